/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 * Keccak-256 as used by the reference through the third-party `sha3` crate (crypto/Cargo.toml:12 `sha3 = "0.10"`,
 * provers/stark/Cargo.toml:20 `sha3 = "0.10.6"`; not vendored under /root/reference).  Restated from the published
 * algorithm: Keccak-f[1600], rate 1088 bits (136 bytes), capacity 512, original Keccak padding 0x01 ... 0x80
 * (NOT the SHA-3 0x06 domain byte), 32-byte output.  Pinned in tests by the Keccak team's known answers
 * (Keccak-256("") and Keccak-256("abc")) and a multi-block message cross-checked against Python's hashlib.sha3_256
 * with the padding byte swapped.
 * Call sites anchored: crypto/src/merkle_tree/backends/field_element_vector.rs:41-58 (leaf = H(concat as_bytes),
 * parent = H(left || right)), crypto/src/merkle_tree/utils.rs:44-72 (level order), merkle.rs:31-56.
 */
#ifndef ORC_KECCAK_H
#define ORC_KECCAK_H
#include <stdint.h>
#include <string.h>

static const uint64_t ORC_KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ORC_KECCAK_ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int ORC_KECCAK_PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static void orc_keccak_f1600(uint64_t st[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = ORC_KECCAK_PIL[i];
            uint64_t b = st[j];
            st[j] = (t << ORC_KECCAK_ROT[i]) | (t >> (64 - ORC_KECCAK_ROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= ORC_KECCAK_RC[round];
    }
}

static void orc_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    uint64_t st[25];
    memset(st, 0, sizeof(st));
    uint8_t block[136];
    while (len >= 136) {
        for (int i = 0; i < 17; i++) {
            uint64_t w;
            memcpy(&w, data + 8 * i, 8);   /* little-endian host */
            st[i] ^= w;
        }
        orc_keccak_f1600(st);
        data += 136;
        len -= 136;
    }
    memset(block, 0, 136);
    memcpy(block, data, len);
    block[len] ^= 0x01;
    block[135] ^= 0x80;
    for (int i = 0; i < 17; i++) {
        uint64_t w;
        memcpy(&w, block + 8 * i, 8);
        st[i] ^= w;
    }
    orc_keccak_f1600(st);
    memcpy(out, st, 32);
}
#endif
