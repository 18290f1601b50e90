"""CPU-side hygiene (no GPU sanitizers on this pool): the oracle and the product's host-side limb code are
rebuilt with AddressSanitizer + UBSan and driven through a small NTT / MSM / field workload in a child process."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = tmp_path / "asan_oracle"
    subprocess.check_call(["gcc", "-O0", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=gnu11",
                           "-o", str(exe), os.path.join(ROOT, "tests", "asan_oracle_main.c"),
                           os.path.join(ROOT, "oracle", "lw_oracle.c"), "-lpthread"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stderr[-2000:])


def test_product_host_limb_code_under_ubsan(tmp_path):
    """field.cuh / ec.cuh compile for the host (twiddle seeds, N^-1, the MSM window fold run there): exercise them
    with UBSan + ASan against Python big integers."""
    src = tmp_path / "t.cpp"
    src.write_text(textwrap.dedent(f"""
        #include <cstdio>
        #include <initializer_list>
        struct uint4 {{ unsigned x, y, z, w; }};
        static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) {{ return uint4{{a, b, c, d}}; }}
        #include "{ROOT}/lambda_elliptic_curves_amd/csrc/ec.cuh"
        using namespace lw;
        template <class F> void run(const char *n) {{
            Fe<F> a = fe_from_u64<F>(123456789ull), b = fe_from_u64<F>(987654321ull);
            Fe<F> c = fe_mul<F>(a, fe_inv<F>(b));
            c = fe_mul<F>(c, b);
            printf("%s %d\\n", n, (int)(c == a));
        }}
        // the 64-bit host product (unsigned __int128 CIOS) against the 32-bit reference loop on pseudo-random residues
        template <class F> int cmp64() {{
            unsigned long long st = 0x9E3779B97F4A7C15ull;
            int bad = 0;
            for (int it = 0; it < 3000; it++) {{
                Fe<F> a, b;
                for (int i = 0; i < F::N; i++) {{
                    st = st * 6364136223846793005ull + 1442695040888963407ull; a.v[i] = (unsigned)(st >> 32);
                    st = st * 6364136223846793005ull + 1442695040888963407ull; b.v[i] = (unsigned)(st >> 32);
                }}
                a.v[F::N - 1] %= F::p(F::N - 1);   // top limb below the modulus' top limb: a, b < p
                b.v[F::N - 1] %= F::p(F::N - 1);
                if (it == 0) {{ for (int i = 0; i < F::N; i++) {{ a.v[i] = F::p(i); b.v[i] = F::p(i); }} a.v[0]--; b.v[0]--; }}   // (p-1)^2
                bad += !(fe_mul<F>(a, b) == fe_mul_portable<F>(a, b));
            }}
            return bad;
        }}
        // the bounded-binary-GCD inverse against the Fermat inverse
        template <class F> int cmpinv() {{
            unsigned long long st = 0x243F6A8885A308D3ull;
            int bad = 0;
            for (int it = 0; it < 400; it++) {{
                Fe<F> a;
                for (int i = 0; i < F::N; i++) {{ st = st * 6364136223846793005ull + 1442695040888963407ull; a.v[i] = (unsigned)(st >> 32); }}
                a.v[F::N - 1] %= F::p(F::N - 1);
                if (it < 40) for (int i = 1 + it % F::N; i < F::N; i++) a.v[i] = 0;          // short values
                if (it == 40) {{ for (int i = 0; i < F::N; i++) a.v[i] = F::p(i); a.v[0]--; }}  // p - 1
                if (it == 41) {{ for (int i = 0; i < F::N; i++) a.v[i] = 0; a.v[0] = 1; }}      // 1
                if (it == 42) {{ for (int i = 0; i < F::N; i++) a.v[i] = 0; }}                  // 0 -> 0
                if (a.is_zero() && it != 42) a.v[0] = 5;
                bad += !(fe_inv_fast<F>(a) == fe_inv<F>(a));
            }}
            return bad;
        }}
        int main() {{
            printf("inv %d\\n", cmpinv<Stark252>() + cmpinv<Fr381>() + cmpinv<Fp381>() + cmpinv<Fp254>() + cmpinv<Fr254>());
            printf("mul64 %d\\n", cmp64<Stark252>() + cmp64<Fr381>() + cmp64<Fp381>() + cmp64<Fp254>() + cmp64<Fr254>());
            run<Stark252>("stark"); run<Fr381>("fr381"); run<Fp381>("fp381"); run<Fp254>("fp254");
            Point<Bls12381G1> id = pt_identity<Bls12381G1>();
            Point<Bls12381G1> s = pt_add<Bls12381G1>(id, pt_dbl<Bls12381G1>(id));
            printf("id %d\\n", (int)pt_is_identity<Bls12381G1>(s));
            printf("bb %u\\n", bb_mul(bb_inv(bb_mul(5, BabyBear::R2)), bb_mul(5, BabyBear::R2)) == BabyBear::ONE);
            Fe<Stark252> big; for (int i = 0; i < 8; i++) big.v[i] = 0xffffffffu;
            Fe<Stark252> r = fe_reduce_full(big);
            printf("rf %d\\n", (int)(reduce_once<Stark252>(r) == r));
            return 0;
        }}
    """))
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["inv", "0", "mul64", "0", "stark", "1", "fr381", "1", "fp381", "1", "fp254", "1", "id", "1", "bb", "1", "rf", "1"], out.stdout
