/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 * Include-template: radix-2 NR-DIT NTT exactly as the reference's CPU path.
 *
 * Parameters (macros) supplied by the includer:
 *   NTT_NAME(x)          name mangler
 *   EL_T                 element type (values, in field E)
 *   TW_T                 twiddle type (domain field F ⊂ E)
 *   CTX_T                context type (const CTX_T *ctx passed everywhere)
 *   TW_MUL(ctx,r,a,b)    r = a*b in F            (pointers)
 *   TW_ONE(ctx,r)        r = 1 in F
 *   EL_MULTW(ctx,r,w,x)  r = w*x  (F × E → E)    (IsSubFieldOf::mul, traits.rs:18-26)
 *   EL_ADD(ctx,r,a,b), EL_SUB(ctx,r,a,b)
 *   EL_IS_ZERO(ctx,a)
 *
 * Follows:
 *   math/src/fft/cpu/fft.rs:20-55            in_place_nr_2radix_fft
 *   math/src/fft/cpu/bit_reversing.rs:2-18   in_place_bit_reverse_permute / reverse_index
 *   math/src/fft/cpu/roots_of_unity.rs:13-48 get_powers_of_primitive_root (running product + bit-reverse)
 *   math/src/fft/cpu/ops.rs:13-26            fft = copy + NR radix-2 + bit-reverse
 */

static inline size_t NTT_NAME(reverse_index)(size_t i, u64 size) {
    if (size == 1) return i;
    int tz = __builtin_ctzll(size);
    u64 r = 0, v = (u64)i;
    for (int b = 0; b < 64; b++) { r = (r << 1) | (v & 1); v >>= 1; }
    return (size_t)(r >> (64 - tz));
}

static void NTT_NAME(bitrev_el)(EL_T *x, size_t n) {
    for (size_t i = 0; i < n; i++) {
        size_t j = NTT_NAME(reverse_index)(i, n);
        if (j > i) { EL_T t = x[i]; x[i] = x[j]; x[j] = t; }
    }
}
static void NTT_NAME(bitrev_tw)(TW_T *x, size_t n) {
    for (size_t i = 0; i < n; i++) {
        size_t j = NTT_NAME(reverse_index)(i, n);
        if (j > i) { TW_T t = x[i]; x[i] = x[j]; x[j] = t; }
    }
}

/* roots_of_unity.rs:13-48.  root is ω (or ω^-1) chosen by the caller; bitrev selects BitReverse*. */
static void NTT_NAME(powers)(const CTX_T *ctx, const TW_T *root, size_t count, int bitrev, TW_T *out) {
    if (count == 0) return;
    size_t up_to = count;
    if (bitrev) { up_to = 1; while (up_to < count) up_to <<= 1; }
    TW_T state;
    TW_ONE(ctx, &state);
    for (size_t i = 0; i < up_to; i++) {
        out[i] = state;
        TW_T nx;
        TW_MUL(ctx, &nx, &state, root);
        state = nx;
    }
    if (bitrev) NTT_NAME(bitrev_tw)(out, up_to);
}

/* fft.rs:20-55 */
static void NTT_NAME(nr_2radix)(const CTX_T *ctx, EL_T *input, size_t len, const TW_T *twiddles) {
    size_t group_count = 1;
    size_t group_size = len;
    while (group_count < len) {
        for (size_t group = 0; group < group_count; group++) {
            size_t first_in_group = group * group_size;
            size_t first_in_next_group = first_in_group + group_size / 2;
            const TW_T *w = &twiddles[group];
            for (size_t i = first_in_group; i < first_in_next_group; i++) {
                EL_T wi, y0, y1;
                EL_MULTW(ctx, &wi, w, &input[i + group_size / 2]);
                EL_ADD(ctx, &y0, &input[i], &wi);
                EL_SUB(ctx, &y1, &input[i], &wi);
                input[i] = y0;
                input[i + group_size / 2] = y1;
            }
        }
        group_count *= 2;
        group_size /= 2;
    }
}

/* ops.rs:13-26 (caller has validated power of two) */
static void NTT_NAME(fft)(const CTX_T *ctx, const EL_T *in, size_t len, const TW_T *twiddles, EL_T *out) {
    if (out != in) memcpy(out, in, len * sizeof(EL_T));
    NTT_NAME(nr_2radix)(ctx, out, len, twiddles);
    NTT_NAME(bitrev_el)(out, len);
}
