// Host side of the 256-bit-field NTT: twiddle cache, pass planning, launches.
#include <stdlib.h>
#include <vector>
#include "context.h"
#include "ntt_kernels.cuh"

namespace lw {

// ---- host field helpers (same limb code as the device, compiled for x86) ----
template <class F>
static Fe<F> host_root_of_unity(uint32_t order, bool inverse) {
    // IsFFTField::get_primitive_root_of_unity (math/src/field/traits.rs:82-94)
    if (order == 0) return Fe<F>::one();
    Fe<F> g;
    for (int i = 0; i < F::N; i++) g.v[i] = F::root(i);
    g = fe_to_mont<F>(g);
    for (uint32_t i = 0; i < F::TWO_ADICITY - order; i++) g = fe_sqr<F>(g);
    if (inverse) g = fe_inv<F>(g);
    return g;
}

// lo[i] = base^i (i < 2^hbits), hi[i] = base^(i * 2^hbits) (i < hi_count); internal 32-byte layout
template <class F>
static int upload_power_tables(const Fe<F> &base, uint32_t hbits, uint64_t hi_count, DeviceBuf &lo, DeviceBuf &hi,
                               const Fe<F> *hi_scale = nullptr) {
    const uint64_t lo_count = 1ull << hbits;
    std::vector<uint32_t> hl(lo_count * 8), hh(hi_count * 8);
    Fe<F> acc = Fe<F>::one();
    for (uint64_t i = 0; i < lo_count; i++) {
        for (int k = 0; k < 8; k++) hl[i * 8 + k] = acc.v[k];
        acc = fe_mul<F>(acc, base);
    }
    Fe<F> step = acc;   // base^(2^hbits)
    acc = hi_scale ? *hi_scale : Fe<F>::one();
    for (uint64_t i = 0; i < hi_count; i++) {
        for (int k = 0; k < 8; k++) hh[i * 8 + k] = acc.v[k];
        acc = fe_mul<F>(acc, step);
    }
    if (lo.ensure(lo_count * 32) || hi.ensure(hi_count * 32)) return LW_ERR_ALLOC;
    LW_HIP_CHECK(hipMemcpy(lo.p, hl.data(), lo_count * 32, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipMemcpy(hi.p, hh.data(), hi_count * 32, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
    return LW_OK;
}

template <class F>
static int ensure_twiddles(Context &c, int field, lw_dir_t dir, uint32_t log2n, hipStream_t stream) {
    TwiddleTable &t = c.tw[field][dir];
    if (t.valid && t.log_n >= log2n) return LW_OK;
    if (log2n < 1) return LW_OK;
    // The tables are shared by all lanes: (re)building one needs every other call out of the library (the old table is
    // freed).  The call's shared hold is given up for the scope and taken back afterwards; another lane may have built the
    // table in between.
    ExclusiveScope excl(c);
    if (t.valid && t.log_n >= log2n) return LW_OK;
    // build for at least 2^16 so small transforms never trigger a rebuild storm
    uint32_t L = log2n < 16 ? 16 : log2n;
    if (L > F::TWO_ADICITY) L = log2n;
    const uint32_t bits = L - 1;
    const uint64_t count = 1ull << bits;
    if (t.buf.ensure(count * 32)) return LW_ERR_ALLOC;
    const uint32_t hbits = (bits + 1) / 2;
    Fe<F> w = host_root_of_unity<F>(L, dir == LW_DIR_INVERSE);
    DeviceBuf lo, hi;
    int rc = upload_power_tables<F>(w, hbits, 1ull << (bits - hbits), lo, hi);
    if (rc) { lo.release(); hi.release(); return rc; }
    const uint32_t threads = 256;
    const uint64_t blocks = (count + threads - 1) / threads;
    hipLaunchKernelGGL((twiddle_fill_kernel<F>), dim3((uint32_t)blocks), dim3(threads), 0, stream, (uint4 *)t.buf.p,
                       (const uint4 *)lo.p, (const uint4 *)hi.p, bits, hbits, count);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipStreamSynchronize(stream), LW_ERR_LAUNCH);
    lo.release();
    hi.release();
    t.log_n = L;
    t.valid = true;
    return LW_OK;
}

// ---- pass planning ----
struct NttPlan {
    int npass;
    uint32_t s0[8], r[8], logC[8];
};

static uint32_t g_ntt_dbg = 0;     // diagnostics: see NttPassParams::dbg (results are wrong when set)
void ntt_set_debug(uint32_t d) { g_ntt_dbg = d; }
uint32_t ntt_get_debug() { return g_ntt_dbg; }
static int cfg_tile_log() { return NttCfgA::TILE_LOG; }
static int cfg_kmax() { return NttCfgA::KMAX; }

// stages [skip, L): the first `skip` stages of a zero-padded input only replicate it (see ntt256_run)
static NttPlan plan_passes(uint32_t L, uint32_t max_r, uint32_t skip = 0) {
    const uint32_t NTT_TILE_LOG = (uint32_t)cfg_tile_log();
    NttPlan pl{};
    const uint32_t Ls = L - skip;
    pl.npass = (int)((Ls + max_r - 1) / max_r);
    if (pl.npass < 1) pl.npass = 1;
    uint32_t rr[8];
    uint32_t base = Ls / pl.npass, extra = Ls % pl.npass;
    for (int i = 0; i < pl.npass; i++) rr[i] = base + ((uint32_t)i < extra ? 1 : 0);
    // From 2^16 up: the last pass (per-element twiddles, bit-reversed stores, wave-local exchanges) takes a full max_r
    // stages and the others share the rest as evenly as possible in EVEN sizes — an odd pass ends in a radix-2 register
    // step with two items per thread.  Measured (tools/ab_ntt_plan.py): 2^20 (7,7,6) 0.1056 -> (6,6,8) 0.1030 ms,
    // 2^22 (8,7,7) 0.3588 -> (6,8,8) 0.3533, 2^26 (7,7,6,6) 6.45 -> (6,6,6,8) 6.19 ms; a short LAST pass is the worst
    // choice (2^26 (8,8,8,2): 9.1 ms).
    if (pl.npass >= 2 && Ls >= 16 && Ls > max_r) {
        const int q = pl.npass - 1;
        rr[q] = max_r;
        const uint32_t R = Ls - max_r;
        base = R / q;
        extra = R % q;
        for (int i = 0; i < q; i++) rr[i] = base + ((uint32_t)i >= (uint32_t)q - extra ? 1 : 0);
        for (int i = 0; i + 1 < q; i++)
            if ((rr[i] & 1) && (rr[i + 1] & 1) && rr[i + 1] < max_r && rr[i] > 1) {
                rr[i]--;
                rr[i + 1]++;
            }
    }
    // tuning only: LW_HIP_NTT_PLAN="8,8,6" fixes the stages per pass for transforms whose stage count matches the sum
    static const char *plan_env = tuning_env("LW_HIP_NTT_PLAN");
    if (plan_env) {
        uint32_t v[8], cnt = 0, sum = 0;
        for (const char *q = plan_env; *q && cnt < 8;) {
            v[cnt] = (uint32_t)atoi(q);
            sum += v[cnt++];
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        bool ok = sum == Ls && cnt >= 1;
        for (uint32_t i = 0; i < cnt && ok; i++) ok = v[i] >= 1 && v[i] <= max_r;
        if (ok) {
            pl.npass = (int)cnt;
            for (uint32_t i = 0; i < cnt; i++) rr[i] = v[i];
        }
    }
    uint32_t s = skip;
    for (int i = 0; i < pl.npass; i++) {
        const uint32_t r = rr[i];
        pl.s0[i] = s;
        pl.r[i] = r;
        uint32_t room = NTT_TILE_LOG - r;
        uint32_t avail = L - s - r;   // non-last: log2 of the row stride; last: 0 unless multi-pass
        if (i == pl.npass - 1) avail = L - r;
        pl.logC[i] = room < avail ? room : avail;
        s += r;
    }
    return pl;
}

static void split_steps(uint32_t r, NttPassParams &p) {
    const uint32_t NTT_KMAX = (uint32_t)cfg_kmax();
    uint32_t nsteps = (r + NTT_KMAX - 1) / NTT_KMAX, left = r;
    p.nsteps = nsteps;
    for (uint32_t i = 0; i < nsteps; i++) {
        uint32_t k = (left + (nsteps - i) - 1) / (nsteps - i);
        p.k[i] = k;
        left -= k;
    }
}

static bool g_ntt_wave_local_all = [] { const char *e = tuning_env("LW_HIP_NTT_WAVE_LOCAL"); return e && atoi(e) == 2; }();   // A/B: also in non-last passes
static bool g_ntt_wave_local = [] { const char *e = tuning_env("LW_HIP_NTT_WAVE_LOCAL"); return !e || atoi(e) != 0; }();   // A/B switch
static uint32_t g_ntt_max_r = 8;
void ntt_set_max_pass_stages(uint32_t r) { g_ntt_max_r = r < 1 ? 1 : (r > 8 ? 8 : r); }

template <class F, class CFG>
static void launch_pass(bool last, dim3 grid, hipStream_t stream, const NttPassParams &p) {
    const bool extra = p.cos_in || p.cos_out || p.scale;
    // Column layout with wave-local exchanges (ntt_kernels.cuh lds_slot): full-size tiles only — every register step has
    // exactly one work-item per thread, so a column's items sit in the same wavefront in every step with the same k.
    // The exchange before step s stays inside the wave when steps s-1 and s both walk rows fastest with the same k
    // (last pass: from step 1; other passes: from step 2, their first step walks columns fastest for coalesced loads).
    // Used by the LAST pass only: there it removes all three inter-step barriers and makes the per-lane twiddle fetches of
    // a wave consecutive table entries (0.537 -> 0.506 ms at 2^24).  In the other passes rows-fastest work-items read 64
    // different staged twiddles per wave where columns-fastest ones read 8 (broadcast over the columns), which costs more
    // LDS bandwidth than the two barriers it saves (0.437 -> 0.461 ms); fetching them from the table instead, as the last
    // pass does, is worse still (0.440 -> 0.478 ms, LW_HIP_NTT_WAVE_LOCAL=2): they keep the plain layout.
    bool wl = g_ntt_wave_local && (last || g_ntt_wave_local_all) && p.r >= 6;
    uint32_t ws = 0;
    for (uint32_t st = 0; st < p.nsteps && wl; st++)
        if ((1u << (p.r + p.logC - p.k[st])) != (uint32_t)CFG::THREADS) wl = false;
    if (wl)
        for (uint32_t st = (last ? 1 : 2); st < p.nsteps; st++)
            if (p.k[st] == p.k[st - 1] && (1u << (p.r - p.k[st])) <= 64u) ws |= 1u << st;
    NttPassParams q = p;
    q.wave_sync = ws;
#define LW_LAUNCH_PASS(LASTV, EXTRAV)                                                                                          \
    do {                                                                                                                       \
        if (fx == 8 && wl) hipLaunchKernelGGL((ntt_pass_kernel<F, LASTV, CFG, EXTRAV, true, FXOK ? 8 : 0>), grid, dim3(CFG::THREADS), 0, stream, q);   \
        else if (fx == 8) hipLaunchKernelGGL((ntt_pass_kernel<F, LASTV, CFG, EXTRAV, false, FXOK ? 8 : 0>), grid, dim3(CFG::THREADS), 0, stream, q);   \
        else if (fx == 6 && !wl) hipLaunchKernelGGL((ntt_pass_kernel<F, LASTV, CFG, EXTRAV, false, FXOK ? 6 : 0>), grid, dim3(CFG::THREADS), 0, stream, q);   \
        else if (fx == 7 && !wl) hipLaunchKernelGGL((ntt_pass_kernel<F, LASTV, CFG, EXTRAV, false, FXOK ? 7 : 0>), grid, dim3(CFG::THREADS), 0, stream, q);   \
        else if (wl) hipLaunchKernelGGL((ntt_pass_kernel<F, LASTV, CFG, EXTRAV, true>), grid, dim3(CFG::THREADS), 0, stream, q);  \
        else hipLaunchKernelGGL((ntt_pass_kernel<F, LASTV, CFG, EXTRAV, false>), grid, dim3(CFG::THREADS), 0, stream, q);      \
    } while (0)
    // full-size tiles (every pass of a 2^24 transform, the last pass from 2^16 on, the 6-stage passes of 2^20 and 2^26):
    // kernels with the tile shape compiled in
    constexpr bool FXOK = CFG::TILE == 2048 && CFG::THREADS == 512;
    static const bool fx_env = [] { const char *e = tuning_env("LW_HIP_NTT_FX"); return !e || atoi(e) != 0; }();   // A/B only
    int fx = 0;
    if (FXOK && fx_env && p.r == 8 && p.logC == 3 && p.nsteps == 4 && p.k[0] == 2 && p.k[1] == 2 && p.k[2] == 2 && p.k[3] == 2) fx = 8;
    if (FXOK && fx_env && p.r == 7 && p.logC == 4 && p.nsteps == 4 && p.k[0] == 2 && p.k[1] == 2 && p.k[2] == 2 && p.k[3] == 1) fx = 7;   // (7,7,8)
    if (FXOK && fx_env && p.r == 6 && p.logC == 5 && p.nsteps == 3 && p.k[0] == 2 && p.k[1] == 2 && p.k[2] == 2) fx = 6;   // (6,6,8), (6,6,6,8)
    if (last) {
        if (extra) LW_LAUNCH_PASS(true, true);
        else LW_LAUNCH_PASS(true, false);
    } else {
        if (extra) LW_LAUNCH_PASS(false, true);
        else LW_LAUNCH_PASS(false, false);
    }
#undef LW_LAUNCH_PASS
}

// cached two-level power tables of `base` (optionally inverted): base^e = lo[e & mask] * hi[e >> hbits]
template <class F>
static int power_tables(Context &c, int field, int slot, const uint32_t *base_words, bool invert, uint32_t hbits,
                        uint32_t hi_bits, hipStream_t stream, const uint4 **lo, const uint4 **hi, bool fold_ninv = false) {
    CosetCache &cc = c.coset[slot];
    bool hit = cc.valid && cc.field == field && cc.hbits == hbits && cc.hi_bits == hi_bits && cc.inverse == invert &&
               cc.fold_ninv == fold_ninv;
    for (int i = 0; i < 8 && hit; i++) hit = cc.words[i] == base_words[i];
    if (!hit) {
        Fe<F> b;
        for (int i = 0; i < 8; i++) b.v[i] = base_words[i];
        if (b.is_zero()) {
            set_error("coset offset is zero");
            return LW_ERR_INV_ZERO;
        }
        if (invert) b = fe_inv<F>(b);
        // The tables are rebuilt by a kernel on the call's stream, behind their previous users (calls of one lane are
        // ordered across streams by Entry); growing a buffer frees the old one, which the runtime does synchronously.
        const uint64_t lo_count = 1ull << hbits, hi_count = 1ull << hi_bits;
        if (cc.lo.ensure(lo_count * 32) || cc.hi.ensure(hi_count * 32)) return LW_ERR_ALLOC;
        FeWords8 bw{}, sw{};
        for (int i = 0; i < 8; i++) bw.w[i] = b.v[i];
        if (fold_ninv) {   // N^-1 folded into the high table
            const Fe<F> ninv = fe_inv<F>(fe_from_u64<F>(1ull << (hbits + hi_bits)));
            for (int i = 0; i < 8; i++) sw.w[i] = ninv.v[i];
        }
        hipLaunchKernelGGL((power_tables_kernel<F>), dim3((uint32_t)((lo_count + hi_count + 255) / 256)), dim3(256), 0, stream, (uint4 *)cc.lo.p,
                           (uint4 *)cc.hi.p, bw, hbits, hi_count, sw, fold_ninv ? 1 : 0);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        cc.valid = true;
        cc.fold_ninv = fold_ninv;
        cc.field = field;
        cc.hbits = hbits;
        cc.hi_bits = hi_bits;
        cc.inverse = invert;
        for (int i = 0; i < 8; i++) cc.words[i] = base_words[i];
    }
    *lo = (const uint4 *)cc.lo.p;
    *hi = (const uint4 *)cc.hi.p;
    return LW_OK;
}

template <class F>
static int ntt256_run(Context &c, int field, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2n,
                      uint32_t batch, uint64_t stride, const uint32_t *coset_words, hipStream_t stream, uint32_t in_log2) {
    const uint64_t n = 1ull << log2n;
    if (stride == 0) stride = n;
    // Low-degree extension (evaluate_fft with blowup / domain_size, math/src/fft/polynomial.rs:30-38): the input has
    // 2^in_log2 coefficients and the rest of the 2^log2n vector is zero padding.  In the NR-DIT dataflow the first
    // `skip` = log2n - in_log2 stages pair every element with a zero, (a + w*0, a - w*0) = (a, a): they only replicate
    // the coefficient block.  So those stages are skipped and the first pass reads element g from in[g mod 2^in_log2]:
    // skip/log2n of the products and all reads of the padding are saved (blow-up 8 at 2^24: 12.5 %).
    uint32_t skip = 0;
    if (dir == LW_DIR_FORWARD && in_log2 >= 1 && in_log2 < log2n) skip = log2n - in_log2;
    const uint64_t in_mask = skip ? ((1ull << in_log2) - 1) : ~0ull;
    const uint64_t in_stride_default = skip ? (1ull << in_log2) : n;
    if (log2n == 0) {   // N = 1: the transform (and N^-1 = 1, h^0 = 1) is the identity
        if (d_in != d_out)
            LW_HIP_CHECK(hipMemcpy2DAsync(d_out, stride * 32, d_in, stride * 32, 32, batch, hipMemcpyDeviceToDevice, stream),
                         LW_ERR_LAUNCH);
        return LW_OK;
    }
    int rc = ensure_twiddles<F>(c, field, dir, log2n, stream);
    if (rc) return rc;
    const uint4 *tw = (const uint4 *)c.tw[field][dir].buf.p;

    NttPlan pl = plan_passes(log2n, g_ntt_max_r, skip);
    const bool need_scratch = pl.npass > 1 || d_in == d_out;
    if (need_scratch && c.scratch.ensure((size_t)n * batch * 32)) return LW_ERR_ALLOC;
    c.timings.scratch_bytes = c.scratch.bytes;

    const void *src = d_in;
    uint64_t src_stride = skip ? in_stride_default : stride;
    const uint4 *cos_lo = nullptr, *cos_hi = nullptr;
    const uint32_t cos_hbits = (log2n + 1) / 2;
    if (coset_words) {   // coset scaling is fused into the first pass's load / the last pass's store
        const bool inv = dir == LW_DIR_INVERSE;
        rc = power_tables<F>(c, field, inv ? 1 : 0, coset_words, inv, cos_hbits, log2n - cos_hbits, stream, &cos_lo, &cos_hi, inv);
        if (rc) return rc;
    }

    if (pl.npass == 1 && src == d_out && !skip) {
        // single pass on aliased buffers: the last pass permutes across tiles, so stage the input first
        LW_HIP_CHECK(hipMemcpy2DAsync(c.scratch.p, n * 32, d_in, stride * 32, n * 32, batch, hipMemcpyDeviceToDevice, stream),
                     LW_ERR_LAUNCH);
        src = c.scratch.p;
        src_stride = n;
    }

    for (int i = 0; i < pl.npass; i++) {
        const bool last = (i == pl.npass - 1);
        NttPassParams p{};
        p.tw = tw;
        p.dbg = g_ntt_dbg;
        p.lazy_in = (F::LAZY && i > 0) ? 1 : 0;
        p.in_mask = i == 0 ? in_mask : ~0ull;
        p.cos_lo = cos_lo;
        p.cos_hi = cos_hi;
        p.cos_hbits = cos_hbits;
        p.cos_in = (coset_words && dir == LW_DIR_FORWARD && i == 0) ? 1 : 0;
        p.cos_out = (coset_words && dir == LW_DIR_INVERSE && last) ? 1 : 0;
        p.L = log2n;
        p.s0 = pl.s0[i];
        p.r = pl.r[i];
        p.logC = pl.logC[i];
        if (p.r > 8) {   // ltw[2][256] (staged twiddles) and the 17p lazy bound both need r <= 8 (ntt_kernels.cuh)
            set_error("internal: NTT pass of %u stages", p.r);
            return LW_ERR_BAD_ARG;
        }
        split_steps(p.r, p);
        p.in = (const uint4 *)src;
        p.in_batch_stride = src_stride;
        if (last) {
            if (src == d_out) {
                set_error("internal: last NTT pass would run in place");
                return LW_ERR_BAD_ARG;
            }
            p.out = (uint4 *)d_out;
            p.out_batch_stride = stride;
            if (dir == LW_DIR_INVERSE) {
                Fe<F> ninv = fe_inv<F>(fe_from_u64<F>(n));   // FieldElement::from(len as u64).inv()
                p.scale = 1;
                for (int k = 0; k < 8; k++) p.sc[k] = ninv.v[k];
            }
        } else {
            p.out = (uint4 *)c.scratch.p;
            p.out_batch_stride = n;
        }
        const uint32_t blocks = 1u << (log2n - p.r - p.logC);
        dim3 grid(blocks, batch);
        hipEvent_t pe = c.prof_begin(stream);
        launch_pass<F, NttCfgA>(last, grid, stream, p);
        c.prof_end(last ? "ntt_pass_kernel<last>" : "ntt_pass_kernel", pe, stream);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        src = p.out;
        src_stride = p.out_batch_stride;
    }

    return LW_OK;
}

// ---- helpers for the multi-GPU cross step (ntt_cross.hip)
int ntt256_power_tables(Context &c, int field, int slot, const uint32_t *base_words, bool invert, uint32_t hbits,
                        uint32_t hi_bits, hipStream_t stream, const uint4 **lo, const uint4 **hi) {
    if (field == LW_FIELD_STARK252) return power_tables<Stark252>(c, field, slot, base_words, invert, hbits, hi_bits, stream, lo, hi);
    return power_tables<Fr381>(c, field, slot, base_words, invert, hbits, hi_bits, stream, lo, hi);
}
int ntt256_root_words(int field, uint32_t order, bool inverse, uint32_t *words) {
    if (field == LW_FIELD_STARK252) {
        Fe<Stark252> w = host_root_of_unity<Stark252>(order, inverse);
        for (int i = 0; i < 8; i++) words[i] = w.v[i];
    } else {
        Fe<Fr381> w = host_root_of_unity<Fr381>(order, inverse);
        for (int i = 0; i < 8; i++) words[i] = w.v[i];
    }
    return LW_OK;
}
int ntt256_inv_u64_words(int field, uint64_t v, uint32_t *words) {
    if (field == LW_FIELD_STARK252) {
        Fe<Stark252> w = fe_inv<Stark252>(fe_from_u64<Stark252>(v));
        for (int i = 0; i < 8; i++) words[i] = w.v[i];
    } else {
        Fe<Fr381> w = fe_inv<Fr381>(fe_from_u64<Fr381>(v));
        for (int i = 0; i < 8; i++) words[i] = w.v[i];
    }
    return LW_OK;
}
const uint4 *ntt256_twiddle_table(Context &c, int field, lw_dir_t dir, uint32_t log2n, hipStream_t stream, int *rc) {
    *rc = field == LW_FIELD_STARK252 ? ensure_twiddles<Stark252>(c, field, dir, log2n, stream)
                                     : ensure_twiddles<Fr381>(c, field, dir, log2n, stream);
    return (const uint4 *)c.tw[field][dir].buf.p;
}

// get_powers_of_primitive_root[_coset] (math/src/fft/cpu/roots_of_unity.rs:13-61) on the device: out[i] = scale * w^e(i),
// e(i) = i or bitrev_bits(i), w the primitive 2^order-th root or its inverse; reference memory layout.
template <class F>
__global__ void powers_export_kernel(uint4 *out, const uint4 *lo, const uint4 *hi, uint32_t bitrev, uint32_t hbits, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t e = bitrev ? bitrev_bits((uint32_t)i, bitrev) : i;
    const Fe<F> x = fe_mul<F>(tw_load<F>(lo, e & ((1ull << hbits) - 1)), tw_load<F>(hi, e >> hbits));
    uint4 q0, q1;
    pack_mem<F>(x, q0, q1);
    out[2 * i] = q0;
    out[2 * i + 1] = q1;
}
template <class F>
static int gen_powers_t(uint32_t order, uint64_t count, uint32_t bitrev, bool inverse, const uint32_t *scale_words, void *d_out,
                        hipStream_t stream) {
    uint32_t bits = 0;
    while ((1ull << bits) < count) bits++;
    const uint32_t hbits = (bits + 1) / 2;
    Fe<F> w = host_root_of_unity<F>(order, inverse), sc;
    if (scale_words)
        for (int i = 0; i < 8; i++) sc.v[i] = scale_words[i];
    DeviceBuf lo, hi;
    int rc = upload_power_tables<F>(w, hbits, 1ull << (bits - hbits), lo, hi, scale_words ? &sc : nullptr);
    if (rc == LW_OK) {
        hipLaunchKernelGGL((powers_export_kernel<F>), dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, stream, (uint4 *)d_out,
                           (const uint4 *)lo.p, (const uint4 *)hi.p, bitrev, hbits, count);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            set_error("power table kernel failed");
            rc = LW_ERR_LAUNCH;
        }
    }
    lo.release();
    hi.release();
    return rc;
}
int ntt256_gen_powers(int field, uint32_t order, uint64_t count, uint32_t bitrev, bool inverse, const uint32_t *scale_words, void *d_out,
                      hipStream_t stream) {
    if (field == LW_FIELD_STARK252) return gen_powers_t<Stark252>(order, count, bitrev, inverse, scale_words, d_out, stream);
    return gen_powers_t<Fr381>(order, count, bitrev, inverse, scale_words, d_out, stream);
}

// in_log2 < log2n: d_in holds `batch` blocks of 2^in_log2 coefficients (dense), d_out 2^log2n evaluations each
int ntt256_device(Context &c, int field, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2n, uint32_t batch,
                  uint64_t stride, const uint32_t *coset_words, hipStream_t stream, uint32_t in_log2) {
    if (field == LW_FIELD_STARK252)
        return ntt256_run<Stark252>(c, field, dir, d_in, d_out, log2n, batch, stride, coset_words, stream, in_log2);
    return ntt256_run<Fr381>(c, field, dir, d_in, d_out, log2n, batch, stride, coset_words, stream, in_log2);
}

}  // namespace lw
