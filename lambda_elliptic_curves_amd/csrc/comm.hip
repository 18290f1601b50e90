// Multi-GPU sharding of the hot path behind the C ABI: one process per GPU, a library-owned RCCL
// communicator over xGMI, no torch types.  The reference has no multi-device code at all; what these entry
// points return is exactly what the single-device entry points return on the concatenated input.
//
//   * lw_hip_ntt_sharded_device   one 2^L-point transform (or `batch` of them) block-distributed over G = 2/4/8 ranks,
//                                 M = N/G elements each.  Bailey four-step with N1 = G:
//                                   A  all-to-all   slice h of my block -> rank h            (RCCL send/recv group)
//                                   B  cross step   G-point transform across the received chunks + w_N^(j2 k1)  (ntt_cross.hip)
//                                   C  all-to-all   row k1 -> rank k1
//                                   D  local M-point NTT  -> X[g + G*k2]                     (the single-GPU pass kernels)
//                                   E  all-to-all + F local interleave -> my block of the natural-order result (optional)
//                                 xGMI is a full mesh, so each all-to-all drives all G-1 links of a GPU at once; the payload
//                                 per rank and exchange is (G-1)/G of the local shard.
//   * lw_hip_msm_sharded_device   points/scalars sharded; every rank accumulates its pairs into the full bucket array, an
//                                 all-to-all gives rank g bucket range g of every window from everyone, rank g adds the G
//                                 contributions and runs the running sums over its slice, and an all-gather of the
//                                 per-slice (S, A) pairs lets every rank fold the result: the north star's "bucket
//                                 all-reduce" as reduce-scatter (all-to-all + local group additions, RCCL has no
//                                 user-defined reduction) + all-gather (second form) — msm_sharded_run below.
//
// The exchange schedule is written once against a small transport interface with two implementations: RCCL (one rank
// per process) and an in-process simulator that walks all G virtual ranks on ONE device with device-to-device copies
// (lw_hip_ntt_sharded_selftest_device) — that is how the index arithmetic below is parity-tested on a one-GPU box.
// librccl is opened with dlopen at lw_hip_comm_init, so the library still loads on hosts without RCCL; every failure on
// this path is reported as LW_ERR_COMM.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <vector>
#include "context.h"

namespace lw {

int ntt_device_locked(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                      uint32_t log2n, uint32_t batch, size_t stride, const void *coset, hipStream_t stream, uint32_t in_log2);
int ntt_cross_device(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                     uint32_t log2_total, uint32_t log2_g, uint64_t j2_begin, uint64_t slice_len, uint64_t chunk_stride,
                     uint32_t batch, uint64_t batch_stride, hipStream_t stream);
int msm_device(Context &c, lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, void *out_host,
               hipStream_t stream, int scalars_montgomery, int affine_points, const void *h_points = nullptr);
int msm_sum_points_host(lw_curve_t curve, const void *pts, size_t n, void *out);   // msm.hip
uint32_t msm_window_bits_for(size_t n);                                              // msm.hip: the single-GPU window rule
int msm_shard_accumulate(Context &c, lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, uint32_t cbits, hipStream_t s,
                         char **buckets);
int msm_shard_reduce(Context &c, lw_curve_t curve, const char *recv, uint32_t G, uint32_t cbits, char *d_sa, hipStream_t s);
int msm_shard_combine(lw_curve_t curve, const char *sa_all, uint32_t G, uint32_t cbits, void *out);
uint32_t field_two_adicity(lw_field_t f);                                          // api.hip
int check_field_layout(lw_field_t field, lw_layout_t layout);                      // api.hip

// ---------------------------------------------------------------- RCCL, loaded on demand
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load() {
    if (g_rccl.handle) return LW_OK;
    // A process that already carries an RCCL (e.g. torch's) gets that one back: dlopen matches loaded objects by SONAME.
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        set_error("cannot load librccl (%s); multi-GPU entry points are unavailable", dlerror());
        return LW_ERR_COMM;
    }
    Rccl r;
    r.handle = h;
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) ok = false; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        set_error("librccl lacks a required symbol");
        dlclose(h);
        return LW_ERR_COMM;
    }
    g_rccl = r;
    return LW_OK;
}

#define LW_NCCL_CHECK(expr)                                                                                  \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) {                                                                             \
            ::lw::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);  \
            return LW_ERR_COMM;                                                                              \
        }                                                                                                    \
    } while (0)

struct CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 0;
};
static CommState g_comm;

void comm_release(Context &) {
    if (g_comm.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_comm.comm);
    g_comm = CommState{};
}

// ---------------------------------------------------------------- transports
// all_to_all: for every local rank g, batch b and peer h, chunk (b, h) of g's send buffer becomes chunk (b, g) of h's
// receive buffer.  Chunks are `chunk_bytes` long; consecutive b are `bstride_bytes` apart.
struct Transport {
    Context *cx = nullptr;       // the running call's lane (lane 0: the communicator's buffers live there)
    int G = 1;
    int first = 0, nlocal = 1;   // ranks [first, first + nlocal) live in this process
    virtual int all_to_all(const char *const *send, char *const *recv, size_t chunk_bytes, uint32_t batch, size_t bstride_bytes,
                           hipStream_t s) = 0;
    // all_gather: recv[h] of every local rank h receives, for every rank g, `bytes` bytes from g's send[g] at offset g * bytes
    virtual int all_gather(const char *const *send, char *const *recv, size_t bytes, hipStream_t s) = 0;
    // collective agreement on a local status: LW_OK only if every rank passed LW_OK, else LW_ERR_COMM on all of them
    // (the failing rank reports its own code); synchronises `s`
    virtual int agree(int local_status, hipStream_t s) = 0;
    virtual ~Transport() {}
};

struct RcclTransport : Transport {
    explicit RcclTransport(Context &c) { cx = &c; G = g_comm.nranks; first = g_comm.rank; nlocal = 1; }
    int all_to_all(const char *const *send, char *const *recv, size_t chunk_bytes, uint32_t batch, size_t bstride_bytes,
                   hipStream_t s) override {
        LW_NCCL_CHECK(g_rccl.GroupStart());
        for (uint32_t b = 0; b < batch; b++)
            for (int h = 0; h < G; h++) {
                ncclResult_t r1 = g_rccl.Send(send[0] + b * bstride_bytes + (size_t)h * chunk_bytes, chunk_bytes, ncclChar, h, g_comm.comm, s);
                ncclResult_t r2 = g_rccl.Recv(recv[0] + b * bstride_bytes + (size_t)h * chunk_bytes, chunk_bytes, ncclChar, h, g_comm.comm, s);
                if (r1 != ncclSuccess || r2 != ncclSuccess) {
                    (void)g_rccl.GroupEnd();
                    set_error("ncclSend/ncclRecv to rank %d failed: %s", h, g_rccl.GetErrorString(r1 != ncclSuccess ? r1 : r2));
                    return LW_ERR_COMM;
                }
            }
        LW_NCCL_CHECK(g_rccl.GroupEnd());
        return LW_OK;
    }
    int all_gather(const char *const *send, char *const *recv, size_t bytes, hipStream_t s) override {
        LW_NCCL_CHECK(g_rccl.AllGather(send[0], recv[0], bytes, ncclChar, g_comm.comm, s));
        return LW_OK;
    }
    int agree(int local_status, hipStream_t s) override {
        Context &c = *cx;
        if (c.small.ensure(8 * (size_t)(G + 1))) return LW_ERR_ALLOC;   // (an allocation of 72 bytes: if this fails nothing works)
        int64_t mine = local_status, all[9] = {0};
        int64_t *d = (int64_t *)c.small.p;
        LW_HIP_CHECK(hipMemcpyAsync(d, &mine, 8, hipMemcpyHostToDevice, s), LW_ERR_LAUNCH);
        LW_NCCL_CHECK(g_rccl.AllGather(d, d + 1, 8, ncclChar, g_comm.comm, s));
        LW_HIP_CHECK(hipMemcpyAsync(all, d + 1, 8 * (size_t)G, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
        if (local_status) return local_status;
        for (int g = 0; g < G; g++)
            if (all[g]) { set_error("rank %d failed to prepare the sharded call (status %lld)", g, (long long)all[g]); return LW_ERR_COMM; }
        return LW_OK;
    }
};

struct SimTransport : Transport {   // G virtual ranks on one device
    SimTransport(Context &c, int g) { cx = &c; G = g; first = 0; nlocal = g; }
    int all_to_all(const char *const *send, char *const *recv, size_t chunk_bytes, uint32_t batch, size_t bstride_bytes,
                   hipStream_t s) override {
        for (int g = 0; g < G; g++)
            for (uint32_t b = 0; b < batch; b++)
                for (int h = 0; h < G; h++)
                    LW_HIP_CHECK(hipMemcpyAsync(recv[h] + b * bstride_bytes + (size_t)g * chunk_bytes,
                                                send[g] + b * bstride_bytes + (size_t)h * chunk_bytes, chunk_bytes,
                                                hipMemcpyDeviceToDevice, s), LW_ERR_LAUNCH);
        return LW_OK;
    }
    int all_gather(const char *const *send, char *const *recv, size_t bytes, hipStream_t s) override {
        for (int h = 0; h < G; h++)
            for (int g = 0; g < G; g++)
                LW_HIP_CHECK(hipMemcpyAsync(recv[h] + (size_t)g * bytes, send[g], bytes, hipMemcpyDeviceToDevice, s), LW_ERR_LAUNCH);
        return LW_OK;
    }
    int agree(int local_status, hipStream_t) override { return local_status; }   // all virtual ranks share one status
};

// ---------------------------------------------------------------- step F: local interleave
// out[b][k2 * G + k1] = in[b][k1 * sl + k2]: the third exchange delivers, from every rank k1, the slice k2 in
// [g*sl, (g+1)*sl) of its cyclic shard X[k1 + G*k2]; natural index within my block is (k2 - g*sl)*G + k1.
template <class VEC, int VPE>
__global__ void shard_interleave_kernel(const VEC *in, VEC *out, uint32_t lg, uint64_t sl, uint64_t in_bstride, uint64_t out_bstride) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // output vector index within one batch entry
    const uint64_t o = t / VPE, v = t % VPE;
    if (o >= (sl << lg)) return;
    const uint64_t k1 = o & ((1ull << lg) - 1), k2 = o >> lg;
    out[((uint64_t)blockIdx.y * out_bstride + o) * VPE + v] = in[((uint64_t)blockIdx.y * in_bstride + k1 * sl + k2) * VPE + v];
}

static int launch_interleave(Context &c, size_t eb, const void *in, void *out, uint32_t lg, uint64_t sl, uint32_t batch,
                             uint64_t in_bstride, uint64_t out_bstride, hipStream_t s) {
    const uint64_t M = sl << lg;
    hipEvent_t pe = c.prof_begin(s);
    if (eb == 32) {
        dim3 grid((uint32_t)((2 * M + 255) / 256), batch);
        hipLaunchKernelGGL((shard_interleave_kernel<uint4, 2>), grid, dim3(256), 0, s, (const uint4 *)in, (uint4 *)out, lg, sl, in_bstride, out_bstride);
    } else if (eb == 8) {
        dim3 grid((uint32_t)((M + 255) / 256), batch);
        hipLaunchKernelGGL((shard_interleave_kernel<uint64_t, 1>), grid, dim3(256), 0, s, (const uint64_t *)in, (uint64_t *)out, lg, sl, in_bstride, out_bstride);
    } else {
        dim3 grid((uint32_t)((M + 255) / 256), batch);
        hipLaunchKernelGGL((shard_interleave_kernel<uint32_t, 1>), grid, dim3(256), 0, s, (const uint32_t *)in, (uint32_t *)out, lg, sl, in_bstride, out_bstride);
    }
    c.prof_end("shard_interleave_kernel", pe, s);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

// ---------------------------------------------------------------- the schedule
// Cross-stream dependencies of one call: events (no timing) taken from a context pool and handed back at the end.
struct SyncEvents {
    Context &c;
    std::vector<hipEvent_t> taken;
    explicit SyncEvents(Context &cc) : c(cc) {}
    hipEvent_t get() {
        hipEvent_t e = nullptr;
        if (!c.sync_pool.empty()) { e = c.sync_pool.back(); c.sync_pool.pop_back(); }
        else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        taken.push_back(e);
        return e;
    }
    ~SyncEvents() { for (hipEvent_t e : taken) c.sync_pool.push_back(e); }
};
// `to` waits for everything enqueued on `from` so far
static int chain(SyncEvents &ev, hipStream_t from, hipStream_t to) {
    hipEvent_t e = ev.get();
    if (!e) { set_error("hipEventCreate failed"); return LW_ERR_LAUNCH; }
    LW_HIP_CHECK(hipEventRecord(e, from), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipStreamWaitEvent(to, e, 0), LW_ERR_LAUNCH);
    return LW_OK;
}
int ensure_aux_stream(Context &c);   // msm.hip: the context's side stream

// in[i] / out[i]: buffers of local rank first + i; batch entries in_bstride / out_bstride elements apart.
//
// Steps per batch column:  1 A exchange   2 B cross step   3 C exchange   4 D local NTT   [5 E exchange   6 F interleave].
// The exchanges run on the context's side stream, the kernels on the caller's stream, one column at a time, enqueued
// diagonal by diagonal (A of column d, C of column d-1, E of column d-2, each followed by its kernel): while column k's
// cross step or local transform runs, column k+1's slices are already on the links — BASELINE config 4 has four columns.
// A single column (batch = 1) degenerates to the serial order.
// stop_after (self-test only, 0 = all): run steps 1..stop_after and hand the buffer that step wrote (batch x M elements per
// rank, dense) to out[] — tests/test_gpu_distributed.py compares the Python transliteration with it step by step.
static int ntt_sharded_run(Context &c, Transport &tp, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const char *const *in,
                           char *const *out, uint64_t in_bstride, uint64_t out_bstride, uint32_t L, uint32_t batch, int natural,
                           hipStream_t s, int stop_after = 0) {
    const int G = tp.G;
    uint32_t lg = 0;
    while ((1 << lg) < G) lg++;
    if ((1 << lg) != G || lg > 3) {
        set_error("sharded NTT supports 1, 2, 4 or 8 ranks (got %d)", G);
        return LW_ERR_BAD_ARG;
    }
    if (L < 2 * lg) {
        set_error("2^%u elements cannot be block-distributed and sliced over %d ranks", L, G);
        return LW_ERR_BAD_ARG;
    }
    const size_t eb = lw_hip_field_elem_bytes(field, layout);
    const uint64_t M = 1ull << (L - lg), sl = M >> lg;
    const int nl = tp.nlocal;
    const size_t per_rank = (size_t)batch * M * eb;
    // Everything that can fail for lack of memory is taken BEFORE the first exchange and the outcome is agreed on by all
    // ranks (one tiny all-gather, first call of a shape only): a rank that returned early would leave the others blocked
    // in the collective.  Later local failures (launch errors) do not skip exchanges either, see below.
    int prep = LW_OK;
    if (c.shard_a.ensure(per_rank * nl) || c.shard_b.ensure(per_rank * nl) || c.scratch.ensure((size_t)M * eb)) prep = LW_ERR_ALLOC;
    if (!prep) prep = ensure_aux_stream(c);
    const uint64_t shape_key = ((uint64_t)field << 60) ^ ((uint64_t)layout << 56) ^ ((uint64_t)dir << 52) ^ ((uint64_t)L << 40) ^
                               ((uint64_t)batch << 8) ^ (uint64_t)G ^ ((uint64_t)(natural != 0) << 48);
    if (prep || c.shard_prepared_key != shape_key) {
        int rc0 = tp.agree(prep, s);
        if (rc0) return rc0;
        c.shard_prepared_key = shape_key;
    }
    hipStream_t cs = c.aux_hi;   // exchanges: high priority, so that their kernels get CU slots under a running NTT kernel
    std::vector<const char *> src(nl);
    std::vector<char *> a(nl), b(nl), dst(nl);
    for (int i = 0; i < nl; i++) {
        a[i] = (char *)c.shard_a.p + per_rank * i;
        b[i] = (char *)c.shard_b.p + per_rank * i;
    }
    int rc = LW_OK, local_rc = LW_OK;
    if (G == 1) {   // degenerate: one exchange with myself, so that a 1-rank communicator still exercises the transport
        for (uint32_t bi = 0; bi < batch; bi++) {
            const char *s1[1] = {in[0] + (size_t)bi * in_bstride * eb};
            char *r1[1] = {a[0] + (size_t)bi * M * eb};
            rc = tp.all_to_all(s1, r1, M * eb, 1, 0, s);
            if (rc) return rc;
        }
        return ntt_device_locked(c, field, layout, dir, a[0], out[0], L, batch, out_bstride, nullptr, s, 0xffffffffu);
    }
    const int last_step = stop_after > 0 ? stop_after : (natural ? 6 : 4);
    const int nex = natural ? 3 : 2;                        // exchanges per column
    SyncEvents ev(c);
    rc = chain(ev, s, cs);                                  // the caller's input is ready once `s` gets here
    if (rc) return rc;
    const bool d_direct = !natural && out_bstride == M;     // step D may write straight into the caller's buffer
    for (uint32_t d = 0; d < batch + (uint32_t)nex - 1; d++) {
        for (int st = 0; st < nex; st++) {
            if (d < (uint32_t)st || d - st >= batch) continue;
            const uint32_t bi = d - (uint32_t)st;
            const size_t boff = (size_t)bi * M * eb;
            const int ex_step = 2 * st + 1, k_step = 2 * st + 2;
            if (ex_step > last_step) continue;
            // ---- exchange `st` of column bi on the side stream (after the kernel that produced its payload)
            if (st > 0) { rc = chain(ev, s, cs); if (rc) return rc; }
            for (int i = 0; i < nl; i++) {
                src[i] = st == 0 ? in[i] + (size_t)bi * in_bstride * eb : b[i] + boff;
                dst[i] = a[i] + boff;
            }
            rc = tp.all_to_all(src.data(), dst.data(), sl * eb, 1, 0, cs);
            if (rc) return rc;                              // a failing collective is fatal for the communicator anyway
            rc = chain(ev, cs, s);
            if (rc) return rc;
            if (k_step > last_step || local_rc) continue;   // after a local failure: keep exchanging, stop computing
            // ---- its kernel on the caller's stream
            for (int i = 0; i < nl && !local_rc; i++) {
                const uint64_t g = (uint64_t)(tp.first + i);
                if (st == 0)        // B: cross-shard step on my j2 slice [g*sl, (g+1)*sl)
                    local_rc = ntt_cross_device(c, field, layout, dir, a[i] + boff, b[i] + boff, L, lg, g * sl, sl, sl, 1, M, s);
                else if (st == 1)   // D: local M-point transform: z[k2] = X[g + G*k2]
                    local_rc = ntt_device_locked(c, field, layout, dir, a[i] + boff,
                                                 d_direct && !stop_after ? out[i] + (size_t)bi * out_bstride * eb : b[i] + boff, L - lg, 1, 0,
                                                 nullptr, s, 0xffffffffu);
                else                // F: interleave into natural order
                    local_rc = launch_interleave(c, eb, a[i] + boff, stop_after ? b[i] + boff : out[i] + (size_t)bi * out_bstride * eb, lg, sl,
                                                 1, M, M, s);
            }
        }
    }
    if (local_rc) return local_rc;
    if (stop_after || (!natural && !d_direct)) {
        // where the last executed step left its result: A, C, E -> a;  B, D, F(self-test) -> b
        const bool in_a = stop_after && (last_step & 1);
        for (int i = 0; i < nl; i++)
            LW_HIP_CHECK(hipMemcpy2DAsync(out[i], out_bstride * eb, (in_a ? a[i] : b[i]), M * eb, M * eb, batch, hipMemcpyDeviceToDevice, s),
                         LW_ERR_LAUNCH);
    }
    return LW_OK;
}

// ---------------------------------------------------------------- sharded MSM: bucket-slice exchange (second form)
// Every rank accumulates its pairs into the full bucket array [W][2^(c-1)] (no running sums yet); an all-to-all hands rank g
// the bucket range [g*Bs, (g+1)*Bs) of every window from everyone; rank g adds the G contributions per bucket and runs the
// running sums over ITS slice only — the bucket reduce, a constant ~4.7 ms per MSM at c = 20 whatever N is, shrinks with G —
// and an all-gather of the per-slice (S, A) pairs (2 W points per rank) lets every rank fold the result.  This is the north
// star's "bucket all-reduce": RCCL has no user-defined reduction, so the reduce-scatter half is an all-to-all plus local
// group additions and the all-gather half carries the already reduced window partials.
// scalars[i] / points[i] / n_local[i]: the pairs of local rank first + i.
static int msm_sharded_run(Context &c, Transport &tp, lw_curve_t curve, const uint64_t *const *scalars, const void *const *points,
                           const size_t *n_local, void *out_host, hipStream_t s) {
    const int G = tp.G, nl = tp.nlocal;
    const size_t pb = lw_hip_curve_point_bytes(curve);
    // 1. window width: every rank must cut its scalars the same way.  All ranks learn all n_local (one tiny all-gather) and
    //    apply the single-GPU rule to the largest shard.
    if (c.small.ensure(8 * (size_t)(nl + G) * (size_t)nl + 256)) return LW_ERR_ALLOC;
    uint64_t all_n[8] = {0};
    {
        std::vector<const char *> snd(nl);
        std::vector<char *> rcv(nl);
        uint64_t *d = (uint64_t *)c.small.p;
        for (int i = 0; i < nl; i++) {
            const uint64_t v = n_local[i];
            LW_HIP_CHECK(hipMemcpyAsync(d + i, &v, 8, hipMemcpyHostToDevice, s), LW_ERR_LAUNCH);
            snd[i] = (const char *)(d + i);
            rcv[i] = (char *)(d + nl + (size_t)i * G);
        }
        LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);   // (&v is a stack temporary)
        int rc = tp.all_gather(snd.data(), rcv.data(), 8, s);
        if (rc) return rc;
        LW_HIP_CHECK(hipMemcpyAsync(all_n, rcv[0], 8 * (size_t)G, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
    }
    uint64_t n_max = 0;
    for (int g = 0; g < G; g++) n_max = all_n[g] > n_max ? all_n[g] : n_max;
    uint32_t cbits = msm_window_bits_for((size_t)n_max);
    uint32_t lg = 0;
    while ((1 << lg) < G) lg++;
    if (cbits < lg + 1) cbits = lg + 1;                       // at least one bucket per slice
    const uint32_t W = (256 + cbits) / cbits;
    const size_t B = (size_t)1 << (cbits - 1), Bs = B / G;
    const size_t bucket_bytes = (size_t)W * B * pb, sa_bytes = 2 * (size_t)W * pb;
    // 2. buffers, then the local accumulations; the outcome is agreed on before the big exchange (nobody may stay behind in it)
    int prep = LW_OK;
    if (c.shard_b.ensure(bucket_bytes * nl) || (nl > 1 && c.shard_a.ensure(bucket_bytes * nl)) ||
        c.shard_c.ensure(sa_bytes * ((size_t)nl + (size_t)nl * G)))
        prep = LW_ERR_ALLOC;
    std::vector<const char *> snd(nl);
    std::vector<char *> rcv(nl);
    for (int i = 0; i < nl && !prep; i++) {
        char *buckets = nullptr;
        prep = msm_shard_accumulate(c, curve, scalars[i], points[i], n_local[i], cbits, s, &buckets);
        if (prep) break;
        if (nl > 1) {   // virtual ranks share one workspace: park this rank's buckets
            char *park = (char *)c.shard_a.p + bucket_bytes * i;
            LW_HIP_CHECK(hipMemcpyAsync(park, buckets, bucket_bytes, hipMemcpyDeviceToDevice, s), LW_ERR_LAUNCH);
            buckets = park;
        }
        snd[i] = buckets;
        rcv[i] = (char *)c.shard_b.p + bucket_bytes * i;
    }
    int rc = tp.agree(prep, s);
    if (rc) return rc;
    // 3. all-to-all: slice h of every window goes to rank h; received as [w][g][Bs]
    rc = tp.all_to_all(snd.data(), rcv.data(), Bs * pb, W, B * pb, s);
    if (rc) return rc;
    // 4. add the G contributions, running sums over my slice
    char *sa_send = (char *)c.shard_c.p, *sa_recv = sa_send + sa_bytes * nl;
    int local_rc = LW_OK;
    for (int i = 0; i < nl && !local_rc; i++) local_rc = msm_shard_reduce(c, curve, rcv[i], (uint32_t)G, cbits, sa_send + sa_bytes * i, s);
    // 5. all-gather of the (S, A) pairs — entered even after a local failure, see ntt_sharded_run — and the fold on the host
    for (int i = 0; i < nl; i++) {
        snd[i] = sa_send + sa_bytes * i;
        rcv[i] = sa_recv + sa_bytes * (size_t)G * i;
    }
    rc = tp.all_gather(snd.data(), rcv.data(), sa_bytes, s);
    if (rc) return rc;
    rc = tp.agree(local_rc, s);
    if (rc) return rc;
    std::vector<uint4> sa_host((sa_bytes * G + 15) / 16);
    LW_HIP_CHECK(hipMemcpyAsync(sa_host.data(), rcv[0], sa_bytes * G, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
    return msm_shard_combine(curve, (const char *)sa_host.data(), (uint32_t)G, cbits, out_host);
}

static int check_sharded_args(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *in, void *out, uint32_t L) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (dir != LW_DIR_FORWARD && dir != LW_DIR_INVERSE) { set_error("bad direction %d", (int)dir); return LW_ERR_BAD_ARG; }
    if (L > 63) { set_error("order %u > 63", L); return LW_ERR_ORDER_TOO_LARGE; }
    if (L > field_two_adicity(field)) { set_error("no primitive 2^%u-th root of unity in this field", L); return LW_ERR_ROOT_OF_UNITY; }
    if (!in || !out) { set_error("null buffer"); return LW_ERR_BAD_ARG; }
    return LW_OK;
}

}  // namespace lw

using namespace lw;

extern "C" {

int lw_hip_comm_unique_id(uint8_t *out_id) {
    if (!out_id) { set_error("null argument"); return LW_ERR_BAD_ARG; }
    Entry en(nullptr, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == LW_HIP_COMM_ID_BYTES, "lw_hip.h: LW_HIP_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId id;
    LW_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return LW_OK;
}

int lw_hip_comm_init(const uint8_t *unique_id, int rank, int nranks) {
    if (!unique_id || nranks < 1 || rank < 0 || rank >= nranks) { set_error("bad communicator arguments (rank %d of %d)", rank, nranks); return LW_ERR_BAD_ARG; }
    if (nranks & (nranks - 1) || nranks > 8) { set_error("communicator size %d: the sharded paths take 1, 2, 4 or 8 ranks (one node)", nranks); return LW_ERR_BAD_ARG; }
    Entry en(nullptr, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    int rc = rccl_load();
    if (rc) return rc;
    comm_release(en.c);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    LW_NCCL_CHECK(g_rccl.CommInitRank(&comm, nranks, id, rank));   // binds to the context's device (Entry set it)
    g_comm.comm = comm;
    g_comm.rank = rank;
    g_comm.nranks = nranks;
    return LW_OK;
}

int lw_hip_comm_shutdown(void) {
    Entry en(nullptr, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    (void)hipDeviceSynchronize();
    comm_release(en.c);
    return LW_OK;
}

int lw_hip_comm_info(int *rank, int *nranks) {
    std::lock_guard<std::mutex> g(lane(0).mu);   // the communicator lives with lane 0 (lock order: lane, then the shared hold)
    std::shared_lock<std::shared_mutex> sl(shared_state().rw);   // ... and shutdown releases it holding rw exclusively
    if (!g_comm.comm) { set_error("no communicator: call lw_hip_comm_init first"); return LW_ERR_COMM; }
    if (rank) *rank = g_comm.rank;
    if (nranks) *nranks = g_comm.nranks;
    return LW_OK;
}

int lw_hip_ntt_sharded_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_local, void *d_out_local,
                              uint32_t log2n_total, uint32_t batch, int natural_output, void *hip_stream) {
    int rc = check_sharded_args(field, layout, dir, d_in_local, d_out_local, log2n_total);
    if (rc) return rc;
    Entry en(hip_stream, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    if (!g_comm.comm) { set_error("no communicator: call lw_hip_comm_init first"); return LW_ERR_COMM; }
    if (batch == 0) return LW_OK;
    RcclTransport tp(en.c);
    uint32_t lg = 0;
    while ((1 << lg) < tp.G) lg++;
    if (log2n_total < lg) { set_error("2^%u elements over %d ranks", log2n_total, tp.G); return LW_ERR_BAD_ARG; }
    const uint64_t M = 1ull << (log2n_total - lg);
    const char *in[1] = {(const char *)d_in_local};
    char *out[1] = {(char *)d_out_local};
    return ntt_sharded_run(en.c, tp, field, layout, dir, in, out, M, M, log2n_total, batch, natural_output, en.stream);
}

static int sharded_selftest(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_full, void *d_out_full,
                            uint32_t log2n_total, uint32_t log2_shards, uint32_t batch, int natural_output, int stop_after,
                            void *hip_stream) {
    int rc = check_sharded_args(field, layout, dir, d_in_full, d_out_full, log2n_total);
    if (rc) return rc;
    if (log2_shards < 1 || log2_shards > 3) { set_error("self-test takes 2, 4 or 8 virtual ranks"); return LW_ERR_BAD_ARG; }
    if (d_in_full == d_out_full) { set_error("self-test needs distinct buffers"); return LW_ERR_BAD_ARG; }
    Entry en(hip_stream, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    if (batch == 0) return LW_OK;
    SimTransport tp(en.c, 1 << log2_shards);
    if (log2n_total < 2 * log2_shards) { set_error("2^%u elements over %d ranks", log2n_total, tp.G); return LW_ERR_BAD_ARG; }
    const size_t eb = lw_hip_field_elem_bytes(field, layout);
    const uint64_t N = 1ull << log2n_total, M = N >> log2_shards;
    std::vector<const char *> in(tp.G);
    std::vector<char *> out(tp.G);
    for (int g = 0; g < tp.G; g++) {
        in[g] = (const char *)d_in_full + (size_t)g * M * eb;
        out[g] = (char *)d_out_full + (size_t)g * M * eb;
    }
    return ntt_sharded_run(en.c, tp, field, layout, dir, in.data(), out.data(), N, N, log2n_total, batch, natural_output, en.stream,
                           stop_after);
}
int lw_hip_ntt_sharded_selftest_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_full, void *d_out_full,
                                       uint32_t log2n_total, uint32_t log2_shards, uint32_t batch, int natural_output, void *hip_stream) {
    return sharded_selftest(field, layout, dir, d_in_full, d_out_full, log2n_total, log2_shards, batch, natural_output, 0, hip_stream);
}
int lw_hip_ntt_sharded_selftest_steps_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_full, void *d_out_full,
                                             uint32_t log2n_total, uint32_t log2_shards, uint32_t batch, int natural_output, int stop_after,
                                             void *hip_stream) {
    if (stop_after < 1 || stop_after > (natural_output ? 6 : 4)) { set_error("stop_after %d out of range", stop_after); return LW_ERR_BAD_ARG; }
    return sharded_selftest(field, layout, dir, d_in_full, d_out_full, log2n_total, log2_shards, batch, natural_output, stop_after, hip_stream);
}

int lw_hip_msm_sharded_device(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n_local, void *out_point_host,
                              void *hip_stream) {
    const size_t pb = lw_hip_curve_point_bytes(curve);
    if (pb == 0 || !out_point_host) { set_error("bad curve or null output"); return LW_ERR_BAD_ARG; }
    Entry en(hip_stream, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    if (!g_comm.comm) { set_error("no communicator: call lw_hip_comm_init first"); return LW_ERR_COMM; }
    RcclTransport tp(en.c);
    // LW_HIP_MSM_SHARD=partials (A/B on real hardware): the first form of SURVEY 8(e) — every rank runs the WHOLE Pippenger
    // on its shard and one partial sum per rank is all-gathered.  No bucket exchange (W x 2^(c-1) points per rank: 654 MB
    // for BN254 G1 at c = 20, ~1.7 ms on seven xGMI links by estimate) but the full ~2.5-4.7 ms of running sums on every
    // rank; the bucket-slice form saves 1.1 ms of them per rank at G = 8 (profiles/r03_msm_sharded_reduce.txt).  Which one
    // wins is a question for an 8-GPU node; the default is the form the north star names.
    static const bool partials = [] { const char *e = tuning_env("LW_HIP_MSM_SHARD"); return e && strcmp(e, "partials") == 0; }();
    if (partials) {
        Context &c = en.c;
        const int G = g_comm.nranks;
        // payload per rank: [status (16-byte header: the point keeps the alignment its stores assume) | partial sum]; the
        // local outcome travels WITH the point, so a failing rank still enters the collective and all return together
        constexpr size_t HDR = 16;
        const size_t slot = HDR + pb;
        std::vector<uint4> mine_v((slot + 15) / 16), all_v((slot * G + 15) / 16), pts_v((pb * G + 15) / 16);
        char *mine = (char *)mine_v.data(), *all = (char *)all_v.data(), *pts = (char *)pts_v.data();
        memset(mine, 0, slot);
        const int local_rc = msm_device(c, curve, d_scalars, d_points, n_local, mine + HDR, en.stream, 0, 0);
        const int64_t st = local_rc;
        memcpy(mine, &st, 8);
        if (local_rc) memset(mine + HDR, 0, pb);
        if (c.shard_c.ensure(slot * (G + 1))) return LW_ERR_ALLOC;
        char *d_send = (char *)c.shard_c.p, *d_recv = d_send + slot;
        LW_HIP_CHECK(hipMemcpyAsync(d_send, mine, slot, hipMemcpyHostToDevice, en.stream), LW_ERR_LAUNCH);
        LW_NCCL_CHECK(g_rccl.AllGather(d_send, d_recv, slot, ncclChar, g_comm.comm, en.stream));
        LW_HIP_CHECK(hipMemcpyAsync(all, d_recv, slot * G, hipMemcpyDeviceToHost, en.stream), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipStreamSynchronize(en.stream), LW_ERR_LAUNCH);
        if (local_rc) return local_rc;
        for (int g = 0; g < G; g++) {
            int64_t sg = 0;
            memcpy(&sg, all + slot * g, 8);
            if (sg) { set_error("rank %d failed its local MSM (status %lld)", g, (long long)sg); return LW_ERR_COMM; }
            memcpy(pts + pb * g, all + slot * g + HDR, pb);
        }
        return msm_sum_points_host(curve, pts, (size_t)G, out_point_host);
    }
    const uint64_t *sc[1] = {d_scalars};
    const void *pt[1] = {d_points};
    const size_t nn[1] = {n_local};
    return msm_sharded_run(en.c, tp, curve, sc, pt, nn, out_point_host, en.stream);
}

// The same run with G = 2^log2_shards virtual ranks on ONE device: virtual rank g owns the pairs [g * n / G, (g + 1) * n / G).
int lw_hip_msm_sharded_selftest_device(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n_total, uint32_t log2_shards,
                                       void *out_point_host, void *hip_stream) {
    const size_t pb = lw_hip_curve_point_bytes(curve);
    if (pb == 0 || !out_point_host) { set_error("bad curve or null output"); return LW_ERR_BAD_ARG; }
    if (log2_shards < 1 || log2_shards > 3) { set_error("self-test takes 2, 4 or 8 virtual ranks"); return LW_ERR_BAD_ARG; }
    Entry en(hip_stream, true);   // lane 0: the communicator and its buffers
    if (en.rc) return en.rc;
    SimTransport tp(en.c, 1 << log2_shards);
    std::vector<const uint64_t *> sc(tp.G);
    std::vector<const void *> pt(tp.G);
    std::vector<size_t> nn(tp.G);
    for (int g = 0; g < tp.G; g++) {
        const size_t b0 = n_total * (size_t)g / tp.G, b1 = n_total * (size_t)(g + 1) / tp.G;
        sc[g] = d_scalars + 4 * b0;
        pt[g] = (const char *)d_points + pb * b0;
        nn[g] = b1 - b0;
    }
    return msm_sharded_run(en.c, tp, curve, sc.data(), pt.data(), nn.data(), out_point_host, en.stream);
}

}  // extern "C"
