// Library context: device selection, error plumbing, twiddle caches, scratch memory.
// The reference creates and drops all device state per call (math/src/fft/gpu/cuda/polynomial.rs:21,37);
// here it is built once and reused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>
#include "../../include/lw_hip.h"

// Ablation switches (skip butterflies / loads / stores: WRONG results, timing only) exist only in builds made with
// -DLW_HIP_ABLATION (tools/ab_ntt.sh); in the shipped library LW_DBG() is the constant 0 and the branches fold away.
#ifdef LW_HIP_ABLATION
#define LW_DBG(p) ((p).dbg)
#else
#define LW_DBG(p) 0u
#endif

namespace lw {

void set_error(const char *fmt, ...);
// Diagnostic / A-B switches (LW_HIP_MSM_C, LW_HIP_NTT_PLAN, ...) are honoured only in processes started with
// LW_HIP_TUNING=1 (read once): a stray variable in a prover's environment cannot change the schedule.
const char *tuning_env(const char *name);
extern thread_local std::string g_last_error;

#define LW_HIP_CHECK(expr, code)                                                            \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ::lw::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (code);                                                                  \
        }                                                                                   \
    } while (0)

struct DeviceBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);   // grow-only; returns LW_OK / LW_ERR_ALLOC
    void release();
};

struct TwiddleTable {
    DeviceBuf buf;
    uint32_t log_n = 0;   // table holds 2^(log_n-1) entries; prefix property serves every smaller size
    bool valid = false;
};

struct CosetCache {
    DeviceBuf lo, hi;
    uint32_t hbits = 0;
    uint32_t hi_bits = 0;
    uint32_t words[8] = {0};
    int field = -1;
    bool inverse = false;
    bool fold_ninv = false;
    bool valid = false;
};

struct ProfSpan {
    const char *name;
    hipEvent_t e0, e1;
};

// State shared by all lanes (below): the twiddle tables (256 MiB per field and direction at 2^24 — not something to keep
// per lane) and the lock that separates ordinary calls from the operations that touch every lane.
//   rw held SHARED  by every entry point for the duration of its call (it may hold table pointers, lane buffers);
//   rw held UNIQUE  by init / shutdown / profile begin + end / timings, and while a twiddle table is rebuilt: then no call is
//                   running, so the old table can be freed (hipFree also drains the device) and every lane may be touched.
struct SharedState {
    std::shared_mutex rw;
    TwiddleTable tw[3][2];   // [field][dir]
    bool initialised = false;
    int device = -1;
};
SharedState &shared_state();

// One LANE of the library: everything a call scribbles on (scratch, staging, MSM workspace, side streams, the cross-stream
// ordering event).  A call takes the first lane whose lock is free, so calls from different host threads — the reference's
// rayon loop over columns, provers/stark/src/trace.rs:186-190, or an NTT caller beside an MSM caller — run CONCURRENTLY on
// different lanes (one caller's download overlaps the other's upload and kernels); a single-threaded caller always gets
// lane 0 and sees exactly the behaviour of a one-context library.  Lanes allocate lazily: memory grows with the
// concurrency actually used.
constexpr int LW_LANES = 4;

struct Context {
    bool profiling = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> event_pool;
    // record a begin/end event pair around one kernel launch when profiling
    hipEvent_t prof_begin(hipStream_t s);
    void prof_end(const char *name, hipEvent_t e0, hipStream_t s);

    bool initialised = false;   // mirrors of SharedState (set for every lane by init)
    int device = -1;
    std::mutex mu;
    TwiddleTable (&tw)[3][2] = shared_state().tw;   // shared by all lanes; rebuilt only under SharedState::rw held unique
    std::shared_lock<std::shared_mutex> *call_lock = nullptr;   // the running call's shared hold on SharedState::rw (Entry)
    DeviceBuf bb_coset;      // two-level power tables of the current BabyBear coset offset (rebuilt per call: 8192 exponentiations)
    DeviceBuf scratch;
    DeviceBuf small;         // staging for small power tables
    CosetCache coset[3];     // forward coset, inverse coset, multi-GPU cross-step twiddle base
    DeviceBuf msm_ws;
    uint32_t *pinned_words = nullptr;   // 64 pinned host words: small device -> host results read while other streams run (msm_core.cuh)
    DeviceBuf msm_scalars;   // canonical scalars when the caller hands Montgomery-form FrElements
    // set by the SRS entry points around msm_device, under the entry lock: the point set holds window-shifted copies
    // (rows w * stride + i = 2^(c w) P_i) and all windows share one bucket set (msm_core.cuh build_fold)
    uint32_t msm_fold_c = 0;
    uint64_t msm_fold_stride = 0;
    // set by msm_device for host-buffer calls: run right after the sort of the scalars is enqueued (upload of the points and
    // their normalisation on the side stream, so that the sort runs under the upload), then cleared
    std::function<int()> msm_after_sort;
    DeviceBuf msm_prefix;    // running products of the batch inversion (msm_to_affine_kernel), one base-field element per point
    DeviceBuf msm_affine;    // per-call affine copy of a large projective point set (msm_device)
    hipStream_t aux_stream = nullptr;   // side stream: the normalisation of the points runs beside the scalar sort
    hipStream_t aux_hi = nullptr;       // high-priority side stream: short kernels that must get CU slots UNDER a long-running kernel of the caller's stream
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    DeviceBuf host_io_a, host_io_b;   // device staging for the host-buffer entry points
    hipStream_t io_stream = nullptr;  // ... and the stream they run on
    DeviceBuf pipe_tmp;               // intermediates of the device-resident pipelines (FRI layer evaluation, Groth16 cosets)
    lw_timings_t timings = {};
    // Cross-stream ordering of the context-owned buffers (scratch, tables, staging, MSM workspace): every entry point
    // records `order_event` on its launch stream when it returns; a call arriving on a different stream first makes
    // that stream wait for it.  (c.mu only serialises the host side.)
    hipEvent_t order_event = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    DeviceBuf shard_a, shard_b;       // exchange buffers of the sharded (multi-GPU) entry points
    DeviceBuf shard_c;                // (S, A) pairs of the sharded MSM's all-gather
    uint64_t shard_prepared_key = 0;  // shape of the last sharded NTT whose allocations every rank agreed on (comm.hip)
    std::vector<hipEvent_t> sync_pool;   // untimed events for cross-stream dependencies inside one call
    void release_all();               // frees every cached device object (shutdown / device change)
};

// Held by every extern "C" entry point for its whole duration: context lock, lazy init, device binding (the HIP
// current device is per thread) and the cross-stream ordering above.
struct Entry {
    // Lock order: the lane first, SharedState::rw second.  (A caller waiting for a lane must not hold rw: the lane's owner may be
    // waiting to upgrade its own hold to exclusive, which needs every shared holder out.)
    std::unique_lock<std::mutex> lock;            // the lane's lock
    Context &c;
    std::shared_lock<std::shared_mutex> shared;   // SharedState::rw, shared
    hipStream_t stream;
    int rc = LW_OK;
    int prev_device = -1;
    // lane0: the call uses process-wide state that lives with lane 0 (the RCCL communicator and its exchange buffers)
    explicit Entry(void *hip_stream, bool lane0 = false);
    // Host-buffer entry points: switch the call to the lane's own (non-blocking) stream instead of the null stream, so that
    // calls running on different lanes do not serialise on it; returns the stream (nullptr + rc set on failure).
    hipStream_t use_lane_stream();
    ~Entry();
    Entry(const Entry &) = delete;
};
// Upgrade of the running call's hold on SharedState::rw to exclusive for the duration of a scope (twiddle rebuild): releases
// the shared hold first — other calls drain — and takes it back afterwards.  Re-check the condition after construction.
struct ExclusiveScope {
    Context &c;
    explicit ExclusiveScope(Context &cc) : c(cc) {
        if (c.call_lock) c.call_lock->unlock();
        shared_state().rw.lock();
    }
    ~ExclusiveScope() {
        shared_state().rw.unlock();
        if (c.call_lock) c.call_lock->lock();
    }
};

Context &lane(int i);
inline Context &ctx() { return lane(0); }
int ensure_init();

}  // namespace lw
