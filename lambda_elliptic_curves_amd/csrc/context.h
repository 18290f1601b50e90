// Library context: device selection, error plumbing, twiddle caches, scratch memory.
// The reference creates and drops all device state per call (math/src/fft/gpu/cuda/polynomial.rs:21,37);
// here it is built once and reused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/lw_hip.h"

namespace lw {

void set_error(const char *fmt, ...);
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

struct Context {
    bool profiling = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> event_pool;
    // record a begin/end event pair around one kernel launch when profiling
    hipEvent_t prof_begin(hipStream_t s);
    void prof_end(const char *name, hipEvent_t e0, hipStream_t s);

    bool initialised = false;
    int device = -1;
    std::mutex mu;
    TwiddleTable tw[3][2];   // [field][dir]
    DeviceBuf scratch;
    DeviceBuf small;         // staging for small power tables
    CosetCache coset[3];     // forward coset, inverse coset, multi-GPU cross-step twiddle base
    DeviceBuf msm_ws;
    DeviceBuf msm_scalars;   // canonical scalars when the caller hands Montgomery-form FrElements
    DeviceBuf msm_affine;    // per-call affine copy of a large projective point set (msm_device)
    hipStream_t aux_stream = nullptr;   // side stream: the normalisation of the points runs beside the scalar sort
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    DeviceBuf host_io_a, host_io_b;   // device staging for the host-buffer entry points
    lw_timings_t timings = {};
};

Context &ctx();
int ensure_init();

}  // namespace lw
