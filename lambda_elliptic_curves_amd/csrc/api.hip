// extern "C" boundary (include/lw_hip.h) + context implementation.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <new>
#include <thread>
#include <vector>
#include "context.h"
#include "field.cuh"

namespace lw {

thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int DeviceBuf::ensure(size_t need) {
    if (need <= bytes && p) return LW_OK;
    if (p) {
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    if (need == 0) need = 256;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        return LW_ERR_ALLOC;
    }
    bytes = need;
    return LW_OK;
}
void DeviceBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

hipEvent_t Context::prof_begin(hipStream_t s) {
    if (!profiling) return nullptr;
    hipEvent_t e = nullptr;
    if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, s);
    return e;
}
void Context::prof_end(const char *name, hipEvent_t e0, hipStream_t s) {
    if (!profiling || !e0) return;
    hipEvent_t e1 = nullptr;
    if (!event_pool.empty()) { e1 = event_pool.back(); event_pool.pop_back(); }
    else if (hipEventCreate(&e1) != hipSuccess) return;
    (void)hipEventRecord(e1, s);
    spans.push_back(ProfSpan{name, e0, e1});
}

SharedState &shared_state() {
    static SharedState s;
    return s;
}
Context &lane(int i) {
    static Context lanes[LW_LANES];
    return lanes[i];
}

const char *tuning_env(const char *name) {
    static const bool on = [] { const char *e = getenv("LW_HIP_TUNING"); return e && atoi(e) == 1; }();
    return on ? getenv(name) : nullptr;
}

void ntt_set_max_pass_stages(uint32_t r);
void ntt_set_debug(uint32_t d);

void host_pool_release_all();
void comm_release(Context &c);   // comm.hip
// every cached device object of every lane and the shared tables (shutdown / device change); SharedState::rw held unique
static void release_everything() {
    (void)hipDeviceSynchronize();
    comm_release(lane(0));
    host_pool_release_all();
    for (int i = 0; i < LW_LANES; i++) lane(i).release_all();
    SharedState &sh = shared_state();
    for (int f = 0; f < 3; f++)
        for (int d = 0; d < 2; d++) {
            sh.tw[f][d].buf.release();
            sh.tw[f][d].valid = false;
        }
}
// SharedState::rw held unique
static int init_locked(const int *device_ids, int n_devices) {
    SharedState &sh = shared_state();
#ifdef LW_HIP_ABLATION
    if (const char *e = tuning_env("LW_HIP_NTT_DBG")) ntt_set_debug((uint32_t)atoi(e));   // wrong results, timing only
#endif
    if (const char *e = tuning_env("LW_HIP_NTT_MAX_R")) ntt_set_max_pass_stages((uint32_t)atoi(e));
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return LW_ERR_NO_DEVICE;
    }
    int dev = 0;
    if (device_ids && n_devices > 1) {   // one process (context) per GPU; multi-GPU jobs use lw_hip_comm_init
        set_error("lw_hip_init takes one device id per process (got %d); shard across processes with lw_hip_comm_init", n_devices);
        return LW_ERR_BAD_ARG;
    }
    if (device_ids && n_devices > 0) {
        dev = device_ids[0];
        if (dev < 0 || dev >= count) {
            set_error("device id %d out of range (0..%d)", dev, count - 1);
            return LW_ERR_NO_DEVICE;
        }
        LW_HIP_CHECK(hipSetDevice(dev), LW_ERR_NO_DEVICE);
    } else {
        LW_HIP_CHECK(hipGetDevice(&dev), LW_ERR_NO_DEVICE);
    }
    hipDeviceProp_t prop;
    LW_HIP_CHECK(hipGetDeviceProperties(&prop, dev), LW_ERR_NO_DEVICE);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library ships gfx950 code objects only", dev, prop.gcnArchName);
        return LW_ERR_NO_DEVICE;
    }
    if (sh.initialised && sh.device != dev) {   // every cached table / workspace lives on the old device
        (void)hipSetDevice(sh.device);
        release_everything();
        (void)hipSetDevice(dev);
    }
    sh.device = dev;
    sh.initialised = true;
    for (int i = 0; i < LW_LANES; i++) {
        lane(i).device = dev;
        lane(i).initialised = true;
    }
    return LW_OK;
}

void Context::release_all() {
    scratch.release();
    small.release();
    for (int i = 0; i < 3; i++) {
        coset[i].lo.release();
        coset[i].hi.release();
        coset[i].valid = false;
    }
    msm_ws.release();
    if (pinned_words) { (void)hipHostFree(pinned_words); pinned_words = nullptr; }
    msm_scalars.release();
    msm_affine.release();
    msm_prefix.release();
    bb_coset.release();
    shard_a.release();
    shard_b.release();
    shard_c.release();
    if (aux_stream) { (void)hipStreamDestroy(aux_stream); aux_stream = nullptr; }
    if (aux_hi) { (void)hipStreamDestroy(aux_hi); aux_hi = nullptr; }
    if (io_stream) { (void)hipStreamDestroy(io_stream); io_stream = nullptr; }
    if (aux_fork) { (void)hipEventDestroy(aux_fork); aux_fork = nullptr; }
    if (aux_join) { (void)hipEventDestroy(aux_join); aux_join = nullptr; }
    if (order_event) { (void)hipEventDestroy(order_event); order_event = nullptr; }
    have_last = false;
    for (auto &sp : spans) { (void)hipEventDestroy(sp.e0); (void)hipEventDestroy(sp.e1); }
    spans.clear();
    for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
    event_pool.clear();
    for (hipEvent_t e : sync_pool) (void)hipEventDestroy(e);
    sync_pool.clear();
    shard_prepared_key = 0;
    profiling = false;
    host_io_a.release();
    host_io_b.release();
    pipe_tmp.release();
    timings.twiddle_bytes = timings.scratch_bytes = 0;
}

// first free lane; when all are busy, wait for one (round robin, so that waiters spread over the lanes)
static Context &pick_lane(std::unique_lock<std::mutex> &lk, bool lane0) {
    if (lane0) {
        lk = std::unique_lock<std::mutex>(lane(0).mu);
        return lane(0);
    }
    for (int i = 0; i < LW_LANES; i++) {
        lk = std::unique_lock<std::mutex>(lane(i).mu, std::try_to_lock);
        if (lk.owns_lock()) return lane(i);
    }
    static std::atomic<unsigned> rr{0};
    const int i = (int)(rr.fetch_add(1) % LW_LANES);
    lk = std::unique_lock<std::mutex>(lane(i).mu);
    return lane(i);
}
static std::shared_lock<std::shared_mutex> enter_shared() {
    for (int attempt = 0;; attempt++) {   // lazily initialised; a shutdown may slip in between the two locks, hence the loop
        const int rc = ensure_init();
        std::shared_lock<std::shared_mutex> sl(shared_state().rw);
        if (shared_state().initialised || rc != LW_OK || attempt > 3) return sl;
    }
}
Entry::Entry(void *hip_stream, bool lane0) : c(pick_lane(lock, lane0)), shared(enter_shared()), stream((hipStream_t)hip_stream) {
    if (!shared_state().initialised) {
        rc = LW_ERR_NO_DEVICE;   // ensure_init() left the reason in the thread's error message
        return;
    }
    c.call_lock = &shared;
    if (hipGetDevice(&prev_device) != hipSuccess) prev_device = -1;
    if (prev_device != c.device && hipSetDevice(c.device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", c.device);
        rc = LW_ERR_NO_DEVICE;
        return;
    }
    if (c.have_last && c.last_stream != stream && c.order_event &&
        hipStreamWaitEvent(stream, c.order_event, 0) != hipSuccess) {
        set_error("hipStreamWaitEvent on the previous call's stream failed");
        rc = LW_ERR_LAUNCH;
    }
}
hipStream_t Entry::use_lane_stream() {
    if (!c.io_stream && hipStreamCreateWithFlags(&c.io_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("cannot create the lane's stream");
        rc = LW_ERR_LAUNCH;
        return nullptr;
    }
    if (c.have_last && c.last_stream != c.io_stream && c.order_event && hipStreamWaitEvent(c.io_stream, c.order_event, 0) != hipSuccess) {
        set_error("hipStreamWaitEvent on the previous call's stream failed");
        rc = LW_ERR_LAUNCH;
        return nullptr;
    }
    stream = c.io_stream;
    return stream;
}
Entry::~Entry() {
    if (rc == LW_OK || c.initialised) {
        if (!c.order_event && c.initialised) (void)hipEventCreateWithFlags(&c.order_event, hipEventDisableTiming);
        if (c.order_event && hipEventRecord(c.order_event, stream) == hipSuccess) {
            c.last_stream = stream;
            c.have_last = true;
        }
    }
    c.call_lock = nullptr;
    if (prev_device >= 0 && prev_device != c.device) (void)hipSetDevice(prev_device);
}

int ensure_init() {
    SharedState &sh = shared_state();
    {
        std::shared_lock<std::shared_mutex> sl(sh.rw);
        if (sh.initialised) return LW_OK;
    }
    std::unique_lock<std::shared_mutex> ul(sh.rw);
    if (sh.initialised) return LW_OK;
    return init_locked(nullptr, 0);
}

// defined in ntt256.hip / ntt_bb.hip / msm.hip
int ntt256_device(Context &c, int field, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2n, uint32_t batch,
                  uint64_t stride, const uint32_t *coset_words, hipStream_t stream, uint32_t in_log2);
int ntt_bb_device(Context &c, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2n,
                  uint32_t batch, uint64_t stride, const void *coset_offset, hipStream_t stream, uint32_t in_log2);
int broadcast_device(size_t elem_bytes, const void *d_in, void *d_out, uint64_t n, uint32_t batch, uint64_t out_stride, hipStream_t stream);
int msm_device(Context &c, lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, void *out_host,
               hipStream_t stream, int scalars_montgomery, int affine_points, const void *h_points = nullptr);
int msm_normalize_device(Context &c, lw_curve_t curve, const void *d_in, size_t n, void *d_out, hipStream_t stream);
size_t msm_affine_bytes(lw_curve_t curve, size_t n);
int msm_fold_build(Context &c, lw_curve_t curve, void *d_rows, size_t n, uint32_t cbits, hipStream_t stream);
int ec_add_outer_device(Context &c, lw_curve_t curve, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out,
                        hipStream_t stream);
int ntt_cross_device(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                     uint32_t log2_total, uint32_t log2_g, uint64_t j2_begin, uint64_t slice_len, uint64_t chunk_stride,
                     uint32_t batch, uint64_t batch_stride, hipStream_t stream);
int ntt_device_locked(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                      uint32_t log2n, uint32_t batch, size_t stride, const void *coset, hipStream_t stream,
                      uint32_t in_log2 = 0xffffffffu);
int merkle_commit_device(Context &c, const void *d_cols, uint32_t n_cols, uint64_t col_stride, uint32_t log2n, int bit_reverse,
                         void *d_nodes, hipStream_t stream, uint32_t elem_bytes = 32);
int fri_layer_device(Context &c, lw_field_t field, const void *d_coeffs, uint64_t n, const uint32_t *d_zeta, const void *offset_ref,
                     uint32_t log2_domain, void *d_poly, uint32_t log2_block, void *d_eval, void *d_eval_br, void *d_nodes,
                     hipStream_t stream);
int groth16_h_device(Context &c, const void *d_l, const void *d_r, const void *d_o, uint32_t log2_gates, void *d_out, void *d_tmp,
                     hipStream_t stream);
int stripped_length_device(const void *d_elems, uint64_t n, uint64_t *d_len, hipStream_t stream);

int ntt256_gen_powers(int field, uint32_t order, uint64_t count, uint32_t bitrev, bool inverse, const uint32_t *scale_words, void *d_out,
                      hipStream_t stream);
int ntt_bb_gen_powers(lw_layout_t layout, uint32_t order, uint64_t count, uint32_t bitrev, bool inverse, const void *scale, void *d_out,
                      hipStream_t stream);
int gen_twiddles_device(Context &c, lw_field_t field, lw_layout_t layout, uint32_t order, int config, void *d_out, hipStream_t stream);
int bitrev_device(size_t elem_bytes, const void *d_in, void *d_out, uint32_t log2n, hipStream_t stream);

uint32_t field_two_adicity(lw_field_t f) {
    switch (f) {
        case LW_FIELD_STARK252: return Stark252::TWO_ADICITY;
        case LW_FIELD_BLS12_381_FR: return Fr381::TWO_ADICITY;
        default: return BabyBear::TWO_ADICITY;
    }
}

int check_field_layout(lw_field_t field, lw_layout_t layout) {
    bool ok = false;
    if (field == LW_FIELD_STARK252 || field == LW_FIELD_BLS12_381_FR) ok = layout == LW_LAYOUT_U64_LIMBS_MS_FIRST;
    if (field == LW_FIELD_BABYBEAR)
        ok = layout == LW_LAYOUT_BABYBEAR_U32_R32 || layout == LW_LAYOUT_BABYBEAR_U64_R64 || layout == LW_LAYOUT_EXT4_INTERLEAVED;
    if (!ok) {
        set_error("field %d does not support layout %d", (int)field, (int)layout);
        return LW_ERR_BAD_ARG;
    }
    return LW_OK;
}

// reference layout (u64, MS limb first) -> internal 8 x u32, LS first
static void words_from_ref(const void *ref, uint32_t *w) {
    const uint32_t *m = (const uint32_t *)ref;
    for (int k = 0; k < 8; k++) w[k] = m[2 * (3 - k / 2) + (k & 1)];
}

// in_log2 < log2n (forward only): low-degree extension of dense blocks of 2^in_log2 coefficients, see ntt256.hip / ntt_bb.hip
int ntt_device_locked(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                      uint32_t log2n, uint32_t batch, size_t stride, const void *coset, hipStream_t stream, uint32_t in_log2) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (dir != LW_DIR_FORWARD && dir != LW_DIR_INVERSE) {
        set_error("bad direction %d", (int)dir);
        return LW_ERR_BAD_ARG;
    }
    if (log2n > 63) {   // get_twiddles: order > 63 -> FFTError::OrderError (roots_of_unity.rs:70-72)
        set_error("order %u > 63", log2n);
        return LW_ERR_ORDER_TOO_LARGE;
    }
    if (log2n > field_two_adicity(field)) {   // traits.rs:88-90
        set_error("no primitive 2^%u-th root of unity in this field", log2n);
        return LW_ERR_ROOT_OF_UNITY;
    }
    if (log2n > 34) {
        set_error("2^%u elements exceed device memory", log2n);
        return LW_ERR_ALLOC;
    }
    if (batch == 0) return LW_OK;
    if (stride != 0 && stride < ((size_t)1 << log2n)) {
        set_error("batch stride %zu < transform length", stride);
        return LW_ERR_BAD_ARG;
    }
    if (!d_in || !d_out) {
        set_error("null buffer");
        return LW_ERR_BAD_ARG;
    }
    if (in_log2 > log2n) in_log2 = log2n;
    if (in_log2 < log2n && (dir != LW_DIR_FORWARD || d_in == d_out)) {
        set_error("low-degree extension needs the forward direction and distinct buffers");
        return LW_ERR_BAD_ARG;
    }
    // grid.y carries the batch: split very wide batches
    const uint32_t max_batch = 32768;
    if (batch > max_batch) {
        const size_t eb = lw_hip_field_elem_bytes(field, layout);
        const size_t s_out = stride ? stride : ((size_t)1 << log2n);
        const size_t s_in = in_log2 < log2n ? ((size_t)1 << in_log2) : s_out;
        for (uint32_t b0 = 0; b0 < batch; b0 += max_batch) {
            const uint32_t nb = batch - b0 < max_batch ? batch - b0 : max_batch;
            int rc2 = ntt_device_locked(c, field, layout, dir, (const char *)d_in + (size_t)b0 * s_in * eb,
                                        (char *)d_out + (size_t)b0 * s_out * eb, log2n, nb, stride, coset, stream, in_log2);
            if (rc2) return rc2;
        }
        return LW_OK;
    }
    if (in_log2 == 0 && log2n > 0)   // a constant polynomial: every evaluation is c_0 (times offset^0), no stage left to run
        return broadcast_device(lw_hip_field_elem_bytes(field, layout), d_in, d_out, 1ull << log2n, batch,
                                stride ? stride : ((uint64_t)1 << log2n), stream);
    if (field == LW_FIELD_BABYBEAR) return ntt_bb_device(c, layout, dir, d_in, d_out, log2n, batch, stride, coset, stream, in_log2);
    uint32_t cw[8];
    if (coset) words_from_ref(coset, cw);
    return ntt256_device(c, (int)field, dir, d_in, d_out, log2n, batch, stride, coset ? cw : nullptr, stream, in_log2);
}

// ---- pinned result buffers (lw_hip_result_acquire / release) ----
// hipHostMalloc'ed, i.e. resident and DMA-able: a device-to-host copy into one runs at the PCIe rate with no page fault.
// Released buffers are kept (up to 4) and handed out again to requests they fit within a factor of two.
struct HostPool {
    struct Buf { void *p; size_t bytes; bool in_use; };
    std::mutex mu;
    std::vector<Buf> bufs;
};
static HostPool g_host_pool;
bool host_pool_owns(const void *p) {
    std::lock_guard<std::mutex> g(g_host_pool.mu);
    for (auto &b : g_host_pool.bufs)
        if ((const char *)p >= (const char *)b.p && (const char *)p < (const char *)b.p + b.bytes) return true;
    return false;
}
void host_pool_release_all() {
    std::lock_guard<std::mutex> g(g_host_pool.mu);
    for (auto &b : g_host_pool.bufs) (void)hipHostFree(b.p);
    g_host_pool.bufs.clear();
}

// A fresh result buffer (a Rust `Vec::with_capacity`, numpy's `empty`) has never been touched: the device-to-host copy
// into it then runs at the kernel's page-fault rate (512 MiB = 131072 first-touch faults: 35-45 ms) instead of the PCIe rate
// (~9 ms).  The host-buffer entry points therefore (a) ask for transparent huge pages on the 2 MiB-aligned interior of a
// large output (madvise MADV_HUGEPAGE: 256 faults instead of 131072 where the host allows it, LW_HIP_HOST_THP=0 skips it),
// (b) populate it with a few threads, chunk by chunk in address order, WHILE the upload and the kernels run
// (MADV_POPULATE_WRITE maps the pages without changing their contents), and (c) copy each chunk back as soon as it is
// mapped, so the download runs behind the populate front instead of after it.  Callers that can hold results in a
// library buffer skip all of this: lw_hip_result_acquire hands out pinned, resident memory.
struct Prefault {
    static constexpr size_t CHUNK = (size_t)32 << 20;
    std::vector<std::thread> th;
    std::vector<std::atomic<int>> done;
    char *base = nullptr;
    size_t bytes = 0, nchunks = 0;
    bool active = false;
    void start(void *p, size_t nbytes, const void *in, size_t in_bytes) {
        const uintptr_t a = (uintptr_t)p, b = (uintptr_t)in;
        base = (char *)p;
        bytes = nbytes;
        if (nbytes < ((size_t)32 << 20) || (a < b + in_bytes && b < a + nbytes) || host_pool_owns(p)) return;   // small, aliases the input (already mapped), or pinned
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
#ifdef MADV_HUGEPAGE
        static const bool thp = [] { const char *e = tuning_env("LW_HIP_HOST_THP"); return !e || atoi(e) != 0; }();
        if (thp) {
            const uintptr_t H = (uintptr_t)2 << 20, hlo = (a + H - 1) & ~(H - 1), hhi = (a + nbytes) & ~(H - 1);
            if (hhi > hlo) (void)madvise((void *)hlo, hhi - hlo, MADV_HUGEPAGE);   // best effort
        }
#endif
        nchunks = (nbytes + CHUNK - 1) / CHUNK;
        done = std::vector<std::atomic<int>>(nchunks);
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
        unsigned T = std::thread::hardware_concurrency();
        T = T < 2 ? 1 : (T > 8 ? 8 : T);
        active = true;
        for (unsigned t = 0; t < T; t++) {
            try {   // nothing may propagate across the C ABI: without a helper thread the copy simply faults the pages itself
                th.emplace_back([this, t, T, page, a] {
                    for (size_t k = t; k < nchunks; k += T) {
                        const uintptr_t c0 = a + k * CHUNK, c1 = c0 + CHUNK < a + bytes ? c0 + CHUNK : a + bytes;
                        const uintptr_t lo = (c0 + page - 1) & ~(uintptr_t)(page - 1), hi = c1 & ~(uintptr_t)(page - 1);
#ifdef MADV_POPULATE_WRITE
                        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_POPULATE_WRITE);   // best effort
#endif
                        done[k].store(1, std::memory_order_release);
                    }
                });
            } catch (...) {
                for (size_t k = t; k < nchunks; k += T) done[k].store(1, std::memory_order_release);   // nobody will populate these
            }
        }
    }
    // dst = base + off: device -> host, chunk by chunk behind the populate front (one plain copy when nothing is being populated)
    int copy_back(const void *d_src, size_t nbytes, hipStream_t s) {
        if (!active) {
            LW_HIP_CHECK(hipMemcpyAsync(base, d_src, nbytes, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
            LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
            return LW_OK;
        }
        for (size_t k = 0; k < nchunks; k++) {
            while (!done[k].load(std::memory_order_acquire)) std::this_thread::yield();
            const size_t off = k * CHUNK, len = off + CHUNK < nbytes ? CHUNK : nbytes - off;
            LW_HIP_CHECK(hipMemcpyAsync(base + off, (const char *)d_src + off, len, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
            LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
        }
        return LW_OK;
    }
    void join() {
        for (auto &t : th) t.join();
        th.clear();
    }
    ~Prefault() { join(); }
};

}  // namespace lw

using namespace lw;

static bool elem_is_zero(const unsigned char *p, size_t eb);

extern "C" {

int lw_hip_init(const int *device_ids, int n_devices) {
    std::unique_lock<std::shared_mutex> g(shared_state().rw);   // no call is running
    return init_locked(device_ids, n_devices);
}

void lw_hip_shutdown(void) {
    SharedState &sh = shared_state();
    std::unique_lock<std::shared_mutex> g(sh.rw);
    if (!sh.initialised) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(sh.device);
    release_everything();
    if (prev >= 0) (void)hipSetDevice(prev);
    sh.initialised = false;
    for (int i = 0; i < LW_LANES; i++) lane(i).initialised = false;
}

int lw_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char *lw_hip_last_error(void) { return g_last_error.c_str(); }

// Exclusive access to every lane with the context's device bound: profile begin / end, timings.
struct AllLanes {
    std::unique_lock<std::shared_mutex> g;
    int rc = LW_OK, prev = -1;
    AllLanes() {
        rc = ensure_init();
        g = std::unique_lock<std::shared_mutex>(shared_state().rw);
        if (!rc && !shared_state().initialised) rc = LW_ERR_NO_DEVICE;
        if (rc) return;
        (void)hipGetDevice(&prev);
        if (prev != shared_state().device) (void)hipSetDevice(shared_state().device);
    }
    ~AllLanes() {
        if (!rc && prev >= 0 && prev != shared_state().device) (void)hipSetDevice(prev);
    }
};

int lw_hip_profile_begin(void) {
    AllLanes all;
    if (all.rc) return all.rc;
    for (int i = 0; i < LW_LANES; i++) {
        Context &c = lane(i);
        for (auto &sp : c.spans) {   // a begin without an end: recycle the pending events
            c.event_pool.push_back(sp.e0);
            c.event_pool.push_back(sp.e1);
        }
        c.spans.clear();
        c.profiling = true;
    }
    return LW_OK;
}

int lw_hip_profile_end(lw_profile_t *out) {
    if (!out) return LW_ERR_BAD_ARG;
    AllLanes all;
    if (all.rc) return all.rc;
    memset(out, 0, sizeof(*out));
    LW_HIP_CHECK(hipDeviceSynchronize(), LW_ERR_LAUNCH);
    for (int i = 0; i < LW_LANES; i++) {
        Context &c = lane(i);
        c.profiling = false;
        for (auto &sp : c.spans) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, sp.e0, sp.e1);
            int idx = -1;
            for (int k = 0; k < out->n; k++)
                if (strcmp(out->k[k].name, sp.name) == 0) idx = k;
            if (idx < 0 && out->n < (int)(sizeof(out->k) / sizeof(out->k[0]))) {
                idx = out->n++;
                strncpy(out->k[idx].name, sp.name, sizeof(out->k[idx].name) - 1);
            }
            if (idx >= 0) {
                out->k[idx].launches++;
                out->k[idx].total_ms += ms;
            }
            c.event_pool.push_back(sp.e0);
            c.event_pool.push_back(sp.e1);
        }
        c.spans.clear();
    }
    return LW_OK;
}

int lw_hip_get_timings(lw_timings_t *out) {
    if (!out) return LW_ERR_BAD_ARG;
    std::unique_lock<std::shared_mutex> g(shared_state().rw);
    lw_timings_t t = {};
    for (int i = 0; i < LW_LANES; i++) {   // calls summed over the lanes; "last" = the longest of the lanes' last calls
        const lw_timings_t &l = lane(i).timings;
        t.ntt_calls += l.ntt_calls;
        t.msm_calls += l.msm_calls;
        t.last_ntt_ms = l.last_ntt_ms > t.last_ntt_ms ? l.last_ntt_ms : t.last_ntt_ms;
        t.last_msm_ms = l.last_msm_ms > t.last_msm_ms ? l.last_msm_ms : t.last_msm_ms;
        t.scratch_bytes += lane(i).scratch.bytes;
    }
    for (int f = 0; f < 3; f++)
        for (int d = 0; d < 2; d++) t.twiddle_bytes += shared_state().tw[f][d].valid ? shared_state().tw[f][d].buf.bytes : 0;
    *out = t;
    return LW_OK;
}

size_t lw_hip_field_elem_bytes(lw_field_t field, lw_layout_t layout) {
    if (field == LW_FIELD_STARK252 || field == LW_FIELD_BLS12_381_FR) return 32;
    switch (layout) {
        case LW_LAYOUT_BABYBEAR_U32_R32: return 4;
        case LW_LAYOUT_BABYBEAR_U64_R64: return 8;
        case LW_LAYOUT_EXT4_INTERLEAVED: return 32;
        default: return 0;
    }
}

size_t lw_hip_curve_point_bytes(lw_curve_t curve) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return 144;
        case LW_CURVE_BN254_G1: return 96;
        case LW_CURVE_BN254_G2: return 192;
        case LW_CURVE_BLS12_381_G2: return 288;
        default: return 0;
    }
}

int lw_hip_result_acquire(size_t bytes, void **out_ptr) {
    if (!out_ptr || bytes == 0) { set_error("null or empty request"); return LW_ERR_BAD_ARG; }
    Entry en(nullptr);   // binds the context's device: pinned memory is registered with it
    if (en.rc) return en.rc;
    {
        std::lock_guard<std::mutex> g(g_host_pool.mu);
        HostPool::Buf *best = nullptr;
        for (auto &b : g_host_pool.bufs)
            if (!b.in_use && b.bytes >= bytes && b.bytes / 2 <= bytes && (!best || b.bytes < best->bytes)) best = &b;
        if (best) { best->in_use = true; *out_ptr = best->p; return LW_OK; }
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return LW_ERR_ALLOC; }
    std::lock_guard<std::mutex> g(g_host_pool.mu);
    g_host_pool.bufs.push_back(HostPool::Buf{p, bytes, true});
    *out_ptr = p;
    return LW_OK;
}

int lw_hip_result_release(void *ptr) {
    if (!ptr) return LW_OK;
    void *to_free = nullptr;
    {
        std::lock_guard<std::mutex> g(g_host_pool.mu);
        size_t idle = 0;
        HostPool::Buf *mine = nullptr;
        for (auto &b : g_host_pool.bufs) {
            if (b.p == ptr) mine = &b;
            else if (!b.in_use) idle++;
        }
        if (!mine || !mine->in_use) { set_error("pointer was not handed out by lw_hip_result_acquire"); return LW_ERR_BAD_ARG; }
        mine->in_use = false;
        if (idle >= 4) {   // keep at most four idle buffers
            to_free = mine->p;
            g_host_pool.bufs.erase(g_host_pool.bufs.begin() + (mine - g_host_pool.bufs.data()));
        }
    }
    if (to_free) (void)hipHostFree(to_free);
    return LW_OK;
}

int lw_hip_ntt_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2n,
                      uint32_t batch, size_t batch_stride_elems, const void *coset_offset_or_null, void *hip_stream) {
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    auto t0 = std::chrono::steady_clock::now();
    rc = ntt_device_locked(c, field, layout, dir, d_in, d_out, log2n, batch, batch_stride_elems, coset_offset_or_null,
                           (hipStream_t)hip_stream);
    c.timings.last_ntt_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.ntt_calls++;
    return rc;
}

// gen_twiddles (math/src/fft/gpu/cuda/ops.rs:45-66 -> math/src/fft/cpu/roots_of_unity.rs:66-75)
int lw_hip_gen_twiddles(lw_field_t field, lw_layout_t layout, uint64_t order, int config, void *out) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (order > 63) { set_error("Order should be less than or equal to 63, but is %llu", (unsigned long long)order); return LW_ERR_ORDER_TOO_LARGE; }
    if (config < 0 || config > 3) { set_error("bad roots config %d", config); return LW_ERR_BAD_ARG; }
    if (order > field_two_adicity(field)) { set_error("no primitive 2^%llu-th root of unity in this field", (unsigned long long)order); return LW_ERR_ROOT_OF_UNITY; }
    const uint64_t count = (1ull << order) / 2;
    if (count == 0) return LW_OK;
    if (!out) { set_error("null buffer"); return LW_ERR_BAD_ARG; }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    // twiddles live in the domain field: one base word per entry for every BabyBear shape
    const size_t eb = field == LW_FIELD_BABYBEAR ? (layout == LW_LAYOUT_BABYBEAR_U32_R32 ? 4 : 8) : 32;
    if (c.host_io_b.ensure(count * eb)) return LW_ERR_ALLOC;
    rc = gen_twiddles_device(c, field, layout, (uint32_t)order, config, c.host_io_b.p, 0);
    if (rc) return rc;
    LW_HIP_CHECK(hipMemcpy(out, c.host_io_b.p, count * eb, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    return LW_OK;
}

// get_powers_of_primitive_root / get_powers_of_primitive_root_coset (math/src/fft/cpu/roots_of_unity.rs:13-61)
int lw_hip_gen_powers(lw_field_t field, lw_layout_t layout, uint64_t order, size_t count, int config, const void *offset_or_null,
                      void *out, size_t *out_len) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (config < 0 || config > 3) { set_error("bad roots config %d", config); return LW_ERR_BAD_ARG; }
    if (offset_or_null && config != 0) { set_error("the coset variant is defined for the Natural configuration only"); return LW_ERR_BAD_ARG; }
    if (out_len) *out_len = 0;
    if (count == 0) return LW_OK;   // roots_of_unity.rs:18-20: nothing computed, not even the root
    if (order > field_two_adicity(field)) { set_error("no primitive 2^%llu-th root of unity in this field", (unsigned long long)order); return LW_ERR_ROOT_OF_UNITY; }
    const bool bitrev = config >= 2, inverse = (config & 1) != 0;
    size_t up_to = count;
    uint32_t bits = 0;
    if (bitrev) {   // "in bit reverse form we could need as many as (1 << count.bits()) - 1 roots": the result has next_power_of_two(count) entries
        while (((size_t)1 << bits) < count) bits++;
        up_to = (size_t)1 << bits;
    }
    if (out_len) *out_len = up_to;
    if (!out) return LW_OK;       // size query
    if (up_to >> 32) { set_error("%zu powers exceed the 32-bit index range", up_to); return LW_ERR_ALLOC; }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    const size_t eb = field == LW_FIELD_BABYBEAR ? (layout == LW_LAYOUT_BABYBEAR_U32_R32 ? 4 : 8) : 32;   // domain-field words
    if (c.host_io_b.ensure(up_to * eb)) return LW_ERR_ALLOC;
    if (field == LW_FIELD_BABYBEAR) {
        rc = ntt_bb_gen_powers(layout, (uint32_t)order, up_to, bitrev ? bits : 0, inverse, offset_or_null, c.host_io_b.p, 0);
    } else {
        uint32_t ow[8];
        if (offset_or_null) words_from_ref(offset_or_null, ow);
        rc = ntt256_gen_powers((int)field, (uint32_t)order, up_to, bitrev ? bits : 0, inverse, offset_or_null ? ow : nullptr, c.host_io_b.p, 0);
    }
    if (rc) return rc;
    LW_HIP_CHECK(hipMemcpy(out, c.host_io_b.p, up_to * eb, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    return LW_OK;
}

// bitrev_permutation (math/src/fft/gpu/cuda/ops.rs:68-77): out[i] = in[bitrev(i)]; in may alias out
int lw_hip_bitrev_permutation(lw_field_t field, lw_layout_t layout, const void *in, void *out, size_t n) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (n == 0) return LW_OK;
    if (n & (n - 1)) { set_error("Input length is %zu, which is not a power of two", n); return LW_ERR_INPUT_NOT_POW2; }
    if (!in || !out) { set_error("null buffer"); return LW_ERR_BAD_ARG; }
    uint32_t log2n = 0;
    while (((size_t)1 << log2n) < n) log2n++;
    if (log2n > 32) { set_error("2^%u elements exceed device memory", log2n); return LW_ERR_ALLOC; }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    const size_t eb = lw_hip_field_elem_bytes(field, layout);
    if (c.host_io_a.ensure(n * eb) || c.host_io_b.ensure(n * eb)) return LW_ERR_ALLOC;
    LW_HIP_CHECK(hipMemcpy(c.host_io_a.p, in, n * eb, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
    rc = bitrev_device(eb, c.host_io_a.p, c.host_io_b.p, log2n, 0);
    if (rc) return rc;
    LW_HIP_CHECK(hipMemcpy(out, c.host_io_b.p, n * eb, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    return LW_OK;
}

int lw_hip_ntt_cross_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                            uint32_t log2n_total, uint32_t log2_shards, uint64_t j2_begin, uint64_t slice_len,
                            uint64_t chunk_stride_elems, uint32_t batch, uint64_t batch_stride_elems, void *hip_stream) {
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (log2n_total > 63) { set_error("order %u > 63", log2n_total); return LW_ERR_ORDER_TOO_LARGE; }
    if (log2n_total > field_two_adicity(field)) { set_error("no primitive 2^%u-th root of unity in this field", log2n_total); return LW_ERR_ROOT_OF_UNITY; }
    if (!d_in || !d_out || d_in == d_out) { set_error("cross step needs distinct non-null buffers"); return LW_ERR_BAD_ARG; }
    if (batch == 0 || slice_len == 0) return LW_OK;
    return ntt_cross_device(c, field, layout, dir, d_in, d_out, log2n_total, log2_shards, j2_begin, slice_len, chunk_stride_elems,
                            batch, batch_stride_elems, (hipStream_t)hip_stream);
}

int lw_hip_ntt_lde_device(lw_field_t field, lw_layout_t layout, const void *d_coeffs, uint32_t log2_coeffs, void *d_out,
                          uint32_t log2n, uint32_t batch, const void *coset_offset_or_null, void *hip_stream) {
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    if (log2_coeffs > log2n) { set_error("2^%u coefficients do not fit a 2^%u domain", log2_coeffs, log2n); return LW_ERR_BAD_ARG; }
    auto t0 = std::chrono::steady_clock::now();
    rc = ntt_device_locked(c, field, layout, LW_DIR_FORWARD, d_coeffs, d_out, log2n, batch, 0, coset_offset_or_null,
                           (hipStream_t)hip_stream, log2_coeffs);
    c.timings.last_ntt_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.ntt_calls++;
    return rc;
}

int lw_hip_ntt(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *in, void *out, uint32_t log2n, uint32_t batch,
               size_t batch_stride_elems, const void *coset_offset_or_null) {
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (log2n > 63) { set_error("order %u > 63", log2n); return LW_ERR_ORDER_TOO_LARGE; }
    if (log2n > field_two_adicity(field)) { set_error("no primitive 2^%u-th root of unity in this field", log2n); return LW_ERR_ROOT_OF_UNITY; }
    if (batch == 0) return LW_OK;
    if (!in || !out) { set_error("null buffer"); return LW_ERR_BAD_ARG; }
    auto t0 = std::chrono::steady_clock::now();
    const size_t eb = lw_hip_field_elem_bytes(field, layout);
    const size_t n = (size_t)1 << log2n;
    const size_t stride = batch_stride_elems ? batch_stride_elems : n;
    if (stride < n) { set_error("batch stride %zu < transform length", stride); return LW_ERR_BAD_ARG; }
    const size_t span = ((size_t)(batch - 1) * stride + n) * eb;
    if (c.host_io_a.ensure(span) || c.host_io_b.ensure(span)) return LW_ERR_ALLOC;
    hipStream_t io = en.use_lane_stream();
    if (!io) return en.rc;
    Prefault pf;
    pf.start(out, span, in, span);
    LW_HIP_CHECK(hipMemcpyAsync(c.host_io_a.p, in, span, hipMemcpyHostToDevice, io), LW_ERR_LAUNCH);
    rc = ntt_device_locked(c, field, layout, dir, c.host_io_a.p, c.host_io_b.p, log2n, batch, stride, coset_offset_or_null, io);
    if (rc) return rc;
    LW_HIP_CHECK(hipStreamSynchronize(io), LW_ERR_LAUNCH);
    if (stride == n) {
        rc = pf.copy_back(c.host_io_b.p, span, io);
        if (rc) return rc;
    } else {   // leave the gaps between strided transforms untouched
        pf.join();
        LW_HIP_CHECK(hipMemcpy2DAsync(out, stride * eb, c.host_io_b.p, stride * eb, n * eb, batch, hipMemcpyDeviceToHost, io), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipStreamSynchronize(io), LW_ERR_LAUNCH);
    }
    c.timings.last_ntt_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.ntt_calls++;
    return LW_OK;
}

static bool elem_is_zero(const unsigned char *p, size_t eb) {
    for (size_t i = 0; i < eb; i++)
        if (p[i]) return false;
    return true;
}

// Polynomial::evaluate_fft / evaluate_offset_fft (math/src/fft/polynomial.rs:25-68,74-82)
int lw_polynomial_evaluate_fft(lw_field_t field, lw_layout_t layout, const void *coeffs, size_t n_coeffs, size_t blowup_factor,
                               size_t domain_size, const void *offset_or_null, void *out, size_t out_capacity_elems,
                               size_t *out_len) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (!out_len || (n_coeffs && !coeffs)) { set_error("null argument"); return LW_ERR_BAD_ARG; }
    const size_t eb = lw_hip_field_elem_bytes(field, layout);
    // Polynomial::new strips trailing zero coefficients (math/src/polynomial/mod.rs:19-31)
    size_t clen = n_coeffs;
    while (clen > 0 && elem_is_zero((const unsigned char *)coeffs + (clen - 1) * eb, eb)) clen--;
    size_t m = clen > domain_size ? clen : domain_size;
    size_t p2 = 1;
    while (p2 < m) p2 <<= 1;
    const size_t len = p2 * blowup_factor;
    *out_len = len;
    if (!out) return LW_OK;
    if (out_capacity_elems < len) { set_error("output capacity %zu < %zu", out_capacity_elems, len); return LW_ERR_BAD_ARG; }
    if (clen == 0) {   // zero polynomial: len zeros, no transform (:33-35)
        memset(out, 0, len * eb);
        return LW_OK;
    }
    if (len == 0 || (len & (len - 1))) {   // ops::fft rejects it (math/src/fft/cpu/ops.rs:17-19)
        set_error("Input length is %zu, which is not a power of two", len);
        return LW_ERR_INPUT_NOT_POW2;
    }
    uint32_t log2n = 0;
    while (((size_t)1 << log2n) < len) log2n++;
    if (log2n > field_two_adicity(field)) { set_error("no primitive 2^%u-th root of unity in this field", log2n); return LW_ERR_ROOT_OF_UNITY; }

    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    auto t0 = std::chrono::steady_clock::now();
    // Zero padding happens after scaling in the reference, so padded slots stay zero either way.  Only the power-of-two
    // block that holds the coefficients is uploaded; the transform extends it (LDE path).
    size_t block = 1;
    while (block < clen) block <<= 1;
    uint32_t in_log2 = 0;
    while (((size_t)1 << in_log2) < block) in_log2++;
    const bool lde = in_log2 >= 1 && in_log2 < log2n;
    const size_t up = lde ? block : len;
    if (c.host_io_a.ensure(up * eb) || c.host_io_b.ensure(len * eb)) return LW_ERR_ALLOC;
    hipStream_t io = en.use_lane_stream();
    if (!io) return en.rc;
    Prefault pf;
    pf.start(out, len * eb, coeffs, n_coeffs * eb);
    LW_HIP_CHECK(hipMemsetAsync(c.host_io_a.p, 0, up * eb, io), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipMemcpyAsync(c.host_io_a.p, coeffs, clen * eb, hipMemcpyHostToDevice, io), LW_ERR_LAUNCH);
    rc = ntt_device_locked(c, field, layout, LW_DIR_FORWARD, c.host_io_a.p, c.host_io_b.p, log2n, 1, len, offset_or_null, io,
                           lde ? in_log2 : log2n);
    if (rc) return rc;
    LW_HIP_CHECK(hipStreamSynchronize(io), LW_ERR_LAUNCH);
    rc = pf.copy_back(c.host_io_b.p, len * eb, io);
    if (rc) return rc;
    c.timings.last_ntt_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.ntt_calls++;
    return LW_OK;
}

// Polynomial::interpolate_fft / interpolate_offset_fft (math/src/fft/polynomial.rs:87-127,159-174)
int lw_polynomial_interpolate_fft(lw_field_t field, lw_layout_t layout, const void *evals, size_t n, const void *offset_or_null,
                                  void *out_coeffs, size_t *coeff_len) {
    int rc = check_field_layout(field, layout);
    if (rc) return rc;
    if (!evals || !out_coeffs) { set_error("null argument"); return LW_ERR_BAD_ARG; }
    if (n == 0 || (n & (n - 1))) {
        set_error("Input length is %zu, which is not a power of two", n);
        return LW_ERR_INPUT_NOT_POW2;
    }
    uint32_t log2n = 0;
    while (((size_t)1 << log2n) < n) log2n++;
    rc = lw_hip_ntt(field, layout, LW_DIR_INVERSE, evals, out_coeffs, log2n, 1, 0, offset_or_null);
    if (rc) return rc;
    if (coeff_len) {
        const size_t eb = lw_hip_field_elem_bytes(field, layout);
        size_t clen = n;
        while (clen > 0 && elem_is_zero((const unsigned char *)out_coeffs + (clen - 1) * eb, eb)) clen--;
        *coeff_len = clen;
    }
    return LW_OK;
}

static int msm_device_entry(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, void *out_point_host,
                            void *hip_stream, int mont) {
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    if (lw_hip_curve_point_bytes(curve) == 0 || !out_point_host) { set_error("bad curve or null output"); return LW_ERR_BAD_ARG; }
    auto t0 = std::chrono::steady_clock::now();
    rc = msm_device(c, curve, d_scalars, d_points, n, out_point_host, (hipStream_t)hip_stream, mont, 0);
    c.timings.last_msm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.msm_calls++;
    return rc;
}
// interpolate_and_commit_main's commitment step (provers/stark/src/prover.rs:229-244) on device-resident LDE columns
// the element size the commitment hashes, or 0 when the (field, layout) pair has no AsBytes in the reference
static uint32_t commit_elem_bytes(lw_field_t field, lw_layout_t layout) {
    if (check_field_layout(field, layout)) return 0;
    if (layout == LW_LAYOUT_EXT4_INTERLEAVED) { set_error("the quartic extension has no AsBytes in the reference (quartic_babybear.rs)"); return 0; }
    return (uint32_t)lw_hip_field_elem_bytes(field, layout);
}
// A Merkle root for the caller's transcript: through the lane's pinned words (a 32-byte copy into pageable memory goes
// through the runtime's staging path, ~2x the latency; FRI reads one root per layer with the GPU idle meanwhile)
static int read_root(Context &c, const void *d_nodes, uint8_t *out_root, hipStream_t s) {
    if (!c.pinned_words) LW_HIP_CHECK(hipHostMalloc((void **)&c.pinned_words, 256, hipHostMallocDefault), LW_ERR_ALLOC);
    uint32_t *stage = c.pinned_words + 32;   // words 0..31: the MSM's small results (msm_core.cuh)
    LW_HIP_CHECK(hipMemcpyAsync(stage, d_nodes, 32, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
    memcpy(out_root, stage, 32);
    return LW_OK;
}
int lw_stark_commit_columns_layout_device(lw_field_t field, lw_layout_t layout, const void *d_columns, uint32_t n_cols, uint64_t col_stride_elems,
                                          uint32_t log2n, int bit_reverse, void *d_nodes, uint8_t *out_root, void *hip_stream) {
    const uint32_t eb = commit_elem_bytes(field, layout);
    if (!eb) return LW_ERR_BAD_ARG;
    if (!d_columns || !d_nodes || n_cols == 0) { set_error("null buffer or no columns"); return LW_ERR_BAD_ARG; }
    if (log2n > 31) { set_error("2^%u leaves", log2n); return LW_ERR_ALLOC; }
    if ((uint64_t)n_cols * eb >= (1ull << 31)) { set_error("%u columns per row", n_cols); return LW_ERR_BAD_ARG; }
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    if (col_stride_elems == 0) col_stride_elems = 1ull << log2n;
    int rc = merkle_commit_device(c, d_columns, n_cols, col_stride_elems, log2n, bit_reverse, d_nodes, (hipStream_t)hip_stream, eb);
    if (rc) return rc;
    if (out_root) return read_root(c, d_nodes, out_root, (hipStream_t)hip_stream);
    return LW_OK;
}
int lw_stark_commit_columns_device(lw_field_t field, const void *d_columns, uint32_t n_cols, uint64_t col_stride_elems, uint32_t log2n,
                                   int bit_reverse, void *d_nodes, uint8_t *out_root, void *hip_stream) {
    if (field != LW_FIELD_STARK252 && field != LW_FIELD_BLS12_381_FR) { set_error("Merkle commitment supports the 256-bit fields"); return LW_ERR_BAD_ARG; }
    if (!d_columns || !d_nodes || n_cols == 0) { set_error("null buffer or no columns"); return LW_ERR_BAD_ARG; }
    if (log2n > 31) { set_error("2^%u leaves", log2n); return LW_ERR_ALLOC; }
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    if (col_stride_elems == 0) col_stride_elems = 1ull << log2n;
    rc = merkle_commit_device(c, d_columns, n_cols, col_stride_elems, log2n, bit_reverse, d_nodes, (hipStream_t)hip_stream);
    if (rc) return rc;
    if (out_root) return read_root(c, d_nodes, out_root, (hipStream_t)hip_stream);
    return LW_OK;
}

int lw_stark_commit_columns(lw_field_t field, const void *columns, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *out_root,
                            uint8_t *out_nodes_or_null) {
    if (field != LW_FIELD_STARK252 && field != LW_FIELD_BLS12_381_FR) { set_error("Merkle commitment supports the 256-bit fields"); return LW_ERR_BAD_ARG; }
    if (!columns || !out_root || n_cols == 0) { set_error("null buffer or no columns"); return LW_ERR_BAD_ARG; }
    if (log2n > 31) { set_error("2^%u leaves", log2n); return LW_ERR_ALLOC; }
    const size_t n = (size_t)1 << log2n;
    {
        Entry en(nullptr);
        if (en.rc) return en.rc;
        Context &c = en.c;
        int rc = LW_OK;
        if (c.host_io_a.ensure((size_t)n_cols * n * 32) || c.host_io_b.ensure((2 * n - 1) * 32)) return LW_ERR_ALLOC;
        LW_HIP_CHECK(hipMemcpy(c.host_io_a.p, columns, (size_t)n_cols * n * 32, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
        rc = merkle_commit_device(c, c.host_io_a.p, n_cols, n, log2n, bit_reverse, c.host_io_b.p, 0);
        if (rc) return rc;
        LW_HIP_CHECK(hipMemcpy(out_root, c.host_io_b.p, 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
        if (out_nodes_or_null) LW_HIP_CHECK(hipMemcpy(out_nodes_or_null, c.host_io_b.p, (2 * n - 1) * 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    }
    return LW_OK;
}

// One layer of the FRI commit phase (provers/stark/src/fri/mod.rs:44-58 + :115-141), host buffers
int lw_stark_fri_layer(lw_field_t field, const void *coeffs, size_t n_coeffs, const void *zeta, const void *coset_offset, size_t domain_size,
                       void *out_poly, size_t *out_poly_len, void *out_evaluation, uint8_t *out_root, uint8_t *out_nodes_or_null) {
    if (field != LW_FIELD_STARK252 && field != LW_FIELD_BLS12_381_FR) { set_error("FRI layer supports the 256-bit fields"); return LW_ERR_BAD_ARG; }
    if (!coeffs || !zeta || !coset_offset || !out_poly || !out_evaluation || !out_root || n_coeffs == 0) { set_error("null or empty argument"); return LW_ERR_BAD_ARG; }
    if (domain_size < 2 || (domain_size & (domain_size - 1))) {
        set_error("Input length is %zu, which is not a power of two", domain_size);
        return LW_ERR_INPUT_NOT_POW2;
    }
    const size_t n_out = (n_coeffs + 1) / 2;
    if (n_out > domain_size) { set_error("folded polynomial of %zu coefficients exceeds the domain %zu", n_out, domain_size); return LW_ERR_BAD_ARG; }
    uint32_t lgd = 0, lgb = 1;
    while (((size_t)1 << lgd) < domain_size) lgd++;
    while (((size_t)1 << lgb) < n_out) lgb++;
    if (lgb > lgd) lgb = lgd;
    if (lgd > field_two_adicity(field)) { set_error("no primitive 2^%u-th root of unity in this field", lgd); return LW_ERR_ROOT_OF_UNITY; }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    const size_t blk = (size_t)1 << lgb;
    // a: [coeffs | zeta words | folded block]   b: [eval | eval_br | nodes]
    const size_t a_coeffs = n_coeffs * 32, a_zeta = 256, a_poly = blk * 32;
    if (c.host_io_a.ensure(a_coeffs + a_zeta + a_poly) || c.host_io_b.ensure(2 * domain_size * 32 + (domain_size - 1) * 32)) return LW_ERR_ALLOC;
    char *da = (char *)c.host_io_a.p, *db = (char *)c.host_io_b.p;
    uint32_t zw[8];
    words_from_ref(zeta, zw);
    LW_HIP_CHECK(hipMemcpy(da, coeffs, a_coeffs, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
    char *d_poly = da + a_coeffs + a_zeta;
    char *d_eval = db, *d_eval_br = db + domain_size * 32, *d_nodes = db + 2 * domain_size * 32;
    rc = fri_layer_device(c, field, da, n_coeffs, zw, coset_offset, lgd, d_poly, lgb, d_eval, d_eval_br, d_nodes, 0);
    if (rc) return rc;
    LW_HIP_CHECK(hipMemcpy(out_poly, d_poly, n_out * 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipMemcpy(out_evaluation, d_eval_br, domain_size * 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipMemcpy(out_root, d_nodes, 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    if (out_nodes_or_null) LW_HIP_CHECK(hipMemcpy(out_nodes_or_null, d_nodes, (domain_size - 1) * 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    if (out_poly_len) {   // Polynomial::new strips trailing zeros of the folded polynomial
        size_t clen = n_out;
        while (clen > 0 && elem_is_zero((const unsigned char *)out_poly + (clen - 1) * 32, 32)) clen--;
        *out_poly_len = clen;
    }
    return LW_OK;
}

// The same layer on device-resident buffers: only zeta (in) and the 32-byte root (out) cross the bus, which is all the
// transcript between two layers of commit_phase needs (provers/stark/src/fri/mod.rs:44-58).
int lw_stark_fri_layer_device(lw_field_t field, const void *d_coeffs, size_t n_coeffs, const void *zeta, const void *coset_offset,
                              size_t domain_size, void *d_out_poly, void *d_out_evaluation_or_null, void *d_nodes_or_null,
                              uint8_t *out_root_or_null, void *hip_stream) {
    if (field != LW_FIELD_STARK252 && field != LW_FIELD_BLS12_381_FR) { set_error("FRI layer supports the 256-bit fields"); return LW_ERR_BAD_ARG; }
    if (!d_coeffs || !zeta || !d_out_poly || n_coeffs == 0 || d_out_poly == d_coeffs) { set_error("null, empty or aliased argument"); return LW_ERR_BAD_ARG; }
    const bool layer = d_nodes_or_null != nullptr;
    if (!layer && (d_out_evaluation_or_null || out_root_or_null)) { set_error("an evaluation or a root needs d_nodes"); return LW_ERR_BAD_ARG; }
    const size_t n_out = (n_coeffs + 1) / 2;
    uint32_t lgd = 0, lgb = 1;
    while (((size_t)1 << lgb) < n_out) lgb++;
    if (layer) {
        if (!coset_offset) { set_error("null coset offset"); return LW_ERR_BAD_ARG; }
        if (domain_size < 2 || (domain_size & (domain_size - 1))) {
            set_error("Input length is %zu, which is not a power of two", domain_size);
            return LW_ERR_INPUT_NOT_POW2;
        }
        while (((size_t)1 << lgd) < domain_size) lgd++;
        if (((size_t)1 << lgb) > domain_size) { set_error("folded polynomial of %zu coefficients exceeds the domain %zu", n_out, domain_size); return LW_ERR_BAD_ARG; }
        if (lgd > field_two_adicity(field)) { set_error("no primitive 2^%u-th root of unity in this field", lgd); return LW_ERR_ROOT_OF_UNITY; }
    }
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    // natural-order evaluation (the tree and the bit-reversed copy are made from it): library scratch
    if (layer && c.pipe_tmp.ensure(domain_size * 32)) return LW_ERR_ALLOC;
    uint32_t zw[8];
    words_from_ref(zeta, zw);
    int rc = fri_layer_device(c, field, d_coeffs, n_coeffs, zw, coset_offset, lgd, d_out_poly, lgb, layer ? c.pipe_tmp.p : nullptr,
                              d_out_evaluation_or_null, d_nodes_or_null, en.stream);
    if (rc) return rc;
    if (out_root_or_null) return read_root(c, d_nodes_or_null, out_root_or_null, en.stream);
    return LW_OK;
}

// calculate_h_coefficients on device-resident coefficient vectors; h stays in HBM for the MSM that consumes it
// (provers/groth16/src/prover.rs:68-72,97-101 -> lw_hip_msm_srs_fr_device)
int lw_groth16_h_coefficients_device(const void *d_l, const void *d_r, const void *d_o, size_t n_coeffs, size_t num_gates, void *d_out_h,
                                     size_t *coeff_len_or_null, void *hip_stream) {
    if (!d_out_h || (n_coeffs && (!d_l || !d_r || !d_o))) { set_error("null argument"); return LW_ERR_BAD_ARG; }
    if (num_gates < 1 || (num_gates & (num_gates - 1))) {
        set_error("Input length is %zu, which is not a power of two", num_gates);
        return LW_ERR_INPUT_NOT_POW2;
    }
    if (n_coeffs > num_gates) { set_error("%zu coefficients for %zu gates", n_coeffs, num_gates); return LW_ERR_BAD_ARG; }
    uint32_t lg = 0;
    while (((size_t)1 << lg) < num_gates) lg++;
    if (lg + 1 > Fr381::TWO_ADICITY) { set_error("no primitive 2^%u-th root of unity in this field", lg + 1); return LW_ERR_ROOT_OF_UNITY; }
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    const size_t n = 2 * num_gates, blk = num_gates < 2 ? 2 : num_gates;
    const bool staged = n_coeffs != blk;   // shorter vectors are zero padded into library scratch first
    if (c.pipe_tmp.ensure((3 * n + (staged ? 3 * blk : 0)) * 32 + 256)) return LW_ERR_ALLOC;
    char *ev = (char *)c.pipe_tmp.p, *st = ev + 3 * n * 32;
    const void *src[3] = {d_l, d_r, d_o};
    if (staged) {
        LW_HIP_CHECK(hipMemsetAsync(st, 0, 3 * blk * 32, en.stream), LW_ERR_LAUNCH);
        for (int k = 0; k < 3; k++) {
            if (n_coeffs) LW_HIP_CHECK(hipMemcpyAsync(st + (size_t)k * blk * 32, src[k], n_coeffs * 32, hipMemcpyDeviceToDevice, en.stream), LW_ERR_LAUNCH);
            src[k] = st + (size_t)k * blk * 32;
        }
    }
    int rc = groth16_h_device(c, src[0], src[1], src[2], lg, d_out_h, ev, en.stream);
    if (rc) return rc;
    if (coeff_len_or_null) {   // Polynomial::new's stripped length, for callers that slice the SRS by it (prover.rs:98-100)
        uint64_t *d_len = (uint64_t *)(ev + 3 * n * 32 + (staged ? 3 * blk * 32 : 0));
        rc = stripped_length_device(d_out_h, n, d_len, en.stream);
        if (rc) return rc;
        uint64_t len = 0;
        LW_HIP_CHECK(hipMemcpyAsync(&len, d_len, 8, hipMemcpyDeviceToHost, en.stream), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipStreamSynchronize(en.stream), LW_ERR_LAUNCH);
        *coeff_len_or_null = (size_t)len;
    }
    return LW_OK;
}

// QuadraticArithmeticProgram::calculate_h_coefficients (provers/groth16/src/qap.rs:15-39) after the variable
// polynomials have been accumulated: three coset LDEs, (l*r - o) / t pointwise, one coset INTT — one device pipeline.
int lw_groth16_h_coefficients(const void *l_coeffs, const void *r_coeffs, const void *o_coeffs, size_t n_coeffs, size_t num_gates,
                              void *out_h, size_t *coeff_len) {
    if (!l_coeffs || !r_coeffs || !o_coeffs || !out_h) { set_error("null argument"); return LW_ERR_BAD_ARG; }
    if (num_gates < 1 || (num_gates & (num_gates - 1))) {   // from_r1cs pads the gate count to a power of two (qap.rs:72-75)
        set_error("Input length is %zu, which is not a power of two", num_gates);
        return LW_ERR_INPUT_NOT_POW2;
    }
    if (n_coeffs > num_gates) { set_error("%zu coefficients for %zu gates", n_coeffs, num_gates); return LW_ERR_BAD_ARG; }
    uint32_t lg = 0;
    while (((size_t)1 << lg) < num_gates) lg++;
    if (lg + 1 > Fr381::TWO_ADICITY) { set_error("no primitive 2^%u-th root of unity in this field", lg + 1); return LW_ERR_ROOT_OF_UNITY; }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    const size_t n = 2 * num_gates;
    const size_t blk = num_gates < 2 ? 2 : num_gates;   // coefficient block per polynomial on the device (zero padded)
    // device staging: [l | r | o] coefficient blocks, then 3 evaluation vectors + output
    if (c.host_io_a.ensure(3 * blk * 32) || c.host_io_b.ensure(4 * n * 32)) return LW_ERR_ALLOC;
    LW_HIP_CHECK(hipMemsetAsync(c.host_io_a.p, 0, 3 * blk * 32, 0), LW_ERR_LAUNCH);
    const void *src[3] = {l_coeffs, r_coeffs, o_coeffs};
    for (int k = 0; k < 3; k++)
        if (n_coeffs)
            LW_HIP_CHECK(hipMemcpy((char *)c.host_io_a.p + (size_t)k * blk * 32, src[k], n_coeffs * 32, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
    char *ev = (char *)c.host_io_b.p;
    rc = groth16_h_device(c, c.host_io_a.p, (char *)c.host_io_a.p + blk * 32, (char *)c.host_io_a.p + 2 * blk * 32, lg,
                          ev + 3 * n * 32, ev, 0);
    if (rc) return rc;
    LW_HIP_CHECK(hipMemcpy(out_h, ev + 3 * n * 32, n * 32, hipMemcpyDeviceToHost), LW_ERR_LAUNCH);
    if (coeff_len) {
        size_t clen = n;
        while (clen > 0 && elem_is_zero((const unsigned char *)out_h + (clen - 1) * 32, 32)) clen--;
        *coeff_len = clen;
    }
    return LW_OK;
}

int lw_hip_msm_device(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, void *out_point_host,
                      void *hip_stream) {
    return msm_device_entry(curve, d_scalars, d_points, n, out_point_host, hip_stream, 0);
}
int lw_hip_msm_fr_device(lw_curve_t curve, const uint64_t *d_fr_elements, const void *d_points, size_t n, void *out_point_host,
                         void *hip_stream) {
    return msm_device_entry(curve, d_fr_elements, d_points, n, out_point_host, hip_stream, 1);
}

static int msm_host_entry(lw_curve_t curve, const uint64_t *scalars, size_t n_scalars, const void *points, size_t n_points,
                          void *out_point, int mont) {
    const size_t pb = lw_hip_curve_point_bytes(curve);
    if (pb == 0 || !out_point) { set_error("bad curve or null output"); return LW_ERR_BAD_ARG; }
    if (n_scalars != n_points) {   // MSMError::LengthMismatch (math/src/msm/pippenger.rs:25-27)
        set_error("scalars and points have different lengths: %zu vs %zu", n_scalars, n_points);
        return LW_ERR_LENGTH_MISMATCH;
    }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    auto t0 = std::chrono::steady_clock::now();
    const size_t n = n_points;
    if (n) {
        if (!scalars || !points) { set_error("null buffer"); return LW_ERR_BAD_ARG; }
        if (c.host_io_a.ensure(n * 32) || c.host_io_b.ensure(n * pb)) return LW_ERR_ALLOC;
    }
    hipStream_t io = en.use_lane_stream();
    if (!io) return en.rc;
    // the scalars first: the sort needs nothing else, and msm_device uploads the points while it runs
    if (n) LW_HIP_CHECK(hipMemcpyAsync(c.host_io_a.p, scalars, n * 32, hipMemcpyHostToDevice, io), LW_ERR_LAUNCH);
    rc = msm_device(c, curve, (const uint64_t *)c.host_io_a.p, c.host_io_b.p, n, out_point, io, mont, 0, n ? points : nullptr);
    c.timings.last_msm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.msm_calls++;
    return rc;
}
int lw_hip_msm(lw_curve_t curve, const uint64_t *scalars, size_t n_scalars, const void *points, size_t n_points, void *out_point) {
    return msm_host_entry(curve, scalars, n_scalars, points, n_points, out_point, 0);
}
int lw_hip_msm_fr(lw_curve_t curve, const uint64_t *fr_elements, size_t n_scalars, const void *points, size_t n_points,
                  void *out_point) {
    return msm_host_entry(curve, fr_elements, n_scalars, points, n_points, out_point, 1);
}

// batched operate_with: out[j*m + i] = rows[i] + cols[j]
int lw_hip_ec_add_outer_device(lw_curve_t curve, const void *d_rows, size_t m, const void *d_cols, size_t k, void *d_out, void *hip_stream) {
    if (lw_hip_curve_point_bytes(curve) == 0) { set_error("bad curve"); return LW_ERR_BAD_ARG; }
    if (m == 0 || k == 0) return LW_OK;
    if (!d_rows || !d_cols || !d_out || (m >> 31) || (k >> 31)) { set_error("null buffer or oversized operand"); return LW_ERR_BAD_ARG; }
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    return ec_add_outer_device(en.c, curve, d_rows, (uint32_t)m, d_cols, (uint32_t)k, d_out, en.stream);
}

// ---- device-resident affine SRS (see include/lw_hip.h) ----
}  // extern "C"
struct lw_srs {
    lw_curve_t curve;
    size_t n;
    lw::DeviceBuf pts;   // n affine rows (2 field elements each), (0,0) = identity; folded: W copies, row w * n + i = 2^(c w) P_i
    uint32_t fold_c = 0; // window width the shifted copies were built for (0: a single copy)
};
extern "C" {

static int srs_build(Context &c, lw_curve_t curve, const void *d_points, size_t n, hipStream_t stream, lw_srs_t **out_srs) {
    const size_t pb = lw_hip_curve_point_bytes(curve);
    lw_srs *h = new (std::nothrow) lw_srs{curve, n, {}};
    if (!h) return LW_ERR_ALLOC;
    // Large sets keep W = 13 window-shifted copies (c = 20), so that every MSM over them runs its 13 windows into ONE set
    // of 2^19 buckets (msm_core.cuh build_fold): 13 x the memory of the affine rows — 27 GiB for 2^24 BLS12-381 G1 points,
    // which is what 288 GB of HBM are for — taken only while it stays below a quarter of the free memory.
    // LW_HIP_SRS_FOLD=0 keeps the single copy.
    static const bool fold_env = [] { const char *e = tuning_env("LW_HIP_SRS_FOLD"); return !e || atoi(e) != 0; }();
    const uint32_t fold_c = 20, fold_w = (256 + fold_c) / fold_c;
    const char *fm = tuning_env("LW_HIP_SRS_FOLD_MIN");   // log2 of the smallest folded set (tests; read per call)
    const int fold_min = fm ? atoi(fm) : 19;
    bool fold = fold_env && n >= ((size_t)1 << (fold_min < 0 ? 0 : fold_min > 40 ? 40 : fold_min)) && (((uint64_t)n * fold_w) >> 31) == 0;
    if (fold) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || msm_affine_bytes(curve, n) * fold_w > free_b / 4) fold = false;
    }
    if (n && h->pts.ensure(msm_affine_bytes(curve, n) * (fold ? fold_w : 1))) { delete h; return LW_ERR_ALLOC; }
    int rc = n ? msm_normalize_device(c, curve, d_points, n, h->pts.p, stream) : LW_OK;
    if (rc == LW_OK && fold) {
        rc = msm_fold_build(c, curve, h->pts.p, n, fold_c, stream);
        if (rc == LW_OK) h->fold_c = fold_c;
    }
    if (rc == LW_OK && n && hipStreamSynchronize(stream) != hipSuccess) { set_error("SRS normalisation failed"); rc = LW_ERR_LAUNCH; }
    if (rc) { h->pts.release(); delete h; return rc; }
    *out_srs = h;
    return LW_OK;
}
int lw_hip_srs_create_device(lw_curve_t curve, const void *d_points, size_t n_points, void *hip_stream, lw_srs_t **out_srs) {
    if (!out_srs || lw_hip_curve_point_bytes(curve) == 0 || (n_points && !d_points)) { set_error("bad curve or null argument"); return LW_ERR_BAD_ARG; }
    Entry en(hip_stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    return srs_build(c, curve, d_points, n_points, (hipStream_t)hip_stream, out_srs);
}
int lw_hip_srs_create(lw_curve_t curve, const void *points, size_t n_points, lw_srs_t **out_srs) {
    const size_t pb = lw_hip_curve_point_bytes(curve);
    if (!out_srs || pb == 0 || (n_points && !points)) { set_error("bad curve or null argument"); return LW_ERR_BAD_ARG; }
    Entry en(nullptr);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    if (n_points) {
        if (c.host_io_b.ensure(n_points * pb)) return LW_ERR_ALLOC;
        LW_HIP_CHECK(hipMemcpy(c.host_io_b.p, points, n_points * pb, hipMemcpyHostToDevice), LW_ERR_LAUNCH);
    }
    return srs_build(c, curve, c.host_io_b.p, n_points, 0, out_srs);
}
int lw_hip_srs_destroy(lw_srs_t *srs) {
    if (!srs) return LW_OK;
    Entry en(nullptr);   // binds the context's device for the hipFree
    srs->pts.release();
    delete srs;
    return LW_OK;
}
static int msm_srs_entry(const lw_srs_t *srs, const uint64_t *scalars, size_t n, void *out_point, hipStream_t stream, int host_scalars,
                         int mont) {
    if (!srs || !out_point) { set_error("null SRS or output"); return LW_ERR_BAD_ARG; }
    if (n > srs->n) {   // MSMError::LengthMismatch (math/src/msm/pippenger.rs:25-27): more scalars than points
        set_error("scalars and points have different lengths: %zu vs %zu", n, srs->n);
        return LW_ERR_LENGTH_MISMATCH;
    }
    if (n && !scalars) { set_error("null buffer"); return LW_ERR_BAD_ARG; }
    Entry en((void *)stream);
    if (en.rc) return en.rc;
    Context &c = en.c;
    int rc = LW_OK;
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t *d_scalars = scalars;
    if (host_scalars) {   // host-buffer form: on the lane's own stream
        stream = en.use_lane_stream();
        if (!stream) return en.rc;
    }
    if (host_scalars && n) {
        if (c.host_io_a.ensure(n * 32)) return LW_ERR_ALLOC;
        LW_HIP_CHECK(hipMemcpyAsync(c.host_io_a.p, scalars, n * 32, hipMemcpyHostToDevice, stream), LW_ERR_LAUNCH);
        d_scalars = (const uint64_t *)c.host_io_a.p;
    }
    // the shifted copies serve calls that use a good part of the set (KZG commits of shorter polynomials take a prefix:
    // below a quarter of it the 2^19 shared buckets would be mostly empty and the plain schedule on copy 0 is faster)
    if (srs->fold_c && n >= srs->n / 4) {
        c.msm_fold_c = srs->fold_c;
        c.msm_fold_stride = srs->n;
    }
    rc = msm_device(c, srs->curve, d_scalars, srs->pts.p, n, out_point, stream, mont, 1);
    c.msm_fold_c = 0;
    c.msm_fold_stride = 0;
    c.timings.last_msm_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c.timings.msm_calls++;
    return rc;
}
int lw_hip_msm_srs(const lw_srs_t *srs, const uint64_t *scalars, size_t n_scalars, void *out_point) {
    return msm_srs_entry(srs, scalars, n_scalars, out_point, 0, 1, 0);
}
int lw_hip_msm_srs_fr(const lw_srs_t *srs, const uint64_t *fr_elements, size_t n_scalars, void *out_point) {
    return msm_srs_entry(srs, fr_elements, n_scalars, out_point, 0, 1, 1);
}
int lw_hip_msm_srs_device(const lw_srs_t *srs, const uint64_t *d_scalars, size_t n_scalars, void *out_point_host, void *hip_stream) {
    return msm_srs_entry(srs, d_scalars, n_scalars, out_point_host, (hipStream_t)hip_stream, 0, 0);
}
int lw_hip_msm_srs_fr_device(const lw_srs_t *srs, const uint64_t *d_fr_elements, size_t n_scalars, void *out_point_host, void *hip_stream) {
    return msm_srs_entry(srs, d_fr_elements, n_scalars, out_point_host, (hipStream_t)hip_stream, 0, 1);
}

}  // extern "C"
