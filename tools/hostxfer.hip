// Host <-> device transfer rates on the GPU box: pageable hipMemcpy, pinned hipMemcpyAsync, hipHostRegister cost, and a
// threaded pinned-staging pipeline (what lw_hip_ntt's host path uses).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -pthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static void staged(char *user, char *dev, size_t bytes, bool to_dev, int T, size_t chunk) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([=] {
            CK(hipSetDevice(0));
            char *pin[2];
            hipStream_t s;
            hipEvent_t ev[2];
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) { CK(hipHostMalloc((void **)&pin[i], chunk)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
            size_t nchunks = (bytes + chunk - 1) / chunk;
            int slot = 0;
            bool used[2] = {false, false};
            size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
            for (size_t k = t; k < nchunks; k += T) {
                size_t off = k * chunk, len = std::min(chunk, bytes - off);
                if (used[slot]) {
                    CK(hipEventSynchronize(ev[slot]));
                    if (!to_dev) memcpy(user + pend_off[slot], pin[slot], pend_len[slot]);
                }
                if (to_dev) {
                    memcpy(pin[slot], user + off, len);
                    CK(hipMemcpyAsync(dev + off, pin[slot], len, hipMemcpyHostToDevice, s));
                } else {
                    CK(hipMemcpyAsync(pin[slot], dev + off, len, hipMemcpyDeviceToHost, s));
                    pend_off[slot] = off; pend_len[slot] = len;
                }
                CK(hipEventRecord(ev[slot], s));
                used[slot] = true;
                slot ^= 1;
            }
            for (int i = 0; i < 2; i++) {
                int sl = slot ^ i;   // oldest first
                if (used[sl]) { CK(hipEventSynchronize(ev[sl])); if (!to_dev) memcpy(user + pend_off[sl], pin[sl], pend_len[sl]); }
            }
            for (int i = 0; i < 2; i++) { CK(hipHostFree(pin[i])); CK(hipEventDestroy(ev[i])); }
            CK(hipStreamDestroy(s));
        });
    for (auto &x : th) x.join();
}

int main() {
    const size_t bytes = 512ull << 20;
    char *user = (char *)aligned_alloc(4096, bytes), *user2 = (char *)aligned_alloc(4096, bytes);
    memset(user, 1, bytes); memset(user2, 2, bytes);
    char *dev; CK(hipMalloc((void **)&dev, bytes));
    char *pin; CK(hipHostMalloc((void **)&pin, bytes));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipMemcpy(dev, user, bytes, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(user2, dev, bytes, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable  H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)\n", t1 - t0, bytes / (t1 - t0) / 1e6, t2 - t1, bytes / (t2 - t1) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(pin, dev, bytes, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pinned    H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)\n", t1 - t0, bytes / (t1 - t0) / 1e6, t2 - t1, bytes / (t2 - t1) / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipHostRegister(user, bytes, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpy(dev, user, bytes, hipMemcpyHostToDevice)); double t2 = now();
        CK(hipMemcpy(user, dev, bytes, hipMemcpyDeviceToHost)); double t3 = now();
        CK(hipHostUnregister(user)); double t4 = now();
        printf("register %.1f ms, H2D %.1f, D2H %.1f, unregister %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    // fresh (never touched) destination, as a new Vec / numpy.empty is: plain copy, then populate-first variants
    for (int variant = 0; variant < 4; variant++) {
        char *fresh = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        double t0 = now(), tp = t0;
        if (variant >= 1) {
            int T = variant == 1 ? 1 : variant == 2 ? 8 : 32;
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([=] {
                    int rc = madvise(fresh + (bytes / T) * t, bytes / T, MADV_POPULATE_WRITE);
                    if (rc) perror("madvise");
                });
            for (auto &x : th) x.join();
            tp = now();
        }
        CK(hipMemcpy(fresh, dev, bytes, hipMemcpyDeviceToHost));
        double t1 = now();
        munmap(fresh, bytes);
        double t2 = now();
        printf("fresh dst variant %d: populate %.1f ms, D2H %.1f ms, munmap %.1f ms\n", variant, tp - t0, t1 - tp, t2 - t1);
    }
    { double t0 = now(); memcpy(user2, user, bytes); double t1 = now(); printf("memcpy 1 thread %.1f ms (%.1f GB/s)\n", t1 - t0, bytes / (t1 - t0) / 1e6); }
    for (int T : {2})
        for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20}) {
            staged(user, dev, bytes, true, T, chunk);
            double t0 = now(); staged(user, dev, bytes, true, T, chunk); double t1 = now();
            staged(user2, dev, bytes, false, T, chunk); double t2 = now();
            printf("staged T=%2d chunk=%2zu MiB  H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)  ok=%d\n", T, chunk >> 20, t1 - t0, bytes / (t1 - t0) / 1e6,
                   t2 - t1, bytes / (t2 - t1) / 1e6, memcmp(user, user2, bytes) == 0);
        }
    return 0;
}
