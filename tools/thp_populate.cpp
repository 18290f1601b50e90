// Cost of first-touching a fresh 512 MiB result buffer with and without transparent huge pages (MADV_HUGEPAGE) and of freeing it,
// by populate threads: the numbers behind the host path of lw_hip_ntt (csrc/api.hip Prefault).  usage: thp_populate HUGE(0|1) THREADS
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <unistd.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t bytes = 512ull << 20;
    int huge = atoi(argv[1]), T = atoi(argv[2]);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        char *p = (char *)malloc(bytes);
        uintptr_t a = (uintptr_t)p;
        const size_t page = 4096, H = 2u << 20;
        uintptr_t lo = (a + page - 1) & ~(uintptr_t)(page - 1), hi = (a + bytes) & ~(uintptr_t)(page - 1);
        if (huge) {
            uintptr_t hlo = (a + H - 1) & ~(uintptr_t)(H - 1), hhi = (a + bytes) & ~(uintptr_t)(H - 1);
            if (hhi > hlo) { int r = madvise((void *)hlo, hhi - hlo, MADV_HUGEPAGE); if (r) perror("madvise hugepage"); }
        }
        double t1 = now();
        std::vector<std::thread> th;
        size_t pages = (hi - lo) / page, per = ((pages + T - 1) / T + 511) & ~(size_t)511;
        for (int t = 0; t < T; t++) {
            size_t p0 = t * per, p1 = std::min(pages, p0 + per);
            if (p0 >= p1) break;
            th.emplace_back([=] { if (madvise((void *)(lo + p0 * page), (p1 - p0) * page, MADV_POPULATE_WRITE)) perror("populate"); });
        }
        for (auto &x : th) x.join();
        double t2 = now();
        memset(p, 1, bytes);
        double t3 = now();
        free(p);
        double t4 = now();
        printf("huge=%d T=%d: alloc+madvise %.2f ms, populate %.2f ms, memset-after %.2f ms, free %.2f ms\n", huge, T, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    FILE *f = fopen("/proc/meminfo", "r"); char line[256]; while (fgets(line, 256, f)) if (strstr(line, "AnonHuge")) printf("%s", line); fclose(f);
    return 0;
}
