// first-touch cost of big host buffers on the GPU box's host: does it scale with threads, do huge pages / MADV_POPULATE_WRITE help?  (the BAM decoder inflates 12 GB into fresh memory)
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t GB = (size_t)1 << 30; const size_t bytes = (argc > 1 ? atoll(argv[1]) : 8) * GB;
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char buf[256] = {0}; if(f) { fgets(buf, 255, f); fclose(f); } printf("THP enabled: %s", buf);
    f = fopen("/sys/kernel/mm/transparent_hugepage/defrag", "r"); if(f) { fgets(buf, 255, f); fclose(f); printf("THP defrag: %s", buf); }
    for(int mode = 0; mode < 4; mode++)             // 0: plain malloc-like, 1: MADV_HUGEPAGE, 2: MADV_HUGEPAGE + POPULATE_WRITE in parallel, 3: plain + POPULATE_WRITE in parallel
        for(int T : {8, 32, 128}) {
            void* p = nullptr; if(posix_memalign(&p, 2 << 20, bytes)) return 1;
            if(mode == 1 || mode == 2) madvise(p, bytes, MADV_HUGEPAGE);
            double t0 = now();
            std::vector<std::thread> th; const size_t per = bytes / T;
            int popErr = 0;
            for(int t = 0; t < T; t++) th.emplace_back([&, t]() {
                char* a = (char*)p + (size_t)t * per;
                if(mode >= 2) { if(madvise(a, per, MADV_POPULATE_WRITE) != 0) popErr = 1; }
                else for(size_t o = 0; o < per; o += 4096) a[o] = 1;
            });
            for(auto& x : th) x.join();
            double t1 = now();
            // second pass: a memset-like streaming write (what the inflate does after the pages exist)
            th.clear();
            for(int t = 0; t < T; t++) th.emplace_back([&, t]() { memset((char*)p + (size_t)t * per, 7, per); });
            for(auto& x : th) x.join();
            double t2 = now();
            printf("mode %d threads %3d: first touch %.3f s (%.1f GB/s)%s, then memset %.3f s (%.1f GB/s)\n", mode, T, t1 - t0, bytes / GB / (t1 - t0), popErr ? " [populate failed]" : "", t2 - t1, bytes / GB / (t2 - t1));
            free(p);
        }
    return 0;
}
