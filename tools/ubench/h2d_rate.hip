// Host -> device copy rate of this box for the shapes the offline run (BASELINE configs[4]) uploads: page-locked memory of the
// different hipHostMalloc kinds, copy sizes from one 720p BGR frame to a chunk of 128, one and two streams, and with a D2H copy
// running beside it.  Build: hipcc --offload-arch=gfx950 -O2 -o h2d_rate h2d_rate.hip ; the numbers are the PCIe ceiling that
// DESIGN.md prices the offline run against.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t frame = (size_t)1280 * 720 * 3, total = frame * 256;
    void *d0, *d1;
    CHK(hipMalloc(&d0, total)); CHK(hipMalloc(&d1, total));
    hipStream_t s0, s1;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    struct Kind { const char *name; unsigned flags; } kinds[] = { { "default", hipHostMallocDefault }, { "noncoherent", hipHostMallocNonCoherent },
        { "writecombined", hipHostMallocWriteCombined }, { "numa_user", hipHostMallocNumaUser }, { "portable|mapped", hipHostMallocPortable | hipHostMallocMapped } };
    for (const Kind &k : kinds) {
        void *h0 = nullptr, *h1 = nullptr;
        if (hipHostMalloc(&h0, total, k.flags) != hipSuccess || hipHostMalloc(&h1, total, k.flags) != hipSuccess) { printf("%-16s allocation failed\n", k.name); (void)hipGetLastError(); continue; }
        memset(h0, 1, total); memset(h1, 2, total);
        for (size_t n : { (size_t)1, (size_t)8, (size_t)32, (size_t)128, (size_t)256 }) {
            const size_t bytes = frame * n;
            const int reps = n >= 128 ? 6 : 40;
            CHK(hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0)); CHK(hipStreamSynchronize(s0));
            double t = now();
            for (int r = 0; r < reps; ++r) CHK(hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0));
            CHK(hipStreamSynchronize(s0));
            const double one = bytes * reps / (now() - t) / 1e9;
            t = now();
            for (int r = 0; r < reps; ++r) { CHK(hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0)); CHK(hipMemcpyAsync(d1, h1, bytes, hipMemcpyHostToDevice, s1)); }
            CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
            const double two = 2.0 * bytes * reps / (now() - t) / 1e9;
            t = now();
            for (int r = 0; r < reps; ++r) { CHK(hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s0)); CHK(hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s1)); }
            CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1));
            const double bidir = bytes * reps / (now() - t) / 1e9;
            printf("%-16s %4zu frames (%8.1f MB): H2D one stream %6.1f GB/s, two streams %6.1f GB/s (sum), H2D beside a D2H %6.1f GB/s each way\n",
                   k.name, n, bytes / 1e6, one, two, bidir);
        }
        (void)hipHostFree(h0); (void)hipHostFree(h1);
    }
    // pageable memory for comparison
    {
        std::vector<char> p(frame * 128, 3);
        CHK(hipMemcpy(d0, p.data(), p.size(), hipMemcpyHostToDevice));
        double t = now();
        for (int r = 0; r < 4; ++r) CHK(hipMemcpy(d0, p.data(), p.size(), hipMemcpyHostToDevice));
        printf("pageable          128 frames: %.1f GB/s\n", 4.0 * p.size() / (now() - t) / 1e9);
    }
    return 0;
}
