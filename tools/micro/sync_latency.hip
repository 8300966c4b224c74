// Measurement only (tools/README.md; the experiment it belongs to is recorded in DESIGN.md section 8 as NOT adopted): what a host thread waits between the end of a small kernel and the return of (a) hipStreamSynchronize,
// (b) a spin on a flag the kernel's last lane stores into page-locked host memory.  hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_work(uint32_t *out, volatile uint32_t *flag, uint32_t v, int spin)
{
    uint32_t a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1664525u + 1013904223u;
    out[threadIdx.x] = a;
    if (flag && threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *d_out, *h_out; volatile uint32_t *flag;
    hipMalloc(&d_out, 1024); hipHostMalloc((void **)&h_out, 1024, hipHostMallocDefault); hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
    *flag = 0;
    for (int spin : { 0, 20000, 100000 }) {
        for (int mode = 0; mode < 3; ++mode) {
            double acc = 0; const int N = 2000;
            for (int i = 0; i < N + 100; ++i) {
                const double t0 = now_us();
                if (mode == 0) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, d_out, (volatile uint32_t *)nullptr, 0u, spin); hipStreamSynchronize(s); }
                else if (mode == 1) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, h_out, flag, (uint32_t)(i + 1), spin); while (*flag != (uint32_t)(i + 1)) { } }
                else { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, h_out, (volatile uint32_t *)nullptr, 0u, spin); hipStreamSynchronize(s); }
                const double t1 = now_us();
                if (i >= 100) acc += t1 - t0;
            }
            hipStreamSynchronize(s);
            printf("spin %6d  %-44s %.2f us per launch + wait\n", spin, mode == 0 ? "device output, hipStreamSynchronize" : mode == 1 ? "host output + flag, host spins on the flag" : "host output, hipStreamSynchronize", acc / N);
        }
    }
    return 0;
}
