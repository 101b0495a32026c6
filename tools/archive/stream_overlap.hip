// developer tool: do kernels of different streams of one process run side by side on this box?  two (four, eight) streams,
// each with one kernel of 64 workgroups spinning ~5 ms: the wall time says whether they overlapped.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned *sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < ticks) x++;
    if (x == 0xffffffffu) *sink = x;
}
int main()
{
    unsigned *sink; hipMalloc(&sink, 4);
    for (int ns : {1, 2, 4, 8}) {
        for (int pri = 0; pri < 2; pri++) {
            hipStream_t st[8];
            int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
            for (int i = 0; i < ns; i++) {
                if (pri && i == ns - 1) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
                else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
            }
            for (int i = 0; i < ns; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[i], 1000ull, sink);   // warm
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < ns; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[i], 500000ull, sink);  // 5 ms at 100 MHz
            hipDeviceSynchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%d streams%s: %.2f ms (one kernel: 5 ms)\n", ns, pri ? " (last one high priority)" : "", ms);
            for (int i = 0; i < ns; i++) hipStreamDestroy(st[i]);
        }
    }
    return 0;
}
