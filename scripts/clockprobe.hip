// Diagnostic: shader clock seen by short kernels launched back to back (s_memtime ticks per 100 MHz realtime tick).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i)
        x = x * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (x == 12345.f)
        out[2] = 1;
}
int main()
{
    unsigned long long *d, h[3];
    hipMalloc(&d, 64);
    for (int iters : {2000, 20000, 200000}) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int k = 0; k < 200; ++k)
                probe<<<1000, 256>>>(d, iters);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("iters=%d shader ticks=%llu realtime ticks=%llu -> %.0f MHz\n", iters, h[0], h[1],
                   100.0 * (double)h[0] / (double)h[1]);
        }
    }
    return 0;
}
