// What does one "cycle" of the trainers' cycle counts last?  ("the 2.4 GHz used to convert ms -> cycles is assumed")
// A single wave64 runs a long dependent v_fma_f32 chain; s_memtime (shader-clock counter), s_memrealtime (constant 100 MHz)
// and the host's HIP events bracket it.  Printed: s_memtime ticks per microsecond = the shader clock the wave actually
// ran at, with the rest of the chip idle and with a second stream keeping every CU busy.
// build: hipcc --offload-arch=gfx950 -O3 -o clock clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void chain(float *out, unsigned long long *t, int iters)
{
    float a = out[threadIdx.x], c = 1.0000001f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(c));) }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
__global__ void burn(float *out, int iters)
{
    float a = out[threadIdx.x & 63], b = a + 1, c = 1.0000001f;
    for (int i = 0; i < iters; i++) { REP16(asm volatile("v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1" : "+v"(a), "+v"(b) : "v"(c));) }
    if (a + b == 12345.f) out[0] = a;
}
int main()
{
    float *out; unsigned long long *t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMemset(out, 0, 4096);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1 << 20;                       // 16.8 M dependent instructions
    for (int busy = 0; busy < 2; busy++) {
        for (int rep = 0; rep < 3; rep++) {
            if (busy) hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), 0, s2, out + 512, 1 << 19);
            hipEventRecord(e0, s1);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, out, t, iters);
            hipEventRecord(e1, s1);
            hipEventSynchronize(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("%s rep %d: kernel %.3f ms (events) | s_memtime %llu ticks = %.1f MHz | s_memrealtime %llu ticks = %.2f MHz | %.3f s_memtime ticks, %.2f ns per dependent v_fma_f32\n",
                   busy ? "chip busy" : "chip idle", rep, ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), (double)h[0] / (16.0 * iters), ms * 1e6 / (16.0 * iters));
        }
    }
    return 0;
}
