// How many wave64 fp32 instructions per cycle does a SIMD issue, as a function of the waves resident on it?
// (the peak of the "valu-issue" roofline bench.py quotes for the segment trainer and the phase search)
// One workgroup of W waves per CU-slot; every wave runs independent (4 accumulators) or dependent v_fma_f32 / v_pk_fma_f32.
// grid = 256 CUs x 1 workgroup, block = 64 * 4 * wps threads -> wps waves on each of the 4 SIMDs of every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o issue issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND> __global__ void __launch_bounds__(1024) k(float *out, unsigned long long *t, int iters)
{
    float a = out[threadIdx.x & 63], b = a + 1, c = a + 2, d = a + 3, m = 1.0000001f;
    float2 p = {a, b}, q = {c, d}, mm = {m, m};
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(m));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n\tv_fma_f32 %1, %1, %4, %1\n\tv_fma_f32 %2, %2, %4, %2\n\tv_fma_f32 %3, %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (KIND == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\tv_pk_fma_f32 %1, %1, %2, %1" : "+v"(p), "+v"(q) : "v"(mm));) }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (a + b + c + d + p.x + q.y == 12345.f) out[0] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
int main()
{
    float *out; unsigned long long *t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMemset(out, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1 << 14;
    const char *names[3] = {"dependent v_fma_f32 (1 chain)", "independent v_fma_f32 (4 chains)", "v_pk_fma_f32 (2 chains)"};
    const int per_iter[3] = {16, 64, 32};
    for (int kind = 0; kind < 3; kind++)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const dim3 grid(256), block(64 * 4 * wps);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, t, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, t, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, t, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
                const double instr = (double)per_iter[kind] * iters;          // per wave
                if (rep == 1)
                    printf("%-34s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.3f instr/cycle/SIMD, whole chip %.1f G wave-instr/s (%.2f ms, %.0f MHz)\n",
                           names[kind], wps, (double)h / instr, instr * wps / (double)h, instr * wps * 1024 / (ms * 1e-3) / 1e9, ms, h / (ms * 1e3));
            }
        }
    return 0;
}
