// Does work on the other 3 SIMDs of a CU slow down a lone dependent-issue wave?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int ITERS = 2048;
template <int MODE> __global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, const float *in, float *buf)
{
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a = in[lane], c = in[64], b = a + 1, d = a + 2;
    lds[threadIdx.x] = a;
    __syncthreads();
    if (wave == 0) {
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITERS; i++) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(c));) }
        unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) cyc[0] = t1 - t0;
    } else {
        for (int i = 0; i < ITERS * 2; i++) {
            if constexpr (MODE == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(b) : "v"(c));) }
            if constexpr (MODE == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double *)&b) : "v"(*(double *)&d));) }
            if constexpr (MODE == 3) { for (int u = 0; u < 16; u++) b += lds[(lane + u * 64 + i) & 4095]; }
            if constexpr (MODE == 4) { for (int u = 0; u < 16; u++) b += buf[(size_t)((i * 16 + u) * 64 + lane) & 0xFFFFF]; }
            if constexpr (MODE == 5) { REP16(asm volatile("s_nop 0");) }
            if constexpr (MODE == 6) { REP16(asm volatile("s_mul_i32 s20, s20, 3" ::: "s20");) }
        }
    }
    out[threadIdx.x] = a + b + d;
}
template <int MODE> void run(const char *name, float *out, unsigned long long *cyc, float *in, float *buf)
{
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, cyc, in, buf); hipDeviceSynchronize(); }
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-50s chain wave: %6.2f cycles per dependent v_fma\n", name, (double)c / ITERS / 16);
}
int main()
{
    float *out, *in, *buf; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&in, 4096); hipMalloc(&cyc, 64); hipMalloc(&buf, 4 << 20);
    hipMemset(in, 0, 4096); hipMemset(buf, 0, 4 << 20);
    run<0>("other 3 waves idle (exit immediately)", out, cyc, in, buf);
    run<1>("other 3 waves: dependent v_fma", out, cyc, in, buf);
    run<2>("other 3 waves: v_pk_fma", out, cyc, in, buf);
    run<3>("other 3 waves: LDS reads", out, cyc, in, buf);
    run<4>("other 3 waves: global loads", out, cyc, in, buf);
    run<5>("other 3 waves: s_nop", out, cyc, in, buf);
    run<6>("other 3 waves: SALU", out, cyc, in, buf);
    return 0;
}
