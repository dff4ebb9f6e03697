// Single-wave latency micro-benchmarks for the sequential equaliser chain (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o lat lat.hip ; run on the GPU box.  Prints cycles (s_memtime) per operation.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int ITERS = 4096;

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <int T> __global__ void k(float *out, unsigned long long *cyc, const float *in)
{
    float a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[128];
    float a2 = a + 1, b2 = b + 1, a3 = a + 2, b3 = b + 3;
    int s0 = 0, s1 = 0;
    unsigned long long t0 = now();
    for (int i = 0; i < ITERS; i++) {
        if constexpr (T == 0) {   // dependent v_fma chain
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(c));)
        } else if constexpr (T == 1) {   // 2 independent fma chains interleaved
            REP16(asm volatile("v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1" : "+v"(a), "+v"(b) : "v"(c));)
        } else if constexpr (T == 2) {   // 4 independent chains
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %0\n\tv_fma_f32 %1, %1, %4, %1\n\tv_fma_f32 %2, %2, %4, %2\n\tv_fma_f32 %3, %3, %4, %3" : "+v"(a), "+v"(b), "+v"(a2), "+v"(b2) : "v"(c));)
        } else if constexpr (T == 3) {   // dependent DPP add chain (1 value): nop 1 + dpp
            REP16(asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));)
        } else if constexpr (T == 4) {   // two interleaved DPP chains (as in the reduction): dpp a, dpp b, nop 0
            REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(a), "+v"(b));)
        } else if constexpr (T == 5) {   // readlane -> valu use -> (dependent)
            REP16(asm volatile("v_readlane_b32 %1, %0, 63\n\ts_nop 3\n\tv_add_f32 %0, %1, %0" : "+v"(a), "=s"(s0));)
        } else if constexpr (T == 6) {   // pk_fma dependent chain
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double *)&a2) : "v"(*(double *)&b2));)
        } else if constexpr (T == 7) {   // row_bcast dependent chain
            REP16(asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a));)
        } else if constexpr (T == 8) {   // the full 2-component wave reduction block (6 levels + readlanes) + dependent use
            asm volatile(
                "s_nop 1\n\t"
                "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\t"
                "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0\n\t"
                "v_readlane_b32 %2, %0, 63\n\tv_readlane_b32 %3, %1, 63\n\ts_nop 3\n\t"
                "v_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %3, %1"
                : "+v"(a), "+v"(b), "=s"(s0), "=s"(s1));
        } else if constexpr (T == 9) {   // s_nop 0 x16 (issue cost of a nop)
            REP16(asm volatile("s_nop 0");)
        } else if constexpr (T == 10) {  // independent v_mov (pure issue rate)
            REP16(asm volatile("v_mov_b32 %0, %1" : "=v"(a3) : "v"(c));)
        } else if constexpr (T == 11) {  // ds_swizzle-free cross-lane: v_permlane32_swap + add (dependent)
            REP16(asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "=&v"(b3));)
        } else if constexpr (T == 12) {  // taken scalar branch
            REP16(asm volatile("s_cbranch_scc0 1f\n\ts_nop 0\n\t1:\n\ts_cmp_eq_u32 0, 1" ::: "scc");)
        }
    }
    unsigned long long t1 = now();
    out[threadIdx.x] = a + b + a2 + b2 + a3 + b3 + __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int T> void run(const char *name, int ops_per_rep, float *out, unsigned long long *cyc, float *in)
{
    hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, out, cyc, in);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<T>, dim3(1), dim3(64), 0, 0, out, cyc, in);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.2f ticks per unit (%d units/iter)\n", name, (double)c / ITERS / ops_per_rep, ops_per_rep);
}

int main()
{
    float *out, *in; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&in, 4096); hipMalloc(&cyc, 64);
    hipMemset(in, 0, 4096);
    // calibrate the counter: wall time of a known loop
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, in); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("counter: %llu ticks in %.3f us kernel -> %.1f MHz tick rate (incl. launch)\n", c, ms * 1e3, c / (ms * 1e3));
    run<0>("dependent v_fma_f32", 16, out, cyc, in);
    run<1>("2 interleaved v_fma chains (per pair)", 16, out, cyc, in);
    run<2>("4 interleaved v_fma chains (per quad)", 16, out, cyc, in);
    run<3>("dependent dpp add (nop1 + dpp)", 16, out, cyc, in);
    run<4>("2 interleaved dpp chains + nop0 (per level)", 16, out, cyc, in);
    run<5>("readlane -> nop3 -> valu (per round trip)", 16, out, cyc, in);
    run<6>("dependent v_pk_fma_f32", 16, out, cyc, in);
    run<7>("dependent row_bcast dpp add (nop1 + dpp)", 16, out, cyc, in);
    run<8>("full 2-comp wave reduction + readlane + use", 1, out, cyc, in);
    run<9>("s_nop 0", 16, out, cyc, in);
    run<10>("independent v_mov", 16, out, cyc, in);
    run<11>("permlane32_swap + add (dependent)", 16, out, cyc, in);
    run<12>("taken s_cbranch", 16, out, cyc, in);
    return 0;
}
