// apply_filter_to_signal on gfx950: strided butterfly FIR + decimate-by-os.
//
// Reference behaviour: qampy/core/equalisation/pythran_equalisation.py:33-76
//     out[j, i] = sum_k sum_t E[k, i*os + t] * wx[modes[j], k, t],   i < N = (L - ntaps + 1)//os
// (k outer, tap inner, no conjugate; the reference parallelises with `omp parallel for collapse(2)`).
//
// Bound: HBM (48 B per symbol period for complex64, 2 modes, os = 2: read 32, write 16) with ~27 flop/B on top, i.e.
// close to the fp32 VALU ridge; MFMA does not apply (the GEMM N dimension is nsel = 2).  Mapping:
//   * one workgroup = one tile of TILE output symbols; the (TILE-1)*os + ntaps input samples of every input mode are
//     staged ONCE into LDS with coalesced loads, so each HBM byte is fetched once per tile (halo = ntaps - os samples);
//   * a thread computes outputs i = tid + r*256 for up to two output modes at a time (register accumulators), reading
//     its window from LDS; the taps are wave-uniform and come through the scalar cache (s_load) as SGPR FMA operands;
//   * grid = ceil(N / TILE) tiles x ceil(nsel / 2): >> 256 workgroups at capture sizes, tile index is the fast grid
//     dimension so neighbouring tiles (which share a halo) land on consecutive dispatches.
#include "common.h"

namespace qh {

constexpr int AP_THREADS = 256;
// P output symbols per thread: 4 by default (1024 per workgroup), 1 when many modes x a large oversampling would not fit the LDS
constexpr int AP_PER_THREAD = 4;
__host__ __device__ constexpr int ap_tile(int P) { return AP_THREADS * P; }
// LDS row pitch in samples: the tile's window, one spare sample, rounded up to an even count
__host__ __device__ constexpr int ap_pitch(int os, int ntaps, int P) { return ((ap_tile(P) - 1) * os + ntaps + 2) & ~1; }
constexpr size_t AP_LDS_MAX = 160 * 1024;

template <typename T> struct ApplyArgs {
    const T *E;
    const T *wx;
    T *out;
    int64_t L, N;
    int nmodes, ntaps, os, nsel;
    int64_t modes[16];
};

// ---- complex
template <typename R, int NJ, int AP_P>
__global__ void __launch_bounds__(AP_THREADS) apply_cplx_kernel(ApplyArgs<Cx<R>> a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Cx<R> *tile = reinterpret_cast<Cx<R> *>(smem);
    const int64_t i0 = (int64_t)blockIdx.x * ap_tile(AP_P);
    const int nout = (int)((a.N - i0) < ap_tile(AP_P) ? (a.N - i0) : ap_tile(AP_P));
    const int span = (nout - 1) * a.os + a.ntaps;                 // samples per input mode needed by this tile
    const int stride = ap_pitch(a.os, a.ntaps, AP_P);                   // LDS row pitch: even, one spare sample for the paired reads
    for (int k = 0; k < a.nmodes; k++) {
        const Cx<R> *src = a.E + (size_t)k * a.L + i0 * a.os;
        for (int s = threadIdx.x; s < span; s += AP_THREADS) tile[k * stride + s] = ldg(src + s);
        if (threadIdx.x == 0) tile[k * stride + span] = Cx<R>{0, 0};            // the spare sample (odd tap counts read it with a zero tap)
    }
    __syncthreads();
    const int j0 = blockIdx.y * NJ;
    const Cx<R> *w0 = a.wx + (size_t)a.modes[j0] * a.nmodes * a.ntaps;
    const Cx<R> *w1 = (NJ > 1 && j0 + 1 < a.nsel) ? a.wx + (size_t)a.modes[j0 + 1] * a.nmodes * a.ntaps : w0;
    R acc[AP_P][NJ][2];
#pragma unroll
    for (int r = 0; r < AP_P; r++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[r][j][0] = acc[r][j][1] = 0;
    if (sizeof(R) == 4 && (a.os & 1) == 0) {
        // even oversampling, complex64: a lane's window starts on a 16-byte boundary, so two consecutive samples (= two taps)
        // come with ONE ds_read_b128, conflict-free at the lanes' 16-byte stride; the complex multiply-adds are written on
        // (re, im) 2-vectors so that they become v_pk_fma_f32 (two per complex MAC instead of four v_fma_f32)
        typedef R v2 __attribute__((ext_vector_type(2)));
        struct alignas(16) Pair { Cx<R> a, b; };
        v2 ac[AP_P][NJ];
#pragma unroll
        for (int r = 0; r < AP_P; r++)
#pragma unroll
            for (int j = 0; j < NJ; j++) ac[r][j] = v2{0, 0};
        auto cmac = [](v2 &acc, const Cx<R> &x, const Cx<R> &c) {
            acc = __builtin_elementwise_fma(v2{x.re, x.re}, v2{c.re, c.im}, acc);
            acc = __builtin_elementwise_fma(v2{x.im, x.im}, v2{-c.im, c.re}, acc);
        };
        // taps, paired and zero-padded, behind the sample rows in LDS: read back as wave-uniform (broadcast) 16-byte loads - no
        // scalar-cache round trip inside the loop, no odd-tap special case
        const int npair = (a.ntaps + 1) / 2;
        Pair *wt = reinterpret_cast<Pair *>(tile + (size_t)a.nmodes * stride);                // [NJ][nmodes][npair]
        for (int e = threadIdx.x; e < NJ * a.nmodes * npair; e += AP_THREADS) {
            const int j = e / (a.nmodes * npair), q = e - j * a.nmodes * npair, k = q / npair, pp = q - k * npair;
            const Cx<R> *w = j == 0 ? w0 : w1;
            Pair v;
            v.a = w[k * a.ntaps + 2 * pp];
            v.b = 2 * pp + 1 < a.ntaps ? w[k * a.ntaps + 2 * pp + 1] : Cx<R>{0, 0};
            wt[e] = v;
        }
        __syncthreads();
        for (int k = 0; k < a.nmodes; k++) {
            const Cx<R> *row = tile + k * stride + threadIdx.x * a.os;
            const Pair *wk0 = wt + k * npair, *wk1 = wt + (a.nmodes + k) * npair;
#pragma unroll 2
            for (int pp = 0; pp < npair; pp++) {
                const Pair c0 = wk0[pp];
                Pair c1 = c0;
                if constexpr (NJ > 1) c1 = wk1[pp];
#pragma unroll
                for (int r = 0; r < AP_P; r++) {
                    const Pair x = *reinterpret_cast<const Pair *>(row + r * AP_THREADS * a.os + 2 * pp);
                    cmac(ac[r][0], x.a, c0.a);
                    cmac(ac[r][0], x.b, c0.b);
                    if constexpr (NJ > 1) {
                        cmac(ac[r][1], x.a, c1.a);
                        cmac(ac[r][1], x.b, c1.b);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < AP_P; r++)
#pragma unroll
            for (int j = 0; j < NJ; j++) { acc[r][j][0] = ac[r][j].x; acc[r][j][1] = ac[r][j].y; }
    } else
    for (int k = 0; k < a.nmodes; k++) {
        const Cx<R> *row = tile + k * stride;
        for (int t = 0; t < a.ntaps; t++) {
            const Cx<R> c0 = w0[k * a.ntaps + t];                 // wave-uniform -> scalar loads
            const Cx<R> c1 = w1[k * a.ntaps + t];
#pragma unroll
            for (int r = 0; r < AP_P; r++) {
                const int il = threadIdx.x + r * AP_THREADS;
                // rows past `nout` read stale-but-in-bounds LDS (stride covers the full tile); they are never stored
                const Cx<R> x = row[il * a.os + t];
                acc[r][0][0] = fma_(x.re, c0.re, fma_(-x.im, c0.im, acc[r][0][0]));
                acc[r][0][1] = fma_(x.re, c0.im, fma_(x.im, c0.re, acc[r][0][1]));
                if constexpr (NJ > 1) {
                    acc[r][1][0] = fma_(x.re, c1.re, fma_(-x.im, c1.im, acc[r][1][0]));
                    acc[r][1][1] = fma_(x.re, c1.im, fma_(x.im, c1.re, acc[r][1][1]));
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < AP_P; r++) {
        const int il = threadIdx.x + r * AP_THREADS;
        if (il < nout) {
            stg(a.out + (size_t)j0 * a.N + i0 + il, Cx<R>{acc[r][0][0], acc[r][0][1]});
            if constexpr (NJ > 1)
                if (j0 + 1 < a.nsel) stg(a.out + (size_t)(j0 + 1) * a.N + i0 + il, Cx<R>{acc[r][1][0], acc[r][1][1]});
        }
    }
}

// ---- real (the real-valued equaliser path, equalisation.py:178-184)
template <typename R, int AP_P>
__global__ void __launch_bounds__(AP_THREADS) apply_real_kernel(ApplyArgs<R> a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R *tile = reinterpret_cast<R *>(smem);
    const int64_t i0 = (int64_t)blockIdx.x * ap_tile(AP_P);
    const int nout = (int)((a.N - i0) < ap_tile(AP_P) ? (a.N - i0) : ap_tile(AP_P));
    const int span = (nout - 1) * a.os + a.ntaps;
    const int stride = (ap_tile(AP_P) - 1) * a.os + a.ntaps;
    for (int k = 0; k < a.nmodes; k++) {
        const R *src = a.E + (size_t)k * a.L + i0 * a.os;
        for (int s = threadIdx.x; s < span; s += AP_THREADS) tile[k * stride + s] = src[s];
    }
    __syncthreads();
    const int j = blockIdx.y;
    const R *w = a.wx + (size_t)a.modes[j] * a.nmodes * a.ntaps;
    R acc[AP_P];
#pragma unroll
    for (int r = 0; r < AP_P; r++) acc[r] = 0;
    for (int k = 0; k < a.nmodes; k++) {
        const R *row = tile + k * stride;
        for (int t = 0; t < a.ntaps; t++) {
            const R c = w[k * a.ntaps + t];
#pragma unroll
            for (int r = 0; r < AP_P; r++) acc[r] = fma_(row[(threadIdx.x + r * AP_THREADS) * a.os + t], c, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < AP_P; r++) {
        const int il = threadIdx.x + r * AP_THREADS;
        if (il < nout) a.out[(size_t)j * a.N + i0 + il] = acc[r];
    }
}

template <typename T> static int apply_check(int nmodes, int64_t L, int os, int ntaps, const int64_t *modes, int nsel, int nrows_w)
{
    QH_REQUIRE(os >= 1, "apply_filter_to_signal: oversampling factor must be larger than 0");
    QH_REQUIRE(nmodes >= 1 && ntaps >= 1 && L >= 0, "apply_filter_to_signal: bad sizes");
    QH_REQUIRE(nsel >= 1 && nsel <= 16, "apply_filter_to_signal: between 1 and 16 modes can be selected");
    for (int j = 0; j < nsel; j++)
        QH_REQUIRE(modes[j] >= 0 && modes[j] < nrows_w, "apply_filter_to_signal: largest mode number is larger than shape of taps");
    const size_t lds = ((size_t)nmodes * ((size_t)(ap_tile(1) - 1) * os + ntaps + 2) + (size_t)2 * nmodes * (ntaps + 1)) * sizeof(T);
    QH_REQUIRE(lds <= AP_LDS_MAX, "apply_filter_to_signal: nmodes*(255*os+ntaps) samples exceed the 160 KiB LDS tile");
    return QH_OK;
}

template <typename R> int apply_cplx_dev(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps,
                                         const int64_t *modes, int nsel, void *out)
{
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = apply_check<Cx<R>>(nmodes, L, os, ntaps, modes, nsel, nmodes))) return rc;
    const int64_t N = (L - ntaps + 1) / os;
    if (N <= 0) return QH_OK;
    ApplyArgs<Cx<R>> a;
    a.E = (const Cx<R> *)E; a.wx = (const Cx<R> *)wx; a.out = (Cx<R> *)out; a.L = L; a.N = N;
    a.nmodes = nmodes; a.ntaps = ntaps; a.os = os; a.nsel = nsel;
    for (int j = 0; j < 16; j++) a.modes[j] = j < nsel ? modes[j] : 0;
    auto lds_for = [&](int P) { return ((size_t)nmodes * (size_t)ap_pitch(os, ntaps, P) + (size_t)2 * nmodes * (ntaps + 1)) * sizeof(Cx<R>); };   // sample rows + paired taps
    const int P = lds_for(AP_PER_THREAD) <= AP_LDS_MAX ? AP_PER_THREAD : 1;
    const size_t lds = lds_for(P);
    const unsigned ntile = (unsigned)((N + ap_tile(P) - 1) / ap_tile(P));
#define QH_AP_LAUNCH(NJ, PP, GY)                                                                                                     \
    do {                                                                                                                             \
        if (lds > 64 * 1024) QH_HIP(hipFuncSetAttribute((const void *)apply_cplx_kernel<R, NJ, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((apply_cplx_kernel<R, NJ, PP>), dim3(ntile, GY), dim3(AP_THREADS), lds, g_stream, a);                      \
    } while (0)
    if (nsel == 1) { if (P == 1) QH_AP_LAUNCH(1, 1, 1); else QH_AP_LAUNCH(1, AP_PER_THREAD, 1); }
    else { if (P == 1) QH_AP_LAUNCH(2, 1, (nsel + 1) / 2); else QH_AP_LAUNCH(2, AP_PER_THREAD, (nsel + 1) / 2); }
#undef QH_AP_LAUNCH
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R> int apply_real_dev(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps,
                                         const int64_t *modes, int nsel, void *out)
{
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = apply_check<R>(nmodes, L, os, ntaps, modes, nsel, nmodes))) return rc;
    const int64_t N = (L - ntaps + 1) / os;
    if (N <= 0) return QH_OK;
    ApplyArgs<R> a;
    a.E = (const R *)E; a.wx = (const R *)wx; a.out = (R *)out; a.L = L; a.N = N;
    a.nmodes = nmodes; a.ntaps = ntaps; a.os = os; a.nsel = nsel;
    for (int j = 0; j < 16; j++) a.modes[j] = j < nsel ? modes[j] : 0;
    auto lds_for = [&](int P) { return (size_t)nmodes * ((size_t)(ap_tile(P) - 1) * os + ntaps) * sizeof(R); };
    const int P = lds_for(AP_PER_THREAD) <= AP_LDS_MAX ? AP_PER_THREAD : 1;
    const size_t lds = lds_for(P);
    const unsigned ntile = (unsigned)((N + ap_tile(P) - 1) / ap_tile(P));
    if (P == 1) {
        if (lds > 64 * 1024) QH_HIP(hipFuncSetAttribute((const void *)apply_real_kernel<R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((apply_real_kernel<R, 1>), dim3(ntile, nsel), dim3(AP_THREADS), lds, g_stream, a);
    } else {
        if (lds > 64 * 1024) QH_HIP(hipFuncSetAttribute((const void *)apply_real_kernel<R, AP_PER_THREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((apply_real_kernel<R, AP_PER_THREAD>), dim3(ntile, nsel), dim3(AP_THREADS), lds, g_stream, a);
    }
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename T, typename F>
static int apply_host(F devfn, const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                      int nsel, void *out)
{
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = apply_check<T>(nmodes, L, os, ntaps, modes, nsel, nmodes))) return rc;
    const int64_t N = (L - ntaps + 1) / os;
    if (N <= 0) return QH_OK;
    DevBuf dE, dw, dout;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * sizeof(T)))) return rc;
    if ((rc = dw.from_host(wx, (size_t)nmodes * nmodes * ntaps * sizeof(T)))) return rc;
    if ((rc = dout.alloc((size_t)nsel * N * sizeof(T)))) return rc;
    if ((rc = devfn(dE.p, nmodes, L, os, dw.p, ntaps, modes, nsel, dout.p))) return rc;
    if ((rc = dout.to_host(out, dout.n))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

}  // namespace qh

extern "C" {
int qh_apply_filter_c64_dev(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes, int nsel, void *out)
{ return qh::apply_cplx_dev<float>(E, nmodes, L, os, wx, ntaps, modes, nsel, out); }
int qh_apply_filter_c128_dev(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes, int nsel, void *out)
{ return qh::apply_cplx_dev<double>(E, nmodes, L, os, wx, ntaps, modes, nsel, out); }
int qh_apply_filter_c64(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes, int nsel, void *out)
{ return qh::apply_host<qh::Cx<float>>(qh::apply_cplx_dev<float>, E, nmodes, L, os, wx, ntaps, modes, nsel, out); }
int qh_apply_filter_c128(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes, int nsel, void *out)
{ return qh::apply_host<qh::Cx<double>>(qh::apply_cplx_dev<double>, E, nmodes, L, os, wx, ntaps, modes, nsel, out); }
int qh_apply_filter_f32(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes, int nsel, void *out)
{ return qh::apply_host<float>(qh::apply_real_dev<float>, E, nmodes, L, os, wx, ntaps, modes, nsel, out); }
int qh_apply_filter_f64(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes, int nsel, void *out)
{ return qh::apply_host<double>(qh::apply_real_dev<double>, E, nmodes, L, os, wx, ntaps, modes, nsel, out); }
}
