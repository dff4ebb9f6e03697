// On-device symbol-error-rate harness: alignment + decisions + error count without moving the recovered
// signal to the host.  Reference: cal_ser (core/signals.py:295-335) = sync_and_adjust / find_sequence_offset_complex
// (core/ber_functions.py:33-160: correlation over the 4 quadrant rotations, pick the best) + make_decision
// (pythran_equalisation.py:304-334) + comparison of the decided symbols.  Here the correlation search is a bounded-lag
// search on DECIDED INDICES: for every transmitted mode, rotation j^k and lag in [-maxlag, maxlag] the matches inside a
// window are counted; the best triple wins (first maximum in (mode, rotation, lag) order), then the whole row is decided
// with that rotation and compared at that lag.
#include "common.h"
#include <vector>

namespace qh {

template <typename R> __device__ __forceinline__ R hyp(R a, R b);
template <> __device__ __forceinline__ float hyp<float>(float a, float b) { return hypotf(a, b); }
template <> __device__ __forceinline__ double hyp<double>(double a, double b) { return hypot(a, b); }

// first minimum of |x - s_k| (np.abs, not squared: det_symbol_argmin, pythran_equalisation.py:233-236)
template <typename R> __device__ __forceinline__ int nearest_index(Cx<R> x, const Cx<R> *symbols, int M)
{
    Cx<R> s = symbols[0];
    R best = hyp<R>(x.re - s.re, x.im - s.im);
    int ib = 0;
    for (int k = 1; k < M; k++) {
        s = symbols[k];
        const R d = hyp<R>(x.re - s.re, x.im - s.im);
        if (d < best) { best = d; ib = k; }
    }
    return ib;
}

template <typename R> __device__ __forceinline__ Cx<R> rot90(Cx<R> x, int k)       // x * j^k
{
    switch (k & 3) {
    case 1: return Cx<R>{-x.im, x.re};
    case 2: return Cx<R>{-x.re, -x.im};
    case 3: return Cx<R>{x.im, -x.re};
    default: return x;
    }
}

// idx[k][i] = decision index of E[start + i] * j^k, k = rot0 .. rot0 + nrot - 1
template <typename R>
__global__ void __launch_bounds__(256) ser_decide_kernel(const Cx<R> *E, int64_t start, int64_t n, int rot0, int nrot, const Cx<R> *symbols, int M,
                                                         int32_t *idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Cx<R> x = ldg(E + start + i);
    for (int k = 0; k < nrot; k++) idx[(size_t)k * n + i] = nearest_index<R>(rot90(x, rot0 + k), symbols, M);
}

// counts[(mode * 4 + rot) * nlag + l] = matches of rx window (rotation rot) against tx mode at lag l - maxlag
__global__ void __launch_bounds__(256) ser_search_kernel(const int32_t *idx4, int64_t W, const int32_t *idx_tx, int64_t ntx, int64_t start, int maxlag,
                                                         unsigned *counts)
{
    const int nlag = 2 * maxlag + 1;
    const int l = blockIdx.x, rot = blockIdx.y & 3, mode = blockIdx.y >> 2;
    const int64_t lag = (int64_t)l - maxlag;
    const int32_t *rx = idx4 + (size_t)rot * W, *tx = idx_tx + (size_t)mode * ntx;
    int c = 0;
    for (int64_t i = threadIdx.x; i < W; i += 256) {
        const int64_t it = start + i - lag;
        if (it >= 0 && it < ntx) c += rx[i] == tx[it];
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[((size_t)mode * 4 + rot) * nlag + l] = (unsigned)(part[0] + part[1] + part[2] + part[3]);
}

__global__ void __launch_bounds__(256) ser_count_kernel(const int32_t *rx, const int32_t *tx, int64_t i0, int64_t i1, int64_t lag, int64_t ntx,
                                                        unsigned long long *out /* [errors, compared] */)
{
    int e = 0, n = 0;
    for (int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t it = i - lag;
        if (it >= 0 && it < ntx) { n++; e += rx[i] != tx[it]; }
    }
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o); n += __shfl_down(n, o); }
    if ((threadIdx.x & 63) == 0) {
        if (e) atomicAdd(out, (unsigned long long)e);
        if (n) atomicAdd(out + 1, (unsigned long long)n);
    }
}

template <typename R>
int ser_dev(const void *E, int64_t N, const int32_t *idx_tx, int nmodes, int64_t ntx, const void *symbols, int M, int maxlag, int64_t window,
            int64_t trim, int64_t *result)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(N > 0 && nmodes >= 1 && nmodes <= 16 && M >= 1 && maxlag >= 0 && maxlag <= 4096 && trim >= 0 && 2 * trim < N,
               "ser: bad sizes");
    int64_t W = window > 0 ? window : 4096;
    if (W > N - 2 * trim) W = N - 2 * trim;
    const int64_t start = trim + (N - 2 * trim - W) / 2;            // window from the middle of the run
    const int nlag = 2 * maxlag + 1;
    void *buf = nullptr;
    const size_t b_idx4 = (size_t)4 * W * sizeof(int32_t), b_cnt = (size_t)nmodes * 4 * nlag * sizeof(unsigned);
    const size_t b_rx = (size_t)N * sizeof(int32_t);
    if ((rc = scratch(8, b_idx4 + b_cnt + b_rx + 64, &buf))) return rc;
    int32_t *idx4 = (int32_t *)buf;
    unsigned *counts = (unsigned *)((char *)buf + b_idx4);
    int32_t *rx = (int32_t *)((char *)buf + b_idx4 + b_cnt);
    unsigned long long *tot = (unsigned long long *)((char *)buf + b_idx4 + b_cnt + ((b_rx + 15) & ~(size_t)15));
    hipLaunchKernelGGL((ser_decide_kernel<R>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, g_stream, (const Cx<R> *)E, start, W, 0, 4,
                       (const Cx<R> *)symbols, M, idx4);
    hipLaunchKernelGGL(ser_search_kernel, dim3(nlag, 4 * nmodes), dim3(256), 0, g_stream, idx4, W, idx_tx, ntx, start, maxlag, counts);
    QH_HIP(hipGetLastError());
    std::vector<unsigned> h((size_t)nmodes * 4 * nlag);
    QH_HIP(hipMemcpyAsync(h.data(), counts, b_cnt, hipMemcpyDeviceToHost, g_stream));
    QH_HIP(hipStreamSynchronize(g_stream));
    size_t best = 0;
    for (size_t k = 1; k < h.size(); k++) if (h[k] > h[best]) best = k;
    const int mode = (int)(best / ((size_t)4 * nlag)), rot = (int)((best / nlag) & 3);
    const int64_t lag = (int64_t)(best % nlag) - maxlag;
    // whole row with the winning rotation, errors inside [trim, N - trim)
    hipLaunchKernelGGL((ser_decide_kernel<R>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, g_stream, (const Cx<R> *)E, (int64_t)0, N, rot, 1,
                       (const Cx<R> *)symbols, M, rx);
    QH_HIP(hipMemsetAsync(tot, 0, 2 * sizeof(unsigned long long), g_stream));
    unsigned nb = (unsigned)((N - 2 * trim + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(ser_count_kernel, dim3(nb), dim3(256), 0, g_stream, rx, idx_tx + (size_t)mode * ntx, trim, N - trim, lag, ntx, tot);
    QH_HIP(hipGetLastError());
    unsigned long long ht[2];
    QH_HIP(hipMemcpyAsync(ht, tot, sizeof(ht), hipMemcpyDeviceToHost, g_stream));
    QH_HIP(hipStreamSynchronize(g_stream));
    result[0] = (int64_t)ht[0]; result[1] = (int64_t)ht[1]; result[2] = mode; result[3] = rot; result[4] = lag;
    result[5] = (int64_t)h[best]; result[6] = W;
    return QH_OK;
}

}  // namespace qh

extern "C" {
int qh_ser_c64_dev(const void *E, int64_t N, const int32_t *idx_tx, int nmodes, int64_t ntx, const void *symbols, int M, int maxlag,
                   int64_t window, int64_t trim, int64_t *result)
{ return qh::ser_dev<float>(E, N, idx_tx, nmodes, ntx, symbols, M, maxlag, window, trim, result); }
int qh_ser_c128_dev(const void *E, int64_t N, const int32_t *idx_tx, int nmodes, int64_t ntx, const void *symbols, int M, int maxlag,
                    int64_t window, int64_t trim, int64_t *result)
{ return qh::ser_dev<double>(E, N, idx_tx, nmodes, ntx, symbols, M, maxlag, window, trim, result); }
}
