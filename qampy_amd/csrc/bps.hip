// Blind phase search carrier recovery on gfx950.
//
// Reference behaviour:
//   qampy/core/pythran_dsp.py:45-85   bps: dists[i, a] = min_k |E[i]*exp(j*theta_a) - s_k|^2   (det_symbol :16-23)
//   qampy/core/pythran_dsp.py:26-42   select_angle_index(dists, 2N): idx[i] = first argmin_a sum_{l=i-N+1}^{i+N} dists[l, a]
//                                     for N <= i < L-N, else 0 (the reference forms the window sum as a difference of
//                                     running sums; here it is a direct 2N-term sum)
//   qampy/core/pythran_dsp.py:133-153 select_angles
//   qampy/core/phaserecovery.py:145-159 host layer: linspace grid, unwrap(4 ph)/4 on [N, L-N), E*exp(j ph)
//
// Bound: fp32 VALU (A*M distance evaluations per symbol, ~25 kflop against 20 B), not HBM.  The (L, A) distance matrix
// the reference materialises (1 GiB per mode at 2^22 symbols) never leaves the CU:
//   * one workgroup owns a tile of T output symbols and computes the (T + 2N - 1) x A distances of tile + halo into
//     LDS (thread <-> (symbol, angle) pairs, alphabet through the scalar cache);
//   * square-QAM alphabets (cartesian products, detected on the device) are searched per axis - bit-identical to the
//     full scan at 2*sqrt(M) instead of M candidates; anything else takes the brute-force scan;
//   * window sums are column sums over that LDS tile (lane <-> angle, conflict free): direct for the first symbol of a
//     short run, sliding for the rest, written to a padded LDS plane;
//   * one thread per symbol scans the A sums for the first strict minimum (tie -> lowest angle, like the reference).
#include "common.h"

namespace qh {

constexpr int BPS_THREADS = 1024;
constexpr size_t BPS_LDS_BUDGET = 120 * 1024;

template <typename R> __device__ __forceinline__ void sincos_(R x, R *s, R *c);
template <> __device__ __forceinline__ void sincos_<float>(float x, float *s, float *c) { sincosf(x, s, c); }
template <> __device__ __forceinline__ void sincos_<double>(double x, double *s, double *c) { sincos(x, s, c); }

// Alphabet descriptor built on the device by analyse_alphabet_kernel (no host round trip).  A square-QAM alphabet is
// a cartesian product {re levels} x {im levels}; then  min_k |t - s_k|^2 = fma(dr*, dr*, di* * di*)  with dr* / di* the
// per-axis minima of |t_re - a_r| / |t_im - b_i| - bit-identical to the brute-force scan (rounding is monotone and
// the product inside the fma is exact), at 2*sqrt(M) instead of M candidate evaluations.
constexpr int BPS_MAX_LEVELS = 32;
template <typename R> struct AlphabetDesc {
    int product;            // 1: cartesian product detected
    int symmetric;          // 1: product alphabet whose sorted levels are mirror images on both axes (l_k == -l_{n-1-k}, n even)
    int nre, nim;
    R re[BPS_MAX_LEVELS], im[BPS_MAX_LEVELS];
};

template <typename R>
__global__ void __launch_bounds__(64) analyse_alphabet_kernel(const Cx<R> *symbols, int M, AlphabetDesc<R> *d)
{
    __shared__ R sre[1024], sim[1024];
    __shared__ unsigned char seen[BPS_MAX_LEVELS * BPS_MAX_LEVELS];
    if (M > 1024) { if (threadIdx.x == 0) d->product = 0; return; }
    for (int k = threadIdx.x; k < M; k += 64) { const Cx<R> s = symbols[k]; sre[k] = s.re; sim[k] = s.im; }
    for (int k = threadIdx.x; k < BPS_MAX_LEVELS * BPS_MAX_LEVELS; k += 64) seen[k] = 0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    R lre[BPS_MAX_LEVELS], lim[BPS_MAX_LEVELS];
    int nre = 0, nim = 0;
    bool ok = true;
    for (int k = 0; k < M && ok; k++) {
        int r = 0, i = 0;
        while (r < nre && lre[r] != sre[k]) r++;
        if (r == nre) { if (nre < BPS_MAX_LEVELS) lre[nre++] = sre[k]; else ok = false; }
        while (i < nim && lim[i] != sim[k]) i++;
        if (i == nim) { if (nim < BPS_MAX_LEVELS) lim[nim++] = sim[k]; else ok = false; }
        // every (re, im) combination must occur exactly once: M distinct points on an nre x nim grid with M == nre*nim
        if (ok) { if (seen[r * BPS_MAX_LEVELS + i]) ok = false; else seen[r * BPS_MAX_LEVELS + i] = 1; }
    }
    ok = ok && (long)nre * nim == M;
    // sorted levels; mirror symmetry lets the search run on |t| against the positive half only (|(-t) - (-l)| == |t - l| exactly)
    bool sym = ok && (nre % 2 == 0) && (nim % 2 == 0);
    if (ok) {
        for (int a = 1; a < nre; a++) { R v = lre[a]; int b = a - 1; while (b >= 0 && lre[b] > v) { lre[b + 1] = lre[b]; b--; } lre[b + 1] = v; }
        for (int a = 1; a < nim; a++) { R v = lim[a]; int b = a - 1; while (b >= 0 && lim[b] > v) { lim[b + 1] = lim[b]; b--; } lim[b + 1] = v; }
        for (int a = 0; a < nre / 2 && sym; a++) sym = lre[a] == -lre[nre - 1 - a];
        for (int a = 0; a < nim / 2 && sym; a++) sym = lim[a] == -lim[nim - 1 - a];
    }
    for (int r = 0; r < BPS_MAX_LEVELS; r++) { d->re[r] = r < nre ? lre[r] : (R)0; d->im[r] = r < nim ? lim[r] : (R)0; }
    d->product = ok ? 1 : 0;
    d->symmetric = sym ? 1 : 0;
    d->nre = nre; d->nim = nim;
}

template <typename R> struct BpsArgs {
    const Cx<R> *E;        // (L,)
    const R *angles;       // (p, A)
    const Cx<R> *symbols;  // (M,)
    const AlphabetDesc<R> *desc;
    int32_t *idx;          // (L,)
    int64_t L, p;
    int A, M, N, T, RUN;
};

constexpr int BPS_EPT = 8;        // (symbol, angle) elements a thread carries through the candidate loop at once

template <typename R>
__global__ void __launch_bounds__(BPS_THREADS) bps_kernel(BpsArgs<R> a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int A = a.A, N = a.N, T = a.T;
    const int rows = T + 2 * N - 1;
    R *dist = reinterpret_cast<R *>(smem);                 // [rows][A]
    R *wsum = dist + (size_t)rows * A;                     // [T][A + 1]
    Cx<R> *rot = reinterpret_cast<Cx<R> *>(wsum + (size_t)T * (A + 1) + ((size_t)T * (A + 1) & 1));   // [A] rotators
    R *lev = reinterpret_cast<R *>(rot + A);               // [2][BPS_MAX_LEVELS] per-axis levels of a product alphabet
    const int64_t i0 = (int64_t)blockIdx.x * T;            // first output symbol of this tile
    const int64_t l0 = i0 - N + 1;                         // symbol index of dist row 0
    const bool per_symbol = a.p > 1;
    const bool product = a.desc->product != 0;
    const bool symmetric = a.desc->symmetric != 0;
    const int nre = product ? a.desc->nre : 0, nim = product ? a.desc->nim : 0;

    if (!per_symbol) {                                     // exp(j*theta_a) once per tile instead of once per (symbol, angle)
        for (int ja = threadIdx.x; ja < A; ja += BPS_THREADS) {
            R sn, cs;
            sincos_<R>(a.angles[ja], &sn, &cs);
            rot[ja] = Cx<R>{cs, sn};
        }
    }
    if (product && threadIdx.x < BPS_MAX_LEVELS) {
        lev[threadIdx.x] = threadIdx.x < nre ? a.desc->re[threadIdx.x] : (R)0;
        lev[BPS_MAX_LEVELS + threadIdx.x] = threadIdx.x < nim ? a.desc->im[threadIdx.x] : (R)0;
    }
    __syncthreads();
    // ---- phase 1: min-distance of every (symbol, test angle) of tile + halo; lanes <-> consecutive angles.  A thread
    // rotates BPS_EPT elements, then walks the candidates ONCE for all of them (one candidate fetch per 8 updates).
    const int total = rows * A;
    const int r_step = BPS_THREADS / A, ja_step = BPS_THREADS - r_step * A;      // (row, angle) advance of one thread-block stride
    for (int e0 = threadIdx.x; e0 < total; e0 += BPS_THREADS * BPS_EPT) {
        R tr[BPS_EPT], ti[BPS_EPT];
        bool ok[BPS_EPT];
        int rq = e0 / A, jq = e0 - rq * A;                 // one division per 8 elements; the rest advance incrementally
#pragma unroll
        for (int q = 0; q < BPS_EPT; q++) {
            const int e = e0 + q * BPS_THREADS;
            const bool in = e < total;
            const int r = in ? rq : rows - 1, ja = in ? jq : A - 1;
            rq += r_step; jq += ja_step;
            if (jq >= A) { jq -= A; rq++; }
            const int64_t l = l0 + r;
            ok[q] = e < total && l >= 0 && l < a.L;        // rows outside [0, L) only feed outputs that are forced to 0
            const int64_t lc = l < 0 ? 0 : (l < a.L ? l : a.L - 1);
            const Cx<R> x = ldg(a.E + lc);
            Cx<R> c;
            if (per_symbol) {
                R sn, cs;
                sincos_<R>(a.angles[(size_t)lc * A + ja], &sn, &cs);
                c = Cx<R>{cs, sn};
            } else {
                c = rot[ja];
            }
            tr[q] = fma_(x.re, c.re, -(x.im * c.im));
            ti[q] = fma_(x.re, c.im, x.im * c.re);
        }
        R d0[BPS_EPT];
        if (symmetric) {
            // mirror-symmetric levels: the nearest level of t is the mirror image of the nearest POSITIVE level of |t|, at exactly
            // the same distance - half the candidates, and |t|, |d| are source modifiers: two instructions per level pair
            R mr[BPS_EPT], mi[BPS_EPT];
#pragma unroll
            for (int q = 0; q < BPS_EPT; q++) mr[q] = mi[q] = (R)3.0e38;
            for (int r = nre / 2; r < nre; r += 2) {
                const R l0 = lev[r], l1 = lev[r + 1 < nre ? r + 1 : r];
#pragma unroll
                for (int q = 0; q < BPS_EPT; q++) mr[q] = min_(min_(mr[q], abs_(abs_(tr[q]) - l0)), abs_(abs_(tr[q]) - l1));
            }
            for (int r = nim / 2; r < nim; r += 2) {
                const R l0 = lev[BPS_MAX_LEVELS + r], l1 = lev[BPS_MAX_LEVELS + (r + 1 < nim ? r + 1 : r)];
#pragma unroll
                for (int q = 0; q < BPS_EPT; q++) mi[q] = min_(min_(mi[q], abs_(abs_(ti[q]) - l0)), abs_(abs_(ti[q]) - l1));
            }
#pragma unroll
            for (int q = 0; q < BPS_EPT; q++) d0[q] = fma_(mr[q], mr[q], mi[q] * mi[q]);
        } else if (product) {
            R mr[BPS_EPT], mi[BPS_EPT];
#pragma unroll
            for (int q = 0; q < BPS_EPT; q++) mr[q] = mi[q] = (R)3.0e38;
            for (int r = 0; r < nre; r++) {
                const R lv = lev[r];
#pragma unroll
                for (int q = 0; q < BPS_EPT; q++) mr[q] = min_(mr[q], abs_(tr[q] - lv));
            }
            for (int r = 0; r < nim; r++) {
                const R lv = lev[BPS_MAX_LEVELS + r];
#pragma unroll
                for (int q = 0; q < BPS_EPT; q++) mi[q] = min_(mi[q], abs_(ti[q] - lv));
            }
#pragma unroll
            for (int q = 0; q < BPS_EPT; q++) d0[q] = fma_(mr[q], mr[q], mi[q] * mi[q]);
        } else {
#pragma unroll
            for (int q = 0; q < BPS_EPT; q++) d0[q] = (R)1000.;            // det_symbol: strict `<` from d0 = 1000 (:17-22)
            for (int k = 0; k < a.M; k++) {
                const Cx<R> sk = a.symbols[k];                              // wave-uniform -> scalar load
#pragma unroll
                for (int q = 0; q < BPS_EPT; q++) {
                    const R dr = tr[q] - sk.re, di = ti[q] - sk.im;
                    const R d = fma_(dr, dr, di * di);
                    d0[q] = d < d0[q] ? d : d0[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BPS_EPT; q++) {
            const int e = e0 + q * BPS_THREADS;
            if (e < total) dist[e] = ok[q] ? (d0[q] < (R)100. ? d0[q] : (R)100.) : (R)0;   // dists start at 100 (:73, :83)
        }
    }
    __syncthreads();
    // ---- phase 2: windowed sums.  A thread owns one angle and a run of RUN consecutive output symbols: the first
    // window is a direct 2N-term sum, the following ones slide (add the entering row, subtract the leaving one); runs
    // are short, so the running sum is re-anchored every RUN symbols (the reference keeps ONE running sum per capture).
    const int nrun = (T + a.RUN - 1) / a.RUN;
    for (int e = threadIdx.x; e < nrun * A; e += BPS_THREADS) {
        const int ir = e / A, ja = e - ir * A;
        const int il0 = ir * a.RUN;
        const int il1 = il0 + a.RUN < T ? il0 + a.RUN : T;
        const R *col = dist + ja;
        R s = 0;
        for (int r = 0; r < 2 * N; r++) s += col[(size_t)(il0 + r) * A];
        wsum[(size_t)il0 * (A + 1) + ja] = s;
        for (int il = il0 + 1; il < il1; il++) {
            s += col[(size_t)(il + 2 * N - 1) * A];
            s -= col[(size_t)(il - 1) * A];
            wsum[(size_t)il * (A + 1) + ja] = s;
        }
    }
    __syncthreads();
    // ---- phase 3: first arg-min over the test angles (dmin starts at 1000, strict `<`, :31-41)
    for (int il = threadIdx.x; il < T; il += BPS_THREADS) {
        const int64_t i = i0 + il;
        if (i >= a.L) break;
        int best = 0;
        if (i >= N && i < a.L - N) {
            R dmin = (R)1000.;
            const R *row = wsum + (size_t)il * (A + 1);
            for (int ja = 0; ja < A; ja++) {
                const R v = row[ja];
                if (v < dmin) { dmin = v; best = ja; }
            }
        }
        a.idx[i] = best;
    }
}

template <typename R> static int bps_tile(int A, int N, size_t *lds)
{
    // largest T with (T + 2N - 1)*A + T*(A + 1) elements (+ the rotator table) inside the LDS budget
    const int64_t cap = (int64_t)(BPS_LDS_BUDGET / sizeof(R)) - 2 * A - 2 - 2 * BPS_MAX_LEVELS;
    int64_t T = (cap - (int64_t)(2 * N - 1) * A) / (2 * A + 1);
    if (T > 1024) T = 1024;
    if (T < 1) return 0;
    *lds = ((size_t)(T + 2 * N - 1) * A + (size_t)T * (A + 1) + 1 + 2 * (size_t)A + 2 * BPS_MAX_LEVELS) * sizeof(R);
    return (int)T;
}

template <typename R>
int bps_dev(const void *E, int64_t L, const void *angles, int64_t p, int A, const void *symbols, int M, int N, int32_t *idx)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(L >= 0 && A >= 1 && M >= 1 && N >= 1, "bps: bad sizes");
    QH_REQUIRE(p == 1 || p == L, "bps: p must be either 1 or the length of the input signal");
    if (L == 0) return QH_OK;
    size_t lds = 0;
    const int T = bps_tile<R>(A, N, &lds);
    QH_REQUIRE(T >= 8, "bps: averaging window 2N x test angles does not fit the LDS tile");
    void *desc = nullptr;
    if ((rc = scratch(3, sizeof(AlphabetDesc<R>), &desc))) return rc;
    hipLaunchKernelGGL((analyse_alphabet_kernel<R>), dim3(1), dim3(64), 0, g_stream, (const Cx<R> *)symbols, M, (AlphabetDesc<R> *)desc);
    BpsArgs<R> a;
    a.E = (const Cx<R> *)E; a.angles = (const R *)angles; a.symbols = (const Cx<R> *)symbols; a.idx = idx;
    a.desc = (const AlphabetDesc<R> *)desc;
    a.L = L; a.p = p; a.A = A; a.M = M; a.N = N; a.T = T;
    // run length of the sliding window sums: enough runs to keep every thread busy, at least 8 symbols each
    int run = (int)(((int64_t)T * A + BPS_THREADS - 1) / BPS_THREADS);
    if (run < 8) run = 8;
    if (run > T) run = T;
    a.RUN = run;
    static bool attr_set = false;
    if (!attr_set) {
        QH_HIP(hipFuncSetAttribute((const void *)bps_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BPS_LDS_BUDGET));
        QH_HIP(hipFuncSetAttribute((const void *)bps_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BPS_LDS_BUDGET));
        attr_set = true;
    }
    const unsigned nblk = (unsigned)((L + T - 1) / T);
    hipLaunchKernelGGL((bps_kernel<R>), dim3(nblk), dim3(BPS_THREADS), lds, g_stream, a);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R>
int bps_host(const void *E, int64_t L, const void *angles, int64_t p, int A, const void *symbols, int M, int N, int32_t *idx)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(L >= 0 && A >= 1 && M >= 1 && N >= 1, "bps: bad sizes");
    QH_REQUIRE(p == 1 || p == L, "bps: p must be either 1 or the length of the input signal");
    if (L == 0) return QH_OK;
    DevBuf dE, da, ds, di;
    if ((rc = dE.from_host(E, (size_t)L * sizeof(Cx<R>)))) return rc;
    if ((rc = da.from_host(angles, (size_t)p * A * sizeof(R)))) return rc;
    if ((rc = ds.from_host(symbols, (size_t)M * sizeof(Cx<R>)))) return rc;
    if ((rc = di.alloc((size_t)L * sizeof(int32_t)))) return rc;
    if ((rc = bps_dev<R>(dE.p, L, da.p, p, A, ds.p, M, N, (int32_t *)di.p))) return rc;
    if ((rc = di.to_host(idx, di.n))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

// ------------------------------------------------------------------------------------------------ unwrap + de-rotation
// Device form of phaserecovery.py:145-159 for the resident pipeline.  With the linspace grid 4*ph = -pi + 2*pi*k/A, so
// np.unwrap's correction is -2*pi when k jumps by more than A/2, +2*pi when it drops by more than A/2 and 0 otherwise
// (|jump| == A/2 maps to 0: numpy keeps dd = +-pi).  The running correction is an integer prefix sum - exact.
constexpr int UW_THREADS = 256;
constexpr int UW_PER_THREAD = 4;
constexpr int UW_CHUNK = UW_THREADS * UW_PER_THREAD;

// np.unwrap's decision for the step from grid entry kprev to kcur of 4 * ph, evaluated with the operations numpy applies to
// the array (numpy/lib/_function_base_impl.py unwrap: dd = diff(p); ddmod = mod(dd + pi, 2 pi) - pi; ddmod = pi where it is
// -pi and dd > 0; correction = ddmod - dd, forced to 0 where |dd| < pi) in the precision of the phase array, on the values
// of the grid the host built - a jump of exactly half the range sits on the edge of that rule and its outcome depends on the
// rounding of the grid.  Returns the correction in units of 2 pi.
template <typename R> __device__ __forceinline__ R fmod_(R a, R b);
template <> __device__ __forceinline__ float fmod_<float>(float a, float b) { return fmodf(a, b); }
template <> __device__ __forceinline__ double fmod_<double>(double a, double b) { return fmod(a, b); }
template <typename R> __device__ __forceinline__ int unwrap_jump(const R *angles, int kprev, int kcur)
{
    const R pi = (R)3.14159265358979323846, twopi = (R)6.28318530717958647692;
    const R dd = (R)4 * angles[kcur] - (R)4 * angles[kprev];
    if (abs_(dd) < pi) return 0;
    R m = fmod_<R>(dd + pi, twopi);
    if (m != 0 && m < 0) m += twopi;
    R ddmod = m - pi;
    if (ddmod == -pi && dd > 0) ddmod = pi;
    const R corr = ddmod - dd;
    return (int)rint((double)corr / 6.28318530717958647692);
}

template <typename R>
__global__ void __launch_bounds__(UW_THREADS) unwrap_partial_kernel(const int32_t *idx, int64_t L, int N, const R *angles, int *chunk_sum,
                                                                     int64_t nchunk)
{
    // sums of the jump indicators of one chunk of one mode; interior = [N, L-N), the first interior element has no jump
    __shared__ int red[UW_THREADS / 64];
    const int64_t mode = blockIdx.y;
    const int32_t *ix = idx + mode * L;
    const int64_t base = (int64_t)blockIdx.x * UW_CHUNK;
    int s = 0;
    for (int r = 0; r < UW_PER_THREAD; r++) {
        const int64_t i = base + threadIdx.x + (int64_t)r * UW_THREADS;
        if (i > N && i < L - N) s += unwrap_jump<R>(angles, ix[i - 1], ix[i]);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < UW_THREADS / 64; w++) t += red[w];
        chunk_sum[mode * nchunk + blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(64) unwrap_scan_kernel(int *chunk_sum, int64_t nchunk)
{
    // exclusive scan of the chunk sums of one mode by a single wave (nchunk is a few thousand at most)
    int *cs = chunk_sum + (int64_t)blockIdx.x * nchunk;
    int carry = 0;
    for (int64_t b = 0; b < nchunk; b += 64) {
        const int64_t i = b + threadIdx.x;
        int v = i < nchunk ? cs[i] : 0;
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += t;
        }
        if (i < nchunk) cs[i] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}

template <typename R>
__global__ void __launch_bounds__(UW_THREADS) unwrap_apply_kernel(const Cx<R> *E, const int32_t *idx, const int *chunk_off,
                                                                   int64_t L, int N, const R *angles, int64_t nchunk, R *ph, Cx<R> *Eout)
{
    __shared__ int wsum[UW_THREADS / 64];
    __shared__ int tsum[UW_THREADS];
    const int64_t mode = blockIdx.y;
    const int32_t *ix = idx + mode * L;
    const int64_t base = (int64_t)blockIdx.x * UW_CHUNK;
    // each thread owns UW_PER_THREAD consecutive symbols so the in-chunk prefix is a short serial run + a block scan
    const int64_t t0 = base + (int64_t)threadIdx.x * UW_PER_THREAD;
    int jmp[UW_PER_THREAD];
    int local = 0;
#pragma unroll
    for (int r = 0; r < UW_PER_THREAD; r++) {
        const int64_t i = t0 + r;
        int j = 0;
        if (i > N && i < L - N) j = unwrap_jump<R>(angles, ix[i - 1], ix[i]);
        local += j;
        jmp[r] = local;                       // inclusive within the thread
    }
    // block exclusive scan of `local`
    int incl = local;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    tsum[threadIdx.x] = incl - local;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += wsum[w];
    const int off = chunk_off[mode * nchunk + blockIdx.x] + woff + tsum[threadIdx.x];
    const R pi = (R)3.14159265358979323846;
#pragma unroll
    for (int r = 0; r < UW_PER_THREAD; r++) {
        const int64_t i = t0 + r;
        if (i < L) {
            const bool interior = (i >= N && i < L - N);
            const int k = ix[i];
            // edges keep the raw grid value of idx = 0, i.e. angles[0] (phaserecovery.py:155 unwraps the interior only)
            R p = angles[k];
            if (interior) p += (pi / 2) * (R)(off + jmp[r]);
            ph[mode * L + i] = p;
            R sn, cs;
            sincos_<R>(p, &sn, &cs);
            const Cx<R> x = ldg(E + mode * L + i);
            stg(Eout + mode * L + i, Cx<R>{fma_(x.re, cs, -(x.im * sn)), fma_(x.re, sn, x.im * cs)});
        }
    }
}

template <typename R> __global__ void linspace_kernel(R *angles, int A)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const R pi = (R)3.14159265358979323846;
    if (j < A) angles[j] = -pi / 4 + ((pi / 2) / (R)A) * (R)j;
}

// angles: the (A,) test-angle grid in HBM as the host layer builds it (np.linspace in double, cast to the signal's precision,
// phaserecovery.py:145), or nullptr for a grid formed on the device in the signal's precision
template <typename R>
int bps_recover_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *symbols, int M, int N, int32_t *idx, void *ph, void *Eout)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(nm >= 1 && L >= 1 && A >= 1, "bps_recover: bad sizes");
    const int64_t nchunk = (L + UW_CHUNK - 1) / UW_CHUNK;
    void *dang = const_cast<void *>(angles), *dchunk = nullptr;     // grow-only library scratch: no allocation / sync in the steady state
    if ((rc = scratch(1, (size_t)nm * nchunk * sizeof(int), &dchunk))) return rc;
    if (!dang) {
        if ((rc = scratch(0, (size_t)A * sizeof(R), &dang))) return rc;
        hipLaunchKernelGGL((linspace_kernel<R>), dim3((A + 63) / 64), dim3(64), 0, g_stream, (R *)dang, A);
    }
    for (int m = 0; m < nm; m++)
        if ((rc = bps_dev<R>((const Cx<R> *)E + (size_t)m * L, L, dang, 1, A, symbols, M, N, idx + (size_t)m * L))) return rc;
    hipLaunchKernelGGL((unwrap_partial_kernel<R>), dim3((unsigned)nchunk, nm), dim3(UW_THREADS), 0, g_stream, idx, L, N, (const R *)dang,
                       (int *)dchunk, nchunk);
    hipLaunchKernelGGL(unwrap_scan_kernel, dim3(nm), dim3(64), 0, g_stream, (int *)dchunk, nchunk);
    hipLaunchKernelGGL((unwrap_apply_kernel<R>), dim3((unsigned)nchunk, nm), dim3(UW_THREADS), 0, g_stream, (const Cx<R> *)E, idx,
                       (const int *)dchunk, L, N, (const R *)dang, nchunk, (R *)ph, (Cx<R> *)Eout);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

// ------------------------------------------------------------------------------------------------ select_angles
template <typename R>
__global__ void select_angles_kernel(const R *angles, int64_t p, int A, const int64_t *idx, int64_t L, R *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) out[i] = angles[(p > 1 ? (size_t)i * A : 0) + idx[i]];
}

template <typename R> int select_angles_host(const void *angles, int64_t p, int A, const int64_t *idx, int64_t L, void *out)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(p >= 1 && A >= 1 && L >= 0, "select_angles: bad sizes");
    QH_REQUIRE(p == 1 || p >= L, "select_angles: angle grid has fewer rows than idx");
    if (L == 0) return QH_OK;
    DevBuf da, di, dout;
    if ((rc = da.from_host(angles, (size_t)p * A * sizeof(R)))) return rc;
    if ((rc = di.from_host(idx, (size_t)L * sizeof(int64_t)))) return rc;
    if ((rc = dout.alloc((size_t)L * sizeof(R)))) return rc;
    hipLaunchKernelGGL((select_angles_kernel<R>), dim3((unsigned)((L + 255) / 256)), dim3(256), 0, g_stream, (const R *)da.p, p, A,
                       (const int64_t *)di.p, L, (R *)dout.p);
    QH_HIP(hipGetLastError());
    if ((rc = dout.to_host(out, dout.n))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

// ------------------------------------------------------------------------------------------------ make_decision
// pythran_equalisation.py:304-334 via det_symbol_argmin :233-236: np.abs distance, first arg-min.
template <typename R> __device__ __forceinline__ R hypot_(R a, R b);
template <> __device__ __forceinline__ float hypot_<float>(float a, float b) { return hypotf(a, b); }
template <> __device__ __forceinline__ double hypot_<double>(double a, double b) { return hypot(a, b); }

template <typename R>
__global__ void __launch_bounds__(256) make_decision_kernel(const Cx<R> *E, int64_t L, const Cx<R> *symbols, int M, Cx<R> *det,
                                                            R *dist, int32_t *idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const Cx<R> x = ldg(E + i);
    Cx<R> s0 = symbols[0];
    R best = hypot_<R>(x.re - s0.re, x.im - s0.im);
    int ib = 0;
    for (int k = 1; k < M; k++) {
        const Cx<R> s = symbols[k];
        const R d = hypot_<R>(x.re - s.re, x.im - s.im);
        if (d < best) { best = d; ib = k; s0 = s; }
    }
    if (det) stg(det + i, s0);             // det / dist are optional for device-resident callers
    if (dist) dist[i] = best;
    idx[i] = ib;
}

template <typename R> int make_decision_dev(const void *E, int64_t L, const void *symbols, int M, void *det, void *dist, int32_t *idx)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(L >= 0 && M >= 1, "make_decision: bad sizes");
    if (L == 0) return QH_OK;
    hipLaunchKernelGGL((make_decision_kernel<R>), dim3((unsigned)((L + 255) / 256)), dim3(256), 0, g_stream, (const Cx<R> *)E, L,
                       (const Cx<R> *)symbols, M, (Cx<R> *)det, (R *)dist, idx);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R> int make_decision_host(const void *E, int64_t L, const void *symbols, int M, void *det, void *dist, int32_t *idx)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(L >= 0 && M >= 1, "make_decision: bad sizes");
    if (L == 0) return QH_OK;
    DevBuf dE, ds, dd, dr, di;
    if ((rc = dE.from_host(E, (size_t)L * sizeof(Cx<R>)))) return rc;
    if ((rc = ds.from_host(symbols, (size_t)M * sizeof(Cx<R>)))) return rc;
    if ((rc = dd.alloc((size_t)L * sizeof(Cx<R>)))) return rc;
    if ((rc = dr.alloc((size_t)L * sizeof(R)))) return rc;
    if ((rc = di.alloc((size_t)L * sizeof(int32_t)))) return rc;
    if ((rc = make_decision_dev<R>(dE.p, L, ds.p, M, dd.p, dr.p, (int32_t *)di.p))) return rc;
    if ((rc = dd.to_host(det, dd.n))) return rc;
    if ((rc = dr.to_host(dist, dr.n))) return rc;
    if ((rc = di.to_host(idx, di.n))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

// ------------------------------------------------------------------------------------------------ frequency-offset removal
// out[k, n] = E[k, n] * exp(-1j * 2 pi (n + 1) fo[k] / os)   (qampy/core/phaserecovery.py:435-473: t = arange(1, L + 1)); the phase is
// formed in double and reduced modulo one turn before the sine / cosine, whatever the precision of the signal
template <typename R>
__global__ void __launch_bounds__(256) comp_freq_offset_kernel(const Cx<R> *E, int64_t L, const double *fo, int os, Cx<R> *out)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (n >= L) return;
    double turns = (double)(n + 1) * fo[k] / (double)os;
    turns -= rint(turns);
    double sn, cs;
    sincos(-6.283185307179586476925 * turns, &sn, &cs);
    const Cx<R> x = ldg(E + (size_t)k * L + n);
    stg(out + (size_t)k * L + n, Cx<R>{(R)((double)x.re * cs - (double)x.im * sn), (R)((double)x.re * sn + (double)x.im * cs)});
}
template <typename R> int comp_freq_offset_host(const void *E, int nmodes, int64_t L, const double *fo, int os, void *out)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(nmodes >= 1 && L >= 0 && os >= 1, "comp_freq_offset: bad sizes");
    if (L == 0) return QH_OK;
    DevBuf dE, df, dout;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * sizeof(Cx<R>)))) return rc;
    if ((rc = df.from_host(fo, (size_t)nmodes * sizeof(double)))) return rc;
    if ((rc = dout.alloc((size_t)nmodes * L * sizeof(Cx<R>)))) return rc;
    hipLaunchKernelGGL((comp_freq_offset_kernel<R>), dim3((unsigned)((L + 255) / 256), nmodes), dim3(256), 0, g_stream, (const Cx<R> *)dE.p, L, (const double *)df.p, os,
                       (Cx<R> *)dout.p);
    QH_HIP(hipGetLastError());
    if ((rc = dout.to_host(out, dout.n))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

// ------------------------------------------------------------------------------------------------ error counter
__global__ void __launch_bounds__(256) count_errors_kernel(const int32_t *rx, const int32_t *tx, int64_t n, int64_t lag, int64_t ntx,
                                                           unsigned long long *count)
{
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t it = i - lag;
        if (it >= 0 && it < ntx) c += rx[i] != tx[it];
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

}  // namespace qh

extern "C" {
int qh_bps_c64(const void *E, int64_t L, const void *t, int64_t p, int A, const void *s, int M, int N, int32_t *idx)
{ return qh::bps_host<float>(E, L, t, p, A, s, M, N, idx); }
int qh_bps_c128(const void *E, int64_t L, const void *t, int64_t p, int A, const void *s, int M, int N, int32_t *idx)
{ return qh::bps_host<double>(E, L, t, p, A, s, M, N, idx); }
int qh_bps_c64_dev(const void *E, int64_t L, const void *t, int64_t p, int A, const void *s, int M, int N, int32_t *idx)
{ return qh::bps_dev<float>(E, L, t, p, A, s, M, N, idx); }
int qh_bps_c128_dev(const void *E, int64_t L, const void *t, int64_t p, int A, const void *s, int M, int N, int32_t *idx)
{ return qh::bps_dev<double>(E, L, t, p, A, s, M, N, idx); }
int qh_bps_recover_c64_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *s, int M, int N, int32_t *idx, void *ph, void *Eout)
{ return qh::bps_recover_dev<float>(E, nm, L, angles, A, s, M, N, idx, ph, Eout); }
int qh_bps_recover_c128_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *s, int M, int N, int32_t *idx, void *ph, void *Eout)
{ return qh::bps_recover_dev<double>(E, nm, L, angles, A, s, M, N, idx, ph, Eout); }
int qh_select_angles_f32(const void *angles, int64_t p, int A, const int64_t *idx, int64_t L, void *out)
{ return qh::select_angles_host<float>(angles, p, A, idx, L, out); }
int qh_select_angles_f64(const void *angles, int64_t p, int A, const int64_t *idx, int64_t L, void *out)
{ return qh::select_angles_host<double>(angles, p, A, idx, L, out); }
int qh_make_decision_c64(const void *E, int64_t L, const void *s, int M, void *det, void *dist, int32_t *idx)
{ return qh::make_decision_host<float>(E, L, s, M, det, dist, idx); }
int qh_make_decision_c128(const void *E, int64_t L, const void *s, int M, void *det, void *dist, int32_t *idx)
{ return qh::make_decision_host<double>(E, L, s, M, det, dist, idx); }
int qh_make_decision_c64_dev(const void *E, int64_t L, const void *s, int M, void *det, void *dist, int32_t *idx)
{ return qh::make_decision_dev<float>(E, L, s, M, det, dist, idx); }
int qh_make_decision_c128_dev(const void *E, int64_t L, const void *s, int M, void *det, void *dist, int32_t *idx)
{ return qh::make_decision_dev<double>(E, L, s, M, det, dist, idx); }
int qh_comp_freq_offset_c64(const void *E, int nmodes, int64_t L, const double *fo, int os, void *out)
{ return qh::comp_freq_offset_host<float>(E, nmodes, L, fo, os, out); }
int qh_comp_freq_offset_c128(const void *E, int nmodes, int64_t L, const double *fo, int os, void *out)
{ return qh::comp_freq_offset_host<double>(E, nmodes, L, fo, os, out); }
int qh_count_errors_dev(const int32_t *rx, const int32_t *tx, int64_t n, int64_t lag, int64_t ntx, unsigned long long *count_dev)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (n <= 0) return QH_OK;
    unsigned nb = (unsigned)((n + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(qh::count_errors_kernel, dim3(nb), dim3(256), 0, qh::g_stream, rx, tx, n, lag, ntx, count_dev);
    QH_HIP(hipGetLastError());
    return QH_OK;
}
}
