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
#include <atomic>
#include <cstdlib>
#include <cstring>

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
constexpr int BPS_DESC_RING = 16;       // descriptors / device-formed angle grids a thread rotates through (one per call)
template <typename R> struct AlphabetDesc {
    int product;            // 1: cartesian product detected
    int symmetric;          // 1: product alphabet whose sorted levels are mirror images on both axes (l_k == -l_{n-1-k}, n even)
    int nre, nim;
    R re[BPS_MAX_LEVELS], im[BPS_MAX_LEVELS];
};

// One wave.  Lanes 0..31 hold the distinct real parts found so far, lanes 32..63 the distinct imaginary parts; a symbol is
// matched against all of them with one ballot, so the pass over the alphabet costs a few instructions per symbol (the former
// single-thread version with private level arrays took 40 us per call - as long as a quarter of the search it prepares).
template <typename R>
__global__ void __launch_bounds__(64) analyse_alphabet_kernel(const Cx<R> *symbols, int M, AlphabetDesc<R> *d)
{
    __shared__ unsigned char seen[BPS_MAX_LEVELS * BPS_MAX_LEVELS];
    __shared__ R sorted[2 * BPS_MAX_LEVELS];
    const int lane = threadIdx.x;
    const bool imside = lane >= BPS_MAX_LEVELS;
    const int slot = lane & (BPS_MAX_LEVELS - 1);
    if (M > 1024) { if (lane == 0) { d->product = 0; d->symmetric = 0; d->nre = d->nim = 0; } return; }
    for (int k = lane; k < BPS_MAX_LEVELS * BPS_MAX_LEVELS; k += 64) seen[k] = 0;
    __syncthreads();
    R mine = 0;
    int nre = 0, nim = 0;
    bool ok = true;
    for (int k = 0; k < M && ok; k++) {
        const Cx<R> sk = symbols[k];                                    // wave-uniform
        const R v = imside ? sk.im : sk.re;
        const unsigned long long hit = __ballot(slot < (imside ? nim : nre) && mine == v);
        int r = __ffsll((unsigned long long)(hit & 0xffffffffull)) - 1, i = __ffsll((unsigned long long)(hit >> 32)) - 1;
        if (r < 0) { if (nre < BPS_MAX_LEVELS) { if (!imside && slot == nre) mine = v; r = nre++; } else ok = false; }
        if (i < 0) { if (nim < BPS_MAX_LEVELS) { if (imside && slot == nim) mine = v; i = nim++; } else ok = false; }
        // every (re, im) combination must occur exactly once: M distinct points on an nre x nim grid with M == nre*nim
        if (ok) {
            const bool dup = seen[r * BPS_MAX_LEVELS + i] != 0;       // same address in every lane
            __syncthreads();
            if (dup) ok = false; else if (lane == 0) seen[r * BPS_MAX_LEVELS + i] = 1;
            __syncthreads();
        }
    }
    ok = ok && (long)nre * nim == M;
    // sorted levels (rank = number of smaller levels of the same axis; levels are distinct)
    const int nmine = imside ? nim : nre;
    int rank = 0;
    for (int j = 0; j < BPS_MAX_LEVELS; j++) {
        const R other = __shfl(mine, (imside ? BPS_MAX_LEVELS : 0) + j);
        if (j < nmine && other < mine) rank++;
    }
    sorted[lane] = 0;
    __syncthreads();
    if (ok && slot < nmine) sorted[(imside ? BPS_MAX_LEVELS : 0) + rank] = mine;
    __syncthreads();
    // mirror symmetry lets the search run on |t| against the positive half only (|(-t) - (-l)| == |t - l| exactly)
    bool symlane = true;
    if (ok && slot < nmine) symlane = sorted[lane] == -sorted[(imside ? BPS_MAX_LEVELS : 0) + nmine - 1 - slot];
    const bool sym = ok && (nre % 2 == 0) && (nim % 2 == 0) && __all(symlane);
    if (imside) d->im[slot] = ok && slot < nim ? sorted[lane] : (R)0; else d->re[slot] = ok && slot < nre ? sorted[lane] : (R)0;
    if (lane == 0) { d->product = ok ? 1 : 0; d->symmetric = sym ? 1 : 0; d->nre = nre; d->nim = nim; }
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

// ------------------------------------------------------------------------------------------------ streaming form (complex64)
// The tile kernel above keeps one workgroup per CU busy with three phases separated by barriers; 2/3 of its time is spent in
// the window sums and the arg-min, with a quarter of the threads idle.  The streaming form removes the phases:
//   * one wave per chunk of C output symbols (one wave per workgroup: no workgroup barrier), lane <-> test angle; the rotator
//     of a lane stays in registers, the symbol of a step is wave-uniform (read once per 16 steps, broadcast with v_readlane);
//   * the distances of the last 2N symbols live in an LDS ring of the wave (2N x 64 floats): per symbol one read (the value
//     leaving the window), one write, s += entering, s -= leaving - the same sliding order as the tile kernel; every 128
//     symbols s is re-formed as the direct 2N-term sum of the ring (oldest to newest);
//   * the arg-min over the angles is taken 16 symbols at a time on a transposed 16 x 64 block in LDS: lane <-> (symbol,
//     quarter of the angles), 16 strided reads (conflict free at pitch 65), then two exchange steps over the quarters;
//   * all three alphabet kinds (mirror-symmetric product / product / anything) are handled in the kernel from the device-side
//     descriptor, so the host never waits for the analysis.
// ~35 wave instructions per symbol against ~75 (and no idle threads): bound by VALU issue, not by HBM (16 B per symbol).
constexpr int BS_G = 16;                 // symbols per arg-min block
constexpr int BS_TP = 65;                // pitch of the transposed block
constexpr int BS_REANCHOR = 8;           // groups between direct re-summations of the window
constexpr int BS_MAXRING = 192;          // rows (2N) the ring may take: 48 KiB

struct BpsStreamArgs {
    const Cx<float> *E;          // (nm, L)
    const float *angles;         // (A,)
    const Cx<float> *symbols;    // (M,)
    const AlphabetDesc<float> *desc;
    int32_t *idx;                // (nm, L)
    int64_t L;
    int A, M, N, C, alpha_lds;
    int chunk0;                  // chunk of block 0 (a launch may cover a part of the chunks: bps_dev's part / nparts)
    int fast_rows;               // 0: the generic rows only (qh_set_form("bps", "plain"): measurements)
    int reg_lo, reg_hi;          // chunks [reg_lo, reg_hi) belong to bps_stream40_kernel IF the alphabet turns out to be one it takes (bs40_takes: both kernels
                                 // read the device-side descriptor and exactly one of them works on such a chunk - the host never waits for the analysis)
    // REC (search + np.unwrap + de-rotation in this one kernel): per-symbol phase and recovered symbols out, and the look-back cells
    float *ph;                   // (nm, L)
    Cx<float> *Eout;             // (nm, L)
    unsigned long long *state;   // (nm, nchunk) zeroed before the launch: flag << 32 | value (1: the chunk's jump total, 2: inclusive prefix)
};

template <int K> struct BsKind { static constexpr int value = K; };
// First arg-min over the angles for the 16 symbols of a transposed block (dmin starts at 1000, strict `<`: pythran_dsp.py:31-41): lane <-> (symbol,
// quarter of the angles), 16 conflict-free reads, then two exchange steps over the quarters with ties to the lower index.  Selects, no branches.
__device__ __forceinline__ int bs_argmin16(const float *tb, int sym, int quarter)
{
    float m = 1000.f;
    int best = 0;
    const float *row = tb + sym * BS_TP + quarter * 16;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float v = row[k];
        const bool lt = v < m;
        m = lt ? v : m;
        best = lt ? quarter * 16 + k : best;
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o);
        const int b2 = __shfl_xor(best, o);
        const bool take = (m2 < m) | ((m2 == m) & (b2 < best));
        m = take ? m2 : m;
        best = take ? b2 : best;
    }
    return best;
}
__device__ __forceinline__ bool bs40_takes(const AlphabetDesc<float> *d);
template <typename R> __device__ __forceinline__ int unwrap_jump(const R *angles, int kprev, int kcur);
typedef float bs_f2 __attribute__((ext_vector_type(2)));
// both axes at once: u = (|t.re|, |t.im|), d_k = u - (l_re[k], l_im[k]) is ONE packed subtraction per level pair; the minima are
// per axis (no packed min).  The same values as the scalar form: subtraction and |.| are exact per component.
template <bool SMALL> __device__ __forceinline__ bs_f2 bs_axes_sym(bs_f2 t, const bs_f2 (&lv)[16], int nl)
{
    const bs_f2 u = {abs_(t.x), abs_(t.y)};
    const bs_f2 d0 = u - lv[0], d1 = u - lv[1], d2 = u - lv[2], d3 = u - lv[3];      // padding levels are 3e38: never the minimum
    bs_f2 m = {min_(min_(abs_(d0.x), abs_(d1.x)), min_(abs_(d2.x), abs_(d3.x))), min_(min_(abs_(d0.y), abs_(d1.y)), min_(abs_(d2.y), abs_(d3.y)))};
    if (!SMALL && nl > 4) {                                                           // wave-uniform
        const bs_f2 d4 = u - lv[4], d5 = u - lv[5], d6 = u - lv[6], d7 = u - lv[7];
        m.x = min_(m.x, min_(min_(abs_(d4.x), abs_(d5.x)), min_(abs_(d6.x), abs_(d7.x))));
        m.y = min_(m.y, min_(min_(abs_(d4.y), abs_(d5.y)), min_(abs_(d6.y), abs_(d7.y))));
        if (nl > 8) {
#pragma unroll
            for (int k = 8; k < 16; k += 2) {
                const bs_f2 da = u - lv[k], db = u - lv[k + 1];
                m.x = min_(m.x, min_(abs_(da.x), abs_(db.x)));
                m.y = min_(m.y, min_(abs_(da.y), abs_(db.y)));
            }
        }
    }
    return m;
}

// REC: the host layer's np.unwrap and de-rotation (phaserecovery.py:155-158) in the same kernel.  The chunk keeps its indices in LDS,
// sums its unwrap jumps, and gets the jumps of everything before it through a decoupled look-back over one 64-bit cell per chunk
// (chunks are dispatched in order, so a predecessor is resident or done; a wave reads 64 cells at a time and stops at the first
// that already carries an inclusive prefix) - then walks the chunk once more, coalesced: phase, rotator, recovered symbol.
template <bool REC>
__global__ void __launch_bounds__(64) bps_stream_kernel(BpsStreamArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char bs_smem[];
    const int lane = threadIdx.x;
    const int W = 2 * a.N;
    float *ring = reinterpret_cast<float *>(bs_smem);                 // [W][64]
    float *tb = ring + (size_t)W * 64;                                // [BS_G][BS_TP]
    float *plev = tb + BS_G * BS_TP + 1;                              // [2][BPS_MAX_LEVELS] levels of a product alphabet
    unsigned char *cidx = reinterpret_cast<unsigned char *>(plev + 2 * BPS_MAX_LEVELS + 1);   // REC: [C + 1] index of output c0 - 1, then of the chunk
    char *after_idx = reinterpret_cast<char *>(cidx) + (REC ? ((a.C + 1 + 15) & ~15) : 0);
    // REC: [C] running unwrap correction of the chunk's symbols - in the ring's memory once the search is over (the search is bound by
    // the waves a CU holds: 5 KiB more LDS per wave cost it a fifth of its speed), behind the indices when the ring is too small
    int *ccorr = (REC && W * 64 < a.C) ? reinterpret_cast<int *>(after_idx) : reinterpret_cast<int *>(ring);
    Cx<float> *alpha = reinterpret_cast<Cx<float> *>(after_idx + ((REC && W * 64 < a.C) ? (size_t)a.C * sizeof(int) : 0));   // [M] any other alphabet (if it fits)
    const int64_t L = a.L;
    const Cx<float> *E = a.E + (size_t)blockIdx.y * L;
    int32_t *idx = a.idx + (size_t)blockIdx.y * L;
    const int64_t c0 = (int64_t)(blockIdx.x + a.chunk0) * a.C;
    const int64_t c1 = c0 + a.C < L ? c0 + a.C : L;                   // outputs [c0, c1)
    const int ngroups = (a.C + W - 1 + (REC ? 1 : 0) + BS_G - 1) / BS_G;      // REC: one output more at the front (the jump into the chunk needs index c0 - 1)
    const int64_t lstart = c0 + a.C - 1 + a.N - (int64_t)ngroups * BS_G + 1;   // first distance row; row l completes the window of output l - N

    // ---- alphabet
    const int product = a.desc->product, symmetric = a.desc->symmetric;
    const int nre = a.desc->nre, nim = a.desc->nim;
    if (!REC && (int)(blockIdx.x + a.chunk0) >= a.reg_lo && (int)(blockIdx.x + a.chunk0) < a.reg_hi && bs40_takes(a.desc)) return;      // (wave-uniform; before any barrier)
    bs_f2 lev2[16];                                                   // positive halves (re, im), in SGPRs
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lev2[k].x = symmetric && k < nre / 2 ? a.desc->re[nre / 2 + k] : 3.0e38f;
        lev2[k].y = symmetric && k < nim / 2 ? a.desc->im[nim / 2 + k] : 3.0e38f;
    }
    const int nlmax = nre > nim ? nre / 2 : nim / 2;
    if (product && !symmetric) {
        if (lane < BPS_MAX_LEVELS) { plev[lane] = a.desc->re[lane]; plev[BPS_MAX_LEVELS + lane] = a.desc->im[lane]; }
    } else if (!product && a.alpha_lds) {
        for (int k = lane; k < a.M; k += 64) alpha[k] = a.symbols[k];
    }
    for (int r = 0; r < W; r++) ring[r * 64 + lane] = 0.f;
    float cs = 1.f, sn = 0.f;
    if (lane < a.A) sincosf(a.angles[lane], &sn, &cs);
    const float bias = lane < a.A ? 0.f : 2000.f;                     // lanes without an angle never win (dmin starts at 1000)
    const bool full = a.A == 64;
    const bs_f2 rot_cs = {cs, sn}, rot_ns = {-sn, cs};
    __syncthreads();

    auto distance = [&](float tr, float ti, auto KIND) -> float {
        constexpr int kind = decltype(KIND)::value;
        float d0;
        if (kind == 0 || kind == 3) {                                 // 3: at most 4 positive levels per axis (up to 64-QAM), no branch at all
            const bs_f2 mm = bs_axes_sym<kind == 3>(bs_f2{tr, ti}, lev2, nlmax);
            d0 = fma_(mm.x, mm.x, mm.y * mm.y);
        } else if (kind == 1) {
            float mr = 3.0e38f, mi = 3.0e38f;
            for (int r = 0; r < nre; r++) mr = min_(mr, abs_(tr - plev[r]));
            for (int r = 0; r < nim; r++) mi = min_(mi, abs_(ti - plev[BPS_MAX_LEVELS + r]));
            d0 = fma_(mr, mr, mi * mi);
        } else {
            d0 = 1000.f;                                              // det_symbol: strict `<` from d0 = 1000 (:17-22)
            for (int k = 0; k < a.M; k++) {
                const Cx<float> sk = a.alpha_lds ? alpha[k] : a.symbols[k];
                const float dr = tr - sk.re, di = ti - sk.im;
                const float d = fma_(dr, dr, di * di);
                d0 = d < d0 ? d : d0;
            }
        }
        return d0 < 100.f ? d0 : 100.f;                               // dists start at 100 (:73, :83)
    };
    auto load_group = [&](int64_t lg) -> Cx<float> {
        int64_t l = lg + (lane & (BS_G - 1));
        l = l < 0 ? 0 : (l < L ? l : L - 1);
        return ldg(E + l);
    };

    float s = 0.f;
    int slot = 0;                                                     // ring row of the oldest distance = the next one written
    const int sym = lane & (BS_G - 1), quarter = lane >> 4;
    Cx<float> xnext = load_group(lstart);
    auto run = [&](auto KIND) {
    constexpr int kind = decltype(KIND)::value;
    for (int g = 0; g < ngroups; g++) {
        const int64_t lg = lstart + (int64_t)g * BS_G;
        const Cx<float> xg = xnext;
        if (g + 1 < ngroups) xnext = load_group(lg + BS_G);
        if (g > 0 && (g % BS_REANCHOR) == 0) {                        // direct 2N-term sum, oldest to newest
            float t = 0.f;
            int r = slot;
            for (int k = 0; k < W; k++) { t += ring[r * 64 + lane]; r = r + 1 == W ? 0 : r + 1; }
            s = t;
        }
        auto rows = [&](auto EDGE) {
#pragma unroll
            for (int k = 0; k < BS_G; k++) {
                const float xr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xg.re), k));
                const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xg.im), k));
                // (tr, ti) = xr (cs, sn) + xi (-sn, cs): a packed multiply and a packed fma with the roundings of the scalar form
                float *cell = ring + slot * 64 + lane;
                const float old = *cell;                                                 // the value leaving the window: read before the distance work, used after it
                const bs_f2 tt = __builtin_elementwise_fma(bs_f2{xr, xr}, rot_cs, bs_f2{xi, xi} * rot_ns);
                const float tr = tt.x, ti = tt.y;
                float d = distance(tr, ti, KIND);
                if (decltype(EDGE)::value && (lg + k < 0 || lg + k >= L)) d = 0.f;   // rows outside the capture only feed outputs that are forced to 0
                *cell = d;
                s += d;
                s -= old;
                slot = slot + 1 == W ? 0 : slot + 1;
                tb[k * BS_TP + lane] = full ? s : s + bias;
            }
        };
        // Fast rows (round 6): a full grid (A = 64: no lane without an angle), at most four positive levels per axis, the 16 rows inside the
        // capture and the 16 ring cells contiguous.  The symbols come through the SCALAR cache (uniform address: two s_load_dwordx16 per
        // group, nothing for the vector unit - the generic rows take them from a coalesced vector load with two v_readlane per row), the ring
        // cells and the transposed block are addressed with immediate offsets from one base register per group, and there is no bias select:
        // 17 vector instructions per row (rotation 2, |t| 2, level differences 4, minima 4, squared distance 3, window 2) against 22.
        // The same arithmetic in the same order: bit-identical distances, sums and indices.
        if (kind == 3 && full && a.fast_rows && lg >= 0 && lg + BS_G <= L && slot + BS_G <= W) {         // wave-uniform
            typedef const __attribute__((address_space(4))) bs_f2 *bs_cptr;
            const bs_cptr eg = (bs_cptr)(const void *)(E + lg);                          // (read-only for the whole launch: constant address space)
            bs_f2 xs[BS_G];
#pragma unroll
            for (int k = 0; k < BS_G; k++) xs[k] = eg[k];
            float *cell0 = ring + slot * 64 + lane;
#pragma unroll
            for (int k = 0; k < BS_G; k++) {
                const float old = cell0[k * 64];
                const bs_f2 tt = __builtin_elementwise_fma(bs_f2{xs[k].x, xs[k].x}, rot_cs, bs_f2{xs[k].y, xs[k].y} * rot_ns);
                const bs_f2 mm = bs_axes_sym<true>(tt, lev2, nlmax);
                float d = fma_(mm.x, mm.x, mm.y * mm.y);
                d = d < 100.f ? d : 100.f;
                cell0[k * 64] = d;
                s += d;
                s -= old;
                tb[k * BS_TP + lane] = s;
            }
            slot = slot + BS_G == W ? 0 : slot + BS_G;
        } else if (lg >= 0 && lg + BS_G <= L) rows(BsKind<0>{}); else rows(BsKind<1>{});        // wave-uniform
        __syncthreads();
        // ---- first arg-min over the angles for the 16 symbols of the block (dmin starts at 1000, strict `<`, :31-41)
        const int best = bs_argmin16(tb, sym, quarter);
        const int64_t i = lg + sym - a.N;
        const int bo = (i >= a.N && i < L - a.N) ? best : 0;
        if (quarter == 0 && i >= c0 && i < c1) idx[i] = bo;
        if (REC && quarter == 0 && i >= c0 - 1 && i < c1) cidx[i - c0 + 1] = (unsigned char)bo;
        __syncthreads();
    }
    };
    if (symmetric && nlmax <= 4) run(BsKind<3>{});                    // wave-uniform: the loop exists once per alphabet kind
    else if (symmetric) run(BsKind<0>{});
    else if (product) run(BsKind<1>{});
    else run(BsKind<2>{});
    if (!REC) return;
    __syncthreads();                                                    // the ring is free now
    // ---- np.unwrap of 4 * ph over the interior [N, L - N): the correction is an integer number of quarter turns, a prefix sum of jumps
    const int per = (a.C + 63) / 64;                                    // consecutive symbols per lane
    int local = 0;
    for (int r = 0; r < per; r++) {
        const int e = lane * per + r;
        const int64_t i = c0 + e;
        int j = 0;
        if (e < a.C && i < c1 && i > a.N && i < L - a.N) j = unwrap_jump<float>(a.angles, cidx[e], cidx[e + 1]);
        local += j;
        if (e < a.C) ccorr[e] = local;                                  // inclusive within the lane
    }
    int incl = local;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 63);
    const int lane_off = incl - local;
    // ---- jumps of the chunks before this one
    unsigned long long *cell = a.state + (size_t)blockIdx.y * gridDim.x;
    const int c = blockIdx.x;
    int before = 0;
    if (c > 0) {
        if (lane == 0) __hip_atomic_store(cell + c, (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int base = c - 1;
        for (;;) {
            const int k = base - lane;
            unsigned long long v = 2ull << 32;                          // before chunk 0: an inclusive prefix of 0
            if (k >= 0) v = __hip_atomic_load(cell + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned flag = (unsigned)(v >> 32);
            if (__ballot(flag == 0) != 0) { __builtin_amdgcn_s_sleep(2); continue; }      // a predecessor has not published yet
            const unsigned long long m2 = __ballot(flag == 2);
            const int val = (int)(unsigned)v;
            const int first2 = m2 ? __builtin_ctzll(m2) : 64;           // nearest cell with an inclusive prefix
            int part = lane <= first2 ? val : 0;
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            before += part;
            if (m2) break;
            base -= 64;
        }
    }
    if (lane == 0) __hip_atomic_store(cell + c, (2ull << 32) | (unsigned)(before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int r = 0; r < per; r++) {
        const int e = lane * per + r;
        if (e < a.C) ccorr[e] += before + lane_off;
    }
    __syncthreads();
    // ---- phase and de-rotation, consecutive symbols per lane
    float *ph = a.ph + (size_t)blockIdx.y * L;
    Cx<float> *out = a.Eout + (size_t)blockIdx.y * L;
    const float pi = 3.14159265358979323846f;
    for (int e = lane; e < a.C; e += 64) {
        const int64_t i = c0 + e;
        if (i >= c1) break;
        const bool interior = (i >= a.N && i < L - a.N);
        float p = a.angles[cidx[e + 1]];                                // edges keep the raw grid value of idx = 0 (phaserecovery.py:155 unwraps the interior only)
        if (interior) p += (pi / 2) * (float)ccorr[e];
        ph[i] = p;
        float sn, cs;
        sincos_<float>(p, &sn, &cs);
        const Cx<float> x = ldg(E + i);
        stg(out + i, Cx<float>{fma_(x.re, cs, -(x.im * sn)), fma_(x.re, sn, x.im * cs)});
    }
}

// ---- Register-ring form of the streaming search (round 6) for the shape the path's own callers run at full size: A = 64 test angles (no idle lane),
// N = 20 (window of 2N = 40 rows), a mirror-symmetric product alphabet with at most four positive levels per axis (QPSK ... 64-QAM), chunks whose rows
// all lie inside the capture.  The window of a lane (its 40 last distances) is private to the lane, so it does not have to live in LDS: here it is 40
// REGISTERS, addressed statically because the loop body is unrolled over lcm(16, 40) = 80 rows (five arg-min blocks).  Per row that removes both ring
// accesses (LDS traffic 3 -> 1 operation per row + the arg-min reads), and per wave 10 KiB of the 14 KiB of LDS: the number of resident waves is then set
// by the registers (~84: 5 per SIMD) instead of the LDS (11 per CU = 2.75 per SIMD), which is what the kernel's speed hangs on (measured round 3: 5 KiB
// more LDS per wave cost a fifth).  Symbols through the scalar cache, 17 vector instructions per row + 3.8 for the arg-min.  The SAME operations in the
// same order as bps_stream_kernel (window start at ring position (16 g) mod 40, direct re-summation oldest to newest every 128 rows): bit-identical.
__device__ __forceinline__ bool bs40_takes(const AlphabetDesc<float> *d)
{
    return d->symmetric != 0 && d->nre <= 8 && d->nim <= 8;
}
__global__ void __launch_bounds__(64) bps_stream40_kernel(BpsStreamArgs a)
{
    constexpr int W = 40, N = 20, GB = 5;                             // GB arg-min blocks of BS_G rows = 2 W rows per trip of the loop
    __shared__ float tb[BS_G * BS_TP + 1];
    if (!bs40_takes(a.desc)) return;                                  // (bps_stream_kernel does these chunks then)
    const int lane = threadIdx.x;
    const int64_t L = a.L;
    const Cx<float> *E = a.E + (size_t)blockIdx.y * L;
    int32_t *idx = a.idx + (size_t)blockIdx.y * L;
    const int64_t c0 = (int64_t)(blockIdx.x + a.chunk0) * a.C;
    const int64_t c1 = c0 + a.C < L ? c0 + a.C : L;
    const int ngroups = (a.C + W - 1 + BS_G - 1) / BS_G;
    const int64_t lstart = c0 + a.C - 1 + N - (int64_t)ngroups * BS_G + 1;
    const int nre = a.desc->nre, nim = a.desc->nim;
    bs_f2 lev2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {                                    // (only the first four are read: bs_axes_sym<true>)
        lev2[k].x = k < nre / 2 ? a.desc->re[nre / 2 + k] : 3.0e38f;
        lev2[k].y = k < nim / 2 ? a.desc->im[nim / 2 + k] : 3.0e38f;
    }
    float cs, sn;
    sincosf(a.angles[lane], &sn, &cs);
    const bs_f2 rot_cs = {cs, sn}, rot_ns = {-sn, cs};
    float ring[W];
#pragma unroll
    for (int r = 0; r < W; r++) ring[r] = 0.f;
    float s = 0.f;
    const int sym = lane & (BS_G - 1), quarter = lane >> 4;
    typedef const __attribute__((address_space(4))) bs_f2 *bs_cptr;
    for (int g0 = 0; g0 < ngroups; g0 += GB) {
#pragma unroll
        for (int gg = 0; gg < GB; gg++) {
            const int g = g0 + gg;
            if (g >= ngroups) break;                                  // wave-uniform
            const int64_t lg = lstart + (int64_t)g * BS_G;
            const bs_cptr eg = (bs_cptr)(const void *)(E + lg);
            bs_f2 xs[BS_G];
#pragma unroll
            for (int k = 0; k < BS_G; k++) xs[k] = eg[k];
            if (g > 0 && (g % BS_REANCHOR) == 0) {                    // direct 2N-term sum, oldest to newest
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < W; k++) t += ring[(gg * BS_G + k) % W];
                s = t;
            }
#pragma unroll
            for (int k = 0; k < BS_G; k++) {
                constexpr int dummy = 0; (void)dummy;
                const int r = (gg * BS_G + k) % W;                    // compile-time after unrolling
                const float old = ring[r];
                const bs_f2 tt = __builtin_elementwise_fma(bs_f2{xs[k].x, xs[k].x}, rot_cs, bs_f2{xs[k].y, xs[k].y} * rot_ns);
                const bs_f2 mm = bs_axes_sym<true>(tt, lev2, 4);
                float d = fma_(mm.x, mm.x, mm.y * mm.y);
                d = d < 100.f ? d : 100.f;
                ring[r] = d;
                s += d;
                s -= old;
                tb[k * BS_TP + lane] = s;
            }
            __syncthreads();
            const int best = bs_argmin16(tb, sym, quarter);
            const int64_t i = lg + sym - N;
            const int bo = (i >= N && i < L - N) ? best : 0;
            if (quarter == 0 && i >= c0 && i < c1) idx[i] = bo;
            __syncthreads();
        }
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

// qh_set_form("bps", ...): 0 automatic, 1 tile kernel for complex64 too, 2 streaming kernel with the LDS ring only (no register-ring kernel), 3 fused
inline bool bps_reg_ring() { return form(FORM_BPS) == 0; }
template <typename R> inline bool bps_stream_ok(int64_t, int, int, int) { return false; }
template <> inline bool bps_stream_ok<float>(int64_t p, int A, int N, int M)
{
    const bool off = form(FORM_BPS) == 1;                              // the tile kernel forced (tests compare the two)
    return !off && p == 1 && A <= 64 && 2 * N <= BS_MAXRING && M >= 1;
}

// nm rows of length L (consecutive in memory) against one angle grid; idx likewise
template <typename R>
int bps_dev(const void *E, int64_t L, const void *angles, int64_t p, int A, const void *symbols, int M, int N, int32_t *idx, int nm = 1,
            void *ph = nullptr, void *Eout = nullptr, bool *recovered = nullptr, int part = 0, int nparts = 1)
{
    // part / nparts (streaming kernel only): this call searches the part-th of nparts runs of chunks - a caller that wants the search in
    // pieces (between the relaxation passes of the next capture's training, pipeline.py) makes nparts calls, each complete in itself (its own
    // alphabet analysis: the parts share no state).  Other kernels: part nparts - 1 does it all.
    if (recovered) *recovered = false;
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(L >= 0 && A >= 1 && M >= 1 && N >= 1 && nm >= 1, "bps: bad sizes");
    QH_REQUIRE(p == 1 || p == L, "bps: p must be either 1 or the length of the input signal");
    QH_REQUIRE(p == 1 || nm == 1, "bps: a per-symbol angle grid goes with a single row");
    if (L == 0) return QH_OK;
    QH_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, "bps: bad part of the search");
    const bool stream = bps_stream_ok<R>(p, A, N, M);
    if (!stream && part != nparts - 1) return QH_OK;
    // The alphabet is analysed by EVERY call (every part of a search in parts) into the next of BPS_DESC_RING descriptors of the calling thread: a part
    // carries no state of the part before it, and another search of the same thread in between - on this or another of its streams - cannot
    // overwrite a descriptor a kernel in flight still reads (round 5 kept ONE descriptor per thread, written by part 0 only).
    void *desc = nullptr;
    {
        void *ring = nullptr;
        static thread_local unsigned desc_next = 0;
        if ((rc = scratch(3, BPS_DESC_RING * sizeof(AlphabetDesc<double>), &ring))) return rc;
        desc = (char *)ring + (size_t)(desc_next++ % BPS_DESC_RING) * sizeof(AlphabetDesc<double>);
    }
    hipLaunchKernelGGL((analyse_alphabet_kernel<R>), dim3(1), dim3(64), 0, g_stream, (const Cx<R> *)symbols, M, (AlphabetDesc<R> *)desc);
    if (stream) {
        BpsStreamArgs s;
        s.E = (const Cx<float> *)E; s.angles = (const float *)angles; s.symbols = (const Cx<float> *)symbols;
        s.desc = (const AlphabetDesc<float> *)desc; s.idx = idx; s.L = L; s.A = A; s.M = M; s.N = N;
        s.alpha_lds = M <= 1024 ? 1 : 0;
        int C = 1024;                                            // longer chunks: less halo (2N - 1 rows each); shorter: more waves
        while (C > 128 && ((L + C - 1) / C) * nm < 4096) C /= 2;
        s.C = C;
        // search + unwrap + de-rotation in ONE kernel (qh_set_form("bps", "fused")).  Opt-in: measured at C3 it is slower than the search
        // followed by the three small unwrap / de-rotation launches (0.80 against 0.70 ms) - the tail of a chunk (jump scan, look-back,
        // sincos, second pass over the symbols) is a latency chain inside a kernel whose speed is the number of waves a CU holds.
        const int fused = form(FORM_BPS) == 3 ? 1 : 0;
        const bool rec = fused && ph != nullptr && Eout != nullptr && A <= 255 && nparts == 1;
        const unsigned nchunk = (unsigned)((L + C - 1) / C);
        const unsigned ch0 = (unsigned)((uint64_t)nchunk * part / nparts), ch1 = (unsigned)((uint64_t)nchunk * (part + 1) / nparts);
        s.chunk0 = (int)ch0;
        s.ph = (float *)ph; s.Eout = (Cx<float> *)Eout; s.state = nullptr;
        if (rec) {
            void *st = nullptr;
            if ((rc = scratch(12, (size_t)nm * nchunk * sizeof(unsigned long long), &st))) return rc;
            QH_HIP(hipMemsetAsync(st, 0, (size_t)nm * nchunk * sizeof(unsigned long long), g_stream));
            s.state = (unsigned long long *)st;
        }
        const size_t lds = ((size_t)2 * N * 64 + BS_G * BS_TP + 1 + 2 * BPS_MAX_LEVELS + 1) * sizeof(float) + (rec ? (2 * N * 64 < C ? (size_t)C * sizeof(int) : 0) + (((size_t)C + 1 + 15) & ~(size_t)15) : 0) +
                           (s.alpha_lds ? (size_t)M * sizeof(Cx<float>) : 0) + 16;
        static std::atomic<bool> sattr{false};
        if (!sattr) {
            QH_HIP(hipFuncSetAttribute((const void *)bps_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            QH_HIP(hipFuncSetAttribute((const void *)bps_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            sattr = true;
        }
        // chunks of the register-ring kernel (A = 64, N = 20, every row of the chunk inside the capture): [lo, hi) of this part
        s.reg_lo = s.reg_hi = 0;
        s.fast_rows = form(FORM_BPS) == 4 ? 0 : 1;
        int64_t lo = 0, hi = 0;
        if (!rec && A == 64 && N == 20 && C > 128) {
            const int ngroups = (C + 2 * N - 1 + BS_G - 1) / BS_G;
            const int64_t back = (int64_t)ngroups * BS_G - (C - 1 + N + 1);            // lstart = c0 - back >= 0
            lo = back > 0 ? (back + C - 1) / C : 0;
            hi = (L - N) / C;                                                            // (c + 1) C + N <= L
            if (lo < (int64_t)ch0) lo = ch0;
            if (hi > (int64_t)ch1) hi = ch1;
            if (hi <= lo) lo = hi = 0;
        }
        if (rec) hipLaunchKernelGGL(bps_stream_kernel<true>, dim3(nchunk, nm), dim3(64), lds, g_stream, s);
        else if (ch1 > ch0 && hi > lo) {
            // The chunks in front of and behind that range hold rows outside the capture and take the generic rows: ONE wave per chunk of 1024 symbols would
            // be a lone, latency-bound wave of ~0.1 ms behind a search of 0.36 ms - they are searched in pieces of 128 symbols instead (8 waves side by
            // side; whatever the form, so that the forms stay bit-identical).  The range itself goes to BOTH kernels: each looks at the device-side
            // descriptor and exactly one of them does the work (an alphabet the register kernel does not take: its waves return at once).
            auto edge = [&](int64_t c_from, int64_t c_to) {
                if (c_to <= c_from) return;
                BpsStreamArgs e = s;
                e.C = 128;
                e.chunk0 = (int)(c_from * (C / 128));
                const int64_t end = c_to * C < L ? c_to * C : L;
                const unsigned n = (unsigned)((end - c_from * C + 127) / 128);
                hipLaunchKernelGGL(bps_stream_kernel<false>, dim3(n, nm), dim3(64), lds, g_stream, e);
            };
            edge(ch0, lo);
            edge(hi, ch1);
            BpsStreamArgs r = s;
            r.chunk0 = (int)lo;
            if (bps_reg_ring()) {
                r.reg_lo = (int)lo; r.reg_hi = (int)hi;
                hipLaunchKernelGGL(bps_stream40_kernel, dim3((unsigned)(hi - lo), nm), dim3(64), 0, g_stream, r);
            }
            hipLaunchKernelGGL(bps_stream_kernel<false>, dim3((unsigned)(hi - lo), nm), dim3(64), lds, g_stream, r);
        } else if (ch1 > ch0)
            hipLaunchKernelGGL(bps_stream_kernel<false>, dim3(ch1 - ch0, nm), dim3(64), lds, g_stream, s);
        if (recovered) *recovered = rec;
        QH_HIP(hipGetLastError());
        return QH_OK;
    }
    size_t lds = 0;
    const int T = bps_tile<R>(A, N, &lds);
    QH_REQUIRE(T >= 8, "bps: averaging window 2N x test angles does not fit the LDS tile");
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        QH_HIP(hipFuncSetAttribute((const void *)bps_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BPS_LDS_BUDGET));
        QH_HIP(hipFuncSetAttribute((const void *)bps_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BPS_LDS_BUDGET));
        attr_set = true;
    }
    // run length of the sliding window sums: enough runs to keep every thread busy, at least 8 symbols each
    int run = (int)(((int64_t)T * A + BPS_THREADS - 1) / BPS_THREADS);
    if (run < 8) run = 8;
    if (run > T) run = T;
    for (int m = 0; m < nm; m++) {
        BpsArgs<R> a;
        a.E = (const Cx<R> *)E + (size_t)m * L; a.angles = (const R *)angles; a.symbols = (const Cx<R> *)symbols; a.idx = idx + (size_t)m * L;
        a.desc = (const AlphabetDesc<R> *)desc;
        a.L = L; a.p = p; a.A = A; a.M = M; a.N = N; a.T = T; a.RUN = run;
        const unsigned nblk = (unsigned)((L + T - 1) / T);
        hipLaunchKernelGGL((bps_kernel<R>), dim3(nblk), dim3(BPS_THREADS), lds, g_stream, a);
    }
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

__global__ void __launch_bounds__(1024) unwrap_scan_kernel(int *chunk_sum, int64_t nchunk)
{
    // exclusive scan of the chunk sums of one mode by one workgroup: a run of consecutive chunks per thread (loaded together), wave
    // scans of the run totals, the 16 wave totals through LDS.  (A single wave walking the array 64 entries at a time was a chain of
    // 64 dependent global round trips: 32 us for 4096 chunks.)
    __shared__ int wtot[16];
    int *cs = chunk_sum + (int64_t)blockIdx.x * nchunk;
    const int64_t len = (nchunk + 1023) / 1024;
    const int64_t i0 = (int64_t)threadIdx.x * len, i1 = i0 + len < nchunk ? i0 + len : nchunk;
    int tot = 0;
    for (int64_t i = i0; i < i1; i++) tot += cs[i];
    int incl = tot;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wtot[w];
    for (int64_t i = i0; i < i1; i++) { const int v = cs[i]; cs[i] = run; run += v; }
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
    // the scan wants consecutive symbols per thread, the memory system consecutive symbols per LANE: the running correction of
    // every symbol goes through LDS and the de-rotation walks the chunk in coalesced order (PMC: the per-thread-consecutive
    // form wrote 3.7 x the algorithmic bytes in partial lines)
    __shared__ int corr[UW_CHUNK];
#pragma unroll
    for (int r = 0; r < UW_PER_THREAD; r++) corr[threadIdx.x * UW_PER_THREAD + r] = off + jmp[r];
    __syncthreads();
    const R pi = (R)3.14159265358979323846;
#pragma unroll
    for (int r = 0; r < UW_PER_THREAD; r++) {
        const int e = r * UW_THREADS + threadIdx.x;
        const int64_t i = base + e;
        if (i < L) {
            const bool interior = (i >= N && i < L - N);
            const int k = ix[i];
            // edges keep the raw grid value of idx = 0, i.e. angles[0] (phaserecovery.py:155 unwraps the interior only)
            R p = angles[k];
            if (interior) p += (pi / 2) * (R)corr[e];
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
int bps_recover_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *symbols, int M, int N, int32_t *idx, void *ph, void *Eout,
                    int part = 0, int nparts = 1)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(nm >= 1 && L >= 1 && A >= 1, "bps_recover: bad sizes");
    // in pieces (qh_bps_recover_part_*): parts 0 .. nparts - 2 search a run of chunks each, the last part searches the rest, unwraps and de-rotates;
    QH_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, "bps_recover: bad part");
    const int64_t nchunk = (L + UW_CHUNK - 1) / UW_CHUNK;
    void *dang = const_cast<void *>(angles), *dchunk = nullptr;     // grow-only library scratch: no allocation / sync in the steady state
    if ((rc = scratch(1, (size_t)nm * nchunk * sizeof(int), &dchunk))) return rc;
    if (!dang) {                                                  // (formed by every part, into the next of a ring of grids: see the alphabet descriptors in bps_dev)
        void *ring = nullptr;
        static thread_local unsigned grid_next = 0;
        const size_t one = ((size_t)A * sizeof(R) + 255) & ~(size_t)255;
        if ((rc = scratch(0, BPS_DESC_RING * one, &ring))) return rc;
        dang = (char *)ring + (size_t)(grid_next++ % BPS_DESC_RING) * one;
        hipLaunchKernelGGL((linspace_kernel<R>), dim3((A + 63) / 64), dim3(64), 0, g_stream, (R *)dang, A);
    }
    bool recovered = false;
    if ((rc = bps_dev<R>(E, L, dang, 1, A, symbols, M, N, idx, nm, ph, Eout, &recovered, part, nparts))) return rc;
    if (recovered) return QH_OK;                                   // the streaming kernel unwrapped and de-rotated as well
    if (part != nparts - 1) return QH_OK;
    hipLaunchKernelGGL((unwrap_partial_kernel<R>), dim3((unsigned)nchunk, nm), dim3(UW_THREADS), 0, g_stream, idx, L, N, (const R *)dang,
                       (int *)dchunk, nchunk);
    hipLaunchKernelGGL(unwrap_scan_kernel, dim3(nm), dim3(1024), 0, g_stream, (int *)dchunk, nchunk);
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

// Pilot-aided phase trace (qampy/core/pilotbased_receiver.py:258-327, the tail of pilot_based_cpe_new): the averaged pilot phases
// knot_phase[k, j] at the symbol positions knots[j] (increasing) are interpolated linearly to every symbol - np.interp: constant before the
// first and after the last knot - and taken out: out[k, i] = E[k, i] exp(-1j trace[k, i]); trace is returned in the signal's complex dtype
// (real part; the reference casts it so).  The interpolation is formed in double like numpy's.
template <typename R>
__global__ void __launch_bounds__(256) pilot_phase_trace_kernel(const Cx<R> *E, int64_t L, const int64_t *knots, const double *kph, int nk, Cx<R> *out, Cx<R> *trace)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= L) return;
    const double *p = kph + (size_t)k * nk;
    double ph;
    if (i <= knots[0]) ph = p[0];
    else if (i >= knots[nk - 1]) ph = p[nk - 1];
    else {
        int lo = 0, hi = nk - 1;                                   // knots[lo] <= i < knots[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (knots[mid] <= i) lo = mid; else hi = mid; }
        const double slope = (p[hi] - p[lo]) / (double)(knots[hi] - knots[lo]);
        ph = slope * (double)(i - knots[lo]) + p[lo];
    }
    const R t = (R)ph;                                             // the trace in the signal's precision, as the reference's cast leaves it
    R sn, cs;
    if constexpr (sizeof(R) == 4) sincosf(-t, &sn, &cs); else sincos(-t, &sn, &cs);
    const Cx<R> x = ldg(E + (size_t)k * L + i);
    stg(out + (size_t)k * L + i, Cx<R>{x.re * cs - x.im * sn, x.re * sn + x.im * cs});
    stg(trace + (size_t)k * L + i, Cx<R>{t, (R)0});
}
template <typename R> int pilot_phase_trace_host(const void *E, int nmodes, int64_t L, const int64_t *knots, const double *kph, int nk, void *out, void *trace)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(nmodes >= 1 && L >= 0 && nk >= 1, "pilot_phase_trace: bad sizes");
    if (L == 0) return QH_OK;
    DevBuf dE, dk, dp, dout, dtr;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * sizeof(Cx<R>)))) return rc;
    if ((rc = dk.from_host(knots, (size_t)nk * sizeof(int64_t)))) return rc;
    if ((rc = dp.from_host(kph, (size_t)nmodes * nk * sizeof(double)))) return rc;
    if ((rc = dout.alloc((size_t)nmodes * L * sizeof(Cx<R>)))) return rc;
    if ((rc = dtr.alloc((size_t)nmodes * L * sizeof(Cx<R>)))) return rc;
    hipLaunchKernelGGL((pilot_phase_trace_kernel<R>), dim3((unsigned)((L + 255) / 256), nmodes), dim3(256), 0, g_stream, (const Cx<R> *)dE.p, L, (const int64_t *)dk.p,
                       (const double *)dp.p, nk, (Cx<R> *)dout.p, (Cx<R> *)dtr.p);
    QH_HIP(hipGetLastError());
    if ((rc = dout.to_host(out, dout.n))) return rc;
    if ((rc = dtr.to_host(trace, dtr.n))) return rc;
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
int qh_bps_recover_part_c64_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *s, int M, int N, int32_t *idx, void *ph, void *Eout, int part, int nparts)
{ return qh::bps_recover_dev<float>(E, nm, L, angles, A, s, M, N, idx, ph, Eout, part, nparts); }
int qh_bps_recover_part_c128_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *s, int M, int N, int32_t *idx, void *ph, void *Eout, int part, int nparts)
{ return qh::bps_recover_dev<double>(E, nm, L, angles, A, s, M, N, idx, ph, Eout, part, nparts); }
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
int qh_pilot_phase_trace_c64(const void *E, int nmodes, int64_t L, const int64_t *knots, const double *kph, int nk, void *out, void *trace)
{ return qh::pilot_phase_trace_host<float>(E, nmodes, L, knots, kph, nk, out, trace); }
int qh_pilot_phase_trace_c128(const void *E, int nmodes, int64_t L, const int64_t *knots, const double *kph, int nk, void *out, void *trace)
{ return qh::pilot_phase_trace_host<double>(E, nmodes, L, knots, kph, nk, out, trace); }
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
