// Look-ahead form of the exact equaliser recurrence (complex, blind methods, fixed step size).
//
// The reference recurrence (pythran_equalisation.py:165-172)
//        y_i = w_i . x_i ,   c_i = mu * errfn(y_i) ,   w_{i+1} = w_i + c_i * conj(x_i)
// is strictly sequential, and a lone wave64 issues one VALU instruction per ~8 cycles (scripts/ubench/lat.hip), so
// the direct form (train_impl.h: dot product + 6-level cross-lane reduction + tap update = ~45 instructions per step)
// costs ~360 cycles per step.  Unrolling the tap update gives the algebraically identical
//        y_i = W . x_i + sum_{l=l0}^{i-1} c_l * G(l, i) ,      G(l, i) = sum_f conj(x_l[f]) * x_i[f]
// for any earlier tap state W = w_{l0}.  G depends on the capture only (not on the taps, the mode, the stage or the
// sweep), so it is computed once, chip-wide, by gram_slide_kernel.  With lanes <-> the 64 steps of a block the critical wave
// then needs per step: one 16 B load, the error function on its 64 pending outputs, two v_readlane and two complex
// multiply-adds - ~11 instructions, no cross-lane reduction, no tap update.  Three helper waves (one per remaining
// SIMD of the CU) keep the taps one block behind ( W_k = W_{k-1} + sum_{l in block k-1} c_l conj(x_l) ) and produce
// the prior outputs  Q_{k+1}[i] = W_k . x_i  of the block after next; one workgroup barrier per 64 steps.
//
// Result: the same numbers as the direct form up to the order of floating-point additions.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace qh {

constexpr int LA_B = 64;          // steps per block = lanes of the chain wave
constexpr int LA_NH = 3;          // helper waves
constexpr int LA_PD = 8;          // Gram-row prefetch distance (steps)
constexpr int LA_HB = 32;         // helper load batch (tap update)
constexpr int LA_MAXSLICE = 32;   // taps per helper held in registers by the prior dot products
constexpr int LA_MAXPART = 8;     // RDE/MRDE partitions handled by the vector select chain

template <typename R> struct GramPair { Cx<R> cur, next; };
// cur-only ("triangular") layout: row j of a block holds its 63 - j entries with target > j at gram_tri_row(j)
constexpr int GRAM_TRI = 2048;     // entries per block slot (2016 used)
__host__ __device__ constexpr int gram_tri_row(int j) { return j * (LA_B - 1) - j * (j - 1) / 2; }   // per (step l, lane i): G(l, blk+i) [i > l-blk] and G(l, blk+64+i)

// ------------------------------------------------------------------------------------------------ Gram precompute
// G(l, l+d) = sum_k sum_t conj(E[k, l os + t]) E[k, (l+d) os + t]  is, along a diagonal (fixed lag d), a SLIDING window sum of
// the products p_d(n) = sum_k conj(E[k,n]) E[k,n + d os]: one step further drops `os` products and adds `os` new ones.  A
// thread therefore owns (a half of) one diagonal of a 64-step block: it forms its first window directly and slides 31 times
// - (ntaps + 31 os) instead of 32 ntaps products per mode, 5x fewer for 41 taps - and stores entry (step j, target j + d)
// where the trainers expect it.  grid = number of 64-step blocks, 256 threads = 128 lags x 2 halves of the block.
// Rounding: a slid sum carries at most 31 x 2 os extra additions; the trainers see Gram terms good to a few 1e-7 relative
// either way (and every form of the trainer is compared with the oracle on its own).
template <typename R, bool PAIR>
__global__ void __launch_bounds__(256) gram_slide_kernel(const Cx<R> *E, int nmodes, int64_t L, int64_t Lp, int os, int ntaps, int64_t TrSyms, Cx<R> *G,
                                                         int64_t e_cs = 0, int64_t g_cs = 0)
{
    QH_WAVE_FIRST();
    E += (int64_t)blockIdx.y * e_cs; G += (int64_t)blockIdx.y * g_cs;          // channel bank: one table per channel, all in ONE launch (blockIdx.y)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Cx<R> *tile = reinterpret_cast<Cx<R> *>(smem);
    constexpr int NT = PAIR ? 2 * LA_B : LA_B;                    // targets reachable from a block: this block (+ the next one)
    constexpr int GS = PAIR ? 2 : 1;                              // Cx elements per (step, lane) entry
    const int64_t blk = (int64_t)blockIdx.x * LA_B;
    const int span = (NT - 1) * os + ntaps;                       // samples per mode covering steps blk .. blk + NT - 1
    for (int k = 0; k < nmodes; k++) {
        const int64_t s0 = blk * os;
        for (int s = threadIdx.x; s < span; s += 256) {
            const int64_t g = s0 + s;
            tile[k * span + s] = g < L ? ldg(E + (size_t)k * Lp + g) : Cx<R>{0, 0};
        }
    }
    // pair layout: entries a lane must never see (targets at or before the source step) are zeros.  The cur-only layout
    // (block-iterative trainer) stores just the 2016 entries with target > step of a block, row after row, in a 2048-entry slot
    if (PAIR) {
        for (int e = threadIdx.x; e < LA_B * LA_B; e += 256) {
            const int j = e >> 6, i = e & 63;
            if (i <= j) stg(G + ((size_t)(blk + j) * LA_B + i) * GS, Cx<R>{0, 0});
        }
    }
    __syncthreads();
    const int d = threadIdx.x & 127, half = threadIdx.x >> 7;     // lag, half of the block
    if (d == 0 || d >= NT) {                                       // PAIR = false: lags >= 64 do not exist
        return;
    }
    const int j0 = half * 32;
    int jend = j0 + 32;                                            // steps j with target j + d < NT
    if (jend > NT - d) jend = NT - d;
    if (j0 >= jend) return;
    // first window, directly
    R sr = 0, si = 0;
    for (int k = 0; k < nmodes; k++) {
        const Cx<R> *ra = tile + k * span + j0 * os, *rb = ra + d * os;
        for (int t = 0; t < ntaps; t++) {
            const Cx<R> a = ra[t], b = rb[t];
            sr = fma_(a.re, b.re, fma_(a.im, b.im, sr));          // conj(a) * b
            si = fma_(a.re, b.im, fma_(-a.im, b.re, si));
        }
    }
    for (int j = j0; j < jend; j++) {
        const int it = j + d;                                      // target relative to the block start
        const bool ok = blk + j < TrSyms && blk + it < TrSyms;
        const Cx<R> v = ok ? Cx<R>{sr, si} : Cx<R>{0, 0};
        if (PAIR) {
            if (it < LA_B) stg(G + ((size_t)(blk + j) * LA_B + it) * GS, v);
            else stg(G + ((size_t)(blk + j) * LA_B + (it - LA_B)) * GS + 1, v);
        } else {
            stg(G + (size_t)blockIdx.x * GRAM_TRI + gram_tri_row(j) + (d - 1), v);
        }
        if (j + 1 < jend) {                                        // slide: drop the first `os` products, add the next `os`
            for (int k = 0; k < nmodes; k++) {
                const Cx<R> *ra = tile + k * span + j * os, *rb = ra + d * os;
                for (int t = 0; t < os; t++) {
                    const Cx<R> a0 = ra[t], b0 = rb[t], a1 = ra[ntaps + t], b1 = rb[ntaps + t];
                    sr = fma_(a1.re, b1.re, fma_(a1.im, b1.im, sr));
                    si = fma_(a1.re, b1.im, fma_(-a1.im, b1.re, si));
                    sr = fma_(-a0.re, b0.re, fma_(-a0.im, b0.im, sr));
                    si = fma_(-a0.re, b0.im, fma_(a0.im, b0.re, si));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ error functions, vector form
// RDE / MRDE tables as a recursive struct of scalars (NPART is a compile-time constant): arrays inside the constants
// struct end up in scratch memory, scalars stay in (scalar) registers.
template <typename R, int N> struct PartTab {
    R part_re, part_im, code_re, code_im;      // partition p and the code that applies above it (code p+1)
    PartTab<R, N - 1> next;
};
template <typename R> struct PartTab<R, 0> {};

template <typename R, int N> __device__ __forceinline__ void tab_fill(PartTab<R, N> &t, const Cx<R> *sy, int p, int ncode)
{
    if constexpr (N > 0) {
        const Cx<R> part = sy[ncode + p], code = sy[p + 1];
        t.part_re = part.re; t.part_im = part.im; t.code_re = code.re; t.code_im = code.im;
        tab_fill<R, N - 1>(t.next, sy, p + 1, ncode);
    }
}
// partition_value (pythran_equalisation.py:4-9) for non-decreasing partitions: the code above the LAST partition below sq
template <typename R, int N, bool IM> __device__ __forceinline__ R tab_lookup(R sq, R r, const PartTab<R, N> &t)
{
    if constexpr (N > 0) {
        r = sq > (IM ? t.part_im : t.part_re) ? (IM ? t.code_im : t.code_re) : r;
        return tab_lookup<R, N - 1, IM>(sq, r, t.next);
    } else {
        return r;
    }
}

// float variant of the look-up without compares: with h = (sq - part) * 2^60 (sign-exact, zero only at equality),
// med3(r, code, h) = code if sq > part else r, because the codes are non-negative and non-decreasing.  The products for
// the two axes come from ONE v_pk_fma, so a partition costs 3 instructions for both axes instead of 4.
template <int N> __device__ __forceinline__ void tab_lookup_med3(float sqr, float sqi, float &rr, float &ri, const PartTab<float, N> &t)
{
    if constexpr (N > 0) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        constexpr float BIG = 1152921504606846976.0f;                 // 2^60
        const v2f h = v2f{sqr, sqi} * BIG - v2f{t.part_re, t.part_im} * BIG;
        rr = __builtin_amdgcn_fmed3f(rr, t.code_re, h.x);
        ri = __builtin_amdgcn_fmed3f(ri, t.code_im, h.y);
        tab_lookup_med3<N - 1>(sqr, sqi, rr, ri, t.next);
    }
}

// the codes of a table times m (the segment trainer folds the step size into them: see seg_errfn_d)
template <typename R, int N> __device__ __forceinline__ void tab_scale_codes(PartTab<R, N> &t, R m)
{
    if constexpr (N > 0) { t.code_re *= m; t.code_im *= m; tab_scale_codes<R, N - 1>(t.next, m); }
}

template <typename R, int NPART> struct LaConst {
    R mu, R_re, R_im, code0_re, code0_im;
    PartTab<R, NPART> tab;
};

// SCALE = true: returns c = mu * errfn(y) with mu folded into the scalar factor (fewer instructions on the critical
// path); SCALE = false: the plain error for the trace.  Written on 2-vectors so that hipcc emits v_pk_* for float.
template <typename R> struct V2 { typedef R type __attribute__((ext_vector_type(2))); };

// the four blind error functions of the form e = d(y) * y (d a real factor: cma, rde; or one factor per axis: mcma, mrde):
// the factor alone, for callers that want both e and a rotated copy of it straight from y (train_seg.h)
template <int METHOD> constexpr bool la_errfn_is_dy = METHOD == QH_M_CMA || METHOD == QH_M_SGNCMA || METHOD == QH_M_MCMA || METHOD == QH_M_RDE || METHOD == QH_M_MRDE;
template <typename R, int METHOD, int NPART, bool SCALE>
__device__ __forceinline__ auto la_errfn_d(Cx<R> y, const LaConst<R, NPART> &k)
{
    using v2 = typename V2<R>::type;
    const v2 yy = {y.re, y.im};
    if constexpr (METHOD == QH_M_CMA || METHOD == QH_M_SGNCMA) {
        R d;
        if constexpr (SCALE) { const R t = fma_(y.re, y.re, y.im * y.im); d = fma_(-k.mu, t, k.mu * k.R_re); }     // (R - |y|^2) mu
        else d = fma_(-y.im, y.im, fma_(-y.re, y.re, k.R_re));                  // R - |y|^2 in two instructions (the adaptive chain counts them)
        return d;
    } else if constexpr (METHOD == QH_M_MCMA) {
        const v2 Rc = {k.R_re, k.R_im};
        v2 d = Rc - yy * yy;
        if constexpr (SCALE) d = d * k.mu;
        return d;
    } else if constexpr (METHOD == QH_M_RDE) {
        const R sq = fma_(y.re, y.re, y.im * y.im);
        R d = tab_lookup<R, NPART, false>(sq, k.code0_re, k.tab) - sq;
        if constexpr (SCALE) d = d * k.mu;
        return d;
    } else if constexpr (METHOD == QH_M_MRDE) {
        const v2 sq = yy * yy;
        v2 r;
        if constexpr (sizeof(R) == 4) {
            float rr = k.code0_re, ri = k.code0_im;
            tab_lookup_med3<NPART>(sq.x, sq.y, rr, ri, k.tab);
            r = v2{rr, ri};
        } else {
            r = v2{tab_lookup<R, NPART, false>(sq.x, k.code0_re, k.tab), tab_lookup<R, NPART, true>(sq.y, k.code0_im, k.tab)};
        }
        v2 d = r - sq;
        if constexpr (SCALE) d = d * k.mu;
        return d;
    } else {
        return (R)0;       // (not of this form: callers test la_errfn_is_dy first; kept instantiable for generic lambdas that name it in a discarded branch)
    }
}

template <typename R, int METHOD, int NPART, bool SCALE>
__device__ __forceinline__ Cx<R> la_errfn(Cx<R> y, const LaConst<R, NPART> &k, Cx<R> sdata = Cx<R>{0, 0})
{
    using v2 = typename V2<R>::type;
    const v2 yy = {y.re, y.im};
    v2 e;
    if constexpr (la_errfn_is_dy<METHOD>) {
        e = yy * la_errfn_d<R, METHOD, NPART, SCALE>(y, k);
    } else if constexpr (METHOD == QH_M_CMA2) {
        const R x2r = fma_(y.re, y.re, -(y.im * y.im)), x2i = (R)2 * y.re * y.im;
        const R m = SCALE ? k.mu : (R)1;
        const R dr = (k.R_re - x2r) * m, di = (k.R_im - x2i) * m;
        e = v2{fma_(dr, y.re, -(di * y.im)), fma_(dr, y.im, di * y.re)};
    } else if constexpr (METHOD == QH_M_SBD_DATA) {        // :219-223, the training symbol of this step comes with the call
        const v2 s = {sdata.re, sdata.im};
        v2 d = (s - yy) * __builtin_elementwise_abs(s);
        if constexpr (SCALE) d = d * k.mu;
        e = d;
    } else if constexpr (METHOD == QH_M_SBD || METHOD == QH_M_MDDMA || METHOD == QH_M_DD) {
        // decision-directed on a square alphabet: the table holds the sorted per-axis levels (codes) and their midpoints
        // (partitions), see slicer_table_kernel in train_bi.h; nearest level per axis == nearest symbol (det_symbol :240-265)
        v2 s = {tab_lookup<R, NPART, false>(y.re, k.code0_re, k.tab), tab_lookup<R, NPART, true>(y.im, k.code0_im, k.tab)};
        const v2 ds = s - yy;
        if (!(fma_(ds.x, ds.x, ds.y * ds.y) < (R)1000)) s = v2{(R)1, (R)0};   // det_symbol's initial value survives (:258-259)
        v2 d;
        if constexpr (METHOD == QH_M_SBD) d = (s - yy) * __builtin_elementwise_abs(s);
        else if constexpr (METHOD == QH_M_MDDMA) d = (s * s - yy * yy) * yy;
        else d = s - yy;
        if constexpr (SCALE) d = d * k.mu;
        e = d;
    }
    return Cx<R>{e.x, e.y};
}

// ------------------------------------------------------------------------------------------------ the sweep kernel
template <typename R> struct LaArgs {
    const Cx<R> *E;
    Cx<R> *wx;
    const Cx<R> *symbols;
    Cx<R> *err;             // row pitch err_pitch, this sweep starts at column err_off
    const GramPair<R> *G;
    int gpair;              // block-iterative kernel only: 1 = G is the look-ahead pair layout (cur/next), 0 = cur only
    const R *mu;
    R *mu_out;              // block-iterative kernel with the adaptive step: final step size, else unused
    int64_t L, Lp, TrSyms, nsy, sy_pitch, err_pitch, err_off;   // L usable samples per row from E on, Lp row pitch (>= L: time chunks)     // symbols row m starts at symbols + m * sy_pitch
    int nmodes, ntaps, os, nsel, method, nch;
    // channel batch: blockIdx.y = channel; element strides between the channels' arrays (0 for a single capture)
    int64_t E_cs, wx_cs, err_cs, G_cs, mu_cs, mu_ms;     // mu_ms: stride between the step sizes of the selected modes (0: one mu)
    int64_t modes[16];
    unsigned long long *prof;   // optional [4 waves][4] cycle counters of workgroup 0 (qh_la_profile), else nullptr
    // segments of ONE sweep as the channels of a batch (parallel-in-time training, train_pit.h): seg != 0 -> channel c
    // trains the steps [c TrSyms + 64 min(c, seg_extra), ...) of the capture at E: TrSyms steps, one block more for the first
    // seg_extra segments, seg_tail (< 64) more for the last one; E, err and G are the whole sweep's arrays, wx per segment
    int seg;
    int64_t seg_extra, seg_tail;
    const int *skip;            // optional device flag: non-zero -> the launch does nothing (device-side early termination)
    int niter;                  // block-iterative kernel: sweeps over the same TrSyms steps inside ONE launch (taps and step size stay on chip;
                                // sweep `it` writes its errors at err_off + it * TrSyms); 0 / 1 = one sweep
    int dd_general = 0;         // block-iterative kernel, decision-directed methods: != 0 - `symbols` is the alphabet itself (nsy <= BI_GEN_MAXSYM entries per
                                // mode, any constellation: 32- / 128-QAM crosses) and every decision is det_symbol's scan over all of it; 0 - slicer tables
};

// per-channel view of the arrays of a launch (channel bank: fixed strides; segmented sweep: see LaArgs::seg)
template <typename R> struct LaView {
    const Cx<R> *E;
    Cx<R> *wx, *err;
    const GramPair<R> *G;
    int64_t L, TrSyms;
};
template <typename R> __device__ __forceinline__ LaView<R> la_view(const LaArgs<R> &a, int64_t ch)
{
    LaView<R> v;
    v.wx = a.wx + ch * a.wx_cs;
    if (a.seg) {
        const int64_t nx = ch < a.seg_extra ? ch : a.seg_extra;
        const int64_t start = ch * a.TrSyms + nx * LA_B;
        v.TrSyms = a.TrSyms + (ch < a.seg_extra ? LA_B : 0) + (ch == a.nch - 1 ? a.seg_tail : 0);
        v.E = a.E + start * a.os;
        v.err = a.err + start;
        v.G = a.G + start * (a.gpair ? LA_B : GRAM_TRI / 2 / LA_B);
        v.L = a.L - start * a.os;
    } else {
        v.TrSyms = a.TrSyms;
        v.E = a.E + ch * a.E_cs;
        v.err = a.err + ch * a.err_cs;
        v.G = a.G + ch * a.G_cs;
        v.L = a.L;
    }
    return v;
}

template <typename R> struct LaLds {
    Cx<R> cbuf[2][LA_B];             // chain -> helpers: c_l of the block just finished
    Cx<R> qbuf[LA_NH][2][LA_B];      // helpers -> chain: partial prior outputs of the block after next
    Cx<R> wbuf[LA_NH][64];           // a helper's tap slice, read back wave-uniformly for the prior dot products
};

// ADAPT (round 5): the reference's adaptive step (adapt_step, pythran_equalisation.py:12-16,171-172) on the chain wave.  The step size in force is a
// wave-uniform value; after step i >= 1 of a sweep it becomes mu / (1 + mu |e_{i-1}|^2) unless the errors of steps i and i - 1 agree in the sign of
// both components.  Kept as r = 1 / mu ( r += |e_{i-1}|^2 : one addition, no rounding drift through repeated divisions) with mu = 1 / r per step;
// the sign test runs on c = mu e (mu > 0: same signs, and c is at hand in vector registers).  One mode per step size: the host launches the modes
// in turn (adaptive = 1, mu carried) or hands every mode its own (adaptive = 2), as for the block-iterative form.  ~12 instructions per step more
// than the fixed step.  niter > 1: the reference's Niter loop inside the launch - taps stay in the helpers' registers, the step size on the chain wave.
struct LaYes { static constexpr bool value = true; };
struct LaNo { static constexpr bool value = false; };
template <typename R> __device__ __forceinline__ R la_recip(R r)
{
    if constexpr (sizeof(R) == 4) return __builtin_amdgcn_rcpf(r);      // 1 ulp; mu is re-derived from the exactly accumulated r every step
    else return (R)1 / r;
}
template <typename R, int METHOD, int NPART, bool ADAPT = false>
__global__ void __launch_bounds__(64 * (1 + LA_NH)) train_la_kernel(LaArgs<R> a)
{
    if (a.skip && *a.skip) return;
    __builtin_amdgcn_s_setprio(3);       // a sequential recurrence: ahead of any streaming kernel of another stream that shares the CU
    // independent captures of a channel bank / segments of a sweep (blockIdx.y): same shapes, own arrays
    const int64_t ch = blockIdx.y;
    const LaView<R> vw = la_view(a, ch);
    const Cx<R> *const aE = vw.E;
    Cx<R> *const awx = vw.wx;
    Cx<R> *const aerr = vw.err;
    const GramPair<R> *const aG = vw.G;
    const int64_t aL = vw.L;
    const R *const amu = a.mu + ch * a.mu_cs + (int64_t)blockIdx.x * a.mu_ms;
    extern __shared__ __attribute__((aligned(16))) char la_smem[];
    LaLds<R> &lds = *reinterpret_cast<LaLds<R> *>(la_smem);
    Cx<R> *lds_win = reinterpret_cast<Cx<R> *>(la_smem + sizeof(LaLds<R>));   // [LA_NH][2][nmodes][wpitch] helper sample windows
    const int lane = threadIdx.x & 63;
    // Roles: 0 = chain, 1..3 = helpers.  The chain wave issues ~4x the instructions of a helper; when several workgroups
    // share a CU (segment / channel batches) the roles are rotated with the workgroup index so that the chain waves of
    // co-resident workgroups land on different SIMDs instead of all on the one that hosts wave 0.
    const int wave = __builtin_amdgcn_readfirstlane(((threadIdx.x >> 6) + blockIdx.x + blockIdx.y) & 3);
    const int mode = (int)a.modes[blockIdx.x];
    const int ntot = a.nmodes * a.ntaps;
    const int64_t TrSyms = vw.TrSyms;
    const int nblk = (int)((TrSyms + LA_B - 1) / LA_B);
    const Cx<R> *sy = a.symbols + (size_t)mode * a.sy_pitch;
    const int nsweep = a.niter > 1 ? a.niter : 1;

    if (wave == 0) {
        // ============================================================ chain wave: lanes <-> the 64 steps of a block
        LaConst<R, NPART> K;
        K.mu = *amu;
        {
            const Cx<R> c0 = sy[0];
            K.R_re = c0.re; K.R_im = c0.im;
        }
        K.code0_re = K.R_re; K.code0_im = K.R_im;       // np.array_split(symbs, 2): NPART + 1 codes, then NPART partitions
        tab_fill<R, NPART>(K.tab, sy, 0, NPART + 1);
        const GramPair<R> *grow = aG;                  // Gram rows: uniform base + lane (scalar base + 32-bit lane offset: no vector address arithmetic per load)
        GramPair<R> ga[LA_PD], gb[LA_PD];               // two register sets: one is consumed while the other one loads
#pragma unroll
        for (int u = 0; u < LA_PD; u++) ga[u] = grow[u * LA_B + lane];
        // adaptive step: r = 1 / mu and mu itself (wave-uniform vector registers), the last step's c and |e|^2
        R r_ad = ADAPT ? (R)1 / K.mu : (R)0, mu_ad = K.mu;
        R cpr = 0, cpi = 0, sqp = 0;
        R mu_vec = 0;                                   // lane j: the step size step j of the block ran with
        Cx<R> sdat{0, 0};                               // data-aided: lane <-> the training symbol of its step
        // one LMS step in look-ahead form: c_j from lane j's (final) output, then both pending output sets move
        // (CHECK: only the partial last block of a sweep holds steps past TrSyms - the chain wave pays ~7 cycles for EVERY instruction it issues,
        // scalar ones included, so full blocks run without the test)
        auto step = [&](Cx<R> &y, Cx<R> &yn, const GramPair<R> &g, int j, auto CHECK, int nvalid) {
            R cr, ci;
            if constexpr (ADAPT) {
                const Cx<R> e = la_errfn<R, METHOD, NPART, false>(y, K, sdat);
                const R er = readlane(e.re, j), ei = readlane(e.im, j);        // e_j, wave-uniform
                cr = mu_ad * er; ci = mu_ad * ei;
                mu_vec = lane == j ? mu_ad : mu_vec;
                // both products positive <=> the smaller one is (one compare instead of two and a scalar AND; NaN: the run has diverged either way)
                const R pr = cr * cpr, pi = ci * cpi;
                bool keep = min_(pr, pi) > 0;
                if constexpr (decltype(CHECK)::value) keep = keep || j >= nvalid;
                r_ad += keep ? (R)0 : sqp;                                     // (sqp = 0 in front of a sweep's first step: no adaptation there)
                mu_ad = la_recip<R>(r_ad);
                cpr = cr; cpi = ci; sqp = fma_(er, er, ei * ei);
            } else {
                const Cx<R> c = la_errfn<R, METHOD, NPART, true>(y, K, sdat);
                cr = readlane(c.re, j); ci = readlane(c.im, j);                // c_j, wave-uniform
            }
            // steps >= nvalid of a partial last block have all-zero Gram rows: they change nothing
            if constexpr (ADAPT && sizeof(R) == 4) {
                // c sits in a vector register pair here: two packed FMAs per output set with the swizzles in the operand modifiers (hipcc builds
                // (-ci, ci) with two extra instructions per step, and the chain wave pays ~7 cycles for every instruction it issues)
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 c2 = {cr, ci};
                auto cmad = [&](Cx<R> &acc, const Cx<R> &gg) {
                    f2 a2 = {acc.re, acc.im};
                    const f2 g2 = {gg.re, gg.im};
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a2) : "v"(c2), "v"(g2));     // (-ci g.im, ci g.re) + acc
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a2) : "v"(c2), "v"(g2));                                  // (cr g.re, cr g.im) + that
                    acc.re = a2.x; acc.im = a2.y;
                };
                cmad(y, g.cur);
                cmad(yn, g.next);
            } else {
                y.re = fma_(cr, g.cur.re, fma_(-ci, g.cur.im, y.re));
                y.im = fma_(cr, g.cur.im, fma_(ci, g.cur.re, y.im));
                yn.re = fma_(cr, g.next.re, fma_(-ci, g.next.im, yn.re));
                yn.im = fma_(cr, g.next.im, fma_(ci, g.next.re, yn.im));
            }
        };
        unsigned long long t_wait = 0, t_work = 0, t_mark = __builtin_readcyclecounter();
        for (int it = 0; it < nsweep; it++) {
        Cx<R> *errow = aerr + (size_t)mode * a.err_pitch + a.err_off + (int64_t)it * TrSyms;
        Cx<R> ynext{0, 0};
        sqp = 0;                                                       // the reference adapts from the second step of a sweep on (:171)
        __syncthreads();                                               // barrier 0 of the sweep: Q_0 is ready
        for (int k = 0; k < nblk; k++) {
            const int64_t s0 = (int64_t)k * LA_B;
            const int nvalid = (int)((TrSyms - s0) < LA_B ? (TrSyms - s0) : LA_B);
            if constexpr (METHOD == QH_M_SBD_DATA) sdat = sy[s0 + lane < TrSyms ? s0 + lane : TrSyms - 1];
            Cx<R> y = ynext;
#pragma unroll
            for (int h = 0; h < LA_NH; h++) {
                const Cx<R> q = lds.qbuf[h][k & 1][lane];
                y.re += q.re; y.im += q.im;
            }
            ynext = Cx<R>{0, 0};
            const GramPair<R> *gr = grow + (size_t)s0 * LA_B;
            // rows behind this block: the next block's, or - last block of a sweep that is followed by another - block 0's again
            const GramPair<R> *gnx = (k + 1 == nblk && it + 1 < nsweep) ? grow : gr + (size_t)LA_B * LA_B;
            auto block = [&](auto CHECK) {
#pragma unroll 1
                for (int j0 = 0; j0 < LA_B; j0 += 2 * LA_PD) {              // keep this loop rolled: 16 steps per trip
#pragma unroll
                    for (int u = 0; u < LA_PD; u++) gb[u] = (gr + (size_t)(j0 + LA_PD) * LA_B)[u * LA_B + lane];
#pragma unroll
                    for (int u = 0; u < LA_PD; u++) step(y, ynext, ga[u], j0 + u, CHECK, nvalid);
                    const GramPair<R> *g2 = j0 + 2 * LA_PD < LA_B ? gr + (size_t)(j0 + 2 * LA_PD) * LA_B : gnx;
#pragma unroll
                    for (int u = 0; u < LA_PD; u++) ga[u] = g2[u * LA_B + lane];
#pragma unroll
                    for (int u = 0; u < LA_PD; u++) step(y, ynext, gb[u], j0 + LA_PD + u, CHECK, nvalid);
                }
            };
            if (ADAPT && nvalid < LA_B) block(LaYes{}); else block(LaNo{});
            // every lane now holds its final output: error trace + step-size-scaled errors for the tap update
            const Cx<R> e = la_errfn<R, METHOD, NPART, false>(y, K, sdat);
            Cx<R> c;
            if constexpr (ADAPT) c = Cx<R>{mu_vec * e.re, mu_vec * e.im};   // exactly the values the steps above used
            else c = la_errfn<R, METHOD, NPART, true>(y, K, sdat);
            if (lane >= nvalid) c = Cx<R>{0, 0};
            if (lane < nvalid) stg(errow + s0 + lane, e);
            lds.cbuf[k & 1][lane] = c;
            if (a.prof) { const unsigned long long t = __builtin_readcyclecounter(); t_work += t - t_mark; t_mark = t; }
            __syncthreads();                                           // barrier k+1
            if (a.prof) { const unsigned long long t = __builtin_readcyclecounter(); t_wait += t - t_mark; t_mark = t; }
        }
        }   // sweeps
        if constexpr (ADAPT) if (lane == 0) a.mu_out[ch * a.mu_cs + (int64_t)blockIdx.x * a.mu_ms] = mu_ad;
        if (a.prof && blockIdx.x == 0 && lane == 0) { a.prof[0] = t_work; a.prof[1] = t_wait; }
        return;
    }

    // ================================================================ helper waves: tap slice [f0, f1) of this mode
    const int h = wave - 1;
    const int per = (ntot + LA_NH - 1) / LA_NH;                        // <= 64 (checked on the host)
    const int f0 = h * per, f1 = (f0 + per) < ntot ? (f0 + per) : ntot;
    const int nf = f1 > f0 ? f1 - f0 : 0;
    // update layout: lane <-> tap f0 + lane
    const bool own = lane < nf;
    const int fl = own ? f0 + lane : 0;
    const int kf = fl / a.ntaps, tf = fl - kf * a.ntaps;
    Cx<R> *wrow = awx + (size_t)mode * ntot;
    Cx<R> w = own ? ldg(wrow + fl) : Cx<R>{0, 0};

    // Sample windows.  A helper stages, per 64-step block, the (63*os + ntaps) samples of every input mode into its own
    // LDS window with coalesced loads (issued early: they do not depend on the chain) and then reads its operands with
    // immediate-offset ds_reads: x_l[f] = win[k_f * wpitch + (l - s0) * os + t_f].
    const int os_ = a.os;
    const int wlen = (LA_B - 1) * os_ + a.ntaps;                      // samples per mode and block
    const int wpitch = (wlen + 1) & ~1;
    const int wsz = a.nmodes * wpitch + 2;                             // + a dummy slot for the surplus staging lanes
    Cx<R> *win_u = lds_win + (size_t)(h * 2 + 0) * wsz;                // window of the block being folded into the taps
    Cx<R> *win_p = lds_win + (size_t)(h * 2 + 1) * wsz;                // window of the block whose prior outputs are due
    constexpr int WREG = 8;                                            // staging registers per lane and window
    const int wtot = a.nmodes * wpitch;
    Cx<R> su[WREG], sp[WREG];
    int64_t soff[WREG];                                                // capture offset of window element lane + 64 q (block 0)
    int sdst[WREG];                                                    // its LDS slot; surplus lanes write a dummy slot
#pragma unroll
    for (int q = 0; q < WREG; q++) {
        const int e = lane + 64 * q;
        const bool v = e < wtot;
        const int k2 = v ? e / wpitch : 0, i2 = v ? e - k2 * wpitch : 0;
        soff[q] = (int64_t)k2 * a.Lp + i2;
        sdst[q] = v ? e : wtot;
    }
    auto stage_load = [&](Cx<R> (&r)[WREG], int kb) {
        const int64_t base = (int64_t)kb * LA_B * os_;
        if (base + wpitch <= aL) {                                    // whole window inside the capture: no clamping
            const Cx<R> *pb = aE + base;
#pragma unroll
            for (int q = 0; q < WREG; q++) r[q] = ldg(pb + soff[q]);
        } else {                                                       // the last block may reach past the capture
#pragma unroll
            for (int q = 0; q < WREG; q++) {
                const int64_t row = soff[q] / a.Lp * a.Lp;
                int64_t g = base + (soff[q] - row);
                if (g > aL - 1) g = aL - 1;
                r[q] = ldg(aE + row + g);
            }
        }
    };
    auto stage_store = [&](const Cx<R> (&r)[WREG], Cx<R> *win) {
#pragma unroll
        for (int q = 0; q < WREG; q++) win[sdst[q]] = r[q];
    };
    const int xoff_u = own ? kf * wpitch + tf : 0;                     // this lane's tap inside a window (update layout)
    // taps after block kb:  w += sum_l c_l conj(x_l);  c_l is zero for the steps past TrSyms of a partial last block
    auto update = [&](int kb) {
        const Cx<R> *xw = win_u + xoff_u;
        const Cx<R> *cb = lds.cbuf[kb & 1];
        if (os_ == 2) {                                                // the common case: immediate ds_read offsets
#pragma unroll 32
            for (int j = 0; j < LA_B; j++) {
                const Cx<R> c = cb[j];
                const Cx<R> x = xw[j * 2];
                w.re = fma_(c.re, x.re, fma_(c.im, x.im, w.re));
                w.im = fma_(c.im, x.re, fma_(-c.re, x.im, w.im));
            }
        } else {
#pragma unroll 8
            for (int j = 0; j < LA_B; j++) {
                const Cx<R> c = cb[j];
                const Cx<R> x = xw[j * os_];
                w.re = fma_(c.re, x.re, fma_(c.im, x.im, w.re));
                w.im = fma_(c.im, x.re, fma_(-c.re, x.im, w.im));
            }
        }
        if (!own) w = Cx<R>{0, 0};
    };
    // prior outputs of block kb from the current taps: lane <-> step kb*64 + lane
    auto prior = [&](int kb) {
        lds.wbuf[h][lane] = w;                                         // same wave writes and reads: LDS keeps order;
        const bool live = (int64_t)kb * LA_B + lane < TrSyms;          // lanes >= nf hold w = 0
        const Cx<R> *xs = win_p + lane * os_;
        Cx<R> acc{0, 0};
        int k2 = f0 / a.ntaps, t2 = f0 - k2 * a.ntaps;
        int off = k2 * wpitch + t2;                                    // wave-uniform window offset of tap f0 + f
        for (int f = 0; f < nf; f++) {
            const Cx<R> wv = lds.wbuf[h][f];
            const Cx<R> x = xs[off];
            acc.re = fma_(x.re, wv.re, fma_(-x.im, wv.im, acc.re));
            acc.im = fma_(x.re, wv.im, fma_(x.im, wv.re, acc.im));
            off++;
            if (++t2 == a.ntaps) { t2 = 0; off += wpitch - a.ntaps; }
        }
        lds.qbuf[h][kb & 1][lane] = live ? acc : Cx<R>{0, 0};
    };

    unsigned long long t_wait = 0, t_upd = 0, t_pri = 0, t_mark = __builtin_readcyclecounter();
    for (int it = 0; it < nsweep; it++) {
    stage_load(sp, 0);
    stage_store(sp, win_p);
    prior(0);
    __syncthreads();                                                   // barrier 0 of the sweep
    for (int k = 0; k < nblk; k++) {
#ifndef LA_SKIP_HELPER
        if (k >= 1) stage_load(su, k - 1);                             // both windows' loads go out first ...
        if (k + 1 < nblk) stage_load(sp, k + 1);
        if (k >= 1) { stage_store(su, win_u); update(k - 1); }         // ... -> W_k
        if (a.prof) { const unsigned long long t = __builtin_readcyclecounter(); t_upd += t - t_mark; t_mark = t; }
        if (k + 1 < nblk) { stage_store(sp, win_p); prior(k + 1); }    // Q_{k+1} = W_k . x
        if (a.prof) { const unsigned long long t = __builtin_readcyclecounter(); t_pri += t - t_mark; t_mark = t; }
#endif
        __syncthreads();                                               // barrier k+1
        if (a.prof) { const unsigned long long t = __builtin_readcyclecounter(); t_wait += t - t_mark; t_mark = t; }
    }
    stage_load(su, nblk - 1);
    stage_store(su, win_u);
    update(nblk - 1);                                                  // taps at the end of the sweep
    }   // sweeps
    if (a.prof && blockIdx.x == 0 && lane == 0) { a.prof[4 * wave] = t_upd; a.prof[4 * wave + 1] = t_pri; a.prof[4 * wave + 2] = t_wait; }
    if (own) stg(wrow + fl, w);
}

// ------------------------------------------------------------------------------------------------ host side
template <typename R> static size_t gram_bytes(int64_t TrSyms)
{
    const int64_t nblk = (TrSyms + LA_B - 1) / LA_B;
    return (size_t)(nblk * LA_B + 2 * LA_PD) * LA_B * sizeof(GramPair<R>);
}

// nch captures (nch, nmodes, L) -> nch Gram tables, gram_bytes() apart, in ONE scratch allocation
// L: usable samples per row starting at E, Lp: row pitch (0 = L), ch_stride: samples between channels (0 = nmodes * Lp)
template <typename R> int gram_build(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram, int nch = 1,
                                     int64_t Lp = 0, int64_t ch_stride = 0, void *into = nullptr)
{
    if (Lp <= 0) Lp = L;
    if (ch_stride <= 0) ch_stride = (int64_t)nmodes * Lp;
    int rc = ensure_init();
    if (rc) return rc;
    const size_t bytes = gram_bytes<R>(TrSyms);
    void *G0 = into;                                             // (the caller's buffer - gram_bytes(TrSyms) x nch - or the library's scratch)
    if (!G0 && (rc = scratch(4, bytes * (size_t)nch, &G0))) return rc;
    const int64_t nblk = (TrSyms + LA_B - 1) / LA_B;
    const size_t lds = (size_t)nmodes * ((2 * LA_B - 1) * os + ntaps) * sizeof(Cx<R>);
    QH_REQUIRE(lds <= 64 * 1024, "gram: nmodes*(127*os+ntaps) samples exceed the LDS tile");
    // rows past the last block are read by the prefetch queue only: keep them zero (one strided fill for the whole bank)
    QH_HIP(hipMemset2DAsync((char *)G0 + (size_t)nblk * LA_B * LA_B * sizeof(GramPair<R>), bytes, 0, (size_t)2 * LA_PD * LA_B * sizeof(GramPair<R>), (size_t)nch, g_stream));
    for (int c0 = 0; c0 < nch && nblk > 0; c0 += 65535) {          // (grid.y limit)
        const int nc = nch - c0 < 65535 ? nch - c0 : 65535;
        hipLaunchKernelGGL((gram_slide_kernel<R, true>), dim3((unsigned)nblk, (unsigned)nc), dim3(256), lds, g_stream, (const Cx<R> *)E + (size_t)c0 * ch_stride, nmodes, L, Lp,
                           os, ntaps, TrSyms, (Cx<R> *)((char *)G0 + bytes * (size_t)c0), ch_stride, (int64_t)(bytes / sizeof(Cx<R>)));
    }
    QH_HIP(hipGetLastError());
    *gram = G0;
    return QH_OK;
}

// can the look-ahead form run this configuration?
// sizes the look-ahead kernel can hold (this alone decides the Gram layout of a capture: pairs when true)
inline bool la_shape_ok(int nmodes, int ntaps, int os)
{
    const char *force = trainer_force();
    if ((force[0] == 'd' || force[0] == 'i')) return false;               // "direct" / "iterative": A/B measurements, tests
    if (nmodes * (((LA_B - 1) * os + ntaps + 1) & ~1) > 8 * 64) return false;      // helper window: 8 staging registers per lane
    return (nmodes * ntaps + LA_NH - 1) / LA_NH <= LA_MAXSLICE;
}

inline bool la_supported(int method, int adaptive, int nmodes, int ntaps, int os, int64_t TrSyms, int64_t nsy)
{
    (void)adaptive;                       // the adaptive step runs on the chain wave (one mode per step size: the caller launches accordingly)
    if (TrSyms < 2 * LA_B) return false;
    if (!la_shape_ok(nmodes, ntaps, os)) return false;
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: case QH_M_CMA2: case QH_M_MCMA: return true;
    case QH_M_RDE: case QH_M_MRDE: return nsy - (nsy + 1) / 2 >= 1 && nsy - (nsy + 1) / 2 <= LA_MAXPART;
    case QH_M_SBD_DATA: return nsy >= TrSyms;              // one training symbol per step of the sweep
    default: return false;
    }
}

template <typename R> static size_t la_lds_bytes(const LaArgs<R> &a)
{
    const int wpitch = ((LA_B - 1) * a.os + a.ntaps + 1) & ~1;
    return sizeof(LaLds<R>) + (size_t)LA_NH * 2 * (a.nmodes * wpitch + 2) * sizeof(Cx<R>);
}

template <typename R, int METHOD, bool ADAPT> static int launch_la_parts(const LaArgs<R> &a, int npart)
{
    dim3 grid(a.nsel, a.nch), block(64 * (1 + LA_NH));
    const size_t lds = la_lds_bytes(a);
#define QH_LA_NP(N) case N: hipLaunchKernelGGL((train_la_kernel<R, METHOD, N, ADAPT>), grid, block, lds, g_stream, a); break;
    switch (npart) {
        QH_LA_NP(1) QH_LA_NP(2) QH_LA_NP(3) QH_LA_NP(4) QH_LA_NP(5) QH_LA_NP(6) QH_LA_NP(7) QH_LA_NP(8)
    default: set_error("look-ahead trainer: unsupported partition count"); return QH_ERR_ARG;
    }
#undef QH_LA_NP
    return QH_OK;
}

template <typename R, bool ADAPT> static int launch_la_t(const LaArgs<R> &a)
{
    dim3 grid(a.nsel, a.nch), block(64 * (1 + LA_NH));
    const int npart = (int)(a.nsy - (a.nsy + 1) / 2);
    const size_t lds = la_lds_bytes(a);
    int rc = QH_OK;
    switch (a.method) {
    case QH_M_CMA: case QH_M_SGNCMA: hipLaunchKernelGGL((train_la_kernel<R, QH_M_CMA, 0, ADAPT>), grid, block, lds, g_stream, a); break;
    case QH_M_CMA2: hipLaunchKernelGGL((train_la_kernel<R, QH_M_CMA2, 0, ADAPT>), grid, block, lds, g_stream, a); break;
    case QH_M_MCMA: hipLaunchKernelGGL((train_la_kernel<R, QH_M_MCMA, 0, ADAPT>), grid, block, lds, g_stream, a); break;
    case QH_M_RDE: rc = launch_la_parts<R, QH_M_RDE, ADAPT>(a, npart); break;
    case QH_M_MRDE: rc = launch_la_parts<R, QH_M_MRDE, ADAPT>(a, npart); break;
    case QH_M_SBD_DATA: hipLaunchKernelGGL((train_la_kernel<R, QH_M_SBD_DATA, 0, ADAPT>), grid, block, lds, g_stream, a); break;
    default: return QH_ERR_METHOD;
    }
    if (rc) return rc;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

// adaptive: one step size per workgroup (mu in / mu_out); the caller launches one mode at a time (nsel = 1) when mu is carried from mode to mode
template <typename R> int launch_la(const LaArgs<R> &a, bool adaptive = false)
{
    return adaptive ? launch_la_t<R, true>(a) : launch_la_t<R, false>(a);
}

}  // namespace qh
