// Look-ahead form of the exact equaliser recurrence (complex, blind methods, fixed step size).
//
// The reference recurrence (pythran_equalisation.py:165-172)
//        y_i = w_i . x_i ,   c_i = mu * errfn(y_i) ,   w_{i+1} = w_i + c_i * conj(x_i)
// is strictly sequential, and a lone wave64 issues one VALU instruction per ~8 cycles (scripts/ubench/lat.hip), so
// the direct form (train_impl.h: dot product + 6-level cross-lane reduction + tap update = ~45 instructions per step)
// costs ~360 cycles per step.  Unrolling the tap update gives the algebraically identical
//        y_i = W . x_i + sum_{l=l0}^{i-1} c_l * G(l, i) ,      G(l, i) = sum_f conj(x_l[f]) * x_i[f]
// for any earlier tap state W = w_{l0}.  G depends on the capture only (not on the taps, the mode, the stage or the
// sweep), so it is computed once, chip-wide, by gram_kernel.  With lanes <-> the 64 steps of a block the critical wave
// then needs per step: one 16 B load, the error function on its 64 pending outputs, two v_readlane and two complex
// multiply-adds - ~11 instructions, no cross-lane reduction, no tap update.  Three helper waves (one per remaining
// SIMD of the CU) keep the taps one block behind ( W_k = W_{k-1} + sum_{l in block k-1} c_l conj(x_l) ) and produce
// the prior outputs  Q_{k+1}[i] = W_k . x_i  of the block after next; one workgroup barrier per 64 steps.
//
// Result: the same numbers as the direct form up to the order of floating-point additions.
#pragma once
#include "common.h"

namespace qh {

constexpr int LA_B = 64;          // steps per block = lanes of the chain wave
constexpr int LA_NH = 3;          // helper waves
constexpr int LA_PD = 8;          // Gram-row prefetch distance (steps)
constexpr int LA_HB = 32;         // helper load batch
constexpr int LA_MAXPART = 8;     // RDE/MRDE partitions handled by the vector select chain

template <typename R> struct GramPair { Cx<R> cur, next; };   // per (step l, lane i): G(l, blk+i) [i > l-blk] and G(l, blk+64+i)

// ------------------------------------------------------------------------------------------------ Gram precompute
// grid = number of 64-step blocks, 256 threads.  Thread (i = t & 63, q = t >> 6) produces the pairs of lane i for the
// 16 steps l = blk + 16 q .. +15, four steps at a time so that every LDS sample feeds 8 complex multiply-adds.
template <typename R>
__global__ void __launch_bounds__(256) gram_kernel(const Cx<R> *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms,
                                                   GramPair<R> *G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Cx<R> *tile = reinterpret_cast<Cx<R> *>(smem);
    const int64_t blk = (int64_t)blockIdx.x * LA_B;
    const int span = (2 * LA_B - 1) * os + ntaps;                 // samples per mode covering steps blk .. blk+127
    for (int k = 0; k < nmodes; k++) {
        const int64_t s0 = blk * os;
        for (int s = threadIdx.x; s < span; s += 256) {
            const int64_t g = s0 + s;
            tile[k * span + s] = g < L ? ldg(E + (size_t)k * L + g) : Cx<R>{0, 0};
        }
    }
    __syncthreads();
    const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
    const bool cur_ok = blk + i < TrSyms, next_ok = blk + LA_B + i < TrSyms;
    for (int l4 = 0; l4 < 16; l4 += 4) {
        const int j0 = q * 16 + l4;                               // first of four source steps (relative to blk)
        R cr[4] = {0, 0, 0, 0}, ci[4] = {0, 0, 0, 0}, nr[4] = {0, 0, 0, 0}, ni[4] = {0, 0, 0, 0};
        for (int k = 0; k < nmodes; k++) {
            const Cx<R> *row = tile + k * span;
            for (int t = 0; t < ntaps; t++) {
                const Cx<R> bc = row[i * os + t];                 // x_{blk+i}[f]
                const Cx<R> bn = row[(i + LA_B) * os + t];        // x_{blk+64+i}[f]
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const Cx<R> a = row[(j0 + u) * os + t];       // x_l[f], wave-uniform address (LDS broadcast)
                    // conj(a) * b = (ar br + ai bi) + j (ar bi - ai br)
                    cr[u] = fma_(a.re, bc.re, fma_(a.im, bc.im, cr[u]));
                    ci[u] = fma_(a.re, bc.im, fma_(-a.im, bc.re, ci[u]));
                    nr[u] = fma_(a.re, bn.re, fma_(a.im, bn.im, nr[u]));
                    ni[u] = fma_(a.re, bn.im, fma_(-a.im, bn.re, ni[u]));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u;
            const bool l_ok = blk + j < TrSyms;
            GramPair<R> p;
            const bool c_ok = l_ok && cur_ok && i > j;            // lanes <= j already hold their final output: add zero
            p.cur = c_ok ? Cx<R>{cr[u], ci[u]} : Cx<R>{0, 0};
            p.next = (l_ok && next_ok) ? Cx<R>{nr[u], ni[u]} : Cx<R>{0, 0};
            using V = typename Cx2T<R>::type;
            V *dst = reinterpret_cast<V *>(G + (size_t)(blk + j) * LA_B + i);
            V v0, v1;
            v0.x = p.cur.re; v0.y = p.cur.im; v1.x = p.next.re; v1.y = p.next.im;
            dst[0] = v0; dst[1] = v1;
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

template <typename R, int NPART> struct LaConst {
    R mu, R_re, R_im, code0_re, code0_im;
    PartTab<R, NPART> tab;
};

// SCALE = true: returns c = mu * errfn(y) with mu folded into the scalar factor (one multiply less on the critical
// path); SCALE = false: the plain error for the trace.
template <typename R, int METHOD, int NPART, bool SCALE>
__device__ __forceinline__ Cx<R> la_errfn(Cx<R> y, const LaConst<R, NPART> &k)
{
    Cx<R> e;
    const R m = SCALE ? k.mu : (R)1;
    if constexpr (METHOD == QH_M_CMA || METHOD == QH_M_SGNCMA) {
        const R d = (k.R_re - fma_(y.re, y.re, y.im * y.im)) * m;
        e.re = d * y.re; e.im = d * y.im;
    } else if constexpr (METHOD == QH_M_CMA2) {
        const R x2r = fma_(y.re, y.re, -(y.im * y.im)), x2i = (R)2 * y.re * y.im;
        const R dr = (k.R_re - x2r) * m, di = (k.R_im - x2i) * m;
        e.re = fma_(dr, y.re, -(di * y.im)); e.im = fma_(dr, y.im, di * y.re);
    } else if constexpr (METHOD == QH_M_MCMA) {
        e.re = ((k.R_re - y.re * y.re) * m) * y.re;
        e.im = ((k.R_im - y.im * y.im) * m) * y.im;
    } else if constexpr (METHOD == QH_M_RDE) {
        const R sq = fma_(y.re, y.re, y.im * y.im);
        const R d = (tab_lookup<R, NPART, false>(sq, k.code0_re, k.tab) - sq) * m;
        e.re = y.re * d; e.im = y.im * d;
    } else {   // QH_M_MRDE
        const R sqr = y.re * y.re, sqi = y.im * y.im;
        e.re = ((tab_lookup<R, NPART, false>(sqr, k.code0_re, k.tab) - sqr) * m) * y.re;
        e.im = ((tab_lookup<R, NPART, true>(sqi, k.code0_im, k.tab) - sqi) * m) * y.im;
    }
    return e;
}

// ------------------------------------------------------------------------------------------------ the sweep kernel
template <typename R> struct LaArgs {
    const Cx<R> *E;
    Cx<R> *wx;
    const Cx<R> *symbols;
    Cx<R> *err;             // row pitch err_pitch, this sweep starts at column err_off
    const GramPair<R> *G;
    const R *mu;
    int64_t L, TrSyms, nsy, err_pitch, err_off;
    int nmodes, ntaps, os, nsel, method;
    int64_t modes[16];
};

template <typename R> struct LaLds {
    Cx<R> cbuf[2][LA_B];             // chain -> helpers: c_l of the block just finished
    Cx<R> qbuf[LA_NH][2][LA_B];      // helpers -> chain: partial prior outputs of the block after next
    Cx<R> wbuf[LA_NH][64];           // a helper's tap slice, read back wave-uniformly for the prior dot products
};

template <typename R, int METHOD, int NPART>
__global__ void __launch_bounds__(64 * (1 + LA_NH)) train_la_kernel(LaArgs<R> a)
{
    __shared__ LaLds<R> lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mode = (int)a.modes[blockIdx.x];
    const int ntot = a.nmodes * a.ntaps;
    const int64_t TrSyms = a.TrSyms;
    const int nblk = (int)((TrSyms + LA_B - 1) / LA_B);
    const Cx<R> *sy = a.symbols + (size_t)mode * a.nsy;

    if (wave == 0) {
        // ============================================================ chain wave: lanes <-> the 64 steps of a block
        LaConst<R, NPART> K;
        K.mu = *a.mu;
        {
            const Cx<R> c0 = sy[0];
            K.R_re = c0.re; K.R_im = c0.im;
        }
        K.code0_re = K.R_re; K.code0_im = K.R_im;       // np.array_split(symbs, 2): NPART + 1 codes, then NPART partitions
        tab_fill<R, NPART>(K.tab, sy, 0, NPART + 1);
        Cx<R> *errow = a.err + (size_t)mode * a.err_pitch + a.err_off;
        Cx<R> ynext{0, 0};
        using V4 = typename Cx2T<R>::type;
        const GramPair<R> *grow = a.G + lane;           // this lane's column of the Gram rows
        GramPair<R> ga[LA_PD], gb[LA_PD];               // two register sets: one is consumed while the other one loads
#pragma unroll
        for (int u = 0; u < LA_PD; u++) ga[u] = grow[(size_t)u * LA_B];
        // one LMS step in look-ahead form: c_j from lane j's (final) output, then both pending output sets move
        auto step = [&](Cx<R> &y, Cx<R> &yn, const GramPair<R> &g, int j) {
            const Cx<R> c = la_errfn<R, METHOD, NPART, true>(y, K);
            const R cr = readlane(c.re, j), ci = readlane(c.im, j);        // c_j, wave-uniform
            // steps >= nvalid of a partial last block have all-zero Gram rows: they change nothing
            y.re = fma_(cr, g.cur.re, fma_(-ci, g.cur.im, y.re));
            y.im = fma_(cr, g.cur.im, fma_(ci, g.cur.re, y.im));
            yn.re = fma_(cr, g.next.re, fma_(-ci, g.next.im, yn.re));
            yn.im = fma_(cr, g.next.im, fma_(ci, g.next.re, yn.im));
        };
        __syncthreads();                                               // barrier 0: Q_0 is ready
        for (int k = 0; k < nblk; k++) {
            const int64_t s0 = (int64_t)k * LA_B;
            const int nvalid = (int)((TrSyms - s0) < LA_B ? (TrSyms - s0) : LA_B);
            Cx<R> y = ynext;
#pragma unroll
            for (int h = 0; h < LA_NH; h++) {
                const Cx<R> q = lds.qbuf[h][k & 1][lane];
                y.re += q.re; y.im += q.im;
            }
            ynext = Cx<R>{0, 0};
            const GramPair<R> *gr = grow + (size_t)s0 * LA_B;
#pragma unroll 1
            for (int j0 = 0; j0 < LA_B; j0 += 2 * LA_PD) {                  // keep this loop rolled: 16 steps per trip
#pragma unroll
                for (int u = 0; u < LA_PD; u++) gb[u] = gr[(size_t)(j0 + LA_PD + u) * LA_B];
#pragma unroll
                for (int u = 0; u < LA_PD; u++) step(y, ynext, ga[u], j0 + u);
#pragma unroll
                for (int u = 0; u < LA_PD; u++) ga[u] = gr[(size_t)(j0 + 2 * LA_PD + u) * LA_B];   // may be the next block's rows
#pragma unroll
                for (int u = 0; u < LA_PD; u++) step(y, ynext, gb[u], j0 + LA_PD + u);
            }
            // every lane now holds its final output: error trace + step-size-scaled errors for the tap update
            const Cx<R> e = la_errfn<R, METHOD, NPART, false>(y, K);
            Cx<R> c = la_errfn<R, METHOD, NPART, true>(y, K);           // exactly the values the steps above used
            if (lane >= nvalid) c = Cx<R>{0, 0};
            if (lane < nvalid) stg(errow + s0 + lane, e);
            lds.cbuf[k & 1][lane] = c;
            __syncthreads();                                           // barrier k+1
        }
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
    const Cx<R> *xl = a.E + (size_t)kf * a.L + tf;                     // x_l[f] = xl[l * os]
    Cx<R> *wrow = a.wx + (size_t)mode * ntot;
    Cx<R> w = own ? ldg(wrow + fl) : Cx<R>{0, 0};

    // The helpers' loads do not depend on the chain, so they are issued LA_HB at a time ahead of the arithmetic (one
    // memory round trip per batch instead of one per step) from per-lane base pointers with small uniform offsets; only
    // the partial last block of a sweep takes the clamped path.
    const int os_ = a.os;
    const int64_t Lrow = a.L;
    // prior outputs of block kb from the current taps: lane <-> step kb*64 + lane
    auto prior = [&](int kb) {
        lds.wbuf[h][lane] = w;                                         // same wave writes and reads: LDS keeps order;
        int64_t s = (int64_t)kb * LA_B + lane;                         // lanes >= nf hold w = 0
        const bool live = s < TrSyms;
        if (!live) s = TrSyms - 1;
        const Cx<R> *ps = a.E + s * os_;                               // x_s[f] = ps[k_f * L + t_f]
        Cx<R> acc{0, 0};
        for (int fb = 0; fb < nf; fb += LA_HB) {
            Cx<R> x[LA_HB];
            int k2 = (f0 + fb) / a.ntaps, t2 = (f0 + fb) - k2 * a.ntaps;   // walk (mode, tap) without a division per tap
            int64_t off = (int64_t)k2 * Lrow + t2;                         // wave-uniform element offset
#pragma unroll
            for (int u = 0; u < LA_HB; u++) {
                x[u] = ldg(ps + off);
                if (fb + u + 1 < nf) {                                     // stop at the slice's last tap
                    off++;
                    if (++t2 == a.ntaps) { t2 = 0; off += Lrow - a.ntaps; }
                }
            }
#pragma unroll
            for (int u = 0; u < LA_HB; u++) {
                const Cx<R> wv = lds.wbuf[h][(fb + u) & 63];              // zero beyond the slice
                acc.re = fma_(x[u].re, wv.re, fma_(-x[u].im, wv.im, acc.re));
                acc.im = fma_(x[u].re, wv.im, fma_(x[u].im, wv.re, acc.im));
            }
        }
        lds.qbuf[h][kb & 1][lane] = live ? acc : Cx<R>{0, 0};
    };
    // taps after block kb:  w += sum_l c_l conj(x_l);  c_l is zero for the steps past TrSyms of a partial last block
    auto update = [&](int kb) {
        const int64_t s0 = (int64_t)kb * LA_B;
        const Cx<R> *pl = xl + s0 * os_;                               // x_l[f_lane] = pl[(l - s0) * os]
        const bool full = s0 + LA_B <= TrSyms;
        for (int jb = 0; jb < LA_B; jb += LA_HB) {
            Cx<R> x[LA_HB];
            if (full) {
                const Cx<R> *pb = pl + jb * os_;
#pragma unroll
                for (int u = 0; u < LA_HB; u++) x[u] = ldg(pb + u * os_);
            } else {
#pragma unroll
                for (int u = 0; u < LA_HB; u++) {
                    int64_t l = s0 + jb + u;
                    if (l > TrSyms - 1) l = TrSyms - 1;
                    x[u] = ldg(xl + l * os_);
                }
            }
#pragma unroll
            for (int u = 0; u < LA_HB; u++) {
                const Cx<R> c = lds.cbuf[kb & 1][jb + u];
                w.re = fma_(c.re, x[u].re, fma_(c.im, x[u].im, w.re));
                w.im = fma_(c.im, x[u].re, fma_(-c.re, x[u].im, w.im));
            }
        }
        if (!own) w = Cx<R>{0, 0};
    };

    prior(0);
    __syncthreads();                                                   // barrier 0
    for (int k = 0; k < nblk; k++) {
        if (k >= 1) update(k - 1);                                     // -> W_k
        if (k + 1 < nblk) prior(k + 1);                                // Q_{k+1} = W_k . x
        __syncthreads();                                               // barrier k+1
    }
    update(nblk - 1);                                                  // taps at the end of the sweep
    if (own) stg(wrow + fl, w);
}

// ------------------------------------------------------------------------------------------------ host side
template <typename R> static size_t gram_bytes(int64_t TrSyms)
{
    const int64_t nblk = (TrSyms + LA_B - 1) / LA_B;
    return (size_t)(nblk * LA_B + 2 * LA_PD) * LA_B * sizeof(GramPair<R>);
}

template <typename R> int gram_build(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram)
{
    int rc = ensure_init();
    if (rc) return rc;
    const size_t bytes = gram_bytes<R>(TrSyms);
    void *G = nullptr;
    if ((rc = scratch(4, bytes, &G))) return rc;
    const int64_t nblk = (TrSyms + LA_B - 1) / LA_B;
    // rows past the last block are read by the prefetch queue only: keep them zero
    QH_HIP(hipMemsetAsync((char *)G + (size_t)nblk * LA_B * LA_B * sizeof(GramPair<R>), 0, (size_t)2 * LA_PD * LA_B * sizeof(GramPair<R>), g_stream));
    const size_t lds = (size_t)nmodes * ((2 * LA_B - 1) * os + ntaps) * sizeof(Cx<R>);
    QH_REQUIRE(lds <= 64 * 1024, "gram: nmodes*(127*os+ntaps) samples exceed the LDS tile");
    if (nblk > 0) hipLaunchKernelGGL((gram_kernel<R>), dim3((unsigned)nblk), dim3(256), lds, g_stream, (const Cx<R> *)E, nmodes, L, os, ntaps,
                                     TrSyms, (GramPair<R> *)G);
    QH_HIP(hipGetLastError());
    *gram = G;
    return QH_OK;
}

// can the look-ahead form run this configuration?
inline bool la_supported(int method, int adaptive, int nmodes, int ntaps, int64_t TrSyms, int64_t nsy)
{
    if (adaptive || TrSyms < 2 * LA_B) return false;
    if ((nmodes * ntaps + LA_NH - 1) / LA_NH > 64) return false;
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: case QH_M_CMA2: case QH_M_MCMA: return true;
    case QH_M_RDE: case QH_M_MRDE: return nsy - (nsy + 1) / 2 >= 1 && nsy - (nsy + 1) / 2 <= LA_MAXPART;
    default: return false;
    }
}

template <typename R, int METHOD> static int launch_la_parts(const LaArgs<R> &a, int npart)
{
    dim3 grid(a.nsel), block(64 * (1 + LA_NH));
#define QH_LA_NP(N) case N: hipLaunchKernelGGL((train_la_kernel<R, METHOD, N>), grid, block, 0, g_stream, a); break;
    switch (npart) {
        QH_LA_NP(1) QH_LA_NP(2) QH_LA_NP(3) QH_LA_NP(4) QH_LA_NP(5) QH_LA_NP(6) QH_LA_NP(7) QH_LA_NP(8)
    default: set_error("look-ahead trainer: unsupported partition count"); return QH_ERR_ARG;
    }
#undef QH_LA_NP
    return QH_OK;
}

template <typename R> int launch_la(const LaArgs<R> &a)
{
    dim3 grid(a.nsel), block(64 * (1 + LA_NH));
    const int npart = (int)(a.nsy - (a.nsy + 1) / 2);
    int rc = QH_OK;
    switch (a.method) {
    case QH_M_CMA: case QH_M_SGNCMA: hipLaunchKernelGGL((train_la_kernel<R, QH_M_CMA, 0>), grid, block, 0, g_stream, a); break;
    case QH_M_CMA2: hipLaunchKernelGGL((train_la_kernel<R, QH_M_CMA2, 0>), grid, block, 0, g_stream, a); break;
    case QH_M_MCMA: hipLaunchKernelGGL((train_la_kernel<R, QH_M_MCMA, 0>), grid, block, 0, g_stream, a); break;
    case QH_M_RDE: rc = launch_la_parts<R, QH_M_RDE>(a, npart); break;
    case QH_M_MRDE: rc = launch_la_parts<R, QH_M_MRDE>(a, npart); break;
    default: return QH_ERR_METHOD;
    }
    if (rc) return rc;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

}  // namespace qh
