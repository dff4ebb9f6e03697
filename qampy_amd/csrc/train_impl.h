// train_equaliser / train_equaliser_realvalued on gfx950.
//
// Reference behaviour: qampy/core/equalisation/pythran_equalisation.py:128-173 (complex), :78-108 (real),
// error functions :178-231 / :110-125, adapt_step :12-22, partition_value :4-9, det_symbol :240-265.
//
// Three kernel forms of the same recurrence live here and in the two headers below; train_dev() picks one per call:
//   direct (this file)      every method, adaptive step, real-valued trainer, window batches, tiny captures
//   look-ahead (train_la.h) cma / cma2 / sgncma / mcma with a fixed step: Gram terms, no reduction on the critical wave
//   block-iterative (train_bi.h)  rde / mrde / sbd / mddma / dd / sbd_data and every method with the adaptive step:
//                           fixed-point sweeps over 64-step blocks on 8 wavefronts
// All three give the reference's numbers up to the order of floating-point additions (tests compare them pairwise).
//
// Direct form.  The recurrence  w[i+1] = w[i] + mu*e(w[i].x[i])*conj(x[i])  is strictly sequential in i, so one output mode
// is ONE dependent chain and this kernel is latency bound, not HBM or MFMA bound.  Mapping:
//   * one wave64 per chain; the nmodes*ntaps taps are spread round-robin over the 64 lanes (TPL taps per lane) and
//     live in VGPRs for the whole sweep;
//   * per step every lane multiplies its taps with its samples, a 6-level DPP butterfly (row_* / row_bcast) sums
//     re and im across the wave, the error function is evaluated wave-uniformly, and every lane updates its taps;
//   * the capture is staged chunk-wise (<= 512 steps) into a double-buffered LDS window with coalesced loads - every
//     HBM byte is read once per sweep - and each step reads its samples from LDS one unrolled group ahead; the error
//     trace is collected 64 steps at a time in registers and written with one coalesced store (no per-step branch);
//   * decision-directed methods search the alphabet lane-parallel (lane j <-> symbol j) and pick the FIRST minimum
//     with ballot + ff1, which is exactly the strict-`<` scan of det_symbol; RDE/MRDE partition look-ups are a ballot
//     over lane-held partitions;
//   * with adaptive step size the modes run back to back in one wave because the reference carries `mu` from one
//     mode into the next when run sequentially; otherwise each mode gets its own workgroup (its own CU).
#pragma once
#include "common.h"
#include "train_bi.h"
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <stdio.h>
#include <string.h>

namespace qh {

constexpr int MAX_TABLE = 64;  // alphabet / partition entries held one per lane; larger tables take the serial path

template <typename R> struct TrainArgs {
    const Cx<R> *E;
    Cx<R> *wx;
    const Cx<R> *symbols;
    Cx<R> *err;
    R *mu;
    int64_t L, TrSyms, nsy;
    int nmodes, ntaps, Niter, os, nsel, adaptive, method;
    int64_t modes[16];
    // segments of one sweep trained concurrently (parallel-in-time training, train_pit.h): blockIdx.y = segment; segment s
    // covers seg_len steps (one 64-step block more for the first seg_extra segments, seg_tail more for the last one) of
    // sweep seg_iter, starts from ITS taps wx + s * nmodes * nmodes * ntaps and leaves its end taps there
    int64_t seg_begin, seg_len, seg_extra, seg_tail;     // seg_begin: first step of segment 0
    int nseg, seg_iter;
    const int *skip;        // optional device flag: non-zero -> the launch does nothing
    Cx<R> *wx_out;          // window batches: (nwin, nmodes, nmodes, ntaps) result taps
    // batch of independent windows (frame synchronisation, qampy/core/pilotbased_receiver.py:395-400): blockIdx.y = window;
    // window v trains on E[:, win_start[v] : win_start[v] + win_len] from the shared initial taps `wx` and step size `mu`
    // and writes its own taps / error trace / final step size
    const int64_t *win_start;
    int64_t win_len;
    int nwin;
    R *win_mu;
    int64_t e_off;          // sample offset of the chain's view into E (0 except for windows)
};

// ------------------------------------------------------------------------------------------------ error functions
template <typename R> struct Tables {
    // lane-resident copies of symbols[mode, :]  (lane j holds entry j and, for the split tables, entry ncode + j)
    R a_re, a_im;   // alphabet / codebook entry of this lane
    R b_re[3], b_im[3];   // decision alphabets of 65 .. 256 symbols: entries lane + 64, lane + 128, lane + 192 (`wide`)
    bool wide;
    R p_re, p_im;   // partition entry of this lane (RDE / MRDE)
    int n, ncode, npart;
    bool serial;        // tables larger than a wave: wave-uniform serial scans over `glob`
    const Cx<R> *glob;  // symbols[mode, :] in global memory (serial fallback, data-aided look-up)
};

// first index with signal > partition failing == number of leading partitions below `signal`
template <typename R> __device__ __forceinline__ int partition_index(R signal, R part, int npart, int lane)
{
    unsigned long long m = __ballot(lane < npart && signal > part);
    unsigned long long stop = ~m;
    return stop ? __builtin_ctzll(stop) : 64;
}

template <typename R> __device__ __forceinline__ Cx<R> nearest_symbol(Cx<R> X, const Tables<R> &T, int lane)
{
    if (!T.serial && !T.wide) {
        R dr = X.re - T.a_re, di = X.im - T.a_im;
        R d = fma_(dr, dr, di * di);
        if (lane >= T.n) d = (R)3.0e38;
        R dmin = wave_min(d);
        if (!(dmin < (R)1000.)) return Cx<R>{(R)1, (R)0};           // det_symbol's initial value survives (:258-259)
        unsigned long long m = __ballot(d == dmin);
        int j = __builtin_ctzll(m);                                 // first minimum == strict `<` scan order
        return Cx<R>{readlane(T.a_re, j), readlane(T.a_im, j)};
    }
    if (T.wide) {
        // 65 .. 256 symbols, four per lane in registers: smallest distance of the wave, then the FIRST index that has it - entries are
        // striped (lane + 64 u), so the first stripe with a hit holds it and its lowest lane is it (det_symbol's strict `<` scan, :258-264)
        R d[4];
        {
            const R dr = X.re - T.a_re, di = X.im - T.a_im;
            d[0] = fma_(dr, dr, di * di);                             // (lane < 64 <= n)
        }
#pragma unroll
        for (int u = 1; u < 4; u++) {
            const R dr = X.re - T.b_re[u - 1], di = X.im - T.b_im[u - 1];
            d[u] = lane + 64 * u < T.n ? fma_(dr, dr, di * di) : (R)3.0e38;
        }
        const R dmin = wave_min(min_(min_(d[0], d[1]), min_(d[2], d[3])));
        if (!(dmin < (R)1000.)) return Cx<R>{(R)1, (R)0};
        unsigned long long m = __ballot(d[0] == dmin);
        if (m) { const int j = __builtin_ctzll(m); return Cx<R>{readlane(T.a_re, j), readlane(T.a_im, j)}; }
#pragma unroll
        for (int u = 1; u < 3; u++) {
            m = __ballot(d[u] == dmin);
            if (m) { const int j = __builtin_ctzll(m); return Cx<R>{readlane(T.b_re[u - 1], j), readlane(T.b_im[u - 1], j)}; }
        }
        m = __ballot(d[3] == dmin);
        const int j = __builtin_ctzll(m);
        return Cx<R>{readlane(T.b_re[2], j), readlane(T.b_im[2], j)};
    }
    // serial scan for alphabets larger than four waves (wave-uniform, every lane does the same work)
    R d0 = (R)1000.;
    Cx<R> s{(R)1, (R)0};
    for (int j = 0; j < T.n; j++) {
        Cx<R> c = ldg(T.glob + j);
        R dr = X.re - c.re, di = X.im - c.im;
        R d = fma_(dr, dr, di * di);
        if (d < d0) { d0 = d; s = c; }
    }
    return s;
}

template <typename R, int METHOD>
__device__ __forceinline__ Cx<R> error_fn(Cx<R> X, const Tables<R> &T, R R_re, R R_im, Cx<R> data_sym, int lane)
{
    Cx<R> e;
    if constexpr (METHOD == QH_M_CMA || METHOD == QH_M_SGNCMA) {      // :178-180 (sgncma -> cma_error, :133-134)
        R d = R_re - fma_(X.re, X.re, X.im * X.im);
        e.re = d * X.re; e.im = d * X.im;
    } else if constexpr (METHOD == QH_M_CMA2) {                       // :182-184, complex X**2
        R x2r = fma_(X.re, X.re, -(X.im * X.im)), x2i = (R)2 * X.re * X.im;
        R dr = R_re - x2r, di = R_im - x2i;
        e.re = fma_(dr, X.re, -(di * X.im)); e.im = fma_(dr, X.im, di * X.re);
    } else if constexpr (METHOD == QH_M_MCMA) {                       // :190-194
        e.re = (R_re - X.re * X.re) * X.re;
        e.im = (R_im - X.im * X.im) * X.im;
    } else if constexpr (METHOD == QH_M_RDE) {                        // :196-200
        R sq = fma_(X.re, X.re, X.im * X.im);
        R r;
        if (!T.serial) {
            int j = partition_index(sq, T.p_re, T.npart, lane);
            r = readlane(T.a_re, j);
        } else {
            int j = 0;
            while (j < T.npart && sq > T.glob[T.ncode + j].re) j++;
            r = T.glob[j].re;
        }
        R d = r - sq;
        e.re = X.re * d; e.im = X.im * d;
    } else if constexpr (METHOD == QH_M_MRDE) {                       // :203-211
        R sqr = X.re * X.re, sqi = X.im * X.im;
        R rr, ri;
        if (!T.serial) {
            int jr = partition_index(sqr, T.p_re, T.npart, lane);
            int ji = partition_index(sqi, T.p_im, T.npart, lane);
            rr = readlane(T.a_re, jr);
            ri = readlane(T.a_im, ji);
        } else {
            int jr = 0, ji = 0;
            while (jr < T.npart && sqr > T.glob[T.ncode + jr].re) jr++;
            while (ji < T.npart && sqi > T.glob[T.ncode + ji].im) ji++;
            rr = T.glob[jr].re; ri = T.glob[ji].im;
        }
        e.re = (rr - sqr) * X.re; e.im = (ri - sqi) * X.im;
    } else if constexpr (METHOD == QH_M_SBD) {                        // :214-217
        Cx<R> s = nearest_symbol(X, T, lane);
        e.re = (s.re - X.re) * abs_(s.re); e.im = (s.im - X.im) * abs_(s.im);
    } else if constexpr (METHOD == QH_M_SBD_DATA) {                   // :219-223
        e.re = (data_sym.re - X.re) * abs_(data_sym.re); e.im = (data_sym.im - X.im) * abs_(data_sym.im);
    } else if constexpr (METHOD == QH_M_MDDMA) {                      // :225-228
        Cx<R> s = nearest_symbol(X, T, lane);
        e.re = (s.re * s.re - X.re * X.re) * X.re; e.im = (s.im * s.im - X.im * X.im) * X.im;
    } else {                                                          // QH_M_DD :230-232
        Cx<R> s = nearest_symbol(X, T, lane);
        e.re = s.re - X.re; e.im = s.im - X.im;
    }
    return e;
}

// ------------------------------------------------------------------------------------------------ one chain sweep
// Staging: the capture is walked in chunks of CH steps.  The (CH-1)*os + ntaps samples of every input mode that a chunk
// needs are copied ONCE from HBM into LDS (coalesced, double buffered), then every step reads its taps' samples from
// LDS with immediate offsets; reads for group g+1 are issued while group g computes.
constexpr int TR_U = 8;            // steps per unrolled group
constexpr int TR_SLACK = 2 * TR_U; // extra steps' worth of LDS behind a chunk so that the look-ahead reads stay in bounds

template <typename R> struct ChainLds {
    Cx<R> *buf;        // [2][nmodes][pitch] chunk double buffer followed by a zero pad
    int pitch;         // samples per input-mode row of one chunk buffer
    int zero_off;      // element offset of the zero pad (read by lanes that own no tap)
    int CH;            // steps per chunk
};

// wave-wide complex sum: on exit re/im hold the totals in every lane
__device__ __forceinline__ void wave_csum(float &re, float &im)
{
    int sr, si;
    // v_add_f32_dpp needs 2 wait states between a VALU write and the DPP read of the same VGPR; the partner
    // component's instruction provides one of them, an s_nop the other.
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 %2, %0, 63\n\t"
        "v_readlane_b32 %3, %1, 63\n\t"
        : "+v"(re), "+v"(im), "=s"(sr), "=s"(si));
    re = __builtin_bit_cast(float, sr);
    im = __builtin_bit_cast(float, si);
}
__device__ __forceinline__ void wave_csum(double &re, double &im) { wave_sum2(re, im); }

// Trains steps [i_begin, i_end) of sweeps [it_begin, it_end) of one output mode.  The exact (reference) semantics is
// i_begin = 0, i_end = TrSyms, all sweeps, taps read from and written back to a.wx.
template <typename R, int TPL, int METHOD>
__device__ __forceinline__ R run_chain(const TrainArgs<R> &a, const ChainLds<R> &lds, int mode, R mu, int lane,
                                       int64_t i_begin, int64_t i_end, int it_begin, int it_end, Cx<R> *w_out)
{
    const int ntot = a.nmodes * a.ntaps;
    const int64_t L = a.L;
    const int os = a.os;
    Cx<R> w[TPL];
    int xoff[TPL];          // element offset of this lane's sample inside a chunk buffer at step 0 of the chunk
    int xstep[TPL];         // per-step advance in elements (0 for lanes that own no tap -> they keep reading the zero pad)
    const Cx<R> *wrow = a.wx + (size_t)mode * ntot;
#pragma unroll
    for (int s = 0; s < TPL; s++) {
        const int f = lane + 64 * s;
        const bool valid = f < ntot;
        const int fc = valid ? f : 0;
        const int k = fc / a.ntaps, t = fc - k * a.ntaps;
        xoff[s] = valid ? k * lds.pitch + t : -1;        // -1: this lane owns no tap in slot s and reads the zero pad
        xstep[s] = valid ? os : 0;
        w[s] = valid ? ldg(wrow + fc) : Cx<R>{0, 0};
    }
    // per-mode constants and tables
    const Cx<R> *sy = a.symbols + (size_t)mode * a.nsy;
    Tables<R> T;
    T.glob = sy;
    T.n = (int)a.nsy;
    T.ncode = (T.n + 1) / 2;           // np.array_split(symbs, 2): the first half takes the extra element
    T.npart = T.n - T.ncode;
    T.a_re = T.a_im = T.p_re = T.p_im = 0;
    for (int u = 0; u < 3; u++) T.b_re[u] = T.b_im[u] = 0;
    T.serial = false; T.wide = false;
    R R_re = 0, R_im = 0;
    if constexpr (METHOD == QH_M_CMA || METHOD == QH_M_SGNCMA || METHOD == QH_M_CMA2 || METHOD == QH_M_MCMA) {
        Cx<R> c = ldg(sy);
        R_re = c.re; R_im = c.im;
    } else if constexpr (METHOD == QH_M_RDE || METHOD == QH_M_MRDE) {
        if (lane < T.ncode && lane < MAX_TABLE) { Cx<R> c = ldg(sy + lane); T.a_re = c.re; T.a_im = c.im; }
        if (lane < T.npart && lane < MAX_TABLE) { Cx<R> c = ldg(sy + T.ncode + lane); T.p_re = c.re; T.p_im = c.im; }
        T.serial = T.ncode > MAX_TABLE;
    } else if constexpr (METHOD == QH_M_SBD || METHOD == QH_M_MDDMA || METHOD == QH_M_DD) {
        if (lane < T.n && lane < MAX_TABLE) { Cx<R> c = ldg(sy + lane); T.a_re = c.re; T.a_im = c.im; }
        T.wide = T.n > MAX_TABLE && T.n <= 4 * MAX_TABLE;
        T.serial = T.n > 4 * MAX_TABLE;
        if (T.wide) {
#pragma unroll
            for (int u = 1; u < 4; u++)
                if (lane + 64 * u < T.n) { Cx<R> c = ldg(sy + lane + 64 * u); T.b_re[u - 1] = c.re; T.b_im[u - 1] = c.im; }
        }
    }

    Cx<R> *errow = a.err + (size_t)mode * (a.TrSyms * a.Niter);
    const int64_t TrSyms = a.TrSyms;
    const int CH = lds.CH;
    const int bufsz = a.nmodes * lds.pitch;              // elements per chunk buffer
    const int64_t nsteps = i_end - i_begin;
    const int nchunk = (int)((nsteps + CH - 1) / CH);
    const int span_full = (CH - 1) * os + a.ntaps;

    // copy the samples of chunk c (steps i_begin + c*CH ...) into buffer (c & 1)
    auto stage = [&](int c) {
        const int64_t s0 = a.e_off + (i_begin + (int64_t)c * CH) * os;
        int64_t span = L - s0;
        if (span > span_full + TR_SLACK * os) span = span_full + TR_SLACK * os;
        Cx<R> *dst = lds.buf + (c & 1) * bufsz;
        for (int k = 0; k < a.nmodes; k++) {
            const Cx<R> *src = a.E + (size_t)k * L + s0;
            for (int j = lane; j < (int)span; j += 64) dst[k * lds.pitch + j] = ldg(src + j);
        }
    };

    R ebr = 0, ebi = 0;          // error-trace staging: lane (i & 63) keeps the error of step i until a 64-step flush
    Cx<R> e_prev{0, 0};

    // one LMS step on samples x[]; i = step index inside the sweep
    auto step = [&](const Cx<R> (&x)[TPL], int64_t i, Cx<R> dsym) {
        R pr = 0, pi = 0;
#pragma unroll
        for (int s = 0; s < TPL; s++) {                                  // Xest = sum x*w, no conjugate (:24-31)
            pr = fma_(x[s].re, w[s].re, pr); pr = fma_(-x[s].im, w[s].im, pr);
            pi = fma_(x[s].re, w[s].im, pi); pi = fma_(x[s].im, w[s].re, pi);
        }
        wave_csum(pr, pi);
        const Cx<R> X{pr, pi};
        const Cx<R> e = error_fn<R, METHOD>(X, T, R_re, R_im, dsym, lane);
        const bool mine = lane == (int)((i - i_begin) & 63);
        ebr = mine ? e.re : ebr;
        ebi = mine ? e.im : ebi;
        const R cr = mu * e.re, ci = mu * e.im;                           // w += (mu*e)*conj(x) (:170)
#pragma unroll
        for (int s = 0; s < TPL; s++) {
            w[s].re = fma_(cr, x[s].re, fma_(ci, x[s].im, w[s].re));
            w[s].im = fma_(ci, x[s].re, fma_(-cr, x[s].im, w[s].im));
        }
        if (a.adaptive) {                                                 // adapt_step(mu, err[i], err[i-1]) :12-16, :171-172
            const bool keep = (i == i_begin) || ((e_prev.re * e.re > 0) && (e_prev.im * e.im > 0));
            const R den = fma_(mu, fma_(e_prev.re, e_prev.re, e_prev.im * e_prev.im), (R)1);
            mu = keep ? mu : mu / den;
            e_prev = e;
        }
    };

    for (int it = it_begin; it < it_end; it++) {
        Cx<R> *eout = errow + (size_t)it * TrSyms;
        stage(0);
        for (int c = 0; c < nchunk; c++) {
            if (c + 1 < nchunk) stage(c + 1);                             // next chunk lands while this one computes
            const Cx<R> *cb = lds.buf + (c & 1) * bufsz;
            const int64_t ibase = i_begin + (int64_t)c * CH;
            const int nst = (int)((i_end - ibase) < CH ? (i_end - ibase) : CH);
            const Cx<R> *xp[TPL];
#pragma unroll
            for (int s = 0; s < TPL; s++) xp[s] = xoff[s] >= 0 ? cb + xoff[s] : lds.buf + lds.zero_off;
            Cx<R> xa[TR_U][TPL], xb[TR_U][TPL];
            Cx<R> dq[TR_U];
#pragma unroll
            for (int u = 0; u < TR_U; u++) dq[u] = Cx<R>{0, 0};
            auto load_group = [&](Cx<R> (&x)[TR_U][TPL], int g) {         // samples of steps g .. g+U-1 of this chunk
#pragma unroll
                for (int u = 0; u < TR_U; u++)
#pragma unroll
                    for (int s = 0; s < TPL; s++) x[u][s] = xp[s][(g + u) * xstep[s]];
            };
            int g = 0;
            load_group(xa, 0);
            for (; g + 2 * TR_U <= nst; g += 2 * TR_U) {
                load_group(xb, g + TR_U);
#pragma unroll
                for (int u = 0; u < TR_U; u++) {
                    if constexpr (METHOD == QH_M_SBD_DATA) dq[u] = ldg(sy + ibase + g + u);
                    step(xa[u], ibase + g + u, dq[u]);
                }
                load_group(xa, g + 2 * TR_U);                             // may run past nst: stays inside the slack
#pragma unroll
                for (int u = 0; u < TR_U; u++) {
                    if constexpr (METHOD == QH_M_SBD_DATA) dq[u] = ldg(sy + ibase + g + TR_U + u);
                    step(xb[u], ibase + g + TR_U + u, dq[u]);
                }
                if (((ibase - i_begin + g + 2 * TR_U) & 63) == 0) {       // 64 errors staged -> one coalesced store
                    stg(eout + ibase + g + 2 * TR_U - 64 + lane, Cx<R>{ebr, ebi});
                }
            }
            for (; g < nst; g++) {                                        // tail of the chunk, one step at a time
                Cx<R> x1[TPL];
#pragma unroll
                for (int s = 0; s < TPL; s++) x1[s] = xp[s][g * xstep[s]];
                Cx<R> d1{0, 0};
                if constexpr (METHOD == QH_M_SBD_DATA) d1 = ldg(sy + ibase + g);
                step(x1, ibase + g, d1);
                if (((ibase - i_begin + g + 1) & 63) == 0) stg(eout + ibase + g + 1 - 64 + lane, Cx<R>{ebr, ebi});
            }
        }
        const int rem = (int)(nsteps & 63);                               // errors still staged at the end of the sweep
        if (lane < rem) stg(eout + (i_end - rem) + lane, Cx<R>{ebr, ebi});
    }
    if (w_out) {
        Cx<R> *wo = w_out + (size_t)mode * ntot;
#pragma unroll
        for (int s = 0; s < TPL; s++)
            if (lane + 64 * s < ntot) stg(wo + lane + 64 * s, w[s]);
    }
    return mu;
}

template <typename R, int TPL, int METHOD>
__global__ void __launch_bounds__(64) train_kernel(TrainArgs<R> a, int CH, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    ChainLds<R> lds;
    lds.buf = reinterpret_cast<Cx<R> *>(smem);
    lds.pitch = pitch;
    lds.CH = CH;
    lds.zero_off = 2 * a.nmodes * pitch;
    if (a.skip && *a.skip) return;
    for (int j = lane; j < TR_SLACK + 64; j += 64) lds.buf[lds.zero_off + j] = Cx<R>{0, 0};
    R mu = *a.mu;
    // adaptive: sequential semantics, ONE wave walks the modes in order and carries mu;
    // otherwise one workgroup (= one CU) per selected mode
    if (a.nwin > 0) {
        // batch of independent windows: one wave per window walks the selected modes (sequential semantics, mu carried)
        TrainArgs<R> b = a;
        const int v = blockIdx.y;
        b.e_off = a.win_start[v];
        b.L = a.L;                                      // row pitch; the staging clamp uses L - e_off - ...
        b.err = a.err + (size_t)v * a.nmodes * a.TrSyms * a.Niter;
        Cx<R> *wout = a.wx_out + (size_t)v * a.nmodes * a.nmodes * a.ntaps;
        const int jb = a.adaptive ? 0 : blockIdx.x, je = a.adaptive ? a.nsel : blockIdx.x + 1;
        for (int j = jb; j < je; j++)
            mu = run_chain<R, TPL, METHOD>(b, lds, (int)a.modes[j], mu, lane, 0, a.TrSyms, 0, a.Niter, wout);
        if (lane == 0 && (a.adaptive || blockIdx.x == 0)) a.win_mu[v] = mu;
        return;
    }
    if (a.nseg > 0) {
        // one workgroup per (mode, segment); fixed step size
        const int64_t sgi = blockIdx.y;
        const int64_t nx = sgi < a.seg_extra ? sgi : a.seg_extra;
        const int64_t b = a.seg_begin + sgi * a.seg_len + nx * 64;
        const int64_t e = b + a.seg_len + (sgi < a.seg_extra ? 64 : 0) + ((int)sgi == a.nseg - 1 ? a.seg_tail : 0);
        TrainArgs<R> sa = a;
        sa.wx = a.wx + (size_t)sgi * a.nmodes * a.nmodes * a.ntaps;
        if (b < e) run_chain<R, TPL, METHOD>(sa, lds, (int)a.modes[blockIdx.x], mu, lane, b, e, a.seg_iter, a.seg_iter + 1, sa.wx);
        return;
    }
    const int jbeg = a.adaptive ? 0 : blockIdx.x, jend = a.adaptive ? a.nsel : blockIdx.x + 1;
    for (int j = jbeg; j < jend; j++)
        mu = run_chain<R, TPL, METHOD>(a, lds, (int)a.modes[j], mu, lane, 0, a.TrSyms, 0, a.Niter, a.wx);
    if (a.adaptive && lane == 0) *a.mu = mu;
}

template <typename R, int TPL> static int launch_tpl(const TrainArgs<R> &a)
{
    dim3 grid(a.nseg > 0 ? a.nsel : (a.adaptive ? 1 : a.nsel), a.nwin > 0 ? a.nwin : (a.nseg > 0 ? a.nseg : 1)), block(64);
    // chunk length: as many steps as fit a 48 KiB double buffer (at most 512)
    int CH = 512;
    int pitch = 0;
    size_t lds = 0;
    for (;; CH /= 2) {
        pitch = (CH - 1 + TR_SLACK) * a.os + a.ntaps;
        pitch = (pitch + 1) & ~1;
        lds = ((size_t)2 * a.nmodes * pitch + TR_SLACK + 64) * sizeof(Cx<R>);
        if (lds <= 48 * 1024 || CH <= 2 * TR_U) break;
    }
    QH_REQUIRE(lds <= 64 * 1024, "train_equaliser: nmodes*ntaps*os too large for the LDS sample window");
#define QH_CASE(M) case M: hipLaunchKernelGGL((train_kernel<R, TPL, M>), grid, block, lds, g_stream, a, CH, pitch); break;
    switch (a.method) {
        QH_CASE(QH_M_CMA) QH_CASE(QH_M_CMA2) QH_CASE(QH_M_SGNCMA) QH_CASE(QH_M_MCMA) QH_CASE(QH_M_RDE) QH_CASE(QH_M_MRDE)
        QH_CASE(QH_M_SBD) QH_CASE(QH_M_MDDMA) QH_CASE(QH_M_DD) QH_CASE(QH_M_SBD_DATA)
    default: return QH_ERR_METHOD;
    }
#undef QH_CASE
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R> static int launch_any(const TrainArgs<R> &a)
{
    const int ntot = a.nmodes * a.ntaps;
    if (ntot <= 64) return launch_tpl<R, 1>(a);
    if (ntot <= 128) return launch_tpl<R, 2>(a);
    if (ntot <= 256) return launch_tpl<R, 4>(a);
    return launch_tpl<R, 16>(a);
}

// dst[i] = src[i / n] (one step size per channel -> one per channel and mode) and back (the last mode's)
template <typename R> __global__ void spread_kernel(R *dst, const R *src, int n, int total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] = src[i / n];
}
// dst[i] = src[i mod n]: one set of start taps -> one per window / channel
template <typename T> __global__ void tile_kernel(T *dst, const T *src, unsigned n, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] = src[i % n];
}
template <typename R> __global__ void gather_kernel(R *dst, const R *src, int n, int nch)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nch) dst[c] = src[(size_t)c * n + n - 1];
}

// scratch the Gram tables of one call may take: qh_set_gram_budget_gb (default 160 of the 288 GB); longer captures / larger channel banks
// are trained in time chunks
static size_t gram_budget()
{
    const double gb = gram_budget_gb();               // (qh_set_gram_budget_gb; no environment look-up in the launch path)
    return (size_t)((gb > 0.001 ? gb : 0.001) * 1073741824.0);
}

// The reference's exact sequential semantics, in whichever of the three kernel forms is fastest for the call.
template <typename R>
int train_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, R *mu_dev, void *wx, int ntaps,
              const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy, int method, void *err,
              int zero_err, const void *gram = nullptr, int nch = 1, int64_t chan_stride = 0, int64_t row_pitch = 0)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    // chan_stride / row_pitch (elements; 0: contiguous captures): "channels" that are equally spaced, overlapping windows of
    // ONE capture whose rows are row_pitch apart (the window batch of frame_sync) - block forms only
    const int64_t Lp = row_pitch ? row_pitch : L, Ecs = chan_stride ? chan_stride : (int64_t)nmodes * L;
    // nch > 1: a bank of independent captures with identical shapes, arrays (nch, ...) contiguous, one mu per channel; the
    // look-ahead / block-iterative kernels take the channel as blockIdx.y, anything else runs channel after channel
    QH_REQUIRE(nch >= 1 && nch <= 65535, "train_equaliser: bad channel count");
    // adaptive: 0 fixed step; 1 the reference's sequential semantics (mu carried from sweep to sweep AND from mode to mode,
    // pythran_equalisation.py:162-172 run with one thread); 2 one step size PER MODE - exactly what one call per selected mode
    // from the initial mu gives (mu out = the last mode's).  The compiled reference adapts a mu that its OpenMP threads share
    // without synchronisation; 2 is the deterministic stand-in for that (every mode starts adapting from the full step).
    QH_REQUIRE(adaptive >= 0 && adaptive <= 2, "train_equaliser: adaptive must be 0, 1 or 2");
    QH_REQUIRE(nmodes >= 1 && ntaps >= 1 && os >= 1 && Niter >= 0 && TrSyms >= 0, "train_equaliser: bad sizes");
    QH_REQUIRE(nsel >= 1 && nsel <= 16, "train_equaliser: between 1 and 16 modes can be selected");
    QH_REQUIRE(TrSyms == 0 || (TrSyms - 1) * os + ntaps <= L, "train_equaliser: field shorter than TrSyms*os + ntaps");
    QH_REQUIRE(nsy >= 1, "train_equaliser: empty symbols array");
    QH_REQUIRE(method != QH_M_SBD_DATA || nsy >= TrSyms, "train_equaliser: sbd_data needs >= TrSyms training symbols");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "train_equaliser: mode number >= nmodes");
    const int ntot = nmodes * ntaps;
    QH_REQUIRE(ntot <= 64 * 16, "train_equaliser: more than 1024 taps per output mode are not supported");
    if (zero_err) QH_HIP(hipMemsetAsync(err, 0, (size_t)nch * nmodes * TrSyms * Niter * sizeof(Cx<R>), g_stream));
    if (TrSyms == 0 || Niter == 0) return QH_OK;
    TrainArgs<R> a;
    a.E = (const Cx<R> *)E; a.wx = (Cx<R> *)wx; a.symbols = (const Cx<R> *)symbols; a.err = (Cx<R> *)err; a.mu = mu_dev;
    a.L = L; a.TrSyms = TrSyms; a.nsy = nsy; a.nmodes = nmodes; a.ntaps = ntaps; a.Niter = Niter; a.os = os;
    a.nsel = nsel; a.adaptive = adaptive ? 1 : 0; a.method = method;
    for (int j = 0; j < 16; j++) a.modes[j] = j < nsel ? modes[j] : 0;
    const bool per_mode = adaptive == 2 && nsel > 1;
    a.nseg = 0; a.seg_begin = 0; a.seg_len = 0; a.seg_extra = 0; a.seg_tail = 0; a.seg_iter = 0; a.skip = nullptr; a.wx_out = nullptr;
    a.win_start = nullptr; a.win_len = 0; a.nwin = 0; a.win_mu = nullptr; a.e_off = 0;
    {
        // exact semantics.  Blind methods with a fixed step run in the look-ahead (train_la.h) or block-iterative
        // (train_bi.h) form, everything else (decision-directed, data-aided, adaptive step, tiny captures) in the direct
        // form below.  Same results up to the order of additions.
        // Trainer choice.  qh_set_trainer / qh_set_form("trainer", "direct" | "lookahead" | "iterative") forces one form (A/B measurements, tests);
        // otherwise the block-iterative form takes the partitioned error functions (rde, mrde: one evaluation per sweep
        // instead of one per step), the look-ahead chain the cheap ones (cma, mcma, cma2), whichever of the two fits.
        const char *force = trainer_force();
        const bool direct = force[0] == 'd';
        bool bi_ok = !direct && !(force[0] == 'l') && bi_supported(method, adaptive, nmodes, ntaps, os, TrSyms, nsy, sizeof(Cx<R>));
        const bool decision = method == QH_M_SBD || method == QH_M_MDDMA || method == QH_M_DD;
        void *dd_table = nullptr;
        int dd_npart = -1;
        bool dd_general = false;
        if (bi_ok && decision) {     // square alphabets: per-axis slicer tables in the rde / mrde layout; any other (32- / 128-QAM crosses): scan of the alphabet
            if ((rc = slicer_tables<R>(symbols, nmodes, nsy, modes, nsel, &dd_table, &dd_npart))) return rc;
            if (!(dd_npart == 1 || dd_npart == 3 || dd_npart == 7 || dd_npart == 15)) bi_ok = dd_general = bi_general_ok(nmodes, ntaps, os, nsy, sizeof(Cx<R>));
        }
        const bool la_ok = !direct && la_supported(method, adaptive, nmodes, ntaps, os, TrSyms, nsy);
        const bool partitioned = method == QH_M_RDE || method == QH_M_MRDE;
        const bool pair = la_shape_ok(nmodes, ntaps, os);          // layout of the Gram terms of this capture (qh_gram_build_*)
        // (round 5: the adaptive step and the data-aided error run on the look-ahead chain too - ~2x fewer cycles per step than the block sweeps)
        const bool use_bi = bi_ok && ((partitioned && !adaptive) || decision || !la_ok || (force[0] == 'i'));
        if (use_bi || la_ok) {
            // block-iterative form (train_bi.h): 8 wavefronts per output mode solve each 64-step block by fixed-point sweeps;
            // look-ahead form (train_la.h): one chain wave + three helper waves per output mode
            const bool pair_tab = use_bi ? pair : true;                      // layout of the Gram table this call reads
            const size_t step_bytes = pair_tab ? sizeof(GramPair<R>) * LA_B : sizeof(Cx<R>) * GRAM_TRI / LA_B;     // per step and channel
            // Time chunks: when the Gram tables of the whole capture (x channels) would not fit the budget, the sweep runs
            // chunk after chunk - table of the chunk, then the trainers over it, taps handed on through HBM exactly as
            // between sweeps - which bounds the scratch memory for any capture length and channel count.  Not for the
            // adaptive step / data-aided training (their state does not live in HBM between launches) nor a caller's table.
            int64_t CH = TrSyms;
            if (!gram && !adaptive && method != QH_M_SBD_DATA) {
                const int64_t fit = (int64_t)(gram_budget() / (step_bytes * (size_t)nch)) / LA_B * LA_B;
                if (fit < TrSyms) CH = fit > 64 * LA_B ? fit : 64 * LA_B;
            }
            LaArgs<R> la;
            la.wx = a.wx; la.symbols = a.symbols; la.err = a.err; la.gpair = pair_tab ? 1 : 0; la.mu = mu_dev; la.mu_out = (R *)mu_dev;
            la.Lp = Lp; la.nsy = nsy; la.sy_pitch = nsy; la.err_pitch = TrSyms * Niter; la.nmodes = nmodes; la.ntaps = ntaps;
            la.os = os; la.nsel = nsel; la.method = method;
            la.nch = nch; la.E_cs = Ecs; la.wx_cs = (int64_t)nmodes * ntot; la.err_cs = (int64_t)nmodes * TrSyms * Niter; la.mu_cs = 1;
            la.mu_ms = 0;
            for (int j = 0; j < 16; j++) la.modes[j] = a.modes[j];
            la.dd_general = use_bi && dd_general;
            if (use_bi && decision && !dd_general) {   // the kernel reads the slicer table like an mrde table: row pitch 2*BI_DD_MAXLEV, 2*npart+1 used
                la.symbols = (const Cx<R> *)dd_table; la.nsy = 2 * dd_npart + 1; la.sy_pitch = 2 * BI_DD_MAXLEV;
            }
            la.prof = nullptr; la.seg = 0; la.seg_extra = 0; la.seg_tail = 0; la.skip = nullptr; la.niter = 1;
            if (form(FORM_LA_PROFILE)) {                         // developer aid: cycle split of workgroup 0 (qh_set_form("la_profile", "1"))
                void *pp = nullptr;
                if ((rc = scratch(5, 16 * sizeof(unsigned long long), &pp))) return rc;
                QH_HIP(hipMemsetAsync(pp, 0, 16 * sizeof(unsigned long long), g_stream));
                la.prof = (unsigned long long *)pp;
            }
            // adaptive = 1: mu is carried from sweep to sweep and from mode to mode -> one mode after the other;
            // adaptive = 2: every mode owns a step size (scratch array, seeded with mu) -> all modes concurrently
            R *mu_modes = nullptr;
            if (per_mode) {
                void *pm = nullptr;
                if ((rc = scratch(6, (size_t)nch * nsel * sizeof(R), &pm))) return rc;
                mu_modes = (R *)pm;
                hipLaunchKernelGGL((spread_kernel<R>), dim3((unsigned)((nch * nsel + 63) / 64)), dim3(64), 0, g_stream, mu_modes, (const R *)mu_dev, nsel, nch * nsel);
                la.mu = mu_modes; la.mu_out = mu_modes; la.mu_cs = nsel; la.mu_ms = 1;
            }
            void *built = nullptr;                              // Gram terms depend on the capture only: built once per chunk, not once per sweep
            int64_t built_step0 = -1, built_n = -1;
            const int nmode_runs = (adaptive && !mu_modes) ? nsel : 1;
            if (adaptive && !mu_modes) la.nsel = 1;
            for (int jm = 0; jm < nmode_runs; jm++) {
                if (adaptive && !mu_modes) la.modes[0] = a.modes[jm];
                // block-iterative form, sweep not chunked: ALL sweeps in one launch (the kernel loops over them with taps and step size on chip)
                const bool sweeps_inside = CH >= TrSyms && Niter > 1;
                for (int it = 0; it < (sweeps_inside ? 1 : Niter); it++) {             // else one launch per sweep (and chunk): taps go through HBM in between
                    la.niter = sweeps_inside ? Niter : 1;
                    for (int64_t step0 = 0; step0 < TrSyms;) {
                        int64_t n = TrSyms - step0 < CH ? TrSyms - step0 : CH;
                        if (TrSyms - (step0 + n) < 2 * LA_B) n = TrSyms - step0;          // never leave a tail the block forms cannot take
                        const Cx<R> *Ec = a.E + step0 * os;
                        void *G = const_cast<void *>(gram);
                        if (!G && built_step0 == step0 && built_n == n) {
                            G = built;                              // same chunk as the last launch (every sweep of an unchunked call): the table is still there
                        } else if (!G) {
                            rc = pair_tab ? gram_build<R>(Ec, nmodes, L - step0 * os, os, ntaps, n, &G, nch, Lp, Ecs)
                                          : gram_cur_build<R>(Ec, nmodes, L - step0 * os, os, ntaps, n, &G, nch, Lp, Ecs);
                            if (rc) return rc;
                            built = G; built_step0 = step0; built_n = n;
                        }
                        la.E = Ec; la.L = L - step0 * os; la.TrSyms = n; la.G = (const GramPair<R> *)G;
                        la.G_cs = (int64_t)((pair_tab ? gram_bytes<R>(n) : gram_cur_bytes<R>(n)) / sizeof(GramPair<R>));
                        la.err_off = (int64_t)it * TrSyms + step0;
                        if ((rc = use_bi ? launch_bi<R>(la, adaptive != 0) : launch_la<R>(la, adaptive != 0))) return rc;
                        step0 += n;
                    }
                }
            }
            if (mu_modes)        // mu out = the last selected mode's, per channel
                hipLaunchKernelGGL((gather_kernel<R>), dim3((unsigned)((nch + 63) / 64)), dim3(64), 0, g_stream, (R *)mu_dev, (const R *)mu_modes, nsel, nch);
            if (la.prof) {
                unsigned long long hp[16];
                QH_HIP(hipMemcpyAsync(hp, la.prof, sizeof(hp), hipMemcpyDeviceToHost, g_stream));
                QH_HIP(hipStreamSynchronize(g_stream));
                if (use_bi) {
                    fprintf(stderr, "[bi profile] method %d blocks %llu: sweeps/block %.2f, cycles/block sweeps %.0f update %.0f prior %.0f\n", method, hp[4],
                            (double)hp[0] / (double)hp[4], (double)hp[1] / (double)hp[4], (double)hp[2] / (double)hp[4], (double)hp[3] / (double)hp[4]);
                } else {
                    fprintf(stderr, "[la profile] method %d TrSyms %lld: chain work %llu wait %llu |", method, (long long)TrSyms, hp[0], hp[1]);
                    for (int h = 1; h <= LA_NH; h++) fprintf(stderr, " helper%d update %llu prior %llu wait %llu |", h, hp[4 * h], hp[4 * h + 1], hp[4 * h + 2]);
                    fprintf(stderr, "\n");
                }
            }
            return QH_OK;
        }
        QH_REQUIRE(!chan_stride && !row_pitch, "train_equaliser: strided channels need a block form of the trainer");
        if (per_mode) {                            // one call per mode, each from the initial step size
            void *pm = nullptr;
            if ((rc = scratch(6, (size_t)nch * sizeof(R), &pm))) return rc;
            QH_HIP(hipMemcpyAsync(pm, mu_dev, (size_t)nch * sizeof(R), hipMemcpyDeviceToDevice, g_stream));
            for (int j = 0; j < nsel; j++) {
                QH_HIP(hipMemcpyAsync(mu_dev, pm, (size_t)nch * sizeof(R), hipMemcpyDeviceToDevice, g_stream));
                if ((rc = train_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes + j, 1, 1, symbols, nsy, method, err, 0, gram, nch)))
                    return rc;
            }
            return QH_OK;
        }
        for (int c = 0; c < nch; c++) {            // direct form: one capture after the other
            TrainArgs<R> ac = a;
            ac.E = a.E + (size_t)c * nmodes * L; ac.wx = a.wx + (size_t)c * nmodes * ntot;
            ac.err = a.err + (size_t)c * nmodes * TrSyms * Niter; ac.mu = mu_dev + c;
            if ((rc = launch_any<R>(ac))) return rc;
        }
        return QH_OK;
    }
}

// variance of the error trace of every (window, mode): var[m * nwin + v] = mean |e - mean(e)|^2  (np.var of a complex row)
template <typename R>
__global__ void __launch_bounds__(256) win_var_kernel(const Cx<R> *err, int nmodes, int64_t n, double *var)
{
    __shared__ double r0[256], r1[256];
    const int v = blockIdx.x, m = blockIdx.y, nwin = gridDim.x;
    const Cx<R> *row = err + ((size_t)v * nmodes + m) * n;
    double sr = 0, si = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) { const Cx<R> e = row[i]; sr += e.re; si += e.im; }
    r0[threadIdx.x] = sr; r1[threadIdx.x] = si;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; } __syncthreads(); }
    const double mr = r0[0] / (double)n, mi = r1[0] / (double)n;
    __syncthreads();
    double q = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) { const Cx<R> e = row[i]; const double dr = e.re - mr, di = e.im - mi; q += dr * dr + di * di; }
    r0[threadIdx.x] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) r0[threadIdx.x] += r0[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) var[(size_t)m * nwin + v] = r0[0] / (double)n;
}
// first minimum of every mode's row of `var`; its window's tap set goes to wx_best[m]
template <typename R>
__global__ void __launch_bounds__(64) win_best_kernel(const double *var, int nwin, const Cx<R> *wx_all, int wset, int *best, Cx<R> *wx_best)
{
    const int m = blockIdx.x;
    __shared__ int bi;
    if (threadIdx.x == 0) {
        int b = 0; double bv = var[(size_t)m * nwin];
        for (int v = 1; v < nwin; v++) { const double x = var[(size_t)m * nwin + v]; if (x < bv) { bv = x; b = v; } }
        bi = b; best[m] = b;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < wset; f += 64) wx_best[(size_t)m * wset + f] = wx_all[(size_t)bi * wset + f];
}

// Batch of independent equaliser runs on windows of one capture (host pointers).  Equivalent to calling train_host once per
// window with E[:, start : start + win_len], the same initial taps and step size; all windows run concurrently.
//   wx0 (nmodes, nmodes, ntaps) in;  wx_out (nwin, nmodes, nmodes, ntaps), err (nwin, nmodes, TrSyms*Niter), mu_out (nwin) out
// Search form (var_out != nullptr; wx_out / err / mu_out may then be nullptr): the error traces stay in HBM, only their
// variances var_out (nmodes, nwin), the window with the smallest variance per mode best (nmodes) and the tap sets of those
// windows wx_best (nmodes, nmodes, nmodes, ntaps) come back - what the frame synchronisation needs (pilotbased_receiver.py:395-405).
// can the look-ahead / block-iterative kernels take this call (without slicer tables)?  (the window batch asks before it
// hands its windows over as strided channels)
template <typename R> static bool block_forms_ok(int method, int adaptive, int nmodes, int ntaps, int os, int64_t TrSyms, int64_t nsy)
{
    if (trainer_force()[0] == 'd') return false;
    if (method == QH_M_SBD || method == QH_M_MDDMA || method == QH_M_DD || method == QH_M_SBD_DATA) return false;
    return la_supported(method, adaptive, nmodes, ntaps, os, TrSyms, nsy) || bi_supported(method, adaptive, nmodes, ntaps, os, TrSyms, nsy, sizeof(Cx<R>));
}

template <typename R>
int train_windows_host(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len, int64_t TrSyms,
                       int Niter, int os, R mu, const void *wx0, int ntaps, const int64_t *modes, int nsel, int adaptive,
                       const void *symbols, int64_t nsy, int method, void *wx_out, void *err, R *mu_out,
                       double *var_out = nullptr, int *best = nullptr, void *wx_best = nullptr)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(nmodes >= 1 && L >= 1 && ntaps >= 1 && nsy >= 1 && TrSyms >= 0 && Niter >= 0 && os >= 1, "train_equaliser_windows: bad sizes");
    QH_REQUIRE(nwin >= 1 && nwin <= 65535, "train_equaliser_windows: between 1 and 65535 windows");
    QH_REQUIRE(nsel >= 1 && nsel <= 16, "train_equaliser_windows: between 1 and 16 modes can be selected");
    QH_REQUIRE(TrSyms == 0 || (TrSyms - 1) * os + ntaps <= win_len, "train_equaliser_windows: window shorter than TrSyms*os + ntaps");
    QH_REQUIRE(method != QH_M_SBD_DATA, "train_equaliser_windows: data-aided training is not supported on window batches");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "train_equaliser_windows: mode number >= nmodes");
    for (int v = 0; v < nwin; v++) QH_REQUIRE(win_start[v] >= 0 && win_start[v] + win_len <= L, "train_equaliser_windows: window outside the field");
    const int ntot = nmodes * ntaps;
    QH_REQUIRE(ntot <= 64 * 16, "train_equaliser_windows: more than 1024 taps per output mode are not supported");
    const size_t cs = sizeof(Cx<R>);
    const size_t wsz = (size_t)nmodes * ntot, esz = (size_t)nmodes * TrSyms * Niter;
    DevBuf dE, dw, ds, dwo, de, dmu, dmo, dst;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * cs))) return rc;
    if ((rc = dw.from_host(wx0, wsz * cs))) return rc;
    if ((rc = ds.from_host(symbols, (size_t)nmodes * nsy * cs))) return rc;
    if ((rc = dmu.from_host(&mu, sizeof(R)))) return rc;
    if ((rc = dst.from_host(win_start, (size_t)nwin * sizeof(int64_t)))) return rc;
    if ((rc = dwo.alloc((size_t)nwin * wsz * cs))) return rc;
    if ((rc = de.alloc((size_t)nwin * esz * cs))) return rc;
    if ((rc = dmo.alloc((size_t)nwin * sizeof(R)))) return rc;
    QH_HIP(hipMemsetAsync(de.p, 0, de.n ? de.n : 1, g_stream));
    // every window starts from (and keeps, for unselected modes) the initial taps (one launch: hundreds of small copies were ~1 ms of the frame search)
    hipLaunchKernelGGL((tile_kernel<Cx<R>>), dim3((unsigned)(((size_t)nwin * wsz + 255) / 256)), dim3(256), 0, g_stream, (Cx<R> *)dwo.p, (const Cx<R> *)dw.p, (unsigned)wsz, (size_t)nwin * wsz);
    // equally spaced windows are the channels of a bank whose captures overlap: a FEW of them go to the latency forms of the
    // trainer (3 x fewer cycles per step on the critical path).  Hundreds of windows fill the chip either way, and then the direct
    // form - one wave per chain instead of four or eight - is the cheaper one: 260 windows x 2 modes of the config-5 frame search
    // take 3.4 ms direct, 5.6 ms block-iterative (measured), so it keeps them, as it does irregular starts and the other methods.
    // (round 5: the look-ahead chain with the adaptive step and all sweeps in one launch - 65 ns per step against the direct form's 165 - takes
    // the windows of the frame search whenever it is the form train_dev would pick and the Gram tables of all windows fit the scratch budget)
    const bool la_pick = la_supported(method, adaptive ? 1 : 0, nmodes, ntaps, os, TrSyms, nsy) && trainer_force()[0] != 'i' && trainer_force()[0] != 'd' &&
                         (adaptive || !(method == QH_M_RDE || method == QH_M_MRDE)) && nwin <= 1024 && (size_t)nwin * gram_bytes<R>(TrSyms) <= gram_budget();
    bool strided = ((int64_t)nwin * nsel <= 64 || la_pick) && (nwin == 1 || win_start[1] > win_start[0]);
    for (int v = 2; v < nwin && strided; v++) strided = win_start[v] - win_start[v - 1] == win_start[1] - win_start[0];
    if (TrSyms > 0 && Niter > 0 && strided && block_forms_ok<R>(method, adaptive ? 1 : 0, nmodes, ntaps, os, TrSyms, nsy)) {
        hipLaunchKernelGGL((spread_kernel<R>), dim3((unsigned)((nwin + 63) / 64)), dim3(64), 0, g_stream, (R *)dmo.p, (const R *)dmu.p, nwin, nwin);
        rc = train_dev<R>((const Cx<R> *)dE.p + win_start[0], nmodes, win_len, TrSyms, Niter, os, (R *)dmo.p, dwo.p, ntaps, modes, nsel, adaptive ? 1 : 0, ds.p, nsy,
                          method, de.p, 0, nullptr, nwin, nwin > 1 ? win_start[1] - win_start[0] : L, L);
        if (rc) return rc;
    } else if (TrSyms > 0 && Niter > 0) {
        TrainArgs<R> a;
        a.E = (const Cx<R> *)dE.p; a.wx = (Cx<R> *)dw.p; a.symbols = (const Cx<R> *)ds.p; a.err = (Cx<R> *)de.p; a.mu = (R *)dmu.p;
        a.L = L; a.TrSyms = TrSyms; a.nsy = nsy; a.nmodes = nmodes; a.ntaps = ntaps; a.Niter = Niter; a.os = os;
        a.nsel = nsel; a.adaptive = adaptive ? 1 : 0; a.method = method;
        for (int j = 0; j < 16; j++) a.modes[j] = j < nsel ? modes[j] : 0;
        a.nseg = 0; a.seg_begin = 0; a.seg_len = 0; a.seg_extra = 0; a.seg_tail = 0; a.seg_iter = 0; a.skip = nullptr; a.wx_out = (Cx<R> *)dwo.p;
        a.win_start = (const int64_t *)dst.p; a.win_len = win_len; a.nwin = nwin; a.win_mu = (R *)dmo.p; a.e_off = 0;
        if ((rc = launch_any<R>(a))) return rc;
    }
    if (wx_out && (rc = dwo.to_host(wx_out, dwo.n))) return rc;
    if (err && (rc = de.to_host(err, de.n))) return rc;
    if (mu_out && (rc = dmo.to_host(mu_out, dmo.n))) return rc;
    if (var_out) {
        QH_REQUIRE(best && wx_best, "train_equaliser_windows: the search form needs best and wx_best");
        DevBuf dv, db, dwb;
        if ((rc = dv.alloc((size_t)nmodes * nwin * sizeof(double)))) return rc;
        if ((rc = db.alloc((size_t)nmodes * sizeof(int)))) return rc;
        if ((rc = dwb.alloc((size_t)nmodes * wsz * cs))) return rc;
        hipLaunchKernelGGL((win_var_kernel<R>), dim3(nwin, nmodes), dim3(256), 0, g_stream, (const Cx<R> *)de.p, nmodes, (int64_t)(TrSyms * Niter), (double *)dv.p);
        hipLaunchKernelGGL((win_best_kernel<R>), dim3(nmodes), dim3(64), 0, g_stream, (const double *)dv.p, nwin, (const Cx<R> *)dwo.p, (int)wsz, (int *)db.p, (Cx<R> *)dwb.p);
        QH_HIP(hipGetLastError());
        if ((rc = dv.to_host(var_out, dv.n))) return rc;
        if ((rc = db.to_host(best, db.n))) return rc;
        if ((rc = dwb.to_host(wx_best, dwb.n))) return rc;
        QH_HIP(hipStreamSynchronize(g_stream));
        return QH_OK;
    }
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

template <typename R>
int train_host(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, R *mu, void *wx, int ntaps,
               const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy, int method, void *err)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(nmodes >= 1 && L >= 1 && ntaps >= 1 && nsy >= 1 && TrSyms >= 0 && Niter >= 0, "train_equaliser: bad sizes");
    const size_t cs = sizeof(Cx<R>);
    DevBuf dE, dw, ds, de, dmu;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * cs))) return rc;
    if ((rc = dw.from_host(wx, (size_t)nmodes * nmodes * ntaps * cs))) return rc;
    if ((rc = ds.from_host(symbols, (size_t)nmodes * nsy * cs))) return rc;
    if ((rc = dmu.from_host(mu, sizeof(R)))) return rc;
    if ((rc = de.alloc((size_t)nmodes * TrSyms * Niter * cs))) return rc;
    rc = train_dev<R>(dE.p, nmodes, L, TrSyms, Niter, os, (R *)dmu.p, dw.p, ntaps, modes, nsel, adaptive, ds.p, nsy, method,
                      de.p, 1);
    if (rc) return rc;
    if ((rc = dw.to_host(wx, dw.n))) return rc;
    if ((rc = de.to_host(err, de.n))) return rc;
    if ((rc = dmu.to_host(mu, sizeof(R)))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

// ================================================================================================ real-valued trainer
template <typename R> struct TrainRealArgs {
    const R *E;
    R *wx;
    const R *symbols;
    R *err;
    R *mu;
    int64_t L, TrSyms, nsy;
    int nmodes, ntaps, Niter, os, nsel, adaptive, method;
    int64_t modes[32];
};

template <typename R, int TPL, int METHOD>
__global__ void __launch_bounds__(64) train_real_kernel(TrainRealArgs<R> a)
{
    // The real-valued reference has no per-mode parallel/serial distinction worth exploiting here: mu is shared and
    // carried exactly like in the complex trainer (pythran_equalisation.py:97-107), so one wave walks all modes.
    const int lane = threadIdx.x;
    const int ntot = a.nmodes * a.ntaps;
    R mu = *a.mu;
    const int jbeg = a.adaptive ? 0 : blockIdx.x, jend = a.adaptive ? a.nsel : blockIdx.x + 1;
    for (int jm = jbeg; jm < jend; jm++) {
        const int mode = (int)a.modes[jm];
        R w[TPL];
        const R *xbase[TPL];
        bool valid[TPL];
        R *wrow = a.wx + (size_t)mode * ntot;
#pragma unroll
        for (int s = 0; s < TPL; s++) {
            int f = lane + 64 * s;
            valid[s] = f < ntot;
            int fc = valid[s] ? f : 0;
            int k = fc / a.ntaps, t = fc - k * a.ntaps;
            xbase[s] = a.E + (size_t)k * a.L + t;
            w[s] = valid[s] ? wrow[fc] : (R)0;
        }
        const R *sy = a.symbols + (size_t)mode * a.nsy;
        const R R0 = sy[0];
        const int n = (int)a.nsy;
        R al = (lane < n) ? sy[lane] : (R)0;          // alphabet entry of this lane (dd)
        R *errow = a.err + (size_t)mode * (a.TrSyms * a.Niter);
        for (int it = 0; it < a.Niter; it++) {
            R e_prev = 0;
            for (int64_t i = 0; i < a.TrSyms; i++) {
                R x[TPL];
                R p = 0;
#pragma unroll
                for (int s = 0; s < TPL; s++) {
                    x[s] = valid[s] ? xbase[s][i * a.os] : (R)0;
                    p = fma_(x[s], w[s], p);
                }
                R X = wave_sum(p);
                R e;
                if constexpr (METHOD == QH_RM_CMA) {                     // :110-112
                    e = (R0 - X * X) * X;
                } else if constexpr (METHOD == QH_RM_SGNCMA) {           // :114-116
                    R v = R0 - X * X;
                    R d = (R)((v > 0) - (v < 0));
                    e = d * (R)((X > 0) - (X < 0));
                } else if constexpr (METHOD == QH_RM_DD) {               // :118-120, det_symbol_argmin :233-236
                    R s;
                    if (n <= MAX_TABLE) {
                        R d = lane < n ? abs_(X - al) : (R)3.0e38;
                        R dmin = wave_min(d);
                        int j = __builtin_ctzll(__ballot(d == dmin));
                        s = readlane(al, j);
                    } else {
                        R best = abs_(X - sy[0]); s = sy[0];
                        for (int j = 1; j < n; j++) { R d = abs_(X - sy[j]); if (d < best) { best = d; s = sy[j]; } }
                    }
                    e = (s - X) * abs_(s);
                } else {                                                 // QH_RM_DD_DATA :122-125
                    R s = sy[i];
                    e = (s - X) * abs_(s);
                }
                if (lane == 0) errow[(size_t)it * a.TrSyms + i] = e;
                R c = mu * e;
#pragma unroll
                for (int s = 0; s < TPL; s++) w[s] = fma_(c, x[s], w[s]);
                if (a.adaptive && i > 0) {                               // adapt_step_real :18-22
                    bool keep = e_prev * e > 0;
                    mu = keep ? mu : mu / fma_(mu, e_prev * e_prev, (R)1);
                }
                e_prev = e;
            }
        }
#pragma unroll
        for (int s = 0; s < TPL; s++)
            if (valid[s]) wrow[lane + 64 * s] = w[s];
    }
    if (a.adaptive && lane == 0) *a.mu = mu;
}

template <typename R, int TPL> static int launch_real_tpl(const TrainRealArgs<R> &a)
{
    dim3 grid(a.adaptive ? 1 : a.nsel), block(64);
#define QH_CASE(M) case M: hipLaunchKernelGGL((train_real_kernel<R, TPL, M>), grid, block, 0, g_stream, a); break;
    switch (a.method) {
        QH_CASE(QH_RM_CMA) QH_CASE(QH_RM_SGNCMA) QH_CASE(QH_RM_DD) QH_CASE(QH_RM_DD_DATA)
    default: return QH_ERR_METHOD;
    }
#undef QH_CASE
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R>
int train_real_host(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, R *mu, void *wx, int ntaps,
                    const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy, int method, void *err)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_RM_DD_DATA) { set_error("unknown real-valued equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(nmodes >= 1 && L >= 1 && ntaps >= 1 && nsy >= 1 && TrSyms >= 0 && Niter >= 0, "train_equaliser_realvalued: bad sizes");
    QH_REQUIRE(nsel >= 1 && nsel <= 32, "train_equaliser_realvalued: between 1 and 32 modes can be selected");
    QH_REQUIRE(TrSyms == 0 || (TrSyms - 1) * os + ntaps <= L, "train_equaliser_realvalued: field shorter than TrSyms*os + ntaps");
    QH_REQUIRE(method != QH_RM_DD_DATA || nsy >= TrSyms, "train_equaliser_realvalued: dd_data needs >= TrSyms training symbols");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "train_equaliser_realvalued: mode number >= nmodes");
    const int ntot = nmodes * ntaps;
    QH_REQUIRE(ntot <= 64 * 16, "train_equaliser_realvalued: more than 1024 taps per output mode are not supported");
    DevBuf dE, dw, ds, de, dmu;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * sizeof(R)))) return rc;
    if ((rc = dw.from_host(wx, (size_t)nmodes * nmodes * ntaps * sizeof(R)))) return rc;
    if ((rc = ds.from_host(symbols, (size_t)nmodes * nsy * sizeof(R)))) return rc;
    if ((rc = dmu.from_host(mu, sizeof(R)))) return rc;
    if ((rc = de.alloc((size_t)nmodes * TrSyms * Niter * sizeof(R)))) return rc;
    QH_HIP(hipMemsetAsync(de.p, 0, de.n ? de.n : 1, g_stream));
    if (TrSyms > 0 && Niter > 0) {
        TrainRealArgs<R> a;
        a.E = (const R *)dE.p; a.wx = (R *)dw.p; a.symbols = (const R *)ds.p; a.err = (R *)de.p; a.mu = (R *)dmu.p;
        a.L = L; a.TrSyms = TrSyms; a.nsy = nsy; a.nmodes = nmodes; a.ntaps = ntaps; a.Niter = Niter; a.os = os;
        a.nsel = nsel; a.adaptive = adaptive ? 1 : 0; a.method = method;
        for (int j = 0; j < 32; j++) a.modes[j] = j < nsel ? modes[j] : 0;
        if (ntot <= 64) rc = launch_real_tpl<R, 1>(a);
        else if (ntot <= 128) rc = launch_real_tpl<R, 2>(a);
        else if (ntot <= 256) rc = launch_real_tpl<R, 4>(a);
        else rc = launch_real_tpl<R, 16>(a);
        if (rc) return rc;
    }
    if ((rc = dw.to_host(wx, dw.n))) return rc;
    if ((rc = de.to_host(err, de.n))) return rc;
    if ((rc = dmu.to_host(mu, sizeof(R)))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

}  // namespace qh

