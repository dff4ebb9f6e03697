// Block-iterative form of the exact equaliser recurrence (complex, blind methods, fixed step size).
//
// Inside a block of 64 steps the look-ahead identity (train_la.h)
//        y_i = W . x_i + sum_{j < i} c_j * G(j, i) ,     c_j = mu * errfn(y_j) ,     G(j, i) = sum_f conj(x_j[f]) x_i[f]
// is a STRICTLY LOWER-TRIANGULAR non-linear system  y = q + L c(y).  The sequential recurrence solves it by forward
// substitution - 64 dependent steps, each paying the ~8 cycles a lone wavefront needs per instruction.  Here it is
// solved by fixed-point sweeps  c <- mu errfn(q + L c)  instead: because L is strictly lower triangular, entry i is
// final once entries < i are, so the sweeps reach the exact fixed point (bit for bit, the reduction order is fixed) after
// at most 64 and in practice 4-8 of them, and every sweep is a 64x64 complex matrix-vector product that BI_W = 8 wavefronts
// share (2 per SIMD): wave w owns the steps j = 8w..8w+7 (its c_j and Gram rows) and the taps of slice w.  The error
// function is evaluated once per sweep instead of once per step.  Per block:
//        q      prior outputs W . x_i, every wave contributes its tap slice
//        sweeps partial_w[i] = q_w[i] + sum_{j in own} c_j G(j, i)  ->  LDS [i][w]  ->  barrier  ->  wave w reduces rows
//               i = 8w..8w+7 over the 8 contributions (one ds_read + an 8-lane DPP tree)  ->  c_i = mu errfn(y_i)
//        taps   W += sum_j c_j conj(x_j): every wave adds its 8 steps to all taps, slice owners reduce over the waves
// The result is the sequential recurrence's result up to the order of floating-point additions (like train_la.h).
#pragma once
#include <stdlib.h>
#include "train_la.h"

#include <atomic>
namespace qh {

constexpr int BI_W = 8;                  // wavefronts per workgroup (2 per SIMD: more only queue up behind the SIMD's issue port)
constexpr int BI_JW = LA_B / BI_W;       // steps owned by a wave
constexpr int BI_PAD = BI_W + 1;         // row pitch of the [i][w] exchange buffers (bank-conflict padding)
constexpr int BI_MAXTAPS = 128;          // taps per output mode (2 per lane in the tap-update layout)
constexpr size_t BI_LDS_MAX = 160 * 1024 - 256;   // LDS a workgroup of this kernel may ask for (the CU's 160 KiB; above 64 KiB the launcher sets the attribute)

// sum over groups of BI_W consecutive lanes; every lane of a group gets the group's total (fixed order -> deterministic)
__device__ __forceinline__ void group_csum(float &re, float &im)
{
    if constexpr (BI_W == 16) {
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
            "s_nop 1\n\t"
            : "+v"(re), "+v"(im));
    } else {
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
            "s_nop 1\n\t"
            : "+v"(re), "+v"(im));
    }
}
__device__ __forceinline__ void group_csum(double &re, double &im)
{
    re += dpp_mov<DPP_QUAD_1032>(re);       im += dpp_mov<DPP_QUAD_1032>(im);
    re += dpp_mov<DPP_QUAD_2301>(re);       im += dpp_mov<DPP_QUAD_2301>(im);
    re += dpp_mov<DPP_ROW_HALF_MIRROR>(re); im += dpp_mov<DPP_ROW_HALF_MIRROR>(im);
    if constexpr (BI_W == 16) { re += dpp_mov<DPP_ROW_MIRROR>(re); im += dpp_mov<DPP_ROW_MIRROR>(im); }
}

// acc += a * b  /  acc += a * conj(b) on 2-vectors (two v_pk_fma for float)
template <typename R> __device__ __forceinline__ void cfma(Cx<R> &acc, R ar, R ai, const Cx<R> &b)
{
    using v2 = typename V2<R>::type;
    v2 t = {acc.re, acc.im};
    t = __builtin_elementwise_fma(v2{ar, ar}, v2{b.re, b.im}, t);
    t = __builtin_elementwise_fma(v2{-ai, ai}, v2{b.im, b.re}, t);
    acc.re = t.x; acc.im = t.y;
}
template <typename R> __device__ __forceinline__ void cfma_conj(Cx<R> &acc, R ar, R ai, const Cx<R> &b)
{
    using v2 = typename V2<R>::type;
    v2 t = {acc.re, acc.im};
    t = __builtin_elementwise_fma(v2{ar, ai}, v2{b.re, b.re}, t);
    t = __builtin_elementwise_fma(v2{ai, -ar}, v2{b.im, b.im}, t);
    acc.re = t.x; acc.im = t.y;
}

// one entry of the sweep exchange buffer: a wave's contribution to an output plus its "my c moved" flag (one ds_write / ds_read)
template <typename R> struct alignas(16) BiEnt { Cx<R> v; unsigned flag; unsigned pad; };

constexpr int BI_NT = 64 * BI_W;         // threads per workgroup
constexpr int BI_RPW = 64 / BI_W;        // rows (steps / taps) a wave reduces per ds_read: lanes = BI_RPW groups of BI_W

// Adaptive step size (adapt_step, pythran_equalisation.py:12-16, :171-172) inside the block-iterative form.  After step
// i > 0 the reference keeps mu when both component products of err[i], err[i-1] are positive and else sets
// mu <- mu / (1 + mu |err[i-1]|^2), i.e. 1/mu grows by |err[i-1]|^2: in r = 1/mu the recurrence is a PREFIX SUM of
// decrements d_i that depend on the errors only.  Every sweep therefore re-derives the step sizes of its 64 steps from the
// current errors (own rows: DPP prefix in a compact lane <-> row layout; other waves: their totals of the last sweep,
// exchanged next to the contributions), c_i = e_i / r_i, and the fixed point is again the sequential recurrence's
// (mu itself differs from the reference's float recurrence only by the rounding of 1/(r + d) vs mu/(1 + mu d)).
template <typename R> struct alignas(16) BiAd { R S, er, ei, pad; };     // a wave's decrement total and last error of the sweep

__device__ __forceinline__ float bperm(float v, int byte_addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double bperm(double v, int byte_addr)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)b), hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114;

// Decision on an ARBITRARY alphabet inside the block-iterative form (round 5; 32- / 128-QAM crosses, where no per-axis slicer exists).  After the
// exchange the 8 lanes of a group hold the same output y; lane v of the group scans the symbols [v ceil(M / 8), (v + 1) ceil(M / 8)) from LDS with the
// reference's strict `<` (det_symbol, pythran_equalisation.py:258-264: first minimum in symbol order, from d0 = 1000 and the symbol 1 + 0j), then the
// group takes the smallest distance with ties to the LOWER index - the reference's first minimum - in three DPP exchanges; the winner's symbol is
// read back from LDS by index.  ceil(M / 8) distance evaluations per lane and sweep instead of M per step.
constexpr int BI_GEN_MAXSYM = 256;
template <typename T> __device__ __forceinline__ T bi_xchg(T v, int lvl)
{
    return lvl == 0 ? dpp_mov<DPP_QUAD_1032>(v) : (lvl == 1 ? dpp_mov<DPP_QUAD_2301>(v) : dpp_mov<DPP_ROW_HALF_MIRROR>(v));
}
template <typename R> __device__ __forceinline__ Cx<R> bi_nearest_general(Cx<R> y, const Cx<R> *alph, int M, int vv)
{
    static_assert(BI_W == 8, "three exchange levels cover a group of 8 lanes");
    const int per = (M + BI_W - 1) / BI_W;
    R best = (R)1000;
    int bidx = 0x7fffffff;                                            // "nothing closer than 1000": det_symbol's initial symbol survives
    const int k0 = vv * per;
    for (int u = 0; u < per; u++) {
        const int k = k0 + u;
        if (k < M) {
            const Cx<R> c = alph[k];
            const R dr = y.re - c.re, di = y.im - c.im;
            const R d = fma_(dr, dr, di * di);
            if (d < best) { best = d; bidx = k; }
        }
    }
#pragma unroll
    for (int lvl = 0; lvl < 3; lvl++) {
        const R od = bi_xchg<R>(best, lvl);
        const int oi = __builtin_bit_cast(int, bi_xchg<float>(__builtin_bit_cast(float, bidx), lvl));
        const bool take = od < best || (od == best && oi < bidx);
        best = take ? od : best; bidx = take ? oi : bidx;
    }
    if (bidx == 0x7fffffff) return Cx<R>{(R)1, (R)0};
    return alph[bidx];
}
// error function of a decision-directed method given the decided symbol (la_errfn's formulas; SCALE: the step size folded in)
template <typename R, int METHOD, bool SCALE> __device__ __forceinline__ Cx<R> bi_dd_error(Cx<R> y, Cx<R> sdec, R mu)
{
    using v2 = typename V2<R>::type;
    const v2 yy = {y.re, y.im}, s = {sdec.re, sdec.im};
    v2 d;
    if constexpr (METHOD == QH_M_SBD) d = (s - yy) * __builtin_elementwise_abs(s);
    else if constexpr (METHOD == QH_M_MDDMA) d = (s * s - yy * yy) * yy;
    else d = s - yy;
    if constexpr (SCALE) d = d * mu;
    return Cx<R>{d.x, d.y};
}
struct BiScaled { static constexpr bool value = true; };
struct BiPlain { static constexpr bool value = false; };

template <typename R, int METHOD, int NPART, bool ADAPT = false>
__global__ void __launch_bounds__(BI_NT) train_bi_kernel(LaArgs<R> a)
{
    if (a.skip && *a.skip) return;
    // independent captures of a channel bank / segments of a sweep (blockIdx.y): same shapes, own arrays
    const int64_t ch = blockIdx.y;
    const LaView<R> vw = la_view(a, ch);
    const Cx<R> *const aE = vw.E;
    Cx<R> *const awx = vw.wx;
    Cx<R> *const aerr = vw.err;
    const GramPair<R> *const aG = vw.G;
    const int64_t aL = vw.L;
    const R *const amu = a.mu + ch * a.mu_cs + (int64_t)blockIdx.x * a.mu_ms;
    extern __shared__ __attribute__((aligned(16))) char bi_smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mode = (int)a.modes[blockIdx.x];
    const int ntot = a.nmodes * a.ntaps;
    const int os_ = a.os;
    const int64_t TrSyms = vw.TrSyms;
    const int nblk = (int)((TrSyms + LA_B - 1) / LA_B);
    const Cx<R> *sy = a.symbols + (size_t)mode * a.sy_pitch;

    // ---- LDS carve-up
    const int wlen = (LA_B - 1) * os_ + a.ntaps;
    const int wpitch = (wlen + 1) & ~1;
    const int wsz = a.nmodes * wpitch;
    BiEnt<R> *P = reinterpret_cast<BiEnt<R> *>(bi_smem);              // [2][64][BI_PAD]   sweep exchange
    Cx<R> *TW = reinterpret_cast<Cx<R> *>(P + 2 * LA_B * BI_PAD);      // [BI_MAXTAPS][BI_PAD] tap-update exchange
    Cx<R> *wbuf = TW + BI_MAXTAPS * BI_PAD;                            // [BI_MAXTAPS]      taps, wave-uniform reads
    Cx<R> *win = wbuf + BI_MAXTAPS;                                    // [2][nmodes][wpitch] sample windows (block parity)
    BiAd<R> *adx = reinterpret_cast<BiAd<R> *>(win + 2 * wsz);         // [2][BI_W] adaptive-step exchange (ADAPT only)
    // decision-directed method without slicer tables: the alphabet itself, scanned per decision (bi_nearest_general)
    constexpr bool GEN = NPART == 0 && (METHOD == QH_M_SBD || METHOD == QH_M_MDDMA || METHOD == QH_M_DD);
    Cx<R> *alph = reinterpret_cast<Cx<R> *>(adx + 2 * BI_W);           // [nsy] (GEN only)
    const int nalph = GEN ? (int)a.nsy : 0;
    if constexpr (GEN) for (int k2 = threadIdx.x; k2 < nalph; k2 += BI_NT) alph[k2] = sy[k2];      // (barriers follow before the first decision)

    // ---- constants of the error function
    LaConst<R, NPART> K;
    K.mu = *amu;
    {
        const Cx<R> c0 = sy[0];
        K.R_re = c0.re; K.R_im = c0.im;
    }
    K.code0_re = K.R_re; K.code0_im = K.R_im;
    tab_fill<R, NPART>(K.tab, sy, 0, NPART + 1);

    // ---- tap ownership: wave w reduces / dots the taps [f0, f0 + nf)
    const int tpw = (((ntot + BI_W - 1) / BI_W) + 3) & ~3;              // multiple of 4: the prior dot products run 4 taps at a time
    const int f0 = w * tpw;
    const int nf = f0 < ntot ? ((f0 + tpw) < ntot ? tpw : ntot - f0) : 0;
    Cx<R> *wrow = awx + (size_t)mode * ntot;
    for (int f = threadIdx.x; f < BI_MAXTAPS; f += BI_NT) wbuf[f] = f < ntot ? wrow[f] : Cx<R>{0, 0};     // taps >= ntot stay zero
    // tap-update layout: lane <-> taps lane and lane + 64
    int xo[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int f = lane + 64 * s;
        const int fc = f < ntot ? f : 0;
        const int k2 = fc / a.ntaps;
        xo[s] = k2 * wpitch + (fc - k2 * a.ntaps);
    }
    const int rr = lane / BI_W, vv = lane % BI_W;                      // reduction layout: group rr <-> row, lane vv of the group <-> wave
    // the error function of an output that is identical in the 8 lanes of its group
    auto errf = [&](Cx<R> yv, Cx<R> sd, auto SC) __attribute__((always_inline)) -> Cx<R> {
        constexpr bool SCALE = decltype(SC)::value;
        if constexpr (GEN) return bi_dd_error<R, METHOD, SCALE>(yv, bi_nearest_general<R>(yv, alph, nalph, vv), K.mu);
        else return la_errfn<R, METHOD, NPART, SCALE>(yv, K, sd);
    };
    // window offset of tap f0 + lane of the own slice (prior dot products fetch it with v_readlane); 0 for padding taps
    int toffv;
    {
        const int f = f0 + lane;
        const int fc = (lane < tpw && f < ntot) ? f : 0;
        const int k2 = fc / a.ntaps;
        toffv = k2 * wpitch + (fc - k2 * a.ntaps);
    }
    // sample staging: thread <-> up to two window elements, source offsets fixed up front
    int st_dst[2]; int64_t st_src[2]; int st_i[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int e = threadIdx.x + s * BI_NT;
        const int ec = e < wsz ? e : 0;
        const int k2 = ec / wpitch;
        st_dst[s] = e < wsz ? ec : -1;
        st_i[s] = ec - k2 * wpitch;
        st_src[s] = (int64_t)k2 * a.Lp;
    }
    Cx<R> stv[2];
    auto stage_load = [&](int kb) {                                    // global loads now ...
        const int64_t base = (int64_t)kb * LA_B * os_;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            int64_t gi = base + st_i[s];
            if (gi > aL - 1) gi = aL - 1;
            stv[s] = aE[st_src[s] + gi];
        }
    };
    auto stage_store = [&](int kb) {                                   // ... LDS writes once the window is free
        const int64_t base = (int64_t)kb * LA_B * os_;
        Cx<R> *dst = win + (size_t)(kb & 1) * wsz;
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (st_dst[s] >= 0) dst[st_dst[s]] = stv[s];
        for (int e = threadIdx.x + 2 * BI_NT; e < wsz; e += BI_NT) {      // very long windows only
            const int k2 = e / wpitch, i2 = e - k2 * wpitch;
            int64_t gi = base + i2;
            if (gi > aL - 1) gi = aL - 1;
            dst[e] = aE[(size_t)k2 * a.Lp + gi];
        }
    };
    // own-slice part of the prior outputs of block kb: lane <-> step kb*64 + lane
    auto prior_part = [&](int kb) {
        const Cx<R> *xs = win + (size_t)(kb & 1) * wsz + lane * os_;
        const Cx<R> *ws = wbuf + f0;
        Cx<R> acc{0, 0};
        for (int f = 0; f < tpw; f += 4) {
            Cx<R> x[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                x[u] = xs[readlane(toffv, f + u)];
                wv[u] = ws[f + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) cfma<R>(acc, x[u].re, x[u].im, wv[u]);
        }
        if ((int64_t)kb * LA_B + lane >= TrSyms) acc = Cx<R>{0, 0};
        return acc;
    };
    // Gram rows of the own steps: either the look-ahead pair layout (cur / next interleaved, zeros where target <= step) or
    // the triangular cur-only layout (only targets > step are stored: lanes at or before the step load a dummy and are
    // zeroed when the prefetched rows are taken over - never at load time, that would stall on the load)
    const int gstride = a.gpair ? LA_B * LA_B * 2 : GRAM_TRI;         // Cx elements per block
    int goff[BI_JW];
    bool gok[BI_JW];
#pragma unroll
    for (int r = 0; r < BI_JW; r++) {
        const int j = BI_JW * w + r;
        gok[r] = a.gpair || lane > j;
        goff[r] = a.gpair ? (j * LA_B + lane) * 2 : (lane > j ? gram_tri_row(j) + lane - j - 1 : 0);
    }
    auto load_gram = [&](Cx<R> (&g)[BI_JW], int kb) {
        const Cx<R> *gp = reinterpret_cast<const Cx<R> *>(aG) + (size_t)kb * gstride;
#pragma unroll
        for (int r = 0; r < BI_JW; r++) g[r] = gp[goff[r]];
    };
    auto mask_gram = [&](Cx<R> (&g)[BI_JW]) {
#pragma unroll
        for (int r = 0; r < BI_JW; r++) g[r] = gok[r] ? g[r] : Cx<R>{0, 0};
    };
    // own tap slice in registers: group rr of round t <-> tap f0 + BI_RPW * t + rr (BI_MAXTAPS / 64 = 2 rounds at most)
    Cx<R> wreg[2] = {Cx<R>{0, 0}, Cx<R>{0, 0}};
    R r_blk = ADAPT ? (R)1 / K.mu : (R)0;                              // adaptive step: r = 1/mu at the start of the block, carried from sweep to sweep
    unsigned long long pf_sweeps = 0, pf_t_sweep = 0, pf_t_upd = 0, pf_t_prior = 0;
    // The reference's Niter loop (pythran_equalisation.py:163-165) INSIDE the launch: taps and step size stay in registers / LDS from sweep
    // to sweep; what a sweep needs from scratch is its first two sample windows, the prior outputs of block 0 and the Gram rows of block 0
    // (the pilot stages of config 5 are 30 sweeps of 1024 steps: a launch per sweep was a third of their time).
    const int nsweep = a.niter > 1 ? a.niter : 1;
    for (int it = 0; it < nsweep; it++) {
    if (it > 0) __syncthreads();                                        // the last block's readers of the sample windows are done
    Cx<R> *errow = aerr + (size_t)mode * a.err_pitch + a.err_off + (int64_t)it * TrSyms;
    stage_load(0); stage_store(0);
    if (nblk > 1) { stage_load(1); stage_store(1); }
    __syncthreads();
    if (it == 0) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int f = f0 + BI_RPW * t + rr;
            wreg[t] = (BI_RPW * t + rr < nf) ? wbuf[f] : Cx<R>{0, 0};
        }
    }
    Cx<R> qpart = prior_part(0);
    Cx<R> g[BI_JW], gn[BI_JW];
    load_gram(g, 0);
    mask_gram(g);

    BiEnt<R> *pw = P + (size_t)lane * BI_PAD + w;                       // this wave's column of the exchange buffer
    const BiEnt<R> *pr = P + (size_t)(BI_JW * w + rr) * BI_PAD + vv;   // the rows it reduces
    constexpr int PBUF = LA_B * BI_PAD;
    Cx<R> *eown = errow + BI_JW * w + rr;                               // this group's slot of the error trace
    // adaptive step: last error of the previous block (none at the start of a sweep: the reference adapts from step 1 on), compact layout helpers
    Cx<R> e_carry{0, 0};
    const int cl = lane & 7;                                            // compact layout: lane <-> row (lane & 7) of this wave
    const int csrc = cl * BI_W * 4;                                     // ds_bpermute byte address of that row's group

    for (int k = 0; k < nblk; k++) {
        const unsigned long long pt0 = a.prof ? clock64() : 0;
        const int64_t s0 = (int64_t)k * LA_B;
        Cx<R> sdat{0, 0};                                              // data-aided: the training symbol of this group's step
        if constexpr (METHOD == QH_M_SBD_DATA) {
            const int64_t gi = s0 + BI_JW * w + rr;
            sdat = sy[gi < TrSyms ? gi : TrSyms - 1];
        }
        load_gram(gn, k + 1 < nblk ? k + 1 : k);                       // next block's rows and the samples of block k+2 arrive during the sweeps
        stage_load(k + 2 < nblk ? k + 2 : k);
        // ---------------------------------------------------------------- fixed-point sweeps
        Cx<R> part = qpart;
        Cx<R> c_old{0, 0}, y{0, 0};
        unsigned changed = 1;
        // one sweep through exchange buffer `pb`; true when the fixed point has been reached
        Cx<R> e_last{0, 0};                                             // ADAPT: own total / last error published with the next exchange
        R S_own = 0;
        R r_next = r_blk;
        auto sweep = [&](const int pb) -> bool {
            BiEnt<R> ent;
            ent.v = part; ent.flag = changed; ent.pad = 0;
            pw[pb] = ent;
            if constexpr (ADAPT) adx[(pb ? BI_W : 0) + w] = BiAd<R>{S_own, e_last.re, e_last.im, 0};
            __syncthreads();
            const BiEnt<R> got = pr[pb];
            y = got.v;
            group_csum(y.re, y.im);                                        // y of row rr, identical in the lanes of the group
            if (a.prof) pf_sweeps++;
            Cx<R> c_new;
            if constexpr (ADAPT) {
                const BiAd<R> ax = adx[(pb ? BI_W : 0) + cl];               // lane <-> wave (lane & 7): totals of the last sweep
                R below = cl < w ? ax.S : (R)0, all = ax.S;                  // decrements of the waves before this one / of the block
                below += dpp_mov<DPP_QUAD_1032>(below);       all += dpp_mov<DPP_QUAD_1032>(all);
                below += dpp_mov<DPP_QUAD_2301>(below);       all += dpp_mov<DPP_QUAD_2301>(all);
                below += dpp_mov<DPP_ROW_HALF_MIRROR>(below); all += dpp_mov<DPP_ROW_HALF_MIRROR>(all);
                r_next = r_blk + all;                                        // 1/mu after the block (final once converged)
                if (!__any(got.flag != 0)) {                                 // converged: remember the block's last error
                    e_carry = Cx<R>{readlane(ax.er, BI_W - 1), readlane(ax.ei, BI_W - 1)};
                    return true;
                }
                const Cx<R> eprev0 = w == 0 ? e_carry : Cx<R>{readlane(ax.er, w > 0 ? w - 1 : 0), readlane(ax.ei, w > 0 ? w - 1 : 0)};
                const Cx<R> e = errf(y, sdat, BiPlain{});
                const Cx<R> ec{bperm(e.re, csrc), bperm(e.im, csrc)};       // compact: lane <-> row cl
                Cx<R> ep{dpp_mov<DPP_ROW_SHR1>(ec.re), dpp_mov<DPP_ROW_SHR1>(ec.im)};
                if (cl == 0) ep = eprev0;
                const int64_t gi = s0 + BI_JW * w + cl;                       // global step of the row
                const bool keep = ec.re * ep.re > 0 && ec.im * ep.im > 0;
                R d = (keep || gi == 0 || gi >= TrSyms) ? (R)0 : fma_(ep.re, ep.re, ep.im * ep.im);
                R inc = d;                                                   // inclusive prefix over the 8 rows
                { const R t = dpp_mov<DPP_ROW_SHR1>(inc); inc += cl >= 1 ? t : (R)0; }
                { const R t = dpp_mov<DPP_ROW_SHR2>(inc); inc += cl >= 2 ? t : (R)0; }
                { const R t = dpp_mov<DPP_ROW_SHR4>(inc); inc += cl >= 4 ? t : (R)0; }
                const R r_row = (r_blk + below) + (inc - d);                 // 1/mu in force at this row's step
                const R mu_row = (R)1 / r_row;
                c_new = Cx<R>{mu_row * ec.re, mu_row * ec.im};
                const R S_new = readlane(inc, BI_JW - 1);
                const Cx<R> el_new{readlane(ec.re, BI_JW - 1), readlane(ec.im, BI_JW - 1)};
                const unsigned long long mv = __builtin_amdgcn_ballot_w64(c_new.re != c_old.re || c_new.im != c_old.im);
                changed = (unsigned)mv | (unsigned)(mv >> 32) | (S_new != S_own) | (el_new.re != e_last.re) | (el_new.im != e_last.im);
                S_own = S_new; e_last = el_new;
            } else {
                if (!__any(got.flag != 0)) return true;                    // nobody's c moved in the last sweep: y is the fixed point
                c_new = errf(y, sdat, BiScaled{});
                const unsigned long long mv = __builtin_amdgcn_ballot_w64(c_new.re != c_old.re || c_new.im != c_old.im);
                changed = (unsigned)mv | (unsigned)(mv >> 32);             // wave-uniform: non-zero when any own c moved
            }
            c_old = c_new;
            part = qpart;                                                  // own steps' contributions to every pending output
#pragma unroll
            for (int r = 0; r < BI_JW; r++) {
                const int src = ADAPT ? r : BI_W * r;                      // compact layout holds row r in lane r
                cfma<R>(part, readlane(c_new.re, src), readlane(c_new.im, src), g[r]);
            }
            return false;
        };
        for (int it = 0; it <= LA_B + 2; it += 2) {
            if (sweep(0)) break;
            if (sweep(PBUF)) break;
        }
        // ---------------------------------------------------------------- results of the block
        const unsigned long long pt1 = a.prof ? clock64() : 0;
        const Cx<R> e = errf(y, sdat, BiPlain{});
        if (vv == 0 && s0 + BI_JW * w + rr < TrSyms) eown[s0] = e;
        // taps: this wave's steps into all taps (lane <-> taps lane, lane + 64).  Steps past TrSyms (partial last block) have
        // all-zero Gram rows, so they never touched the sweeps; their c (non-zero for decision-directed functions) is dropped here
        if (s0 + BI_JW * w + (ADAPT ? cl : rr) >= TrSyms) c_old = Cx<R>{0, 0};
        if constexpr (ADAPT) r_blk = r_next;
        {
            const Cx<R> *xw = win + (size_t)(k & 1) * wsz + (BI_JW * w) * os_;
            Cx<R> dw[2] = {{0, 0}, {0, 0}};
#pragma unroll
            for (int r = 0; r < BI_JW; r++) {
                const R cr = readlane(c_old.re, ADAPT ? r : BI_W * r), ci = readlane(c_old.im, ADAPT ? r : BI_W * r);
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const Cx<R> x = xw[xo[s] + r * os_];
                    cfma_conj<R>(dw[s], cr, ci, x);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; s++) TW[(size_t)(lane + 64 * s) * BI_PAD + w] = dw[s];
        }
        __syncthreads();
        // taps: slice owners sum the contributions of all waves (BI_RPW taps per read, one lane group each)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (BI_RPW * t < tpw) {
                const int f = f0 + BI_RPW * t + rr;
                Cx<R> d = TW[(size_t)f * BI_PAD + vv];
                group_csum(d.re, d.im);
                if (BI_RPW * t + rr < nf) { wreg[t].re += d.re; wreg[t].im += d.im; }
                if (BI_RPW * t + rr < tpw) wbuf[f] = wreg[t];              // all lanes of a group write the same value
            }
        }
        // samples of block k+2 replace block k's: every wave finished reading those before the barrier above, and the next
        // readers (prior outputs of block k+2) sit behind the sweep barriers of block k+1
        if (k + 2 < nblk) stage_store(k + 2);
        const unsigned long long pt2c = a.prof ? clock64() : 0;
        if (k + 1 < nblk) qpart = prior_part(k + 1);                      // reads the own slice of wbuf only: no barrier needed
#pragma unroll
        for (int r = 0; r < BI_JW; r++) g[r] = gok[r] ? gn[r] : Cx<R>{0, 0};
        if (a.prof) {
            const unsigned long long pt3 = clock64();
            pf_t_sweep += pt1 - pt0; pf_t_upd += pt2c - pt1; pf_t_prior += pt3 - pt2c;
        }
    }
    }   // sweeps
    __syncthreads();
    if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) {
        a.prof[0] = pf_sweeps; a.prof[1] = pf_t_sweep; a.prof[2] = pf_t_upd; a.prof[3] = pf_t_prior; a.prof[4] = (unsigned long long)nblk;
    }
    for (int f = threadIdx.x; f < ntot; f += BI_NT) wrow[f] = wbuf[f];
    if constexpr (ADAPT) if (threadIdx.x == 0) a.mu_out[ch * a.mu_cs + (int64_t)blockIdx.x * a.mu_ms] = (R)1 / r_blk;
}

// ------------------------------------------------------------------------------------------------ slicer tables
// Decision-directed error functions (sbd, mddma, dd) pick the nearest symbol of an arbitrary alphabet (det_symbol,
// pythran_equalisation.py:240-265: linear scan).  For a square alphabet - every (re level, im level) combination
// present once, equally many levels per axis - that is the nearest level per axis, i.e. the same partition look-up rde /
// mrde use.  One workgroup per mode analyses the row `symbols[mode]` and writes the table in the rde / mrde layout:
// n codes (sorted levels, re / im axis in .re / .im) followed by n-1 partitions (midpoints); info[mode] = n-1, or -1
// when the row is not such an alphabet (the direct-form kernel then scans it).
constexpr int BI_DD_MAXLEV = 16;
template <typename R>
__global__ void __launch_bounds__(64) slicer_table_kernel(const Cx<R> *symbols, int M, Cx<R> *table, int *info)
{
    __shared__ R sre[1024], sim[1024];
    __shared__ unsigned char seen[BI_DD_MAXLEV * BI_DD_MAXLEV];
    const Cx<R> *row = symbols + (size_t)blockIdx.x * M;
    Cx<R> *tab = table + (size_t)blockIdx.x * (2 * BI_DD_MAXLEV);
    if (M > 1024 || M < 4) { if (threadIdx.x == 0) info[blockIdx.x] = -1; return; }
    for (int k = threadIdx.x; k < M; k += 64) { const Cx<R> s = row[k]; sre[k] = s.re; sim[k] = s.im; }
    for (int k = threadIdx.x; k < BI_DD_MAXLEV * BI_DD_MAXLEV; k += 64) seen[k] = 0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    R lre[BI_DD_MAXLEV], lim[BI_DD_MAXLEV];
    int nre = 0, nim = 0;
    bool ok = true;
    for (int k = 0; k < M && ok; k++) {
        int r = 0, i = 0;
        ok = sre[k] == sre[k] && sim[k] == sim[k];                     // NaN alphabets stay with the scan
        while (r < nre && lre[r] != sre[k]) r++;
        if (r == nre) { if (nre < BI_DD_MAXLEV) lre[nre++] = sre[k]; else ok = false; }
        while (i < nim && lim[i] != sim[k]) i++;
        if (i == nim) { if (nim < BI_DD_MAXLEV) lim[nim++] = sim[k]; else ok = false; }
        if (ok) { if (seen[r * BI_DD_MAXLEV + i]) ok = false; else seen[r * BI_DD_MAXLEV + i] = 1; }
    }
    ok = ok && nre == nim && nre >= 2 && (long)nre * nim == M;
    if (!ok) { info[blockIdx.x] = -1; return; }
    for (int a = 1; a < nre; a++) {                                    // insertion sort, <= 16 levels
        R v = lre[a]; int b = a - 1;
        while (b >= 0 && lre[b] > v) { lre[b + 1] = lre[b]; b--; }
        lre[b + 1] = v;
        v = lim[a]; b = a - 1;
        while (b >= 0 && lim[b] > v) { lim[b + 1] = lim[b]; b--; }
        lim[b + 1] = v;
    }
    for (int a = 0; a < nre; a++) tab[a] = Cx<R>{lre[a], lim[a]};
    for (int a = 0; a + 1 < nre; a++) tab[nre + a] = Cx<R>{(lre[a] + lre[a + 1]) / 2, (lim[a] + lim[a + 1]) / 2};
    info[blockIdx.x] = nre - 1;
}

// tables for all modes; *npart = common partition count of the selected modes, or -1 (one small D2H copy + sync)
template <typename R> int slicer_tables(const void *symbols, int nmodes, int64_t nsy, const int64_t *modes, int nsel, void **table, int *npart)
{
    *npart = -1;
    if (nsy > 1024 || nsy < 4) return QH_OK;
    void *buf = nullptr;
    int rc = scratch(7, (size_t)nmodes * 2 * BI_DD_MAXLEV * sizeof(Cx<R>) + (size_t)nmodes * sizeof(int), &buf);
    if (rc) return rc;
    int *info = reinterpret_cast<int *>((char *)buf + (size_t)nmodes * 2 * BI_DD_MAXLEV * sizeof(Cx<R>));
    hipLaunchKernelGGL((slicer_table_kernel<R>), dim3(nmodes), dim3(64), 0, g_stream, (const Cx<R> *)symbols, (int)nsy, (Cx<R> *)buf, info);
    QH_HIP(hipGetLastError());
    int h[16];
    QH_REQUIRE(nmodes <= 16, "train_equaliser: more than 16 modes");
    QH_HIP(hipMemcpyAsync(h, info, (size_t)nmodes * sizeof(int), hipMemcpyDeviceToHost, g_stream));
    QH_HIP(hipStreamSynchronize(g_stream));
    int np = h[modes[0]];
    for (int j = 1; j < nsel; j++) if (h[modes[j]] != np) np = -1;
    *npart = np;
    *table = buf;
    return QH_OK;
}

// ------------------------------------------------------------------------------------------------ host side
template <typename R> static size_t gram_cur_bytes(int64_t TrSyms)
{
    const int64_t nblk = (TrSyms + LA_B - 1) / LA_B;
    return (size_t)(nblk + 1) * GRAM_TRI * sizeof(Cx<R>);
}

template <typename R> int gram_cur_build(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram, int nch = 1,
                                         int64_t Lp = 0, int64_t ch_stride = 0)
{
    if (Lp <= 0) Lp = L;
    if (ch_stride <= 0) ch_stride = (int64_t)nmodes * Lp;
    int rc = ensure_init();
    if (rc) return rc;
    void *G = nullptr;
    const size_t bytes = gram_cur_bytes<R>(TrSyms);
    if ((rc = scratch(4, bytes * (size_t)nch, &G))) return rc;
    const int64_t nblk = (TrSyms + LA_B - 1) / LA_B;
    const size_t lds = (size_t)nmodes * ((LA_B - 1) * os + ntaps) * sizeof(Cx<R>);
    QH_REQUIRE(lds <= 64 * 1024, "gram: nmodes*(63*os+ntaps) samples exceed the LDS tile");
    for (int c0 = 0; c0 < nch && nblk > 0; c0 += 65535) {          // one launch for the bank (blockIdx.y = channel; grid.y limit)
        const int nc = nch - c0 < 65535 ? nch - c0 : 65535;
        hipLaunchKernelGGL((gram_slide_kernel<R, false>), dim3((unsigned)nblk, (unsigned)nc), dim3(256), lds, g_stream, (const Cx<R> *)E + (size_t)c0 * ch_stride,
                           nmodes, L, Lp, os, ntaps, TrSyms, (Cx<R> *)((char *)G + bytes * (size_t)c0), ch_stride, (int64_t)(bytes / sizeof(Cx<R>)));
    }
    QH_HIP(hipGetLastError());
    *gram = G;
    return QH_OK;
}

template <typename R> static size_t bi_lds_bytes(int nmodes, int ntaps, int os)
{
    const int wpitch = ((LA_B - 1) * os + ntaps + 1) & ~1;
    return (size_t)2 * LA_B * BI_PAD * sizeof(BiEnt<R>) + ((size_t)BI_MAXTAPS * BI_PAD + BI_MAXTAPS + (size_t)2 * nmodes * wpitch) * sizeof(Cx<R>) +
           2 * BI_W * sizeof(BiAd<R>);
}

// sizes the block-iterative kernel can hold (the Gram layout of a capture follows from this alone)
inline bool bi_shape_ok(int nmodes, int ntaps, int os, size_t elem)
{
    const char *force = trainer_force();
    if ((force[0] == 'd' || force[0] == 'l')) return false;      // "direct" / "lookahead": A/B measurements, tests
    if (nmodes * ntaps > BI_MAXTAPS) return false;
    const int wpitch = ((LA_B - 1) * os + ntaps + 1) & ~1;
    return (size_t)2 * LA_B * BI_PAD * (elem + 16) + ((size_t)BI_MAXTAPS * BI_PAD + BI_MAXTAPS + (size_t)2 * nmodes * wpitch) * elem + 2 * BI_W * 2 * elem <= BI_LDS_MAX;
}

// decision-directed methods on an alphabet without per-axis slicer (bi_nearest_general): the alphabet rides in LDS behind the kernel's other arrays
inline bool bi_general_ok(int nmodes, int ntaps, int os, int64_t nsy, size_t elem)
{
    if (nsy < 1 || nsy > BI_GEN_MAXSYM || !bi_shape_ok(nmodes, ntaps, os, elem)) return false;
    if (elem > 8 && nsy > 64) return false;           // double precision, > 64 symbols: the direct form's in-register search is as fast (measured 32.5 against 35 ms, 128-QAM)
    const int wpitch = ((LA_B - 1) * os + ntaps + 1) & ~1;
    return (size_t)2 * LA_B * BI_PAD * (elem + 16) + ((size_t)BI_MAXTAPS * BI_PAD + BI_MAXTAPS + (size_t)2 * nmodes * wpitch + (size_t)nsy) * elem + 2 * BI_W * 2 * elem <= BI_LDS_MAX;
}

inline bool bi_supported(int method, int adaptive, int nmodes, int ntaps, int os, int64_t TrSyms, int64_t nsy, size_t elem)
{
    (void)adaptive;                       // the adaptive step runs in this form too (one mode after the other, mu carried)
    if (TrSyms < 2 * LA_B) return false;
    if (!bi_shape_ok(nmodes, ntaps, os, elem)) return false;
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: case QH_M_CMA2: case QH_M_MCMA: return true;
    case QH_M_RDE: case QH_M_MRDE: return nsy - (nsy + 1) / 2 >= 1 && nsy - (nsy + 1) / 2 <= LA_MAXPART;
    case QH_M_SBD: case QH_M_MDDMA: case QH_M_DD: return true;      // square alphabets: slicer_tables(); any other: bi_general_ok()
    case QH_M_SBD_DATA: return true;
    default: return false;
    }
}

// One launch of the kernel; shapes whose arrays need more than the 64 KiB a workgroup gets by default (complex128 with two or more modes) ask for the
// CU's whole LDS once per kernel instance (160 KiB on gfx950; one workgroup per CU then, which is what a latency chain wants anyway).
template <typename R, int METHOD, int NPART, bool ADAPT> static void bi_launch(dim3 grid, dim3 block, size_t lds, const LaArgs<R> &a)
{
    if (lds > 64 * 1024) {
        static std::atomic<bool> allowed{false};
        if (!allowed.load(std::memory_order_acquire)) {
            (void)hipFuncSetAttribute((const void *)train_bi_kernel<R, METHOD, NPART, ADAPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BI_LDS_MAX);
            allowed.store(true, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((train_bi_kernel<R, METHOD, NPART, ADAPT>), grid, block, lds, g_stream, a);
}

template <typename R, int METHOD, bool ADAPT> static int launch_bi_dd(const LaArgs<R> &a, int npart, size_t lds)
{
    dim3 grid(a.nsel, a.nch), block(BI_NT);
#define QH_BI_DD(N) case N: bi_launch<R, METHOD, N, ADAPT>(grid, block, lds, a); break;
    if (a.dd_general) {         // any alphabet: the symbols themselves in LDS (bi_nearest_general)
        if (a.nsy < 1 || a.nsy > BI_GEN_MAXSYM) { set_error("block-iterative trainer: alphabet too large for the general decision"); return QH_ERR_ARG; }
        bi_launch<R, METHOD, 0, ADAPT>(grid, block, lds + (size_t)a.nsy * sizeof(Cx<R>), a);
        return QH_OK;
    }
    switch (npart) {            // 4-, 16-, 64-, 256-QAM
        QH_BI_DD(1) QH_BI_DD(3) QH_BI_DD(7) QH_BI_DD(15)
    default: set_error("block-iterative trainer: unsupported slicer size"); return QH_ERR_ARG;
    }
#undef QH_BI_DD
    return QH_OK;
}

template <typename R, int METHOD, bool ADAPT> static int launch_bi_parts(const LaArgs<R> &a, int npart, size_t lds)
{
    dim3 grid(a.nsel, a.nch), block(BI_NT);
#define QH_BI_NP(N) case N: bi_launch<R, METHOD, N, ADAPT>(grid, block, lds, a); break;
    switch (npart) {
        QH_BI_NP(1) QH_BI_NP(2) QH_BI_NP(3) QH_BI_NP(4) QH_BI_NP(5) QH_BI_NP(6) QH_BI_NP(7) QH_BI_NP(8)
    default: set_error("block-iterative trainer: unsupported partition count"); return QH_ERR_ARG;
    }
#undef QH_BI_NP
    return QH_OK;
}

template <typename R, bool ADAPT> static int launch_bi_t(const LaArgs<R> &a)
{
    dim3 grid(a.nsel, a.nch), block(BI_NT);
    const int npart = (int)(a.nsy - (a.nsy + 1) / 2);
    const size_t lds = bi_lds_bytes<R>(a.nmodes, a.ntaps, a.os);
    int rc = QH_OK;
    switch (a.method) {
    case QH_M_CMA: case QH_M_SGNCMA: bi_launch<R, QH_M_CMA, 0, ADAPT>(grid, block, lds, a); break;
    case QH_M_CMA2: bi_launch<R, QH_M_CMA2, 0, ADAPT>(grid, block, lds, a); break;
    case QH_M_MCMA: bi_launch<R, QH_M_MCMA, 0, ADAPT>(grid, block, lds, a); break;
    case QH_M_RDE: rc = launch_bi_parts<R, QH_M_RDE, ADAPT>(a, npart, lds); break;
    case QH_M_MRDE: rc = launch_bi_parts<R, QH_M_MRDE, ADAPT>(a, npart, lds); break;
    case QH_M_SBD: rc = launch_bi_dd<R, QH_M_SBD, ADAPT>(a, npart, lds); break;
    case QH_M_MDDMA: rc = launch_bi_dd<R, QH_M_MDDMA, ADAPT>(a, npart, lds); break;
    case QH_M_DD: rc = launch_bi_dd<R, QH_M_DD, ADAPT>(a, npart, lds); break;
    case QH_M_SBD_DATA: bi_launch<R, QH_M_SBD_DATA, 0, ADAPT>(grid, block, lds, a); break;
    default: return QH_ERR_METHOD;
    }
    if (rc) return rc;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

// adaptive: the caller launches one mode at a time (nsel = 1) so that mu is carried from mode to mode like in the reference
template <typename R> int launch_bi(const LaArgs<R> &a, bool adaptive = false)
{
    return adaptive ? launch_bi_t<R, true>(a) : launch_bi_t<R, false>(a);
}

}  // namespace qh
