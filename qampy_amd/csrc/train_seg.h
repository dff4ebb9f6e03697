// Throughput form of the equaliser recurrence for MANY concurrent chains (segments of a sweep, train_pit.h).
//
// The exact-path kernels minimise the latency of ONE chain (look-ahead: 13 instructions per step on the critical wave, paid
// for with a 1 KiB-per-step Gram table; block-iterative: 8 waves per chain).  With thousands of chains in flight the stage is
// bound by instruction issue (and the look-ahead form by streaming its table: 4.3 GB per pass at C3), so this kernel spends
// as few instructions per chain and step as the recurrence allows and reads nothing but the capture:
//   * LPC = 16 or 8 lanes per chain, 4 or 8 chains per wave64 (one wave per workgroup, no workgroup barrier anywhere);
//   * the nmodes*ntaps taps of a chain are spread over its lanes, TPL consecutive taps of one input mode per lane, and live
//     in registers for the whole segment; per step a lane reads its TPL samples from an LDS window (staged chunk-wise with
//     coalesced loads, double buffered; chains of the same segment - its output modes - share the window);
//   * the dot product is reduced over the chain's lanes with 3 or 4 DPP steps (quad_perm, row_half_mirror, row_mirror: every
//     lane ends up with the total), the error function is evaluated redundantly in all lanes, every lane updates its own taps;
//   * the complex multiply-adds are packed FMAs with operand selectors (pk_re / pk_im / pk_im_rot below);
//   * per wave and step at 41 taps x 2 modes: 68 instructions for 4 chains (16 lanes, 6 taps per lane), 94 for 8 chains (8 lanes,
//     11 taps per lane) - 17 / 11.8 per chain and step, against ~25 of the look-ahead form over its 4 waves and ~125 of the
//     block-iterative one.  8 lanes per chain is used from 3000 chains on (single precision, layouts with 6 or 11 taps per lane).
// Same recurrence, same error functions (la_errfn); results equal the other forms up to the order of the additions in the dot
// product (lane tree instead of 64-lane tree / look-ahead identity).
#pragma once
#include "train_bi.h"

namespace qh {

constexpr int SG_CH = 64;          // steps per staged chunk

template <typename R> struct SegArgs {
    const Cx<R> *E;
    Cx<R> *wx;                // (S, nmodes, nmodes*ntaps) tap sets, trained in place
    const Cx<R> *symbols;
    Cx<R> *err;               // rows of err_pitch, this sweep starts at column err_off
    const R *mu;
    int64_t L, TrSyms, nsy, sy_pitch, err_pitch, err_off;
    int nmodes, ntaps, os, nsel, S;
    int64_t seg_len, seg_extra, seg_tail, seg_begin;
    int64_t modes[16];
    const int *skip;
    int q_first, q_count;          // chains [q_first, q_first + q_count) of the S * nsel are trained by this launch (q_count = 0: all)
    int lpm, pitch, rag, nslots;   // lanes per input mode, LDS row pitch (samples), padding taps in the last lane of a mode, segment windows per wave
    int ch;                        // steps per staged chunk (SG_CH, or SG_CH_LONG where the layout has the longer rows and the filter fits them)
    // adaptive step (ADAPT kernels, pythran_equalisation.py:12-16, :171-172): per chain r = 1 / mu and the previous error at the start of
    // its segment in, at the end out, and the sum of the step sizes its steps used (the coarse model's mu T); adapt_first = 0: the
    // sweep's step 0 belongs to this grid (no adaptation after it)
    const R *r_in; R *r_out; const Cx<R> *e_in; Cx<R> *e_out; R *mu_sum; int adapt_first;
};

// sum over the 16 lanes of a row; every lane gets the total
__device__ __forceinline__ void row16_csum(float &re, float &im)
{
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
        : "+v"(re), "+v"(im));
}
__device__ __forceinline__ void row16_csum(double &re, double &im)
{
    re += dpp_mov<DPP_QUAD_1032>(re);       im += dpp_mov<DPP_QUAD_1032>(im);
    re += dpp_mov<DPP_QUAD_2301>(re);       im += dpp_mov<DPP_QUAD_2301>(im);
    re += dpp_mov<DPP_ROW_HALF_MIRROR>(re); im += dpp_mov<DPP_ROW_HALF_MIRROR>(im);
    re += dpp_mov<DPP_ROW_MIRROR>(re);      im += dpp_mov<DPP_ROW_MIRROR>(im);
}

// acc + x.re * (b.re, b.im) / acc + x.im * (b.re, b.im) / acc + x.im * (b.im, -b.re): ONE packed instruction each - the halves of
// the operands are picked with op_sel / neg_hi instead of being copied into place (the compiler materialises the broadcasts
// with v_mov: 24 of them per step)
typedef float sg_f2 __attribute__((ext_vector_type(2)));
typedef double sg_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sg_f2 pk_re(sg_f2 x, sg_f2 b, sg_f2 acc)
{
    sg_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(x), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ sg_f2 pk_im(sg_f2 x, sg_f2 b, sg_f2 acc)
{
    sg_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(x), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ sg_f2 pk_im_rot(sg_f2 x, sg_f2 b, sg_f2 acc)
{
    sg_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(b), "v"(acc));
    return d;
}
// acc + x.im * (-b.im, b.re): the second half of a complex multiply-add acc + x * b
__device__ __forceinline__ sg_f2 pk_imw(sg_f2 x, sg_f2 b, sg_f2 acc)
{
    sg_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(x), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ sg_d2 pk_imw(sg_d2 x, sg_d2 b, sg_d2 acc) { return __builtin_elementwise_fma(sg_d2{x.y, x.y}, sg_d2{-b.y, b.x}, acc); }
__device__ __forceinline__ sg_d2 pk_re(sg_d2 x, sg_d2 b, sg_d2 acc) { return __builtin_elementwise_fma(sg_d2{x.x, x.x}, b, acc); }
__device__ __forceinline__ sg_d2 pk_im(sg_d2 x, sg_d2 b, sg_d2 acc) { return __builtin_elementwise_fma(sg_d2{x.y, x.y}, b, acc); }
__device__ __forceinline__ sg_d2 pk_im_rot(sg_d2 x, sg_d2 b, sg_d2 acc) { return __builtin_elementwise_fma(sg_d2{x.y, x.y}, sg_d2{b.y, -b.x}, acc); }
// first terms of the accumulators: x.re * (b.re, b.im) / x.im * (b.re, b.im) without a zero to add to
__device__ __forceinline__ sg_f2 pk_re0(sg_f2 x, sg_f2 b)
{
    sg_f2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(b));
    return d;
}
__device__ __forceinline__ sg_f2 pk_im0(sg_f2 x, sg_f2 b)
{
    sg_f2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(b));
    return d;
}
__device__ __forceinline__ sg_d2 pk_re0(sg_d2 x, sg_d2 b) { return sg_d2{x.x, x.x} * b; }
__device__ __forceinline__ sg_d2 pk_im0(sg_d2 x, sg_d2 b) { return sg_d2{x.y, x.y} * b; }
template <int N> struct SgInt { static constexpr int value = N; };

// sum over the 8 lanes of a half row
__device__ __forceinline__ void row8_csum(float &re, float &im)
{
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
        : "+v"(re), "+v"(im));
}
__device__ __forceinline__ void row8_csum(double &re, double &im)
{
    re += dpp_mov<DPP_QUAD_1032>(re);       im += dpp_mov<DPP_QUAD_1032>(im);
    re += dpp_mov<DPP_QUAD_2301>(re);       im += dpp_mov<DPP_QUAD_2301>(im);
    re += dpp_mov<DPP_ROW_HALF_MIRROR>(re); im += dpp_mov<DPP_ROW_HALF_MIRROR>(im);
}
template <int LPC, typename R> __device__ __forceinline__ void chain_csum(R &re, R &im)
{
    if (LPC == 8) row8_csum(re, im); else row16_csum(re, im);
}


// ---- one step of the 16-lane layout as hand-scheduled blocks (single precision, 4 or 6 taps per lane, unrolled groups of steps)
// A lone wave issues one instruction per turn of its SIMD (~5 cycles) whatever the instruction is - an s_nop costs as much as
// a packed FMA - and gfx950 wants 2 wait states between a VALU write and a DPP read of it and 1 after every packed operation
// before its result is read.  Round 4's step paid 8 s_nop and 2 s_waitcnt for that (58 issue slots, 46 of them VALU), most of them
// padding the compiler puts around inline-assembly instructions: to it they are opaque nodes that it places anywhere and then pads
// (an operand written by the instruction right before an asm statement costs an s_nop, whatever the statement does with it).  So the
// step is made of statements that schedule themselves, with the error function (the compiler's) in between:
//   block A  y = sum w x over the lane's taps as P = sum x.re w, R = sum x.im w on the even / odd taps (four accumulators, every FMA
//            four instructions after the one it depends on), P0 + P1, R0 + R1, y = P + i R (ONE packed add with a rotated, half-negated
//            operand), the four DPP levels, (y.re^2, y.im^2).  The wait states in there hold work that has to be done anyway and does
//            not depend on it: the two selects that park the PREVIOUS step's error in its lane of the trace group (KEEP), the
//            ds_read_b128 of the window of the NEXT pair of steps (NL of them, at immediate offsets from one address register per group -
//            the compiler is out of the LDS business in this loop, so it also stops placing an s_waitcnt before every first use: one
//            per pair, sg_wait_lds), the samples of the padding taps times the lane's 0 / 1 mask;
//   block B  c = d y and its rotated copy, then w += c conj(x) as two rounds of packed FMAs, x.re (c.re, c.im) and x.im (c.im, -c.re),
//            a tap's second FMA TPL - 1 instructions behind its first - in ONE statement with block A of the next step (seg_block_ba).
// v[244:255] are scratch of the blocks (the halves of a packed sum feed the DPP adds and the selects; operands of inline assembly have no
// sub-registers to name).  DESIGN.md 3.2.2.
typedef float sg_f4 __attribute__((ext_vector_type(4)));
// x[0 .. TPL), w[0 .. TPL): the samples and taps of this step; y: the chain's output, in every lane (the pair v[246:247]: the DPP adds write its
// halves, what follows reads it as a packed operand), sq = (y.re^2, y.im^2) - every blind error function starts with it, and behind the block's
// last DPP add it costs no wait state; ebr / ebi: the lane's slot of the trace group, (per, pei) parked there in the lanes of mask mk; d0, d1:
// window pieces read from LDS address la + O0 (+ 16); xm0 .. xm[NR - 1]: the samples of the NR padding taps (the last ones) times tm - 0 in the
// last lane of an input mode, 1 elsewhere: block B updates the padding taps with THOSE, so they stay the zeros they start as, and zeroing them
// costs a wait state of the tree instead of an instruction of block B.
// The 32 statements (taps per lane x KEEP x pieces x padding taps) are written by scripts/gen_seg_blocks.py, which fills the wait states of the
// sums and of the tree from one list of independent instructions in a fixed order.
template <int TPL, bool KEEP, int NL, int NR, int O0>
__device__ __forceinline__ void seg_block_a(const sg_f2 (&x)[TPL], const sg_f2 (&w)[TPL], sg_f2 &y, sg_f2 &sq, float &ebr, float &ebi,
                                            float per, float pei, unsigned long long mk, unsigned la, sg_f4 &d0, sg_f4 &d1, sg_f4 &d2, sg_f4 &d3, sg_f2 tm, sg_f2 &xm0, sg_f2 &xm1, sg_f2 &xm2)
{
    static_assert(((TPL == 4 || TPL == 6) && (NL == 1 || NL == 2)) || (TPL == 11 && (NL == 3 || NL == 4)), "4 or 6 taps per lane: one or two window pieces per step; 11: three or four");
    static_assert(NR >= 0 && NR <= 3, "up to three padding taps");
#include "train_seg_blocks.inc"
    if constexpr (!KEEP) { (void)ebr; (void)ebi; (void)per; (void)pei; (void)mk; }
    if constexpr (NL < 2) (void)d1;
    if constexpr (NL < 3) (void)d2;
    if constexpr (NL < 4) (void)d3;
    if constexpr (NR < 1) { (void)tm; (void)xm0; }
    if constexpr (NR < 2) (void)xm1;
    if constexpr (NR < 3) (void)xm2;
}
// block B of the previous step (its samples xp - padding taps masked -, its output yp and factor dp: c = dp yp) and block A of this step in ONE
// statement: the compiler has nothing to pad between them.  c lives in v[244:245] for the tap update and for the selects that park it (lane mask mk);
// written by scripts/gen_seg_blocks.py like block A.
template <int TPL, bool D1, int NL, int NR, int O0>
__device__ __forceinline__ void seg_block_ba(const sg_f2 (&xp)[TPL], sg_f2 yp, sg_f2 dp, const sg_f2 (&x)[TPL], sg_f2 (&w)[TPL], sg_f2 &y, sg_f2 &sq, float &ebr, float &ebi,
                                             unsigned long long mk, unsigned la, sg_f4 &d0, sg_f4 &d1, sg_f4 &d2, sg_f4 &d3, sg_f2 tm, sg_f2 &xm0, sg_f2 &xm1, sg_f2 &xm2)
{
    static_assert(((TPL == 4 || TPL == 6) && (NL == 1 || NL == 2)) || (TPL == 11 && (NL == 3 || NL == 4)), "4 or 6 taps per lane: one or two window pieces per step; 11: three or four");
    static_assert(NR >= 0 && NR <= 3, "up to three padding taps");
    sg_f2 cr;
#include "train_seg_blocks_ba.inc"
    if constexpr (NL < 2) (void)d1;
    if constexpr (NL < 3) (void)d2;
    if constexpr (NL < 4) (void)d3;
    if constexpr (NR < 1) { (void)tm; (void)xm0; }
    if constexpr (NR < 2) (void)xm1;
    if constexpr (NR < 3) (void)xm2;
}
// block B on its own (behind the last step of a group, and for the error functions that are not of the form d(y) y): w[j] += x[j].re (c.re, c.im) +
// x[j].im (cr.re, cr.im) with cr = (c.im, -c.re); x: the step's samples with those of the padding taps masked (block A's xm)
#define SG_UR(j) "v_pk_fma_f32 %[w" #j "], %[x" #j "], %[c], %[w" #j "] op_sel_hi:[0,1,1]\n\t"
#define SG_UI(j) "v_pk_fma_f32 %[w" #j "], %[x" #j "], %[cr], %[w" #j "] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define SG_W4_ [w0] "+v"(w[0]), [w1] "+v"(w[1]), [w2] "+v"(w[2]), [w3] "+v"(w[3])
#define SG_W6_ SG_W4_, [w4] "+v"(w[4]), [w5] "+v"(w[5])
#define SG_W11_ SG_W6_, [w6] "+v"(w[6]), [w7] "+v"(w[7]), [w8] "+v"(w[8]), [w9] "+v"(w[9]), [w10] "+v"(w[10])
#define SG_X4_ [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [c] "v"(c), [cr] "v"(cr)
#define SG_X6_ SG_X4_, [x4] "v"(x[4]), [x5] "v"(x[5])
#define SG_X11_ SG_X6_, [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10])
#define SG_UR11 SG_UR(0) SG_UR(1) SG_UR(2) SG_UR(3) SG_UR(4) SG_UR(5) SG_UR(6) SG_UR(7) SG_UR(8) SG_UR(9) SG_UR(10)
#define SG_UI11 SG_UI(0) SG_UI(1) SG_UI(2) SG_UI(3) SG_UI(4) SG_UI(5) SG_UI(6) SG_UI(7) SG_UI(8) SG_UI(9) SG_UI(10)
template <int TPL>
__device__ __forceinline__ void seg_block_b(const sg_f2 (&x)[TPL], sg_f2 (&w)[TPL], sg_f2 c, sg_f2 cr)
{
    static_assert(TPL == 4 || TPL == 6 || TPL == 11, "layouts with 4, 6 or 11 taps per lane");
    // (a tap's second FMA comes TPL - 1 instructions after its first; the first instruction of the next block A reads w[0], written TPL before the end)
    if constexpr (TPL == 11) asm volatile(SG_UR11 SG_UI11 : SG_W11_ : SG_X11_);
    else if constexpr (TPL == 6) asm volatile(SG_UR(0) SG_UR(1) SG_UR(2) SG_UR(3) SG_UR(4) SG_UR(5) SG_UI(0) SG_UI(1) SG_UI(2) SG_UI(3) SG_UI(4) SG_UI(5) : SG_W6_ : SG_X6_);
    else asm volatile(SG_UR(0) SG_UR(1) SG_UR(2) SG_UR(3) SG_UI(0) SG_UI(1) SG_UI(2) SG_UI(3) : SG_W4_ : SG_X4_);
}
// the same for error functions of the form e = d(y) y: c = mu e and its rotated copy are the block's first two instructions (d: the factor
// with the step size folded in - one per axis, or ONE real factor in the low half of its operand (D1) - written by the compiler's last
// instruction before the block; c is returned for the trace)
#define SG_CV "v_pk_mul_f32 %[c], %[yy], %[d]\n\tv_pk_mul_f32 %[cr], %[yy], %[d] op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
#define SG_C1 "v_pk_mul_f32 %[c], %[yy], %[d] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %[cr], %[yy], %[d] op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
#define SG_Y4_ [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [yy] "v"(yy), [d] "v"(d)
#define SG_Y6_ SG_Y4_, [x4] "v"(x[4]), [x5] "v"(x[5])
#define SG_Y11_ SG_Y6_, [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10])
#define SG_CO_ , [c] "=&v"(c), [cr] "=&v"(cr)
template <int TPL, bool D1>
__device__ __forceinline__ sg_f2 seg_block_b2(const sg_f2 (&x)[TPL], sg_f2 (&w)[TPL], sg_f2 yy, sg_f2 d)
{
    static_assert(TPL == 4 || TPL == 6 || TPL == 11, "layouts with 4, 6 or 11 taps per lane");
    sg_f2 c, cr;
#define SG_B2(CM) \
    if constexpr (TPL == 11) asm volatile(CM SG_UR11 SG_UI11 : SG_W11_ SG_CO_ : SG_Y11_); \
    else if constexpr (TPL == 6) asm volatile(CM SG_UR(0) SG_UR(1) SG_UR(2) SG_UR(3) SG_UR(4) SG_UR(5) SG_UI(0) SG_UI(1) SG_UI(2) SG_UI(3) SG_UI(4) SG_UI(5) : SG_W6_ SG_CO_ : SG_Y6_); \
    else asm volatile(CM SG_UR(0) SG_UR(1) SG_UR(2) SG_UR(3) SG_UI(0) SG_UI(1) SG_UI(2) SG_UI(3) : SG_W4_ SG_CO_ : SG_Y4_);
    if constexpr (D1) { SG_B2(SG_C1) } else { SG_B2(SG_CV) }
#undef SG_B2
    return c;
}
#undef SG_CV
#undef SG_C1
#undef SG_Y4_
#undef SG_Y6_
#undef SG_Y11_
#undef SG_CO_
#undef SG_UR11
#undef SG_UI11
#undef SG_UR
#undef SG_UI
#undef SG_W4_
#undef SG_W6_
#undef SG_W11_
#undef SG_X4_
#undef SG_X6_
#undef SG_X11_
// (e.im, -e.re) of e = d * y, component by component, from y and d: the rotated operand of the second round of the tap update
__device__ __forceinline__ sg_f2 pk_rot_mul(sg_f2 y, sg_f2 d)
{
    sg_f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(o) : "v"(y), "v"(d));
    return o;
}
// the same with one real factor d in the low half of its operand (the high half is not read)
__device__ __forceinline__ sg_f2 pk_rot_mul1(sg_f2 y, sg_f2 d)
{
    sg_f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(o) : "v"(y), "v"(d));
    return o;
}
// the window pieces asked for during the previous pair of steps have arrived (their only reader waits here: data dependence pins the order)
template <int NQ> __device__ __forceinline__ void sg_wait_lds(sg_f4 (&q)[NQ])
{
    static_assert(NQ >= 2 && NQ <= 7, "window of 4 to 14 samples");
    if constexpr (NQ == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]));
    else if constexpr (NQ == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]));
    else if constexpr (NQ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]));
    else if constexpr (NQ == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]));
    else if constexpr (NQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]));
}

// the factor d of c = mu e = d y for the blind error functions of that form, from y and (y.re^2, y.im^2), with the step size folded in as far
// as it goes (cma: mu R - mu |y|^2 is one FMA; mcma: one packed FMA; mrde: the codes of the table come pre-scaled by mu - Ks - so that
// mu (r - y^2) is one packed FMA behind the look-up; rde: as la_errfn has it).  Same functions as la_errfn, rounded differently in the last bit.
// tab_lookup_med3 (train_la.h) in two phases: h_p = (sq - part_p) 2^60 for every partition, then r = med3(r, code_p, h_p) partition by partition
template <int N, int P, int NH> __device__ __forceinline__ void seg_med3_diffs(sg_f2 sq, const PartTab<float, N> &t, sg_f2 (&h)[NH])
{
    if constexpr (N > 0) {
        constexpr float BIG = 1152921504606846976.0f;                 // 2^60
        h[P] = sq * BIG - sg_f2{t.part_re, t.part_im} * BIG;
        seg_med3_diffs<N - 1, P + 1>(sq, t.next, h);
    }
}
template <int N, int P, int NH> __device__ __forceinline__ void seg_med3_pick(float &rr, float &ri, const PartTab<float, N> &t, const sg_f2 (&h)[NH])
{
    if constexpr (N > 0) {
        rr = __builtin_amdgcn_fmed3f(rr, t.code_re, h[P].x);
        ri = __builtin_amdgcn_fmed3f(ri, t.code_im, h[P].y);
        seg_med3_pick<N - 1, P + 1>(rr, ri, t.next, h);
    }
}
template <int METHOD, int NPART>
__device__ __forceinline__ auto seg_errfn_d(sg_f2 sq, const LaConst<float, NPART> &k, const LaConst<float, NPART> &ks)
{
    if constexpr (METHOD == QH_M_CMA || METHOD == QH_M_SGNCMA) {
        return __builtin_fmaf(-k.mu, sq.x + sq.y, k.mu * k.R_re);
    } else if constexpr (METHOD == QH_M_MCMA) {
        return __builtin_elementwise_fma(sq, sg_f2{-k.mu, -k.mu}, sg_f2{k.mu * k.R_re, k.mu * k.R_im});
    } else if constexpr (METHOD == QH_M_RDE) {
        const float s = sq.x + sq.y;
        return (tab_lookup<float, NPART, false>(s, k.code0_re, k.tab) - s) * k.mu;
    } else {   // QH_M_MRDE
        // (the compiler's own order - one packed FMA for a partition, its two med3, the next FMA into the same registers - pays a wait state per
        // partition: all the differences first, then the two med3 chains, interleaved)
        sg_f2 h[NPART > 0 ? NPART : 1];
        seg_med3_diffs<NPART, 0>(sq, ks.tab, h);
        __builtin_amdgcn_sched_barrier(0);
        float rr = ks.code0_re, ri = ks.code0_im;
        seg_med3_pick<NPART, 0>(rr, ri, ks.tab, h);
        return __builtin_elementwise_fma(sq, sg_f2{-k.mu, -k.mu}, sg_f2{rr, ri});
    }
}

constexpr int SG_PITCH = 192;      // samples per LDS row (one segment window of one input mode): 3 pieces of 64
constexpr int SG_PIECES = SG_PITCH / 64;
// The 16-lane layout in single precision (fixed step) stages chunks of 128 steps where they fit - rows of 5 pieces: what a chunk costs beside its steps
// (address arithmetic of the staging loads, stores, the two fences, the first window, the dispatch on the padding count) is ~170 vector instructions,
// 2.6 per step at 64 steps per chunk - 6 % of the wave's instruction stream (SQ_INSTS_VALU 45.7 per step against 43.1 in the unrolled loop).
// (256 steps per chunk - 23 KiB of LDS per wave - are faster one capture at a time, 4.48 against 4.57 ms, and slower over consecutive captures: the phase
// search beside the passes is bound by the waves a CU holds, i.e. by the LDS the passes leave it - 976 against 1107 MSym/s)
constexpr int SG_CH_LONG = 128, SG_PITCH_LONG = 320;
template <typename R, int LPC, bool ADAPT> struct SgRow { static constexpr int pitch = (sizeof(R) == 4 && LPC == 16 && !ADAPT) ? SG_PITCH_LONG : SG_PITCH; };
// rows per buffer (segment windows of the wave x input modes) = chains per wave; staging registers per lane = rows x pieces
constexpr int SG_MAXRAG = 3;       // padding taps a lane may hold (handled by selects on its last three tap slots)

#ifdef QH_SEG_KERNELS        // the kernels and their per-method launchers live in train_seg_{a,b}_{f32,f64}.hip (build time: ~190 instantiations)
template <typename R, int METHOD, int NPART, int TPL, int LPC, bool ADAPT = false>
__global__ void __launch_bounds__(64) train_seg_kernel(SegArgs<R> a)
{
    constexpr int CPW = 64 / LPC;                               // chains per wave
    constexpr int PITCH = SgRow<R, LPC, ADAPT>::pitch, PIECES = PITCH / 64;
    constexpr int SG_ROWS = ADAPT ? 2 * CPW : CPW, SG_NSTG = SG_ROWS * PIECES;   // (ADAPT: one output mode per launch - every chain of the wave its own segment window)
    const int CH = a.ch;                                         // steps per staged chunk
    if (a.skip && *a.skip) return;
    // One wave per SIMD, by construction.  A launch of this kernel has about as many single-wave workgroups as the chip has SIMDs
    // (992 at C3's mrde stage); with <= 256 VGPRs two of them fit on a SIMD and the dispatcher does pair them up while other SIMDs
    // of the same CU stay empty - each of the pair then issues every other turn (measured: 783 instead of 553 cycles per step).
    // Naming the last VGPR and one AGPR as clobbered pushes the allocation past half the register file, so that a second wave
    // never fits (any layout, any precision; with more chains than SIMDs the waves queue up and still run alone).
#ifndef QH_SEG_DUAL                                             // (build switch for measurements: two waves may share a SIMD)
    asm volatile("" ::: "v255", "a0");                          // allocation = 256 VGPRs + the first AGPR granule > half the file
#endif
    // What CAN share the SIMD is a narrow streaming kernel of another stream (the phase search of the previous capture, 64 registers a
    // wave: pipeline.py run(overlap=True)); this wave is the latency-bound one, so it goes first whenever both have an instruction ready.
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char sg_smem[];
    Cx<R> *lds = reinterpret_cast<Cx<R> *>(sg_smem);          // [SG_ROWS][PITCH] + zero row [PITCH]  (ONE buffer: the next chunk waits in registers
                                                              // while this one computes and is stored after it - one wave per workgroup, LDS operations in program order)
    using v2 = typename V2<R>::type;
    const int lane = threadIdx.x;
    const int l16 = lane & (LPC - 1);                          // lane within the chain
    const int nq = a.q_count > 0 ? a.q_first + a.q_count : a.S * a.nsel;      // one past the last chain of the launch
    const int q0 = a.q_first + blockIdx.x * CPW;
    const int q = q0 + lane / LPC;
    const bool alive = q < nq;
    const int qc = alive ? q : nq - 1;
    const int seg = qc / a.nsel, jsel = qc - seg * a.nsel;
    const int seg0 = q0 / a.nsel;                              // first segment of this wave
    const int segl = ((q0 + CPW - 1 < nq ? q0 + CPW - 1 : nq - 1)) / a.nsel;
    const int nslot = segl - seg0 + 1;
    const int slot = seg - seg0;
    const int mode = (int)a.modes[jsel];
    const int ntot = a.nmodes * a.ntaps;
    const int os_ = a.os;
    auto seg_start = [&](int64_t s) { return a.seg_begin + s * a.seg_len + (s < a.seg_extra ? s : a.seg_extra) * LA_B; };
    auto seg_steps = [&](int64_t s) { return (int)(a.seg_len + (s < a.seg_extra ? LA_B : 0) + (s == a.S - 1 ? a.seg_tail : 0)); };
    const int64_t my_start = seg_start(seg);
    const int my_steps = alive ? seg_steps(seg) : 0;
    int max_steps = 0, min_steps = 0x7fffffff;                  // over the chains of the wave (a wave with a dead chain checks every step)
    for (int s = seg0; s <= segl; s++) { const int n = seg_steps(s); max_steps = n > max_steps ? n : max_steps; min_steps = n < min_steps ? n : min_steps; }
    if (q0 + CPW - 1 >= nq) min_steps = 0;

    // ---- taps: lane <-> TPL consecutive taps of input mode kin
    const int kin = l16 / a.lpm, t0 = (l16 - kin * a.lpm) * TPL;
    const bool has = kin < a.nmodes;
    Cx<R> *wrow = a.wx + ((size_t)seg * a.nmodes + mode) * ntot;
    v2 w[TPL];
#pragma unroll
    for (int j = 0; j < TPL; j++) {
        const bool ok = has && t0 + j < a.ntaps;
        const Cx<R> v = ok ? wrow[kin * a.ntaps + t0 + j] : Cx<R>{0, 0};
        w[j] = v2{v.re, v.im};
    }
    // ---- error-function constants
    const Cx<R> *sy = a.symbols + (size_t)mode * a.sy_pitch;
    LaConst<R, NPART> K;
    K.mu = *a.mu;
    { const Cx<R> c0 = sy[0]; K.R_re = c0.re; K.R_im = c0.im; }
    K.code0_re = K.R_re; K.code0_im = K.R_im;
    tab_fill<R, NPART>(K.tab, sy, 0, NPART + 1);
    LaConst<R, NPART> Ks = K;                                   // codes times the step size (fixed step, single precision: seg_errfn_d)
    Ks.code0_re *= K.mu; Ks.code0_im *= K.mu;
    tab_scale_codes<R, NPART>(Ks.tab, K.mu);
    // ADAPT: the chain's step size as r = 1 / mu (adapt_step adds |e_prev|^2 to it unless both component products of successive
    // errors are positive), the previous error, the sum of the step sizes used
    R ad_r = 1, ad_sum = 0;
    Cx<R> ad_ep{0, 0};
    bool ad_skip0 = false;                                      // this chain's step 0 is step 0 of the sweep: no adaptation after it
    if constexpr (ADAPT) { ad_r = a.r_in[qc]; ad_ep = a.e_in[qc]; ad_skip0 = a.adapt_first == 0 && my_start == 0; }

    // ---- LDS windows: SG_ROWS rows of PITCH samples per buffer, one per (segment window, input mode)
    constexpr int rowsz = PITCH;
    constexpr int bufsz = SG_ROWS * PITCH;
    const int nrow = nslot * a.nmodes;                          // <= SG_ROWS (checked on the host)
    Cx<R> *zero_row = lds + bufsz;
    for (int e = lane; e < rowsz; e += 64) zero_row[e] = Cx<R>{0, 0};
    int64_t rowbase[SG_ROWS], rowlim[SG_ROWS];                  // sample offset of row r at chunk 0, last sample of its capture row (wave-uniform)
#pragma unroll
    for (int r = 0; r < SG_ROWS; r++) {
        const int rc = r < nrow ? r : 0;
        const int sl = rc / a.nmodes, k = rc - sl * a.nmodes;
        rowbase[r] = (int64_t)k * a.L + seg_start(seg0 + sl) * os_;
        rowlim[r] = (int64_t)k * a.L + a.L - 1;
    }
    Cx<R> stg_r[SG_NSTG];
    // global loads of chunk `chunk` into registers (no wait) ...
    auto stage_load = [&](int chunk) __attribute__((always_inline)) {
        const int64_t adv = (int64_t)chunk * CH * os_;
#pragma unroll
        for (int u = 0; u < SG_NSTG; u++) {
            const int row = u / PIECES, piece = u % PIECES;
            if (row < nrow) {
                int64_t g = rowbase[row] + adv + piece * 64 + lane;
                if (g > rowlim[row]) g = rowlim[row];           // reads stay inside their row of the capture
                stg_r[u] = ldg(a.E + g);
            }
        }
    };
    // ... and from there into the buffer of that chunk once the previous user of the buffer is done
    auto stage_store = [&](int chunk) __attribute__((always_inline)) {
        Cx<R> *dst = lds;
#pragma unroll
        for (int u = 0; u < SG_NSTG; u++)
            if (u / PIECES < nrow) dst[u * 64 + lane] = stg_r[u];
    };
    Cx<R> *errow = a.err + (size_t)mode * a.err_pitch + a.err_off + my_start;
    // The error of a step is the same in all lanes of its chain; lane l keeps the error of step l of a group of LPC steps (two
    // selects per step - the groups are unrolled, so "is this my step" is a loop-invariant lane mask in a scalar register pair, not a
    // compare per step) and stores it when the group is complete: one coalesced store per chain and group.
    R ebr = 0, ebi = 0;
    unsigned long long mks[LPC];                                // "lane u of its chain" as wave-wide lane masks (the selects of the fused lane tree)
#pragma unroll
    for (int u = 0; u < LPC; u++) mks[u] = __builtin_amdgcn_ballot_w64(l16 == u);
    // fixed step: the step-size-scaled error mu e is what the update needs; the trace is taken from it (x 1 / mu when a group of LPC errors is
    // stored) instead of evaluating the error function a second time without the factor (2-3 instructions per step; the host sends
    // sweeps with mu = 0 to the exact path)
    const R inv_mu = ADAPT ? (R)1 : (R)1 / K.mu;
    // padding taps: the last lane of an input mode may hold up to SG_MAXRAG of them, in its last slots.  They start as zeros and
    // stay zeros because that lane's update of those slots is multiplied by 0 (tailmask; 1 in every other lane).
    const bool lastlane = has && (l16 - kin * a.lpm) == a.lpm - 1;
    const v2 tailmask = lastlane ? v2{0, 0} : v2{1, 1};
    const int rag = a.rag;

    constexpr int WIN = TPL + 2;                                 // samples of a pair of steps at 2 samples per symbol: they share TPL - 2 of them
    auto load_x = [&](v2 (&x)[WIN], const Cx<R> *p, auto NLOAD) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < decltype(NLOAD)::value; j++) { const Cx<R> v = p[j]; x[j] = v2{v.re, v.im}; }
    };
    // one step on the samples x[OFF .. OFF + TPL); returns the error (unscaled) of the step
    // (always_inline: a lambda left as a call takes the taps - captured by reference - through scratch memory, as the 2-taps-per-lane
    // layout did: 5000 instead of 300 cycles per step)
    // y = sum w x over this lane's taps (no conjugate, pythran_equalisation.py:24-31), accumulated as the complex number it is: tap j
    // adds x.re * (w.re, w.im) and x.im * (-w.im, w.re) to accumulator j mod NACC (pk_re / pk_imw: the rotation of w is an operand
    // selector, not an instruction), so the partial sums need NACC - 1 packed additions and nothing else before the lane tree
    // (round 4 kept the two products per tap apart: 4 additions to combine them + 2 more to form re / im).  A packed FMA that
    // reads the result of another one wants three or four independent instructions in between (a lone wave issues every ~4.3
    // cycles, the packed pipe answers after ~13): NACC chains, the re-parts of a group of NACC taps first, then their im-parts.
    constexpr int NACC = TPL == 6 ? 3 : (TPL <= 2 ? TPL : 4);
    auto dot = [&](auto xof) __attribute__((always_inline)) -> v2 {
        v2 acc[NACC];
#pragma unroll
        for (int b = 0; b < TPL; b += NACC) {
#pragma unroll
            for (int q = 0; q < NACC; q++) if (b + q < TPL) acc[q] = b == 0 ? pk_re0(xof(q), w[q]) : pk_re(xof(b + q), w[b + q], acc[q]);
#pragma unroll
            for (int q = 0; q < NACC; q++) if (b + q < TPL) acc[q] = pk_imw(xof(b + q), w[b + q], acc[q]);
        }
        v2 ysum = acc[0];
#pragma unroll
        for (int q = 1; q < NACC; q++) ysum += acc[q];
        return ysum;
    };
    // error function of the chain's output y and the tap update with it; returns the error of the step (fixed step: times mu)
    auto finish = [&](Cx<R> y, auto xof, int gstep, auto CHK, auto RG) __attribute__((always_inline)) -> Cx<R> {
        Cx<R> e{0, 0};
        Cx<R> cc;
        if constexpr (ADAPT) {
            e = la_errfn<R, METHOD, NPART, false>(y, K);
            const bool live = !decltype(CHK)::value || gstep < my_steps;
            R m;
            if constexpr (sizeof(R) == 4) m = __builtin_amdgcn_rcpf(ad_r); else m = (R)1 / ad_r;
            cc = Cx<R>{m * e.re, m * e.im};
            const bool keep = (e.re * ad_ep.re > 0) && (e.im * ad_ep.im > 0);
            const bool first = ad_skip0 && gstep == 0;
            if (live) {
                if (!keep && !first) ad_r += ad_ep.re * ad_ep.re + ad_ep.im * ad_ep.im;
                ad_ep = e;
                ad_sum += m;
            }
        } else {
            cc = la_errfn<R, METHOD, NPART, true>(y, K);       // mu * e with mu folded in
            e = cc;                                            // (the trace: x inv_mu at the store)
        }
        if (decltype(CHK)::value && gstep >= my_steps) cc = Cx<R>{0, 0};   // past the end of this chain's segment: nothing moves
        // w += c conj(x):  (re, im) += x.re (c.re, c.im) + x.im (c.im, -c.re)
        const v2 c1 = {cc.re, cc.im};
        constexpr int NR = decltype(RG)::value;
        v2 ct = c1;
        if (NR > 0) ct = c1 * tailmask;
#pragma unroll
        for (int j = 0; j < TPL; j++) w[j] = pk_re(xof(j), j >= TPL - NR ? ct : c1, w[j]);        // two rounds: no instruction waits
#pragma unroll
        for (int j = 0; j < TPL; j++) w[j] = pk_im_rot(xof(j), j >= TPL - NR ? ct : c1, w[j]);    // for the one right before it
        return e;
    };
    // one step on the samples x[OFF .. OFF + TPL); returns the error of the step
    // (always_inline: a lambda left as a call takes the taps - captured by reference - through scratch memory, as the 2-taps-per-lane
    // layout did: 5000 instead of 300 cycles per step)
    auto step = [&](auto XO, v2 (&x)[WIN], int gstep, auto CHK, auto RG) __attribute__((always_inline)) -> Cx<R> {
        constexpr int OFF = decltype(XO)::value;
        auto xof = [&](int j) __attribute__((always_inline)) -> v2 { return x[OFF + j]; };
        const v2 ysum = dot(xof);
        R yr = ysum.x, yi = ysum.y;
        chain_csum<LPC>(yr, yi);
        return finish(Cx<R>{yr, yi}, xof, gstep, CHK, RG);
    };
    auto keep = [&](Cx<R> e, int u) __attribute__((always_inline)) { const bool mine = l16 == u; ebr = mine ? e.re : ebr; ebi = mine ? e.im : ebi; };
    auto run_chunk = [&](const Cx<R> *xs, int xstep, int ibase, int nst, auto CHK, auto RG, auto OS2) __attribute__((always_inline)) {
        constexpr bool os2 = decltype(OS2)::value != 0;
        v2 xa[WIN], xb[WIN];
        int i = 0;
        // ---- whole groups of LPC steps
        constexpr bool FAST = sizeof(R) == 4 && !ADAPT && ((LPC == 16 && (TPL == 4 || TPL == 6)) || (LPC == 8 && TPL == 11));
        if constexpr (FAST) {
            // single precision, 16 lanes per chain, 2 samples per symbol: the same pairs of steps on windows of TPL + 2 samples, the
            // windows as 128-bit pieces; the pieces of the next pair's window are asked for from inside the lane trees of this pair
            // and the previous step's error is parked there too (tree16_fused).  Lanes without taps walk through the zero row.
            if (os2) {
                // (11 taps per lane - the 8-lane layout, round 6: a window of 13 samples, seven pieces with the last half unused; a lane's window starts
                // at a multiple of 88 bytes, so the pieces are pairs of 64-bit reads there - ds_read2_b64 in the blocks, v2 loads here)
                constexpr int NQ = (WIN + 1) / 2, NLA = (NQ + 1) / 2;
                sg_f4 qa[NQ], qb[NQ];
                unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(const char *)xs;   // LDS byte address of step i's window
                if constexpr (TPL == 11) {
                    const v2 *xv = reinterpret_cast<const v2 *>(xs);
#pragma unroll
                    for (int j = 0; j < NQ; j++) { const v2 lo = xv[2 * j], hi = xv[2 * j + 1]; qa[j] = sg_f4{lo.x, lo.y, hi.x, hi.y}; }
                } else {
                    const sg_f4 *xq = reinterpret_cast<const sg_f4 *>(xs);
#pragma unroll
                    for (int j = 0; j < NQ; j++) qa[j] = xq[j];
                }
                // The steps are software-pipelined by half a step: block B of step i - 1 and block A of step i are ONE statement (seg_block_ba) wherever
                // the error function has the form d(y) y, so a group of 16 steps is  A | e B+A e B+A ... e B+A e | B  - (xp, yp, dp) carry step i - 1's samples
                // (padding taps masked), output and factor to it.  Other error functions: A, e, B per step.
                v2 xp[TPL], yp = v2{0, 0}, dp = v2{0, 0};
                auto fstep = [&](auto XO, sg_f4 (&q)[NQ], sg_f4 (&qn)[NQ], auto P0, auto NL_, auto BOFF, auto KU, Cx<R> pend, int gstep) __attribute__((always_inline)) -> Cx<R> {
                    constexpr int OFF = decltype(XO)::value, p0 = decltype(P0)::value, nl = decltype(NL_)::value, ku = decltype(KU)::value;
                    constexpr bool chk = decltype(CHK)::value != 0;
                    constexpr int NR = decltype(RG)::value;
                    v2 x[TPL];
#pragma unroll
                    for (int j = 0; j < TPL; j++) {
                        const sg_f4 v = q[(OFF + j) >> 1];
                        x[j] = ((OFF + j) & 1) ? v2{v.z, v.w} : v2{v.x, v.y};
                    }
                    v2 y, sq, xm0, xm1, xm2;
                    if constexpr (la_errfn_is_dy<METHOD> && ku >= 0) {
                        constexpr bool d1 = sizeof(decltype(seg_errfn_d<METHOD, NPART>(sq, K, Ks))) == sizeof(R);
                        seg_block_ba<TPL, d1, nl, NR, decltype(BOFF)::value + 16 * p0>(xp, yp, dp, x, w, y, sq, ebr, ebi, mks[ku >= 0 ? ku : 0], la,
                                                                                         qn[p0], qn[nl > 1 ? p0 + 1 : p0], qn[nl > 2 ? p0 + 2 : p0], qn[nl > 3 ? p0 + 3 : p0],
                                                                                         tailmask, xm0, xm1, xm2);
                    } else {
                        seg_block_a<TPL, (ku >= 0), nl, NR, decltype(BOFF)::value + 16 * p0>(x, w, y, sq, ebr, ebi, pend.re, pend.im, mks[ku >= 0 ? ku : 0], la,
                                                                                             qn[p0], qn[nl > 1 ? p0 + 1 : p0], qn[nl > 2 ? p0 + 2 : p0], qn[nl > 3 ? p0 + 3 : p0],
                                                                                             tailmask, xm0, xm1, xm2);
                    }
                    if constexpr (NR >= 1) x[TPL - 1] = xm0;                  // (the padding taps are updated with masked samples: they stay zero)
                    if constexpr (NR >= 2) x[TPL - 2] = xm1;
                    if constexpr (NR >= 3) x[TPL - 3] = xm2;
                    // c = mu e(y) and w += c conj(x)
                    v2 c1 = v2{0, 0};
                    if constexpr (la_errfn_is_dy<METHOD>) {
                        // (the same arithmetic with and without the end-of-segment test: which waves take which variant depends on how the chains of a
                        // launch fall into waves, and the result of a chain must not - a capture split over several processes trains other runs of chains)
                        auto d = seg_errfn_d<METHOD, NPART>(sq, K, Ks);
                        if constexpr (sizeof(d) == sizeof(R)) {
                            if (chk && gstep >= my_steps) d = 0;                    // past the end of this chain's segment: nothing moves
                            dp = __builtin_nondeterministic_value(dp); dp.x = d;
                        } else {
                            if (chk && gstep >= my_steps) d = v2{0, 0};
                            dp = d;
                        }
                        yp = y;                                                     // block B of this step: in front of the next step's block A, or behind the group
#pragma unroll
                        for (int j = 0; j < TPL; j++) xp[j] = x[j];
                    } else {
                        Cx<R> cc = la_errfn<R, METHOD, NPART, true>(Cx<R>{y.x, y.y}, K);
                        if (chk && gstep >= my_steps) cc = Cx<R>{0, 0};              // past the end of this chain's segment: nothing moves
                        c1 = v2{cc.re, cc.im};
                        seg_block_b<TPL>(x, w, c1, v2{cc.im, -cc.re});
                    }
                    return Cx<R>{c1.x, c1.y};
                };
                // block B of the last step of a group (error functions of the form d(y) y); returns its c = mu e
                auto last_b = [&]() __attribute__((always_inline)) -> Cx<R> {
                    constexpr bool d1 = sizeof(decltype(seg_errfn_d<METHOD, NPART>(yp, K, Ks))) == sizeof(R);
                    const v2 c = seg_block_b2<TPL, d1>(xp, w, yp, dp);
                    return Cx<R>{c.x, c.y};
                };
                for (; i + LPC <= nst; i += LPC) {
                    Cx<R> pend{0, 0};
                    auto quad = [&](auto U) __attribute__((always_inline)) {
                        constexpr int u = decltype(U)::value;
                        const int g = ibase + i + u;
                        pend = fstep(SgInt<0>{}, qa, qb, SgInt<0>{}, SgInt<NLA>{}, SgInt<(u + 2) * 16>{}, SgInt<u - 1>{}, pend, g);
                        pend = fstep(SgInt<2>{}, qa, qb, SgInt<NLA>{}, SgInt<NQ - NLA>{}, SgInt<(u + 2) * 16>{}, SgInt<u>{}, pend, g + 1);
                        sg_wait_lds(qb);
                        pend = fstep(SgInt<0>{}, qb, qa, SgInt<0>{}, SgInt<NLA>{}, SgInt<(u + 4) * 16>{}, SgInt<u + 1>{}, pend, g + 2);    // (may look past the chunk: inside the row's slack)
                        pend = fstep(SgInt<2>{}, qb, qa, SgInt<NLA>{}, SgInt<NQ - NLA>{}, SgInt<(u + 4) * 16>{}, SgInt<u + 2>{}, pend, g + 3);
                        sg_wait_lds(qa);
                    };
                    quad(SgInt<0>{}); quad(SgInt<4>{});
                    if constexpr (LPC == 16) { quad(SgInt<8>{}); quad(SgInt<12>{}); }
                    if constexpr (la_errfn_is_dy<METHOD>) pend = last_b();
                    keep(pend, LPC - 1);
                    const int gi = ibase + i + l16;
                    if (gi < my_steps) stg(errow + gi, Cx<R>{ebr * inv_mu, ebi * inv_mu});
                    la += LPC * 16;                                 // 16 steps x 2 samples x 8 bytes
                }
            }
        } else
        if (os2) {
            // 2 samples per symbol: steps i and i + 1 read x[2 i .. 2 i + TPL + 2) - ONE window of TPL + 2 samples serves both (13 LDS
            // words instead of 22 at 11 taps per lane); the window of the next pair is read while this one computes
            load_x(xa, xs, SgInt<WIN>{});
            for (; i + LPC <= nst; i += LPC) {
#pragma unroll
                for (int u = 0; u < LPC; u += 4) {
                    load_x(xb, xs + (i + u + 2) * xstep, SgInt<WIN>{});
                    keep(step(SgInt<0>{}, xa, ibase + i + u, CHK, RG), u);
                    keep(step(SgInt<2>{}, xa, ibase + i + u + 1, CHK, RG), u + 1);
                    load_x(xa, xs + (i + u + 4) * xstep, SgInt<WIN>{});       // may look past the chunk: inside the row's slack
                    keep(step(SgInt<0>{}, xb, ibase + i + u + 2, CHK, RG), u + 2);
                    keep(step(SgInt<2>{}, xb, ibase + i + u + 3, CHK, RG), u + 3);
                }
                const int gi = ibase + i + l16;
                if (gi < my_steps) stg(errow + gi, Cx<R>{ebr * inv_mu, ebi * inv_mu});
            }
        }
        // (other sampling rates, and the ragged end of the last chunk, take the plain loop below)
        // ---- step by step: LPC errors per chain staged, then one store per chain
        for (int i0 = i; i0 < nst; i0 += LPC) {
            const int ng = nst - i0 < LPC ? nst - i0 : LPC;
#pragma unroll 1
            for (int u = 0; u < ng; u++) {
                load_x(xa, xs + (i0 + u) * xstep, SgInt<TPL>{});
                keep(step(SgInt<0>{}, xa, ibase + i0 + u, CHK, RG), u);
            }
            const int gi = ibase + i0 + l16;
            if (l16 < ng && gi < my_steps) stg(errow + gi, Cx<R>{ebr * inv_mu, ebi * inv_mu});
        }
    };

    const int nchunk = (max_steps + CH - 1) / CH;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) stage_load(c + 1);                  // in flight while this chunk computes
        const Cx<R> *xs = has ? lds + (slot * a.nmodes + kin) * rowsz + t0 : zero_row;
        const int xstep = has ? os_ : 0;
        const int ibase = c * CH;
        const int nst = (max_steps - ibase) < CH ? (max_steps - ibase) : CH;
        auto run2 = [&](auto RG, auto OS2) __attribute__((always_inline)) {
            if (ibase + nst <= min_steps) run_chunk(xs, xstep, ibase, nst, SgInt<0>{}, RG, OS2);
            else run_chunk(xs, xstep, ibase, nst, SgInt<1>{}, RG, OS2);
        };
        auto run = [&](auto RG) __attribute__((always_inline)) {   // (wave-uniform)
            if (os_ == 2) run2(RG, SgInt<1>{}); else run2(RG, SgInt<0>{});
        };
        switch (rag) {                                         // wave-uniform: the loops exist once per padding count
        case 0: run(SgInt<0>{}); break;
        case 1: run(SgInt<1>{}); break;
        case 2: run(SgInt<(TPL > 2 ? 2 : 0)>{}); break;
        default: run(SgInt<(TPL > 3 ? 3 : 0)>{}); break;
        }
        __syncthreads();                                         // (compiler fence: no store of the next chunk above a read of this one)
        if (c + 1 < nchunk) stage_store(c + 1);                  // same buffer: its readers (this wave) are done with chunk c
        __syncthreads();
    }
    if (alive) {
#pragma unroll
        for (int j = 0; j < TPL; j++)
            if (has && t0 + j < a.ntaps) stg(wrow + kin * a.ntaps + t0 + j, Cx<R>{w[j].x, w[j].y});
        if constexpr (ADAPT) if (l16 == 0) { a.r_out[q] = ad_r; a.e_out[q] = ad_ep; a.mu_sum[q] = ad_sum; }
    }
}

#endif  // QH_SEG_KERNELS

// ------------------------------------------------------------------------------------------------ host side
// taps per lane for `lpc` lanes per chain (0: no layout).  16 lanes per chain is the general form; 8 lanes per chain (8 chains
// per wave, instantiated for single precision and 6 / 11 taps per lane) issues 0.64 x the instructions per chain and step and
// is what launches with many chains use (see launch_seg).
inline int seg_tpl(int nmodes, int ntaps, int lpc = 16)
{
    for (int tpl : {2, 4, 6, 8, 11}) {
        if (lpc == 16 ? tpl == 11 : (tpl != 6 && tpl != 11)) continue;
        const int lpm = (ntaps + tpl - 1) / tpl;
        if (nmodes * lpm <= lpc && lpm * tpl - ntaps <= SG_MAXRAG && lpm * tpl - ntaps < tpl) return tpl;
    }
    return 0;
}
inline int seg_slots(int nsel, int cpw = 4)                         // distinct segments among the chains of a wave
{
    int n = 1;
    for (int q0 = 0; q0 <= cpw * nsel; q0 += cpw) { const int m = (q0 + cpw - 1) / nsel - q0 / nsel + 1; n = m > n ? m : n; }
    return n;
}
inline bool seg_supported(int method, int nmodes, int ntaps, int os, int64_t nsy, size_t elem, int nsel = 1, int rows = 4)
{
    (void)elem;
    if (seg_tpl(nmodes, ntaps) == 0) return false;
    if ((SG_CH + 4) * os + ntaps + 8 > SG_PITCH && os == 2) return false;      // chunk + the look-ahead of the window pairs + padding taps fit a row
    if ((SG_CH + 2) * os + ntaps + 8 > SG_PITCH) return false;
    if (seg_slots(nsel) * nmodes > rows) return false;
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: case QH_M_CMA2: case QH_M_MCMA: return true;
    case QH_M_RDE: case QH_M_MRDE: return nsy - (nsy + 1) / 2 >= 1 && nsy - (nsy + 1) / 2 <= LA_MAXPART;
    case QH_M_SBD: case QH_M_MDDMA: case QH_M_DD: return true;      // square alphabets through slicer tables (caller checks)
    default: return false;
    }
}
constexpr int SG_LPC8_MIN = 4096 + 1;    // chains from which 8 lanes per chain pay: as long as 4 chains per wave fit one wave per SIMD (1024 SIMDs), the
                                         // shorter instruction stream of the 16-lane layout wins (C3, 3840 chains: 275 against 390 us per pass)
template <typename R> inline int seg_lanes(int nmodes, int ntaps, int nsel, int nq)
{
    const int e = form(FORM_SEG_LANES);                             // qh_set_form("seg_lanes", "8" | "16"): force (measurements, tests)
    const bool can8 = sizeof(R) == 4 && seg_tpl(nmodes, ntaps, 8) != 0 && seg_slots(nsel, 8) * nmodes <= 8;
    if (e == 16) return 16;
    if (e == 8) return can8 ? 8 : 16;
    return can8 && nq >= SG_LPC8_MIN ? 8 : 16;
}

#ifdef QH_SEG_KERNELS
template <typename R, int METHOD, int NPART> static int launch_seg_tpl(const SegArgs<R> &a, int tpl, int lpc, dim3 grid, size_t lds)
{
    if (lpc == 8) {
        if constexpr (sizeof(R) == 4) {
            switch (tpl) {
            case 6: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 6, 8>), grid, dim3(64), lds, g_stream, a); return QH_OK;
            case 11: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 11, 8>), grid, dim3(64), lds, g_stream, a); return QH_OK;
            default: break;
            }
        }
        set_error("segment trainer: unsupported tap layout"); return QH_ERR_ARG;
    }
    switch (tpl) {
    case 2: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 2, 16>), grid, dim3(64), lds, g_stream, a); break;
    case 4: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 4, 16>), grid, dim3(64), lds, g_stream, a); break;
    case 6: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 6, 16>), grid, dim3(64), lds, g_stream, a); break;
    case 8: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 8, 16>), grid, dim3(64), lds, g_stream, a); break;
    default: set_error("segment trainer: unsupported tap layout"); return QH_ERR_ARG;
    }
    return QH_OK;
}
template <typename R, int METHOD> static int launch_seg_parts(const SegArgs<R> &a, int npart, int tpl, int lpc, dim3 grid, size_t lds)
{
#define QH_SG_NP(N) case N: return launch_seg_tpl<R, METHOD, N>(a, tpl, lpc, grid, lds);
    switch (npart) {
        QH_SG_NP(1) QH_SG_NP(2) QH_SG_NP(3) QH_SG_NP(4) QH_SG_NP(5) QH_SG_NP(6) QH_SG_NP(7) QH_SG_NP(8)
    default: set_error("segment trainer: unsupported partition count"); return QH_ERR_ARG;
    }
#undef QH_SG_NP
}
template <typename R, int METHOD> static int launch_seg_dd(const SegArgs<R> &a, int npart, int tpl, int lpc, dim3 grid, size_t lds)
{
    switch (npart) {            // 4-, 16-, 64-, 256-QAM
    case 1: return launch_seg_tpl<R, METHOD, 1>(a, tpl, lpc, grid, lds);
    case 3: return launch_seg_tpl<R, METHOD, 3>(a, tpl, lpc, grid, lds);
    case 7: return launch_seg_tpl<R, METHOD, 7>(a, tpl, lpc, grid, lds);
    case 15: return launch_seg_tpl<R, METHOD, 15>(a, tpl, lpc, grid, lds);
    default: set_error("segment trainer: unsupported slicer size"); return QH_ERR_ARG;
    }
}

// the adaptive-step kernels (single precision, 16 lanes per chain): translation units train_seg_<method>_f32_ad.hip
template <typename R, int METHOD, int NPART> static int launch_seg_tpl_ad(const SegArgs<R> &a, int tpl, dim3 grid, size_t lds)
{
    switch (tpl) {
    case 2: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 2, 16, true>), grid, dim3(64), lds, g_stream, a); break;
    case 4: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 4, 16, true>), grid, dim3(64), lds, g_stream, a); break;
    case 6: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 6, 16, true>), grid, dim3(64), lds, g_stream, a); break;
    case 8: hipLaunchKernelGGL((train_seg_kernel<R, METHOD, NPART, 8, 16, true>), grid, dim3(64), lds, g_stream, a); break;
    default: set_error("segment trainer: unsupported tap layout"); return QH_ERR_ARG;
    }
    return QH_OK;
}
template <typename R, int METHOD> int launch_seg_ad(const SegArgs<R> &a, int npart, int tpl, dim3 grid, size_t lds)
{
    if constexpr (METHOD == QH_M_SBD || METHOD == QH_M_MDDMA || METHOD == QH_M_DD) {
        switch (npart) {            // 4-, 16-, 64-, 256-QAM
        case 1: return launch_seg_tpl_ad<R, METHOD, 1>(a, tpl, grid, lds);
        case 3: return launch_seg_tpl_ad<R, METHOD, 3>(a, tpl, grid, lds);
        case 7: return launch_seg_tpl_ad<R, METHOD, 7>(a, tpl, grid, lds);
        case 15: return launch_seg_tpl_ad<R, METHOD, 15>(a, tpl, grid, lds);
        default: set_error("segment trainer: unsupported slicer size"); return QH_ERR_ARG;
        }
    } else return launch_seg_tpl_ad<R, METHOD, 0>(a, tpl, grid, lds);
}

// one launcher per error function (translation units train_seg_<method>_{f32,f64}.hip instantiate them: the kernels' main loops are
// unrolled eight or sixteen steps deep in four padding x two end-of-segment variants - build time is spread over many units)
template <typename R, int METHOD> int launch_seg_m(const SegArgs<R> &a, int npart, int tpl, int lpc, dim3 grid, size_t lds)
{
    if constexpr (METHOD == QH_M_RDE || METHOD == QH_M_MRDE) return launch_seg_parts<R, METHOD>(a, npart, tpl, lpc, grid, lds);
    else if constexpr (METHOD == QH_M_SBD || METHOD == QH_M_MDDMA || METHOD == QH_M_DD) return launch_seg_dd<R, METHOD>(a, npart, tpl, lpc, grid, lds);
    else return launch_seg_tpl<R, METHOD, 0>(a, tpl, lpc, grid, lds);
}
#else
template <typename R, int METHOD> int launch_seg_ad(const SegArgs<R> &a, int npart, int tpl, dim3 grid, size_t lds);
extern template int launch_seg_ad<float, QH_M_CMA>(const SegArgs<float> &, int, int, dim3, size_t);
extern template int launch_seg_ad<float, QH_M_MCMA>(const SegArgs<float> &, int, int, dim3, size_t);
extern template int launch_seg_ad<float, QH_M_MDDMA>(const SegArgs<float> &, int, int, dim3, size_t);
extern template int launch_seg_ad<float, QH_M_SBD>(const SegArgs<float> &, int, int, dim3, size_t);
template <typename R, int METHOD> int launch_seg_m(const SegArgs<R> &a, int npart, int tpl, int lpc, dim3 grid, size_t lds);
#define QH_SEG_EXTERN(M) \
    extern template int launch_seg_m<float, M>(const SegArgs<float> &, int, int, int, dim3, size_t); \
    extern template int launch_seg_m<double, M>(const SegArgs<double> &, int, int, int, dim3, size_t);
QH_SEG_EXTERN(QH_M_CMA) QH_SEG_EXTERN(QH_M_CMA2) QH_SEG_EXTERN(QH_M_MCMA) QH_SEG_EXTERN(QH_M_RDE) QH_SEG_EXTERN(QH_M_MRDE)
QH_SEG_EXTERN(QH_M_SBD) QH_SEG_EXTERN(QH_M_MDDMA) QH_SEG_EXTERN(QH_M_DD)
#undef QH_SEG_EXTERN
#endif  // QH_SEG_KERNELS

// `a` complete except lpm / pitch / rag; method-specific table layout as for launch_bi (slicer tables for sbd / mddma / dd)
inline bool seg_adaptive_supported(int method) { return method == QH_M_CMA || method == QH_M_SGNCMA || method == QH_M_MCMA || method == QH_M_MDDMA || method == QH_M_SBD; }
template <typename R> int launch_seg(SegArgs<R> a, int method, bool adaptive = false)
{
    const int nq = a.q_count > 0 ? a.q_count : a.S * a.nsel;
    if (a.q_count <= 0) a.q_first = 0;
    const int lpc = adaptive ? 16 : seg_lanes<R>(a.nmodes, a.ntaps, a.nsel, nq), cpw = 64 / lpc;
    const int tpl = seg_tpl(a.nmodes, a.ntaps, lpc);
    a.lpm = (a.ntaps + tpl - 1) / tpl;
    a.rag = a.lpm * tpl - a.ntaps;
    const bool long_rows = sizeof(R) == 4 && lpc == 16 && !adaptive;          // SgRow<R, LPC, ADAPT>::pitch
    a.pitch = long_rows ? SG_PITCH_LONG : SG_PITCH;
    a.ch = (long_rows && (SG_CH_LONG + 4) * a.os + a.ntaps + 8 <= SG_PITCH_LONG) ? SG_CH_LONG : SG_CH;
    a.nslots = seg_slots(a.nsel, cpw);
    const size_t lds = (size_t)((adaptive ? 2 * cpw : cpw) + 1) * a.pitch * sizeof(Cx<R>);
    dim3 grid((nq + cpw - 1) / cpw);
    const int npart = (int)(a.nsy - (a.nsy + 1) / 2);
    int rc;
    if (adaptive) {
        if constexpr (sizeof(R) == 4) {
            switch (method) {
            case QH_M_CMA: case QH_M_SGNCMA: rc = launch_seg_ad<R, QH_M_CMA>(a, npart, tpl, grid, lds); break;
            case QH_M_MCMA: rc = launch_seg_ad<R, QH_M_MCMA>(a, npart, tpl, grid, lds); break;
            case QH_M_MDDMA: rc = launch_seg_ad<R, QH_M_MDDMA>(a, npart, tpl, grid, lds); break;
            case QH_M_SBD: rc = launch_seg_ad<R, QH_M_SBD>(a, npart, tpl, grid, lds); break;
            default: rc = QH_ERR_METHOD;
            }
        } else rc = QH_ERR_ARG;
        if (rc) return rc;
        QH_HIP(hipGetLastError());
        return QH_OK;
    }
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: rc = launch_seg_m<R, QH_M_CMA>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_CMA2: rc = launch_seg_m<R, QH_M_CMA2>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_MCMA: rc = launch_seg_m<R, QH_M_MCMA>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_RDE: rc = launch_seg_m<R, QH_M_RDE>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_MRDE: rc = launch_seg_m<R, QH_M_MRDE>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_SBD: rc = launch_seg_m<R, QH_M_SBD>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_MDDMA: rc = launch_seg_m<R, QH_M_MDDMA>(a, npart, tpl, lpc, grid, lds); break;
    case QH_M_DD: rc = launch_seg_m<R, QH_M_DD>(a, npart, tpl, lpc, grid, lds); break;
    default: rc = QH_ERR_METHOD;
    }
    if (rc) return rc;
    QH_HIP(hipGetLastError());
    return QH_OK;
}

}  // namespace qh
