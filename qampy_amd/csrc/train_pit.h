// Parallel-in-time training ("tier B", opt-in; include/qampy_hip.h): the sweep
//        w_{i+1} = w_i + mu errfn(w_i . x_i) conj(x_i) ,   i = 0 .. TrSyms-1     (pythran_equalisation.py:165-172)
// is cut into S contiguous segments that the exact kernels train concurrently (one workgroup per segment and output
// mode: the segments are the channels of a batch that happen to be adjacent in one capture) and that waveform relaxation
// makes consistent: pass p starts segment s from the end taps of segment s-1 in pass p-1.  What is sequential is
//   * a gear-shifted acquisition on a prefix when the start taps are not converged (cold start), and
//   * the number of passes (each costs one segment length of dependent steps),
// and both end on device-side criteria (error plateau; boundary defect), so the host only enqueues.
//
// Buffers per call (scratch slot 2):  X (S tap sets) start taps of the current pass,  Y (S tap sets) trained in place,
// rot (S x nsel) pass-0 phase rotations, z (S x nsel) 4th-power sums, dfc (S x nsel) boundary defects.
#pragma once
#include <atomic>
#include <chrono>
#include <vector>
#include "train_impl.h"
#include "train_seg.h"

namespace qh {

typedef qh_pit_report PitCtrl;

constexpr double PIT_CONTRACT = 0.2; // expected defect ratio of consecutive passes with the coarse correction (measured 0.1 .. 0.4)
constexpr int PIT_PROBE = 128;      // symbol periods of the capture a boundary defect is measured on
constexpr int PIT_SEEDWIN = 512;
constexpr int PIT_MAXSEG = 65536;  // segments of a sweep (was 4096: the 10^7-symbol capture already sat on it)    // symbol periods of a segment's head the phase seed is estimated on

// symmetry group of an error function: errfn(g y) = g errfn(y).  0: every phase (cma, rde); n: the n-th roots of unity
__host__ __device__ inline int pit_symmetry(int method)
{
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: case QH_M_RDE: return 0;
    case QH_M_CMA2: return 2;
    default: return 4;               // mcma, mrde, sbd, mddma, dd on the 4-fold symmetric QAM alphabets
    }
}

struct PitSeg {                      // segment grid of a sweep: see LaArgs::seg
    int S;
    int64_t len, extra, tail;
    int64_t begin = 0;               // first step of segment 0 (adaptive step: the head of the sweep runs in the exact form)
    __host__ __device__ int64_t start(int64_t s) const { return begin + s * len + (s < extra ? s : extra) * 64; }
    __host__ __device__ int64_t steps(int64_t s) const { return len + (s < extra ? 64 : 0) + (s == S - 1 ? tail : 0); }
};

// in-place inclusive prefix sum of a[0..n) in LDS by a 256-thread block: chunk sums, one serial pass over the 256 chunk
// totals, chunk-local prefixes (n is a few thousand: ~1 us instead of a serial loop over n)
// Exclusive scan over the NT threads of a block (thread order) with an associative op(earlier, later) and its identity; buf: LDS,
// [NT / 64] of T.  Inclusive scan inside each wave with shuffles, the wave totals through LDS: two barriers (the Hillis-Steele
// version on ping-pong buffers took log2(NT) of them - 2-3 us per call at 1024 threads; the one before that ended in ONE thread
// walking all partial results).
template <typename T> __device__ __forceinline__ T shfl_up_any(T v, int d)
{
    static_assert(sizeof(T) % 4 == 0, "shfl_up_any: 32-bit words");
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = __shfl_up(w[i], d);
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}
template <typename T, typename Op, int NT = 256> __device__ __forceinline__ T block_scan_excl(T v, Op op, T ident, T *buf)      // buf: [NT / 64]
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    T x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T y = shfl_up_any(x, o);
        if (lane >= o) x = op(y, x);
    }
    if (lane == 63) buf[wave] = x;
    __syncthreads();
    T pre = ident;
    for (int w = 0; w < wave; w++) pre = op(pre, buf[w]);
    T ex = shfl_up_any(x, 1);
    if (lane == 0) ex = ident;
    const T r = op(pre, ex);
    __syncthreads();
    return r;
}

// inclusive prefix sums of a[0..n) in place (a in LDS, 256 threads, tot: [512] scratch in LDS)
template <typename T> __device__ __forceinline__ void block_prefix_sum(T *a, int n, T *tot)
{
    const int len = (n + 255) / 256;
    const int i0 = threadIdx.x * len, i1 = i0 + len < n ? i0 + len : n;
    T s = 0;
    for (int i = i0; i < i1; i++) s += a[i];
    T run = block_scan_excl<T>(s, [](T x, T y) { return x + y; }, (T)0, tot);
    for (int i = i0; i < i1; i++) { run += a[i]; a[i] = run; }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ control kernels
// power of the capture, gear-shifted step size, report header
template <typename R>
__global__ void __launch_bounds__(256) pit_setup_kernel(const Cx<R> *E, int nmodes, int64_t L, int64_t npow, int ntot, const R *mu, double gear,
                                                        double bound, double tol, PitSeg sg, PitCtrl *c, R *mu_acq)
{
    QH_WAVE_FIRST();
    __shared__ double red[256];
    double acc = 0;
    for (int k = 0; k < nmodes; k++)
        for (int64_t i = threadIdx.x; i < npow; i += 256) {
            const Cx<R> v = E[(size_t)k * L + i];
            acc += (double)v.re * v.re + (double)v.im * v.im;
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double p = red[0] / ((double)nmodes * (double)npow);
        const double m = (double)*mu;
        double ma = gear * m;
        const double cap = bound / ((double)ntot * (p > 1e-30 ? p : 1e-30));
        if (ma > cap) ma = cap;
        if (!(ma > m)) ma = m;
        c->segments = sg.S; c->seg_len = sg.len; c->mu = m; c->mu_acq = ma; c->power = p; c->tol = tol;
        for (int q = 0; q < QH_PIT_MAXPASS; q++) { c->result_change[q] = -1; c->deviation[q] = -1; c->deviation_rms[q] = -1; c->deviation_taps[q] = -1; c->deviation_taps_worst[q] = -1; } c->passes = 0; c->converged = 0; c->acq_chunks = 0; c->acq_steps = 0; c->acq_done = 0; c->done = 0; c->diverged = 0; c->corr_on = 0; c->gain = 0; c->out_power = 0;
        for (int i = 0; i < QH_PIT_MAXPASS; i++) c->defect[i] = -1;
        for (int i = 0; i < QH_PIT_MAXCHUNK; i++) c->acq_err[i] = -1;
        *mu_acq = (R)ma;
    }
}

// after an acquisition chunk: mean |err|^2 over the chunk -> plateau / divergence test
template <typename R>
__global__ void __launch_bounds__(256) pit_acq_monitor_kernel(const Cx<R> *err, int64_t err_pitch, int64_t step0, int64_t n, int nsel,
                                                              const int64_t *modes_dev, double plateau, PitCtrl *c, R *mu_acq, const R *mu, double floor_gear)
{
    QH_WAVE_FIRST();
    // gear-down between the acquisition chunks: the next chunk runs at half the step, never below floor_gear x mu (floor_gear < 0: every
    // chunk at the gear-shifted step).  Nothing in this kernel reads mu_acq; the chunk behind it does.  (Was a launch of its own.)
    if (threadIdx.x == 0 && floor_gear >= 0) {
        const R lo = (R)((double)*mu * floor_gear), h = *mu_acq * (R)0.5;
        *mu_acq = h > lo ? h : (*mu_acq > lo ? lo : *mu_acq);
    }
    if (c->acq_done) return;
    __shared__ double red[256];
    double acc = 0;
    for (int j = 0; j < nsel; j++) {
        const Cx<R> *row = err + (size_t)modes_dev[j] * err_pitch + step0;
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const Cx<R> v = row[i];
            acc += (double)v.re * v.re + (double)v.im * v.im;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v = red[0] / ((double)n * nsel);
        const int k = c->acq_chunks;
        if (k < QH_PIT_MAXCHUNK) c->acq_err[k] = v;
        c->acq_chunks = k + 1;
        c->acq_steps = step0 + n;
        if (!(v == v) || v > 1e30) { c->diverged = 1; c->acq_done = 1; return; }
        if (k >= 1) {
            const double prev = c->acq_err[k - 1 < QH_PIT_MAXCHUNK ? k - 1 : QH_PIT_MAXCHUNK - 1];
            if (v > 100 * c->acq_err[0]) { c->diverged = 1; c->acq_done = 1; return; }
            if (v > plateau * prev) c->acq_done = 1;
        }
    }
}

// a diverged acquisition (step size too bold for this capture) is undone: the sweep then starts from the original taps
template <typename R> __global__ void pit_acq_finish_kernel(Cx<R> *wx, const Cx<R> *w_start, int n, PitCtrl *c)
{
    QH_WAVE_FIRST();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (c->diverged && i < n) wx[i] = w_start[i];
}

// z[s, j] = sum over the head of segment s of (w[mode_j] . x_i)^4.  The head's samples are staged in LDS once, split into the
// `os` sampling phases (plane r holds x[j os + r]), so that for a fixed tap the lanes - consecutive outputs - read consecutive
// words; the taps are broadcast reads.  (The former version fetched every operand of every product from global memory: 2 x the
// time of the full-capture filter for half its outputs.)
inline size_t pit_phase_pitch(int ntaps, int os, int nwin) { return (size_t)nwin + (size_t)(ntaps + os - 1) / os + 1; }
template <typename R>
__global__ void __launch_bounds__(256) pit_phase_kernel(const Cx<R> *E, int nmodes, int64_t L, int os, const Cx<R> *wx, int ntaps, PitSeg sg,
                                                        const int64_t *modes_dev, int nwin, double *z)
{
    QH_WAVE_FIRST();
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    Cx<R> *w = reinterpret_cast<Cx<R> *>(pit_smem);
    const int ntot = nmodes * ntaps;
    const int pitch = nwin + (ntaps + os - 1) / os + 1;
    Cx<R> *xs = w + ntot;                                         // [nmodes][os][pitch]
    __shared__ double rr[256], ri[256];
    const int s = blockIdx.x, j = blockIdx.y;
    const int mode = (int)modes_dev[j];
    for (int f = threadIdx.x; f < ntot; f += 256) w[f] = wx[(size_t)mode * ntot + f];
    const int64_t st = sg.start(s);
    int64_t n = sg.steps(s);
    if (n > nwin) n = nwin;
    const int ns = n > 0 ? (int)(n - 1) * os + ntaps : 0;         // samples of the head per input mode
    for (int k = 0; k < nmodes; k++) {
        const Cx<R> *x = E + (size_t)k * L + st * os;
        for (int g = threadIdx.x; g < ns; g += 256) xs[((size_t)k * os + g % os) * pitch + g / os] = x[g];
    }
    __syncthreads();
    double zr = 0, zi = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        R yr = 0, yi = 0;
        for (int k = 0; k < nmodes; k++) {
            const Cx<R> *wk = w + k * ntaps;
            for (int r = 0; r < os; r++) {
                const Cx<R> *xp = xs + ((size_t)k * os + r) * pitch + i;
                int u = 0;
                for (int t = r; t < ntaps; t += os, u++) {
                    const Cx<R> a = xp[u], b = wk[t];
                    yr = fma_(a.re, b.re, fma_(-a.im, b.im, yr));
                    yi = fma_(a.re, b.im, fma_(a.im, b.re, yi));
                }
            }
        }
        const double y2r = (double)yr * yr - (double)yi * yi, y2i = 2.0 * yr * yi;
        zr += y2r * y2r - y2i * y2i;
        zi += 2.0 * y2r * y2i;
    }
    rr[threadIdx.x] = zr; ri[threadIdx.x] = zi;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if (threadIdx.x < q) { rr[threadIdx.x] += rr[threadIdx.x + q]; ri[threadIdx.x] += ri[threadIdx.x + q]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { z[2 * ((size_t)s * gridDim.y + j)] = rr[0]; z[2 * ((size_t)s * gridDim.y + j) + 1] = ri[0]; }
}

// rot[s, j] = exp(-i phi_s): phi = arg(-z)/4 (E[s^4] is negative real for the QAM alphabets), unwrapped along the segments
static __global__ void __launch_bounds__(256) pit_unwrap_kernel(const double *z, int S, int nsel, double *rot, double *phi, int *jump)
{
    QH_WAVE_FIRST();
    // phi[nsel][S], jump[nsel][S]: work arrays in device memory (S may be tens of thousands); one block per mode
    __shared__ int ptot[512];
    const double q = 1.5707963267948966;
    phi += (size_t)blockIdx.x * S; jump += (size_t)blockIdx.x * S;
    for (int j = blockIdx.x; j <= (int)blockIdx.x; j++) {
        for (int s = threadIdx.x; s < S; s += 256) {
            const double zr = z[2 * ((size_t)s * nsel + j)], zi = z[2 * ((size_t)s * nsel + j) + 1];
            phi[s] = (zr == 0 && zi == 0) ? 0.0 : (double)atan2f((float)-zi, (float)-zr) / 4;     // a seed: single precision is plenty
        }
        __syncthreads();
        for (int s = threadIdx.x; s < S; s += 256) jump[s] = s == 0 ? 0 : (int)rint((phi[s - 1] - phi[s]) / q);
        __syncthreads();
        block_prefix_sum<int>(jump, S, ptot);
        for (int s = threadIdx.x; s < S; s += 256) {
            const double p = phi[s] + q * (jump[s] & 3);
            float sn, cs;
            sincosf((float)p, &sn, &cs);
            rot[2 * ((size_t)s * nsel + j)] = cs;
            rot[2 * ((size_t)s * nsel + j) + 1] = -sn;
        }
        __syncthreads();
    }
}

// X[s] = wx with the rows of the selected modes rotated by rot[s, j] (rot == nullptr: plain copies).  w0 != nullptr: segment 0
// starts from w0 as it is - the taps the sweep starts from in the reference - whatever seeds the other segments get.
template <typename R>
__global__ void __launch_bounds__(256) pit_seed_kernel(const Cx<R> *wx, int nmodes, int ntot, const int64_t *modes_dev, int nsel, const double *rot, Cx<R> *X,
                                                       const Cx<R> *w0, Cx<R> *Y, PitCtrl *c)
{
    QH_WAVE_FIRST();
    const int s = blockIdx.x;
    const int n = nmodes * ntot;
    if (s == 0 && c) {                                            // a new sweep: the control block's per-sweep fields (was a launch of its own)
        if (threadIdx.x == 0) { c->done = 0; c->converged = 0; c->passes = 0; }
        for (int q = threadIdx.x; q < QH_PIT_MAXPASS; q += 256) {
            c->result_change[q] = -1; c->deviation[q] = -1; c->deviation_rms[q] = -1; c->deviation_taps[q] = -1; c->deviation_taps_worst[q] = -1; c->defect[q] = -1;
        }
    }
    const bool exact0 = s == 0 && w0 != nullptr;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int row = e / ntot;
        Cx<R> v = exact0 ? w0[e] : wx[e];
        if (rot && !exact0) {
            for (int j = 0; j < nsel; j++)
                if ((int)modes_dev[j] == row) {
                    const double c = rot[2 * ((size_t)s * nsel + j)], d = rot[2 * ((size_t)s * nsel + j) + 1];
                    const double vr = (double)v.re * c - (double)v.im * d, vi = (double)v.re * d + (double)v.im * c;
                    v = Cx<R>{(R)vr, (R)vi};
                    break;
                }
        }
        X[(size_t)s * n + e] = v;
        Y[(size_t)s * n + e] = v;                                 // pass 0 trains Y in place (was a copy of X before the launch)
    }
}

// defect of boundary b = blockIdx.x + 1: outputs of the start taps X[b] against the end taps Y[b-1] of the left neighbour
// on PIT_PROBE symbol periods from the boundary on, modulo the symmetry of the error function
template <typename R>
__global__ void __launch_bounds__(PIT_PROBE) pit_defect_kernel(const Cx<R> *E, int nmodes, int64_t L, int os, int ntaps, PitSeg sg, int64_t TrSyms,
                                                                const int64_t *modes_dev, const Cx<R> *X, const Cx<R> *Y, int sym, const PitCtrl *c, double *dfc, double *pw, double *gph,
                                                                const Cx<R> *wx_prev)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    Cx<R> *wa = reinterpret_cast<Cx<R> *>(pit_smem);
    const int ntot = nmodes * ntaps;
    Cx<R> *wb = wa + ntot;
    const int pitch = PIT_PROBE + (ntaps + os - 1) / os + 1;
    Cx<R> *xs = wb + ntot;                                        // [nmodes][os][pitch]: the probe's samples by sampling phase (as in pit_phase_kernel)
    __shared__ double red[4][PIT_PROBE];
    const int b = blockIdx.x + 1, j = blockIdx.y;
    const int mode = (int)modes_dev[j];
    const size_t wset = (size_t)nmodes * ntot;
    // b == S (one block row more than there are boundaries): the sweep's RESULT - end taps of the last segment - against the
    // result of the previous pass (wx_prev), probed on the last PIT_PROBE steps of the sweep: how far the output still moved
    const bool result_probe = b == sg.S;
    for (int f = threadIdx.x; f < ntot; f += PIT_PROBE) {
        wa[f] = result_probe ? wx_prev[(size_t)mode * ntot + f] : X[(size_t)b * wset + (size_t)mode * ntot + f];
        wb[f] = Y[(size_t)(b - 1) * wset + (size_t)mode * ntot + f];
    }
    const int64_t st0 = result_probe ? (TrSyms > PIT_PROBE ? TrSyms - PIT_PROBE : 0) : sg.start(b);
    int64_t nout = TrSyms - st0;
    if (nout > PIT_PROBE) nout = PIT_PROBE;
    const int ns = nout > 0 ? (int)(nout - 1) * os + ntaps : 0;
    for (int k = 0; k < nmodes; k++) {
        const Cx<R> *x = E + (size_t)k * L + st0 * os;
        for (int g = threadIdx.x; g < ns; g += PIT_PROBE) xs[((size_t)k * os + g % os) * pitch + g / os] = x[g];
    }
    __syncthreads();
    double aa = 0, bb = 0, cr = 0, ci = 0;
    if (threadIdx.x < nout) {
        R ar = 0, ai = 0, br = 0, bi = 0;
        for (int k = 0; k < nmodes; k++)
            for (int r = 0; r < os; r++) {
                const Cx<R> *xp = xs + ((size_t)k * os + r) * pitch + threadIdx.x;
                int u = 0;
                for (int t = r; t < ntaps; t += os, u++) {
                    const Cx<R> v = xp[u], p = wa[k * ntaps + t], q = wb[k * ntaps + t];
                    ar = fma_(v.re, p.re, fma_(-v.im, p.im, ar)); ai = fma_(v.re, p.im, fma_(v.im, p.re, ai));
                    br = fma_(v.re, q.re, fma_(-v.im, q.im, br)); bi = fma_(v.re, q.im, fma_(v.im, q.re, bi));
                }
            }
        aa = (double)ar * ar + (double)ai * ai;
        bb = (double)br * br + (double)bi * bi;
        cr = (double)br * ar + (double)bi * ai;          // yB conj(yA)
        ci = (double)bi * ar - (double)br * ai;
    }
    red[0][threadIdx.x] = aa; red[1][threadIdx.x] = bb; red[2][threadIdx.x] = cr; red[3][threadIdx.x] = ci;
    __syncthreads();
    for (int q = PIT_PROBE / 2; q > 0; q >>= 1) {
        if (threadIdx.x < q)
            for (int u = 0; u < 4; u++) red[u][threadIdx.x] += red[u][threadIdx.x + q];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double A = red[0][0], B = red[1][0], Cr = red[2][0], Ci = red[3][0];
        double proj, gr = 1, gi = 0;                         // best group element g (yB ~ g yA) and Re(conj(g) c)
        if (sym == 0) {
            proj = sqrt(Cr * Cr + Ci * Ci);
            if (proj > 0) { gr = Cr / proj; gi = Ci / proj; }
        } else if (sym == 4 || sym == 2 || sym == 1) {          // nearest of +-1 (, +-i): signs and one comparison, no trigonometry
            if (sym == 4 && fabs(Ci) > fabs(Cr)) { gr = 0; gi = Ci >= 0 ? 1 : -1; proj = fabs(Ci); }
            else if (sym == 1) { gr = 1; proj = Cr; }
            else { gr = Cr >= 0 ? 1 : -1; proj = fabs(Cr); }
        } else {
            const double step = 6.283185307179586 / sym;
            const double k = rint(atan2(Ci, Cr) / step) * step;
            gr = cos(k); gi = sin(k);
            proj = Cr * gr + Ci * gi;
        }
        if (!result_probe) { gph[2 * ((size_t)blockIdx.x * gridDim.y + j)] = gr; gph[2 * ((size_t)blockIdx.x * gridDim.y + j) + 1] = gi; }
        double d2 = (A + B - 2 * proj) / (B > 1e-300 ? B : 1e-300);
        if (d2 < 0) d2 = 0;
        dfc[(size_t)blockIdx.x * gridDim.y + j] = (A == A && B == B) ? sqrt(d2) : 1e30;
        if (!result_probe) pw[(size_t)blockIdx.x * gridDim.y + j] = B / PIT_PROBE;
    }
}

// mean gain g of the error function around a converged output of power Py: d errfn / dy ~ -g (holomorphic part, decisions
// held), so that a tap perturbation decays like exp(-mu g Rc t).  cma / mcma from their constants; the radius- and decision-
// directed functions behave like (s - y) |s|^2 resp. (s - y) around the decided point.  0: no usable linearisation.
template <typename R> __device__ inline double pit_gain(int method, double Py, Cx<R> c0)
{
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: return 2 * Py - (double)c0.re;
    case QH_M_MCMA: return 1.5 * Py - 0.5 * ((double)c0.re + (double)c0.im);
    case QH_M_RDE: case QH_M_MRDE: case QH_M_MDDMA: return Py;
    case QH_M_DD: return 1.0;
    case QH_M_SBD: return 0.866 * sqrt(Py / 2);
    default: return 0.0;
    }
}

// end of a pass: largest boundary defect and the deviation estimate -> report; the pass's end taps become the sweep's result;
// converged -> later passes skip.  devmax: per-block maxima of sum_k lambda_k |D~_k[col]|^2 from pit_devest_kernel (ndev of them;
// nullptr / corr_on = 0: no estimate, the defect rule decides).
constexpr double PIT_DEV_SAFETY = 1.0, PIT_DEV_WORST = 3.0, PIT_DEV_TAPS = 2.0, PIT_DEV_TAPS_WORST = 3.0;    // rule: safety x rms estimate < tol, worst segment < 3 tol, taps (relative norm, rms over segments) < 2 tol
template <typename R> struct PitDecideArgs {
    const double *dfc, *pw;
    int nb;
    const Cx<R> *Ylast;
    int n;
    Cx<R> *wx;
    PitCtrl *c;
    float *host_view;
    int nrow;
    const float *devmax;
    int ndev;
    double safety;
    const float2 *Ye;
    float2 *Yprev;
    int ne, ncol_e;
    const double *theta;
    const int64_t *modes_dev;
    int ntot_w, S, sym, corr_wanted;
    // Final extrapolation (eigen-space analysis, from the second pass on): the result is the last segment's end taps PLUS the linearised
    // effect of the correction its start taps would get next, J D[S-1] - what the next pass would give to first order.  It matters for the
    // weakly excited directions (J ~ 1, model exact): their deviation - the bulk of the TAP deviation, invisible in the output - goes
    // away without another pass; the tap part of the stop rule is discounted by the contraction the last pass showed.
    const float2 *Dfin;              // D~ (ntot x ncol, after the scan) or nullptr
    const float2 *Vfin;              // V[i][k]
    const double *lam_fin;
    int stall_from;                  // first pass at which "nothing gained over two passes" ends the sweep (2; damped adaptive sweeps: 5)
    const float *extra;              // one more figure the criterion must cover (adaptive step: largest relative change of a segment's start step size), or nullptr
};
// (256 threads of ONE block: the kernel below, or the last block of pit_devest_kernel to finish)
constexpr int PIT_FIN_MAX = 128;     // rows of the eigenbasis the final extrapolation stages (= PIT_EIGMAX, asserted where that is defined)
template <typename R> __device__ __forceinline__ void pit_decide_body(const PitDecideArgs<R> &a)
{
    const double *dfc = a.dfc; const int nb = a.nb; const Cx<R> *Ylast = a.Ylast; const int n = a.n; Cx<R> *wx = a.wx; PitCtrl *c = a.c; float *host_view = a.host_view;
    const int nrow = a.nrow; const float *devmax = a.devmax; const int ndev = a.ndev; const double safety = a.safety; const float2 *Ye = a.Ye; float2 *Yprev = a.Yprev;
    const int ne = a.ne, ncol_e = a.ncol_e; const double *theta = a.theta; const int64_t *modes_dev = a.modes_dev; const int ntot_w = a.ntot_w, S = a.S, sym = a.sym, corr_wanted = a.corr_wanted;
    // the fields of the control block the decision reads, fetched now (uniform loads, one round trip under the reductions below;
    // read one by one between the stores at the end they were most of this function's time)
    const int p = c->passes, corr_on0 = c->corr_on;
    const double c_out_power = c->out_power, c_gain = c->gain, c_power = c->power, c_mu = c->mu, c_tol = c->tol;
    const double c_seg_len = (double)c->seg_len;
    const int pl = p - 1 < QH_PIT_MAXPASS ? p - 1 : QH_PIT_MAXPASS - 1, pl2 = p - 2 < QH_PIT_MAXPASS ? p - 2 : QH_PIT_MAXPASS - 1;
    const double drms1 = p >= 1 ? c->deviation_rms[pl] : 0.0, drms2 = p >= 2 ? c->deviation_rms[pl2] : 0.0, dfc2 = p >= 2 ? c->defect[pl2] : 0.0;
    if (Yprev)                                                // eigen-space copy of this pass's result, for the next pass's "how far did the result move"
        for (int e = threadIdx.x; e < ne * nrow; e += 256) { const int k = e / nrow, j = e - k * nrow; Yprev[e] = Ye[(size_t)k * ncol_e + (ncol_e - nrow) + j]; }
    __shared__ double red[4], redd[4], reds[4], redt[4], redw[4], redm[4];
    __shared__ int s_flag[2];
    __shared__ float s_crit;
    double m = 0, dv = 0, ds = 0, dt = 0, wn = 0, dm = 0;
    for (int i0 = threadIdx.x; i0 < nb; i0 += 8 * 256) {       // eight loads in flight per thread (one by one they were most of this function's time)
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = i0 + 256 * q < nb ? dfc[i0 + 256 * q] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) m = (v[q] > m || !(v[q] == v[q])) ? (v[q] == v[q] ? v[q] : 1e30) : m;
    }
    if (devmax)
        for (int i = threadIdx.x; i < ndev; i += 256) {
            const double v = (double)devmax[4 * i];
            dv = (v > dv || !(v == v)) ? (v == v ? v : 1e30) : dv;
            ds += (double)devmax[4 * i + 1];
            dt += (double)devmax[4 * i + 2];
            const double vm = (double)devmax[4 * i + 3];
            dm = (vm > dm || !(vm == vm)) ? (vm == vm ? vm : 1e30) : dm;
        }
    // how far the sweep's result still moved in this pass, in output terms (pit_defect_kernel's extra block row; nrow entries after the nb boundaries)
    double chg = 0;
    for (int r = 0; r < nrow; r++) { const double v = dfc[nb + r]; chg = (v > chg || !(v == v)) ? v : chg; }
    // The pass's result (and |taps|^2 of all output modes).  The functions with a continuous symmetry (cma, rde: every common phase)
    // leave the phase of the taps free, and the end taps of the last segment sit in that segment's own frame: theta_{S-1}, the
    // product of the boundary rotations of this pass, takes them into the frame of segment 0 - the caller's start taps, i.e. the
    // frame of the sequential recurrence.  (A few 1e-3 rad over thousands of boundaries; the quarter-turn functions lock the phase
    // themselves.)  Everything this needs is there before the function starts: issued with the loads above, not after the reduction.
    const bool extrap = a.Dfin != nullptr && p >= 1 && corr_on0 && c_gain > 0;
    for (int e = threadIdx.x; e < n; e += 256) {
        Cx<R> v = Ylast[e];
        wn += (double)v.re * v.re + (double)v.im * v.im;
        const int row = e / ntot_w;
        int jj = -1;
        for (int j = 0; j < nrow; j++) if ((int)modes_dev[j] == row) { jj = j; break; }
        if (jj >= 0 && theta) {
            const double tr = theta[2 * ((size_t)(S - 1) * nrow + jj)], ti = theta[2 * ((size_t)(S - 1) * nrow + jj) + 1];
            if (sym == 0) v = Cx<R>{(R)(tr * v.re - ti * v.im), (R)(tr * v.im + ti * v.re)};       // into the frame of segment 0
        }
        wx[e] = v;
    }
    // six reductions over the block: shuffles inside the waves, the four wave results through LDS
    for (int o = 32; o > 0; o >>= 1) {
        const double m2 = __shfl_xor(m, o), dv2 = __shfl_xor(dv, o), dm2 = __shfl_xor(dm, o);
        m = m > m2 ? m : m2; dv = dv > dv2 ? dv : dv2; dm = dm > dm2 ? dm : dm2;
        ds += __shfl_xor(ds, o); dt += __shfl_xor(dt, o); wn += __shfl_xor(wn, o);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[w] = m; redd[w] = dv; reds[w] = ds; redt[w] = dt; redw[w] = wn; redm[w] = dm;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < 4; w++) {
            red[0] = red[0] > red[w] ? red[0] : red[w]; redd[0] = redd[0] > redd[w] ? redd[0] : redd[w]; redm[0] = redm[0] > redm[w] ? redm[0] : redm[w];
            reds[0] += reds[w]; redt[0] += redt[w]; redw[0] += redw[w];
        }
    if (threadIdx.x == 0) {
        const bool have_dev = devmax != nullptr && corr_on0 && c_out_power > 0;
        const double dev = have_dev ? sqrt(redd[0] / c_out_power) : -1.0;                            // worst segment
        const double dev_rms = have_dev ? sqrt(reds[0] / ((double)(nb + nrow) * c_out_power)) : -1.0;  // rms over segments and modes
        // taps: rms over segments of |D[s]| / |w|, |w|^2 = the squared tap norm of one output mode (mean over the modes that have taps)
        const double wnorm2 = redw[0] / (double)(nrow > 0 ? nrow : 1);
        const double dev_tap = (have_dev && wnorm2 > 0) ? sqrt(redt[0] / ((double)(nb + nrow) * wnorm2)) : -1.0;
        const double dev_tap_worst = (have_dev && wnorm2 > 0) ? sqrt(redm[0] / wnorm2) : -1.0;
        if (p < QH_PIT_MAXPASS) { c->defect[p] = red[0]; c->result_change[p] = chg; c->deviation[p] = dev; c->deviation_rms[p] = dev_rms; c->deviation_taps[p] = dev_tap; c->deviation_taps_worst[p] = dev_tap_worst; }
        c->passes = p + 1;
        // The criterion.  With the coarse correction: `dev`, the first-order estimate of how far this pass's trajectory is from the
        // sequential recurrence (rms of the output, worst segment, relative to the output rms; include/qampy_hip.h "Stop rule"),
        // times a safety factor for what the linearised model misses.  Without it: the largest boundary defect, which says how
        // far the trajectory is only through how much a segment forgets - errors of the start taps decay by rho = exp(-mu g T
        // lambda) per segment, so defects of size d everywhere leave ~ d / (1 - rho); `tol` is then calibrated on the automatic
        // grid (mu T = 0.2) and the defect of shorter segments is held to a proportionally smaller value.
        double crit;
        if (have_dev) {
            crit = safety * dev_rms;
            if (safety * dev / PIT_DEV_WORST > crit) crit = safety * dev / PIT_DEV_WORST;
            // with the final extrapolation the taps are ahead of this pass by what one more would gain: the contraction the estimate just
            // showed (never taken better than 0.25)
            double rho = 1.0;
            if (extrap && drms1 > 0) { rho = dev_rms / drms1; rho = rho < 0.25 ? 0.25 : (rho > 1.0 ? 1.0 : rho); }
            if (dev_tap >= 0 && safety * rho * dev_tap / PIT_DEV_TAPS > crit) crit = safety * rho * dev_tap / PIT_DEV_TAPS;
            if (dev_tap_worst >= 0 && safety * rho * dev_tap_worst / PIT_DEV_TAPS_WORST > crit) crit = safety * rho * dev_tap_worst / PIT_DEV_TAPS_WORST;    // (the final taps sit at the worst segment's value)
        }
        else {
            double amp = 1.0;
            if (c_gain > 0 && c_power > 0 && c_mu > 0) {
                const double gl = c_gain * 2.0 * c_power, a = c_mu * c_seg_len * gl;
                if (a > 0) amp = (1.0 - exp(-0.2 * gl)) / (1.0 - exp(-a));
                if (!(amp > 1.0)) amp = 1.0;
            }
            crit = red[0] * amp;
        }
        // the correction is given up only when the estimate has GROWN two passes in a row (a model that drives the iteration apart);
        // a pass without progress is not a reason - plain relaxation leaves the weakly excited directions where they are
        if (have_dev && p >= 2 && p < QH_PIT_MAXPASS && drms1 > 0 && drms2 > 0 && dev_rms > 1.5 * drms1 && drms1 > 1.5 * drms2) c->corr_on = 0;
        // Certified only by the deviation estimate when a coarse model exists: small boundary defects alone say nothing about the
        // weakly excited directions (round 2's defect rule certified 64-QAM mrde runs whose taps were 3e-2 off).  Without a model
        // (more than 96 taps per output mode) the defect rule is all there is.
        if (a.extra) { const double ex = (double)a.extra[0]; if (p < QH_PIT_MAXPASS) c->result_change[p] = ex; if (ex > crit || !(ex == ex)) crit = ex == ex ? ex : 1e30; }   // (adaptive step: result_change[] reports the change of the start step sizes)
        int done = 0, conv = 0;
        if (crit < c_tol && (have_dev || !corr_wanted)) { c->converged = 1; done = 1; conv = 1; }
        // nothing gained over two passes: the trajectory has no fixed point the passes can agree on (a stage that cannot track the
        // carrier) - stop, NOT converged; further passes would only cost time.  (Slow but steady gains - rde's ring decisions - go on
        // to max_passes: the uncertified result keeps improving with them.)
        else if (p >= a.stall_from && p < QH_PIT_MAXPASS) {
            const bool use_dev = have_dev && drms2 > 0;
            const double prev2 = use_dev ? drms2 : dfc2, now = use_dev ? dev_rms : red[0];
            if (!(now < prev2)) done = 1;
        }
        if (done) c->done = 1;
        s_flag[0] = done; s_flag[1] = conv; s_crit = (float)crit;
    }
    __syncthreads();
    if (s_flag[1] && extrap) {                                    // the certified last pass: its result + J D[S-1] (see PitDecideArgs::Dfin)
        // delta = V (c o D~[S-1]) in the frame of segment 0, one output mode at a time: c o D~ staged in LDS (one exponential per component,
        // not one per component and tap), V read through its transposed copy so that the threads of a wave read consecutive addresses
        // (round 4's first version - a strided row of V per thread, 82 dependent exponentials - was 24 us of the last pass of every stage)
        __shared__ float2 s_cd[PIT_FIN_MAX];
        const float2 *VT = a.Vfin + (size_t)ntot_w * ntot_w;
        for (int jj = 0; jj < nrow; jj++) {
            const size_t col = (size_t)(S - 1) * nrow + jj;
            for (int k = threadIdx.x; k < ntot_w; k += 256) {
                const double ak = c_mu * c_gain * c_seg_len * a.lam_fin[k];
                const float ck = ak > 0 ? __expf(-(float)ak) : 1.f;
                const float2 dk = a.Dfin[(size_t)k * ncol_e + col];
                s_cd[k] = float2{ck * dk.x, ck * dk.y};
            }
            __syncthreads();
            const int row = (int)modes_dev[jj];
            const double tr = theta[2 * ((size_t)(S - 1) * nrow + jj)], ti = theta[2 * ((size_t)(S - 1) * nrow + jj) + 1];
            for (int f = threadIdx.x; f < ntot_w; f += 256) {
                float dr = 0.f, di = 0.f;
#pragma unroll 8
                for (int k = 0; k < ntot_w; k++) {
                    const float2 vk = VT[(size_t)k * ntot_w + f], cd = s_cd[k];
                    dr += vk.x * cd.x - vk.y * cd.y; di += vk.x * cd.y + vk.y * cd.x;
                }
                const int e = row * ntot_w + f;
                const Cx<R> v = wx[e];
                if (sym == 0) wx[e] = Cx<R>{(R)(v.re + dr), (R)(v.im + di)};                                                   // (already in the frame of segment 0)
                else wx[e] = Cx<R>{(R)(v.re + tr * dr + ti * di), (R)(v.im + tr * di - ti * dr)};                                 // own frame: conj(theta) delta
            }
            __syncthreads();
        }
        __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // what the host reads after the pass: criterion (to decide whether the pass after the next one is worth enqueueing early), then
        // the flag it polls for (host_view is pinned, coherent host memory: no copy kernel, no event in between)
        host_view[1] = s_crit;
        __threadfence_system();
        __hip_atomic_store(&host_view[0], s_flag[0] ? (s_flag[1] ? 1.f : 2.f) : 0.f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (2: stopped, NOT certified)
    }
}

template <typename R> __global__ void __launch_bounds__(256) pit_decide_kernel(PitDecideArgs<R> a)
{
    QH_WAVE_FIRST();
    if (a.c->done) return;
    pit_decide_body<R>(a);
}

// Error trace of the final pass into the frame of segment 0 (functions with a continuous symmetry only, see pit_decide_kernel):
// errfn(g y) = g errfn(y), so the errors of segment s turn with theta_s.  Launched once, after the pass that ended the sweep (the
// passes enqueued ahead of the host's knowledge return at `done` and leave theta alone).
template <typename R>
__global__ void __launch_bounds__(256) pit_rotate_err_kernel(Cx<R> *err, int64_t err_pitch, int64_t err_off, PitSeg sg, const int64_t *modes_dev, int nsel,
                                                             const double *theta, const PitCtrl *c, int p)
{
    QH_WAVE_FIRST();
    if (p >= 0 && (!c->done || c->passes != p + 1)) return;       // p < 0: launched once, after the last pass
    const int s = blockIdx.x, j = blockIdx.y;
    if (s == 0) return;                                           // theta_0 = 1
    const double tr = theta[2 * ((size_t)s * nsel + j)], ti = theta[2 * ((size_t)s * nsel + j) + 1];
    Cx<R> *row = err + (size_t)modes_dev[j] * err_pitch + err_off + sg.start(s);
    const int64_t n = sg.steps(s);
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const Cx<R> v = row[i];
        row[i] = Cx<R>{(R)(tr * v.re - ti * v.im), (R)(tr * v.im + ti * v.re)};
    }
}

// ------------------------------------------------------------------------------------------------ coarse correction
typedef double2 Z;
__device__ __forceinline__ Z zmul(Z a, Z b) { return Z{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// Rc[f][f'] = sum over a subsample of the training windows of conj(x_i[f]) x_i[f'],  x_i[f = k ntaps + t] = E[k, i os + t].
// Block b stages PIT_COVW windows in LDS and writes its partial sums to part[b]; pit_cov_reduce_kernel adds the blocks up.
constexpr int PIT_COVW = 32, PIT_COVB = 256;                       // windows per block, blocks (one per CU: the eigensolver waits for this)
template <typename R, int EPT>
__global__ void __launch_bounds__(256) pit_cov_kernel(const Cx<R> *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, int nwin, Z *part)
{
    QH_WAVE_FIRST();
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    Cx<R> *x = reinterpret_cast<Cx<R> *>(pit_smem);               // [PIT_COVW][ntot]
    const int ntot = nmodes * ntaps;
    const int nent = ntot * ntot;
    const int64_t stride = TrSyms / nwin > 0 ? TrSyms / nwin : 1;
    for (int e = threadIdx.x; e < PIT_COVW * ntot; e += 256) {
        const int w = e / ntot, f = e - w * ntot;
        const int k = f / ntaps, t = f - k * ntaps;
        const int64_t i = ((int64_t)blockIdx.x * PIT_COVW + w) * stride;
        x[e] = (i < TrSyms && (int64_t)blockIdx.x * PIT_COVW + w < nwin) ? E[(size_t)k * L + i * os + t] : Cx<R>{0, 0};
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < EPT; q++) {
        const int e = threadIdx.x + 256 * q;
        if (e >= nent) break;
        const int fa = e / ntot, fb = e - fa * ntot;
        float ar = 0, ai = 0;
#pragma unroll 8
        for (int w = 0; w < PIT_COVW; w++) {
            const Cx<R> u = x[w * ntot + fa], v = x[w * ntot + fb];
            ar += (float)(u.re * v.re + u.im * v.im);
            ai += (float)(u.re * v.im - u.im * v.re);
        }
        part[(size_t)blockIdx.x * nent + e] = Z{(double)ar, (double)ai};
    }
}
// 64 entries per block, 4 threads per entry (each a quarter of the partial sums, loads unrolled so that they overlap)
static __global__ void __launch_bounds__(256) pit_cov_reduce_kernel(const Z *part, int nent, int nblk, Z *Rc)
{
    QH_WAVE_FIRST();
    __shared__ double sr[256], si[256];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    double ar = 0, ai = 0;
    if (e < nent) {
#pragma unroll 16
        for (int b = grp; b < nblk; b += 4) { const Z v = part[(size_t)b * nent + e]; ar += v.x; ai += v.y; }
    }
    sr[threadIdx.x] = ar; si[threadIdx.x] = ai;
    __syncthreads();
    if (grp == 0 && e < nent) {
        const int t = threadIdx.x;
        Rc[e] = Z{(sr[t] + sr[t + 64]) + (sr[t + 128] + sr[t + 192]), (si[t] + si[t + 64]) + (si[t + 128] + si[t + 192])};
    }
}

// Eigenbasis of the covariance: cyclic Jacobi (parallel round-robin ordering, n/2 disjoint rotations per round) on the
// Hermitian matrix Rc / nwin, whole problem in the LDS of ONE workgroup, single precision (the basis only preconditions the
// relaxation).  Out: lam[n] eigenvalues, V[i][k] = component i of eigenvector k.  n <= PIT_EIGMAX.
typedef float2 Zf;
constexpr int PIT_EIGMAX = 128, PIT_EIGSWEEPS = 5;
static_assert(PIT_FIN_MAX >= PIT_EIGMAX, "final extrapolation stages one eigen-space column");
// PIT_EIGMAX: up to 2 x 64 taps (round 2: 96; above 96 the solver must log its rotations: A alone fills the LDS)        // 5 sweeps: off-diagonal norm 2e-3, smallest eigenvalues good to 2 % - a preconditioner (3 sweeps: twice the passes on some captures)
__device__ __forceinline__ Zf cmulf(Zf a, Zf b) { return Zf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// glog != nullptr: the rotations are LOGGED ([sweep][round][pair] -> (c, s, e.re, e.im)) instead of being accumulated in V - a third of
// the LDS traffic that bounds this kernel - and pit_jacobi_v_kernel applies them to the rows of V afterwards, which are independent
// of each other (1.12 -> ~0.78 ms until the basis is there: it had become what the first correction of a cold start waits for).
static __global__ void __launch_bounds__(1024) pit_jacobi_kernel(const Z *Rc, int n, double norm, double *lam, Zf *Vout, int nsweep, float4 *glog)
{
    QH_WAVE_FIRST();
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    const int ld = n + 1;
    Zf *A = reinterpret_cast<Zf *>(pit_smem);                 // [n][ld]
    Zf *V = A + (size_t)n * ld;                               // [n][ld]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < n; i += 16)
        for (int j = lane; j < n; j += 64) {
            const Z v = Rc[(size_t)i * n + j];
            A[i * ld + j] = Zf{(float)(v.x * norm), (float)(v.y * norm)};
            if (!glog) V[i * ld + j] = Zf{i == j ? 1.f : 0.f, 0.f};
        }
    __syncthreads();
    const int m = (n + 1) & ~1;                               // players of the round-robin (a dummy when n is odd)
    const int npair = m / 2;
    constexpr int NP = (PIT_EIGMAX / 2 + 15) / 16, NR = (PIT_EIGMAX + 63) / 64;
    for (int sweep = 0; sweep < nsweep; sweep++) {
        for (int r = 0; r < m - 1; r++) {
            // A wave owns pairs wave, wave + 16, ... of the round (circle method: player m-1 stays, the others rotate) in BOTH
            // phases, so it works out their rotations itself (one lane per pair, broadcast) instead of waiting for a few threads
            // to do it for the whole workgroup behind one more barrier.
            int2 ixs[NP];
            float4 gs[NP];
            auto pair_of = [&](int i) -> int2 {
                int pa, pb;
                if (i == 0) { pa = m - 1; pb = r; }
                else {
                    pa = r + i; if (pa >= m - 1) pa -= m - 1;
                    pb = r - i; if (pb < 0) pb += m - 1;
                }
                const int p = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
                return (i < npair && q < n) ? int2{p, q} : int2{0, n};
            };
            float4 gl = {1.f, 0.f, 1.f, 0.f};                     // lane a < NP works out the rotation of the wave's pair a
            {
                const int2 ix = pair_of(wave + 16 * (lane < NP ? lane : 0));
                if (lane < NP && ix.y < n) {
                    const Zf apq = A[ix.x * ld + ix.y];
                    const float app = A[ix.x * ld + ix.x].x, aqq = A[ix.y * ld + ix.y].x;
                    const float m2 = apq.x * apq.x + apq.y * apq.y;
                    if (m2 > 1e-36f) {                                // hardware reciprocals (1 ulp): the basis is a preconditioner
                        const float rmag = __builtin_amdgcn_rsqf(m2);
                        const float tau = (aqq - app) * 0.5f * rmag;
                        const float t = (tau >= 0 ? 1.f : -1.f) * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(1.f + tau * tau));
                        const float c = __builtin_amdgcn_rsqf(1.f + t * t);
                        gl = float4{c, t * c, apq.x * rmag, apq.y * rmag};
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < NP; a++) {
                ixs[a] = pair_of(wave + 16 * a);
                gs[a] = float4{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl.x), a)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl.y), a)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl.z), a)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl.w), a))};
            }
            if (glog && lane < NP) {                              // lane a logs the wave's pair a of this round
                const int pi = wave + 16 * lane;
                if (pi < npair) glog[((size_t)sweep * (m - 1) + r) * npair + pi] = gl;
            }
            // (no barrier here: the pivots of a pair sit in its own rows / columns, which only its own wave rotates)
            // columns p, q of A and V:  col_p' = c col_p - s conj(e) col_q ;  col_q' = s col_p + c conj(e) col_q
            // (a thread's items - up to 3 pairs x 2 rows x 2 matrices - are all read before any is rotated: one LDS latency)
            {
                Zf xp[NP][NR][2], xq[NP][NR][2];
#pragma unroll
                for (int a = 0; a < NP; a++) {
                    const int2 ix = ixs[a];
                    if (ix.y >= n) continue;                      // wave-uniform
#pragma unroll
                    for (int b = 0; b < NR; b++) {
                        if (64 * b >= n) continue;                    // wave-uniform; lanes past the last row redo row n-1 (same inputs, same result)
                        const int row = lane + 64 * b < n ? lane + 64 * b : n - 1;
                        xp[a][b][0] = A[row * ld + ix.x]; xq[a][b][0] = A[row * ld + ix.y];
                        if (!glog) { xp[a][b][1] = V[row * ld + ix.x]; xq[a][b][1] = V[row * ld + ix.y]; }
                    }
                }
#pragma unroll
                for (int a = 0; a < NP; a++) {
                    const int2 ix = ixs[a];
                    const float4 g = gs[a];
                    if (ix.y >= n) continue;
#pragma unroll
                    for (int b = 0; b < NR; b++) {
                        if (64 * b >= n) continue;
                        const int row = lane + 64 * b < n ? lane + 64 * b : n - 1;
#pragma unroll
                        for (int which = 0; which < 2; which++) {
                            if (which && glog) continue;                  // (uniform) V is built from the log afterwards
                            Zf *Mx = which ? V : A;
                            const Zf up = xp[a][b][which];
                            const Zf xqe = cmulf(Zf{g.z, -g.w}, xq[a][b][which]);        // conj(e) x_q
                            Mx[row * ld + ix.x] = Zf{g.x * up.x - g.y * xqe.x, g.x * up.y - g.y * xqe.y};
                            Mx[row * ld + ix.y] = Zf{g.y * up.x + g.x * xqe.x, g.y * up.y + g.x * xqe.y};
                        }
                    }
                }
            }
            __syncthreads();
            // rows p, q of A:  row_p' = c row_p - s e row_q ;  row_q' = s row_p + c e row_q
            {
                Zf xp[NP][NR], xq[NP][NR];
#pragma unroll
                for (int a = 0; a < NP; a++) {
                    const int2 ix = ixs[a];
                    if (ix.y >= n) continue;
#pragma unroll
                    for (int b = 0; b < NR; b++) {
                        if (64 * b >= n) continue;
                        const int col = lane + 64 * b < n ? lane + 64 * b : n - 1;
                        xp[a][b] = A[ix.x * ld + col]; xq[a][b] = A[ix.y * ld + col];
                    }
                }
#pragma unroll
                for (int a = 0; a < NP; a++) {
                    const int2 ix = ixs[a];
                    const float4 g = gs[a];
                    if (ix.y >= n) continue;
#pragma unroll
                    for (int b = 0; b < NR; b++) {
                        if (64 * b >= n) continue;
                        const int col = lane + 64 * b < n ? lane + 64 * b : n - 1;
                        const Zf up = xp[a][b];
                        const Zf xqe = cmulf(Zf{g.z, g.w}, xq[a][b]);
                        A[ix.x * ld + col] = Zf{g.x * up.x - g.y * xqe.x, g.x * up.y - g.y * xqe.y};
                        A[ix.y * ld + col] = Zf{g.y * up.x + g.x * xqe.x, g.y * up.y + g.x * xqe.y};
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int k = tid; k < n; k += 1024) lam[k] = (double)A[k * ld + k].x;
    if (glog) return;
    for (int i = wave; i < n; i += 16)
        for (int k = lane; k < n; k += 64) { Vout[(size_t)i * n + k] = V[i * ld + k]; Vout[(size_t)n * n + (size_t)i * n + k] = V[k * ld + i]; }   // V, then V^T (both products read rows)
}

// The same sweeps in 2 x 2 BLOCK form, one barrier per round: the pairs of a round are disjoint, so A' = J^H A J falls apart into
// the blocks A'[I][J] = G_I^H A[I][J] G_J over pairs I = (p, q), J = (r, s) - four elements read, four written, by ONE thread, which
// works out G_I and G_J itself from the pivots (three LDS reads and a dozen flops each: cheaper than waiting for them).  Reading
// from one copy of A and writing the other (ping-pong) removes the hazard between a block holding pivots and its readers, so a
// round is: the rotations of its pairs (one thread each), barrier, every thread its block of the upper triangle and the mirror image, barrier.  (The row / column form above: two passes over A and two barriers per
// round, 2.2 us per round at n = 82 - the basis was what the first correction of a cold sweep waited for.)  Rotations logged for
// pit_jacobi_v_kernel; two copies of A: n <= 96.
static __global__ void __launch_bounds__(1024) pit_jacobi_blk_kernel(const Z *Rc, int n, double norm, double *lam, int nsweep, float4 *glog)
{
    QH_WAVE_FIRST();
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    const int ld = n + 1;
    Zf *src = reinterpret_cast<Zf *>(pit_smem), *dst = src + (size_t)n * ld;
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 1024) {
        const int i = e / n, j = e - i * n;
        const Z v = Rc[e];
        src[i * ld + j] = Zf{(float)(v.x * norm), (float)(v.y * norm)};
    }
    __syncthreads();
    const int m = (n + 1) & ~1, npair = m / 2, ntri = npair * (npair + 1) / 2;
    // A stays Hermitian: a thread takes the blocks (I, J), I <= J, of the upper triangle - the same ones in every round - and writes
    // their mirror images too
    __shared__ float4 grot[PIT_EIGMAX / 2 + 1];
    __shared__ int2 gpair[PIT_EIGMAX / 2 + 1];
    constexpr int NB = (PIT_EIGMAX / 2) * (PIT_EIGMAX / 2 + 1) / 2 / 1024 + 1;
    int bI[NB], bJ[NB], nbt = 0;
    {
        auto off = [&](int I) { return I * npair - I * (I - 1) / 2; };
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int bq = tid + 1024 * k;
            bI[k] = 0; bJ[k] = 0;
            if (bq < ntri) {
                int I = 0;
                while (bq >= off(I + 1)) I++;
                bI[k] = I; bJ[k] = I + (bq - off(I));
                nbt = k + 1;
            }
        }
    }
    for (int sweep = 0; sweep < nsweep; sweep++) {
        for (int r = 0; r < m - 1; r++) {
            // the round's pairs and rotations, once (circle method: player m - 1 stays, the others rotate; q = n: the partner of the
            // dummy player - n odd - sits this round out).  (c, s, e.re, e.im); hardware reciprocals (1 ulp): the basis is a preconditioner
            if (tid < npair) {
                int pa, pb;
                if (tid == 0) { pa = m - 1; pb = r; }
                else {
                    pa = r + tid; if (pa >= m - 1) pa -= m - 1;
                    pb = r - tid; if (pb < 0) pb += m - 1;
                }
                const int2 ix{pa < pb ? pa : pb, (pa < pb ? pb : pa) < n ? (pa < pb ? pb : pa) : n};
                float4 g = {1.f, 0.f, 1.f, 0.f};
                if (ix.y < n) {
                    const Zf apq = src[ix.x * ld + ix.y];
                    const float app = src[ix.x * ld + ix.x].x, aqq = src[ix.y * ld + ix.y].x;
                    const float m2 = apq.x * apq.x + apq.y * apq.y;
                    if (m2 > 1e-36f) {
                        const float rmag = __builtin_amdgcn_rsqf(m2);
                        const float tau = (aqq - app) * 0.5f * rmag;
                        const float t = (tau >= 0 ? 1.f : -1.f) * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(1.f + tau * tau));
                        const float c = __builtin_amdgcn_rsqf(1.f + t * t);
                        g = float4{c, t * c, apq.x * rmag, apq.y * rmag};
                    }
                }
                grot[tid] = g; gpair[tid] = ix;
                glog[((size_t)sweep * (m - 1) + r) * npair + tid] = g;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NB; k++) {
                if (k >= nbt) break;
                const int I = bI[k], J = bJ[k];
                const int2 pi = gpair[I], pj = gpair[J];
                const float4 gi = grot[I], gj = grot[J];
                const bool qi = pi.y < n, qj = pj.y < n;
                const Zf z{0.f, 0.f};
                const Zf a_pr = src[pi.x * ld + pj.x], a_ps = qj ? src[pi.x * ld + pj.y] : z;
                const Zf a_qr = qi ? src[pi.y * ld + pj.x] : z, a_qs = (qi && qj) ? src[pi.y * ld + pj.y] : z;
                // T = A[I][J] G_J:  col_r' = c col_r - s conj(e) col_s ;  col_s' = s col_r + c conj(e) col_s
                const Zf ejc{gj.z, -gj.w};
                const Zf ps_e = cmulf(ejc, a_ps), qs_e = cmulf(ejc, a_qs);
                const Zf t_pr{gj.x * a_pr.x - gj.y * ps_e.x, gj.x * a_pr.y - gj.y * ps_e.y}, t_ps{gj.y * a_pr.x + gj.x * ps_e.x, gj.y * a_pr.y + gj.x * ps_e.y};
                const Zf t_qr{gj.x * a_qr.x - gj.y * qs_e.x, gj.x * a_qr.y - gj.y * qs_e.y}, t_qs{gj.y * a_qr.x + gj.x * qs_e.x, gj.y * a_qr.y + gj.x * qs_e.y};
                // A' = G_I^H T:  row_p' = c row_p - s e row_q ;  row_q' = s row_p + c e row_q
                const Zf ei{gi.z, gi.w};
                const Zf qr_e = cmulf(ei, t_qr), qs_e2 = cmulf(ei, t_qs);
                const Zf n_pr{gi.x * t_pr.x - gi.y * qr_e.x, gi.x * t_pr.y - gi.y * qr_e.y}, n_ps{gi.x * t_ps.x - gi.y * qs_e2.x, gi.x * t_ps.y - gi.y * qs_e2.y};
                const Zf n_qr{gi.y * t_pr.x + gi.x * qr_e.x, gi.y * t_pr.y + gi.x * qr_e.y}, n_qs{gi.y * t_ps.x + gi.x * qs_e2.x, gi.y * t_ps.y + gi.x * qs_e2.y};
                dst[pi.x * ld + pj.x] = n_pr;
                if (qj) dst[pi.x * ld + pj.y] = n_ps;
                if (qi) dst[pi.y * ld + pj.x] = n_qr;
                if (qi && qj) dst[pi.y * ld + pj.y] = n_qs;
                if (I != J) {                                         // the mirror block A'[J][I] = A'[I][J]^H
                    dst[pj.x * ld + pi.x] = Zf{n_pr.x, -n_pr.y};
                    if (qj) dst[pj.y * ld + pi.x] = Zf{n_ps.x, -n_ps.y};
                    if (qi) dst[pj.x * ld + pi.y] = Zf{n_qr.x, -n_qr.y};
                    if (qi && qj) dst[pj.y * ld + pi.y] = Zf{n_qs.x, -n_qs.y};
                }
            }
            __syncthreads();
            Zf *t = src; src = dst; dst = t;
        }
    }
    for (int k = tid; k < n; k += 1024) lam[k] = (double)src[k * ld + k].x;
}

// Row i of V = e_i times the logged rotations, in their order: one wave per row (the rows do not interact), lane a = pair a of a
// round (the pairs of a round are disjoint), the log staged through LDS 27 rounds at a time.  Writes V and V^T like the kernel above.
constexpr int PIT_JV_CH = 27;
static __global__ void __launch_bounds__(64) pit_jacobi_v_kernel(const float4 *glog, int n, int nsweep, Zf *Vout)
{
    QH_WAVE_FIRST();
    __shared__ Zf row[PIT_EIGMAX + 2];
    __shared__ float4 gch[PIT_JV_CH * (PIT_EIGMAX / 2 + 1)];
    const int lane = threadIdx.x, i = blockIdx.x;
    const int m = (n + 1) & ~1, npair = m / 2;
    for (int k = lane; k < n + 2; k += 64) row[k] = Zf{k == i ? 1.f : 0.f, 0.f};
    const int nround = nsweep * (m - 1);
    for (int r0 = 0; r0 < nround; r0 += PIT_JV_CH) {
        const int nr = nround - r0 < PIT_JV_CH ? nround - r0 : PIT_JV_CH;
        __syncthreads();
        for (int e = lane; e < nr * npair; e += 64) gch[e] = glog[(size_t)r0 * npair + e];
        __syncthreads();
        for (int rr = 0; rr < nr; rr++) {
            const int r = (r0 + rr) % (m - 1);
            if (lane < npair) {
                int pa, pb;
                if (lane == 0) { pa = m - 1; pb = r; }
                else {
                    pa = r + lane; if (pa >= m - 1) pa -= m - 1;
                    pb = r - lane; if (pb < 0) pb += m - 1;
                }
                const int p = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
                if (q < n) {
                    const float4 g = gch[rr * npair + lane];
                    const Zf up = row[p];
                    const Zf xqe = cmulf(Zf{g.z, -g.w}, row[q]);                 // conj(e) x_q
                    row[p] = Zf{g.x * up.x - g.y * xqe.x, g.x * up.y - g.y * xqe.y};
                    row[q] = Zf{g.y * up.x + g.x * xqe.x, g.y * up.y + g.x * xqe.y};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // one wave: its LDS operations are in program order; no barrier per round
        }
    }
    for (int k = lane; k < n; k += 64) { Vout[(size_t)i * n + k] = row[k]; Vout[(size_t)n * n + (size_t)k * n + i] = row[k]; }
}

// C[m][c] = sum_k op(A)[m][k] B[k][c],  op(A) = A (CONJT = false) or A^H; A is n x n (n <= PIT_EIGMAX), B and C are n x ncol
// (row-major).  A block takes 16 columns (a few thousand columns in all: twice as many blocks as CUs rather than half as many):
// op(A) and the B tile are staged in LDS, a thread accumulates 3 rows x 2 columns in registers (8 x 32 threads).
constexpr int PIT_GT = 16, PIT_GC = PIT_GT / 8;
// what the products read / write besides the matrices: the defect vectors are formed while the B tile is staged (CONJT: B[f][col] =
// theta_{s-1} Y[s-1][mode_j][f] - theta_s X[s][mode_j][f], 0 for s = 0) and the corrected start taps are written by the
// epilogue of the back-transform (X[s] = Y[s] = theta_s X[s] + C[.][col]) - no defect / correction arrays in HBM in between
template <typename R> struct PitFuse {
    Cx<R> *X, *Y;
    const double *theta;
    const int64_t *modes_dev;
    int nmodes, nsel;
    float damp = 1.f;                // X = theta X + damp V D~ (1: the full correction)
    // forward product of TWO tap-set arrays in one launch (MODE 0): blocks [0, nb) take T -> Out, blocks [nb, 2 nb) take T2 -> Out2 (nb = ceil(ncol / 16))
    const Cx<R> *T2 = nullptr;
    float2 *Out2 = nullptr;
};
// A2 = the matrix whose ROWS are read: V for the forward product (op(A) = V^H: op(A)[m][k] = conj(V[k][m])), V^T for the back
// transform (op(A) = V: op(A)[m][k] = V^T[k][m]) - consecutive threads read consecutive m either way.
template <typename R, bool CONJT>
__global__ void __launch_bounds__(256) pit_cgemm_kernel(const Zf *A2, const Zf *B, Zf *C, int n, int ncol, const PitCtrl *c, PitFuse<R> fz)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    Zf *As = reinterpret_cast<Zf *>(pit_smem);                // [k][PIT_EIGMAX]  op(A)[m][k] stored k-major: rows m contiguous
    Zf *Bs = As + (size_t)PIT_EIGMAX * PIT_EIGMAX;            // [k][PIT_GT]
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;    // columns PIT_GC tx .., rows ty + 32 u
    const int col0 = blockIdx.x * PIT_GT;
    const size_t wset = (size_t)fz.nmodes * n;
    for (int e = threadIdx.x; e < PIT_EIGMAX * n; e += 256) {
        const int k = e / PIT_EIGMAX, m = e - k * PIT_EIGMAX;
        Zf v{0.f, 0.f};
        if (m < n) { v = A2[(size_t)k * n + m]; if (CONJT) v.y = -v.y; }
        As[e] = v;
    }
    if (CONJT) {                                              // defect vectors, formed here (consecutive threads: consecutive f of one column)
        for (int e = threadIdx.x; e < n * PIT_GT; e += 256) {
            const int cc = e / n, k = e - cc * n, col = col0 + cc;
            Zf v{0.f, 0.f};
            if (col < ncol) {
                const int s = col / fz.nsel, j = col - s * fz.nsel;
                if (s > 0) {
                    const size_t ro = (size_t)fz.modes_dev[j] * n;
                    const Cx<R> b = fz.Y[(size_t)(s - 1) * wset + ro + k], a = fz.X[(size_t)s * wset + ro + k];
                    const double pr = fz.theta[2 * (size_t)(col - fz.nsel)], pi = fz.theta[2 * (size_t)(col - fz.nsel) + 1];
                    const double qr = fz.theta[2 * (size_t)col], qi = fz.theta[2 * (size_t)col + 1];
                    v = Zf{(float)((pr * b.re - pi * b.im) - (qr * a.re - qi * a.im)), (float)((pr * b.im + pi * b.re) - (qr * a.im + qi * a.re))};
                }
            }
            Bs[k * PIT_GT + cc] = v;
        }
    } else {
        for (int e = threadIdx.x; e < n * PIT_GT; e += 256) {
            const int k = e / PIT_GT, cc = e - k * PIT_GT;
            Bs[e] = col0 + cc < ncol ? B[(size_t)k * ncol + col0 + cc] : Zf{0.f, 0.f};
        }
    }
    __syncthreads();
    constexpr int RU = PIT_EIGMAX / 32;                       // 3 rows per thread
    Zf acc[RU][PIT_GC];
#pragma unroll
    for (int u = 0; u < RU; u++)
#pragma unroll
        for (int v = 0; v < PIT_GC; v++) acc[u][v] = Zf{0.f, 0.f};
    for (int k = 0; k < n; k++) {
        Zf b[PIT_GC], a[RU];
#pragma unroll
        for (int v = 0; v < PIT_GC; v++) b[v] = Bs[k * PIT_GT + PIT_GC * tx + v];
#pragma unroll
        for (int u = 0; u < RU; u++) a[u] = As[k * PIT_EIGMAX + ty + 32 * u];
#pragma unroll
        for (int u = 0; u < RU; u++)
#pragma unroll
            for (int v = 0; v < PIT_GC; v++) {
                acc[u][v].x += a[u].x * b[v].x - a[u].y * b[v].y;
                acc[u][v].y += a[u].x * b[v].y + a[u].y * b[v].x;
            }
    }
    if (CONJT) {
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const int m = ty + 32 * u;
            if (m < n)
#pragma unroll
                for (int v = 0; v < PIT_GC; v++)
                    if (col0 + PIT_GC * tx + v < ncol) C[(size_t)m * ncol + col0 + PIT_GC * tx + v] = acc[u][v];
        }
    } else {                                                  // X[s] = Y[s] = theta_s X[s] + D[.][col]  (Y: the copy the next pass trains in place)
#pragma unroll
        for (int v = 0; v < PIT_GC; v++) {
            const int col = col0 + PIT_GC * tx + v;
            if (col >= ncol) continue;
            const int s = col / fz.nsel, j = col - s * fz.nsel;
            const size_t base = (size_t)s * wset + (size_t)fz.modes_dev[j] * n;
            const double qr = fz.theta[2 * (size_t)col], qi = fz.theta[2 * (size_t)col + 1];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const int m = ty + 32 * u;
                if (m < n) {
                    const Cx<R> x = fz.X[base + m];
                    const Cx<R> w{(R)(qr * x.re - qi * x.im + acc[u][v].x), (R)(qr * x.im + qi * x.re + acc[u][v].y)};
                    fz.X[base + m] = w;
                    fz.Y[base + m] = w;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ measured coarse model
// What a segment does to a perturbation of its start taps, to first order and on average:  delta' = delta - mu H delta, H the mean
// Jacobian of the update around the converged output.  With dy = x^T delta and de = -G dy (G the real 2 x 2 gain matrix of the error
// function at the output y: its trace / 2 is the holomorphic gain, the rest - different pull along and across y - the anti-holomorphic
// part) H = E[P^T G P].  The symbols are independent, so everything in H that is NOT g Rc (g = E[tr G] / 2, Rc the input covariance) sits
// on ONE direction per output mode: u = Rc w / |Rc w| (the window's response to the symbol under the centre tap: E[conj(x) y]), along which
// a perturbation is an amplitude change (real multiple) or a phase rotation (imaginary multiple) of the output.  For the constant-modulus
// functions these two have gains 3<|y|^4>/<|y|^2> - R (2.76 at 64-QAM) and ~0 against g = 0.62 for everything else: a model with ONE gain
// per eigen-direction of Rc (round 3) is off by 0.3-0.5 in the segment map along u, and that mismatch - not the sample covariance of a
// segment - was the contraction floor of the passes (0.35-0.45 per pass; mcma, whose amplitude direction had no extra damping, stalled).
// The model is therefore MEASURED per sweep on PIT_MODW windows of the capture at the sweep's seed taps:
//   g        = E[tr G] / 2                                  (replaces the per-method formulas of pit_gain)
//   K (2x2)  = E[Z^T G Z],  Z = the 2 x 2 real form of z = x^T u:  the restriction of H to span{u, i u}
// and the scan of the corrections treats the component of a defect along u with exp(-mu T K) (two scalar recurrences in the eigenbasis of
// K), the rest with exp(-mu g T lambda_k) as before.  G per error function (pythran_equalisation.py:178-231), decisions held.
constexpr int PIT_MODW = 8192, PIT_MODB = 128;                 // sample windows, blocks (4 waves x 4 window groups each: 4 windows per group and mode)
struct PitModel { float g, k1, k2, q11, q12, q21, q22, lam_u; int ok; int pad[3]; };     // per selected mode: gain, eigenvalues of K, Q (columns = eigenvectors), <|z|^2>
template <typename R> struct PitModelArgs {
    const Cx<R> *E, *wx, *symbols;
    int64_t L, TrSyms, nsy, sy_pitch;
    int nmodes, ntaps, os, nsel, method, tables;               // tables: symbols hold codes / partitions in the rde / mrde layout (also the slicer tables)
    int64_t modes[16];
    float *part;                                               // [PIT_MODB blocks][nsel][2 * PIT_EIGMAX + 8] partial sums
    Cx<R> *u;                                                  // [nsel][ntot]: phase A writes p = sum conj(x) y normalised, phase B reads it
    PitModel *model;                                           // [nsel]
    unsigned *ticket;
    // The windows sit in the HEADS of the segments, and for the phase-sensitive error functions their outputs are turned by the
    // segment's phase seed rot[s] (pit_phase_kernel / pit_unwrap_kernel: the 4th-power phase of the seed taps' output there) - G depends on
    // where y sits relative to the axes, and with a drifting carrier the seed taps' output is only aligned after that rotation (measured
    // without it on 16-QAM / mcma / 50 kHz linewidth: a model so wrong that the passes diverged).  rot == nullptr: no rotation.
    PitSeg sg;
    const double *rot;
};
// G = [[g11, g12], [g12, g22]] at the output y
constexpr int PIT_MODTAB = 128;                              // table entries staged in LDS (more: the constants only)
__device__ __forceinline__ void pit_gain_matrix(int method, float yr, float yi, const Zf *sy, int nsy, int tables, float &g11, float &g12, float &g22)
{
    const float y2 = yr * yr + yi * yi;
    const Zf c0 = sy[0];
    const int ncode = (nsy + 1) / 2, npart = nsy - ncode;
    auto look = [&](float v, bool im) -> float {                 // partition_value: the code above the last partition below v
        float r = im ? c0.y : c0.x;
        for (int p = 0; p < npart; p++) { const Zf pt = sy[ncode + p], cd = sy[p + 1]; if (v > (im ? pt.y : pt.x)) r = im ? cd.y : cd.x; }
        return r;
    };
    g12 = 0.f;
    switch (method) {
    case QH_M_CMA: case QH_M_SGNCMA: { const float r = c0.x; g11 = 3 * yr * yr + yi * yi - r; g22 = yr * yr + 3 * yi * yi - r; g12 = 2 * yr * yi; break; }
    case QH_M_MCMA: g11 = 3 * yr * yr - c0.x; g22 = 3 * yi * yi - c0.y; break;
    case QH_M_RDE: { const float r = tables ? look(y2, false) : c0.x; g11 = 3 * yr * yr + yi * yi - r; g22 = yr * yr + 3 * yi * yi - r; g12 = 2 * yr * yi; break; }
    case QH_M_MRDE: { const float rr = tables ? look(yr * yr, false) : c0.x, ri = tables ? look(yi * yi, true) : c0.y; g11 = 3 * yr * yr - rr; g22 = 3 * yi * yi - ri; break; }
    case QH_M_SBD: { const float sr = look(yr, false), si = look(yi, true); g11 = fabsf(sr); g22 = fabsf(si); break; }
    case QH_M_MDDMA: { const float sr = look(yr, false), si = look(yi, true); g11 = 3 * yr * yr - sr * sr; g22 = 3 * yi * yi - si * si; break; }
    default: g11 = 1.f; g22 = 1.f; break;                      // dd
    }
}
// PHASE 0: p = sum conj(x_i) y_i (-> u), g.  PHASE 1: K, <|z|^2>.  FOUR windows per wave at a time (16 lanes each, up to 8 consecutive taps
// per lane, 4-step shuffle reductions): a window is one round trip to memory plus a chain of reductions, so what counts is how many are
// in flight (one window per wave: 70 us per launch; so: ~10).
template <typename R, int PHASE>
__global__ void __launch_bounds__(256) pit_model_kernel(PitModelArgs<R> a)
{
    QH_WAVE_FIRST();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, l16 = lane & 15;
    const int gg = (blockIdx.x * 4 + wave) * 4 + grp, ng = gridDim.x * 16;            // window groups of the launch
    const int ntot = a.nmodes * a.ntaps;
    constexpr int TP = PIT_EIGMAX / 16;                          // taps per lane (slots)
    const int pw = 2 * PIT_EIGMAX + 8;                           // floats per (block, mode) partial
    const int tpl = (ntot + 15) / 16;                            // taps per lane in use
    for (int j = 0; j < a.nsel; j++) {
        const int mode = (int)a.modes[j];
        __shared__ Zf tab[PIT_MODTAB];                                // this mode's constants / tables (read per window: not from global memory)
        const int nsy_l = a.nsy <= PIT_MODTAB ? (int)a.nsy : 1;
        const int tables_l = a.nsy <= PIT_MODTAB ? a.tables : 0;
        __syncthreads();
        for (int e = threadIdx.x; e < nsy_l; e += 256) { const Cx<R> v = a.symbols[(size_t)mode * a.sy_pitch + e]; tab[e] = Zf{(float)v.re, (float)v.im}; }
        __syncthreads();
        Zf wv[TP], uv[TP], pacc[TP];
        int64_t xoff[TP];                                           // where tap f of a window sits relative to the window's first sample
        float gacc = 0.f, k11 = 0.f, k12 = 0.f, k22 = 0.f, zz = 0.f, cnt = 0.f;
#pragma unroll
        for (int q = 0; q < TP; q++) {
            const int f = l16 * tpl + q;
            const bool ok = q < tpl && f < ntot;
            const int k = ok ? f / a.ntaps : 0;
            xoff[q] = ok ? (int64_t)k * a.L + (f - k * a.ntaps) : -1;
            wv[q] = Zf{0.f, 0.f}; uv[q] = Zf{0.f, 0.f}; pacc[q] = Zf{0.f, 0.f};
            if (ok) { const Cx<R> t = a.wx[(size_t)mode * ntot + f]; wv[q] = Zf{(float)t.re, (float)t.im}; if (PHASE == 1) { const Cx<R> v = a.u[(size_t)j * ntot + f]; uv[q] = Zf{(float)v.re, (float)v.im}; } }
        }
        int64_t st = ((int64_t)PIT_SEEDWIN * a.sg.S) / PIT_MODW;
        st = st < 1 ? 1 : (st > 37 ? 37 : st);
        for (int wi0 = gg - grp; wi0 < PIT_MODW; wi0 += ng) {     // (wave-uniform trip count; a group without a valid window idles through it)
            // window wi: segment wi mod S, the (wi / S)-th of the sampled steps of its head (the first PIT_SEEDWIN steps, up to 37 steps apart)
            const int wi = wi0 + grp;
            const int sgi = wi % a.sg.S;
            const int64_t off = (int64_t)(wi / a.sg.S) * st;
            const int64_t i = a.sg.start(sgi) + off;
            const bool valid = wi < PIT_MODW && off < PIT_SEEDWIN && off < a.sg.steps(sgi) && i < a.TrSyms;
            float rc = 1.f, rs = 0.f;
            if (a.rot && valid) { rc = (float)a.rot[2 * ((size_t)sgi * a.nsel + j)]; rs = (float)a.rot[2 * ((size_t)sgi * a.nsel + j) + 1]; }
            Zf xv[TP];
            float yr = 0.f, yi = 0.f, zr = 0.f, zi = 0.f;
#pragma unroll
            for (int q = 0; q < TP; q++) {
                xv[q] = Zf{0.f, 0.f};
                if (valid && xoff[q] >= 0) { const Cx<R> v = a.E[xoff[q] + i * a.os]; xv[q] = Zf{(float)v.re, (float)v.im}; }
            }
#pragma unroll
            for (int q = 0; q < TP; q++) {
                yr += xv[q].x * wv[q].x - xv[q].y * wv[q].y; yi += xv[q].x * wv[q].y + xv[q].y * wv[q].x;
                if (PHASE == 1) { zr += xv[q].x * uv[q].x - xv[q].y * uv[q].y; zi += xv[q].x * uv[q].y + xv[q].y * uv[q].x; }
            }
            for (int o = 8; o > 0; o >>= 1) { yr += __shfl_xor(yr, o); yi += __shfl_xor(yi, o); if (PHASE == 1) { zr += __shfl_xor(zr, o); zi += __shfl_xor(zi, o); } }
            const float yr0 = yr, yi0 = yi;                                                  // (u = E[conj(x) y] belongs to the seed taps as they are)
            { const float t = yr * rc - yi * rs; yi = yr * rs + yi * rc; yr = t; }            // the segment's frame (seed taps x rot[s])
            if (PHASE == 1) { const float t = zr * rc - zi * rs; zi = zr * rs + zi * rc; zr = t; }
            float g11, g12, g22;
            pit_gain_matrix(a.method, yr, yi, tab, nsy_l, tables_l, g11, g12, g22);
            if (valid) {
                if (PHASE == 0) {
#pragma unroll
                    for (int q = 0; q < TP; q++) { pacc[q].x += xv[q].x * yr0 + xv[q].y * yi0; pacc[q].y += xv[q].x * yi0 - xv[q].y * yr0; }   // conj(x) y
                    gacc += 0.5f * (g11 + g22);
                } else {
                    k11 += zr * zr * g11 + 2 * zr * zi * g12 + zi * zi * g22;
                    k22 += zi * zi * g11 - 2 * zr * zi * g12 + zr * zr * g22;
                    k12 += -zr * zi * g11 + (zr * zr - zi * zi) * g12 + zr * zi * g22;
                    zz += zr * zr + zi * zi;
                }
                cnt += 1.f;
            }
        }
        // the four window groups of the wave add up (shuffles), then the four waves of the block in LDS (fixed order): one partial per block and mode
        __shared__ float blk[4][2 * PIT_EIGMAX + 8];
        for (int e = lane; e < 2 * PIT_EIGMAX + 8; e += 64) blk[wave][e] = 0.f;
        if (PHASE == 0) {
#pragma unroll
            for (int q = 0; q < TP; q++) {
                float pr = pacc[q].x, pi = pacc[q].y;
                pr += __shfl_xor(pr, 16); pi += __shfl_xor(pi, 16); pr += __shfl_xor(pr, 32); pi += __shfl_xor(pi, 32);
                const int f = l16 * tpl + q;
                if (grp == 0 && q < tpl && f < ntot) { blk[wave][2 * f] = pr; blk[wave][2 * f + 1] = pi; }
            }
            gacc += __shfl_xor(gacc, 16); gacc += __shfl_xor(gacc, 32); cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 32);
            if (lane == 0) { blk[wave][2 * PIT_EIGMAX] = gacc; blk[wave][2 * PIT_EIGMAX + 1] = cnt; }
        } else {
            for (int o = 16; o <= 32; o <<= 1) { k11 += __shfl_xor(k11, o); k12 += __shfl_xor(k12, o); k22 += __shfl_xor(k22, o); zz += __shfl_xor(zz, o); cnt += __shfl_xor(cnt, o); }
            if (lane == 0) { blk[wave][0] = k11; blk[wave][1] = k12; blk[wave][2] = k22; blk[wave][3] = zz; blk[wave][4] = cnt; }
        }
        __syncthreads();
        float *dst = a.part + ((size_t)blockIdx.x * a.nsel + j) * pw;
        const int nval = PHASE == 0 ? 2 * PIT_EIGMAX + 2 : 5;
        for (int e = threadIdx.x; e < nval; e += 256) dst[e] = (blk[0][e] + blk[1][e]) + (blk[2][e] + blk[3][e]);
        __syncthreads();
    }
    // the block that finishes last adds the partial sums up (fixed order: reproducible) and writes the result
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); const unsigned t = atomicAdd(a.ticket, 1u); last = t == gridDim.x - 1; if (last) { *a.ticket = 0; __threadfence(); } }
    __syncthreads();
    if (!last) return;
    __shared__ float red[256];
    const int nblk = gridDim.x;
    // sum over the blocks' partials of element e of mode j: loads batched 16 at a time (one by one they were 0.25 ms of latency)
    auto psum = [&](int j, int e) -> float {
        float acc = 0.f;
        for (int b0 = 0; b0 < nblk; b0 += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = b0 + q < nblk ? a.part[((size_t)(b0 + q) * a.nsel + j) * pw + e] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; q++) acc += v[q];
        }
        return acc;
    };
    __shared__ float vals[2 * PIT_EIGMAX + 8];
    const int nval = PHASE == 0 ? 2 * PIT_EIGMAX + 2 : 5;
    for (int j = 0; j < a.nsel; j++) {
        // every value of this mode summed over the blocks' partials by its own thread (fixed order: reproducible)
        for (int e = threadIdx.x; e < nval; e += 256) vals[e] = (PHASE == 1 || e >= 2 * PIT_EIGMAX || e < 2 * ntot) ? psum(j, e) : 0.f;
        __syncthreads();
        if (PHASE == 0) {
            float nrm = 0.f;
            for (int f = threadIdx.x; f < ntot; f += 256) nrm += vals[2 * f] * vals[2 * f] + vals[2 * f + 1] * vals[2 * f + 1];
            red[threadIdx.x] = nrm; __syncthreads();
            for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
            const float inv = red[0] > 0.f ? rsqrtf(red[0]) : 0.f;
            for (int f = threadIdx.x; f < ntot; f += 256) a.u[(size_t)j * ntot + f] = Cx<R>{(R)(vals[2 * f] * inv), (R)(vals[2 * f + 1] * inv)};
            if (threadIdx.x == 0) {
                const float gs = vals[2 * PIT_EIGMAX], cs = vals[2 * PIT_EIGMAX + 1];
                a.model[j].g = cs > 0.f ? gs / cs : 0.f;
                a.model[j].ok = 0;
            }
        } else if (threadIdx.x == 0) {
            float k11 = vals[0], k12 = vals[1], k22 = vals[2], zz = vals[3];
            const float cs = vals[4];
            const float n = cs > 0.f ? 1.f / cs : 0.f;
            k11 *= n; k12 *= n; k22 *= n; zz *= n;
            // symmetric 2 x 2 eigen-decomposition: K = Q diag(k1, k2) Q^T
            const float tr = 0.5f * (k11 + k22), df = 0.5f * (k11 - k22), rad = sqrtf(df * df + k12 * k12);
            float c = 1.f, sn = 0.f;
            if (rad > 1e-12f * (fabsf(tr) + 1e-30f)) { const float ang = 0.5f * atan2f(k12, df); c = cosf(ang); sn = sinf(ang); }
            PitModel m = a.model[j];
            m.k1 = tr + rad; m.k2 = tr - rad; m.q11 = c; m.q21 = sn; m.q12 = -sn; m.q22 = c; m.lam_u = zz;
            m.ok = (m.g > 0.f && m.g == m.g && m.k1 == m.k1 && m.k2 == m.k2) ? 1 : 0;
            a.model[j] = m;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ eigen-space analysis (single precision)
// The analysis of a pass needs the boundary states in the eigenbasis anyway (defect vectors -> scan -> correction).  For complex64
// everything between the two basis products therefore lives THERE: x~[s] = V^H X[s] (start taps, kept up to date through the passes:
// x~ <- theta x~ + D~, one product per sweep) and y~[s] = V^H Y[s] (end taps, one product per pass).  Boundary defects, gauge
// elements, output powers and the deviation estimate are lambda-weighted sums over the 82 components of a column
// (E[(x^T b) conj(x^T a)] = a~^H Lambda b~) - no probe of the capture, no filtering: the 35 us probe kernel of round 2 becomes a 5 us
// reduction, and the products themselves are blocked for registers (3 rows x 4 columns per thread, 32 columns per block:
// 5 LDS reads per 12 complex multiply-adds instead of 5 per 6; 42 -> ~15 us at 7936 columns).
// The two basis products on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact single precision at the vector
// rate, MI355X_MICROARCH.md) - the one place on this path where the work IS a dense GEMM: (n x n) x (n x S nsel), n = nmodes ntaps = 82,
// thousands of columns, and what matters is its LATENCY (it sits between two passes).  A block takes 16 columns, wave w of its
// ceil(n / 16) waves the output rows 16 w .. 16 w + 15: 224 blocks of 6 waves at C3 - every CU busy, 23 k-steps x 4 MFMAs x 32 cycles
// = 1.2 us of MFMA issue per wave (the 32x32x2 tiling of the first version: 112 blocks, 4.8 us per wave, 18-22 us per product).
// Complex product from three real accumulators: rr += Ar Br, ii += Ai Bi, im += Ar Bi + Ai Br (C = rr - ii + i im).  Operand
// layout: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]; C/D: column lane & 15, row 4 (lane >> 4) + reg.
typedef float pit_f4 __attribute__((ext_vector_type(4)));
constexpr int PIT_MC = 16, PIT_MBP = PIT_MC + 1;               // columns per block, row pitch of the B tile in LDS
inline int pit_mfma_threads(int n) { return 64 * ((n + 15) / 16); }
inline size_t pit_mfma_lds(int n) { return ((size_t)(n + 4) * PIT_MBP + (size_t)PIT_MC * (PIT_EIGMAX + 1)) * sizeof(Zf); }
template <typename R, int MODE>
__global__ void __launch_bounds__(64 * (PIT_EIGMAX / 16)) pit_basis_mfma_kernel(const Zf *__restrict__ A2, const Cx<R> *__restrict__ T, const Zf *__restrict__ D, Zf *__restrict__ Out,
                                                                               int n, int ncol, const PitCtrl *c, PitFuse<R> fz)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    extern __shared__ __attribute__((aligned(16))) char pit_smem[];
    const int np = (n + 3) & ~3;                              // contraction length padded to whole MFMA steps (zero rows)
    Zf *Bs = reinterpret_cast<Zf *>(pit_smem);                // [np][PIT_MBP]
    Zf *Ct = Bs + (size_t)(n + 4) * PIT_MBP;                  // [PIT_MC][PIT_EIGMAX + 1] (MODE 1: transposition of the result)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    const int nb1 = (ncol + PIT_MC - 1) / PIT_MC;
    const bool second = MODE == 0 && (int)blockIdx.x >= nb1;   // (block-uniform: the second array of a two-array forward product)
    if (second) { T = fz.T2; Out = reinterpret_cast<Zf *>(fz.Out2); }
    const int col0 = ((int)blockIdx.x - (second ? nb1 : 0)) * PIT_MC;
    const size_t wset = (size_t)fz.nmodes * n;
    // op(A) - 54 KB, the same for every block and launch, L2 resident - goes from global memory straight into the A operand
    // (lane: row 16 wave + (lane & 15), k + (lane >> 4); 8 steps prefetched), the B tile through LDS (shared by the waves)
    const int am = 16 * wave + (lane & 15), ak = lane >> 4;
    const Zf *ag = A2 + (size_t)ak * n + am;
    auto lda = [&](int k) -> Zf { return (am < n && k + ak < n) ? ag[(size_t)k * n] : Zf{0.f, 0.f}; };
    constexpr int PF = 8;
    Zf abuf[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) abuf[i] = lda(4 * i);
    constexpr int NCW = 3;                                      // columns per wave in the column-wise phases: ceil(16 / nw), nw >= 6 ... (nw < 6: loop)
    if (MODE == 0) {
        // wave w: columns w, w + nw, ...; lanes: f = lane, lane + 64 (one column's taps are contiguous: coalesced); all loads in flight together
        for (int cb = 0; cb < PIT_MC; cb += NCW * nw) {
            Zf tmp[NCW][2];
#pragma unroll
            for (int q = 0; q < NCW; q++) {
                const int cc = cb + wave + nw * q, col = col0 + cc;
                tmp[q][0] = Zf{0.f, 0.f}; tmp[q][1] = Zf{0.f, 0.f};
                if (cc < PIT_MC && col < ncol) {
                    const int s = col / fz.nsel, j = col - s * fz.nsel;               // wave-uniform
                    const Cx<R> *src = T + (size_t)s * wset + (size_t)fz.modes_dev[j] * n;
                    if (lane < n) { const Cx<R> t = src[lane]; tmp[q][0] = Zf{(float)t.re, (float)t.im}; }
                    if (lane + 64 < n) { const Cx<R> t = src[lane + 64]; tmp[q][1] = Zf{(float)t.re, (float)t.im}; }
                }
            }
#pragma unroll
            for (int q = 0; q < NCW; q++) {
                const int cc = cb + wave + nw * q;
                if (cc < PIT_MC) {
                    if (lane < np) Bs[lane * PIT_MBP + cc] = tmp[q][0];
                    if (lane + 64 < np) Bs[(lane + 64) * PIT_MBP + cc] = tmp[q][1];
                }
            }
        }
    } else {
        // thread: column tid & 15, rows tid >> 4, + nt / 16, ...: a row of the tile is 128 contiguous bytes; up to 8 rows of a thread in flight
        const int cc = tid & 15, k0 = tid >> 4, kstep = nt >> 4;
        const bool okc = col0 + cc < ncol;
        for (int kb = 0; kb < np; kb += 8 * kstep) {
            Zf tmp[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int k = kb + k0 + kstep * q;
                tmp[q] = (okc && k < n) ? D[(size_t)k * ncol + col0 + cc] : Zf{0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int k = kb + k0 + kstep * q;
                if (k < np) Bs[k * PIT_MBP + cc] = tmp[q];
            }
        }
    }
    // MODE 1: the start taps and frames the epilogue combines with the product are fetched now, under the MFMAs (nw >= 6 waves: up to 3
    // columns per wave; fewer waves - small n - fetch in the epilogue)
    const bool pre = MODE == 1 && NCW * nw >= PIT_MC;
    Cx<R> xv[MODE == 1 ? NCW : 1][2];
    double th[MODE == 1 ? NCW : 1][2];
    if (pre) {
#pragma unroll
        for (int q = 0; q < NCW; q++) {
            const int cc = wave + nw * q, col = col0 + cc;
            xv[q][0] = Cx<R>{0, 0}; xv[q][1] = Cx<R>{0, 0}; th[q][0] = 1; th[q][1] = 0;
            if (cc < PIT_MC && col < ncol) {
                const int s = col / fz.nsel, j = col - s * fz.nsel;
                const size_t base = (size_t)s * wset + (size_t)fz.modes_dev[j] * n;
                th[q][0] = fz.theta[2 * (size_t)col]; th[q][1] = fz.theta[2 * (size_t)col + 1];
                if (lane < n) xv[q][0] = fz.X[base + lane];
                if (lane + 64 < n) xv[q][1] = fz.X[base + lane + 64];
            }
        }
    }
    __syncthreads();
    pit_f4 rr = {0.f, 0.f, 0.f, 0.f}, ii = rr, im = rr;
    const Zf *bp = Bs + (lane >> 4) * PIT_MBP + (lane & 15);
    for (int k0 = 0; k0 < np; k0 += 4 * PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const int k = k0 + 4 * i;
            if (k < np) {                                       // block-uniform
                Zf a = abuf[i];
                abuf[i] = lda(k + 4 * PF);
                if (MODE == 0) a.y = -a.y;                      // V^H
                const Zf b = bp[(size_t)k * PIT_MBP];
                rr = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, rr, 0, 0, 0);
                ii = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, ii, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.y, im, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, im, 0, 0, 0);
            }
        }
    }
    const int cl = lane & 15;
    if (MODE == 0) {
        const int col = col0 + cl;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int m = 16 * wave + 4 * (lane >> 4) + q;
            if (m < n && col < ncol) Out[(size_t)m * ncol + col] = Zf{rr[q] - ii[q], im[q]};      // 16 consecutive columns per quarter wave
        }
    } else {
        // X[s] = Y[s] = theta_s X[s] + C[.][col]: through LDS, so that the tap sets are written along m (their fast axis)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int m = 16 * wave + 4 * (lane >> 4) + q;
            Ct[cl * (PIT_EIGMAX + 1) + m] = Zf{rr[q] - ii[q], im[q]};
        }
        __syncthreads();
        // wave: columns wave, wave + nw, ...; lanes along m (the tap sets' fast axis)
        for (int cb = 0; cb < PIT_MC; cb += NCW * nw) {
#pragma unroll
            for (int q = 0; q < NCW; q++) {
                const int cc = cb + wave + nw * q, col = col0 + cc;
                if (cc < PIT_MC && col < ncol) {
                    const int s = col / fz.nsel, j = col - s * fz.nsel;
                    const size_t base = (size_t)s * wset + (size_t)fz.modes_dev[j] * n;
                    const double qr = pre ? th[q][0] : fz.theta[2 * (size_t)col], qi = pre ? th[q][1] : fz.theta[2 * (size_t)col + 1];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int m = lane + 64 * h;
                        if (m < n) {
                            const Cx<R> x = pre ? xv[q][h] : fz.X[base + m];
                            const Zf d = Ct[cc * (PIT_EIGMAX + 1) + m];
                            const Cx<R> w{(R)(qr * x.re - qi * x.im + fz.damp * d.x), (R)(qr * x.im + qi * x.re + fz.damp * d.y)};
                            fz.X[base + m] = w;
                            fz.Y[base + m] = w;
                        }
                    }
                }
            }
        }
    }
}

// Boundary b = 1 .. S-1 of mode j (one wave each; entries (b-1) nsel + j): a = x~[b] against b = y~[b-1], lambda-weighted:
// best group element g (y_B ~ g y_A), defect |b - g a|_Lambda / |b|_Lambda, output power |b|^2_Lambda.  nsel extra rows: how far the
// sweep's RESULT moved - y~[S-1] of this pass against the previous pass's (pass 0: against the start taps of the last segment).
// ualpha != nullptr (measured model, see pit_model_kernel): per boundary the component of the local defect along the signal direction
// u = Lambda y~[s-1] / |Lambda y~[s-1]| (Rc w in the eigenbasis) - alpha = u^H (y~[s-1] - g x~[s]), the same number in every frame because
// u turns with the taps it is made of - and 1 / |Lambda y~[s-1]|: ualpha[bi] = (Re alpha, Im alpha, 1 / norm, 0).
static __global__ void __launch_bounds__(256) pit_bound_kernel(const Zf *Xe, const Zf *Ye, const Zf *Yprev, const double *lam, int n, int S, int nsel, int sym,
                                                              const PitCtrl *c, double *dfc, double *pw, double *gph, float4 *ualpha = nullptr)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    const int lane = threadIdx.x & 63, bi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nb = (S - 1) * nsel, ncol = S * nsel;
    if (bi >= nb + nsel) return;
    const bool result_probe = bi >= nb;
    const int j = result_probe ? bi - nb : bi % nsel;
    const int sb = result_probe ? S - 1 : bi / nsel;             // segment whose end taps are `b`
    const Zf *pa = result_probe ? (c->passes == 0 ? Xe + (size_t)(S - 1) * nsel + j : Yprev + j) : Xe + (size_t)(sb + 1) * nsel + j;
    const size_t sa = (result_probe && c->passes != 0) ? (size_t)nsel : (size_t)ncol;
    const Zf *pb = Ye + (size_t)sb * nsel + j;
    Zf av[(PIT_EIGMAX + 63) / 64], bv[(PIT_EIGMAX + 63) / 64];
    float lv[(PIT_EIGMAX + 63) / 64];
    double A = 0, B = 0, Cr = 0, Ci = 0;
#pragma unroll
    for (int q = 0; q < (PIT_EIGMAX + 63) / 64; q++) {
        const int k = lane + 64 * q;
        av[q] = Zf{0.f, 0.f}; bv[q] = Zf{0.f, 0.f}; lv[q] = 0.f;
        if (k < n) {
            av[q] = pa[(size_t)k * sa]; bv[q] = pb[(size_t)k * ncol];
            const float l = (float)lam[k];
            lv[q] = l > 0.f ? l : 0.f;
        }
        A += (double)lv[q] * ((double)av[q].x * av[q].x + (double)av[q].y * av[q].y);
        B += (double)lv[q] * ((double)bv[q].x * bv[q].x + (double)bv[q].y * bv[q].y);
        Cr += (double)lv[q] * ((double)bv[q].x * av[q].x + (double)bv[q].y * av[q].y);      // b conj(a)
        Ci += (double)lv[q] * ((double)bv[q].y * av[q].x - (double)bv[q].x * av[q].y);
    }
    for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o); B += __shfl_xor(B, o); Cr += __shfl_xor(Cr, o); Ci += __shfl_xor(Ci, o); }
    double gr = 1, gi = 0;
    if (sym == 0) {
        const double pr = sqrt(Cr * Cr + Ci * Ci);
        if (pr > 0) { gr = Cr / pr; gi = Ci / pr; }
    } else if (sym == 4 || sym == 2 || sym == 1) {              // nearest of +-1 (, +-i)
        if (sym == 4 && fabs(Ci) > fabs(Cr)) { gr = 0; gi = Ci >= 0 ? 1 : -1; }
        else if (sym == 1) gr = 1;
        else gr = Cr >= 0 ? 1 : -1;
    } else {
        const double step = 6.283185307179586 / sym;
        const double kk = rint(atan2(Ci, Cr) / step) * step;
        gr = cos(kk); gi = sin(kk);
    }
    // the defect from the DIFFERENCES (A + B - 2 Re(conj(g) C) cancels to nothing in single precision once the passes agree)
    double d2 = 0;
    const float grf = (float)gr, gif = (float)gi;
#pragma unroll
    for (int q = 0; q < (PIT_EIGMAX + 63) / 64; q++) {
        const float dr = bv[q].x - (grf * av[q].x - gif * av[q].y), di = bv[q].y - (grf * av[q].y + gif * av[q].x);
        d2 += (double)lv[q] * ((double)dr * dr + (double)di * di);
    }
    for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o);
    if (ualpha && !result_probe) {
        float nn = 0.f, ar = 0.f, ai = 0.f;
#pragma unroll
        for (int q = 0; q < (PIT_EIGMAX + 63) / 64; q++) {
            const float dr = bv[q].x - (grf * av[q].x - gif * av[q].y), di = bv[q].y - (grf * av[q].y + gif * av[q].x);
            const float ur = lv[q] * bv[q].x, ui = lv[q] * bv[q].y;                // Lambda y~ (unnormalised u)
            nn += ur * ur + ui * ui;
            ar += ur * dr + ui * di;                                                // conj(u) d
            ai += ur * di - ui * dr;
        }
        for (int o = 32; o > 0; o >>= 1) { nn += __shfl_xor(nn, o); ar += __shfl_xor(ar, o); ai += __shfl_xor(ai, o); }
        const float inv = nn > 0.f ? rsqrtf(nn) : 0.f;
        if (lane == 0) ualpha[bi] = float4{ar * inv, ai * inv, inv, 0.f};
    }
    if (lane == 0) {
        if (!result_probe) { gph[2 * (size_t)bi] = gr; gph[2 * (size_t)bi + 1] = gi; pw[bi] = B; }
        dfc[bi] = (A == A && B == B) ? sqrt(d2 / (B > 1e-300 ? B : 1e-300)) : 1e30;
    }
}

// The scan of pit_recur_kernel on defect vectors formed on the fly, d~_k[s] = theta_{s-1} y~_k[s-1] - theta_s x~_k[s] (0 for s = 0), and the
// eigen-space start taps of the next pass: x~_k[s] <- theta_s x~_k[s] + D~_k[s] (what the back product adds to X, V being unitary).
constexpr int PIT_RT = 1024;            // threads of the eigen-space scan: up to 4 segments per thread stay in registers (S <= 4096: one round of loads)
// ualpha / uq != nullptr (measured model): the component of every defect along u[s] = theta_{s-1} Lambda y~[s-1] / |Lambda y~[s-1]| is taken out
// before the scan (d_perp = d - u alpha) and its own scan - pit_gauge_kernel's q - put back afterwards: D~[s] = scan(d_perp) + u[s] q[s].
template <typename R>
__global__ void __launch_bounds__(PIT_RT) pit_recur_eig_kernel(Zf *Xe, const Zf *Ye, Zf *D, const double *theta, const double *lam, int nsel, int S, int64_t T, const R *mu,
                                                               double beta, const PitCtrl *c, const float *Msum = nullptr, float damp = 1.f,
                                                               const float4 *ualpha = nullptr, const float2 *uq = nullptr)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    __shared__ float4 aff[2 * PIT_RT];
    const int k = blockIdx.x, j = blockIdx.y;
    const int ncol = S * nsel;
    Zf *row = D + (size_t)k * ncol + j, *xr = Xe + (size_t)k * ncol + j;
    const Zf *yr = Ye + (size_t)k * ncol + j;
    double a = (double)*mu * c->gain * (double)T * lam[k];
    if (a < 0) a = 0;
    const float coef = c->corr_on ? (float)exp(-a * (1 + beta * a)) : 0.f;       // (beta: see pit_recur_kernel)
    // adaptive step: the step sizes a segment used add up to Msum[s] instead of mu T - one coefficient per segment, c_s takes D[s] to D[s + 1]
    const float gl = (float)(c->gain * lam[k] > 0 ? c->gain * lam[k] : 0.0), betaf = (float)beta;
    const bool corr = c->corr_on != 0;
    auto cf = [&](int s) -> float {                              // coefficient of segment s (s >= 0)
        if (!Msum) return coef;
        const float as = gl * Msum[(size_t)s * nsel + j];
        return corr ? __expf(-as * (1.f + betaf * as)) : 0.f;
    };
    const int len = (S + PIT_RT - 1) / PIT_RT;
    const int s0 = threadIdx.x * len, s1 = s0 + len < S ? s0 + len : S;
    auto th = [&](int s) { return Zf{(float)theta[2 * ((size_t)s * nsel + j)], (float)theta[2 * ((size_t)s * nsel + j) + 1]}; };
    auto scan_op = [](float4 e, float4 l) { return float4{l.x * e.x, l.x * e.y + l.y, l.x * e.z + l.z, 0.f}; };
    if (len <= 4) {
        // everything a thread touches is loaded once, together, and stays in registers across the scan
        Zf y[4], x[4], t[5], r[4];
        t[0] = s0 > 0 && s0 < S ? th(s0 - 1) : Zf{1.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = s0 + q;
            const bool ok = s < s1;
            y[q] = (ok && s > 0) ? yr[(size_t)(s - 1) * nsel] : Zf{0.f, 0.f};
            x[q] = ok ? xr[(size_t)s * nsel] : Zf{0.f, 0.f};
            t[q + 1] = ok ? th(s) : Zf{1.f, 0.f};
        }
        Zf run{0.f, 0.f};
        float cq[4], clen = 1.f;
        float4 ua[4];
        float2 qq[4];
        const float lk = (float)(lam[k] > 0 ? lam[k] : 0.0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = s0 + q;
            cq[q] = (s < s1 && s > 0) ? cf(s - 1) : (s < s1 ? 0.f : 1.f);
            ua[q] = (ualpha && s < s1 && s > 0) ? ualpha[(size_t)(s - 1) * nsel + j] : float4{0.f, 0.f, 0.f, 0.f};
            qq[q] = (uq && s < s1 && s > 0) ? uq[(size_t)s * nsel + j] : float2{0.f, 0.f};
        }
        Zf uk[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = s0 + q;
            uk[q] = Zf{0.f, 0.f};
            if (s < s1) {
                x[q] = cmulf(t[q + 1], x[q]);                   // theta_s x~[s]
                Zf d{0.f, 0.f};
                if (s > 0) {
                    const Zf aa = cmulf(t[q], y[q]);
                    d = Zf{aa.x - x[q].x, aa.y - x[q].y};
                    uk[q] = Zf{aa.x * lk * ua[q].z, aa.y * lk * ua[q].z};          // component k of u[s] (frame 0)
                    const Zf ual = cmulf(uk[q], Zf{ua[q].x, ua[q].y});
                    d = Zf{d.x - ual.x, d.y - ual.y};                              // d_perp
                }
                run = Zf{d.x + cq[q] * run.x, d.y + cq[q] * run.y};
                r[q] = run;
                clen *= cq[q];
            }
        }
        const float4 comp = block_scan_excl<float4, decltype(scan_op), PIT_RT>(float4{clen, run.x, run.y, 0.f}, scan_op, float4{1.f, 0.f, 0.f, 0.f}, aff);
        float pw = 1.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = s0 + q;
            if (s < s1) {
                pw *= cq[q];
                const Zf uqv = cmulf(uk[q], Zf{qq[q].x, qq[q].y});
                const Zf v{r[q].x + pw * comp.y + uqv.x, r[q].y + pw * comp.z + uqv.y};
                row[(size_t)s * nsel] = v;
                xr[(size_t)s * nsel] = Zf{x[q].x + damp * v.x, x[q].y + damp * v.y};
            }
        }
        return;
    }
    // long sweeps: batches of 4 segments (their loads are issued together), running values through memory
    Zf run{0.f, 0.f};
    float clen = 1.f;
    for (int sb = s0; sb < s1; sb += 4) {
        Zf y[4], x[4], t[5];
        t[0] = sb > 0 ? th(sb - 1) : Zf{1.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = sb + q;
            const bool ok = s < s1;
            y[q] = (ok && s > 0) ? yr[(size_t)(s - 1) * nsel] : Zf{0.f, 0.f};
            x[q] = ok ? xr[(size_t)s * nsel] : Zf{0.f, 0.f};
            t[q + 1] = ok ? th(s) : Zf{1.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = sb + q;
            if (s < s1) {
                Zf d{0.f, 0.f};
                if (s > 0) {
                    const Zf aa = cmulf(t[q], y[q]), bb = cmulf(t[q + 1], x[q]);
                    d = Zf{aa.x - bb.x, aa.y - bb.y};
                    if (ualpha) {
                        const float4 al = ualpha[(size_t)(s - 1) * nsel + j];
                        const float lk2 = (float)(lam[k] > 0 ? lam[k] : 0.0);
                        const Zf ual = cmulf(Zf{aa.x * lk2 * al.z, aa.y * lk2 * al.z}, Zf{al.x, al.y});
                        d = Zf{d.x - ual.x, d.y - ual.y};
                    }
                }
                const float cs = s > 0 ? cf(s - 1) : 0.f;
                run = Zf{d.x + cs * run.x, d.y + cs * run.y};
                clen *= cs;
                row[(size_t)s * nsel] = run;
            }
        }
    }
    const float4 comp = block_scan_excl<float4, decltype(scan_op), PIT_RT>(float4{clen, run.x, run.y, 0.f}, scan_op, float4{1.f, 0.f, 0.f, 0.f}, aff);
    const Zf cin{comp.y, comp.z};
    float pw = 1.f;
    for (int sb = s0; sb < s1; sb += 4) {
        Zf v[4], x[4], t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = sb + q;
            const bool ok = s < s1;
            v[q] = ok ? row[(size_t)s * nsel] : Zf{0.f, 0.f};
            x[q] = ok ? xr[(size_t)s * nsel] : Zf{0.f, 0.f};
            t[q] = ok ? th(s) : Zf{1.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int s = sb + q;
            if (s < s1) {
                pw *= s > 0 ? cf(s - 1) : 0.f;
                v[q].x += pw * cin.x; v[q].y += pw * cin.y;
                if (uq && s > 0) {                               // the scanned component along u[s] back in
                    const float4 al = ualpha[(size_t)(s - 1) * nsel + j];
                    const float2 qv = uq[(size_t)s * nsel + j];
                    const float lk2 = (float)(lam[k] > 0 ? lam[k] : 0.0);
                    const Zf ts = s > 0 ? th(s - 1) : Zf{1.f, 0.f};
                    const Zf aa = cmulf(ts, yr[(size_t)(s - 1) * nsel]);
                    const Zf uqv = cmulf(Zf{aa.x * lk2 * al.z, aa.y * lk2 * al.z}, Zf{qv.x, qv.y});
                    v[q].x += uqv.x; v[q].y += uqv.y;
                }
                row[(size_t)s * nsel] = v[q];
                const Zf xx = cmulf(t[q], x[q]);
                xr[(size_t)s * nsel] = Zf{xx.x + damp * v[q].x, xx.y + damp * v[q].y};
            }
        }
    }
}

// In the eigenbasis the linearised segment map is diagonal, J = diag(exp(-mu g T lam_k)), and D[s] = d[s] + J D[s-1] is one
// first-order recurrence with a constant coefficient per (eigen-direction k, mode j): block (k, j) runs it over the S
// segments (chunks per thread, then the carries).  Correction off -> coefficient 0 (D = d: plain relaxation).
template <typename R>
__global__ void __launch_bounds__(256) pit_recur_kernel(Zf *D, const double *lam, int nsel, int S, int64_t T, const R *mu, double beta, const PitCtrl *c)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    __shared__ float4 aff[512];
    const int k = blockIdx.x, j = blockIdx.y;
    const int ncol = S * nsel;
    Zf *row = D + (size_t)k * ncol + j;
    double a = (double)*mu * c->gain * (double)T * lam[k];
    if (a < 0) a = 0;
    // beta > 0: stronger damping of the well-excited directions than the mean gain gives.  The error functions pull several
    // times harder on the output's amplitude than on anything else (cma: 2 <|s|^4>/<|s|^2> = 2.76 against g = 0.62); that
    // direction lives in the strongly excited part of the spectrum, and a map that damps it too LITTLE makes the iteration
    // stall (error factor |J - J'| / (1 - J')), one that damps the rest a little too much only slows it down.  The weakly
    // excited directions (a << 1), which are the ones that need the propagation, are unaffected.
    const float coef = c->corr_on ? (float)exp(-a * (1 + beta * a)) : 0.f;
    const int len = (S + 255) / 256;
    const int s0 = threadIdx.x * len, s1 = s0 + len < S ? s0 + len : S;
    Zf run{0.f, 0.f};
    for (int s = s0; s < s1; s++) {
        const Zf d = row[(size_t)s * nsel];
        run = Zf{d.x + coef * run.x, d.y + coef * run.y};
        row[(size_t)s * nsel] = run;
    }
    // carry into chunk t: the maps x -> clen x + tot[t] of the chunks before it, composed (a scan of affine maps (A, B))
    const float clen = powf(coef, (float)len);
    const float4 comp = block_scan_excl<float4>(float4{clen, run.x, run.y, 0.f},
                                                [](float4 e, float4 l) { return float4{l.x * e.x, l.x * e.y + l.y, l.x * e.z + l.z, 0.f}; },
                                                float4{1.f, 0.f, 0.f, 0.f}, aff);
    const Zf cin{comp.y, comp.z};
    float pw = coef;
    for (int s = s0; s < s1; s++) {
        Zf v = row[(size_t)s * nsel];
        v.x += pw * cin.x; v.y += pw * cin.y;
        row[(size_t)s * nsel] = v;
        pw *= coef;
    }
}

// Gauge fixing.  The error functions are equivariant under their symmetry group (errfn(g y) = g errfn(y)), so the trajectory
// from rotated start taps is the rotated trajectory and a relative rotation g_s between the start taps of segment s and the
// end taps of segment s-1 (yB ~ g_s yA, from the defect kernel) is not an error: theta_s = g_1 ... g_s brings every segment
// into the frame of segment 0, where the boundary defects are formed and corrected (a rotation treated as an additive
// defect would be wrong in second order and keep the iteration from converging below ~theta^2).
// ---- adaptive step (pythran_equalisation.py:12-16, :171-172) across segments: r = 1 / mu and the previous error are boundary states like the taps.
// After the head of the sweep (exact form): every segment starts from the head's r (pass 0 only), segment 0 also from its last error.
template <typename R>
__global__ void __launch_bounds__(256) pit_adapt_init_kernel(const R *mu, const Cx<R> *e_last, int n, R *rS, Cx<R> *eS)
{
    QH_WAVE_FIRST();
    const R r0 = (R)1 / *mu;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < n; s += gridDim.x * 256) { rS[s] = r0; eS[s] = s == 0 ? *e_last : Cx<R>{0, 0}; }
}
// After a pass: what a segment adds to r does not depend on the r it started from (to first order), so the next start values are
// r[0] + the prefix sums of this pass's increments - one pass carries a change of r through ALL later segments; the previous error of
// segment s + 1 is the last error of segment s.  chg[0] = the largest relative change of a start value (part of the stop rule).
// newton != 0 (from the third pass on): a segment's increment d_s is not quite independent of the r it started from - in a blind
// stage the tap-noise part of |e|^2 shrinks with the step, d ln d / d ln r = -kappa with kappa ~ 1-3 - and the plain prefix sum hands
// that on to every later segment with the opposite sign, pass after pass (profiles/r03_adaptive_tier_b.txt).  With kappa_s from the
// last two passes (secant, clamped to [0, 4]) the new start values solve r'[s+1] = r'[s] + d_s (1 - kappa_s (r'[s] - r[s]) / r[s]):
// a first-order recurrence with one coefficient per segment - a scan of affine maps like the one of the taps.
template <typename R>
__global__ void __launch_bounds__(1024) pit_adapt_scan_kernel(R *rS, const R *rE, Cx<R> *eS, const Cx<R> *eE, int n, float *chg, const PitCtrl *c, float relax,
                                                              float *rPrev, float *dPrev, int newton)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    __shared__ double2 buf[16];
    __shared__ float red[16];
    const int pass = c->passes;
    const int len = (n + 1023) / 1024;
    const int s0 = threadIdx.x * len, s1 = s0 + len < n ? s0 + len : n;
    auto seg_map = [&](int s, double &A, double &B) {             // r'[s+1] = A r'[s] + B
        const double rho = (double)rS[s], d = (double)rE[s] - rho;
        double kappa = 0;
        if (newton && pass >= 2) {
            const double rp = (double)rPrev[s], dp = (double)dPrev[s];
            if (fabs(rho - rp) > 1e-6 * rho && fabs(d) > 1e-30) kappa = -((d - dp) / d) / ((rho - rp) / rho);
            kappa = kappa == kappa ? (kappa < 0 ? 0 : (kappa > 4 ? 4 : kappa)) : 0;
        }
        A = 1.0 - kappa * d / rho;
        B = d * (1.0 + kappa);
    };
    double Ac = 1, Bc = 0;
    for (int s = s0; s < s1; s++) { double A, B; seg_map(s, A, B); Bc = A * Bc + B; Ac = A * Ac; }
    auto comp = [](double2 e, double2 l) { return double2{l.x * e.x, l.x * e.y + l.y}; };
    const double2 pre = block_scan_excl<double2, decltype(comp), 1024>(double2{Ac, Bc}, comp, double2{1.0, 0.0}, buf);
    double run = pre.x * (double)rS[0] + pre.y;                   // r'[s0]
    float worst = 0.f;
    for (int s = s0; s < s1; s++) {
        double A, B;
        seg_map(s, A, B);
        const float old = (float)rS[s];
        rPrev[s] = old; dPrev[s] = (float)((double)rE[s] - (double)old);
        if (s > 0) {
            const float nw = pass == 0 ? (float)run : old + relax * ((float)run - old);
            const float rel = fabsf(nw - old) / (fabsf(old) > 1e-30f ? fabsf(old) : 1e-30f);
            worst = (rel > worst || !(rel == rel)) ? (rel == rel ? rel : 3.0e38f) : worst;
            rS[s] = (R)nw;
            eS[s] = eE[s - 1];
        }
        run = A * run + B;
    }
    for (int o = 32; o > 0; o >>= 1) { const float w2 = __shfl_xor(worst, o); worst = worst > w2 ? worst : w2; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = worst;
    __syncthreads();
    if (threadIdx.x == 0) { float m = 0.f; for (int w = 0; w < 16; w++) m = m > red[w] ? m : red[w]; chg[0] = m; }
}
template <typename R> __global__ void pit_adapt_finish_kernel(R *mu, const R *rE, int n) { *mu = (R)1 / rE[n - 1]; }

// The small initialisations of a call's prologue in ONE launch (they were a memset / copy / one-thread kernel each, ~5 us of idle stream apiece:
// 35 us before the first kernel of a cold sweep that does any work): up to three regions to zero, two to copy, the selected modes.
struct PitInit { void *z[3]; unsigned zn[3]; void *cd[2]; const void *cs[2]; unsigned cn[2]; int64_t *modes_dev; int64_t modes[16]; int nsel; };
static __global__ void __launch_bounds__(256) pit_init_kernel(PitInit a)
{
    QH_WAVE_FIRST();
    const unsigned t = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (a.z[r]) for (unsigned i = t; i < a.zn[r] / 4; i += nt) reinterpret_cast<uint32_t *>(a.z[r])[i] = 0u;
#pragma unroll
    for (int r = 0; r < 2; r++)
        if (a.cd[r]) for (unsigned i = t; i < a.cn[r] / 8; i += nt) reinterpret_cast<uint64_t *>(a.cd[r])[i] = reinterpret_cast<const uint64_t *>(a.cs[r])[i];
    if ((int)t < a.nsel) a.modes_dev[t] = a.modes[t];
}
constexpr int PIT_GT_THREADS = 1024;
// model != nullptr (measured coarse model): the gain comes from it, and the components alpha[s] of the defects along the signal direction
// (pit_bound_kernel) are scanned HERE with exp(-mu T K) - in the eigenbasis of K two real first-order recurrences per mode -
// q[s] = alpha[s] + C q[s - 1]; uq[col] = (Re q, Im q) is what pit_recur_eig_kernel adds back along u.  Msum: per-segment sums of the step
// sizes (adaptive step), else mu T.
template <typename R>
__global__ void __launch_bounds__(PIT_GT_THREADS) pit_gauge_kernel(const double *gph, int S, int nsel, PitCtrl *c, double *theta, const double *pw, int nb, int method, const Cx<R> *sy0,
                                                                   int want_corr, const PitModel *model = nullptr, const float4 *ualpha = nullptr, float2 *uq = nullptr,
                                                                   const R *mu = nullptr, int64_t T = 0, const float *Msum = nullptr)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    constexpr int NT = PIT_GT_THREADS;
    __shared__ double redp[NT / 64];
    __shared__ double2 gbuf[NT / 64];
    __shared__ float4 abuf[NT / 64];
    if (gridDim.x > 1 && blockIdx.x == 1 && !(model && uq)) return;
    if (model && uq && (gridDim.x == 1 || blockIdx.x == 1)) {
        const bool corr = c->passes == 0 ? true : c->corr_on != 0;       // (pass 0: decided below; the model being there, it will be on)
        const int len = (S + NT - 1) / NT;
        const int s0 = threadIdx.x * len, s1 = s0 + len < S ? s0 + len : S;
        const float muT = mu ? (float)((double)*mu * (double)T) : 0.f;
        auto op = [](float4 e, float4 l) { return float4{l.x * e.x, l.x * e.y + l.y, l.z * e.z, l.z * e.w + l.w}; };     // two affine maps side by side: (A1, B1, A2, B2)
        for (int j = 0; j < nsel; j++) {
            const PitModel m = model[j];
            const bool on = corr && m.ok;
            auto cf = [&](int s, float &c1, float &c2) {               // coefficients of segment s (takes q[s] to q[s + 1])
                const float t = Msum ? Msum[(size_t)s * nsel + j] : muT;
                c1 = on ? __expf(-fmaxf(m.k1, 0.f) * t) : 0.f; c2 = on ? __expf(-fmaxf(m.k2, 0.f) * t) : 0.f;
            };
            float4 loc{1.f, 0.f, 1.f, 0.f};
            for (int s = s0; s < s1; s++) {
                float b1 = 0.f, b2 = 0.f, c1 = 0.f, c2 = 0.f;
                if (s > 0) { const float4 al = ualpha[(size_t)(s - 1) * nsel + j]; b1 = m.q11 * al.x + m.q21 * al.y; b2 = m.q12 * al.x + m.q22 * al.y; cf(s - 1, c1, c2); }
                loc = op(loc, float4{c1, b1, c2, b2});
            }
            const float4 pre = block_scan_excl<float4, decltype(op), NT>(loc, op, float4{1.f, 0.f, 1.f, 0.f}, abuf);
            float r1 = pre.y, r2 = pre.w;
            for (int s = s0; s < s1; s++) {
                float b1 = 0.f, b2 = 0.f, c1 = 0.f, c2 = 0.f;
                if (s > 0) { const float4 al = ualpha[(size_t)(s - 1) * nsel + j]; b1 = m.q11 * al.x + m.q21 * al.y; b2 = m.q12 * al.x + m.q22 * al.y; cf(s - 1, c1, c2); }
                r1 = c1 * r1 + b1; r2 = c2 * r2 + b2;
                uq[(size_t)s * nsel + j] = float2{m.q11 * r1 + m.q12 * r2, m.q21 * r1 + m.q22 * r2};
            }
        }
        if (gridDim.x > 1) return;                                   // (block 1 of a two-block launch: the frames are block 0's)
    }
    if (c->passes == 0) {                                     // first pass of a sweep: mean output power -> gain of the linearised map (the scan below needs it)
        double ps = 0;
        for (int i = threadIdx.x; i < nb; i += NT) ps += pw[i];
        for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o);
        if ((threadIdx.x & 63) == 0) redp[threadIdx.x >> 6] = ps;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0;
            for (int w = 0; w < NT / 64; w++) tot += redp[w];
            const double Py = tot / (nb > 0 ? nb : 1);
            double g = pit_gain<R>(method, Py, sy0[0]);
            if (model) {                                          // the measured gain (mean over the modes) instead of the formula
                double gs = 0; int ng = 0;
                for (int j = 0; j < nsel; j++) if (model[j].ok) { gs += (double)model[j].g; ng++; }
                if (ng == nsel && gs > 0) g = gs / ng;
            }
            c->out_power = Py; c->gain = g;
            c->corr_on = (want_corr && g > 0 && g == g) ? 1 : 0;
        }
        __syncthreads();
    }
    // theta_s = g_1 ... g_s as a running PRODUCT of unit complex numbers (chunk per thread, scan of the chunk products):
    // no angles, no trigonometry; renormalised on output.  Up to 4 segments per thread: every load is issued before the first use.
    const int len = (S + NT - 1) / NT;
    const int s0 = threadIdx.x * len, s1 = s0 + len < S ? s0 + len : S;
    auto cm = [](double2 x, double2 y) { return double2{x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x}; };
    const double2 *g2 = reinterpret_cast<const double2 *>(gph);
    double2 *th2 = reinterpret_cast<double2 *>(theta);
    if (len <= 4 && nsel <= 2) {
        double2 g[2][4];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int s = s0 + q;
                g[j][q] = (j < nsel && s < s1 && s > 0) ? g2[(size_t)(s - 1) * nsel + j] : double2{1.0, 0.0};
            }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (j >= nsel) break;
            const double2 prod = cm(cm(g[j][0], g[j][1]), cm(g[j][2], g[j][3]));
            double2 run = block_scan_excl<double2, decltype(cm), NT>(prod, cm, double2{1.0, 0.0}, gbuf);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int s = s0 + q;
                run = cm(run, g[j][q]);
                if (s < s1) {
                    const double r = rsqrt(run.x * run.x + run.y * run.y);
                    th2[(size_t)s * nsel + j] = double2{run.x * r, run.y * r};
                }
            }
        }
        return;
    }
    for (int j = 0; j < nsel; j++) {
        double2 prod{1.0, 0.0};
        for (int s = s0; s < s1; s++)
            if (s > 0) prod = cm(prod, g2[(size_t)(s - 1) * nsel + j]);
        double2 run = block_scan_excl<double2, decltype(cm), NT>(prod, cm, double2{1.0, 0.0}, gbuf);
        for (int s = s0; s < s1; s++) {
            if (s > 0) run = cm(run, g2[(size_t)(s - 1) * nsel + j]);
            const double r = rsqrt(run.x * run.x + run.y * run.y);
            th2[(size_t)s * nsel + j] = double2{run.x * r, run.y * r};
        }
    }
}
// Output power of the accumulated corrections: dev2[col] = sum_k lambda_k |D~_k[col]|^2 (D~ in the eigenbasis, after the scan), the
// block's maximum -> devmax[blockIdx.x].  D~[s] is the first-order estimate of how far the start taps of segment s are from the
// sequential trajectory; lambda-weighted it is the power of the output deviation they cause.
template <typename R>
__global__ void __launch_bounds__(256) pit_devest_kernel(const Zf *D, const double *lam, int n, int ncol, const PitCtrl *c, float *devmax, unsigned *ticket, PitDecideArgs<R> da)
{
    QH_WAVE_FIRST();
    if (c->done) return;
    // 64 columns per block, 4 threads per column (each a quarter of the k, loads of consecutive columns coalesce and overlap)
    __shared__ float part[2][4][64], red[64], reds[64], redt[64], redm[64];
    const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float acc = 0.f, tap = 0.f;
    if (col < ncol) {
#pragma unroll 8
        for (int k = kg; k < n; k += 4) {
            const Zf v = D[(size_t)k * ncol + col];
            const float l = (float)lam[k], m2 = v.x * v.x + v.y * v.y;
            acc += (l > 0.f ? l : 0.f) * m2;
            tap += m2;                                            // V is unitary: the squared norm of the tap deviation itself
        }
    }
    part[0][kg][cl] = acc; part[1][kg][cl] = tap;
    __syncthreads();
    if (kg == 0) {
        acc = (part[0][0][cl] + part[0][1][cl]) + (part[0][2][cl] + part[0][3][cl]);
        tap = (part[1][0][cl] + part[1][1][cl]) + (part[1][2][cl] + part[1][3][cl]);
        red[cl] = (acc == acc) ? acc : 3.0e38f;
        reds[cl] = (acc == acc) ? acc : 3.0e38f;
        redt[cl] = (tap == tap) ? tap : 3.0e38f;
        redm[cl] = (tap == tap) ? tap : 3.0e38f;
    }
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
            reds[threadIdx.x] += reds[threadIdx.x + s];
            redt[threadIdx.x] += redt[threadIdx.x + s];
            redm[threadIdx.x] = redm[threadIdx.x] > redm[threadIdx.x + s] ? redm[threadIdx.x] : redm[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {                                       // worst column / sum over the columns of the output power; sum / worst column of the tap norm
        devmax[4 * blockIdx.x] = red[0]; devmax[4 * blockIdx.x + 1] = reds[0]; devmax[4 * blockIdx.x + 2] = redt[0]; devmax[4 * blockIdx.x + 3] = redm[0];
    }
    // The block that finishes last takes the decision of the pass (one launch less per pass): results released device-wide, a ticket
    // drawn; whoever draws the last one acquires and goes on.
    __shared__ int last;
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned t = atomicAdd(ticket, 1u);
        last = t == gridDim.x - 1;
        if (last) { *ticket = 0; __threadfence(); }
    }
    __syncthreads();
    if (!last) return;
    pit_decide_body<R>(da);
}
// last column whose estimated output deviation sum_k lambda_k |D~_k|^2 exceeds thr (way out of a stalled sweep: where does the trouble end?)
static __global__ void __launch_bounds__(256) pit_front_kernel(const Zf *D, const double *lam, int n, int ncol, double thr, int *front, int *bad)
{
    QH_WAVE_FIRST();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    float acc = 0.f;
    for (int k = 0; k < n; k++) { const Zf v = D[(size_t)k * ncol + col]; const float l = (float)lam[k]; acc += (l > 0.f ? l : 0.f) * (v.x * v.x + v.y * v.y); }
    if (!(acc == acc) || acc > 1e30f) { atomicMax(bad, 1); return; }
    if ((double)acc > thr) atomicMax(front, col);
}
// ------------------------------------------------------------------------------------------------ host side
// kernel time of the most recent call (HIP events around the trainer launches; the host synchronises after each anyway)
struct PitTiming { int npass; float pass_ms[QH_PIT_MAXPASS]; float acq_ms; };
inline PitTiming &pit_timing() { static thread_local PitTiming t; return t; }
// Events and a pinned landing area for the flags the host reads: one set per pass / acquisition chunk, because the work of
// pass p + 1 is enqueued BEFORE the host looks at the flag of pass p (every kernel of a pass starts with `if (done) return`,
// so a pass enqueued in vain costs a few empty launches instead of an idle GPU during every host round trip).
constexpr int PIT_NEV = (QH_PIT_MAXPASS > QH_PIT_MAXCHUNK ? QH_PIT_MAXPASS : QH_PIT_MAXCHUNK) + 1;
struct PitEvents { hipEvent_t t0[PIT_NEV], t1[PIT_NEV], flag[PIT_NEV]; int32_t *hflag; float *hview; bool ok; };
inline PitEvents &pit_events()
{
    static thread_local PitEvents e = {{nullptr}, {nullptr}, {nullptr}, nullptr, nullptr, false};
    if (!e.ok) {
        for (int i = 0; i < PIT_NEV; i++) { (void)hipEventCreate(&e.t0[i]); (void)hipEventCreate(&e.t1[i]); (void)hipEventCreateWithFlags(&e.flag[i], hipEventDisableTiming); }
        (void)hipHostMalloc((void **)&e.hflag, PIT_NEV * sizeof(int32_t), hipHostMallocDefault);
        (void)hipHostMalloc((void **)&e.hview, 2 * PIT_NEV * sizeof(float), hipHostMallocCoherent);
        e.ok = e.hflag != nullptr && e.hview != nullptr;
    }
    return e;
}

// Automatic segment grid.  Two things decide it:
//   * convergence: a segment should span a fixed fraction of the time constant 1/(mu g lambda) of the well-excited tap directions,
//     long enough for the linearised coarse correction (mean covariance) to describe what a segment does to its start taps -
//     measured at C3 (mu = 2e-4): segments of >= ~2300 steps contract the deviation 3x per pass all the way down (7 passes to
//     1e-3 from the acquisition taps, at any length from 2340 to 3280), 2176 steps stall at 2x after four passes (8 passes), 2048
//     need 11; from converged taps 2048 steps need 4 passes, 1024 need 5-7.  Target: 0.45 / mu cold, 0.4 / mu warm.
//   * the machine: the segment kernel runs one wave per SIMD (train_seg.h), 1024 SIMDs; a pass costs
//     rounds x steps per segment x cycles per wave and step (~303 with 16 lanes per chain = 4 chains per wave, ~428 with 8 lanes
//     = 8 chains per wave, chosen by seg_lanes from the chain count), so a grid that needs 1.05 rounds costs two.  Cold sweeps
//     leave one CU per shader engine (32 of 256) to the basis build that runs beside the first pass: 896 waves per round.
// The grid is the cheapest of: the target length as it is, exactly one round of 16-lane waves, whole rounds of 8-lane waves -
// never with segments shorter than the target.
inline int pit_auto_segments(int64_t TrSyms, double mu, int nsel, int cold)
{
    double target = (cold ? 0.45 : 0.4) / (mu > 1e-12 ? mu : 1e-12);
    if (target < 256) target = 256;
    if (target > 1048576) target = 1048576;
    const int64_t seg = ((int64_t)target + 63) / 64 * 64;
    int64_t St = TrSyms / seg;
    if (St > PIT_MAXSEG) St = PIT_MAXSEG;
    if (St < 4) return 1;
    if (nsel < 1) nsel = 1;
    const int64_t cap = cold ? 896 : 1024;                     // waves per round
    auto cost = [&](int64_t S) {                               // cycles of one pass
        const int64_t chains = S * nsel;
        const bool l16 = chains <= 4096;                       // seg_lanes
        const int64_t waves = (chains + (l16 ? 3 : 7)) / (l16 ? 4 : 8);
        return (double)((waves + cap - 1) / cap) * (double)(TrSyms / S) * (l16 ? 303.0 : 428.0);
    };
    int64_t best = St;
    double cbest = cost(St);
    auto consider = [&](int64_t S) {
        if (S < 4 || S > PIT_MAXSEG || S > St) return;
        const double c = cost(S);
        if (c < cbest * 0.999 || (c < cbest * 1.001 && S > best)) { best = S; cbest = c; }
    };
    const int64_t S16 = cap * 4 / nsel, S8 = cap * 8 / nsel;
    if (S16 * nsel <= 4096) consider(S16 < St ? S16 : St);
    for (int64_t r = 1; r * S8 <= St && r <= 64; r++) consider(r * S8);
    return (int)best;
}

// Eigenbasis of the input covariance of a capture (depends on E, os, ntaps, TrSyms only - one build serves every stage):
// basis = [ntot eigenvalues (double)][ntot x ntot eigenvectors (float complex, V[i][k])] in device memory.
inline size_t pit_basis_bytes(int ntot) { return (size_t)ntot * sizeof(double) + 2 * (size_t)ntot * ntot * sizeof(Zf); }        // eigenvalues, V, V^T
inline bool pit_basis_ok(int ntot, size_t elem) { return ntot <= PIT_EIGMAX && (size_t)PIT_COVW * ntot * elem <= 64 * 1024; }
// overlap != 0: the build runs on the library's other stream (one workgroup grinding through the Jacobi sweeps next to
// whatever the current stream does next - the acquisition and the first pass do not need the basis); the trainer waits for
// it right before its first correction.
struct PitBasisSync { hipEvent_t in = nullptr, out = nullptr; bool pending = false; };
inline PitBasisSync &pit_basis_sync() { static thread_local PitBasisSync b; return b; }
template <typename R>
int pit_basis(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void *basis, int overlap = 0)
{
    int rc = ensure_init();
    if (rc) return rc;
    hipStream_t st = g_stream;
    PitBasisSync &bs = pit_basis_sync();
    if (overlap) {
        if (!bs.in) { QH_HIP(hipEventCreateWithFlags(&bs.in, hipEventDisableTiming)); QH_HIP(hipEventCreateWithFlags(&bs.out, hipEventDisableTiming)); }
        st = side_stream();
        QH_HIP(hipEventRecord(bs.in, g_stream));                 // the capture is complete on the current stream
        QH_HIP(hipStreamWaitEvent(st, bs.in, 0));
    }
    const int ntot = nmodes * ntaps;
    QH_REQUIRE(pit_basis_ok(ntot, sizeof(Cx<R>)), "pit basis: nmodes*ntaps too large for the eigen-solver");
    QH_REQUIRE(TrSyms >= 1 && (TrSyms - 1) * os + ntaps <= L, "pit basis: field shorter than TrSyms*os + ntaps");
    const size_t msz = (size_t)ntot * ntot;
    void *cb = nullptr;
    if ((rc = scratch(9, (1 + (size_t)PIT_COVB) * msz * sizeof(Z), &cb))) return rc;
    Z *Rc = (Z *)cb, *part = Rc + msz;
    const int ncov = (int)(TrSyms < PIT_COVW * PIT_COVB ? TrSyms : PIT_COVW * PIT_COVB);
    const int ept = (int)((msz + 255) / 256);
    const size_t lds = (size_t)PIT_COVW * ntot * sizeof(Cx<R>);
    if (ept <= 8) hipLaunchKernelGGL((pit_cov_kernel<R, 8>), dim3(PIT_COVB), dim3(256), lds, st, (const Cx<R> *)E, nmodes, L, os, ntaps, TrSyms, ncov, part);
    else if (ept <= 32) hipLaunchKernelGGL((pit_cov_kernel<R, 32>), dim3(PIT_COVB), dim3(256), lds, st, (const Cx<R> *)E, nmodes, L, os, ntaps, TrSyms, ncov, part);
    else hipLaunchKernelGGL((pit_cov_kernel<R, 64>), dim3(PIT_COVB), dim3(256), lds, st, (const Cx<R> *)E, nmodes, L, os, ntaps, TrSyms, ncov, part);
    hipLaunchKernelGGL(pit_cov_reduce_kernel, dim3((unsigned)((msz + 63) / 64)), dim3(256), 0, st, (const Z *)part, (int)msz, PIT_COVB, Rc);
    static std::atomic<bool> attr_set{false};                     // (several host threads may make their first call at the same time: setting it twice is harmless, a torn flag is not)
    if (!attr_set.load(std::memory_order_acquire)) { QH_HIP(hipFuncSetAttribute((const void *)pit_jacobi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); attr_set.store(true, std::memory_order_release); }
    const int nsweep = PIT_EIGSWEEPS;
    // rotations logged by the solver, applied to V row by row afterwards; block form while two copies of A fit the LDS (n <= 96), else
    // the row / column form
    float4 *glog = nullptr;
    const bool logged = true;
    const size_t jlds = (size_t)ntot * (ntot + 1) * sizeof(Zf) + 256;
    {
        void *gl = nullptr;
        const int mm = (ntot + 1) & ~1;
        if ((rc = scratch(11, (size_t)nsweep * (mm - 1) * (mm / 2) * sizeof(float4) + 64, &gl))) return rc;
        glog = (float4 *)gl;
    }
    const size_t blds = 2 * (size_t)ntot * (ntot + 1) * sizeof(Zf) + 256;
    if (logged && blds <= 160 * 1024 - 2048) {          // (the kernel's static LDS: the round's pairs and rotations)
        static std::atomic<bool> battr{false};
        if (!battr.load(std::memory_order_acquire)) { QH_HIP(hipFuncSetAttribute((const void *)pit_jacobi_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048)); battr.store(true, std::memory_order_release); }
        hipLaunchKernelGGL(pit_jacobi_blk_kernel, dim3(1), dim3(1024), blds, st, (const Z *)Rc, ntot, 1.0 / (double)ncov, (double *)basis, nsweep, glog);
    } else
    hipLaunchKernelGGL(pit_jacobi_kernel, dim3(1), dim3(1024), jlds, st, (const Z *)Rc, ntot, 1.0 / (double)ncov, (double *)basis,
                       (Zf *)((char *)basis + (size_t)ntot * sizeof(double)), nsweep, glog);
    if (glog) hipLaunchKernelGGL(pit_jacobi_v_kernel, dim3(ntot), dim3(64), 0, st, (const float4 *)glog, ntot, nsweep, (Zf *)((char *)basis + (size_t)ntot * sizeof(double)));
    QH_HIP(hipGetLastError());
    if (overlap) { QH_HIP(hipEventRecord(bs.out, st)); bs.pending = true; }
    return QH_OK;
}

// ---- the acquisition of a cold sweep ahead of time (qh_pit_prepare_*_dev / qh_pit_opts.prepared) -----------------------------------------------
// Layout of a preparation buffer: [PitCtrl][mu_acq, 64 B][modes_dev, 128 B][acquired taps][start taps][error trace of the acquisition range][Gram table]
struct PitPrepLayout { size_t ctrl, mu_acq, modes, wx, w0, err, gram, total; int64_t err_pitch; };
template <typename R> inline PitPrepLayout pit_prep_layout(int nmodes, int ntaps, int64_t acq_steps)
{
    PitPrepLayout l;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t wset = (size_t)nmodes * nmodes * ntaps * sizeof(Cx<R>);
    l.err_pitch = (acq_steps + LA_B + 63) / 64 * 64;
    l.ctrl = 0; l.mu_acq = up(sizeof(PitCtrl)); l.modes = l.mu_acq + 64; l.wx = up(l.modes + 128); l.w0 = up(l.wx + wset);
    l.err = up(l.w0 + wset); l.gram = up(l.err + (size_t)nmodes * l.err_pitch * sizeof(Cx<R>));
    l.total = up(l.gram + gram_bytes<R>(l.err_pitch));
    return l;
}
// the training call adopts a prepared acquisition: its taps (unless it diverged: then the sweep starts from the caller's taps, as pit_acq_finish_kernel
// leaves them) and what the report says about it
template <typename R> __global__ void __launch_bounds__(256) pit_adopt_kernel(Cx<R> *wx, const Cx<R> *wx_prep, int n, const PitCtrl *p, PitCtrl *c, R *mu_acq, const R *mu_acq_prep)
{
    QH_WAVE_FIRST();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (!p->diverged && i < n) wx[i] = wx_prep[i];
    if (i == 0) {
        c->acq_chunks = p->acq_chunks; c->acq_steps = p->acq_steps; c->acq_done = 1; c->diverged = p->diverged; c->mu_acq = p->mu_acq;
        for (int q = 0; q < QH_PIT_MAXCHUNK; q++) c->acq_err[q] = p->acq_err[q];
        *mu_acq = *mu_acq_prep;
    }
}
template <typename R>
int pit_prepare(const void *E, int nmodes, int64_t L, int64_t TrSyms, int os, const R *mu_dev, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                const void *symbols, int64_t nsy, int method, const qh_pit_opts *opts, void *prep, size_t prep_bytes)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(opts && prep, "pit prepare: options and a preparation buffer are needed");
    QH_REQUIRE(nmodes >= 1 && ntaps >= 1 && os >= 1 && TrSyms >= 1 && nsel >= 1 && nsel <= 16 && nsy >= 1, "pit prepare: bad sizes");
    QH_REQUIRE((TrSyms - 1) * os + ntaps <= L, "pit prepare: field shorter than TrSyms*os + ntaps");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "pit prepare: mode number >= nmodes");
    const qh_pit_opts &o = *opts;
    QH_REQUIRE(o.acquire != 0 && o.adaptive == 0 && o.segments > 1 && o.mu_hint > 0 && o.acq_chunk > 0 && o.head_steps == 0 && !o.exchange,
               "pit prepare: not preparable (a cold fixed-step sweep with segments, mu_hint and acq_chunk given)");
    const int ntot = nmodes * ntaps;
    // ---- the decisions of train_pit_dev that the acquisition depends on (same rules: the two must run the same kernels)
    int S = o.segments;
    const int64_t nblk_all = TrSyms / LA_B;
    if ((int64_t)S * 4 > nblk_all) S = (int)(nblk_all / 4);
    QH_REQUIRE(S >= 2, "pit prepare: not preparable (nothing to parallelise)");
    const int64_t seg_len = nblk_all / S * LA_B;
    const char *force = trainer_force();
    const bool decision = method == QH_M_SBD || method == QH_M_MDDMA || method == QH_M_DD;
    const bool bi_ok = force[0] != 'd' && force[0] != 'l' && bi_supported(method, 0, nmodes, ntaps, os, seg_len, nsy, sizeof(Cx<R>));
    bool seg_ok = force[0] == 0 && seg_supported(method, nmodes, ntaps, os, nsy, sizeof(Cx<R>), nsel);
    {
        const int pf = form(FORM_PIT);                            // qh_set_form("pit_form", "segment" | "block")
        if (pf == 2) seg_ok = false;
        else if (pf != 1 && (int64_t)S * nsel < 512) seg_ok = false;
    }
    const bool la_ok = force[0] != 'd' && method != QH_M_SBD_DATA && la_supported(method, 0, nmodes, ntaps, os, seg_len, nsy);
    const bool use_bi = bi_ok && (decision || !la_ok || force[0] == 'i');
    const bool block_form = use_bi || la_ok;
    QH_REQUIRE(seg_ok && block_form && !decision && method != QH_M_SBD_DATA && la_shape_ok(nmodes, ntaps, os),
               "pit prepare: not preparable (throughput-form passes, block-form acquisition, blind error function)");
    const bool pair_tab = use_bi ? la_shape_ok(nmodes, ntaps, os) : true;
    const double gear = o.gear > 0 ? o.gear : 8.0, bound = o.acq_bound > 0 ? o.acq_bound : 0.08, plateau = o.acq_plateau > 0 ? o.acq_plateau : 0.8;
    int64_t acq_ch = (o.acq_chunk + LA_B - 1) / LA_B * LA_B;
    int64_t amax = o.acq_max > 0 ? o.acq_max : 2 * acq_ch;
    if (amax > TrSyms / 2 && o.acq_max <= 0) amax = TrSyms / 2;
    if (amax > TrSyms) amax = TrSyms;
    const PitPrepLayout lay = pit_prep_layout<R>(nmodes, ntaps, amax);
    QH_REQUIRE(prep_bytes >= lay.total, "pit prepare: preparation buffer too small (qh_pit_prepare_bytes)");
    char *pb = (char *)prep;
    PitCtrl *ctrl = (PitCtrl *)(pb + lay.ctrl);
    R *mu_acq = (R *)(pb + lay.mu_acq);
    int64_t *modes_dev = (int64_t *)(pb + lay.modes);
    Cx<R> *wxp = (Cx<R> *)(pb + lay.wx), *w0 = (Cx<R> *)(pb + lay.w0), *errp = (Cx<R> *)(pb + lay.err);
    const size_t wset = (size_t)nmodes * ntot, wbytes = wset * sizeof(Cx<R>);
    PitSeg sg; sg.S = S; sg.len = seg_len; sg.extra = nblk_all - (seg_len / LA_B) * S; sg.tail = TrSyms - nblk_all * LA_B; sg.begin = 0;
    const int64_t npow = L < 4096 ? L : 4096;
    hipLaunchKernelGGL((pit_setup_kernel<R>), dim3(1), dim3(256), 0, g_stream, (const Cx<R> *)E, nmodes, L, npow, ntot, mu_dev, gear, bound, o.tol > 0 ? o.tol : 1e-3, sg, ctrl, mu_acq);
    PitInit init{};
    init.cd[0] = wxp; init.cs[0] = wx0; init.cn[0] = (unsigned)wbytes;
    init.cd[1] = w0; init.cs[1] = wx0; init.cn[1] = (unsigned)wbytes;
    init.modes_dev = modes_dev; init.nsel = nsel;
    for (int j = 0; j < 16; j++) init.modes[j] = j < nsel ? modes[j] : 0;
    hipLaunchKernelGGL(pit_init_kernel, dim3(32), dim3(256), 0, g_stream, init);
    const int64_t ngram = (amax / LA_B + 1) * LA_B < TrSyms ? (amax / LA_B + 1) * LA_B : TrSyms;
    QH_REQUIRE(ngram <= lay.err_pitch, "pit prepare: acquisition range exceeds the preparation buffer");
    void *G = nullptr;
    rc = pair_tab ? gram_build<R>(E, nmodes, L, os, ntaps, ngram, &G, 1, 0, 0, pb + lay.gram) : QH_ERR_ARG;
    if (rc) return rc;
    const int64_t g_per_step = LA_B;
    int64_t CH = acq_ch;
    if (CH * QH_PIT_MAXCHUNK < amax) CH = (amax + QH_PIT_MAXCHUNK - 1) / QH_PIT_MAXCHUNK;
    CH = (CH + LA_B - 1) / LA_B * LA_B;
    if (CH < 4 * LA_B) CH = 4 * LA_B;
    const int nchunks = (int)(amax / CH);
    LaArgs<R> la;
    la.E = (const Cx<R> *)E; la.symbols = (const Cx<R> *)symbols; la.err = errp; la.G = (const GramPair<R> *)G; la.gpair = 1;
    la.mu = mu_acq; la.mu_out = nullptr; la.mu_cs = 0; la.mu_ms = 0;
    la.L = L; la.Lp = L; la.nsy = nsy; la.sy_pitch = nsy; la.err_pitch = lay.err_pitch;
    la.nmodes = nmodes; la.ntaps = ntaps; la.os = os; la.nsel = nsel; la.method = method;
    la.E_cs = 0; la.err_cs = 0; la.G_cs = 0; la.wx_cs = 0;
    for (int j = 0; j < 16; j++) la.modes[j] = j < nsel ? modes[j] : 0;
    la.prof = nullptr; la.seg = 0; la.seg_extra = 0; la.seg_tail = 0; la.skip = &ctrl->acq_done; la.niter = 1;
    for (int c = 0; c < nchunks; c++) {
        const int64_t step0 = (int64_t)c * CH;
        LaArgs<R> lp = la;
        lp.E = (const Cx<R> *)E + step0 * os; lp.L = L - step0 * os; lp.TrSyms = CH; lp.nch = 1; lp.wx = wxp;
        lp.G = (const GramPair<R> *)G + step0 * g_per_step; lp.err_off = step0;
        if ((rc = use_bi ? launch_bi<R>(lp) : launch_la<R>(lp))) return rc;
        hipLaunchKernelGGL((pit_acq_monitor_kernel<R>), dim3(1), dim3(256), 0, g_stream, (const Cx<R> *)errp, (int64_t)lay.err_pitch, step0, CH,
                           nsel, (const int64_t *)modes_dev, plateau, ctrl, mu_acq, mu_dev, o.acq_anneal < 0 ? -1.0 : (o.acq_anneal > 0 ? (double)o.acq_anneal : 2.0));
    }
    hipLaunchKernelGGL((pit_acq_finish_kernel<R>), dim3((unsigned)((wset + 255) / 256)), dim3(256), 0, g_stream, wxp, (const Cx<R> *)w0, (int)wset, ctrl);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R>
int train_pit_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, R *mu_dev, void *wx, int ntaps,
                  const int64_t *modes, int nsel, const void *symbols, int64_t nsy, int method, void *err, int zero_err,
                  const void *gram, const qh_pit_opts *opts, void *report_dev, int depth = 0)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(nmodes >= 1 && ntaps >= 1 && os >= 1 && Niter >= 0 && TrSyms >= 0 && nsel >= 1 && nsel <= 16 && nsy >= 1, "train_equaliser: bad sizes");
    QH_REQUIRE(TrSyms == 0 || (TrSyms - 1) * os + ntaps <= L, "train_equaliser: field shorter than TrSyms*os + ntaps");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "train_equaliser: mode number >= nmodes");
    qh_pit_opts o;
    memset(&o, 0, sizeof(o));
    o.phase_seed = -1; o.corr_beta = -1;
    if (opts) o = *opts;
    QH_REQUIRE(o.segments >= 0 && o.segments <= PIT_MAXSEG && o.max_passes >= 0 && o.max_passes <= QH_PIT_MAXPASS, "train_equaliser: bad segment / pass count");
    QH_REQUIRE(o.adaptive >= 0 && o.adaptive <= 2, "train_equaliser: adaptive must be 0, 1 or 2");
    // Adaptive step (opts.adaptive = 1): ONE output mode per call (the reference carries one step size from mode to mode: the caller runs
    // the modes in turn), one sweep, single precision, the error functions of train_seg_*_f32_ad.hip.  The head of the sweep runs in
    // the exact form; the segments cover the rest, with r = 1 / mu and the previous error as boundary states.
    // Tier b is TOTAL: a call for which no parallel-in-time solver exists - data-aided training (the symbols are indexed by the step:
    // nothing to certify against a segment grid yet), the adaptive step with another error function / precision / several sweeps /
    // several modes at once / one step size per mode (adaptive = 2) - takes the exact form right away and says so (report:
    // segments = 1, converged = 2); the result is the reference's either way.
    bool adaptive = o.adaptive != 0;
    {
        bool exact_only = method == QH_M_SBD_DATA;
        if (adaptive && (o.adaptive == 2 || sizeof(R) != 4 || nsel != 1 || Niter != 1 || !seg_adaptive_supported(method))) exact_only = true;
        QH_REQUIRE(!(adaptive && o.exchange), "train_equaliser: a capture split over processes is trained with a fixed step");
        if (exact_only) {
            QH_REQUIRE(!o.exchange, "train_equaliser: a capture split over processes needs the parallel-in-time solver (blind / decision-directed method, fixed step)");
            void *cb0 = nullptr;
            if ((rc = scratch(8, sizeof(PitCtrl) + 64, &cb0))) return rc;
            PitCtrl *c0 = report_dev ? (PitCtrl *)report_dev : (PitCtrl *)cb0;
            PitSeg s1; s1.S = 1; s1.len = TrSyms; s1.extra = 0; s1.tail = 0;
            hipLaunchKernelGGL((pit_setup_kernel<R>), dim3(1), dim3(256), 0, g_stream, (const Cx<R> *)E, nmodes, L, (int64_t)(L < 4096 ? L : 4096), nmodes * ntaps, (const R *)mu_dev, 1.0, 1.0,
                               o.tol > 0 ? o.tol : 1e-3, s1, c0, (R *)((char *)cb0 + sizeof(PitCtrl)));
            if ((rc = train_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, o.adaptive, symbols, nsy, method, err, zero_err, adaptive ? nullptr : gram))) return rc;
            const int32_t hdr[3] = {1, 0, 2};                          // segments, passes, converged = 2: the exact form
            QH_HIP(hipMemcpyAsync(&c0->segments, hdr, sizeof(hdr), hipMemcpyHostToDevice, g_stream));
            QH_HIP(hipStreamSynchronize(g_stream));
            return QH_OK;
        }
    }
    if (adaptive) o.acquire = 0;
    const int ntot = nmodes * ntaps;
    QH_REQUIRE(ntot <= 64 * 16, "train_equaliser: more than 1024 taps per output mode are not supported");
    // Adaptive sweeps apply 0.7 of every correction after the first (the full one overshoots: the estimate then GROWS 1.7 x per pass on the
    // blind stage of the reference script's recipe; damped it falls 0.3-0.5 x per pass, profiles/r03_adaptive_tier_b.txt) and what is left
    // of the early corrections sits in the result, so they are held to a third of the tolerance and may take 24 passes.
    const int npass = o.max_passes > 0 ? o.max_passes : (adaptive ? QH_PIT_MAXPASS : 16);
    const double tol = (o.tol > 0 ? o.tol : 1e-3) * (adaptive ? 1.0 / 3.0 : 1.0);
    const double safety = o.dev_safety > 0 ? o.dev_safety : PIT_DEV_SAFETY;
    const double gear = o.gear > 0 ? o.gear : 8.0;
    const double bound = o.acq_bound > 0 ? o.acq_bound : 0.08;
    const double plateau = o.acq_plateau > 0 ? o.acq_plateau : 0.8;
    const int sym = pit_symmetry(method);
    const bool seed_phase = o.phase_seed < 0 ? sym == 4 : o.phase_seed != 0;
    const bool want_corr = o.correction != 0 && pit_basis_ok(ntot, sizeof(Cx<R>));
    double beta = o.corr_beta >= 0 ? o.corr_beta : (sym == 0 ? 1.5 : 0.0);       // (round-3 model only: the measured model needs no extra damping, see below)

    // ---- control block / report
    void *cbuf = nullptr;
    if ((rc = scratch(8, sizeof(PitCtrl) + 64, &cbuf))) return rc;
    PitCtrl *ctrl = report_dev ? (PitCtrl *)report_dev : (PitCtrl *)cbuf;
    R *mu_acq = (R *)((char *)cbuf + sizeof(PitCtrl));            // 8-byte aligned: sizeof(PitCtrl) is a multiple of 8

    // ---- segment grid
    int S = o.segments;
    // adaptive step: the first 16384 steps (while the step is large) in the exact form, then segments of 2048 steps
    const float ad_relax = 1.0f, ad_damp = 0.7f;                  // (measured: profiles/r03_adaptive_tier_b.txt)
    const int ad_newton = 1;
    const int64_t head_want = 16384;
    // Fixed step: opts.head_steps > 0 - the first head_steps steps of every sweep in the exact form, the segments cover the rest (what the
    // automatic way out below uses when the passes stall on a transient at the start of the sweep that no linear model describes)
    int64_t head = adaptive ? ((TrSyms / 4 < head_want ? TrSyms / 4 : head_want) / LA_B * LA_B) : 0;
    if (!adaptive && o.head_steps > 0) {
        QH_REQUIRE(!o.exchange, "train_equaliser: a capture split over processes takes no exact head");
        head = (int64_t)o.head_steps / LA_B * LA_B;
        if (head > TrSyms) head = TrSyms / LA_B * LA_B;
        o.acquire = 0;                                            // the head's end taps seed the segments: nothing to acquire
    }
    // Segment length of an adaptive sweep: 2048 steps; a BLIND stage gets at most ~512 segments - the transient of its iteration grows
    // with the number of segments (2^22 symbols: 2039 segments of 2048 steps meet tol / 3 only at the 24-pass cap or not at all, 509
    // segments of 8192 steps in 12-14 passes; the decision-directed stage needs 5-7 passes either way and is faster with short segments)
    int64_t seg_want = 2048;
    if (adaptive && !(method == QH_M_SBD || method == QH_M_MDDMA || method == QH_M_DD)) {
        const int64_t t = ((TrSyms - head) / 512 + LA_B - 1) / LA_B * LA_B;
        if (t > seg_want) seg_want = t;
    }
    if (S == 0 && adaptive) S = (int)((TrSyms - head) / seg_want < PIT_MAXSEG ? (TrSyms - head) / seg_want : PIT_MAXSEG);
    if (S == 0) {
        R mu_h = (R)o.mu_hint;                                    // the caller's host copy of the step, if it gave one: no read-back, no synchronisation
        if (!(o.mu_hint > 0)) {
            QH_HIP(hipMemcpyAsync(&mu_h, mu_dev, sizeof(R), hipMemcpyDeviceToHost, g_stream));
            QH_HIP(hipStreamSynchronize(g_stream));
        }
        S = pit_auto_segments(TrSyms, (double)mu_h, nsel, o.acquire);
    }
    const int64_t nblk_all = (TrSyms - head) / LA_B;
    if ((int64_t)S * 4 > nblk_all) S = (int)(nblk_all / 4);       // at least 4 blocks per segment
    if (adaptive && S < 16) S = 1;                                // (too short to be worth it: the exact form)
    PitSeg sg;
    sg.S = S > 1 ? S : 1;
    sg.len = sg.S > 0 ? nblk_all / sg.S * LA_B : 0;
    sg.extra = nblk_all - (sg.len / LA_B) * sg.S;
    sg.tail = (TrSyms - head) - nblk_all * LA_B;
    sg.begin = head;
    if (zero_err) QH_HIP(hipMemsetAsync(err, 0, (size_t)nmodes * TrSyms * Niter * sizeof(Cx<R>), g_stream));
    const int64_t npow = L < 4096 ? L : 4096;
    hipLaunchKernelGGL((pit_setup_kernel<R>), dim3(1), dim3(256), 0, g_stream, (const Cx<R> *)E, nmodes, L, npow, ntot, (const R *)mu_dev, gear, bound, tol, sg, ctrl, mu_acq);
    QH_HIP(hipGetLastError());
    if (TrSyms == 0 || Niter == 0) return QH_OK;
    if (sg.S < 2) {                 // nothing to parallelise: the sequential path (report: one segment, one pass, defect 0)
        if ((rc = train_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive ? 1 : 0, symbols, nsy, method, err, 0, gram))) return rc;
        const double zero = 0;
        const int32_t hdr[3] = {1, 1, 2};                          // segments, passes, converged = 2: the exact form
        QH_HIP(hipMemcpyAsync(&ctrl->segments, hdr, sizeof(hdr), hipMemcpyHostToDevice, g_stream));
        QH_HIP(hipMemcpyAsync(&ctrl->defect[0], &zero, sizeof(double), hipMemcpyHostToDevice, g_stream));
        QH_HIP(hipStreamSynchronize(g_stream));
        return QH_OK;
    }

    // ---- which exact form takes the segments (same rules as the sequential path, fixed step)
    const char *force = trainer_force();
    bool bi_ok = force[0] != 'd' && force[0] != 'l' && bi_supported(method, 0, nmodes, ntaps, os, sg.len, nsy, sizeof(Cx<R>));
    const bool decision = method == QH_M_SBD || method == QH_M_MDDMA || method == QH_M_DD;
    void *dd_table = nullptr;
    int dd_npart = -1;
    bool dd_general = false;
    bool seg_ok = force[0] == 0 && seg_supported(method, nmodes, ntaps, os, nsy, sizeof(Cx<R>), nsel);
    if ((bi_ok || seg_ok) && decision) {
        if ((rc = slicer_tables<R>(symbols, nmodes, nsy, modes, nsel, &dd_table, &dd_npart))) return rc;
        const bool sq = dd_npart == 1 || dd_npart == 3 || dd_npart == 7 || dd_npart == 15;
        dd_general = bi_ok && !sq && bi_general_ok(nmodes, ntaps, os, nsy, sizeof(Cx<R>));        // crosses: block form with the alphabet scan
        bi_ok = bi_ok && (sq || dd_general); seg_ok = seg_ok && sq;
    }
    // Form of the passes.  Few chains: the latency forms (look-ahead / block-iterative, one workgroup per chain).  Many
    // chains: the throughput form (train_seg.h: 16 lanes per chain, no Gram table).  qh_set_form("pit_form", "segment" | "block") forces.
    {
        const int pf = form(FORM_PIT);
        if (pf == 2) seg_ok = false;
        else if (pf != 1 && (int64_t)sg.S * nsel < 512) seg_ok = false;
    }
    if (adaptive) {
        // the adaptive solver needs the throughput form of the passes; a tap layout / alphabet it cannot take: the exact form (converged = 2)
        if (!(seg_supported(method, nmodes, ntaps, os, nsy, sizeof(Cx<R>), nsel, 8) && (!decision || dd_npart == 1 || dd_npart == 3 || dd_npart == 7 || dd_npart == 15))) {
            if ((rc = train_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, 1, symbols, nsy, method, err, 0, nullptr))) return rc;
            const int32_t hdr[3] = {1, 0, 2};
            QH_HIP(hipMemcpyAsync(&ctrl->segments, hdr, sizeof(hdr), hipMemcpyHostToDevice, g_stream));
            QH_HIP(hipStreamSynchronize(g_stream));
            return QH_OK;
        }
        seg_ok = true;
    }
    const bool seg_form = seg_ok;
    const bool split = o.exchange != nullptr;
    const int own_first = split ? o.seg_first : 0, own_count = split ? o.seg_count : sg.S;
    const bool xchg_async = split && o.exchange_on_stream != 0;
    if (split) {
        QH_REQUIRE(seg_form, "train_equaliser: a capture split over processes needs the throughput form of the passes (>= 512 chains, supported tap layout)");
        QH_REQUIRE(own_first >= 0 && own_count >= 0 && own_first + own_count <= sg.S, "train_equaliser: owned segments outside the segment grid");
        QH_REQUIRE(Niter == 1, "train_equaliser: a capture split over processes is trained in one sweep");
    }
    const bool la_ok = force[0] != 'd' && method != QH_M_SBD_DATA && la_supported(method, 0, nmodes, ntaps, os, sg.len, nsy);
    const bool partitioned = method == QH_M_RDE || method == QH_M_MRDE;
    (void)partitioned;
    // With the chip full the stage is bound by instruction issue, not by one chain's latency: the look-ahead form (~25
    // instructions per step and mode over its 4 waves) beats the block-iterative one (~125: every block is swept ~8 times),
    // so it takes whatever it can (cma-type AND rde / mrde); block-iterative for the decision-directed functions.
    const bool use_bi = bi_ok && (decision || !la_ok || force[0] == 'i');
    const bool block_form = use_bi || la_ok;
    const bool pair_tab = use_bi ? la_shape_ok(nmodes, ntaps, os) : true;

    // ---- buffers
    const size_t wset = (size_t)nmodes * ntot;
    const size_t wbytes = wset * sizeof(Cx<R>);
    void *wbuf = nullptr;
    const size_t nsj = (size_t)sg.S * nsel;
    const size_t bytes_w = ((2 * (size_t)sg.S + 2) * wbytes + 63) / 64 * 64;
    if ((rc = scratch(2, bytes_w + 10 * nsj * sizeof(double) + (size_t)nsel * sizeof(int64_t) + 320 + nsj * 16 + 4 * ((nsj + 63) / 64) * sizeof(float) + 64, &wbuf))) return rc;
    Cx<R> *X = (Cx<R> *)wbuf, *Y = X + (size_t)sg.S * wset, *w_start = Y + (size_t)sg.S * wset, *w_call = w_start + wset;
    double *z = (double *)((char *)wbuf + bytes_w), *rot = z + 2 * nsj, *dfc = rot + 2 * nsj, *pw = dfc + nsj, *gph = pw + nsj, *theta = gph + 2 * nsj;
    int64_t *modes_dev = (int64_t *)(theta + 2 * nsj);
    double *uw_phi = (double *)(modes_dev + ((nsel + 7) / 8 * 8));
    int *uw_jump = (int *)(uw_phi + (size_t)sg.S * nsel);
    float *devmax = (float *)(uw_jump + (size_t)sg.S * nsel);                    // per-block maxima of the deviation estimate (ndev of them)
    const int ndev = (int)((nsj + 63) / 64);                     // (four floats per block: worst column, sum, sum / worst of the tap norms)
    unsigned *ticket = (unsigned *)(devmax + 4 * (size_t)ndev);  // pit_devest_kernel: which block finishes last
    PitInit init{};                                              // (launched once, right before the sweeps: pit_init_kernel)
    init.z[0] = ticket; init.zn[0] = sizeof(unsigned);
    // adaptive step: r and previous error at the start / end of every segment, the sum of its step sizes, the change of the start values;
    // error rows of the head (the exact form writes rows of its own length)
    R *ad_rS = nullptr, *ad_rE = nullptr;
    Cx<R> *ad_eS = nullptr, *ad_eE = nullptr, *ad_errh = nullptr;
    float *ad_M = nullptr, *ad_chg = nullptr, *ad_rP = nullptr, *ad_dP = nullptr;
    if (!adaptive && head > 0) {
        void *ab = nullptr;
        if ((rc = scratch(13, (size_t)nmodes * head * sizeof(Cx<R>) + 64, &ab))) return rc;
        ad_errh = (Cx<R> *)ab;
    }
    if (adaptive) {
        void *ab = nullptr;
        const size_t nS = ((size_t)sg.S + 15) / 16 * 16;
        if ((rc = scratch(13, nS * (2 * sizeof(R) + 2 * sizeof(Cx<R>) + 3 * sizeof(float)) + 64 + (size_t)nmodes * head * sizeof(Cx<R>), &ab))) return rc;
        ad_eS = (Cx<R> *)ab; ad_eE = ad_eS + nS; ad_errh = ad_eE + nS;
        ad_rS = (R *)(ad_errh + (size_t)nmodes * head); ad_rE = ad_rS + nS;
        ad_M = (float *)(ad_rE + nS); ad_rP = ad_M + nS; ad_dP = ad_rP + nS; ad_chg = ad_dP + nS;
    }
    // (the selected modes go to the device as a kernel argument - pit_init_kernel - : no copy from the caller's memory to wait for)
    init.modes_dev = modes_dev; init.nsel = nsel;
    for (int j = 0; j < 16; j++) init.modes[j] = j < nsel ? modes[j] : 0;
    // What the host needs to know before it can enqueue the rest: the step size (a sweep with mu = 0 takes the exact form) and, for a cold
    // start, the gear-shifted step size pit_setup_kernel chose (it sizes the acquisition chunks).  A caller that hands over its host copy of
    // the step (mu_hint) and the chunk length (acq_chunk: from the report of an earlier capture) spares the call its one synchronisation -
    // ~90 us of idle GPU per stage at C3.
    const bool prepared = o.prepared != nullptr && o.acquire && !adaptive && head == 0 && Niter >= 1;
    const bool need_mu = adaptive || !(o.mu_hint > 0), need_acq = o.acquire && !(o.acq_chunk > 0) && !prepared;
    R mu_acq_h = 0;                                               // the gear-shifted step size pit_setup_kernel chose (sizes the acquisition run)
    if (need_acq) QH_HIP(hipMemcpyAsync(&mu_acq_h, mu_acq, sizeof(R), hipMemcpyDeviceToHost, g_stream));
    R mu_host = need_mu ? (R)1 : (R)o.mu_hint;
    if (need_mu) QH_HIP(hipMemcpyAsync(&mu_host, mu_dev, sizeof(R), hipMemcpyDeviceToHost, g_stream));
    if (need_mu || need_acq) QH_HIP(hipStreamSynchronize(g_stream));
    if (!adaptive && !(mu_host != 0)) {                           // a sweep that moves nothing (or a NaN step): the exact form, as it is
        if ((rc = train_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, 0, symbols, nsy, method, err, 0, gram))) return rc;
        const int32_t hdr[3] = {1, 0, 2};
        QH_HIP(hipMemcpyAsync(&ctrl->segments, hdr, sizeof(hdr), hipMemcpyHostToDevice, g_stream));
        QH_HIP(hipStreamSynchronize(g_stream));
        return QH_OK;
    }

    // ---- Gram table: of the whole sweep when the passes run in a block form (acquisition chunks and segments index into
    // it), of the acquisition range only when they run in the throughput form
    int64_t amax = 0, acq_ch = 0;
    if (o.acquire) {
        // Length of the acquisition run: two chunks of 2 / mu_acq steps.  Measured (profiles/r03_acquisition.txt): the passes that follow
        // do not care whether the run was 2/mu_acq or 8/mu_acq steps long (C3, mu_acq = 9.6e-4: 6-7 passes after 2048 .. 8192 steps, 8-10
        // after 1024; C2, 1.9e-3: 6-7 after 2048, 6-9 after 896 or 4096, 12-16 after 512) - and every one of its steps is sequential.
        if (o.acq_chunk > 0) acq_ch = o.acq_chunk;
        else {
            // 2 / mu_acq, rounded to the nearest power of two: mu_acq follows the measured signal power, and a receiver hands the chunk of
            // its first capture back for the later ones (acq_chunk) - a few per cent of power must not make the result of a capture depend
            // on which capture the receiver saw first
            const double m = (double)mu_acq_h > 1e-12 ? (double)mu_acq_h : 1e-12;
            acq_ch = (int64_t)1 << (int)floor(log2(2.0 / m) + 0.5);
            acq_ch = acq_ch < 256 ? 256 : (acq_ch > 4096 ? 4096 : acq_ch);
        }
        acq_ch = (acq_ch + LA_B - 1) / LA_B * LA_B;
        amax = o.acq_max > 0 ? o.acq_max : 2 * acq_ch;            // two chunks, the second at HALF the gear-shifted step (pit_acq_monitor_kernel, never below 2 mu): seeds with
                                                                  // less misadjustment noise - with the measured model, whose passes contract 5-6 x, that is worth the fifth pass of a
                                                                  // cold cma sweep at C3 (estimates 0.14, 0.025, 0.0044, 0.00074 against 0.20, 0.036, 0.0067, 0.0012, 0.0002 without;
                                                                  // one chunk, or chunks of 1024 steps, cost passes on other recipes: profiles/r04_acquisition.txt)
        if (amax > TrSyms / 2 && o.acq_max <= 0) amax = TrSyms / 2;
        if (amax > TrSyms) amax = TrSyms;
    }
    void *G = const_cast<void *>(gram);
    if (prepared) QH_REQUIRE(seg_form && o.acq_chunk > 0, "train_equaliser: a prepared acquisition belongs to throughput-form passes with acq_chunk given (qh_pit_prepare_*_dev said so)");
    if (block_form && !G && (!seg_form || amax > 0) && !prepared) {
        const int64_t n = seg_form ? (amax / LA_B + 1) * LA_B < TrSyms ? (amax / LA_B + 1) * LA_B : TrSyms : TrSyms;
        rc = pair_tab ? gram_build<R>(E, nmodes, L, os, ntaps, n, &G) : gram_cur_build<R>(E, nmodes, L, os, ntaps, n, &G);
        if (rc) return rc;
    }
    const int64_t g_per_step = pair_tab ? LA_B : GRAM_TRI / 2 / LA_B;      // GramPair elements per step

    // ---- coarse correction: eigenbasis of the input covariance (the caller's, or built here), defect vectors in that basis
    const int ncol = sg.S * nsel;
    const double *lam = nullptr;
    const Zf *Vb = nullptr;
    Zf *Dz[2] = {nullptr, nullptr};
    Zf *Xe = nullptr, *Ye = nullptr, *Yprev = nullptr;
    if (want_corr) {
        void *cb = nullptr;
        if ((rc = scratch(10, pit_basis_bytes(ntot) + 4 * (size_t)ntot * ncol * sizeof(Zf) + (size_t)ntot * nsel * sizeof(Zf) + 256, &cb))) return rc;
        void *basis = o.basis ? o.basis : cb;
        if (!o.basis && (rc = pit_basis<R>(E, nmodes, L, os, ntaps, TrSyms, basis))) return rc;
        lam = (const double *)basis; Vb = (const Zf *)((const char *)basis + (size_t)ntot * sizeof(double));
        Dz[0] = (Zf *)((char *)cb + (pit_basis_bytes(ntot) + 63) / 64 * 64); Dz[1] = Dz[0] + (size_t)ntot * ncol;
        Xe = Dz[1] + (size_t)ntot * ncol; Ye = Xe + (size_t)ntot * ncol; Yprev = Ye + (size_t)ntot * ncol;
    }
    // complex64: the analysis of a pass runs in the eigenbasis (pit_basis_gemm_kernel / pit_bound_kernel / pit_recur_eig_kernel); complex128
    // keeps the probe-based analysis, whose defect vectors are formed in double precision before they are projected
    // (qh_set_form("pit_probe", "1") forces it for complex64 too: tests compare the two)
    const bool eig = want_corr && sizeof(R) == 4 && form(FORM_PIT_PROBE) == 0;
    const bool eig_path = eig && !(o.exchange != nullptr);        // (a capture split over processes keeps everything on one stream)
    // Measured coarse model (pit_model_kernel): gain and the 2 x 2 block of the signal direction from the capture itself.  opts.correction = 2
    // keeps round 3's model (one formula gain per error function, diagonal in the eigenbasis, extra damping beta) for comparisons.
    const bool tables_ok = method == QH_M_CMA || method == QH_M_SGNCMA || method == QH_M_MCMA || method == QH_M_RDE || method == QH_M_MRDE ||
                           (decision && (use_bi || seg_form));
    const bool ssb = eig && o.correction != 2 && tables_ok && method != QH_M_CMA2;
    PitModel *model = nullptr;
    float4 *ualpha = nullptr;
    float2 *uqv = nullptr;
    PitModelArgs<R> ma;
    hipEvent_t model_event = nullptr;                             // recorded behind the model kernels of the current sweep
    if (ssb) {
        void *mb = nullptr;
        const size_t pw_ = 2 * PIT_EIGMAX + 8;
        const size_t b_part = (size_t)PIT_MODB * 4 * nsel * pw_ * sizeof(float), b_u = ((size_t)nsel * ntot * sizeof(Cx<R>) + 63) / 64 * 64;
        const size_t b_ua = ((size_t)ncol * sizeof(float4) + 63) / 64 * 64, b_uq = ((size_t)ncol * sizeof(float2) + 63) / 64 * 64;
        if ((rc = scratch(14, b_part + b_u + b_ua + b_uq + 16 * sizeof(PitModel) + 128, &mb))) return rc;
        ma.part = (float *)mb; ma.u = (Cx<R> *)((char *)mb + b_part);
        ualpha = (float4 *)((char *)mb + b_part + b_u); uqv = (float2 *)((char *)ualpha + b_ua);
        model = (PitModel *)((char *)uqv + b_uq); ma.model = model; ma.ticket = (unsigned *)(model + 16);
        init.z[1] = ma.ticket; init.zn[1] = sizeof(unsigned);
        init.z[2] = ualpha; init.zn[2] = (unsigned)(b_ua + b_uq);
        ma.E = (const Cx<R> *)E; ma.wx = (const Cx<R> *)wx; ma.L = L; ma.TrSyms = TrSyms; ma.nmodes = nmodes; ma.ntaps = ntaps; ma.os = os; ma.nsel = nsel; ma.method = method;
        if (decision) { ma.symbols = (const Cx<R> *)dd_table; ma.nsy = 2 * dd_npart + 1; ma.sy_pitch = 2 * BI_DD_MAXLEV; ma.tables = 1; }
        else { ma.symbols = (const Cx<R> *)symbols; ma.nsy = nsy; ma.sy_pitch = nsy; ma.tables = (method == QH_M_RDE || method == QH_M_MRDE) ? 1 : 0; }
        for (int j = 0; j < 16; j++) ma.modes[j] = j < nsel ? modes[j] : 0;
        if (o.corr_beta < 0) beta = 0.0;
    }
    const size_t glds = ((size_t)PIT_EIGMAX * PIT_EIGMAX + (size_t)PIT_EIGMAX * PIT_GT) * sizeof(Zf);
    static std::atomic<bool> gemm_attr{false};
    if (want_corr && !gemm_attr.load(std::memory_order_acquire)) {
        QH_HIP(hipFuncSetAttribute((const void *)pit_cgemm_kernel<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        QH_HIP(hipFuncSetAttribute((const void *)pit_cgemm_kernel<R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        QH_HIP(hipFuncSetAttribute((const void *)pit_basis_mfma_kernel<R, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        QH_HIP(hipFuncSetAttribute((const void *)pit_basis_mfma_kernel<R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        gemm_attr.store(true, std::memory_order_release);
    }

    LaArgs<R> la;
    la.E = (const Cx<R> *)E; la.symbols = (const Cx<R> *)symbols; la.err = (Cx<R> *)err; la.G = (const GramPair<R> *)G; la.gpair = pair_tab ? 1 : 0;
    la.mu = mu_dev; la.mu_out = nullptr; la.mu_cs = 0; la.mu_ms = 0;
    la.L = L; la.Lp = L; la.nsy = nsy; la.sy_pitch = nsy; la.err_pitch = TrSyms * Niter;
    la.nmodes = nmodes; la.ntaps = ntaps; la.os = os; la.nsel = nsel; la.method = method;
    la.E_cs = 0; la.err_cs = 0; la.G_cs = 0; la.wx_cs = (int64_t)wset;
    for (int j = 0; j < 16; j++) la.modes[j] = j < nsel ? modes[j] : 0;
    la.dd_general = use_bi && !seg_form && dd_general;
    if ((use_bi || seg_form) && decision && !la.dd_general) { la.symbols = (const Cx<R> *)dd_table; la.nsy = 2 * dd_npart + 1; la.sy_pitch = 2 * BI_DD_MAXLEV; }
    la.prof = nullptr; la.seg = 0; la.seg_extra = 0; la.seg_tail = 0; la.skip = nullptr; la.niter = 1;
    TrainArgs<R> ta;
    ta.E = (const Cx<R> *)E; ta.symbols = (const Cx<R> *)symbols; ta.err = (Cx<R> *)err; ta.mu = mu_dev;
    ta.L = L; ta.TrSyms = TrSyms; ta.nsy = nsy; ta.nmodes = nmodes; ta.ntaps = ntaps; ta.Niter = Niter; ta.os = os;
    ta.nsel = nsel; ta.adaptive = 0; ta.method = method;
    for (int j = 0; j < 16; j++) ta.modes[j] = j < nsel ? modes[j] : 0;
    ta.win_start = nullptr; ta.win_len = 0; ta.nwin = 0; ta.win_mu = nullptr; ta.e_off = 0; ta.wx_out = nullptr;
    ta.nseg = 0; ta.seg_begin = 0; ta.seg_len = 0; ta.seg_extra = 0; ta.seg_tail = 0; ta.seg_iter = 0; ta.skip = nullptr;

    PitTiming &tm = pit_timing();
    tm.npass = 0; tm.acq_ms = 0;
    bool acq_timed = false;                                       // the acquisition's event pair (read when the passes are through)
    PitEvents &ev = pit_events();
    QH_REQUIRE(ev.ok, "train_equaliser: no pinned memory for the pass flags");
    // Way out (every tier-b call returns the reference's result): a sweep the passes do not certify - estimate stuck above tol, pass
    // budget used up, no coarse model to certify with - is redone in the EXACT form from the taps the call started with, inside the
    // call (report: converged = 2, `exact_form`).  opts.exact_redo_off != 0 leaves the uncertified result in place (converged = 0): for
    // the tests of the iteration itself.
    const bool redo_ok = o.exact_redo_off == 0;
    bool fell_back = false;
    if (redo_ok && !adaptive) { init.cd[0] = w_call; init.cs[0] = wx; init.cn[0] = (unsigned)wbytes; }
    if (o.acquire && Niter > 0) { init.cd[1] = w_start; init.cs[1] = wx; init.cn[1] = (unsigned)wbytes; }     // (the taps the acquisition starts from)
    hipLaunchKernelGGL(pit_init_kernel, dim3(32), dim3(256), 0, g_stream, init);
    QH_HIP(hipGetLastError());
    for (int it = 0; it < Niter; it++) {
        // ================================================================ acquisition (first sweep of a cold start)
        if (it == 0 && prepared) {
            // the acquisition ran ahead of the call (qh_pit_prepare_*_dev, on another stream beside the previous capture's training): adopt it
            const PitPrepLayout lay = pit_prep_layout<R>(nmodes, ntaps, amax);
            const char *pb = (const char *)o.prepared;
            hipLaunchKernelGGL((pit_adopt_kernel<R>), dim3((unsigned)((wset + 255) / 256)), dim3(256), 0, g_stream, (Cx<R> *)wx, (const Cx<R> *)(pb + lay.wx), (int)wset,
                               (const PitCtrl *)(pb + lay.ctrl), ctrl, mu_acq, (const R *)(pb + lay.mu_acq));
            QH_HIP(hipGetLastError());
        } else if (it == 0 && o.acquire) {
            int64_t CH = acq_ch;
            if (CH * QH_PIT_MAXCHUNK < amax) CH = (amax + QH_PIT_MAXCHUNK - 1) / QH_PIT_MAXCHUNK;
            CH = (CH + LA_B - 1) / LA_B * LA_B;
            if (CH < 4 * LA_B) CH = 4 * LA_B;
            const int nchunks = (int)(amax / CH);
            auto enqueue_chunk = [&](int c) -> int {
                const int64_t step0 = (int64_t)c * CH;
                if (block_form) {
                    LaArgs<R> lp = la;
                    lp.E = (const Cx<R> *)E + step0 * os; lp.L = L - step0 * os; lp.TrSyms = CH; lp.nch = 1; lp.wx = (Cx<R> *)wx; lp.wx_cs = 0;
                    lp.G = (const GramPair<R> *)G + step0 * g_per_step; lp.err_off = step0; lp.mu = mu_acq; lp.skip = &ctrl->acq_done;
                    int r = use_bi ? launch_bi<R>(lp) : launch_la<R>(lp);
                    if (r) return r;
                } else {
                    TrainArgs<R> tp = ta;
                    tp.wx = (Cx<R> *)wx; tp.mu = mu_acq; tp.nseg = 1; tp.seg_begin = step0; tp.seg_len = CH; tp.seg_iter = 0; tp.skip = &ctrl->acq_done;
                    int r = launch_any<R>(tp);
                    if (r) return r;
                }
                hipLaunchKernelGGL((pit_acq_monitor_kernel<R>), dim3(1), dim3(256), 0, g_stream, (const Cx<R> *)err, (int64_t)(TrSyms * Niter), step0, CH,
                                   nsel, (const int64_t *)modes_dev, plateau, ctrl, mu_acq, (const R *)mu_dev, o.acq_anneal < 0 ? -1.0 : (o.acq_anneal > 0 ? (double)o.acq_anneal : 2.0));
                return QH_OK;
            };
            // All chunks are enqueued at once and the host does not wait for any of them: once the plateau is reached (or the run diverged) the
            // monitor sets acq_done on the device and the chunks behind it return at once, and everything downstream reads the device's flags.
            // (Rounds 2-3 waited for every chunk's flag: ~30 us of idle GPU after the last one.)  One event pair around the run times it.
            if (nchunks > 0) { QH_HIP(hipEventRecord(ev.t0[PIT_NEV - 1], g_stream)); acq_timed = true; }
            for (int c = 0; c < nchunks; c++)
                if ((rc = enqueue_chunk(c))) return rc;
            if (nchunks > 0) QH_HIP(hipEventRecord(ev.t1[PIT_NEV - 1], g_stream));
            hipLaunchKernelGGL((pit_acq_finish_kernel<R>), dim3((unsigned)((wset + 255) / 256)), dim3(256), 0, g_stream, (Cx<R> *)wx, (const Cx<R> *)w_start, (int)wset, ctrl);
            QH_HIP(hipGetLastError());
        }
        // ================================================================ adaptive step: the head of the sweep in the exact form
        if (adaptive) {
            QH_HIP(hipMemcpyAsync(w_start, wx, wbytes, hipMemcpyDeviceToDevice, g_stream));           // (kept for the way out below)
            QH_HIP(hipMemcpyAsync(ad_chg + 1, mu_dev, sizeof(R), hipMemcpyDeviceToDevice, g_stream));
            if ((rc = train_dev<R>(E, nmodes, L, head, 1, os, mu_dev, wx, ntaps, modes, 1, 1, symbols, nsy, method, ad_errh, 0, nullptr))) return rc;
            QH_HIP(hipMemcpyAsync((Cx<R> *)err + (size_t)modes[0] * TrSyms, ad_errh + (size_t)modes[0] * head, (size_t)head * sizeof(Cx<R>), hipMemcpyDeviceToDevice, g_stream));
            hipLaunchKernelGGL((pit_adapt_init_kernel<R>), dim3((sg.S + 255) / 256), dim3(256), 0, g_stream, (const R *)mu_dev, (const Cx<R> *)(ad_errh + (size_t)modes[0] * head + head - 1),
                               sg.S, ad_rS, ad_eS);
            QH_HIP(hipGetLastError());
        }
        // ================================================================ fixed step: the head of the sweep in the exact form
        if (!adaptive && head > 0) {
            if ((rc = train_dev<R>(E, nmodes, L, head, 1, os, mu_dev, wx, ntaps, modes, nsel, 0, symbols, nsy, method, ad_errh, 0, nullptr))) return rc;
            for (int j = 0; j < nsel; j++)
                QH_HIP(hipMemcpyAsync((Cx<R> *)err + (size_t)modes[j] * TrSyms * Niter + (size_t)it * TrSyms, ad_errh + (size_t)modes[j] * head, (size_t)head * sizeof(Cx<R>),
                                      hipMemcpyDeviceToDevice, g_stream));
        }
        // ================================================================ pass-0 start taps
        // (the per-sweep fields of the control block are reset by pit_seed_kernel below)
        const double *rot_use = nullptr;
        if (seed_phase) {
            int nwin = PIT_SEEDWIN;                                   // head staged in LDS: shortened until it fits the default 64 KiB
            auto phase_lds = [&](int nw) { return ((size_t)ntot + (size_t)nmodes * os * pit_phase_pitch(ntaps, os, nw)) * sizeof(Cx<R>); };
            while (nwin > 32 && phase_lds(nwin) > 60 * 1024) nwin /= 2;
            QH_REQUIRE(phase_lds(nwin) <= 60 * 1024, "train_equaliser: phase seeding does not fit the LDS for this filter shape (pass phase_seed = 0)");
            hipLaunchKernelGGL((pit_phase_kernel<R>), dim3(sg.S, nsel), dim3(256), phase_lds(nwin), g_stream, (const Cx<R> *)E, nmodes, L, os,
                               (const Cx<R> *)wx, ntaps, sg, (const int64_t *)modes_dev, nwin, z);
            hipLaunchKernelGGL(pit_unwrap_kernel, dim3(nsel), dim3(256), 0, g_stream, (const double *)z, sg.S, nsel, rot, uw_phi, uw_jump);
            rot_use = rot;
        }
        // segment 0 starts from the taps the sweep starts from in the reference (before the acquisition moved them), unrotated
        const Cx<R> *w_exact = o.start == 1 ? nullptr : ((it == 0 && o.acquire) ? (const Cx<R> *)w_start : (const Cx<R> *)wx);
        hipLaunchKernelGGL((pit_seed_kernel<R>), dim3(sg.S), dim3(256), 0, g_stream, (const Cx<R> *)wx, nmodes, ntot, (const int64_t *)modes_dev, nsel, rot_use, X, w_exact, Y, ctrl);
        if (ssb) {
            // the coarse model of this sweep, measured at its seed taps: two small launches on the library's helper stream, beside pass 0 (which
            // does not need it; its analysis waits for it) and beside the basis build of a cold sweep on the other stream.
            ma.sg = sg; ma.rot = rot_use;
            static thread_local hipEvent_t ev_seed = nullptr, ev_model = nullptr;
            if (!ev_seed) { QH_HIP(hipEventCreateWithFlags(&ev_seed, hipEventDisableTiming)); QH_HIP(hipEventCreateWithFlags(&ev_model, hipEventDisableTiming)); }
            hipStream_t ss = helper_stream();
            QH_HIP(hipEventRecord(ev_seed, g_stream));
            QH_HIP(hipStreamWaitEvent(ss, ev_seed, 0));
            hipLaunchKernelGGL((pit_model_kernel<R, 0>), dim3(PIT_MODB), dim3(256), 0, ss, ma);
            hipLaunchKernelGGL((pit_model_kernel<R, 1>), dim3(PIT_MODB), dim3(256), 0, ss, ma);
            QH_HIP(hipEventRecord(ev_model, ss));
            model_event = ev_model;
        }
        QH_HIP(hipGetLastError());
        // ================================================================ relaxation passes
        // HIP events around the trainer launch of a pass (qh_pit_last_timing): an event is ~5.6 us of idle stream, so by default only
        // pass 1 of a sweep is timed (pass 0 of a cold sweep shares the chip with the basis build); QAMPY_HIP_PIT_TIMING = all | none
        const int timing_mode = pit_timing_mode();                // (qh_set_pit_timing; the environment variable QAMPY_HIP_PIT_TIMING = all | none is its initial value)
        auto decide_args = [&](int p, const float *dm, const float2 *ye, float2 *yprev, int ne, int ncol_e, int corr_wanted) {
            PitDecideArgs<R> d;
            d.dfc = dfc; d.pw = pw; d.nb = (int)((sg.S - 1) * nsel); d.Ylast = (const Cx<R> *)(Y + (size_t)(sg.S - 1) * wset); d.n = (int)wset; d.wx = (Cx<R> *)wx; d.c = ctrl;
            d.host_view = &ev.hview[2 * p]; d.nrow = nsel; d.devmax = dm; d.ndev = ndev; d.safety = safety; d.Ye = ye; d.Yprev = yprev; d.ne = ne; d.ncol_e = ncol_e;
            d.theta = theta; d.modes_dev = (const int64_t *)modes_dev; d.ntot_w = ntot; d.S = (int)sg.S; d.sym = sym; d.corr_wanted = corr_wanted;
            d.extra = adaptive ? (const float *)ad_chg : nullptr;
            d.Dfin = (ye && !adaptive) ? (const float2 *)Dz[1] : nullptr; d.Vfin = (const float2 *)Vb; d.lam_fin = lam;
            d.stall_from = adaptive ? 5 : 2;
            return d;
        };
        auto timed = [&](int p) { return timing_mode == 2 || (timing_mode == 1 && p == 1); };
        auto enqueue_pass = [&](int p) -> int {
            ((volatile float *)ev.hview)[2 * p] = -1.f;             // "not decided yet": pit_decide_kernel overwrites it (polled below)
            PitFuse<R> fz;
            fz.X = X; fz.Y = Y; fz.theta = theta; fz.modes_dev = (const int64_t *)modes_dev; fz.nmodes = nmodes; fz.nsel = nsel;
            fz.damp = (adaptive && p > 1) ? ad_damp : 1.f;      // (the first correction - from the seeds to the trajectory - in full)
            if (p > 0 && want_corr) {
                // start taps = theta X + V D~ with D[s+1] = d[s+1] + J D[s] from the analysis that closed pass p - 1 (below); with the
                // correction switched off on the device (corr_on = 0) the scan ran with coefficient 0, D = d: plain relaxation
                if (eig)
                    hipLaunchKernelGGL((pit_basis_mfma_kernel<R, 1>), dim3((ncol + PIT_MC - 1) / PIT_MC), dim3(pit_mfma_threads(ntot)), pit_mfma_lds(ntot), g_stream, Vb + (size_t)ntot * ntot,
                                       (const Cx<R> *)nullptr, (const Zf *)Dz[1], (Zf *)nullptr, ntot, ncol, (const PitCtrl *)ctrl, fz);
                else
                    hipLaunchKernelGGL((pit_cgemm_kernel<R, false>), dim3((ncol + PIT_GT - 1) / PIT_GT), dim3(256), glds, g_stream, Vb + (size_t)ntot * ntot, (const Zf *)Dz[1], (Zf *)nullptr, ntot, ncol, (const PitCtrl *)ctrl, fz);
                QH_HIP(hipGetLastError());
            } else if (p > 0) {
                QH_HIP(hipMemcpyAsync(X + wset, Y, (size_t)(sg.S - 1) * wbytes, hipMemcpyDeviceToDevice, g_stream));   // X[s] = end taps of s-1
                QH_HIP(hipMemcpyAsync(Y, X, (size_t)sg.S * wbytes, hipMemcpyDeviceToDevice, g_stream));
            }                                                       // (pass 0: pit_seed_kernel wrote the start taps into X and Y)
            // The start taps of this pass go into the eigenbasis BESIDE the pass (helper stream): they are final once the back product (or the seed
            // kernel) has run, the product is 8-10 us on a chip the pass leaves half empty, and the analysis only needs it after the pass.
            static thread_local hipEvent_t ev_x = nullptr, ev_xe = nullptr;
            // (measured, C3 tol 1e-4, same box, alternating: 1021-1023 MSym/s with the product aside, 1029-1031 with it in line - the two cross-stream
            // event waits per pass cost what the 8 us product costs.  Off; qh_set_form("pit_xaside", "1") switches it on for measurements.)
            const bool x_aside = eig_path && form(FORM_PIT_XASIDE) != 0;
            if (x_aside) {
                if (!ev_x) { QH_HIP(hipEventCreateWithFlags(&ev_x, hipEventDisableTiming)); QH_HIP(hipEventCreateWithFlags(&ev_xe, hipEventDisableTiming)); }
                hipStream_t hs = helper_stream();
                QH_HIP(hipEventRecord(ev_x, g_stream));
                QH_HIP(hipStreamWaitEvent(hs, ev_x, 0));
                if (o.basis && pit_basis_sync().pending) QH_HIP(hipStreamWaitEvent(hs, pit_basis_sync().out, 0));      // (a basis still being built on the other stream)
                hipLaunchKernelGGL((pit_basis_mfma_kernel<R, 0>), dim3((ncol + PIT_MC - 1) / PIT_MC), dim3(pit_mfma_threads(ntot)), pit_mfma_lds(ntot), hs, Vb, (const Cx<R> *)X, (const Zf *)nullptr, Xe,
                                   ntot, ncol, (const PitCtrl *)ctrl, fz);
                QH_HIP(hipEventRecord(ev_xe, hs));
            }
            if (o.on_pass) {
                // the caller's chip-wide work for other streams, to run BESIDE this pass's trainer: an event recorded here is behind everything of the
                // pass before (its back product), so what is gated on it starts with the trainer - not in the analysis behind it, where it would share
                // the chip with the control path, which is the critical path (hook behind the trainer launch: gaps of 88-90 instead of 70-75 us between
                // the passes, C3 1137-1146 against 1163-1167 MSym/s)
                hipStream_t keep = g_stream;
                o.on_pass(o.on_pass_user, it, p);
                g_stream = keep;
            }
            if (timed(p)) QH_HIP(hipEventRecord(ev.t0[p], g_stream));
            if (seg_form) {
                SegArgs<R> sa;
                sa.E = (const Cx<R> *)E; sa.wx = Y; sa.symbols = la.symbols; sa.err = (Cx<R> *)err; sa.mu = mu_dev;
                sa.L = L; sa.TrSyms = TrSyms; sa.nsy = la.nsy; sa.sy_pitch = la.sy_pitch; sa.err_pitch = TrSyms * Niter; sa.err_off = (int64_t)it * TrSyms;
                sa.nmodes = nmodes; sa.ntaps = ntaps; sa.os = os; sa.nsel = nsel; sa.S = sg.S;
                sa.seg_len = sg.len; sa.seg_extra = sg.extra; sa.seg_tail = sg.tail; sa.seg_begin = sg.begin;
                sa.r_in = ad_rS; sa.r_out = ad_rE; sa.e_in = ad_eS; sa.e_out = ad_eE; sa.mu_sum = (R *)ad_M; sa.adapt_first = sg.begin > 0 ? 1 : 0;
                for (int j = 0; j < 16; j++) sa.modes[j] = j < nsel ? modes[j] : 0;
                sa.skip = &ctrl->done;
                sa.q_first = own_first * nsel; sa.q_count = split ? own_count * nsel : 0;
                if (!split || own_count > 0) { int r = launch_seg<R>(sa, method, adaptive); if (r) return r; }
                if (adaptive) hipLaunchKernelGGL((pit_adapt_scan_kernel<R>), dim3(1), dim3(1024), 0, g_stream, ad_rS, (const R *)ad_rE, ad_eS, (const Cx<R> *)ad_eE, sg.S, ad_chg, (const PitCtrl *)ctrl, ad_relax, ad_rP, ad_dP, ad_newton);
            } else if (block_form) {
                LaArgs<R> ls = la;
                ls.TrSyms = sg.len; ls.nch = sg.S; ls.wx = Y; ls.err_off = (int64_t)it * TrSyms; ls.seg = 1; ls.seg_extra = sg.extra; ls.seg_tail = sg.tail;
                ls.skip = &ctrl->done;
                if (sg.begin > 0) {                                  // an exact head: the segments cover the steps from sg.begin on
                    ls.E = (const Cx<R> *)E + sg.begin * os; ls.L = L - sg.begin * os; ls.err_off += sg.begin;
                    ls.G = (const GramPair<R> *)G + sg.begin * g_per_step;
                }
                { int r = use_bi ? launch_bi<R>(ls) : launch_la<R>(ls); if (r) return r; }
            } else {
                TrainArgs<R> ts = ta;
                ts.wx = Y; ts.nseg = sg.S; ts.seg_begin = sg.begin; ts.seg_len = sg.len; ts.seg_extra = sg.extra; ts.seg_tail = sg.tail; ts.seg_iter = it; ts.skip = &ctrl->done;
                { int r = launch_any<R>(ts); if (r) return r; }
            }
            if (timed(p)) QH_HIP(hipEventRecord(ev.t1[p], g_stream));
            if (split) {
                // one capture over several processes: the end taps of the segments trained elsewhere arrive through the caller's
                // all-reduce (zeros here, the trained taps there); from then on every process works on identical data
                if (own_first > 0) QH_HIP(hipMemsetAsync(Y, 0, (size_t)own_first * wbytes, g_stream));
                if (own_first + own_count < sg.S)
                    QH_HIP(hipMemsetAsync(Y + (size_t)(own_first + own_count) * wset, 0, (size_t)(sg.S - own_first - own_count) * wbytes, g_stream));
                if (!xchg_async) QH_HIP(hipStreamSynchronize(g_stream));
                if (o.exchange(o.exchange_user, Y, (size_t)sg.S * wbytes) != 0) { set_error("train_equaliser: the exchange callback failed"); return QH_ERR_ARG; }
            }
            // ---- analysis of the pass: boundary defects -> gauge -> defect vectors in the eigenbasis -> scan (the next correction,
            // and - weighted with the eigenvalues - the estimate of how far this pass is from the sequential recurrence) -> decision
            const int nbnd = (int)((sg.S - 1) * nsel);
            if (eig) {
                if (o.basis && pit_basis_sync().pending) {           // a basis still being built on the other stream
                    QH_HIP(hipStreamWaitEvent(g_stream, pit_basis_sync().out, 0));
                    pit_basis_sync().pending = false;
                }
                auto forward = [&](const Cx<R> *src, Zf *dst) {      // dst = V^H src
                    hipLaunchKernelGGL((pit_basis_mfma_kernel<R, 0>), dim3((ncol + PIT_MC - 1) / PIT_MC), dim3(pit_mfma_threads(ntot)), pit_mfma_lds(ntot), g_stream, Vb, src, (const Zf *)nullptr, dst, ntot, ncol, (const PitCtrl *)ctrl, fz);
                };
                auto forward2 = [&](const Cx<R> *src, Zf *dst, const Cx<R> *src2, Zf *dst2) {      // both products in ONE launch (a launch is ~4.5 us whatever it does)
                    PitFuse<R> f2 = fz;
                    f2.T2 = src2; f2.Out2 = reinterpret_cast<float2 *>(dst2);
                    hipLaunchKernelGGL((pit_basis_mfma_kernel<R, 0>), dim3(2 * ((ncol + PIT_MC - 1) / PIT_MC)), dim3(pit_mfma_threads(ntot)), pit_mfma_lds(ntot), g_stream, Vb, src, (const Zf *)nullptr, dst, ntot, ncol, (const PitCtrl *)ctrl, f2);
                };
                if (p == 0 && model_event) QH_HIP(hipStreamWaitEvent(g_stream, model_event, 0));       // the sweep's coarse model (other stream)
                // Start taps of this pass into the eigenbasis - in EVERY pass (round 5).  Rounds 3-4 projected them once per sweep and kept x~ up to
                // date in the eigenbasis (x~ <- theta x~ + D~, pit_recur_eig_kernel) while the back product updated X itself (X <- theta X + V D~).
                // The two copies part by the rounding of the products and by what V lacks to be unitary (7e-6 per entry, single-precision Jacobi)
                // times the correction - ~1e-6 per boundary, the same sign from boundary to boundary - and a fixed point of the TRACKED copy
                // leaves exactly that as a true defect at every boundary; the weakly excited directions (coefficient ~1) add them up over the
                // whole sweep: 1.0-1.7e-3 of tap deviation at C3 whatever the tolerance, invisible to the estimate (profiles/r05_tap_floor.txt).
                if (x_aside) { QH_HIP(hipStreamWaitEvent(g_stream, ev_xe, 0)); forward((const Cx<R> *)Y, Ye); }       // (the product of X ran beside the pass: see above)
                else forward2((const Cx<R> *)X, Xe, (const Cx<R> *)Y, Ye);
                hipLaunchKernelGGL(pit_bound_kernel, dim3((nbnd + nsel + 3) / 4), dim3(256), 0, g_stream, (const Zf *)Xe, (const Zf *)Ye, (const Zf *)Yprev, lam, ntot, sg.S, nsel, sym,
                                   (const PitCtrl *)ctrl, dfc, pw, gph, ualpha);
                hipLaunchKernelGGL((pit_gauge_kernel<R>), dim3(ssb ? 2 : 1), dim3(PIT_GT_THREADS), 0, g_stream, (const double *)gph, sg.S, nsel, ctrl, theta, (const double *)pw,
                                   nbnd, method, (const Cx<R> *)symbols + (size_t)modes[0] * nsy, 1, (const PitModel *)model, (const float4 *)ualpha, uqv,
                                   (const R *)mu_dev, sg.len, (const float *)ad_M);
                hipLaunchKernelGGL((pit_recur_eig_kernel<R>), dim3(ntot, nsel), dim3(PIT_RT), 0, g_stream, Xe, (const Zf *)Ye, Dz[1], (const double *)theta, lam, nsel, sg.S, sg.len,
                                   (const R *)mu_dev, beta, (const PitCtrl *)ctrl, (const float *)ad_M, (adaptive && p >= 1) ? ad_damp : 1.f,
                                   (const float4 *)ualpha, (const float2 *)uqv);
                hipLaunchKernelGGL((pit_devest_kernel<R>), dim3(ndev), dim3(256), 0, g_stream, (const Zf *)Dz[1], lam, ntot, ncol, (const PitCtrl *)ctrl, devmax, ticket,
                                   decide_args(p, (const float *)devmax, (const float2 *)Ye, (float2 *)Yprev, ntot, ncol, 1));
            } else {
            const size_t dlds = (2 * (size_t)ntot + (size_t)nmodes * os * pit_phase_pitch(ntaps, os, PIT_PROBE)) * sizeof(Cx<R>);
            QH_REQUIRE(dlds <= 60 * 1024, "train_equaliser: boundary probe does not fit the LDS for this filter shape");
            hipLaunchKernelGGL((pit_defect_kernel<R>), dim3(sg.S, nsel), dim3(PIT_PROBE), dlds, g_stream, (const Cx<R> *)E, nmodes, L, os,
                               ntaps, sg, TrSyms, (const int64_t *)modes_dev, (const Cx<R> *)X, (const Cx<R> *)Y, sym, (const PitCtrl *)ctrl, dfc, pw, gph, (const Cx<R> *)wx);
            hipLaunchKernelGGL((pit_gauge_kernel<R>), dim3(1), dim3(PIT_GT_THREADS), 0, g_stream, (const double *)gph, sg.S, nsel, ctrl, theta, (const double *)pw,
                               nbnd, method, (const Cx<R> *)symbols + (size_t)modes[0] * nsy, want_corr ? 1 : 0);
            if (want_corr) {
                if (o.basis && pit_basis_sync().pending) {           // a basis still being built on the other stream
                    QH_HIP(hipStreamWaitEvent(g_stream, pit_basis_sync().out, 0));
                    pit_basis_sync().pending = false;
                }
                hipLaunchKernelGGL((pit_cgemm_kernel<R, true>), dim3((ncol + PIT_GT - 1) / PIT_GT), dim3(256), glds, g_stream, Vb, (const Zf *)nullptr, Dz[1], ntot, ncol, (const PitCtrl *)ctrl, fz);
                hipLaunchKernelGGL((pit_recur_kernel<R>), dim3(ntot, nsel), dim3(256), 0, g_stream, Dz[1], lam, nsel, sg.S, sg.len, (const R *)mu_dev, beta, (const PitCtrl *)ctrl);
                hipLaunchKernelGGL((pit_devest_kernel<R>), dim3(ndev), dim3(256), 0, g_stream, (const Zf *)Dz[1], lam, ntot, ncol, (const PitCtrl *)ctrl, devmax, ticket,
                                   decide_args(p, (const float *)devmax, (const float2 *)nullptr, (float2 *)nullptr, 0, 0, 1));
            } else
                hipLaunchKernelGGL((pit_decide_kernel<R>), dim3(1), dim3(256), 0, g_stream, decide_args(p, (const float *)nullptr, (const float2 *)nullptr, (float2 *)nullptr, 0, 0, 0));
            }
            QH_HIP(hipGetLastError());
            return QH_OK;
        };
        // Pass p + 1 is enqueued BEFORE the host has seen the flag of pass p - unless pass p is expected to be the last one, since
        // a pass enqueued in vain costs a dozen empty launches (~55 us), about as much as the idle round trip it would save:
        // expected defect of pass p = PIT_CONTRACT x the defect of pass p - 1 (the first pass never converges from seeds).
        if ((rc = enqueue_pass(0))) return rc;
        if (it == 0 && o.on_pass0) {                              // the caller's work for the other streams: enqueued while pass 0 keeps the device busy
            hipStream_t keep = g_stream;
            o.on_pass0(o.on_pass0_user);
            g_stream = keep;
        }
        bool ahead = false;                                       // pass p + 1 already in the stream
        bool certified = false;                                   // the sweep's last decision: stopped AND below the tolerance
        for (int p = 0; p < npass; p++) {
            // (expected criterion of pass p: the last one times the contraction the last two passes showed; the first guess is PIT_CONTRACT)
            double contr = PIT_CONTRACT;
            if (p >= 2 && ev.hview[2 * (p - 2) + 1] > 0) {
                contr = (double)ev.hview[2 * (p - 1) + 1] / (double)ev.hview[2 * (p - 2) + 1];
                contr = contr < 0.1 ? 0.1 : (contr > 0.9 ? 0.9 : contr);
            }
            const bool expect_last = p > 0 && contr * (double)ev.hview[2 * (p - 1) + 1] < tol;
            ahead = false;
            if (p + 1 < npass && !expect_last && (!split || xchg_async)) { if ((rc = enqueue_pass(p + 1))) return rc; ahead = true; }   // (a host-side exchange cannot be enqueued ahead)
            // the decision of pass p: polled in pinned memory (an event here would idle the stream for ~5.6 us per pass); the stream
            // going idle without a decision means a launch failed or the sweep was already done
            {
                volatile float *hv = (volatile float *)ev.hview;
                unsigned spins = 0;
                auto t_last = std::chrono::steady_clock::now();
                while (hv[2 * p] < 0.f) {
                    if ((++spins & 255u) != 0) continue;
                    const auto now = std::chrono::steady_clock::now();
                    if (now - t_last < std::chrono::milliseconds(20)) continue;     // (a stream query puts a marker into the queue: rarely)
                    t_last = now;
                    const hipError_t q = hipStreamQuery(g_stream);
                    if (q == hipSuccess) { if (hv[2 * p] < 0.f) hv[2 * p] = 2.f; break; }       // (nothing decided: treated as not certified)
                    if (q != hipErrorNotReady) QH_HIP(q);
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            if (timed(p) && tm.npass < QH_PIT_MAXPASS) {
                float ms = 0;
                QH_HIP(hipEventSynchronize(ev.t1[p]));
                QH_HIP(hipEventElapsedTime(&ms, ev.t0[p], ev.t1[p]));
                tm.pass_ms[tm.npass++] = ms;
            }
            if (ev.hview[2 * p] != 0.f) { certified = ev.hview[2 * p] == 1.f; break; }
            if (p + 1 < npass && !ahead && (rc = enqueue_pass(p + 1))) return rc;
        }
        if (acq_timed) {                                          // (complete long before the passes were)
            float ms = 0;
            QH_HIP(hipEventSynchronize(ev.t1[PIT_NEV - 1]));
            QH_HIP(hipEventElapsedTime(&ms, ev.t0[PIT_NEV - 1], ev.t1[PIT_NEV - 1]));
            tm.acq_ms = ms; acq_timed = false;
        }
        if (!adaptive && !certified && redo_ok && !split && depth < 3 && o.head_auto_off == 0) {
            // Stalled on the START of the sweep?  A stage that begins far from its own fixed point - a decision-directed stage whose start
            // taps were locked to the carrier phase at the END of the capture - goes through a non-linear pull-in that the passes can only
            // follow at the speed of plain relaxation (measured, 64-QAM mcma -> sbd: the bulk of the sweep converged after two passes, a
            // front of ~0.1 deviation moved 1.5 segments per pass).  If the estimate of the last pass sits in the first quarter of the
            // segments only, the call is repeated with that stretch (+ a margin) as an EXACT head: sequential there, parallel after it.
            double hp = 0;
            int hfront[2] = {-1, 0};
            bool finite = false;
            if (eig) {
                int *fr = (int *)(ticket + 4);                               // (two ints behind the ticket: scratch slot 2 has the room)
                QH_HIP(hipMemcpyAsync(&hp, &ctrl->out_power, sizeof(double), hipMemcpyDeviceToHost, g_stream));
                QH_HIP(hipStreamSynchronize(g_stream));
                QH_HIP(hipMemcpyAsync(fr, hfront, sizeof(hfront), hipMemcpyHostToDevice, g_stream));
                hipLaunchKernelGGL(pit_front_kernel, dim3((ncol + 255) / 256), dim3(256), 0, g_stream, (const Zf *)Dz[1], lam, ntot, ncol, tol * tol * (hp > 0 ? hp : 1.0), fr, fr + 1);
                QH_HIP(hipMemcpyAsync(hfront, fr, sizeof(hfront), hipMemcpyDeviceToHost, g_stream));
                QH_HIP(hipStreamSynchronize(g_stream));
                finite = hp > 0 && hfront[1] == 0;
            }
            const int front = hfront[0];
            const int64_t front_seg = front < 0 ? -1 : front / nsel + 1;      // one past the last segment above the tolerance
            if (eig && finite && front >= 0 && front_seg <= sg.S / 4) {
                int64_t hs = front_seg + 8 + (front_seg + 1) / 2;                 // margin: the front keeps moving while the head is redone
                if (hs > sg.S - 4) hs = sg.S - 4;
                const int64_t new_head = sg.start(hs);
                if (new_head > head && new_head < TrSyms && new_head <= 0x7fffffff) {
                    QH_HIP(hipMemcpyAsync(wx, w_call, wbytes, hipMemcpyDeviceToDevice, g_stream));
                    qh_pit_opts o2 = opts ? *opts : o;
                    if (!opts) { memset(&o2, 0, sizeof(o2)); o2.phase_seed = -1; o2.corr_beta = -1; }
                    o2.head_steps = (int32_t)new_head; o2.acquire = 0; o2.segments = 0;
                    return train_pit_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, symbols, nsy, method, err, 0, gram, &o2, report_dev, depth + 1);
                }
            }
        }
        if (!adaptive && !certified && redo_ok) {
            // not certified: the whole call again in the exact form, from the taps it started with (all sweeps: a later sweep starts from
            // the result of this one).  Identical on every process of a split capture (they all took the same decision).
            QH_HIP(hipMemcpyAsync(wx, w_call, wbytes, hipMemcpyDeviceToDevice, g_stream));
            if ((rc = train_dev<R>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, 0, symbols, nsy, method, err, 0, gram))) return rc;
            const int32_t two = 2;
            QH_HIP(hipMemcpyAsync(&ctrl->converged, &two, sizeof(two), hipMemcpyHostToDevice, g_stream));
            QH_HIP(hipStreamSynchronize(g_stream));               // (`two` lives on this stack frame)
            fell_back = true;
            break;
        }
        if (adaptive) {
            hipLaunchKernelGGL((pit_adapt_finish_kernel<R>), dim3(1), dim3(1), 0, g_stream, mu_dev, (const R *)ad_rE, sg.S);
            // A sweep the passes cannot agree on - the later modes of a BLIND stage, which start from centre-spike taps with the tiny step
            // the first mode left behind: no linear model describes that - ends not converged after a few passes; the call then does what
            // the reference does: the exact form from the saved taps and step size (report: converged = 2).
            int32_t conv = 0;
            QH_HIP(hipMemcpyAsync(&conv, &ctrl->converged, sizeof(conv), hipMemcpyDeviceToHost, g_stream));
            QH_HIP(hipStreamSynchronize(g_stream));
            if (!conv && redo_ok) {
                fell_back = true;
                QH_HIP(hipMemcpyAsync(wx, w_start, wbytes, hipMemcpyDeviceToDevice, g_stream));
                QH_HIP(hipMemcpyAsync(mu_dev, ad_chg + 1, sizeof(R), hipMemcpyDeviceToDevice, g_stream));
                if ((rc = train_dev<R>(E, nmodes, L, TrSyms, 1, os, mu_dev, wx, ntaps, modes, 1, 1, symbols, nsy, method, err, 0, nullptr))) return rc;
                const int32_t two = 2;
                QH_HIP(hipMemcpyAsync(&ctrl->converged, &two, sizeof(two), hipMemcpyHostToDevice, g_stream));
                QH_HIP(hipStreamSynchronize(g_stream));
            }
        }
        if (sym == 0 && !fell_back)
            hipLaunchKernelGGL((pit_rotate_err_kernel<R>), dim3(sg.S, nsel), dim3(256), 0, g_stream, (Cx<R> *)err, (int64_t)(TrSyms * Niter), (int64_t)it * TrSyms, sg,
                               (const int64_t *)modes_dev, nsel, (const double *)theta, (const PitCtrl *)ctrl, -1);
        QH_HIP(hipGetLastError());
    }
    return QH_OK;
}

// ---- the drop-in host-array trainer through tier b (qh_set_default_tier(1, tol)): what qh_train_equaliser_c64 / _c128 run when the
// process-wide default tier is b.  Same arguments, same results as train_host up to the stated tolerance: the arrays are staged exactly as there, the
// sweep is solved in parallel in time (cold start - gear-shifted acquisition first - when every trained mode starts from a centre-spike tap set, as the
// reference's wrappers initialise them, equalisation.py:243-249), the reference's adaptive step is solved mode after mode (one shared step size), and
// a call no parallel-in-time solver exists for, or one the passes do not certify, runs in the exact form inside train_pit_dev - the report of the
// (last) solve stays readable through qh_last_pit_report.
inline qh_pit_report &last_host_report() { static thread_local qh_pit_report r; return r; }
template <typename R>
int train_host_tier_b(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, R *mu, void *wx, int ntaps,
                      const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy, int method, void *err, double tol)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(nmodes >= 1 && L >= 1 && ntaps >= 1 && nsy >= 1 && TrSyms >= 0 && Niter >= 0 && nsel >= 1 && nsel <= 16, "train_equaliser: bad sizes");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "train_equaliser: mode number >= nmodes");
    const size_t cs = sizeof(Cx<R>);
    // cold start? every trained mode's tap set is a centre spike (one tap equal to 1, the rest 0)
    bool cold = true;
    {
        const Cx<R> *w = (const Cx<R> *)wx;
        const size_t ntot = (size_t)nmodes * ntaps;
        for (int j = 0; j < nsel && cold; j++) {
            int nz = 0;
            for (size_t f = 0; f < ntot; f++) {
                const Cx<R> v = w[(size_t)modes[j] * ntot + f];
                if (v.re != 0 || v.im != 0) { nz++; if (!(v.re == 1 && v.im == 0)) nz = 2; }
                if (nz > 1) break;
            }
            cold = nz == 1;
        }
    }
    DevBuf dE, dw, ds, de, dmu, drep;
    if ((rc = dE.from_host(E, (size_t)nmodes * L * cs))) return rc;
    if ((rc = dw.from_host(wx, (size_t)nmodes * nmodes * ntaps * cs))) return rc;
    if ((rc = ds.from_host(symbols, (size_t)nmodes * nsy * cs))) return rc;
    if ((rc = dmu.from_host(mu, sizeof(R)))) return rc;
    if ((rc = de.alloc((size_t)nmodes * TrSyms * Niter * cs))) return rc;
    if ((rc = drep.alloc(sizeof(qh_pit_report)))) return rc;
    QH_HIP(hipMemsetAsync(drep.p, 0, sizeof(qh_pit_report), g_stream));
    qh_pit_opts o;
    memset(&o, 0, sizeof(o));
    o.phase_seed = -1; o.corr_beta = -1; o.correction = -1;
    o.tol = tol;
    o.adaptive = adaptive;
    if (adaptive == 1) {
        // the reference carries ONE step size from mode to mode (pythran_equalisation.py:163-172): the modes in turn, each from the step the one before left
        for (int j = 0; j < nsel; j++)
            if ((rc = train_pit_dev<R>(dE.p, nmodes, L, TrSyms, Niter, os, (R *)dmu.p, dw.p, ntaps, modes + j, 1, ds.p, nsy, method, de.p, j == 0 ? 1 : 0, nullptr, &o, drep.p))) return rc;
    } else {
        o.acquire = cold ? 1 : 0;
        if (!adaptive && *mu > 0) o.mu_hint = (double)*mu;
        if (!adaptive && *mu > 0) o.segments = pit_auto_segments(TrSyms, (double)*mu, nsel, o.acquire);
        if ((rc = train_pit_dev<R>(dE.p, nmodes, L, TrSyms, Niter, os, (R *)dmu.p, dw.p, ntaps, modes, nsel, ds.p, nsy, method, de.p, 1, nullptr, &o, drep.p))) return rc;
    }
    if ((rc = dw.to_host(wx, dw.n))) return rc;
    if ((rc = de.to_host(err, de.n))) return rc;
    if ((rc = dmu.to_host(mu, sizeof(R)))) return rc;
    if ((rc = drep.to_host(&last_host_report(), sizeof(qh_pit_report)))) return rc;
    QH_HIP(hipStreamSynchronize(g_stream));
    return QH_OK;
}

}  // namespace qh
