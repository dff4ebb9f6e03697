// On-device synthesis of impaired dual-polarisation captures, so that measurement runs - many channels per
// GPU, 8 GPUs - are not fed by minutes of host numpy.  What the reference does with its signal classes and
// core/impairments.py, in one fused time-domain kernel that reads NOTHING from HBM:
//   symbols      uniformly random Gray-labelled M-QAM indices, counter-based Philox4x32-10 (seed, mode, symbol index)
//   shaping      root-raised-cosine pulse at `os` samples/symbol (Kaiser-windowed FIR spanning +-SYNTH_HSPAN symbols; the
//                host generator shapes in the frequency domain, core/resample.py:73-126), unit mean power
//   phase noise  Wiener process per transmitted mode, variance 2 pi linewidth / fs per sample (core/impairments.py:155-158):
//                per-tile totals first (phase_tile_kernel + scan), then the in-tile prefix
//   PMD          R(-theta) diag(delay +tau/2, -tau/2) R(theta) (core/impairments.py:94-104) as a real 2x2 matrix FIR
//                (windowed-sinc fractional delays), 2 modes only
//   AWGN         sigma = 10^(-snr/20) sqrt(os) on the unit-power signal, split over I and Q (core/impairments.py:205, :230-233)
// Everything is circular in time like the host generator.  One workgroup = SYNTH_T output samples of all modes.
#include "common.h"
#include <math.h>
#include <vector>

namespace qh {

constexpr int SYNTH_T = 1024;          // output samples per workgroup
constexpr int SYNTH_HSPAN = 48;        // pulse half-length in symbols
constexpr int SYNTH_NP = 33;           // taps of the PMD matrix FIR (fractional delays)
constexpr int SYNTH_MAXOS = 4;

struct Philox { unsigned x, y, z, w; };
__device__ __forceinline__ Philox philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox{c0, c1, c2, c3};
}
// two independent standard normals from two 32-bit words (Box-Muller)
__device__ __forceinline__ void gauss2(unsigned a, unsigned b, float &g0, float &g1)
{
    const float u = ((float)a + 1.0f) * 2.3283064365386963e-10f;           // (0, 1]
    const float v = (float)b * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.0f * __logf(u));
    float s, c;
    __sincosf(6.283185307179586f * v, &s, &c);
    g0 = r * c; g1 = r * s;
}

struct SynthArgs {
    Cx<float> *E;              // (nmodes, L) out
    Cx<float> *symbols;        // (nmodes, nsym) out
    int32_t *idx_tx;           // (nmodes, nsym) out
    const Cx<float> *alphabet; // (M,)
    const float *hrrc;         // [2*HSPAN*os + 1] shaping FIR, scaled to unit output power
    const float *pmd;          // [3][NP]: m00, m01 (= m10), m11
    const float *tile_phase;   // [nmodes][ntiles] exclusive per-tile phase offsets (or nullptr)
    int64_t L, nsym;
    int nmodes, M, os, do_pmd;
    unsigned seed_lo, seed_hi;
    float sigma_noise, sigma_phase;
};

__device__ __forceinline__ float phase_increment(const SynthArgs &a, int mode, int64_t n)
{
    const Philox p = philox4x32_10((unsigned)n, (unsigned)(n >> 32), (unsigned)mode, 1u, a.seed_lo, a.seed_hi);
    float g0, g1;
    gauss2(p.x, p.y, g0, g1);
    return a.sigma_phase * g0;
}

// per-tile sums of the phase increments: totals[mode][tile]
__global__ void __launch_bounds__(256) phase_tile_kernel(SynthArgs a, float *totals, int ntiles)
{
    const int tile = blockIdx.x, mode = blockIdx.y;
    float s = 0;
    for (int i = threadIdx.x; i < SYNTH_T; i += 256) {
        const int64_t n = (int64_t)tile * SYNTH_T + i;
        if (n < a.L) s += phase_increment(a, mode, n);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) totals[(size_t)mode * ntiles + tile] = part[0] + part[1] + part[2] + part[3];
}
// exclusive scan of the tile totals, one workgroup per mode (double accumulation: millions of samples)
__global__ void __launch_bounds__(64) phase_scan_kernel(float *totals, int ntiles)
{
    if (threadIdx.x != 0) return;
    float *t = totals + (size_t)blockIdx.x * ntiles;
    double acc = 0;
    for (int i = 0; i < ntiles; i++) { const double v = t[i]; t[i] = (float)acc; acc += v; }
}

__global__ void __launch_bounds__(256) synth_kernel(SynthArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int os = a.os, HP = (SYNTH_NP - 1) / 2, NH = 2 * SYNTH_HSPAN * os + 1, HH = SYNTH_HSPAN * os;
    const int W = SYNTH_T + 2 * HP;                       // shaped samples needed per mode (tile + PMD halo)
    const int64_t n0 = (int64_t)blockIdx.x * SYNTH_T;
    const int64_t k_first = (n0 - HP - HH) >= 0 ? (n0 - HP - HH) / os : -((HP + HH - n0 + os - 1) / os);     // floor
    const int nk = (SYNTH_T + 2 * HP + 2 * HH) / os + 2;  // symbols needed per mode
    Cx<float> *sym = reinterpret_cast<Cx<float> *>(smem);                      // [nmodes][nk]
    Cx<float> *xs = sym + (size_t)a.nmodes * nk;                               // [nmodes][W] shaped, phase-noisy samples
    float *ph = reinterpret_cast<float *>(xs + (size_t)a.nmodes * W);          // [nmodes][W] phase of every sample
    float *hl = ph + (size_t)a.nmodes * W;                                     // [NH] shaping taps
    for (int i = threadIdx.x; i < NH; i += 256) hl[i] = a.hrrc[i];
    // ---- symbols (circular in the symbol index)
    for (int e = threadIdx.x; e < a.nmodes * nk; e += 256) {
        const int m = e / nk, j = e - m * nk;
        int64_t k = (k_first + j) % a.nsym;
        if (k < 0) k += a.nsym;
        const Philox p = philox4x32_10((unsigned)k, (unsigned)(k >> 32), (unsigned)m, 0u, a.seed_lo, a.seed_hi);
        const int idx = (int)(p.x % (unsigned)a.M);
        const Cx<float> s = a.alphabet[idx];
        sym[e] = s;
        const int64_t kk = k_first + j;
        if (kk * os >= n0 && kk * os < n0 + SYNTH_T && kk < a.nsym && kk >= 0) {   // this tile owns the symbol: publish it
            a.symbols[(size_t)m * a.nsym + kk] = s;
            a.idx_tx[(size_t)m * a.nsym + kk] = idx;
        }
    }
    // ---- phase of tile + halo: cumulative sums like np.cumsum, anchored at the tile's exclusive offset
    if (a.sigma_phase > 0.f) {
        for (int e = threadIdx.x; e < a.nmodes * W; e += 256) {
            const int m = e / W, i = e - m * W;
            int64_t n = (n0 - HP + i) % a.L;
            if (n < 0) n += a.L;
            ph[e] = phase_increment(a, m, n);
        }
        __syncthreads();
        if (threadIdx.x < a.nmodes) {                     // serial prefix per mode: 1056 adds, once per tile
            const int m = threadIdx.x;
            float *p = ph + (size_t)m * W;
            // p[i] holds the increment g[n] of sample n = n0 - HP + i; turn it into phi[n] = sum_{j <= n} g[j] (np.cumsum).
            // Samples of the PMD halo that wrap around the ends of the capture simply continue the local sum (the host
            // generator has a phase jump there; +-16 samples at the edges, irrelevant)
            const float base = a.tile_phase[(size_t)m * gridDim.x + blockIdx.x];     // sum_{j < n0} g[j]
            float acc = base;
            for (int i = HP; i < W; i++) { acc += p[i]; p[i] = acc; }
            float gnext = p[HP - 1], cur = base;                                      // phi[n0 - 1] = base
            p[HP - 1] = base;
            for (int i = HP - 2; i >= 0; i--) {                                       // phi[n] = phi[n + 1] - g[n + 1]
                const float gi = p[i];
                cur -= gnext;
                p[i] = cur;
                gnext = gi;
            }
        }
    }
    __syncthreads();
    // ---- pulse shaping + phase rotation into xs
    for (int e = threadIdx.x; e < a.nmodes * W; e += 256) {
        const int m = e / W, i = e - m * W;
        const int64_t n = n0 - HP + i;                                          // sample index (may wrap; symbols already circular)
        // x[n] = sum_k s[k] h[n - k os + HH], |n - k os| <= HH
        const int64_t kmin = (n - HH) >= 0 ? (n - HH + os - 1) / os : -((HH - n) / os);      // ceil((n - HH)/os)
        float xr = 0.f, xi = 0.f;
        int tap = (int)(n - kmin * os) + HH;                                    // index into hl for k = kmin
        const Cx<float> *sp = sym + (size_t)m * nk + (kmin - k_first);
        for (; tap >= 0; tap -= os, sp++) {
            const float h = hl[tap];
            xr = fmaf(h, sp->re, xr); xi = fmaf(h, sp->im, xi);
        }
        if (a.sigma_phase > 0.f) {
            float s, c;
            __sincosf(ph[e], &s, &c);
            const float r = xr * c - xi * s; xi = xr * s + xi * c; xr = r;
        }
        xs[e] = Cx<float>{xr, xi};
    }
    __syncthreads();
    // ---- PMD matrix FIR + noise + store
    for (int e = threadIdx.x; e < a.nmodes * SYNTH_T; e += 256) {
        const int m = e / SYNTH_T, i = e - m * SYNTH_T;
        const int64_t n = n0 + i;
        if (n >= a.L) continue;
        float yr, yi;
        if (a.do_pmd) {
            const float *c_self = a.pmd + (m == 0 ? 0 : 2) * SYNTH_NP, *c_other = a.pmd + SYNTH_NP;
            const Cx<float> *x_self = xs + (size_t)m * W + i + 2 * HP, *x_other = xs + (size_t)(1 - m) * W + i + 2 * HP;
            yr = yi = 0.f;
            for (int j = 0; j < SYNTH_NP; j++) {                                // y[n] = sum_j c[j] x[n + HP - j]
                const Cx<float> u = x_self[-j], v = x_other[-j];
                yr = fmaf(c_self[j], u.re, fmaf(c_other[j], v.re, yr));
                yi = fmaf(c_self[j], u.im, fmaf(c_other[j], v.im, yi));
            }
        } else {
            const Cx<float> u = xs[(size_t)m * W + i + HP];
            yr = u.re; yi = u.im;
        }
        if (a.sigma_noise > 0.f) {
            const Philox p = philox4x32_10((unsigned)n, (unsigned)(n >> 32), (unsigned)m, 2u, a.seed_lo, a.seed_hi);
            float g0, g1;
            gauss2(p.x, p.y, g0, g1);
            yr = fmaf(a.sigma_noise, g0, yr); yi = fmaf(a.sigma_noise, g1, yi);
        }
        a.E[(size_t)m * a.L + n] = Cx<float>{yr, yi};
    }
}

// ---- host side: filter design in double precision
static double bessel_i0(double x)
{
    double s = 1, t = 1;
    for (int k = 1; k < 50; k++) { t *= (x / (2 * k)) * (x / (2 * k)); s += t; if (t < 1e-18 * s) break; }
    return s;
}
static double kaiser(int i, int n, double beta)
{
    const double r = 2.0 * i / (n - 1) - 1.0;
    return bessel_i0(beta * sqrt(1 - r * r > 0 ? 1 - r * r : 0)) / bessel_i0(beta);
}
static double rrc_impulse(double t, double beta)        // t in symbol periods
{
    const double pi = 3.14159265358979323846;
    if (fabs(t) < 1e-12) return 1 - beta + 4 * beta / pi;
    if (beta > 0 && fabs(fabs(t) - 1 / (4 * beta)) < 1e-9)
        return beta / sqrt(2.0) * ((1 + 2 / pi) * sin(pi / (4 * beta)) + (1 - 2 / pi) * cos(pi / (4 * beta)));
    return (sin(pi * t * (1 - beta)) + 4 * beta * t * cos(pi * t * (1 + beta))) / (pi * t * (1 - (4 * beta * t) * (4 * beta * t)));
}

int synth_capture(void *E, void *symbols, int32_t *idx_tx, const void *alphabet, int M, int nmodes, int64_t nsym, int os, double beta,
                  double snr_db, int have_snr, double theta, double dgd_samples, int have_pmd, double phase_var, uint64_t seed)
{
    int rc = ensure_init();
    if (rc) return rc;
    QH_REQUIRE(nmodes >= 1 && nmodes <= 8 && nsym >= 2 * SYNTH_HSPAN && os >= 1 && os <= SYNTH_MAXOS && M >= 2 && M <= 4096 && beta >= 0 && beta <= 1,
               "synth: bad sizes");
    QH_REQUIRE(!have_pmd || nmodes == 2, "synth: PMD needs two modes");
    QH_REQUIRE(fabs(dgd_samples) < SYNTH_NP / 2 - 4, "synth: DGD too long for the fractional-delay filter");
    const int64_t L = nsym * os;
    const int NH = 2 * SYNTH_HSPAN * os + 1;
    std::vector<float> h(NH), pm(3 * SYNTH_NP, 0.f);
    {
        std::vector<double> hd(NH);
        double e = 0;
        for (int i = 0; i < NH; i++) {
            hd[i] = rrc_impulse((double)(i - SYNTH_HSPAN * os) / os, beta) * kaiser(i, NH, 6.0);
            e += hd[i] * hd[i];
        }
        // unit-power symbols: mean |x|^2 over the samples = sum h^2 / os
        const double g = 1 / sqrt(e / os);
        for (int i = 0; i < NH; i++) h[i] = (float)(hd[i] * g);
    }
    if (have_pmd) {
        const double pi = 3.14159265358979323846, c = cos(theta), s = sin(theta);
        const int HP = (SYNTH_NP - 1) / 2;
        std::vector<double> hp(SYNTH_NP), hm(SYNTH_NP);
        double sp = 0, sm = 0;
        for (int j = 0; j < SYNTH_NP; j++) {               // windowed-sinc fractional delays of +-dgd/2 samples
            const double tp = (j - HP) - dgd_samples / 2, tm = (j - HP) + dgd_samples / 2, w = kaiser(j, SYNTH_NP, 8.0);
            hp[j] = (fabs(tp) < 1e-12 ? 1 : sin(pi * tp) / (pi * tp)) * w;
            hm[j] = (fabs(tm) < 1e-12 ? 1 : sin(pi * tm) / (pi * tm)) * w;
            sp += hp[j]; sm += hm[j];
        }
        for (int j = 0; j < SYNTH_NP; j++) {
            hp[j] /= sp; hm[j] /= sm;
            pm[j] = (float)(c * c * hp[j] + s * s * hm[j]);                    // x0 -> y0
            pm[SYNTH_NP + j] = (float)(c * s * (hm[j] - hp[j]));               // x1 -> y0 and x0 -> y1
            pm[2 * SYNTH_NP + j] = (float)(s * s * hp[j] + c * c * hm[j]);     // x1 -> y1
        }
    }
    const int ntiles = (int)((L + SYNTH_T - 1) / SYNTH_T);
    void *buf = nullptr;
    const size_t b_h = (size_t)NH * 4, b_pm = 3 * SYNTH_NP * 4, b_ph = (size_t)nmodes * ntiles * 4;
    if ((rc = scratch(9, b_h + b_pm + b_ph + 64, &buf))) return rc;
    float *d_h = (float *)buf, *d_pm = d_h + NH, *d_ph = d_pm + 3 * SYNTH_NP;
    QH_HIP(hipMemcpyAsync(d_h, h.data(), b_h, hipMemcpyHostToDevice, g_stream));
    QH_HIP(hipMemcpyAsync(d_pm, pm.data(), b_pm, hipMemcpyHostToDevice, g_stream));
    QH_HIP(hipStreamSynchronize(g_stream));                 // the host vectors go out of scope
    SynthArgs a;
    a.E = (Cx<float> *)E; a.symbols = (Cx<float> *)symbols; a.idx_tx = idx_tx; a.alphabet = (const Cx<float> *)alphabet;
    a.hrrc = d_h; a.pmd = d_pm; a.tile_phase = d_ph; a.L = L; a.nsym = nsym; a.nmodes = nmodes; a.M = M; a.os = os; a.do_pmd = have_pmd ? 1 : 0;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    a.sigma_noise = have_snr ? (float)(pow(10.0, -snr_db / 20) * sqrt((double)os) / sqrt(2.0)) : 0.f;
    a.sigma_phase = phase_var > 0 ? (float)sqrt(phase_var) : 0.f;
    if (a.sigma_phase > 0.f) {
        hipLaunchKernelGGL(phase_tile_kernel, dim3(ntiles, nmodes), dim3(256), 0, g_stream, a, d_ph, ntiles);
        hipLaunchKernelGGL(phase_scan_kernel, dim3(nmodes), dim3(64), 0, g_stream, d_ph, ntiles);
    }
    const int HP = (SYNTH_NP - 1) / 2, W = SYNTH_T + 2 * HP, nk = (SYNTH_T + 2 * HP + 2 * SYNTH_HSPAN * os) / os + 2;
    const size_t lds = (size_t)nmodes * nk * 8 + (size_t)nmodes * W * 8 + (size_t)nmodes * W * 4 + (size_t)NH * 4;
    QH_REQUIRE(lds <= 64 * 1024, "synth: tile does not fit the LDS");
    hipLaunchKernelGGL(synth_kernel, dim3(ntiles), dim3(256), lds, g_stream, a);
    QH_HIP(hipGetLastError());
    return QH_OK;
}

}  // namespace qh

extern "C" int qh_synth_capture_c64_dev(void *E, void *symbols, int32_t *idx_tx, const void *alphabet, int M, int nmodes, int64_t nsym, int os,
                                        double beta, double snr_db, int have_snr, double theta, double dgd_samples, int have_pmd,
                                        double phase_var, uint64_t seed)
{
    return qh::synth_capture(E, symbols, idx_tx, alphabet, M, nmodes, nsym, os, beta, snr_db, have_snr, theta, dgd_samples, have_pmd, phase_var, seed);
}
