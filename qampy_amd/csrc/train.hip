// train_equaliser / train_equaliser_realvalued on gfx950.
//
// Reference behaviour: qampy/core/equalisation/pythran_equalisation.py:128-173 (complex), :78-108 (real),
// error functions :178-231 / :110-125, adapt_step :12-22, partition_value :4-9, det_symbol :240-265.
//
// The recurrence  w[i+1] = w[i] + mu*e(w[i].x[i])*conj(x[i])  is strictly sequential in i, so one output mode is ONE
// dependent chain and this kernel is latency bound, not HBM or MFMA bound.  Mapping:
//   * one wave64 per chain; the nmodes*ntaps taps are spread round-robin over the 64 lanes (TPL taps per lane) and
//     live in VGPRs for the whole sweep;
//   * per step every lane multiplies its taps with its samples, a 6-level DPP butterfly (row_* / row_bcast) sums
//     re and im across the wave, the error function is evaluated wave-uniformly, and every lane updates its taps;
//   * the samples of step i+PD are loaded while step i computes (rotating register buffer, no LDS round trip);
//     the window slides by `os` samples per step so these loads hit the CU's L1/L2;
//   * decision-directed methods search the alphabet lane-parallel (lane j <-> symbol j) and pick the FIRST minimum
//     with ballot + ff1, which is exactly the strict-`<` scan of det_symbol; RDE/MRDE partition look-ups are a ballot
//     over lane-held partitions;
//   * with adaptive step size the modes run back to back in one wave because the reference carries `mu` from one
//     mode into the next when run sequentially; otherwise each mode gets its own workgroup (its own CU).
#include "common.h"

namespace qh {

// sample prefetch distance in steps (shrinks with the taps-per-lane count to keep the queue in registers)
template <int TPL> struct Prefetch { static constexpr int PD = TPL <= 2 ? 4 : (TPL <= 4 ? 2 : 1); };
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
};

// ------------------------------------------------------------------------------------------------ error functions
template <typename R> struct Tables {
    // lane-resident copies of symbols[mode, :]  (lane j holds entry j and, for the split tables, entry ncode + j)
    R a_re, a_im;   // alphabet / codebook entry of this lane
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
    if (!T.serial) {
        R dr = X.re - T.a_re, di = X.im - T.a_im;
        R d = fma_(dr, dr, di * di);
        if (lane >= T.n) d = (R)3.0e38;
        R dmin = wave_min(d);
        if (!(dmin < (R)1000.)) return Cx<R>{(R)1, (R)0};           // det_symbol's initial value survives (:258-259)
        unsigned long long m = __ballot(d == dmin);
        int j = __builtin_ctzll(m);                                 // first minimum == strict `<` scan order
        return Cx<R>{readlane(T.a_re, j), readlane(T.a_im, j)};
    }
    // serial scan for alphabets larger than a wave (wave-uniform, every lane does the same work)
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
template <typename R, int TPL, int METHOD>
__device__ __forceinline__ R run_chain(const TrainArgs<R> &a, int mode, R mu, int lane)
{
    const int ntot = a.nmodes * a.ntaps;
    const int64_t L = a.L;
    Cx<R> w[TPL];
    const Cx<R> *xbase[TPL];
    bool valid[TPL];
    Cx<R> *wrow = a.wx + (size_t)mode * ntot;
#pragma unroll
    for (int s = 0; s < TPL; s++) {
        int f = lane + 64 * s;
        valid[s] = f < ntot;
        int fc = valid[s] ? f : 0;
        int k = fc / a.ntaps, t = fc - k * a.ntaps;
        xbase[s] = a.E + (size_t)k * L + t;
        w[s] = valid[s] ? ldg(wrow + fc) : Cx<R>{0, 0};
    }
    // per-mode constants and tables
    const Cx<R> *sy = a.symbols + (size_t)mode * a.nsy;
    Tables<R> T;
    T.glob = sy;
    T.n = (int)a.nsy;
    T.ncode = (T.n + 1) / 2;           // np.array_split(symbs, 2): the first half takes the extra element
    T.npart = T.n - T.ncode;
    T.a_re = T.a_im = T.p_re = T.p_im = 0;
    T.serial = false;
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
        T.serial = T.n > MAX_TABLE;
    }

    Cx<R> *errow = a.err + (size_t)mode * (a.TrSyms * a.Niter);
    const int64_t TrSyms = a.TrSyms;
    const int os = a.os;

    constexpr int PD = Prefetch<TPL>::PD;
    for (int it = 0; it < a.Niter; it++) {
        Cx<R> xq[PD][TPL];
        Cx<R> dq[PD];
#pragma unroll
        for (int u = 0; u < PD; u++) dq[u] = Cx<R>{0, 0};
        // prime the prefetch queue
#pragma unroll
        for (int u = 0; u < PD; u++) {
            int64_t ii = u < TrSyms ? u : TrSyms - 1;
#pragma unroll
            for (int s = 0; s < TPL; s++) xq[u][s] = valid[s] ? ldg(xbase[s] + ii * os) : Cx<R>{0, 0};
            if constexpr (METHOD == QH_M_SBD_DATA) dq[u] = ldg(sy + ii);
        }
        Cx<R> e_prev{0, 0};
        Cx<R> *eout = errow + (size_t)it * TrSyms;
        for (int64_t i0 = 0; i0 < TrSyms; i0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; u++) {
                const int64_t i = i0 + u;
                if (i < TrSyms) {
                    Cx<R> x[TPL];
#pragma unroll
                    for (int s = 0; s < TPL; s++) x[s] = xq[u][s];
                    Cx<R> dsym = dq[u];
                    // refill this queue slot with the samples of step i + PD
                    {
                        int64_t ii = i + PD < TrSyms ? i + PD : TrSyms - 1;
#pragma unroll
                        for (int s = 0; s < TPL; s++) xq[u][s] = valid[s] ? ldg(xbase[s] + ii * os) : Cx<R>{0, 0};
                        if constexpr (METHOD == QH_M_SBD_DATA) dq[u] = ldg(sy + ii);
                    }
                    // Xest = sum_f x[f] * w[f]   (no conjugate, :24-31)
                    R pr = 0, pi = 0;
#pragma unroll
                    for (int s = 0; s < TPL; s++) {
                        pr = fma_(x[s].re, w[s].re, pr); pr = fma_(-x[s].im, w[s].im, pr);
                        pi = fma_(x[s].re, w[s].im, pi); pi = fma_(x[s].im, w[s].re, pi);
                    }
                    wave_sum2(pr, pi);
                    Cx<R> X{pr, pi};
                    Cx<R> e = error_fn<R, METHOD>(X, T, R_re, R_im, dsym, lane);
                    if (lane == 0) stg(eout + i, e);
                    // w += (mu*e) * conj(x)   (:170)
                    R cr = mu * e.re, ci = mu * e.im;
#pragma unroll
                    for (int s = 0; s < TPL; s++) {
                        w[s].re = fma_(cr, x[s].re, fma_(ci, x[s].im, w[s].re));
                        w[s].im = fma_(ci, x[s].re, fma_(-cr, x[s].im, w[s].im));
                    }
                    if (a.adaptive && i > 0) {                 // adapt_step(mu, err[i], err[i-1]) :12-16, :171-172
                        bool keep = (e_prev.re * e.re > 0) && (e_prev.im * e.im > 0);
                        R den = fma_(mu, fma_(e_prev.re, e_prev.re, e_prev.im * e_prev.im), (R)1);
                        mu = keep ? mu : mu / den;
                    }
                    e_prev = e;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < TPL; s++)
        if (valid[s]) stg(wrow + lane + 64 * s, w[s]);
    return mu;
}

template <typename R, int TPL, int METHOD>
__global__ void __launch_bounds__(64) train_kernel(TrainArgs<R> a)
{
    const int lane = threadIdx.x;
    R mu = *a.mu;
    if (a.adaptive) {
        // sequential semantics: one wave walks the modes in order and carries mu
        for (int j = 0; j < a.nsel; j++) mu = run_chain<R, TPL, METHOD>(a, (int)a.modes[j], mu, lane);
        if (lane == 0) *a.mu = mu;
    } else {
        run_chain<R, TPL, METHOD>(a, (int)a.modes[blockIdx.x], mu, lane);
    }
}

template <typename R, int TPL> static int launch_tpl(const TrainArgs<R> &a)
{
    dim3 grid(a.adaptive ? 1 : a.nsel), block(64);
#define QH_CASE(M) case M: hipLaunchKernelGGL((train_kernel<R, TPL, M>), grid, block, 0, g_stream, a); break;
    switch (a.method) {
        QH_CASE(QH_M_CMA) QH_CASE(QH_M_CMA2) QH_CASE(QH_M_SGNCMA) QH_CASE(QH_M_MCMA) QH_CASE(QH_M_RDE) QH_CASE(QH_M_MRDE)
        QH_CASE(QH_M_SBD) QH_CASE(QH_M_MDDMA) QH_CASE(QH_M_DD) QH_CASE(QH_M_SBD_DATA)
    default: return QH_ERR_METHOD;
    }
#undef QH_CASE
    QH_HIP(hipGetLastError());
    return QH_OK;
}

template <typename R>
int train_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, R *mu_dev, void *wx, int ntaps,
              const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy, int method, void *err,
              int zero_err)
{
    int rc = ensure_init();
    if (rc) return rc;
    if (method < 0 || method > QH_M_SBD_DATA) { set_error("unknown equaliser method id"); return QH_ERR_METHOD; }
    QH_REQUIRE(nmodes >= 1 && ntaps >= 1 && os >= 1 && Niter >= 0 && TrSyms >= 0, "train_equaliser: bad sizes");
    QH_REQUIRE(nsel >= 1 && nsel <= 16, "train_equaliser: between 1 and 16 modes can be selected");
    QH_REQUIRE(TrSyms == 0 || (TrSyms - 1) * os + ntaps <= L, "train_equaliser: field shorter than TrSyms*os + ntaps");
    QH_REQUIRE(nsy >= 1, "train_equaliser: empty symbols array");
    QH_REQUIRE(method != QH_M_SBD_DATA || nsy >= TrSyms, "train_equaliser: sbd_data needs >= TrSyms training symbols");
    for (int j = 0; j < nsel; j++) QH_REQUIRE(modes[j] >= 0 && modes[j] < nmodes, "train_equaliser: mode number >= nmodes");
    const int ntot = nmodes * ntaps;
    QH_REQUIRE(ntot <= 64 * 16, "train_equaliser: more than 1024 taps per output mode are not supported");
    if (zero_err) QH_HIP(hipMemsetAsync(err, 0, (size_t)nmodes * TrSyms * Niter * sizeof(Cx<R>), g_stream));
    if (TrSyms == 0 || Niter == 0) return QH_OK;
    TrainArgs<R> a;
    a.E = (const Cx<R> *)E; a.wx = (Cx<R> *)wx; a.symbols = (const Cx<R> *)symbols; a.err = (Cx<R> *)err; a.mu = mu_dev;
    a.L = L; a.TrSyms = TrSyms; a.nsy = nsy; a.nmodes = nmodes; a.ntaps = ntaps; a.Niter = Niter; a.os = os;
    a.nsel = nsel; a.adaptive = adaptive ? 1 : 0; a.method = method;
    for (int j = 0; j < 16; j++) a.modes[j] = j < nsel ? modes[j] : 0;
    if (ntot <= 64) return launch_tpl<R, 1>(a);
    if (ntot <= 128) return launch_tpl<R, 2>(a);
    if (ntot <= 256) return launch_tpl<R, 4>(a);
    if (ntot <= 512) return launch_tpl<R, 8>(a);
    return launch_tpl<R, 16>(a);
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
        else if (ntot <= 512) rc = launch_real_tpl<R, 8>(a);
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

// ================================================================================================ C ABI
extern "C" {

int qh_train_equaliser_c64(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu, void *wx,
                           int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                           int method, void *err)
{
    return qh::train_host<float>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_train_equaliser_c128(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu, void *wx,
                            int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                            int method, void *err)
{
    return qh::train_host<double>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_train_equaliser_c64_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                               void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                               int64_t nsy, int method, void *err, int zero_err)
{
    return qh::train_dev<float>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err);
}
int qh_train_equaliser_c128_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                int64_t nsy, int method, void *err, int zero_err)
{
    return qh::train_dev<double>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err);
}
int qh_train_equaliser_real_f32(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu, void *wx,
                                int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                                int method, void *err)
{
    return qh::train_real_host<float>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_train_equaliser_real_f64(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu, void *wx,
                                int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                                int method, void *err)
{
    return qh::train_real_host<double>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}

}  // extern "C"
