/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product; see oracle/README.md.
 *
 * Precision-generic body of the CPU restatement.  Included twice by qampy_oracle.c with
 *   R      = float / double
 *   FN(x)  = x##_f32 / x##_f64      (complex entry points: c64 / c128, real ones: f32 / f64)
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Arithmetic is written out on (re, im) pairs in the order numpy evaluates the reference's scalar
 * expressions, so that the strict build (-O2 -ffp-contract=off, no OpenMP) reproduces the pure-Python
 * run of the reference to rounding.
 */

typedef struct { R re, im; } FN(cplx);

static inline FN(cplx) FN(cmul)(FN(cplx) a, FN(cplx) b)
{
    FN(cplx) r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

static inline R FN(abs2_hyp)(R re, R im)
{
    /* abs(x)**2 with numpy semantics: hypot first, then square */
    R a = (R)HYPOT(re, im);
    return a * a;
}

/* partition_value, qampy/core/equalisation/pythran_equalisation.py:4-9.  `imag` selects np.imag instead of np.real */
static inline R FN(partition_value)(R signal, const FN(cplx) *partitions, int npart, const FN(cplx) *codebook, int imag)
{
    int index = 0;
    while (index < npart && signal > (imag ? partitions[index].im : partitions[index].re))
        index++;
    return imag ? codebook[index].im : codebook[index].re;
}

/* det_symbol, pythran_equalisation.py:240-265: first strict minimum of abs(X-s)**2, d0 = 1000, s = 1 */
static inline FN(cplx) FN(det_symbol_eq)(FN(cplx) X, const FN(cplx) *symbs, int M)
{
    R d0 = (R)1000.;
    FN(cplx) s = {(R)1, (R)0};
    for (int j = 0; j < M; j++) {
        R d = FN(abs2_hyp)(X.re - symbs[j].re, X.im - symbs[j].im);
        if (d < d0) { d0 = d; s = symbs[j]; }
    }
    return s;
}

/* error functions, pythran_equalisation.py:178-231; `i` is the training-step index (data-aided only) */
static inline FN(cplx) FN(errfn)(int method, FN(cplx) X, const FN(cplx) *sy, int nsy, long i)
{
    FN(cplx) e = {0, 0};
    switch (method) {
    case QO_CMA: case QO_SGNCMA: {            /* :178-180 (sgncma dispatches to cma_error, :133-134) */
        R d = sy[0].re - FN(abs2_hyp)(X.re, X.im);
        e.re = d * X.re; e.im = d * X.im;
        break; }
    case QO_CMA2: {                            /* :182-184  d = s1[0] - X**2 ; d*X (complex) */
        FN(cplx) x2 = FN(cmul)(X, X);
        FN(cplx) d = {sy[0].re - x2.re, sy[0].im - x2.im};
        e = FN(cmul)(d, X);
        break; }
    case QO_MCMA: {                            /* :190-194 */
        R dr = sy[0].re - X.re * X.re;
        R di = sy[0].im - X.im * X.im;
        e.re = dr * X.re; e.im = di * X.im;
        break; }
    case QO_RDE: {                             /* :196-200  array_split: first half gets the extra element */
        int ncode = (nsy + 1) / 2;
        R sq = FN(abs2_hyp)(X.re, X.im);
        R r = FN(partition_value)(sq, sy + ncode, nsy - ncode, sy, 0);
        R d = r - sq;
        e.re = X.re * d; e.im = X.im * d;
        break; }
    case QO_MRDE: {                            /* :203-211 */
        int ncode = (nsy + 1) / 2;
        R sqr = X.re * X.re, sqi = X.im * X.im;
        R rr = FN(partition_value)(sqr, sy + ncode, nsy - ncode, sy, 0);
        R ri = FN(partition_value)(sqi, sy + ncode, nsy - ncode, sy, 1);
        e.re = (rr - sqr) * X.re; e.im = (ri - sqi) * X.im;
        break; }
    case QO_SBD: {                             /* :214-217 */
        FN(cplx) s = FN(det_symbol_eq)(X, sy, nsy);
        e.re = (s.re - X.re) * FABS(s.re); e.im = (s.im - X.im) * FABS(s.im);
        break; }
    case QO_SBD_DATA: {                        /* :219-223 */
        FN(cplx) s = sy[i];
        e.re = (s.re - X.re) * FABS(s.re); e.im = (s.im - X.im) * FABS(s.im);
        break; }
    case QO_MDDMA: {                           /* :225-228 */
        FN(cplx) s = FN(det_symbol_eq)(X, sy, nsy);
        e.re = (s.re * s.re - X.re * X.re) * X.re; e.im = (s.im * s.im - X.im * X.im) * X.im;
        break; }
    case QO_DD: {                              /* :230-232 */
        FN(cplx) s = FN(det_symbol_eq)(X, sy, nsy);
        e.re = s.re - X.re; e.im = s.im - X.im;
        break; }
    }
    return e;
}

/* adapt_step, pythran_equalisation.py:12-16.  Called as adapt_step(mu, err[i], err[i-1]) (:172), i.e. the
 * magnitude in the denominator is that of the PREVIOUS error. */
static inline R FN(adapt_step)(R mu, FN(cplx) e_now, FN(cplx) e_prev)
{
    if (e_prev.re * e_now.re > 0 && e_prev.im * e_now.im > 0)
        return mu;
    return mu / (1 + mu * (e_prev.re * e_prev.re + e_prev.im * e_prev.im));
}

/*
 * train_equaliser, pythran_equalisation.py:128-173 (complex).
 *   E (nmodes, L) C-contiguous; wx (nmodes, nmodes, ntaps) in/out; symbols (nmodes, nsy); err (nmodes, TrSyms*Niter) out,
 *   caller zero-fills; modes[nsel]; *mu in/out.  Semantics = sequential (mu carried across modes and iterations),
 *   which is the compiled reference with OMP_NUM_THREADS=1.  With QO_OPENMP and !adaptive the mode loop is
 *   parallel exactly like :162.
 */
int FN(qo_train_equaliser)(const FN(cplx) *E, int nmodes, long L, long TrSyms, int Niter, int os, R *mu_io,
                           FN(cplx) *wx, int ntaps, const long *modes, int nsel, int adaptive,
                           const FN(cplx) *symbols, long nsy, int method, FN(cplx) *err)
{
    if (method < 0 || method > QO_SBD_DATA) return 1;   /* ValueError("Unknown method") :151-152 */
    R mu = *mu_io;
    const long nerr = TrSyms * (long)Niter;
#ifdef QO_OPENMP
#pragma omp parallel for if (!adaptive)
#endif
    for (int im = 0; im < nsel; im++) {
        const long mode = modes[im];
        FN(cplx) *w = wx + (size_t)mode * nmodes * ntaps;
        const FN(cplx) *sy = symbols + (size_t)mode * nsy;
        FN(cplx) *er = err + (size_t)mode * nerr;
        R mu_l = mu;                      /* private copy; written back only in the sequential (adaptive) case */
        for (int it = 0; it < Niter; it++) {
            for (long i = 0; i < TrSyms; i++) {
                FN(cplx) X = {0, 0};
                for (int k = 0; k < nmodes; k++) {          /* apply_filter :24-31, k outer, tap inner, no conjugate */
                    const FN(cplx) *x = E + (size_t)k * L + i * os;
                    const FN(cplx) *wk = w + (size_t)k * ntaps;
                    for (int t = 0; t < ntaps; t++) {
                        FN(cplx) p = FN(cmul)(x[t], wk[t]);
                        X.re += p.re; X.im += p.im;
                    }
                }
                FN(cplx) e = FN(errfn)(method, X, sy, (int)nsy, i);
                er[it * TrSyms + i] = e;
                FN(cplx) c = {mu_l * e.re, mu_l * e.im};     /* (mu*err) then * conj(X) then += , :170 */
                for (int k = 0; k < nmodes; k++) {
                    const FN(cplx) *x = E + (size_t)k * L + i * os;
                    FN(cplx) *wk = w + (size_t)k * ntaps;
                    for (int t = 0; t < ntaps; t++) {
                        FN(cplx) xc = {x[t].re, -x[t].im};
                        FN(cplx) p = FN(cmul)(c, xc);
                        wk[t].re += p.re; wk[t].im += p.im;
                    }
                }
                if (adaptive && i > 0)
                    mu_l = FN(adapt_step)(mu_l, e, er[it * TrSyms + i - 1]);
            }
        }
        if (adaptive) mu = mu_l;          /* sequential carry-over into the next mode */
    }
    *mu_io = mu;
    return 0;
}

/* real-valued error functions, pythran_equalisation.py:110-125 */
static inline R FN(errfn_real)(int method, R X, const R *sy, int nsy, long i)
{
    switch (method) {
    case QO_R_CMA: {
        R a = FABS(X);
        return (sy[0] - a * a) * X; }
    case QO_R_SGNCMA: {
        R a = FABS(X);
        R v = sy[0] - a * a;
        R d = (R)((v > 0) - (v < 0));
        return d * (R)((X > 0) - (X < 0)); }
    case QO_R_DD: {                       /* det_symbol_argmin :233-236 : first argmin of |X - s| */
        R best = FABS(X - sy[0]); int ib = 0;
        for (int j = 1; j < nsy; j++) { R d = FABS(X - sy[j]); if (d < best) { best = d; ib = j; } }
        return (sy[ib] - X) * FABS(sy[ib]); }
    case QO_R_DD_DATA: {
        R s = sy[i];
        return (s - X) * FABS(s); }
    }
    return 0;
}

/* train_equaliser_realvalued, pythran_equalisation.py:78-108; update has no conjugate, adapt_step_real :18-22 */
int FN(qo_train_equaliser_real)(const R *E, int nmodes, long L, long TrSyms, int Niter, int os, R *mu_io,
                                R *wx, int ntaps, const long *modes, int nsel, int adaptive,
                                const R *symbols, long nsy, int method, R *err)
{
    if (method < 0 || method > QO_R_DD_DATA) return 1;
    R mu = *mu_io;
    const long nerr = TrSyms * (long)Niter;
    for (int im = 0; im < nsel; im++) {
        const long mode = modes[im];
        R *w = wx + (size_t)mode * nmodes * ntaps;
        const R *sy = symbols + (size_t)mode * nsy;
        R *er = err + (size_t)mode * nerr;
        for (int it = 0; it < Niter; it++) {
            for (long i = 0; i < TrSyms; i++) {
                R X = 0;
                for (int k = 0; k < nmodes; k++)
                    for (int t = 0; t < ntaps; t++)
                        X += E[(size_t)k * L + i * os + t] * w[(size_t)k * ntaps + t];
                R e = FN(errfn_real)(method, X, sy, (int)nsy, i);
                er[it * TrSyms + i] = e;
                R c = mu * e;
                for (int k = 0; k < nmodes; k++)
                    for (int t = 0; t < ntaps; t++)
                        w[(size_t)k * ntaps + t] += c * E[(size_t)k * L + i * os + t];
                if (adaptive && i > 0) {
                    R ep = er[it * TrSyms + i - 1];
                    if (!(ep * e > 0)) mu = mu / (1 + mu * (ep * ep));
                }
            }
        }
    }
    *mu_io = mu;
    return 0;
}

/* apply_filter_to_signal, pythran_equalisation.py:33-76 (complex).  out (nsel, N), N = (L-ntaps+1)//os */
int FN(qo_apply_filter)(const FN(cplx) *E, int nmodes, long L, int os, const FN(cplx) *wx, int ntaps,
                        const long *modes, int nsel, FN(cplx) *out)
{
    const long N = (L - ntaps + 1) / os;
#ifdef QO_OPENMP
#pragma omp parallel for collapse(2)
#endif
    for (int j = 0; j < nsel; j++)
        for (long i = 0; i < N; i++) {
            const FN(cplx) *w = wx + (size_t)modes[j] * nmodes * ntaps;
            FN(cplx) X = {0, 0};
            for (int k = 0; k < nmodes; k++)
                for (int t = 0; t < ntaps; t++) {
                    FN(cplx) p = FN(cmul)(E[(size_t)k * L + i * os + t], w[(size_t)k * ntaps + t]);
                    X.re += p.re; X.im += p.im;
                }
            out[(size_t)j * N + i] = X;
        }
    return 0;
}

/* apply_filter_to_signal, real overloads (:33-34) */
int FN(qo_apply_filter_real)(const R *E, int nmodes, long L, int os, const R *wx, int ntaps,
                             const long *modes, int nsel, R *out)
{
    const long N = (L - ntaps + 1) / os;
#ifdef QO_OPENMP
#pragma omp parallel for collapse(2)
#endif
    for (int j = 0; j < nsel; j++)
        for (long i = 0; i < N; i++) {
            const R *w = wx + (size_t)modes[j] * nmodes * ntaps;
            R X = 0;
            for (int k = 0; k < nmodes; k++)
                for (int t = 0; t < ntaps; t++)
                    X += E[(size_t)k * L + i * os + t] * w[(size_t)k * ntaps + t];
            out[(size_t)j * N + i] = X;
        }
    return 0;
}

/*
 * bps, qampy/core/pythran_dsp.py:45-85 with det_symbol :16-23 and select_angle_index :26-42.
 *   comp = exp(1j*testangles) is formed by the caller (numpy) so the oracle sees the reference's exact rotators;
 *   comp has p rows (p == 1: one grid for all symbols, p == L: per-symbol grid of the two-stage search).
 *   idx (L,) int32 out.  dists and csum are materialised (L, A) like the reference does.
 */
int FN(qo_bps)(const FN(cplx) *E, long L, const FN(cplx) *comp, long p, int A, const FN(cplx) *symbols, int M,
               int N, int *idx)
{
    R *dists = (R *)malloc((size_t)L * A * sizeof(R));
    R *csum = (R *)calloc((size_t)L * A, sizeof(R));
    if (!dists || !csum) { free(dists); free(csum); return 2; }
#ifdef QO_OPENMP
#pragma omp parallel for
#endif
    for (long i = 0; i < L; i++) {
        const FN(cplx) *c = comp + (p > 1 ? (size_t)i * A : 0);
        for (int j = 0; j < A; j++) {
            FN(cplx) tmp = FN(cmul)(E[i], c[j]);
            R d0 = (R)1000.;
            for (int k = 0; k < M; k++) {
                R dr = tmp.re - symbols[k].re, di = tmp.im - symbols[k].im;
                R d = dr * dr + di * di;                      /* cabsq :3-4 */
                if (d < d0) d0 = d;
            }
            dists[(size_t)i * A + j] = d0 < (R)100. ? d0 : (R)100.;   /* dists starts at 100 (:73), strict < (:83) */
        }
    }
    /* select_angle_index(dists, 2N): serial in the reference (no omp pragma) */
    const long W = 2L * N;
    for (long i = 0; i < L; i++) idx[i] = 0;
    for (long i = 1; i < L; i++) {
        R dmin = (R)1000.;
        const R *x = dists + (size_t)i * A;
        R *cs = csum + (size_t)i * A;
        const R *cp = cs - A;
        if (i < W) {
            for (int k = 0; k < A; k++) cs[k] = cp[k] + x[k];
        } else {
            const R *cw = csum + (size_t)(i - W) * A;
            for (int k = 0; k < A; k++) {
                cs[k] = cp[k] + x[k];
                R dtmp = cs[k] - cw[k];
                if (dtmp < dmin) { idx[i - W / 2] = k; dmin = dtmp; }
            }
        }
    }
    free(dists); free(csum);
    return 0;
}

/* select_angles, pythran_dsp.py:133-153 */
int FN(qo_select_angles)(const R *angles, long p, int A, const long *idx, long L, R *out)
{
    for (long i = 0; i < L; i++)
        out[i] = angles[(p > 1 ? (size_t)i * A : 0) + idx[i]];
    return 0;
}

/* make_decision, pythran_equalisation.py:304-334 via det_symbol_argmin :233-236 (np.abs distance, first argmin) */
int FN(qo_make_decision)(const FN(cplx) *E, long L, const FN(cplx) *symbols, int M, FN(cplx) *det, R *dist, int *idx)
{
#ifdef QO_OPENMP
#pragma omp parallel for
#endif
    for (long i = 0; i < L; i++) {
        R best = (R)HYPOT(E[i].re - symbols[0].re, E[i].im - symbols[0].im);
        int ib = 0;
        for (int k = 1; k < M; k++) {
            R d = (R)HYPOT(E[i].re - symbols[k].re, E[i].im - symbols[k].im);
            if (d < best) { best = d; ib = k; }
        }
        det[i] = symbols[ib]; dist[i] = best; idx[i] = ib;
    }
    return 0;
}
