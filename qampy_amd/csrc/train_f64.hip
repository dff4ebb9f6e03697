// C ABI of the trainers, double precision (complex128 / float64).  Kernels: train_impl.h
#include "train_pit.h"

extern "C" {
int qh_train_equaliser_c128(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu, void *wx,
                            int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                            int method, void *err)
{
    // the process-wide default tier (qh_set_default_tier): a = the exact sequential recurrence, b = the same recurrence solved in parallel in time
    if (qh::default_tier() == 1)
        return qh::train_host_tier_b<double>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, qh::default_tier_tol());
    return qh::train_host<double>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_train_equaliser_c128_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                int64_t nsy, int method, void *err, int zero_err)
{
    return qh::train_dev<double>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err);
}
int qh_train_equaliser_real_f64(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu, void *wx,
                                int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                                int method, void *err)
{
    return qh::train_real_host<double>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_gram_build_c128_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram)
{
    // pairs (look-ahead layout, also read by the block-iterative kernel) whenever the look-ahead kernel fits the shape
    if (!qh::la_shape_ok(nmodes, ntaps, os) && qh::bi_shape_ok(nmodes, ntaps, os, 2 * sizeof(double))) return qh::gram_cur_build<double>(E, nmodes, L, os, ntaps, TrSyms, gram);
    return qh::gram_build<double>(E, nmodes, L, os, ntaps, TrSyms, gram);
}
int qh_train_equaliser_c128_gram_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                    void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                    int64_t nsy, int method, void *err, int zero_err, const void *gram)
{
    return qh::train_dev<double>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err, gram);
}
int qh_train_equaliser_c128_pit_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev, void *wx, int ntaps,
                                    const int64_t *modes, int nsel, const void *symbols, int64_t nsy, int method, void *err, int zero_err,
                                    const void *gram, const qh_pit_opts *opts, void *report_dev)
{
    return qh::train_pit_dev<double>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, symbols, nsy, method, err, zero_err, gram, opts, report_dev);
}
int qh_pit_basis_c128_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void *basis, int overlap)
{
    return qh::pit_basis<double>(E, nmodes, L, os, ntaps, TrSyms, basis, overlap);
}
int qh_gram_build_c128_batch_dev(const void *E, int nch, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram)
{
    if (!qh::la_shape_ok(nmodes, ntaps, os) && qh::bi_shape_ok(nmodes, ntaps, os, 2 * sizeof(double))) return qh::gram_cur_build<double>(E, nmodes, L, os, ntaps, TrSyms, gram, nch);
    return qh::gram_build<double>(E, nmodes, L, os, ntaps, TrSyms, gram, nch);
}
int qh_train_equaliser_c128_batch_dev(const void *E, int nch, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                      void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                      int64_t nsy, int method, void *err, int zero_err, const void *gram)
{
    return qh::train_dev<double>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err, gram, nch);
}
int qh_train_equaliser_windows_search_c128(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                            int64_t TrSyms, int Niter, int os, double mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                            int adaptive, const void *symbols, int64_t nsy, int method, double *var, int32_t *best, void *wx_best)
{
    return qh::train_windows_host<double>(E, nmodes, L, win_start, nwin, win_len, TrSyms, Niter, os, mu, wx0, ntaps, modes, nsel, adaptive,
                                      symbols, nsy, method, nullptr, nullptr, nullptr, var, best, wx_best);
}
int qh_train_equaliser_windows_c128(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                     int64_t TrSyms, int Niter, int os, double mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                     int adaptive, const void *symbols, int64_t nsy, int method, void *wx_out, void *err, double *mu_out)
{
    return qh::train_windows_host<double>(E, nmodes, L, win_start, nwin, win_len, TrSyms, Niter, os, mu, wx0, ntaps, modes, nsel, adaptive,
                                      symbols, nsy, method, wx_out, err, mu_out);
}
}
