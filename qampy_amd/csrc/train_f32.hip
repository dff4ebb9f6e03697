// C ABI of the trainers, single precision (complex64 / float32).  Kernels: train_impl.h
#include "train_pit.h"

extern "C" {
int qh_train_equaliser_c64(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu, void *wx,
                           int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                           int method, void *err)
{
    // the process-wide default tier (qh_set_default_tier): a = the exact sequential recurrence, b = the same recurrence solved in parallel in time
    if (qh::default_tier() == 1)
        return qh::train_host_tier_b<float>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, qh::default_tier_tol());
    return qh::train_host<float>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_train_equaliser_c64_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                               void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                               int64_t nsy, int method, void *err, int zero_err)
{
    return qh::train_dev<float>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err);
}
int qh_train_equaliser_real_f32(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu, void *wx,
                                int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                                int method, void *err)
{
    return qh::train_real_host<float>(E, nmodes, L, TrSyms, Niter, os, mu, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err);
}
int qh_gram_build_c64_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram)
{
    // pairs (look-ahead layout, also read by the block-iterative kernel) whenever the look-ahead kernel fits the shape
    if (!qh::la_shape_ok(nmodes, ntaps, os) && qh::bi_shape_ok(nmodes, ntaps, os, 2 * sizeof(float))) return qh::gram_cur_build<float>(E, nmodes, L, os, ntaps, TrSyms, gram);
    return qh::gram_build<float>(E, nmodes, L, os, ntaps, TrSyms, gram);
}
int qh_train_equaliser_c64_gram_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                                    void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                    int64_t nsy, int method, void *err, int zero_err, const void *gram)
{
    return qh::train_dev<float>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err, gram);
}
int qh_train_equaliser_c64_pit_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev, void *wx, int ntaps,
                                    const int64_t *modes, int nsel, const void *symbols, int64_t nsy, int method, void *err, int zero_err,
                                    const void *gram, const qh_pit_opts *opts, void *report_dev)
{
    return qh::train_pit_dev<float>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, symbols, nsy, method, err, zero_err, gram, opts, report_dev);
}
int qh_last_pit_report(qh_pit_report *out)
{
    *out = qh::last_host_report();                   // the calling thread's most recent host-array solve through tier b (qh_set_default_tier)
    return QH_OK;
}
int qh_pit_last_timing(float *pass_ms, int max_passes, int *npass, float *acq_ms)
{
    const qh::PitTiming &t = qh::pit_timing();
    for (int i = 0; i < t.npass && i < max_passes; i++) pass_ms[i] = t.pass_ms[i];
    *npass = t.npass < max_passes ? t.npass : max_passes;
    *acq_ms = t.acq_ms;
    return QH_OK;
}
int qh_pit_prepare_bytes(int nmodes, int ntaps, int64_t acq_steps, size_t elem_bytes, size_t *bytes)
{
    *bytes = elem_bytes == 16 ? qh::pit_prep_layout<double>(nmodes, ntaps, acq_steps).total : qh::pit_prep_layout<float>(nmodes, ntaps, acq_steps).total;
    return QH_OK;
}
int qh_pit_prepare_c64_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int os, const float *mu_dev, const void *wx0, int ntaps,
                           const int64_t *modes, int nsel, const void *symbols, int64_t nsy, int method, const qh_pit_opts *opts, void *prep, size_t prep_bytes)
{
    return qh::pit_prepare<float>(E, nmodes, L, TrSyms, os, mu_dev, wx0, ntaps, modes, nsel, symbols, nsy, method, opts, prep, prep_bytes);
}
int qh_pit_auto_segments(int64_t TrSyms, double mu, int nsel, int cold, int *segments)
{
    *segments = qh::pit_auto_segments(TrSyms, mu, nsel, cold);
    return QH_OK;
}
int qh_pit_basis_c64_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void *basis, int overlap)
{
    return qh::pit_basis<float>(E, nmodes, L, os, ntaps, TrSyms, basis, overlap);
}
int qh_pit_basis_bytes(int ntot, size_t *bytes)
{
    *bytes = qh::pit_basis_bytes(ntot);
    return QH_OK;
}
int qh_gram_build_c64_batch_dev(const void *E, int nch, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram)
{
    if (!qh::la_shape_ok(nmodes, ntaps, os) && qh::bi_shape_ok(nmodes, ntaps, os, 2 * sizeof(float))) return qh::gram_cur_build<float>(E, nmodes, L, os, ntaps, TrSyms, gram, nch);
    return qh::gram_build<float>(E, nmodes, L, os, ntaps, TrSyms, gram, nch);
}
int qh_train_equaliser_c64_batch_dev(const void *E, int nch, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                                      void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                      int64_t nsy, int method, void *err, int zero_err, const void *gram)
{
    return qh::train_dev<float>(E, nmodes, L, TrSyms, Niter, os, mu_dev, wx, ntaps, modes, nsel, adaptive, symbols, nsy, method, err, zero_err, gram, nch);
}
int qh_train_equaliser_windows_search_c64(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                            int64_t TrSyms, int Niter, int os, float mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                            int adaptive, const void *symbols, int64_t nsy, int method, double *var, int32_t *best, void *wx_best)
{
    return qh::train_windows_host<float>(E, nmodes, L, win_start, nwin, win_len, TrSyms, Niter, os, mu, wx0, ntaps, modes, nsel, adaptive,
                                      symbols, nsy, method, nullptr, nullptr, nullptr, var, best, wx_best);
}
int qh_train_equaliser_windows_c64(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                     int64_t TrSyms, int Niter, int os, float mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                     int adaptive, const void *symbols, int64_t nsy, int method, void *wx_out, void *err, float *mu_out)
{
    return qh::train_windows_host<float>(E, nmodes, L, win_start, nwin, win_len, TrSyms, Niter, os, mu, wx0, ntaps, modes, nsel, adaptive,
                                      symbols, nsy, method, wx_out, err, mu_out);
}
}
