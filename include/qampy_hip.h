/*
 * qampy_hip.h - C ABI of libqampy_hip.so: hand-written HIP (gfx950 / MI355X) kernels behind QAMpy's compiled hot path.
 *
 * What this replaces.  QAMpy's only native code are two pythran-compiled modules; their `#pythran export` lines are the
 * de-facto FFI contract of the hot path (paths relative to the QAMpy tree):
 *   qampy/core/equalisation/pythran_equalisation.py:33-36    apply_filter_to_signal  {f32,f64,c64,c128}
 *   qampy/core/equalisation/pythran_equalisation.py:78-79    train_equaliser_realvalued {f32,f64}
 *   qampy/core/equalisation/pythran_equalisation.py:128-129  train_equaliser {c64,c128}
 *   qampy/core/equalisation/pythran_equalisation.py:304-305  make_decision {c64,c128}
 *   qampy/core/pythran_dsp.py:45-46                          bps {c64,c128}
 *   qampy/core/pythran_dsp.py:133-136                        select_angles {f32,f64}
 * Call sites that bind to them: qampy/core/equalisation/equalisation.py:177,180,555,557;
 * qampy/core/phaserecovery.py:28-29,149-150; qampy/core/signal_quality.py:26.
 *
 * Conventions
 *   - plain pointers and sizes only; complex data is interleaved (re, im) exactly like numpy complex64/complex128;
 *   - all arrays C-contiguous, row-major, shapes given in the comments;
 *   - every function returns a status code (QH_OK == 0); the library never frees or keeps caller memory;
 *   - `qh_*`      : pointers are HOST memory (numpy arrays); the call stages through HBM and is synchronous;
 *   - `qh_*_dev`  : array pointers are DEVICE memory obtained from qh_malloc; the call only enqueues work on the
 *                   library's stream (use qh_sync / events).  Small control arrays (`modes`) stay host pointers.
 *   - method ids are the QH_M_* / QH_RM_* enums below (the reference matches strings, pythran_equalisation.py:131-152).
 *
 * Semantics are those of the reference run sequentially (OMP_NUM_THREADS=1): modes are processed in the order given
 * and, with adaptive != 0, the adapted step size is carried from one mode into the next.
 */
#ifndef QAMPY_HIP_H
#define QAMPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ABI version: bumped whenever an entry point changes its argument list or a public struct its layout, so that a caller
 * built against another header fails loudly (compare with qh_abi_version() at load time) instead of passing shifted
 * arguments.  History: 1 = round 1; 2 = round 2 (qh_bps_recover_*_dev gained `angles`, qh_train_equaliser_*_pit_dev takes
 * (gram, opts, report), the *_seg_dev entry points were removed - unversioned at the time); 3 = round 3 (qh_pit_opts:
 * start, dev_safety; qh_pit_report: deviation[]); 4 = qh_pit_opts: adaptive. */
#define QH_ABI_VERSION 10
int qh_abi_version(void);

/* ---- status codes (python shim: 1,2 -> ValueError, 3,4 -> RuntimeError) */
#define QH_OK 0
#define QH_ERR_METHOD 1   /* unknown method id   (reference: ValueError, pythran_equalisation.py:151-152) */
#define QH_ERR_ARG 2      /* inconsistent sizes  (reference: asserts that the compiled build drops)          */
#define QH_ERR_HIP 3      /* HIP runtime failure, text in qh_last_error()                                      */
#define QH_ERR_NODEVICE 4 /* no usable gfx950 device                                                           */

/* ---- complex trainer methods, pythran_equalisation.py:131-150 */
enum { QH_M_CMA = 0, QH_M_CMA2, QH_M_SGNCMA, QH_M_MCMA, QH_M_RDE, QH_M_MRDE, QH_M_SBD, QH_M_MDDMA, QH_M_DD, QH_M_SBD_DATA };
/* ---- real-valued trainer methods, pythran_equalisation.py:81-88 */
enum { QH_RM_CMA = 0, QH_RM_SGNCMA, QH_RM_DD, QH_RM_DD_DATA };

/* ---- device / runtime --------------------------------------------------------------------------------------- */
int qh_device_count(int *count);
int qh_init(int device);                       /* select the device and create the library streams; idempotent for the same device; ONE device
                                                * per process: a second call with another index fails with QH_ERR_ARG (run one process per GPU) */
int qh_device_name(char *buf, size_t n);
const char *qh_last_error(void);
int qh_sync(void);                             /* wait for the library stream */
int qh_malloc(void **dptr, size_t bytes);
int qh_free(void *dptr);
int qh_memset(void *dptr, int value, size_t bytes);
int qh_memcpy_h2d(void *dptr, const void *hptr, size_t bytes);
int qh_memcpy_d2h(void *hptr, const void *dptr, size_t bytes);
int qh_memcpy_d2d(void *dst, const void *src, size_t bytes);
/* Pinned host memory from a library pool + asynchronous copies on the current library stream (ABI 8): the mirrored host layers return
 * ndarrays that view pooled pinned buffers (one DMA at the PCIe rate, no bounce copy, no first-touch page faults), and copy a stage's
 * error trace back while the next stage trains.  qh_stream_sync waits for the current stream only. */
int qh_pinned_alloc(void **hptr, size_t bytes);
int qh_pinned_free(void *hptr);
int qh_memcpy_h2d_async(void *dptr, const void *hptr, size_t bytes);
int qh_memcpy_d2h_async(void *hptr, const void *dptr, size_t bytes);
int qh_stream_sync(void);
/* Threading: the library keeps per-process state (current device and stream, grow-only scratch buffers, trainer selection) and
 * is meant to be driven by ONE host thread, like the reference's extension modules under the GIL; error text is per thread.
 * qh_release_scratch frees the grow-only scratch buffers (Gram tables above all - up to qh_set_gram_budget_gb) after
 * draining both streams; they are re-allocated on demand. */
int qh_release_scratch(void);
int qh_thread_release(void);                   /* destroy the calling thread's streams and scratch buffers (they come back on the next call): a worker
                                                  thread before it ends, the main thread at exit (qampy_amd._lib registers it with atexit) */
/* Three library streams.  Every entry point enqueues on the CURRENT one (0 after qh_init); qh_use_stream(0..2) switches it,
 * qh_stream_wait_event makes the current stream wait for an event recorded on another one, qh_sync drains all of them.
 * Streams 0 and 1 are equals (a tier-b trainer puts its eigen-solver on whichever of the two is not current); stream 2 has the
 * lowest queue priority and is meant for chip-wide streaming work (phase search of capture k) overlapped with the latency-bound
 * trainers of capture k+1 on stream 0 (ResidentReceiver.run(overlap=True)).
 * Streams, scratch buffers and the events of the tier-b solver are PER HOST THREAD: a thread that calls into the library gets its own
 * set on first use (same device), so several captures can be in flight on one GPU, one driving thread each (pipeline.py ReceiverGroup);
 * qh_sync / qh_release_scratch act on the calling thread's set.  Device memory and the staging pool are per process.
 * Within a thread scratch buffers are per library, not per stream: overlap only stages that use different ones (trainers + Gram tables on
 * one stream; filter, phase search and SER harness on the other - what ChannelBank.run_pipelined does). */
int qh_use_stream(int idx);
int qh_stream_wait_event(void *ev);
int qh_stream_handle(void **stream);           /* the current library stream as a hipStream_t, for collectives enqueued next to the kernels (RCCL: qampy_amd/comm.py) */
/* HIP events recorded on the current library stream (bench.py measures kernel time with these) */
int qh_event_create(void **ev);
int qh_event_destroy(void *ev);
int qh_event_record(void *ev);
int qh_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on `stop` */

/* ---- train_equaliser (complex) ----------------------------------------------------------------------------------
 *   E        (nmodes, L)                 input field
 *   mu       in/out step size (the adapted value is returned when adaptive != 0)
 *   adaptive 0 fixed step; 1 adapt_step with the reference's sequential semantics (mu carried from sweep to sweep and
 *            from mode to mode, pythran_equalisation.py:162-172 with one thread); 2 (extension) one step size per mode:
 *            exactly the result of one call per selected mode from the initial mu, modes trained concurrently, mu out =
 *            the last mode's (the compiled reference's OpenMP threads share and race on mu; 2 is its deterministic stand-in)
 *   wx       (nmodes, nmodes, ntaps)     taps, updated in place
 *   modes    (nsel,) int64               output modes to train, processed in this order
 *   symbols  (nmodes, nsy)               per-method constants / alphabet / training symbols
 *   err      (nmodes, TrSyms*Niter)      out; rows of unselected modes are zeroed
 * Requires (TrSyms-1)*os + ntaps <= L, nsel >= 1, modes[i] < nmodes, and nsy >= TrSyms for QH_M_SBD_DATA.
 */
int qh_train_equaliser_c64(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu, void *wx,
                           int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                           int method, void *err);
int qh_train_equaliser_c128(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu, void *wx,
                            int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                            int method, void *err);
/* device-resident: E, wx, symbols, err, mu are device pointers; err is NOT zeroed for unselected modes unless zero_err */
int qh_train_equaliser_c64_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                               void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                               int64_t nsy, int method, void *err, int zero_err);
int qh_train_equaliser_c128_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                int64_t nsy, int method, void *err, int zero_err);

/* Gram terms of the look-ahead trainer: G(l, i) = sum_f conj(x_l[f]) x_i[f] for the 127 steps after l, laid out for the
 * kernel.  They depend on the capture only, so one build serves every mode, stage and sweep over the same
 * (E, os, ntaps, TrSyms).  The buffer is library-owned scratch: valid until the next qh_gram_build_* call; pass it to
 * qh_train_equaliser_*_gram_dev, or pass NULL there (and to the plain _dev / host entry points) to build it internally. */
int qh_gram_build_c64_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram);
int qh_gram_build_c128_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram);
int qh_train_equaliser_c64_gram_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                                    void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                    int64_t nsy, int method, void *err, int zero_err, const void *gram);
int qh_train_equaliser_c128_gram_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                     void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols,
                                     int64_t nsy, int method, void *err, int zero_err, const void *gram);

/* Batch of independent equaliser runs on `nwin` windows E[:, win_start[v] : win_start[v] + win_len] of one capture, all from
 * the same initial taps wx0 and step size mu - what the frame synchronisation of the pilot receiver does in a Python loop
 * (qampy/core/pilotbased_receiver.py:395-400).  Results are those of nwin separate qh_train_equaliser_* calls; the windows
 * run concurrently (one wavefront each).  wx_out (nwin, nmodes, nmodes, ntaps), err (nwin, nmodes, TrSyms*Niter), mu_out (nwin). */
int qh_train_equaliser_windows_c64(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                   int64_t TrSyms, int Niter, int os, float mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                   int adaptive, const void *symbols, int64_t nsy, int method, void *wx_out, void *err, float *mu_out);
int qh_train_equaliser_windows_c128(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                    int64_t TrSyms, int Niter, int os, double mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                    int adaptive, const void *symbols, int64_t nsy, int method, void *wx_out, void *err, double *mu_out);

/* Search form of the window batch: the error traces stay in HBM; out come var (nmodes, nwin) = variance of every window's
 * error trace per mode (np.var of the complex row), best (nmodes) = window with the smallest variance per mode (first
 * minimum) and wx_best (nmodes, nmodes, nmodes, ntaps) = the tap sets of those windows. */
int qh_train_equaliser_windows_search_c64(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                          int64_t TrSyms, int Niter, int os, float mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                          int adaptive, const void *symbols, int64_t nsy, int method, double *var, int32_t *best, void *wx_best);
int qh_train_equaliser_windows_search_c128(const void *E, int nmodes, int64_t L, const int64_t *win_start, int nwin, int64_t win_len,
                                           int64_t TrSyms, int Niter, int os, double mu, const void *wx0, int ntaps, const int64_t *modes, int nsel,
                                           int adaptive, const void *symbols, int64_t nsy, int method, double *var, int32_t *best, void *wx_best);

/* ---- train_equaliser_realvalued: same layout with real arrays, update without conjugate ---------------------- */
int qh_train_equaliser_real_f32(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu, void *wx,
                                int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                                int method, void *err);
int qh_train_equaliser_real_f64(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu, void *wx,
                                int ntaps, const int64_t *modes, int nsel, int adaptive, const void *symbols, int64_t nsy,
                                int method, void *err);

/* ---- apply_filter_to_signal -------------------------------------------------------------------------------------
 *   out (nsel, N), N = (L - ntaps + 1) / os ;  out[j, i] = sum_k sum_t E[k, i*os + t] * wx[modes[j], k, t]
 */
int qh_apply_filter_c64(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                        int nsel, void *out);
int qh_apply_filter_c128(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                         int nsel, void *out);
int qh_apply_filter_f32(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                        int nsel, void *out);
int qh_apply_filter_f64(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                        int nsel, void *out);
int qh_apply_filter_c64_dev(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                            int nsel, void *out);
int qh_apply_filter_c128_dev(const void *E, int nmodes, int64_t L, int os, const void *wx, int ntaps, const int64_t *modes,
                             int nsel, void *out);

/* ---- bps: blind phase search index ---------------------------------------------------------------------------
 *   E (L,) one mode; testangles (p, A) real with p == 1 (one grid) or p == L (per-symbol grid); symbols (M,)
 *   idx (L,) int32 out: first arg-min over the A angles of the 2N-symbol windowed min-distance sum; 0 in [0,N) and [L-N,L)
 */
int qh_bps_c64(const void *E, int64_t L, const void *testangles, int64_t p, int A, const void *symbols, int M, int N,
               int32_t *idx);
int qh_bps_c128(const void *E, int64_t L, const void *testangles, int64_t p, int A, const void *symbols, int M, int N,
                int32_t *idx);
int qh_bps_c64_dev(const void *E, int64_t L, const void *testangles, int64_t p, int A, const void *symbols, int M, int N,
                   int32_t *idx);
int qh_bps_c128_dev(const void *E, int64_t L, const void *testangles, int64_t p, int A, const void *symbols, int M, int N,
                    int32_t *idx);

/* device-resident carrier recovery of the host layer qampy/core/phaserecovery.py:145-159 for `nm` modes at once:
 *   idx = bps(E[m], grid); ph = grid[idx]; ph[N:-N] = unwrap(4 ph)/4; Eout = E * exp(1j ph).
 * angles: the grid (A,) real in HBM, as the host builds it (np.linspace(-pi/4, pi/4, A, endpoint=False) in double, cast to
 * the signal's precision), or NULL for a grid formed on the device.  np.unwrap's jump decisions are evaluated with numpy's own
 * operations on the grid values (a jump of exactly half the range depends on their rounding), the running correction is an
 * exact integer prefix sum.  E, Eout (nm, L) complex; ph (nm, L) real; idx (nm, L) int32 scratch/out. */
int qh_bps_recover_c64_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *symbols, int M, int N, int32_t *idx,
                           void *ph, void *Eout);
int qh_bps_recover_c128_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *symbols, int M, int N, int32_t *idx,
                            void *ph, void *Eout);
/* the same in `nparts` calls (ABI 9): parts 0 .. nparts - 2 search a run of the signal each, part nparts - 1 searches the rest, unwraps and de-rotates.
 * A receiver that processes capture after capture enqueues one part beside each relaxation pass of the NEXT capture's training (qh_pit_opts.on_pass):
 * a part of an eighth of the search is one wave per SIMD: it starts with a pass's trainer and lives on the issue slots that pass leaves, where the whole search at once slows a pass by a third.
 * Same kernels on the same data: the results do not depend on nparts.  The parts of one search are issued by one thread, in order, on one stream,
 * with no other phase search of that thread in between. */
int qh_bps_recover_part_c64_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *symbols, int M, int N, int32_t *idx,
                                void *ph, void *Eout, int part, int nparts);
int qh_bps_recover_part_c128_dev(const void *E, int nm, int64_t L, const void *angles, int A, const void *symbols, int M, int N, int32_t *idx,
                                 void *ph, void *Eout, int part, int nparts);

/* ---- comp_freq_offset (pilot receiver; qampy/core/phaserecovery.py:435-473): out[k, n] = E[k, n] exp(-2 pi i (n + 1) fo[k] / os),
 * fo (nmodes,) in units of the symbol rate, E / out (nmodes, L) host arrays */
int qh_comp_freq_offset_c64(const void *E, int nmodes, int64_t L, const double *fo, int os, void *out);
int qh_comp_freq_offset_c128(const void *E, int nmodes, int64_t L, const double *fo, int os, void *out);
/* Tail of pilot_based_cpe_new (qampy/core/pilotbased_receiver.py:318-327): the averaged pilot phases kph (nmodes, nk) at the symbol
 * positions knots (nk, increasing) interpolated linearly to every symbol (np.interp) and taken out: out = E exp(-1j trace); trace in the
 * signal's complex dtype like the reference returns it.  Host arrays. */
int qh_pilot_phase_trace_c64(const void *E, int nmodes, int64_t L, const int64_t *knots, const double *kph, int nk, void *out, void *trace);
int qh_pilot_phase_trace_c128(const void *E, int nmodes, int64_t L, const int64_t *knots, const double *kph, int nk, void *out, void *trace);

/* ---- select_angles: out[i] = angles[(p > 1 ? i : 0), idx[i]] ;  idx int64 (L,) ------------------------------- */
int qh_select_angles_f32(const void *angles, int64_t p, int A, const int64_t *idx, int64_t L, void *out);
int qh_select_angles_f64(const void *angles, int64_t p, int A, const int64_t *idx, int64_t L, void *out);

/* ---- make_decision: nearest alphabet point, |distance| (not squared) and index (first minimum) --------------- */
int qh_make_decision_c64(const void *E, int64_t L, const void *symbols, int M, void *det, void *dist, int32_t *idx);
int qh_make_decision_c128(const void *E, int64_t L, const void *symbols, int M, void *det, void *dist, int32_t *idx);
int qh_make_decision_c64_dev(const void *E, int64_t L, const void *symbols, int M, void *det, void *dist, int32_t *idx);
int qh_make_decision_c128_dev(const void *E, int64_t L, const void *symbols, int M, void *det, void *dist, int32_t *idx);

/* ---- measurement helpers ---------------------------------------------------------------------------------------
 * symbol errors of decided indices against a reference index sequence with a lag: count(idx_rx[i] != idx_tx[i - lag]) */
int qh_count_errors_dev(const int32_t *idx_rx, const int32_t *idx_tx, int64_t n, int64_t lag, int64_t ntx,
                        unsigned long long *count_dev);

/* Form of the exact trainer for subsequent calls: 0 automatic (default), 1 direct, 2 look-ahead, 3 block-iterative (the same
 * switch as qh_set_form("trainer", ...)).  All forms give the
 * reference's results up to the order of floating-point additions; a form that cannot take a call falls through. */
int qh_set_trainer(int form);
/* Production knobs (ABI 8; rounds 1-4 read environment variables in the launch paths).  The environment variables of the same meaning
 * (QAMPY_HIP_RESERVED_CUS, QAMPY_HIP_GRAM_BUDGET_GB) are read ONCE, as the initial value, when the value is first needed.
 *   qh_set_reserved_cus(n)     compute units library stream 2 stays off (default 32; 0: none); applies to streams created afterwards - call it before a
 *                              thread's first library call, or after qh_thread_release
 *   qh_set_gram_budget_gb(gb)  scratch the Gram tables of one call may take (default 160 of the 288 GB); longer captures / larger banks train in time chunks
 *   qh_set_default_tier(tier, tol)
 *                              what the DROP-IN host-array trainers (qh_train_equaliser_c64 / _c128 - the entry points a binding of the reference's
 *                              `train_equaliser` export calls) run: 0 = tier a, the exact sequential recurrence (default); 1 = tier b, the
 *                              same recurrence solved in parallel in time to `tol` (0 = 1e-3; SURVEY.md 8c's complex64 bar is 1e-4), total: a call no
 *                              parallel-in-time solver exists for, or one the passes do not certify, runs in the exact form inside the call.
 *                              qh_last_pit_report: the device's report of the calling thread's most recent such solve. */
int qh_set_pit_timing(int mode);      /* which relaxation passes of a tier-b sweep get HIP events (qh_pit_last_timing): 0 none, 1 pass 1 of every sweep (default: an event
                                        * idles the stream for ~5.6 us), 2 every pass (bench.py's roofline run) */
int qh_set_reserved_cus(int n);
int qh_set_gram_budget_gb(double gb);
int qh_get_gram_budget_gb(double *gb);
int qh_set_default_tier(int tier, double tol);
int qh_get_default_tier(int *tier, double *tol);
/* Test / measurement hooks (ABI 10; rounds 1-5 read environment variables in the launch paths): each forces a kernel form the automatic choice would
 * not take at that size, none selects a different algorithm, every one is exercised through this C ABI by the -m gpu tests named.  The launch paths read
 * ONE table of atomics; the environment variables of the same meaning (QAMPY_HIP_TRAINER, QAMPY_HIP_PIT_FORM, QAMPY_HIP_SEG_LANES, QAMPY_HIP_PIT_PROBE,
 * QAMPY_HIP_BPS, QAMPY_HIP_BPS_FUSED, QAMPY_HIP_PIT_XASIDE, QAMPY_HIP_LA_PROFILE) are read ONCE, when the library is loaded, as the table's initial values.
 *   key          values                                        what
 *   "trainer"    auto | direct | lookahead | iterative          form of the exact trainer, like qh_set_trainer (tests/test_gpu_parity.py)
 *   "pit_form"   auto | segment | block                         parallel in time: throughput / latency form of the passes (tests/test_gpu_pit.py)
 *   "seg_lanes"  0 | 8 | 16                                     throughput form: lanes per chain (tests/test_gpu_pit.py)
 *   "pit_probe"  0 | 1                                          complex64: the complex128 analysis of a pass (probe of the capture) (tests/test_gpu_pit.py)
 *   "bps"        auto | tile | lds | fused                      phase search: tile kernel for complex64 / streaming kernel with the LDS ring only (no
 *                                                               register-ring kernel) / search + unwrap + de-rotation in one kernel (tests/test_gpu_parity.py)
 *   "pit_xaside" 0 | 1                                          start taps into the eigenbasis beside the pass (measurement)
 *   "la_profile" 0 | 1                                          developer aid: cycle split of workgroup 0 of the block trainers on stderr
 * qh_set_form returns QH_ERR_ARG for an unknown key or value; a NULL or empty value resets the key to automatic. */
int qh_set_form(const char *key, const char *value);
int qh_get_form(const char *key, int *value);

/* ---- parallel-in-time training ("tier B": opt-in, NOT the reference's order of evaluation) -----------
 * The sweep of TrSyms steps is cut into S contiguous segments that are trained CONCURRENTLY with the exact kernels
 * (segments = the channels of a batch that happen to be adjacent in one capture) and made consistent by waveform
 * relaxation: in pass p segment s starts from the taps segment s-1 ended with in pass p-1 (segment 0 always from the start
 * taps).  The map is triangular in s, its only fixed point is the sequential recurrence, reached exactly after S passes;
 * in practice the LMS recursion forgets its start within a few 1/(mu lambda) steps and the passes stop as soon as the
 * boundary DEFECT - the relative rms difference, on a probe window of the capture, between the outputs of the taps a
 * segment started from and the taps its left neighbour ended with, modulo the symmetry of the error function (common
 * phase for cma / rde, quarter turns for the square-grid functions) - is below `tol` for every boundary.  Everything is
 * decided on the device (skip flags): the call only enqueues, nothing synchronises.
 *   acquire != 0 (cold start, e.g. centre-spike taps): a sequential "acquisition" first trains a prefix of the capture with
 *     a gear-shifted step size mu_acq = clamp(gear * mu, mu, acq_bound / (nmodes ntaps <|x|^2>)) in chunks until the mean
 *     squared error stops improving (or acq_max steps).  The acquired taps SEED the start taps of segments 1 .. S-1; segment 0
 *     starts from the caller's taps (start = 0, the default), so the fixed point of the passes is the reference's recurrence from
 *     the caller's taps - the cold-start trajectory itself.  start = 1 is the round-2 behaviour: segment 0 starts from the
 *     acquired taps too, i.e. the result is the recurrence WARM-STARTED from them (one or two passes fewer, taps of the weakly
 *     excited directions ~1e-2 away from the cold-start result).
 *   Stop rule (`tol`): with the coarse correction on, the passes stop when the ESTIMATED DEVIATION of the pass's trajectory from
 *     the sequential recurrence - relative rms deviation of the equaliser OUTPUT, worst segment - is below tol.  The estimate is
 *     the first-order solution of the error equation E[s+1] = F'_s E[s] + d[s+1] (d the boundary defects of the pass, F'_s the
 *     segment Jacobian replaced by its linearised model J), measured in output power: sum_k lambda_k |E~_k[s]|^2 / <|y|^2> in the
 *     eigenbasis of the input covariance - the same scan that forms the next correction; the rule takes its rms over all segments
 *     and modes (times dev_safety, default 1: measured against the exact path the estimate is within a factor 1.5 of the rms
 *     deviation of the error trace) and additionally holds the worst segment below 3 tol and the estimated relative deviation of
 *     the taps themselves (unweighted norm of the same vectors, rms over the segments: the weakly excited directions count fully)
 *     below 2 tol.  Measured against the exact path the FINAL taps sit at the worst-segment value of that estimate (deviation_taps_worst,
 *     up to 1.6 x the rms: deviations of the weakly excited directions accumulate along the sweep), i.e. within 3 tol.
 *     Without the correction (nmodes * ntaps > 128, or switched off) the round-2 rule applies: largest boundary
 *     defect x a segment-length factor below tol.
 *   phase_seed: for the phase-sensitive functions (mcma, mrde, sbd, mddma, dd) the pass-0 start taps of segment s are the
 *     start taps rotated by an unwrapped 4th-power phase estimate of their output at the head of the segment, so that
 *     every segment starts phase-locked however far the carrier has drifted (-1: by method, 0 off, 1 on).
 *   correction: plain relaxation hands information on by ONE segment per pass, which is fine for the tap directions the
 *     signal excites (they forget within a segment) but not for the weakly excited ones (out-of-band directions, time
 *     constants 1/(mu g lambda) of millions of steps), whose state depends on the whole history.  Between the passes
 *     the boundary defects d[s] are therefore propagated through the LINEARISED segment map J = exp(-mu g T Rc), Rc the
 *     input covariance <conj(x) x^T> and g the mean gain of the error function: D[s+1] = d[s+1] + J D[s] (in the
 *     eigenbasis of Rc: one scalar first-order recurrence per direction, run as a parallel scan), start taps += D.  With J = 0 this is plain relaxation; J only preconditions the
 *     iteration - at the fixed point all defects vanish and the result is the sequential recurrence either way.
 * Fixed step, or the adaptive step through qh_pit_opts.adaptive (one output mode per call, see there); no data-aided methods.
 * EVERY call returns the reference's result: a sweep that is not certified is redone in the exact form inside the call
 * (qh_pit_opts.exact_redo_off, qh_pit_report.converged = 2).  gram: table from qh_gram_build_*_dev for this (E, os, ntaps,
 * TrSyms), or NULL.  report_dev: device memory for one qh_pit_report (read it after qh_sync), or NULL. */
#define QH_PIT_MAXPASS 24
#define QH_PIT_MAXCHUNK 32
typedef struct qh_pit_opts {
    int32_t segments;       /* 0 = automatic: segments of about 0.2 / mu (warm) or 0.4 / mu (cold start) steps, qh_pit_auto_segments */
    int32_t max_passes;     /* 0 = 16 (at most QH_PIT_MAXPASS) */
    int32_t acquire;        /* 0 warm start, 1 cold start: gear-shifted sequential acquisition first */
    int32_t phase_seed;     /* -1 by method, 0 off, 1 on */
    double tol;             /* 0 = 1e-3: accepted estimated rms deviation of the equaliser output from the sequential recurrence, relative to the
                             * output rms (see "Stop rule"); without the coarse correction: largest boundary defect accepted */
    double gear;            /* 0 = 8 */
    double acq_bound;       /* 0 = 0.08 */
    double acq_plateau;     /* 0 = 0.8: a chunk whose mean |err|^2 exceeds this fraction of the previous one's ends the acquisition */
    int64_t acq_chunk;      /* 0 = automatic: 2 / mu_acq steps (mu_acq = the gear-shifted step size) rounded to the nearest power of two, 256 .. 4096 */
    int64_t acq_max;        /* 0 = two chunks (at most TrSyms / 2 steps) */
    int32_t correction;     /* -1 / 1: linearised coarse correction between the passes (see below), 0: plain relaxation */
    int32_t head_steps;     /* fixed step: > 0 - the first head_steps steps of every sweep run in the EXACT form, the segments cover the rest; 0: none, unless
                             * the passes stall on the start of the sweep (see head_auto_off) (ABI 6) */
    void *basis;            /* NULL, or the eigenbasis of this capture's input covariance from qh_pit_basis_*_dev (device memory) */
    double corr_beta;       /* extra damping of the well-excited directions in the coarse map, exp(-a (1 + beta a)); < 0: by method */
    /* One capture over several processes / GPUs (optional; every process holds the whole capture and makes the same call):
     * this process trains segments [seg_first, seg_first + seg_count) only (exchange == NULL: all of them; with an exchange
     * callback seg_count = 0 means "none": the process still takes part in every exchange); after the training
     * launch of every pass the library zeroes the end taps of the segments it does not own, synchronises its stream (unless
     * exchange_on_stream) and calls
     * exchange(exchange_user, taps, bytes) - the caller sums the buffers of all processes in place (an all-reduce: RCCL over
     * xGMI) and returns 0 - after which every process evaluates the (cheap) boundary defects and the coarse correction on
     * identical data and takes identical decisions.  Error traces are written for the owned segments only. */
    int32_t seg_first, seg_count;
    int (*exchange)(void *user, void *taps_dev, size_t bytes);
    void *exchange_user;
    int32_t start;          /* 0: segment 0 starts from the caller's taps (fixed point = the reference's recurrence); 1: from the acquired taps */
    int32_t exchange_on_stream; /* != 0: `exchange` only ENQUEUES the all-reduce on the library stream (RCCL with qh_stream_handle): the library
                             * does not synchronise around it and enqueues the next pass ahead as in the single-process case */
    double dev_safety;      /* 0 = 1: factor on the deviation estimate in the stop rule */
    int32_t adaptive;       /* != 0: the step size adapts (adapt_step); ONE output mode per call (nsel = 1), one sweep, complex64, cma / mcma / sbd /
                             * mddma; mu is in/out like in the exact entry points.  The first 16384 steps run in the exact form, the sweep is held to
                             * tol / 3 with up to 24 passes (damped corrections); one that is not certified is redone in the exact form
                             * (report: converged = 2) (ABI 4) */
    int32_t exact_redo_off; /* 0 (default): a sweep the passes do not certify (estimate above tol when they stop, pass budget used up, no coarse
                             * model) is redone in the EXACT form from the taps the call started with, inside the call - every call returns the
                             * reference's result; report: converged = 2.  != 0: the uncertified result stays (converged = 0) (ABI 5) */
    int32_t head_auto_off;  /* 0 (default): a sweep whose passes stall with the estimated deviation confined to the first quarter of the segments (a
                             * non-linear transient at the start of the stage: e.g. a decision-directed stage pulling in from taps locked to another
                             * carrier phase) is repeated with that stretch as an exact head - sequential there, parallel in time after it - before the
                             * whole call is given to the exact form; != 0: off (ABI 6) */
    int32_t acq_anneal;     /* acquisition: 0 (default) the second and later chunks run at half the step of the one before, never below 2 mu; > 0: never
                             * below acq_anneal x mu; < 0: every chunk at the gear-shifted step (rounds 2-3) (ABI 6) */
    double mu_hint;         /* fixed step: the caller's HOST copy of *mu (> 0), so that the call does not have to read it back; with it - and acq_chunk
                             * given when acquire != 0 - the call enqueues its whole prologue without a host synchronisation (0: read back) (ABI 7) */
    void *prepared;         /* cold start (acquire != 0): NULL, or the device buffer a qh_pit_prepare_*_dev call for THIS capture, these start taps, this step
                             * size and method filled: the acquisition has run already (beside the previous capture's training, on another stream) -
                             * the call adopts its taps and report fields instead of running it.  The caller orders the two (qh_stream_wait_event). (ABI 8) */
    void (*on_pass0)(void *user);   /* NULL, or a function the call invokes ONCE, on the calling thread, right after it has enqueued the first relaxation pass: the
                             * place to enqueue work for OTHER library streams (the previous capture's phase search, the next capture's preparation) whose
                             * launches would otherwise sit in front of this sweep's; it must leave the current library stream as it found it (ABI 8) */
    void *on_pass0_user;
    void (*on_pass)(void *user, int sweep, int pass);   /* NULL, or a function the call invokes on the calling thread right BEFORE it enqueues the trainer launch of
                             * relaxation pass `pass` of sweep `sweep`: the place to enqueue chip-wide work of other library streams behind an event recorded
                             * here - it then starts together with the trainer and runs beside it (qh_bps_recover_part_*_dev: a part of the phase search small
                             * enough to be one wave per SIMD costs the pass 0-3 %), not in the ~70 us of analysis behind the trainer, which are the critical
                             * path.  A pass enqueued ahead of the decision that ends the sweep still calls it (its launches do nothing).  It must leave the
                             * current library stream as it found it (ABI 9) */
    void *on_pass_user;
} qh_pit_opts;
typedef struct qh_pit_report {
    int32_t segments, passes, converged, acq_chunks;
    int64_t seg_len, acq_steps;
    double mu, mu_acq, power, tol;
    double defect[QH_PIT_MAXPASS];      /* largest boundary defect after pass p of the last sweep; -1 = pass not run */
    double acq_err[QH_PIT_MAXCHUNK];    /* mean |err|^2 of the acquisition chunks */
    double gain, out_power;             /* linearised error-function gain g and mean output power used by the correction */
    int32_t acq_done, done, diverged, corr_on;   /* device-side flags */
    double result_change[QH_PIT_MAXPASS]; /* how far the sweep's RESULT (end taps of the last segment) moved from pass p-1 to pass p, as the relative
                                            rms difference of the two outputs on the last 128 steps (modulo the error function's symmetry);
                                            pass 0: against the start taps; -1 = not run */
    double deviation[QH_PIT_MAXPASS];   /* estimated relative rms output deviation of pass p's trajectory from the sequential recurrence (worst
                                            segment, without the safety factor); -1 = not available (no coarse correction) */
    double deviation_rms[QH_PIT_MAXPASS]; /* the same estimate as an rms over all segments and modes */
    double deviation_taps[QH_PIT_MAXPASS]; /* estimated deviation of the TAPS from the sequential recurrence's: |D[s]| / |w|, rms over segments and modes */
    double deviation_taps_worst[QH_PIT_MAXPASS]; /* ... of the worst segment */
} qh_pit_report;
int qh_pit_auto_segments(int64_t TrSyms, double mu, int nsel, int cold, int *segments);   /* cold: the sweep starts from unconverged taps (acquire) */
/* Eigenbasis of the input covariance <conj(x) x^T> of the training windows of a capture, for the coarse correction: depends
 * on (E, os, ntaps, TrSyms) only, so one build serves every stage and sweep over the same capture.  basis: device memory of
 * qh_pit_basis_bytes(nmodes*ntaps) bytes.  nmodes*ntaps <= 128.  overlap != 0: built on the library's other stream while the
 * current stream goes on (the trainer waits for it before its first correction; qh_sync waits for both streams). */
int qh_pit_basis_bytes(int ntot, size_t *bytes);
int qh_pit_basis_c64_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void *basis, int overlap);
int qh_pit_basis_c128_dev(const void *E, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void *basis, int overlap);
/* kernel time (HIP events on the library stream) of the trainer launches of the most recent qh_train_equaliser_*_pit_dev call:
 * the timed relaxation passes in order (all sweeps) and the sum of the acquisition chunks.  An event idles the stream for ~5.6 us,
 * so by default ONE pass per sweep is timed (pass 1); qh_set_pit_timing(2) times every pass. */
int qh_pit_last_timing(float *pass_ms, int max_passes, int *npass, float *acq_ms);
/* The sequential part of a COLD sweep ahead of time (ABI 8): the gear-shifted acquisition of a capture depends on the capture, the start taps, the step
 * size and the error function only - not on anything the previous capture's training produces - so a receiver that is handed capture after capture runs
 * it for capture k + 1 on another library stream while capture k trains (pipeline.ResidentReceiver.run(prefetch=True)), and hands the result to the
 * training call through qh_pit_opts.prepared.  Same kernels on the same data as the acquisition inside the call: bit-identical results.
 * opts: as for the training call (segments, mu_hint and acq_chunk must be given - the call never synchronises); prep: device memory of
 * qh_pit_prepare_bytes(...) bytes.  Returns QH_ERR_ARG with "not preparable" for a sweep whose acquisition cannot run ahead (decision-directed or
 * adaptive stages, latency-form passes): the caller then simply does not pass `prepared`. */
int qh_pit_prepare_bytes(int nmodes, int ntaps, int64_t acq_steps, size_t elem_bytes, size_t *bytes);
int qh_pit_prepare_c64_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int os, const float *mu_dev, const void *wx0, int ntaps,
                           const int64_t *modes, int nsel, const void *symbols, int64_t nsy, int method, const qh_pit_opts *opts, void *prep, size_t prep_bytes);
int qh_last_pit_report(qh_pit_report *out);      /* host copy of the report of the calling thread's most recent host-array solve through tier b (qh_set_default_tier) */
int qh_train_equaliser_c64_pit_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, float *mu_dev,
                                   void *wx, int ntaps, const int64_t *modes, int nsel, const void *symbols, int64_t nsy,
                                   int method, void *err, int zero_err, const void *gram, const qh_pit_opts *opts,
                                   void *report_dev);
int qh_train_equaliser_c128_pit_dev(const void *E, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os, double *mu_dev,
                                    void *wx, int ntaps, const int64_t *modes, int nsel, const void *symbols, int64_t nsy,
                                    int method, void *err, int zero_err, const void *gram, const qh_pit_opts *opts,
                                    void *report_dev);

/* ---- channel bank: nch independent captures of identical shape processed together -------------------------------
 * (SURVEY.md 8e "within a GPU": one exact training chain occupies one workgroup, so a GPU holds hundreds of channels).
 * All arrays carry a leading channel dimension and are contiguous: E (nch, nmodes, L), wx (nch, nmodes, nmodes, ntaps),
 * err (nch, nmodes, TrSyms*Niter), mu_dev (nch,), gram = nch tables from qh_gram_build_*_batch_dev (or NULL: built
 * internally); symbols and modes are shared.  Semantics per channel: exactly qh_train_equaliser_*_dev. */
int qh_gram_build_c64_batch_dev(const void *E, int nch, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram);
int qh_gram_build_c128_batch_dev(const void *E, int nch, int nmodes, int64_t L, int os, int ntaps, int64_t TrSyms, void **gram);
int qh_train_equaliser_c64_batch_dev(const void *E, int nch, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os,
                                     float *mu_dev, void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive,
                                     const void *symbols, int64_t nsy, int method, void *err, int zero_err, const void *gram);
int qh_train_equaliser_c128_batch_dev(const void *E, int nch, int nmodes, int64_t L, int64_t TrSyms, int Niter, int os,
                                      double *mu_dev, void *wx, int ntaps, const int64_t *modes, int nsel, int adaptive,
                                      const void *symbols, int64_t nsy, int method, void *err, int zero_err, const void *gram);

/* On-device SER harness (reference: cal_ser core/signals.py:295-335 = sync_and_adjust /
 * find_sequence_offset_complex core/ber_functions.py:33-160 + make_decision + compare).  E: one recovered row (N,) in
 * HBM; idx_tx: decided indices of the transmitted symbols (nmodes, ntx) int32 in HBM (qh_make_decision_*_dev); symbols
 * (M,) in HBM.  Bounded search over tx mode x quadrant rotation x lag in [-maxlag, maxlag] on a `window` of decided
 * symbols from the middle of the run, then errors over [trim, N - trim).  result (HOST, 7 x int64): errors, compared,
 * tx mode, rotation k (row was multiplied by j^k), lag (rx[i] <-> tx[i - lag]), matches in the window, window length. */
int qh_ser_c64_dev(const void *E, int64_t N, const int32_t *idx_tx, int nmodes, int64_t ntx, const void *symbols, int M,
                   int maxlag, int64_t window, int64_t trim, int64_t *result);
int qh_ser_c128_dev(const void *E, int64_t N, const int32_t *idx_tx, int nmodes, int64_t ntx, const void *symbols, int M,
                    int maxlag, int64_t window, int64_t trim, int64_t *result);

/* On-device synthesis of an impaired capture (the reference builds its test signals with signals.py,
 * core/resample.py:73-126 and core/impairments.py:94-233): random Gray-labelled M-QAM symbols (Philox, keyed by seed /
 * mode / symbol index), root-raised-cosine shaping at `os` samples per symbol, optional Wiener phase noise (phase_var =
 * 2 pi linewidth / fs per sample), first-order PMD (theta, DGD in samples; 2 modes) and AWGN at snr_db (reference
 * convention sigma = 10^(-snr/20) sqrt(os)).  Outputs in HBM: E (nmodes, nsym*os) complex64, symbols (nmodes, nsym)
 * complex64, idx_tx (nmodes, nsym) int32 (index into alphabet).  alphabet (M,) complex64 in HBM. */
int qh_synth_capture_c64_dev(void *E, void *symbols, int32_t *idx_tx, const void *alphabet, int M, int nmodes, int64_t nsym,
                             int os, double beta, double snr_db, int have_snr, double theta, double dgd_samples, int have_pmd,
                             double phase_var, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
