/*
 * ORACLE - TEST INFRASTRUCTURE ONLY (never imported, linked or executed by the product path).
 *
 * Plain-C restatement of the reference's compiled hot loops
 *   /root/reference/qampy/core/equalisation/pythran_equalisation.py  (train_equaliser, train_equaliser_realvalued,
 *                                                                    apply_filter_to_signal, make_decision)
 *   /root/reference/qampy/core/pythran_dsp.py                        (bps, select_angle_index, select_angles)
 * used (a) as the checker of the HIP path in tests/, __graft_entry__.smoke() and (b) as bench.py's `cpu_baseline`
 * ("port").  Parity pinned: oracle/README.md - validated against golden vectors captured from the imported
 * reference (tests/golden/*.npz, generator tests/golden/gen_golden.py) by tests/test_oracle_golden.py.
 *
 * Two builds from this one source (oracle/Makefile):
 *   libqampy_oracle.so       strict: -O2 -ffp-contract=off, no OpenMP, sequential semantics  -> the checker
 *   libqampy_oracle_fast.so  the reference's flag set (setup.py:24-33) + OpenMP placed as in the reference -> CPU baseline
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

/* method ids shared with the python wrappers (oracle/oracle.py) and, by value, with include/qampy_hip.h */
enum { QO_CMA = 0, QO_CMA2, QO_SGNCMA, QO_MCMA, QO_RDE, QO_MRDE, QO_SBD, QO_MDDMA, QO_DD, QO_SBD_DATA };
enum { QO_R_CMA = 0, QO_R_SGNCMA, QO_R_DD, QO_R_DD_DATA };

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define R float
#define FN(x) CAT(x, _f32)
#define HYPOT hypotf
#define FABS fabsf
#include "qampy_oracle_impl.h"
#undef R
#undef FN
#undef HYPOT
#undef FABS

#define R double
#define FN(x) CAT(x, _f64)
#define HYPOT hypot
#define FABS fabs
#include "qampy_oracle_impl.h"

int qo_openmp(void)
{
#ifdef QO_OPENMP
    return 1;
#else
    return 0;
#endif
}
