/* The C ABI of libqampy_hip without any Python: train a 2x2 T/2-spaced equaliser on a synthetic QPSK capture (CMA), apply it,
 * recover the carrier phase (blind phase search) and count symbol errors.  Build and run on an MI355X:
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -o c_abi_demo -Lqampy_amd -lqampy_hip -Wl,-rpath,$PWD/qampy_amd -lm && ./c_abi_demo
 * Entry points and their reference counterparts: include/qampy_hip.h. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "qampy_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != QH_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, qh_last_error()); return 1; } } while (0)

static unsigned rng_state = 12345u;
static float urand(void) { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 16777216.0f; }
static float nrand(void) { return sqrtf(-2.0f * logf(urand() + 1e-12f)) * cosf(6.2831853f * urand()); }

int main(void)
{
    enum { NMODES = 2, NSYM = 1 << 14, OS = 2, NTAPS = 11, A = 16, NBPS = 8 };
    const int64_t L = (int64_t)NSYM * OS;
    float *E = calloc((size_t)NMODES * L * 2, sizeof(float)), *tx = malloc((size_t)NMODES * NSYM * 2 * sizeof(float));
    /* QPSK symbols, held over both samples of a symbol (a crude 2 samples/symbol pulse), polarisation rotation, phase offset, noise */
    const float th = 0.4f, ph = 0.3f, s2 = 0.70710678f;
    for (int m = 0; m < NMODES; m++)
        for (int k = 0; k < NSYM; k++) { tx[((size_t)m * NSYM + k) * 2] = urand() < 0.5f ? -s2 : s2; tx[((size_t)m * NSYM + k) * 2 + 1] = urand() < 0.5f ? -s2 : s2; }
    for (int64_t n = 0; n < L; n++) {
        const float *a = tx + ((size_t)0 * NSYM + n / OS) * 2, *b = tx + ((size_t)1 * NSYM + n / OS) * 2;
        const float xr[2] = {cosf(th) * a[0] - sinf(th) * b[0], sinf(th) * a[0] + cosf(th) * b[0]};
        const float xi[2] = {cosf(th) * a[1] - sinf(th) * b[1], sinf(th) * a[1] + cosf(th) * b[1]};
        for (int m = 0; m < NMODES; m++) {
            E[((size_t)m * L + n) * 2] = cosf(ph) * xr[m] - sinf(ph) * xi[m] + 0.05f * nrand();
            E[((size_t)m * L + n) * 2 + 1] = sinf(ph) * xr[m] + cosf(ph) * xi[m] + 0.05f * nrand();
        }
    }
    int ndev = 0;
    CHECK(qh_device_count(&ndev));
    if (ndev == 0) { fprintf(stderr, "no gfx950 device: %s\n", "libqampy_hip has no CPU fallback"); return 2; }
    CHECK(qh_init(0));
    /* taps (nmodes, nmodes, ntaps) with a centre spike, CMA radius R = <|s|^4>/<|s|^2> = 1 for unit-power QPSK */
    float *wx = calloc((size_t)NMODES * NMODES * NTAPS * 2, sizeof(float));
    for (int m = 0; m < NMODES; m++) wx[(((size_t)m * NMODES + m) * NTAPS + NTAPS / 2) * 2] = 1.0f;
    float symbols[NMODES][2] = {{1.0f, 0.0f}, {1.0f, 0.0f}};
    const int64_t modes[NMODES] = {0, 1};
    const int64_t TrSyms = (L / OS / NTAPS - 1) * NTAPS;
    float *err = malloc((size_t)NMODES * TrSyms * 2 * 2 * sizeof(float));
    float mu = 2e-3f;
    CHECK(qh_train_equaliser_c64(E, NMODES, L, TrSyms, 2, OS, &mu, wx, NTAPS, modes, NMODES, 0, symbols, 1, QH_M_CMA, err));
    const int64_t N = (L - NTAPS + 1) / OS;
    float *eq = malloc((size_t)NMODES * N * 2 * sizeof(float));
    CHECK(qh_apply_filter_c64(E, NMODES, L, OS, wx, NTAPS, modes, NMODES, eq));
    /* blind phase search per mode against the QPSK alphabet, A test angles in [-pi/4, pi/4) */
    float alphabet[4][2] = {{s2, s2}, {s2, -s2}, {-s2, s2}, {-s2, -s2}}, angles[A];
    for (int j = 0; j < A; j++) angles[j] = -0.78539816f + 1.57079633f * j / A;
    int32_t *idx = malloc((size_t)N * sizeof(int32_t));
    long worst = 0;
    for (int m = 0; m < NMODES; m++) {
        CHECK(qh_bps_c64(eq + (size_t)m * N * 2, N, angles, 1, A, alphabet, 4, NBPS, idx));
        /* de-rotate, decide, and compare with both transmitted modes under the 4 quadrant rotations and small lags */
        long best = N;
        for (int t = 0; t < NMODES; t++)
            for (int q = 0; q < 4; q++)
                for (int lag = 0; lag < NTAPS; lag++) {
                    long e = 0;
                    for (int64_t i = 200; i < N - 200; i++) {
                        const float a0 = angles[idx[i]] + 1.57079633f * q, re = eq[((size_t)m * N + i) * 2], im = eq[((size_t)m * N + i) * 2 + 1];
                        const float r = cosf(a0) * re - sinf(a0) * im, s = sinf(a0) * re + cosf(a0) * im;
                        const float *ref = tx + ((size_t)t * NSYM + i + lag) * 2;
                        e += ((r > 0) != (ref[0] > 0)) || ((s > 0) != (ref[1] > 0));
                    }
                    if (e < best) best = e;
                }
        printf("mode %d: %ld symbol errors of %ld\n", m, best, (long)(N - 400));
        if (best > worst) worst = best;
    }
    printf("mu %.3g, centre taps |w00| %.3f |w11| %.3f\n", mu, hypotf(wx[(NTAPS / 2) * 2], wx[(NTAPS / 2) * 2 + 1]),
           hypotf(wx[((size_t)3 * NTAPS + NTAPS / 2) * 2], wx[((size_t)3 * NTAPS + NTAPS / 2) * 2 + 1]));
    return worst < (N - 400) / 100 ? 0 : 3;
}
