// Shared device/host helpers of libqampy_hip (gfx950 only; wave64 is hard-coded throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/qampy_hip.h"

namespace qh {

// ---------------------------------------------------------------------------------------------- host-side state
extern thread_local hipStream_t g_stream;     // per host thread (api.hip)
extern int g_device;
void set_error(const std::string &s);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define QH_HIP(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return qh::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define QH_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) { qh::set_error(msg); return QH_ERR_ARG; } \
    } while (0)

// First statement of the kernels on a tier-b trainer's critical path (chains, eigen-solver, control kernels): issue priority over the
// waves of a streaming kernel of another stream that shares the SIMD (the phase search of the previous capture, pipeline.py
// run(overlap=True)).  Costs one scalar instruction; without a co-runner it changes nothing.
#define QH_WAVE_FIRST() __builtin_amdgcn_s_setprio(3)

int ensure_init();
hipStream_t side_stream();       // the library stream that is NOT the current one (work overlapped with the current stream)
hipStream_t helper_stream();     // a third stream for small launches beside both (the coarse model of a tier-b sweep)
// Test / measurement hooks (qh_set_form): every switch that forces a kernel form the automatic choice would not take at that size lives in ONE table of
// atomics, set through the C ABI; the launch paths read the table, never the environment.  (The environment variables of rounds 1-5 - QAMPY_HIP_TRAINER,
// QAMPY_HIP_PIT_FORM, ... - are read ONCE, when the library is loaded, as the table's initial values: scripts/ that export them keep working.)
enum FormKey { FORM_TRAINER = 0, FORM_PIT, FORM_SEG_LANES, FORM_PIT_PROBE, FORM_BPS, FORM_PIT_XASIDE, FORM_LA_PROFILE, FORM_COUNT };
int form(FormKey k);
const char *trainer_force();   // "" (automatic) or "direct" / "lookahead" / "iterative": qh_set_trainer(), else QAMPY_HIP_TRAINER
double gram_budget_gb();         // scratch the Gram tables of one call may take (qh_set_gram_budget_gb; default: QAMPY_HIP_GRAM_BUDGET_GB read once, else 160)
int pit_timing_mode();            // which relaxation passes of a tier-b sweep get HIP events: 0 none, 1 pass 1 (default), 2 all (qh_set_pit_timing)
int default_tier();               // 0: tier a (exact), 1: tier b - what the drop-in host-array trainers run (qh_set_default_tier)
double default_tier_tol();        // tolerance of the default tier b (qh_set_default_tier)
int scratch(int slot, size_t bytes, void **p);   // grow-only device scratch, slots 0..11

// Staging memory of the host-pointer entry points: power-of-two size classes kept in a small pool (api.hip) instead of a
// hipMalloc / hipFree pair per call - hipFree synchronises the device, and the pilot receiver makes dozens of small calls.
// The entry points synchronise their stream before they return, so a buffer is idle when it goes back to the pool.
int pool_alloc(size_t bytes, void **p, size_t *cap);
void pool_free(void *p, size_t cap);

// RAII device scratch used by the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    size_t n = 0, cap = 0;
    ~DevBuf() { if (p) pool_free(p, cap); }
    int alloc(size_t bytes) {
        n = bytes;
        return pool_alloc(bytes ? bytes : 1, &p, &cap);
    }
    int from_host(const void *h, size_t bytes) {
        int rc = alloc(bytes);
        if (rc) return rc;
        if (bytes) QH_HIP(hipMemcpyAsync(p, h, bytes, hipMemcpyHostToDevice, g_stream));
        return QH_OK;
    }
    int to_host(void *h, size_t bytes) const {
        if (bytes) QH_HIP(hipMemcpyAsync(h, p, bytes, hipMemcpyDeviceToHost, g_stream));
        return QH_OK;
    }
};

// ---------------------------------------------------------------------------------------------- complex pairs
template <typename R> struct Cx { R re, im; };

template <typename R> struct Cx2T;
template <> struct Cx2T<float> { using type = float2; };
template <> struct Cx2T<double> { using type = double2; };

template <typename R> __device__ __forceinline__ Cx<R> ldg(const Cx<R> *p)
{
    using V = typename Cx2T<R>::type;
    V v = *reinterpret_cast<const V *>(p);
    return Cx<R>{v.x, v.y};
}
template <typename R> __device__ __forceinline__ void stg(Cx<R> *p, Cx<R> v)
{
    using V = typename Cx2T<R>::type;
    V o; o.x = v.re; o.y = v.im;
    *reinterpret_cast<V *>(p) = o;
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float abs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double abs_(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return __builtin_fmin(a, b); }

// ---------------------------------------------------------------------------------------------- wave64 cross-lane
// DPP controls (gfx9 encoding)
constexpr int DPP_QUAD_1032 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_mov(float v)
{
    // lanes whose row is masked off keep `old` (= 0): adding it is a no-op, exactly what the row_bcast steps need
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ double dpp_mov(double v)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ float readlane(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double readlane(double v, int lane)
{
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// sum over the 64 lanes of a wave; the total is returned wave-uniformly (read back from lane 63)
template <typename R> __device__ __forceinline__ R wave_sum(R v)
{
    v += dpp_mov<DPP_QUAD_1032>(v);
    v += dpp_mov<DPP_QUAD_2301>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    v += dpp_mov<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_mov<DPP_ROW_BCAST31, 0xC>(v);
    return readlane(v, 63);
}
// two sums with their DPP chains interleaved (re / im of a complex dot product)
template <typename R> __device__ __forceinline__ void wave_sum2(R &a, R &b)
{
    a += dpp_mov<DPP_QUAD_1032>(a);            b += dpp_mov<DPP_QUAD_1032>(b);
    a += dpp_mov<DPP_QUAD_2301>(a);            b += dpp_mov<DPP_QUAD_2301>(b);
    a += dpp_mov<DPP_ROW_HALF_MIRROR>(a);      b += dpp_mov<DPP_ROW_HALF_MIRROR>(b);
    a += dpp_mov<DPP_ROW_MIRROR>(a);           b += dpp_mov<DPP_ROW_MIRROR>(b);
    a += dpp_mov<DPP_ROW_BCAST15, 0xA>(a);     b += dpp_mov<DPP_ROW_BCAST15, 0xA>(b);
    a += dpp_mov<DPP_ROW_BCAST31, 0xC>(a);     b += dpp_mov<DPP_ROW_BCAST31, 0xC>(b);
    a = readlane(a, 63);
    b = readlane(b, 63);
}
// min over the 64 lanes, returned wave-uniformly
template <typename R> __device__ __forceinline__ R wave_min(R v)
{
    // masked-off rows would read `old` = 0, so the bcast steps are written with explicit selects instead
    v = min_(v, dpp_mov<DPP_QUAD_1032>(v));
    v = min_(v, dpp_mov<DPP_QUAD_2301>(v));
    v = min_(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = min_(v, dpp_mov<DPP_ROW_MIRROR>(v));
    R r0 = readlane(v, 0), r1 = readlane(v, 16), r2 = readlane(v, 32), r3 = readlane(v, 48);
    return min_(min_(r0, r1), min_(r2, r3));
}

}  // namespace qh
