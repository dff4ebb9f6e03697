// Runtime part of libqampy_hip: device selection, the library stream, device memory, events, error text.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>
#include <unordered_map>
#include <string.h>

namespace qh {

// Streams, scratch slots and the tier-b solver's events are PER HOST THREAD: a thread that calls into the library gets its own set on first
// use, so that several captures can be in flight on one GPU, each driven by its own thread (pipeline.py ReceiverGroup); within a thread
// everything is ordered as before.  Device memory, the staging pool and the trainer selection are per process.
thread_local hipStream_t g_stream = nullptr;          // the stream every entry point enqueues on: one of g_streams (qh_use_stream)
static thread_local hipStream_t g_streams[4] = {nullptr, nullptr, nullptr, nullptr};    // three the caller can switch between (qh_use_stream 0..2) + one for small helper launches
int g_device = -1;
static thread_local std::string g_err;
static std::mutex g_mu;

void set_error(const std::string &s) { g_err = s; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    g_err = std::string(hipGetErrorString(e)) + " in " + what + " (" + file + ":" + std::to_string(line) + ")";
    (void)hipGetLastError();
    return QH_ERR_HIP;
}

// Hardware queues (see qampy_amd/_lib.py load()): four streams per host thread need more than the runtime's default of 4 hardware queues as soon as a
// second thread drives a receiver.  Set when the library is loaded - before the process's first HIP call unless someone else made one - and never
// over a value the user chose.
static const int g_hw_queues_set = [] { return setenv("GPU_MAX_HW_QUEUES", "16", 0); }();

// Production knobs: set through the C ABI (qh_set_reserved_cus / qh_set_gram_budget_gb / qh_set_default_tier); the environment variables of
// the same name are read ONCE, as the initial value, never in a launch path.
static std::atomic<int> g_reserved_cus{-1};
static std::atomic<double> g_gram_budget_gb{-1.0};
static std::atomic<int> g_default_tier{0};
static std::atomic<double> g_default_tol{1e-3};
static int reserved_cus()
{
    int v = g_reserved_cus.load();
    if (v < 0) { const char *e = getenv("QAMPY_HIP_RESERVED_CUS"); v = e ? atoi(e) : 32; if (v < 0) v = 0; g_reserved_cus.store(v); }
    return v;
}
double gram_budget_gb()
{
    double v = g_gram_budget_gb.load();
    if (v < 0) { const char *e = getenv("QAMPY_HIP_GRAM_BUDGET_GB"); v = e ? atof(e) : 160.0; if (!(v > 0)) v = 160.0; g_gram_budget_gb.store(v); }
    return v;
}
static std::atomic<int> g_pit_timing{-1};
int pit_timing_mode()
{
    int v = g_pit_timing.load();
    if (v < 0) { const char *e = getenv("QAMPY_HIP_PIT_TIMING"); v = !e ? 1 : (e[0] == 'a' ? 2 : (e[0] == 'n' ? 0 : 1)); g_pit_timing.store(v); }
    return v;
}
int default_tier() { return g_default_tier.load(); }
double default_tier_tol() { return g_default_tol.load(); }

static int init_device(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_stream && device == g_device) return QH_OK;
    if (g_device >= 0 && device != g_device) {
        // One device per process: scratch buffers, streams, events and the kernels' per-device attributes belong to the device
        // the library was initialised on; run one process per GPU (bench.py --gpus N does).
        set_error("libqampy_hip is already initialised on device " + std::to_string(g_device) + ": one device per process");
        return QH_ERR_ARG;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible: libqampy_hip needs a gfx950 (MI355X) GPU, there is no CPU fallback");
        return QH_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) { set_error("device index out of range"); return QH_ERR_ARG; }
    hipDeviceProp_t prop;
    QH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", libqampy_hip is built for gfx950 only");
        return QH_ERR_NODEVICE;
    }
    QH_HIP(hipSetDevice(device));
    int prio_least = 0, prio_greatest = 0;
    QH_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    for (int i = 0; i < 4; i++) {
        if (g_streams[i]) { (void)hipStreamDestroy(g_streams[i]); g_streams[i] = nullptr; }
        // stream 2 is where a caller overlaps chip-wide streaming work (filter output -> phase search) with the latency-bound trainers
        // of the next capture on stream 0: lowest queue priority, so that the trainers' workgroups are dispatched first
        if (i == 2) {
            // ... and kept off the first 32 compute units (qh_set_reserved_cus): its single-wave workgroups fill every CU they may
            // use to the LDS limit, and a trainer's one-workgroup kernels that need most of a CU's LDS (eigen-solver: 107 KiB; acquisition)
            // would otherwise wait for the whole phase search to drain (measured at C3: eigen-solver 1.1 ms instead of 0.5 ms, step
            // 3.95 ms with 0..16 CUs kept free, 3.50 ms with 24..48)
            const int reserved = reserved_cus();
            const int ncu = prop.multiProcessorCount;
            if (reserved > 0 && reserved < ncu) {
                std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
                for (int c = reserved; c < ncu; c++) mask[c / 32] |= 1u << (c % 32);
                // (a runtime that refuses the mask - an environment-wide CU mask is in force, say - gets the plain low-priority stream below:
                // overlapped receivers then run slower, nothing else changes)
                if (hipExtStreamCreateWithCUMask(&g_streams[i], (uint32_t)mask.size(), mask.data()) == hipSuccess) continue;
                (void)hipGetLastError();
                g_streams[i] = nullptr;
            }
        }
        QH_HIP(hipStreamCreateWithPriority(&g_streams[i], hipStreamNonBlocking, i == 2 ? prio_least : prio_greatest));
    }
    g_stream = g_streams[0];
    g_device = device;
    return QH_OK;
}

hipStream_t side_stream() { return g_stream == g_streams[0] ? g_streams[1] : g_streams[0]; }
hipStream_t helper_stream() { return g_streams[3]; }

int ensure_init()
{
    if (g_stream) return QH_OK;
    return init_device(g_device >= 0 ? g_device : 0);       // first call of this thread: the process's device, this thread's streams
}

// grow-only scratch slots so that the resident pipeline never allocates (and never synchronises) inside a timed region
thread_local void *g_scratch[16] = {nullptr};
thread_local size_t g_scratch_n[16] = {0};
// ---- the table of form switches (common.h FormKey; qh_set_form / qh_set_trainer).  Values:
//   trainer    0 automatic, 1 direct, 2 lookahead, 3 iterative          pit_form   0 automatic, 1 segment (throughput form), 2 block (latency forms)
//   seg_lanes  0 automatic, 8, 16                                       pit_probe  1: complex64 takes the complex128 analysis of a pass
//   bps        0 automatic, 1 tile kernel for complex64, 2 streaming kernel with the LDS ring only, 3 search + unwrap + de-rotation fused, 4 plain (round 5's rows)
//   pit_xaside 1: start taps into the eigenbasis beside the pass         la_profile 1: cycle split of workgroup 0 of the block trainers
static std::atomic<int> g_form[FORM_COUNT];
struct FormName { const char *key, *env; };
static const FormName FORM_NAMES[FORM_COUNT] = {
    {"trainer", "QAMPY_HIP_TRAINER"}, {"pit_form", "QAMPY_HIP_PIT_FORM"}, {"seg_lanes", "QAMPY_HIP_SEG_LANES"}, {"pit_probe", "QAMPY_HIP_PIT_PROBE"},
    {"bps", "QAMPY_HIP_BPS"}, {"pit_xaside", "QAMPY_HIP_PIT_XASIDE"}, {"la_profile", "QAMPY_HIP_LA_PROFILE"}};
static int form_parse(int k, const char *v, int *out)
{
    if (!v || !v[0]) { *out = 0; return 0; }
    switch (k) {
    case FORM_TRAINER: *out = v[0] == 'd' ? 1 : (v[0] == 'l' ? 2 : (v[0] == 'i' ? 3 : (v[0] == 'a' || v[0] == '0' ? 0 : -1))); break;
    case FORM_PIT: *out = v[0] == 's' ? 1 : (v[0] == 'b' ? 2 : (v[0] == 'a' || v[0] == '0' ? 0 : -1)); break;
    case FORM_SEG_LANES: { const int n = atoi(v); *out = (n == 8 || n == 16 || n == 0) ? n : -1; break; }
    case FORM_BPS: *out = v[0] == 't' ? 1 : (v[0] == 'l' ? 2 : (v[0] == 'f' ? 3 : (v[0] == 'p' ? 4 : (v[0] == 'a' || v[0] == '0' ? 0 : -1)))); break;
    default: *out = atoi(v) != 0 ? 1 : 0; break;
    }
    return *out < 0 ? -1 : 0;
}
// the environment, ONCE, when the library is loaded (QAMPY_HIP_BPS_FUSED=1 was round 3's spelling of bps = fused)
static const int g_form_env_read = [] {
    for (int k = 0; k < FORM_COUNT; k++) {
        int v = 0;
        if (form_parse(k, getenv(FORM_NAMES[k].env), &v) == 0) g_form[k].store(v);
    }
    const char *f = getenv("QAMPY_HIP_BPS_FUSED");
    if (f && f[0] == '1' && g_form[FORM_BPS].load() == 0) g_form[FORM_BPS].store(3);
    return 0;
}();
int form(FormKey k) { return g_form[k].load(std::memory_order_relaxed); }
const char *trainer_force()
{
    switch (form(FORM_TRAINER)) {
    case 1: return "direct";
    case 2: return "lookahead";
    case 3: return "iterative";
    default: return "";
    }
}
// ---- pool of staging buffers (DevBuf): size classes 2^k bytes, at most POOL_KEEP idle buffers per class and POOL_BYTES in total
static constexpr int POOL_CLASSES = 48, POOL_KEEP = 8;
static constexpr size_t POOL_BYTES = (size_t)24 << 30;
static std::vector<void *> g_pool[POOL_CLASSES];
static size_t g_pool_bytes = 0;
static void pool_release();
static int pool_class(size_t bytes) { int k = 8; while (((size_t)1 << k) < bytes && k < POOL_CLASSES - 1) k++; return k; }
int pool_alloc(size_t bytes, void **p, size_t *cap)
{
    const int k = pool_class(bytes);
    *cap = (size_t)1 << k;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_pool[k].empty()) { *p = g_pool[k].back(); g_pool[k].pop_back(); g_pool_bytes -= *cap; return QH_OK; }
    }
    hipError_t e = hipMalloc(p, *cap);
    if (e == hipErrorOutOfMemory) {
        // the idle buffers of the pool (up to POOL_BYTES) are the first thing to give back before the caller sees an allocation fail
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        pool_release();
        e = hipMalloc(p, *cap);
    }
    QH_HIP(e);
    return QH_OK;
}
void pool_free(void *p, size_t cap)
{
    const int k = pool_class(cap);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((int)g_pool[k].size() < POOL_KEEP && g_pool_bytes + cap <= POOL_BYTES) { g_pool[k].push_back(p); g_pool_bytes += cap; return; }
    }
    (void)hipFree(p);
}
static std::unordered_map<void *, size_t> g_dlive;                 // pooled buffers handed out by qh_malloc: pointer -> capacity
static void pool_release()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &v : g_pool) { for (void *p : v) (void)hipFree(p); v.clear(); }
    g_pool_bytes = 0;
}

// ---- pinned host memory for results (size classes like the device pool): the mirrored host layers hand back ndarrays that VIEW these buffers, so
// a result crosses PCIe once, by DMA at the link rate, into memory whose pages exist already - a fresh pageable array costs a bounce copy inside the
// runtime plus a page fault per 4 KiB on first touch (measured round 4: 20 GB/s device -> host against 54 GB/s the other way)
static std::vector<void *> g_hpool[POOL_CLASSES];
static size_t g_hpool_bytes = 0;
static constexpr size_t HPOOL_BYTES = (size_t)8 << 30;
static std::vector<std::pair<void *, size_t>> g_hlive;            // buffers handed out: pointer -> capacity
int pinned_alloc(size_t bytes, void **p)
{
    const int k = pool_class(bytes ? bytes : 1);
    const size_t cap = (size_t)1 << k;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_hpool[k].empty()) { *p = g_hpool[k].back(); g_hpool[k].pop_back(); g_hpool_bytes -= cap; g_hlive.emplace_back(*p, cap); return QH_OK; }
    }
    QH_HIP(hipHostMalloc(p, cap, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(g_mu);
    g_hlive.emplace_back(*p, cap);
    return QH_OK;
}
int pinned_free(void *p)
{
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_hlive.size(); i++)
            if (g_hlive[i].first == p) { cap = g_hlive[i].second; g_hlive[i] = g_hlive.back(); g_hlive.pop_back(); break; }
        if (!cap) { set_error("qh_pinned_free: not a live buffer of qh_pinned_alloc"); return QH_ERR_ARG; }
        const int k = pool_class(cap);
        if ((int)g_hpool[k].size() < POOL_KEEP && g_hpool_bytes + cap <= HPOOL_BYTES) { g_hpool[k].push_back(p); g_hpool_bytes += cap; return QH_OK; }
    }
    (void)hipHostFree(p);
    return QH_OK;
}
static void hpool_release()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &v : g_hpool) { for (void *p : v) (void)hipHostFree(p); v.clear(); }
    g_hpool_bytes = 0;
}

int scratch(int slot, size_t bytes, void **p)
{
    if (bytes > g_scratch_n[slot]) {
        if (g_scratch[slot]) QH_HIP(hipFree(g_scratch[slot]));
        g_scratch[slot] = nullptr; g_scratch_n[slot] = 0;
        QH_HIP(hipMalloc(&g_scratch[slot], bytes));
        g_scratch_n[slot] = bytes;
    }
    *p = g_scratch[slot];
    return QH_OK;
}

}  // namespace qh

extern "C" {
int qh_set_trainer(int form)
{
    if (form < 0 || form > 3) { qh::set_error("qh_set_trainer: 0 automatic, 1 direct, 2 lookahead, 3 iterative"); return QH_ERR_ARG; }
    qh::g_form[qh::FORM_TRAINER].store(form);
    return QH_OK;
}
int qh_set_form(const char *key, const char *value)
{
    if (!key) { qh::set_error("qh_set_form: key"); return QH_ERR_ARG; }
    for (int k = 0; k < qh::FORM_COUNT; k++) {
        if (strcmp(key, qh::FORM_NAMES[k].key) != 0) continue;
        int v = 0;
        if (qh::form_parse(k, value, &v) != 0) { qh::set_error(std::string("qh_set_form: value '") + (value ? value : "") + "' is not one of key '" + key + "'"); return QH_ERR_ARG; }
        qh::g_form[k].store(v);
        return QH_OK;
    }
    qh::set_error(std::string("qh_set_form: unknown key '") + key + "'");
    return QH_ERR_ARG;
}
int qh_get_form(const char *key, int *value)
{
    for (int k = 0; key && k < qh::FORM_COUNT; k++)
        if (strcmp(key, qh::FORM_NAMES[k].key) == 0) { *value = qh::g_form[k].load(); return QH_OK; }
    qh::set_error("qh_get_form: unknown key");
    return QH_ERR_ARG;
}

int qh_set_reserved_cus(int n)
{
    if (n < 0) { qh::set_error("qh_set_reserved_cus: n >= 0"); return QH_ERR_ARG; }
    qh::g_reserved_cus.store(n);
    return QH_OK;
}
int qh_set_gram_budget_gb(double gb)
{
    if (!(gb > 0)) { qh::set_error("qh_set_gram_budget_gb: gb > 0"); return QH_ERR_ARG; }
    qh::g_gram_budget_gb.store(gb);
    return QH_OK;
}
int qh_set_pit_timing(int mode)
{
    if (mode < 0 || mode > 2) { qh::set_error("qh_set_pit_timing: 0 none, 1 the second pass of every sweep (default), 2 every pass"); return QH_ERR_ARG; }
    qh::g_pit_timing.store(mode);
    return QH_OK;
}
int qh_get_gram_budget_gb(double *gb) { *gb = qh::gram_budget_gb(); return QH_OK; }
int qh_set_default_tier(int tier, double tol)
{
    if (tier != 0 && tier != 1) { qh::set_error("qh_set_default_tier: 0 = tier a (exact sequential recurrence), 1 = tier b (parallel in time)"); return QH_ERR_ARG; }
    if (tol < 0 || tol > 0.1) { qh::set_error("qh_set_default_tier: 0 <= tol <= 0.1 (0 = the library default 1e-3)"); return QH_ERR_ARG; }
    qh::g_default_tier.store(tier);
    qh::g_default_tol.store(tol > 0 ? tol : 1e-3);
    return QH_OK;
}
int qh_get_default_tier(int *tier, double *tol) { *tier = qh::default_tier(); *tol = qh::default_tier_tol(); return QH_OK; }

int qh_abi_version(void) { return QH_ABI_VERSION; }

int qh_device_count(int *count)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return QH_OK;
}
int qh_init(int device) { return qh::init_device(device); }
int qh_device_name(char *buf, size_t n)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    hipDeviceProp_t prop;
    QH_HIP(hipGetDeviceProperties(&prop, qh::g_device));
    snprintf(buf, n, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return QH_OK;
}
const char *qh_last_error(void) { return qh::g_err.c_str(); }
int qh_sync(void)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    for (int i = 0; i < 4; i++) QH_HIP(hipStreamSynchronize(qh::g_streams[i]));
    return QH_OK;
}
int qh_release_scratch(void)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    for (int i = 0; i < 4; i++) QH_HIP(hipStreamSynchronize(qh::g_streams[i]));
    for (int i = 0; i < 16; i++) {
        if (qh::g_scratch[i]) QH_HIP(hipFree(qh::g_scratch[i]));
        qh::g_scratch[i] = nullptr; qh::g_scratch_n[i] = 0;
    }
    qh::pool_release();
    qh::hpool_release();
    return QH_OK;
}
int qh_thread_release(void)
{
    // the calling thread's streams and scratch buffers (a worker thread before it ends; the main thread before the process exits - a
    // stream created with a CU mask must not be left to the runtime's own teardown when a profiler is attached)
    if (!qh::g_stream) return QH_OK;
    for (int i = 0; i < 4; i++) if (qh::g_streams[i]) (void)hipStreamSynchronize(qh::g_streams[i]);
    for (int i = 0; i < 16; i++) {
        if (qh::g_scratch[i]) (void)hipFree(qh::g_scratch[i]);
        qh::g_scratch[i] = nullptr; qh::g_scratch_n[i] = 0;
    }
    for (int i = 0; i < 4; i++) if (qh::g_streams[i]) { (void)hipStreamDestroy(qh::g_streams[i]); qh::g_streams[i] = nullptr; }
    qh::g_stream = nullptr;
    return QH_OK;
}
int qh_use_stream(int idx)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (idx < 0 || idx > 2) { qh::set_error("qh_use_stream: the library has streams 0, 1 and 2"); return QH_ERR_ARG; }
    qh::g_stream = qh::g_streams[idx];
    return QH_OK;
}
int qh_stream_handle(void **stream)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    *stream = (void *)qh::g_stream;
    return QH_OK;
}
int qh_stream_wait_event(void *ev)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    QH_HIP(hipStreamWaitEvent(qh::g_stream, (hipEvent_t)ev, 0));
    return QH_OK;
}
// Device memory for the callers' arrays.  Requests up to 1 GiB come from the staging pool (size classes 2^k; a freed buffer goes back to it): the
// mirrored host layers allocate their capture, error traces and outputs per call, and hipMalloc / hipFree of 64 MiB buffers cost ~0.5 ms each
// (19 allocations = 10 ms of a 28 ms C3 call chain, measured round 5).  Larger requests (channel banks) go to hipMalloc directly.
static constexpr size_t QH_POOLED_MAX = (size_t)1 << 30;
int qh_malloc(void **dptr, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes <= QH_POOLED_MAX) {
        size_t cap = 0;
        if ((rc = qh::pool_alloc(bytes ? bytes : 1, dptr, &cap))) return rc;
        std::lock_guard<std::mutex> lk(qh::g_mu);
        qh::g_dlive[*dptr] = cap;
        return QH_OK;
    }
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        qh::pool_release();
        e = hipMalloc(dptr, bytes);
    }
    QH_HIP(e);
    return QH_OK;
}
int qh_free(void *dptr)
{
    if (!dptr) return QH_OK;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> lk(qh::g_mu);
        auto it = qh::g_dlive.find(dptr);
        if (it != qh::g_dlive.end()) { cap = it->second; qh::g_dlive.erase(it); }
    }
    if (cap) {
        // like hipFree: nothing on the device may still use the buffer when it becomes available again (any stream of any thread)
        QH_HIP(hipDeviceSynchronize());
        qh::pool_free(dptr, cap);
        return QH_OK;
    }
    QH_HIP(hipFree(dptr));
    return QH_OK;
}
int qh_memset(void *dptr, int value, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes) QH_HIP(hipMemsetAsync(dptr, value, bytes, qh::g_stream));
    return QH_OK;
}
int qh_memcpy_h2d(void *dptr, const void *hptr, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes) QH_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, qh::g_stream));
    QH_HIP(hipStreamSynchronize(qh::g_stream));
    return QH_OK;
}
int qh_memcpy_d2h(void *hptr, const void *dptr, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes) QH_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, qh::g_stream));
    QH_HIP(hipStreamSynchronize(qh::g_stream));
    return QH_OK;
}
/* asynchronous forms on the current library stream (the host buffer should be pinned: qh_pinned_alloc, else the runtime stages the copy) */
int qh_memcpy_h2d_async(void *dptr, const void *hptr, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes) QH_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, qh::g_stream));
    return QH_OK;
}
int qh_memcpy_d2h_async(void *hptr, const void *dptr, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes) QH_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, qh::g_stream));
    return QH_OK;
}
int qh_stream_sync(void)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    QH_HIP(hipStreamSynchronize(qh::g_stream));
    return QH_OK;
}
int qh_pinned_alloc(void **hptr, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    return qh::pinned_alloc(bytes, hptr);
}
int qh_pinned_free(void *hptr) { return hptr ? qh::pinned_free(hptr) : QH_OK; }
int qh_memcpy_d2d(void *dst, const void *src, size_t bytes)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    if (bytes) QH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, qh::g_stream));
    return QH_OK;
}
int qh_event_create(void **ev)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    hipEvent_t e;
    QH_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return QH_OK;
}
int qh_event_destroy(void *ev)
{
    if (ev) QH_HIP(hipEventDestroy((hipEvent_t)ev));
    return QH_OK;
}
int qh_event_record(void *ev)
{
    int rc = qh::ensure_init();
    if (rc) return rc;
    QH_HIP(hipEventRecord((hipEvent_t)ev, qh::g_stream));
    return QH_OK;
}
int qh_event_elapsed_ms(void *start, void *stop, float *ms)
{
    QH_HIP(hipEventSynchronize((hipEvent_t)stop));
    QH_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return QH_OK;
}

}  // extern "C"
