// Error strings, launch accounting and the per-launch HIP-event profiler of libsgv_hip.so.
#include "sgv_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
static std::atomic<int64_t> g_launches{0};

int sgv_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* sgv_last_error(void) { return g_err; }
extern "C" int sgv_version(void) { return SGV_VERSION; }
extern "C" int64_t sgv_launch_count(void) { return g_launches.load(); }

static std::atomic<int64_t> g_variants[SGV_V_COUNT];
static const char* const g_variant_names[SGV_V_COUNT] = {
#define SGV_V_NAME(name) #name,
    SGV_VARIANTS(SGV_V_NAME)
#undef SGV_V_NAME
};
static thread_local int g_scope_slot = -1;   // profiler record of the sgv_launch_scope that is open on this thread, if any
static void prof_note_variant(int v);
void sgv_note_variant(int v) {
    if (v < 0 || v >= SGV_V_COUNT) return;
    g_variants[v].fetch_add(1, std::memory_order_relaxed);
    prof_note_variant(v);
}
extern "C" int64_t sgv_variant_count(int32_t v) { return (v >= 0 && v < SGV_V_COUNT) ? g_variants[v].load() : -1; }
extern "C" const char* sgv_variant_name(int32_t v) { return (v >= 0 && v < SGV_V_COUNT) ? g_variant_names[v] : nullptr; }

// ---------------------------------------------------------------------------------------------
// Profiler: a fixed pool of event pairs; one record per launch while enabled.

struct prof_record {
    hipEvent_t start, stop;
    int family, variant;
    double bytes, flops;
    bool stamped;       // bracketed by device-timestamp kernels instead of events (a launch recorded while its stream was being captured)
};

static std::mutex g_prof_mu;
static std::vector<prof_record> g_prof_pool;
static std::atomic<int> g_prof_next{0};
static std::atomic<bool> g_prof_on{false};
static std::atomic<uint64_t> g_prof_mask{~0ull};     // bit f: launches of family f are bracketed
static unsigned long long* g_stamps = nullptr;       // device: [2 * pool size] constant-rate clock readings (record i: 2 i = before, 2 i + 1 = behind its kernel)
static size_t g_stamps_cap = 0;
static void prof_note_variant(int v) {
    const int i = g_scope_slot;
    if (i >= 0 && i < (int)g_prof_pool.size() && g_prof_pool[i].variant < 0) g_prof_pool[i].variant = v;   // the first note of a call names it
}

extern "C" int sgv_prof_enable(int32_t max_records) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (max_records <= 0) return sgv_fail(SGV_ERR_INVALID_ARG, "sgv_prof_enable: max_records must be > 0");
    while ((int)g_prof_pool.size() < max_records) {
        prof_record r{};
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess)
            return sgv_fail(SGV_ERR_LAUNCH, "sgv_prof_enable: hipEventCreate failed");
        g_prof_pool.push_back(r);
    }
    if (g_stamps_cap < g_prof_pool.size()) {
        // A table that captured graphs may still write to is never freed: a replay of a graph captured before the pool grew stores into the RETIRED table (16 bytes per
        // record, kept for the life of the process) -- its durations are no longer collected, nothing is corrupted.
        unsigned long long* fresh = nullptr;
        const size_t bytes = g_prof_pool.size() * 2 * sizeof(unsigned long long);
        if (hipMalloc((void**)&fresh, bytes) != hipSuccess || hipMemset(fresh, 0, bytes) != hipSuccess)
            return sgv_fail(SGV_ERR_LAUNCH, "sgv_prof_enable: no device memory for the timestamp table");
        g_stamps = fresh;       // (zeroed: a pair that no replay has written yet reads as "no duration")
        g_stamps_cap = g_prof_pool.size();
    }
    g_prof_next = 0;
    g_prof_on = true;
    return SGV_OK;
}

extern "C" int sgv_prof_disable(void) {
    g_prof_on = false;
    return SGV_OK;
}

extern "C" int sgv_prof_resume(void) {
    if (g_prof_pool.empty()) return sgv_fail(SGV_ERR_INVALID_ARG, "sgv_prof_resume: no pool (call sgv_prof_enable first)");
    g_prof_on = true;
    return SGV_OK;
}

extern "C" int sgv_prof_families(uint64_t mask) {
    g_prof_mask = mask ? mask : ~0ull;
    return SGV_OK;
}

// host copy of the timestamp pairs of the first n records (after the device has drained) + the clock's rate; empty if none of them is stamped
static bool prof_read_stamps(int n, std::vector<unsigned long long>& host, double& ticks_per_ms) {
    bool any = false;
    for (int i = 0; i < n; i++) any = any || g_prof_pool[i].stamped;
    if (!any || !g_stamps) return false;
    if (hipDeviceSynchronize() != hipSuccess) return false;
    host.resize((size_t)2 * n);
    if (hipMemcpy(host.data(), g_stamps, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return false;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    ticks_per_ms = (double)khz;
    return true;
}

extern "C" int sgv_prof_collect(sgv_prof_entry* out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!out) return sgv_fail(SGV_ERR_INVALID_ARG, "sgv_prof_collect: out is NULL");
    for (int k = 0; k < SGV_K_COUNT; k++) out[k] = sgv_prof_entry{0, 0.0, 0.0, 0.0};
    int n = g_prof_next.load();
    if (n > (int)g_prof_pool.size()) n = (int)g_prof_pool.size();
    std::vector<unsigned long long> stamps;
    double ticks_per_ms = 1e5;
    const bool have_stamps = prof_read_stamps(n, stamps, ticks_per_ms);
    for (int i = 0; i < n; i++) {
        prof_record& r = g_prof_pool[i];
        float ms = 0.f;
        if (r.stamped) {
            if (!have_stamps || stamps[2 * i + 1] <= stamps[2 * i]) continue;
            ms = (float)((double)(stamps[2 * i + 1] - stamps[2 * i]) / ticks_per_ms);
        } else {
            if (hipEventSynchronize(r.stop) != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "sgv_prof_collect: event sync failed");
            if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
        }
        sgv_prof_entry& e = out[r.family];
        e.launches += 1;
        e.ms += ms;
        e.bytes += r.bytes;
        e.flops += r.flops;
    }
    g_prof_next = 0;
    return SGV_OK;
}

extern "C" int sgv_prof_collect_records(sgv_prof_record* out, int32_t max_records) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!out || max_records < 0) return sgv_fail(SGV_ERR_INVALID_ARG, "sgv_prof_collect_records: bad output array");
    int n = g_prof_next.load();
    if (n > (int)g_prof_pool.size()) n = (int)g_prof_pool.size();
    std::vector<unsigned long long> stamps;
    double ticks_per_ms = 1e5;
    const bool have_stamps = prof_read_stamps(n, stamps, ticks_per_ms);
    for (int i = 0; i < n; i++) {
        prof_record& r = g_prof_pool[i];
        float ms = 0.f;
        if (r.stamped) {
            if (have_stamps && stamps[2 * i + 1] > stamps[2 * i]) ms = (float)((double)(stamps[2 * i + 1] - stamps[2 * i]) / ticks_per_ms);
        } else {
            if (hipEventSynchronize(r.stop) != hipSuccess) return sgv_fail(SGV_ERR_LAUNCH, "sgv_prof_collect_records: event sync failed");
            if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) ms = 0.f;
        }
        if (i < max_records) out[i] = sgv_prof_record{r.family, r.variant, ms, 0.f, r.bytes, r.flops};
    }
    g_prof_next = 0;
    return n;
}

static thread_local float* g_amax_sink = nullptr;     // armed by sgv_amax_sink(), moved into the next launch scope of this thread
static thread_local int g_amax_consumed = 0;

extern "C" int sgv_amax_sink(float* out) {
    g_amax_sink = out;
    g_amax_consumed = 0;
    return SGV_OK;
}
extern "C" int sgv_amax_sink_consumed(void) { return g_amax_consumed; }

float* sgv_launch_scope::take_amax_sink() {
    float* p = amax_sink;
    amax_sink = nullptr;
    if (!p) return nullptr;
    g_amax_consumed = 1;
    amax_taken = p;
    return p;
}

// sink[0] = max(sink[0], sink[1 .. SGV_AMAX_SLOTS]): one workgroup, behind the producing kernel on its stream
__global__ __launch_bounds__(256) void sgv_amax_reduce_kernel(unsigned* sink) {
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < SGV_AMAX_SLOTS / 256; i++) m = max(m, sink[1 + i * 256 + threadIdx.x]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(sink, m);
}

// Timing a launch INSIDE a captured graph.  An ordinary hipEventRecord during stream capture leaves no node whose time can be read back, and the external
// event-record form (hipEventRecordWithFlags(..., hipEventRecordExternal): works under the ROCm 7.2 runtime, tools/graph_event_lab.hip) is refused with "invalid
// argument" by the HIP 7.0 runtime that PyTorch 2.10 bundles and this library shares a process with (profiles/r05_c13_graph_event_repro.log).  So a launch that
// is recorded while its stream is being captured is bracketed by two one-thread KERNELS that store the device's constant-rate clock (wall_clock64, 100 MHz):
// ordinary kernel nodes, replayed with the graph in stream order; the difference of a pair read after a replay is the launch's duration inside that replay plus the
// dispatch gap in front of it (~1-2 us against launches of 30-2,000 us).  bench.py: the roofline objects of the captured headline step.
__global__ void sgv_stamp_kernel(unsigned long long* out) { *out = wall_clock64(); }

static bool prof_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

sgv_launch_scope::sgv_launch_scope(int family, hipStream_t s, double bytes, double flops, bool count, bool own_stamps)
    : slot(-1), stamp_mode(0), stream(s), amax_sink(g_amax_sink), amax_taken(nullptr) {
    g_amax_sink = nullptr;
    if (count) g_launches.fetch_add(1, std::memory_order_relaxed);
    if (!g_prof_on.load(std::memory_order_relaxed) || !((g_prof_mask.load(std::memory_order_relaxed) >> family) & 1ull)) return;
    int i = g_prof_next.fetch_add(1);
    if (i >= (int)g_prof_pool.size()) return;  // pool exhausted: launch is simply not recorded
    prof_record& r = g_prof_pool[i];
    r.family = family;
    r.variant = -1;
    r.bytes = bytes;
    r.flops = flops;
    slot = i;
    g_scope_slot = i;
    r.stamped = g_stamps != nullptr && prof_capturing(stream);
    if (r.stamped) {
        if (own_stamps) stamp_mode = 2;
        else { stamp_mode = 1; hipLaunchKernelGGL(sgv_stamp_kernel, dim3(1), dim3(1), 0, stream, g_stamps + 2 * (size_t)i); }
    } else (void)hipEventRecord(r.start, stream);
}

unsigned long long* sgv_launch_scope::kernel_stamps() {
    if (slot < 0 || stamp_mode != 2) return nullptr;
    stamp_mode = 3;
    return g_stamps + 2 * (size_t)slot;
}

void sgv_launch_scope::begin_stamp() {
    if (slot < 0 || stamp_mode != 2) return;
    stamp_mode = 1;
    hipLaunchKernelGGL(sgv_stamp_kernel, dim3(1), dim3(1), 0, stream, g_stamps + 2 * (size_t)slot);
}

sgv_launch_scope::~sgv_launch_scope() {
    if (slot >= 0) {
        if (stamp_mode == 1) hipLaunchKernelGGL(sgv_stamp_kernel, dim3(1), dim3(1), 0, stream, g_stamps + 2 * (size_t)slot + 1);
        else if (stamp_mode == 0 && !g_prof_pool[slot].stamped) (void)hipEventRecord(g_prof_pool[slot].stop, stream);
        // (mode 3: the kernel writes both; mode 2 left undecided: the pair stays as it was -- zero or stale -- and the record reads as "no duration")
        g_scope_slot = -1;
    }
    // the one-workgroup fold of the partial maxima: behind the producer on its stream, outside the call's event bracket (the bracket times the op's own
    // kernel for the roofline tables; the fold is ~3 us of a single workgroup)
    if (amax_taken) hipLaunchKernelGGL(sgv_amax_reduce_kernel, dim3(1), dim3(256), 0, stream, (unsigned*)amax_taken);
}
