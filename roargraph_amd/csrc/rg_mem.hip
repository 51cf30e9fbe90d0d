// rg_mem.hip -- device memory for the large, randomly read buffers of an index (base rows, split rows, adjacency, visited
// tags, id logs), BALANCED OVER THE MEMORY CLASSES of the device.
//
// What was measured (round 4; scripts/exp/alloc_map*.hip, profiles/r04/alloc_map*.jsonl): the 288 GB of an MI355X fall into
// three classes of physical memory, met in runs of 4 ... 100+ GiB along the order in which a process is handed memory.
// K1's access mix -- random 768-byte rows plus random byte tests -- runs 10 % SLOWER when the rows and the tags lie in the
// same class than when they lie in two (5.97 vs 5.43 ms in the probe), and fastest when each of them is spread over all
// classes (5.23 ms); a row gather alone gains 4 % from being spread, byte tests alone 10 %.  "Conflict" is an equivalence
// relation with exactly three classes, each a third of the device (one pairwise matrix over 17 chunks: profiles/r04/).  A
// plain hipMalloc of 8 - 19 GiB comes out of one run, i.e. one class -- which one, relative to the buffer it is used with,
// is the coin that round 3's "two modes, 10 % apart" tossed.  No allocation flag changes this (contiguous, VMM, first-in-
// process: scripts/exp/alloc_place.hip); what does is WHERE the physical pages come from.
//
// So buffers of 2 GiB and more are built with the HIP virtual-memory API from granules of 1 GiB whose class is
// measured (a 1-ms probe kernel against one representative granule per known class: slow = same class), taken round robin
// over the classes.  A request that lacks granules of some class walks on, granule by granule, keeping everything it
// classifies in a pool that the buffers of one index open share; dev_trim hands the pool back.  Every step can fail softly:
// no VMM, no second class found, no memory to walk -- the buffer is then a plain hipMalloc, exactly what round 3 shipped.
// RG_BALANCED_ALLOC=0 turns the whole thing off.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "rg.h"
#include "rg_internal.h"
#include "rg_mem.h"

namespace rg {

namespace {

constexpr size_t kGranule = (size_t)1 << 30;
constexpr size_t kMinBalanced = (size_t)2 << 30;      // smaller buffers: hipMalloc
constexpr int kMaxClasses = 4;

// ---- JOURNAL OF ADDRESS-SPACE EVENTS + FAULT REPORT (round 6) ------------------------------------------------------------------
// The runtime answers a GPU page fault with one line ("Memory access fault by GPU ... on address 0x...") and abort().  Two bench
// runs of round 5 died that way and the address could not be attributed to anything.  Every range this file maps, caches, hands
// out again or unmaps is therefore noted in a ring (4096 events, no allocation, one atomic per event), and -- RG_FAULT_REPORT=<path>
// in the environment, or rg_mem_fault_report(path) -- a SIGABRT / SIGSEGV / SIGBUS handler writes the ring, the live and cached
// buffers and /proc/self/maps to <path> before the process dies: the fault address then names its buffer and that buffer's life
// (benchlib/fault.py reads the report).  Kinds: g = granule mapped at its pool address (1 GiB), u = granule unmapped
// there (moved into a buffer, or dropped: aux 1), B = balanced buffer mapped, C = freed into the cache (still mapped), H = handed
// out again from the cache, F = unmapped (freed, or released from the cache: aux 1), P = plain hipMalloc of a large request,
// f = hipFree of a pointer the pools do not know, A = an arena of address space reserved (round 6: every g / B address lies inside one).
struct JEvent { uint64_t t_us, va, bytes; char kind; int8_t device; uint16_t aux; };
constexpr uint32_t kJournal = 4096;
JEvent g_journal[kJournal];
std::atomic<uint64_t> g_jn{0};
const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
void jlog(char kind, const void *va, size_t bytes, int device, int aux = 0) {
    const uint64_t i = g_jn.fetch_add(1, std::memory_order_relaxed);
    JEvent &e = g_journal[i % kJournal];
    e.t_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - g_t0).count();
    e.va = (uint64_t)(uintptr_t)va; e.bytes = (uint64_t)bytes; e.kind = kind; e.device = (int8_t)device; e.aux = (uint16_t)aux;
}

__device__ __forceinline__ uint32_t mixu(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// the access mix of K1 in miniature: every wave reads 32 random 768-byte rows of granule A (eight passes of four in flight,
// one dword per lane and step) and makes 64 random byte tests + 32 byte marks in its own slice of granule B, `steps` times
__global__ void __launch_bounds__(64) rg_mem_probe_kernel(const float *__restrict__ rows, uint32_t nrows, uint8_t *tags, size_t slot_bytes,
                                                          uint32_t steps, uint32_t seed, float *out) {
    const int lane = threadIdx.x, g = lane >> 4, a = lane & 15;
    uint8_t *my = tags + (size_t)blockIdx.x * slot_bytes;
    float acc = 0.0f;
    uint32_t s = mixu(seed ^ (blockIdx.x * 0x9E3779B1u));
    for (uint32_t it = 0; it < steps; ++it) {
        s = mixu(s + it);
        const uint32_t t = mixu(s ^ (uint32_t)lane * 0x85EBCA6Bu);
        const size_t off = (size_t)(((uint64_t)t * (uint64_t)slot_bytes) >> 32);
        const uint8_t v = __hip_atomic_load(my + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane & 1) __hip_atomic_store(my + off, (uint8_t)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += (float)v;
        float r[8][12];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t rid = (uint32_t)(((uint64_t)mixu(s ^ (uint32_t)(p * 4 + g + 1) * 0xC2B2AE35u) * nrows) >> 32);
            const float *src = rows + (size_t)rid * 192 + a;
#pragma unroll
            for (int k = 0; k < 12; ++k) r[p][k] = src[16 * k];
        }
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int k = 0; k < 12; ++k) acc += r[p][k];
    }
    if (acc == 123.456f) out[0] = acc;
}

struct Granule {
    hipMemGenericAllocationHandle_t h{};
    void *va = nullptr;     // where it is mapped while it sits in the pool (probes read it there)
    int cls = -1;
};

struct Buffer {
    size_t bytes = 0;       // mapped size (a multiple of the granule)
    std::vector<hipMemGenericAllocationHandle_t> handles;
    int per_class[kMaxClasses] = {0, 0, 0, 0};
};
// a freed balanced buffer, still mapped: the next request of (about) its size takes it as it is
struct Cached {
    void *va = nullptr;
    Buffer buf;
};

struct Pool {
    std::mutex mu;
    bool tried = false, usable = false;
    hipMemAllocationProp prop{};
    std::vector<Granule> reps;                 // one mapped granule per known class, never handed out
    std::vector<Granule> spare[kMaxClasses];   // classified, mapped in the pool, free
    float t_same = 0.0f;                       // probe time of two granules of one class
    float *d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::map<void *, Buffer> live;
    // BUFFER CACHE (round 5).  A virtual range can be used once (see va_reserve below), so a process that opens and closes indexes
    // would go through address space without bound -- and pay the walk and its probes on every open.  A freed balanced buffer
    // therefore stays as it is, mapped, in this list (up to RG_MEM_CACHE_GIB, default 64 GiB per device), and a later request of
    // its size -- the same index opened again, the tags of the next context -- takes it whole: no new range, no probe, no remap.
    // The cache goes back to the device when a request cannot be served otherwise, and through rg_mem_release.
    std::vector<Cached> cache;
    size_t cached_bytes = 0;
    int want_classes = 3;                      // classes a buffer is spread over: three on MI355X; fewer once a full walk found fewer
    // statistics (rg_mem_stats / rg_mem_stats_ex)
    uint64_t n_buffers = 0, n_plain = 0, n_probes = 0, n_ballast = 0, n_cache_hits = 0;
    double probe_ms = 0.0;                     // wall time spent classifying granules
    size_t walked_epoch = 0;                   // bytes walked since the pool was last trimmed
};

Pool g_pool[16];
bool g_off = false, g_off_read = false, g_trace = false, g_trace2 = false;
thread_local bool t_last_plain = false;       // dev_last_plain(): the calling thread's last large request fell back to hipMalloc

// cap of the cache of freed buffers: RG_MEM_CACHE_GIB, default min(64 GiB, a quarter of the device) -- on a 64-GiB part the fixed
// 64 GiB of round 5 would have kept every closed index on the device
size_t cache_cap_bytes() {
    static const size_t cap = [] {
        const char *e = getenv("RG_MEM_CACHE_GIB");
        if (e) return (size_t)std::max(0, atoi(e)) << 30;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)256 << 30; }
        return std::min<size_t>((size_t)64 << 30, total_b / 4);
    }();
    return cap;
}

bool off() {
    if (!g_off_read) {
        const char *e = getenv("RG_BALANCED_ALLOC");
        g_off = e && atoi(e) == 0;
        g_trace = getenv("RG_TRACE_ALLOC") != nullptr;
        g_trace2 = g_trace && atoi(getenv("RG_TRACE_ALLOC")) >= 2;
        g_off_read = true;
    }
    return g_off;
}

// debugging knobs of the mapping path (round 4: stale translations after unmap were suspected): RG_MEM_NOVAFREE=1 never hands a
// virtual range back (no range is ever mapped twice), RG_MEM_FENCE=1 synchronises the device around every map / unmap
bool knob(const char *name) { const char *e = getenv(name); return e && atoi(e) != 0; }
void va_unmap(void *va, size_t bytes) {
    static const bool fence = knob("RG_MEM_FENCE");
    if (fence) (void)hipDeviceSynchronize();
    (void)hipMemUnmap(va, bytes);
    if (fence) (void)hipDeviceSynchronize();
}
// Virtual ranges are NEVER handed back to the runtime.  Measured (scripts/exp/mem_stress.py, profiles/r04/mem_stress_*.log; HIP
// 7.0.51831 on the GPU box): after hipMemAddressFree a later reservation can return the same range, and accesses through it
// then reach the physical memory of the OLD mapping -- whole granules of a buffer read back another buffer's data, or the GPU
// faults when that memory went back to the driver (4 of 4 stress runs; 0 of 4 when no range is ever freed; device
// synchronisation around every map / unmap changes nothing).  RG_MEM_VA: "leak" (default) = a range is used once;
// "reuse" = ranges of equal size are kept in a list of the library's own and mapped again (experiment);
// "free" = hipMemAddressFree (the behaviour that corrupts: only to reproduce it).
// ROUND 6: "arena" (the default now).  The fault report of the one GPU abort of round 6 (profiles/r06/fault_report_box4_pytest_abort.txt: a
// full -m gpu run hung for 207 s and was aborted while the allocator was classifying the 13th freshly mapped granule of a walk) shows what
// "leak" still allowed: hipMemAddressReserve hands out address ranges that plain hipMalloc'ed buffers occupied SECONDS earlier (a buffer
// freed at t = 56.69 s lies inside a granule reserved and mapped at t = 59.44 s) -- the runtime recycles addresses between hipFree and the
// virtual-memory API, and a range is "used once" only as far as this file's own reservations go.  Both faults on record (round 5's address was
// 76 pages into a granule-sized region; round 6's hang sits on a granule's first touch) are FIRST TOUCHES OF A FRESH MAPPING.  In the arena
// mode one large range (RG_MEM_ARENA_GIB, default 4096 GiB of address space, no memory) is reserved when a device's pool is first used, and
// every granule and buffer address is carved from it by a bump pointer with a 2-MiB guard gap: no address of a mapping this file makes
// was ever a hipMalloc'ed buffer's after that moment, and none is used twice.  A spent arena is followed by another.
// A/B on one box (scripts/r06/walk_stress.py: the walk in a loop, torch and plain allocations churned between rounds;
// profiles/r06/walk_stress_summary.txt): "leak" 5 of 5 runs died of `Memory access fault by GPU` within 2,100 granules, every time on a
// granule whose address is 1-GiB aligned (one reservation in 512 is; large plain allocations sit at such addresses too: the signature of a
// stale huge-page translation of the buffer that had the address before); arena 0 faults in 57,024 granules, ~110 of them GiB-aligned.
enum VaMode { kVaLeak, kVaReuse, kVaFree, kVaArena };
VaMode va_mode() {
    static const VaMode m = [] {
        const char *e = getenv("RG_MEM_VA");
        if (e && !strcmp(e, "reuse")) return kVaReuse;
        if (e && !strcmp(e, "free")) return kVaFree;
        if (e && !strcmp(e, "leak")) return kVaLeak;
        return kVaArena;
    }();
    return m;
}
std::mutex g_va_mu;
std::map<size_t, std::vector<void *>> g_va_spare;
uint64_t g_va_reserved = 0;        // bytes of virtual address space reserved so far (rg_mem_stats)
char *g_arena = nullptr;           // current arena: [g_arena, g_arena + g_arena_bytes), next free address g_arena + g_arena_used
size_t g_arena_bytes = 0, g_arena_used = 0;
void va_free(void *va, size_t bytes) {
    if (va_mode() == kVaFree) (void)hipMemAddressFree(va, bytes);
    else if (va_mode() == kVaReuse) { std::lock_guard<std::mutex> lk(g_va_mu); g_va_spare[bytes].push_back(va); }
}
bool va_reserve(void **va, size_t bytes) {
    if (va_mode() == kVaArena) {
        std::lock_guard<std::mutex> lk(g_va_mu);
        constexpr size_t kGuard = (size_t)2 << 20;
        const size_t need = (bytes + kGuard + kGuard - 1) / kGuard * kGuard;
        if (!g_arena || g_arena_used + need > g_arena_bytes) {
            static const size_t arena_bytes = [] {
                const char *e = getenv("RG_MEM_ARENA_GIB");
                return (size_t)std::max(64, e ? atoi(e) : 4096) << 30;
            }();
            void *a = nullptr;
            size_t got = 0;
            for (size_t sz = arena_bytes; sz >= need && sz >= ((size_t)64 << 30) && !got; sz /= 4) {      // (a smaller arena where the address space is short)
                if (g_va_reserved + sz > ((uint64_t)64 << 40)) continue;
                if (hipMemAddressReserve(&a, sz, kGranule, nullptr, 0) == hipSuccess) got = sz;
                else (void)hipGetLastError();
            }
            if (!got) return false;
            g_arena = (char *)a; g_arena_bytes = got; g_arena_used = 0;
            g_va_reserved += got;
            jlog('A', a, got, -1);
        }
        *va = g_arena + g_arena_used;
        g_arena_used += need;
        return true;
    }
    if (va_mode() == kVaReuse) {
        std::lock_guard<std::mutex> lk(g_va_mu);
        auto it = g_va_spare.find(bytes);
        if (it != g_va_spare.end() && !it->second.empty()) { *va = it->second.back(); it->second.pop_back(); return true; }
    }
    {   // ranges are used once: a process that has gone through 64 TiB of them (hundreds of index opens) gets plain allocations from then on
        std::lock_guard<std::mutex> lk(g_va_mu);
        if (va_mode() == kVaLeak && g_va_reserved + bytes > ((uint64_t)64 << 40)) return false;
    }
    if (hipMemAddressReserve(va, bytes, kGranule, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lk(g_va_mu);
    g_va_reserved += bytes;
    return true;
}

bool map_at(void *va, size_t bytes, hipMemGenericAllocationHandle_t h, int device) {
    static const bool fence = knob("RG_MEM_FENCE");
    if (fence) (void)hipDeviceSynchronize();
    if (hipMemMap(va, bytes, 0, h, 0) != hipSuccess) return false;
    hipMemAccessDesc d{};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = device;
    d.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, bytes, &d, 1) != hipSuccess) { va_unmap(va, bytes); return false; }
    if (fence) (void)hipDeviceSynchronize();
    return true;
}

bool new_granule(Pool &P, int device, Granule *g) {
    if (hipMemCreate(&g->h, kGranule, &P.prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (!va_reserve(&g->va, kGranule)) { (void)hipMemRelease(g->h); return false; }
    if (!map_at(g->va, kGranule, g->h, device)) { (void)hipGetLastError(); va_free(g->va, kGranule); (void)hipMemRelease(g->h); return false; }
    g->cls = -1;
    jlog('g', g->va, kGranule, device);
    return true;
}

void drop_granule(Granule &g) {
    if (g.va) { jlog('u', g.va, kGranule, -1, 1); va_unmap(g.va, kGranule); va_free(g.va, kGranule); }
    (void)hipMemRelease(g.h);
    g.va = nullptr;
}

// milliseconds of the probe with its rows in granule a and its tags in granule b (two timed launches behind a warm-up one)
float probe(Pool &P, const Granule &a, const Granule &b) {
    struct Timer { Pool &P; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Timer() { P.probe_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } timer{P};
    const uint32_t slots = 2048, steps = 64;
    const uint32_t nrows = (uint32_t)(kGranule / 768);
    const size_t slot_bytes = kGranule / slots;
    ++P.n_probes;
    hipLaunchKernelGGL(rg_mem_probe_kernel, dim3(slots), dim3(64), 0, 0, (const float *)a.va, nrows, (uint8_t *)b.va, slot_bytes, steps, 3u, P.d_out);
    float best = 1e30f;
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(P.e0, 0);
        hipLaunchKernelGGL(rg_mem_probe_kernel, dim3(slots), dim3(64), 0, 0, (const float *)a.va, nrows, (uint8_t *)b.va, slot_bytes, steps, 11u + r, P.d_out);
        (void)hipEventRecord(P.e1, 0);
        if (hipEventSynchronize(P.e1) != hipSuccess) return -1.0f;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, P.e0, P.e1) != hipSuccess) return -1.0f;
        best = std::min(best, ms);
    }
    return hipGetLastError() == hipSuccess ? best : -1.0f;
}

bool init_pool(Pool &P, int device) {
    if (P.tried) return P.usable;
    P.tried = true;
    P.prop.type = hipMemAllocationTypePinned;
    P.prop.location.type = hipMemLocationTypeDevice;
    P.prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &P.prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || kGranule % gran) {
        (void)hipGetLastError();
        return false;
    }
    if (hipMalloc(&P.d_out, 64) != hipSuccess || hipEventCreate(&P.e0) != hipSuccess || hipEventCreate(&P.e1) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    // calibration: three granules handed out one after the other; runs of a class are gigabytes long, so at least two of
    // them share a class, and the slowest of the three pairs is a same-class pair
    Granule g[3];
    int n = 0;
    for (; n < 3; ++n)
        if (!new_granule(P, device, &g[n])) break;
    if (n < 3) { for (int i = 0; i < n; ++i) drop_granule(g[i]); return false; }
    for (int i = 0; i < 3; ++i) (void)hipMemsetAsync(g[i].va, 0, kGranule, 0);
    const float t01 = probe(P, g[0], g[1]), t02 = probe(P, g[0], g[2]), t12 = probe(P, g[1], g[2]);
    if (t01 <= 0 || t02 <= 0 || t12 <= 0) { for (int i = 0; i < 3; ++i) drop_granule(g[i]); return false; }
    P.t_same = std::max(t01, std::max(t02, t12));
    g[0].cls = 0;
    P.reps.push_back(g[0]);
    // the two others join the pool through the ordinary classification below
    for (int i = 1; i < 3; ++i) {
        const float t = i == 1 ? t01 : t02;
        if (t >= 0.955f * P.t_same) { g[i].cls = 0; P.spare[0].push_back(g[i]); }
        else { g[i].cls = 1; if (P.reps.size() == 1) P.reps.push_back(g[i]); else P.spare[1].push_back(g[i]); }
    }
    if (g_trace) fprintf(stderr, "[rg_mem] device %d: probe of one class %.3f ms (pairs %.3f %.3f %.3f)\n", device, P.t_same, t01, t02, t12);
    P.usable = true;
    return true;
}

// class of a fresh granule: the known class whose representative it conflicts with (probe within 4.5 % of the same-class
// time), else a new class (it becomes that class's representative; *is_rep says so)
int classify(Pool &P, Granule &g, bool *is_rep) {
    *is_rep = false;
    (void)hipMemsetAsync(g.va, 0, kGranule, 0);
    int slowest = -1;
    float ts = 0.0f;
    for (size_t c = 0; c < P.reps.size(); ++c) {
        const float t = probe(P, P.reps[c], g);
        if (t <= 0) return -1;
        if (t >= 0.955f * P.t_same) return (int)c;
        if (t > ts) { ts = t; slowest = (int)c; }
    }
    if ((int)P.reps.size() < kMaxClasses - 1) { *is_rep = true; return (int)P.reps.size(); }
    return slowest;
}

// (P.mu held) everything the pool holds beyond the live buffers goes back to the device: spare granules, cached buffers
void release_spare(Pool &P) {
    for (int k = 0; k < kMaxClasses; ++k) {
        for (Granule &g : P.spare[k]) drop_granule(g);
        P.spare[k].clear();
    }
    P.walked_epoch = 0;
}
void release_cache(Pool &P) {
    if (P.cache.empty()) return;
    (void)hipDeviceSynchronize();
    for (Cached &c : P.cache) {
        if (g_trace) fprintf(stderr, "[rg_mem] %.2f GiB at %p: released from the cache\n", (double)c.buf.bytes / (1u << 30), c.va);
        jlog('F', c.va, c.buf.bytes, -1, 1);
        for (size_t i = 0; i < c.buf.handles.size(); ++i) {
            va_unmap((char *)c.va + i * kGranule, kGranule);
            (void)hipMemRelease(c.buf.handles[i]);
        }
        va_free(c.va, c.buf.bytes);
    }
    P.cache.clear();
    P.cached_bytes = 0;
}

}  // namespace

bool dev_last_plain() { return t_last_plain; }

rg_status dev_alloc(int device, size_t bytes, void **out) {
    *out = nullptr;
    if (bytes == 0) bytes = 16;
    t_last_plain = false;
    Pool *Pp = (bytes >= kMinBalanced && !off() && device >= 0 && device < 16) ? &g_pool[device] : nullptr;
    // the plain buffer.  Whatever the pool still holds goes back first (a request that the pool's own spare granules
    // made impossible must not fail with the memory sitting in the pool)
    auto plain = [&]() -> rg_status {
        if (Pp) { release_spare(*Pp); release_cache(*Pp); t_last_plain = true; }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return set_error(RG_ERR_OOM, std::string("hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e)); }
        if (bytes >= kMinBalanced) jlog('P', *out, bytes, device);
        return RG_OK;
    };
    if (!Pp) return plain();
    Pool &P = *Pp;
    std::lock_guard<std::mutex> lk(P.mu);
    if (!init_pool(P, device)) { ++P.n_plain; return plain(); }
    const size_t n = (bytes + kGranule - 1) / kGranule;
    {   // a cached buffer of this size (at most an eighth larger): taken as it is
        int best = -1;
        for (size_t i = 0; i < P.cache.size(); ++i) {
            const size_t cn = P.cache[i].buf.bytes / kGranule;
            if (cn >= n && cn <= n + std::max<size_t>(1, n / 8) && (best < 0 || cn < P.cache[(size_t)best].buf.bytes / kGranule)) best = (int)i;
        }
        if (best >= 0) {
            Cached c = P.cache[(size_t)best];
            P.cache.erase(P.cache.begin() + best);
            P.cached_bytes -= c.buf.bytes;
            P.live[c.va] = c.buf;
            ++P.n_buffers; ++P.n_cache_hits;
            jlog('H', c.va, c.buf.bytes, device);
            if (g_trace) fprintf(stderr, "[rg_mem] %.2f GiB at %p: taken from the cache (%d / %d / %d / %d granules of the classes)\n", (double)c.buf.bytes / (1u << 30), c.va,
                                 c.buf.per_class[0], c.buf.per_class[1], c.buf.per_class[2], c.buf.per_class[3]);
            *out = c.va;
            return RG_OK;
        }
    }
    // what the buffer should get of each class: an equal share of every class the walk can reach -- three on this part (fewer once a
    // whole walk has found fewer: a device or partition without the third class must not walk its budget on every request)
    const int want_classes = P.want_classes;
    const size_t share = (n + want_classes - 1) / want_classes;
    size_t free_b = 0, total_b = 0;
    // Walk: new granules, one after the other, each classified, ALL kept in the pool -- until every class has its share there, or
    // the pool has walked its budget since it was last trimmed.  The pool lives across the buffers of one index open (and is
    // handed back by dev_trim at its end), so the four or five buffers of an open share one walk.  Runs of a class are 4 ...
    // 100+ GiB long: a walk that stepped over stretches with ballast (first version) found a scarce class in some processes and
    // not in others (boxes 19, 22: tags spread 12 / 0 / 8 over the classes, 65.6 / 57.5 % at L_pq 1000 / 2000 against 67.3 / 59.3);
    // granule by granule nothing is skipped, at about 10 ms per GiB walked.
    auto have = [&](int c) { return c < kMaxClasses ? P.spare[c].size() : 0; };
    auto satisfied = [&]() {
        size_t tot = 0;
        int classes = 0;
        for (int c = 0; c < kMaxClasses; ++c) { tot += std::min(have(c), share); classes += have(c) >= share ? 1 : 0; }
        return classes >= want_classes && tot >= n;
    };
    // (the pool never holds more than half of what was free when the walk began: other processes may live on the device)
    (void)hipMemGetInfo(&free_b, &total_b);
    const size_t epoch_budget = std::min<size_t>((size_t)160 << 30, free_b / 2);
    size_t walked = 0;
    bool budget_spent = false;
    while (!satisfied()) {
        (void)hipMemGetInfo(&free_b, &total_b);
        size_t pooled = 0;
        for (int c = 0; c < kMaxClasses; ++c) pooled += have(c);
        if (free_b < ((size_t)16 << 30) && !P.cache.empty()) { release_cache(P); continue; }   // cached buffers are the first thing to give up
        if (free_b < ((size_t)16 << 30)) break;                                     // the device is nearly full (others may live on it): take what there is
        if (pooled >= n && (P.walked_epoch >= epoch_budget || pooled * kGranule >= epoch_budget)) { budget_spent = true; break; }   // a long walk did not find enough of some class: take what there is
        Granule g;
        if (!new_granule(P, device, &g)) break;
        walked += kGranule;
        P.walked_epoch += kGranule;
        bool is_rep = false;
        const int c = classify(P, g, &is_rep);
        if (g_trace2) fprintf(stderr, "[rg_mem]   granule at %p (free %.1f GiB, pooled %zu) -> class %d%s\n", g.va, (double)free_b / (1u << 30), pooled, c, is_rep ? " (new)" : "");
        if (c < 0) { drop_granule(g); break; }
        g.cls = c;
        if (is_rep) { P.reps.push_back(g); continue; }
        P.spare[c].push_back(g);
    }
    if (budget_spent) {   // the whole budget walked and some class never showed: aim for the classes there are from now on
        int found = 0;
        for (int c = 0; c < kMaxClasses; ++c) found += have(c) > 0 ? 1 : 0;
        if (found >= 1 && found < P.want_classes) {
            P.want_classes = found;
            if (g_trace) fprintf(stderr, "[rg_mem] device %d: a full walk found %d class(es): buffers are spread over those from now on\n", device, found);
        }
    }
    size_t pooled = 0;
    for (int c = 0; c < kMaxClasses; ++c) pooled += have(c);
    if (pooled < n) {     // not enough granules (memory): the plain buffer
        ++P.n_plain;
        return plain();
    }
    void *va = nullptr;
    if (!va_reserve(&va, n * kGranule)) { ++P.n_plain; return plain(); }
    Buffer buf;
    buf.bytes = n * kGranule;
    std::vector<Granule> taken;
    int c = 0;
    for (size_t i = 0; i < n; ++i) {
        int tries = 0;
        while (have(c) == 0 && tries < kMaxClasses) { c = (c + 1) % kMaxClasses; ++tries; }
        Granule g = P.spare[c].back();
        P.spare[c].pop_back();
        taken.push_back(g);
        buf.per_class[c]++;
        c = (c + 1) % kMaxClasses;
    }
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) {
        Granule &g = taken[i];
        jlog('u', g.va, kGranule, device, 0);
        va_unmap(g.va, kGranule);
        va_free(g.va, kGranule);
        g.va = nullptr;
        ok = map_at((char *)va + i * kGranule, kGranule, g.h, device);
        if (ok) buf.handles.push_back(g.h);
    }
    if (!ok) {
        (void)hipGetLastError();
        for (size_t i = 0; i < buf.handles.size(); ++i) va_unmap((char *)va + i * kGranule, kGranule);
        for (Granule &g : taken) {
            if (g.va) { va_unmap(g.va, kGranule); va_free(g.va, kGranule); g.va = nullptr; }   // (not moved yet: still mapped in the pool)
            (void)hipMemRelease(g.h);
        }
        va_free(va, n * kGranule);
        ++P.n_plain;
        return plain();
    }
    // (what is left in the pool serves the next buffer; dev_trim hands it back)
    if (g_trace)
        fprintf(stderr, "[rg_mem] %.2f GiB at %p: %d / %d / %d / %d granules of the classes, %zu classes known, walked %.1f GiB\n", (double)buf.bytes / (1u << 30), va,
                buf.per_class[0], buf.per_class[1], buf.per_class[2], buf.per_class[3], P.reps.size(), (double)walked / (1u << 30));
    P.live[va] = buf;
    ++P.n_buffers;
    jlog('B', va, buf.bytes, device);
    *out = va;
    return RG_OK;
}

void dev_free(void *p) {
    if (!p) return;
    for (int d = 0; d < 16; ++d) {
        Pool &P = g_pool[d];
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) continue;
        Buffer &b = it->second;
        if (P.cached_bytes + b.bytes <= cache_cap_bytes()) {     // kept mapped for the next request of its size
            (void)hipDeviceSynchronize();                        // (nothing queued reads it any more when it is handed out again)
            if (g_trace) fprintf(stderr, "[rg_mem] %.2f GiB at %p: freed into the cache\n", (double)b.bytes / (1u << 30), p);
            jlog('C', p, b.bytes, d);
            P.cache.push_back({p, b});
            P.cached_bytes += b.bytes;
            P.live.erase(it);
            return;
        }
        (void)hipDeviceSynchronize();
        if (g_trace) fprintf(stderr, "[rg_mem] %.2f GiB at %p: freed (unmapped)\n", (double)b.bytes / (1u << 30), p);
        jlog('F', p, b.bytes, d, 0);
        for (size_t i = 0; i < b.handles.size(); ++i) {
            va_unmap((char *)p + i * kGranule, kGranule);
            (void)hipMemRelease(b.handles[i]);
        }
        va_free(p, b.bytes);
        P.live.erase(it);
        return;
    }
    jlog('f', p, 0, -1);
    (void)hipFree(p);
}

// hipMalloc for the library's plain (small or uncached) buffers: a request the device refuses is tried once more after the cache of
// freed balanced buffers and the pool's spare granules went back to the device (memory the library itself keeps mapped
// must not make its own next request fail)
hipError_t dev_malloc_retry(void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16) return e;
    {
        Pool &P = g_pool[device];
        std::lock_guard<std::mutex> lk(P.mu);
        if (P.cache.empty() && P.walked_epoch == 0) return e;      // nothing of ours to give back
        release_spare(P);
        release_cache(P);
    }
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

rg_status upload_staged(void *d_dst, const void *h_src, size_t bytes) {
    constexpr size_t kChunk = (size_t)64 << 20;
    if (bytes <= kChunk) {
        hipError_t e = hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice);
        return e == hipSuccess ? RG_OK : set_error(RG_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t s = nullptr;
    bool ok = hipHostMalloc(&pin[0], kChunk) == hipSuccess && hipHostMalloc(&pin[1], kChunk) == hipSuccess &&
              hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
    hipError_t e = hipSuccess;
    if (ok) {
        size_t off = 0;
        for (int p = 0; off < bytes && e == hipSuccess; p ^= 1) {
            const size_t n = std::min(kChunk, bytes - off);
            if (off >= 2 * kChunk) e = hipEventSynchronize(ev[p]);       // the copy that last used this chunk is done
            if (e != hipSuccess) break;
            memcpy(pin[p], (const char *)h_src + off, n);
            e = hipMemcpyAsync((char *)d_dst + off, pin[p], n, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipEventRecord(ev[p], s);
            off += n;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    } else {
        (void)hipGetLastError();
        e = hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice);
    }
    if (s) (void)hipStreamDestroy(s);
    for (int p = 0; p < 2; ++p) { if (ev[p]) (void)hipEventDestroy(ev[p]); if (pin[p]) (void)hipHostFree(pin[p]); }
    return e == hipSuccess ? RG_OK : set_error(RG_ERR_DEVICE, std::string("staged upload: ") + hipGetErrorString(e));
}

void dev_trim(int device) {
    if (device < 0 || device >= 16) return;
    Pool &P = g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    release_spare(P);
}

}  // namespace rg

extern "C" rg_status rg_mem_release(int device) {
    if (device < 0 || device >= 16) return rg::set_error(RG_ERR_ARG, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return rg::set_error(RG_ERR_DEVICE, "cannot select the device");
    rg::Pool &P = rg::g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    rg::release_spare(P);
    rg::release_cache(P);
    return RG_OK;
}

// ---- fault report -------------------------------------------------------------------------------------------------------------
namespace rg {
namespace {
char g_report_path[512] = {0};
struct sigaction g_old_act[3];
const int g_sigs[3] = {SIGABRT, SIGSEGV, SIGBUS};
void wr(int fd, const char *s, size_t n) { while (n) { ssize_t k = write(fd, s, n); if (k <= 0) return; s += k; n -= (size_t)k; } }
// (snprintf and the unlocked look at the pools are not async-signal-safe by the letter; the process is about to die, and a report that
// is right nearly always is worth more than none)
void write_report(int fd, const char *why) {
    char line[256];
    int n = snprintf(line, sizeof line, "rg_mem fault report v1 (%s)\n", why);
    wr(fd, line, (size_t)n);
    const uint64_t jn = g_jn.load(std::memory_order_relaxed);
    const uint64_t t_now = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - g_t0).count();
    n = snprintf(line, sizeof line, "now_us %llu\njournal %llu events (the last %u are kept): t_us kind va bytes device aux\n", (unsigned long long)t_now,
                 (unsigned long long)jn, kJournal);
    wr(fd, line, (size_t)n);
    for (uint64_t i = jn > kJournal ? jn - kJournal : 0; i < jn; ++i) {
        const JEvent &e = g_journal[i % kJournal];
        n = snprintf(line, sizeof line, "J %llu %c 0x%llx %llu %d %u\n", (unsigned long long)e.t_us, e.kind, (unsigned long long)e.va, (unsigned long long)e.bytes,
                     (int)e.device, (unsigned)e.aux);
        wr(fd, line, (size_t)n);
    }
    for (int d = 0; d < 16; ++d) {
        Pool &P = g_pool[d];
        if (!P.tried) continue;
        const bool locked = P.mu.try_lock();
        for (auto &kv : P.live) { n = snprintf(line, sizeof line, "LIVE %d 0x%llx %llu\n", d, (unsigned long long)(uintptr_t)kv.first, (unsigned long long)kv.second.bytes); wr(fd, line, (size_t)n); }
        for (auto &c : P.cache) { n = snprintf(line, sizeof line, "CACHED %d 0x%llx %llu\n", d, (unsigned long long)(uintptr_t)c.va, (unsigned long long)c.buf.bytes); wr(fd, line, (size_t)n); }
        for (auto &g : P.reps) { n = snprintf(line, sizeof line, "REP %d 0x%llx %llu\n", d, (unsigned long long)(uintptr_t)g.va, (unsigned long long)kGranule); wr(fd, line, (size_t)n); }
        for (int k = 0; k < kMaxClasses; ++k)
            for (auto &g : P.spare[k]) { n = snprintf(line, sizeof line, "SPARE %d 0x%llx %llu\n", d, (unsigned long long)(uintptr_t)g.va, (unsigned long long)kGranule); wr(fd, line, (size_t)n); }
        if (locked) P.mu.unlock();
    }
    wr(fd, "MAPS\n", 5);
    const int mf = open("/proc/self/maps", O_RDONLY);
    if (mf >= 0) {
        char buf[4096];
        ssize_t k;
        while ((k = read(mf, buf, sizeof buf)) > 0) wr(fd, buf, (size_t)k);
        close(mf);
    }
    wr(fd, "END\n", 4);
}
void on_fatal(int sig, siginfo_t *si, void *uc) {
    static std::atomic<int> once{0};
    int idx = 0;
    for (int i = 0; i < 3; ++i) if (g_sigs[i] == sig) idx = i;
    if (once.fetch_add(1) == 0 && g_report_path[0]) {
        const int fd = open(g_report_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd >= 0) {
            write_report(fd, sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : "SIGBUS");
            close(fd);
        }
    }
    // the handler that was there before (Python's faulthandler, or the default: die of the signal)
    const struct sigaction &o = g_old_act[idx];
    if ((o.sa_flags & SA_SIGINFO) && o.sa_sigaction) { o.sa_sigaction(sig, si, uc); return; }
    if (!(o.sa_flags & SA_SIGINFO) && o.sa_handler != SIG_DFL && o.sa_handler != SIG_IGN && o.sa_handler) { o.sa_handler(sig); return; }
    signal(sig, SIG_DFL);
    raise(sig);
}
struct ReportFromEnv {
    ReportFromEnv() {
        const char *e = getenv("RG_FAULT_REPORT");
        if (e && *e) (void)rg_mem_fault_report(e);
    }
} g_report_from_env;
}  // namespace
}  // namespace rg

extern "C" rg_status rg_mem_fault_report(const char *path) {
    if (!path || !*path || strlen(path) >= sizeof rg::g_report_path) return rg::set_error(RG_ERR_ARG, "rg_mem_fault_report: bad path");
    const bool first = rg::g_report_path[0] == 0;
    strcpy(rg::g_report_path, path);
    if (first) {
        struct sigaction a;
        memset(&a, 0, sizeof a);
        a.sa_sigaction = rg::on_fatal;
        a.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigemptyset(&a.sa_mask);
        for (int i = 0; i < 3; ++i) (void)sigaction(rg::g_sigs[i], &a, &rg::g_old_act[i]);
    }
    return RG_OK;
}

extern "C" rg_status rg_mem_journal_dump(const char *path) {
    if (!path || !*path) return rg::set_error(RG_ERR_ARG, "rg_mem_journal_dump: bad path");
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return rg::set_error(RG_ERR_IO, std::string("cannot write ") + path);
    rg::write_report(fd, "on request");
    close(fd);
    return RG_OK;
}

extern "C" rg_status rg_mem_stats_ex(int device, uint64_t *vals, int nvals) {
    if (device < 0 || device >= 16 || !vals || nvals < 1) return rg::set_error(RG_ERR_ARG, "bad argument");
    rg::Pool &P = rg::g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    uint64_t va = 0;
    { std::lock_guard<std::mutex> lk2(rg::g_va_mu); va = rg::g_va_reserved; }
    uint64_t live = 0;
    for (auto &kv : P.live) live += kv.second.bytes;
    const uint64_t all[10] = {P.n_buffers, P.n_plain, (uint64_t)P.reps.size(), P.n_probes, (uint64_t)(P.probe_ms * 1000.0), va, P.n_cache_hits,
                              (uint64_t)P.cached_bytes, live, (uint64_t)P.want_classes};
    for (int i = 0; i < nvals && i < 10; ++i) vals[i] = all[i];
    return RG_OK;
}

extern "C" rg_status rg_mem_stats(int device, uint64_t *buffers, uint64_t *plain, uint32_t *classes, uint64_t *granules_per_class /* [4] */) {
    if (device < 0 || device >= 16) return rg::set_error(RG_ERR_ARG, "device index out of range");
    rg::Pool &P = rg::g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    if (buffers) *buffers = P.n_buffers;
    if (plain) *plain = P.n_plain;
    if (classes) *classes = (uint32_t)P.reps.size();
    if (granules_per_class) {
        for (int k = 0; k < 4; ++k) granules_per_class[k] = 0;
        for (auto &kv : P.live)
            for (int k = 0; k < 4; ++k) granules_per_class[k] += (uint64_t)kv.second.per_class[k];
    }
    return RG_OK;
}

// diagnostics (scripts/r06/walk_stress.py): the allocator's WALK in a loop -- `per_round` granules created, mapped at fresh addresses, zeroed and
// probed against the class representatives (classify), then all dropped (unmapped, released) -- `rounds` times.  This is the step both GPU
// faults on record sit in (the first touch of a freshly mapped granule); the caller churns plain allocations between calls so that the
// runtime has freed addresses to recycle.  *granules = how many were mapped and touched.
extern "C" rg_status rg_mem_walk_stress(int device, uint32_t per_round, uint32_t rounds, uint64_t *granules) {
    if (device < 0 || device >= 16 || !granules) return rg::set_error(RG_ERR_ARG, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return rg::set_error(RG_ERR_DEVICE, "cannot select the device");
    if (rg::off()) return rg::set_error(RG_ERR_ARG, "the balanced allocator is switched off");
    rg::Pool &P = rg::g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    if (!rg::init_pool(P, device)) return rg::set_error(RG_ERR_DEVICE, "the virtual-memory pool cannot be set up on this device");
    *granules = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        std::vector<rg::Granule> got;
        for (uint32_t i = 0; i < per_round; ++i) {
            rg::Granule g;
            if (!rg::new_granule(P, device, &g)) break;
            bool is_rep = false;
            const int c = rg::classify(P, g, &is_rep);
            ++*granules;
            if (c < 0) { rg::drop_granule(g); return rg::set_error(RG_ERR_DEVICE, "a probe failed"); }
            if (is_rep) { g.cls = c; P.reps.push_back(g); continue; }
            got.push_back(g);
        }
        for (rg::Granule &g : got) rg::drop_granule(g);
    }
    return RG_OK;
}

// diagnostics (scripts/exp/mem_stress.py): `rounds` times, buffers of the given sizes are allocated, every page of each is
// written and read back by a kernel, and all are freed again in a shuffled order.  Returns the number of mismatching words.
__global__ void rg_mem_fill_kernel(uint32_t *p, size_t words, uint32_t tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (uint32_t)i;
}
__global__ void rg_mem_check_kernel(const uint32_t *p, size_t words, uint32_t tag, unsigned long long *bad) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) mine += p[i] != (tag ^ (uint32_t)i) ? 1u : 0u;
    if (mine) atomicAdd(bad, mine);
}
extern "C" rg_status rg_mem_selftest(int device, const uint64_t *sizes, int nsizes, int rounds, uint64_t *mismatches) {
    if (!sizes || nsizes <= 0 || !mismatches) return rg::set_error(RG_ERR_ARG, "null argument");
    if (hipSetDevice(device) != hipSuccess) return rg::set_error(RG_ERR_DEVICE, "cannot select the device");
    unsigned long long *d_bad = nullptr;
    if (hipMalloc(&d_bad, 8) != hipSuccess) return rg::set_error(RG_ERR_OOM, "hipMalloc");
    (void)hipMemset(d_bad, 0, 8);
    rg_status st = RG_OK;
    for (int r = 0; r < rounds && st == RG_OK; ++r) {
        std::vector<void *> bufs((size_t)nsizes, nullptr);
        for (int i = 0; i < nsizes && st == RG_OK; ++i) {
            st = rg::dev_alloc(device, (size_t)sizes[i], &bufs[(size_t)i]);
            if (st != RG_OK) break;
            hipLaunchKernelGGL(rg_mem_fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)bufs[(size_t)i], (size_t)sizes[i] / 4, 0x5bd1e995u * (uint32_t)(r * 131 + i + 1));
        }
        for (int i = 0; i < nsizes && st == RG_OK; ++i)
            if (bufs[(size_t)i])
                hipLaunchKernelGGL(rg_mem_check_kernel, dim3(4096), dim3(256), 0, 0, (const uint32_t *)bufs[(size_t)i], (size_t)sizes[i] / 4,
                                   0x5bd1e995u * (uint32_t)(r * 131 + i + 1), d_bad);
        if (hipDeviceSynchronize() != hipSuccess) st = rg::set_error(RG_ERR_DEVICE, "a self-test kernel failed");
        for (int i = 0; i < nsizes; ++i) rg::dev_free(bufs[(size_t)((i * 7 + r) % nsizes)]), bufs[(size_t)((i * 7 + r) % nsizes)] = nullptr;
        rg::dev_trim(device);
    }
    unsigned long long h = 0;
    (void)hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    *mismatches = h;
    return st;
}
