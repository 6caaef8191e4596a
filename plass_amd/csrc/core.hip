// plasship: context + sequence DB residency (C-ABI part 1).  Product code.
//   replaces DBReader<unsigned int>::open/getData/getSeqLen/getDbKey for the hot modules
//   (mm/commons/DBReader.cpp:150-215,548-589; DBReader.h:185-213) and DBWriter for sequence DBs.
#include "common.hpp"
#include "host_util.hpp"
#include "device_utils.hpp"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <unordered_map>

namespace plasship {
// ---- device memory arena (see common.hpp) ---------------------------------------------------------------
// hipMalloc / hipFree synchronise the device and cost ~27 ms per GB on this stack; an assembly iteration on 50 M reads asks for
// ~40 buffers between a few bytes and 85 GB whose sizes change from iteration to iteration.  Memory therefore comes from SLABS
// obtained from HIP once and never returned while in use: a request takes the best-fitting free range of any slab and splits
// it, a release merges the range with its free neighbours.  (Round 1 cached whole blocks by size class: a 10 GB request could
// occupy a cached 77 GB block, the next 77 GB request went to hipMalloc, and at 50 M reads the job spent two minutes in
// hipMalloc / hipFree and finally ran out of HBM with most of it idle.)
// Slabs: small jobs grow in slabs of >= 1 GB; once a process has taken 8 GB (or asks for >= 4 GB at once) the next slab takes
// 90 % of what the device still has free, so a large job pays one hipMalloc.  A free range remembers the stream whose queued
// work may still touch it: ranges never cross devices, and a range last used on another stream (several contexts on one GPU:
// the in-process rank groups of the tests) is handed out only after that stream has drained.
namespace {
std::mutex g_poolMu;
constexpr size_t POOL_ALIGN = 256;
struct FreeRange { size_t size; hipStream_t stream; bool mixed; };
struct Slab { char *base; size_t size; size_t used; std::map<size_t, FreeRange> free; };       // free: offset -> range
struct DevicePool { std::vector<Slab> slabs; size_t total = 0; };
std::map<int, DevicePool> g_pools;
struct PoolLive { int device; int slab; size_t off, size; hipStream_t stream; };   // stream: the one the block was handed out on (its user)
std::unordered_map<void *, PoolLive> g_poolLive;
thread_local hipStream_t tl_poolStream = nullptr;    // stream of the API call this thread is in (poolEnter)
int g_ctxCount = 0;
// plasship_ctx_reserve_async: the arena's big slab is being allocated by a background thread; an allocation that finds no range waits
std::mutex g_reserveMu; std::condition_variable g_reserveCv; int g_reserving = 0;
thread_local bool tl_isReserver = false;
// A process that ends while the reservation thread is still inside hipMalloc (a CLI failure right after plasship_ctx_reserve_async: ADVICE r4):
// this object is constructed after — so destroyed before — the mutexes, the condition variable and the pools above, and before the HIP
// runtime's own exit handlers (registered earlier, when the library's dependency was loaded); its destructor waits for the thread.
struct ReserveJoin { ~ReserveJoin() { std::unique_lock<std::mutex> lk(g_reserveMu); g_reserveCv.wait(lk, [] { return g_reserving == 0; }); } } g_reserveJoin;
size_t g_poolHits = 0, g_poolMisses = 0; double g_poolMissMs = 0, g_poolMissBytes = 0;

// best fit over all free ranges of the device (a few hundred at most)
bool takeRange(DevicePool &dp, int dev, size_t n, void **p, hipStream_t *waitFor, bool *waitAll, bool longLived) {
    int bs = -1; size_t bo = 0, bsz = ~(size_t) 0; const char *bestEnd = nullptr;
    for (size_t i = 0; i < dp.slabs.size(); i++)
        for (auto &kv : dp.slabs[i].free) {
            if (kv.second.size < n) continue;
            if (longLived) {        // the free range that ends highest in the largest slab (the big one of a large job)
                const char *end = dp.slabs[i].base + kv.first + kv.second.size;
                if (bs < 0 || dp.slabs[i].size > dp.slabs[bs].size || (dp.slabs[i].size == dp.slabs[bs].size && end > bestEnd)) { bs = (int) i; bo = kv.first; bsz = kv.second.size; bestEnd = end; }
            } else if (kv.second.size < bsz) { bs = (int) i; bo = kv.first; bsz = kv.second.size; }
        }
    if (bs < 0) return false;
    Slab &sl = dp.slabs[bs];
    const FreeRange fr = sl.free[bo];
    sl.free.erase(bo);
    size_t at = bo;
    if (longLived) { at = bo + fr.size - n; if (fr.size > n) sl.free[bo] = FreeRange{fr.size - n, fr.stream, fr.mixed}; }
    else if (fr.size > n) sl.free[bo + n] = FreeRange{fr.size - n, fr.stream, fr.mixed};
    sl.used += n;
    *p = sl.base + at;
    bo = at;
    g_poolLive[*p] = PoolLive{dev, bs, bo, n, tl_poolStream};
    *waitAll = fr.mixed; *waitFor = (!fr.mixed && fr.stream != tl_poolStream) ? fr.stream : nullptr;
    return true;
}
void trimLocked(int onlyDevice) {      // give completely free slabs back to HIP
    int cur = 0; (void) hipGetDevice(&cur);
    for (auto &dv : g_pools) {
        if (onlyDevice >= 0 && dv.first != onlyDevice) continue;
        bool any = false; for (auto &sl : dv.second.slabs) any |= (sl.used == 0 && sl.base != nullptr);
        if (!any) continue;
        (void) hipSetDevice(dv.first); (void) hipDeviceSynchronize();          // queued work may still reference released ranges
        for (auto &sl : dv.second.slabs) if (sl.used == 0 && sl.base) { (void) hipFree(sl.base); dv.second.total -= sl.size; sl.base = nullptr; sl.size = 0; sl.free.clear(); }
    }
    (void) hipSetDevice(cur);
}
}  // namespace
// debugging aid: PLASSHIP_POOL_POISON=<0..255> fills every block handed out with that byte, so that a kernel reading
// memory it did not write fails the same way on every run (recycled blocks otherwise hold the previous call's data)
static int poisonByte() { static const int v = [] { const char *e = getenv("PLASSHIP_POOL_POISON"); return e ? atoi(e) : -1; }(); return v; }
static hipError_t poolMallocRaw(void **p, size_t n, bool longLived);
hipError_t poolMalloc(void **p, size_t n, bool longLived) {
    const hipError_t e = poolMallocRaw(p, n, longLived);
    if (e == hipSuccess && poisonByte() >= 0) { (void) hipDeviceSynchronize(); (void) hipMemset(*p, poisonByte(), n); (void) hipDeviceSynchronize(); }
    return e;
}
void poolEnter(hipStream_t stream) { tl_poolStream = stream; }
// a context goes away (its stream has been drained): its released ranges no longer wait for anybody
static void poolForgetStream(hipStream_t stream) {
    std::lock_guard<std::mutex> g(g_poolMu);
    for (auto &dv : g_pools) for (auto &sl : dv.second.slabs) for (auto &kv : sl.free) if (!kv.second.mixed && kv.second.stream == stream) kv.second.stream = nullptr;
    for (auto &kv : g_poolLive) if (kv.second.stream == stream) kv.second.stream = nullptr;     // blocks that outlive their context: its queued work is through
    if (tl_poolStream == stream) tl_poolStream = nullptr;
}
static double poolFraction() { static const double v = [] { const char *e = getenv("PLASSHIP_POOL_FRACTION"); const double x = e ? atof(e) : 0.0; return (x > 0.05 && x <= 0.98) ? x : 0.88; }(); return v; }
static hipError_t poolMallocRaw(void **p, size_t n, bool longLived) {
    n = std::max<size_t>((n + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN, POOL_ALIGN);
    int dev = 0; (void) hipGetDevice(&dev);
    const size_t MB2 = (size_t) 2 << 20;
    for (int fails = 0;;) {
        hipStream_t waitFor = nullptr; bool waitAll = false, hit = false;
        size_t slabBytes = 0;
        {
            std::lock_guard<std::mutex> g(g_poolMu);
            DevicePool &dp = g_pools[dev];
            hit = takeRange(dp, dev, n, p, &waitFor, &waitAll, longLived);
            if (hit) g_poolHits++;
            else {
                // a new slab: modest while the process is small, most of the remaining HBM once it is not
                // (round 5: ONE slab of most of the HBM per process — a process that already has it grows in modest steps instead of taking 88 % of
                //  what is left again and again — and never the last 1.5 GB: the HIP runtime allocates the kernels' scratch there on demand.
                //  A pytest process that had run the 50 M-read tests held 99.97 % of the HBM, and the child process of a later test died in a
                //  kernel launch with "out of resources, available free memory 94 MB"; profiles/r05_calls/)
                size_t fr = 0, tt = 0; (void) hipMemGetInfo(&fr, &tt);
                const size_t keepFree = (size_t) 3 << 29;
                const size_t most = std::min<size_t>((size_t) ((double) fr * poolFraction()), fr > keepFree ? fr - keepFree : 0);
                bool haveBigSlab = false;
                for (const auto &sl : dp.slabs) if (sl.base && sl.size >= ((size_t) 16 << 30)) haveBigSlab = true;
                const bool bigJob = !haveBigSlab && (dp.total >= ((size_t) 8 << 30) || n >= ((size_t) 4 << 30));
                slabBytes = bigJob ? most : std::min<size_t>(std::max<size_t>(2 * n, (size_t) 1 << 30), most);
                if (slabBytes < n || fails) slabBytes = n;                  // last resort: exactly what is asked for
                slabBytes = (slabBytes + MB2 - 1) / MB2 * MB2;
            }
        }
        if (hit) {
            if (waitAll) (void) hipDeviceSynchronize();
            else if (waitFor) (void) plasship::streamSync(waitFor);          // the previous user's queued work must be through
            return hipSuccess;
        }
        if (!tl_isReserver) {                                    // the arena is on its way (plasship_ctx_reserve_async): wait for it, look again
            std::unique_lock<std::mutex> lk(g_reserveMu);
            if (g_reserving) { g_reserveCv.wait(lk, [] { return g_reserving == 0; }); continue; }
        }
        const auto t0 = std::chrono::steady_clock::now();
        void *base = nullptr;
        const hipError_t e = hipMalloc(&base, slabBytes);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> g(g_poolMu);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            if (++fails >= 3) return e;
            trimLocked(dev);                                     // completely free slabs go back to HIP, then try again (smaller)
            continue;
        }
        DevicePool &dp = g_pools[dev];
        Slab sl; sl.base = static_cast<char *>(base); sl.size = slabBytes; sl.used = 0; sl.free[0] = FreeRange{slabBytes, nullptr, false};
        size_t idx = dp.slabs.size();                            // reuse the table entry of a slab that was given back
        for (size_t i = 0; i < dp.slabs.size(); i++) if (!dp.slabs[i].base) { idx = i; break; }
        if (idx == dp.slabs.size()) dp.slabs.push_back(std::move(sl)); else dp.slabs[idx] = std::move(sl);
        dp.total += slabBytes; g_poolMisses++; g_poolMissMs += ms; g_poolMissBytes += (double) slabBytes;
    }
}
void poolFree(void *p) {
    std::lock_guard<std::mutex> g(g_poolMu);
    auto it = g_poolLive.find(p);
    if (it == g_poolLive.end()) { (void) hipFree(p); return; }
    const PoolLive lv = it->second;
    g_poolLive.erase(it);
    Slab &sl = g_pools[lv.device].slabs[lv.slab];
    sl.used -= lv.size;
    // the range is tagged with the stream that USED the block (recorded when it was handed out), not with whatever stream the
    // releasing thread entered last: `*_free(NULL, h)` and frees from another context's thread are legal.  A release from a
    // thread that is inside another context's call marks the range `mixed` (reuse waits for the whole device).
    size_t off = lv.off, size = lv.size; hipStream_t stream = lv.stream; bool mixed = (tl_poolStream != nullptr && tl_poolStream != lv.stream);
    auto nx = sl.free.lower_bound(off);
    if (nx != sl.free.end() && off + size == nx->first) {        // merge with the range behind
        if (nx->second.mixed || nx->second.stream != stream) mixed = !(nx->second.stream == nullptr && !nx->second.mixed) || mixed;
        size += nx->second.size; nx = sl.free.erase(nx);
    }
    if (nx != sl.free.begin()) {
        auto pv = std::prev(nx);
        if (pv->first + pv->second.size == off) {                // merge with the range in front
            if (pv->second.mixed || pv->second.stream != stream) mixed = !(pv->second.stream == nullptr && !pv->second.mixed) || mixed;
            off = pv->first; size += pv->second.size; sl.free.erase(pv);
        }
    }
    sl.free[off] = FreeRange{size, stream, mixed};
}
void poolTrim() {
    std::lock_guard<std::mutex> g(g_poolMu);
    trimLocked(-1);
}
static thread_local std::string g_err;
int tuneInt(const char *name, int dflt) {
    char key[96]; snprintf(key, sizeof(key), "PLASSHIP_TUNE_%s", name);
    const char *e = getenv(key); const int v = e ? atoi(e) : 0;
    return v > 0 ? v : dflt;
}
bool traceOn() { static const bool v = getenv("PLASSHIP_TRACE") != nullptr; return v; }
void setError(const std::string &msg) { g_err = msg; }
static std::atomic<unsigned long long> g_hostSyncs(0);
// ---- a wait that can end (sharded runs): see common.hpp, watchEnter ----
thread_local int tl_watchRank = -1, tl_watchWorld = 1;
thread_local const char *tl_watchLast = "none yet";
thread_local unsigned long long tl_watchCount = 0;
thread_local std::string tl_watchMsg;
void watchEnter(const plasship_ctx *ctx) {
    if (ctx && ctx->hasComm && ctx->comm.world > 1) { tl_watchRank = ctx->comm.rank; tl_watchWorld = ctx->comm.world; }
    else { tl_watchRank = -1; tl_watchWorld = 1; }
    tl_watchMsg.clear();
}
void watchCollective(const char *what) { tl_watchLast = what; tl_watchCount++; }
static double commTimeoutSeconds() { static const double v = [] { const char *e = getenv("PLASSHIP_COMM_TIMEOUT_S"); return e ? atof(e) : 300.0; }(); return v; }
hipError_t streamSync(hipStream_t st) {
    g_hostSyncs++;
    const double limit = tl_watchWorld > 1 ? commTimeoutSeconds() : 0.0;
    if (limit <= 0.0) return hipStreamSynchronize(st);
    // poll: a few hundred queries back to back (a kernel chain that is nearly through), then short sleeps
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (spins < 512) continue;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > limit) {
            tl_watchMsg = "sharded run, rank " + std::to_string(tl_watchRank) + " of " + std::to_string(tl_watchWorld) + ": the stream did not drain within " + std::to_string((int) limit) +
                          " s (PLASSHIP_COMM_TIMEOUT_S); last collective enqueued: " + tl_watchLast + " (#" + std::to_string(tl_watchCount) + " on this thread) - a peer that left the call, "
                          "or a link that is down (NCCL_DEBUG=INFO shows RCCL's view)";
            fprintf(stderr, "[plasship] %s\n", tl_watchMsg.c_str());
            return hipErrorNotReady;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(el < 0.01 ? 20 : 200));
    }
}
std::string hipErrStr(hipError_t e, const char *what, const char *file, int line) {
    const std::string std_ = std::string("HIP error ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" + std::to_string(line);
    if (!tl_watchMsg.empty()) { const std::string m = tl_watchMsg + " [" + std_ + "]"; tl_watchMsg.clear(); return m; }
    return std_;
}
__global__ void keysDifferKernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t n, uint32_t *__restrict__ flag) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) if (a[i] != b[i]) *flag = 1u;
}
int deviceKeysDiffer(plasship_ctx *ctx, const uint32_t *a, const uint32_t *b, size_t n, bool *differ) {
    DevBuf dFlag; uint32_t d = 0;
    if (dFlag.alloc(4) != hipSuccess) { setError("out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dFlag.p, 0, 4, ctx->stream));
    if (n) hipLaunchKernelGGL(keysDifferKernel, dim3(std::min<uint32_t>(((uint32_t) n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, a, b, (uint32_t) n, dFlag.as<uint32_t>());
    PH_COPY_SYNC(ctx->stream, &d, dFlag.p, 4, hipMemcpyDeviceToHost);
    *differ = d != 0;
    return PLASSHIP_OK;
}
}  // namespace plasship
using namespace plasship;

extern "C" const char *plasship_last_error(void) { return g_err.c_str(); }
extern "C" const char *plasship_version(void) { return "plasship 0.1 (gfx950)"; }

extern "C" int plasship_ctx_create(int device_ordinal, plasship_ctx **out) {
    if (!out) { setError("plasship_ctx_create: out is NULL"); return PLASSHIP_ERR_ARG; }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        setError("plasship_ctx_create: no HIP device available (this library has no CPU fallback)");
        return PLASSHIP_ERR_DEVICE;
    }
    if (device_ordinal < 0) {
        const char *lr = getenv("LOCAL_RANK");
        device_ordinal = lr ? atoi(lr) % count : 0;
    }
    if (device_ordinal >= count) { setError("plasship_ctx_create: device ordinal out of range"); return PLASSHIP_ERR_ARG; }
    PH_CHECK(hipSetDevice(device_ordinal));
    std::unique_ptr<plasship_ctx> holder(new plasship_ctx());
    plasship_ctx *c = holder.get();
    c->device = device_ordinal;
    hipDeviceProp_t prop;
    PH_CHECK(hipGetDeviceProperties(&prop, device_ordinal));
    c->numCU = prop.multiProcessorCount;
    PH_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    poolEnter(c->stream);
    for (auto &ev : c->ev) PH_CHECK(hipEventCreate(&ev));
    { std::lock_guard<std::mutex> g(g_poolMu); g_ctxCount++; }
    *out = holder.release();
    return PLASSHIP_OK;
}

extern "C" int plasship_ctx_reserve_async(plasship_ctx *ctx) {
    if (!ctx) { setError("plasship_ctx_reserve_async: ctx is NULL"); return PLASSHIP_ERR_ARG; }
    {
        std::lock_guard<std::mutex> g(g_poolMu);
        for (const auto &sl : g_pools[ctx->device].slabs) if (sl.base && sl.size >= ((size_t) 16 << 30)) return PLASSHIP_OK;      // there is one already
    }
    { std::lock_guard<std::mutex> lk(g_reserveMu); if (g_reserving) return PLASSHIP_OK; g_reserving = 1; }
    const int dev = ctx->device;
    std::thread([dev] {
        (void) hipSetDevice(dev);
        tl_isReserver = true;
        void *p = nullptr;
        if (poolMalloc(&p, (size_t) 4 << 30) == hipSuccess) poolFree(p);      // >= 4 GB: "a large job" -> one slab of most of the free HBM
        { std::lock_guard<std::mutex> lk(g_reserveMu); g_reserving = 0; }
        g_reserveCv.notify_all();
    }).detach();
    return PLASSHIP_OK;
}

uint64_t plasship::newDbGeneration() { static std::atomic<uint64_t> g(0); return ++g; }

extern "C" void plasship_ctx_destroy(plasship_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) plasship::streamSync(ctx->stream);
    for (auto &ev : ctx->ev) if (ev) (void) hipEventDestroy(ev);
    for (int i = 0; i < 2; i++) { if (ctx->stage[i]) (void) hipHostFree(ctx->stage[i]); if (ctx->stageEv[i]) (void) hipEventDestroy(ctx->stageEv[i]); }
    if (ctx->pinnedTable) (void) hipHostFree(ctx->pinnedTable);
    // the context's own device buffers (comparator tables, the 11 GB of selected-window cache lines) go back to the arena BEFORE it is
    // trimmed: a slab is returned to the driver only when nothing in it is live (round 4: the cache lines kept the 250 GB slab of a
    // closed context alive, and the next process on the GPU — bench.py's fused-driver child, a test's subprocess — ran out of memory)
    { std::unique_lock<std::mutex> lk(g_reserveMu); g_reserveCv.wait(lk, [] { return g_reserving == 0; }); }      // (a reservation in flight must land first)
    ctx->kmCache.lines.release(); ctx->d_cmpCache.release(); ctx->d_ambKeys.release(); ctx->d_ambVals.release();
    if (ctx->stream) { poolForgetStream(ctx->stream); (void) hipStreamDestroy(ctx->stream); }
    bool last; { std::lock_guard<std::mutex> g(g_poolMu); last = (--g_ctxCount <= 0); }
    if (last) {
        if (getenv("PLASSHIP_POOL_STATS")) fprintf(stderr, "plasship pool: %zu hits, %zu misses (%.1f ms in hipMalloc, %.1f MB)\n", g_poolHits, g_poolMisses, g_poolMissMs, g_poolMissBytes / 1e6);
        poolTrim();
    }
    delete ctx;
}

extern "C" int plasship_ctx_sync(plasship_ctx *ctx) {
    if (!ctx) { setError("ctx is NULL"); return PLASSHIP_ERR_ARG; }
    PH_CHECK(plasship::streamSync(ctx->stream));
    return PLASSHIP_OK;
}
extern "C" unsigned long long plasship_host_syncs(void) { return plasship::g_hostSyncs.load(); }
extern "C" void *plasship_ctx_stream(plasship_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

// ---- host boundary: pinned staging -------------------------------------------------------------------
// Pageable memory reaches the GPU through a driver-side bounce buffer, one synchronous chunk at a time; here the bounce buffers are
// ours (pinned, 2 x 32 MB per context), the host side of a chunk (packing entries into id order, memcpy into a caller's array,
// fwrite) runs on the host threads while the other chunk is on the link, and nothing is allocated per call.
namespace plasship {
static int stageReady(plasship_ctx *ctx) {
    if (ctx->stageBytes) return PLASSHIP_OK;
    size_t want = 32u << 20;
    if (const char *e = getenv("PLASSHIP_STAGE_MB")) { const long v = atol(e); if (v >= 1 && v <= 4096) want = (size_t) v << 20; }
    for (int i = 0; i < 2; i++) {
        PH_CHECK(hipHostMalloc((void **) &ctx->stage[i], want, hipHostMallocDefault));
        PH_CHECK(hipEventCreateWithFlags(&ctx->stageEv[i], hipEventDisableTiming));
    }
    ctx->stageBytes = want;
    return PLASSHIP_OK;
}
void *ctxPinnedTable(plasship_ctx *ctx, size_t bytes) {
    if (bytes > (1u << 20)) return nullptr;
    if (!ctx->pinnedTable && hipHostMalloc(&ctx->pinnedTable, 1u << 20, hipHostMallocDefault) != hipSuccess) { ctx->pinnedTable = nullptr; (void) hipGetLastError(); }
    return ctx->pinnedTable;
}
int stagedUpload(plasship_ctx *ctx, void *dDst, uint64_t total, const std::function<void(char *, uint64_t, uint64_t)> &produce) {
    if (!total) return PLASSHIP_OK;
    int rc = stageReady(ctx); if (rc) return rc;
    const uint64_t CH = ctx->stageBytes; int b = 0; bool used[2] = {false, false};
    for (uint64_t o = 0; o < total; o += CH, b ^= 1) {
        const uint64_t n = std::min<uint64_t>(CH, total - o);
        if (used[b]) PH_CHECK(hipEventSynchronize(ctx->stageEv[b]));            // the copy that read this buffer two chunks ago
        produce(ctx->stage[b], o, n);
        PH_CHECK(hipMemcpyAsync((char *) dDst + o, ctx->stage[b], n, hipMemcpyHostToDevice, ctx->stream));
        PH_CHECK(hipEventRecord(ctx->stageEv[b], ctx->stream)); used[b] = true;
    }
    PH_CHECK(plasship::streamSync(ctx->stream));
    return PLASSHIP_OK;
}
int stagedDownload(plasship_ctx *ctx, const void *dSrc, uint64_t total, const std::function<bool(const char *, uint64_t, uint64_t)> &consume) {
    if (!total) return PLASSHIP_OK;
    int rc = stageReady(ctx); if (rc) return rc;
    const uint64_t CH = ctx->stageBytes;
    auto issue = [&](uint64_t o, int b) -> int {
        const uint64_t n = std::min<uint64_t>(CH, total - o);
        PH_CHECK(hipMemcpyAsync(ctx->stage[b], (const char *) dSrc + o, n, hipMemcpyDeviceToHost, ctx->stream));
        PH_CHECK(hipEventRecord(ctx->stageEv[b], ctx->stream));
        return PLASSHIP_OK;
    };
    rc = issue(0, 0); if (rc) return rc;
    int b = 0;
    for (uint64_t o = 0; o < total; o += CH, b ^= 1) {
        if (o + CH < total) { rc = issue(o + CH, b ^ 1); if (rc) return rc; }   // the other buffer was consumed in the previous round
        PH_CHECK(hipEventSynchronize(ctx->stageEv[b]));
        if (!consume(ctx->stage[b], o, std::min<uint64_t>(CH, total - o))) { (void) plasship::streamSync(ctx->stream); return PLASSHIP_ERR_IO; }
    }
    return PLASSHIP_OK;
}
static void parallelCopy(char *dst, const char *src, uint64_t n) {
    const size_t SL = 1u << 20;
    parallelRanges((size_t) ((n + SL - 1) / SL), [&](int, size_t b, size_t e) {
        const uint64_t o = (uint64_t) b * SL, end = std::min<uint64_t>(n, (uint64_t) e * SL);
        if (end > o) memcpy(dst + o, src + o, (size_t) (end - o));
    }, nullptr, 4);
}
int stagedCopyToDevice(plasship_ctx *ctx, void *dDst, const void *hSrc, uint64_t bytes) {
    return stagedUpload(ctx, dDst, bytes, [&](char *dst, uint64_t o, uint64_t n) { parallelCopy(dst, (const char *) hSrc + o, n); });
}
int stagedCopyToHost(plasship_ctx *ctx, void *hDst, const void *dSrc, uint64_t bytes) {
    return stagedDownload(ctx, dSrc, bytes, [&](const char *src, uint64_t o, uint64_t n) { parallelCopy((char *) hDst + o, src, n); return true; });
}
}  // namespace plasship
using namespace plasship;

// ---- sequence DB ---------------------------------------------------------------------------------
// Bytes the kernels can take: the scoring kernels index their substitution tables with the residue BYTES ((a << 7) | b over
// 123 rows) and blank columns with byte 0, so an entry must consist of bytes 1..122 and end in "\n\0" — what every DB the reference
// writes looks like (letters, '*', the terminators).  One pass over the uploaded data: bytes above 122, NUL bytes, entries without the
// final NUL.  counts[0] = bytes > 122, counts[1] = NUL bytes (must be one per entry), counts[2] = entries whose last byte is not NUL.
namespace plasship {
__global__ __launch_bounds__(256) void validateBytesKernel(const uint4 *__restrict__ data, uint64_t nWords, const uint64_t *__restrict__ off, uint64_t n, unsigned long long *__restrict__ counts) {
    unsigned long long hi = 0, zero = 0, bad = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nWords; i += (uint64_t) gridDim.x * 256) {
        const uint4 w = data[i];
        const uint32_t v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = v[j];
            hi += (unsigned) __popc((((x & 0x7F7F7F7Fu) + 0x05050505u) | x) & 0x80808080u);          // a byte >= 123 (0x7B): +5 carries into bit 7, or bit 7 is set
            zero += (unsigned) __popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu));        // exact zero-byte test
        }
    }
    const char *bytes = reinterpret_cast<const char *>(data);
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t) gridDim.x * 256) bad += bytes[off[i + 1] - 1] != 0;
    hi = waveReduceSumU64(hi); zero = waveReduceSumU64(zero); bad = waveReduceSumU64(bad);
    if ((threadIdx.x & 63) == 0) { if (hi) atomicAdd(&counts[0], hi); if (zero) atomicAdd(&counts[1], zero); if (bad) atomicAdd(&counts[2], bad); }
}
}  // namespace plasship

extern "C" int plasship_seqdb_upload(plasship_ctx *ctx, const char *data, size_t data_bytes, const uint64_t *off,
                                     const uint32_t *elen, const uint32_t *key, size_t n, int dbtype,
                                     plasship_seqdb **out) {
    if (!ctx || !out || (n && (!data || !off || !elen || !key))) { setError("plasship_seqdb_upload: bad argument"); return PLASSHIP_ERR_ARG; }
    if (dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS && dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES) {
        setError("plasship_seqdb_upload: dbtype must be amino acids (0) or nucleotides (1)"); return PLASSHIP_ERR_UNSUPPORTED;
    }
    if (n >= 0xFFFFFFFFull) { setError("plasship_seqdb_upload: too many sequences"); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    // ids are ranks in key order (DBReader::getId); repack the data in id order so that device offsets
    // are monotone and neighbouring ids are neighbours in HBM.
    // (every loop over the entries runs on all host threads: 88 M entries at 50 M reads, and the GPU waits for this — round 4)
    std::vector<uint32_t> perm(n);
    std::atomic<bool> sortedA(true);
    parallelRanges(n, [&](int, size_t b, size_t e) {
        bool ok = true;
        for (size_t i = b; i < e; i++) { perm[i] = (uint32_t) i; if (i > 0 && key[i - 1] > key[i]) ok = false; }
        if (!ok) sortedA = false;
    });
    const bool sorted = sortedA;
    const double tu0 = ioNow();
    if (!sorted) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    std::unique_ptr<plasship_seqdb> holder(new plasship_seqdb());   // released to the caller on success only
    plasship_seqdb *db = holder.get();
    db->dbtype = dbtype; db->n = n;
    db->h_key.resize(n); db->h_elen.resize(n); db->h_off.resize(n);
    std::vector<uint32_t> hlen(n);
    uint64_t total = 0; uint32_t maxE = 0;
    {   // offsets in id order = prefix sums of the entry lengths: per-range sums first, then every range fills its part
        const int maxT = hostThreads();
        std::vector<uint64_t> rangeSum((size_t) maxT + 1, 0), rangeBeg((size_t) maxT + 1, 0); std::vector<uint32_t> rangeMax((size_t) maxT + 1, 0);
        std::vector<std::pair<size_t, size_t>> ranges((size_t) maxT + 1, std::make_pair((size_t) 0, (size_t) 0));
        std::atomic<int> bad(0);
        const int used = parallelRanges(n, [&](int t, size_t b, size_t e) {
            uint64_t sum = 0; uint32_t mx = 0;
            for (size_t i = b; i < e; i++) {
                const uint32_t sI = perm[i];
                if (off[sI] + elen[sI] > data_bytes) { bad = 1; return; }
                if (elen[sI] < 2) { bad = 2; return; }
                sum += elen[sI]; mx = std::max(mx, elen[sI]);
            }
            rangeSum[(size_t) t] = sum; rangeMax[(size_t) t] = mx; ranges[(size_t) t] = std::make_pair(b, e);
        });
        if (bad == 1) { setError("plasship_seqdb_upload: entry beyond data"); return PLASSHIP_ERR_ARG; }
        if (bad == 2) { setError("plasship_seqdb_upload: sequence entry shorter than \"\\n\\0\""); return PLASSHIP_ERR_ARG; }
        // (the ranges are contiguous and ascending in t: parallelRanges hands out [0, n) in order)
        std::vector<int> order((size_t) used); for (int t = 0; t < used; t++) order[(size_t) t] = t;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return ranges[(size_t) x].first < ranges[(size_t) y].first; });
        for (int q = 0; q < used; q++) { const int t = order[(size_t) q]; rangeBeg[(size_t) t] = total; total += rangeSum[(size_t) t]; maxE = std::max(maxE, rangeMax[(size_t) t]); }
        std::vector<std::pair<size_t, size_t>> rs(ranges.begin(), ranges.begin() + used);
        parallelRanges((size_t) used, [&](int, size_t qb, size_t qe) {
            for (size_t q = qb; q < qe; q++) {
                uint64_t o = rangeBeg[q];
                for (size_t i = rs[q].first; i < rs[q].second; i++) {
                    const uint32_t sI = perm[i];
                    db->h_key[i] = key[sI]; db->h_elen[i] = elen[sI]; db->h_off[i] = o; hlen[i] = elen[sI] - 2;
                    o += elen[sI];
                }
            }
        }, nullptr, 1);
    }
    db->dataBytes = total; db->maxEntryLen = maxE; db->residues = total - 2 * (uint64_t) n; db->hostIndexValid = true;
    // pad the data buffer so 16-byte vector loads at the tail stay in bounds
    if (db->d_data.allocLong(total + 64) != hipSuccess || db->d_off.allocLong((n + 1) * 8) != hipSuccess ||
        db->d_len.allocLong((n + 1) * 4) != hipSuccess || db->d_key.allocLong((n + 1) * 4) != hipSuccess) {
        setError("plasship_seqdb_upload: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    const double tu1 = ioNow();
    PH_CHECK(hipMemsetAsync((char *) db->d_data.p + total, 0, 64, ctx->stream));     // the padding only; the entries are copied below
    // entries packed into id order, chunk by chunk, straight into the pinned staging buffers (an entry may straddle two chunks)
    {
        const uint64_t *newOff = db->h_off.data();
        const int rc = stagedUpload(ctx, db->d_data.p, total, [&](char *dst, uint64_t o, uint64_t nb) {
            const size_t first = (size_t) (std::upper_bound(newOff, newOff + n, o) - newOff) - 1;           // entry holding byte o
            const size_t last = (size_t) (std::upper_bound(newOff, newOff + n, o + nb - 1) - newOff);      // one past the entry holding the last byte
            parallelRanges(last - first, [&](int, size_t b, size_t e) {
                for (size_t i = first + b; i < first + e; i++) {
                    const uint64_t eb = newOff[i], ee = eb + db->h_elen[i];
                    const uint64_t cb = std::max(eb, o), ce = std::min(ee, o + nb);
                    if (ce > cb) memcpy(dst + (cb - o), data + off[perm[i]] + (cb - eb), (size_t) (ce - cb));
                }
            }, nullptr, 1024);
        });
        if (rc) return rc;
    }
    const double tu2 = ioNow();
    // file order != key order (a DB several writer threads left behind): remember every entry's rank in the data file
    {
        bool fileSorted = true;
        for (size_t i = 1; i < n && fileSorted; i++) fileSorted = off[perm[i - 1]] <= off[perm[i]];
        if (!fileSorted) {
            std::vector<uint32_t> byOff(n), rank(n);
            std::iota(byOff.begin(), byOff.end(), 0u);
            std::stable_sort(byOff.begin(), byOff.end(), [&](uint32_t x, uint32_t y) { return off[perm[x]] < off[perm[y]]; });
            for (size_t r = 0; r < n; r++) rank[byOff[r]] = (uint32_t) r;
            if (db->d_fileRank.allocLong(n * 4) != hipSuccess) { setError("plasship_seqdb_upload: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            const int rc = stagedCopyToDevice(ctx, db->d_fileRank.p, rank.data(), n * 4); if (rc) return rc;
        }
    }
    std::vector<uint64_t> hoff(n + 1);
    for (size_t i = 0; i < n; i++) hoff[i] = db->h_off[i];
    hoff[n] = total;
    { int rc = stagedCopyToDevice(ctx, db->d_off.p, hoff.data(), (n + 1) * 8); if (rc) return rc; }
    if (n) {
        int rc = stagedCopyToDevice(ctx, db->d_len.p, hlen.data(), n * 4); if (rc) return rc;
        rc = stagedCopyToDevice(ctx, db->d_key.p, db->h_key.data(), n * 4); if (rc) return rc;
    }
    // the bytes: 1..122 inside the entries, one NUL per entry, at its end (the 64 bytes of padding behind the data are NUL: subtracted)
    unsigned long long bc[3] = {0, 0, 0};
    if (n) {
        DevBuf dBC;
        if (dBC.alloc(24) != hipSuccess) { setError("plasship_seqdb_upload: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemsetAsync(dBC.p, 0, 24, ctx->stream));
        const uint64_t nWords = (total + 15) / 16;                  // the last word may reach into the padding
        hipLaunchKernelGGL(validateBytesKernel, dim3((unsigned) std::min<uint64_t>((nWords + 255) / 256, (uint64_t) ctx->numCU * 16)), dim3(256), 0, ctx->stream,
                           (const uint4 *) db->d_data.p, nWords, (const uint64_t *) db->d_off.as<uint64_t>(), (uint64_t) n, dBC.as<unsigned long long>());
        PH_CHECK(hipMemcpyAsync(bc, dBC.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    PH_CHECK(plasship::streamSync(ctx->stream));
    if (ioTimingOn()) fprintf(stderr, "[plasship io] seqdb_upload: %zu entries, %.2f GB: host index arrays + device allocation (waits for the arena) %.3f s, packed upload %.3f s, index upload + validation %.3f s\n",
                              n, (double) total / 1e9, tu1 - tu0, tu2 - tu1, ioNow() - tu2);
    const unsigned long long padZeros = (16 - total % 16) % 16;
    if (bc[0]) { setError("plasship_seqdb_upload: " + std::to_string(bc[0]) + " byte(s) above 'z' (122) in the sequence data: not a sequence DB the kernels' score tables can index"); return PLASSHIP_ERR_ARG; }
    if (bc[2] || bc[1] != (unsigned long long) n + padZeros) {
        setError("plasship_seqdb_upload: every entry must end in \"\\n\\0\" and hold no other NUL byte (" + std::to_string(bc[2]) + " entries without the final NUL, " +
                 std::to_string((long long) bc[1] - (long long) padZeros) + " NUL bytes in " + std::to_string(n) + " entries)"); return PLASSHIP_ERR_ARG;
    }
    *out = holder.release();
    return PLASSHIP_OK;
}

extern "C" int plasship_seqdb_read(plasship_ctx *ctx, const char *db_path, plasship_seqdb **out) {
    if (!ctx || !db_path || !out) { setError("plasship_seqdb_read: bad argument"); return PLASSHIP_ERR_ARG; }
    HostDB h; std::string err;
    const double t0 = ioNow();
    if (!readDBFiles(db_path, h, err)) { setError(err); return PLASSHIP_ERR_IO; }
    const double t1 = ioNow();
    const int rc = plasship_seqdb_upload(ctx, h.data.data(), h.data.size(), h.off.data(), h.elen.data(), h.key.data(), h.key.size(), h.dbtype, out);
    if (ioTimingOn()) fprintf(stderr, "[plasship io] seqdb_read %s: files %.3f s, index arrays + upload + validation %.3f s\n", db_path, t1 - t0, ioNow() - t1);
    return rc;
}

static int ensureHostIndex(plasship_ctx *ctx, plasship_seqdb *db) {
    if (db->hostIndexValid) return PLASSHIP_OK;
    size_t n = db->n;
    db->h_key.resize(n); db->h_off.resize(n + 1); db->h_elen.resize(n);
    std::vector<uint32_t> len(n);
    PH_CHECK(plasship::streamSync(ctx->stream));
    { int rc = stagedCopyToHost(ctx, db->h_off.data(), db->d_off.p, (n + 1) * 8); if (rc) return rc; }
    if (n) {
        int rc = stagedCopyToHost(ctx, db->h_key.data(), db->d_key.p, n * 4); if (rc) return rc;
        rc = stagedCopyToHost(ctx, len.data(), db->d_len.p, n * 4); if (rc) return rc;
    }
    parallelRanges(n, [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) db->h_elen[i] = len[i] + 2; });
    db->h_off.resize(n);
    db->hostIndexValid = true;
    return PLASSHIP_OK;
}

extern "C" int plasship_seqdb_info(const plasship_seqdb *db, size_t *n, uint64_t *residues, uint32_t *max_entry_len,
                                   int *dbtype, uint64_t *data_bytes) {
    if (!db) { setError("plasship_seqdb_info: db is NULL"); return PLASSHIP_ERR_ARG; }
    if (n) *n = db->n;
    if (residues) *residues = db->residues;
    if (max_entry_len) *max_entry_len = db->maxEntryLen;
    if (dbtype) *dbtype = db->dbtype;
    if (data_bytes) *data_bytes = db->dataBytes;
    return PLASSHIP_OK;
}

extern "C" int plasship_seqdb_download(plasship_ctx *ctx, const plasship_seqdb *cdb, char *data, uint64_t *off,
                                       uint32_t *elen, uint32_t *key) {
    if (!ctx || !cdb) { setError("plasship_seqdb_download: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    // (a DB that shares an append-only heap with its ancestors is packed into a buffer of its own first: what leaves is the data file)
    std::unique_ptr<plasship_seqdb> packedTmp;
    if (!cdb->contiguous) { const int rcP = packedCopyOf(ctx, cdb, packedTmp); if (rcP) return rcP; cdb = packedTmp.get(); }
    plasship_seqdb *db = const_cast<plasship_seqdb *>(cdb);
    int rc = ensureHostIndex(ctx, db); if (rc) return rc;
    PH_CHECK(plasship::streamSync(ctx->stream));
    if (data && db->dataBytes) { rc = stagedCopyToHost(ctx, data, db->dataPtr(), db->dataBytes); if (rc) return rc; }
    if (off) memcpy(off, db->h_off.data(), db->n * 8);
    if (elen) memcpy(elen, db->h_elen.data(), db->n * 4);
    if (key) memcpy(key, db->h_key.data(), db->n * 4);
    return PLASSHIP_OK;
}

extern "C" int plasship_seqdb_write(plasship_ctx *ctx, const plasship_seqdb *cdb, const char *db_path) {
    if (!ctx || !cdb || !db_path) { setError("plasship_seqdb_write: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    std::unique_ptr<plasship_seqdb> packedTmp;              // (see plasship_seqdb_download)
    if (!cdb->contiguous) { const int rcP = packedCopyOf(ctx, cdb, packedTmp); if (rcP) return rcP; cdb = packedTmp.get(); }
    plasship_seqdb *db = const_cast<plasship_seqdb *>(cdb);
    int rc = ensureHostIndex(ctx, db); if (rc) return rc;
    PH_CHECK(plasship::streamSync(ctx->stream));
    // the device layout is the file layout (entries "SEQ\n\0" back to back in key order): the data file is the device buffer, streamed
    // through the pinned staging buffers while the previous chunk is being written; the index is formatted on the host threads
    std::string err; DBFileWriter w;
    const double tw0 = ioNow();
    if (!w.open(db_path, db->dbtype, err)) { setError(err); return PLASSHIP_ERR_IO; }
    std::atomic<bool> packed(true);
    parallelRanges(db->n, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; i++) if (db->h_off[i] + db->h_elen[i] != (i + 1 < db->n ? db->h_off[i + 1] : db->dataBytes)) { packed = false; return; }
    });
    if (packed) {
        // (round 6) the index — 88 M lines at 50 M reads, 1.2 s of formatting on the host threads — is written WHILE the data streams off the device
        // (one thread in fwrite, the others idle until now): the two touch different files and different members of the writer
        double tIdx = 0;
        std::thread idxThread([&]() { const double a = ioNow(); w.index(db->h_key.data(), db->h_elen.data(), db->n); tIdx = ioNow() - a; });
        rc = stagedDownload(ctx, db->dataPtr(), db->dataBytes, [&](const char *src, uint64_t, uint64_t nb) { w.data(src, (size_t) nb); return !w.failed; });
        idxThread.join();
        if (rc == PLASSHIP_ERR_IO) setError(std::string("error while writing ") + db_path);
        if (rc) return rc;
        const double tw1 = ioNow() - tIdx;      // (for the timing line below: data, then "index" = what the index cost on its own thread)
        if (ioTimingOn()) fprintf(stderr, "[plasship io] seqdb_write %s: data %.2f GB in %.3f s (%.2f GB/s), index of %zu entries %.3f s\n", db_path, (double) db->dataBytes / 1e9, tw1 - tw0,
                                  (double) db->dataBytes / 1e9 / std::max(tw1 - tw0, 1e-9), db->n, ioNow() - tw1);
    } else {                                                  // a DB with gaps between its entries (none of the producers here makes one)
        HostBytes data;
        if (!data.alloc(db->dataBytes)) { setError("plasship_seqdb_write: out of host memory"); return PLASSHIP_ERR_IO; }
        rc = stagedCopyToHost(ctx, data.data(), db->dataPtr(), db->dataBytes); if (rc) return rc;
        for (size_t i = 0; i < db->n; i++) w.add(db->h_key[i], data.data() + db->h_off[i], db->h_elen[i] - 1);
    }
    if (!w.close(err)) { setError(err); return PLASSHIP_ERR_IO; }
    return PLASSHIP_OK;
}

// ---- digest of a resident DB (include/plasship.h): one thread per entry, FNV-1a is a serial chain per entry ----
namespace plasship {
__global__ __launch_bounds__(256) void digestKernel(const char *__restrict__ data, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len, const uint32_t *__restrict__ key,
                                                    uint32_t n, unsigned long long *__restrict__ out) {
    unsigned long long sum = 0, bytes = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const uint32_t el = len[e] + 2;                                       // the entry as indexed: sequence + "\n\0"
        const unsigned char *p = reinterpret_cast<const unsigned char *>(data) + off[e];
        uint64_t h = 0xCBF29CE484222325ull ^ ((uint64_t) key[e] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t) el << 40);
        uint32_t j = 0;
        // eight bytes per load once the address is aligned (entries start anywhere)
        for (; j < el && ((reinterpret_cast<uintptr_t>(p + j)) & 7u); j++) { h ^= p[j]; h *= 0x100000001B3ull; }
        for (; j + 8 <= el; j += 8) {
            uint64_t w = *reinterpret_cast<const uint64_t *>(p + j);
#pragma unroll
            for (int b = 0; b < 8; b++) { h ^= (w & 0xFFu); h *= 0x100000001B3ull; w >>= 8; }
        }
        for (; j < el; j++) { h ^= p[j]; h *= 0x100000001B3ull; }
        h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
        sum += h; bytes += el;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); bytes += __shfl_xor(bytes, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], sum); atomicAdd(&out[1], bytes); }
}
__global__ void packOffLenKernel(const uint64_t *__restrict__ off, const uint32_t *__restrict__ len, uint64_t n, uint64_t *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = (off[i] << 24) | (uint64_t) (len[i] & 0xFFFFFFu);
}
// plasship_seqdb::d_offLen, built on the first call that looks up random entries of the DB (sequences are shorter than 2^20, a DB
// smaller than 2^40 bytes).  Stream-ordered: the caller's kernels follow on ctx->stream.
int ensureOffLen(plasship_ctx *ctx, const plasship_seqdb *db) {
    if (db->n == 0) return PLASSHIP_OK;
    std::lock_guard<std::mutex> g(db->offLenMu);
    if (db->d_offLen.p) {       // built by another context: its kernel must have run before this stream reads the words
        if (db->offLenStream != ctx->stream && db->offLenEv) PH_CHECK(hipStreamWaitEvent(ctx->stream, db->offLenEv, 0));
        return PLASSHIP_OK;
    }
    if (db->maxEntryLen >= (1u << 24) || db->dataBytes >= (1ull << 40)) { setError("a sequence DB with an entry of 2^24 bytes or more, or of 2^40 bytes or more in total, is not supported"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (db->d_offLen.allocLong(db->n * 8) != hipSuccess) { setError("out of device memory for the packed offsets of a sequence DB"); return PLASSHIP_ERR_DEVICE; }
    hipLaunchKernelGGL(packOffLenKernel, dim3((unsigned) std::min<uint64_t>((db->n + 255) / 256, (uint64_t) ctx->numCU * 16)), dim3(256), 0, ctx->stream,
                       db->d_off.as<uint64_t>(), db->d_len.as<uint32_t>(), (uint64_t) db->n, db->d_offLen.as<uint64_t>());
    db->offLenStream = ctx->stream;
    if (!db->offLenEv) PH_CHECK(hipEventCreateWithFlags(&db->offLenEv, hipEventDisableTiming));
    PH_CHECK(hipEventRecord(db->offLenEv, ctx->stream));
    return PLASSHIP_OK;
}
}  // namespace plasship
extern "C" int plasship_seqdb_digest(plasship_ctx *ctx, const plasship_seqdb *db, uint64_t *digest, uint64_t *entry_bytes) {
    if (!ctx || !db || !digest) { setError("plasship_seqdb_digest: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    DevBuf d; unsigned long long h[2] = {0, 0};
    if (d.alloc(16) != hipSuccess) { setError("plasship_seqdb_digest: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(d.p, 0, 16, ctx->stream));
    if (db->n) hipLaunchKernelGGL(digestKernel, dim3((unsigned) std::min<size_t>((db->n + 255) / 256, (size_t) ctx->numCU * 64)), dim3(256), 0, ctx->stream,
                                  db->dataPtr(), db->d_off.as<uint64_t>(), db->d_len.as<uint32_t>(), db->d_key.as<uint32_t>(), (uint32_t) db->n, d.as<unsigned long long>());
    PH_COPY_SYNC(ctx->stream, h, d.p, 16, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    *digest = h[0]; if (entry_bytes) *entry_bytes = h[1];
    return PLASSHIP_OK;
}

extern "C" void plasship_seqdb_free(plasship_ctx *ctx, plasship_seqdb *db) {
    if (!db) return;
    if (ctx) { (void) hipSetDevice(ctx->device); plasship::poolEnter(ctx->stream); }
    delete db;
}
