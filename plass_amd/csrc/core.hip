// plasship: context + sequence DB residency (C-ABI part 1).  Product code.
//   replaces DBReader<unsigned int>::open/getData/getSeqLen/getDbKey for the hot modules
//   (mm/commons/DBReader.cpp:150-215,548-589; DBReader.h:185-213) and DBWriter for sequence DBs.
#include "common.hpp"
#include "host_util.hpp"
#include <algorithm>
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
// ---- caching allocator (see common.hpp) -----------------------------------------------------------
namespace {
std::mutex g_poolMu;
// A cached block remembers the device it lives on and the stream whose work may still be using it: blocks never cross devices,
// and a block last used on another stream is handed out only after that stream has drained (several contexts on one GPU — the
// in-process rank groups of the tests — share the pool; within one stream reuse is ordered by the stream itself).
struct PoolBlock { void *p; hipStream_t stream; };
std::map<int, std::multimap<size_t, PoolBlock>> g_poolFree;   // device -> size class -> cached block
struct PoolLive { size_t cls; int device; };
std::unordered_map<void *, PoolLive> g_poolLive;     // block -> size class, device
thread_local hipStream_t tl_poolStream = nullptr;    // stream of the API call this thread is in (poolEnter)
int g_ctxCount = 0;
size_t g_poolHits = 0, g_poolMisses = 0; double g_poolMissMs = 0, g_poolMissBytes = 0;
size_t sizeClass(size_t n) {
    if (n < 4096) return 4096;
    int e = 63 - __builtin_clzll((unsigned long long) n);      // 2^e <= n
    const size_t step = (size_t) 1 << (e - 3);                 // eight classes per octave
    return (n + step - 1) / step * step;
}
}  // namespace
// debugging aid: PLASSHIP_POOL_POISON=<0..255> fills every block handed out with that byte, so that a kernel reading
// memory it did not write fails the same way on every run (recycled blocks otherwise hold the previous call's data)
static int poisonByte() { static const int v = [] { const char *e = getenv("PLASSHIP_POOL_POISON"); return e ? atoi(e) : -1; }(); return v; }
static hipError_t poolMallocRaw(void **p, size_t n);
hipError_t poolMalloc(void **p, size_t n) {
    const hipError_t e = poolMallocRaw(p, n);
    if (e == hipSuccess && poisonByte() >= 0) { (void) hipDeviceSynchronize(); (void) hipMemset(*p, poisonByte(), n); (void) hipDeviceSynchronize(); }
    return e;
}
void poolEnter(hipStream_t stream) { tl_poolStream = stream; }
// a context goes away (its stream has been drained): its cached blocks no longer wait for anybody
static void poolForgetStream(hipStream_t stream) {
    std::lock_guard<std::mutex> g(g_poolMu);
    for (auto &dv : g_poolFree) for (auto &kv : dv.second) if (kv.second.stream == stream) kv.second.stream = nullptr;
    if (tl_poolStream == stream) tl_poolStream = nullptr;
}
static hipError_t poolMallocRaw(void **p, size_t n) {
    const size_t c = sizeClass(n);
    int dev = 0; (void) hipGetDevice(&dev);
    {
        hipStream_t waitFor = nullptr; bool hit = false;
        {
        std::lock_guard<std::mutex> g(g_poolMu);
        auto &freeMap = g_poolFree[dev];
        // best fit: the smallest cached block that is large enough, as long as it is not absurdly larger
        // (iterations shrink and grow their arrays; an exact-class match would miss and fall into hipMalloc)
        // (small requests may take a block up to 8x their size; a large request only one with <= 25 % slack — at 50 M reads a
        //  10 GB request that takes a cached 77 GB record array makes the next iteration's record arrays a fresh hipMalloc, and
        //  the job runs out of HBM with most of it idle inside oversized blocks)
        auto it = freeMap.lower_bound(c);
        if (it != freeMap.end() && (it->first <= (size_t) 1 << 20 || (c <= ((size_t) 64 << 20) ? it->first <= 8 * c : it->first <= c + c / 4))) {
            *p = it->second.p; const size_t got = it->first;
            if (it->second.stream != tl_poolStream) waitFor = it->second.stream;
            freeMap.erase(it); g_poolLive[*p] = PoolLive{got, dev}; g_poolHits++; hit = true;
        }
        }
        if (hit) {
            if (waitFor) (void) hipStreamSynchronize(waitFor);          // the previous user's queued work must be through
            return hipSuccess;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, c);
    if (e != hipSuccess) { (void) hipGetLastError(); poolTrim(); e = hipMalloc(p, c); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (e == hipSuccess) { std::lock_guard<std::mutex> g(g_poolMu); g_poolLive[*p] = PoolLive{c, dev}; g_poolMisses++; g_poolMissMs += ms; g_poolMissBytes += (double) c; }
    return e;
}
void poolFree(void *p) {
    std::lock_guard<std::mutex> g(g_poolMu);
    auto it = g_poolLive.find(p);
    if (it == g_poolLive.end()) { (void) hipFree(p); return; }
    g_poolFree[it->second.device].emplace(it->second.cls, PoolBlock{p, tl_poolStream});
    g_poolLive.erase(it);
}
void poolTrim() {
    std::lock_guard<std::mutex> g(g_poolMu);
    // cached blocks may still be referenced by queued work of their last stream: drain it before the memory goes back to HIP
    int cur = 0; (void) hipGetDevice(&cur);
    for (auto &dv : g_poolFree) {
        if (dv.second.empty()) continue;
        (void) hipSetDevice(dv.first); (void) hipDeviceSynchronize();
        for (auto &kv : dv.second) (void) hipFree(kv.second.p);
        dv.second.clear();
    }
    (void) hipSetDevice(cur);
}
static thread_local std::string g_err;
int tuneInt(const char *name, int dflt) {
    char key[96]; snprintf(key, sizeof(key), "PLASSHIP_TUNE_%s", name);
    const char *e = getenv(key); const int v = e ? atoi(e) : 0;
    return v > 0 ? v : dflt;
}
bool traceOn() { static const bool v = getenv("PLASSHIP_TRACE") != nullptr; return v; }
void setError(const std::string &msg) { g_err = msg; }
std::string hipErrStr(hipError_t e, const char *what, const char *file, int line) {
    return std::string("HIP error ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" + std::to_string(line);
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

extern "C" void plasship_ctx_destroy(plasship_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->ev) if (ev) (void) hipEventDestroy(ev);
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
    PH_CHECK(hipStreamSynchronize(ctx->stream));
    return PLASSHIP_OK;
}
extern "C" void *plasship_ctx_stream(plasship_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

// ---- sequence DB ---------------------------------------------------------------------------------
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
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = key[i - 1] <= key[i];
    if (!sorted) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    std::unique_ptr<plasship_seqdb> holder(new plasship_seqdb());   // released to the caller on success only
    plasship_seqdb *db = holder.get();
    db->dbtype = dbtype; db->n = n;
    db->h_key.resize(n); db->h_elen.resize(n); db->h_off.resize(n);
    uint64_t total = 0; uint32_t maxE = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t s = perm[i];
        if (off[s] + elen[s] > data_bytes) { setError("plasship_seqdb_upload: entry beyond data"); return PLASSHIP_ERR_ARG; }
        if (elen[s] < 2) { setError("plasship_seqdb_upload: sequence entry shorter than \"\\n\\0\""); return PLASSHIP_ERR_ARG; }
        db->h_key[i] = key[s]; db->h_elen[i] = elen[s]; db->h_off[i] = total;
        total += elen[s]; maxE = std::max(maxE, elen[s]);
    }
    db->dataBytes = total; db->maxEntryLen = maxE; db->residues = total - 2 * (uint64_t) n; db->hostIndexValid = true;
    std::vector<uint32_t> hlen(n);
    for (size_t i = 0; i < n; i++) hlen[i] = db->h_elen[i] - 2;
    // pad the data buffer so 16-byte vector loads at the tail stay in bounds
    if (db->d_data.alloc(total + 64) != hipSuccess || db->d_off.alloc((n + 1) * 8) != hipSuccess ||
        db->d_len.alloc((n + 1) * 4) != hipSuccess || db->d_key.alloc((n + 1) * 4) != hipSuccess) {
        setError("plasship_seqdb_upload: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipMemsetAsync((char *) db->d_data.p + total, 0, 64, ctx->stream));     // the padding only; the entries are copied below
    // stage in id order through a pinned-size bounce buffer
    {
        const size_t CH = 64u << 20;
        std::vector<char> bounce(std::min<uint64_t>(CH, std::max<uint64_t>(total, 1)));
        uint64_t done = 0; size_t i = 0;
        while (i < n) {
            size_t fill = 0; uint64_t base = db->h_off[i];
            while (i < n && fill + db->h_elen[i] <= bounce.size()) {
                memcpy(bounce.data() + fill, data + off[perm[i]], db->h_elen[i]); fill += db->h_elen[i]; i++;
            }
            if (fill == 0) {   // single entry larger than the bounce buffer
                PH_COPY_SYNC(ctx->stream, (char *) db->d_data.p + base, data + off[perm[i]], db->h_elen[i], hipMemcpyHostToDevice);
                i++;
            } else {
                PH_COPY_SYNC(ctx->stream, (char *) db->d_data.p + base, bounce.data(), fill, hipMemcpyHostToDevice);
            }
            done = base + fill;
        }
        (void) done;
    }
    std::vector<uint64_t> hoff(n + 1);
    for (size_t i = 0; i < n; i++) hoff[i] = db->h_off[i];
    hoff[n] = total;
    PH_COPY_SYNC(ctx->stream, db->d_off.p, hoff.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    if (n) {
        PH_COPY_SYNC(ctx->stream, db->d_len.p, hlen.data(), n * 4, hipMemcpyHostToDevice);
        PH_COPY_SYNC(ctx->stream, db->d_key.p, db->h_key.data(), n * 4, hipMemcpyHostToDevice);
    }
    PH_CHECK(hipStreamSynchronize(ctx->stream));
    *out = holder.release();
    return PLASSHIP_OK;
}

extern "C" int plasship_seqdb_read(plasship_ctx *ctx, const char *db_path, plasship_seqdb **out) {
    if (!ctx || !db_path || !out) { setError("plasship_seqdb_read: bad argument"); return PLASSHIP_ERR_ARG; }
    HostDB h; std::string err;
    if (!readDBFiles(db_path, h, err)) { setError(err); return PLASSHIP_ERR_IO; }
    return plasship_seqdb_upload(ctx, h.data.data(), h.data.size(), h.off.data(), h.elen.data(), h.key.data(), h.key.size(), h.dbtype, out);
}

static int ensureHostIndex(plasship_ctx *ctx, plasship_seqdb *db) {
    if (db->hostIndexValid) return PLASSHIP_OK;
    size_t n = db->n;
    db->h_key.resize(n); db->h_off.resize(n + 1); db->h_elen.resize(n);
    std::vector<uint32_t> len(n);
    PH_CHECK(hipStreamSynchronize(ctx->stream));
    PH_COPY_SYNC(ctx->stream, db->h_off.data(), db->d_off.p, (n + 1) * 8, hipMemcpyDeviceToHost);
    if (n) {
        PH_COPY_SYNC(ctx->stream, db->h_key.data(), db->d_key.p, n * 4, hipMemcpyDeviceToHost);
        PH_COPY_SYNC(ctx->stream, len.data(), db->d_len.p, n * 4, hipMemcpyDeviceToHost);
    }
    for (size_t i = 0; i < n; i++) db->h_elen[i] = len[i] + 2;
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
    plasship_seqdb *db = const_cast<plasship_seqdb *>(cdb);
    PH_ENTER(ctx);
    int rc = ensureHostIndex(ctx, db); if (rc) return rc;
    PH_CHECK(hipStreamSynchronize(ctx->stream));
    if (data && db->dataBytes) PH_COPY_SYNC(ctx->stream, data, db->d_data.p, db->dataBytes, hipMemcpyDeviceToHost);
    if (off) memcpy(off, db->h_off.data(), db->n * 8);
    if (elen) memcpy(elen, db->h_elen.data(), db->n * 4);
    if (key) memcpy(key, db->h_key.data(), db->n * 4);
    return PLASSHIP_OK;
}

extern "C" int plasship_seqdb_write(plasship_ctx *ctx, const plasship_seqdb *cdb, const char *db_path) {
    if (!ctx || !cdb || !db_path) { setError("plasship_seqdb_write: bad argument"); return PLASSHIP_ERR_ARG; }
    plasship_seqdb *db = const_cast<plasship_seqdb *>(cdb);
    PH_ENTER(ctx);
    int rc = ensureHostIndex(ctx, db); if (rc) return rc;
    std::vector<char> data(db->dataBytes);
    PH_CHECK(hipStreamSynchronize(ctx->stream));
    if (db->dataBytes) PH_COPY_SYNC(ctx->stream, data.data(), db->d_data.p, db->dataBytes, hipMemcpyDeviceToHost);
    // device layout already is the canonical DB layout: entries in key order, each "SEQ\n\0"
    std::string err; DBFileWriter w;
    if (!w.open(db_path, db->dbtype, err)) { setError(err); return PLASSHIP_ERR_IO; }
    for (size_t i = 0; i < db->n; i++) w.add(db->h_key[i], data.data() + db->h_off[i], db->h_elen[i] - 1);
    if (!w.close(err)) { setError(err); return PLASSHIP_ERR_IO; }
    return PLASSHIP_OK;
}

extern "C" void plasship_seqdb_free(plasship_ctx *ctx, plasship_seqdb *db) {
    if (!db) return;
    if (ctx) { (void) hipSetDevice(ctx->device); plasship::poolEnter(ctx->stream); }
    delete db;
}
