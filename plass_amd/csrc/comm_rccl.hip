// plasship_rccl: the native RCCL communicator of a sharded run (include/plasship_rccl.h).  Product code, host only.
// replaces the reference's MPI split of kmermatcher (mm/linclust/kmermatcher.cpp:631-660,736-778; $RUNNER in data/assemble.sh:92).
#include "common.hpp"
#include "../../include/plasship_rccl.h"
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {
using namespace plasship;

// the few RCCL entry points used, resolved at run time (rccl.h: ncclResult_t is an int enum, ncclSuccess == 0, ncclUint8 == 1)
typedef struct { char internal[PLASSHIP_RCCL_ID_BYTES]; } UniqueId;
typedef void *Comm;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};
Rccl &rccl() {
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("PLASSHIP_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { if (!n || !*n) continue; r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) { const char *e = dlerror(); r.error = std::string("cannot load RCCL (librccl.so.1): ") + (e ? e : "?"); return; }   // dlerror() clears the message: call it once
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}
constexpr int NCCL_UINT8 = 1;
constexpr uint64_t PIECE = 256ull << 20;       // largest single message (see the header)
}  // namespace

struct plasship_rccl_comm {
    plasship_ctx *ctx = nullptr; Comm comm = nullptr; int rank = 0, world = 1;
    void *hStage = nullptr, *dStage = nullptr; size_t stageBytes = 0;     // pinned host + device staging of the host all-gather
    uint64_t bytesSent = 0, calls = 0; double seconds = 0; bool failed = false;
};

namespace {
#define RC(call, what)                                                                                              \
    do { const int e_ = (call); if (e_ != 0) { Rccl &r_ = rccl(); setError(std::string("RCCL: ") + (what) + ": " + (r_.GetErrorString ? r_.GetErrorString(e_) : "error")); c->failed = true; return 1; } } while (0)
#define HC(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { setError(hipErrStr(e_, #call, __FILE__, __LINE__)); c->failed = true; return 1; } } while (0)

int ensureStage(plasship_rccl_comm *c, size_t bytes) {
    if (bytes <= c->stageBytes) return 0;
    if (c->hStage) (void) hipHostFree(c->hStage);
    if (c->dStage) (void) hipFree(c->dStage);
    c->hStage = c->dStage = nullptr; c->stageBytes = 0;
    const size_t n = std::max<size_t>(bytes, 1 << 20);
    HC(hipHostMalloc(&c->hStage, n, hipHostMallocDefault));
    HC(hipMalloc(&c->dStage, n));
    c->stageBytes = n;
    return 0;
}
struct Clock { plasship_rccl_comm *c; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
               ~Clock() { c->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); c->calls++; } };

// recv[r * n ...] = send of rank r: host -> pinned -> device, ncclAllGather, device -> pinned -> host
int cbAllgatherHost(void *user, const void *send, void *recv, uint64_t n) {
    plasship_rccl_comm *c = static_cast<plasship_rccl_comm *>(user); Clock clk{c};
    hipStream_t st = c->ctx->stream;
    if (c->world == 1) { memcpy(recv, send, n); return 0; }
    const size_t slot = (size_t) ((n + 15) / 16 * 16);
    if (ensureStage(c, slot * (size_t) (c->world + 1))) return 1;
    char *h = static_cast<char *>(c->hStage), *d = static_cast<char *>(c->dStage);
    memcpy(h, send, n);
    HC(hipMemcpyAsync(d, h, slot, hipMemcpyHostToDevice, st));
    RC(rccl().AllGather(d, d + slot, slot, NCCL_UINT8, c->comm, st), "ncclAllGather");
    HC(hipMemcpyAsync(h + slot, d + slot, slot * (size_t) c->world, hipMemcpyDeviceToHost, st));
    HC(plasship::streamSync(st));
    for (int r = 0; r < c->world; r++) memcpy(static_cast<char *>(recv) + (size_t) r * n, h + slot * (size_t) (r + 1), n);
    return 0;
}
// send[soff[r] .. + sb[r]) -> rank r, recv[roff[r] .. + rb[r]) <- rank r, in rounds of at most PIECE bytes per pair, one group per round
int exchange(plasship_rccl_comm *c, const char *send, const uint64_t *soff, const uint64_t *sb, char *recv, const uint64_t *roff, const uint64_t *rb) {
    hipStream_t st = c->ctx->stream; const int W = c->world, me = c->rank;
    if (sb[me]) HC(hipMemcpyAsync(recv + roff[me], send + soff[me], sb[me], hipMemcpyDeviceToDevice, st));
    uint64_t rounds = 0;
    for (int r = 0; r < W; r++) if (r != me) rounds = std::max(rounds, (std::max(sb[r], rb[r]) + PIECE - 1) / PIECE);
    // a failing call inside an open group closes the group before it returns (the thread must not stay in group mode: the
    // ncclCommAbort of the destroy path runs on it)
#define RCG(call, what) do { const int g_ = (call); if (g_ != 0) { (void) rccl().GroupEnd(); RC(g_, what); } } while (0)
    for (uint64_t k = 0; k < rounds; k++) {
        RC(rccl().GroupStart(), "ncclGroupStart");
        for (int r = 0; r < W; r++) {
            if (r == me) continue;
            uint64_t lo = std::min(k * PIECE, sb[r]), hi = std::min((k + 1) * PIECE, sb[r]);
            if (hi > lo) { RCG(rccl().Send(send + soff[r] + lo, hi - lo, NCCL_UINT8, r, c->comm, st), "ncclSend"); c->bytesSent += hi - lo; }
            lo = std::min(k * PIECE, rb[r]); hi = std::min((k + 1) * PIECE, rb[r]);
            if (hi > lo) RCG(rccl().Recv(recv + roff[r] + lo, hi - lo, NCCL_UINT8, r, c->comm, st), "ncclRecv");
        }
        RC(rccl().GroupEnd(), "ncclGroupEnd");
    }
#undef RCG
    return 0;
}
int cbAlltoallv(void *user, const void *dSend, const uint64_t *sb, void *dRecv, const uint64_t *rb) {
    plasship_rccl_comm *c = static_cast<plasship_rccl_comm *>(user); Clock clk{c};
    std::vector<uint64_t> soff(c->world), roff(c->world); uint64_t s = 0, r = 0;
    for (int i = 0; i < c->world; i++) { soff[i] = s; s += sb[i]; roff[i] = r; r += rb[i]; }
    return exchange(c, static_cast<const char *>(dSend), soff.data(), sb, static_cast<char *>(dRecv), roff.data(), rb);
}
int cbAllgatherv(void *user, const void *dSend, uint64_t n, void *dRecv, const uint64_t *rb) {
    plasship_rccl_comm *c = static_cast<plasship_rccl_comm *>(user); Clock clk{c};
    if (rb[c->rank] != n) { setError("RCCL: all-gather size mismatch"); return 1; }
    std::vector<uint64_t> soff(c->world, 0), sb(c->world, n), roff(c->world); uint64_t r = 0;
    for (int i = 0; i < c->world; i++) { roff[i] = r; r += rb[i]; }
    return exchange(c, static_cast<const char *>(dSend), soff.data(), sb.data(), static_cast<char *>(dRecv), roff.data(), rb);
}
}  // namespace

extern "C" int plasship_rccl_get_unique_id(void *id_out) {
    if (!id_out) { setError("plasship_rccl_get_unique_id: bad argument"); return PLASSHIP_ERR_ARG; }
    Rccl &r = rccl();
    if (!r.error.empty()) { setError(r.error); return PLASSHIP_ERR_UNSUPPORTED; }
    UniqueId id;
    const int e = r.GetUniqueId(&id);
    if (e != 0) { setError(std::string("RCCL: ncclGetUniqueId: ") + r.GetErrorString(e)); return PLASSHIP_ERR_DEVICE; }
    memcpy(id_out, &id, PLASSHIP_RCCL_ID_BYTES);
    return PLASSHIP_OK;
}

extern "C" int plasship_rccl_comm_create(plasship_ctx *ctx, int rank, int world, const void *id, plasship_rccl_comm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) { setError("plasship_rccl_comm_create: bad argument"); return PLASSHIP_ERR_ARG; }
    Rccl &r = rccl();
    if (!r.error.empty()) { setError(r.error); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    std::unique_ptr<plasship_rccl_comm> c(new plasship_rccl_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    UniqueId uid; memcpy(&uid, id, PLASSHIP_RCCL_ID_BYTES);
    // ncclCommInitRank is a rendezvous of all ranks: on hardware nobody has run this on yet it is the first place a sharded job can wait for
    // ever (a rank that never arrives, a fabric RCCL cannot bring up).  It runs on a helper thread with a deadline (PLASSHIP_COMM_TIMEOUT_S,
    // default 300, 0 = none); past it the call fails with the rank in the message and the caller can choose another communicator
    // (bench.py --comm auto: torch.distributed).  The helper thread is left behind in that case — RCCL offers no way to cancel the call.
    struct InitBox { std::mutex mu; std::condition_variable cv; bool done = false; int e = 0; Comm comm = nullptr; };
    auto box = std::make_shared<InitBox>();
    const int dev = ctx->device;
    std::thread th([box, &r, world, uid, rank, dev]() {
        (void) hipSetDevice(dev);
        Comm cm = nullptr;
        const int e = r.CommInitRank(&cm, world, uid, rank);
        std::lock_guard<std::mutex> g(box->mu); box->e = e; box->comm = cm; box->done = true; box->cv.notify_all();
    });
    {
        const char *te = getenv("PLASSHIP_COMM_TIMEOUT_S"); const double limit = te ? atof(te) : 300.0;
        std::unique_lock<std::mutex> lk(box->mu);
        const bool ok = limit <= 0.0 ? (box->cv.wait(lk, [&] { return box->done; }), true) : box->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return box->done; });
        if (!ok) {
            lk.unlock(); th.detach();
            setError("RCCL: rank " + std::to_string(rank) + " of " + std::to_string(world) + ": ncclCommInitRank did not return within " + std::to_string((int) limit) +
                     " s (PLASSHIP_COMM_TIMEOUT_S) - not every rank reached the rendezvous, or RCCL cannot bring the links up (NCCL_DEBUG=INFO)");
            fprintf(stderr, "[plasship] %s\n", plasship_last_error());
            return PLASSHIP_ERR_PEER;
        }
    }
    th.join();
    c->comm = box->comm;
    const int e = box->e;
    if (e != 0) { setError(std::string("RCCL: rank ") + std::to_string(rank) + ": ncclCommInitRank: " + r.GetErrorString(e)); return PLASSHIP_ERR_DEVICE; }
    plasship_comm pc; memset(&pc, 0, sizeof(pc));
    pc.rank = rank; pc.world = world; pc.user = c.get();
    pc.allgather_host = cbAllgatherHost; pc.alltoallv_dev = cbAlltoallv; pc.allgatherv_dev = cbAllgatherv;
    pc.stream_ordered = 1;
    const int rc = plasship_ctx_set_comm(ctx, &pc);
    if (rc) { (void) r.CommDestroy(c->comm); return rc; }
    *out = c.release();
    return PLASSHIP_OK;
}

extern "C" int plasship_rccl_comm_stats(plasship_rccl_comm *c, uint64_t *bytes_sent, double *seconds, uint64_t *calls, int reset) {
    if (!c) { setError("plasship_rccl_comm_stats: bad argument"); return PLASSHIP_ERR_ARG; }
    if (bytes_sent) *bytes_sent = c->bytesSent;
    if (seconds) *seconds = c->seconds;
    if (calls) *calls = c->calls;
    if (reset) { c->bytesSent = 0; c->seconds = 0; c->calls = 0; }
    return PLASSHIP_OK;
}

extern "C" void plasship_rccl_comm_destroy(plasship_ctx *ctx, plasship_rccl_comm *c) {
    if (!c) return;
    if (ctx) { (void) hipSetDevice(ctx->device); (void) plasship::streamSync(ctx->stream); (void) plasship_ctx_set_comm(ctx, nullptr); }
    Rccl &r = rccl();
    if (c->comm) { if (c->failed && r.CommAbort) (void) r.CommAbort(c->comm); else if (r.CommDestroy) (void) r.CommDestroy(c->comm); }
    if (c->hStage) (void) hipHostFree(c->hStage);
    if (c->dStage) (void) hipFree(c->dStage);
    delete c;
}
