// rccl_stub — TEST INFRASTRUCTURE (never linked into the product): the ten nccl* entry points plass_amd/csrc/comm_rccl.hip resolves
// with dlsym, implemented over an in-process rendezvous so that the NATIVE communicator (include/plasship_rccl.h) runs with
// W > 1 ranks on a box with ONE GPU: W threads, W plasship contexts (W streams) on the same device, `PLASSHIP_RCCL_LIB` pointing
// at this library.  What executes is the product's own exchange(): its 256 MiB piece rounds, offsets, ncclGroupStart/End pairs,
// the pinned-buffer ncclAllGather of the host arrays and the 8-byte status rounds — only the wire is replaced.
//
// Semantics kept from RCCL: calls are STREAM ORDERED (a send reads its buffer after the work queued on the sender's stream, a
// receive is visible to work queued behind it on the receiver's stream, the sender's later work waits until its buffer has been
// read), sends and receives between a pair of ranks match in posting order, a group is atomic, sizes of a matched pair must agree.
// Differences: ncclGroupEnd blocks the calling thread until its peers have posted the matching operations (RCCL would return and
// let the proxy thread wait) — the library issues collectives from all ranks, so this only serialises what would overlap.
//
// Build: hipcc -shared -fPIC tests/tools/rccl_stub.cpp -o tests/tools/librccl_stub.so   (done by __graft_entry__.build())
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {
enum { OK = 0, ERR_UNHANDLED = 1, ERR_SYSTEM = 2, ERR_INTERNAL = 3, ERR_ARG = 4, ERR_USAGE = 5 };
struct UniqueId { char internal[128]; };

struct SendPost {                 // a send waiting for its receive
    const char *src; size_t bytes; hipEvent_t ready;      // `ready`: recorded on the sender's stream when the send was posted
    hipEvent_t done = nullptr; bool taken = false, finished = false;   // `done`: recorded on the receiver's stream behind the copy
    bool sizeMismatch = false;
};
struct Group {
    int world = 0; int joined = 0, left = 0;
    std::mutex mu; std::condition_variable cv;
    // mailbox[from][to]: sends posted by `from` for `to`, in order
    std::vector<std::vector<std::deque<SendPost *>>> box;
    // all-gather rounds: a reusable barrier (generation + arrivals) and every rank's contribution
    uint64_t agGen = 0; int agPosted = 0;
    std::vector<const char *> agSrc; std::vector<hipEvent_t> agReady, agDone; std::vector<size_t> agSize;
    bool aborted = false;
    unsigned long long nSend = 0, nAllGather = 0, bytes = 0;
};
struct Comm { Group *g; int rank; std::string key; std::vector<hipEvent_t> events; };

std::mutex g_mu;
std::map<std::string, Group *> g_groups;

struct Op { bool send; const char *src; char *dst; size_t bytes; int peer; Comm *c; hipStream_t st; };
thread_local int tl_depth = 0;
thread_local std::vector<Op> tl_ops;

size_t typeBytes(int dt) {          // ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8, bf16 9
    switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}
hipEvent_t newEvent(Comm *c) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    c->events.push_back(e);           // destroyed with the communicator: every wait on it has been queued by then
    return e;
}

// runs the posted operations of one group: all sends are published first, then every receive waits for its send
int flush(std::vector<Op> &ops) {
    if (ops.empty()) return OK;
    int rc = OK;
    std::vector<SendPost *> mine;     // my sends, to be waited for at the end
    // 1. publish the sends
    for (Op &o : ops) {
        if (!o.send) continue;
        Group *g = o.c->g;
        SendPost *p = new SendPost();
        p->src = o.src; p->bytes = o.bytes; p->ready = newEvent(o.c);
        if (!p->ready || hipEventRecord(p->ready, o.st) != hipSuccess) { delete p; return ERR_UNHANDLED; }
        { std::lock_guard<std::mutex> l(g->mu); g->box[o.c->rank][o.peer].push_back(p); g->nSend++; g->bytes += o.bytes; }
        g->cv.notify_all();
        mine.push_back(p);
    }
    // 2. receives: wait for the matching send, copy on MY stream behind the sender's `ready`
    for (Op &o : ops) {
        if (o.send) continue;
        Group *g = o.c->g; SendPost *p = nullptr;
        {
            std::unique_lock<std::mutex> l(g->mu);
            auto &q = g->box[o.peer][o.c->rank];
            g->cv.wait(l, [&] { if (g->aborted) return true; for (SendPost *s : q) if (!s->taken) return true; return false; });
            if (g->aborted) { rc = ERR_SYSTEM; break; }
            for (SendPost *s : q) if (!s->taken) { p = s; break; }
            p->taken = true;
        }
        bool bad = p->bytes != o.bytes;
        hipEvent_t done = newEvent(o.c);
        if (!bad && (!done || hipStreamWaitEvent(o.st, p->ready, 0) != hipSuccess ||
                     (o.bytes && hipMemcpyAsync(o.dst, p->src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) ||
                     hipEventRecord(done, o.st) != hipSuccess)) { rc = ERR_UNHANDLED; done = nullptr; }
        if (bad) { fprintf(stderr, "rccl_stub: size mismatch, rank %d receives %zu bytes from rank %d which sends %zu\n", o.c->rank, o.bytes, o.peer, p->bytes); rc = ERR_ARG; }
        { std::lock_guard<std::mutex> l(g->mu); p->done = done; p->sizeMismatch = bad; p->finished = true; }
        g->cv.notify_all();
    }
    // 3. my sends: later work on my stream must not touch the send buffers before the receivers have read them
    for (size_t i = 0, j = 0; i < ops.size(); i++) {
        if (!ops[i].send) continue;
        Op &o = ops[i]; SendPost *p = mine[j++]; Group *g = o.c->g;
        {
            std::unique_lock<std::mutex> l(g->mu);
            g->cv.wait(l, [&] { return p->finished || g->aborted; });
            auto &q = g->box[o.c->rank][o.peer];
            for (auto it = q.begin(); it != q.end(); ++it) if (*it == p) { q.erase(it); break; }
            if (!p->finished) { rc = ERR_SYSTEM; continue; }     // aborted: the post is leaked on purpose (a peer may still look at it)
        }
        if (p->sizeMismatch) rc = ERR_ARG;
        else if (p->done && hipStreamWaitEvent(o.st, p->done, 0) != hipSuccess) rc = ERR_UNHANDLED;
        delete p;
    }
    ops.clear();
    return rc;
}
}  // namespace

extern "C" {
const char *ncclGetErrorString(int e) {
    switch (e) { case OK: return "no error"; case ERR_UNHANDLED: return "unhandled hip error (stub)"; case ERR_SYSTEM: return "communicator aborted (stub)";
                 case ERR_INTERNAL: return "internal error (stub)"; case ERR_ARG: return "invalid argument (stub)"; default: return "invalid usage (stub)"; }
}
int ncclGetUniqueId(UniqueId *id) {
    if (!id) return ERR_ARG;
    static std::mutex m; static std::mt19937_64 rng(std::random_device{}()); static uint64_t counter = 0;
    std::lock_guard<std::mutex> l(m);
    memset(id->internal, 0, sizeof(id->internal));
    const uint64_t a = rng(), b = ++counter; memcpy(id->internal, "rccl_stub", 9); memcpy(id->internal + 16, &a, 8); memcpy(id->internal + 24, &b, 8);
    return OK;
}
int ncclCommInitRank(void **out, int world, UniqueId id, int rank) {
    if (!out || world < 1 || rank < 0 || rank >= world) return ERR_ARG;
    const std::string key(id.internal, sizeof(id.internal));
    Group *g;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_groups.find(key);
        if (it == g_groups.end()) {
            g = new Group(); g->world = world; g->box.assign(world, std::vector<std::deque<SendPost *>>(world));
            g->agSrc.assign(world, nullptr); g->agReady.assign(world, nullptr); g->agDone.assign(world, nullptr); g->agSize.assign(world, 0);
            g_groups[key] = g;
        } else g = it->second;
        if (g->world != world) return ERR_ARG;
    }
    {   // like RCCL: the call returns when every rank has joined
        std::unique_lock<std::mutex> l(g->mu);
        g->joined++;
        g->cv.notify_all();
        g->cv.wait(l, [&] { return g->joined >= g->world || g->aborted; });
    }
    Comm *c = new Comm(); c->g = g; c->rank = rank; c->key = key;
    *out = c;
    return OK;
}
static int leave(Comm *c, bool abort) {
    if (!c) return ERR_ARG;
    Group *g = c->g; bool last;
    {
        std::lock_guard<std::mutex> l(g->mu);
        if (abort) g->aborted = true;
        last = (++g->left >= g->world);
        g->cv.notify_all();               // under the lock: the last rank to leave deletes the group
    }
    (void) hipDeviceSynchronize();
    for (hipEvent_t e : c->events) (void) hipEventDestroy(e);
    if (last) {
        if (getenv("RCCL_STUB_STATS")) fprintf(stderr, "rccl_stub: group of %d ranks: %llu sends (%llu bytes), %llu all-gathers\n", g->world, g->nSend, g->bytes, g->nAllGather);
        std::lock_guard<std::mutex> l(g_mu); g_groups.erase(c->key); delete g;
    }
    delete c;
    return OK;
}
int ncclCommDestroy(void *c) { return leave(static_cast<Comm *>(c), false); }
int ncclCommAbort(void *c) { return leave(static_cast<Comm *>(c), true); }
int ncclGroupStart() { tl_depth++; return OK; }
int ncclGroupEnd() {
    if (tl_depth <= 0) return ERR_USAGE;
    if (--tl_depth > 0) return OK;
    return flush(tl_ops);
}
int ncclSend(const void *buf, size_t count, int dt, int peer, void *comm, hipStream_t st) {
    Comm *c = static_cast<Comm *>(comm); const size_t tb = typeBytes(dt);
    if (!c || !tb || peer < 0 || peer >= c->g->world || peer == c->rank) return ERR_ARG;
    tl_ops.push_back(Op{true, static_cast<const char *>(buf), nullptr, count * tb, peer, c, st});
    return tl_depth > 0 ? OK : flush(tl_ops);
}
int ncclRecv(void *buf, size_t count, int dt, int peer, void *comm, hipStream_t st) {
    Comm *c = static_cast<Comm *>(comm); const size_t tb = typeBytes(dt);
    if (!c || !tb || peer < 0 || peer >= c->g->world || peer == c->rank) return ERR_ARG;
    tl_ops.push_back(Op{false, nullptr, static_cast<char *>(buf), count * tb, peer, c, st});
    return tl_depth > 0 ? OK : flush(tl_ops);
}
// reusable barrier over the ranks of a group; false when the group was aborted
static bool barrier(Group *g) {
    std::unique_lock<std::mutex> l(g->mu);
    const uint64_t gen = g->agGen;
    if (++g->agPosted >= g->world) { g->agPosted = 0; g->agGen++; g->cv.notify_all(); return !g->aborted; }
    g->cv.wait(l, [&] { return g->agGen != gen || g->aborted; });
    return !g->aborted;
}
// recv[r * count ...] = send of rank r, on every rank.  Three barriers: all posted / all copies queued / all done-events taken
// (the slots of a round are free again only behind the third)
int ncclAllGather(const void *send, void *recv, size_t count, int dt, void *comm, hipStream_t st) {
    Comm *c = static_cast<Comm *>(comm); const size_t tb = typeBytes(dt);
    if (!c || !tb) return ERR_ARG;
    Group *g = c->g; const size_t bytes = count * tb; const int W = g->world;
    hipEvent_t ready = newEvent(c), done = newEvent(c);
    if (!ready || !done || hipEventRecord(ready, st) != hipSuccess) return ERR_UNHANDLED;
    { std::lock_guard<std::mutex> l(g->mu); g->agSrc[c->rank] = static_cast<const char *>(send); g->agReady[c->rank] = ready; g->agSize[c->rank] = bytes; g->nAllGather++; }
    if (!barrier(g)) return ERR_SYSTEM;
    int rc = OK;
    for (int r = 0; r < W; r++) if (g->agSize[r] != bytes) rc = ERR_ARG;          // every rank sees the same table: all of them return it
    for (int r = 0; r < W && rc == OK; r++) {
        if (hipStreamWaitEvent(st, g->agReady[r], 0) != hipSuccess ||
            (bytes && hipMemcpyAsync(static_cast<char *>(recv) + (size_t) r * bytes, g->agSrc[r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)) rc = ERR_UNHANDLED;
    }
    if (hipEventRecord(done, st) != hipSuccess) rc = ERR_UNHANDLED;
    { std::lock_guard<std::mutex> l(g->mu); g->agDone[c->rank] = done; }
    if (!barrier(g)) return ERR_SYSTEM;
    // nobody's send buffer may be rewritten before every rank has read it
    for (int r = 0; r < W; r++) if (hipStreamWaitEvent(st, g->agDone[r], 0) != hipSuccess) rc = ERR_UNHANDLED;
    if (!barrier(g)) return ERR_SYSTEM;
    return rc;
}
}  // extern "C"
