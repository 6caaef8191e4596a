// Sharded operation: one read set over several GPUs, one context per GPU (include/plasship.h, plasship_ctx_set_comm).
// The reference splits kmermatcher over MPI ranks by k-mer hash range (mm/linclust/kmermatcher.cpp:312,736-778) and merges
// the ranks' files on disk; here the ranks exchange device buffers through three collectives the caller supplies
// (RCCL via torch.distributed in bench.py).  This file holds the wrappers the stages use; the exchanges themselves are in
// kmermatch.hip (k-mer records -> bucket owner, grouped records -> rep owner) and assemble.hip (extended sequences).
#include "common.hpp"
#include <cstring>
#include <string>

using namespace plasship;

extern "C" int plasship_ctx_set_comm(plasship_ctx *ctx, const plasship_comm *comm) {
    if (!ctx) { setError("plasship_ctx_set_comm: ctx is NULL"); return PLASSHIP_ERR_ARG; }
    if (!comm) { ctx->hasComm = false; memset(&ctx->comm, 0, sizeof(ctx->comm)); return PLASSHIP_OK; }
    if (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || !comm->allgather_host || !comm->alltoallv_dev || !comm->allgatherv_dev) {
        setError("plasship_ctx_set_comm: bad communicator (rank/world out of range or a collective is missing)"); return PLASSHIP_ERR_ARG;
    }
    if (comm->world > 1024) { setError("plasship_ctx_set_comm: more than 1024 ranks"); return PLASSHIP_ERR_UNSUPPORTED; }
    ctx->comm = *comm; ctx->hasComm = true;
    return PLASSHIP_OK;
}

extern "C" int plasship_ctx_debug_fail_collective(plasship_ctx *ctx, int nth) {
    if (!ctx) { setError("plasship_ctx_debug_fail_collective: ctx is NULL"); return PLASSHIP_ERR_ARG; }
    ctx->debugFailCollective = nth;
    return PLASSHIP_OK;
}

extern "C" int plasship_ctx_copy_d2d(plasship_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    if (!ctx) { setError("plasship_ctx_copy_d2d: ctx is NULL"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    if (bytes) PH_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    PH_CHECK(plasship::streamSync(ctx->stream));
    return PLASSHIP_OK;
}

namespace plasship {

// one status round: every rank contributes 0 (fine) or its error code; a non-zero code of ANY rank ends the call on every rank
static int commStatus(plasship_ctx *ctx, int myCode) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm || cm->world == 1) return PLASSHIP_OK;
    std::vector<int64_t> all((size_t) cm->world, 0);
    const int64_t mine = myCode;
    if (cm->allgather_host(cm->user, &mine, all.data(), 8) != 0) { if (!myCode) setError("sharded run: the caller's allgather_host failed"); return PLASSHIP_ERR_DEVICE; }
    for (int r = 0; r < cm->world; r++)
        if (all[(size_t) r] != 0 && r != cm->rank) {
            if (!myCode) setError("sharded run: rank " + std::to_string(r) + " failed inside this call (code " + std::to_string((long long) all[(size_t) r]) + "); all ranks leave it");
            return PLASSHIP_ERR_PEER;
        }
    return PLASSHIP_OK;
}
int commFinish(plasship_ctx *ctx, int rc) {
    if (rc == PLASSHIP_ERR_PEER) return rc;                  // everybody saw the same status round and is leaving: no further round
    const int peer = commStatus(ctx, rc);
    return rc ? rc : peer;
}

int commAllgatherHost(plasship_ctx *ctx, const void *send, void *recv, uint64_t bytesPerRank) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm) { memcpy(recv, send, bytesPerRank); return PLASSHIP_OK; }
    PH_CHECK(plasship::streamSync(ctx->stream));
    if (ctx->debugFailCollective >= 0 && ctx->debugFailCollective-- == 0) { setError("sharded run: injected rank-local failure (plasship_ctx_debug_fail_collective)"); return PLASSHIP_ERR_DEVICE; }
    { const int rc = commStatus(ctx, 0); if (rc) return rc; }
    watchCollective("host all-gather");
    if (cm->allgather_host(cm->user, send, recv, bytesPerRank) != 0) { setError("sharded run: the caller's allgather_host failed"); return PLASSHIP_ERR_DEVICE; }
    return PLASSHIP_OK;
}

template <typename F> static int allReduceU64(plasship_ctx *ctx, uint64_t *v, size_t n, F f) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm || cm->world == 1 || n == 0) return PLASSHIP_OK;
    std::vector<uint64_t> all((size_t) cm->world * n);
    const int rc = commAllgatherHost(ctx, v, all.data(), n * 8);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) { uint64_t x = all[i]; for (int r = 1; r < cm->world; r++) x = f(x, all[(size_t) r * n + i]); v[i] = x; }
    return PLASSHIP_OK;
}
int commAllReduceSumU64(plasship_ctx *ctx, uint64_t *v, size_t n) { return allReduceU64(ctx, v, n, [](uint64_t a, uint64_t b) { return a + b; }); }
int commAllReduceMaxU64(plasship_ctx *ctx, uint64_t *v, size_t n) { return allReduceU64(ctx, v, n, [](uint64_t a, uint64_t b) { return a > b ? a : b; }); }
int commAllReduceMinU64(plasship_ctx *ctx, uint64_t *v, size_t n) { return allReduceU64(ctx, v, n, [](uint64_t a, uint64_t b) { return a < b ? a : b; }); }

// a rank-local condition that precedes a collective: all ranks learn whether it held everywhere and leave together if not
// (a rank that returned alone would leave the others waiting inside the collective)
int commAgreeOk(plasship_ctx *ctx, bool ok, const char *what) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm || cm->world == 1) { if (!ok) { setError(what); return PLASSHIP_ERR_DEVICE; } return PLASSHIP_OK; }
    uint64_t bad = ok ? 0 : 1;
    const int rc = commAllReduceMaxU64(ctx, &bad, 1);
    if (rc) return rc;
    if (bad) { setError(ok ? std::string(what) + " (on another rank)" : std::string(what)); return PLASSHIP_ERR_DEVICE; }
    return PLASSHIP_OK;
}

int commAlltoallvRecords(plasship_ctx *ctx, const void *dSend, const uint64_t *sendCount, size_t recordBytes, DevBuf &recv,
                         uint64_t *recvTotal, uint64_t slackRecords, uint64_t *allTotal) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm) { setError("sharded run: no communicator"); return PLASSHIP_ERR_ARG; }
    const int W = cm->world;
    // counts first: row r of the gathered matrix = what rank r sends to everybody
    std::vector<uint64_t> mine(sendCount, sendCount + W), all((size_t) W * W);
    int rc = commAllgatherHost(ctx, mine.data(), all.data(), (uint64_t) W * 8);
    if (rc) return rc;
    std::vector<uint64_t> sendBytes(W), recvBytes(W); uint64_t tot = 0;
    for (int r = 0; r < W; r++) { sendBytes[r] = sendCount[r] * recordBytes; const uint64_t c = all[(size_t) r * W + cm->rank]; recvBytes[r] = c * recordBytes; tot += c; }
    rc = commAgreeOk(ctx, recv.alloc(std::max<uint64_t>(tot + slackRecords, 1) * recordBytes) == hipSuccess, "sharded run: out of device memory for the receive buffer");
    if (rc) return rc;
    if (!cm->stream_ordered) PH_CHECK(plasship::streamSync(ctx->stream));
    watchCollective("all-to-all(v) of device records");
    if (cm->alltoallv_dev(cm->user, dSend, sendBytes.data(), recv.p, recvBytes.data()) != 0) { setError("sharded run: the caller's alltoallv_dev failed"); return PLASSHIP_ERR_DEVICE; }
    *recvTotal = tot;
    if (allTotal) { uint64_t a = 0; for (uint64_t c : all) a += c; *allTotal = a; }
    return PLASSHIP_OK;
}

int commAllgathervBytes(plasship_ctx *ctx, const void *dSend, uint64_t sendBytes, DevBuf &recv, std::vector<uint64_t> &recvBytes) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm) { setError("sharded run: no communicator"); return PLASSHIP_ERR_ARG; }
    recvBytes.assign(cm->world, 0);
    const int rc = commAllgatherHost(ctx, &sendBytes, recvBytes.data(), 8);
    if (rc) return rc;
    return commAllgathervBytesKnown(ctx, dSend, sendBytes, recv, recvBytes);
}

int commAllgathervBytesKnown(plasship_ctx *ctx, const void *dSend, uint64_t sendBytes, DevBuf &recv, const std::vector<uint64_t> &recvBytes) {
    const plasship_comm *cm = commOf(ctx);
    if (!cm) { setError("sharded run: no communicator"); return PLASSHIP_ERR_ARG; }
    const int W = cm->world;
    if ((int) recvBytes.size() != W || recvBytes[cm->rank] != sendBytes) { setError("sharded run: inconsistent all-gather sizes"); return PLASSHIP_ERR_ARG; }
    uint64_t tot = 0; for (int r = 0; r < W; r++) tot += recvBytes[r];
    const int rcA = commAgreeOk(ctx, recv.alloc(tot + 64) == hipSuccess, "sharded run: out of device memory for the gather buffer");
    if (rcA) return rcA;
    if (!cm->stream_ordered) PH_CHECK(plasship::streamSync(ctx->stream));
    watchCollective("all-gather(v) of device bytes");
    if (cm->allgatherv_dev(cm->user, dSend, sendBytes, recv.p, recvBytes.data()) != 0) { setError("sharded run: the caller's allgatherv_dev failed"); return PLASSHIP_ERR_DEVICE; }
    return PLASSHIP_OK;
}

}  // namespace plasship
