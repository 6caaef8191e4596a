// plasship: findassemblystart on gfx950.  Product code.
//
// Reference behaviour reproduced (src/assembler/findassemblystart.cpp:35-176): every query that contains an 'M' votes
// together with its non-self alignments on "the residue in front of the M column is a stop": the query contributes its first
// 'M' (:73-84), an alignment contributes the target residue in the same column if `qStart >= posM && posM <= qEnd` (the
// condition as the reference writes it, :108) — otherwise a vote without position.  If at least 20 % of the votes say "stop"
// (float division, :128-130) the M column becomes a candidate start of every voter; a sequence's start is the MAXIMUM over
// all queries it voted with (:131-139, an atomic compare-exchange loop in the reference, atomicMax here), and the output
// sequence is "*" + the sequence from that column on (:162-166).  The plass workflow runs it once, inside iteration 0,
// between two kmermatcher / rescorediagonal passes (data/assemble.sh:110-141).
// One thread per query: a read fragment is ~50 residues and has ~3 alignments.
#include "common.hpp"
#include "device_utils.hpp"
#include <algorithm>
#include <memory>
#include <cstring>

namespace plasship {

__global__ __launch_bounds__(256) void findStartVoteKernel(SeqView s, const uint64_t *__restrict__ qoff, const AlnRec *__restrict__ recs, int *__restrict__ addStop) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < s.n; q += gridDim.x * blockDim.x) {
        const uint64_t a0 = qoff[q], a1 = qoff[q + 1];
        if (a1 == a0) continue;                                  // no entry for this query (sharded run: not owned)
        const char *seq = s.data + s.off[q];
        const uint32_t L = s.len[q];
        int posM = -1;
        for (uint32_t i = 0; i < L && posM < 0; i += 8) {       // first 'M', eight residues per load (entries are padded)
            uint64_t w; __builtin_memcpy(&w, seq + i, 8);
            const uint64_t x = w ^ 0x4D4D4D4D4D4D4D4Dull;       // 'M' -> zero byte
            uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
            if (z) { const uint32_t p = i + (uint32_t) (__builtin_ctzll(z) >> 3); if (p < L) posM = (int) p; else break; }
        }
        if (posM < 0) continue;
        uint32_t votes = 1, stops = (posM > 0 && seq[posM - 1] == '*') ? 1u : 0u;
        for (uint64_t j = a0; j < a1; j++) {
            const AlnRec r = recs[j];
            if (r.target == q || !r.accepted) continue;        // the self hit; a hole of a sparse list (common.hpp)
            votes++;
            if (r.qStart >= posM && posM <= r.qEnd) {
                const int dbMPos = r.dbStart + (posM - r.qStart);
                const char *t = s.data + s.off[r.target];
                if (dbMPos > 0 && t[dbMPos] == 'M' && t[dbMPos - 1] == '*') stops++;
            }
        }
        if (votes > 1 && (float) stops / (float) votes >= 0.2f) {
            atomicMax(&addStop[q], posM);
            for (uint64_t j = a0; j < a1; j++) {
                const AlnRec r = recs[j];
                if (r.target == q || !r.accepted) continue;
                if (r.qStart >= posM && posM <= r.qEnd) atomicMax(&addStop[r.target], r.dbStart + (posM - r.qStart));
            }
        }
    }
}

__global__ void findStartLenKernel(SeqView s, const int *__restrict__ addStop, uint32_t *__restrict__ flags, uint32_t *__restrict__ newLen, uint64_t *__restrict__ bytes) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < s.n; id += gridDim.x * blockDim.x) {
        const int m = addStop[id];
        const bool cut = m >= 0;
        const uint32_t L = cut ? 1u + (s.len[id] - (uint32_t) m) : 0u;
        flags[id] = cut ? 0x20u : 0u; newLen[id] = L; bytes[id] = L;
    }
}

__global__ void findStartFillKernel(SeqView s, const int *__restrict__ addStop, const uint64_t *__restrict__ start, char *__restrict__ arena) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < s.n; id += gridDim.x * blockDim.x) {
        const int m = addStop[id];
        if (m < 0) continue;
        char *d = arena + start[id];
        const char *src = s.data + s.off[id] + m;
        const uint32_t n = s.len[id] - (uint32_t) m;
        d[0] = '*';
        for (uint32_t i = 0; i < n; i++) d[1 + i] = src[i];
    }
}

__global__ void maxIntRowsKernel(const int *__restrict__ all, uint32_t n, int world, int *__restrict__ out) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += gridDim.x * blockDim.x) {
        int m = -1;
        for (int r = 0; r < world; r++) m = max(m, all[(size_t) r * n + id]);
        out[id] = m;
    }
}

}  // namespace plasship
using namespace plasship;

static int findStartImpl(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_alns *al, plasship_seqdb **out, plasship_findstart_stats *stats);
extern "C" int plasship_find_assembly_start(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_alns *al, plasship_seqdb **out,
                                            plasship_findstart_stats *stats) {
    if (!ctx || !db || !al || !out) { setError("plasship_find_assembly_start: bad argument"); return PLASSHIP_ERR_ARG; }
    if (db->dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS) { setError("plasship_find_assembly_start: needs a protein sequence DB"); return PLASSHIP_ERR_ARG; }
    if (al->nQueries != db->n) { setError("plasship_find_assembly_start: alignment list does not belong to the DB"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    return commFinish(ctx, findStartImpl(ctx, db, al, out, stats));
}
static int findStartImpl(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_alns *al, plasship_seqdb **out, plasship_findstart_stats *stats) {
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const SeqView sv = db->view();
    DevBuf dStop, dFlags, dNewLen, dBytes, dStart, dTmp, dArena;
    const size_t tmpBytes = exclusiveScanTmpBytes((size_t) N + 2);
    if (dStop.alloc(((size_t) N + 1) * 4) != hipSuccess || dFlags.alloc(((size_t) N + 1) * 4) != hipSuccess || dNewLen.alloc(((size_t) N + 1) * 4) != hipSuccess ||
        dBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dStart.alloc(((size_t) N + 2) * 8) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) {
        setError("plasship_find_assembly_start: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    PH_CHECK(hipMemsetAsync(dStop.p, 0xFF, ((size_t) N + 1) * 4, st));                   // -1
    const unsigned grid = std::min<uint32_t>((N + 255) / 256 + 1, (uint32_t) ctx->numCU * 16);
    if (N) hipLaunchKernelGGL(findStartVoteKernel, dim3(grid), dim3(256), 0, st, sv, al->d_qoff.as<uint64_t>(), al->d_recs.as<AlnRec>(), dStop.as<int>());
    if (const plasship_comm *cm = commOf(ctx)) {
        // sharded run: a rank has voted with the queries it owns; a sequence's start is the maximum over all ranks
        DevBuf gAll; std::vector<uint64_t> rb((size_t) cm->world, (uint64_t) N * 4);
        const int rc = commAllgathervBytesKnown(ctx, dStop.p, (uint64_t) N * 4, gAll, rb);
        if (rc) return rc;
        if (N) hipLaunchKernelGGL(maxIntRowsKernel, dim3(grid), dim3(256), 0, st, gAll.as<int>(), N, cm->world, dStop.as<int>());
        PH_CHECK(plasship::streamSync(st));
    }
    if (N) hipLaunchKernelGGL(findStartLenKernel, dim3(grid), dim3(256), 0, st, sv, dStop.as<int>(), dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dBytes.as<uint64_t>());
    if (exclusiveScanU64(st, dBytes.as<uint64_t>(), dStart.as<uint64_t>(), N, dTmp.p, tmpBytes)) { setError("plasship_find_assembly_start: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t arenaBytes = 0;
    PH_COPY_SYNC(st, &arenaBytes, dStart.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost);
    if (dArena.alloc(arenaBytes + 64) != hipSuccess) { setError("plasship_find_assembly_start: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (N) hipLaunchKernelGGL(findStartFillKernel, dim3(grid), dim3(256), 0, st, sv, dStop.as<int>(), dStart.as<uint64_t>(), dArena.as<char>());
    plasship_seqdb *o = nullptr;
    const int rc = buildOutputDB(ctx, db, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dStart.as<uint64_t>(), dArena.as<char>(), 1, dTmp.p, tmpBytes, &o,
                                 nullptr, nullptr, 0, ctx->ev[1]);
    if (rc != PLASSHIP_OK) return rc;
    o->dbtype = PLASSHIP_DBTYPE_AMINO_ACIDS;                                              // findassemblystart.cpp:47
    if (stats) {
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        stats->ms_kernel = ms; stats->n_alignments = al->nLines;
        stats->out_residues = o->residues;
    }
    *out = o;
    return PLASSHIP_OK;
}
