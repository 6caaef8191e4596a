// plasship: kmermatcher on gfx950 (rows K1–K8 of SURVEY.md §8a).  Product code.
//
// Reference behaviour reproduced (file:line in /root/reference/lib/mmseqs/src):
//   linclust/kmermatcher.cpp:77-385    fillKmerPositionArray: letter mapping, k-mer index, XXH64 score,
//                                      per-sequence selection of the lowest-hash k-mers, identity record
//   linclust/kmermatcher.cpp:408-412   sort #1 (kmer, seqLen desc, id, pos)
//   linclust/kmermatcher.cpp:450-559   assignGroup: rep = first of each equal-kmer run, diagonal, filter
//   linclust/kmermatcher.cpp:427-431   sort #2 (rep, target, diagonal)
//   linclust/kmermatcher.cpp:809-924   writeKmerMatcherResult: best diagonal per (rep, target)
//   linclust/kmermatcher.cpp:705-724   every key gets an entry ("key\t0\t0" self line first)
//
// MI355X design (not a translation of the CPU algorithm):
//   * extraction: one wavefront per sequence, codes staged through LDS in 64-position tiles; the
//     reference's 65 536-bin threshold walk becomes a two-level 256-bin LDS radix select, the per
//     sequence std::sort becomes an LDS bitonic sort of only the <= ~60 candidate k-mers.
//   * sort #1 is NOT a sort: only grouping by equal k-mer and the identity of the run's first record
//     matter (its order is destroyed by sort #2 anyway).  Records are hash-partitioned over the line
//     store (linepart.hpp: 1–2 levels, unstable, no histogram pass) into buckets that fit an LDS hash
//     table; the run head is one atomicMin over a packed (seqLen desc, id, pos, strand) word.  One
//     read+write per level instead of the 8+ passes of an LSD radix sort over 16-byte records.
//   * sort #2 must be a true sort (the reference scans across rep boundaries, Appendix A.3): records
//     are range-partitioned by rep id (order preserving) and each bucket is bitonic-sorted in LDS.
//   * per-(rep,target) reduction: one thread per run head walks its run.
// Integer work only; the single float expression (Util::canBeCovered) is IEEE-exact.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include "linepart.hpp"
#include "xxh64_u64.hpp"
#include <algorithm>
#include <memory>
#include <type_traits>
#include <cstdlib>
#include <climits>
#include <cstring>

namespace plasship {

#include "kmermatch_extract.hpp"      // K1-K3: slot bounds, extraction + selection (sections 1, 2, 2b, 2c)
#include "kmermatch_group.hpp"        // K5: assignGroup (section 4)
#include "kmermatch_repsort.hpp"      // K6-K8: rep sort as aggregation, run reduction, stale-record scan (sections 5-7)
}  // namespace plasship
using namespace plasship;

// ---- host orchestration --------------------------------------------------------------------------------
namespace {

struct Timer {
    plasship_ctx *ctx; int slot;
    void start(int s) { slot = s; (void) hipEventRecord(ctx->ev[s], ctx->stream); }
    float stop(int s2) { (void) hipEventRecord(ctx->ev[s2], ctx->stream); (void) hipEventSynchronize(ctx->ev[s2]); float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[slot], ctx->ev[s2]); return ms; }
};

static inline unsigned gridFor(uint64_t n, unsigned block, unsigned cap = 65535u * 8) {
    uint64_t g = (n + block - 1) / block; if (g < 1) g = 1; if (g > cap) g = cap; return (unsigned) g;
}
static int ceilLog2(uint64_t x) { int b = 0; while ((1ULL << b) < x) b++; return b; }

// what the host needs about the target of the last run (the stale-record check), gathered on the device so that it travels with
// the group kernel's counts in one round trip: {maxRT, slotOff[T], slotOff[T+1], len[T]}
__global__ void lastRunInfoKernel(const unsigned long long *__restrict__ maxRT, const uint64_t *__restrict__ slotOff, const uint32_t *__restrict__ len,
                                  uint32_t n, unsigned long long *__restrict__ out) {
    const unsigned long long m = *maxRT;
    const uint32_t t = (uint32_t) (m & 0xFFFFFFFFull);
    out[0] = m;
    if (t < n) { out[1] = slotOff[t]; out[2] = slotOff[t + 1]; out[3] = len[t]; } else { out[1] = out[2] = out[3] = 0; }
}

// PLASSHIP_TRACE: records whose sequence id is out of range (sentinels excepted) — a consistency probe between the stages
template <bool LONG>
__global__ void countBadIdsKernel(const void *recs, uint64_t n, uint32_t nSeq, unsigned long long *out) {
    const Rec<LONG> *g = reinterpret_cast<const Rec<LONG> *>(recs);
    unsigned long long bad = 0, sen = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const Rec<LONG> r = g[i];
        if (isSentinel(r)) sen++; else if (r.id >= nSeq) bad++;
    }
    if (bad) atomicAdd(&out[0], bad);
    if (sen) atomicAdd(&out[1], sen);
}
template <bool LONG>
static void traceBadIds(plasship_ctx *ctx, const char *what, const void *recs, uint64_t n, uint32_t nSeq) {
    if (!traceOn()) return;
    DevBuf d; unsigned long long h[2] = {0, 0};
    if (d.alloc(16) != hipSuccess) return;
    (void) hipMemsetAsync(d.p, 0, 16, ctx->stream);
    hipLaunchKernelGGL((countBadIdsKernel<LONG>), dim3(1024), dim3(256), 0, ctx->stream, recs, n, nSeq, d.as<unsigned long long>());
    (void) hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e = plasship::streamSync(ctx->stream);
    fprintf(stderr, "[plasship] %s: %llu records, %llu with an id out of range, %llu sentinels (%s)\n", what, (unsigned long long) n, h[0], h[1], hipGetErrorString(e));
}

// sharded run: id ranges with equal shares of the k-mer record slots.  out[r] = first id of rank r (r = 0..W), out[W+1+r] = its slot
__global__ void splitIdsKernel(const uint64_t *__restrict__ slotOff, uint32_t n, int W, uint64_t *__restrict__ out) {
    const uint64_t total = slotOff[n];
    for (int r = threadIdx.x; r <= W; r += blockDim.x) {
        uint32_t lo = 0, hi = n;
        if (r == W) lo = n;
        else {
            const uint64_t want = (uint64_t) (((unsigned __int128) total * (unsigned) r) / (unsigned) W);
            while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (slotOff[mid] < want) lo = mid + 1; else hi = mid; }
        }
        out[r] = lo; out[W + 1 + r] = slotOff[lo];
    }
}

// =====================================================================================================
// single GPU: hash grouping and rep sort over the line store (linepart.hpp)
// =====================================================================================================
// geometry of the k-mer side, fixed by the number of record slots (known before the extraction): partition levels, piece sizes,
// and how many lines the two record buffers must hold
struct LineGeo {
    int b1 = 0, b2 = 0; uint32_t nb1 = 1, nb2 = 0, PL1 = 1, PL2 = 0;
    uint64_t totalLines = 0, nP1 = 0, cap1 = 0, maxP2 = 0, cap2 = 0;
    uint32_t lastValid = RPL;
};
static uint32_t pieceLinesFor(uint64_t lines, uint32_t nb, int numCU, uint32_t minFactor) {
    // a piece leaves one partial line per bucket: >= minFactor * nb lines per piece bounds that waste; beyond that, enough pieces
    // to give every CU a few
    const uint64_t want = (lines + 4ull * (uint64_t) numCU - 1) / (4ull * (uint64_t) numCU);
    return (uint32_t) std::max<uint64_t>((uint64_t) nb * minFactor, std::min<uint64_t>((uint64_t) nb * 64, std::max<uint64_t>(want, 1)));
}
// totalSlots: slots of THIS rank's sequences (what level 1 reads); sharded run: slotsAll = slots of all ranks — the bucket bits are
// those of the whole run, the same on every rank, and there are at least W level-1 buckets (rank r owns a contiguous range of them)
static LineGeo lineGeometry(uint64_t totalSlots, bool lng, int numCU, uint64_t slotsAll = 0, int W = 1) {
    LineGeo g;
    const int maxBits = lng ? 9 : 10;                         // LDS: 2^bits open lines of RPL records
    const int lw = ceilLog2((uint64_t) std::max(W, 1));
    const int totalBits = std::min(2 * maxBits, std::max(lw, std::max(0, ceilLog2(((slotsAll ? slotsAll : totalSlots) + 1535) / 1536))));   // ~1000-1500 records per final bucket
    g.b1 = totalBits <= maxBits ? totalBits : std::max((totalBits + 1) / 2, lw); g.b2 = totalBits - g.b1;
    g.nb1 = 1u << g.b1; g.nb2 = g.b2 ? 1u << g.b2 : 0u;
    g.totalLines = (totalSlots + RPL - 1) / RPL;
    g.lastValid = g.totalLines ? (uint32_t) (totalSlots - (g.totalLines - 1) * RPL) : (uint32_t) RPL;
    g.PL1 = pieceLinesFor(g.totalLines, g.nb1, numCU, 8);
    g.nP1 = (g.totalLines + g.PL1 - 1) / g.PL1;
    g.cap1 = std::max<uint64_t>(g.nP1 * ((uint64_t) g.PL1 + g.nb1), 1);
    if (g.nb2) {
        // level 2: ONE piece per level-1 bucket (its output range is the bucket's "region"; hash buckets are evenly filled)
        g.PL2 = 0xFFFFFFFFu; g.maxP2 = g.nb1; g.cap2 = g.cap1 + (uint64_t) g.nb1 * g.nb2;     // (sharded run: cap2 follows from what the exchange delivers)
    }
    return g;
}
// lines a range partition of `nRec` records in `nSeg` dense segments can need (level 1) and a second level on top of it
static uint64_t repLevel1Cap(uint64_t nRec, uint64_t nSeg, uint32_t PL, uint32_t nb) {
    const uint64_t lines = (nRec + RPL - 1) / RPL + nSeg;
    return lines + (lines / PL + nSeg + 1) * (uint64_t) nb;
}

__global__ void arenaStartKernel(const uint32_t *__restrict__ lineBeg, uint32_t bpb, uint32_t gGrid, uint32_t nBuckets, uint64_t *__restrict__ arenaStart) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < gGrid; j += gridDim.x * blockDim.x)
        arenaStart[j] = (uint64_t) lineBeg[std::min(j * bpb, nBuckets - 1)] * RPL;
}
// scratch need of the aggregation kernel per bucket (buckets beyond its LDS capacity): 2 (nucleotides: 3) * pow2ceil(records) 8-byte words
__global__ void bigNeedKernel(const uint32_t *__restrict__ lineCnt, const uint32_t *__restrict__ unique, uint32_t nBuckets, uint64_t *__restrict__ need, uint32_t wordsPerEntry) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nBuckets; b += gridDim.x * blockDim.x) {
        const uint64_t c = (uint64_t) lineCnt[b] * RPL;
        uint64_t v = 0;
        if (unique[b] == AGG_NEEDS_SCRATCH) { uint64_t P = 1; while (P < c) P <<= 1; v = wordsPerEntry * P; }      // only what pass 1 left over (keys, counts[, rank words])
        need[b] = v;
    }
}
__global__ void sumU32Kernel(const uint32_t *__restrict__ v, uint32_t n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += v[i];
    s = waveReduceSumU64(s);
    if (laneId() == 0 && s) atomicAdd(out, s);
}

template <class K> static int setDynLds(K k, size_t bytes) {
    PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    return PLASSHIP_OK;
}
// workgroup geometry of the partition kernel (measured, tools/linepart_bench.hip on 2^30 records): 1024 threads x 4 records with
// the next tile prefetched for 1024 buckets (one workgroup per CU: 4.8 TB/s read + written at level 1), 512 x 4 with prefetch
// when two workgroups fit a CU (<= 512 buckets: 4.9 TB/s)
template <bool NUCL, bool LONG, int MODE, bool LIST, bool EXTRAS>
static int launchLinePart(plasship_ctx *ctx, const LinePartArgs &a, uint64_t nPiecesBound) {
    const size_t lds = linePartLdsBytes(a.nb, sizeof(Rec<LONG>), EXTRAS);
    const bool big = lds > 72 * 1024;
    const unsigned grid = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>(nPiecesBound, (uint64_t) ctx->numCU * (big ? 1u : 2u)));
    if (big) {
        auto k = linePartKernel<NUCL, LONG, MODE, LIST, EXTRAS, 1024, 4, true>;
        const int rc = setDynLds(k, lds); if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, ctx->stream, a);
    } else {
        auto k = linePartKernel<NUCL, LONG, MODE, LIST, EXTRAS, 512, 4, true>;
        const int rc = setDynLds(k, lds); if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, ctx->stream, a);
    }
    return PLASSHIP_OK;
}

// line lists of one partition level over `capLines` output lines: list[start[b] .. start[b + 1]) = lines of bucket b
static int buildLineLists(plasship_ctx *ctx, const uint32_t *dTags, uint64_t capLines, uint32_t nb, uint32_t *dCount, uint32_t *dStart, uint32_t *dCursor, uint32_t *dList) {
    hipStream_t st = ctx->stream;
    PH_CHECK(hipMemsetAsync(dCount, 0, (size_t) nb * 4, st));
    const unsigned g = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((capLines + 4095) / 4096, (uint64_t) ctx->numCU * 8));
    const unsigned gs = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((capLines + TS_CHUNK - 1) / TS_CHUNK, (uint64_t) ctx->numCU * 2));
    hipLaunchKernelGGL(tagHistKernel, dim3(g), dim3(256), 0, st, dTags, capLines, nb, dCount);
    hipLaunchKernelGGL(tagScanKernel, dim3(1), dim3(1024), 0, st, (const uint32_t *) dCount, nb, dStart, dCursor);
    hipLaunchKernelGGL(tagScatterKernel, dim3(gs), dim3(TS_BLOCK), 0, st, dTags, capLines, nb, dCursor, dList);
    return PLASSHIP_OK;
}

// Sharded run, how the k-mer records reach the rank that owns their level-1 bucket:
//   exchange (rounds 1-4): a rank extracts 1/W of the sequences and ships the level-1 lines of the other ranks' buckets (all-to-all);
//   owner-filtered (round 5; the reference's own MPI scheme — every rank scans all sequences and keeps its hash range,
//   kmermatcher.cpp:312,736-778): every rank extracts ALL sequences (the DB is replicated anyway) and level 1 drops what it does not
//   own — no exchange 1 at all.  Extraction is not sharded then (75 of 350 ms per iteration at 50 M reads) but nothing crosses the
//   links: by the cost model of DESIGN.md section 6 the better choice up to 4 ranks, where one xGMI link per pair carries the exchange.
// PLASSHIP_TUNE_SHARD_EXTRACT: 1 = exchange, 2 = owner-filtered, unset = by the number of ranks.
static bool shardOwnerFiltered(int W) { const int m = tuneInt("SHARD_EXTRACT", 0); return m == 2 || (m != 1 && W <= 4); }

constexpr int KM_RETRY_EARLY_OVERFLOW_CHECK = -1000;     // internal: kmermatchImpl is to be called again, waiting for the extraction's overflow count
// What the line path hands to the run reduction
struct LinesOut { void *triples = nullptr; uint64_t nTriples = 0, Nk = 0, Nm = 0; std::vector<int64_t> stalePos; uint32_t staleT = 0; float msSort1 = 0, msGroup = 0, msSort2 = 0, msPart = 0; int nPart = 1;
                  uint64_t exchangedRecordBytes = 0, exchangedTripleBytes = 0; };

constexpr uint64_t HALO_SLACK = 1u << 16;

// ---- sort #2 over the line store: range partition of `in` (the level-1 pieces `hp` with `outLine` output lines in total) by ranges of
// the bit-reversed (rep - repBase), aggregation + sort per bucket, then every representative's triples to their place in
// representative order.  TRIPLES: `in` holds weighted triples (owner side of a sharded run), else grouped records.
// out: `dOut` = nTriples triples in (rep, target, diagonal) order (+ slackTriples of room behind them); dRepStart[nReps + 1] (optional)
template <bool NUCL, bool LONG, bool TRIPLES, bool ORDOUT = false>
static int repSortLines(plasship_ctx *ctx, const void *in, const std::vector<std::pair<uint64_t, uint64_t>> &segs, uint64_t nIn, uint32_t nReps, uint32_t repBase, uint32_t nTargets,
                        uint64_t slackTriples, const std::function<void()> &inputConsumed, DevBuf &dOut, uint64_t &nTriples, DevBuf *dRepStartOut) {
    constexpr bool PL = TRIPLES ? NUCL : LONG;                 // record layout the partition kernels move: triples are 16 bytes like Rec<false>, nucleotide triples (TripleX) 24 like Rec<true>
    typedef Rec<PL> R;
    typedef typename std::conditional<ORDOUT, TripleX, Triple>::type OutT;      // ORDOUT (sharded nucleotide run, before exchange 2): triples with their rank word
    hipStream_t st = ctx->stream;
    const int numCU = ctx->numCU;
    const int maxBits = PL ? 9 : 10;
    const int idBits = std::max(1, ceilLog2((uint64_t) nTargets));
    const int repBits = std::max(1, ceilLog2((uint64_t) std::max<uint32_t>(nReps, 1)));
    const int wantBits = std::min(2 * maxBits, std::max(0, ceilLog2((nIn + 511) / 512)));    // ~512 records per sort bucket
    const int allowedLocal = 62 - idBits - DiagPack<LONG>::BITS;                             // packed sort key = [rep - bucketBase | target | diagonal | strand] must fit 63 bits
    const int sBits = std::max(std::min(wantBits, repBits), std::max(0, repBits - allowedLocal));
    if (sBits > 2 * maxBits) { setError("kmermatch: too many sequences for the packed rep-sort key"); return PLASSHIP_ERR_UNSUPPORTED; }
    const int s1 = sBits <= maxBits ? sBits : (sBits + 1) / 2, s2 = sBits - s1;
    const uint32_t nS1 = 1u << s1, nS2 = s2 ? 1u << s2 : 0u, nSort = 1u << sBits;
    // level-1 pieces: `segs` = dense runs (first line, elements) of `in` — the group kernel's arenas, or one run of received triples;
    // the table is built here (a few thousand entries at most)
    std::vector<LinePiece> hp; uint64_t outLine = 0;
    {
        uint64_t totLines = 0; for (const auto &sg : segs) totLines += (sg.second + RPL - 1) / RPL;
        const uint32_t PLr1 = pieceLinesFor(totLines, nS1, numCU, 8);
        for (const auto &sg : segs) {
            const uint64_t cnt = sg.second, lines = (cnt + RPL - 1) / RPL;
            for (uint64_t l0 = 0; l0 < lines; l0 += PLr1) {
                LinePiece pc; pc.in0 = sg.first + l0; pc.nLines = (uint32_t) std::min<uint64_t>(PLr1, lines - l0);
                pc.lastValid = (l0 + pc.nLines == lines) ? (uint32_t) (cnt - (lines - 1) * RPL) : (uint32_t) RPL;
                pc.out0 = outLine; pc.outCap = pc.nLines + nS1; pc.tagBase = 0;
                outLine += pc.outCap; hp.push_back(pc);
            }
        }
    }
    // level 2 of a RANGE partition: representatives are not evenly spread over the id range (a contig is the representative of
    // everything it overlaps), so a level-1 bucket is cut into pieces like any other input (hash buckets are even: one piece each)
    const uint32_t PLr2 = s2 ? pieceLinesFor(std::max<uint64_t>(outLine, 1), nS2, numCU, 16) : 0;
    const uint64_t maxPR2 = s2 ? std::max<uint64_t>(outLine, 1) / PLr2 + nS1 + 1 : 0;
    const uint64_t capR1 = std::max<uint64_t>(outLine, 1), capR2 = s2 ? capR1 + maxPR2 * nS2 : 0;
    const uint32_t nPR1 = (uint32_t) hp.size();
    DevBuf dR1, dRTag1, dRList1, dRPieces, dRNP, dRCnt1, dRStart1, dRCur1, dR2, dRTag2, dRList2, dRPieces2, dRNP2, dRRegBeg, dRRegEnd, dRTot2, dSortBeg, dSortCnt;
    if (dR1.alloc(capR1 * RPL * sizeof(R)) != hipSuccess || dRTag1.alloc(capR1 * 4) != hipSuccess || dRList1.alloc(capR1 * 4) != hipSuccess || dRPieces.alloc(((size_t) nPR1 + 1) * sizeof(LinePiece)) != hipSuccess ||
        dRNP.alloc(4) != hipSuccess || dRCnt1.alloc(LP_MAXB * 4) != hipSuccess || dRStart1.alloc((LP_MAXB + 1) * 4) != hipSuccess || dRCur1.alloc(LP_MAXB * 4) != hipSuccess ||
        dSortBeg.alloc((size_t) nSort * 4) != hipSuccess || dSortCnt.alloc((size_t) nSort * 4) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
    // the piece table travels from pinned memory when it fits (no wait for the copy), else from the vector (waited for below)
    void *hpPinned = nPR1 ? ctxPinnedTable(ctx, (size_t) nPR1 * sizeof(LinePiece)) : nullptr;
    if (hpPinned) memcpy(hpPinned, hp.data(), (size_t) nPR1 * sizeof(LinePiece));
    if (nPR1) PH_CHECK(hipMemcpyAsync(dRPieces.p, hpPinned ? hpPinned : (const void *) hp.data(), (size_t) nPR1 * sizeof(LinePiece), hipMemcpyHostToDevice, st));
    else PH_CHECK(hipMemsetAsync(dRTag1.p, 0xFF, capR1 * 4, st));                       // nothing to sort: no piece will write the tag array
    PH_CHECK(hipMemcpyAsync(dRNP.p, &nPR1, 4, hipMemcpyHostToDevice, st));
    // grouped records arrive with many records per representative: lines that complete inside a tile are written directly (linepart.hpp)
    static const int directLines = [] { const char *e = getenv("PLASSHIP_DIRECT_LINES"); return e ? atoi(e) : 1; }();
    LineKey rkey; rkey.rangeBits = repBits; rkey.repBase = repBase; rkey.shift = s1 ? 64 - s1 : 63; rkey.scrambleBits = repBits;      // ranges of the bit-reversed id
    int rc;
    {
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = in; a.out = dR1.p; a.tags = dRTag1.as<uint32_t>(); a.pieces = dRPieces.as<LinePiece>(); a.nPieces = dRNP.as<uint32_t>(); a.nb = nS1; a.key = rkey;
        a.direct = directLines;
        rc = launchLinePart<NUCL, PL, KEY_RANGE, false, false>(ctx, a, std::max<uint32_t>(nPR1, 1)); if (rc) return rc;
    }
    if (nPR1 && !hpPinned) PH_CHECK(plasship::streamSync(st));   // hp goes out of use (async copy of a pageable host vector)
    rc = buildLineLists(ctx, dRTag1.as<uint32_t>(), capR1, nS1, dRCnt1.as<uint32_t>(), dRStart1.as<uint32_t>(), dRCur1.as<uint32_t>(), dRList1.as<uint32_t>()); if (rc) return rc;
    inputConsumed();                                        // the caller's input buffer is dead (stream order): it may release it
    const void *sortRecs = dR1.p; const uint32_t *sortList = dRList1.as<uint32_t>(); uint64_t sortCap = capR1;
    if (s2) {
        if (dR2.alloc(capR2 * RPL * sizeof(R)) != hipSuccess || dRTag2.alloc(capR2 * 4) != hipSuccess || dRList2.alloc(capR2 * 4) != hipSuccess || dRPieces2.alloc(((size_t) maxPR2 + 1) * sizeof(LinePiece)) != hipSuccess ||
            dRNP2.alloc(4) != hipSuccess || dRRegBeg.alloc(LP_MAXB * 8) != hipSuccess || dRRegEnd.alloc(LP_MAXB * 8) != hipSuccess || dRTot2.alloc(8) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(planListKernel, dim3(1), dim3(1024), 0, st, (const uint32_t *) dRStart1.as<uint32_t>(), nS1, PLr2, nS2, dRPieces2.as<LinePiece>(), dRNP2.as<uint32_t>(),
                           dRRegBeg.as<uint64_t>(), dRRegEnd.as<uint64_t>(), dRTot2.as<uint64_t>());
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = dR1.p; a.list = dRList1.as<uint32_t>(); a.out = dR2.p; a.tags = dRTag2.as<uint32_t>(); a.pieces = dRPieces2.as<LinePiece>(); a.nPieces = dRNP2.as<uint32_t>(); a.nb = nS2;
        a.direct = directLines;
        a.key = rkey; a.key.shift = 64 - s1 - s2;
        rc = launchLinePart<NUCL, PL, KEY_RANGE, true, false>(ctx, a, maxPR2); if (rc) return rc;
        hipLaunchKernelGGL(tagSortRegionKernel, dim3(std::min<uint32_t>(nS1, (uint32_t) numCU * 4)), dim3(512), 0, st, (const uint32_t *) dRTag2.as<uint32_t>(), (const uint64_t *) dRRegBeg.as<uint64_t>(),
                           (const uint64_t *) dRRegEnd.as<uint64_t>(), nS1, nS2, dRList2.as<uint32_t>(), dSortBeg.as<uint32_t>(), dSortCnt.as<uint32_t>());
        sortRecs = dR2.p; sortList = dRList2.as<uint32_t>(); sortCap = capR2;
    } else {
        hipLaunchKernelGGL(listRangesKernel, dim3(4), dim3(256), 0, st, (const uint32_t *) dRStart1.as<uint32_t>(), nS1, dSortBeg.as<uint32_t>(), dSortCnt.as<uint32_t>());
    }
    // aggregate + sort each bucket; buckets beyond the LDS capacity use HBM scratch
    DevBuf dBigNeed, dBigOff, dBigScratch, dUnique, dScanTmp3, dSparse;
    const size_t scanTmp3Bytes = exclusiveScanTmpBytes((size_t) nSort + 2);
    if (dBigNeed.alloc(((size_t) nSort + 1) * 8) != hipSuccess || dBigOff.alloc(((size_t) nSort + 2) * 8) != hipSuccess || dUnique.alloc(((size_t) nSort + 1) * 4) != hipSuccess ||
        dScanTmp3.alloc(scanTmp3Bytes) != hipSuccess || dSparse.alloc(sortCap * RPL * sizeof(OutT)) != hipSuccess) {
        setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE;
    }
    // pass 1: every bucket aggregated in LDS (no scratch); pass 2: the few buckets with more distinct triples than LDS holds
    const AggLines aggLn{sortList, dSortBeg.as<uint32_t>(), dSortCnt.as<uint32_t>()};
    const unsigned aggGrid = std::min<uint32_t>(nSort, (uint32_t) numCU * (uint32_t) tuneInt("AGGSORT", 16));
    hipLaunchKernelGGL((aggSortKernel<NUCL, LONG, true, TRIPLES, ORDOUT>), dim3(aggGrid), dim3(LS_BLOCK), 0, st, sortRecs, dSparse.p, (const uint64_t *) nullptr, nSort,
                       (unsigned long long *) nullptr, (const uint64_t *) nullptr, dUnique.as<uint32_t>(), repBits - sBits, idBits, (uint64_t) repBase, aggLn, repBits);
    hipLaunchKernelGGL(bigNeedKernel, dim3(gridFor(nSort, 256, 1024)), dim3(256), 0, st, (const uint32_t *) dSortCnt.as<uint32_t>(), (const uint32_t *) dUnique.as<uint32_t>(), nSort, dBigNeed.as<uint64_t>(), NUCL ? 3u : 2u);
    if (exclusiveScanU64(st, dBigNeed.as<uint64_t>(), dBigOff.as<uint64_t>(), nSort, dScanTmp3.p, scanTmp3Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t bigTot = 0;
    PH_COPY_SYNC(st, &bigTot, dBigOff.as<uint64_t>() + nSort, 8, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    if (bigTot) {
        if (dBigScratch.alloc(bigTot * 8) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL((aggSortKernel<NUCL, LONG, true, TRIPLES, ORDOUT>), dim3(aggGrid), dim3(LS_BLOCK), 0, st, sortRecs, dSparse.p, (const uint64_t *) nullptr, nSort,
                           dBigScratch.as<unsigned long long>(), (const uint64_t *) dBigOff.as<uint64_t>(), dUnique.as<uint32_t>(), repBits - sBits, idBits, (uint64_t) repBase, aggLn, repBits);
    }
    // every bucket now holds its representatives' triples, each representative's contiguous and in (target, diagonal) order, but
    // the buckets are ranges of the bit-reversed id: count the triples per representative, prefix-sum over the ids, and move every
    // triple to its place in id order — the (rep, target, diagonal)-sorted array the run reduction walks
    DevBuf dRepCnt, dRepStartLocal, dScanTmp4;
    DevBuf &dRepStart = dRepStartOut ? *dRepStartOut : dRepStartLocal;
    const size_t scanTmp4Bytes = exclusiveScanTmpBytes((size_t) nReps + 2);
    if (dRepCnt.alloc(((size_t) nReps + 1) * 4) != hipSuccess || dRepStart.alloc(((size_t) nReps + 2) * 8) != hipSuccess ||
        dScanTmp4.alloc(scanTmp4Bytes) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dRepCnt.p, 0, ((size_t) nReps + 1) * 4, st));
    const unsigned runGrid = std::min<uint32_t>((nSort + 3) / 4, (uint32_t) numCU * 8);
    hipLaunchKernelGGL(repRunsKernel<OutT>, dim3(runGrid), dim3(256), 0, st, (const OutT *) dSparse.p, (const uint32_t *) dSortBeg.as<uint32_t>(),
                       (const uint32_t *) dUnique.as<uint32_t>(), nSort, repBase, dRepCnt.as<uint32_t>());
    if (exclusiveScanU32(st, dRepCnt.as<uint32_t>(), dRepStart.as<uint64_t>(), nReps, dScanTmp4.p, scanTmp4Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    nTriples = 0;
    PH_COPY_SYNC(st, &nTriples, dRepStart.as<uint64_t>() + nReps, 8, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    dR1.release(); dR2.release();
    if (dOut.alloc((std::max<uint64_t>(nTriples, 1) + slackTriples) * sizeof(OutT)) != hipSuccess) { setError("kmermatch: out of device memory for the sorted triples"); return PLASSHIP_ERR_DEVICE; }
    if (nTriples) hipLaunchKernelGGL(placeRunsKernel<OutT>, dim3(runGrid), dim3(256), 0, st, (const OutT *) dSparse.p, (const uint32_t *) dSortBeg.as<uint32_t>(), (const uint32_t *) dUnique.as<uint32_t>(), nSort,
                                     repBase, (const uint64_t *) dRepStart.as<uint64_t>(), (OutT *) dOut.p);
    PH_CHECK(hipGetLastError());
    return PLASSHIP_OK;
}

static void moveBuf(DevBuf &dst, DevBuf &src) { dst.release(); dst.p = src.p; dst.bytes = src.bytes; src.p = nullptr; src.bytes = 0; }

// extraction has filled dA (`total` record slots of this rank's sequences, sentinels in unused slots).  Buffers dA / dB hold geo.cap2
// resp. geo.cap1 lines (single GPU) or geo.cap1 lines each (sharded run).
// Sharded run (commOf(ctx) != nullptr; `totalAll` = slots of all ranks): the bucket geometry is that of the WHOLE run, rank r owns the
// level-1 buckets [ceil(r nb1 / W), ceil((r+1) nb1 / W)).  Exchange 1 ships the level-1 LINES of the other ranks' buckets (gathered by
// destination through the line list: one extra pass over this rank's records), the receiver builds a line list over what arrived and
// runs level 2 and the group kernel exactly as a single GPU does.  Exchange 2 ships AGGREGATED (rep, target, diagonal, count) triples —
// every rank first runs the whole rep sort on the grouped records of its own buckets — and the owner of a representative merges the
// triples of all ranks with the same kernels (aggSortKernel<TRIPLES>).
template <bool NUCL, bool LONG>
static int kmermatchLines(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par, const LineGeo &geo, uint64_t total,
                          DevBuf &dA, DevBuf &dB, const DevBuf &dSlotOff, const DevBuf &dKStats, const ExtractArgs &ea, int keyBits, LinesOut &res,
                          const uint32_t *dLateOverflow, bool filtered) {
    typedef Rec<LONG> R;
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const int numCU = ctx->numCU;
    const plasship_comm *cm = commOf(ctx);
    const int W = cm ? cm->world : 1, rk = cm ? cm->rank : 0;
    const bool exch = cm && !filtered;                       // exchange 1 happens (see shardOwnerFiltered); otherwise the buffers are a single GPU's
    // stage boundaries are events read when the call is over (kmermatchImpl): nobody waits for the stream just to time a stage
    const int valueShift = std::max(0, keyBits - 11);         // VH_BINS = 2^11 monotone bins
    // ---- hash partition (replaces sort #1): level 1 over the slot array, level 2 over every level-1 bucket's line list ----
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    const uint32_t bLo = cm ? (uint32_t) ownedBegin(geo.nb1, rk, W) : 0u, bHi = cm ? (uint32_t) ownedBegin(geo.nb1, rk + 1, W) : geo.nb1;
    const uint32_t nbL = bHi - bLo;                           // level-1 buckets this rank groups (all of them on a single GPU)
    DevBuf dVHist, dMinKey, dTag1, dList1, dCnt1, dStart1, dCur1, dTag2, dList2, dPieces2, dNP2, dRegBeg, dRegEnd, dTot2, dFineBeg, dFineCnt;
    DevBuf dRx, dRxList, dRxSegs, dRxStart;                   // sharded run: received lines, their list, per-bucket list offsets
    const uint32_t nBuckets = geo.nb2 ? nbL * geo.nb2 : nbL;
    if (dVHist.alloc(VH_BINS * 4) != hipSuccess || dMinKey.alloc(8) != hipSuccess || dTag1.alloc(geo.cap1 * 4) != hipSuccess || dList1.alloc(geo.cap1 * 4) != hipSuccess ||
        dCnt1.alloc(LP_MAXB * 4) != hipSuccess || dStart1.alloc((LP_MAXB + 1) * 4) != hipSuccess || dCur1.alloc(LP_MAXB * 4) != hipSuccess ||
        dFineBeg.alloc((size_t) std::max<uint32_t>(nBuckets, 1) * 4) != hipSuccess || dFineCnt.alloc((size_t) std::max<uint32_t>(nBuckets, 1) * 4) != hipSuccess ||
        (geo.nb2 && (dPieces2.alloc((geo.maxP2 + 1) * sizeof(LinePiece)) != hipSuccess ||
                     dNP2.alloc(4) != hipSuccess || dRegBeg.alloc(LP_MAXB * 8) != hipSuccess || dRegEnd.alloc(LP_MAXB * 8) != hipSuccess || dTot2.alloc(8) != hipSuccess))) {
        setError("kmermatch: out of device memory for the line lists"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipMemsetAsync(dVHist.p, 0, VH_BINS * 4, st));
    PH_CHECK(hipMemsetAsync(dMinKey.p, 0xFF, 8, st));
    if (geo.nP1 == 0) PH_CHECK(hipMemsetAsync(dTag1.p, 0xFF, geo.cap1 * 4, st));       // no piece will write the (one-line) tag array
    {
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = dA.p; a.out = dB.p; a.tags = dTag1.as<uint32_t>(); a.totalLines = geo.totalLines; a.lastValidAll = geo.lastValid; a.pieceLines = geo.PL1; a.nb = geo.nb1;
        a.key.shift = geo.b1 ? 64 - geo.b1 : 63; a.key.rangeBits = 0; a.key.repBase = 0;
        a.minKey = NUCL ? dMinKey.as<unsigned long long>() : nullptr; a.valueHist = dVHist.as<uint32_t>(); a.valueShift = valueShift;
        if (cm && filtered) { a.keepLo = bLo; a.keepHi = bHi; }      // every rank has extracted everything: keep the records of the own buckets
        PH_CHECK(hipEventRecord(ctx->ev[8], st));
        const int rc = launchLinePart<NUCL, LONG, KEY_HASH, false, true>(ctx, a, geo.nP1); if (rc) return rc;
        PH_CHECK(hipEventRecord(ctx->ev[9], st));
    }
    int rc = buildLineLists(ctx, dTag1.as<uint32_t>(), geo.cap1, geo.nb1, dCnt1.as<uint32_t>(), dStart1.as<uint32_t>(), dCur1.as<uint32_t>(), dList1.as<uint32_t>()); if (rc) return rc;
    // what level 2 (or, with a single level, the group kernel) reads: records, their line list, the list offsets of the nbL buckets
    void *l1Recs = dB.p; const uint32_t *l1List = dList1.as<uint32_t>(); const uint32_t *l1Start = dStart1.as<uint32_t>() + bLo; uint64_t l1Lines = geo.cap1;
    const uint32_t *l1Tags = dTag1.as<uint32_t>();
    uint64_t NkAll = 0;
    if (cm && filtered && NUCL) {              // the globally smallest key (first-run quirk of the group kernel): every rank saw its own buckets' records
        uint64_t mk = 0; PH_COPY_SYNC(st, &mk, dMinKey.p, 8, hipMemcpyDeviceToHost);
        rc = commAllReduceMinU64(ctx, &mk, 1); if (rc) return rc;
        PH_COPY_SYNC(st, dMinKey.p, &mk, 8, hipMemcpyHostToDevice);
    }
    if (exch) {
        // ---- exchange 1: the lines of every level-1 bucket to the bucket's owner ----
        std::vector<uint32_t> hStart1(geo.nb1 + 1);
        PH_COPY_SYNC(st, hStart1.data(), dStart1.p, ((size_t) geo.nb1 + 1) * 4, hipMemcpyDeviceToHost);
        PH_CHECK(hipGetLastError());
        const uint64_t myLines = hStart1[geo.nb1];
        // lines in list order = by bucket = by destination.  The lines of the OTHER ranks' buckets are packed into a send buffer of
        // exactly their size (the slot array is consumed: it goes first, the receive buffer will need the room); the lines of this
        // rank's own buckets never pass through it: they are gathered straight into the receive buffer, behind what the others sent
        // (one pass over them instead of a gather and a device-to-device copy — all of the data in a 1-rank group, 1/W of it otherwise).
        const uint64_t selfBeg = hStart1[ownedBegin(geo.nb1, rk, W)], selfEnd = hStart1[ownedBegin(geo.nb1, rk + 1, W)];
        const uint64_t selfLines = selfEnd - selfBeg, sendLines = myLines - selfLines;
        const uint32_t chunksPerLine = (uint32_t) (RPL * sizeof(R) / 16);
        auto gather = [&](uint64_t listFrom, uint64_t n, void *dst) {
            if (n) hipLaunchKernelGGL(gatherLinesKernel, dim3(gridFor(n * chunksPerLine, 256, (unsigned) numCU * 16)), dim3(256), 0, st, (const uint4 *) dB.p, (const uint32_t *) dList1.as<uint32_t>() + listFrom,
                                      n, chunksPerLine, (uint4 *) dst);
        };
        dA.release();
        if (dA.alloc(std::max<uint64_t>(sendLines, 1) * RPL * sizeof(R)) != hipSuccess) { setError("kmermatch: out of device memory for the send buffer"); return PLASSHIP_ERR_DEVICE; }
        gather(0, selfBeg, dA.p);
        gather(selfEnd, myLines - selfEnd, (char *) dA.p + selfBeg * RPL * sizeof(R));
        dTag1.release();
        // per-bucket line counts of every rank + the records this rank extracted
        std::vector<uint64_t> mine((size_t) geo.nb1 + 1), all(((size_t) geo.nb1 + 1) * (size_t) W);
        for (uint32_t j = 0; j < geo.nb1; j++) mine[j] = hStart1[j + 1] - hStart1[j];
        { unsigned long long ks[4] = {0, 0, 0, 0}; PH_COPY_SYNC(st, ks, dKStats.p, 32, hipMemcpyDeviceToHost); mine[geo.nb1] = ks[1] + ks[3]; }
        rc = commAllgatherHost(ctx, mine.data(), all.data(), mine.size() * 8); if (rc) return rc;
        std::vector<uint64_t> sendCount(W);
        for (int r = 0; r < W; r++) { sendCount[r] = (r == rk) ? 0 : hStart1[ownedBegin(geo.nb1, r + 1, W)] - hStart1[ownedBegin(geo.nb1, r, W)]; NkAll += all[(size_t) r * mine.size() + geo.nb1]; }
        uint64_t gotOthers = 0;
        // (room behind the received lines: this rank's own lines, and the group kernel's arenas, which are addressed by level-2 line
        // numbers, up to nbL * nb2 beyond)
        rc = commAlltoallvRecords(ctx, dA.p, sendCount.data(), RPL * sizeof(R), dRx, &gotOthers, selfLines + (uint64_t) nbL * geo.nb2 + 1); if (rc) return rc;
        res.exchangedRecordBytes = sendLines * RPL * sizeof(R);
        PH_TRACE(st, "kmermatch: exchange 1 (level-1 lines)");
        dA.release();
        gather(selfBeg, selfLines, (char *) dRx.p + gotOthers * RPL * sizeof(R));
        dB.release(); dList1.release();    // (stream order: the gathers have read them before anything reuses the memory)
        const uint64_t gotLines = gotOthers + selfLines;
        if (gotLines >= 0xFFFFFFFFull) { rc = PLASSHIP_ERR_UNSUPPORTED; setError("kmermatch: more than 2^32 lines on one rank"); }
        rc = commAgreeOk(ctx, rc == 0, "kmermatch: more than 2^32 lines on one rank"); if (rc) return rc;
        // what arrived: from every source s the lines of my buckets bLo .. bHi-1, bucket after bucket.  List: bucket-major, source-minor.
        std::vector<RxSeg> segs((size_t) nbL * W); std::vector<uint32_t> rxStart((size_t) nbL + 1);
        // (in the receive buffer: the other ranks' lines in rank order, then this rank's own)
        std::vector<uint64_t> srcBase(W); { uint64_t o = 0; auto linesFrom = [&](int s) { uint64_t c = 0; for (uint32_t j = bLo; j < bHi; j++) c += all[(size_t) s * mine.size() + j]; return c; };
          for (int s = 0; s < W; s++) if (s != rk) { srcBase[s] = o; o += linesFrom(s); }
          if (o != gotOthers || linesFrom(rk) != selfLines) { setError("kmermatch: internal error, exchanged line counts do not add up"); return PLASSHIP_ERR_DEVICE; }
          srcBase[rk] = o; }
        { uint64_t d = 0; std::vector<uint64_t> run(srcBase);
          for (uint32_t j = 0; j < nbL; j++) { rxStart[j] = (uint32_t) d; for (int s = 0; s < W; s++) { const uint64_t c = all[(size_t) s * mine.size() + bLo + j]; segs[(size_t) j * W + s] = RxSeg{run[s], (uint32_t) c, (uint32_t) d}; run[s] += c; d += c; } }
          rxStart[nbL] = (uint32_t) d; }
        if (dRxList.alloc(std::max<uint64_t>(gotLines, 1) * 4) != hipSuccess || dRxSegs.alloc(std::max<size_t>(segs.size(), 1) * sizeof(RxSeg)) != hipSuccess || dRxStart.alloc(((size_t) nbL + 1) * 4) != hipSuccess) { setError("kmermatch: out of device memory for the received line list"); return PLASSHIP_ERR_DEVICE; }
        PH_COPY_SYNC(st, dRxSegs.p, segs.data(), segs.size() * sizeof(RxSeg), hipMemcpyHostToDevice);
        PH_COPY_SYNC(st, dRxStart.p, rxStart.data(), ((size_t) nbL + 1) * 4, hipMemcpyHostToDevice);
        if (!segs.empty()) hipLaunchKernelGGL(rxListKernel, dim3((unsigned) std::min<size_t>(segs.size(), (size_t) numCU * 8)), dim3(256), 0, st, (const RxSeg *) dRxSegs.p, (uint32_t) segs.size(), dRxList.as<uint32_t>());
        if (NUCL) {              // the globally smallest key (first-run quirk of the group kernel)
            uint64_t mk = 0; PH_COPY_SYNC(st, &mk, dMinKey.p, 8, hipMemcpyDeviceToHost);
            rc = commAllReduceMinU64(ctx, &mk, 1); if (rc) return rc;
            PH_COPY_SYNC(st, dMinKey.p, &mk, 8, hipMemcpyHostToDevice);
        }
        l1Recs = dRx.p; l1List = dRxList.as<uint32_t>(); l1Start = dRxStart.as<uint32_t>(); l1Lines = gotLines; l1Tags = nullptr;
    }
    void *finalRecs = l1Recs; const uint32_t *finalTags = l1Tags, *finalList = l1List; uint64_t finalCap = l1Lines;
    DevBuf dL2;                                               // sharded run: level-2 output (a single GPU writes level 2 into dA)
    res.nPart = 1;
    if (geo.nb2) {
        const uint64_t cap2 = exch ? l1Lines + (uint64_t) nbL * geo.nb2 : geo.cap2;
        void *l2Out = dA.p;
        if (exch) { if (dL2.alloc(std::max<uint64_t>(cap2, 1) * RPL * sizeof(R)) != hipSuccess) { setError("kmermatch: out of device memory for the k-mer record arrays"); return PLASSHIP_ERR_DEVICE; } l2Out = dL2.p; }
        if (dTag2.alloc(std::max<uint64_t>(cap2, 1) * 4) != hipSuccess || dList2.alloc(std::max<uint64_t>(cap2, 1) * 4) != hipSuccess) { setError("kmermatch: out of device memory for the line lists"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(planListKernel, dim3(1), dim3(1024), 0, st, l1Start, nbL, geo.PL2, geo.nb2, dPieces2.as<LinePiece>(), dNP2.as<uint32_t>(),
                           dRegBeg.as<uint64_t>(), dRegEnd.as<uint64_t>(), dTot2.as<uint64_t>());
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = l1Recs; a.list = l1List; a.out = l2Out; a.tags = dTag2.as<uint32_t>(); a.pieces = dPieces2.as<LinePiece>(); a.nPieces = dNP2.as<uint32_t>(); a.nb = geo.nb2;
        a.key.shift = 64 - geo.b1 - geo.b2;
        PH_CHECK(hipEventRecord(ctx->ev[10], st));
        rc = launchLinePart<NUCL, LONG, KEY_HASH, true, false>(ctx, a, geo.maxP2); if (rc) return rc;
        PH_CHECK(hipEventRecord(ctx->ev[11], st));
        hipLaunchKernelGGL(tagSortRegionKernel, dim3(std::min<uint32_t>(nbL, (uint32_t) numCU * 4)), dim3(512), 0, st, (const uint32_t *) dTag2.as<uint32_t>(), (const uint64_t *) dRegBeg.as<uint64_t>(),
                           (const uint64_t *) dRegEnd.as<uint64_t>(), nbL, geo.nb2, dList2.as<uint32_t>(), dFineBeg.as<uint32_t>(), dFineCnt.as<uint32_t>());
        finalRecs = l2Out; finalTags = dTag2.as<uint32_t>(); finalList = dList2.as<uint32_t>(); finalCap = cap2;
        res.nPart = 2;
    } else {
        hipLaunchKernelGGL(listRangesKernel, dim3(4), dim3(256), 0, st, l1Start, nbL, dFineBeg.as<uint32_t>(), dFineCnt.as<uint32_t>());
    }
    PH_CHECK(hipEventRecord(ctx->ev[6], st));
    PH_TRACE(st, "kmermatch: hash partition (line store)");
    PH_CHECK(hipGetLastError());

    // ---- assignGroup: every workgroup writes its grouped records into an arena that begins where its first bucket's lines begin ----

    // the arenas: the buffer level 2 read (dead now).  single GPU: dB (level 1's output) when there are two levels, else dA (the slot
    // array); sharded run: the receive buffer when there are two levels, else a buffer of its own
    DevBuf dArena;
    void *arenaBuf;
    if (exch) {
        if (geo.nb2) arenaBuf = dRx.p;
        else { if (dArena.alloc(std::max<uint64_t>(finalCap, 1) * RPL * sizeof(R)) != hipSuccess) { setError("kmermatch: out of device memory for the grouped records"); return PLASSHIP_ERR_DEVICE; } arenaBuf = dArena.p; }
    } else arenaBuf = geo.nb2 ? dB.p : dA.p;
    const uint32_t gBlocks = std::max<uint32_t>(1, std::min<uint32_t>(nBuckets, (uint32_t) numCU * (uint32_t) tuneInt("GROUP", 24)));      // (6 until round 6; with the look-ahead 24 measures 1 ms better at 50 M reads — smaller arenas for the rep sort: profiles/r06_ab_knobs.txt, call 26)
    const uint32_t bpb = (std::max<uint32_t>(nBuckets, 1) + gBlocks - 1) / gBlocks;
    const uint32_t gGrid = (std::max<uint32_t>(nBuckets, 1) + bpb - 1) / bpb;
    DevBuf dOutCnt, dArenaStart, dMaxRT, dLastRun;
    if (dOutCnt.alloc((size_t) gGrid * 8) != hipSuccess || dArenaStart.alloc((size_t) gGrid * 8) != hipSuccess || dMaxRT.alloc(8) != hipSuccess || dLastRun.alloc(32) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    GroupArgs ga; memset(&ga, 0, sizeof(ga));
    ga.in = finalRecs; ga.out = arenaBuf; ga.list = finalList; ga.lineBeg = dFineBeg.as<uint32_t>(); ga.lineCnt = dFineCnt.as<uint32_t>();
    ga.nBuckets = nBuckets; ga.bucketsPerBlock = bpb; ga.outCount = dOutCnt.as<uint64_t>();
    PH_CHECK(hipMemsetAsync(dMaxRT.p, 0, 8, st));
    ga.maxRepTarget = dMaxRT.as<unsigned long long>();
    ga.includeOnlyExtendable = par->include_only_extendable; ga.covMode = par->cov_mode; ga.covThr = par->cov_thr; ga.minKey = NUCL ? dMinKey.as<unsigned long long>() : nullptr;
    // positions per bucket (sentinel padding included) decide the workgroup shape of the 16-byte-record kernel
    const uint64_t avgPos = finalCap * RPL / std::max<uint32_t>(nBuckets, 1);
    const bool wideGroup = !LONG && (getenv("PLASSHIP_GROUP_WIDE") ? atoi(getenv("PLASSHIP_GROUP_WIDE")) != 0 : avgPos > 1600);
    if (nBuckets == 0) PH_CHECK(hipMemsetAsync(dOutCnt.p, 0, (size_t) gGrid * 8, st));
    else if constexpr (LONG) hipLaunchKernelGGL((groupKernel<NUCL, LONG, true>), dim3(gGrid), dim3(GR_BLOCK), 0, st, ga);
    else if (wideGroup && tuneInt("GROUP_WPE", 4) == 4 && tuneInt("GROUP_LAHEAD", 1) == 1) hipLaunchKernelGGL((groupLinesKernel<NUCL, 512, 4096, 4, true>), dim3(gGrid), dim3(512), 0, st, ga);      // PLASSHIP_TUNE_GROUP_LAHEAD=2: without the list look-ahead
    else if (wideGroup && tuneInt("GROUP_WPE", 4) == 4) hipLaunchKernelGGL((groupLinesKernel<NUCL, 512, 4096, 4>), dim3(gGrid), dim3(512), 0, st, ga);
    else if (wideGroup) hipLaunchKernelGGL((groupLinesKernel<NUCL, 512, 4096, 2>), dim3(gGrid), dim3(512), 0, st, ga);
    else if (tuneInt("GROUP_LAHEAD", 1) == 1) hipLaunchKernelGGL((groupLinesKernel<NUCL, GR_BLOCK, GR_HT, 3, true>), dim3(gGrid), dim3(GR_BLOCK), 0, st, ga);
    else hipLaunchKernelGGL((groupLinesKernel<NUCL, GR_BLOCK, GR_HT, 3>), dim3(gGrid), dim3(GR_BLOCK), 0, st, ga);
    if (nBuckets) hipLaunchKernelGGL(arenaStartKernel, dim3(gridFor(gGrid, 256, 64)), dim3(256), 0, st, (const uint32_t *) dFineBeg.as<uint32_t>(), bpb, gGrid, nBuckets, dArenaStart.as<uint64_t>());
    else PH_CHECK(hipMemsetAsync(dArenaStart.p, 0, (size_t) gGrid * 8, st));
    std::vector<uint64_t> hOutCnt(gGrid), hArena(gGrid);
    unsigned long long hLastRun[4] = {0, 0, 0, 0}, ks[4] = {0, 0, 0, 0}; std::vector<uint32_t> hVHist(VH_BINS);
    hipLaunchKernelGGL(lastRunInfoKernel, dim3(1), dim3(1), 0, st, dMaxRT.as<unsigned long long>(), dSlotOff.as<uint64_t>(), db->d_len.as<uint32_t>(), N, dLastRun.as<unsigned long long>());
    PH_CHECK(hipMemcpyAsync(hLastRun, dLastRun.p, 32, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(hVHist.data(), dVHist.p, VH_BINS * 4, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(hOutCnt.data(), dOutCnt.p, (size_t) gGrid * 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(hArena.data(), dArenaStart.p, (size_t) gGrid * 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(ks, dKStats.p, 32, hipMemcpyDeviceToHost, st));
    uint32_t lateOverflow = 0;
    if (dLateOverflow) PH_CHECK(hipMemcpyAsync(&lateOverflow, dLateOverflow, 4, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    if (lateOverflow) return KM_RETRY_EARLY_OVERFLOW_CHECK;          // (never in a sharded run: its ranks wait for the count right after the extraction)
    uint64_t NmLocal = 0;
    for (uint32_t j = 0; j < gGrid; j++) NmLocal += hOutCnt[j];
    uint64_t Nm = NmLocal;
    const uint64_t NkLocal = ks[1] + ks[3];                  // records the extraction kernels of this rank wrote (sentinels excluded)
    const uint64_t Nk = exch ? NkAll : NkLocal;              // ... and of the whole run (owner-filtered: every rank extracted everything)
    std::vector<uint64_t> hVHistG(hVHist.begin(), hVHist.end());
    if (cm) {
        // the stale-record check below is a property of the WHOLE run: N_m, the last (rep, target) run and the value histogram are
        // reduced over the ranks; every rank then takes the same decisions (and the same collectives)
        std::vector<uint64_t> mine(2 + (size_t) VH_BINS), all((2 + (size_t) VH_BINS) * (size_t) W);
        mine[0] = NmLocal; mine[1] = hLastRun[0]; std::copy(hVHistG.begin(), hVHistG.end(), mine.begin() + 2);
        rc = commAllgatherHost(ctx, mine.data(), all.data(), mine.size() * 8); if (rc) return rc;
        uint64_t mx = 0; Nm = 0; std::fill(hVHistG.begin(), hVHistG.end(), 0);
        for (int r = 0; r < W; r++) {
            const uint64_t *row = all.data() + (size_t) r * mine.size();
            Nm += row[0]; mx = std::max(mx, row[1]);
            for (uint32_t b = 0; b < VH_BINS; b++) hVHistG[b] += row[2 + b];
        }
        PH_CHECK(hipMemcpyAsync(dMaxRT.p, &mx, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lastRunInfoKernel, dim3(1), dim3(1), 0, st, dMaxRT.as<unsigned long long>(), dSlotOff.as<uint64_t>(), db->d_len.as<uint32_t>(), N, dLastRun.as<unsigned long long>());
        PH_COPY_SYNC(st, hLastRun, dLastRun.p, 32, hipMemcpyDeviceToHost);
    }
    res.Nk = NkLocal;                                         // sharded run: the records THIS rank extracted (the ranks' sum is the run's N_k)
    if (cm && filtered) { res.Nk = 0; for (uint32_t b = 0; b < VH_BINS; b++) res.Nk += hVHist[b]; }      // ... owner-filtered: the records it KEPT (level 1 counts each in its value histogram)
    res.Nm = NmLocal;
    PH_CHECK(hipEventRecord(ctx->ev[7], st));
    PH_TRACE(st, "kmermatch: group (line store)");

    // ---- stale records behind the compaction point that continue the last run (see section 7 above) ----
    if (Nm > 0 && Nm < Nk) {
        const unsigned long long maxRT = hLastRun[0];
        res.staleT = (uint32_t) (maxRT & 0xFFFFFFFFull);
        const uint64_t so[2] = {hLastRun[1], hLastRun[2]}; const uint32_t tLen = (uint32_t) hLastRun[3];
        const uint32_t tb = (uint32_t) (so[1] - so[0]);
        DevBuf dTRec, dTId, dTScr, dTOff, dTCap, dDiff;
        uint32_t cap = 64; while (cap < tLen + 1) cap <<= 1;
        const uint64_t zero = 0;
        if (dTRec.alloc((size_t) tb * sizeof(R)) != hipSuccess || dTId.alloc(4) != hipSuccess || dTScr.alloc((size_t) cap * sizeof(Cand)) != hipSuccess ||
            dTOff.alloc(8) != hipSuccess || dTCap.alloc(4) != hipSuccess || dDiff.alloc(((size_t) tb + 1) * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemcpyAsync(dTId.p, &res.staleT, 4, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemcpyAsync(dTOff.p, &zero, 8, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemcpyAsync(dTCap.p, &cap, 4, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemsetAsync(dDiff.p, 0, ((size_t) tb + 1) * 8, st));
        // re-extract the records of T into a scratch array with the very kernel that produced them
        ExtractArgs ta = ea; ta.waveList = nullptr; ta.waveCount = nullptr; ta.kstats = nullptr; ta.arr = dTRec.p; ta.slotBias = so[0];
        ta.idList = dTId.as<uint32_t>(); ta.nIds = 1; ta.scratch = dTScr.as<Cand>(); ta.scratchOff = dTOff.as<uint64_t>(); ta.scratchCap = dTCap.as<uint32_t>();
        hipLaunchKernelGGL((extractKernel<NUCL, LONG, 1, true>), dim3(1), dim3(64), 0, st, ta);
        std::vector<R> trec(tb);
        PH_CHECK(hipMemcpyAsync(trec.data(), dTRec.p, (size_t) tb * sizeof(R), hipMemcpyDeviceToHost, st));
        PH_CHECK(plasship::streamSync(st));
        trec.erase(std::remove_if(trec.begin(), trec.end(), [](const R &r) { return r.kmer == ~0ULL && r.id == 0xFFFFFFFFu; }), trec.end());
        std::sort(trec.begin(), trec.end(), [](const R &x, const R &y) { return recLess1<NUCL, LONG>(x, y); });
        const uint32_t m = (uint32_t) trec.size();
        // cheap exact filter first: the value histogram bounds the sort-#1 rank of every record of T; the scan can only reach
        // a record of T if rank N_m itself can be one of them
        bool mayHit = false;
        if (m) {
            std::vector<uint64_t> cum(VH_BINS + 1, 0);
            for (uint32_t b = 0; b < VH_BINS; b++) cum[b + 1] = cum[b] + hVHistG[b];
            for (uint32_t j = 0; j < m && !mayHit; j++) { const uint32_t b = valueBin<NUCL>(trec[j].kmer, valueShift); mayHit = Nm >= cum[b] && Nm < cum[b + 1]; }
        }
        if (m && mayHit) {
            PH_CHECK(hipMemcpyAsync(dTRec.p, trec.data(), (size_t) m * sizeof(R), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((rankLinesKernel<NUCL, LONG>), dim3(gridFor(finalCap * RPL, 256, (unsigned) numCU * 8)), dim3(256), 0, st, (const void *) finalRecs, finalTags, finalCap, geo.nb2 ? (const uint64_t *) dTot2.as<uint64_t>() : (const uint64_t *) nullptr,
                               (const void *) dTRec.p, m, dDiff.as<unsigned long long>());
            std::vector<unsigned long long> diff((size_t) m + 1);
            PH_CHECK(hipMemcpyAsync(diff.data(), dDiff.p, ((size_t) m + 1) * 8, hipMemcpyDeviceToHost, st));
            PH_CHECK(plasship::streamSync(st));
            if (cm) {            // every rank counted the records of its own buckets: the sort-#1 ranks are the sums
                static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
                rc = commAllReduceSumU64(ctx, reinterpret_cast<uint64_t *>(diff.data()), diff.size()); if (rc) return rc;
            }
            unsigned long long rank = 0, expect = Nm;
            for (uint32_t j = 0; j < m; j++) {
                rank += diff[j];                         // records strictly before trec[j] in sort-#1 order
                if (rank == expect) { res.stalePos.push_back((int64_t) trec[j].pos); expect++; }
                else if (rank > expect) break;
            }
        }
        PH_TRACE(st, "kmermatch: stale-record check (line store)");
    }

    // ---- sort #2: range partition of the grouped records by rep id over the line store + aggregation / sort per bucket ----
    PH_CHECK(hipEventRecord(ctx->ev[12], st));
    // the hash-bucketed records are dead now (the group kernel's arenas live in another buffer): free them for the rep side
    if (exch) { dL2.release(); if (!geo.nb2) dRx.release(); }
    else (finalRecs == dA.p ? dA : dB).release();
    dTag1.release(); dList1.release(); dTag2.release(); dList2.release(); dRxList.release();
    std::vector<std::pair<uint64_t, uint64_t>> arenas(gGrid);                 // the arenas are dense segments: (first line, records)
    for (uint32_t j = 0; j < gGrid; j++) arenas[j] = std::make_pair(hArena[j] / RPL, hOutCnt[j]);
    DevBuf dTriples, dRepStart; uint64_t nTriples = 0;
    auto arenasConsumed = [&]() { if (exch) { if (geo.nb2) dRx.release(); else dArena.release(); } else (arenaBuf == dA.p ? dA : dB).release(); };
    // (sharded nucleotide run: the triples leave for their owners with the rank word of their top member, see TripleX)
    if (NUCL && cm) rc = repSortLines<NUCL, LONG, false, NUCL>(ctx, arenaBuf, arenas, NmLocal, N, 0u, N, 0, arenasConsumed, dTriples, nTriples, &dRepStart);
    else rc = repSortLines<NUCL, LONG, false>(ctx, arenaBuf, arenas, NmLocal, N, 0u, N, 0, arenasConsumed, dTriples, nTriples, cm ? &dRepStart : nullptr);
    if (rc) return rc;
    if (cm) {
        // ---- exchange 2: every representative's aggregated triples to the representative's owner (contiguous id ranges: the triples
        //      are in id order, so a rank's share is one run), merged there with the triples of the other ranks ----
        DevBuf dOB; std::vector<uint64_t> ob((size_t) W + 1), sendCount(W);
        if (dOB.alloc(((size_t) W + 1) * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(ownerBoundsKernel, dim3(1), dim3(256), 0, st, (const uint64_t *) dRepStart.as<uint64_t>(), (uint64_t) N, W, dOB.as<uint64_t>());
        PH_COPY_SYNC(st, ob.data(), dOB.p, ((size_t) W + 1) * 8, hipMemcpyDeviceToHost);
        for (int r = 0; r < W; r++) sendCount[r] = ob[r + 1] - ob[r];
        DevBuf dRxT; uint64_t gotT = 0;
        constexpr size_t xBytes = NUCL ? sizeof(TripleX) : sizeof(Triple);
        rc = commAlltoallvRecords(ctx, dTriples.p, sendCount.data(), xBytes, dRxT, &gotT, RPL); if (rc) return rc;
        res.exchangedTripleBytes = (nTriples - sendCount[rk]) * xBytes;
        PH_TRACE(st, "kmermatch: exchange 2 (aggregated triples)");
        dTriples.release(); dRepStart.release();
        const uint32_t repBase = (uint32_t) ownedBegin(N, rk, W), ownedN = (uint32_t) (ownedBegin(N, rk + 1, W) - repBase);
        DevBuf dMerged; uint64_t nMerged = 0;
        rc = repSortLines<NUCL, LONG, true>(ctx, dRxT.p, {std::make_pair((uint64_t) 0, gotT)}, gotT, ownedN, repBase, N, HALO_SLACK, [&]() { dRxT.release(); }, dMerged, nMerged, nullptr);
        if (rc) return rc;
        moveBuf(dA, dMerged);                                 // the caller's dA owns the result
        nTriples = nMerged;
    } else moveBuf(dA, dTriples);
    PH_CHECK(hipEventRecord(ctx->ev[13], st));
    PH_TRACE(st, "kmermatch: rep sort (line store)");
    PH_CHECK(hipGetLastError());
    res.triples = dA.p; res.nTriples = nTriples;
    { float msS = 0, msS2 = 0; (void) hipEventElapsedTime(&msS, ctx->ev[8], ctx->ev[9]); if (res.nPart == 2) (void) hipEventElapsedTime(&msS2, ctx->ev[10], ctx->ev[11]); res.msPart = msS + msS2; }
    return PLASSHIP_OK;
}


// ---- best diagonal per (rep, target) run over the sorted weighted triples, CSR of the candidate list (kmermatcher.cpp:809-924) ----
// `cur`: nTriples triples in (rep, target, diagonal) order (sharded run: room for HALO_SLACK more behind them)
template <bool NUCL, bool LONG>
static int reduceToCandidates(plasship_ctx *ctx, const plasship_seqdb *db, void *cur, uint64_t nTriples, const std::vector<int64_t> &stalePos, uint32_t staleT,
                              std::unique_ptr<plasship_cands> &holder, uint64_t &Nc, float &msReduce, bool lazyClock = false) {
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    Timer tm{ctx, 0};
    const plasship_comm *cm = commOf(ctx);
    const int W = cm ? cm->world : 1, rk = cm ? cm->rank : 0;
    const uint64_t repBase = ownedBegin(N, rk, W), ownedN = ownedBegin(N, rk + 1, W) - repBase;
    // ---- per-(rep,target) reduction + CSR ----
    if (!lazyClock) tm.start(0);                             // (lazyClock: the caller recorded ev[13] and reads ev[13] .. ev[14] when the call is over)
    DevBuf dTmpHits, dEmit, dEpos, dPerRep, dQoff;
    if (dTmpHits.alloc(std::max<uint64_t>(nTriples, 1) * sizeof(CandHit)) != hipSuccess || dEmit.alloc((nTriples / 64 + 2) * 4) != hipSuccess ||
        dEpos.alloc((nTriples / 64 + 3) * 8) != hipSuccess || dPerRep.alloc(((size_t) N + 1) * 4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    DevBuf dScanTmp2; const size_t scanTmp2Bytes = exclusiveScanTmpBytes(std::max<uint64_t>(nTriples, N) + 2);
    if (dScanTmp2.alloc(scanTmp2Bytes) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    uint64_t nHalo = 0;
    if (cm) {
        // The reference's run scan tests only the target id (Appendix A.3): the last run of this rank continues into the
        // triples of the next ranks while they carry the same target, and behind the last rank into the stale records.
        // Every rank publishes the head of its triples (the leading ones with one target); each rank appends what its last
        // run can reach behind its own triples.
        Triple *dTr = reinterpret_cast<Triple *>(cur);
        std::vector<Triple> head; uint64_t headCnt = 0;
        if (nTriples) {
            uint64_t want = std::min<uint64_t>(nTriples, 1024);
            for (;;) {
                head.resize(want);
                PH_COPY_SYNC(st, head.data(), dTr, want * sizeof(Triple), hipMemcpyDeviceToHost);
                headCnt = 0; while (headCnt < want && head[headCnt].target == head[0].target) headCnt++;
                if (headCnt < want || want == nTriples) break;
                want = std::min<uint64_t>(nTriples, want * 2);
            }
            head.resize(headCnt);
        }
        Triple last; memset(&last, 0, sizeof(last));
        if (nTriples) PH_COPY_SYNC(st, &last, dTr + (nTriples - 1), sizeof(Triple), hipMemcpyDeviceToHost);
        uint64_t hdr[2] = {nTriples, headCnt}; std::vector<uint64_t> hdrs(2 * (size_t) W);
        int rc = commAllgatherHost(ctx, hdr, hdrs.data(), 16); if (rc) return rc;
        uint64_t maxHead = 0, sumHead = 0; for (int r = 0; r < W; r++) { maxHead = std::max(maxHead, hdrs[2 * (size_t) r + 1]); sumHead += hdrs[2 * (size_t) r + 1]; }
        // the halo of any rank is at most all heads plus the stale records: decide on THAT, so every rank takes the same exit
        // (a rank that fails alone would leave the others waiting in the next collective)
        if (sumHead + stalePos.size() > HALO_SLACK) { setError("kmermatch: a (rep, target) run continues over more than 65536 records of other ranks"); return PLASSHIP_ERR_UNSUPPORTED; }
        std::vector<Triple> heads;
        if (maxHead) {
            std::vector<Triple> mine(maxHead); memset(mine.data(), 0, maxHead * sizeof(Triple));
            std::copy(head.begin(), head.end(), mine.begin());
            heads.resize(maxHead * (size_t) W);
            rc = commAllgatherHost(ctx, mine.data(), heads.data(), maxHead * sizeof(Triple)); if (rc) return rc;
        }
        if (nTriples) {
            std::vector<Triple> halo; bool open = true;          // open: the scan has not met another target yet
            for (int r = rk + 1; r < W && open; r++) {
                const uint64_t nr = hdrs[2 * (size_t) r], hr = hdrs[2 * (size_t) r + 1];
                if (nr == 0) continue;
                const Triple *hp = heads.data() + maxHead * (size_t) r;
                if (hp[0].target != last.target) { open = false; break; }
                halo.insert(halo.end(), hp, hp + hr);
                if (hr < nr) open = false;
            }
            if (open && !stalePos.empty() && staleT == last.target) {
                for (int64_t sp : stalePos) {      // stale records: pos = original k-mer position, kmer field = SIZE_T_MAX (forward)
                    Triple t; t.rep = 0xFFFFFFFFu; t.target = staleT; t.diag = LONG ? (int32_t) sp : (int32_t) (int16_t) sp; t.cnt = 1u | 0x80000000u;
                    halo.push_back(t);
                }
            }
            nHalo = halo.size();
            if (nHalo) PH_COPY_SYNC(st, dTr + nTriples, halo.data(), nHalo * sizeof(Triple), hipMemcpyHostToDevice);
        }
        PH_CHECK(hipMemsetAsync(dPerRep.p, 0, ((size_t) N + 1) * 4, st));
        if (ownedN) hipLaunchKernelGGL(fillU32Kernel, dim3(gridFor(ownedN, 256, 4096)), dim3(256), 0, st, dPerRep.as<uint32_t>() + repBase, 1u, ownedN);
    } else
    hipLaunchKernelGGL(fillU32Kernel, dim3(gridFor((uint64_t) N + 1, 256, 4096)), dim3(256), 0, st, dPerRep.as<uint32_t>(), 1u, (uint64_t) N);
    if (nTriples) hipLaunchKernelGGL((reduceRunsKernel<NUCL>), dim3(gridFor(nTriples, 256, 65535)), dim3(256), 0, st, (const Triple *) cur, nTriples, nTriples + nHalo, dTmpHits.as<CandHit>(), dEmit.as<uint32_t>(), dPerRep.as<uint32_t>());
    const uint64_t nWaves = (nTriples + 63) / 64;              // candidates are counted per 64 triples (reduceRunsKernel)
    if (exclusiveScanU32(st, dEmit.as<uint32_t>(), dEpos.as<uint64_t>(), nWaves, dScanTmp2.p, scanTmp2Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    holder.reset(new plasship_cands());                              // released to the caller on success only
    plasship_cands *c = holder.get();
    c->reverseCapable = NUCL; c->nQueries = N;
    if (c->d_qoff.alloc(((size_t) N + 1) * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (exclusiveScanU32(st, dPerRep.as<uint32_t>(), c->d_qoff.as<uint64_t>(), N, dScanTmp2.p, scanTmp2Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    Nc = 0;
    PH_CHECK(hipMemcpyAsync(&Nc, dEpos.as<uint64_t>() + nWaves, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    const uint32_t qLo = cm ? (uint32_t) repBase : 0u, qHi = cm ? (uint32_t) (repBase + ownedN) : N;      // queries with a self line
    c->nHits = Nc + (qHi - qLo); c->nNonSelf = Nc;
    if (c->d_hits.alloc(std::max<uint64_t>(c->nHits, 1) * sizeof(CandHit)) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (qHi > qLo) hipLaunchKernelGGL(placeSelfKernel, dim3(gridFor(qHi - qLo, 256, 4096)), dim3(256), 0, st, c->d_qoff.as<uint64_t>(), qLo, qHi, c->d_hits.as<CandHit>());
    if (nTriples) hipLaunchKernelGGL(placeHitsKernel, dim3(gridFor(nTriples, 256, 65535)), dim3(256), 0, st, dTmpHits.as<CandHit>(), dEpos.as<uint64_t>(), nTriples, qLo, c->d_hits.as<CandHit>());
    if (lazyClock) PH_CHECK(hipEventRecord(ctx->ev[14], st)); else msReduce = tm.stop(1);
    PH_TRACE(st, "kmermatch: reduce");
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    if (!stalePos.empty() && nTriples > 0 && !cm) {
        // the runs that end at the very end of the sorted array (the last (rep,T) run, and the T-runs of directly preceding
        // reps whose scan the reference lets run across the rep boundary) continue into the stale records: redo them
        const Triple *dTr = reinterpret_cast<const Triple *>(cur);
        std::vector<Triple> tail; uint64_t want = std::min<uint64_t>(nTriples, 4096);
        for (;;) {
            tail.resize(want);
            PH_COPY_SYNC(st, tail.data(), dTr + (nTriples - want), want * sizeof(Triple), hipMemcpyDeviceToHost);
            if (tail.front().target != staleT || want == nTriples) break;
            want = std::min<uint64_t>(nTriples, want * 2);
        }
        if (tail.back().target == staleT) {
            size_t b0 = tail.size(); while (b0 > 0 && tail[b0 - 1].target == staleT) b0--;
            for (size_t h0 = b0; h0 < tail.size(); h0++) {
                if (!(h0 == b0 || tail[h0].rep != tail[h0 - 1].rep)) continue;       // not a run head
                int32_t diagonal = tail[h0].diag, prevDiagonal = tail[h0].diag;
                uint64_t maxDiagonal = 0, diagonalCnt = 0, topScore = 0;
                int bestRev = NUCL ? ((tail[h0].cnt & 0x80000000u) == 0) : 0;
                for (size_t j = h0; j < tail.size(); j++) {
                    const uint64_t cc = tail[j].cnt & 0x7FFFFFFFu;
                    if (prevDiagonal == tail[j].diag) diagonalCnt += cc; else diagonalCnt = cc;
                    if (diagonalCnt >= maxDiagonal) { diagonal = tail[j].diag; maxDiagonal = diagonalCnt; if (NUCL) bestRev = ((tail[j].cnt & 0x80000000u) == 0); }
                    prevDiagonal = tail[j].diag; topScore += cc;
                }
                for (int64_t sp : stalePos) {          // stale records: pos = original k-mer position, kmer field = SIZE_T_MAX (forward)
                    const int32_t d = LONG ? (int32_t) sp : (int32_t) (int16_t) sp;
                    if (prevDiagonal == d) diagonalCnt++; else diagonalCnt = 1;
                    if (diagonalCnt >= maxDiagonal) { diagonal = d; maxDiagonal = diagonalCnt; if (NUCL) bestRev = 0; }
                    prevDiagonal = d; topScore++;
                }
                const uint32_t rep = tail[h0].rep;
                if (rep == staleT) continue;                                       // self run: scanned but never emitted
                uint64_t qn = 0;
                PH_COPY_SYNC(st, &qn, c->d_qoff.as<uint64_t>() + rep + 1, 8, hipMemcpyDeviceToHost);
                CandHit hh;
                PH_COPY_SYNC(st, &hh, c->d_hits.as<CandHit>() + (qn - 1), sizeof(CandHit), hipMemcpyDeviceToHost);
                if (hh.target != staleT || hh.query != rep) { setError("kmermatch: internal error while patching the last run"); return PLASSHIP_ERR_DEVICE; }
                hh.prefScore = bestRev ? -(int) topScore : (int) topScore; hh.diag16 = (uint32_t) (uint16_t) diagonal;
                PH_COPY_SYNC(st, c->d_hits.as<CandHit>() + (qn - 1), &hh, sizeof(CandHit), hipMemcpyHostToDevice);
            }
        }
    }
    return PLASSHIP_OK;
}

template <bool NUCL, bool LONG>
int kmermatchImpl(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par, plasship_cands **out,
                  plasship_kmermatch_stats *stats, bool overflowCheckEarly) {
    typedef Rec<LONG> R;
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const int k = par->kmer_size;
    float msExtract = 0;
    // sharded run (plasship_ctx_set_comm): this rank extracts the sequences [sLo, sHi), owns the k-mer hash buckets that map to
    // it, and owns the representatives / queries [repBase, repBase + ownedN)
    const plasship_comm *cm = commOf(ctx);
    const int W = cm ? cm->world : 1, rk = cm ? cm->rank : 0;

    // ---- slot bounds + offsets ----
    DevBuf dBound, dSlotOff, dScanTmp;
    const size_t scanTmpBytes = exclusiveScanTmpBytes((size_t) N + 2) + (1u << 20);
    if (dBound.alloc(((size_t) N + 1) * 4) != hipSuccess || dSlotOff.alloc(((size_t) N + 2) * 8) != hipSuccess || dScanTmp.alloc(scanTmpBytes) != hipSuccess) {
        setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    if (N) hipLaunchKernelGGL(boundsKernel, dim3(gridFor(N, 256, 4096)), dim3(256), 0, st, db->d_len.as<uint32_t>(), N, k, par->kmers_per_seq, par->kmers_per_seq_scale, dBound.as<uint32_t>());
    if (exclusiveScanU32(st, dBound.as<uint32_t>(), dSlotOff.as<uint64_t>(), N, dScanTmp.p, scanTmpBytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t total = 0;
    uint32_t sLo = 0, sHi = N; uint64_t slotBias = 0, totalAll = 0;
    const bool filtered = cm && shardOwnerFiltered(W);       // every rank extracts all sequences and keeps its own buckets' records
    if (cm && !filtered) {
        DevBuf dSplit; std::vector<uint64_t> hSplit(2 * (size_t) W + 2);
        if (dSplit.alloc(hSplit.size() * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(splitIdsKernel, dim3(1), dim3(256), 0, st, dSlotOff.as<uint64_t>(), N, W, dSplit.as<uint64_t>());
        PH_COPY_SYNC(st, hSplit.data(), dSplit.p, hSplit.size() * 8, hipMemcpyDeviceToHost);
        sLo = (uint32_t) hSplit[rk]; sHi = (uint32_t) hSplit[rk + 1]; slotBias = hSplit[W + 1 + rk];
        total = hSplit[W + 1 + rk + 1] - slotBias;              // slots of this rank's share
        totalAll = hSplit[2 * (size_t) W + 1];                  // ... and of all sequences
    } else {
    PH_CHECK(hipMemcpyAsync(&total, dSlotOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    totalAll = total;
    }
    const uint32_t nMine = sHi - sLo;

    DevBuf dA, dB;   // ping-pong record arrays
    // the line-store partition (linepart.hpp): its record buffers hold whole lines plus one partial line per bucket and piece
    if (cm && W > 1024) { setError("kmermatch: more than 1024 ranks"); return PLASSHIP_ERR_UNSUPPORTED; }
    const LineGeo geo = lineGeometry(total, LONG, ctx->numCU, cm ? totalAll : 0, W);
    // single GPU: both buffers serve level 1 and level 2 (and the group kernel's arenas); sharded run: the slot array / level-1 output,
    // and the packed send buffer of exchange 1 (at most cap1 lines)
    const uint64_t recCap = std::max<uint64_t>(total, (uint64_t) RPL * ((cm && !filtered) ? geo.cap1 : std::max(geo.cap1, geo.cap2)));
    if (dA.alloc(std::max<uint64_t>(recCap, 1) * sizeof(R)) != hipSuccess || dB.alloc(std::max<uint64_t>(recCap, 1) * sizeof(R)) != hipSuccess) {
        setError("kmermatch: out of device memory for the k-mer record arrays (" + std::to_string(2 * recCap * sizeof(R)) + " bytes)"); return PLASSHIP_ERR_DEVICE;
    }

    // ---- extraction ----
    DevBuf dMap, dOvIds, dOvCnt;
    if (dMap.alloc(256) != hipSuccess || dOvIds.alloc(((size_t) N + 1) * 4) != hipSuccess || dOvCnt.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    const unsigned char *map = aa2numTable(NUCL, par->alphabet_size);
    PH_CHECK(hipMemcpyAsync(dMap.p, map, 256, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemsetAsync(dOvCnt.p, 0, 4, st));
    ExtractArgs ea; memset(&ea, 0, sizeof(ea));
    ea.s = db->view(); ea.slotOff = dSlotOff.as<uint64_t>(); ea.arr = dA.p; ea.map = dMap.as<unsigned char>();
    const int alph = NUCL ? 5 : par->alphabet_size;
    { uint64_t p = 1; for (int i = 0; i < 24; i++) { ea.powers[i] = p; p *= (uint64_t) (alph - 1); } }
    ea.k = k; ea.xCode = map[(int) 'X']; ea.kps = par->kmers_per_seq; ea.ignoreMulti = par->ignore_multi_kmer; ea.scale = par->kmers_per_seq_scale;
    ea.seed = (uint64_t) par->hash_shift; ea.overflowIds = dOvIds.as<uint32_t>(); ea.overflowCount = dOvCnt.as<uint32_t>();
    ea.idLo = sLo; ea.idHi = sHi; ea.slotBias = slotBias;
    // candidate k-mers per sequence held in LDS.  128 covers every protein sequence (59 considered k-mers) and every
    // nucleotide sequence up to ~690 nt (59 + 0.1 L); the 1024-candidate instantiation (35 KB of LDS, one wavefront per SIMD)
    // only sees the longer nucleotide contigs, queued by the first launch; what does not fit there either goes to the
    // HBM-scratch launch.
    constexpr int CAP = 128, CAP2 = NUCL ? 1024 : 128;
    DevBuf dWaveList, dWaveCount, dLongList, dLongCount, dKStats;
    static const bool tier0 = [] { const char *e = getenv("PLASSHIP_TIER0"); return e ? atoi(e) != 0 : true; }();
    constexpr uint32_t TIER0_WINDOWS = 256;                  // 4 scores per lane (an 8-scores tier at 5 wavefronts per SIMD gained 0.3 %: not kept)
    if (dWaveList.alloc(((size_t) N + 1) * 4) != hipSuccess || dWaveCount.alloc(4) != hipSuccess || dLongList.alloc(((size_t) N + 1) * 4) != hipSuccess || dLongCount.alloc(4) != hipSuccess ||
        dKStats.alloc(64) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dWaveCount.p, 0, 4, st));
    PH_CHECK(hipMemsetAsync(dLongCount.p, 0, 4, st));
    PH_CHECK(hipMemsetAsync(dKStats.p, 0, 64, st));
    ea.kstats = dKStats.as<unsigned long long>();
    // ---- selected-window cache (section 2c): which sequences keep last call's selection, and where this call's selection is kept ----
    plasship_ctx::KmCache &kc = ctx->kmCache;
    const bool fastIndex = !NUCL && k <= 14 && ea.powers[1] <= 16;      // kmerIndexCore
    const bool cacheEligible = !NUCL && !LONG && (!cm || filtered) && fastIndex && k <= 16 && par->kmers_per_seq >= 1 && par->kmers_per_seq <= (int) KMC_POS + 1 && par->kmers_per_seq_scale == 0.0f &&
                               tuneInt("KMCACHE", 1) == 1;      // PLASSHIP_TUNE_KMCACHE=2 switches the cache off
    const bool cacheReuse = cacheEligible && kc.valid && kc.n == N && kc.gen == db->parentGen && db->d_changed.p && kc.k == k && kc.alph == par->alphabet_size &&
                            kc.kps == par->kmers_per_seq && kc.ignoreMulti == par->ignore_multi_kmer && kc.hashShift == par->hash_shift && !overflowCheckEarly;
    kc.valid = false;                                         // (set again when this call has succeeded)
    DevBuf dCachedList, dCachedCount;
    unsigned long long cacheLinesPtr = 0;                    // -> kstats[4] (see ExtractArgs)
    // The lines are only worth their N x 128 bytes (11 GB at 88 M sequences) when a later call can use them: a DB that was DERIVED from another
    // one (a chained iteration), or a context that has made an eligible call before (a single `plass-hip kmermatcher` call on a DB
    // read from files paid for them in memory and in 128-byte stores per long sequence for nothing)
    const bool cacheWanted = cacheEligible && N && (cacheReuse || db->parentGen != 0 || db->ancestorGen != 0 || kc.seenEligible || tuneInt("KMCACHE_EAGER", 0) == 1);
    if (cacheEligible) kc.seenEligible = true;
    if (cacheWanted) {
        if (kc.lines.bytes < (size_t) N * KMC_LINE) { kc.lines.release(); if (kc.lines.allocLong((size_t) N * KMC_LINE) != hipSuccess) { setError("kmermatch: out of device memory for the selected-window cache"); return PLASSHIP_ERR_DEVICE; } }
        cacheLinesPtr = (unsigned long long) (uintptr_t) kc.lines.p;
        if (cacheReuse) {
            if (dCachedList.alloc(((size_t) N + 1) * 4) != hipSuccess || dCachedCount.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            PH_CHECK(hipMemsetAsync(dCachedCount.p, 0, 4, st));
        }
    } else kc.lines.release();
    // ---- position cache of nucleotide runs (kmermatch_extract.hpp section 2d) ----
    typedef typename std::conditional<LONG, uint32_t, unsigned short>::type PosT;
    plasship_ctx::KmPosCache &pc = ctx->kmPosCache;
    const bool posEligible = NUCL && !cm && N && tier0 && tuneInt("CLASSIFY", 1) == 1 && tuneInt("KMCACHE", 1) == 1;      // PLASSHIP_TUNE_KMCACHE=2 switches it off
    bool posReuse = posEligible && pc.valid && pc.gen != 0 && db->originGen == pc.gen && db->d_origin.p && pc.lng == LONG && pc.k == k && pc.kps == par->kmers_per_seq &&
                          pc.scale == par->kmers_per_seq_scale && pc.ignoreMulti == par->ignore_multi_kmer && pc.hashShift == par->hash_shift;
    // (like the lines of section 2c, the positions are only written when a later call can use them: a derived DB, or a context that has run kmermatcher before)
    bool posWanted = posEligible && (posReuse || db->parentGen != 0 || db->ancestorGen != 0 || db->originGen != 0 || pc.seen || ctx->kmermatchCalls > 0 || tuneInt("KMCACHE_EAGER", 0) == 1);
    if (posEligible) pc.seen = true;
    ctx->kmermatchCalls++;
    DevBuf dPosNew, dIdHashNew;
    unsigned long long cachePtrs[3] = {cacheLinesPtr, 0, 0};
    if (posWanted && (dPosNew.allocLong((size_t) (total + 1) * sizeof(PosT)) != hipSuccess || dIdHashNew.allocLong(((size_t) N + 1) * 8) != hipSuccess)) {
        dPosNew.release(); dIdHashNew.release(); posWanted = posReuse = false;      // no room for the positions: the call works without them (and the next one hashes everything)
    }
    if (posWanted) {
        cachePtrs[1] = (unsigned long long) (uintptr_t) dPosNew.p; cachePtrs[2] = (unsigned long long) (uintptr_t) dIdHashNew.p;
        if (posReuse) {
            if (dCachedList.alloc(((size_t) N + 1) * 4) != hipSuccess || dCachedCount.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            PH_CHECK(hipMemsetAsync(dCachedCount.p, 0, 4, st));
        }
    }
    PH_CHECK(hipMemcpyAsync(dKStats.as<unsigned long long>() + 4, cachePtrs, 24, hipMemcpyHostToDevice, st));      // (cachePtrs lives until the function's last wait)
    PH_CHECK(hipEventRecord(ctx->ev[2], st));
    bool twoLists = false;
    if (!NUCL && k <= 16 && nMine) {
        // short sequences: one thread each; everything else is queued for the wave-per-sequence kernels, in two lists by length
        ShortArgs sa; memset(&sa, 0, sizeof(sa));
        sa.s = ea.s; sa.slotOff = ea.slotOff; sa.arr = ea.arr; sa.map = ea.map; sa.k = k; sa.xCode = ea.xCode; sa.kps = ea.kps; sa.ignoreMulti = ea.ignoreMulti;
        sa.scale = ea.scale; sa.seed = ea.seed; sa.base = (uint64_t) (alph - 1); sa.top = ea.powers[k - 1];
        { uint64_t b = sa.base; int tz = 0; while ((b & 1) == 0) { b >>= 1; tz++; } uint64_t inv = b; for (int i = 0; i < 6; i++) inv *= 2 - b * inv; sa.tz = tz; sa.inv = inv; }
        sa.waveList = dWaveList.as<uint32_t>(); sa.waveCount = dWaveCount.as<uint32_t>(); sa.kstats = dKStats.as<unsigned long long>();
        if (tier0) {
            sa.longList = dLongList.as<uint32_t>(); sa.longCount = dLongCount.as<uint32_t>(); sa.longWindows = TIER0_WINDOWS; twoLists = true;
            // more than 16 scores per lane: straight into the queue the 48-scores tier reads (tiers 0 and 1 append their rare overflows
            // to the same queue, one atomic per sequence)
            sa.hugeList = dOvIds.as<uint32_t>(); sa.hugeCount = dOvCnt.as<uint32_t>(); sa.hugeWindows = 64 * 16;
        }
        sa.idLo = sLo; sa.idHi = sHi; sa.slotBias = slotBias;
        if (cacheReuse) { sa.changed = db->d_changed.as<unsigned char>(); sa.cachedList = dCachedList.as<uint32_t>(); sa.cachedCount = dCachedCount.as<uint32_t>(); }
        // the fast restatement needs k = 14 and the half-indices (7 digits of the base, each at most the X code = base) in 32 bits
        constexpr int KF = 14, HF = KF / 2;
        bool fast = k == KF && tuneInt("SHORT_FAST", 1) == 1;      // PLASSHIP_TUNE_SHORT_FAST=2: the kernel above
        uint64_t pwH = 1; for (int i = 0; i < HF; i++) pwH *= sa.base;
        if (pwH * 2 >= (1ull << 32)) fast = false;
        sa.topLo = (uint32_t) (pwH / sa.base); sa.topHi = (uint32_t) ea.powers[KF - HF - 1]; sa.baseH = (uint32_t) pwH;
        const dim3 shortGrid(std::min<uint32_t>((nMine + 63) / 64, (uint32_t) ctx->numCU * (uint32_t) tuneInt("SHORT", nMine > 20000000u ? 288 : 18)));
        // Resident wavefronts (round 4): every working lane has one partly written 128-byte line of records open; at the 18 wavefronts per CU
        // the kernel's own LDS allows, those are 4.7 MB per XCD against 4 MB of L2 — lines leave the L2 half written and the kernel moves
        // 71 GB for 35 GB of records (profiles/r03_pmc_hbm_traffic.txt).  Unused dynamic LDS caps the residency: 8 KB more per wavefront
        // (10 per CU) took 34.6 -> 29.9 ms off iteration 0 and 28.4 -> 24.2 off iteration 1; with 16 KB (6 per CU) iteration 0, where
        // every lane works, fell to 26.4 ms but the later iterations, where most lanes only queue their sequence, rose by 4-6 ms
        // (profiles/r04_ab_knobs.txt, call 18).  PLASSHIP_TUNE_SHORT_PAD_KB overrides (1 = none).
        const bool allWork = (int64_t) db->maxEntryLen - 2 - k + 1 <= (int64_t) par->kmers_per_seq - 1;      // no sequence long enough to be queued for the wave kernels
        const int padKb = tuneInt("SHORT_PAD_KB", allWork ? 16 : 8);
        const size_t padLds = padKb > 1 ? (size_t) padKb << 10 : 0;
        // records through LDS (extractShortFastKernel<..., STAGE>; 1: on, no residency pad; 3: on, with the pad; 2: off): measured at 50 M reads (profiles/r06_ab_knobs.txt,
        // call 15) it takes 2.8 ms off iteration 0, where every lane works (30.5 -> 27.7 ms), and costs 1.5-2 ms in the later iterations, where a third of
        // the lanes only queue their sequence and the flushes' barriers are paid for fewer records -> on exactly when no sequence can be queued
        const int stageMode = LONG ? 0 : tuneInt("SHORT_STAGE", allWork ? 1 : 2);
        if (fast && sa.base < (1u << 8) && sa.topLo < (1u << 24) && sa.topHi < (1u << 24) && (stageMode == 1 || stageMode == 3)) {
            if constexpr (!LONG) hipLaunchKernelGGL((extractShortFastKernel<LONG, KF, true, true>), shortGrid, dim3(64), stageMode == 3 ? padLds : 0, st, sa);
        }
        else if (fast && sa.base < (1u << 8) && sa.topLo < (1u << 24) && sa.topHi < (1u << 24)) hipLaunchKernelGGL((extractShortFastKernel<LONG, KF, true>), shortGrid, dim3(64), padLds, st, sa);
        else if (fast) hipLaunchKernelGGL((extractShortFastKernel<LONG, KF, false>), shortGrid, dim3(64), 0, st, sa);
        else hipLaunchKernelGGL((extractShortKernel<LONG>), shortGrid, dim3(64), 0, st, sa);   // 18 wavefronts fit a CU; on large sets a finer grid evens out the tail (50 M reads, round 3: 35.7 -> 34.4 ms at 72 per CU; round 4, after the residency cap: another 0.3-1.1 ms per iteration at 144, nothing more at 288 / 576)
        ea.waveList = dWaveList.as<uint32_t>(); ea.waveCount = dWaveCount.as<uint32_t>();
        if (cacheReuse) {
            CachedArgs ca; memset(&ca, 0, sizeof(ca));
            ca.s = ea.s; ca.slotOff = ea.slotOff; ca.arr = ea.arr; ca.map = ea.map; ca.lines = kc.lines.as<unsigned char>();
            ca.list = dCachedList.as<uint32_t>(); ca.count = dCachedCount.as<uint32_t>(); ca.k = k; ca.xCode = ea.xCode;
            ca.base = (uint32_t) ea.powers[1]; ca.base7 = (uint32_t) ea.powers[7]; ca.slotBias = slotBias; ca.seed = ea.seed; ca.kstats = dKStats.as<unsigned long long>();
            hipLaunchKernelGGL(extractCachedKernel, dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) tuneInt("CACHED", 32))), dim3(64), 0, st, ca);
        }
    }
    else if (nMine && tier0 && tuneInt("CLASSIFY", 1) == 1) {                   // PLASSHIP_TUNE_CLASSIFY=2: the 16-scores tier takes every id itself
        hipLaunchKernelGGL(classifyWindowsKernel, dim3(std::min<uint32_t>((nMine + 2047) / 2048, (uint32_t) ctx->numCU * 8)), dim3(256), 0, st, (const uint32_t *) db->d_len.as<uint32_t>(), sLo, sHi, (uint32_t) k,
                           TIER0_WINDOWS, 64u * 16u, dWaveList.as<uint32_t>(), dWaveCount.as<uint32_t>(), dLongList.as<uint32_t>(), dLongCount.as<uint32_t>(), dOvIds.as<uint32_t>(), dOvCnt.as<uint32_t>(),
                           posReuse ? (const uint32_t *) db->d_origin.as<uint32_t>() : (const uint32_t *) nullptr, dCachedList.as<uint32_t>(), dCachedCount.as<uint32_t>());
        ea.waveList = dWaveList.as<uint32_t>(); ea.waveCount = dWaveCount.as<uint32_t>();
        twoLists = true;
        if constexpr (NUCL) {
            if (posReuse) {      // entries the position cache describes: their records rebuilt from the cached window positions
                CachedPosArgs<LONG> ca; memset(&ca, 0, sizeof(ca));
                ca.s = ea.s; ca.slotOff = ea.slotOff; ca.arr = ea.arr; ca.map = ea.map; ca.posOld = pc.pos.p; ca.slotOffOld = pc.slotOff.as<uint64_t>();
                ca.idHashOld = pc.idHash.as<unsigned long long>(); ca.origin = db->d_origin.as<uint32_t>(); ca.posNew = dPosNew.p; ca.idHashNew = dIdHashNew.as<unsigned long long>();
                ca.list = dCachedList.as<uint32_t>(); ca.count = dCachedCount.as<uint32_t>(); ca.k = k; ca.seed = ea.seed; ca.kstats = dKStats.as<unsigned long long>();
                hipLaunchKernelGGL((extractCachedPosKernel<LONG>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) tuneInt("CACHEDPOS", 32))), dim3(64), 0, st, ca);
            }
        }
    }
    PH_CHECK(hipEventRecord(ctx->ev[3], st));
    PH_CHECK(hipEventRecord(ctx->ev[4], st));
    // one-wavefront workgroups per CU of the register tiers' grids.  32 was measured best of 12..64 on the 1 M-read set; on the 50 M-read chain
    // (round 6, profiles/r06_ab_knobs.txt call 7) a CU holds 20 / 16 of them at once and a grid of 32 leaves a second, 60 %-filled round: 40: 57.5 ms
    // for the wave tiers per iteration, 64: 55.8, 96: 54.5, 128: 54.1, 192: 53.7, 256: 53.7, 512: 54.7 against 58.1 at 32 -> 192 for large sets
    static const int waveBlocksEnv = [] { const char *e = getenv("PLASSHIP_EXTRACT_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    const int waveBlocksPerCU = waveBlocksEnv ? waveBlocksEnv : (nMine > 5000000u ? 192 : 32);
    // launch chain, each tier queueing what it cannot hold for the next: (1) register front end, up to 1024 windows (every read,
    // most contigs); (2) the same with 48 scores per lane, up to 3072 windows (proteins longer than that are rare) and, for
    // nucleotides, up to 1024 candidates; (3) three-pass path with codes and scores of up to 8160 residues resident in LDS;
    // (4) the HBM-scratch launch below
    // (0) the same register front end with 4 scores per lane (up to 256 windows: merged read fragments and young contigs) at six
    // wavefronts per SIMD instead of four: the per-sequence steps (staging, ballots, LDS round trips between barriers) are what a
    // 100-250 residue sequence mostly consists of, and more resident wavefronts hide them (2.1 instead of 2.8 ns per sequence).
    // Tiers 0 and 1 take their sequences from the two lists of the thread-per-sequence kernel; what either cannot hold goes to
    // one queue for tier 2, and on to tier 3.
    DevBuf dOv2Ids, dOv2Cnt;
    DevBuf *lastIds = &dOvIds, *lastCnt = &dOvCnt;
    if (nMine) {
        if (dOv2Ids.alloc(((size_t) N + 1) * 4) != hipSuccess || dOv2Cnt.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemsetAsync(dOv2Cnt.p, 0, 4, st));
        const uint32_t wide = std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) waveBlocksPerCU);
        // (round 6) four sequences per wavefront for the sequences of the 4-scores tier (kmermatch_extract.hpp section 2e): the thread-per-sequence kernel's
        // list sorted into four lists by window count, one launch per list; what a launch cannot finish (surplus in the threshold bin, a possible
        // repeat) is queued for the 4-scores tier, which reads that queue instead of the list
        ExtractArgs e0 = ea;
        DevBuf dRowLists, dRowCounts, dRowFall, dRowFallCnt;
        const bool rowTier = !NUCL && !LONG && twoLists && ea.waveList == dWaveList.as<uint32_t>() && fastIndex && k <= 14 && par->kmers_per_seq >= 1 && par->kmers_per_seq <= (int) KMC_POS + 1 &&
                             par->kmers_per_seq_scale == 0.0f && (tuneInt("ROWTIER", nMine > 4000000u ? 1 : 2) & 1) == 1;      // PLASSHIP_TUNE_ROWTIER=2: the 4-scores tier takes the whole list; 3: rows whatever the size
        // (large sets only: at 1 M reads the five extra launches in a row — each a chain of dependent round trips with the chip a tenth full — cost the
        //  wave tiers 0.35 ms per iteration, 1.64 -> 1.99 ms; at 12.5 M reads the rows win 1.9 ms, at 50 M 5.8: profiles/r06_ab_knobs.txt)
        if (rowTier) {
            const uint32_t rs = N + 1;
            if (dRowLists.alloc((size_t) 4 * rs * 4) != hipSuccess || dRowCounts.alloc(16) != hipSuccess || dRowFall.alloc((size_t) rs * 4) != hipSuccess || dRowFallCnt.alloc(4) != hipSuccess) {
                setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE;
            }
            PH_CHECK(hipMemsetAsync(dRowCounts.p, 0, 16, st));
            PH_CHECK(hipMemsetAsync(dRowFallCnt.p, 0, 4, st));
            hipLaunchKernelGGL(binWaveListKernel, dim3(std::min<uint32_t>((nMine + 2047) / 2048, (uint32_t) ctx->numCU * 8)), dim3(256), 0, st, (const uint32_t *) dWaveList.as<uint32_t>(), (const uint32_t *) dWaveCount.as<uint32_t>(),
                               (const uint32_t *) db->d_len.as<uint32_t>(), (uint32_t) k, 96u, 128u, 192u, dRowLists.as<uint32_t>(), rs, dRowCounts.as<uint32_t>());
            RowArgs ra; memset(&ra, 0, sizeof(ra));
            ra.s = ea.s; ra.slotOff = ea.slotOff; ra.arr = ea.arr; ra.map = ea.map; ra.fallList = dRowFall.as<uint32_t>(); ra.fallCount = dRowFallCnt.as<uint32_t>();
            ra.k = k; ra.xCode = ea.xCode; ra.kps = ea.kps; ra.ignoreMulti = ea.ignoreMulti; ra.seed = ea.seed; ra.base = (uint32_t) ea.powers[1]; ra.base7 = (uint32_t) ea.powers[7];
            ra.slotBias = slotBias; ra.kstats = dKStats.as<unsigned long long>();
            const dim3 rowGrid(std::min<uint32_t>((nMine + 3) / 4, (uint32_t) ctx->numCU * (uint32_t) tuneInt("ROWGRID", 32)));
            const bool w5 = tuneInt("ROW_WPE", 4) == 5;
            for (int b = 0; b < 4; b++) {
                ra.list = dRowLists.as<uint32_t>() + (size_t) b * rs; ra.count = dRowCounts.as<uint32_t>() + b;
                if (b == 0) { if (w5) hipLaunchKernelGGL((extractRowKernel<6, 5>), rowGrid, dim3(64), 0, st, ra); else hipLaunchKernelGGL((extractRowKernel<6, 4>), rowGrid, dim3(64), 0, st, ra); }
                else if (b == 1) { if (w5) hipLaunchKernelGGL((extractRowKernel<8, 5>), rowGrid, dim3(64), 0, st, ra); else hipLaunchKernelGGL((extractRowKernel<8, 4>), rowGrid, dim3(64), 0, st, ra); }
                else if (b == 2) hipLaunchKernelGGL((extractRowKernel<12, 4>), rowGrid, dim3(64), 0, st, ra);
                else hipLaunchKernelGGL((extractRowKernel<16, 4>), rowGrid, dim3(64), 0, st, ra);
            }
            e0.waveList = dRowFall.as<uint32_t>(); e0.waveCount = dRowFallCnt.as<uint32_t>();
        }
        // tiers 0 and 1 both queue into dOvIds
        if (twoLists) {
            // (round 5: tiers 0 and 1 — disjoint lists — and the cached kernel side by side on three streams: extraction 81.7 against 81.2 ms
            //  per iteration, nothing gained; profiles/r05_ab_knobs.txt)
            if (tuneInt("TIER0_WPE", 5) == 6) hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 4, 256, 6>), dim3(wide), dim3(64), 0, st, e0);
            else hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 4, 256>), dim3(wide), dim3(64), 0, st, e0);
            ExtractArgs e1 = ea; e1.waveList = dLongList.as<uint32_t>(); e1.waveCount = dLongCount.as<uint32_t>();
            hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 16, 992>), dim3(wide), dim3(64), 0, st, e1);
        } else hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 16, 992>), dim3(wide), dim3(64), 0, st, ea);
        ExtractArgs e2 = ea; e2.waveList = dOvIds.as<uint32_t>(); e2.waveCount = dOvCnt.as<uint32_t>();
        e2.overflowIds = dOv2Ids.as<uint32_t>(); e2.overflowCount = dOv2Cnt.as<uint32_t>();
        // (round 4: a nucleotide sequence of this tier has at most 3 072 windows, i.e. 60 + 0.1 L <= 370 selected k-mers at the workflow's
        //  scaling: 512 candidates instead of 1 024 halve its LDS and let eight instead of four wavefronts work per CU; a candidate set
        //  that does not fit is queued for the next tier like any other overflow)
        constexpr int CAP48 = NUCL ? 512 : 128;
        if (NUCL && tuneInt("NUCL_CAP48", 1) == 1) hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP48, false, 48, 992, NUCL ? 2 : 0>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) tuneInt("TIER2", 8))), dim3(64), 0, st, e2);
        else hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP2, false, 48, 992>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) tuneInt("TIER2", NUCL ? 4 : 16))), dim3(64), 0, st, e2);
        PH_CHECK(hipMemsetAsync(dOvCnt.p, 0, 4, st));            // tier 2 has consumed the first queue: it becomes tier 3's output queue
        ExtractArgs e3 = ea; e3.waveList = dOv2Ids.as<uint32_t>(); e3.waveCount = dOv2Cnt.as<uint32_t>();
        e3.overflowIds = dOvIds.as<uint32_t>(); e3.overflowCount = dOvCnt.as<uint32_t>();
        hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP2, false, 0, 8160>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) tuneInt("TIER3", NUCL ? 2 : 5))), dim3(64), 0, st, e3);
    }
    // what the last stage could not hold either (candidate sets beyond LDS)
    DevBuf &dLastIds = *lastIds, &dLastCnt = *lastCnt;
    PH_CHECK(hipEventRecord(ctx->ev[5], st));
    uint32_t nOv = 0;
    // protein DBs: a candidate set hardly ever exceeds the tiers' 128 entries (59 considered k-mers), and the last tier keeps up to
    // 8160 residues in LDS.  "Hardly ever": every window whose score is <= the threshold score is a candidate, and equal k-mers
    // share a score — a homopolymer or tandem repeat of >= ~85 residues whose k-mer lies at or below the threshold overfills the set
    // (ADVICE r3).  So the count of the last tier's hand-overs is not waited for here when an overflow is UNLIKELY; it travels with
    // the group stage's counts (kmermatchLines), the last tier has left such a sequence's slots as sentinels, and if the count turns
    // out non-zero the call starts over with the wait in place (`overflowCheckEarly`).
    const bool overflowPossible = overflowCheckEarly || cm != nullptr || NUCL || db->maxEntryLen > 8160u || par->kmers_per_seq > 120 || par->kmers_per_seq_scale != 0.0f;
    if (nMine && overflowPossible) {
        PH_CHECK(hipMemcpyAsync(&nOv, dLastCnt.p, 4, hipMemcpyDeviceToHost, st));
        PH_CHECK(plasship::streamSync(st));
    } else if (nMine) hipLaunchKernelGGL((fillOverflowSlotsKernel<LONG>), dim3(64), dim3(64), 0, st, (const uint32_t *) dLastIds.as<uint32_t>(), (const uint32_t *) dLastCnt.as<uint32_t>(),
                                         (const uint64_t *) dSlotOff.as<uint64_t>(), slotBias, dA.p);
    PH_CHECK(hipGetLastError());
    if (nOv) {   // sequences whose candidate set did not fit LDS: same kernel, candidates in HBM scratch
        std::vector<uint32_t> ids(nOv), lens(nOv);
        DevBuf dLens, dSOff, dSCap, dScratch;
        if (dLens.alloc((size_t) nOv * 4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(gatherU32Kernel, dim3(gridFor(nOv, 256, 1024)), dim3(256), 0, st, db->d_len.as<uint32_t>(), dLastIds.as<uint32_t>(), nOv, dLens.as<uint32_t>());
        PH_CHECK(hipMemcpyAsync(lens.data(), dLens.p, (size_t) nOv * 4, hipMemcpyDeviceToHost, st));
        PH_CHECK(plasship::streamSync(st));
        std::vector<uint64_t> soff(nOv); std::vector<uint32_t> scap(nOv); uint64_t tot = 0;
        for (uint32_t i = 0; i < nOv; i++) { uint32_t c = 64; while (c < lens[i] + 1) c <<= 1; scap[i] = c; soff[i] = tot; tot += c; }
        if (dSOff.alloc((size_t) nOv * 8) != hipSuccess || dSCap.alloc((size_t) nOv * 4) != hipSuccess || dScratch.alloc(tot * sizeof(Cand)) != hipSuccess) {
            setError("kmermatch: out of device memory for the long-sequence scratch"); return PLASSHIP_ERR_DEVICE;
        }
        PH_CHECK(hipMemcpyAsync(dSOff.p, soff.data(), (size_t) nOv * 8, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemcpyAsync(dSCap.p, scap.data(), (size_t) nOv * 4, hipMemcpyHostToDevice, st));
        ExtractArgs fa = ea; fa.idList = dLastIds.as<uint32_t>(); fa.nIds = nOv; fa.scratch = dScratch.as<Cand>(); fa.scratchOff = dSOff.as<uint64_t>(); fa.scratchCap = dSCap.as<uint32_t>();
        hipLaunchKernelGGL((extractKernel<NUCL, LONG, 1, true>), dim3(std::min<uint32_t>(nOv, (uint32_t) ctx->numCU * 8)), dim3(64), 0, st, fa);
        PH_CHECK(plasship::streamSync(st));
        PH_CHECK(hipGetLastError());
    }
    PH_TRACE(st, "kmermatch: extraction");
    traceBadIds<LONG>(ctx, "kmermatch: extracted slots", dA.p, total, N);

    {
        int keyBitsL = 0;
        if (NUCL) keyBitsL = 2 * k; else { long double v = 1; for (int i = 0; i < k; i++) v *= (long double) (alph - 1); while (keyBitsL < 63 && (long double) (1ULL << keyBitsL) < v) keyBitsL++; }
        LinesOut lo;
        int rcL = kmermatchLines<NUCL, LONG>(ctx, db, par, geo, total, dA, dB, dSlotOff, dKStats, ea, keyBitsL, lo, (nMine && !overflowPossible) ? dLastCnt.as<uint32_t>() : nullptr, filtered);
        if (rcL) return rcL;                                  // (KM_RETRY_EARLY_OVERFLOW_CHECK: the caller starts over)
        std::unique_ptr<plasship_cands> holderL; uint64_t NcL = 0; float msReduceL = 0;
        rcL = reduceToCandidates<NUCL, LONG>(ctx, db, lo.triples, lo.nTriples, lo.stalePos, lo.staleT, holderL, NcL, msReduceL, true);
        if (rcL) return rcL;
        {   // the stage boundaries (all recorded, all complete: reduceToCandidates ended with a wait for the stream)
            auto ms = [&](int a, int b) { float v = 0; (void) hipEventElapsedTime(&v, ctx->ev[a], ctx->ev[b]); return v; };
            msExtract = ms(0, 1); lo.msSort1 = ms(1, 6); lo.msGroup = ms(6, 7); lo.msSort2 = ms(12, 13); msReduceL = ms(13, 14);
        }
        if (stats) {
            stats->n_kmer_records = lo.Nk; stats->n_grouped = lo.Nm; stats->n_candidates = NcL; stats->record_bytes = LONG ? 20 : 16;
            float ms = 0, ms2 = 0; (void) hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); (void) hipEventElapsedTime(&ms2, ctx->ev[4], ctx->ev[5]);
            stats->ms_extract_short_kernel = ms; stats->ms_extract_wave_kernel = ms2; stats->ms_extract_kernel = ms + ms2;
            stats->ms_part_scatter = lo.msPart; stats->n_part_scatter = lo.nPart;
            unsigned long long ks[4] = {0, 0, 0, 0}; uint32_t nCachedSeqs = 0;
            if (cacheReuse || posReuse) PH_CHECK(hipMemcpyAsync(&nCachedSeqs, dCachedCount.p, 4, hipMemcpyDeviceToHost, st));
            PH_COPY_SYNC(st, ks, dKStats.p, 32, hipMemcpyDeviceToHost);
            stats->n_cached_sequences = nCachedSeqs; stats->reserved0 = 0;
            stats->short_residues = ks[0]; stats->short_records = ks[1]; stats->wave_residues = ks[2]; stats->wave_records = ks[3];
            stats->residues = db->residues;
            stats->ms_extract = msExtract; stats->ms_sort1 = lo.msSort1; stats->ms_group = lo.msGroup; stats->ms_sort2 = lo.msSort2; stats->ms_reduce = msReduceL;
            stats->n_scratch_sequences = nOv; stats->n_restarts = overflowCheckEarly ? 1u : 0u;
        }
        if (cacheWanted) {     // the lines now describe THIS DB (the wave kernels rewrote what changed, the rest was kept)
            kc.valid = true; kc.n = N; kc.gen = db->gen; kc.k = k; kc.alph = par->alphabet_size; kc.kps = par->kmers_per_seq; kc.ignoreMulti = par->ignore_multi_kmer; kc.hashShift = par->hash_shift;
        }
        if (posWanted) {       // the positions now describe THIS DB: it becomes the anchor the derived DBs' d_origin refers to
            if (pc.slotOff.allocLong(((size_t) N + 2) * 8) != hipSuccess) { pc.valid = false; setError("kmermatch: out of device memory for the position cache"); return PLASSHIP_ERR_DEVICE; }
            PH_CHECK(hipMemcpyAsync(pc.slotOff.p, dSlotOff.p, ((size_t) N + 1) * 8, hipMemcpyDeviceToDevice, st));
            PH_CHECK(plasship::streamSync(st));
            std::swap(pc.pos.p, dPosNew.p); std::swap(pc.pos.bytes, dPosNew.bytes); std::swap(pc.idHash.p, dIdHashNew.p); std::swap(pc.idHash.bytes, dIdHashNew.bytes);
            pc.valid = true; pc.n = N; pc.gen = db->gen; pc.lng = LONG; pc.k = k; pc.kps = par->kmers_per_seq; pc.scale = par->kmers_per_seq_scale;
            pc.ignoreMulti = par->ignore_multi_kmer; pc.hashShift = par->hash_shift;
        }
        *out = holderL.release();
        return PLASSHIP_OK;
    }
}
}  // namespace

extern "C" int plasship_kmermatch(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par,
                                  plasship_cands **out, plasship_kmermatch_stats *stats) {
    if (!ctx || !db || !par || !out) { setError("plasship_kmermatch: bad argument"); return PLASSHIP_ERR_ARG; }
    const bool nucl = db->dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES;
    if (par->kmer_size < 2 || par->kmer_size > (nucl ? 31 : 23)) { setError("plasship_kmermatch: unsupported k"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (!nucl && par->alphabet_size != 13 && par->alphabet_size != 21) { setError("plasship_kmermatch: --alph-size must be 13 or 21"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (db->maxEntryLen >= (1u << 20)) { setError("plasship_kmermatch: sequences of 2^20 residues or more are not supported"); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    // kmermatcher.cpp:797-802: KmerPosition<short> unless a sequence is too long for it.  (Nucleotide k > 23: the 16-byte grouped
    // record has no room for the k-mer the strand rule needs — embedOrd — so such a run takes the 24-byte layout, whose arithmetic
    // is the same on sequences that short.)
    // (PLASSHIP_TUNE_FORCE_LONG=1: the 24-byte layout whatever the lengths — the tests run the short chains through the long kernels with it)
    const bool lng = !(db->maxEntryLen < (uint32_t) SHRT_MAX) || (nucl && par->kmer_size > 23) || tuneInt("FORCE_LONG", 0) == 1;
    int rc = KM_RETRY_EARLY_OVERFLOW_CHECK;
    for (int attempt = 0; attempt < 2 && rc == KM_RETRY_EARLY_OVERFLOW_CHECK; attempt++) {
        const bool early = attempt > 0;
        if (nucl) rc = lng ? kmermatchImpl<true, true>(ctx, db, par, out, stats, early) : kmermatchImpl<true, false>(ctx, db, par, out, stats, early);
        else rc = lng ? kmermatchImpl<false, true>(ctx, db, par, out, stats, early) : kmermatchImpl<false, false>(ctx, db, par, out, stats, early);
    }
    return commFinish(ctx, rc);
}
