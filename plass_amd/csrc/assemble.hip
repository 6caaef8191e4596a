// plasship: assembleresults / nuclassembleresults on gfx950 (rows A1–A5 of SURVEY.md §8a).  Product code.
//
// Reference behaviour reproduced (file:line in /root/reference):
//   src/assembler/assembleresult.cpp:19-36     CompareResultByScore (score, alnLength, smaller key) — strict
//   src/assembler/assembleresult.cpp:40-57     selectFragmentToExtend: pop until an extendable non-self hit
//   src/assembler/assembleresult.cpp:161-189   queue fill: raw score from bit score, score/seqId rescaling
//   src/assembler/assembleresult.cpp:193-314   extension rounds, deferred hits re-scored on the new query
//   src/assembler/assembleresult.cpp:70-108    updateAlignment (identity count end-EXCLUSIVE, exact chars)
//   src/assembler/assembleresult.cpp:316-342   flags 0x20 (became contig) / 0x80 (consumed), carry-over pass
//   lib/mmseqs/src/alignment/Matcher.cpp:248-320  the text round trip the reference reads its input through:
//       seqId has 3 truncated decimals ("1.00" for 1.0), alnLength = max(|qE-qS|,|tE-tS|)+1, score = bit score
//
// Kernel design: the queue of a query lives in registers, one alignment per lane — 16, 32 or 64 lanes per query by queue size
// (assembleGroupKernel<16/32/64>; work lists by prefix sums after an exact pre-screen), an HBM-resident queue above 64 alignments
// (assembleBigKernel, one wavefront per query).  A ROUND of the reference's pop loop is computed from the queued set: the comparator
// is a strict total order, so std::priority_queue's pop order is "max of what is in the queue", the best right-extendable and the
// best left-extendable hit (two arg-max reductions on a packed priority) are the round's two extensions, and every other hit is
// dropped or deferred by the geometry tests with the final offsets, in parallel; deferred hits are re-scored by their own lane
// (wide queues) or by the 16-lane group.  The growing query lives in an HBM arena sized by an exact upper bound (query + every
// target that could ever be attached on either side), so no allocation happens in the loop.
// The nucleotide variant (non-strict Bayesian comparator, heap-order dependent) has its own kernel below that
// replays libstdc++'s heap operations (assembleNuclKernel).
#include "common.hpp"
#include "device_utils.hpp"
#include "posterior_class.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <memory>
#include <cstring>

namespace plasship {

struct Item {             // one alignment of the current query (reference: Matcher::result_t in the queue)
    uint32_t target; int32_t score; uint32_t alnLength; float seqId;
    int32_t qStart, qEnd; uint32_t qLen; int32_t dbStart, dbEnd; uint32_t dbLen;
    uint32_t state;       // 0 = in queue, 1 = deferred (tmpAlignments), 2 = gone
    uint32_t pad;
};

struct AsmArgs {
    SeqView s;
    const uint64_t *qoff; const AlnRec *recs;   // CSR of accepted alignments
    Item *items;                                // [nLines] scratch, same indexing as recs
    const uint64_t *arenaOff;                   // [n+1] byte offsets into arena
    const uint32_t *leftCap;                    // [n] bytes reserved left of the original query
    char *arena;
    uint32_t *flags;                            // [n] wasExtended bits (0x20, 0x80)
    uint32_t *newLen;                           // [n] length of the extended query (valid if flag 0x20)
    uint64_t *newStart;                         // [n] absolute arena offset of the extended query
    const signed char *mat;
    double lambda, logK, ln2;
    float seqIdThr; uint64_t maxSeqLen; int rescoreMode;
    uint32_t ownLaneMin;                        // wide register queues: at least this many deferred hits of a round are re-scored each by its own lane, fewer one after the other by the whole group
    unsigned long long *stats;                  // [0] extended, [1] rescored hits, [2] rescored overlap residues
    uint32_t *bigList; uint32_t nBig;   // queries with more than 64 alignments (HBM-resident queue)
    uint32_t *midList; uint32_t nMid;   // 33..64 alignments: one wavefront per query
    uint32_t *mid32List; uint32_t nMid32;   // 17..32 alignments: half a wavefront per query
    uint32_t *smallList; uint32_t nSmall;                         // <= 16 alignments: 16 lanes per query
    uint32_t *heap;                             // nucleotide variant: [3*nLines] index heap + deferral list + consumed targets per query
    // nucleotide variant: comparator decisions on a threshold come from a host-evaluated table (see nuclLess)
    const uint32_t *ambKeys; const uint8_t *ambVals; uint32_t ambMask;    // open addressing, 4 words per key, empty = alpha1 0
    uint32_t *needKeys; uint32_t *needCount; uint32_t needCap;            // tuples the table lacks
    uint32_t *redoList; uint32_t *redoCount;                              // queries that met such a tuple: run again
    const uint32_t *queryList; uint32_t nQueryList;                       // list-driven pass (nullptr = all queries)
    uint8_t *cmpCache;                          // nucleotide variants: memo of the comparator's posterior class (see nuclLess)
    // guided variant: the protein twins (same ids) and their arena
    SeqView aa; char *aaArena; const uint64_t *aaArenaOff; const uint32_t *aaLeftCap; uint32_t *aaNewLen; uint64_t *aaNewStart;
};

// text round trip of seqId (Util.cpp:278-307 + strtod in Matcher.cpp:265)
__device__ __forceinline__ float seqIdThroughText(float f) {
    if (f == 1.0f) return 1.0f;
    const int t = (int) (f * 1000);
    int zeros = 0;
    if ((double) f < 0.10) zeros++;
    if ((double) f < 0.01) zeros++;
    int digits = 1; for (int x = t; x >= 10; x /= 10) digits++;
    double den = 1.0; for (int i = 0; i < digits + zeros; i++) den *= 10.0;
    return (float) ((double) t / den);
}

// 8 residues per lane and memory round trip: the loops below are latency bound (a query is a chain of dependent
// loads), so fewer, wider accesses are what counts.  gfx950 global loads/stores take unaligned addresses; sequence and
// arena buffers are padded past their ends, bytes beyond the range are masked off.
__device__ __forceinline__ uint64_t loadU64Unaligned(const char *p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }
__device__ __forceinline__ void storeU64Unaligned(char *p, uint64_t w) { __builtin_memcpy(p, &w, 8); }
// the last, partial word of a copy: the source word is read whole (the buffers are padded), the destination gets its 1-7 bytes as one
// 4-, 2- and 1-byte store each — never a byte beyond n (the next sequence of the arena / the output DB belongs to another group).
// (Byte by byte the tail was up to seven loads and seven stores of the whole wavefront per copy: round 3, profiles/r03_pmc.)
__device__ __forceinline__ void storeTail(char *d, uint64_t v, unsigned r) {
    if (r & 4u) { const uint32_t x = (uint32_t) v; __builtin_memcpy(d, &x, 4); d += 4; v >>= 32; }
    if (r & 2u) { const uint16_t x = (uint16_t) v; __builtin_memcpy(d, &x, 2); d += 2; v >>= 16; }
    if (r & 1u) *d = (char) v;
}
template <int G> __device__ __forceinline__ void copyBytesG(char *dst, const char *src, unsigned n, int gl) {
    for (unsigned i = 8u * (unsigned) gl; i < n; i += 8u * G) {
        const uint64_t w = loadU64Unaligned(src + i);
        if (i + 8 <= n) storeU64Unaligned(dst + i, w);
        else storeTail(dst + i, w, n - i);
    }
}
// The score table in LDS.  Entry [0][0] is forced to 0: the scoring loops below BLANK the columns outside [first, last] (byte 0
// in both words) instead of predicating them, so that the lookups of a step are unconditional — the compiler issues them together
// and waits once (with one predicated block per column every `ds_read` had its own wait: profiles/r03_pmc_sq_per_kernel.txt).
// No residue is byte 0.
__device__ __forceinline__ void stageScoreTable(signed char *smat, const signed char *__restrict__ mat, int nThreads) {
    for (int i = threadIdx.x; i < 123 * 123; i += nThreads) smat[i] = i ? mat[i] : (signed char) 0;
}
// 0xFF in every byte j of the result with lo <= j < hi
__device__ __forceinline__ uint64_t byteRangeMask(int lo, int hi) {
    const uint64_t belowHi = hi >= 8 ? ~0ULL : (hi <= 0 ? 0ULL : ((1ULL << (8 * hi)) - 1ULL));
    const uint64_t belowLo = lo >= 8 ? ~0ULL : (lo <= 0 ? 0ULL : ((1ULL << (8 * lo)) - 1ULL));
    return belowHi & ~belowLo;
}
// number of zero bytes of x among the bytes selected by mask (exact zero-byte test)
__device__ __forceinline__ int zeroBytes(uint64_t x, uint64_t mask) {
    const uint64_t lo7 = 0x7F7F7F7F7F7F7F7FULL;
    return __popcll(~(((x & lo7) + lo7) | x | lo7) & mask);
}
// computeGlobalSubstitutionStartEndDistance (DistanceCalculator.h:204-220) over an ungapped overlap of `len` columns:
// first/last column ('*' trimming), score sum over [first, last], identities over [first, last).  The boundary bytes
// and the first 8-residue words are requested together (one memory round trip for overlaps up to 8*G residues).
template <int G> __device__ __forceinline__ void scoreColumnsG(const char *q, const char *t, unsigned len, const signed char *smat, int gl,
                                                               unsigned &first, unsigned &last, int &s, int &ids) {
    const unsigned p0 = 8u * (unsigned) gl;
    uint64_t qw = 0, tw = 0;
    if (p0 < len) { qw = loadU64Unaligned(q + p0); tw = loadU64Unaligned(t + p0); }
    const char q0 = q[0], t0 = t[0], qe = q[len - 1], te = t[len - 1];
    first = (q0 == '*' || t0 == '*') ? 1u : 0u;
    last = len - 1;
    if (last > 0 && (qe == '*' || te == '*')) last--;
    for (unsigned p = p0; p < len; p += 8u * G) {
        if (p != p0) { qw = loadU64Unaligned(q + p); tw = loadU64Unaligned(t + p); }
        const int lo = (int) first - (int) p, hi = (int) last - (int) p;       // columns [lo, hi] of this word are scored
        uint64_t qv = qw, tv = tw;
        if (lo <= 0 && hi >= 8) ids += zeroBytes(qw ^ tw, ~0ULL);               // an interior word: all eight columns scored and counted (round 5: the two
        else {                                                                  // range masks cost as much as the eight lookups)
            const uint64_t m = byteRangeMask(lo, hi + 1);
            ids += zeroBytes(qw ^ tw, byteRangeMask(lo, hi));                   // [qStart, qEnd): the last aligned column is not counted
            qv = qw & m; tv = tw & m;
        }
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
            const unsigned a = (unsigned) (qv >> (8 * j)) & 0xFFu, b = (unsigned) (tv >> (8 * j)) & 0xFFu;
            s += (int) smat[a * 123 + b];
        }
    }
}

// the same for ONE lane walking the whole overlap: 32 residues of both sequences are requested per round trip (two
// unaligned 16-byte loads each) — the walk is a chain of dependent memory round trips, so fewer and wider is what counts
__device__ __forceinline__ void scoreColumnsSerial(const char *q, const char *t, unsigned len, const signed char *smat,
                                                   unsigned &first, unsigned &last, int &s, int &ids) {
    const char q0 = q[0], t0 = t[0], qe = q[len - 1], te = t[len - 1];
    uint64_t qw[4], tw[4];
    __builtin_memcpy(qw, q, 32); __builtin_memcpy(tw, t, 32);
    first = (q0 == '*' || t0 == '*') ? 1u : 0u;
    last = len - 1;
    if (last > 0 && (qe == '*' || te == '*')) last--;
    for (unsigned p = 0; p < len; p += 32) {
        if (p) { __builtin_memcpy(qw, q + p, 32); __builtin_memcpy(tw, t + p, 32); }
#pragma unroll
        for (unsigned k = 0; k < 4; k++) {
            const int lo = (int) first - (int) (p + 8 * k), hi = (int) last - (int) (p + 8 * k);
            uint64_t qv = qw[k], tv = tw[k];
            if (lo <= 0 && hi >= 8) ids += zeroBytes(qw[k] ^ tw[k], ~0ULL);     // an interior word (see scoreColumnsG)
            else {
                const uint64_t m = byteRangeMask(lo, hi + 1);
                ids += zeroBytes(qw[k] ^ tw[k], byteRangeMask(lo, hi));
                qv = qw[k] & m; tv = tw[k] & m;
            }
#pragma unroll
            for (unsigned j = 0; j < 8; j++) {
                const unsigned a = (unsigned) (qv >> (8 * j)) & 0xFFu, b = (unsigned) (tv >> (8 * j)) & 0xFFu;
                s += (int) smat[a * 123 + b];
            }
        }
    }
}

// ungappedAlignmentByDiagonal, mode 3 (DistanceCalculator.h:115-175,204-220) + the counts updateAlignment needs
struct Rescored { int startPos, endPos; unsigned score, diagonalLen; int idExcl; };
__device__ __forceinline__ Rescored rescoreOnDiagonal(const char *q, unsigned qLen, const char *t, unsigned tLen, int diagonal,
                                                      const signed char *smat) {
    Rescored r; r.startPos = -1; r.endPos = -1; r.score = 0; r.diagonalLen = 0; r.idExcl = 0;
    const unsigned dist = (unsigned) abs(diagonal);
    unsigned qo, to, len;
    if (diagonal >= 0 && dist < qLen) { qo = dist; to = 0; len = min(tLen, qLen - dist); }
    else if (diagonal < 0 && dist < tLen) { qo = 0; to = dist; len = min(tLen - dist, qLen); }
    else return r;
    r.diagonalLen = len;
    if (len == 0) return r;
    unsigned first, last;
    int s = 0, ids = 0;
    scoreColumnsG<64>(q + qo, t + to, len, smat, laneId(), first, last, s, ids);
    s = waveReduceSum(s); ids = waveReduceSum(ids);
    r.score = (unsigned) max(s, 0); r.startPos = (int) first; r.endPos = (int) last; r.idExcl = ids;
    return r;
}

__device__ __forceinline__ void waveMemSync() {   // make this wave's global stores visible to its own later loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Four wavefronts per workgroup share the 15 KB score table (one wavefront per workgroup let LDS cap the CU at 10 wavefronts); a
// wavefront works on its own queries and orders its own memory operations with fences — there is no workgroup barrier in the loop.
__global__ __launch_bounds__(256) void assembleBigKernel(AsmArgs a) {
    __shared__ signed char smat[123 * 123 + 7];
    stageScoreTable(smat, a.mat, 256);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long nExt = 0, nResc = 0, nRescRes = 0, nAln = 0, nQRes = 0;
    for (uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6); w < a.nBig; w += gridDim.x * 4) {
        const uint32_t id = a.bigList[w];
        const uint64_t h0 = a.qoff[id], h1 = a.qoff[id + 1];
        const uint32_t h = (uint32_t) (h1 - h0);
        if (h == 0) continue;
        const uint64_t aoff = a.arenaOff[id];
        if (a.arenaOff[id + 1] == aoff) continue;          // no non-self hit: can never be extended
        Item *it = a.items + h0;
        const char *orig = a.s.data + seqOff(a.s, id);
        unsigned querySeqLen = seqLen(a.s, id);
        nAln += h; nQRes += querySeqLen;
        // ---- queue fill (assembleresult.cpp:161-189) ----
        for (uint32_t i = lane; i < h; i += 64) {
            const AlnRec r = a.recs[h0 + i];
            Item x;
            x.target = r.target;
            const int aq = (r.qStart == -1) ? 0 : r.qStart, ad = (r.dbStart == -1) ? 0 : r.dbStart;
            x.alnLength = (uint32_t) (max(abs(r.qEnd - aq), abs(r.dbEnd - ad)) + 1);           // Matcher::computeAlnLength
            const int rawScore = (int) (fma((double) r.bitScore, a.ln2, a.logK) / a.lambda + 0.5);
            const float scorePerCol = (float) rawScore / (float) ((double) x.alnLength + 0.5);
            const float sid = r.fromText ? r.seqId : seqIdThroughText(r.seqId);
            const float alnLen = (float) x.alnLength;
            const float ids = sid * alnLen;
            x.seqId = (float) ((double) ids / ((double) alnLen + 0.5));
            x.score = (int) (scorePerCol * 100);
            x.qStart = r.qStart; x.qEnd = r.qEnd; x.qLen = (uint32_t) r.qLen; x.dbStart = r.dbStart; x.dbEnd = r.dbEnd; x.dbLen = (uint32_t) r.dbLen;
            x.state = r.accepted ? 0u : 2u; x.pad = 0;                                    // a hole of a sparse list was never queued
            it[i] = x;
        }
        // the query starts in the middle of its arena slice
        char *buf = a.arena + aoff;
        uint64_t curStart = a.leftCap[id];
        copyBytesG<64>(buf + curStart, orig, querySeqLen, lane);
        uint64_t curLen = querySeqLen;
        waveMemSync();
        bool couldExtend = false;
        uint32_t inQueue = h;
        while (inQueue > 0) {
            unsigned leftOff = 0, rightOff = 0;
            bool brokeOut = false;
            // ---- one round as a function of the set of queued hits (see assembleGroupKernel): best right-extendable and
            //      best left-extendable hit, then every popped hit classified with the final offsets ----
            unsigned long long bestR = 0, bestL = 0; uint32_t idxR = 0xFFFFFFFFu, idxL = 0xFFFFFFFFu;
            for (uint32_t i = lane; i < h; i += 64) {
                Item x = it[i];
                if (x.state == 1) { it[i].state = 2; continue; }      // leftovers of the previous round
                if (x.state != 0) continue;
                // ranks inside the queue are not precomputed here: ties on (score, alnLength) are broken by the smaller target id
                const unsigned long long key = ((unsigned long long) ((uint32_t) x.score ^ 0x80000000u) << 32) | (unsigned long long) x.alnLength;
                const bool notBoth = !(x.dbStart == 0 && x.qStart == 0);
                const bool rightStart = x.dbStart == 0 && (x.dbEnd != (int) x.dbLen - 1);
                const bool leftStart = x.qStart == 0 && (x.qEnd != (int) x.qLen - 1);
                if (!((rightStart || leftStart) && notBoth) || x.target == id) continue;
                const unsigned tLen = seqLen(a.s, x.target);
                if (x.dbStart == 0) {
                    const unsigned fragR = tLen - ((unsigned) x.dbEnd + 1);
                    if (fragR > 0 && (unsigned) x.qEnd == (querySeqLen - 1) && (key > bestR || (key == bestR && x.target < it[idxR].target))) { bestR = key; idxR = i; }
                } else if (x.qStart == 0) {
                    if (x.dbStart > 0 && (unsigned) x.dbEnd == (tLen - 1) && (key > bestL || (key == bestL && x.target < it[idxL].target))) { bestL = key; idxL = i; }
                }
            }
            // wave arg-max with the target id as the final tie-break
            uint32_t tgtR = idxR != 0xFFFFFFFFu ? it[idxR].target : 0xFFFFFFFFu, tgtL = idxL != 0xFFFFFFFFu ? it[idxL].target : 0xFFFFFFFFu;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long okR = __shfl_xor(bestR, o, 64), okL = __shfl_xor(bestL, o, 64);
                const uint32_t oiR = __shfl_xor(idxR, o, 64), oiL = __shfl_xor(idxL, o, 64), otR = __shfl_xor(tgtR, o, 64), otL = __shfl_xor(tgtL, o, 64);
                if (oiR != 0xFFFFFFFFu && (idxR == 0xFFFFFFFFu || okR > bestR || (okR == bestR && otR < tgtR))) { bestR = okR; idxR = oiR; tgtR = otR; }
                if (oiL != 0xFFFFFFFFu && (idxL == 0xFFFFFFFFu || okL > bestL || (okL == bestL && otL < tgtL))) { bestL = okL; idxL = oiL; tgtL = otL; }
            }
            const bool haveR = idxR != 0xFFFFFFFFu, haveL = idxL != 0xFFFFFFFFu;
            // full priority order between the two: (score, alnLength) then the smaller target id wins
            const bool rFirst = haveR && (!haveL || bestR > bestL || (bestR == bestL && tgtR < tgtL));
            waveMemSync();
            auto extendRight = [&]() {
                const Item x = it[idxR];
                const char *tSeq = a.s.data + seqOff(a.s, x.target);
                const unsigned tLen = seqLen(a.s, x.target), dbEnd = (unsigned) x.dbEnd, fragLen = tLen - (dbEnd + 1);
                copyBytesG<64>(buf + curStart + curLen, tSeq + dbEnd + 1, fragLen, lane);
                curLen += fragLen; rightOff += fragLen;
                if (lane == 0) atomicOr(&a.flags[x.target], 0x80u);
            };
            if (rFirst) extendRight();
            if (haveL) {
                const Item x = it[idxL];
                const unsigned fragLen = (unsigned) x.dbStart;
                if (curLen + fragLen >= a.maxSeqLen) brokeOut = true;
                else {
                    const char *tSeq = a.s.data + seqOff(a.s, x.target);
                    curStart -= fragLen;
                    copyBytesG<64>(buf + curStart, tSeq, fragLen, lane);
                    curLen += fragLen; leftOff += fragLen;
                    if (lane == 0) atomicOr(&a.flags[x.target], 0x80u);
                }
            }
            if (haveR && !rFirst && !brokeOut) extendRight();
            waveMemSync();
            uint32_t still = 0;
            for (uint32_t i = lane; i < h; i += 64) {
                const Item x = it[i];
                if (x.state != 0) continue;
                if (brokeOut) {                                    // hits ranked below the left hit were never popped
                    const unsigned long long key = ((unsigned long long) ((uint32_t) x.score ^ 0x80000000u) << 32) | (unsigned long long) x.alnLength;
                    const bool below = key < bestL || (key == bestL && x.target > tgtL);
                    if (below) { still++; continue; }
                }
                uint32_t st = 2;
                const bool used = (i == idxR && (rFirst || !brokeOut)) || i == idxL;
                const bool notBoth = !(x.dbStart == 0 && x.qStart == 0);
                const bool rightStart = x.dbStart == 0 && (x.dbEnd != (int) x.dbLen - 1);
                const bool leftStart = x.qStart == 0 && (x.qEnd != (int) x.qLen - 1);
                if (!used && (rightStart || leftStart) && notBoth && x.target != id) {
                    const unsigned tLen = seqLen(a.s, x.target);
                    if (x.dbStart == 0) { if ((tLen - ((unsigned) x.dbEnd + 1)) > rightOff && (unsigned) x.qEnd == (querySeqLen - 1) && rightOff > 0) st = 1; }
                    else if (x.qStart == 0) { if (x.dbStart > (int) leftOff && (unsigned) x.dbEnd == (tLen - 1) && leftOff > 0) st = 1; }
                }
                it[i].state = st;
            }
            inQueue = (uint32_t) waveReduceSum((int) still);
            waveMemSync();
            if (leftOff > 0 || rightOff > 0) couldExtend = true;
            if (brokeOut && inQueue > 0) break;
            // ---- re-score deferred hits on the extended query (assembleresult.cpp:288-313) ----
            querySeqLen = (unsigned) curLen;
            const char *qs = buf + curStart;
            waveMemSync();
            // every lane re-scores its own deferred hits (32 residues per step), all hits of the round in parallel: a queue this long
            // defers dozens of hits per round, and a 50-150 residue overlap would leave most of a wavefront idle if the hits were
            // taken one after the other (same arithmetic as the wide register queues of assembleGroupKernel)
            for (uint32_t i = lane; i < h; i += 64) {
                Item x = it[i];
                if (x.state != 1) continue;
                const char *tSeq = a.s.data + seqOff(a.s, x.target);
                const unsigned tLen = seqLen(a.s, x.target);
                const int diag = (int) ((unsigned) x.qStart + leftOff) - x.dbStart;
                const unsigned dist = (unsigned) abs(diag);
                unsigned qo = 0, to = 0, len = 0; bool hit = true;
                if (diag >= 0 && dist < querySeqLen) { qo = dist; to = 0; len = min(tLen, querySeqLen - dist); }
                else if (diag < 0 && dist < tLen) { qo = 0; to = dist; len = min(tLen - dist, querySeqLen); }
                else hit = false;
                unsigned first = 0, last = 0; int sc = 0, ids = 0; int startPos = -1, endPos = -1;
                if (hit && len > 0) { scoreColumnsSerial(qs + qo, tSeq + to, len, smat, first, last, sc, ids); startPos = (int) first; endPos = (int) last; }
                const unsigned score = (unsigned) max(sc, 0);
                nResc++; nRescRes += hit ? len : 0;
                // updateAlignment
                int qS, qE, dS, dE;
                if (diag >= 0) { qS = startPos + (int) dist; qE = endPos + (int) dist; dS = startPos; dE = endPos; }
                else { qS = startPos; qE = endPos; dS = startPos + (int) dist; dE = endPos + (int) dist; }
                const float seqId = (float) ids / ((float) qE - (float) qS);
                x.seqId = seqId; x.qLen = querySeqLen; x.dbLen = tLen; x.alnLength = hit ? len : 0u;
                const float spc = (float) score / (float) ((double) x.alnLength + 0.5);
                x.score = (int) (spc * 100);
                x.qStart = qS; x.qEnd = qE; x.dbStart = dS; x.dbEnd = dE;
                x.state = (seqId >= a.seqIdThr) ? 0u : 2u;
                it[i] = x;
            }
            waveMemSync();
            // recount what is really queued (defensive: the loop condition must match the item states)
            {
                uint32_t c = 0;
                for (uint32_t i = lane; i < h; i += 64) c += (it[i].state == 0) ? 1u : 0u;
                inQueue = (uint32_t) waveReduceSum((int) c);
            }
        }
        if (couldExtend) {
            if (lane == 0) { atomicOr(&a.flags[id], 0x20u); a.newLen[id] = (uint32_t) curLen; a.newStart[id] = aoff + curStart; }
            nExt++;
        }
        waveMemSync();
    }
    nResc = waveReduceSumU64(nResc); nRescRes = waveReduceSumU64(nRescRes);        // counted per lane (every lane re-scores its own hits)
    if (lane == 0) {
        if (nExt) atomicAdd(&a.stats[0], nExt); if (nResc) atomicAdd(&a.stats[1], nResc); if (nRescRes) atomicAdd(&a.stats[2], nRescRes);
        if (nAln) { atomicAdd(&a.stats[9], nAln); atomicAdd(&a.stats[10], nQRes); atomicAdd(&a.stats[11], nRescRes); }
    }
}

// =====================================================================================================
// nuclassembleresults (src/assembler/nuclassembleresult.cpp): same greedy loop on nucleotide reads with
//   * hits on the reverse strand (qStart > qEnd): coordinates mirrored, fragments reverse-complemented (:196-224,59-68)
//   * a Bayesian comparator (beta-binomial tail via lgamma/log/exp, :36-70) that is NOT a strict weak ordering,
//     so the pop order is whatever libstdc++'s binary heap produces: std::push_heap / std::pop_heap are replayed
//     verbatim (bits/stl_heap.h __push_heap / __adjust_heap) on an index heap
//   * the length cap on both sides (:271-275,301-305); seqId of the parsed hits is not rescaled.
// The comparator's doubles come from the device math library; a decision can differ from glibc's only when p lies
// within rounding distance of 0.45 / 0.55 (the reference itself depends on the host's libm variant there).
// One wavefront per query; lane 0 runs the heap, all lanes copy and re-score.
// =====================================================================================================
__device__ __forceinline__ char nuclRevN(char c) {          // getNuclRevFragment: num2aa[reverse(aa2num[c])], X -> N
    switch (c & ~0x20) {
        case 'A': return 'T';
        case 'C': case 'M': case 'Y': case 'H': return 'G';
        case 'T': case 'U': case 'W': return 'A';
        case 'G': case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': return 'C';
        default: return 'N';
    }
}
// comparator state of one query: the first decision the table cannot answer aborts the query (it is re-run after
// the host has evaluated the tuple with its libm)
constexpr unsigned CMP_LEN = 512, CMP_MM = 8;          // memoised range: overlaps shorter than 512 columns with fewer than 8 mismatches
struct NuclCmp { const AsmArgs *a; bool abort; };
__device__ __forceinline__ uint32_t ambHash(uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
    return (a1 * 0x9E3779B1u) ^ (b1 * 0x85EBCA77u) ^ (a2 * 0xC2B2AE3Du) ^ (b2 * 0x27D4EB2Fu);
}
// posterior class of one (alpha, beta) tuple the memo does not hold yet: 0 (p < 0.45), 1 (p > 0.55), 2 (between), or -1 when the
// decision sits on a threshold and the host table lacks the tuple (the query is then re-run).  Kept out of line: it is rare
// (the memo absorbs it) and its double-precision lgamma / exp / log would otherwise dictate the register budget of the callers.
// (the value of p and the width of the band around the thresholds: posterior_class.hpp, plain C++ as well — tests/test_host.py compares it
// with the reference's formula under glibc on the host)
__device__ __attribute__((noinline)) int nuclPosteriorClassDev(unsigned alpha1, unsigned beta1, unsigned alpha2, unsigned beta2, const AsmArgs *ap) {
    const double p = nuclPosteriorP(alpha1, beta1, alpha2, beta2);
    int cls = (p < 0.45) ? 0 : ((p > 0.55) ? 1 : 2);
    // The reference's decision is glibc's rounding of p whenever p sits on a threshold (zero mismatches on both sides and
    // overlap lengths 9 : 11 give exactly 0.45).  Device and host libm agree to ~1e-15; inside a 1e-9 band the class is
    // taken from the host-evaluated table instead.
    const double AMB_EPS = nuclPosteriorBand(alpha1, beta1, alpha2, beta2);
    if (fabs(p - 0.45) < AMB_EPS || fabs(p - 0.55) < AMB_EPS) {
        const AsmArgs &a = *ap;
        if (a.ambMask) {
            uint32_t slot = ambHash(alpha1, beta1, alpha2, beta2) & a.ambMask;
            for (;;) {
                const uint32_t *k = a.ambKeys + 4 * (size_t) slot;
                if (k[0] == 0) break;
                if (k[0] == alpha1 && k[1] == beta1 && k[2] == alpha2 && k[3] == beta2) return (int) a.ambVals[slot];
                slot = (slot + 1) & a.ambMask;
            }
        }
        const uint32_t w = atomicAdd(a.needCount, 1u);
        if (w < a.needCap) { uint32_t *k = a.needKeys + 4 * (size_t) w; k[0] = alpha1; k[1] = beta1; k[2] = alpha2; k[3] = beta2; }
        return -1;
    }
    return cls;
}
__device__ __forceinline__ bool nuclLess(const Item &r1, const Item &r2, NuclCmp &c) {   // CompareNuclResultByScore (nuclassembleresult.cpp:36-70)
    if (c.abort) return false;
    const unsigned mm1 = (unsigned) ((double) ((1.0f - r1.seqId) * (float) r1.alnLength) + 0.5);
    const unsigned mm2 = (unsigned) ((double) ((1.0f - r2.seqId) * (float) r2.alnLength) + 0.5);
    // The posterior class depends on (mismatches, overlap length) of the two hits only, and read overlaps use a small part of
    // that space over and over: classes are memoised in a direct-mapped table (1 byte per tuple, 0 = not yet known; racing
    // writers store the same value).  A hit skips four lgamma and a loop of exp / log in double precision.
    const bool cacheable = c.a->cmpCache && mm1 < CMP_MM && mm2 < CMP_MM && r1.alnLength < CMP_LEN && r2.alnLength < CMP_LEN;
    const size_t cidx = cacheable ? ((((size_t) r1.alnLength * CMP_MM + mm1) * CMP_LEN + r2.alnLength) * CMP_MM + mm2) : 0;
    int cls = cacheable ? (int) c.a->cmpCache[cidx] - 1 : -1;
    if (cls < 0) {
        cls = nuclPosteriorClassDev(mm1 + 1, r1.alnLength - mm1 + 1, mm2 + 1, r2.alnLength - mm2 + 1, c.a);
        if (cls < 0) { c.abort = true; return false; }
        if (cacheable) c.a->cmpCache[cidx] = (uint8_t) (cls + 1);     // (classes on a threshold come from the host table and are stored alike)
    }
    if (cls == 0) return true;
    if (cls == 1) return false;
    if (r1.dbLen - r1.alnLength < r2.dbLen - r2.alnLength) return true;
    if (r1.dbLen - r1.alnLength > r2.dbLen - r2.alnLength) return false;
    return true;
}
__device__ void heapPush(uint32_t *hp, uint32_t &n, uint32_t v, const Item *it, NuclCmp &c) {      // vector::push_back + std::push_heap
    uint32_t hole = n++;
    while (hole > 0) {
        const uint32_t parent = (hole - 1) / 2;
        if (!nuclLess(it[hp[parent]], it[v], c)) break;
        hp[hole] = hp[parent]; hole = parent;
    }
    hp[hole] = v;
}
__device__ uint32_t heapPop(uint32_t *hp, uint32_t &n, const Item *it, NuclCmp &c) {               // top() + std::pop_heap + pop_back
    const uint32_t top = hp[0];
    if (n > 1) {
        const uint32_t value = hp[n - 1];
        const uint32_t len = n - 1;
        uint32_t hole = 0, second = 0;
        while ((int64_t) second < ((int64_t) len - 1) / 2) {
            second = 2 * (second + 1);
            if (nuclLess(it[hp[second]], it[hp[second - 1]], c)) second--;
            hp[hole] = hp[second]; hole = second;
        }
        if ((len & 1) == 0 && (int64_t) second == ((int64_t) len - 2) / 2) {
            second = 2 * (second + 1);
            hp[hole] = hp[second - 1]; hole = second - 1;
        }
        while (hole > 0) {                                         // __push_heap(first, hole, 0, value)
            const uint32_t parent = (hole - 1) / 2;
            if (!nuclLess(it[hp[parent]], it[value], c)) break;
            hp[hole] = hp[parent]; hole = parent;
        }
        hp[hole] = value;
    }
    n--;
    return top;
}

// mode-3 score of query vs (optionally reverse-complemented) target on one diagonal + the counts updateNuclAlignment needs
__device__ __forceinline__ Rescored rescoreOnDiagonalNucl(const char *q, unsigned qLen, const char *t, unsigned tLen, int diagonal,
                                                          const signed char *smat, bool rev) {
    Rescored r; r.startPos = -1; r.endPos = -1; r.score = 0; r.diagonalLen = 0; r.idExcl = 0;
    const unsigned dist = (unsigned) abs(diagonal);
    unsigned qo, to, len;
    if (diagonal >= 0 && dist < qLen) { qo = dist; to = 0; len = min(tLen, qLen - dist); }
    else if (diagonal < 0 && dist < tLen) { qo = 0; to = dist; len = min(tLen - dist, qLen); }
    else return r;
    r.diagonalLen = len;
    if (len == 0) return r;
    auto T = [&](unsigned i) -> char { return rev ? nuclRevN(t[tLen - 1 - (to + i)]) : t[to + i]; };
    unsigned first = (q[qo] == '*' || T(0) == '*') ? 1u : 0u;
    unsigned last = len - 1;
    if (last > 0 && (q[qo + len - 1] == '*' || T(len - 1) == '*')) last--;
    int s = 0, ids = 0;
    for (unsigned p = first + (unsigned) laneId(); p <= last; p += 64) {
        const char a = q[qo + p], b = T(p);
        s += (int) smat[(int) a * 123 + (int) b];
        if (p < last) ids += (a == b) ? 1 : 0;
    }
    s = waveReduceSum(s); ids = waveReduceSum(ids);
    r.score = (unsigned) max(s, 0); r.startPos = (int) first; r.endPos = (int) last; r.idExcl = ids;
    return r;
}

// GUIDED = guidedassembleresults (src/assembler/guidedassembleresult.cpp:136-343): the same loop on nucleotide ORFs without
// strand handling; hits are taken as parsed (no rescaling, :194-205 drops those below --min-seq-id), an extension never
// crosses a '*' of the protein twins (:183-184,234-244) and is mirrored on the twin (:258-259,267-271,291-296).
template <bool GUIDED>
__global__ __launch_bounds__(64) void assembleNuclKernel(AsmArgs a) {
    __shared__ signed char smat[123 * 123 + 7];
    __shared__ uint32_t sPop;
    __shared__ int sAbort;
    __shared__ uint32_t sPushed;
    stageScoreTable(smat, a.mat, 64);
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned long long nExt = 0, nResc = 0, nRescRes = 0, nAln = 0, nQRes = 0;
    const uint32_t nWork = a.queryList ? a.nQueryList : a.s.n;
    NuclCmp cmp; cmp.a = &a; cmp.abort = false;
    for (uint32_t w = blockIdx.x; w < nWork; w += gridDim.x) {
        const uint32_t id = a.queryList ? a.queryList[w] : w;
        const uint64_t h0 = a.qoff[id], h1 = a.qoff[id + 1];
        const uint32_t h = (uint32_t) (h1 - h0);
        if (h == 0) continue;
        const uint64_t aoff = a.arenaOff[id];
        if (a.arenaOff[id + 1] == aoff) continue;          // only the self hit: nothing can happen
        Item *it = a.items + h0;
        uint32_t *hp = a.heap + 3 * h0;          // binary heap of item indices
        uint32_t *def = hp + h;                  // tmpAlignments: deferred items in pop order
        uint32_t *used = def + h;                // targets attached so far (flag 0x80 is committed when the query completes)
        uint32_t nUsed = 0;
        cmp.abort = false;
        if (lane == 0) sAbort = 0;
        unsigned long long qResc = 0, qRescRes = 0;
        const char *orig = a.s.data + seqOff(a.s, id);
        unsigned querySeqLen = seqLen(a.s, id);
        // ---- queue fill (nuclassembleresult.cpp:196-224); pad = useReverse of the hit's target ----
        for (uint32_t i = lane; i < h; i += 64) {
            const AlnRec r = a.recs[h0 + i];
            Item x;
            x.target = r.target;
            const int aq = (r.qStart == -1) ? 0 : r.qStart, ad = (r.dbStart == -1) ? 0 : r.dbStart;
            x.alnLength = (uint32_t) (max(abs(r.qEnd - aq), abs(r.dbEnd - ad)) + 1);
            const int rawScore = (int) (fma((double) r.bitScore, a.ln2, a.logK) / a.lambda + 0.5);
            const float scorePerCol = (float) rawScore / (float) ((double) x.alnLength + 0.5);
            x.seqId = r.fromText ? r.seqId : seqIdThroughText(r.seqId);
            x.score = GUIDED ? r.bitScore : (int) (scorePerCol * 100);
            x.qStart = r.qStart; x.qEnd = r.qEnd; x.qLen = (uint32_t) r.qLen; x.dbStart = r.dbStart; x.dbEnd = r.dbEnd; x.dbLen = (uint32_t) r.dbLen;
            x.pad = 0;
            if (!GUIDED && x.qStart > x.qEnd) {
                x.pad = 1;
                const int t0 = x.qStart; x.qStart = x.qEnd; x.qEnd = t0;
                const unsigned dbs = (unsigned) x.dbStart;
                x.dbStart = (int) (x.dbLen - (unsigned) x.dbEnd - 1);
                x.dbEnd = (int) (x.dbLen - dbs - 1);
            }
            x.state = (!r.accepted || (GUIDED && x.seqId < a.seqIdThr)) ? 2u : 0u;     // a hole of a sparse list; guided: re-evaluated on the nucleotide level, never queued
            it[i] = x;
        }
        __syncthreads();
        uint32_t nHeap = 0;
        if (lane == 0) {
            for (uint32_t i = 0; i < h && !cmp.abort; i++) if (it[i].state == 0) heapPush(hp, nHeap, i, it, cmp);
            if (cmp.abort) sAbort = 1;
            sPushed = nHeap;
        }
        __syncthreads();
        nHeap = sPushed;
        // guided: the protein twin of the query and its own arena slice
        const char *aaQ = nullptr; char *aaBuf = nullptr; uint64_t aaStart = 0, aaLen = 0; bool exclL = false, exclR = false;
        if (GUIDED) {
            aaQ = a.aa.data + a.aa.off[id]; aaLen = a.aa.len[id];
            exclL = aaQ[0] == '*'; exclR = aaQ[aaLen - 1] == '*';
            aaBuf = a.aaArena + a.aaArenaOff[id]; aaStart = a.aaLeftCap[id];
            copyBytesG<64>(aaBuf + aaStart, aaQ, (unsigned) aaLen, lane);
        }
        char *buf = a.arena + aoff;
        uint64_t curStart = a.leftCap[id];
        // the query is copied into its arena slice when the first fragment is attached: most queries that pass the pre-screen still end
        // without an extension (their extendable hit is not the one the comparator pops first, or it names a consumed side)
        const unsigned origLen = querySeqLen;
        bool copied = false;
        uint64_t curLen = querySeqLen;
        __syncthreads();
        bool couldExtend = false;
        bool aborted = sAbort != 0;
        while (nHeap > 0 && !aborted) {
            unsigned leftOff = 0, rightOff = 0;
            bool brokeOut = false;
            uint32_t nDef = 0;
            while (nHeap > 0) {
                // ---- selectNuclFragmentToExtend: pop the heap's top until one is extendable ----
                if (lane == 0) { uint32_t n2 = nHeap; sPop = heapPop(hp, n2, it, cmp); if (cmp.abort) sAbort = 1; }
                nHeap--;
                __syncthreads();
                if (sAbort) { aborted = true; break; }
                const uint32_t bi = sPop;
                const Item best = it[bi];
                __syncthreads();
                const bool notBoth = !(best.dbStart == 0 && best.qStart == 0);
                const bool rightStart = best.dbStart == 0 && (best.dbEnd != (int) best.dbLen - 1);
                const bool leftStart = best.qStart == 0 && (best.qEnd != (int) best.qLen - 1);
                if (!((rightStart || leftStart) && notBoth && best.target != id)) continue;
                const char *tSeq = a.s.data + seqOff(a.s, best.target);
                const unsigned tLen = seqLen(a.s, best.target);
                const bool rev = best.pad != 0;
                const char *aaT = nullptr; unsigned aaTLen = 0;
                if (GUIDED) { aaT = a.aa.data + a.aa.off[best.target]; aaTLen = a.aa.len[best.target]; }
                if (best.dbStart == 0) { if ((tLen - ((unsigned) best.dbEnd + 1)) <= rightOff || (GUIDED && (exclR || aaT[0] == '*'))) continue; }
                else if (best.qStart == 0) { if (best.dbStart <= (int) leftOff || (GUIDED && (exclL || aaT[aaTLen - 1] == '*'))) continue; }
                const unsigned dbStart = (unsigned) best.dbStart, dbEnd = (unsigned) best.dbEnd, qStart = (unsigned) best.qStart, qEnd = (unsigned) best.qEnd;
                if (dbStart == 0 && qEnd == (querySeqLen - 1)) {            // right extension
                    if (rightOff > 0) { if (lane == 0) def[nDef] = bi; nDef++; continue; }
                    const unsigned fragLen = tLen - (dbEnd + 1);
                    if (curLen + fragLen >= a.maxSeqLen) { brokeOut = true; break; }
                    if (!copied) { copyBytesG<64>(buf + curStart, orig, origLen, lane); copied = true; }
                    for (unsigned i = lane; i < fragLen; i += 64)
                        buf[curStart + curLen + i] = rev ? nuclRevN(tSeq[fragLen - 1 - i]) : tSeq[dbEnd + 1 + i];
                    curLen += fragLen; rightOff += fragLen;
                    if (GUIDED) {                                    // protein twin: aaTargetSeq + dbEnd/3 + 1, (tLen/3 - dbEnd/3) - 1 bytes
                        const unsigned aaFrag = (tLen / 3 - dbEnd / 3) - 1;
                        if (aaStart + aaLen + aaFrag > a.aaArenaOff[id + 1] - a.aaArenaOff[id]) { if (lane == 0) atomicAdd(&a.stats[12], 1ull); }
                        else { copyBytesG<64>(aaBuf + aaStart + aaLen, aaT + dbEnd / 3 + 1, aaFrag, lane); aaLen += aaFrag; }
                    }
                    if (lane == 0) used[nUsed] = best.target;
                    nUsed++;
                } else if (qStart == 0 && dbEnd == (tLen - 1)) {            // left extension
                    if (leftOff > 0) { if (lane == 0) def[nDef] = bi; nDef++; continue; }
                    const unsigned fragLen = dbStart;
                    if (curLen + fragLen >= a.maxSeqLen) { brokeOut = true; break; }
                    if (!copied) { copyBytesG<64>(buf + curStart, orig, origLen, lane); copied = true; }
                    curStart -= fragLen;
                    for (unsigned i = lane; i < fragLen; i += 64)
                        buf[curStart + i] = rev ? nuclRevN(tSeq[(tLen - dbStart) + (fragLen - 1 - i)]) : tSeq[i];
                    curLen += fragLen; leftOff += fragLen;
                    if (GUIDED) {                                    // protein twin: the first fragLen/3 (+1 behind a start '*') residues
                        const unsigned aaFrag = fragLen / 3 + ((aaT[0] == '*') ? 1u : 0u);
                        if (aaFrag > aaStart) { if (lane == 0) atomicAdd(&a.stats[12], 1ull); }
                        else { aaStart -= aaFrag; copyBytesG<64>(aaBuf + aaStart, aaT, aaFrag, lane); aaLen += aaFrag; }
                    }
                    if (lane == 0) used[nUsed] = best.target;
                    nUsed++;
                }
                __syncthreads();
            }
            if (aborted) break;
            if (leftOff > 0 || rightOff > 0) couldExtend = true;
            if (brokeOut && nHeap > 0) break;
            // ---- re-score deferred hits on the extended query (nuclassembleresult.cpp:332-355) in deferral order ----
            querySeqLen = (unsigned) curLen;
            const char *qs = buf + curStart;
            __syncthreads();
            for (uint32_t d = 0; d < nDef; d++) {
                const uint32_t found = def[d];
                Item x = it[found];
                const char *tSeq = a.s.data + seqOff(a.s, x.target);
                const unsigned tLen = seqLen(a.s, x.target);
                const int diag = (int) ((unsigned) x.qStart + leftOff) - x.dbStart;
                const Rescored rs = rescoreOnDiagonalNucl(qs, querySeqLen, tSeq, tLen, diag, smat, x.pad != 0);
                qResc++; qRescRes += rs.diagonalLen;
                const int dist = abs(diag);
                int qS, qE, dS, dE;
                if (diag >= 0) { qS = rs.startPos + dist; qE = rs.endPos + dist; dS = rs.startPos; dE = rs.endPos; }
                else { qS = rs.startPos; qE = rs.endPos; dS = rs.startPos + dist; dE = rs.endPos + dist; }
                const float seqId = (float) rs.idExcl / ((float) qE - (float) qS);
                x.seqId = seqId; x.qLen = querySeqLen; x.dbLen = tLen; x.alnLength = rs.diagonalLen;
                const float spc = (float) rs.score / (float) ((double) x.alnLength + 0.5);
                x.score = (int) (spc * 100);
                x.qStart = qS; x.qEnd = qE; x.dbStart = dS; x.dbEnd = dE;
                const bool requeue = seqId >= a.seqIdThr;
                __syncthreads();
                if (lane == 0) { it[found] = x; if (requeue) { uint32_t n2 = nHeap; heapPush(hp, n2, found, it, cmp); if (cmp.abort) sAbort = 1; } }
                if (requeue) nHeap++;
                __syncthreads();
                if (sAbort) { aborted = true; break; }
            }
        }
        __syncthreads();
        if (aborted) {                                       // nothing of this query has been published
            if (lane == 0) a.redoList[atomicAdd(a.redoCount, 1u)] = id;
        } else {
            nAln += h; nQRes += seqLen(a.s, id); nResc += qResc; nRescRes += qRescRes;
            for (uint32_t i = lane; i < nUsed; i += 64) atomicOr(&a.flags[used[i]], 0x80u);
            if (couldExtend) {
                if (lane == 0) {
                    atomicOr(&a.flags[id], 0x20u); a.newLen[id] = (uint32_t) curLen; a.newStart[id] = aoff + curStart;
                    if (GUIDED) { a.aaNewLen[id] = (uint32_t) aaLen; a.aaNewStart[id] = a.aaArenaOff[id] + aaStart; }
                }
                nExt++;
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        if (nExt) atomicAdd(&a.stats[0], nExt); if (nResc) atomicAdd(&a.stats[1], nResc); if (nRescRes) atomicAdd(&a.stats[2], nRescRes);
        if (nAln) { atomicAdd(&a.stats[3], nAln); atomicAdd(&a.stats[4], nQRes); atomicAdd(&a.stats[5], nRescRes); }
    }
}

// ---- nucleotide / guided variants, ONE THREAD per query (queries with up to a few hundred hits: the reads and most contigs).
//      The loop is the reference's, statement by statement; what a wavefront per query wastes here is 63 lanes during every
//      heap operation (each a chain of dependent loads).  Queue, deferral list and hit records live in the query's slice of
//      the global scratch arrays like in assembleNuclKernel; copies and re-scoring walk 8 / 32 bytes per step. ----
__device__ __forceinline__ void copyBytesSerial(char *dst, const char *src, unsigned n) {
    unsigned i = 0;
    for (; i + 8 <= n; i += 8) storeU64Unaligned(dst + i, loadU64Unaligned(src + i));
    for (; i < n; i++) dst[i] = src[i];
}
// Reverse-strand targets eight residues per load: the word that ENDS at the mirrored position, byte-swapped, every byte through the
// complement table in LDS.
// bytes src[hi - 7 .. hi] in reverse order (byte j of the result = src[hi - j]); for hi < 7 the bytes below src[0] are not touched
// and the upper bytes of the result are unspecified
__device__ __forceinline__ uint64_t loadRevWord(const char *src, unsigned hi) {
    if (hi >= 7) return __builtin_bswap64(loadU64Unaligned(src + (hi - 7)));
    return __builtin_bswap64(loadU64Unaligned(src)) >> (8u * (7u - hi));
}
__device__ __forceinline__ uint64_t compWord(uint64_t w, const unsigned char *sComp) {
    uint64_t r = 0;
#pragma unroll
    for (unsigned j = 0; j < 8; j++) r |= (uint64_t) sComp[(unsigned) (w >> (8 * j)) & 0xFFu] << (8 * j);
    return r;
}
// ---- cooperative halves of the kernel below: one job of ONE lane at a time, worked on by all 64 lanes (coalesced) ----
template <typename T> __device__ __forceinline__ T bcast(T v, int srcLane) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "bcast");
    if (sizeof(T) == 4) { int x; __builtin_memcpy(&x, &v, 4); x = __builtin_amdgcn_readlane(x, srcLane); T r; __builtin_memcpy(&r, &x, 4); return r; }
    int x[2]; __builtin_memcpy(x, &v, 8); x[0] = __builtin_amdgcn_readlane(x[0], srcLane); x[1] = __builtin_amdgcn_readlane(x[1], srcLane);
    T r; __builtin_memcpy(&r, x, 8); return r;
}
// dst[i] = nuclRevN(src[n - 1 - i]), 8 bytes per lane and step
__device__ __forceinline__ void copyRevCompWave(char *dst, const char *src, unsigned n, const unsigned char *sComp, int lane) {
    for (unsigned i = 8u * (unsigned) lane; i < n; i += 512u) {
        const uint64_t w = compWord(loadRevWord(src, n - 1 - i), sComp);
        if (i + 8 <= n) storeU64Unaligned(dst + i, w);
        else storeTail(dst + i, w, n - i);
    }
}
// scoreColumnsG<64> against the reverse complement of the target: column p is nuclRevN(tSeq[tHi - p]); partial sums per lane
__device__ __forceinline__ void scoreColumnsRevWave(const char *q, const char *tSeq, unsigned tHi, unsigned len, const signed char *smat, const unsigned char *sComp, int lane,
                                                    unsigned &first, unsigned &last, int &s, int &ids) {
    const char q0 = q[0], qe = q[len - 1];
    const char t0 = (char) sComp[(unsigned char) tSeq[tHi]], te = (char) sComp[(unsigned char) tSeq[tHi - (len - 1)]];
    first = (q0 == '*' || t0 == '*') ? 1u : 0u;
    last = len - 1;
    if (last > 0 && (qe == '*' || te == '*')) last--;
    for (unsigned p = 8u * (unsigned) lane; p < len; p += 512u) {
        const uint64_t qw = loadU64Unaligned(q + p), tc = compWord(loadRevWord(tSeq, tHi - p), sComp);
        const int lo = (int) first - (int) p, hi = (int) last - (int) p;
        const uint64_t m = byteRangeMask(lo, hi + 1);
        ids += zeroBytes(qw ^ tc, byteRangeMask(lo, hi));
        const uint64_t qv = qw & m, tv = tc & m;
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
            const unsigned a = (unsigned) (qv >> (8 * j)) & 0xFFu, b = (unsigned) (tv >> (8 * j)) & 0xFFu;
            s += (int) smat[a * 123 + b];
        }
    }
}

// ---- nucleotide / guided variants, ONE LANE per queue, ONE WAVEFRONT per copy and per re-scored overlap (round 4).
//      Rounds 2-3 ran the reference's loop statement by statement in one thread per query: what a wavefront per query wastes is 63
//      lanes during every heap operation (a chain of dependent loads and, for overlaps the memo does not cover, a posterior class in
//      double precision), so the heaps stay one per lane.  But a lane that copies a fragment or walks an overlap alone issues loads
//      whose 64 addresses lie in 64 different lines, 2 048 such streams per CU evict each other from the L1, and the lanes of a
//      wavefront wait for the one with the longest overlaps: in the late nucleotide iterations of configs[4] (contigs of 5-100 kb
//      with up to 256 hits) the kernel spent 700 ms per call that way.  Now a round of the loop is three phases per wavefront:
//      (1) every lane pops its own heap and only RECORDS what it attaches (at most the query itself, one fragment per side);
//      (2) the recorded copies, then (3) the deferred hits' overlaps, are worked through one lane's job at a time by all 64 lanes,
//      8 bytes per lane and step, partial scores reduced across the wavefront; the owner takes the result and pushes the hit back.
//      Queue, deferral list and hit records live in the query's slice of the global scratch arrays like in assembleNuclKernel. ----
// 256 threads share one score table: the number of queues in flight per CU is what counts in phase 1
constexpr int NT_BLOCK = 256;
struct CopyJob { char *dst; const char *src; unsigned n; unsigned rev; };
template <bool GUIDED>
__global__ __launch_bounds__(NT_BLOCK, 4) void assembleNuclThreadKernel(AsmArgs a) {    // 4 blocks per CU: at most 128 VGPRs
    __shared__ signed char smat[123 * 123 + 7];
    __shared__ unsigned char sComp[256];                     // nuclRevN of every byte
    stageScoreTable(smat, a.mat, NT_BLOCK);
    for (int i = threadIdx.x; i < 256; i += NT_BLOCK) sComp[i] = (unsigned char) nuclRevN((char) i);
    __syncthreads();
    const int lane = laneId();
    unsigned long long nExt = 0, nResc = 0, nRescRes = 0, nAln = 0, nQRes = 0;
    NuclCmp cmp; cmp.a = &a; cmp.abort = false;
    for (uint32_t w0 = blockIdx.x * NT_BLOCK + (threadIdx.x & ~63u); w0 < a.nQueryList; w0 += gridDim.x * NT_BLOCK) {      // wavefront-uniform
        const uint32_t w = w0 + (uint32_t) lane;
        const bool valid = w < a.nQueryList;
        const uint32_t id = valid ? a.queryList[w] : 0u;
        const uint64_t h0 = valid ? a.qoff[id] : 0ull;
        const uint32_t h = valid ? (uint32_t) (a.qoff[id + 1] - h0) : 0u;
        const uint64_t aoff = valid ? a.arenaOff[id] : 0ull;
        Item *it = a.items + h0;
        uint32_t *hp = a.heap + 3 * h0, *def = hp + h, *used = def + h;
        uint32_t nUsed = 0;
        cmp.abort = false;
        unsigned long long qResc = 0, qRescRes = 0;
        const char *orig = a.s.data + (valid ? seqOff(a.s, id) : 0ull);
        unsigned querySeqLen = valid ? seqLen(a.s, id) : 0u;
        uint32_t nHeap = 0;
        for (uint32_t i = 0; i < h && !cmp.abort; i++) {                 // queue fill
            const AlnRec r = a.recs[h0 + i];
            Item x;
            x.target = r.target;
            const int aq = (r.qStart == -1) ? 0 : r.qStart, ad = (r.dbStart == -1) ? 0 : r.dbStart;
            x.alnLength = (uint32_t) (max(abs(r.qEnd - aq), abs(r.dbEnd - ad)) + 1);
            const int rawScore = (int) (fma((double) r.bitScore, a.ln2, a.logK) / a.lambda + 0.5);
            const float scorePerCol = (float) rawScore / (float) ((double) x.alnLength + 0.5);
            x.seqId = r.fromText ? r.seqId : seqIdThroughText(r.seqId);
            x.score = GUIDED ? r.bitScore : (int) (scorePerCol * 100);
            x.qStart = r.qStart; x.qEnd = r.qEnd; x.qLen = (uint32_t) r.qLen; x.dbStart = r.dbStart; x.dbEnd = r.dbEnd; x.dbLen = (uint32_t) r.dbLen;
            x.pad = 0;
            if (!GUIDED && x.qStart > x.qEnd) {
                x.pad = 1;
                const int t0 = x.qStart; x.qStart = x.qEnd; x.qEnd = t0;
                const unsigned dbs = (unsigned) x.dbStart;
                x.dbStart = (int) (x.dbLen - (unsigned) x.dbEnd - 1);
                x.dbEnd = (int) (x.dbLen - dbs - 1);
            }
            x.state = (!r.accepted || (GUIDED && x.seqId < a.seqIdThr)) ? 2u : 0u;
            it[i] = x;
            if (x.state == 0) heapPush(hp, nHeap, i, it, cmp);
        }
        const char *aaQ = nullptr; char *aaBuf = nullptr; uint64_t aaStart = 0, aaLen = 0; bool exclL = false, exclR = false;
        if (GUIDED && valid) {
            aaQ = a.aa.data + a.aa.off[id]; aaLen = a.aa.len[id];
            exclL = aaQ[0] == '*'; exclR = aaQ[aaLen - 1] == '*';
            aaBuf = a.aaArena + a.aaArenaOff[id]; aaStart = a.aaLeftCap[id];
            copyBytesSerial(aaBuf + aaStart, aaQ, (unsigned) aaLen);      // (the protein twins are a third of the length, their fragments a few residues)
        }
        char *buf = a.arena + aoff;
        uint64_t curStart = valid ? a.leftCap[id] : 0ull;
        // the query is copied into its arena slice when the first fragment is attached: many queries that pass the pre-screen still end
        // without an extension (their extendable hit names a target that offers nothing new)
        const unsigned origLen = querySeqLen;
        bool copied = false;
        uint64_t curLen = querySeqLen;
        bool couldExtend = false;
        bool active = valid && nHeap > 0 && !cmp.abort;
        while (__any(active)) {
            // ---- phase 1, per lane: pop until the heap is empty; attachments are recorded, not made ----
            unsigned leftOff = 0, rightOff = 0;
            bool brokeOut = false;
            uint32_t nDef = 0;
            CopyJob job[3];
            job[0].n = 0; job[1].n = 0; job[2].n = 0;
            if (active) {
                while (nHeap > 0) {
                    const uint32_t bi = heapPop(hp, nHeap, it, cmp);
                    if (cmp.abort) break;
                    const Item best = it[bi];
                    const bool notBoth = !(best.dbStart == 0 && best.qStart == 0);
                    const bool rightStart = best.dbStart == 0 && (best.dbEnd != (int) best.dbLen - 1);
                    const bool leftStart = best.qStart == 0 && (best.qEnd != (int) best.qLen - 1);
                    if (!((rightStart || leftStart) && notBoth && best.target != id)) continue;
                    const char *tSeq = a.s.data + seqOff(a.s, best.target);
                    const unsigned tLen = seqLen(a.s, best.target);
                    const bool rev = best.pad != 0;
                    const char *aaT = nullptr; unsigned aaTLen = 0;
                    if (GUIDED) { aaT = a.aa.data + a.aa.off[best.target]; aaTLen = a.aa.len[best.target]; }
                    if (best.dbStart == 0) { if ((tLen - ((unsigned) best.dbEnd + 1)) <= rightOff || (GUIDED && (exclR || aaT[0] == '*'))) continue; }
                    else if (best.qStart == 0) { if (best.dbStart <= (int) leftOff || (GUIDED && (exclL || aaT[aaTLen - 1] == '*'))) continue; }
                    const unsigned dbStart = (unsigned) best.dbStart, dbEnd = (unsigned) best.dbEnd, qStart = (unsigned) best.qStart, qEnd = (unsigned) best.qEnd;
                    if (dbStart == 0 && qEnd == (querySeqLen - 1)) {            // right extension
                        if (rightOff > 0) { def[nDef++] = bi; continue; }
                        const unsigned fragLen = tLen - (dbEnd + 1);
                        if (curLen + fragLen >= a.maxSeqLen) { brokeOut = true; break; }
                        if (!copied) { job[0].dst = buf + curStart; job[0].src = orig; job[0].n = origLen; job[0].rev = 0; copied = true; }
                        job[1].dst = buf + curStart + curLen; job[1].src = rev ? tSeq : tSeq + dbEnd + 1; job[1].n = fragLen; job[1].rev = rev ? 1u : 0u;
                        curLen += fragLen; rightOff += fragLen;
                        if (GUIDED) {
                            const unsigned aaFrag = (tLen / 3 - dbEnd / 3) - 1;
                            if (aaStart + aaLen + aaFrag > a.aaArenaOff[id + 1] - a.aaArenaOff[id]) atomicAdd(&a.stats[12], 1ull);
                            else { copyBytesSerial(aaBuf + aaStart + aaLen, aaT + dbEnd / 3 + 1, aaFrag); aaLen += aaFrag; }
                        }
                        used[nUsed++] = best.target;
                    } else if (qStart == 0 && dbEnd == (tLen - 1)) {            // left extension
                        if (leftOff > 0) { def[nDef++] = bi; continue; }
                        const unsigned fragLen = dbStart;
                        if (curLen + fragLen >= a.maxSeqLen) { brokeOut = true; break; }
                        if (!copied) { job[0].dst = buf + curStart; job[0].src = orig; job[0].n = origLen; job[0].rev = 0; copied = true; }
                        curStart -= fragLen;
                        job[2].dst = buf + curStart; job[2].src = rev ? tSeq + (tLen - dbStart) : tSeq; job[2].n = fragLen; job[2].rev = rev ? 1u : 0u;
                        curLen += fragLen; leftOff += fragLen;
                        if (GUIDED) {
                            const unsigned aaFrag = fragLen / 3 + ((aaT[0] == '*') ? 1u : 0u);
                            if (aaFrag > aaStart) atomicAdd(&a.stats[12], 1ull);
                            else { aaStart -= aaFrag; copyBytesSerial(aaBuf + aaStart, aaT, aaFrag); aaLen += aaFrag; }
                        }
                        used[nUsed++] = best.target;
                    }
                }
                if (cmp.abort) active = false;
            }
            // ---- phase 2, per wavefront: the recorded copies, one lane's job at a time ----
#pragma unroll
            for (int k = 0; k < 3; k++) {
                unsigned long long m = __ballot(job[k].n > 0);
                while (m) {
                    const int j = __ffsll((long long) m) - 1; m &= m - 1;
                    char *dst = bcast(job[k].dst, j); const char *src = bcast(job[k].src, j);
                    const unsigned n = bcast(job[k].n, j), rv = bcast(job[k].rev, j);
                    if (rv) copyRevCompWave(dst, src, n, sComp, lane); else copyBytesG<64>(dst, src, n, lane);
                }
            }
            waveMemSync();                                            // phase 3 reads what phase 2 wrote through other lanes
            if (active) {
                if (leftOff > 0 || rightOff > 0) couldExtend = true;
                if (brokeOut && nHeap > 0) active = false;
            }
            // ---- phase 3: the deferred hits on the extended query, in deferral order; overlaps walked by the wavefront ----
            querySeqLen = (unsigned) curLen;
            const char *qs = buf + curStart;
            const uint32_t myDef = active ? nDef : 0u;
            uint32_t maxDef = myDef;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) maxDef = max(maxDef, (uint32_t) __shfl_xor((int) maxDef, o, 64));
            for (uint32_t d = 0; d < maxDef; d++) {
                const bool mine = d < myDef && !cmp.abort;
                uint32_t found = 0; Item x; memset(&x, 0, sizeof(x));
                const char *tSeq = nullptr; unsigned tLen = 0, qo = 0, to = 0, len = 0, dist = 0; int diag = 0; bool hit = false;
                if (mine) {
                    found = def[d];
                    x = it[found];
                    tSeq = a.s.data + seqOff(a.s, x.target);
                    tLen = seqLen(a.s, x.target);
                    diag = (int) ((unsigned) x.qStart + leftOff) - x.dbStart;
                    dist = (unsigned) abs(diag);
                    hit = true;
                    if (diag >= 0 && dist < querySeqLen) { qo = dist; to = 0; len = min(tLen, querySeqLen - dist); }
                    else if (diag < 0 && dist < tLen) { qo = 0; to = dist; len = min(tLen - dist, querySeqLen); }
                    else hit = false;
                }
                int startPos = -1, endPos = -1, sc = 0, ids = 0;
                unsigned long long m = __ballot(mine && hit && len > 0);
                while (m) {
                    const int j = __ffsll((long long) m) - 1; m &= m - 1;
                    const char *jq = bcast(qs + qo, j); const char *jt = bcast(tSeq, j);
                    const unsigned jTo = bcast(to, j), jLen = bcast(len, j), jTLen = bcast(tLen, j), jRev = bcast((unsigned) x.pad, j);
                    unsigned first = 0, last = 0; int s = 0, idn = 0;
                    if (jRev) scoreColumnsRevWave(jq, jt, jTLen - 1 - jTo, jLen, smat, sComp, lane, first, last, s, idn);
                    else scoreColumnsG<64>(jq, jt + jTo, jLen, smat, lane, first, last, s, idn);
                    s = waveReduceSum(s); idn = waveReduceSum(idn);
                    if (lane == j) { startPos = (int) first; endPos = (int) last; sc = s; ids = idn; }
                }
                if (mine) {
                    const unsigned score = (unsigned) max(sc, 0);
                    qResc++; qRescRes += hit ? len : 0;
                    int qS, qE, dS, dE;
                    if (diag >= 0) { qS = startPos + (int) dist; qE = endPos + (int) dist; dS = startPos; dE = endPos; }
                    else { qS = startPos; qE = endPos; dS = startPos + (int) dist; dE = endPos + (int) dist; }
                    const float seqId = (float) ids / ((float) qE - (float) qS);
                    x.seqId = seqId; x.qLen = querySeqLen; x.dbLen = tLen; x.alnLength = hit ? len : 0u;
                    const float spc = (float) score / (float) ((double) x.alnLength + 0.5);
                    x.score = (int) (spc * 100);
                    x.qStart = qS; x.qEnd = qE; x.dbStart = dS; x.dbEnd = dE;
                    it[found] = x;
                    if (seqId >= a.seqIdThr) heapPush(hp, nHeap, found, it, cmp);
                }
            }
            active = active && nHeap > 0 && !cmp.abort;
        }
        if (valid) {
            if (cmp.abort) a.redoList[atomicAdd(a.redoCount, 1u)] = id;          // nothing of this query has been published
            else {
                nAln += h; nQRes += seqLen(a.s, id); nResc += qResc; nRescRes += qRescRes;
                for (uint32_t i = 0; i < nUsed; i++) atomicOr(&a.flags[used[i]], 0x80u);
                if (couldExtend) {
                    atomicOr(&a.flags[id], 0x20u); a.newLen[id] = (uint32_t) curLen; a.newStart[id] = aoff + curStart;
                    if (GUIDED) { a.aaNewLen[id] = (uint32_t) aaLen; a.aaNewStart[id] = a.aaArenaOff[id] + aaStart; }
                    nExt++;
                }
            }
        }
    }
    nExt = waveReduceSumU64(nExt); nResc = waveReduceSumU64(nResc); nRescRes = waveReduceSumU64(nRescRes); nAln = waveReduceSumU64(nAln); nQRes = waveReduceSumU64(nQRes);
    if (laneId() == 0) {
        if (nExt) atomicAdd(&a.stats[0], nExt); if (nResc) atomicAdd(&a.stats[1], nResc); if (nRescRes) atomicAdd(&a.stats[2], nRescRes);
        if (nAln) { atomicAdd(&a.stats[3], nAln); atomicAdd(&a.stats[4], nQRes); atomicAdd(&a.stats[5], nRescRes); }
    }
}

// ---- the common cases: the whole queue of a query lives in registers, one alignment per lane.
//      G = 16: four queries per wavefront (a read has a handful of overlaps); G = 64: one query per wavefront.
//      Sixteen/four independent groups per block share the LDS score table; no block barriers in the loop. ----
// reductions over a group: inside a row of 16 lanes on the VALU (DPP), across rows with shuffles
template <int G> __device__ __forceinline__ int groupSum(int v) {
    v = rowSum16(v);
#pragma unroll
    for (int o = 16; o < G; o <<= 1) v += __shfl_xor(v, o, G);
    return v;
}
template <int G> __device__ __forceinline__ unsigned long long groupMax(unsigned long long v) {
    v = rowMax16(v);
#pragma unroll
    for (int o = 16; o < G; o <<= 1) { const unsigned long long ok = __shfl_xor(v, o, G); v = (ok > v) ? ok : v; }
    return v;
}
// rank of `mine` among the group's values (number of strictly smaller ones)
template <int G> __device__ __forceinline__ uint32_t groupRank(uint32_t mine) {
    uint32_t r = rowCountLess16(mine, mine, false);
#pragma unroll
    for (int o = 16; o < G; o += 16) r += rowCountLess16((uint32_t) __shfl_xor((int) mine, o, G), mine, true);
    return r;
}
template <int G> __device__ __forceinline__ unsigned long long groupBallot(bool p) {
    const unsigned long long m = __ballot(p);
    if (G == 64) return m;
    return (m >> ((laneId() / G) * G)) & ((1ULL << G) - 1ULL);
}

template <int G>
__device__ __forceinline__ Rescored rescoreOnDiagonalG(const char *q, unsigned qLen, const char *t, unsigned tLen, int diagonal,
                                                       const signed char *smat, int gl) {
    Rescored r; r.startPos = -1; r.endPos = -1; r.score = 0; r.diagonalLen = 0; r.idExcl = 0;
    const unsigned dist = (unsigned) abs(diagonal);
    unsigned qo, to, len;
    if (diagonal >= 0 && dist < qLen) { qo = dist; to = 0; len = min(tLen, qLen - dist); }
    else if (diagonal < 0 && dist < tLen) { qo = 0; to = dist; len = min(tLen - dist, qLen); }
    else return r;
    r.diagonalLen = len;
    if (len == 0) return r;
    unsigned first, last;
    int s = 0, ids = 0;
    scoreColumnsG<G>(q + qo, t + to, len, smat, gl, first, last, s, ids);
    s = groupSum<G>(s); ids = groupSum<G>(ids);
    r.score = (unsigned) max(s, 0); r.startPos = (int) first; r.endPos = (int) last; r.idExcl = ids;
    return r;
}

template <int G, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void assembleGroupKernel(AsmArgs a) {
    __shared__ signed char smat[123 * 123 + 7];
    stageScoreTable(smat, a.mat, 256);
    __syncthreads();
    const int gl = threadIdx.x & (G - 1);                         // lane within the group
    const uint32_t groupsTotal = gridDim.x * (256 / G);
    const uint32_t nWork = (G == 64) ? a.nMid : ((G == 32) ? a.nMid32 : a.nSmall);
    unsigned long long nExt = 0, nResc = 0, nRescRes = 0, nAln = 0, nQRes = 0;
    // Round 6: a query starts with a chain of dependent round trips — work list entry -> CSR offsets, arena offset, packed (offset, length), left capacity ->
    // alignment records -> the targets' metadata -> bytes — and the counters show the wavefronts parked 57-72 % of their cycles.  The first two links travel
    // ahead: the list entry two items ahead in every lane (one register), and the five per-query words of the next item ONE WORD PER LANE (lanes 0..4 of the
    // group, two registers; ten registers per lane went to scratch and cost more than the look-ahead saved: profiles/r06_ab_knobs.txt, call 21), handed round
    // with shuffles when the item's turn comes.
    const uint32_t *workList = (G == 64) ? a.midList : ((G == 32) ? a.mid32List : a.smallList);      // work lists: arenaSizeKernel
    auto loadWord = [&](uint32_t pid) -> uint64_t {
        uint64_t v = 0;
        if (pid != 0xFFFFFFFFu) {
            if (gl == 0) v = a.qoff[pid]; else if (gl == 1) v = a.qoff[pid + 1]; else if (gl == 2) v = a.arenaOff[pid]; else if (gl == 3) v = a.s.offLen[pid]; else if (gl == 4) v = a.leftCap[pid];
        }
        return v;
    };
    uint32_t w = blockIdx.x * (256 / G) + threadIdx.x / G;
    uint32_t idNext = (w < nWork) ? workList[w] : 0xFFFFFFFFu;
    uint64_t wordNext = loadWord(idNext);
    uint32_t idAhead = (w + groupsTotal < nWork) ? workList[w + groupsTotal] : 0xFFFFFFFFu;
    for (; w < nWork; w += groupsTotal) {
        const uint32_t id = idNext;
        const uint64_t h0 = __shfl(wordNext, 0, G), h1 = __shfl(wordNext, 1, G), aoff = __shfl(wordNext, 2, G), qOffLen = __shfl(wordNext, 3, G), qLeftCap = __shfl(wordNext, 4, G);
        idNext = idAhead; wordNext = loadWord(idNext);
        idAhead = (w + 2u * groupsTotal < nWork) ? workList[w + 2u * groupsTotal] : 0xFFFFFFFFu;
        const uint32_t h = (uint32_t) (h1 - h0);
        const char *orig = a.s.data + (qOffLen >> 24);
        unsigned querySeqLen = (unsigned) ((uint32_t) qOffLen & 0xFFFFFFu);
        if (gl == 0) { nAln += h; nQRes += querySeqLen; }
        // ---- queue fill (assembleresult.cpp:161-189): lane i owns alignment i ----
        uint32_t xTarget = 0xFFFFFFFFu, xAlnLen = 0, xQLen = 0, xDbLen = 0, xState = 2, xTLen = 0; uint64_t xTOff = 0;
        int xScore = 0, xQStart = 0, xQEnd = 0, xDbStart = 0, xDbEnd = 0;
        if ((uint32_t) gl < h) {
            const AlnRec r = a.recs[h0 + gl];
            xTarget = r.target;
            const int aq = (r.qStart == -1) ? 0 : r.qStart, ad = (r.dbStart == -1) ? 0 : r.dbStart;
            xAlnLen = (uint32_t) (max(abs(r.qEnd - aq), abs(r.dbEnd - ad)) + 1);
            const int rawScore = (int) (fma((double) r.bitScore, a.ln2, a.logK) / a.lambda + 0.5);
            const float scorePerCol = (float) rawScore / (float) ((double) xAlnLen + 0.5);
            xScore = (int) (scorePerCol * 100);
            xQStart = r.qStart; xQEnd = r.qEnd; xQLen = (uint32_t) r.qLen; xDbStart = r.dbStart; xDbEnd = r.dbEnd; xDbLen = (uint32_t) r.dbLen;
            // the self alignment is popped and discarded without any effect (selectFragmentToExtend: isNotIdentity):
            // it never enters the register queue
            xState = (xTarget == id || !r.accepted) ? 2u : 0u;                 // (nor does a hole of a sparse list)
            // offset / length of the target, fetched up front (one memory round trip less per pop) — for the hits that can take one of the
            // two extension branches at all: the geometry tests below read them for nobody else, and a re-scored hit came through a branch
            // (round 5: a random 128-byte line of metadata per hit that is popped and dropped in its first round)
            const bool branchType = ((xDbStart == 0 && xDbEnd != (int) xDbLen - 1) || (xQStart == 0 && xQEnd != (int) xQLen - 1)) && !(xDbStart == 0 && xQStart == 0);
            if (xState == 0 && branchType) { xTOff = seqOff(a.s, xTarget); xTLen = seqLen(a.s, xTarget); }
        }
        // tie-break of CompareResultByScore (smaller key wins) as a rank among the group's targets
        const uint32_t tRank = groupRank<G>(xTarget);
        char *buf = a.arena + aoff;
        uint64_t curStart = qLeftCap;
        copyBytesG<G>(buf + curStart, orig, querySeqLen, gl);
        uint64_t curLen = querySeqLen;
        bool couldExtend = false;
        uint32_t inQueue = (uint32_t) __popcll(groupBallot<G>(xState == 0));
        while (inQueue > 0) {
            unsigned leftOff = 0, rightOff = 0;
            bool brokeOut = false;
            if (xState == 1) xState = 2;
            // ---- one round of the reference's pop loop (assembleresult.cpp:203-283), without popping hit by hit.
            // The comparator is a strict total order, so the round is a function of the SET of queued hits: the best hit
            // that can extend to the right does so, the best that can extend to the left does so (whichever ranks higher
            // first: the length cap of the left extension sees the right fragment only then); every other hit is popped
            // after "its" extension (a hit popped earlier would have been the extension itself, or is one the geometry
            // tests drop whatever the offsets are) and is dropped or deferred by the same tests with the final offsets.
            // O(1) reductions per round instead of one arg-max per popped hit (O(h) per round).
            unsigned long long key = 0;
            bool rType = false, lType = false, rBranch = false, lBranch = false;
            unsigned fragR = 0;
            if (xState == 0) {
                key = ((unsigned long long) ((uint32_t) xScore ^ 0x80000000u) << 32) | ((unsigned long long) xAlnLen << 6) | (unsigned long long) (63u - tRank);
                const bool notBoth = !(xDbStart == 0 && xQStart == 0);
                const bool rightStart = xDbStart == 0 && (xDbEnd != (int) xDbLen - 1);
                const bool leftStart = xQStart == 0 && (xQEnd != (int) xQLen - 1);
                if ((rightStart || leftStart) && notBoth) {
                    if (xDbStart == 0) { rBranch = true; fragR = xTLen - ((unsigned) xDbEnd + 1); rType = fragR > 0 && (unsigned) xQEnd == (querySeqLen - 1); }
                    else if (xQStart == 0) { lBranch = true; lType = xDbStart > 0 && (unsigned) xDbEnd == (xTLen - 1); }
                }
            }
            const unsigned long long bestR = groupMax<G>(rType ? key : 0ULL), bestL = groupMax<G>(lType ? key : 0ULL);
            const bool rFirst = bestR != 0 && bestR > bestL;
            const bool isR = rType && key == bestR, isL = lType && key == bestL;
            auto extendRight = [&]() {
                const int bi = __ffsll((long long) groupBallot<G>(isR)) - 1;
                const char *tSeq = a.s.data + __shfl(xTOff, bi, G);
                const uint32_t bTarget = __shfl(xTarget, bi, G);
                const unsigned fragLen = __shfl(fragR, bi, G), dbEnd = (unsigned) __shfl(xDbEnd, bi, G);
                copyBytesG<G>(buf + curStart + curLen, tSeq + dbEnd + 1, fragLen, gl);
                curLen += fragLen; rightOff += fragLen;
                if (gl == 0) atomicOr(&a.flags[bTarget], 0x80u);
            };
            if (rFirst) extendRight();
            if (bestL != 0) {
                const int bi = __ffsll((long long) groupBallot<G>(isL)) - 1;
                const unsigned fragLen = (unsigned) __shfl(xDbStart, bi, G);
                if (curLen + fragLen >= a.maxSeqLen) brokeOut = true;                  // assembleresult.cpp:259-263
                else {
                    const char *tSeq = a.s.data + __shfl(xTOff, bi, G);
                    const uint32_t bTarget = __shfl(xTarget, bi, G);
                    curStart -= fragLen;
                    copyBytesG<G>(buf + curStart, tSeq, fragLen, gl);
                    curLen += fragLen; leftOff += fragLen;
                    if (gl == 0) atomicOr(&a.flags[bTarget], 0x80u);
                }
            }
            if (bestR != 0 && !rFirst && !brokeOut) extendRight();
            // every hit of the round that was popped: used, dropped or deferred
            if (xState == 0 && (!brokeOut || key >= bestL)) {       // at the length cap the hits ranked below the left hit stay queued
                uint32_t st = 2;
                if (!(isR && (rFirst || !brokeOut)) && !isL) {
                    if (rBranch) { if (fragR > rightOff && (unsigned) xQEnd == (querySeqLen - 1) && rightOff > 0) st = 1; }
                    else if (lBranch) { if (xDbStart > (int) leftOff && (unsigned) xDbEnd == (xTLen - 1) && leftOff > 0) st = 1; }
                }
                xState = st;
            }
            inQueue = (uint32_t) __popcll(groupBallot<G>(xState == 0));
            if (leftOff > 0 || rightOff > 0) couldExtend = true;
            if (brokeOut && inQueue > 0) break;
            // ---- re-score deferred hits on the extended query (assembleresult.cpp:288-313) ----
            querySeqLen = (unsigned) curLen;
            const char *qs = buf + curStart;
            unsigned long long deferred = groupBallot<G>(xState == 1);
            if (deferred) waveMemSync();
            if (G >= 32 && (uint32_t) __popcll(deferred) >= a.ownLaneMin) {
                // wide queues: every deferred lane re-scores its own hit (8 residues per step), all hits of the round in
                // parallel and without cross-lane traffic; with a dozen or more deferred hits this beats taking them one
                // after the other with the whole group (most of whose lanes idle on a 50-150 residue overlap)
                if (xState == 1) {
                    const char *tSeq = a.s.data + xTOff;
                    const int diag = (int) ((unsigned) xQStart + leftOff) - xDbStart;
                    const unsigned dist = (unsigned) abs(diag);
                    unsigned qo = 0, to = 0, len = 0; bool hit = true;
                    if (diag >= 0 && dist < querySeqLen) { qo = dist; to = 0; len = min(xTLen, querySeqLen - dist); }
                    else if (diag < 0 && dist < xTLen) { qo = 0; to = dist; len = min(xTLen - dist, querySeqLen); }
                    else hit = false;
                    unsigned first = 0, last = 0; int sc = 0, ids = 0; int startPos = -1, endPos = -1;
                    if (hit && len > 0) { scoreColumnsSerial(qs + qo, tSeq + to, len, smat, first, last, sc, ids); startPos = (int) first; endPos = (int) last; }
                    const unsigned score = (unsigned) max(sc, 0);
                    nResc += 1; nRescRes += hit ? len : 0;
                    int qS, qE, dS, dE;
                    if (diag >= 0) { qS = startPos + (int) dist; qE = endPos + (int) dist; dS = startPos; dE = endPos; }
                    else { qS = startPos; qE = endPos; dS = startPos + (int) dist; dE = endPos + (int) dist; }
                    const float seqId = (float) ids / ((float) qE - (float) qS);
                    const float spc = (float) score / (float) ((double) (hit ? len : 0u) + 0.5);
                    xQLen = querySeqLen; xDbLen = xTLen; xAlnLen = hit ? len : 0u; xScore = (int) (spc * 100);
                    xQStart = qS; xQEnd = qE; xDbStart = dS; xDbEnd = dE;
                    xState = (seqId >= a.seqIdThr) ? 0u : 2u;
                }
                deferred = 0;
            }
            while (deferred) {
                const int dl = __ffsll((long long) deferred) - 1;
                deferred &= deferred - 1;
                const int dqs = __shfl(xQStart, dl, G), dds = __shfl(xDbStart, dl, G);
                const char *tSeq = a.s.data + __shfl(xTOff, dl, G);
                const unsigned tLen = __shfl(xTLen, dl, G);
                const int diag = (int) ((unsigned) dqs + leftOff) - dds;
                const Rescored rs = rescoreOnDiagonalG<G>(qs, querySeqLen, tSeq, tLen, diag, smat, gl);
                nResc += (gl == 0); nRescRes += (gl == 0) ? rs.diagonalLen : 0;
                const int dist = abs(diag);
                int qS, qE, dS, dE;
                if (diag >= 0) { qS = rs.startPos + dist; qE = rs.endPos + dist; dS = rs.startPos; dE = rs.endPos; }
                else { qS = rs.startPos; qE = rs.endPos; dS = rs.startPos + dist; dE = rs.endPos + dist; }
                const float seqId = (float) rs.idExcl / ((float) qE - (float) qS);
                const float spc = (float) rs.score / (float) ((double) rs.diagonalLen + 0.5);
                if (gl == dl) {
                    xQLen = querySeqLen; xDbLen = tLen; xAlnLen = rs.diagonalLen; xScore = (int) (spc * 100);
                    xQStart = qS; xQEnd = qE; xDbStart = dS; xDbEnd = dE;
                    xState = (seqId >= a.seqIdThr) ? 0u : 2u;
                }
            }
            inQueue = (uint32_t) __popcll(groupBallot<G>(xState == 0));
        }
        if (couldExtend) {
            if (gl == 0) { atomicOr(&a.flags[id], 0x20u); a.newLen[id] = (uint32_t) curLen; a.newStart[id] = aoff + curStart; nExt++; }
        }
    }
    nExt = waveReduceSumU64(nExt); nResc = waveReduceSumU64(nResc); nRescRes = waveReduceSumU64(nRescRes); nAln = waveReduceSumU64(nAln); nQRes = waveReduceSumU64(nQRes);
    if (laneId() == 0) {
        if (nExt) atomicAdd(&a.stats[0], nExt); if (nResc) atomicAdd(&a.stats[1], nResc); if (nRescRes) atomicAdd(&a.stats[2], nRescRes);
        const int tier = (G == 16) ? 0 : 1;      // the 32- and 64-lane kernels share tier 1 in the statistics
        if (nAln) { atomicAdd(&a.stats[3 + 3 * tier], nAln); atomicAdd(&a.stats[4 + 3 * tier], nQRes); atomicAdd(&a.stats[5 + 3 * tier], nRescRes); }
    }
}

// Pass 1, one thread per ALIGNMENT (a wavefront reads 64 records = 4 KB in a row; one thread per query walking its records was 64
// lines per load instruction: 9 ms for 226 M records): the alignments of a query are contiguous, so the lanes of a wavefront form
// segments by query — segmented sums of the target lengths, one atomic per query and wavefront.
// mirror: nuclassembleresults puts a reverse-strand hit (qStart > qEnd) on the query's strand before anything looks at its coordinates
// (nuclassembleresult.cpp:207-216); the pre-screen tests the mirrored coordinates then.
__global__ __launch_bounds__(256) void arenaSumKernel(const AlnRec *__restrict__ recs, uint64_t nAln, uint64_t maxSeqLen, unsigned long long *__restrict__ qSum,
                                                      unsigned long long *__restrict__ qSumAa, uint32_t *__restrict__ qCan, int mirror) {
    const int lane = laneId();
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nAln; i += (uint64_t) gridDim.x * blockDim.x) {
        const AlnRec r = recs[i];
        const uint32_t id = r.query;
        uint32_t add = 0, addAa = 0; bool can = false;
        if (r.target != id && r.accepted) {                                              // (a hole of a sparse list counts for nothing)
            add = (uint32_t) r.dbLen; addAa = (uint32_t) r.dbLen / 3 + 2;               // guided: a twin fragment is at most dbLen/3 + 1 residues
            // exact pre-screen: the first extension of a query is decided by coordinates the alignment already
            // carries (selectFragmentToExtend + the two geometry tests, assembleresult.cpp:40-57,211-263); if no
            // alignment can start an extension the greedy loop drains its queue without changing anything.
            int qS = r.qStart, qE = r.qEnd, dS = r.dbStart, dE = r.dbEnd;
            if (mirror && qS > qE) { qS = r.qEnd; qE = r.qStart; dS = r.dbLen - r.dbEnd - 1; dE = r.dbLen - r.dbStart - 1; }
            const bool notBoth = !(dS == 0 && qS == 0);
            const bool rightStart = dS == 0 && (dE != r.dbLen - 1);
            const bool leftStart = qS == 0 && (qE != r.qLen - 1);
            if ((rightStart || leftStart) && notBoth) {
                if (dS == 0) can = (qE == r.qLen - 1) && (r.dbLen - (dE + 1) > 0);
                else if (qS == 0) can = (dE == r.dbLen - 1) && (dS > 0) && ((uint64_t) r.qLen + (uint64_t) dS < maxSeqLen);
            }
        }
        // (the lanes of a wavefront leave the loop together except in its last round, where the active ones are the low lanes)
        const unsigned long long act = __ballot(1);
        const uint32_t prevId = (uint32_t) __shfl_up((int) id, 1, 64);
        const bool head = lane == 0 || prevId != id;
        const unsigned long long heads = __ballot(head), cans = __ballot(can);
        const int segStart = 63 - __clzll((long long) (heads & (lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL))));
        const unsigned long long later = lane == 63 ? 0ULL : (heads & ~((2ULL << lane) - 1ULL));
        const int segEnd = (later ? __ffsll((long long) later) - 1 : (int) __popcll(act)) - 1;            // last lane of my segment
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t) __shfl_up((int) add, d, 64), ta = (uint32_t) __shfl_up((int) addAa, d, 64);
            if (lane - d >= segStart) { add += t; addAa += ta; }
        }
        if (lane == segEnd) {
            const unsigned long long seg = (segEnd == 63 ? ~0ULL : ((2ULL << segEnd) - 1ULL)) & ~((1ULL << segStart) - 1ULL);
            if (add) atomicAdd(&qSum[id], (unsigned long long) add);
            if (qSumAa && addAa) atomicAdd(&qSumAa[id], (unsigned long long) addAa);
            if (cans & seg) atomicOr(&qCan[id], 1u);
        }
    }
}
// arena sizing: query + all targets on either side (a hit is attached at most once, to one side)
// ... and the tier of each query that passed the pre-screen, by queue size (<= 16, <= 32, <= 64, more), as flags packed
// for two 64-bit prefix sums (tierA: tier 0 | tier 1 << 32, tierB: tier 2 | tier 3 << 32); listKernel turns the scanned
// positions into the work lists of the extension kernels (id order, no atomics)
__global__ void arenaSizeKernel(SeqView s, const uint64_t *__restrict__ qoff, const unsigned long long *__restrict__ qSum, const unsigned long long *__restrict__ qSumAa,
                                const uint32_t *__restrict__ qCan, uint32_t *__restrict__ leftCap,
                                uint64_t *__restrict__ bytes, int nuclTiers, uint64_t threadBytes,
                                uint64_t *__restrict__ tierA, uint64_t *__restrict__ tierB,
                                const uint32_t *__restrict__ aaLen, uint32_t *__restrict__ aaLeftCap, uint64_t *__restrict__ aaBytes) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < s.n; id += gridDim.x * blockDim.x) {
        int tier = -1;
        const uint64_t sum = qSum[id], sumAa = aaBytes ? qSumAa[id] : 0;
        const bool can = qCan[id] != 0;
        leftCap[id] = (uint32_t) std::min<uint64_t>(sum, 0xFFFFFFFFull);
        const uint64_t slice = (sum && can) ? (2 * sum + s.len[id] + 40) : 0;
        bytes[id] = slice;      // slack: the re-scoring loops read up to 32 bytes past the query
        if (aaBytes) { aaLeftCap[id] = (uint32_t) std::min<uint64_t>(sumAa, 0xFFFFFFFFull); aaBytes[id] = (sum && can) ? (2 * sumAa + aaLen[id] + 8) : 0; }
        // nucleotide variants: thread-per-query list (up to 256 hits AND an arena slice — query + twice the targets — of at most threadBytes:
        // a thread copies and re-scores byte ranges alone, and its wavefront waits for it; round 4: the contigs of the late iterations of
        // configs[4], 10-100 kb with a few dozen hits, kept their wavefronts for hundreds of ms), wave-per-query list for the rest
        if (sum && can) { const uint64_t h = qoff[id + 1] - qoff[id]; tier = nuclTiers ? ((h <= 256 && slice <= threadBytes) ? 0 : 1) : (h <= 16 ? 0 : (h <= 32 ? 1 : (h <= 64 ? 2 : 3))); }
        tierA[id] = (tier == 0) ? 1ull : ((tier == 1) ? (1ull << 32) : 0ull);
        tierB[id] = (tier == 2) ? 1ull : ((tier == 3) ? (1ull << 32) : 0ull);
    }
}
__global__ void listKernel(uint32_t n, const uint64_t *__restrict__ tierA, const uint64_t *__restrict__ tierB, const uint64_t *__restrict__ posA,
                           const uint64_t *__restrict__ posB, uint32_t *__restrict__ list0, uint32_t *__restrict__ list1, uint32_t *__restrict__ list2,
                           uint32_t *__restrict__ list3) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += gridDim.x * blockDim.x) {
        const uint64_t a = tierA[id], b = tierB[id];
        if (a & 0xFFFFFFFFull) list0[(uint32_t) posA[id]] = id;
        else if (a) list1[(uint32_t) (posA[id] >> 32)] = id;
        else if (b & 0xFFFFFFFFull) list2[(uint32_t) posB[id]] = id;
        else if (b) list3[(uint32_t) (posB[id] >> 32)] = id;
    }
}

// Thread-per-query list of the nucleotide variants in CLASSES of similar work (round 4): the lanes of a wavefront walk their queries in
// lockstep, and the work of a query — re-scoring its deferred hits after every pair of extensions — spans three orders of magnitude
// (a read with 4 hits: 500 residues; a contig with 160 hits: 290 000; profiles/r04_ab_knobs.txt).  The class is the binary order of
// magnitude of the query's arena slice (query + twice its targets); the list is rewritten class by class, heaviest first, in whatever
// order the atomics give inside a class (a query's result does not depend on when it runs).
constexpr int NW_CLASSES = 16;
__device__ __forceinline__ int nuclWorkClass(uint64_t slice) {
    const int k = 63 - __clzll((long long) (slice | 1ull));
    return NW_CLASSES - 1 - (min(max(k, 9), 9 + NW_CLASSES - 1) - 9);
}
__global__ __launch_bounds__(256) void nuclClassCountKernel(const uint32_t *__restrict__ list, uint32_t n, const uint64_t *__restrict__ bytes, uint32_t *__restrict__ counts) {
    __shared__ uint32_t sCnt[NW_CLASSES];
    if (threadIdx.x < NW_CLASSES) sCnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&sCnt[nuclWorkClass(bytes[list[i]])], 1u);
    __syncthreads();
    if (threadIdx.x < NW_CLASSES && sCnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sCnt[threadIdx.x]);
}
// counts[0..15] class sizes, counts[16..31] fill counters (zeroed by the caller)
__global__ __launch_bounds__(256) void nuclClassScatterKernel(const uint32_t *__restrict__ list, uint32_t n, const uint64_t *__restrict__ bytes, uint32_t *__restrict__ counts,
                                                              uint32_t *__restrict__ out) {
    __shared__ uint32_t sCnt[NW_CLASSES], sBase[NW_CLASSES];
    for (uint32_t t0 = blockIdx.x * 256; t0 < n; t0 += gridDim.x * 256) {
        if (threadIdx.x < NW_CLASSES) sCnt[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t i = t0 + threadIdx.x;
        uint32_t id = 0, r = 0; int c = -1;
        if (i < n) { id = list[i]; c = nuclWorkClass(bytes[id]); r = atomicAdd(&sCnt[c], 1u); }
        __syncthreads();
        if (threadIdx.x < NW_CLASSES) {
            uint32_t before = 0;
            for (int k = 0; k < (int) threadIdx.x; k++) before += counts[k];
            sBase[threadIdx.x] = before + (sCnt[threadIdx.x] ? atomicAdd(&counts[NW_CLASSES + threadIdx.x], sCnt[threadIdx.x]) : 0u);
        }
        __syncthreads();
        if (c >= 0) out[sBase[c] + r] = id;
        __syncthreads();
    }
}

__global__ void outLenKernel(SeqView s, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ newLen, int keepTarget,
                             uint64_t *__restrict__ outBytes, uint32_t *__restrict__ keep, uint64_t *__restrict__ appBytes) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < s.n; id += gridDim.x * blockDim.x) {
        const uint32_t f = flags[id];
        uint64_t b = 0, ap = 0; uint32_t k = 0;
        if (f & 0x20u) { b = (uint64_t) newLen[id] + 2; k = 1; ap = b; }
        else if (keepTarget || !(f & 0x80u)) { b = (uint64_t) s.len[id] + 2; k = 1; }
        outBytes[id] = b; keep[id] = k;
        if (appBytes) appBytes[id] = ap;                      // what the entry adds to a shared heap: only rewritten entries have new bytes
    }
}

// The output DB as an INDEX over the heap the input DB lives in (SeqHeap, common.hpp): an entry that is carried over keeps its bytes
// where they are, a rewritten one (flag 0x20: extended, cut, chopped) is copied from the arena to heapBase + appOff[id] behind
// everything the heap held.  G lanes per entry; what moves per iteration is the rewritten 10-20 % of the sequences instead of
// 2 x all residues (writeOutKernel below: 15 ms per iteration at 50 M reads, the same again on every rank of a sharded run).
// Round 5: ONE THREAD per entry for the index (coalesced: rounds 4's eight lanes per entry read every index word eight times over and moved
// 14.6 + 5.4 GB per launch for ~3 GB of rewritten entries), then the wavefront copies the rewritten entries of its 64 ids one after the other
// with all 64 lanes, 8 bytes per lane and step (a rewritten entry is a contig of a few hundred residues: one to three steps).
__global__ __launch_bounds__(256) void appendOutKernel(SeqView s, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ newLen,
                                                       const uint64_t *__restrict__ newStart, const char *arena,
                                                       const uint64_t *__restrict__ appOff, uint64_t heapBase, const uint32_t *__restrict__ keep,
                                                       const uint64_t *__restrict__ keepPos, const uint32_t *__restrict__ inKey,
                                                       char *heap, uint64_t *__restrict__ outOffArr, uint32_t *__restrict__ outLen, uint32_t *__restrict__ outKey,
                                                       unsigned char *__restrict__ changedOut) {
    // (arena and heap are NOT restrict: cyclecheck and findassemblystart pass the heap itself as the arena — disjoint ranges of one buffer)
    const int lane = laneId();
    for (uint32_t id0 = (blockIdx.x * 256 + (threadIdx.x & ~63u)); id0 < s.n; id0 += gridDim.x * 256) {
        const uint32_t id = id0 + (uint32_t) lane;
        bool ext = false; uint32_t L = 0; uint64_t srcOff = 0, o = 0;
        if (id < s.n && keep[id]) {
            ext = (flags[id] & 0x20u) != 0;
            if (ext) { L = newLen[id]; o = heapBase + appOff[id]; srcOff = newStart[id]; }
            else { L = s.len[id]; o = s.off[id]; }
            const uint64_t j = keepPos[id];
            outOffArr[j] = o; outLen[j] = L; outKey[j] = inKey[id];
            if (changedOut) changedOut[j] = ext ? 1 : 0;
        }
        unsigned long long m = __ballot(ext);
        while (m) {
            const int src = __ffsll((long long) m) - 1;
            m &= m - 1;
            const uint32_t cl = (uint32_t) __shfl((int) L, src, 64);
            const uint64_t co = (uint64_t) __shfl((unsigned long long) o, src, 64), cs = (uint64_t) __shfl((unsigned long long) srcOff, src, 64);
            const char *from = arena + cs; char *dst = heap + co;
            for (unsigned p = 8u * (unsigned) lane; p < cl; p += 512u) {
                const uint64_t x = loadU64Unaligned(from + p);                           // buffers are padded past their ends
                if (p + 8 <= cl) storeU64Unaligned(dst + p, x); else storeTail(dst + p, x, cl - p);
            }
            if (lane == 0) { dst[cl] = '\n'; dst[cl + 1] = '\0'; }
        }
    }
}

// origin of every entry of an output DB in the anchor DB of kmermatcher's position cache (plasship_seqdb::d_origin; round 6): a carried-over
// entry keeps the id its source had there (the source's own id when the source IS the anchor), a rewritten entry has none
__global__ __launch_bounds__(256) void originOutKernel(uint32_t n, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ keep, const uint64_t *__restrict__ keepPos,
                                                       const uint32_t *__restrict__ srcOrigin, uint32_t *__restrict__ outOrigin) {
    for (uint32_t id = blockIdx.x * 256 + threadIdx.x; id < n; id += gridDim.x * 256)
        if (keep[id]) outOrigin[keepPos[id]] = (flags[id] & 0x20u) ? 0xFFFFFFFFu : (srcOrigin ? srcOrigin[id] : id);
}

template <int G, int U>
__global__ __launch_bounds__(256) void writeOutKernel(SeqView s, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ newLen,
                                                      const uint64_t *__restrict__ newStart, const char *__restrict__ arena,
                                                      const uint64_t *__restrict__ outOff, const uint32_t *__restrict__ keep,
                                                      const uint64_t *__restrict__ keepPos, const uint32_t *__restrict__ inKey,
                                                      char *__restrict__ outData, uint64_t *__restrict__ outOffArr, uint32_t *__restrict__ outLen, uint32_t *__restrict__ outKey,
                                                      unsigned char *__restrict__ changedOut) {
    // G lanes per sequence, 8 bytes per lane and step: 8 lanes take a read fragment (~46 residues) in one step with most lanes busy,
    // eight sequences per wavefront; contigs take a few steps of contiguous 64-byte pieces.  A sequence is a chain of dependent
    // round trips (what to copy -> the bytes -> the store) and the kernel is bound by the number of chains in flight (round 3: the
    // wavefronts were parked on memory 89 % of their cycles at 2 TB/s), so a group works on U sequences at a time: the U sets of
    // metadata are requested together, then the U first pieces.
    const int gl = threadIdx.x & (G - 1);
    constexpr int groupsPerBlock = 256 / G;
    const uint32_t stride = gridDim.x * groupsPerBlock;
    for (uint32_t id0 = blockIdx.x * groupsPerBlock + (threadIdx.x / G); id0 < s.n; id0 += stride * U) {
        uint32_t L[U]; const char *src[U]; uint64_t o[U]; bool k[U]; uint64_t w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t id = id0 + (uint32_t) u * stride;
            k[u] = id < s.n && keep[id] != 0;
            L[u] = 0; src[u] = s.data; o[u] = 0;
            if (k[u]) {
                const uint32_t f = flags[id]; const uint32_t l0 = s.len[id]; const uint64_t o0 = s.off[id];
                o[u] = outOff[id];
                const bool ext = (f & 0x20u) != 0;
                L[u] = ext ? newLen[id] : l0;
                src[u] = ext ? (arena + newStart[id]) : (s.data + o0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) w[u] = (k[u] && 8u * (unsigned) gl < L[u]) ? loadU64Unaligned(src[u] + 8 * gl) : 0ULL;     // buffers are padded past their ends
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!k[u]) continue;
            const uint32_t id = id0 + (uint32_t) u * stride;
            const unsigned i = 8u * (unsigned) gl;
            char *dst = outData + o[u];
            if (i + 8 <= L[u]) storeU64Unaligned(dst + i, w[u]);
            else if (i < L[u]) storeTail(dst + i, w[u], L[u] - i);
            for (unsigned p = i + 8u * G; p < L[u]; p += 8u * G) {                       // the rest of a contig
                const uint64_t x = loadU64Unaligned(src[u] + p);
                if (p + 8 <= L[u]) storeU64Unaligned(dst + p, x);
                else storeTail(dst + p, x, L[u] - p);
            }
            if (gl == 0) {
                dst[L[u]] = '\n'; dst[L[u] + 1] = '\0';
                const uint64_t j = keepPos[id];
                outOffArr[j] = o[u]; outLen[j] = L[u]; outKey[j] = inKey[id];
                if (changedOut) changedOut[j] = (flags[id] & 0x20u) ? 1 : 0;
            }
        }
    }
}
// ---- sharded run: the extended sequences of the owned queries travel to every rank (all-gather), see mergeExtended below ----
struct __attribute__((aligned(16))) ExtMeta { uint32_t id, len, aaLen, pad; };
__global__ void extMarkKernel(uint32_t n, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ newLen, const uint32_t *__restrict__ aaNewLen,
                              uint32_t *__restrict__ keep, uint64_t *__restrict__ bytes, uint64_t *__restrict__ aaBytes) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += gridDim.x * blockDim.x) {
        const bool ext = (flags[id] & 0x20u) != 0;
        keep[id] = ext ? 1u : 0u; bytes[id] = ext ? newLen[id] : 0u;
        if (aaBytes) aaBytes[id] = ext ? aaNewLen[id] : 0u;
    }
}
__global__ __launch_bounds__(256) void extPackKernel(uint32_t n, const uint32_t *__restrict__ keep, const uint64_t *__restrict__ pos, const uint64_t *__restrict__ off,
                                                     const uint64_t *__restrict__ aaOff, const uint32_t *__restrict__ newLen, const uint64_t *__restrict__ newStart,
                                                     const char *__restrict__ arena, const uint32_t *__restrict__ aaNewLen, const uint64_t *__restrict__ aaNewStart,
                                                     const char *__restrict__ aaArena, ExtMeta *__restrict__ meta, char *__restrict__ packed, char *__restrict__ aaPacked) {
    const int G = 16, groupsPerBlock = 256 / G;
    const int gl = threadIdx.x & (G - 1);
    for (uint32_t id = blockIdx.x * groupsPerBlock + (threadIdx.x / G); id < n; id += gridDim.x * groupsPerBlock) {
        if (!keep[id]) continue;
        const uint32_t L = newLen[id];
        copyBytesG<G>(packed + off[id], arena + newStart[id], L, gl);
        uint32_t La = 0;
        if (aaPacked) { La = aaNewLen[id]; copyBytesG<G>(aaPacked + aaOff[id], aaArena + aaNewStart[id], La, gl); }
        if (gl == 0) { ExtMeta m; m.id = id; m.len = L; m.aaLen = La; m.pad = 0; meta[pos[id]] = m; }
    }
}
__global__ void extLensKernel(const ExtMeta *__restrict__ meta, uint64_t m, uint64_t *__restrict__ lens, uint64_t *__restrict__ aaLens) {
    for (uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t) gridDim.x * blockDim.x) { lens[j] = meta[j].len; if (aaLens) aaLens[j] = meta[j].aaLen; }
}
__global__ void extMergeKernel(const ExtMeta *__restrict__ meta, uint64_t m, const uint64_t *__restrict__ start, const uint64_t *__restrict__ aaStart,
                               uint32_t *__restrict__ flags, uint32_t *__restrict__ newLen, uint64_t *__restrict__ newStart,
                               uint32_t *__restrict__ aaNewLen, uint64_t *__restrict__ aaNewStart) {
    for (uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t) gridDim.x * blockDim.x) {
        const ExtMeta e = meta[j];
        atomicOr(&flags[e.id], 0x20u); newLen[e.id] = e.len; newStart[e.id] = start[j];
        if (aaStart) { aaNewLen[e.id] = e.aaLen; aaNewStart[e.id] = aaStart[j]; }
    }
}
__global__ void orFlagsKernel(const uint32_t *__restrict__ all, uint32_t n, int world, uint32_t *__restrict__ flags) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += gridDim.x * blockDim.x) {
        uint32_t f = flags[id];
        for (int r = 0; r < world; r++) f |= all[(size_t) r * n + id] & 0x80u;       // "consumed as a target" by any rank's query
        flags[id] = f;
    }
}

__global__ void maxU32Kernel(const uint32_t *__restrict__ v, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) m = max(m, v[i]);
    m = (uint32_t) waveReduceMax((int) m);
    if (laneId() == 0) atomicMax(out, m);
}

}  // namespace plasship
using namespace plasship;

// host side of the comparator table: evaluate new tuples with the platform libm, rebuild the open-addressing table
static int ambTableInsert(plasship_ctx *ctx, const uint32_t *tuples, uint32_t n) {
    auto hashOf = [](const uint32_t *k) { return (k[0] * 0x9E3779B1u) ^ (k[1] * 0x85EBCA77u) ^ (k[2] * 0xC2B2AE3Du) ^ (k[3] * 0x27D4EB2Fu); };
    auto insertInto = [&](std::vector<uint32_t> &keys, std::vector<uint8_t> &vals, uint32_t slots, const uint32_t *k, uint8_t v) -> bool {
        uint32_t s = hashOf(k) & (slots - 1);
        while (keys[4 * (size_t) s] != 0) {
            if (memcmp(&keys[4 * (size_t) s], k, 16) == 0) return false;
            s = (s + 1) & (slots - 1);
        }
        memcpy(&keys[4 * (size_t) s], k, 16); vals[s] = v;
        return true;
    };
    size_t have = 0;
    for (size_t i = 0; i < ctx->ambVals.size(); i++) have += ctx->ambKeys[4 * i] != 0;
    uint32_t slots = ctx->ambSlots ? ctx->ambSlots : 1024;
    while ((have + n) * 2 > slots) slots *= 2;
    if (slots != ctx->ambSlots) {                             // grow: re-insert what is there
        std::vector<uint32_t> nk((size_t) slots * 4, 0); std::vector<uint8_t> nv(slots, 0);
        for (size_t i = 0; i < ctx->ambVals.size(); i++)
            if (ctx->ambKeys[4 * i] != 0) insertInto(nk, nv, slots, &ctx->ambKeys[4 * i], ctx->ambVals[i]);
        ctx->ambKeys.swap(nk); ctx->ambVals.swap(nv); ctx->ambSlots = slots;
        ctx->d_ambKeys.release(); ctx->d_ambVals.release();
        if (ctx->d_ambKeys.alloc((size_t) slots * 16) != hipSuccess || ctx->d_ambVals.alloc(slots) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t *k = tuples + 4 * (size_t) i;
        if (k[0] == 0) continue;
        insertInto(ctx->ambKeys, ctx->ambVals, slots, k, (uint8_t) nuclPosteriorClass(k[0], k[1], k[2], k[3]));
    }
    PH_COPY_SYNC(ctx->stream, ctx->d_ambKeys.p, ctx->ambKeys.data(), (size_t) slots * 16, hipMemcpyHostToDevice);
    PH_COPY_SYNC(ctx->stream, ctx->d_ambVals.p, ctx->ambVals.data(), slots, hipMemcpyHostToDevice);
    return PLASSHIP_OK;
}

// builds one output DB (extended entries from the arena + carried-over entries of `db`), entries in key order.
// mode 0: the DB shares `db`'s heap when it has one with room (appendOutKernel: only the rewritten entries move); else its entries are
//         written back to back into a NEW heap with room for the iterations to come (as much again as the data, at most PLASSHIP_TUNE_DBHEAP_GB = 16 GB;
//         PLASSHIP_TUNE_DBHEAP=2: no heaps, every DB in an exact buffer of its own as in rounds 1-3)
// mode 1: a packed copy in an exact buffer (packedCopyOf)
static int buildOutputDBImpl(plasship_ctx *ctx, const plasship_seqdb *db, const uint32_t *dFlags, const uint32_t *dNewLen, const uint64_t *dNewStart,
                             const char *dArena, int keepTarget, void *dTmp, size_t tmpBytes, plasship_seqdb **out,
                             const void *dExtra, void *hExtra, size_t extraBytes, hipEvent_t doneEvent, int mode, bool noAppend = false) {
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const SeqView sv = db->view();
    const bool useHeaps = mode == 0 && tuneInt("DBHEAP", 1) == 1;
    // noAppend (the protein-guided path, round 5): the DB is written back to back in key order — the layout of the DB file.  The
    // reference's proteinaln2nucl and guidedassembleresults read PAST the end of an entry when a protein twin and its ORF / 3 differ in
    // length, i.e. into the entry that FOLLOWS in the data file; behind an entry appended to a shared heap lies
    // whatever was appended next.  tests/test_gpu_deep.py::test_four_guided_iterations (round 5) found the difference in the third
    // guided iteration, the first one whose input DB had appended entries: 3 bytes of one alignment DB in 382 MB.
    const bool mayAppend = useHeaps && db->heap != nullptr && !noAppend;
    DevBuf dOutBytes, dKeep, dOutOff, dKeepPos, dMaxLen, dAppBytes, dAppOff;
    if (dOutBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dKeep.alloc(((size_t) N + 1) * 4) != hipSuccess || dOutOff.alloc(((size_t) N + 2) * 8) != hipSuccess ||
        dKeepPos.alloc(((size_t) N + 2) * 8) != hipSuccess || dMaxLen.alloc(4) != hipSuccess ||
        (mayAppend && (dAppBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dAppOff.alloc(((size_t) N + 2) * 8) != hipSuccess))) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (N) hipLaunchKernelGGL(outLenKernel, dim3(std::min<uint32_t>((N + 255) / 256, 8192)), dim3(256), 0, st, sv, dFlags, dNewLen, keepTarget, dOutBytes.as<uint64_t>(), dKeep.as<uint32_t>(),
                              mayAppend ? dAppBytes.as<uint64_t>() : (uint64_t *) nullptr);
    if (exclusiveScanU64(st, dOutBytes.as<uint64_t>(), dOutOff.as<uint64_t>(), N, dTmp, tmpBytes) ||
        exclusiveScanU32(st, dKeep.as<uint32_t>(), dKeepPos.as<uint64_t>(), N, dTmp, tmpBytes) ||
        (mayAppend && exclusiveScanU64(st, dAppBytes.as<uint64_t>(), dAppOff.as<uint64_t>(), N, dTmp, tmpBytes))) { setError("plasship_assemble: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t outBytes = 0, outN = 0, appTotal = 0;
    PH_CHECK(hipMemcpyAsync(&outBytes, dOutOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&outN, dKeepPos.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    if (mayAppend) PH_CHECK(hipMemcpyAsync(&appTotal, dAppOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    std::unique_ptr<plasship_seqdb> holder(new plasship_seqdb());   // released to the caller on success only
    plasship_seqdb *o = holder.get();
    o->dbtype = db->dbtype; o->n = (size_t) outN; o->dataBytes = outBytes; o->residues = outBytes - 2 * outN; o->hostIndexValid = false;
    if (o->d_off.allocLong((outN + 1) * 8) != hipSuccess || o->d_len.allocLong((outN + 1) * 4) != hipSuccess || o->d_key.allocLong((outN + 1) * 4) != hipSuccess) {
        setError("plasship_assemble: out of device memory for the output DB's index"); return PLASSHIP_ERR_DEVICE;
    }
    // lineage: the extended / cut entries marked (one byte per NEW id); same ids as `db` if nothing was dropped (kmermatcher's
    // selected-window cache needs that), else only "an unmarked entry is an entry of `db`" (cyclecheck's known-linear entries)
    if (outN && mode == 0) {
        if (o->d_changed.allocLong((size_t) outN) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        o->ancestorGen = db->gen;
        if (outN == N) o->parentGen = db->gen;
    }
    // ... and, when `db` is the anchor of kmermatcher's position cache or descends from it, the id every carried-over entry had in the anchor
    if (outN && mode == 0 && ctx->kmPosCache.valid && ctx->kmPosCache.gen != 0) {
        const uint64_t anchor = ctx->kmPosCache.gen;
        const bool isAnchor = db->gen == anchor, descends = !isAnchor && db->originGen == anchor && db->d_origin.p != nullptr;
        if (isAnchor || descends) {
            if (o->d_origin.allocLong((size_t) outN * 4) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            o->originGen = anchor;
            hipLaunchKernelGGL(originOutKernel, dim3(std::min<uint32_t>((N + 255) / 256, 4096)), dim3(256), 0, st, N, dFlags, (const uint32_t *) dKeep.as<uint32_t>(), (const uint64_t *) dKeepPos.as<uint64_t>(),
                               descends ? (const uint32_t *) db->d_origin.as<uint32_t>() : (const uint32_t *) nullptr, o->d_origin.as<uint32_t>());
        }
    }
    // room in the shared heap?  (the reservation is atomic: two DBs derived from one parent get disjoint ranges)
    bool append = false; uint64_t heapBase = 0;
    if (mayAppend) {                                          // (the reservation commits only if it fits: a DB that does not fit leaves the heap's room to its siblings)
        uint64_t cur = db->heap->used.load();
        while (cur + appTotal + 64 <= db->heap->buf.bytes) {
            if (db->heap->used.compare_exchange_weak(cur, cur + appTotal)) { heapBase = cur; append = true; break; }
        }
    }
    if (append) {
        o->heap = db->heap; o->contiguous = false;
        if (N) hipLaunchKernelGGL(appendOutKernel, dim3(std::min<uint32_t>((N + 255) / 256, (uint32_t) ctx->numCU * (uint32_t) tuneInt("WRITEOUT", 16))), dim3(256), 0, st, sv, dFlags, dNewLen, dNewStart, dArena,
                                  (const uint64_t *) dAppOff.as<uint64_t>(), heapBase, (const uint32_t *) dKeep.as<uint32_t>(), (const uint64_t *) dKeepPos.as<uint64_t>(), (const uint32_t *) db->d_key.as<uint32_t>(),
                                  db->heap->buf.as<char>(), o->d_off.as<uint64_t>(), o->d_len.as<uint32_t>(), o->d_key.as<uint32_t>(), o->d_changed.as<unsigned char>());
    } else {
        char *dst = nullptr;
        if (useHeaps && !noAppend) {                           // (noAppend: nobody will ever append behind this DB — an exact buffer, no slack: ADVICE r5)
            o->heap = std::make_shared<SeqHeap>();
            // room for the entries the next iterations rewrite (25-35 % of the data per iteration at 50 M reads): as much again as the
            // data, but no more than PLASSHIP_TUNE_DBHEAP_GB (default 16; round 4: 8 — the sparse alignment lists of round 5 freed 14 GB) — kmermatcher's record arrays need 170 of the 288 GB there
            const uint64_t cap = outBytes + std::min<uint64_t>(outBytes, (uint64_t) tuneInt("DBHEAP_GB", 16) << 30) + 4096;
            if (o->heap->buf.allocLong(cap) == hipSuccess) { o->heap->used = outBytes; dst = o->heap->buf.as<char>(); }
            else o->heap.reset();                             // no room for the slack: an exact buffer will do
        }
        if (!dst) {
            if (o->d_data.allocLong(outBytes + 64) != hipSuccess) {
                size_t fr = 0, tt = 0; (void) hipMemGetInfo(&fr, &tt);
                setError("plasship_assemble: out of device memory for the output DB (" + std::to_string(outBytes) + " bytes, " + std::to_string(outN) + " sequences; device free " + std::to_string(fr) + " of " + std::to_string(tt) + ")"); return PLASSHIP_ERR_DEVICE;
            }
            dst = o->d_data.as<char>();
        }
        o->contiguous = true;
        PH_CHECK(hipMemsetAsync(dst + outBytes, 0, 64, st));
        if (N) {
            // (2 and 4 sequences in flight per lane group changed nothing — profiles/r03_ab_knobs.txt: the kernel is not bound by its chains of round trips)
            const unsigned woGrid = std::min<uint32_t>((N + 15) / 16, (uint32_t) ctx->numCU * (uint32_t) tuneInt("WRITEOUT", 16));
            hipLaunchKernelGGL((writeOutKernel<8, 1>), dim3(woGrid), dim3(256), 0, st, sv, dFlags, dNewLen,
                               dNewStart, dArena, dOutOff.as<uint64_t>(), dKeep.as<uint32_t>(), dKeepPos.as<uint64_t>(), db->d_key.as<uint32_t>(),
                               dst, o->d_off.as<uint64_t>(), o->d_len.as<uint32_t>(), o->d_key.as<uint32_t>(), o->d_changed.as<unsigned char>());
        }
    }
    PH_CHECK(hipMemcpyAsync(o->d_off.as<uint64_t>() + outN, &outBytes, 8, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemsetAsync(dMaxLen.p, 0, 4, st));
    if (outN) hipLaunchKernelGGL(maxU32Kernel, dim3(std::min<uint64_t>((outN + 255) / 256, 1024)), dim3(256), 0, st, o->d_len.as<uint32_t>(), outN, dMaxLen.as<uint32_t>());
    uint32_t maxLen = 0;
    if (doneEvent) PH_CHECK(hipEventRecord(doneEvent, st));
    PH_CHECK(hipMemcpyAsync(&maxLen, dMaxLen.p, 4, hipMemcpyDeviceToHost, st));
    if (dExtra) PH_CHECK(hipMemcpyAsync(hExtra, dExtra, extraBytes, hipMemcpyDeviceToHost, st));      // the caller's counters ride along
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    o->maxEntryLen = maxLen + 2;
    o->buildAppendedBytes = append ? appTotal : 0; o->buildCopiedBytes = append ? 0 : outBytes;
    *out = holder.release();
    return PLASSHIP_OK;
}
int plasship::buildOutputDB(plasship_ctx *ctx, const plasship_seqdb *db, const uint32_t *dFlags, const uint32_t *dNewLen, const uint64_t *dNewStart,
                            const char *dArena, int keepTarget, void *dTmp, size_t tmpBytes, plasship_seqdb **out,
                            const void *dExtra, void *hExtra, size_t extraBytes, hipEvent_t doneEvent, bool noAppend) {
    return buildOutputDBImpl(ctx, db, dFlags, dNewLen, dNewStart, dArena, keepTarget, dTmp, tmpBytes, out, dExtra, hExtra, extraBytes, doneEvent, 0, noAppend);
}
int plasship::packedCopyOf(plasship_ctx *ctx, const plasship_seqdb *db, std::unique_ptr<plasship_seqdb> &out) {
    const uint32_t N = (uint32_t) db->n;
    DevBuf dFlags, dTmp; const size_t tmpBytes = exclusiveScanTmpBytes((size_t) N + 2);
    if (dFlags.alloc(((size_t) N + 1) * 4) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship: out of device memory while packing a sequence DB"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dFlags.p, 0, ((size_t) N + 1) * 4, ctx->stream));
    plasship_seqdb *o = nullptr;
    const int rc = buildOutputDBImpl(ctx, db, dFlags.as<uint32_t>(), nullptr, nullptr, nullptr, 1, dTmp.p, tmpBytes, &o, nullptr, nullptr, 0, nullptr, 1);
    if (rc) return rc;
    o->maxEntryLen = db->maxEntryLen;
    out.reset(o);
    return PLASSHIP_OK;
}

// Sharded run (plasship_ctx_set_comm): every rank has extended the queries it owns.  The extended sequences (and, for the
// guided variant, their protein twins) are packed, all-gathered and entered into flags / newLen / newStart of every rank as if
// it had produced them itself, so that buildOutputDB writes the complete DB everywhere; the "consumed" bits only matter with
// --keep-target 0 and are OR-ed over the ranks then.  gathered / gatheredAa replace the local arenas.
static int mergeExtended(plasship_ctx *ctx, uint32_t N, bool guided, int keepTarget, uint32_t *dFlags, uint32_t *dNewLen, uint64_t *dNewStart, const char *dArena,
                         uint32_t *dAaNewLen, uint64_t *dAaNewStart, const char *dAaArena, void *dTmp, size_t tmpBytes, DevBuf &gathered, DevBuf &gatheredAa) {
    hipStream_t st = ctx->stream;
    const plasship_comm *cm = commOf(ctx);
    const int W = cm->world;
    DevBuf dKeep, dBytes, dAaBytes, dPos, dOff, dAaOff;
    if (dKeep.alloc(((size_t) N + 1) * 4) != hipSuccess || dBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dPos.alloc(((size_t) N + 2) * 8) != hipSuccess ||
        dOff.alloc(((size_t) N + 2) * 8) != hipSuccess || (guided && (dAaBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dAaOff.alloc(((size_t) N + 2) * 8) != hipSuccess))) {
        setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    const unsigned gridN = std::min<uint32_t>((N + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(extMarkKernel, dim3(gridN), dim3(256), 0, st, N, dFlags, dNewLen, guided ? dAaNewLen : (const uint32_t *) nullptr, dKeep.as<uint32_t>(), dBytes.as<uint64_t>(),
                       guided ? dAaBytes.as<uint64_t>() : (uint64_t *) nullptr);
    if (exclusiveScanU32(st, dKeep.as<uint32_t>(), dPos.as<uint64_t>(), N, dTmp, tmpBytes) || exclusiveScanU64(st, dBytes.as<uint64_t>(), dOff.as<uint64_t>(), N, dTmp, tmpBytes) ||
        (guided && exclusiveScanU64(st, dAaBytes.as<uint64_t>(), dAaOff.as<uint64_t>(), N, dTmp, tmpBytes))) { setError("plasship_assemble: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t nExt = 0, extBytes = 0, aaExtBytes = 0;
    PH_CHECK(hipMemcpyAsync(&nExt, dPos.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&extBytes, dOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    if (guided) PH_CHECK(hipMemcpyAsync(&aaExtBytes, dAaOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    DevBuf dMeta, dPacked, dAaPacked;
    if (dMeta.alloc(std::max<uint64_t>(nExt, 1) * sizeof(ExtMeta)) != hipSuccess || dPacked.alloc(extBytes + 64) != hipSuccess || (guided && dAaPacked.alloc(aaExtBytes + 64) != hipSuccess)) {
        setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    if (N) hipLaunchKernelGGL(extPackKernel, dim3(std::min<uint32_t>((N + 15) / 16, (uint32_t) ctx->numCU * 16)), dim3(256), 0, st, N, dKeep.as<uint32_t>(), dPos.as<uint64_t>(), dOff.as<uint64_t>(),
                              guided ? dAaOff.as<uint64_t>() : (const uint64_t *) nullptr, dNewLen, dNewStart, dArena, dAaNewLen, dAaNewStart, dAaArena,
                              dMeta.as<ExtMeta>(), dPacked.as<char>(), guided ? dAaPacked.as<char>() : (char *) nullptr);
    PH_TRACE(st, "assemble: packed the extended sequences");
    DevBuf gMeta; std::vector<uint64_t> rb(W), rbMeta(W), rbSeq(W), rbAa(W);
    uint64_t M = 0;
    {   // sizes of all three payloads in one host all-gather
        const uint64_t mine[3] = {nExt, extBytes, aaExtBytes}; std::vector<uint64_t> all(3 * (size_t) W);
        const int rc0 = commAllgatherHost(ctx, mine, all.data(), 24); if (rc0) return rc0;
        for (int r = 0; r < W; r++) { M += all[3 * (size_t) r]; rbMeta[r] = all[3 * (size_t) r] * sizeof(ExtMeta); rbSeq[r] = all[3 * (size_t) r + 1]; rbAa[r] = all[3 * (size_t) r + 2]; }
    }
    int rc = commAllgathervBytesKnown(ctx, dMeta.p, nExt * sizeof(ExtMeta), gMeta, rbMeta); if (rc) return rc;
    rc = commAllgathervBytesKnown(ctx, dPacked.p, extBytes, gathered, rbSeq); if (rc) return rc;
    if (guided) { rc = commAllgathervBytesKnown(ctx, dAaPacked.p, aaExtBytes, gatheredAa, rbAa); if (rc) return rc; }
    PH_TRACE(st, "assemble: gathered the extended sequences");
    // the gathered byte blocks are the ranks' packed blocks in rank order = the gathered meta order: starts are a prefix sum
    DevBuf dLens, dStart, dAaLens, dAaStart, dTmp2; const size_t tmp2Bytes = exclusiveScanTmpBytes(M + 2);
    if (dLens.alloc((M + 1) * 8) != hipSuccess || dStart.alloc((M + 2) * 8) != hipSuccess || dTmp2.alloc(tmp2Bytes) != hipSuccess ||
        (guided && (dAaLens.alloc((M + 1) * 8) != hipSuccess || dAaStart.alloc((M + 2) * 8) != hipSuccess))) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (M) {
        const unsigned gridM = (unsigned) std::min<uint64_t>((M + 255) / 256, 8192);
        hipLaunchKernelGGL(extLensKernel, dim3(gridM), dim3(256), 0, st, gMeta.as<ExtMeta>(), M, dLens.as<uint64_t>(), guided ? dAaLens.as<uint64_t>() : (uint64_t *) nullptr);
        if (exclusiveScanU64(st, dLens.as<uint64_t>(), dStart.as<uint64_t>(), M, dTmp2.p, tmp2Bytes) ||
            (guided && exclusiveScanU64(st, dAaLens.as<uint64_t>(), dAaStart.as<uint64_t>(), M, dTmp2.p, tmp2Bytes))) { setError("plasship_assemble: scan failed"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(extMergeKernel, dim3(gridM), dim3(256), 0, st, gMeta.as<ExtMeta>(), M, dStart.as<uint64_t>(), guided ? dAaStart.as<uint64_t>() : (const uint64_t *) nullptr,
                           dFlags, dNewLen, dNewStart, dAaNewLen, dAaNewStart);
    }
    PH_TRACE(st, "assemble: merged the extended sequences");
    if (!keepTarget) {
        DevBuf gFlags;
        rc = commAllgathervBytes(ctx, dFlags, (uint64_t) N * 4, gFlags, rb); if (rc) return rc;
        if (N) hipLaunchKernelGGL(orFlagsKernel, dim3(gridN), dim3(256), 0, st, gFlags.as<uint32_t>(), N, W, dFlags);
        PH_CHECK(plasship::streamSync(st));       // gFlags is released on return
    }
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    return PLASSHIP_OK;
}

// aaDb == nullptr: assembleresults (protein DB) / nuclassembleresults (nucleotide DB); aaDb != nullptr: guidedassembleresults
static int assembleImpl(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_seqdb *aaDb, const plasship_alns *al,
                        const plasship_assemble_params *par, plasship_seqdb **out, plasship_seqdb **outAa, plasship_assemble_stats *stats) {
    const bool guided = aaDb != nullptr;
    // protein DB -> assembleresults; nucleotide DB -> nuclassembleresults (what the nuclassemble workflow runs on reads)
    const bool nucl = db->dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES;
    if (!nucl && db->dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS) { setError("plasship_assemble: the sequence DB is neither amino acids nor nucleotides"); return PLASSHIP_ERR_ARG; }
    if (par->rescore_mode != 3) { setError("plasship_assemble: only --rescore-mode 3"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (al->nQueries != db->n) { setError("plasship_assemble: alignment list does not belong to the DB"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    // nuclassembleresults / guidedassembleresults rank a query's alignment with itself among the others (it enters the heap like any hit);
    // assembleresults pops and discards it (assembleresult.cpp:203-209): only the former need the identity pairs scored (common.hpp: selfPending)
    if (nucl) { const int rcS = finishSelfAlns(ctx, al); if (rcS) return rcS; }
    const uint64_t nLines = al->nSlots;                     // record slots (a sparse list: holes included, common.hpp); al->nLines of them are alignments
    DevBuf dLeftCap, dBytes, dArenaOff, dTmp, dItems, dFlags, dNewLen, dNewStart, dMat, dStats, dArena;
    const size_t tmpBytes = exclusiveScanTmpBytes((size_t) N + 2);
    if (dLeftCap.alloc(((size_t) N + 1) * 4) != hipSuccess || dBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dArenaOff.alloc(((size_t) N + 2) * 8) != hipSuccess ||
        dTmp.alloc(tmpBytes) != hipSuccess || dItems.alloc(std::max<uint64_t>(nLines, 1) * sizeof(Item)) != hipSuccess || dFlags.alloc(((size_t) N + 1) * 4) != hipSuccess ||
        dNewLen.alloc(((size_t) N + 1) * 4) != hipSuccess || dNewStart.alloc(((size_t) N + 1) * 8) != hipSuccess || dMat.alloc(123 * 123) != hipSuccess || dStats.alloc(128) != hipSuccess) {
        setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipMemsetAsync(dFlags.p, 0, ((size_t) N + 1) * 4, st));
    PH_CHECK(hipMemsetAsync(dNewLen.p, 0, ((size_t) N + 1) * 4, st));
    PH_CHECK(hipMemsetAsync(dStats.p, 0, 128, st));
    PH_CHECK(hipMemcpyAsync(dMat.p, asciiSubMat(nucl), 123 * 123, hipMemcpyHostToDevice, st));
    { const int rcOL = ensureOffLen(ctx, db); if (rcOL) return rcOL; }          // the extension kernels look up random targets
    const SeqView sv = db->view();
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    // work lists by queue size: [0] <= 16 alignments, [1] <= 32, [2] <= 64, [3] more (filled by arenaSizeKernel)
    DevBuf dBigList, dMidList, dMid32List, dSmallList, dTierA, dTierB, dPosA, dPosB, dAaLeftCap, dAaBytes, dAaArenaOff, dAaArena, dAaNewLen, dAaNewStart;
    if (guided && (dAaLeftCap.alloc(((size_t) N + 1) * 4) != hipSuccess || dAaBytes.alloc(((size_t) N + 1) * 8) != hipSuccess || dAaArenaOff.alloc(((size_t) N + 2) * 8) != hipSuccess ||
                   dAaNewLen.alloc(((size_t) N + 1) * 4) != hipSuccess || dAaNewStart.alloc(((size_t) N + 1) * 8) != hipSuccess)) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (guided) PH_CHECK(hipMemsetAsync(dAaNewLen.p, 0, ((size_t) N + 1) * 4, st));
    if (dBigList.alloc(((size_t) N + 1) * 4) != hipSuccess || dMidList.alloc(((size_t) N + 1) * 4) != hipSuccess || dMid32List.alloc(((size_t) N + 1) * 4) != hipSuccess ||
        dSmallList.alloc(((size_t) N + 1) * 4) != hipSuccess || dTierA.alloc(((size_t) N + 1) * 8) != hipSuccess || dTierB.alloc(((size_t) N + 1) * 8) != hipSuccess ||
        dPosA.alloc(((size_t) N + 2) * 8) != hipSuccess || dPosB.alloc(((size_t) N + 2) * 8) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    DevBuf dQSum, dQSumAa, dQCan;
    if (dQSum.alloc(((size_t) N + 1) * 8) != hipSuccess || dQCan.alloc(((size_t) N + 1) * 4) != hipSuccess || (guided && dQSumAa.alloc(((size_t) N + 1) * 8) != hipSuccess)) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dQSum.p, 0, ((size_t) N + 1) * 8, st));
    PH_CHECK(hipMemsetAsync(dQCan.p, 0, ((size_t) N + 1) * 4, st));
    if (guided) PH_CHECK(hipMemsetAsync(dQSumAa.p, 0, ((size_t) N + 1) * 8, st));
    if (nLines) hipLaunchKernelGGL(arenaSumKernel, dim3((unsigned) std::min<uint64_t>((nLines + 255) / 256, (uint64_t) ctx->numCU * 32)), dim3(256), 0, st, al->d_recs.as<AlnRec>(), (uint64_t) nLines,
                                       (uint64_t) par->max_seq_len, dQSum.as<unsigned long long>(), guided ? dQSumAa.as<unsigned long long>() : (unsigned long long *) nullptr, dQCan.as<uint32_t>(),
                                       (nucl && !guided) ? 1 : 0);
    if (N) hipLaunchKernelGGL(arenaSizeKernel, dim3(std::min<uint32_t>((N + 255) / 256, 8192)), dim3(256), 0, st, sv, al->d_qoff.as<uint64_t>(), (const unsigned long long *) dQSum.as<unsigned long long>(),
                              (const unsigned long long *) dQSumAa.as<unsigned long long>(), (const uint32_t *) dQCan.as<uint32_t>(), dLeftCap.as<uint32_t>(), dBytes.as<uint64_t>(), nucl ? 1 : 0, (uint64_t) tuneInt("NUCL_THREAD_BYTES", 1 << 30),
                              dTierA.as<uint64_t>(), dTierB.as<uint64_t>(),
                              guided ? aaDb->d_len.as<uint32_t>() : (const uint32_t *) nullptr, dAaLeftCap.as<uint32_t>(), guided ? dAaBytes.as<uint64_t>() : (uint64_t *) nullptr);
    if (guided && exclusiveScanU64(st, dAaBytes.as<uint64_t>(), dAaArenaOff.as<uint64_t>(), N, dTmp.p, tmpBytes)) { setError("plasship_assemble: scan failed"); return PLASSHIP_ERR_DEVICE; }
    if (exclusiveScanU64(st, dTierA.as<uint64_t>(), dPosA.as<uint64_t>(), N, dTmp.p, tmpBytes) || exclusiveScanU64(st, dTierB.as<uint64_t>(), dPosB.as<uint64_t>(), N, dTmp.p, tmpBytes)) {
        setError("plasship_assemble: scan failed"); return PLASSHIP_ERR_DEVICE;
    }
    if (N) hipLaunchKernelGGL(listKernel, dim3(std::min<uint32_t>((N + 255) / 256, 8192)), dim3(256), 0, st, N, dTierA.as<uint64_t>(), dTierB.as<uint64_t>(), dPosA.as<uint64_t>(), dPosB.as<uint64_t>(),
                              dSmallList.as<uint32_t>(), dMid32List.as<uint32_t>(), dMidList.as<uint32_t>(), dBigList.as<uint32_t>());
    if (exclusiveScanU64(st, dBytes.as<uint64_t>(), dArenaOff.as<uint64_t>(), N, dTmp.p, tmpBytes)) { setError("plasship_assemble: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t arenaBytes = 0, aaArenaBytes = 0;
    uint64_t totA = 0, totB = 0;
    if (guided) PH_CHECK(hipMemcpyAsync(&aaArenaBytes, dAaArenaOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&arenaBytes, dArenaOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&totA, dPosA.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&totB, dPosB.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    const uint32_t cnts[4] = {(uint32_t) totA, (uint32_t) (totA >> 32), (uint32_t) totB, (uint32_t) (totB >> 32)};
    if (dArena.alloc(arenaBytes + 64) != hipSuccess || (guided && dAaArena.alloc(aaArenaBytes + 64) != hipSuccess)) {
        size_t fr = 0, tt = 0; (void) hipMemGetInfo(&fr, &tt);
        setError("plasship_assemble: out of device memory for the extension arena (" + std::to_string(arenaBytes) + " bytes; device free " + std::to_string(fr) + " of " + std::to_string(tt) + ")"); return PLASSHIP_ERR_DEVICE;
    }
    HostEvaluer ev(nucl, db->residues);
    AsmArgs a; memset(&a, 0, sizeof(a));
    a.s = sv; a.qoff = al->d_qoff.as<uint64_t>(); a.recs = al->d_recs.as<AlnRec>(); a.items = dItems.as<Item>(); a.arenaOff = dArenaOff.as<uint64_t>();
    a.leftCap = dLeftCap.as<uint32_t>(); a.arena = dArena.as<char>(); a.flags = dFlags.as<uint32_t>(); a.newLen = dNewLen.as<uint32_t>(); a.newStart = dNewStart.as<uint64_t>();
    a.mat = dMat.as<signed char>(); a.lambda = ev.g[0]; a.logK = ev.logK; a.ln2 = ev.ln2; a.seqIdThr = par->seq_id_thr; a.maxSeqLen = par->max_seq_len; a.rescoreMode = par->rescore_mode;
    a.ownLaneMin = (uint32_t) tuneInt("ASM_OWN", 6);      // swept 1 / 3 / 6 / 12 / never: 30.4 / 28.4 / 27.9 / 29.9 / 40.3 ms for the two wide tiers (profiles/r05_ab_knobs.txt, call 11)
    a.stats = dStats.as<unsigned long long>();
    a.smallList = dSmallList.as<uint32_t>(); a.nSmall = cnts[0];
    a.mid32List = dMid32List.as<uint32_t>(); a.nMid32 = cnts[1];
    a.midList = dMidList.as<uint32_t>(); a.nMid = cnts[2];
    a.bigList = dBigList.as<uint32_t>(); a.nBig = cnts[3];
    // assembleBigKernel keeps the self hit in its HBM-resident queue (its rank decides what is left queued when the length cap ends the pop
    // loop, assembleresult.cpp:259-263,285): the identity pairs of ITS queries are scored now (common.hpp: selfPending); the register-queue
    // kernels never queue the self hit
    if (!nucl && a.nBig) { const int rcS = finishSelfAlns(ctx, al, dBigList.as<uint32_t>(), a.nBig); if (rcS) return rcS; }
    std::unique_ptr<plasship_seqdb> aaPacked;                // guided: the twins are read past their end like the reference does (buildOutputDBImpl, noAppend):
    if (guided && !aaDb->contiguous) {                       // a twin DB that lives in a shared heap is laid out like its DB file first
        const int rcP = packedCopyOf(ctx, aaDb, aaPacked); if (rcP) return rcP;
    }
    if (guided) {
        a.aa = (aaPacked ? aaPacked.get() : aaDb)->view(); a.aaArena = dAaArena.as<char>(); a.aaArenaOff = dAaArenaOff.as<uint64_t>(); a.aaLeftCap = dAaLeftCap.as<uint32_t>();
        a.aaNewLen = dAaNewLen.as<uint32_t>(); a.aaNewStart = dAaNewStart.as<uint64_t>();
    }
    DevBuf dHeap, dNeed, dRedo[2], dCnt;
    if (nucl) {
        const uint32_t needCap = 1u << 16;
        if (dHeap.alloc(std::max<uint64_t>(nLines, 1) * 12) != hipSuccess || dNeed.alloc((size_t) needCap * 16) != hipSuccess || dRedo[0].alloc(((size_t) N + 1) * 4) != hipSuccess ||
            dRedo[1].alloc(((size_t) N + 1) * 4) != hipSuccess || dCnt.alloc(8) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        a.heap = dHeap.as<uint32_t>();
        if (!ctx->d_cmpCache.p) {
            const size_t cb = (size_t) CMP_LEN * CMP_MM * CMP_LEN * CMP_MM;
            if (ctx->d_cmpCache.alloc(cb) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            PH_CHECK(hipMemsetAsync(ctx->d_cmpCache.p, 0, cb, st));
        }
        a.cmpCache = ctx->d_cmpCache.as<uint8_t>();
        a.needKeys = dNeed.as<uint32_t>(); a.needCount = dCnt.as<uint32_t>(); a.needCap = needCap; a.redoCount = dCnt.as<uint32_t>() + 1;
        PH_CHECK(hipEventRecord(ctx->ev[2], st));
        // Pass 0 runs every query; a query that needs a comparator decision the table lacks leaves no trace and is
        // run again once the host has evaluated the tuple (the set of such tuples is small and recurs, so the table
        // kept in the context makes later calls single-pass).
        // pass 0: queries with up to 256 hits one thread each, the rest one wavefront each; later passes (queries that met
        // a comparator tuple the table lacked) one wavefront each
        uint32_t nWork = cnts[0] + cnts[1];
        for (int pass = 0; nWork > 0; pass++) {
            if (pass > 256) { setError("plasship_assemble: comparator table did not converge"); return PLASSHIP_ERR_DEVICE; }
            a.ambKeys = ctx->d_ambKeys.as<uint32_t>(); a.ambVals = ctx->d_ambVals.as<uint8_t>(); a.ambMask = ctx->ambSlots ? ctx->ambSlots - 1 : 0;
            a.redoList = dRedo[pass & 1].as<uint32_t>();
            PH_CHECK(hipMemsetAsync(dCnt.p, 0, 8, st));
            auto launchWave = [&](const uint32_t *list, uint32_t n) {
                if (!n) return;
                a.queryList = list; a.nQueryList = n;
                if (guided) hipLaunchKernelGGL(assembleNuclKernel<true>, dim3(std::min<uint32_t>(n, (uint32_t) ctx->numCU * 16)), dim3(64), 0, st, a);
                else hipLaunchKernelGGL(assembleNuclKernel<false>, dim3(std::min<uint32_t>(n, (uint32_t) ctx->numCU * 16)), dim3(64), 0, st, a);
            };
            if (pass == 0) {
                if (cnts[0]) {
                    a.queryList = dSmallList.as<uint32_t>(); a.nQueryList = cnts[0];
                    if (tuneInt("NUCL_CLASSES", 1) == 1) {                    // PLASSHIP_TUNE_NUCL_CLASSES=2: id order
                        DevBuf dCls;
                        if (dCls.alloc(2 * NW_CLASSES * 4) != hipSuccess) { setError("plasship_assemble: out of device memory"); return PLASSHIP_ERR_DEVICE; }
                        PH_CHECK(hipMemsetAsync(dCls.p, 0, 2 * NW_CLASSES * 4, st));
                        const unsigned g = std::min<uint32_t>((cnts[0] + 255) / 256, (uint32_t) ctx->numCU * 8);
                        hipLaunchKernelGGL(nuclClassCountKernel, dim3(g), dim3(256), 0, st, (const uint32_t *) dSmallList.as<uint32_t>(), cnts[0], (const uint64_t *) dBytes.as<uint64_t>(), dCls.as<uint32_t>());
                        hipLaunchKernelGGL(nuclClassScatterKernel, dim3(g), dim3(256), 0, st, (const uint32_t *) dSmallList.as<uint32_t>(), cnts[0], (const uint64_t *) dBytes.as<uint64_t>(), dCls.as<uint32_t>(),
                                           dMidList.as<uint32_t>());          // (the 64-lane list is a protein tier: unused here)
                        a.queryList = dMidList.as<uint32_t>();
                    }
                    const uint32_t grid = std::min<uint32_t>((cnts[0] + NT_BLOCK - 1) / NT_BLOCK, (uint32_t) ctx->numCU * 8);
                    if (guided) hipLaunchKernelGGL(assembleNuclThreadKernel<true>, dim3(grid), dim3(NT_BLOCK), 0, st, a);
                    else hipLaunchKernelGGL(assembleNuclThreadKernel<false>, dim3(grid), dim3(NT_BLOCK), 0, st, a);
                }
                launchWave(dMid32List.as<uint32_t>(), cnts[1]);
            } else launchWave(dRedo[(pass + 1) & 1].as<uint32_t>(), nWork);
            uint32_t cnt[2] = {0, 0};
            PH_CHECK(hipMemcpyAsync(cnt, dCnt.p, 8, hipMemcpyDeviceToHost, st));
            PH_CHECK(plasship::streamSync(st));
            PH_CHECK(hipGetLastError());
            nWork = cnt[1];
            if (nWork == 0) break;
            const uint32_t nNeed = std::min(cnt[0], needCap);
            if (nNeed == 0) { setError("plasship_assemble: queries aborted without a missing comparator tuple"); return PLASSHIP_ERR_DEVICE; }
            std::vector<uint32_t> need((size_t) nNeed * 4);
            PH_COPY_SYNC(ctx->stream, need.data(), dNeed.p, need.size() * 4, hipMemcpyDeviceToHost);
            const int rc = ambTableInsert(ctx, need.data(), nNeed);
            if (rc != PLASSHIP_OK) return rc;
        }
        PH_CHECK(hipEventRecord(ctx->ev[3], st));
        for (int e = 4; e <= 7; e++) PH_CHECK(hipEventRecord(ctx->ev[e], st));
    } else {
    // (round 4: the protein tiers' lists in work classes like the nucleotide list changed nothing — 82.2 against 81.0 ms for the stage: the
    //  tiers already group the queries by queue size, and id order keeps a wavefront's alignment records adjacent; profiles/r04_ab_knobs.txt)
    // wavefronts per SIMD of the register-queue kernels (PLASSHIP_TUNE_ASM16 / ASM64): the grid is what the CUs hold at once
    const int w16 = tuneInt("ASM16", 5), w64 = tuneInt("ASM64", 4);      // round 3 (after the copy tails went word-wise): 16.6 ms at 5 wavefronts, 17.0 at 6, 18.0 at 4
    const uint32_t gx = (uint32_t) std::max(1, tuneInt("ASM_GRIDX", 1));      // grid = gx x what the CUs hold at once (round 6 A/B: finer shares against the tail)
    const dim3 g16(std::min<uint32_t>((a.nSmall + 15) / 16, (uint32_t) ctx->numCU * (uint32_t) w16 * gx)), g64(std::min<uint32_t>((a.nMid + 3) / 4, (uint32_t) ctx->numCU * (uint32_t) w64 * gx));
    // (round 5: the four tiers — disjoint queries — launched side by side on four streams, so that one tier's tail of long queues lies under
    //  the next tier: 88.8 against 82.7 ms for the stage; the tiers' wavefronts evict each other's lines.  profiles/r05_ab_knobs.txt)
    PH_CHECK(hipEventRecord(ctx->ev[2], st));
    if (a.nSmall) {
        if (w16 == 6) hipLaunchKernelGGL((assembleGroupKernel<16, 6>), g16, dim3(256), 0, st, a);
        else if (w16 == 5) hipLaunchKernelGGL((assembleGroupKernel<16, 5>), g16, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((assembleGroupKernel<16, 4>), g16, dim3(256), 0, st, a);
    }
    PH_CHECK(hipEventRecord(ctx->ev[3], st));
    PH_CHECK(hipEventRecord(ctx->ev[4], st));
    if (a.nMid32) hipLaunchKernelGGL((assembleGroupKernel<32, 5>), dim3(std::min<uint32_t>((a.nMid32 + 7) / 8, (uint32_t) ctx->numCU * 5u * gx)), dim3(256), 0, st, a);      // (4 wavefronts per SIMD: 28.3 against 28.0 ms)
    if (a.nMid) {
        if (w64 == 5) hipLaunchKernelGGL((assembleGroupKernel<64, 5>), g64, dim3(256), 0, st, a);
        else if (w64 == 4) hipLaunchKernelGGL((assembleGroupKernel<64, 4>), g64, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((assembleGroupKernel<64, 3>), g64, dim3(256), 0, st, a);
    }
    PH_CHECK(hipEventRecord(ctx->ev[5], st));
    PH_CHECK(hipEventRecord(ctx->ev[6], st));
    if (a.nBig) hipLaunchKernelGGL(assembleBigKernel, dim3(std::min<uint32_t>((a.nBig + 3) / 4, (uint32_t) ctx->numCU * (uint32_t) tuneInt("ASMBIG", 4))), dim3(256), 0, st, a);
    PH_CHECK(hipEventRecord(ctx->ev[7], st));
    }
    PH_TRACE(st, "assemble: extension kernels");
    // ---- output DB(s): extended queries + carried-over sequences, in key order ----
    plasship_seqdb *o = nullptr, *oAa = nullptr;
    unsigned long long hs[16] = {0};
    DevBuf dGathered, dGatheredAa;
    const char *arenaP = dArena.as<char>(), *aaArenaP = dAaArena.as<char>();
    if (commOf(ctx)) {
        const int rc = mergeExtended(ctx, N, guided, par->keep_target, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>(), dArena.as<char>(),
                                     dAaNewLen.as<uint32_t>(), dAaNewStart.as<uint64_t>(), dAaArena.as<char>(), dTmp.p, tmpBytes, dGathered, dGatheredAa);
        if (rc != PLASSHIP_OK) return rc;
        arenaP = dGathered.as<char>(); aaArenaP = dGatheredAa.as<char>();
    }
    int rcOut = buildOutputDB(ctx, db, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>(), arenaP, par->keep_target, dTmp.p, tmpBytes, &o,
                              dStats.p, hs, 128, guided ? (hipEvent_t) nullptr : ctx->ev[1], guided);
    if (rcOut != PLASSHIP_OK) return rcOut;
    std::unique_ptr<plasship_seqdb> holdO(o), holdAa;              // released to the caller on success only
    if (guided) {
        rcOut = buildOutputDB(ctx, aaDb, dFlags.as<uint32_t>(), dAaNewLen.as<uint32_t>(), dAaNewStart.as<uint64_t>(), aaArenaP, par->keep_target, dTmp.p, tmpBytes, &oAa,
                              nullptr, nullptr, 0, nullptr, true);
        if (rcOut != PLASSHIP_OK) return rcOut;
        holdAa.reset(oAa);
    }
    if (guided) { PH_CHECK(hipEventRecord(ctx->ev[1], st)); PH_CHECK(plasship::streamSync(st)); }
    if (hs[12]) { setError("plasship_guided_assemble: an alignment asks for a protein fragment the twin does not have (coordinates are not codon aligned)"); return PLASSHIP_ERR_ARG; }
    if (stats) {
        stats->n_extended = hs[0]; stats->n_rescored = hs[1]; stats->out_residues = o->residues;
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); stats->ms_kernel = ms;
        float sum = 0;
        for (int t = 0; t < 3; t++) {
            ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[2 + 2 * t], ctx->ev[3 + 2 * t]); stats->ms_tier_kernel[t] = ms; sum += ms;
            stats->tier_alignments[t] = hs[3 + 3 * t]; stats->tier_query_residues[t] = hs[4 + 3 * t]; stats->tier_rescored_residues[t] = hs[5 + 3 * t];
        }
        stats->ms_assemble_kernel = sum;
        stats->n_alignments = al->nLines; stats->rescored_residues = hs[2];
        stats->db_appended_bytes = o->buildAppendedBytes; stats->db_copied_bytes = o->buildCopiedBytes;
    }
    *out = holdO.release();
    if (guided) *outAa = holdAa.release();
    return PLASSHIP_OK;
}

extern "C" int plasship_assemble(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_alns *al,
                                 const plasship_assemble_params *par, plasship_seqdb **out, plasship_assemble_stats *stats) {
    if (!ctx || !db || !al || !par || !out) { setError("plasship_assemble: bad argument"); return PLASSHIP_ERR_ARG; }
    return commFinish(ctx, assembleImpl(ctx, db, nullptr, al, par, out, nullptr, stats));
}

extern "C" int plasship_guided_assemble(plasship_ctx *ctx, const plasship_seqdb *nucl_db, const plasship_seqdb *aa_db, const plasship_alns *al,
                                        const plasship_assemble_params *par, plasship_seqdb **out_nucl, plasship_seqdb **out_aa,
                                        plasship_assemble_stats *stats) {
    if (!ctx || !nucl_db || !aa_db || !al || !par || !out_nucl || !out_aa) { setError("plasship_guided_assemble: bad argument"); return PLASSHIP_ERR_ARG; }
    if (nucl_db->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES || aa_db->dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS) { setError("plasship_guided_assemble: needs a nucleotide DB and its protein twin DB"); return PLASSHIP_ERR_ARG; }
    if (nucl_db->n != aa_db->n) { setError("plasship_guided_assemble: the two DBs differ in size"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    // the reference addresses the twin of entry i by the same id (guidedassembleresult.cpp:363): the key sets must be equal
    bool differ = false;
    // (every return behind PH_ENTER goes through commFinish: a rank that left alone would leave its peers waiting in their next collective)
    { const int rc = deviceKeysDiffer(ctx, nucl_db->d_key.as<uint32_t>(), aa_db->d_key.as<uint32_t>(), nucl_db->n, &differ); if (rc) return commFinish(ctx, rc); }
    if (differ) { setError("plasship_guided_assemble: nucleotide and protein DB have different keys"); return commFinish(ctx, PLASSHIP_ERR_ARG); }
    return commFinish(ctx, assembleImpl(ctx, nucl_db, aa_db, al, par, out_nucl, out_aa, stats));
}
