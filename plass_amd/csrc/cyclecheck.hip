// plasship: cyclecheck on gfx950.  Product code.
//
// Reference behaviour reproduced (src/assembler/cyclecheck.cpp:94-283): per nucleotide contig the k-mers (k = 22, index
// = sum code[i] * 4^i over the codes A0 C1 T2 G3 X4, windows with X included) of the front, middle and back third are
// matched against each other — the FIRST (smallest position) front occurrence of a k-mer against every middle and back
// occurrence, the first middle occurrence against every back occurrence (:151-216) — and every match on a diagonal
// >= seqLen/3 increments a histogram.  The first diagonal d whose band (+- 1 % of the diagonal's length, only bins not
// larger than its own) holds more than 0.2 hits per k-mer position makes the contig circular (:241-269); it is written
// out whole or cut at that diagonal (--chop-cycle).  The third a k-mer belongs to is decided by the position BEFORE the
// reference's iterator advances (:118-136): k-mer p counts as front if 1 <= p <= L/3 + 1, middle if p <= 2(L/3) + 1,
// and the k-mer at position 0 ("-1" as unsigned) lands in the back list.
//
// MI355X-first: the reference sorts three k-mer lists per contig; only "first occurrence per k-mer" matters, so a hash
// table of (k-mer -> smallest front position, smallest middle position) replaces the sorts — in LDS, one wavefront per
// sequence, for everything up to 3 000 nt (reads and most contigs; the table is sized to the sequence), in HBM scratch with one workgroup per sequence for the
// long contigs.  Floating point: the hit rate is one float division compared with the double 0.2, as in the reference.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <memory>
#include <cstring>

namespace plasship {

constexpr int CC_K = 22;
constexpr unsigned long long CC_EMPTY = ~0ULL;

struct CycArgs {
    SeqView s;
    const unsigned char *map;        // 256-entry letter -> code
    const uint32_t *list; uint32_t nList;
    uint32_t *split;                 // [n] split diagonal, 0 = not circular
    // long tier: per list entry scratch
    unsigned long long *keys; uint32_t *minF, *minM, *hits;
    const uint64_t *slotOff; const uint32_t *slotCnt; const uint64_t *hitOff;
};

__device__ __forceinline__ int kmerClass(uint32_t p, uint32_t third) {      // 0 front, 1 middle, 2 back
    if (p == 0) return 2;
    if (p <= third + 1) return 0;
    if (p <= 2 * third + 1) return 1;
    return 2;
}
__device__ __forceinline__ uint32_t hashSlot(unsigned long long k, uint32_t mask) { return (uint32_t) ((k * 0x9E3779B97F4A7C15ULL) >> 40) & mask; }

// the reference's band test for diagonal bin d (cyclecheck.cpp:245-266); hits has 2 * third + 1 bins
__device__ __forceinline__ bool bandPasses(const uint32_t *hits, uint32_t d, uint32_t third, uint32_t L) {
    const uint32_t hd = hits[d];
    const uint32_t diag = d + third, diaglen = L - diag;
    const uint32_t gapwindow = (uint32_t) ((double) diaglen * 0.01);
    const uint32_t lower = (uint32_t) max(0, (int) (d - gapwindow));
    const uint32_t upper = min(d + gapwindow, 2 * third);
    uint32_t band = 0;
    for (uint32_t i = lower; i <= upper; i++) { const uint32_t h = hits[i]; if (h <= hd) band += h; }
    const float rate = (float) band / (float) ((unsigned long long) diaglen - (unsigned long long) CC_K + 1ULL);
    return (double) rate > 0.2;
}

// ---- round 4: one table entry = (k-mer << 18 | position), the smallest position of a k-mer kept by a 64-bit atomicMin; two phases
//      per sequence (front k-mers in the table, middle and back k-mers looked up; middle k-mers in the table, back k-mers looked up)
//      instead of two minima per entry, so an entry is 8 bytes instead of 16 and a phase holds a third of the sequence instead of two:
//      a 3 000-nt contig needs 16 KB of LDS instead of 64 KB + staging + histogram (rounds 1-3: 4-5 wavefronts per CU in the two larger
//      wave tiers, 125 + 82 ms per call on the contigs of configs[4]'s last iteration).  The histogram of diagonals lives in HBM
//      scratch and is only made when a first pass over both phases has seen a hit at all — a contig without an internal repeat of
//      22 letters between its thirds, i.e. nearly every one, never touches it.  k-mers are rolled along contiguous ranges of
//      positions (one letter per position instead of 22).  Contigs longer than the table holds take several passes per phase, each
//      over the k-mers of one hash range: a workgroup with 64 KB of LDS handles any length below 2^18 (rounds 1-3: tables in HBM,
//      global atomics, batches of 2 GB with a host wait each — 20 launches of 9 ms in that iteration).
constexpr int CC_POS_BITS = 18;
constexpr uint32_t CC_PACK_MAXL = (1u << CC_POS_BITS) - 1;
struct CycTabArgs {
    SeqView s;
    const unsigned char *map;
    const uint32_t *list; uint32_t nList;
    uint32_t *split;
    uint32_t *hist; uint64_t histStride;          // per workgroup: 2 * (longest sequence of the tier / 3) + 2 bins
    uint32_t *fallList; uint32_t *fallCount;      // sequences whose table overflowed (never seen; the HBM kernel below takes them)
    uint32_t forcePasses;                         // tests only (PLASSHIP_TUNE_CYC_PASSES): too few passes, so that long contigs DO overflow
};
template <int THREADS, int SLOTS, int STAGE>      // STAGE: longest sequence whose letter codes are staged in LDS (0: read from HBM)
__global__ __launch_bounds__(THREADS) void cycleTableKernel(CycTabArgs a) {
    __shared__ unsigned char sMap[256];
    __shared__ __attribute__((aligned(16))) unsigned char sNum[STAGE ? STAGE + 32 : 16];
    __shared__ unsigned long long sTab[SLOTS];
    __shared__ uint32_t sHits, sFirst, sOver;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t mask = SLOTS - 1;
    constexpr int slotShift = 64 - __builtin_ctz(SLOTS);
    for (uint32_t i = tid; i < 256; i += THREADS) sMap[i] = a.map[i];
    __syncthreads();
    uint32_t *hist = a.hist + (uint64_t) blockIdx.x * a.histStride;
    // Round 6: a sequence used to start with list entry -> length, offset -> its letters BYTE BY BYTE (one 64- or 256-byte load instruction per trip of the
    // staging loop, each waited for: 6 / 24 / 48 dependent loads in the three staged tiers).  Now the list entry travels two sequences ahead, length and
    // offset one ahead, and the letters are fetched 16 bytes per thread with all loads of a sequence in flight together.
    uint32_t idNext = blockIdx.x < a.nList ? a.list[blockIdx.x] : 0u;
    uint32_t idAhead = blockIdx.x + gridDim.x < a.nList ? a.list[blockIdx.x + gridDim.x] : 0u;
    uint32_t LNext = blockIdx.x < a.nList ? a.s.len[idNext] : 0u; uint64_t offNext = blockIdx.x < a.nList ? a.s.off[idNext] : 0ull;
    for (uint32_t w = blockIdx.x; w < a.nList; w += gridDim.x) {
        const uint32_t id = idNext;
        const uint32_t L = LNext;
        const char *seq = a.s.data + offNext;
        idNext = idAhead;
        if (w + gridDim.x < a.nList) { LNext = a.s.len[idNext]; offNext = a.s.off[idNext]; }
        idAhead = (w + 2 * gridDim.x < a.nList) ? a.list[w + 2 * gridDim.x] : 0u;
        const uint32_t third = L / 3, nk = L - CC_K + 1, nBins = 2 * third + 1;
        // positions by class (kmerClass): front [1, third + 1], middle [third + 2, 2 third + 1], back [2 third + 2, nk - 1]; position 0 is
        // "back" too, but its diagonals are <= 0 and never counted
        const uint32_t fLo = 1, fHi = min(third + 1, nk - 1), mLo = third + 2, mHi = min(2 * third + 1, nk - 1), bLo = 2 * third + 2, bHi = nk - 1;
        const uint32_t passes = a.forcePasses ? a.forcePasses : (third + 2 + SLOTS / 2 - 1) / (SLOTS / 2);          // load <= 0.5 per pass on average
        if (STAGE) {
            constexpr int NCH = STAGE ? (STAGE + 16 * THREADS - 1) / (16 * THREADS) : 1;
            uint4 wv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) { const uint32_t p = 16u * ((uint32_t) c * THREADS + tid); wv[c] = make_uint4(0, 0, 0, 0); if (p < L) __builtin_memcpy(&wv[c], seq + p, 16); }      // (buffers are padded past their ends)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint32_t p = 16u * ((uint32_t) c * THREADS + tid);
                if (p < L) {
                    const uint32_t r4[4] = {wv[c].x, wv[c].y, wv[c].z, wv[c].w};
                    uint32_t cw[4];
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        cw[d] = (uint32_t) sMap[r4[d] & 0xFFu] | ((uint32_t) sMap[(r4[d] >> 8) & 0xFFu] << 8) | ((uint32_t) sMap[(r4[d] >> 16) & 0xFFu] << 16) | ((uint32_t) sMap[r4[d] >> 24] << 24);
                    *reinterpret_cast<uint4 *>(&sNum[p]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);      // (codes behind L are never read)
                }
            }
        }
        if (tid == 0) { sHits = 0; sFirst = 0xFFFFFFFFu; sOver = 0; }
        __syncthreads();
        auto code = [&](uint32_t i) -> unsigned long long { return STAGE ? (unsigned long long) sNum[i] : (unsigned long long) sMap[(unsigned char) seq[i]]; };
        // every thread walks a contiguous part of [lo, hi] with a rolling index: idx' = (idx - code[p]) / 4 + code[p + k] * 4^(k-1)
        auto walk = [&](uint32_t lo, uint32_t hi, uint32_t pass, auto &&visit) {
            if (lo > hi) return;
            const uint32_t n = hi - lo + 1, chunk = (n + THREADS - 1) / THREADS;
            const uint32_t p0 = lo + tid * chunk, p1 = min(hi + 1, p0 + chunk);
            if (p0 >= p1) return;
            unsigned long long k = 0;
            for (int i = CC_K - 1; i >= 0; i--) k = k * 4ULL + code(p0 + (uint32_t) i);
            for (uint32_t p = p0; p < p1; p++) {
                if (passes == 1 || (uint32_t) ((((k * 0xD6E8FEB86659FD93ULL) >> 40) * passes) >> 24) == pass) visit(p, k);
                if (p + 1 < p1) k = ((k - code(p)) >> 2) + (code(p + CC_K) << (2 * (CC_K - 1)));
            }
        };
        auto insert = [&](uint32_t p, unsigned long long k) {
            const unsigned long long word = (k << CC_POS_BITS) | p;
            uint32_t sl = (uint32_t) ((k * 0x9E3779B97F4A7C15ULL) >> slotShift);
            for (uint32_t probes = 0; probes < (uint32_t) SLOTS; probes++) {
                unsigned long long cur = sTab[sl];
                if (cur == CC_EMPTY) { cur = atomicCAS(&sTab[sl], CC_EMPTY, word); if (cur == CC_EMPTY) return; }
                if ((cur >> CC_POS_BITS) == k) { atomicMin(&sTab[sl], word); return; }
                sl = (sl + 1) & mask;
            }
            sOver = 1;
        };
        for (int round = 0; round < 2; round++) {                 // round 0: is there any hit?  round 1: the histogram
            if (round == 1) {
                for (uint32_t i = tid; i < nBins; i += THREADS) hist[i] = 0;
                __threadfence();
                __syncthreads();
            }
            auto lookup = [&](uint32_t p, unsigned long long k) {
                uint32_t sl = (uint32_t) ((k * 0x9E3779B97F4A7C15ULL) >> slotShift);
                for (uint32_t probes = 0; probes < (uint32_t) SLOTS; probes++) {
                    const unsigned long long cur = sTab[sl];
                    if (cur == CC_EMPTY) return;
                    if ((cur >> CC_POS_BITS) == k) {
                        const int diag = (int) p - (int) (uint32_t) (cur & CC_PACK_MAXL);
                        if (diag >= (int) third) { if (round == 0) sHits = 1; else atomicAdd(&hist[diag - (int) third], 1u); }
                        return;
                    }
                    sl = (sl + 1) & mask;
                }
            };
            for (int phase = 0; phase < 2; phase++) {
                for (uint32_t pass = 0; pass < passes; pass++) {
                    for (uint32_t i = tid; i < (uint32_t) SLOTS; i += THREADS) sTab[i] = CC_EMPTY;
                    __syncthreads();
                    if (phase == 0) walk(fLo, fHi, pass, insert); else walk(mLo, mHi, pass, insert);
                    __syncthreads();
                    if (phase == 0) walk(mLo, mHi, pass, lookup);
                    walk(bLo, bHi, pass, lookup);
                    __syncthreads();
                }
            }
            if (sOver || sHits == 0) break;
        }
        if (sOver) {
            if (tid == 0) a.fallList[atomicAdd(a.fallCount, 1u)] = id;
        } else {
            if (sHits) {
                __threadfence();
                __syncthreads();
                for (uint32_t base = 0; base < 2 * third; base += THREADS) {
                    const uint32_t d = base + tid;
                    if (d < 2 * third && hist[d] != 0 && bandPasses(hist, d, third, L)) atomicMin(&sFirst, d);
                    __syncthreads();
                    if (sFirst != 0xFFFFFFFFu) break;
                }
            }
            if (tid == 0) a.split[id] = (sFirst != 0xFFFFFFFFu) ? sFirst + third : 0u;
        }
        __syncthreads();
    }
}

// ---- one workgroup per long sequence, table and histogram in HBM scratch -----------------------------------------------
__global__ __launch_bounds__(256) void cycleBlockKernel(CycArgs a) {
    __shared__ unsigned char sMap[256];
    __shared__ uint32_t sAny, sFirst;
    for (int i = threadIdx.x; i < 256; i += 256) sMap[i] = a.map[i];
    __syncthreads();
    for (uint32_t w = blockIdx.x; w < a.nList; w += gridDim.x) {
        const uint32_t id = a.list[w];
        const uint32_t L = a.s.len[id];
        const char *seq = a.s.data + a.s.off[id];
        const uint32_t third = L / 3, nk = L - CC_K + 1, nBins = 2 * third + 1;
        const uint32_t slots = a.slotCnt[w], mask = slots - 1;
        unsigned long long *keys = a.keys + a.slotOff[w];
        uint32_t *minF = a.minF + a.slotOff[w], *minM = a.minM + a.slotOff[w], *hits = a.hits + a.hitOff[w];
        for (uint32_t i = threadIdx.x; i < slots; i += 256) { keys[i] = CC_EMPTY; minF[i] = 0xFFFFFFFFu; minM[i] = 0xFFFFFFFFu; }
        for (uint32_t i = threadIdx.x; i < nBins; i += 256) hits[i] = 0;
        if (threadIdx.x == 0) { sAny = 0; sFirst = 0xFFFFFFFFu; }
        __syncthreads();
        // every thread walks a contiguous range of positions with a rolling index: idx' = (idx - code[p]) / 4 + code[p + k] * 4^(k-1)
        const uint32_t chunk = (nk + 255) / 256;
        const uint32_t p0 = threadIdx.x * chunk, p1 = min(nk, p0 + chunk);
        auto code = [&](uint32_t i) { return (unsigned long long) sMap[(unsigned char) seq[i]]; };
        auto first = [&](uint32_t p) { unsigned long long v = 0; for (int i = CC_K - 1; i >= 0; i--) v = v * 4ULL + code(p + i); return v; };
        if (p0 < p1) {
            unsigned long long k = first(p0);
            for (uint32_t p = p0; p < p1; p++) {
                const int c = kmerClass(p, third);
                if (c != 2) {
                    uint32_t sl = hashSlot(k, mask);
                    for (;;) {
                        const unsigned long long prev = atomicCAS(&keys[sl], CC_EMPTY, k);
                        if (prev == CC_EMPTY || prev == k) break;
                        sl = (sl + 1) & mask;
                    }
                    atomicMin(c == 0 ? &minF[sl] : &minM[sl], p);
                }
                if (p + 1 < p1) k = ((k - code(p)) >> 2) + (code(p + CC_K) << (2 * (CC_K - 1)));
            }
        }
        __threadfence();
        __syncthreads();
        if (p0 < p1) {
            unsigned long long k = first(p0);
            for (uint32_t p = p0; p < p1; p++) {
                const int c = kmerClass(p, third);
                if (c != 0) {
                    uint32_t sl = hashSlot(k, mask);
                    for (;;) {
                        const unsigned long long kk = keys[sl];
                        if (kk == k || kk == CC_EMPTY) { if (kk != k) sl = 0xFFFFFFFFu; break; }
                        sl = (sl + 1) & mask;
                    }
                    if (sl != 0xFFFFFFFFu) {
                        const uint32_t mf = minF[sl], mm = minM[sl];
                        if (mf != 0xFFFFFFFFu) { const int diag = (int) p - (int) mf; if (diag >= (int) third) { atomicAdd(&hits[diag - (int) third], 1u); sAny = 1; } }
                        if (c == 2 && mm != 0xFFFFFFFFu) { const int diag = (int) p - (int) mm; if (diag >= (int) third) { atomicAdd(&hits[diag - (int) third], 1u); sAny = 1; } }
                    }
                }
                if (p + 1 < p1) k = ((k - code(p)) >> 2) + (code(p + CC_K) << (2 * (CC_K - 1)));
            }
        }
        __threadfence();
        __syncthreads();
        if (sAny) {
            for (uint32_t base = 0; base < 2 * third; base += 256) {
                const uint32_t d = base + threadIdx.x;
                if (d < 2 * third && hits[d] != 0 && bandPasses(hits, d, third, L)) atomicMin(&sFirst, d);
                __syncthreads();
                if (sFirst != 0xFFFFFFFFu) break;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) a.split[id] = (sFirst != 0xFFFFFFFFu) ? sFirst + third : 0u;
        __syncthreads();
    }
}

// tier of every sequence: 0-2 one wavefront per sequence with a table of 256 / 1024 / 2048 entries in LDS (a phase inserts at most
// L/3 + 2 k-mers, load <= 0.5; the LDS a wavefront holds decides how many sequences a CU works on at a time: reads and merged read
// pairs get the small instantiation), 3 one workgroup of 256 threads with 8192 entries and as many passes as the length asks for,
// 4 (2^18 letters and more: the position no longer fits beside the k-mer) the HBM kernel; sequences without a k-mer or at / above
// --max-seq-len are not circular
constexpr uint32_t CC_L0 = 380, CC_L1 = 1532, CC_L2 = 3068;
constexpr int CC_TIERS = 5;
// `known` (nullable): one byte per id, 0 = the entry is byte for byte an entry of a DB whose entries are all known not to be circular
// (the "rest" DB of the last call, plasship_ctx::cycKnownGen) — it is not looked at again.  A workgroup classifies 2 048 ids per round
// and reserves its part of every list with one atomic per tier (rounds 1-3: one per wavefront and tier, 10 ms at 35 M sequences).
constexpr int CC_TIER_PER = 8;
__global__ __launch_bounds__(256) void cycleTierKernel(SeqView s, uint64_t maxSeqLen, const unsigned char *__restrict__ known, uint32_t *__restrict__ lists, uint32_t *__restrict__ counts,
                                                       uint32_t *__restrict__ split) {
    __shared__ uint32_t sCnt[CC_TIERS + 1], sBase[CC_TIERS + 1];
    for (uint64_t b0 = (uint64_t) blockIdx.x * (256 * CC_TIER_PER); b0 < s.n; b0 += (uint64_t) gridDim.x * (256 * CC_TIER_PER)) {
        if (threadIdx.x <= CC_TIERS) sCnt[threadIdx.x] = 0;
        __syncthreads();
        int tier[CC_TIER_PER]; uint32_t rank[CC_TIER_PER];
#pragma unroll
        for (int j = 0; j < CC_TIER_PER; j++) {
            const uint64_t id = b0 + (uint64_t) j * 256 + threadIdx.x;
            tier[j] = -1; rank[j] = 0;
            if (id < s.n) {
                const uint32_t L = s.len[id];
                split[id] = 0;
                if (L >= (uint32_t) CC_K && (uint64_t) L < maxSeqLen) {
                    if (known && known[id] == 0) tier[j] = CC_TIERS;                      // counted, not listed
                    else tier[j] = L <= CC_L0 ? 0 : (L <= CC_L1 ? 1 : (L <= CC_L2 ? 2 : (L <= CC_PACK_MAXL ? 3 : 4)));
                    rank[j] = atomicAdd(&sCnt[tier[j]], 1u);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x <= CC_TIERS) sBase[threadIdx.x] = sCnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], sCnt[threadIdx.x]) : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CC_TIER_PER; j++)
            if (tier[j] >= 0 && tier[j] < CC_TIERS) lists[(size_t) tier[j] * s.n + sBase[tier[j]] + rank[j]] = (uint32_t) (b0 + (uint64_t) j * 256 + threadIdx.x);
        __syncthreads();
    }
}

__global__ void cycleGatherLenKernel(const uint32_t *__restrict__ len, const uint32_t *__restrict__ ids, uint32_t n, uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = len[ids[i]];
}

__global__ void cycleFlagsKernel(SeqView s, const uint32_t *__restrict__ split, int chop, int invert, uint32_t *__restrict__ flags, uint32_t *__restrict__ newLen,
                                 uint64_t *__restrict__ newStart) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < s.n; id += gridDim.x * blockDim.x) {
        const uint32_t sp = split[id];
        const bool cyc = sp != 0;
        if (invert) { flags[id] = cyc ? 0x80u : 0u; newLen[id] = 0; newStart[id] = 0; }            // the remainder: everything that is not circular
        else { flags[id] = cyc ? 0x20u : 0x80u; newLen[id] = cyc ? (chop ? sp : s.len[id]) : 0u; newStart[id] = s.off[id]; }
    }
}

}  // namespace plasship
using namespace plasship;

extern "C" int plasship_cyclecheck(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_cyclecheck_params *par, plasship_seqdb **out_cycle,
                                   plasship_seqdb **out_rest, plasship_cyclecheck_stats *stats) {
    if (!ctx || !db || !par || !out_cycle) { setError("plasship_cyclecheck: bad argument"); return PLASSHIP_ERR_ARG; }
    if (db->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES) { setError("Module cyclecheck only supports nucleotide input database"); return PLASSHIP_ERR_ARG; }
    if (db->maxEntryLen >= (1u << 30)) { setError("plasship_cyclecheck: sequence too long"); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const SeqView sv = db->view();
    DevBuf dMap, dLists, dCounts, dSplit, dFlags, dNewLen, dNewStart, dTmp;
    const size_t tmpBytes = exclusiveScanTmpBytes((size_t) N + 2);
    if (dMap.alloc(256) != hipSuccess || dLists.alloc(((size_t) CC_TIERS * N + 1) * 4) != hipSuccess || dCounts.alloc(32) != hipSuccess || dSplit.alloc(((size_t) N + 1) * 4) != hipSuccess ||
        dFlags.alloc(((size_t) N + 1) * 4) != hipSuccess || dNewLen.alloc(((size_t) N + 1) * 4) != hipSuccess || dNewStart.alloc(((size_t) N + 1) * 8) != hipSuccess ||
        dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_cyclecheck: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    PH_CHECK(hipMemcpyAsync(dMap.p, aa2numTable(true, 5), 256, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemsetAsync(dCounts.p, 0, 32, st));
    const unsigned gridN = std::min<uint32_t>((N + 255) / 256 + 1, (uint32_t) ctx->numCU * 16);
    // entries the last call on this context has already found linear (they sit unchanged in a descendant of its "rest" DB) are skipped;
    // PLASSHIP_TUNE_CYCSKIP=2 checks everything
    const bool skipKnown = db->ancestorGen != 0 && db->ancestorGen == ctx->cycKnownGen && ctx->cycKnownMaxLen == (uint64_t) par->max_seq_len && db->d_changed.p && tuneInt("CYCSKIP", 1) == 1;
    if (N) hipLaunchKernelGGL(cycleTierKernel, dim3(std::min<uint32_t>((N + 2047) / 2048, (uint32_t) ctx->numCU * 8)), dim3(256), 0, st, sv, (uint64_t) par->max_seq_len,
                              skipKnown ? db->d_changed.as<unsigned char>() : (const unsigned char *) nullptr, dLists.as<uint32_t>(), dCounts.as<uint32_t>(), dSplit.as<uint32_t>());
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PH_COPY_SYNC(st, cnt, dCounts.p, 32, hipMemcpyDeviceToHost);
    CycArgs a; memset(&a, 0, sizeof(a));
    a.s = sv; a.map = dMap.as<unsigned char>(); a.split = dSplit.as<uint32_t>();
    // tiers 0-3: tables in LDS; a sequence whose table overflowed (more distinct k-mers in a hash range than entries: not seen) joins tier 4
    DevBuf dHist, dFall;
    {
        const uint32_t maxL = (uint32_t) std::min<uint64_t>(db->maxEntryLen, CC_PACK_MAXL);
        const uint32_t grid[4] = {std::min<uint32_t>(cnt[0], (uint32_t) ctx->numCU * 32), std::min<uint32_t>(cnt[1], (uint32_t) ctx->numCU * 16),
                                  std::min<uint32_t>(cnt[2], (uint32_t) ctx->numCU * 8), std::min<uint32_t>(cnt[3], (uint32_t) ctx->numCU * 2)};
        const uint64_t stride[4] = {2 * (uint64_t) (CC_L0 / 3) + 2, 2 * (uint64_t) (CC_L1 / 3) + 2, 2 * (uint64_t) (CC_L2 / 3) + 2, 2 * (uint64_t) (maxL / 3) + 2};
        uint64_t histWords = 0, histOff[4];
        for (int t = 0; t < 4; t++) { histOff[t] = histWords; histWords += cnt[t] ? grid[t] * stride[t] : 0; }
        if (dHist.alloc((histWords + 1) * 4) != hipSuccess || dFall.alloc(4) != hipSuccess) { setError("plasship_cyclecheck: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemsetAsync(dFall.p, 0, 4, st));
        CycTabArgs ta; memset(&ta, 0, sizeof(ta));
        ta.s = sv; ta.map = dMap.as<unsigned char>(); ta.split = dSplit.as<uint32_t>();
        ta.fallList = dLists.as<uint32_t>() + 4 * (size_t) N + cnt[4]; ta.fallCount = dFall.as<uint32_t>();
        ta.forcePasses = (uint32_t) tuneInt("CYC_PASSES", 0);
        for (int t = 0; t < 4; t++) {
            if (!cnt[t]) continue;
            ta.list = dLists.as<uint32_t>() + (size_t) t * N; ta.nList = cnt[t]; ta.hist = dHist.as<uint32_t>() + histOff[t]; ta.histStride = stride[t];
            if (t == 0) hipLaunchKernelGGL((cycleTableKernel<64, 256, CC_L0>), dim3(grid[t]), dim3(64), 0, st, ta);
            else if (t == 1) hipLaunchKernelGGL((cycleTableKernel<64, 1024, CC_L1>), dim3(grid[t]), dim3(64), 0, st, ta);
            else if (t == 2) hipLaunchKernelGGL((cycleTableKernel<64, 2048, CC_L2>), dim3(grid[t]), dim3(64), 0, st, ta);
            else hipLaunchKernelGGL((cycleTableKernel<256, 8192, 0>), dim3(grid[t]), dim3(256), 0, st, ta);
        }
        uint32_t nFall = 0;
        PH_COPY_SYNC(st, &nFall, dFall.p, 4, hipMemcpyDeviceToHost);
        cnt[4] += nFall;
    }
    if (cnt[4]) {
        // long contigs: scratch tables in HBM, in batches of at most ~2 GB
        const uint32_t nLong = cnt[4];
        std::vector<uint32_t> lens(nLong);
        DevBuf dLens;
        if (dLens.alloc((size_t) nLong * 4) != hipSuccess) { setError("plasship_cyclecheck: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        // lengths of the long sequences in list order (the host has no copy of a device-produced DB's index)
        hipLaunchKernelGGL(cycleGatherLenKernel, dim3(std::min<uint32_t>((nLong + 255) / 256, 1024)), dim3(256), 0, st, db->d_len.as<uint32_t>(), dLists.as<uint32_t>() + 4 * (size_t) N, nLong, dLens.as<uint32_t>());
        PH_COPY_SYNC(st, lens.data(), dLens.p, (size_t) nLong * 4, hipMemcpyDeviceToHost);
        const uint64_t budgetSlots = 1ull << 27;        // 16 bytes per slot
        uint32_t b0 = 0;
        while (b0 < nLong) {
            std::vector<uint64_t> slotOff, hitOff; std::vector<uint32_t> slotCnt;
            uint64_t so = 0, ho = 0; uint32_t b1 = b0;
            while (b1 < nLong) {
                uint32_t slots = 1024; while ((uint64_t) slots * 3 < (uint64_t) lens[b1] * 4 + 64) slots <<= 1;      // load <= 0.5 for the 2L/3 front + middle k-mers
                if (b1 > b0 && so + slots > budgetSlots) break;
                slotOff.push_back(so); slotCnt.push_back(slots); hitOff.push_back(ho);
                so += slots; ho += 2 * (uint64_t) (lens[b1] / 3) + 2; b1++;
            }
            const uint32_t nb = b1 - b0;
            DevBuf dKeys, dMinF, dMinM, dHits, dSO, dSC, dHO;
            if (dKeys.alloc(so * 8) != hipSuccess || dMinF.alloc(so * 4) != hipSuccess || dMinM.alloc(so * 4) != hipSuccess || dHits.alloc(ho * 4) != hipSuccess ||
                dSO.alloc((size_t) nb * 8) != hipSuccess || dSC.alloc((size_t) nb * 4) != hipSuccess || dHO.alloc((size_t) nb * 8) != hipSuccess) {
                setError("plasship_cyclecheck: out of device memory for the long-contig tables"); return PLASSHIP_ERR_DEVICE;
            }
            PH_CHECK(hipMemcpyAsync(dSO.p, slotOff.data(), (size_t) nb * 8, hipMemcpyHostToDevice, st));
            PH_CHECK(hipMemcpyAsync(dSC.p, slotCnt.data(), (size_t) nb * 4, hipMemcpyHostToDevice, st));
            PH_CHECK(hipMemcpyAsync(dHO.p, hitOff.data(), (size_t) nb * 8, hipMemcpyHostToDevice, st));
            a.list = dLists.as<uint32_t>() + 4 * (size_t) N + b0; a.nList = nb;
            a.keys = dKeys.as<unsigned long long>(); a.minF = dMinF.as<uint32_t>(); a.minM = dMinM.as<uint32_t>(); a.hits = dHits.as<uint32_t>();
            a.slotOff = dSO.as<uint64_t>(); a.slotCnt = dSC.as<uint32_t>(); a.hitOff = dHO.as<uint64_t>();
            hipLaunchKernelGGL(cycleBlockKernel, dim3(std::min<uint32_t>(nb, (uint32_t) ctx->numCU * 4)), dim3(256), 0, st, a);
            PH_CHECK(plasship::streamSync(st));
            PH_CHECK(hipGetLastError());
            b0 = b1;
        }
    }
    // the circular sequences (whole or cut) ...
    plasship_seqdb *oc = nullptr, *orest = nullptr;
    if (N) hipLaunchKernelGGL(cycleFlagsKernel, dim3(gridN), dim3(256), 0, st, sv, dSplit.as<uint32_t>(), par->chop_cycle ? 1 : 0, 0, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>());
    int rc = buildOutputDB(ctx, db, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>(), sv.data, 0, dTmp.p, tmpBytes, &oc, nullptr, nullptr, 0, ctx->ev[1],
                           true);      // (noAppend: the cycle DB is written out and freed at once — its entries do not belong in the chain's long-lived heap)
    if (rc != PLASSHIP_OK) return rc;
    std::unique_ptr<plasship_seqdb> holdC(oc);
    oc->dbtype = PLASSHIP_DBTYPE_NUCLEOTIDES;
    // ... and, if asked for, everything else (what the workflow continues with: "<db>_noneCycle", data/nuclassemble.sh:27-35)
    if (out_rest) {
        if (N) hipLaunchKernelGGL(cycleFlagsKernel, dim3(gridN), dim3(256), 0, st, sv, dSplit.as<uint32_t>(), 0, 1, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>());
        rc = buildOutputDB(ctx, db, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>(), sv.data, 0, dTmp.p, tmpBytes, &orest);
        if (rc != PLASSHIP_OK) return rc;
        ctx->cycKnownGen = orest->gen; ctx->cycKnownMaxLen = (uint64_t) par->max_seq_len;      // every entry of it has split == 0
    }
    if (stats) {
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        stats->ms_kernel = ms; stats->n_cyclic = oc->n; stats->n_wave_small = cnt[0]; stats->n_wave_large = cnt[1] + cnt[2]; stats->n_block = cnt[3] + cnt[4]; stats->n_known = cnt[5];
    }
    *out_cycle = holdC.release();
    if (out_rest) *out_rest = orest;
    return PLASSHIP_OK;
}
