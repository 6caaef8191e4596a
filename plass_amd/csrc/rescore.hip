// plasship: rescorediagonal on gfx950 (rows R1–R5 of SURVEY.md §8a).  Product code.
//
// Reference behaviour reproduced (file:line in /root/reference/lib/mmseqs/src):
//   alignment/rescorediagonal.cpp:193-334   per-hit loop: canBeCovered gate, isIdentity, filters
//   alignment/DistanceCalculator.h:93-113   computeUngappedAlignment: try all ±65536 wraps of the u16
//                                           diagonal, strictly-better score wins, first wins ties
//   alignment/DistanceCalculator.h:115-175  overlap geometry for one diagonal
//   alignment/DistanceCalculator.h:204-220  mode 3 end-to-end score, '*' trimmed at either end
//   alignment/rescorediagonal.cpp:251-297   alnLen, coordinates, identity count, seqId, coverage
//
// Kernel design: ONE THREAD per candidate pair whenever the shorter sequence has at most 768 residues (rescoreKernel<1>: a read
// overlap is 30-150 columns; a wavefront per pair would idle most lanes and pay two reductions per pair), 16 lanes per pair on a
// list of the rest (rescoreKernel<16>, 8 residues per lane and step).  The 123x123 ASCII-indexed score table
// (SubstitutionMatrix.h:56-73; 15 KB) lives in LDS; a thread streams 16 residues of both sequences per round trip (unaligned
// 16-byte loads, the next 16 requested before the current ones are scored), looks the scores up in LDS and counts the identities
// of the 16 columns with one zero-byte test.  The integer and IEEE-exact fp arithmetic (float divisions, double fma/div for the
// bit score) follows per pair; the only transcendental piece — the E-value — is replaced by a host-built per-query-length
// minimum-score table, which is exact because E(score) is monotone (host_util.cpp).
// -ffp-contract=off: no float expression here may be fused behind our back.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <memory>
#include <cfloat>
#include <cstring>

namespace plasship {

struct RescoreArgs {
    SeqView q, t;
    const uint64_t *qoff;        // CSR over queries (only for nHits)
    const CandHit *hits;
    uint64_t nHits;
    AlnRec *out;                 // [nHits]: the record of pair h in slot h, accepted or not (sparse list, common.hpp)
    const uint32_t *minScore;    // [maxQLen+1] minimum raw score passing -e for that query length
    uint32_t minScoreLen;
    const signed char *mat;      // 123*123 in global, staged to LDS
    int sameDB, includeIdentity, reverseCapable;
    int covMode; float covThr;
    float seqIdThr; int alnLenThr, seqIdMode;
    double lambda, logK, ln2;
    unsigned long long *stats;   // [0] accepted, [1] overlap residues
    unsigned long long *longList, *longCount;   // hit indices queued for the 16-lane kernel
    uint32_t shortMax;           // min(qLen, tLen) up to which the thread-per-pair kernel scores a pair itself
    // lazy self hits (common.hpp: plasship_alns::selfPending).  mode 0: the pairs of a candidate list, identity pairs left as stubs when
    // lazySelf; mode 1 (finishSelfAlns): the record slots of an alignment list, only the stubs are scored (hits == nullptr: the pair is rebuilt
    // from its stub); mode 2: the same for the stubs of the queries of `queryList` only (the extension kernel for queues beyond 64 alignments
    // keeps the self hit in its HBM-resident queue: assembleBigKernel)
    int mode, lazySelf;
    const uint32_t *queryList; uint32_t nQueryList;
};

__device__ __forceinline__ bool canBeCoveredDev(float covThr, int covMode, float q, float t) {   // Util.cpp:533-550
    switch (covMode) {
        case 0: return (q / t >= covThr) && (t / q >= covThr);
        case 1: return (q / t) >= covThr;     // COV_MODE_TARGET = 1, COV_MODE_QUERY = 2 (mm/commons/Parameters.h:246-251)
        case 2: return (t / q) >= covThr;
        case 3: return ((t / q) >= covThr) && (t / q) <= 1.0f;
        case 4: return ((q / t) >= covThr) && (q / t) <= 1.0f;
        case 5: return (fminf(t, q) / fmaxf(t, q)) >= covThr;
        default: return true;
    }
}
__device__ __forceinline__ bool hasCoverageDev(float covThr, int covMode, float qc, float tc) {  // Util.cpp:552-568
    switch (covMode) {
        case 0: return (qc >= covThr) && (tc >= covThr);
        case 1: return tc >= covThr;
        case 2: return qc >= covThr;
        default: return true;
    }
}
__device__ __forceinline__ float computeCovDev(unsigned s, unsigned e, unsigned len) {            // StripedSmithWaterman.cpp:1055-1057
    return (float) (min(len, max(s, e)) - min(s, e) + 1) / (float) len;
}
// complement of an ASCII nucleotide exactly as rescorediagonal.cpp:175-178 builds the reverse query:
// num2aa[reverse(aa2num[c])] with aa2num = NucleotideMatrix letter mapping (NucleotideMatrix.cpp:17-61)
__device__ __forceinline__ char nuclRevCompChar(char c) {
    switch (c & ~0x20) {
        case 'A': return 'T';
        case 'C': case 'M': case 'Y': case 'H': return 'G';
        case 'T': case 'U': case 'W': return 'A';
        case 'G': case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': return 'C';
        default: return 'X';
    }
}

constexpr int RS_BLOCK = 256;
// lanes per candidate pair: G = 1 (one thread per pair: short read overlaps, ~6 wave-instructions per pair)
// or G = 16 (long overlaps, queued by the first kernel)
constexpr uint32_t RS_SHORT_MAX = 768;   // min(qLen, tLen) handled by one thread (round 5: 512 -> 768 once the identity pairs were gone: 31.2 -> 30.7 ms; 256: 46, 2048: 30.9)

template <int G> __device__ __forceinline__ int groupReduceSumG(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}
__device__ __forceinline__ uint32_t loadU32Unaligned(const char *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }

// Scores ONE diagonal with a 16-lane group (8 residues per lane and step); returns (all lanes of the group)
// score/first/last/idCnt; valid=false if the diagonal does not intersect.  Mode 3 only.
struct DiagScore { bool valid; unsigned score; int first, last; unsigned diagLen; int idCnt; };

template <bool REV, int G>
__device__ __forceinline__ DiagScore scoreDiagonal(const char *__restrict__ q, unsigned qLen, const char *__restrict__ t,
                                                   unsigned tLen, int diagonal, const signed char *__restrict__ smat, const unsigned char *__restrict__ comp, int sl) {
    DiagScore r; r.valid = false; r.score = 0; r.first = -1; r.last = -1; r.diagLen = 0; r.idCnt = 0;
    const unsigned dist = (unsigned) abs(diagonal);
    unsigned qo, to, len;
    if (diagonal >= 0 && dist < qLen) { qo = dist; to = 0; len = min(tLen, qLen - dist); }
    else if (diagonal < 0 && dist < tLen) { qo = 0; to = dist; len = min(tLen - dist, qLen); }
    else return r;
    r.valid = true; r.diagLen = len;
    if (len == 0) { r.first = 0; r.last = -1; return r; }   // empty sequence: reference reads out of bounds; unsupported
    // REV: the aligned query is the reverse complement of the stored one: qrev[i] = comp(q[qLen-1-i])
    auto Q = [&](unsigned i) -> char { return REV ? (char) comp[(unsigned char) q[qLen - 1 - (qo + i)]] : q[qo + i]; };
    unsigned first, last;
    int s = 0, ids = 0;
    // Columns outside [first, last] are BLANKED (byte 0 in both sequences' words; entry [0][0] of the LDS score table and entry 0
    // of the complement table are 0), so that the lookups of a step are unconditional: the compiler issues them together and waits
    // once.  (Round 3: with one predicated block per column every `ds_read` was followed by its own wait.)
    if (G == 1) {
        // one thread per pair: 16 residues of both sequences per round trip (unaligned 16-byte loads; buffers are padded), the next
        // 16 requested before the current ones are scored.  A pair is a chain of dependent round trips per lane and the kernel is
        // bound by their number: the first 16 columns are requested together with the bytes of the last column, before `first` and
        // `last` are known (column 0 is blanked afterwards when it holds a '*').
        // REV: the aligned query is the reverse complement of the stored one — the 16 stored bytes that end at the mirrored position
        // are loaded and byte-reversed (near the start of the sequence byte by byte, never reading before the buffer); the
        // complement is looked up per column.
        auto fetchQ = [&](unsigned p, uint32_t *w) {
            if (!REV) __builtin_memcpy(w, q + qo + p, 16);
            else {
                const unsigned rem = qLen - (qo + p);            // stored residues left of (and including) the mirrored position
                if (rem >= 16) {
                    uint32_t v[4]; __builtin_memcpy(v, q + (qLen - 1 - (qo + p)) - 15, 16);
                    w[0] = __builtin_bswap32(v[3]); w[1] = __builtin_bswap32(v[2]); w[2] = __builtin_bswap32(v[1]); w[3] = __builtin_bswap32(v[0]);
                } else {
                    uint64_t lo = 0, hi = 0;
                    for (unsigned j = rem; j-- > 0;) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | (uint64_t) (unsigned char) q[qLen - 1 - (qo + p + j)]; }
                    w[0] = (uint32_t) lo; w[1] = (uint32_t) (lo >> 32); w[2] = (uint32_t) hi; w[3] = (uint32_t) (hi >> 32);
                }
            }
        };
        uint32_t qn[4], tn[4];
        __builtin_memcpy(tn, t + to, 16); fetchQ(0, qn);
        const char te = t[to + len - 1];
        const char qeStored = REV ? q[qLen - 1 - (qo + len - 1)] : q[qo + len - 1];
        const char qe = REV ? (char) comp[(unsigned char) qeStored] : qeStored;
        const char t0 = (char) (tn[0] & 0xFFu);
        const char q0 = REV ? (char) comp[qn[0] & 0xFFu] : (char) (qn[0] & 0xFFu);
        first = (q0 == '*' || t0 == '*') ? 1u : 0u;
        last = len - 1;
        if (last > 0 && (qe == '*' || te == '*')) last--;
        for (unsigned p = 0; p <= last; p += 16u) {
            uint32_t qw[4], tw[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { qw[k] = qn[k]; tw[k] = tn[k]; }
            if (p + 16u <= last) { __builtin_memcpy(tn, t + to + p + 16, 16); fetchQ(p + 16, qn); }
            const unsigned n = min(16u, last - p + 1);
            const unsigned skip = p ? 0u : first;                // column 0 of the overlap is not scored when it holds a '*'
            const unsigned blanks = (16u - n) + skip;
            if (blanks) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int nb = (int) n - 4 * k;
                    uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                    if (k == 0 && skip) m &= 0xFFFFFF00u;
                    qw[k] &= m; tw[k] &= m;
                }
            }
            if (!REV) {
                // identities of the 16 columns at once: bytes equal up to the case bit are zero bytes of (q ^ t) & 0xDF..; exact
                // zero-byte test; the blanked columns count as equal and are taken off again
                int z = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t x = (qw[k] ^ tw[k]) & 0xDFDFDFDFu;
                    z += __popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu));
                }
                ids += z - (int) blanks;
            }
#pragma unroll
            for (unsigned j = 0; j < 16; j++) {
                unsigned a = (qw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                if (REV) a = (unsigned) comp[a];
                const unsigned b = (tw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                s += (int) smat[(a << 7) | b];
                if (REV) ids += ((a & ~0x20u) == (b & ~0x20u)) ? 1 : 0;
            }
            if (REV) ids -= (int) blanks;
        }
    } else {
    const char q0 = Q(0), t0 = t[to], qe = Q(len - 1), te = t[to + len - 1];
    first = (q0 == '*' || t0 == '*') ? 1u : 0u;
    last = len - 1;
    if (last > 0 && (qe == '*' || te == '*')) last--;
    for (unsigned p = first + 8u * (unsigned) sl; p <= last; p += 8u * G) {
        // 8 consecutive residues of both sequences (unaligned 8-byte loads; the DB buffer is padded past its end)
        uint64_t tw; __builtin_memcpy(&tw, t + to + p, 8);
        uint64_t qw;
        if (REV) {
            const unsigned rem = qLen - (qo + p);            // stored residues left of (and including) this one
            if (rem >= 8) { uint64_t v; __builtin_memcpy(&v, q + (qLen - 1 - (qo + p)) - 7, 8); qw = __builtin_bswap64(v); }
            else { qw = 0; for (unsigned j = 0; j < rem; j++) qw |= (uint64_t) (unsigned char) q[qLen - 1 - (qo + p + j)] << (8 * j); }   // never read before the buffer
        }
        else __builtin_memcpy(&qw, q + qo + p, 8);
        const unsigned n = min(8u, last - p + 1);
        if (n < 8u) { const uint64_t m = (1ULL << (8 * n)) - 1ULL; qw &= m; tw &= m; }
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
            unsigned a = (unsigned) (qw >> (8 * j)) & 0xFFu;
            const unsigned b = (unsigned) (tw >> (8 * j)) & 0xFFu;
            if (REV) a = (unsigned) comp[a];
            s += (int) smat[(a << 7) | b];
            ids += ((a & ~0x20u) == (b & ~0x20u)) ? 1 : 0;
        }
        ids -= (int) (8u - n);
    }
    }
    if (G > 1) { s = groupReduceSumG<G>(s); ids = groupReduceSumG<G>(ids); }
    r.score = (unsigned) max(s, 0); r.first = (int) first; r.last = (int) last; r.idCnt = ids;
    return r;
}

// MODE (round 6): RescoreArgs::mode as a compile-time constant — the launch over a candidate list (0) does not carry the registers of the
// stub-finishing paths (1, 2) through its loop
template <int G, int WPE, int MODE = -1>
__global__ __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void rescoreKernel(RescoreArgs a) {
    const int mode = MODE >= 0 ? MODE : a.mode;
    __shared__ signed char smat[123 * 128];              // row stride 128: the index of a column is (a << 7) | b
    __shared__ unsigned char sComp[256];                 // reverse-strand hits: complement of a stored letter (getRevFragment's mapping)
    for (int i = threadIdx.x; i < 123 * 128; i += RS_BLOCK) smat[i] = (i & 127) < 123 ? a.mat[(i >> 7) * 123 + (i & 127)] : (signed char) 0;
    for (int i = threadIdx.x; i < 256; i += RS_BLOCK) sComp[i] = i ? (unsigned char) nuclRevCompChar((char) i) : (unsigned char) 0;
    __syncthreads();
    if (threadIdx.x == 0) smat[0] = 0;      // blanked columns (scoreDiagonal) look up [0][0]; no residue is byte 0
    __syncthreads();
    const int groupsPerBlock = RS_BLOCK / G;
    const int sl = threadIdx.x & (G - 1);
    const uint64_t stride = (uint64_t) gridDim.x * groupsPerBlock;
    unsigned long long accLocal = 0, ovLocal = 0;
    const uint64_t nWork = (G == 1) ? (mode == 2 ? (uint64_t) a.nQueryList : a.nHits) : (uint64_t) *a.longCount;
    const bool finish = mode != 0;                     // (wave-uniform)
    // G == 1: the candidate of the next round and its sequences' offsets / lengths are requested while the current pair is scored
    // (a pair is a chain of dependent round trips: candidate -> offsets and lengths -> residues; two of them leave the chain)
    struct Meta { uint64_t qOff, tOff; uint32_t qLen, tLen; };
    // (offset, length) of an entry come from ONE packed word: a random target costs one line of metadata instead of two
    auto loadMeta = [&](const CandHit &c) { Meta m; const uint64_t qv = a.q.offLen[c.query], tv = a.t.offLen[c.target]; m.qOff = qv >> 24; m.qLen = (uint32_t) qv & 0xFFFFFFu; m.tOff = tv >> 24; m.tLen = (uint32_t) tv & 0xFFFFFFu; return m; };
    uint64_t w = (uint64_t) blockIdx.x * groupsPerBlock + (threadIdx.x / G);
    CandHit hitNext; Meta metaNext; CandHit hitAfter;
    memset(&hitNext, 0, sizeof(hitNext)); memset(&hitAfter, 0, sizeof(hitAfter)); memset(&metaNext, 0, sizeof(metaNext));
    if (G == 1 && !finish) {
        if (w < nWork) { hitNext = a.hits[w]; metaNext = loadMeta(hitNext); }
        if (w + stride < nWork) hitAfter = a.hits[w + stride];
    }
    for (; w < nWork; w += stride) {
        uint64_t h = (G == 1) ? w : a.longList[w];
        CandHit hit; Meta me;
        if (G == 1 && mode == 2) {                          // the stub among the record slots of query queryList[w]
            const uint32_t qq = a.queryList[w];
            uint64_t j = a.qoff[qq]; const uint64_t j1 = a.qoff[qq + 1];
            while (j < j1 && a.out[j].btKind != ALN_SELF_PENDING) j++;
            if (j == j1) continue;
            h = j;
        }
        if (finish) {                                         // the pair behind a stub: score and diagonal of the candidate were stashed in it
            const AlnRec stub = a.out[h];
            if (stub.btKind != ALN_SELF_PENDING) continue;
            hit.query = stub.query; hit.target = stub.target; hit.prefScore = stub.rawScore; hit.diag16 = (uint32_t) stub.qStart;
            me = loadMeta(hit);
        } else if (G == 1) {
            hit = hitNext; me = metaNext;
            hitNext = hitAfter;
            if (w + stride < nWork) metaNext = loadMeta(hitNext);
            if (w + 2 * stride < nWork) hitAfter = a.hits[w + 2 * stride];
        } else { hit = a.hits[h]; me = loadMeta(hit); }
        const uint32_t qid = hit.query, tid = hit.target;
        const char *q = a.q.data + me.qOff;
        const unsigned qLen = me.qLen;
        const char *t = a.t.data + me.tOff;
        const unsigned tLen = me.tLen;
        if (G == 1 && !finish && a.lazySelf && qid == tid && (a.includeIdentity || a.sameDB)) {
            // an identity pair: accepted whatever it scores, read by few consumers — left as a stub for finishSelfAlns (common.hpp)
            AlnRec stub; memset(&stub, 0, sizeof(stub));
            stub.query = qid; stub.target = tid; stub.rawScore = hit.prefScore; stub.qStart = (int32_t) hit.diag16; stub.qLen = (int) qLen; stub.dbLen = (int) tLen;
            stub.accepted = 1; stub.btKind = ALN_SELF_PENDING;
            a.out[h] = stub;
            accLocal += 1;
            continue;
        }
        if (G == 1) {                                         // long overlap: 16 lanes will score it (one atomic per wavefront, not per pair)
            // (round 4: sending every SELF hit — a third of the list, and the lane the others wait for — to the 16-lane kernel as well
            //  doubled the stage, 45 -> 89 ms: its per-pair epilogue on 16 lanes costs more than the waiting; profiles/r04_ab_knobs.txt)
            // (nor do the self hits hold the other lanes up measurably: the self hits in a launch of their own and the rest in a second
            //  one — every lane of a wavefront on pairs of one kind — took 47.3 ms against 44.9 ms for the one launch; same file)
            const bool toLong = min(qLen, tLen) > a.shortMax;
            const unsigned long long m = __ballot(toLong);
            if (m) {
                const int leader = __ffsll((long long) m) - 1;
                unsigned long long basePos = 0;
                if (laneId() == leader) basePos = atomicAdd(a.longCount, (unsigned long long) __popcll(m));
                basePos = __shfl(basePos, leader, 64);
                if (toLong) a.longList[basePos + (unsigned long long) __popcll(m & ((1ULL << laneId()) - 1ULL))] = h;
            }
            if (toLong) continue;
        }
        const bool isReverse = a.reverseCapable && hit.prefScore < 0;
        const bool isIdentity = (qid == tid) && (a.includeIdentity || a.sameDB);
        AlnRec rec; memset(&rec, 0, sizeof(rec));
        rec.query = qid; rec.target = tid;
        bool accepted = false;
        if (canBeCoveredDev(a.covThr, a.covMode, (float) qLen, (float) tLen)) {
            // computeUngappedAlignment: best over all wraps (first the negative ones, then the positive ones; strictly better wins);
            // default LocalAlignment if none scores > 0.  A wrap whose diagonal misses the sequences scores 0 and is skipped.
            int bStart = -1, bEnd = -1, bDiag = 0, bIds = 0; unsigned bScore = 0, bDiagLen = 0, bDist = 0;
            const unsigned d16 = hit.diag16 & 0xFFFFu;
            const unsigned nNeg = 1 + tLen / 32768, nPos = 1 + qLen / 65536;
            // Every lane walks ITS OWN wraps that meet the sequences, in the same order as before (round 4, last GPU calls: SQ_THREAD_CYCLES_VALU
            // showed 16 of 64 lanes active per vector instruction of this kernel — with one loop over the wrap index for all lanes, the
            // lanes whose diagonal is the negative wrap scored in the first round and those with the positive wrap in the second, each
            // round with about half the wavefront masked; a pair of sequences below 32 768 residues has exactly one such wrap)
            const unsigned nWrap = nNeg + nPos;
            auto wrapDiag = [&](unsigned c) -> int { return c < nNeg ? (int) (d16 - (c + 1) * 65536u) : (int) ((c - nNeg) * 65536u + d16); };
            auto nextWrap = [&](unsigned c) -> unsigned {
                for (; c < nWrap; c++) { const int real = wrapDiag(c); const unsigned dist = (unsigned) abs(real); if (real >= 0 ? dist < qLen : dist < tLen) break; }
                return c;
            };
            for (unsigned c = nextWrap(0); c < nWrap; c = nextWrap(c + 1)) {
                const int real = wrapDiag(c);
                const unsigned dist = (unsigned) abs(real);
                DiagScore s = isReverse ? scoreDiagonal<true, G>(q, qLen, t, tLen, real, smat, sComp, sl) : scoreDiagonal<false, G>(q, qLen, t, tLen, real, smat, sComp, sl);
                if (s.score > bScore) { bScore = s.score; bStart = s.first; bEnd = s.last; bDiag = real; bDiagLen = s.diagLen; bDist = dist; bIds = s.idCnt; }
            }
            if (sl == 0) ovLocal += bDiagLen;
            // ---- group-uniform finish (rescorediagonal.cpp:251-314) ----
            const int distance = (int) bScore;
            const int bitScore = (int) (fma(a.lambda, (double) distance, -a.logK) / a.ln2 + 0.5);
            const int alnLen = (bEnd - bStart) + 1;
            int qS, qE, dS, dE;
            if (bDiag >= 0) { qS = bStart + (int) bDist; qE = bEnd + (int) bDist; dS = bStart; dE = bEnd; }
            else { qS = bStart; qE = bEnd; dS = bStart + (int) bDist; dE = bEnd + (int) bDist; }
            const uint32_t ms = (qLen < a.minScoreLen) ? a.minScore[qLen] : 0xFFFFFFFFu;
            const bool hasEvalue = (uint32_t) distance >= ms;
            // default alignment (no wrap scored > 0): the reference's identity loop then runs over the
            // single index -1 of both strings; for the identity pair both bytes are the same byte.
            int idCnt = bIds;
            if (bStart < 0) idCnt = isIdentity ? 1 : 0;
            float seqId = 0.0f;
            if (hasEvalue || isIdentity) {
                switch (a.seqIdMode) {                                                     // Util.cpp:588-598
                    case 0: seqId = (float) idCnt / (float) alnLen; break;
                    case 1: seqId = (float) idCnt / (float) min((int) qLen, (int) tLen); break;
                    case 2: seqId = (float) idCnt / (float) max((int) qLen, (int) tLen); break;
                    default: seqId = 0.0f;
                }
            }
            const float queryCov = computeCovDev((unsigned) qS, (unsigned) qE, qLen);
            const float targetCov = computeCovDev((unsigned) dS, (unsigned) dE, tLen);
            if (isReverse) { qS = (int) qLen - qS - 1; qE = (int) qLen - qE - 1; }
            const bool hasCov = hasCoverageDev(a.covThr, a.covMode, queryCov, targetCov);
            const bool hasSeqId = (double) seqId >= (double) (a.seqIdThr - FLT_EPSILON);
            const bool hasAlnLen = alnLen >= a.alnLenThr;
            accepted = isIdentity || (hasAlnLen && hasCov && hasSeqId && hasEvalue);
            rec.bitScore = bitScore; rec.rawScore = distance; rec.seqId = seqId;
            rec.qStart = qS; rec.qEnd = qE; rec.qLen = (int) qLen; rec.dbStart = dS; rec.dbEnd = dE; rec.dbLen = (int) tLen;
            rec.alnLen = alnLen; rec.reversed = isReverse ? 1 : 0;
        }
        rec.accepted = accepted ? 1 : 0; rec.btKind = 1;        // ungapped: the backtrace is one run of alnLen 'M'
        if (sl == 0) {
            a.out[h] = rec;
            accLocal += accepted ? 1 : 0;
        }
    }
    accLocal = waveReduceSumU64(accLocal); ovLocal = waveReduceSumU64(ovLocal);
    if (laneId() == 0) {
        if (accLocal) atomicAdd(&a.stats[0], accLocal);
        if (ovLocal) atomicAdd(&a.stats[1], ovLocal);
    }
}

// (Round 6, VERDICT r5 item 2: a PERSISTENT wavefront whose lanes refill — one 16-column step per loop trip for every lane with an open
//  diagonal, the lanes that are through waiting until T of them run the epilogue together and draw new pairs from a device-wide cursor — was
//  built, byte-identical on every test, and measured against this kernel on the 12 iterations of the 50 M-read chain
//  (profiles/r06_calls/call1_rescore_refill_sweep.txt): 36.3 ms at T = 32, 33.4 at 48, 32.1 at 64, 42.7 at 16, against 30.8 ms for the
//  lock-step kernel.  The time grows linearly with the number of refill events per 64 pairs (~2.4 ms each) while the stepping part does not
//  shrink with the fuller lanes: the kernel is bound by the round trips a pair consists of, not by vector issue — a refill exposes one per
//  event for T pairs where the lock-step kernel exposes one per 64 — and the lanes-per-instruction figure does not measure that.  Removed.)
// dense copy of a sparse list (host paths only since round 5): four lanes per 64-byte record, 16 bytes each, both sides coalesced
__global__ void acceptFlagsKernel(const AlnRec *__restrict__ recs, uint32_t *__restrict__ accept, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) accept[i] = recs[i].accepted ? 1u : 0u;
}
__global__ void compactAlnKernel(const uint4 *__restrict__ in, const uint32_t *__restrict__ accept,
                                 const uint64_t *__restrict__ pos, uint4 *__restrict__ out, uint64_t n) {
    static_assert(sizeof(AlnRec) == 64, "four 16-byte pieces per record");
    const uint64_t total = n * 4;
    for (uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t i = t >> 2;
        if (accept[i]) out[pos[i] * 4 + (t & 3)] = in[t];
    }
}
__global__ void gatherOffsetsKernel(const uint64_t *__restrict__ candQoff, const uint64_t *__restrict__ pos,
                                    uint64_t *__restrict__ alnQoff, uint64_t nQ) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= nQ; i += (uint64_t) gridDim.x * blockDim.x)
        alnQoff[i] = pos[candQoff[i]];
}
__global__ void markLengthsKernel(const uint32_t *__restrict__ len, uint32_t n, uint32_t *__restrict__ present, uint32_t cap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t l = len[i];
        if (l < cap) present[l] = 1;
    }
}


int denseAlnsCopy(plasship_ctx *ctx, const plasship_alns *a, DevBuf &qoffBuf, DevBuf &recsBuf, const uint64_t **qoff, const AlnRec **recs) {
    *qoff = a->d_qoff.as<uint64_t>(); *recs = a->d_recs.as<AlnRec>();
    if (!a->sparse) return PLASSHIP_OK;
    const uint64_t n = a->nSlots;
    DevBuf dAccept, dPos, dTmp;
    const size_t tmpBytes = exclusiveScanTmpBytes(n);
    if (dAccept.alloc(std::max<uint64_t>(n, 1) * 4) != hipSuccess || dPos.alloc((n + 1) * 8) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess ||
        qoffBuf.alloc((a->nQueries + 1) * 8) != hipSuccess || recsBuf.alloc(std::max<uint64_t>(a->nLines, 1) * sizeof(AlnRec)) != hipSuccess) {
        setError("alignment list: out of device memory for the dense copy"); return PLASSHIP_ERR_DEVICE;
    }
    if (n) hipLaunchKernelGGL(acceptFlagsKernel, dim3((unsigned) std::min<uint64_t>((n + 255) / 256, (uint64_t) ctx->numCU * 32)), dim3(256), 0, ctx->stream, a->d_recs.as<AlnRec>(), dAccept.as<uint32_t>(), n);
    if (exclusiveScanU32(ctx->stream, dAccept.as<uint32_t>(), dPos.as<uint64_t>(), n, dTmp.p, tmpBytes)) { (void) plasship::streamSync(ctx->stream); setError("scan failed"); return PLASSHIP_ERR_DEVICE; }   // (the local buffers go with the function: their kernel must be through)
    if (n) hipLaunchKernelGGL(compactAlnKernel, dim3((unsigned) std::min<uint64_t>((4 * n + 255) / 256, (uint64_t) ctx->numCU * 64)), dim3(256), 0, ctx->stream,
                              a->d_recs.as<uint4>(), dAccept.as<uint32_t>(), dPos.as<uint64_t>(), recsBuf.as<uint4>(), n);
    hipLaunchKernelGGL(gatherOffsetsKernel, dim3((unsigned) std::min<uint64_t>((a->nQueries + 256) / 256, 65535)), dim3(256), 0, ctx->stream,
                       a->d_qoff.as<uint64_t>(), dPos.as<uint64_t>(), qoffBuf.as<uint64_t>(), (uint64_t) a->nQueries);
    uint64_t nAcc = 0;
    PH_COPY_SYNC(ctx->stream, &nAcc, dPos.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost);
    if (nAcc != a->nLines) { setError("alignment list: the accepted records do not add up to the list's count"); return PLASSHIP_ERR_DEVICE; }
    *qoff = qoffBuf.as<uint64_t>(); *recs = recsBuf.as<AlnRec>();
    return PLASSHIP_OK;
}

// The identity pairs plasship_rescore left as stubs (common.hpp: plasship_alns::selfPending), scored by the same kernels with the list's own
// parameters.  An identity pair is accepted whatever it scores and its sequence identity does not depend on the E-value gate
// (rescorediagonal.cpp:262-297), so the per-length minimum-score table of plasship_rescore is not needed here (an empty table: "no E-value").
int finishSelfAlns(plasship_ctx *ctx, const plasship_alns *al, const uint32_t *dQueryList, uint32_t nQueryList) {
    if (!al->selfPending) return PLASSHIP_OK;
    if (!al->qdb || !al->tdb) { setError("alignment list: the DBs it was made from are gone (they must outlive the list)"); return PLASSHIP_ERR_ARG; }
    const bool nucl = al->nucl;
    const uint64_t n = al->nSlots;
    DevBuf dMat, dStats, dLongList, dLongCount;
    if (dMat.alloc(123 * 123) != hipSuccess || dStats.alloc(16) != hipSuccess || dLongList.alloc(((uint64_t) al->nQueries + 1) * 8) != hipSuccess || dLongCount.alloc(8) != hipSuccess) {
        setError("alignment list: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipMemsetAsync(dStats.p, 0, 16, ctx->stream));
    PH_CHECK(hipMemsetAsync(dLongCount.p, 0, 8, ctx->stream));
    PH_CHECK(hipMemcpyAsync(dMat.p, asciiSubMat(nucl), 123 * 123, hipMemcpyHostToDevice, ctx->stream));
    { int rcOL = ensureOffLen(ctx, al->qdb); if (!rcOL) rcOL = ensureOffLen(ctx, al->tdb); if (rcOL) return rcOL; }
    HostEvaluer ev(nucl, al->dbResidues);
    const plasship_rescore_params &par = al->rsPar;
    RescoreArgs a; memset(&a, 0, sizeof(a));
    a.q = al->qdb->view(); a.t = al->tdb->view(); a.qoff = al->d_qoff.as<uint64_t>(); a.hits = nullptr; a.nHits = n;
    a.out = al->d_recs.as<AlnRec>(); a.minScore = nullptr; a.minScoreLen = 0;
    a.mat = dMat.as<signed char>(); a.sameDB = al->rsSameDB; a.includeIdentity = par.include_identity; a.reverseCapable = al->rsReverseCapable;
    a.covMode = par.cov_mode; a.covThr = par.cov_thr; a.seqIdThr = par.seq_id_thr; a.alnLenThr = par.min_aln_len; a.seqIdMode = par.seq_id_mode;
    a.lambda = ev.g[0]; a.logK = ev.logK; a.ln2 = ev.ln2; a.stats = dStats.as<unsigned long long>();
    a.longList = dLongList.as<unsigned long long>(); a.longCount = dLongCount.as<unsigned long long>();
    a.shortMax = (uint32_t) tuneInt("RESCORE_SHORT", (int) RS_SHORT_MAX);
    a.mode = dQueryList ? 2 : 1; a.lazySelf = 0; a.queryList = dQueryList; a.nQueryList = nQueryList;
    const uint64_t work = dQueryList ? (uint64_t) nQueryList : n;
    if (work) {
        const unsigned grid = (unsigned) std::min<uint64_t>((work + 255) / 256 + 1, (uint64_t) ctx->numCU * 32);
        hipLaunchKernelGGL((rescoreKernel<1, 4>), dim3(grid), dim3(RS_BLOCK), 0, ctx->stream, a);
        hipLaunchKernelGGL((rescoreKernel<16, 6>), dim3((unsigned) ctx->numCU * 8), dim3(RS_BLOCK), 0, ctx->stream, a);
    }
    PH_CHECK(plasship::streamSync(ctx->stream));          // (the local buffers above are released with the function: their kernels must be through)
    PH_CHECK(hipGetLastError());
    if (!dQueryList) al->selfPending = false;
    return PLASSHIP_OK;
}

}  // namespace plasship
using namespace plasship;

extern "C" int plasship_rescore(plasship_ctx *ctx, const plasship_seqdb *qdb, const plasship_seqdb *tdb,
                                const plasship_cands *c, const plasship_rescore_params *par, plasship_alns **out,
                                plasship_rescore_stats *stats) {
    if (!ctx || !qdb || !tdb || !c || !par || !out) { setError("plasship_rescore: bad argument"); return PLASSHIP_ERR_ARG; }
    if (par->rescore_mode != 3) { setError("plasship_rescore: only --rescore-mode 3 (end-to-end) runs on the GPU path"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (c->nQueries != qdb->n) { setError("plasship_rescore: candidate list does not belong to the query DB"); return PLASSHIP_ERR_ARG; }
    if (qdb->dbtype != tdb->dbtype) { setError("plasship_rescore: query and target DB types differ"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    const bool nucl = qdb->dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES;
    const uint64_t nHits = c->nHits;
    const uint32_t maxQLen = qdb->maxEntryLen >= 2 ? qdb->maxEntryLen - 2 : 0;

    // E-value gate table: which query lengths exist (device) -> min passing score per length (host, exact doubles)
    DevBuf dPresent, dMinScore, dMat, dStats;
    const uint32_t tabLen = maxQLen + 1;
    if (dPresent.alloc((size_t) tabLen * 4) != hipSuccess || dMinScore.alloc((size_t) tabLen * 4) != hipSuccess ||
        dMat.alloc(123 * 123) != hipSuccess || dStats.alloc(16) != hipSuccess) { setError("plasship_rescore: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dPresent.p, 0, (size_t) tabLen * 4, ctx->stream));
    PH_CHECK(hipMemsetAsync(dStats.p, 0, 16, ctx->stream));
    if (qdb->n) hipLaunchKernelGGL(markLengthsKernel, dim3(std::min<size_t>(4096, (qdb->n + 255) / 256)), dim3(256), 0, ctx->stream,
                                   qdb->d_len.as<uint32_t>(), (uint32_t) qdb->n, dPresent.as<uint32_t>(), tabLen);
    std::vector<uint32_t> present(tabLen), minScore(tabLen, 0xFFFFFFFFu);
    PH_CHECK(hipMemcpyAsync(present.data(), dPresent.p, (size_t) tabLen * 4, hipMemcpyDeviceToHost, ctx->stream));
    PH_CHECK(plasship::streamSync(ctx->stream));
    HostEvaluer ev(nucl, tdb->residues);
    {
        // max raw score of an overlap of length L: max matrix entry (11 for BLOSUM62 W-W, 2 for nucl) * L
        const int maxEntry = nucl ? 2 : 11;
        int guess = -1;                                   // neighbouring lengths have neighbouring thresholds
        for (uint32_t l = 1; l < tabLen; l++)
            if (present[l]) { guess = ev.minScoreForEvalue(par->eval_thr, (int) l, maxEntry * (int) l + 1, guess); minScore[l] = (uint32_t) guess; }
    }
    PH_CHECK(hipMemcpyAsync(dMinScore.p, minScore.data(), (size_t) tabLen * 4, hipMemcpyHostToDevice, ctx->stream));
    PH_CHECK(hipMemcpyAsync(dMat.p, asciiSubMat(nucl), 123 * 123, hipMemcpyHostToDevice, ctx->stream));

    std::unique_ptr<plasship_alns> holder(new plasship_alns());     // released to the caller on success only
    plasship_alns *al = holder.get();
    al->nQueries = qdb->n; al->nucl = nucl; al->addBacktrace = par->add_backtrace != 0; al->dbResidues = tdb->residues;
    if (al->d_qoff.alloc((qdb->n + 1) * 8) != hipSuccess || al->d_recs.alloc(std::max<uint64_t>(nHits, 1) * sizeof(AlnRec)) != hipSuccess) {
        setError("plasship_rescore: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }

    { int rcOL = ensureOffLen(ctx, qdb); if (!rcOL) rcOL = ensureOffLen(ctx, tdb); if (rcOL) return rcOL; }
    RescoreArgs a;
    a.q = qdb->view(); a.t = tdb->view(); a.qoff = c->d_qoff.as<uint64_t>(); a.hits = c->d_hits.as<CandHit>(); a.nHits = nHits;
    a.out = al->d_recs.as<AlnRec>(); a.minScore = dMinScore.as<uint32_t>(); a.minScoreLen = tabLen;
    a.mat = dMat.as<signed char>(); a.sameDB = (qdb == tdb); a.includeIdentity = par->include_identity; a.reverseCapable = c->reverseCapable;
    a.covMode = par->cov_mode; a.covThr = par->cov_thr; a.seqIdThr = par->seq_id_thr; a.alnLenThr = par->min_aln_len; a.seqIdMode = par->seq_id_mode;
    a.lambda = ev.g[0]; a.logK = ev.logK; a.ln2 = ev.ln2; a.stats = dStats.as<unsigned long long>();
    DevBuf dLongList, dLongCount;
    if (dLongList.alloc(std::max<uint64_t>(nHits, 1) * 8) != hipSuccess || dLongCount.alloc(8) != hipSuccess) { setError("plasship_rescore: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dLongCount.p, 0, 8, ctx->stream));
    a.longList = dLongList.as<unsigned long long>(); a.longCount = dLongCount.as<unsigned long long>();
    a.shortMax = (uint32_t) tuneInt("RESCORE_SHORT", (int) RS_SHORT_MAX);
    a.mode = 0; a.lazySelf = tuneInt("LAZY_SELF", 1) == 1 ? 1 : 0;      // PLASSHIP_TUNE_LAZY_SELF=2: every identity pair scored here (rounds 1-4)
    const unsigned grid = (unsigned) std::min<uint64_t>((nHits + 255) / 256 + 1, (uint64_t) ctx->numCU * (uint64_t) tuneInt("RESCORE", nHits > 50000000ull ? 32 : 12));   // large lists: smaller shares per workgroup even out the tail (37.8 -> 35.9 ms at 250 M pairs)
    PH_CHECK(hipEventRecord(ctx->ev[0], ctx->stream));
    // wavefronts per SIMD of the thread-per-pair kernel (registers against chains in flight).  Round 5: 4 — with the stub and finishing paths
    // the kernel spills 88 bytes per lane at 5 (96 VGPRs) and nothing at 4 (125): 35.5 -> 31.2 ms per iteration at 50 M reads; 3: 33.2, 6: 43.9
    // (profiles/r05_ab_knobs.txt, calls 12-13)
    static const int wpe = [] { const int v = tuneInt("RESCORE_WPE", 4); if (v < 4 || v > 6) fprintf(stderr, "[plasship] PLASSHIP_TUNE_RESCORE_WPE=%d: only 4, 5 and 6 are built, using 4\n", v); return v; }();
    // (16 or 8 lanes per pair for EVERY pair, and a second thread-per-pair pass for the overlaps of 128-512 columns, were both
    // slower — 76 / 53 ms and 83 ms against 45 ms per iteration at 50 M reads: profiles/r03_ab_knobs.txt)
    if (wpe == 5) hipLaunchKernelGGL((rescoreKernel<1, 5, 0>), dim3(grid), dim3(RS_BLOCK), 0, ctx->stream, a);
    else if (wpe == 6) hipLaunchKernelGGL((rescoreKernel<1, 6, 0>), dim3(grid), dim3(RS_BLOCK), 0, ctx->stream, a);
    else hipLaunchKernelGGL((rescoreKernel<1, 4, 0>), dim3(grid), dim3(RS_BLOCK), 0, ctx->stream, a);
    hipLaunchKernelGGL((rescoreKernel<16, 6>), dim3((unsigned) ctx->numCU * 8), dim3(RS_BLOCK), 0, ctx->stream, a);     // long overlaps (count read on the device)
    PH_CHECK(hipEventRecord(ctx->ev[1], ctx->stream));
    // the list stays SPARSE (common.hpp: plasship_alns): record h belongs to candidate pair h, the CSR is the candidate list's
    PH_CHECK(hipMemcpyAsync(al->d_qoff.p, c->d_qoff.p, (qdb->n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    unsigned long long hs[2] = {0, 0};
    PH_CHECK(hipMemcpyAsync(hs, dStats.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PH_CHECK(plasship::streamSync(ctx->stream));
    PH_CHECK(hipGetLastError());
    const uint64_t nAcc = hs[0];
    al->nLines = nAcc; al->nSlots = nHits; al->sparse = true;
    al->qdb = qdb; al->tdb = tdb;
    // (stubs exist only where an identity pair can: the same DB on both sides, or --add-self-matches — ADVICE r5: a list without any made every
    //  later consumer walk all its slots once)
    al->selfPending = a.lazySelf != 0 && (a.sameDB || a.includeIdentity); al->rsPar = *par; al->rsSameDB = (qdb == tdb); al->rsReverseCapable = c->reverseCapable;
    if (stats) {
        stats->n_scored = nHits; stats->n_accepted = nAcc; stats->overlap_residues = hs[1];
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); stats->ms_kernel = ms;
    }
    *out = holder.release();
    return PLASSHIP_OK;
}
