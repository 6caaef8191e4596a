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
// Kernel design: ONE THREAD per candidate pair whenever the shorter sequence has at most 512 residues (rescoreKernel<1>: a read
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
    AlnRec *out;                 // [nHits]
    uint32_t *accept;            // [nHits]
    const uint32_t *minScore;    // [maxQLen+1] minimum raw score passing -e for that query length
    uint32_t minScoreLen;
    const signed char *mat;      // 123*123 in global, staged to LDS
    int sameDB, includeIdentity, reverseCapable;
    int covMode; float covThr;
    float seqIdThr; int alnLenThr, seqIdMode;
    double lambda, logK, ln2;
    unsigned long long *stats;   // [0] accepted, [1] overlap residues
    unsigned long long *longList, *longCount;   // hit indices queued for the 16-lane kernel
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
constexpr uint32_t RS_SHORT_MAX = 512;   // min(qLen, tLen) handled by one thread

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
    const char q0 = Q(0), t0 = t[to], qe = Q(len - 1), te = t[to + len - 1];
    const unsigned first = (q0 == '*' || t0 == '*') ? 1u : 0u;
    unsigned last = len - 1;
    if (last > 0 && (qe == '*' || te == '*')) last--;
    int s = 0, ids = 0;
    if (G == 1) {
        // one thread per pair: 16 residues of both sequences per round trip (unaligned 16-byte loads; buffers are padded).
        // The loop is bound by the number of memory requests, not by bytes: every lane streams its own two sequences.
        // REV: the aligned query is the reverse complement of the stored one — the 16 stored bytes that end at the mirrored
        // position are loaded and walked backwards (never reading before the start of the buffer).
        // (forward strand: the next 16 residues of both sequences are requested before the current ones are scored — the walk is a
        // chain of dependent round trips per lane, and the score lookups of one step hide most of the next step's latency)
        uint64_t qn[2] = {0, 0}, tn[2] = {0, 0};
        if (!REV && first <= last) { __builtin_memcpy(tn, t + to + first, 16); __builtin_memcpy(qn, q + qo + first, 16); }
        for (unsigned p = first; p <= last; p += 16u) {
            uint64_t qw[2], tw[2];
            if (!REV) { tw[0] = tn[0]; tw[1] = tn[1]; qw[0] = qn[0]; qw[1] = qn[1]; if (p + 16u <= last) { __builtin_memcpy(tn, t + to + p + 16, 16); __builtin_memcpy(qn, q + qo + p + 16, 16); } }
            else __builtin_memcpy(tw, t + to + p, 16);
            const unsigned n = min(16u, last - p + 1);
            bool wide = true;
            if (REV) {
                const unsigned rem = qLen - (qo + p);            // stored residues left of (and including) the mirrored position
                wide = rem >= 16;
                if (wide) __builtin_memcpy(qw, q + (qLen - 1 - (qo + p)) - 15, 16);
            }
            if (!REV) {
                // identities of the 16 columns at once: bytes equal up to the case bit are zero bytes of (q ^ t) & 0xDF..; exact
                // zero-byte test, columns beyond n masked off (4 instructions per residue less than comparing byte by byte)
                const uint64_t lo7 = 0x7F7F7F7F7F7F7F7FULL;
                const uint64_t x0 = (qw[0] ^ tw[0]) & 0xDFDFDFDFDFDFDFDFULL, x1 = (qw[1] ^ tw[1]) & 0xDFDFDFDFDFDFDFDFULL;
                uint64_t z0 = ~(((x0 & lo7) + lo7) | x0 | lo7), z1 = ~(((x1 & lo7) + lo7) | x1 | lo7);     // 0x80 in every zero byte
                if (n < 8) { z0 &= (1ULL << (8 * n)) - 1ULL; z1 = 0; } else if (n < 16) z1 &= (1ULL << (8 * (n - 8))) - 1ULL;
                ids += __popcll(z0) + __popcll(z1);
            }
#pragma unroll
            for (unsigned j = 0; j < 16; j++) {
                if (j < n) {
                    unsigned a;
                    if (!REV) a = (unsigned) (qw[j >> 3] >> (8 * (j & 7))) & 0xFFu;
                    else {
                        const unsigned jj = 15u - j;                 // byte of the loaded block that holds stored position mirror - j
                        const char c = wide ? (char) (qw[jj >> 3] >> (8 * (jj & 7))) : q[qLen - 1 - (qo + p + j)];
                        a = (unsigned) comp[(unsigned char) c];
                    }
                    const unsigned b = (unsigned) (tw[j >> 3] >> (8 * (j & 7))) & 0xFFu;
                    s += (int) smat[a * 123 + b];
                    if (REV) ids += ((a & ~0x20u) == (b & ~0x20u)) ? 1 : 0;
                }
            }
        }
    } else
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
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
            if (j < n) {
                char a = (char) (qw >> (8 * j)), b = (char) (tw >> (8 * j));
                if (REV) a = (char) comp[(unsigned char) a];
                s += (int) smat[(int) a * 123 + (int) b];
                ids += ((a & ~0x20) == (b & ~0x20)) ? 1 : 0;
            }
        }
    }
    if (G > 1) { s = groupReduceSumG<G>(s); ids = groupReduceSumG<G>(ids); }
    r.score = (unsigned) max(s, 0); r.first = (int) first; r.last = (int) last; r.idCnt = ids;
    return r;
}

template <int G>
__global__ __launch_bounds__(RS_BLOCK) void rescoreKernel(RescoreArgs a) {
    __shared__ signed char smat[123 * 123 + 7];
    __shared__ unsigned char sComp[256];                 // reverse-strand hits: complement of a stored letter (getRevFragment's mapping)
    for (int i = threadIdx.x; i < 123 * 123; i += RS_BLOCK) smat[i] = a.mat[i];
    for (int i = threadIdx.x; i < 256; i += RS_BLOCK) sComp[i] = (unsigned char) nuclRevCompChar((char) i);
    __syncthreads();
    const int groupsPerBlock = RS_BLOCK / G;
    const int sl = threadIdx.x & (G - 1);
    const uint64_t stride = (uint64_t) gridDim.x * groupsPerBlock;
    unsigned long long accLocal = 0, ovLocal = 0;
    const uint64_t nWork = (G == 1) ? a.nHits : (uint64_t) *a.longCount;
    for (uint64_t w = (uint64_t) blockIdx.x * groupsPerBlock + (threadIdx.x / G); w < nWork; w += stride) {
        const uint64_t h = (G == 1) ? w : a.longList[w];
        const CandHit hit = a.hits[h];
        const uint32_t qid = hit.query, tid = hit.target;
        const char *q = a.q.data + a.q.off[qid];
        const unsigned qLen = a.q.len[qid];
        const char *t = a.t.data + a.t.off[tid];
        const unsigned tLen = a.t.len[tid];
        if (G == 1 && min(qLen, tLen) > RS_SHORT_MAX) {       // long overlap: 16 lanes will score it
            const unsigned long long o = atomicAdd(a.longCount, 1ULL); a.longList[o] = h;
            continue;
        }
        const bool isReverse = a.reverseCapable && hit.prefScore < 0;
        const bool isIdentity = (qid == tid) && (a.includeIdentity || a.sameDB);
        AlnRec rec; memset(&rec, 0, sizeof(rec));
        rec.query = qid; rec.target = tid;
        bool accepted = false;
        if (canBeCoveredDev(a.covThr, a.covMode, (float) qLen, (float) tLen)) {
            // computeUngappedAlignment: best over all wraps; default LocalAlignment if none scores > 0
            int bStart = -1, bEnd = -1, bDiag = 0, bIds = 0; unsigned bScore = 0, bDiagLen = 0, bDist = 0;
            const unsigned d16 = hit.diag16 & 0xFFFFu;
            for (unsigned d = 1; d <= 1 + tLen / 32768; d++) {
                const int real = (int) (d16 - d * 65536u);
                DiagScore s = isReverse ? scoreDiagonal<true, G>(q, qLen, t, tLen, real, smat, sComp, sl) : scoreDiagonal<false, G>(q, qLen, t, tLen, real, smat, sComp, sl);
                if (s.score > bScore) { bScore = s.score; bStart = s.first; bEnd = s.last; bDiag = real; bDiagLen = s.diagLen; bDist = (unsigned) abs(real); bIds = s.idCnt; }
            }
            for (unsigned d = 0; d <= qLen / 65536; d++) {
                const int real = (int) (d * 65536u + d16);
                DiagScore s = isReverse ? scoreDiagonal<true, G>(q, qLen, t, tLen, real, smat, sComp, sl) : scoreDiagonal<false, G>(q, qLen, t, tLen, real, smat, sComp, sl);
                if (s.score > bScore) { bScore = s.score; bStart = s.first; bEnd = s.last; bDiag = real; bDiagLen = s.diagLen; bDist = (unsigned) abs(real); bIds = s.idCnt; }
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
            a.accept[h] = accepted ? 1u : 0u;
            accLocal += accepted ? 1 : 0;
        }
    }
    accLocal = waveReduceSumU64(accLocal); ovLocal = waveReduceSumU64(ovLocal);
    if (laneId() == 0) {
        if (accLocal) atomicAdd(&a.stats[0], accLocal);
        if (ovLocal) atomicAdd(&a.stats[1], ovLocal);
    }
}

__global__ void compactAlnKernel(const AlnRec *__restrict__ in, const uint32_t *__restrict__ accept,
                                 const uint64_t *__restrict__ pos, AlnRec *__restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        if (accept[i]) out[pos[i]] = in[i];
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

    DevBuf dAll, dAccept, dPos, dTmp;
    const size_t tmpBytes = exclusiveScanTmpBytes(nHits);
    if (dAll.alloc(std::max<uint64_t>(nHits, 1) * sizeof(AlnRec)) != hipSuccess || dAccept.alloc(std::max<uint64_t>(nHits, 1) * 4) != hipSuccess ||
        dPos.alloc((nHits + 1) * 8) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_rescore: out of device memory"); return PLASSHIP_ERR_DEVICE; }

    RescoreArgs a;
    a.q = qdb->view(); a.t = tdb->view(); a.qoff = c->d_qoff.as<uint64_t>(); a.hits = c->d_hits.as<CandHit>(); a.nHits = nHits;
    a.out = dAll.as<AlnRec>(); a.accept = dAccept.as<uint32_t>(); a.minScore = dMinScore.as<uint32_t>(); a.minScoreLen = tabLen;
    a.mat = dMat.as<signed char>(); a.sameDB = (qdb == tdb); a.includeIdentity = par->include_identity; a.reverseCapable = c->reverseCapable;
    a.covMode = par->cov_mode; a.covThr = par->cov_thr; a.seqIdThr = par->seq_id_thr; a.alnLenThr = par->min_aln_len; a.seqIdMode = par->seq_id_mode;
    a.lambda = ev.g[0]; a.logK = ev.logK; a.ln2 = ev.ln2; a.stats = dStats.as<unsigned long long>();
    DevBuf dLongList, dLongCount;
    if (dLongList.alloc(std::max<uint64_t>(nHits, 1) * 8) != hipSuccess || dLongCount.alloc(8) != hipSuccess) { setError("plasship_rescore: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dLongCount.p, 0, 8, ctx->stream));
    a.longList = dLongList.as<unsigned long long>(); a.longCount = dLongCount.as<unsigned long long>();
    const unsigned grid = (unsigned) std::min<uint64_t>((nHits + 255) / 256 + 1, (uint64_t) ctx->numCU * (uint64_t) tuneInt("RESCORE", nHits > 50000000ull ? 32 : 12));   // large lists: smaller shares per workgroup even out the tail (37.8 -> 35.9 ms at 250 M pairs)
    PH_CHECK(hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(rescoreKernel<1>, dim3(grid), dim3(RS_BLOCK), 0, ctx->stream, a);
    hipLaunchKernelGGL(rescoreKernel<16>, dim3((unsigned) ctx->numCU * 8), dim3(RS_BLOCK), 0, ctx->stream, a);     // long overlaps (count read on the device)
    PH_CHECK(hipEventRecord(ctx->ev[1], ctx->stream));
    if (exclusiveScanU32(ctx->stream, dAccept.as<uint32_t>(), dPos.as<uint64_t>(), nHits, dTmp.p, tmpBytes)) { setError("scan failed"); return PLASSHIP_ERR_DEVICE; }
    // the accepted alignments are compacted into a buffer sized for ALL pairs (nearly all candidates of an assembly iteration are
    // accepted): their number is read with the statistics at the end instead of costing a wait of its own here
    uint64_t nAcc = 0;
    PH_CHECK(hipMemcpyAsync(&nAcc, dPos.as<uint64_t>() + nHits, 8, hipMemcpyDeviceToHost, ctx->stream));

    std::unique_ptr<plasship_alns> holder(new plasship_alns());     // released to the caller on success only
    plasship_alns *al = holder.get();
    al->nQueries = qdb->n; al->nucl = nucl; al->addBacktrace = par->add_backtrace != 0; al->dbResidues = tdb->residues;
    if (al->d_qoff.alloc((qdb->n + 1) * 8) != hipSuccess || al->d_recs.alloc(std::max<uint64_t>(nHits, 1) * sizeof(AlnRec)) != hipSuccess) {
        setError("plasship_rescore: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    if (nHits) hipLaunchKernelGGL(compactAlnKernel, dim3((unsigned) std::min<uint64_t>((nHits + 255) / 256, 65535)), dim3(256), 0, ctx->stream,
                                  dAll.as<AlnRec>(), dAccept.as<uint32_t>(), dPos.as<uint64_t>(), al->d_recs.as<AlnRec>(), nHits);
    hipLaunchKernelGGL(gatherOffsetsKernel, dim3((unsigned) std::min<uint64_t>((qdb->n + 256) / 256, 65535)), dim3(256), 0, ctx->stream,
                       c->d_qoff.as<uint64_t>(), dPos.as<uint64_t>(), al->d_qoff.as<uint64_t>(), (uint64_t) qdb->n);
    unsigned long long hs[2] = {0, 0};
    PH_CHECK(hipMemcpyAsync(hs, dStats.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PH_CHECK(plasship::streamSync(ctx->stream));
    PH_CHECK(hipGetLastError());
    al->nLines = nAcc;
    al->qdb = qdb; al->tdb = tdb;
    if (stats) {
        stats->n_scored = nHits; stats->n_accepted = nAcc; stats->overlap_residues = hs[1];
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); stats->ms_kernel = ms;
    }
    *out = holder.release();
    return PLASSHIP_OK;
}
