// ORACLE (test infrastructure): assembleresults / nuclassembleresults restated (rows A1–A5).
//   src/assembler/assembleresult.cpp:19-57     comparator + selectFragmentToExtend
//   src/assembler/assembleresult.cpp:59-108    getRevFragment, updateAlignment
//   src/assembler/assembleresult.cpp:110-356   doassembly
//   src/assembler/nuclassembleresult.cpp:19-92,173-398  nucleotide variant (Bayesian comparator,
//                                                        length cap on both sides, seqId not rescaled)
// std::priority_queue is used exactly like the reference so that the non-strict nucleotide comparator
// replays libstdc++'s heap operations identically (SURVEY.md §7 "hard parts").
#include "oracle.hpp"
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <queue>

namespace oracle {

struct CompareResultByScore {                        // assembleresult.cpp:19-36
    bool operator()(const Result &r1, const Result &r2) const {
        if (r1.score < r2.score) return true;
        if (r2.score < r1.score) return false;
        if (r1.alnLength < r2.alnLength) return true;
        if (r2.alnLength < r1.alnLength) return false;
        if (r1.dbKey > r2.dbKey) return true;
        if (r2.dbKey > r1.dbKey) return false;
        return false;
    }
};

struct CompareNuclResultByScore {                    // nuclassembleresult.cpp:36-70
    bool operator()(const Result &r1, const Result &r2) const {
        unsigned mm1 = (unsigned) ((1 - r1.seqId) * r1.alnLength + 0.5);
        unsigned mm2 = (unsigned) ((1 - r2.seqId) * r2.alnLength + 0.5);
        unsigned alpha1 = mm1 + 1, alpha2 = mm2 + 1;
        unsigned beta1 = r1.alnLength - mm1 + 1, beta2 = r2.alnLength - mm2 + 1;
        double log_c = (std::lgamma(beta1 + beta2) + std::lgamma(alpha1 + beta1)) -
                       (std::lgamma(alpha1 + beta1 + beta2) + std::lgamma(beta1));
        double log_r = 0.0, p = 0.0;
        for (size_t idx = 0; idx < alpha2; idx++) {
            p += exp(log_r + log_c);
            log_r = log(alpha1 + idx) + log(beta2 + idx) - (log(idx + 1) + log(idx + alpha1 + beta1 + beta2)) + log_r;
        }
        if (p < 0.45) return true;
        if (p > 0.55) return false;
        if (r1.dbLen - r1.alnLength < r2.dbLen - r2.alnLength) return true;
        if (r1.dbLen - r1.alnLength > r2.dbLen - r2.alnLength) return false;
        return true;
    }
};

template <typename Q>
static Result selectFragmentToExtend(Q &alignments, unsigned queryKey) {   // assembleresult.cpp:40-57
    while (!alignments.empty()) {
        Result res = alignments.top();
        alignments.pop();
        const bool notRightStartAndLeftStart = !(res.dbStartPos == 0 && res.qStartPos == 0);
        const bool rightStart = res.dbStartPos == 0 && (res.dbEndPos != (int) res.dbLen - 1);
        const bool leftStart = res.qStartPos == 0 && (res.qEndPos != (int) res.qLen - 1);
        const bool isNotIdentity = (res.dbKey != queryKey);
        if ((rightStart || leftStart) && notRightStartAndLeftStart && isNotIdentity) return res;
    }
    Result none; none.dbKey = UINT_MAX;
    return none;
}

static std::string getRevFragment(const char *fragment, size_t fragLen) {   // assembleresult.cpp:59-68
    static const char num2aa[] = "ACTGX"; static const unsigned char rev[5] = {2, 3, 0, 1, 4};
    const unsigned char *a2n = aa2num(true, 0);
    std::string out(fragLen, 'N');
    for (int pos = (int) fragLen - 1; pos > -1; pos--) {
        char r = num2aa[rev[a2n[(unsigned char) fragment[pos]]]];
        out[(fragLen - 1) - (size_t) pos] = (r == 'X') ? 'N' : r;
    }
    return out;
}

static void updateAlignment(Result &a, const LocalAlignment &aln, const char *q, size_t qLen,
                            const char *t, size_t tLen) {                     // assembleresult.cpp:70-108
    int qS, qE, dS, dE;
    int diag = aln.diagonal, dist = std::max(std::abs(diag), 0);
    if (diag >= 0) { qS = aln.startPos + dist; qE = aln.endPos + dist; dS = aln.startPos; dE = aln.endPos; }
    else { qS = aln.startPos; qE = aln.endPos; dS = aln.startPos + dist; dE = aln.endPos + dist; }
    int idCnt = 0;
    for (int i = qS; i < qE; i++) idCnt += (q[i] == t[dS + (i - qS)]) ? 1 : 0;
    float seqId = (float) idCnt / ((float) qE - (float) qS);
    a.seqId = seqId; a.qLen = (unsigned) qLen; a.dbLen = (unsigned) tLen;
    a.alnLength = aln.diagonalLen;
    float scorePerCol = (float) aln.score / (float) (a.alnLength + 0.5);
    a.score = (int) (scorePerCol * 100);
    a.qStartPos = qS; a.qEndPos = qE; a.dbStartPos = dS; a.dbEndPos = dE;
}

template <bool NUCLVARIANT, typename Cmp>
static DB doAssembly(const DB &seqDb, const DB &alnDb, const Params &par) {
    const bool nucl = seqDb.dbtype == DBTYPE_NUCLEOTIDES;
    const signed char *mat = asciiSubMat(nucl);
    Evaluer evaluer(nucl, seqDb.aminoAcidDBSize());
    const size_t N = seqDb.size();
    std::vector<unsigned char> wasExtended(N, 0);
    auto mark = [&](size_t i, unsigned char bits) { __atomic_fetch_or(&wasExtended[i], bits, __ATOMIC_RELAXED); };   // shared between the threads, like the reference's array
    DB out; out.dbtype = seqDb.dbtype;
    std::vector<std::string> extended(N);          // contig of query id, if it was extended (added in id order below)
#pragma omp parallel num_threads(std::max(1, par.threads))
    {
    std::vector<char> useReverse(N, 0);            // per thread, as in the reference
    std::vector<Result> alignments, tmpAlignments;
#pragma omp for schedule(dynamic, 64)
    for (size_t id = 0; id < N; id++) {
        unsigned queryKey = seqDb.key[id];
        const char *querySeq = seqDb.entry(id);
        unsigned querySeqLen = seqDb.seqLen(id);
        std::string query(querySeq, querySeqLen);
        size_t alnId = alnDb.getId(queryKey);
        alignments.clear();
        if (alnId != (size_t) -1) readAlignmentResults(alignments, alnDb.entry(alnId));
        bool queryCouldBeExtended = false;
        std::priority_queue<Result, std::vector<Result>, Cmp> alnQueue;
        for (size_t i = 0; i < alignments.size(); i++) {                        // :161-189 / nucl :196-224
            Result &a = alignments[i];
            int rawScore = (int) (evaluer.rawFromBit(a.score) + 0.5);
            float scorePerCol = (float) rawScore / (float) (a.alnLength + 0.5);
            if (!NUCLVARIANT) {
                float alnLen = (float) a.alnLength;
                float ids = (float) a.seqId * alnLen;
                a.seqId = ids / (alnLen + 0.5);
            }
            a.score = (int) (scorePerCol * 100);
            if (nucl) {
                size_t tid = seqDb.getId(a.dbKey);
                if (a.qStartPos > a.qEndPos) {
                    useReverse[tid] = 1;
                    std::swap(a.qStartPos, a.qEndPos);
                    unsigned dbStartPos = (unsigned) a.dbStartPos;
                    a.dbStartPos = (int) (a.dbLen - (unsigned) a.dbEndPos - 1);
                    a.dbEndPos = (int) (a.dbLen - dbStartPos - 1);
                } else useReverse[tid] = 0;
            }
            alnQueue.push(a);
            if (alignments.size() > 1) mark(seqDb.getId(a.dbKey), 0x40);
        }
        tmpAlignments.clear();
        while (!alnQueue.empty()) {
            unsigned leftQueryOffset = 0, rightQueryOffset = 0;
            tmpAlignments.clear();
            Result best;
            while ((best = selectFragmentToExtend(alnQueue, queryKey)).dbKey != UINT_MAX) {
                size_t targetId = seqDb.getId(best.dbKey);
                const char *targetSeq = seqDb.entry(targetId);
                unsigned targetSeqLen = seqDb.seqLen(targetId);
                if (best.dbStartPos == 0) {
                    if ((targetSeqLen - ((unsigned) best.dbEndPos + 1)) <= rightQueryOffset) continue;
                } else if (best.qStartPos == 0) {
                    if (best.dbStartPos <= (int) leftQueryOffset) continue;
                }
                mark(targetId, 0x10);
                unsigned dbStartPos = (unsigned) best.dbStartPos, dbEndPos = (unsigned) best.dbEndPos;
                unsigned qStartPos = (unsigned) best.qStartPos, qEndPos = (unsigned) best.qEndPos;
                if (dbStartPos == 0 && qEndPos == (querySeqLen - 1)) {             // right extension
                    if (rightQueryOffset > 0) { tmpAlignments.push_back(best); continue; }
                    unsigned fragLen = targetSeqLen - (dbEndPos + 1);
                    if (NUCLVARIANT && query.size() + fragLen >= par.maxSeqLen) break;   // nucl :271-275
                    std::string fragment = useReverse[targetId] ? getRevFragment(targetSeq, fragLen)
                                                                : std::string(targetSeq + dbEndPos + 1, fragLen);
                    query += fragment; rightQueryOffset += fragLen;
                    mark(targetId, 0x80);
                } else if (qStartPos == 0 && dbEndPos == (targetSeqLen - 1)) {     // left extension
                    if (leftQueryOffset > 0) { tmpAlignments.push_back(best); continue; }
                    unsigned fragLen = dbStartPos;
                    if (query.size() + fragLen >= par.maxSeqLen) break;            // :259-263
                    std::string fragment = useReverse[targetId] ? getRevFragment(targetSeq + (targetSeqLen - dbStartPos), fragLen)
                                                                : std::string(targetSeq, fragLen);
                    query = fragment + query; leftQueryOffset += fragLen;
                    mark(targetId, 0x80);
                }
            }
            if (leftQueryOffset > 0 || rightQueryOffset > 0) queryCouldBeExtended = true;
            if (!alnQueue.empty()) break;
            querySeqLen = (unsigned) query.length();
            const char *qs = query.c_str();
            for (size_t i = 0; i < tmpAlignments.size(); i++) {                    // :292-313
                size_t tId = seqDb.getId(tmpAlignments[i].dbKey);
                unsigned tSeqLen = seqDb.seqLen(tId);
                const char *tSeq = seqDb.entry(tId);
                std::string revHolder;
                if (useReverse[tId]) { revHolder = getRevFragment(tSeq, tSeqLen); tSeq = revHolder.c_str(); }
                int qStartPos = tmpAlignments[i].qStartPos, dbStartPos = tmpAlignments[i].dbStartPos;
                int diag = (int) ((unsigned) qStartPos + leftQueryOffset) - dbStartPos;
                LocalAlignment aln = ungappedAlignmentByDiagonal(qs, querySeqLen, tSeq, tSeqLen, diag, mat, par.rescoreMode);
                updateAlignment(tmpAlignments[i], aln, qs, querySeqLen, tSeq, tSeqLen);
                if (tmpAlignments[i].seqId >= par.seqIdThr) alnQueue.push(tmpAlignments[i]);
            }
        }
        if (queryCouldBeExtended) {
            query.push_back('\n');
            mark(id, 0x20);
            extended[id] = std::move(query);
        }
    }
    }   // omp parallel
    for (size_t id = 0; id < N; id++)
        if (wasExtended[id] & 0x20) out.add(seqDb.key[id], extended[id].data(), extended[id].size());
    for (size_t id = 0; id < N; id++) {                                            // :326-342
        bool isNotContig = !(wasExtended[id] & 0x20), wasNotExtended = !(wasExtended[id] & 0x80);
        if (isNotContig && (par.keepTarget || wasNotExtended))
            out.add(seqDb.key[id], seqDb.entry(id), seqDb.elen[id] - 1);
    }
    out.sortByKey();
    return out;
}

DB assembleresults(const DB &seqDb, const DB &alnDb, const Params &par) {
    return doAssembly<false, CompareResultByScore>(seqDb, alnDb, par);
}
DB nuclassembleresults(const DB &seqDb, const DB &alnDb, const Params &par) {
    return doAssembly<true, CompareNuclResultByScore>(seqDb, alnDb, par);
}

// ---- guidedassembleresults (src/assembler/guidedassembleresult.cpp:136-385) ---------------------------------------
// Nucleotide ORFs are extended exactly like nuclassembleresults does (same Bayesian comparator :23-75, no reverse
// strand, no score / seqId rescaling of the parsed hits), and every extension is mirrored on the protein twin; an
// extension never crosses a '*' (:183-184,234-244).
bool guidedassembleresults(const DB &nuclDb, const DB &aaDb, const DB &alnDb, const Params &par, DB &outNucl, DB &outAa, std::string &err) {
    const signed char *mat = asciiSubMat(true);                                            // :152-153
    const size_t N = nuclDb.size();
    if (aaDb.size() != N) { err = "guidedassembleresults: nucleotide and protein DB differ in size"; return false; }
    std::vector<unsigned char> wasExtended(N, 0);
    outNucl = DB(); outNucl.dbtype = nuclDb.dbtype; outAa = DB(); outAa.dbtype = aaDb.dbtype;
    std::vector<Result> nuclAlignments, tmp;
    for (size_t id = 0; id < N; id++) {
        const unsigned queryKey = nuclDb.key[id];
        unsigned nuclQuerySeqLen = nuclDb.seqLen(id);
        const size_t aaQueryId = aaDb.getId(queryKey);
        if (aaQueryId == (size_t) -1) { err = "guidedassembleresults: protein twin missing"; return false; }
        std::string nuclQuery(nuclDb.entry(id), nuclQuerySeqLen);
        const unsigned aaQuerySeqLen = aaDb.seqLen(aaQueryId);
        std::string aaQuery(aaDb.entry(aaQueryId), aaQuerySeqLen);
        const bool excludeLeftExtension = (aaQuery[0] == '*');                             // :183-184 (reads [0] of an empty string like the reference)
        const bool excludeRightExtension = (aaQuery[aaQuerySeqLen - 1] == '*');
        nuclAlignments.clear();
        const size_t alnId = alnDb.getId(queryKey);
        if (alnId != (size_t) -1) readAlignmentResults(nuclAlignments, alnDb.entry(alnId));
        bool queryCouldBeExtended = false;
        std::priority_queue<Result, std::vector<Result>, CompareNuclResultByScore> alnQueue;
        for (size_t i = 0; i < nuclAlignments.size(); i++) {                               // :194-205
            if (nuclAlignments[i].seqId < par.seqIdThr) continue;
            alnQueue.push(nuclAlignments[i]);
            if (nuclAlignments.size() > 1) wasExtended[nuclDb.getId(nuclAlignments[i].dbKey)] |= 0x40;
        }
        while (!alnQueue.empty()) {
            unsigned leftOff = 0, rightOff = 0;
            tmp.clear();
            Result best;
            while ((best = selectFragmentToExtend(alnQueue, queryKey)).dbKey != UINT_MAX) {
                const size_t tId = nuclDb.getId(best.dbKey), aaTId = aaDb.getId(best.dbKey);
                if (tId == (size_t) -1 || aaTId == (size_t) -1) { err = "guidedassembleresults: target missing"; return false; }
                const char *nuclTargetSeq = nuclDb.entry(tId);
                const unsigned nuclTargetSeqLen = nuclDb.seqLen(tId);
                const char *aaTargetSeq = aaDb.entry(aaTId);
                const unsigned aaTargetSeqLen = aaDb.seqLen(aaTId);
                if (best.dbStartPos == 0) {                                                // :234-244
                    if (((nuclTargetSeqLen - ((unsigned) best.dbEndPos + 1)) <= rightOff) || excludeRightExtension || aaTargetSeq[0] == '*') continue;
                } else if (best.qStartPos == 0) {
                    if ((best.dbStartPos <= (int) leftOff) || excludeLeftExtension || aaTargetSeq[aaTargetSeqLen - 1] == '*') continue;
                }
                wasExtended[tId] |= 0x10;
                const int nuclDbStartPos = best.dbStartPos, nuclDbEndPos = best.dbEndPos, qStartPos = best.qStartPos, qEndPos = best.qEndPos;
                if (nuclDbStartPos == 0 && qEndPos == ((int) nuclQuerySeqLen - 1)) {        // right extension :251-275
                    if (rightOff > 0) { tmp.push_back(best); continue; }
                    const unsigned nuclDbFragLen = (nuclTargetSeqLen - (unsigned) nuclDbEndPos) - 1;
                    const unsigned aaDbFragLen = (nuclTargetSeqLen / 3 - (unsigned) (nuclDbEndPos / 3)) - 1;
                    if (nuclQuery.size() + nuclDbFragLen >= par.maxSeqLen) break;
                    nuclQuery += std::string(nuclTargetSeq + nuclDbEndPos + 1, nuclDbFragLen);
                    aaQuery += std::string(aaTargetSeq + nuclDbEndPos / 3 + 1, aaDbFragLen);
                    rightOff += nuclDbFragLen;
                    wasExtended[tId] |= 0x80;
                } else if (qStartPos == 0 && nuclDbEndPos == ((int) nuclTargetSeqLen - 1)) { // left extension :276-301
                    if (leftOff > 0) { tmp.push_back(best); continue; }
                    const unsigned nuclDbFragLen = (unsigned) nuclDbStartPos;
                    if (nuclQuery.size() + nuclDbFragLen >= par.maxSeqLen) break;
                    const int hasStart = (aaTargetSeq[0] == '*') ? 1 : 0;
                    nuclQuery = std::string(nuclTargetSeq, nuclDbFragLen) + nuclQuery;
                    aaQuery = std::string(aaTargetSeq, nuclDbFragLen / 3 + hasStart) + aaQuery;
                    leftOff += nuclDbFragLen;
                    wasExtended[tId] |= 0x80;
                }
            }
            if (leftOff > 0 || rightOff > 0) queryCouldBeExtended = true;
            if (!alnQueue.empty()) break;
            nuclQuerySeqLen = (unsigned) nuclQuery.length();
            const char *qs = nuclQuery.c_str();
            for (size_t i = 0; i < tmp.size(); i++) {                                       // :315-334
                const size_t tId = nuclDb.getId(tmp[i].dbKey);
                const unsigned tSeqLen = nuclDb.seqLen(tId);
                const char *tSeq = nuclDb.entry(tId);
                const int diag = (int) ((unsigned) tmp[i].qStartPos + leftOff) - tmp[i].dbStartPos;
                LocalAlignment aln = ungappedAlignmentByDiagonal(qs, nuclQuerySeqLen, tSeq, tSeqLen, diag, mat, par.rescoreMode);
                updateAlignment(tmp[i], aln, qs, nuclQuerySeqLen, tSeq, tSeqLen);
                if (tmp[i].seqId >= par.seqIdThr) alnQueue.push(tmp[i]);
            }
        }
        if (queryCouldBeExtended) {                                                        // :336-342
            nuclQuery.push_back('\n'); aaQuery.push_back('\n');
            wasExtended[id] |= 0x20;
            outNucl.add(queryKey, nuclQuery.data(), nuclQuery.size());
            outAa.add(queryKey, aaQuery.data(), aaQuery.size());
        }
    }
    for (size_t id = 0; id < N; id++) {                                                     // :346-367 (protein entry taken BY THE SAME id)
        const bool isNotContig = !(wasExtended[id] & 0x20), wasNotExtended = !(wasExtended[id] & 0x80);
        if (isNotContig && (par.keepTarget || wasNotExtended)) {
            outNucl.add(nuclDb.key[id], nuclDb.entry(id), nuclDb.elen[id] - 1);
            outAa.add(aaDb.key[id], aaDb.entry(id), aaDb.elen[id] - 1);
        }
    }
    outNucl.sortByKey(); outAa.sortByKey();
    return true;
}

}  // namespace oracle
