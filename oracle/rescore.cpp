// ORACLE (test infrastructure): rescorediagonal restated (rows R1–R5 of SURVEY.md §8a).
//   mm/alignment/rescorediagonal.cpp:45-379      doRescorediagonal
//   mm/alignment/DistanceCalculator.h:93-175       computeUngappedAlignment / ungappedAlignmentByDiagonal
//   mm/alignment/DistanceCalculator.h:14-38,179-220  scoring loops (modes 1,2,3)
//   mm/alignment/Matcher.cpp:190-203,248-370       alignment record text <-> result_t
//   mm/commons/Util.cpp:278-307,533-598            fastSeqIdToBuffer, canBeCovered, hasCoverage, computeSeqId
// Supported rescore modes: 1 (substitution), 2 (local start/end), 3 (end-to-end; the plass/penguin
// default).  Mode 0 (Hamming, AVX2) and 4 (window quality) are never used by the workflows.
#include "oracle.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace oracle {

bool canBeCoveredPublic(float covThr, int covMode, float q, float t);

static inline int M(const signed char *m, char a, char b) { return m[(int) a * 123 + (int) b]; }

static LocalAlignment scoreSpan(const char *s1, const char *s2, unsigned length, const signed char *m, int mode) {
    LocalAlignment r;
    if (mode == 1) {                      // computeSubstitutionDistance (local, no coordinates)
        int max = 0, score = 0;
        for (unsigned p = 0; p < length; p++) {
            score += M(m, s1[p], s2[p]); score = (score < 0) ? 0 : score; max = (score > max) ? score : max;
        }
        r.score = (unsigned) max;         // startPos/endPos keep their -1 defaults
    } else if (mode == 2) {               // computeSubstitutionStartEndDistance
        int maxScore = 0, maxEnd = 0, maxStart = 0, minPos = -1, score = 0;
        for (unsigned p = 0; p < length; p++) {
            score += M(m, s1[p], s2[p]);
            const bool isMin = (score <= 0);
            score = isMin ? 0 : score; minPos = isMin ? (int) p : minPos;
            const bool isNewMax = (score > maxScore);
            maxEnd = isNewMax ? (int) p : maxEnd; maxStart = isNewMax ? minPos + 1 : maxStart;
            maxScore = isNewMax ? score : maxScore;
        }
        r.startPos = maxStart; r.endPos = maxEnd; r.score = (unsigned) maxScore;
    } else {                              // mode 3: computeGlobalSubstitutionStartEndDistance
        unsigned first = (s1[0] == '*' || s2[0] == '*') ? 1 : 0;
        unsigned last = length - 1;
        if (last > 0 && (s1[length - 1] == '*' || s2[length - 1] == '*')) last--;
        int64_t score = 0;
        for (unsigned p = first; p <= last; p++) score += M(m, s1[p], s2[p]);
        score = std::max(score, (int64_t) 0);
        r.startPos = (int) first; r.endPos = (int) last; r.score = (unsigned) (int) score;
    }
    return r;
}

LocalAlignment ungappedAlignmentByDiagonal(const char *q, unsigned qLen, const char *t, unsigned tLen,
                                           int diagonal, const signed char *m, int mode) {
    unsigned dist = (unsigned) std::abs(diagonal);
    LocalAlignment res; res.distToDiagonal = dist; res.diagonal = diagonal;
    if (diagonal >= 0 && dist < qLen) {
        unsigned minLen = std::min(tLen, qLen - dist);
        res.diagonalLen = minLen;
        LocalAlignment tmp = scoreSpan(q + dist, t, minLen, m, mode);
        res.score = tmp.score; res.startPos = tmp.startPos; res.endPos = tmp.endPos;
    } else if (diagonal < 0 && dist < tLen) {
        unsigned minLen = std::min(tLen - dist, qLen);
        res.diagonalLen = minLen;
        LocalAlignment tmp = scoreSpan(q, t + dist, minLen, m, mode);
        res.score = tmp.score; res.startPos = tmp.startPos; res.endPos = tmp.endPos;
    }
    return res;
}

LocalAlignment computeUngappedAlignment(const char *q, unsigned qLen, const char *t, unsigned tLen,
                                        uint16_t diagonal, const signed char *m, int mode) {
    LocalAlignment max;
    for (unsigned d = 1; d <= 1 + tLen / 32768; d++) {
        int real = (int) (-d * 65536u + diagonal);
        LocalAlignment tmp = ungappedAlignmentByDiagonal(q, qLen, t, tLen, real, m, mode);
        if (tmp.score > max.score) max = tmp;
    }
    for (unsigned d = 0; d <= qLen / 65536; d++) {
        int real = (int) (d * 65536u + diagonal);
        LocalAlignment tmp = ungappedAlignmentByDiagonal(q, qLen, t, tLen, real, m, mode);
        if (tmp.score > max.score) max = tmp;
    }
    return max;
}

char *fastSeqIdToBuffer(float seqId, char *buffer) {                      // Util.cpp:278-307
    if (seqId == 1.0) {
        // The reference returns the pointer ON the '\0' here (not past it, unlike the itoa path), so the
        // caller's `*(p-1) = '\t'` overwrites the last '0': identity 1.0 is printed as "1.00".
        memcpy(buffer, "1.000", 6);
        return buffer + 5;
    }
    *buffer++ = '0'; *buffer++ = '.';
    if (seqId < 0.10) *buffer++ = '0';
    if (seqId < 0.01) *buffer++ = '0';
    return i32toa((int) (seqId * 1000), buffer);
}

size_t resultToBuffer(char *buf, const Result &r, bool addBacktrace) {    // Matcher.cpp:323-370
    char *t = u32toa(r.dbKey, buf); *(t - 1) = '\t';
    t = i32toa(r.score, t); *(t - 1) = '\t';
    t = fastSeqIdToBuffer(r.seqId, t); *(t - 1) = '\t';
    t += sprintf(t, "%.3E", r.eval); t++; *(t - 1) = '\t';
    t = i32toa(r.qStartPos, t); *(t - 1) = '\t';
    t = i32toa(r.qEndPos, t); *(t - 1) = '\t';
    t = i32toa((int) r.qLen, t); *(t - 1) = '\t';
    t = i32toa(r.dbStartPos, t); *(t - 1) = '\t';
    t = i32toa(r.dbEndPos, t); *(t - 1) = '\t';
    t = i32toa((int) r.dbLen, t);
    if (addBacktrace) {
        *(t - 1) = '\t';
        memcpy(t, r.backtrace.data(), r.backtrace.size());
        t += r.backtrace.size() + 1;
    }
    *(t - 1) = '\n'; *t = '\0';
    return (size_t) (t - buf);
}

static float computeCov(unsigned s, unsigned e, unsigned len) {           // StripedSmithWaterman.cpp:1055-1057
    return (std::min(len, std::max(s, e)) - std::min(s, e) + 1) / (float) len;
}

static int atoiFast(const char *s) { int sg = 1; if (*s == '-') { sg = -1; s++; } int v = 0; while (*s >= '0' && *s <= '9') v = v * 10 + (*s++ - '0'); return sg * v; }

Result parseAlignmentRecord(const char *data) {                            // Matcher.cpp:248-320
    const char *e[16]; size_t cols = 0; const char *p = data;
    while (*p != '\n' && *p != '\0' && cols < 15) {
        while (*p == ' ' || *p == '\t') p++;
        e[cols++] = p;
        while (*p != ' ' && *p != '\t' && *p != '\n' && *p != '\0') p++;
    }
    e[cols] = p;
    Result r;
    r.dbKey = (uint32_t) strtoul(e[0], nullptr, 10);
    r.score = atoiFast(e[1]);
    r.seqId = (float) strtod(e[2], nullptr);
    r.eval = strtod(e[3], nullptr);
    r.qStartPos = atoiFast(e[4]); r.qEndPos = atoiFast(e[5]); r.qLen = (unsigned) atoiFast(e[6]);
    r.dbStartPos = atoiFast(e[7]); r.dbEndPos = atoiFast(e[8]); r.dbLen = (unsigned) atoiFast(e[9]);
    int aq = (r.qStartPos == -1) ? 0 : r.qStartPos, ad = (r.dbStartPos == -1) ? 0 : r.dbStartPos;
    r.qcov = computeCov((unsigned) aq, (unsigned) r.qEndPos, r.qLen);
    r.dbcov = computeCov((unsigned) ad, (unsigned) r.dbEndPos, r.dbLen);
    r.alnLength = (unsigned) (std::max(std::abs(r.qEndPos - aq), std::abs(r.dbEndPos - ad)) + 1);
    if (cols >= 11) { const char *b = e[10]; const char *q = b; while (*q != '\n' && *q != '\0' && *q != '\t' && *q != ' ') q++; r.backtrace.assign(b, (size_t) (q - b)); }
    return r;
}

void readAlignmentResults(std::vector<Result> &out, const char *data) {     // Matcher.cpp:190-199
    if (!data) return;
    while (*data != '\0') {
        out.push_back(parseAlignmentRecord(data));
        while (*data != '\n') data++;
        data++;
    }
}

static bool hasCoverage(float covThr, int covMode, float qCov, float tCov) { // Util.cpp:552-568
    switch (covMode) {
        case 0: return (qCov >= covThr) && (tCov >= covThr);
        case 1: return tCov >= covThr;        // COV_MODE_TARGET = 1, COV_MODE_QUERY = 2 (mm/commons/Parameters.h:246-251)
        case 2: return qCov >= covThr;
        default: return true;
    }
}
static float computeSeqId(int mode, int ids, int qLen, int tLen, int alnLen) {  // Util.cpp:588-598
    switch (mode) {
        case 0: return (float) ids / (float) alnLen;
        case 1: return (float) ids / (float) std::min(qLen, tLen);
        case 2: return (float) ids / (float) std::max(qLen, tLen);
    }
    return 0.0f;
}

DB rescorediagonal(const DB &qDb, const DB &tDb, bool sameQTDB, const DB &prefDb, const Params &par) {
    const bool nucl = qDb.dbtype == DBTYPE_NUCLEOTIDES;
    const bool reversePref = prefDb.dbtype == DBTYPE_PREFILTER_REV_RES;
    const signed char *mat = asciiSubMat(nucl);
    const unsigned char *a2n = aa2num(true, 0);
    static const char nucNum2aa[] = "ACTGX"; static const unsigned char nucRev[5] = {2, 3, 0, 1, 4};
    Evaluer evaluer(nucl, tDb.aminoAcidDBSize());
    DB out;
    out.dbtype = (par.rescoreMode >= 2) ? DBTYPE_ALIGNMENT_RES : prefDb.dbtype;
    // (OpenMP loop over the prefilter entries in the reference, rescorediagonal.cpp:133-135; one result string per entry,
    //  appended in entry order afterwards)
    std::vector<std::string> results(prefDb.size());
#pragma omp parallel num_threads(std::max(1, par.threads))
    {
    std::string resultBuffer, queryRev;
    std::vector<char> buffer(1024 + 32768 * 4);
#pragma omp for schedule(dynamic, 256)
    for (size_t id = 0; id < prefDb.size(); id++) {
        const char *data = prefDb.entry(id);
        const uint32_t queryKey = prefDb.key[id];
        const char *querySeq = nullptr; size_t queryId = (size_t) -1; int queryLen = -1;
        if (*data != '\0') {
            queryId = qDb.getId(queryKey);
            querySeq = qDb.entry(queryId);
            queryLen = (int) qDb.seqLen(queryId);
            if (reversePref) {                                              // :173-179
                queryRev.resize((size_t) queryLen);
                for (int pos = queryLen - 1; pos > -1; pos--)
                    queryRev[(size_t) ((queryLen - 1) - pos)] = nucNum2aa[nucRev[a2n[(unsigned char) querySeq[pos]]]];
            }
        }
        resultBuffer.clear();
        std::vector<Hit> hits = parsePrefilterHits(data);
        for (const Hit &hit : hits) {
            const char *qAln = querySeq; bool isReverse = false;
            if (reversePref && hit.prefScore < 0) { qAln = queryRev.data(); isReverse = true; }
            size_t targetId = tDb.getId(hit.seqId);
            const bool isIdentity = (queryId == targetId && (par.includeIdentity || sameQTDB));
            const char *targetSeq = tDb.entry(targetId);
            int dbLen = (int) tDb.seqLen(targetId);
            if (!canBeCoveredPublic(par.covThr, par.covMode, (float) queryLen, (float) dbLen)) continue;
            LocalAlignment aln = computeUngappedAlignment(qAln, (unsigned) queryLen, targetSeq, (unsigned) dbLen,
                                                          hit.diagonal, mat, par.rescoreMode);
            unsigned distanceToDiagonal = aln.distToDiagonal;
            int diagonalLen = (int) aln.diagonalLen, distance = (int) aln.score, diagonal = aln.diagonal;
            double seqId = 0, evalue = 0.0; int bitScore = 0, alnLen = 0;
            float targetCov = (float) diagonalLen / (float) dbLen;
            float queryCov = (float) diagonalLen / (float) queryLen;
            Result result;
            evalue = evaluer.evalue(distance, queryLen);
            bitScore = (int) (evaluer.bitScore(distance) + 0.5);
            if (par.rescoreMode >= 2) {
                alnLen = (aln.endPos - aln.startPos) + 1;
                int qS, qE, dS, dE;
                if (diagonal >= 0) { qS = aln.startPos + (int) distanceToDiagonal; qE = aln.endPos + (int) distanceToDiagonal; dS = aln.startPos; dE = aln.endPos; }
                else { qS = aln.startPos; qE = aln.endPos; dS = aln.startPos + (int) distanceToDiagonal; dE = aln.endPos + (int) distanceToDiagonal; }
                if (evalue <= par.evalThr || isIdentity) {
                    int idCnt = 0;
                    for (int i = qS; i <= qE; i++) {
                        char ql = qAln[i] & (char) ~0x20, tl = targetSeq[dS + (i - qS)] & (char) ~0x20;
                        idCnt += (ql == tl) ? 1 : 0;
                    }
                    seqId = computeSeqId(par.seqIdMode, idCnt, queryLen, dbLen, alnLen);
                }
                std::string bt;
                if (par.addBacktrace) { bt = std::to_string(alnLen); bt.push_back('M'); }
                queryCov = computeCov((unsigned) qS, (unsigned) qE, (unsigned) queryLen);
                targetCov = computeCov((unsigned) dS, (unsigned) dE, (unsigned) dbLen);
                if (isReverse) { qS = queryLen - qS - 1; qE = queryLen - qE - 1; }
                result.dbKey = hit.seqId; result.score = bitScore; result.qcov = queryCov; result.dbcov = targetCov;
                result.seqId = (float) seqId; result.eval = evalue; result.alnLength = (unsigned) alnLen;
                result.qStartPos = qS; result.qEndPos = qE; result.qLen = (unsigned) queryLen;
                result.dbStartPos = dS; result.dbEndPos = dE; result.dbLen = (unsigned) dbLen; result.backtrace = bt;
            }
            bool hasCov = hasCoverage(par.covThr, par.covMode, queryCov, targetCov);
            bool hasSeqId = seqId >= (par.seqIdThr - std::numeric_limits<float>::epsilon());
            bool hasEvalue = (evalue <= par.evalThr);
            bool hasAlnLen = (alnLen >= par.alnLenThr);
            if (isIdentity || (hasAlnLen && hasCov && hasSeqId && hasEvalue)) {
                if (par.rescoreMode >= 2) {
                    size_t len = resultToBuffer(buffer.data(), result, par.addBacktrace);
                    resultBuffer.append(buffer.data(), len);
                } else {
                    Hit h2; h2.seqId = hit.seqId; h2.prefScore = isReverse ? -bitScore : bitScore; h2.diagonal = (uint16_t) diagonal;
                    resultBuffer.append(buffer.data(), prefilterHitToBuffer(buffer.data(), h2));
                }
            }
        }
        results[id] = resultBuffer;
    }
    }   // omp parallel
    for (size_t id = 0; id < prefDb.size(); id++) out.add(prefDb.key[id], results[id].data(), results[id].size());
    out.sortByKey();
    return out;
}

}  // namespace oracle
