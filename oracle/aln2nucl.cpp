// ORACLE (test infrastructure): proteinaln2nucl restated.
//   mm/util/proteinaln2nucl.cpp:13-204  (coordinates x3, start-codon shift, identity / score recount over the
//                                        backtrace, gapped nucleotide evaluer, bit score TRUNCATED, not rounded)
#include "oracle.hpp"
#include <cctype>
#include <cstdlib>
#include <cstring>

namespace oracle {

bool proteinaln2nucl(const DB &qNucl, const DB &tNucl, const DB &qAa, const DB &tAa, const DB &alnDb, const Params &par, DB &out, std::string &err) {
    if (qNucl.dbtype != DBTYPE_NUCLEOTIDES || tNucl.dbtype != DBTYPE_NUCLEOTIDES || qAa.dbtype != DBTYPE_AMINO_ACIDS || tAa.dbtype != DBTYPE_AMINO_ACIDS) {
        err = "Wrong query and target database input"; return false;                       // :46-52
    }
    const signed char *mat = asciiSubMat(true);
    bool ok = false;
    const Evaluer evaluer = Evaluer::nuclGapped(par.gapOpenNucl, par.gapExtendNucl, tNucl.aminoAcidDBSize(), &ok);   // :54-58
    if (!ok) { err = "oracle: no captured Gumbel parameters for these nucleotide gap penalties"; return false; }
    const int gapOpen = par.gapOpenNucl, gapExtend = par.gapExtendNucl;
    out = DB(); out.dbtype = DBTYPE_ALIGNMENT_RES;
    std::vector<char> buffer(1024 + 32768 * 4);
    for (size_t i = 0; i < alnDb.size(); i++) {                                            // :83-188
        const uint32_t alnKey = alnDb.key[i];
        const char *data = alnDb.entry(i);
        const size_t queryId = qNucl.getId(alnKey);
        const size_t aaQueryId = qAa.getId(alnKey);
        if (queryId == (size_t) -1 || aaQueryId == (size_t) -1) { err = "query key missing"; return false; }
        const char *nuclQuerySeq = qNucl.entry(queryId);
        const unsigned nuclQuerySeqLen = qNucl.seqLen(queryId);
        const bool qStartCodon = qAa.entry(aaQueryId)[0] == '*';
        std::string result;
        while (*data != '\0') {
            Result res = parseAlignmentRecord(data);
            while (*data != '\n') data++;
            data++;
            if (qStartCodon && res.qStartPos == 0) { err = "Alignment contains unalignable character"; return false; }
            if (res.backtrace.empty()) { err = "This module only supports database input with backtrace string"; return false; }
            const size_t targetId = tNucl.getId(res.dbKey), aaTargetId = tAa.getId(res.dbKey);
            if (targetId == (size_t) -1 || aaTargetId == (size_t) -1) { err = "target key missing"; return false; }
            const char *nuclTargetSeq = tNucl.entry(targetId);
            const unsigned nuclTargetSeqLen = tNucl.seqLen(targetId);
            const bool tStartCodon = tAa.entry(aaTargetId)[0] == '*';
            if (tStartCodon && res.dbStartPos == 0) { err = "Alignment contains unalignable character"; return false; }
            res.dbStartPos = res.dbStartPos * 3 + (tStartCodon ? -3 : 0);                   // :128-133
            res.dbEndPos = res.dbEndPos * 3 + 2 + (tStartCodon ? -3 : 0);
            res.dbLen = nuclTargetSeqLen;
            res.qStartPos = res.qStartPos * 3 + (qStartCodon ? -3 : 0);
            res.qEndPos = res.qEndPos * 3 + 2 + (qStartCodon ? -3 : 0);
            res.qLen = nuclQuerySeqLen;
            size_t idCnt = 0, alnLen = 0;
            int qPos = res.qStartPos, tPos = res.dbStartPos, score = 0;
            std::string newBacktrace;
            const std::string &bt = res.backtrace;
            for (size_t pos = 0; pos < bt.size(); pos++) {                                 // :141-176
                int cnt = 0;
                if (isdigit((unsigned char) bt[pos])) {
                    cnt += atoi(bt.c_str() + pos);
                    while (pos < bt.size() && isdigit((unsigned char) bt[pos])) pos++;
                }
                if (pos >= bt.size()) break;
                bool update = false;
                switch (bt[pos]) {
                    case 'M':
                        for (int b = 0; b < cnt * 3; b++) {
                            idCnt += (nuclQuerySeq[qPos] == nuclTargetSeq[tPos]);
                            score += mat[(int) nuclQuerySeq[qPos] * 123 + (int) nuclTargetSeq[tPos]];
                            tPos++; qPos++;
                        }
                        update = true; break;
                    case 'D': tPos += cnt * 3; score -= gapOpen + ((cnt - 1) * 3) * gapExtend; update = true; break;
                    case 'I': qPos += cnt * 3; score -= gapOpen + ((cnt - 1) * 3) * gapExtend; update = true; break;
                }
                if (update) { alnLen += (size_t) cnt * 3; newBacktrace += std::to_string(cnt * 3); newBacktrace.push_back(bt[pos]); }
            }
            res.score = (int) evaluer.bitScore(score);                                     // :177 (implicit double -> int)
            res.eval = evaluer.evalue(score, nuclQuerySeqLen);
            res.backtrace = newBacktrace;
            res.seqId = (float) idCnt / (float) alnLen;
            const size_t len = resultToBuffer(buffer.data(), res, true);
            result.append(buffer.data(), len);
        }
        out.add(alnKey, result.data(), result.size());
    }
    out.sortByKey();
    return true;
}

}  // namespace oracle
