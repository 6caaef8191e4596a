// ORACLE (test infrastructure): findassemblystart restated.
//   src/assembler/findassemblystart.cpp:12-24   findPosOfM: position of the first 'M' of a sequence, -1 if none
//   src/assembler/findassemblystart.cpp:35-176  per query with an 'M': the query and every non-self alignment vote on
//                                               "the residue before the M is a stop ('*')"; if >= 20 % of them agree, the
//                                               M column becomes the new start of all of them (maximum over all queries),
//                                               and the output sequence is "*" + the sequence from that column on
// The workflow runs it once, between the first kmermatcher/rescorediagonal of iteration 0 and a second pair on the
// corrected sequences (data/assemble.sh:110-141).
#include "oracle.hpp"

namespace oracle {

static int findPosOfM(const char *seq) {                                                   // :12-24
    for (int pos = 0; seq[pos] != '\0'; pos++)
        if (seq[pos] == 'M') return pos;
    return -1;
}

bool findassemblystart(const DB &seqDb, const DB &alnDb, DB &out, std::string &err) {
    struct PositionOfM { size_t id; int mPos; bool hasM; bool hasStopM; };               // :26-33
    std::vector<int> addStopAtPosition(seqDb.size(), -1);                                  // :50-51
    const float threshold = 0.2;                                                           // :53
    std::vector<PositionOfM> stopPositions;
    for (size_t id = 0; id < alnDb.size(); id++) {                                         // :67-140
        const uint32_t queryKey = alnDb.key[id];
        const size_t qId = seqDb.getId(queryKey);
        if (qId == (size_t) -1) { err = "query key missing"; return false; }
        const char *querySeqData = seqDb.entry(qId);
        const int queryPosOfM = findPosOfM(querySeqData);
        if (queryPosOfM == -1) continue;                                                   // :76-78
        bool hasStopMq = false;
        if (queryPosOfM > 0) hasStopMq = querySeqData[queryPosOfM - 1] == '*';
        stopPositions.clear();
        stopPositions.push_back({qId, queryPosOfM, true, hasStopMq});                      // :84
        const char *results = alnDb.entry(id);
        while (*results != '\0') {                                                         // :87-121
            const uint32_t key = (uint32_t) strtoul(results, nullptr, 10);
            const size_t edgeId = seqDb.getId(key);
            if (edgeId == (size_t) -1) { err = "target key missing"; return false; }
            if (edgeId == qId) { while (*results != '\n') results++; results++; continue; }
            {   // :95-99 at least the ten columns of a result without backtrace
                int columns = 0; const char *p = results;
                while (*p != '\n' && *p != '\0') { while (*p == '\t' || *p == ' ') p++; if (*p == '\n' || *p == '\0') break; columns++; while (*p != '\t' && *p != ' ' && *p != '\n' && *p != '\0') p++; }
                if (columns < 10) { err = "ERROR: Backtrace is missing for at result"; return false; }
            }
            const Result res = parseAlignmentRecord(results);
            while (*results != '\n') results++;
            results++;
            const char *dbSeqData = seqDb.entry(edgeId);
            int posOfM = -1; bool hasM = false, hasStopM = false;
            if (res.qStartPos >= queryPosOfM && queryPosOfM <= res.qEndPos) {              // :108 (as written in the reference)
                const int queryMoffset = queryPosOfM - res.qStartPos;
                const int dbMPos = res.dbStartPos + queryMoffset;
                posOfM = dbMPos;
                hasM = dbMPos >= 0 && (dbSeqData[dbMPos] == 'M');
                if (dbMPos > 0 && hasM) hasStopM = dbSeqData[dbMPos - 1] == '*';
            }
            stopPositions.push_back({edgeId, posOfM, hasM, hasStopM});                     // :120
        }
        int stopMCount = 0;
        for (const PositionOfM &p : stopPositions) stopMCount += p.hasStopM;               // :122-127
        if (stopPositions.size() > 1) {
            const float frequency = static_cast<float>(stopMCount) / static_cast<float>(stopPositions.size());
            if (frequency >= threshold)
                for (const PositionOfM &p : stopPositions)                                 // :131-139 atomic maximum
                    if (addStopAtPosition[p.id] < p.mPos) addStopAtPosition[p.id] = p.mPos;
        }
    }
    out = DB(); out.dbtype = DBTYPE_AMINO_ACIDS;                                           // :47
    for (size_t id = 0; id < seqDb.size(); id++) {                                         // :156-169
        const char *querySeqData = seqDb.entry(id);
        const int mPos = addStopAtPosition[id];
        if (mPos == -1) out.add(seqDb.key[id], querySeqData, seqDb.elen[id] - 1);
        else { std::string str("*"); str.append(querySeqData + mPos); out.add(seqDb.key[id], str.data(), str.size()); }
    }
    return true;
}

}  // namespace oracle
