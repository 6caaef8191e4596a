// ORACLE (test infrastructure): cyclecheck restated.
//   src/assembler/cyclecheck.cpp:30-291   per nucleotide contig: k-mers (k = 22) of the front / middle / back third, k-mer matches
//                                          between the thirds on diagonals >= seqLen/3, hit rate of a diagonal band > 0.2 =>
//                                          the contig is circular (or terminally redundant) and is written out, cut at the
//                                          split diagonal with --chop-cycle.  Called after every nuclassembleresults of the
//                                          penguin workflows (data/nuclassemble.sh:19-61,132).
#include "oracle.hpp"
#include <algorithm>

namespace oracle {

bool cyclecheck(const DB &seqDb, const Params &par, bool chopCycle, DB &out, std::string &err) {
    if (seqDb.dbtype != DBTYPE_NUCLEOTIDES) { err = "Module cyclecheck only supports nucleotide input database"; return false; }   // :46-51
    const size_t kmerSize = 22;                                                            // setCycleCheckDefaults :25-28
    const unsigned char *map = aa2num(true, 5);
    struct kmerSeqPos { size_t kmer; unsigned int pos; };
    auto compareByKmer = [](const kmerSeqPos &a, const kmerSeqPos &b) {                   // :57-67
        if (a.kmer < b.kmer) return true;
        if (b.kmer < a.kmer) return false;
        return a.pos < b.pos;
    };
    uint64_t powers[32]; { uint64_t p = 1; for (int i = 0; i < 32; i++) { powers[i] = p; p *= 4; } }   // Indexer(alphabetSize - 1 = 4, k), Indexer.h:20-83
    out = DB(); out.dbtype = DBTYPE_NUCLEOTIDES;
    std::vector<kmerSeqPos> frontKmers, middleKmers, backKmers;
    std::vector<unsigned int> diagHits;
    std::vector<unsigned char> num;
    for (size_t id = 0; id < seqDb.size(); id++) {                                         // :94-282
        const char *nuclSeq = seqDb.entry(id);
        const unsigned int seqLen = seqDb.seqLen(id);
        if (seqLen >= par.maxSeqLen) continue;                                             // :100-106 (warning, skipped)
        num.resize(seqLen);
        for (unsigned int i = 0; i < seqLen; i++) num[i] = map[(unsigned char) nuclSeq[i]];  // Sequence::mapSequence
        frontKmers.clear(); middleKmers.clear(); backKmers.clear();
        const unsigned int thirdSeqLen = seqLen / 3;
        // Sequence::hasNextKmer / nextKmer (Sequence.h:98-113): currItPos starts at -1; the loop reads the position BEFORE
        // nextKmer() advances it (:118-119), so the third a k-mer lands in is decided by (unsigned) (position - 1) — the very
        // first k-mer (position 0, "-1" = 4294967295) lands in the back list
        int currItPos = -1;
        while ((currItPos + 1) + (int) kmerSize <= (int) seqLen) {
            const unsigned int pos = (unsigned int) currItPos;
            currItPos++;
            uint64_t kmerIdx = 0;
            for (size_t i = 0; i < kmerSize; i++) kmerIdx += (uint64_t) num[currItPos + i] * powers[i];
            const kmerSeqPos e = {(size_t) kmerIdx, (unsigned int) currItPos};
            if (pos < thirdSeqLen + 1) frontKmers.push_back(e);
            else if (pos < 2 * thirdSeqLen + 1) middleKmers.push_back(e);
            else backKmers.push_back(e);
        }
        std::sort(frontKmers.begin(), frontKmers.end(), compareByKmer);
        std::sort(middleKmers.begin(), middleKmers.end(), compareByKmer);
        std::sort(backKmers.begin(), backKmers.end(), compareByKmer);
        unsigned int nMatches = 0;
        diagHits.assign((size_t) 2 * thirdSeqLen + 1, 0);
        const int third = static_cast<int>(seqLen / 3);
        // a match between a k-mer at `lo` (earlier third) and the same k-mer at `hi` (later third) counts when its diagonal is >= L/3
        auto hit = [&](unsigned int hi, unsigned int lo) {
            const int diag = hi - lo;
            if (diag >= third) { diagHits[diag - seqLen / 3]++; nMatches++; }
        };
        const size_t nF = frontKmers.size(), nM = middleKmers.size(), nB = backKmers.size();
        // front against back and middle (:151-188): only the FIRST entry of a run of equal front k-mers (smallest position) is used,
        // against every back and every middle entry of that k-mer
        for (size_t f = 0, b = 0, m = 0; f < nF && (b < nB || m < nM); ) {
            const size_t key = frontKmers[f].kmer;
            const unsigned int fpos = frontKmers[f].pos;
            while (b < nB && backKmers[b].kmer < key) b++;
            while (m < nM && middleKmers[m].kmer < key) m++;
            for (; b < nB && backKmers[b].kmer == key; b++) hit(backKmers[b].pos, fpos);
            for (; m < nM && middleKmers[m].kmer == key; m++) hit(middleKmers[m].pos, fpos);
            do f++; while (f < nF && frontKmers[f].kmer == key);
        }
        // middle against back (:191-216): first middle entry of a k-mer against every back entry of it
        for (size_t m = 0, b = 0; m < nM && b < nB; ) {
            if (middleKmers[m].kmer < backKmers[b].kmer) m++;
            else if (middleKmers[m].kmer > backKmers[b].kmer) b++;
            else {
                const size_t key = middleKmers[m].kmer;
                const unsigned int mpos = middleKmers[m].pos;
                for (; b < nB && backKmers[b].kmer == key; b++) hit(backKmers[b].pos, mpos);
                while (m < nM && middleKmers[m].kmer == key) m++;
            }
        }
        const unsigned int kmermatches = nMatches;
        unsigned int splitDiagonal = 0;                                                    // :241-269 hit rate of diagonal bands
        if (kmermatches > 0) {
            for (unsigned int d = 0; d < 2 * thirdSeqLen; d++) {
                if (diagHits[d] != 0) {
                    const unsigned int diag = d + thirdSeqLen;
                    const unsigned int diaglen = seqLen - diag;
                    const unsigned int gapwindow = diaglen * 0.01;
                    const unsigned int lower = std::max(0, static_cast<int>(d - gapwindow));
                    const unsigned int upper = std::min(d + gapwindow, 2 * thirdSeqLen);
                    unsigned int diagbandHits = 0;
                    for (size_t i = lower; i <= upper; i++)
                        if (diagHits[i] <= diagHits[d]) diagbandHits += diagHits[i];
                    const float diagbandHitRate = static_cast<float>(diagbandHits) / (diaglen - kmerSize + 1);
                    if (diagbandHitRate > 0.2) { splitDiagonal = diag; break; }            // HIT_RATE_THRESHOLD (double)
                }
            }
        }
        if (splitDiagonal != 0) {                                                          // :271-283
            if (chopCycle) { std::string s(nuclSeq, splitDiagonal); s.push_back('\n'); out.add(seqDb.key[id], s.data(), s.size()); }
            else out.add(seqDb.key[id], nuclSeq, seqDb.elen[id] - 1);
        }
    }
    return true;
}

}  // namespace oracle
