// ORACLE (test infrastructure): kmermatcher restated (rows K1–K8 of SURVEY.md §8a).
//   mm/linclust/kmermatcher.cpp:77-385   fillKmerPositionArray  (extraction, hashing, selection)
//   mm/linclust/kmermatcher.cpp:387-448  doComputation          (sort, assignGroup, sort)
//   mm/linclust/kmermatcher.cpp:450-559  assignGroup
//   mm/linclust/kmermatcher.cpp:809-924  writeKmerMatcherResult (threads = 1)
//   mm/linclust/kmermatcher.cpp:705-724  self-only back-fill
//   mm/linclust/kmermatcher.h:10-130     record layouts + comparators
// Single split only (K9 is out of scope: SURVEY.md §8a).  The in-place array of the reference is
// emulated exactly, including what is left behind the compaction point, because
// writeKmerMatcherResult scans into it.
#include "oracle.hpp"
#include <omp.h>
#include <parallel/algorithm>
#include <algorithm>
#include <climits>
#include <cstring>

namespace oracle {

static const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                      P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// XXH64 of exactly 8 little-endian bytes (xxhash 0.8.0: XXH64_endian_align len<32 path +
// XXH64_finalize one 8-byte lane + XXH64_avalanche); call site kmermatcher.cpp:33-38.
uint64_t xxh64_u64(uint64_t v, uint64_t seed) {
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = rotl64(v * P2, 31) * P1;          // XXH64_round(0, v)
    h ^= k1;
    h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// 2-bit alphabet A0 C1 T2 G3: complement = code ^ 2; reverse the k letters (Util.cpp:601-638).
uint64_t revComplement(uint64_t kmer, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; i++) { r = (r << 2) | ((kmer & 3) ^ 2); kmer >>= 2; }
    return r;
}

#define BIT63 (1ULL << 63)

struct SeqPos { uint16_t score; uint64_t kmer; uint32_t pos; };   // kmermatcher.h:10-47

template <typename T> struct KPos { uint64_t kmer; uint32_t id; T seqLen; T pos; };   // kmermatcher.h:49-55

template <typename T, bool NUCL>
static bool cmpKmerLenIdPos(const KPos<T> &a, const KPos<T> &b) {       // kmermatcher.h:56-96
    uint64_t ak = NUCL ? (a.kmer | BIT63) : a.kmer, bk = NUCL ? (b.kmer | BIT63) : b.kmer;
    if (ak != bk) return ak < bk;
    if (a.seqLen != b.seqLen) return a.seqLen > b.seqLen;
    if (a.id != b.id) return a.id < b.id;
    if (a.pos != b.pos) return a.pos < b.pos;
    // reference comparator ties here (only possible for NUCL records of ONE sequence that differ in nothing but the strand bit:
    // an inverted repeat whose two copies mirror each other's position); ips4o's order is unspecified => reverse strand (bit63 = 0) first.
    return a.kmer < b.kmer;
}
// Sort #2 (kmermatcher.h:98-130, compareRepSequenceAndIdAndDiag[Reverse]) compares (rep | bit 63, target id, diagonal) and nothing else:
// nucleotide records of one (rep, target, diagonal) triple that differ in the strand bit of the rep field TIE, and the strand
// writeKmerMatcherResult reports for the pair (kmermatcher.cpp:866-893) is that of the LAST record of the best diagonal's run.  ips4o
// leaves the members of such a small tied group in the order it found them — the order assignGroup wrote them, i.e. sort-#1 order =
// ascending k-mer — so the reference's effective rule is "the strand of the triple's member with the LARGEST k-mer" (measured:
// tests/golden/make_strand_ties.sh — the unmodified reference at --threads 1 and 8 against this file).  Restated as a total order:
// ties of sort #2 are resolved by the k-mer of the record the grouped record was made from (`ord`), then by the strand bit (forward
// last) — the latter only decides between two records of one k-mer run that name the same target on the same diagonal with opposite
// strands (one sequence holding the k-mer and its reverse complement at mirrored positions), where the reference's order follows the
// positions; the GPU path carries exactly (k-mer, strand), so oracle and product agree everywhere.
template <typename T> struct Grouped { KPos<T> r; uint64_t ord; };
template <typename T, bool NUCL>
static bool cmpRepIdDiag(const Grouped<T> &x, const Grouped<T> &y) {
    const KPos<T> &a = x.r, &b = y.r;
    uint64_t ak = NUCL ? (a.kmer | BIT63) : a.kmer, bk = NUCL ? (b.kmer | BIT63) : b.kmer;
    if (ak != bk) return ak < bk;
    if (a.id != b.id) return a.id < b.id;
    if (a.pos != b.pos) return a.pos < b.pos;
    if (x.ord != y.ord) return x.ord < y.ord;       // protein records never get here with different fields: equal triples are equal records
    return a.kmer < b.kmer;
}

static bool canBeCovered(float covThr, int covMode, float q, float t) {  // Util.cpp:533-550
    switch (covMode) {
        case 0: return (q / t >= covThr) && (t / q >= covThr);
        case 1: return (q / t) >= covThr;     // COV_MODE_TARGET = 1, COV_MODE_QUERY = 2 (mm/commons/Parameters.h:246-251)
        case 2: return (t / q) >= covThr;
        case 3: return ((t / q) >= covThr) && (t / q) <= 1.0;
        case 4: return ((q / t) >= covThr) && (q / t) <= 1.0;
        case 5: return (std::min(t, q) / std::max(t, q)) >= covThr;
        default: return true;
    }
}
bool canBeCoveredPublic(float covThr, int covMode, float q, float t) { return canBeCovered(covThr, covMode, q, t); }

std::vector<Hit> parsePrefilterHits(const char *data) {                  // QueryMatcher.h:81-112
    std::vector<Hit> r;
    while (*data != '\0') {
        Hit h; const char *p = data;
        uint32_t a = 0; while (*p >= '0' && *p <= '9') a = a * 10 + (uint32_t) (*p++ - '0');
        while (*p == '\t' || *p == ' ') p++;
        int sg = 1; if (*p == '-') { sg = -1; p++; }
        int b = 0; while (*p >= '0' && *p <= '9') b = b * 10 + (*p++ - '0');
        while (*p == '\t' || *p == ' ') p++;
        int sg2 = 1; if (*p == '-') { sg2 = -1; p++; }
        short c = 0; while (*p >= '0' && *p <= '9') c = (short) (c * 10 + (*p++ - '0'));
        h.seqId = a; h.prefScore = sg * b; h.diagonal = (uint16_t) (short) (sg2 * c);
        r.push_back(h);
        while (*data != '\n') data++;
        data++;
    }
    return r;
}

size_t prefilterHitToBuffer(char *buf, const Hit &h) {                   // QueryMatcher.h:114-126
    char *t = u32toa(h.seqId, buf); *(t - 1) = '\t';
    t = i32toa(h.prefScore, t); *(t - 1) = '\t';
    t = i32toa((int32_t) (short) h.diagonal, t); *(t - 1) = '\n'; *t = '\0';
    return (size_t) (t - buf);
}

// the reference sorts with ips4o's parallel sorter (FastSort.h:8-15); both comparators are total orders on the records that
// occur, so any correct sort gives the same array
template <typename It, typename Cmp> static void sortRecords(It b, It e, Cmp cmp, int threads) {
    if (threads > 1) { omp_set_num_threads(threads); __gnu_parallel::sort(b, e, cmp); }
    else std::sort(b, e, cmp);
}

template <typename T, bool NUCL>
static DB kmermatcherT(const DB &seqDb, const Params &par, KmerStats *stats) {
    const int k = par.kmerSize;
    const unsigned char *map = aa2num(NUCL, par.alphabetSizeAA);
    const unsigned char xCode = map[(int) 'X'];
    const int alphabetSize = NUCL ? 5 : par.alphabetSizeAA;
    // Indexer(subMat->alphabetSize - 1, k): powers[i] = (alphabetSize-1)^i  (Indexer.h / Indexer.cpp)
    std::vector<uint64_t> powers((size_t) k);
    { uint64_t p = 1; for (int i = 0; i < k; i++) { powers[(size_t) i] = p; p *= (uint64_t) (alphabetSize - 1); } }
    const float scale = NUCL ? par.kmersPerSequenceScaleNucl : par.kmersPerSequenceScaleAA;
    const size_t N = seqDb.size();

    // computeKmerCount (kmermatcher.cpp:576-585) -> array size (kmermatcher.cpp:617-622, :41-54)
    size_t totalKmers = 0;
    for (size_t id = 0; id < N; id++) {
        int seqLen = (int) seqDb.seqLen(id);
        int adj = std::max(1, seqLen - k + 2);
        totalKmers += (size_t) std::min(adj, (int) ((size_t) par.kmersPerSequence + (scale * seqLen)));
    }
    const size_t totalKmersPerSplit = std::max((size_t) 1025, totalKmers + 1);
    std::vector<KPos<T>> arr(totalKmersPerSplit + 1);
    memset((void *) arr.data(), 0xFF, sizeof(KPos<T>) * arr.size());

    // ---- K1..K3: fillKmerPositionArray ------------------------------------------------------
    // (the reference runs this loop under OpenMP, kmermatcher.cpp:103-105; here thread t takes the id range [N t / nThr, N (t+1) / nThr)
    //  into its own buffer and the buffers are concatenated in id order, so the array is the single-threaded one)
    const int nThr = std::max(1, par.threads);
    std::vector<std::vector<KPos<T>>> partOut((size_t) nThr);
#pragma omp parallel num_threads(nThr)
    {
    const int tIdx = omp_get_thread_num();
    std::vector<KPos<T>> &arr = partOut[(size_t) tIdx];
    arr.resize(1);
    size_t offset = 0;
    auto put = [&](size_t o) -> KPos<T> & { if (o >= arr.size()) arr.resize(std::max(o + 1, arr.size() * 2)); return arr[o]; };
    std::vector<unsigned char> code;
    std::vector<SeqPos> kmers;
    std::vector<uint16_t> scoreDist(65536);
    uint32_t hier[128];
    for (size_t id = N * (size_t) tIdx / (size_t) nThr; id < N * ((size_t) tIdx + 1) / (size_t) nThr; id++) {
        std::fill(scoreDist.begin(), scoreDist.end(), 0);
        memset(hier, 0, sizeof(hier));
        const char *s = seqDb.entry(id);
        const uint32_t dataLen = seqDb.seqLen(id);
        // Sequence::mapSequence (Sequence.cpp:476-489)
        code.clear();
        for (uint32_t l = 0; l < dataLen && s[l] != '\0' && s[l] != '\n'; l++) code.push_back(map[(unsigned char) s[l]]);
        const int L = (int) code.size();
        const uint32_t seqId = seqDb.key[id];

        uint64_t seqHash = 0;                                  // Util::hash (Util.h:337-345)
        for (int i = 0; i < L; i++) seqHash = seqHash * 31 + code[(size_t) i];
        seqHash = xxh64_u64(seqHash, (uint64_t) par.hashShift);

        kmers.clear();
        for (int pos = 0; pos + k <= L; pos++) {               // hasNextKmer (Sequence.h:98-100)
            const unsigned char *w = code.data() + pos;
            bool hasX = false;
            for (int i = 0; i < k; i++) hasX |= (w[i] == xCode);
            if (hasX) continue;
            SeqPos sp;
            if (NUCL) {
                uint64_t f = 0;
                for (int i = 0; i < k; i++) f = (f << 2) | w[i];            // Indexer::computeKmerIdx
                uint64_t r = revComplement(f, k);
                if (r == f) continue;
                bool pickRev = r < f;
                uint64_t c = pickRev ? r : f;
                sp.score = (uint16_t) xxh64_u64(c, (uint64_t) par.hashShift);
                sp.kmer = pickRev ? (c & ~BIT63) : (c | BIT63);
                sp.pos = (uint32_t) (pickRev ? (L - pos - k) : pos);
            } else {
                uint64_t idx = 0;
                for (int i = 0; i < k; i++) idx += (uint64_t) w[i] * powers[(size_t) i];   // Indexer::int2index
                sp.kmer = idx; sp.pos = (uint32_t) pos;
                sp.score = (uint16_t) xxh64_u64(idx, (uint64_t) par.hashShift);
            }
            scoreDist[sp.score]++; hier[sp.score >> 9]++;
            kmers.push_back(sp);
        }
        const size_t n = kmers.size();
        size_t considered = std::min((size_t) (par.kmersPerSequence - 1 + (scale * L)), n);   // :223
        unsigned threshold = 0; size_t inBins = 0;
        if (n > 0) {                                                                        // :227-237
            size_t h = 0;
            for (h = 0; h < 128 && inBins < considered; h++) inBins += hier[h];
            h -= (h > 0) ? 1 : 0;
            inBins -= hier[h];
            for (threshold = (unsigned) (h * 512); threshold <= USHRT_MAX && inBins < considered; threshold++)
                inBins += scoreDist[threshold];
        }
        int tooMuch = (int) (inBins - considered);

        // identity record (:241-249)
        { KPos<T> &r = put(offset); r.kmer = seqHash; r.id = seqId; r.pos = 0; r.seqLen = (T) L; offset++; }

        if (par.ignoreMultiKmer) {                                                           // :266-272
            std::sort(kmers.begin(), kmers.end(), [](const SeqPos &a, const SeqPos &b) {
                if (a.score != b.score) return a.score < b.score;
                uint64_t ak = NUCL ? (a.kmer | BIT63) : a.kmer, bk = NUCL ? (b.kmer | BIT63) : b.kmer;
                if (ak != bk) return ak < bk;
                return a.pos < b.pos;
            });
        }
        auto K = [&](size_t i) { return NUCL ? (kmers[i].kmer | BIT63) : kmers[i].kmer; };
        size_t selected = 0;
        for (size_t i = 0; i < n && selected < considered; i++) {                            // :274-347
            if (par.ignoreMultiKmer) {
                uint64_t km = K(i);
                if (i + 1 < n) {
                    uint64_t next = K(i + 1);
                    if (km == next) {
                        while (km == next && i < n) {
                            i++;
                            if (i >= n) break;
                            next = K(i);
                        }
                    }
                }
                if (i >= n) break;
            }
            if (kmers[i].score < threshold) {
                if (kmers[i].score == (threshold - 1) && tooMuch) {
                    tooMuch--;
                    threshold -= (tooMuch == 0) ? 1 : 0;
                }
                selected++;
                KPos<T> &r = put(offset); r.kmer = kmers[i].kmer; r.id = seqId;
                r.pos = (T) kmers[i].pos; r.seqLen = (T) L; offset++;
            }
        }
    }
    arr.resize(offset);
    }   // omp parallel
    size_t offset = 0;
    for (int t = 0; t < nThr; t++) { memcpy((void *) (arr.data() + offset), (const void *) partOut[(size_t) t].data(), partOut[(size_t) t].size() * sizeof(KPos<T>)); offset += partOut[(size_t) t].size(); partOut[(size_t) t] = std::vector<KPos<T>>(); }
    const size_t elementsToSort = offset;
    if (stats) stats->nKmerRecords = elementsToSort;

    // ---- K4: sort #1 (:408-412) ----------------------------------------------------------------
    sortRecords(arr.begin(), arr.begin() + (ptrdiff_t) elementsToSort, cmpKmerLenIdPos<T, NUCL>, nThr);

    // ---- K5: assignGroup (:450-559), exact in-place emulation -----------------------------------
    size_t writePos = 0;
    std::vector<uint64_t> ordOf;                    // NUCL: k-mer (| bit 63) of the sort-#1 record each grouped record was made from
    {
        KPos<T> *h = arr.data();
        const size_t splitKmerCount = totalKmersPerSplit;
        uint64_t prevHash = h[0].kmer;
        uint64_t repSeqId = h[0].id;
        if (NUCL) {
            bool isReverse = (h[0].kmer & BIT63) == 0;
            repSeqId = isReverse ? (repSeqId & ~BIT63) : (repSeqId | BIT63);
            prevHash |= BIT63;
        }
        size_t prevHashStart = 0, prevSetSize = 0;
        T queryLen = h[0].seqLen;
        bool repIsReverse = false;
        T repPos = h[0].pos;
        for (size_t e = 0; e < splitKmerCount + 1; e++) {
            uint64_t currKmer = h[e].kmer;
            if (NUCL) currKmer |= BIT63;
            if (prevHash != currKmer) {
                for (size_t i = prevHashStart; i < e; i++) {
                    uint64_t kmer = NUCL ? (h[i].kmer | BIT63) : h[i].kmer;
                    uint64_t rId = (kmer != UINT64_MAX) ? ((prevSetSize == 1) ? UINT64_MAX : repSeqId) : UINT64_MAX;
                    if (rId != UINT64_MAX) {
                        int diagonal = repPos - h[i].pos;
                        if (NUCL) {
                            bool targetIsReverse = (h[i].kmer & BIT63) == 0;
                            bool queryNeedsToBeRev;
                            T queryPos, targetPos;
                            if (repIsReverse && !targetIsReverse) {
                                queryPos = repPos; targetPos = h[i].pos; queryNeedsToBeRev = true;
                            } else if (repIsReverse && targetIsReverse) {
                                queryPos = (T) ((queryLen - 1) - repPos); targetPos = (T) ((h[i].seqLen - 1) - h[i].pos); queryNeedsToBeRev = false;
                            } else if (!repIsReverse && targetIsReverse) {
                                queryPos = (T) ((queryLen - 1) - repPos); targetPos = (T) ((h[i].seqLen - 1) - h[i].pos); queryNeedsToBeRev = true;
                            } else {
                                queryPos = repPos; targetPos = h[i].pos; queryNeedsToBeRev = false;
                            }
                            diagonal = queryPos - targetPos;
                            rId = queryNeedsToBeRev ? (rId & ~BIT63) : (rId | BIT63);
                        }
                        bool canBeExtended = diagonal < 0 || (diagonal > (queryLen - h[i].seqLen));
                        bool cov = canBeCovered(par.covThr, par.covMode, (float) queryLen, (float) h[i].seqLen);
                        if ((!par.includeOnlyExtendable && cov) || (canBeExtended && par.includeOnlyExtendable)) {
                            if (NUCL) ordOf.push_back(kmer);
                            h[writePos].kmer = rId; h[writePos].pos = (T) diagonal;
                            h[writePos].seqLen = h[i].seqLen; h[writePos].id = h[i].id;
                            writePos++;
                        }
                    }
                    h[i].kmer = (i != writePos - 1) ? UINT64_MAX : h[i].kmer;
                }
                prevSetSize = 0; prevHashStart = e;
                repSeqId = h[e].id;
                if (NUCL) {
                    repIsReverse = (h[e].kmer & BIT63) == 0;
                    repSeqId = repIsReverse ? repSeqId : (repSeqId | BIT63);
                }
                queryLen = h[e].seqLen; repPos = h[e].pos;
            }
            if (h[e].kmer == UINT64_MAX) break;
            prevSetSize++;
            prevHash = h[e].kmer;
            if (NUCL) prevHash |= BIT63;
        }
    }
    if (stats) stats->nGrouped = writePos;

    // ---- K6: sort #2 (:427-431) ------------------------------------------------------------------
    {
        std::vector<Grouped<T>> g(writePos);
        for (size_t i = 0; i < writePos; i++) { g[i].r = arr[i]; g[i].ord = (NUCL && !par.debugOldStrandTies) ? ordOf[i] : 0; }
        ordOf = std::vector<uint64_t>();
        sortRecords(g.begin(), g.end(), cmpRepIdDiag<T, NUCL>, nThr);
        for (size_t i = 0; i < writePos; i++) arr[i] = g[i].r;
        if (NUCL && stats) {     // how often the tie-break above decides anything
            size_t triples = 0, pairs = 0;
            for (size_t i = 0; i < writePos;) {
                size_t j = i; bool pairMixed = false;
                while (j < writePos && (arr[j].kmer | BIT63) == (arr[i].kmer | BIT63) && arr[j].id == arr[i].id) {
                    size_t e = j; bool fwd = false, rev = false;
                    while (e < writePos && (arr[e].kmer | BIT63) == (arr[j].kmer | BIT63) && arr[e].id == arr[j].id && arr[e].pos == arr[j].pos) { ((arr[e].kmer & BIT63) ? fwd : rev) = true; e++; }
                    if (fwd && rev) { triples++; pairMixed = true; }
                    j = e;
                }
                pairs += pairMixed; i = j;
            }
            stats->nStrandTieTriples = triples; stats->nStrandTiePairs = pairs;
        }
    }

    // ---- K7/K8: writeKmerMatcherResult, threads = 1 (:809-924) -------------------------------------
    DB out; out.dbtype = NUCL ? DBTYPE_PREFILTER_REV_RES : DBTYPE_PREFILTER_RES;
    uint32_t lastKey = 0; for (uint32_t kk : seqDb.key) lastKey = std::max(lastKey, kk);
    std::vector<char> repSequence((size_t) lastKey + 1, 0);
    size_t nCand = 0;
    {
        const KPos<T> *h = arr.data();
        const size_t end = par.debugNoStaleScan ? writePos : totalKmersPerSplit;
        std::string buf; char tmp[100];
        uint64_t lastTargetId = UINT64_MAX, repSeqId = UINT64_MAX;
        unsigned writeSets = 0;
        for (size_t kp = 0; kp < end && h[kp].kmer != UINT64_MAX; kp++) {
            uint64_t currKmer = h[kp].kmer;
            int reverMask = 0;
            if (NUCL) { reverMask = (currKmer & BIT63) == 0; currKmer &= ~BIT63; }
            if (repSeqId != currKmer) {
                if (writeSets > 0) { repSequence[repSeqId] = 1; out.add((uint32_t) repSeqId, buf.data(), buf.size()); }
                else if (repSeqId != UINT64_MAX) repSequence[repSeqId] = 0;
                lastTargetId = UINT64_MAX; buf.clear(); repSeqId = currKmer;
                Hit hh{(uint32_t) repSeqId, 0, 0};
                buf.append(tmp, prefilterHitToBuffer(tmp, hh));
            }
            unsigned targetId = h[kp].id;
            T diagonal = h[kp].pos;
            size_t kOff = 0; T prevDiagonal = diagonal;
            size_t maxDiagonal = 0, diagonalCnt = 0, topScore = 0;
            int bestReverMask = reverMask;
            while (lastTargetId != targetId && kp + kOff < end && h[kp + kOff].id == targetId) {
                if (prevDiagonal == h[kp + kOff].pos) diagonalCnt++; else diagonalCnt = 1;
                if (diagonalCnt >= maxDiagonal) {
                    diagonal = h[kp + kOff].pos; maxDiagonal = diagonalCnt;
                    if (NUCL) bestReverMask = (h[kp + kOff].kmer & BIT63) == 0;
                }
                prevDiagonal = h[kp + kOff].pos; kOff++; topScore++;
            }
            if (targetId != repSeqId && lastTargetId != targetId) { /* emit below */ }
            else { lastTargetId = targetId; continue; }
            Hit hh; hh.seqId = targetId;
            hh.prefScore = bestReverMask ? -(int) topScore : (int) topScore;
            hh.diagonal = (uint16_t) diagonal;
            buf.append(tmp, prefilterHitToBuffer(tmp, hh));
            lastTargetId = targetId; writeSets++; nCand++;
        }
        if (writeSets > 0) { repSequence[repSeqId] = 1; out.add((uint32_t) repSeqId, buf.data(), buf.size()); }
        else if (repSeqId != UINT64_MAX) repSequence[repSeqId] = 0;
    }
    // self-only back-fill (:705-724)
    for (size_t id = 0; id < N; id++) {
        uint32_t dbKey = seqDb.key[id];
        if (!repSequence[dbKey]) {
            char tmp[100]; Hit hh{dbKey, 0, 0};
            size_t len = prefilterHitToBuffer(tmp, hh);
            out.add(dbKey, tmp, len);
        }
    }
    out.sortByKey();
    if (stats) { stats->nCandidates = nCand; stats->shortT = (sizeof(T) == 2); }
    return out;
}

DB kmermatcher(const DB &seqDb, const Params &par, KmerStats *stats) {
    const bool nucl = seqDb.dbtype == DBTYPE_NUCLEOTIDES;
    const bool shortT = seqDb.maxEntryLen() < SHRT_MAX;                // kmermatcher.cpp:797-802
    if (nucl) return shortT ? kmermatcherT<short, true>(seqDb, par, stats) : kmermatcherT<int, true>(seqDb, par, stats);
    return shortT ? kmermatcherT<short, false>(seqDb, par, stats) : kmermatcherT<int, false>(seqDb, par, stats);
}

}  // namespace oracle
