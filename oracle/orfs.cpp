// ORACLE (test infrastructure): extractorfs / translatenucs / concatdbs restated — the once-per-run
// preprocessing that turns the nucleotide read DB into the protein fragment DB the hot path starts from (data/assemble.sh:40-77).
//   mm/util/extractorfs.cpp:20-159      per read: open reading frames of the six frames, filtered by length / gaps / contig
//                                       start and end modes, written with the read's key, then renumbered 0..M-1 by (key, offset)
//   mm/commons/Orf.cpp:124-336          setSequence (only 'u' -> 't'; reverse complement over the IUPAC table, non-IUPAC -> 'N'),
//                                       findForward (one pass, three frame state machines), writeOrfHeader
//   mm/util/translatenucs.cpp:14-117    codon translation, '*' added in front / behind complete ORF ends (--add-orf-stop)
//   mm/commons/TranslateNucl.h:330-503  IUPAC-aware codon table built by expanding ambiguity codes (B / Z / J / X merges)
//   mm/commons/DBConcat.cpp:19-145      concatdbs: keys of A kept, entry i of B gets key i + max(keyA) + 1
// Only translation table 1 (the canonical code, what the workflows pass) is restated.
#include "oracle.hpp"
#include <algorithm>
#include <climits>
#include <cstring>

namespace oracle {

// ---- IUPAC helpers -----------------------------------------------------------------------------------------------------
static char iupacComplement(char c) {                       // Orf.cpp:47-51 as a function: IUPAC letters, case kept, '.' otherwise
    static const char *from = "ABCDGHKMNRSTUVWY", *to = "TVGHCDMKNYSAABWR";
    const bool lower = (c >= 'a' && c <= 'z');
    const char u = lower ? (char) (c - 32) : c;
    const char *p = (u >= 'A' && u <= 'Z') ? strchr(from, u) : nullptr;
    if (!p) return '.';
    const char r = to[p - from];
    return lower ? (char) (r + 32) : r;
}

// 4-bit base code: A 1, C 2, G 4, T 8, unions for the ambiguity letters (TranslateNucl.h:337-365)
static int baseCode(unsigned char ch) {
    static int tab[256]; static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; i++) tab[i] = 0;
        static const char charToBase[17] = "-ACMGRSVTWYHKDBN";
        for (int i = 0; i <= 15; i++) { tab[(unsigned char) charToBase[i]] = i; tab[(unsigned char) tolower(charToBase[i])] = i; }
        tab['U'] = 8; tab['u'] = 8; tab['X'] = 15; tab['x'] = 15;
        for (int i = 0; i <= 15; i++) tab[i] = i;           // "also map ncbi4na alphabet"
        init = true;
    }
    return tab[ch];
}

// amino acid of a codon of three base codes under translation table 1 (TranslateNucl.h:392-480)
static char codonResidue(int i, int j, int k) {
    static const char *ncbieaa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const int expansions[4] = {1, 2, 4, 8};          // A, C, G, T
    static const int codonIdx[9] = {0, 2, 1, 0, 3, 0, 0, 0, 0};   // T = 0, C = 1, A = 2, G = 3
    char aa = '\0';
    for (int p = 0; p < 4; p++) { const int x = expansions[p]; if (!(x & i)) continue;
        for (int q = 0; q < 4; q++) { const int y = expansions[q]; if (!(y & j)) continue;
            for (int r = 0; r < 4; r++) { const int z = expansions[r]; if (!(z & k)) continue;
                const char ch = ncbieaa[16 * codonIdx[x] + 4 * codonIdx[y] + codonIdx[z]];
                if (aa == '\0') aa = ch;
                else if (aa != ch) {
                    if ((aa == 'B' || aa == 'D' || aa == 'N') && (ch == 'D' || ch == 'N')) aa = 'B';
                    else if ((aa == 'Z' || aa == 'E' || aa == 'Q') && (ch == 'E' || ch == 'Q')) aa = 'Z';
                    else if ((aa == 'J' || aa == 'I' || aa == 'L') && (ch == 'I' || ch == 'L')) aa = 'J';
                    else aa = 'X';
                }
            } } }
    return aa == '\0' ? 'X' : aa;
}

static void translate(char *aa, const char *nucl, int L) {                  // TranslateNucl.h:488-503
    for (int i = 0; i < L; i += 3) {
        bool isLowerCase = false;
        for (int k = 0; k < 3; k++) isLowerCase |= (islower((unsigned char) nucl[i + k]) != 0);
        const char residue = codonResidue(baseCode((unsigned char) nucl[i]), baseCode((unsigned char) nucl[i + 1]), baseCode((unsigned char) nucl[i + 2]));
        aa[i / 3] = isLowerCase ? (char) tolower(residue) : residue;
    }
}

// ---- Orf --------------------------------------------------------------------------------------------------------------
struct SequenceLocation { size_t from, to; bool hasIncompleteStart, hasIncompleteEnd; int strand; };

static bool isStartCodon(const char *c) { return c[0] == 'A' && c[1] == 'T' && c[2] == 'G'; }                 // --use-all-table-starts 0
static bool isStopCodon(const char *c) {                                                                       // table 1: TAA TAG TGA
    return c[0] == 'T' && ((c[1] == 'A' && (c[2] == 'A' || c[2] == 'G')) || (c[1] == 'G' && c[2] == 'A'));
}

static void findForward(const char *sequence, size_t sequenceLength, std::vector<SequenceLocation> &result, size_t minLength, size_t maxLength,
                        size_t maxGaps, unsigned frames, unsigned startMode, int strand) {                     // Orf.cpp:227-336
    const int FRAMES = 3;
    bool isInsideOrf[3] = {true, true, true};
    bool hasStartCodon[3] = {false, false, false};
    size_t countGaps[3] = {0, 0, 0}, countLength[3] = {0, 0, 0};
    size_t from[3] = {0, 1, 2};
    auto isIncomplete = [](const char *c) { return c[0] == CHAR_MAX || c[1] == CHAR_MAX || c[2] == CHAR_MAX; };
    auto isGapOrN = [](const char *c) {
        return c[0] == 'N' || iupacComplement(c[0]) == '.' || c[1] == 'N' || iupacComplement(c[1]) == '.' || c[2] == 'N' || iupacComplement(c[2]) == '.';
    };
    char codon[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < sequenceLength - (FRAMES - 1); i += FRAMES) {
        for (size_t position = i; position < i + FRAMES; position++) {
            for (int c = 0; c < 3; c++) codon[c] = sequence[position + c] == CHAR_MAX ? (char) CHAR_MAX : (char) (sequence[position + c] & (unsigned char) ~0x20);
            const size_t frame = position % FRAMES;
            if (!(frames & (1u << frame))) continue;
            const bool thisIncomplete = isIncomplete(codon);
            const bool isLast = !thisIncomplete && isIncomplete(sequence + position + FRAMES);
            bool shouldStart;
            if (startMode == 0) shouldStart = isInsideOrf[frame] == false && isStartCodon(codon);                // START_TO_STOP
            else if (startMode == 1) shouldStart = isInsideOrf[frame] == false;                                 // ANY_TO_STOP
            else shouldStart = isStartCodon(codon);                                                             // LAST_START_TO_STOP
            if (shouldStart) { isInsideOrf[frame] = true; hasStartCodon[frame] = true; from[frame] = position; countGaps[frame] = 0; countLength[frame] = 0; }
            const bool stop = isStopCodon(codon);
            if (isInsideOrf[frame]) {
                if (!stop) countLength[frame]++;
                if (isGapOrN(codon)) countGaps[frame]++;
            }
            if (isInsideOrf[frame] && (stop || isLast)) {
                isInsideOrf[frame] = false;
                if (countLength[frame] == 0 && stop) continue;
                const size_t to = position + ((isLast && stop == false) ? 2 : -1);
                if (countGaps[frame] > maxGaps || countLength[frame] > maxLength || countLength[frame] < minLength) continue;
                result.push_back({from[frame], to, !hasStartCodon[frame], !stop, strand});
            }
        }
    }
}

static size_t writeOrfHeader(char *buffer, unsigned key, size_t fromPos, size_t toPos, bool hasIncompleteStart, bool hasIncompleteEnd) {   // Orf.cpp:438-456
    int n = sprintf(buffer, "%u\t%u%c%d", key, (unsigned) fromPos, (fromPos < toPos) ? '+' : '-', abs((int) fromPos - (int) toPos));
    const int complete = (hasIncompleteStart ? 1 : 0) | ((hasIncompleteEnd ? 1 : 0) << 1);
    if (complete != 0) n += sprintf(buffer + n, "\t%d", complete);
    buffer[n++] = '\n'; buffer[n] = '\0';
    return (size_t) n;
}

static bool parseOrfHeader(const char *data, bool &hasIncompleteStart, bool &hasIncompleteEnd) {                // Orf.cpp:339-436 (the fields translatenucs uses)
    // "<key>\t<from>[+-]<len>[\t<complete>]\n"
    hasIncompleteStart = false; hasIncompleteEnd = false;
    const char *p = data;
    while (*p && *p != '\t' && *p != '\n') p++;
    if (*p != '\t') return false;
    p++;
    const char *q = p; while (*q >= '0' && *q <= '9') q++;
    if (q == p || (*q != '+' && *q != '-')) return false;
    q++;
    const char *r = q; while (*r >= '0' && *r <= '9') r++;
    if (r == q) return false;
    if (*r == '\t') {
        const int complete = atoi(r + 1);
        // exactly three columns are needed for the flags to be read (columns == 3)
        const char *e = r + 1; while (*e && *e != '\t' && *e != '\n' && *e != ' ') e++;
        const bool third = (*e == '\n' || *e == '\0');
        if (third) { hasIncompleteStart = (complete & 1) != 0; hasIncompleteEnd = (complete & 2) != 0; }
    }
    return true;
}

bool extractorfs(const DB &seqDb, const OrfParams &op, DB &outSeq, DB &outHdr, std::string &err) {
    if (op.translationTable != 1 || op.useAllTableStarts) { err = "oracle extractorfs: only --translation-table 1 --use-all-table-starts 0"; return false; }
    if (op.orfStartMode == 1 && op.contigStartMode < 2) { err = "Parameter combination is illegal, orf-start-mode 1 can only go with contig-start-mode 2"; return false; }   // :38-41
    const int outputDbtype = op.translate ? DBTYPE_AMINO_ACIDS : DBTYPE_NUCLEOTIDES;
    outSeq = DB(); outSeq.dbtype = outputDbtype;
    outHdr = DB(); outHdr.dbtype = 12;                                    // DBTYPE_GENERIC_DB
    const size_t VEC = 8;                                                 // padding behind the sequence (VECSIZE_INT CHAR_MAX bytes; AVX2 build: 8)
    std::vector<char> sequence, reverseComplement, aa;
    std::vector<SequenceLocation> res;
    char buffer[1024];
    unsigned newKey = 0;                                                  // createRenumberedDB: rank by (key, offset) = emission order, keys ascending
    for (size_t i = 0; i < seqDb.size(); i++) {                           // :66-134
        const unsigned key = seqDb.key[i];
        const char *data = seqDb.entry(i);
        const size_t sequenceLength = seqDb.seqLen(i);
        if (sequenceLength < 3) continue;                                 // Orf::setSequence false -> wrongSeqCnt
        sequence.assign(sequenceLength + VEC, (char) CHAR_MAX); reverseComplement.assign(sequenceLength + VEC, (char) CHAR_MAX);
        for (size_t p = 0; p < sequenceLength; p++) sequence[p] = (data[p] == 'u') ? 't' : data[p];            // Orf.cpp:141-144 (the 'U' line is overwritten)
        for (size_t p = 0; p < sequenceLength; p++) {
            char c = iupacComplement(sequence[sequenceLength - p - 1]);
            reverseComplement[p] = (c == '.') ? 'N' : c;
        }
        res.clear();
        if (op.forwardFrames) findForward(sequence.data(), sequenceLength, res, op.orfMinLength, op.orfMaxLength, op.orfMaxGaps, op.forwardFrames, op.orfStartMode, 1);
        if (op.reverseFrames) findForward(reverseComplement.data(), sequenceLength, res, op.orfMinLength, op.orfMaxLength, op.orfMaxGaps, op.reverseFrames, op.orfStartMode, -1);
        for (const SequenceLocation &loc : res) {
            if (op.contigStartMode < 2 && ((int) loc.hasIncompleteStart == op.contigStartMode)) continue;       // :84-89
            if (op.contigEndMode < 2 && ((int) loc.hasIncompleteEnd == op.contigEndMode)) continue;
            const char *first = (loc.strand == 1 ? sequence.data() : reverseComplement.data()) + loc.from;
            size_t second = (loc.to - loc.from) + 1;
            size_t fromPos = loc.from, toPos = loc.to;
            if (loc.strand == -1) { fromPos = (sequenceLength - 1) - loc.from; toPos = (sequenceLength - 1) - loc.to; }
            const size_t hl = writeOrfHeader(buffer, key, fromPos, toPos, loc.hasIncompleteStart, loc.hasIncompleteEnd);
            std::string entry;
            if (op.translate) {                                           // :103-117
                if ((data[second] != '\n' && second % 3 != 0) && (data[second - 1] == '\n' && (second - 1) % 3 != 0)) second = second - (second % 3);
                if (second < 3) continue;
                if (second > 3 * op.maxSeqLen) second = 3 * op.maxSeqLen;
                aa.resize(second / 3 + 4);
                translate(aa.data(), first, (int) second);
                entry.assign(aa.data(), second / 3);
            } else entry.assign(first, second);
            entry.push_back('\n');
            outSeq.add(newKey, entry.data(), entry.size());
            outHdr.add(newKey, buffer, hl);
            newKey++;
        }
    }
    return true;
}

bool translatenucs(const DB &seqDb, const DB *hdrDb, const OrfParams &op, DB &out, std::string &err) {
    if (op.translationTable != 1) { err = "oracle translatenucs: only --translation-table 1"; return false; }
    if (op.addOrfStop && !hdrDb) { err = "translatenucs --add-orf-stop needs the header DB"; return false; }
    out = DB(); out.dbtype = DBTYPE_AMINO_ACIDS;
    std::vector<char> aa;
    for (size_t i = 0; i < seqDb.size(); i++) {                           // :47-104
        const unsigned key = seqDb.key[i];
        const char *data = seqDb.entry(i);
        if (*data == '\0') continue;
        bool addStopAtStart = false, addStopAtEnd = false;
        if (op.addOrfStop) {
            const size_t hid = hdrDb->getId(key);
            if (hid == (size_t) -1) { err = "header entry missing"; return false; }
            bool incS, incE;
            if (!parseOrfHeader(hdrDb->entry(hid), incS, incE)) { incS = false; incE = false; }
            addStopAtStart = !incS; addStopAtEnd = !incE;
        }
        size_t length = seqDb.elen[i] - 1;
        if ((data[length] != '\n' && length % 3 != 0) && (data[length - 1] == '\n' && (length - 1) % 3 != 0)) length = length - (length % 3);
        if (length < 3) continue;
        if (length > 3 * op.maxSeqLen) length = 3 * op.maxSeqLen;
        aa.assign(length / 3 + 8, 0);
        char *writeAA = aa.data();
        if (addStopAtStart) { aa[0] = '*'; writeAA = aa.data() + 1; }
        translate(writeAA, data, (int) length);
        if (addStopAtEnd && writeAA[(length / 3) - 1] != '*') { writeAA[length / 3] = '*'; writeAA[length / 3 + 1] = '\n'; }
        else { addStopAtEnd = false; writeAA[length / 3] = '\n'; }
        out.add(key, aa.data(), (length / 3) + 1 + (addStopAtStart ? 1 : 0) + (addStopAtEnd ? 1 : 0));
    }
    return true;
}

// concatdbs = DBConcat with preserveKeysA = true, preserveKeysB = --preserve-keys (DBConcat.cpp:379-382): A's entries keep their keys;
// B is opened LINEAR_ACCCESS (DBConcat.cpp:46-47), i.e. its ids run in DATA FILE order (DBReader.cpp:335-360 sorts the index by
// offset), and entry id of B becomes key id + max(keyA) + 1 (DBConcat.cpp:113-118).  For a B whose data lies in key order that is
// "B renumbered behind A in key order"; translatenucs run on several threads leaves its data in thread order, and B's new keys then
// follow the FILE (tests/golden/concat_noncanonical.tar.gz: written by the reference with 8 threads).
// --preserve-keys (preserveKeysB, DBConcat.cpp:113-118): B's entries keep their keys as well — the union of the two DBs.
bool concatdbs(const DB &a, const DB &b, DB &out, std::string &, bool preserveKeysB) {
    out = DB(); out.dbtype = a.dbtype;
    if (preserveKeysB) {
        for (size_t i = 0; i < a.size(); i++) out.add(a.key[i], a.entry(i), a.elen[i] - 1);
        for (size_t i = 0; i < b.size(); i++) out.add(b.key[i], b.entry(i), b.elen[i] - 1);
        out.sortByKey();
        return true;
    }
    unsigned maxKeyA = 0;
    for (size_t i = 0; i < a.size(); i++) { out.add(a.key[i], a.entry(i), a.elen[i] - 1); maxKeyA = std::max(maxKeyA, a.key[i]); }
    maxKeyA++;
    std::vector<size_t> ob(b.size());
    for (size_t i = 0; i < ob.size(); i++) ob[i] = i;
    std::stable_sort(ob.begin(), ob.end(), [&](size_t x, size_t y) { return b.off[x] < b.off[y]; });
    for (size_t i = 0; i < ob.size(); i++) out.add((unsigned) i + maxKeyA, b.entry(ob[i]), b.elen[ob[i]] - 1);
    out.sortByKey();
    return true;
}

}  // namespace oracle
