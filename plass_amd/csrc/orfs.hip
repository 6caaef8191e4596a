// plasship: extractorfs + translatenucs + concatdbs on gfx950.  Product code.
//
// The once-per-run preprocessing of data/assemble.sh:41-77 / data/guidedNuclAssemble.sh:46-72 that turns the nucleotide read DB
// into the protein fragment DB the hot path starts from.  Reference behaviour reproduced:
//   mm/util/extractorfs.cpp:20-159      per read: open reading frames of the six frames (Orf::findAll), filtered by
//                                       --contig-start-mode / --contig-end-mode, written with the read's key and renumbered
//                                       0..M-1 in emission order (createRenumberedDB); optional --translate
//   mm/commons/Orf.cpp:124-151          setSequence: only 'u' -> 't' (the 'U' line is overwritten), reverse complement over the
//                                       IUPAC table with unknown letters -> 'N', CHAR_MAX padding behind the sequence
//   mm/commons/Orf.cpp:227-347          findForward: one pass over the positions, three frame state machines
//   mm/commons/Orf.cpp:438-456          writeOrfHeader: "<readKey>\t<from>[+-]<len>[\t<incomplete flags>]\n"
//   mm/util/translatenucs.cpp:14-117    codon translation; '*' in front of / behind ORFs with complete ends (--add-orf-stop)
//   mm/commons/TranslateNucl.h:330-503  IUPAC-aware codon table (ambiguity codes expanded; B / Z / J / X merges); bytes 0..15
//                                       count as ncbi4na codes, so a '\n' inside the last codon reads as 'Y'
//   mm/commons/DBConcat.cpp:19-145      concatdbs: keys of A kept, entry i of B gets key i + max(keyA) + 1
//
// MI355X design: the stop-codon scan is a tiny sequential state machine per (read, strand), so one THREAD runs each of the
// 2 N machines (a 150-nt read is 150 steps of ~30 integer instructions on a rolling six-letter window); a counting pass and an
// emitting pass around a device-wide prefix sum give every ORF its final key and byte offset without atomics, i.e. the
// reference's renumbering by (key, offset) falls out of the layout.  Translation is a 4096-entry LUT over three 4-bit base
// codes (built on the host by expanding the ambiguity codes exactly like the reference).  Integer / byte work only.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <atomic>
#include <memory>
#include <cstring>
#include <climits>

// header DB of an ORF DB (what <orfDB>_h holds), device resident: one record per ORF in key order
struct OrfInfo { uint32_t key, readKey, fromPos, toPos, flags; };      // flags: 1 incomplete start, 2 incomplete end, 4 header unparsable
struct plasship_orfhdr {
    size_t n = 0;
    size_t nUnparsable = 0;             // entries that are not "<id>\t<from>[+-]<len>[\t<flags>]" (flags bit 2)
    plasship::DevBuf d_info;            // OrfInfo[n], in key order
    // rank of every entry in DATA FILE order (empty: the file lay in key order) — like plasship_seqdb::d_fileRank: concatdbs numbers
    // its second DB by it (DBConcat.cpp:46-47,113-118), and extractorfs writes ORFs and headers in the same thread order, so the
    // header DB of a concatenation must be renumbered exactly like its sequence DB or the keys no longer correspond
    plasship::DevBuf d_fileRank;
};

namespace plasship {

// ---- tables ------------------------------------------------------------------------------------------------------------
namespace {
// Orf.cpp:47-51 as a function: IUPAC letters, case kept, '.' otherwise
char iupacComplementHost(char c) {
    static const char *from = "ABCDGHKMNRSTUVWY", *to = "TVGHCDMKNYSAABWR";
    const bool lower = (c >= 'a' && c <= 'z');
    const char u = lower ? (char) (c - 32) : c;
    const char *p = (u >= 'A' && u <= 'Z') ? strchr(from, u) : nullptr;
    if (!p) return '.';
    const char r = to[p - from];
    return lower ? (char) (r + 32) : r;
}
struct OrfTables {
    unsigned char comp[256];        // reverse-strand letter of a forward letter ('.' already turned into 'N')
    unsigned char gap[256];         // 1: the (upper-cased) codon letter counts as gap / unknown (isGapOrN, Orf.cpp:158-162)
    unsigned char base[256];        // TranslateNucl.h:337-365: letter -> 4-bit base code
    unsigned char aa[4096];         // residue of three base codes, translation table 1
};
int baseCodeHost(unsigned char ch) {
    static int tab[256]; static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; i++) tab[i] = 0;
        static const char charToBase[17] = "-ACMGRSVTWYHKDBN";
        for (int i = 0; i <= 15; i++) { tab[(unsigned char) charToBase[i]] = i; tab[(unsigned char) tolower(charToBase[i])] = i; }
        tab['U'] = 8; tab['u'] = 8; tab['X'] = 15; tab['x'] = 15;
        for (int i = 0; i <= 15; i++) tab[i] = i;               // "also map ncbi4na alphabet"
        init = true;
    }
    return tab[ch];
}
char codonResidueHost(int i, int j, int k) {                     // TranslateNucl.h:392-480, canonical code
    static const char *ncbieaa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const int expansions[4] = {1, 2, 4, 8};               // A, C, G, T
    static const int codonIdx[9] = {0, 2, 1, 0, 3, 0, 0, 0, 0};  // T = 0, C = 1, A = 2, G = 3
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
const OrfTables &orfTables() {
    static OrfTables t; static bool init = false;
    if (!init) {
        for (int c = 0; c < 256; c++) {
            const char r = iupacComplementHost((char) c);
            t.comp[c] = (unsigned char) (r == '.' ? 'N' : r);
            t.gap[c] = (unsigned char) ((c == 'N' || r == '.') ? 1 : 0);
            t.base[c] = (unsigned char) baseCodeHost((unsigned char) c);
        }
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 16; k++) t.aa[(i << 8) | (j << 4) | k] = (unsigned char) codonResidueHost(i, j, k);
        init = true;
    }
    return t;
}
}  // namespace

// ---- extractorfs -------------------------------------------------------------------------------------------------------
struct OrfArgs {
    SeqView s; const uint32_t *key;
    const unsigned char *comp, *gap, *base, *aa;
    uint32_t minLength, maxLength, maxGaps;
    int contigStartMode, contigEndMode, orfStartMode;
    uint32_t forwardFrames, reverseFrames;
    int translate; uint64_t maxSeqLen;
    // pass 0: per (read, strand) slot the number of ORFs and of entry bytes; pass 1: their exclusive prefix sums, and the output
    uint32_t *cnt; uint32_t *bytes;
    const uint64_t *orfBase, *byteBase;
    char *outData; uint64_t *outOff; uint32_t *outLen, *outKey; OrfInfo *info;
};

// raw letters of one strand as Orf::setSequence leaves them, CHAR_MAX behind the end; eight bytes per global load
struct StrandReader {
    const char *base; uint32_t L; bool rev; const unsigned char *comp;
    uint64_t word; int64_t wordIdx;
    __device__ __forceinline__ unsigned char fwd(uint32_t q) {               // q < L
        const int64_t wi = (int64_t) (q >> 3);
        if (wi != wordIdx) { __builtin_memcpy(&word, base + ((size_t) wi << 3), 8); wordIdx = wi; }    // DB data is padded behind its end
        unsigned char c = (unsigned char) (word >> (8 * (q & 7)));
        return c == 'u' ? (unsigned char) 't' : c;
    }
    __device__ __forceinline__ unsigned char at(uint32_t p) {
        if (p >= L) return (unsigned char) CHAR_MAX;
        return rev ? comp[fwd(L - 1 - p)] : fwd(p);
    }
};

template <int PASS>
__global__ __launch_bounds__(256) void orfKernel(OrfArgs a) {
    __shared__ unsigned char sComp[256], sGap[256], sBase[256];
    for (int i = threadIdx.x; i < 256; i += 256) { sComp[i] = a.comp[i]; sGap[i] = a.gap[i]; sBase[i] = a.base[i]; }
    __syncthreads();
    const uint64_t nSlots = 2ull * a.s.n;
    for (uint64_t slot = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; slot < nSlots; slot += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t) (slot >> 1); const bool rev = (slot & 1) != 0;
        const uint32_t L = a.s.len[id];
        const uint32_t frames = rev ? a.reverseFrames : a.forwardFrames;
        uint32_t nOrf = 0, nBytes = 0;
        uint64_t orfIdx = 0, byteOff = 0;
        if (PASS == 1) { orfIdx = a.orfBase[slot]; byteOff = a.byteBase[slot]; }
        if (L >= 3 && frames != 0) {                                         // Orf::setSequence refuses shorter sequences
            const char *data = a.s.data + a.s.off[id];
            StrandReader rd; rd.base = data; rd.L = L; rd.rev = rev; rd.comp = sComp; rd.word = 0; rd.wordIdx = -1;
            // w[j]: letter at position + j as findForward sees it: upper-cased (& ~0x20), CHAR_MAX kept
            unsigned char w[6];
            auto up = [](unsigned char c) { return c == (unsigned char) CHAR_MAX ? c : (unsigned char) (c & 0xDF); };
#pragma unroll
            for (int j = 0; j < 6; j++) w[j] = up(rd.at((uint32_t) j));
            bool inside[3] = {true, true, true}, hasStart[3] = {false, false, false};
            uint32_t gaps[3] = {0, 0, 0}, length[3] = {0, 0, 0}, from[3] = {0, 1, 2};
            const uint32_t nPos = 3 * (L / 3);                               // for (i = 0; i < L - 2; i += 3) for (position = i .. i + 2)
            uint32_t frame = 0;
            for (uint32_t position = 0; position < nPos; position++) {
                if (frames & (1u << frame)) {
                    const bool thisIncomplete = w[0] == CHAR_MAX || w[1] == CHAR_MAX || w[2] == CHAR_MAX;
                    const bool isLast = !thisIncomplete && (w[3] == CHAR_MAX || w[4] == CHAR_MAX || w[5] == CHAR_MAX);
                    const bool isStart = w[0] == 'A' && w[1] == 'T' && w[2] == 'G';                 // --use-all-table-starts 0
                    const bool stop = w[0] == 'T' && ((w[1] == 'A' && (w[2] == 'A' || w[2] == 'G')) || (w[1] == 'G' && w[2] == 'A'));   // table 1
                    // state of this frame (indexing registers by a runtime frame would spill: select explicitly)
                    bool in = frame == 0 ? inside[0] : (frame == 1 ? inside[1] : inside[2]);
                    bool hs = frame == 0 ? hasStart[0] : (frame == 1 ? hasStart[1] : hasStart[2]);
                    uint32_t gp = frame == 0 ? gaps[0] : (frame == 1 ? gaps[1] : gaps[2]);
                    uint32_t ln = frame == 0 ? length[0] : (frame == 1 ? length[1] : length[2]);
                    uint32_t fr = frame == 0 ? from[0] : (frame == 1 ? from[1] : from[2]);
                    bool shouldStart;
                    if (a.orfStartMode == 0) shouldStart = !in && isStart;                         // START_TO_STOP
                    else if (a.orfStartMode == 1) shouldStart = !in;                               // ANY_TO_STOP
                    else shouldStart = isStart;                                                    // LAST_START_TO_STOP
                    if (shouldStart) { in = true; hs = true; fr = position; gp = 0; ln = 0; }
                    if (in) {
                        if (!stop) ln++;
                        if (sGap[w[0]] | sGap[w[1]] | sGap[w[2]]) gp++;
                    }
                    if (in && (stop || isLast)) {
                        in = false;
                        if (!(ln == 0 && stop)) {
                            const uint32_t to = (isLast && !stop) ? position + 2 : position - 1;
                            if (!(gp > a.maxGaps || ln > a.maxLength || ln < a.minLength)) {
                                const bool incS = !hs, incE = !stop;
                                bool keep = true;
                                if (a.contigStartMode < 2 && ((int) incS == a.contigStartMode)) keep = false;     // extractorfs.cpp:84-89
                                if (a.contigEndMode < 2 && ((int) incE == a.contigEndMode)) keep = false;
                                uint32_t second = to - fr + 1;
                                if (keep && a.translate) {                                            // extractorfs.cpp:103-113 (indexes the READ, as written there)
                                    if ((data[second] != '\n' && second % 3 != 0) && (data[second - 1] == '\n' && (second - 1) % 3 != 0)) second = second - (second % 3);
                                    if (second < 3) keep = false;
                                    else if ((uint64_t) second > 3 * a.maxSeqLen) second = (uint32_t) (3 * a.maxSeqLen);
                                }
                                if (keep) {
                                    const uint32_t seqBytes = a.translate ? second / 3 : second;
                                    if (PASS == 1) {
                                        char *d = a.outData + byteOff;
                                        StrandReader r2 = rd; r2.wordIdx = -1;
                                        if (a.translate) {
                                            for (uint32_t r = 0; r < seqBytes; r++) {
                                                const unsigned char c0 = r2.at(fr + 3 * r), c1 = r2.at(fr + 3 * r + 1), c2 = r2.at(fr + 3 * r + 2);
                                                const bool lower = (c0 >= 'a' && c0 <= 'z') || (c1 >= 'a' && c1 <= 'z') || (c2 >= 'a' && c2 <= 'z');
                                                unsigned char res = a.aa[((uint32_t) sBase[c0] << 8) | ((uint32_t) sBase[c1] << 4) | sBase[c2]];
                                                if (lower && res >= 'A' && res <= 'Z') res = (unsigned char) (res + 32);
                                                d[r] = (char) res;
                                            }
                                        } else {
                                            for (uint32_t i = 0; i < second; i++) d[i] = (char) r2.at(fr + i);
                                        }
                                        d[seqBytes] = '\n'; d[seqBytes + 1] = '\0';
                                        a.outOff[orfIdx] = byteOff; a.outLen[orfIdx] = seqBytes; a.outKey[orfIdx] = (uint32_t) orfIdx;
                                        OrfInfo oi; oi.key = (uint32_t) orfIdx; oi.readKey = a.key[id];
                                        oi.fromPos = rev ? (L - 1) - fr : fr; oi.toPos = rev ? (L - 1) - to : to;
                                        oi.flags = (incS ? 1u : 0u) | (incE ? 2u : 0u);
                                        a.info[orfIdx] = oi;
                                        orfIdx++; byteOff += seqBytes + 2;
                                    }
                                    nOrf++; nBytes += seqBytes + 2;
                                }
                            }
                        }
                    }
                    if (frame == 0) { inside[0] = in; hasStart[0] = hs; gaps[0] = gp; length[0] = ln; from[0] = fr; }
                    else if (frame == 1) { inside[1] = in; hasStart[1] = hs; gaps[1] = gp; length[1] = ln; from[1] = fr; }
                    else { inside[2] = in; hasStart[2] = hs; gaps[2] = gp; length[2] = ln; from[2] = fr; }
                }
                // slide the window by one position
                w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; w[5] = up(rd.at(position + 6));
                frame = (frame == 2) ? 0 : frame + 1;
            }
        }
        if (PASS == 0) { a.cnt[slot] = nOrf; a.bytes[slot] = nBytes; }
    }
}

// ---- translatenucs -----------------------------------------------------------------------------------------------------
struct TransArgs {
    SeqView s; const uint32_t *key; const OrfInfo *info; const unsigned char *base, *aa;
    int addOrfStop; uint64_t maxSeqLen;
    uint32_t *bytes;                 // pass 0: entry bytes (0 = entry skipped)
    const uint64_t *byteBase, *idxBase;      // pass 1: prefix sums of bytes / of (bytes != 0)
    char *outData; uint64_t *outOff; uint32_t *outLen, *outKey;
};
template <int PASS>
__global__ __launch_bounds__(256) void translateKernel(TransArgs a) {
    __shared__ unsigned char sBase[256];
    for (int i = threadIdx.x; i < 256; i += 256) sBase[i] = a.base[i];
    __syncthreads();
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < a.s.n; id += gridDim.x * blockDim.x) {
        const char *data = a.s.data + a.s.off[id];
        uint32_t length = a.s.len[id] + 1;                                   // entry without its '\0' (translatenucs.cpp:75)
        uint32_t out = 0;
        bool stopStart = false, stopEnd = false;
        if (*data != '\0') {
            if (a.addOrfStop) { const uint32_t fl = a.info[id].flags; stopStart = !(fl & 1u); stopEnd = !(fl & 2u); if (fl & 4u) { stopStart = false; stopEnd = false; } }
            if ((data[length] != '\n' && length % 3 != 0) && (data[length - 1] == '\n' && (length - 1) % 3 != 0)) length = length - (length % 3);
            if (length >= 3) {
                if ((uint64_t) length > 3 * a.maxSeqLen) length = (uint32_t) (3 * a.maxSeqLen);
                const uint32_t nRes = length / 3;
                // the last residue decides whether a '*' is appended (translatenucs.cpp:97-103)
                auto residue = [&](uint32_t r) {
                    const unsigned char c0 = (unsigned char) data[3 * r], c1 = (unsigned char) data[3 * r + 1], c2 = (unsigned char) data[3 * r + 2];
                    const bool lower = (c0 >= 'a' && c0 <= 'z') || (c1 >= 'a' && c1 <= 'z') || (c2 >= 'a' && c2 <= 'z');
                    unsigned char res = a.aa[((uint32_t) sBase[c0] << 8) | ((uint32_t) sBase[c1] << 4) | sBase[c2]];
                    if (lower && res >= 'A' && res <= 'Z') res = (unsigned char) (res + 32);
                    return res;
                };
                if (stopEnd && residue(nRes - 1) == '*') stopEnd = false;
                out = nRes + (stopStart ? 1u : 0u) + (stopEnd ? 1u : 0u) + 2u;                     // + "\n\0"
                if (PASS == 1) {
                    const uint64_t o = a.byteBase[id], k = a.idxBase[id];
                    char *d = a.outData + o;
                    uint32_t q = 0;
                    if (stopStart) d[q++] = '*';
                    for (uint32_t r = 0; r < nRes; r++) d[q++] = (char) residue(r);
                    if (stopEnd) d[q++] = '*';
                    d[q] = '\n'; d[q + 1] = '\0';
                    a.outOff[k] = o; a.outLen[k] = q; a.outKey[k] = a.key[id];
                }
            }
        }
        if (PASS == 0) a.bytes[id] = out;
    }
}
__global__ void nonZeroKernel(const uint32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] ? 1u : 0u;
}

// ---- concatdbs ---------------------------------------------------------------------------------------------------------
__global__ void concatIndexKernel(const uint64_t *__restrict__ offA, const uint32_t *__restrict__ lenA, const uint32_t *__restrict__ keyA, uint32_t nA,
                                  const uint64_t *__restrict__ offB, const uint32_t *__restrict__ lenB, uint32_t nB, uint64_t bytesA, uint32_t keyBase,
                                  uint64_t *__restrict__ off, uint32_t *__restrict__ len, uint32_t *__restrict__ key) {
    const uint32_t n = nA + nB;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        if (i < nA) { off[i] = offA[i]; len[i] = lenA[i]; key[i] = keyA[i]; }
        else if (i < n) { off[i] = bytesA + offB[i - nA]; len[i] = lenB[i - nA]; key[i] = keyBase + (i - nA); }
        else off[i] = bytesA + (nB ? offB[nB] : 0);
    }
}
// B in data file order (a DB read from thread-written files, plasship_seqdb::d_fileRank): entry j of B goes to position nA + rank[j]
__global__ void concatRankLenKernel(const uint32_t *__restrict__ lenB, const uint32_t *__restrict__ rank, uint32_t nB, uint32_t *__restrict__ lenOut, uint64_t *__restrict__ bytesOut) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nB; j += gridDim.x * blockDim.x) { lenOut[rank[j]] = lenB[j]; bytesOut[rank[j]] = (uint64_t) lenB[j] + 2; }
}
__global__ __launch_bounds__(256) void concatRankCopyKernel(const char *__restrict__ dataB, const uint64_t *__restrict__ offB, const uint32_t *__restrict__ lenB, const uint32_t *__restrict__ rank, uint32_t nB,
                                                            const uint64_t *__restrict__ newOff, uint64_t bytesA, uint32_t nA, uint32_t keyBase, char *__restrict__ data, uint64_t *__restrict__ off, uint32_t *__restrict__ key) {
    const int G = 16, gl = threadIdx.x & (G - 1);
    for (uint32_t j = blockIdx.x * (256 / G) + threadIdx.x / G; j < nB; j += gridDim.x * (256 / G)) {
        const uint32_t r = rank[j], el = lenB[j] + 2; const uint64_t d = bytesA + newOff[r], s = offB[j];
        for (uint32_t i = gl; i < el; i += G) data[d + i] = dataB[s + i];
        if (gl == 0) { off[nA + r] = d; key[nA + r] = keyBase + r; }
    }
}
// rank != nullptr: entry j of B (key order) lay at place rank[j] of B's data file and is numbered by that
__global__ void concatInfoKernel(const OrfInfo *__restrict__ a, uint32_t nA, const OrfInfo *__restrict__ b, uint32_t nB, const uint32_t *__restrict__ rank, uint32_t keyBase, OrfInfo *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB; i += gridDim.x * blockDim.x) {
        if (i < nA) out[i] = a[i];
        else { const uint32_t j = i - nA, r = rank ? rank[j] : j; OrfInfo o = b[j]; o.key = keyBase + r; out[nA + r] = o; }
    }
}
__global__ void maxLenKernel(const uint32_t *__restrict__ v, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
    m = (uint32_t) waveReduceMax((int) m);                                    // sequence lengths are far below 2^31
    if (laneId() == 0 && m) atomicMax(out, m);
}

static unsigned gridOf(uint64_t n, int numCU) { return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, (uint64_t) numCU * 32)); }

// finishes a device-built sequence DB: host-side totals that the handle carries
static int finishSeqdb(plasship_ctx *ctx, plasship_seqdb *o, size_t n, uint64_t dataBytes, int dbtype, const uint32_t *dLen) {
    o->dbtype = dbtype; o->n = n; o->dataBytes = dataBytes; o->residues = dataBytes - 2 * (uint64_t) n; o->hostIndexValid = false;
    // longest entry (drives the KmerPosition<short>/<int> choice of kmermatcher)
    DevBuf dMax; uint32_t mx = 0;
    if (dMax.alloc(4) != hipSuccess) { setError("out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dMax.p, 0, 4, ctx->stream));
    if (n) hipLaunchKernelGGL(maxLenKernel, dim3(gridOf(n, ctx->numCU)), dim3(256), 0, ctx->stream, dLen, (uint32_t) n, dMax.as<uint32_t>());
    PH_COPY_SYNC(ctx->stream, &mx, dMax.p, 4, hipMemcpyDeviceToHost);
    o->maxEntryLen = n ? mx + 2 : 0;
    return PLASSHIP_OK;
}

}  // namespace plasship
using namespace plasship;

static int uploadTables(plasship_ctx *ctx, DevBuf &d, const unsigned char **comp, const unsigned char **gap, const unsigned char **base, const unsigned char **aa) {
    const OrfTables &t = orfTables();
    if (d.alloc(sizeof(OrfTables)) != hipSuccess) { setError("out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemcpyAsync(d.p, &t, sizeof(OrfTables), hipMemcpyHostToDevice, ctx->stream));
    const unsigned char *p = d.as<unsigned char>();
    *comp = p + offsetof(OrfTables, comp); *gap = p + offsetof(OrfTables, gap); *base = p + offsetof(OrfTables, base); *aa = p + offsetof(OrfTables, aa);
    return PLASSHIP_OK;
}

extern "C" int plasship_extract_orfs(plasship_ctx *ctx, const plasship_seqdb *reads, const plasship_orf_params *par, plasship_seqdb **out_orfs,
                                     plasship_orfhdr **out_hdr, plasship_orf_stats *stats) {
    if (!ctx || !reads || !par || !out_orfs) { setError("plasship_extract_orfs: bad argument"); return PLASSHIP_ERR_ARG; }
    if (reads->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES) { setError("plasship_extract_orfs: needs a nucleotide sequence DB"); return PLASSHIP_ERR_ARG; }
    if (par->translation_table != 1 || par->use_all_table_starts) { setError("plasship_extract_orfs: only --translation-table 1 --use-all-table-starts 0 are supported"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (par->orf_start_mode == 1 && par->contig_start_mode < 2) {                                   // extractorfs.cpp:38-41
        setError("Parameter combination is illegal, orf-start-mode 1 can only go with contig-start-mode 2"); return PLASSHIP_ERR_ARG;
    }
    if (par->orf_start_mode < 0 || par->orf_start_mode > 2 || par->min_length < 0 || par->max_length < 0 || par->max_gaps < 0 ||
        (par->forward_frames & ~7) || (par->reverse_frames & ~7) || (par->translate && par->max_seq_len == 0)) { setError("plasship_extract_orfs: bad parameter"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) reads->n;
    const uint64_t nSlots = 2ull * N;
    if (nSlots >= 0xFFFFFFFFull) { setError("plasship_extract_orfs: too many sequences"); return PLASSHIP_ERR_UNSUPPORTED; }
    DevBuf dTab, dCnt, dBytes, dOrfBase, dByteBase, dTmp;
    const size_t tmpBytes = exclusiveScanTmpBytes((size_t) nSlots + 2);
    OrfArgs a; memset(&a, 0, sizeof(a));
    int rc = uploadTables(ctx, dTab, &a.comp, &a.gap, &a.base, &a.aa); if (rc) return rc;
    if (dCnt.alloc((nSlots + 1) * 4) != hipSuccess || dBytes.alloc((nSlots + 1) * 4) != hipSuccess || dOrfBase.alloc((nSlots + 2) * 8) != hipSuccess ||
        dByteBase.alloc((nSlots + 2) * 8) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_extract_orfs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    a.s = reads->view(); a.key = reads->d_key.as<uint32_t>();
    a.minLength = (uint32_t) par->min_length; a.maxLength = (uint32_t) par->max_length; a.maxGaps = (uint32_t) par->max_gaps;
    a.contigStartMode = par->contig_start_mode; a.contigEndMode = par->contig_end_mode; a.orfStartMode = par->orf_start_mode;
    a.forwardFrames = (uint32_t) par->forward_frames; a.reverseFrames = (uint32_t) par->reverse_frames;
    a.translate = par->translate ? 1 : 0; a.maxSeqLen = par->max_seq_len;
    a.cnt = dCnt.as<uint32_t>(); a.bytes = dBytes.as<uint32_t>();
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    const unsigned grid = gridOf(nSlots, ctx->numCU);
    if (nSlots) hipLaunchKernelGGL((orfKernel<0>), dim3(grid), dim3(256), 0, st, a);
    if (exclusiveScanU32(st, dCnt.as<uint32_t>(), dOrfBase.as<uint64_t>(), nSlots, dTmp.p, tmpBytes) ||
        exclusiveScanU32(st, dBytes.as<uint32_t>(), dByteBase.as<uint64_t>(), nSlots, dTmp.p, tmpBytes)) { setError("plasship_extract_orfs: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t tot[2] = {0, 0};
    PH_CHECK(hipMemcpyAsync(&tot[0], dOrfBase.as<uint64_t>() + nSlots, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&tot[1], dByteBase.as<uint64_t>() + nSlots, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    const uint64_t M = tot[0], dataBytes = tot[1];
    if (M >= 0xFFFFFFFFull) { setError("plasship_extract_orfs: too many ORFs"); return PLASSHIP_ERR_UNSUPPORTED; }
    std::unique_ptr<plasship_seqdb> o(new plasship_seqdb());
    std::unique_ptr<plasship_orfhdr> h(new plasship_orfhdr());
    if (o->d_data.allocLong(dataBytes + 64) != hipSuccess || o->d_off.allocLong((M + 1) * 8) != hipSuccess || o->d_len.allocLong((M + 1) * 4) != hipSuccess ||
        o->d_key.allocLong((M + 1) * 4) != hipSuccess || h->d_info.alloc((M + 1) * sizeof(OrfInfo)) != hipSuccess) { setError("plasship_extract_orfs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync((char *) o->d_data.p + dataBytes, 0, 64, st));
    PH_CHECK(hipMemcpyAsync(o->d_off.as<uint64_t>() + M, &dataBytes, 8, hipMemcpyHostToDevice, st));
    a.orfBase = dOrfBase.as<uint64_t>(); a.byteBase = dByteBase.as<uint64_t>();
    a.outData = o->d_data.as<char>(); a.outOff = o->d_off.as<uint64_t>(); a.outLen = o->d_len.as<uint32_t>(); a.outKey = o->d_key.as<uint32_t>(); a.info = h->d_info.as<OrfInfo>();
    if (nSlots) hipLaunchKernelGGL((orfKernel<1>), dim3(grid), dim3(256), 0, st, a);
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    rc = finishSeqdb(ctx, o.get(), (size_t) M, dataBytes, par->translate ? PLASSHIP_DBTYPE_AMINO_ACIDS : PLASSHIP_DBTYPE_NUCLEOTIDES, o->d_len.as<uint32_t>());
    if (rc) return rc;
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    h->n = (size_t) M;
    if (stats) {
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        stats->ms_kernel = ms; stats->n_out = M; stats->out_residues = o->residues; stats->in_residues = reads->residues;
    }
    *out_orfs = o.release();
    if (out_hdr) *out_hdr = h.release();
    return PLASSHIP_OK;
}

extern "C" int plasship_translate_nucs(plasship_ctx *ctx, const plasship_seqdb *orfs, const plasship_orfhdr *hdr, const plasship_translate_params *par,
                                       plasship_seqdb **out_aa, plasship_orf_stats *stats) {
    if (!ctx || !orfs || !par || !out_aa) { setError("plasship_translate_nucs: bad argument"); return PLASSHIP_ERR_ARG; }
    if (orfs->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES) { setError("plasship_translate_nucs: needs a nucleotide sequence DB"); return PLASSHIP_ERR_ARG; }
    if (par->translation_table != 1) { setError("plasship_translate_nucs: only --translation-table 1 is supported"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (par->max_seq_len == 0) { setError("plasship_translate_nucs: bad --max-seq-len"); return PLASSHIP_ERR_ARG; }
    if (par->add_orf_stop && (!hdr || hdr->n != orfs->n)) { setError("plasship_translate_nucs: --add-orf-stop needs the header DB of the ORF DB (same keys)"); return PLASSHIP_ERR_ARG; }
    // Orf::parseOrfHeader leaves the two flags uninitialised when a header is not in ORF format (Orf.cpp:401-405): no defined result to reproduce
    if (par->add_orf_stop && hdr->nUnparsable) { setError("plasship_translate_nucs: --add-orf-stop with headers that are not ORF headers is undefined in the reference; refused"); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) orfs->n;
    DevBuf dTab, dBytes, dFlag, dByteBase, dIdxBase, dTmp;
    const size_t tmpBytes = exclusiveScanTmpBytes((size_t) N + 2);
    TransArgs a; memset(&a, 0, sizeof(a));
    const unsigned char *comp, *gap;
    int rc = uploadTables(ctx, dTab, &comp, &gap, &a.base, &a.aa); if (rc) return rc;
    if (dBytes.alloc(((size_t) N + 1) * 4) != hipSuccess || dFlag.alloc(((size_t) N + 1) * 4) != hipSuccess || dByteBase.alloc(((size_t) N + 2) * 8) != hipSuccess ||
        dIdxBase.alloc(((size_t) N + 2) * 8) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_translate_nucs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (par->add_orf_stop && N) {                  // the header of ORF i must be the header OF ORF i (translatenucs looks it up by key)
        DevBuf dHK; if (dHK.alloc((size_t) N * 4) != hipSuccess) { setError("plasship_translate_nucs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemcpy2DAsync(dHK.p, 4, hdr->d_info.p, sizeof(OrfInfo), 4, N, hipMemcpyDeviceToDevice, st));      // column `key`
        bool differ = false; rc = deviceKeysDiffer(ctx, dHK.as<uint32_t>(), orfs->d_key.as<uint32_t>(), N, &differ); if (rc) return rc;
        if (differ) { setError("plasship_translate_nucs: header DB and ORF DB hold different keys"); return PLASSHIP_ERR_ARG; }
    }
    a.s = orfs->view(); a.key = orfs->d_key.as<uint32_t>(); a.info = hdr ? hdr->d_info.as<OrfInfo>() : nullptr;
    a.addOrfStop = par->add_orf_stop ? 1 : 0; a.maxSeqLen = par->max_seq_len; a.bytes = dBytes.as<uint32_t>();
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    const unsigned grid = gridOf(N, ctx->numCU);
    if (N) {
        hipLaunchKernelGGL((translateKernel<0>), dim3(grid), dim3(256), 0, st, a);
        hipLaunchKernelGGL(nonZeroKernel, dim3(grid), dim3(256), 0, st, dBytes.as<uint32_t>(), N, dFlag.as<uint32_t>());
    }
    if (exclusiveScanU32(st, dBytes.as<uint32_t>(), dByteBase.as<uint64_t>(), N, dTmp.p, tmpBytes) ||
        exclusiveScanU32(st, dFlag.as<uint32_t>(), dIdxBase.as<uint64_t>(), N, dTmp.p, tmpBytes)) { setError("plasship_translate_nucs: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t tot[2] = {0, 0};
    PH_CHECK(hipMemcpyAsync(&tot[0], dIdxBase.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(&tot[1], dByteBase.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    const uint64_t M = tot[0], dataBytes = tot[1];
    std::unique_ptr<plasship_seqdb> o(new plasship_seqdb());
    if (o->d_data.allocLong(dataBytes + 64) != hipSuccess || o->d_off.allocLong((M + 1) * 8) != hipSuccess || o->d_len.allocLong((M + 1) * 4) != hipSuccess ||
        o->d_key.allocLong((M + 1) * 4) != hipSuccess) { setError("plasship_translate_nucs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync((char *) o->d_data.p + dataBytes, 0, 64, st));
    PH_CHECK(hipMemcpyAsync(o->d_off.as<uint64_t>() + M, &dataBytes, 8, hipMemcpyHostToDevice, st));
    a.byteBase = dByteBase.as<uint64_t>(); a.idxBase = dIdxBase.as<uint64_t>();
    a.outData = o->d_data.as<char>(); a.outOff = o->d_off.as<uint64_t>(); a.outLen = o->d_len.as<uint32_t>(); a.outKey = o->d_key.as<uint32_t>();
    if (N) hipLaunchKernelGGL((translateKernel<1>), dim3(grid), dim3(256), 0, st, a);
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    rc = finishSeqdb(ctx, o.get(), (size_t) M, dataBytes, PLASSHIP_DBTYPE_AMINO_ACIDS, o->d_len.as<uint32_t>());
    if (rc) return rc;
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    if (stats) {
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        stats->ms_kernel = ms; stats->n_out = M; stats->out_residues = o->residues; stats->in_residues = orfs->residues;
    }
    *out_aa = o.release();
    return PLASSHIP_OK;
}

// concatdbs <A> <B> <out> (DBConcat.cpp:63-135 with preserveKeysA, !preserveKeysB, DBConcat.cpp:379-382): A's entries with their keys,
// followed by B's with keys max(keyA) + 1 + i where i runs over B in DATA FILE order (B is opened LINEAR_ACCCESS: DBConcat.cpp:46-47,
// 113-118).  Handles hold their entries in key order; one whose file order was different carries d_fileRank (common.hpp)
static int maxKeyOf(plasship_ctx *ctx, const uint32_t *dKey, size_t n, uint32_t *out) {
    *out = 0;
    if (!n) return PLASSHIP_OK;
    PH_COPY_SYNC(ctx->stream, out, dKey + (n - 1), 4, hipMemcpyDeviceToHost);       // ids are ranks in key order: the last key is the largest
    return PLASSHIP_OK;
}
extern "C" int plasship_seqdb_concat(plasship_ctx *ctx, const plasship_seqdb *a, const plasship_seqdb *b, plasship_seqdb **out) {
    if (!ctx || !a || !b || !out) { setError("plasship_seqdb_concat: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    uint32_t maxKeyA = 0; int rc = maxKeyOf(ctx, a->d_key.as<uint32_t>(), a->n, &maxKeyA); if (rc) return rc;
    const uint64_t nn = (uint64_t) a->n + b->n;
    if (nn >= 0xFFFFFFFFull || (uint64_t) maxKeyA + 1 + b->n > 0xFFFFFFFFull) { setError("plasship_seqdb_concat: too many sequences"); return PLASSHIP_ERR_UNSUPPORTED; }
    // (concatenation copies the two data blocks as they are: a DB that lives in a shared heap is packed first)
    std::unique_ptr<plasship_seqdb> packedA, packedB;
    if (!a->contiguous) { const int rcP = packedCopyOf(ctx, a, packedA); if (rcP) return rcP; a = packedA.get(); }
    if (!b->contiguous) { const int rcP = packedCopyOf(ctx, b, packedB); if (rcP) return rcP; b = packedB.get(); }
    const uint64_t dataBytes = a->dataBytes + b->dataBytes;
    std::unique_ptr<plasship_seqdb> o(new plasship_seqdb());
    if (o->d_data.allocLong(dataBytes + 64) != hipSuccess || o->d_off.allocLong((nn + 1) * 8) != hipSuccess || o->d_len.allocLong((nn + 1) * 4) != hipSuccess ||
        o->d_key.allocLong((nn + 1) * 4) != hipSuccess) { setError("plasship_seqdb_concat: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (a->dataBytes) PH_CHECK(hipMemcpyAsync(o->d_data.p, a->dataPtr(), a->dataBytes, hipMemcpyDeviceToDevice, st));
    if (b->dataBytes && !b->d_fileRank.p) PH_CHECK(hipMemcpyAsync((char *) o->d_data.p + a->dataBytes, b->dataPtr(), b->dataBytes, hipMemcpyDeviceToDevice, st));
    PH_CHECK(hipMemsetAsync((char *) o->d_data.p + dataBytes, 0, 64, st));
    hipLaunchKernelGGL(concatIndexKernel, dim3(gridOf(nn + 1, ctx->numCU)), dim3(256), 0, st, a->d_off.as<uint64_t>(), a->d_len.as<uint32_t>(), a->d_key.as<uint32_t>(), (uint32_t) a->n,
                       b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), (uint32_t) b->n, a->dataBytes, maxKeyA + 1, o->d_off.as<uint64_t>(), o->d_len.as<uint32_t>(), o->d_key.as<uint32_t>());
    if (b->d_fileRank.p && b->n) {
        // B was read from files whose data is not in key order: the reference numbers B's entries in FILE order (see common.hpp),
        // so the B half of the output is B permuted by its file rank (lengths scattered, offsets by a prefix sum, entries copied)
        DevBuf dBytes, dNewOff, dTmp; const size_t tmpBytes = exclusiveScanTmpBytes(b->n + 2);
        if (dBytes.alloc((b->n + 1) * 8) != hipSuccess || dNewOff.alloc((b->n + 2) * 8) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_seqdb_concat: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(concatRankLenKernel, dim3(gridOf(b->n, ctx->numCU)), dim3(256), 0, st, b->d_len.as<uint32_t>(), b->d_fileRank.as<uint32_t>(), (uint32_t) b->n, o->d_len.as<uint32_t>() + a->n, dBytes.as<uint64_t>());
        if (exclusiveScanU64(st, dBytes.as<uint64_t>(), dNewOff.as<uint64_t>(), b->n, dTmp.p, tmpBytes)) { setError("plasship_seqdb_concat: scan failed"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(concatRankCopyKernel, dim3(gridOf((b->n + 15) / 16, ctx->numCU)), dim3(256), 0, st, b->dataPtr(), b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), b->d_fileRank.as<uint32_t>(), (uint32_t) b->n,
                           dNewOff.as<uint64_t>(), a->dataBytes, (uint32_t) a->n, maxKeyA + 1, o->d_data.as<char>(), o->d_off.as<uint64_t>(), o->d_key.as<uint32_t>());
        PH_CHECK(plasship::streamSync(st));
    }
    o->dbtype = a->dbtype; o->n = (size_t) nn; o->dataBytes = dataBytes; o->residues = a->residues + b->residues;
    o->maxEntryLen = std::max(a->maxEntryLen, b->maxEntryLen); o->hostIndexValid = false;
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    *out = o.release();
    return PLASSHIP_OK;
}
// concatdbs <A> <B> <out> --preserve-keys (DBConcat.cpp:113-118 with preserveKeysB; data/nuclassemble.sh:41,145: the circular contigs taken out
// by cyclecheck rejoin the linear ones under their own keys): the union of the two DBs, every entry under its own key.  A handle holds its entries
// in key order, so the result is the MERGE of the two key-sorted lists — rank of A's entry i = i + |{keys of B < keyA[i]}|, of B's entry j =
// j + |{keys of A <= keyB[j]}| (a binary search per entry) — and the bytes are gathered into that order, back to back like the DB file.
// A key held by both DBs (the reference then writes two entries under one key, which no reader of its own resolves) is refused.
__global__ void mergeRankKernel(const uint32_t *__restrict__ keyA, const uint32_t *__restrict__ lenA, uint32_t nA, const uint32_t *__restrict__ keyB, const uint32_t *__restrict__ lenB, uint32_t nB,
                                uint32_t *__restrict__ rankA, uint32_t *__restrict__ rankB, uint32_t *__restrict__ len, uint32_t *__restrict__ key, uint64_t *__restrict__ bytes, uint32_t *__restrict__ dup) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nA + nB; t += gridDim.x * blockDim.x) {
        const bool fromA = t < nA; const uint32_t i = fromA ? t : t - nA;
        const uint32_t k = fromA ? keyA[i] : keyB[i];
        const uint32_t *o = fromA ? keyB : keyA; uint32_t lo = 0, hi = fromA ? nB : nA;
        while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (fromA ? (o[mid] < k) : (o[mid] <= k)) lo = mid + 1; else hi = mid; }
        if (!fromA && lo > 0 && o[lo - 1] == k) atomicOr(dup, 1u);
        const uint32_t r = i + lo, l = fromA ? lenA[i] : lenB[i];
        (fromA ? rankA : rankB)[i] = r; len[r] = l; key[r] = k; bytes[r] = (uint64_t) l + 2;
    }
}
__global__ __launch_bounds__(256) void mergeCopyKernel(const char *__restrict__ src, const uint64_t *__restrict__ offS, const uint32_t *__restrict__ lenS, const uint32_t *__restrict__ rank, uint32_t n,
                                                       const uint64_t *__restrict__ newOff, char *__restrict__ data) {
    const int G = 16, gl = threadIdx.x & (G - 1);
    for (uint32_t j = blockIdx.x * (256 / G) + threadIdx.x / G; j < n; j += gridDim.x * (256 / G)) {
        const uint32_t r = rank[j], el = lenS[j] + 2; const uint64_t d = newOff[r], s = offS[j];
        for (uint32_t i = gl; i < el; i += G) data[d + i] = src[s + i];
    }
}
extern "C" int plasship_seqdb_concat_keys(plasship_ctx *ctx, const plasship_seqdb *a, const plasship_seqdb *b, int preserve_keys_b, plasship_seqdb **out) {
    if (!preserve_keys_b) return plasship_seqdb_concat(ctx, a, b, out);
    if (!ctx || !a || !b || !out) { setError("plasship_seqdb_concat_keys: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint64_t nn = (uint64_t) a->n + b->n;
    if (nn >= 0xFFFFFFFFull) { setError("plasship_seqdb_concat_keys: too many sequences"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (a->dbtype != b->dbtype) { setError("plasship_seqdb_concat_keys: the DB types differ"); return PLASSHIP_ERR_ARG; }
    std::unique_ptr<plasship_seqdb> packedA, packedB;
    if (!a->contiguous) { const int rcP = packedCopyOf(ctx, a, packedA); if (rcP) return rcP; a = packedA.get(); }
    if (!b->contiguous) { const int rcP = packedCopyOf(ctx, b, packedB); if (rcP) return rcP; b = packedB.get(); }
    const uint64_t dataBytes = a->dataBytes + b->dataBytes;
    std::unique_ptr<plasship_seqdb> o(new plasship_seqdb());
    DevBuf dRankA, dRankB, dBytes, dDup, dTmp; const size_t tmpBytes = exclusiveScanTmpBytes(nn + 2);
    if (o->d_data.allocLong(dataBytes + 64) != hipSuccess || o->d_off.allocLong((nn + 1) * 8) != hipSuccess || o->d_len.allocLong((nn + 1) * 4) != hipSuccess ||
        o->d_key.allocLong((nn + 1) * 4) != hipSuccess || dRankA.alloc(((uint64_t) a->n + 1) * 4) != hipSuccess || dRankB.alloc(((uint64_t) b->n + 1) * 4) != hipSuccess ||
        dBytes.alloc((nn + 1) * 8) != hipSuccess || dDup.alloc(4) != hipSuccess || dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_seqdb_concat_keys: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dDup.p, 0, 4, st));
    PH_CHECK(hipMemsetAsync((char *) o->d_data.p + dataBytes, 0, 64, st));
    if (nn) {
        hipLaunchKernelGGL(mergeRankKernel, dim3(gridOf(nn, ctx->numCU)), dim3(256), 0, st, a->d_key.as<uint32_t>(), a->d_len.as<uint32_t>(), (uint32_t) a->n, b->d_key.as<uint32_t>(), b->d_len.as<uint32_t>(), (uint32_t) b->n,
                           dRankA.as<uint32_t>(), dRankB.as<uint32_t>(), o->d_len.as<uint32_t>(), o->d_key.as<uint32_t>(), dBytes.as<uint64_t>(), dDup.as<uint32_t>());
        if (exclusiveScanU64(st, dBytes.as<uint64_t>(), o->d_off.as<uint64_t>(), nn, dTmp.p, tmpBytes)) { setError("plasship_seqdb_concat_keys: scan failed"); return PLASSHIP_ERR_DEVICE; }
        // (the scan leaves its total at [nn]: o->d_off is the finished index)
        if (a->n) hipLaunchKernelGGL(mergeCopyKernel, dim3(gridOf((a->n + 15) / 16, ctx->numCU)), dim3(256), 0, st, a->dataPtr(), a->d_off.as<uint64_t>(), a->d_len.as<uint32_t>(), (const uint32_t *) dRankA.as<uint32_t>(), (uint32_t) a->n,
                                     (const uint64_t *) o->d_off.as<uint64_t>(), o->d_data.as<char>());
        if (b->n) hipLaunchKernelGGL(mergeCopyKernel, dim3(gridOf((b->n + 15) / 16, ctx->numCU)), dim3(256), 0, st, b->dataPtr(), b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), (const uint32_t *) dRankB.as<uint32_t>(), (uint32_t) b->n,
                                     (const uint64_t *) o->d_off.as<uint64_t>(), o->d_data.as<char>());
    } else PH_CHECK(hipMemsetAsync(o->d_off.p, 0, 8, st));
    uint32_t dup = 0;
    PH_COPY_SYNC(st, &dup, dDup.p, 4, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    if (dup) { setError("plasship_seqdb_concat_keys: a key occurs in both DBs (the reference writes two entries under one key then; not reproduced)"); return PLASSHIP_ERR_UNSUPPORTED; }
    o->dbtype = a->dbtype; o->n = (size_t) nn; o->dataBytes = dataBytes; o->residues = a->residues + b->residues;
    o->maxEntryLen = std::max(a->maxEntryLen, b->maxEntryLen); o->hostIndexValid = false;
    *out = o.release();
    return PLASSHIP_OK;
}
extern "C" int plasship_orfhdr_concat(plasship_ctx *ctx, const plasship_orfhdr *a, const plasship_orfhdr *b, plasship_orfhdr **out) {
    if (!ctx || !a || !b || !out) { setError("plasship_orfhdr_concat: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    uint32_t maxKeyA = 0;
    if (a->n) { OrfInfo last; PH_COPY_SYNC(st, &last, a->d_info.as<OrfInfo>() + (a->n - 1), sizeof(OrfInfo), hipMemcpyDeviceToHost); maxKeyA = last.key; }
    const uint64_t nn = (uint64_t) a->n + b->n;
    std::unique_ptr<plasship_orfhdr> o(new plasship_orfhdr());
    if (o->d_info.alloc((nn + 1) * sizeof(OrfInfo)) != hipSuccess) { setError("plasship_orfhdr_concat: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (nn) hipLaunchKernelGGL(concatInfoKernel, dim3(gridOf(nn, ctx->numCU)), dim3(256), 0, st, a->d_info.as<OrfInfo>(), (uint32_t) a->n, b->d_info.as<OrfInfo>(), (uint32_t) b->n, (const uint32_t *) b->d_fileRank.as<uint32_t>(), maxKeyA + 1, o->d_info.as<OrfInfo>());
    o->n = (size_t) nn; o->nUnparsable = a->nUnparsable + b->nUnparsable;
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    *out = o.release();
    return PLASSHIP_OK;
}

// ---- header DB <-> files (Orf::writeOrfHeader / Orf::parseOrfHeader, Orf.cpp:350-456) ------------------------------------------
extern "C" int plasship_orfhdr_write(plasship_ctx *ctx, const plasship_orfhdr *h, const char *db_path) {
    if (!ctx || !h || !db_path) { setError("plasship_orfhdr_write: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    std::vector<OrfInfo> info(h->n);
    { const int rc = stagedCopyToHost(ctx, info.data(), h->d_info.p, h->n * sizeof(OrfInfo)); if (rc) return rc; }
    std::vector<uint32_t> keys(h->n);
    std::atomic<int> foreign(0);
    parallelRanges(h->n, [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) { keys[i] = info[i].key; if (info[i].flags & 4u) foreign = 1; } });
    if (foreign) { setError("plasship_orfhdr_write: the header DB holds entries that are not ORF headers"); return PLASSHIP_ERR_UNSUPPORTED; }
    std::string err;
    const bool ok = writeTextDB(db_path, 12, keys.data(), h->n, nullptr, [&](size_t i, std::string &out) {                    // DBTYPE_GENERIC_DB
        const OrfInfo &o = info[i]; char buf[96];
        char *q = fmtU32(o.readKey, buf); *q++ = '\t'; q = fmtU32(o.fromPos, q); *q++ = (o.fromPos < o.toPos) ? '+' : '-';
        const int d = (int) o.fromPos - (int) o.toPos; q = fmtI32(d < 0 ? -d : d, q);
        if (o.flags & 3u) { *q++ = '\t'; q = fmtI32((int) (o.flags & 3u), q); }
        *q++ = '\n';
        out.append(buf, (size_t) (q - buf));
        return true;
    }, err);
    if (!ok) { setError(err); return PLASSHIP_ERR_IO; }
    return PLASSHIP_OK;
}
extern "C" int plasship_orfhdr_read(plasship_ctx *ctx, const char *db_path, plasship_orfhdr **out) {
    if (!ctx || !db_path || !out) { setError("plasship_orfhdr_read: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    HostDB h; std::string err;
    if (!readDBFiles(db_path, h, err)) { setError(err); return PLASSHIP_ERR_IO; }
    const size_t n = h.key.size();
    std::vector<uint32_t> perm(n); for (size_t i = 0; i < n; i++) perm[i] = (uint32_t) i;
    if (!std::is_sorted(h.key.begin(), h.key.end())) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return h.key[x] < h.key[y]; });
    std::vector<OrfInfo> info(n); std::atomic<size_t> nBadAll(0);
    parallelRanges(n, [&](int, size_t rb, size_t re) {
    size_t nBad = 0;
    for (size_t i = rb; i < re; i++) {
        const uint32_t s = perm[i];
        OrfInfo o; o.key = h.key[s]; o.readKey = 0; o.fromPos = 0; o.toPos = 0; o.flags = 4u;
        // "<id>\t<from>[+-]<len>[\t<flags>]": words are separated by blanks / tabs; the flags count only as the third and last word
        const char *p = h.data.data() + h.off[s], *e = p + (h.elen[s] ? h.elen[s] - 1 : 0);
        const char *w[4] = {nullptr, nullptr, nullptr, nullptr}; size_t wl[4] = {0, 0, 0, 0}; int nw = 0;
        while (p < e && *p != '\n' && *p != '\0') {
            while (p < e && (*p == ' ' || *p == '\t')) p++;
            if (p >= e || *p == '\n' || *p == '\0') break;
            const char *b = p; while (p < e && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\0') p++;
            if (nw < 4) { w[nw] = b; wl[nw] = (size_t) (p - b); }
            nw++;
        }
        if (nw >= 2) {
            size_t q = 0; uint64_t from = 0, len = 0;
            while (q < wl[1] && w[1][q] >= '0' && w[1][q] <= '9') { from = from * 10 + (uint64_t) (w[1][q] - '0'); q++; }
            if (q > 0 && q < wl[1] && (w[1][q] == '+' || w[1][q] == '-')) {
                const bool plus = w[1][q] == '+'; q++; const size_t q0 = q;
                while (q < wl[1] && w[1][q] >= '0' && w[1][q] <= '9') { len = len * 10 + (uint64_t) (w[1][q] - '0'); q++; }
                if (q > q0) {
                    uint64_t id = 0; for (size_t j = 0; j < wl[0] && w[0][j] >= '0' && w[0][j] <= '9'; j++) id = id * 10 + (uint64_t) (w[0][j] - '0');
                    o.readKey = (uint32_t) id; o.fromPos = (uint32_t) from; o.toPos = (uint32_t) (plus ? from + len : from - len); o.flags = 0;
                    if (nw == 3) { uint64_t c = 0; for (size_t j = 0; j < wl[2] && w[2][j] >= '0' && w[2][j] <= '9'; j++) c = c * 10 + (uint64_t) (w[2][j] - '0'); if (c <= 3) o.flags = (uint32_t) c; }
                }
            }
        }
        info[i] = o;
        if (o.flags & 4u) nBad++;
    }
    nBadAll += nBad;
    });
    const size_t nBad = nBadAll;
    std::unique_ptr<plasship_orfhdr> o(new plasship_orfhdr());
    if (o->d_info.alloc((n + 1) * sizeof(OrfInfo)) != hipSuccess) { setError("plasship_orfhdr_read: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    { const int rc = stagedCopyToDevice(ctx, o->d_info.p, info.data(), n * sizeof(OrfInfo)); if (rc) return rc; }
    {   // file order != key order (a DB several writer threads left behind): every entry's rank in the data file (as plasship_seqdb_upload does)
        bool fileSorted = true;
        for (size_t i = 1; i < n && fileSorted; i++) fileSorted = h.off[perm[i - 1]] <= h.off[perm[i]];
        if (!fileSorted) {
            std::vector<uint32_t> byOff(n), rank(n);
            for (size_t i = 0; i < n; i++) byOff[i] = (uint32_t) i;
            std::stable_sort(byOff.begin(), byOff.end(), [&](uint32_t x, uint32_t y) { return h.off[perm[x]] < h.off[perm[y]]; });
            for (size_t r = 0; r < n; r++) rank[byOff[r]] = (uint32_t) r;
            if (o->d_fileRank.allocLong(n * 4) != hipSuccess) { setError("plasship_orfhdr_read: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            const int rc = stagedCopyToDevice(ctx, o->d_fileRank.p, rank.data(), n * 4); if (rc) return rc;
        }
    }
    o->n = n; o->nUnparsable = nBad;
    *out = o.release();
    return PLASSHIP_OK;
}
extern "C" int plasship_orfhdr_count(const plasship_orfhdr *h, size_t *n) {
    if (!h || !n) { setError("plasship_orfhdr_count: bad argument"); return PLASSHIP_ERR_ARG; }
    *n = h->n; return PLASSHIP_OK;
}
extern "C" void plasship_orfhdr_free(plasship_ctx *ctx, plasship_orfhdr *h) {
    if (!h) return;
    if (ctx) { (void) hipSetDevice(ctx->device); plasship::poolEnter(ctx->stream); }
    delete h;
}
