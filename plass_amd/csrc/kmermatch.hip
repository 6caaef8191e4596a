// plasship: kmermatcher on gfx950 (rows K1–K8 of SURVEY.md §8a).  Product code.
//
// Reference behaviour reproduced (file:line in /root/reference/lib/mmseqs/src):
//   linclust/kmermatcher.cpp:77-385    fillKmerPositionArray: letter mapping, k-mer index, XXH64 score,
//                                      per-sequence selection of the lowest-hash k-mers, identity record
//   linclust/kmermatcher.cpp:408-412   sort #1 (kmer, seqLen desc, id, pos)
//   linclust/kmermatcher.cpp:450-559   assignGroup: rep = first of each equal-kmer run, diagonal, filter
//   linclust/kmermatcher.cpp:427-431   sort #2 (rep, target, diagonal)
//   linclust/kmermatcher.cpp:809-924   writeKmerMatcherResult: best diagonal per (rep, target)
//   linclust/kmermatcher.cpp:705-724   every key gets an entry ("key\t0\t0" self line first)
//
// MI355X design (not a translation of the CPU algorithm):
//   * extraction: one wavefront per sequence, codes staged through LDS in 64-position tiles; the
//     reference's 65 536-bin threshold walk becomes a two-level 256-bin LDS radix select, the per
//     sequence std::sort becomes an LDS bitonic sort of only the <= ~60 candidate k-mers.
//   * sort #1 is NOT a sort: only grouping by equal k-mer and the identity of the run's first record
//     matter (its order is destroyed by sort #2 anyway).  Records are hash-partitioned over the line
//     store (linepart.hpp: 1–2 levels, unstable, no histogram pass) into buckets that fit an LDS hash
//     table; the run head is one atomicMin over a packed (seqLen desc, id, pos, strand) word.  One
//     read+write per level instead of the 8+ passes of an LSD radix sort over 16-byte records.
//   * sort #2 must be a true sort (the reference scans across rep boundaries, Appendix A.3): records
//     are range-partitioned by rep id (order preserving) and each bucket is bitonic-sorted in LDS.
//   * per-(rep,target) reduction: one thread per run head walks its run.
// Integer work only; the single float expression (Util::canBeCovered) is IEEE-exact.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include "linepart.hpp"
#include <algorithm>
#include <memory>
#include <type_traits>
#include <cstdlib>
#include <climits>
#include <cstring>

namespace plasship {

// ---- XXH64 of one little-endian u64 (xxhash 0.8.0, call site kmermatcher.cpp:33-38) ---------------------
__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ __forceinline__ uint64_t xxh64U64(uint64_t v, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = rotl64(v * P2, 31) * P1;
    h ^= k1;
    h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}
// 2-bit alphabet A0 C1 T2 G3, complement = code ^ 2 (Util.cpp:601-638)
__device__ __forceinline__ uint64_t revComplementDev(uint64_t kmer, int k) {
    uint64_t x = kmer ^ 0xAAAAAAAAAAAAAAAAULL;                         // complement every 2-bit letter
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);                                          // reverse the 32 letters
    return x >> (64 - 2 * k);
}

// =====================================================================================================
// 1. slot bounds (computeKmerCount, kmermatcher.cpp:576-585): every sequence owns a fixed slot range
//    of the record array, exactly like the reference's pre-sized array; unused slots keep 0xFF.
// =====================================================================================================
__global__ void boundsKernel(const uint32_t *__restrict__ len, uint32_t n, int k, int kps, float scale, uint32_t *__restrict__ bound) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int L = (int) len[i];
        const int adj = max(1, L - k + 2);
        bound[i] = (uint32_t) min(adj, (int) ((float) (size_t) kps + (scale * (float) L)));
    }
}

// =====================================================================================================
// 2. extraction + selection, one wavefront (= one 64-thread block) per sequence
// =====================================================================================================
struct Cand { uint64_t kmer; uint32_t pos; uint32_t score; };   // score: low 16 bits hash, bit 31 = skipped
// selected-window cache (section 2c): one 128-byte line per sequence = {u64 identity hash BEFORE its XXH64 (Util::hash of the letter
// codes: seed-independent), u16 position of up to 59 selected windows (0xFFFF = none), u16 flags}
constexpr uint32_t KMC_LINE = 128, KMC_POS = 59, KMC_FLAGS = 63, KMC_CLEAN = 1;     // flags word at u16 index 63; CLEAN: every candidate was selected (unordered path)

#define FALLBACK_NOSTATS(a) ((a).kstats == nullptr)       // the scratch launch that re-extracts one sequence for the stale-record check: no statistics, no cache
struct ExtractArgs {
    SeqView s;
    const uint64_t *slotOff;        // [n+1]
    void *arr;                      // Rec<LONG>[total]
    const unsigned char *map;       // 256-entry letter -> code
    uint64_t powers[24];            // AA: (alphabet-1)^i
    int k, xCode, kps, ignoreMulti;
    float scale;
    uint64_t seed;
    uint32_t *overflowIds, *overflowCount;
    // fallback launch: explicit id list and per-sequence global scratch
    const uint32_t *idList; uint32_t nIds; Cand *scratch; const uint64_t *scratchOff; const uint32_t *scratchCap;
    // regular launch after the one-thread-per-sequence kernel: only the queued ids (count read on the device)
    const uint32_t *waveList; const uint32_t *waveCount;
    unsigned long long *kstats;     // [2] residues, [3] records handled by the wave-per-sequence kernel (incl. its HBM-scratch launch)
    uint64_t slotBias;              // subtracted from every slot offset (re-extraction of one sequence into a scratch array;
                                    // sharded run: first slot of this rank's id range)
    uint32_t idLo, idHi;            // regular launch without a wave list: ids [idLo, idHi) (sharded run: this rank's share)
    // One more input travels OUTSIDE this struct — the tuned tiers are at the edge of their register budgets (round 4: a block of code
    // that filled an overflowing sequence's slots with sentinels cost the 4-scores tier 40 bytes of scratch per lane and 8-13 ms per
    // iteration; that fill is a kernel of its own now, fillOverflowSlotsKernel):
    //   kstats[4] = address of the selected-window cache lines (section 2c; 0 = no cache), fetched per sequence where it is used.
};

__device__ __forceinline__ bool candLess(const Cand &a, const Cand &b, bool nucl) {
    const uint32_t sa = a.score, sb = b.score;
    if (sa != sb) return sa < sb;
    const uint64_t ka = nucl ? (a.kmer | BIT63) : a.kmer, kb = nucl ? (b.kmer | BIT63) : b.kmer;
    if (ka != kb) return ka < kb;
    return a.pos < b.pos;
}

// bitonic sort of p[0..P) (P power of two) by one wavefront; padding entries carry score 0xFFFFFFFF
__device__ void waveBitonicSortCands(Cand *p, uint32_t P, bool nucl) {
    for (uint32_t kk = 2; kk <= P; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += 64) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    Cand a = p[i], b = p[l];
                    const bool up = (i & kk) == 0;
                    if (candLess(b, a, nucl) == up) { p[i] = b; p[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// k-mer of the window starting at p from a code accessor (Indexer::int2index / computeKmerIdx + canonical strand,
// kmermatcher.cpp:149-213); returns false for windows containing X or (nucleotide) reverse-palindromes
template <bool NUCL, class F>
__device__ __forceinline__ bool kmerFromCodes(F codeAt, int k, unsigned char xCode, const uint64_t *powers, uint32_t L, uint32_t p,
                                              uint64_t &kmer, uint32_t &pos) {
    bool hasX = false; kmer = 0; pos = p;
    if (NUCL) {
        uint64_t f = 0;
        for (int i = 0; i < k; i++) { const unsigned char ci = codeAt(i); hasX |= (ci == xCode); f = (f << 2) | (ci & 3); }
        const uint64_t r = revComplementDev(f, k);
        if (hasX || r == f) return false;
        const bool pickRev = r < f;
        const uint64_t cc = pickRev ? r : f;
        kmer = pickRev ? cc : (cc | BIT63);
        pos = pickRev ? (L - p - k) : p;
        return true;
    }
    for (int i = 0; i < k; i++) { const unsigned char ci = codeAt(i); hasX |= (ci == xCode); kmer += (uint64_t) ci * powers[i]; }
    return !hasX;
}

// Protein k-mer index straight from the code bytes in LDS (Indexer::int2index, mm/prefiltering/Indexer.h:20-83: sum code[i] *
// base^i): two unaligned 8-byte LDS reads fetch all k <= 14 codes, an 'X' is found with a SWAR zero-byte test, the sum is two
// 24-bit Horner halves (base <= 16 keeps every partial sum below 2^24) joined by one 32x32->64 multiply-add.  The window
// loop is issue bound; this replaces 14 LDS byte reads and 14 64-bit multiply-adds per window.
__device__ __forceinline__ bool kmerIndexCore(uint64_t w0, uint64_t w1, int k, unsigned xCode, uint32_t base, uint32_t base7, uint64_t &kmer);
__device__ __forceinline__ bool kmerIndexFast(const unsigned char *w, int k, unsigned xCode, uint32_t base, uint32_t base7, uint64_t &kmer) {
    uint64_t w0, w1; __builtin_memcpy(&w0, w, 8); __builtin_memcpy(&w1, w + 8, 8);
    return kmerIndexCore(w0, w1, k, xCode, base, base7, kmer);
}
// the same from a 4-byte-aligned LDS array: five aligned dword reads and four v_alignbyte_b32 fetch the 16 codes of the window at
// byte p (a byte pointer makes the compiler read LDS byte by byte: 16 ds_read_u8 and their shifts per window)
__device__ __forceinline__ bool kmerIndexFastAligned(const unsigned char *codes, uint32_t p, int k, unsigned xCode, uint32_t base, uint32_t base7, uint64_t &kmer) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(codes + (p & ~3u));
    const uint32_t sh = p & 3u;
    const uint32_t W0 = w[0], W1 = w[1], W2 = w[2], W3 = w[3], W4 = w[4];
    const uint32_t d0 = __builtin_amdgcn_alignbyte(W1, W0, sh), d1 = __builtin_amdgcn_alignbyte(W2, W1, sh),
                   d2 = __builtin_amdgcn_alignbyte(W3, W2, sh), d3 = __builtin_amdgcn_alignbyte(W4, W3, sh);
    return kmerIndexCore((uint64_t) d0 | ((uint64_t) d1 << 32), (uint64_t) d2 | ((uint64_t) d3 << 32), k, xCode, base, base7, kmer);
}
__device__ __forceinline__ bool kmerIndexCore(uint64_t w0, uint64_t w1, int k, unsigned xCode, uint32_t base, uint32_t base7, uint64_t &kmer) {
    const uint64_t m0 = (k >= 8) ? ~0ULL : ((1ULL << (8 * k)) - 1ULL);
    const uint64_t m1 = (k <= 8) ? 0ULL : ((1ULL << (8 * (k - 8))) - 1ULL);       // k <= 14
    w0 &= m0; w1 &= m1;
    const uint64_t ones = 0x0101010101010101ULL, highs = 0x8080808080808080ULL, xs = ones * xCode;
    const uint64_t x0 = w0 ^ xs, x1 = w1 ^ xs;
    const uint64_t z = (((x0 - ones) & ~x0 & highs) & m0) | (((x1 - ones) & ~x1 & highs) & m1);
    const uint32_t a0 = (uint32_t) w0, a1 = (uint32_t) (w0 >> 32), b0 = (uint32_t) w1, b1 = (uint32_t) (w1 >> 32);
    // codes c0..c6 = bytes 0..6 of w0; c7 = byte 7 of w0; c8..c13 = bytes 0..5 of w1
    uint32_t lo = (a1 >> 16) & 0xFFu;                                   // c6
    lo = __umul24(lo, base) + ((a1 >> 8) & 0xFFu);                      // c5
    lo = __umul24(lo, base) + (a1 & 0xFFu);                             // c4
    lo = __umul24(lo, base) + (a0 >> 24);                               // c3
    lo = __umul24(lo, base) + ((a0 >> 16) & 0xFFu);                     // c2
    lo = __umul24(lo, base) + ((a0 >> 8) & 0xFFu);                      // c1
    lo = __umul24(lo, base) + (a0 & 0xFFu);                             // c0
    uint32_t hi = (b1 >> 8) & 0xFFu;                                    // c13
    hi = __umul24(hi, base) + (b1 & 0xFFu);                             // c12
    hi = __umul24(hi, base) + (b0 >> 24);                               // c11
    hi = __umul24(hi, base) + ((b0 >> 16) & 0xFFu);                     // c10
    hi = __umul24(hi, base) + ((b0 >> 8) & 0xFFu);                      // c9
    hi = __umul24(hi, base) + (b0 & 0xFFu);                             // c8
    hi = __umul24(hi, base) + (a1 >> 24);                               // c7
    kmer = (uint64_t) lo + (uint64_t) hi * (uint64_t) base7;
    return z == 0;
}

// Nucleotide k-mer (Indexer::computeKmerIdx, mm/prefiltering/Indexer.h:124-131: 2 bits per letter, first letter most significant)
// straight from the code bytes in LDS: up to four unaligned 8-byte reads fetch the k <= 31 codes (0..3, X = 4), an X is any byte
// with bit 2 set, and each word's eight 2-bit fields are gathered with three shift-or-mask steps after a byte swap.  Replaces k
// LDS byte reads and k shift/or pairs per window.
__device__ __forceinline__ uint32_t pack2x8(uint64_t w) {            // bytes b0..b7 (codes, b0 first) -> 16 bits, b0 most significant
    uint64_t y = __builtin_bswap64(w & 0x0303030303030303ULL);
    y = (y | (y >> 6)) & 0x000F000F000F000FULL;
    y = (y | (y >> 12)) & 0x000000FF000000FFULL;
    y = (y | (y >> 24)) & 0xFFFFULL;
    return (uint32_t) y;
}
__device__ __forceinline__ bool kmerNuclFast(const unsigned char *w, int k, uint64_t &f) {
    uint64_t x = 0; f = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (8 * j < k) {
            uint64_t v; __builtin_memcpy(&v, w + 8 * j, 8);
            const int nb = k - 8 * j;                               // codes of this word that belong to the k-mer
            if (nb < 8) v &= (1ULL << (8 * nb)) - 1ULL;
            x |= v;
            const int sh = 2 * (k - 8 * (j + 1));
            const uint64_t g = pack2x8(v);
            f |= (sh >= 0) ? (g << sh) : (g >> (-sh));
        }
    }
    return (x & 0x0404040404040404ULL) == 0;                        // no X among the k codes
}

// canonical strand, palindromes dropped, position mirrored for the reverse strand (kmermatcher.cpp:149-187)
__device__ __forceinline__ bool kmerNuclCanonical(const unsigned char *w, int k, uint32_t L, uint32_t p, uint64_t &kmer, uint32_t &pos) {
    uint64_t f;
    const bool noX = kmerNuclFast(w, k, f);
    const uint64_t r = revComplementDev(f, k);
    kmer = 0; pos = p;
    if (!noX || r == f) return false;
    const bool pickRev = r < f;
    kmer = pickRev ? r : (f | BIT63);
    pos = pickRev ? (L - p - k) : p;
    return true;
}


// REGS > 0 (the regular launch): sequences of up to 64 * REGS windows — every read, every contig up to ~1000 residues — keep the
//   16-bit score of every window in a REGISTER (window p = j * 64 + lane): the k-mers are hashed once, the reference's
//   65 536-bin threshold walk is a 16-step bisection over the score bits whose counts are wave ballots (no LDS histogram, no
//   atomics, no barriers), and only the <= ~60 selected windows rebuild their k-mer.  Longer sequences are queued for the next
//   launch.  REGS == 0: the three-pass path below; RESL = longest sequence whose codes and scores stay resident in LDS.
//   Wavefronts per SIMD (amdgpu_waves_per_eu), measured with tools/extract_probe.py: the per-sequence phases are chains of LDS round
//   trips, so resident wavefronts count for more than registers — 6 for the 4-scores tier (80 VGPRs; 4: +30 % time; 8 would gain
//   another 4 % but its 144 bytes of scratch per lane turn into 90 GB of memory traffic per launch), 4 for the 16-scores tier (5 gains nothing), 4 for the 48-scores tier of protein runs (128 VGPRs
//   and 200+ bytes of scratch, yet 3.0 instead of 4.7 ms per 120 k sequences of 2500 residues at 2 wavefronts).
//   Round 3 (kernel-resource-usage remarks of the compiler + an A/B run, profiles/r03_ab_tier0_wpe.log): at 6 wavefronts the 4-scores tier
//   has 80 VGPRs and spills 17 of them (64 bytes of scratch per lane — half of the tier's HBM traffic in round 2's PMC pass); at 5 it has
//   96, spills one, and the wave-per-sequence extraction of the 50 M-read chain is 9 % faster (85.8 -> 78.0 ms per iteration).
template <bool NUCL, bool LONG, int CAP, bool FALLBACK, int REGS = 0, int RESL = 992, int WPE = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE ? WPE : ((REGS > 0 && REGS <= 4) ? 5 : (REGS == 16 ? 4 : ((REGS > 16 && !NUCL) ? 4 : 1)))))) void extractKernel(ExtractArgs a) {
    constexpr uint32_t RES_L = RESL;
    constexpr uint32_t CODES = (RESL > 64 * REGS + 32 ? RESL : 64 * REGS + 32) + 32;
    __shared__ unsigned char sMap[256];
    __shared__ unsigned char sCode[64 + 32];
    __shared__ uint32_t sHist[(REGS > 0 && !FALLBACK) ? 1 : 256];  // radix select of the three-pass path only
    __shared__ Cand sCand[FALLBACK ? 1 : CAP];
    __shared__ unsigned long long sSet[FALLBACK ? 1 : 2 * CAP];     // duplicate-k-mer detection without sorting (after the passes)
    __shared__ unsigned short sScoreBig[(RESL > 4 * CAP && !FALLBACK) ? RESL : 1];
    unsigned short *sScore = (RESL > 4 * CAP) ? sScoreBig : reinterpret_cast<unsigned short *>(sSet); // per-window hash scores (during the passes; aliases the set when 2*CAP*8 >= RESL*2 bytes)
    __shared__ __attribute__((aligned(16))) unsigned char sCodeAll[FALLBACK ? 1 : CODES];          // codes of a resident sequence
    __shared__ unsigned long long sPow64[(REGS > 0 && !FALLBACK) ? REGS + 2 : 1];                  // 31^(64 q) (identity hash of the register front end)
    __shared__ unsigned long long sValid[RESL / 64 + 2];             // per-tile validity masks
    __shared__ unsigned short sKmcPos[64];                           // positions of the ordered path's selection, for the cache line (section 2c)
    typedef Rec<LONG> R;
    R *arr = reinterpret_cast<R *>(a.arr);
    const int lane = threadIdx.x;
    const int k = a.k;
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    __syncthreads();
    const bool fastIdx = !NUCL && k <= 14 && a.powers[1] <= 16;      // see kmerIndexFast
    uint64_t pow31 = 1;                                    // 31^lane
    unsigned long long stRes = 0, stRec = 0;
    for (int i = 0; i < lane; i++) pow31 *= 31;
    // register front end: 31^(63 - lane), the inverse of 31^64 modulo 2^64 and 31^(64 q) — the identity hash
    // h = sum code[p] * 31^(L-1-p) is then one running product per lane and ONE wave reduction per sequence
    uint64_t pow31rev = 0, inv64 = 0;
    if (REGS > 0 && !FALLBACK) {
        pow31rev = __shfl(pow31, 63 - lane, 64);
        const uint64_t p64 = __shfl(pow31, 63, 64) * 31ull;                       // 31^64 (odd: invertible mod 2^64)
        uint64_t iv = p64; for (int i = 0; i < 6; i++) iv *= 2ull - p64 * iv;     // Newton: doubles the correct low bits each step
        inv64 = iv;
        if (lane == 0) { uint64_t t = 1; for (int q = 0; q < REGS + 2; q++) { sPow64[q] = t; t *= p64; } }
        __syncthreads();
    }

    const uint32_t nWork = FALLBACK ? a.nIds : (a.waveList ? *a.waveCount : (a.idHi - a.idLo));
    auto idAt = [&](uint32_t w) { return a.waveList ? a.waveList[w] : (a.idLo + w); };
    // software pipeline over sequences (regular launch): the index entry of sequence w+2*grid and the first 128 bytes
    // of sequence w+grid are in flight while sequence w is processed, so a short read never waits on HBM latency
    struct Meta { uint32_t L; uint64_t off, slot, slot1; };
    auto loadMeta = [&](uint32_t id) { Meta m; m.L = a.s.len[id]; m.off = a.s.off[id]; m.slot = a.slotOff[id] - a.slotBias; m.slot1 = a.slotOff[id + 1] - a.slotBias; return m; };
    Meta mNext = {0, 0, 0, 0}, mNext2 = {0, 0, 0, 0};
    char pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;       // the next sequence's first 256 bytes (the average contig of a metagenomic run is ~250 residues)
    if (!FALLBACK && blockIdx.x < nWork) {
        mNext = loadMeta(idAt(blockIdx.x));
        if (blockIdx.x + gridDim.x < nWork) mNext2 = loadMeta(idAt(blockIdx.x + gridDim.x));
        if ((uint32_t) lane < mNext.L) pb0 = a.s.data[mNext.off + lane];
        if ((uint32_t) lane + 64 < mNext.L) pb1 = a.s.data[mNext.off + lane + 64];
        if ((uint32_t) lane + 128 < mNext.L) pb2 = a.s.data[mNext.off + lane + 128];
        if ((uint32_t) lane + 192 < mNext.L) pb3 = a.s.data[mNext.off + lane + 192];
    }
    for (uint32_t w = blockIdx.x; w < nWork; w += gridDim.x) {
        const uint32_t id = FALLBACK ? a.idList[w] : idAt(w);
        Meta cur;
        char cb0 = 0, cb1 = 0, cb2 = 0, cb3 = 0;
        if (FALLBACK) cur = loadMeta(id);
        else {
            cur = mNext; cb0 = pb0; cb1 = pb1; cb2 = pb2; cb3 = pb3;
            mNext = mNext2;
            if (w + gridDim.x < nWork) {
                pb0 = ((uint32_t) lane < mNext.L) ? a.s.data[mNext.off + lane] : (char) 0;
                pb1 = ((uint32_t) lane + 64 < mNext.L) ? a.s.data[mNext.off + lane + 64] : (char) 0;
                pb2 = ((uint32_t) lane + 128 < mNext.L) ? a.s.data[mNext.off + lane + 128] : (char) 0;
                pb3 = ((uint32_t) lane + 192 < mNext.L) ? a.s.data[mNext.off + lane + 192] : (char) 0;
            }
            if (w + 2 * gridDim.x < nWork) mNext2 = loadMeta(idAt(w + 2 * gridDim.x));
        }
        const uint32_t L = cur.L;
        const char *base = a.s.data + cur.off;
        const uint64_t slot = cur.slot;
        const uint32_t bound = (uint32_t) (cur.slot1 - slot);
        Cand *cand = FALLBACK ? (a.scratch + a.scratchOff[w]) : sCand;
        const uint32_t cap = FALLBACK ? a.scratchCap[w] : (uint32_t) CAP;
        const uint32_t nWin = (L >= (uint32_t) k) ? (L - k + 1) : 0;
        const size_t consideredRaw = (size_t) ((float) (a.kps - 1) + (a.scale * (float) L));   // kmermatcher.cpp:223
        const bool allCand = (size_t) nWin <= consideredRaw;
        if (!FALLBACK && std::min((size_t) nWin, consideredRaw) > (size_t) cap) {       // cannot fit this instantiation's LDS: next tier
            if (lane == 0) { const uint32_t o = atomicAdd(a.overflowCount, 1u); a.overflowIds[o] = id; }
            continue;
        }

        uint32_t C = 0;            // candidates pushed (wave-uniform)
        uint32_t n = 0;            // valid k-mers
        bool overflow = false;
        uint64_t seqHash = 0;      // Util::hash (Util.h:337-345): h = h*31 + code
        uint32_t sStar = 0; int tooMuch = 0; size_t considered = 0;
        uint32_t b1 = 0, cumBefore1 = 0;

        constexpr bool useRegs = REGS > 0 && !FALLBACK;
        // Code-generation aid, not logic: phaseSplit is a wave-uniform value that is ALWAYS ZERO (a letter code is below 2^30), which
        // the compiler cannot know.  The never-taken uniform branches on it end the scheduling region between the phases of the
        // register front end (identity hash / window hashing of a row / bisection / candidate rebuild); without them the scheduler
        // merges the phases, and the 16-scores tier measures 8-14 % slower (300 k sequences of 1000 residues: 2.85 instead of 2.61 ms;
        // 750 k of 400: 4.03 instead of 3.47 ms — `__builtin_amdgcn_sched_barrier` at the same places does not have that effect).
        const uint32_t phaseSplit = (uint32_t) a.xCode >> 30;
        if (useRegs && nWin > 64u * (uint32_t) (REGS > 0 ? REGS : 1)) {     // too long for the register front end: next tier
            if (lane == 0) { const uint32_t o = atomicAdd(a.overflowCount, 1u); a.overflowIds[o] = id; }
            continue;
        }
        if (useRegs) {
            // ---- codes to LDS (padded with X so that every window read stays inside the staged bytes) ----
            for (uint32_t i = lane; i < L + 31; i += 64) {
                const char ch = (i < 64) ? cb0 : ((i < 128) ? cb1 : ((i < 192) ? cb2 : ((i < 256) ? cb3 : ((i < L) ? base[i] : (char) 0))));      // first 256 bytes were prefetched
                sCodeAll[i] = (i < L) ? sMap[(unsigned char) ch] : (unsigned char) a.xCode;
            }
            __syncthreads();
            // identity hash (Util::hash, Util.h:337-345: h = h*31 + code, i.e. sum code[p] * 31^(L-1-p) modulo 2^64): lane l owns the
            // positions l, l + 64, …; its power starts at 31^(L-1-l) and shrinks by 31^64 (a multiplication by the inverse) per step
            if (!(phaseSplit & 1)) {
                uint64_t pw;
                if (L >= 64) { const uint32_t e = L - 64; pw = sPow64[e >> 6] * __shfl(pow31, (int) (e & 63u), 64) * pow31rev; }
                else pw = ((uint32_t) lane < L) ? __shfl(pow31, (int) (L - 1 - min((uint32_t) lane, L - 1)), 64) : 0ull;
                uint64_t acc = 0;
                for (uint32_t t0 = 0; t0 < L; t0 += 64) {
                    const uint32_t p = t0 + lane;
                    if (p < L) acc += (uint64_t) sCodeAll[p] * pw;
                    pw *= inv64;
                }
                seqHash = waveReduceSumU64(acc);
            }
            auto windowKmer = [&](uint32_t p, uint64_t &kmer, uint32_t &pos) -> bool {
                pos = p; kmer = 0;
                if (!NUCL && fastIdx) return kmerIndexFastAligned(sCodeAll, p, k, (unsigned) a.xCode, (uint32_t) a.powers[1], (uint32_t) a.powers[7], kmer);
                if (NUCL && a.xCode == 4) return kmerNuclCanonical(&sCodeAll[p], k, L, p, kmer, pos);
                return kmerFromCodes<NUCL>([&](int i) { return sCodeAll[p + i]; }, k, (unsigned char) a.xCode, a.powers, L, p, kmer, pos);
            };
            uint32_t *sPick = reinterpret_cast<uint32_t *>(sSet);     // [cap] (score << 16 | window) of the candidates; the set is not in use yet
            // ---- one score per window, in registers: 0xFFFFFFFF = no k-mer here ----
            const uint32_t nWinU = (uint32_t) __builtin_amdgcn_readfirstlane((int) nWin);     // wave-uniform loop guards stay scalar
            uint32_t sc[REGS > 0 ? REGS : 1];
#pragma unroll
            for (int j = 0; j < REGS; j++) {
                sc[j] = 0xFFFFFFFFu;
                if ((uint32_t) j * 64u < nWinU) {                   // wave-uniform
                    const uint32_t p = (uint32_t) j * 64u + (uint32_t) lane;
                    if (p < nWin) {
                        uint64_t kmer; uint32_t pos;
                        if (phaseSplit & 32) sc[j] = p;
                        else if (windowKmer(p, kmer, pos)) sc[j] = (uint32_t) (xxh64U64(NUCL ? (kmer & ~BIT63) : kmer, a.seed) & 0xFFFFu);
                    }
                    n += (uint32_t) __popcll(__ballot(sc[j] != 0xFFFFFFFFu));
                }
            }
            considered = min(consideredRaw, (size_t) n);
            if (!allCand && considered > 0) {
                // the reference walks 65 536 score bins until `considered` k-mers are covered (kmermatcher.cpp:224-239): s* is the
                // considered-th smallest score = the largest t with fewer than `considered` scores below it
                uint32_t t = 0;
                if (phaseSplit & 2) t = 1;
                else
#pragma unroll 1
                for (int bit = 15; bit >= 0; bit--) {
                    const uint32_t tr = t | (1u << bit);
                    uint32_t below = 0;
#pragma unroll
                    for (int j = 0; j < REGS; j++) if ((uint32_t) j * 64u < nWinU) below += (uint32_t) __popcll(__ballot(sc[j] < tr));
                    if ((size_t) below < considered) t = tr;
                }
                sStar = t;
                uint32_t upTo = 0;
#pragma unroll
                for (int j = 0; j < REGS; j++) if ((uint32_t) j * 64u < nWinU) upTo += (uint32_t) __popcll(__ballot(sc[j] <= t));
                tooMuch = (int) upTo - (int) considered;
            }
            // ---- candidates: every k-mer (allCand) or those with score <= s* ----
            if (allCand || considered > 0) {
#pragma unroll
                for (int j = 0; j < REGS; j++) {
                    if ((uint32_t) j * 64u < nWinU) {
                        const bool push = sc[j] != 0xFFFFFFFFu && (allCand || sc[j] <= sStar);
                        const unsigned long long mask = __ballot(push);
                        const uint32_t rank = (uint32_t) __popcll(mask & ((1ULL << lane) - 1ULL));
                        const uint32_t cnt = (uint32_t) __popcll(mask);
                        // (window, score) of the candidates first; their k-mers are rebuilt below, once per candidate — under this
                        // loop the whole wavefront would rebuild them in every round
                        if (C + cnt > cap) overflow = true;
                        else if (push) sPick[C + rank] = (sc[j] << 16) | ((uint32_t) j * 64u + (uint32_t) lane);
                        C += cnt;
                    }
                }
            }
            __syncthreads();
            if (!overflow) {
                for (uint32_t i = lane; i < C; i += 64) {
                    const uint32_t pk = sPick[i];
                    Cand cd; uint32_t pos = 0; cd.kmer = pk;
                    if (!(phaseSplit & 4)) (void) windowKmer(pk & 0xFFFFu, cd.kmer, pos);
                    cd.pos = pos; cd.score = pk >> 16; cand[i] = cd;
                }
            }
            __syncthreads();
        } else {
        // pass 0: all candidates pushed / or coarse histogram; pass 1: fine histogram; pass 2: push score <= s*
        const int nPass = allCand ? 1 : 3;
        const bool resident = !FALLBACK && L <= RES_L;      // whole sequence staged once; later passes reuse codes and scores
        const bool useCache = resident && !allCand;
        if (resident) {
            for (uint32_t i = lane; i < L + 31; i += 64) {
                const char ch = (i < 64) ? cb0 : ((i < 128) ? cb1 : ((i < 192) ? cb2 : ((i < 256) ? cb3 : ((i < L) ? base[i] : (char) 0))));      // first 256 bytes were prefetched
                sCodeAll[i] = (i < L) ? sMap[(unsigned char) ch] : (unsigned char) a.xCode;
            }
            __syncthreads();
        }
        for (int pass = 0; pass < nPass; pass++) {
            if (pass < 2 && !allCand) { for (int i = lane; i < 256; i += 64) sHist[i] = 0; }
            __syncthreads();
            for (uint32_t t0 = 0; t0 < L; t0 += 64) {
                const uint32_t p = t0 + lane;
                bool valid; uint64_t kmer = 0; uint32_t pos = p, score = 0;
                const bool cached = useCache && pass > 0;          // scores come from LDS: no staging, no hashing
                if (!cached) {
                    unsigned char c;
                    if (resident) c = (p < L) ? sCodeAll[p] : (unsigned char) a.xCode;
                    else {
                        // stage codes of positions [t0, t0+64+k-1)
                        c = (p < L) ? sMap[(unsigned char) base[p]] : (unsigned char) a.xCode;
                        sCode[lane] = c;
                        if (lane < k - 1) { const uint32_t p2 = t0 + 64 + lane; sCode[64 + lane] = (p2 < L) ? sMap[(unsigned char) base[p2]] : (unsigned char) a.xCode; }
                    }
                    if (pass == 0) {   // identity hash, tile-wise Horner: h = h*31^m + sum code[j]*31^(m-1-j)
                        const uint32_t m = min(64u, L - t0);
                        const uint64_t pw = __shfl(pow31, (int) (m - 1 - min((uint32_t) lane, m - 1)), 64);
                        uint64_t term = ((uint32_t) lane < m) ? (uint64_t) c * pw : 0ull;
                        term = waveReduceSumU64(term);
                        const uint64_t pm = __shfl(pow31, (int) (m - 1), 64) * 31ull;      // 31^m
                        seqHash = seqHash * pm + term;
                    }
                    if (!resident) __syncthreads();
                    valid = (p < nWin);
                    if (valid) {
                        if (!NUCL && fastIdx) { pos = p; valid = kmerIndexFast(resident ? &sCodeAll[p] : &sCode[lane], k, (unsigned) a.xCode, (uint32_t) a.powers[1], (uint32_t) a.powers[7], kmer); }
                        else if (NUCL && a.xCode == 4) valid = kmerNuclCanonical(resident ? &sCodeAll[p] : &sCode[lane], k, L, p, kmer, pos);
                        else if (resident) valid = kmerFromCodes<NUCL>([&](int i) { return sCodeAll[p + i]; }, k, (unsigned char) a.xCode, a.powers, L, p, kmer, pos);
                        else valid = kmerFromCodes<NUCL>([&](int i) { return sCode[lane + i]; }, k, (unsigned char) a.xCode, a.powers, L, p, kmer, pos);
                    }
                    if (valid) score = (uint32_t) (xxh64U64(NUCL ? (kmer & ~BIT63) : kmer, a.seed) & 0xFFFFu);
                    if (useCache) {
                        if (p < RES_L) sScore[p] = (unsigned short) score;
                        const unsigned long long vm = __ballot(valid);
                        if (lane == 0) sValid[t0 >> 6] = vm;
                    }
                } else {
                    valid = ((sValid[t0 >> 6] >> lane) & 1ULL) != 0;
                    score = valid ? (uint32_t) sScore[p] : 0u;
                    if (pass == 2 && valid && score <= sStar)      // only the ~60 selected windows rebuild their k-mer
                    {
                        if (!NUCL && fastIdx) { pos = p; (void) kmerIndexFast(&sCodeAll[p], k, (unsigned) a.xCode, (uint32_t) a.powers[1], (uint32_t) a.powers[7], kmer); }
                        else if (NUCL && a.xCode == 4) (void) kmerNuclCanonical(&sCodeAll[p], k, L, p, kmer, pos);
                        else (void) kmerFromCodes<NUCL>([&](int i) { return sCodeAll[p + i]; }, k, (unsigned char) a.xCode, a.powers, L, p, kmer, pos);
                    }
                }
                bool push = false;
                if (allCand) push = valid;
                else if (pass == 0) { if (valid) atomicAdd(&sHist[score >> 8], 1u); }
                else if (pass == 1) { if (valid && (score >> 8) == b1) atomicAdd(&sHist[score & 255], 1u); }
                else push = valid && score <= sStar;
                if (pass == 0) n += (uint32_t) __popcll(__ballot(valid));
                if (allCand || pass == 2) {
                    const unsigned long long mask = __ballot(push);
                    const uint32_t rank = (uint32_t) __popcll(mask & ((1ULL << lane) - 1ULL));
                    const uint32_t cnt = (uint32_t) __popcll(mask);
                    if (C + cnt > cap) overflow = true;
                    else if (push) { Cand cd; cd.kmer = kmer; cd.pos = pos; cd.score = score; cand[C + rank] = cd; }
                    C += cnt;
                }
                __syncthreads();
                if (overflow) break;
            }
            if (overflow) break;
            if (!allCand && pass < 2) {
                // radix-select step over the 256-bin histogram: first bin where the running count reaches `target`
                __syncthreads();
                if (pass == 0) considered = min(consideredRaw, (size_t) n);
                const uint32_t target = (pass == 0) ? (uint32_t) considered : (uint32_t) considered - cumBefore1;
                const uint32_t h0 = sHist[lane * 4], h1 = sHist[lane * 4 + 1], h2 = sHist[lane * 4 + 2], h3 = sHist[lane * 4 + 3];
                const uint32_t mine = h0 + h1 + h2 + h3;
                const uint32_t incl = waveInclusiveScan(mine);
                const unsigned long long reach = __ballot(incl >= target && target > 0);
                uint32_t bin = 255, before = 0, upto = 0;
                if (reach) {
                    const int fl = __ffsll((long long) reach) - 1;
                    const uint32_t exB = __shfl(incl - mine, fl, 64);
                    const uint32_t q0 = __shfl(h0, fl, 64), q1 = __shfl(h1, fl, 64), q2 = __shfl(h2, fl, 64), q3 = __shfl(h3, fl, 64);
                    uint32_t run = exB; bin = (uint32_t) fl * 4;
                    if (run + q0 >= target) { before = run; upto = run + q0; }
                    else if (run + q0 + q1 >= target) { bin += 1; before = run + q0; upto = before + q1; }
                    else if (run + q0 + q1 + q2 >= target) { bin += 2; before = run + q0 + q1; upto = before + q2; }
                    else { bin += 3; before = run + q0 + q1 + q2; upto = before + q3; }
                }
                if (pass == 0) { b1 = bin; cumBefore1 = before; }
                else { sStar = (b1 << 8) | bin; tooMuch = (int) (cumBefore1 + upto) - (int) considered; }
                if (pass == 0 && (considered == 0)) break;     // nothing can be selected (n == 0)
            }
        }
        }      // three-pass path
        if (overflow) {
            if (!FALLBACK && lane == 0) { const uint32_t o = atomicAdd(a.overflowCount, 1u); a.overflowIds[o] = id; }
            __syncthreads();
            continue;
        }
        if (allCand) {
            considered = min(consideredRaw, (size_t) n);   // == n
            // threshold walk ends one past the largest score present; no surplus in the last bin
            uint32_t mx = 0;
            for (uint32_t i = lane; i < C; i += 64) mx = max(mx, cand[i].score);
            sStar = (uint32_t) waveReduceMax((int) mx); tooMuch = 0;
        }
        // ---- fast path: when no candidate k-mer repeats and the threshold bin has no surplus, the reference's
        //      sort + walk selects exactly the candidate set (C == considered), in an order that does not matter ----
        bool needOrder = (tooMuch != 0);
        if (!needOrder && a.ignoreMulti && C > 1) {
            if (FALLBACK) needOrder = true;
            else {
                for (uint32_t i = lane; i < 2 * CAP; i += 64) sSet[i] = ~0ULL;
                __syncthreads();
                bool dup = false;
                for (uint32_t i = lane; i < C; i += 64) {
                    const unsigned long long K = NUCL ? (cand[i].kmer | BIT63) : cand[i].kmer;
                    uint32_t slot = (uint32_t) ((K * 0x9E3779B97F4A7C15ULL) >> 40) & (2 * CAP - 1);
                    for (;;) {
                        const unsigned long long prev = atomicCAS(&sSet[slot], ~0ULL, K);
                        if (prev == ~0ULL) break;
                        if (prev == K) { dup = true; break; }
                        slot = (slot + 1) & (2 * CAP - 1);
                    }
                }
                needOrder = __ballot(dup) != 0ULL;
                __syncthreads();
            }
        }
        if (!needOrder) {
            // (selected-window cache, section 2c: the positions — C <= 59 here, no surplus — and the identity hash go to the sequence's line
            //  from the registers that hold them for the records; a separate block cost the tuned tiers 40 bytes of scratch per lane)
            for (uint32_t i = lane; i < C; i += 64) {
                const Cand cd = cand[i];
                R r; r.kmer = cd.kmer; r.id = id; r.len = (decltype(r.len)) L; r.pos = (decltype(r.pos)) cd.pos;
                if constexpr (LONG) r.pad = 0;
                arr[slot + 1 + i] = r;
            }
            if (lane == 0) {   // identity record (kmermatcher.cpp:241-249)
                R r; r.kmer = xxh64U64(seqHash, a.seed); r.id = id; r.len = (decltype(r.len)) L; r.pos = 0;
                if constexpr (LONG) r.pad = 0;
                arr[slot] = r;
            }
            for (uint32_t i = 1 + C + lane; i < bound; i += 64) { R r; memset(&r, 0xFF, sizeof(R)); arr[slot + i] = r; }
            if (!LONG && !FALLBACK_NOSTATS(a)) {
                unsigned char *cl = reinterpret_cast<unsigned char *>(a.kstats[4]);
                if (cl) {
                    unsigned short *ln = reinterpret_cast<unsigned short *>(cl + (size_t) id * KMC_LINE);
                    if ((uint32_t) lane < KMC_POS) ln[4 + lane] = ((uint32_t) lane < C) ? (unsigned short) cand[lane].pos : (unsigned short) 0xFFFFu;
                    if (lane == 0) *reinterpret_cast<unsigned long long *>(ln) = seqHash;
                    if (lane == 63) ln[KMC_FLAGS] = (unsigned short) KMC_CLEAN;      // every candidate was selected: no surplus in the threshold bin, no repeated k-mer
                }
            }
            stRes += L; stRec += 1 + C;
            __syncthreads();
            continue;
        }
        // ---- order candidates like SequencePosition::compareByScore[Reverse] (kmermatcher.h:13-45) ----
        uint32_t P = 1; while (P < C) P <<= 1;
        if (P > cap) P = cap;      // cap is a power of two >= C in both modes
        for (uint32_t i = C + lane; i < P; i += 64) { Cand cd; cd.kmer = ~0ULL; cd.pos = 0xFFFFFFFFu; cd.score = 0xFFFFFFFFu; cand[i] = cd; }
        __syncthreads();
        bool sortedNeeded = true;
        if (sortedNeeded && C > 1) waveBitonicSortCands(cand, P, NUCL);
        // ---- repeated k-mer skipping (kmermatcher.cpp:277-301), exact emulation of the index walk ----
        if (a.ignoreMulti) {
            bool rep = false;
            for (uint32_t i = 1 + lane; i < C; i += 64) {
                const uint64_t x = NUCL ? (cand[i].kmer | BIT63) : cand[i].kmer, y = NUCL ? (cand[i - 1].kmer | BIT63) : cand[i - 1].kmer;
                rep |= (x == y);
            }
            if (__ballot(rep)) {
                for (uint32_t i = lane; i < C; i += 64) cand[i].score |= 0x80000000u;      // skipped until visited
                __syncthreads();
                if (lane == 0) {
                    uint32_t i = 0;
                    while (i < C) {
                        const uint64_t km = NUCL ? (cand[i].kmer | BIT63) : cand[i].kmer;
                        if (i + 1 < C && (NUCL ? (cand[i + 1].kmer | BIT63) : cand[i + 1].kmer) == km) {
                            do { i++; if (i >= C) break; } while ((NUCL ? (cand[i].kmer | BIT63) : cand[i].kmer) == km);
                            if (i >= C) break;
                        }
                        cand[i].score &= 0x7FFFFFFFu;
                        i++;
                    }
                }
                __syncthreads();
            }
        }
        // ---- selection walk (kmermatcher.cpp:274-347) as prefix counts over the sorted candidates ----
        uint32_t binCarry = 0, selCarry = 0;
        for (uint32_t c0 = 0; c0 < C; c0 += 64) {
            const uint32_t i = c0 + lane;
            Cand cd; cd.kmer = 0; cd.pos = 0; cd.score = 0x80000000u;
            if (i < C) cd = cand[i];
            const bool v = (i < C) && !(cd.score & 0x80000000u);
            const uint32_t sc = cd.score & 0xFFFFu;
            const bool isBin = v && sc == sStar;
            const unsigned long long mb = __ballot(isBin);
            const uint32_t binRank = binCarry + (uint32_t) __popcll(mb & ((1ULL << lane) - 1ULL));
            const bool selectable = v && (sc < sStar || (isBin && (tooMuch == 0 || (int) binRank < tooMuch)));
            const unsigned long long ms = __ballot(selectable);
            const uint32_t selRank = selCarry + (uint32_t) __popcll(ms & ((1ULL << lane) - 1ULL));
            if (selectable && (size_t) selRank < considered) {
                R r; r.kmer = cd.kmer; r.id = id; r.len = (decltype(r.len)) L; r.pos = (decltype(r.pos)) cd.pos;
                if constexpr (LONG) r.pad = 0;
                arr[slot + 1 + selRank] = r;
                if (selRank < 64u) sKmcPos[selRank] = (unsigned short) cd.pos;       // (for the selected-window cache line below)
            }
            binCarry += (uint32_t) __popcll(mb); selCarry += (uint32_t) __popcll(ms);
        }
        const uint32_t numSel = (uint32_t) min((size_t) selCarry, considered);
        if (!LONG && !FALLBACK_NOSTATS(a)) {
            unsigned char *cl = reinterpret_cast<unsigned char *>(a.kstats[4]);
            if (cl) {
                __syncthreads();
                unsigned short *ln = reinterpret_cast<unsigned short *>(cl + (size_t) id * KMC_LINE);
                if ((uint32_t) lane < KMC_POS) ln[4 + lane] = ((uint32_t) lane < numSel) ? sKmcPos[lane] : (unsigned short) 0xFFFFu;
                if (lane == 0) *reinterpret_cast<unsigned long long *>(ln) = seqHash;
                if (lane == 63) ln[KMC_FLAGS] = 0;                   // the ordered path: candidates were dropped (surplus / repeats)
            }
        }
        if (lane == 0) {   // identity record (kmermatcher.cpp:241-249)
            R r; r.kmer = xxh64U64(seqHash, a.seed); r.id = id; r.len = (decltype(r.len)) L; r.pos = 0;
            if constexpr (LONG) r.pad = 0;
            arr[slot] = r;
        }
        for (uint32_t i = 1 + numSel + lane; i < bound; i += 64) {
            R r; memset(&r, 0xFF, sizeof(R));
            arr[slot + i] = r;
        }
        stRes += L; stRec += 1 + numSel;
        __syncthreads();
    }
    if (lane == 0 && a.kstats) { atomicAdd(&a.kstats[2], stRes); atomicAdd(&a.kstats[3], stRec); }
}

// =====================================================================================================
// 2b. short protein sequences, ONE THREAD per sequence.
//     When a sequence has no more valid k-mers than kmermatcher would consider (n <= kmer-per-seq - 1 + scale*L, true
//     for every <= ~72-residue read fragment), the reference selects ALL its k-mers unless one repeats inside the
//     sequence.  Then no per-sequence threshold, sort or wave coordination is needed: a lane rolls the k-mer index
//     along its sequence (exact division by the alphabet base via the modular inverse), hashes, and writes its slot
//     range.  Sequences that are longer, or in which two k-mers share a 16-bit hash score (possible repeat), are
//     queued for the wave-per-sequence kernel, which then owns their slot range.  ~60 instructions per sequence
//     instead of ~850 wave-instructions.
// =====================================================================================================
constexpr uint32_t SHORT_MAXL = 128;
struct ShortArgs {
    SeqView s; const uint64_t *slotOff; void *arr; const unsigned char *map;
    int k, xCode, kps, ignoreMulti; float scale; uint64_t seed;
    uint64_t base, top, inv; int tz;     // alphabet base; base^(k-1); exact division by base = (x >> tz) * inv
    uint32_t topLo, topHi, baseH;    // extractShortFastKernel: base^(H-1), base^(K-H-1), base^H for the two halves of the k-mer index (H = K / 2)
    uint32_t *waveList, *waveCount;      // sequences for the wave kernels ...
    uint32_t *longList, *longCount; uint32_t longWindows;   // ... those with more than longWindows windows go to this list instead (nullptr: one list)
    uint32_t *hugeList, *hugeCount; uint32_t hugeWindows;   //     and those with more than hugeWindows to this one (nullptr: no such list)
    unsigned long long *kstats;          // [0] residues, [1] records handled by this kernel
    uint32_t idLo, idHi; uint64_t slotBias;   // ids [idLo, idHi) (sharded run: this rank's share), records at arr[slotOff[id] - slotBias]
    const unsigned char *changed; uint32_t *cachedList, *cachedCount;   // selected-window cache (section 2c): a sequence that is too long for this
                                              // kernel and whose bytes are those of the last call's DB goes to this list, not to the wave kernels'
};

template <bool LONG>
__global__ __launch_bounds__(64) void extractShortKernel(ShortArgs a) {
    __shared__ unsigned char sMap[256];
    __shared__ __attribute__((aligned(16))) unsigned short sSet[64 * 64];    // per-lane open-addressing set of (score + 1): 64 slots (8 KB per
                                                                             // wavefront keeps ~4 wavefronts per SIMD resident; this kernel is latency bound)
    typedef Rec<LONG> R;
    R *arr = reinterpret_cast<R *>(a.arr);
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    __syncthreads();
    unsigned short *mySet = sSet + lane * 64;
    const int k = a.k;
    unsigned long long stRes = 0, stRec = 0;
    for (uint32_t b0 = a.idLo + blockIdx.x * 64; b0 < a.idHi; b0 += gridDim.x * 64) {
        const uint32_t id = b0 + lane;
        const bool active = id < a.idHi;
        bool toWave = false, lenWave = false;
        if (active) {
            const uint32_t L = a.s.len[id];
            const uint32_t nWin = (L >= (uint32_t) k) ? (L - k + 1) : 0;
            const size_t consideredRaw = (size_t) ((float) (a.kps - 1) + (a.scale * (float) L));
            if (L > SHORT_MAXL || (size_t) nWin > consideredRaw) toWave = lenWave = true;
            else {
                const char *base = a.s.data + a.s.off[id];
                const uint64_t slot = a.slotOff[id] - a.slotBias;
                const uint32_t bound = (uint32_t) (a.slotOff[id + 1] - a.slotOff[id]);
                if (a.ignoreMulti) { uint4 z = make_uint4(0, 0, 0, 0); uint4 *q = reinterpret_cast<uint4 *>(mySet); for (int i = 0; i < 8; i++) q[i] = z; }
                uint64_t idx = 0, seqHash = 0, fifoLo = 0, fifoHi = 0;   // fifo: the k codes of the current window, 8 bits each
                uint64_t pw = 1;
                int lastX = -1;
                uint32_t nOut = 0;
                uint32_t word = 0;
                R pend0, pend1, pend2;
                for (uint32_t i = 0; i < L; i++) {
                    if ((i & 3) == 0) __builtin_memcpy(&word, base + i, 4);                 // buffer is padded past its end
                    const unsigned char c = sMap[(word >> (8 * (i & 3))) & 0xFF];
                    seqHash = seqHash * 31 + c;
                    if (c == (unsigned char) a.xCode) lastX = (int) i;
                    if (i < (uint32_t) k) {                         // first window: idx = sum code[i] * base^i
                        idx += (uint64_t) c * pw; pw *= a.base;
                        if (i < 8) fifoLo |= (uint64_t) c << (8 * i); else fifoHi |= (uint64_t) c << (8 * (i - 8));
                    } else {                                        // roll: drop the oldest digit, append the new one on top
                        const uint64_t cOut = fifoLo & 0xFF;
                        idx = (((idx - cOut) >> a.tz) * a.inv) + (uint64_t) c * a.top;
                        fifoLo = (fifoLo >> 8) | (fifoHi << 56); fifoHi >>= 8;
                        if (k - 1 < 8) fifoLo |= (uint64_t) c << (8 * (k - 1)); else fifoHi |= (uint64_t) c << (8 * (k - 1 - 8));
                    }
                    if (i + 1 >= (uint32_t) k) {
                        const uint32_t p = i + 1 - k;
                        if (lastX < (int) p) {
                            // every window of such a sequence is selected whatever its XXH64 score, so the score is not needed here: the
                            // per-lane set only has to notice a POSSIBLE repeat (equal k-mers give equal tags), and any 16-bit function of
                            // the k-mer does that — one multiplication instead of XXH64's five (round 3: the kernel is issue bound)
                            const uint32_t score = (uint32_t) ((idx * 0x9E3779B97F4A7C15ULL) >> 48);
                            if (a.ignoreMulti) {
                                const unsigned short tag = (unsigned short) (score + 1);
                                if (tag == 0 || nOut >= 48) toWave = true;      // table nearly full: let the wave kernel do this one
                                else {
                                    uint32_t sl = (score * 40503u >> 7) & 63;
                                    for (;;) { const unsigned short v = mySet[sl]; if (v == tag) { toWave = true; break; } if (v == 0) { mySet[sl] = tag; break; } sl = (sl + 1) & 63; }
                                }
                            }
                            R r; r.kmer = idx; r.id = id; r.len = (decltype(r.len)) L; r.pos = (decltype(r.pos)) p;
                            if constexpr (LONG) r.pad = 0;
                            // four records at a time: the lane's stores to one cache line leave together instead of a k-mer apart
                            switch (nOut & 3u) { case 0: pend0 = r; break; case 1: pend1 = r; break; case 2: pend2 = r; break;
                                default: { R *d = arr + slot + 1 + (nOut - 3u); d[0] = pend0; d[1] = pend1; d[2] = pend2; d[3] = r; } }
                            nOut++;
                        }
                    }
                }
                if (!toWave) {
                    { R *d = arr + slot + 1 + (nOut & ~3u); const uint32_t rem = nOut & 3u; if (rem > 0) d[0] = pend0; if (rem > 1) d[1] = pend1; if (rem > 2) d[2] = pend2; }
                    R r; r.kmer = xxh64U64(seqHash, a.seed); r.id = id; r.len = (decltype(r.len)) L; r.pos = 0;
                    if constexpr (LONG) r.pad = 0;
                    arr[slot] = r;
                    R sen; memset(&sen, 0xFF, sizeof(R));
                    for (uint32_t i = 1 + nOut; i < bound; i++) arr[slot + i] = sen;
                    stRes += L; stRec += 1 + nOut;
                }
            }
        }
        // the queued sequences, one atomic per wavefront and list (the wave kernels' tiers are fed from these lists directly: a
        // queue filled one sequence at a time — one atomic on one counter per sequence — costs more than the tier it feeds)
        // (too long for this kernel by its LENGTH and unchanged since the last call: the cached kernel takes it, section 2c)
        const bool isCached = lenWave && a.cachedList && a.changed[id] == 0;
        if (isCached) toWave = false;
        const uint32_t nw = (toWave && active && a.s.len[id] >= (uint32_t) k) ? a.s.len[id] - (uint32_t) k + 1 : 0u;
        const bool isHuge = toWave && a.hugeList && nw > a.hugeWindows;
        const bool isLong = toWave && !isHuge && a.longList && nw > a.longWindows;
        auto append = [&](bool mine, uint32_t *list, uint32_t *count) {
            const unsigned long long m = __ballot(mine);
            if (!m) return;
            uint32_t basePos = 0;
            if (lane == 0) basePos = atomicAdd(count, (uint32_t) __popcll(m));
            basePos = __shfl(basePos, 0, 64);
            if (mine) list[basePos + (uint32_t) __popcll(m & ((1ULL << lane) - 1ULL))] = id;
        };
        append(toWave && !isLong && !isHuge, a.waveList, a.waveCount);
        if (a.longList) append(isLong, a.longList, a.longCount);
        if (a.hugeList) append(isHuge, a.hugeList, a.hugeCount);
        if (a.cachedList) append(isCached, a.cachedList, a.cachedCount);
    }
    stRes = waveReduceSumU64(stRes); stRec = waveReduceSumU64(stRec);
    if (lane == 0 && a.kstats) { atomicAdd(&a.kstats[0], stRes); atomicAdd(&a.kstats[1], stRec); }
}

// Nucleotide DBs (and protein k > 16) have no thread-per-sequence kernel in front of the wave kernels; this one only sorts the ids into the
// tiers' lists by window count (round 4: until then the 16-scores tier walked every sequence itself, reads of 130 windows included).
__global__ __launch_bounds__(256) void classifyWindowsKernel(const uint32_t *__restrict__ len, uint32_t idLo, uint32_t idHi, uint32_t k, uint32_t longWindows, uint32_t hugeWindows,
                                                             uint32_t *__restrict__ waveList, uint32_t *__restrict__ waveCount, uint32_t *__restrict__ longList, uint32_t *__restrict__ longCount,
                                                             uint32_t *__restrict__ hugeList, uint32_t *__restrict__ hugeCount) {
    __shared__ uint32_t sCnt[3], sBase[3];
    constexpr int PER = 8;
    for (uint64_t b0 = (uint64_t) idLo + (uint64_t) blockIdx.x * (256 * PER); b0 < idHi; b0 += (uint64_t) gridDim.x * (256 * PER)) {
        if (threadIdx.x < 3) sCnt[threadIdx.x] = 0;
        __syncthreads();
        int cls[PER]; uint32_t rank[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t id = b0 + (uint64_t) j * 256 + threadIdx.x;
            cls[j] = -1; rank[j] = 0;
            if (id < idHi) {
                const uint32_t L = len[id], nw = L >= k ? L - k + 1 : 0u;
                cls[j] = nw > hugeWindows ? 2 : (nw > longWindows ? 1 : 0);
                rank[j] = atomicAdd(&sCnt[cls[j]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 3) sBase[threadIdx.x] = sCnt[threadIdx.x] ? atomicAdd(threadIdx.x == 0 ? waveCount : (threadIdx.x == 1 ? longCount : hugeCount), sCnt[threadIdx.x]) : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (cls[j] >= 0) (cls[j] == 0 ? waveList : (cls[j] == 1 ? longList : hugeList))[sBase[cls[j]] + rank[j]] = (uint32_t) (b0 + (uint64_t) j * 256 + threadIdx.x);
        __syncthreads();
    }
}

// The same, restated for the instruction mix (round 3).  The PMC pass over the kernel above (profiles/r03_pmc) showed it bound by
// SCALAR issue — 1.3 M scalar against 0.67 M vector instructions per wavefront: per residue a chain of divergent branches (first
// window or not, word boundary, X, the four-way switch of the pending stores, the probe loop), each paid in exec-mask bookkeeping —
// and a third of its vector time in 64-bit multiplications (the rolling index and the repeat tag: twelve quarter-rate instructions
// per residue).  Here
//  * the lanes of a wavefront walk their sequences in lockstep, four residues (one 32-bit load) per iteration of a wave-uniform loop;
//    the first window needs no code of its own: the index starts from K virtual letters of code 0 and is rolled forward;
//  * the k-mer index base^0 d_0 + ... + base^(K-1) d_(K-1) is kept as TWO 32-bit halves (low H digits, high K - H digits): a roll
//    is (x - d) >> tz times the 32-bit inverse of the odd part of the base plus digit * power — one quarter-rate multiplication per
//    half, the digit products on the 24-bit multiplier — and the 64-bit index lo + hi * base^H is built only for a record that is written;
//  * the digits that leave the halves come from a 16-byte register FIFO of the last letter codes at compile-time byte positions;
//  * the repeat tag is any 16-bit function of the k-mer (see above): two 24-bit products of the halves.
// Records, lists and statistics are those of the kernel above (which stays for k != 14 and for alphabets whose half-index does not
// fit 32 bits).
// the 24-bit multiplier (full rate; the compiler prefers the quarter-rate 32-bit one when it can prove the results equal)
__device__ __forceinline__ uint32_t mulU24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t waveMaxU32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t) __shfl_xor((int) v, o, 64));
    return v;
}
template <bool LONG, int K, bool MUL24>
__global__ __launch_bounds__(64) void extractShortFastKernel(ShortArgs a) {
    constexpr int H = K / 2;
    static_assert(K <= 16 && 16 - K + H + 3 <= 15, "the digits that leave the halves are read from the 16-byte FIFO before this iteration's codes enter it");
    __shared__ unsigned char sMap[256];
    __shared__ __attribute__((aligned(16))) unsigned short sSet[64 * 64];    // per-lane open-addressing set of tags, as above
    typedef Rec<LONG> R;
    R *arr = reinterpret_cast<R *>(a.arr);
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    __syncthreads();
    unsigned short *mySet = sSet + lane * 64;
    const uint32_t inv32 = (uint32_t) a.inv, tz = (uint32_t) a.tz, topLo = a.topLo, topHi = a.topHi, xCode = (uint32_t) a.xCode;
    const uint32_t baseH = a.baseH;
    const bool multi = a.ignoreMulti != 0;
    auto digitMul = [&](uint32_t d, uint32_t pw) -> uint32_t { return MUL24 ? __umul24(d, pw) : d * pw; };
    unsigned long long stRes = 0, stRec = 0;
    for (uint32_t b0 = a.idLo + blockIdx.x * 64; b0 < a.idHi; b0 += gridDim.x * 64) {
        const uint32_t id = b0 + lane;
        const bool active = id < a.idHi;
        bool toWave = false, lenWave = false;
        uint32_t L = 0;
        if (active) {
            L = a.s.len[id];
            const uint32_t nWin = (L >= (uint32_t) K) ? (L - K + 1) : 0;
            const size_t consideredRaw = (size_t) ((float) (a.kps - 1) + (a.scale * (float) L));
            // (more than 48 windows would overfill the 64-slot tag set: the kernel above hands such a sequence over at its 49th
            // record, here it goes at once — the wave kernel's records are the same either way)
            if (L > SHORT_MAXL || (size_t) nWin > consideredRaw || (multi && nWin > 48)) toWave = lenWave = true;
        }
        const bool work = active && !toWave;
        const uint32_t Lmax = (uint32_t) __builtin_amdgcn_readfirstlane((int) waveMaxU32(work ? L : 0u));
        if (Lmax) {
            const char *base = a.s.data;
            uint64_t slot = 0; uint32_t bound = 0;
            if (work) {
                base += a.s.off[id];
                slot = a.slotOff[id] - a.slotBias; bound = (uint32_t) (a.slotOff[id + 1] - a.slotOff[id]);
                if (multi) { uint4 z = make_uint4(0, 0, 0, 0); uint4 *q = reinterpret_cast<uint4 *>(mySet); for (int i = 0; i < 8; i++) q[i] = z; }
            }
            const uint32_t Lw = work ? L : 0u;                 // a lane without work has no residue inside
            uint32_t lo = 0, hi = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, nOut = 0;
            uint64_t seqHash = 0;
            int lastX = -1;
            uint32_t wordNext = 0;
            if (Lw) __builtin_memcpy(&wordNext, base, 4);                                   // the buffer is padded past its end
            for (uint32_t i = 0; i < Lmax; i += 4) {
                // the four residues of the NEXT step are requested before this step's are used: the load is the head of the step's
                // chain of dependent round trips (residues -> letter codes in LDS -> tag set in LDS -> store)
                const uint32_t word = wordNext;
                wordNext = 0;
                if (i + 4 < Lw) __builtin_memcpy(&wordNext, base + i + 4, 4);
                const uint32_t cw = (uint32_t) sMap[word & 0xFFu] | ((uint32_t) sMap[(word >> 8) & 0xFFu] << 8) |
                                    ((uint32_t) sMap[(word >> 16) & 0xFFu] << 16) | ((uint32_t) sMap[word >> 24] << 24);
                const uint32_t f[4] = {f0, f1, f2, f3};
                // identity hash (Util::hash: h = h * 31 + code), the up to four residues of this step at once: h * 31^m + their Horner sum
                uint32_t part = 0, mult = 1;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool in = i + (uint32_t) j < Lw;
                    const uint32_t c = (cw >> (8 * j)) & 0xFFu;
                    part = in ? __umul24(part, 31u) + c : part;
                    mult = in ? __umul24(mult, 31u) : mult;
                }
                seqHash = seqHash * (uint64_t) mult + (uint64_t) part;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t ii = i + (uint32_t) j;
                    const bool in = ii < Lw;
                    const uint32_t c = (cw >> (8 * j)) & 0xFFu;
                    lastX = (in && c == xCode) ? (int) ii : lastX;
                    constexpr int B0 = 16 - K, BH = 16 - K + H;
                    const uint32_t d0 = (f[(B0 + j) >> 2] >> (8 * ((B0 + j) & 3))) & 0xFFu;    // the digits of window ii - K that leave the halves
                    const uint32_t dh = (f[(BH + j) >> 2] >> (8 * ((BH + j) & 3))) & 0xFFu;
                    lo = ((lo - d0) >> tz) * inv32 + digitMul(dh, topLo);
                    hi = ((hi - dh) >> tz) * inv32 + digitMul(c, topHi);
                    const int p = (int) ii + 1 - K;
                    if (in && !toWave && p >= 0 && lastX < p) {
                        if (multi) {
                            const uint32_t t = mulU24(lo, 0x9E3779u) ^ mulU24(hi, 0x85EBCBu) ^ (lo >> 9) ^ (hi >> 7);
                            const unsigned short tag = (unsigned short) max((t >> 8) & 0xFFFFu, 1u);      // 0 marks an empty slot
                            uint32_t sl = (t >> 3) & 63u;
                            unsigned short v = mySet[sl];
                            while (v != 0 && v != tag) { sl = (sl + 1) & 63u; v = mySet[sl]; }     // one exit condition: the first probe almost always ends it
                            if (v == tag) toWave = true; else mySet[sl] = tag;
                        }
                        R r; r.kmer = (uint64_t) lo + (uint64_t) hi * (uint64_t) baseH; r.id = id; r.len = (decltype(r.len)) L; r.pos = (decltype(r.pos)) p;
                        if constexpr (LONG) r.pad = 0;
                        arr[slot + 1 + nOut] = r;
                        nOut++;
                    }
                }
                f0 = f1; f1 = f2; f2 = f3; f3 = cw;
            }
            if (work && !toWave) {
                R r; r.kmer = xxh64U64(seqHash, a.seed); r.id = id; r.len = (decltype(r.len)) L; r.pos = 0;
                if constexpr (LONG) r.pad = 0;
                arr[slot] = r;
                R sen; memset(&sen, 0xFF, sizeof(R));
                for (uint32_t i = 1 + nOut; i < bound; i++) arr[slot + i] = sen;
                stRes += L; stRec += 1 + nOut;
            }
        }
        const bool isCached = lenWave && a.cachedList && a.changed[id] == 0;      // (section 2c)
        if (isCached) toWave = false;
        const uint32_t nw = (toWave && active && L >= (uint32_t) K) ? L - (uint32_t) K + 1 : 0u;
        const bool isHuge = toWave && a.hugeList && nw > a.hugeWindows;
        const bool isLong = toWave && !isHuge && a.longList && nw > a.longWindows;
        auto append = [&](bool mine, uint32_t *list, uint32_t *count) {
            const unsigned long long m = __ballot(mine);
            if (!m) return;
            uint32_t basePos = 0;
            if (lane == 0) basePos = atomicAdd(count, (uint32_t) __popcll(m));
            basePos = __shfl(basePos, 0, 64);
            if (mine) list[basePos + (uint32_t) __popcll(m & ((1ULL << lane) - 1ULL))] = id;
        };
        append(toWave && !isLong && !isHuge, a.waveList, a.waveCount);
        if (a.longList) append(isLong, a.longList, a.longCount);
        if (a.hugeList) append(isHuge, a.hugeList, a.hugeCount);
        if (a.cachedList) append(isCached, a.cachedList, a.cachedCount);
    }
    stRes = waveReduceSumU64(stRes); stRec = waveReduceSumU64(stRec);
    if (lane == 0 && a.kstats) { atomicAdd(&a.kstats[0], stRes); atomicAdd(&a.kstats[1], stRec); }
}

// =====================================================================================================
// 2c. the selected-window cache.  Plass changes the hash seed every other iteration (Assembler.cpp:99-110: hashShift += i % 2), and
//     an iteration extends 10-20 % of the sequences: in iterations 2, 4, 6, ... most sequences are byte for byte what they were when
//     kmermatcher last hashed them with this very seed, so the windows it selects are the same.  The wave-per-sequence kernels
//     therefore leave, per sequence, one 128-byte line {identity hash, positions of the <= 59 selected windows (0xFFFF = none), flags};
//     the next call with the same selection parameters sends every sequence its DB inherited UNCHANGED (plasship_seqdb::parentGen /
//     d_changed) here instead: the k-mers at the cached positions are rebuilt from the sequence's bytes and the records written — no
//     window is hashed, nothing is selected.  Protein DBs with --kmer-per-seq <= 60 and no length scaling (the Plass workflow) only:
//     that is where a line holds every selected window.  The record set is exactly what the full kernels write (tests: the chained
//     iterations of every protein parity test pass through here; PLASSHIP_TUNE_KMCACHE=2 switches it off).
// =====================================================================================================
struct CachedArgs {
    SeqView s; const uint64_t *slotOff; void *arr; const unsigned char *map; const unsigned char *lines;
    const uint32_t *list, *count; int k, xCode; uint32_t base, base7; uint64_t slotBias, seed; unsigned long long *kstats;
};
__global__ __launch_bounds__(64) void extractCachedKernel(CachedArgs a) {
    __shared__ unsigned char sMap[256];
    typedef Rec<false> R;
    R *arr = reinterpret_cast<R *>(a.arr);
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    __syncthreads();
    const uint32_t nWork = *a.count;
    unsigned long long stRes = 0, stRec = 0;
    // A sequence is a chain of dependent round trips (list entry -> index entry and cache line -> the windows' bytes -> the records):
    // the list entry of sequence w + 2 * grid, and index entry, slot range and cache line of sequence w + grid, are in flight while
    // sequence w is written, so one round trip per sequence — its bytes — is left on the critical path.
    struct Meta { uint32_t id, L, pos; uint64_t off, slot, slot1; unsigned long long idh; };
    auto loadMeta = [&](uint32_t id) {
        Meta m; m.id = id; m.L = a.s.len[id]; m.off = a.s.off[id]; m.slot = a.slotOff[id] - a.slotBias; m.slot1 = a.slotOff[id + 1] - a.slotBias;
        const unsigned short *ln = reinterpret_cast<const unsigned short *>(a.lines + (size_t) id * KMC_LINE);
        m.pos = ((uint32_t) lane < KMC_POS) ? (uint32_t) ln[4 + lane] : 0xFFFFu;
        m.idh = *reinterpret_cast<const unsigned long long *>(ln);
        return m;
    };
    Meta nxt; nxt.id = 0; nxt.L = 0; nxt.pos = 0xFFFFu; nxt.off = 0; nxt.slot = 0; nxt.slot1 = 0; nxt.idh = 0;
    uint32_t id2 = 0;
    if (blockIdx.x < nWork) nxt = loadMeta(a.list[blockIdx.x]);
    if (blockIdx.x + gridDim.x < nWork) id2 = a.list[blockIdx.x + gridDim.x];
    for (uint32_t w = blockIdx.x; w < nWork; w += gridDim.x) {
        const Meta cur = nxt;
        if (w + gridDim.x < nWork) nxt = loadMeta(id2);
        if (w + 2 * gridDim.x < nWork) id2 = a.list[w + 2 * gridDim.x];
        const char *base = a.s.data + cur.off;
        const uint32_t bound = (uint32_t) (cur.slot1 - cur.slot);
        const bool have = cur.pos != 0xFFFFu;
        const uint32_t n = (uint32_t) __popcll(__ballot(have));          // the positions fill the line from its front
        if (have) {
            // the 14 residues of the window (the entry is "SEQ\n\0": 16 bytes from pos <= L - 14 stay inside it), mapped to letter codes
            uint64_t r0, r1; __builtin_memcpy(&r0, base + cur.pos, 8); __builtin_memcpy(&r1, base + cur.pos + 8, 8);
            uint64_t w0 = 0, w1 = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) { w0 |= (uint64_t) sMap[(r0 >> (8 * b)) & 0xFF] << (8 * b); w1 |= (uint64_t) sMap[(r1 >> (8 * b)) & 0xFF] << (8 * b); }
            uint64_t kmer; (void) kmerIndexCore(w0, w1, a.k, (unsigned) a.xCode, a.base, a.base7, kmer);
            R r; r.kmer = kmer; r.id = cur.id; r.len = (uint16_t) cur.L; r.pos = (int16_t) cur.pos;
            arr[cur.slot + 1 + (uint32_t) lane] = r;
        }
        if (lane == 63) { R r; r.kmer = xxh64U64(cur.idh, a.seed); r.id = cur.id; r.len = (uint16_t) cur.L; r.pos = 0; arr[cur.slot] = r; }     // identity record (kmermatcher.cpp:241-249)
        for (uint32_t i = 1 + n + (uint32_t) lane; i < bound; i += 64) { R r; memset(&r, 0xFF, sizeof(R)); arr[cur.slot + i] = r; }
        stRes += cur.L; stRec += 1 + n;
    }
    if (lane == 0 && a.kstats) { atomicAdd(&a.kstats[2], stRes); atomicAdd(&a.kstats[3], stRec); }
}

// the sequences the last tier handed to the HBM-scratch launch: their slots become sentinels.  (Protein runs learn of such a hand-over
// only with the group stage's counts — kmermatchImpl, `overflowPossible` — and start over then; until that point the partition and
// the group kernel read defined bytes.)  One wavefront per queued sequence; the queue is almost always empty.
template <bool LONG>
__global__ __launch_bounds__(64) void fillOverflowSlotsKernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ count, const uint64_t *__restrict__ slotOff, uint64_t slotBias, void *arrV) {
    Rec<LONG> *arr = reinterpret_cast<Rec<LONG> *>(arrV);
    const uint32_t n = *count;
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        const uint32_t id = ids[w];
        const uint64_t s0 = slotOff[id] - slotBias, s1 = slotOff[id + 1] - slotBias;
        for (uint64_t i = s0 + threadIdx.x; i < s1; i += 64) { Rec<LONG> r; memset(&r, 0xFF, sizeof(r)); arr[i] = r; }
    }
}

__global__ void gatherU32Kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// =====================================================================================================
// 4. assignGroup over hash buckets with an LDS hash table (kmermatcher.cpp:450-559)
// =====================================================================================================
constexpr int GR_BLOCK = 256;
constexpr uint32_t GR_HT = 2048;            // slots
constexpr uint32_t GR_MAXKEYS = 1536;       // distinct k-mers per sub-pass before splitting further

struct GroupArgs {
    const void *in; void *out;
    const uint64_t *bucketStart;     // [nBuckets+1] (dense input)
    const uint32_t *list, *lineBeg, *lineCnt;   // LINES input: bucket b = the lines list[lineBeg[b] .. + lineCnt[b]) of `in` (linepart.hpp)
    uint32_t nBuckets, bucketsPerBlock;
    uint64_t *outCount;              // [gridDim.x] records written by block j at out[bucketStart[j*bucketsPerBlock] ...]
    int includeOnlyExtendable, covMode; float covThr;
    const unsigned long long *minKey;   // NUCL: K of the globally first run
    unsigned long long *maxRepTarget;   // max over emitted records of (rep << 32 | member): the last run of sort #2
};

__device__ __forceinline__ bool canBeCoveredK(float covThr, int covMode, float q, float t) {   // Util.cpp:533-550
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

// Nucleotide strand ties of sort #2 (kmermatcher.h:98-130 compares rep, target and diagonal only; kmermatcher.cpp:866-893 reports the
// strand of the LAST record of the best diagonal's run): the reference's ips4o leaves the records of one (rep, target, diagonal) triple
// in the order assignGroup wrote them — sort-#1 order, ascending k-mer — so the strand that counts is that of the member with the
// LARGEST k-mer (tests/golden/make_strand_ties.py measures it against the unmodified reference).  A grouped nucleotide record
// therefore carries the k-mer it was made from in its spare bits: bits 32..62 of the rep field (the rep id needs 32, bit 63 is the
// strand) hold the k-mer's low 31 bits, the length field — which nothing reads after assignGroup — the rest (16 bits in the 16-byte
// layout: k <= 23; plasship_kmermatch moves a longer k to the 24-byte layout).  The aggregation keeps, per triple, the strand of the
// largest (k-mer, strand) word (aggSortKernel).
template <bool LONG> __device__ __forceinline__ void embedOrd(Rec<LONG> &o, uint64_t memberKmerField) {
    const uint64_t K = memberKmerField & ~BIT63;
    o.kmer |= (K & 0x7FFFFFFFull) << 32;
    o.len = (decltype(o.len)) (K >> 31);
}
// (k-mer << 1 | forward strand) of a grouped nucleotide record: what the members of a triple are ranked by
template <bool LONG> __device__ __forceinline__ unsigned long long ordWordOf(const Rec<LONG> &r) {
    const uint64_t K = ((r.kmer >> 32) & 0x7FFFFFFFull) | ((uint64_t) (LONG ? (uint32_t) r.len : (uint32_t) (uint16_t) r.len) << 31);
    return (K << 1) | (r.kmer >> 63);
}

template <bool NUCL, bool LONG, bool LINES>
__global__ __launch_bounds__(GR_BLOCK) void groupKernel(GroupArgs a) {
    __shared__ unsigned long long hKey[GR_HT];
    __shared__ unsigned long long hBest[GR_HT];
    __shared__ uint32_t hCnt[GR_HT];
    __shared__ uint32_t hLen[GR_HT];
    __shared__ uint32_t sFlag[2];
    __shared__ uint32_t sCursor;
    typedef Rec<LONG> R;
    const R *in = reinterpret_cast<const R *>(a.in);
    R *out = reinterpret_cast<R *>(a.out);
    const uint32_t bBegin = blockIdx.x * a.bucketsPerBlock;
    const uint32_t bEnd = min(a.nBuckets, bBegin + a.bucketsPerBlock);
    if (bBegin >= a.nBuckets) { if (threadIdx.x == 0) a.outCount[blockIdx.x] = 0; return; }
    unsigned long long written = 0;                  // block-uniform
    unsigned long long maxRT = 0;
    // the workgroup writes its grouped records where its input begins: it never emits more records than it read
    const uint64_t arena = LINES ? (uint64_t) a.lineBeg[bBegin] * RPL : a.bucketStart[bBegin];
    const unsigned long long firstRunKey = (NUCL && a.minKey) ? *a.minKey : 0ull;
    for (uint32_t b = bBegin; b < bEnd; b++) {
        // records [s0, s1) of the bucket; LINES: positions in the bucket's line list (padding sentinels are skipped)
        const uint64_t s0 = LINES ? 0ull : a.bucketStart[b], s1 = LINES ? (uint64_t) a.lineCnt[b] * RPL : a.bucketStart[b + 1];
        const uint32_t lb = LINES ? a.lineBeg[b] : 0u;
        auto recAt = [&](uint64_t i) -> R { if (LINES) return in[(uint64_t) a.list[lb + (uint32_t) (i / RPL)] * RPL + (i % RPL)]; return in[i]; };
        if (s1 <= s0) continue;
        uint32_t nSub = 1;                           // sub-passes by a secondary hash when too many distinct k-mers
        const unsigned long long writtenAtBucketStart = written;
        for (;;) {
            bool redo = false;
            for (uint32_t sub = 0; sub < nSub && !redo; sub++) {
                for (uint32_t i = threadIdx.x; i < GR_HT; i += GR_BLOCK) { hKey[i] = ~0ULL; hBest[i] = ~0ULL; hCnt[i] = 0; hLen[i] = 0; }
                if (threadIdx.x == 0) { sFlag[0] = 0; sFlag[1] = 0; sCursor = 0; }
                __syncthreads();
                // phase A: insert keys, count members, longest sequence
                for (uint64_t i = s0 + threadIdx.x; i < s1; i += GR_BLOCK) {
                    const R r = recAt(i);
                    if (LINES && isSentinel(r)) continue;
                    const unsigned long long K = NUCL ? (r.kmer | BIT63) : r.kmer;
                    const uint64_t hh = K * 0xD6E8FEB86659FD93ULL;
                    if (nSub > 1 && (uint32_t) ((hh >> 40) % nSub) != sub) continue;
                    uint32_t slot = (uint32_t) (hh >> 32) & (GR_HT - 1);
                    for (uint32_t probe = 0; probe < GR_HT; probe++) {
                        const unsigned long long prev = atomicCAS(&hKey[slot], ~0ULL, K);
                        if (prev == ~0ULL) { atomicAdd(&sFlag[0], 1u); }
                        if (prev == ~0ULL || prev == K) { atomicAdd(&hCnt[slot], 1u); atomicMax(&hLen[slot], (uint32_t) r.len); break; }
                        slot = (slot + 1) & (GR_HT - 1);
                        if (probe == GR_HT - 1) atomicExch(&sFlag[1], 1u);
                    }
                }
                __syncthreads();
                if (sFlag[1] || sFlag[0] > GR_MAXKEYS) { redo = true; __syncthreads(); break; }
                // phase B: head of the run = (longest, smallest id, smallest pos[, reverse strand first])
                for (uint64_t i = s0 + threadIdx.x; i < s1; i += GR_BLOCK) {
                    const R r = recAt(i);
                    if (LINES && isSentinel(r)) continue;
                    const unsigned long long K = NUCL ? (r.kmer | BIT63) : r.kmer;
                    const uint64_t hh = K * 0xD6E8FEB86659FD93ULL;
                    if (nSub > 1 && (uint32_t) ((hh >> 40) % nSub) != sub) continue;
                    uint32_t slot = (uint32_t) (hh >> 32) & (GR_HT - 1);
                    while (hKey[slot] != K) slot = (slot + 1) & (GR_HT - 1);
                    if ((uint32_t) r.len == hLen[slot]) {
                        const unsigned long long packed = ((unsigned long long) r.id << 22) | ((unsigned long long) (uint32_t) r.pos << 1) | (NUCL ? ((r.kmer >> 63) & 1ULL) : 0ULL);
                        atomicMin(&hBest[slot], packed);
                    }
                }
                __syncthreads();
                // phase C: every member of a run of size >= 2 becomes (rep, member, diagonal) if it passes the filter
                for (uint64_t i0 = s0; i0 < s1; i0 += GR_BLOCK) {
                    const uint64_t i = i0 + threadIdx.x;
                    bool keep = false; R o; memset(&o, 0, sizeof(R));
                    if (i < s1) {
                        const R r = recAt(i);
                        const unsigned long long K = NUCL ? (r.kmer | BIT63) : r.kmer;
                        const uint64_t hh = K * 0xD6E8FEB86659FD93ULL;
                        if (!(LINES && isSentinel(r)) && !(nSub > 1 && (uint32_t) ((hh >> 40) % nSub) != sub)) {
                            uint32_t slot = (uint32_t) (hh >> 32) & (GR_HT - 1);
                            while (hKey[slot] != K) slot = (slot + 1) & (GR_HT - 1);
                            if (hCnt[slot] >= 2) {
                                const unsigned long long best = hBest[slot];
                                const uint32_t repId = (uint32_t) (best >> 22);
                                const int repPos = (int) ((best >> 1) & 0x1FFFFFu);
                                const int queryLen = (int) hLen[slot];
                                const int mLen = (int) r.len, mPos = (int) r.pos;
                                int diagonal = repPos - mPos;
                                unsigned long long rId = repId;
                                if (NUCL) {
                                    bool repIsReverse = ((best & 1ULL) == 0);
                                    if (K == firstRunKey) repIsReverse = false;       // kmermatcher.cpp:463 (never refreshed for run 0)
                                    const bool targetIsReverse = ((r.kmer & BIT63) == 0);
                                    int qp, tp; bool qRev;
                                    // positions are truncated to T exactly like the reference's T queryPos/targetPos
                                    if (repIsReverse && !targetIsReverse) { qp = repPos; tp = mPos; qRev = true; }
                                    else if (repIsReverse && targetIsReverse) { qp = (queryLen - 1) - repPos; tp = (mLen - 1) - mPos; qRev = false; }
                                    else if (!repIsReverse && targetIsReverse) { qp = (queryLen - 1) - repPos; tp = (mLen - 1) - mPos; qRev = true; }
                                    else { qp = repPos; tp = mPos; qRev = false; }
                                    if (!LONG) { qp = (int) (short) qp; tp = (int) (short) tp; }
                                    diagonal = qp - tp;
                                    rId = qRev ? (rId & ~BIT63) : (rId | BIT63);
                                }
                                const bool canBeExtended = diagonal < 0 || (diagonal > (queryLen - mLen));
                                const bool cov = canBeCoveredK(a.covThr, a.covMode, (float) queryLen, (float) mLen);
                                keep = (!a.includeOnlyExtendable && cov) || (canBeExtended && a.includeOnlyExtendable);
                                o.kmer = rId; o.id = r.id; o.len = r.len; o.pos = (decltype(o.pos)) diagonal;
                                if (NUCL) embedOrd(o, r.kmer);
                                if (keep) maxRT = max(maxRT, (unsigned long long) (((rId & ~BIT63) << 32) | (unsigned long long) r.id));
                            }
                        }
                    }
                    // compaction into this block's arena: the order inside the arena is irrelevant (the next stage is a
                    // partition), so every wavefront just claims a run from an LDS cursor — no block barrier in this loop
                    const unsigned long long mk = __ballot(keep);
                    const uint32_t wr = (uint32_t) __popcll(mk & ((1ULL << laneId()) - 1ULL));
                    uint32_t wbase = 0;
                    if (mk) {
                        if (laneId() == 0) wbase = atomicAdd(&sCursor, (uint32_t) __popcll(mk));
                        wbase = __shfl(wbase, 0, 64);
                    }
                    if (keep) out[arena + written + wbase + wr] = o;
                }
                __syncthreads();
                written += sCursor;
                __syncthreads();
                if (threadIdx.x == 0) sCursor = 0;
            }
            if (!redo) break;
            // a retry discards what completed sub-passes of this attempt wrote: rewind the arena cursor
            nSub *= 2;
            written = writtenAtBucketStart;
            __syncthreads();
        }
    }
    if (a.maxRepTarget) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxRT = max(maxRT, (unsigned long long) __shfl_xor(maxRT, o, 64));
        if (laneId() == 0 && maxRT) atomicMax(a.maxRepTarget, maxRT);
    }
    if (threadIdx.x == 0) a.outCount[blockIdx.x] = written;
}

// ---- the same over a line store, 16-byte records: the bucket is read ONCE --------------------------------------------------
// A bucket (~1000-2000 records behind a list of 128-byte lines) stays in REGISTERS (8 records per thread) for both phases, and
// the run head is ONE atomicMin per record on a packed (longest, smallest id, smallest position, reverse strand first) word —
// sequences of the KmerPosition<short> layout are shorter than 32 767, so the four fields fit 63 bits.  (The three-phase kernel
// above reads every record three times and needs a barrier more per bucket; it remains for 24-byte records, dense input and
// buckets beyond 2048 positions.)
// BLOCK x HT: 256 threads and 2048 slots hold buckets of up to 2048 positions (~1500 distinct k-mers); the 50 M-read sets fill the
// 2^20 buckets two partition levels can make with ~4000 positions each, which 512 threads and 4096 slots take in one go (a
// bucket beyond the registers would be read from HBM once per phase and sub-pass).  "At least two members" is one bit per slot.
constexpr int GL_RMAX = 8;
template <bool NUCL, int BLOCK, uint32_t HT, int WPE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void groupLinesKernel(GroupArgs a) {
    typedef Rec<false> R;
    constexpr uint32_t MAXKEYS = HT / 4 * 3;                 // distinct k-mers per sub-pass before splitting further
    __shared__ unsigned long long hKey[HT];
    __shared__ unsigned long long hBest[HT];
    __shared__ uint32_t hMulti[HT / 32];                     // bit = a second record met this slot's k-mer
    __shared__ uint32_t sFlag[2];
    __shared__ uint32_t sCursor[2];                          // arena cursor of a sub-pass; two, used alternately, save a barrier per sub-pass
    const R *in = reinterpret_cast<const R *>(a.in);
    R *out = reinterpret_cast<R *>(a.out);
    const uint32_t bBegin = blockIdx.x * a.bucketsPerBlock;
    const uint32_t bEnd = min(a.nBuckets, bBegin + a.bucketsPerBlock);
    if (bBegin >= a.nBuckets) { if (threadIdx.x == 0) a.outCount[blockIdx.x] = 0; return; }
    unsigned long long written = 0;                  // block-uniform
    unsigned long long maxRT = 0;
    const uint64_t arena = (uint64_t) a.lineBeg[bBegin] * RPL;
    const unsigned long long firstRunKey = (NUCL && a.minKey) ? *a.minKey : 0ull;
    const R none = [] { R r; r.kmer = ~0ULL; r.id = 0xFFFFFFFFu; r.len = 0; r.pos = 0; return r; }();
    // the records of a bucket are fetched (line list entry, then the record: two dependent round trips) as soon as the registers
    // of the previous bucket are dead — behind its last phase, ahead of the barriers that close it and of the table reset
    R rg[GL_RMAX];
    uint32_t nNext = 0, lbNext = 0;
    auto fetch = [&](uint32_t b) {
        nNext = (b < bEnd) ? a.lineCnt[b] * RPL : 0u; lbNext = (b < bEnd) ? a.lineBeg[b] : 0u;
        if (nNext && nNext <= (uint32_t) GL_RMAX * BLOCK) {
#pragma unroll
            for (int j = 0; j < GL_RMAX; j++) { const uint32_t i = (uint32_t) j * BLOCK + threadIdx.x; rg[j] = (i < nNext) ? in[(uint64_t) a.list[lbNext + i / RPL] * RPL + (i % RPL)] : none; }
        }
    };
    uint32_t par = 0;                                // which cursor the current sub-pass uses (workgroup-uniform)
    fetch(bBegin);
    for (uint32_t b = bBegin; b < bEnd; b++) {
        const uint32_t n = nNext;                    // record positions of the bucket (padding sentinels included)
        const uint32_t lb = lbNext;
        if (n == 0) { fetch(b + 1); continue; }
        auto recAt = [&](uint32_t i) -> R { return in[(uint64_t) a.list[lb + i / RPL] * RPL + (i % RPL)]; };
        const bool inRegs = n <= (uint32_t) GL_RMAX * BLOCK;
        bool fetched = false;
        // phase A on one record: claim the k-mer's slot, mark a second member, and bid for the run head; called by whole wavefronts
        // (new k-mers are counted once per wavefront, not with one LDS atomic per record on a single word)
        auto phaseA = [&](const R &r, uint32_t nSub, uint32_t sub) {
            bool claimed = false, full = false;
            const unsigned long long K = NUCL ? (r.kmer | BIT63) : r.kmer;
            const uint64_t hh = K * 0xD6E8FEB86659FD93ULL;
            if (!isSentinel(r) && !(nSub > 1 && (uint32_t) ((hh >> 40) % nSub) != sub)) {
                uint32_t slot = (uint32_t) (hh >> 32) & (HT - 1);
                full = true;
                for (uint32_t probe = 0; probe < HT; probe++) {
                    const unsigned long long prev = atomicCAS(&hKey[slot], ~0ULL, K);
                    if (prev == ~0ULL || prev == K) {
                        claimed = (prev == ~0ULL);
                        if (!claimed) atomicOr(&hMulti[slot >> 5], 1u << (slot & 31));
                        const unsigned long long packed = ((unsigned long long) (0x7FFFu - (uint32_t) r.len) << 48) | ((unsigned long long) r.id << 16) |
                                                          ((unsigned long long) ((uint32_t) r.pos & 0x7FFFu) << 1) | (NUCL ? ((r.kmer >> 63) & 1ULL) : 0ULL);
                        atomicMin(&hBest[slot], packed);
                        full = false;
                        break;
                    }
                    slot = (slot + 1) & (HT - 1);
                }
            }
            const unsigned long long cm = __ballot(claimed);
            if (cm && laneId() == 0) atomicAdd(&sFlag[0], (uint32_t) __popcll(cm));
            if (full) atomicExch(&sFlag[1], 1u);      // table full
        };
        // phase C on one record: (rep, member, diagonal) if the run has at least two members and the filter keeps it
        auto phaseC = [&](const R &r, uint32_t nSub, uint32_t sub) {
            bool keep = false; R o; o.kmer = 0; o.id = 0; o.len = 0; o.pos = 0;
            if (!isSentinel(r)) {
                const unsigned long long K = NUCL ? (r.kmer | BIT63) : r.kmer;
                const uint64_t hh = K * 0xD6E8FEB86659FD93ULL;
                if (!(nSub > 1 && (uint32_t) ((hh >> 40) % nSub) != sub)) {
                    uint32_t slot = (uint32_t) (hh >> 32) & (HT - 1);
                    while (hKey[slot] != K) slot = (slot + 1) & (HT - 1);
                    if ((hMulti[slot >> 5] >> (slot & 31)) & 1u) {
                        const unsigned long long best = hBest[slot];
                        const uint32_t repId = (uint32_t) (best >> 16);
                        const int repPos = (int) ((best >> 1) & 0x7FFFu);
                        const int queryLen = (int) (0x7FFFu - (uint32_t) (best >> 48));
                        const int mLen = (int) r.len, mPos = (int) r.pos;
                        int diagonal = repPos - mPos;
                        unsigned long long rId = repId;
                        if (NUCL) {
                            bool repIsReverse = ((best & 1ULL) == 0);
                            if (K == firstRunKey) repIsReverse = false;       // kmermatcher.cpp:463 (never refreshed for run 0)
                            const bool targetIsReverse = ((r.kmer & BIT63) == 0);
                            int qp, tp; bool qRev;
                            if (repIsReverse && !targetIsReverse) { qp = repPos; tp = mPos; qRev = true; }
                            else if (repIsReverse && targetIsReverse) { qp = (queryLen - 1) - repPos; tp = (mLen - 1) - mPos; qRev = false; }
                            else if (!repIsReverse && targetIsReverse) { qp = (queryLen - 1) - repPos; tp = (mLen - 1) - mPos; qRev = true; }
                            else { qp = repPos; tp = mPos; qRev = false; }
                            qp = (int) (short) qp; tp = (int) (short) tp;     // positions are truncated to T exactly like the reference's T queryPos/targetPos
                            diagonal = qp - tp;
                            rId = qRev ? (rId & ~BIT63) : (rId | BIT63);
                        }
                        const bool canBeExtended = diagonal < 0 || (diagonal > (queryLen - mLen));
                        const bool cov = canBeCoveredK(a.covThr, a.covMode, (float) queryLen, (float) mLen);
                        keep = (!a.includeOnlyExtendable && cov) || (canBeExtended && a.includeOnlyExtendable);
                        o.kmer = rId; o.id = r.id; o.len = r.len; o.pos = (int16_t) diagonal;
                        if (NUCL) embedOrd(o, r.kmer);
                        if (keep) maxRT = max(maxRT, (unsigned long long) (((rId & ~BIT63) << 32) | (unsigned long long) r.id));
                    }
                }
            }
            const unsigned long long mk = __ballot(keep);
            const uint32_t wr = (uint32_t) __popcll(mk & ((1ULL << laneId()) - 1ULL));
            uint32_t wbase = 0;
            if (mk) {
                if (laneId() == 0) wbase = atomicAdd(&sCursor[par], (uint32_t) __popcll(mk));
                wbase = __shfl(wbase, 0, 64);
            }
            if (keep) out[arena + written + wbase + wr] = o;
        };
        uint32_t nSub = 1;                           // sub-passes by a secondary hash when too many distinct k-mers
        const unsigned long long writtenAtBucketStart = written;
        for (;;) {
            bool redo = false;
            for (uint32_t sub = 0; sub < nSub && !redo; sub++) {
                for (uint32_t i = threadIdx.x; i < HT; i += BLOCK) { hKey[i] = ~0ULL; hBest[i] = ~0ULL; if (i < HT / 32) hMulti[i] = 0; }
                if (threadIdx.x == 0) { sFlag[0] = 0; sFlag[1] = 0; sCursor[par] = 0; }   // (the other cursor may still be being read)
                __syncthreads();
                if (inRegs) {
#pragma unroll
                    for (int j = 0; j < GL_RMAX; j++) if ((uint32_t) j * BLOCK < n) phaseA(rg[j], nSub, sub);
                } else for (uint32_t i0 = 0; i0 < n; i0 += BLOCK) { const uint32_t i = i0 + threadIdx.x; phaseA(i < n ? recAt(i) : none, nSub, sub); }
                __syncthreads();
                if (sFlag[1] || sFlag[0] > MAXKEYS) { redo = true; __syncthreads(); break; }
                if (inRegs) {
#pragma unroll
                    for (int j = 0; j < GL_RMAX; j++) if ((uint32_t) j * BLOCK < n) phaseC(rg[j], nSub, sub);
                } else for (uint32_t i0 = 0; i0 < n; i0 += BLOCK) { const uint32_t i = i0 + threadIdx.x; phaseC(i < n ? recAt(i) : none, nSub, sub); }
                if (sub + 1 == nSub) { fetch(b + 1); fetched = true; }     // last sub-pass: nothing reads this bucket's registers again
                __syncthreads();
                written += sCursor[par];
                par ^= 1u;
            }
            if (!redo) break;
            nSub *= 2;                               // a retry discards what completed sub-passes of this attempt wrote
            written = writtenAtBucketStart;
            __syncthreads();
        }
        if (!fetched) fetch(b + 1);                  // (cannot happen: the last sub-pass always completes; kept for the invariant)
    }
    if (a.maxRepTarget) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxRT = max(maxRT, (unsigned long long) __shfl_xor(maxRT, o, 64));
        if (laneId() == 0 && maxRT) atomicMax(a.maxRepTarget, maxRT);
    }
    if (threadIdx.x == 0) a.outCount[blockIdx.x] = written;
}

// =====================================================================================================
// 5. sort #2 + run reduction, per rep-range bucket.
//    compareRepSequenceAndIdAndDiag[Reverse] (kmermatcher.h:98-130) orders by (rep, target, diagonal); what
//    writeKmerMatcherResult (kmermatcher.cpp:835-923) needs from that order is, per (rep,target), the multiset of
//    diagonals in ascending order.  Overlapping reads share many k-mers on ONE diagonal (N_m/N_c ~ 3..10), so the
//    bucket is first aggregated in an LDS hash table to unique (rep,target,diagonal) triples with multiplicities,
//    and only the triples are bitonic-sorted (packed 64-bit keys).  Buckets that do not fit sort all their packed
//    keys in HBM scratch and run-length encode them.  Output: weighted triples in global (rep,target,diagonal) order.
// =====================================================================================================
constexpr int LS_BLOCK = 256;
constexpr uint32_t AGG_CAP = 1024;          // records per bucket handled in LDS (=> at most 1024 distinct triples)
constexpr uint32_t AGG_HT = 2048;           // hash slots
constexpr uint32_t AGG_NEEDS_SCRATCH = 0xFFFFFFFFu;   // uniqueCount of a bucket pass 1 left to the chunked pass
template <bool LONG> struct DiagPack { static constexpr int BITS = LONG ? 22 : 16; static constexpr int64_t BIAS = LONG ? (1 << 21) : 32768; };
struct __attribute__((aligned(16))) Triple { uint32_t rep, target; int32_t diag; uint32_t cnt; };   // cnt bit 31 (nucleotides): the run's member with the largest k-mer is forward-strand
// sharded nucleotide run, exchange 2: a triple some rank aggregated from ITS k-mer buckets, with the (k-mer << 1 | strand) word of its
// top-ranked member — the owner merges the ranks' partial triples and needs it to name the strand of the whole triple.  Laid out
// like Rec<true> (24 bytes: the line store moves it as such; the first 8 bytes are rep | target << 32 like a Triple's).
struct __attribute__((aligned(8))) TripleX { uint32_t rep, target; int32_t diag; uint32_t cnt; uint64_t ord; };
static_assert(sizeof(TripleX) == sizeof(Rec<true>) && sizeof(Triple) == sizeof(Rec<false>), "triples travel through the line store as records");

// LINES: bucket b = the lines list[lineBeg[b] .. + lineCnt[b]) of `arr` (linepart.hpp); its triples go to outTriples[lineBeg[b] * RPL ...]
struct AggLines { const uint32_t *list, *lineBeg, *lineCnt; };
__device__ __forceinline__ bool isSentinel(const Triple &t) { return t.rep == 0xFFFFFFFFu && t.target == 0xFFFFFFFFu; }   // a padding slot of a line of triples
__device__ __forceinline__ bool isSentinel(const TripleX &t) { return t.rep == 0xFFFFFFFFu && t.target == 0xFFFFFFFFu; }
// TRIPLES (sharded run, owner side): the input elements are weighted triples other ranks aggregated from THEIR k-mer buckets (Triple;
// TripleX for nucleotides); equal (rep, target, diagonal) triples of several ranks merge here: counts add, the larger (k-mer, strand) word wins.
// ORDOUT (sharded nucleotide run, every rank's own rep sort): the output elements are TripleX.
// A representative is keyed by (rep - repBase) [bit-reversed over scrambleBits when != 0] relative to its bucket's first key.
template <bool NUCL, bool LONG, bool LINES, bool TRIPLES = false, bool ORDOUT = false>
__global__ __launch_bounds__(LS_BLOCK) void aggSortKernel(const void *arr, void *outTriples, const uint64_t *__restrict__ bucketStart, uint32_t nBuckets,
                                                          unsigned long long *bigScratch, const uint64_t *__restrict__ bigOff,
                                                          uint32_t *__restrict__ uniqueCount, int localBits, int idBits, uint64_t repBase, AggLines ln, int scrambleBits) {
    static_assert(!ORDOUT || NUCL, "only nucleotide triples carry a strand");
    typedef typename std::conditional<TRIPLES, typename std::conditional<NUCL, TripleX, Triple>::type, Rec<LONG>>::type R;
    typedef typename std::conditional<ORDOUT, TripleX, Triple>::type O;
    __shared__ unsigned long long hKey[AGG_HT];
    __shared__ uint32_t hVal[AGG_HT];
    __shared__ unsigned long long hOrd[NUCL ? AGG_HT : 1];       // (k-mer << 1 | forward) of the slot's top-ranked member
    __shared__ unsigned long long lKey[AGG_CAP];
    __shared__ uint32_t lVal[AGG_CAP];
    __shared__ unsigned long long lOrd[NUCL ? AGG_CAP : 1];
    __shared__ uint32_t sCount;
    __shared__ uint32_t sDistinct, sOver;
    __shared__ uint32_t sWave[LS_BLOCK / 64];
    const R *g = reinterpret_cast<const R *>(arr);
    O *out = reinterpret_cast<O *>(outTriples);
    constexpr int DB = DiagPack<LONG>::BITS;
    for (uint32_t b = blockIdx.x; b < nBuckets; b += gridDim.x) {
        // cnt record positions; LINES: positions in the bucket's line list, padding sentinels are skipped when read
        const uint64_t s0 = LINES ? (uint64_t) ln.lineBeg[b] * RPL : bucketStart[b];        // where the bucket's triples are written
        const uint64_t cnt = LINES ? (uint64_t) ln.lineCnt[b] * RPL : bucketStart[b + 1] - s0;
        const uint32_t lb = LINES ? ln.lineBeg[b] : 0u;
        auto recAt = [&](uint64_t i) -> R { if (LINES) return g[(uint64_t) ln.list[lb + (uint32_t) (i / RPL)] * RPL + (i % RPL)]; return g[s0 + i]; };
        if (cnt == 0) { if (threadIdx.x == 0) uniqueCount[b] = 0; continue; }
        const uint64_t bucketBase = (uint64_t) b << localBits;             // first (relative, possibly bit-reversed) rep key of the bucket
        auto decode = [&](unsigned long long key, uint32_t val, unsigned long long ord) {
            O t;
            t.diag = (int32_t) ((int64_t) (key & ((1ULL << DB) - 1)) - DiagPack<LONG>::BIAS);
            const uint64_t k2 = key >> DB;
            t.target = (uint32_t) (k2 & ((1ULL << idBits) - 1));
            const uint64_t rp = (k2 >> idBits) + bucketBase;
            t.rep = (uint32_t) ((scrambleBits ? scrambleRep(rp, scrambleBits) : rp) + repBase);       // an involution: back to the id
            t.cnt = val;
            if constexpr (ORDOUT) t.ord = ord;
            return t;
        };
        // packed sort key, count and (nucleotides) rank word of one input element
        auto packRec = [&](const R &r, uint32_t &val, unsigned long long &ord) {
            uint64_t rep, target; int64_t diag; ord = 0;
            if constexpr (TRIPLES) { rep = r.rep; target = r.target; diag = r.diag; val = r.cnt & 0x7FFFFFFFu; if constexpr (NUCL) ord = r.ord; }
            else { rep = (uint32_t) r.kmer; target = r.id; diag = r.pos; val = 1u; if constexpr (NUCL) ord = ordWordOf(r); }     // (nucleotides: bits 32..62 of the rep field hold k-mer bits, embedOrd)
            rep -= repBase;                                                    // repBase: first rep of this rank's range (sharded run, owner side), else 0
            if (scrambleBits) rep = scrambleRep(rep, scrambleBits);            // buckets are ranges of the bit-reversed id (linepart.hpp)
            return (unsigned long long) (((((rep - bucketBase) << idBits) | target) << DB) | (uint64_t) (diag + DiagPack<LONG>::BIAS));
        };
        auto clearTable = [&]() {
            for (uint32_t i = threadIdx.x; i < AGG_HT; i += LS_BLOCK) { hKey[i] = ~0ULL; hVal[i] = 0; if (NUCL) hOrd[i] = 0; }
            if (threadIdx.x == 0) sCount = 0;
            __syncthreads();
        };
        auto insert = [&](unsigned long long key, uint32_t val, unsigned long long ord) {      // counts add, the larger rank word wins
            uint32_t slot = (uint32_t) ((key * 0x9E3779B97F4A7C15ULL) >> 40) & (AGG_HT - 1);
            for (;;) {
                const unsigned long long prev = atomicCAS(&hKey[slot], ~0ULL, key);
                if (prev == ~0ULL || prev == key) break;
                slot = (slot + 1) & (AGG_HT - 1);
            }
            atomicAdd(&hVal[slot], val);
            if (NUCL) atomicMax(&hOrd[slot], ord);
        };
        // table -> lKey/lVal/lOrd (unordered); returns the number of distinct keys (block-uniform).  lVal: count | strand of the top-ranked member << 31
        auto extract = [&]() {
            for (uint32_t i = threadIdx.x; i < AGG_HT; i += LS_BLOCK) {
                const unsigned long long k = hKey[i];
                if (k != ~0ULL) {
                    const uint32_t o = atomicAdd(&sCount, 1u); lKey[o] = k;
                    if (NUCL) { const unsigned long long od = hOrd[i]; lOrd[o] = od; lVal[o] = hVal[i] | ((uint32_t) (od & 1ULL) << 31); }
                    else lVal[o] = hVal[i];
                }
            }
            __syncthreads();
            return sCount;
        };
        auto sortListAndWrite = [&](uint32_t U) {
            uint32_t P = 1; while (P < U) P <<= 1;
            for (uint32_t i = U + threadIdx.x; i < P; i += LS_BLOCK) { lKey[i] = ~0ULL; lVal[i] = 0; if (ORDOUT) lOrd[i] = 0; }
            __syncthreads();
            for (uint32_t kk = 2; kk <= P; kk <<= 1) {
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    for (uint32_t t = threadIdx.x; t < (P >> 1); t += LS_BLOCK) {
                        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const uint32_t l = i | j;
                        const unsigned long long x = lKey[i], y = lKey[l];
                        const bool up = (i & kk) == 0;
                        if ((x > y) == up) {
                            lKey[i] = y; lKey[l] = x; const uint32_t vx = lVal[i]; lVal[i] = lVal[l]; lVal[l] = vx;
                            if (ORDOUT) { const unsigned long long ox = lOrd[i]; lOrd[i] = lOrd[l]; lOrd[l] = ox; }
                        }
                    }
                    __syncthreads();
                }
            }
            for (uint32_t i = threadIdx.x; i < U; i += LS_BLOCK) out[s0 + i] = decode(lKey[i], lVal[i], ORDOUT ? lOrd[i] : 0ull);
            if (threadIdx.x == 0) uniqueCount[b] = U;
            __syncthreads();
        };
        // pass 2 (bigScratch != nullptr) only revisits the buckets pass 1 could not aggregate in LDS
        if (bigScratch && uniqueCount[b] != AGG_NEEDS_SCRATCH) continue;
        // What bounds the LDS path is the number of DISTINCT (rep, target, diagonal) triples, not the number of records: overlapping
        // reads share many k-mers on one diagonal (N_m / N_c is 3..14), so a bucket of several thousand records usually holds a few
        // hundred triples.  Every bucket is therefore first aggregated straight into the table; only when more than AGG_CAP
        // distinct triples turn up is it left to the chunked path below (second launch, with HBM scratch).
        bool done = false;
        if (!bigScratch && cnt <= (1ull << 22)) {
            clearTable();
            if (threadIdx.x == 0) { sDistinct = 0; sOver = 0; }
            __syncthreads();
            for (uint64_t i = threadIdx.x; i < cnt && !*(volatile uint32_t *) &sOver; i += LS_BLOCK) {
                const R r = recAt(i); if (LINES && isSentinel(r)) continue;
                uint32_t v; unsigned long long od; const unsigned long long key = packRec(r, v, od);
                uint32_t slot = (uint32_t) ((key * 0x9E3779B97F4A7C15ULL) >> 40) & (AGG_HT - 1);
                bool placed = false;
                for (uint32_t probe = 0; probe < AGG_HT; probe++) {
                    const unsigned long long prev = atomicCAS(&hKey[slot], ~0ULL, key);
                    if (prev == ~0ULL) { if (atomicAdd(&sDistinct, 1u) >= AGG_CAP) atomicExch(&sOver, 1u); placed = true; break; }
                    if (prev == key) { placed = true; break; }
                    slot = (slot + 1) & (AGG_HT - 1);
                }
                if (!placed) { atomicExch(&sOver, 1u); break; }
                atomicAdd(&hVal[slot], v);
                if (NUCL) atomicMax(&hOrd[slot], od);
            }
            __syncthreads();
            if (!sOver) { sortListAndWrite(extract()); done = true; }
            __syncthreads();
        }
        if (done) continue;
        if (!bigScratch) { if (threadIdx.x == 0) uniqueCount[b] = AGG_NEEDS_SCRATCH; continue; }
        {
            // oversized bucket (hot representatives): aggregate chunk by chunk in LDS, spill the partial (key, count[, rank word])
            // entries to HBM scratch, then merge the partials — in LDS again when they fit, else by sorting them in HBM
            unsigned long long *pk = bigScratch + bigOff[b];           // [2 * P0 (protein) or 3 * P0 (nucleotides)]: keys, values, rank words
            uint64_t P0 = 1; while (P0 < cnt) P0 <<= 1;
            unsigned long long *pv = pk + P0, *po = pv + P0;
            uint64_t nPart = 0;
            for (uint64_t c0 = 0; c0 < cnt; c0 += AGG_CAP) {
                const uint64_t c1 = min(cnt, c0 + (uint64_t) AGG_CAP);
                clearTable();
                for (uint64_t i = c0 + threadIdx.x; i < c1; i += LS_BLOCK) { const R r = recAt(i); if (LINES && isSentinel(r)) continue; uint32_t v; unsigned long long od; const unsigned long long key = packRec(r, v, od); insert(key, v, od); }
                __syncthreads();
                const uint32_t U = extract();
                for (uint32_t i = threadIdx.x; i < U; i += LS_BLOCK) { pk[nPart + i] = lKey[i]; pv[nPart + i] = lVal[i] & 0x7FFFFFFFu; if (NUCL) po[nPart + i] = lOrd[i]; }
                nPart += U;
                __syncthreads();
            }
            // second level: distinct keys among the partials
            bool fits = true;
            if (nPart <= (uint64_t) AGG_HT) {
                clearTable();
                for (uint64_t i = threadIdx.x; i < nPart; i += LS_BLOCK) insert(pk[i], (uint32_t) pv[i], NUCL ? po[i] : 0ull);
                __syncthreads();
                // count distinct keys before extracting (the list holds AGG_CAP entries)
                uint32_t mine = 0;
                for (uint32_t i = threadIdx.x; i < AGG_HT; i += LS_BLOCK) mine += (hKey[i] != ~0ULL) ? 1u : 0u;
                mine = (uint32_t) waveReduceSum((int) mine);
                if (laneId() == 0) sWave[threadIdx.x >> 6] = mine;
                __syncthreads();
                uint32_t tot = 0;
#pragma unroll
                for (int w = 0; w < LS_BLOCK / 64; w++) tot += sWave[w];
                __syncthreads();
                fits = tot <= AGG_CAP;
                if (fits) sortListAndWrite(extract());
            } else fits = false;
            if (!fits) {
                // sort the partial entries by key in HBM scratch, then merge runs of equal keys
                uint64_t P = 1; while (P < nPart) P <<= 1;
                for (uint64_t i = nPart + threadIdx.x; i < P; i += LS_BLOCK) { pk[i] = ~0ULL; pv[i] = 0; if (NUCL) po[i] = 0; }
                __syncthreads();
                for (uint64_t kk = 2; kk <= P; kk <<= 1) {
                    for (uint64_t j = kk >> 1; j > 0; j >>= 1) {
                        for (uint64_t t = threadIdx.x; t < (P >> 1); t += LS_BLOCK) {
                            const uint64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                            const uint64_t l = i | j;
                            const unsigned long long x = pk[i], y = pk[l];
                            const bool up = (i & kk) == 0;
                            if ((x > y) == up) {
                                pk[i] = y; pk[l] = x; const unsigned long long vx = pv[i]; pv[i] = pv[l]; pv[l] = vx;
                                if (NUCL) { const unsigned long long ox = po[i]; po[i] = po[l]; po[l] = ox; }
                            }
                        }
                        __syncthreads();
                    }
                }
                uint32_t written = 0;
                for (uint64_t i0 = 0; i0 < nPart; i0 += LS_BLOCK) {
                    const uint64_t i = i0 + threadIdx.x;
                    bool head = false; O t; t.rep = t.target = t.cnt = 0; t.diag = 0;
                    if (i < nPart) {
                        const unsigned long long k = pk[i];
                        head = (i == 0) || (pk[i - 1] != k);
                        if (head) {
                            uint32_t c = 0; unsigned long long om = 0;
                            for (uint64_t j = i; j < nPart && pk[j] == k; j++) { c += (uint32_t) pv[j]; if (NUCL) om = max(om, po[j]); }
                            t = decode(k, c | ((uint32_t) (om & 1ULL) << 31), om);
                        }
                    }
                    const unsigned long long mk = __ballot(head);
                    const uint32_t wr = (uint32_t) __popcll(mk & ((1ULL << laneId()) - 1ULL));
                    if (laneId() == 0) sWave[threadIdx.x >> 6] = (uint32_t) __popcll(mk);
                    __syncthreads();
                    uint32_t woff = 0, tot = 0;
#pragma unroll
                    for (int w = 0; w < LS_BLOCK / 64; w++) { if (w < (int) (threadIdx.x >> 6)) woff += sWave[w]; tot += sWave[w]; }
                    if (head) out[s0 + written + woff + wr] = t;
                    written += tot;
                    __syncthreads();
                }
                if (threadIdx.x == 0) uniqueCount[b] = written;
                __syncthreads();
            }
        }
    }
}

// line-store path: the triples of bucket b lie at in[lineBeg[b] * RPL ...] (unique[b] of them), grouped by representative.  Every
// representative occurs in exactly one bucket: the number of its triples and the position of the first go to cnt[rep - repBase] /
// pos[rep - repBase] (cnt is zeroed beforehand).  One wavefront per bucket; a representative's triples are counted 64 at a time
// (a long contig is the representative of 10^5..10^6 triples: neither a serial walk per run nor one atomic per triple would do).
template <class T>
__global__ __launch_bounds__(256) void repRunsKernel(const T *__restrict__ in, const uint32_t *__restrict__ lineBeg, const uint32_t *__restrict__ unique, uint32_t nBuckets,
                                                     uint32_t repBase, uint32_t *__restrict__ cnt) {
    for (uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6); b < nBuckets; b += gridDim.x * 4) {
        const uint64_t s0 = (uint64_t) lineBeg[b] * RPL; const uint32_t n = unique[b];
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + laneId();
            const bool valid = i < n;
            const uint32_t rep = valid ? in[s0 + i].rep : 0xFFFFFFFFu;
            const uint32_t prevLane = __shfl_up(rep, 1, 64);
            const bool segHead = valid && (laneId() == 0 || prevLane != rep);            // first of its triples within these 64
            const unsigned long long heads = __ballot(segHead), vmask = __ballot(valid);
            if (segHead) {
                const unsigned long long later = heads & ~((2ULL << laneId()) - 1ULL);   // heads behind this lane
                const int end = later ? __ffsll((long long) later) - 1 : (int) __popcll(vmask);
                atomicAdd(&cnt[rep - repBase], (uint32_t) (end - laneId()));
            }
        }
    }
}
// every triple to its place in representative order: start[rep - repBase] + its offset within the representative's run.  The
// offset comes from the run heads among the 64 triples a wavefront holds (a run that began earlier is carried along as a
// wave-uniform pair), so the only random access per representative is its start (round 3: an array of run positions, written by
// the kernel above and read here, was a second random line per representative in both kernels).
template <class T>
__global__ __launch_bounds__(256) void placeRunsKernel(const T *__restrict__ in, const uint32_t *__restrict__ lineBeg, const uint32_t *__restrict__ unique, uint32_t nBuckets,
                                                       uint32_t repBase, const uint64_t *__restrict__ start, T *__restrict__ out) {
    const int lane = laneId();
    for (uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6); b < nBuckets; b += gridDim.x * 4) {
        const uint64_t s0 = (uint64_t) lineBeg[b] * RPL; const uint32_t n = unique[b];
        uint32_t carryRep = 0xFFFFFFFFu, carryHead = 0;           // the run that reaches into these 64 from the left: its representative, the bucket index of its first triple
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + (uint32_t) lane;
            const bool valid = i < n;
            T t; t.rep = 0xFFFFFFFFu; t.target = 0; t.diag = 0; t.cnt = 0;
            if (valid) t = in[s0 + i];
            const uint32_t rep = t.rep;
            const uint32_t prevLane = __shfl_up(rep, 1, 64);
            const bool segHead = valid && (lane == 0 || prevLane != rep);
            const unsigned long long heads = __ballot(segHead), vmask = __ballot(valid);
            const unsigned long long below = heads & ((lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL));       // heads at or before this lane (lane 0 is one)
            const int headLane = 63 - __clzll((long long) below);
            const uint32_t rep0 = (uint32_t) __shfl((int) rep, 0, 64);
            const uint32_t runHead = (headLane == 0 && rep0 == carryRep) ? carryHead : i0 + (uint32_t) headLane;
            if (valid) out[start[rep - repBase] + (uint64_t) (i - runHead)] = t;
            const int lastLane = (int) __popcll(vmask) - 1;
            carryRep = (uint32_t) __shfl((int) rep, lastLane, 64); carryHead = (uint32_t) __shfl((int) runHead, lastLane, 64);
        }
    }
}
// sharded run: the first triple of every rank's share of the (rep-ordered) triples: out[r] = start[first rep rank r owns], r = 0..W
__global__ void ownerBoundsKernel(const uint64_t *__restrict__ start, uint64_t n, int W, uint64_t *__restrict__ out) {
    for (int r = threadIdx.x; r <= W; r += blockDim.x) out[r] = start[((uint64_t) r * n + (uint64_t) W - 1) / (uint64_t) W];
}
// sharded run: the lines of the level-1 buckets in list order, packed for the exchange (16-byte chunks; a line is 128 or 192 bytes)
__global__ __launch_bounds__(256) void gatherLinesKernel(const uint4 *__restrict__ in, const uint32_t *__restrict__ list, uint64_t nLines, uint32_t chunksPerLine, uint4 *__restrict__ out) {
    const uint64_t total = nLines * chunksPerLine;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t) gridDim.x * blockDim.x)
        out[i] = in[(uint64_t) list[i / chunksPerLine] * chunksPerLine + (i % chunksPerLine)];
}
// sharded run: line list over the receive buffer — the lines of own bucket j from source s are a contiguous run of it
struct RxSeg { uint64_t src; uint32_t cnt, dst; };      // first line in the receive buffer, lines, first position in the list
__global__ __launch_bounds__(256) void rxListKernel(const RxSeg *__restrict__ segs, uint32_t nSegs, uint32_t *__restrict__ list) {
    for (uint32_t q = blockIdx.x; q < nSegs; q += gridDim.x) { const RxSeg g = segs[q]; for (uint32_t i = threadIdx.x; i < g.cnt; i += 256) list[g.dst + i] = (uint32_t) (g.src + i); }
}

// =====================================================================================================
// 6. best diagonal per (rep, target) run (writeKmerMatcherResult, kmermatcher.cpp:835-923) over weighted triples
// =====================================================================================================
template <bool NUCL>
__global__ __launch_bounds__(256) void reduceRunsKernel(const Triple *__restrict__ h, uint64_t n, uint64_t nScan, CandHit *__restrict__ tmpHits, uint32_t *__restrict__ emitW,
                                 uint32_t *__restrict__ perRep) {
    // h[n .. nScan): what follows this rank's triples in the global (rep, target, diagonal) order as far as the last run's scan
    // can reach (sharded run: the head of the next ranks' triples and the stale records; nScan == n otherwise)
    // A wavefront takes 64 consecutive triples; the candidates it emits are packed to the front of ITS 64 slots of tmpHits (one
    // coalesced store per wavefront instead of 16-byte stores scattered over the slots of the run heads) and counted once per
    // wavefront (emitW[i / 64]); the triples are in id order, so the candidates of one representative are counted with one atomic per
    // wavefront.  (Round 3: the kernel's memory pipe was busy all the time with partial-line writes: profiles/r03_pmc.)
    const int lane = threadIdx.x & 63;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const Triple r = h[i];
        bool head = (i == 0);
        if (!head) { const Triple q = h[i - 1]; head = (q.rep != r.rep) || (q.target != r.target); }
        uint32_t e = 0;
        CandHit c; c.target = 0; c.prefScore = 0; c.diag16 = 0; c.query = 0;
        if (head) {
            const uint32_t targetId = r.target;
            int32_t diagonal = r.diag, prevDiagonal = r.diag;
            uint64_t maxDiagonal = 0, diagonalCnt = 0, topScore = 0;
            int bestRev = NUCL ? ((r.cnt & 0x80000000u) == 0) : 0;
            // NOTE: the reference's scan tests only the target id, so it runs across a rep boundary when the
            // next rep starts with the same target (Appendix A.3) — reproduced (a run of equal diagonals then
            // continues across the boundary); it can also run past the compaction point into stale sort-#1
            // records (probability ~1/N per run) — not reproduced.
            for (uint64_t j = i; j < nScan; j++) {
                const Triple x = h[j];
                if (x.target != targetId) break;
                const uint64_t cc = x.cnt & 0x7FFFFFFFu;
                if (prevDiagonal == x.diag) diagonalCnt += cc; else diagonalCnt = cc;
                // every record of the run is checked against the running maximum; the count only grows inside a
                // run, so the state after the run is what the record-by-record walk leaves behind
                if (diagonalCnt >= maxDiagonal) { diagonal = x.diag; maxDiagonal = diagonalCnt; if (NUCL) bestRev = ((x.cnt & 0x80000000u) == 0); }
                prevDiagonal = x.diag; topScore += cc;
            }
            if (targetId != r.rep) {
                c.target = targetId; c.prefScore = bestRev ? -(int) topScore : (int) topScore;
                c.diag16 = (uint32_t) (uint16_t) diagonal; c.query = r.rep;
                e = 1;
            }
        }
        // (the lanes of a wavefront leave the loop together except in its last round, where the active ones are the low lanes)
        const unsigned long long act = __ballot(1), em = __ballot(e != 0);
        const uint64_t i0 = i - (uint64_t) lane;
        if (e) tmpHits[i0 + (uint64_t) __popcll(em & ((1ULL << lane) - 1ULL))] = c;
        if (lane == 0) emitW[i0 >> 6] = (uint32_t) __popcll(em);
        const uint32_t prevRep = (uint32_t) __shfl_up((int) r.rep, 1, 64);
        const bool segHead = lane == 0 || prevRep != r.rep;
        const unsigned long long hm = __ballot(segHead);
        if (segHead) {
            const unsigned long long above = lane == 63 ? 0ULL : (hm & ~((2ULL << lane) - 1ULL));
            const int end = above ? __ffsll((long long) above) - 1 : 64;
            const unsigned long long seg = (end == 64 ? ~0ULL : ((1ULL << end) - 1ULL)) & ~((1ULL << lane) - 1ULL) & act;
            const uint32_t cnt = (uint32_t) __popcll(em & seg);
            if (cnt) atomicAdd(&perRep[r.rep], cnt);
        }
    }
}

// =====================================================================================================
// 7. The reference's run scan does not stop at the compaction point of assignGroup: if the sort-#1 record that
//    happens to sit right behind it belongs to the target of the very last (rep,target) run, it is counted too
//    (kmermatcher.cpp:880-898).  Those "stale" records are the sort-#1 records of rank
//    N_m, N_m+1, ...  This path has no k-mer-sorted array, so the rank of every record of that one target is
//    counted directly: one streaming pass over the N_k records.
// =====================================================================================================
template <bool NUCL, bool LONG> __host__ __device__ __forceinline__ bool recLess1(const Rec<LONG> &a, const Rec<LONG> &b) {   // kmermatcher.h:56-96
    const uint64_t ak = NUCL ? (a.kmer | BIT63) : a.kmer, bk = NUCL ? (b.kmer | BIT63) : b.kmer;
    if (ak != bk) return ak < bk;
    if (a.len != b.len) return a.len > b.len;
    if (a.id != b.id) return a.id < b.id;
    if (a.pos != b.pos) return a.pos < b.pos;
    return a.kmer < b.kmer;      // records of one sequence that differ in the strand only: reverse first (as oracle/kmermatcher.cpp)
}
// over the line store: every written line (tag != TAG_NONE) of the hash-partitioned records, padding sentinels skipped
template <bool NUCL, bool LONG>
__global__ __launch_bounds__(256) void rankLinesKernel(const void *recs, const uint32_t *__restrict__ tags, uint64_t nLines, const uint64_t *__restrict__ nLinesDev,
                                                       const void *tkeys, uint32_t m, unsigned long long *diff) {
    typedef Rec<LONG> R;
    if (nLinesDev) nLines = min(nLines, (uint64_t) *nLinesDev);        // lines the last partition level laid out (the rest of the tag array was never written)
    const R *g = reinterpret_cast<const R *>(recs);
    const R *tk = reinterpret_cast<const R *>(tkeys);
    __shared__ uint32_t sDiff[1025];
    const bool useLds = m <= 1024;
    if (useLds) { for (uint32_t i = threadIdx.x; i <= m; i += 256) sDiff[i] = 0; __syncthreads(); }
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nLines * RPL; i += (uint64_t) gridDim.x * 256) {
        if (tags && tags[i / RPL] == TAG_NONE) continue;                // tags == nullptr: every line is valid (received lines of a sharded run)
        const R r = g[i];
        if (isSentinel(r)) continue;
        uint32_t lo = 0, hi = m;                      // first j with r < tk[j]
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (recLess1<NUCL, LONG>(r, tk[mid])) hi = mid; else lo = mid + 1; }
        if (useLds) atomicAdd(&sDiff[lo], 1u); else atomicAdd(&diff[lo], 1ULL);
    }
    if (useLds) { __syncthreads(); for (uint32_t i = threadIdx.x; i <= m; i += 256) { const uint32_t c = sDiff[i]; if (c) atomicAdd(&diff[i], (unsigned long long) c); } }
}

__global__ void fillU32Kernel(uint32_t *p, uint32_t v, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) p[i] = v;
}
__global__ void placeHitsKernel(const CandHit *__restrict__ tmpHits, const uint64_t *__restrict__ eposW,
                                uint64_t n, uint32_t qLo, CandHit *__restrict__ hits) {
    // the candidates of the 64 triples [64 w, 64 w + 64) sit packed at the front of those slots of tmpHits (reduceRunsKernel)
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t w = i >> 6, l = i & 63, e0 = eposW[w];
        if (l < eposW[w + 1] - e0) { const CandHit c = tmpHits[i]; hits[e0 + l + (uint64_t) (c.query - qLo) + 1] = c; }      // self lines of queries qLo..query come first
    }
}
// queries [qLo, qHi) have a self line (all of them; sharded run: the ones this rank owns)
__global__ void placeSelfKernel(const uint64_t *__restrict__ qoff, uint32_t qLo, uint32_t qHi, CandHit *__restrict__ hits) {
    for (uint32_t q = qLo + blockIdx.x * blockDim.x + threadIdx.x; q < qHi; q += gridDim.x * blockDim.x) {
        CandHit c; c.target = q; c.prefScore = 0; c.diag16 = 0; c.query = q;
        hits[qoff[q]] = c;
    }
}

}  // namespace plasship
using namespace plasship;

// ---- host orchestration --------------------------------------------------------------------------------
namespace {

struct Timer {
    plasship_ctx *ctx; int slot;
    void start(int s) { slot = s; (void) hipEventRecord(ctx->ev[s], ctx->stream); }
    float stop(int s2) { (void) hipEventRecord(ctx->ev[s2], ctx->stream); (void) hipEventSynchronize(ctx->ev[s2]); float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[slot], ctx->ev[s2]); return ms; }
};

static inline unsigned gridFor(uint64_t n, unsigned block, unsigned cap = 65535u * 8) {
    uint64_t g = (n + block - 1) / block; if (g < 1) g = 1; if (g > cap) g = cap; return (unsigned) g;
}
static int ceilLog2(uint64_t x) { int b = 0; while ((1ULL << b) < x) b++; return b; }

// what the host needs about the target of the last run (the stale-record check), gathered on the device so that it travels with
// the group kernel's counts in one round trip: {maxRT, slotOff[T], slotOff[T+1], len[T]}
__global__ void lastRunInfoKernel(const unsigned long long *__restrict__ maxRT, const uint64_t *__restrict__ slotOff, const uint32_t *__restrict__ len,
                                  uint32_t n, unsigned long long *__restrict__ out) {
    const unsigned long long m = *maxRT;
    const uint32_t t = (uint32_t) (m & 0xFFFFFFFFull);
    out[0] = m;
    if (t < n) { out[1] = slotOff[t]; out[2] = slotOff[t + 1]; out[3] = len[t]; } else { out[1] = out[2] = out[3] = 0; }
}

// PLASSHIP_TRACE: records whose sequence id is out of range (sentinels excepted) — a consistency probe between the stages
template <bool LONG>
__global__ void countBadIdsKernel(const void *recs, uint64_t n, uint32_t nSeq, unsigned long long *out) {
    const Rec<LONG> *g = reinterpret_cast<const Rec<LONG> *>(recs);
    unsigned long long bad = 0, sen = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const Rec<LONG> r = g[i];
        if (isSentinel(r)) sen++; else if (r.id >= nSeq) bad++;
    }
    if (bad) atomicAdd(&out[0], bad);
    if (sen) atomicAdd(&out[1], sen);
}
template <bool LONG>
static void traceBadIds(plasship_ctx *ctx, const char *what, const void *recs, uint64_t n, uint32_t nSeq) {
    if (!traceOn()) return;
    DevBuf d; unsigned long long h[2] = {0, 0};
    if (d.alloc(16) != hipSuccess) return;
    (void) hipMemsetAsync(d.p, 0, 16, ctx->stream);
    hipLaunchKernelGGL((countBadIdsKernel<LONG>), dim3(1024), dim3(256), 0, ctx->stream, recs, n, nSeq, d.as<unsigned long long>());
    (void) hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e = plasship::streamSync(ctx->stream);
    fprintf(stderr, "[plasship] %s: %llu records, %llu with an id out of range, %llu sentinels (%s)\n", what, (unsigned long long) n, h[0], h[1], hipGetErrorString(e));
}

// sharded run: id ranges with equal shares of the k-mer record slots.  out[r] = first id of rank r (r = 0..W), out[W+1+r] = its slot
__global__ void splitIdsKernel(const uint64_t *__restrict__ slotOff, uint32_t n, int W, uint64_t *__restrict__ out) {
    const uint64_t total = slotOff[n];
    for (int r = threadIdx.x; r <= W; r += blockDim.x) {
        uint32_t lo = 0, hi = n;
        if (r == W) lo = n;
        else {
            const uint64_t want = (uint64_t) (((unsigned __int128) total * (unsigned) r) / (unsigned) W);
            while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (slotOff[mid] < want) lo = mid + 1; else hi = mid; }
        }
        out[r] = lo; out[W + 1 + r] = slotOff[lo];
    }
}

// =====================================================================================================
// single GPU: hash grouping and rep sort over the line store (linepart.hpp)
// =====================================================================================================
// geometry of the k-mer side, fixed by the number of record slots (known before the extraction): partition levels, piece sizes,
// and how many lines the two record buffers must hold
struct LineGeo {
    int b1 = 0, b2 = 0; uint32_t nb1 = 1, nb2 = 0, PL1 = 1, PL2 = 0;
    uint64_t totalLines = 0, nP1 = 0, cap1 = 0, maxP2 = 0, cap2 = 0;
    uint32_t lastValid = RPL;
};
static uint32_t pieceLinesFor(uint64_t lines, uint32_t nb, int numCU, uint32_t minFactor) {
    // a piece leaves one partial line per bucket: >= minFactor * nb lines per piece bounds that waste; beyond that, enough pieces
    // to give every CU a few
    const uint64_t want = (lines + 4ull * (uint64_t) numCU - 1) / (4ull * (uint64_t) numCU);
    return (uint32_t) std::max<uint64_t>((uint64_t) nb * minFactor, std::min<uint64_t>((uint64_t) nb * 64, std::max<uint64_t>(want, 1)));
}
// totalSlots: slots of THIS rank's sequences (what level 1 reads); sharded run: slotsAll = slots of all ranks — the bucket bits are
// those of the whole run, the same on every rank, and there are at least W level-1 buckets (rank r owns a contiguous range of them)
static LineGeo lineGeometry(uint64_t totalSlots, bool lng, int numCU, uint64_t slotsAll = 0, int W = 1) {
    LineGeo g;
    const int maxBits = lng ? 9 : 10;                         // LDS: 2^bits open lines of RPL records
    const int lw = ceilLog2((uint64_t) std::max(W, 1));
    const int totalBits = std::min(2 * maxBits, std::max(lw, std::max(0, ceilLog2(((slotsAll ? slotsAll : totalSlots) + 1535) / 1536))));   // ~1000-1500 records per final bucket
    g.b1 = totalBits <= maxBits ? totalBits : std::max((totalBits + 1) / 2, lw); g.b2 = totalBits - g.b1;
    g.nb1 = 1u << g.b1; g.nb2 = g.b2 ? 1u << g.b2 : 0u;
    g.totalLines = (totalSlots + RPL - 1) / RPL;
    g.lastValid = g.totalLines ? (uint32_t) (totalSlots - (g.totalLines - 1) * RPL) : (uint32_t) RPL;
    g.PL1 = pieceLinesFor(g.totalLines, g.nb1, numCU, 8);
    g.nP1 = (g.totalLines + g.PL1 - 1) / g.PL1;
    g.cap1 = std::max<uint64_t>(g.nP1 * ((uint64_t) g.PL1 + g.nb1), 1);
    if (g.nb2) {
        // level 2: ONE piece per level-1 bucket (its output range is the bucket's "region"; hash buckets are evenly filled)
        g.PL2 = 0xFFFFFFFFu; g.maxP2 = g.nb1; g.cap2 = g.cap1 + (uint64_t) g.nb1 * g.nb2;     // (sharded run: cap2 follows from what the exchange delivers)
    }
    return g;
}
// lines a range partition of `nRec` records in `nSeg` dense segments can need (level 1) and a second level on top of it
static uint64_t repLevel1Cap(uint64_t nRec, uint64_t nSeg, uint32_t PL, uint32_t nb) {
    const uint64_t lines = (nRec + RPL - 1) / RPL + nSeg;
    return lines + (lines / PL + nSeg + 1) * (uint64_t) nb;
}

__global__ void arenaStartKernel(const uint32_t *__restrict__ lineBeg, uint32_t bpb, uint32_t gGrid, uint32_t nBuckets, uint64_t *__restrict__ arenaStart) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < gGrid; j += gridDim.x * blockDim.x)
        arenaStart[j] = (uint64_t) lineBeg[std::min(j * bpb, nBuckets - 1)] * RPL;
}
// scratch need of the aggregation kernel per bucket (buckets beyond its LDS capacity): 2 (nucleotides: 3) * pow2ceil(records) 8-byte words
__global__ void bigNeedKernel(const uint32_t *__restrict__ lineCnt, const uint32_t *__restrict__ unique, uint32_t nBuckets, uint64_t *__restrict__ need, uint32_t wordsPerEntry) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nBuckets; b += gridDim.x * blockDim.x) {
        const uint64_t c = (uint64_t) lineCnt[b] * RPL;
        uint64_t v = 0;
        if (unique[b] == AGG_NEEDS_SCRATCH) { uint64_t P = 1; while (P < c) P <<= 1; v = wordsPerEntry * P; }      // only what pass 1 left over (keys, counts[, rank words])
        need[b] = v;
    }
}
__global__ void sumU32Kernel(const uint32_t *__restrict__ v, uint32_t n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += v[i];
    s = waveReduceSumU64(s);
    if (laneId() == 0 && s) atomicAdd(out, s);
}

template <class K> static int setDynLds(K k, size_t bytes) {
    PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
    return PLASSHIP_OK;
}
// workgroup geometry of the partition kernel (measured, tools/linepart_bench.hip on 2^30 records): 1024 threads x 4 records with
// the next tile prefetched for 1024 buckets (one workgroup per CU: 4.8 TB/s read + written at level 1), 512 x 4 with prefetch
// when two workgroups fit a CU (<= 512 buckets: 4.9 TB/s)
template <bool NUCL, bool LONG, int MODE, bool LIST, bool EXTRAS>
static int launchLinePart(plasship_ctx *ctx, const LinePartArgs &a, uint64_t nPiecesBound) {
    const size_t lds = linePartLdsBytes(a.nb, sizeof(Rec<LONG>), EXTRAS);
    const bool big = lds > 72 * 1024;
    const unsigned grid = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>(nPiecesBound, (uint64_t) ctx->numCU * (big ? 1u : 2u)));
    if (big) {
        auto k = linePartKernel<NUCL, LONG, MODE, LIST, EXTRAS, 1024, 4, true>;
        const int rc = setDynLds(k, lds); if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, ctx->stream, a);
    } else {
        auto k = linePartKernel<NUCL, LONG, MODE, LIST, EXTRAS, 512, 4, true>;
        const int rc = setDynLds(k, lds); if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, ctx->stream, a);
    }
    return PLASSHIP_OK;
}

// line lists of one partition level over `capLines` output lines: list[start[b] .. start[b + 1]) = lines of bucket b
static int buildLineLists(plasship_ctx *ctx, const uint32_t *dTags, uint64_t capLines, uint32_t nb, uint32_t *dCount, uint32_t *dStart, uint32_t *dCursor, uint32_t *dList) {
    hipStream_t st = ctx->stream;
    PH_CHECK(hipMemsetAsync(dCount, 0, (size_t) nb * 4, st));
    const unsigned g = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((capLines + TS_CHUNK - 1) / TS_CHUNK, (uint64_t) ctx->numCU * 8));
    hipLaunchKernelGGL(tagHistKernel, dim3(g), dim3(256), 0, st, dTags, capLines, nb, dCount);
    hipLaunchKernelGGL(tagScanKernel, dim3(1), dim3(1024), 0, st, (const uint32_t *) dCount, nb, dStart, dCursor);
    hipLaunchKernelGGL(tagScatterKernel, dim3(g), dim3(256), 0, st, dTags, capLines, nb, dCursor, dList);
    return PLASSHIP_OK;
}

constexpr int KM_RETRY_EARLY_OVERFLOW_CHECK = -1000;     // internal: kmermatchImpl is to be called again, waiting for the extraction's overflow count
// What the line path hands to the run reduction
struct LinesOut { void *triples = nullptr; uint64_t nTriples = 0, Nk = 0, Nm = 0; std::vector<int64_t> stalePos; uint32_t staleT = 0; float msSort1 = 0, msGroup = 0, msSort2 = 0, msPart = 0; int nPart = 1;
                  uint64_t exchangedRecordBytes = 0, exchangedTripleBytes = 0; };

constexpr uint64_t HALO_SLACK = 1u << 16;

// ---- sort #2 over the line store: range partition of `in` (the level-1 pieces `hp` with `outLine` output lines in total) by ranges of
// the bit-reversed (rep - repBase), aggregation + sort per bucket, then every representative's triples to their place in
// representative order.  TRIPLES: `in` holds weighted triples (owner side of a sharded run), else grouped records.
// out: `dOut` = nTriples triples in (rep, target, diagonal) order (+ slackTriples of room behind them); dRepStart[nReps + 1] (optional)
template <bool NUCL, bool LONG, bool TRIPLES, bool ORDOUT = false>
static int repSortLines(plasship_ctx *ctx, const void *in, const std::vector<std::pair<uint64_t, uint64_t>> &segs, uint64_t nIn, uint32_t nReps, uint32_t repBase, uint32_t nTargets,
                        uint64_t slackTriples, const std::function<void()> &inputConsumed, DevBuf &dOut, uint64_t &nTriples, DevBuf *dRepStartOut) {
    constexpr bool PL = TRIPLES ? NUCL : LONG;                 // record layout the partition kernels move: triples are 16 bytes like Rec<false>, nucleotide triples (TripleX) 24 like Rec<true>
    typedef Rec<PL> R;
    typedef typename std::conditional<ORDOUT, TripleX, Triple>::type OutT;      // ORDOUT (sharded nucleotide run, before exchange 2): triples with their rank word
    hipStream_t st = ctx->stream;
    const int numCU = ctx->numCU;
    const int maxBits = PL ? 9 : 10;
    const int idBits = std::max(1, ceilLog2((uint64_t) nTargets));
    const int repBits = std::max(1, ceilLog2((uint64_t) std::max<uint32_t>(nReps, 1)));
    const int wantBits = std::min(2 * maxBits, std::max(0, ceilLog2((nIn + 511) / 512)));    // ~512 records per sort bucket
    const int allowedLocal = 62 - idBits - DiagPack<LONG>::BITS;                             // packed sort key = [rep - bucketBase | target | diagonal | strand] must fit 63 bits
    const int sBits = std::max(std::min(wantBits, repBits), std::max(0, repBits - allowedLocal));
    if (sBits > 2 * maxBits) { setError("kmermatch: too many sequences for the packed rep-sort key"); return PLASSHIP_ERR_UNSUPPORTED; }
    const int s1 = sBits <= maxBits ? sBits : (sBits + 1) / 2, s2 = sBits - s1;
    const uint32_t nS1 = 1u << s1, nS2 = s2 ? 1u << s2 : 0u, nSort = 1u << sBits;
    // level-1 pieces: `segs` = dense runs (first line, elements) of `in` — the group kernel's arenas, or one run of received triples;
    // the table is built here (a few thousand entries at most)
    std::vector<LinePiece> hp; uint64_t outLine = 0;
    {
        uint64_t totLines = 0; for (const auto &sg : segs) totLines += (sg.second + RPL - 1) / RPL;
        const uint32_t PLr1 = pieceLinesFor(totLines, nS1, numCU, 8);
        for (const auto &sg : segs) {
            const uint64_t cnt = sg.second, lines = (cnt + RPL - 1) / RPL;
            for (uint64_t l0 = 0; l0 < lines; l0 += PLr1) {
                LinePiece pc; pc.in0 = sg.first + l0; pc.nLines = (uint32_t) std::min<uint64_t>(PLr1, lines - l0);
                pc.lastValid = (l0 + pc.nLines == lines) ? (uint32_t) (cnt - (lines - 1) * RPL) : (uint32_t) RPL;
                pc.out0 = outLine; pc.outCap = pc.nLines + nS1; pc.tagBase = 0;
                outLine += pc.outCap; hp.push_back(pc);
            }
        }
    }
    // level 2 of a RANGE partition: representatives are not evenly spread over the id range (a contig is the representative of
    // everything it overlaps), so a level-1 bucket is cut into pieces like any other input (hash buckets are even: one piece each)
    const uint32_t PLr2 = s2 ? pieceLinesFor(std::max<uint64_t>(outLine, 1), nS2, numCU, 16) : 0;
    const uint64_t maxPR2 = s2 ? std::max<uint64_t>(outLine, 1) / PLr2 + nS1 + 1 : 0;
    const uint64_t capR1 = std::max<uint64_t>(outLine, 1), capR2 = s2 ? capR1 + maxPR2 * nS2 : 0;
    const uint32_t nPR1 = (uint32_t) hp.size();
    DevBuf dR1, dRTag1, dRList1, dRPieces, dRNP, dRCnt1, dRStart1, dRCur1, dR2, dRTag2, dRList2, dRPieces2, dRNP2, dRRegBeg, dRRegEnd, dRTot2, dSortBeg, dSortCnt;
    if (dR1.alloc(capR1 * RPL * sizeof(R)) != hipSuccess || dRTag1.alloc(capR1 * 4) != hipSuccess || dRList1.alloc(capR1 * 4) != hipSuccess || dRPieces.alloc(((size_t) nPR1 + 1) * sizeof(LinePiece)) != hipSuccess ||
        dRNP.alloc(4) != hipSuccess || dRCnt1.alloc(LP_MAXB * 4) != hipSuccess || dRStart1.alloc((LP_MAXB + 1) * 4) != hipSuccess || dRCur1.alloc(LP_MAXB * 4) != hipSuccess ||
        dSortBeg.alloc((size_t) nSort * 4) != hipSuccess || dSortCnt.alloc((size_t) nSort * 4) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
    // the piece table travels from pinned memory when it fits (no wait for the copy), else from the vector (waited for below)
    void *hpPinned = nPR1 ? ctxPinnedTable(ctx, (size_t) nPR1 * sizeof(LinePiece)) : nullptr;
    if (hpPinned) memcpy(hpPinned, hp.data(), (size_t) nPR1 * sizeof(LinePiece));
    if (nPR1) PH_CHECK(hipMemcpyAsync(dRPieces.p, hpPinned ? hpPinned : (const void *) hp.data(), (size_t) nPR1 * sizeof(LinePiece), hipMemcpyHostToDevice, st));
    else PH_CHECK(hipMemsetAsync(dRTag1.p, 0xFF, capR1 * 4, st));                       // nothing to sort: no piece will write the tag array
    PH_CHECK(hipMemcpyAsync(dRNP.p, &nPR1, 4, hipMemcpyHostToDevice, st));
    // grouped records arrive with many records per representative: lines that complete inside a tile are written directly (linepart.hpp)
    static const int directLines = [] { const char *e = getenv("PLASSHIP_DIRECT_LINES"); return e ? atoi(e) : 1; }();
    LineKey rkey; rkey.rangeBits = repBits; rkey.repBase = repBase; rkey.shift = s1 ? 64 - s1 : 63; rkey.scrambleBits = repBits;      // ranges of the bit-reversed id
    int rc;
    {
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = in; a.out = dR1.p; a.tags = dRTag1.as<uint32_t>(); a.pieces = dRPieces.as<LinePiece>(); a.nPieces = dRNP.as<uint32_t>(); a.nb = nS1; a.key = rkey;
        a.direct = directLines;
        rc = launchLinePart<NUCL, PL, KEY_RANGE, false, false>(ctx, a, std::max<uint32_t>(nPR1, 1)); if (rc) return rc;
    }
    if (nPR1 && !hpPinned) PH_CHECK(plasship::streamSync(st));   // hp goes out of use (async copy of a pageable host vector)
    rc = buildLineLists(ctx, dRTag1.as<uint32_t>(), capR1, nS1, dRCnt1.as<uint32_t>(), dRStart1.as<uint32_t>(), dRCur1.as<uint32_t>(), dRList1.as<uint32_t>()); if (rc) return rc;
    inputConsumed();                                        // the caller's input buffer is dead (stream order): it may release it
    const void *sortRecs = dR1.p; const uint32_t *sortList = dRList1.as<uint32_t>(); uint64_t sortCap = capR1;
    if (s2) {
        if (dR2.alloc(capR2 * RPL * sizeof(R)) != hipSuccess || dRTag2.alloc(capR2 * 4) != hipSuccess || dRList2.alloc(capR2 * 4) != hipSuccess || dRPieces2.alloc(((size_t) maxPR2 + 1) * sizeof(LinePiece)) != hipSuccess ||
            dRNP2.alloc(4) != hipSuccess || dRRegBeg.alloc(LP_MAXB * 8) != hipSuccess || dRRegEnd.alloc(LP_MAXB * 8) != hipSuccess || dRTot2.alloc(8) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(planListKernel, dim3(1), dim3(1024), 0, st, (const uint32_t *) dRStart1.as<uint32_t>(), nS1, PLr2, nS2, dRPieces2.as<LinePiece>(), dRNP2.as<uint32_t>(),
                           dRRegBeg.as<uint64_t>(), dRRegEnd.as<uint64_t>(), dRTot2.as<uint64_t>());
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = dR1.p; a.list = dRList1.as<uint32_t>(); a.out = dR2.p; a.tags = dRTag2.as<uint32_t>(); a.pieces = dRPieces2.as<LinePiece>(); a.nPieces = dRNP2.as<uint32_t>(); a.nb = nS2;
        a.direct = directLines;
        a.key = rkey; a.key.shift = 64 - s1 - s2;
        rc = launchLinePart<NUCL, PL, KEY_RANGE, true, false>(ctx, a, maxPR2); if (rc) return rc;
        hipLaunchKernelGGL(tagSortRegionKernel, dim3(std::min<uint32_t>(nS1, (uint32_t) numCU * 4)), dim3(512), 0, st, (const uint32_t *) dRTag2.as<uint32_t>(), (const uint64_t *) dRRegBeg.as<uint64_t>(),
                           (const uint64_t *) dRRegEnd.as<uint64_t>(), nS1, nS2, dRList2.as<uint32_t>(), dSortBeg.as<uint32_t>(), dSortCnt.as<uint32_t>());
        sortRecs = dR2.p; sortList = dRList2.as<uint32_t>(); sortCap = capR2;
    } else {
        hipLaunchKernelGGL(listRangesKernel, dim3(4), dim3(256), 0, st, (const uint32_t *) dRStart1.as<uint32_t>(), nS1, dSortBeg.as<uint32_t>(), dSortCnt.as<uint32_t>());
    }
    // aggregate + sort each bucket; buckets beyond the LDS capacity use HBM scratch
    DevBuf dBigNeed, dBigOff, dBigScratch, dUnique, dScanTmp3, dSparse;
    const size_t scanTmp3Bytes = exclusiveScanTmpBytes((size_t) nSort + 2);
    if (dBigNeed.alloc(((size_t) nSort + 1) * 8) != hipSuccess || dBigOff.alloc(((size_t) nSort + 2) * 8) != hipSuccess || dUnique.alloc(((size_t) nSort + 1) * 4) != hipSuccess ||
        dScanTmp3.alloc(scanTmp3Bytes) != hipSuccess || dSparse.alloc(sortCap * RPL * sizeof(OutT)) != hipSuccess) {
        setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE;
    }
    // pass 1: every bucket aggregated in LDS (no scratch); pass 2: the few buckets with more distinct triples than LDS holds
    const AggLines aggLn{sortList, dSortBeg.as<uint32_t>(), dSortCnt.as<uint32_t>()};
    const unsigned aggGrid = std::min<uint32_t>(nSort, (uint32_t) numCU * (uint32_t) tuneInt("AGGSORT", 16));
    hipLaunchKernelGGL((aggSortKernel<NUCL, LONG, true, TRIPLES, ORDOUT>), dim3(aggGrid), dim3(LS_BLOCK), 0, st, sortRecs, dSparse.p, (const uint64_t *) nullptr, nSort,
                       (unsigned long long *) nullptr, (const uint64_t *) nullptr, dUnique.as<uint32_t>(), repBits - sBits, idBits, (uint64_t) repBase, aggLn, repBits);
    hipLaunchKernelGGL(bigNeedKernel, dim3(gridFor(nSort, 256, 1024)), dim3(256), 0, st, (const uint32_t *) dSortCnt.as<uint32_t>(), (const uint32_t *) dUnique.as<uint32_t>(), nSort, dBigNeed.as<uint64_t>(), NUCL ? 3u : 2u);
    if (exclusiveScanU64(st, dBigNeed.as<uint64_t>(), dBigOff.as<uint64_t>(), nSort, dScanTmp3.p, scanTmp3Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t bigTot = 0;
    PH_COPY_SYNC(st, &bigTot, dBigOff.as<uint64_t>() + nSort, 8, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    if (bigTot) {
        if (dBigScratch.alloc(bigTot * 8) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL((aggSortKernel<NUCL, LONG, true, TRIPLES, ORDOUT>), dim3(aggGrid), dim3(LS_BLOCK), 0, st, sortRecs, dSparse.p, (const uint64_t *) nullptr, nSort,
                           dBigScratch.as<unsigned long long>(), (const uint64_t *) dBigOff.as<uint64_t>(), dUnique.as<uint32_t>(), repBits - sBits, idBits, (uint64_t) repBase, aggLn, repBits);
    }
    // every bucket now holds its representatives' triples, each representative's contiguous and in (target, diagonal) order, but
    // the buckets are ranges of the bit-reversed id: count the triples per representative, prefix-sum over the ids, and move every
    // triple to its place in id order — the (rep, target, diagonal)-sorted array the run reduction walks
    DevBuf dRepCnt, dRepStartLocal, dScanTmp4;
    DevBuf &dRepStart = dRepStartOut ? *dRepStartOut : dRepStartLocal;
    const size_t scanTmp4Bytes = exclusiveScanTmpBytes((size_t) nReps + 2);
    if (dRepCnt.alloc(((size_t) nReps + 1) * 4) != hipSuccess || dRepStart.alloc(((size_t) nReps + 2) * 8) != hipSuccess ||
        dScanTmp4.alloc(scanTmp4Bytes) != hipSuccess) { setError("kmermatch: out of device memory for the rep sort"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dRepCnt.p, 0, ((size_t) nReps + 1) * 4, st));
    const unsigned runGrid = std::min<uint32_t>((nSort + 3) / 4, (uint32_t) numCU * 8);
    hipLaunchKernelGGL(repRunsKernel<OutT>, dim3(runGrid), dim3(256), 0, st, (const OutT *) dSparse.p, (const uint32_t *) dSortBeg.as<uint32_t>(),
                       (const uint32_t *) dUnique.as<uint32_t>(), nSort, repBase, dRepCnt.as<uint32_t>());
    if (exclusiveScanU32(st, dRepCnt.as<uint32_t>(), dRepStart.as<uint64_t>(), nReps, dScanTmp4.p, scanTmp4Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    nTriples = 0;
    PH_COPY_SYNC(st, &nTriples, dRepStart.as<uint64_t>() + nReps, 8, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    dR1.release(); dR2.release();
    if (dOut.alloc((std::max<uint64_t>(nTriples, 1) + slackTriples) * sizeof(OutT)) != hipSuccess) { setError("kmermatch: out of device memory for the sorted triples"); return PLASSHIP_ERR_DEVICE; }
    if (nTriples) hipLaunchKernelGGL(placeRunsKernel<OutT>, dim3(runGrid), dim3(256), 0, st, (const OutT *) dSparse.p, (const uint32_t *) dSortBeg.as<uint32_t>(), (const uint32_t *) dUnique.as<uint32_t>(), nSort,
                                     repBase, (const uint64_t *) dRepStart.as<uint64_t>(), (OutT *) dOut.p);
    PH_CHECK(hipGetLastError());
    return PLASSHIP_OK;
}

static void moveBuf(DevBuf &dst, DevBuf &src) { dst.release(); dst.p = src.p; dst.bytes = src.bytes; src.p = nullptr; src.bytes = 0; }

// extraction has filled dA (`total` record slots of this rank's sequences, sentinels in unused slots).  Buffers dA / dB hold geo.cap2
// resp. geo.cap1 lines (single GPU) or geo.cap1 lines each (sharded run).
// Sharded run (commOf(ctx) != nullptr; `totalAll` = slots of all ranks): the bucket geometry is that of the WHOLE run, rank r owns the
// level-1 buckets [ceil(r nb1 / W), ceil((r+1) nb1 / W)).  Exchange 1 ships the level-1 LINES of the other ranks' buckets (gathered by
// destination through the line list: one extra pass over this rank's records), the receiver builds a line list over what arrived and
// runs level 2 and the group kernel exactly as a single GPU does.  Exchange 2 ships AGGREGATED (rep, target, diagonal, count) triples —
// every rank first runs the whole rep sort on the grouped records of its own buckets — and the owner of a representative merges the
// triples of all ranks with the same kernels (aggSortKernel<TRIPLES>).
template <bool NUCL, bool LONG>
static int kmermatchLines(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par, const LineGeo &geo, uint64_t total,
                          DevBuf &dA, DevBuf &dB, const DevBuf &dSlotOff, const DevBuf &dKStats, const ExtractArgs &ea, int keyBits, LinesOut &res,
                          const uint32_t *dLateOverflow) {
    typedef Rec<LONG> R;
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const int numCU = ctx->numCU;
    const plasship_comm *cm = commOf(ctx);
    const int W = cm ? cm->world : 1, rk = cm ? cm->rank : 0;
    // stage boundaries are events read when the call is over (kmermatchImpl): nobody waits for the stream just to time a stage
    const int valueShift = std::max(0, keyBits - 11);         // VH_BINS = 2^11 monotone bins
    // ---- hash partition (replaces sort #1): level 1 over the slot array, level 2 over every level-1 bucket's line list ----
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    const uint32_t bLo = cm ? (uint32_t) ownedBegin(geo.nb1, rk, W) : 0u, bHi = cm ? (uint32_t) ownedBegin(geo.nb1, rk + 1, W) : geo.nb1;
    const uint32_t nbL = bHi - bLo;                           // level-1 buckets this rank groups (all of them on a single GPU)
    DevBuf dVHist, dMinKey, dTag1, dList1, dCnt1, dStart1, dCur1, dTag2, dList2, dPieces2, dNP2, dRegBeg, dRegEnd, dTot2, dFineBeg, dFineCnt;
    DevBuf dRx, dRxList, dRxSegs, dRxStart;                   // sharded run: received lines, their list, per-bucket list offsets
    const uint32_t nBuckets = geo.nb2 ? nbL * geo.nb2 : nbL;
    if (dVHist.alloc(VH_BINS * 4) != hipSuccess || dMinKey.alloc(8) != hipSuccess || dTag1.alloc(geo.cap1 * 4) != hipSuccess || dList1.alloc(geo.cap1 * 4) != hipSuccess ||
        dCnt1.alloc(LP_MAXB * 4) != hipSuccess || dStart1.alloc((LP_MAXB + 1) * 4) != hipSuccess || dCur1.alloc(LP_MAXB * 4) != hipSuccess ||
        dFineBeg.alloc((size_t) std::max<uint32_t>(nBuckets, 1) * 4) != hipSuccess || dFineCnt.alloc((size_t) std::max<uint32_t>(nBuckets, 1) * 4) != hipSuccess ||
        (geo.nb2 && (dPieces2.alloc((geo.maxP2 + 1) * sizeof(LinePiece)) != hipSuccess ||
                     dNP2.alloc(4) != hipSuccess || dRegBeg.alloc(LP_MAXB * 8) != hipSuccess || dRegEnd.alloc(LP_MAXB * 8) != hipSuccess || dTot2.alloc(8) != hipSuccess))) {
        setError("kmermatch: out of device memory for the line lists"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipMemsetAsync(dVHist.p, 0, VH_BINS * 4, st));
    PH_CHECK(hipMemsetAsync(dMinKey.p, 0xFF, 8, st));
    if (geo.nP1 == 0) PH_CHECK(hipMemsetAsync(dTag1.p, 0xFF, geo.cap1 * 4, st));       // no piece will write the (one-line) tag array
    {
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = dA.p; a.out = dB.p; a.tags = dTag1.as<uint32_t>(); a.totalLines = geo.totalLines; a.lastValidAll = geo.lastValid; a.pieceLines = geo.PL1; a.nb = geo.nb1;
        a.key.shift = geo.b1 ? 64 - geo.b1 : 63; a.key.rangeBits = 0; a.key.repBase = 0;
        a.minKey = NUCL ? dMinKey.as<unsigned long long>() : nullptr; a.valueHist = dVHist.as<uint32_t>(); a.valueShift = valueShift;
        PH_CHECK(hipEventRecord(ctx->ev[8], st));
        const int rc = launchLinePart<NUCL, LONG, KEY_HASH, false, true>(ctx, a, geo.nP1); if (rc) return rc;
        PH_CHECK(hipEventRecord(ctx->ev[9], st));
    }
    int rc = buildLineLists(ctx, dTag1.as<uint32_t>(), geo.cap1, geo.nb1, dCnt1.as<uint32_t>(), dStart1.as<uint32_t>(), dCur1.as<uint32_t>(), dList1.as<uint32_t>()); if (rc) return rc;
    // what level 2 (or, with a single level, the group kernel) reads: records, their line list, the list offsets of the nbL buckets
    void *l1Recs = dB.p; const uint32_t *l1List = dList1.as<uint32_t>(); const uint32_t *l1Start = dStart1.as<uint32_t>() + bLo; uint64_t l1Lines = geo.cap1;
    const uint32_t *l1Tags = dTag1.as<uint32_t>();
    uint64_t NkAll = 0;
    if (cm) {
        // ---- exchange 1: the lines of every level-1 bucket to the bucket's owner ----
        std::vector<uint32_t> hStart1(geo.nb1 + 1);
        PH_COPY_SYNC(st, hStart1.data(), dStart1.p, ((size_t) geo.nb1 + 1) * 4, hipMemcpyDeviceToHost);
        PH_CHECK(hipGetLastError());
        const uint64_t myLines = hStart1[geo.nb1];
        // lines in list order = by bucket = by destination.  The lines of the OTHER ranks' buckets are packed into a send buffer of
        // exactly their size (the slot array is consumed: it goes first, the receive buffer will need the room); the lines of this
        // rank's own buckets never pass through it: they are gathered straight into the receive buffer, behind what the others sent
        // (one pass over them instead of a gather and a device-to-device copy — all of the data in a 1-rank group, 1/W of it otherwise).
        const uint64_t selfBeg = hStart1[ownedBegin(geo.nb1, rk, W)], selfEnd = hStart1[ownedBegin(geo.nb1, rk + 1, W)];
        const uint64_t selfLines = selfEnd - selfBeg, sendLines = myLines - selfLines;
        const uint32_t chunksPerLine = (uint32_t) (RPL * sizeof(R) / 16);
        auto gather = [&](uint64_t listFrom, uint64_t n, void *dst) {
            if (n) hipLaunchKernelGGL(gatherLinesKernel, dim3(gridFor(n * chunksPerLine, 256, (unsigned) numCU * 16)), dim3(256), 0, st, (const uint4 *) dB.p, (const uint32_t *) dList1.as<uint32_t>() + listFrom,
                                      n, chunksPerLine, (uint4 *) dst);
        };
        dA.release();
        if (dA.alloc(std::max<uint64_t>(sendLines, 1) * RPL * sizeof(R)) != hipSuccess) { setError("kmermatch: out of device memory for the send buffer"); return PLASSHIP_ERR_DEVICE; }
        gather(0, selfBeg, dA.p);
        gather(selfEnd, myLines - selfEnd, (char *) dA.p + selfBeg * RPL * sizeof(R));
        dTag1.release();
        // per-bucket line counts of every rank + the records this rank extracted
        std::vector<uint64_t> mine((size_t) geo.nb1 + 1), all(((size_t) geo.nb1 + 1) * (size_t) W);
        for (uint32_t j = 0; j < geo.nb1; j++) mine[j] = hStart1[j + 1] - hStart1[j];
        { unsigned long long ks[4] = {0, 0, 0, 0}; PH_COPY_SYNC(st, ks, dKStats.p, 32, hipMemcpyDeviceToHost); mine[geo.nb1] = ks[1] + ks[3]; }
        rc = commAllgatherHost(ctx, mine.data(), all.data(), mine.size() * 8); if (rc) return rc;
        std::vector<uint64_t> sendCount(W);
        for (int r = 0; r < W; r++) { sendCount[r] = (r == rk) ? 0 : hStart1[ownedBegin(geo.nb1, r + 1, W)] - hStart1[ownedBegin(geo.nb1, r, W)]; NkAll += all[(size_t) r * mine.size() + geo.nb1]; }
        uint64_t gotOthers = 0;
        // (room behind the received lines: this rank's own lines, and the group kernel's arenas, which are addressed by level-2 line
        // numbers, up to nbL * nb2 beyond)
        rc = commAlltoallvRecords(ctx, dA.p, sendCount.data(), RPL * sizeof(R), dRx, &gotOthers, selfLines + (uint64_t) nbL * geo.nb2 + 1); if (rc) return rc;
        res.exchangedRecordBytes = sendLines * RPL * sizeof(R);
        PH_TRACE(st, "kmermatch: exchange 1 (level-1 lines)");
        dA.release();
        gather(selfBeg, selfLines, (char *) dRx.p + gotOthers * RPL * sizeof(R));
        dB.release(); dList1.release();    // (stream order: the gathers have read them before anything reuses the memory)
        const uint64_t gotLines = gotOthers + selfLines;
        if (gotLines >= 0xFFFFFFFFull) { rc = PLASSHIP_ERR_UNSUPPORTED; setError("kmermatch: more than 2^32 lines on one rank"); }
        rc = commAgreeOk(ctx, rc == 0, "kmermatch: more than 2^32 lines on one rank"); if (rc) return rc;
        // what arrived: from every source s the lines of my buckets bLo .. bHi-1, bucket after bucket.  List: bucket-major, source-minor.
        std::vector<RxSeg> segs((size_t) nbL * W); std::vector<uint32_t> rxStart((size_t) nbL + 1);
        // (in the receive buffer: the other ranks' lines in rank order, then this rank's own)
        std::vector<uint64_t> srcBase(W); { uint64_t o = 0; auto linesFrom = [&](int s) { uint64_t c = 0; for (uint32_t j = bLo; j < bHi; j++) c += all[(size_t) s * mine.size() + j]; return c; };
          for (int s = 0; s < W; s++) if (s != rk) { srcBase[s] = o; o += linesFrom(s); }
          if (o != gotOthers || linesFrom(rk) != selfLines) { setError("kmermatch: internal error, exchanged line counts do not add up"); return PLASSHIP_ERR_DEVICE; }
          srcBase[rk] = o; }
        { uint64_t d = 0; std::vector<uint64_t> run(srcBase);
          for (uint32_t j = 0; j < nbL; j++) { rxStart[j] = (uint32_t) d; for (int s = 0; s < W; s++) { const uint64_t c = all[(size_t) s * mine.size() + bLo + j]; segs[(size_t) j * W + s] = RxSeg{run[s], (uint32_t) c, (uint32_t) d}; run[s] += c; d += c; } }
          rxStart[nbL] = (uint32_t) d; }
        if (dRxList.alloc(std::max<uint64_t>(gotLines, 1) * 4) != hipSuccess || dRxSegs.alloc(std::max<size_t>(segs.size(), 1) * sizeof(RxSeg)) != hipSuccess || dRxStart.alloc(((size_t) nbL + 1) * 4) != hipSuccess) { setError("kmermatch: out of device memory for the received line list"); return PLASSHIP_ERR_DEVICE; }
        PH_COPY_SYNC(st, dRxSegs.p, segs.data(), segs.size() * sizeof(RxSeg), hipMemcpyHostToDevice);
        PH_COPY_SYNC(st, dRxStart.p, rxStart.data(), ((size_t) nbL + 1) * 4, hipMemcpyHostToDevice);
        if (!segs.empty()) hipLaunchKernelGGL(rxListKernel, dim3((unsigned) std::min<size_t>(segs.size(), (size_t) numCU * 8)), dim3(256), 0, st, (const RxSeg *) dRxSegs.p, (uint32_t) segs.size(), dRxList.as<uint32_t>());
        if (NUCL) {              // the globally smallest key (first-run quirk of the group kernel)
            uint64_t mk = 0; PH_COPY_SYNC(st, &mk, dMinKey.p, 8, hipMemcpyDeviceToHost);
            rc = commAllReduceMinU64(ctx, &mk, 1); if (rc) return rc;
            PH_COPY_SYNC(st, dMinKey.p, &mk, 8, hipMemcpyHostToDevice);
        }
        l1Recs = dRx.p; l1List = dRxList.as<uint32_t>(); l1Start = dRxStart.as<uint32_t>(); l1Lines = gotLines; l1Tags = nullptr;
    }
    void *finalRecs = l1Recs; const uint32_t *finalTags = l1Tags, *finalList = l1List; uint64_t finalCap = l1Lines;
    DevBuf dL2;                                               // sharded run: level-2 output (a single GPU writes level 2 into dA)
    res.nPart = 1;
    if (geo.nb2) {
        const uint64_t cap2 = cm ? l1Lines + (uint64_t) nbL * geo.nb2 : geo.cap2;
        void *l2Out = dA.p;
        if (cm) { if (dL2.alloc(std::max<uint64_t>(cap2, 1) * RPL * sizeof(R)) != hipSuccess) { setError("kmermatch: out of device memory for the k-mer record arrays"); return PLASSHIP_ERR_DEVICE; } l2Out = dL2.p; }
        if (dTag2.alloc(std::max<uint64_t>(cap2, 1) * 4) != hipSuccess || dList2.alloc(std::max<uint64_t>(cap2, 1) * 4) != hipSuccess) { setError("kmermatch: out of device memory for the line lists"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(planListKernel, dim3(1), dim3(1024), 0, st, l1Start, nbL, geo.PL2, geo.nb2, dPieces2.as<LinePiece>(), dNP2.as<uint32_t>(),
                           dRegBeg.as<uint64_t>(), dRegEnd.as<uint64_t>(), dTot2.as<uint64_t>());
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = l1Recs; a.list = l1List; a.out = l2Out; a.tags = dTag2.as<uint32_t>(); a.pieces = dPieces2.as<LinePiece>(); a.nPieces = dNP2.as<uint32_t>(); a.nb = geo.nb2;
        a.key.shift = 64 - geo.b1 - geo.b2;
        PH_CHECK(hipEventRecord(ctx->ev[10], st));
        rc = launchLinePart<NUCL, LONG, KEY_HASH, true, false>(ctx, a, geo.maxP2); if (rc) return rc;
        PH_CHECK(hipEventRecord(ctx->ev[11], st));
        hipLaunchKernelGGL(tagSortRegionKernel, dim3(std::min<uint32_t>(nbL, (uint32_t) numCU * 4)), dim3(512), 0, st, (const uint32_t *) dTag2.as<uint32_t>(), (const uint64_t *) dRegBeg.as<uint64_t>(),
                           (const uint64_t *) dRegEnd.as<uint64_t>(), nbL, geo.nb2, dList2.as<uint32_t>(), dFineBeg.as<uint32_t>(), dFineCnt.as<uint32_t>());
        finalRecs = l2Out; finalTags = dTag2.as<uint32_t>(); finalList = dList2.as<uint32_t>(); finalCap = cap2;
        res.nPart = 2;
    } else {
        hipLaunchKernelGGL(listRangesKernel, dim3(4), dim3(256), 0, st, l1Start, nbL, dFineBeg.as<uint32_t>(), dFineCnt.as<uint32_t>());
    }
    PH_CHECK(hipEventRecord(ctx->ev[6], st));
    PH_TRACE(st, "kmermatch: hash partition (line store)");
    PH_CHECK(hipGetLastError());

    // ---- assignGroup: every workgroup writes its grouped records into an arena that begins where its first bucket's lines begin ----

    // the arenas: the buffer level 2 read (dead now).  single GPU: dB (level 1's output) when there are two levels, else dA (the slot
    // array); sharded run: the receive buffer when there are two levels, else a buffer of its own
    DevBuf dArena;
    void *arenaBuf;
    if (cm) {
        if (geo.nb2) arenaBuf = dRx.p;
        else { if (dArena.alloc(std::max<uint64_t>(finalCap, 1) * RPL * sizeof(R)) != hipSuccess) { setError("kmermatch: out of device memory for the grouped records"); return PLASSHIP_ERR_DEVICE; } arenaBuf = dArena.p; }
    } else arenaBuf = geo.nb2 ? dB.p : dA.p;
    const uint32_t gBlocks = std::max<uint32_t>(1, std::min<uint32_t>(nBuckets, (uint32_t) numCU * (uint32_t) tuneInt("GROUP", 6)));
    const uint32_t bpb = (std::max<uint32_t>(nBuckets, 1) + gBlocks - 1) / gBlocks;
    const uint32_t gGrid = (std::max<uint32_t>(nBuckets, 1) + bpb - 1) / bpb;
    DevBuf dOutCnt, dArenaStart, dMaxRT, dLastRun;
    if (dOutCnt.alloc((size_t) gGrid * 8) != hipSuccess || dArenaStart.alloc((size_t) gGrid * 8) != hipSuccess || dMaxRT.alloc(8) != hipSuccess || dLastRun.alloc(32) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    GroupArgs ga; memset(&ga, 0, sizeof(ga));
    ga.in = finalRecs; ga.out = arenaBuf; ga.list = finalList; ga.lineBeg = dFineBeg.as<uint32_t>(); ga.lineCnt = dFineCnt.as<uint32_t>();
    ga.nBuckets = nBuckets; ga.bucketsPerBlock = bpb; ga.outCount = dOutCnt.as<uint64_t>();
    PH_CHECK(hipMemsetAsync(dMaxRT.p, 0, 8, st));
    ga.maxRepTarget = dMaxRT.as<unsigned long long>();
    ga.includeOnlyExtendable = par->include_only_extendable; ga.covMode = par->cov_mode; ga.covThr = par->cov_thr; ga.minKey = NUCL ? dMinKey.as<unsigned long long>() : nullptr;
    // positions per bucket (sentinel padding included) decide the workgroup shape of the 16-byte-record kernel
    const uint64_t avgPos = finalCap * RPL / std::max<uint32_t>(nBuckets, 1);
    const bool wideGroup = !LONG && (getenv("PLASSHIP_GROUP_WIDE") ? atoi(getenv("PLASSHIP_GROUP_WIDE")) != 0 : avgPos > 1600);
    if (nBuckets == 0) PH_CHECK(hipMemsetAsync(dOutCnt.p, 0, (size_t) gGrid * 8, st));
    else if constexpr (LONG) hipLaunchKernelGGL((groupKernel<NUCL, LONG, true>), dim3(gGrid), dim3(GR_BLOCK), 0, st, ga);
    else if (wideGroup && tuneInt("GROUP_WPE", 4) == 4) hipLaunchKernelGGL((groupLinesKernel<NUCL, 512, 4096, 4>), dim3(gGrid), dim3(512), 0, st, ga);
    else if (wideGroup) hipLaunchKernelGGL((groupLinesKernel<NUCL, 512, 4096, 2>), dim3(gGrid), dim3(512), 0, st, ga);
    else hipLaunchKernelGGL((groupLinesKernel<NUCL, GR_BLOCK, GR_HT, 3>), dim3(gGrid), dim3(GR_BLOCK), 0, st, ga);
    if (nBuckets) hipLaunchKernelGGL(arenaStartKernel, dim3(gridFor(gGrid, 256, 64)), dim3(256), 0, st, (const uint32_t *) dFineBeg.as<uint32_t>(), bpb, gGrid, nBuckets, dArenaStart.as<uint64_t>());
    else PH_CHECK(hipMemsetAsync(dArenaStart.p, 0, (size_t) gGrid * 8, st));
    std::vector<uint64_t> hOutCnt(gGrid), hArena(gGrid);
    unsigned long long hLastRun[4] = {0, 0, 0, 0}, ks[4] = {0, 0, 0, 0}; std::vector<uint32_t> hVHist(VH_BINS);
    hipLaunchKernelGGL(lastRunInfoKernel, dim3(1), dim3(1), 0, st, dMaxRT.as<unsigned long long>(), dSlotOff.as<uint64_t>(), db->d_len.as<uint32_t>(), N, dLastRun.as<unsigned long long>());
    PH_CHECK(hipMemcpyAsync(hLastRun, dLastRun.p, 32, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(hVHist.data(), dVHist.p, VH_BINS * 4, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(hOutCnt.data(), dOutCnt.p, (size_t) gGrid * 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(hArena.data(), dArenaStart.p, (size_t) gGrid * 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(hipMemcpyAsync(ks, dKStats.p, 32, hipMemcpyDeviceToHost, st));
    uint32_t lateOverflow = 0;
    if (dLateOverflow) PH_CHECK(hipMemcpyAsync(&lateOverflow, dLateOverflow, 4, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    if (lateOverflow) return KM_RETRY_EARLY_OVERFLOW_CHECK;          // (never in a sharded run: its ranks wait for the count right after the extraction)
    uint64_t NmLocal = 0;
    for (uint32_t j = 0; j < gGrid; j++) NmLocal += hOutCnt[j];
    uint64_t Nm = NmLocal;
    const uint64_t NkLocal = ks[1] + ks[3];                  // records the extraction kernels of this rank wrote (sentinels excluded)
    const uint64_t Nk = cm ? NkAll : NkLocal;                // ... and of the whole run
    std::vector<uint64_t> hVHistG(hVHist.begin(), hVHist.end());
    if (cm) {
        // the stale-record check below is a property of the WHOLE run: N_m, the last (rep, target) run and the value histogram are
        // reduced over the ranks; every rank then takes the same decisions (and the same collectives)
        std::vector<uint64_t> mine(2 + (size_t) VH_BINS), all((2 + (size_t) VH_BINS) * (size_t) W);
        mine[0] = NmLocal; mine[1] = hLastRun[0]; std::copy(hVHistG.begin(), hVHistG.end(), mine.begin() + 2);
        rc = commAllgatherHost(ctx, mine.data(), all.data(), mine.size() * 8); if (rc) return rc;
        uint64_t mx = 0; Nm = 0; std::fill(hVHistG.begin(), hVHistG.end(), 0);
        for (int r = 0; r < W; r++) {
            const uint64_t *row = all.data() + (size_t) r * mine.size();
            Nm += row[0]; mx = std::max(mx, row[1]);
            for (uint32_t b = 0; b < VH_BINS; b++) hVHistG[b] += row[2 + b];
        }
        PH_CHECK(hipMemcpyAsync(dMaxRT.p, &mx, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lastRunInfoKernel, dim3(1), dim3(1), 0, st, dMaxRT.as<unsigned long long>(), dSlotOff.as<uint64_t>(), db->d_len.as<uint32_t>(), N, dLastRun.as<unsigned long long>());
        PH_COPY_SYNC(st, hLastRun, dLastRun.p, 32, hipMemcpyDeviceToHost);
    }
    res.Nk = NkLocal;                                         // sharded run: the records THIS rank extracted (the ranks' sum is the run's N_k)
    res.Nm = NmLocal;
    PH_CHECK(hipEventRecord(ctx->ev[7], st));
    PH_TRACE(st, "kmermatch: group (line store)");

    // ---- stale records behind the compaction point that continue the last run (see section 7 above) ----
    if (Nm > 0 && Nm < Nk) {
        const unsigned long long maxRT = hLastRun[0];
        res.staleT = (uint32_t) (maxRT & 0xFFFFFFFFull);
        const uint64_t so[2] = {hLastRun[1], hLastRun[2]}; const uint32_t tLen = (uint32_t) hLastRun[3];
        const uint32_t tb = (uint32_t) (so[1] - so[0]);
        DevBuf dTRec, dTId, dTScr, dTOff, dTCap, dDiff;
        uint32_t cap = 64; while (cap < tLen + 1) cap <<= 1;
        const uint64_t zero = 0;
        if (dTRec.alloc((size_t) tb * sizeof(R)) != hipSuccess || dTId.alloc(4) != hipSuccess || dTScr.alloc((size_t) cap * sizeof(Cand)) != hipSuccess ||
            dTOff.alloc(8) != hipSuccess || dTCap.alloc(4) != hipSuccess || dDiff.alloc(((size_t) tb + 1) * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemcpyAsync(dTId.p, &res.staleT, 4, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemcpyAsync(dTOff.p, &zero, 8, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemcpyAsync(dTCap.p, &cap, 4, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemsetAsync(dDiff.p, 0, ((size_t) tb + 1) * 8, st));
        // re-extract the records of T into a scratch array with the very kernel that produced them
        ExtractArgs ta = ea; ta.waveList = nullptr; ta.waveCount = nullptr; ta.kstats = nullptr; ta.arr = dTRec.p; ta.slotBias = so[0];
        ta.idList = dTId.as<uint32_t>(); ta.nIds = 1; ta.scratch = dTScr.as<Cand>(); ta.scratchOff = dTOff.as<uint64_t>(); ta.scratchCap = dTCap.as<uint32_t>();
        hipLaunchKernelGGL((extractKernel<NUCL, LONG, 1, true>), dim3(1), dim3(64), 0, st, ta);
        std::vector<R> trec(tb);
        PH_CHECK(hipMemcpyAsync(trec.data(), dTRec.p, (size_t) tb * sizeof(R), hipMemcpyDeviceToHost, st));
        PH_CHECK(plasship::streamSync(st));
        trec.erase(std::remove_if(trec.begin(), trec.end(), [](const R &r) { return r.kmer == ~0ULL && r.id == 0xFFFFFFFFu; }), trec.end());
        std::sort(trec.begin(), trec.end(), [](const R &x, const R &y) { return recLess1<NUCL, LONG>(x, y); });
        const uint32_t m = (uint32_t) trec.size();
        // cheap exact filter first: the value histogram bounds the sort-#1 rank of every record of T; the scan can only reach
        // a record of T if rank N_m itself can be one of them
        bool mayHit = false;
        if (m) {
            std::vector<uint64_t> cum(VH_BINS + 1, 0);
            for (uint32_t b = 0; b < VH_BINS; b++) cum[b + 1] = cum[b] + hVHistG[b];
            for (uint32_t j = 0; j < m && !mayHit; j++) { const uint32_t b = valueBin<NUCL>(trec[j].kmer, valueShift); mayHit = Nm >= cum[b] && Nm < cum[b + 1]; }
        }
        if (m && mayHit) {
            PH_CHECK(hipMemcpyAsync(dTRec.p, trec.data(), (size_t) m * sizeof(R), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((rankLinesKernel<NUCL, LONG>), dim3(gridFor(finalCap * RPL, 256, (unsigned) numCU * 8)), dim3(256), 0, st, (const void *) finalRecs, finalTags, finalCap, geo.nb2 ? (const uint64_t *) dTot2.as<uint64_t>() : (const uint64_t *) nullptr,
                               (const void *) dTRec.p, m, dDiff.as<unsigned long long>());
            std::vector<unsigned long long> diff((size_t) m + 1);
            PH_CHECK(hipMemcpyAsync(diff.data(), dDiff.p, ((size_t) m + 1) * 8, hipMemcpyDeviceToHost, st));
            PH_CHECK(plasship::streamSync(st));
            if (cm) {            // every rank counted the records of its own buckets: the sort-#1 ranks are the sums
                static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
                rc = commAllReduceSumU64(ctx, reinterpret_cast<uint64_t *>(diff.data()), diff.size()); if (rc) return rc;
            }
            unsigned long long rank = 0, expect = Nm;
            for (uint32_t j = 0; j < m; j++) {
                rank += diff[j];                         // records strictly before trec[j] in sort-#1 order
                if (rank == expect) { res.stalePos.push_back((int64_t) trec[j].pos); expect++; }
                else if (rank > expect) break;
            }
        }
        PH_TRACE(st, "kmermatch: stale-record check (line store)");
    }

    // ---- sort #2: range partition of the grouped records by rep id over the line store + aggregation / sort per bucket ----
    PH_CHECK(hipEventRecord(ctx->ev[12], st));
    // the hash-bucketed records are dead now (the group kernel's arenas live in another buffer): free them for the rep side
    if (cm) { dL2.release(); if (!geo.nb2) dRx.release(); }
    else (finalRecs == dA.p ? dA : dB).release();
    dTag1.release(); dList1.release(); dTag2.release(); dList2.release(); dRxList.release();
    std::vector<std::pair<uint64_t, uint64_t>> arenas(gGrid);                 // the arenas are dense segments: (first line, records)
    for (uint32_t j = 0; j < gGrid; j++) arenas[j] = std::make_pair(hArena[j] / RPL, hOutCnt[j]);
    DevBuf dTriples, dRepStart; uint64_t nTriples = 0;
    auto arenasConsumed = [&]() { if (cm) { if (geo.nb2) dRx.release(); else dArena.release(); } else (arenaBuf == dA.p ? dA : dB).release(); };
    // (sharded nucleotide run: the triples leave for their owners with the rank word of their top member, see TripleX)
    if (NUCL && cm) rc = repSortLines<NUCL, LONG, false, NUCL>(ctx, arenaBuf, arenas, NmLocal, N, 0u, N, 0, arenasConsumed, dTriples, nTriples, &dRepStart);
    else rc = repSortLines<NUCL, LONG, false>(ctx, arenaBuf, arenas, NmLocal, N, 0u, N, 0, arenasConsumed, dTriples, nTriples, cm ? &dRepStart : nullptr);
    if (rc) return rc;
    if (cm) {
        // ---- exchange 2: every representative's aggregated triples to the representative's owner (contiguous id ranges: the triples
        //      are in id order, so a rank's share is one run), merged there with the triples of the other ranks ----
        DevBuf dOB; std::vector<uint64_t> ob((size_t) W + 1), sendCount(W);
        if (dOB.alloc(((size_t) W + 1) * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(ownerBoundsKernel, dim3(1), dim3(256), 0, st, (const uint64_t *) dRepStart.as<uint64_t>(), (uint64_t) N, W, dOB.as<uint64_t>());
        PH_COPY_SYNC(st, ob.data(), dOB.p, ((size_t) W + 1) * 8, hipMemcpyDeviceToHost);
        for (int r = 0; r < W; r++) sendCount[r] = ob[r + 1] - ob[r];
        DevBuf dRxT; uint64_t gotT = 0;
        constexpr size_t xBytes = NUCL ? sizeof(TripleX) : sizeof(Triple);
        rc = commAlltoallvRecords(ctx, dTriples.p, sendCount.data(), xBytes, dRxT, &gotT, RPL); if (rc) return rc;
        res.exchangedTripleBytes = (nTriples - sendCount[rk]) * xBytes;
        PH_TRACE(st, "kmermatch: exchange 2 (aggregated triples)");
        dTriples.release(); dRepStart.release();
        const uint32_t repBase = (uint32_t) ownedBegin(N, rk, W), ownedN = (uint32_t) (ownedBegin(N, rk + 1, W) - repBase);
        DevBuf dMerged; uint64_t nMerged = 0;
        rc = repSortLines<NUCL, LONG, true>(ctx, dRxT.p, {std::make_pair((uint64_t) 0, gotT)}, gotT, ownedN, repBase, N, HALO_SLACK, [&]() { dRxT.release(); }, dMerged, nMerged, nullptr);
        if (rc) return rc;
        moveBuf(dA, dMerged);                                 // the caller's dA owns the result
        nTriples = nMerged;
    } else moveBuf(dA, dTriples);
    PH_CHECK(hipEventRecord(ctx->ev[13], st));
    PH_TRACE(st, "kmermatch: rep sort (line store)");
    PH_CHECK(hipGetLastError());
    res.triples = dA.p; res.nTriples = nTriples;
    { float msS = 0, msS2 = 0; (void) hipEventElapsedTime(&msS, ctx->ev[8], ctx->ev[9]); if (res.nPart == 2) (void) hipEventElapsedTime(&msS2, ctx->ev[10], ctx->ev[11]); res.msPart = msS + msS2; }
    return PLASSHIP_OK;
}


// ---- best diagonal per (rep, target) run over the sorted weighted triples, CSR of the candidate list (kmermatcher.cpp:809-924) ----
// `cur`: nTriples triples in (rep, target, diagonal) order (sharded run: room for HALO_SLACK more behind them)
template <bool NUCL, bool LONG>
static int reduceToCandidates(plasship_ctx *ctx, const plasship_seqdb *db, void *cur, uint64_t nTriples, const std::vector<int64_t> &stalePos, uint32_t staleT,
                              std::unique_ptr<plasship_cands> &holder, uint64_t &Nc, float &msReduce, bool lazyClock = false) {
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    Timer tm{ctx, 0};
    const plasship_comm *cm = commOf(ctx);
    const int W = cm ? cm->world : 1, rk = cm ? cm->rank : 0;
    const uint64_t repBase = ownedBegin(N, rk, W), ownedN = ownedBegin(N, rk + 1, W) - repBase;
    // ---- per-(rep,target) reduction + CSR ----
    if (!lazyClock) tm.start(0);                             // (lazyClock: the caller recorded ev[13] and reads ev[13] .. ev[14] when the call is over)
    DevBuf dTmpHits, dEmit, dEpos, dPerRep, dQoff;
    if (dTmpHits.alloc(std::max<uint64_t>(nTriples, 1) * sizeof(CandHit)) != hipSuccess || dEmit.alloc((nTriples / 64 + 2) * 4) != hipSuccess ||
        dEpos.alloc((nTriples / 64 + 3) * 8) != hipSuccess || dPerRep.alloc(((size_t) N + 1) * 4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    DevBuf dScanTmp2; const size_t scanTmp2Bytes = exclusiveScanTmpBytes(std::max<uint64_t>(nTriples, N) + 2);
    if (dScanTmp2.alloc(scanTmp2Bytes) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    uint64_t nHalo = 0;
    if (cm) {
        // The reference's run scan tests only the target id (Appendix A.3): the last run of this rank continues into the
        // triples of the next ranks while they carry the same target, and behind the last rank into the stale records.
        // Every rank publishes the head of its triples (the leading ones with one target); each rank appends what its last
        // run can reach behind its own triples.
        Triple *dTr = reinterpret_cast<Triple *>(cur);
        std::vector<Triple> head; uint64_t headCnt = 0;
        if (nTriples) {
            uint64_t want = std::min<uint64_t>(nTriples, 1024);
            for (;;) {
                head.resize(want);
                PH_COPY_SYNC(st, head.data(), dTr, want * sizeof(Triple), hipMemcpyDeviceToHost);
                headCnt = 0; while (headCnt < want && head[headCnt].target == head[0].target) headCnt++;
                if (headCnt < want || want == nTriples) break;
                want = std::min<uint64_t>(nTriples, want * 2);
            }
            head.resize(headCnt);
        }
        Triple last; memset(&last, 0, sizeof(last));
        if (nTriples) PH_COPY_SYNC(st, &last, dTr + (nTriples - 1), sizeof(Triple), hipMemcpyDeviceToHost);
        uint64_t hdr[2] = {nTriples, headCnt}; std::vector<uint64_t> hdrs(2 * (size_t) W);
        int rc = commAllgatherHost(ctx, hdr, hdrs.data(), 16); if (rc) return rc;
        uint64_t maxHead = 0, sumHead = 0; for (int r = 0; r < W; r++) { maxHead = std::max(maxHead, hdrs[2 * (size_t) r + 1]); sumHead += hdrs[2 * (size_t) r + 1]; }
        // the halo of any rank is at most all heads plus the stale records: decide on THAT, so every rank takes the same exit
        // (a rank that fails alone would leave the others waiting in the next collective)
        if (sumHead + stalePos.size() > HALO_SLACK) { setError("kmermatch: a (rep, target) run continues over more than 65536 records of other ranks"); return PLASSHIP_ERR_UNSUPPORTED; }
        std::vector<Triple> heads;
        if (maxHead) {
            std::vector<Triple> mine(maxHead); memset(mine.data(), 0, maxHead * sizeof(Triple));
            std::copy(head.begin(), head.end(), mine.begin());
            heads.resize(maxHead * (size_t) W);
            rc = commAllgatherHost(ctx, mine.data(), heads.data(), maxHead * sizeof(Triple)); if (rc) return rc;
        }
        if (nTriples) {
            std::vector<Triple> halo; bool open = true;          // open: the scan has not met another target yet
            for (int r = rk + 1; r < W && open; r++) {
                const uint64_t nr = hdrs[2 * (size_t) r], hr = hdrs[2 * (size_t) r + 1];
                if (nr == 0) continue;
                const Triple *hp = heads.data() + maxHead * (size_t) r;
                if (hp[0].target != last.target) { open = false; break; }
                halo.insert(halo.end(), hp, hp + hr);
                if (hr < nr) open = false;
            }
            if (open && !stalePos.empty() && staleT == last.target) {
                for (int64_t sp : stalePos) {      // stale records: pos = original k-mer position, kmer field = SIZE_T_MAX (forward)
                    Triple t; t.rep = 0xFFFFFFFFu; t.target = staleT; t.diag = LONG ? (int32_t) sp : (int32_t) (int16_t) sp; t.cnt = 1u | 0x80000000u;
                    halo.push_back(t);
                }
            }
            nHalo = halo.size();
            if (nHalo) PH_COPY_SYNC(st, dTr + nTriples, halo.data(), nHalo * sizeof(Triple), hipMemcpyHostToDevice);
        }
        PH_CHECK(hipMemsetAsync(dPerRep.p, 0, ((size_t) N + 1) * 4, st));
        if (ownedN) hipLaunchKernelGGL(fillU32Kernel, dim3(gridFor(ownedN, 256, 4096)), dim3(256), 0, st, dPerRep.as<uint32_t>() + repBase, 1u, ownedN);
    } else
    hipLaunchKernelGGL(fillU32Kernel, dim3(gridFor((uint64_t) N + 1, 256, 4096)), dim3(256), 0, st, dPerRep.as<uint32_t>(), 1u, (uint64_t) N);
    if (nTriples) hipLaunchKernelGGL((reduceRunsKernel<NUCL>), dim3(gridFor(nTriples, 256, 65535)), dim3(256), 0, st, (const Triple *) cur, nTriples, nTriples + nHalo, dTmpHits.as<CandHit>(), dEmit.as<uint32_t>(), dPerRep.as<uint32_t>());
    const uint64_t nWaves = (nTriples + 63) / 64;              // candidates are counted per 64 triples (reduceRunsKernel)
    if (exclusiveScanU32(st, dEmit.as<uint32_t>(), dEpos.as<uint64_t>(), nWaves, dScanTmp2.p, scanTmp2Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    holder.reset(new plasship_cands());                              // released to the caller on success only
    plasship_cands *c = holder.get();
    c->reverseCapable = NUCL; c->nQueries = N;
    if (c->d_qoff.alloc(((size_t) N + 1) * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (exclusiveScanU32(st, dPerRep.as<uint32_t>(), c->d_qoff.as<uint64_t>(), N, dScanTmp2.p, scanTmp2Bytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    Nc = 0;
    PH_CHECK(hipMemcpyAsync(&Nc, dEpos.as<uint64_t>() + nWaves, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    const uint32_t qLo = cm ? (uint32_t) repBase : 0u, qHi = cm ? (uint32_t) (repBase + ownedN) : N;      // queries with a self line
    c->nHits = Nc + (qHi - qLo); c->nNonSelf = Nc;
    if (c->d_hits.alloc(std::max<uint64_t>(c->nHits, 1) * sizeof(CandHit)) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    if (qHi > qLo) hipLaunchKernelGGL(placeSelfKernel, dim3(gridFor(qHi - qLo, 256, 4096)), dim3(256), 0, st, c->d_qoff.as<uint64_t>(), qLo, qHi, c->d_hits.as<CandHit>());
    if (nTriples) hipLaunchKernelGGL(placeHitsKernel, dim3(gridFor(nTriples, 256, 65535)), dim3(256), 0, st, dTmpHits.as<CandHit>(), dEpos.as<uint64_t>(), nTriples, qLo, c->d_hits.as<CandHit>());
    if (lazyClock) PH_CHECK(hipEventRecord(ctx->ev[14], st)); else msReduce = tm.stop(1);
    PH_TRACE(st, "kmermatch: reduce");
    PH_CHECK(plasship::streamSync(st));
    PH_CHECK(hipGetLastError());
    if (!stalePos.empty() && nTriples > 0 && !cm) {
        // the runs that end at the very end of the sorted array (the last (rep,T) run, and the T-runs of directly preceding
        // reps whose scan the reference lets run across the rep boundary) continue into the stale records: redo them
        const Triple *dTr = reinterpret_cast<const Triple *>(cur);
        std::vector<Triple> tail; uint64_t want = std::min<uint64_t>(nTriples, 4096);
        for (;;) {
            tail.resize(want);
            PH_COPY_SYNC(st, tail.data(), dTr + (nTriples - want), want * sizeof(Triple), hipMemcpyDeviceToHost);
            if (tail.front().target != staleT || want == nTriples) break;
            want = std::min<uint64_t>(nTriples, want * 2);
        }
        if (tail.back().target == staleT) {
            size_t b0 = tail.size(); while (b0 > 0 && tail[b0 - 1].target == staleT) b0--;
            for (size_t h0 = b0; h0 < tail.size(); h0++) {
                if (!(h0 == b0 || tail[h0].rep != tail[h0 - 1].rep)) continue;       // not a run head
                int32_t diagonal = tail[h0].diag, prevDiagonal = tail[h0].diag;
                uint64_t maxDiagonal = 0, diagonalCnt = 0, topScore = 0;
                int bestRev = NUCL ? ((tail[h0].cnt & 0x80000000u) == 0) : 0;
                for (size_t j = h0; j < tail.size(); j++) {
                    const uint64_t cc = tail[j].cnt & 0x7FFFFFFFu;
                    if (prevDiagonal == tail[j].diag) diagonalCnt += cc; else diagonalCnt = cc;
                    if (diagonalCnt >= maxDiagonal) { diagonal = tail[j].diag; maxDiagonal = diagonalCnt; if (NUCL) bestRev = ((tail[j].cnt & 0x80000000u) == 0); }
                    prevDiagonal = tail[j].diag; topScore += cc;
                }
                for (int64_t sp : stalePos) {          // stale records: pos = original k-mer position, kmer field = SIZE_T_MAX (forward)
                    const int32_t d = LONG ? (int32_t) sp : (int32_t) (int16_t) sp;
                    if (prevDiagonal == d) diagonalCnt++; else diagonalCnt = 1;
                    if (diagonalCnt >= maxDiagonal) { diagonal = d; maxDiagonal = diagonalCnt; if (NUCL) bestRev = 0; }
                    prevDiagonal = d; topScore++;
                }
                const uint32_t rep = tail[h0].rep;
                if (rep == staleT) continue;                                       // self run: scanned but never emitted
                uint64_t qn = 0;
                PH_COPY_SYNC(st, &qn, c->d_qoff.as<uint64_t>() + rep + 1, 8, hipMemcpyDeviceToHost);
                CandHit hh;
                PH_COPY_SYNC(st, &hh, c->d_hits.as<CandHit>() + (qn - 1), sizeof(CandHit), hipMemcpyDeviceToHost);
                if (hh.target != staleT || hh.query != rep) { setError("kmermatch: internal error while patching the last run"); return PLASSHIP_ERR_DEVICE; }
                hh.prefScore = bestRev ? -(int) topScore : (int) topScore; hh.diag16 = (uint32_t) (uint16_t) diagonal;
                PH_COPY_SYNC(st, c->d_hits.as<CandHit>() + (qn - 1), &hh, sizeof(CandHit), hipMemcpyHostToDevice);
            }
        }
    }
    return PLASSHIP_OK;
}

template <bool NUCL, bool LONG>
int kmermatchImpl(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par, plasship_cands **out,
                  plasship_kmermatch_stats *stats, bool overflowCheckEarly) {
    typedef Rec<LONG> R;
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const int k = par->kmer_size;
    float msExtract = 0;
    // sharded run (plasship_ctx_set_comm): this rank extracts the sequences [sLo, sHi), owns the k-mer hash buckets that map to
    // it, and owns the representatives / queries [repBase, repBase + ownedN)
    const plasship_comm *cm = commOf(ctx);
    const int W = cm ? cm->world : 1, rk = cm ? cm->rank : 0;

    // ---- slot bounds + offsets ----
    DevBuf dBound, dSlotOff, dScanTmp;
    const size_t scanTmpBytes = exclusiveScanTmpBytes((size_t) N + 2) + (1u << 20);
    if (dBound.alloc(((size_t) N + 1) * 4) != hipSuccess || dSlotOff.alloc(((size_t) N + 2) * 8) != hipSuccess || dScanTmp.alloc(scanTmpBytes) != hipSuccess) {
        setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    if (N) hipLaunchKernelGGL(boundsKernel, dim3(gridFor(N, 256, 4096)), dim3(256), 0, st, db->d_len.as<uint32_t>(), N, k, par->kmers_per_seq, par->kmers_per_seq_scale, dBound.as<uint32_t>());
    if (exclusiveScanU32(st, dBound.as<uint32_t>(), dSlotOff.as<uint64_t>(), N, dScanTmp.p, scanTmpBytes)) { setError("kmermatch: scan failed"); return PLASSHIP_ERR_DEVICE; }
    uint64_t total = 0;
    uint32_t sLo = 0, sHi = N; uint64_t slotBias = 0, totalAll = 0;
    if (cm) {
        DevBuf dSplit; std::vector<uint64_t> hSplit(2 * (size_t) W + 2);
        if (dSplit.alloc(hSplit.size() * 8) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(splitIdsKernel, dim3(1), dim3(256), 0, st, dSlotOff.as<uint64_t>(), N, W, dSplit.as<uint64_t>());
        PH_COPY_SYNC(st, hSplit.data(), dSplit.p, hSplit.size() * 8, hipMemcpyDeviceToHost);
        sLo = (uint32_t) hSplit[rk]; sHi = (uint32_t) hSplit[rk + 1]; slotBias = hSplit[W + 1 + rk];
        total = hSplit[W + 1 + rk + 1] - slotBias;              // slots of this rank's share
        totalAll = hSplit[2 * (size_t) W + 1];                  // ... and of all sequences
    } else {
    PH_CHECK(hipMemcpyAsync(&total, dSlotOff.as<uint64_t>() + N, 8, hipMemcpyDeviceToHost, st));
    PH_CHECK(plasship::streamSync(st));
    }
    const uint32_t nMine = sHi - sLo;

    DevBuf dA, dB;   // ping-pong record arrays
    // the line-store partition (linepart.hpp): its record buffers hold whole lines plus one partial line per bucket and piece
    if (cm && W > 1024) { setError("kmermatch: more than 1024 ranks"); return PLASSHIP_ERR_UNSUPPORTED; }
    const LineGeo geo = lineGeometry(total, LONG, ctx->numCU, cm ? totalAll : 0, W);
    // single GPU: both buffers serve level 1 and level 2 (and the group kernel's arenas); sharded run: the slot array / level-1 output,
    // and the packed send buffer of exchange 1 (at most cap1 lines)
    const uint64_t recCap = std::max<uint64_t>(total, (uint64_t) RPL * (cm ? geo.cap1 : std::max(geo.cap1, geo.cap2)));
    if (dA.alloc(std::max<uint64_t>(recCap, 1) * sizeof(R)) != hipSuccess || dB.alloc(std::max<uint64_t>(recCap, 1) * sizeof(R)) != hipSuccess) {
        setError("kmermatch: out of device memory for the k-mer record arrays (" + std::to_string(2 * recCap * sizeof(R)) + " bytes)"); return PLASSHIP_ERR_DEVICE;
    }

    // ---- extraction ----
    DevBuf dMap, dOvIds, dOvCnt;
    if (dMap.alloc(256) != hipSuccess || dOvIds.alloc(((size_t) N + 1) * 4) != hipSuccess || dOvCnt.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    const unsigned char *map = aa2numTable(NUCL, par->alphabet_size);
    PH_CHECK(hipMemcpyAsync(dMap.p, map, 256, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemsetAsync(dOvCnt.p, 0, 4, st));
    ExtractArgs ea; memset(&ea, 0, sizeof(ea));
    ea.s = db->view(); ea.slotOff = dSlotOff.as<uint64_t>(); ea.arr = dA.p; ea.map = dMap.as<unsigned char>();
    const int alph = NUCL ? 5 : par->alphabet_size;
    { uint64_t p = 1; for (int i = 0; i < 24; i++) { ea.powers[i] = p; p *= (uint64_t) (alph - 1); } }
    ea.k = k; ea.xCode = map[(int) 'X']; ea.kps = par->kmers_per_seq; ea.ignoreMulti = par->ignore_multi_kmer; ea.scale = par->kmers_per_seq_scale;
    ea.seed = (uint64_t) par->hash_shift; ea.overflowIds = dOvIds.as<uint32_t>(); ea.overflowCount = dOvCnt.as<uint32_t>();
    ea.idLo = sLo; ea.idHi = sHi; ea.slotBias = slotBias;
    // candidate k-mers per sequence held in LDS.  128 covers every protein sequence (59 considered k-mers) and every
    // nucleotide sequence up to ~690 nt (59 + 0.1 L); the 1024-candidate instantiation (35 KB of LDS, one wavefront per SIMD)
    // only sees the longer nucleotide contigs, queued by the first launch; what does not fit there either goes to the
    // HBM-scratch launch.
    constexpr int CAP = 128, CAP2 = NUCL ? 1024 : 128;
    DevBuf dWaveList, dWaveCount, dLongList, dLongCount, dKStats;
    static const bool tier0 = [] { const char *e = getenv("PLASSHIP_TIER0"); return e ? atoi(e) != 0 : true; }();
    constexpr uint32_t TIER0_WINDOWS = 256;                  // 4 scores per lane (an 8-scores tier at 5 wavefronts per SIMD gained 0.3 %: not kept)
    if (dWaveList.alloc(((size_t) N + 1) * 4) != hipSuccess || dWaveCount.alloc(4) != hipSuccess || dLongList.alloc(((size_t) N + 1) * 4) != hipSuccess || dLongCount.alloc(4) != hipSuccess ||
        dKStats.alloc(64) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync(dWaveCount.p, 0, 4, st));
    PH_CHECK(hipMemsetAsync(dLongCount.p, 0, 4, st));
    PH_CHECK(hipMemsetAsync(dKStats.p, 0, 64, st));
    ea.kstats = dKStats.as<unsigned long long>();
    // ---- selected-window cache (section 2c): which sequences keep last call's selection, and where this call's selection is kept ----
    plasship_ctx::KmCache &kc = ctx->kmCache;
    const bool fastIndex = !NUCL && k <= 14 && ea.powers[1] <= 16;      // kmerIndexCore
    const bool cacheEligible = !NUCL && !LONG && !cm && fastIndex && k <= 16 && par->kmers_per_seq >= 1 && par->kmers_per_seq <= (int) KMC_POS + 1 && par->kmers_per_seq_scale == 0.0f &&
                               tuneInt("KMCACHE", 1) == 1;      // PLASSHIP_TUNE_KMCACHE=2 switches the cache off
    const bool cacheReuse = cacheEligible && kc.valid && kc.n == N && kc.gen == db->parentGen && db->d_changed.p && kc.k == k && kc.alph == par->alphabet_size &&
                            kc.kps == par->kmers_per_seq && kc.ignoreMulti == par->ignore_multi_kmer && kc.hashShift == par->hash_shift && !overflowCheckEarly;
    kc.valid = false;                                         // (set again when this call has succeeded)
    DevBuf dCachedList, dCachedCount;
    unsigned long long cacheLinesPtr = 0;                    // -> kstats[4] (see ExtractArgs)
    if (cacheEligible && N) {
        if (kc.lines.bytes < (size_t) N * KMC_LINE) { kc.lines.release(); if (kc.lines.allocLong((size_t) N * KMC_LINE) != hipSuccess) { setError("kmermatch: out of device memory for the selected-window cache"); return PLASSHIP_ERR_DEVICE; } }
        cacheLinesPtr = (unsigned long long) (uintptr_t) kc.lines.p;
        if (cacheReuse) {
            if (dCachedList.alloc(((size_t) N + 1) * 4) != hipSuccess || dCachedCount.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
            PH_CHECK(hipMemsetAsync(dCachedCount.p, 0, 4, st));
        }
    } else kc.lines.release();
    PH_CHECK(hipMemcpyAsync(dKStats.as<unsigned long long>() + 4, &cacheLinesPtr, 8, hipMemcpyHostToDevice, st));
    PH_CHECK(hipEventRecord(ctx->ev[2], st));
    bool twoLists = false;
    if (!NUCL && k <= 16 && nMine) {
        // short sequences: one thread each; everything else is queued for the wave-per-sequence kernels, in two lists by length
        ShortArgs sa; memset(&sa, 0, sizeof(sa));
        sa.s = ea.s; sa.slotOff = ea.slotOff; sa.arr = ea.arr; sa.map = ea.map; sa.k = k; sa.xCode = ea.xCode; sa.kps = ea.kps; sa.ignoreMulti = ea.ignoreMulti;
        sa.scale = ea.scale; sa.seed = ea.seed; sa.base = (uint64_t) (alph - 1); sa.top = ea.powers[k - 1];
        { uint64_t b = sa.base; int tz = 0; while ((b & 1) == 0) { b >>= 1; tz++; } uint64_t inv = b; for (int i = 0; i < 6; i++) inv *= 2 - b * inv; sa.tz = tz; sa.inv = inv; }
        sa.waveList = dWaveList.as<uint32_t>(); sa.waveCount = dWaveCount.as<uint32_t>(); sa.kstats = dKStats.as<unsigned long long>();
        if (tier0) {
            sa.longList = dLongList.as<uint32_t>(); sa.longCount = dLongCount.as<uint32_t>(); sa.longWindows = TIER0_WINDOWS; twoLists = true;
            // more than 16 scores per lane: straight into the queue the 48-scores tier reads (tiers 0 and 1 append their rare overflows
            // to the same queue, one atomic per sequence)
            sa.hugeList = dOvIds.as<uint32_t>(); sa.hugeCount = dOvCnt.as<uint32_t>(); sa.hugeWindows = 64 * 16;
        }
        sa.idLo = sLo; sa.idHi = sHi; sa.slotBias = slotBias;
        if (cacheReuse) { sa.changed = db->d_changed.as<unsigned char>(); sa.cachedList = dCachedList.as<uint32_t>(); sa.cachedCount = dCachedCount.as<uint32_t>(); }
        // the fast restatement needs k = 14 and the half-indices (7 digits of the base, each at most the X code = base) in 32 bits
        constexpr int KF = 14, HF = KF / 2;
        bool fast = k == KF && tuneInt("SHORT_FAST", 1) == 1;      // PLASSHIP_TUNE_SHORT_FAST=2: the kernel above
        uint64_t pwH = 1; for (int i = 0; i < HF; i++) pwH *= sa.base;
        if (pwH * 2 >= (1ull << 32)) fast = false;
        sa.topLo = (uint32_t) (pwH / sa.base); sa.topHi = (uint32_t) ea.powers[KF - HF - 1]; sa.baseH = (uint32_t) pwH;
        const dim3 shortGrid(std::min<uint32_t>((nMine + 63) / 64, (uint32_t) ctx->numCU * (uint32_t) tuneInt("SHORT", nMine > 20000000u ? 72 : 18)));
        if (fast && sa.base < (1u << 8) && sa.topLo < (1u << 24) && sa.topHi < (1u << 24)) hipLaunchKernelGGL((extractShortFastKernel<LONG, KF, true>), shortGrid, dim3(64), 0, st, sa);
        else if (fast) hipLaunchKernelGGL((extractShortFastKernel<LONG, KF, false>), shortGrid, dim3(64), 0, st, sa);
        else hipLaunchKernelGGL((extractShortKernel<LONG>), shortGrid, dim3(64), 0, st, sa);   // 18 wavefronts fit a CU; on large sets twice that evens out the tail (50 M reads: 35.7 -> 34.4 ms)
        ea.waveList = dWaveList.as<uint32_t>(); ea.waveCount = dWaveCount.as<uint32_t>();
        if (cacheReuse) {
            CachedArgs ca; memset(&ca, 0, sizeof(ca));
            ca.s = ea.s; ca.slotOff = ea.slotOff; ca.arr = ea.arr; ca.map = ea.map; ca.lines = kc.lines.as<unsigned char>();
            ca.list = dCachedList.as<uint32_t>(); ca.count = dCachedCount.as<uint32_t>(); ca.k = k; ca.xCode = ea.xCode;
            ca.base = (uint32_t) ea.powers[1]; ca.base7 = (uint32_t) ea.powers[7]; ca.slotBias = slotBias; ca.seed = ea.seed; ca.kstats = dKStats.as<unsigned long long>();
            hipLaunchKernelGGL(extractCachedKernel, dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) tuneInt("CACHED", 32))), dim3(64), 0, st, ca);
        }
    }
    else if (nMine && tier0 && tuneInt("CLASSIFY", 1) == 1) {                   // PLASSHIP_TUNE_CLASSIFY=2: the 16-scores tier takes every id itself
        hipLaunchKernelGGL(classifyWindowsKernel, dim3(std::min<uint32_t>((nMine + 2047) / 2048, (uint32_t) ctx->numCU * 8)), dim3(256), 0, st, (const uint32_t *) db->d_len.as<uint32_t>(), sLo, sHi, (uint32_t) k,
                           TIER0_WINDOWS, 64u * 16u, dWaveList.as<uint32_t>(), dWaveCount.as<uint32_t>(), dLongList.as<uint32_t>(), dLongCount.as<uint32_t>(), dOvIds.as<uint32_t>(), dOvCnt.as<uint32_t>());
        ea.waveList = dWaveList.as<uint32_t>(); ea.waveCount = dWaveCount.as<uint32_t>();
        twoLists = true;
    }
    PH_CHECK(hipEventRecord(ctx->ev[3], st));
    PH_CHECK(hipEventRecord(ctx->ev[4], st));
    static const int waveBlocksPerCU = [] { const char *e = getenv("PLASSHIP_EXTRACT_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 32; }();     // 32 one-wavefront workgroups per CU: measured best of 12..64 on the 1 M-read set
    // launch chain, each tier queueing what it cannot hold for the next: (1) register front end, up to 1024 windows (every read,
    // most contigs); (2) the same with 48 scores per lane, up to 3072 windows (proteins longer than that are rare) and, for
    // nucleotides, up to 1024 candidates; (3) three-pass path with codes and scores of up to 8160 residues resident in LDS;
    // (4) the HBM-scratch launch below
    // (0) the same register front end with 4 scores per lane (up to 256 windows: merged read fragments and young contigs) at six
    // wavefronts per SIMD instead of four: the per-sequence steps (staging, ballots, LDS round trips between barriers) are what a
    // 100-250 residue sequence mostly consists of, and more resident wavefronts hide them (2.1 instead of 2.8 ns per sequence).
    // Tiers 0 and 1 take their sequences from the two lists of the thread-per-sequence kernel; what either cannot hold goes to
    // one queue for tier 2, and on to tier 3.
    DevBuf dOv2Ids, dOv2Cnt;
    DevBuf *lastIds = &dOvIds, *lastCnt = &dOvCnt;
    if (nMine) {
        if (dOv2Ids.alloc(((size_t) N + 1) * 4) != hipSuccess || dOv2Cnt.alloc(4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        PH_CHECK(hipMemsetAsync(dOv2Cnt.p, 0, 4, st));
        const uint32_t wide = std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (uint32_t) waveBlocksPerCU);
        // tiers 0 and 1 both queue into dOvIds
        if (twoLists) {
            if (tuneInt("TIER0_WPE", 5) == 6) hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 4, 256, 6>), dim3(wide), dim3(64), 0, st, ea);
            else hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 4, 256>), dim3(wide), dim3(64), 0, st, ea);
            ExtractArgs e1 = ea; e1.waveList = dLongList.as<uint32_t>(); e1.waveCount = dLongCount.as<uint32_t>();
            hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 16, 992>), dim3(wide), dim3(64), 0, st, e1);
        } else hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP, false, 16, 992>), dim3(wide), dim3(64), 0, st, ea);
        ExtractArgs e2 = ea; e2.waveList = dOvIds.as<uint32_t>(); e2.waveCount = dOvCnt.as<uint32_t>();
        e2.overflowIds = dOv2Ids.as<uint32_t>(); e2.overflowCount = dOv2Cnt.as<uint32_t>();
        // (round 4: a nucleotide sequence of this tier has at most 3 072 windows, i.e. 60 + 0.1 L <= 370 selected k-mers at the workflow's
        //  scaling: 512 candidates instead of 1 024 halve its LDS and let eight instead of four wavefronts work per CU; a candidate set
        //  that does not fit is queued for the next tier like any other overflow)
        constexpr int CAP48 = NUCL ? 512 : 128;
        if (NUCL && tuneInt("NUCL_CAP48", 1) == 1) hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP48, false, 48, 992, NUCL ? 2 : 0>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * 8u)), dim3(64), 0, st, e2);
        else hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP2, false, 48, 992>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (NUCL ? 4u : 16u))), dim3(64), 0, st, e2);
        PH_CHECK(hipMemsetAsync(dOvCnt.p, 0, 4, st));            // tier 2 has consumed the first queue: it becomes tier 3's output queue
        ExtractArgs e3 = ea; e3.waveList = dOv2Ids.as<uint32_t>(); e3.waveCount = dOv2Cnt.as<uint32_t>();
        e3.overflowIds = dOvIds.as<uint32_t>(); e3.overflowCount = dOvCnt.as<uint32_t>();
        hipLaunchKernelGGL((extractKernel<NUCL, LONG, CAP2, false, 0, 8160>), dim3(std::min<uint32_t>(nMine, (uint32_t) ctx->numCU * (NUCL ? 2u : 5u))), dim3(64), 0, st, e3);
    }
    // what the last stage could not hold either (candidate sets beyond LDS)
    DevBuf &dLastIds = *lastIds, &dLastCnt = *lastCnt;
    PH_CHECK(hipEventRecord(ctx->ev[5], st));
    uint32_t nOv = 0;
    // protein DBs: a candidate set hardly ever exceeds the tiers' 128 entries (59 considered k-mers), and the last tier keeps up to
    // 8160 residues in LDS.  "Hardly ever": every window whose score is <= the threshold score is a candidate, and equal k-mers
    // share a score — a homopolymer or tandem repeat of >= ~85 residues whose k-mer lies at or below the threshold overfills the set
    // (ADVICE r3).  So the count of the last tier's hand-overs is not waited for here when an overflow is UNLIKELY; it travels with
    // the group stage's counts (kmermatchLines), the last tier has left such a sequence's slots as sentinels, and if the count turns
    // out non-zero the call starts over with the wait in place (`overflowCheckEarly`).
    const bool overflowPossible = overflowCheckEarly || cm != nullptr || NUCL || db->maxEntryLen > 8160u || par->kmers_per_seq > 120 || par->kmers_per_seq_scale != 0.0f;
    if (nMine && overflowPossible) {
        PH_CHECK(hipMemcpyAsync(&nOv, dLastCnt.p, 4, hipMemcpyDeviceToHost, st));
        PH_CHECK(plasship::streamSync(st));
    } else if (nMine) hipLaunchKernelGGL((fillOverflowSlotsKernel<LONG>), dim3(64), dim3(64), 0, st, (const uint32_t *) dLastIds.as<uint32_t>(), (const uint32_t *) dLastCnt.as<uint32_t>(),
                                         (const uint64_t *) dSlotOff.as<uint64_t>(), slotBias, dA.p);
    PH_CHECK(hipGetLastError());
    if (nOv) {   // sequences whose candidate set did not fit LDS: same kernel, candidates in HBM scratch
        std::vector<uint32_t> ids(nOv), lens(nOv);
        DevBuf dLens, dSOff, dSCap, dScratch;
        if (dLens.alloc((size_t) nOv * 4) != hipSuccess) { setError("kmermatch: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        hipLaunchKernelGGL(gatherU32Kernel, dim3(gridFor(nOv, 256, 1024)), dim3(256), 0, st, db->d_len.as<uint32_t>(), dLastIds.as<uint32_t>(), nOv, dLens.as<uint32_t>());
        PH_CHECK(hipMemcpyAsync(lens.data(), dLens.p, (size_t) nOv * 4, hipMemcpyDeviceToHost, st));
        PH_CHECK(plasship::streamSync(st));
        std::vector<uint64_t> soff(nOv); std::vector<uint32_t> scap(nOv); uint64_t tot = 0;
        for (uint32_t i = 0; i < nOv; i++) { uint32_t c = 64; while (c < lens[i] + 1) c <<= 1; scap[i] = c; soff[i] = tot; tot += c; }
        if (dSOff.alloc((size_t) nOv * 8) != hipSuccess || dSCap.alloc((size_t) nOv * 4) != hipSuccess || dScratch.alloc(tot * sizeof(Cand)) != hipSuccess) {
            setError("kmermatch: out of device memory for the long-sequence scratch"); return PLASSHIP_ERR_DEVICE;
        }
        PH_CHECK(hipMemcpyAsync(dSOff.p, soff.data(), (size_t) nOv * 8, hipMemcpyHostToDevice, st));
        PH_CHECK(hipMemcpyAsync(dSCap.p, scap.data(), (size_t) nOv * 4, hipMemcpyHostToDevice, st));
        ExtractArgs fa = ea; fa.idList = dLastIds.as<uint32_t>(); fa.nIds = nOv; fa.scratch = dScratch.as<Cand>(); fa.scratchOff = dSOff.as<uint64_t>(); fa.scratchCap = dSCap.as<uint32_t>();
        hipLaunchKernelGGL((extractKernel<NUCL, LONG, 1, true>), dim3(std::min<uint32_t>(nOv, (uint32_t) ctx->numCU * 8)), dim3(64), 0, st, fa);
        PH_CHECK(plasship::streamSync(st));
        PH_CHECK(hipGetLastError());
    }
    PH_TRACE(st, "kmermatch: extraction");
    traceBadIds<LONG>(ctx, "kmermatch: extracted slots", dA.p, total, N);

    {
        int keyBitsL = 0;
        if (NUCL) keyBitsL = 2 * k; else { long double v = 1; for (int i = 0; i < k; i++) v *= (long double) (alph - 1); while (keyBitsL < 63 && (long double) (1ULL << keyBitsL) < v) keyBitsL++; }
        LinesOut lo;
        int rcL = kmermatchLines<NUCL, LONG>(ctx, db, par, geo, total, dA, dB, dSlotOff, dKStats, ea, keyBitsL, lo, (nMine && !overflowPossible) ? dLastCnt.as<uint32_t>() : nullptr);
        if (rcL) return rcL;                                  // (KM_RETRY_EARLY_OVERFLOW_CHECK: the caller starts over)
        std::unique_ptr<plasship_cands> holderL; uint64_t NcL = 0; float msReduceL = 0;
        rcL = reduceToCandidates<NUCL, LONG>(ctx, db, lo.triples, lo.nTriples, lo.stalePos, lo.staleT, holderL, NcL, msReduceL, true);
        if (rcL) return rcL;
        {   // the stage boundaries (all recorded, all complete: reduceToCandidates ended with a wait for the stream)
            auto ms = [&](int a, int b) { float v = 0; (void) hipEventElapsedTime(&v, ctx->ev[a], ctx->ev[b]); return v; };
            msExtract = ms(0, 1); lo.msSort1 = ms(1, 6); lo.msGroup = ms(6, 7); lo.msSort2 = ms(12, 13); msReduceL = ms(13, 14);
        }
        if (stats) {
            stats->n_kmer_records = lo.Nk; stats->n_grouped = lo.Nm; stats->n_candidates = NcL; stats->record_bytes = LONG ? 20 : 16;
            float ms = 0, ms2 = 0; (void) hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]); (void) hipEventElapsedTime(&ms2, ctx->ev[4], ctx->ev[5]);
            stats->ms_extract_short_kernel = ms; stats->ms_extract_wave_kernel = ms2; stats->ms_extract_kernel = ms + ms2;
            stats->ms_part_scatter = lo.msPart; stats->n_part_scatter = lo.nPart;
            unsigned long long ks[4] = {0, 0, 0, 0}; uint32_t nCachedSeqs = 0;
            if (cacheReuse) PH_CHECK(hipMemcpyAsync(&nCachedSeqs, dCachedCount.p, 4, hipMemcpyDeviceToHost, st));
            PH_COPY_SYNC(st, ks, dKStats.p, 32, hipMemcpyDeviceToHost);
            stats->n_cached_sequences = nCachedSeqs; stats->reserved0 = 0;
            stats->short_residues = ks[0]; stats->short_records = ks[1]; stats->wave_residues = ks[2]; stats->wave_records = ks[3];
            stats->residues = db->residues;
            stats->ms_extract = msExtract; stats->ms_sort1 = lo.msSort1; stats->ms_group = lo.msGroup; stats->ms_sort2 = lo.msSort2; stats->ms_reduce = msReduceL;
            stats->n_scratch_sequences = nOv; stats->n_restarts = overflowCheckEarly ? 1u : 0u;
        }
        if (cacheEligible && N) {     // the lines now describe THIS DB (the wave kernels rewrote what changed, the rest was kept)
            kc.valid = true; kc.n = N; kc.gen = db->gen; kc.k = k; kc.alph = par->alphabet_size; kc.kps = par->kmers_per_seq; kc.ignoreMulti = par->ignore_multi_kmer; kc.hashShift = par->hash_shift;
        }
        *out = holderL.release();
        return PLASSHIP_OK;
    }
}
}  // namespace

extern "C" int plasship_kmermatch(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par,
                                  plasship_cands **out, plasship_kmermatch_stats *stats) {
    if (!ctx || !db || !par || !out) { setError("plasship_kmermatch: bad argument"); return PLASSHIP_ERR_ARG; }
    const bool nucl = db->dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES;
    if (par->kmer_size < 2 || par->kmer_size > (nucl ? 31 : 23)) { setError("plasship_kmermatch: unsupported k"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (!nucl && par->alphabet_size != 13 && par->alphabet_size != 21) { setError("plasship_kmermatch: --alph-size must be 13 or 21"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (db->maxEntryLen >= (1u << 20)) { setError("plasship_kmermatch: sequences of 2^20 residues or more are not supported"); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    // kmermatcher.cpp:797-802: KmerPosition<short> unless a sequence is too long for it.  (Nucleotide k > 23: the 16-byte grouped
    // record has no room for the k-mer the strand rule needs — embedOrd — so such a run takes the 24-byte layout, whose arithmetic
    // is the same on sequences that short.)
    const bool lng = !(db->maxEntryLen < (uint32_t) SHRT_MAX) || (nucl && par->kmer_size > 23);
    int rc = KM_RETRY_EARLY_OVERFLOW_CHECK;
    for (int attempt = 0; attempt < 2 && rc == KM_RETRY_EARLY_OVERFLOW_CHECK; attempt++) {
        const bool early = attempt > 0;
        if (nucl) rc = lng ? kmermatchImpl<true, true>(ctx, db, par, out, stats, early) : kmermatchImpl<true, false>(ctx, db, par, out, stats, early);
        else rc = lng ? kmermatchImpl<false, true>(ctx, db, par, out, stats, early) : kmermatchImpl<false, false>(ctx, db, par, out, stats, early);
    }
    return commFinish(ctx, rc);
}
