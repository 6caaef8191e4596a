// plasship: kmermatcher on gfx950, stage K1-K3: slot bounds, extraction and selection.  Product code; part of kmermatch.hip's translation unit (included there, inside
// namespace plasship, after common.hpp / device_utils.hpp / linepart.hpp) — split out by stage in round 4, see kmermatch.hip for the
// reference lines the stage reproduces and DESIGN.md section 4 for the kernels' bounds.
// Kernels: boundsKernel, extractKernel (the wave-per-sequence tiers), extractShortKernel / extractShortFastKernel (thread per sequence), classifyWindowsKernel, extractCachedKernel (selected-window cache).
#pragma once

// (XXH64 of one u64 and the windows' 16-bit score: xxh64_u64.hpp, included by kmermatch.hip)
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
    //   kstats[4] = address of the selected-window cache lines (section 2c; 0 = no cache), fetched per sequence where it is used;
    //   kstats[5] / kstats[6] = position cache of nucleotide runs (section 2d; 0 = none): window position per record slot, identity hash per id.
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
    typedef typename std::conditional<LONG, uint32_t, unsigned short>::type PosT;      // one entry of the position cache (section 2d)
    R *arr = reinterpret_cast<R *>(a.arr);
    const int lane = threadIdx.x;
    const int k = a.k;
    // (round 6: a block behind the end of its list leaves before the set-up below — 63 dependent 64-bit products for 31^lane; the grids are
    //  192 blocks per CU and the lists of the later tiers, and of the 4-scores tier behind the row kernels, are often shorter than that)
    if (blockIdx.x >= (FALLBACK ? a.nIds : (a.waveList ? *a.waveCount : (a.idHi - a.idLo)))) return;
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
            // Round 6: what lies behind the 256 prefetched bytes is fetched SIXTEEN bytes per lane, all loads of a sequence in flight together.  (Until
            // then the loop below went on byte by byte — 64 bytes per load instruction, each load waited for before the next was issued: up to 13
            // dependent round trips for a 1 000-residue sequence, 45 in the 48-scores tier; the counters showed the tiers parked 50-60 % of their cycles.)
            {
                const uint32_t head = min(L + 31u, 256u);
                for (uint32_t i = lane; i < head; i += 64) {
                    const char ch = (i < 64) ? cb0 : ((i < 128) ? cb1 : ((i < 192) ? cb2 : cb3));      // the first 256 bytes were prefetched
                    sCodeAll[i] = (i < L) ? sMap[(unsigned char) ch] : (unsigned char) a.xCode;
                }
                if (L + 31u > 256u) {
                    constexpr int NCH = (int) ((CODES - 256u + 1023u) / 1024u);
                    uint4 wv[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; c++) { const uint32_t p = 256u + 1024u * (uint32_t) c + 16u * (uint32_t) lane; wv[c] = make_uint4(0, 0, 0, 0); if (p < L) __builtin_memcpy(&wv[c], base + p, 16); }      // (buffers are padded past their ends)
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        const uint32_t p = 256u + 1024u * (uint32_t) c + 16u * (uint32_t) lane;
                        if (p < L + 31u && p + 16u <= CODES) {
                            const uint32_t r4[4] = {wv[c].x, wv[c].y, wv[c].z, wv[c].w};
                            uint32_t cw[4];
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                uint32_t v = 0;
#pragma unroll
                                for (int b = 0; b < 4; b++) { const uint32_t q = p + 4u * (uint32_t) d + (uint32_t) b; v |= (uint32_t) ((q < L) ? sMap[(r4[d] >> (8 * b)) & 0xFFu] : (unsigned char) a.xCode) << (8 * b); }
                                cw[d] = v;
                            }
                            *reinterpret_cast<uint4 *>(&sCodeAll[p]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                        }
                    }
                }
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
                        else if (windowKmer(p, kmer, pos)) sc[j] = xxh64Score16(NUCL ? (kmer & ~BIT63) : kmer, a.seed);
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
                    if (valid) score = xxh64Score16(NUCL ? (kmer & ~BIT63) : kmer, a.seed);
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
            if constexpr (NUCL) {      // position cache (section 2d): window positions of the records just written, their count in the identity record's place
                if (!FALLBACK_NOSTATS(a)) {
                    PosT *pa = reinterpret_cast<PosT *>(a.kstats[5]);
                    if (pa) {
                        for (uint32_t i = lane; i < C; i += 64) { const Cand cd = cand[i]; pa[slot + 1 + i] = (PosT) ((cd.kmer & BIT63) ? cd.pos : (L - cd.pos - (uint32_t) k)); }
                        if (lane == 0) { pa[slot] = (PosT) C; reinterpret_cast<unsigned long long *>(a.kstats[6])[id] = seqHash; }
                    }
                }
            }
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
        PosT *posCache = nullptr;                                      // (section 2d)
        if constexpr (NUCL) { if (!FALLBACK_NOSTATS(a)) posCache = reinterpret_cast<PosT *>(a.kstats[5]); }
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
                if constexpr (NUCL) { if (posCache) posCache[slot + 1 + selRank] = (PosT) ((cd.kmer & BIT63) ? cd.pos : (L - cd.pos - (uint32_t) k)); }
            }
            binCarry += (uint32_t) __popcll(mb); selCarry += (uint32_t) __popcll(ms);
        }
        const uint32_t numSel = (uint32_t) min((size_t) selCarry, considered);
        if constexpr (NUCL) { if (posCache && lane == 0) { posCache[slot] = (PosT) numSel; reinterpret_cast<unsigned long long *>(a.kstats[6])[id] = seqHash; } }
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
// (round 6, section 2d: `origin` != nullptr — ids whose bytes are those of a sequence the position cache describes go to a fourth list)
__global__ __launch_bounds__(256) void classifyWindowsKernel(const uint32_t *__restrict__ len, uint32_t idLo, uint32_t idHi, uint32_t k, uint32_t longWindows, uint32_t hugeWindows,
                                                             uint32_t *__restrict__ waveList, uint32_t *__restrict__ waveCount, uint32_t *__restrict__ longList, uint32_t *__restrict__ longCount,
                                                             uint32_t *__restrict__ hugeList, uint32_t *__restrict__ hugeCount,
                                                             const uint32_t *__restrict__ origin, uint32_t *__restrict__ cachedList, uint32_t *__restrict__ cachedCount) {
    __shared__ uint32_t sCnt[4], sBase[4];
    constexpr int PER = 8;
    for (uint64_t b0 = (uint64_t) idLo + (uint64_t) blockIdx.x * (256 * PER); b0 < idHi; b0 += (uint64_t) gridDim.x * (256 * PER)) {
        if (threadIdx.x < 4) sCnt[threadIdx.x] = 0;
        __syncthreads();
        int cls[PER]; uint32_t rank[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t id = b0 + (uint64_t) j * 256 + threadIdx.x;
            cls[j] = -1; rank[j] = 0;
            if (id < idHi) {
                const uint32_t L = len[id], nw = L >= k ? L - k + 1 : 0u;
                cls[j] = nw > hugeWindows ? 2 : (nw > longWindows ? 1 : 0);
                if (origin && origin[id] != 0xFFFFFFFFu) cls[j] = 3;
                rank[j] = atomicAdd(&sCnt[cls[j]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 4) sBase[threadIdx.x] = sCnt[threadIdx.x] ? atomicAdd(threadIdx.x == 0 ? waveCount : (threadIdx.x == 1 ? longCount : (threadIdx.x == 2 ? hugeCount : cachedCount)), sCnt[threadIdx.x]) : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (cls[j] >= 0) (cls[j] == 0 ? waveList : (cls[j] == 1 ? longList : (cls[j] == 2 ? hugeList : cachedList)))[sBase[cls[j]] + rank[j]] = (uint32_t) (b0 + (uint64_t) j * 256 + threadIdx.x);
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
// STAGE (round 6): a lane's records go to LDS first — eight per lane, rows swizzled so that both the lanes' own 16-byte stores and the cooperative
// reads are conflict-free — and every eight residues the wavefront writes them out TOGETHER: eight lanes per source lane, i.e. up to 128 contiguous
// bytes per source instead of one 16-byte piece per lane and window into 64 different lines (the kernel was bound by those pieces: 44 GB of
// write traffic for 28 GB of records at 2.2-2.6 TB/s, with resident wavefronts capped so that the half-written lines stay in the L2).
template <bool LONG, int K, bool MUL24, bool STAGE = false>
__global__ __launch_bounds__(64) void extractShortFastKernel(ShortArgs a) {
    static_assert(!(STAGE && LONG), "the staging rows hold 16-byte records");
    constexpr int H = K / 2;
    static_assert(K <= 16 && 16 - K + H + 3 <= 15, "the digits that leave the halves are read from the 16-byte FIFO before this iteration's codes enter it");
    __shared__ unsigned char sMap[256];
    __shared__ __attribute__((aligned(16))) unsigned short sSet[64 * 64];    // per-lane open-addressing set of tags, as above
    typedef Rec<LONG> R;
    __shared__ __attribute__((aligned(16))) R sStage[STAGE ? 8 * 64 : 1];     // record k of lane l since the last flush: [k * 64 + ((l + 8 k) & 63)]
    __shared__ unsigned long long sDst[STAGE ? 64 : 1];
    __shared__ uint32_t sCnt[STAGE ? 64 : 1];
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
    // Round 6: the ids of a block's rounds are arithmetic, so the index words of the round after next (length, offset, slot range) and the first sixteen
    // residues of the next round's sequences are requested while the current round is worked on: a round used to START with three dependent round trips
    // (length -> offset and slots -> residues), and the counters showed the wavefronts parked 57 % of their cycles (profiles/r06_calls/call13_pmc_b.txt).
    struct Ahead { uint32_t L; uint64_t off, slot, slot1; };
    auto loadAhead = [&](uint64_t idw) { Ahead m; m.L = 0; m.off = 0; m.slot = 0; m.slot1 = 0; if (idw < (uint64_t) a.idHi) { m.L = a.s.len[idw]; m.off = a.s.off[idw]; m.slot = a.slotOff[idw]; m.slot1 = a.slotOff[idw + 1]; } return m; };
    auto isShort = [&](uint32_t Lx) { const uint32_t nw = (Lx >= (uint32_t) K) ? (Lx - K + 1) : 0; const size_t cr = (size_t) ((float) (a.kps - 1) + (a.scale * (float) Lx)); return !(Lx > SHORT_MAXL || (size_t) nw > cr || (multi && nw > 48)); };
    auto firstBytes = [&](const Ahead &m, uint64_t idw) { uint4 v = make_uint4(0, 0, 0, 0); if (idw < (uint64_t) a.idHi && m.L && isShort(m.L)) __builtin_memcpy(&v, a.s.data + m.off, 16); return v; };      // (the buffer is padded past its end)
    // (not with STAGE — iteration 0, every lane working, the kernel at its write bandwidth: measured 27.7 -> 32.7 ms with the look-ahead, call 23 — there the
    //  words of a round are requested together at its start, one round trip ahead of the residues)
    constexpr bool AHEAD = !STAGE;
    const uint64_t strideIds = (uint64_t) gridDim.x * 64;
    uint64_t bw = (uint64_t) a.idLo + (uint64_t) blockIdx.x * 64;
    Ahead mCur = {0, 0, 0, 0}, mNext = {0, 0, 0, 0};
    uint4 bytesCur = make_uint4(0, 0, 0, 0);
    if (AHEAD) { mCur = loadAhead(bw + lane); mNext = loadAhead(bw + strideIds + lane); bytesCur = firstBytes(mCur, bw + lane); }
    for (uint32_t b0 = a.idLo + blockIdx.x * 64; b0 < a.idHi; b0 += gridDim.x * 64, bw += strideIds) {
        const uint32_t id = b0 + lane;
        const bool active = id < a.idHi;
        Ahead cur; uint4 curBytes;
        if (AHEAD) {
            cur = mCur; curBytes = bytesCur;
            mCur = mNext; bytesCur = firstBytes(mCur, bw + strideIds + lane);
            mNext = loadAhead(bw + 2 * strideIds + lane);
        } else { cur = loadAhead(bw + lane); curBytes = firstBytes(cur, bw + lane); }
        bool toWave = false, lenWave = false;
        uint32_t L = 0;
        if (active) {
            L = cur.L;
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
                base += cur.off;
                slot = cur.slot - a.slotBias; bound = (uint32_t) (cur.slot1 - cur.slot);
                if (multi) { uint4 z = make_uint4(0, 0, 0, 0); uint4 *q = reinterpret_cast<uint4 *>(mySet); for (int i = 0; i < 8; i++) q[i] = z; }
            }
            const uint32_t Lw = work ? L : 0u;                 // a lane without work has no residue inside
            uint32_t lo = 0, hi = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0, nOut = 0;
            uint32_t nFlushed = 0;                             // STAGE: records of this lane already written out
            auto flush = [&]() {
                const uint32_t cnt = (work && !toWave) ? nOut - nFlushed : 0u;
                if (__ballot(cnt != 0u)) {
                    sDst[STAGE ? lane : 0] = slot + 1 + nFlushed; sCnt[STAGE ? lane : 0] = cnt;
                    __syncthreads();
#pragma unroll
                    for (int r8 = 0; r8 < 8; r8++) {
                        const int sl = r8 * 8 + (lane >> 3), j = lane & 7;
                        if ((uint32_t) j < sCnt[STAGE ? sl : 0]) arr[sDst[STAGE ? sl : 0] + (uint32_t) j] = sStage[STAGE ? (j * 64 + ((sl + 8 * j) & 63)) : 0];
                    }
                    __syncthreads();
                }
                nFlushed = nOut;
            };
            uint64_t seqHash = 0;
            int lastX = -1;
            // sixteen residues per load (round 4: four per load fetched every 128-byte line of residues four to five times — the lanes of
            // the resident wavefronts keep more lines open than the L1 and L2 hold — 23 GB read for 4 GB of residues); the NEXT sixteen
            // are requested before this round's are used: the load is the head of a chain of dependent round trips (residues -> letter
            // codes in LDS -> tag set in LDS -> store)
            uint4 bufNext = Lw ? curBytes : make_uint4(0, 0, 0, 0);                        // (requested a round ago)
            for (uint32_t i0 = 0; i0 < Lmax; i0 += 16) {
                const uint4 buf = bufNext;
                bufNext = make_uint4(0, 0, 0, 0);
                if (i0 + 16 < Lw) __builtin_memcpy(&bufNext, base + i0 + 16, 16);
#pragma unroll
              for (int step = 0; step < 4; step++) {
                const uint32_t i = i0 + 4u * (uint32_t) step;
                if (i >= Lmax) break;                                                        // wave-uniform
                const uint32_t word = step == 0 ? buf.x : (step == 1 ? buf.y : (step == 2 ? buf.z : buf.w));
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
                        if constexpr (STAGE) { const uint32_t pi = nOut - nFlushed; sStage[pi * 64u + (((uint32_t) lane + 8u * pi) & 63u)] = r; }
                        else arr[slot + 1 + nOut] = r;
                        nOut++;
                    }
                }
                f0 = f1; f1 = f2; f2 = f3; f3 = cw;
                if constexpr (STAGE) { if (step & 1) flush(); }      // eight residues: at most eight records per lane
              }
            }
            if constexpr (STAGE) flush();
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

// =====================================================================================================
// 2e. FOUR SEQUENCES PER WAVEFRONT (round 6).  The wave-per-sequence tiers pay their control — staging, the
//     identity hash, the 16 steps of the threshold bisection, the candidate list, the repeat check, the stores — once per
//     wavefront and sequence, and half of what they issue is scalar bookkeeping for ONE sequence (SQ_INSTS_SALU ~ SQ_INSTS_VALU,
//     profiles/r05_pmc_lanes_per_kernel.txt).  Here a ROW of 16 lanes owns a sequence of up to 16 * REGS windows: window p is
//     lane p % 16's register p / 16, every count is a per-lane count plus a row reduction on the DPP path (no ballots, no
//     scalar loop per sequence), and one instruction stream drives four sequences.  Only the common outcome is finished here —
//     the threshold bin holds no surplus and no selected k-mer repeats: the reference then selects exactly the windows at or
//     below the threshold, in an order that does not matter (see the fast path of extractKernel) — everything else (surplus in
//     the threshold bin, a possible repeat, a sequence too long for the instantiation) is queued for the 4-scores tier, which
//     then owns the slot range.  Lists by window count (binWaveListKernel) keep the four sequences of a wavefront alike.
//     Protein runs with k <= 14, alphabet base <= 16, no length scaling, --kmer-per-seq <= 60 (the Plass workflow);
//     PLASSHIP_TUNE_ROWTIER=2 switches it off.
// =====================================================================================================
struct RowArgs {
    SeqView s; const uint64_t *slotOff; void *arr; const unsigned char *map;
    const uint32_t *list, *count;             // ids of this launch (count on the device)
    uint32_t *fallList, *fallCount;           // what this kernel leaves to the 4-scores tier
    int k, xCode, kps, ignoreMulti; uint64_t seed; uint32_t base, base7; uint64_t slotBias;
    unsigned long long *kstats;               // [2] residues, [3] records (the wave tiers' counters); [4] selected-window cache lines
};
constexpr uint64_t rowPowU64(uint64_t b, unsigned e) { uint64_t r = 1; while (e) { if (e & 1u) r *= b; b *= b; e >>= 1; } return r; }      // modulo 2^64
constexpr uint64_t rowInv31() { uint64_t x = 31; for (int i = 0; i < 6; i++) x *= 2ull - 31ull * x; return x; }      // Newton: 31 * 31 = 1 modulo 8, every step doubles the correct bits
static_assert(rowInv31() * 31ull == 1ull, "inverse of 31 modulo 2^64");
__device__ __forceinline__ uint32_t rowExclusiveScan16(uint32_t v) {      // exclusive prefix sum inside each row of 16 lanes
    uint32_t x = v;
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xF, 0xF, true);      // row_shr:1, lanes without a source add 0
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xF, 0xF, true);
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xF, 0xF, true);
    x += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xF, 0xF, true);
    return x - v;
}
__device__ __forceinline__ unsigned long long rowSum16U64(unsigned long long v) {
    v += dppMov64<0xB1>(v); v += dppMov64<0x4E>(v); v += dppMov64<0x141>(v); v += dppMov64<0x140>(v);
    return v;
}
// the ids of a list sorted into four lists by window count (<= w0, <= w1, <= w2, the rest): the rows of a wavefront share their loop bounds
__global__ __launch_bounds__(256) void binWaveListKernel(const uint32_t *__restrict__ list, const uint32_t *__restrict__ count, const uint32_t *__restrict__ len, uint32_t k,
                                                         uint32_t w0, uint32_t w1, uint32_t w2, uint32_t *__restrict__ out, uint32_t outStride, uint32_t *__restrict__ outCount) {
    __shared__ uint32_t sCnt[4], sBase[4];
    const uint32_t n = *count;
    constexpr int PER = 8;
    for (uint64_t b0 = (uint64_t) blockIdx.x * (256 * PER); b0 < n; b0 += (uint64_t) gridDim.x * (256 * PER)) {
        if (threadIdx.x < 4) sCnt[threadIdx.x] = 0;
        __syncthreads();
        int cls[PER]; uint32_t rank[PER], ids[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint64_t i = b0 + (uint64_t) j * 256 + threadIdx.x;
            cls[j] = -1; rank[j] = 0; ids[j] = 0;
            if (i < n) {
                ids[j] = list[i];
                const uint32_t L = len[ids[j]], nw = L >= k ? L - k + 1 : 0u;
                cls[j] = nw <= w0 ? 0 : (nw <= w1 ? 1 : (nw <= w2 ? 2 : 3));
                rank[j] = atomicAdd(&sCnt[cls[j]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 4) sBase[threadIdx.x] = sCnt[threadIdx.x] ? atomicAdd(&outCount[threadIdx.x], sCnt[threadIdx.x]) : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++) if (cls[j] >= 0) out[(size_t) cls[j] * outStride + sBase[cls[j]] + rank[j]] = ids[j];
        __syncthreads();
    }
}

template <int REGS, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void extractRowKernel(RowArgs a) {
    constexpr uint32_t MAXWIN = 16u * REGS, MAXL = MAXWIN + 13u;           // k <= 14
    constexpr uint32_t ROWB = (MAXL + 31u + 15u) & ~15u;                    // staged codes of a row, X padding included
    constexpr bool TWO = MAXL > 256u;                                       // a second 16-byte chunk per lane
    constexpr uint32_t NPOW = MAXL + 17u;
    __shared__ unsigned char sMap[256];
    __shared__ __attribute__((aligned(16))) unsigned char sCode[4][ROWB];
    __shared__ unsigned long long sPow[NPOW];                               // 31^(x - 15) modulo 2^64 (31 is odd: the inverse exists)
    __shared__ unsigned short sSel[4][64];                                  // window positions of the selection, row by row
    __shared__ __attribute__((aligned(16))) uint32_t sSet[4][128];          // repeat check: 32-bit tags of the selected k-mers
    typedef Rec<false> R;
    R *arr = reinterpret_cast<R *>(a.arr);
    const int lane = threadIdx.x, row = lane >> 4, sub = lane & 15;
    const int k = a.k;
    if (blockIdx.x * 4u >= *a.count) return;                               // (a list shorter than the grid: nothing to set up)
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    {
        // 31^(lane - 15) by the bits of the lane (compile-time squares), then steps of 31^64: a block's set-up is a dozen multiplications
        // (a first version walked 31^lane up one factor at a time — 63 dependent 64-bit products per block, and the launch preferred small grids)
        constexpr uint64_t inv31 = rowInv31(), c0 = rowPowU64(inv31, 15), p64 = rowPowU64(31ull, 64);
        uint64_t m = c0;
#pragma unroll
        for (int b = 0; b < 6; b++) if ((lane >> b) & 1) m *= rowPowU64(31ull, 1u << b);
        for (uint32_t x = lane; x < NPOW; x += 64) { sPow[x] = m; m *= p64; }
    }
    __syncthreads();
    const uint32_t nWork = *a.count;
    const uint32_t stride = gridDim.x * 4u;
    const uint32_t xC = (uint32_t) a.xCode;
    const uint32_t considerRaw = (uint32_t) (a.kps - 1);                    // kmermatcher.cpp:223 with --kmer-per-seq-scale 0
    unsigned long long stRes = 0, stRec = 0;
    struct Meta { uint32_t id, L; uint64_t off, slot, slot1; };
    auto loadMeta = [&](uint32_t w) {
        Meta m; m.id = 0xFFFFFFFFu; m.L = 0; m.off = 0; m.slot = 0; m.slot1 = 0;
        if (w < nWork) { m.id = a.list[w]; m.L = a.s.len[m.id]; m.off = a.s.off[m.id]; m.slot = a.slotOff[m.id] - a.slotBias; m.slot1 = a.slotOff[m.id + 1] - a.slotBias; }
        return m;
    };
    auto loadRaw = [&](const Meta &m, uint32_t at) { uint4 v = make_uint4(0, 0, 0, 0); if (at < m.L) __builtin_memcpy(&v, a.s.data + m.off + at, 16); return v; };      // (buffers are padded past their ends)
    uint32_t w = blockIdx.x * 4u + (uint32_t) row;
    auto loadId = [&](uint32_t ww) { return ww < nWork ? a.list[ww] : 0xFFFFFFFFu; };
    auto metaOf = [&](uint32_t id) {
        Meta m; m.id = id; m.L = 0; m.off = 0; m.slot = 0; m.slot1 = 0;
        if (id != 0xFFFFFFFFu) { m.L = a.s.len[id]; m.off = a.s.off[id]; m.slot = a.slotOff[id] - a.slotBias; m.slot1 = a.slotOff[id + 1] - a.slotBias; }
        return m;
    };
    Meta mNext = loadMeta(w); uint32_t id2 = loadId(w + stride);
    uint4 rawN0 = loadRaw(mNext, (uint32_t) sub * 16u);
    for (uint32_t w0 = blockIdx.x * 4u; w0 < nWork; w0 += stride, w += stride) {
        const Meta cur = mNext; const uint4 raw0 = rawN0;
        uint4 raw1 = make_uint4(0, 0, 0, 0);
        if (TWO) raw1 = loadRaw(cur, 256u + (uint32_t) sub * 16u);         // (the last 13 residues of the longest sequences: requested here, used behind the first chunk)
        mNext = metaOf(id2);
        rawN0 = loadRaw(mNext, (uint32_t) sub * 16u);
        id2 = loadId(w + 2u * stride);
        const bool active = cur.id != 0xFFFFFFFFu;
        const uint32_t L = cur.L, id = cur.id;
        const uint32_t nWin = (L >= (uint32_t) k) ? L - (uint32_t) k + 1u : 0u;
        const bool tooLong = active && nWin > MAXWIN;                       // (the lists are binned: does not happen)
        const uint32_t nWinU = (uint32_t) __builtin_amdgcn_readfirstlane((int) waveMaxU32((active && !tooLong) ? nWin : 0u));
        // ---- letter codes of the row's sequence into LDS (16 per lane and chunk), the chunk's part of the identity hash on the way ----
        unsigned long long hAcc = 0;
        auto stageChunk = [&](const uint4 &raw, uint32_t c) {
            const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
            uint32_t cw[4];
#pragma unroll
            for (int d = 0; d < 4; d++)
                cw[d] = (uint32_t) sMap[r[d] & 0xFFu] | ((uint32_t) sMap[(r[d] >> 8) & 0xFFu] << 8) | ((uint32_t) sMap[(r[d] >> 16) & 0xFFu] << 16) | ((uint32_t) sMap[r[d] >> 24] << 24);
            if (16u * c < ROWB) *reinterpret_cast<uint4 *>(&sCode[row][16u * c]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            // identity hash (Util::hash, Util.h:337-345: sum code[p] * 31^(L-1-p)): the chunk's 16 codes as four base-31 quads, the quads
            // joined by 31^4; codes at and behind L count as 0, so a partial chunk is its sum times 31^(16-m) — undone by the NEGATIVE
            // exponent of the chunk's power 31^(L - 16 (c + 1))
            const int m = (int) L - (int) (16u * c);
            if (m > 0 && !tooLong) {
                uint32_t q[4];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int nb = m - 4 * d;
                    const uint32_t z = nb >= 4 ? cw[d] : (nb <= 0 ? 0u : (cw[d] & ((1u << (8 * nb)) - 1u)));
                    q[d] = __umul24(__umul24(__umul24(z & 0xFFu, 31u) + ((z >> 8) & 0xFFu), 31u) + ((z >> 16) & 0xFFu), 31u) + (z >> 24);
                }
                constexpr unsigned long long K4 = 923521ull;                 // 31^4
                const unsigned long long H = (((unsigned long long) q[0] * K4 + q[1]) * K4 + q[2]) * K4 + q[3];
                hAcc += H * sPow[(uint32_t) ((int) L - 16 * (int) (c + 1) + 15)];
            }
        };
        stageChunk(raw0, (uint32_t) sub);
        if (TWO) stageChunk(raw1, 16u + (uint32_t) sub);
        if (active && !tooLong) for (uint32_t i = (uint32_t) sub; i < 31u; i += 16u) sCode[row][L + i] = (unsigned char) xC;      // X behind the sequence: every window read stays defined
        for (uint32_t i = (uint32_t) sub * 4u; i < 128u; i += 64u) *reinterpret_cast<uint4 *>(&sSet[row][i]) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const unsigned long long seqHash = rowSum16U64(hAcc);
        // ---- one score per window, in registers ----
        uint32_t sc[REGS];
        uint32_t nLane = 0;
#pragma unroll
        for (int j = 0; j < REGS; j++) {
            sc[j] = 0xFFFFFFFFu;
            if ((uint32_t) j * 16u < nWinU) {
                // (every lane hashes — a window behind the row's last one reads stale codes inside the row's buffer and is discarded: the rows of a
                //  wavefront differ in length, and a branch per row and window costs more than the masked lanes it saves)
                const uint32_t p = (uint32_t) j * 16u + (uint32_t) sub;
                uint64_t kmer;
                const bool v = kmerIndexFastAligned(sCode[row], p, k, xC, a.base, a.base7, kmer);
                const uint32_t h = xxh64Score16(kmer, a.seed);
                const bool use = v && p < nWin && !tooLong;
                sc[j] = use ? h : 0xFFFFFFFFu; nLane += use ? 1u : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);      // one window's hash at a time: the temporaries of interleaved windows cost a wavefront per SIMD
        }
        const uint32_t n = (uint32_t) rowSum16((int) nLane);
        const uint32_t considered = min(considerRaw, n);
        // ---- the reference's walk over 65 536 score bins (kmermatcher.cpp:224-239) as a bisection: every count a per-lane count + a row sum ----
        uint32_t t = 0;
#pragma unroll 1
        for (int bit = 15; bit >= 0; bit--) {
            const uint32_t tr = t | (1u << bit);
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < REGS; j++) if ((uint32_t) j * 16u < nWinU) c += (sc[j] < tr) ? 1u : 0u;
            if ((uint32_t) rowSum16((int) c) < considered) t = tr;
        }
        if (n <= considered) t = 0xFFFFu;                                   // every valid window is selected
        uint32_t selMask = 0;
#pragma unroll
        for (int j = 0; j < REGS; j++) if ((uint32_t) j * 16u < nWinU) selMask |= (sc[j] <= t) ? (1u << j) : 0u;
        const uint32_t cLane = (uint32_t) __popc(selMask);
        const uint32_t C = (uint32_t) rowSum16((int) cLane);
        bool ok = active && !tooLong && C == considered;                    // no surplus in the threshold bin
        // ---- the selected windows' positions, row by row (any order: see the fast path of extractKernel) ----
        {
            uint32_t at = rowExclusiveScan16(cLane);
            if (ok) {
#pragma unroll
                for (int j = 0; j < REGS; j++) if ((selMask >> j) & 1u) { sSel[row][at] = (unsigned short) ((uint32_t) j * 16u + (uint32_t) sub); at++; }
            }
        }
        __syncthreads();
        const uint64_t slot = cur.slot;
        const uint32_t bound = (uint32_t) (cur.slot1 - cur.slot);
        bool rep = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = (uint32_t) sub + 16u * (uint32_t) i;
            if (ok && r < C) {
                const uint32_t p = sSel[row][r];
                uint64_t kmer; (void) kmerIndexFastAligned(sCode[row], p, k, xC, a.base, a.base7, kmer);
                if (a.ignoreMulti) {      // equal k-mers have equal tags: a tag met twice sends the sequence to the tier that compares k-mers
                    const uint64_t hx = kmer * 0x9E3779B97F4A7C15ULL;
                    const uint32_t tag = (uint32_t) (hx >> 32) | 1u;
                    uint32_t sl = (uint32_t) (hx >> 20) & 127u;
                    for (;;) {
                        const uint32_t prev = atomicCAS(&sSet[row][sl], 0u, tag);
                        if (prev == 0u) break;
                        if (prev == tag) { rep = true; break; }
                        sl = (sl + 1u) & 127u;
                    }
                }
                R rec; rec.kmer = kmer; rec.id = id; rec.len = (uint16_t) L; rec.pos = (int16_t) p;
                arr[slot + 1 + r] = rec;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rowSum16(rep ? 1 : 0) != 0) ok = false;
        if (ok) {
            if (sub == 15) { R rec; rec.kmer = xxh64U64(seqHash, a.seed); rec.id = id; rec.len = (uint16_t) L; rec.pos = 0; arr[slot] = rec; }      // identity record (kmermatcher.cpp:241-249)
            for (uint32_t i = 1u + C + (uint32_t) sub; i < bound; i += 16u) { R rec; memset(&rec, 0xFF, sizeof(R)); arr[slot + i] = rec; }
            unsigned char *cl = reinterpret_cast<unsigned char *>(a.kstats[4]);
            if (cl) {       // selected-window cache (section 2c): the same line the wave tiers' fast path leaves
                unsigned short *ln = reinterpret_cast<unsigned short *>(cl + (size_t) id * KMC_LINE);
#pragma unroll
                for (int i = 0; i < 4; i++) { const uint32_t r = (uint32_t) sub + 16u * (uint32_t) i; if (r < KMC_POS) ln[4 + r] = (r < C) ? sSel[row][r] : (unsigned short) 0xFFFFu; }
                if (sub == 0) *reinterpret_cast<unsigned long long *>(ln) = seqHash;
                if (sub == 15) ln[KMC_FLAGS] = (unsigned short) KMC_CLEAN;
            }
            if (sub == 0) { stRes += L; stRec += 1 + C; }
        }
        // ---- what is left to the 4-scores tier: one atomic per wavefront ----
        {
            const bool fall = active && !ok && sub == 0;
            const unsigned long long fm = __ballot(fall);
            if (fm) {
                uint32_t basePos = 0;
                if (lane == 0) basePos = atomicAdd(a.fallCount, (uint32_t) __popcll(fm));
                basePos = __shfl(basePos, 0, 64);
                if (fall) a.fallList[basePos + (uint32_t) __popcll(fm & ((1ULL << lane) - 1ULL))] = id;
            }
        }
        __syncthreads();
    }
    stRes = waveReduceSumU64(stRes); stRec = waveReduceSumU64(stRec);
    if (lane == 0 && a.kstats) { atomicAdd(&a.kstats[2], stRes); atomicAdd(&a.kstats[3], stRec); }
}

// =====================================================================================================
// 2d. the position cache of nucleotide runs (round 6).  PenguiN's nucleotide chains never change the hash seed (Nuclassembler.cpp:24,
//     GuidedNuclassembler.cpp:25: --hash-shift stays what it is), and an iteration rewrites a minority of the sequences, so from the
//     second call on most sequences are byte for byte what they were when kmermatcher last selected their windows.  The selection is
//     variable-length there (--kmer-per-seq-scale 0.1: 59 + 0.1 L windows) and nuclassembleresults / cyclecheck DROP entries, so the
//     128-byte lines of section 2c do not fit; instead the extraction kernels leave, beside the record array, ONE WINDOW POSITION PER
//     RECORD SLOT (u16, u32 in the long layout; the slot of the identity record holds the count) and the identity hash per id, and the
//     DBs derived from the DB of that call carry, per entry, the id it had there (plasship_seqdb::d_origin, 0xFFFFFFFF = rewritten —
//     buildOutputDB, assemble.hip).  The next call with the same selection parameters rebuilds the records of every such entry here:
//     canonical k-mer and strand from the bytes at the cached positions, nothing hashed, nothing selected.  Same records in the same
//     slots as the full kernels write (the nucleotide chain tests pass through here from their third call on; PLASSHIP_TUNE_KMCACHE=2
//     switches it off).
// =====================================================================================================
template <bool LONG>
struct CachedPosArgs {
    SeqView s; const uint64_t *slotOff; void *arr; const unsigned char *map;
    const void *posOld; const uint64_t *slotOffOld; const unsigned long long *idHashOld; const uint32_t *origin;
    void *posNew; unsigned long long *idHashNew;
    const uint32_t *list, *count; int k; uint64_t seed; unsigned long long *kstats;
};
template <bool LONG>
__global__ __launch_bounds__(64) void extractCachedPosKernel(CachedPosArgs<LONG> a) {
    __shared__ unsigned char sMap[256];
    typedef Rec<LONG> R;
    typedef typename std::conditional<LONG, uint32_t, unsigned short>::type PosT;
    R *arr = reinterpret_cast<R *>(a.arr);
    const PosT *posOld = reinterpret_cast<const PosT *>(a.posOld);
    PosT *posNew = reinterpret_cast<PosT *>(a.posNew);
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    __syncthreads();
    const int k = a.k;
    const uint32_t nWork = *a.count;
    unsigned long long stRes = 0, stRec = 0;
    // the chain of a sequence: list entry -> origin, index entry, slot range -> old slot range -> count, positions -> bytes -> records.
    // Metadata of sequence w + grid is in flight while sequence w is written.
    struct Meta { uint32_t id, L, cnt; uint64_t off, slot, slot1, so; unsigned long long idh; };
    auto loadMeta = [&](uint32_t id) {
        Meta m; m.id = id; m.L = a.s.len[id]; m.off = a.s.off[id]; m.slot = a.slotOff[id]; m.slot1 = a.slotOff[id + 1];
        const uint32_t src = a.origin[id];
        m.so = a.slotOffOld[src]; m.idh = a.idHashOld[src]; m.cnt = (uint32_t) posOld[m.so];
        return m;
    };
    Meta nxt; nxt.id = 0; nxt.L = 0; nxt.cnt = 0; nxt.off = 0; nxt.slot = 0; nxt.slot1 = 0; nxt.so = 0; nxt.idh = 0;
    if (blockIdx.x < nWork) nxt = loadMeta(a.list[blockIdx.x]);
    for (uint32_t w = blockIdx.x; w < nWork; w += gridDim.x) {
        const Meta cur = nxt;
        if (w + gridDim.x < nWork) nxt = loadMeta(a.list[w + gridDim.x]);
        const char *base = a.s.data + cur.off;
        const uint32_t bound = (uint32_t) (cur.slot1 - cur.slot), L = cur.L;
        for (uint32_t i = lane; i < cur.cnt; i += 64) {
            const uint32_t p = (uint32_t) posOld[cur.so + 1 + i];
            // the k <= 31 letters of the window as codes (0..3, X = 4), eight per word; the entry is "SEQ\n\0" in a padded buffer
            uint64_t x = 0, f = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (8 * j < k) {
                    uint64_t raw; __builtin_memcpy(&raw, base + p + 8 * j, 8);
                    uint64_t v = 0;
#pragma unroll
                    for (int b = 0; b < 8; b++) v |= (uint64_t) sMap[(raw >> (8 * b)) & 0xFF] << (8 * b);
                    const int nb = k - 8 * j;
                    if (nb < 8) v &= (1ULL << (8 * nb)) - 1ULL;
                    x |= v;
                    const int sh = 2 * (k - 8 * (j + 1));
                    const uint64_t g = pack2x8(v);
                    f |= (sh >= 0) ? (g << sh) : (g >> (-sh));
                }
            }
            (void) x;                                                  // (a cached window held no X and was no palindrome when it was selected)
            const uint64_t rc = revComplementDev(f, k);
            const bool pickRev = rc < f;
            R r; r.kmer = pickRev ? rc : (f | BIT63); r.id = cur.id; r.len = (decltype(r.len)) L; r.pos = (decltype(r.pos)) (pickRev ? (L - p - (uint32_t) k) : p);
            if constexpr (LONG) r.pad = 0;
            arr[cur.slot + 1 + i] = r;
            posNew[cur.slot + 1 + i] = (PosT) p;
        }
        if (lane == 63) {
            R r; r.kmer = xxh64U64(cur.idh, a.seed); r.id = cur.id; r.len = (decltype(r.len)) L; r.pos = 0;
            if constexpr (LONG) r.pad = 0;
            arr[cur.slot] = r;
            posNew[cur.slot] = (PosT) cur.cnt; a.idHashNew[cur.id] = cur.idh;
        }
        for (uint32_t i = 1 + cur.cnt + (uint32_t) lane; i < bound; i += 64) { R r; memset(&r, 0xFF, sizeof(R)); arr[cur.slot + i] = r; }
        stRes += L; stRec += 1 + cur.cnt;
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

