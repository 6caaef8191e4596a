// plasship: k-mer record layouts and the LINE-STORE PARTITION that replaces both sorts of kmermatcher (product code).
//
//   reference: sort #1 by (kmer, seqLen desc, id, pos) and sort #2 by (rep, target, diagonal), two ips4o sorts of 16-byte
//   records (mm/linclust/kmermatcher.cpp:408-412, 427-431).  Only the GROUPING by k-mer (resp. by representative range)
//   matters to what follows, so the GPU path partitions instead of sorting (see kmermatch.hip).
//
// Why a new partition (round 2): the round-1 scatter moved 3.7x the algorithmic bytes — a histogram pass before every scatter
// pass, and scattered 16-byte stores that leave partial cache lines (WRITE_SIZE 1.6x the record bytes).  Here
//   * records travel in LINES of 8 (128 bytes for 16-byte records): a workgroup keeps one open line per bucket in LDS (write
//     combining, up to 1024 buckets x 128 B = 128 KB of the CU's 160 KB) and appends every completed line to ITS OWN sequential
//     output range, tagged with the bucket — all global stores are full, aligned lines written as one stream per workgroup;
//   * no histogram pass and no global atomics: a piece of input (a fixed number of lines) owns a fixed output range
//     (input lines + one partial line per bucket), so nothing has to be counted before it is written and the layout is
//     deterministic in size whatever the skew of the keys;
//   * the next stage finds the lines of a bucket through a line LIST: a counting sort of the 4-byte tags (3 % of the record
//     bytes), not of the records.  It reads whole 128-byte lines wherever they lie.
// Traffic per level: one read and one write of the records (+ 3 % tags and lists) — the one-pass ideal of SURVEY.md section 8d
// per level.  A level handles up to 1024 buckets (512 for 24-byte records), two levels give the ~10^6 buckets of a 50 M-read set.
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace plasship {

#define BIT63 (1ULL << 63)

// ---- record layouts (kmermatcher.h:49-55: KmerPosition<short> 16 B, KmerPosition<int> 20 B) -------------
template <bool LONG> struct Rec;
template <> struct __attribute__((aligned(16))) Rec<false> { uint64_t kmer; uint32_t id; uint16_t len; int16_t pos; };
template <> struct __attribute__((aligned(8))) Rec<true> { uint64_t kmer; uint32_t id; int32_t len; int32_t pos; uint32_t pad; };

template <bool LONG> __device__ __forceinline__ bool isSentinel(const Rec<LONG> &r) { return r.kmer == ~0ULL && r.id == 0xFFFFFFFFu; }

enum { KEY_HASH = 0, KEY_RANGE = 1 };
template <bool NUCL> __device__ __forceinline__ uint64_t kmerMix(uint64_t kmerField) {
    const uint64_t K = NUCL ? (kmerField & ~BIT63) : kmerField;
    uint64_t x = K * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    return x;
}

// histogram of the records by k-mer VALUE (monotone bins of the sort-#1 key): bounds the sort-#1 rank of any record without
// sorting, which is all the stale-record check (kmermatch.hip section 7) needs most of the time
constexpr uint32_t VH_BINS = 2048;
template <bool NUCL> __device__ __host__ __forceinline__ uint32_t valueBin(uint64_t kmerField, int shift) {
    const uint64_t v = NUCL ? (kmerField & ~BIT63) : kmerField;       // sort #1 compares (kmer | bit 63) for nucleotides
    const uint64_t b = v >> shift;
    return b < VH_BINS ? (uint32_t) b : VH_BINS - 1;                   // identity records (64-bit hashes) collect in the last bin
}

// =====================================================================================================
// line store
// =====================================================================================================
#ifndef PLASSHIP_RPL
#define PLASSHIP_RPL 8
#endif
constexpr int RPL = PLASSHIP_RPL;            // records per line
constexpr uint32_t TAG_NONE = 0xFFFFFFFFu;   // tag of an output line nobody wrote
// workgroup geometry: BLOCK threads take ITEMS records each per tile; with PREFETCH the next tile's records are in flight
// while the current tile goes through LDS (one workgroup per CU at 1024 buckets: nothing else would hide the HBM latency)
constexpr uint32_t LP_MAXB = 1024;           // buckets per level (LDS: nb * RPL * sizeof(record) + 10 * nb bytes)

struct LineKey {
    int shift, rangeBits;        // KEY_HASH: bucket = (mix >> shift) & (nb - 1);  KEY_RANGE: ((rep - repBase) << (64 - rangeBits)) >> shift
    uint64_t repBase;
    int scrambleBits;            // KEY_RANGE, != 0: the ranges are taken on the bit-REVERSED rep id (scrambleRep)
};
// The representative of a k-mer group is its longest sequence and, among equally long ones, the one with the SMALLEST id
// (kmermatcher.cpp:450-559).  In iteration 0 all fragments are ~50 residues, so representatives pile up at the low end of the id
// range (the minimum of ~25 uniform ids) and equal-width id ranges would put half of the grouped records into 3 % of the buckets.
// The rep sort therefore ranges over the bit-reversed id — a bijection on [0, 2^bits) whose top bits are the id's evenly spread
// low bits; every representative's triples still end up contiguous and sorted, and are moved to the representative's place in id
// order afterwards (placeRunsKernel, kmermatch.hip).
__device__ __host__ __forceinline__ uint64_t scrambleRep(uint64_t rep, int bits) {
    uint32_t x = (uint32_t) rep;
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
    return (uint64_t) (x >> (32 - bits));      // an involution on [0, 2^bits): applying it again gives the id back
}
template <bool NUCL, int MODE> __device__ __forceinline__ uint32_t lineBucket(const LineKey &k, uint64_t kmerField, uint32_t nb) {
    if (MODE == KEY_HASH) return (uint32_t) (kmerMix<NUCL>(kmerField) >> k.shift) & (nb - 1);
    uint64_t r = (kmerField & ~BIT63) - k.repBase;
    if (k.scrambleBits) r = scrambleRep(r, k.scrambleBits);
    return (uint32_t) ((r << (64 - k.rangeBits)) >> k.shift) & (nb - 1);   // left-aligned rep id: top bits = id range
}

// a piece of input and the output range it owns
struct LinePiece {
    uint64_t in0;         // first input line: index into the line list (LIST) or line number in the input array (dense)
    uint64_t out0;        // first output line
    uint32_t nLines;      // input lines (>= 1)
    uint32_t lastValid;   // records of the last input line that exist (1..RPL); dense inputs whose length is no multiple of RPL
    uint32_t outCap;      // output lines reserved: nLines + nb
    uint32_t tagBase;     // added to the bucket to form the tag (level 2: level-1 bucket * nb)
};

struct LinePartArgs {
    const void *in; const uint32_t *list;       // list != nullptr: the input lines are in[list[in0 + j]]
    void *out; uint32_t *tags;
    const LinePiece *pieces; const uint32_t *nPieces;   // piece table on the device, or (pieces == nullptr) uniform pieces:
    uint64_t totalLines; uint32_t lastValidAll, pieceLines;   //   piece p = input lines [p * pieceLines, …), output lines [p * (pieceLines + nb), …)
    uint32_t *pieceOut;                          // optional: output lines a piece used
    LineKey key; uint32_t nb;
    int direct;                                  // records of lines that complete inside a tile go straight to HBM (see linePartKernel)
    unsigned long long *minKey;                  // optional (EXTRAS, NUCL): global minimum of (kmer | BIT63) (first-run quirk)
    uint32_t *valueHist; int valueShift;         // optional (EXTRAS)
    uint32_t keepLo, keepHi;                     // (EXTRAS, keepHi != 0) owner-filtered extraction of a sharded run: records of the buckets outside
                                                 // [keepLo, keepHi) are dropped here — another rank, which extracted the same sequences, keeps them
};

static inline size_t linePartLdsBytes(uint32_t nb, size_t recBytes, bool extras) {
    return (size_t) nb * RPL * recBytes + (size_t) nb * 8 + (extras ? VH_BINS * 4 : 0) + (size_t) nb * 2 + (size_t) nb * 4 + 16;
}

template <bool NUCL, bool LONG, int MODE, bool LIST, bool EXTRAS, int LP_BLOCK, int LP_ITEMS, bool PREFETCH>
__global__ __launch_bounds__(LP_BLOCK) void linePartKernel(LinePartArgs a) {
    constexpr int LP_TILE = LP_BLOCK * LP_ITEMS;
    typedef Rec<LONG> R;
    extern __shared__ __attribute__((aligned(16))) unsigned char lpDyn[];
    const uint32_t nb = a.nb;
    R *buf = reinterpret_cast<R *>(lpDyn);                                                    // [nb][RPL] the open line of every bucket
    uint32_t *cnt = reinterpret_cast<uint32_t *>(lpDyn + (size_t) nb * RPL * sizeof(R));      // [nb] records seen of the bucket (this piece)
    uint32_t *flushed = cnt + nb;                                                             // [nb] lines written of the bucket (this piece)
    uint32_t *vh = flushed + nb;                                                              // [VH_BINS] (EXTRAS)
    uint32_t *lbase = vh + (EXTRAS ? VH_BINS : 0);                                            // [nb] (direct) first output line of the bucket's lines of this tile
    unsigned short *queue = reinterpret_cast<unsigned short *>(lbase + nb);                   // [nb] buckets whose line completed this round
    __shared__ uint32_t sQ[2];
    __shared__ uint32_t sScan[LP_BLOCK / 64 + 1];
    const uint32_t tid = threadIdx.x;
    const R *in = reinterpret_cast<const R *>(a.in);
    R *out = reinterpret_cast<R *>(a.out);
    if (tid < 2) sQ[tid] = 0;
    if (EXTRAS && a.valueHist) for (uint32_t i = tid; i < VH_BINS; i += LP_BLOCK) vh[i] = 0;
    unsigned long long mn = ~0ULL;
    uint32_t round = 0;
    const uint32_t nPieces = a.pieces ? *a.nPieces : (uint32_t) ((a.totalLines + a.pieceLines - 1) / a.pieceLines);
    R sen; __builtin_memset(&sen, 0xFF, sizeof(R));
    for (uint32_t piece = blockIdx.x; piece < nPieces; piece += gridDim.x) {
        LinePiece pc;
        if (a.pieces) pc = a.pieces[piece];
        else {
            pc.in0 = (uint64_t) piece * a.pieceLines;
            pc.nLines = (uint32_t) (a.totalLines - pc.in0 < (uint64_t) a.pieceLines ? a.totalLines - pc.in0 : (uint64_t) a.pieceLines);
            pc.lastValid = (pc.in0 + pc.nLines == a.totalLines) ? a.lastValidAll : (uint32_t) RPL;
            pc.out0 = (uint64_t) piece * ((uint64_t) a.pieceLines + nb); pc.outCap = a.pieceLines + nb; pc.tagBase = 0;
        }
        for (uint32_t i = tid; i < nb; i += LP_BLOCK) { cnt[i] = 0; flushed[i] = 0; }
        __syncthreads();
        uint64_t outLine = pc.out0;                                  // workgroup-uniform
        const uint64_t nRec = (uint64_t) (pc.nLines - 1) * RPL + pc.lastValid;
        R nxt[LP_ITEMS];
        // LIST input: a record is two dependent round trips away (list entry, then the record).  Round 5: the list entries run ONE TILE AHEAD
        // of the records (lnNext = the lines of tile t + 2 while the records of tile t + 1 are in flight) — with both inside one prefetch the
        // chain had a single tile's time (~4 us per CU at 4 TB/s) to complete, and level 2 took 29 ms where level 1 (dense input) takes 24.
        uint32_t lnNext[LP_ITEMS];
        auto loadLines = [&](uint64_t t0) {
#pragma unroll
            for (int u = 0; u < LP_ITEMS; u++) {
                const uint64_t i = t0 + (uint64_t) u * LP_BLOCK + tid;
                lnNext[u] = (LIST && i < nRec) ? a.list[pc.in0 + i / RPL] : 0u;
            }
        };
        auto loadTile = [&](uint64_t t0) {
#pragma unroll
            for (int u = 0; u < LP_ITEMS; u++) {
                const uint64_t i = t0 + (uint64_t) u * LP_BLOCK + tid;
                if (i < nRec) {
                    const uint64_t line = LIST ? (uint64_t) lnNext[u] : pc.in0 + i / RPL;
                    nxt[u] = in[line * RPL + (i % RPL)];
                } else nxt[u] = sen;
            }
        };
        if (LIST) loadLines(0);
        if (PREFETCH) { loadTile(0); if (LIST) loadLines(LP_TILE); }
        for (uint64_t t0 = 0; t0 < nRec; t0 += LP_TILE) {
            R rec[LP_ITEMS]; uint32_t bk[LP_ITEMS], sq[LP_ITEMS]; uint32_t pending = 0;
            if (!PREFETCH) { loadTile(t0); if (LIST) loadLines(t0 + LP_TILE); }
#pragma unroll
            for (int u = 0; u < LP_ITEMS; u++) rec[u] = nxt[u];
            if (PREFETCH && t0 + LP_TILE < nRec) { loadTile(t0 + LP_TILE); if (LIST) loadLines(t0 + 2 * (uint64_t) LP_TILE); }
#pragma unroll
            for (int u = 0; u < LP_ITEMS; u++) {
                bk[u] = 0; sq[u] = 0;
                {
                    if (!isSentinel(rec[u])) {
                        const uint32_t b = lineBucket<NUCL, MODE>(a.key, rec[u].kmer, nb);
                        if (EXTRAS && a.keepHi && (b < a.keepLo || b >= a.keepHi)) continue;
                        bk[u] = b; sq[u] = atomicAdd(&cnt[b], 1u); pending |= 1u << u;
                        if (EXTRAS) {
                            if (a.valueHist) atomicAdd(&vh[valueBin<NUCL>(rec[u].kmer, a.valueShift)], 1u);
                            if (NUCL && a.minKey) mn = min(mn, (unsigned long long) (rec[u].kmer | BIT63));
                        }
                    }
                }
            }
            // DIRECT (range partitions of grouped records): a tile of grouped records names few representatives with many records
            // each, so a bucket completes many lines per tile and the round scheme below (one line per bucket and round) would take
            // 2-13 rounds per tile.  Here `flushed` holds the record COUNT at the start of the tile; the lines a bucket completes in
            // this tile get consecutive output lines (one scan over the buckets), their records go straight from registers to HBM
            // (the 16-byte pieces of a line are written within the same tile and merge in L2), what was waiting in the open line
            // goes with them, and only the new open line stays in LDS: one round per tile whatever the distribution.
            if (MODE == KEY_RANGE && a.direct) {                    // (compiled into the range-partition instantiations only: hash partitions
                                                                    //  are evenly filled and gain nothing from it — measured)
                __syncthreads();
                uint32_t mine = 0;
                for (uint32_t b = tid; b < nb; b += LP_BLOCK) mine += cnt[b] / RPL - flushed[b] / RPL;
                const uint32_t incl = waveInclusiveScan(mine);
                if (laneId() == 63) sScan[tid >> 6] = incl;
                __syncthreads();
                uint32_t woff = 0, total = 0;
#pragma unroll
                for (int w = 0; w < LP_BLOCK / 64; w++) { const uint32_t v = sScan[w]; if (w < (int) (tid >> 6)) woff += v; total += v; }
                uint32_t run = woff + incl - mine;
                for (uint32_t b = tid; b < nb; b += LP_BLOCK) {
                    const uint32_t oldC = flushed[b], done = cnt[b] / RPL - oldC / RPL;
                    lbase[b] = run;
                    if (done && (oldC % RPL)) {                      // the open line completes: its earlier records wait in LDS
                        const uint64_t ol = outLine + run;
                        for (uint32_t r = 0; r < oldC % RPL; r++) out[ol * RPL + r] = buf[(size_t) b * RPL + r];
                        a.tags[ol] = pc.tagBase + b;
                    }
                    run += done;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < LP_ITEMS; u++) {
                    if ((pending >> u) & 1u) {
                        const uint32_t b = bk[u], L = sq[u] / RPL, oldL = flushed[b] / RPL, newL = cnt[b] / RPL;
                        if (L < newL) {
                            const uint64_t ol = outLine + lbase[b] + (L - oldL);
                            out[ol * RPL + (sq[u] % RPL)] = rec[u];
                            if (sq[u] % RPL == 0) a.tags[ol] = pc.tagBase + b;
                        } else buf[(size_t) b * RPL + (sq[u] % RPL)] = rec[u];
                    }
                }
                __syncthreads();
                for (uint32_t b = tid; b < nb; b += LP_BLOCK) flushed[b] = cnt[b];
                outLine += total;
                // The next tile's first step is an atomicAdd on cnt[]: it must not run ahead of another wavefront's copy above, or that
                // wavefront files the next tile's records under "flushed" and the line accounting of the bucket is off for the rest of
                // the piece (lines written twice or beyond the piece's range).  Round 2 had no barrier here; the window only opened when
                // another stream's kernels shared the CUs — the two rank threads of tests/test_gpu_sharded.py::test_full_size… at
                // 1 M reads faulted in this kernel (profiles/r03_call3_gdb_fault.log).
                __syncthreads();
                continue;
            }
            // rounds: a record joins the open line of its bucket when that line is the one it belongs to (its running number / 8);
            // a round closes at most one line per bucket, the workgroup then writes the closed lines as one contiguous run
            for (;;) {
                const uint32_t par = round & 1u;
#pragma unroll
                for (int u = 0; u < LP_ITEMS; u++) {
                    if ((pending >> u) & 1u) {
                        const uint32_t b = bk[u];
                        if (sq[u] / RPL == flushed[b]) {
                            buf[(size_t) b * RPL + (sq[u] % RPL)] = rec[u];
                            pending &= ~(1u << u);
                            if (sq[u] % RPL == RPL - 1) { const uint32_t q = atomicAdd(&sQ[par], 1u); queue[q] = (unsigned short) b; }
                        }
                    }
                }
                __syncthreads();
                const uint32_t nQ = sQ[par];
                for (uint32_t j = tid; j < nQ * RPL; j += LP_BLOCK) { const uint32_t b = queue[j / RPL]; out[(outLine + j / RPL) * RPL + (j % RPL)] = buf[(size_t) b * RPL + (j % RPL)]; }
                for (uint32_t j = tid; j < nQ; j += LP_BLOCK) { const uint32_t b = queue[j]; a.tags[outLine + j] = pc.tagBase + b; flushed[b] += 1u; }
                const int more = __syncthreads_or(pending != 0u);
                outLine += nQ;
                if (tid == 0) sQ[par] = 0;                           // next used two rounds from now, behind the next round's barriers
                round++;
                if (!more) break;
            }
        }
        // the open lines of the piece, padded with sentinels
        {
            const uint32_t par = round & 1u;
            __syncthreads();
            auto openRecords = [&](uint32_t b) { return (MODE == KEY_RANGE && a.direct) ? cnt[b] % RPL : cnt[b] - flushed[b] * RPL; };
            for (uint32_t b = tid; b < nb; b += LP_BLOCK) if (openRecords(b)) { const uint32_t q = atomicAdd(&sQ[par], 1u); queue[q] = (unsigned short) b; }
            __syncthreads();
            const uint32_t nQ = sQ[par];
            for (uint32_t j = tid; j < nQ * RPL; j += LP_BLOCK) {
                const uint32_t b = queue[j / RPL]; const uint32_t r = openRecords(b);
                out[(outLine + j / RPL) * RPL + (j % RPL)] = ((j % RPL) < r) ? buf[(size_t) b * RPL + (j % RPL)] : sen;
            }
            for (uint32_t j = tid; j < nQ; j += LP_BLOCK) a.tags[outLine + j] = pc.tagBase + queue[j];
            __syncthreads();
            outLine += nQ;
            if (tid == 0) sQ[par] = 0;
            round++;
        }
        for (uint64_t j = outLine + tid; j < pc.out0 + pc.outCap; j += LP_BLOCK) a.tags[j] = TAG_NONE;
        if (tid == 0 && a.pieceOut) a.pieceOut[piece] = (uint32_t) (outLine - pc.out0);
    }
    if (EXTRAS) {
        __syncthreads();
        if (a.valueHist) for (uint32_t i = tid; i < VH_BINS; i += LP_BLOCK) { const uint32_t c = vh[i]; if (c) atomicAdd(&a.valueHist[i], c); }
        if (NUCL && a.minKey) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned long long) __shfl_xor(mn, o, 64));
            if (laneId() == 0 && mn != ~0ULL) atomicMin(a.minKey, mn);
        }
    }
}

// =====================================================================================================
// line lists: counting sort of the tags
// =====================================================================================================
// (a) whole tag array -> per-bucket line lists (level 1, or the only level)
__global__ __launch_bounds__(256) void tagHistKernel(const uint32_t *__restrict__ tags, uint64_t nLines, uint32_t nb, uint32_t *__restrict__ count) {
    __shared__ uint32_t sh[LP_MAXB];
    for (uint32_t i = threadIdx.x; i < nb; i += 256) sh[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nLines; i += (uint64_t) gridDim.x * 256) { const uint32_t t = tags[i]; if (t != TAG_NONE) atomicAdd(&sh[t], 1u); }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += 256) { const uint32_t c = sh[i]; if (c) atomicAdd(&count[i], c); }
}
// exclusive prefix sums of at most LP_MAXB counts by one workgroup: start[nb + 1], cursor[nb] = start
__global__ __launch_bounds__(1024) void tagScanKernel(const uint32_t *__restrict__ count, uint32_t nb, uint32_t *__restrict__ start, uint32_t *__restrict__ cursor) {
    __shared__ uint32_t sWave[16];
    const uint32_t v = threadIdx.x < nb ? count[threadIdx.x] : 0u;
    const uint32_t incl = waveInclusiveScan(v);
    if (laneId() == 63) sWave[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) woff += sWave[w];
    const uint32_t ex = woff + incl - v;
    if (threadIdx.x < nb) { start[threadIdx.x] = ex; cursor[threadIdx.x] = ex; }
    if (threadIdx.x == nb - 1) start[nb] = ex + v;
}
// A chunk of tags per workgroup round: a bucket's entries of one chunk are reserved with ONE atomic on the bucket's cursor and land next to
// each other in the list.  Round 5: 1024 threads x 32 tags (was 256 x 16): with 1024 buckets a chunk of 4 096 tags gave every bucket 4
// entries — 16-byte pieces of list, 6.3 GB written for 1.1 GB of tags read per launch (profiles/r04_pmc_traffic.json) — a chunk of 32 768
// gives it a full 128-byte line on average.
constexpr int TS_BLOCK = 1024, TS_PER = 32, TS_CHUNK = TS_BLOCK * TS_PER;
__global__ __launch_bounds__(TS_BLOCK) void tagScatterKernel(const uint32_t *__restrict__ tags, uint64_t nLines, uint32_t nb, uint32_t *__restrict__ cursor, uint32_t *__restrict__ list) {
    __shared__ uint32_t sh[LP_MAXB];      // count of the chunk, then running position
    for (uint64_t c0 = (uint64_t) blockIdx.x * TS_CHUNK; c0 < nLines; c0 += (uint64_t) gridDim.x * TS_CHUNK) {
        for (uint32_t i = threadIdx.x; i < nb; i += TS_BLOCK) sh[i] = 0;
        __syncthreads();
        uint32_t t[TS_PER];
#pragma unroll
        for (int u = 0; u < TS_PER; u++) {
            const uint64_t i = c0 + (uint64_t) u * TS_BLOCK + threadIdx.x;
            t[u] = (i < nLines) ? tags[i] : TAG_NONE;
        }
#pragma unroll
        for (int u = 0; u < TS_PER; u++) if (t[u] != TAG_NONE) atomicAdd(&sh[t[u]], 1u);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += TS_BLOCK) { const uint32_t c = sh[i]; sh[i] = c ? atomicAdd(&cursor[i], c) : 0u; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TS_PER; u++) {
            const uint64_t i = c0 + (uint64_t) u * TS_BLOCK + threadIdx.x;
            if (t[u] != TAG_NONE) list[atomicAdd(&sh[t[u]], 1u)] = (uint32_t) i;
        }
        __syncthreads();
    }
}
// bucket b of a single-level partition: its lines are list[beg[b] .. beg[b] + cnt[b])
__global__ void listRangesKernel(const uint32_t *__restrict__ start, uint32_t nb, uint32_t *__restrict__ beg, uint32_t *__restrict__ cnt) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) { beg[i] = start[i]; cnt[i] = start[i + 1] - start[i]; }
}

// (b) level 2: the output range of level-1 bucket g (its "region") holds lines tagged g * nb2 + f; one workgroup per region
// builds the region's part of the list and the (begin, count) of its nb2 fine buckets
constexpr int TS_STAGE = 8;
__global__ __launch_bounds__(512) void tagSortRegionKernel(const uint32_t *__restrict__ tags, const uint64_t *__restrict__ regionBeg, const uint64_t *__restrict__ regionEnd,
                                                           uint32_t nRegions, uint32_t nb2, uint32_t *__restrict__ list, uint32_t *__restrict__ fineBeg, uint32_t *__restrict__ fineCnt) {
    __shared__ uint32_t sh[LP_MAXB];
    __shared__ uint32_t sWave[8];
    __shared__ uint32_t sCnt[LP_MAXB];                      // entries in a bucket's row
    __shared__ uint32_t sStage[TS_STAGE * LP_MAXB];         // the rows, entry-major (conflict-free for consecutive buckets)
    for (uint32_t i = threadIdx.x; i < LP_MAXB; i += 512) sCnt[i] = 0;
    for (uint32_t g = blockIdx.x; g < nRegions; g += gridDim.x) {
        const uint64_t r0 = regionBeg[g], r1 = regionEnd[g];
        const uint32_t base = g * nb2;
        for (uint32_t i = threadIdx.x; i < nb2; i += 512) sh[i] = 0;
        __syncthreads();
        // four independent tags per thread and round: the loop is a chain of global load -> LDS atomic, bound by latency
        for (uint64_t i = r0 + threadIdx.x; i < r1; i += 2048) {
            uint32_t t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = (i + 512u * u < r1) ? tags[i + 512u * u] : TAG_NONE;
#pragma unroll
            for (int u = 0; u < 4; u++) if (t[u] != TAG_NONE) atomicAdd(&sh[t[u] - base], 1u);
        }
        __syncthreads();
        // exclusive scan of the nb2 <= 1024 counts: two per thread
        const uint32_t i0 = 2 * threadIdx.x, i1 = i0 + 1;
        const uint32_t c0 = i0 < nb2 ? sh[i0] : 0u, c1 = i1 < nb2 ? sh[i1] : 0u;
        const uint32_t incl = waveInclusiveScan(c0 + c1);
        if (laneId() == 63) sWave[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) woff += sWave[w];
        const uint32_t ex = (uint32_t) r0 + woff + incl - (c0 + c1);        // list positions are output line numbers: a region's list lies in its own range
        if (i0 < nb2) { fineBeg[base + i0] = ex; fineCnt[base + i0] = c0; sh[i0] = ex; }
        if (i1 < nb2) { fineBeg[base + i1] = ex + c0; fineCnt[base + i1] = c1; sh[i1] = ex + c0; }
        __syncthreads();
        // The list entries of a fine bucket are staged in LDS, TS_STAGE at a time, and leave as one row (round 3: one 4-byte store per
        // line scattered over the region's 1024 open lists reached the HBM as 1.7e8 partial-line writes per launch — 6.5 bytes written
        // per byte of list).  Per batch of 2048 tags: an entry takes the next place of its bucket's row (or, when the row is full, its
        // final place in the list directly); after the barrier every full row is written and the bucket's position moves on.
        for (uint64_t i0 = r0; i0 < r1; i0 += 2048) {
            const uint64_t i = i0 + threadIdx.x;
            uint32_t t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = (i + 512u * u < r1) ? tags[i + 512u * u] : TAG_NONE;
#pragma unroll
            for (int u = 0; u < 4; u++) if (t[u] != TAG_NONE) {
                const uint32_t b = t[u] - base, k = atomicAdd(&sCnt[b], 1u);
                if (k < (uint32_t) TS_STAGE) sStage[k * LP_MAXB + b] = (uint32_t) (i + 512u * u); else list[sh[b] + k] = (uint32_t) (i + 512u * u);
            }
            __syncthreads();
            for (uint32_t b = threadIdx.x; b < nb2; b += 512) {
                const uint32_t c = sCnt[b];
                if (c >= (uint32_t) TS_STAGE) {
                    uint32_t row[TS_STAGE];
#pragma unroll
                    for (int j = 0; j < TS_STAGE; j++) row[j] = sStage[j * LP_MAXB + b];
                    __builtin_memcpy(list + sh[b], row, sizeof(row));
                    sh[b] += c; sCnt[b] = 0;
                }
            }
            __syncthreads();
        }
        for (uint32_t b = threadIdx.x; b < nb2; b += 512) {             // what is left in the rows
            const uint32_t c = sCnt[b], p = sh[b];
            for (uint32_t j = 0; j < c; j++) list[p + j] = sStage[j * LP_MAXB + b];
            sCnt[b] = 0;
        }
        __syncthreads();
    }
}

// =====================================================================================================
// piece tables
// =====================================================================================================
// exclusive scan over the 1024 threads of a workgroup of (pieces, output lines); totals to every thread
__device__ __forceinline__ void blockScan1024(uint32_t a, uint64_t b, uint32_t &exA, uint64_t &exB, uint32_t &totA, uint64_t &totB, uint32_t *sA, unsigned long long *sB) {
    const uint32_t inA = waveInclusiveScan(a); const unsigned long long inB = waveInclusiveScanU64(b);
    const uint32_t w = threadIdx.x >> 6;
    if (laneId() == 63) { sA[w] = inA; sB[w] = inB; }
    __syncthreads();
    uint32_t oa = 0, ta = 0; unsigned long long ob = 0, tb = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) { if (k < w) { oa += sA[k]; ob += sB[k]; } ta += sA[k]; tb += sB[k]; }
    exA = oa + inA - a; exB = ob + inB - b; totA = ta; totB = tb;
    __syncthreads();
}
// level 2: pieces of every level-1 bucket's line list, and the bucket's output region
__global__ __launch_bounds__(1024) void planListKernel(const uint32_t *__restrict__ start, uint32_t nb1, uint32_t pieceLines, uint32_t nb2,
                                                       LinePiece *__restrict__ pieces, uint32_t *__restrict__ nPieces, uint64_t *__restrict__ regionBeg,
                                                       uint64_t *__restrict__ regionEnd, uint64_t *__restrict__ totalOut) {
    __shared__ uint32_t sA[16]; __shared__ unsigned long long sB[16];
    const uint32_t g = threadIdx.x;                                   // nb1 <= 1024: one level-1 bucket per thread
    const uint32_t s0 = g < nb1 ? start[g] : 0u;
    const uint32_t c = g < nb1 ? start[g + 1] - s0 : 0u;
    const uint32_t np = (uint32_t) (((uint64_t) c + pieceLines - 1) / pieceLines);      // pieceLines = 0xFFFFFFFF: one piece per bucket
    uint32_t p0, pt; uint64_t o0, ot;
    blockScan1024(np, (uint64_t) c + (uint64_t) np * nb2, p0, o0, pt, ot, sA, sB);
    if (g == 0) { *nPieces = pt; *totalOut = ot; }
    if (g < nb1) {
        regionBeg[g] = o0; regionEnd[g] = o0 + (uint64_t) c + (uint64_t) np * nb2;
        for (uint32_t p = 0; p < np; p++) {
            LinePiece pc;
            pc.in0 = (uint64_t) s0 + (uint64_t) p * pieceLines;
            pc.nLines = (uint32_t) min((uint64_t) pieceLines, (uint64_t) c - (uint64_t) p * pieceLines);
            pc.lastValid = RPL;
            pc.out0 = o0 + (uint64_t) p * ((uint64_t) pieceLines + nb2);
            pc.outCap = pc.nLines + nb2;
            pc.tagBase = g * nb2;
            pieces[p0 + p] = pc;
        }
    }
}
// level 1 over segments of a dense record array (the arenas the group kernel wrote): segStart in records (multiples of RPL)
__global__ __launch_bounds__(1024) void planSegKernel(const uint64_t *__restrict__ segStart, const uint64_t *__restrict__ segCount, uint32_t nSeg, uint32_t pieceLines,
                                                      uint32_t nb, LinePiece *__restrict__ pieces, uint32_t *__restrict__ nPieces, uint64_t *__restrict__ totalOut) {
    __shared__ uint32_t sA[16]; __shared__ unsigned long long sB[16];
    const uint32_t per = (nSeg + 1023) / 1024;                        // consecutive segments per thread
    const uint32_t sBeg = threadIdx.x * per, sEnd = min(nSeg, sBeg + per);
    uint32_t myP = 0; uint64_t myO = 0;
    for (uint32_t s = sBeg; s < sEnd; s++) {
        const uint64_t lines = (segCount[s] + RPL - 1) / RPL;
        const uint32_t np = (uint32_t) ((lines + pieceLines - 1) / pieceLines);
        myP += np; myO += lines + (uint64_t) np * nb;
    }
    uint32_t p0, pt; uint64_t o0, ot;
    blockScan1024(myP, myO, p0, o0, pt, ot, sA, sB);
    if (threadIdx.x == 0) { *nPieces = pt; *totalOut = ot; }
    for (uint32_t s = sBeg; s < sEnd; s++) {
        const uint64_t cnt = segCount[s];
        const uint64_t lines = (cnt + RPL - 1) / RPL;
        const uint32_t np = (uint32_t) ((lines + pieceLines - 1) / pieceLines);
        for (uint32_t p = 0; p < np; p++) {
            LinePiece pc;
            pc.in0 = segStart[s] / RPL + (uint64_t) p * pieceLines;
            const uint64_t left = lines - (uint64_t) p * pieceLines;
            pc.nLines = (uint32_t) (left < (uint64_t) pieceLines ? left : (uint64_t) pieceLines);
            pc.lastValid = (p == np - 1) ? (uint32_t) (cnt - (lines - 1) * RPL) : (uint32_t) RPL;
            pc.out0 = o0 + (uint64_t) p * ((uint64_t) pieceLines + nb);
            pc.outCap = pc.nLines + nb;
            pc.tagBase = 0;
            pieces[p0 + p] = pc;
        }
        p0 += np; o0 += lines + (uint64_t) np * nb;
    }
}

// record accessor of a bucket held as a line list: record i of the bucket (sentinels included) — used by the group and the
// aggregation kernels
template <class R> struct LineBucket {
    const R *recs; const uint32_t *list; uint32_t beg, lines;
    __device__ __forceinline__ uint32_t size() const { return lines * RPL; }
    __device__ __forceinline__ R at(uint32_t i) const { return recs[(uint64_t) list[beg + i / RPL] * RPL + (i % RPL)]; }
};

}  // namespace plasship
