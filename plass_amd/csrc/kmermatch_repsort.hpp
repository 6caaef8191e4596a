// plasship: kmermatcher on gfx950, stages K6-K8: sort #2 as aggregation, run reduction, the stale-record scan.  Product code; part of kmermatch.hip's translation unit (included there, inside
// namespace plasship, after common.hpp / device_utils.hpp / linepart.hpp) — split out by stage in round 4, see kmermatch.hip for the
// reference lines the stage reproduces and DESIGN.md section 4 for the kernels' bounds.
// Kernels: aggSortKernel, repRunsKernel / placeRunsKernel, reduceRunsKernel, rankLinesKernel and helpers.
#pragma once

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

