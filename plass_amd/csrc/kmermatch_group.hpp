// plasship: kmermatcher on gfx950, stage K5: assignGroup over hash buckets.  Product code; part of kmermatch.hip's translation unit (included there, inside
// namespace plasship, after common.hpp / device_utils.hpp / linepart.hpp) — split out by stage in round 4, see kmermatch.hip for the
// reference lines the stage reproduces and DESIGN.md section 4 for the kernels' bounds.
// Kernels: groupKernel (24-byte records, sharded run), groupLinesKernel (16-byte records over the line store).
#pragma once

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
// LAHEAD (round 6): the fetch of a bucket is two dependent round trips (line list entry, then the record), and the counters say the
// wavefronts are parked two thirds of their cycles (SQ_WAIT_ANY 65 %, LDS pipe 10 %).  A second set of record registers does not fit (126 of 128 VGPRs:
// profiles/r06_ab_knobs.txt), but the LIST ENTRIES of the next bucket do — one load per thread at the start of a bucket, parked in LDS behind its first
// phase — so that the record loads at the bucket's end start from LDS: one round trip on the critical path instead of two.
template <bool NUCL, int BLOCK, uint32_t HT, int WPE, bool LAHEAD = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void groupLinesKernel(GroupArgs a) {
    __shared__ uint32_t sList[LAHEAD ? GL_RMAX * BLOCK / RPL : 1];
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
    auto fetch = [&](uint32_t b, bool fromLds) {
        nNext = (b < bEnd) ? a.lineCnt[b] * RPL : 0u; lbNext = (b < bEnd) ? a.lineBeg[b] : 0u;
        if (nNext && nNext <= (uint32_t) GL_RMAX * BLOCK) {
#pragma unroll
            for (int j = 0; j < GL_RMAX; j++) {
                const uint32_t i = (uint32_t) j * BLOCK + threadIdx.x;
                const uint32_t le = (LAHEAD && fromLds) ? sList[LAHEAD ? i / RPL : 0] : a.list[lbNext + i / RPL];
                rg[j] = (i < nNext) ? in[(uint64_t) le * RPL + (i % RPL)] : none;
            }
        }
    };
    uint32_t par = 0;                                // which cursor the current sub-pass uses (workgroup-uniform)
    fetch(bBegin, false);
    for (uint32_t b = bBegin; b < bEnd; b++) {
        const uint32_t n = nNext;                    // record positions of the bucket (padding sentinels included)
        const uint32_t lb = lbNext;
        if (n == 0) { fetch(b + 1, false); continue; }
        // (LAHEAD) the next bucket's list entries: requested here, stored to LDS behind this bucket's first phase
        uint32_t leAhead = 0; bool aheadStored = false;
        if (LAHEAD && b + 1 < bEnd) { const uint32_t nl = a.lineCnt[b + 1]; if (threadIdx.x < nl && nl <= (uint32_t) GL_RMAX * BLOCK / RPL) leAhead = a.list[a.lineBeg[b + 1] + threadIdx.x]; }
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
                // (round 6: double hashing — an odd step from other bits of the hash instead of 1 — changed nothing, 38.7 -> 38.3 ms at 50 M reads: the
                //  table is loaded to a third, probing is not where the time is; profiles/r06_ab_knobs.txt, call 3)
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
                if (LAHEAD && !aheadStored) { if (threadIdx.x < GL_RMAX * BLOCK / RPL) sList[LAHEAD ? threadIdx.x : 0] = leAhead; aheadStored = true; }
                __syncthreads();
                if (sFlag[1] || sFlag[0] > MAXKEYS) { redo = true; __syncthreads(); break; }
                if (inRegs && LAHEAD && sub + 1 == nSub) {
                    // last sub-pass: a record register is dead once its phase C is through — the next bucket's record takes it at once (list entries from
                    // LDS), so the loads of registers 0..6 travel under the rest of this phase instead of starting behind it
                    const uint32_t nN = (b + 1 < bEnd) ? a.lineCnt[b + 1] * RPL : 0u, lbN = (b + 1 < bEnd) ? a.lineBeg[b + 1] : 0u;
                    const bool nextRegs = nN && nN <= (uint32_t) GL_RMAX * BLOCK;
#pragma unroll
                    for (int j = 0; j < GL_RMAX; j++) {
                        if ((uint32_t) j * BLOCK < n) phaseC(rg[j], nSub, sub);
                        if (nextRegs) { const uint32_t i = (uint32_t) j * BLOCK + threadIdx.x; rg[j] = (i < nN) ? in[(uint64_t) sList[LAHEAD ? i / RPL : 0] * RPL + (i % RPL)] : none; }
                    }
                    nNext = nN; lbNext = lbN; fetched = true;
                } else {
                if (inRegs) {
#pragma unroll
                    for (int j = 0; j < GL_RMAX; j++) if ((uint32_t) j * BLOCK < n) phaseC(rg[j], nSub, sub);
                } else for (uint32_t i0 = 0; i0 < n; i0 += BLOCK) { const uint32_t i = i0 + threadIdx.x; phaseC(i < n ? recAt(i) : none, nSub, sub); }
                if (sub + 1 == nSub) { fetch(b + 1, true); fetched = true; }     // last sub-pass: nothing reads this bucket's registers again
                }
                __syncthreads();
                written += sCursor[par];
                par ^= 1u;
            }
            if (!redo) break;
            nSub *= 2;                               // a retry discards what completed sub-passes of this attempt wrote
            written = writtenAtBucketStart;
            __syncthreads();
        }
        if (!fetched) fetch(b + 1, false);           // (cannot happen: the last sub-pass always completes; kept for the invariant)
    }
    if (a.maxRepTarget) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxRT = max(maxRT, (unsigned long long) __shfl_xor(maxRT, o, 64));
        if (laneId() == 0 && maxRT) atomicMax(a.maxRepTarget, maxRT);
    }
    if (threadIdx.x == 0) a.outCount[blockIdx.x] = written;
}

