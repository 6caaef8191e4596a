// plasship: cyclecheck on gfx950.  Product code.
//
// Reference behaviour reproduced (src/assembler/cyclecheck.cpp:94-283): per nucleotide contig the k-mers (k = 22, index
// = sum code[i] * 4^i over the codes A0 C1 T2 G3 X4, windows with X included) of the front, middle and back third are
// matched against each other — the FIRST (smallest position) front occurrence of a k-mer against every middle and back
// occurrence, the first middle occurrence against every back occurrence (:151-216) — and every match on a diagonal
// >= seqLen/3 increments a histogram.  The first diagonal d whose band (+- 1 % of the diagonal's length, only bins not
// larger than its own) holds more than 0.2 hits per k-mer position makes the contig circular (:241-269); it is written
// out whole or cut at that diagonal (--chop-cycle).  The third a k-mer belongs to is decided by the position BEFORE the
// reference's iterator advances (:118-136): k-mer p counts as front if 1 <= p <= L/3 + 1, middle if p <= 2(L/3) + 1,
// and the k-mer at position 0 ("-1" as unsigned) lands in the back list.
//
// MI355X-first: the reference sorts three k-mer lists per contig; only "first occurrence per k-mer" matters, so a hash
// table of (k-mer -> smallest front position, smallest middle position) replaces the sorts — in LDS, one wavefront per
// sequence, for everything up to 3 000 nt (reads and most contigs; the table is sized to the sequence), in HBM scratch with one workgroup per sequence for the
// long contigs.  Floating point: the hit rate is one float division compared with the double 0.2, as in the reference.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <memory>
#include <cstring>

namespace plasship {

constexpr int CC_K = 22;
constexpr unsigned long long CC_EMPTY = ~0ULL;

struct CycArgs {
    SeqView s;
    const unsigned char *map;        // 256-entry letter -> code
    const uint32_t *list; uint32_t nList;
    uint32_t *split;                 // [n] split diagonal, 0 = not circular
    // long tier: per list entry scratch
    unsigned long long *keys; uint32_t *minF, *minM, *hits;
    const uint64_t *slotOff; const uint32_t *slotCnt; const uint64_t *hitOff;
};

__device__ __forceinline__ int kmerClass(uint32_t p, uint32_t third) {      // 0 front, 1 middle, 2 back
    if (p == 0) return 2;
    if (p <= third + 1) return 0;
    if (p <= 2 * third + 1) return 1;
    return 2;
}
__device__ __forceinline__ uint32_t hashSlot(unsigned long long k, uint32_t mask) { return (uint32_t) ((k * 0x9E3779B97F4A7C15ULL) >> 40) & mask; }

// the reference's band test for diagonal bin d (cyclecheck.cpp:245-266); hits has 2 * third + 1 bins
__device__ __forceinline__ bool bandPasses(const uint32_t *hits, uint32_t d, uint32_t third, uint32_t L) {
    const uint32_t hd = hits[d];
    const uint32_t diag = d + third, diaglen = L - diag;
    const uint32_t gapwindow = (uint32_t) ((double) diaglen * 0.01);
    const uint32_t lower = (uint32_t) max(0, (int) (d - gapwindow));
    const uint32_t upper = min(d + gapwindow, 2 * third);
    uint32_t band = 0;
    for (uint32_t i = lower; i <= upper; i++) { const uint32_t h = hits[i]; if (h <= hd) band += h; }
    const float rate = (float) band / (float) ((unsigned long long) diaglen - (unsigned long long) CC_K + 1ULL);
    return (double) rate > 0.2;
}

// ---- one wavefront per sequence, everything in LDS -------------------------------------------------------------------
template <int SLOTS, int MAXL>
__global__ __launch_bounds__(64) void cycleWaveKernel(CycArgs a) {
    __shared__ unsigned char sMap[256];
    __shared__ unsigned char sNum[MAXL + CC_K + 8];
    __shared__ unsigned long long sKey[SLOTS];
    __shared__ uint32_t sMinF[SLOTS], sMinM[SLOTS];
    __shared__ uint32_t sHits[2 * (MAXL / 3) + 2];
    __shared__ uint32_t sAny;
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sMap[i] = a.map[i];
    __syncthreads();
    for (uint32_t w = blockIdx.x; w < a.nList; w += gridDim.x) {
        const uint32_t id = a.list[w];
        const uint32_t L = a.s.len[id];
        const char *seq = a.s.data + a.s.off[id];
        const uint32_t third = L / 3, nk = L - CC_K + 1, nBins = 2 * third + 1;
        // table sized to the sequence (load <= 0.5 for its 2L/3 front + middle k-mers): clearing it is most of the work
        // for a sequence without repeats
        uint32_t slots = 256; while (slots * 3 < L * 4 + 64 && slots < (uint32_t) SLOTS) slots <<= 1;
        const uint32_t mask = slots - 1;
        for (uint32_t i = lane; i < L; i += 64) sNum[i] = sMap[(unsigned char) seq[i]];
        for (uint32_t i = lane; i < slots; i += 64) { sKey[i] = CC_EMPTY; sMinF[i] = 0xFFFFFFFFu; sMinM[i] = 0xFFFFFFFFu; }
        for (uint32_t i = lane; i < nBins; i += 64) sHits[i] = 0;
        if (lane == 0) sAny = 0;
        __syncthreads();
        auto kmerAt = [&](uint32_t p) { unsigned long long v = 0; for (int i = CC_K - 1; i >= 0; i--) v = v * 4ULL + sNum[p + i]; return v; };
        // phase 1: front and middle k-mers -> table (smallest position per k-mer and third)
        for (uint32_t p = lane; p < nk; p += 64) {
            const int c = kmerClass(p, third);
            if (c == 2) continue;
            const unsigned long long k = kmerAt(p);
            uint32_t sl = hashSlot(k, mask);
            for (;;) {
                const unsigned long long prev = atomicCAS(&sKey[sl], CC_EMPTY, k);
                if (prev == CC_EMPTY || prev == k) break;
                sl = (sl + 1) & mask;
            }
            atomicMin(c == 0 ? &sMinF[sl] : &sMinM[sl], p);
        }
        __syncthreads();
        // phase 2: middle k-mers against the first front occurrence, back k-mers against the first front and first middle one
        for (uint32_t p = lane; p < nk; p += 64) {
            const int c = kmerClass(p, third);
            if (c == 0) continue;
            const unsigned long long k = kmerAt(p);
            uint32_t sl = hashSlot(k, mask);
            for (;;) {
                const unsigned long long kk = sKey[sl];
                if (kk == k || kk == CC_EMPTY) { if (kk != k) sl = 0xFFFFFFFFu; break; }
                sl = (sl + 1) & mask;
            }
            if (sl == 0xFFFFFFFFu) continue;
            const uint32_t mf = sMinF[sl], mm = sMinM[sl];
            if (mf != 0xFFFFFFFFu) { const int diag = (int) p - (int) mf; if (diag >= (int) third) { atomicAdd(&sHits[diag - (int) third], 1u); sAny = 1; } }
            if (c == 2 && mm != 0xFFFFFFFFu) { const int diag = (int) p - (int) mm; if (diag >= (int) third) { atomicAdd(&sHits[diag - (int) third], 1u); sAny = 1; } }
        }
        __syncthreads();
        uint32_t split = 0;
        if (sAny) {
            for (uint32_t base = 0; base < 2 * third; base += 64) {
                const uint32_t d = base + lane;
                const bool pass = d < 2 * third && sHits[d] != 0 && bandPasses(sHits, d, third, L);
                const unsigned long long m = __ballot(pass);
                if (m) { split = base + (uint32_t) __builtin_ctzll(m) + third; break; }
            }
        }
        if (lane == 0) a.split[id] = split;
        __syncthreads();
    }
}

// ---- one workgroup per long sequence, table and histogram in HBM scratch -----------------------------------------------
__global__ __launch_bounds__(256) void cycleBlockKernel(CycArgs a) {
    __shared__ unsigned char sMap[256];
    __shared__ uint32_t sAny, sFirst;
    for (int i = threadIdx.x; i < 256; i += 256) sMap[i] = a.map[i];
    __syncthreads();
    for (uint32_t w = blockIdx.x; w < a.nList; w += gridDim.x) {
        const uint32_t id = a.list[w];
        const uint32_t L = a.s.len[id];
        const char *seq = a.s.data + a.s.off[id];
        const uint32_t third = L / 3, nk = L - CC_K + 1, nBins = 2 * third + 1;
        const uint32_t slots = a.slotCnt[w], mask = slots - 1;
        unsigned long long *keys = a.keys + a.slotOff[w];
        uint32_t *minF = a.minF + a.slotOff[w], *minM = a.minM + a.slotOff[w], *hits = a.hits + a.hitOff[w];
        for (uint32_t i = threadIdx.x; i < slots; i += 256) { keys[i] = CC_EMPTY; minF[i] = 0xFFFFFFFFu; minM[i] = 0xFFFFFFFFu; }
        for (uint32_t i = threadIdx.x; i < nBins; i += 256) hits[i] = 0;
        if (threadIdx.x == 0) { sAny = 0; sFirst = 0xFFFFFFFFu; }
        __syncthreads();
        // every thread walks a contiguous range of positions with a rolling index: idx' = (idx - code[p]) / 4 + code[p + k] * 4^(k-1)
        const uint32_t chunk = (nk + 255) / 256;
        const uint32_t p0 = threadIdx.x * chunk, p1 = min(nk, p0 + chunk);
        auto code = [&](uint32_t i) { return (unsigned long long) sMap[(unsigned char) seq[i]]; };
        auto first = [&](uint32_t p) { unsigned long long v = 0; for (int i = CC_K - 1; i >= 0; i--) v = v * 4ULL + code(p + i); return v; };
        if (p0 < p1) {
            unsigned long long k = first(p0);
            for (uint32_t p = p0; p < p1; p++) {
                const int c = kmerClass(p, third);
                if (c != 2) {
                    uint32_t sl = hashSlot(k, mask);
                    for (;;) {
                        const unsigned long long prev = atomicCAS(&keys[sl], CC_EMPTY, k);
                        if (prev == CC_EMPTY || prev == k) break;
                        sl = (sl + 1) & mask;
                    }
                    atomicMin(c == 0 ? &minF[sl] : &minM[sl], p);
                }
                if (p + 1 < p1) k = ((k - code(p)) >> 2) + (code(p + CC_K) << (2 * (CC_K - 1)));
            }
        }
        __threadfence();
        __syncthreads();
        if (p0 < p1) {
            unsigned long long k = first(p0);
            for (uint32_t p = p0; p < p1; p++) {
                const int c = kmerClass(p, third);
                if (c != 0) {
                    uint32_t sl = hashSlot(k, mask);
                    for (;;) {
                        const unsigned long long kk = keys[sl];
                        if (kk == k || kk == CC_EMPTY) { if (kk != k) sl = 0xFFFFFFFFu; break; }
                        sl = (sl + 1) & mask;
                    }
                    if (sl != 0xFFFFFFFFu) {
                        const uint32_t mf = minF[sl], mm = minM[sl];
                        if (mf != 0xFFFFFFFFu) { const int diag = (int) p - (int) mf; if (diag >= (int) third) { atomicAdd(&hits[diag - (int) third], 1u); sAny = 1; } }
                        if (c == 2 && mm != 0xFFFFFFFFu) { const int diag = (int) p - (int) mm; if (diag >= (int) third) { atomicAdd(&hits[diag - (int) third], 1u); sAny = 1; } }
                    }
                }
                if (p + 1 < p1) k = ((k - code(p)) >> 2) + (code(p + CC_K) << (2 * (CC_K - 1)));
            }
        }
        __threadfence();
        __syncthreads();
        if (sAny) {
            for (uint32_t base = 0; base < 2 * third; base += 256) {
                const uint32_t d = base + threadIdx.x;
                if (d < 2 * third && hits[d] != 0 && bandPasses(hits, d, third, L)) atomicMin(&sFirst, d);
                __syncthreads();
                if (sFirst != 0xFFFFFFFFu) break;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) a.split[id] = (sFirst != 0xFFFFFFFFu) ? sFirst + third : 0u;
        __syncthreads();
    }
}

// tier of every sequence: 0-3 wave kernel with a table of up to 256 / 512 / 2048 / 4096 slots in LDS (the LDS a wavefront
// holds decides how many sequences a CU works on at a time: reads and merged read pairs get the small instantiations),
// 4 workgroup kernel; sequences without a k-mer or at / above --max-seq-len are not circular
constexpr uint32_t CC_L0 = 190, CC_LA = 380, CC_L1 = 1536, CC_L2 = 3000;
constexpr int CC_TIERS = 5;
__global__ void cycleTierKernel(SeqView s, uint64_t maxSeqLen, uint32_t *__restrict__ lists, uint32_t *__restrict__ counts, uint32_t *__restrict__ split) {
    for (uint32_t b0 = blockIdx.x * blockDim.x; b0 < s.n; b0 += gridDim.x * blockDim.x) {
        const uint32_t id = b0 + threadIdx.x;
        int tier = -1;
        if (id < s.n) {
            const uint32_t L = s.len[id];
            split[id] = 0;
            if (L >= (uint32_t) CC_K && (uint64_t) L < maxSeqLen) tier = L <= CC_L0 ? 0 : (L <= CC_LA ? 1 : (L <= CC_L1 ? 2 : (L <= CC_L2 ? 3 : 4)));
        }
        for (int t = 0; t < CC_TIERS; t++) {
            const unsigned long long m = __ballot(tier == t);
            if (!m) continue;
            uint32_t base = 0;
            if (laneId() == 0) base = atomicAdd(&counts[t], (uint32_t) __popcll(m));
            base = __shfl(base, 0, 64);
            if (tier == t) lists[(size_t) t * s.n + base + (uint32_t) __popcll(m & ((1ULL << laneId()) - 1ULL))] = id;
        }
    }
}

__global__ void cycleGatherLenKernel(const uint32_t *__restrict__ len, const uint32_t *__restrict__ ids, uint32_t n, uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = len[ids[i]];
}

__global__ void cycleFlagsKernel(SeqView s, const uint32_t *__restrict__ split, int chop, int invert, uint32_t *__restrict__ flags, uint32_t *__restrict__ newLen,
                                 uint64_t *__restrict__ newStart) {
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < s.n; id += gridDim.x * blockDim.x) {
        const uint32_t sp = split[id];
        const bool cyc = sp != 0;
        if (invert) { flags[id] = cyc ? 0x80u : 0u; newLen[id] = 0; newStart[id] = 0; }            // the remainder: everything that is not circular
        else { flags[id] = cyc ? 0x20u : 0x80u; newLen[id] = cyc ? (chop ? sp : s.len[id]) : 0u; newStart[id] = s.off[id]; }
    }
}

}  // namespace plasship
using namespace plasship;

extern "C" int plasship_cyclecheck(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_cyclecheck_params *par, plasship_seqdb **out_cycle,
                                   plasship_seqdb **out_rest, plasship_cyclecheck_stats *stats) {
    if (!ctx || !db || !par || !out_cycle) { setError("plasship_cyclecheck: bad argument"); return PLASSHIP_ERR_ARG; }
    if (db->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES) { setError("Module cyclecheck only supports nucleotide input database"); return PLASSHIP_ERR_ARG; }
    if (db->maxEntryLen >= (1u << 30)) { setError("plasship_cyclecheck: sequence too long"); return PLASSHIP_ERR_UNSUPPORTED; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t N = (uint32_t) db->n;
    const SeqView sv = db->view();
    DevBuf dMap, dLists, dCounts, dSplit, dFlags, dNewLen, dNewStart, dTmp;
    const size_t tmpBytes = exclusiveScanTmpBytes((size_t) N + 2);
    if (dMap.alloc(256) != hipSuccess || dLists.alloc(((size_t) CC_TIERS * N + 1) * 4) != hipSuccess || dCounts.alloc(32) != hipSuccess || dSplit.alloc(((size_t) N + 1) * 4) != hipSuccess ||
        dFlags.alloc(((size_t) N + 1) * 4) != hipSuccess || dNewLen.alloc(((size_t) N + 1) * 4) != hipSuccess || dNewStart.alloc(((size_t) N + 1) * 8) != hipSuccess ||
        dTmp.alloc(tmpBytes) != hipSuccess) { setError("plasship_cyclecheck: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    PH_CHECK(hipMemcpyAsync(dMap.p, aa2numTable(true, 5), 256, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemsetAsync(dCounts.p, 0, 32, st));
    const unsigned gridN = std::min<uint32_t>((N + 255) / 256 + 1, (uint32_t) ctx->numCU * 16);
    if (N) hipLaunchKernelGGL(cycleTierKernel, dim3(gridN), dim3(256), 0, st, sv, (uint64_t) par->max_seq_len, dLists.as<uint32_t>(), dCounts.as<uint32_t>(), dSplit.as<uint32_t>());
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PH_COPY_SYNC(st, cnt, dCounts.p, 32, hipMemcpyDeviceToHost);
    CycArgs a; memset(&a, 0, sizeof(a));
    a.s = sv; a.map = dMap.as<unsigned char>(); a.split = dSplit.as<uint32_t>();
    if (cnt[0]) { a.list = dLists.as<uint32_t>(); a.nList = cnt[0]; hipLaunchKernelGGL((cycleWaveKernel<256, CC_L0>), dim3(std::min<uint32_t>(cnt[0], (uint32_t) ctx->numCU * 32)), dim3(64), 0, st, a); }
    if (cnt[1]) { a.list = dLists.as<uint32_t>() + N; a.nList = cnt[1]; hipLaunchKernelGGL((cycleWaveKernel<512, CC_LA>), dim3(std::min<uint32_t>(cnt[1], (uint32_t) ctx->numCU * 24)), dim3(64), 0, st, a); }
    if (cnt[2]) { a.list = dLists.as<uint32_t>() + 2 * (size_t) N; a.nList = cnt[2]; hipLaunchKernelGGL((cycleWaveKernel<2048, CC_L1>), dim3(std::min<uint32_t>(cnt[2], (uint32_t) ctx->numCU * 8)), dim3(64), 0, st, a); }
    if (cnt[3]) { a.list = dLists.as<uint32_t>() + 3 * (size_t) N; a.nList = cnt[3]; hipLaunchKernelGGL((cycleWaveKernel<4096, CC_L2>), dim3(std::min<uint32_t>(cnt[3], (uint32_t) ctx->numCU * 4)), dim3(64), 0, st, a); }
    if (cnt[4]) {
        // long contigs: scratch tables in HBM, in batches of at most ~2 GB
        const uint32_t nLong = cnt[4];
        std::vector<uint32_t> lens(nLong);
        DevBuf dLens;
        if (dLens.alloc((size_t) nLong * 4) != hipSuccess) { setError("plasship_cyclecheck: out of device memory"); return PLASSHIP_ERR_DEVICE; }
        // lengths of the long sequences in list order (the host has no copy of a device-produced DB's index)
        hipLaunchKernelGGL(cycleGatherLenKernel, dim3(std::min<uint32_t>((nLong + 255) / 256, 1024)), dim3(256), 0, st, db->d_len.as<uint32_t>(), dLists.as<uint32_t>() + 4 * (size_t) N, nLong, dLens.as<uint32_t>());
        PH_COPY_SYNC(st, lens.data(), dLens.p, (size_t) nLong * 4, hipMemcpyDeviceToHost);
        const uint64_t budgetSlots = 1ull << 27;        // 16 bytes per slot
        uint32_t b0 = 0;
        while (b0 < nLong) {
            std::vector<uint64_t> slotOff, hitOff; std::vector<uint32_t> slotCnt;
            uint64_t so = 0, ho = 0; uint32_t b1 = b0;
            while (b1 < nLong) {
                uint32_t slots = 1024; while ((uint64_t) slots * 3 < (uint64_t) lens[b1] * 4 + 64) slots <<= 1;      // load <= 0.5 for the 2L/3 front + middle k-mers
                if (b1 > b0 && so + slots > budgetSlots) break;
                slotOff.push_back(so); slotCnt.push_back(slots); hitOff.push_back(ho);
                so += slots; ho += 2 * (uint64_t) (lens[b1] / 3) + 2; b1++;
            }
            const uint32_t nb = b1 - b0;
            DevBuf dKeys, dMinF, dMinM, dHits, dSO, dSC, dHO;
            if (dKeys.alloc(so * 8) != hipSuccess || dMinF.alloc(so * 4) != hipSuccess || dMinM.alloc(so * 4) != hipSuccess || dHits.alloc(ho * 4) != hipSuccess ||
                dSO.alloc((size_t) nb * 8) != hipSuccess || dSC.alloc((size_t) nb * 4) != hipSuccess || dHO.alloc((size_t) nb * 8) != hipSuccess) {
                setError("plasship_cyclecheck: out of device memory for the long-contig tables"); return PLASSHIP_ERR_DEVICE;
            }
            PH_CHECK(hipMemcpyAsync(dSO.p, slotOff.data(), (size_t) nb * 8, hipMemcpyHostToDevice, st));
            PH_CHECK(hipMemcpyAsync(dSC.p, slotCnt.data(), (size_t) nb * 4, hipMemcpyHostToDevice, st));
            PH_CHECK(hipMemcpyAsync(dHO.p, hitOff.data(), (size_t) nb * 8, hipMemcpyHostToDevice, st));
            a.list = dLists.as<uint32_t>() + 4 * (size_t) N + b0; a.nList = nb;
            a.keys = dKeys.as<unsigned long long>(); a.minF = dMinF.as<uint32_t>(); a.minM = dMinM.as<uint32_t>(); a.hits = dHits.as<uint32_t>();
            a.slotOff = dSO.as<uint64_t>(); a.slotCnt = dSC.as<uint32_t>(); a.hitOff = dHO.as<uint64_t>();
            hipLaunchKernelGGL(cycleBlockKernel, dim3(std::min<uint32_t>(nb, (uint32_t) ctx->numCU * 4)), dim3(256), 0, st, a);
            PH_CHECK(plasship::streamSync(st));
            PH_CHECK(hipGetLastError());
            b0 = b1;
        }
    }
    // the circular sequences (whole or cut) ...
    plasship_seqdb *oc = nullptr, *orest = nullptr;
    if (N) hipLaunchKernelGGL(cycleFlagsKernel, dim3(gridN), dim3(256), 0, st, sv, dSplit.as<uint32_t>(), par->chop_cycle ? 1 : 0, 0, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>());
    int rc = buildOutputDB(ctx, db, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>(), sv.data, 0, dTmp.p, tmpBytes, &oc, nullptr, nullptr, 0, ctx->ev[1]);
    if (rc != PLASSHIP_OK) return rc;
    std::unique_ptr<plasship_seqdb> holdC(oc);
    oc->dbtype = PLASSHIP_DBTYPE_NUCLEOTIDES;
    // ... and, if asked for, everything else (what the workflow continues with: "<db>_noneCycle", data/nuclassemble.sh:27-35)
    if (out_rest) {
        if (N) hipLaunchKernelGGL(cycleFlagsKernel, dim3(gridN), dim3(256), 0, st, sv, dSplit.as<uint32_t>(), 0, 1, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>());
        rc = buildOutputDB(ctx, db, dFlags.as<uint32_t>(), dNewLen.as<uint32_t>(), dNewStart.as<uint64_t>(), sv.data, 0, dTmp.p, tmpBytes, &orest);
        if (rc != PLASSHIP_OK) return rc;
    }
    if (stats) {
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        stats->ms_kernel = ms; stats->n_cyclic = oc->n; stats->n_wave_small = cnt[0] + cnt[1]; stats->n_wave_large = cnt[2] + cnt[3]; stats->n_block = cnt[4];
    }
    *out_cycle = holdC.release();
    if (out_rest) *out_rest = orest;
    return PLASSHIP_OK;
}
