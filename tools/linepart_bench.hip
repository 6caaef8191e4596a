// Stand-alone check + microbenchmark of the line-store partition (plass_amd/csrc/linepart.hpp) — development tool.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plass_amd/csrc tools/linepart_bench.hip -o tools/linepart_bench
//   run:   tools/linepart_bench [records (default 2^26)] [sentinel percent (default 30)]
// Correctness (always): two partition levels over random 16-byte records with skewed key multiplicities; every record must come
// out exactly once, in the bucket its key hashes to (per-bucket multiset comparison with a host partition).
// Timing: HIP events around each kernel for several bucket counts and grid sizes; GB/s = (bytes read + bytes written) / time.
#include "linepart.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace plasship;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef Rec<false> R;

static uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ULL; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; }
static uint64_t hostKmerMix(uint64_t K) { uint64_t x = K * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32; return x; }

__global__ void genKernel(R *out, uint64_t n, uint32_t sentinelPct, uint64_t distinct) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        uint64_t x = i + 0x9E3779B97F4A7C15ULL; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
        R r;
        if ((x % 100) < sentinelPct) { memset(&r, 0xFF, sizeof(R)); }
        else {
            uint64_t y = x * 0xD6E8FEB86659FD93ULL; y ^= y >> 32;
            // skew: a quarter of the records share 1/1000 of the keys
            uint64_t key = ((y & 3) == 0) ? (y >> 8) % (distinct / 1000 + 1) : (y >> 8) % distinct;
            r.kmer = key * 2654435761ULL % (1ULL << 50); r.id = (uint32_t) i; r.len = (uint16_t) (x >> 40); r.pos = (int16_t) (x >> 20);
        }
        out[i] = r;
    }
}

struct Timer { hipEvent_t a, b; Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); } void start() { CK(hipEventRecord(a, 0)); } float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; } };

template <class K> static void setLds(K k, size_t bytes) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes)); }

int main(int argc, char **argv) {
    const uint64_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 26);
    const uint32_t senPct = argc > 2 ? (uint32_t) atoi(argv[2]) : 30;
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int numCU = prop.multiProcessorCount;
    printf("device %s, %d CUs, LDS per block max %zu\n", prop.name, numCU, (size_t) prop.sharedMemPerBlock);
    const uint64_t totalLines = (N + RPL - 1) / RPL;
    const uint32_t lastValid = (uint32_t) (N - (totalLines - 1) * RPL);
    R *dIn; CK(hipMalloc(&dIn, totalLines * RPL * sizeof(R)));
    genKernel<<<numCU * 8, 256>>>(dIn, N, senPct, std::max<uint64_t>(N / 8, 16));
    CK(hipDeviceSynchronize());
    Timer tm;
    int fails = 0;
    // geo: 0 = 512 threads x 8 records, 1 = 1024 x 4 + prefetch, 2 = 512 x 8 + prefetch, 3 = 1024 x 4, 4 = 512 x 4 + prefetch
    struct Cfg { int b1, b2; int blocksPerCU; int geo; };
    std::vector<Cfg> cfgs;
    const int maxBits = 10;
    for (int geo = 0; geo < 5; geo++) { cfgs.push_back({maxBits, maxBits, 1, geo}); cfgs.push_back({maxBits - 1, maxBits - 1, 2, geo}); if (RPL == 4) cfgs.push_back({maxBits, maxBits, 2, geo}); }
    cfgs.push_back({maxBits, 0, 1, 1}); cfgs.push_back({7, 0, 4, 1}); cfgs.push_back({8, 8, 4, 4});
    for (const Cfg &cf : cfgs) {
        const uint32_t nb1 = 1u << cf.b1, nb2 = cf.b2 ? 1u << cf.b2 : 0;
        const uint32_t PL1 = (uint32_t) std::max<uint64_t>(nb1 * 8, std::min<uint64_t>((uint64_t) nb1 * 64, (totalLines + 2 * numCU - 1) / (2 * numCU)));
        const uint64_t nP1 = (totalLines + PL1 - 1) / PL1;
        const uint64_t cap1 = nP1 * ((uint64_t) PL1 + nb1);
        const uint32_t PL2 = nb2 ? (uint32_t) std::max<uint64_t>(nb2 * 16, (cap1 + 2 * numCU - 1) / (2 * numCU)) : 0;
        const uint64_t maxP2 = nb2 ? cap1 / PL2 + nb1 + 1 : 0;
        const uint64_t cap2 = nb2 ? cap1 + maxP2 * nb2 : 0;
        R *dL1, *dL2 = nullptr; uint32_t *dTag1, *dTag2 = nullptr, *dList1, *dList2 = nullptr, *dCnt, *dStart, *dCur, *dBeg, *dFCnt, *dNP; uint64_t *dRB, *dRE, *dTot; LinePiece *dPieces = nullptr;
        CK(hipMalloc(&dL1, cap1 * RPL * sizeof(R))); CK(hipMalloc(&dTag1, cap1 * 4)); CK(hipMalloc(&dList1, cap1 * 4));
        CK(hipMalloc(&dCnt, 2 * LP_MAXB * 4)); CK(hipMalloc(&dStart, (2 * LP_MAXB + 1) * 4)); CK(hipMalloc(&dCur, 2 * LP_MAXB * 4));
        const uint64_t nFine = nb2 ? (uint64_t) nb1 * nb2 : nb1;
        CK(hipMalloc(&dBeg, nFine * 4)); CK(hipMalloc(&dFCnt, nFine * 4)); CK(hipMalloc(&dNP, 4)); CK(hipMalloc(&dRB, LP_MAXB * 8)); CK(hipMalloc(&dRE, LP_MAXB * 8)); CK(hipMalloc(&dTot, 8));
        if (nb2) { CK(hipMalloc(&dL2, cap2 * RPL * sizeof(R))); CK(hipMalloc(&dTag2, cap2 * 4)); CK(hipMalloc(&dList2, cap2 * 4)); CK(hipMalloc(&dPieces, maxP2 * sizeof(LinePiece))); }
        CK(hipMemset(dCnt, 0, LP_MAXB * 4));
        LinePartArgs a; memset(&a, 0, sizeof(a));
        a.in = dIn; a.out = dL1; a.tags = dTag1; a.totalLines = totalLines; a.lastValidAll = lastValid; a.pieceLines = PL1; a.nb = nb1;
        a.key.shift = 64 - cf.b1; a.key.rangeBits = 0; a.key.repBase = 0;
        const size_t lds1 = linePartLdsBytes(nb1, sizeof(R), false);
        typedef void (*KernelT)(LinePartArgs);
        static const KernelT K1[5] = {linePartKernel<false, false, KEY_HASH, false, false, 512, 8, false>, linePartKernel<false, false, KEY_HASH, false, false, 1024, 4, true>,
                                      linePartKernel<false, false, KEY_HASH, false, false, 512, 8, true>, linePartKernel<false, false, KEY_HASH, false, false, 1024, 4, false>,
                                      linePartKernel<false, false, KEY_HASH, false, false, 512, 4, true>};
        static const KernelT K2[5] = {linePartKernel<false, false, KEY_HASH, true, false, 512, 8, false>, linePartKernel<false, false, KEY_HASH, true, false, 1024, 4, true>,
                                      linePartKernel<false, false, KEY_HASH, true, false, 512, 8, true>, linePartKernel<false, false, KEY_HASH, true, false, 1024, 4, false>,
                                      linePartKernel<false, false, KEY_HASH, true, false, 512, 4, true>};
        static const int BLK[5] = {512, 1024, 512, 1024, 512};
        const KernelT k1 = K1[cf.geo], k2 = K2[cf.geo]; const int LPB = BLK[cf.geo];
        setLds(k1, lds1);
        const unsigned grid1 = (unsigned) std::min<uint64_t>(nP1, (uint64_t) numCU * cf.blocksPerCU);
        float msP1 = 0, msT1 = 0, msPlan = 0, msP2 = 0, msT2 = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(dCnt, 0, LP_MAXB * 4));
            tm.start(); hipLaunchKernelGGL(k1, dim3(grid1), dim3(LPB), lds1, 0, a); msP1 = tm.stop();
            CK(hipGetLastError());
            tm.start();
            hipLaunchKernelGGL(tagHistKernel, dim3(numCU * 4), dim3(256), 0, 0, dTag1, cap1, nb1, dCnt);
            hipLaunchKernelGGL(tagScanKernel, dim3(1), dim3(1024), 0, 0, dCnt, nb1, dStart, dCur);
            hipLaunchKernelGGL(tagScatterKernel, dim3(numCU * 4), dim3(256), 0, 0, dTag1, cap1, nb1, dCur, dList1);
            msT1 = tm.stop();
            CK(hipGetLastError());
            if (nb2) {
                tm.start();
                hipLaunchKernelGGL(planListKernel, dim3(1), dim3(1024), 0, 0, dStart, nb1, PL2, nb2, dPieces, dNP, dRB, dRE, dTot);
                msPlan = tm.stop();
                LinePartArgs b; memset(&b, 0, sizeof(b));
                b.in = dL1; b.list = dList1; b.out = dL2; b.tags = dTag2; b.pieces = dPieces; b.nPieces = dNP; b.nb = nb2;
                b.key.shift = 64 - cf.b1 - cf.b2;
                const size_t lds2 = linePartLdsBytes(nb2, sizeof(R), false);
                setLds(k2, lds2);
                const unsigned grid2 = (unsigned) std::min<uint64_t>(maxP2, (uint64_t) numCU * cf.blocksPerCU);
                tm.start(); hipLaunchKernelGGL(k2, dim3(grid2), dim3(LPB), lds2, 0, b); msP2 = tm.stop();
                CK(hipGetLastError());
                tm.start();
                hipLaunchKernelGGL(tagSortRegionKernel, dim3(std::min<uint32_t>(nb1, numCU * 4)), dim3(512), 0, 0, dTag2, dRB, dRE, nb1, nb2, dList2, dBeg, dFCnt);
                msT2 = tm.stop();
                CK(hipGetLastError());
            } else {
                hipLaunchKernelGGL(listRangesKernel, dim3(4), dim3(256), 0, 0, dStart, nb1, dBeg, dFCnt);
                CK(hipDeviceSynchronize());
            }
        }
        // ---- verification: per fine bucket, the multiset of records ----
        uint64_t valid = 0;
        bool ok = true;
        if (N <= (1ull << 27)) {
            std::vector<R> hin(N); CK(hipMemcpy(hin.data(), dIn, N * sizeof(R), hipMemcpyDeviceToHost));
            const R *dRec = nb2 ? dL2 : dL1; const uint32_t *dList = nb2 ? dList2 : dList1; const uint64_t cap = nb2 ? cap2 : cap1;
            std::vector<R> hrec(cap * RPL); CK(hipMemcpy(hrec.data(), dRec, cap * RPL * sizeof(R), hipMemcpyDeviceToHost));
            std::vector<uint32_t> hlist(cap), hbeg(nFine), hcnt(nFine);
            CK(hipMemcpy(hlist.data(), dList, cap * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hbeg.data(), dBeg, nFine * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hcnt.data(), dFCnt, nFine * 4, hipMemcpyDeviceToHost));
            const int bits = cf.b1 + cf.b2;
            std::vector<uint64_t> want(nFine, 0), wantCnt(nFine, 0), got(nFine, 0), gotCnt(nFine, 0);
            for (uint64_t i = 0; i < N; i++) {
                const R &r = hin[i];
                if (r.kmer == ~0ULL && r.id == 0xFFFFFFFFu) continue;
                valid++;
                const uint64_t b = (hostKmerMix(r.kmer) >> (64 - bits)) & (nFine - 1);
                want[b] += mix(r.kmer ^ ((uint64_t) r.id << 20) ^ r.len ^ ((uint64_t) (uint16_t) r.pos << 44)); wantCnt[b]++;
            }
            uint64_t lastEnd = 0; bool mono = true;
            for (uint64_t b = 0; b < nFine; b++) {
                if (hcnt[b] && hbeg[b] < lastEnd) mono = false;
                if (hcnt[b]) lastEnd = (uint64_t) hbeg[b] + hcnt[b];
                for (uint32_t j = 0; j < hcnt[b]; j++) {
                    const uint64_t line = hlist[(uint64_t) hbeg[b] + j];
                    for (int s = 0; s < RPL; s++) {
                        const R &r = hrec[line * RPL + s];
                        if (r.kmer == ~0ULL && r.id == 0xFFFFFFFFu) continue;
                        const uint64_t bb = (hostKmerMix(r.kmer) >> (64 - bits)) & (nFine - 1);
                        if (bb != b) { ok = false; continue; }
                        got[b] += mix(r.kmer ^ ((uint64_t) r.id << 20) ^ r.len ^ ((uint64_t) (uint16_t) r.pos << 44)); gotCnt[b]++;
                    }
                }
            }
            for (uint64_t b = 0; b < nFine; b++) if (want[b] != got[b] || wantCnt[b] != gotCnt[b]) ok = false;
            if (!mono) { ok = false; printf("  bucket list ranges are not monotone\n"); }
        } else valid = N * (100 - senPct) / 100;
        if (!ok) fails++;
        const double inB = (double) N * sizeof(R), recB = (double) valid * sizeof(R);
        printf("RPL=%d geo=%d b1=%d b2=%d blocks/CU=%d PL1=%u PL2=%u | P1 %.3f ms (%.0f GB/s r+w) tags1 %.3f ms | plan %.3f P2 %.3f ms (%.0f GB/s r+w) tags2 %.3f ms | total %.3f ms = %.0f GB/s of 2*s*N_valid | %s\n",
               RPL, cf.geo, cf.b1, cf.b2, cf.blocksPerCU, PL1, PL2, msP1, (inB + recB) / msP1 / 1e6, msT1, msPlan, msP2, nb2 ? 2 * recB / msP2 / 1e6 : 0.0, msT2,
               msP1 + msT1 + msPlan + msP2 + msT2, 2 * recB / (msP1 + msT1 + msPlan + msP2 + msT2) / 1e6, (N <= (1ull << 27)) ? (ok ? "VERIFIED" : "MISMATCH") : "not verified (large)");
        fflush(stdout);
        CK(hipFree(dL1)); CK(hipFree(dTag1)); CK(hipFree(dList1)); CK(hipFree(dCnt)); CK(hipFree(dStart)); CK(hipFree(dCur)); CK(hipFree(dBeg)); CK(hipFree(dFCnt)); CK(hipFree(dNP)); CK(hipFree(dRB)); CK(hipFree(dRE)); CK(hipFree(dTot));
        if (nb2) { CK(hipFree(dL2)); CK(hipFree(dTag2)); CK(hipFree(dList2)); CK(hipFree(dPieces)); }
    }
    printf("linepart_bench: %d failing configurations\n", fails);
    return fails ? 1 : 0;
}
