// Stand-alone check + microbenchmark of the thread-per-pair scoring loop of rescorediagonal (plass_amd/csrc/rescore.hip, scoreDiagonal<false, 1>)
// — development tool (end of round 4): the SQ counters say the rescoring kernel is bound by vector issue (profiles/r03_pmc, profiles/r04_pmc_c5,
// profiles/r04_pmc_lanes_per_kernel.txt) and its loop costs 85 vector instructions per 16 columns; this file compares the loop with a cheaper variant.
// Result on the MI355X (profiles/r04_rescore_loop_bench.log): both run at ~600 G columns/s on random pairs — memory bound; the loop is not where
// the product kernel's instructions go.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rescore_loop_bench.hip -o tools/rescore_loop_bench
//   run:   tools/rescore_loop_bench [pairs (default 2^22)] [mean overlap (default 115)]
// Both kernels score the same synthetic pairs (random protein letters, overlaps of random length at random places of one buffer, so
// that the loads behave like the product's: every lane its own two streams):
//   A  the product's loop: 16 columns per step, index (q << 7) | t into a 123 x 128 table, bytes taken out with bit-field extracts;
//   B  a 123 x 256 table (31.5 KB of LDS): ONE v_perm_b32 interleaves two query and two target bytes into two 16-bit table indices
//      (q << 8 | t), so a column costs a half-word select, the lookup and a share of an add3 instead of two extracts and a shift-or.
// Correctness (always): both sums against a host loop.  Timing: HIP events, columns per second.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Pair { uint64_t qOff, tOff; uint32_t len; uint32_t pad; };

// ---- A: as scoreDiagonal<false, 1> (columns beyond `len` are blanked: byte 0 in both words, entry [0][0] = 0) ----
__global__ __launch_bounds__(256) void loopA(const char *__restrict__ seq, const Pair *__restrict__ pairs, uint32_t n, const signed char *__restrict__ mat, int *__restrict__ out) {
    __shared__ signed char smat[123 * 128];
    for (int i = threadIdx.x; i < 123 * 128; i += 256) smat[i] = (i & 127) < 123 ? mat[(i >> 7) * 123 + (i & 127)] : (signed char) 0;
    __syncthreads();
    if (threadIdx.x == 0) smat[0] = 0;
    __syncthreads();
    for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
        const Pair p = pairs[w];
        const char *q = seq + p.qOff, *t = seq + p.tOff;
        int s = 0;
        uint32_t qn[4], tn[4];
        __builtin_memcpy(qn, q, 16); __builtin_memcpy(tn, t, 16);
        for (uint32_t c = 0; c < p.len; c += 16) {
            uint32_t qw[4], tw[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { qw[k] = qn[k]; tw[k] = tn[k]; }
            if (c + 16 < p.len) { __builtin_memcpy(qn, q + c + 16, 16); __builtin_memcpy(tn, t + c + 16, 16); }
            const uint32_t nCol = min(16u, p.len - c);
            if (nCol < 16u) {
#pragma unroll
                for (int k = 0; k < 4; k++) { const int nb = (int) nCol - 4 * k; const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u)); qw[k] &= m; tw[k] &= m; }
            }
#pragma unroll
            for (unsigned j = 0; j < 16; j++) {
                const unsigned a = (qw[j >> 2] >> (8 * (j & 3))) & 0xFFu, b = (tw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                s += (int) smat[(a << 7) | b];
            }
        }
        out[w] = s;
    }
}

// ---- B: 256-byte rows, two table indices per permute ----
__global__ __launch_bounds__(256) void loopB(const char *__restrict__ seq, const Pair *__restrict__ pairs, uint32_t n, const signed char *__restrict__ mat, int *__restrict__ out) {
    __shared__ signed char smat[123 * 256];
    for (int i = threadIdx.x; i < 123 * 256; i += 256) smat[i] = (i & 255) < 123 ? mat[(i >> 8) * 123 + (i & 255)] : (signed char) 0;
    __syncthreads();
    if (threadIdx.x == 0) smat[0] = 0;
    __syncthreads();
    for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
        const Pair p = pairs[w];
        const char *q = seq + p.qOff, *t = seq + p.tOff;
        int s = 0;
        uint32_t qn[4], tn[4];
        __builtin_memcpy(qn, q, 16); __builtin_memcpy(tn, t, 16);
        for (uint32_t c = 0; c < p.len; c += 16) {
            uint32_t qw[4], tw[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { qw[k] = qn[k]; tw[k] = tn[k]; }
            if (c + 16 < p.len) { __builtin_memcpy(qn, q + c + 16, 16); __builtin_memcpy(tn, t + c + 16, 16); }
            const uint32_t nCol = min(16u, p.len - c);
            if (nCol < 16u) {
#pragma unroll
                for (int k = 0; k < 4; k++) { const int nb = (int) nCol - 4 * k; const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u)); qw[k] &= m; tw[k] &= m; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // v_perm_b32 D, S0, S1, sel: selector bytes 0-3 take bytes of S1, 4-7 bytes of S0.  lo = (t0, q0, t1, q1), hi = (t2, q2, t3, q3)
                const uint32_t lo = __builtin_amdgcn_perm(qw[k], tw[k], 0x05010400u), hi = __builtin_amdgcn_perm(qw[k], tw[k], 0x07030602u);
                s += (int) smat[lo & 0xFFFFu] + (int) smat[lo >> 16] + (int) smat[hi & 0xFFFFu] + (int) smat[hi >> 16];
            }
        }
        out[w] = s;
    }
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t) atoll(argv[1]) : (1u << 22);
    const uint32_t meanLen = argc > 2 ? (uint32_t) atoi(argv[2]) : 115;
    const char *aa = "ACDEFGHIKLMNPQRSTVWY";
    const size_t seqBytes = (size_t) 1 << 30;
    std::mt19937_64 g(1);
    std::vector<char> seq(seqBytes + 64, 0);
    for (size_t i = 0; i < seqBytes; i++) seq[i] = aa[g() % 20];
    std::vector<signed char> mat(123 * 123, 0);
    for (int a = 0; a < 123; a++) for (int b = 0; b < 123; b++) mat[a * 123 + b] = (a == b) ? 5 : (signed char) ((int) ((a * 31 + b * 17) % 7) - 4);
    mat[0] = 0;
    std::vector<Pair> pairs(n);
    for (uint32_t i = 0; i < n; i++) {
        Pair p; p.len = 16 + (uint32_t) (g() % (2 * meanLen - 31)); p.pad = 0;
        p.qOff = g() % (seqBytes - 1024); p.tOff = g() % (seqBytes - 1024);
        pairs[i] = p;
    }
    char *dSeq; Pair *dPairs; signed char *dMat; int *dOutA, *dOutB;
    CK(hipMalloc(&dSeq, seqBytes + 64)); CK(hipMalloc(&dPairs, (size_t) n * sizeof(Pair))); CK(hipMalloc(&dMat, 123 * 123)); CK(hipMalloc(&dOutA, (size_t) n * 4)); CK(hipMalloc(&dOutB, (size_t) n * 4));
    CK(hipMemcpy(dSeq, seq.data(), seqBytes + 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dPairs, pairs.data(), (size_t) n * sizeof(Pair), hipMemcpyHostToDevice));
    CK(hipMemcpy(dMat, mat.data(), 123 * 123, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint64_t cols = 0; for (auto &p : pairs) cols += p.len;
    for (int variant = 0; variant < 2; variant++) {
        int *dOut = variant ? dOutB : dOutA;
        for (int blocksPerCU : {4, 5, 8}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, 0));
                if (variant) hipLaunchKernelGGL(loopB, dim3(cus * blocksPerCU), dim3(256), 0, 0, dSeq, dPairs, n, dMat, dOut);
                else hipLaunchKernelGGL(loopA, dim3(cus * blocksPerCU), dim3(256), 0, 0, dSeq, dPairs, n, dMat, dOut);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            printf("variant %c, %d workgroups per CU: %.3f ms, %.1f G columns/s\n", variant ? 'B' : 'A', blocksPerCU, best, cols / (best * 1e-3) / 1e9);
        }
    }
    std::vector<int> a(n), b(n);
    CK(hipMemcpy(a.data(), dOutA, (size_t) n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dOutB, (size_t) n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t i = 0; i < n; i += 97) {
        int s = 0; for (uint32_t c = 0; c < pairs[i].len; c++) s += mat[(int) seq[pairs[i].qOff + c] * 123 + (int) seq[pairs[i].tOff + c]];
        if (s != a[i] || s != b[i]) { if (bad < 5) fprintf(stderr, "pair %u: host %d, A %d, B %d\n", i, s, a[i], b[i]); bad++; }
    }
    printf("%s (%zu of %u checked pairs differ)\n", bad ? "MISMATCH" : "sums agree with the host loop", bad, (n + 96) / 97);
    return bad ? 1 : 0;
}
