// Device utilities shared by the plasship kernels: wave64 reductions/scans and a device-wide
// exclusive scan.  gfx950 only: wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace plasship {

constexpr int WAVE = 64;

__device__ __forceinline__ int laneId() { return (int) (threadIdx.x & 63); }

__device__ __forceinline__ int waveReduceSum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long waveReduceSumU64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int waveReduceMax(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ unsigned waveInclusiveScan(unsigned v) {
    const int l = laneId();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}
__device__ __forceinline__ unsigned long long waveInclusiveScanU64(unsigned long long v) {
    const int l = laneId();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned long long t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}

// ---- device-wide exclusive scan of uint32 counts into uint64 offsets ------------------------------
// out[i] = sum_{j<i} in[j], out[n] = total.  Three-kernel scheme (reduce / scan partials / downsweep);
// the partial array is scanned recursively.  Traffic 12 B/element read + 8 B written: negligible next
// to the record arrays it indexes.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                       // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;  // 2048 elements per block

int exclusiveScanU32(hipStream_t stream, const uint32_t *d_in, uint64_t *d_out, size_t n, void *d_tmp, size_t tmpBytes);
size_t exclusiveScanTmpBytes(size_t n);
// same, input already uint64 (in place allowed: d_in == d_out)
int exclusiveScanU64(hipStream_t stream, const uint64_t *d_in, uint64_t *d_out, size_t n, void *d_tmp, size_t tmpBytes);

}  // namespace plasship
