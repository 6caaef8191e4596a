// Device utilities shared by the plasship kernels: wave64 reductions/scans and a device-wide
// exclusive scan.  gfx950 only: wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace plasship {

constexpr int WAVE = 64;

__device__ __forceinline__ int laneId() { return (int) (threadIdx.x & 63); }

// ---- DPP row operations (a row = 16 consecutive lanes): cross-lane moves on the VALU, no LDS crossbar ----
// ctrl: quad_perm 0x00-0xFF, row_ror:n 0x120+n, row_mirror 0x140, row_half_mirror 0x141
template <int CTRL> __device__ __forceinline__ uint32_t dppMov(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ unsigned long long dppMov64(unsigned long long v) {
    const uint32_t lo = dppMov<CTRL>((uint32_t) v), hi = dppMov<CTRL>((uint32_t) (v >> 32));
    return ((unsigned long long) hi << 32) | lo;
}
// all-reduce over each row of 16 lanes: xor 1, xor 2 (quad_perm), then mirror inside 8, mirror inside 16
__device__ __forceinline__ int rowSum16(int v) {
    v += (int) dppMov<0xB1>((uint32_t) v); v += (int) dppMov<0x4E>((uint32_t) v);
    v += (int) dppMov<0x141>((uint32_t) v); v += (int) dppMov<0x140>((uint32_t) v);
    return v;
}
__device__ __forceinline__ unsigned long long rowMax16(unsigned long long v) {
    unsigned long long o;
    o = dppMov64<0xB1>(v); v = o > v ? o : v;
    o = dppMov64<0x4E>(v); v = o > v ? o : v;
    o = dppMov64<0x141>(v); v = o > v ? o : v;
    o = dppMov64<0x140>(v); v = o > v ? o : v;
    return v;
}
// number of lanes in the own row of 16 whose value is smaller than `mine` (own lane excluded by the strict compare
// when other == mine); `other` is the value to rotate: the own value, or another row's value fetched by a shuffle
template <int N> struct RowCountLess {
    static __device__ __forceinline__ uint32_t run(uint32_t other, uint32_t mine) {
        return ((dppMov<0x120 + N>(other) < mine) ? 1u : 0u) + RowCountLess<N - 1>::run(other, mine);
    }
};
template <> struct RowCountLess<0> { static __device__ __forceinline__ uint32_t run(uint32_t, uint32_t) { return 0u; } };
__device__ __forceinline__ uint32_t rowCountLess16(uint32_t other, uint32_t mine, bool includeUnrotated) {
    return RowCountLess<15>::run(other, mine) + ((includeUnrotated && other < mine) ? 1u : 0u);
}

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
