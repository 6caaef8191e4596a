// Device-wide exclusive scan (see device_utils.hpp).  Product code.
#include "device_utils.hpp"

namespace plasship {

template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void scanReduceKernel(const T *__restrict__ in, uint64_t *__restrict__ partial, size_t n) {
    __shared__ unsigned long long wsum[SCAN_BLOCK / WAVE];
    const size_t base = (size_t) blockIdx.x * SCAN_TILE;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + (size_t) k * SCAN_BLOCK + threadIdx.x;   // strided: coalesced
        if (i < n) s += (unsigned long long) in[i];
    }
    s = waveReduceSumU64(s);
    if (laneId() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < SCAN_BLOCK / WAVE; w++) t += wsum[w];
        partial[blockIdx.x] = t;
    }
}

// Exclusive scan of one tile per block; blockOff (nullable) supplies the scanned tile offsets.
// If writeTotal, the block that owns the end writes out[n] = grand total.
template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void scanTileKernel(const T *in, uint64_t *out, size_t n,
                                                             const uint64_t *__restrict__ blockOff, int writeTotal) {
    __shared__ unsigned long long wsum[SCAN_BLOCK / WAVE];
    const size_t base = (size_t) blockIdx.x * SCAN_TILE + (size_t) threadIdx.x * SCAN_ITEMS;   // thread-contiguous
    unsigned long long v[SCAN_ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + k;
        v[k] = (i < n) ? (unsigned long long) in[i] : 0ull;
        s += v[k];
    }
    unsigned long long incl = waveInclusiveScanU64(s);
    const int w = threadIdx.x >> 6;
    if (laneId() == 63) wsum[w] = incl;
    __syncthreads();
    unsigned long long woff = 0, tileTotal = 0;
#pragma unroll
    for (int j = 0; j < SCAN_BLOCK / WAVE; j++) { if (j < w) woff += wsum[j]; tileTotal += wsum[j]; }
    unsigned long long run = (blockOff ? blockOff[blockIdx.x] : 0ull) + woff + (incl - s);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        size_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (writeTotal && threadIdx.x == 0) {
        const size_t lastBlock = (n == 0) ? 0 : (n - 1) / SCAN_TILE;
        if (blockIdx.x == lastBlock) out[n] = (blockOff ? blockOff[blockIdx.x] : 0ull) + tileTotal;
    }
}

size_t exclusiveScanTmpBytes(size_t n) {
    size_t bytes = 0;
    while (n > (size_t) SCAN_TILE) { size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE; bytes += (nb + 1) * sizeof(uint64_t); n = nb; }
    return bytes + 64;
}

template <typename T>
static int scanImpl(hipStream_t stream, const T *d_in, uint64_t *d_out, size_t n, uint64_t *tmp) {
    if (n <= (size_t) SCAN_TILE) {
        hipLaunchKernelGGL(scanTileKernel<T>, dim3(1), dim3(SCAN_BLOCK), 0, stream, d_in, d_out, n, (const uint64_t *) nullptr, 1);
        return 0;
    }
    size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scanReduceKernel<T>, dim3((unsigned) nb), dim3(SCAN_BLOCK), 0, stream, d_in, tmp, n);
    // NOTE: reduce reads strided within the tile while the downsweep reads thread-contiguous; both
    // cover exactly [tile*SCAN_TILE, (tile+1)*SCAN_TILE) so the per-tile sums agree.
    int rc = scanImpl<uint64_t>(stream, tmp, tmp, nb, tmp + nb + 1);
    if (rc) return rc;
    hipLaunchKernelGGL(scanTileKernel<T>, dim3((unsigned) nb), dim3(SCAN_BLOCK), 0, stream, d_in, d_out, n, (const uint64_t *) tmp, 1);
    return 0;
}

int exclusiveScanU32(hipStream_t stream, const uint32_t *d_in, uint64_t *d_out, size_t n, void *d_tmp, size_t tmpBytes) {
    if (tmpBytes < exclusiveScanTmpBytes(n)) return -1;
    return scanImpl<uint32_t>(stream, d_in, d_out, n, (uint64_t *) d_tmp);
}
int exclusiveScanU64(hipStream_t stream, const uint64_t *d_in, uint64_t *d_out, size_t n, void *d_tmp, size_t tmpBytes) {
    if (tmpBytes < exclusiveScanTmpBytes(n)) return -1;
    return scanImpl<uint64_t>(stream, d_in, d_out, n, (uint64_t *) d_tmp);
}

}  // namespace plasship
