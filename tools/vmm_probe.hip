// GPU box probe (round 6): what does the arena's one big hipMalloc cost, is the cost proportional to the size, does the virtual-memory API
// (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess) map the same memory in pieces at a comparable rate, and can a kernel on
// already-mapped memory run WHILE another host thread maps more?  (plass-hip assemble-chain waits 6 s for hipMalloc(271 GB): profiles/r06_calls/call3.)
//   hipcc --offload-arch=gfx950 -O2 tools/vmm_probe.hip -o tools/vmm_probe && tools/vmm_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(unsigned long long *p, size_t n, unsigned long long v) { for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v + i; }
int main() {
    size_t fr = 0, tt = 0; CK(hipMemGetInfo(&fr, &tt)); printf("free %.1f GB of %.1f GB\n", fr / 1e9, tt / 1e9);
    for (size_t gb : {8ul, 64ul, 128ul}) {
        void *p = nullptr; double t0 = now(); CK(hipMalloc(&p, gb << 30)); double t1 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (unsigned long long *) p, (gb << 30) / 8, 1ull); CK(hipDeviceSynchronize()); double t2 = now();
        CK(hipFree(p)); double t3 = now();
        printf("hipMalloc %3zu GB: %.3f s (%.1f GB/s), first touch by a kernel %.3f s, hipFree %.3f s\n", gb, t1 - t0, gb * 1.0737 / (t1 - t0), t2 - t1, t3 - t2);
    }
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended)); printf("VMM granularity %zu\n", gran);
    const size_t total = 128ull << 30, chunk = 2ull << 30; const size_t nch = total / chunk;
    void *va = nullptr; double t0 = now(); CK(hipMemAddressReserve(&va, total, 0, nullptr, 0)); printf("reserve 128 GB of addresses: %.4f s\n", now() - t0);
    std::vector<hipMemGenericAllocationHandle_t> h(nch);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    // map the first 16 GB, then run a kernel on them in a loop while a second thread maps the rest
    t0 = now();
    for (size_t i = 0; i < 8; i++) { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemMap((char *) va + i * chunk, chunk, 0, h[i], 0)); CK(hipMemSetAccess((char *) va + i * chunk, chunk, &acc, 1)); }
    printf("create + map + access 16 GB in 2 GB chunks: %.3f s (%.1f GB/s)\n", now() - t0, 16 * 1.0737 / (now() - t0));
    std::atomic<int> done(0); double tMap = 0; int mapErr = 0;
    std::thread mapper([&] { (void) hipSetDevice(0); const double a = now();
        for (size_t i = 8; i < nch; i++) { if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess || hipMemMap((char *) va + i * chunk, chunk, 0, h[i], 0) != hipSuccess || hipMemSetAccess((char *) va + i * chunk, chunk, &acc, 1) != hipSuccess) { mapErr = 1; break; } }
        tMap = now() - a; done = 1; });
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int launches = 0; double worst = 0; const double k0 = now();
    while (!done) { const double a = now(); hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, (unsigned long long *) va, (16ull << 30) / 8, (unsigned long long) launches); CK(hipStreamSynchronize(st)); const double d = now() - a; if (d > worst) worst = d; launches++; }
    mapper.join();
    printf("second thread mapped %zu GB in %.3f s (%.1f GB/s, error %d) while %d kernels over the first 16 GB ran: %.4f s each on average, worst %.4f s\n", (nch - 8) * 2, tMap, (nch - 8) * 2 * 1.0737 / tMap, mapErr, launches, (now() - k0) / (launches ? launches : 1), worst);
    { const double a = now(); hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, (unsigned long long *) va, total / 8, 7ull); CK(hipStreamSynchronize(st)); printf("one kernel over all 128 GB of the mapped range: %.3f s\n", now() - a); }
    t0 = now();
    for (size_t i = 0; i < nch; i++) { (void) hipMemUnmap((char *) va + i * chunk, chunk); (void) hipMemRelease(h[i]); }
    (void) hipMemAddressFree(va, total); printf("unmap + release: %.3f s\n", now() - t0);
    // the same hipMalloc again, after memory has gone back once (is the second one cheaper?)
    { void *p = nullptr; const double a = now(); CK(hipMalloc(&p, 128ull << 30)); printf("hipMalloc 128 GB again: %.3f s\n", now() - a); CK(hipFree(p)); }
    return 0;
}
