// plasship: XXH64 of one little-endian u64 and the 16-bit window score derived from it.  Product code (kmermatch.hip); plain C++ as
// well, so that tests/test_host.py can compile it with g++ and compare the two functions without a GPU.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define PLASSHIP_HD __host__ __device__ __forceinline__
#else
#define PLASSHIP_HD static inline
#endif
namespace plasship {
// ---- XXH64 of one little-endian u64 (xxhash 0.8.0, call site kmermatcher.cpp:33-38) ---------------------
PLASSHIP_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
PLASSHIP_HD uint64_t xxh64U64(uint64_t v, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = rotl64(v * P2, 31) * P1;
    h ^= k1;
    h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}
// The 16-bit SCORE of a window is all the selection uses of its hash (kmermatcher.cpp:184-208 store it in the unsigned short `score`):
// bits 0-15 of the last product xor its bits 32-47.  Those need one 32 x 32 -> 64 product and the low 16 bits of the two cross terms —
// 16 x 16-bit products on the full-rate 24-bit multiplier — instead of a whole 64-bit multiplication (round 4: 13 instead of 15
// quarter-rate multiplier passes per window).  tests/test_host.py compiles this file with g++ and compares the two functions.
PLASSHIP_HD uint32_t mulLow16(uint32_t a, uint32_t c16) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a & 0xFFFFu, c16);
#else
    return (a & 0xFFFFu) * c16;
#endif
}
PLASSHIP_HD uint32_t xxh64Score16(uint64_t v, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    uint64_t h = seed + P5 + 8;
    h ^= rotl64(v * P2, 31) * P1;
    h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33; h *= P2; h ^= h >> 29;
    const uint32_t lo = (uint32_t) h, hi = (uint32_t) (h >> 32);
    const uint64_t ll = (uint64_t) lo * 0x9E3779F9ULL;                                       // P3 = 0x165667B1'9E3779F9
    const uint32_t cross = mulLow16(lo, 0x67B1u) + mulLow16(hi, 0x79F9u);
    return ((uint32_t) ll ^ ((uint32_t) (ll >> 32) + cross)) & 0xFFFFu;
}
}  // namespace plasship
