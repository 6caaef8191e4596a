// plasship_synth core: the synthetic community and read model (include/plasship_synth.h) as plain functions that compile both for
// the GPU (synth.hip: one thread per base / per read) and for the host — tests regenerate the very same reads on the CPU
// (the test tools' `synthreads` module) to pin large GPU runs against CPU-oracle checksums made in a container without a
// GPU.  MEASUREMENT INFRASTRUCTURE — no reference counterpart.  Everything is integer arithmetic on a 64-bit mix of
// (seed, indices), except one float multiply-add for the insert length (single IEEE operations, -ffp-contract=off) and exp() for the
// abundances, which runs on the host in both cases.
#pragma once
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
#ifdef __HIPCC__
#define PH_HD __host__ __device__ __forceinline__
#else
#define PH_HD inline
#endif

namespace plasship {

PH_HD uint64_t synthMix(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t x = seed + a * 0x9E3779B97F4A7C15ULL + b * 0xC2B2AE3D27D4EB4FULL;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
    return x;
}

struct SynthGenome {
    const uint64_t *geneStart;      // [nGenes + 1] first base of gene j in the concatenated genomes (its spacer follows the gene)
    const uint32_t *geneCodons;     // [nGenes] sense codons between ATG and the stop; bit 31 = gene lies on the reverse strand
    uint64_t nGenes, totalBases, seed;
};

// codon index = 16 b0 + 4 b1 + b2 with A0 C1 G2 T3; stops TAA 48, TAG 50, TGA 56
PH_HD uint32_t senseCodon(uint32_t r) {       // r in [0, 61): the r-th non-stop codon
    uint32_t c = r;
    if (c >= 48) c++;                // skip TAA
    if (c >= 50) c++;                // skip TAG
    if (c >= 56) c++;                // skip TGA
    return c;
}

PH_HD char synthGenomeBase(const SynthGenome &g, uint64_t x) {
    uint64_t lo = 0, hi = g.nGenes;                               // last gene with geneStart <= x
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (g.geneStart[mid] <= x) lo = mid; else hi = mid; }
    const uint64_t j = lo;
    const uint32_t gc = g.geneCodons[j];
    const uint32_t n = gc & 0x7FFFFFFFu; const bool rev = (gc >> 31) != 0;
    const uint64_t off = x - g.geneStart[j];
    const uint64_t geneLen = 3ull * ((uint64_t) n + 2);
    uint32_t b;
    if (off < geneLen) {
        const uint64_t o = rev ? geneLen - 1 - off : off;
        const uint32_t c = (uint32_t) (o / 3), p = (uint32_t) (o % 3);
        uint32_t cod;
        if (c == 0) cod = 14;                                      // ATG
        else if (c == n + 1) { const uint32_t s = (uint32_t) (synthMix(g.seed, j, 0xFFFFFFFFull) % 3); cod = s == 0 ? 48u : (s == 1 ? 50u : 56u); }
        else cod = senseCodon((uint32_t) (synthMix(g.seed, j, c) % 61));
        b = (cod >> (2 * (2 - p))) & 3u;
        if (rev) b = 3u - b;                                       // A<->T, C<->G in this code
    } else b = (uint32_t) (synthMix(g.seed ^ 0x5BD1E995ull, x, 1) & 3u);
    return b == 0 ? 'A' : (b == 1 ? 'C' : (b == 2 ? 'G' : 'T'));
}

struct SynthReads {
    const char *genome; const uint64_t *genomeStart;   // [nGenomes + 1]
    const uint64_t *cum;                               // [nGenomes] inclusive cumulative pair probability scaled to 2^63
    uint32_t nGenomes, readLen, insertMin; float insertMean, insertSd; uint32_t errThresh;   // error probability * 2^30
    uint64_t nPairs, seed;
    char *out; uint64_t *off; uint32_t *len, *key;
};

PH_HD char synthCompLetter(char c) { return c == 'A' ? 'T' : (c == 'C' ? 'G' : (c == 'G' ? 'C' : 'A')); }

// entry of read r (mate r & 1 of pair r >> 1): readLen letters, '\n', '\0'; also its index row
PH_HD void synthRead(const SynthReads &a, uint64_t r) {
    const uint64_t nReads = 2 * a.nPairs;
    const uint32_t entry = a.readLen + 2;
    const uint64_t i = r >> 1; const uint32_t mate = (uint32_t) (r & 1);
    const uint64_t h0 = synthMix(a.seed ^ 0xA24BAED4963EE407ull, i, 0);
    const uint64_t u = h0 >> 1;
    uint32_t lo = 0, hi = a.nGenomes - 1;                          // first genome with cum > u
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.cum[mid] > u) hi = mid; else lo = mid + 1; }
    const uint64_t g0 = a.genomeStart[lo], gl = a.genomeStart[lo + 1] - g0;
    // insert length: mean + sd * z, z = sum of twelve uniforms - 6 (Irwin-Hall), from 16-bit slices of three hashes
    int64_t s = 0;
    for (int q = 0; q < 3; q++) { const uint64_t h = synthMix(a.seed ^ 0x9FB21C651E98DF25ull, i, 1 + q); s += (int64_t) (h & 0xFFFF) + (int64_t) ((h >> 16) & 0xFFFF) + (int64_t) ((h >> 32) & 0xFFFF) + (int64_t) (h >> 48); }
    const float z = (float) (s - 393210) * (1.0f / 65536.0f);                    // twelve values in [0, 65535]: mean 393210, sd 65536
    // two separately rounded IEEE operations on both sides: the intrinsics are never contracted into an fma on the GPU, the host side
    // is compiled with -ffp-contract=off (oracle/Makefile)
#ifdef __HIP_DEVICE_COMPILE__
    int64_t ins = (int64_t) __fadd_rn(a.insertMean, __fmul_rn(a.insertSd, z));
#else
    volatile float scaled = a.insertSd * z;
    int64_t ins = (int64_t) (a.insertMean + scaled);
#endif
    if (ins < (int64_t) a.insertMin) ins = a.insertMin;
    if (ins < (int64_t) a.readLen) ins = a.readLen;
    if ((uint64_t) ins + 2 > gl) ins = (int64_t) gl - 2;
    const uint64_t pos = g0 + synthMix(a.seed ^ 0xD6E8FEB86659FD93ull, i, 5) % (gl - (uint64_t) ins - 1);
    const bool flip = (h0 & 1) != 0;
    const bool rev = (mate == 0) ? flip : !flip;                   // mate 0: forward end unless flipped; mate 1: the other end
    char *d = a.out + r * entry;
    uint64_t he = 0;
    for (uint32_t t = 0; t < a.readLen; t++) {
        char c = rev ? synthCompLetter(a.genome[pos + (uint64_t) ins - 1 - t]) : a.genome[pos + t];
        if ((t & 1) == 0) he = synthMix(a.seed ^ 0x2545F4914F6CDD1Dull, r, t >> 1);
        const uint32_t e = (t & 1) ? (uint32_t) (he >> 32) : (uint32_t) he;
        if ((e & 0x3FFFFFFFu) < a.errThresh) { const uint32_t b = e >> 30; c = b == 0 ? 'A' : (b == 1 ? 'C' : (b == 2 ? 'G' : 'T')); }
        d[t] = c;
    }
    d[a.readLen] = '\n'; d[a.readLen + 1] = '\0';
    a.off[r] = r * entry; a.len[r] = a.readLen; a.key[r] = (uint32_t) r;
    if (r == 0) a.off[nReads] = nReads * entry;
}

// the community and the gene layout (host; a few hundred thousand entries at most)
struct SynthCommunity {
    std::vector<uint64_t> gStart, geneStart, cum; std::vector<uint32_t> geneCodons; std::vector<double> abund; double wsum = 0; uint64_t total = 0;
    void build(uint64_t seed, uint32_t G, uint64_t genomeMinLen, uint64_t genomeMaxLen, float abundanceSigma) {
        uint64_t rs = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ull;
        auto next = [&]() { rs += 0x9E3779B97F4A7C15ULL; uint64_t x = rs; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; };
        auto unif = [&]() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); };
        gStart.assign(G + 1, 0); geneStart.clear(); geneCodons.clear(); abund.assign(G, 0.0); cum.assign(G, 0);
        std::vector<double> weight(G);
        total = 0;
        for (uint32_t g = 0; g < G; g++) {
            const uint64_t want = genomeMinLen + (uint64_t) (unif() * (double) (genomeMaxLen - genomeMinLen));
            double z = -6.0; for (int q = 0; q < 12; q++) z += unif();
            abund[g] = std::exp((double) abundanceSigma * z);
            gStart[g] = total;
            uint64_t len = 0;
            while (len < want) {
                const uint64_t r = next();
                const uint32_t n = 300 + (uint32_t) (r % 1201); const uint32_t revBit = (uint32_t) ((r >> 32) & 1); const uint32_t spacer = 50 + (uint32_t) ((r >> 33) % 151);
                geneStart.push_back(total + len); geneCodons.push_back(n | (revBit << 31));
                len += 3ull * (n + 2) + spacer;
            }
            total += len;
            weight[g] = abund[g] * (double) len;
        }
        gStart[G] = total; geneStart.push_back(total);
        wsum = 0; for (double w : weight) wsum += w;
        double acc = 0;
        for (uint32_t g = 0; g < G; g++) { acc += weight[g] / wsum; const double v = std::min(acc, 1.0) * 9223372036854775808.0; cum[g] = v >= 9223372036854775807.0 ? 0x7FFFFFFFFFFFFFFFull : (uint64_t) v; }
        cum[G - 1] = 0x8000000000000000ull;                                // u < 2^63 always lands somewhere
    }
};

}  // namespace plasship
