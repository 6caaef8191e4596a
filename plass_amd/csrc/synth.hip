// plasship_synth: seeded synthetic read pairs generated in HBM (include/plasship_synth.h).
// MEASUREMENT INFRASTRUCTURE — no reference counterpart; the model is SURVEY.md section 8d's (gene-dense genomes, log-normal
// community, paired 2 x 150 nt reads with substitution errors).  Everything is a pure function of (seed, indices).
#include "common.hpp"
#include "device_utils.hpp"
#include "../../include/plasship_synth.h"
#include <algorithm>
#include <cmath>
#include <memory>
#include <cstring>

namespace plasship {

__host__ __device__ __forceinline__ uint64_t synthMix(uint64_t seed, uint64_t a, uint64_t b) {
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
__device__ __forceinline__ uint32_t senseCodon(uint32_t r) {       // r in [0, 61): the r-th non-stop codon
    uint32_t c = r;
    if (c >= 48) c++;                // skip TAA
    if (c >= 50) c++;                // skip TAG
    if (c >= 56) c++;                // skip TGA
    return c;
}

__global__ __launch_bounds__(256) void synthGenomeKernel(SynthGenome g, char *__restrict__ out) {
    const char LET[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t x = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; x < g.totalBases; x += (uint64_t) gridDim.x * blockDim.x) {
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
        out[x] = LET[b];
    }
}

struct SynthReads {
    const char *genome; const uint64_t *genomeStart;   // [nGenomes + 1]
    const uint64_t *cum;                               // [nGenomes] inclusive cumulative pair probability scaled to 2^63
    uint32_t nGenomes, readLen, insertMin; float insertMean, insertSd; uint32_t errThresh;   // error probability * 2^30
    uint64_t nPairs, seed;
    char *out; uint64_t *off; uint32_t *len, *key;
};

__device__ __forceinline__ char compLetter(char c) { return c == 'A' ? 'T' : (c == 'C' ? 'G' : (c == 'G' ? 'C' : 'A')); }

__global__ __launch_bounds__(256) void synthReadsKernel(SynthReads a) {
    const uint64_t nReads = 2 * a.nPairs;
    const uint32_t entry = a.readLen + 2;
    const char LET[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < nReads; r += (uint64_t) gridDim.x * blockDim.x) {
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
        int64_t ins = (int64_t) (a.insertMean + a.insertSd * z);
        if (ins < (int64_t) a.insertMin) ins = a.insertMin;
        if (ins < (int64_t) a.readLen) ins = a.readLen;
        if ((uint64_t) ins + 2 > gl) ins = (int64_t) gl - 2;
        const uint64_t pos = g0 + synthMix(a.seed ^ 0xD6E8FEB86659FD93ull, i, 5) % (gl - (uint64_t) ins - 1);
        const bool flip = (h0 & 1) != 0;
        const bool rev = (mate == 0) ? flip : !flip;                   // mate 0: forward end unless flipped; mate 1: the other end
        char *d = a.out + r * entry;
        uint64_t he = 0;
        for (uint32_t t = 0; t < a.readLen; t++) {
            char c = rev ? compLetter(a.genome[pos + (uint64_t) ins - 1 - t]) : a.genome[pos + t];
            if ((t & 1) == 0) he = synthMix(a.seed ^ 0x2545F4914F6CDD1Dull, r, t >> 1);
            const uint32_t e = (t & 1) ? (uint32_t) (he >> 32) : (uint32_t) he;
            if ((e & 0x3FFFFFFFu) < a.errThresh) c = LET[e >> 30];
            d[t] = c;
        }
        d[a.readLen] = '\n'; d[a.readLen + 1] = '\0';
        a.off[r] = r * entry; a.len[r] = a.readLen; a.key[r] = (uint32_t) r;
        if (r == 0) a.off[nReads] = nReads * entry;
    }
}

}  // namespace plasship
using namespace plasship;

extern "C" int plasship_synth_read_pairs(plasship_ctx *ctx, const plasship_synth_params *par, plasship_seqdb **out_reads, plasship_synth_stats *stats) {
    if (!ctx || !par || !out_reads) { setError("plasship_synth_read_pairs: bad argument"); return PLASSHIP_ERR_ARG; }
    if (par->n_genomes == 0 || par->genome_min_len < 1000 || par->genome_max_len < par->genome_min_len || par->read_len < 8 || par->read_len > 100000 ||
        2 * par->n_pairs >= 0xFFFFFFFFull || par->error_rate < 0 || par->error_rate >= 1 || par->abundance_sigma < 0 ||
        par->genome_min_len < 4ull * (uint64_t) (par->insert_mean + 8 * par->insert_sd + par->read_len)) { setError("plasship_synth_read_pairs: bad parameter"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    // ---- community and gene layout (host, tiny) ----
    uint64_t rs = par->seed * 0x9E3779B97F4A7C15ULL + 0x1234567ull;
    auto next = [&]() { rs += 0x9E3779B97F4A7C15ULL; uint64_t x = rs; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31; return x; };
    auto unif = [&]() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); };
    const uint32_t G = par->n_genomes;
    std::vector<uint64_t> gStart(G + 1, 0), geneStart; std::vector<uint32_t> geneCodons; std::vector<double> weight(G), abund(G);
    uint64_t total = 0;
    for (uint32_t g = 0; g < G; g++) {
        const uint64_t want = par->genome_min_len + (uint64_t) (unif() * (double) (par->genome_max_len - par->genome_min_len));
        double z = -6.0; for (int q = 0; q < 12; q++) z += unif();
        abund[g] = std::exp((double) par->abundance_sigma * z);
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
    double wsum = 0; for (double w : weight) wsum += w;
    std::vector<uint64_t> cum(G); double acc = 0;
    for (uint32_t g = 0; g < G; g++) { acc += weight[g] / wsum; const double v = std::min(acc, 1.0) * 9223372036854775808.0; cum[g] = v >= 9223372036854775807.0 ? 0x7FFFFFFFFFFFFFFFull : (uint64_t) v; }
    cum[G - 1] = 0x8000000000000000ull;                                // u < 2^63 always lands somewhere
    const uint64_t nGenes = geneCodons.size();
    // ---- device ----
    DevBuf dGenome, dGeneStart, dGeneCodons, dGStart, dCum;
    if (dGenome.alloc(total + 64) != hipSuccess || dGeneStart.alloc((nGenes + 1) * 8) != hipSuccess || dGeneCodons.alloc(nGenes * 4) != hipSuccess ||
        dGStart.alloc((G + 1) * 8) != hipSuccess || dCum.alloc((size_t) G * 8) != hipSuccess) { setError("plasship_synth_read_pairs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemcpyAsync(dGeneStart.p, geneStart.data(), (nGenes + 1) * 8, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemcpyAsync(dGeneCodons.p, geneCodons.data(), nGenes * 4, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemcpyAsync(dGStart.p, gStart.data(), (G + 1) * 8, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemcpyAsync(dCum.p, cum.data(), (size_t) G * 8, hipMemcpyHostToDevice, st));
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    SynthGenome sg; sg.geneStart = dGeneStart.as<uint64_t>(); sg.geneCodons = dGeneCodons.as<uint32_t>(); sg.nGenes = nGenes; sg.totalBases = total; sg.seed = par->seed;
    hipLaunchKernelGGL(synthGenomeKernel, dim3((unsigned) std::min<uint64_t>((total + 255) / 256, (uint64_t) ctx->numCU * 64)), dim3(256), 0, st, sg, dGenome.as<char>());
    const uint64_t nReads = 2 * par->n_pairs; const uint32_t entry = par->read_len + 2;
    std::unique_ptr<plasship_seqdb> o(new plasship_seqdb());
    const uint64_t dataBytes = nReads * entry;
    if (o->d_data.alloc(dataBytes + 64) != hipSuccess || o->d_off.alloc((nReads + 1) * 8) != hipSuccess || o->d_len.alloc((nReads + 1) * 4) != hipSuccess ||
        o->d_key.alloc((nReads + 1) * 4) != hipSuccess) { setError("plasship_synth_read_pairs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync((char *) o->d_data.p + dataBytes, 0, 64, st));
    PH_CHECK(hipMemsetAsync(o->d_off.p, 0, 8, st));
    SynthReads sr; memset(&sr, 0, sizeof(sr));
    sr.genome = dGenome.as<char>(); sr.genomeStart = dGStart.as<uint64_t>(); sr.cum = dCum.as<uint64_t>(); sr.nGenomes = G; sr.readLen = par->read_len;
    sr.insertMin = par->insert_min; sr.insertMean = par->insert_mean; sr.insertSd = par->insert_sd; sr.errThresh = (uint32_t) ((double) par->error_rate * 1073741824.0);
    sr.nPairs = par->n_pairs; sr.seed = par->seed; sr.out = o->d_data.as<char>(); sr.off = o->d_off.as<uint64_t>(); sr.len = o->d_len.as<uint32_t>(); sr.key = o->d_key.as<uint32_t>();
    if (nReads) hipLaunchKernelGGL(synthReadsKernel, dim3((unsigned) std::min<uint64_t>((nReads + 255) / 256, (uint64_t) ctx->numCU * 64)), dim3(256), 0, st, sr);
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    PH_CHECK(hipStreamSynchronize(st));
    PH_CHECK(hipGetLastError());
    o->dbtype = PLASSHIP_DBTYPE_NUCLEOTIDES; o->n = (size_t) nReads; o->dataBytes = dataBytes; o->residues = nReads * par->read_len; o->maxEntryLen = nReads ? entry : 0; o->hostIndexValid = false;
    if (stats) {
        float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        stats->ms_kernel = ms; stats->genome_bases = total; stats->n_genes = nGenes;
        stats->mean_coverage = total ? (double) (nReads * par->read_len) / (double) total : 0.0;
        double mx = 0; for (uint32_t g = 0; g < G; g++) mx = std::max(mx, abund[g] / wsum);
        stats->max_coverage = mx * (double) (nReads * par->read_len);
    }
    *out_reads = o.release();
    return PLASSHIP_OK;
}
