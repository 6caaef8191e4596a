// plasship_synth: seeded synthetic read pairs generated in HBM (include/plasship_synth.h).
// MEASUREMENT INFRASTRUCTURE — no reference counterpart; the model is SURVEY.md section 8d's (gene-dense genomes, log-normal
// community, paired 2 x 150 nt reads with substitution errors).  Everything is a pure function of (seed, indices).
#include "common.hpp"
#include "device_utils.hpp"
#include "../../include/plasship_synth.h"
#include "synth_core.hpp"
#include <algorithm>
#include <cmath>
#include <memory>
#include <cstring>

namespace plasship {

__global__ __launch_bounds__(256) void synthGenomeKernel(SynthGenome g, char *__restrict__ out) {
    for (uint64_t x = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; x < g.totalBases; x += (uint64_t) gridDim.x * blockDim.x) out[x] = synthGenomeBase(g, x);
}
__global__ __launch_bounds__(256) void synthReadsKernel(SynthReads a) {
    const uint64_t nReads = 2 * a.nPairs;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < nReads; r += (uint64_t) gridDim.x * blockDim.x) synthRead(a, r);
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
    SynthCommunity cmty; cmty.build(par->seed, par->n_genomes, par->genome_min_len, par->genome_max_len, par->abundance_sigma);
    const uint32_t G = par->n_genomes;
    const std::vector<uint64_t> &gStart = cmty.gStart, &geneStart = cmty.geneStart, &cum = cmty.cum; const std::vector<uint32_t> &geneCodons = cmty.geneCodons;
    const std::vector<double> &abund = cmty.abund; const double wsum = cmty.wsum; const uint64_t total = cmty.total;
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
    if (o->d_data.allocLong(dataBytes + 64) != hipSuccess || o->d_off.allocLong((nReads + 1) * 8) != hipSuccess || o->d_len.allocLong((nReads + 1) * 4) != hipSuccess ||
        o->d_key.allocLong((nReads + 1) * 4) != hipSuccess) { setError("plasship_synth_read_pairs: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemsetAsync((char *) o->d_data.p + dataBytes, 0, 64, st));
    PH_CHECK(hipMemsetAsync(o->d_off.p, 0, 8, st));
    SynthReads sr; memset(&sr, 0, sizeof(sr));
    sr.genome = dGenome.as<char>(); sr.genomeStart = dGStart.as<uint64_t>(); sr.cum = dCum.as<uint64_t>(); sr.nGenomes = G; sr.readLen = par->read_len;
    sr.insertMin = par->insert_min; sr.insertMean = par->insert_mean; sr.insertSd = par->insert_sd; sr.errThresh = (uint32_t) ((double) par->error_rate * 1073741824.0);
    sr.nPairs = par->n_pairs; sr.seed = par->seed; sr.out = o->d_data.as<char>(); sr.off = o->d_off.as<uint64_t>(); sr.len = o->d_len.as<uint32_t>(); sr.key = o->d_key.as<uint32_t>();
    if (nReads) hipLaunchKernelGGL(synthReadsKernel, dim3((unsigned) std::min<uint64_t>((nReads + 255) / 256, (uint64_t) ctx->numCU * 64)), dim3(256), 0, st, sr);
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    PH_CHECK(plasship::streamSync(st));
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
