/* plasship_synth — seeded synthetic read sets, generated on the GPU.
 *
 * MEASUREMENT INFRASTRUCTURE, not a reference interface: the reference has no read simulator.  bench.py and the large
 * parity tests need read sets of up to 50 M reads (BASELINE.json configs[2]) in HBM within seconds; a host generator
 * plus a PCIe upload of 7.6 GB would dominate every run.  The reads come out as an ordinary nucleotide plasship_seqdb
 * (entries "SEQ\n\0", keys 0..2*pairs-1, mates of pair i under keys 2i and 2i+1), so everything downstream — extractorfs,
 * translatenucs, the hot path, plasship_seqdb_write for the CPU oracle — treats them like reads that came from disk.
 *
 * Model: a community of `n_genomes` genomes with lengths uniform in [genome_min_len, genome_max_len] and
 * log-normal abundances (sigma = abundance_sigma; 0 = equal coverage); a genome is a chain of genes (ATG + 300..1500 sense
 * codons drawn uniformly from the 61 non-stop codons + a stop codon, random strand) separated by 50..200 random bases; a pair
 * is an insert of about N(insert_mean, insert_sd) (Irwin-Hall approximation in integer arithmetic, >= insert_min) at a uniform
 * position of a genome drawn with probability ~ abundance * length, read from both ends (read_len bases each, second mate
 * reverse-complemented, orientation of the pair random) with `error_rate` substitutions.  Everything is a pure function of
 * (seed, indices) through a 64-bit integer mix: the same parameters give the same bytes on every run and every GPU.
 */
#ifndef PLASSHIP_SYNTH_H
#define PLASSHIP_SYNTH_H
#include "plasship.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct plasship_synth_params {
    uint64_t n_pairs;
    uint64_t seed;
    uint32_t n_genomes;
    uint64_t genome_min_len, genome_max_len;
    float abundance_sigma;
    float insert_mean, insert_sd;
    uint32_t insert_min;
    uint32_t read_len;
    float error_rate;
} plasship_synth_params;

typedef struct plasship_synth_stats {
    uint64_t genome_bases;     /* bases of all genomes together                        */
    uint64_t n_genes;
    double mean_coverage;      /* read bases / genome bases                            */
    double max_coverage;       /* of the most abundant genome                          */
    float ms_kernel;
} plasship_synth_stats;

int plasship_synth_read_pairs(plasship_ctx *ctx, const plasship_synth_params *par, plasship_seqdb **out_reads, plasship_synth_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
