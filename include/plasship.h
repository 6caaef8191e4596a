/* plasship — C-ABI of the MI355X-native Plass/PenguiN hot path
 *   kmermatcher -> rescorediagonal -> assembleresults | nuclassembleresults
 *   kmermatcher -> rescorediagonal -a 1 -> proteinaln2nucl -> guidedassembleresults      (penguin guided_nuclassemble)
 *
 * This is the drop-in boundary.  In the reference each of the three steps is an
 * MMseqs2 "module" `int f(int argc, const char** argv, const Command&)` (mm/commons/Command.h:91-102)
 * registered by name (src/plass.cpp:24-30, src/penguin.cpp:30-47) and run as a sub-process by
 * data/assemble.sh:92,103,145; modules exchange DBReader/DBWriter databases on disk.  A derived tool
 * overrides a module by registering the same name (mm/commons/Application.cpp:24-36).  The entry
 * points below are what such an override binds: plain pointers and sizes, no C++
 * or torch types.  Host buffers are owned by the caller, device buffers by the library.  All functions
 * return 0 on success and a negative code on failure; plasship_last_error() gives the message (thread
 * local).  One context per process/GPU; calls on one context are not re-entrant.  There is NO CPU
 * fallback: if no gfx950 device is usable, plasship_ctx_create fails.
 */
#ifndef PLASSHIP_H
#define PLASSHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct plasship_ctx plasship_ctx;       /* one GPU + stream + scratch arenas              */
typedef struct plasship_seqdb plasship_seqdb;   /* sequence DB resident in HBM                    */
typedef struct plasship_cands plasship_cands;   /* prefilter result (candidate pairs) in HBM      */
typedef struct plasship_alns plasship_alns;     /* alignment result (verified overlaps) in HBM    */

#define PLASSHIP_OK 0
#define PLASSHIP_ERR_ARG (-1)
#define PLASSHIP_ERR_IO (-2)
#define PLASSHIP_ERR_DEVICE (-3)
#define PLASSHIP_ERR_UNSUPPORTED (-4)
#define PLASSHIP_ERR_PEER (-5)        /* sharded run: another rank failed inside this call; every rank returns an error from it */

/* MMseqs2 dbtype codes (mm/commons/Parameters.h:65-84) */
#define PLASSHIP_DBTYPE_AMINO_ACIDS 0
#define PLASSHIP_DBTYPE_NUCLEOTIDES 1
#define PLASSHIP_DBTYPE_ALIGNMENT_RES 5
#define PLASSHIP_DBTYPE_PREFILTER_RES 7
#define PLASSHIP_DBTYPE_PREFILTER_REV_RES 14

const char *plasship_last_error(void);
const char *plasship_version(void);

/* ---- context ------------------------------------------------------------------------------ */
/* replaces: process start-up of a module (MMseqsMPI::init + Parameters singleton,
 * mm/linclust/kmermatcher.cpp:780-785).  device_ordinal < 0 = use LOCAL_RANK or 0.            */
int plasship_ctx_create(int device_ordinal, plasship_ctx **out);
void plasship_ctx_destroy(plasship_ctx *ctx);
int plasship_ctx_sync(plasship_ctx *ctx);
/* A caller that knows a large job is coming (the fused chain drivers: a multi-GB input DB) lets the library take its device arena —
 * one hipMalloc of most of the free HBM, 3-4 s on an MI355X whose memory another process has just used — in a background thread
 * while the caller reads and parses its input on the host.  Allocations wait for it; without the call the first large allocation
 * takes the arena itself.  No reference counterpart (the reference has no device). */
int plasship_ctx_reserve_async(plasship_ctx *ctx);
/* raw hipStream_t of the context, so a caller can bracket work with its own events */
void *plasship_ctx_stream(plasship_ctx *ctx);
/* diagnostic: how often the host has waited for a stream since the library was loaded (all contexts of the process).  A module
 * call is a chain of kernels on the context stream; the host waits only where it needs a size to allocate the next buffer. */
unsigned long long plasship_host_syncs(void);
/* test hook: the nth collective this context enters from now on (0 = the next one) fails locally BEFORE anything is exchanged, as
 * an allocation between two exchanges would; nth < 0 disarms.  tests/test_gpu_sharded.py uses it to check that one failing rank
 * makes every rank return an error from the same call instead of leaving the others inside a collective. */
int plasship_ctx_debug_fail_collective(plasship_ctx *ctx, int nth);

/* ---- one read set sharded over several GPUs (one process / context per GPU) ------------------
 * replaces: the reference's split of kmermatcher over MPI ranks by k-mer hash range
 * (`$RUNNER` = mpirun in data/assemble.sh:92,103; hashStartRange/hashEndRange, mm/linclust/kmermatcher.cpp:312,736-778;
 * MMseqsMPI::rank/numProc) — here every stage of the iteration is sharded and the result is the single-process one.
 *
 * Every rank holds the WHOLE sequence DB (replicated, identical on all ranks) and calls the same entry points with the
 * same arguments in the same order.  With a communicator set:
 *   plasship_kmermatch   extracts the k-mers of its share of the sequences, routes the records to the owner of their
 *                        k-mer hash bucket (all-to-all), groups them there, routes the grouped (rep, member, diagonal)
 *                        records to the owner of the representative (all-to-all) and returns the candidates of the
 *                        queries it owns (rank r owns ids [ceil(r*n/world), ceil((r+1)*n/world)); other queries have no
 *                        lines);
 *   plasship_rescore     as before (it sees only the owned queries' candidates);
 *   plasship_assemble /  extend the owned queries, all-gather the extended sequences and return the complete output
 *   _guided_assemble     DB(s) on every rank.
 * The library does not talk to a network itself: the caller supplies the three collectives (bench.py: torch.distributed
 * = RCCL over xGMI on device pointers; tests: an in-process implementation that runs several ranks on one GPU).
 * All callbacks are collective, return 0 on success, and must have completed (device buffers ready on the context's
 * stream or the whole device) when they return.  The library synchronises its stream before calling them.            */
typedef struct plasship_comm {
    int rank, world;
    void *user;
    /* recv[r * bytes_per_rank ...] = send of rank r (host memory) */
    int (*allgather_host)(void *user, const void *send, void *recv, uint64_t bytes_per_rank);
    /* device buffers; send is laid out by destination rank (send_bytes[world]), recv by source rank (recv_bytes[world]) */
    int (*alltoallv_dev)(void *user, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes);
    /* device buffers; every rank contributes send_bytes (its own entry of recv_bytes[world]); recv laid out by rank */
    int (*allgatherv_dev)(void *user, const void *d_send, uint64_t send_bytes, void *d_recv, const uint64_t *recv_bytes);
    /* != 0: the device collectives enqueue their work on the CONTEXT'S stream (plasship_ctx_stream) and may return before it
     * has run: the library then neither drains its stream before calling them nor assumes completion afterwards — everything is
     * ordered by the stream (the native RCCL communicator, include/plasship_rccl.h).  0: as described above. */
    int stream_ordered;
} plasship_comm;
/* comm == NULL returns the context to single-GPU operation.  The struct is copied. */
int plasship_ctx_set_comm(plasship_ctx *ctx, const plasship_comm *comm);
/* device-to-device copy on the context's stream, waited for (used by in-process communicators) */
int plasship_ctx_copy_d2d(plasship_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes);

/* ---- sequence DB  (replaces DBReader<unsigned int>::open/getData/getSeqLen/getDbKey,
 *      mm/commons/DBReader.cpp:150-215,548-589; DBReader.h:185-213) --------------------------- */
/* data = concatenation of entries "SEQ\n\0"; off/elen index it (elen includes "\n\0");
 * keys need not be sorted — ids are ranks in key order like DBReader::getId.                   */
int plasship_seqdb_upload(plasship_ctx *ctx, const char *data, size_t data_bytes, const uint64_t *off,
                          const uint32_t *elen, const uint32_t *key, size_t n, int dbtype,
                          plasship_seqdb **out);
int plasship_seqdb_read(plasship_ctx *ctx, const char *db_path, plasship_seqdb **out);
/* replaces DBWriter::writeData/close for a sequence DB (mm/commons/DBWriter.cpp:362-419,522-614) */
int plasship_seqdb_write(plasship_ctx *ctx, const plasship_seqdb *db, const char *db_path);
int plasship_seqdb_info(const plasship_seqdb *db, size_t *n, uint64_t *residues, uint32_t *max_entry_len,
                        int *dbtype, uint64_t *data_bytes);
/* Order-independent digest of a resident sequence DB: sum over the entries of a 64-bit hash of (key, entry length, entry bytes
 * with the trailing "\n\0") — FNV-1a over the bytes, seeded with key and length, then a 64-bit finaliser.  Not a reference
 * interface: it is how DBs too large to write out are compared (bench.py puts the digest of every iteration's output DB on its
 * JSON line; the CPU oracle's `plass_oracle dbsum` computes the same number from DB files, tests/golden/large_chain.json). */
int plasship_seqdb_digest(plasship_ctx *ctx, const plasship_seqdb *db, uint64_t *digest, uint64_t *entry_bytes);
/* download in key order; any pointer may be NULL */
int plasship_seqdb_download(plasship_ctx *ctx, const plasship_seqdb *db, char *data, uint64_t *off,
                            uint32_t *elen, uint32_t *key);
void plasship_seqdb_free(plasship_ctx *ctx, plasship_seqdb *db);

/* ---- kmermatcher  (replaces int kmermatcher(int, const char**, const Command&),
 *      mm/linclust/kmermatcher.cpp:780-807; flags of Parameters.cpp:872-892) ------------------- */
typedef struct plasship_kmermatch_params {
    int32_t kmer_size;               /* -k                       (14 plass / 22 penguin nucl)     */
    int32_t alphabet_size;           /* --alph-size aa           (13; 21 = full)                  */
    int32_t kmers_per_seq;           /* --kmer-per-seq           (60)                             */
    float kmers_per_seq_scale;       /* --kmer-per-seq-scale for the DB's alphabet               */
    int32_t hash_shift;              /* --hash-shift             (67, 68, 68, 69 …)               */
    int32_t include_only_extendable; /* --include-only-extendable                                 */
    int32_t ignore_multi_kmer;       /* --ignore-multi-kmer                                       */
    int32_t cov_mode;                /* --cov-mode                                                */
    float cov_thr;                   /* -c                                                        */
} plasship_kmermatch_params;

typedef struct plasship_kmermatch_stats {
    uint64_t n_kmer_records;   /* N_k: records emitted by extraction (incl. identity records)    */
    uint64_t n_grouped;        /* N_m: records after assignGroup                                 */
    uint64_t n_candidates;     /* N_c: non-self prefilter lines                                  */
    uint32_t record_bytes;     /* 16 (T=short) or 20 (T=int) in the reference layout             */
    float ms_extract, ms_sort1, ms_group, ms_sort2, ms_reduce; /* HIP-event stage times (incl. scans)  */
    float ms_extract_kernel;   /* both extraction kernel launches (HIP events on the ctx stream)     */
    uint64_t residues;         /* R: residues read by the extraction                              */
    /* per kernel: one-thread-per-sequence kernel (short reads) and wave-per-sequence kernel (the rest) */
    float ms_extract_short_kernel, ms_extract_wave_kernel;
    uint64_t short_residues, short_records, wave_residues, wave_records;
    float ms_part_scatter;      /* the partScatterKernel launches of the hash partition (sort #1), summed            */
    int32_t n_part_scatter;     /* how many (1 or 2 levels)                                                          */
    uint32_t n_scratch_sequences; /* sequences whose candidate k-mers did not fit LDS (HBM-scratch launch of the extraction)  */
    uint32_t n_restarts;        /* 1: the call started over because the extraction's overflow count, checked late, was not 0 */
    uint32_t n_cached_sequences; /* sequences whose selected windows came from the previous call's cache (same hash seed, bytes unchanged) */
    uint32_t reserved0;
} plasship_kmermatch_stats;

int plasship_kmermatch(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_kmermatch_params *par,
                       plasship_cands **out, plasship_kmermatch_stats *stats);

/* prefilter DB <-> device candidate list (hit_t lines "seqId\tprefScore\tdiagonal\n",
 * mm/prefiltering/QueryMatcher.h:35-51,81-126; back-fill of self-only entries kmermatcher.cpp:705-724) */
int plasship_cands_write(plasship_ctx *ctx, const plasship_cands *c, const plasship_seqdb *db, const char *db_path);
int plasship_cands_read(plasship_ctx *ctx, const plasship_seqdb *qdb, const plasship_seqdb *tdb,
                        const char *db_path, plasship_cands **out);
int plasship_cands_count(const plasship_cands *c, uint64_t *n_hits, int *reverse_capable);
/* arrays of n_hits (non-self) entries in prefilter order (query key, then target key ascending);
 * qdb/tdb are the DBs the list was built on (ids are mapped back to DB keys) */
int plasship_cands_download(plasship_ctx *ctx, const plasship_cands *c, const plasship_seqdb *qdb,
                            const plasship_seqdb *tdb, uint32_t *query_key, uint32_t *target_key,
                            int32_t *pref_score, uint16_t *diagonal);
void plasship_cands_free(plasship_ctx *ctx, plasship_cands *c);

/* ---- rescorediagonal  (replaces int rescorediagonal(int, const char**, const Command&),
 *      mm/alignment/rescorediagonal.cpp:381-431; flags of Parameters.cpp:423-439) -------------- */
typedef struct plasship_rescore_params {
    int32_t rescore_mode;     /* --rescore-mode: 2 local start/end, 3 end-to-end (default)        */
    double eval_thr;          /* -e                                                               */
    float seq_id_thr;         /* --min-seq-id                                                     */
    int32_t cov_mode;         /* --cov-mode                                                       */
    float cov_thr;            /* -c                                                               */
    int32_t min_aln_len;      /* --min-aln-len                                                    */
    int32_t seq_id_mode;      /* --seq-id-mode                                                    */
    int32_t add_backtrace;    /* -a                                                               */
    int32_t include_identity; /* --add-self-matches                                               */
} plasship_rescore_params;

typedef struct plasship_rescore_stats {
    uint64_t n_scored;    /* candidate pairs scored (incl. the self hit of every query)          */
    uint64_t n_accepted;  /* alignment lines kept                                                */
    uint64_t overlap_residues; /* Σ diagonal overlap length over the pairs scored BY THIS CALL: the identity pairs it leaves
                                * as stubs (below) are not in it                                 */
    float ms_kernel;
} plasship_rescore_stats;

/* qdb == tdb (same handle) is the plass case (sameQTDB, rescorediagonal.cpp:59-69).
 * LIFETIME AND THREADING of the list this returns: every query's alignment with itself is left as an unscored stub and scored by the
 * first plasship_* call that READS a self record (plasship_alns_write / _download / _count of a dense copy, plasship_aln2nucl, the
 * nucleotide and guided extension).  Until then the list refers to qdb and tdb and to this call's parameters: both DBs must outlive
 * the list (they must anyway: ids become keys when the list is written), and the first such reader MODIFIES the list's records in
 * place — one context / one thread at a time may use a list made by plasship_rescore. */
int plasship_rescore(plasship_ctx *ctx, const plasship_seqdb *qdb, const plasship_seqdb *tdb,
                     const plasship_cands *c, const plasship_rescore_params *par, plasship_alns **out,
                     plasship_rescore_stats *stats);

/* alignment DB <-> device (Matcher::resultToBuffer / parseAlignmentRecord, mm/alignment/Matcher.cpp:248-370) */
int plasship_alns_write(plasship_ctx *ctx, const plasship_alns *a, const char *db_path);
int plasship_alns_read(plasship_ctx *ctx, const plasship_seqdb *db, const char *db_path, plasship_alns **out);
int plasship_alns_count(const plasship_alns *a, uint64_t *n_lines);
typedef struct plasship_aln_record {   /* one accepted alignment line, binary */
    uint32_t query_key, target_key;
    int32_t bit_score, raw_score;
    float seq_id;              /* exact float before text truncation                              */
    int32_t q_start, q_end, q_len, db_start, db_end, db_len, aln_len;
    int32_t reversed;
} plasship_aln_record;
int plasship_alns_download(plasship_ctx *ctx, const plasship_alns *a, plasship_aln_record *out);
void plasship_alns_free(plasship_ctx *ctx, plasship_alns *a);

/* ---- assembleresults / nuclassembleresults
 *      protein DB    -> replaces int assembleresult(int, const char**, const Command&),
 *                       src/assembler/assembleresult.cpp:358-368
 *      nucleotide DB -> replaces int nuclassembleresult(int, const char**, const Command&),
 *                       src/assembler/nuclassembleresult.cpp:400-410 (reverse-strand hits, Bayesian comparator,
 *                       libstdc++ heap order; comparator decisions that fall on the 0.45 / 0.55 thresholds are
 *                       evaluated with the host's libm like the reference does)
 *      flags LocalParameters.h:96-102 ----------------------------------------------------------- */
typedef struct plasship_assemble_params {
    float seq_id_thr;       /* --min-seq-id                                                      */
    uint64_t max_seq_len;   /* --max-seq-len                                                     */
    int32_t keep_target;    /* --keep-target                                                     */
    int32_t rescore_mode;   /* --rescore-mode                                                    */
} plasship_assemble_params;

typedef struct plasship_assemble_stats {
    uint64_t n_extended;        /* queries that became (longer) contigs                          */
    uint64_t n_rescored;        /* deferred hits re-scored on an extended query (A4)             */
    uint64_t out_residues;
    float ms_kernel;            /* whole stage (arena sizing .. output DB)                          */
    float ms_assemble_kernel;   /* the per-query extension kernel alone                              */
    uint64_t n_alignments;      /* alignment lines consumed                                          */
    uint64_t rescored_residues; /* overlap residues re-scored                                        */
    /* per kernel tier: [0] 16 lanes per query (<= 16 alignments), [1] one wave per query (17..64), [2] HBM queue (> 64) */
    float ms_tier_kernel[3];
    uint64_t tier_alignments[3], tier_query_residues[3], tier_rescored_residues[3];
    /* how the output DB was made: bytes appended to the heap it shares with the input DB (the rewritten entries only), or bytes of a
     * full copy (no shared heap yet, or no room left in it) — one of the two is 0 */
    uint64_t db_appended_bytes, db_copied_bytes;
} plasship_assemble_stats;

int plasship_assemble(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_alns *a,
                      const plasship_assemble_params *par, plasship_seqdb **out, plasship_assemble_stats *stats);

/* ---- guidedassembleresults  (replaces int guidedassembleresults(int, const char**, const Command&),
 *      src/assembler/guidedassembleresult.cpp:387-397; positional args <nuclDB> <aaDB> <nuclAlnDB> <outNuclDB> <outAaDB>,
 *      flags as for assembleresults).  nucl_db and aa_db hold the same keys (ORF i and its translation); `a` is the
 *      nucleotide-level alignment list (plasship_aln2nucl, or plasship_alns_read on nucl_db). ------------------------ */
int plasship_guided_assemble(plasship_ctx *ctx, const plasship_seqdb *nucl_db, const plasship_seqdb *aa_db, const plasship_alns *a,
                             const plasship_assemble_params *par, plasship_seqdb **out_nucl, plasship_seqdb **out_aa,
                             plasship_assemble_stats *stats);

/* ---- proteinaln2nucl  (replaces int proteinaln2nucl(int, const char**, const Command&), mm/util/proteinaln2nucl.cpp:13-204;
 *      positional args <qNuclDB> <tNuclDB> <qAaDB> <tAaDB> <alnDB> <outAlnDB>).  `a` is a protein alignment list with
 *      backtrace (plasship_rescore with add_backtrace, or plasship_alns_read on the protein DB); the result refers to the
 *      nucleotide DBs.  Ungapped alignments only (what --rescore-mode 3 produces). ------------------------------------- */
typedef struct plasship_aln2nucl_params {
    int32_t gap_open;      /* --gap-open   (nucleotide value; the penguin workflow passes 5)                        */
    int32_t gap_extend;    /* --gap-extend (nucleotide value; the penguin workflow passes 2)                        */
} plasship_aln2nucl_params;
typedef struct plasship_aln2nucl_stats {
    uint64_t n_alignments;
    float ms_kernel;
} plasship_aln2nucl_stats;
int plasship_aln2nucl(plasship_ctx *ctx, const plasship_seqdb *q_nucl, const plasship_seqdb *t_nucl, const plasship_seqdb *q_aa,
                      const plasship_seqdb *t_aa, const plasship_alns *a, const plasship_aln2nucl_params *par, plasship_alns **out,
                      plasship_aln2nucl_stats *stats);

/* ---- findassemblystart  (replaces int findassemblystart(int, const char**, const Command&),
 *      src/assembler/findassemblystart.cpp:35-176; positional args <seqDB> <alnDB> <outSeqDB>; called once, inside
 *      iteration 0 of data/assemble.sh:110-141).  Sequences whose alignments agree on a "*M" start column are cut to
 *      "*" + the sequence from that column on; all others are carried over.  Protein DBs only. ------------------------- */
typedef struct plasship_findstart_stats {
    uint64_t n_alignments;
    uint64_t out_residues;
    float ms_kernel;
} plasship_findstart_stats;
int plasship_find_assembly_start(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_alns *a, plasship_seqdb **out,
                                 plasship_findstart_stats *stats);

/* ---- cyclecheck  (replaces int cyclecheck(int, const char**, const Command&), src/assembler/cyclecheck.cpp:30-291;
 *      positional args <sequenceDB> <cycleDB>, flags --max-seq-len --chop-cycle; called after every nuclassembleresults of
 *      the penguin workflows, data/nuclassemble.sh:19-61,132).  out_cycle = the circular / terminally redundant contigs
 *      (cut at the split diagonal with chop_cycle); out_rest (may be NULL) = all other sequences, what the workflow's
 *      "<db>_noneCycle" is and the next iteration continues with.  Nucleotide DBs only, k = 22 like the reference. ------ */
typedef struct plasship_cyclecheck_params {
    uint64_t max_seq_len;  /* --max-seq-len : sequences of this length or more are skipped (not reported)             */
    int32_t chop_cycle;    /* --chop-cycle  : write only the first <split diagonal> residues                           */
} plasship_cyclecheck_params;
typedef struct plasship_cyclecheck_stats {
    uint64_t n_cyclic;
    uint64_t n_wave_small, n_wave_large, n_block;   /* sequences per kernel tier (<= 380 nt, <= 3068 nt, longer)       */
    float ms_kernel;
    uint64_t n_known;      /* entries not looked at: unchanged since the last call's "rest" DB, all of whose entries are linear */
} plasship_cyclecheck_stats;
int plasship_cyclecheck(plasship_ctx *ctx, const plasship_seqdb *db, const plasship_cyclecheck_params *par, plasship_seqdb **out_cycle,
                        plasship_seqdb **out_rest, plasship_cyclecheck_stats *stats);

/* ---- extractorfs / translatenucs / concatdbs  (the once-per-run preprocessing of
 *      data/assemble.sh:41-77 and data/guidedNuclAssemble.sh:46-72 that makes the DB the hot path uploads).
 *      replaces int extractorfs(int, const char**, const Command&), mm/util/extractorfs.cpp:20-159 (+ mm/commons/Orf.cpp);
 *               int translatenucs(…), mm/util/translatenucs.cpp:14-117 (+ mm/commons/TranslateNucl.h);
 *               int concatdbs(…), mm/util/concatdbs.cpp via DBConcat (mm/commons/DBConcat.cpp:19-145).
 *      An ORF DB travels with its header DB (<db>_h: "<readKey>\t<from>[+-]<len>[\t<incomplete flags>]\n",
 *      Orf::writeOrfHeader, Orf.cpp:438-456), kept on the device as a plasship_orfhdr. ------------------------------------ */
typedef struct plasship_orfhdr plasship_orfhdr;
typedef struct plasship_orf_params {
    int32_t min_length;          /* --min-length   (codons)                                                        */
    int32_t max_length;          /* --max-length   (codons)                                                        */
    int32_t max_gaps;            /* --max-gaps     (codons with N / unknown letters)                               */
    int32_t contig_start_mode;   /* --contig-start-mode 0 incomplete, 1 complete, 2 both                           */
    int32_t contig_end_mode;     /* --contig-end-mode                                                              */
    int32_t orf_start_mode;      /* --orf-start-mode 0 ATG-to-stop, 1 any-to-stop, 2 last-ATG-to-stop              */
    int32_t forward_frames;      /* --forward-frames as a bit mask (frame 1 = bit 0)                               */
    int32_t reverse_frames;      /* --reverse-frames                                                               */
    int32_t translation_table;   /* --translation-table (1 only)                                                   */
    int32_t translate;           /* --translate                                                                    */
    int32_t use_all_table_starts;/* --use-all-table-starts (0 only)                                                */
    uint64_t max_seq_len;        /* --max-seq-len (only read with --translate)                                     */
} plasship_orf_params;
typedef struct plasship_translate_params {
    int32_t translation_table;   /* --translation-table (1 only)                                                   */
    int32_t add_orf_stop;        /* --add-orf-stop                                                                 */
    uint64_t max_seq_len;        /* --max-seq-len                                                                  */
} plasship_translate_params;
typedef struct plasship_orf_stats {
    uint64_t n_out;              /* entries written                                                                */
    uint64_t in_residues, out_residues;
    float ms_kernel;
} plasship_orf_stats;
/* out_hdr may be NULL.  Keys of the output are 0..M-1 in the reference's order (read key, then position of discovery). */
int plasship_extract_orfs(plasship_ctx *ctx, const plasship_seqdb *reads, const plasship_orf_params *par, plasship_seqdb **out_orfs,
                          plasship_orfhdr **out_hdr, plasship_orf_stats *stats);
/* hdr: the header DB of `orfs` (same keys); only read with add_orf_stop, may be NULL otherwise */
int plasship_translate_nucs(plasship_ctx *ctx, const plasship_seqdb *orfs, const plasship_orfhdr *hdr, const plasship_translate_params *par,
                            plasship_seqdb **out_aa, plasship_orf_stats *stats);
/* concatdbs <A> <B> <out> without --preserve-keys: keys of A kept, entry i of B (in key order) gets key max(keyA) + 1 + i */
int plasship_seqdb_concat(plasship_ctx *ctx, const plasship_seqdb *a, const plasship_seqdb *b, plasship_seqdb **out);
/* concatdbs <A> <B> <out> --preserve-keys (mm/commons/DBConcat.cpp:113-118 with preserveKeysB = true; data/nuclassemble.sh:41,145): the union of
 * the two DBs, every entry under its own key.  preserve_keys_b == 0: plasship_seqdb_concat.  A key that occurs in both DBs: PLASSHIP_ERR_UNSUPPORTED. */
int plasship_seqdb_concat_keys(plasship_ctx *ctx, const plasship_seqdb *a, const plasship_seqdb *b, int preserve_keys_b, plasship_seqdb **out);
int plasship_orfhdr_concat(plasship_ctx *ctx, const plasship_orfhdr *a, const plasship_orfhdr *b, plasship_orfhdr **out);
int plasship_orfhdr_read(plasship_ctx *ctx, const char *db_path, plasship_orfhdr **out);
int plasship_orfhdr_write(plasship_ctx *ctx, const plasship_orfhdr *h, const char *db_path);
int plasship_orfhdr_count(const plasship_orfhdr *h, size_t *n);
void plasship_orfhdr_free(plasship_ctx *ctx, plasship_orfhdr *h);

#ifdef __cplusplus
}
#endif
#endif /* PLASSHIP_H */
