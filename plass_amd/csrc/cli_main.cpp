// plass-hip: the reference's module command lines over the plasship C-ABI.
//
// The reference workflow scripts run `"$MMSEQS" <module> <dbs…> <flags…>` (data/assemble.sh:92,103,145);
// pointing $MMSEQS at this binary for the three hot modules makes them run on the MI355X while every
// database on disk keeps the DBReader/DBWriter format.  Module names, positional arguments and flag names
// are the reference's (mm/commons/Parameters.cpp:423-439,872-892; src/commons/LocalParameters.h:96-102);
// flags that do not influence the hot path (--threads, -v, --sub-mat, --db-load-mode …) are accepted and
// ignored; unsupported values fail loudly like Debug(Debug::ERROR)+EXIT(EXIT_FAILURE) does.
#include "../../include/plasship.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool multiParam(const std::string &v, const char *which, std::string &out) {
    if (v.find(':') == std::string::npos) { out = v; return true; }
    size_t p = 0;
    while (p < v.size()) {
        size_t c = v.find(',', p); if (c == std::string::npos) c = v.size();
        std::string part = v.substr(p, c - p);
        size_t col = part.find(':');
        if (col != std::string::npos && part.substr(0, col) == which) { out = part.substr(col + 1); return true; }
        p = c + 1;
    }
    return false;
}

struct Flags {
    int k = 14, alph = 13, kps = 60, hashShift = 67, onlyExt = 0, ignoreMulti = 1, covMode = 0;
    float scaleAA = 0.0f, scaleNucl = 0.2f, covThr = 0.0f, seqIdThr = 0.9f;
    int rescoreMode = 3, minAlnLen = 0, seqIdMode = 0, addBt = 0, addSelf = 0, keepTarget = 1, wrapped = 0, filterHits = 0, sortResults = 0;
    double evalThr = 1e-5;
    unsigned long long maxSeqLen = 65535;
    int gapOpenNucl = 5, gapExtendNucl = 2;
    int chopCycle = 0;                   // cyclecheck: setCycleCheckDefaults (cyclecheck.cpp:25-28); the workflow passes --chop-cycle
};

static int fail(const char *what) { fprintf(stdout, "%s: %s\n", what, plasship_last_error()); return EXIT_FAILURE; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stdout, "usage: plass-hip <kmermatcher|rescorediagonal|assembleresults|nuclassembleresults|guidedassembleresults|proteinaln2nucl|findassemblystart|cyclecheck> <dbs…> [flags]\n"); return EXIT_FAILURE; }
    const std::string mod = argv[1];
    Flags f; std::vector<std::string> pos;
    if (mod == "kmermatcher") f.covThr = 0.8f;   // setLinearFilterDefault (kmermatcher.cpp:566-573); workflows pass -c
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) {
            if (i + 1 >= argc) { fprintf(stdout, "Missing value for %s\n", a.c_str()); return EXIT_FAILURE; }
            std::string v = argv[++i], t;
            if (a == "-k") f.k = atoi(v.c_str());
            else if (a == "--alph-size") { if (multiParam(v, "aa", t)) f.alph = atoi(t.c_str()); }
            else if (a == "--kmer-per-seq") f.kps = atoi(v.c_str());
            else if (a == "--kmer-per-seq-scale") { if (multiParam(v, "aa", t)) f.scaleAA = strtof(t.c_str(), nullptr); if (multiParam(v, "nucl", t)) f.scaleNucl = strtof(t.c_str(), nullptr); }
            else if (a == "--hash-shift") f.hashShift = atoi(v.c_str());
            else if (a == "--include-only-extendable") f.onlyExt = atoi(v.c_str());
            else if (a == "--ignore-multi-kmer") f.ignoreMulti = atoi(v.c_str());
            else if (a == "--cov-mode") f.covMode = atoi(v.c_str());
            else if (a == "-c") f.covThr = strtof(v.c_str(), nullptr);
            else if (a == "--rescore-mode") f.rescoreMode = atoi(v.c_str());
            else if (a == "-e") f.evalThr = strtod(v.c_str(), nullptr);
            else if (a == "--min-seq-id") f.seqIdThr = strtof(v.c_str(), nullptr);
            else if (a == "--min-aln-len") f.minAlnLen = atoi(v.c_str());
            else if (a == "--seq-id-mode") f.seqIdMode = atoi(v.c_str());
            else if (a == "-a") f.addBt = atoi(v.c_str());
            else if (a == "--add-self-matches") f.addSelf = atoi(v.c_str());
            else if (a == "--max-seq-len") f.maxSeqLen = strtoull(v.c_str(), nullptr, 10);
            else if (a == "--chop-cycle") f.chopCycle = atoi(v.c_str());
            else if (a == "--keep-target") f.keepTarget = atoi(v.c_str());
            else if (a == "--gap-open") { if (multiParam(v, "nucl", t)) f.gapOpenNucl = atoi(t.c_str()); }
            else if (a == "--gap-extend") { if (multiParam(v, "nucl", t)) f.gapExtendNucl = atoi(t.c_str()); }
            else if (a == "--wrapped-scoring") f.wrapped = atoi(v.c_str());
            else if (a == "--filter-hits") f.filterHits = atoi(v.c_str());
            else if (a == "--sort-results") f.sortResults = atoi(v.c_str());
            else if (a == "--spaced-kmer-mode" || a == "--mask" || a == "--mask-lower-case" || a == "--adjust-kmer-len" || a == "--compressed") {
                if (atoi(v.c_str()) != 0) { fprintf(stdout, "%s %s is not supported by plass-hip\n", a.c_str(), v.c_str()); return EXIT_FAILURE; }
            }
            // everything else (--threads, -v, --sub-mat, --db-load-mode, --split-memory-limit …): ignored
        } else pos.push_back(a);
    }
    if (f.wrapped || f.filterHits || f.sortResults) { fprintf(stdout, "--wrapped-scoring/--filter-hits/--sort-results are not supported by plass-hip\n"); return EXIT_FAILURE; }
    plasship_ctx *ctx = nullptr;
    if (plasship_ctx_create(-1, &ctx)) return fail("plass-hip");
    const double t0 = now();
    int rc = EXIT_SUCCESS;
    if (mod == "kmermatcher") {
        if (pos.size() != 2) { fprintf(stdout, "kmermatcher <i:sequenceDB> <o:prefDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr; plasship_cands *c = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &db)) return fail("kmermatcher");
        int dbtype = 0; plasship_seqdb_info(db, nullptr, nullptr, nullptr, &dbtype, nullptr);
        plasship_kmermatch_params p; memset(&p, 0, sizeof(p));
        p.kmer_size = f.k; p.alphabet_size = f.alph; p.kmers_per_seq = f.kps; p.kmers_per_seq_scale = (dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES) ? f.scaleNucl : f.scaleAA;
        p.hash_shift = f.hashShift; p.include_only_extendable = f.onlyExt; p.ignore_multi_kmer = f.ignoreMulti; p.cov_mode = f.covMode; p.cov_thr = f.covThr;
        plasship_kmermatch_stats st; memset(&st, 0, sizeof(st));
        if (plasship_kmermatch(ctx, db, &p, &c, &st)) return fail("kmermatcher");
        fprintf(stdout, "k-mer records: %llu grouped: %llu candidates: %llu | kernels ms: extract %.3f partition %.3f group %.3f sort %.3f reduce %.3f\n",
                (unsigned long long) st.n_kmer_records, (unsigned long long) st.n_grouped, (unsigned long long) st.n_candidates,
                st.ms_extract, st.ms_sort1, st.ms_group, st.ms_sort2, st.ms_reduce);
        if (plasship_cands_write(ctx, c, db, pos[1].c_str())) return fail("kmermatcher");
        plasship_cands_free(ctx, c); plasship_seqdb_free(ctx, db);
    } else if (mod == "rescorediagonal") {
        if (pos.size() != 4) { fprintf(stdout, "rescorediagonal <i:queryDB> <i:targetDB> <i:prefDB> <o:alnDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *q = nullptr, *t = nullptr; plasship_cands *c = nullptr; plasship_alns *al = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &q)) return fail("rescorediagonal");
        if (pos[1] == pos[0]) t = q; else if (plasship_seqdb_read(ctx, pos[1].c_str(), &t)) return fail("rescorediagonal");
        if (plasship_cands_read(ctx, q, t, pos[2].c_str(), &c)) return fail("rescorediagonal");
        plasship_rescore_params p; memset(&p, 0, sizeof(p));
        p.rescore_mode = f.rescoreMode; p.eval_thr = f.evalThr; p.seq_id_thr = f.seqIdThr; p.cov_mode = f.covMode; p.cov_thr = f.covThr;
        p.min_aln_len = f.minAlnLen; p.seq_id_mode = f.seqIdMode; p.add_backtrace = f.addBt; p.include_identity = f.addSelf;
        plasship_rescore_stats st; memset(&st, 0, sizeof(st));
        if (plasship_rescore(ctx, q, t, c, &p, &al, &st)) return fail("rescorediagonal");
        fprintf(stdout, "scored: %llu accepted: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_scored, (unsigned long long) st.n_accepted, st.ms_kernel);
        if (plasship_alns_write(ctx, al, pos[3].c_str())) return fail("rescorediagonal");
        plasship_alns_free(ctx, al); plasship_cands_free(ctx, c); if (t != q) plasship_seqdb_free(ctx, t); plasship_seqdb_free(ctx, q);
    } else if (mod == "assembleresults" || mod == "nuclassembleresults") {
        if (pos.size() != 3) { fprintf(stdout, "%s <i:sequenceDB> <i:alnResult> <o:reprSeqDB>\n", mod.c_str()); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr; plasship_alns *al = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &db)) return fail("assembleresults");
        // the library picks the variant from the DB type (protein -> assembleresults, nucleotide -> nuclassembleresults);
        // the module name has to agree, the reference's assembleresults on nucleotides uses another comparator
        int dbtype = -1; plasship_seqdb_info(db, nullptr, nullptr, nullptr, &dbtype, nullptr);
        if ((mod == "nuclassembleresults") != (dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES)) {
            fprintf(stdout, "plass-hip: %s needs a %s sequence DB\n", mod.c_str(), mod == "nuclassembleresults" ? "nucleotide" : "protein");
            return EXIT_FAILURE;
        }
        if (plasship_alns_read(ctx, db, pos[1].c_str(), &al)) return fail("assembleresults");
        plasship_assemble_params p; memset(&p, 0, sizeof(p));
        p.seq_id_thr = f.seqIdThr; p.max_seq_len = f.maxSeqLen; p.keep_target = f.keepTarget; p.rescore_mode = f.rescoreMode;
        plasship_assemble_stats st; memset(&st, 0, sizeof(st));
        if (plasship_assemble(ctx, db, al, &p, &o, &st)) return fail("assembleresults");
        fprintf(stdout, "extended: %llu rescored: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_extended, (unsigned long long) st.n_rescored, st.ms_kernel);
        if (plasship_seqdb_write(ctx, o, pos[2].c_str())) return fail("assembleresults");
        plasship_seqdb_free(ctx, o); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, db);
    } else if (mod == "guidedassembleresults") {
        if (pos.size() != 5) { fprintf(stdout, "guidedassembleresults <i:nuclSequenceDB> <i:aaSequenceDB> <i:nuclAlnResult> <o:nuclAssembly> <o:aaAssembly>\n"); return EXIT_FAILURE; }
        plasship_seqdb *nu = nullptr, *aa = nullptr, *on = nullptr, *oa = nullptr; plasship_alns *al = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &nu) || plasship_seqdb_read(ctx, pos[1].c_str(), &aa)) return fail("guidedassembleresults");
        if (plasship_alns_read(ctx, nu, pos[2].c_str(), &al)) return fail("guidedassembleresults");
        plasship_assemble_params p; memset(&p, 0, sizeof(p));
        p.seq_id_thr = f.seqIdThr; p.max_seq_len = f.maxSeqLen; p.keep_target = f.keepTarget; p.rescore_mode = f.rescoreMode;
        plasship_assemble_stats st; memset(&st, 0, sizeof(st));
        if (plasship_guided_assemble(ctx, nu, aa, al, &p, &on, &oa, &st)) return fail("guidedassembleresults");
        fprintf(stdout, "extended: %llu rescored: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_extended, (unsigned long long) st.n_rescored, st.ms_kernel);
        if (plasship_seqdb_write(ctx, on, pos[3].c_str()) || plasship_seqdb_write(ctx, oa, pos[4].c_str())) return fail("guidedassembleresults");
        plasship_seqdb_free(ctx, on); plasship_seqdb_free(ctx, oa); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, aa); plasship_seqdb_free(ctx, nu);
    } else if (mod == "proteinaln2nucl") {
        if (pos.size() != 6) { fprintf(stdout, "proteinaln2nucl <i:queryNuclDB> <i:targetNuclDB> <i:queryAaDB> <i:targetAaDB> <i:alnDB> <o:alnDB>\n"); return EXIT_FAILURE; }
        if ((pos[0] == pos[1]) != (pos[2] == pos[3])) { fprintf(stdout, "Either query database == target database for nucleotide and amino acid or != for both\n"); return EXIT_FAILURE; }
        if (pos[0] != pos[1]) { fprintf(stdout, "plass-hip: proteinaln2nucl with separate query and target DBs is not supported (the assembly workflows use one DB)\n"); return EXIT_FAILURE; }
        plasship_seqdb *nu = nullptr, *aa = nullptr; plasship_alns *al = nullptr, *o = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &nu) || plasship_seqdb_read(ctx, pos[2].c_str(), &aa)) return fail("proteinaln2nucl");
        if (plasship_alns_read(ctx, aa, pos[4].c_str(), &al)) return fail("proteinaln2nucl");
        plasship_aln2nucl_params p; p.gap_open = f.gapOpenNucl; p.gap_extend = f.gapExtendNucl;
        plasship_aln2nucl_stats st; memset(&st, 0, sizeof(st));
        if (plasship_aln2nucl(ctx, nu, nu, aa, aa, al, &p, &o, &st)) return fail("proteinaln2nucl");
        fprintf(stdout, "alignments: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_alignments, st.ms_kernel);
        if (plasship_alns_write(ctx, o, pos[5].c_str())) return fail("proteinaln2nucl");
        plasship_alns_free(ctx, o); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, aa); plasship_seqdb_free(ctx, nu);
    } else if (mod == "findassemblystart") {
        if (pos.size() != 3) { fprintf(stdout, "findassemblystart <i:sequenceDB> <i:alnDB> <o:sequenceDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr; plasship_alns *al = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &db)) return fail("findassemblystart");
        if (plasship_alns_read(ctx, db, pos[1].c_str(), &al)) return fail("findassemblystart");
        plasship_findstart_stats st; memset(&st, 0, sizeof(st));
        if (plasship_find_assembly_start(ctx, db, al, &o, &st)) return fail("findassemblystart");
        fprintf(stdout, "alignments: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_alignments, st.ms_kernel);
        if (plasship_seqdb_write(ctx, o, pos[2].c_str())) return fail("findassemblystart");
        plasship_seqdb_free(ctx, o); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, db);
    } else if (mod == "cyclecheck") {
        if (pos.size() != 2) { fprintf(stdout, "cyclecheck <i:sequenceDB> <o:sequenceDBcycle>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr;
        if (plasship_seqdb_read(ctx, pos[0].c_str(), &db)) return fail("cyclecheck");
        plasship_cyclecheck_params p; p.max_seq_len = f.maxSeqLen; p.chop_cycle = f.chopCycle;
        plasship_cyclecheck_stats st; memset(&st, 0, sizeof(st));
        if (plasship_cyclecheck(ctx, db, &p, &o, nullptr, &st)) return fail("cyclecheck");
        fprintf(stdout, "circular: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_cyclic, st.ms_kernel);
        if (plasship_seqdb_write(ctx, o, pos[1].c_str())) return fail("cyclecheck");
        plasship_seqdb_free(ctx, o); plasship_seqdb_free(ctx, db);
    } else {
        fprintf(stdout, "plass-hip: module \"%s\" is not part of the GPU hot path (use the reference binary for it)\n", mod.c_str());
        rc = EXIT_FAILURE;
    }
    fprintf(stdout, "Time for processing: %.3fs\n", now() - t0);
    plasship_ctx_destroy(ctx);
    return rc;
}
