// plass-hip: the reference's module command lines over the plasship C-ABI.
//
// The reference workflow scripts run `"$MMSEQS" <module> <dbs…> <flags…>` (data/assemble.sh:43-145); pointing $MMSEQS at this
// binary for the hot modules makes them run on the MI355X while every database on disk keeps the DBReader/DBWriter format.
// Module names, positional arguments and flag names are the reference's (mm/commons/Parameters.cpp:423-439,640-655,763-767,
// 872-892,970-974; src/commons/LocalParameters.h:96-117,171-176).  Parsing follows Parameters::parseParameters
// (Parameters.cpp:1560-1700): a flag the module does not own is an error, a bool flag without a value toggles its default, and
// defaults are the MODULE's defaults (Parameters::setDefaults, setLinearFilterDefault), not the assemble workflow's — a call
// that does not come from createParameterString therefore means the same thing here and there.  Flags that cannot influence
// the result (--threads, -v, --db-load-mode, --compressed 0 …) are accepted and ignored; values this build cannot honour
// (another substitution matrix, spaced k-mers, masking, automatic k …) fail loudly like Debug(Debug::ERROR) + EXIT(EXIT_FAILURE).
#include "../../include/plasship.h"
#include <chrono>
#include <unistd.h>
#include <cstdarg>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <sys/stat.h>
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
static std::string baseName(const std::string &p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? p : p.substr(s + 1); }
static int frameMask(const std::string &v, bool &ok) {                   // Orf::getFrames: "1,2,3"
    int m = 0; size_t p = 0; ok = true;
    while (p <= v.size()) {
        size_t c = v.find(',', p); if (c == std::string::npos) c = v.size();
        const std::string t = v.substr(p, c - p);
        if (!t.empty()) { const int f = atoi(t.c_str()); if (f < 1 || f > 3) ok = false; else m |= 1 << (f - 1); }
        p = c + 1;
    }
    return m;
}

struct Flags {
    // Parameters::setDefaults (Parameters.cpp:2093-2338); kmermatcher additionally setLinearFilterDefault (kmermatcher.cpp:566-573)
    int k = 0, alph = 21, kps = 21, hashShift = 67, onlyExt = 0, ignoreMulti = 0, covMode = 0;
    float scaleAA = 0.0f, scaleNucl = 0.2f, covThr = 0.0f, seqIdThr = 0.0f;
    int rescoreMode = 0, minAlnLen = 0, seqIdMode = 0, addBt = 0, addSelf = 0, keepTarget = 1, wrapped = 0, filterHits = 0, sortResults = 0;
    double evalThr = 1e-3;
    unsigned long long maxSeqLen = 65535;
    int gapOpenNucl = 5, gapExtendNucl = 2;
    int chopCycle = 1;                   // LocalParameters.h:202
    int orfMin = 30, orfMax = 32734, orfGaps = INT_MAX, contigStart = 2, contigEnd = 2, orfStart = 1, fwdFrames = 7, revFrames = 7;
    int translationTable = 1, translate = 0, allStarts = 0, addOrfStop = 0, preserveKeys = 0, takeLarger = 0;
    int numIterations = 0, fromReads = 0; std::string writeIntermediate; float seqIdThrNucl = 0.99f;
    std::set<std::string> seen;
};

// Exit codes beyond EXIT_SUCCESS / EXIT_FAILURE: a request that is well-formed for the REFERENCE module but lies
// outside what the GPU path implements (--rescore-mode 0, --wrapped-scoring 1, masking, spaced k-mers, automatic -k, a module that is not
// part of the hot path, PLASSHIP_ERR_UNSUPPORTED from the library) ends with EXIT_UNSUPPORTED before anything is written, so that a
// wrapper can hand exactly that call to the reference binary: `linclust` at the end of `penguin guided_nuclassemble` calls
// `rescorediagonal --rescore-mode 0 --wrapped-scoring 1` (lib/mmseqs/data/workflow/linclust.sh:30) through the same $MMSEQS.
// EXIT_DRYRUN_ACCEPTED: with PLASSHIP_CLI_DRYRUN=1 the command line is parsed and validated, nothing is read or computed (no GPU needed) —
// what tools/workflow_dropin_check.sh uses to exercise the routing of every call the unmodified workflow scripts make.
enum { EXIT_UNSUPPORTED = 95, EXIT_DRYRUN_ACCEPTED = 96 };
static int g_lastRc = 0;
static bool g_wrote = false;             // an output DB of this call exists on disk (a completed *_write): exit code 95 would make a wrapper re-run the call over it
static inline int K(int rc) { g_lastRc = rc; return rc; }
static inline int KW(int rc) { g_lastRc = rc; if (rc == 0) g_wrote = true; return rc; }      // K() for the calls that write a DB
static int unsupported(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stdout, fmt, ap); va_end(ap);
    fprintf(stdout, "plass-hip: outside the GPU hot path, exit code %d (nothing was written)\n", (int) EXIT_UNSUPPORTED);
    return EXIT_UNSUPPORTED;
}
static int fail(const char *what) {
    fprintf(stdout, "%s: %s\n", what, plasship_last_error());
    if (g_lastRc == PLASSHIP_ERR_UNSUPPORTED && !g_wrote) { fprintf(stdout, "plass-hip: outside the GPU hot path, exit code %d (nothing was written)\n", (int) EXIT_UNSUPPORTED); return EXIT_UNSUPPORTED; }
    // (an "unsupported" that arrives after an output DB of this call was written is an ordinary failure: the files stay as they are and no wrapper
    //  should run the reference over them — ADVICE r5)
    if (g_lastRc == PLASSHIP_ERR_UNSUPPORTED) fprintf(stdout, "plass-hip: part of the output was already written; exit code %d, not %d\n", (int) EXIT_FAILURE, (int) EXIT_UNSUPPORTED);
    return EXIT_FAILURE;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// which flags a module owns (its parameter vector in the reference)
static const std::map<std::string, std::set<std::string>> &moduleFlags() {
    static const std::set<std::string> common = {"--threads", "-v", "--compressed"};
    static std::map<std::string, std::set<std::string>> m;
    if (m.empty()) {
        m["kmermatcher"] = {"--sub-mat", "--alph-size", "--min-seq-id", "--kmer-per-seq", "--spaced-kmer-mode", "--spaced-kmer-pattern", "--kmer-per-seq-scale",
                            "--adjust-kmer-len", "--mask", "--mask-lower-case", "--cov-mode", "-k", "-c", "--max-seq-len", "--hash-shift", "--split-memory-limit",
                            "--include-only-extendable", "--ignore-multi-kmer"};
        m["rescorediagonal"] = {"--sub-mat", "--rescore-mode", "--wrapped-scoring", "--filter-hits", "-e", "-c", "-a", "--cov-mode", "--min-seq-id", "--min-aln-len",
                                "--seq-id-mode", "--add-self-matches", "--sort-results", "--db-load-mode"};
        m["assembleresults"] = {"--min-seq-id", "--max-seq-len", "--keep-target", "--rescore-mode"};
        m["nuclassembleresults"] = m["assembleresults"];
        m["guidedassembleresults"] = m["assembleresults"];
        m["proteinaln2nucl"] = {"--sub-mat", "--gap-open", "--gap-extend"};
        m["findassemblystart"] = {};
        m["cyclecheck"] = {"--max-seq-len", "--chop-cycle"};
        m["extractorfs"] = {"--min-length", "--max-length", "--max-gaps", "--contig-start-mode", "--contig-end-mode", "--orf-start-mode", "--forward-frames",
                            "--reverse-frames", "--translation-table", "--translate", "--use-all-table-starts", "--id-offset", "--create-lookup"};
        m["translatenucs"] = {"--translation-table", "--add-orf-stop"};
        m["concatdbs"] = {"--preserve-keys", "--take-larger-entry"};
        // the fused drivers (not reference modules: the iteration loops of data/assemble.sh:85-156, nuclassemble.sh:95-137 and
        // guidedNuclAssemble.sh:77-126 with the DBs chained in HBM); flags = the workflow's own, defaults = the workflow's
        m["assemble-chain"] = {"--num-iterations", "--write-intermediate", "--from-reads", "-k", "--alph-size", "--kmer-per-seq", "--kmer-per-seq-scale", "--min-seq-id", "-e", "-c",
                               "--cov-mode", "--max-seq-len", "--keep-target", "--hash-shift", "--ignore-multi-kmer", "--rescore-mode", "--min-aln-len", "--seq-id-mode"};
        m["nuclassemble-chain"] = m["assemble-chain"]; m["nuclassemble-chain"].insert("--chop-cycle");
        m["guidedassemble-chain"] = m["assemble-chain"];
        for (auto &kv : m) kv.second.insert(common.begin(), common.end());
    }
    return m;
}
// bool-typed parameters of the reference (typeid(bool)): a missing value toggles the default (Parameters.cpp:1670-1677)
static const std::set<std::string> &boolFlags() {
    static const std::set<std::string> b = {"-a", "--add-self-matches", "--wrapped-scoring", "--filter-hits", "--include-only-extendable", "--ignore-multi-kmer",
                                            "--keep-target", "--chop-cycle", "--adjust-kmer-len", "--use-all-table-starts", "--add-orf-stop", "--preserve-keys",
                                            "--take-larger-entry"};
    return b;
}
static bool parseBool(const std::string &v, bool &ok) {                 // Parameters::parseBool: TRUE/1 | FALSE/0
    ok = true;
    if (v == "1" || v == "TRUE" || v == "true") return true;
    if (v == "0" || v == "FALSE" || v == "false") return false;
    ok = false; return false;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stdout, "usage: plass-hip <kmermatcher|rescorediagonal|assembleresults|nuclassembleresults|guidedassembleresults|proteinaln2nucl|findassemblystart|"
                        "cyclecheck|extractorfs|translatenucs|concatdbs> <dbs…> [flags]\n"
                        "       plass-hip assemble-chain <i:fragmentDB|readDB> <o:assemblyDB> [--num-iterations 12] [--from-reads 1] [--write-intermediate DIR]\n"
                        "       plass-hip nuclassemble-chain <i:nuclDB> <o:assemblyDB> [--num-iterations 8]        (writes <o>_cycle_<i> for circular contigs)\n"
                        "       plass-hip guidedassemble-chain <i:readDB> <o:nuclAssemblyDB> <o:aaAssemblyDB> [--num-iterations 5]\n");
        return EXIT_FAILURE;
    }
    const std::string mod = argv[1];
    const auto mf = moduleFlags().find(mod);
    if (mf == moduleFlags().end()) {
        return unsupported("plass-hip: module \"%s\" is not part of the GPU hot path (use the reference binary for it)\n", mod.c_str());
    }
    Flags f; std::vector<std::string> pos;
    if (mod == "kmermatcher") { f.covThr = 0.8f; f.alph = 13; f.kps = 0; }                    // setLinearFilterDefault
    const bool chain = mod == "assemble-chain" || mod == "nuclassemble-chain" || mod == "guidedassemble-chain";
    if (chain) {
        // the workflows' defaults: src/workflow/Assembler.cpp:10-27 (plass assemble), Nuclassembler.cpp:10-31 (penguin nuclassemble),
        // GuidedNuclassembler.cpp:10-41 (penguin guided_nuclassemble, protein-guided stage)
        f.alph = 13; f.kps = 60; f.ignoreMulti = 1; f.covThr = 0.0f; f.covMode = 0; f.evalThr = 1e-5; f.rescoreMode = 3; f.keepTarget = 1; f.hashShift = 67;
        if (mod == "assemble-chain") { f.k = 14; f.scaleAA = 0.0f; f.seqIdThr = 0.9f; f.maxSeqLen = 65535; f.numIterations = 12; }
        else if (mod == "nuclassemble-chain") { f.k = 22; f.scaleNucl = 0.1f; f.seqIdThr = 0.99f; f.maxSeqLen = 200000; f.numIterations = 8; f.onlyExt = 1; f.chopCycle = 1; }
        else { f.k = 14; f.scaleAA = 0.1f; f.seqIdThr = 0.97f; f.maxSeqLen = 200000; f.numIterations = 5; f.onlyExt = 1; f.covMode = 1; f.addBt = 1; }
    }
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (!(a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9'))) { pos.push_back(a); continue; }
        if (!mf->second.count(a)) { fprintf(stdout, "Unrecognized parameter \"%s\" for module %s\n", a.c_str(), mod.c_str()); return EXIT_FAILURE; }
        std::string v, t; bool haveValue = true;
        if (boolFlags().count(a)) {
            if (i + 1 >= argc || argv[i + 1][0] == '-') haveValue = false; else v = argv[++i];
        } else {
            if (i + 1 >= argc) { fprintf(stdout, "Missing argument %s\n", a.c_str()); return EXIT_FAILURE; }
            v = argv[++i];
        }
        f.seen.insert(a);
        auto setBool = [&](int &field) -> bool {
            if (!haveValue) { field = !field; return true; }
            bool ok; const bool b = parseBool(v, ok);
            if (!ok) { fprintf(stdout, "Error in argument %s\n", a.c_str()); return false; }
            field = b ? 1 : 0; return true;
        };
        if (a == "-k") f.k = atoi(v.c_str());
        else if (a == "--num-iterations") { std::string t2; f.numIterations = atoi((multiParam(v, mod == "assemble-chain" || mod == "guidedassemble-chain" ? "aa" : "nucl", t2) ? t2 : v).c_str()); }
        else if (a == "--write-intermediate") f.writeIntermediate = v;
        else if (a == "--from-reads") f.fromReads = atoi(v.c_str());
        else if (a == "--alph-size") { if (multiParam(v, "aa", t)) f.alph = atoi(t.c_str()); }
        else if (a == "--kmer-per-seq") f.kps = atoi(v.c_str());
        else if (a == "--kmer-per-seq-scale") { if (multiParam(v, "aa", t)) f.scaleAA = strtof(t.c_str(), nullptr); if (multiParam(v, "nucl", t)) f.scaleNucl = strtof(t.c_str(), nullptr); }
        else if (a == "--hash-shift") f.hashShift = atoi(v.c_str());
        else if (a == "--include-only-extendable") { if (!setBool(f.onlyExt)) return EXIT_FAILURE; }
        else if (a == "--ignore-multi-kmer") { if (!setBool(f.ignoreMulti)) return EXIT_FAILURE; }
        else if (a == "--cov-mode") f.covMode = atoi(v.c_str());
        else if (a == "-c") f.covThr = strtof(v.c_str(), nullptr);
        else if (a == "--rescore-mode") f.rescoreMode = atoi(v.c_str());
        else if (a == "-e") f.evalThr = strtod(v.c_str(), nullptr);
        else if (a == "--min-seq-id") {
            if (mod == "guidedassemble-chain") { if (multiParam(v, "aa", t)) f.seqIdThr = strtof(t.c_str(), nullptr); if (multiParam(v, "nucl", t)) f.seqIdThrNucl = strtof(t.c_str(), nullptr); }
            else if (mod == "nuclassemble-chain") { if (multiParam(v, "nucl", t)) f.seqIdThr = strtof(t.c_str(), nullptr); }
            else if (mod == "assemble-chain") { if (multiParam(v, "aa", t)) f.seqIdThr = strtof(t.c_str(), nullptr); }
            else f.seqIdThr = strtof(v.c_str(), nullptr);
        }
        else if (a == "--min-aln-len") f.minAlnLen = atoi(v.c_str());
        else if (a == "--seq-id-mode") f.seqIdMode = atoi(v.c_str());
        else if (a == "-a") { if (!setBool(f.addBt)) return EXIT_FAILURE; }
        else if (a == "--add-self-matches") { if (!setBool(f.addSelf)) return EXIT_FAILURE; }
        else if (a == "--max-seq-len") f.maxSeqLen = strtoull(v.c_str(), nullptr, 10);
        else if (a == "--chop-cycle") { if (!setBool(f.chopCycle)) return EXIT_FAILURE; }
        else if (a == "--keep-target") { if (!setBool(f.keepTarget)) return EXIT_FAILURE; }
        else if (a == "--gap-open") { if (multiParam(v, "nucl", t)) f.gapOpenNucl = atoi(t.c_str()); }
        else if (a == "--gap-extend") { if (multiParam(v, "nucl", t)) f.gapExtendNucl = atoi(t.c_str()); }
        else if (a == "--wrapped-scoring") { if (!setBool(f.wrapped)) return EXIT_FAILURE; }
        else if (a == "--filter-hits") { if (!setBool(f.filterHits)) return EXIT_FAILURE; }
        else if (a == "--sort-results") f.sortResults = atoi(v.c_str());
        else if (a == "--min-length") f.orfMin = atoi(v.c_str());
        else if (a == "--max-length") f.orfMax = atoi(v.c_str());
        else if (a == "--max-gaps") f.orfGaps = atoi(v.c_str());
        else if (a == "--contig-start-mode") f.contigStart = atoi(v.c_str());
        else if (a == "--contig-end-mode") f.contigEnd = atoi(v.c_str());
        else if (a == "--orf-start-mode") f.orfStart = atoi(v.c_str());
        else if (a == "--forward-frames" || a == "--reverse-frames") {
            bool ok; const int m = frameMask(v, ok);
            if (!ok) { fprintf(stdout, "Error in argument %s\n", a.c_str()); return EXIT_FAILURE; }
            (a == "--forward-frames" ? f.fwdFrames : f.revFrames) = m;
        }
        else if (a == "--translation-table") f.translationTable = atoi(v.c_str());
        else if (a == "--translate") f.translate = atoi(v.c_str());
        else if (a == "--use-all-table-starts") { if (!setBool(f.allStarts)) return EXIT_FAILURE; }
        else if (a == "--add-orf-stop") { if (!setBool(f.addOrfStop)) return EXIT_FAILURE; }
        else if (a == "--preserve-keys") { if (!setBool(f.preserveKeys)) return EXIT_FAILURE; }
        else if (a == "--take-larger-entry") { if (!setBool(f.takeLarger)) return EXIT_FAILURE; }
        else if (a == "--sub-mat") {
            // only the matrices the tables were captured from: blosum62.out (amino acids) and nucleotide.out
            std::string aa, nu;
            const bool okA = multiParam(v, "aa", aa), okN = multiParam(v, "nucl", nu);
            if ((okA && baseName(aa) != "blosum62.out" && v.find(':') != std::string::npos) || (okN && baseName(nu) != "nucleotide.out" && v.find(':') != std::string::npos) ||
                (v.find(':') == std::string::npos && baseName(v) != "blosum62.out" && baseName(v) != "nucleotide.out")) {
                return unsupported("plass-hip: --sub-mat %s is not supported (built for blosum62.out / nucleotide.out)\n", v.c_str());
            }
        }
        else if (a == "--spaced-kmer-mode" || a == "--mask" || a == "--mask-lower-case" || a == "--compressed" || a == "--create-lookup" || a == "--id-offset") {
            if (atoi(v.c_str()) != 0) return unsupported("%s %s is not supported by plass-hip\n", a.c_str(), v.c_str());
        }
        else if (a == "--adjust-kmer-len") { int x = 0; if (!setBool(x)) return EXIT_FAILURE; if (x) return unsupported("--adjust-kmer-len is not supported by plass-hip\n"); }
        else if (a == "--spaced-kmer-pattern") { if (!v.empty()) return unsupported("--spaced-kmer-pattern is not supported by plass-hip\n"); }
        // --threads, -v, --db-load-mode, --split-memory-limit: no influence on the result
    }
    if (f.wrapped || f.filterHits || f.sortResults) return unsupported("--wrapped-scoring/--filter-hits/--sort-results are not supported by plass-hip\n");
    if (mod == "kmermatcher") {
        // the module's own defaults for these two are "choose automatically" (k = 0: from the DB size, kmermatcher.cpp:607-613;
        // --kmer-per-seq 0): not reproduced — every workflow passes them
        if (f.k <= 0 || f.kps <= 0) return unsupported("plass-hip kmermatcher: -k and --kmer-per-seq must be given (the automatic choice of the reference is not implemented)\n");
        // the reference's module default is spaced k-mers and masking ON unless told otherwise; the workflows run with both off
        // (setLinearFilterDefault turns them off for kmermatcher itself)
    }
    if (mod == "rescorediagonal" && f.rescoreMode != 3)
        return unsupported("plass-hip rescorediagonal: --rescore-mode %d is not part of the GPU path (the assembly workflows use mode 3)\n", f.rescoreMode);
    // Requests that are valid for the reference and lie outside the GPU path are all refused HERE, before the dry-run exit and before anything is
    // read or written (data/nuclassemble.sh:41,145 calls `concatdbs ... --preserve-keys` whenever circular contigs exist; with exit
    // code 1 the wrapper ended the workflow instead of handing the call to the reference, and the dry run reported it as accepted)
    if (mod == "concatdbs") {
        if (f.takeLarger) return unsupported("plass-hip concatdbs: --take-larger-entry is not part of the GPU path\n");
        if (pos.size() == 3) {      // sequence DBs, or the header DBs of ORF DBs (dbtype 12: data/assemble.sh:75); a DB that is not there yet cannot be probed (dry run on names only)
            const std::string tp = pos[0] + ".dbtype"; FILE *ft = fopen(tp.c_str(), "rb"); unsigned ty = 0;
            if (ft) {
                const bool got = fread(&ty, 4, 1, ft) == 1; fclose(ft);
                const int dbtype = got ? (int) (ty & 0x3FFFFFFFu) : -1;
                if (got && dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS && dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES && dbtype != 12)
                    return unsupported("plass-hip concatdbs: database type %d is not part of the GPU path (sequence DBs and ORF header DBs only)\n", dbtype);
                if (got && dbtype == 12 && f.preserveKeys) return unsupported("plass-hip concatdbs: --preserve-keys on a header DB is not part of the GPU path\n");
            }
        }
    }
    if (mod == "proteinaln2nucl" && pos.size() == 6 && (pos[0] == pos[1]) != (pos[2] == pos[3])) { fprintf(stdout, "Either query database == target database for nucleotide and amino acid or != for both\n"); return EXIT_FAILURE; }
    if (mod == "proteinaln2nucl" && pos.size() == 6 && pos[0] != pos[1])
        return unsupported("plass-hip proteinaln2nucl: separate query and target DBs are not part of the GPU path (the assembly workflows use one DB)\n");
    if (getenv("PLASSHIP_CLI_DRYRUN") && atoi(getenv("PLASSHIP_CLI_DRYRUN")) != 0) {
        fprintf(stdout, "plass-hip dry run: %s accepted (%zu positional arguments, %zu flags); nothing read or computed\n", mod.c_str(), pos.size(), f.seen.size());
        return EXIT_DRYRUN_ACCEPTED;
    }
    plasship_ctx *ctx = nullptr;
    if (K(plasship_ctx_create(-1, &ctx))) return fail("plass-hip");
    const double t0 = now();
    int rc = EXIT_SUCCESS;
    if (mod == "kmermatcher") {
        if (pos.size() != 2) { fprintf(stdout, "kmermatcher <i:sequenceDB> <o:prefDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr; plasship_cands *c = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &db))) return fail("kmermatcher");
        int dbtype = 0; plasship_seqdb_info(db, nullptr, nullptr, nullptr, &dbtype, nullptr);
        plasship_kmermatch_params p; memset(&p, 0, sizeof(p));
        p.kmer_size = f.k; p.alphabet_size = f.alph; p.kmers_per_seq = f.kps; p.kmers_per_seq_scale = (dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES) ? f.scaleNucl : f.scaleAA;
        p.hash_shift = f.hashShift; p.include_only_extendable = f.onlyExt; p.ignore_multi_kmer = f.ignoreMulti; p.cov_mode = f.covMode; p.cov_thr = f.covThr;
        plasship_kmermatch_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_kmermatch(ctx, db, &p, &c, &st))) return fail("kmermatcher");
        fprintf(stdout, "k-mer records: %llu grouped: %llu candidates: %llu | kernels ms: extract %.3f partition %.3f group %.3f sort %.3f reduce %.3f\n",
                (unsigned long long) st.n_kmer_records, (unsigned long long) st.n_grouped, (unsigned long long) st.n_candidates,
                st.ms_extract, st.ms_sort1, st.ms_group, st.ms_sort2, st.ms_reduce);
        if (KW(plasship_cands_write(ctx, c, db, pos[1].c_str()))) return fail("kmermatcher");
        plasship_cands_free(ctx, c); plasship_seqdb_free(ctx, db);
    } else if (mod == "rescorediagonal") {
        if (pos.size() != 4) { fprintf(stdout, "rescorediagonal <i:queryDB> <i:targetDB> <i:prefDB> <o:alnDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *q = nullptr, *t = nullptr; plasship_cands *c = nullptr; plasship_alns *al = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &q))) return fail("rescorediagonal");
        if (pos[1] == pos[0]) t = q; else if (K(plasship_seqdb_read(ctx, pos[1].c_str(), &t))) return fail("rescorediagonal");
        if (K(plasship_cands_read(ctx, q, t, pos[2].c_str(), &c))) return fail("rescorediagonal");
        plasship_rescore_params p; memset(&p, 0, sizeof(p));
        p.rescore_mode = f.rescoreMode; p.eval_thr = f.evalThr; p.seq_id_thr = f.seqIdThr; p.cov_mode = f.covMode; p.cov_thr = f.covThr;
        p.min_aln_len = f.minAlnLen; p.seq_id_mode = f.seqIdMode; p.add_backtrace = f.addBt; p.include_identity = f.addSelf;
        plasship_rescore_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_rescore(ctx, q, t, c, &p, &al, &st))) return fail("rescorediagonal");
        fprintf(stdout, "scored: %llu accepted: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_scored, (unsigned long long) st.n_accepted, st.ms_kernel);
        if (KW(plasship_alns_write(ctx, al, pos[3].c_str()))) return fail("rescorediagonal");
        plasship_alns_free(ctx, al); plasship_cands_free(ctx, c); if (t != q) plasship_seqdb_free(ctx, t); plasship_seqdb_free(ctx, q);
    } else if (mod == "assembleresults" || mod == "nuclassembleresults") {
        if (pos.size() != 3) { fprintf(stdout, "%s <i:sequenceDB> <i:alnResult> <o:reprSeqDB>\n", mod.c_str()); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr; plasship_alns *al = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &db))) return fail("assembleresults");
        // the library picks the variant from the DB type (protein -> assembleresults, nucleotide -> nuclassembleresults);
        // the module name has to agree, the reference's assembleresults on nucleotides uses another comparator
        int dbtype = -1; plasship_seqdb_info(db, nullptr, nullptr, nullptr, &dbtype, nullptr);
        if ((mod == "nuclassembleresults") != (dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES)) {
            fprintf(stdout, "plass-hip: %s needs a %s sequence DB\n", mod.c_str(), mod == "nuclassembleresults" ? "nucleotide" : "protein");
            return EXIT_FAILURE;
        }
        if (K(plasship_alns_read(ctx, db, pos[1].c_str(), &al))) return fail("assembleresults");
        plasship_assemble_params p; memset(&p, 0, sizeof(p));
        p.seq_id_thr = f.seqIdThr; p.max_seq_len = f.maxSeqLen; p.keep_target = f.keepTarget; p.rescore_mode = f.rescoreMode;
        plasship_assemble_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_assemble(ctx, db, al, &p, &o, &st))) return fail("assembleresults");
        fprintf(stdout, "extended: %llu rescored: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_extended, (unsigned long long) st.n_rescored, st.ms_kernel);
        if (KW(plasship_seqdb_write(ctx, o, pos[2].c_str()))) return fail("assembleresults");
        plasship_seqdb_free(ctx, o); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, db);
    } else if (mod == "guidedassembleresults") {
        if (pos.size() != 5) { fprintf(stdout, "guidedassembleresults <i:nuclSequenceDB> <i:aaSequenceDB> <i:nuclAlnResult> <o:nuclAssembly> <o:aaAssembly>\n"); return EXIT_FAILURE; }
        plasship_seqdb *nu = nullptr, *aa = nullptr, *on = nullptr, *oa = nullptr; plasship_alns *al = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &nu)) || K(plasship_seqdb_read(ctx, pos[1].c_str(), &aa))) return fail("guidedassembleresults");
        if (K(plasship_alns_read(ctx, nu, pos[2].c_str(), &al))) return fail("guidedassembleresults");
        plasship_assemble_params p; memset(&p, 0, sizeof(p));
        p.seq_id_thr = f.seqIdThr; p.max_seq_len = f.maxSeqLen; p.keep_target = f.keepTarget; p.rescore_mode = f.rescoreMode;
        plasship_assemble_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_guided_assemble(ctx, nu, aa, al, &p, &on, &oa, &st))) return fail("guidedassembleresults");
        fprintf(stdout, "extended: %llu rescored: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_extended, (unsigned long long) st.n_rescored, st.ms_kernel);
        if (KW(plasship_seqdb_write(ctx, on, pos[3].c_str())) || KW(plasship_seqdb_write(ctx, oa, pos[4].c_str()))) return fail("guidedassembleresults");
        plasship_seqdb_free(ctx, on); plasship_seqdb_free(ctx, oa); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, aa); plasship_seqdb_free(ctx, nu);
    } else if (mod == "proteinaln2nucl") {
        if (pos.size() != 6) { fprintf(stdout, "proteinaln2nucl <i:queryNuclDB> <i:targetNuclDB> <i:queryAaDB> <i:targetAaDB> <i:alnDB> <o:alnDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *nu = nullptr, *aa = nullptr; plasship_alns *al = nullptr, *o = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &nu)) || K(plasship_seqdb_read(ctx, pos[2].c_str(), &aa))) return fail("proteinaln2nucl");
        if (K(plasship_alns_read(ctx, aa, pos[4].c_str(), &al))) return fail("proteinaln2nucl");
        plasship_aln2nucl_params p; p.gap_open = f.gapOpenNucl; p.gap_extend = f.gapExtendNucl;
        plasship_aln2nucl_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_aln2nucl(ctx, nu, nu, aa, aa, al, &p, &o, &st))) return fail("proteinaln2nucl");
        fprintf(stdout, "alignments: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_alignments, st.ms_kernel);
        if (KW(plasship_alns_write(ctx, o, pos[5].c_str()))) return fail("proteinaln2nucl");
        plasship_alns_free(ctx, o); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, aa); plasship_seqdb_free(ctx, nu);
    } else if (mod == "findassemblystart") {
        if (pos.size() != 3) { fprintf(stdout, "findassemblystart <i:sequenceDB> <i:alnDB> <o:sequenceDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr; plasship_alns *al = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &db))) return fail("findassemblystart");
        if (K(plasship_alns_read(ctx, db, pos[1].c_str(), &al))) return fail("findassemblystart");
        plasship_findstart_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_find_assembly_start(ctx, db, al, &o, &st))) return fail("findassemblystart");
        fprintf(stdout, "alignments: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_alignments, st.ms_kernel);
        if (KW(plasship_seqdb_write(ctx, o, pos[2].c_str()))) return fail("findassemblystart");
        plasship_seqdb_free(ctx, o); plasship_alns_free(ctx, al); plasship_seqdb_free(ctx, db);
    } else if (mod == "cyclecheck") {
        if (pos.size() != 2) { fprintf(stdout, "cyclecheck <i:sequenceDB> <o:sequenceDBcycle>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &db))) return fail("cyclecheck");
        plasship_cyclecheck_params p; p.max_seq_len = f.maxSeqLen; p.chop_cycle = f.chopCycle;
        plasship_cyclecheck_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_cyclecheck(ctx, db, &p, &o, nullptr, &st))) return fail("cyclecheck");
        fprintf(stdout, "circular: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_cyclic, st.ms_kernel);
        if (KW(plasship_seqdb_write(ctx, o, pos[1].c_str()))) return fail("cyclecheck");
        plasship_seqdb_free(ctx, o); plasship_seqdb_free(ctx, db);
    } else if (mod == "extractorfs") {
        // writes <out> and <out>_h like the reference (extractorfs.cpp:28-32)
        if (pos.size() != 2) { fprintf(stdout, "extractorfs <i:sequenceDB> <o:sequenceDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr; plasship_orfhdr *h = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &db))) return fail("extractorfs");
        plasship_orf_params p; memset(&p, 0, sizeof(p));
        p.min_length = f.orfMin; p.max_length = f.orfMax; p.max_gaps = f.orfGaps; p.contig_start_mode = f.contigStart; p.contig_end_mode = f.contigEnd;
        p.orf_start_mode = f.orfStart; p.forward_frames = f.fwdFrames; p.reverse_frames = f.revFrames; p.translation_table = f.translationTable;
        p.translate = f.translate; p.use_all_table_starts = f.allStarts; p.max_seq_len = f.maxSeqLen;
        plasship_orf_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_extract_orfs(ctx, db, &p, &o, &h, &st))) return fail("extractorfs");
        fprintf(stdout, "orfs: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_out, st.ms_kernel);
        if (KW(plasship_seqdb_write(ctx, o, pos[1].c_str())) || KW(plasship_orfhdr_write(ctx, h, (pos[1] + "_h").c_str()))) return fail("extractorfs");
        plasship_orfhdr_free(ctx, h); plasship_seqdb_free(ctx, o); plasship_seqdb_free(ctx, db);
    } else if (mod == "translatenucs") {
        if (pos.size() != 2) { fprintf(stdout, "translatenucs <i:sequenceDB> <o:sequenceDB>\n"); return EXIT_FAILURE; }
        plasship_seqdb *db = nullptr, *o = nullptr; plasship_orfhdr *h = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &db))) return fail("translatenucs");
        if (f.addOrfStop && K(plasship_orfhdr_read(ctx, (pos[0] + "_h").c_str(), &h))) return fail("translatenucs");      // translatenucs.cpp:29-34
        plasship_translate_params p; p.translation_table = f.translationTable; p.add_orf_stop = f.addOrfStop; p.max_seq_len = f.maxSeqLen;
        plasship_orf_stats st; memset(&st, 0, sizeof(st));
        if (K(plasship_translate_nucs(ctx, db, h, &p, &o, &st))) return fail("translatenucs");
        fprintf(stdout, "translated: %llu | kernel ms: %.3f\n", (unsigned long long) st.n_out, st.ms_kernel);
        if (KW(plasship_seqdb_write(ctx, o, pos[1].c_str()))) return fail("translatenucs");
        plasship_orfhdr_free(ctx, h); plasship_seqdb_free(ctx, o); plasship_seqdb_free(ctx, db);
    } else if (mod == "concatdbs") {
        if (pos.size() != 3) { fprintf(stdout, "concatdbs <i:DB> <i:DB> <o:DB>\n"); return EXIT_FAILURE; }
        // sequence DBs, or the header DBs of ORF DBs (dbtype 12: data/assemble.sh:75)
        int dbtype = -1;
        { std::string tp = pos[0] + ".dbtype"; FILE *ft = fopen(tp.c_str(), "rb"); unsigned ty = 0; if (ft && fread(&ty, 4, 1, ft) == 1) dbtype = (int) (ty & 0x3FFFFFFFu); if (ft) fclose(ft); }
        if (dbtype == PLASSHIP_DBTYPE_AMINO_ACIDS || dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES) {
            plasship_seqdb *a = nullptr, *b = nullptr, *o = nullptr;
            if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &a)) || K(plasship_seqdb_read(ctx, pos[1].c_str(), &b))) return fail("concatdbs");
            if (K(plasship_seqdb_concat_keys(ctx, a, b, f.preserveKeys, &o))) return fail("concatdbs");
            if (KW(plasship_seqdb_write(ctx, o, pos[2].c_str()))) return fail("concatdbs");
            plasship_seqdb_free(ctx, o); plasship_seqdb_free(ctx, b); plasship_seqdb_free(ctx, a);
        } else if (dbtype == 12) {
            plasship_orfhdr *a = nullptr, *b = nullptr, *o = nullptr;
            if (K(plasship_orfhdr_read(ctx, pos[0].c_str(), &a)) || K(plasship_orfhdr_read(ctx, pos[1].c_str(), &b))) return fail("concatdbs");
            if (K(plasship_orfhdr_concat(ctx, a, b, &o))) return fail("concatdbs");
            if (KW(plasship_orfhdr_write(ctx, o, pos[2].c_str()))) return fail("concatdbs");
            plasship_orfhdr_free(ctx, o); plasship_orfhdr_free(ctx, b); plasship_orfhdr_free(ctx, a);
        } else { fprintf(stdout, "plass-hip concatdbs: cannot read the database type of %s\n", pos[0].c_str()); return EXIT_FAILURE; }      // (other types were refused above)
    } else if (chain) {
        // ---- fused drivers: the iteration loop of a workflow script with every DB of the loop resident in HBM; only the first DB is
        //      read from disk and only the last one written (plus, with --write-intermediate DIR, every iteration's assembly by a host
        //      thread on a second context while the next iteration runs, with the workflow's .done sentinels) ----
        const bool prot = mod == "assemble-chain", nuc = mod == "nuclassemble-chain", gd = mod == "guidedassemble-chain";
        if (pos.size() != (gd ? 3u : 2u)) { fprintf(stdout, "%s: wrong number of databases\n", mod.c_str()); return EXIT_FAILURE; }
        if (f.numIterations < 1) { fprintf(stdout, "--num-iterations must be at least 1\n"); return EXIT_FAILURE; }
        plasship_ctx *wctx = nullptr; std::thread writer; int writerRc = 0; std::string writerErr;
        // every exit path below joins the writer first (a joinable std::thread that goes out of scope terminates the process, and an
        // intermediate DB must not be left half written behind an ordinary error message: ADVICE r3)
        struct JoinOnExit { std::thread &t; ~JoinOnExit() { if (t.joinable()) t.join(); } } joinOnExit{writer};
        if (!f.writeIntermediate.empty() && K(plasship_ctx_create(-1, &wctx))) return fail(mod.c_str());
        auto joinWriter = [&]() { if (writer.joinable()) writer.join(); return writerRc; };
        auto writeAsync = [&](const plasship_seqdb *d, const std::string &name) {
            if (!wctx) return;
            joinWriter();
            const std::string path = f.writeIntermediate + "/" + name;
            writer = std::thread([&, d, path]() {
                if (plasship_seqdb_write(wctx, d, path.c_str())) { writerRc = 1; writerErr = plasship_last_error(); return; }      // (the error text is the writer THREAD's)
                FILE *fd = fopen((path + ".done").c_str(), "w"); if (fd) fclose(fd); else writerRc = 1;     // data/assemble.sh:147 `touch assembly_$STEP.done`
            });
        };
        // a multi-GB input: the library takes its device arena (seconds of hipMalloc) while this thread reads and parses the DB files
        { struct stat stIn; if (stat(pos[0].c_str(), &stIn) == 0 && stIn.st_size >= ((off_t) 1 << 30)) (void) plasship_ctx_reserve_async(ctx); }
        plasship_seqdb *in = nullptr;
        if (K(plasship_seqdb_read(ctx, pos[0].c_str(), &in))) return fail(mod.c_str());
        const double tRead = now();
        int dbtype = -1; plasship_seqdb_info(in, nullptr, nullptr, nullptr, &dbtype, nullptr);
        auto orfPar = [&](bool start) {       // the two extractorfs passes (Assembler.cpp:116-130, GuidedNuclassembler.cpp:133-145)
            plasship_orf_params p; memset(&p, 0, sizeof(p));
            p.min_length = start ? 20 : 45; p.max_length = start ? 45 : 32734; p.max_gaps = 0; p.contig_start_mode = start ? 1 : 2; p.contig_end_mode = start ? 0 : 2;
            p.orf_start_mode = 0; p.forward_frames = 7; p.reverse_frames = 7; p.translation_table = 1; p.max_seq_len = 65535;
            return p;
        };
        plasship_translate_params tp; tp.translation_table = 1; tp.add_orf_stop = 1; tp.max_seq_len = 65535;
        plasship_seqdb *db = in, *aa = nullptr;          // protein / nucleotide chain: db; guided chain: db = nucleotide ORFs, aa = their twins
        if (gd || (prot && (f.fromReads || dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES))) {
            if (dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES) { fprintf(stdout, "%s: the input must be a nucleotide read DB\n", mod.c_str()); return EXIT_FAILURE; }
            plasship_seqdb *ol = nullptr, *os = nullptr; plasship_orfhdr *hl = nullptr, *hs = nullptr; plasship_orf_stats ost;
            const plasship_orf_params pl = orfPar(false), ps = orfPar(true);
            if (K(plasship_extract_orfs(ctx, in, &pl, &ol, &hl, &ost)) || K(plasship_extract_orfs(ctx, in, &ps, &os, &hs, &ost))) return fail(mod.c_str());
            if (gd) {      // concatdbs of ORFs and headers, then one translatenucs (data/guidedNuclAssemble.sh:56-75)
                plasship_seqdb *nu = nullptr; plasship_orfhdr *hh = nullptr;
                if (K(plasship_seqdb_concat(ctx, ol, os, &nu)) || K(plasship_orfhdr_concat(ctx, hl, hs, &hh)) || K(plasship_translate_nucs(ctx, nu, hh, &tp, &aa, &ost))) return fail(mod.c_str());
                plasship_orfhdr_free(ctx, hh); db = nu;
            } else {       // translatenucs x2, then concatdbs (data/assemble.sh:41-77)
                plasship_seqdb *al = nullptr, *as = nullptr;
                if (K(plasship_translate_nucs(ctx, ol, hl, &tp, &al, &ost)) || K(plasship_translate_nucs(ctx, os, hs, &tp, &as, &ost)) || K(plasship_seqdb_concat(ctx, al, as, &db))) return fail(mod.c_str());
                plasship_seqdb_free(ctx, al); plasship_seqdb_free(ctx, as);
            }
            plasship_orfhdr_free(ctx, hl); plasship_orfhdr_free(ctx, hs); plasship_seqdb_free(ctx, ol); plasship_seqdb_free(ctx, os); plasship_seqdb_free(ctx, in);
        } else if ((prot && dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS) || (nuc && dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES)) {
            fprintf(stdout, "%s: wrong input DB type %d\n", mod.c_str(), dbtype); return EXIT_FAILURE;
        }
        const double tPrep = now();
        plasship_rescore_params rp; memset(&rp, 0, sizeof(rp));
        rp.rescore_mode = f.rescoreMode; rp.eval_thr = f.evalThr; rp.seq_id_thr = f.seqIdThr; rp.cov_mode = f.covMode; rp.cov_thr = f.covThr; rp.min_aln_len = f.minAlnLen;
        rp.seq_id_mode = f.seqIdMode; rp.add_backtrace = gd ? 1 : 0;
        plasship_assemble_params ap; memset(&ap, 0, sizeof(ap));
        ap.seq_id_thr = gd ? f.seqIdThrNucl : f.seqIdThr; ap.max_seq_len = f.maxSeqLen; ap.keep_target = f.keepTarget; ap.rescore_mode = f.rescoreMode;
        unsigned long long overlaps = 0; double kernelMs = 0;
        int hashShift = f.hashShift;
        for (int it = 0; it < f.numIterations; it++) {
            plasship_kmermatch_params kp; memset(&kp, 0, sizeof(kp));
            kp.kmer_size = f.k; kp.alphabet_size = f.alph; kp.kmers_per_seq = f.kps; kp.kmers_per_seq_scale = nuc ? f.scaleNucl : f.scaleAA; kp.ignore_multi_kmer = f.ignoreMulti;
            kp.cov_mode = f.covMode; kp.cov_thr = f.covThr;
            // plass assemble: --hash-shift grows with every second iteration, iteration 0 keeps non-extendable matches (Assembler.cpp:99-110)
            if (prot) { hashShift += it % 2; kp.hash_shift = hashShift; kp.include_only_extendable = it > 0; } else { kp.hash_shift = f.hashShift; kp.include_only_extendable = 1; }
            plasship_seqdb *q = gd ? aa : db;
            plasship_cands *c = nullptr; plasship_alns *al = nullptr; plasship_kmermatch_stats ks; plasship_rescore_stats rs; plasship_assemble_stats as;
            if (K(plasship_kmermatch(ctx, q, &kp, &c, &ks)) || K(plasship_rescore(ctx, q, q, c, &rp, &al, &rs))) return fail(mod.c_str());
            if (prot && it == 0) {         // data/assemble.sh:110-141: findassemblystart, then k-mer matching and re-scoring again on the corrected sequences
                plasship_seqdb *corr = nullptr; plasship_findstart_stats fs;
                if (K(plasship_find_assembly_start(ctx, db, al, &corr, &fs))) return fail(mod.c_str());
                plasship_alns_free(ctx, al); plasship_cands_free(ctx, c); plasship_seqdb_free(ctx, db); db = corr; q = db;
                if (K(plasship_kmermatch(ctx, q, &kp, &c, &ks)) || K(plasship_rescore(ctx, q, q, c, &rp, &al, &rs))) return fail(mod.c_str());
            }
            overlaps += ks.n_candidates; kernelMs += ks.ms_extract + ks.ms_sort1 + ks.ms_group + ks.ms_sort2 + ks.ms_reduce + rs.ms_kernel;
            plasship_seqdb *next = nullptr, *nextAa = nullptr;
            if (gd) {
                plasship_alns *na = nullptr; plasship_aln2nucl_params np; np.gap_open = f.gapOpenNucl; np.gap_extend = f.gapExtendNucl; plasship_aln2nucl_stats ns;
                if (K(plasship_aln2nucl(ctx, db, db, aa, aa, al, &np, &na, &ns)) || K(plasship_guided_assemble(ctx, db, aa, na, &ap, &next, &nextAa, &as))) return fail(mod.c_str());
                plasship_alns_free(ctx, na);
            } else if (K(plasship_assemble(ctx, db, al, &ap, &next, &as))) return fail(mod.c_str());
            kernelMs += as.ms_kernel;
            plasship_alns_free(ctx, al); plasship_cands_free(ctx, c);
            if (joinWriter()) { fprintf(stdout, "%s: writing an intermediate DB failed: %s\n", mod.c_str(), writerErr.c_str()); return EXIT_FAILURE; }   // the writer read `db`
            plasship_seqdb_free(ctx, db); if (gd) plasship_seqdb_free(ctx, aa);
            db = next; aa = nextAa;
            if (nuc) {     // data/nuclassemble.sh:19-61,132: circular contigs leave the loop, the rest goes on
                plasship_seqdb *cyc = nullptr, *rest = nullptr; plasship_cyclecheck_params cp; cp.max_seq_len = f.maxSeqLen; cp.chop_cycle = f.chopCycle; plasship_cyclecheck_stats cs;
                if (K(plasship_cyclecheck(ctx, db, &cp, &cyc, &rest, &cs))) return fail(mod.c_str());
                if (cs.n_cyclic && KW(plasship_seqdb_write(ctx, cyc, (pos[1] + "_cycle_" + std::to_string(it)).c_str()))) return fail(mod.c_str());
                plasship_seqdb_free(ctx, cyc); plasship_seqdb_free(ctx, db); db = rest;
            }
            fprintf(stdout, "iteration %d: candidates %llu verified %llu extended %llu (%.3f s since the DB was read)\n", it, (unsigned long long) ks.n_candidates, (unsigned long long) rs.n_accepted, (unsigned long long) as.n_extended, now() - tPrep);
            if (it + 1 < f.numIterations) writeAsync(db, (gd ? "assembly_nucl_" : "assembly_") + std::to_string(it));
        }
        const double tLoop = now();
        if (joinWriter()) { fprintf(stdout, "%s: writing an intermediate DB failed: %s\n", mod.c_str(), writerErr.c_str()); return EXIT_FAILURE; }
        if (KW(plasship_seqdb_write(ctx, db, pos[1].c_str())) || (gd && KW(plasship_seqdb_write(ctx, aa, pos[2].c_str())))) return fail(mod.c_str());
        const double tEnd = now();
        fprintf(stdout, "chain: %d iterations, %llu candidate overlaps | read %.3fs preprocessing %.3fs iterations %.3fs (kernels %.3fs) write %.3fs\n", f.numIterations, overlaps,
                tRead - t0, tPrep - tRead, tLoop - tPrep, kernelMs * 1e-3, tEnd - tLoop);
        plasship_seqdb_free(ctx, db); if (gd) plasship_seqdb_free(ctx, aa);
        if (wctx) plasship_ctx_destroy(wctx);
    }
    fprintf(stdout, "Time for processing: %.3fs\n", now() - t0);
    // Every output DB is complete and renamed into place by now.  Taking the context apart — unmapping an arena of up to 270 GB, the runtime's own
    // shutdown — is work the driver does for an exiting process anyway; a command-line tool leaves it to it (round 6: ~1 s of the fused driver's wall
    // clock at 50 M reads).  PLASSHIP_CLI_FULL_TEARDOWN=1: the orderly way (leak checkers).
    fflush(stdout); fflush(stderr);
    if (!getenv("PLASSHIP_CLI_FULL_TEARDOWN")) _exit(rc);
    plasship_ctx_destroy(ctx);
    return rc;
}
