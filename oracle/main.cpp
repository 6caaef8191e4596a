// ORACLE (test infrastructure) command line: the reference module names and flag names
// (mm/commons/Parameters.cpp:423-439,872-892; src/commons/LocalParameters.h:96-102) over the
// CPU restatement.  Usage mirrors `plass <module> …`:
//   plass_oracle kmermatcher <seqDB> <prefDB> [flags]
//   plass_oracle rescorediagonal <qDB> <tDB> <prefDB> <alnDB> [flags]
//   plass_oracle assembleresults|nuclassembleresults <seqDB> <alnDB> <outDB> [flags]
//   plass_oracle guidedassembleresults <nuclDB> <aaDB> <nuclAlnDB> <outNuclDB> <outAaDB> [flags]
//   plass_oracle proteinaln2nucl <qNuclDB> <tNuclDB> <qAaDB> <tAaDB> <alnDB> <outAlnDB> [flags]
//   plass_oracle findassemblystart <seqDB> <alnDB> <outSeqDB>
//   plass_oracle cyclecheck <seqDB> <outCycleDB> [--max-seq-len N --chop-cycle 0|1]
//   plass_oracle synthreads <outReadDB> --pairs N [--seed S --genomes G --genome-min-len A --genome-max-len B --abundance-sigma X …]
//       the synthetic read pairs of include/plasship_synth.h, byte for byte what plasship_synth_read_pairs makes on the GPU
//   plass_oracle dbsum <DB>…   entries, data bytes and an order-independent digest (sum over entries of a 64-bit hash of key, length and
//       bytes) of each DB: the large parity tests compare DBs of hundreds of megabytes through it, whatever the order of the entries in the
//       data file
//   plass_oracle extractorfs <seqDB> <outDB> [flags] | translatenucs <nuclDB> <outAaDB> [--add-orf-stop 1] | concatdbs <dbA> <dbB> <outDB> [--preserve-keys 1]
#include "oracle.hpp"
#include "../plass_amd/csrc/synth_core.hpp"   // the read model of include/plasship_synth.h (measurement infrastructure shared with the GPU generator)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace oracle;

static bool multiParam(const std::string &v, const char *which, std::string &out) {
    // "aa:13,nucl:5" or "13"
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

static oracle::OrfParams orfPar;    // extractorfs / translatenucs flags
static bool preserveKeys = false;   // concatdbs --preserve-keys 1
static bool chopCycle = false;      // --chop-cycle (cyclecheck; setCycleCheckDefaults: off unless the workflow passes it)
static int parseFlags(int argc, char **argv, int from, Params &par, std::vector<std::string> &pos) {
    for (int i = from; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); return 1; }
            std::string v = argv[++i], t;
            if (a == "-k") par.kmerSize = atoi(v.c_str());
            else if (a == "--alph-size") { if (multiParam(v, "aa", t)) par.alphabetSizeAA = atoi(t.c_str()); }
            else if (a == "--kmer-per-seq") par.kmersPerSequence = atoi(v.c_str());
            else if (a == "--kmer-per-seq-scale") {
                if (multiParam(v, "aa", t)) par.kmersPerSequenceScaleAA = strtof(t.c_str(), nullptr);
                if (multiParam(v, "nucl", t)) par.kmersPerSequenceScaleNucl = strtof(t.c_str(), nullptr);
            }
            else if (a == "--hash-shift") par.hashShift = atoi(v.c_str());
            else if (a == "--include-only-extendable") par.includeOnlyExtendable = atoi(v.c_str()) != 0;
            else if (a == "--ignore-multi-kmer") par.ignoreMultiKmer = atoi(v.c_str()) != 0;
            else if (a == "--cov-mode") par.covMode = atoi(v.c_str());
            else if (a == "-c") par.covThr = strtof(v.c_str(), nullptr);
            else if (a == "--rescore-mode") par.rescoreMode = atoi(v.c_str());
            else if (a == "-e") par.evalThr = strtod(v.c_str(), nullptr);
            else if (a == "--min-seq-id") par.seqIdThr = strtof(v.c_str(), nullptr);
            else if (a == "--min-aln-len") par.alnLenThr = atoi(v.c_str());
            else if (a == "--seq-id-mode") par.seqIdMode = atoi(v.c_str());
            else if (a == "-a") par.addBacktrace = atoi(v.c_str()) != 0;
            else if (a == "--add-self-matches") par.includeIdentity = atoi(v.c_str()) != 0;
            else if (a == "--max-seq-len") par.maxSeqLen = (size_t) strtoull(v.c_str(), nullptr, 10);
            else if (a == "--keep-target") par.keepTarget = atoi(v.c_str()) != 0;
            else if (a == "--chop-cycle") chopCycle = atoi(v.c_str()) != 0;
            else if (a == "--preserve-keys") preserveKeys = atoi(v.c_str()) != 0;
            else if (a == "--min-length") orfPar.orfMinLength = (size_t) strtoull(v.c_str(), nullptr, 10);
            else if (a == "--max-length") orfPar.orfMaxLength = (size_t) strtoull(v.c_str(), nullptr, 10);
            else if (a == "--max-gaps") orfPar.orfMaxGaps = (size_t) strtoull(v.c_str(), nullptr, 10);
            else if (a == "--contig-start-mode") orfPar.contigStartMode = atoi(v.c_str());
            else if (a == "--contig-end-mode") orfPar.contigEndMode = atoi(v.c_str());
            else if (a == "--orf-start-mode") orfPar.orfStartMode = atoi(v.c_str());
            else if (a == "--forward-frames" || a == "--reverse-frames") {
                unsigned m = 0; for (char c : v) if (c >= '1' && c <= '3') m |= 1u << (c - '1');
                (a == "--forward-frames" ? orfPar.forwardFrames : orfPar.reverseFrames) = m;
            }
            else if (a == "--translation-table") orfPar.translationTable = atoi(v.c_str());
            else if (a == "--translate") orfPar.translate = atoi(v.c_str()) != 0;
            else if (a == "--use-all-table-starts") orfPar.useAllTableStarts = atoi(v.c_str()) != 0;
            else if (a == "--add-orf-stop") orfPar.addOrfStop = atoi(v.c_str()) != 0;
            else if (a == "--threads") par.threads = std::max(1, atoi(v.c_str()));
            else if (a == "--oracle-no-stale-scan") par.debugNoStaleScan = atoi(v.c_str()) != 0;
            else if (a == "--oracle-old-strand-ties") par.debugOldStrandTies = atoi(v.c_str()) != 0;
            else if (a == "--gap-open") { if (multiParam(v, "nucl", t)) par.gapOpenNucl = atoi(t.c_str()); }
            else if (a == "--gap-extend") { if (multiParam(v, "nucl", t)) par.gapExtendNucl = atoi(t.c_str()); }
            else { /* accepted and ignored: --sub-mat --threads -v --compressed --mask … */ }
        } else pos.push_back(a);
    }
    return 0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// synthetic read pairs on the CPU: the same functions the GPU generator runs (plass_amd/csrc/synth_core.hpp), one loop iteration per base / read
static int synthreads(int argc, char **argv) {
    uint64_t pairs = 0, seed = 1, gmin = 7500000, gmax = 7500000; uint32_t genomes = 1, insertMin = 160, readLen = 150;
    float sigma = 0.0f, insertMean = 320.0f, insertSd = 40.0f, errorRate = 0.002f; std::string out;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) == 0) {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); return 1; }
            const char *v = argv[++i];
            if (a == "--pairs") pairs = strtoull(v, nullptr, 10); else if (a == "--seed") seed = strtoull(v, nullptr, 10);
            else if (a == "--genomes") genomes = (uint32_t) strtoul(v, nullptr, 10); else if (a == "--genome-min-len") gmin = strtoull(v, nullptr, 10);
            else if (a == "--genome-max-len") gmax = strtoull(v, nullptr, 10); else if (a == "--abundance-sigma") sigma = strtof(v, nullptr);
            else if (a == "--insert-mean") insertMean = strtof(v, nullptr); else if (a == "--insert-sd") insertSd = strtof(v, nullptr);
            else if (a == "--insert-min") insertMin = (uint32_t) strtoul(v, nullptr, 10); else if (a == "--read-len") readLen = (uint32_t) strtoul(v, nullptr, 10);
            else if (a == "--error-rate") errorRate = strtof(v, nullptr);
            else { fprintf(stderr, "synthreads: unknown flag %s\n", a.c_str()); return 1; }
        } else out = a;
    }
    if (out.empty() || pairs == 0 || genomes == 0 || gmax < gmin || gmin < 4ull * (uint64_t) (insertMean + 8 * insertSd + readLen)) { fprintf(stderr, "synthreads <outReadDB> --pairs N [...]\n"); return 1; }
    plasship::SynthCommunity c; c.build(seed, genomes, gmin, gmax, sigma);
    std::vector<char> genome(c.total + 64);
    plasship::SynthGenome sg; sg.geneStart = c.geneStart.data(); sg.geneCodons = c.geneCodons.data(); sg.nGenes = c.geneCodons.size(); sg.totalBases = c.total; sg.seed = seed;
#pragma omp parallel for schedule(static)
    for (int64_t x = 0; x < (int64_t) c.total; x++) genome[x] = plasship::synthGenomeBase(sg, (uint64_t) x);
    const uint64_t nReads = 2 * pairs; const uint32_t entry = readLen + 2;
    DB db; db.dbtype = DBTYPE_NUCLEOTIDES; db.key.resize(nReads); db.off.resize(nReads + 1); db.elen.resize(nReads); db.data.resize(nReads * entry);
    std::vector<uint32_t> len(nReads);
    plasship::SynthReads sr; memset(&sr, 0, sizeof(sr));
    sr.genome = genome.data(); sr.genomeStart = c.gStart.data(); sr.cum = c.cum.data(); sr.nGenomes = genomes; sr.readLen = readLen; sr.insertMin = insertMin;
    sr.insertMean = insertMean; sr.insertSd = insertSd; sr.errThresh = (uint32_t) ((double) errorRate * 1073741824.0); sr.nPairs = pairs; sr.seed = seed;
    sr.out = &db.data[0]; sr.off = db.off.data(); sr.len = len.data(); sr.key = db.key.data();
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t) nReads; r++) plasship::synthRead(sr, (uint64_t) r);
    db.off.resize(nReads);
    for (uint64_t r = 0; r < nReads; r++) db.elen[r] = entry;
    std::string err;
    if (!writeDB(out, db, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    fprintf(stderr, "oracle synthreads: %llu reads, %llu genome bases, %zu genes\n", (unsigned long long) nReads, (unsigned long long) c.total, c.geneCodons.size());
    return 0;
}

static int dbsum(int argc, char **argv) {
    for (int i = 2; i < argc; i++) {
        DB db; std::string err;
        if (!readDB(argv[i], db, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        uint64_t sum = 0, bytes = 0;
        const int64_t n = (int64_t) db.size();
#pragma omp parallel for schedule(static) reduction(+ : sum, bytes)
        for (int64_t e = 0; e < n; e++) {
            const unsigned char *p = (const unsigned char *) db.entry((size_t) e); const uint32_t len = db.elen[e];
            uint64_t h = 0xCBF29CE484222325ull ^ ((uint64_t) db.key[e] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t) len << 40);
            for (uint32_t j = 0; j < len; j++) { h ^= p[j]; h *= 0x100000001B3ull; }                 // FNV-1a over the entry as indexed (with its '\0')
            h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
            sum += h; bytes += len;
        }
        printf("%s\tentries=%zu\tbytes=%llu\tdbtype=%d\tdigest=%016llx\n", argv[i], db.size(), (unsigned long long) bytes, db.dbtype, (unsigned long long) sum);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: plass_oracle <module> <dbs…> [flags]\n"); return 1; }
    if (std::string(argv[1]) == "dbsum") return dbsum(argc, argv);
    if (std::string(argv[1]) == "synthreads") return synthreads(argc, argv);
    std::string mod = argv[1];
    Params par; std::vector<std::string> pos; std::string err;
    // module defaults: kmermatcher's setLinearFilterDefault sets covThr 0.8 (kmermatcher.cpp:566-573);
    // the workflows always pass -c explicitly, and so do the tests.
    if (parseFlags(argc, argv, 2, par, pos)) return 1;
    if (mod == "kmermatcher") {
        if (pos.size() != 2) { fprintf(stderr, "kmermatcher <seqDB> <prefDB>\n"); return 1; }
        DB seq; if (!readDB(pos[0], seq, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now(); KmerStats st;
        DB pref = kmermatcher(seq, par, &st);
        double t1 = now();
        if (st.nStrandTieTriples) fprintf(stderr, "oracle kmermatcher: %zu (rep, target, diagonal) triples in %zu pairs hold both strands (sort-#2 ties)\n", st.nStrandTieTriples, st.nStrandTiePairs);
        fprintf(stderr, "oracle kmermatcher: %zu seqs, N_k=%zu N_m=%zu N_c=%zu, %.3f s\n", seq.size(), st.nKmerRecords, st.nGrouped, st.nCandidates, t1 - t0);
        if (!writeDB(pos[1], pref, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "rescorediagonal") {
        if (pos.size() != 4) { fprintf(stderr, "rescorediagonal <qDB> <tDB> <prefDB> <alnDB>\n"); return 1; }
        DB q, t, pref;
        if (!readDB(pos[0], q, err) || !readDB(pos[2], pref, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        bool same = pos[0] == pos[1];
        if (!same && !readDB(pos[1], t, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now();
        DB aln = rescorediagonal(q, same ? q : t, same, pref, par);
        fprintf(stderr, "oracle rescorediagonal: %.3f s\n", now() - t0);
        if (!writeDB(pos[3], aln, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "assembleresults" || mod == "nuclassembleresults") {
        if (pos.size() != 3) { fprintf(stderr, "%s <seqDB> <alnDB> <outDB>\n", mod.c_str()); return 1; }
        DB seq, aln;
        if (!readDB(pos[0], seq, err) || !readDB(pos[1], aln, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now();
        DB out = (mod == "assembleresults") ? assembleresults(seq, aln, par) : nuclassembleresults(seq, aln, par);
        fprintf(stderr, "oracle %s: %.3f s\n", mod.c_str(), now() - t0);
        if (!writeDB(pos[2], out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "guidedassembleresults") {
        if (pos.size() != 5) { fprintf(stderr, "guidedassembleresults <nuclDB> <aaDB> <nuclAlnDB> <outNuclDB> <outAaDB>\n"); return 1; }
        DB nucl, aa, aln, on, oa;
        if (!readDB(pos[0], nucl, err) || !readDB(pos[1], aa, err) || !readDB(pos[2], aln, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now();
        if (!guidedassembleresults(nucl, aa, aln, par, on, oa, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        fprintf(stderr, "oracle guidedassembleresults: %.3f s\n", now() - t0);
        if (!writeDB(pos[3], on, err) || !writeDB(pos[4], oa, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "proteinaln2nucl") {
        if (pos.size() != 6) { fprintf(stderr, "proteinaln2nucl <qNuclDB> <tNuclDB> <qAaDB> <tAaDB> <alnDB> <outAlnDB>\n"); return 1; }
        DB qn, tn, qa, ta, aln, out;
        const bool same = pos[0] == pos[1] && pos[2] == pos[3];
        if (!same && (pos[0] == pos[1] || pos[2] == pos[3])) { fprintf(stderr, "Either query database == target database for nucleotide and amino acid or != for both\n"); return 1; }
        if (!readDB(pos[0], qn, err) || !readDB(pos[2], qa, err) || !readDB(pos[4], aln, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (!same && (!readDB(pos[1], tn, err) || !readDB(pos[3], ta, err))) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now();
        if (!proteinaln2nucl(qn, same ? qn : tn, qa, same ? qa : ta, aln, par, out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        fprintf(stderr, "oracle proteinaln2nucl: %.3f s\n", now() - t0);
        if (!writeDB(pos[5], out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "findassemblystart") {
        if (pos.size() != 3) { fprintf(stderr, "findassemblystart <seqDB> <alnDB> <outSeqDB>\n"); return 1; }
        DB seq, aln, out;
        if (!readDB(pos[0], seq, err) || !readDB(pos[1], aln, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now();
        if (!findassemblystart(seq, aln, out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        fprintf(stderr, "oracle findassemblystart: %.3f s\n", now() - t0);
        if (!writeDB(pos[2], out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "cyclecheck") {
        if (pos.size() != 2) { fprintf(stderr, "cyclecheck <seqDB> <outCycleDB> [--max-seq-len N --chop-cycle 0|1]\n"); return 1; }
        DB seq, out;
        if (!readDB(pos[0], seq, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        double t0 = now();
        if (!cyclecheck(seq, par, chopCycle, out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        fprintf(stderr, "oracle cyclecheck: %.3f s\n", now() - t0);
        if (!writeDB(pos[1], out, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "extractorfs") {
        if (pos.size() != 2) { fprintf(stderr, "extractorfs <seqDB> <outDB>   (writes <outDB> and <outDB>_h)\n"); return 1; }
        DB seq, o, oh;
        if (!readDB(pos[0], seq, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        orfPar.maxSeqLen = par.maxSeqLen;
        double t0 = now();
        if (!extractorfs(seq, orfPar, o, oh, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        fprintf(stderr, "oracle extractorfs: %zu orfs, %.3f s\n", o.size(), now() - t0);
        if (!writeDB(pos[1], o, err) || !writeDB(pos[1] + "_h", oh, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "translatenucs") {
        if (pos.size() != 2) { fprintf(stderr, "translatenucs <nuclDB> <outAaDB>   (--add-orf-stop 1 reads <nuclDB>_h)\n"); return 1; }
        DB seq, hdr, o;
        if (!readDB(pos[0], seq, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (orfPar.addOrfStop && !readDB(pos[0] + "_h", hdr, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        orfPar.maxSeqLen = par.maxSeqLen;
        double t0 = now();
        if (!translatenucs(seq, orfPar.addOrfStop ? &hdr : nullptr, orfPar, o, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        fprintf(stderr, "oracle translatenucs: %.3f s\n", now() - t0);
        if (!writeDB(pos[1], o, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (mod == "concatdbs") {
        if (pos.size() != 3) { fprintf(stderr, "concatdbs <dbA> <dbB> <outDB>\n"); return 1; }
        DB a, b, o;
        if (!readDB(pos[0], a, err) || !readDB(pos[1], b, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (!concatdbs(a, b, o, err, preserveKeys)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (!writeDB(pos[2], o, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else { fprintf(stderr, "unknown module %s\n", mod.c_str()); return 1; }
    return 0;
}
