// ORACLE (test infrastructure): tiny C surface for ctypes (tests/ and bench.py's cpu_baseline only).
#include "oracle.hpp"
#include "ref_tables.h"
#include <cstring>
using namespace oracle;
extern "C" {
unsigned long long oracle_xxh64_u64(unsigned long long v, unsigned long long seed) { return xxh64_u64(v, seed); }
unsigned long long oracle_revcomp(unsigned long long kmer, int k) { return revComplement(kmer, k); }
double oracle_bitscore(int nucl, double s) { return Evaluer(nucl != 0, 1000000).bitScore(s); }
double oracle_raw_from_bit(int nucl, double b) { return Evaluer(nucl != 0, 1000000).rawFromBit(b); }
double oracle_evalue(int nucl, unsigned long long dbRes, double s, double qLen) { return Evaluer(nucl != 0, dbRes).evalue(s, qLen); }
const double *oracle_kat_aa() { return REF_KAT_AA; }
const double *oracle_kat_nuc() { return REF_KAT_NUC; }
const unsigned long long *oracle_kat_xxh64() { return &REF_KAT_XXH64[0][0]; }
const unsigned long long *oracle_kat_revcomp() { return &REF_KAT_REVCOMP[0][0]; }
const signed char *oracle_ascii_submat(int nucl) { return asciiSubMat(nucl != 0); }
const unsigned char *oracle_aa2num(int nucl, int alph) { return aa2num(nucl != 0, alph); }
// run a module on DB files; returns 0 on success. Stats: N_k, N_m, N_c.
int oracle_kmermatcher_files(const char *seqDb, const char *outDb, int k, int alph, int kps, float scaleAA, float scaleNucl,
                             int hashShift, int onlyExt, int ignoreMulti, int covMode, float covThr, unsigned long long *stats) {
    DB s; std::string err; if (!readDB(seqDb, s, err)) return 1;
    Params p; p.kmerSize = k; p.alphabetSizeAA = alph; p.kmersPerSequence = kps; p.kmersPerSequenceScaleAA = scaleAA;
    p.kmersPerSequenceScaleNucl = scaleNucl; p.hashShift = hashShift; p.includeOnlyExtendable = onlyExt; p.ignoreMultiKmer = ignoreMulti;
    p.covMode = covMode; p.covThr = covThr;
    KmerStats st; DB o = kmermatcher(s, p, &st);
    if (stats) { stats[0] = st.nKmerRecords; stats[1] = st.nGrouped; stats[2] = st.nCandidates; }
    return writeDB(outDb, o, err) ? 0 : 1;
}
int oracle_rescorediagonal_files(const char *qDb, const char *tDb, const char *prefDb, const char *outDb, int mode, double evalThr,
                                 float seqIdThr, int covMode, float covThr, int alnLenThr, int seqIdMode, int addBt) {
    DB q, t, pr; std::string err; bool same = strcmp(qDb, tDb) == 0;
    if (!readDB(qDb, q, err) || !readDB(prefDb, pr, err)) return 1;
    if (!same && !readDB(tDb, t, err)) return 1;
    Params p; p.rescoreMode = mode; p.evalThr = evalThr; p.seqIdThr = seqIdThr; p.covMode = covMode; p.covThr = covThr;
    p.alnLenThr = alnLenThr; p.seqIdMode = seqIdMode; p.addBacktrace = addBt;
    DB o = rescorediagonal(q, same ? q : t, same, pr, p);
    return writeDB(outDb, o, err) ? 0 : 1;
}
int oracle_assembleresults_files(const char *seqDb, const char *alnDb, const char *outDb, int nuclVariant, float seqIdThr,
                                 unsigned long long maxSeqLen, int keepTarget, int rescoreMode) {
    DB s, a; std::string err; if (!readDB(seqDb, s, err) || !readDB(alnDb, a, err)) return 1;
    Params p; p.seqIdThr = seqIdThr; p.maxSeqLen = maxSeqLen; p.keepTarget = keepTarget; p.rescoreMode = rescoreMode;
    DB o = nuclVariant ? nuclassembleresults(s, a, p) : assembleresults(s, a, p);
    return writeDB(outDb, o, err) ? 0 : 1;
}
}
