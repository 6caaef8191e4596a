// Host-side helpers of the product: DB files, text formats, E-value arithmetic.
// Format contracts: SURVEY.md §8b; text formats QueryMatcher.h:114-126, Matcher.cpp:323-370.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace plasship {

struct HostDB {
    int dbtype = 0;
    std::string data;                       // all data files concatenated
    std::vector<uint32_t> key, elen;        // index lines in file order
    std::vector<uint64_t> off;
};
bool readDBFiles(const std::string &path, HostDB &db, std::string &err);

// streaming writer: entries must be appended in key order; writes NAME, NAME.index, NAME.dbtype
struct DBFileWriter {
    FILE *fd = nullptr, *fi = nullptr; std::string path; uint64_t off = 0; int dbtype = 0;
    std::string ibuf;
    bool open(const std::string &p, int type, std::string &err);
    void add(uint32_t key, const char *bytes, size_t n);   // appends '\0'
    bool close(std::string &err);
};

// decimal formatting without the libc (hot in DB writing)
char *fmtU32(uint32_t v, char *p);          // returns pointer past last digit (no terminator)
char *fmtI32(int32_t v, char *p);
char *fmtU64(uint64_t v, char *p);

// E-value machinery (EvalueComputation.h + ALP sls_pvalues.cpp:366-545), gapless parameter sets only
struct HostEvaluer {
    const double *g; double logK, ln2, dbRes;
    HostEvaluer(bool nucl, uint64_t dbResidues);
    // gapped evaluer on the nucleotide matrix as proteinaln2nucl builds it (mm/util/proteinaln2nucl.cpp:54-58); parameters exist
    // for the penguin workflow's --gap-open 5 --gap-extend 2 only (captured from the reference's ALP run)
    static bool nuclGapped(int gapOpen, int gapExtend, uint64_t dbResidues, HostEvaluer &out);
    double evalue(double score, double qLen) const;
    double bitScore(double score) const;
    double rawFromBit(double bits) const;
    // smallest integer score s in [0, maxScore] with evalue(s, qLen) <= thr, or maxScore+1 if none
    int minScoreForEvalue(double thr, int qLen, int maxScore, int guess = -1) const;
};

// CompareNuclResultByScore's posterior (nuclassembleresult.cpp:36-70) with the platform libm, as the reference computes
// it: the decision p < 0.45 / p > 0.55 is defined by glibc's lgamma/log/exp whenever p falls on a threshold
// (e.g. zero mismatches on both sides and overlap lengths 9 : 11 give p = 0.45 up to rounding).
// returns 0 (p < 0.45), 1 (p > 0.55) or 2 (in between)
int nuclPosteriorClass(uint32_t alpha1, uint32_t beta1, uint32_t alpha2, uint32_t beta2);

const signed char *asciiSubMat(bool nucl);                 // 123 x 123
const unsigned char *aa2numTable(bool nucl, int alphabetSize);
}  // namespace plasship
