// Host-side helpers of the product: DB files, text formats, E-value arithmetic.
// Format contracts: SURVEY.md §8b; text formats QueryMatcher.h:114-126, Matcher.cpp:323-370.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace plasship {

// host threads for file reading, index / text parsing and formatting (PLASSHIP_HOST_THREADS; default: the cores this process may
// run on, at most 16 — the reference formats and parses on its OpenMP threads)
int hostThreads();
// f(t, begin, end) on up to hostThreads() threads over contiguous ranges of [0, n); with `prefix` (n + 1 running weights, e.g. a CSR
// offset array) the ranges carry about equal weight.  Returns the number of ranges used (range t was given to f exactly once).
int parallelRanges(size_t n, const std::function<void(int, size_t, size_t)> &f, const uint64_t *prefix = nullptr, size_t minPerThread = 4096);

// uninitialised byte buffer (a std::string would zero-fill gigabytes on one thread before the file is read into it)
struct HostBytes {
    char *p = nullptr; size_t n = 0;
    HostBytes() {}
    HostBytes(const HostBytes &) = delete; HostBytes &operator=(const HostBytes &) = delete;
    HostBytes(HostBytes &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    HostBytes &operator=(HostBytes &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~HostBytes() { free(p); }
    bool alloc(size_t bytes) { free(p); n = bytes; p = (char *) malloc(bytes + 64); if (p) memset(p + bytes, 0, 64); return p != nullptr; }
    const char *data() const { return p; }
    char *data() { return p; }
    size_t size() const { return n; }
};

struct HostDB {
    int dbtype = 0;
    HostBytes data;                         // all data files concatenated (64 zero bytes follow)
    std::vector<uint32_t> key, elen;        // index lines in file order
    std::vector<uint64_t> off;
};
// NAME or NAME.0 .. NAME.k (the unmerged per-thread files of the reference's DBWriter; offsets run over their concatenation),
// NAME.index, NAME.dbtype.  The data is read with pread() on all host threads, the index is parsed on all host threads.
bool readDBFiles(const std::string &path, HostDB &db, std::string &err);
bool ioTimingOn();      // PLASSHIP_IO_TIMING=1: phase timings of the host boundary on stderr
double ioNow();

// Writer of NAME, NAME.index, NAME.dbtype.  Everything goes to "<file>.tmp.<pid>" first; close() checks every write and renames
// the three files into place, so a reader never sees a half-written DB and a failed write never leaves one behind under NAME.
//   large DBs:  data(bytes…) in file order, then index(keys, entry lengths) — offsets are the running sum, lines are formatted on
//               all host threads
//   small DBs:  add(key, bytes) per entry
struct DBFileWriter {
    FILE *fd = nullptr, *fi = nullptr; std::string path, tmpSuffix; uint64_t off = 0, dataPos = 0; int dbtype = 0; std::atomic<bool> failed{false}; bool open_ = false;   // failed: data() and index() may run on two threads (plasship_seqdb_write)
    std::string ibuf;
    DBFileWriter() {}
    DBFileWriter(const DBFileWriter &) = delete; DBFileWriter &operator=(const DBFileWriter &) = delete;
    ~DBFileWriter();
    bool open(const std::string &p, int type, std::string &err);
    void data(const char *bytes, size_t n);
    void index(const uint32_t *keys, const uint32_t *elen, size_t n);   // elen counts the entry's '\0'
    void add(uint32_t key, const char *bytes, size_t n);   // appends '\0'
    bool close(std::string &err);
};
// text DB (prefilter / alignment / header DBs): fmt(q, out) appends the lines of entry q; entries are formatted on all host threads
// (ranges balanced by `prefix`, n + 1 running line counts, if given) and written in key order.  fmt returns false to abort.
bool writeTextDB(const std::string &path, int dbtype, const uint32_t *keys, size_t n, const uint64_t *prefix,
                 const std::function<bool(size_t, std::string &)> &fmt, std::string &err);

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
