// Host-side helpers of the product (see host_util.hpp).  Product code: independent of oracle/.
#include "host_util.hpp"
#include "tables_data.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>

namespace plasship {

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

static bool appendFile(const std::string &p, std::string &out) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    struct stat st; if (fstat(fileno(f), &st) != 0) { fclose(f); return false; }
    size_t n = (size_t) st.st_size, old = out.size();
    out.resize(old + n);
    size_t got = n ? fread(&out[old], 1, n, f) : 0;
    fclose(f);
    return got == n;
}

// DB layout: NAME or NAME.0..NAME.k data (offsets global over the concatenation, FileUtil.cpp:336-352),
// NAME.index "key\toffset\tlength\n", NAME.dbtype int32 LE (bit 31 = compressed, unsupported here).
bool readDBFiles(const std::string &path, HostDB &db, std::string &err) {
    db = HostDB();
    std::string t;
    if (!appendFile(path + ".dbtype", t) || t.size() < 4) { err = "cannot read " + path + ".dbtype"; return false; }
    uint32_t ty; memcpy(&ty, t.data(), 4);
    if (ty & 0x80000000u) { err = "compressed database not supported: " + path; return false; }
    db.dbtype = (int) (ty & 0x3FFFFFFFu);
    if (exists(path)) {
        if (!appendFile(path, db.data)) { err = "cannot read " + path; return false; }
    } else {
        int i = 0;
        for (;; i++) {
            std::string p = path + "." + std::to_string(i);
            if (!exists(p)) break;
            if (!appendFile(p, db.data)) { err = "cannot read " + p; return false; }
        }
        if (i == 0) { err = "no data file for " + path; return false; }
    }
    std::string idx;
    if (!appendFile(path + ".index", idx)) { err = "cannot read " + path + ".index"; return false; }
    size_t lines = (size_t) std::count(idx.begin(), idx.end(), '\n');
    db.key.reserve(lines); db.off.reserve(lines); db.elen.reserve(lines);
    const char *p = idx.data(), *e = p + idx.size();
    while (p < e) {
        uint64_t v[3] = {0, 0, 0};
        for (int c = 0; c < 3; c++) {
            while (p < e && (*p == '\t' || *p == ' ')) p++;
            while (p < e && *p >= '0' && *p <= '9') v[c] = v[c] * 10 + (uint64_t) (*p++ - '0');
        }
        while (p < e && *p != '\n') p++;
        if (p < e) p++;
        if (v[1] + v[2] > db.data.size()) { err = "index entry points past the data of " + path; return false; }
        db.key.push_back((uint32_t) v[0]); db.off.push_back(v[1]); db.elen.push_back((uint32_t) v[2]);
    }
    return true;
}

bool DBFileWriter::open(const std::string &p, int type, std::string &err) {
    path = p; dbtype = type; off = 0;
    fd = fopen(p.c_str(), "wb"); fi = fopen((p + ".index").c_str(), "wb");
    if (!fd || !fi) { err = "cannot open " + p + " for writing"; return false; }
    setvbuf(fd, nullptr, _IOFBF, 1 << 22);
    ibuf.clear(); ibuf.reserve(1 << 22);
    return true;
}
void DBFileWriter::add(uint32_t key, const char *bytes, size_t n) {
    fwrite(bytes, 1, n, fd); fputc('\0', fd);
    char tmp[64]; char *q = fmtU32(key, tmp); *q++ = '\t'; q = fmtU64(off, q); *q++ = '\t'; q = fmtU64(n + 1, q); *q++ = '\n';
    ibuf.append(tmp, (size_t) (q - tmp));
    if (ibuf.size() > (1 << 22) - 128) { fwrite(ibuf.data(), 1, ibuf.size(), fi); ibuf.clear(); }
    off += n + 1;
}
bool DBFileWriter::close(std::string &err) {
    bool ok = true;
    if (fi) { fwrite(ibuf.data(), 1, ibuf.size(), fi); ok &= (fclose(fi) == 0); fi = nullptr; }
    if (fd) { ok &= (fclose(fd) == 0); fd = nullptr; }
    FILE *ft = fopen((path + ".dbtype").c_str(), "wb");
    if (!ft) ok = false; else { uint32_t ty = (uint32_t) dbtype; fwrite(&ty, 4, 1, ft); fclose(ft); }
    if (!ok) err = "error while writing " + path;
    return ok;
}

char *fmtU64(uint64_t v, char *p) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char) ('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
char *fmtU32(uint32_t v, char *p) { return fmtU64(v, p); }
char *fmtI32(int32_t v, char *p) {
    if (v < 0) { *p++ = '-'; return fmtU64((uint64_t) (-(int64_t) v), p); }
    return fmtU64((uint64_t) v, p);
}

const signed char *asciiSubMat(bool nucl) { return nucl ? PH_NUC_ASCII_SUBMAT : PH_AA_ASCII_SUBMAT; }
const unsigned char *aa2numTable(bool nucl, int alphabetSize) {
    if (nucl) return PH_NUC_AA2NUM;
    return alphabetSize == 21 ? PH_AA21_AA2NUM : PH_AA13_AA2NUM;
}

// ---- E-values -------------------------------------------------------------------------------------
// The reference binary is built with FMA contraction; the two affine forms below are single
// fused multiply-adds there (checked against known answers captured from it).
HostEvaluer::HostEvaluer(bool nucl, uint64_t dbResidues)
    : g(nucl ? PH_NUC_GAPLESS_GUMBEL : PH_AA_GAPLESS_GUMBEL), logK(std::log(g[1])), ln2(std::log(2.0)), dbRes((double) dbResidues) {}
bool HostEvaluer::nuclGapped(int gapOpen, int gapExtend, uint64_t dbResidues, HostEvaluer &out) {
    if (gapOpen != 5 || gapExtend != 2) return false;
    out = HostEvaluer(true, dbResidues);
    out.g = PH_NUC_GAPPED_5_2_GUMBEL; out.logK = std::log(out.g[1]);
    return true;
}
double HostEvaluer::bitScore(double s) const { return std::fma(g[0], s, -logK) / ln2; }
double HostEvaluer::rawFromBit(double b) const { return std::fma(b, ln2, logK) / g[0]; }

static inline double normalCdf(double x) { return 0.5 * std::erfc(-std::sqrt(0.5) * x); }

double HostEvaluer::evalue(double y, double qLen) const {
    const double pi = 3.1415926535897932384626433832795;
    const double cv = 1 / std::sqrt(2.0 * pi);
    // finite-size corrected area (query side "j", database side "i")
    double mli = dbRes - (g[2] * y + g[3]);
    double svi = std::sqrt(std::fmax(g[12], g[4] * y + g[5]));
    double mF = (svi == 0.0) ? 1e100 : mli / svi;
    double PmF = normalCdf(mF), EmF = -cv * std::exp(-0.5 * mF * mF);
    double p1 = mli * PmF - svi * EmF;
    double nlj = qLen - (g[6] * y + g[7]);
    double svj = std::sqrt(std::fmax(g[13], g[8] * y + g[9]));
    double nF = (svj == 0.0) ? 1e100 : nlj / svj;
    double PnF = normalCdf(nF), EnF = -cv * std::exp(-0.5 * nF * nF);
    double p2 = nlj * PnF - svj * EnF;
    double cy = std::fmax(g[14], g[10] * y + g[11]);
    double area = p1 * p2 + cy * PmF * PnF;
    return g[1] * std::exp(-g[0] * y) * area;
}

int HostEvaluer::minScoreForEvalue(double thr, int qLen, int maxScore, int guess) const {
    // E(s) is strictly decreasing in s for fixed lengths (exponential factor times a decreasing,
    // positive area); search on the exact double predicate the reference evaluates.  `guess` (the answer of a
    // neighbouring length, or < 0) only picks where the bracketing starts: a gallop around it, then bisection.
    auto pred = [&](int s) { return evalue(s, qLen) <= thr; };
    int lo, hi;                            // invariant: pred(lo) false, pred(hi) true
    if (guess >= 1 && guess <= maxScore) {
        if (pred(guess)) {
            hi = guess; int step = 1; lo = -1;
            while (hi - step >= 0) { if (pred(hi - step)) { hi -= step; step *= 2; } else { lo = hi - step; break; } }
            if (lo < 0) { if (pred(0)) return 0; lo = 0; }
        } else {
            lo = guess; int step = 1; hi = -1;
            while (lo + step <= maxScore) { if (!pred(lo + step)) { lo += step; step *= 2; } else { hi = lo + step; break; } }
            if (hi < 0) { if (!pred(maxScore)) return maxScore + 1; hi = maxScore; }
        }
    } else {
        if (pred(0)) return 0;
        if (!pred(maxScore)) return maxScore + 1;
        lo = 0; hi = maxScore;
    }
    while (hi - lo > 1) { int mid = lo + (hi - lo) / 2; if (pred(mid)) hi = mid; else lo = mid; }
    return hi;
}

int nuclPosteriorClass(uint32_t alpha1, uint32_t beta1, uint32_t alpha2, uint32_t beta2) {
    const unsigned a1 = alpha1, b1 = beta1, a2 = alpha2, b2 = beta2;
    const double log_c = (std::lgamma(b1 + b2) + std::lgamma(a1 + b1)) - (std::lgamma(a1 + b1 + b2) + std::lgamma(b1));
    double log_r = 0.0, p = 0.0;
    for (size_t idx = 0; idx < a2; idx++) {
        p += exp(log_r + log_c);
        log_r = log(a1 + idx) + log(b2 + idx) - (log(idx + 1) + log(idx + a1 + b1 + b2)) + log_r;
    }
    if (p < 0.45) return 0;
    if (p > 0.55) return 1;
    return 2;
}

}  // namespace plasship
