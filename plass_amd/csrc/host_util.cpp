// Host-side helpers of the product (see host_util.hpp).  Product code: independent of oracle/.
#include "host_util.hpp"
#include <chrono>
#include "tables_data.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <sched.h>
#include <atomic>
#include <thread>

namespace plasship {

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

int hostThreads() {
    static const int n = [] {
        if (const char *e = getenv("PLASSHIP_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 256); }
        cpu_set_t cs; CPU_ZERO(&cs);
        int c = (sched_getaffinity(0, sizeof(cs), &cs) == 0) ? CPU_COUNT(&cs) : (int) std::thread::hardware_concurrency();
        return std::max(1, std::min(c, 32));
    }();
    return n;
}

int parallelRanges(size_t n, const std::function<void(int, size_t, size_t)> &f, const uint64_t *prefix, size_t minPerThread) {
    int T = hostThreads();
    if (minPerThread && n / minPerThread < (size_t) T) T = (int) std::max<size_t>(1, n / minPerThread);
    if (T <= 1) { f(0, 0, n); return 1; }
    std::vector<size_t> cut((size_t) T + 1, 0);
    cut[(size_t) T] = n;
    for (int t = 1; t < T; t++) {
        if (prefix) {
            const uint64_t want = prefix[0] + (uint64_t) (((unsigned __int128) (prefix[n] - prefix[0]) * (unsigned) t) / (unsigned) T);
            cut[(size_t) t] = (size_t) (std::lower_bound(prefix, prefix + n, want) - prefix);
        } else cut[(size_t) t] = (size_t) (((unsigned __int128) n * (unsigned) t) / (unsigned) T);
        cut[(size_t) t] = std::max(cut[(size_t) t], cut[(size_t) t - 1]);
    }
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back([&, t] { f(t, cut[(size_t) t], cut[(size_t) t + 1]); });
    f(0, cut[0], cut[1]);
    for (auto &x : th) x.join();
    return T;
}

static bool readSmallFile(const std::string &p, std::string &out) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    char buf[256]; size_t got; out.clear();
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, got);
    fclose(f);
    return true;
}
static bool fileSize(const std::string &p, uint64_t &n) { struct stat st; if (stat(p.c_str(), &st) != 0) return false; n = (uint64_t) st.st_size; return true; }
// [dst, dst + n) = bytes [0, n) of the file, read in slices on all host threads
static bool readInto(const std::string &p, char *dst, uint64_t n) {
    const int fd = ::open(p.c_str(), O_RDONLY);
    if (fd < 0) return false;
    std::atomic<bool> ok(true);
    const size_t SL = 8u << 20;                                    // slice
    const size_t nSl = (size_t) ((n + SL - 1) / SL);
    parallelRanges(nSl, [&](int, size_t b, size_t e) {
        for (size_t s = b; s < e && ok; s++) {
            uint64_t o = (uint64_t) s * SL; const uint64_t end = std::min<uint64_t>(n, o + SL);
            while (o < end) { const ssize_t g = pread(fd, dst + o, (size_t) (end - o), (off_t) o); if (g <= 0) { ok = false; break; } o += (uint64_t) g; }
        }
    }, nullptr, 4);
    ::close(fd);
    return ok;
}

// DB layout: NAME or NAME.0..NAME.k data (offsets global over the concatenation, FileUtil.cpp:336-352),
// NAME.index "key\toffset\tlength\n", NAME.dbtype int32 LE (bit 31 = compressed, unsupported here).
// PLASSHIP_IO_TIMING=1: the phases of reading / writing a DB on stderr (tools/chain_wall_probe.py)
bool ioTimingOn() { static const bool v = getenv("PLASSHIP_IO_TIMING") != nullptr; return v; }
double ioNow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool readDBFiles(const std::string &path, HostDB &db, std::string &err) {
    db = HostDB();
    const double tio0 = ioNow();
    std::string t;
    if (!readSmallFile(path + ".dbtype", t) || t.size() < 4) { err = "cannot read " + path + ".dbtype"; return false; }
    uint32_t ty; memcpy(&ty, t.data(), 4);
    if (ty & 0x80000000u) { err = "compressed database not supported: " + path; return false; }
    db.dbtype = (int) (ty & 0x3FFFFFFFu);
    std::vector<std::pair<std::string, uint64_t>> files; uint64_t total = 0;
    if (exists(path)) { uint64_t n; if (!fileSize(path, n)) { err = "cannot read " + path; return false; } files.push_back({path, n}); total = n; }
    else {
        for (int i = 0;; i++) {
            const std::string p = path + "." + std::to_string(i);
            uint64_t n; if (!exists(p) || !fileSize(p, n)) break;
            files.push_back({p, n}); total += n;
        }
        if (files.empty()) { err = "no data file for " + path; return false; }
    }
    if (!db.data.alloc((size_t) total)) { err = "out of host memory reading " + path; return false; }
    { uint64_t o = 0; for (auto &f : files) { if (!readInto(f.first, db.data.data() + o, f.second)) { err = "cannot read " + f.first; return false; } o += f.second; } }
    const double tio1 = ioNow();
    uint64_t idxBytes = 0; HostBytes idx;
    if (!fileSize(path + ".index", idxBytes) || !idx.alloc((size_t) idxBytes) || !readInto(path + ".index", idx.data(), idxBytes)) { err = "cannot read " + path + ".index"; return false; }
    const double tio2 = ioNow();
    // parse on all threads: a range of bytes handles the lines that START in it
    const char *base = idx.data(); const size_t nb = idx.size();
    const int maxT = hostThreads();
    struct Part { std::vector<uint32_t> key, elen; std::vector<uint64_t> off; bool bad = false; };
    std::vector<Part> parts((size_t) maxT);
    const uint64_t dataSize = total;
    const int used = parallelRanges(nb, [&](int tix, size_t b, size_t e) {
        Part &pt = parts[(size_t) tix];
        { const size_t guess = (e - b) / 16 + 16; pt.key.reserve(guess); pt.off.reserve(guess); pt.elen.reserve(guess); }      // an index line is >= 6, typically 15-25 bytes
        const char *p = base + b, *end = base + e, *fileEnd = base + nb;
        if (b > 0) { while (p < fileEnd && p[-1] != '\n') p++; }              // first line start at or after b
        while (p < end) {
            uint64_t v[3] = {0, 0, 0};
            for (int c = 0; c < 3; c++) {
                while (p < fileEnd && (*p == '\t' || *p == ' ')) p++;
                while (p < fileEnd && *p >= '0' && *p <= '9') v[c] = v[c] * 10 + (uint64_t) (*p++ - '0');
            }
            while (p < fileEnd && *p != '\n') p++;
            if (p < fileEnd) p++;
            if (v[1] + v[2] > dataSize) { pt.bad = true; return; }
            pt.key.push_back((uint32_t) v[0]); pt.off.push_back(v[1]); pt.elen.push_back((uint32_t) v[2]);
        }
    }, nullptr, 1u << 16);
    size_t lines = 0;
    for (int t = 0; t < used; t++) { if (parts[(size_t) t].bad) { err = "index entry points past the data of " + path; return false; } lines += parts[(size_t) t].key.size(); }
    db.key.resize(lines); db.off.resize(lines); db.elen.resize(lines);
    std::vector<size_t> at((size_t) used + 1, 0);
    for (int t = 0; t < used; t++) at[(size_t) t + 1] = at[(size_t) t] + parts[(size_t) t].key.size();
    parallelRanges((size_t) used, [&](int, size_t tb, size_t te) {
        for (size_t t = tb; t < te; t++) {
            const Part &pt = parts[t];
            if (!pt.key.empty()) { memcpy(&db.key[at[t]], pt.key.data(), pt.key.size() * 4); memcpy(&db.off[at[t]], pt.off.data(), pt.off.size() * 8); memcpy(&db.elen[at[t]], pt.elen.data(), pt.elen.size() * 4); }
        }
    }, nullptr, 1);
    if (ioTimingOn()) fprintf(stderr, "[plasship io] read %s: data %.2f GB in %.3f s (%.2f GB/s), index %.2f GB in %.3f s, parse %zu lines %.3f s (%d host threads)\n", path.c_str(),
                              (double) total / 1e9, tio1 - tio0, (double) total / 1e9 / std::max(tio1 - tio0, 1e-9), (double) idxBytes / 1e9, tio2 - tio1, lines, ioNow() - tio2, maxT);
    return true;
}

// ---- writer ---------------------------------------------------------------------------------------
DBFileWriter::~DBFileWriter() {
    if (!open_) return;                                      // never opened, or closed properly
    if (fd) fclose(fd);
    if (fi) fclose(fi);
    for (const char *sfx : {"", ".index", ".dbtype"}) (void) remove((path + sfx + tmpSuffix).c_str());
}
bool DBFileWriter::open(const std::string &p, int type, std::string &err) {
    path = p; dbtype = type; off = 0; dataPos = 0; failed = false;
    tmpSuffix = ".tmp." + std::to_string((long) getpid());
    fd = fopen((p + tmpSuffix).c_str(), "wb"); fi = fopen((p + ".index" + tmpSuffix).c_str(), "wb");
    open_ = true;
    if (!fd || !fi) { err = "cannot open " + p + " for writing"; return false; }
    setvbuf(fd, nullptr, _IOFBF, 1 << 22);
    ibuf.clear(); ibuf.reserve(1 << 22);
    return true;
}
void DBFileWriter::data(const char *bytes, size_t n) {
    if (!n) return;
    // a chunk of the data file (32 MB from the staging buffers): slices written with pwrite() on all host threads — one fwrite() copies
    // into the page cache at ~4 GB/s, and the 12.7 GB final DB of a 50 M-read assembly spent 3 s there (round 4)
    if (n < ((size_t) 4 << 20)) { if (fwrite(bytes, 1, n, fd) != n) failed = true; dataPos += n; return; }
    if (fflush(fd) != 0) { failed = true; return; }
    const int fdn = fileno(fd);
    const uint64_t base = dataPos;
    std::atomic<bool> ok(true);
    const size_t SL = (size_t) 2 << 20;
    const size_t nSl = (n + SL - 1) / SL;
    parallelRanges(nSl, [&](int, size_t b, size_t e) {
        for (size_t sl = b; sl < e && ok; sl++) {
            size_t o = sl * SL; const size_t end = std::min(n, o + SL);
            while (o < end) { const ssize_t w = pwrite(fdn, bytes + o, end - o, (off_t) (base + o)); if (w <= 0) { ok = false; break; } o += (size_t) w; }
        }
    }, nullptr, 2);
    if (!ok) { failed = true; return; }
    dataPos += n;
    if (fseeko(fd, (off_t) dataPos, SEEK_SET) != 0) failed = true;
}
void DBFileWriter::index(const uint32_t *keys, const uint32_t *elen, size_t n) {
    if (!n) return;
    const int maxT = hostThreads();
    std::vector<uint64_t> sum((size_t) maxT + 1, 0);
    std::vector<std::pair<size_t, size_t>> range((size_t) maxT, {0, 0});
    const int used = parallelRanges(n, [&](int t, size_t b, size_t e) { uint64_t s = 0; for (size_t i = b; i < e; i++) s += elen[i]; sum[(size_t) t + 1] = s; range[(size_t) t] = {b, e}; });
    sum[0] = off;
    for (int t = 0; t < used; t++) sum[(size_t) t + 1] += sum[(size_t) t];
    std::vector<std::string> text((size_t) used);
    std::vector<std::thread> th;
    auto work = [&](int t) {
        std::string &s = text[(size_t) t]; s.reserve((range[(size_t) t].second - range[(size_t) t].first) * 24);
        uint64_t o = sum[(size_t) t]; char tmp[80];
        for (size_t i = range[(size_t) t].first; i < range[(size_t) t].second; i++) {
            char *q = fmtU32(keys[i], tmp); *q++ = '\t'; q = fmtU64(o, q); *q++ = '\t'; q = fmtU64(elen[i], q); *q++ = '\n';
            s.append(tmp, (size_t) (q - tmp)); o += elen[i];
        }
    };
    for (int t = 1; t < used; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (!ibuf.empty()) { if (fwrite(ibuf.data(), 1, ibuf.size(), fi) != ibuf.size()) failed = true; ibuf.clear(); }
    for (const std::string &s : text) if (!s.empty() && fwrite(s.data(), 1, s.size(), fi) != s.size()) failed = true;
    off = sum[(size_t) used];
}
void DBFileWriter::add(uint32_t key, const char *bytes, size_t n) {
    if (n && fwrite(bytes, 1, n, fd) != n) failed = true;
    if (fputc('\0', fd) == EOF) failed = true;
    dataPos += n + 1;
    char tmp[64]; char *q = fmtU32(key, tmp); *q++ = '\t'; q = fmtU64(off, q); *q++ = '\t'; q = fmtU64(n + 1, q); *q++ = '\n';
    ibuf.append(tmp, (size_t) (q - tmp));
    if (ibuf.size() > (1 << 22) - 128) { if (fwrite(ibuf.data(), 1, ibuf.size(), fi) != ibuf.size()) failed = true; ibuf.clear(); }
    off += n + 1;
}
bool DBFileWriter::close(std::string &err) {
    bool ok = !failed;
    if (fi) { if (!ibuf.empty() && fwrite(ibuf.data(), 1, ibuf.size(), fi) != ibuf.size()) ok = false; ok &= (fflush(fi) == 0); ok &= (fclose(fi) == 0); fi = nullptr; }
    if (fd) { ok &= (fflush(fd) == 0); ok &= (fclose(fd) == 0); fd = nullptr; }
    FILE *ft = fopen((path + ".dbtype" + tmpSuffix).c_str(), "wb");
    if (!ft) ok = false; else { uint32_t ty = (uint32_t) dbtype; ok &= (fwrite(&ty, 4, 1, ft) == 1); ok &= (fclose(ft) == 0); }
    if (ok) for (const char *sfx : {"", ".index", ".dbtype"}) ok &= (rename((path + sfx + tmpSuffix).c_str(), (path + sfx).c_str()) == 0);
    if (!ok) { err = "error while writing " + path + " (disk full?)"; for (const char *sfx : {"", ".index", ".dbtype"}) (void) remove((path + sfx + tmpSuffix).c_str()); }
    open_ = false;
    return ok;
}

bool writeTextDB(const std::string &path, int dbtype, const uint32_t *keys, size_t n, const uint64_t *prefix,
                 const std::function<bool(size_t, std::string &)> &fmt, std::string &err) {
    DBFileWriter w;
    if (!w.open(path, dbtype, err)) return false;
    const int maxT = hostThreads();
    struct Part { std::string text; std::vector<uint32_t> elen; size_t b = 0, e = 0; bool bad = false; };
    std::vector<Part> parts((size_t) maxT);
    const int used = parallelRanges(n, [&](int t, size_t b, size_t e) {
        Part &pt = parts[(size_t) t]; pt.b = b; pt.e = e; pt.elen.reserve(e - b);
        if (prefix) pt.text.reserve((size_t) (prefix[e] - prefix[b]) * 24 + (e - b));
        for (size_t q = b; q < e; q++) {
            const size_t before = pt.text.size();
            if (!fmt(q, pt.text)) { pt.bad = true; return; }
            pt.text.push_back('\0');
            pt.elen.push_back((uint32_t) (pt.text.size() - before));
        }
    }, prefix);
    for (int t = 0; t < used; t++) if (parts[(size_t) t].bad) { err = "formatting failed"; return false; }
    std::vector<uint32_t> elen(n);
    for (int t = 0; t < used; t++) {
        Part &pt = parts[(size_t) t];
        w.data(pt.text.data(), pt.text.size());
        if (!pt.elen.empty()) memcpy(&elen[pt.b], pt.elen.data(), pt.elen.size() * 4);
        std::string().swap(pt.text);
    }
    w.index(keys, elen.data(), n);
    return w.close(err);
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
