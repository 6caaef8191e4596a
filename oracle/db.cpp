// ORACLE (test infrastructure): DBReader/DBWriter on-disk format, restated.
//   reader: mm/commons/DBReader.cpp:150-215 (open), :770-831 (readIndex), FileUtil.cpp:336-352 (findDatafiles)
//   writer: mm/commons/DBWriter.cpp:362-419 (entry + '\0', index line), :193-213 (dbtype)
#include "oracle.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <sys/stat.h>

namespace oracle {

static bool fileExists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

static bool slurp(const std::string &p, std::string &out) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    size_t old = out.size(); out.resize(old + (size_t) n);
    size_t rd = n ? fread(&out[old], 1, (size_t) n, f) : 0;
    fclose(f);
    return rd == (size_t) n;
}

size_t DB::getId(uint32_t k) const {
    auto it = std::lower_bound(key.begin(), key.end(), k);
    if (it == key.end() || *it != k) return (size_t) -1;
    return (size_t) (it - key.begin());
}
uint64_t DB::aminoAcidDBSize() const {
    uint64_t s = 0; for (uint32_t l : elen) s += l; return s - 2 * (uint64_t) key.size();
}
uint32_t DB::maxEntryLen() const { uint32_t m = 0; for (uint32_t l : elen) m = std::max(m, l); return m; }

void DB::add(uint32_t k, const char *bytes, size_t n) {
    key.push_back(k); off.push_back(data.size()); elen.push_back((uint32_t) (n + 1));
    data.append(bytes, n); data.push_back('\0');
}
void DB::sortByKey() {
    std::vector<size_t> p(key.size()); std::iota(p.begin(), p.end(), 0);
    std::stable_sort(p.begin(), p.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    std::vector<uint32_t> k2(p.size()), l2(p.size()); std::vector<uint64_t> o2(p.size());
    for (size_t i = 0; i < p.size(); i++) { k2[i] = key[p[i]]; o2[i] = off[p[i]]; l2[i] = elen[p[i]]; }
    key.swap(k2); off.swap(o2); elen.swap(l2);
}

bool readDB(const std::string &name, DB &db, std::string &err) {
    db = DB();
    std::string t;
    if (!slurp(name + ".dbtype", t) || t.size() < 4) { err = "cannot read " + name + ".dbtype"; return false; }
    uint32_t ty; memcpy(&ty, t.data(), 4);
    if (ty & 0x80000000u) { err = "compressed DBs are not supported (" + name + ")"; return false; }
    db.dbtype = (int) (ty & 0x3FFFFFFF);   // Parameters::isEqualDbtype mask (Parameters.h:1107-1109)
    if (fileExists(name)) {
        if (!slurp(name, db.data)) { err = "cannot read " + name; return false; }
    } else {
        for (int i = 0;; i++) {
            std::string p = name + "." + std::to_string(i);
            if (!fileExists(p)) break;
            if (!slurp(p, db.data)) { err = "cannot read " + p; return false; }
        }
    }
    std::string idx;
    if (!slurp(name + ".index", idx)) { err = "cannot read " + name + ".index"; return false; }
    const char *p = idx.data(), *e = p + idx.size();
    while (p < e) {
        uint64_t v[3] = {0, 0, 0};
        for (int c = 0; c < 3; c++) {
            while (p < e && (*p == '\t' || *p == ' ')) p++;
            while (p < e && *p >= '0' && *p <= '9') v[c] = v[c] * 10 + (uint64_t) (*p++ - '0');
        }
        while (p < e && *p != '\n') p++;
        if (p < e) p++;
        db.key.push_back((uint32_t) v[0]); db.off.push_back(v[1]); db.elen.push_back((uint32_t) v[2]);
    }
    for (size_t i = 0; i < db.key.size(); i++)
        if (db.off[i] + db.elen[i] > db.data.size()) { err = "index entry beyond data in " + name; return false; }
    db.sortByKey();
    return true;
}

bool writeDB(const std::string &name, const DB &db, std::string &err) {
    // canonical layout: one data file, entries in key order, index sorted by key
    std::vector<size_t> p(db.key.size()); std::iota(p.begin(), p.end(), 0);
    std::stable_sort(p.begin(), p.end(), [&](size_t a, size_t b) { return db.key[a] < db.key[b]; });
    FILE *fd = fopen(name.c_str(), "wb"), *fi = fopen((name + ".index").c_str(), "wb"),
         *ft = fopen((name + ".dbtype").c_str(), "wb");
    if (!fd || !fi || !ft) { err = "cannot open output DB " + name; return false; }
    uint64_t o = 0;
    for (size_t i : p) {
        fwrite(db.data.data() + db.off[i], 1, db.elen[i], fd);
        fprintf(fi, "%u\t%llu\t%u\n", db.key[i], (unsigned long long) o, db.elen[i]);
        o += db.elen[i];
    }
    uint32_t ty = (uint32_t) db.dbtype; fwrite(&ty, 4, 1, ft);
    fclose(fd); fclose(fi); fclose(ft);
    return true;
}

// itoa.h convention: writes digits and a terminating '\0', returns pointer PAST the '\0';
// callers then overwrite *(ret-1) with the separator.
char *u32toa(uint32_t v, char *buf) { int n = sprintf(buf, "%u", v); return buf + n + 1; }
char *i32toa(int32_t v, char *buf) { int n = sprintf(buf, "%d", v); return buf + n + 1; }

}  // namespace oracle
