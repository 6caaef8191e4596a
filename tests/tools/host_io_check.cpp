// Host boundary of the product without a GPU (tests/test_host.py compiles this against plass_amd/csrc/host_util.cpp): the threaded DB
// reader / writers must produce and accept exactly the reference's DB files (NAME, NAME.index "key\toffset\tlength\n", NAME.dbtype),
// including data split over NAME.0 .. NAME.k (the reference's unmerged per-thread files, FileUtil.cpp:336-352), whatever the
// number of host threads; writes go through temporary names and leave nothing behind.
#include "../../plass_amd/csrc/host_util.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <dirent.h>
#include <unistd.h>
using namespace plasship;

static std::string slurp(const std::string &p) { std::string s; FILE *f = fopen(p.c_str(), "rb"); if (!f) return s; char b[65536]; size_t g; while ((g = fread(b, 1, sizeof(b), f)) > 0) s.append(b, g); fclose(f); return s; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_io_check: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const size_t N = 200000;
    // entries of 0 .. 40 lines, keys with gaps
    std::vector<uint32_t> keys(N); std::vector<uint64_t> prefix(N + 1, 0);
    for (size_t i = 0; i < N; i++) { keys[i] = (uint32_t) (3 * i + (i % 2)); prefix[i + 1] = prefix[i] + (i * 2654435761u >> 7) % 41; }
    auto fmt = [&](size_t q, std::string &out) { for (uint64_t l = prefix[q]; l < prefix[q + 1]; l++) { char t[64]; int n = snprintf(t, sizeof(t), "%u\t%llu\t-%zu\n", keys[q], (unsigned long long) l, q % 7); out.append(t, (size_t) n); } return true; };
    std::string err, ref, refIdx;
    // reference layout made here, single-threaded
    { uint64_t off = 0; for (size_t q = 0; q < N; q++) { std::string e; fmt(q, e); e.push_back('\0'); ref += e; char t[96]; int n = snprintf(t, sizeof(t), "%u\t%llu\t%zu\n", keys[q], (unsigned long long) off, e.size()); refIdx.append(t, (size_t) n); off += e.size(); } }
    const std::string a = dir + "/textdb";
    CHECK(writeTextDB(a, 7, keys.data(), N, prefix.data(), fmt, err));
    CHECK(slurp(a) == ref); CHECK(slurp(a + ".index") == refIdx);
    { std::string t = slurp(a + ".dbtype"); uint32_t ty = 0; CHECK(t.size() == 4); memcpy(&ty, t.data(), 4); CHECK(ty == 7); }
    // the small-DB interface writes the same files
    { DBFileWriter w; CHECK(w.open(dir + "/adddb", 7, err)); for (size_t q = 0; q < N; q++) { std::string e; fmt(q, e); w.add(keys[q], e.data(), e.size()); } CHECK(w.close(err)); }
    CHECK(slurp(dir + "/adddb") == ref); CHECK(slurp(dir + "/adddb.index") == refIdx);
    // read back: one data file, and the same data split over three files
    HostDB h; CHECK(readDBFiles(a, h, err));
    CHECK(h.dbtype == 7 && h.key.size() == N && h.data.size() == ref.size() && memcmp(h.data.data(), ref.data(), ref.size()) == 0);
    { uint64_t off = 0; for (size_t q = 0; q < N; q++) { CHECK(h.key[q] == keys[q] && h.off[q] == off); off += h.elen[q]; } CHECK(off == ref.size()); }
    const size_t c1 = ref.size() / 3 + 5, c2 = 2 * ref.size() / 3 + 1;
    { const std::string b = dir + "/splitdb"; FILE *f;
      f = fopen((b + ".0").c_str(), "wb"); fwrite(ref.data(), 1, c1, f); fclose(f);
      f = fopen((b + ".1").c_str(), "wb"); fwrite(ref.data() + c1, 1, c2 - c1, f); fclose(f);
      f = fopen((b + ".2").c_str(), "wb"); fwrite(ref.data() + c2, 1, ref.size() - c2, f); fclose(f);
      f = fopen((b + ".index").c_str(), "wb"); fwrite(refIdx.data(), 1, refIdx.size(), f); fclose(f);
      f = fopen((b + ".dbtype").c_str(), "wb"); uint32_t ty = 7; fwrite(&ty, 4, 1, f); fclose(f);
      HostDB s; CHECK(readDBFiles(b, s, err));
      CHECK(s.data.size() == ref.size() && memcmp(s.data.data(), ref.data(), ref.size()) == 0 && s.key == h.key && s.off == h.off && s.elen == h.elen); }
    // an index entry beyond the data is refused
    { const std::string b = dir + "/baddb"; FILE *f = fopen(b.c_str(), "wb"); fwrite("abc\n\0", 1, 5, f); fclose(f);
      f = fopen((b + ".index").c_str(), "wb"); fputs("0\t0\t5\n1\t5\t9\n", f); fclose(f);
      f = fopen((b + ".dbtype").c_str(), "wb"); uint32_t ty = 0; fwrite(&ty, 4, 1, f); fclose(f);
      HostDB s; CHECK(!readDBFiles(b, s, err)); }
    // a writer that is abandoned, or whose formatter fails, leaves neither the DB nor temporary files
    { DBFileWriter w; CHECK(w.open(dir + "/gone", 0, err)); w.add(1, "x\n", 2); }
    CHECK(!writeTextDB(dir + "/gone2", 7, keys.data(), N, prefix.data(), [&](size_t q, std::string &o) { return q != N / 2 && fmt(q, o); }, err));
    CHECK(!writeTextDB(dir + "/no/such/dir/db", 7, keys.data(), N, prefix.data(), fmt, err));
    { DIR *d = opendir(dir.c_str()); CHECK(d); while (dirent *e = readdir(d)) { const std::string n = e->d_name; CHECK(n.find(".tmp.") == std::string::npos); CHECK(n.rfind("gone", 0) != 0); } closedir(d); }
    printf("host_io_check ok (%d host threads)\n", hostThreads());
    return 0;
}
