// plasship: prefilter / alignment DB text formats <-> device lists (C-ABI part 2).  Product code.
//   hit_t line       "seqId\tprefScore\t(int16)diagonal\n"            mm/prefiltering/QueryMatcher.h:81-126
//   alignment line   "dbKey\tbits\tseqId\teval\tqS\tqE\tqLen\ttS\ttE\ttLen[\t<n>M]\n"   mm/alignment/Matcher.cpp:248-370
//   seqId text       truncated to 3 decimals, 1.0 printed as "1.00" (Util.cpp:278-307 + Matcher.cpp:329-330)
#include "common.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <atomic>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace plasship;

static int hostKeys(plasship_ctx *ctx, const plasship_seqdb *cdb, const std::vector<uint32_t> **keys) {
    plasship_seqdb *db = const_cast<plasship_seqdb *>(cdb);
    if (!db->hostIndexValid && db->h_key.size() != db->n) {
        db->h_key.resize(db->n);
        PH_CHECK(plasship::streamSync(ctx->stream));
        if (db->n) { const int rc = stagedCopyToHost(ctx, db->h_key.data(), db->d_key.p, db->n * 4); if (rc) return rc; }
    }
    *keys = &db->h_key;
    return PLASSHIP_OK;
}

static inline long keyToId(const std::vector<uint32_t> &keys, uint32_t k) {
    auto it = std::lower_bound(keys.begin(), keys.end(), k);
    if (it == keys.end() || *it != k) return -1;
    return (long) (it - keys.begin());
}

// ---- candidates -------------------------------------------------------------------------------------
extern "C" int plasship_cands_count(const plasship_cands *c, uint64_t *n_hits, int *reverse_capable) {
    if (!c) { setError("plasship_cands_count: NULL"); return PLASSHIP_ERR_ARG; }
    if (n_hits) *n_hits = c->nNonSelf;
    if (reverse_capable) *reverse_capable = c->reverseCapable ? 1 : 0;
    return PLASSHIP_OK;
}

extern "C" void plasship_cands_free(plasship_ctx *ctx, plasship_cands *c) {
    if (!c) return;
    if (ctx) { (void) hipSetDevice(ctx->device); plasship::poolEnter(ctx->stream); }
    delete c;
}

extern "C" int plasship_cands_read(plasship_ctx *ctx, const plasship_seqdb *qdb, const plasship_seqdb *tdb,
                                   const char *db_path, plasship_cands **out) {
    if (!ctx || !qdb || !tdb || !db_path || !out) { setError("plasship_cands_read: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    HostDB h; std::string err;
    if (!readDBFiles(db_path, h, err)) { setError(err); return PLASSHIP_ERR_IO; }
    if (h.dbtype != PLASSHIP_DBTYPE_PREFILTER_RES && h.dbtype != PLASSHIP_DBTYPE_PREFILTER_REV_RES) {
        setError(std::string("not a prefilter DB: ") + db_path); return PLASSHIP_ERR_ARG;
    }
    const std::vector<uint32_t> *qk, *tk;
    int rc = hostKeys(ctx, qdb, &qk); if (rc) return rc;
    rc = hostKeys(ctx, tdb, &tk); if (rc) return rc;
    const size_t nQ = qdb->n;
    // entries -> query ids, lines per query, then the lines themselves: all three on the host threads (the reference parses its
    // prefilter DB on its OpenMP threads, one query at a time)
    std::vector<long> entryOf(nQ, -1);
    std::atomic<int> bad(0);
    parallelRanges(h.key.size(), [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; i++) { const long id = keyToId(*qk, h.key[i]); if (id < 0) { bad = 1; return; } entryOf[(size_t) id] = (long) i; }
    });
    if (bad) { setError("prefilter entry for a key that is not in the query DB"); return PLASSHIP_ERR_ARG; }
    std::vector<uint64_t> qoff(nQ + 1, 0);
    parallelRanges(nQ, [&](int, size_t b, size_t e) {
        for (size_t q = b; q < e; q++) {
            if (entryOf[q] < 0) continue;
            const char *p = h.data.data() + h.off[(size_t) entryOf[q]]; uint64_t lines = 0;
            while (*p != '\0') { lines++; while (*p != '\n' && *p != '\0') p++; if (*p == '\n') p++; }
            qoff[q + 1] = lines;
        }
    });
    for (size_t q = 0; q < nQ; q++) qoff[q + 1] += qoff[q];
    std::vector<CandHit> hits(qoff[nQ]);
    std::vector<uint64_t> nonSelfPart((size_t) hostThreads(), 0);
    parallelRanges(nQ, [&](int t, size_t b, size_t e) {
        uint64_t ns = 0;
        for (size_t q = b; q < e; q++) {
            if (entryOf[q] < 0) continue;
            const char *p = h.data.data() + h.off[(size_t) entryOf[q]]; uint64_t at = qoff[q];
            while (*p != '\0') {
                uint32_t key = 0; while (*p >= '0' && *p <= '9') key = key * 10 + (uint32_t) (*p++ - '0');
                while (*p == '\t' || *p == ' ') p++;
                int sg = 1; if (*p == '-') { sg = -1; p++; }
                int sc = 0; while (*p >= '0' && *p <= '9') sc = sc * 10 + (*p++ - '0');
                while (*p == '\t' || *p == ' ') p++;
                int sg2 = 1; if (*p == '-') { sg2 = -1; p++; }
                short dg = 0; while (*p >= '0' && *p <= '9') dg = (short) (dg * 10 + (*p++ - '0'));
                while (*p != '\n' && *p != '\0') p++;
                if (*p == '\n') p++;
                const long tid = keyToId(*tk, key);
                if (tid < 0) { bad = 2; return; }
                CandHit ch; ch.target = (uint32_t) tid; ch.prefScore = sg * sc; ch.diag16 = (uint32_t) (uint16_t) (short) (sg2 * dg); ch.query = (uint32_t) q;
                hits[at++] = ch;
                if (!(qdb == tdb && (size_t) tid == q)) ns++;
            }
        }
        nonSelfPart[(size_t) t] = ns;
    }, qoff.data());
    if (bad) { setError("prefilter hit for a key that is not in the target DB"); return PLASSHIP_ERR_ARG; }
    uint64_t nonSelf = 0; for (uint64_t v : nonSelfPart) nonSelf += v;
    std::unique_ptr<plasship_cands> holder(new plasship_cands());   // released to the caller on success only
    plasship_cands *c = holder.get();
    c->reverseCapable = (h.dbtype == PLASSHIP_DBTYPE_PREFILTER_REV_RES);
    c->nQueries = nQ; c->nHits = hits.size(); c->nNonSelf = nonSelf;
    if (c->d_qoff.alloc((nQ + 1) * 8) != hipSuccess || c->d_hits.alloc(std::max<size_t>(hits.size(), 1) * sizeof(CandHit)) != hipSuccess) {
        setError("plasship_cands_read: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    rc = stagedCopyToDevice(ctx, c->d_qoff.p, qoff.data(), (nQ + 1) * 8); if (rc) return rc;
    rc = stagedCopyToDevice(ctx, c->d_hits.p, hits.data(), hits.size() * sizeof(CandHit)); if (rc) return rc;
    *out = holder.release();
    return PLASSHIP_OK;
}

static int fetchCands(plasship_ctx *ctx, const plasship_cands *c, std::vector<uint64_t> &qoff, std::vector<CandHit> &hits) {
    qoff.resize(c->nQueries + 1); hits.resize(c->nHits);
    PH_CHECK(plasship::streamSync(ctx->stream));
    int rc = stagedCopyToHost(ctx, qoff.data(), c->d_qoff.p, (c->nQueries + 1) * 8); if (rc) return rc;
    return stagedCopyToHost(ctx, hits.data(), c->d_hits.p, c->nHits * sizeof(CandHit));
}

extern "C" int plasship_cands_write(plasship_ctx *ctx, const plasship_cands *c, const plasship_seqdb *db, const char *db_path) {
    if (!ctx || !c || !db || !db_path) { setError("plasship_cands_write: bad argument"); return PLASSHIP_ERR_ARG; }
    if (c->nQueries != db->n) { setError("plasship_cands_write: DB mismatch"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    const std::vector<uint32_t> *keys; int rc = hostKeys(ctx, db, &keys); if (rc) return rc;
    std::vector<uint64_t> qoff; std::vector<CandHit> hits;
    rc = fetchCands(ctx, c, qoff, hits); if (rc) return rc;
    std::string err;
    const bool ok = writeTextDB(db_path, c->reverseCapable ? PLASSHIP_DBTYPE_PREFILTER_REV_RES : PLASSHIP_DBTYPE_PREFILTER_RES, keys->data(), c->nQueries, qoff.data(),
                                [&](size_t q, std::string &out) {
        for (uint64_t i = qoff[q]; i < qoff[q + 1]; i++) {
            char tmp[64]; char *p = fmtU32((*keys)[hits[i].target], tmp); *p++ = '\t';
            p = fmtI32(hits[i].prefScore, p); *p++ = '\t';
            p = fmtI32((int32_t) (int16_t) (uint16_t) hits[i].diag16, p); *p++ = '\n';
            out.append(tmp, (size_t) (p - tmp));
        }
        return true;
    }, err);
    if (!ok) { setError(err); return PLASSHIP_ERR_IO; }
    return PLASSHIP_OK;
}

extern "C" int plasship_cands_download(plasship_ctx *ctx, const plasship_cands *c, const plasship_seqdb *qdb,
                                       const plasship_seqdb *tdb, uint32_t *query_key, uint32_t *target_key,
                                       int32_t *pref_score, uint16_t *diagonal) {
    if (!ctx || !c || !qdb || !tdb) { setError("plasship_cands_download: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    const std::vector<uint32_t> *qk, *tk;
    int rc = hostKeys(ctx, qdb, &qk); if (rc) return rc;
    rc = hostKeys(ctx, tdb, &tk); if (rc) return rc;
    std::vector<uint64_t> qoff; std::vector<CandHit> hits;
    rc = fetchCands(ctx, c, qoff, hits); if (rc) return rc;
    uint64_t o = 0;
    for (uint64_t i = 0; i < c->nHits; i++) {
        if (qdb == tdb && hits[i].query == hits[i].target && hits[i].prefScore == 0 && hits[i].diag16 == 0) continue;   // implicit self line (same DB only: ids of two DBs are unrelated)
        if (query_key) query_key[o] = (*qk)[hits[i].query];
        if (target_key) target_key[o] = (*tk)[hits[i].target];
        if (pref_score) pref_score[o] = hits[i].prefScore;
        if (diagonal) diagonal[o] = (uint16_t) hits[i].diag16;
        o++;
    }
    return PLASSHIP_OK;
}

// ---- alignments ---------------------------------------------------------------------------------------
extern "C" int plasship_alns_count(const plasship_alns *a, uint64_t *n_lines) {
    if (!a) { setError("plasship_alns_count: NULL"); return PLASSHIP_ERR_ARG; }
    if (n_lines) *n_lines = a->nLines;
    return PLASSHIP_OK;
}
extern "C" void plasship_alns_free(plasship_ctx *ctx, plasship_alns *a) {
    if (!a) return;
    if (ctx) { (void) hipSetDevice(ctx->device); plasship::poolEnter(ctx->stream); }
    delete a;
}

static int fetchAlns(plasship_ctx *ctx, const plasship_alns *a, std::vector<uint64_t> &qoff, std::vector<AlnRec> &recs) {
    qoff.resize(a->nQueries + 1); recs.resize(a->nLines);
    // a list made by plasship_rescore is sparse (common.hpp: plasship_alns): the host sees a dense copy, made on the device
    DevBuf dQoff, dRecs; const uint64_t *pq = nullptr; const AlnRec *pr = nullptr;
    int rc = finishSelfAlns(ctx, a); if (rc) return rc;          // (identity pairs left as stubs are scored now: the host reads every record)
    rc = denseAlnsCopy(ctx, a, dQoff, dRecs, &pq, &pr); if (rc) return rc;
    PH_CHECK(plasship::streamSync(ctx->stream));
    rc = stagedCopyToHost(ctx, qoff.data(), pq, (a->nQueries + 1) * 8); if (rc) return rc;
    return stagedCopyToHost(ctx, recs.data(), pr, a->nLines * sizeof(AlnRec));
}

extern "C" int plasship_alns_download(plasship_ctx *ctx, const plasship_alns *a, plasship_aln_record *out) {
    if (!ctx || !a || !out) { setError("plasship_alns_download: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    std::vector<uint64_t> qoff; std::vector<AlnRec> recs;
    int rc = fetchAlns(ctx, a, qoff, recs); if (rc) return rc;
    const std::vector<uint32_t> *qk, *tk;
    rc = hostKeys(ctx, a->qdb, &qk); if (rc) return rc;
    rc = hostKeys(ctx, a->tdb, &tk); if (rc) return rc;
    parallelRanges(a->nLines, [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) {
        const AlnRec &r = recs[i]; plasship_aln_record &o = out[i];
        o.query_key = (*qk)[r.query]; o.target_key = (*tk)[r.target];
        o.bit_score = r.bitScore; o.raw_score = r.rawScore; o.seq_id = r.seqId;
        o.q_start = r.qStart; o.q_end = r.qEnd; o.q_len = r.qLen; o.db_start = r.dbStart; o.db_end = r.dbEnd; o.db_len = r.dbLen;
        o.aln_len = r.alnLen; o.reversed = r.reversed;
    } });
    return PLASSHIP_OK;
}

// Util::fastSeqIdToBuffer + the separator write of Matcher::resultToBuffer
static char *fmtSeqId(float seqId, char *p) {
    if (seqId == 1.0f) { memcpy(p, "1.00", 4); return p + 4; }   // the '\t' lands on the third zero
    *p++ = '0'; *p++ = '.';
    if ((double) seqId < 0.10) *p++ = '0';
    if ((double) seqId < 0.01) *p++ = '0';
    return fmtI32((int) (seqId * 1000), p);
}

extern "C" int plasship_alns_write(plasship_ctx *ctx, const plasship_alns *a, const char *db_path) {
    if (!ctx || !a || !db_path) { setError("plasship_alns_write: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    std::vector<uint64_t> qoff; std::vector<AlnRec> recs;
    int rc = fetchAlns(ctx, a, qoff, recs); if (rc) return rc;
    const std::vector<uint32_t> *qk, *tk;
    rc = hostKeys(ctx, a->qdb, &qk); if (rc) return rc;
    rc = hostKeys(ctx, a->tdb, &tk); if (rc) return rc;
    HostEvaluer ev(a->nucl, a->dbResidues);
    if (a->gappedOpen && !HostEvaluer::nuclGapped(a->gappedOpen, a->gappedExtend, a->dbResidues, ev)) { setError("plasship_alns_write: no Gumbel parameters for these gap penalties"); return PLASSHIP_ERR_UNSUPPORTED; }
    std::string err; std::atomic<int> fromText(0);
    // E-values (erfc / exp per line) and the text are made on the host threads, like the reference's writer threads
    const bool ok = writeTextDB(db_path, PLASSHIP_DBTYPE_ALIGNMENT_RES, qk->data(), a->nQueries, qoff.data(), [&](size_t q, std::string &out) {
        for (uint64_t i = qoff[q]; i < qoff[q + 1]; i++) {
            const AlnRec &r = recs[i];
            if (r.fromText) { fromText = 1; return false; }
            char tmp[256]; char *p = fmtU32((*tk)[r.target], tmp); *p++ = '\t';
            p = fmtI32(r.bitScore, p); *p++ = '\t';
            p = fmtSeqId(r.seqId, p); *p++ = '\t';
            p += snprintf(p, 32, "%.3E", ev.evalue((double) r.rawScore, (double) r.qLen)); *p++ = '\t';
            p = fmtI32(r.qStart, p); *p++ = '\t'; p = fmtI32(r.qEnd, p); *p++ = '\t'; p = fmtI32(r.qLen, p); *p++ = '\t';
            p = fmtI32(r.dbStart, p); *p++ = '\t'; p = fmtI32(r.dbEnd, p); *p++ = '\t'; p = fmtI32(r.dbLen, p);
            if (a->addBacktrace) { *p++ = '\t'; p = fmtI32(r.alnLen, p); *p++ = 'M'; }
            *p++ = '\n';
            out.append(tmp, (size_t) (p - tmp));
        }
        return true;
    }, err);
    if (fromText) { setError("plasship_alns_write: list was read from text (no raw scores)"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (!ok) { setError(err); return PLASSHIP_ERR_IO; }
    return PLASSHIP_OK;
}

extern "C" int plasship_alns_read(plasship_ctx *ctx, const plasship_seqdb *db, const char *db_path, plasship_alns **out) {
    if (!ctx || !db || !db_path || !out) { setError("plasship_alns_read: bad argument"); return PLASSHIP_ERR_ARG; }
    PH_ENTER(ctx);
    HostDB h; std::string err;
    if (!readDBFiles(db_path, h, err)) { setError(err); return PLASSHIP_ERR_IO; }
    if (h.dbtype != PLASSHIP_DBTYPE_ALIGNMENT_RES) { setError(std::string("not an alignment DB: ") + db_path); return PLASSHIP_ERR_ARG; }
    const std::vector<uint32_t> *keys; int rc = hostKeys(ctx, db, &keys); if (rc) return rc;
    const size_t nQ = db->n;
    std::vector<long> entryOf(nQ, -1);
    parallelRanges(h.key.size(), [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) { const long id = keyToId(*keys, h.key[i]); if (id >= 0) entryOf[(size_t) id] = (long) i; } });
    std::vector<uint64_t> qoff(nQ + 1, 0);
    parallelRanges(nQ, [&](int, size_t b, size_t e) {
        for (size_t q = b; q < e; q++) {
            if (entryOf[q] < 0) continue;
            const char *p = h.data.data() + h.off[(size_t) entryOf[q]]; uint64_t lines = 0;
            while (*p != '\0') { lines++; while (*p != '\n' && *p != '\0') p++; if (*p == '\n') p++; }
            qoff[q + 1] = lines;
        }
    });
    for (size_t q = 0; q < nQ; q++) qoff[q + 1] += qoff[q];
    std::vector<AlnRec> recs(qoff[nQ]);
    std::atomic<int> bad(0);
    parallelRanges(nQ, [&](int, size_t qb, size_t qe) {
    for (size_t q = qb; q < qe; q++) {
        if (entryOf[q] < 0) continue;
        const char *p = h.data.data() + h.off[(size_t) entryOf[q]]; uint64_t at = qoff[q];
        while (*p != '\0') {
            const char *f[16]; int nf = 0; const char *s = p;
            while (*s != '\n' && *s != '\0' && nf < 15) {
                while (*s == ' ' || *s == '\t') s++;
                f[nf++] = s;
                while (*s != ' ' && *s != '\t' && *s != '\n' && *s != '\0') s++;
            }
            if (nf < 10) { bad = 1; return; }
            AlnRec r; memset(&r, 0, sizeof(r));
            long tid = keyToId(*keys, (uint32_t) strtoul(f[0], nullptr, 10));
            if (tid < 0) { bad = 2; return; }
            r.query = (uint32_t) q; r.target = (uint32_t) tid; r.bitScore = atoi(f[1]); r.rawScore = -1;
            r.seqId = (float) strtod(f[2], nullptr);
            r.qStart = atoi(f[4]); r.qEnd = atoi(f[5]); r.qLen = atoi(f[6]); r.dbStart = atoi(f[7]); r.dbEnd = atoi(f[8]); r.dbLen = atoi(f[9]);
            int aq = r.qStart == -1 ? 0 : r.qStart, ad = r.dbStart == -1 ? 0 : r.dbStart;
            r.alnLen = std::max(std::abs(r.qEnd - aq), std::abs(r.dbEnd - ad)) + 1;          // Matcher::computeAlnLength
            r.reversed = 0; r.accepted = 1; r.fromText = 1;
            if (nf >= 11) {     // backtrace column: "<alnLen>M" is what rescorediagonal -a 1 writes (DistanceCalculator: ungapped)
                const char *b = f[10]; long v = 0; bool digits = false;
                while (*b >= '0' && *b <= '9') { v = v * 10 + (*b - '0'); b++; digits = true; }
                const bool oneRun = digits && *b == 'M' && (b[1] == '\n' || b[1] == '\0' || b[1] == ' ' || b[1] == '\t');
                r.btKind = (oneRun && v == r.alnLen) ? 1 : 2;
            }
            recs[at++] = r;
            while (*p != '\n' && *p != '\0') p++;
            if (*p == '\n') p++;
        }
    }
    }, qoff.data());
    if (bad == 1) { setError("invalid alignment record"); return PLASSHIP_ERR_ARG; }
    if (bad == 2) { setError("alignment line for a key that is not in the DB"); return PLASSHIP_ERR_ARG; }
    std::unique_ptr<plasship_alns> holder(new plasship_alns());     // released to the caller on success only
    plasship_alns *a = holder.get();
    a->nQueries = nQ; a->nLines = recs.size(); a->nSlots = recs.size(); a->nucl = db->dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES; a->dbResidues = db->residues;
    a->qdb = db; a->tdb = db;
    if (a->d_qoff.alloc((nQ + 1) * 8) != hipSuccess || a->d_recs.alloc(std::max<size_t>(recs.size(), 1) * sizeof(AlnRec)) != hipSuccess) {
        setError("plasship_alns_read: out of device memory"); return PLASSHIP_ERR_DEVICE;
    }
    rc = stagedCopyToDevice(ctx, a->d_qoff.p, qoff.data(), (nQ + 1) * 8); if (rc) return rc;
    rc = stagedCopyToDevice(ctx, a->d_recs.p, recs.data(), recs.size() * sizeof(AlnRec)); if (rc) return rc;
    *out = holder.release();
    return PLASSHIP_OK;
}
