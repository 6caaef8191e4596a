// plasship internal declarations (product code; never includes anything from oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/plasship.h"

namespace plasship {

void setError(const std::string &msg);
// every wait of the host for the stream goes through here and is counted (plasship_host_syncs(): bench.py reports waits per iteration)
hipError_t streamSync(hipStream_t st);
}
struct plasship_seqdb; struct plasship_ctx;
namespace plasship {
int ensureOffLen(plasship_ctx *ctx, const plasship_seqdb *db);     // core.hip: builds plasship_seqdb::d_offLen once (stream-ordered)
std::string hipErrStr(hipError_t e, const char *what, const char *file, int line);

#define PH_CHECK(call)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            plasship::setError(plasship::hipErrStr(e_, #call, __FILE__, __LINE__));     \
            return PLASSHIP_ERR_DEVICE;                                                  \
        }                                                                                \
    } while (0)
// The context stream is non-blocking: a plain hipMemcpy runs on the null stream and is NOT ordered with work queued on
// it.  Every blocking copy in the library goes through this: queued on the context stream, then waited for.
#define PH_COPY_SYNC(st, dst, src, bytes, kind)                                          \
    do {                                                                                 \
        PH_CHECK(hipMemcpyAsync((dst), (src), (bytes), (kind), (st)));                   \
        PH_CHECK(plasship::streamSync(st));                                              \
    } while (0)

// launch-geometry knobs (workgroups per CU of a persistent kernel's grid): default unless PLASSHIP_TUNE_<name> is set in the
// environment (tools/tune_sweep.sh).  The defaults were swept on the 1 M-read set: the best grid is the number of workgroups
// a CU holds at once, or a small multiple — one more leaves a tail round (extractShortKernel: 18 -> 0.69 ms, 20 -> 0.81 ms)
int tuneInt(const char *name, int dflt);
// PLASSHIP_TRACE=1: wait for the stream at every marked stage boundary and say so on stderr (localises a faulting kernel)
bool traceOn();
#define PH_TRACE(st, what)                                                                \
    do {                                                                                 \
        if (plasship::traceOn()) {                                                       \
            hipError_t e_ = plasship::streamSync(st);                                    \
            fprintf(stderr, "[plasship] %s: %s\n", (what), hipGetErrorString(e_));       \
        }                                                                                \
    } while (0)

// Caching device allocator: hipMalloc/hipFree synchronise the device and cost 0.1–1 ms each, which would
// dominate an assembly iteration on a 1 M-read set.  Freed blocks are kept (size classes with <= 12.5 % slack)
// and handed out again; everything is returned to HIP when the last context is destroyed or on out-of-memory.
// longLived: a buffer that outlives the module call that makes it (sequence DBs and their heaps, the selected-window cache).  It is
// placed at the TOP of the highest free range that holds it instead of at the front of the best-fitting one, so that the long-lived
// buffers collect at one end of the arena and the 60 GB record arrays of kmermatcher keep finding contiguous room (round 4: with
// heaps and cache lines scattered over the arena the 50 M-read chain ran out of memory with 80 GB free).
hipError_t poolMalloc(void **p, size_t n, bool longLived = false);
// every C-ABI entry that allocates calls this first (PH_ENTER): the stream the calling thread's allocations belong to
void poolEnter(hipStream_t stream);
// sharded run (round 6): which rank the calling thread's context is, so that a wait for the stream that never ends — a collective a peer
// never joined, a link that is down: what the FIRST run on real xGMI links can meet — becomes an error with the rank and the last
// collective in it instead of a hang (core.hip: streamSync polls with a deadline when a communicator of more than one rank is installed;
// PLASSHIP_COMM_TIMEOUT_S, default 300, 0 = wait for ever).  watchCollective names what was enqueued last.
void watchEnter(const plasship_ctx *ctx);
void watchCollective(const char *what);
#define PH_ENTER(ctx)                                                                    \
    do {                                                                                 \
        PH_CHECK(hipSetDevice((ctx)->device));                                           \
        plasship::poolEnter((ctx)->stream);                                              \
        plasship::watchEnter(ctx);                                                       \
    } while (0)
void poolFree(void *p);
void poolTrim();

// RAII device buffer (untyped bytes)
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n, bool longLived = false) { release(); bytes = n; if (n == 0) { p = nullptr; return hipSuccess; } return poolMalloc(&p, n, longLived); }
    hipError_t allocLong(size_t n) { return alloc(n, true); }
    void release() { if (p) { poolFree(p); p = nullptr; } bytes = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- device-side views (POD, passed to kernels by value) --------------------------------------
struct SeqView {
    const char *data;          // entries "SEQ\n\0"
    const uint64_t *off;       // [n] byte offset of entry i (id = rank in key order)
    const uint32_t *len;       // [n] sequence length (entry length - 2)
    const uint64_t *offLen;    // [n] off << 24 | len in one word, or nullptr (ensureOffLen): ONE line instead of two for a random entry
    uint32_t n;
    int nucl;
};

#ifdef __HIPCC__
// offset / length of a RANDOM entry from the packed word (requires ensureOffLen on the DB)
__device__ __forceinline__ uint64_t seqOff(const SeqView &s, uint32_t id) { return s.offLen[id] >> 24; }
__device__ __forceinline__ uint32_t seqLen(const SeqView &s, uint32_t id) { return (uint32_t) s.offLen[id] & 0xFFFFFFu; }
#endif

// one candidate pair (query id implied by CSR)
struct __attribute__((aligned(8))) CandHit { uint32_t target; int32_t prefScore; uint32_t diag16; uint32_t query; };

// one scored/accepted alignment (device + host layout)
struct AlnRec {
    uint32_t query, target;    // ids (ranks)
    int32_t bitScore, rawScore;
    float seqId;               // exact float (or parsed text value if fromText)
    int32_t qStart, qEnd, qLen, dbStart, dbEnd, dbLen, alnLen;
    int32_t reversed;
    int32_t accepted;
    int32_t fromText;
    int32_t btKind;            // backtrace of the record: 0 unknown / none, 1 one run of alnLen 'M' (ungapped), 2 anything else,
                               // ALN_SELF_PENDING: an identity pair not scored yet (plasship_alns::selfPending)
};
constexpr int32_t ALN_SELF_PENDING = 3;

}  // namespace plasship

struct plasship_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[16] = {};
    int numCU = 0;
    // nuclassembleresults: comparator decisions that lie on a threshold, resolved with the host libm (assemble.hip);
    // kept for the lifetime of the context (the same few (alpha, beta) tuples recur in every iteration)
    std::vector<uint32_t> ambKeys;      // 4 words per entry
    std::vector<uint8_t> ambVals;
    plasship::DevBuf d_ambKeys, d_ambVals;
    plasship::DevBuf d_cmpCache;        // memo of the nucleotide comparator's posterior classes (assemble.hip, nuclLess)
    uint32_t ambSlots = 0;              // power of two, 0 = no table yet
    // host boundary: two pinned staging buffers (allocated on first use) every bulk host<->device copy goes through — a chunk is
    // packed / consumed on the host threads while the previous one is on the PCIe link (stagedUpload / stagedDownload, core.hip)
    char *stage[2] = {nullptr, nullptr};
    hipEvent_t stageEv[2] = {nullptr, nullptr};
    size_t stageBytes = 0;
    void *pinnedTable = nullptr;        // 1 MB of pinned host memory for small tables a kernel chain reads (ctxPinnedTable, core.hip)
    // one read set sharded over several GPUs (plasship_ctx_set_comm); world == 1 and hasComm == false: single GPU
    bool hasComm = false;
    plasship_comm comm = {};
    int debugFailCollective = -1;       // plasship_ctx_debug_fail_collective: countdown to an injected rank-local failure
    // kmermatcher's selected-window cache (kmermatch.hip section 2c): per sequence one 128-byte line {identity-record hash, the positions
    // of the <= 60 selected k-mers} of the LAST plasship_kmermatch call on this context; the next call re-uses the line of every
    // sequence its DB inherited unchanged from that call's DB when the selection parameters (hash seed included) are the same
    struct KmCache {
        plasship::DevBuf lines; uint64_t n = 0, gen = 0; bool valid = false, seenEligible = false;      // seenEligible: an earlier call on this context could have used lines (they are allocated from the second such call on, or for a derived DB)
        int k = 0, alph = 0, kps = 0, ignoreMulti = 0, hashShift = 0;
    } kmCache;
    // kmermatcher's position cache of nucleotide runs (kmermatch_extract.hpp section 2d): per record slot of the LAST nucleotide
    // plasship_kmermatch call on this context the window position it selected (the identity slot: the count), that call's slot offsets
    // and identity hashes.  `gen` (that call's DB) is the ANCHOR: every DB buildOutputDB derives from it, directly or through other derived
    // DBs, carries per entry the id it had in the anchor (plasship_seqdb::d_origin) — entries may be dropped on the way.
    struct KmPosCache {
        plasship::DevBuf pos, slotOff, idHash; uint64_t n = 0, gen = 0; bool valid = false, seen = false, lng = false;
        int k = 0, kps = 0, ignoreMulti = 0, hashShift = 0; float scale = 0;
    } kmPosCache;
    uint64_t kmermatchCalls = 0;       // plasship_kmermatch calls this context has seen (a second call suggests a chain: the caches are written from then on)
    // cyclecheck.hip: generation of the last "rest" DB plasship_cyclecheck made on this context (every entry of it is known not to be
    // circular at that --max-seq-len); a later call on a DB that descends from it checks only the entries rewritten since
    uint64_t cycKnownGen = 0, cycKnownMaxLen = 0;
};

namespace plasship {
uint64_t newDbGeneration();
// An append-only byte heap shared by a chain of sequence DBs (assemble.hip, buildOutputDB): an assembly iteration changes 10-20 % of
// the sequences, so the DB it makes keeps the bytes of the unchanged entries where they are — its index points into the heap of the
// DB it derives from — and only the extended entries are appended behind `used`.  Nothing is ever overwritten, so every DB of the
// chain stays valid for as long as its handle lives; the buffer goes when the last of them does.
struct SeqHeap { DevBuf buf; std::atomic<uint64_t> used{0}; };
}      // core.hip: a process-wide counter; every sequence DB handle gets its own number
struct plasship_seqdb {
    int dbtype = 0;
    // Lineage (kmermatch.hip, the selected-window cache): `gen` names this handle; a DB that buildOutputDB derived from another one
    // WITHOUT dropping entries (same ids, same keys) names it in `parentGen` and marks in d_changed (one byte per id) the entries whose
    // bytes differ from the parent's.  0 = no such parent (read from disk, generated, concatenated, entries dropped).
    // `ancestorGen` is the weaker statement that survives dropped entries (nuclassembleresults removes the consumed targets): every
    // entry with d_changed == 0 is byte for byte an entry of the DB with that generation, under whatever id (cyclecheck.hip).
    uint64_t gen = plasship::newDbGeneration(), parentGen = 0, ancestorGen = 0;
    plasship::DevBuf d_changed;
    // `originGen` != 0: d_origin[i] = the id entry i had in the DB of that generation (the anchor of kmermatcher's position cache,
    // plasship_ctx::kmPosCache), 0xFFFFFFFF if its bytes are not an entry of that DB; survives dropped entries and chains of derivations
    uint64_t originGen = 0;
    plasship::DevBuf d_origin;
    size_t n = 0;
    uint64_t dataBytes = 0, residues = 0;
    uint32_t maxEntryLen = 0;
    plasship::DevBuf d_data, d_off, d_len, d_key;
    // heap != nullptr: the entries' bytes are in heap->buf (d_data is empty) at d_off[i], NOT necessarily back to back or in key order;
    // `contiguous` says whether they are (a freshly compacted heap is; then, like d_data, the bytes [0, dataBytes) are the data file)
    std::shared_ptr<plasship::SeqHeap> heap;
    bool contiguous = true;
    uint64_t buildAppendedBytes = 0, buildCopiedBytes = 0;  // how buildOutputDB made this DB: bytes appended to the shared heap / bytes of a full copy
    const char *dataPtr() const { return heap ? heap->buf.as<char>() : d_data.as<char>(); }
    // rank of every entry in DATA FILE order (empty: the file lay in key order, rank == id).  Only DBs read from files written by
    // several threads have one; concatdbs renumbers its second DB by it (DBConcat.cpp:46-47,113-118 opens it LINEAR_ACCCESS).
    plasship::DevBuf d_fileRank;
    // offset and length of every entry packed into one word (plasship::ensureOffLen): the kernels that look up RANDOM entries
    // (the targets of candidate pairs and alignments) fetch one line per entry instead of one of d_off and one of d_len
    mutable plasship::DevBuf d_offLen;
    // (built lazily on the stream of the first context that needs it; a second context on the same handle — the chain driver's writer
    //  thread, ranks sharing a DB — takes the lock and waits for that kernel on its own stream: ensureOffLen, core.hip)
    mutable std::mutex offLenMu; mutable hipEvent_t offLenEv = nullptr; mutable hipStream_t offLenStream = nullptr;
    ~plasship_seqdb() { if (offLenEv) (void) hipEventDestroy(offLenEv); }
    // host mirror of the index (lazily filled for device-produced DBs)
    bool hostIndexValid = false;
    std::vector<uint32_t> h_key, h_elen;
    std::vector<uint64_t> h_off;
    plasship::SeqView view() const {
        plasship::SeqView v; v.data = dataPtr(); v.off = d_off.as<uint64_t>(); v.len = d_len.as<uint32_t>(); v.offLen = d_offLen.as<uint64_t>();
        v.n = (uint32_t) n; v.nucl = (dbtype == PLASSHIP_DBTYPE_NUCLEOTIDES); return v;
    }
};

struct plasship_cands {
    bool reverseCapable = false;   // DBTYPE_PREFILTER_REV_RES
    size_t nQueries = 0;           // == query DB size; every query has an entry
    uint64_t nHits = 0;            // including the explicit self hits
    uint64_t nNonSelf = 0;
    plasship::DevBuf d_qoff;       // uint64 [nQueries+1]
    plasship::DevBuf d_hits;       // CandHit [nHits], sorted by (query, target order of the reference)
};

struct plasship_alns {
    size_t nQueries = 0;
    uint64_t nLines = 0;           // ACCEPTED alignments (what plasship_alns_count reports, what a DB file holds)
    // Round 5: a list made by plasship_rescore is SPARSE — one record slot per candidate pair, in the candidate list's own CSR, a rejected
    // pair's record carrying accepted == 0 — because 95 % of the pairs of an assembly iteration are accepted: compacting the list moved
    // 38 GB per iteration (8 ms at 50 M reads) to close 5 % of holes.  Every kernel that walks a query's records skips the holes (the
    // extension kernels' queue fill, arenaSumKernel, findStartVoteKernel, aln2nuclKernel); the host paths (download, DB files) take a
    // dense copy first (denseAlnsCopy, rescore.hip).  Lists read from DB files or made dense are not sparse: nSlots == nLines.
    uint64_t nSlots = 0;           // records in d_recs (holes included)
    bool sparse = false;
    // Round 5, LAZY SELF HITS: every query's alignment with itself (a third of the pairs, and in the late iterations of an assembly — contigs of
    // thousands of residues against themselves — most of the columns rescorediagonal scores: the stage grew 20 -> 60 ms over the twelve
    // iterations at 50 M reads while the candidate pairs fell) is accepted whatever it scores, and assembleresults pops and discards it
    // (assembleresult.cpp:203-209, isNotIdentity), findassemblystart skips it.  plasship_rescore therefore leaves the identity pairs as STUBS
    // (accepted = 1, btKind = ALN_SELF_PENDING, the candidate's score / diagonal stashed in rawScore / qStart) and the first consumer that
    // READS a self record — DB files, downloads, proteinaln2nucl, the nucleotide and guided extension (whose comparator ranks the self hit
    // with the others) — has them scored then, by the same kernels (finishSelfAlns, rescore.hip): every record any reader sees is the
    // record rounds 1-4 made.  selfPending is the list's state, not its value: mutable, set back by finishSelfAlns.
    mutable bool selfPending = false;
    plasship_rescore_params rsPar = {};      // what finishSelfAlns scores with
    bool rsSameDB = false; int rsReverseCapable = 0;
    bool nucl = false;
    bool addBacktrace = false;
    uint64_t dbResidues = 0;       // of the target DB (E-value area)
    int gappedOpen = 0, gappedExtend = 0;   // != 0: E-values of the gapped nucleotide evaluer (list made by plasship_aln2nucl)
    plasship::DevBuf d_qoff;       // uint64 [nQueries+1]
    plasship::DevBuf d_recs;       // AlnRec [nSlots]
    // DBs the list refers to (ids -> keys for text output); they must outlive this object
    const plasship_seqdb *qdb = nullptr, *tdb = nullptr;
};

namespace plasship {
// ---- sharded operation (comm.hip): thin wrappers over the caller's collectives; all of them synchronise the stream first ----
inline const plasship_comm *commOf(const plasship_ctx *ctx) { return ctx->hasComm ? &ctx->comm : nullptr; }
// first id owned by rank r of `world` when n ids are split into contiguous ranges: ceil(r*n/world)
inline uint64_t ownedBegin(uint64_t n, int r, int world) { return ((uint64_t) r * n + (uint64_t) world - 1) / (uint64_t) world; }
int commAllgatherHost(plasship_ctx *ctx, const void *send, void *recv, uint64_t bytesPerRank);
int commAllReduceSumU64(plasship_ctx *ctx, uint64_t *v, size_t n);       // in place
int commAllReduceMaxU64(plasship_ctx *ctx, uint64_t *v, size_t n);
int commAllReduceMinU64(plasship_ctx *ctx, uint64_t *v, size_t n);
int commAgreeOk(plasship_ctx *ctx, bool ok, const char *what);          // collective: error on every rank unless `ok` on every rank
// Failure protocol of a sharded call (comm.hip): every collective starts with an 8-byte status round; a rank that fails on its own
// between two collectives returns to its C-ABI entry, whose commFinish() sends its error code as the status of ONE more round — the
// round the other ranks make before their next collective, or their own commFinish() — and every rank leaves the call with an error
// (PLASSHIP_ERR_PEER on the ranks that did not fail themselves).  Nobody is left waiting inside a collective.
int commFinish(plasship_ctx *ctx, int rc);                              // last statement of every C-ABI entry that can run sharded
// all-to-all(v) of fixed-size records laid out by destination; allocates `recv` (capacity (total + slackRecords) records)
// *allTotal (optional): records sent by all ranks together
int commAlltoallvRecords(plasship_ctx *ctx, const void *dSend, const uint64_t *sendCount, size_t recordBytes, DevBuf &recv,
                         uint64_t *recvTotal, uint64_t slackRecords, uint64_t *allTotal = nullptr);
// all-gather(v) of bytes; allocates `recv`; recvBytes[world] / recvOff[world+1] filled
int commAllgathervBytes(plasship_ctx *ctx, const void *dSend, uint64_t sendBytes, DevBuf &recv, std::vector<uint64_t> &recvBytes);
// the same when every rank already knows everybody's size (recvBytes[world] given)
int commAllgathervBytesKnown(plasship_ctx *ctx, const void *dSend, uint64_t sendBytes, DevBuf &recv, const std::vector<uint64_t> &recvBytes);
// builds an output sequence DB in key order (assemble.hip): entries with flag 0x20 come from `dArena + dNewStart[id]` (dNewLen[id]
// residues), the others are carried over from `db` (dropped when !keepTarget and flag 0x80 is set)
// a copy of `db` with its entries back to back in key order in a buffer of its own (assemble.hip); callers that stream or copy the
// data of a DB as one block (DB files, downloads, concatdbs) take it when !db->contiguous
int packedCopyOf(plasship_ctx *ctx, const plasship_seqdb *db, std::unique_ptr<plasship_seqdb> &out);
int buildOutputDB(plasship_ctx *ctx, const plasship_seqdb *db, const uint32_t *dFlags, const uint32_t *dNewLen, const uint64_t *dNewStart,
                  const char *dArena, int keepTarget, void *dTmp, size_t tmpBytes, plasship_seqdb **out,
                  const void *dExtra = nullptr, void *hExtra = nullptr, size_t extraBytes = 0, hipEvent_t doneEvent = nullptr, bool noAppend = false);
// ---- host boundary (core.hip): bulk copies through the context's pinned double buffer, on the context stream ----
// H2D of `total` bytes the caller produces chunk by chunk: produce(dst, byteOffset, bytes) fills a pinned chunk (consecutive chunks,
// in order; it may use the host threads) while the previous chunk is in flight.  Returns after the last copy has completed.
int stagedUpload(plasship_ctx *ctx, void *dDst, uint64_t total, const std::function<void(char *, uint64_t, uint64_t)> &produce);
// D2H: consume(src, byteOffset, bytes) sees consecutive chunks in order (false aborts with PLASSHIP_ERR_IO); the next chunk is
// already being copied while it runs
int stagedDownload(plasship_ctx *ctx, const void *dSrc, uint64_t total, const std::function<bool(const char *, uint64_t, uint64_t)> &consume);
// 1 MB of pinned host memory owned by the context (nullptr if `bytes` does not fit or it cannot be allocated): a table copied from
// it with hipMemcpyAsync needs no wait before the host moves on.  One user per module call; the next call's use comes after that
// call's own waits, by which time the copy has long completed.
void *ctxPinnedTable(plasship_ctx *ctx, size_t bytes);
// plain arrays: host -> device / device -> host through the staging buffers (memcpy on the host threads)
int stagedCopyToDevice(plasship_ctx *ctx, void *dDst, const void *hSrc, uint64_t bytes);
int stagedCopyToHost(plasship_ctx *ctx, void *hDst, const void *dSrc, uint64_t bytes);
// sets *differ when two key arrays (device, n entries) are not identical
int deviceKeysDiffer(plasship_ctx *ctx, const uint32_t *a, const uint32_t *b, size_t n, bool *differ);
// dense (hole-free) copy of an alignment list's CSR and records on the device (rescore.hip); for a list that is not sparse the buffers
// stay empty and *qoff / *recs point at the list's own arrays
int denseAlnsCopy(plasship_ctx *ctx, const plasship_alns *a, DevBuf &qoffBuf, DevBuf &recsBuf, const uint64_t **qoff, const AlnRec **recs);
// scores the identity pairs plasship_rescore left as stubs (plasship_alns::selfPending); no-op for a list that has none
// (dQueryList: only the stubs of these nQueryList queries — ids on the device)
int finishSelfAlns(plasship_ctx *ctx, const plasship_alns *a, const uint32_t *dQueryList = nullptr, uint32_t nQueryList = 0);
}
