// plasship: proteinaln2nucl on gfx950.  Product code.
//
// Reference behaviour reproduced (mm/util/proteinaln2nucl.cpp:83-188): every protein alignment of the list is moved onto
// the nucleotide twins — coordinates x3 (shifted by one codon behind a leading '*' of the translated ORF, :128-133),
// identities and the nucleotide score recounted over the aligned columns (:141-176), bit score from the GAPPED
// nucleotide evaluer truncated to int (:177), seqId = identities / columns (:180).  The lists this path sees come from
// rescorediagonal --rescore-mode 3 -a 1: the backtrace is one run of 'M' (ungapped); anything else is refused.
// One thread per alignment: 150-450 byte compares against a 15 KB ASCII score table in LDS.
#include "common.hpp"
#include "device_utils.hpp"
#include "host_util.hpp"
#include <algorithm>
#include <memory>
#include <cstring>

namespace plasship {

struct A2NArgs {
    SeqView qn, tn, qa, ta;
    const AlnRec *in; AlnRec *out; uint64_t n;
    const signed char *mat;
    double lambda, logK, ln2;
    uint32_t *err;        // [0] alignment starts on an unalignable '*', [1] gapped backtrace
};

__global__ __launch_bounds__(256) void aln2nuclKernel(A2NArgs a) {
    __shared__ signed char smat[123 * 123 + 7];
    for (int i = threadIdx.x; i < 123 * 123; i += 256) smat[i] = a.mat[i];
    __syncthreads();
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < a.n; i += (uint64_t) gridDim.x * 256) {
        AlnRec r = a.in[i];
        if (!r.accepted) { a.out[i] = r; continue; }                // a hole of a sparse list stays a hole (common.hpp: plasship_alns)
        const char *nq = a.qn.data + a.qn.off[r.query], *nt = a.tn.data + a.tn.off[r.target];
        const bool qStartCodon = a.qa.data[a.qa.off[r.query]] == '*', tStartCodon = a.ta.data[a.ta.off[r.target]] == '*';
        if ((qStartCodon && r.qStart == 0) || (tStartCodon && r.dbStart == 0)) { a.err[0] = 1u; continue; }     // :103-106,123-126
        if (r.btKind != 1) { a.err[1] = 1u; continue; }
        const int cnt = r.alnLen;                                   // the single "<cnt>M" run
        r.dbStart = r.dbStart * 3 + (tStartCodon ? -3 : 0);
        r.dbEnd = r.dbEnd * 3 + 2 + (tStartCodon ? -3 : 0);
        r.dbLen = (int) a.tn.len[r.target];
        r.qStart = r.qStart * 3 + (qStartCodon ? -3 : 0);
        r.qEnd = r.qEnd * 3 + 2 + (qStartCodon ? -3 : 0);
        r.qLen = (int) a.qn.len[r.query];
        const char *q = nq + r.qStart, *t = nt + r.dbStart;
        const int n3 = cnt * 3;
        int ids = 0, score = 0;
        // (round 6: sixteen columns per load and the NEXT sixteen requested before these are scored — the walk was a chain of one dependent round trip
        //  per eight columns, 15 of them for a 40-residue protein alignment)
        uint64_t qn2[2], tn2[2];
        __builtin_memcpy(qn2, q, 16); __builtin_memcpy(tn2, t, 16);                                  // buffers are padded past their ends
        for (int p0 = 0; p0 < n3; p0 += 16) {
            const uint64_t q2[2] = {qn2[0], qn2[1]}, t2[2] = {tn2[0], tn2[1]};
            if (p0 + 16 < n3) { __builtin_memcpy(qn2, q + p0 + 16, 16); __builtin_memcpy(tn2, t + p0 + 16, 16); }
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int p = p0 + 8 * h;
            if (p >= n3) break;
            uint64_t qw = q2[h], tw = t2[h];
            const int m = min(8, n3 - p);
            // columns behind the end are blanked (byte 0 in both words) and their lookups of entry [0][0] taken off again: eight
            // unconditional lookups per step that go out together.  ([0][0] itself is a legitimate entry here: like the reference,
            // the walk runs into the terminator bytes when a protein twin is longer than its ORF / 3.)
            const uint64_t mask = m >= 8 ? ~0ULL : ((1ULL << (8 * m)) - 1ULL);
            const uint64_t lo7 = 0x7F7F7F7F7F7F7F7FULL, x = qw ^ tw;
            ids += __popcll(~(((x & lo7) + lo7) | x | lo7) & mask);                                 // equal bytes among the first m
            qw &= mask; tw &= mask;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned a = (unsigned) (qw >> (8 * j)) & 0xFFu, b = (unsigned) (tw >> (8 * j)) & 0xFFu;
                score += (int) smat[a * 123 + b];
            }
            score -= (8 - m) * (int) smat[0];
          }
        }
        r.rawScore = score;
        r.bitScore = (int) (fma(a.lambda, (double) score, -a.logK) / a.ln2);      // implicit double -> int in the reference: truncation
        r.seqId = (float) ids / (float) n3;
        r.alnLen = n3; r.reversed = 0; r.accepted = 1; r.fromText = 0; r.btKind = 1;
        a.out[i] = r;
    }
}

}  // namespace plasship
using namespace plasship;

extern "C" int plasship_aln2nucl(plasship_ctx *ctx, const plasship_seqdb *q_nucl, const plasship_seqdb *t_nucl, const plasship_seqdb *q_aa,
                                 const plasship_seqdb *t_aa, const plasship_alns *al, const plasship_aln2nucl_params *par, plasship_alns **out,
                                 plasship_aln2nucl_stats *stats) {
    if (!ctx || !q_nucl || !t_nucl || !q_aa || !t_aa || !al || !par || !out) { setError("plasship_aln2nucl: bad argument"); return PLASSHIP_ERR_ARG; }
    if (q_nucl->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES || t_nucl->dbtype != PLASSHIP_DBTYPE_NUCLEOTIDES || q_aa->dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS ||
        t_aa->dbtype != PLASSHIP_DBTYPE_AMINO_ACIDS) { setError("plasship_aln2nucl: Wrong query and target database input"); return PLASSHIP_ERR_ARG; }
    if ((q_nucl == t_nucl) != (q_aa == t_aa)) { setError("plasship_aln2nucl: Either query database == target database for nucleotide and amino acid or != for both"); return PLASSHIP_ERR_ARG; }
    if (al->qdb != q_aa || al->tdb != t_aa) { setError("plasship_aln2nucl: the alignment list does not belong to these protein DBs"); return PLASSHIP_ERR_ARG; }
    if (q_nucl->n != q_aa->n || t_nucl->n != t_aa->n) { setError("plasship_aln2nucl: nucleotide and protein DB differ in size"); return PLASSHIP_ERR_ARG; }
    HostEvaluer ev(true, t_nucl->residues);
    if (!HostEvaluer::nuclGapped(par->gap_open, par->gap_extend, t_nucl->residues, ev)) {
        setError("plasship_aln2nucl: Gumbel parameters exist for --gap-open 5 --gap-extend 2 only (the reference simulates others at start-up)"); return PLASSHIP_ERR_UNSUPPORTED;
    }
    PH_ENTER(ctx);
    hipStream_t st = ctx->stream;
    bool differ = false;
    int rc = deviceKeysDiffer(ctx, q_nucl->d_key.as<uint32_t>(), q_aa->d_key.as<uint32_t>(), q_nucl->n, &differ); if (rc) return rc;
    if (!differ && t_nucl != q_nucl) { rc = deviceKeysDiffer(ctx, t_nucl->d_key.as<uint32_t>(), t_aa->d_key.as<uint32_t>(), t_nucl->n, &differ); if (rc) return rc; }
    if (differ) { setError("plasship_aln2nucl: nucleotide and protein DB have different keys"); return PLASSHIP_ERR_ARG; }
    std::unique_ptr<plasship_alns> holder(new plasship_alns());      // released to the caller on success only
    plasship_alns *o = holder.get();
    o->nQueries = al->nQueries; o->nLines = al->nLines; o->nSlots = al->nSlots; o->sparse = al->sparse; o->nucl = true; o->addBacktrace = true; o->dbResidues = t_nucl->residues;
    o->gappedOpen = par->gap_open; o->gappedExtend = par->gap_extend; o->qdb = q_nucl; o->tdb = t_nucl;
    DevBuf dMat, dErr;
    if (o->d_qoff.alloc((al->nQueries + 1) * 8) != hipSuccess || o->d_recs.alloc(std::max<uint64_t>(al->nSlots, 1) * sizeof(AlnRec)) != hipSuccess ||
        dMat.alloc(123 * 123) != hipSuccess || dErr.alloc(8) != hipSuccess) { setError("plasship_aln2nucl: out of device memory"); return PLASSHIP_ERR_DEVICE; }
    PH_CHECK(hipMemcpyAsync(o->d_qoff.p, al->d_qoff.p, (al->nQueries + 1) * 8, hipMemcpyDeviceToDevice, st));
    PH_CHECK(hipMemcpyAsync(dMat.p, asciiSubMat(true), 123 * 123, hipMemcpyHostToDevice, st));
    PH_CHECK(hipMemsetAsync(dErr.p, 0, 8, st));
    // The walk over 3 * alnLen nucleotide columns runs past the end of an entry when a protein twin is longer than its ORF / 3, like the
    // reference's — into the entry that follows in the DB FILE.  A DB whose rewritten entries live in a shared heap (common.hpp: SeqHeap)
    // is laid out like its file first (round 5: tests/test_gpu_deep.py::test_four_guided_iterations).
    rc = finishSelfAlns(ctx, al); if (rc) return rc;         // every protein alignment is converted, the identity pairs included
    std::unique_ptr<plasship_seqdb> qPacked, tPacked;
    if (!q_nucl->contiguous) { rc = packedCopyOf(ctx, q_nucl, qPacked); if (rc) return rc; }
    if (t_nucl != q_nucl && !t_nucl->contiguous) { rc = packedCopyOf(ctx, t_nucl, tPacked); if (rc) return rc; }
    const plasship_seqdb *qn = qPacked ? qPacked.get() : q_nucl, *tn = (t_nucl == q_nucl) ? qn : (tPacked ? tPacked.get() : t_nucl);
    A2NArgs a; memset(&a, 0, sizeof(a));
    a.qn = qn->view(); a.tn = tn->view(); a.qa = q_aa->view(); a.ta = t_aa->view();
    a.in = al->d_recs.as<AlnRec>(); a.out = o->d_recs.as<AlnRec>(); a.n = al->nSlots; a.mat = dMat.as<signed char>();
    a.lambda = ev.g[0]; a.logK = ev.logK; a.ln2 = ev.ln2; a.err = dErr.as<uint32_t>();
    PH_CHECK(hipEventRecord(ctx->ev[0], st));
    if (al->nSlots) hipLaunchKernelGGL(aln2nuclKernel, dim3((unsigned) std::min<uint64_t>((al->nSlots + 255) / 256, (uint64_t) ctx->numCU * 16)), dim3(256), 0, st, a);
    PH_CHECK(hipEventRecord(ctx->ev[1], st));
    uint32_t herr[2] = {0, 0};
    PH_COPY_SYNC(st, herr, dErr.p, 8, hipMemcpyDeviceToHost);
    PH_CHECK(hipGetLastError());
    if (herr[0]) { setError("plasship_aln2nucl: Alignment contains unalignable character"); return PLASSHIP_ERR_ARG; }
    if (herr[1]) { setError("plasship_aln2nucl: only ungapped alignments (one 'M' run, rescorediagonal -a 1) are supported"); return PLASSHIP_ERR_UNSUPPORTED; }
    if (stats) { stats->n_alignments = al->nLines; float ms = 0; (void) hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); stats->ms_kernel = ms; }
    *out = holder.release();
    return PLASSHIP_OK;
}
