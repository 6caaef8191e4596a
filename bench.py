#!/usr/bin/env python3
"""bench.py — candidate read-overlaps per second per assembly iteration on MI355X (BASELINE.json metric).

A "step" is one assembly iteration of the hot path — kmermatcher -> rescorediagonal -> assembleresults — over
the synthetic protein-fragment DB of BASELINE.json configs[1] (1 M synthetic 2x150 bp protein-coding reads,
500 k pairs, ~1.6 M protein fragments), chained on the device like `plass assemble --num-iterations K`
(iteration 0: --hash-shift 67 --include-only-extendable 0; later: 68,68,69,… and 1; src/workflow/Assembler.cpp:99-110).
The input DB is resident in HBM before the timed region; each of the W warm-up steps is one untimed traversal of
the same K-iteration chain (results discarded), so every iteration's kernels and buffers are warm.  value = sum over the K timed iterations of the candidate overlaps kmermatcher
emitted (non-self prefilter lines) / wall time, max over ranks, summed over ranks.

N > 1 (launched by torch.distributed.run, one rank per GPU), default `--mode sharded`: ONE read set of N x 1 M reads
(N genomes, one seeded part per rank's worth) is sharded over the N GPUs by k-mer bucket — every rank extracts the k-mers
of its share of the sequences, the records travel to the owner of their hash bucket and the grouped records to the owner
of the representative by RCCL all-to-all(v) over xGMI, every rank re-scores and extends the queries it owns and the
extended sequences are all-gathered (include/plasship.h: plasship_ctx_set_comm).  Per-GPU work is
fixed as N grows: weak scaling.  `--mode partitions` is the older independent-partitions run (no data-path collective).

Also reported on the same JSON line:
  roofline     — dominant kernel of the timed run: algorithmic bytes / its HIP-event time
  cpu_baseline — the CPU oracle (a single-threaded port of the reference algorithm, oracle/) timed on a bounded
                 sample of the same workload on this host (rank 0, N=1 only)
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WALL = []               # host wall ms per module call (kmermatcher, rescorediagonal, assembleresults)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def hash_shift(it):
    hs = 67
    for i in range(it + 1):
        hs += i % 2
    return hs


def load_workload(pairs, seed):
    import numpy as np
    from plass_amd import synth
    cache = os.path.join(tempfile.gettempdir(), "plasship_bench_cache")
    os.makedirs(cache, exist_ok=True)
    f = os.path.join(cache, "frag_p%d_s%d.npz" % (pairs, seed))
    if os.path.exists(f):
        try:
            z = np.load(f)
            return z["data"].tobytes(), z["off"], z["elen"], z["key"]
        except Exception:          # a cache file from a run that was killed while writing: regenerate
            pass
    data, off, elen, key = synth.protein_fragment_db(pairs, seed=seed)
    try:
        tmp = os.path.join(cache, "tmp_%d_frag_p%d_s%d.npz" % (os.getpid(), pairs, seed))
        np.savez(tmp, data=np.frombuffer(data, dtype=np.uint8), off=off, elen=elen, key=key)
        os.replace(tmp, f)         # atomic: other ranks / later runs never see a partial file
    except OSError:
        pass
    return data, off, elen, key


def load_sharded_workload(pairs, rank, world, dist):
    """ONE read set of `world` parts (part r = the set a single GPU gets with seed 1 + r).  Rank r generates part r into the
    node-local cache, then everybody loads all parts; keys are made unique by a per-part offset."""
    import numpy as np
    load_workload(pairs, seed=1 + rank)
    if dist is not None:
        dist.barrier()
    datas, offs, elens, keys = [], [], [], []
    base, kbase = 0, 0
    for r in range(world):
        d, o, e, k = load_workload(pairs, seed=1 + r)
        datas.append(d); offs.append(np.asarray(o, dtype=np.uint64) + np.uint64(base)); elens.append(np.asarray(e, dtype=np.uint32))
        keys.append(np.asarray(k, dtype=np.uint32) + np.uint32(kbase))
        base += len(d); kbase += int(np.max(k)) + 1 if len(k) else 0
    return b"".join(datas), np.concatenate(offs), np.concatenate(elens), np.concatenate(keys)


def sharded_preflight(ctx, dist, device):
    """one sharded iteration on a small read set (the same on every rank): every rank must end with the same DB"""
    import torch
    from plass_amd import synth
    data, off, elen, key = synth.protein_fragment_db(3000, seed=7)
    db = ctx.upload_seqdb(data, off, elen, key, 0)
    out, kst, _, _ = one_iteration(ctx, db, 0)
    WALL.pop()
    i = out.info()
    t = torch.tensor([i["residues"], -i["residues"], kst.n_candidates], dtype=torch.int64, device=device)
    mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if int(mx[0]) != -int(mx[1]):
        raise RuntimeError("sharded preflight: the ranks ended with different output DBs")
    out.free(); db.free()


def one_iteration(ctx, db, it):
    import plass_amd
    par = plass_amd.KmermatchParams(k=14, alph_size=13, kmer_per_seq=60, kmer_per_seq_scale=0.0, hash_shift=hash_shift(it),
                                    include_only_extendable=(it > 0), ignore_multi_kmer=True, cov_mode=0, c=0.0)
    t0 = time.perf_counter()
    cands, kst = ctx.kmermatcher(db, par)
    t1 = time.perf_counter()
    alns, rst = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9, e=1e-5))
    t2 = time.perf_counter()
    out, ast = ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9, max_seq_len=65535, keep_target=True))
    t3 = time.perf_counter()
    alns.free(); cands.free()
    WALL.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    return out, kst, rst, ast


def stage_table(kst, rst, ast):
    """per kernel / stage: (HIP-event ms per iteration, algorithmic bytes per SURVEY.md §8d, is_single_kernel, launches per iteration).
    The hash partition (the reference's sort #1, ideal traffic 2*s*N_k = one read + one write of the records) runs as two
    partScatterKernel launches (coarse, fine) plus two histogram launches: the scatter kernel is listed on its own with the
    stage's ideal bytes split over its launches, so the second level counts against the achieved fraction."""
    s = kst.record_bytes
    Nk, Nm, Nc = kst.n_kmer_records, kst.n_grouped, kst.n_candidates
    t = {
        "extractShortKernel": (kst.ms_extract_short_kernel, kst.short_residues + s * kst.short_records, True),
        "extractKernel": (kst.ms_extract_wave_kernel, kst.wave_residues + s * kst.wave_records, True),
        "hash_partition(partHist+partScatter x levels)": (kst.ms_sort1, 2 * s * Nk, False, 1),
        "partScatterKernel": (kst.ms_part_scatter, 2 * s * Nk, True, max(kst.n_part_scatter, 1)),
        "groupKernel": (kst.ms_group, s * Nk + s * Nm, True, 1),
        "rep_sort(partition+aggSortKernel)": (kst.ms_sort2, 2 * s * Nm, False, 1),
        "run_reduce(reduceRunsKernel+CSR)": (kst.ms_reduce, s * Nm + 12 * Nc, False, 1),
        "rescoreKernel": (rst.ms_kernel, 12 * rst.n_scored + 2 * rst.overlap_residues + 32 * rst.n_scored, True, 1),
    }
    t["extractShortKernel"] += (1,); t["extractKernel"] += (1,)
    # whole kmermatcher stage against SURVEY.md section 8d's B_K = R + 4*s*N_k + 4*s*N_m + 12*N_c (every kernel and the host
    # round trips in between count): the number the north star's "achieved HBM bandwidth in kmermatcher" refers to
    t["kmermatcher_stage"] = (kst.ms_extract + kst.ms_sort1 + kst.ms_group + kst.ms_sort2 + kst.ms_reduce,
                              kst.residues + 4 * s * Nk + 4 * s * Nm + 12 * Nc, False, 1)
    for i, (name, single) in enumerate((("assembleGroupKernel<16>", True), ("assembleGroupKernel<32>+<64>", False), ("assembleBigKernel", True))):
        t[name] = (ast.ms_tier_kernel[i], 32 * ast.tier_alignments[i] + 2 * ast.tier_query_residues[i] + 2 * ast.tier_rescored_residues[i], single, 1)
    return t


def cpu_baseline(sample_pairs, iters):
    """the CPU oracle (a port of the reference algorithm; OpenMP over the loops the reference threads: extraction, the two sorts,
    re-scoring, extension) on a bounded sample of the same workload, on the host cores of this box; module compute time only"""
    import __graft_entry__ as g
    from plass_amd import synth
    if not os.path.exists(g.oracle_bin()):
        subprocess.check_call(["make", "-j", "4"], cwd=os.path.join(ROOT, "oracle"))
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    if sample_pairs <= 0:                                    # default: ~10-30 s of CPU work whatever the core count
        sample_pairs = 40000 if threads < 8 else 120000
    data, off, elen, key = synth.protein_fragment_db(sample_pairs, seed=101)
    thr = ["--threads", str(threads)]
    tot_t, tot_c = 0.0, 0
    with tempfile.TemporaryDirectory() as td:
        synth.write_db(os.path.join(td, "seq_0"), data, off, elen, key, 0)
        for it in range(iters):
            s, p, a, o = (os.path.join(td, x) for x in ("seq_%d" % it, "pref", "aln", "seq_%d" % (it + 1)))
            e1 = g.run_oracle(["kmermatcher", s, p, "--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0", "-k", "14", "-c", "0",
                               "--hash-shift", str(hash_shift(it)), "--include-only-extendable", "1" if it else "0", "--ignore-multi-kmer", "1"] + thr)
            e2 = g.run_oracle(["rescorediagonal", s, s, p, a, "--min-seq-id", "0.9", "-e", "1e-5", "-c", "0"] + thr)
            e3 = g.run_oracle(["assembleresults", s, a, o, "--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1"] + thr)
            tot_c += int(re.search(r"N_c=(\d+)", e1).group(1))
            for e in (e1, e2, e3):
                tot_t += float(re.search(r"([0-9.]+) s\s*$", e.strip()).group(1))
    return {"value": tot_c / tot_t, "unit": "overlaps/s", "cores": threads, "kind": "port",
            "sample": "%d read pairs (%d protein fragments), %d iterations, oracle module compute time (no DB I/O), %d OpenMP threads "
                      "(grouping and result writing are single-threaded, as in the reference)" % (sample_pairs, len(key), iters, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=500000, help="read pairs per GPU (500000 = 1 M reads, BASELINE configs[1])")
    ap.add_argument("--parts", type=int, default=1, help="single GPU: build the read set from this many independently seeded parts of --pairs each "
                    "(a 50 M-read set as 10 x 2.5 M pairs keeps the generator's host memory at one part)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = 40000 below 8 host cores, 120000 otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["auto", "sharded", "partitions"], default="auto",
                    help="N > 1: 'sharded' = one read set over the GPUs with RCCL all-to-all (default), 'partitions' = independent sets")
    args = ap.parse_args()

    import torch
    import plass_amd
    from plass_amd import dist as pdist
    rank, local, world = pdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("PLASS_BENCH_FORCE_DIST"):      # one process per GPU over RCCL ("nccl" backend on ROCm); the env
        # switch runs the same collectives in a 1-rank group (what a 1-GPU box can check of the N > 1 path)
        dist = pdist.init("nccl", rank, world, device=torch.device("cuda", local), timeout_s=180)
    plan = pdist.partition_plan(world)
    mode = args.mode
    if mode == "auto":
        mode = "sharded" if dist is not None else "partitions"
    if mode == "sharded" and dist is None:
        raise SystemExit("--mode sharded needs torch.distributed (launch with torch.distributed.run, or PLASS_BENCH_FORCE_DIST=1 on one GPU)")

    ctx = plass_amd.Context(local)
    comm = None
    sharded_error = None
    if mode == "sharded":
        from plass_amd.shard import TorchComm
        comm = TorchComm(dist, torch.device("cuda", local))
        comm.install(ctx)
        try:
            sharded_preflight(ctx, dist, torch.device("cuda", local))
        except Exception as e:       # e.g. a collective this RCCL / torch build lacks: say so and fall back
            if args.mode == "sharded":
                raise
            sharded_error = "%s: %s" % (type(e).__name__, e)
        # every rank takes the same decision (a rank that failed alone must not leave the others in a collective)
        ok = torch.tensor([0 if sharded_error else 1], dtype=torch.int64, device=torch.device("cuda", local))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            sharded_error = sharded_error or "the preflight failed on another rank"
            TorchComm.uninstall(ctx)
            comm, mode = None, "partitions"
    if mode == "sharded":
        data, off, elen, key = load_sharded_workload(args.pairs, rank, world, dist)
    elif args.parts > 1:
        data, off, elen, key = load_sharded_workload(args.pairs, 0, args.parts, None)
    else:
        data, off, elen, key = load_workload(args.pairs, seed=plan["seeds"][rank])
    ctx.sync(); tu0 = time.perf_counter()
    db0 = ctx.upload_seqdb(data, off, elen, key, 0)
    ctx.sync(); upload_s = time.perf_counter() - tu0            # host buffers -> HBM (PCIe), outside the timed region
    n_frag = len(key)

    def barrier():
        ctx.sync(); torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.sync(); torch.cuda.synchronize()

    # warm-up: W untimed traversals of the same K-iteration chain (iteration i of the chain works on the output of
    # iteration i-1, so warming with iteration 0 alone would leave the later iterations' kernels, tiers and buffer sizes cold)
    for _ in range(args.warmup):
        wdb = db0
        for it in range(args.steps):
            out, _, _, _ = one_iteration(ctx, wdb, it)
            if wdb is not db0:
                wdb.free()
            wdb = out
        if wdb is not db0:
            wdb.free()
    barrier()
    if comm is not None:
        comm.bytes_moved = 0; comm.seconds = 0.0; comm.calls = 0
    t0 = time.perf_counter()
    db = db0
    stats = []
    overlaps = 0
    for it in range(args.steps):
        out, kst, rst, ast = one_iteration(ctx, db, it)
        overlaps += kst.n_candidates
        stats.append(stage_table(kst, rst, ast))
        if os.environ.get("PLASS_BENCH_VERBOSE") and rank == 0:     # per-iteration counters, to stderr
            print("it%d kmer: Nk=%d Nm=%d Nc=%d | rescore: scored=%d accepted=%d ov=%d | assemble: aln=%d ext=%d resc=%d rescRes=%d tiers ms=%s aln=%s qres=%s rres=%s | wall %s" % (
                it, kst.n_kmer_records, kst.n_grouped, kst.n_candidates, rst.n_scored, rst.n_accepted, rst.overlap_residues,
                ast.n_alignments, ast.n_extended, ast.n_rescored, ast.rescored_residues, ["%.2f" % x for x in ast.ms_tier_kernel],
                list(ast.tier_alignments), list(ast.tier_query_residues), list(ast.tier_rescored_residues), ["%.2f" % x for x in WALL[-1]]), file=sys.stderr)
        if db is not db0:
            db.free()
        db = out
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    td0 = time.perf_counter()
    final_bytes = len(db.download()[0]) if args.steps else 0     # final contig DB back to host buffers (PCIe), outside the timed region
    download_s = time.perf_counter() - td0
    local_overlaps = overlaps
    elapsed, overlaps = pdist.reduce_step(dist, elapsed, overlaps, device="cuda")

    if rank == 0:
        # dominant kernel over the timed iterations (rank 0's HIP-event times)
        tot = {}
        for st in stats:
            for k, (ms, b, single, launches) in st.items():
                a = tot.setdefault(k, [0.0, 0, single, 0])
                a[0] += ms; a[1] += b; a[3] += launches
        # the kernel with the largest total time over the timed iterations; its numbers are per launch
        dom = max((k for k in tot if tot[k][2]), key=lambda k: tot[k][0])
        ms_avg = tot[dom][0] / max(tot[dom][3], 1)
        bytes_avg = tot[dom][1] / max(tot[dom][3], 1)
        achieved = bytes_avg / (ms_avg * 1e-3) / 1e9 if ms_avg > 0 else 0.0
        line = {
            "metric": "read-overlaps/s per assembly iteration", "value": overlaps / elapsed, "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64 (integer hash, byte compare; f32 ratios)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic 2x150 bp protein-coding reads per GPU (%d read pairs, %d protein fragments), "
                                   "--num-iterations %d, k=14, alph 13, kmer-per-seq 60, min-seq-id 0.9, e 1e-5" % (
                                       2 * args.pairs * (args.parts if mode != "sharded" else 1), args.pairs * (args.parts if mode != "sharded" else 1), n_frag, args.steps),
                       "parallelism": ("1 GPU" if world == 1 and mode != "sharded" else
                                       "%d GPUs, one read set of %d fragments sharded by k-mer bucket (RCCL all-to-all(v) of k-mer and grouped records, "
                                       "all-gather of extended sequences), sequence DB replicated" % (world, n_frag) if mode == "sharded" else
                                       "1 process per GPU, independent partitions"),
                       "candidate_overlaps": overlaps},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "ms_per_launch": ms_avg, "algorithmic_bytes_per_launch": bytes_avg, "launches_per_step": tot[dom][3] / len(stats),
                         "stage_ms_per_step": {k: v[0] / len(stats) for k, v in tot.items()},
                         "kmermatcher_stage": {"algorithmic_bytes_per_step": tot["kmermatcher_stage"][1] / len(stats),
                                               "ms_per_step": tot["kmermatcher_stage"][0] / len(stats),
                                               "frac": (tot["kmermatcher_stage"][1] / (tot["kmermatcher_stage"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tot["kmermatcher_stage"][0] > 0 else 0.0},
                         "module_wall_ms_per_step": [round(sum(w[i] for w in WALL[-args.steps:]) / args.steps, 3) for i in range(3)]},
        }
        # what the job costs a caller that hands over host buffers and wants host buffers back (never `value`): rank 0's share
        line["host_boundary"] = {"upload_ms": upload_s * 1e3, "upload_bytes": len(data), "download_ms": download_s * 1e3, "download_bytes": final_bytes,
                                 "pcie_inclusive_overlaps_per_s_rank0": local_overlaps / (t1 - t0 + upload_s + download_s)}
        if sharded_error is not None:
            line["sharded_mode_error"] = sharded_error
        if comm is not None:
            line["exchange"] = {"device_bytes_sent_per_step_rank0": comm.bytes_moved / max(args.steps, 1), "collective_calls_per_step": comm.calls / max(args.steps, 1),
                                "ms_in_collectives_per_step_rank0": comm.seconds * 1e3 / max(args.steps, 1)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_pairs, min(args.steps, 3))
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
