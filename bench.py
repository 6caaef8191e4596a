#!/usr/bin/env python3
"""bench.py — candidate read-overlaps per second per assembly iteration on MI355X (BASELINE.json metric).

A "step" is ONE assembly iteration of the hot path — kmermatcher -> rescorediagonal -> assembleresults — chained on the
device like `plass assemble` chains them through DBs (iteration 0: --hash-shift 67 --include-only-extendable 0; later: 68, 68,
69, … and 1; src/workflow/Assembler.cpp:99-110).  The workload is a CONFIG of BASELINE.json and brings its own iteration
count (the chain length); --steps K times K iterations: iteration (s mod chain) of the chain, restarting from the input DB
when the chain is through — so K never changes what an iteration works on.  --warmup W runs the first W iterations of the
chain untimed.

  default (no flags), N = 1: BASELINE configs[2] — 50 M synthetic 2x150 bp metagenomic reads (25 M pairs from a community of
      200 genomes of 1-5 Mbp with log-normal abundances, sigma 1, seed 2), default parameters (12 iterations, k = 14,
      alphabet 13, 60 k-mers per sequence, min-seq-id 0.9, e 1e-5).  --config c2 = configs[1] (1 M reads, one 7.5 Mbp
      genome, seed 1, 6 iterations).  --pairs P scales the community with the reads (same coverage).
  N > 1 (torch.distributed.run, one rank per GPU): configs[3] — the SAME read set, sharded by k-mer bucket over the N GPUs
      (RCCL all-to-all(v) of k-mer and grouped records, all-gather of extended sequences; include/plasship.h:
      plasship_ctx_set_comm).  Total work is fixed as N grows: strong scaling.

The reads are generated in HBM (include/plasship_synth.h) and turned into the protein fragment DB by the GPU versions of the
workflow's own preprocessing (extractorfs x2, translatenucs x2, concatdbs: data/assemble.sh:41-77) before the timed region;
every rank generates the identical DB from the seed.  value = candidate overlaps kmermatcher emitted in the K timed
iterations (non-self prefilter lines) / wall time (max over ranks), inputs resident in HBM.

Also on the JSON line:
  iterations   — one row per timed step: chain iteration, ms, N_k / N_m / N_c, stage times
  roofline     — dominant kernel of the timed run: algorithmic bytes / its HIP-event time, per launch;
                 `traffic` = HBM bytes per launch of that kernel from the stored rocprofv3 PMC pass of the same command
                 (PMC_FILES below: the newest profiles/rNN_pmc_traffic.json taken from these sources), null if that file has no row for it
  cpu_baseline — the CPU oracle (a port of the reference's algorithm, oracle/) on a bounded sample of the same community
                 model on this host's cores (rank 0, N = 1), plus the I/O-inclusive rate of the drop-in command line
                 (plass-hip: DB files in, DB files out) on the same sample
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

VERBOSE = bool(os.environ.get("PLASS_BENCH_VERBOSE"))
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # name: (BASELINE.json index, read pairs, genomes, min/max genome length, abundance sigma, seed, chain length)
    "c3": (2, 25000000, 200, 1000000, 5000000, 1.0, 2, 12),
    "c2": (1, 500000, 1, 7500000, 7500000, 0.0, 1, 6),
    # configs[4]: PenguiN's nucleotide-level chains (kmermatcher k=22 -> rescorediagonal -> nuclassembleresults -> cyclecheck, and the
    # protein-guided chain) on 20 M reads of the same community model, seed 3; chain = nucleotide iterations timed
    "c5": (4, 10000000, 200, 1000000, 5000000, 1.0, 3, 5),
}


def hash_shift(it):
    hs = 67
    for i in range(it + 1):
        hs += i % 2
    return hs


def synth_params(cfg, pairs=None, min_genomes=0):
    """SynthParams of a config; another number of pairs keeps the coverage (the community shrinks / grows with the reads).
    min_genomes: a small sample of a community config keeps at least that many genomes (shorter ones, same total bases), so that
    its coverage stays SKEWED like the full workload's (the CPU baseline's sample)"""
    import plass_amd
    _, p0, g, lo, hi, sigma, seed, _ = CONFIGS[cfg]
    pairs = p0 if not pairs else pairs
    if pairs != p0:
        if g > 1:
            bases = (lo + hi) / 2.0 * g * pairs / p0                  # total genome bases at the config's coverage
            g = max(1, int(round(g * pairs / p0)))
            if g < min_genomes:
                g = min_genomes
                lo, hi = max(20000, int(0.5 * bases / g)), max(30000, int(1.5 * bases / g))
            elif g == 1:
                lo = hi = max(30000, int(3000000 * 200 * pairs / p0))
        else:
            lo = hi = max(30000, int(lo * pairs / p0))
    return plass_amd.SynthParams(n_pairs=pairs, seed=seed, n_genomes=g, genome_min_len=lo, genome_max_len=hi, abundance_sigma=sigma)


def build_workload(ctx, cfg, pairs=None, min_genomes=0):
    """reads in HBM -> protein fragment DB (the DB iteration 0 starts from); returns (db, description dict)"""
    t0 = time.perf_counter()
    sp = synth_params(cfg, pairs, min_genomes)
    reads, sst = ctx.synth_read_pairs(sp)
    t1 = time.perf_counter()
    frag = ctx.plass_fragments(reads)
    ctx.sync()
    t2 = time.perf_counter()
    ri = reads.info(); fi = frag.info()
    reads.free()
    return frag, {"read_pairs": sp.n_pairs, "reads": ri["n"], "genomes": sp.n_genomes, "genome_bases": int(sst.genome_bases), "genes": int(sst.n_genes),
                  "mean_coverage": round(sst.mean_coverage, 2), "max_genome_coverage": round(sst.max_coverage, 1), "seed": sp.seed,
                  "protein_fragments": fi["n"], "fragment_residues": fi["residues"],
                  "generate_reads_s": round(t1 - t0, 3), "extractorfs_translatenucs_concatdbs_s": round(t2 - t1, 3)}


def one_iteration(ctx, db, it, xstat=None):
    """xstat (sharded runs): callable returning the communicator's cumulative (bytes sent, seconds, calls) — sampled around every module so
    that the N > 1 line can say which module's exchanges cost what"""
    import plass_amd
    par = plass_amd.KmermatchParams(k=14, alph_size=13, kmer_per_seq=60, kmer_per_seq_scale=0.0, hash_shift=hash_shift(it),
                                    include_only_extendable=(it > 0), ignore_multi_kmer=True, cov_mode=0, c=0.0)
    xs = (lambda: tuple(xstat())) if xstat else (lambda: (0, 0.0, 0))
    t0 = time.perf_counter(); s0 = ctx.host_syncs(); x0 = xs()
    cands, kst = ctx.kmermatcher(db, par)
    t1 = time.perf_counter(); s1 = ctx.host_syncs(); x1 = xs()
    alns, rst = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9, e=1e-5))
    t2 = time.perf_counter(); s2 = ctx.host_syncs(); x2 = xs()
    out, ast = ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9, max_seq_len=65535, keep_target=True))
    t3 = time.perf_counter(); s3 = ctx.host_syncs(); x3 = xs()
    alns.free(); cands.free()
    xd = tuple(tuple(b[i] - a[i] for i in range(3)) for a, b in ((x0, x1), (x1, x2), (x2, x3)))
    # wall: ms per module, then the number of times the host waited for the stream inside each module, then (sharded runs) the
    # communicator's (bytes sent, seconds, calls) inside each module
    return out, kst, rst, ast, ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, s1 - s0, s2 - s1, s3 - s2, xd)


def stage_table(kst, rst, ast, nucl_queue=False):
    """per kernel / stage: (HIP-event ms, algorithmic bytes per SURVEY.md section 8d, is_single_kernel, launches).
    The hash partition (the reference's sort #1, ideal traffic 2*s*N_k = one read + one write of the records) runs as several
    partition-kernel launches: the kernel is listed on its own with the stage's ideal bytes split over its launches, so every
    extra pass counts against the achieved fraction."""
    s = kst.record_bytes
    Nk, Nm, Nc = kst.n_kmer_records, kst.n_grouped, kst.n_candidates
    t = {
        "extractShortKernel": (kst.ms_extract_short_kernel, kst.short_residues + s * kst.short_records, True, 1),
        "extractKernel": (kst.ms_extract_wave_kernel, kst.wave_residues + s * kst.wave_records, True, 1),
        "hash_partition(all passes)": (kst.ms_sort1, 2 * s * Nk, False, 1),
        "partitionKernel(k-mer records)": (kst.ms_part_scatter, 2 * s * Nk, True, max(kst.n_part_scatter, 1)),
        "groupKernel": (kst.ms_group, s * Nk + s * Nm, True, 1),
        "rep_sort(partition+aggSortKernel)": (kst.ms_sort2, 2 * s * Nm, False, 1),
        "run_reduce(reduceRunsKernel+CSR)": (kst.ms_reduce, s * Nm + 12 * Nc, False, 1),
        "rescoreKernel": (rst.ms_kernel, 12 * rst.n_scored + 2 * rst.overlap_residues + 32 * rst.n_scored, True, 1),
    }
    # whole kmermatcher stage against SURVEY.md section 8d's B_K = R + 4*s*N_k + 4*s*N_m + 12*N_c (every kernel and the host
    # round trips in between count): the number the north star's "achieved HBM bandwidth in kmermatcher" refers to
    t["kmermatcher_stage"] = (kst.ms_extract + kst.ms_sort1 + kst.ms_group + kst.ms_sort2 + kst.ms_reduce,
                              kst.residues + 4 * s * Nk + 4 * s * Nm + 12 * Nc, False, 1)
    if nucl_queue:
        # nuclassembleresults / guidedassembleresults: the heap-replay queue kernels (one lane per queue up to 256 hits — copies and
        # re-scored overlaps by the whole wavefront —, one wavefront per query beyond, re-runs of the queries that met an unknown
        # comparator tuple) are timed as ONE interval (assemble.hip)
        t["assembleNuclKernel(+assembleNuclThreadKernel, all passes)"] = (ast.ms_tier_kernel[0], 32 * ast.tier_alignments[0] + 2 * ast.tier_query_residues[0] + 2 * ast.tier_rescored_residues[0], True, 1)
    else:
        for i, (name, single) in enumerate((("assembleGroupKernel<16>", True), ("assembleGroupKernel<32>+<64>", False), ("assembleBigKernel", True))):
            t[name] = (ast.ms_tier_kernel[i], 32 * ast.tier_alignments[i] + 2 * ast.tier_query_residues[i] + 2 * ast.tier_rescored_residues[i], single, 1)
    # the other two modules as whole stages, against SURVEY.md section 8d's B_R = (12 + 2 ov + 32) N_c and B_A = 32 N_aln + 2 R + 2 ov N_resc
    # (every kernel of the module: work lists, compaction, the extension tiers, writing the next DB)
    t["rescore_stage"] = (rst.ms_kernel, 12 * rst.n_scored + 2 * rst.overlap_residues + 32 * rst.n_scored, False, 1)
    t["assemble_stage"] = (ast.ms_kernel, 32 * ast.n_alignments + 2 * kst.residues + 2 * ast.rescored_residues, False, 1)
    return t


def source_sha():
    """sha256 over the product sources the library is built from (plass_amd/csrc, include/): a stored profile is only quoted for
    the code it was taken from (there is no .git on the GPU box to ask)"""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "plass_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".hpp", ".cpp", ".h")):
                h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# rocprofv3 names of the kernels behind a row of the stage table (all instantiations of a template are one row)
KERNEL_SYMBOLS = {"extractKernel": "(extractKernel<|extractRowKernel<|binWaveListKernel)", "extractShortKernel": "extractShort(Fast)?Kernel<", "groupKernel": "group(Lines)?Kernel<", "rescoreKernel": "rescoreKernel<",
                  # (MODE 0 = the hash partition of the k-mer records; the MODE 1 instantiations are the rep sort's range partitions: VERDICT r4 weak #10)
                  "partitionKernel(k-mer records)": "linePartKernel<(true|false), (true|false), 0, .*\\(plasship::LinePartArgs\\)", "assembleGroupKernel<16>": "assembleGroupKernel<16", "assembleBigKernel": "assembleBigKernel",
                  "assembleNuclKernel(+assembleNuclThreadKernel, all passes)": "assembleNucl(Thread)?Kernel<"}
ROUND = "r06"                          # prefix of the profiles/ files this round's evidence run writes (tools/gpu_round6_final.sh)
KERNEL_STATS_FILE = "profiles/%s_kernel_stats_driver_cmd.txt" % ROUND
# stored PMC passes, newest first: the first one whose recorded source hash equals this build's is quoted (stored_traffic)
PMC_FILES = {"c3": ["%s_pmc_traffic.json" % ROUND, "r05_pmc_traffic.json"], "c5": ["%s_pmc_traffic_c5.json" % ROUND, "r05_pmc_traffic_c5.json"]}


def stored_traffic(kernel, launches_per_step, cfg="c3"):
    """(HBM bytes per launch of `kernel` — all its instantiations together — from the stored PMC passes of the driver's command, note).
    profiles/rNN_pmc_traffic[_c5].json (PMC_FILES): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; the file records the hash of the sources it was measured on and is refused for any
    other code.  A stored figure, not measured in this run: PMC collection serialises the kernels."""
    names = PMC_FILES.get(cfg)
    if not names:
        return None, "no PMC pass is stored for --config %s" % cfg
    rows, why = None, "no stored PMC profile (profiles/%s)" % names[0]
    for name in names:
        try:
            cand = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if cand.get("source_sha") == source_sha():
            rows = cand
            break
        why = "profiles/%s was taken from other sources (%s, this build %s): not quoted" % (name, cand.get("source_sha"), source_sha())
    if rows is None:
        return None, why
    pat = KERNEL_SYMBOLS.get(kernel, re.escape(re.sub(r"[<(].*", "", kernel)) + "[<(]")
    tot = sum(r["hbm_bytes_per_launch"] * r["launches"] for name, r in rows.get("kernels", {}).items() if re.search(pat, name))
    steps = rows.get("steps", 0)
    return (tot / steps / max(launches_per_step, 1e-9), rows.get("source", "")) if tot and steps else (None, "the stored profile has no row for " + kernel)


# the cost model of a sharded iteration, so that a SCALE record can be read against it phase by phase.
# Inputs measured on ONE MI355X at 50 M reads (profiles/r06_bench_driver_cmd.log): ms per iteration of the single-GPU path by module, the part
# of kmermatcher that is NOT sharded when every rank extracts all sequences (owner-filtered extraction, the default up to 4 ranks), the
# 1-rank overhead of the sharded orchestration (12.5 M reads: the owner's merge of the exchanged triples, packing the extended sequences);
# link: one xGMI link per GPU pair, 76 GB/s per direction assumed.
# (round 6: the module times of profiles/r06_bench_driver_cmd.log — kmermatcher 193.6 of which extraction 63.5, rescorediagonal 28.6, assembleresults 75.0)
MODEL_50M = {"kmermatcher_ms": 193.6, "extraction_ms": 63.5, "rescore_ms": 28.6, "assemble_ms": 75.0, "other_ms": 0.0,
             "shard_overhead": 0.10, "level1_line_bytes": 63e9, "triple_bytes": 5e9, "extended_bytes": 3.5e9, "link_GBs": 76.0, "host_rounds_ms": 2.0}


def scaling_model(world, reads, measured_module_wall=None):
    """predicted ms per iteration of the sharded run at `world` ranks for both ways the k-mer records reach their owner, scaled linearly
    with the reads from the 50 M-read figures; `measured_module_wall_ms` = this run's [kmermatcher, rescorediagonal, assembleresults]"""
    m = MODEL_50M
    f = reads / 50e6
    W = max(world, 1)
    ov = 1.0 + m["shard_overhead"]
    t2 = (m["triple_bytes"] + m["extended_bytes"]) * f / (W * W * m["link_GBs"] * 1e9) * 1e3 * (W - 1) if W > 1 else 0.0
    filt_km = m["extraction_ms"] * f + (m["kmermatcher_ms"] - m["extraction_ms"]) * f * ov / W
    exch_km = m["kmermatcher_ms"] * f * ov / W + (m["level1_line_bytes"] * f / (W * W * m["link_GBs"] * 1e9) * 1e3 if W > 1 else 0.0)
    rest = (m["rescore_ms"] + m["assemble_ms"] * ov) * f / W + t2 + (m["host_rounds_ms"] if W > 1 else 0.0)
    single = (m["kmermatcher_ms"] + m["rescore_ms"] + m["assemble_ms"]) * f
    out = {"inputs": m, "single_gpu_ms": round(single, 1),
           "owner_filtered": {"kmermatcher_ms": round(filt_km, 1), "total_ms": round(filt_km + rest, 1), "speedup": round(single / (filt_km + rest), 2)},
           "exchange": {"kmermatcher_ms": round(exch_km, 1), "total_ms": round(exch_km + rest, 1), "speedup": round(single / (exch_km + rest), 2)},
           "library_default": "owner_filtered" if W <= 4 else "exchange",
           "note": "kmermatch.hip shardOwnerFiltered: owner-filtered extraction up to 4 ranks (nothing crosses the links for the k-mer records; extraction is "
                   "replicated), the all-to-all of level-1 lines beyond (PLASSHIP_TUNE_SHARD_EXTRACT=1 / 2 forces one)"}
    if measured_module_wall:
        out["measured_module_wall_ms"] = [round(x, 1) for x in measured_module_wall]
    return out


def furthest_below(tot, n_steps, cfg):
    """the single kernel of the stage table that sits furthest below the HBM roofline by ALGORITHMIC bytes (kernels below 2 % of the
    step's kernel time are left out), with its stored PMC traffic over its algorithmic bytes — so that the worst kernel is on the line
    every round"""
    whole = sum(v[0] for k, v in tot.items() if k in ("kmermatcher_stage", "rescore_stage", "assemble_stage"))
    worst = None
    for k, (ms, b, single, launches) in tot.items():
        if not single or ms <= 0.02 * whole or b <= 0:
            continue
        frac = b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if worst is None or frac < worst["frac"]:
            traffic, _ = stored_traffic(k, launches / n_steps, cfg)
            worst = {"kernel": k, "frac": frac, "ms_per_step": ms / n_steps, "algorithmic_bytes_per_step": b / n_steps,
                     "traffic_ratio": (traffic * launches / b) if traffic else None}
    return worst


# The unmodified reference on the WHOLE bench workloads, measured in the build container (8 threads of a Xeon at 2.1 GHz; it cannot travel to the
# GPU box): seconds inside its kmermatcher / rescorediagonal / assembleresults (c3: twelve iterations on 50 M reads, kmermatcher unsplit) and inside
# the ten PenguiN steps of c5 (20 M reads) — the runs whose results are the digests `verify` compares with (profiles/r05_headline_pin_reference.txt).
REFERENCE_FULL_WORKLOAD = {
    "c3": {"seconds": 8420.0, "seconds_by_module": {"kmermatcher": 3879.0, "rescorediagonal": 2596.0, "assembleresults": 1945.0}, "threads": 8, "cpu": "Xeon 2.1 GHz (build container)",
           "candidate_overlaps": 2.40e9, "overlaps_per_s": 2.40e9 / 8420.0, "iterations": 12, "reads": 50000000, "source": "profiles/r05_headline_pin_reference.txt",
           "note": "the unmodified `plass` binary, AVX2, 8 OpenMP threads, kmermatcher in one part (--split-memory-limit 64G); its seq_1 .. seq_12 are tests/golden/c3_chain_digests.json"},
    "c5": {"seconds": 4126.0, "threads": 8, "cpu": "Xeon 2.1 GHz (build container)", "overlaps_per_s": 0.16e6, "iterations": 10, "reads": 20000000,
           "source": "profiles/r05_headline_pin_reference.txt", "note": "the unmodified `penguin` binary: 5 guided iterations (2 156 s) + 5 nucleotide iterations with cyclecheck (1 970 s)"},
    # round 6, 6 threads: the reference-only fixture tests/golden/reference_chain.json (5 M reads, twelve iterations; profiles/r06_record_chain_reference.txt)
    "c3_5M_reads": {"seconds": 746.6, "seconds_by_module": {"kmermatcher": 454.5, "rescorediagonal": 141.0, "assembleresults": 151.1}, "threads": 6, "cpu": "Xeon 2.1 GHz (build container)",
                    "iterations": 12, "reads": 5000000, "source": "profiles/r06_record_chain_reference.txt"},
}


def host_cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ctx, cfg, sample_pairs, iters):
    """(1) the CPU oracle — a port of the reference algorithm with OpenMP over the loops the reference threads — on a bounded
    sample of the same community model (same generator, same coverage), module compute time only;
    (2) the drop-in command line on the same DB files: plass-hip kmermatcher / rescorediagonal / assembleresults, DB files in
    and out, its own "Time for processing" summed = the I/O-inclusive rate of the GPU path through the module boundary."""
    import __graft_entry__ as g
    if not os.path.exists(g.oracle_bin()):
        subprocess.check_call(["make", "-j", "4"], cwd=os.path.join(ROOT, "oracle"))
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    if sample_pairs <= 0:                                    # ~10-30 s of CPU work whatever the core count (the GPU box has 64+ cores: 1 M pairs there — VERDICT r5 item 8)
        sample_pairs = 40000 if threads < 8 else (120000 if threads < 32 else 1000000)
    db, desc = build_workload(ctx, cfg, sample_pairs, min_genomes=5)      # >= 5 genomes with log-normal abundances: skewed like the GPU workload
    thr = ["--threads", str(threads)]
    tot_t, tot_c, cli_t, cli_c = 0.0, 0, 0.0, 0
    hip = os.path.join(ROOT, "plass_amd", "plass-hip")
    with tempfile.TemporaryDirectory() as td:
        db.write(os.path.join(td, "seq_0")); db.free()
        for it in range(iters):
            s, p, a, o = (os.path.join(td, x) for x in ("seq_%d" % it, "pref", "aln", "seq_%d" % (it + 1)))
            km = ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0", "-k", "14", "-c", "0", "--hash-shift", str(hash_shift(it)),
                  "--include-only-extendable", "1" if it else "0", "--ignore-multi-kmer", "1"]
            rs = ["--rescore-mode", "3", "--min-seq-id", "0.9", "-e", "1e-5", "-c", "0"]
            asm = ["--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1", "--rescore-mode", "3"]
            if os.path.exists(hip) and it < 3:               # the GPU path through the module boundary, on the same files (first three iterations: nine process starts)
                for args in (["kmermatcher", s, p + "_g"] + km, ["rescorediagonal", s, s, p + "_g", a + "_g"] + rs, ["assembleresults", s, a + "_g", o + "_g"] + asm):
                    out = subprocess.run([hip] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=g.child_env())
                    m = re.search(r"Time for processing: ([0-9.]+)s", out.stdout)
                    if out.returncode != 0 or not m:
                        raise RuntimeError("plass-hip %s failed: %s" % (args[0], out.stdout[-500:]))
                    cli_t += float(m.group(1))
                    m = re.search(r"candidates: (\d+)", out.stdout)
                    if m:
                        cli_c += int(m.group(1))
            e1 = g.run_oracle(["kmermatcher", s, p] + km + thr)
            e2 = g.run_oracle(["rescorediagonal", s, s, p, a] + rs[2:] + thr)
            e3 = g.run_oracle(["assembleresults", s, a, o] + asm[:6] + thr)
            tot_c += int(re.search(r"N_c=(\d+)", e1).group(1))
            for e in (e1, e2, e3):                           # "oracle <module>: …, 1.234 s" (other lines may follow: a profiler attached to the child)
                tot_t += float(re.search(r"^oracle \w+:.*?([0-9.]+) s\s*$", e, re.M).group(1))
    res = {"value": tot_c / tot_t, "unit": "overlaps/s", "cores": threads, "kind": "port", "cpu": host_cpu_model(), "seconds": round(tot_t, 2),
           "reference_full_workload": REFERENCE_FULL_WORKLOAD.get(cfg),
           "sample": "%d read pairs of the same community model at the same mean coverage, skewed (%d genomes, log-normal abundances sigma 1; %d protein fragments), iterations 0..%d of the chain, "
                     "oracle module compute time (no DB I/O), %d OpenMP threads (grouping and result writing are single-threaded, as in the reference)"
                     % (desc["read_pairs"], desc["genomes"], desc["protein_fragments"], iters - 1, threads),
           # the unmodified reference (AVX2, 8 threads) against this port in the build container
           "reference_vs_port_build_container": {"reference_overlaps_per_s": 0.40e6, "port_overlaps_per_s": 0.30e6, "threads": 8,
                                                 "note": "reference source cannot travel to the GPU box; scale `value` by 0.40/0.30 for the AVX2 reference"}}
    if cli_t > 0:
        res["drop_in_cli_same_sample"] = {"value": cli_c / cli_t, "unit": "overlaps/s", "seconds": round(cli_t, 3),
                                          "what": "plass-hip kmermatcher + rescorediagonal + assembleresults, DB files in / DB files out, sum of `Time for processing`, iterations 0..%d of the sample" % (min(iters, 3) - 1)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed iterations (default: one traversal of the config's chain)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed iterations first (default: one traversal of the chain)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="c3 = BASELINE configs[2]/[3] (50 M reads, default); c2 = configs[1] (1 M reads); c5 = configs[4] (PenguiN chains, 20 M reads, one GPU)")
    ap.add_argument("--pairs", type=int, default=0, help="read pairs of the whole job (0 = the config's own; the community scales with it)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = 40000 below 8 host cores, 120000 otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wall", action="store_true", help="skip the wall-clock-to-contigs run of the fused C++ driver (DB files on disk -> final DB on disk)")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed traversal that digests every iteration's output DB")
    ap.add_argument("--comm", choices=["auto", "native", "torch"], default="auto",
                    help="sharded run: 'native' = the library's own RCCL communicator (include/plasship_rccl.h), 'torch' = torch.distributed P2P; auto = native, torch if that fails")
    ap.add_argument("--mode", choices=["auto", "sharded", "partitions"], default="auto",
                    help="N > 1: 'sharded' = the one read set over the GPUs with RCCL all-to-all (default); 'partitions' = N independent sets of 1/N the size")
    args = ap.parse_args()
    if args.config == "c5":
        return main_c5(args)

    import torch
    import plass_amd
    from plass_amd import dist as pdist
    rank, local, world = pdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    cfg_idx, cfg_pairs, _, _, _, _, _, chain = CONFIGS[args.config]
    pairs = args.pairs or cfg_pairs
    steps = chain if args.steps is None else args.steps
    warmup = chain if args.warmup is None else args.warmup
    dist = None
    if world > 1 or os.environ.get("PLASS_BENCH_FORCE_DIST"):      # one process per GPU over RCCL ("nccl" backend on ROCm); the env
        # switch runs the same collectives in a 1-rank group (what a 1-GPU box can check of the N > 1 path)
        dist = pdist.init("nccl", rank, world, device=torch.device("cuda", local), timeout_s=300)
    mode = args.mode
    if mode == "auto":
        mode = "sharded" if dist is not None else "single"
    if mode == "sharded" and dist is None:
        raise SystemExit("--mode sharded needs torch.distributed (launch with torch.distributed.run, or PLASS_BENCH_FORCE_DIST=1 on one GPU)")

    ctx = plass_amd.Context(local)
    dev = torch.device("cuda", local)
    comm = None; native = None
    sharded_error = None
    comm_timeout = float(os.environ.get("PLASS_BENCH_COMM_TIMEOUT", "240"))
    setup_s = {}
    # the preprocessing is not sharded: every rank builds the identical DB from the seed (before a communicator is installed)
    db0, wl = build_workload(ctx, args.config, pairs if mode != "partitions" else max(1000, pairs // world))
    if mode == "sharded":
        from plass_amd.shard import TorchComm, RcclComm, rccl_unique_id
        comm_error = None
        if args.comm in ("auto", "native"):
            try:                     # rank 0 makes the id, torch.distributed carries it (control plane only)
                box = [rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, device=dev)
                native = bounded(lambda: RcclComm(ctx, rank, world, box[0]), comm_timeout, "creating the native RCCL communicator (ncclCommInitRank)", rank)
            except Exception as e:
                comm_error = "%s: %s" % (type(e).__name__, e)
            ok = torch.tensor([0 if comm_error else 1], dtype=torch.int64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if native is not None:
                    native.destroy(); native = None
                if args.comm == "native":
                    raise SystemExit("native RCCL communicator failed: %s" % (comm_error or "on another rank"))
        if native is None:
            comm = TorchComm(dist, dev)
            comm.install(ctx)
        try:
            t_pre = time.perf_counter()
            bounded(lambda: sharded_preflight(ctx, dist, dev), comm_timeout, "the sharded preflight iteration (first exchanges over the links, %s communicator)" % ("native RCCL" if native is not None else "torch.distributed"), rank)
            setup_s["preflight_s"] = round(time.perf_counter() - t_pre, 3)      # (the links' first use, RCCL's channel set-up: outside the timed steps)
        except TimeoutError as e:
            # a collective that never completes leaves this context's stream blocked: nothing can be salvaged in this process.  Say which rank and
            # which phase, and end the job with an error instead of hanging until somebody's outer limit (the peers time out in the same phase)
            print(json.dumps({"error": str(e), "rank": rank, "world": world}), file=sys.stderr, flush=True)
            os._exit(86)
        except Exception as e:       # e.g. a collective this RCCL / torch build lacks: say so and fall back
            if args.mode == "sharded":
                raise
            sharded_error = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([0 if sharded_error else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # every rank takes the same decision
        if int(ok.item()) == 0:
            sharded_error = sharded_error or "the preflight failed on another rank"
            if native is not None:
                native.destroy(); native = None
            else:
                TorchComm.uninstall(ctx)
            comm, mode = None, "partitions"
            db0.free()
            db0, wl = build_workload(ctx, args.config, max(1000, pairs // world))
        elif comm_error:
            sharded_error = "native communicator unavailable (%s); torch.distributed P2P used" % comm_error
    if mode == "partitions" and world > 1:
        wl["note"] = "independent partition of 1/%d of the job per rank (same seed model, no data-path collective)" % world
    n_frag = wl["protein_fragments"]

    def barrier():
        ctx.sync(); torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.sync(); torch.cuda.synchronize()

    xstat = (lambda: native.stats()) if native is not None else ((lambda: (comm.bytes_moved, comm.seconds, comm.calls)) if comm is not None else None)

    def run(n_steps, record):
        db, rows, total = db0, [], 0
        for s in range(n_steps):
            it = s % chain
            if it == 0 and db is not db0:
                db.free(); db = db0
            ts = time.perf_counter()
            if VERBOSE and rank == 0:
                i = db.info(); print("step %d iteration %d: %d sequences, %d residues, longest entry %d" % (s, it, i["n"], i["residues"], i["max_entry_len"]), file=sys.stderr, flush=True)
            out, kst, rst, ast, wall = one_iteration(ctx, db, it, xstat)
            if VERBOSE and rank == 0:
                print("   N_k=%d N_m=%d N_c=%d cached=%d extract %.1f (short %.1f wave %.1f) | scored=%d accepted=%d | aln=%d extended=%d rescored=%d db +%.2f GB / copy %.2f GB asm %.1f | wall ms %s" % (
                    kst.n_kmer_records, kst.n_grouped, kst.n_candidates, kst.n_cached_sequences, kst.ms_extract, kst.ms_extract_short_kernel, kst.ms_extract_wave_kernel,
                    rst.n_scored, rst.n_accepted, ast.n_alignments, ast.n_extended, ast.n_rescored, ast.db_appended_bytes / 1e9, ast.db_copied_bytes / 1e9, ast.ms_kernel,
                    ["%.1f" % x for x in wall[:3]]), file=sys.stderr, flush=True)
            if record:
                ctx.sync()
                rows.append((it, (time.perf_counter() - ts) * 1e3, kst, rst, ast, wall))
                total += kst.n_candidates
            if db is not db0:
                db.free()
            db = out
        return db, rows, total

    wdb, _, _ = run(warmup, False)
    if wdb is not db0:
        wdb.free()
    barrier()
    if comm is not None:
        comm.bytes_moved = 0; comm.seconds = 0.0; comm.calls = 0
    if native is not None:
        native.stats(reset=True)
    t0 = time.perf_counter()
    db, rows, overlaps = run(steps, True)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    final_info = db.info() if steps else db0.info()
    local_overlaps = overlaps
    elapsed, overlaps = pdist.reduce_step(dist, elapsed, overlaps, device="cuda")

    if rank == 0:
        stats = [stage_table(k, r, a) for (_, _, k, r, a, _) in rows]
        tot = {}
        for st in stats:
            for k, (ms, b, single, launches) in st.items():
                a = tot.setdefault(k, [0.0, 0, single, 0])
                a[0] += ms; a[1] += b; a[3] += launches
        # the single kernel with the largest total time over the timed iterations; its numbers are per launch
        dom = max((k for k in tot if tot[k][2]), key=lambda k: tot[k][0]) if tot else None
        roof = None
        if dom:
            ms_avg = tot[dom][0] / max(tot[dom][3], 1)
            bytes_avg = tot[dom][1] / max(tot[dom][3], 1)
            achieved = bytes_avg / (ms_avg * 1e-3) / 1e9 if ms_avg > 0 else 0.0
            km = tot["kmermatcher_stage"]
            traffic, traffic_note = stored_traffic(dom, tot[dom][3] / len(stats), args.config)
            stage = lambda key: {"algorithmic_bytes_per_step": tot[key][1] / len(stats), "ms_per_step": tot[key][0] / len(stats),
                                 "frac": (tot[key][1] / (tot[key][0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tot[key][0] > 0 else 0.0}
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "kernel_note": "`%s` = every instantiation of that kernel template launched in a step, timed together with HIP events on the library's stream "
                                   "(rocprofv3 lists the instantiations as separate symbols: %s)" % (dom, KERNEL_STATS_FILE),
                    "traffic": traffic, "traffic_note": traffic_note, "rescore_stage": stage("rescore_stage"), "assemble_stage": stage("assemble_stage"), "ms_per_launch": ms_avg, "algorithmic_bytes_per_launch": bytes_avg, "launches_per_step": tot[dom][3] / len(stats),
                    "stage_ms_per_step": {k: round(v[0] / len(stats), 4) for k, v in tot.items()},
                    "kmermatcher_stage": {"algorithmic_bytes_per_step": km[1] / len(stats), "ms_per_step": km[0] / len(stats),
                                          "frac": (km[1] / (km[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if km[0] > 0 else 0.0},
                    "furthest_below": furthest_below(tot, len(stats), args.config),
                    "module_wall_ms_per_step": [round(sum(r[5][i] for r in rows) / len(rows), 3) for i in range(3)],
                    "host_waits_per_step": [round(sum(r[5][3 + i] for r in rows) / len(rows), 1) for i in range(3)]}
        line = {
            "metric": "read-overlaps/s per assembly iteration", "value": overlaps / elapsed if elapsed > 0 else 0.0, "unit": "overlaps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / max(steps, 1),
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "u8/u64 (integer hash, byte compare; f32 ratios)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]%s: %d synthetic 2x150 bp reads (%d read pairs from %d genomes, %.1fx mean coverage, seed %d) -> %d protein fragments; "
                                   "step s = iteration (s mod %d) of the plass assemble chain (--num-iterations %d, k=14, alph 13, kmer-per-seq 60, min-seq-id 0.9, e 1e-5)"
                                   % (cfg_idx if world == 1 or args.config != "c3" else 3, "" if not args.pairs else " model at another size", wl["reads"], wl["read_pairs"], wl["genomes"], wl["mean_coverage"], wl["seed"],
                                      n_frag, chain, chain),
                       "parallelism": ("1 GPU" if world == 1 and mode != "sharded" else
                                       "%d GPUs, one read set of %d fragments sharded by k-mer bucket (RCCL all-to-all(v) of k-mer and grouped records, "
                                       "all-gather of extended sequences), sequence DB replicated" % (world, n_frag) if mode == "sharded" else
                                       "1 process per GPU, independent partitions"),
                       "candidate_overlaps": overlaps, "workload_detail": wl,
                       "final_db": {"sequences": final_info["n"], "residues": final_info["residues"]}},
            "iterations": [{"step": i, "iteration": it, "ms": round(ms, 3), "N_k": k.n_kmer_records, "N_m": k.n_grouped, "N_c": k.n_candidates,
                            "verified": r.n_accepted, "extended": a.n_extended, "residues": k.residues,
                            "kmermatcher_ms": round(k.ms_extract + k.ms_sort1 + k.ms_group + k.ms_sort2 + k.ms_reduce, 3),
                            "extract_ms": round(k.ms_extract, 3), "partition_ms": round(k.ms_sort1, 3), "group_ms": round(k.ms_group, 3),
                            "repsort_ms": round(k.ms_sort2, 3), "reduce_ms": round(k.ms_reduce, 3),
                            "rescore_ms": round(r.ms_kernel, 3), "assemble_ms": round(a.ms_kernel, 3), "module_wall_ms": [round(x, 3) for x in w[:3]], "host_waits": list(w[3:6])}
                           for i, (it, ms, k, r, a, w) in enumerate(rows)],
            "roofline": roof,
        }
        if sharded_error is not None:
            line["sharded_mode_error"] = sharded_error
        if comm is not None:
            line["exchange"] = {"communicator": "torch.distributed P2P (RCCL)", "device_bytes_sent_per_step_rank0": comm.bytes_moved / max(steps, 1),
                                "collective_calls_per_step": comm.calls / max(steps, 1), "ms_in_collectives_per_step_rank0": comm.seconds * 1e3 / max(steps, 1)}
        if native is not None:
            nb, nsec, ncalls = native.stats()
            line["exchange"] = {"communicator": "native RCCL (plasship_rccl, ncclSend/ncclRecv groups on the context stream)", "device_bytes_sent_per_step_rank0": nb / max(steps, 1),
                                "collective_calls_per_step": ncalls / max(steps, 1), "host_ms_in_collectives_per_step_rank0": nsec * 1e3 / max(steps, 1)}
        if "exchange" in line:
            # which module's exchanges: kmermatcher = exchange 1 (k-mer lines to their bucket's owner), exchange 2 (aggregated
            # triples to their representative's owner), the run heads, status rounds; rescorediagonal = none; assembleresults = the all-gather of
            # the extended sequences
            per = {}
            for mi, name in enumerate(("kmermatcher", "rescorediagonal", "assembleresults")):
                xt = [sum(r[5][6][mi][f] for r in rows) for f in range(3)]
                per[name] = {"device_bytes_sent_per_step_rank0": xt[0] / max(len(rows), 1), "host_ms_in_collectives_per_step_rank0": xt[1] * 1e3 / max(len(rows), 1),
                             "collective_calls_per_step": xt[2] / max(len(rows), 1)}
            line["exchange"]["per_module"] = per
            line["exchange"]["setup"] = setup_s                  # one-off set-up that happened BEFORE the timed steps (the preflight iteration = the links' first use)
            line["exchange"]["model"] = scaling_model(world, wl["reads"], line["roofline"]["module_wall_ms_per_step"] if line.get("roofline") else None)
    if db is not db0:
        db.free()
    # ---- untimed verification: one more traversal of the chain, digest of every iteration's output DB (include/plasship.h:
    # plasship_seqdb_digest, the number the CPU oracle's `dbsum` computes from DB files) against the committed digests of this
    # workload (tests/golden/c3_chain_digests.json: agreed on by the line-store, the dense-partition and the sharded path,
    # tests/test_gpu_large.py).  A 32-bit slip above 2^32 record slots can no longer print a plausible rate unnoticed.
    verify = None
    if not args.no_verify:
        vdb, digests = db0, []
        for it in range(chain):
            out, _, _, _, _ = one_iteration(ctx, vdb, it)
            digests.append(out.digest()[0])
            if vdb is not db0:
                vdb.free()
            vdb = out
        if vdb is not db0:
            vdb.free()
        verify = {"seq_digests": digests, "reference": None, "match": None}
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "%s_chain_digests.json" % args.config)))
        except (OSError, ValueError):
            gold = None
        if gold and gold.get("pairs") == wl["read_pairs"] and mode != "partitions":
            verify["reference"] = "tests/golden/%s_chain_digests.json" % args.config
            verify["match"] = digests == gold["digests"][:len(digests)]
            # (round 5: those digests are what the UNMODIFIED REFERENCE computes for this workload — c3: all twelve iterations of the 50 M-read chain,
            # c2: the oracle's digests, themselves pinned against the reference; profiles/r05_headline_pin_reference.txt, r05_deep_pin_reference.txt)
            verify["reference_is"] = "identical to the unmodified reference binaries' run of this workload (profiles/r05_headline_pin_reference.txt, profiles/r05_deep_pin_reference.txt)"
    # ---- CPU baseline + the drop-in command line on the same sample.  BEFORE the wall-clock leg below: that leg's child process takes and
    # returns 270 GB of HBM, and for seconds afterwards every process that allocates device memory waits for the driver to scrub it — the
    # nine short `plass-hip` invocations of this leg then measure that (round 4's first evidence run: 4.8 s instead of 1.3 s;
    # tools/cli_dropin_probe.py, profiles/r04_calls/call21_cli_dropin_probe.log) ----
    cpu_line = None
    if rank == 0 and world == 1 and comm is None and native is None and not args.no_cpu_baseline:
        # (64 host cores and more: 1 M pairs through ALL iterations of the chain, about a minute; fewer: three iterations of a smaller sample)
        cpu_line = cpu_baseline(ctx, args.config, args.cpu_sample_pairs, chain if len(os.sched_getaffinity(0)) >= 32 else min(chain, 3))
    # ---- wall clock to final contigs (the metric's second half): the fragment DB goes to disk, then the fused C++ driver
    # (`plass-hip assemble-chain`: the loop of data/assemble.sh:85-156 incl. findassemblystart, DBs chained in HBM) runs as its own
    # process from DB files on disk to the final assembly DB on disk.  This process gives its HBM back first.
    # Round 6: the child runs TWICE.  Straight behind this process, which has just freed 250 GB, the child's one large hipMalloc waits 5-6 s
    # while the driver clears that memory (profiles/r06_ab_knobs.txt, calls 3-5: 25-44 GB/s INSIDE hipMalloc; memory nobody has touched since
    # it was cleared costs nothing; the driver also clears freed memory in the background) — a property of starting behind a job that has
    # just ended, not of the path.  The second run starts after PLASS_BENCH_SCRUB_WAIT seconds (default 20) of an idle GPU: `seconds` is
    # that run, the first one is kept as `seconds_behind_a_job_that_just_freed_the_hbm`.
    wall = None
    if rank == 0 and world == 1 and mode == "single" and args.config in ("c2", "c3") and not args.no_wall:
        import shutil
        wall = {"seconds": None}
        tdir = tempfile.mkdtemp(prefix="plass_wall_", dir=os.environ.get("PLASS_BENCH_TMPDIR"))

        def run_chain_child():
            import __graft_entry__ as g
            t0w = time.perf_counter()
            pr = subprocess.run([os.path.join(ROOT, "plass_amd", "plass-hip"), "assemble-chain", os.path.join(tdir, "aa_6f_start_long"), os.path.join(tdir, "assembly_final"),
                                 "--num-iterations", str(chain)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=g.child_env())
            secs = round(time.perf_counter() - t0w, 3)
            m = re.search(r"chain: .*", pr.stdout)
            for f in os.listdir(tdir):
                if f.startswith("assembly_final"):
                    os.unlink(os.path.join(tdir, f))
            return (secs if pr.returncode == 0 else None), (m.group(0) if m else pr.stdout[-400:]), pr.returncode

        try:
            need = 6 * db0.info()["data_bytes"] + (8 << 30)
            if shutil.disk_usage(tdir).free < need:
                wall["skipped"] = "less than %d GB free under %s" % (need >> 30, tdir)
            else:
                db0.write(os.path.join(tdir, "aa_6f_start_long"))
                db0.free(); db0 = None
                ctx.close()                                   # the arena (88 % of the HBM) goes back to the driver: the child needs it
                secs, brk, rc = run_chain_child()
                wall["seconds"], wall["breakdown"] = secs, brk
                if rc != 0:
                    wall["error"] = brk
                wall["what"] = ("`plass-hip assemble-chain aa_6f_start_long assembly_final --num-iterations %d`: process start to exit, fragment DB files on disk -> final "
                                "assembly DB files on disk, iteration 0 with findassemblystart like the workflow" % chain)
                # the reference's three modules take 4.4 s per iteration on 0.4 M reads with 8 threads, scaled linearly in the reads
                wall["reference_scaled_seconds"] = round(4.4 * chain * wl["reads"] / 400000.0, 0)
                wall["reference_note"] = "BASELINE.md section 2 (8 threads, AVX2, build container), module times only, scaled linearly with the number of reads"
                wait_s = float(os.environ.get("PLASS_BENCH_SCRUB_WAIT", "20"))
                if rc == 0 and wait_s > 0 and args.config == "c3":
                    time.sleep(wait_s)
                    secs2, brk2, rc2 = run_chain_child()
                    if rc2 == 0:
                        wall["seconds_behind_a_job_that_just_freed_the_hbm"], wall["breakdown_behind_a_job_that_just_freed_the_hbm"] = secs, brk
                        wall["seconds"], wall["breakdown"] = secs2, brk2
                        wall["note"] = ("two runs of the same command: `seconds` started after %.0f s of an idle GPU (the driver clears freed memory in the background), the other "
                                        "one straight behind this process, whose 250 GB the driver then clears inside the child's one large hipMalloc") % wait_s
                ctx = plass_amd.Context(local)
        finally:
            shutil.rmtree(tdir, ignore_errors=True)
    if db0 is not None:
        db0.free()
    if rank == 0:
        line["verify"] = verify
        line["wall_to_contigs"] = wall
        line["cpu_baseline"] = cpu_line
    if native is not None:
        native.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        # the ONE JSON line is the last thing on stdout: RCCL prints a version banner through C stdio, which (redirected to a file or a
        # pipe) would otherwise be flushed at exit, behind the line
        import ctypes
        sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


# ---------------------------------------------------------------------------------------------------------------------------
# --config c5: BASELINE configs[4] — PenguiN's nucleotide-level chains on the 20 M-read set (single GPU)
# ---------------------------------------------------------------------------------------------------------------------------
def penguin_step(ctx, st, it, gd_iters):
    """one step of the c5 chain: iterations 0..gd_iters-1 = protein-guided (kmermatcher on the ORFs' protein twins, rescorediagonal -a 1,
    proteinaln2nucl, guidedassembleresults: data/guidedNuclAssemble.sh:77-126), the rest = nucleotide (kmermatcher -k 22, rescorediagonal,
    nuclassembleresults, cyclecheck --chop-cycle: data/nuclassemble.sh:95-137) on the reads.  Returns (kst, rst, ast, extra ms, kind)"""
    import plass_amd
    if it < gd_iters:
        if it == 0:
            st["nu"], st["aa"] = st["nu0"], st["aa0"]
        nu, aa = st["nu"], st["aa"]
        c, kst = ctx.kmermatcher(aa, plass_amd.KmermatchParams(k=14, alph_size=13, kmer_per_seq=60, kmer_per_seq_scale=0.1, hash_shift=67, include_only_extendable=True,
                                                             ignore_multi_kmer=True, cov_mode=1, c=0.0))
        a, rst = ctx.rescorediagonal(aa, aa, c, plass_amd.RescoreParams(min_seq_id=0.97, cov_mode=1, a=True))
        na, nst = ctx.proteinaln2nucl(nu, aa, a)
        nu2, aa2, ast = ctx.guidedassembleresults(nu, aa, na)
        for x in (na, a, c):
            x.free()
        if nu is not st["nu0"]:
            nu.free(); aa.free()
        st["nu"], st["aa"] = nu2, aa2
        return kst, rst, ast, nst.ms_kernel, "guided"
    if it == gd_iters:
        if st.get("nu") is not None and st["nu"] is not st["nu0"]:
            st["nu"].free(); st["aa"].free()
        st["nu"] = st["aa"] = None
        if st.get("db") is not None and st["db"] is not st["reads"]:
            st["db"].free()                                  # the rest of the previous traversal
        st["db"] = st["reads"]
    db = st["db"]
    c, kst = ctx.kmermatcher(db, plass_amd.KmermatchParams(k=22, alph_size=5, kmer_per_seq=60, kmer_per_seq_scale=0.1, hash_shift=67, include_only_extendable=True,
                                                         ignore_multi_kmer=True, cov_mode=0, c=0.0))
    a, rst = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.99))
    out, ast = ctx.assembleresults(db, a, plass_amd.AssembleParams(min_seq_id=0.99, max_seq_len=200000))
    cyc, rest, cst = ctx.cyclecheck(out, max_seq_len=200000, chop_cycle=True, with_rest=True)
    for x in (a, c, cyc, out):
        x.free()
    if db is not st["reads"]:
        db.free()
    st["db"] = rest
    return kst, rst, ast, cst.ms_kernel, "nucleotide"


def main_c5(args):
    import torch
    import plass_amd
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1:
        raise SystemExit("--config c5 runs on one GPU (the sharded bench is --config c3)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    cfg_idx, cfg_pairs, _, _, _, _, _, chain = CONFIGS["c5"]
    chain, gd_iters = 10, 5                                   # 5 protein-guided + 5 nucleotide iterations
    pairs = args.pairs or cfg_pairs
    steps = chain if args.steps is None else args.steps
    warmup = chain if args.warmup is None else args.warmup
    ctx = plass_amd.Context(0)
    sp = synth_params("c5", pairs)
    reads, sst = ctx.synth_read_pairs(sp)
    nu0, aa0 = ctx.penguin_guided_inputs(reads)
    st = {"reads": reads, "nu0": nu0, "aa0": aa0}
    ri, ni = reads.info(), nu0.info()

    def run(n, record, digests=None):
        rows, total = [], 0
        for s_ in range(n):
            it = s_ % chain
            ts = time.perf_counter()
            kst, rst, ast, extra, kind = penguin_step(ctx, st, it, gd_iters)
            if record:
                ctx.sync()
                rows.append((it, (time.perf_counter() - ts) * 1e3, kst, rst, ast, extra, kind)); total += kst.n_candidates
            if digests is not None:                          # what the step left behind: the extended ORFs and their twins, or the non-circular contigs
                digests.append("+".join(x.digest()[0] for x in ((st["nu"], st["aa"]) if kind == "guided" else (st["db"],))))
        return rows, total

    run(warmup, False)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, overlaps = run(steps, True)
    ctx.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # ---- untimed verification: one more traversal, digest of every step's output DB(s) against the committed digests of this workload
    # (tests/golden/c5_chain_digests.json — written down from the GPU path's output in round 4; round 5: identical to the unmodified reference's
    # run of the same 20 M reads, profiles/r05_headline_pin_reference.txt) ----
    verify = None
    if not args.no_verify:
        digests = []
        run(chain, False, digests)
        verify = {"step_digests": digests, "reference": None, "match": None}
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_chain_digests.json")))
        except (OSError, ValueError):
            gold = None
        if gold and gold.get("pairs") == sp.n_pairs:
            verify["reference"] = "tests/golden/c5_chain_digests.json"
            verify["match"] = digests == gold["digests"][:len(digests)]
            verify["reference_is"] = "identical to the unmodified reference binaries' run of this workload (profiles/r05_headline_pin_reference.txt)"
    tot = {}
    for (_, _, k, r, a, extra, kind) in rows:
        tab = stage_table(k, r, a, nucl_queue=True)
        tab["proteinaln2nucl / cyclecheck"] = (extra, 0, False, 1)
        for key, (ms, b, single, launches) in tab.items():
            v = tot.setdefault(key, [0.0, 0, single, 0]); v[0] += ms; v[1] += b; v[3] += launches
    dom = max((k for k in tot if tot[k][2]), key=lambda k: tot[k][0])
    ms_avg = tot[dom][0] / max(tot[dom][3], 1); bytes_avg = tot[dom][1] / max(tot[dom][3], 1)
    achieved = bytes_avg / (ms_avg * 1e-3) / 1e9 if ms_avg > 0 else 0.0
    traffic, traffic_note = stored_traffic(dom, tot[dom][3] / len(rows), "c5")
    stage = lambda key: {"algorithmic_bytes_per_step": tot[key][1] / len(rows), "ms_per_step": tot[key][0] / len(rows),
                         "frac": (tot[key][1] / (tot[key][0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tot[key][0] > 0 else 0.0}
    line = {"metric": "read-overlaps/s per assembly iteration", "value": overlaps / elapsed if elapsed > 0 else 0.0, "unit": "overlaps/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed * 1e3 / max(steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u64 (integer hash, byte compare; f32/f64 comparator)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] on ONE GPU: %d synthetic 2x150 nt reads (%d pairs from %d genomes, %.1fx mean coverage, seed %d); step s = iteration (s mod 10) of: "
                                   "5 protein-guided iterations on the %d ORFs of the reads and their protein twins (kmermatcher k=14 -> rescorediagonal -a 1 -> proteinaln2nucl -> "
                                   "guidedassembleresults), then 5 nucleotide iterations on the reads (kmermatcher k=22 -> rescorediagonal -> nuclassembleresults -> cyclecheck --chop-cycle). "
                                   "The workflow's own nucleotide stage starts from the extended ORFs + the reads (an index filter of the workflow script, not a hot-path module)"
                                   % (ri["n"], sp.n_pairs, sp.n_genomes, sst.mean_coverage, sp.seed, ni["n"]),
                       "parallelism": "1 GPU", "candidate_overlaps": overlaps},
            "iterations": [{"step": i, "iteration": it, "kind": kind, "ms": round(ms, 3), "N_k": k.n_kmer_records, "N_m": k.n_grouped, "N_c": k.n_candidates, "verified": r.n_accepted,
                            "extended": a.n_extended, "kmermatcher_ms": round(k.ms_extract + k.ms_sort1 + k.ms_group + k.ms_sort2 + k.ms_reduce, 3), "extract_ms": round(k.ms_extract, 3),
                            "partition_ms": round(k.ms_sort1, 3), "group_ms": round(k.ms_group, 3), "repsort_ms": round(k.ms_sort2, 3), "reduce_ms": round(k.ms_reduce, 3),
                            "rescore_ms": round(r.ms_kernel, 3), "assemble_ms": round(a.ms_kernel, 3), "aln2nucl_or_cyclecheck_ms": round(extra, 3)}
                           for i, (it, ms, k, r, a, extra, kind) in enumerate(rows)],
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": traffic_note, "ms_per_launch": ms_avg, "algorithmic_bytes_per_launch": bytes_avg,
                         "stage_ms_per_step": {k: round(v[0] / len(rows), 4) for k, v in tot.items()},
                         "kmermatcher_stage": stage("kmermatcher_stage"), "rescore_stage": stage("rescore_stage"), "assemble_stage": stage("assemble_stage"),
                         "furthest_below": furthest_below(tot, len(rows), "c5")}}
    for key in ("nu", "aa", "db"):
        x = st.get(key)
        if x is not None and x is not reads and x is not nu0 and x is not aa0:
            x.free()
    nu0.free(); aa0.free(); reads.free()
    ctx.close()
    line["verify"] = verify
    line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline_c5(args.cpu_sample_pairs)
    import ctypes
    sys.stderr.flush(); ctypes.CDLL(None).fflush(None)
    print(json.dumps(line), flush=True)


def cpu_baseline_c5(sample_pairs):
    """the CPU oracle on a bounded sample of the c5 community (>= 5 genomes, skewed): 2 nucleotide and 2 protein-guided iterations,
    candidate overlaps / module compute time"""
    import __graft_entry__ as g
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest as T
    from plass_amd import _lib
    if not os.path.exists(g.oracle_bin()):
        subprocess.check_call(["make", "-j", "4"], cwd=os.path.join(ROOT, "oracle"))
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    pairs = sample_pairs or (20000 if threads < 8 else 60000)
    sp = synth_params("c5", pairs, min_genomes=5)
    thr = ["--threads", str(threads)]
    tot_t, tot_c = 0.0, 0

    def timed(args):
        nonlocal tot_t, tot_c
        e = g.run_oracle(args + thr)
        tot_t += float(re.search(r"^oracle \w+:.*?([0-9.]+) s\s*$", e, re.M).group(1))
        m = re.search(r"N_c=(\d+)", e)
        if m:
            tot_c += int(m.group(1))

    with tempfile.TemporaryDirectory() as td:
        P = lambda n: os.path.join(td, n)
        g.run_oracle(["synthreads", P("reads"), "--pairs", str(sp.n_pairs), "--seed", str(sp.seed), "--genomes", str(sp.n_genomes), "--genome-min-len", str(sp.genome_min_len),
                      "--genome-max-len", str(sp.genome_max_len), "--abundance-sigma", repr(sp.abundance_sigma), "--insert-mean", repr(sp.insert_mean), "--insert-sd", repr(sp.insert_sd),
                      "--insert-min", str(sp.insert_min), "--read-len", str(sp.read_len), "--error-rate", repr(sp.error_rate)])
        src = P("reads")
        for it in range(2):
            timed(["kmermatcher", src, P("p")] + T.NUCL_KM); timed(["rescorediagonal", src, src, P("p"), P("a")] + T.NUCL_RS)
            timed(["nuclassembleresults", src, P("a"), P("s%d" % it)] + T.NUCL_AS[:6]); timed(["cyclecheck", P("s%d" % it), P("c"), "--max-seq-len", "200000", "--chop-cycle", "1"])
            src = P("s%d" % it)
        for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
            fl = []
            for k, v in par.items():
                fl += ["--" + k.replace("_", "-"), str(v)]
            g.run_oracle(["extractorfs", P("reads"), P("n_" + name)] + fl)
        g.run_oracle(["concatdbs", P("n_long"), P("n_start"), P("nu0")]); g.run_oracle(["concatdbs", P("n_long_h"), P("n_start_h"), P("nu0_h")])
        g.run_oracle(["translatenucs", P("nu0"), P("aa0"), "--add-orf-stop", "1"])
        for it in range(2):
            nu, aa = P("nu%d" % it), P("aa%d" % it)
            timed(["kmermatcher", aa, P("p")] + T.GD_KM); timed(["rescorediagonal", aa, aa, P("p"), P("a")] + T.GD_RS)
            timed(["proteinaln2nucl", nu, nu, aa, aa, P("a"), P("an")] + T.GD_P2N); timed(["guidedassembleresults", nu, aa, P("an"), P("nu%d" % (it + 1)), P("aa%d" % (it + 1))] + T.GD_AS[:6])
    return {"value": tot_c / tot_t, "unit": "overlaps/s", "cores": threads, "kind": "port", "cpu": host_cpu_model(), "seconds": round(tot_t, 2),
            "reference_full_workload": REFERENCE_FULL_WORKLOAD["c5"],
            "sample": "%d read pairs of the c5 community model at the same mean coverage, skewed (%d genomes, sigma 1): 2 nucleotide iterations (incl. cyclecheck) and 2 protein-guided "
                      "iterations (incl. proteinaln2nucl), oracle module compute time (no DB I/O), %d OpenMP threads" % (sp.n_pairs, sp.n_genomes, threads)}


def bounded(fn, seconds, what, rank):
    """runs fn() on a helper thread and waits at most `seconds` for it: the first sharded run on hardware nobody has touched must not be able to hang the
    job without a word.  A timeout raises with the rank and the phase in the message; the helper thread is left behind."""
    import threading
    box = {}

    def work():
        try:
            box["r"] = fn()
        except BaseException as e:                               # noqa: B902 (carried to the caller's thread)
            box["e"] = e
    th = threading.Thread(target=work, daemon=True)
    th.start(); th.join(seconds if seconds > 0 else None)
    if th.is_alive():
        msg = "rank %d: %s did not finish within %.0f s (PLASS_BENCH_COMM_TIMEOUT); NCCL_DEBUG=INFO shows RCCL's view of the links" % (rank, what, seconds)
        print("[bench] " + msg, file=sys.stderr, flush=True)
        raise TimeoutError(msg)
    if "e" in box:
        raise box["e"]
    return box.get("r")


def sharded_preflight(ctx, dist, device):
    """one sharded iteration on a small read set (the same on every rank): every rank must end with the same DB (checksum)"""
    import hashlib
    import torch
    from plass_amd import synth
    data, off, elen, key = synth.protein_fragment_db(3000, seed=7)
    db = ctx.upload_seqdb(data, off, elen, key, 0)
    out, kst, _, _, _ = one_iteration(ctx, db, 0)
    h = int.from_bytes(hashlib.sha256(out.download()[0]).digest()[:7], "little")
    t = torch.tensor([h, -h], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError("sharded preflight: the ranks ended with different output DBs")
    out.free(); db.free()


if __name__ == "__main__":
    main()
