#!/usr/bin/env python3
"""Where the fused driver's wall clock goes (GPU box): writes the 50 M-read fragment DB, then runs `plass-hip assemble-chain` as its own
process with PLASSHIP_POOL_STATS=1 and prints everything it says (per-iteration lines carry the time since the DB was read)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, plass_amd, __graft_entry__ as g
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 25000000
ctx = plass_amd.Context(0)
db, wl = bench.build_workload(ctx, "c3", pairs)
td = tempfile.mkdtemp(prefix="plass_wall_probe_")
db.write(os.path.join(td, "frag")); db.free(); ctx.close()
import time as _t
runs = ({}, {"_sleep": "20"}, {"_sleep": "20", "PLASSHIP_CLI_FULL_TEARDOWN": "1"}, {"_sleep": "20"})      # (round 6: straight behind this process; after an idle wait; with the orderly teardown)
for extra in runs:
    extra = dict(extra); _t.sleep(float(extra.pop("_sleep", "0")))
    env = dict(g.child_env()); env["PLASSHIP_POOL_STATS"] = "1"; env["PLASSHIP_IO_TIMING"] = "1"; env.update(extra)
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(ROOT, "plass_amd", "plass-hip"), "assemble-chain", os.path.join(td, "frag"), os.path.join(td, "out"), "--num-iterations", "12"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    print("=== extra env", extra, "wall %.2f s rc %d" % (time.perf_counter() - t0, p.returncode)); print(p.stdout[-3000:], flush=True)
    for f in os.listdir(td):                      # (a run that overwrites the previous run's 17 GB pays for their write-back: 6.1 instead of 2.5 s of "write")
        if f.startswith("out"):
            os.unlink(os.path.join(td, f))
import shutil; shutil.rmtree(td, ignore_errors=True)
