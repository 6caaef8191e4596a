#!/usr/bin/env python3
"""round 5, GPU call 3 (debug): tests/test_gpu_deep.py::test_four_guided_iterations found 3 bytes of difference in the proteinaln2nucl DB of
guided iteration 2 (2 M reads of the configs[4] model; pref and aln of that iteration equal the oracle's).  Reproduce the chain up to there,
run the ORACLE's proteinaln2nucl on the GPU-written inputs of that iteration and print the lines that differ."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, plass_amd, __graft_entry__ as g
import conftest as T
from test_gpu_parity import gd_km_params, gd_rs_params
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "deep_chains.json")))["c5_deep"]
ctx = plass_amd.Context(0)
sp = bench.synth_params(gold["config"], gold["pairs"])
reads, _ = ctx.synth_read_pairs(sp)
nu, aa = ctx.penguin_guided_inputs(reads); reads.free()
td = tempfile.mkdtemp(prefix="dbg_guided_")
P = lambda n: os.path.join(td, n)
for it in range(3):
    cands, _ = ctx.kmermatcher(aa, gd_km_params())
    alns, _ = ctx.rescorediagonal(aa, aa, cands, gd_rs_params()); cands.free()
    naln, _ = ctx.proteinaln2nucl(nu, aa, alns)
    if it == 2:
        nu.write(P("nu")); aa.write(P("aa")); alns.write(P("aln")); naln.write(P("gpu_aln_nucl"))
        print(g.run_oracle(["proteinaln2nucl", P("nu"), P("nu"), P("aa"), P("aa"), P("aln"), P("ora_aln_nucl")] + T.GD_P2N + ["--threads", "32"]).strip())
        a = open(P("gpu_aln_nucl"), "rb").read().split(b"\0"); b = open(P("ora_aln_nucl"), "rb").read().split(b"\0")
        print("entries", len(a), len(b))
        ia = {int(l.split(b"\t")[0]): (int(l.split(b"\t")[1]), int(l.split(b"\t")[2])) for l in open(P("gpu_aln_nucl.index"), "rb")}
        ib = {int(l.split(b"\t")[0]): (int(l.split(b"\t")[1]), int(l.split(b"\t")[2])) for l in open(P("ora_aln_nucl.index"), "rb")}
        da = open(P("gpu_aln_nucl"), "rb").read(); dbb = open(P("ora_aln_nucl"), "rb").read()
        nd = 0
        for k in sorted(ia):
            ea = da[ia[k][0]:ia[k][0] + ia[k][1]]; eb = dbb[ib[k][0]:ib[k][0] + ib[k][1]]
            if ea != eb:
                la = ea.split(b"\n"); lb = eb.split(b"\n")
                for x, y in zip(la, lb):
                    if x != y:
                        print("key", k, "\n  gpu:", x.decode(), "\n  ora:", y.decode()); nd += 1
                        # the protein alignment line behind it
                        tgt = x.split(b"\t")[0]
                        ai = {int(l.split(b"\t")[0]): (int(l.split(b"\t")[1]), int(l.split(b"\t")[2])) for l in open(P("aln.index"), "rb")} if nd == 1 else None
                        if ai:
                            e = open(P("aln"), "rb").read()[ai[k][0]:ai[k][0] + ai[k][1]]
                            print("  protein lines of that query:", [l.decode() for l in e.split(b"\n") if l.split(b"\t")[0] == tgt])
                if len(la) != len(lb): print("key", k, "line counts differ", len(la), len(lb))
                if nd > 8: break
        print("differing lines shown:", nd)
    alns.free()
    nu2, aa2, _ = ctx.guidedassembleresults(nu, aa, naln)
    naln.free(); nu.free(); aa.free(); nu, aa = nu2, aa2
ctx.close()
