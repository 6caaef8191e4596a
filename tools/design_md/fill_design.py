# Regenerates DESIGN.md from design_head.md + design_tail.md and the evidence of a round (bench lines, kernel trace): the numbers in the text
# are placeholders filled from the logs, so the document cannot quote a figure the logs do not hold.  Usage (repo root): python tools/design_md/fill_design.py <dir with bench_driver_cmd.log, bench_c5.log, bench_12M_single.log, bench_12M_sharded_1rank.log, kernel_stats_driver_cmd.txt>
import json, re, sys
O = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r04_final'
def last(f): return json.loads([x for x in open(f) if x.startswith("{")][-1])
j = last(O + '/bench_driver_cmd.log')
old = json.loads(open('profiles/r03_bench_driver_cmd.log').read())
r = j["roofline"]
rows = {}; orows = {}
for x in j["iterations"]: rows.setdefault(x["iteration"], []).append(x)
for x in old["iterations"]: orows.setdefault(x["iteration"], []).append(x)
lines = []
for it in range(12):
    x = rows[it][-1]; o = orows[it][-1]
    lines.append("| %d | %.2f G | %.2f G | %.2f G | %d M | %d | %d | %d | %d | %d | %d | %d | %d (%d) |" % (it, x["residues"]/1e9, x["N_k"]/1e9, x["N_m"]/1e9, round(x["N_c"]/1e6),
        round(x["extract_ms"]), round(x["partition_ms"]), round(x["group_ms"]), round(x["repsort_ms"]), round(x["reduce_ms"]), round(x["rescore_ms"]), round(x["assemble_ms"]), round(x["ms"]), round(o["ms"])))
c5 = last(O + '/bench_c5.log')
s12 = last(O + '/bench_12M_single.log'); sh12 = last(O + '/bench_12M_sharded_1rank.log')
w = j.get("wall_to_contigs") or {}
st = r["stage_ms_per_step"]
ks = open(O + '/kernel_stats_driver_cmd.txt').read()
def avg(pat):
    m = re.search(pat + r".*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s*$", ks, re.M)
    return float(m.group(3)) / 1000 if m else float('nan')
sub = {
 "MS": "%.1f" % j["ms_per_step"], "MOPS": "%d" % round(j["value"]/1e6), "CPU": "%.2f" % ((j.get("cpu_baseline") or {}).get("value", 0)/1e6),
 "KM_MS": "%.1f" % r["kmermatcher_stage"]["ms_per_step"], "KM_FRAC": "%.1f" % (100*r["kmermatcher_stage"]["frac"]),
 "R_GB": "%.1f" % (r["rescore_stage"]["algorithmic_bytes_per_step"]/1e9), "R_MS": "%.1f" % r["rescore_stage"]["ms_per_step"], "R_FRAC": "%.1f" % (100*r["rescore_stage"]["frac"]),
 "A_GB": "%.1f" % (r["assemble_stage"]["algorithmic_bytes_per_step"]/1e9), "A_MS": "%.1f" % r["assemble_stage"]["ms_per_step"], "A_FRAC": "%.1f" % (100*r["assemble_stage"]["frac"]),
 "ITER_ROWS": "\n".join(lines), "EXT": "%.1f" % (st["extractShortKernel"] + st["extractKernel"]),
 "C5_MS": "%.0f" % c5["ms_per_step"], "C5_MOPS": "%.1f" % (c5["value"]/1e6), "C5_TRAFFIC": ("%.0f" % (c5["roofline"]["traffic"]/1e9)) if c5["roofline"].get("traffic") else "n/a",
 "WALL": ("%.1f" % w["seconds"]) if w.get("seconds") else "n/a", "WALL_BREAK": (w.get("breakdown") or "").replace("chain: ", ""),
 "WALL_VERDICT": "**not reached**",
 "S12": "%.1f" % s12["ms_per_step"], "SH12": "%.1f" % sh12["ms_per_step"], "SH_OVER": "%+.0f %%" % (100*(sh12["ms_per_step"]/s12["ms_per_step"]-1)), "S12NC": "n/m",
 "DOM_MS": "%.1f" % r["ms_per_launch"], "DOM_FRAC": "%.1f" % (100 * r["frac"]), "EXTW": "%.1f" % st["extractKernel"], "PART_MS": "%.1f" % st["partitionKernel(k-mer records)"],
 "SK": "%.1f" % avg(r"extractShortFastKernel<false, 14, true>"),
 "C5_IT9": "%.0f" % c5["iterations"][9]["ms"], "C5_EXT9": "%.0f" % c5["iterations"][9]["extract_ms"], "C5_ASM9": "%.0f" % c5["iterations"][9]["assemble_ms"],
 "C5_CYC9": "%.0f" % c5["iterations"][9]["aln2nucl_or_cyclecheck_ms"], "C5_RESC9": "%.0f" % c5["iterations"][9]["rescore_ms"],
 "C5_DOM": "`" + c5["roofline"]["kernel"] + "`", "C5_DOM_MS": "%.1f" % c5["roofline"]["ms_per_launch"], "C5_DOM_FRAC": "%.1f" % (100 * c5["roofline"]["frac"]),
 "C5_VERDICT": ("**reached, by a hair** (%.0f ms in the evidence run)" % c5["ms_per_step"] if c5["ms_per_step"] <= 300 else "**not reached** (%.0f ms)" % c5["ms_per_step"]),
 "T2": "%.1f" % avg(r"extractKernel<false, false, 128, false, 48, 992, 0>"), "CK": "%.1f" % avg(r"extractCachedKernel"), "AK": "%.1f" % avg(r"appendOutKernel<8>"),
}
t = open('tools/design_md/design_head.md').read() + open('tools/design_md/design_tail.md').read()
for k, v in sub.items(): t = t.replace("«" + k + "»", v)
left = re.findall(r"«[A-Z0-9_]+»", t)
print("unfilled:", sorted(set(left)))
open('DESIGN.md', 'w').write(t)
print(len(t.splitlines()), "lines,", len(t), "bytes")
