#!/usr/bin/env python3
"""Prints the numbers DESIGN.md section 8 quotes from an evidence run (tools/gpu_round6_final.sh): python tools/evidence_numbers.py [gpurun_out/r06_final]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
O = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06_final")
def last(p):
    return json.loads([x for x in open(os.path.join(O, p)) if x.startswith('{')][-1])
d = last('bench_driver_cmd.log')
print('c3 %.1f ms  %.1f M/s  verify %s' % (d['ms_per_step'], d['value'] / 1e6, d['verify'].get('match')))
r = d['roofline']
print('dominant', r['kernel'], 'frac %.3f' % r['frac'], 'traffic %.1f GB' % (r['traffic'] / 1e9 if r['traffic'] else -1), 'ms/launch %.1f' % r['ms_per_launch'])
print({k: round(v, 1) for k, v in r['stage_ms_per_step'].items()})
for k in ('kmermatcher_stage', 'rescore_stage', 'assemble_stage'):
    print(k, '%.1f ms  %.1f %%  %.1f GB' % (r[k]['ms_per_step'], 100 * r[k]['frac'], r[k]['algorithmic_bytes_per_step'] / 1e9))
print('furthest', r['furthest_below'])
w = d['wall_to_contigs']; print('wall', w.get('seconds'), '|', w.get('breakdown'), '| behind:', w.get('seconds_behind_a_job_that_just_freed_the_hbm'))
c = d['cpu_baseline']; print('cpu %.2f M/s in %.0f s; drop-in %.2f M/s' % (c['value'] / 1e6, c['seconds'], c['drop_in_cli_same_sample']['value'] / 1e6))
its = {}
for it in d['iterations']:
    its[it['iteration']] = it
for i in range(12):
    x = its[i]
    print('| %d | %.2f G | %.2f G | %.2f G | %d M | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f |' % (i, x.get('residues', 0) / 1e9, x['N_k'] / 1e9, x['N_m'] / 1e9, x['N_c'] / 1e6, x['extract_ms'], x['partition_ms'], x['group_ms'], x['repsort_ms'], x['reduce_ms'], x['rescore_ms'], x['assemble_ms'], x['ms']))
t = json.load(open(os.path.join(O, 'pmc_traffic.json')))
print('pmc source_sha', t['source_sha'], 'build', bench.source_sha())
st = {'km': 0, 'rs': 0, 'as': 0}
for name, rr in t['kernels'].items():
    b = rr['hbm_bytes_per_launch'] * rr['launches'] / t.get('steps', 25)
    if re.search(r'synth|orfKernel|translate|concat|digest|rocclr|keysDiffer|nonZero|maxLen', name): continue
    k = 'rs' if re.search(r'rescoreKernel|packOffLen|markLengths', name) else ('as' if re.search(r'assemble|arenaSum|arenaSize|listKernel|outLen|appendOut|writeOut|maxU32|bigNeed', name) else 'km')
    st[k] += b
print('stage traffic GB', {k: round(v / 1e9, 1) for k, v in st.items()})
d5 = last('bench_c5.log'); print('c5 %.1f ms %.1f M/s verify %s' % (d5['ms_per_step'], d5['value'] / 1e6, d5['verify'].get('match'))); print({k: round(v, 1) for k, v in d5['roofline']['stage_ms_per_step'].items()}); print(d5['roofline'].get('furthest_below'))
d2 = last('bench_c2.log'); print('c2 %.2f ms %.1f M/s verify %s' % (d2['ms_per_step'], d2['value'] / 1e6, d2['verify'].get('match')))
for f in ('bench_12M_single.log', 'bench_12M_sharded_1rank.log'):
    x = last(f); ms = [it['ms'] for it in x['iterations'] if it['iteration'] > 0][:11]; print(f, '%.1f' % (sum(ms) / len(ms)))
for W in (2, 4, 8):
    m = bench.scaling_model(W, 50e6); print('model', W, m['owner_filtered']['speedup'], m['exchange']['speedup'])
print(open(os.path.join(O, 'pytest_gpu.log')).read().strip().splitlines()[-1])
