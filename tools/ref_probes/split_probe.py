# BUILD CONTAINER ONLY (needs the survey-time reference build, /tmp/plass-build): one of the three probes behind DESIGN.md section 2 (i)-(iii) / section 5
# (split_probe: see profiles/r05_deep_pin_reference.txt for what it showed).  Scratch under /tmp/pin.
import sys,os,subprocess,shutil,time,glob,json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import bench, __graft_entry__ as g
from plass_amd import _lib
from make_deep_chains import synth
import pin_deep_chains_against_reference as pin
D='/tmp/pin/sp'
shutil.rmtree(D, ignore_errors=True); os.makedirs(D)
P=lambda n:D+'/'+n
fx=json.load(open('/root/repo/tests/golden/deep_chains.json'))["c3_deep"]
sp=bench.synth_params("c3",fx["pairs"])
synth(g,sp,P('reads')); pin.write_header_db(P('reads'),P('reads_h'))
q=["--threads","8","-v","1"]
for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
    pin.ref(pin.PLASS, ["extractorfs", P("reads"), P("nucl_" + name)] + pin.orf_flags(par), pin.Q1)
    pin.ref(pin.PLASS, ["translatenucs", P("nucl_" + name), P("aa_" + name), "--add-orf-stop", "1"], pin.Q1)
pin.ref(pin.PLASS, ["concatdbs", P("aa_long"), P("aa_start"), P("seq_0")], q)
print("fragments", pin.same(P("seq_0"), fx["fragments"]))
km = ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "nucl:0.200,aa:0.000", "-k", "14", "-c", "0", "--cov-mode", "0", "--ignore-multi-kmer", "1",
      "--max-seq-len", "65535", "--hash-shift", "67", "--include-only-extendable", "0"]
for lim in ("0", "2G", "1G", "500M"):
    out = subprocess.run([pin.PLASS, "kmermatcher", P("seq_0"), P("pref_"+lim)] + km + ["--split-memory-limit", lim, "--threads", "8", "-v", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    parts=[l for l in out.splitlines() if "parts" in l.lower() or "split" in l.lower()]
    got=pin.db_sums(P("pref_"+lim))
    print("split-memory-limit", lim, "->", got, "fixture", fx["iterations"][0]["pref"]["digest"], "|", parts[:3], flush=True)
