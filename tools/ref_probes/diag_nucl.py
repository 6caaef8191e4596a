# BUILD CONTAINER ONLY (needs the survey-time reference build, /tmp/plass-build): one of the three probes behind DESIGN.md section 2 (i)-(iii) / section 5
# (diag_nucl: see profiles/r05_deep_pin_reference.txt for what it showed).  Scratch under /tmp/pin.
import sys,os,subprocess,shutil,time,glob
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import bench, __graft_entry__ as g
import conftest as T
from plass_amd import _lib
from make_deep_chains import synth
import pin_deep_chains_against_reference as pin
D='/tmp/pin/dn'
shutil.rmtree(D, ignore_errors=True); os.makedirs(D)
P=lambda n:D+'/'+n
sp=bench.synth_params("c5",1000000)
synth(g,sp,P('reads'))
q=["--threads","8","-v","1"]
def load(p):
    if os.path.exists(p): d=open(p,'rb').read()
    else: d=b''.join(open(f,'rb').read() for f in sorted(glob.glob(p+'.[0-9]*'), key=lambda f:int(f.rsplit('.',1)[1])))
    r={}
    for l in open(p+'.index','rb'):
        k,o,n=map(int,l.split()); r[k]=d[o:o+n]
    return r
src=P('reads')
for it in range(3):
    p, al, o, cy, rest = P("pref"), P("aln"), P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it)
    pin.ref(pin.PENGUIN, ["kmermatcher", src, p] + T.NUCL_KM + ["--max-seq-len", "200000"], q)
    if it == 2:
        pin.ref(pin.PENGUIN, ["kmermatcher", src, P("pref1")] + T.NUCL_KM + ["--max-seq-len", "200000"], pin.Q1)
        pin.ref(pin.PENGUIN, ["kmermatcher", src, P("pref8b")] + T.NUCL_KM + ["--max-seq-len", "200000"], q)
        g.run_oracle(["kmermatcher", src, P("pref_or")] + T.NUCL_KM + ["--threads","8"])
        a=load(p); a1=load(P("pref1")); a8=load(P("pref8b")); b=load(P("pref_or"))
        for nm,x in (("reference 1 thread",a1),("reference 8 threads, second run",a8),("oracle",b)):
            bad=[k for k in a if a[k]!=x.get(k)]
            print("pref entries differing between the reference (8 threads) and", nm, ":", len(bad), bad[:10])
            for k in bad[:6]:
                sa=set(a[k].split(b'\n')); sx=set(x[k].split(b'\n'))
                print("   entry",k,"only in reference-8:",sorted(sa-sx),"only in",nm,":",sorted(sx-sa))
        break
    pin.ref(pin.PENGUIN, ["rescorediagonal", src, src, p, al] + T.NUCL_RS, q)
    pin.ref(pin.PENGUIN, ["nuclassembleresults", src, al, o] + T.NUCL_AS, q)
    pin.ref(pin.PENGUIN, ["cyclecheck", o, cy, "--max-seq-len", "200000", "--chop-cycle", "1"], q)
    pin.rest_db(o, cy, rest)
    src=rest
