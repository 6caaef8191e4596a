# BUILD CONTAINER ONLY (needs the survey-time reference build, /tmp/plass-build): one of the three probes behind DESIGN.md section 2 (i)-(iii) / section 5
# (diag_guided: see profiles/r05_deep_pin_reference.txt for what it showed).  Scratch under /tmp/pin.
import sys,os,subprocess,shutil,time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import bench, __graft_entry__ as g
import conftest as T
from plass_amd import _lib
from make_deep_chains import synth
import pin_deep_chains_against_reference as pin
D='/tmp/pin/dg'
shutil.rmtree(D, ignore_errors=True); os.makedirs(D)
P=lambda n:D+'/'+n
pairs=int(sys.argv[1]) if len(sys.argv)>1 else 200000
sp=bench.synth_params("c5",pairs)
synth(g,sp,P('reads')); pin.write_header_db(P('reads'),P('reads_h'))
q=["--threads","8","-v","1"]
for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
    pin.ref(pin.PENGUIN, ["extractorfs", P("reads"), P("nucl_" + name)] + pin.orf_flags(par), pin.Q1)
pin.ref(pin.PENGUIN, ["concatdbs", P("nucl_long"), P("nucl_start"), P("nucl_0")], q)
pin.ref(pin.PENGUIN, ["concatdbs", P("nucl_long_h"), P("nucl_start_h"), P("nucl_0_h")], q)
pin.ref(pin.PENGUIN, ["translatenucs", P("nucl_0"), P("aa_0"), "--add-orf-stop", "1"], pin.Q1)
def load(p):
    import glob
    if os.path.exists(p): d=open(p,'rb').read()
    else: d=b''.join(open(f,'rb').read() for f in sorted(glob.glob(p+'.[0-9]*'), key=lambda f:int(f.rsplit('.',1)[1])))
    r={}
    for l in open(p+'.index','rb'):
        k,o,n=map(int,l.split()); r[k]=d[o:o+n]
    return r
for it in range(3):
    nu, aa, p, al = P("nucl_%d" % it), P("aa_%d" % it), P("pref"), P("aln")
    nu2, aa2 = P("nucl_%d" % (it + 1)), P("aa_%d" % (it + 1))
    pin.ref(pin.PENGUIN, ["kmermatcher", aa, p] + T.GD_KM + ["--max-seq-len", "200000"], q)
    pin.ref(pin.PENGUIN, ["rescorediagonal", aa, aa, p, al] + T.GD_RS, q)
    pin.ref(pin.PENGUIN, ["proteinaln2nucl", nu, nu, aa, aa, al, P("an_ref")] + T.GD_P2N, q)
    g.run_oracle(["proteinaln2nucl", nu, nu, aa, aa, al, P("an_or")] + T.GD_P2N + ["--threads","8"])
    a=load(P("an_ref")); b=load(P("an_or"))
    bad=[k for k in a if a[k]!=b.get(k)]
    print("iteration",it,"aln_nucl entries differing between the reference and the oracle on the SAME (reference-written) inputs:",len(bad),flush=True)
    for k in bad[:5]:
        la=a[k].split(b'\n'); lb=b[k].split(b'\n')
        for x,y in zip(la,lb):
            if x!=y: print("  query",k,"\n    ref   ",x,"\n    oracle",y)
    # layout: is the nucl DB's data file in key order?
    idx=[tuple(map(int,l.split())) for l in open(nu+'.index','rb')]
    inorder=all(idx[i][1]<idx[i+1][1] for i in range(len(idx)-1))
    print("   nucl_%d data file in key order: %s" % (it, inorder))
    pin.ref(pin.PENGUIN, ["guidedassembleresults", nu, aa, P("an_ref"), nu2, aa2] + T.GD_AS, q)
    pin.rm(p, al)
