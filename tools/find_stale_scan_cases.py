import numpy as np, subprocess, os, sys
sys.path.insert(0,'/root/repo')
from plass_amd import synth
O='/root/repo/oracle/build/plass_oracle'
KM="--alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale 0 -k 14 -c 0 --hash-shift 67 --ignore-multi-kmer 1".split()
os.makedirs('/tmp/q1',exist_ok=True)
aa="ACDEFGHIKLMNPQRSTVWY"
found=0
for seed in range(20000):
    rng=np.random.default_rng(seed)
    base="".join(rng.choice(list(aa), size=int(rng.integers(60,140))))
    n=int(rng.integers(3,9))
    seqs=[]
    for i in range(n):
        p=int(rng.integers(0,len(base)-30)); l=int(rng.integers(25,60))
        seqs.append(base[p:p+l])
    arrs=[np.frombuffer(s.encode(),dtype=np.uint8) for s in seqs]
    d=synth.pack_db(arrs)
    synth.write_db('/tmp/q1/s',*d,0)
    for ext in ("0","1"):
        subprocess.run([O,'kmermatcher','/tmp/q1/s','/tmp/q1/a']+KM+['--include-only-extendable',ext],stderr=subprocess.DEVNULL)
        subprocess.run([O,'kmermatcher','/tmp/q1/s','/tmp/q1/b']+KM+['--include-only-extendable',ext,'--oracle-no-stale-scan','1'],stderr=subprocess.DEVNULL)
        if open('/tmp/q1/a','rb').read()!=open('/tmp/q1/b','rb').read():
            print("FOUND seed",seed,"ext",ext,"n",n); found+=1
            os.makedirs('/tmp/q1/case%d'%found,exist_ok=True)
            for f in ('s','s.index','s.dbtype'): open('/tmp/q1/case%d/%s'%(found,f),'wb').write(open('/tmp/q1/'+f,'rb').read())
            open('/tmp/q1/case%d/ext'%found,'w').write(ext)
            if found>=4: sys.exit(0)
print("found",found)
