#!/bin/bash
# GENERATION-TIME ONLY (build container): tests/golden/concat_noncanonical.tar.gz — inputs of concatdbs whose data files are NOT in
# key order (translatenucs run on 8 threads leaves them in thread order) and the reference's own concatenation of them, raw files, not
# passed through tools/dbcanon.py.  Reference binaries: the survey-time out-of-tree build (REF_BUILD, default /tmp/plass-build).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); B=${REF_BUILD:-/tmp/plass-build}; PLASS=$B/src/plass; EX=${REFERENCE:-/root/reference}/examples
W=$(mktemp -d); mkdir $W/w $W/concat
LONG="--min-length 45 --max-length 32734 --max-gaps 0 --contig-start-mode 2 --contig-end-mode 2 --orf-start-mode 0"
START="--min-length 20 --max-length 45 --max-gaps 0 --contig-start-mode 1 --contig-end-mode 0 --orf-start-mode 0"
$PLASS createdb $EX/reads_1.fastq.gz $W/w/reads -v 0
$PLASS extractorfs $W/w/reads $W/w/A $LONG --threads 8 -v 0
$PLASS extractorfs $W/w/reads $W/w/B $START --threads 8 -v 0
$PLASS translatenucs $W/w/A $W/w/aaA --add-orf-stop 1 --threads 8 -v 0
$PLASS translatenucs $W/w/B $W/w/aaB --add-orf-stop 1 --threads 8 -v 0
$PLASS concatdbs $W/w/aaA $W/w/aaB $W/w/aaC -v 0 --threads 1
# round 4: the header DBs of the same two passes — extractorfs writes ORFs and headers in the same thread order, the
# workflow concatenates both (data/assemble.sh:72-77), and the keys of the two concatenations must keep corresponding
$PLASS concatdbs $W/w/A_h $W/w/B_h $W/w/C_h -v 0 --threads 1
$PLASS concatdbs $W/w/A $W/w/B $W/w/C -v 0 --threads 1
# a header DB whose DATA FILE is not in key order (what a multi-threaded writer without renumbering leaves; extractorfs itself renumbers):
# B_h's entries laid out in a seeded random order, index offsets to match — and the reference's concatenation with it as second DB
python3 - $W/w/B_h $W/w/Bs_h <<'PY'
import random, sys
src, dst = sys.argv[1], sys.argv[2]
data = open(src, 'rb').read()
ent = [tuple(int(x) for x in l.split()[:3]) for l in open(src + '.index', 'rb')]
order = list(range(len(ent))); random.Random(7).shuffle(order)
off = {}; pos = 0
with open(dst, 'wb') as f:
    for i in order:
        k, o, l = ent[i]; f.write(data[o:o + l]); off[k] = (pos, l); pos += l
with open(dst + '.index', 'wb') as f:
    for k, _, _ in ent:
        f.write(b'%d\t%d\t%d\n' % (k, off[k][0], off[k][1]))
open(dst + '.dbtype', 'wb').write(open(src + '.dbtype', 'rb').read())
PY
$PLASS concatdbs $W/w/A_h $W/w/Bs_h $W/w/Cs_h -v 0 --threads 1
for f in aaA aaB aaC A_h B_h C_h Bs_h Cs_h A B C; do cp $W/w/$f $W/w/$f.index $W/w/$f.dbtype $W/concat/; done
for f in aaA aaB A_h B_h Bs_h; do awk -v f=$f 'NR>1 && $2<prev {c++} {prev=$2} END {print f": offset inversions in key order:", c+0, "of", NR}' $W/concat/$f.index; done
printf 'aaA, aaB = plass translatenucs --add-orf-stop 1 --threads 8 of the two extractorfs passes on examples/reads_1.fastq.gz (raw files, thread order)\naaC = plass concatdbs aaA aaB\nA, B, A_h, B_h = the nucleotide ORFs and their header DBs as the two extractorfs passes (--threads 8) left them; C = concatdbs A B, C_h = concatdbs A_h B_h\nBs_h = B_h with its data file in a seeded random order (index offsets to match); Cs_h = plass concatdbs A_h Bs_h\n' > $W/concat/MANIFEST
tar -C $W -czf $HERE/concat_noncanonical.tar.gz concat
rm -rf $W
