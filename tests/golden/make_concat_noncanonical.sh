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
for f in aaA aaB aaC; do cp $W/w/$f $W/w/$f.index $W/w/$f.dbtype $W/concat/; done
for f in aaA aaB; do awk -v f=$f 'NR>1 && $2<prev {c++} {prev=$2} END {print f": offset inversions in key order:", c+0, "of", NR}' $W/concat/$f.index; done
printf 'aaA, aaB = plass translatenucs --add-orf-stop 1 --threads 8 of the two extractorfs passes on examples/reads_1.fastq.gz (raw files, thread order)\naaC = plass concatdbs aaA aaB\n' > $W/concat/MANIFEST
tar -C $W -czf $HERE/concat_noncanonical.tar.gz concat
rm -rf $W
