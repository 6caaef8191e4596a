#!/bin/bash
# GENERATION-TIME ONLY (build container): tests/golden/concat_preserve.tar.gz — `concatdbs --preserve-keys` as data/nuclassemble.sh:19-61,145
# uses it, made by the UNMODIFIED reference (REF_BUILD, default /tmp/plass-build): the crafted contigs of cyclecheck.tar.gz (`in`) and the
# circular ones the reference's own cyclecheck found among them (`cycle_chop1`) give
#   noneCycle          the workflow's "<assembly>_noneCycle": an INDEX SUBSET over the assembly's data file (awk over the two indices, symlinked data)
#   cycA, cycB         two disjoint halves of the cycle DB (index subsets as well): "the cycles of an earlier iteration" and "of this one"
#   cycle_all          penguin concatdbs cycA cycB cycle_all --preserve-keys          (nuclassemble.sh:41)
#   merged             penguin concatdbs noneCycle cycle_all merged --preserve-keys   (nuclassemble.sh:145)
# raw files as the reference's single-threaded writer leaves them.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); B=${REF_BUILD:-/tmp/plass-build}; PENGUIN=$B/src/penguin
W=$(mktemp -d); mkdir $W/w $W/concat_preserve
tar -C $W/w -xzf $HERE/cyclecheck.tar.gz
S=$W/w/cyc; O=$W/concat_preserve

awk 'NR==FNR { a[$1]=$0; next } !($1 in a) {print $0}' $S/cycle_chop1.index $S/in.index > $O/noneCycle.index
cp $S/in $O/noneCycle; cp $S/in.dbtype $O/noneCycle.dbtype
awk 'NR%2==1' $S/cycle_chop1.index > $O/cycA.index; awk 'NR%2==0' $S/cycle_chop1.index > $O/cycB.index
for x in cycA cycB; do cp $S/cycle_chop1 $O/$x; cp $S/cycle_chop1.dbtype $O/$x.dbtype; done
$PENGUIN concatdbs $O/cycA $O/cycB $O/cycle_all --preserve-keys --threads 1 -v 0
$PENGUIN concatdbs $O/noneCycle $O/cycle_all $O/merged --preserve-keys --threads 1 -v 0
printf 'noneCycle = index subset of the crafted contigs of cyclecheck.tar.gz (not circular per the reference cyclecheck --chop-cycle 1), data file = in\ncycA / cycB = odd / even index lines of the reference cycle DB (index subsets over its data file)\ncycle_all = penguin concatdbs cycA cycB --preserve-keys; merged = penguin concatdbs noneCycle cycle_all --preserve-keys (data/nuclassemble.sh:41,145)\n' > $O/MANIFEST
wc -l $O/*.index
tar -C $W -czf $HERE/concat_preserve.tar.gz concat_preserve
rm -rf $W
