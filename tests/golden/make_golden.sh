#!/bin/bash
# GENERATION-TIME ONLY (build container). Produces the golden fixtures in this directory by running
# the UNMODIFIED reference binaries step by step and canonicalising every DB (key order).
#
# The reference binaries come from the survey-time out-of-tree build:
#   cmake -DCMAKE_BUILD_TYPE=Release -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DHAVE_MPI=0 \
#         -DHAVE_TESTS=0 -DVERSION_OVERRIDE=survey /root/reference && make -j8 plass penguin
# (REF_BUILD, default /tmp/plass-build).  The reference cannot be rebuilt with a plain g++ recipe
# (Parameters.cpp needs cmake-generated headers), so there is no oracle/_ref; these fixtures are the pin.
# Fixtures are DATA: inputs are the reference's bundled example reads turned into DBs by the
# reference's own (out-of-scope) preprocessing; expected outputs are what its hot modules wrote.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
B=${REF_BUILD:-/tmp/plass-build}
PLASS=$B/src/plass; PENGUIN=$B/src/penguin
EX=/root/reference/examples
W=$(mktemp -d /tmp/golden.XXXXXX)
CANON="python3 $REPO/tools/dbcanon.py"
Q="--threads 4 -v 1"

# ---------- protein example (config C1): preprocessing by the reference, then step-level modules ----
$PLASS assemble $EX/reads_1.fastq.gz $EX/reads_2.fastq.gz $W/out.fas $W/tmp --num-iterations 1 \
       --remove-tmp-files 0 --delete-tmp-inc 0 $Q > $W/aa.log
T=$(ls -d $W/tmp/[0-9]*/)
S=$W/aa; mkdir -p $S
$CANON ${T}aa_6f_start_long $S/seq_0
KM="--alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --max-seq-len 65535"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
AS="--min-seq-id 0.9 --max-seq-len 65535 --keep-target 1 --rescore-mode 3"
for i in 0 1 2; do
  if [ $i -eq 0 ]; then HS=67; EXT=0; else HS=68; EXT=1; fi
  $PLASS kmermatcher $S/seq_$i $W/p $KM --hash-shift $HS --include-only-extendable $EXT $Q >> $W/aa.log
  $PLASS rescorediagonal $S/seq_$i $S/seq_$i $W/p $W/a $RS $Q >> $W/aa.log
  $PLASS assembleresults $S/seq_$i $W/a $W/s $AS $Q >> $W/aa.log
  $CANON $W/p $S/pref_$i; $CANON $W/a $S/aln_$i; $CANON $W/s $S/seq_$((i+1))
  rm -f $W/p* $W/a.* $W/a $W/a_* $W/s $W/s.* 2>/dev/null || true
done
# variants: -a 1 backtrace, --rescore-mode 2 (local), --keep-target 0
$PLASS kmermatcher $S/seq_0 $W/p $KM --hash-shift 67 --include-only-extendable 0 $Q >> $W/aa.log
$PLASS rescorediagonal $S/seq_0 $S/seq_0 $W/p $W/a --rescore-mode 2 -e 1e-05 -c 0 -a 1 --cov-mode 0 --min-seq-id 0.9 $Q >> $W/aa.log
$CANON $W/a $S/aln_0_mode2_bt
$PLASS assembleresults $S/seq_0 $S/aln_0 $W/s --min-seq-id 0.9 --max-seq-len 65535 --keep-target 0 --rescore-mode 3 $Q >> $W/aa.log
$CANON $W/s $S/seq_1_keeptarget0
rm -f $W/p* $W/a.* $W/a $W/s $W/s.*
cat > $S/MANIFEST <<M
protein example: seq_0 = aa_6f_start_long of 'plass assemble examples/reads_{1,2}.fastq.gz'
iteration i: kmermatcher seq_i -> pref_i [$KM --hash-shift 67|68|68 --include-only-extendable 0|1|1]
             rescorediagonal seq_i seq_i pref_i -> aln_i [$RS]
             assembleresults seq_i aln_i -> seq_{i+1} [$AS]
aln_0_mode2_bt: rescorediagonal --rescore-mode 2 -a 1 on pref_0; seq_1_keeptarget0: assembleresults --keep-target 0
M
tar -C $W -czf $HERE/example_aa.tar.gz aa

# ---------- findassemblystart (row N3): iteration 0 of data/assemble.sh:110-150 on the protein example -----------
F=$W/fs; mkdir -p $F
$PLASS findassemblystart $S/seq_0 $S/aln_0 $W/corr $Q >> $W/aa.log
$CANON $W/corr $F/corrected_seqs
$PLASS kmermatcher $F/corrected_seqs $W/pc $KM --hash-shift 67 --include-only-extendable 0 $Q >> $W/aa.log
$PLASS rescorediagonal $F/corrected_seqs $F/corrected_seqs $W/pc $W/ac $RS $Q >> $W/aa.log
$PLASS assembleresults $F/corrected_seqs $W/ac $W/as0 $AS $Q >> $W/aa.log
$CANON $W/as0 $F/assembly_0
# (guided_corrected_seqs = findassemblystart on guided/aa_0 + guided/aln_0 is appended after the guided section below)

# ---------- nucleotide example (penguin nuclassemble stage, config C5's nucleotide path) -----------
$PENGUIN nuclassemble $EX/reads_1.fastq.gz $EX/reads_2.fastq.gz $W/outn.fas $W/tmpn --num-iterations 1 \
       --remove-tmp-files 0 --delete-tmp-inc 0 $Q > $W/nucl.log
T=$(ls -d $W/tmpn/[0-9]*/)
S=$W/nucl; mkdir -p $S
$CANON ${T}nucl_reads $S/seq_0
KM="--alph-size 5 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 22 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --max-seq-len 200000 --hash-shift 67 --include-only-extendable 1"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.99 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
AS="--min-seq-id 0.99 --max-seq-len 200000 --keep-target 1 --rescore-mode 3"
for i in 0 1; do
  $PENGUIN kmermatcher $S/seq_$i $W/p $KM $Q >> $W/nucl.log
  $PENGUIN rescorediagonal $S/seq_$i $S/seq_$i $W/p $W/a $RS $Q >> $W/nucl.log
  $PENGUIN nuclassembleresults $S/seq_$i $W/a $W/s $AS $Q >> $W/nucl.log
  $CANON $W/p $S/pref_$i; $CANON $W/a $S/aln_$i; $CANON $W/s $S/seq_$((i+1))
  rm -f $W/p* $W/a.* $W/a $W/s $W/s.* 2>/dev/null || true
done
cat > $S/MANIFEST <<M
nucleotide example: seq_0 = nucl_reads of 'penguin nuclassemble examples/reads_{1,2}.fastq.gz'
iteration i: kmermatcher seq_i -> pref_i [$KM]; rescorediagonal -> aln_i [$RS]; nuclassembleresults -> seq_{i+1} [$AS]
(no cyclecheck between iterations: out of scope, SURVEY.md §2)
M
tar -C $W -czf $HERE/example_nucl.tar.gz nucl

# ---------- guided example (penguin guided_nuclassemble stage: protein-guided nucleotide assembly, config C5) ----------
$PENGUIN guided_nuclassemble $EX/reads_1.fastq.gz $EX/reads_2.fastq.gz $W/outg.fas $W/tmpg --num-iterations 2 \
       --remove-tmp-files 0 --delete-tmp-inc 0 $Q > $W/guided.log
T=$(ls -d $W/tmpg/[0-9]*/)guidedassembly_tmp/
S=$W/guided; mkdir -p $S
$CANON ${T}nucl_6f_start_long $S/nucl_0; $CANON ${T}aa_6f_start_long $S/aa_0
KM="--alph-size nucl:5,aa:13 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 14 -c 0 --cov-mode 1 --ignore-multi-kmer 1 --max-seq-len 200000 --hash-shift 67 --include-only-extendable 1"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 1 --cov-mode 1 --min-seq-id 0.97 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
P2N="--gap-open 5 --gap-extend 2"
AS="--min-seq-id 0.99 --max-seq-len 200000 --keep-target 1 --rescore-mode 3"
for i in 0 1; do
  $PENGUIN kmermatcher $S/aa_$i $W/p $KM $Q >> $W/guided.log
  $PENGUIN rescorediagonal $S/aa_$i $S/aa_$i $W/p $W/a $RS $Q >> $W/guided.log
  $PENGUIN proteinaln2nucl $S/nucl_$i $S/nucl_$i $S/aa_$i $S/aa_$i $W/a $W/an $P2N $Q >> $W/guided.log
  $PENGUIN guidedassembleresults $S/nucl_$i $S/aa_$i $W/an $W/sn $W/sa $AS $Q >> $W/guided.log
  if [ $i -eq 0 ]; then $CANON $W/p $S/pref_$i; $CANON $W/a $S/aln_$i; $CANON $W/an $S/aln_nucl_$i; fi   # iteration 1: outputs only (size)
  $CANON $W/sn $S/nucl_$((i+1)); $CANON $W/sa $S/aa_$((i+1))
  rm -f $W/p* $W/a.* $W/a $W/an $W/an.* $W/sn $W/sn.* $W/sa $W/sa.* 2>/dev/null || true
done
cat > $S/MANIFEST <<M
guided example: nucl_0 / aa_0 = nucl_6f_start_long / aa_6f_start_long of 'penguin guided_nuclassemble examples/reads_{1,2}.fastq.gz'
iteration i: kmermatcher aa_i -> pref_i [$KM]; rescorediagonal aa_i aa_i pref_i -> aln_i [$RS];
             proteinaln2nucl nucl_i nucl_i aa_i aa_i aln_i -> aln_nucl_i [$P2N];
             guidedassembleresults nucl_i aa_i aln_nucl_i -> nucl_{i+1} aa_{i+1} [$AS]
(pref / aln / aln_nucl kept for iteration 0 only)
M
tar -C $W -czf $HERE/example_guided.tar.gz guided

# ---------- long nucleotide contigs (KmerPosition<int>, 16-bit diagonal wrap-around: a contig grows past 65 535 nt) ----------
S=$W/longnucl; mkdir -p $S
python3 - "$S/seq_0" "$REPO" <<'PY'
import sys
sys.path.insert(0, sys.argv[2])
import numpy as np
from plass_amd import synth
rng = np.random.default_rng(77)
B = np.frombuffer(b"ACGT", dtype=np.uint8)
g = rng.integers(0, 4, size=70000, dtype=np.uint8)
rc = lambda x: (3 - x)[::-1]
seqs = [g[0:36000], g[35200:52000], rc(g[30000:47000]), g[51000:70000], rc(g[60000:69000])]
for i in range(400):
    p = int(rng.integers(0, 69800)); r = g[p:p + 150].copy()
    if rng.random() < 0.5: r = rc(r)
    m = rng.random(150) < 0.003; r[m] = rng.integers(0, 4, size=int(m.sum()))
    seqs.append(r)
synth.write_db(sys.argv[1], *synth.pack_db([B[s] for s in seqs]), 1)
PY
KM="--alph-size 5 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 22 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --max-seq-len 200000 --hash-shift 67 --include-only-extendable 1"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.99 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
AS="--min-seq-id 0.99 --max-seq-len 200000 --keep-target 1 --rescore-mode 3"
for i in 0 1; do
  $PENGUIN kmermatcher $S/seq_$i $W/p $KM $Q >> $W/nucl.log
  $PENGUIN rescorediagonal $S/seq_$i $S/seq_$i $W/p $W/a $RS $Q >> $W/nucl.log
  $PENGUIN nuclassembleresults $S/seq_$i $W/a $W/s $AS $Q >> $W/nucl.log
  $CANON $W/p $S/pref_$i; $CANON $W/a $S/aln_$i; $CANON $W/s $S/seq_$((i+1))
  rm -f $W/p* $W/a.* $W/a $W/s $W/s.* 2>/dev/null || true
done
tar -C $W -czf $HERE/long_nucl.tar.gz longnucl
# ---------- findassemblystart on the translated ORFs of the guided example (alignment lines with backtrace) ----------
$PLASS findassemblystart $W/guided/aa_0 $W/guided/aln_0 $W/gcorr $Q >> $W/aa.log
$CANON $W/gcorr $W/fs/guided_corrected_seqs
printf 'corrected_seqs = plass findassemblystart aa/seq_0 aa/aln_0; assembly_0 = kmermatcher + rescorediagonal + assembleresults on it\nguided_corrected_seqs = plass findassemblystart guided/aa_0 guided/aln_0\n' > $W/fs/MANIFEST
tar -C $W -czf $HERE/findstart.tar.gz fs
# ---------- cyclecheck (row N4): crafted circular / linear contigs + the example's contigs ----------
C=$W/cyc; mkdir -p $C
python3 $HERE/make_cyclecheck.py $C/in
for c in 0 1; do
  $PENGUIN cyclecheck $C/in $W/cy --max-seq-len 50000 --chop-cycle $c $Q > /dev/null; $CANON $W/cy $C/cycle_chop$c; rm -f $W/cy $W/cy.*
done
for d in nucl/seq_2 longnucl/seq_0 longnucl/seq_2; do
  $PENGUIN cyclecheck $W/$d $W/cy --max-seq-len 200000 --chop-cycle 1 $Q > /dev/null; $CANON $W/cy $C/$(echo $d | tr / _)_cycle; rm -f $W/cy $W/cy.*
done
printf 'in = make_cyclecheck.py; cycle_chop0|1 = penguin cyclecheck in --max-seq-len 50000 --chop-cycle 0|1; <db>_cycle = penguin cyclecheck <db> --max-seq-len 200000 --chop-cycle 1\n' > $C/MANIFEST
tar -C $W -czf $HERE/cyclecheck.tar.gz cyc
# ---------- extractorfs / translatenucs (row N2): hostile reads, the two workflow flag sets and two odd ones ----------
R=$W/orf; mkdir -p $R
python3 $HERE/make_orfs_input.py $R/in
cat > $R/FLAGS <<F
--min-length 20 --max-length 45 --max-gaps 0 --contig-start-mode 1 --contig-end-mode 0 --orf-start-mode 0 --forward-frames 1,2,3 --reverse-frames 1,2,3 --translation-table 1 --translate 0 --use-all-table-starts 0
--min-length 45 --max-length 32734 --max-gaps 0 --contig-start-mode 2 --contig-end-mode 2 --orf-start-mode 0 --forward-frames 1,2,3 --reverse-frames 1,2,3 --translation-table 1 --translate 0
--min-length 5 --max-length 100 --max-gaps 3 --contig-start-mode 2 --contig-end-mode 2 --orf-start-mode 2 --forward-frames 1,3 --reverse-frames 2 --translation-table 1 --translate 0
--min-length 10 --max-length 32734 --max-gaps 1 --contig-start-mode 0 --contig-end-mode 1 --orf-start-mode 0 --forward-frames 1,2,3 --reverse-frames 1,2,3 --translation-table 1 --translate 0
F
n=0
while read -r FL; do
  n=$((n+1)); i=$(echo "1 2 4 5" | cut -d" " -f$n)
  $PLASS extractorfs $R/in $W/ro $FL $Q > /dev/null; $CANON $W/ro $R/orfs_$i; $CANON $W/ro_h $R/orfs_${i}_h
  $PLASS translatenucs $W/ro $W/ra --translation-table 1 --add-orf-stop 1 $Q > /dev/null; $CANON $W/ra $R/aa_stop_$i
  $PLASS translatenucs $W/ro $W/rb --translation-table 1 --add-orf-stop 0 $Q > /dev/null; $CANON $W/rb $R/aa_$i
  rm -f $W/ro* $W/ra* $W/rb*
done < $R/FLAGS
# a small second input (first 120 entries) for the any-to-stop start mode and extractorfs' own --translate 1
python3 - "$R" <<'PY'
import sys
r = sys.argv[1]
for suf in ("", "_h"):
    lines = open(r + "/in" + suf + ".index").read().splitlines()[:120]
    data = open(r + "/in" + suf, "rb").read()
    end = int(lines[-1].split()[1]) + int(lines[-1].split()[2])
    open(r + "/in_small" + suf, "wb").write(data[:end]); open(r + "/in_small" + suf + ".index", "w").write("\n".join(lines) + "\n")
    open(r + "/in_small" + suf + ".dbtype", "wb").write(open(r + "/in" + suf + ".dbtype", "rb").read())
PY
ANY="--min-length 1 --max-length 32734 --max-gaps 2147483647 --contig-start-mode 2 --contig-end-mode 2 --orf-start-mode 1 --forward-frames 1,2,3 --reverse-frames 1,2,3 --translation-table 1"
printf '%s --translate 0\n%s --translate 1\n' "$ANY" "$ANY" > $R/FLAGS_SMALL
$PLASS extractorfs $R/in_small $W/ro $ANY --translate 0 $Q > /dev/null; $CANON $W/ro $R/small_orfs_any; $CANON $W/ro_h $R/small_orfs_any_h; rm -f $W/ro*
$PLASS extractorfs $R/in_small $W/ro $ANY --translate 1 $Q > /dev/null; $CANON $W/ro $R/small_orfs_translated; $CANON $W/ro_h $R/small_orfs_translated_h; rm -f $W/ro*
rm -f $R/in_small_h $R/in_small_h.index $R/in_small_h.dbtype
rm -f $R/in_h $R/in_h.index $R/in_h.dbtype
printf 'in = make_orfs_input.py; orfs_<i>, orfs_<i>_h = plass extractorfs in <out> <line of FLAGS>; aa_stop_<i> / aa_<i> = plass translatenucs --add-orf-stop 1 / 0\n' > $R/MANIFEST
tar -C $W -czf $HERE/orfs.tar.gz orf
ls -la $HERE/*.tar.gz
rm -rf $W
