#!/bin/bash
# GPU-box check through the command-line boundary: run every module of plass-hip on the golden
# inputs and compare every output DB with what the unmodified reference wrote (tests/golden/*.tar.gz).
# Usage: tests/gpu_cli_check.sh [workdir]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/plasship_cli_check}
rm -rf "$W"; mkdir -p "$W"
tar -C "$W" -xzf "$ROOT/tests/golden/example_aa.tar.gz"
tar -C "$W" -xzf "$ROOT/tests/golden/example_nucl.tar.gz"
tar -C "$W" -xzf "$ROOT/tests/golden/example_guided.tar.gz"
tar -C "$W" -xzf "$ROOT/tests/golden/findstart.tar.gz"
tar -C "$W" -xzf "$ROOT/tests/golden/cyclecheck.tar.gz"
P="timeout 300 $ROOT/plass_amd/plass-hip"
D="python3 $ROOT/tools/dbdiff.py"
fails=0
check() { if $D "$1" "$2" > "$W/diff.log" 2>&1; then echo "PASS $3"; else echo "FAIL $3"; head -12 "$W/diff.log"; fails=$((fails+1)); fi; }
S=$W/aa
KM="--alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --max-seq-len 65535"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
AS="--min-seq-id 0.9 --max-seq-len 65535 --keep-target 1 --rescore-mode 3"
for i in 0 1 2; do
  if [ $i -eq 0 ]; then HS=67; EXT=0; else HS=68; EXT=1; fi
  $P kmermatcher $S/seq_$i $W/o_pref $KM --hash-shift $HS --include-only-extendable $EXT | tail -2 || echo "kmermatcher rc=$?"
  check $S/pref_$i $W/o_pref "aa kmermatcher it$i"
  $P rescorediagonal $S/seq_$i $S/seq_$i $S/pref_$i $W/o_aln $RS | tail -2 || echo "rescorediagonal rc=$?"
  check $S/aln_$i $W/o_aln "aa rescorediagonal it$i"
  $P assembleresults $S/seq_$i $S/aln_$i $W/o_seq $AS | tail -2 || echo "assembleresults rc=$?"
  check $S/seq_$((i+1)) $W/o_seq "aa assembleresults it$i"
done
$P assembleresults $S/seq_0 $S/aln_0 $W/o_seq --min-seq-id 0.9 --max-seq-len 65535 --keep-target 0 --rescore-mode 3 | tail -1
check $S/seq_1_keeptarget0 $W/o_seq "aa assembleresults keep-target 0"
S=$W/nucl
KM="--alph-size 5 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 22 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --max-seq-len 200000 --hash-shift 67 --include-only-extendable 1"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.99 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
for i in 0 1; do
  $P kmermatcher $S/seq_$i $W/o_pref $KM | tail -2 || echo "kmermatcher rc=$?"
  check $S/pref_$i $W/o_pref "nucl kmermatcher it$i"
  $P rescorediagonal $S/seq_$i $S/seq_$i $S/pref_$i $W/o_aln $RS | tail -2 || echo "rescorediagonal rc=$?"
  check $S/aln_$i $W/o_aln "nucl rescorediagonal it$i"
  $P nuclassembleresults $S/seq_$i $S/aln_$i $W/o_seq --min-seq-id 0.99 --max-seq-len 200000 --keep-target 1 --rescore-mode 3 | tail -2 || echo "nuclassembleresults rc=$?"
  check $S/seq_$((i+1)) $W/o_seq "nucl nuclassembleresults it$i"
done
# penguin's protein-guided stage: the reference passes "--kmer-per-seq-scale 0.100 --alph-size nucl:5,aa:13 --gap-open 5 --gap-extend 2"
S=$W/guided
KM="--alph-size nucl:5,aa:13 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 14 -c 0 --cov-mode 1 --ignore-multi-kmer 1 --max-seq-len 200000 --hash-shift 67 --include-only-extendable 1"
RS="--rescore-mode 3 -e 1e-05 -c 0 -a 1 --cov-mode 1 --min-seq-id 0.97 --min-aln-len 0 --seq-id-mode 0 --sort-results 0"
$P kmermatcher $S/aa_0 $W/o_pref $KM | tail -2 || echo "kmermatcher rc=$?"
check $S/pref_0 $W/o_pref "guided kmermatcher"
$P rescorediagonal $S/aa_0 $S/aa_0 $S/pref_0 $W/o_aln $RS | tail -2 || echo "rescorediagonal rc=$?"
check $S/aln_0 $W/o_aln "guided rescorediagonal -a 1"
$P proteinaln2nucl $S/nucl_0 $S/nucl_0 $S/aa_0 $S/aa_0 $S/aln_0 $W/o_aln_nucl --gap-open 5 --gap-extend 2 | tail -2 || echo "proteinaln2nucl rc=$?"
check $S/aln_nucl_0 $W/o_aln_nucl "proteinaln2nucl"
$P guidedassembleresults $S/nucl_0 $S/aa_0 $S/aln_nucl_0 $W/o_nucl $W/o_aa --min-seq-id 0.99 --max-seq-len 200000 --keep-target 1 --rescore-mode 3 | tail -2 || echo "guidedassembleresults rc=$?"
check $S/nucl_1 $W/o_nucl "guidedassembleresults nucl"
check $S/aa_1 $W/o_aa "guidedassembleresults aa"
# iteration 0 of plass assemble runs findassemblystart between two kmermatcher / rescorediagonal passes (data/assemble.sh:110-141)
$P findassemblystart $W/aa/seq_0 $W/aa/aln_0 $W/o_corr --threads 4 | tail -2 || echo "findassemblystart rc=$?"
check $W/fs/corrected_seqs $W/o_corr "findassemblystart"
$P findassemblystart $W/guided/aa_0 $W/guided/aln_0 $W/o_gcorr --threads 4 | tail -2 || echo "findassemblystart rc=$?"
check $W/fs/guided_corrected_seqs $W/o_gcorr "findassemblystart (ORFs, alignments with backtrace)"
# penguin runs cyclecheck after every nuclassembleresults (data/nuclassemble.sh:19-61,132)
for c in 0 1; do
  $P cyclecheck $W/cyc/in $W/o_cyc$c --max-seq-len 50000 --chop-cycle $c --threads 4 | tail -2 || echo "cyclecheck rc=$?"
  check $W/cyc/cycle_chop$c $W/o_cyc$c "cyclecheck --chop-cycle $c"
done
echo "failures: $fails"
exit $fails
