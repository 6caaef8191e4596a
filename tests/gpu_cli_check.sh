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
tar -C "$W" -xzf "$ROOT/tests/golden/orfs.tar.gz"
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
# row N2: the once-per-run preprocessing (data/assemble.sh:41-77) on the reference's hostile reads and on the example's reads
n=0
while read -r FL; do
  n=$((n+1)); i=$(echo "1 2 4 5" | cut -d' ' -f$n)
  $P extractorfs $W/orf/in $W/o_orfs$i $FL --threads 4 | tail -1 || echo "extractorfs rc=$?"
  check $W/orf/orfs_$i $W/o_orfs$i "extractorfs flag set $i"
  check $W/orf/orfs_${i}_h $W/o_orfs${i}_h "extractorfs headers flag set $i"
  $P translatenucs $W/o_orfs$i $W/o_aa$i --translation-table 1 --add-orf-stop 1 | tail -1 || echo "translatenucs rc=$?"
  check $W/orf/aa_stop_$i $W/o_aa$i "translatenucs --add-orf-stop 1 flag set $i"
done < $W/orf/FLAGS
CM="--max-gaps 0 --orf-start-mode 0 --forward-frames 1,2,3 --reverse-frames 1,2,3 --translation-table 1 --translate 0 --use-all-table-starts 0"
$P extractorfs $W/nucl/seq_0 $W/nucl_6f_start --min-length 20 --max-length 45 --contig-start-mode 1 --contig-end-mode 0 $CM | tail -1
$P extractorfs $W/nucl/seq_0 $W/nucl_6f_long --min-length 45 --max-length 32734 --contig-start-mode 2 --contig-end-mode 2 $CM | tail -1
$P translatenucs $W/nucl_6f_start $W/aa_6f_start --translation-table 1 --add-orf-stop | tail -1
$P translatenucs $W/nucl_6f_long $W/aa_6f_long --translation-table 1 --add-orf-stop 1 | tail -1
$P concatdbs $W/aa_6f_long $W/aa_6f_start $W/aa_6f_start_long -v 3 | tail -1
check $W/aa/seq_0 $W/aa_6f_start_long "extractorfs x2 + translatenucs x2 + concatdbs = iteration-0 input"
$P concatdbs $W/nucl_6f_long_h $W/nucl_6f_start_h $W/aa_6f_start_long_h | tail -1 || echo "concatdbs (headers) rc=$?"
# the drop-in CLI refuses what it does not know instead of ignoring it
if $P kmermatcher $W/aa/seq_0 $W/o_bad -k 14 --kmer-per-seq 60 --no-such-flag 1 > $W/bad.log 2>&1; then echo "FAIL unknown flag accepted"; fails=$((fails+1)); else echo "PASS unknown flag refused"; fi
if $P rescorediagonal $W/aa/seq_0 $W/aa/seq_0 $W/aa/pref_0 $W/o_bad --rescore-mode 3 --sub-mat aa:VTML80.out,nucl:nucleotide.out > $W/bad.log 2>&1; then echo "FAIL foreign matrix accepted"; fails=$((fails+1)); else echo "PASS foreign substitution matrix refused"; fi
echo "failures: $fails"
exit $fails
