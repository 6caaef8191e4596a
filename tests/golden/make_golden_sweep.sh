#!/bin/bash
# GENERATION-TIME ONLY (build container).  Flag sweep of the three protein modules on the bundled example: the UNMODIFIED
# reference binary (REF_BUILD, see make_golden.sh) is run once per variant; inputs are the golden DBs of example_aa.tar.gz
# (seq_0, pref_0, aln_0), outputs are canonicalised and stored with the flags that produced them (variants.tsv).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
B=${REF_BUILD:-/tmp/plass-build}
PLASS=$B/src/plass
W=$(mktemp -d /tmp/golden.XXXXXX)
CANON="python3 $REPO/tools/dbcanon.py"
Q="--threads 4 -v 1"
tar -C $W -xzf $HERE/example_aa.tar.gz
S=$W/aa; O=$W/sweep; mkdir -p $O
KM0="--alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0"
RS0="--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0"
AS0="--min-seq-id 0.9 --max-seq-len 65535 --keep-target 1 --rescore-mode 3"
: > $O/variants.tsv
km() { local name=$1; shift; $PLASS kmermatcher $S/seq_0 $W/o "$@" $Q > /dev/null; $CANON $W/o $O/$name; rm -f $W/o $W/o.*; printf "%s\tkmermatcher\t%s\n" "$name" "$*" >> $O/variants.tsv; }
rs() { local name=$1; shift; $PLASS rescorediagonal $S/seq_0 $S/seq_0 $S/pref_0 $W/o "$@" $Q > /dev/null; $CANON $W/o $O/$name; rm -f $W/o $W/o.*; printf "%s\trescorediagonal\t%s\n" "$name" "$*" >> $O/variants.tsv; }
as() { local name=$1; shift; $PLASS assembleresults $S/seq_0 $S/aln_0 $W/o "$@" $Q > /dev/null; $CANON $W/o $O/$name; rm -f $W/o $W/o.*; printf "%s\tassembleresults\t%s\n" "$name" "$*" >> $O/variants.tsv; }
# kmermatcher: alphabet 21 + shorter k, fewer / length-scaled k-mers per sequence, repeated k-mers kept, coverage filters, other seed
km km_alph21_k12    --alph-size 21 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 12 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0
km km_kps20         --alph-size 13 --kmer-per-seq 20 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0
km km_scale05       --alph-size 13 --kmer-per-seq 10 --kmer-per-seq-scale nucl:0.200,aa:0.500 -k 14 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 1
km km_multi0        --alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0 --cov-mode 0 --ignore-multi-kmer 0 --hash-shift 67 --include-only-extendable 0
km km_cov0_c08      --alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0.8 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0
km km_cov1_c09      --alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0.9 --cov-mode 1 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0
km km_cov2_c09      --alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 14 -c 0.9 --cov-mode 2 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0
km km_shift5_k10    --alph-size 13 --kmer-per-seq 60 --kmer-per-seq-scale nucl:0.200,aa:0.000 -k 10 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 5 --include-only-extendable 1
# rescorediagonal: loose E-value, identity / length thresholds, identity modes, coverage filters, self matches, backtrace
rs rs_e10           --rescore-mode 3 -e 10 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0
rs rs_id05_len30    --rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.5 --min-aln-len 30 --seq-id-mode 0
rs rs_idmode1       --rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.8 --min-aln-len 0 --seq-id-mode 1
rs rs_idmode2       --rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.6 --min-aln-len 0 --seq-id-mode 2
rs rs_cov0_c07      --rescore-mode 3 -e 1e-05 -c 0.7 -a 0 --cov-mode 0 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0
rs rs_cov1_c09      --rescore-mode 3 -e 1e-05 -c 0.9 -a 0 --cov-mode 1 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0
rs rs_cov2_c09      --rescore-mode 3 -e 1e-05 -c 0.9 -a 0 --cov-mode 2 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0
rs rs_self_bt       --rescore-mode 3 -e 1e-05 -c 0 -a 1 --cov-mode 0 --min-seq-id 0.9 --min-aln-len 0 --seq-id-mode 0 --add-self-matches 1
# assembleresults: low identity threshold, length cap, consumed targets dropped
as as_id05          --min-seq-id 0.5 --max-seq-len 65535 --keep-target 1 --rescore-mode 3
as as_cap100_keep0  --min-seq-id 0.9 --max-seq-len 100 --keep-target 0 --rescore-mode 3
# nucleotide and guided modules (inputs: example_nucl.tar.gz / example_guided.tar.gz); module names carry the input set
PENGUIN=$B/src/penguin
tar -C $W -xzf $HERE/example_nucl.tar.gz; tar -C $W -xzf $HERE/example_guided.tar.gz
N=$W/nucl; G=$W/guided
nkm() { local name=$1; shift; $PENGUIN kmermatcher $N/seq_0 $W/o "$@" $Q > /dev/null; $CANON $W/o $O/$name; rm -f $W/o $W/o.*; printf "%s\tnucl:kmermatcher\t%s\n" "$name" "$*" >> $O/variants.tsv; }
nrs() { local name=$1; shift; $PENGUIN rescorediagonal $N/seq_0 $N/seq_0 $N/pref_0 $W/o "$@" $Q > /dev/null; $CANON $W/o $O/$name; rm -f $W/o $W/o.*; printf "%s\tnucl:rescorediagonal\t%s\n" "$name" "$*" >> $O/variants.tsv; }
nas() { local name=$1; shift; $PENGUIN nuclassembleresults $N/seq_0 $N/aln_0 $W/o "$@" $Q > /dev/null; $CANON $W/o $O/$name; rm -f $W/o $W/o.*; printf "%s\tnucl:nuclassembleresults\t%s\n" "$name" "$*" >> $O/variants.tsv; }
gas() { local name=$1; shift; $PENGUIN guidedassembleresults $G/nucl_0 $G/aa_0 $G/aln_nucl_0 $W/o $W/o2 "$@" $Q > /dev/null; $CANON $W/o $O/$name; $CANON $W/o2 $O/${name}_aa; rm -f $W/o $W/o.* $W/o2 $W/o2.*; printf "%s\tguided:guidedassembleresults\t%s\n" "$name" "$*" >> $O/variants.tsv; }
nkm nkm_ext0_k15    --alph-size 5 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 15 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --hash-shift 67 --include-only-extendable 0
nkm nkm_scale03_c08 --alph-size 5 --kmer-per-seq 20 --kmer-per-seq-scale 0.300 -k 22 -c 0.8 --cov-mode 1 --ignore-multi-kmer 1 --hash-shift 3 --include-only-extendable 0
nkm nkm_multi0      --alph-size 5 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 22 -c 0 --cov-mode 0 --ignore-multi-kmer 0 --hash-shift 67 --include-only-extendable 1
nrs nrs_bt_self     --rescore-mode 3 -e 1e-05 -c 0 -a 1 --cov-mode 0 --min-seq-id 0.99 --min-aln-len 0 --seq-id-mode 0 --add-self-matches 1
nrs nrs_id09_cov2   --rescore-mode 3 -e 0.001 -c 0.5 -a 0 --cov-mode 2 --min-seq-id 0.9 --min-aln-len 40 --seq-id-mode 1
nas nas_id09_keep0  --min-seq-id 0.9 --max-seq-len 200000 --keep-target 0 --rescore-mode 3
nas nas_cap300      --min-seq-id 0.99 --max-seq-len 300 --keep-target 1 --rescore-mode 3
gas gas_id09_keep0  --min-seq-id 0.9 --max-seq-len 200000 --keep-target 0 --rescore-mode 3
gas gas_cap400      --min-seq-id 0.99 --max-seq-len 400 --keep-target 1 --rescore-mode 3
( cd $W && tar -czf $HERE/example_aa_sweep.tar.gz sweep )
ls -la $HERE/example_aa_sweep.tar.gz; cat $O/variants.tsv | cut -c1-60
rm -rf $W
