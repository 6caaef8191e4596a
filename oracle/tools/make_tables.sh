#!/bin/bash
# GENERATION-TIME ONLY (build container). Regenerates oracle/ref_tables.h from the unmodified
# reference. Needs the survey-time out-of-tree build of /root/reference:
#   cmake -DCMAKE_BUILD_TYPE=Release -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DHAVE_MPI=0 \
#         -DHAVE_TESTS=0 -DVERSION_OVERRIDE=survey /root/reference && make -j8 plass penguin
# in ${REF_BUILD:-/tmp/plass-build}. Nothing from that build is copied into the repo;
# only the printed constant tables (data) are.
set -euo pipefail
REF=/root/reference
B=${REF_BUILD:-/tmp/plass-build}
HERE=$(cd "$(dirname "$0")" && pwd)
M=$REF/lib/mmseqs
INC="-I$REF/lib -I$REF/src/commons -I$M/lib/zstd/lib -I$M/lib/tinyexpr -I$M/lib/microtar -I$M/lib/simde -I$M/lib -I$M/lib/simd -I$M/lib/gzstream -I$M/lib/alp -I$M/lib/cacode -I$M/lib/ksw2 -I$M/lib/xxhash -I$M/lib/ips4o -I$B/generated -I$B/lib/mmseqs/generated -I$M/src/alignment -I$M/src/clustering -I$M/src/commons -I$M/src/multihit -I$M/src/prefiltering -I$M/src/linclust -I$M/src/taxonomy -I$M/src/util -I$M/src"
OUT=${TMPDIR:-/tmp}/dump_ref_tables
g++ -O3 -DNDEBUG -w -fsigned-char -march=native -std=c++1y -fopenmp -DENABLE_IPS4O=1 -DHAVE_ZLIB=1 -DOPENMP=1 $INC \
    "$HERE/dump_ref_tables.cpp" -o "$OUT" \
    $B/lib/mmseqs/src/libmmseqs-framework.a $B/src/version/libversion.a $B/lib/mmseqs/src/libmmseqs-framework.a \
    -latomic $B/lib/mmseqs/lib/tinyexpr/libtinyexpr.a -lm $B/lib/mmseqs/lib/zstd/build/cmake/lib/libzstd.a \
    $B/lib/mmseqs/lib/microtar/libmicrotar.a -lz
"$OUT" dump | grep -v "^Reduced amino acid alphabet\|^Time for processing" > "$HERE/../ref_tables.h"
echo "wrote $HERE/../ref_tables.h"
# The product keeps its own copy of the same DATA (it must not include anything under oracle/).
sed -e 's/REF_/PH_/g' -e '1s|.*|/* GENERATED DATA (oracle/tools/make_tables.sh): constant tables of the reference matrices. */|' \
    "$HERE/../ref_tables.h" > "$HERE/../../plass_amd/csrc/tables_data.h"
echo "wrote plass_amd/csrc/tables_data.h"
