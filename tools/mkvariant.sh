#!/bin/bash
# Build a variant of libmrchip.so for same-box A/B runs (tools/abn.sh): the objects of the current build are reused, the
# listed sources are recompiled with the extra flags.
# Usage: bash tools/mkvariant.sh <V> "<extra hipcc flags>" k_sauvola [k_optimise ...]
set -e
V=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/archive-pdf-tools_amd/lib
mkdir -p $L/abv/$V $L/ab
rm -rf $L/abv/$V/obj; cp -rp $L/obj $L/abv/$V/obj
for s in "$@"; do rm -f $L/abv/$V/obj/$s.o; done
make -s -j8 -C $R/archive-pdf-tools_amd/csrc OUT=../lib/abv/$V EXTRA="$FLAGS" ../lib/abv/$V/libmrchip.so
cp $L/abv/$V/libmrchip.so $L/ab/libmrchip_$V.so
echo built $L/ab/libmrchip_$V.so
