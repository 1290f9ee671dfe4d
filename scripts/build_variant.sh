#!/bin/bash
# development: build_variant.sh NAME 'sed-expr on kge_rank_screen.h' -> build_variants/NAME/libamdkge.so (only kge_rank.o differs)
set -e
N=$1; R=/root/repo; D=/tmp/variant_$N/ampligraph_amd/csrc; mkdir -p $D /tmp/variant_$N/include $R/build_variants/$N
cp $R/ampligraph_amd/csrc/*.h $R/ampligraph_amd/csrc/kge_rank.hip $D/; cp $R/include/amdkge.h /tmp/variant_$N/include/
[ -n "${2:-}" ] && sed -i "$2" $D/kge_rank_screen.h
cd $D && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function ${EXTRA:-} -c kge_rank.hip -o /tmp/variant_$N/kge_rank.o
OBJS=$(ls $R/build/obj/*.o | grep -v kge_rank.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$N/kge_rank.o -ldl -o $R/build_variants/$N/libamdkge.so
ls -la $R/build_variants/$N/libamdkge.so
