#!/bin/bash
# A measurement build of libqv.so with extra -D switches on ONE kernel file:  tools/build_variant.sh <name> <file.hip> <flags...>
# -> quiver_amd/lib/libqv_<name>.so (use with QV_LIB_PATH=...).  The other objects are the product build's.
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../quiver_amd/csrc"
base=${file%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c -o ../lib/obj/${base}_$name.o $file
objs=$(ls ../lib/obj/*.o | grep -v "_[a-z0-9]*\.o$" | grep -v "${base}.o" | tr '\n' ' ')
objs=$(for f in qv_scan qv_mq64 qv_rank qv_select qv_batched qv_qreg qv_hnsw qv_build qv_misc qv_api qv_graph_api qv_sharded_api; do if [ "$f" = "$base" ]; then echo ../lib/obj/${base}_$name.o; else echo ../lib/obj/$f.o; fi; done | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libqv_$name.so $objs -L/opt/rocm/lib -lrccl
echo built quiver_amd/lib/libqv_$name.so
