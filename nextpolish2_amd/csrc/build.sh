#!/bin/bash
# Builds libnp2_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.  Translation units are compiled in parallel;
# an object is rebuilt when its source, any header next to it, the C ABI headers or this script changed.
set -e
cd "$(dirname "$0")"
ARCH=${NP2_ARCH:-gfx950}
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-omit-frame-pointer ${NP2_EXTRA_FLAGS:-}"
SRCS="np2_dense.hip np2_kernels.hip np2_graph.hip np2_passfront.hip np2_cand.hip np2_regions.hip np2_front.hip np2_inflate.hip np2_prims.hip np2_host.cpp np2_io.cpp np2_batch.cpp"
mkdir -p obj
pids=()
objs=()
for f in $SRCS; do
  [ -f "$f" ] || continue
  o=obj/${f%.*}.o
  objs+=("$o")
  stale=0
  [ -f "$o" ] || stale=1
  for d in "$f" *.hpp ../../include/*.h build.sh; do [ "$d" -nt "$o" ] && stale=1; done
  if [ $stale = 1 ]; then
    echo "hipcc $f"
    hipcc $FLAGS -x hip -c "$f" -o "$o" &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
[ $rc = 0 ] || { echo "build failed"; exit 1; }
hipcc --offload-arch=$ARCH -shared -fPIC -o ../libnp2_hip.so "${objs[@]}" -lz -lpthread -rdynamic
echo "built nextpolish2_amd/libnp2_hip.so"
