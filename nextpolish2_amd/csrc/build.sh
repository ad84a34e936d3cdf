#!/bin/bash
# Builds libnp2_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
ARCH=${NP2_ARCH:-gfx950}
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p obj
for f in np2_kernels.hip np2_graph.hip np2_cand.hip np2_regions.hip np2_front.hip np2_prims.hip np2_host.cpp np2_io.cpp; do
  o=obj/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ np2_kernels.hpp -nt "$o" ] || [ np2_common.hpp -nt "$o" ] || [ np2_phase_host.hpp -nt "$o" ] || [ ../../include/np2.h -nt "$o" ] || [ ../../include/np2_io.h -nt "$o" ] || [ np2_ctx.hpp -nt "$o" ] || [ np2_blockscan.hpp -nt "$o" ]; then
    echo "hipcc $f"
    hipcc $FLAGS -x hip -c $f -o $o
  fi
done
hipcc --offload-arch=$ARCH -shared -fPIC -o ../libnp2_hip.so obj/np2_kernels.o obj/np2_graph.o obj/np2_cand.o obj/np2_regions.o obj/np2_front.o obj/np2_prims.o obj/np2_host.o obj/np2_io.o -lz -lpthread
echo "built nextpolish2_amd/libnp2_hip.so"
