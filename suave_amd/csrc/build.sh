#!/bin/bash
# Build libsuave_hip.so for gfx950 in-tree (the .so travels to the GPU box; it is git-ignored).
set -e
cd "$(dirname "$0")"
OUT=../libsuave_hip.so
OBJ=../../build/obj
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $SUAVE_EXTRA_FLAGS"
pids=()
for f in gemm decode loss wide embed latent optim sample frame head api; do
  [ -f $f.hip ] || continue
  stale=0  # an object is rebuilt when its source, ANY header here or under include/, or this script is newer
  for dep in $f.hip *.h ../../include/*.h build.sh; do
    if [ ! -f $OBJ/$f.o ] || [ $dep -nt $OBJ/$f.o ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    extra=""
    # deterministic binary32 paths (detmath.h): no FMA contraction, correctly rounded sqrt / divide
    if [ $f = sample ] || [ $f = latent ] || [ $f = decode ]; then extra="-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"; fi
    hipcc $FLAGS $extra -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in gemm decode loss wide embed latent optim sample frame head api; do [ -f $OBJ/$f.o ] && objs="$objs $OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $(realpath $OUT)"
