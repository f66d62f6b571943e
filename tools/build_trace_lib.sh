#!/bin/bash
# Trace build of the library (-DWS_TRACE: s_memtime stamps in the kernels that have them; every symbol visible so that
# the tools can reach the stamp buffers through their mangled accessors) -> tools/bin/libws_trace.so.
# Runs here (hipcc cross-compiles); tools/bin/ is git-ignored but travels with gpurun.
cd "$(dirname "$0")/.."
REPO=$(pwd); OBJ=/tmp/ws_trace_obj; mkdir -p $OBJ tools/bin
pids=()
for f in wespeaker_amd/csrc/*.hip; do
  o=$OBJ/$(basename $f).o
  if [ ! -s $o ] || [ $f -nt $o ] || [ wespeaker_amd/csrc/kernels.h -nt $o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWS_TRACE -Wno-unused-function -I include -c $f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libws_trace.so $OBJ/*.o && ls -la tools/bin/libws_trace.so
