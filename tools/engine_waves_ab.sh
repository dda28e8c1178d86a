#!/bin/bash
# The doorbell-fed engine: round 5's eight-wave period code against the twelve-wave code on ONE box, alternating (diagnostic build,
# GAB_ENGINE_WAVES): pipelined rate (tools/engine_conv.py) and ONE buffer in flight (tools/engine_latency.py).
#   bash tools/engine_waves_ab.sh [tag] -> gpurun_out/engine_waves_ab_<tag>.txt
TAG=${1:-r06}
OUT=$PWD/gpurun_out/engine_waves_ab_$TAG.txt
export GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so
: > $OUT
for rep in 1 2; do
  for W in 8 12; do
    export GAB_ENGINE_WAVES=$W
    echo "waves $W (rep $rep)" >> $OUT
    ENGINE_AHEADS="16 16 8" timeout -k 10 200 python3 tools/engine_conv.py 1024 63 2>&1 | grep -v amdgpu >> $OUT || { echo "FAILED" >> $OUT; exit 1; }
    timeout -k 10 200 python3 tools/engine_latency.py 1024 1000 2>&1 | grep -v amdgpu >> $OUT || { echo "FAILED" >> $OUT; exit 1; }
  done
done
cat $OUT
