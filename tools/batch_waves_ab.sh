#!/bin/bash
# gab_conv_process_batch on the split cut: the eight-wave launch (conv_split_batch_kernel, two waves per SIMD) against the
# twelve-wave launch (conv_split_batch12_kernel, three per SIMD) on ONE box, alternating (diagnostic build, GAB_BATCH_WAVES):
#   bash tools/batch_waves_ab.sh [tag] [channels ...] -> gpurun_out/batch_waves_ab_<tag>.txt
TAG=${1:-r06}; shift
CH=${@:-1024}
OUT=$PWD/gpurun_out/batch_waves_ab_$TAG.txt
export GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so NBUF=${NBUF:-128}
: > $OUT
for rep in 1 2 3; do
  for W in ${WAVES:-8 12 6}; do
    export GAB_BATCH_WAVES=$W
    echo "waves $W (rep $rep)" >> $OUT
    timeout -k 10 200 python3 tools/batch_conv.py $CH >> $OUT 2>&1 || { echo "FAILED rc=$?" >> $OUT; exit 1; }
  done
done
cat $OUT
