#!/bin/bash
# Round-2 profile set (run through gpurun): bash tools/profile_r02.sh <tag>  -> gpurun_out/prof_<tag>/
# rocprofv3 kernel-trace statistics of bench.py's timed region (two range streams, and one launch per
# buffer), PMC traffic passes (one counter group per run, kernel-trace only), secondary kernels, FDTD.
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--steps 150 --warmup 15 --no-cpu-baseline --no-side-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench_s2 -- python3 bench.py $B --streams 2 > $OUT/bench_s2_line.json 2> $OUT/bench_s2.err
python3 tools/trace_period.py $OUT/bench_s2_kernel_trace.csv --kernel conv_split_range --last 6400 --per-buffer 2 > $OUT/bench_s2_trace_period.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench_s1 -- python3 bench.py $B --streams 1 > $OUT/bench_s1_line.json 2> $OUT/bench_s1.err
python3 tools/trace_period.py $OUT/bench_s1_kernel_trace.csv --kernel "conv_split_kernel" --last 3200 > $OUT/bench_s1_trace_period.json
rm -f $OUT/bench_s2_kernel_trace.csv $OUT/bench_s1_kernel_trace.csv      # tens of MB; the derived JSON stays
echo "bench traces done"
python3 bench.py --steps 150 --warmup 15 > $OUT/bench_line_untraced.json 2> $OUT/bench_untraced.err
echo "untraced bench done"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- python3 tools/conv_loop.py 300 1024 > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o secondary -- python3 tools/secondary_bench.py --reps 200 > $OUT/secondary.jsonl 2> $OUT/secondary.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o sec_pmc_$C -- python3 tools/secondary_bench.py --reps 30 > /dev/null 2>&1
done
rm -f $OUT/secondary_kernel_trace.csv
echo "secondary done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fdtd128 -- python3 tools/fdtd_loop.py 128 334 128 > $OUT/fdtd128.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fdtd52 -- python3 tools/fdtd_loop.py 52 512 128 > $OUT/fdtd52.txt 2>&1
rm -f $OUT/fdtd128_kernel_trace.csv $OUT/fdtd52_kernel_trace.csv
# device-clock view of the two range streams (diagnostic build with phase stamps; no tracer attached)
if [ -f gpuaudiobench_amd/libgab_hip_ablate.so ]; then
  GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so python3 tools/stamp_split.py 1024 x ranges > $OUT/stamps_split.txt 2>&1
fi
bash tools/ablate_sweep.sh > $OUT/ablations.txt 2>&1
python3 tools/multiqueue_conv.py --buffers 4000 > $OUT/multiqueue.jsonl 2>/dev/null
tools/ubench/bin/library_baseline > $OUT/library_baseline.json
tools/ubench/bin/mfma_dft > $OUT/mfma_dft.jsonl
ls $OUT
