#!/bin/bash
# Round-6 profile set (run through gpurun): bash tools/profile_r06.sh <tag> -> gpurun_out/prof_<tag>/
#  1. the driver's command, untraced and under the kernel tracer (kernel stats + the bench line it printed)
#  2. PMC passes (one counter group per run, kernel-trace only) over the same launches
#  3. the batch launch: twelve waves against eight on this box (diagnostic build), barrier timeline of the twelve-wave launch
#  4. the round trip: what the check launch costs under each rule, device timeline
#  5. the engine: pipelined rate, ONE buffer in flight; gab_conv_process_batch across channel counts; five bench runs
# (DWG: tools/dwg_ab.sh, tools/pmc_dwg.sh; the per-size kernel table: tools/kernel_table.sh r06)
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DRIVER="python3 bench.py --steps 20 --warmup 5"                                  # the driver's command, as it is
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs"  # the same launches without the CPU legs (PMC passes)
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20_warmup5.json 2> $OUT/bench.err; echo "untraced bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- $DRIVER > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- $BENCH > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N rc=$?"
done
python3 tools/pmc_means.py $OUT conv_split_batch12_kernel 4831838208 5 > $OUT/conv_batch_pmc_means.json; echo "pmc means rc=$?"
WAVES="8 12" bash tools/batch_waves_ab.sh $TAG 1024 > /dev/null 2>&1; grep -v amdgpu gpurun_out/batch_waves_ab_$TAG.txt > $OUT/batch_waves_ab.txt; echo "waves ab rc=$?"
GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so python3 tools/stamp_batch12.py 2>&1 | grep -v amdgpu > $OUT/batch12_stamps.txt; echo "stamps rc=$?"
python3 tools/roundtrip_check_cost.py 2>&1 | grep -v amdgpu > $OUT/roundtrip_check_cost.txt; echo "check cost rc=$?"
GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so python3 tools/roundtrip_timeline.py 2>&1 | grep -v amdgpu > $OUT/roundtrip_timeline.txt; echo "timeline rc=$?"
python3 tools/roundtrip_conv.py 1024 520 2>&1 | grep -v amdgpu > $OUT/roundtrip_conv.txt; echo "roundtrip rc=$?"
python3 tools/engine_latency.py 1024 2000 2>&1 | grep -v amdgpu > $OUT/engine_latency.txt; echo "engine latency rc=$?"
ENGINE_AHEADS="16 16 8 48" python3 tools/engine_conv.py 1024 63 2>&1 | grep -v amdgpu > $OUT/engine_conv.txt; echo "engine rc=$?"
python3 tools/batch_conv.py 2>&1 | grep -v amdgpu > $OUT/batch_channels.txt; echo "batch rc=$?"
for i in 1 2 3 4 5; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f buffers/s  frac %.4f' % (d['value'], d['roofline']['frac']))"; done > $OUT/bench_five_runs.txt 2>&1
ls $OUT | head -60
