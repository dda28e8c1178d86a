#!/bin/bash
# Round-4 profile set (run through gpurun): bash tools/profile_r04.sh <tag> -> gpurun_out/prof_<tag>/
#  1. the driver's command, untraced and under the kernel tracer (kernel stats + the bench line it printed)
#  2. PMC passes (one counter group per run, kernel-trace only) over the same launches, LDS counters included
#  3. the round trip: device-stamp timeline of gab_conv_round_trip (diagnostic library) and a kernel + memory-copy trace
#  4. the LDS-resident FDTD kernel: PMC passes and its own bound (tools/fdtd_bound.py)
#  5. gab_conv_process_batch across channel counts
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DRIVER="python3 bench.py --steps 20 --warmup 5"                                  # the driver's command, as it is
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs"  # the same launches without the CPU legs (PMC passes)
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20_warmup5.json 2> $OUT/bench.err; echo "untraced bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- $DRIVER > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err; echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- $BENCH > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N rc=$?"
done
python3 tools/pmc_means.py $OUT conv_split_batch_kernel 4831838208 5 > $OUT/conv_batch_pmc_means.json; echo "pmc means rc=$?"
GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so python3 tools/roundtrip_timeline.py > $OUT/roundtrip_timeline.txt 2>&1; echo "timeline rc=$?"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o roundtrip -- python3 tools/roundtrip_conv.py 1024 120 > $OUT/roundtrip_conv_traced.txt 2>&1; echo "roundtrip trace rc=$?"
python3 tools/roundtrip_conv.py 1024 520 > $OUT/roundtrip_conv.txt 2>&1; echo "roundtrip rc=$?"
bash tools/profile_fdtd_resident.sh $TAG > $OUT/fdtd_prof.log 2>&1; echo "fdtd pmc rc=$?"
python3 tools/pmc_means.py $PWD/gpurun_out/prof_fdtd_$TAG fdtd_resident_kernel 67637084160 1 > $OUT/fdtd_resident_pmc_raw.json; echo "fdtd means rc=$?"
python3 tools/fdtd_bound.py $OUT/fdtd_resident_pmc_raw.json $OUT/fdtd_bound.json $OUT/fdtd_bound.md > $OUT/fdtd_bound.log 2>&1; echo "fdtd bound rc=$?"
python3 tools/batch_conv.py > $OUT/batch_channels.txt 2>&1; echo "batch rc=$?"
ls $OUT | head -50
