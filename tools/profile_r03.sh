#!/bin/bash
# Round-3 profile set (run through gpurun): bash tools/profile_r03.sh <tag> -> gpurun_out/prof_<tag>/
#  1. the driver's command under the kernel tracer (kernel stats + the bench line it printed)
#  2. PMC passes (one counter group per run, kernel-trace only) over the same launches
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DRIVER="python3 bench.py --steps 20 --warmup 5"                                  # the driver's command, as it is
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs"  # the same launches without the CPU legs (PMC passes)
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20_warmup5.json 2> $OUT/bench.err
echo "untraced bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- $DRIVER > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_rocprof.err
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- $BENCH > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N done"
done
ls $OUT
