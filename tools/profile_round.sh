#!/bin/bash
# Round-end profile set (run through gpurun): kernel-trace stats of bench.py and of the FDTD3D /
# modal loops, then PMC passes (one counter group per run, kernel-trace only) on the conv loop.
# Usage: bash tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --steps 3000 --warmup 300 --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fdtd128 -- python3 tools/fdtd_loop.py 128 334 128 > $OUT/fdtd128.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o modal -- python3 tools/modal_loop.py > $OUT/modal.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o batch -- python3 tools/batch_conv.py 1024 > $OUT/batch.txt 2>&1
echo "traces done"
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- python3 tools/conv_loop.py 200 1024 > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N done"
done
ls $OUT
