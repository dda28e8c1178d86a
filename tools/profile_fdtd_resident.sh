#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over the LDS-resident FDTD kernel at 128^3:
#   bash tools/profile_fdtd_resident.sh <tag>  ->  gpurun_out/prof_fdtd_<tag>/
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_fdtd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 tools/fdtd_loop.py 128 334 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $CMD > $OUT/trace.txt 2>&1
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- $CMD > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N done"
done
