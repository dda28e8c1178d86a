#!/bin/bash
# One rocprofv3 --kernel-trace --stats pass per (benchmark, size) case -> gpurun_out/ktab_<tag>/ ; then the table.
# Usage (through gpurun): bash tools/kernel_table.sh <tag> [case ...]
TAG=${1:-r03}; shift
OUT=$PWD/gpurun_out/ktab_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CASES=${@:-$(python3 tools/kernel_table.py list)}
for c in $CASES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $c -- python3 tools/kernel_table.py run $c > $OUT/$c.json 2> $OUT/$c.err
  rm -f $OUT/${c}_kernel_trace.csv $OUT/${c}_agent_info.csv $OUT/${c}_domain_stats.csv
  echo "$c done: $(grep -c . $OUT/$c.json) line(s)"
done
python3 tools/kernel_table.py collect $OUT $OUT/kernels_by_size.csv $OUT/kernels_by_size.md
