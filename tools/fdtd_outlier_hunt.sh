#!/bin/bash
# Round 3 saw ONE 12.06 ms iteration among 20 of `gpubench --benchmark FDTD3D --fdtdGrid 128 --fdtdSteps 1000 --nTracks 16`
# (median 2.83 ms).  This runs that very command N times (default 30) and prints every run's slowest iteration; the
# next K runs (default 10) go under the kernel tracer and list the resident kernel's durations and the gaps between
# its launches, which tells a slow KERNEL from a slow HOST.   bash tools/fdtd_outlier_hunt.sh [N] [K] -> gpurun_out/fdtd_outlier/
N=${1:-30}; K=${2:-10}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/fdtd_outlier; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="$ROOT/gpuaudiobench_amd/gpubench --benchmark FDTD3D --fdtdGrid 128 --fdtdSteps 1000 --nTracks 16 --nRuns 20 --json --cpu-threads 0"
stats() { python3 -c "
import re,sys
t=open(sys.argv[1]).read()
g=lambda k: re.search(r'\"%s\": ([\\w.]+)'%k,t).group(1)
print('%s: p50 %s ms, slowest %s ms, meets_deadline %s' % (sys.argv[2], g('p50_ms'), g('max_ms'), g('meets_deadline')))" "$1" "$2"; }
for i in $(seq 1 $N); do $CMD > $OUT/run_$i.txt 2>&1; stats $OUT/run_$i.txt "run $i"; done > $OUT/plain_runs.txt
for i in $(seq 1 $K); do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$i -o t -- $CMD > $OUT/traced_$i.txt 2>&1
  stats $OUT/traced_$i.txt "traced run $i"
  python3 - $OUT/trace_$i/t_kernel_trace.csv <<'PY'
import sys, csv
res=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in csv.DictReader(open(sys.argv[1])) if 'fdtd_resident' in r['Kernel_Name'])
d=[(e-s)/1e3 for s,e in res]; g=[(res[k+1][0]-res[k][1])/1e3 for k in range(len(res)-1)]
print("    resident kernel us: min %.0f med %.0f max %.0f over %d launches; gap between launches us: min %.0f med %.0f max %.0f (after launch #%d)" % (
    min(d), sorted(d)[len(d)//2], max(d), len(d), min(g), sorted(g)[len(g)//2], max(g), g.index(max(g))), flush=True)
PY
done > $OUT/traced_runs.txt
cat $OUT/plain_runs.txt $OUT/traced_runs.txt
