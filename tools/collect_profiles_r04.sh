#!/bin/bash
# Copies what tools/profile_r04.sh <tag> + tools/kernel_table.sh r04 left under gpurun_out/ into profiles/ (tracked).
TAG=${1:-r04b}
S=gpurun_out/prof_$TAG; D=profiles
cp $S/bench_line_steps20_warmup5.json $D/r04_bench_line_steps20_warmup5.json
cp $S/bench_line_under_rocprof.json $D/r04_bench_line_under_rocprof.json
cp $S/bench_kernel_stats.csv $D/r04_bench_kernel_stats.csv
cp $S/conv_batch_pmc_means.json $D/r04_conv_batch_pmc_means.json
cp $S/roundtrip_timeline.txt $D/r04_roundtrip_timeline.txt
cp $S/roundtrip_conv.txt $D/r04_roundtrip_conv.txt
cp $S/roundtrip_memory_copy_trace.csv $D/r04_roundtrip_memory_copy_trace.csv
python3 - $S/roundtrip_kernel_trace.csv $D/r04_roundtrip_kernel_trace_round_trip_kernel.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "round_trip" in r["Kernel_Name"]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Stream_Id", "Start_Timestamp", "End_Timestamp"])
    w.writeheader()
    for r in keep:
        w.writerow({"Kernel_Name": "conv_round_trip_kernel", "Stream_Id": r["Stream_Id"], "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"]})
PY
cp $S/fdtd_resident_pmc_raw.json $D/r04_fdtd_resident_pmc_means.json
cp $S/fdtd_bound.json $D/r04_fdtd_bound.json
cp $S/fdtd_bound.md $D/r04_fdtd_bound.md
cp gpurun_out/prof_fdtd_$TAG/trace_kernel_stats.csv $D/r04_fdtd_resident_kernel_stats.csv 2>/dev/null
cp $S/batch_channels.txt $D/r04_batch_channels.txt
cp gpurun_out/ktab_r04/kernels_by_size.md $D/r04_kernels_by_size.md
cp gpurun_out/ktab_r04/kernels_by_size.csv $D/r04_kernels_by_size.csv
ls -la $D | grep r04_
