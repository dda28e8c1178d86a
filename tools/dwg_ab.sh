#!/bin/bash
# DWG1DAccel at 8192 lines: the cells kernel's forms side by side on ONE box (diagnostic build, GAB_DWG_FORM):
#   0 round 5's (8 lines per workgroup, input staged in LDS)   1, 2 no staging, 8 / 16 lines   3, 4 the lean kernel, 8 / 16 lines (FORMS="0 3 4")
# rocprofv3 --kernel-trace --stats per form, twice, alternating -> gpurun_out/dwg_ab_<tag>/summary.txt
TAG=${1:-r06}
OUT=$PWD/gpurun_out/dwg_ab_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so
for rep in 1 2; do
  for FORM in ${FORMS:-0 3 4}; do
    export GAB_DWG_FORM=$FORM
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o form${FORM}_rep$rep -- python3 tools/kernel_table.py run dwg_accel_8192 > $OUT/form${FORM}_rep$rep.json 2> $OUT/form${FORM}_rep$rep.err
    rm -f $OUT/form${FORM}_rep${rep}_kernel_trace.csv $OUT/form${FORM}_rep${rep}_agent_info.csv $OUT/form${FORM}_rep${rep}_domain_stats.csv
  done
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys, json
d = sys.argv[1]
print("DWG1DAccel, 8192 lines x 512 samples (67.2 MB algorithmic): average kernel durations, us (rocprofv3 --kernel-trace --stats)")
for f in sorted(glob.glob(os.path.join(d, "form*_kernel_stats.csv"))):
    name = os.path.basename(f).replace("_kernel_stats.csv", "")
    rows = {r["Name"].split("(")[0].replace("void gab::(anonymous namespace)::", ""): r for r in csv.DictReader(open(f)) if "dwg" in r["Name"]}
    line = json.loads(open(os.path.join(d, name + ".json")).read().strip().splitlines()[-1])
    tot = sum(float(r["AverageNs"]) for r in rows.values()) / 1e3
    print("%-12s %s | sum %.2f us = %.3f of 8 TB/s | valid %s" % (name, "  ".join("%s %.2f" % (k[:28], float(r["AverageNs"]) / 1e3) for k, r in sorted(rows.items())),
          tot, line["algorithmic_bytes"] / (tot * 1e-6) / 8e12, line["valid"]))
PY
