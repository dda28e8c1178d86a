#!/bin/bash
# PMC passes (one counter group per run) over ONE case of tools/kernel_table.py, mean per launch per kernel:
#   bash tools/pmc_case.sh <case> [tag]  -> gpurun_out/pmc_<case>_<tag>/means.txt
CASE=$1; TAG=${2:-r04}
OUT=$PWD/gpurun_out/pmc_${CASE}_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- python3 tools/kernel_table.py run $CASE > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N rc=$?"
done
python3 - $OUT <<'PY' | tee $OUT/means.txt
import csv, glob, sys, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "gab::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 4:]
        print("   %-22s %16.1f  (mean of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
