#!/bin/bash
# PMC passes over DWG1DAccel at 8192 lines for one form of the cells kernel (diagnostic build, GAB_DWG_FORM):
#   bash tools/pmc_dwg.sh <form> [tag] -> gpurun_out/pmc_dwg_form<form>_<tag>/means.txt
FORM=$1; TAG=${2:-r06}
OUT=$PWD/gpurun_out/pmc_dwg_form${FORM}_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export GAB_LIB_PATH=$PWD/gpuaudiobench_amd/libgab_hip_ablate.so GAB_DWG_FORM=$FORM
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$N -- python3 tools/kernel_table.py run dwg_accel_8192 > $OUT/pmc_$N.txt 2>&1
  echo "pmc $N rc=$?"
done
python3 - $OUT <<'PY' | tee $OUT/means.txt
import csv, glob, sys, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "dwg" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 4:]
        print("   %-28s %16.1f  (mean of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
