#!/bin/bash
# counters of ONE batch launch and ONE engine launch over the same 4032 buffers, one --pmc pass each
# (the program directly after --), per-buffer means -> gpurun_out/pmc_evb_<tag>/engine_vs_batch_pmc.json
TAG=${1:-r05}
OUT=$PWD/gpurun_out/pmc_evb_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for M in batch engine; do
  python3 tools/engine_vs_batch.py $M > $OUT/untraced_$M.txt 2>&1; grep buffers $OUT/untraced_$M.txt
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o ${M}_$N -- python3 tools/engine_vs_batch.py $M > $OUT/${M}_$N.txt 2>&1
    echo "$M pmc $N rc=$? $(grep buffers $OUT/${M}_$N.txt | cut -c1-80)"
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
d = sys.argv[1]
N = 4032
res = {}
for mode, kern in (("batch", "conv_split_batch_kernel"), ("engine", "conv_split_engine_kernel")):
    cnt = {}
    for f in sorted(glob.glob(os.path.join(d, mode + "_*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]:
                continue
            grid = int(r.get("Grid_Size", 0) or 0)
            v = float(r["Counter_Value"])
            # the measured launch is the LAST one of that kernel in the file (batch: the warm-up launches carry the same name)
            cnt[r["Counter_Name"]] = v
    res[mode] = cnt
out = {"_comment": "rocprofv3 --kernel-trace --pmc <group> -- python3 tools/engine_vs_batch.py batch|engine: counters of ONE launch over the same "
                   "4032 buffers (1024 channels), one counter group per pass; per-buffer = / 4032; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950",
       "per_launch": res, "per_buffer": {}, "engine_minus_batch_per_buffer": {}}
for k in sorted(set(res["batch"]) | set(res["engine"])):
    b, e = res["batch"].get(k), res["engine"].get(k)
    out["per_buffer"][k] = {"batch": None if b is None else b / N, "engine": None if e is None else e / N}
    if b is not None and e is not None:
        out["engine_minus_batch_per_buffer"][k] = (e - b) / N
json.dump(out, open(os.path.join(d, "engine_vs_batch_pmc.json"), "w"), indent=1)
for k, v in out["per_buffer"].items():
    print("%-24s batch %14.1f  engine %14.1f  per buffer" % (k, v["batch"] or float("nan"), v["engine"] or float("nan")))
PY
