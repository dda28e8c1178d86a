#!/bin/bash
# conv_split_batch_kernel at 1024 channels, short against long launches: what the L2s exchange with memory per buffer
# (separate --pmc passes, the program directly after --):   bash tools/pmc_batch_long.sh [tag]   -> gpurun_out/<tag>/long.txt
TAG=${1:-pmc_batch_long}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for N in 128 2048; do
  for C in FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum; do
    NBUF=$N rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_${N}_$C -- python3 tools/batch_conv.py 1024 > $OUT/pmc_${N}_$C.txt 2>&1
    echo "pmc $N $C rc=$?"
  done
done
python3 - $OUT <<'PY' | tee $OUT/long.txt
import csv, glob, sys, os, re
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "pmc_*_counter_collection.csv"))):
    n = int(re.search(r"pmc_(\d+)_", os.path.basename(f)).group(1))
    vals = {}
    for r in csv.DictReader(open(f)):
        if "conv_split_batch_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in vals.items():
        full = [x for x in v if x >= 0.5 * max(v)]          # the full-length launches (the parity launch is 9 buffers)
        rows.setdefault(c, {})[n] = sum(full) / len(full) / n
print("%-22s %16s %16s   (per buffer; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950)" % ("counter", "128 per launch", "2048 per launch"))
for c in sorted(rows):
    print("%-22s %16.1f %16.1f" % (c, rows[c].get(128, float("nan")), rows[c].get(2048, float("nan"))))
PY
