#!/bin/bash
# Address-translation counters of conv_split_batch_kernel at 1024 channels for two launch lengths (separate --pmc passes,
# the program directly after --):   bash tools/pmc_batch_tlb.sh [tag]   -> gpurun_out/<tag>/tlb.txt
TAG=${1:-pmc_batch_tlb}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -o "UTCL[0-9A-Z_a-z]*\|TCP_[A-Z0-9_]*TRANSLATION[A-Za-z0-9_]*" $OUT/counters.txt | sort -u > $OUT/utcl_names.txt
for N in 128 2048; do
  for C in TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum; do
    NBUF=$N rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_${N}_$C -- python3 tools/batch_conv.py 1024 > $OUT/pmc_${N}_$C.txt 2>&1
    echo "pmc $N $C rc=$?"
  done
done
python3 - $OUT <<'PY' | tee $OUT/tlb.txt
import csv, glob, sys, os, re
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "pmc_*_counter_collection.csv"))):
    n = int(re.search(r"pmc_(\d+)_", os.path.basename(f)).group(1))
    vals = {}
    for r in csv.DictReader(open(f)):
        if "conv_split_batch_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in vals.items():
        v = [x for x in v if x > 0] or v
        full = [x for x in v if x >= 0.5 * max(v)]          # the full-length launches (the parity launch is 9 buffers)
        m = sum(full) / len(full)
        print("%5d buffers per launch  %-36s %14.0f per launch  %10.1f per buffer" % (n, c, m, m / n))
PY
