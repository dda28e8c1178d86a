#!/bin/bash
# L2-miss (fabric) traffic of conv_split_batch_kernel at a channel count whose state is far beyond the Infinity Cache:
#   bash tools/pmc_batch_channels.sh 16384   -> gpurun_out/pmc_batch_<T>/means.txt
T=${1:-16384}
OUT=$PWD/gpurun_out/pmc_batch_$T; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o pmc_$C -- python3 tools/batch_conv.py $T > $OUT/pmc_$C.txt 2>&1
  echo "pmc $C rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 tools/batch_conv.py $T > $OUT/stats.txt 2>&1
python3 - $OUT $T <<'PY' | tee $OUT/means.txt
import csv, glob, sys, os
out, T = sys.argv[1], int(sys.argv[2])
n = int(os.environ.get("NBUF", 64 if T <= 16384 else 32))      # as tools/batch_conv.py chooses
alg = 4 * T * (2 * 512 + 2 * 4096) * n
vals = {}
for f in sorted(glob.glob(os.path.join(out, "pmc_*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "conv_split_batch_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = None
for r in csv.DictReader(open(os.path.join(out, "stats_kernel_stats.csv"))):
    if "conv_split_batch_kernel" in r["Name"]:
        dur = float(r["AverageNs"])
print("T=%d, %d buffers per launch, algorithmic bytes per launch %d, average launch %.1f us (rocprofv3 --stats)" % (T, n, alg, dur / 1e3))
for c, v in sorted(vals.items()):
    v = v[len(v) // 4:]
    m = sum(v) / len(v)
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B: doubled (MI355X_MICROARCH.md, HBM section)
    b = m * 1024 * (2 if c == "FETCH_SIZE" else 1)
    print("   %-12s %14.1f KiB raw -> %.1f MB per launch = %.3f x algorithmic, %.2f TB/s over the launch" % (c, m, b / 1e6, b / alg, b / dur / 1e3))
PY
