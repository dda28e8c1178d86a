"""Mean per launch of every counter rocprofv3 collected for one kernel, from the
<dir>/pmc_*_counter_collection.csv files of separate --pmc passes (tools/profile_r05.sh).

    python tools/pmc_means.py <dir> <kernel name part> <algorithmic bytes per launch> [skip first N launches] > out.json

FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B,
MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

d, part, alg = sys.argv[1], sys.argv[2], int(sys.argv[3])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
means, counts = {}, {}
for f in sorted(glob.glob(os.path.join(d, "pmc_*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if part in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[skip:]
        means[k] = sum(v) / len(v)
        counts[k] = len(v)
out = {"_comment": "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                   "--no-side-legs: mean per launch of %s, launches %d.., one counter group per run; FETCH_SIZE / WRITE_SIZE "
                   "in KiB, FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request)" % (part, skip),
       "launches_averaged": counts, "counters_mean_per_launch": means}
if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    out["fetch_bytes_corrected"] = means["FETCH_SIZE"] * 1024 * 2
    out["write_bytes"] = means["WRITE_SIZE"] * 1024
    out["hbm_traffic_bytes_per_launch"] = out["fetch_bytes_corrected"] + out["write_bytes"]
    out["algorithmic_bytes_per_launch"] = alg
    out["traffic_over_algorithmic"] = out["hbm_traffic_bytes_per_launch"] / alg
print(json.dumps(out, indent=1))
