#!/usr/bin/env python3
"""Per-buffer period of the streaming convolution from a rocprofv3 --kernel-trace CSV.

    python tools/trace_period.py <..._kernel_trace.csv> [--kernel conv_split] [--last N] [--per-buffer R]

With one launch per buffer the period is what --stats reports as the average duration plus the gap.
With R channel-range launches per buffer on R streams the kernels overlap, so the per-kernel average
no longer says what a buffer costs; the trace itself does: (last end - first start) of the last N
launches divided by the N / R buffers they carry.  Prints one JSON line.
"""
import argparse, csv, json, statistics

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--kernel", default="conv_split")
ap.add_argument("--last", type=int, default=6400)
ap.add_argument("--per-buffer", type=int, default=0, help="launches per buffer (0: infer from the kernel name)")
ap.add_argument("--alg-bytes", type=float, default=37748736.0)
args = ap.parse_args()
rows = [r for r in csv.DictReader(open(args.trace)) if args.kernel in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-args.last:]
m = __import__("re").search(r"(\w+_kernel)", rows[0]["Kernel_Name"])
name = m.group(1) if m else rows[0]["Kernel_Name"][:60]
R = args.per_buffer or (2 if "range" in name else 1)
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
dur = [e - s for s, e in zip(st, en)]
span = max(en) - min(st)
buffers = len(rows) / R
busy = 0
cur_s, cur_e = None, None
for s, e in sorted(zip(st, en)):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
period_us = span / buffers / 1e3
print(json.dumps({"kernel": name, "launches": len(rows), "launches_per_buffer": R, "buffers": buffers,
                  "avg_kernel_duration_us": statistics.mean(dur) / 1e3, "median_kernel_duration_us": statistics.median(dur) / 1e3,
                  "span_us": span / 1e3, "period_us_per_buffer": period_us,
                  "device_busy_fraction_of_span": busy / span,
                  "avg_kernels_in_flight": sum(dur) / span,
                  "alg_GBps": args.alg_bytes / period_us / 1e3, "frac_of_8TBps": args.alg_bytes / period_us / 1e3 / 8000.0}))
