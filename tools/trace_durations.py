"""Per-launch durations of one kernel from a rocprofv3 kernel trace: python tools/trace_durations.py <kernel_trace.csv> <name part>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; us:", " ".join("%.0f" % x for x in d))
