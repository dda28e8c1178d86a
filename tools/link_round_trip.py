"""datatransfer with both link directions busy at once (gab_datatransfer_round_trip) against the one-after-the-other
H2D -> kernel -> D2H of the reference, at the five datacopy splits of 10 MiB.  Checks bit-identity against the device
kernel on the same input, prints wall medians and the link rate (in + out bytes / wall).
GAB_LIB_PATH=.../libgab_hip_ablate.so + LINK_WGS="64 128 256 512": also sweeps the workgroup count."""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab

BASE = 2621440
SPLITS = [("datacopy0199", 0.01, 0.99), ("datacopy2080", 0.20, 0.80), ("datacopy5050", 0.50, 0.50),
          ("datacopy8020", 0.80, 0.20), ("datacopy9901", 0.99, 0.01)]
ITERS = int(os.environ.get("ITERS", "300"))
wgs = os.environ.get("LINK_WGS", "").split()
out = {"iters": ITERS, "rows": []}
for name, fi, fo in SPLITS:
    n_in, n_out = int(BASE * np.float32(fi)), int(BASE * np.float32(fo))
    h_in = torch.from_numpy(gab.harness.noise(n_in, seed=3)).abs().pin_memory()
    h_out = torch.zeros(n_out).pin_memory()
    d_in = torch.empty(n_in, device="cuda"); d_out = torch.empty(n_out, device="cuda")
    h_seq = torch.zeros(n_out).pin_memory()

    def seq():
        d_in.copy_(h_in, non_blocking=True)
        gab.check(gab.lib.gab_datatransfer(d_in.data_ptr(), d_out.data_ptr(), n_in, n_out, torch.cuda.current_stream().cuda_stream))
        h_seq.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream().synchronize()

    def timed(fn):
        for _ in range(20): fn()
        t = []
        for _ in range(ITERS):
            t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
        return float(np.median(t)) * 1e6, float(np.percentile(t, 95)) * 1e6

    seq_p50, seq_p95 = timed(seq)
    row = {"name": name, "in_bytes": 4 * n_in, "out_bytes": 4 * n_out, "sequential_us": round(seq_p50, 1),
           "sequential_GBps": round(4 * (n_in + n_out) / seq_p50 / 1e3, 1), "round_trip": []}
    for w in (wgs or [""]):
        if w: os.environ["GAB_LINK_WGS"] = w
        plan = gab.LinkPlan(n_in)
        h_out.zero_()
        plan.round_trip(h_in, h_out)
        same = bool(torch.equal(h_out.view(torch.int32), h_seq.view(torch.int32)))
        p50, p95 = timed(lambda: plan.round_trip(h_in, h_out))
        same = same and bool(torch.equal(h_out.view(torch.int32), h_seq.view(torch.int32)))
        floor = max(4 * n_in, 4 * n_out) / 55e3          # us at 55 GB/s one way
        row["round_trip"].append({"workgroups": w or "default", "p50_us": round(p50, 1), "p95_us": round(p95, 1),
                                  "GBps": round(4 * (n_in + n_out) / p50 / 1e3, 1), "bit_identical": same,
                                  "larger_direction_alone_us_at_55GBps": round(floor, 1)})
        plan.close()
    out["rows"].append(row)
    print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
