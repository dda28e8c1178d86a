#!/usr/bin/env python3
"""Per-kernel, per-SIZE device times of every benchmark on the path (one rocprofv3 pass per case).

    python tools/kernel_table.py list                      -> case names
    python tools/kernel_table.py run <case> [--iters N]    -> runs the case through the harness, one JSON line
    python tools/kernel_table.py collect <dir> <out.csv> <out.md>
        <dir> holds, per case, <case>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and <case>.json
        (the line `run` printed); writes one row per (case, kernel) with the kernel's average duration,
        the case's algorithmic bytes and the fraction of 8 TB/s they make.

tools/kernel_table.sh drives it: rocprofv3 ... -- python3 tools/kernel_table.py run <case>, case by case, so
that no two sizes of one kernel share a stats row."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_GBS = 8000.0
PCIE_GEN5_X16_GBS = 63.0        # spec, one direction

# case -> (registry name, harness config, iterations)
CASES = {
    "gain_128": ("gain", dict(n_tracks=128), 300), "gain_8192": ("gain", dict(n_tracks=8192), 300),
    "gain_65536": ("gain", dict(n_tracks=65536), 100),
    "gainstats_128": ("GainStats", dict(n_tracks=128), 300), "gainstats_8192": ("GainStats", dict(n_tracks=8192), 300),
    "gainstats_65536": ("GainStats", dict(n_tracks=65536), 100),
    "iir_128": ("IIRFilter", dict(n_tracks=128), 300), "iir_8192": ("IIRFilter", dict(n_tracks=8192), 300),
    "iir_65536": ("IIRFilter", dict(n_tracks=65536), 100),
    "fft_128": ("FFT1D", dict(n_tracks=128), 300), "fft_8192": ("FFT1D", dict(n_tracks=8192), 200),
    "fft_65536": ("FFT1D", dict(n_tracks=65536), 20),
    "rndmem_128": ("RndMemRead", dict(n_tracks=128), 300), "rndmem_8192": ("RndMemRead", dict(n_tracks=8192), 300),
    "rndmem_65536": ("RndMemRead", dict(n_tracks=65536), 100),
    "conv1d_c2_256x256": ("Conv1D", dict(n_tracks=256, ir_length=256), 300),
    "conv1d_128x1024": ("Conv1D", dict(n_tracks=128), 300),
    "conv_accel_c3_1024x4096": ("Conv1D_accel", dict(n_tracks=1024, ir_length=4096), 500),
    "conv_accel_128x512": ("Conv1D_accel", dict(n_tracks=128), 300),
    "dwg_accel_128": ("DWG1DAccel", dict(n_tracks=128), 300), "dwg_accel_8192": ("DWG1DAccel", dict(n_tracks=8192), 100),
    "dwg_naive_128": ("DWG1DNaive", dict(n_tracks=128), 200), "dwg_naive_8192": ("DWG1DNaive", dict(n_tracks=8192), 50),
    "modal_placeholder": ("ModalFilterBank", dict(), 30), "modal_bank_1024": ("ModalFilterBank", dict(n_tracks=1024, modal_mode=1), 100),
    "datacopy0199": ("datacopy0199", dict(), 200), "datacopy2080": ("datacopy2080", dict(), 200),
    "datacopy5050": ("datacopy5050", dict(), 200), "datacopy8020": ("datacopy8020", dict(), 200),
    "datacopy9901": ("datacopy9901", dict(), 200),
    "datacopy0199_seq": ("datacopy0199", dict(datacopy_mode=1), 200), "datacopy2080_seq": ("datacopy2080", dict(datacopy_mode=1), 200),
    "datacopy5050_seq": ("datacopy5050", dict(datacopy_mode=1), 200), "datacopy8020_seq": ("datacopy8020", dict(datacopy_mode=1), 200),
    "datacopy9901_seq": ("datacopy9901", dict(datacopy_mode=1), 200),
    "fdtd_52": ("FDTD3D", dict(n_tracks=128, buffer_size=128), 20),
    "fdtd_128": ("FDTD3D", dict(n_tracks=128, buffer_size=128, fdtd_grid=128), 10),
    "noop_128": ("NoOp", dict(n_tracks=128), 300),
}


# Cases whose algorithmic bytes are dominated by what crosses the LINK in an iteration (the modal placeholder
# uploads its 32 MiB parameter table every iteration and its kernel reads 32 records of it): the KERNEL is priced
# against what the kernel itself reads and writes, the iteration against the link.
KERNEL_BYTES = {"modal_placeholder": 32 * 8 * 4 + 32 * 512 * 4}
# The LDS-resident FDTD kernel moves no field through HBM: its per-step floor is the longer of (a) VALU issue — the
# counted instructions per SIMD and step x 2.63 cycles per wave64 instruction (measured) at the part's clock — and (b) the neighbour
# hand-off's request-to-data round trip with every workgroup asking at once; profiles/r04_fdtd_bound.md holds both.
# The modal bank is VALU-bound: the golden's own phasor recurrence (unfused, so that every mode's sequence is the
# oracle's bit for bit) is 3 packed + 2 plain fp32 instructions per mode and sample.  A SIMD that holds four or more
# waves issues a plain fp32 wave64 instruction every 2.63 clocks and a packed one every 5.2 (tools/ubench/valu_rate,
# profiles/r04_valu_rate.txt).  Floor = modes x samples / 64 lanes / 1024 SIMDs x (3 x 5.2 + 2 x 2.63) clk / 2.4 GHz.
MODAL_CLK_PER_WAVE_MODE_SAMPLE = 3 * 5.2 + 2 * 2.63


def modal_bank_floor_us(n_modes, bufsize):
    return n_modes * bufsize / 64.0 / 1024 * MODAL_CLK_PER_WAVE_MODE_SAMPLE / 2.4e3


FDTD_RESIDENT_FLOOR_US_PER_STEP = {"fdtd_128": None, "fdtd_52": None}     # filled from profiles/r04_fdtd_bound.json when present


def run(case, iters=None):
    import gpuaudiobench_amd as gab
    name, cfg, n = CASES[case]
    n = iters or n
    b = gab.Benchmark(name, **cfg)
    b.setup()
    r = b.run(iterations=n, warmup=5)
    v, _ = b.validate()
    alg = b.algorithmic_bytes()
    row = dict(case=case, benchmark=name, config=cfg, iterations=n, wall_median_ms=r.median_ms, wall_p95_ms=r.p95_ms,
               device_median_ms=r.gpu_median_ms, algorithmic_bytes=alg, valid=(v.status == 0), max_error=v.max_error)
    if name.startswith("datacopy"):
        # in + out bytes over the link.  Sequential (datacopy_mode 1): the directions take turns, the bound is the sum
        # of both at the link's one-way rate.  Overlap (default): both at once, the bound is the LARGER direction alone.
        import numpy as np
        base = 10 * 1024 * 1024 // 4
        n_in, n_out = int(base * np.float32(int(name[8:10]) / 100.0)), int(base * np.float32(int(name[10:12]) / 100.0))
        assert 4 * (n_in + n_out) == alg, (n_in, n_out, alg)
        row["link_GBps_device"] = alg / (r.gpu_median_ms * 1e-3) / 1e9 if r.gpu_median_ms > 0 else None
        row["link_GBps_wall"] = alg / (r.median_ms * 1e-3) / 1e9
        bound_bytes = alg if cfg.get("datacopy_mode") == 1 else 4 * max(n_in, n_out)
        row["link_bound"] = "both directions in turn" if cfg.get("datacopy_mode") == 1 else "the larger direction"
        row["frac_of_pcie_gen5_x16"] = bound_bytes / (r.median_ms * 1e-3) / 1e9 / PCIE_GEN5_X16_GBS
    print(json.dumps(row), flush=True)
    b.close()


def short(kernel):
    k = kernel.replace("(anonymous namespace)::", "").replace("void ", "").replace("gab::", "")
    return re.sub(r"\(.*", "", k).strip()


def collect(d, out_csv, out_md):
    rows = []
    for case in CASES:
        jf, sf = os.path.join(d, case + ".json"), os.path.join(d, case + "_kernel_stats.csv")
        if not (os.path.exists(jf) and os.path.exists(sf)):
            continue
        lines = [ln for ln in open(jf).read().splitlines() if ln.startswith("{")]
        if not lines:
            continue
        meta = json.loads(lines[-1])
        ks = [r for r in csv.DictReader(open(sf)) if "gab::" in r["Name"]]
        total_avg = sum(float(r["AverageNs"]) * int(r["Calls"]) for r in ks)
        per_iter_ns = total_avg / max(1, meta["iterations"] + 5 + 1)      # timed + warm-up + the validation iteration
        kbytes = KERNEL_BYTES.get(case, meta["algorithmic_bytes"])
        floor = None
        bf = os.path.join(ROOT, "profiles", "r04_fdtd_bound.json")
        if case in FDTD_RESIDENT_FLOOR_US_PER_STEP and os.path.exists(bf):
            floor = json.load(open(bf)).get(case, {}).get("floor_us_per_step")
        link = case.startswith("datacopy")        # the iteration is bound by the link, whatever kernel runs beside the copies
        valu_floor = None
        if case.startswith("modal_bank"):
            nt = meta["config"].get("n_tracks", 128)
            valu_floor = modal_bank_floor_us(min(1024 * nt, 1024 * 1024), meta["config"].get("buffer_size", 512))
        for r in ks:
            calls = int(r["Calls"])
            rows.append(dict(case=case, benchmark=meta["benchmark"], kernel=short(r["Name"]), calls=calls,
                             avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3,
                             stddev_us=float(r["StdDev"]) / 1e3, kernels_us_per_iteration=per_iter_ns / 1e3,
                             algorithmic_bytes=kbytes,
                             alg_GBps=kbytes / per_iter_ns if per_iter_ns > 0 else 0.0,
                             # rooms resident in LDS: not an HBM figure at all — the fraction is of the kernel's own floor
                             frac_of_8TBps=(kbytes / per_iter_ns / PEAK_GBS if per_iter_ns > 0 else 0.0) if floor is None and not link and valu_floor is None else None,
                             bound=("PCIe Gen5 x16, 63 GB/s one way, %s" % meta.get("link_bound", "") if link else
                                    "VALU issue, 3 packed + 2 plain fp32 per mode-sample: %.1f us" % valu_floor if valu_floor is not None else
                                    "hbm" if floor is None else "issue + neighbour hand-off, %.2f us per step" % floor),
                             frac_of_bound=(meta.get("frac_of_pcie_gen5_x16") if link else
                                            valu_floor / (per_iter_ns / 1e3) if valu_floor is not None else None if floor is None else
                                            floor * meta["config"].get("buffer_size", 512) * 3 / (per_iter_ns / 1e3)),
                             harness_device_median_us=meta["device_median_ms"] * 1e3, wall_median_us=meta["wall_median_ms"] * 1e3,
                             valid=meta["valid"], link_GBps_wall=meta.get("link_GBps_wall"),
                             link_GBps_device=meta.get("link_GBps_device")))
    keys = list(rows[0].keys()) if rows else []
    with open(out_csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys)
        w.writeheader()
        w.writerows(rows)
    with open(out_md, "w") as f:
        f.write("Per-kernel, per-size device times (rocprofv3 --kernel-trace --stats, ONE pass per case; `kernels us/iter` = all\n"
                "gab kernels of one harness iteration; frac = algorithmic bytes / that time / 8 TB/s.  datacopy rows: link GB/s\n"
                "= (in + out bytes) / wall median of the iteration; their fraction is of PCIe Gen5 x16's 63 GB/s one way — of the\n"
                "LARGER direction's bytes / wall for the default overlapped schedule (both directions at once: the kernel in the row\n"
                "runs for the whole transfer), of in + out bytes / wall for `_seq` (H2D, kernel, D2H in turn, the reference's).\n\n")
        f.write("Rooms the FDTD3D kernel keeps resident in LDS cross no HBM roofline: their fraction is of the kernel's own per-step floor\n"
                "(profiles/r04_fdtd_bound.md).  The modal placeholder's kernel is priced against the 66 560 B it reads and writes (its\n"
                "iteration is the 32 MiB upload).\n\n")
        f.write("| case | kernel | calls | avg us | min us | kernels us/iter | bytes priced | GB/s | bound | frac of bound | link GB/s | valid |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            frac = r["frac_of_8TBps"] if r["frac_of_bound"] is None else r["frac_of_bound"]
            f.write("| %s | %s | %d | %.2f | %.2f | %.2f | %d | %.0f | %s | %.3f | %s | %s |\n" % (
                r["case"], r["kernel"], r["calls"], r["avg_us"], r["min_us"], r["kernels_us_per_iteration"], r["algorithmic_bytes"],
                r["alg_GBps"], "8 TB/s HBM" if r["bound"] == "hbm" else r["bound"], frac,
                "%.1f" % r["link_GBps_wall"] if r["link_GBps_wall"] else "", r["valid"]))
    print("wrote", out_csv, out_md, len(rows), "rows")


if __name__ == "__main__":
    if sys.argv[1] == "list":
        print("\n".join(CASES))
    elif sys.argv[1] == "run":
        it = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else None
        run(sys.argv[2], it)
    elif sys.argv[1] == "collect":
        collect(sys.argv[2], sys.argv[3], sys.argv[4])
