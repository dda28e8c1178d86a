"""Device time and algorithmic-byte rate of every registered benchmark through the
harness (GPUABenchmark::runBenchmark): median wall ms, median device ms (event
pair around the kernels), algorithmic GB/s = algorithmicBytes() / device time."""
import json, sys
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab

CASES = [
    ("NoOp", dict(n_tracks=128)), ("gain", dict(n_tracks=128)), ("gain", dict(n_tracks=65536)),
    ("GainStats", dict(n_tracks=128)), ("GainStats", dict(n_tracks=65536)),
    ("datacopy2080", {}), ("datacopy5050", {}), ("datacopy8020", {}),
    ("FFT1D", dict(n_tracks=128)), ("FFT1D", dict(n_tracks=8192)),
    ("IIRFilter", dict(n_tracks=128)), ("IIRFilter", dict(n_tracks=8192)),
    ("Conv1D", dict(n_tracks=256, ir_length=256)), ("Conv1D", dict(n_tracks=128)),
    ("Conv1D_accel", dict(n_tracks=1024, ir_length=4096)),
    ("Conv1D_accel", dict(n_tracks=1024, ir_length=4096, conv_mode=gab.CONV_STATELESS)),
    ("Conv1D_accel", dict(n_tracks=128)),
    ("ModalFilterBank", {}), ("ModalFilterBank", dict(n_tracks=1024, modal_mode=1)), ("DWG1DNaive", dict(n_tracks=128)), ("DWG1DAccel", dict(n_tracks=128)),
    ("DWG1DAccel", dict(n_tracks=1024)),
    ("FDTD3D", dict(n_tracks=128, buffer_size=64)), ("FDTD3D", dict(n_tracks=128, buffer_size=64, fdtd_grid=128)),
    ("FDTD3D", dict(n_tracks=16, buffer_size=16, fdtd_grid=256)),
    ("RndMemRead", dict(n_tracks=128)), ("RndMemRead", dict(n_tracks=8192)), ("RndMemRead", dict(n_tracks=65536)),
]
only = sys.argv[1:]
rows = []
for name, cfg in CASES:
    if only and name not in only:
        continue
    b = gab.Benchmark(name, **cfg)
    b.setup()
    r = b.run(iterations=30, warmup=3)
    v, _ = b.validate()
    alg = b.algorithmic_bytes()
    dev_ms = r.gpu_median_ms
    row = dict(benchmark=name, cfg={k: v for k, v in cfg.items()}, wall_median_ms=round(r.median_ms, 4),
               device_median_ms=round(dev_ms, 5), algorithmic_bytes=alg,
               alg_GBps=round(alg / (dev_ms * 1e-3) / 1e9, 1) if dev_ms > 0 else None,
               frac_of_8TBps=round(alg / (dev_ms * 1e-3) / 8e12, 4) if dev_ms > 0 else None,
               valid=(v.status == 0))
    rows.append(row)
    print(json.dumps(row), flush=True)
    b.close()
