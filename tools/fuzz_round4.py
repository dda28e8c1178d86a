"""Randomised shapes through the entries added or rebuilt in round 4, each against the oracle or the device-buffer
kernel, bit for bit: gab_datatransfer_round_trip (one plan, many shapes), gab_conv1d (tap counts, tracks, buffer sizes).
    python tools/fuzz_round4.py [seed] [cases]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
import oracle as orc
orc.build()
seed, cases = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
rng = np.random.default_rng(seed)
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
plan = gab.LinkPlan(400000)
bad = 0
for c in range(cases):
    n_in, n_out = int(rng.integers(0, 400001)), int(rng.integers(0, 400001))
    if rng.random() < 0.3: n_out = n_in + int(rng.integers(-3, 4))
    n_out = max(0, n_out)
    x = rng.random(n_in, dtype=np.float32)
    if n_in and rng.random() < 0.2: x.view(np.uint32)[rng.integers(0, n_in, size=min(n_in, 50))] = 0xffa5c3e1      # the sentinel itself
    h_in, h_out = torch.from_numpy(x).pin_memory(), torch.full((n_out,), -3.0).pin_memory()
    plan.round_trip(h_in, h_out)
    want = gab.datatransfer(torch.from_numpy(x).cuda(), n_out).cpu().numpy() if n_out else np.zeros(0, np.float32)
    if not np.array_equal(bits(h_out.numpy()), bits(want)):
        bad += 1; print("datatransfer MISMATCH", n_in, n_out, flush=True)
print("datatransfer round trip: %d shapes, %d mismatches" % (cases, bad), flush=True)
plan.close()
bad1 = 0
for c in range(cases // 3):
    L = int(rng.choice([1, 2, 17, 31, 32, 33, 64, 100, 255, 256, 257, 500, 1000, 1024, 1025, 1500, 2049, 3000]))
    T, B = int(rng.integers(1, 12)), int(rng.choice([64, 100, 256, 300, 512, 513, 1000]))
    ir = orc.conv1d_ir(L, T)
    if rng.random() < 0.3: ir[rng.integers(0, L * T, size=3)] = 0.0
    x = orc.noise(T * B, seed=int(rng.integers(1, 10000)))
    y = gab.conv1d(torch.from_numpy(x).cuda(), torch.from_numpy(ir).cuda(), L, T, B).cpu().numpy()
    if not np.array_equal(bits(y), bits(orc.conv1d(x, ir, L, B, T))):
        bad1 += 1; print("conv1d MISMATCH L=%d T=%d B=%d" % (L, T, B), flush=True)
print("conv1d: %d shapes, %d mismatches" % (cases // 3, bad1), flush=True)
sys.exit(1 if bad or bad1 else 0)
