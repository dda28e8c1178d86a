"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so): the barrier timeline of tools/stamp_batch.py for the
eight-wave doorbell-fed ENGINE launch (period 4000 of a run whose buffers were all published before the launch looked: ~25 ms in, clocks settled)."""
import ctypes, os, sys
os.environ.setdefault("GAB_CONV_SPLIT_DEBUG", "64")
os.environ.setdefault("GAB_ENGINE_WAVES", "8")            # (the eight-wave engine: the product launches the twelve-wave one, tools/stamp_engine12.py)
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T, B, L, NB = 1024, 512, 4096, 64
plan = gab.ConvPlan(T, B, L, scheme="split"); plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
x = torch.from_numpy(np.concatenate([gab.harness.noise(T * B, seed=s) for s in range(NB)])).cuda()
in_ring, out_ring = plan.engine_rings(NB)
in_ring.copy_(x.view(NB, T * B)); torch.cuda.synchronize()
side = torch.cuda.Stream()
plan.engine_start(NB, stream=side)
plan.engine_publish(70 * NB)
plan.engine_stop()
n = 256 * 8 * 8
buf = (ctypes.c_ulonglong * n)()
fn = gab.lib.gab_debug_split_stamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 8, 8) * 0.01      # us
far = st[:, 4, :]
start = far[:, 6]
rel = far[:, :6] - start[:, None]
print("ENGINE, far view: barrier release times after the period's start (us), median over workgroups:")
print("   " + "  ".join("b%d %.2f" % (i + 1, np.median(rel[:, i])) for i in range(6)))
iv = np.diff(np.concatenate([np.zeros((256, 1)), rel], axis=1), axis=1)
print("   far wave, slot 7 (%s) after the period's start: %.2f us" % ("request burst issued" if int(os.environ.get("GAB_CONV_SPLIT_DEBUG", "64")) & 512 else "spectral product done", np.median(far[:, 7] - start)))
print("   interval lengths: " + "  ".join("%.2f" % np.median(iv[:, i]) for i in range(6)) + "   period %.2f" % np.median(rel[:, 5]))
for wv, name in ((0, "forward (pair 0)"), (1, "forward (pair 1)"), (2, "inverse (pair 0)"), (3, "inverse (pair 1)")):
    arr = st[:, wv, :6] - start[:, None]
    wait = rel - arr
    print("%-17s arrival: " % name + "  ".join("%.2f" % np.median(arr[:, i]) for i in range(6)) +
          "   waits: " + "  ".join("%.2f" % np.median(wait[:, i]) for i in range(6)))
