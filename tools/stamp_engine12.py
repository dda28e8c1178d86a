"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so, GAB_ENGINE_WAVES=12): tools/stamp_batch12.py's barrier timeline for the
twelve-wave ENGINE launch (periods 4000 and 4001 of a run whose buffers were all published before the launch looked)."""
import ctypes, os, sys
os.environ.setdefault("GAB_CONV_SPLIT_DEBUG", "64")
os.environ.setdefault("GAB_ENGINE_WAVES", "12")
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
n = 256 * 12 * 2 * 6
buf = (ctypes.c_ulonglong * n)()
fn = gab.lib.gab_debug_split_stamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 12, 12) * 0.01
rel = st.max(axis=1)
names = ["fwd0", "fwd1", "inv0", "inv1"] + ["farA%d" % i for i in range(4)] + ["farB%d" % i for i in range(4)]
iv = np.diff(rel, axis=1)
lab = ["4000.%d" % (i + 1) for i in range(1, 6)] + ["4001.%d" % (i + 1) for i in range(6)]
print("ENGINE, twelve waves.  interval (ends at barrier):   " + "  ".join("%6s" % l for l in lab))
print("median length, us:                                 " + "  ".join("%6.2f" % np.median(iv[:, i]) for i in range(11)))
print("period 4001 = %.2f us" % np.median(rel[:, 11] - rel[:, 5]))
work = st[:, :, 1:] - rel[:, None, :-1]
last = (st[:, :, 1:] == rel[:, None, 1:])
for wv in range(12):
    print("%-6s work: " % names[wv] + "  ".join("%6.2f" % np.median(work[:, wv, i]) for i in range(11)) +
          "   last: " + "  ".join("%3.0f%%" % (100 * last[:, wv, i].mean()) for i in range(11)))
