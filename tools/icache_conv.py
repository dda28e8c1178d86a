"""Diagnostic (GAB_ABLATE build): runs partition A three times and B twice inside one launch
(same code addresses) to see how much of a stage's time is first-touch instruction fetch."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["GAB_CONV_ABLATE"] = "7"
os.environ["GAB_CONV_SCHEME"] = "classic"      # the stamps live in the classic kernel
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
plan = gab.ConvPlan(T, B, L)
plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
xs = [torch.from_numpy(gab.harness.noise(T * B, seed=s)).cuda() for s in range(4)]
out = torch.empty(T * B, device="cuda")
for i in range(20):
    plan.process(xs[i % 4], out=out)
torch.cuda.synchronize()
NB = (T + 1) // 2
buf = (ctypes.c_ulonglong * (8 * NB))()
fn = gab.lib.gab_debug_conv_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 8 * NB) == 0
st = np.array(buf[:], dtype=np.int64).reshape(NB, 8)
d = np.diff(st, axis=1) * 10.0 / 1000.0
names = ["loads issued", "first data + A #1", "A #2", "A #3", "B #1", "B #2", "store"]
for i, n in enumerate(names):
    print("%-20s min %6.2f  median %6.2f  max %6.2f us" % (n, d[:, i].min(), np.median(d[:, i]), d[:, i].max()))
