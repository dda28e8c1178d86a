"""Diagnostic (GAB_ABLATE build): phase timeline of the streaming conv kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["GAB_CONV_ABLATE"] = "6"
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
t0 = st[:, 0].min()
rel = (st - t0) * 10.0 / 1000.0      # 100 MHz ticks -> us
names = ["start", "first data (A inputs) back", "A done", "-", "-", "-", "B done", "stored"]
for i, n in enumerate(names):
    print("%-16s min %6.2f  median %6.2f  max %6.2f us" % (n, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))
