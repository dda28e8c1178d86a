"""Diagnostic (GAB_ABLATE build): phase timeline of the streaming conv kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["GAB_CONV_ABLATE"] = "6"
import gpuaudiobench_amd as gab
T, B, L = 1024, 512, 4096
plan = gab.ConvPlan(T, B, L)
plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
xs = [torch.from_numpy(gab.harness.noise(T * B, seed=s)).cuda() for s in range(4)]
out = torch.empty(T * B, device="cuda")
for i in range(20):
    plan.process(xs[i % 4], out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 512))()
fn = gab.lib.gab_debug_conv_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 8 * 512) == 0
st = np.array(buf[:], dtype=np.int64).reshape(512, 8)
t0 = st[:, 0].min()
rel = (st - t0) * 10.0 / 1000.0      # 100 MHz ticks -> us
names = ["start", "loads issued", "A done", "history landed", "B fwd done", "B product done", "B inv done", "stored"]
for i, n in enumerate(names):
    print("%-16s min %6.2f  median %6.2f  max %6.2f us" % (n, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))
