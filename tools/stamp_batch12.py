"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so): barrier timeline of conv_split_batch12_kernel.  Every wave of
every workgroup stamps its ARRIVAL at each of the six barriers of periods 32 and 33 of a 64-buffer launch (two periods: the two
far groups are one period out of step).  A barrier releases when its last wave arrives: per interval, the median length and,
per wave, the median time from the previous release to its arrival (its work in that interval) and how often it was the last.
    GAB_BATCH_WAVES=12 python tools/stamp_batch12.py"""
import ctypes, os, sys
os.environ.setdefault("GAB_CONV_SPLIT_DEBUG", "64")
os.environ.setdefault("GAB_BATCH_WAVES", "12")
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T, B, L, NB = 1024, 512, 4096, 64
plan = gab.ConvPlan(T, B, L); plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
x = torch.from_numpy(np.concatenate([gab.harness.noise(T * B, seed=s) for s in range(NB)])).cuda(); y = torch.empty_like(x)
a = plan.prepare_batch(x, NB, y)
for _ in range(150): plan.launch_batch(a)
torch.cuda.synchronize()
n = 256 * 12 * 2 * 6
buf = (ctypes.c_ulonglong * n)()
fn = gab.lib.gab_debug_split_stamps; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 12, 12) * 0.01      # us; [block][wave][period-32][barrier] flattened to 12 barriers
rel = st.max(axis=1)                                                   # [block][12]: release = the last arrival
names = ["fwd0", "fwd1", "inv0", "inv1"] + ["farA%d" % i for i in range(4)] + ["farB%d" % i for i in range(4)]
iv = np.diff(rel, axis=1)                                              # 11 intervals: barrier 2 of period 32 .. barrier 6 of period 33
lab = ["32.%d" % (i + 1) for i in range(1, 6)] + ["33.%d" % (i + 1) for i in range(6)]
print("interval (ends at barrier):   " + "  ".join("%5s" % l for l in lab))
print("median length, us:            " + "  ".join("%5.2f" % np.median(iv[:, i]) for i in range(11)))
print("period 33 = %.2f us (median of the sum of its six intervals)" % np.median(rel[:, 11] - rel[:, 5]))
work = st[:, :, 1:] - rel[:, None, :-1]                                # [block][wave][11]: previous release -> this wave's arrival
last = (st[:, :, 1:] == rel[:, None, 1:])
for wv in range(12):
    print("%-6s work: " % names[wv] + "  ".join("%5.2f" % np.median(work[:, wv, i]) for i in range(11)) +
          "   last: " + "  ".join("%3.0f%%" % (100 * last[:, wv, i].mean()) for i in range(11)))
