"""Diagnostic (GAB_ABLATE build): phase timeline of conv_split_kernel over two consecutive launches.

    GAB_BUILD_TAG=ablate GAB_ABLATE=1 python gpuaudiobench_amd/build.py
    GAB_LIB_PATH=gpuaudiobench_amd/libgab_hip_ablate.so python tools/stamp_split.py [T] [drain]
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
drain = len(sys.argv) > 2 and sys.argv[2] == "drain"
os.environ["GAB_CONV_SPLIT_DEBUG"] = str(64 + (128 if drain else 0))
os.environ["GAB_CONV_STAMP_AT"] = "1200"          # range launches: stamp buffers 1200 and 1201 of 2001 (mid-run)
import gpuaudiobench_amd as gab
B, L = 512, 4096
plan = gab.ConvPlan(T, B, L, scheme="split")
plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
xs = [torch.from_numpy(gab.harness.noise(T * B, seed=s)).cuda() for s in range(8)]
out = torch.empty(T * B, device="cuda")
args = [plan.prepare(x, out) for x in xs]
N = 2001                      # odd: the last launch has head parity 0, the one before parity 1
for i in range(N):
    plan.launch(args[i % 8])
torch.cuda.synchronize()
NB = T // 2
buf = (ctypes.c_ulonglong * (2 * 8192 * 8))()
fn = gab.lib.gab_debug_split_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 2 * 8192 * 8) == 0
st = np.array(buf[:], dtype=np.int64).reshape(2, 8192, 8)[:, :NB]
prev, last = st[1], st[0]                     # launch N-2 (parity 1), launch N-1 (parity 0)
if (N - 1) % 2 == 1:
    prev, last = st[0], st[1]
duos = NB // 2
if NB % 512 == 0:
    run = np.arange(NB) >> 8
    far = (run & 1) != 0
else:
    far = np.arange(NB) >= duos
t0 = prev[:, 0].min()
us = lambda a: (a - t0) / 100.0
names = {False: ["entry", "loads issued" + ("+drained" if drain else ""), "forward done", "product + barrier", "inverse done", "stores issued"],
         True: ["entry", "loads issued" + ("+drained" if drain else ""), "forward done", "product done", "inverse done", "stores issued"]}
for label, L_ in (("launch k", prev), ("launch k+1", last)):
    for role in (False, True):
        m = far == role
        print("%s, %s workgroups (%d):" % (label, "far" if role else "near", m.sum()))
        for i, n in enumerate(names[role]):
            v = us(L_[m, i])
            print("   %-22s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (
                n, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
print("launch k: last stamp %.2f us;  launch k+1: first entry %.2f us  (gap %.2f us);  period %.2f us" % (
    us(prev[:, 5]).max(), us(last[:, 0]).min(), us(last[:, 0]).min() - us(prev[:, 5]).max(),
    us(last[:, 0]).min() - us(prev[:, 0]).min()))
xcc = prev[:, 7] & 15
for x in range(8):
    m = xcc == x
    print("xcc %d: %3d wgs (%3d far)  entry median %.2f  end median %.2f max %.2f" % (
        x, m.sum(), (m & far).sum(), np.median(us(prev[m, 0])), np.median(us(prev[m, 5])), us(prev[m, 5]).max()))

# ---- the same plan launched as two channel ranges on two streams (gab_conv_stream_ranges): are the
# two kernels of a buffer on the device at the same time?  Device-side clock, so no tracer in the way.
if T % 8 == 0 and len(sys.argv) > 2 and sys.argv[-1] == "ranges":
    plan.reset()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    halves = [(0, T // 2), (T // 2, T // 2)]
    plan.stream_ranges(xs, out, halves, streams, 2001)
    torch.cuda.synchronize()
    assert fn(buf, 2 * 8192 * 8) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(2, 8192, 8)[:, :NB]
    rows = {}
    for parity, label in ((0, "buffer k"), (1, "buffer k+1")):        # buffers 1200 (head parity 0) and 1201
        for r, (a, c) in enumerate(halves):
            blk = st[parity, a // 2:(a + c) // 2]
            rows[(label, r)] = (blk[:, 0].min(), blk[:, 5].max())
    t0 = min(v[0] for v in rows.values())
    print("two range launches per buffer on two streams (device clock, us after the earliest start):")
    for (label, r), (a, b) in sorted(rows.items(), key=lambda kv: kv[1][0]):
        print("   %-10s range %d (stream %d): first workgroup starts %6.2f, last store issued %6.2f  (%.2f us)"
              % (label, r, r, (a - t0) / 100.0, (b - t0) / 100.0, (b - a) / 100.0))
    for r in (0, 1):
        print("   stream %d: period %.2f us" % (r, (rows[("buffer k+1", r)][0] - rows[("buffer k", r)][0]) / 100.0))
