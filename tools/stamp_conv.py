"""Diagnostic (GAB_ABLATE build): phase timeline of the streaming conv kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["GAB_CONV_ABLATE"] = "6"
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
split = os.environ.get("GAB_CONV_VARIANT") == "4"
if split:
    rel = (st - st[:, 0].min()) * 10.0 / 1000.0
    for i, n in [(0, "start"), (1, "A: data back"), (2, "A: done"), (3, "B: forward done"), (6, "B: done"), (7, "stored")]:
        print("%-16s min %6.2f  median %6.2f  max %6.2f us" % (n, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))
    half = NB // 2
    print("stored: first half median %.2f, second half median %.2f" % (np.median(rel[:half, 7]), np.median(rel[half:, 7])))
    sys.exit(0)
hw, xcc = st[:, 3].copy(), st[:, 4].copy() & 15
st[:, 3] = st[:, 0]; st[:, 4] = st[:, 0]
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
where = xcc * 1000 + se * 100 + sh * 16 + cu
print("distinct (xcc,se,sh,cu):", len(set(where.tolist())))
for b in list(range(0, 20)) + list(range(256, 266)):
    if b < NB:
        print("block %3d xcc %d se %d sh %d cu %2d wave %d simd %d" % (b, xcc[b], se[b], sh[b], cu[b], hw[b] & 15, (hw[b] >> 4) & 3))
from collections import defaultdict
groups = defaultdict(list)
for b in range(NB):
    groups[int(where[b])].append(b)
pairs = [g for g in groups.values() if len(g) == 2]
print("CUs holding 2 blocks:", len(pairs), " examples:", pairs[:8])
print("block-index difference of co-resident blocks:", sorted(set(abs(g[1] - g[0]) for g in pairs))[:20])
t0 = st[:, 0].min()
rel = (st - t0) * 10.0 / 1000.0      # 100 MHz ticks -> us
names = ["start", "first data (A inputs) back", "A done", "-", "-", "-", "B done", "stored"]
for i, n in enumerate(names):
    print("%-16s min %6.2f  median %6.2f  max %6.2f us" % (n, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))

fin = rel[:, 7]
first = rel[:, 1]
half = NB // 2
print("finish: blocks < %d: median %.2f max %.2f | blocks >= %d: median %.2f max %.2f" % (
    half, np.median(fin[:half]), fin[:half].max(), half, np.median(fin[half:]), fin[half:].max()))
print("first data: blocks < %d: median %.2f max %.2f | blocks >= %d: median %.2f max %.2f" % (
    half, np.median(first[:half]), first[:half].max(), half, np.median(first[half:]), first[half:].max()))
for x in range(8):
    m = xcc == x
    print("xcc %d: first data median %.2f  A done %.2f  B done %.2f  finish median %.2f max %.2f" % (
        x, np.median(first[m]), np.median(rel[m, 2]), np.median(rel[m, 6]), np.median(fin[m]), fin[m].max()))
print("corr(first data, finish) = %.2f" % np.corrcoef(first, fin)[0, 1])
dA = rel[:, 2] - rel[:, 1]; dB = rel[:, 6] - rel[:, 2]
print("A duration: min %.2f median %.2f max %.2f | B duration: min %.2f median %.2f max %.2f" % (
    dA.min(), np.median(dA), dA.max(), dB.min(), np.median(dB), dB.max()))
order = np.argsort(fin)[-12:]
for b in order:
    print("slow block %3d xcc %d se %d cu %2d: first %.2f A %.2f B %.2f fin %.2f" % (b, xcc[b], se[b], cu[b], first[b], rel[b, 2], rel[b, 6], fin[b]))
