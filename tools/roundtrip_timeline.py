"""Timeline of ONE gab_conv_round_trip call (diagnostic build: GAB_LIB_PATH=.../libgab_hip_ablate.so [GAB_RT_GROUPS=G]).
Device side: per channel group, the first workgroup's entry, the moment the group's last rows had landed, the start and
the end of its drain to the pinned output (s_memrealtime, 100 MHz, relative to the first workgroup of the launch);
host side: the call's own marks on the host clock.  Prints a table (median over the calls) and the p50 of the call."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
assert hasattr(gab.lib, "gab_debug_rt_stamps"), "needs the diagnostic build (GAB_LIB_PATH=.../libgab_hip_ablate.so)"
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
plan = gab.ConvPlan(T, B, L, scheme="classic")
plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B).pin_memory()
args = plan.prepare_round_trip(h_in, h_out)
for _ in range(50):
    plan.launch_round_trip(args)
buf = (C.c_ulonglong * (65 * 4))()
rows, walls = [], []
for _ in range(101):
    gab.lib.gab_debug_rt_stamps(buf, 1)
    t0 = time.perf_counter()
    plan.launch_round_trip(args)
    walls.append((time.perf_counter() - t0) * 1e6)
    gab.lib.gab_debug_rt_stamps(buf, 0)
    a = np.array(buf[:], dtype=np.uint64).reshape(65, 4)
    G = int((a[:64, 0] != np.uint64(0xffffffffffffffff)).sum())
    t00 = a[:G, 0].min()
    rel = (a[:G].astype(np.int64) - np.int64(t00)) / 100.0          # us
    done = (np.int64(a[64, 0]) - np.int64(t00)) / 100.0
    extra = [(np.int64(a[64, k]) - np.int64(t00)) / 100.0 if a[64, k] not in (np.uint64(0), np.uint64(0xffffffffffffffff)) else np.nan for k in (1, 2, 3)]
    rows.append(np.concatenate([rel.ravel(), [done], extra]))
rows = np.nanmedian(np.array(rows), axis=0)
G = (len(rows) - 4) // 4
print("gab_conv_round_trip, %d channels, %d groups of %d channels: host clock p50 %.1f us (p95 %.1f) per call" % (
    T, G, T // G, np.percentile(walls, 50), np.percentile(walls, 95)))
print("device marks, us after the launch's first workgroup entered (median of %d calls):" % len(walls))
print("group  entered  rows landed  drain starts  drain done")
for g in range(G):
    print("%5d  %7.1f  %11.1f  %12.1f  %10.1f" % (g, *rows[4 * g:4 * g + 4]))
print("completion word written at %.1f us" % rows[-4])
print("the consumed words' check against the completed upload (round 6): the first waiting workgroup saw `landed` at %.1f us, the last "
      "check at park time started at %.1f us, the last workgroup had checked and re-armed its rows at %.1f us" % (rows[-2], rows[-1], rows[-3]))
