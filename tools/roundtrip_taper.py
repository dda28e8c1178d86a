"""gab_conv_round_trip with the channel groups cut in different ways (diagnostic build: GAB_RT_TAPER is read when a plan's
round-trip state is created): host clock per call and the device timeline's last marks.
    GAB_LIB_PATH=gpuaudiobench_amd/libgab_hip_ablate.so python tools/roundtrip_taper.py [channels] [calls]"""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 520
NAMES = {0: "16 equal groups (round 4)", 1: "first and last group cut into quarter, quarter, half", 2: "the first group only", 3: "the last group only"}
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B).pin_memory()
buf = (C.c_ulonglong * (65 * 4))()
for rep in range(2):
    for mode in (0, 1, 2, 3):
        os.environ["GAB_RT_TAPER"] = str(mode)
        plan = gab.ConvPlan(T, B, L, scheme="classic")
        plan.set_ir(ir)
        args = plan.prepare_round_trip(h_in, h_out)
        ts = []
        for i in range(N):
            t0 = time.perf_counter(); plan.launch_round_trip(args); ts.append((time.perf_counter() - t0) * 1e6)
        ts = np.array(ts[20:])
        marks = []
        for _ in range(51):
            gab.lib.gab_debug_rt_stamps(buf, 1)
            plan.launch_round_trip(args)
            gab.lib.gab_debug_rt_stamps(buf, 0)
            a = np.array(buf[:], dtype=np.uint64).reshape(65, 4)
            G = int((a[:64, 0] != np.uint64(0xffffffffffffffff)).sum())
            t00 = a[:G, 0].min()
            rel = (a[:G].astype(np.int64) - np.int64(t00)) / 100.0
            marks.append([rel[0, 1], rel[0, 2], rel[0, 3], rel[G - 1, 1], rel[G - 1, 2], rel[G - 1, 3], (np.int64(a[64, 0]) - np.int64(t00)) / 100.0, G])
        m = np.median(np.array(marks), axis=0)
        print("pass %d  taper %d  p50 %6.1f us  p95 %6.1f   groups %2d: first landed %5.1f drain starts %5.1f done %5.1f | last landed %5.1f drain starts %5.1f done %5.1f | word %5.1f   %s"
              % (rep, mode, np.percentile(ts, 50), np.percentile(ts, 95), int(m[7]), m[0], m[1], m[2], m[3], m[4], m[5], m[6], NAMES[mode]), flush=True)
        plan.close()
