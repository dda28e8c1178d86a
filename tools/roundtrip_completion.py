"""How gab_conv_round_trip learns that its launch has ended, priced on the host clock (diagnostic build: the rule is
picked per call through GAB_RT_COMPLETION; the product library has ONE rule compiled in, kRtCompletion).
    GAB_LIB_PATH=gpuaudiobench_amd/libgab_hip_ablate.so python tools/roundtrip_completion.py [channels] [calls]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 520
NAMES = {0: "hipStreamSynchronize", 1: "event recorded behind the launch, hipEventQuery", 2: "the launch's own stop event (hipExtLaunchKernelGGL), hipEventQuery",
         3: "hipStreamQuery", 9: "round 4's rule: the pinned hint word alone (NOT a stated guarantee)"}
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B).pin_memory()
plan = gab.ConvPlan(T, B, L, scheme="classic")
plan.set_ir(ir)
args = plan.prepare_round_trip(h_in, h_out)
ref = None
for rep in range(2):
    for mode in (0, 1, 2, 3, 9):
        os.environ["GAB_RT_COMPLETION"] = str(mode)
        plan.reset(); torch.cuda.synchronize()
        ts = []
        for i in range(N):
            t0 = time.perf_counter(); plan.launch_round_trip(args); ts.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
        ts = np.array(ts[20:])
        plan.reset()
        for _ in range(10):
            plan.launch_round_trip(args)
        torch.cuda.synchronize()
        same = True if ref is None else bool(torch.equal(ref.view(torch.int32), h_out.view(torch.int32)))
        if ref is None:
            ref = h_out.clone()
        print("pass %d  mode %d  p50 %6.1f us  p95 %6.1f  min %6.1f  max %7.1f  same bits %s   %s"
              % (rep, mode, np.percentile(ts, 50), np.percentile(ts, 95), ts.min(), ts.max(), same, NAMES[mode]), flush=True)
plan.close()
