"""Host cost of one ConvPlan.process() call from Python: a 2-channel plan is GPU-trivial, so the
loop rate is the host's."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
from gpuaudiobench_amd import _capi
import ctypes as C
B, L = 512, 4096
for T in (2, 128, 1024):
    plan = gab.ConvPlan(T, B, L)
    plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
    x = torch.from_numpy(gab.harness.noise(T * B, seed=1)).cuda()
    out = torch.empty(T * B, device="cuda")
    for i in range(300):
        plan.process(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5000):
        plan.process(x, out=out)
    t_issue = (time.perf_counter() - t0) * 1e6 / 5000
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e6 / 5000
    # raw ctypes call, pointers prepared once
    h, px, po = plan._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = gab.lib.gab_conv_process
    t0 = time.perf_counter()
    for i in range(5000):
        fn(h, px, po, 1, st)
    t_raw_issue = (time.perf_counter() - t0) * 1e6 / 5000
    torch.cuda.synchronize()
    t_raw_all = (time.perf_counter() - t0) * 1e6 / 5000
    print("T=%4d  process(): issue %.2f us, with drain %.2f us | raw ctypes: issue %.2f us, with drain %.2f us"
          % (T, t_issue, t_all, t_raw_issue, t_raw_all), flush=True)
