"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so): where one workgroup of conv1d_direct_kernel spends its
time — s_memtime (shader clocks) at: tile entered, window staged (after the barrier), chain done, back in the kernel."""
import ctypes, sys
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
fn = gab.lib.gab_debug_conv1d_stamps; fn.argtypes = [ctypes.c_void_p]
for T, L in ((256, 256), (128, 1024), (128, 4096)):
    B = 512
    x = torch.from_numpy(gab.harness.noise(T * B, seed=1)).cuda()
    ir = torch.from_numpy(gab.harness.noise(T * L, seed=2)).cuda()
    rows = []
    for rep in range(30):
        gab.conv1d(x, ir, L, T, B)
        buf = (ctypes.c_ulonglong * 4)()
        assert fn(buf) == 0
        s = np.array(buf[:], dtype=np.int64)
        rows.append(np.diff(s))
    d = np.median(np.array(rows[5:]), axis=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): gab.conv1d(x, ir, L, T, B)
    e1.record(); torch.cuda.synchronize()
    print("T=%d L=%d: stage %d clk, chain %d clk (%.1f clk per tap), exit %d clk; back-to-back launches %.2f us each"
          % (T, L, d[0], d[1], d[1] / min(L, 1024), d[2], e0.elapsed_time(e1) * 1e3 / 200), flush=True)
