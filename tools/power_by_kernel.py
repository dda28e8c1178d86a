"""Socket power and shader clock (rocm-smi, a child process) beside steady loops of different launches on ONE box: the gain kernel
at 65 536 x 512 (a pure HBM stream), the headline batch launch, the doorbell engine's pipelined run — what the board's power limit
means for each.     python tools/power_by_kernel.py        (profiles/r06_box_clocks.txt)"""
import os, re, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
dev = torch.device("cuda:0")
SMI = [sys.executable, os.path.realpath("/opt/rocm/bin/rocm-smi"), "--showclocks", "--showpower", "--showtemp"]


def sample(step, name, warm=6.0):
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < warm:
        step(); torch.cuda.synchronize(); n += 1
    rate = n / (time.perf_counter() - t0)
    child = subprocess.Popen(SMI, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    while child.poll() is None:
        step(); torch.cuda.synchronize()
    text = "\n".join(l for l in child.communicate()[0].splitlines() if l.startswith("GPU[0]"))
    g = lambda p: (re.search(p, text) or [None, "?"])[1]
    print("%-64s sclk %s MHz  mclk %s MHz  power %s W  junction %s C   (%.0f steps/s)" % (
        name, g(r"sclk clock level: *\d+: *\((\d+)Mhz\)"), g(r"mclk clock level: *\d+: *\((\d+)Mhz\)"), g(r"Power \(W\): *([\d.]+)"),
        g(r"Sensor junction\) \(C\): *([\d.]+)"), rate), flush=True)


x = torch.empty(65536 * 512, device=dev).uniform_(-1, 1); y = torch.empty_like(x)
sample(lambda: [gab.ops.gain(x, 2.0, out=y) for _ in range(16)], "gain, 65 536 x 512 (268 MB per launch, 16 launches per step)")
del x, y
T, B, L, NB = 1024, 512, 4096, 128
plan = gab.ConvPlan(T, B, L); plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).to(dev))
xb = torch.empty(NB * T * B, device=dev).uniform_(-1, 1); yb = torch.empty_like(xb)
args = plan.prepare_batch(xb, NB, yb)
sample(lambda: [plan.launch_batch(args) for _ in range(8)], "conv_split_batch12_kernel, 128 buffers per launch (8 launches per step)")
one = [plan.prepare(xb[i * T * B:(i + 1) * T * B], yb[:T * B], gab.CONV_STREAMING) for i in range(NB)]
sample(lambda: [plan.launch(one[i]) for i in range(NB)], "conv_split_kernel, one launch per buffer (128 launches per step)")
plan.close()
