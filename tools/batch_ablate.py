"""Diagnostic build only: us per buffer of gab_conv_process_batch (64 per launch) under GAB_CONV_SPLIT_DEBUG role ablations."""
import sys, os
sys.path.insert(0, ".")
import torch, gpuaudiobench_amd as gab
T, B, L, NB = 1024, 512, 4096, 64
plan = gab.ConvPlan(T, B, L); plan.set_ir(torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda())
x = torch.randn(NB * T * B, device="cuda"); y = torch.empty_like(x)
a = plan.prepare_batch(x, NB, y)
for _ in range(150): plan.launch_batch(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): plan.launch_batch(a)
e1.record(); torch.cuda.synchronize()
print("debug=%s form=%s: %.3f us per buffer" % (os.environ.get("GAB_CONV_SPLIT_DEBUG", "0"), os.environ.get("GAB_CONV_BATCH_FORM", "2"), e0.elapsed_time(e1) * 1e3 / (50 * NB)))
