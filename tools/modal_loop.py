"""Times the real modal bank: n_modes x bufsize, 32 output tracks."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
p = torch.from_numpy(oracle.modal_params(n)).cuda()
from gpuaudiobench_amd import ops
ws = ops.modal_bank_workspace(n, B, T)
y = torch.empty(T * B, device="cuda")
for _ in range(3):
    gab.modal_bank(p, n, B, T, out=y, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
R = 20
for _ in range(R):
    gab.modal_bank(p, n, B, T, out=y, workspace=ws)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / R
print("modes %d x %d samples: %.1f us per buffer, %.2f G mode-samples/s, %.1f TFLOP/s (9 flop per mode-sample)"
      % (n, B, us, n * B / us / 1e3, 9.0 * n * B / us / 1e6))
