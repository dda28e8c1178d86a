"""Soak check of the LDS-resident FDTD kernel's neighbour hand-off (diagnostic build: GAB_LIB_PATH=.../libgab_hip_ablate.so):
many thousand steps of a room on the resident kernel and, from the same input, on the step kernels (GAB_FDTD_RESIDENT=0,
read at plan creation), each beside another stream that keeps the chip's memory system busy.  Same operations in the
same order: the pressure fields and the outputs must agree bit for bit; one stale or torn boundary value in the
resident run would not.  python tools/fdtd_soak.py [grid] [buffers] [samples per buffer]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gpuaudiobench_amd as gab  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
buffers = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
T = 4
xs = [gab.harness.noise(T * B, seed=11 + b) for b in range(4)]
big = torch.ones(64 * 1024 * 1024, device="cuda")
hog, work = torch.cuda.Stream(), torch.cuda.Stream()
fields, outs = [], []
for resident in (1, 0):
    os.environ["GAB_FDTD_RESIDENT"] = str(resident)
    plan = gab.FdtdPlan(gab.fdtd_default_params(n))
    if plan.resident()[0] != bool(resident):
        raise SystemExit("needs the diagnostic build (GAB_FDTD_RESIDENT is read there only) and a room that fits the LDS")
    xd = [torch.from_numpy(x).cuda() for x in xs]
    out = torch.zeros(T * B, device="cuda")
    acc = torch.zeros(T * B, device="cuda")
    torch.cuda.synchronize()
    for b in range(buffers):
        with torch.cuda.stream(hog):
            for _ in range(4):
                big.mul_(1.0000001)
        with torch.cuda.stream(work):
            plan.process(xd[b % 4], out, T, B, 0, B)
            acc += out
    torch.cuda.synchronize()
    fields.append(plan.pressure().cpu().numpy().ravel())
    outs.append(acc.cpu().numpy())
    plan.close()
steps = buffers * B * 3
same_field = np.array_equal(fields[0].view(np.uint32), fields[1].view(np.uint32))
same_out = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
print("grid %d^3, %d steps: resident and step kernels leave %s pressure field (%d non-zero cells) and %s outputs"
      % (n, steps, "the SAME" if same_field else "DIFFERENT", int(np.count_nonzero(fields[0])), "the SAME" if same_out else "DIFFERENT"))
sys.exit(0 if same_field and same_out and np.count_nonzero(fields[0]) > fields[0].size // 2 else 1)
