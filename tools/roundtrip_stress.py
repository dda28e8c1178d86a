"""gab_conv_round_trip against device-buffer launches of the same cut at a large channel count, many buffers, pinned and
pageable inputs mixed.  Every compared call is set up so that ONE mismatch classifies itself (tests/rt_diag.py): h_out is
refilled with NaN, the block the kernel consumed is read back and compared with h_in, the previous call's input and
output are kept.
    python tools/roundtrip_stress.py [channels] [buffers]        (KEEP_WARM=1: with gab_conv_round_trip_keep_warm on)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpuaudiobench_amd as gab
import rt_diag
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B, L = 512, 4096
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
a, b = gab.ConvPlan(T, B, L, scheme="classic"), gab.ConvPlan(T, B, L, scheme="classic")
a.set_ir(ir); b.set_ir(ir)
if os.environ.get("KEEP_WARM"):
    b.round_trip_keep_warm(True)
h_in, h_out = torch.empty(T * B).pin_memory(), torch.empty(T * B).pin_memory()
bad = 0
prev_out = prev_in = None
for i in range(N):
    x = gab.harness.noise(T * B, seed=1000 + i)
    ya = a.process(torch.from_numpy(x).cuda()).cpu().numpy()
    h_out.fill_(float("nan"))
    pageable = i % 3 == 1
    if pageable:
        yb = b.round_trip(torch.from_numpy(x.copy()), h_out).numpy().copy()
    else:
        h_in.copy_(torch.from_numpy(x))
        yb = b.round_trip(h_in, h_out).numpy().copy()
    consumed = b.newest_block().cpu().numpy()
    report = rt_diag.classify(ya, yb, T, B, prev_out=prev_out, h_in=x, consumed=consumed, prev_in=prev_in,
                              label="buffer %d (%s input)" % (i, "pageable" if pageable else "pinned"))
    if report:
        bad += 1
        print(report, flush=True)
        # both plans go on from the device-buffer result's state: re-synchronise b's history with a's
        b.reset(); a.reset()
        prev_out = prev_in = None
    else:
        prev_out, prev_in = yb, x
print("%d buffers at %d channels, %d with a mismatch" % (N, T, bad))
sys.exit(1 if bad else 0)
