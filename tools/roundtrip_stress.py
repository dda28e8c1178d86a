"""gab_conv_round_trip against device-buffer launches of the same cut at a large channel count, many buffers, pinned and
pageable inputs mixed; on a mismatch prints WHERE (channels, samples, channel groups) instead of only that.
    python tools/roundtrip_stress.py [channels] [buffers]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B, L = 512, 4096
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
a, b = gab.ConvPlan(T, B, L, scheme="classic"), gab.ConvPlan(T, B, L, scheme="classic")
a.set_ir(ir); b.set_ir(ir)
h_in, h_out = torch.empty(T * B).pin_memory(), torch.empty(T * B).pin_memory()
bad = 0
for i in range(N):
    x = gab.harness.noise(T * B, seed=1000 + i)
    ya = a.process(torch.from_numpy(x).cuda()).cpu().numpy()
    h_out.fill_(float("nan"))
    if i % 3 == 1:
        yb = b.round_trip(torch.from_numpy(x.copy()), h_out).numpy().copy()       # pageable input
    else:
        h_in.copy_(torch.from_numpy(x))
        yb = b.round_trip(h_in, h_out).numpy().copy()
    d = ya.view(np.uint32) != yb.view(np.uint32)
    if d.any():
        bad += 1
        idx = np.flatnonzero(d)
        smp, ch = idx // T, idx % T
        print("buffer %d (%s input): %d words differ; samples %d..%d (%d distinct), channels %d..%d (%d distinct), groups %s; NaN in round trip: %d; first: got %r want %r"
              % (i, "pageable" if i % 3 == 1 else "pinned", idx.size, smp.min(), smp.max(), np.unique(smp).size, ch.min(), ch.max(), np.unique(ch).size,
                 sorted(set((ch // 512).tolist()))[:8], int(np.isnan(yb).sum()), yb[idx[0]], ya[idx[0]]), flush=True)
        # both plans go on from the device-buffer result's state: re-synchronise b's history with a's
        b.reset(); a.reset()
print("%d buffers, %d with a mismatch" % (N, bad))
