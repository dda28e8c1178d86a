"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so).  GAB_RT_TEAR=<word> makes gab_conv_round_trip show the kernel
a word whose early value is NOT what the completed upload leaves (one bit wrong in the staging buffer before the launch, the
right value only when the rest of the upload is through): what a torn or reordered engine write would look like.  The call
must return GAB_ERR_RUNTIME and say so — the consumed words are checked against the completed upload before the launch
ends — and after a reset the plan must match device-buffer launches again.  Run in a child process by the test suite; also
prints the p50 of the call with and without nothing to report (the check's cost is in the product build's p50)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)

T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 512, 4096
ir = gab.harness.conv_accel_ir(L, T)
a, b = gab.ConvPlan(T, B, L, scheme="classic"), gab.ConvPlan(T, B, L, scheme="classic")
a.set_ir(torch.from_numpy(ir).cuda()); b.set_ir(torch.from_numpy(ir).cuda())
xs = [gab.harness.noise(T * B, seed=40 + s) for s in range(6)]
hi = [torch.from_numpy(v).pin_memory() for v in xs]
ho = torch.zeros(T * B).pin_memory()
for k in range(2):                                   # two ordinary calls: right, and nothing reported
    a.round_trip(hi[k], ho)
    ref = b.process(torch.from_numpy(xs[k]).cuda()).cpu().numpy()
    assert np.array_equal(bits(ho.numpy()), bits(ref)), "buffer %d" % k
for word in (5, T * B // 2 + 3, T * B - 1):          # a word of the first group, of the middle, the very last word
    os.environ["GAB_RT_TEAR"] = str(word)
    try:
        a.round_trip(hi[2], ho)
        raise SystemExit("conv: the call whose word %d was consumed with a wrong bit did not fail" % word)
    except gab.GabError as e:
        assert "not the word the completed upload left" in str(e), str(e)
        print("conv: word %d: failed at that call: %s" % (word, str(e)[:120]))
    del os.environ["GAB_RT_TEAR"]
    a.reset(); b.reset()
    for k in range(3, 6):                            # a fresh stream on the re-armed plan against device-buffer launches
        a.round_trip(hi[k], ho)
        ref = b.process(torch.from_numpy(xs[k]).cuda()).cpu().numpy()
        assert np.array_equal(bits(ho.numpy()), bits(ref)), "after word %d, buffer %d" % (word, k)
print("conv: after each reset the plan matches device-buffer launches bit for bit: ok")
ts = []
for i in range(300):
    t0 = time.perf_counter()
    a.round_trip(hi[i % 6], ho)
    ts.append((time.perf_counter() - t0) * 1e6)
print("conv: %d channels, p50 of the call %.1f us (the check is part of every call)" % (T, np.percentile(ts[50:], 50)))
