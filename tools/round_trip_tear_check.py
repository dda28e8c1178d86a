"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so).  GAB_RT_TEAR=<word> makes gab_conv_round_trip show the kernel
a word whose early value is NOT what the completed upload leaves (one bit wrong in the staging buffer before the launch, the
right value only when the rest of the upload is through): what a torn or reordered engine write would look like.  The check
launch behind the call compares the consumed words with the completed upload's: with the default (deferred) rule the torn
call returns and gab_conv_round_trip_check — or the next call — reports it; with set_check(2) the call itself fails.  After a
reset the plan must match device-buffer launches again.  Run in a child process by the test suite; also prints the p50 of the
call under both rules."""
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
for n, word in enumerate((5, T * B // 2 + 3, T * B - 1)):          # a word of the first group, of the middle, the very last word
    in_the_call = n == 1                             # the middle one with the verdict read IN the call (set_check 2)
    a.round_trip_set_check(2 if in_the_call else 1)
    os.environ["GAB_RT_TEAR"] = str(word)
    try:
        a.round_trip(hi[2], ho)
        if in_the_call:
            raise SystemExit("conv: the call whose word %d was consumed with a wrong bit did not fail" % word)
        del os.environ["GAB_RT_TEAR"]
        if n == 0:
            a.round_trip_check()                     # deferred: asked for explicitly ...
        else:
            a.round_trip(hi[3], ho)                  # ... or found by a following call on the plan: the next one if the check launch is
            a.round_trip(hi[4], ho)                  # through by then (a paced caller), the one after it for a back-to-back caller
        raise SystemExit("conv: the wrong bit in word %d was never reported" % word)
    except gab.GabError as e:
        assert "not the word the completed upload left" in str(e), str(e)
        print("conv: word %d: reported (%s): %s" % (word, "at that call" if in_the_call else ("by round_trip_check" if n == 0 else "by a following call"), str(e)[:100]))
    os.environ.pop("GAB_RT_TEAR", None)
    a.round_trip_set_check(1)
    a.reset(); b.reset()
    for k in range(3, 6):                            # a fresh stream on the re-armed plan against device-buffer launches
        a.round_trip(hi[k], ho)
        ref = b.process(torch.from_numpy(xs[k]).cuda()).cpu().numpy()
        assert np.array_equal(bits(ho.numpy()), bits(ref)), "after word %d, buffer %d" % (word, k)
print("conv: after each reset the plan matches device-buffer launches bit for bit: ok")
for mode in (1, 2, 0):
    a.round_trip_set_check(mode)
    ts = []
    for i in range(300):
        t0 = time.perf_counter()
        a.round_trip(hi[i % 6], ho)
        ts.append((time.perf_counter() - t0) * 1e6)
    print("conv: %d channels, back to back, set_check(%d): p50 of the call %.1f us" % (T, mode, np.percentile(ts[50:], 50)))
