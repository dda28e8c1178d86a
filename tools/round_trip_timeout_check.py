"""Diagnostic build only (GAB_LIB_PATH=.../libgab_hip_ablate.so).  With GAB_RT_SKIP_UPLOAD=1 neither round trip uploads
its input and nothing is announced as landed: every wait inside the launches must run out, the launches must END, the
calls must return GAB_ERR_RUNTIME — and the next calls (variable removed) on the same plans must be right again, the
convolver after a gab_conv_reset.  Run in a child process by the test suite."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
import oracle as orc
orc.build()
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)

# ---- datatransfer
n_in, n_out = 50000, 70000
plan = gab.LinkPlan(n_in)
x = orc.Rand(3).unit(n_in)
h_in, h_out = torch.from_numpy(x).pin_memory(), torch.zeros(n_out).pin_memory()
plan.round_trip(h_in, h_out)
want = h_out.numpy().copy()
assert np.array_equal(bits(want[:n_in]), bits(x))
os.environ["GAB_RT_SKIP_UPLOAD"] = "1"
t0 = time.time()
try:
    plan.round_trip(h_in, h_out)
    raise SystemExit("datatransfer: the call without an upload did not fail")
except gab.GabError as e:
    print("datatransfer: failed at that call after %.1f s: %s" % (time.time() - t0, str(e)[:90]))
del os.environ["GAB_RT_SKIP_UPLOAD"]
for _ in range(3):
    h_out.zero_()
    plan.round_trip(h_in, h_out)
    assert np.array_equal(bits(h_out.numpy()), bits(want))
print("datatransfer: the plan is right again: ok")
plan.close()

# ---- the convolver's round trip (classic cut)
T, B, L = 64, 512, 4096
ir = gab.harness.conv_accel_ir(L, T)
a, b = gab.ConvPlan(T, B, L, scheme="classic"), gab.ConvPlan(T, B, L, scheme="classic")
a.set_ir(torch.from_numpy(ir).cuda()); b.set_ir(torch.from_numpy(ir).cuda())
xs = [gab.harness.noise(T * B, seed=s) for s in range(4)]
hi = [torch.from_numpy(v).pin_memory() for v in xs]
ho = torch.zeros(T * B).pin_memory()
a.round_trip(hi[0], ho)
os.environ["GAB_RT_SKIP_UPLOAD"] = "1"
t0 = time.time()
try:
    a.round_trip(hi[1], ho)
    raise SystemExit("conv: the call without an upload did not fail")
except gab.GabError as e:
    print("conv: failed at that call after %.1f s: %s" % (time.time() - t0, str(e)[:90]))
del os.environ["GAB_RT_SKIP_UPLOAD"]
a.reset()
for k in range(4):                                   # a fresh stream on the re-armed plan against device-buffer launches
    a.round_trip(hi[k], ho)
    ref = b.process(torch.from_numpy(xs[k]).cuda()).cpu().numpy()
    assert np.array_equal(bits(ho.numpy()), bits(ref)), "buffer %d" % k
print("conv: after the reset the plan matches device-buffer launches bit for bit: ok")
