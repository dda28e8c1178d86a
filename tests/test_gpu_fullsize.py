"""GPU parity at BASELINE.json's full sizes, against the CPU oracle (not against
the HIP path itself):

* conv1d_accel C3 (4096 taps x 1024 channels x 512) in STEADY STATE — ten
  streamed buffers, every channel, vs the float64 direct form
  (cuda/bench_conv1d_accel.cu:234-252 extended with carried history);
* the split kernel's large-grid block -> (role, duo) mapping (grids that are a
  multiple of 512 workgroups: T = 2048, 4096, 8192) on sampled channel duos;
* fdtd3d C4 as BASELINE states it: 128^3 for 1000 leapfrog steps
  (cuda/bench_fdtd3d.cu:14-139, 384-438), output AND pressure field bit-exact.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def gab():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import gpuaudiobench_amd as g
    return g


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


class ChunkedStreamOracle:
    """orc_conv_accel_stream_f64 on a fixed set of channels, cut over host threads (the library
    call releases the interpreter lock; channels are independent)."""

    def __init__(self, orc, ir_rows, L, B):
        self.orc, self.L, self.B = orc, L, B
        n = ir_rows.shape[0]
        k = min(_threads(), n)
        edges = [n * i // k for i in range(k + 1)]
        self.cuts = [(edges[i], edges[i + 1]) for i in range(k)]
        self.irs = [np.ascontiguousarray(ir_rows[lo:hi]).ravel() for lo, hi in self.cuts]
        self.hists = [np.zeros((hi - lo) * L, np.float32) for lo, hi in self.cuts]
        self.pool = ThreadPoolExecutor(max_workers=k)

    def step(self, x_rows):
        """x_rows: (channels, B) -> (B, channels) sample-major float64-accumulated reference."""
        def one(i):
            lo, hi = self.cuts[i]
            return self.orc.conv_accel_stream(np.ascontiguousarray(x_rows[lo:hi]).ravel(), self.irs[i],
                                              self.hists[i], self.L, self.B, hi - lo, f64=True)
        parts = list(self.pool.map(one, range(len(self.cuts))))
        return np.concatenate([p.reshape(self.B, -1) for p in parts], axis=1)

    def close(self):
        self.pool.shutdown()


def test_conv_accel_c3_steady_state_every_channel(gab, orc):
    """BASELINE C3 at full size for 12 buffers: the carry ring, the far role and the parked shares
    all contribute from buffer 2 on; every channel is compared with the float64 direct form."""
    T, B, L, nbuf = 1024, 512, 4096, 12
    ir = orc.conv_accel_ir(L, T)
    ref = ChunkedStreamOracle(orc, ir.reshape(T, L), L, B)
    plan = gab.ConvPlan(T, B, L)
    assert plan.scheme == "split"
    plan.set_ir(dev(ir))
    max_abs, peak = 0.0, 0.0
    for n in range(nbuf):
        x = orc.noise(T * B, seed=4200 + n)
        y = host(plan.process(dev(x), mode=gab.CONV_STREAMING)).reshape(B, T)
        r = ref.step(x.reshape(T, B))
        err = float(np.abs(y - r).max())
        if n == 0 or n >= 8:                       # per-buffer gate once the window is full (and on the
            assert err <= TOL * np.abs(r).max(), n  # reference-semantics first buffer)
        max_abs, peak = max(max_abs, err), max(peak, float(np.abs(r).max()))
    assert max_abs <= TOL * peak
    assert peak > 1e-4                              # a developed steady-state signal, not the onset
    ref.close()
    plan.close()


@pytest.mark.parametrize("T", [2048, 4096, 8192])
def test_conv_accel_split_large_grid_role_mapping(gab, orc, T):
    """Grids that are a multiple of 512 workgroups alternate the near/far roles in runs of 256
    (k_conv_accel.hip conv_split_buffer): run >= 2 is only reached from T = 2048 on.  Sampled
    channel duos — the first and last of every run of 256 workgroups plus random ones — are checked
    against the oracle with the impulse responses generated at the GLOBAL channel index; the rest of
    the channels must be finite and non-zero."""
    B, L, nbuf = 512, 4096, 11
    rng = np.random.default_rng(T)
    duos = T // 4
    picked = {0, duos - 1}
    for run_start in range(0, duos, 256):
        picked.update((run_start, min(run_start + 255, duos - 1)))
    picked.update(int(v) for v in rng.integers(0, duos, 24))
    chans = np.array(sorted(4 * d + j for d in picked for j in range(4)))
    ir_full = orc.conv_accel_ir(L, T)
    ir_rows = ir_full.reshape(T, L)[chans]
    # the oracle's own global-index generator gives the same rows
    one = chans[5]
    assert np.array_equal(orc.conv_accel_ir(L, 1, track_offset=int(one), total_tracks=T), ir_full.reshape(T, L)[one])
    ref = ChunkedStreamOracle(orc, ir_rows, L, B)
    plan = gab.ConvPlan(T, B, L, scheme="split")
    plan.set_ir(dev(ir_full))
    del ir_full
    max_abs, peak = 0.0, 0.0
    for n in range(nbuf):
        x = orc.noise(T * B, seed=77 + n)
        y = host(plan.process(dev(x), mode=gab.CONV_STREAMING)).reshape(B, T)
        r = ref.step(x.reshape(T, B)[chans])
        err = float(np.abs(y[:, chans] - r).max())
        if n == 0 or n >= 8:
            assert err <= TOL * np.abs(r).max(), n
        max_abs, peak = max(max_abs, err), max(peak, float(np.abs(r).max()))
        assert np.isfinite(y).all()
        if n >= 8:
            assert (np.abs(y).max(axis=0) > 0).all()        # no channel left unwritten
    assert max_abs <= TOL * peak
    assert plan.scheme == "split"
    ref.close()
    plan.close()


def test_conv_accel_large_grid_equals_shards(gab, orc):
    """The whole T = 4096 output, bit for bit, against four 1024-channel plans fed the same
    channels (a 512-workgroup grid, whose mapping the C3 test above pins to the oracle)."""
    T, B, L, nbuf, S = 4096, 512, 4096, 10, 1024
    ir = orc.conv_accel_ir(L, T).reshape(T, L)
    big = gab.ConvPlan(T, B, L, scheme="split")
    big.set_ir(dev(ir.ravel()))
    shards = []
    for k in range(T // S):
        p = gab.ConvPlan(S, B, L, scheme="split")
        p.set_ir(dev(ir[k * S:(k + 1) * S].ravel()))
        shards.append(p)
    for n in range(nbuf):
        x = orc.noise(T * B, seed=900 + n).reshape(T, B)
        y = host(big.process(dev(x.ravel()), mode=gab.CONV_STREAMING)).reshape(B, T)
        for k, p in enumerate(shards):
            ys = host(p.process(dev(x[k * S:(k + 1) * S].ravel()), mode=gab.CONV_STREAMING)).reshape(B, S)
            assert np.array_equal(bits(y[:, k * S:(k + 1) * S]), bits(ys)), (n, k)
    for p in shards + [big]:
        p.close()


def test_conv_accel_reset_then_launch_on_another_stream_without_host_sync(gab, orc):
    """gab_conv_reset queues its memsets on the caller's stream; launches on ANOTHER stream must
    still see the cleared rings (the library orders them with events).  No host synchronisation
    between the dirtying stream, the reset and the launches that follow."""
    import torch
    T, B, L = 1024, 512, 4096
    ir = dev(orc.conv_accel_ir(L, T))
    xs = [dev(orc.noise(T * B, seed=3100 + i)) for i in range(4)]
    xcat = torch.cat(xs[:3])
    a, b = gab.ConvPlan(T, B, L, scheme="split"), gab.ConvPlan(T, B, L, scheme="split")
    a.set_ir(ir)
    b.set_ir(ir)
    side = torch.cuda.Stream()
    out = torch.empty(3 * T * B, device="cuda")
    for i in range(3):
        ya = a.process(xs[i], mode=gab.CONV_STREAMING)
    want = host(ya)
    for trial in range(5):
        for i in range(9):                              # dirty every ring slot
            b.process(xs[(i + trial) % 4], mode=gab.CONV_STREAMING)
        b.reset()
        with torch.cuda.stream(side):
            if trial % 2:
                b.process_batch(xcat, 3, out=out)
            else:
                for i in range(3):
                    b.process(xs[i], out=out[2 * T * B:], mode=gab.CONV_STREAMING)
        torch.cuda.synchronize()
        assert np.array_equal(bits(host(out[2 * T * B:])), bits(want)), trial
    a.close()
    b.close()


# ---------------------------------------------------------------------------
def test_fdtd_c4_as_baseline_states_it(gab, orc):
    """BASELINE C4: 128^3 grid, 1000 leapfrog steps (334 samples x 3 steps = 1002).  Output and
    pressure field bit-exact against orc_fdtd(fused) — the restated kernels of
    cuda/bench_fdtd3d.cu:14-139 in the order of runFDTD3DTimeStep (:384-438) — for BOTH forms the room can take:
    the LDS-resident kernel (gab_fdtd_set_form AUTO: what this grid takes by default, one launch per call) and the
    step kernels (STEP: one launch per step, the LDS-halo kernels — also what a plan falls back to after a resident
    launch has timed out), each for the full 1002 steps.  By then the front has crossed the room several times: the
    receiver hears it and the planes next to the damped shell hold developed values on every face."""
    import torch
    n, T, B = 128, 8, 334
    P = orc.fdtd_params(n)
    G = gab.fdtd_default_params(n)
    x = orc.Rand(1).bipolar(T * B)
    grids = orc.fdtd_grids(P)
    ref = np.zeros(T * B, np.float32)
    calls = ((0, 100), (100, 134), (234, 100))                   # state carries across calls
    for first, cnt in calls:
        orc.fdtd(P, grids, x, ref, T, B, first, cnt, fused=True)
    p_ref = grids[0].reshape(n, n, n)
    xd = dev(x)
    for form in ("auto", "step"):
        plan = gab.FdtdPlan(G)
        plan.set_form(form)
        assert plan.resident()[0] == (form == "auto")
        out = torch.zeros(T * B, device="cuda")
        for first, cnt in calls:
            plan.process(xd, out, T, B, first, cnt)
            plan.status()                                        # synchronises; raises if the launch had given up
        got = host(out)
        assert np.array_equal(bits(got), bits(ref)), form
        p_got = host(plan.pressure())
        assert np.array_equal(bits(p_got.ravel()), bits(p_ref.ravel())), form
        plan.close()
    # the run is long enough to mean something
    assert np.count_nonzero(ref.reshape(T, B)[0]) > 200          # the receiver hears the source
    assert np.abs(ref).max() > 1e-6
    # the outermost shell only ever sees p *= 0.8 from zero, so it is identically zero by
    # construction (bench_fdtd3d.cu:86-97); the planes next to it carry the reflected field
    assert not p_ref[0].any() and not p_ref[:, 0].any() and not p_ref[:, :, -1].any()
    for face in (p_ref[1], p_ref[-2], p_ref[:, 1], p_ref[:, -2], p_ref[:, :, 1], p_ref[:, :, -2]):
        inner = face[1:-1, 1:-1]
        assert np.count_nonzero(np.abs(inner) > 1e-12) > 0.9 * inner.size     # developed values next to every face
        assert np.abs(inner).max() > 1e-7


def test_fdtd_c4_against_an_order_the_reference_atomics_can_produce(gab, orc):
    """The source cell: the reference lets every track atomicAdd 0.1f*in[t,s] into p[src] (cuda/bench_fdtd3d.cu:101-120,
    order unspecified), i.e. the samples accumulate INTO THE CELL one by one; the kernels (and orc_fdtd, which the
    bit-exact tests pin) add the tracks' own sum once — a grouping no atomic order produces, at most the cell's last
    bit apart per sample.  What that choice is worth over C4's 1002 steps, against two orders the reference CAN produce
    (ascending and descending tracks; orc_fdtd_trackwise): both forms of the room within 1e-5 of the output's peak."""
    import torch
    n, T, B = 128, 8, 334
    P = orc.fdtd_params(n)
    G = gab.fdtd_default_params(n)
    x = orc.Rand(1).bipolar(T * B)
    refs = []
    for order in (None, list(range(T - 1, -1, -1))):
        grids = orc.fdtd_grids(P)
        ref = np.zeros(T * B, np.float32)
        orc.fdtd_trackwise(P, grids, x, ref, T, B, 0, B, fused=True, order=order)
        refs.append((ref, grids[0].copy()))
    xd = dev(x)
    for form in ("auto", "step"):
        plan = gab.FdtdPlan(G)
        plan.set_form(form)
        out = torch.zeros(T * B, device="cuda")
        plan.process(xd, out, T, B, 0, B)
        plan.status()
        got, p_got = host(out), host(plan.pressure()).ravel()
        for ref, p_ref in refs:
            peak = float(np.abs(ref).max())
            assert peak > 1e-6
            assert float(np.abs(got - ref).max()) <= 1e-5 * peak, (form, float(np.abs(got - ref).max()) / peak)     # tolerance: 1e-5 of peak
            ppeak = float(np.abs(p_ref).max())
            assert float(np.abs(p_got - p_ref).max()) <= 1e-5 * ppeak, form
        plan.close()
    # the two atomic orders themselves differ from each other by no more than that
    assert float(np.abs(refs[0][0] - refs[1][0]).max()) <= 1e-5 * float(np.abs(refs[0][0]).max())


def test_fdtd_resident_launch_that_gives_up_fails_at_that_call():
    """Diagnostic build (workgroup 0 of the resident kernel never publishes: GAB_FDTD_RES_ABLATE=2), in a child
    process: the neighbours' bounded polls give up, the launch ENDS, THAT call's output is NaN in every sample,
    gab_fdtd_status reports GAB_ERR_RUNTIME for it (once), and after a reset the
    plan runs the step kernels and matches the oracle bit for bit (tools/fdtd_timeout_check.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "gpuaudiobench_amd", "libgab_hip_ablate.so")
    if not os.path.exists(lib):
        pytest.skip("the diagnostic library is not built (GAB_BUILD_TAG=ablate GAB_ABLATE=1 python gpuaudiobench_amd/build.py)")
    env = dict(os.environ, GAB_LIB_PATH=lib, GAB_FDTD_RES_ABLATE="2")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fdtd_timeout_check.py")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "matches the oracle bit for bit: ok" in r.stdout


def test_round_trips_whose_input_never_lands_fail_at_that_call_and_work_again():
    """Diagnostic build (GAB_RT_SKIP_UPLOAD=1: no upload, nothing announced as landed), in a child process: the
    bounded waits inside gab_datatransfer_round_trip's and gab_conv_round_trip's launches run out, the launches END,
    the calls return GAB_ERR_RUNTIME; the staging buffers are re-armed: the next calls (the convolver after
    gab_conv_reset) are right again, bit for bit (tools/round_trip_timeout_check.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "gpuaudiobench_amd", "libgab_hip_ablate.so")
    if not os.path.exists(lib):
        pytest.skip("the diagnostic library is not built (GAB_BUILD_TAG=ablate GAB_ABLATE=1 python gpuaudiobench_amd/build.py)")
    env = dict(os.environ, GAB_LIB_PATH=lib)
    env.pop("GAB_RT_SKIP_UPLOAD", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "round_trip_timeout_check.py")], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "datatransfer: the plan is right again: ok" in r.stdout
    assert "conv: after the reset the plan matches device-buffer launches bit for bit: ok" in r.stdout


def test_fdtd_wide_slab_cut_developed_field(gab, orc):
    """One >= 68-wide slab cut (rows that take the LDS-halo step kernel) run long enough for the
    front to cross the cut planes many times: 96^3, three uneven slabs, 120 samples = 360 steps,
    bit-exact against the oracle's single grid."""
    import torch
    from gpuaudiobench_amd import fdtd_slabs
    n, T, B = 96, 4, 120
    P = orc.fdtd_params(n)
    G = gab.fdtd_default_params(n)
    x = orc.Rand(1).bipolar(T * B)
    grids = orc.fdtd_grids(P)
    ref = np.zeros(T * B, np.float32)
    orc.fdtd(P, grids, x, ref, T, B, 0, B, fused=True)
    slabs = [fdtd_slabs.FdtdSlab(G, a, b) for a, b in [(0, 30), (30, 71), (71, 96)]]
    out = torch.zeros(T * B, device="cuda")
    fdtd_slabs.process_local(slabs, dev(x), out, T, B)
    field = torch.cat([s.pressure() for s in slabs]).cpu().numpy().ravel()
    assert np.array_equal(bits(host(out)), bits(ref))
    assert np.array_equal(bits(field), bits(grids[0]))
    assert np.count_nonzero(ref.reshape(T, B)[0]) > 60
    for s in slabs:
        s.close()


def test_conv_accel_mixed_launch_paths_stay_bit_identical(gab, orc):
    """A random mix of the ways a streaming buffer can be launched — one launch, a batch of 1-9
    buffers, pinned host buffers, on changing streams, resets in between, no host synchronisation
    except where a result is read — walks the same history as a plan that only ever sees
    gab_conv_process on device buffers: same bits, every buffer."""
    import torch
    T, B, L = 256, 512, 4096
    ir = dev(orc.conv_accel_ir(L, T))
    ref, mix = gab.ConvPlan(T, B, L, scheme="split"), gab.ConvPlan(T, B, L, scheme="split")
    ref.set_ir(ir)
    mix.set_ir(ir)
    hx = [orc.noise(T * B, seed=5000 + i) for i in range(6)]
    xs = [dev(x) for x in hx]
    pinned = [torch.from_numpy(x).pin_memory() for x in hx]
    h_out = torch.empty(T * B).pin_memory()
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    rng = np.random.default_rng(11)
    n = 0
    for step in range(60):
        how = int(rng.integers(0, 4))
        if how == 3 and step > 0:
            ref.reset()
            mix.reset()
            continue
        count = int(rng.integers(1, 10)) if how == 1 else 1
        for j in range(count):
            want = ref.process(xs[(n + j) % 6], mode=gab.CONV_STREAMING)
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[int(rng.integers(0, 3))]):
            if how == 0:
                got = mix.process(xs[n % 6], mode=gab.CONV_STREAMING)
            elif how == 1:
                xb = torch.cat([xs[(n + j) % 6] for j in range(count)])
                got = mix.process_batch(xb, count)[(count - 1) * T * B:]
            else:
                mix.process(pinned[n % 6], out=h_out, mode=gab.CONV_STREAMING)
                got = h_out
        n += count
        torch.cuda.synchronize()
        assert mix.scheme == "split"
        assert np.array_equal(bits(got.cpu().numpy()), bits(host(want))), (step, how)
    ref.close()
    mix.close()


@pytest.mark.parametrize("nx,ny,nz,samples", [
    (48, 20, 36, 30),      # one launch per sample (tile kernel), clipped boxes on every side
    (52, 33, 7, 24),       # fewer planes than a box is tall; odd row count
    (21, 56, 40, 20),      # rows that are not a multiple of 4 cells
    (100, 72, 36, 12),     # LDS-halo step kernel, non-cubic
    (36, 100, 24, 12),     # wider than 56 in y only: one launch per step
])
def test_fdtd_non_cubic_rooms_bit_exact(gab, orc, nx, ny, nz, samples):
    """Rooms that are not cubes (the reference's parameters are per axis, bench_fdtd3d.cuh:68-86):
    every kernel form, output and pressure field bit-exact, state carried across two calls."""
    import torch
    T, B = 5, samples
    P = orc.fdtd_params(nx, ny, nz)
    G = gab.fdtd_default_params(nx, ny, nz)
    x = orc.Rand(7).bipolar(T * B)
    grids = orc.fdtd_grids(P)
    ref = np.zeros(T * B, np.float32)
    plan = gab.FdtdPlan(G)
    out = torch.zeros(T * B, device="cuda")
    half = samples // 2
    for first, cnt in ((0, half), (half, samples - half)):
        orc.fdtd(P, grids, x, ref, T, B, first, cnt, fused=True)
        plan.process(dev(x), out, T, B, first, cnt)
    assert np.array_equal(bits(host(out)), bits(ref))
    assert np.array_equal(bits(host(plan.pressure()).ravel()), bits(grids[0]))
    assert np.abs(grids[0]).max() > 0
    plan.close()


@pytest.mark.parametrize("ranks,distribution", [(1, "broadcast"), (1, "slices"), (2, "slices")])
def test_bench_under_a_launcher_walks_the_rccl_path_on_one_gpu(ranks, distribution):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the rank creates the RCCL
    process group (backend nccl), distributes the impulse-response bank through it (as a broadcast, or as per-rank
    slices), meets the barriers and all-reduces the timing — the N > 1 code path with one rank, on the hardware the
    box has — and the line still carries the in-run parity check.  Two ranks: GAB_BENCH_REHEARSE=1, both on device 0
    over gloo (the rate means nothing then, the line says so): rank 1's rows of the bank arrive by a send, its input rows
    and the parity check at GLOBAL channel indices."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ranks > 1:
        env["GAB_BENCH_REHEARSE"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--clock-warm-steps", "5",
                        "--no-side-legs", "--no-cpu-baseline", "--ir-distribution", distribution],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    d = json.loads(line)
    assert d["n_gpus"] == ranks and d["config"]["collective_backend"] == ("nccl" if ranks == 1 else "gloo")
    assert d["config"]["ir_distribution"] == distribution and d["config"]["channels_total"] == 1024 * ranks
    assert d["config"]["ir_broadcast_ms"] is not None and d["config"]["ir_broadcast_ms"] >= 0
    assert d["parity_checked"]["ok"] is True and d["value"] > 0
    pr = d["config"]["per_rank"]                       # every rank's own launch period, bank time and bytes received
    assert len(pr["us_per_buffer"]) == ranks and all(v > 0 for v in pr["us_per_buffer"]) and len(pr["ir_broadcast_ms"]) == ranks
    bank = 4 * 4096 * 1024 * ranks
    assert pr["ir_bytes_received"] == ([0] if ranks == 1 else [0, bank if distribution == "broadcast" else bank // 2])
