"""tests/golden/: the committed fixtures.

reference_pins.json  — known-answer values of the reference's own CPU goldens.
derived_fixtures.npz — answers of the pinned oracle on small seeded inputs for what the
                       reference cannot pin (made by tests/golden/make_fixtures.py).
CPU tests: the oracle still reproduces both.  GPU tests: the HIP path reproduces the derived
fixtures through the C ABI without consulting the live oracle for the expected values.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "reference_pins.json")))
FIX = np.load(os.path.join(HERE, "golden", "derived_fixtures.npz"))


def f9(x):
    return float("%.9g" % float(x))


def hex_of(arr):
    return bytes(arr.tolist()).hex()


# ---------------------------------------------------------------- CPU: oracle vs reference pins
def test_reference_pins_file_against_oracle(orc):
    p = PINS["noise_seed42_n65536"]
    x = orc.noise(65536, 42)
    assert [f9(v) for v in x[:4]] == p["first4"] and f9(x[65535]) == p["at65535"]
    assert orc.fnv_survey(x) == p["fnv"]

    x = orc.noise(128 * 512)
    p = PINS["gain_c1_128x512"]
    g = orc.gain(x, 2.0)
    assert [f9(g[0]), f9(g[1])] == p["first2"] and orc.fnv_survey(g) == p["fnv"]
    assert f9(g.astype(np.float64).sum()) == p["sum"]

    p = PINS["gainstats_128x512"]
    out, stats = orc.gainstats(x, 128, 512)
    assert orc.fnv_survey(out) == p["out_fnv"] and orc.fnv_survey(stats) == p["stats_fnv"]
    assert [f9(v) for v in stats[:4]] == p["stats_first4"]

    p = PINS["iir_128x512"]
    c = orc.iir_coeffs(0.25)
    state = np.zeros(256, np.float32)
    y = orc.iir(x, c, state, 128, 512)
    assert f9(c[3]) == p["a1"] and [f9(y[0]), f9(y[1])] == p["first2"]
    assert orc.fnv_survey(y) == p["out_fnv"] and orc.fnv_survey(state) == p["state_fnv"]

    p = PINS["conv1d_c2_ir_256x256"]
    ir = orc.conv1d_ir(256, 256)
    assert f9(np.abs(ir).max()) == p["maxabs"] and f9(ir[0]) == p["first"] and orc.fnv_survey(ir) == p["fnv"]
    p = PINS["conv1d_c2_golden_256x512"]
    y = orc.conv1d(orc.noise(256 * 512), ir, 256, 512, 256)
    assert [f9(y[0]), f9(y[1])] == p["first2"] and f9(y[-1]) == p["last"]
    assert f9(np.abs(y).max()) == p["maxabs"] and orc.fnv_survey(y) == p["fnv"]

    p = PINS["glibc_rand_unseeded"]
    assert orc.Rand(1).next() == p["first_raw"]
    assert f9(orc.Rand(1).unit(1)[0]) == p["first_unit"] and f9(orc.Rand(1).bipolar(1)[0]) == p["first_bipolar"]

    p = PINS["datatransfer_sizes"]
    for r in ("0.01", "0.20", "0.50", "0.80", "0.99"):
        assert orc.datatransfer_size(float(r)) == p[r]

    p = PINS["dwg_128wg"]
    wg, _ = orc.dwg_init(128, 512)
    assert wg[0]["length"] == p["wg0_length"] and f9(wg[0]["gain"]) == p["wg0_gain"]
    assert wg["length"].min() == p["shortest_length"] and int((wg["length"] <= 512).sum()) == p["lengths_le_512"]

    p = PINS["modal_32x512"]
    y = orc.modal(orc.modal_params(64), 64, 512)
    assert f9(y[0]) == p["first"] and f9(y[-1]) == p["last"] and orc.fnv_survey(y) == p["fnv"]


def test_reference_pins_conv_accel_c3(orc):
    p = PINS["conv_accel_c3_ir_1024x4096"]
    ir = orc.conv_accel_ir(4096, 1024)
    assert f9(np.abs(ir).max()) == p["maxabs"] and f9(ir[0]) == p["first"] and orc.fnv_survey(ir) == p["fnv"]
    p = PINS["conv_accel_c3_golden_1024x512"]
    y = orc.conv_accel(orc.noise(1024 * 512), ir, 4096, 512, 1024)
    assert [f9(y[0]), f9(y[1])] == p["first2"] and f9(np.abs(y).max()) == p["maxabs"]
    assert orc.fnv_survey(y) == p["fnv"]


# ---------------------------------------------------------------- CPU: oracle vs derived fixtures
def test_derived_fixtures_are_what_the_oracle_still_says(orc):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(HERE, "golden", "make_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    fd_out, fd_p = mk.fdtd_case()
    assert np.array_equal(fd_out, FIX["fdtd_16_out"]) and np.array_equal(fd_p, FIX["fdtd_16_pressure"])
    y, state = mk.iir_case()
    assert orc.fnv(y) == hex_of(FIX["iir_3rd_buffer_fnv"]) and np.array_equal(state, FIX["iir_state_after_3"])
    fwd, bwd = mk.dwg_case()
    assert orc.fnv(fwd) == hex_of(FIX["dwg_fwd_fnv"]) and orc.fnv(bwd) == hex_of(FIX["dwg_bwd_fnv"])
    a, b = mk.modal_case()
    assert np.array_equal(a, FIX["modal_bank_20000x64_f32"]) and np.array_equal(b, FIX["modal_bank_20000x64_f64"])
    conv = mk.conv_stream_case()
    assert np.array_equal(conv[:, ::7], FIX["conv_stream_T8_L4096_B512_x12_f64"])


# ---------------------------------------------------------------- GPU: HIP path vs derived fixtures
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_gpu_streaming_convolution_against_fixture():
    import gpuaudiobench_amd as gab
    T, B, L, N = 8, 512, 4096, 12
    want = FIX["conv_stream_T8_L4096_B512_x12_f64"]
    peaks = FIX["conv_stream_peak"]
    plan = gab.ConvPlan(T, B, L)
    plan.set_ir(_dev(gab.harness.conv_accel_ir(L, T)))
    stream_peak = float(peaks.max())
    for i in range(N):
        y = plan.process(_dev(gab.harness.noise(T * B, seed=100 + i)), mode=gab.CONV_STREAMING).cpu().numpy()
        # north-star tolerance 1e-5 relative.  The first buffer and buffers >= 8 (full history) are
        # gated on their own peak; buffers 1..7 after a reset are ~1e-6 in magnitude while round-off
        # follows the centre taps, so they are gated on the stream's peak.
        scale = float(peaks[i]) if (i == 0 or i >= 8) else stream_peak
        assert np.abs(y[::7] - want[i]).max() <= 1e-5 * scale, "buffer %d" % i
    plan.close()


@pytest.mark.gpu
def test_gpu_fdtd_iir_dwg_modal_against_fixtures():
    import torch
    import gpuaudiobench_amd as gab
    import oracle as orc                       # input generators only (seeded); expectations come from FIX
    # FDTD3D 16^3: bit-exact
    n, T, B = 16, 4, 24
    plan = gab.FdtdPlan(gab.fdtd_default_params(n))
    x = _dev(orc.Rand(1).bipolar(T * B))
    out = torch.zeros(T * B, device="cuda")
    plan.process(x, out, T, B, 0, 10)
    plan.process(x, out, T, B, 10, 14)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), FIX["fdtd_16_out"].view(np.uint32))
    assert np.array_equal(plan.pressure().cpu().numpy().ravel().view(np.uint32), FIX["fdtd_16_pressure"].view(np.uint32))
    plan.close()
    # IIR (sequential form): bit-exact, three buffers of carried state
    T, B = 128, 512
    c = orc.iir_coeffs(0.25)
    state = torch.zeros(2 * T, device="cuda")
    for k in range(3):
        y = gab.iir(_dev(orc.noise(T * B, seed=7 + k)), _dev(c), state, T, B, sequential=True)
    assert orc.fnv(y.cpu().numpy()) == hex_of(FIX["iir_3rd_buffer_fnv"])
    assert np.array_equal(state.cpu().numpy().view(np.uint32), FIX["iir_state_after_3"].view(np.uint32))
    # ... and the wave-scan form within 1e-5 of the same buffer's peak
    state2 = torch.zeros(2 * T, device="cuda")
    for k in range(3):
        y2 = gab.iir(_dev(orc.noise(T * B, seed=7 + k)), _dev(c), state2, T, B)
    assert np.abs(y2.cpu().numpy()[:64] - FIX["iir_3rd_buffer_head"]).max() <= 1e-5 * np.abs(FIX["iir_3rd_buffer_head"]).max()
    # DWG delay lines after three iterations: bit-exact
    n_wg, B, ML = 128, 512, 2000
    wg, xin = orc.dwg_init(n_wg, B)
    fwd = torch.zeros(n_wg * ML, device="cuda")
    bwd = torch.zeros(n_wg * ML, device="cuda")
    wg_d = _dev(wg.view(np.uint8))
    for _ in range(3):
        gab.dwg(wg_d, fwd, bwd, _dev(xin), B, ML)
    assert orc.fnv(fwd.cpu().numpy()) == hex_of(FIX["dwg_fwd_fnv"])
    assert orc.fnv(bwd.cpu().numpy()) == hex_of(FIX["dwg_bwd_fnv"])
    # modal bank: 1e-5 of the peak
    nm, B, Tr = 20000, 64, 32
    y = gab.modal_bank(_dev(orc.modal_params(nm)), nm, B, Tr).cpu().numpy()
    ref64 = FIX["modal_bank_20000x64_f64"]
    assert np.abs(y - ref64).max() <= 1e-5 * np.abs(ref64).max()
