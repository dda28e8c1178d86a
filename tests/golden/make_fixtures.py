#!/usr/bin/env python3
"""Writes tests/golden/derived_fixtures.npz: outputs of the pinned CPU oracle on small seeded
inputs, for the cases the reference itself cannot pin (SURVEY §8c "parity unpinned": streaming
conv1d_accel beyond the first buffer, the real FDTD3D field evolution, IIR / DWG state at
iteration k > 0, the real modal bank).  The reference cannot be built or run in this image, so
these are the restatement's answers, frozen: a later change to oracle/ or to a kernel that moves
any of them shows up as a diff against this file.  Inputs are regenerated from seeds, only
outputs (or strided samples + FNV-1a-64 of them) are stored.

    python tests/golden/make_fixtures.py          # rewrites the .npz next to this script
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402


def conv_stream_case():
    """8 channels x 4096 taps x 512-sample buffers, 12 buffers: float64 direct form."""
    T, B, L, N = 8, 512, 4096, 12
    ir = orc.conv_accel_ir(L, T)
    hist = np.zeros(T * L, np.float32)
    outs = []
    for i in range(N):
        x = orc.noise(T * B, seed=100 + i)
        outs.append(orc.conv_accel_stream(x, ir, hist, L, B, T, f64=True))
    return np.stack(outs)                      # [N][B*T] float64, sample-major


def fdtd_case():
    """16^3 grid, 4 tracks, 24 samples in two calls (state carried)."""
    n, T, B = 16, 4, 24
    P = orc.fdtd_params(n)
    grids = orc.fdtd_grids(P)
    x = orc.Rand(1).bipolar(T * B)
    out = np.zeros(T * B, np.float32)
    orc.fdtd(P, grids, x, out, T, B, 0, 10, fused=True)
    orc.fdtd(P, grids, x, out, T, B, 10, 14, fused=True)
    return out, grids[0]


def iir_case():
    """128 tracks x 512, three buffers with carried state."""
    T, B = 128, 512
    c = orc.iir_coeffs(0.25)
    state = np.zeros(2 * T, np.float32)
    ys = [orc.iir(orc.noise(T * B, seed=7 + k), c, state, T, B) for k in range(3)]
    return ys[-1], state


def dwg_case():
    n_wg, B, ML = 128, 512, 2000
    wg, x = orc.dwg_init(n_wg, B)
    fwd = np.zeros(n_wg * ML, np.float32)
    bwd = np.zeros(n_wg * ML, np.float32)
    for _ in range(3):
        orc.dwg(wg, fwd, bwd, x, B, ML)
    return fwd, bwd


def modal_case():
    n, B, T = 20000, 64, 32
    p = orc.modal_params(n)
    return orc.modal_bank(p, n, B, T), orc.modal_bank_f64acc(p, n, B, T)


def main():
    conv = conv_stream_case()
    fd_out, fd_p = fdtd_case()
    iir_y, iir_state = iir_case()
    fwd, bwd = dwg_case()
    mb32, mb64 = modal_case()
    np.savez_compressed(
        os.path.join(HERE, "derived_fixtures.npz"),
        conv_stream_T8_L4096_B512_x12_f64=conv[:, ::7].copy(),          # every 7th output sample
        conv_stream_peak=np.abs(conv).max(axis=1),
        fdtd_16_out=fd_out, fdtd_16_pressure=fd_p,
        iir_3rd_buffer_fnv=np.frombuffer(bytes.fromhex(orc.fnv(iir_y)), np.uint8),
        iir_3rd_buffer_head=iir_y[:64].copy(), iir_state_after_3=iir_state,
        dwg_fwd_nonzero_idx=np.flatnonzero(fwd).astype(np.int32)[:4096],
        dwg_fwd_fnv=np.frombuffer(bytes.fromhex(orc.fnv(fwd)), np.uint8),
        dwg_bwd_fnv=np.frombuffer(bytes.fromhex(orc.fnv(bwd)), np.uint8),
        modal_bank_20000x64_f32=mb32, modal_bank_20000x64_f64=mb64,
    )
    print("wrote", os.path.join(HERE, "derived_fixtures.npz"),
          os.path.getsize(os.path.join(HERE, "derived_fixtures.npz")), "bytes")


if __name__ == "__main__":
    main()
