#!/usr/bin/env python3
"""Cost of cutting the FDTD room into z-slabs, on ONE device.

    python tools/fdtd_slabs_bench.py [--grid 128] [--parts 1 2 4] [--samples 16]

Times one buffer of `samples` samples (3 steps each) through the single-grid plan and through
`parts` slabs with the local plane exchange, both as plain launches and replayed from a graph
captured with torch.cuda.graph (the step kernels run on torch's current stream, so they are
captured together with the plane copies).  On one device the slabs only add work — the halo
copies and the smaller launches — so this measures the overhead a multi-device run has to
amortise, not a speed-up.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--parts", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--samples", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import gpuaudiobench_amd as gab
    from gpuaudiobench_amd import fdtd_slabs as fs

    n, T, B = args.grid, 16, args.samples
    G = gab.fdtd_default_params(n)
    steps = B * G.steps_per_sample
    x = torch.from_numpy(gab.harness.noise(T * B, seed=1)).cuda()
    out = torch.zeros(T * B, device="cuda")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps / steps * 1e6

    plan = gab.FdtdPlan(G)
    print("grid %d^3, %d samples x %d steps" % (n, B, G.steps_per_sample))
    plan.process(x, out, T, B, 0, B)
    want = out.clone()                                             # the first buffer from rest
    field = plan.pressure().clone()
    print("single-grid plan (its own graph replay): %.2f us/step" % timed(lambda: plan.process(x, out, T, B, 0, B)))
    plan.close()

    for parts in args.parts:
        slabs = [fs.FdtdSlab(G, a, b) for a, b in fs.slab_ranges(n, parts)]
        fs.process_local(slabs, x, out, T, B)                      # allocates the strips
        torch.cuda.synchronize()
        for s in slabs:
            s.reset()
        fs.process_local(slabs, x, out, T, B)
        same = bool(torch.equal(out, want)) and bool(torch.equal(torch.cat([s.pressure() for s in slabs]), field))
        eager = timed(lambda: fs.process_local(slabs, x, out, T, B))
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                fs.process_local(slabs, x, out, T, B)
        # an even number of steps per buffer leaves the ping-pong where the capture saw it
        assert steps % 2 == 0, "pick an even samples x steps_per_sample for the graph replay"
        replay = timed(g.replay)
        print("%d slab(s): launches %.2f us/step, graph replay %.2f us/step, halo %d B/step, first buffer and field identical: %s"
              % (parts, eager, replay, 24 * n * n * (parts - 1), same), flush=True)
        del g
        for s in slabs:
            s.close()


if __name__ == "__main__":
    main()
