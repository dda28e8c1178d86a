#!/usr/bin/env python3
"""Channel ranges of one streaming convolution plan on several HIP streams, queued by one host
thread per stream (gab_conv_stream_ranges) or by a single thread: us per buffer.

    python tools/multiqueue_conv.py [--buffers 3000]

T = 1024 is BASELINE C3; T = 2048 / 4096 with 1024-channel ranges are "two / four full-size chains
on one device" (what two processes sharing the GPU do), reported per 1024 channels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buffers", type=int, default=3000)
    ap.add_argument("--tracks", type=int, nargs="*", default=[1024, 2048, 4096])
    args = ap.parse_args()
    import numpy as np
    import torch
    import gpuaudiobench_amd as gab
    B, L = 512, 4096
    dev = torch.device("cuda", 0)
    rows = []
    for T in args.tracks:
        ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).to(dev)
        plan = gab.ConvPlan(T, B, L, scheme="split")
        plan.set_ir(ir)
        del ir
        inputs = [torch.from_numpy(gab.harness.noise(T * B, seed=42 + i)).to(dev) for i in range(16)]
        out = torch.empty(T * B, dtype=torch.float32, device=dev)
        streams = [torch.cuda.Stream() for _ in range(8)]
        for R in (1, 2, 3, 4, 8):
            if T // R < 64:
                continue
            if R == 3:                       # 16-channel granules (whole XCD groups of workgroups)
                g = T // 16
                cuts = [0, 16 * ((g + 2) // 3), 16 * ((g + 2) // 3 + (g + 1) // 3), T]
                ranges = [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)]
            else:
                n = T // R
                ranges = [(r * n, n) for r in range(R)]
            for threaded in ((True,) if R == 1 else (True, False)):
                os.environ["GAB_CONV_RANGE_THREADS"] = "1" if threaded else "0"
                # the library reads the variable once: it is static there, so single-thread runs use
                # the python-side loop instead
                plan.reset()
                torch.cuda.synchronize()

                def run(k):
                    if threaded:
                        plan.stream_ranges(inputs, out, ranges, streams[:R], k)
                    else:
                        prepared = [[plan.prepare_range(x, out, a, c, streams[r]) for r, (a, c) in enumerate(ranges)]
                                    for x in inputs]
                        for i in range(k):
                            for a in prepared[i % 16]:
                                plan.launch_range(a)
                            plan.advance()
                run(500)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(args.buffers)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) * 1e6 / args.buffers
                row = {"tracks": T, "ranges": R, "host_threads": R if threaded else 1,
                       "us_per_buffer": round(us, 3), "us_per_1024ch": round(us * 1024 / T, 3),
                       "alg_TBps": round(4 * T * (2 * B + 2 * L) / us / 1e6, 3)}
                rows.append(row)
                print(json.dumps(row), flush=True)
        plan.close()
        del inputs, out
    return rows


if __name__ == "__main__":
    main()
