"""gab_datatransfer_round_trip at the benchmark's real sizes (10 MiB per iteration: inputs that cross the runtime's 4 MiB - 1 byte
engine-packet limit once or twice) against the device kernel, bit for bit, fresh random data every call; a mismatch says which
words (a word cut by a packet boundary shows the input's low bytes under the sentinel's high ones: profiles/r05_incident_torn_word.txt).
    python tools/link_stress.py [calls per split]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
TOTAL = 2621440
rng = np.random.default_rng(3)
plan = gab.LinkPlan(TOTAL)
bad = 0
for ratio in (0.01, 0.2, 0.5, 0.8, 0.99):
    n_in = int(np.float32(TOTAL) * np.float32(ratio))
    n_out = int(np.float32(TOTAL) * np.float32(1.0 - ratio))
    h_in, h_out = torch.empty(n_in).pin_memory(), torch.empty(n_out).pin_memory()
    worst = 0
    for c in range(N):
        x = rng.random(n_in, dtype=np.float32)
        h_in.copy_(torch.from_numpy(x))
        h_out.fill_(float("nan"))
        plan.round_trip(h_in, h_out)
        want = gab.datatransfer(torch.from_numpy(x).cuda(), n_out).cpu().numpy()
        d = np.flatnonzero(h_out.numpy().view(np.uint32) != want.view(np.uint32))
        if d.size:
            bad += 1
            worst += 1
            print("in %d out %d call %d: %d words differ, first at %d (byte %d): got %08x want %08x" % (
                n_in, n_out, c, d.size, d[0], 4 * d[0], h_out.numpy().view(np.uint32)[d[0]], want.view(np.uint32)[d[0]]), flush=True)
    print("in %7d out %7d words: %d calls, %d with a mismatch" % (n_in, n_out, N, worst), flush=True)
plan.close()
sys.exit(1 if bad else 0)
