"""A seeded random producer at the doorbell-fed engine (gab_conv_engine_submit): bursts of 1..9 buffers, flushed or not, pauses of
0..3 ms, slots reused as soon as their buffer came back (copy ENGINES fill and empty the rings: at 1024 channels the launch holds
every compute unit); every output against one gab_conv_process launch per buffer, bit for bit.
    python tools/engine_fuzz.py [channels] [buffers] [seeds...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seeds = [int(a) for a in sys.argv[3:]] or [11, 12, 13]
B, L, R = 512, 4096, 16
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
bad = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    a, b = gab.ConvPlan(T, B, L, scheme="split"), gab.ConvPlan(T, B, L, scheme="split")
    a.set_ir(ir); b.set_ir(ir)
    xs = [gab.harness.noise(T * B, seed=9000 + 977 * seed + i) for i in range(N)]
    want = [a.process(torch.from_numpy(x).cuda()).cpu() for x in xs]
    side = torch.cuda.Stream()
    in_ring, out_ring = b.engine_start(R, stream=side)
    cur = torch.cuda.current_stream()
    h_in, h_out = torch.empty(T * B).pin_memory(), torch.empty(T * B).pin_memory()
    taken = published = bursts = flushed = 0

    def take_until(k_done):
        global taken, bad
        while taken < k_done:
            h_out.copy_(out_ring[taken % R], non_blocking=True); cur.synchronize()
            if not torch.equal(h_out.view(torch.int32), want[taken].view(torch.int32)):
                bad += 1
                print("seed %d: buffer %d differs" % (seed, taken), flush=True)
            taken += 1

    t0 = time.time()
    while published < N:
        n = int(min(N - published, rng.integers(1, 10)))
        flush = bool(rng.integers(0, 2)) or published + n == N
        if published + n - taken > R:
            b.engine_submit(0, flush=True); b.engine_wait(published, timeout=8.0); take_until(published)
        for j in range(n):
            h_in.copy_(torch.from_numpy(xs[published + j]))
            in_ring[(published + j) % R].copy_(h_in, non_blocking=True); cur.synchronize()
        b.engine_submit(n, flush=flush)
        published += n; bursts += 1; flushed += flush
        if flush and rng.integers(0, 2):
            b.engine_wait(published, timeout=8.0); take_until(published)
        time.sleep(float(rng.integers(0, 4)) * 1e-3)
    b.engine_wait(N, timeout=8.0); take_until(N)
    b.engine_stop()
    print("seed %d: %d buffers in %d bursts (%d flushed), %.1f s: %s" % (seed, N, bursts, flushed, time.time() - t0, "all bit-identical" if not bad else "%d MISMATCHES" % bad), flush=True)
    a.close(); b.close()
sys.exit(1 if bad else 0)
