"""gab_conv_round_trip under DAW pacing (one buffer per 512/48000 s slot, the device idle between): where do the ~10 us
over the back-to-back p50 come from, and what does gab_keep_warm buy?  The same 100 paced calls as they are; with a
gab_keep_warm launch of 1 / 8 / 64 / 256 waves kicked after every call (diagnostic builds: GAB_KEEP_WARM_NAPS = how many
4 us sleeps between looks); through the product's switch gab_conv_round_trip_keep_warm; and with the host thread making a
runtime call (hipStreamQuery) between slots instead of spinning on the clock alone.
    python tools/paced_keepwarm.py [channels] [slots]        (GAB_LIB_PATH=...ablate.so for the naps)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gpuaudiobench_amd as gab
T, B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 4096
SLOTS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).to(dev)
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B, dtype=torch.float32).pin_memory()
rplan = gab.ConvPlan(T, B, L, scheme="classic")
rplan.set_ir(ir)
args = rplan.prepare_round_trip(h_in, h_out)

torch.cuda.synchronize()


def back_to_back(n=520, skip=20):
    ts = []
    for i in range(n):
        t1 = time.perf_counter()
        rplan.launch_round_trip(args)
        if i >= skip:
            ts.append((time.perf_counter() - t1) * 1e6)
    return np.array(ts)


def paced(between=None, after=None):
    daw = gab.harness.DawSim(buffer_seconds=float(B) / 48000, mode="spin")
    ts = []
    for i in range(SLOTS + 5):
        if between is not None:
            t_end = time.perf_counter() + 0.0095
            while time.perf_counter() < t_end:
                between()
        daw.wait()
        t1 = time.perf_counter()
        rplan.launch_round_trip(args)
        if i >= 5:
            ts.append((time.perf_counter() - t1) * 1e6)
        if after is not None:
            after()
    waits, missed = daw.stats()
    daw.close()
    return np.array(ts), missed


def line(name, ts, missed=None):
    print("%-64s p50 %6.1f us  p95 %6.1f  min %6.1f  max %6.1f%s" % (name, np.percentile(ts, 50), np.percentile(ts, 95), ts.min(), ts.max(),
          "" if missed is None else "  missed slots %d" % missed), flush=True)


import os
line("back to back", back_to_back())
cur = torch.cuda.current_stream()
tiny_h, tiny_d, side = torch.zeros(16).pin_memory(), torch.zeros(16, device=dev), torch.cuda.Stream()


def tiny_copies():
    # tiny pinned -> device copies on a side stream until the slot comes (what a caller who knows its period could do from a
    # timer thread; here to see whether the rest of the gap is the copy engines' own idle state)
    with torch.cuda.stream(side):
        tiny_d.copy_(tiny_h, non_blocking=True)
    side.synchronize()


def with_keep_warm(wgs, naps, label=None):
    os.environ["GAB_KEEP_WARM_NAPS"] = str(naps)                # (read by diagnostic builds only)
    warm = gab.KeepWarm(workgroups=wgs, idle_seconds=0.25)
    warm.kick()
    ts, missed = paced(after=warm.kick)
    alive = warm.running()
    where = gab.ops.placement_summary(warm.placement())          # where the waves landed: a run that buys nothing classifies itself
    warm.close()
    line((label or "paced, keep-warm: %3d waves, a look every %d naps" % (wgs, naps)) + (" [launch alive: %s]" % alive), ts, missed)
    print("%-64s %s" % ("", where), flush=True)


def with_switch(between=None, label="paced, gab_conv_round_trip_keep_warm(plan, 1)"):
    rplan.round_trip_keep_warm(True)                            # the product's switch: 8 waves, kicked by the call itself
    rplan.launch_round_trip(args)
    line(label, *paced(between))
    print("%-64s %s" % ("", gab.ops.placement_summary(rplan.round_trip_keep_warm_placement())), flush=True)
    rplan.round_trip_keep_warm(False)
    time.sleep(0.4)


for rep in range(2):
    line("paced, device idle between slots", *paced())
    for wgs, naps in ((1, 1), (8, 1), (8, 16), (16, 16), (64, 1), (256, 16)):
        with_keep_warm(wgs, naps)
    with_switch()
    with_keep_warm(8, 16, "paced, keep-warm:   8 waves again, after the switch has been used")
    with_switch(tiny_copies, "paced, keep-warm switch + tiny engine copies between slots")
    line("paced, tiny engine copies between slots alone", *paced(tiny_copies))
    line("paced, the host queries a stream between slots", *paced(lambda: cur.query()))
    with_keep_warm(8, 16, "paced, keep-warm:   8 waves once more")
line("back to back", back_to_back())
rplan.close()
