"""gab_conv_round_trip back to back: a fresh process, the default stream against a stream of its own, each check rule."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gpuaudiobench_amd as gab
T, B, L = 1024, 512, 4096
ir = torch.from_numpy(gab.harness.conv_accel_ir(L, T)).cuda()
h_in = torch.from_numpy(gab.harness.noise(T * B, seed=7)).pin_memory()
h_out = torch.empty(T * B).pin_memory()
def p50(plan, stream, n=420):
    args = plan.prepare_round_trip(h_in, h_out, stream=stream)
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); plan.launch_round_trip(args); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts[20:])
    return "p50 %.1f p95 %.1f" % (np.percentile(ts, 50), np.percentile(ts, 95))
own = torch.cuda.Stream()
for label, stream in (("default stream", None), ("own stream", own), ("default stream", None)):
    for mode in (1, 0, 2):
        plan = gab.ConvPlan(T, B, L, scheme="classic"); plan.set_ir(ir); plan.round_trip_set_check(mode)
        print("%-15s set_check(%d): %s" % (label, mode, p50(plan, stream)), flush=True)
        plan.close()
# what bench.py does before its round-trip leg: other plans, an engine on a side stream
e = gab.ConvPlan(T, B, L, scheme="split"); e.set_ir(ir)
side = torch.cuda.Stream()
e.engine_start(8, stream=side); e.engine_stop(); e.close()
plan = gab.ConvPlan(T, B, L, scheme="classic"); plan.set_ir(ir)
print("after an engine has run on a side stream, default stream, set_check(1): %s" % p50(plan, None), flush=True)
plan.close()
# the same scenario, rule by rule, and with streams alone (the runtime maps streams onto a few hardware queues)
for mode in (1, 0, 2, 1):
    plan = gab.ConvPlan(T, B, L, scheme="classic"); plan.set_ir(ir); plan.round_trip_set_check(mode)
    print("after the engine, default stream, set_check(%d): %s" % (mode, p50(plan, None)), flush=True)
    print("after the engine, own stream,     set_check(%d): %s" % (mode, p50(plan, own)), flush=True)
    plan.close()
extra = [torch.cuda.Stream() for _ in range(6)]
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(16, device="cuda")
torch.cuda.synchronize()
plan = gab.ConvPlan(T, B, L, scheme="classic"); plan.set_ir(ir)
print("six more streams in the process, default stream, set_check(1): %s" % p50(plan, None), flush=True)
plan.close()
