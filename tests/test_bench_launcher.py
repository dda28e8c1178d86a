"""`python bench.py --gpus N` as a plain command (no torch.distributed.run around it): the parent
starts the ranks itself, relays rank 0's JSON line and returns the ranks' exit code.  On CPU the
ranks run bench.py's dry-run path (GAB_BENCH_DRYRUN=1: rendezvous, bank broadcast and the timing
collectives over gloo, no device work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, GAB_BENCH_DRYRUN="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_2_launches_its_own_ranks():
    r = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                   # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["ir_broadcast_ms"] is not None
    assert d["config"]["ir_slices_match_global_bank"] is True
    assert d["value"] is None and "DRY RUN" in d["data"]          # never mistaken for a measurement
    # rank 0's line carries every rank's own figures: the source receives nothing, rank 1 the whole bank (16 x 64 taps)
    pr = d["config"]["per_rank"]
    assert pr["ir_bytes_received"] == [0, 4 * 16 * 64] and len(pr["ir_broadcast_ms"]) == 2 and all(v >= 0 for v in pr["ir_broadcast_ms"])


def test_bench_single_rank_needs_no_launcher():
    r = _run({}, "--gpus", "1", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1


def test_bench_launcher_propagates_a_failing_rank():
    # an impossible shard count for the dry run's bank makes a rank exit non-zero
    r = _run({"GAB_BENCH_DRYRUN_TRACKS": "0"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0


def test_bench_gpus_8_dry_run_at_c5_shapes():
    """The driver's 8-GPU command shape on the CPU path: `bench.py --gpus 8` starts eight ranks, which rendezvous over
    gloo and distribute BASELINE configs[4]'s bank (8192 channels x 4096 taps = 128 MiB) as 1024-channel slices at
    global indices; every rank checks its slice against the formula at its global rows."""
    r = _run({"GAB_BENCH_DRYRUN_TRACKS": "1024", "GAB_BENCH_DRYRUN_TAPS": "4096"},
             "--gpus", "8", "--steps", "2", "--warmup", "1", "--ir-distribution", "slices")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 8 and d["config"]["channels_total"] == 8192 and d["config"]["channels_per_rank"] == 1024
    assert d["config"]["taps"] == 4096 and d["config"]["ir_distribution"] == "slices"
    assert d["config"]["ir_slices_match_global_bank"] is True
    pr = d["config"]["per_rank"]                       # 16 MiB of rows per rank instead of the 128 MiB bank
    assert pr["ir_bytes_received"] == [0] + [4 * 1024 * 4096] * 7 and len(pr["ir_broadcast_ms"]) == 8
