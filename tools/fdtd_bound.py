"""The LDS-resident FDTD3D kernel against its OWN bound (it moves no field through HBM, so an HBM fraction means nothing).

A step cannot be shorter than the longer of
  (a) VALU issue:   counted VALU instructions per SIMD and step (rocprofv3 --pmc SQ_INSTS_VALU, tools/profile_fdtd_resident.sh)
                    x 2.63 cycles per wave64 fp32 instruction — what a SIMD with four or more waves issues, measured
                    (tools/ubench/valu_rate: one wave alone gets one per 5.25) — / the clock the kernel ran at;
  (b) the hand-off: a block's boundary pressures of step t feed its neighbours' face rows of step t+1, so one request-to-data
                    round trip through memory sits on every step's chain — taken from the diagnostic build's clock marks (the
                    mark "quads there" of a wave that asked at the step's start, all 256 workgroups asking at once).
Also measured: the step with the hand-off ablated (diagnostic build: compute, LDS and barriers only).

    python tools/fdtd_bound.py <pmc_means.json> <out.json> <out.md>      (on a GPU box; needs libgab_hip_ablate.so)
"""
kClkPerInstr = 2.63        # a SIMD holding >= 4 waves: profiles/r04_valu_rate.txt
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "gpuaudiobench_amd", "libgab_hip_ablate.so")


def loop(n, samples, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fdtd_loop.py"), str(n), str(samples), "8"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    if r.returncode != 0:
        raise SystemExit(r.stdout + r.stderr)
    us = [float(m.group(1)) for m in re.finditer(r"-> ([\d.]+) us/step", r.stdout)]
    waves = [[float(v) for v in ln.split(":")[1].split()] for ln in r.stdout.splitlines() if ln.strip().startswith("wave")]
    return min(us[1:]), waves, r.stdout


def main():
    pmc = json.load(open(sys.argv[1])) if os.path.exists(sys.argv[1]) else {}
    out = {}
    lines = []
    for case, n, samples in (("fdtd_128", 128, 334), ("fdtd_52", 52, 334)):
        prod, _, _ = loop(n, samples, {})
        diag, _, _ = loop(n, samples, {"GAB_LIB_PATH": ABL, "GAB_FDTD_RES_ABLATE": "0"})      # the diagnostic library as it is
        noh, _, _ = loop(n, samples, {"GAB_LIB_PATH": ABL, "GAB_FDTD_RES_ABLATE": "1"})
        stamped, waves, text = loop(n, samples, {"GAB_LIB_PATH": ABL, "GAB_FDTD_RES_ABLATE": "16"})
        # marks: clocks from a step's start to: low faces | barrier A | interior pressures | quads there | face rows + stores | barrier B
        waves = [w for w in waves if any(w)]
        step_clk = max(w[5] for w in waves)
        ghz = step_clk / (stamped * 1e3)                     # clocks per step / ns per step
        quads_us = max(w[3] for w in waves) / (ghz * 1e3)
        interior_us = max(w[2] for w in waves) / (ghz * 1e3)
        # the PMC passes ran tools/fdtd_loop.py 128 334 8: 1002 steps per counted launch, 256 CUs x 4 SIMDs
        per_launch = pmc.get("counters_mean_per_launch", {}).get("SQ_INSTS_VALU")
        valu = per_launch / 1002.0 / 1024.0 if (per_launch and case == "fdtd_128") else None
        issue_us = valu * kClkPerInstr / (ghz * 1e3) if valu else None
        floor = max(quads_us, issue_us or 0.0)
        out[case] = dict(grid=n, us_per_step=prod, us_per_step_diagnostic_build=diag, us_per_step_handoff_ablated=noh,
                         us_per_step_with_marks=stamped, clock_GHz=ghz,
                         handoff_request_to_data_us=quads_us, interior_pressures_done_us=interior_us,
                         valu_instructions_per_simd_per_step=valu, valu_issue_us=issue_us, floor_us_per_step=floor,
                         frac_of_floor=floor / prod)
        lines.append("| %d^3 | %.2f | %.2f / %.2f / %.2f | %s | %.2f | %.2f | %.2f | **%.2f** |" % (
            n, prod, diag, noh, stamped, "%.2f (%.0f instr x %.2f clk at %.2f GHz)" % (issue_us, valu, kClkPerInstr, ghz) if issue_us else "not counted",
            interior_us, quads_us, floor, floor / prod))
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    with open(sys.argv[3], "w") as f:
        f.write("# r04 — the LDS-resident FDTD3D kernel against its own bound\n\n"
                "The kernel keeps the room in LDS and registers for a whole buffer; per step it moves only its blocks' boundary pressures\n"
                "through memory.  An HBM roofline does not bound it (algorithmic bytes / time is 2.9 x 8 TB/s at 128^3), so the fraction\n"
                "reported from round 4 on is of its own per-step floor = max(VALU issue, the neighbour hand-off's request-to-data round trip).\n"
                "Measured by `tools/fdtd_bound.py` (product library; the diagnostic library for the ablated step and the clock marks of one\n"
                "workgroup; `SQ_INSTS_VALU` from `%s`).  us per step:\n\n" % os.path.basename(sys.argv[1]))
        f.write("| room | step (product library) | diagnostic library: as it is / hand-off ablated / with the clock marks | (a) VALU issue | interior pressures done at | (b) asked quads there at | floor = max(a, b) | floor / step |\n|---|---|---|---|---|---|---|---|\n")
        f.write("\n".join(lines) + "\n\n")
        f.write("Reading: the floor is the hand-off — the time from a step's start, when a wave asks for its neighbours' boundary pressures, to the\n"
                "moment they are in its registers, all 256 workgroups asking at once (taken in the library that carries the marks, whose step is\n"
                "longer than the product's: the mark is an upper estimate of the product's round trip).  The diagnostic library's own step with the\n"
                "hand-off ablated against its step as it is says what the hand-off still costs on top of compute, LDS traffic and the two barriers;\n"
                "VALU issue alone is about half a step.\n")
    print(open(sys.argv[3]).read())


if __name__ == "__main__":
    main()
