#!/usr/bin/env python3
"""valu_mix.py - measure, on the GPU box, the rate at which gfx950 issues the instruction mix
of each BASELINE leaf kernel's hot loop (tools/isa_mix.py reads the mix off the shipped
liba2amd.so; tools/ubench/valu_mix.hip issues it, verbatim and with independent operands).

    python tools/valu_mix.py > profiles/r03_valu_mix.json

Per kernel and hot loop: the static histogram (by mnemonic and issue class), the code hash of the
kernel it was read from, and the measured rates at 1..4 wavefronts per SIMD.  bench.py takes
roofline_valu.peak from here (independent operands, at the kernel's own occupancy)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402

CASES = [("k_leaf_osc2pan", "taps"), ("k_leaf_oscfiltpan", "taps"), ("k_leaf_oscfiltpan", "recurrence"),
         ("k_leaf_oscpan", "taps")]


def main():
    rates, src = isa_mix.rates_table()
    funcs = isa_mix.extract()
    ub = os.path.join(ROOT, "tools", "ubench")
    os.makedirs(os.path.join(ub, "variants"), exist_ok=True)
    out = {"rates_from": os.path.relpath(src, ROOT) if src else None, "kernels": {}}
    for kern, rule in CASES:
        for name, ins in funcs:
            if kern in name and "commit" not in name:
                res, body = isa_mix.analyse(name, ins, rates, rule)
                if rule not in res["hot_loops"]:
                    break
                tag = f"{kern[7:]}_{rule}"
                inc = os.path.join(ub, "variants", f"mix_{tag}.inc")
                n = isa_mix.emit(body, inc, f"{kern}:{rule}")
                exe = os.path.join(ub, "variants", f"valu_mix_{tag}")
                subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-w", f"-DMIX_INC=\"variants/mix_{tag}.inc\"",
                                "-o", exe, "valu_mix.hip"], cwd=ub, check=True)
                r = subprocess.run(["timeout", "120", exe], capture_output=True, text=True)
                try:
                    meas = json.loads(r.stdout.strip().splitlines()[-1])
                except (ValueError, IndexError):
                    meas = {"error": (r.stdout + r.stderr)[-300:]}
                e = out["kernels"].setdefault(kern, {"code_sha": res["code_sha"], "hot_loops": {}})
                h = res["hot_loops"][rule]
                e["hot_loops"][rule] = {k: h[k] for k in ("start", "end", "instructions", "valu", "by_class", "fraction_2_cycle",
                                                          "other", "additive_mix_rate_T_lane_ops")}
                e["hot_loops"][rule]["by_mnemonic"] = {k: v["count"] for k, v in h["by_mnemonic"].items()}
                e["hot_loops"][rule]["valu_issued_by_microbenchmark"] = n
                e["hot_loops"][rule]["measured"] = meas.get("rates", meas)
                break
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
