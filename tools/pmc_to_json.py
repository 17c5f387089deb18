#!/usr/bin/env python3
"""Fold the "JSON {...}" lines tools/pmc_summary.py printed (one per counter pass)
into the per-workload table bench.py reads (profiles/rNN_pmc.json).

    python tools/pmc_to_json.py gpurun_out/round6/pmc_summary.txt tools/profile_round6.sh > profiles/r06_pmc.json

(second argument: the script that made the summary - it is named in every entry's "source".)
Keys are "chain/voices/groups/fragments-per-step"; per key the DOMINANT kernel - the one
with the most time in the kernel trace of the counter runs (round 6; rounds 2-5 guessed it
from the chain's name, which labelled every scripted workload k_leaf_fmpan) - with its HBM traffic per launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950
correction of MI355X_MICROARCH.md) and its wave-level instruction counts per
voice-fragment, plus the same for every other kernel of the step under "kernels".
"""
import json
import sys

LEAF = {"osc-pan": "k_leaf_oscpan", "osc-filter-pan": "k_leaf_oscfiltpan", "osc2-pan": "k_leaf_osc2pan"}


def code_shas():
    """kernel symbol -> hash of its disassembly in the liba2amd.so the counters were collected on"""
    import hashlib
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    try:
        import isa_mix
        return {n: hashlib.sha256("\n".join(f"{m} {ops}" for _, m, ops, _ in ins).encode()).hexdigest()[:16]
                for n, ins in isa_mix.extract()}
    except Exception:  # noqa: BLE001
        return {}


def main():
    shas = code_shas()
    script = sys.argv[2] if len(sys.argv) > 2 else "an unnamed script"
    merged = {}
    for line in open(sys.argv[1]):
        if not line.startswith("JSON "):
            continue
        for label, kernels in json.loads(line[5:]).items():
            for k, d in kernels.items():
                # (counters: one pass each; the trace time of a kernel: the longest of the passes' totals)
                t = max(merged.get(label, {}).get(k, {}).get("trace_ns_total", 0.0), d.get("trace_ns_total", 0.0))
                merged.setdefault(label, {}).setdefault(k, {}).update(d)
                if t:
                    merged[label][k]["trace_ns_total"] = t
    out = {}
    for label, kernels in merged.items():
        chain, voices, groups, B = label.split("/")
        voices, groups, B = int(voices), int(groups), int(B)
        # (the runtime's own kernels - __amd_rocclr_copyBuffer / fillBuffer: hundreds of tiny set-up copies - are not candidates)
        timed = {k: d.get("trace_ns_total", 0.0) for k, d in kernels.items() if not k.startswith("__amd_")}
        if any(timed.values()):
            leaf = max(timed, key=timed.get)
            how = "by time in the counter runs' kernel trace"
        else:       # (a summary from before round 6: no trace times in it)
            leaf = LEAF.get(chain, "k_leaf_oscpan" if chain.startswith("osc-pan") else "k_leaf_fmpan")
            how = "by the chain's name (no trace times in the summary)"
        e = {"kernel": leaf, "kernel_chosen": how,
             "source": f"rocprofv3 --pmc passes of workload {label} ({script}), median launch", "kernels": {}}
        for k, d in kernels.items():
            kd = dict(d)
            if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
                kd["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM"):
                if c in d:
                    kd[c.lower()[3:] + "_per_voice_fragment"] = d[c] / (voices * B)
            if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"]:
                kd["valu_active_over_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
            e["kernels"][k] = kd
        e["code_sha"] = next((h for n, h in shas.items() if leaf in n and "commit" not in n), None)
        lk = e["kernels"].get(leaf, {})
        e["hbm_bytes_per_launch"] = lk.get("hbm_bytes_per_launch")
        e["valu_insts_per_voice_fragment"] = lk.get("insts_valu_per_voice_fragment")
        e["salu_insts_per_voice_fragment"] = lk.get("insts_salu_per_voice_fragment")
        e["formula"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, MI355X_MICROARCH.md)"
        out[label] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
