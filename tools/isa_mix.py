#!/usr/bin/env python3
"""isa_mix.py - what the shipped kernels' hot loops are made of, read off the shipped binary.

    python tools/isa_mix.py                          # the three BASELINE leaf kernels, JSON on stdout
    python tools/isa_mix.py --kernel k_leaf_osc2pan --loops          # every loop of a kernel
    python tools/isa_mix.py --kernel k_leaf_osc2pan --emit tools/ubench/variants/mix_osc2pan.inc
                                                     # ... and the microbenchmark body for tools/ubench/valu_mix.hip

Extracts the gfx950 code objects from audiality2_amd/liba2amd.so (llvm-objdump --offloading),
disassembles them, finds a kernel's loops (backward branches) and - for its hot loops, chosen by
a stated rule (pick_loops) - prints the VALU histogram by mnemonic and by ISSUE CLASS.  The classes come from measurement, not from a
manual: profiles/r02_valu_rates.txt (tools/ubench/valu_rates.hip) has the chip-wide issue rate of
every integer VALU instruction the kernels use; an instruction is "2-cycle" when its measured rate
is above 50 T lane-ops/s (v_add_u32, v_sub_u32, v_and/or/xor_b32, v_ashrrev_i32, v_lshrrev_b32,
v_mov_b32: ~2.5 cycles per wave64 instruction and SIMD) and "4-cycle" otherwise (multiplies, v_bfe,
v_lshlrev, v_max, v_add_co, v_cndmask, SDWA forms ...: ~4.3 cycles); unmeasured mnemonics count as
4-cycle.  'code_sha' is a hash of the kernel's disassembly: PMC-derived figures kept in profiles/
carry the hash of the binary they were measured on, and bench.py says so when it differs.
"""
import argparse
import collections
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SO = os.path.join(ROOT, "audiality2_amd", "liba2amd.so")

# mnemonic (suffix-stripped) -> name in profiles/r02_valu_rates.txt
RATE_NAME = {
    "v_add_u32": "add_u32", "v_sub_u32": "sub_u32", "v_subrev_u32": "sub_u32", "v_and_b32": "and_b32",
    "v_or_b32": "and_b32", "v_xor_b32": "xor_b32", "v_ashrrev_i32": "ashrrev", "v_lshrrev_b32": "lshrrev",
    "v_mov_b32": "mov", "v_lshlrev_b32": "lshlrev", "v_mul_lo_u32": "mul_lo_u32", "v_mul_hi_i32": "mul_hi_i32",
    "v_mul_hi_u32": "mul_hi_i32", "v_mul_i32_i24": "mul_i32_i24", "v_mul_u32_u24": "mul_u32_u24",
    "v_mul_hi_i32_i24": "mul_hi_i24", "v_mul_hi_u32_u24": "mul_hi_i24", "v_mad_i32_i24": "mad_i32_i24",
    "v_mad_u32_u24": "mad_i32_i24", "v_mad_i64_i32": "mad_i64_i32", "v_mad_u64_u32": "mad_i64_i32",
    "v_bfe_i32": "bfe_i32", "v_bfe_u32": "bfe_u32", "v_alignbit_b32": "alignbit", "v_lshl_add_u32": "lshl_add",
    "v_add3_u32": "add3", "v_cndmask_b32": "cndmask_sgpr", "v_max_i32": "max_i32", "v_min_i32": "max_i32",
    "v_max_u32": "max_i32", "v_min_u32": "max_i32", "v_add_co_u32": "add_co", "v_addc_co_u32": "add_co",
    "v_sub_co_u32": "add_co", "v_subb_co_u32": "add_co", "v_bfi_b32": "bfi", "v_perm_b32": "perm",
    "v_and_or_b32": "and_or", "v_lshl_or_b32": "and_or", "v_add_lshl_u32": "lshl_add", "v_or3_b32": "add3",
    "v_xad_u32": "add3", "v_dot2_i32_i16": "dot2_i32_i16", "v_pk_mad_i16": "pk_mad_i16",
}
FAST_T = 50.0       # T lane-ops/s above which an instruction is in the 2-cycle class
TAP_LOADS = ("buffer_load_dwordx3", "buffer_load_dwordx4", "global_load_dwordx3")   # one Hermite coefficient entry (a2amd_fast.hip: coef_at)


def rates_table():
    """name -> T lane-ops/s at 8 wavefronts per SIMD from the newest profiles/r*_valu_rates.txt"""
    best = None
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if fn.endswith("_valu_rates.txt"):
            best = os.path.join(ROOT, "profiles", fn)
    tab = {}
    if best:
        for ln in open(best):
            m = re.match(r"(\S+)\s+waves/blk 8:\s+[\d.]+ ms\s+([\d.]+) T lane-ops/s", ln)
            if m:
                tab[m.group(1)] = float(m.group(2))
    return tab, best


def extract(so=SO):
    """[(kernel symbol, [(offset, mnemonic, operands)])] of every gfx950 code object in the library"""
    tmp = tempfile.mkdtemp(prefix="isa_mix_")
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(so, dst)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], check=True, capture_output=True, cwd=tmp)
        funcs = []
        for fn in sorted(os.listdir(tmp)):
            if "gfx950" not in fn:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, fn)],
                                 check=True, capture_output=True, text=True).stdout
            cur = None
            for ln in txt.splitlines():
                m = re.match(r"^([0-9a-f]+) <(\S+)>:", ln)
                if m:
                    cur = (m.group(2), int(m.group(1), 16), [])
                    funcs.append(cur)
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", ln)
                if m and cur:
                    cur[2].append((int(m.group(3), 16) - cur[1], m.group(1), m.group(2), ln))
        return [(n, ins) for n, _, ins in funcs]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def base_mnemonic(m):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", m)


def is_valu(m):
    return m.startswith("v_") and not m.startswith(("v_readlane", "v_readfirstlane", "v_writelane"))


def find_loops(ins):
    """[(start index, end index)] of backward branches (end = the branch)"""
    byoff = {o: i for i, (o, _, _, _) in enumerate(ins)}
    loops = []
    for i, (o, m, ops, ln) in enumerate(ins):
        if m.startswith(("s_cbranch", "s_branch")):
            t = re.search(r"<\S+\+0x([0-9a-f]+)>", ln)
            if t:
                to = int(t.group(1), 16)
                if to <= o and to in byoff:
                    loops.append((byoff[to], i))
    return loops


def classify(ins, rates):
    hist = collections.Counter()
    classes = collections.Counter()
    detail = {}
    other = collections.Counter()
    for _, m, _, _ in ins:
        b = base_mnemonic(m)
        if is_valu(m):
            hist[b] += 1
            nm = RATE_NAME.get(b)
            if m.endswith("_sdwa"):
                nm = "add_sdwa"
            r = rates.get(nm) if nm else None
            cls = "2-cycle" if (r is not None and r > FAST_T) else "4-cycle" if r is not None else "4-cycle (unmeasured)"
            classes[cls] += 1
            detail[b] = {"count": hist[b], "class": cls, "measured_T_lane_ops": r}
        elif m.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            other["lane<->scalar"] += 1
        elif m.startswith("s_"):
            other["salu" if not m.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer", "s_cbranch", "s_branch", "s_barrier")) else
                  "smem" if m.startswith(("s_load", "s_buffer")) else "control"] += 1
        elif m.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
        elif m.startswith("ds_"):
            other["lds"] += 1
        else:
            other["other"] += 1
    return hist, classes, detail, other


def pick_loops(ins):
    """The loops worth looking at, by a rule that can be checked against the listing (--loops):
    'taps'       the SMALLEST loop that contains every wave-data tap load of the kernel
                 (buffer_load_dwordx3 ... idxen: one Hermite coefficient entry, a2amd_fast.hip) - one trip
                 renders one chunk of fragments for the voices of a wavefront: the oscillator /
                 pan / mix-down work.  Where no loop holds them all (k_leaf_oscfiltpan has one copy of
                 its all-settled loop per number of voices and bus mode): the loops with the MOST tap
                 loads that hold a whole pipeline step (a barrier; back edges to the same header count as one
                 loop, its longest), of those the one with the fewest vector
                 instructions, then the shortest - six
                 voices, sums meeting in LDS: what the BASELINE configs run;
    'recurrence' the smallest loop of at least 100 vector instructions without any vector memory
                 instruction that works on an LDS tile (k_leaf_oscfiltpan: the filter wavefront's 16
                 frames of filter12 per trip, the pure low pass shape the BASELINE voices have).
    Kernels without tap loads: the innermost loop with the most VALU instructions."""
    loops = find_loops(ins)
    nx3 = sum(m in TAP_LOADS for _, m, _, _ in ins)
    out = {}
    if nx3:
        full = [l for l in loops if sum(m in TAP_LOADS for _, m, _, _ in ins[l[0]:l[1] + 1]) == nx3]
        if full:
            out["taps"] = min(full, key=lambda l: l[1] - l[0])
        else:
            ntap = lambda l: sum(m in TAP_LOADS for _, m, _, _ in ins[l[0]:l[1] + 1])
            most = max(ntap(l) for l in loops)
            cand = [l for l in loops if ntap(l) == most]
            steps = [l for l in cand if any(m == "s_barrier" for _, m, _, _ in ins[l[0]:l[1] + 1])]    # (a whole pipeline step)
            # (several back edges to one header - 'continue' paths of short steps - are one loop: its last back edge)
            steps = [l for l in steps if not any(o is not l and abs(o[0] - l[0]) <= 8 and o[1] > l[1] for o in steps)]
            out["taps"] = min(steps or cand,
                              key=lambda l: (sum(is_valu(m) for _, m, _, _ in ins[l[0]:l[1] + 1]), l[1] - l[0]))
    novm = [l for l in loops if not any(m.startswith(("global_", "buffer_", "flat_")) for _, m, _, _ in ins[l[0]:l[1] + 1])]
    novm = [l for l in novm if any(m.startswith("ds_") for _, m, _, _ in ins[l[0]:l[1] + 1])]     # (rows of an LDS tile)
    novm = [l for l in novm if sum(is_valu(m) for _, m, _, _ in ins[l[0]:l[1] + 1]) >= 100]
    if novm:
        out["recurrence"] = min(novm, key=lambda l: l[1] - l[0])
    if not out:
        inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)] or loops
        if inner:
            out["innermost"] = max(inner, key=lambda l: sum(is_valu(m) for _, m, _, _ in ins[l[0]:l[1] + 1]))
    return out, loops


def describe(ins, lo, hi, rates):
    body = ins[lo:hi + 1]
    hist, classes, detail, other = classify(body, rates)
    nv = sum(hist.values())
    r4 = [d["measured_T_lane_ops"] for d in detail.values() if d["class"] == "4-cycle"]
    t = 0.0
    for b, d in detail.items():     # issue time, additive: n / rate (unmeasured: the 4-cycle class's mean)
        t += d["count"] / (d["measured_T_lane_ops"] or (sum(r4) / len(r4) if r4 else 36.5))
    return {"start": hex(body[0][0]), "end": hex(body[-1][0]), "instructions": len(body), "valu": nv,
            "by_class": dict(classes), "fraction_2_cycle": classes.get("2-cycle", 0) / nv if nv else None,
            "other": dict(other), "by_mnemonic": dict(sorted(detail.items(), key=lambda kv: -kv[1]["count"])),
            "additive_mix_rate_T_lane_ops": nv / t if t else None}, body


def analyse(name, ins, rates, which=None):
    picks, loops = pick_loops(ins)
    sha = hashlib.sha256("\n".join(f"{m} {ops}" for _, m, ops, _ in ins).encode()).hexdigest()[:16]
    res = {"kernel": name, "code_sha": sha, "instructions_in_kernel": len(ins), "loops_in_kernel": len(loops),
           "hot_loops": {}, "note": "additive_mix_rate = VALU count / sum(count / measured single-instruction rate): the "
                                    "rate the SIMDs would issue this mix at if issue times simply added up; the measured "
                                    "rate of the mix itself comes from tools/ubench/valu_mix.hip (profiles/r03_valu_mix.json)"}
    body = None
    for rule, (lo, hi) in picks.items():
        res["hot_loops"][rule], b = describe(ins, lo, hi, rates)
        if body is None or rule == which:
            body = b
    return res, body


def vregs(ops):
    """VGPR numbers named in an operand string"""
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", ops):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"(?<![\w\[])v(\d+)\b", ops):
        out.add(int(a))
    return out


def sregs_written(m, ops):
    """SGPRs a VALU instruction writes (carry / compare results), vcc excluded"""
    out = set()
    first = ops.split(",")[0].strip() if ops else ""
    cands = [first]
    if ("_co_" in m or m.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale"))) and "," in ops:
        cands.append(ops.split(",")[1].strip())     # (the carry / flag output)
    for c in cands:
        mm = re.match(r"s\[(\d+):(\d+)\]$", c)
        if mm:
            out.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        mm = re.match(r"s(\d+)$", c)
        if mm:
            out.add(int(mm.group(1)))
    return out


def emit(body, path, name):
    """The loop's VALU instructions as two inline-asm bodies for tools/ubench/valu_mix.hip:
    verbatim (registers and with them the dependencies as in the kernel; loads, LDS, scalar
    instructions and waits left out) and with the operands renamed so that no instruction
    depends on another (destinations rotate through a pool, sources are two fixed registers)."""
    kept, clob_v, clob_s = [], set(), set()
    for _, m, ops, _ in body:
        if not is_valu(m) or m.startswith("v_cmpx") or "exec" in ops.split(",")[0]:
            continue
        if not (base_mnemonic(m) in RATE_NAME or m.endswith("_sdwa")):
            continue        # (compares, conversions, the division helper of the cold path: not part of the hot mix)
        if sregs_written(m, ops) & set(range(0, 34)):
            continue        # (would overwrite the benchmark kernel's own arguments)
        kept.append((m, ops))
        clob_v |= vregs(ops)
        clob_s |= sregs_written(m, ops)
    pool = 48
    indep = []
    for i, (m, ops) in enumerate(kept):
        parts = [p.strip() for p in ops.split(",")]
        # destination: a register of its own, same width as in the original (compares and
        # carries keep their scalar destination)
        new = [parts[0]]
        mm = re.match(r"v\[(\d+):(\d+)\]$", parts[0])
        if mm:
            w = int(mm.group(2)) - int(mm.group(1)) + 1
            base = 2 + (i % (pool // 2)) * 2
            new = [f"v[{base}:{base + w - 1}]"]
        elif re.match(r"v\d+$", parts[0]):
            new = [f"v{2 + i % pool}"]
        for p in parts[1:]:
            q = re.sub(r"(?<![\w\[])v\d+\b", "v0", p)
            q = re.sub(r"v\[(\d+):(\d+)\]", lambda g: f"v[{60}:{60 + int(g.group(2)) - int(g.group(1))}]", q)
            new.append(q)
        indep.append((m, ", ".join(new)))
    with open(path, "w") as f:
        f.write(f"// generated by tools/isa_mix.py from the hot loop of {name}: {len(kept)} VALU instructions\n")
        f.write(f"#define MIX_NAME \"{name}\"\n#define MIX_NVALU {len(kept)}\n")
        for tag, seq in (("KERNEL", kept), ("INDEP", indep)):
            f.write(f"#define MIX_BODY_{tag} \\\n")
            for m, ops in seq:
                f.write(f"\t\"{m} {ops}\\n\\t\" \\\n")
            f.write("\t\"\"\n")
        allv = sorted(clob_v | set(range(0, 2 + pool + 2)) | set(range(60, 64)))
        f.write("#define MIX_CLOBBERS " + ", ".join([f"\"v{v}\"" for v in allv] + [f"\"s{s}\"" for s in sorted(clob_s)] + ["\"vcc\""]) + "\n")
    return len(kept)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=SO)
    ap.add_argument("--kernel", action="append", help="substring of the kernel symbol (default: the BASELINE leaf kernels)")
    ap.add_argument("--rule", default=None, help="which hot loop to --emit: taps (default) / recurrence / innermost")
    ap.add_argument("--loops", action="store_true", help="list every loop of the kernel (start, end, VALU, tap loads)")
    ap.add_argument("--emit", help="write the microbenchmark include for tools/ubench/valu_mix.hip (one --kernel)")
    args = ap.parse_args()
    rates, src = rates_table()
    want = args.kernel or ["k_leaf_osc2pan", "k_leaf_oscfiltpan", "k_leaf_oscpan"]
    funcs = extract(args.so)
    out = {"rates_from": os.path.relpath(src, ROOT) if src else None, "kernels": {}}
    for w in want:
        for name, ins in funcs:
            if w in name and "commit" not in name:
                res, body = analyse(name, ins, rates, args.rule)
                out["kernels"][w] = res
                if args.loops:
                    res["all_loops"] = [{"start": hex(ins[a][0]), "end": hex(ins[b][0]), "insts": b - a + 1,
                                         "valu": sum(is_valu(m) for _, m, _, _ in ins[a:b + 1]),
                                         "tap_loads": sum(m in TAP_LOADS for _, m, _, _ in ins[a:b + 1])}
                                        for a, b in sorted(find_loops(ins), key=lambda l: l[1] - l[0])]
                if args.emit:
                    res["emitted_valu"] = emit(body, args.emit, w)
                break
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
