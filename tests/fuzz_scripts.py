"""Random A2S scripts for differential testing (tests/test_fuzz_dropin.py): voice
programs over every replaced unit with random structures, parameters, ramps and
timing, spawned by a random main program.  Deterministic per seed."""
import random

WAVES = ["sine", "saw", "triangle", "pulse10", "pulse30", "pulse50", "asine", "hsine", "qsine", "noise"]
FMS = [("fm1", 1), ("fm2", 2), ("fm3", 3), ("fm4", 4), ("fm3p", 3), ("fm4p", 4), ("fm2r", 2), ("fm4r", 4)]


def r(rng, lo, hi, nd=3):
    return f"{rng.uniform(lo, hi):.{nd}f}".rstrip("0").rstrip(".") or "0"


def pos(x):
    """A2S lexes '-.5' after a name as a literal only in some places: always parenthesise negatives."""
    return f"({x})" if x.startswith("-") else x


def delay(rng):
    return r(rng, 0.3, 35.0, 2)


def osc_steps(rng, n, pre=""):
    out = []
    for _ in range(n):
        k = rng.random()
        if k < 0.3:
            out.append(f"\t{pre}p (P + {pos(r(rng, -1.5, 1.5))}); d {delay(rng)}")
        elif k < 0.55:
            out.append(f"\t{pre}a (V * {r(rng, 0, 1.2)}); d {delay(rng)}")
        elif k < 0.65:
            out.append(f"\t@{pre}p (P + {pos(r(rng, -1, 2))}); {pre}phase {r(rng, 0, 1)}; d {delay(rng)}")
        elif k < 0.8:
            out.append(f"\t{pre}w {rng.choice(WAVES)}; d {delay(rng)}")
        elif k < 0.9:
            out.append(f"\t+{pre}p (rand .2 - .1); d {delay(rng)}")
        else:
            out.append(f"\td {delay(rng)}")
    return out


def pan_steps(rng, n):
    return [f"\tpan {pos(r(rng, -1.6, 1.6))}; vol {r(rng, 0, 1.5)}; d {delay(rng)}" for _ in range(n)]


def voice_program(rng, name):
    kind = rng.randrange(12)
    L = [f"{name}(P V)", "{"]
    n = rng.randint(2, 7)
    if kind == 0:
        L += ["\tstruct { wtosc; panmix }", f"\tw {rng.choice(WAVES)}; @p P; @a V; pan {pos(r(rng, -1, 1))}; set"]
        L += osc_steps(rng, n) + pan_steps(rng, rng.randint(0, 2))
    elif kind == 1:
        L += ["\tstruct { wtosc; filter12; panmix }",
              f"\tw {rng.choice(WAVES[:9])}; @p P; @a V; cutoff (P + {r(rng, 0, 6)}); q {r(rng, 0.001, 30)}; "
              f"lp {r(rng, 0, 1)}; bp {r(rng, 0, 1)}; hp {r(rng, 0, 1)}; set"]
        for _ in range(n):
            if rng.random() < 0.5:
                L.append(f"\tcutoff (P + {pos(r(rng, -2, 9))}); d {delay(rng)}")
            else:
                L.append(f"\t@cutoff (P + {r(rng, 0, 5)}); q {r(rng, 0.3, 20)}; d {delay(rng)}")
        L += osc_steps(rng, 2)
    elif kind == 2:
        L += ["\tstruct { wtosc A; wtosc B; panmix }",
              f"\tA.w {rng.choice(WAVES)}; B.w {rng.choice(WAVES)}; A.p P; B.p (P + {pos(r(rng, -1, 1))}); "
              f"A.a V; B.a (V * {r(rng, 0, 1)}); set"]
        L += osc_steps(rng, n // 2 + 1, "A.") + osc_steps(rng, n // 2 + 1, "B.")
    elif kind == 3:
        fm, nops = rng.choice(FMS)
        ring = fm.endswith("r")
        init = [f"@p P; @a {'1' if ring else 'V'}; fb {r(rng, 0, 1.2)}"]
        for j in range(1, nops):
            init.append(f"@p{j} {r(rng, 0.25, 4)}; @a{j} {('V' if (ring and j == 1) else r(rng, 0, 2.5))}; fb{j} {r(rng, 0, 1)}")
        L += [f"\tstruct {{ {fm}; panmix }}", "\t" + "; ".join(init) + f"; pan {pos(r(rng, -1, 1))}; set"]
        for _ in range(n):
            j = rng.randrange(nops)
            sfx = "" if j == 0 else str(j)
            k = rng.random()
            if k < 0.35:
                L.append(f"\ta{sfx} {r(rng, 0, 2) if j else '(V * ' + r(rng, 0, 1) + ')'}; d {delay(rng)}")
            elif k < 0.6:
                L.append(f"\tfb{sfx} {r(rng, 0, 1.5)}; d {delay(rng)}")
            elif k < 0.85:
                L.append(f"\tp{sfx} {('(P + ' + pos(r(rng, -1, 1)) + ')') if j == 0 else r(rng, 0.25, 4)}; d {delay(rng)}")
            else:
                L.append(f"\tphase {r(rng, 0, 1)}; d {delay(rng)}")
    elif kind == 4:
        L += ["\tstruct { wtosc; waveshaper; panmix }", f"\tw {rng.choice(WAVES[:9])}; @p P; @a V; amount {r(rng, 0, 8)}; set"]
        L += [f"\tamount {r(rng, 0, 10)}; d {delay(rng)}" for _ in range(n)] + osc_steps(rng, 2)
    elif kind == 5:
        L += ["\tstruct { wtosc; dcblock; panmix }", f"\tw {rng.choice(['pulse10', 'pulse30', 'hsine', 'qsine'])}; @p P; @a V; set"]
        L += [f"\tcutoff {pos(r(rng, -8, 3))}; d {delay(rng)}" for _ in range(n)] + osc_steps(rng, 2)
    elif kind == 6:
        L += ["\tstruct { wtosc; limiter; panmix }",
              f"\tw {rng.choice(WAVES[:9])}; @p P; @a (V * {r(rng, 1, 20)}); threshold {r(rng, 0.02, 0.6)}; release {r(rng, 1, 120)}; set"]
        L += [f"\ta (V * {r(rng, 0, 25)}); d {delay(rng)}" for _ in range(n)]
    elif kind == 7:
        L += ["\tstruct { dc; panmix }", f"\tvalue (V * {pos(r(rng, -1, 1))}); d {delay(rng)}"]
        for _ in range(n):
            if rng.random() < 0.3:
                L.append(f"\tmode {rng.choice(['STEP', 'LINEAR'])}")
            L.append(f"\tvalue (V * {pos(r(rng, -1, 1))}); d {delay(rng)}")
        L.append("\tvalue 0; d 3")
    elif kind == 8:
        L += ["\tstruct { env E; wtosc; panmix; wire E.out a }", f"\tw {rng.choice(WAVES[:9])}; @p P; pan {pos(r(rng, -1, 1))}; set pan"]
        L += [f"\tE.target (V * {r(rng, 0, 1)}); d {delay(rng)}" for _ in range(n)]
        L.append("\tE.target 0; d 8")
    elif kind == 9:
        L += ["\tstruct { wtosc }", f"\tw {rng.choice(WAVES)}; @p P; @a V; d {delay(rng)}"] + osc_steps(rng, n)
    elif kind == 10:
        L += ["\tstruct { wtosc; panmix P1 1 2; filter12 2 2; panmix P2 2 > }",
              f"\tw {rng.choice(WAVES[:9])}; @p P; @a V; P1.pan {pos(r(rng, -1, 1))}; cutoff (P + {r(rng, 0, 5)}); q {r(rng, 0.5, 9)}; set"]
        L += [f"\tP2.pan {pos(r(rng, -1.8, 1.8))}; cutoff (P + {pos(r(rng, -1, 7))}); d {delay(rng)}" for _ in range(n)]
    else:
        L += ["\tstruct { wtosc; panmix 1 2; fbdelay 2 > }",
              f"\tw {rng.choice(WAVES[:9])}; @p P; @a V; fbdelay {r(rng, 0.1, 30)}; ldelay {r(rng, 0.1, 40)}; "
              f"rdelay {r(rng, 0.1, 40)}; fbgain {pos(r(rng, -0.6, 0.6))}; set"]
        L += [f"\tldelay {r(rng, 0.05, 60)}; a (V * {r(rng, 0, 1)}); d {delay(rng)}" for _ in range(n)]
    L += ["\ta 0; d 2" if kind in (0, 1, 4, 5, 9, 10, 11) else "\td 1", "}"]
    return "\n".join(L)


def bus_program(rng, name, children):
    L = [f"{name}(P V)", "{"]
    if rng.random() < 0.5:
        L += ["\tstruct { inline 0 2; fbdelay D 2 2; panmix 2 > }",
              f"\tD.fbdelay {r(rng, 1, 50)}; D.ldelay {r(rng, 1, 60)}; D.rdelay {r(rng, 1, 60)}; D.fbgain {r(rng, 0, 0.5)}; "
              f"pan {pos(r(rng, -1, 1))}; set"]
    else:
        L += ["\tstruct { inline 0 2; panmix 2 2; xinsert 2 > }", f"\tvol {r(rng, 0.3, 1.2)}; pan {pos(r(rng, -1, 1))}; set"]
    for _ in range(rng.randint(2, 5)):
        L.append(f"\t{rng.choice(children)} (P + {pos(r(rng, -1, 1))}) (V * {r(rng, 0.3, 1)}); d {delay(rng)}")
    L += [f"\tvol {r(rng, 0, 1)}; d {delay(rng)}", "\td 60", "}"]
    return "\n".join(L)


def held_program(rng, name):
    """A note that sustains until it is told to stop: `end` parks the VM, message 1
    forces the release section (the idiom of the reference's test programs)."""
    k = rng.randrange(3)
    if k == 0:
        st, setup = "wtosc; panmix", f"w {rng.choice(WAVES)}; @p P; a V; pan {pos(r(rng, -1, 1))}"
    elif k == 1:
        st, setup = "wtosc; filter12; panmix", f"w {rng.choice(WAVES[:9])}; @p P; a V; cutoff (P + {r(rng, 1, 5)}); q {r(rng, 1, 9)}"
    else:
        fm, nops = rng.choice(FMS[:4])
        st = f"{fm}; panmix"
        setup = "@p P; a V; fb " + r(rng, 0, 1) + "".join(f"; @p{j} {r(rng, .5, 3)}; a{j} {r(rng, 0, 2)}" for j in range(1, nops))
    return "\n".join([f"{name}(P V)", "{", f"\tstruct {{ {st} }}", f"\t{setup}; d {delay(rng)}",
                      f"\ta (V * {r(rng, .2, 1)}); d {delay(rng)}", "\tend",
                      f".rel\ta 0; d {r(rng, 1, 40, 1)}", "\t1() { force rel }", "}"])


def pad_programs(rng):
    """Seeds >= 1000: sustained, sleeping voices in (nested) groups next to everything else - what the
    replaced voice walk (INTEGRATION.md option C) skips, holds and has to let go of again: pads that
    sleep for good, pads that wake a few times a second, groups that end with all their voices asleep."""
    L = []
    shapes = [("wtosc; panmix", "w {w}; p P; a V; pan {pan}"),
              ("wtosc; filter12; panmix", "w {w}; p P; a V; pan {pan}; cutoff (P + 2); q 3"),
              ("wtosc A; wtosc B; panmix", "A.w {w}; B.w sine; A.p P; B.p (P + .01); A.a V; B.a V; pan {pan}")]
    for i, (st, setup) in enumerate(shapes):
        su = setup.format(w=rng.choice(WAVES[:9]), pan=pos(r(rng, -1, 1)))
        L.append(f"Pad{i}(P V)\n{{\n\tstruct {{ {st} }}\n\t{su}; set\n\tfor {{ d 100000 }}\n}}\n")
    pre = "A." if False else ""
    L.append("PadW(P V)\n{\n\tstruct { wtosc; panmix }\n\tw triangle; p P; a V; set\n"
             f"\tfor {{ d {r(rng, 90, 400, 1)}; p (P + rand 1); d {r(rng, 60, 300, 1)}; p P }}\n}}\n")
    # PG: a group of pads (N of them), optionally with a life time; PGG: a group of such groups
    bus = rng.choice(["inline 0 2; panmix 2 2; xinsert 2 >", "inline 0 2; fbdelay D 2 2; panmix 2 >"])
    L.append(f"PG(P V N Life)\n{{\n\tstruct {{ {bus} }}\n\t!Q P\n"
             f"\tN {{ Pad0 Q V; +Q .07; Pad1 Q V; +Q .05; Pad2 Q V; d .02 }}\n\tPadW (P + 1) V\n"
             "\tif Life > 0 { d Life } else { for { d 100000 } }\n}\n")
    L.append("PGG(P V N Life)\n{\n\tstruct { inline 0 2; panmix 2 2; xinsert 2 > }\n"
             "\tPG P V N 0\n\tPG (P + .3) V N (Life * .6)\n\tPad1 (P - 1) V\n"
             "\tif Life > 0 { d Life } else { for { d 100000 } }\n}\n")
    return L


def loop_programs(rng, envs=False):
    """Seeds >= 2000: voices whose programs keep LOOPING over the part of the instruction set the device VM
    runs (include/a2amd_vm.h: SURVEY 8 f4) - random delays (milliseconds and ticks), counted and
    conditional loops inside the endless one, sets, ramps of their own length, arithmetic on work
    registers - plus the ways out of it: a handler that is sent messages, a kill, a detach, a group that
    cuts its voices' windows, a parent that ends.  Some loops contain what the VM must refuse (rand, a
    wave switch): those voices simply stay with the engine."""
    shapes = [
        ("wtosc; panmix", "w {w}; p P; a V; pan {pan}", ["p", "a", "pan", "vol", "phase"]),
        ("wtosc; filter12; panmix", "w {w}; p P; a V; pan {pan}; cutoff (P + 2); q 3", ["p", "a", "cutoff", "q", "lp", "bp", "hp", "pan"]),
        ("wtosc A; wtosc B; panmix", "A.w {w}; B.w sine; A.p P; B.p (P + .01); A.a V; B.a V; pan {pan}", ["A.p", "B.p", "A.a", "B.a", "pan"]),
        ("wtosc A; wtosc B; filter12; panmix", "A.w {w}; B.w saw; A.p P; B.p (P - .01); A.a V; B.a V; cutoff (P + 3); q 5",
         ["A.p", "B.a", "cutoff", "q", "pan"]),
        ("fm2; panmix", "p P; a V; fb .2; p1 2; a1 .5", ["p", "a", "fb", "p1", "a1", "fb1", "pan"]),
        ("wtosc; waveshaper; panmix", "w {w}; p P; a V; amount 2", ["amount", "a", "p"]),
        ("wtosc; panmix 1 2; fbdelay 2 >", "w {w}; p P; a V; fbdelay 13; ldelay 17; rdelay 19", ["fbgain", "lgain", "rgain", "drygain", "a"]),
    ]

    if envs:
        # seeds >= 3000: env units in those voices - amplitude, pan / volume and cutoff envelopes, in front
        # of and behind what they drive, table modes at random: their segments travel with the voice to the device VM
        shapes = shapes[:3] + [
            ("env E; wtosc; panmix; wire E.out a", "w {w}; p P; pan {pan}; E.mode C.{m1}; E.down C.{m2}", ["E.target", "E.target", "p", "pan", "E.time"]),
            ("wtosc; env E; panmix; wire E.out a", "w {w}; p P; pan {pan}; E.mode C.{m1}", ["E.target", "E.target", "p", "E.time"]),
            ("env EP; env EV; wtosc; panmix; wire EP.out pan; wire EV.out vol",
             "w {w}; p P; a V; EP.mode C.{m1}; EV.mode C.{m2}; EV.down C.{m1}", ["EP.target", "EV.target", "p", "a"]),
            ("env EA; wtosc; filter12; env EF; panmix; wire EA.out a; wire EF.out cutoff",
             "w {w}; p P; q 4; cutoff (P + 2); EA.mode C.{m1}; EA.down C.{m2}; EF.mode C.{m2}", ["EA.target", "EF.target", "q", "pan"]),
            ("env EF; wtosc; filter12; panmix; wire EF.out cutoff", "w {w}; p P; a V; q 6; cutoff (P + 1); EF.mode C.{m1}; EF.down C.{m2}",
             ["EF.target", "EF.target", "q", "a"]),
        ]
    MODES = ["LINEAR", "SPLINE", "LINK"] + [f"EXP{k}" for k in range(1, 8)] + [f"IEXP{k}" for k in range(1, 8)]

    def value(reg):
        base = reg.split(".")[-1]
        if base == "target":
            return value({"EP": "pan", "EF": "cutoff"}.get(reg.split(".")[0], "a"))
        if base == "time":
            return rng.choice(["0", r(rng, 1, 45, 1)])
        if base in ("p", "cutoff"):
            return f"(P + {pos(r(rng, -1.5, 3))})"
        if base in ("a", "vol"):
            return f"(V * {r(rng, 0, 1.3)})"
        if base == "pan":
            return pos(r(rng, -1.3, 1.3))
        if base == "q":
            return r(rng, 0.002, 25)
        if base in ("p1",):
            return r(rng, 0.25, 4)
        if base == "phase":
            return r(rng, 0, 1)
        if base == "amount":
            return r(rng, 0, 8)
        return r(rng, 0, 1)

    def step(regs, depth=0):
        k = rng.random()
        reg = rng.choice(regs)
        dly = f"d {r(rng, 0.4, 40, 2)}" if rng.random() < 0.8 else f"td {r(rng, 0.05, 0.5, 3)}"
        if k < 0.30:
            return [f"{reg} {value(reg)}; {dly}"]
        if k < 0.42:
            return [f"@{reg} {value(reg)}; {dly}"]
        if k < 0.54:
            return [f"+{reg} {pos(r(rng, -0.05, 0.05))}; {dly}"]
        if k < 0.64:
            return [f"{reg} {value(reg)}; ramp {reg} {r(rng, 1, 60, 1)}; {dly}"]
        if k < 0.72:
            return [f"*{reg} {r(rng, 0.7, 1.1)}; +X 1; {dly}"]
        if k < 0.80 and depth < 2:
            inner = sum((step(regs, depth + 1) for _ in range(rng.randint(1, 2))), [])
            return [f"{rng.randint(2, 5)} {{ " + "; ".join(inner) + " }"]
        if k < 0.88 and depth < 2:
            inner = step(regs, depth + 1)
            return [f"if X > {rng.randint(1, 6)} {{ X 0; " + "; ".join(inner) + " }"]
        if k < 0.92:
            return [f"+tr {pos(r(rng, -0.05, 0.05))}; {reg} {value(reg)}; {dly}", "if tr > .3 { tr 0 }", "if tr < -.3 { tr 0 }"]
        if k < 0.95:
            return [f"+{reg} (rand .02); {dly}"]           # the engine's RNG: not for the device
        return [f"set; {dly}"]

    L, names = [], []
    for i in range(rng.randint(4, 7)):
        st, setup, regs = rng.choice(shapes)
        su = setup.format(w=rng.choice(WAVES[:9]), pan=pos(r(rng, -1, 1)), m1=rng.choice(MODES), m2=rng.choice(MODES))
        body = sum((step(regs) for _ in range(rng.randint(2, 6))), [])
        if rng.random() < 0.9:      # (else: an iteration may go by without a delay - the engine's A2_OVERLOAD ends such a voice,
            body.append(f"d {r(rng, 0.5, 12, 2)}")     # and the device VM's analysis must have refused it)
        tempo = f"\ttempo {rng.randint(90, 200)} {rng.choice([2, 4, 8])}\n" if rng.random() < 0.5 else ""
        handler = ""
        if rng.random() < 0.4:
            reg = rng.choice(regs)
            off = "; ".join(dict.fromkeys(f"{a} 0" for a in regs if a.split(".")[-1] == "a" or a in ("E.target", "EA.target", "EV.target"))) or "vol 0"
            handler = f".rel\t{off}; d {r(rng, 2, 30, 1)}\n\t1(NP) {{ P NP; {reg} {value(reg)} }}\n\t2() {{ force rel }}\n"
        name = f"L{i}"
        names.append((name, bool(handler)))
        L.append(f"{name}(P V)\n{{\n\tstruct {{ {st} }}\n{tempo}\t{su}; set\n\t!X 0\n\tfor {{\n\t\t" +
                 "\n\t\t".join(body) + f"\n\t}}\n{handler}}}\n")
    plain = [n for n, h in names]
    # a group whose own program wakes every few ms: the voices under it are processed window by window
    L.append("LG(P V)\n{\n\tstruct { inline 0 2; panmix 2 2; xinsert 2 > }\n\tvol .8; set\n" +
             "".join(f"\t{rng.choice(plain)} (P + {pos(r(rng, -1, 1))}) V\n" for _ in range(rng.randint(1, 3))) +
             f"\tfor {{ +pan .07; d {r(rng, 1.5, 9, 2)}; if pan > 1 {{ pan -1 }} }}\n}}\n")
    # ... and one that ends, with looping voices under it
    L.append("LM(P V Life)\n{\n\tstruct { inline 0 2; panmix 2 2; xinsert 2 > }\n" +
             "".join(f"\t{rng.choice(plain)} (P + {pos(r(rng, -1, 1))}) V\n" for _ in range(rng.randint(1, 3))) +
             "\td Life\n}\n")
    return L, names


def make_script(seed):
    rng = random.Random(seed)
    nv = rng.randint(4, 8)
    names = [f"V{i}" for i in range(nv)]
    parts = [f'def title\t"fuzz{seed}"', 'def a2sversion\t"1.9"', 'def C\t\tunits.env.constants', ""]
    parts += [voice_program(rng, n) + "\n" for n in names]
    held = [f"H{i}" for i in range(2)]
    parts += [held_program(rng, h) + "\n" for h in held]
    buses = [f"B{i}" for i in range(rng.randint(1, 2))]
    parts += [bus_program(rng, b, names) + "\n" for b in buses]
    main = ["export Main(V=.15)", "{"]
    if seed % 3 == 0:
        # Main is a group (a2_NewGroup's driver shape): the test attaches a sink, and for
        # every other such seed a source as well, to its xinsert (A2REF_SINK / A2REF_SOURCE)
        main += ["\tstruct { inline 0 2; panmix 2 2; xinsert 2 > }", f"\tvol {r(rng, 0.4, 1)}; set"]
    if seed >= 1000:
        parts += pad_programs(rng)
        for _ in range(rng.randint(1, 3)):
            life = rng.choice(["0", r(rng, 200, 1500, 0)])
            main.append(f"\t{rng.choice(['PG', 'PGG'])} {pos(r(rng, -1.5, 1))} (V * .2) {rng.randint(2, 9)} {life}")
    if seed >= 2000:
        lp, lnames = loop_programs(rng, envs=seed >= 3000)
        parts += lp
        for name, _h in lnames:
            main.append(f"\t{name} {pos(r(rng, -1.5, 1))} (V * .3)")
        main.append(f"\tLG {pos(r(rng, -1, 1))} (V * .3)")
        main.append(f"\tLM {pos(r(rng, -1, 1))} (V * .3) {r(rng, 150, 900, 0)}")
        main.append("\td 40")
    main += ["\t!P 0", "\tfor {"]
    for _ in range(rng.randint(6, 14)):
        if rng.random() < 0.25:
            # an attached voice with an id: started, held, then released by message, killed or detached
            vid = rng.randint(1, 6)
            how = rng.choice([f"{vid}<1", f"kill {vid}", f"detach {vid}"])
            main.append(f"\t\t{vid}:{rng.choice(held)} (P + {pos(r(rng, -1, 1))}) V; d {delay(rng)}; {how}; d {delay(rng)}")
            continue
        if seed >= 2000 and rng.random() < 0.35:
            # a looping voice with an id: sent messages while its loop runs, then released, killed or detached
            name, has_handler = rng.choice(lnames)
            vid = rng.randint(7, 12)
            how = rng.choice([f"kill {vid}", f"detach {vid}"] + ([f"{vid}<2"] if has_handler else []))
            msg = f"{vid}<1 (P + {pos(r(rng, -1, 1))}); d {delay(rng)}; " if has_handler else ""
            main.append(f"\t\t{vid}:{name} (P + {pos(r(rng, -1, 1))}) (V * .3); d {delay(rng)}; {msg}{how}; d {delay(rng)}")
            continue
        prog = rng.choice(names + buses)
        main.append(f"\t\t{prog} (P + {pos(r(rng, -1.5, 1))}) V; d {delay(rng)}")
    main += ["\t\t+P (rand 1 - .5)", "\t\tif P > 1 { P 0 }", "\t\tif P < -2 { P 0 }", "\t}", "}"]
    return "\n".join(parts + main) + "\n"


if __name__ == "__main__":
    import sys
    print(make_script(int(sys.argv[1]) if len(sys.argv) > 1 else 0))
