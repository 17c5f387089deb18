"""Build recipes: liba2amd.so (hipcc, gfx950) and the test oracle (gcc)."""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


MARK = b"A2AMD_SRCHASH:"


def source_hash(sources, cmd=()):
    """sha256 over the CONTENTS of everything a library is built from (+ its recipe): the
    stamp compiled into the library (`a2amd_srchash`, `-DA2AMD_SRCHASH`), so that a stale
    binary is found by what it was made of, not by file times (which a checkout, a copy to
    the GPU box or a touch changes)."""
    h = hashlib.sha256()
    for s in sources:
        h.update(os.path.basename(s).encode() + b"\0")
        with open(s, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(cmd).encode())
    return h.hexdigest()[:32]


def stamped_hash(lib, mark=MARK):
    """the stamp a built library carries (None: no library, or one from before the stamps)"""
    if not os.path.exists(lib):
        return None
    with open(lib, "rb") as f:
        blob = f.read()
    i = blob.find(mark)
    if i < 0:
        return None
    return blob[i + len(mark):i + len(mark) + 32].decode("ascii", "replace")


ENGMARK = b"A2AMD_ENGHASH:"


def engine_hash(engine):
    """Hash of the engine headers liba2amd_walk.so is compiled against (it reads A2_voice / A2_state members by
    name, i.e. by THAT version's offsets): everything under the engine's include/ and src/ that is a header or the
    template of one, plus liba2amd_units.so's own source hash (the walk links it)."""
    files = []
    for d in ("include", "src", os.path.join("src", "units"), os.path.join("src", "drivers")):
        p = os.path.join(engine, d)
        if os.path.isdir(p):
            files += [os.path.join(p, f) for f in sorted(os.listdir(p)) if f.endswith((".h", ".h.cmake", ".h.in"))]
    return source_hash(files, (stamped_hash(os.path.join(HERE, "liba2amd_units.so")) or "",))


def _stale(target, want):
    return stamped_hash(target) != want


def _define(h):
    return '-DA2AMD_SRCHASH="%s"' % h


def build_lib(force=False):
    csrc = os.path.join(HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in ("a2amd_host.cpp", "a2amd_sched.cpp", "a2amd_render.cpp", "a2amd_dist.cpp",
                                            "a2amd_vm.cpp", "a2amd_kernels.hip", "a2amd_fast.hip", "a2amd_vm.hip", "a2amd_wavecap.hip", "a2amd_win.hip", "a2amd_vmwin.hip")]
    deps = srcs + [os.path.join(csrc, "a2amd_host.h"), os.path.join(csrc, "a2amd_device.h"), os.path.join(csrc, "a2amd_dsp.h"),
                   os.path.join(csrc, "a2amd_fm.h"), os.path.join(csrc, "a2amd_vmcore.h"), os.path.join(csrc, "a2amd_taps.h"), os.path.join(csrc, "a2amd_winctl.h"), os.path.join(csrc, "a2amd_vmdev.h"), os.path.join(ROOT, "include", "a2amd.h"),
                   os.path.join(ROOT, "include", "a2amd_vm.h")]
    out = os.path.join(HERE, "liba2amd.so")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]
    want = source_hash(deps, flags)
    if force or _stale(out, want):
        subprocess.run([HIPCC] + flags + [_define(want), "-o", out] + srcs, check=True)
    return out


def build_units(force=False):
    """The drop-in unit descriptors (plain C) on top of liba2amd.so."""
    src = os.path.join(HERE, "csrc", "a2amd_units.c")
    deps = [src, os.path.join(ROOT, "include", "a2amd.h"), os.path.join(ROOT, "include", "a2amd_plugin.h"),
            os.path.join(ROOT, "include", "a2amd_walk.h"), os.path.join(ROOT, "include", "a2amd_vm.h")]
    out = os.path.join(HERE, "liba2amd_units.so")
    flags = ["-O2", "-Wall", "-fPIC", "-shared"]
    want = source_hash(deps, flags)
    if force or _stale(out, want):
        subprocess.run(["gcc"] + flags + [_define(want), "-o", out, src,
                        "-L" + HERE, "-la2amd", "-ldl", "-lpthread", "-lm", "-Wl,-rpath,$ORIGIN"], check=True)
    return out


def build_walk(force=False, engine=None):
    """INTEGRATION.md option C: liba2amd_walk.so, the engine's voice walk with the short cut for
    sleeping voices.  Compiled against the ENGINE's internal headers (src/internals.h), so it
    is built where the engine's source tree is (A2_ENGINE_SRC, default /root/reference) and for
    that engine version; elsewhere the library built before travels as it is."""
    engine = engine or os.environ.get("A2_ENGINE_SRC", "/root/reference")
    out = os.path.join(HERE, "liba2amd_walk.so")
    src = os.path.join(HERE, "csrc", "a2amd_walk.c")
    # the stamp covers OUR sources only: the engine's headers are not on the GPU box, and a
    # library made for another engine version is the maintainer's to rebuild
    own = [src, os.path.join(ROOT, "include", "a2amd_walk.h"), os.path.join(ROOT, "include", "a2amd_vm.h"),
           os.path.join(ROOT, "include", "a2amd_plugin.h")]
    flags = ["-O2", "-Wall", "-fPIC", "-shared"]
    want = source_hash(own, flags)
    if not (os.path.exists(os.path.join(engine, "src", "internals.h")) and shutil.which("cmake")):
        if os.path.exists(out) and _stale(out, want):
            raise RuntimeError(f"{out} was not built from this tree's a2amd_walk.c (stamp {stamped_hash(out)}, "
                               f"sources {want}) and the engine's source tree ({engine}) is not here to rebuild it")
        return out if os.path.exists(out) else None
    inc = os.path.join(HERE, "_engine_include")
    # where the engine's tree is, the library must also have been compiled against THESE headers (another engine
    # version = another A2_voice layout: a silent mismatch otherwise); boxes without the tree keep the stamp of our
    # own sources only, above
    weng = engine_hash(engine)
    if force or _stale(out, want) or stamped_hash(out, ENGMARK) != weng:
        subprocess.run(["cmake", f"-DENGINE={engine}", f"-DOUT={inc}", "-P", os.path.join(HERE, "csrc", "engine_header.cmake")],
                       check=True, stdout=subprocess.DEVNULL)
        subprocess.run(["gcc"] + flags + [_define(want), '-DA2AMD_ENGHASH="%s"' % weng, "-I" + inc] +
                       ["-I" + os.path.join(engine, d) for d in ("include", "src", "src/units", "src/drivers")] +
                       ["-o", out, src, "-L" + HERE, "-la2amd_units", "-la2amd", "-ldl", "-lpthread", "-Wl,-rpath,$ORIGIN"],
                       check=True)
    return out


def build_oracle(force=False):
    """Test infrastructure: the CPU restatement and, when the reference tree
    is present, the compiled reference + its harness (oracle/_ref)."""
    odir = os.path.join(ROOT, "oracle")
    targets = ["restate"]
    if os.path.exists("/root/reference/src/core.c") and shutil.which("cmake"):
        targets += ["ref", "tools", "optionb"]    # (optionb links the drop-in built just before)
    subprocess.run(["make", "-s", "-C", odir] + (["-B"] if force else []) + targets, check=True)
    return os.path.join(odir, "liba2oracle.so")


def build_all(force=False):
    lib = build_lib(force)
    build_units(force)
    build_walk(force)
    return lib, build_oracle(force)
