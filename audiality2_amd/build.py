"""Build recipes: liba2amd.so (hipcc, gfx950) and the test oracle (gcc)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_lib(force=False):
    csrc = os.path.join(HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in ("a2amd_host.cpp", "a2amd_sched.cpp", "a2amd_render.cpp", "a2amd_dist.cpp",
                                            "a2amd_vm.cpp", "a2amd_kernels.hip", "a2amd_fast.hip", "a2amd_vm.hip", "a2amd_wavecap.hip")]
    deps = srcs + [os.path.join(csrc, "a2amd_host.h"), os.path.join(csrc, "a2amd_device.h"), os.path.join(csrc, "a2amd_dsp.h"),
                   os.path.join(csrc, "a2amd_fm.h"), os.path.join(csrc, "a2amd_vmcore.h"), os.path.join(ROOT, "include", "a2amd.h"),
                   os.path.join(ROOT, "include", "a2amd_vm.h")]
    out = os.path.join(HERE, "liba2amd.so")
    if force or _newer(out, deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wno-unused-value", "-o", out] + srcs
        subprocess.run(cmd, check=True)
    return out


def build_units(force=False):
    """The drop-in unit descriptors (plain C) on top of liba2amd.so."""
    src = os.path.join(HERE, "csrc", "a2amd_units.c")
    deps = [src, os.path.join(ROOT, "include", "a2amd.h"), os.path.join(ROOT, "include", "a2amd_plugin.h"),
            os.path.join(ROOT, "include", "a2amd_walk.h"), os.path.join(ROOT, "include", "a2amd_vm.h")]
    out = os.path.join(HERE, "liba2amd_units.so")
    if force or _newer(out, deps):
        subprocess.run(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-o", out, src,
                        "-L" + HERE, "-la2amd", "-ldl", "-lpthread", "-lm", "-Wl,-rpath,$ORIGIN"], check=True)
    return out


def build_walk(force=False, engine=None):
    """INTEGRATION.md option C: liba2amd_walk.so, the engine's voice walk with the short cut for
    sleeping voices.  Compiled against the ENGINE's internal headers (src/internals.h), so it
    is built where the engine's source tree is (A2_ENGINE_SRC, default /root/reference) and for
    that engine version; elsewhere the library built before travels as it is."""
    engine = engine or os.environ.get("A2_ENGINE_SRC", "/root/reference")
    out = os.path.join(HERE, "liba2amd_walk.so")
    if not (os.path.exists(os.path.join(engine, "src", "internals.h")) and shutil.which("cmake")):
        return out if os.path.exists(out) else None
    src = os.path.join(HERE, "csrc", "a2amd_walk.c")
    inc = os.path.join(HERE, "_engine_include")
    deps = [src, os.path.join(ROOT, "include", "a2amd_walk.h"), os.path.join(ROOT, "include", "a2amd_vm.h"),
            os.path.join(HERE, "liba2amd_units.so")] + \
           [os.path.join(engine, "src", f) for f in ("internals.h", "config.h")] + \
           [os.path.join(engine, "include", f) for f in ("a2_vm.h", "a2_units.h", "audiality2.h.cmake")]
    if force or _newer(out, deps):
        subprocess.run(["cmake", f"-DENGINE={engine}", f"-DOUT={inc}", "-P", os.path.join(HERE, "csrc", "engine_header.cmake")],
                       check=True, stdout=subprocess.DEVNULL)
        subprocess.run(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-I" + inc] +
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
