"""The C-ABI library: loads without a GPU, exports every symbol the header
declares, and refuses to open (rather than fall back to the CPU) when there is
no HIP device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols(header="a2amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(a2amd_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for need in ("a2amd_open", "a2amd_unit_init", "a2amd_unit_write", "a2amd_unit_process",
                 "a2amd_unit_deinit", "a2amd_inline_end", "a2amd_wave_upload", "a2amd_render",
                 "a2amd_fragment", "a2amd_rootbus"):
        assert need in syms


def test_library_exports_every_declared_symbol(gpu_lib):
    for s in declared_symbols():
        assert hasattr(gpu_lib, s), f"liba2amd.so lacks {s}"
    # the device VM's entry points
    vm = declared_symbols("a2amd_vm.h")
    assert "a2amd_vm_adopt" in vm and "a2amd_vm_recall" in vm and "a2amd_vm_analyze" in vm
    for s in vm:
        assert hasattr(gpu_lib, s), f"liba2amd.so lacks {s}"


def test_walk_and_units_libraries_export_their_interface():
    """include/a2amd_walk.h: what liba2amd_walk.so asks of liba2amd_units.so."""
    units = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
    if not os.path.exists(units):
        pytest.skip("liba2amd_units.so not built")
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", units], capture_output=True, text=True, check=True).stdout
    text = open(os.path.join(ROOT, "include", "a2amd_walk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for s in sorted(set(re.findall(r"\b(a2amd_units_[a-z_0-9]+)\s*\(", text))):
        assert f" {s}\n" in out, f"liba2amd_units.so lacks {s}"


def test_oracle_mirrors_the_call_protocol(oracle_lib):
    for s in declared_symbols():
        o = s.replace("a2amd_", "a2o_")
        if s in ("a2amd_version", "a2amd_source_stamp", "a2amd_device_count", "a2amd_rootbus", "a2amd_rootbus_copy", "a2amd_get_stats", "a2amd_set_profiling",
                 "a2amd_replay", "a2amd_collect", "a2amd_fragment_offset", "a2amd_capture_begin", "a2amd_capture_end", "a2amd_capture_frames", "a2amd_capture_free", "a2amd_wave_upload_captured", "a2amd_wave_stats", "a2amd_voice_process", "a2amd_voice_slot", "a2amd_voice_markable", "a2amd_default_hold", "a2amd_default_release_all", "a2amd_default_map", "a2amd_dist_unique_id", "a2amd_dist_init", "a2amd_dist_init_local", "a2amd_render_group", "a2amd_unit_insertable", "a2amd_unit_insert", "a2amd_render_paused"):
            continue
        assert hasattr(oracle_lib, o), f"oracle lacks {o}"


def test_open_fails_loudly_without_gpu(gpu_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from audiality2_amd.replay import Backend
    with pytest.raises(RuntimeError, match="no usable HIP device|a2amd_open"):
        Backend(gpu_lib, "a2amd_", 48000, -492789, 2)


def test_product_does_not_link_the_oracle():
    """Nothing under audiality2_amd/ may reference oracle/ (it is the checker)."""
    pkg = os.path.join(ROOT, "audiality2_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".c")) and fn != "build.py":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "a2o_" not in text, fn
                assert "liba2oracle" not in text and "oracle/" not in text.replace("oracle/_ref", ""), fn


def test_built_libraries_are_stamped_with_their_sources(gpu_lib):
    """build() decides by a hash of the sources compiled into each library (not by file times):
    the libraries in the tree must carry the stamp of the tree."""
    from audiality2_amd import build as b
    lib = os.path.join(ROOT, "audiality2_amd", "liba2amd.so")
    gpu_lib.a2amd_source_stamp.restype = ctypes.c_char_p
    stamp = gpu_lib.a2amd_source_stamp().decode()
    assert stamp.startswith("A2AMD_SRCHASH:") and stamp[14:] == b.stamped_hash(lib)
    assert b.build_lib() == lib and b.stamped_hash(lib) == stamp[14:]      # (nothing to rebuild)
    for name in ("liba2amd_units.so", "liba2amd_walk.so"):
        p = os.path.join(ROOT, "audiality2_amd", name)
        if os.path.exists(p):
            assert b.stamped_hash(p), f"{name} carries no source stamp"
    # a stale walk library is refused where it cannot be rebuilt
    walk = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")
    if os.path.exists(walk):
        assert b.build_walk(engine="/nonexistent") == walk
