"""include/a2amd_plugin.h must lay the plugin-boundary structs out exactly like
the reference's public headers.  Compiles a C file that includes both and
static-asserts sizes and offsets (only where the reference tree is mounted)."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
REFINC = os.path.join(ROOT, "oracle", "_ref", "include")

SRC = r'''
#include <stddef.h>
#include "audiality2.h"
#include "a2_units.h"
#include "a2_waves.h"
#include "a2_drivers.h"
#include "a2_properties.h"
#include "xinsert.h"            /* engine-internal: src/units/xinsert.h */
/* the plugin header declares the a2_*_unitdesc symbols with its own type; keep
 * the two sets of declarations apart */
#define a2_wtosc_unitdesc p_wtosc
#define a2_panmix_unitdesc p_panmix
#define a2_filter12_unitdesc p_f12
#define a2_fbdelay_unitdesc p_fbd
#define a2_inline_unitdesc p_inl
#define a2_xinsert_unitdesc p_xi
#define a2_xsink_unitdesc p_xsink
#define a2_xsource_unitdesc p_xsrc
#include "a2amd_plugin.h"
#define SAME(A, B, m) _Static_assert(offsetof(A, m) == offsetof(B, m), #m)
#define SIZE(A, B) _Static_assert(sizeof(A) == sizeof(B), #A)
SIZE(A2_unit, A2P_unit); SAME(A2_unit, A2P_unit, next); SAME(A2_unit, A2P_unit, descriptor);
SAME(A2_unit, A2P_unit, ninputs); SAME(A2_unit, A2P_unit, noutputs); SAME(A2_unit, A2P_unit, inputs);
SAME(A2_unit, A2P_unit, outputs); SAME(A2_unit, A2P_unit, registers); SAME(A2_unit, A2P_unit, coutputs);
SAME(A2_unit, A2P_unit, Process);
SAME(A2_xinsert, A2P_xinsert, state); SAME(A2_xinsert, A2P_xinsert, clients); SAME(A2_xinsert, A2P_xinsert, SetProcess);
SIZE(A2_xinsert_client, A2P_xinsert_client); SAME(A2_xinsert_client, A2P_xinsert_client, next);
SAME(A2_xinsert_client, A2P_xinsert_client, callback); SAME(A2_xinsert_client, A2P_xinsert_client, userdata);
SAME(A2_xinsert_client, A2P_xinsert_client, flags);
_Static_assert(A2_XI_READ == A2P_XI_READ && A2_XI_WRITE == A2P_XI_WRITE, "xiflags");
SIZE(A2_unitdesc, A2P_unitdesc); SAME(A2_unitdesc, A2P_unitdesc, name); SAME(A2_unitdesc, A2P_unitdesc, flags);
SAME(A2_unitdesc, A2P_unitdesc, registers); SAME(A2_unitdesc, A2P_unitdesc, coutputs);
SAME(A2_unitdesc, A2P_unitdesc, constants); SAME(A2_unitdesc, A2P_unitdesc, mininputs);
SAME(A2_unitdesc, A2P_unitdesc, maxoutputs); SAME(A2_unitdesc, A2P_unitdesc, instancesize);
SAME(A2_unitdesc, A2P_unitdesc, Initialize); SAME(A2_unitdesc, A2P_unitdesc, Deinitialize);
SAME(A2_unitdesc, A2P_unitdesc, OpenState); SAME(A2_unitdesc, A2P_unitdesc, CloseState);
SIZE(A2_vmstate, A2P_vmstate); SAME(A2_vmstate, A2P_vmstate, waketime); SAME(A2_vmstate, A2P_vmstate, r);
SIZE(A2_config, A2P_config); SAME(A2_config, A2P_config, interface); SAME(A2_config, A2P_config, samplerate);
SAME(A2_config, A2P_config, channels); SAME(A2_config, A2P_config, basepitch);
SIZE(A2_driver, A2P_driver); SAME(A2_driver, A2P_driver, config); SAME(A2_driver, A2P_driver, type);
SAME(A2_driver, A2P_driver, Destroy);
SIZE(A2_audiodriver, A2P_audiodriver); SAME(A2_audiodriver, A2P_audiodriver, Run);
SAME(A2_audiodriver, A2P_audiodriver, state); SAME(A2_audiodriver, A2P_audiodriver, Process);
SAME(A2_audiodriver, A2P_audiodriver, buffers);
_Static_assert(A2_AUDIODRIVER == A2P_AUDIODRIVER, "drivertype");
_Static_assert(A2_OOMEMORY == A2P_OOMEMORY && A2_NOTIMPLEMENTED == A2P_NOTIMPLEMENTED &&
		A2_DEVICEOPEN == A2P_DEVICEOPEN && A2_INTERNAL == A2P_INTERNAL, "error codes");
_Static_assert(sizeof(A2_wave) == sizeof(A2P_wave), "A2_wave");
_Static_assert(offsetof(A2_wave, d.wave.data) == offsetof(A2P_wave, data), "data");
_Static_assert(offsetof(A2_wave, d.wave.size) == offsetof(A2P_wave, size), "size");
_Static_assert(offsetof(A2_wave, period) == offsetof(A2P_wave, period), "period");
_Static_assert(A2_BLOCK_SIZE == A2P_BLOCK_SIZE, "block");
_Static_assert(A2_PNOISESEED == A2P_PNOISESEED, "noiseseed");
_Static_assert(R_TRANSPOSE == A2P_R_TRANSPOSE, "transpose");
_Static_assert(A2_MATCHIO == A2P_MATCHIO && A2_XINSERT == A2P_XINSERT, "flags");
_Static_assert(A2_MAXFRAG == 64 && A2_MAXCHANNELS == 8 && A2_MIPLEVELS == 10, "limits");
_Static_assert(A2_WAVEPRE == 1 && A2_WAVEPOST == 131, "pads");
/* the device VM (include/a2amd_vm.h) interprets the engine's bytecode: opcode numbers, VM state
 * layout and limits are the engine's (src/internals.h:152-224, include/a2_vm.h, src/config.h:119) */
#include "internals.h"
#include "a2amd_vm.h"
#define A2V(x) _Static_assert((int)OP_##x == (int)A2AMD_OP_##x, "opcode " #x);
A2AMD_VM_ALLOPS
#undef A2V
_Static_assert((int)A2_OPCODES == (int)A2AMD_VM_OPCODES, "opcodes");
_Static_assert(A2_REGISTERS == A2AMD_VM_REGISTERS && A2_INSLIMIT == A2AMD_VM_INSLIMIT, "VM limits");
_Static_assert(sizeof(A2_vmstate) == sizeof(a2amd_vm_state) && offsetof(A2_vmstate, r) == offsetof(a2amd_vm_state, r) &&
		offsetof(A2_vmstate, pc) == offsetof(a2amd_vm_state, pc) && offsetof(A2_vmstate, func) == offsetof(a2amd_vm_state, func) &&
		offsetof(A2_vmstate, state) == offsetof(a2amd_vm_state, state), "A2_vmstate");
_Static_assert(sizeof(A2_instruction) == 8 && offsetof(A2_instruction, a3) == 4, "A2_instruction");
_Static_assert((int)A2_WAITING == (int)A2AMD_VM_WAITING && (int)A2_ENDING == (int)A2AMD_VM_ENDING, "VM states");
_Static_assert(R_TICK == A2AMD_VM_R_TICK && R_TRANSPOSE == A2AMD_VM_R_TRANSPOSE, "fixed registers");
int main(void) { return 0; }
'''


def test_plugin_structs_match_reference_headers(tmp_path):
    if not (os.path.isdir(REF) and os.path.exists(os.path.join(REFINC, "audiality2.h"))):
        pytest.skip("reference tree / generated header not available here")
    src = tmp_path / "abi.c"
    src.write_text(SRC)
    subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", f"-I{REFINC}", f"-I{REF}/include", f"-I{REF}/src", f"-I{REF}/src/units", f"-I{REF}/src/drivers",
                    f"-I{ROOT}/include", str(src)], check=True)
