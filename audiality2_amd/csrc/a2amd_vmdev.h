// a2amd_vmdev.h - what the kernels that run the device VM lane = voice share (a2amd_vm.hip, a2amd_vmwin.hip):
// a voice's working copy in LDS, and the program text of the wavefront's voices staged there.
#ifndef A2AMD_VMDEV_H
#define A2AMD_VMDEV_H
#include "a2amd_device.h"
#include "a2amd_vmcore.h"

// A voice's working copy lives in LDS, one per lane: the interpreter indexes the register file, the control
// map, the env units and the cutoff rampers with run-time values, and a struct indexed like that in "registers"
// is a struct in scratch memory (round 4: 476 bytes of private segment per lane, every VM register access a
// trip to the vector cache).  One word of padding makes a lane's stride odd: the same word of 64 voices lies in
// 32 different banks.
struct VmSlot { A2DVmVoice v; a2vm::Tracker rt; int32_t pad[((sizeof(A2DVmVoice) + sizeof(a2vm::Tracker)) / 4) % 2 ? 0 : 1]; };
static_assert((sizeof(VmSlot) / 4) % 2 == 1, "an odd stride in words");

// The interpreter fetches one instruction word (two for the long forms) per trip through its loop, each fetch a
// round trip the next one waits for: from memory that is most of a VM run's time (measured, 16 384 voices of one
// script: 60 % of k_vm_win).  The voices of a wavefront mostly run a handful of programs - an instrument's voices
// all the same one - so the wavefront copies the text of up to VM_NPROGS distinct functions of its voices into LDS
// (VM_CODEWORDS words in all) and a lane whose function made it there fetches from the copy; the others - a
// wavefront of many different or very long programs - keep reading memory.  Returns the lane's text.
#define VM_CODEWORDS 2048
#define VM_NPROGS    8
__device__ inline const uint32_t *vm_stage_code(const uint32_t *pool, const A2DVmVoice &v, bool has, uint32_t *s_code,
		unsigned room = VM_CODEWORDS)
{
	const int lane = (int)(threadIdx.x & 63);
	const uint32_t *mine = pool + v.code;
	unsigned long long todo = __ballot(has);
	unsigned used = 0;
	for(int k = 0; k < VM_NPROGS && todo; ++k) {
		const int leader = __ffsll((long long)todo) - 1;
		const uint32_t base = (uint32_t)__shfl((int)v.code, leader, 64), n = (uint32_t)__shfl((int)v.ncode, leader, 64);
		const unsigned long long same = __ballot(has && v.code == base && v.ncode == n);
		if(used + n <= room) {
			for(uint32_t q = (uint32_t)lane; q < n; q += 64)
				s_code[used + q] = pool[base + q];
			if(has && v.code == base && v.ncode == n)
				mine = s_code + used;
			used += n;
		}
		todo &= ~same;
	}
	return mine;
}

#endif /* A2AMD_VMDEV_H */
