// Device-side launch trace (opt-in; rlzero_amd/trace.py, profiles/lane_timeline.py): a kernel that is handed a trace buffer leaves one
// record per workgroup -- start and end on the chip's constant 100 MHz clock (s_memrealtime), which step it worked on, which CU
// it ran on -- so the schedule of the lanes' kernels can be read WITHOUT a profiler in the way (rocprofv3 serialises the four
// queues of the shipped layout).  No atomics, nothing returned to the wave: a record is four fire-and-forget stores at a place
// computed from (kind, step, block), so a search overwrites the records of the search before it and the buffer holds the LAST one.
//   buffer (one per lane of games): u64 [2 + 4 * 2 * slots * blocks]: [0] slots (steps of a search), [1] blocks (games of the lane),
//   then records of 4 x u64 at ((kind - 1) * slots + step) * blocks + block: {t0, t1, kind << 56 | step << 32 | block, xcc << 32 | HW_ID}
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { RZ_TRACE_TRUNK = 1, RZ_TRACE_TREE = 2 };

__device__ __forceinline__ unsigned long long rz_trace_now() { return __builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ void rz_trace_write(unsigned long long *buf, int kind, int step, unsigned block, unsigned long long t0) {
    const unsigned long long t1 = rz_trace_now();
    const unsigned long long slots = buf[0], blocks = buf[1];
    if (step < 0 || (unsigned long long)step >= slots || block >= blocks) return;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: wave, SIMD, CU, SH, SE
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID, bits 0 .. 3
    unsigned long long *r = buf + 2 + 4 * (((unsigned long long)(kind - 1) * slots + (unsigned)step) * blocks + block);
    r[0] = t0;
    r[1] = t1;
    r[2] = ((unsigned long long)kind << 56) | ((unsigned long long)(unsigned)step << 32) | block;
    r[3] = ((unsigned long long)(xcc & 15u) << 32) | hw;
}
