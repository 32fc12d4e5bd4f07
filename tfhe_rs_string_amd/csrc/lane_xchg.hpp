// lane_xchg.hpp -- cross-lane exchanges of f64 values with v_permlane32_swap / v_permlane16_swap (gfx950), shared by
// the blind-rotation kernels.  Device-only (the host emulation in tests/emu performs the same exchanges by hand).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe_hip {
namespace xchg {

// between two wave-local stages: a wavefront's DS instructions execute in order, the compiler must too
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// lanes [32,63] of `a` <-> lanes [0,31] of `b`
__device__ __forceinline__ void swap32(double &a, double &b)
{
    uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
    uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// rows (16 lanes) 1, 3 of `a` <-> rows 0, 2 of `b`
__device__ __forceinline__ void swap16(double &a, double &b)
{
    uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
    uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// 4 x 4 transposition between the register index of R[0..3] (complex values) and lane bits (5,4)
template <class LaneT> __device__ __forceinline__ void transpose54(LaneT &s)
{
    swap32(s.R[0].re, s.R[2].re); swap32(s.R[0].im, s.R[2].im);
    swap32(s.R[1].re, s.R[3].re); swap32(s.R[1].im, s.R[3].im);
    swap16(s.R[0].re, s.R[1].re); swap16(s.R[0].im, s.R[1].im);
    swap16(s.R[2].re, s.R[3].re); swap16(s.R[2].im, s.R[3].im);
}

} // namespace xchg
} // namespace tfhe_hip
