// wave_dpp.hpp -- wave-uniform minima over the 64 lanes with the DPP modifier ON the v_min (gfx9 DPP controls:
// four steps inside the rows of 16 lanes, row_bcast:15 / row_bcast:31 across them; lane 63 ends up with the minimum).
// Six vector instructions + one v_readlane, where mov_dpp + min pairs, four v_readlane and three scalar minima took
// fifteen (SN_WAVE_MIN_PLAIN restores that form for A/B).
//
// Hazards: nobody inserts wait states inside inline assembly -- the hazard recognizer cannot see a DPP there.
//   * a DPP operand written by the previous VALU instruction needs 2 wait states  -> "s_nop 1" in front of every step;
//   * a VALU write of EXEC (v_cmpx ...) followed by a DPP needs 5 wait states      -> "s_nop 4" in front of the FIRST
//     step (the later ones follow a DPP step + its own nop, far enough from anything the compiler put before).
// row_bcast and the absence of bound_ctrl need a FULL exec mask: every caller invokes these from wave-uniform code with
// all 64 lanes active (asserted in debug builds, SN_DPP_ASSERT).  row_bcast exists on gfx9 only (gone in gfx10+).
#pragma once
#include <hip/hip_runtime.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "wave_dpp.hpp: row_bcast DPP controls exist on gfx9 (CDNA) only; this library is built for gfx950"
#endif

namespace sn {

__device__ __forceinline__ void dpp_assert_full_exec() {
#ifdef SN_DPP_ASSERT
  if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#ifdef SN_WAVE_MIN_PLAIN
  auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
  v = mn(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));   // quad_perm 1,0,3,2
  v = mn(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));   // quad_perm 2,3,0,1
  v = mn(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));  // row_half_mirror
  v = mn(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));  // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return mn(mn(a, b), mn(c, d));
#else
  dpp_assert_full_exec();
  asm volatile("s_nop 4\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

// the same for finite floats (the renderer's tile minimum)
__device__ __forceinline__ float wave_min_f32(float m) {
  dpp_assert_full_exec();
  asm volatile("s_nop 4\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(m));
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(m));
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(m));
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(m));
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(m));
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(m));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
}

}  // namespace sn
