// dpp.h -- cross-lane moves and wave-wide sums on the VALU (DPP) instead of ds_bpermute (shared by scan.hip and pcg_persist.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace pplie {

// ---- cross-lane moves on the VALU (DPP) instead of ds_bpermute --------------------------------------------
// A wave-wide inclusive scan is 7 combine steps (GCN3 cross-lane recipe): row_shr:1,2,3 of the ORIGINAL values,
// row_shr:4 (banks 1-3), row_shr:8 (banks 2-3), row_bcast:15 (rows 1,3), row_bcast:31 (rows 2,3).  A lane a step
// does not reach keeps `old` (bound_ctrl = 0), which is the identity of the scan's operation, so every lane can
// combine unconditionally.  Each move is one VALU instruction; __shfl_up/down compile to LDS permutes with an
// address computation and a wait each -- on these latency-bound scan kernels that was most of the time.
template <int CTRL, int RM, int BM> __device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, RM, BM, false));
}
template <int CTRL, int RM, int BM> __device__ __forceinline__ double dpp_mov(double old, double src) {
  const unsigned long long o = __builtin_bit_cast(unsigned long long, old), v = __builtin_bit_cast(unsigned long long, src);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)v, CTRL, RM, BM, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(v >> 32), CTRL, RM, BM, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
enum { DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR3 = 0x113, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118,
       DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143 };

// value of lane-1 (lane 0 receives `first`)
template <class T> __device__ __forceinline__ T lane_shift_up1(T v, T first) { return dpp_mov<DPP_WAVE_SHR1, 0xf, 0xf>(first, v); }
__device__ __forceinline__ float lane_bcast63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ double lane_bcast63(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// inclusive prefix sum over the 64 lanes
template <class T> __device__ __forceinline__ T wave_prefix_add(T v) {
  const T v0 = v, z = T(0);
  v = dpp_mov<DPP_ROW_SHR1, 0xf, 0xf>(z, v0) + v;
  v = dpp_mov<DPP_ROW_SHR2, 0xf, 0xf>(z, v0) + v;
  v = dpp_mov<DPP_ROW_SHR3, 0xf, 0xf>(z, v0) + v;
  v = dpp_mov<DPP_ROW_SHR4, 0xf, 0xe>(z, v) + v;
  v = dpp_mov<DPP_ROW_SHR8, 0xf, 0xc>(z, v) + v;
  v = dpp_mov<DPP_ROW_BCAST15, 0xa, 0xf>(z, v) + v;
  v = dpp_mov<DPP_ROW_BCAST31, 0xc, 0xf>(z, v) + v;
  return v;
}

// sum over the 64 lanes, valid in lane 63 (the last lane of the inclusive scan above)
template <class T> __device__ __forceinline__ T wave_total63(T v) { return wave_prefix_add<T>(v); }

}  // namespace pplie
