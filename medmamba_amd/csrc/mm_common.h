// Shared device helpers for the gfx950 kernels (wave64 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm {

constexpr int kWave = 64;
constexpr int kNState = 16;      // d_state of every MedMamba block (MedMamba.py:329, 457)
constexpr int kTile = 64;        // timesteps staged through LDS per tile
constexpr int kTileStride = 68;  // LDS row stride in floats: +16 B pad -> conflict-free b128 rows
constexpr int kChunk = 16;       // checkpoint interval (steps) of x_chk
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// DPP controls (gfx9 encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;    // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_ROR8 = 0x128;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Butterfly sum over the WIDTH (1,2,4,8,16) consecutive lanes that share lane/WIDTH; every lane gets the sum.
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (WIDTH >= 2) v += dpp_f<DPP_QUAD_XOR1>(v);
  if constexpr (WIDTH >= 4) v += dpp_f<DPP_QUAD_XOR2>(v);
  if constexpr (WIDTH >= 8) v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
  if constexpr (WIDTH >= 16) v += dpp_f<DPP_ROW_MIRROR>(v);
  return v;
}

// softplus with F.softplus semantics (beta 1, threshold 20: temp.py:63-64), built from v_exp/v_log.
// e = exp(x); for e < 2^-12 the series log1p(e) = e - e^2/2 is exact to fp32 and avoids 1+e rounding.
__device__ __forceinline__ float softplus_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * kLog2e);
  const float big = __builtin_amdgcn_logf(1.0f + e) * kLn2;   // v_log_f32 is log2
  const float small = e - 0.5f * e * e;
  float r = e < 2.44140625e-4f ? small : big;
  return x > 20.0f ? x : r;
}

__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * kLog2e));
}

}  // namespace mm
