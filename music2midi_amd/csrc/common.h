// Shared host/device helpers for the music2midi_amd HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/music2midi_amd.h"

namespace m2m {

// ---------------------------------------------------------------- errors ---
void set_error(const char* fmt, ...);

#define M2M_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::m2m::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                       __FILE__, __LINE__);                                        \
      return M2M_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

#define M2M_REQUIRE(cond, ...)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      ::m2m::set_error(__VA_ARGS__);                                               \
      return M2M_ERR_INVALID;                                                      \
    }                                                                              \
  } while (0)

// ------------------------------------------------------------ bf16 / f32 ---
// Storage element types: float (parity mode) or bf16_t (throughput mode).
struct bf16_t {
  uint16_t bits;
};

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)v.bits << 16;
  return c.f;
}

// round-to-nearest-even; a plain cast keeps NaNs NaN (v_cvt_pk_bf16_f32 on gfx950).
__device__ inline bf16_t f32_to_bf16(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);
  bf16_t r;
  r.bits = *reinterpret_cast<uint16_t*>(&h);
  return r;
}

template <typename T> __device__ inline T from_f32(float f);
template <> __device__ inline float from_f32<float>(float f) { return f; }
template <> __device__ inline bf16_t from_f32<bf16_t>(float f) { return f32_to_bf16(f); }

__device__ inline float to_f32(float v) { return v; }
__device__ inline float to_f32(bf16_t v) { return bf16_to_f32(v); }

// 16-byte vectors of storage elements.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 v;
  __device__ inline float get(int i) const { return reinterpret_cast<const float*>(&v)[i]; }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 v;
  __device__ inline float get(int i) const {
    uint32_t w = reinterpret_cast<const uint32_t*>(&v)[i >> 1];
    union { uint32_t u; float f; } c;
    c.u = (i & 1) ? (w & 0xFFFF0000u) : (w << 16);
    return c.f;
  }
};

// ------------------------------------------------------------- wave ops ---
// Cross-lane exchanges without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: an LDS-pipe
// round trip of ~100 clocks per step, and the steps of a reduction are dependent):
//   xor 1, 2   DPP quad_perm            xor 4   two DPP row shifts under complementary bank masks
//   xor 8      DPP row_ror:8            xor 16 / 32   v_permlane16_swap / v_permlane32_swap (gfx950)
// lane_xor<M>(v) returns v of lane (lane ^ M) for ANY data, so a reduction built from it pairs the same
// operands as the __shfl_xor butterfly it replaces and is bit-identical to it.
// (The swap instructions are issued as inline asm: __builtin_amdgcn_permlane32_swap with both operands
// the same value is miscompiled by ROCm 7.2 - tools/permlane_test.hip.)
template <int CTRL, int BANK_MASK = 0xF>
__device__ inline float dpp_mov(float v, float old = 0.f) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               0xF, BANK_MASK, BANK_MASK == 0xF));
}
template <int M>
__device__ inline float lane_xor(float v) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "lane_xor: power of two below 64");
  if constexpr (M == 1) return dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  else if constexpr (M == 4) {
    // banks (4-lane groups) 0 and 2 of every row take lane + 4 (row_shl:4), banks 1 and 3 lane - 4 (row_shr:4)
    const float up = dpp_mov<0x104, 0x5>(v, v);
    return dpp_mov<0x114, 0xA>(v, up);
  } else if constexpr (M == 8) return dpp_mov<0x128>(v);  // row_ror:8
  else {
    float a = v, b = v;
    if constexpr (M == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    // M == 32: a = [lo, lo], b = [hi, hi] (halves of the wave); M == 16: a = [r0, r0, r2, r2], b = [r1, r1, r3, r3]
    const int lane = __lane_id();
    return (lane & M) ? a : b;
  }
}
__device__ inline float wave_sum(float v) {
  v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v);
  v += lane_xor<4>(v);  v += lane_xor<2>(v);  v += lane_xor<1>(v);
  return v;
}
__device__ inline float wave_max(float v) {
  v = fmaxf(v, lane_xor<32>(v)); v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v));
  v = fmaxf(v, lane_xor<4>(v));  v = fmaxf(v, lane_xor<2>(v));  v = fmaxf(v, lane_xor<1>(v));
  return v;
}

// Sum over aligned groups of 8 or 16 lanes with DPP operand modifiers (plain VALU adds: no LDS
// crossbar round trip as __shfl_xor's ds_bpermute has).  Same pairing as an xor-butterfly 1,2,4(,8):
// after the two quad steps the value is quad-uniform, so the half-row / row mirrors fetch exactly the
// partner quad / half a butterfly would — results are bit-identical to it.
template <int LANES>
__device__ inline float group_sum(float v) {
  static_assert(LANES == 8 || LANES == 16, "group_sum: 8 or 16 lanes");
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  if constexpr (LANES == 16) v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}

// gelu_new (hf: activations.py NewGELUActivation), fp32, accurate tanhf.
__device__ inline float gelu_new(float x) {
  const float k = 0.7978845608028654f;  // sqrt(2/pi)
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
// The same function for results that are rounded to bf16 anyway: tanh(z) = 1 - 2 / (exp(2z) + 1) through the
// hardware exp2 / rcp (relative error ~1e-6 against bf16's 4e-3); ~8 instructions instead of ~40 for tanhf,
// which was half of the gated GEMM's time (32 M activations per launch).  Saturates correctly: exp2 -> inf / 0.
__device__ inline float gelu_new_fast(float x) {
  const float z2 = 2.0f * 0.7978845608028654f * 1.4426950408889634f * (x + 0.044715f * x * x * x);   // 2 z log2(e)
  const float t = __builtin_amdgcn_exp2f(z2);
  const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
  return 0.5f * x * (1.0f + th);
}
// gelu_new(x) and its derivative from ONE tanh: accurate (tanhf) in the fp32 parity mode, through the hardware exp2 / rcp where the
// results are rounded to bf16 (the training step's gate gradient)
template <typename T> __device__ inline void gelu_new_both_t(float x, float* g, float* dg) {
  const float k = 0.7978845608028654f;
  const float u = k * (x + 0.044715f * x * x * x);
  float th;
  if constexpr (sizeof(T) == 2) th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.0f * 1.4426950408889634f * u) + 1.0f);
  else th = tanhf(u);
  *g = 0.5f * x * (1.0f + th);
  *dg = 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * k * (1.0f + 3.0f * 0.044715f * x * x);
}
// accurate in the fp32 (parity) mode, fast where the result is stored as bf16
template <typename T> __device__ inline float gelu_new_t(float x) {
  if constexpr (sizeof(T) == 2) return gelu_new_fast(x);
  else return gelu_new(x);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remember it per device (a process may
// use cuda:1 after cuda:0), race-free (the threaded Flask server calls in from several threads).
#define M2M_OPT_IN_LDS(kernel_ptr, bytes)                                                             \
  do {                                                                                                \
    static std::atomic<bool> m2m_lds_set[64];                                                         \
    int m2m_dev = 0;                                                                                  \
    M2M_CHECK_HIP(hipGetDevice(&m2m_dev));                                                            \
    if (m2m_dev < 0 || m2m_dev >= 64 || !m2m_lds_set[m2m_dev].load(std::memory_order_acquire)) {      \
      M2M_CHECK_HIP(hipFuncSetAttribute((const void*)(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
      if (m2m_dev >= 0 && m2m_dev < 64) m2m_lds_set[m2m_dev].store(true, std::memory_order_release);  \
    }                                                                                                 \
  } while (0)

// Counter-based dropout: element i of site `key` is KEPT iff the high 32 bits of splitmix64(key + i) are >= thresh
// (thresh = p * 2^32), and kept values are scaled by 1 / (1 - p) — the same masks are regenerated in the backward pass,
// nothing is stored.  (oracle/train.py mirrors the hash in numpy to reproduce the masks.)
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline bool drop_keep(uint64_t key, int64_t i, uint32_t thresh) {
  // FOUR consecutive elements share one hash: element i takes the 16-bit field (i & 3) of splitmix64(key + (i >> 2)) and is kept
  // iff field >= thresh >> 16 (p in steps of 2^-16).  The kernels touch elements four at a time (drop_keep4), so a step hashes a
  // quarter of what one hash per element cost (~150 M elements at 16 clips: ~0.25 ms of VALU time).
  const uint64_t h = splitmix64(key + ((uint64_t)i >> 2));
  return (uint32_t)((h >> (16 * (int)(i & 3))) & 0xFFFFu) >= (thresh >> 16);
}
// i4 a multiple of 4: bit e of the result = drop_keep(key, i4 + e, thresh)
__host__ __device__ inline uint32_t drop_keep4(uint64_t key, int64_t i4, uint32_t thresh) {
  const uint64_t h = splitmix64(key + ((uint64_t)i4 >> 2));
  const uint32_t t16 = thresh >> 16;
  return ((uint32_t)(h & 0xFFFFu) >= t16 ? 1u : 0u) | ((uint32_t)((h >> 16) & 0xFFFFu) >= t16 ? 2u : 0u) |
         ((uint32_t)((h >> 32) & 0xFFFFu) >= t16 ? 4u : 0u) | ((uint32_t)(h >> 48) >= t16 ? 8u : 0u);
}

// The key of a dropout site is splitmix64(*step + salt): `step` is a device word the trainer advances once per
// forward/backward (so a captured HIP graph replays with fresh masks), `salt` identifies the site.
struct DropKey {
  const uint64_t* step;
  uint64_t salt;
};
__device__ inline uint64_t drop_site_key(DropKey k) { return splitmix64(*k.step + k.salt); }

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

}  // namespace m2m
