// MFMA fragment helpers shared by the encoder GEMM/attention and the decode projections.
//
// One "macro step" multiplies a 32-row A block by a 32-column B block over 16 consecutive k:
//   bf16:  one v_mfma_f32_32x32x16_bf16       (lane (r,h) supplies A[r][8h+j], B[8h+j][r], j=0..7)
//   fp32:  eight v_mfma_f32_32x32x2_f32        (MFMA j: lane (r,h) supplies A[r][8h+j], B[8h+j][r])
// so for both storage types a lane's fragment is the SAME 8 contiguous elements
// [k0 + 8h, k0 + 8h + 8) of its row/column.  The f32 form is an exact fp32 FMA chain
// (MI355X_MICROARCH.md "Matrix cores"), which is what the parity mode needs.
//
// Accumulator layout (both): element i of lane l is  row (i&3) + 8*(i>>2) + 4*(l>>5),  col l&31.
#pragma once

#include "common.h"

namespace m2m {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  uint4 v;
};
template <> struct Frag<float> {
  float4 lo, hi;
};

// 8 contiguous elements from a 16-byte aligned address (LDS or global).
__device__ inline Frag<bf16_t> load_frag(const bf16_t* p) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const uint4*>(p);
  return f;
}
__device__ inline Frag<float> load_frag(const float* p) {
  Frag<float> f;
  f.lo = *reinterpret_cast<const float4*>(p);
  f.hi = *reinterpret_cast<const float4*>(p + 4);
  return f;
}

template <typename T> __device__ inline Frag<T> zero_frag();
template <> __device__ inline Frag<bf16_t> zero_frag<bf16_t>() {
  Frag<bf16_t> f;
  f.v = make_uint4(0, 0, 0, 0);
  return f;
}
template <> __device__ inline Frag<float> zero_frag<float>() {
  Frag<float> f;
  f.lo = make_float4(0, 0, 0, 0);
  f.hi = f.lo;
  return f;
}

// 8 fp32 values -> fragment of storage type T (round-to-nearest-even for bf16).
template <typename T> __device__ inline Frag<T> pack_frag(const float (&x)[8]);
template <> __device__ inline Frag<float> pack_frag<float>(const float (&x)[8]) {
  Frag<float> f;
  f.lo = make_float4(x[0], x[1], x[2], x[3]);
  f.hi = make_float4(x[4], x[5], x[6], x[7]);
  return f;
}
// two fp32 -> packed bf16 (a in the low half), round-to-nearest-even: ONE v_cvt_pk_bf16_f32 (element-wise casts + a merge cost three)
__device__ inline uint32_t pack2_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t x = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
}
template <> __device__ inline Frag<bf16_t> pack_frag<bf16_t>(const float (&x)[8]) {
  Frag<bf16_t> f;
  f.v = make_uint4(pack2_bf16(x[0], x[1]), pack2_bf16(x[2], x[3]), pack2_bf16(x[4], x[5]), pack2_bf16(x[6], x[7]));
  return f;
}

__device__ inline void mma16(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                acc, 0, 0, 0);
}
__device__ inline void mma16(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
}

__device__ inline f32x16 zero_acc() {
  f32x16 a;
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.f;
  return a;
}

// accumulator element i of lane -> row within the 32x32 tile
__device__ inline int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

}  // namespace m2m
