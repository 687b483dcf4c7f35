// MXFP8 products for the training step (BASELINE configs[4]: "fp8 MFMA GEMMs"; SURVEY §8f-1: OCP e4m3 / e5m2, block-scaled).
//
// Format (OCP Microscaling v1.0, as gfx950's v_mfma_scale_f32_32x32x64_f8f6f4 consumes it): along the REDUCTION dimension
// every 32 consecutive elements share one power-of-two scale 2^(e - 127), e an E8M0 byte; the elements are OCP FP8 —
// e4m3fn for forward operands (activations, weights), e5m2 for gradients.  scale = 2^(floor(log2(amax)) - emax_elem)
// with emax_elem = 8 (e4m3) / 15 (e5m2); elements that still exceed the format's largest finite value are clamped.
// No per-tensor amax pass exists: a quantiser is ONE launch, and the matrix cores dequantise for free — the scaled
// K = 64 MFMA runs at twice the bf16 rate (MI355X_MICROARCH.md, matrix cores).
//
// Lane map of the K = 64 MFMA, measured with one-hot operands (tools/mxprobe.py): lane l (r = l & 31, h = l >> 5) supplies
// row / column r; its operand bytes 0-15 are k = 16 h + j and bytes 16-31 are k = 32 + 16 h + j, while its scale byte
// (bits 0-7 of the scale operand, opsel 0) is the E8M0 scale of the k-block [32 h, 32 h + 32).  C/D layout is the 32x32
// one of every other MFMA (mma.h acc_row).
#include "mma.h"
#include "train.h"

#include <stdlib.h>

namespace m2m {

typedef int v8i_t __attribute__((ext_vector_type(8)));

// ---- quantisers ----------------------------------------------------------------------------------------------------
__device__ inline float mx_block_scale_exp(float amax, int emax_elem) {          // floor(log2(amax)) - emax_elem, clamped to E8M0
  if (!(amax > 0.f)) return -127.f;
  int e;
  (void)frexpf(amax, &e);                                                         // amax = m * 2^e, m in [0.5, 1) -> floor(log2) = e - 1
  int se = e - 1 - emax_elem;
  se = se < -127 ? -127 : (se > 127 ? 127 : se);
  return (float)se;
}
template <int FMT> __device__ inline uint32_t mx_pack4(float a, float b, float c, float d) {   // 4 floats -> 4 fp8 bytes (RNE, pre-clamped)
  constexpr float LIM = FMT == 0 ? 448.0f : 57344.0f;
  a = fminf(fmaxf(a, -LIM), LIM); b = fminf(fmaxf(b, -LIM), LIM); c = fminf(fmaxf(c, -LIM), LIM); d = fminf(fmaxf(d, -LIM), LIM);
  int w = 0;
  if constexpr (FMT == 0) { w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true); }
  else { w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true); }
  return (uint32_t)w;
}

// rows: src [R][C] (row stride ld_s) -> q [R][Cp] bytes, scales [R][Cp / 32]; blocks of 32 along the columns; columns >= C are zero.
// One thread per block.
template <typename TS, int FMT>
__global__ void mxq_rows_kernel(const TS* __restrict__ src, int64_t ld_s, uint8_t* __restrict__ q, uint8_t* __restrict__ sc, int R, int C, int Cp) {
  const int nb = Cp / 32;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)R * nb, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t row = i / nb;
    const int c0 = (int)(i - row * nb) * 32;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      v[j] = (c0 + j < C) ? to_f32(src[row * ld_s + c0 + j]) : 0.f;
      amax = fmaxf(amax, fabsf(v[j]));
    }
    const float se = mx_block_scale_exp(amax, FMT == 0 ? 8 : 15);
    const float inv = exp2f(-se);
    uint32_t* dst = reinterpret_cast<uint32_t*>(q + row * Cp + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = mx_pack4<FMT>(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
    sc[i] = (uint8_t)((int)se + 127);
  }
}

// columns: src [R][C] -> qt [C][Rp] bytes (the TRANSPOSE), scales [C][Rp / 32]; blocks of 32 along the rows; rows >= R are zero.
// A workgroup takes 32 rows x 64 columns through LDS; thread c < 64 owns column c's block.
template <typename TS, int FMT>
__global__ __launch_bounds__(64) void mxq_cols_kernel(const TS* __restrict__ src, int64_t ld_s, uint8_t* __restrict__ qt, uint8_t* __restrict__ sc,
                                                      int R, int C, int Rp) {
  __shared__ float tile[32][65];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 32 * 64; i += 64) {
    const int rl = i >> 6, cl = i & 63;
    tile[rl][cl] = (r0 + rl < R && c0 + cl < C) ? to_f32(src[(int64_t)(r0 + rl) * ld_s + c0 + cl]) : 0.f;
  }
  __syncthreads();
  const int c = c0 + threadIdx.x;
  if (c >= C) return;
  float v[32];
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) { v[j] = tile[j][threadIdx.x]; amax = fmaxf(amax, fabsf(v[j])); }
  const float se = mx_block_scale_exp(amax, FMT == 0 ? 8 : 15);
  const float inv = exp2f(-se);
  uint32_t* dst = reinterpret_cast<uint32_t*>(qt + (int64_t)c * Rp + r0);
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = mx_pack4<FMT>(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
  sc[(int64_t)c * (Rp / 32) + r0 / 32] = (uint8_t)((int)se + 127);
}

// ---- GEMM ------------------------------------------------------------------------------------------------------------
// C[M,N] (epi)= A[M,K] . B[N,K]^T on MXFP8 operands: A fp8 [M][lda] + scales [M][lda/32], B fp8 [N][ldb] + scales [N][ldb/32];
// lda, ldb multiples of 128, K <= lda (the padding is zero).  64x64 tile, BK = 128 bytes, 2 x 2 waves of one 32x32 MFMA tile.
constexpr int MX_BK = 128, MX_PITCH = MX_BK + 16;

template <int FA, int FB, int EPI>
__global__ __launch_bounds__(256) void mxgemm_kernel(MxGemmArgs g) {
  __shared__ __align__(16) uint8_t As[64 * MX_PITCH];
  __shared__ __align__(16) uint8_t Bs[64 * MX_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const bool split = g.ksplit > 1;
  const int kbeg = split ? blockIdx.z * g.kchunk : 0, kend = split ? min(g.K, kbeg + g.kchunk) : g.K;   // multiples of 128 by construction
  const int arow = min(m0 + wm * 32 + r, g.M - 1), brow = min(n0 + wn * 32 + r, g.N - 1);
  const uint32_t* sA = reinterpret_cast<const uint32_t*>(g.sA) + (int64_t)arow * (g.lda / 128);
  const uint32_t* sB = reinterpret_cast<const uint32_t*>(g.sB) + (int64_t)brow * (g.ldb / 128);

  f32x16 acc = zero_acc();
  for (int k0 = kbeg; k0 < kend; k0 += MX_BK) {
    __syncthreads();
    for (int c = tid; c < 64 * (MX_BK / 16); c += 256) {
      const int rl = c / (MX_BK / 16), kc = (c % (MX_BK / 16)) * 16;
      const int ra = m0 + rl, rb = n0 + rl;
      *reinterpret_cast<uint4*>(As + rl * MX_PITCH + kc) =
          ra < g.M ? *reinterpret_cast<const uint4*>(g.A + (int64_t)ra * g.lda + k0 + kc) : make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(Bs + rl * MX_PITCH + kc) =
          rb < g.N ? *reinterpret_cast<const uint4*>(g.B + (int64_t)rb * g.ldb + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
    const uint32_t sa4 = sA[k0 / 128], sb4 = sB[k0 / 128];     // the 4 scale bytes of this row's 4 blocks
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // operand bytes 0-15 of lane half h are k = 16 h + j, bytes 16-31 are k = 32 + 16 h + j of the 64-wide step
      // (measured with one-hot operands, tools/mxprobe.py); the lane's scale byte is the one of k-block h = [32 h, 32 h + 32)
      const uint8_t* ap = As + (wm * 32 + r) * MX_PITCH + s * 64 + h * 16;
      const uint8_t* bp = Bs + (wn * 32 + r) * MX_PITCH + s * 64 + h * 16;
      const uint4 a0 = *reinterpret_cast<const uint4*>(ap), a1 = *reinterpret_cast<const uint4*>(ap + 32);
      const uint4 b0 = *reinterpret_cast<const uint4*>(bp), b1 = *reinterpret_cast<const uint4*>(bp + 32);
      const v8i_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
      const v8i_t bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
      const int sa = (int)((sa4 >> (8 * (2 * s + h))) & 0xFF), sb = (int)((sb4 >> (8 * (2 * s + h))) & 0xFF);
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, FA, FB, 0, sa, 0, sb);
    }
  }
  const int col = n0 + wn * 32 + r;
  if (col >= g.N) return;
  const uint64_t dkey = (EPI == TG_RESID_F32 && g.drop_thresh) ? splitmix64(*g.drop_step + g.drop_key) : 0ull;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + wm * 32 + acc_row(i, lane);
    if (row >= g.M) continue;
    if (split) { g.Cpart[((int64_t)blockIdx.z * g.M + row) * g.N + col] = acc[i]; continue; }
    const int64_t at = (int64_t)row * g.ldc + col;
    const float v = acc[i];
    if constexpr (EPI == TG_STORE_T) reinterpret_cast<bf16_t*>(g.C)[at] = f32_to_bf16(v);
    else if constexpr (EPI == TG_STORE_F32) reinterpret_cast<float*>(g.C)[at] = v;
    else if constexpr (EPI == TG_ACC_F32) reinterpret_cast<float*>(g.C)[at] += v;
    else {
      float u = v;
      if (g.drop_thresh) u = drop_keep(dkey, at, g.drop_thresh) ? v * g.drop_scale : 0.f;
      reinterpret_cast<float*>(g.C)[at] = g.R[at] + u;
    }
  }
}

// The same product with the A operand given in bf16 and quantised IN the operand staging (no quantiser launch, no fp8 image of
// the activations in memory): a thread stages 16-byte chunks of 8 bf16, the 4 lanes that hold the 4 chunks of a 32-element block
// agree on the block maximum with two DPP exchanges, and the chunk goes to LDS as 8 fp8 bytes (+ one scale byte per block) —
// the arithmetic of mxq_rows_kernel, bit for bit.  64x64 tile, 128 k per step, next step's chunks prefetched into registers.
__device__ inline float mxq_quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));   // lane ^ 1
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));   // lane ^ 2
  return v;
}
// TN = 2: 64 x 128 tile (each wave 32 x 64): a workgroup's quantised rows serve twice the columns — every column tile of a product
// re-quantises its rows, so for the wide products (N >= 512) this halves the kernel's VALU time
template <int FA, int EPI, int TN>
__global__ __launch_bounds__(256) void mxgemm_q_kernel(MxGemmArgs g) {
  constexpr int BN = 64 * TN;
  __shared__ __align__(16) uint8_t As[64 * MX_PITCH];
  __shared__ __align__(16) uint8_t Bs[BN * MX_PITCH];
  __shared__ __align__(4) uint8_t Asc[64 * 4];                 // scale bytes of the A tile: [row][k-block of this step]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_total > 0) {                            // XCD-contiguous order: see MxGemmArgs
    const int per = (g.xcd_total + 7) >> 3, l = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (l >= g.xcd_total || (int)(blockIdx.x >> 3) >= per) return;        // padding workgroups (uniform, before any barrier)
    bx = l % g.xcd_nx; by = l / g.xcd_nx;
  }
  const int m0 = by * 64, n0 = bx * BN;
  const uint32_t* sB[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) sB[j] = reinterpret_cast<const uint32_t*>(g.sB) + (int64_t)min(n0 + (wn * TN + j) * 32 + r, g.N - 1) * (g.ldb / 128);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.Asrc);
  // staging assignments: A: 64 rows x 16 chunks of 8 bf16 = 4 per thread (chunk c: row c / 16, k chunk c % 16: the 4 lanes of a
  // quad hold one 32-element block); B: 64 rows x 8 chunks of 16 fp8 bytes = 2 per thread
  uint4 ra[4], rb[2 * TN];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = m0 + c / 16, k = k0 + (c % 16) * 8;
      ra[i] = (row < g.M && k < g.Kvalid) ? *reinterpret_cast<const uint4*>(A + (int64_t)row * g.ld_src + k) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) {
      const int c = tid + 256 * i, row = n0 + c / 8, kc = (c % 8) * 16;
      rb[i] = row < g.N ? *reinterpret_cast<const uint4*>(g.B + (int64_t)row * g.ldb + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, rl = c / 16, kc = c % 16;
      const uint32_t w4[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      float v[8];
      float amax = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(w4[j] << 16);
        v[2 * j + 1] = __uint_as_float(w4[j] & 0xFFFF0000u);
        amax = fmaxf(amax, fmaxf(fabsf(v[2 * j]), fabsf(v[2 * j + 1])));
      }
      amax = mxq_quad_max(amax);
      // floor(log2 amax) - emax from the exponent field (== mx_block_scale_exp's frexpf form for every finite amax: subnormals
      // and zero clamp to -127 either way), 2^-se built from its bits: ~45 instructions per chunk instead of ~70 — every
      // column tile of the product re-quantises its rows, so this is the kernel's VALU time
      int se = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - (FA == 0 ? 8 : 15);
      se = se < -127 ? -127 : se;                                  // (the upper clamp cannot bind: amax is a finite bf16)
      const float inv = __uint_as_float((uint32_t)(127 - se) << 23);
      constexpr float LIM = FA == 0 ? 448.0f : 57344.0f;
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = __builtin_amdgcn_fmed3f(v[j] * inv, -LIM, LIM);
      int q0 = 0, q1 = 0;
      if constexpr (FA == 0) {
        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(w[0], w[1], q0, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(w[2], w[3], q0, true);
        q1 = __builtin_amdgcn_cvt_pk_fp8_f32(w[4], w[5], q1, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(w[6], w[7], q1, true);
      } else {
        q0 = __builtin_amdgcn_cvt_pk_bf8_f32(w[0], w[1], q0, false); q0 = __builtin_amdgcn_cvt_pk_bf8_f32(w[2], w[3], q0, true);
        q1 = __builtin_amdgcn_cvt_pk_bf8_f32(w[4], w[5], q1, false); q1 = __builtin_amdgcn_cvt_pk_bf8_f32(w[6], w[7], q1, true);
      }
      *reinterpret_cast<uint2*>(As + rl * MX_PITCH + kc * 8) = make_uint2((uint32_t)q0, (uint32_t)q1);
      if ((kc & 3) == 0) Asc[rl * 4 + (kc >> 2)] = (uint8_t)(se + 127);
    }
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<uint4*>(Bs + (c / 8) * MX_PITCH + (c % 8) * 16) = rb[i];
    }
  };
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = zero_acc();
  gload(0);
  for (int k0 = 0; k0 < g.K; k0 += MX_BK) {
    __syncthreads();
    sstore();
    uint32_t sb4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) sb4[j] = sB[j][k0 / 128];
    __syncthreads();
    if (k0 + MX_BK < g.K) gload(k0 + MX_BK);
    const uint32_t sa4 = *reinterpret_cast<const uint32_t*>(Asc + (wm * 32 + r) * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint8_t* ap = As + (wm * 32 + r) * MX_PITCH + s * 64 + h * 16;
      const uint4 a0 = *reinterpret_cast<const uint4*>(ap), a1 = *reinterpret_cast<const uint4*>(ap + 32);
      const v8i_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
      const int sa = (int)((sa4 >> (8 * (2 * s + h))) & 0xFF);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint8_t* bp = Bs + ((wn * TN + j) * 32 + r) * MX_PITCH + s * 64 + h * 16;
        const uint4 b0 = *reinterpret_cast<const uint4*>(bp), b1 = *reinterpret_cast<const uint4*>(bp + 32);
        const v8i_t bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        const int sb = (int)((sb4[j] >> (8 * (2 * s + h))) & 0xFF);
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[j], FA, 0, 0, sa, 0, sb);
      }
    }
  }
  const uint64_t dkey = (EPI == TG_RESID_F32 && g.drop_thresh) ? splitmix64(*g.drop_step + g.drop_key) : 0ull;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + r;
    if (col >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + wm * 32 + acc_row(i, lane);
      if (row >= g.M) continue;
      const int64_t at = (int64_t)row * g.ldc + col;
      const float v = acc[j][i];
      if constexpr (EPI == TG_STORE_T) reinterpret_cast<bf16_t*>(g.C)[at] = f32_to_bf16(v);
      else if constexpr (EPI == TG_STORE_F32) reinterpret_cast<float*>(g.C)[at] = v;
      else if constexpr (EPI == TG_ACC_F32) reinterpret_cast<float*>(g.C)[at] += v;
      else {
        float u = v;
        if (g.drop_thresh) u = drop_keep(dkey, at, g.drop_thresh) ? v * g.drop_scale : 0.f;
        reinterpret_cast<float*>(g.C)[at] = g.R[at] + u;
      }
    }
  }
}

// Both operands pre-quantised (the activations' fp8 image is written by the kernel that produces them — RMSNorm, gated GELU, the
// gradient converters), 64 x (64 TN) tile, next k-step's 16-byte chunks prefetched into registers: the operand staging is byte
// copies only, half the bytes of the bf16 product, and no VALU work in the loop.
template <int FA, int EPI, int TN>
__global__ __launch_bounds__(256) void mxgemm_p_kernel(MxGemmArgs g) {
  constexpr int BN = 64 * TN;
  __shared__ __align__(16) uint8_t As[64 * MX_PITCH];
  __shared__ __align__(16) uint8_t Bs[BN * MX_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_total > 0) {                            // XCD-contiguous order: see MxGemmArgs
    const int per = (g.xcd_total + 7) >> 3, l = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (l >= g.xcd_total || (int)(blockIdx.x >> 3) >= per) return;        // padding workgroups (uniform, before any barrier)
    bx = l % g.xcd_nx; by = l / g.xcd_nx;
  }
  const int m0 = by * 64, n0 = bx * BN;
  const uint32_t* sA = reinterpret_cast<const uint32_t*>(g.sA) + (int64_t)min(m0 + wm * 32 + r, g.M - 1) * (g.lda / 128);
  const uint32_t* sB[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) sB[j] = reinterpret_cast<const uint32_t*>(g.sB) + (int64_t)min(n0 + (wn * TN + j) * 32 + r, g.N - 1) * (g.ldb / 128);
  uint4 ra[2], rb[2 * TN];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i, row = m0 + c / 8, kc = (c % 8) * 16;
      ra[i] = row < g.M ? *reinterpret_cast<const uint4*>(g.A + (int64_t)row * g.lda + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) {
      const int c = tid + 256 * i, row = n0 + c / 8, kc = (c % 8) * 16;
      rb[i] = row < g.N ? *reinterpret_cast<const uint4*>(g.B + (int64_t)row * g.ldb + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
  };
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = zero_acc();
  gload(0);
  for (int k0 = 0; k0 < g.K; k0 += MX_BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int c = tid + 256 * i; *reinterpret_cast<uint4*>(As + (c / 8) * MX_PITCH + (c % 8) * 16) = ra[i]; }
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) { const int c = tid + 256 * i; *reinterpret_cast<uint4*>(Bs + (c / 8) * MX_PITCH + (c % 8) * 16) = rb[i]; }
    const uint32_t sa4 = sA[k0 / 128];
    uint32_t sb4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) sb4[j] = sB[j][k0 / 128];
    __syncthreads();
    if (k0 + MX_BK < g.K) gload(k0 + MX_BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint8_t* ap = As + (wm * 32 + r) * MX_PITCH + s * 64 + h * 16;
      const uint4 a0 = *reinterpret_cast<const uint4*>(ap), a1 = *reinterpret_cast<const uint4*>(ap + 32);
      const v8i_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
      const int sa = (int)((sa4 >> (8 * (2 * s + h))) & 0xFF);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint8_t* bp = Bs + ((wn * TN + j) * 32 + r) * MX_PITCH + s * 64 + h * 16;
        const uint4 b0 = *reinterpret_cast<const uint4*>(bp), b1 = *reinterpret_cast<const uint4*>(bp + 32);
        const v8i_t bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
        const int sb = (int)((sb4[j] >> (8 * (2 * s + h))) & 0xFF);
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[j], FA, 0, 0, sa, 0, sb);
      }
    }
  }
  const uint64_t dkey = (EPI == TG_RESID_F32 && g.drop_thresh) ? splitmix64(*g.drop_step + g.drop_key) : 0ull;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + r;
    if (col >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m0 + wm * 32 + acc_row(i, lane);
      if (row >= g.M) continue;
      const int64_t at = (int64_t)row * g.ldc + col;
      const float v = acc[j][i];
      if constexpr (EPI == TG_STORE_T) reinterpret_cast<bf16_t*>(g.C)[at] = f32_to_bf16(v);
      else if constexpr (EPI == TG_STORE_F32) reinterpret_cast<float*>(g.C)[at] = v;
      else if constexpr (EPI == TG_ACC_F32) reinterpret_cast<float*>(g.C)[at] += v;
      else {
        float u = v;
        if (g.drop_thresh) u = drop_keep(dkey, at, g.drop_thresh) ? v * g.drop_scale : 0.f;
        reinterpret_cast<float*>(g.C)[at] = g.R[at] + u;
      }
    }
  }
}
static dim3 mx_xcd_grid(MxGemmArgs& g, dim3 grid);
template <int FA, int TN>
static int launch_mxgemm_p_ft(int epi, const MxGemmArgs& g_in, hipStream_t st) {
  MxGemmArgs g = g_in;
  const dim3 grid = mx_xcd_grid(g, dim3((unsigned)ceil_div(g.N, 64 * TN), (unsigned)ceil_div(g.M, 64)));
  switch (epi) {
    case TG_STORE_T: hipLaunchKernelGGL((mxgemm_p_kernel<FA, TG_STORE_T, TN>), grid, dim3(256), 0, st, g); break;
    case TG_STORE_F32: hipLaunchKernelGGL((mxgemm_p_kernel<FA, TG_STORE_F32, TN>), grid, dim3(256), 0, st, g); break;
    case TG_ACC_F32: hipLaunchKernelGGL((mxgemm_p_kernel<FA, TG_ACC_F32, TN>), grid, dim3(256), 0, st, g); break;
    case TG_RESID_F32: hipLaunchKernelGGL((mxgemm_p_kernel<FA, TG_RESID_F32, TN>), grid, dim3(256), 0, st, g); break;
    default: set_error("mxgemm_p: bad epilogue %d", epi); return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

static bool mx_xcd_order() { static const bool on = [] { const char* v = getenv("M2M_XCD_ORDER"); return !(v && v[0] == '0'); }(); return on; }
static dim3 mx_xcd_grid(MxGemmArgs& g, dim3 grid) {
  if (!mx_xcd_order() || grid.x * grid.y < 64) return grid;
  g.xcd_nx = (int)grid.x; g.xcd_total = (int)(grid.x * grid.y);
  return dim3((unsigned)(8 * ceil_div(g.xcd_total, 8)));
}
template <int FA, int TN>
static int launch_mxgemm_q_ft(int epi, const MxGemmArgs& g_in, hipStream_t st) {
  MxGemmArgs g = g_in;
  const dim3 grid = mx_xcd_grid(g, dim3((unsigned)ceil_div(g.N, 64 * TN), (unsigned)ceil_div(g.M, 64)));
  switch (epi) {
    case TG_STORE_T: hipLaunchKernelGGL((mxgemm_q_kernel<FA, TG_STORE_T, TN>), grid, dim3(256), 0, st, g); break;
    case TG_STORE_F32: hipLaunchKernelGGL((mxgemm_q_kernel<FA, TG_STORE_F32, TN>), grid, dim3(256), 0, st, g); break;
    case TG_ACC_F32: hipLaunchKernelGGL((mxgemm_q_kernel<FA, TG_ACC_F32, TN>), grid, dim3(256), 0, st, g); break;
    case TG_RESID_F32: hipLaunchKernelGGL((mxgemm_q_kernel<FA, TG_RESID_F32, TN>), grid, dim3(256), 0, st, g); break;
    default: set_error("mxgemm_q: bad epilogue %d", epi); return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}
template <int FA>
static int launch_mxgemm_q_f(int epi, const MxGemmArgs& g, hipStream_t st) {
  static const int wide_from = [] { const char* v = getenv("M2M_MXQ_WIDE_FROM"); return v ? atoi(v) : 512; }();      // N from which the 64 x 128 tile is used
  return g.N >= wide_from ? launch_mxgemm_q_ft<FA, 2>(epi, g, st) : launch_mxgemm_q_ft<FA, 1>(epi, g, st);
}
int launch_mxgemm_q(int fmt_a, int epi, const MxGemmArgs& g, hipStream_t st) {
  M2M_REQUIRE(g.Asrc && g.M >= 1 && g.N >= 1 && g.K >= 128 && g.K % 128 == 0 && g.ldb % 128 == 0 && g.K <= g.ldb && g.Kvalid <= g.K && g.ld_src % 8 == 0 &&
                  g.Kvalid % 8 == 0 && (reinterpret_cast<uintptr_t>(g.Asrc) & 15) == 0,
              "mxgemm_q: K=%d (valid %d), ld_src=%lld, ldb=%lld", g.K, g.Kvalid, (long long)g.ld_src, (long long)g.ldb);
  M2M_REQUIRE(fmt_a == 0 || fmt_a == 1, "mxgemm_q: A format e4m3 | e5m2");
  return fmt_a == 0 ? launch_mxgemm_q_f<0>(epi, g, st) : launch_mxgemm_q_f<1>(epi, g, st);
}

__global__ void mx_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int M, int N, int64_t ldc, int ksplit) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float acc = 0.f;
    for (int z = 0; z < ksplit; ++z) acc += part[(int64_t)z * n + i];
    const int64_t row = i / N;
    C[row * ldc + (i - row * N)] = acc;
  }
}

template <int FA, int FB>
static int launch_mxgemm_f(int epi, const MxGemmArgs& g, hipStream_t st) {
  // (a 128x128-tile variant with register prefetch was built and measured: 34.1 vs 29.8 ms per 64-clip step — slower; removed)
  dim3 grid((unsigned)ceil_div(g.N, 64), (unsigned)ceil_div(g.M, 64), (unsigned)(g.ksplit > 1 ? g.ksplit : 1));
  switch (epi) {
    case TG_STORE_T: hipLaunchKernelGGL((mxgemm_kernel<FA, FB, TG_STORE_T>), grid, dim3(256), 0, st, g); break;
    case TG_STORE_F32: hipLaunchKernelGGL((mxgemm_kernel<FA, FB, TG_STORE_F32>), grid, dim3(256), 0, st, g); break;
    case TG_ACC_F32: hipLaunchKernelGGL((mxgemm_kernel<FA, FB, TG_ACC_F32>), grid, dim3(256), 0, st, g); break;
    case TG_RESID_F32: hipLaunchKernelGGL((mxgemm_kernel<FA, FB, TG_RESID_F32>), grid, dim3(256), 0, st, g); break;
    default: set_error("mxgemm: bad epilogue %d", epi); return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

int launch_mxgemm(int fmt_a, int fmt_b, int epi, const MxGemmArgs& g, hipStream_t st) {
  M2M_REQUIRE(g.M >= 1 && g.N >= 1 && g.K >= 1 && g.K % 128 == 0 && g.lda % 128 == 0 && g.ldb % 128 == 0 && g.K <= g.lda && g.K <= g.ldb,
              "mxgemm: K=%d, lda=%lld, ldb=%lld must be multiples of 128 with K <= ld", g.K, (long long)g.lda, (long long)g.ldb);
  M2M_REQUIRE(fmt_b == 0 && (fmt_a == 0 || fmt_a == 1), "mxgemm: formats (A e4m3|e5m2, B e4m3) only");
  int rc;
  if (g.ksplit > 1) {
    M2M_REQUIRE(epi == TG_STORE_F32 && g.Cpart && g.kchunk % 128 == 0, "mxgemm: split-K is for plain fp32-store products");
  }
  static const bool plain = getenv("M2M_MXGEMM_PLAIN") != nullptr;      // diagnostic: the unprefetched 64x64 kernel
  static const int wide_from = [] { const char* v = getenv("M2M_MXP_WIDE_FROM"); return v ? atoi(v) : 512; }();
  if (g.ksplit <= 1 && !plain) {
    if (g.N >= wide_from) return fmt_a == 0 ? launch_mxgemm_p_ft<0, 2>(epi, g, st) : launch_mxgemm_p_ft<1, 2>(epi, g, st);
    return fmt_a == 0 ? launch_mxgemm_p_ft<0, 1>(epi, g, st) : launch_mxgemm_p_ft<1, 1>(epi, g, st);
  }
  rc = fmt_a == 0 ? launch_mxgemm_f<0, 0>(epi, g, st) : launch_mxgemm_f<1, 0>(epi, g, st);
  if (rc != M2M_OK) return rc;
  if (g.ksplit > 1) {
    const int64_t n = (int64_t)g.M * g.N;
    hipLaunchKernelGGL(mx_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st, g.Cpart,
                       reinterpret_cast<float*>(g.C), g.M, g.N, g.ldc, g.ksplit);
    M2M_CHECK_HIP(hipGetLastError());
  }
  return M2M_OK;
}

template <typename TS>
static int launch_mxq_rows_t(const TS* src, int64_t ld_s, uint8_t* q, uint8_t* sc, int R, int C, int Cp, int fmt, hipStream_t st) {
  const int64_t n = (int64_t)R * (Cp / 32);
  const unsigned grid = (unsigned)((n + 127) / 128 > 4096 ? 4096 : (n + 127) / 128);
  if (fmt == 0) hipLaunchKernelGGL((mxq_rows_kernel<TS, 0>), dim3(grid), dim3(128), 0, st, src, ld_s, q, sc, R, C, Cp);
  else hipLaunchKernelGGL((mxq_rows_kernel<TS, 1>), dim3(grid), dim3(128), 0, st, src, ld_s, q, sc, R, C, Cp);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}
template <typename TS>
static int launch_mxq_cols_t(const TS* src, int64_t ld_s, uint8_t* qt, uint8_t* sc, int R, int C, int Rp, int fmt, hipStream_t st) {
  // rows [R, Rp) of every column are zero blocks: the grid covers Rp / 32 row blocks
  dim3 grid((unsigned)ceil_div(C, 64), (unsigned)(Rp / 32));
  if (fmt == 0) hipLaunchKernelGGL((mxq_cols_kernel<TS, 0>), grid, dim3(64), 0, st, src, ld_s, qt, sc, R, C, Rp);
  else hipLaunchKernelGGL((mxq_cols_kernel<TS, 1>), grid, dim3(64), 0, st, src, ld_s, qt, sc, R, C, Rp);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}
// src_kind: 0 = fp32, 1 = bf16
int launch_mxq_rows(int src_kind, const void* src, int64_t ld_s, uint8_t* q, uint8_t* sc, int R, int C, int Cp, int fmt, hipStream_t st) {
  M2M_REQUIRE(Cp % 128 == 0 && Cp >= C, "mxq_rows: padded width %d must be a multiple of 128 and >= %d", Cp, C);
  return src_kind == 0 ? launch_mxq_rows_t<float>((const float*)src, ld_s, q, sc, R, C, Cp, fmt, st)
                       : launch_mxq_rows_t<bf16_t>((const bf16_t*)src, ld_s, q, sc, R, C, Cp, fmt, st);
}
int launch_mxq_cols(int src_kind, const void* src, int64_t ld_s, uint8_t* qt, uint8_t* sc, int R, int C, int Rp, int fmt, hipStream_t st) {
  M2M_REQUIRE(Rp % 128 == 0 && Rp >= R, "mxq_cols: padded height %d must be a multiple of 128 and >= %d", Rp, R);
  return src_kind == 0 ? launch_mxq_cols_t<float>((const float*)src, ld_s, qt, sc, R, C, Rp, fmt, st)
                       : launch_mxq_cols_t<bf16_t>((const bf16_t*)src, ld_s, qt, sc, R, C, Rp, fmt, st);
}

}  // namespace m2m

// ------------------------------------------------------------------ C ABI: one MXFP8 product, for tests and callers ---
using namespace m2m;

extern "C" int m2m_mx8_matmul_f32(const float* a_dev, const float* b_dev, int M, int N, int K, int a_is_e5m2, float* c_dev, void* stream) {
  M2M_REQUIRE(a_dev && b_dev && c_dev && M >= 1 && N >= 1 && K >= 1, "m2m_mx8_matmul_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int Kp = (int)align_up(K, 128);
  uint8_t* buf = nullptr;
  const size_t qa = (size_t)M * Kp, qb = (size_t)N * Kp, sa = (size_t)M * (Kp / 32), sb = (size_t)N * (Kp / 32);
  M2M_CHECK_HIP(hipMalloc((void**)&buf, qa + qb + align_up((int64_t)sa, 256) + align_up((int64_t)sb, 256) + 1024));
  uint8_t *A8 = buf, *B8 = buf + qa, *SA = B8 + qb, *SB = SA + align_up((int64_t)sa, 256);
  int rc = launch_mxq_rows(0, a_dev, K, A8, SA, M, K, Kp, a_is_e5m2 ? 1 : 0, st);
  if (rc == M2M_OK) rc = launch_mxq_rows(0, b_dev, K, B8, SB, N, K, Kp, 0, st);
  MxGemmArgs g{};
  g.A = A8; g.B = B8; g.sA = SA; g.sB = SB; g.C = c_dev; g.M = M; g.N = N; g.K = Kp; g.lda = Kp; g.ldb = Kp; g.ldc = N;
  if (rc == M2M_OK) rc = launch_mxgemm(a_is_e5m2 ? 1 : 0, 0, TG_STORE_F32, g, st);
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  if (rc == M2M_OK && e != hipSuccess) { set_error("m2m_mx8_matmul_f32: %s", hipGetErrorString(e)); rc = M2M_ERR_HIP; }
  return rc;
}

// The same product with A given in bf16 (what the training path feeds it): `fused` != 0 quantises A inside the product's operand
// staging (mxgemm_q_kernel), 0 through the separate row quantiser — the two must agree bit for bit (tests/test_mx8_gpu.py).
extern "C" int m2m_mx8_matmul_bf16a(const uint16_t* a_bf16_dev, const float* b_dev, int M, int N, int K, int a_is_e5m2, int fused, float* c_dev,
                                    void* stream) {
  M2M_REQUIRE(a_bf16_dev && b_dev && c_dev && M >= 1 && N >= 1 && K >= 8 && K % 8 == 0, "m2m_mx8_matmul_bf16a: bad argument (K must be a multiple of 8)");
  hipStream_t st = (hipStream_t)stream;
  const int Kp = (int)align_up(K, 128);
  uint8_t* buf = nullptr;
  const size_t qa = (size_t)M * Kp, qb = (size_t)N * Kp, sa = (size_t)M * (Kp / 32), sb = (size_t)N * (Kp / 32);
  M2M_CHECK_HIP(hipMalloc((void**)&buf, qa + qb + align_up((int64_t)sa, 256) + align_up((int64_t)sb, 256) + 1024));
  uint8_t *A8 = buf, *B8 = buf + qa, *SA = B8 + qb, *SB = SA + align_up((int64_t)sa, 256);
  int rc = launch_mxq_rows(0, b_dev, K, B8, SB, N, K, Kp, 0, st);
  MxGemmArgs g{};
  g.A = A8; g.B = B8; g.sA = SA; g.sB = SB; g.C = c_dev; g.M = M; g.N = N; g.K = Kp; g.lda = Kp; g.ldb = Kp; g.ldc = N;
  if (rc == M2M_OK) {
    if (fused) {
      g.Asrc = a_bf16_dev; g.ld_src = K; g.Kvalid = K;
      rc = launch_mxgemm_q(a_is_e5m2 ? 1 : 0, TG_STORE_F32, g, st);
    } else {
      rc = launch_mxq_rows(1, a_bf16_dev, K, A8, SA, M, K, Kp, a_is_e5m2 ? 1 : 0, st);
      if (rc == M2M_OK) rc = launch_mxgemm(a_is_e5m2 ? 1 : 0, 0, TG_STORE_F32, g, st);
    }
  }
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  if (rc == M2M_OK && e != hipSuccess) { set_error("m2m_mx8_matmul_bf16a: %s", hipGetErrorString(e)); rc = M2M_ERR_HIP; }
  return rc;
}
