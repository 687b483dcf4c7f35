// Training-step kernels: what `loss.backward()` + `Adafactor.step()` run for
// ref: music2midi/model.py:27-43 (training_step -> T5Transformer.forward with labels, ref transformer.py:28-39;
// optimizer = transformers Adafactor(warmup_init=True) + AdafactorSchedule), written for gfx950.
//
// Layout rules of the training path (train_api.hip drives these kernels):
//   * every activation is a plain row-major [rows, features] matrix — fp32 for the residual stream, its
//     gradient and the logits, storage type T (bf16, or fp32 in the parity mode) for every GEMM input;
//   * attention is done with MATERIALISED probabilities: training sequences are short (S = 190..261 encoder
//     positions, <= ~360 labels), so P [B,H,Sq,Sk] is a few MB per layer and the backward is four batched
//     GEMMs + one row kernel instead of a second flash kernel;
//   * ONE strided/batched MFMA GEMM (`bgemm_kernel`) serves every product of the forward and the backward:
//     either operand may be stored k-major ("transposed"), so dX = dY . W and dW = dY^T . X read the same
//     buffers the forward wrote — no transposed copies of weights or activations exist;
//   * every reduction has a fixed order (per-block partials + a second pass, never float atomics), so a
//     training step is bit-reproducible run to run.
#include "mma.h"
#include "t5.h"
#include "train.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <string>
#include <utility>
#include <vector>

namespace m2m {

// ============================================================ strided batched GEMM ====
// C[z][M,N] (op)= alpha * A[z][M,K] . B[z][N,K]^T,  z = (b1, b2) with independent strides per operand.
// 64x64 output tile per 256-thread workgroup (2x2 waves, one 32x32 MFMA tile each), BK = 32.
// Operand storage: a_kmajor == 0: element (m, k) at A[m*lda + k]; a_kmajor == 1: at A[k*lda + m]
// (the transposed product reads the buffer as it is: the tile is transposed on its way into LDS).
constexpr int TG_BM = 64, TG_BN = 64;
// k extent staged per step: 96 (bf16) / 64 (fp32).  The attention products reduce over <= ~360 keys / queries, so the dependent
// load -> LDS -> MFMA chain of a workgroup is 3 steps instead of the 9 of BK = 32 (16 -> us per launch).  96 rather than 128 for
// bf16 (round 3): 26.6 KB of LDS instead of 34.8 — five workgroups per CU (97 registers allow them) instead of four, so the
// 1 280 workgroups of the paired dV | dK launch of a 16-clip step are ONE round of the chip, and 261 keys are three steps either way
// (18 launches per step, 17.8 -> 16.3 us each; M2M_TG_BK at build time for measurements).
constexpr int TG_BK_MAX = 128;

template <typename T> struct TgCfg;
#ifndef M2M_TG_BK
#define M2M_TG_BK 96
#endif
template <> struct TgCfg<bf16_t> { static constexpr int E = 8, BK = M2M_TG_BK, PITCH = BK + 8; };
template <> struct TgCfg<float> { static constexpr int E = 4, BK = 64, PITCH = BK + 4; };

__device__ inline uint32_t u4_get(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
// Column j of an E x E block held as E 16-byte rows (E = 8 bf16 / 4 fp32), as one 16-byte row of the transposed block.
// bf16: halves (2d, 2d+1) of dword j/2 through v_perm_b32; fp32: register renaming.
template <typename T, int E>
__device__ inline uint4 tr_col(const uint4 (&rg)[E], int j) {
  if constexpr (sizeof(T) == 2) {
    const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
    return make_uint4(__builtin_amdgcn_perm(u4_get(rg[1], j >> 1), u4_get(rg[0], j >> 1), sel),
                      __builtin_amdgcn_perm(u4_get(rg[3], j >> 1), u4_get(rg[2], j >> 1), sel),
                      __builtin_amdgcn_perm(u4_get(rg[5], j >> 1), u4_get(rg[4], j >> 1), sel),
                      __builtin_amdgcn_perm(u4_get(rg[7], j >> 1), u4_get(rg[6], j >> 1), sel));
  } else {
    return make_uint4(u4_get(rg[0], j), u4_get(rg[1], j), u4_get(rg[2], j), u4_get(rg[3], j));
  }
}

// one operand tile [64 rows][32 k] -> LDS (row-major, k contiguous), zero-filled outside (rows_valid, k_valid).
// k-major operands: E x E blocks transposed in registers (16-byte LDS rows instead of 2-byte scatters), threads
// [tshift, tshift + blocks) do the work so that the A and the B tile of a step are staged by different waves; the row
// (non-reduction) index may run past `rows` inside the 16-byte chunk: those tile rows only feed outputs that are never stored
// (launch_bgemm checks that the padded row still lies inside the operand's row stride).
template <typename T>
__device__ inline void tg_stage(T* __restrict__ S, const T* __restrict__ G, int64_t ld, int kmajor, int row0, int k0,
                                int rows, int K, int tid, int tshift) {
  using Cfg = TgCfg<T>;
  constexpr int E = Cfg::E;
  if (!kmajor) {
    constexpr int TG_BK = Cfg::BK;
    constexpr int CPR = TG_BK / E;                      // chunks per row
    for (int c = tid; c < TG_BM * CPR; c += 256) {
      const int rl = c / CPR, kc = (c % CPR) * E;
      const int row = row0 + rl, k = k0 + kc;
      T* dst = S + rl * Cfg::PITCH + kc;
      if (row < rows && k + E <= K) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(G + (int64_t)row * ld + k);
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = (row < rows && k + e < K) ? G[(int64_t)row * ld + k + e] : from_f32<T>(0.f);
      }
    }
  } else {
    constexpr int KBn = Cfg::BK / E, RBn = TG_BM / E;   // blocks along k / along the tile rows
    const int c0 = tid - tshift;                                             // 128 threads per operand
    for (int c = c0; c0 >= 0 && c0 < 128 && c < KBn * RBn; c += 128) {
      const int kb = c % KBn, rb = c / KBn;
      const int row = row0 + rb * E;
      uint4 rg[E];
#pragma unroll
      for (int kk = 0; kk < E; ++kk) {
        const int k = k0 + kb * E + kk;
        rg[kk] = (k < K && row < rows) ? *reinterpret_cast<const uint4*>(G + (int64_t)k * ld + row) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < E; ++j) *reinterpret_cast<uint4*>(S + (rb * E + j) * Cfg::PITCH + kb * E) = tr_col<T, E>(rg, j);
    }
  }
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void bgemm_kernel(BGemmArgs g) {
  using Cfg = TgCfg<T>;
  __shared__ __align__(16) T As[TG_BM * Cfg::PITCH];
  __shared__ __align__(16) T Bs[TG_BN * Cfg::PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const bool split = g.ksplit > 1;
  const int nz = g.nb1 * g.nb2;
  // XCD-contiguous workgroup order (g.xcd_total > 0: 1-D launch; workgroup id % 8 picks the XCD): an XCD walks one eighth of the
  // (x, y, z) list front to back, so the row tiles of one (clip, head) product — which all read the same k-major operand — run
  // side by side on ONE XCD and meet in its L2 (dV / dK: 124 MB fetched per launch for 33 MB of operands with the launch order)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.xcd_total > 0) {
    const int per = (g.xcd_total + 7) >> 3, l = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (l >= g.xcd_total || (int)(blockIdx.x >> 3) >= per) return;        // padding workgroups (uniform)
    bx = l % g.xcd_nx; by = (l / g.xcd_nx) % g.xcd_ny; bz = l / (g.xcd_nx * g.xcd_ny);
  }
  const bool second = !split && g.A2 && bz >= nz;            // pair mode: the second product of the launch
  const int zz = second ? bz - nz : bz;
  const int z1 = split ? 0 : zz / g.nb2, z2 = split ? 0 : zz - z1 * g.nb2;
  const T* A = reinterpret_cast<const T*>(second ? g.A2 : g.A) + z1 * g.sA1 + z2 * g.sA2;
  const T* B = second ? reinterpret_cast<const T*>(g.B2) + z1 * g.sB1_2 + z2 * g.sB2_2 : reinterpret_cast<const T*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  const int64_t ldb = second ? g.ldb2 : g.ldb;
  void* const Cout = second ? g.C2 : g.C;
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2;
  const int m0 = by * TG_BM, n0 = bx * TG_BN;
  const int kbeg = split ? bz * g.kchunk : 0, kend = split ? min(g.K, kbeg + g.kchunk) : g.K;

  f32x16 acc = zero_acc();
  constexpr int TG_BK = Cfg::BK;
  for (int k0 = kbeg; k0 < kend; k0 += TG_BK) {
    __syncthreads();
    tg_stage<T>(As, A, g.lda, g.a_kmajor, m0, k0, g.M, kend, tid, 0);
    tg_stage<T>(Bs, B, ldb, g.b_kmajor, n0, k0, g.N, kend, tid, 128);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TG_BK / 16; ++s) {
      const Frag<T> fa = load_frag(As + (wm * 32 + r) * Cfg::PITCH + s * 16 + 8 * h);
      const Frag<T> fb = load_frag(Bs + (wn * 32 + r) * Cfg::PITCH + s * 16 + 8 * h);
      mma16(acc, fa, fb);
    }
  }
  const int col = n0 + wn * 32 + r;
  if (col >= g.N) return;
  const uint64_t dkey = (EPI == TG_RESID_F32 && g.drop_thresh) ? splitmix64(*g.drop_step + g.drop_key) : 0ull;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + wm * 32 + acc_row(i, lane);
    if (row >= g.M) continue;
    if (split) {                                   // partial tile of this k-slice
      g.Cpart[((int64_t)bz * g.M + row) * g.N + col] = acc[i];
      continue;
    }
    const int64_t at = coff + (int64_t)row * g.ldc + col;
    const float v = g.alpha * acc[i];
    if constexpr (EPI == TG_STORE_T) reinterpret_cast<T*>(Cout)[at] = from_f32<T>(v);
    else if constexpr (EPI == TG_STORE_F32) reinterpret_cast<float*>(Cout)[at] = v;
    else if constexpr (EPI == TG_ACC_F32) reinterpret_cast<float*>(Cout)[at] += v;
    else {                                                                       // TG_RESID_F32: C = R + dropout(acc)
      float u = v;
      if (g.drop_thresh) u = drop_keep(dkey, at, g.drop_thresh) ? v * g.drop_scale : 0.f;
      reinterpret_cast<float*>(Cout)[at] = g.R[at] + u;
    }
  }
}

template <typename T>
static int launch_bgemm_t(int epi, const BGemmArgs& g_in, hipStream_t st) {
  BGemmArgs g = g_in;
  dim3 grid((unsigned)ceil_div(g.N, TG_BN), (unsigned)ceil_div(g.M, TG_BM), (unsigned)(g.ksplit > 1 ? g.ksplit : g.nb1 * g.nb2 * (g.A2 ? 2 : 1)));
  static const bool xcd = [] { const char* v = getenv("M2M_XCD_ORDER"); return !(v && v[0] == '0'); }();
  if (xcd && (int64_t)grid.x * grid.y * grid.z >= 64 && (int64_t)grid.x * grid.y * grid.z < (1 << 30)) {
    g.xcd_nx = (int)grid.x; g.xcd_ny = (int)grid.y; g.xcd_total = (int)(grid.x * grid.y * grid.z);
    grid = dim3((unsigned)(8 * ceil_div(g.xcd_total, 8)));
  }
  switch (epi) {
    case TG_STORE_T: hipLaunchKernelGGL((bgemm_kernel<T, TG_STORE_T>), grid, dim3(256), 0, st, g); break;
    case TG_STORE_F32: hipLaunchKernelGGL((bgemm_kernel<T, TG_STORE_F32>), grid, dim3(256), 0, st, g); break;
    case TG_ACC_F32: hipLaunchKernelGGL((bgemm_kernel<T, TG_ACC_F32>), grid, dim3(256), 0, st, g); break;
    case TG_RESID_F32: hipLaunchKernelGGL((bgemm_kernel<T, TG_RESID_F32>), grid, dim3(256), 0, st, g); break;
    default: set_error("bgemm: bad epilogue %d", epi); return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// C[row][col] = alpha * sum_z part[z][row][col]   (z ascending: fixed order)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int M, int N, int64_t ldc, int ksplit, float alpha) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float acc = 0.f;
    for (int z = 0; z < ksplit; ++z) acc += part[(int64_t)z * n + i];
    const int64_t row = i / N;
    C[row * ldc + (i - row * N)] = alpha * acc;
  }
}

// ============================================================ weight-gradient GEMM ====
// C[N1, N2] (fp32) = A[K, N1]^T . B[K, N2]: both operands row-major with the REDUCTION index as the row (a weight
// gradient dW = dY^T . X reads dY [rows, N1] and X [rows, N2] exactly as the forward / backward passes left them).
// The MFMA wants 8 consecutive k per lane, memory has 8 consecutive n: every thread loads an E x E block (E rows of k,
// 16 bytes = E elements of n each; a wave's request is eight full 128-byte lines), transposes it IN REGISTERS (bf16:
// 32 v_perm_b32 per 8x8 block; fp32: pure renaming) and writes E 16-byte rows of the usual [n][k] LDS tile — the same
// number of LDS writes as a k-contiguous operand, where the generic kernel above scatters 2-byte elements.
// Tiles 128x128 (TF = 2) or 64x64 (TF = 1), BK = 64 / 32, next k-step prefetched into registers, split over k on
// blockIdx.z with fp32 partial tiles summed in z order by splitk_reduce_kernel (deterministic).
struct DwGemmArgs {
  const void *A, *B;
  float* C;
  float* Cpart;
  int N1, N2, K;
  int64_t lda, ldb, ldc;
  int ksplit, kchunk;
};

template <typename T> struct DwCfg;
template <> struct DwCfg<bf16_t> { static constexpr int BK = 64, E = 8; };
template <> struct DwCfg<float> { static constexpr int BK = 32, E = 4; };

// one output tile [n1_0, +BT) x [n2_0, +BT) over k in [kbeg, kend), written to out (row stride ldo)
template <typename T, int TF>
__device__ inline void dw_tile(const DwGemmArgs& g, T* __restrict__ AB, int n1_0, int n2_0, int kbeg, int kend, float* __restrict__ out, int64_t ldo) {
  using Cfg = DwCfg<T>;
  constexpr int BK = Cfg::BK, E = Cfg::E, PITCH = BK + E;
  constexpr int BT = 64 * TF, WT = 32 * TF;
  constexpr int KB = BK / E, NB = BT / E, BPO = KB * NB, TB = 2 * BPO, NBT = (TB + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // this thread's blocks: (operand, n block, k block); kb runs fastest over the lanes, so one store instruction of a wave
  // covers whole 128-byte LDS rows and one load instruction whole 128-byte lines of 8 k-rows
  const T* src[NBT];
  int64_t ld[NBT];
  int kofs[NBT], lds_off[NBT];
  bool ok[NBT];
#pragma unroll
  for (int u = 0; u < NBT; ++u) {
    const int bi = tid + 256 * u;
    const int op = bi / BPO, id = bi - op * BPO;
    const int kb = id % KB, nb = id / KB;
    const int n0 = op ? n2_0 : n1_0, nlim = op ? g.N2 : g.N1;
    ok[u] = bi < TB && n0 + nb * E + E <= nlim;
    src[u] = reinterpret_cast<const T*>(op ? g.B : g.A) + (ok[u] ? n0 + nb * E : 0);
    ld[u] = op ? g.ldb : g.lda;
    kofs[u] = kb * E;
    lds_off[u] = (min(op, 1) * BT + nb * E) * PITCH + kb * E;
  }
  uint4 rg[NBT][E];
  auto gload = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NBT; ++u)
#pragma unroll
      for (int kk = 0; kk < E; ++kk) {
        const int k = k0 + kofs[u] + kk;
        const uint4 v = *reinterpret_cast<const uint4*>(src[u] + (int64_t)min(k, kend - 1) * ld[u]);   // clamped, never predicated
        rg[u][kk] = (ok[u] && k < kend) ? v : make_uint4(0, 0, 0, 0);
      }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int u = 0; u < NBT; ++u) {
      if (TB < 256 && tid >= TB) continue;
#pragma unroll
      for (int j = 0; j < E; ++j) *reinterpret_cast<uint4*>(AB + lds_off[u] + j * PITCH) = tr_col<T, E>(rg[u], j);
    }
  };

  f32x16 acc[TF][TF];
#pragma unroll
  for (int i = 0; i < TF; ++i)
#pragma unroll
    for (int j = 0; j < TF; ++j) acc[i][j] = zero_acc();
  const T* As = AB;
  const T* Bs = AB + BT * PITCH;
  gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      Frag<T> fa[TF], fb[TF];
#pragma unroll
      for (int i = 0; i < TF; ++i) {
        fa[i] = load_frag(As + (wm * WT + i * 32 + r) * PITCH + s * 16 + 8 * h);
        fb[i] = load_frag(Bs + (wn * WT + i * 32 + r) * PITCH + s * 16 + 8 * h);
      }
#pragma unroll
      for (int i = 0; i < TF; ++i)
#pragma unroll
        for (int j = 0; j < TF; ++j) mma16(acc[i][j], fa[i], fb[j]);
    }
  }
#pragma unroll
  for (int mi = 0; mi < TF; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = n1_0 + wm * WT + mi * 32 + acc_row(e, lane);
      if (row >= g.N1) continue;
#pragma unroll
      for (int ni = 0; ni < TF; ++ni) {
        const int col = n2_0 + wn * WT + ni * 32 + r;
        if (col < g.N2) out[(int64_t)row * ldo + col] = acc[mi][ni][e];
      }
    }
}

// The same 128 x 128 tile for bf16 WITHOUT the register transposes: the operands' k-major rows are copied to LDS as they lie in
// memory ([k][n], 320-byte rows: four consecutive k-rows fall into four different 16-bank groups), and an MFMA fragment — 8
// consecutive k of one column — is two `ds_read_b64_tr_b16` (gfx950's transposing LDS read: per 16 lanes a 4 x 16 block comes back
// column-major; lane map checked by tools/trprobe.hip).  What this buys is on the LOAD side: with an 8 x 8 register block per
// thread, neighbouring lanes had to sit in different k-rows (so that the transposed block lands in one LDS row), and every
// 16-byte request was a cache access of its own — PMC: 2.5e8 L1 accesses per launch for 3.9 GB, 77 % of the kernel's cycles per
// CU; here 16 neighbouring lanes read one 256-byte row segment.  ~140 VALU instructions per k-step and wave disappear with it.
constexpr int DWT_PITCH = 160;                       // bf16 elements per staged k-row: 128 + 32 of padding
typedef short dw_v4s __attribute__((ext_vector_type(4)));
__device__ inline Frag<bf16_t> dw_tr_frag(const bf16_t* p) {            // p: this lane's address of the first 4 k-rows (see dw_tile_tr)
  typedef dw_v4s __attribute__((address_space(3))) * lds_v4s;
  const dw_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p));
  const dw_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 4 * DWT_PITCH));
  Frag<bf16_t> f;
  f.v = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
  return f;
}
// FULL: the tile lies inside the matrix on both sides (every tile of the projection weights; not the last tiles of lm_head):
// no column predicate anywhere.  The k loop runs whole 64-row steps from pointers that advance by a constant — no clamp, no
// select, no 64-bit multiply per load — and one guarded step takes the ragged end (M = 4 176 rows = 65 steps + 16 rows).
template <bool FULL>
__device__ inline void dw_tile_tr(const DwGemmArgs& g, bf16_t* __restrict__ AB, int n1_0, int n2_0, int kbeg, int kend, float* __restrict__ out,
                                  int64_t ldo) {
  constexpr int BK = 64, P = DWT_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // staging: threads 0..127 copy operand A (dY), 128..255 operand B (X); a thread owns 16-byte chunk `ch` of the k-rows
  // rw, rw + 8, ... rw + 56: 16 neighbouring lanes cover one 256-byte row segment
  const int op = tid >> 7, c = tid & 127, ch = c & 15, rw = c >> 4;
  const int n0 = op ? n2_0 : n1_0, nlim = op ? g.N2 : g.N1;
  const bool ok = FULL || n0 + ch * 8 + 8 <= nlim;
  const int64_t ld = op ? g.ldb : g.lda;
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(op ? g.B : g.A) + (ok ? n0 + ch * 8 : 0) + (int64_t)(kbeg + rw) * ld;   // row rw of the current step
  const int64_t step = (int64_t)BK * ld, ld8 = 8 * ld;
  bf16_t* dst = AB + op * (BK * P) + rw * P + ch * 8;
  uint4 r0, r1, r2, r3, r4, r5, r6, r7;               // named: arrays across the k loop end up in scratch (hipcc 7.2)
#define M2M_DW_LD(u) { const uint4 v = *reinterpret_cast<const uint4*>(pk + (u) * ld8); r##u = ok ? v : make_uint4(0, 0, 0, 0); }
#define M2M_DW_LD_TAIL(u, k0) { const int k = (k0) + rw + 8 * (u); const uint4 v = *reinterpret_cast<const uint4*>(pk + (int64_t)(min(k, kend - 1) - ((k0) + rw)) * ld); \
                                r##u = (ok && k < kend) ? v : make_uint4(0, 0, 0, 0); }
#define M2M_DW_GLOAD() { M2M_DW_LD(0) M2M_DW_LD(1) M2M_DW_LD(2) M2M_DW_LD(3) M2M_DW_LD(4) M2M_DW_LD(5) M2M_DW_LD(6) M2M_DW_LD(7) }
#define M2M_DW_GLOAD_TAIL(k0) { M2M_DW_LD_TAIL(0, k0) M2M_DW_LD_TAIL(1, k0) M2M_DW_LD_TAIL(2, k0) M2M_DW_LD_TAIL(3, k0) M2M_DW_LD_TAIL(4, k0) \
                                M2M_DW_LD_TAIL(5, k0) M2M_DW_LD_TAIL(6, k0) M2M_DW_LD_TAIL(7, k0) }
#define M2M_DW_ST(u) *reinterpret_cast<uint4*>(dst + 8 * (u) * P) = r##u;
#define M2M_DW_SSTORE() { M2M_DW_ST(0) M2M_DW_ST(1) M2M_DW_ST(2) M2M_DW_ST(3) M2M_DW_ST(4) M2M_DW_ST(5) M2M_DW_ST(6) M2M_DW_ST(7) }
  // fragment addresses: lane = 16 g + 4 q + p supplies row q of its group's 4 x 16 block, columns 4p .. 4p+3; groups 0 / 1 are the
  // column halves of k 0..7, groups 2 / 3 those of k 8..15 (h = g >> 1) of a 16-deep substep
  const int li = lane & 15, q = li >> 2, pp = li & 3, gq = lane >> 4, h = lane >> 5;
  const bf16_t* fa = AB + (8 * h + q) * P + wm * 64 + 16 * (gq & 1) + 4 * pp;
  const bf16_t* fb = AB + BK * P + (8 * h + q) * P + wn * 64 + 16 * (gq & 1) + 4 * pp;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero_acc();
  const int nfull = (kend - kbeg) / BK;               // whole steps; a ragged one may follow
  const bool ragged = kbeg + nfull * BK < kend;
  const int nsteps = nfull + (ragged ? 1 : 0);
  if (nfull > 0) M2M_DW_GLOAD() else M2M_DW_GLOAD_TAIL(kbeg)
  for (int it = 0; it < nsteps; ++it) {
    __syncthreads();
    M2M_DW_SSTORE()
    __syncthreads();
    pk += step;
    if (it + 1 < nfull) M2M_DW_GLOAD()
    else if (it + 1 < nsteps) M2M_DW_GLOAD_TAIL(kbeg + (it + 1) * BK)
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      Frag<bf16_t> a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = dw_tr_frag(fa + 16 * s * P + 32 * i);
        b[i] = dw_tr_frag(fb + 16 * s * P + 32 * i);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma16(acc[i][j], a[i], b[j]);
    }
  }
#undef M2M_DW_LD
#undef M2M_DW_LD_TAIL
#undef M2M_DW_GLOAD
#undef M2M_DW_GLOAD_TAIL
#undef M2M_DW_ST
#undef M2M_DW_SSTORE
  const int r = lane & 31;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = n1_0 + wm * 64 + mi * 32 + acc_row(e, lane);
      if (!FULL && row >= g.N1) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n2_0 + wn * 64 + ni * 32 + r;
        if (FULL || col < g.N2) out[(int64_t)row * ldo + col] = acc[mi][ni][e];
      }
    }
}

template <typename T, int TF>
__global__ __launch_bounds__(256) void dw_gemm_kernel(DwGemmArgs g) {
  constexpr int BT = 64 * TF, PITCH = DwCfg<T>::BK + DwCfg<T>::E;
  __shared__ __align__(16) T AB[2 * BT * PITCH];
  const bool split = g.ksplit > 1;
  const int kbeg = split ? blockIdx.z * g.kchunk : 0, kend = split ? min(g.K, kbeg + g.kchunk) : g.K;
  dw_tile<T, TF>(g, AB, blockIdx.y * BT, blockIdx.x * BT, kbeg, kend, split ? g.Cpart + (int64_t)blockIdx.z * g.N1 * g.N2 : g.C,
                 split ? (int64_t)g.N2 : g.ldc);
}

// Every weight gradient of a step in ONE launch: the backward pass only records (dY, X, dW) triples — each sub-layer
// keeps its dY operands in buffers of its own, 0.55 GB at 16 clips, nothing for a 288 GB part — and this kernel walks
// the table: workgroup -> (problem, 128x128 tile), full reduction length per tile.  ~1 850 tiles fill the 256 CUs
// seven times over, so no product is split over k: the split-K partial images (3.8 GB written and re-read per step
// for the 67 products, 134 launches) are gone.
struct DwProb {
  DwGemmArgs g;
  int tile0, tn2;          // first tile of this problem in the launch; tiles along N2
};
template <typename T, bool TR>
__global__ __launch_bounds__(256) void dw_group_kernel(const DwProb* __restrict__ probs, int n_probs, int n_tiles, int xcd_order) {
  constexpr int PITCH = DwCfg<T>::BK + DwCfg<T>::E;
  constexpr int LDS_ELEMS = TR ? 2 * 64 * DWT_PITCH : 2 * 128 * PITCH;
  __shared__ __align__(16) T AB[LDS_ELEMS];
  // XCD-contiguous tile order (workgroup id % 8 picks the XCD): each XCD walks one eighth of the tile list front to back, so the
  // tiles that run side by side on an XCD are neighbours in the list — the same product, the same dY column block (tiles of a
  // product are numbered along X fastest), a handful of X column blocks — and their operand slices meet in that XCD's L2
  const int per = (n_tiles + 7) >> 3;
  const int tile = xcd_order ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= n_tiles || (xcd_order && (int)(blockIdx.x >> 3) >= per)) return;       // padding workgroups (uniform)
  int p = 0;
  while (p + 1 < n_probs && tile >= probs[p + 1].tile0) ++p;                          // uniform scan of <= ~70 entries
  const DwProb pr = probs[p];
  const int tl = tile - pr.tile0;
  if constexpr (TR) {
    const int n1_0 = (tl / pr.tn2) * 128, n2_0 = (tl % pr.tn2) * 128;
    if (n1_0 + 128 <= pr.g.N1 && n2_0 + 128 <= pr.g.N2) dw_tile_tr<true>(pr.g, AB, n1_0, n2_0, 0, pr.g.K, pr.g.C, pr.g.ldc);      // (workgroup-uniform)
    else dw_tile_tr<false>(pr.g, AB, n1_0, n2_0, 0, pr.g.K, pr.g.C, pr.g.ldc);
  } else dw_tile<T, 2>(pr.g, AB, (tl / pr.tn2) * 128, (tl % pr.tn2) * 128, 0, pr.g.K, pr.g.C, pr.g.ldc);
}

// C = A^T . B with the split chosen here: enough workgroups to fill the chip, as few k-slices as that allows (every slice
// writes and re-reads an fp32 image of C).
int launch_dw_gemm(int precision, const void* A, int64_t lda, int N1, const void* B, int64_t ldb, int N2, int K, float* C, int64_t ldc,
                   float* kpart, int64_t kpart_floats, hipStream_t st) {
  const int E = precision == M2M_PREC_BF16 ? 8 : 4, BK = precision == M2M_PREC_BF16 ? 64 : 32;
  M2M_REQUIRE(N1 % E == 0 && N2 % E == 0 && lda % E == 0 && ldb % E == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(B) & 15) == 0,
              "dw_gemm: operands must be 16-byte aligned with row lengths that are multiples of %d", E);
  static const int force_tf = [] { const char* v = getenv("M2M_DW_TILE"); return v ? atoi(v) : 0; }();      // 64 / 128: diagnostic
  static const int target = [] { const char* v = getenv("M2M_DW_WGS"); return v ? atoi(v) : 512; }();
  const int t128 = ceil_div(N1, 128) * ceil_div(N2, 128);
  const int tf = force_tf == 64 ? 1 : 2;   // 128x128 measured faster for every weight shape of the model (16 clips: 8.80 vs 9.90 ms per step)
  (void)t128;
  const int bt = 64 * tf;
  const int tiles = ceil_div(N1, bt) * ceil_div(N2, bt);
  int ks = ceil_div(target, tiles);
  if (ks > 32) ks = 32;
  while (ks > 1 && ((int64_t)ks * N1 * N2 > kpart_floats || K / ks < 2 * BK)) --ks;
  DwGemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.Cpart = kpart; g.N1 = N1; g.N2 = N2; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.ksplit = 1; g.kchunk = K;
  if (ks > 1) { g.kchunk = (int)align_up(ceil_div(K, ks), BK); g.ksplit = ceil_div(K, g.kchunk); }
  dim3 grid((unsigned)ceil_div(N2, bt), (unsigned)ceil_div(N1, bt), (unsigned)g.ksplit);
  if (precision == M2M_PREC_BF16) {
    if (tf == 2) hipLaunchKernelGGL((dw_gemm_kernel<bf16_t, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((dw_gemm_kernel<bf16_t, 1>), grid, dim3(256), 0, st, g);
  } else {
    if (tf == 2) hipLaunchKernelGGL((dw_gemm_kernel<float, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((dw_gemm_kernel<float, 1>), grid, dim3(256), 0, st, g);
  }
  M2M_CHECK_HIP(hipGetLastError());
  if (g.ksplit > 1) {
    const int64_t n = (int64_t)N1 * N2;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st, kpart, C, N1, N2,
                       ldc, g.ksplit, 1.0f);
    M2M_CHECK_HIP(hipGetLastError());
  }
  return M2M_OK;
}

// dst[c][r] = src[r][c]: src [R][C] (row stride ld_s) -> dst [C][ld_d]; the columns [R, Rpad) of dst are zeroed
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ src, int64_t ld_s, TD* __restrict__ dst, int64_t ld_d, int R, int C,
                                                        int Rpad) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int rl = i >> 6, cl = i & 63;
    tile[rl][cl] = (r0 + rl < R && c0 + cl < C) ? to_f32(src[(int64_t)(r0 + rl) * ld_s + c0 + cl]) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cl = i >> 6, rl = i & 63;
    if (c0 + cl < C && r0 + rl < Rpad) dst[(int64_t)(c0 + cl) * ld_d + r0 + rl] = from_f32<TD>(tile[rl][cl]);
  }
}

// all weight matrices of the model in one launch: WT[off .. ] = W[off ..]^T in the storage type (block -> (matrix, tile) table)
struct WtBlock { int64_t off; int N, K, tn, tk; int il_half; };     // il_half > 0: a gate pair [wi_0 | wi_1], il_half = d_ff rows each
// ... and, in the same pass over the fp32 master (Wc != null: bf16 mode), the plain copy in the storage type the forward products
// read: every matrix a product reads is in the table, so the separate conversion of the whole parameter buffer (30 us, 121 MB read
// a second time) is gone.  WT == null (a forward-only pass): the copy only.
// Wil (bf16 mode): the gate-pair matrices once more with their rows interleaved in 32-row chunks (wi_0 rows [32c, 32c + 32), then the
// matching wi_1 rows), the B operand of the fused gated-GELU product (EPI_GATED_TRAIN): a wave's 64 output columns are then a | b of
// the same 32 hidden units.
template <typename TD>
__global__ __launch_bounds__(256) void weights_transpose_kernel(const WtBlock* __restrict__ blocks, const float* __restrict__ P, TD* __restrict__ WT,
                                                                TD* __restrict__ Wc, TD* __restrict__ Wil) {
  __shared__ float tile[64][65];
  const WtBlock b = blocks[blockIdx.x];
  const int n0 = b.tn * 64, k0 = b.tk * 64;
  const float* src = P + b.off;
  TD* dst = WT + b.off;
  // four elements per thread and access (16-byte loads, 8-byte bf16 stores; the element-wise form moved 2 bytes per store: 62 us for the
  // step's 265 MB); tiles at a ragged edge (K or N not a multiple of 4 there) take the element-wise loop
  const bool vec = sizeof(TD) == 2 && (b.K & 3) == 0 && (b.N & 3) == 0 && ((b.off & 3) == 0);
  if (vec) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int nl = i >> 4, kl = (i & 15) * 4;
      const bool in = n0 + nl < b.N && k0 + kl < b.K;          // (K % 4 == 0: the whole piece is inside or outside)
      const int64_t at = (int64_t)(n0 + nl) * b.K + k0 + kl;
      const float4 v = in ? *reinterpret_cast<const float4*>(src + at) : make_float4(0.f, 0.f, 0.f, 0.f);
      tile[nl][kl] = v.x; tile[nl][kl + 1] = v.y; tile[nl][kl + 2] = v.z; tile[nl][kl + 3] = v.w;
      if (in && (Wc || (Wil && b.il_half))) {
        const uint2 pk = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
        if (Wc) *reinterpret_cast<uint2*>(Wc + b.off + at) = pk;
        if (Wil && b.il_half) {
          const int n = n0 + nl, m = n < b.il_half ? n : n - b.il_half;
          *reinterpret_cast<uint2*>(Wil + b.off + (int64_t)((m >> 5) * 64 + (n < b.il_half ? 0 : 32) + (m & 31)) * b.K + k0 + kl) = pk;
        }
      }
    }
    if (!WT) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int kl = i >> 4, nl = (i & 15) * 4;
      if (k0 + kl < b.K && n0 + nl < b.N)
        *reinterpret_cast<uint2*>(dst + (int64_t)(k0 + kl) * b.N + n0 + nl) =
            make_uint2(pack2_bf16(tile[nl][kl], tile[nl + 1][kl]), pack2_bf16(tile[nl + 2][kl], tile[nl + 3][kl]));
    }
    return;
  }
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int nl = i >> 6, kl = i & 63;
    const bool in = n0 + nl < b.N && k0 + kl < b.K;
    const float v = in ? src[(int64_t)(n0 + nl) * b.K + k0 + kl] : 0.f;
    tile[nl][kl] = v;
    if (Wc && in) Wc[b.off + (int64_t)(n0 + nl) * b.K + k0 + kl] = from_f32<TD>(v);
    if (Wil && in && b.il_half) {
      const int n = n0 + nl, m = n < b.il_half ? n : n - b.il_half;
      Wil[b.off + (int64_t)((m >> 5) * 64 + (n < b.il_half ? 0 : 32) + (m & 31)) * b.K + k0 + kl] = from_f32<TD>(v);
    }
  }
  if (!WT) return;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int kl = i >> 6, nl = i & 63;
    if (k0 + kl < b.K && n0 + nl < b.N) dst[(int64_t)(k0 + kl) * b.N + n0 + nl] = from_f32<TD>(tile[nl][kl]);
  }
}

// fp8 mode: every projection matrix W [N][K] (fp32 master) -> MXFP8 in BOTH layouts in one launch: row-major with blocks
// along K (the forward's B operand) and transposed [K][Np] with blocks along N (the B operand of dX = dY . W).
// One 64-thread workgroup per 32 x 64 tile (table entry); e4m3.
struct W8Tile { int64_t off; int N, K, Np; int64_t q, qs, qt, qts; int tn, tk; };
__device__ inline float w8_scale_exp(float amax) {
  if (!(amax > 0.f)) return -127.f;
  int e;
  (void)frexpf(amax, &e);
  int se = e - 1 - 8;
  return (float)(se < -127 ? -127 : (se > 127 ? 127 : se));
}
__device__ inline uint32_t w8_pack4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f); c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
__global__ __launch_bounds__(64) void mxq_weights_kernel(const W8Tile* __restrict__ tiles, const float* __restrict__ P, uint8_t* __restrict__ w8) {
  __shared__ float tile[32][65];
  const W8Tile tl = tiles[blockIdx.x];
  const int n0 = tl.tn * 32, k0 = tl.tk * 64;
  const float* src = P + tl.off;
  for (int i = threadIdx.x; i < 32 * 64; i += 64) {
    const int nl = i >> 6, kl = i & 63;
    tile[nl][kl] = (n0 + nl < tl.N && k0 + kl < tl.K) ? src[(int64_t)(n0 + nl) * tl.K + k0 + kl] : 0.f;
  }
  __syncthreads();
  {  // rows: thread -> (row, k-block)
    const int nl = threadIdx.x >> 1, kb = threadIdx.x & 1;
    if (n0 + nl < tl.N && k0 + 32 * kb < tl.K) {
      float v[32], amax = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) { v[j] = tile[nl][32 * kb + j]; amax = fmaxf(amax, fabsf(v[j])); }
      const float se = w8_scale_exp(amax), inv = exp2f(-se);
      uint32_t* dst = reinterpret_cast<uint32_t*>(w8 + tl.q + (int64_t)(n0 + nl) * tl.K + k0 + 32 * kb);
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = w8_pack4(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
      w8[tl.qs + (int64_t)(n0 + nl) * (tl.K / 32) + (k0 + 32 * kb) / 32] = (uint8_t)((int)se + 127);
    }
  }
  {  // columns: thread -> column k, block of 32 rows n0 .. n0 + 31
    const int kl = threadIdx.x;
    if (k0 + kl < tl.K) {
      float v[32], amax = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) { v[j] = tile[j][kl]; amax = fmaxf(amax, fabsf(v[j])); }
      const float se = w8_scale_exp(amax), inv = exp2f(-se);
      uint32_t* dst = reinterpret_cast<uint32_t*>(w8 + tl.qt + (int64_t)(k0 + kl) * tl.Np + n0);
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = w8_pack4(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
      w8[tl.qts + (int64_t)(k0 + kl) * (tl.Np / 32) + n0 / 32] = (uint8_t)((int)se + 127);
    }
  }
}

int launch_bgemm(int precision, int epi, const BGemmArgs& g, hipStream_t st) {
  const int E = precision == M2M_PREC_BF16 ? 8 : 4;
  M2M_REQUIRE(g.M >= 1 && g.N >= 1 && g.K >= 1 && g.nb1 >= 1 && g.nb2 >= 1, "bgemm: empty problem");
  M2M_REQUIRE(g.lda % E == 0 && g.ldb % E == 0, "bgemm: operand row strides (%lld, %lld) must be multiples of %d elements (16-byte rows)",
              (long long)g.lda, (long long)g.ldb, E);
  M2M_REQUIRE(g.sA1 % E == 0 && g.sA2 % E == 0 && g.sB1 % E == 0 && g.sB2 % E == 0, "bgemm: batch strides must keep 16-byte alignment");
  M2M_REQUIRE((int64_t)g.nb1 * g.nb2 * (g.A2 ? 2 : 1) <= 65535, "bgemm: too many batch entries");
  M2M_REQUIRE(!g.A2 || (g.B2 && g.C2 && g.ksplit <= 1 && g.ldb2 % E == 0 && g.sB1_2 % E == 0 && g.sB2_2 % E == 0 && (!g.b_kmajor || g.ldb2 >= align_up(g.N, E))),
              "bgemm: bad second product");
  M2M_REQUIRE((!g.a_kmajor || g.lda >= align_up(g.M, E)) && (!g.b_kmajor || g.ldb >= align_up(g.N, E)),
              "bgemm: a k-major operand's row stride must cover its rows padded to %d", E);
  if (g.ksplit > 1) {
    M2M_REQUIRE(g.nb1 == 1 && g.nb2 == 1 && epi == TG_STORE_F32 && g.Cpart && g.kchunk % TG_BK_MAX == 0, "bgemm: split-K is for plain fp32-store products");
    int rc = precision == M2M_PREC_BF16 ? launch_bgemm_t<bf16_t>(epi, g, st) : launch_bgemm_t<float>(epi, g, st);
    if (rc != M2M_OK) return rc;
    const int64_t n = (int64_t)g.M * g.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st, g.Cpart,
                       reinterpret_cast<float*>(g.C), g.M, g.N, g.ldc, g.ksplit, g.alpha);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  return precision == M2M_PREC_BF16 ? launch_bgemm_t<bf16_t>(epi, g, st) : launch_bgemm_t<float>(epi, g, st);
}

// ============================================================ element / row kernels ====
template <typename T>
__global__ void cvt_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = from_f32<T>(src[i]);
}
// four consecutive elements of the storage type <-> four floats (8- / 16-byte accesses)
template <typename T> __device__ inline void st_store4(T* p, float a, float b, float c, float d);
template <> __device__ inline void st_store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(a, b), pack2_bf16(c, d));
}
template <> __device__ inline void st_store4<float>(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
template <typename T> __device__ inline float4 st_load4(const T* p);
template <> __device__ inline float4 st_load4<bf16_t>(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xFFFF0000u));
}
template <> __device__ inline float4 st_load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
// dst = T(keep(i) * scale * src)  (the gradient entering a dropped branch)
template <typename T>
__global__ void cvt_drop_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n, DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = from_f32<T>(drop_keep(key, i, thresh) ? src[i] * scale : 0.f);
}
static inline int grid_1d(int64_t n, int per_block = 256) {
  int64_t g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
int launch_cvt(int precision, const float* src, void* dst, int64_t n, hipStream_t st) {
  if (precision == M2M_PREC_BF16) hipLaunchKernelGGL(cvt_kernel<bf16_t>, dim3(grid_1d(n)), dim3(256), 0, st, src, (bf16_t*)dst, n);
  else hipLaunchKernelGGL(cvt_kernel<float>, dim3(grid_1d(n)), dim3(256), 0, st, src, (float*)dst, n);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// ---- attention probabilities: P = softmax(scores + bias) row by row (hf: modeling_t5.py:159-170: no 1/sqrt(d)) ----
// scores fp32 [BH][Sq][ldp]; bias_tab [H][tab_stride] by (key - query) + tab_center, or null; causal: keys > query masked.
// One wave per row; P is written in T with the padding columns [Sk, ldp) zeroed (they are GEMM operand columns).
// With dropout (hf: modeling_t5.py "attn_weights = dropout(attn_weights)") the kept-and-scaled copy Pd feeds the P.V product;
// P itself is what the softmax backward needs.
constexpr int SM_NR = 8;      // row elements per lane kept in registers (rows up to 512 keys)
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ sc, T* __restrict__ P, int rows_total, int H,
                                                          int Sq, int Sk, int ldp, const float* __restrict__ bias_tab,
                                                          int tab_stride, int tab_center, int causal, T* __restrict__ Pd,
                                                          DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = (Pd && thresh) ? drop_site_key(dk) : 0ull;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  const int q = row % Sq, bh = row / Sq, hh = bh % H;
  const float* s = sc + (int64_t)row * ldp;
  T* p = P + (int64_t)row * ldp;
  const int kend = causal ? min(Sk, q + 1) : Sk;
  const float* bt = bias_tab ? bias_tab + (int64_t)hh * tab_stride + tab_center - q : nullptr;
  if (ldp <= 64 * SM_NR) {
    // rows of up to 512 keys (every shape of the model): the row lives in registers between the three passes — one read of
    // the scores and one exponential per element instead of three reads and two exponentials; same values, same summation order
    float v[SM_NR];
    float mx = -1e30f;
#pragma unroll
    for (int u = 0; u < SM_NR; ++u) {
      const int k = lane + 64 * u;
      v[u] = k < kend ? s[k] + (bt ? bt[k] : 0.f) : -1e30f;
      if (k < kend) mx = fmaxf(mx, v[u]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < SM_NR; ++u) {
      const int k = lane + 64 * u;
      v[u] = k < kend ? expf(v[u] - mx) : 0.f;
      if (k < kend) sum += v[u];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int u = 0; u < SM_NR; ++u) {
      const int k = lane + 64 * u;
      if (k < ldp) {
        const T pt = from_f32<T>(v[u] * inv);
        p[k] = pt;
        if (Pd) Pd[(int64_t)row * ldp + k] = drop_keep(key, (int64_t)row * ldp + k, thresh) ? from_f32<T>(to_f32(pt) * scale) : from_f32<T>(0.f);
      }
    }
    return;
  }
  float mx = -1e30f;
  for (int k = lane; k < kend; k += 64) mx = fmaxf(mx, s[k] + (bt ? bt[k] : 0.f));
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < kend; k += 64) sum += expf(s[k] + (bt ? bt[k] : 0.f) - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int k = lane; k < ldp; k += 64) {
    const float pv = k < kend ? expf(s[k] + (bt ? bt[k] : 0.f) - mx) * inv : 0.f;
    const T pt = from_f32<T>(pv);
    p[k] = pt;
    if (Pd) Pd[(int64_t)row * ldp + k] = drop_keep(key, (int64_t)row * ldp + k, thresh) ? from_f32<T>(to_f32(pt) * scale) : from_f32<T>(0.f);
  }
}
// K and V of every (clip, head) transposed for the fused products of the stripe kernels: src element (b, s, h, d) at
// src + (b*S + s)*ld + which*wstride + h*64 + d  ->  dst[which][(b*H + h)*64 + d][s], row pitch Sp (a multiple of 32), zero in [S, Sp).
template <typename T>
__global__ __launch_bounds__(256) void kv_transpose_kernel(const T* __restrict__ src, int64_t ld, int64_t wstride, T* __restrict__ dst, int S, int Sp, int H,
                                                           int64_t dst_wstride) {
  // register transposes only (tr_col): a thread takes an E x E block — E rows of s, 16 bytes of d each — and writes E rows of d,
  // 16 bytes of s each; for a fixed register the lanes of a wave cover whole 128-byte lines on both sides (the LDS form with
  // 2-byte accesses took 9.3 us per launch)
  constexpr int E = 16 / sizeof(T), DB = DK / E;              // d blocks per row: 8 (bf16) / 16 (fp32)
  const int bh = blockIdx.y, which = blockIdx.z, b = bh / H, hh = bh - b * H;
  const int blk = blockIdx.x * 256 + threadIdx.x;             // block id: d block fastest on the load side
  const int sb = blk / DB, db = blk - sb * DB;
  const int s0 = sb * E;
  if (s0 >= Sp) return;
  const T* sp = src + (int64_t)b * S * ld + which * wstride + hh * DK + db * E;
  uint4 rg[E];
#pragma unroll
  for (int k = 0; k < E; ++k) rg[k] = (s0 + k < S) ? *reinterpret_cast<const uint4*>(sp + (int64_t)(s0 + k) * ld) : make_uint4(0, 0, 0, 0);
  T* dp = dst + which * dst_wstride + ((int64_t)bh * DK + db * E) * Sp + s0;
#pragma unroll
  for (int jj = 0; jj < E; ++jj) *reinterpret_cast<uint4*>(dp + (int64_t)jj * Sp) = tr_col<T, E>(rg, jj);
}

// ---- attention stripes: scores -> probabilities (forward) and dP -> dS (backward) WITHOUT the fp32 [Sq, Sk] image ----
// Training sequences are short (S, L <= 320 here), so one wave can hold a whole stripe of 32 queries x all keys in MFMA
// accumulators: T^T = X . Y^T with the KEYS as tile rows (X = K or V, [Sk, 64]) and the QUERIES as tile columns
// (Y = Q or dO), so a lane owns ONE query (column l & 31) and its row reductions are in-lane plus one exchange with lane ^ 32
// (the orientation of the inference flash kernel).  Forward: + T5 bias (per-head row of the (key - query) table, staged in
// LDS), causal mask, softmax, P (and its dropped copy) written in T.  Backward: dP = V . dO^T, dS = P o (dP~ - sum P o dP~)
// with dP~ the dropout-masked dP.  Replaces a batched GEMM that wrote the fp32 image + a row kernel that read it back
// (35 MB each way per call at 16 clips): 15.6 + 25.7 us -> one launch (forward), 15.6 + 18 us -> one launch (backward).
#ifdef M2M_ST_STAMP      // diagnostic builds only: phase times of one workgroup of the stripe kernel (100 MHz s_memrealtime)
__device__ unsigned long long g_st_stamp[4][12];
#define ST_STAMP(i) do { if (threadIdx.x == 0 && bh == 5 && sx == 4) g_st_stamp[(BWD ? 2 : 0) + (BIAS ? 1 : 0)][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ST_WAITLOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ST_STAMP(i) do {} while (0)
#define ST_WAITLOADS() do {} while (0)
#endif
struct StripeArgs {
  const void *X, *Y;                 // (key, d) at X + b*sX1 + h*sX2 + key*ldx + d;  (query, d) at Y + b*sY1 + h*sY2 + q*ldy + d
  int64_t ldx, sX1, sX2, ldy, sY1, sY2;
  void* P;                           // [nB*H][Sq][ldp] T: forward out / backward in
  void* Pd;                          // forward: dropped copy of P, or null
  void* dS;                          // backward out
  float* diag_part;                  // backward, self-attention with bias: [nB*H][stripes][Sk + 31] diagonal sums of dS per stripe, or null
  // optional fused product with the rows this workgroup has just formed: forward O = P~ . V, backward dQ = dS . K.  Xt is the
  // second operand TRANSPOSED ([nB*H][64][xt_ld], zero beyond Sk: kv_transpose_kernel), (query, d) of the result at
  // O + b*sO1 + h*sO2 + q*ldo + d; null: the product is a launch of its own
  const void* Xt;
  int64_t xt_ld;
  void* O;
  int64_t ldo, sO1, sO2;
  const float* bias_tab;             // forward, self-attention: [H][tab_stride] by (key - query + tab_center); null: no bias
  int tab_stride, tab_center;
  int H, Sq, Sk, ldp, causal;
  DropKey dk;
  uint32_t thresh;
  float scale;
  int n_stripes, xcd_total;          // set by launch_attn_stripe: stripes per (clip, head); > 0: 1-D launch in XCD-contiguous order
};
constexpr int ST_NT = 16;            // key tiles per stripe: Sk <= 512 (the X operand of a (clip, head) is staged in LDS)
// softmax exponential: accurate in the fp32 (parity) mode; multiply + v_exp_f32 where the result is rounded to bf16
template <typename T> __device__ inline float m2m_exp_t(float x) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  else return expf(x);
}


// SLIM: the operand fragments are NOT kept across the two passes (reloaded, as the fp32 form always does): ~95 registers, FIVE
// workgroups per CU.  Chosen by the launch when the grid has more workgroups than the 1 024 that four per CU hold but not more
// than 1 280: the encoder's 9 stripes x 128 (clip, head) = 1 152 ran as a full round plus a round of 128 — twice one
// workgroup's time (stamps: 17.6 us of the launch's 38) — and now run as one.
template <typename T, bool BWD, bool DROP, bool BIAS, bool SLIM = false>
__global__ __launch_bounds__(256, SLIM ? 5 : 4) void attn_stripe_kernel(StripeArgs a) {
  // A workgroup = 32 queries of one (clip, head); its four waves split the key tiles (wave w: tiles w, w + 4, ...), so the
  // dependent chain of a wave is at most ST_NT / 4 tiles per pass and a 16-clip launch has > 1 000 workgroups.  Operand
  // fragments come straight from memory, all of a wave's loads in flight at once (bf16: kept in registers for both passes);
  // LDS holds only this head's bias row and the per-wave row statistics.  (First forms: all tiles' accumulators alive in
  // one wave — 368 registers, 37 us per launch; K / V staged in LDS for 128 queries per workgroup — 25 us.)
  extern __shared__ __align__(16) unsigned char st_smem[];
  constexpr int TPW = ST_NT / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h2 = lane >> 5;
  // XCD-contiguous order (workgroup id % 8 picks the XCD): the stripes of one (clip, head) — which all read its K, V, K^T / V^T —
  // run side by side on one XCD instead of on all eight (PMC: 113 MB fetched per backward launch with the launch order)
  int sx = blockIdx.x, bh = blockIdx.y;
  const int n_stripes = a.n_stripes;
  if (a.xcd_total > 0) {
    const int per = (a.xcd_total + 7) >> 3, l = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (l >= a.xcd_total || (int)(blockIdx.x >> 3) >= per) return;       // padding workgroups (uniform, before any barrier)
    sx = l % n_stripes; bh = l / n_stripes;
  }
  const int b = bh / a.H, hh = bh - b * a.H;
  const int q0 = sx * 32;
  __shared__ float red_a[4][32], red_b[4][32];       // per-wave partial row statistics
  constexpr bool has_bias = !BWD && BIAS;
  float* st_bias = reinterpret_cast<float*>(st_smem);
  const int nt = (a.Sk + 31) >> 5;
  // row staging: the 32 x ldp block of P / dS this workgroup produces (or consumes) goes through LDS, so memory sees whole
  // rows in 16-byte chunks instead of 8-byte pieces of 32 different rows per store (forward 18.7 -> us, backward 25.3 -> us)
  const int LP = nt * 32 + 16 / (int)sizeof(T);                                   // row pitch, elements
  T* pl = reinterpret_cast<T*>(st_smem + (has_bias ? ((size_t)(a.tab_stride + 32) * 4 + 15) / 16 * 16 : 0));
  T* pl2 = pl + 32 * LP;                                                          // forward with dropout: the dropped copy
  const int q = q0 + r, qc = min(q, a.Sq - 1);
  const T* X = reinterpret_cast<const T*>(a.X) + b * a.sX1 + hh * a.sX2;
  const T* Y = reinterpret_cast<const T*>(a.Y) + b * a.sY1 + hh * a.sY2;
  Frag<T> yf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) yf[s] = load_frag(Y + (int64_t)qc * a.ldy + 16 * s + 8 * h2);
  constexpr bool KEEP = sizeof(T) == 2 && !SLIM;     // fp32 fragments are twice the registers: reloaded per pass instead
  Frag<T> xf[KEEP ? TPW : 1][4];
  if constexpr (KEEP) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const T* xr = X + (int64_t)min((wave + 4 * j) * 32 + r, a.Sk - 1) * a.ldx + 8 * h2;      // clamped: keys >= Sk are masked below
#pragma unroll
      for (int s = 0; s < 4; ++s) xf[j][s] = load_frag(xr + 16 * s);
    }
  }
  ST_STAMP(0);
  ST_WAITLOADS();
  ST_STAMP(1);
  constexpr int EC = 16 / sizeof(T);
  const int cpr = a.ldp / EC;                        // 16-byte chunks per row (ldp is a multiple of 8)
  const int64_t blk = ((int64_t)bh * a.Sq + q0) * a.ldp;
  const uint64_t key = DROP ? drop_site_key(a.dk) : 0ull;
  if constexpr (BWD) {                               // P rows -> LDS (read twice below), coalesced
    const T* src = reinterpret_cast<const T*>(a.P) + blk;
    const int cpl = nt * 32 / EC;                      // the WHOLE LDS row: a causal stripe never rewrites the tiles above its diagonal,
                                                       // and the fused product reads them (uninitialised LDS x 0 is not 0)
    // all of a thread's chunks requested before the first is stored: ONE round trip instead of one per loop iteration (stamps: 4.6 us
    // of the workgroup's 17.6 with the plain loop).  32 rows x at most ST_NT * 32 elements = at most 8 (bf16) / 16 (fp32) chunks per thread.
    constexpr int PCH = ST_NT * 32 / EC * 32 / 256 / 2;       // in two batches: a single one spills next to the kept operand fragments
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half * PCH * 256 >= 32 * cpl) break;                 // (uniform)
      uint4 pv[PCH];
#pragma unroll
      for (int u = 0; u < PCH; ++u) {
        const int c = threadIdx.x + 256 * (half * PCH + u), row = c / cpl, col = (c - row * cpl) * EC;
        pv[u] = (c < 32 * cpl && q0 + row < a.Sq && col < a.ldp) ? *reinterpret_cast<const uint4*>(src + (int64_t)row * a.ldp + col) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < PCH; ++u) {
        const int c = threadIdx.x + 256 * (half * PCH + u), row = c / cpl, col = (c - row * cpl) * EC;
        if (c < 32 * cpl) *reinterpret_cast<uint4*>(pl + row * LP + col) = pv[u];
      }
    }
    __syncthreads();
    if constexpr (DROP) {
      // The dropped probabilities again (the dV product's operand; a scratch buffer in the forward pass) — what drop_copy_kernel
      // produced in a launch of its own, 18 launches / 0.28 ms of a step: masked straight from the staged P rows into memory, chunk by
      // chunk, before the passes overwrite them.  No second row block, so the five-per-CU variant stays available with dropout on.
      if (a.Pd) {
        T* dst2 = reinterpret_cast<T*>(a.Pd) + blk;
        for (int c = threadIdx.x; c < 32 * cpr; c += 256) {
          const int row = c / cpr, col = (c - row * cpr) * EC;
          if (q0 + row >= a.Sq) continue;
          const int64_t at = ((int64_t)bh * a.Sq + q0 + row) * a.ldp + col;
#pragma unroll
          for (int g4 = 0; g4 < EC / 4; ++g4) {
            const float4 v = st_load4<T>(pl + row * LP + col + 4 * g4);
            const uint32_t kb = drop_keep4(key, at + 4 * g4, a.thresh);
            st_store4<T>(dst2 + (int64_t)row * a.ldp + col + 4 * g4, (kb & 1u) ? v.x * a.scale : 0.f, (kb & 2u) ? v.y * a.scale : 0.f,
                         (kb & 4u) ? v.z * a.scale : 0.f, (kb & 8u) ? v.w * a.scale : 0.f);
          }
        }
      }
    }
  }
  if (has_bias) {
    for (int i = threadIdx.x; i < a.tab_stride + 32; i += 256) st_bias[i] = i < a.tab_stride ? a.bias_tab[(int64_t)hh * a.tab_stride + i] : 0.f;
    __syncthreads();
  }
  ST_STAMP(2);
  // element i of a tile: key 32 kt + (i & 3) + 8 (i >> 2) + 4 h2, query q (this lane's column); the tile product is
  // RECOMPUTED in the second pass (4 MFMAs) instead of being kept
  auto tile = [&](int j) {
    f32x16 acc = zero_acc();
    if constexpr (KEEP) {
#pragma unroll
      for (int s = 0; s < 4; ++s) mma16(acc, xf[j][s], yf[s]);
    } else {
      const T* xr = X + (int64_t)min((wave + 4 * j) * 32 + r, a.Sk - 1) * a.ldx + 8 * h2;
#pragma unroll
      for (int s = 0; s < 4; ++s) mma16(acc, load_frag(xr + 16 * s), yf[s]);
    }
    return acc;
  };
  // fused product rows . Xt^T (see StripeArgs): waves 0 and 1 take the two 32-wide halves of d, 8 k-steps' fragments in flight at a time
  auto rows_times_xt = [&](const T* rows) {
    if (!a.Xt || wave >= 2) return;
    const T* xt = reinterpret_cast<const T*>(a.Xt) + ((int64_t)bh * DK + wave * 32 + r) * a.xt_ld + 8 * h2;
    f32x16 acc = zero_acc();
    const int nks = nt * 2;                           // k-steps of 16 keys
    for (int k8 = 0; k8 < nks; k8 += 8) {
      Frag<T> bf[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) bf[u] = load_frag(xt + 16 * min(k8 + u, nks - 1));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k8 + u < nks) mma16(acc, load_frag(rows + r * LP + 16 * (k8 + u) + 8 * h2), bf[u]);
    }
    T* o = reinterpret_cast<T*>(a.O) + b * a.sO1 + hh * a.sO2 + wave * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = q0 + acc_row(i, lane);
      if (qq < a.Sq) o[(int64_t)qq * a.ldo] = from_f32<T>(acc[i]);
    }
  };
  const int kend = BWD ? a.Sk : (a.causal ? min(a.Sk, q + 1) : a.Sk);
  const int64_t prow = ((int64_t)bh * a.Sq + qc) * a.ldp;
  if constexpr (!BWD) {
    const float* bt = st_bias + a.tab_center - qc;
    // pass 1: running (max, sum) of this lane's half of the keys
    float m = -1e30f, l = 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int kt = wave + 4 * j;
      if (kt >= nt) break;                             // wave-uniform
      if (a.causal && kt * 32 > q0 + 31) break;        // causal: this and every later tile of the wave lie above the stripe's diagonal
      f32x16 acc = tile(j);
      float tm = -1e30f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
        float v = acc[i];
        if (has_bias) v += bt[k];                      // the LDS row is padded by 32 entries: no clamp
        v = k < kend ? v : -1e30f;
        acc[i] = v;
        tm = fmaxf(tm, v);
      }
      const float mn = fmaxf(m, tm);
      float ts = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) ts += m2m_exp_t<T>(acc[i] - mn);      // masked entries are -1e30: exp gives exactly 0, no branch
      l = l * m2m_exp_t<T>(m - mn) + ts;
      m = mn;
    }
    {
      const float mo = lane_xor<32>(m), lo = lane_xor<32>(l);
      const float mt = fmaxf(m, mo);
      l = l * m2m_exp_t<T>(m - mt) + lo * m2m_exp_t<T>(mo - mt);
      m = mt;
    }
    ST_STAMP(3);
    if (h2 == 0) { red_a[wave][r] = m; red_b[wave][r] = l; }
    __syncthreads();
    ST_STAMP(4);
    {                                                // the four waves' (max, sum) of this query, fixed order
      const float m0 = red_a[0][r], m1 = red_a[1][r], m2 = red_a[2][r], m3 = red_a[3][r];
      m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
      l = red_b[0][r] * m2m_exp_t<T>(m0 - m) + red_b[1][r] * m2m_exp_t<T>(m1 - m) + red_b[2][r] * m2m_exp_t<T>(m2 - m) + red_b[3][r] * m2m_exp_t<T>(m3 - m);
    }
    const float inv = 1.0f / l;
    // pass 2: probabilities
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int kt = wave + 4 * j;
      if (kt >= nt) break;                             // wave-uniform
      if (a.causal && kt * 32 > q0 + 31) {             // fully masked tile: zeros, no product, no exponentials
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          st_store4<T>(pl + r * LP + kt * 32 + 8 * g + 4 * h2, 0.f, 0.f, 0.f, 0.f);
          if constexpr (DROP) st_store4<T>(pl2 + r * LP + kt * 32 + 8 * g + 4 * h2, 0.f, 0.f, 0.f, 0.f);
        }
        continue;
      }
      const f32x16 acc = tile(j);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = kt * 32 + 8 * g + 4 * h2;
        float pv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k0 + e;
          float v = acc[4 * g + e];
          if (has_bias) v += bt[k];
          v = k < kend ? v : -1e30f;
          pv[e] = to_f32(from_f32<T>(m2m_exp_t<T>(v - m) * inv));
        }
        st_store4<T>(pl + r * LP + k0, pv[0], pv[1], pv[2], pv[3]);          // keys >= kend are zeros
        if constexpr (DROP) {
          if (a.Pd) {                                   // the dropped copy is wanted in memory: a second row block
            float dv[4];
            const uint32_t kb = drop_keep4(key, prow + k0, a.thresh);
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[e] = ((kb >> e) & 1u) ? pv[e] * a.scale : 0.f;
            st_store4<T>(pl2 + r * LP + k0, dv[0], dv[1], dv[2], dv[3]);
          }
        }
      }
    }
    ST_STAMP(5);
    __syncthreads();
    ST_STAMP(6);
    {
      T* dst = reinterpret_cast<T*>(a.P) + blk;
      T* dst2 = (DROP && a.Pd) ? reinterpret_cast<T*>(a.Pd) + blk : nullptr;
      for (int c = threadIdx.x; c < 32 * cpr; c += 256) {
        const int row = c / cpr, col = (c - row * cpr) * EC;
        if (q0 + row < a.Sq) {
          *reinterpret_cast<uint4*>(dst + (int64_t)row * a.ldp + col) = *reinterpret_cast<const uint4*>(pl + row * LP + col);
          if (DROP && dst2) *reinterpret_cast<uint4*>(dst2 + (int64_t)row * a.ldp + col) = *reinterpret_cast<const uint4*>(pl2 + row * LP + col);
        }
      }
    }
    if constexpr (DROP) {
      // The dropped probabilities are only the operand of the fused product (the backward pass re-forms them): no second row block —
      // half the LDS, four (or five) workgroups per CU instead of three — the rows are masked IN PLACE once they have been copied out,
      // every lane rewriting the elements it wrote
      if (!a.Pd && a.Xt) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const int kt = wave + 4 * j;
          if (kt >= nt) break;                             // wave-uniform
          if (a.causal && kt * 32 > q0 + 31) continue;     // zeros stay zeros
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int k0 = kt * 32 + 8 * g + 4 * h2;
            const float4 pv = st_load4<T>(pl + r * LP + k0);
            const float pe[4] = {pv.x, pv.y, pv.z, pv.w};
            const uint32_t kb4 = drop_keep4(key, prow + k0, a.thresh);
            float dv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) dv[e] = ((kb4 >> e) & 1u) ? pe[e] * a.scale : 0.f;
            st_store4<T>(pl + r * LP + k0, dv[0], dv[1], dv[2], dv[3]);
          }
        }
        __syncthreads();
      }
    }
    ST_STAMP(7);
    rows_times_xt((DROP && a.Pd) ? pl2 : pl);         // O = P~ . V
    ST_STAMP(9);
  } else {
    T* prd = pl + r * LP;                            // this lane's query row of P, in LDS
    // dP arrives for the DROPPED probabilities: through the mask first.  The keep bits of a lane's 16 elements per tile are hashed
    // ONCE (pass 1) and kept as a bit mask for pass 2 and for the dropped copy (three hashes per element before: 22 -> 36 us per launch)
    uint32_t kbits[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) kbits[j] = 0u;
    auto masked = [&](float dp, int k, bool keep) {
      if constexpr (DROP) dp = keep ? dp * a.scale : 0.f;
      return k < a.Sk ? dp : 0.f;
    };
    // pass 1: t = sum_k P dP~
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int kt = wave + 4 * j;
      if (kt >= nt) break;                             // wave-uniform
      if (a.causal && kt * 32 > q0 + 31) break;        // causal: P is zero there
      const f32x16 acc = tile(j);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = kt * 32 + 8 * g + 4 * h2;
        const float4 pv = st_load4<T>(prd + k0);
        const float pe[4] = {pv.x, pv.y, pv.z, pv.w};
        uint32_t kb = 0xFu;
        if constexpr (DROP) { kb = drop_keep4(key, prow + k0, a.thresh); kbits[j] |= kb << (4 * g); }
#pragma unroll
        for (int e = 0; e < 4; ++e) t += (k0 + e < a.Sk ? pe[e] : 0.f) * masked(acc[4 * g + e], k0 + e, (kb >> e) & 1u);
      }
    }
    t += lane_xor<32>(t);
    ST_STAMP(3);
    if (h2 == 0) red_a[wave][r] = t;
    __syncthreads();
    ST_STAMP(4);
    t = (red_a[0][r] + red_a[1][r]) + (red_a[2][r] + red_a[3][r]);
    // pass 2: dS = P (dP~ - t)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int kt = wave + 4 * j;
      if (kt >= nt) break;                             // wave-uniform
      if (a.causal && kt * 32 > q0 + 31) {             // dS = P (...) = 0: the staged P rows already hold zeros there
        continue;
      }
      const f32x16 acc = tile(j);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = kt * 32 + 8 * g + 4 * h2;
        const float4 pv = st_load4<T>(prd + k0);
        const float pe[4] = {pv.x, pv.y, pv.z, pv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (k0 + e < a.Sk) ? pe[e] * (masked(acc[4 * g + e], k0 + e, (kbits[j] >> (4 * g + e)) & 1u) - t) : 0.f;
        st_store4<T>(prd + k0, o[0], o[1], o[2], o[3]);                        // in place: the same lane read these four
      }
    }
    ST_STAMP(5);
    __syncthreads();
    ST_STAMP(6);
    {
      T* dst = reinterpret_cast<T*>(a.dS) + blk;
      for (int c = threadIdx.x; c < 32 * cpr; c += 256) {
        const int row = c / cpr, col = (c - row * cpr) * EC;
        if (q0 + row < a.Sq) *reinterpret_cast<uint4*>(dst + (int64_t)row * a.ldp + col) = *reinterpret_cast<const uint4*>(pl + row * LP + col);
      }
    }
    // relative-position-bias gradient, stage 1 (replaces bias_diag_kernel's pass over dS in memory): the sums of this stripe's
    // dS along the diagonals key - row = i - 31, i in [0, Sk + 31), rows in fixed order; bias_bucket2_kernel adds the stripes
    ST_STAMP(7);
    if (a.diag_part) {
      const int dl = a.Sk + 31, rows = min(32, a.Sq - q0);
      float* out = a.diag_part + ((int64_t)bh * n_stripes + sx) * dl;
      for (int i = threadIdx.x; i < dl; i += 256) {
        float acc = 0.f;
        for (int r8 = 0; r8 < 32; r8 += 8) {          // eight LDS reads in flight, added in row order (the sum is what the plain loop gave)
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int rr = r8 + u, k = i - 31 + rr;
            v[u] = (rr < rows && k >= 0 && k < a.Sk) ? to_f32(pl[rr * LP + k]) : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += v[u];
        }
        out[i] = acc;
      }
    }
    ST_STAMP(8);
    rows_times_xt(pl);                               // dQ = dS . K
    ST_STAMP(9);
  }
}

template <typename T>
static int launch_attn_stripe(bool bwd, const StripeArgs& a_in, int nB, hipStream_t st) {
  StripeArgs a = a_in;
  dim3 grid((unsigned)ceil_div(a.Sq, 32), (unsigned)(nB * a.H));
  a.n_stripes = (int)grid.x;
  static const bool xcd = [] { const char* v = getenv("M2M_XCD_ORDER"); return !(v && v[0] == '0'); }();
  if (xcd && grid.x * grid.y >= 64) {
    a.xcd_total = (int)(grid.x * grid.y);
    grid = dim3((unsigned)(8 * ceil_div(a.xcd_total, 8)));
  }
  const bool drop = a.thresh != 0, bias = !bwd && a.bias_tab != nullptr;
  const size_t bias_bytes = bias ? ((size_t)(a.tab_stride + 32) * 4 + 15) / 16 * 16 : 0;
  const size_t row_bytes = (size_t)32 * (ceil_div(a.Sk, 32) * 32 + 16 / sizeof(T)) * sizeof(T);
  const size_t smem = bias_bytes + row_bytes * ((!bwd && drop && a.Pd) ? 2 : 1);      // forward with a dropped copy of P wanted in memory: two row blocks
  // five workgroups per CU when that turns two rounds into one (and their LDS fits): see SLIM above
  static const bool slim_on = [] { const char* v = getenv("M2M_ST_SLIM"); return !(v && v[0] == '0'); }();
  const int64_t n_wgs = (int64_t)a.n_stripes * nB * a.H;
  const bool slim = slim_on && sizeof(T) == 2 && n_wgs > 4 * 256 && n_wgs <= 5 * 256 && 5 * (smem + 1024) <= 160 * 1024;
#define M2M_ST_LAUNCH(B_, D_, I_)                                                                              \
  do {                                                                                                         \
    if (slim) {                                                                                                \
      if constexpr (sizeof(T) == 2) {                                                                          \
        M2M_OPT_IN_LDS((attn_stripe_kernel<T, B_, D_, I_, true>), 158 * 1024);                                    \
        hipLaunchKernelGGL((attn_stripe_kernel<T, B_, D_, I_, true>), grid, dim3(256), smem, st, a);           \
      }                                                                                                        \
    } else {                                                                                                   \
      M2M_OPT_IN_LDS((attn_stripe_kernel<T, B_, D_, I_>), 158 * 1024);      /* + 1 KB of static LDS */            \
      hipLaunchKernelGGL((attn_stripe_kernel<T, B_, D_, I_>), grid, dim3(256), smem, st, a);                   \
    }                                                                                                          \
  } while (0)
  if (bwd) {
    if (drop) M2M_ST_LAUNCH(true, true, false); else M2M_ST_LAUNCH(true, false, false);
  } else if (bias) {
    if (drop) M2M_ST_LAUNCH(false, true, true); else M2M_ST_LAUNCH(false, false, true);
  } else {
    if (drop) M2M_ST_LAUNCH(false, true, false); else M2M_ST_LAUNCH(false, false, false);
  }
#undef M2M_ST_LAUNCH
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// Pd = dropout(P) again (backward: the dV product needs it, it was a scratch buffer in the forward)
template <typename T>
__global__ void drop_copy_kernel(const T* __restrict__ P, T* __restrict__ Pd, int64_t n, DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) Pd[i] = drop_keep(key, i, thresh) ? from_f32<T>(to_f32(P[i]) * scale) : from_f32<T>(0.f);
}

// dS = P o (dP - rowsum(P o dP))   (softmax backward; P in T as the forward stored it, dP fp32), dS in T, padding zeroed
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, const float* __restrict__ dP, T* __restrict__ dS,
                                                          int rows_total, int Sk, int ldp, DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  const T* p = P + (int64_t)row * ldp;
  const float* dp = dP + (int64_t)row * ldp;
  T* ds = dS + (int64_t)row * ldp;
  // dP arrives for the DROPPED probabilities: through the mask first (thresh == 0: identity)
  auto dpe = [&](int k) { return thresh ? (drop_keep(key, (int64_t)row * ldp + k, thresh) ? dp[k] * scale : 0.f) : dp[k]; };
  if (ldp <= 64 * SM_NR) {                 // P and dP of the row stay in registers between the two passes
    float pv[SM_NR], dv[SM_NR];
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < SM_NR; ++u) {
      const int k = lane + 64 * u;
      pv[u] = k < Sk ? to_f32(p[k]) : 0.f;
      dv[u] = k < Sk ? dpe(k) : 0.f;
      if (k < Sk) t += pv[u] * dv[u];
    }
    t = wave_sum(t);
#pragma unroll
    for (int u = 0; u < SM_NR; ++u) {
      const int k = lane + 64 * u;
      if (k < ldp) ds[k] = from_f32<T>(k < Sk ? pv[u] * (dv[u] - t) : 0.f);
    }
    return;
  }
  float t = 0.f;
  for (int k = lane; k < Sk; k += 64) t += to_f32(p[k]) * dpe(k);
  t = wave_sum(t);
  for (int k = lane; k < ldp; k += 64) ds[k] = from_f32<T>(k < Sk ? to_f32(p[k]) * (dpe(k) - t) : 0.f);
}


// relative-position-bias gradient.  Stage 1: one block per (clip, head): part[b][h][rel] = sum over the diagonal
// key - query = rel - (Sq-1) of dS[b,h,q,k]; consecutive threads take consecutive diagonals, so every pass over q reads
// consecutive addresses.  Stage 2 sums the clips and the diagonals of each bucket.  Fixed order throughout.
template <typename T>
__global__ __launch_bounds__(256) void bias_diag_kernel(const T* __restrict__ dS, float* __restrict__ part, int H, int Sq, int Sk, int ldp) {
  // block = (clip-head, 64 consecutive diagonals); thread = (quarter of the queries, diagonal): consecutive threads read
  // consecutive addresses; the four partial sums of a diagonal are added in a fixed order through LDS
  __shared__ float sred[4][64];
  const int nrel = Sq + Sk - 1;
  const int bh = blockIdx.x, rl = threadIdx.x & 63, qg = threadIdx.x >> 6;
  const int rel = blockIdx.y * 64 + rl;
  const T* base = dS + (int64_t)bh * Sq * ldp;
  float acc = 0.f;
  if (rel < nrel) {
    const int off = rel - (Sq - 1);                              // key - query
    const int q_lo = max(0, -off), q_hi = min(Sq, Sk - off);     // q with 0 <= q + off < Sk
    const int per = (Sq + 3) / 4;
    const int a0 = max(q_lo, qg * per), a1 = min(q_hi, (qg + 1) * per);
    for (int q = a0; q < a1; ++q) acc += to_f32(base[(int64_t)q * ldp + q + off]);
  }
  sred[qg][rl] = acc;
  __syncthreads();
  if (qg == 0 && rel < nrel) part[(int64_t)bh * nrel + rel] = (sred[0][rl] + sred[1][rl]) + (sred[2][rl] + sred[3][rl]);
}
// stage 2: one block per (bucket, head): dtable[bucket][h] (+)= sum over clips b and the rels of that bucket of part[b][h][rel]
__global__ __launch_bounds__(256) void bias_bucket_kernel(const float* __restrict__ part, const int* __restrict__ bucket_of_rel,
                                                          float* __restrict__ dtable, int B, int H, int nrel, int accumulate) {
  __shared__ float sred[256];
  const int bk = blockIdx.x / H, hh = blockIdx.x - bk * H;
  float acc = 0.f;
  for (int rel = threadIdx.x; rel < nrel; rel += 256)
    if (bucket_of_rel[rel] == bk)
      for (int b = 0; b < B; ++b) acc += part[((int64_t)b * H + hh) * nrel + rel];
  sred[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) sred[threadIdx.x] += sred[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) dtable[blockIdx.x] = accumulate ? dtable[blockIdx.x] + sred[0] : sred[0];
}

// stage 1b for the per-stripe diagonal sums of attn_stripe_kernel: part2 [B*H][stripes][Sk + 31] -> tmp [H][nrel], summed over
// clips and stripes in a fixed order; the diagonal with global index rel (= key - query + Sq - 1) is entry
// rel + 31 + 32 s - (Sq - 1) of stripe s.  Block = (head, 64 consecutive rels) x 16 clip groups; bias_bucket_kernel (B = 1) follows.
__global__ __launch_bounds__(1024) void bias_stripes_sum_kernel(const float* __restrict__ part2_all, float* __restrict__ tmp_all, int B, int H, int stripes,
                                                                int Sq, int Sk, int64_t layer_stride) {
  const float* part2 = part2_all + (int64_t)blockIdx.z * layer_stride;        // blockIdx.z: layer (grouped mode: all layers of a stack in one launch)
  float* tmp = tmp_all + (int64_t)blockIdx.z * H * (Sq + Sk - 1);
  __shared__ float sred[16][64];
  const int nrel = Sq + Sk - 1, dl = Sk + 31;
  const int hh = blockIdx.x, rl = threadIdx.x & 63, bg = threadIdx.x >> 6;     // sixteen clip groups: at 16 clips one clip x all stripes per thread,
  const int rel = blockIdx.y * 64 + rl;                                        // loads unconditional (clamped) so that they are all in flight at once
  float acc = 0.f;
  if (rel < nrel) {
    const int per = (B + 15) / 16;
    for (int b = bg * per; b < min(B, (bg + 1) * per); ++b) {
      const float* pb = part2 + ((int64_t)b * H + hh) * stripes * dl;
#pragma unroll 3
      for (int sidx = 0; sidx < stripes; ++sidx) {
        const int loc = rel + 31 + 32 * sidx - (Sq - 1);
        const float v = pb[(int64_t)sidx * dl + min(max(loc, 0), dl - 1)];
        acc += (loc >= 0 && loc < dl) ? v : 0.f;
      }
    }
  }
  sred[bg][rl] = acc;
  __syncthreads();
  if (bg == 0 && rel < nrel) {
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) sum += sred[g][rl];
    tmp[(int64_t)hh * nrel + rel] = sum;
  }
}

// ---- gated GELU (hf: modeling_t5.py T5DenseGatedActDense: gelu_new(wi_0 x) * (wi_1 x)); ab = [a | b], [M, 2*dff] ----
// four consecutive columns per thread (8- / 16-byte accesses; the first form moved 2-byte elements: 11.4 / 16.5 us per launch)
template <typename T>
__global__ void gated_fwd_kernel(const T* __restrict__ ab, T* __restrict__ mid, int64_t M, int dff, DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = dff >> 2;
  const int64_t n4 = M * q, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i4 < n4; i4 += stride) {
    const int64_t row = i4 / q;
    const int c = (int)(i4 - row * q) * 4;
    const float4 a = st_load4<T>(ab + row * 2 * dff + c), b = st_load4<T>(ab + row * 2 * dff + dff + c);
    float v[4] = {gelu_new_t<T>(a.x) * b.x, gelu_new_t<T>(a.y) * b.y, gelu_new_t<T>(a.z) * b.z, gelu_new_t<T>(a.w) * b.w};      // (as the fused epilogue)
    if (thresh) {                                                       // hf: T5DenseGatedActDense dropout before wo
      const uint32_t kb = drop_keep4(key, row * dff + c, thresh);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ((kb >> e) & 1u) ? v[e] * scale : 0.f;
    }
    st_store4<T>(mid + row * dff + c, v[0], v[1], v[2], v[3]);
  }
}
template <typename T>
__global__ void gated_bwd_kernel(const T* __restrict__ ab, const T* __restrict__ dmid, T* __restrict__ dab, int64_t M, int dff, DropKey dk,
                                 uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = dff >> 2;
  const int64_t n4 = M * q, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i4 < n4; i4 += stride) {
    const int64_t row = i4 / q;
    const int c = (int)(i4 - row * q) * 4;
    const float4 a4 = st_load4<T>(ab + row * 2 * dff + c), b4 = st_load4<T>(ab + row * 2 * dff + dff + c), d4 = st_load4<T>(dmid + row * dff + c);
    const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    float dm[4] = {d4.x, d4.y, d4.z, d4.w}, da[4], db[4];
    const uint32_t kb = thresh ? drop_keep4(key, row * dff + c, thresh) : 0xFu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (thresh) dm[e] = ((kb >> e) & 1u) ? dm[e] * scale : 0.f;
      float g, dg;
      gelu_new_both_t<T>(a[e], &g, &dg);                    // (as the fused epilogue: bit-identical either way)
      da[e] = dm[e] * b[e] * dg;
      db[e] = dm[e] * g;
    }
    st_store4<T>(dab + row * 2 * dff + c, da[0], da[1], da[2], da[3]);
    st_store4<T>(dab + row * 2 * dff + dff + c, db[0], db[1], db[2], db[3]);
  }
}

// ---- RMSNorm backward (forward: y = w * x * r, r = rsqrt(mean(x^2) + eps), hf: modeling_t5.py:59-72) ----
// dx_out[row] = dx_res[row] (gradient arriving over the residual connection, may be null)
//             + r * (w o dy) - x * r^3 * mean(w o dy o x);   dw partial per block (fixed order), reduced by colsum_kernel.
constexpr int RN_BLOCKS = 512;      // (256 until round 3: four rows per wave at 16 clips, one memory round trip each — two now, both in flight from the start)
// out_t (optional): the gradient again in the GEMM-input type, through the dropout mask of the branch that consumes it next
// (what cvt_kernel / cvt_drop_kernel would produce in a launch of its own)
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, const float* __restrict__ dx_res,
                                                          float* __restrict__ dx_out, float* __restrict__ dw_part, int M, int d, float eps,
                                                          T* __restrict__ out_t, DropKey dk, uint32_t thresh, float scale,
                                                          DropKey dk_in, uint32_t thresh_in) {
  // thresh_in != 0: dy arrives through the dropout that follows THIS norm in the forward pass (the final norm of a stack): masked
  // as it is loaded — a drop_inplace launch over the whole buffer before
  const uint64_t key = (out_t && thresh) ? drop_site_key(dk) : 0ull;
  const uint64_t key_in = thresh_in ? drop_site_key(dk_in) : 0ull;
  extern __shared__ float red[];          // [4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dwacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dwacc[j] = 0.f;
  // d <= 512: a lane's (at most two) 4-column pieces of x, dy and the weight stay in registers between the statistics
  // and the output pass (the first form read them twice, two dependent round trips per row)
  float4 gv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) gv[j] = (lane * 4 + 256 * j < d) ? *reinterpret_cast<const float4*>(w + lane * 4 + 256 * j) : make_float4(0, 0, 0, 0);
  // A wave walks its rows (4 at M = 4 176) with the NEXT row's loads in flight while it reduces the current one: the first form
  // paid a dependent load -> reduce -> store chain per row (10 us per launch, 32 launches per step).
  const int stride = gridDim.x * 4;
  float4 xv[2], dv[2], rv[2], xn[2], dn[2], rn[2];
  const float* const res_src = dx_res ? dx_res : x;     // (never predicated: clamped addresses, unconditional loads, then a select)
  auto load_row = [&](int row, float4 (&xa)[2], float4 (&da)[2], float4 (&ra)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = lane * 4 + 256 * j;
      const bool in = col < d && row < M;
      const int64_t at = (int64_t)min(row, M - 1) * d + min(col, d - 4);
      const float4 a = *reinterpret_cast<const float4*>(x + at), b = *reinterpret_cast<const float4*>(dy + at),
                   c4 = *reinterpret_cast<const float4*>(res_src + at);
      const float4 z = make_float4(0, 0, 0, 0);
      xa[j] = in ? a : z;
      float4 bm = b;
      if (thresh_in) {
        const uint32_t kb = drop_keep4(key_in, at, thresh_in);
        bm = make_float4((kb & 1u) ? b.x * scale : 0.f, (kb & 2u) ? b.y * scale : 0.f, (kb & 4u) ? b.z * scale : 0.f, (kb & 8u) ? b.w * scale : 0.f);
      }
      da[j] = in ? bm : z;
      ra[j] = (in && dx_res) ? c4 : z;
    }
  };
  int row = blockIdx.x * 4 + wave;
  if (row < M) load_row(row, xv, dv, rv);
  for (; row < M; row += stride) {
    load_row(row + stride, xn, dn, rn);                 // (past the end: no loads, zeros)
    float ss = 0.f, c = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {       // same association as the two-pass form: per piece, then pieces in order
      ss += xv[j].x * xv[j].x + xv[j].y * xv[j].y + xv[j].z * xv[j].z + xv[j].w * xv[j].w;
      c += gv[j].x * dv[j].x * xv[j].x + gv[j].y * dv[j].y * xv[j].y + gv[j].z * dv[j].z * xv[j].z + gv[j].w * dv[j].w * xv[j].w;
    }
    ss = wave_sum(ss);
    c = wave_sum(c);
    const float r = rsqrtf(ss / (float)d + eps);
    const float k2 = r * r * r * c / (float)d;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = lane * 4 + 256 * j;
      if (col < d) {
        float4 o = make_float4(r * gv[j].x * dv[j].x - xv[j].x * k2, r * gv[j].y * dv[j].y - xv[j].y * k2, r * gv[j].z * dv[j].z - xv[j].z * k2,
                               r * gv[j].w * dv[j].w - xv[j].w * k2);
        if (dx_res) { o.x += rv[j].x; o.y += rv[j].y; o.z += rv[j].z; o.w += rv[j].w; }
        *reinterpret_cast<float4*>(dx_out + (int64_t)row * d + col) = o;
        if (out_t) {
          const int64_t at = (int64_t)row * d + col;
          float v[4] = {o.x, o.y, o.z, o.w};
          if (thresh) {
            const uint32_t kb = drop_keep4(key, at, thresh);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((kb >> e) & 1u) ? v[e] * scale : 0.f;
          }
          st_store4<T>(out_t + at, v[0], v[1], v[2], v[3]);
        }
        dwacc[4 * j + 0] += dv[j].x * xv[j].x * r; dwacc[4 * j + 1] += dv[j].y * xv[j].y * r;
        dwacc[4 * j + 2] += dv[j].z * xv[j].z * r; dwacc[4 * j + 3] += dv[j].w * xv[j].w * r;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { xv[j] = xn[j]; dv[j] = dn[j]; rv[j] = rn[j]; }
  }
  int j = 0;
  for (int col = lane * 4; col < d; col += 256, ++j) {
    red[wave * d + col + 0] = dwacc[4 * j + 0]; red[wave * d + col + 1] = dwacc[4 * j + 1];
    red[wave * d + col + 2] = dwacc[4 * j + 2]; red[wave * d + col + 3] = dwacc[4 * j + 3];
  }
  __syncthreads();
  for (int col = threadIdx.x; col < d; col += 256)
    dw_part[(int64_t)blockIdx.x * d + col] = red[col] + red[d + col] + red[2 * d + col] + red[3 * d + col];
}
// out[col] (+)= sum over `parts` rows of part[p][col]   (second pass of every column reduction): 32 columns per block,
// 8 row groups per column, each walking its rows 8 at a time (independent loads: the first form issued one dependent
// round trip per row, 64 in a row = 16.7 us per launch), summed through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int parts, int d, int accumulate) {
  __shared__ float sred[8][32];
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cx;
  float acc = 0.f;
  if (col < d) {
    int p = py;
    for (; p + 56 < parts; p += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + 8 * u) * d + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; p < parts; p += 8) acc += part[(int64_t)p * d + col];
  }
  sred[py][cx] = acc;
  __syncthreads();
  if (py == 0 && col < d) {
    float v = sred[0][cx];
#pragma unroll
    for (int u = 1; u < 8; ++u) v += sred[u][cx];
    out[col] = accumulate ? out[col] + v : v;
  }
}

// every RMSNorm weight gradient of a step in one launch: partial image j ([parts][d]) -> out_base + off[j]
__global__ __launch_bounds__(256) void colsum_group_kernel(const float* __restrict__ part_all, const int64_t* __restrict__ offs, float* __restrict__ out_base,
                                                           int parts, int d) {
  // block = 64 columns (16 lanes x 16 bytes: 256-byte row pieces) x 16 row groups, eight rows in flight per thread; the first form read
  // 128-byte pieces one 4-byte element per lane (54 us for the step's 17 MB of partial rows)
  __shared__ float4 sred[16][16];
  const int j = blockIdx.y;
  const float* part = part_all + (int64_t)j * parts * d;
  const int cx = threadIdx.x & 15, py = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cx;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < d) {
    for (int p = py; p < parts; p += 128) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (int64_t)min(p + 16 * u, parts - 1) * d + col);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p + 16 * u < parts) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  sred[py][cx] = acc;
  __syncthreads();
  if (py == 0 && col < d) {
    float4 v = sred[0][cx];
#pragma unroll
    for (int u = 1; u < 16; ++u) { const float4 o = sred[u][cx]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    float* out = out_base + offs[j] + col;
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
}

// ---- cross entropy (mean over labels != -100, hf: modeling_t5.py:1049-1054) + gradient of the logits ----
// one wave per row: row_loss[row] = logsumexp - logit[label] (0 if ignored); dlogits (T) = (softmax - onehot) / n_valid
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 const float* __restrict__ inv_n, float* __restrict__ row_loss, T* __restrict__ dlogits,
                                                 int M, int V, int ldd) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* lg = logits + (int64_t)row * V;
  const int64_t lab = labels[row];
  const bool valid = lab >= 0 && lab < V;          // -100 (and anything out of range) is ignored, as torch's ignore_index
  float mx = -1e30f;
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg[v]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int v = lane; v < V; v += 64) sum += expf(lg[v] - mx);
  sum = wave_sum(sum);
  const float lse = mx + logf(sum), scale = inv_n[0];
  if (lane == 0) row_loss[row] = valid ? lse - lg[lab] : 0.f;
  T* dl = dlogits + (int64_t)row * ldd;
  for (int v = lane; v < ldd; v += 64) {
    float gval = 0.f;
    if (valid && v < V) gval = (expf(lg[v] - lse) - (v == (int)lab ? 1.f : 0.f)) * scale;
    dl[v] = from_f32<T>(gval);
  }
}
// loss = inv_n * sum(row_loss)  (single block, fixed order)
__global__ void loss_reduce_kernel(const float* __restrict__ row_loss, int M, const float* __restrict__ inv_n, float* __restrict__ loss) {
  __shared__ float part[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) acc += row_loss[i];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0] * inv_n[0];
}

// ---- embeddings ----
// conditioning rows of the encoder input: x[b, i, :] = table_i[idx[b, i], :]   (ref: music2midi/input.py:57-59)
__global__ void cond_gather_kernel(const float* __restrict__ params, const int64_t* __restrict__ tab_off, const int* __restrict__ tab_rows,
                                   int n_tab, const int64_t* __restrict__ idx, float* __restrict__ x, int S, int d) {
  const int b = blockIdx.x / n_tab, i = blockIdx.x - b * n_tab;
  int64_t id = idx[(int64_t)b * n_tab + i];
  if (id < 0 || id >= tab_rows[i]) id = 0;
  const float* src = params + tab_off[i] + id * d;
  float* dst = x + ((int64_t)b * S + i) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = src[c];
}
// gradient of an embedding table: one block per table row v, G[v] = sum of dx[row] over the activation rows whose id is v, in a fixed
// order.  Pass 1 scans the ids 256 at a time and lists the matching rows in LDS (ballot + prefix counts keep them in order); pass 2
// gives every fourth listed row to one wave, eight rows in flight per wave (a lane reads 16 bytes of each), and the four waves' partial
// rows meet in LDS in wave order.  (Round 2 walked the matches one after another inside the scan: a hot row — the pad / start token of
// every clip's ragged tail, ~700 activation rows at 16 clips — paid one memory latency per match, 140 us of the step.)
// ids[i * id_stride + id_off] is the id of activation row (i * x_row_stride + x_row_off).
// thresh != 0: dx arrives through the dropout on the embeddings (hf: T5Stack dropout(inputs_embeds)); only the rows that are READ
// here are masked, as they are read.
// NW waves per block: 16 for the shared embedding (round 4: the hot row's ~700 matches are 6 batches of loads per wave instead of 22;
// the launch's time IS that row), 4 for the small conditioning tables.
constexpr int EMB_ROWS_IN_FLIGHT = 8;
template <int NW>
__global__ __launch_bounds__(64 * NW) void embed_bwd_kernel(const int64_t* __restrict__ ids, int n_ids, int id_stride, int id_off,
                                                        const float* __restrict__ dx, int64_t x_row_stride, int64_t x_row_off,
                                                        float* __restrict__ gtab, int d, int pad_to_zero_id, int V, DropKey dk, uint32_t thresh,
                                                        float scale) {
  extern __shared__ __align__(16) int emb_smem[];
  int* list = emb_smem;                                          // [round_up_4(n_ids)]
  float* part = reinterpret_cast<float*>(list + ((n_ids + 3) & ~3));      // [NW][256]: one column block of the waves' sums
  __shared__ int wave_cnt[NW];
  const uint64_t dkey = thresh ? drop_site_key(dk) : 0ull;
  const int v = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cnt = 0;
  for (int i0 = 0; i0 < n_ids; i0 += 64 * NW) {
    const int i = i0 + threadIdx.x;
    bool match = false;
    if (i < n_ids) {
      int64_t id = ids[(int64_t)i * id_stride + id_off];
      if (id < 0 || id >= V) id = pad_to_zero_id;
      match = id == v;
    }
    const unsigned long long m = __ballot(match);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = cnt, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = wave_cnt[w]; if (w < wave) base += c; all += c; }
    if (match) list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    cnt += all;
    __syncthreads();
  }
  for (int c0 = 0; c0 < d; c0 += 256) {                          // column blocks of 256: lane -> columns c0 + 4 lane .. + 3
    const int c = c0 + 4 * lane;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < d) {
      for (int k0 = wave; k0 < cnt; k0 += NW * EMB_ROWS_IN_FLIGHT) {
        float4 rows[EMB_ROWS_IN_FLIGHT];
        uint32_t keep[EMB_ROWS_IN_FLIGHT];
#pragma unroll
        for (int rr = 0; rr < EMB_ROWS_IN_FLIGHT; ++rr) {
          const int k = k0 + NW * rr;
          rows[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
          keep[rr] = 0xFu;
          if (k < cnt) {
            const int64_t roff = ((int64_t)list[k] * x_row_stride + x_row_off) * d + c;
            rows[rr] = *reinterpret_cast<const float4*>(dx + roff);
            if (thresh) keep[rr] = drop_keep4(dkey, roff, thresh);
          }
        }
#pragma unroll
        for (int rr = 0; rr < EMB_ROWS_IN_FLIGHT; ++rr) {
          const float sc = thresh ? scale : 1.0f;
          acc.x += (keep[rr] & 1u) ? rows[rr].x * sc : 0.f;
          acc.y += (keep[rr] & 2u) ? rows[rr].y * sc : 0.f;
          acc.z += (keep[rr] & 4u) ? rows[rr].z * sc : 0.f;
          acc.w += (keep[rr] & 8u) ? rows[rr].w * sc : 0.f;
        }
      }
    }
    *reinterpret_cast<float4*>(part + wave * 256 + 4 * lane) = acc;
    __syncthreads();
    const int cc = c0 + threadIdx.x;
    if (threadIdx.x < 256 && cc < d) {                           // the waves' partial rows in wave order: a fixed order of additions
      float sum = part[threadIdx.x];
#pragma unroll
      for (int w = 1; w < NW; ++w) sum += part[256 * w + threadIdx.x];
      gtab[(int64_t)v * d + cc] = sum;
    }
    __syncthreads();
  }
}
constexpr int EMB_NW_SHARED = 16;      // waves per block of the shared-embedding launch
static size_t embed_bwd_smem(int n_ids, int nw = EMB_NW_SHARED) { return (size_t)((n_ids + 3) & ~3) * 4 + (size_t)nw * 256 * 4; }
// Final norm of a stack with the dropout that follows it (hf: T5Stack dropout(final_layer_norm(x))): y = T(x * rstd * w), then the
// mask on the ROUNDED value, as the separate in-place pass did — one wave per row, one launch instead of two.
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_drop_kernel(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ out, int M, int d,
                                                           float eps, DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * d;
  float ss = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)d + eps);
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const float4 gw = *reinterpret_cast<const float4*>(w + c);
    float y[4] = {to_f32(from_f32<T>(gw.x * (v.x * rstd))), to_f32(from_f32<T>(gw.y * (v.y * rstd))), to_f32(from_f32<T>(gw.z * (v.z * rstd))),
                  to_f32(from_f32<T>(gw.w * (v.w * rstd)))};
    const int64_t at = (int64_t)row * d + c;
    const uint32_t kb = drop_keep4(key, at, thresh);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = ((kb >> e) & 1u) ? y[e] * scale : 0.f;
    st_store4<T>(out + at, y[0], y[1], y[2], y[3]);
  }
}
// Encoder input of a pass in place: rows [0, n_tab) of every clip from the conditioning tables (ref: music2midi/input.py:57-59), then
// the dropout on the embeddings over ALL rows (cond_gather_kernel + drop_inplace_kernel before).
__global__ __launch_bounds__(256) void enc_input_kernel(const float* __restrict__ params, const int64_t* __restrict__ tab_off, const int* __restrict__ tab_rows,
                                                        int n_tab, const int64_t* __restrict__ idx, float* __restrict__ x, int B, int S, int d, DropKey dk,
                                                        uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  const int q = d >> 2;
  const int64_t n4 = (int64_t)B * S * q, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
    const int64_t row = i4 / q;
    const int c = (int)(i4 - row * q) * 4, b = (int)(row / S), s = (int)(row - (int64_t)b * S);
    float4 v;
    if (s < n_tab) {
      int64_t id = idx[(int64_t)b * n_tab + s];
      if (id < 0 || id >= tab_rows[s]) id = 0;
      v = *reinterpret_cast<const float4*>(params + tab_off[s] + id * d + c);
    } else {
      if (!thresh) continue;                                 // nothing to do for a feature row
      v = *reinterpret_cast<const float4*>(x + row * d + c);
    }
    if (thresh) {
      const uint32_t kb = drop_keep4(key, row * d + c, thresh);
      v = make_float4((kb & 1u) ? v.x * scale : 0.f, (kb & 2u) ? v.y * scale : 0.f, (kb & 4u) ? v.z * scale : 0.f, (kb & 8u) ? v.w * scale : 0.f);
    }
    *reinterpret_cast<float4*>(x + row * d + c) = v;
  }
}
// Decoder input embedding of the teacher-forced pass with its dropout: x[row] = dropout(table[ids[row]])
__global__ __launch_bounds__(256) void embed_rows_drop_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table, float* __restrict__ x, int M,
                                                              int d, int V, int pad_id, DropKey dk, uint32_t thresh, float scale) {
  const uint64_t key = thresh ? drop_site_key(dk) : 0ull;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int tok = (int)ids[row];
  if (tok < 0 || tok >= V) tok = pad_id;
  for (int c = lane * 4; c < d; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(table + (int64_t)tok * d + c);
    const uint32_t kb = drop_keep4(key, (int64_t)row * d + c, thresh);
    v = make_float4((kb & 1u) ? v.x * scale : 0.f, (kb & 2u) ? v.y * scale : 0.f, (kb & 4u) ? v.z * scale : 0.f, (kb & 8u) ? v.w * scale : 0.f);
    *reinterpret_cast<float4*>(x + (int64_t)row * d + c) = v;
  }
}

// the three inputs of a pass into the trainer's own buffers (what a captured graph reads): x as 16-byte pieces, labels, conditioning indices
__global__ void stage_inputs_kernel(const float4* __restrict__ x, float4* __restrict__ x_dst, int64_t n4, const int64_t* __restrict__ lab,
                                    int64_t* __restrict__ lab_dst, int64_t nl, const int64_t* __restrict__ cond, int64_t* __restrict__ cond_dst, int64_t nc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, n = n4 + nl + nc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (i < n4) x_dst[i] = x[i];
    else if (i < n4 + nl) lab_dst[i - n4] = lab[i - n4];
    else cond_dst[i - n4 - nl] = cond[i - n4 - nl];
  }
}

// fp32 c = a + b (either may be null -> treated as 0)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) c[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f);
}

// ============================================================ Adafactor ====
// transformers.optimization.Adafactor as ref: music2midi/model.py:27-30 builds it:
//   Adafactor(params, lr=None, eps=(1e-30, 1e-3), clip_threshold=1.0, decay_rate=-0.8, beta1=None, weight_decay=0.0,
//             scale_parameter=True, relative_step=True, warmup_init=True)
// per tensor p with gradient g at step t (1-based):
//   rho   = min(1e-6 * t, 1/sqrt(t))                       (relative step with warm-up init)
//   lr    = max(1e-3, rms(p)) * rho                        (scale_parameter)
//   b2    = 1 - t^-0.8
//   u     = g^2 + 1e-30
//   2-D:  R <- b2 R + (1-b2) mean_cols(u);  C <- b2 C + (1-b2) mean_rows(u);  upd = g * rsqrt(R / mean(R)) [row] * rsqrt(C) [col]
//   1-D:  V <- b2 V + (1-b2) u;             upd = g * rsqrt(V)
//   upd  /= max(1, rms(upd) / 1.0);   p <- p - lr * upd
// Three passes over (p, g) with the reductions between them; one launch per pass for ALL tensors (block -> (tensor,
// row block) through a table), every reduction in a fixed order.
constexpr int AF_ROWS = 16;      // rows of a matrix per block (32 until round 3: 235 registers in pass A, two blocks per CU; 16 rows run pass A / B / C in
                                 // 51 / 23 / 61 us instead of 64 / 31 / 69; 8 rows gain nothing more and cost pass A2 its partial sums)
// (vectors: one "row" of up to AF_VEC elements per block)

// pass A: per block: sum p^2, per-row sum of (g^2 + eps1) -> rowsum[tensor rows], per-block column partial sums
__global__ __launch_bounds__(256) void af_pass_a(const AfBlock* __restrict__ blocks, const AfTensor* __restrict__ tensors,
                                                 const float* __restrict__ P, const float* __restrict__ G, float* __restrict__ rowsum,
                                                 float* __restrict__ colpart, float* __restrict__ blk_p2) {
  __shared__ float sred[256];
  const AfBlock bk = blocks[blockIdx.x];
  const AfTensor t = tensors[bk.tensor];
  const float* p = P + t.offset;
  const float* g = G + t.offset;
  float p2 = 0.f;
  const int r1 = min(bk.row0 + AF_ROWS, t.rows);
  // thread tid owns columns tid, tid + 256, ... ; rows are walked in order -> fixed summation order per column.  ONE read of
  // g and p (the first form read g a second time for the row sums, one dependent load per row: 157 us per step): the per-row
  // partial sums of this thread's columns stay in registers and are reduced by wave, then over the four waves, in a fixed order
  __shared__ float rred[4][AF_ROWS];
  float rs[AF_ROWS];
#pragma unroll
  for (int j = 0; j < AF_ROWS; ++j) rs[j] = 0.f;
  for (int c = threadIdx.x; c < t.cols; c += 256) {
    float gv[AF_ROWS], pv[AF_ROWS];
#pragma unroll
    for (int j = 0; j < AF_ROWS; ++j) {
      const int r = min(bk.row0 + j, r1 - 1);          // clamped: every load of the tile is in flight at once
      gv[j] = g[(int64_t)r * t.cols + c];
      pv[j] = p[(int64_t)r * t.cols + c];
    }
    float cs = 0.f;
#pragma unroll
    for (int j = 0; j < AF_ROWS; ++j) {
      if (bk.row0 + j < r1) {
        const float q = gv[j] * gv[j] + 1e-30f;
        cs += q;
        rs[j] += q;
        p2 += pv[j] * pv[j];
      }
    }
    colpart[bk.col_off + c] = cs;
  }
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < AF_ROWS; ++j) {
      const float v = wave_sum(rs[j]);
      if (lane == 0) rred[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < AF_ROWS && bk.row0 + (int)threadIdx.x < r1)
      rowsum[t.row_off + bk.row0 + threadIdx.x] = (rred[0][threadIdx.x] + rred[1][threadIdx.x]) + (rred[2][threadIdx.x] + rred[3][threadIdx.x]);
  }
  sred[threadIdx.x] = p2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) blk_p2[blockIdx.x] = sred[0];
}

// pass A2: one block per tensor: finish rms(p), update the factored second moments, derive the row / column factors
__global__ __launch_bounds__(256) void af_pass_a2(const AfTensor* __restrict__ tensors, const float* __restrict__ rowsum,
                                                  const float* __restrict__ colpart, const float* __restrict__ blk_p2,
                                                  float* __restrict__ state, float* __restrict__ rfac, float* __restrict__ cfac,
                                                  float* __restrict__ tstat, float beta2t) {
  __shared__ float sred[256];
  const AfTensor t = tensors[blockIdx.x];
  float acc = 0.f;
  for (int b = threadIdx.x; b < t.nblocks; b += 256) acc += blk_p2[t.block0 + b];
  sred[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s]; __syncthreads(); }
  const float p_rms = sqrtf(sred[0] / (float)((int64_t)t.rows * t.cols));
  __syncthreads();
  float* R = state + t.state_off;               // [rows] (matrix) or [cols] full second moment (vector: rows == 1)
  float* C = R + t.rows;                        // [cols] (matrix only)
  if (t.rows > 1) {
    // rows
    float racc = 0.f;
    for (int r = threadIdx.x; r < t.rows; r += 256) {
      const float v = beta2t * R[r] + (1.f - beta2t) * (rowsum[t.row_off + r] / (float)t.cols);
      R[r] = v;
      racc += v;
    }
    sred[threadIdx.x] = racc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s]; __syncthreads(); }
    const float rmean = sred[0] / (float)t.rows;
    __syncthreads();
    for (int r = threadIdx.x; r < t.rows; r += 256) rfac[t.row_off + r] = rsqrtf(R[r] / rmean);
    for (int c = threadIdx.x; c < t.cols; c += 256) {
      float cs = 0.f;
      int b = 0;
      for (; b + 8 <= t.nblocks; b += 8) {            // eight partial rows in flight (the plain loop paid a memory round trip per row block: 28 us per step)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = colpart[t.col_off + (int64_t)(b + u) * t.cols + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) cs += v[u];
      }
      for (; b < t.nblocks; ++b) cs += colpart[t.col_off + (int64_t)b * t.cols + c];
      const float v = beta2t * C[c] + (1.f - beta2t) * (cs / (float)t.rows);
      C[c] = v;
      cfac[t.cfac_off + c] = rsqrtf(v);
    }
  } else {
    for (int c = threadIdx.x; c < t.cols; c += 256) {
      const float v = beta2t * R[c] + (1.f - beta2t) * colpart[t.col_off + c];     // one block, one row: colpart = g^2 + eps
      R[c] = v;
      cfac[t.cfac_off + c] = rsqrtf(v);
    }
    if (threadIdx.x == 0) rfac[t.row_off] = 1.0f;
  }
  if (threadIdx.x == 0) tstat[2 * blockIdx.x] = p_rms;
}

// pass B: per block sum of upd^2, upd = g * rfac[row] * cfac[col]
__global__ __launch_bounds__(256) void af_pass_b(const AfBlock* __restrict__ blocks, const AfTensor* __restrict__ tensors,
                                                 const float* __restrict__ G, const float* __restrict__ rfac, const float* __restrict__ cfac,
                                                 float* __restrict__ blk_u2) {
  __shared__ float sred[256];
  const AfBlock bk = blocks[blockIdx.x];
  const AfTensor t = tensors[bk.tensor];
  const float* g = G + t.offset;
  const int r1 = min(bk.row0 + AF_ROWS, t.rows);
  float u2 = 0.f;
  for (int c = threadIdx.x; c < t.cols; c += 256) {
    const float cf = cfac[t.cfac_off + c];
    float gv[AF_ROWS];
#pragma unroll
    for (int j = 0; j < AF_ROWS; ++j) gv[j] = g[(int64_t)min(bk.row0 + j, r1 - 1) * t.cols + c];
#pragma unroll
    for (int j = 0; j < AF_ROWS; ++j) {
      if (bk.row0 + j < r1) {
        const float u = gv[j] * rfac[t.row_off + bk.row0 + j] * cf;
        u2 += u * u;
      }
    }
  }
  sred[threadIdx.x] = u2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) blk_u2[blockIdx.x] = sred[0];
}
// pass B2: per tensor: step size = lr / max(1, rms(upd))
__global__ __launch_bounds__(256) void af_pass_b2(const AfTensor* __restrict__ tensors, const float* __restrict__ blk_u2,
                                                  float* __restrict__ tstat, float rho) {
  __shared__ float sred[256];
  const AfTensor t = tensors[blockIdx.x];
  float acc = 0.f;
  for (int b = threadIdx.x; b < t.nblocks; b += 256) acc += blk_u2[t.block0 + b];
  sred[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float u_rms = sqrtf(sred[0] / (float)((int64_t)t.rows * t.cols));
    const float lr = fmaxf(1e-3f, tstat[2 * blockIdx.x]) * rho;
    tstat[2 * blockIdx.x + 1] = lr / fmaxf(1.0f, u_rms);
  }
}
// pass C: p -= step * g * rfac[row] * cfac[col]
__global__ __launch_bounds__(256) void af_pass_c(const AfBlock* __restrict__ blocks, const AfTensor* __restrict__ tensors,
                                                 float* __restrict__ P, const float* __restrict__ G, const float* __restrict__ rfac,
                                                 const float* __restrict__ cfac, const float* __restrict__ tstat) {
  const AfBlock bk = blocks[blockIdx.x];
  const AfTensor t = tensors[bk.tensor];
  float* p = P + t.offset;
  const float* g = G + t.offset;
  const float step = tstat[2 * bk.tensor + 1];
  const int r1 = min(bk.row0 + AF_ROWS, t.rows);
  // eight rows of a column in flight per thread (a row-by-row loop kept one load pair per thread in flight: 78 us for the pass's 366 MB; 70 us this way, same arithmetic)
  for (int c = threadIdx.x; c < t.cols; c += 256) {
    const float cf = cfac[t.cfac_off + c];
    for (int r = bk.row0; r < r1; r += 8) {
      float gv[8], pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t at = (int64_t)min(r + j, r1 - 1) * t.cols + c;
        gv[j] = g[at];
        pv[j] = p[at];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r + j < r1) p[(int64_t)(r + j) * t.cols + c] = pv[j] - gv[j] * (rfac[t.row_off + r + j] * step) * cf;
    }
  }
}

int launch_adafactor(const AfPlan& pl, float* P, const float* G, int step, hipStream_t st) {
  const double t = (double)step;
  const float beta2t = (float)(1.0 - pow(t, -0.8));
  const double rho_d = fmin(1e-6 * t, 1.0 / sqrt(t));
  const float rho = (float)rho_d;
  hipLaunchKernelGGL(af_pass_a, dim3(pl.n_blocks), dim3(256), 0, st, pl.blocks, pl.tensors, P, G, pl.rowsum, pl.colpart, pl.blk_a);
  hipLaunchKernelGGL(af_pass_a2, dim3(pl.n_tensors), dim3(256), 0, st, pl.tensors, pl.rowsum, pl.colpart, pl.blk_a, pl.state, pl.rfac,
                     pl.cfac, pl.tstat, beta2t);
  hipLaunchKernelGGL(af_pass_b, dim3(pl.n_blocks), dim3(256), 0, st, pl.blocks, pl.tensors, G, pl.rfac, pl.cfac, pl.blk_b);
  hipLaunchKernelGGL(af_pass_b2, dim3(pl.n_tensors), dim3(256), 0, st, pl.tensors, pl.blk_b, pl.tstat, rho);
  hipLaunchKernelGGL(af_pass_c, dim3(pl.n_blocks), dim3(256), 0, st, pl.blocks, pl.tensors, P, G, pl.rfac, pl.cfac, pl.tstat);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}


// declared in enc_kernels.hip (the forward RMSNorm is the inference kernel)
int launch_rmsnorm(int precision, const float* x, const float* w, void* out, int M, int d, float eps, hipStream_t st);
int launch_embed_rows(const int64_t* ids, const float* table, float* x, int M, int d, int V, int pad_id, hipStream_t st);

}  // namespace m2m

// ============================================================ trainer ====
using namespace m2m;

namespace {

struct TensorDesc {
  std::string name;        // state-dict key under `model.` of the reference's LightningModule
  int64_t off;             // floats into the flat buffers
  int rows, cols;          // 1-D tensors: rows = length, cols = 0
};

struct EncOff { int64_t ln0, qkv, o, ln1, wi, wo; };
struct DecOff { int64_t ln0, qkv, o, ln1, cq, ckv, co, ln2, wi, wo; };

// build the relative-position tables for one (Sq, Sk) geometry: bucket of every (key - query) offset
std::vector<int> bucket_table(const m2m_t5_geometry& g, int Sq, int Sk, bool bidirectional) {
  std::vector<int> t((size_t)Sq + Sk - 1);
  for (int i = 0; i < Sq + Sk - 1; ++i) t[i] = m2m_rel_bucket(i - (Sq - 1), bidirectional ? 1 : 0, g.num_buckets, g.max_distance);
  return t;
}

}  // namespace

struct m2m_trainer {
  m2m_t5_geometry g;
  int precision, inner, n_cond;
  size_t es;
  int max_batch, max_enc, max_dec;
  std::vector<int> cond_rows;
  std::vector<TensorDesc> tensors;
  int64_t n_floats = 0;
  // offsets
  int64_t o_shared, o_lm, o_erb, o_drb, o_eln, o_dln;
  std::vector<EncOff> enc;
  std::vector<DecOff> dec;
  std::vector<int64_t> o_cond;
  // device memory owned by the trainer
  unsigned char* arena = nullptr;
  int64_t arena_bytes = 0;
  void* Wc = nullptr;                    // T copy of the parameters (bf16 mode; fp32 mode reads the master buffer)
  void* Wil = nullptr;                   // bf16 mode: the gate-pair matrices again, rows interleaved (same offsets; see weights_transpose_kernel)
  void* WT = nullptr;                    // every weight matrix TRANSPOSED, in T, at the same offsets (dX = dY . W as an NT product)
  void *tA = nullptr, *tB = nullptr;     // transposed activations of the current dW product: [features][Mp]
  float* kpart = nullptr;                // split-K partial tiles
  int64_t kpart_floats = 0;
  void* wt_blocks = nullptr;             // WtBlock table on the device
  int n_wt_blocks = 0;
  // activations (pointers into the arena)
  std::vector<float*> xe, xd;            // residual streams: 2*Le + 1 and 3*Ld + 1 buffers
  std::vector<void*> h0e, h1e, qkve, Pe, aoe, abe, mide;
  std::vector<void*> h0d, h1d, h2d, qkvd, Pd, aod, cqd, ckvd, Pcd, aocd, abd, midd;
  std::vector<float*> lse_e, lse_d, lse_c;   // per attention layer: row log-sum-exp [B*H][queries] of the whole-head attention kernels (attn_train.hip)
  std::vector<void*> kte, ktd, ktc;      // per layer: K and V transposed, [2][B*H][64][Sp] (kv_transpose_kernel), for the fused stripe products
  void *hE = nullptr, *hD = nullptr;
  float *logits = nullptr, *sc = nullptr, *dxa = nullptr, *dxb = nullptr, *dh = nullptr, *dhE = nullptr, *dw_part = nullptr,
        *row_loss = nullptr, *inv_n = nullptr, *drel = nullptr, *etab = nullptr, *dtab = nullptr;
  void *dlog = nullptr, *dxT = nullptr, *dmid = nullptr, *dab = nullptr, *dO = nullptr, *dqkv = nullptr, *dS = nullptr, *dcq = nullptr,
       *dckv = nullptr;
  int64_t* dec_in = nullptr;
  int *ebucket = nullptr, *dbucket = nullptr, *counter = nullptr;
  int64_t* cond_off_dev = nullptr;
  int* cond_rows_dev = nullptr;
  int64_t* pads_dev = nullptr;           // (offset, count) of the alignment gaps of the flat layout
  int n_pads = 0;
  int tab_S = -1, tab_L = -1;            // geometry the bucket tables on the device were built for
  // fp8 mode (M2M_PREC_FP8): storage type stays bf16, the projection products run on MXFP8 (mx8.hip)
  bool fp8 = false;
  // which projection products run on MXFP8 (M2M_FP8_PARTS = subset of fwd,dx,dw).  Default: forward and dX; the weight gradients
  // take the grouped bf16 launch — on fp8 they need two transposing quantiser launches + a split-K product + a reduce EACH
  // (16 clips: 8.9 ms per step against 7.5 ms), for the least accuracy-critical third of the products
  bool fp8_fwd = true, fp8_dx = true, fp8_dw = false;
  int grad_fmt = 0;                      // element format of the gradient operands: 0 = e4m3 (default), 1 = e5m2 (M2M_FP8_GRAD=e5m2)
  struct LinW { int64_t off; int N, K, Np; int64_t q, qs, qt, qts; };
  std::vector<LinW> lin;                 // every projection matrix (fused groups), by parameter offset
  uint8_t* w8 = nullptr;                 // fp8 weights, both layouts, + scales
  void* w8_tiles = nullptr;
  int n_w8_tiles = 0;
  uint8_t *q8a = nullptr, *s8a = nullptr, *q8ta = nullptr, *s8ta = nullptr, *q8tb = nullptr, *s8tb = nullptr;
  // dropout (hf T5Config.dropout_rate; the reference trains in model.train() mode, ref train.py:33): off unless set
  float drop_p = 0.f, drop_scale = 1.f;
  uint32_t drop_thresh = 0;
  uint64_t drop_seed = 0;
  uint64_t *step_key_dev = nullptr, *step_ctr_dev = nullptr;   // key of the current pass / passes since set_dropout (device words:
                                                               // a captured graph advances them itself, see train_prologue_kernel)
  // The operands the weight-gradient products read (dxT, dab, dqkv, dcq, dckv) live in per-sub-layer buffers (rings, one
  // entry per use in a pass): the products of a whole step are issued as ONE grouped launch after the backward pass
  // (or, in fp8 mode, on the side stream while the main stream moves on), so nothing may be overwritten before.
  enum { K_DXT = 0, K_DAB, K_DQKV, K_DCQ, K_DCKV, K_KINDS };
  std::vector<void*> ring[K_KINDS];
  bool use_group = true;                 // one grouped weight-gradient launch per step (bf16 / fp32 modes)
  void* dw_probs_dev = nullptr;          // DwProb tables on the device: [N_SLOTS + 1 table slots][2 phases][128 entries] (see GraphSlot)
  int dw_tiles = 0;
  int64_t drel_slot_floats = 0;          // one self-attention layer's per-stripe diagonal sums (t->drel holds Le + Ld of them, then the scratch)
  int64_t* norm_offs_dev = nullptr;      // parameter offsets of the RMSNorm weights, in the order the backward pass meets them: [N_SLOTS + 1][2][32]
  // data-parallel overlap (m2m_trainer_set_sync_stream): the backward pass is issued in two parts — decoder side, then encoder side —
  // and `sync_stream` is released (ev_mid) as soon as the decoder-side gradients are final, so their all-reduce runs beside the
  // encoder backward.  The layout keeps them in two flat ranges: [0, o_erb) (shared embedding, lm_head) and the decoder blocks.
  hipStream_t sync_stream = nullptr;
  hipEvent_t ev_mid = nullptr;
  int64_t dec_begin = 0, dec_end = 0;
  // streams / graph of the step (trainer-owned: the caller's stream may be the legacy default stream, which cannot capture)
  hipStream_t s_main = nullptr, s_side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_ready = nullptr, ev_free[2] = {nullptr, nullptr};
  bool use_side = true, use_graph = true;
  int64_t* labels_buf = nullptr;
  int64_t* cond_buf = nullptr;
  float* loss_dev = nullptr;
  struct GraphKey {
    const float* P = nullptr; float* G = nullptr; int B = 0, S = 0, L = 0; uint32_t thresh = 0; uint64_t seed = 0; bool split = false;
    bool operator==(const GraphKey& o) const {
      return P == o.P && G == o.G && B == o.B && S == o.S && L == o.L && thresh == o.thresh && seed == o.seed && split == o.split;
    }
  };
  // One captured graph PER SHAPE.  The grouped weight-gradient launch and the norm column sums read device-resident tables
  // (operand pointers, reduction lengths, tile counts) that depend on (B, S, L); a graph bakes the table's ADDRESS in, so every
  // cached shape owns a table slot of its own and a pass with another shape can never rewrite the table a kept graph replays
  // (training batches are padded to their longest label sequence, ref: music2midi/tokenizer.py:86-96, so L changes from batch
  // to batch and comes back).  The first call with a key runs directly and uploads the slot's tables, the second captures, later
  // ones replay; the least recently used slot is recycled (its graphs destroyed first).  Slot N_SLOTS serves the passes issued on
  // the caller's stream (no graph).
  static constexpr int N_SLOTS = 8;
  struct GraphSlot {
    GraphKey key;
    bool valid = false;
    int calls = 0;
    uint64_t last_use = 0;
    hipGraphExec_t gexec = nullptr, gexec2 = nullptr;      // gexec2: second half of a split pass
    int nodes = 0;                                         // nodes of the captured graph(s): launches per step
    std::vector<unsigned char> dw_probs_host[2];           // host images of the slot's device tables (re-uploaded only when they change);
    std::vector<int64_t> norm_offs_host[2];                // [1]: the second half of a split pass
  };
  GraphSlot slots[N_SLOTS + 1];
  int cur_slot = N_SLOTS;                // the table slot the pass being issued uses
  uint64_t tick = 0;
  // optimizer
  AfPlan af;
  unsigned char* af_mem = nullptr;
  int step = 0;
};

namespace {

int64_t add_tensor(m2m_trainer* t, const std::string& name, int rows, int cols, int64_t& off) {
  const int64_t o = off;
  t->tensors.push_back({name, o, rows, cols});
  off = align_up(off + (int64_t)rows * (cols ? cols : 1), 64);
  return o;
}

void build_layout(m2m_trainer* t) {
  const m2m_t5_geometry& g = t->g;
  const int d = g.d_model, dff = g.d_ff, inner = t->inner, V = g.vocab_size, H = g.num_heads;
  int64_t off = 0;
  const std::string T5 = "transformer.";
  t->o_shared = add_tensor(t, T5 + "shared.weight", V, d, off);
  t->o_lm = add_tensor(t, T5 + "lm_head.weight", V, d, off);
  t->o_erb = add_tensor(t, T5 + "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", g.num_buckets, H, off);
  t->o_drb = add_tensor(t, T5 + "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", g.num_buckets, H, off);
  t->o_eln = add_tensor(t, T5 + "encoder.final_layer_norm.weight", d, 0, off);
  t->o_dln = add_tensor(t, T5 + "decoder.final_layer_norm.weight", d, 0, off);
  auto attn3 = [&](const std::string& p, int64_t& first) {   // q, k, v back to back WITHOUT padding: one fused [3*inner, d] matrix
    first = off;
    for (const char* n : {"q", "k", "v"}) {
      t->tensors.push_back({p + "." + n + ".weight", off, inner, d});
      off += (int64_t)inner * d;
    }
    off = align_up(off, 64);
  };
  t->enc.resize(g.num_layers);
  for (int l = 0; l < g.num_layers; ++l) {
    const std::string p = T5 + "encoder.block." + std::to_string(l) + ".layer.";
    EncOff& e = t->enc[l];
    e.ln0 = add_tensor(t, p + "0.layer_norm.weight", d, 0, off);
    attn3(p + "0.SelfAttention", e.qkv);
    e.o = add_tensor(t, p + "0.SelfAttention.o.weight", d, inner, off);
    e.ln1 = add_tensor(t, p + "1.layer_norm.weight", d, 0, off);
    e.wi = off;
    t->tensors.push_back({p + "1.DenseReluDense.wi_0.weight", off, dff, d}); off += (int64_t)dff * d;
    t->tensors.push_back({p + "1.DenseReluDense.wi_1.weight", off, dff, d}); off = align_up(off + (int64_t)dff * d, 64);
    e.wo = add_tensor(t, p + "1.DenseReluDense.wo.weight", d, dff, off);
  }
  t->dec.resize(g.num_decoder_layers);
  t->dec_begin = off;
  for (int l = 0; l < g.num_decoder_layers; ++l) {
    const std::string p = T5 + "decoder.block." + std::to_string(l) + ".layer.";
    DecOff& e = t->dec[l];
    e.ln0 = add_tensor(t, p + "0.layer_norm.weight", d, 0, off);
    attn3(p + "0.SelfAttention", e.qkv);
    e.o = add_tensor(t, p + "0.SelfAttention.o.weight", d, inner, off);
    e.ln1 = add_tensor(t, p + "1.layer_norm.weight", d, 0, off);
    e.cq = add_tensor(t, p + "1.EncDecAttention.q.weight", inner, d, off);
    e.ckv = off;
    t->tensors.push_back({p + "1.EncDecAttention.k.weight", off, inner, d}); off += (int64_t)inner * d;
    t->tensors.push_back({p + "1.EncDecAttention.v.weight", off, inner, d}); off = align_up(off + (int64_t)inner * d, 64);
    e.co = add_tensor(t, p + "1.EncDecAttention.o.weight", d, inner, off);
    e.ln2 = add_tensor(t, p + "2.layer_norm.weight", d, 0, off);
    e.wi = off;
    t->tensors.push_back({p + "2.DenseReluDense.wi_0.weight", off, dff, d}); off += (int64_t)dff * d;
    t->tensors.push_back({p + "2.DenseReluDense.wi_1.weight", off, dff, d}); off = align_up(off + (int64_t)dff * d, 64);
    e.wo = add_tensor(t, p + "2.DenseReluDense.wo.weight", d, dff, off);
  }
  t->dec_end = off;
  t->o_cond.resize(t->n_cond);
  for (int i = 0; i < t->n_cond; ++i)
    t->o_cond[i] = add_tensor(t, "conditioning.embeds." + std::to_string(i) + ".weight", t->cond_rows[i], d, off);
  t->n_floats = off;
}

// -------------------------------------------------------------- arena
struct Carver {
  int64_t off = 0;
  int64_t take(int64_t bytes) { const int64_t o = off; off = align_up(off + bytes, 256); return o; }
};

int build_optimizer(m2m_trainer* t) {
  std::vector<AfTensor> at;
  std::vector<AfBlock> ab;
  int row_off = 0, cfac_off = 0;
  int64_t col_off = 0, state_off = 0;
  for (const TensorDesc& td : t->tensors) {
    AfTensor a{};
    a.offset = td.off;
    a.rows = td.cols ? td.rows : 1;
    a.cols = td.cols ? td.cols : td.rows;
    a.row_off = row_off; a.cfac_off = cfac_off; a.col_off = col_off; a.state_off = state_off;
    a.block0 = (int)ab.size();
    a.nblocks = ceil_div(a.rows, AF_ROWS);
    for (int b = 0; b < a.nblocks; ++b) ab.push_back({(int)at.size(), b * AF_ROWS, col_off + (int64_t)b * a.cols});
    row_off += a.rows; cfac_off += a.cols; col_off += (int64_t)a.nblocks * a.cols;
    state_off += (a.rows > 1 ? a.rows : 0) + a.cols;
    at.push_back(a);
  }
  AfPlan& p = t->af;
  p.n_tensors = (int)at.size(); p.n_blocks = (int)ab.size(); p.state_floats = state_off;
  Carver c;
  const int64_t o_t = c.take((int64_t)at.size() * sizeof(AfTensor)), o_b = c.take((int64_t)ab.size() * sizeof(AfBlock));
  const int64_t o_rs = c.take((int64_t)row_off * 4), o_cp = c.take(col_off * 4), o_ba = c.take((int64_t)ab.size() * 4),
                o_bb = c.take((int64_t)ab.size() * 4), o_st = c.take(state_off * 4), o_rf = c.take((int64_t)row_off * 4),
                o_cf = c.take((int64_t)cfac_off * 4), o_ts = c.take((int64_t)at.size() * 2 * 4);
  M2M_CHECK_HIP(hipMalloc((void**)&t->af_mem, (size_t)c.off));
  M2M_CHECK_HIP(hipMemset(t->af_mem, 0, (size_t)c.off));
  unsigned char* b = t->af_mem;
  p.tensors = (AfTensor*)(b + o_t); p.blocks = (AfBlock*)(b + o_b); p.rowsum = (float*)(b + o_rs); p.colpart = (float*)(b + o_cp);
  p.blk_a = (float*)(b + o_ba); p.blk_b = (float*)(b + o_bb); p.state = (float*)(b + o_st); p.rfac = (float*)(b + o_rf);
  p.cfac = (float*)(b + o_cf); p.tstat = (float*)(b + o_ts);
  M2M_CHECK_HIP(hipMemcpy(p.tensors, at.data(), at.size() * sizeof(AfTensor), hipMemcpyHostToDevice));
  M2M_CHECK_HIP(hipMemcpy(p.blocks, ab.data(), ab.size() * sizeof(AfBlock), hipMemcpyHostToDevice));
  return M2M_OK;
}

int build_arena(m2m_trainer* t) {
  const m2m_t5_geometry& g = t->g;
  const int64_t es = (int64_t)t->es, d = g.d_model, dff = g.d_ff, inner = t->inner, V = g.vocab_size, H = g.num_heads;
  const int64_t B = t->max_batch, S = t->max_enc, L = t->max_dec;
  const int64_t Me = B * S, Md = B * L, Mx = Me > Md ? Me : Md;
  const int64_t lps = align_up(S, 8), lpl = align_up(L, 8), lpm = lps > lpl ? lps : lpl, Sm = S > L ? S : L;
  const int Le = g.num_layers, Ld = g.num_decoder_layers;
  Carver c;
  std::vector<int64_t> o;
  auto T = [&](int64_t elems) { return c.take(elems * es); };
  auto F = [&](int64_t elems) { return c.take(elems * 4); };
  // a layer's probability buffer [B*H][Sq][ldp]; the whole-head attention path keeps only its dropout keep words there ([B*H][ceil(Sk/32)][round_up_32(Sq)] x 4 bytes)
  auto PB = [&](int64_t Sq, int64_t Sk, int64_t ldp) { return std::max<int64_t>(B * H * Sq * ldp, B * H * ((Sk + 31) / 32) * align_up(Sq, 32) * 2); };
  // order of `o` must match the assignment below
  for (int i = 0; i < 2 * Le + 1; ++i) o.push_back(F(Me * d));
  for (int i = 0; i < 3 * Ld + 1; ++i) o.push_back(F(Md * d));
  for (int l = 0; l < Le; ++l) { o.push_back(T(Me * d)); o.push_back(T(Me * d)); o.push_back(T(Me * 3 * inner)); o.push_back(T(PB(S, S, lps)));
                                  o.push_back(T(Me * inner)); o.push_back(T(Me * 2 * dff)); o.push_back(T(Me * dff)); }
  for (int l = 0; l < Ld; ++l) { o.push_back(T(Md * d)); o.push_back(T(Md * d)); o.push_back(T(Md * d)); o.push_back(T(Md * 3 * inner));
                                  o.push_back(T(PB(L, L, lpl))); o.push_back(T(Md * inner)); o.push_back(T(Md * inner)); o.push_back(T(Me * 2 * inner));
                                  o.push_back(T(PB(L, S, lps))); o.push_back(T(Md * inner)); o.push_back(T(Md * 2 * dff)); o.push_back(T(Md * dff)); }
  const int64_t Sp32 = align_up(S, 32), Lp32 = align_up(L, 32);
  std::vector<int64_t> o_kte, o_ktd, o_ktc;
  std::vector<int64_t> o_lse;
  for (int l = 0; l < Le + 2 * Ld; ++l) o_lse.push_back(F(B * H * Sm));
  for (int l = 0; l < Le; ++l) o_kte.push_back(T(2 * B * H * 64 * Sp32));
  for (int l = 0; l < Ld; ++l) { o_ktd.push_back(T(2 * B * H * 64 * Lp32)); o_ktc.push_back(T(2 * B * H * 64 * Sp32)); }
  const int64_t o_hE = T(Me * d), o_hD = T(Md * d), o_logits = F(Md * V), o_sc = F(B * H * Sm * lpm), o_dxa = F(Mx * d), o_dxb = F(Mx * d),
                o_dh = F(Mx * d), o_dhE = F(Me * d), o_dwp = F((int64_t)RN_BLOCKS * d * (2 * Le + 3 * Ld + 2)), o_nofs = c.take((int64_t)(m2m_trainer::N_SLOTS + 1) * 64 * 8), o_rl = F(Md), o_inv = F(64), o_drel = F((int64_t)(Le + Ld) * B * H * std::max<int64_t>(2 * Sm, ((Sm + 31) / 32) * (Sm + 32)) + (int64_t)std::max(Le, Ld) * H * 2 * Sm + B * H * 2 * Sm),
                o_etab = F(H * (2 * S)), o_dtab = F(H * (2 * L)), o_dlog = T(Md * align_up(V, 8)), o_dxT = T(Mx * d), o_dmid = T(Mx * dff),
                o_dab = T(Mx * 2 * dff), o_dO = T(Mx * inner), o_dqkv = T(Mx * 3 * inner), o_dS = T(B * H * Sm * lpm), o_dcq = T(Md * inner),
                o_dckv = T(Me * 2 * inner), o_lab = c.take(Md * 8), o_cnd = c.take(B * 8 * 8), o_skey = c.take(256), o_decin = c.take(Md * 8), o_eb = c.take(2 * S * 4), o_db = c.take(2 * L * 4),
                o_co = c.take(64 * 8), o_cr = c.take(64 * 4), o_cnt = c.take(256), o_wc = (t->precision == M2M_PREC_BF16) ? T(t->n_floats) : 0, o_wil = (t->precision == M2M_PREC_BF16) ? T(t->n_floats) : 0,
                o_wt = T(t->n_floats);
  std::vector<int64_t> o_ring[m2m_trainer::K_KINDS];
  for (int i = 1; i < 2 * Le + 3 * Ld; ++i) o_ring[m2m_trainer::K_DXT].push_back(T(Mx * d));
  for (int i = 1; i < Le + Ld; ++i) { o_ring[m2m_trainer::K_DAB].push_back(T(Mx * 2 * dff)); o_ring[m2m_trainer::K_DQKV].push_back(T(Mx * 3 * inner)); }
  for (int i = 1; i < Ld; ++i) { o_ring[m2m_trainer::K_DCQ].push_back(T(Md * inner)); o_ring[m2m_trainer::K_DCKV].push_back(T(Me * 2 * inner)); }
  const int64_t o_dwp_tab = c.take((int64_t)(m2m_trainer::N_SLOTS + 1) * 256 * (int64_t)sizeof(DwProb));
  const int64_t Mxp = align_up(Mx, 8), fmax = std::max<int64_t>(std::max<int64_t>(3 * inner, 2 * dff), align_up(V, 8));
  const int64_t o_tA = T(fmax * Mxp), o_tB = T(std::max<int64_t>(std::max<int64_t>(dff, inner), d) * Mxp);
  t->kpart_floats = std::max<int64_t>((int64_t)8 << 20, fmax * std::max<int64_t>(dff, d) + 64);
  const int64_t o_kp = F(t->kpart_floats);
  // weight-transpose table: every 2-D weight group as one matrix [N][K] (q|k|v, wi_0|wi_1, cross k|v are fused groups)
  std::vector<WtBlock> wtb;
  {
    auto add = [&](int64_t off, int N, int K) {
      const int il = (N == 2 * (int)dff && K == (int)d && dff % 32 == 0) ? (int)dff : 0;      // the gate pairs
      for (int tn = 0; tn < ceil_div(N, 64); ++tn)
        for (int tk = 0; tk < ceil_div(K, 64); ++tk) wtb.push_back({off, N, K, tn, tk, il});
    };
    add(t->o_lm, (int)V, (int)d);
    for (const EncOff& e : t->enc) { add(e.qkv, 3 * (int)inner, (int)d); add(e.o, (int)d, (int)inner); add(e.wi, 2 * (int)dff, (int)d); add(e.wo, (int)d, (int)dff); }
    for (const DecOff& e : t->dec) { add(e.qkv, 3 * (int)inner, (int)d); add(e.o, (int)d, (int)inner); add(e.cq, (int)inner, (int)d);
                                     add(e.ckv, 2 * (int)inner, (int)d); add(e.co, (int)d, (int)inner); add(e.wi, 2 * (int)dff, (int)d); add(e.wo, (int)d, (int)dff); }
  }
  const int64_t o_wtb = c.take((int64_t)wtb.size() * sizeof(WtBlock));
  std::vector<int64_t> pads;             // gaps between consecutive tensors of the flat layout (and behind the last one)
  {
    std::vector<std::pair<int64_t, int64_t>> spans;
    for (const TensorDesc& td : t->tensors) spans.push_back({td.off, (int64_t)td.rows * (td.cols ? td.cols : 1)});
    std::sort(spans.begin(), spans.end());
    int64_t pos = 0;
    for (const auto& sp : spans) {
      if (sp.first > pos) { pads.push_back(pos); pads.push_back(sp.first - pos); }
      pos = std::max(pos, sp.first + sp.second);
    }
    if (t->n_floats > pos) { pads.push_back(pos); pads.push_back(t->n_floats - pos); }
  }
  const int64_t o_pads = c.take((int64_t)std::max<size_t>(pads.size(), 2) * 8);
  // fp8 mode: MXFP8 copies of every projection matrix (lm_head stays bf16) + quantised-activation scratch
  std::vector<W8Tile> w8t;
  int64_t w8_bytes = 0, o_w8 = 0, o_w8t = 0, o_q8a = 0, o_s8a = 0, o_q8ta = 0, o_s8ta = 0, o_q8tb = 0, o_s8tb = 0;
  if (t->fp8) {
    auto addl = [&](int64_t off, int N, int K) {
      m2m_trainer::LinW w{off, N, K, (int)align_up(N, 128), 0, 0, 0, 0};
      w.q = w8_bytes; w8_bytes = align_up(w8_bytes + (int64_t)N * K, 256);
      w.qs = w8_bytes; w8_bytes = align_up(w8_bytes + (int64_t)N * (K / 32), 256);
      w.qt = w8_bytes; w8_bytes = align_up(w8_bytes + (int64_t)K * w.Np, 256);
      w.qts = w8_bytes; w8_bytes = align_up(w8_bytes + (int64_t)K * (w.Np / 32), 256);
      t->lin.push_back(w);
      for (int tn = 0; tn < ceil_div(N, 32); ++tn)
        for (int tk = 0; tk < ceil_div(K, 64); ++tk) w8t.push_back({off, N, K, w.Np, w.q, w.qs, w.qt, w.qts, tn, tk});
    };
    for (const EncOff& e : t->enc) { addl(e.qkv, 3 * (int)inner, (int)d); addl(e.o, (int)d, (int)inner); addl(e.wi, 2 * (int)dff, (int)d); addl(e.wo, (int)d, (int)dff); }
    for (const DecOff& e : t->dec) { addl(e.qkv, 3 * (int)inner, (int)d); addl(e.o, (int)d, (int)inner); addl(e.cq, (int)inner, (int)d);
                                     addl(e.ckv, 2 * (int)inner, (int)d); addl(e.co, (int)d, (int)inner); addl(e.wi, 2 * (int)dff, (int)d); addl(e.wo, (int)d, (int)dff); }
    const int64_t Mp128 = align_up(Mx, 128), f8 = align_up(fmax, 128);
    o_w8 = c.take(w8_bytes); o_w8t = c.take((int64_t)w8t.size() * sizeof(W8Tile));
    o_q8a = c.take(Mx * f8); o_s8a = c.take(Mx * (f8 / 32));
    o_q8ta = c.take(f8 * Mp128); o_s8ta = c.take(f8 * (Mp128 / 32));
    o_q8tb = c.take(align_up(std::max<int64_t>(std::max<int64_t>(dff, inner), d), 128) * Mp128);
    o_s8tb = c.take(align_up(std::max<int64_t>(std::max<int64_t>(dff, inner), d), 128) * (Mp128 / 32));
  }
  t->arena_bytes = c.off;
  hipError_t e = hipMalloc((void**)&t->arena, (size_t)c.off);
  if (e != hipSuccess) { set_error("m2m_trainer_create: hipMalloc(%lld) failed: %s", (long long)c.off, hipGetErrorString(e)); return M2M_ERR_NOMEM; }
  M2M_CHECK_HIP(hipMemset(t->arena, 0, (size_t)c.off));
  unsigned char* b = t->arena;
  size_t k = 0;
  auto nextF = [&]() { return (float*)(b + o[k++]); };
  auto nextT = [&]() { return (void*)(b + o[k++]); };
  for (int i = 0; i < 2 * Le + 1; ++i) t->xe.push_back(nextF());
  for (int i = 0; i < 3 * Ld + 1; ++i) t->xd.push_back(nextF());
  for (int l = 0; l < Le; ++l) { t->h0e.push_back(nextT()); t->h1e.push_back(nextT()); t->qkve.push_back(nextT()); t->Pe.push_back(nextT());
                                  t->aoe.push_back(nextT()); t->abe.push_back(nextT()); t->mide.push_back(nextT()); }
  for (int l = 0; l < Ld; ++l) { t->h0d.push_back(nextT()); t->h1d.push_back(nextT()); t->h2d.push_back(nextT()); t->qkvd.push_back(nextT());
                                  t->Pd.push_back(nextT()); t->aod.push_back(nextT()); t->cqd.push_back(nextT()); t->ckvd.push_back(nextT());
                                  t->Pcd.push_back(nextT()); t->aocd.push_back(nextT()); t->abd.push_back(nextT()); t->midd.push_back(nextT()); }
  for (int l = 0; l < Le; ++l) t->lse_e.push_back((float*)(b + o_lse[l]));
  for (int l = 0; l < Ld; ++l) { t->lse_d.push_back((float*)(b + o_lse[Le + 2 * l])); t->lse_c.push_back((float*)(b + o_lse[Le + 2 * l + 1])); }
  for (int64_t off : o_kte) t->kte.push_back(b + off);
  for (int64_t off : o_ktd) t->ktd.push_back(b + off);
  for (int64_t off : o_ktc) t->ktc.push_back(b + off);
  t->hE = b + o_hE; t->hD = b + o_hD; t->logits = (float*)(b + o_logits); t->sc = (float*)(b + o_sc); t->dxa = (float*)(b + o_dxa);
  t->dxb = (float*)(b + o_dxb); t->dh = (float*)(b + o_dh); t->dhE = (float*)(b + o_dhE); t->dw_part = (float*)(b + o_dwp); t->norm_offs_dev = (int64_t*)(b + o_nofs);
  t->row_loss = (float*)(b + o_rl); t->inv_n = (float*)(b + o_inv); t->drel = (float*)(b + o_drel); t->drel_slot_floats = B * H * std::max<int64_t>(2 * Sm, ((Sm + 31) / 32) * (Sm + 32)); t->etab = (float*)(b + o_etab);
  t->dtab = (float*)(b + o_dtab); t->dlog = b + o_dlog; t->dxT = b + o_dxT; t->dmid = b + o_dmid; t->dab = b + o_dab; t->dO = b + o_dO;
  t->dqkv = b + o_dqkv; t->dS = b + o_dS; t->dcq = b + o_dcq; t->dckv = b + o_dckv; t->dec_in = (int64_t*)(b + o_decin);
  {
    void* first[m2m_trainer::K_KINDS] = {t->dxT, t->dab, t->dqkv, t->dcq, t->dckv};
    for (int k = 0; k < m2m_trainer::K_KINDS; ++k) {
      t->ring[k].push_back(first[k]);
      for (int64_t off : o_ring[k]) t->ring[k].push_back(b + off);
    }
    t->dw_probs_dev = b + o_dwp_tab;
  }
  t->labels_buf = (int64_t*)(b + o_lab); t->cond_buf = (int64_t*)(b + o_cnd);
  t->step_key_dev = (uint64_t*)(b + o_skey); t->step_ctr_dev = t->step_key_dev + 1; t->loss_dev = (float*)(t->step_key_dev + 4);
  t->ebucket = (int*)(b + o_eb); t->dbucket = (int*)(b + o_db); t->counter = (int*)(b + o_cnt); t->cond_off_dev = (int64_t*)(b + o_co); t->cond_rows_dev = (int*)(b + o_cr);
  t->Wc = (t->precision == M2M_PREC_BF16) ? (void*)(b + o_wc) : nullptr;
  t->Wil = (t->precision == M2M_PREC_BF16) ? (void*)(b + o_wil) : nullptr;
  t->WT = b + o_wt; t->tA = b + o_tA; t->tB = b + o_tB; t->kpart = (float*)(b + o_kp); t->wt_blocks = b + o_wtb; t->n_wt_blocks = (int)wtb.size();
  M2M_CHECK_HIP(hipMemcpy(t->wt_blocks, wtb.data(), wtb.size() * sizeof(WtBlock), hipMemcpyHostToDevice));
  t->pads_dev = (int64_t*)(b + o_pads); t->n_pads = (int)(pads.size() / 2);
  if (!pads.empty()) M2M_CHECK_HIP(hipMemcpy(t->pads_dev, pads.data(), pads.size() * 8, hipMemcpyHostToDevice));
  if (t->fp8) {
    t->w8 = b + o_w8; t->w8_tiles = b + o_w8t; t->n_w8_tiles = (int)w8t.size();
    t->q8a = b + o_q8a; t->s8a = b + o_s8a; t->q8ta = b + o_q8ta; t->s8ta = b + o_s8ta; t->q8tb = b + o_q8tb; t->s8tb = b + o_s8tb;
    M2M_CHECK_HIP(hipMemcpy(t->w8_tiles, w8t.data(), w8t.size() * sizeof(W8Tile), hipMemcpyHostToDevice));
  }
  M2M_CHECK_HIP(hipMemcpy(t->cond_off_dev, t->o_cond.data(), t->o_cond.size() * 8, hipMemcpyHostToDevice));
  M2M_CHECK_HIP(hipMemcpy(t->cond_rows_dev, t->cond_rows.data(), t->cond_rows.size() * 4, hipMemcpyHostToDevice));
  return M2M_OK;
}

// ------------------------------------------------------------ typed helpers
template <typename T>
struct Ops {
  m2m_trainer* t;
  hipStream_t st;
  const float* P;      // master parameters (fp32)
  bool use_tuned = getenv("M2M_TRAIN_PLAIN_GEMM") == nullptr;   // diagnostic switch: everything through bgemm
  int dw_kmajor = getenv("M2M_TRAIN_DW_KMAJOR") ? atoi(getenv("M2M_TRAIN_DW_KMAJOR")) : -1;   // -1 = by size, 0 / 1 = forced
  const T* W(int64_t off) const { return (t->precision == M2M_PREC_BF16 ? reinterpret_cast<const T*>(t->Wc) : reinterpret_cast<const T*>(P)) + off; }

  static bool mxq_fused() { static const bool on = [] { const char* v = getenv("M2M_FP8_FUSED_Q"); return !(v && v[0] == '0'); }(); return on; }
  // fp8 mode: the projection matrix that starts at parameter offset `off`, or null (lm_head, non-projection operands)
  const m2m_trainer::LinW* lin8(int64_t off) const {
    if (!t->fp8) return nullptr;
    for (const auto& w : t->lin) if (w.off == off) return &w;
    return nullptr;
  }
  float* Gbase = nullptr;     // flat gradient buffer of this call (dW recognises its weight by the output offset)

  // dropout sites: one key per (layer, place); site < 0 or p == 0: no dropout
  bool dropping(int site) const { return site >= 0 && t->drop_thresh != 0; }
  DropKey key(int site) const { return DropKey{t->step_key_dev, (uint64_t)site * 0x9E3779B97F4A7C15ull}; }

  // ---- weight-gradient products.  A backward sub-layer brackets itself with begin_sub(kinds) / end_sub(): begin_sub
  // points t->dxT ... at fresh ring entries, dW() calls in between are queued.  Grouped mode: the queue becomes ONE launch
  // after the backward pass (flush_group).  Side-stream mode (fp8): end_sub() hands the queue to the side stream behind
  // everything the main stream has issued so far.
  hipStream_t st2 = nullptr;
  bool group = false;
  mutable std::vector<std::function<int(hipStream_t)>> pending;
  mutable std::vector<DwProb> probs;
  mutable std::vector<int64_t> norm_offs;
  struct BiasJob { const int* buckets; float* Gtab; int nB, Sq, Sk, first_slot, layers; };
  mutable std::vector<BiasJob> bias_jobs;          // grouped mode: per stack, the self-attention layers whose diagonal sums wait in t->drel slots
  mutable int drel_slots = 0;
  // grouped mode: the RMSNorm backward of a sub-layer also emits its dx in the GEMM-input type for the sub-layer that runs next
  // (site = that sub-layer's branch-output dropout site; -2: nobody), straight into that sub-layer's dxT ring entry
  mutable int after_site = -2;
  struct PreCvt { const float* src = nullptr; const void* dst = nullptr; int site = -2; };
  mutable PreCvt pre;
  mutable int pos[m2m_trainer::K_KINDS] = {0, 0, 0, 0, 0};
  mutable int sub = 0;
  mutable bool used[2] = {false, false};
  int begin_sub(unsigned kinds) const {
    void** dst[m2m_trainer::K_KINDS] = {&t->dxT, &t->dab, &t->dqkv, &t->dcq, &t->dckv};
    if (group) {                           // a ring entry per use: nothing is overwritten before the grouped launch
      for (int k = 0; k < m2m_trainer::K_KINDS; ++k)
        if (kinds & (1u << k)) { *dst[k] = t->ring[k][pos[k] % t->ring[k].size()]; pos[k] += 1; }
      return M2M_OK;
    }
    // side-stream / single-stream mode: two alternating sets (they stay in the Infinity Cache; fresh buffers per
    // sub-layer measured 10.9 vs 10.0 ms per fp8 step); a set is rewritten only after the side stream has read it
    const int slot = sub & 1;
    if (st2 && used[slot]) M2M_CHECK_HIP(hipStreamWaitEvent(st, t->ev_free[slot], 0));
    for (int k = 0; k < m2m_trainer::K_KINDS; ++k)
      if (kinds & (1u << k)) *dst[k] = t->ring[k][slot];
    return M2M_OK;
  }
  int end_sub() const {
    if (group) return M2M_OK;
    const int slot = sub & 1;
    sub += 1;
    if (!st2 || pending.empty()) return M2M_OK;
    M2M_CHECK_HIP(hipEventRecord(t->ev_ready, st));
    M2M_CHECK_HIP(hipStreamWaitEvent(st2, t->ev_ready, 0));
    for (auto& f : pending) { const int rc = f(st2); if (rc != M2M_OK) { pending.clear(); return rc; } }
    pending.clear();
    M2M_CHECK_HIP(hipEventRecord(t->ev_free[slot], st2));
    used[slot] = true;
    return M2M_OK;
  }
  int join_side() const {                  // the main stream continues only after every queued product has finished
    if (!st2) return M2M_OK;
    for (int p = 0; p < 2; ++p)
      if (used[p]) M2M_CHECK_HIP(hipStreamWaitEvent(st, t->ev_free[p], 0));
    return M2M_OK;
  }
  // A split pass (t->sync_stream) flushes twice: `phase` picks the half of the device tables (and the host image) a flush uses,
  // so both halves stay constant from step to step and a captured graph never sees a table change.
  mutable int phase = 0;
  mutable int norm_slot_base = 0;          // partial images already handed to an earlier flush of this pass keep their slices
  int flush_norms() const {
    if (!group || norm_offs.empty()) return M2M_OK;
    M2M_REQUIRE(norm_offs.size() <= 32 && (int)norm_offs.size() <= 2 * t->g.num_layers + 3 * t->g.num_decoder_layers + 2, "training: too many norms");
    int64_t* const offs_dev = t->norm_offs_dev + 64 * t->cur_slot + 32 * phase;
    std::vector<int64_t>& offs_host = t->slots[t->cur_slot].norm_offs_host[phase];
    if (offs_host != norm_offs) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cs);
      M2M_REQUIRE(cs == hipStreamCaptureStatusNone, "training: the norm table changed inside a graph capture");
      M2M_CHECK_HIP(hipStreamSynchronize(st));
      M2M_CHECK_HIP(hipMemcpy(offs_dev, norm_offs.data(), norm_offs.size() * 8, hipMemcpyHostToDevice));
      offs_host = norm_offs;
    }
    const int d = t->g.d_model;
    hipLaunchKernelGGL(colsum_group_kernel, dim3(ceil_div(d, 64), (unsigned)norm_offs.size()), dim3(256), 0, st,
                       t->dw_part + (int64_t)norm_slot_base * RN_BLOCKS * d, offs_dev, Gbase, RN_BLOCKS, d);
    M2M_CHECK_HIP(hipGetLastError());
    norm_slot_base += (int)norm_offs.size();
    norm_offs.clear();
    return M2M_OK;
  }
  int flush_group() const {
    if (!group || probs.empty()) return M2M_OK;
    M2M_REQUIRE(probs.size() <= 128, "training: %zu weight-gradient products exceed the table", probs.size());
    unsigned char* const tab_dev = reinterpret_cast<unsigned char*>(t->dw_probs_dev) + (size_t)(2 * t->cur_slot + phase) * 128 * sizeof(DwProb);
    std::vector<unsigned char>& tab_host = t->slots[t->cur_slot].dw_probs_host[phase];
    int tiles = 0;
    for (DwProb& p : probs) { p.tn2 = ceil_div(p.g.N2, 128); p.tile0 = tiles; tiles += ceil_div(p.g.N1, 128) * p.tn2; }
    const size_t bytes = probs.size() * sizeof(DwProb);
    if (tab_host.size() != bytes || memcmp(tab_host.data(), probs.data(), bytes) != 0) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cs);
      M2M_REQUIRE(cs == hipStreamCaptureStatusNone, "training: the weight-gradient table changed inside a graph capture");
      M2M_CHECK_HIP(hipStreamSynchronize(st));                 // nothing in flight still reads the old table
      M2M_CHECK_HIP(hipMemcpy(tab_dev, probs.data(), bytes, hipMemcpyHostToDevice));
      tab_host.assign(reinterpret_cast<const unsigned char*>(probs.data()), reinterpret_cast<const unsigned char*>(probs.data()) + bytes);
    }
    static const bool tr_reads = [] { const char* v = getenv("M2M_DW_TR"); return !(v && v[0] == '0'); }();      // 0: the register-transposing tile
    static const int xcd_order = [] { const char* v = getenv("M2M_DW_XCD"); return v ? atoi(v) : 1; }();
    const int grid = xcd_order ? 8 * ceil_div(tiles, 8) : tiles;
    if (t->precision == M2M_PREC_BF16 && tr_reads)
      hipLaunchKernelGGL((dw_group_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const DwProb*)tab_dev, (int)probs.size(), tiles, xcd_order);
    else if (t->precision == M2M_PREC_BF16)
      hipLaunchKernelGGL((dw_group_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const DwProb*)tab_dev, (int)probs.size(), tiles, xcd_order);
    else
      hipLaunchKernelGGL((dw_group_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const DwProb*)tab_dev, (int)probs.size(), tiles, xcd_order);
    M2M_CHECK_HIP(hipGetLastError());
    probs.clear();
    return M2M_OK;
  }
  int mm(int epi, const void* A, int64_t lda, int akm, const void* B, int64_t ldb, int bkm, void* C, int64_t ldc, int M, int N, int K,
         const float* R = nullptr, int drop_site = -1) const {
    BGemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.R = R; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_kmajor = akm; g.b_kmajor = bkm;
    g.nb1 = 1; g.nb2 = 1; g.alpha = 1.0f;
    if (dropping(drop_site)) { g.drop_thresh = t->drop_thresh; g.drop_scale = t->drop_scale; g.drop_key = key(drop_site).salt; g.drop_step = t->step_key_dev; }
    if (!akm && !bkm && t->fp8 && t->fp8_fwd) {          // a forward projection Y = X . W^T on MXFP8 operands
      const int64_t off = reinterpret_cast<const T*>(B) - W(0);
      if (const m2m_trainer::LinW* w = lin8(off)) {
        MxGemmArgs m{};
        m.A = t->q8a; m.sA = t->s8a; m.B = t->w8 + w->q; m.sB = t->w8 + w->qs; m.C = C; m.R = R; m.M = M; m.N = N; m.K = K;
        m.lda = K; m.ldb = K; m.ldc = ldc; m.drop_thresh = g.drop_thresh; m.drop_scale = g.drop_scale; m.drop_key = g.drop_key; m.drop_step = g.drop_step;
        if (mxq_fused() && K % 128 == 0 && lda % 8 == 0) {      // activations quantised inside the product's staging: one launch
          m.Asrc = A; m.ld_src = lda; m.Kvalid = K;
          return launch_mxgemm_q(0, epi, m, st);
        }
        int rc = launch_mxq_rows(1, A, lda, t->q8a, t->s8a, M, K, K, 0, st);
        if (rc != M2M_OK) return rc;
        return launch_mxgemm(0, 0, epi, m, st);
      }
    }
    // dense NT products with K % 64 == 0 go through the inference path's tuned kernel (128x128 tiles, register-prefetched
    // staging, XCD-aware tile order): every forward projection and every dX product qualifies
    if (!akm && !bkm && K % 64 == 0 && lda == K && ldb == K && use_tuned) {
      GemmArgs a{};
      a.A = A; a.W = B; a.M = M; a.N = N; a.K = K; a.out = C; a.ldo = (int)ldc; a.vt_which = -1;
      if (epi == TG_STORE_T) return launch_gemm(t->precision, EPI_STORE, a, st);
      if (epi == TG_STORE_F32) return launch_gemm(t->precision, EPI_STORE_F32, a, st);
      if (epi == TG_ACC_F32) return launch_gemm(t->precision, EPI_RESID, a, st);
      a.resid = R; a.drop_thresh = g.drop_thresh; a.drop_scale = g.drop_scale; a.drop_key = g.drop_key; a.drop_step = g.drop_step;
      return launch_gemm(t->precision, EPI_RESID, a, st);
    }
    return launch_bgemm(t->precision, epi, g, st);
  }
  // gradient entering a (possibly dropped) branch, in the GEMM-input type
  int cvt_branch(const float* src, void* dst, int64_t n, int site) const {
    if (pre.src == src && pre.dst == dst && pre.site == site) { pre = PreCvt{}; return M2M_OK; }      // the producing norm already wrote it
    pre = PreCvt{};
    if (!dropping(site)) return launch_cvt(t->precision, src, dst, n, st);
    hipLaunchKernelGGL(cvt_drop_kernel<T>, dim3(grid_1d(n)), dim3(256), 0, st, src, (T*)dst, n, key(site), t->drop_thresh, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  // batched over (clip b, head h): operand X of clip b / head h starts at X + b*s1 + h*s2 (elements of its own type)
  int mmbh(int epi, const T* A, int64_t lda, int akm, int64_t sA1, int64_t sA2, const T* B, int64_t ldb, int bkm, int64_t sB1, int64_t sB2,
           void* C, int64_t ldc, int64_t sC1, int64_t sC2, int nB, int M, int N, int K) const {
    BGemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.R = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_kmajor = akm; g.b_kmajor = bkm;
    g.nb1 = nB; g.nb2 = t->g.num_heads; g.sA1 = sA1; g.sA2 = sA2; g.sB1 = sB1; g.sB2 = sB2; g.sC1 = sC1; g.sC2 = sC2; g.alpha = 1.0f;
    return launch_bgemm(t->precision, epi, g, st);
  }
  // two products of one shape in one launch (BGemmArgs pair mode): (A, B) -> C and (A2, B2) -> C2; A / A2 and C / C2 share strides
  int mmbh2(int epi, const T* A, const T* A2, int64_t lda, int akm, int64_t sA1, int64_t sA2, const T* B, int64_t ldb, int64_t sB1, int64_t sB2,
            const T* B2, int64_t ldb2, int64_t sB1_2, int64_t sB2_2, int bkm, void* C, void* C2, int64_t ldc, int64_t sC1, int64_t sC2, int nB, int M, int N,
            int K) const {
    BGemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.R = nullptr; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.a_kmajor = akm; g.b_kmajor = bkm;
    g.nb1 = nB; g.nb2 = t->g.num_heads; g.sA1 = sA1; g.sA2 = sA2; g.sB1 = sB1; g.sB2 = sB2; g.sC1 = sC1; g.sC2 = sC2; g.alpha = 1.0f;
    g.A2 = A2; g.B2 = B2; g.C2 = C2; g.ldb2 = ldb2; g.sB1_2 = sB1_2; g.sB2_2 = sB2_2;
    return launch_bgemm(t->precision, epi, g, st);
  }
  // dX[M, Kw] (epi) = dY[M, Nw] . W   for a weight stored [Nw][Kw]: an NT product against the transposed copy WT [Kw][Nw]
  int dX(int epi, const void* dY, int64_t ldy, int64_t w_off, int Nw, int Kw, void* C, int64_t ldc, int M) const {
    if (const m2m_trainer::LinW* w = t->fp8_dx ? lin8(w_off) : nullptr) {    // fp8 mode: dY in e5m2 (gradient format), W^T in e4m3
      MxGemmArgs m{};
      m.A = t->q8a; m.sA = t->s8a; m.B = t->w8 + w->qt; m.sB = t->w8 + w->qts; m.C = C; m.M = M; m.N = Kw; m.K = w->Np;
      m.lda = w->Np; m.ldb = w->Np; m.ldc = ldc;
      if (mxq_fused() && Nw % 8 == 0 && ldy % 8 == 0) {
        m.Asrc = dY; m.ld_src = ldy; m.Kvalid = Nw;
        return launch_mxgemm_q(t->grad_fmt, epi, m, st);
      }
      int rc = launch_mxq_rows(1, dY, ldy, t->q8a, t->s8a, M, Nw, w->Np, t->grad_fmt, st);
      if (rc != M2M_OK) return rc;
      return launch_mxgemm(t->grad_fmt, 0, epi, m, st);
    }
    return mm(epi, dY, ldy, 0, reinterpret_cast<const T*>(t->WT) + w_off, Nw, 0, C, ldc, M, Kw, Nw);
  }
  // G[Ny, Kx] = dY[M, Ny]^T . X[M, Kx]: both operands are transposed once (coalesced, through LDS) into [features][Mp]
  // scratch, then it is a plain NT product with the M rows as the reduction, split over k so that the few output tiles
  // of a weight gradient still fill the chip; the k-slices are summed in a fixed order.
  int dW(const void* dY, int64_t ldy, int Ny, const void* X, int64_t ldx, int Kx, float* Gout, int M) const {
    if (group) {
      const int Ea = t->precision == M2M_PREC_BF16 ? 8 : 4;
      if (Ny % Ea == 0 && Kx % Ea == 0 && ldy % Ea == 0 && ldx % Ea == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0 &&
          (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        DwProb p;
        memset(&p, 0, sizeof(p));                     // the table is compared bytewise: no uninitialised padding
        p.g.A = dY; p.g.B = X; p.g.C = Gout; p.g.N1 = Ny; p.g.N2 = Kx; p.g.K = M; p.g.lda = ldy; p.g.ldb = ldx; p.g.ldc = Kx; p.g.ksplit = 1; p.g.kchunk = M;
        probs.push_back(p);
        return M2M_OK;
      }
      return dW_on(st, dY, ldy, Ny, X, ldx, Kx, Gout, M);
    }
    if (!st2) return dW_on(st, dY, ldy, Ny, X, ldx, Kx, Gout, M);
    pending.push_back([=](hipStream_t s) { return dW_on(s, dY, ldy, Ny, X, ldx, Kx, Gout, M); });
    return M2M_OK;
  }
  int dW_on(hipStream_t st, const void* dY, int64_t ldy, int Ny, const void* X, int64_t ldx, int Kx, float* Gout, int M) const {
    if (Gbase && t->fp8_dw && lin8(Gout - Gbase)) {      // fp8 mode: dY^T (e5m2) . X^T (e4m3), blocks along the M rows
      const int Mp8 = (int)align_up(M, 128);
      int rc = launch_mxq_cols(1, dY, ldy, t->q8ta, t->s8ta, M, Ny, Mp8, t->grad_fmt, st);
      if (rc == M2M_OK) rc = launch_mxq_cols(1, X, ldx, t->q8tb, t->s8tb, M, Kx, Mp8, 0, st);
      if (rc != M2M_OK) return rc;
      MxGemmArgs m{};
      m.A = t->q8ta; m.sA = t->s8ta; m.B = t->q8tb; m.sB = t->s8tb; m.C = Gout; m.M = Ny; m.N = Kx; m.K = Mp8; m.lda = Mp8; m.ldb = Mp8; m.ldc = Kx;
      const int tiles = ceil_div(Ny, 64) * ceil_div(Kx, 64);
      int ks = 1024 / tiles;
      if (ks > 32) ks = 32;
      while (ks > 1 && ((int64_t)ks * Ny * Kx > t->kpart_floats || Mp8 / ks < 128)) --ks;
      if (ks > 1) { m.kchunk = (int)align_up(ceil_div(Mp8, ks), 128); m.ksplit = ceil_div(Mp8, m.kchunk); m.Cpart = t->kpart; }
      return launch_mxgemm(t->grad_fmt, 0, TG_STORE_F32, m, st);
    }
    // (the generic kernel's k-major staging — 2-byte LDS scatters — and the transpose-then-NT route it replaced cost
    //  ~30 us per weight gradient at 16 clips; M2M_TRAIN_DW_OLD=1 keeps them for comparison)
    static const bool old_path = getenv("M2M_TRAIN_DW_OLD") != nullptr;
    const int Ealign = t->precision == M2M_PREC_BF16 ? 8 : 4;
    if (!old_path && Ny % Ealign == 0 && Kx % Ealign == 0 && ldy % Ealign == 0 && ldx % Ealign == 0)
      return launch_dw_gemm(t->precision, dY, ldy, Ny, X, ldx, Kx, M, Gout, Kx, t->kpart, t->kpart_floats, st);
    const int Mp = (int)align_up(M, 8);
    BGemmArgs g{};
    g.C = Gout; g.M = Ny; g.N = Kx; g.K = M; g.ldc = Kx; g.nb1 = 1; g.nb2 = 1; g.alpha = 1.0f;
    if (dw_kmajor == 1 || (dw_kmajor < 0 && M < 8192)) {
      g.A = dY; g.B = X; g.lda = ldy; g.ldb = ldx; g.a_kmajor = 1; g.b_kmajor = 1;
    } else {
      hipLaunchKernelGGL((transpose_kernel<T, T>), dim3(ceil_div(Ny, 64), ceil_div(Mp, 64)), dim3(256), 0, st, (const T*)dY, ldy, (T*)t->tA, (int64_t)Mp, M, Ny, Mp);
      hipLaunchKernelGGL((transpose_kernel<T, T>), dim3(ceil_div(Kx, 64), ceil_div(Mp, 64)), dim3(256), 0, st, (const T*)X, ldx, (T*)t->tB, (int64_t)Mp, M, Kx, Mp);
      M2M_CHECK_HIP(hipGetLastError());
      g.A = t->tA; g.B = t->tB; g.lda = Mp; g.ldb = Mp;
    }
    const int tiles = ceil_div(Ny, TG_BM) * ceil_div(Kx, TG_BN);
    int ks = 1024 / tiles;
    if (ks > 32) ks = 32;
    while (ks > 1 && ((int64_t)ks * Ny * Kx > t->kpart_floats || ceil_div(M, ks) < 64)) --ks;
    if (ks > 1) { g.ksplit = ks; g.kchunk = (int)align_up(ceil_div(M, ks), TG_BK_MAX); g.ksplit = ceil_div(M, g.kchunk); g.Cpart = t->kpart; }
    return launch_bgemm(t->precision, TG_STORE_F32, g, st);
  }
  int cvt(const float* src, void* dst, int64_t n) const { return launch_cvt(t->precision, src, dst, n, st); }
  // (A row-complete residual product that carries the next sub-layer's RMSNorm in its epilogue — 32 whole rows per workgroup, the
  //  weight matrix streamed through every one of the 131 workgroups by DMA — was built and measured in round 3 (commit 0a97748,
  //  tools/shared_stream.hip for the streaming ceiling): 13.4 us at K = 512 and 23.9 us at K = 1 152 against 9.6 + 5.2 and 13.9 + 5.2 us
  //  for the product and the norm as two launches.  Per 64-deep step a workgroup moves 52 KB into LDS and reads 64 KB of fragments
  //  back: ~0.6 us of LDS and L2-to-CU time per step on HALF the CUs, where the tiled product spreads the same 54 MB over all of them.)
  int mm_resid(const void* A, int64_t w_off, float* x_out, int M, int N, int K, const float* x_in, int site) const {
    return mm(TG_RESID_F32, A, K, 0, W(w_off), K, 0, x_out, N, M, N, K, x_in, site);
  }
  int norm(const float* x, int64_t w_off, void* out, int M) const { return launch_rmsnorm(t->precision, x, P + w_off, out, M, t->g.d_model, t->g.layer_norm_eps, st); }
  int norm_drop(const float* x, int64_t w_off, void* out, int M, int site) const {
    if (!dropping(site)) return norm(x, w_off, out, M);
    hipLaunchKernelGGL(rmsnorm_drop_kernel<T>, dim3(ceil_div(M, 4)), dim3(256), 0, st, x, P + w_off, (T*)out, M, t->g.d_model, t->g.layer_norm_eps, key(site),
                       t->drop_thresh, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  int norm_bwd(const float* x, int64_t w_off, const float* dy, const float* dx_res, float* dx_out, float* G, int M, int dy_site = -1) const {
    const int d = t->g.d_model;
    // (summing the partials in the last block to finish, behind a __threadfence() + counter, was measured: the agent-scope
    //  fence of 256 blocks costs ~100 us per launch on this machine — 8.7 -> 12.3 ms per step; the second launch stays)
    // grouped mode: the partial image of every norm goes to a slice of its own and ONE launch sums them all after the backward
    // pass (flush_norms); otherwise the column sum follows right away
    const int slot = group ? norm_slot_base + (int)norm_offs.size() : 0;
    float* part = t->dw_part + (int64_t)slot * RN_BLOCKS * d;
    T* out_t = nullptr;
    DropKey dk{nullptr, 0};
    uint32_t thr = 0;
    if (group && after_site != -2) {
      out_t = (T*)t->ring[m2m_trainer::K_DXT][pos[m2m_trainer::K_DXT] % t->ring[m2m_trainer::K_DXT].size()];      // what the next begin_sub() selects
      if (dropping(after_site)) { dk = key(after_site); thr = t->drop_thresh; }
      pre.src = dx_out; pre.dst = out_t; pre.site = after_site;
    }
    const bool din = dropping(dy_site);
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<T>, dim3(RN_BLOCKS), dim3(256), (size_t)4 * d * sizeof(float), st, x, P + w_off, dy, dx_res, dx_out,
                       part, M, d, t->g.layer_norm_eps, out_t, dk, thr, t->drop_scale, din ? key(dy_site) : DropKey{nullptr, 0}, din ? t->drop_thresh : 0u);
    if (group) norm_offs.push_back(w_off);
    else hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(d, 32)), dim3(256), 0, st, part, G + w_off, RN_BLOCKS, d, 0);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  // Whole-head attention kernels (attn_train.hip): bf16 storage, both lengths within an LDS image.  M2M_TRAIN_ATTN=stripes keeps round 2's path.
  static bool head_on() { const char* v = getenv("M2M_TRAIN_ATTN"); return !(v && v[0] == 's'); }      // (read per pass: tests run both paths in one process)
  bool head_ok(int Sq, int Sk) const { return sizeof(T) == 2 && head_on() && use_tuned && Sq <= AH_MAX_S && Sk <= AH_MAX_S; }
  HeadAttnArgs head_args(const void* Q, int64_t ldq, int64_t sQb, const void* K, int64_t ldk, const void* V, int64_t ldv, int64_t sKb, void* O, float* lse,
                         int Sq, int Sk, const float* tab, int causal, int site, void* keep_bits) const {
    HeadAttnArgs a{};
    const int inner = t->inner;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sQb = sQb; a.sKb = sKb; a.sVb = sKb;
    a.O = (bf16_t*)O; a.ldo = inner; a.sOb = (int64_t)Sq * inner; a.lse = lse;
    a.bias_tab = tab; a.tab_stride = Sq + Sk - 1; a.tab_center = Sq - 1;
    a.H = t->g.num_heads; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.ldp = (int)align_up(Sk, 8);
    const bool dr = dropping(site);
    a.dk = dr ? key(site) : DropKey{nullptr, 0}; a.thresh = dr ? t->drop_thresh : 0u; a.scale = t->drop_scale;
    a.keep_bits = (uint32_t*)keep_bits;      // (the layer's probability buffer, unused on this path: B*H*Sq*Sk/8 bytes of its B*H*Sq*ldp*2)
    return a;
  }
  // Fused scores + softmax (attn_stripe_kernel) when a wave can hold all keys; K = (key, d) operand, Q = (query, d) operand.
  static bool stripes_on() { static const bool on = [] { const char* v = getenv("M2M_TRAIN_STRIPES"); return !(v && v[0] == '0'); }(); return on; }
  bool stripe_ok(int Sk) const {      // the bias row + (two, with dropout) 32-row blocks of P must fit the LDS
    return stripes_on() && Sk <= 32 * ST_NT && 2 * (size_t)32 * (ceil_div(Sk, 32) * 32 + 8) * sizeof(T) + 8192 <= 158 * 1024;
  }
  // K | V of a layer ([rows, ld], K at column 0 and V at column `inner` of `kv`) -> kt [2][nB*H][64][Sp]
  int kv_transpose(const T* kv, int64_t ld, void* kt, int nB, int S) const {
    const int H = t->g.num_heads, Sp = (int)align_up(S, 32);
    constexpr int Et = 16 / sizeof(T);
    hipLaunchKernelGGL(kv_transpose_kernel<T>, dim3(ceil_div((Sp / Et) * (DK / Et), 256), nB * H, 2), dim3(256), 0, st, kv, ld, (int64_t)t->inner, (T*)kt, S, Sp, H,
                       (int64_t)nB * H * DK * Sp);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  // M2M_TRAIN_FUSE_PV: 0 = off, fwd / bwd = only that pass (diagnostics), default both
  static int fuse_mode() { static const int m = [] { const char* v = getenv("M2M_TRAIN_FUSE_PV"); return !v ? 3 : v[0] == '0' ? 0 : v[0] == 'f' ? 1 : v[0] == 'b' ? 2 : 3; }(); return m; }
  static bool fuse_on() { return fuse_mode() != 0; }
  static bool pair_on() { static const bool on = [] { const char* v = getenv("M2M_TRAIN_PAIR_DVDK"); return !(v && v[0] == '0'); }(); return on; }
  // Xt / O: the fused product of the stripe kernel (StripeArgs), or null
  int attn_probs(const T* K, int64_t ldk, int64_t sK1, int64_t sK2, const T* Q, int64_t ldq, int64_t sQ1, int64_t sQ2, void* Pm, int nB, int Sq, int Sk,
            int ldp, const float* tab, int causal, int site, const T** Puse, const T* Xt = nullptr, void* O = nullptr, int64_t ldo = 0, int64_t sO1 = 0,
            int64_t sO2 = 0) const {
    const bool dr = dropping(site);
    StripeArgs a{};
    a.X = K; a.ldx = ldk; a.sX1 = sK1; a.sX2 = sK2; a.Y = Q; a.ldy = ldq; a.sY1 = sQ1; a.sY2 = sQ2;
    static const bool keep_pd = [] { const char* v = getenv("M2M_TRAIN_FWD_PD"); return v && v[0] == '1'; }();      // 1: always write the dropped copy (measurement)
    a.P = Pm; a.Pd = (dr && (keep_pd || !(Xt && O))) ? t->dS : nullptr;      // fused P.V: the dropped copy never leaves the kernel
    a.bias_tab = tab; a.tab_stride = Sq + Sk - 1; a.tab_center = Sq - 1;
    a.H = t->g.num_heads; a.Sq = Sq; a.Sk = Sk; a.ldp = ldp; a.causal = causal;
    a.dk = dr ? key(site) : DropKey{nullptr, 0}; a.thresh = dr ? t->drop_thresh : 0u; a.scale = t->drop_scale;
    a.Xt = Xt; a.xt_ld = align_up(Sk, 32); a.O = O; a.ldo = ldo; a.sO1 = sO1; a.sO2 = sO2;
    *Puse = dr ? (const T*)t->dS : (const T*)Pm;
    return launch_attn_stripe<T>(false, a, nB, st);
  }
  // Fused dP + softmax backward: V = (key, d) operand, dO = (query, d) operand; dS out
  int dscores(const T* V, int64_t ldv, int64_t sV1, int64_t sV2, const T* dO, int64_t ldo, int64_t sO1, int64_t sO2, const void* Pm, void* dS, int nB,
              int Sq, int Sk, int ldp, int site, bool want_diag = false, const T* Xt = nullptr, void* Out = nullptr, int64_t ld_out = 0, int64_t s1_out = 0,
              int64_t s2_out = 0, int causal = 0, void* Pd_out = nullptr) const {
    const bool dr = dropping(site);
    StripeArgs a{};
    a.Pd = dr ? Pd_out : nullptr;        // the dropped probabilities again, for the dV product (instead of a drop_copy launch)
    a.X = V; a.ldx = ldv; a.sX1 = sV1; a.sX2 = sV2; a.Y = dO; a.ldy = ldo; a.sY1 = sO1; a.sY2 = sO2;
    a.P = const_cast<void*>(Pm); a.dS = dS; a.H = t->g.num_heads; a.Sq = Sq; a.Sk = Sk; a.ldp = ldp;
    a.causal = causal;                   // only lets the kernel skip key tiles above the diagonal (P is zero there anyway)
    a.diag_part = want_diag ? drel_slot() : nullptr;
    a.Xt = Xt; a.xt_ld = align_up(Sk, 32); a.O = Out; a.ldo = ld_out; a.sO1 = s1_out; a.sO2 = s2_out;
    a.dk = dr ? key(site) : DropKey{nullptr, 0}; a.thresh = dr ? t->drop_thresh : 0u; a.scale = t->drop_scale;
    return launch_attn_stripe<T>(true, a, nB, st);
  }
  // P (kept for the backward) and, with dropout, the dropped copy the P.V product reads (scratch: t->dS); returns it through *Puse
  int softmax(const float* sc, void* Pm, int nB, int Sq, int Sk, int ldp, const float* tab, int causal, int site, const T** Puse) const {
    const int H = t->g.num_heads, rows = nB * H * Sq;
    const bool dr = dropping(site);
    hipLaunchKernelGGL(softmax_fwd_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, sc, (T*)Pm, rows, H, Sq, Sk, ldp, tab, Sq + Sk - 1,
                       Sq - 1, causal, dr ? (T*)t->dS : (T*)nullptr, dr ? key(site) : DropKey{nullptr, 0}, dr ? t->drop_thresh : 0u, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    *Puse = dr ? (const T*)t->dS : (const T*)Pm;
    return M2M_OK;
  }
  // backward: the dropped probabilities again (into t->dS, consumed by the dV product before dS overwrites it)
  // (into_sc: the dropped copy goes to the fp32 score scratch, unused on the stripe path, so that it survives the dS written later
  //  and dV can share a launch with dK)
  static bool pd_fuse_on() { static const bool on = [] { const char* v = getenv("M2M_TRAIN_FUSE_PD"); return !(v && v[0] == '0'); }(); return on; }
  int redrop(const void* Pm, int64_t n, int site, const T** Puse, bool into_sc = false) const {
    if (!dropping(site)) { *Puse = (const T*)Pm; return M2M_OK; }
    T* dst = into_sc ? (T*)t->sc : (T*)t->dS;
    hipLaunchKernelGGL(drop_copy_kernel<T>, dim3(grid_1d(n)), dim3(256), 0, st, (const T*)Pm, dst, n, key(site), t->drop_thresh, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    *Puse = dst;
    return M2M_OK;
  }
  int softmax_bwd(const void* Pm, const float* dP, void* dS, int rows, int Sk, int ldp, int site) const {
    const bool dr = dropping(site);
    hipLaunchKernelGGL(softmax_bwd_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, st, (const T*)Pm, dP, (T*)dS, rows, Sk, ldp,
                       dr ? key(site) : DropKey{nullptr, 0}, dr ? t->drop_thresh : 0u, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  int gated(const void* ab, void* mid, int64_t M, int site) const {
    const bool dr = dropping(site);
    hipLaunchKernelGGL(gated_fwd_kernel<T>, dim3(grid_1d(M * t->g.d_ff / 4)), dim3(256), 0, st, (const T*)ab, (T*)mid, M, t->g.d_ff,
                       dr ? key(site) : DropKey{nullptr, 0}, dr ? t->drop_thresh : 0u, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  int gated_bwd(const void* ab, const void* dmid, void* dab, int64_t M, int site) const {
    const bool dr = dropping(site);
    hipLaunchKernelGGL(gated_bwd_kernel<T>, dim3(grid_1d(M * t->g.d_ff / 4)), dim3(256), 0, st, (const T*)ab, (const T*)dmid, (T*)dab, M, t->g.d_ff,
                       dr ? key(site) : DropKey{nullptr, 0}, dr ? t->drop_thresh : 0u, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  // bias gradient from the per-stripe diagonal sums the backward stripe kernel left in t->drel
  float* drel_slot() const { return t->drel + (int64_t)(group ? drel_slots : 0) * t->drel_slot_floats; }      // where the next stripe launch leaves its sums
  int bias_reduce(const BiasJob& j, int accumulate) const {
    const int H = t->g.num_heads, nrel = j.Sq + j.Sk - 1, stripes = ceil_div(j.Sq, 32);
    float* tmp = t->drel + (int64_t)(t->g.num_layers + t->g.num_decoder_layers) * t->drel_slot_floats;
    hipLaunchKernelGGL(bias_stripes_sum_kernel, dim3(H, ceil_div(nrel, 64), j.layers), dim3(1024), 0, st, t->drel + (int64_t)j.first_slot * t->drel_slot_floats,
                       tmp, j.nB, H, stripes, j.Sq, j.Sk, t->drel_slot_floats);
    hipLaunchKernelGGL(bias_bucket_kernel, dim3(t->g.num_buckets * H), dim3(256), 0, st, tmp, j.buckets, j.Gtab, j.layers, H, nrel, accumulate);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
  // grouped mode: the layers of a stack share one table, so their sums are reduced together after the backward pass (two
  // launches per stack instead of two per layer); otherwise right away
  int bias_grad_stripes(const int* buckets, float* Gtab, int nB, int Sq, int Sk, int accumulate) const {
    if (!group) return bias_reduce(BiasJob{buckets, Gtab, nB, Sq, Sk, 0, 1}, accumulate);
    if (!bias_jobs.empty() && bias_jobs.back().buckets == buckets && bias_jobs.back().Gtab == Gtab) bias_jobs.back().layers += 1;
    else bias_jobs.push_back(BiasJob{buckets, Gtab, nB, Sq, Sk, drel_slots, 1});
    drel_slots += 1;
    return M2M_OK;
  }
  int flush_bias() const {
    for (const BiasJob& j : bias_jobs) { const int rc = bias_reduce(j, 0); if (rc != M2M_OK) return rc; }
    bias_jobs.clear();
    return M2M_OK;
  }
  int bias_grad(const void* dS, const int* buckets, float* Gtab, int nB, int Sq, int Sk, int ldp, int accumulate) const {
    const int H = t->g.num_heads, nrel = Sq + Sk - 1;
    // scratch of its own, behind the stripe kernels' slots and their reduction scratch: those may hold other layers' sums
    // that wait for the end of the pass (a model whose encoder is past the stripe kernels' reach and whose decoder is not)
    float* part = t->drel + (int64_t)(t->g.num_layers + t->g.num_decoder_layers) * t->drel_slot_floats +
                  (int64_t)std::max(t->g.num_layers, t->g.num_decoder_layers) * H * 2 * std::max(t->max_enc, t->max_dec);
    hipLaunchKernelGGL(bias_diag_kernel<T>, dim3(nB * H, ceil_div(nrel, 64)), dim3(256), 0, st, (const T*)dS, part, H, Sq, Sk, ldp);
    hipLaunchKernelGGL(bias_bucket_kernel, dim3(t->g.num_buckets * H), dim3(256), 0, st, part, buckets, Gtab, nB, H, nrel, accumulate);
    M2M_CHECK_HIP(hipGetLastError());
    return M2M_OK;
  }
};

#define RC(expr) do { if ((rc = (expr)) != M2M_OK) return rc; } while (0)

// dropout sites (hf: modeling_t5.py — T5Stack dropout on the embeddings and after the final norm, T5Attention on the
// probabilities, T5LayerSelfAttention / CrossAttention / FF on the branch output, T5DenseGatedActDense before wo).
// site = stack base + 16 * layer + place; oracle/train.py uses the same numbering.
enum { SITE_ENC = 0, SITE_DEC = 1000, SITE_EMB = 900, SITE_FIN = 901,
       PL_PROBS_SELF = 1, PL_SELF_OUT = 2, PL_PROBS_CROSS = 3, PL_CROSS_OUT = 4, PL_MID = 5, PL_FF_OUT = 6 };

// self-attention block, forward: x_in -> x_out = x_in + Attn(norm(x_in)).  Buffers of this layer are passed in.
template <typename T>
int attn_self_fwd(const Ops<T>& o, const float* x_in, float* x_out, int64_t ln, int64_t wqkv, int64_t wo, void* h, void* qkv, void* Pm, void* ao,
                  int nB, int S, const float* tab, int causal, int site0, void* kt, float* lse) {
  m2m_trainer* t = o.t;
  const int d = t->g.d_model, inner = t->inner, M = nB * S, ldp = (int)align_up(S, 8), H = t->g.num_heads;
  int rc;
  RC(o.norm(x_in, ln, h, M));
  RC(o.mm(TG_STORE_T, h, d, 0, o.W(wqkv), d, 0, qkv, 3 * inner, M, 3 * inner, d));
  const T* q = (const T*)qkv;
  if (lse && o.head_ok(S, S)) {                                  // whole-head kernel: no P, no K^T | V^T, only O and the row log-sum-exp
    const HeadAttnArgs a = o.head_args(q, 3 * inner, (int64_t)S * 3 * inner, q + inner, 3 * inner, q + 2 * inner, 3 * inner, (int64_t)S * 3 * inner, ao, lse, S, S,
                                       tab, causal, site0 + PL_PROBS_SELF, Pm);
    RC(launch_attn_head_fwd(a, nB, o.st));
    RC(o.mm_resid(ao, wo, x_out, M, d, inner, x_in, site0 + PL_SELF_OUT));
    return M2M_OK;
  }
  const T* Pu;
  const bool fuse = o.stripe_ok(S) && o.fuse_on() && kt;         // P . V inside the stripe kernel, against the transposed V
  const bool fuse_pv = fuse && (o.fuse_mode() & 1);
  const T* vt = (const T*)kt + (int64_t)nB * H * DK * align_up(S, 32);
  // (K^T | V^T from the projection's own epilogue was built twice in round 3 — the tile staged through LDS, then the transpose taken
  //  from the matrix core with the operands swapped — and both cost the projection as much as these launches take: 64-byte row
  //  pieces instead of whole lines; removed again)
  if (fuse) RC(o.kv_transpose(q + inner, 3 * inner, kt, nB, S));
  if (o.stripe_ok(S)) {
    RC(o.attn_probs(q + inner, 3 * inner, (int64_t)S * 3 * inner, DK, q, 3 * inner, (int64_t)S * 3 * inner, DK, Pm, nB, S, S, ldp, tab, causal,
               site0 + PL_PROBS_SELF, &Pu, fuse_pv ? vt : nullptr, ao, inner, (int64_t)S * inner, DK));
  } else {
    RC(o.mmbh(TG_STORE_F32, q, 3 * inner, 0, (int64_t)S * 3 * inner, DK, q + inner, 3 * inner, 0, (int64_t)S * 3 * inner, DK, t->sc, ldp,
              (int64_t)H * S * ldp, (int64_t)S * ldp, nB, S, S, DK));
    RC(o.softmax(t->sc, Pm, nB, S, S, ldp, tab, causal, site0 + PL_PROBS_SELF, &Pu));
  }
  if (!fuse_pv)
    RC(o.mmbh(TG_STORE_T, Pu, ldp, 0, (int64_t)H * S * ldp, (int64_t)S * ldp, q + 2 * inner, 3 * inner, 1, (int64_t)S * 3 * inner, DK, ao,
              inner, (int64_t)S * inner, DK, nB, S, DK, S));
  RC(o.mm_resid(ao, wo, x_out, M, d, inner, x_in, site0 + PL_SELF_OUT));
  return M2M_OK;
}

// ... and backward: dx_out (fp32, gradient wrt x_out) -> dx_in; weight gradients into G
template <typename T>
int attn_self_bwd(const Ops<T>& o, const float* x_in, const float* dx_out, float* dx_in, float* G, int64_t ln, int64_t wqkv, int64_t wo,
                  const void* h, const void* qkv, const void* Pm, const void* ao, int nB, int S, const int* buckets, int64_t bias_off,
                  int bias_accumulate, int site0, const void* kt, float* lse, const float* tab) {
  m2m_trainer* t = o.t;
  const int d = t->g.d_model, inner = t->inner, M = nB * S, ldp = (int)align_up(S, 8), H = t->g.num_heads;
  int rc;
  RC(o.begin_sub(1u << m2m_trainer::K_DXT | 1u << m2m_trainer::K_DQKV));
  RC(o.cvt_branch(dx_out, t->dxT, (int64_t)M * d, site0 + PL_SELF_OUT));
  RC(o.dW(t->dxT, d, d, ao, inner, inner, G + wo, M));                                            // dWo = dx^T . ao
  RC(o.dX(TG_STORE_T, t->dxT, d, wo, d, inner, t->dO, inner, M));                                 // dO = dx . Wo
  const T* q = (const T*)qkv;
  const T* dO = (const T*)t->dO;
  T* dq = (T*)t->dqkv;
  const int64_t sP1 = (int64_t)H * S * ldp, sP2 = (int64_t)S * ldp, sQ1 = (int64_t)S * 3 * inner, sO1 = (int64_t)S * inner;
  if (lse && o.head_ok(S, S)) {                                  // whole-head kernel: dQ | dK | dV (and the bias gradient's diagonal sums) in one launch
    HeadAttnArgs a = o.head_args(q, 3 * inner, sQ1, q + inner, 3 * inner, q + 2 * inner, 3 * inner, sQ1, const_cast<void*>(ao), lse, S, S, tab,
                                 buckets == t->dbucket ? 1 : 0, site0 + PL_PROBS_SELF, const_cast<void*>((const void*)Pm));
    a.dO = (const bf16_t*)dO;
    a.dQ = (bf16_t*)dq; a.dK = (bf16_t*)(dq + inner); a.dV = (bf16_t*)(dq + 2 * inner);
    a.lddq = a.lddk = a.lddv = 3 * inner; a.sdQb = a.sdKb = a.sdVb = sQ1;
    a.diag_part = buckets ? o.drel_slot() : nullptr;
    RC(launch_attn_head_bwd(a, nB, o.st));
    if (buckets) RC(o.bias_grad_stripes(buckets, G + bias_off, nB, S, S, bias_accumulate));
    RC(o.dW(dq, 3 * inner, 3 * inner, h, d, d, G + wqkv, M));                                     // dWqkv = dqkv^T . h
    RC(o.dX(TG_STORE_F32, dq, 3 * inner, wqkv, 3 * inner, d, t->dh, d, M));                       // dh = dqkv . Wqkv
    RC(o.norm_bwd(x_in, ln, t->dh, dx_out, dx_in, G, M));
    RC(o.end_sub());
    return M2M_OK;
  }
  const T* Pu;
  const bool pair = o.stripe_ok(S) && o.pair_on();            // dV and dK in one launch (after dS exists)
  const bool pd_fused = pair && o.dropping(site0 + PL_PROBS_SELF) && o.pd_fuse_on();      // the stripe kernel re-emits the dropped P itself
  if (pd_fused) Pu = (const T*)t->sc;
  else RC(o.redrop(Pm, (int64_t)nB * H * S * ldp, site0 + PL_PROBS_SELF, &Pu, pair));
  if (!pair) RC(o.mmbh(TG_STORE_T, Pu, ldp, 1, sP1, sP2, dO, inner, 1, sO1, DK, dq + 2 * inner, 3 * inner, sQ1, DK, nB, S, DK, S));  // dV = Pd^T dO
  const bool fuse = o.stripe_ok(S) && (o.fuse_mode() & 2) && kt;  // dQ = dS . K inside the stripe kernel, against the transposed K
  if (o.stripe_ok(S)) {
    RC(o.dscores(q + 2 * inner, 3 * inner, sQ1, DK, dO, inner, sO1, DK, Pm, t->dS, nB, S, S, ldp, site0 + PL_PROBS_SELF, buckets != nullptr,   // dS from dPd = dO V^T
                 fuse ? (const T*)kt : nullptr, dq, 3 * inner, sQ1, DK, buckets == t->dbucket ? 1 : 0, pd_fused ? t->sc : nullptr));
    if (buckets) RC(o.bias_grad_stripes(buckets, G + bias_off, nB, S, S, bias_accumulate));
  } else {
    RC(o.mmbh(TG_STORE_F32, dO, inner, 0, sO1, DK, q + 2 * inner, 3 * inner, 0, sQ1, DK, t->sc, ldp, sP1, sP2, nB, S, S, DK));      // dPd = dO V^T
    RC(o.softmax_bwd(Pm, t->sc, t->dS, nB * H * S, S, ldp, site0 + PL_PROBS_SELF));
    if (buckets) RC(o.bias_grad(t->dS, buckets, G + bias_off, nB, S, S, ldp, bias_accumulate));
  }
  const T* dS = (const T*)t->dS;
  if (!fuse) RC(o.mmbh(TG_STORE_T, dS, ldp, 0, sP1, sP2, q + inner, 3 * inner, 1, sQ1, DK, dq, 3 * inner, sQ1, DK, nB, S, DK, S));   // dQ = dS K
  if (pair)                                                                                                                          // dV = Pd^T dO | dK = dS^T Q
    RC(o.mmbh2(TG_STORE_T, Pu, dS, ldp, 1, sP1, sP2, dO, inner, sO1, DK, q, 3 * inner, sQ1, DK, 1, dq + 2 * inner, dq + inner, 3 * inner, sQ1, DK, nB, S, DK, S));
  else
    RC(o.mmbh(TG_STORE_T, dS, ldp, 1, sP1, sP2, q, 3 * inner, 1, sQ1, DK, dq + inner, 3 * inner, sQ1, DK, nB, S, DK, S));            // dK = dS^T Q
  RC(o.dW(dq, 3 * inner, 3 * inner, h, d, d, G + wqkv, M));                                       // dWqkv = dqkv^T . h
  RC(o.dX(TG_STORE_F32, dq, 3 * inner, wqkv, 3 * inner, d, t->dh, d, M));                         // dh = dqkv . Wqkv
  RC(o.norm_bwd(x_in, ln, t->dh, dx_out, dx_in, G, M));
  RC(o.end_sub());
  return M2M_OK;
}

template <typename T>
int ff_fwd(const Ops<T>& o, const float* x_in, float* x_out, int64_t ln, int64_t wi, int64_t wo, void* h, void* ab, void* mid, int M, int site0) {
  m2m_trainer* t = o.t;
  const int d = t->g.d_model, dff = t->g.d_ff;
  int rc;
  RC(o.norm(x_in, ln, h, M));
  // the gate product with the activation in its epilogue (bf16, grids the 128x128 tile takes anyway): a | b and
  // mid = dropout(gelu_new(a) * b) leave the product together — the gated_fwd_kernel launch and its read of the pair are gone
  static const bool gate_epi = [] { const char* v = getenv("M2M_TRAIN_GATE_EPI"); return !(v && v[0] == '0'); }();
  if (gate_epi && !t->fp8 && o.use_tuned && t->Wil && gemm_takes_gated_train(t->precision, M, 2 * dff, d)) {
    GemmArgs a{};
    a.A = h; a.W = reinterpret_cast<const T*>(t->Wil) + wi; a.M = M; a.N = 2 * dff; a.K = d; a.out = mid; a.ldo = dff; a.ab_out = ab; a.vt_which = -1;
    if (o.dropping(site0 + PL_MID)) { a.drop_thresh = t->drop_thresh; a.drop_scale = t->drop_scale; a.drop_key = o.key(site0 + PL_MID).salt; a.drop_step = t->step_key_dev; }
    RC(launch_gemm(t->precision, EPI_GATED_TRAIN, a, o.st));
  } else {
    RC(o.mm(TG_STORE_T, h, d, 0, o.W(wi), d, 0, ab, 2 * dff, M, 2 * dff, d));
    RC(o.gated(ab, mid, M, site0 + PL_MID));
  }
  RC(o.mm_resid(mid, wo, x_out, M, d, dff, x_in, site0 + PL_FF_OUT));
  return M2M_OK;
}
template <typename T>
int ff_bwd(const Ops<T>& o, const float* x_in, const float* dx_out, float* dx_in, float* G, int64_t ln, int64_t wi, int64_t wo, const void* h,
           const void* ab, const void* mid, int M, int site0) {
  m2m_trainer* t = o.t;
  const int d = t->g.d_model, dff = t->g.d_ff;
  int rc;
  RC(o.begin_sub(1u << m2m_trainer::K_DXT | 1u << m2m_trainer::K_DAB));
  RC(o.cvt_branch(dx_out, t->dxT, (int64_t)M * d, site0 + PL_FF_OUT));
  RC(o.dW(t->dxT, d, d, mid, dff, dff, G + wo, M));                                               // dWo = dx^T . mid
  // dmid = dx . Wo, turned into the gate pair's gradient by the product's own epilogue where it can (bf16): dmid is never stored,
  // the gated_bwd_kernel launch and its reads are gone
  static const bool gate_epi = [] { const char* v = getenv("M2M_TRAIN_GATE_EPI"); return !(v && v[0] == '0'); }();
  if (gate_epi && !t->fp8 && o.use_tuned && gemm_takes_gated_bwd(t->precision, M, dff, d)) {
    GemmArgs a{};
    a.A = t->dxT; a.W = reinterpret_cast<const T*>(t->WT) + wo; a.M = M; a.N = dff; a.K = d; a.out = t->dab; a.ldo = 2 * dff; a.ab_out = const_cast<void*>(ab);
    a.vt_which = -1;
    if (o.dropping(site0 + PL_MID)) { a.drop_thresh = t->drop_thresh; a.drop_scale = t->drop_scale; a.drop_key = o.key(site0 + PL_MID).salt; a.drop_step = t->step_key_dev; }
    RC(launch_gemm(t->precision, EPI_GATED_BWD, a, o.st));
  } else {
    RC(o.dX(TG_STORE_T, t->dxT, d, wo, d, dff, t->dmid, dff, M));                                 // dmid = dx . Wo
    RC(o.gated_bwd(ab, t->dmid, t->dab, M, site0 + PL_MID));
  }
  RC(o.dW(t->dab, 2 * dff, 2 * dff, h, d, d, G + wi, M));                                         // dWi = dab^T . h
  RC(o.dX(TG_STORE_F32, t->dab, 2 * dff, wi, 2 * dff, d, t->dh, d, M));                           // dh = dab . Wi
  RC(o.norm_bwd(x_in, ln, t->dh, dx_out, dx_in, G, M));
  RC(o.end_sub());
  return M2M_OK;
}

// Everything a pass needs before its first product and that depends on nothing but the inputs, in ONE launch.  (The pad ranges: the
// flat gradient buffer is OVERWRITTEN by every pass — each tensor in full, by its own product / reduction — so all that needs zeroing
// is the alignment padding between tensors, <= 63 floats each, instead of a 121 MB memset; tests fill the buffer with NaN before a
// pass, so a tensor nobody wrote would show.)  In ONE launch (they were six:
// two bias-table gathers, the dropout key, the pad zeroing, the valid-label count, the decoder inputs — ~5 us of launch each for
// microseconds of work).  Block roles by index: [0, nb_e) encoder bias table, [nb_e, nb_e + nb_d) decoder bias table, then the pad
// ranges (one wave each), then shift_right; block 0 also advances the dropout key (thread 0) and, the LAST block counts
// the scored labels (a single-block reduction).
struct PrologueArgs {
  const float *w_e, *w_d;              // relative-position-bias weights [buckets][H]
  const int *bucket_e, *bucket_d;
  float *tab_e, *tab_d;
  int H, nrel_e, nrel_d, nb_e, nb_d;
  uint64_t seed;
  uint64_t *ctr, *key;
  const int64_t* pads;
  int n_pads, nb_p;
  float* G;                            // null: forward only (no pads)
  const int64_t* labels;
  int64_t* dec_in;
  int B, L, start_id, pad_id, nb_s;
  float* inv_n;
};
__global__ __launch_bounds__(256) void train_prologue_kernel(PrologueArgs a) {
  int blk = blockIdx.x;
  if (blk == 0 && threadIdx.x == 0) {                      // key of this pass = splitmix64(seed + passes since set_dropout); the counter lives on the device so that a replayed graph advances it
    const uint64_t c = *a.ctr;
    *a.key = splitmix64(a.seed + c);
    *a.ctr = c + 1;
  }
  if (blk < a.nb_e + a.nb_d) {                             // bias tables from the CURRENT (trainable) bucket weights
    const bool dec = blk >= a.nb_e;
    const int idx = (dec ? blk - a.nb_e : blk) * 256 + threadIdx.x, nrel = dec ? a.nrel_d : a.nrel_e;
    if (idx < a.H * nrel) {
      const int hh = idx / nrel, i = idx - hh * nrel;
      (dec ? a.tab_d : a.tab_e)[idx] = (dec ? a.w_d : a.w_e)[(int64_t)(dec ? a.bucket_d : a.bucket_e)[i] * a.H + hh];
    }
    return;
  }
  blk -= a.nb_e + a.nb_d;
  if (blk < a.nb_p) {                                      // alignment padding of the flat gradient buffer
    const int i = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i < a.n_pads) {
      const int64_t off = a.pads[2 * i], cnt = a.pads[2 * i + 1];
      for (int64_t j = lane; j < cnt; j += 64) a.G[off + j] = 0.f;
    }
    return;
  }
  blk -= a.nb_p;
  if (blk < a.nb_s) {                                      // decoder inputs = shift_right(labels)
    const int i = blk * 256 + threadIdx.x;
    if (i < a.B * a.L) {
      const int tpos = i % a.L;
      int64_t v = tpos == 0 ? a.start_id : a.labels[i - 1];
      if (v == -100) v = a.pad_id;
      a.dec_in[i] = v;
    }
    return;
  }
  // last block: number of scored labels
  __shared__ int cnt[256];
  int c = 0;
  const int n = a.B * a.L;
  for (int i = threadIdx.x; i < n; i += 256) c += a.labels[i] != -100;
  cnt[threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) cnt[threadIdx.x] += cnt[threadIdx.x + st];
    __syncthreads();
  }
  // no label to score: the mean over zero rows is NaN, as torch's CrossEntropyLoss gives (its gradient is zero there too)
  if (threadIdx.x == 0) { a.inv_n[0] = cnt[0] > 0 ? 1.0f / (float)cnt[0] : __builtin_nanf(""); a.inv_n[1] = (float)cnt[0]; }
}

template <typename T>
int forward_backward_t(m2m_trainer* t, const float* P, const float* enc_inputs, const int64_t* cond_idx, const int64_t* labels, int B, int S,
                       int L, float* loss_out, float* G, float* logits_out, hipStream_t st, hipStream_t st_side,
                       const std::function<int()>* at_split = nullptr) {
  const m2m_t5_geometry& g = t->g;
  const int d = g.d_model, inner = t->inner, V = g.vocab_size, H = g.num_heads, Le = g.num_layers, Ld = g.num_decoder_layers;
  const int Me = B * S, Md = B * L, lps = (int)align_up(S, 8), ldv = (int)align_up(V, 8);
  Ops<T> o{t, st, P};
  o.Gbase = G;
  o.group = G && t->use_group && !(t->fp8 && t->fp8_dw);
  o.st2 = (G && !o.group) ? st_side : nullptr;
  int rc;
  // (the bucket tables of this geometry are on the device already: ensure_tables())
  {
    PrologueArgs a{};
    a.w_e = P + t->o_erb; a.w_d = P + t->o_drb; a.bucket_e = t->ebucket; a.bucket_d = t->dbucket; a.tab_e = t->etab; a.tab_d = t->dtab;
    a.H = H; a.nrel_e = 2 * S - 1; a.nrel_d = 2 * L - 1; a.nb_e = ceil_div(H * a.nrel_e, 256); a.nb_d = ceil_div(H * a.nrel_d, 256);
    a.seed = t->drop_seed; a.ctr = t->step_ctr_dev; a.key = t->step_key_dev;
    a.pads = t->pads_dev; a.n_pads = G ? t->n_pads : 0; a.nb_p = ceil_div(a.n_pads, 4); a.G = G;
    a.labels = labels; a.dec_in = t->dec_in; a.B = B; a.L = L; a.start_id = g.decoder_start_token_id; a.pad_id = g.pad_token_id;
    a.nb_s = ceil_div(Md, 256); a.inv_n = t->inv_n;
    hipLaunchKernelGGL(train_prologue_kernel, dim3(a.nb_e + a.nb_d + a.nb_p + a.nb_s + 1), dim3(256), 0, st, a);
    M2M_CHECK_HIP(hipGetLastError());
  }
  if (G || t->precision == M2M_PREC_BF16) {      // W^T for the dX products (with gradients) and the bf16 copy the forward products read, one pass over P
    hipLaunchKernelGGL(weights_transpose_kernel<T>, dim3(t->n_wt_blocks), dim3(256), 0, st, (const WtBlock*)t->wt_blocks, P, G ? (T*)t->WT : (T*)nullptr,
                       t->precision == M2M_PREC_BF16 ? (T*)t->Wc : (T*)nullptr, (t->precision == M2M_PREC_BF16 && !t->fp8) ? (T*)t->Wil : (T*)nullptr);
    M2M_CHECK_HIP(hipGetLastError());
  }
  if (t->fp8) {
    hipLaunchKernelGGL(mxq_weights_kernel, dim3(t->n_w8_tiles), dim3(64), 0, st, (const W8Tile*)t->w8_tiles, P, t->w8);
    M2M_CHECK_HIP(hipGetLastError());
  }

  // ================= forward =================
  if (enc_inputs != t->xe[0]) M2M_CHECK_HIP(hipMemcpyAsync(t->xe[0], enc_inputs, (size_t)Me * d * 4, hipMemcpyDeviceToDevice, st));
  if (o.dropping(SITE_ENC + SITE_EMB)) {     // conditioning rows + the dropout on the embeddings, one launch
    hipLaunchKernelGGL(enc_input_kernel, dim3(grid_1d((int64_t)Me * d / 4)), dim3(256), 0, st, P, t->cond_off_dev, t->cond_rows_dev, t->n_cond, cond_idx,
                       t->xe[0], B, S, d, o.key(SITE_ENC + SITE_EMB), t->drop_thresh, t->drop_scale);
  } else if (t->n_cond > 0) {
    hipLaunchKernelGGL(cond_gather_kernel, dim3(B * t->n_cond), dim3(128), 0, st, P, t->cond_off_dev, t->cond_rows_dev, t->n_cond, cond_idx,
                       t->xe[0], S, d);
  }
  for (int l = 0; l < Le; ++l) {
    const EncOff& e = t->enc[l];
    RC(attn_self_fwd<T>(o, t->xe[2 * l], t->xe[2 * l + 1], e.ln0, e.qkv, e.o, t->h0e[l], t->qkve[l], t->Pe[l], t->aoe[l], B, S, t->etab, 0,
                        SITE_ENC + 16 * l, t->kte[l], t->lse_e[l]));
    RC(ff_fwd<T>(o, t->xe[2 * l + 1], t->xe[2 * l + 2], e.ln1, e.wi, e.wo, t->h1e[l], t->abe[l], t->mide[l], Me, SITE_ENC + 16 * l));
  }
  RC(o.norm_drop(t->xe[2 * Le], t->o_eln, t->hE, Me, SITE_ENC + SITE_FIN));
  // decoder
  if (o.dropping(SITE_DEC + SITE_EMB)) {
    hipLaunchKernelGGL(embed_rows_drop_kernel, dim3(ceil_div(Md, 4)), dim3(256), 0, st, t->dec_in, P + t->o_shared, t->xd[0], Md, d, V, g.pad_token_id,
                       o.key(SITE_DEC + SITE_EMB), t->drop_thresh, t->drop_scale);
    M2M_CHECK_HIP(hipGetLastError());
  } else {
    RC(launch_embed_rows(t->dec_in, P + t->o_shared, t->xd[0], Md, d, V, g.pad_token_id, st));
  }
  const int64_t sPc1 = (int64_t)H * L * lps, sPc2 = (int64_t)L * lps;
  const bool fuse_c = o.stripe_ok(S) && o.fuse_on();       // cross-attention: P . V and dQ = dS . K inside the stripe kernels
  const bool fuse_c_pv = fuse_c && (o.fuse_mode() & 1), fuse_c_dq = fuse_c && (o.fuse_mode() & 2);
  for (int l = 0; l < Ld; ++l) {
    const DecOff& e = t->dec[l];
    RC(attn_self_fwd<T>(o, t->xd[3 * l], t->xd[3 * l + 1], e.ln0, e.qkv, e.o, t->h0d[l], t->qkvd[l], t->Pd[l], t->aod[l], B, L, t->dtab, 1,
                        SITE_DEC + 16 * l, t->ktd[l], t->lse_d[l]));
    // cross-attention (hf: modeling_t5.py:319-342: K/V from the encoder output, zero bias, no mask)
    RC(o.norm(t->xd[3 * l + 1], e.ln1, t->h1d[l], Md));
    RC(o.mm(TG_STORE_T, t->h1d[l], d, 0, o.W(e.cq), d, 0, t->cqd[l], inner, Md, inner, d));
    RC(o.mm(TG_STORE_T, t->hE, d, 0, o.W(e.ckv), d, 0, t->ckvd[l], 2 * inner, Me, 2 * inner, d));
    const T* cq = (const T*)t->cqd[l];
    const T* ckv = (const T*)t->ckvd[l];
    const T* Pu;
    const bool head_c = o.head_ok(L, S);
    if (head_c) {
      const HeadAttnArgs a = o.head_args(cq, inner, (int64_t)L * inner, ckv, 2 * inner, ckv + inner, 2 * inner, (int64_t)S * 2 * inner, t->aocd[l], t->lse_c[l], L, S,
                                         nullptr, 0, SITE_DEC + 16 * l + PL_PROBS_CROSS, t->Pcd[l]);
      RC(launch_attn_head_fwd(a, B, o.st));
    } else if (o.stripe_ok(S)) {
      if (fuse_c) RC(o.kv_transpose(ckv, 2 * inner, t->ktc[l], B, S));          // serves P . V here and dQ = dS . K in the backward pass
      RC(o.attn_probs(ckv, 2 * inner, (int64_t)S * 2 * inner, DK, cq, inner, (int64_t)L * inner, DK, t->Pcd[l], B, L, S, lps, nullptr, 0,
                 SITE_DEC + 16 * l + PL_PROBS_CROSS, &Pu, fuse_c_pv ? (const T*)t->ktc[l] + (int64_t)B * H * DK * align_up(S, 32) : nullptr, t->aocd[l], inner,
                 (int64_t)L * inner, DK));
    } else {
      RC(o.mmbh(TG_STORE_F32, cq, inner, 0, (int64_t)L * inner, DK, ckv, 2 * inner, 0, (int64_t)S * 2 * inner, DK, t->sc, lps, sPc1, sPc2, B, L, S, DK));
      RC(o.softmax(t->sc, t->Pcd[l], B, L, S, lps, nullptr, 0, SITE_DEC + 16 * l + PL_PROBS_CROSS, &Pu));
    }
    if (!head_c && !fuse_c_pv)
      RC(o.mmbh(TG_STORE_T, Pu, lps, 0, sPc1, sPc2, ckv + inner, 2 * inner, 1, (int64_t)S * 2 * inner, DK, t->aocd[l], inner,
                (int64_t)L * inner, DK, B, L, DK, S));
    RC(o.mm_resid(t->aocd[l], e.co, t->xd[3 * l + 2], Md, d, inner, t->xd[3 * l + 1], SITE_DEC + 16 * l + PL_CROSS_OUT));
    RC(ff_fwd<T>(o, t->xd[3 * l + 2], t->xd[3 * l + 3], e.ln2, e.wi, e.wo, t->h2d[l], t->abd[l], t->midd[l], Md, SITE_DEC + 16 * l));
  }
  RC(o.norm_drop(t->xd[3 * Ld], t->o_dln, t->hD, Md, SITE_DEC + SITE_FIN));
  RC(o.mm(TG_STORE_F32, t->hD, d, 0, o.W(t->o_lm), d, 0, t->logits, V, Md, V, d));
  if (logits_out) M2M_CHECK_HIP(hipMemcpyAsync(logits_out, t->logits, (size_t)Md * V * 4, hipMemcpyDeviceToDevice, st));
  // loss + gradient of the logits
  hipLaunchKernelGGL(ce_kernel<T>, dim3(ceil_div(Md, 4)), dim3(256), 0, st, t->logits, labels, t->inv_n, t->row_loss, (T*)t->dlog, Md, V, ldv);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, t->row_loss, Md, t->inv_n, loss_out);
  M2M_CHECK_HIP(hipGetLastError());
  if (!G) return M2M_OK;

  // ================= backward =================
  RC(o.begin_sub(0));
  RC(o.dW(t->dlog, ldv, V, t->hD, d, d, G + t->o_lm, Md));                                        // dW_lm = dlogits^T . hD
  RC(o.end_sub());
  RC(o.dX(TG_STORE_F32, t->dlog, ldv, t->o_lm, V, d, t->dh, d, Md));                              // dhD = dlogits . W_lm
  float* dcur = t->dxa;
  float* dnext = t->dxb;
  o.after_site = SITE_DEC + 16 * (Ld - 1) + PL_FF_OUT;
  RC(o.norm_bwd(t->xd[3 * Ld], t->o_dln, t->dh, nullptr, dcur, G, Md, SITE_DEC + SITE_FIN));
  for (int l = Ld - 1; l >= 0; --l) {
    const DecOff& e = t->dec[l];
    o.after_site = SITE_DEC + 16 * l + PL_CROSS_OUT;
    RC(ff_bwd<T>(o, t->xd[3 * l + 2], dcur, dnext, G, e.ln2, e.wi, e.wo, t->h2d[l], t->abd[l], t->midd[l], Md, SITE_DEC + 16 * l));
    std::swap(dcur, dnext);
    // ---- cross-attention backward: dcur = d x[3l+2] ----
    RC(o.begin_sub(1u << m2m_trainer::K_DXT | 1u << m2m_trainer::K_DCQ | 1u << m2m_trainer::K_DCKV));
    RC(o.cvt_branch(dcur, t->dxT, (int64_t)Md * d, SITE_DEC + 16 * l + PL_CROSS_OUT));
    RC(o.dW(t->dxT, d, d, t->aocd[l], inner, inner, G + e.co, Md));
    RC(o.dX(TG_STORE_T, t->dxT, d, e.co, d, inner, t->dO, inner, Md));
    const T* cq = (const T*)t->cqd[l];
    const T* ckv = (const T*)t->ckvd[l];
    const T* dO = (const T*)t->dO;
    T* dckv = (T*)t->dckv;
    T* dcq = (T*)t->dcq;
    const int64_t sK1 = (int64_t)S * 2 * inner, sQ1 = (int64_t)L * inner;
    const T* Pu;
    const bool head_cb = o.head_ok(L, S);
    if (head_cb) {
      HeadAttnArgs a = o.head_args(cq, inner, sQ1, ckv, 2 * inner, ckv + inner, 2 * inner, sK1, t->aocd[l], t->lse_c[l], L, S, nullptr, 0,
                                   SITE_DEC + 16 * l + PL_PROBS_CROSS, t->Pcd[l]);
      a.dO = (const bf16_t*)dO;
      a.dQ = (bf16_t*)dcq; a.lddq = inner; a.sdQb = sQ1;
      a.dK = (bf16_t*)dckv; a.dV = (bf16_t*)(dckv + inner); a.lddk = a.lddv = 2 * inner; a.sdKb = a.sdVb = sK1;
      RC(launch_attn_head_bwd(a, B, o.st));
    }
    if (!head_cb) {
      const bool pair_c = o.stripe_ok(S) && o.pair_on();
      const bool pd_fused_c = pair_c && o.dropping(SITE_DEC + 16 * l + PL_PROBS_CROSS) && o.pd_fuse_on();
      if (pd_fused_c) Pu = (const T*)t->sc;
      else RC(o.redrop(t->Pcd[l], (int64_t)B * H * L * lps, SITE_DEC + 16 * l + PL_PROBS_CROSS, &Pu, pair_c));
      if (!pair_c) RC(o.mmbh(TG_STORE_T, Pu, lps, 1, sPc1, sPc2, dO, inner, 1, sQ1, DK, dckv + inner, 2 * inner, sK1, DK, B, S, DK, L));      // dV = Pd^T dO
      if (o.stripe_ok(S)) {
        RC(o.dscores(ckv + inner, 2 * inner, sK1, DK, dO, inner, sQ1, DK, t->Pcd[l], t->dS, B, L, S, lps, SITE_DEC + 16 * l + PL_PROBS_CROSS, false,
                     fuse_c_dq ? (const T*)t->ktc[l] : nullptr, dcq, inner, sQ1, DK, 0, pd_fused_c ? t->sc : nullptr));
      } else {
        RC(o.mmbh(TG_STORE_F32, dO, inner, 0, sQ1, DK, ckv + inner, 2 * inner, 0, sK1, DK, t->sc, lps, sPc1, sPc2, B, L, S, DK));             // dPd = dO V^T
        RC(o.softmax_bwd(t->Pcd[l], t->sc, t->dS, B * H * L, S, lps, SITE_DEC + 16 * l + PL_PROBS_CROSS));
      }
      const T* dS = (const T*)t->dS;
      if (!fuse_c_dq) RC(o.mmbh(TG_STORE_T, dS, lps, 0, sPc1, sPc2, ckv, 2 * inner, 1, sK1, DK, dcq, inner, sQ1, DK, B, L, DK, S));               // dQ = dS K
      if (pair_c)                                                                                                                              // dV | dK
        RC(o.mmbh2(TG_STORE_T, Pu, dS, lps, 1, sPc1, sPc2, dO, inner, sQ1, DK, cq, inner, sQ1, DK, 1, dckv + inner, dckv, 2 * inner, sK1, DK, B, S, DK, L));
      else
        RC(o.mmbh(TG_STORE_T, dS, lps, 1, sPc1, sPc2, cq, inner, 1, sQ1, DK, dckv, 2 * inner, sK1, DK, B, S, DK, L));                          // dK = dS^T Q
    }
    RC(o.dW(dcq, inner, inner, t->h1d[l], d, d, G + e.cq, Md));
    RC(o.dX(TG_STORE_F32, dcq, inner, e.cq, inner, d, t->dh, d, Md));
    o.after_site = SITE_DEC + 16 * l + PL_SELF_OUT;
    RC(o.norm_bwd(t->xd[3 * l + 1], e.ln1, t->dh, dcur, dnext, G, Md));
    std::swap(dcur, dnext);
    RC(o.dW(dckv, 2 * inner, 2 * inner, t->hE, d, d, G + e.ckv, Me));                               // dWckv = dckv^T . hE
    RC(o.dX(l == Ld - 1 ? TG_STORE_F32 : TG_ACC_F32, dckv, 2 * inner, e.ckv, 2 * inner, d, t->dhE, d, Me));                  // dhE (+)= dckv . Wckv
    RC(o.end_sub());
    // ---- causal self-attention backward ----
    o.after_site = l > 0 ? SITE_DEC + 16 * (l - 1) + PL_FF_OUT : -2;
    RC(attn_self_bwd<T>(o, t->xd[3 * l], dcur, dnext, G, e.ln0, e.qkv, e.o, t->h0d[l], t->qkvd[l], t->Pd[l], t->aod[l], B, L, t->dbucket, t->o_drb,
                        l == Ld - 1 ? 0 : 1, SITE_DEC + 16 * l, t->ktd[l], t->lse_d[l], t->dtab));
    std::swap(dcur, dnext);
  }
  // token embedding (decoder inputs; the encoder is fed inputs_embeds) — hf: modeling_t5.py embed_tokens = shared
  {
    const bool dr = o.dropping(SITE_DEC + SITE_EMB);
    M2M_OPT_IN_LDS(embed_bwd_kernel<EMB_NW_SHARED>, 158 * 1024);
    hipLaunchKernelGGL(embed_bwd_kernel<EMB_NW_SHARED>, dim3(V), dim3(64 * EMB_NW_SHARED), embed_bwd_smem(Md), st, t->dec_in, Md, 1, 0, dcur, (int64_t)1, (int64_t)0, G + t->o_shared, d,
                       g.pad_token_id, V, dr ? o.key(SITE_DEC + SITE_EMB) : DropKey{nullptr, 0}, dr ? t->drop_thresh : 0u, t->drop_scale);
  }
  // Split pass (data-parallel overlap): everything the decoder side deferred is issued now, so the gradients of the shared embedding,
  // lm_head and every decoder block are FINAL here — the caller's hook releases whoever waits for them — and the encoder side
  // flushes again at the end (into the other half of the tables).  Same products, same reductions: bit-identical gradients.
  if (at_split && o.group) {
    RC(o.flush_group());
    RC(o.flush_norms());
    RC(o.flush_bias());
    RC((*at_split)());
    o.phase = 1;
  }
  // (Single-GPU passes flush ONCE, at the end.  Round 3 measured the alternative — the decoder half of the grouped launch and the
  //  decoder's small reductions issued here on the side stream, beside the encoder's backward chain: 3.73 -> 3.86 ms per pass at
  //  16 clips x S = 261, unchanged at 64 clips and at S = 190.  The chain's launches are latency-bound, one or two workgroups per CU,
  //  and every one of them ends when its slowest workgroup does; sharing CUs with a matrix-core-bound launch stretches all of them by
  //  more than the 0.18 ms it hides.  Stream priorities (main highest, side lowest) did not change that and cost 0.18 ms on their own.)
  // encoder
  o.after_site = SITE_ENC + 16 * (Le - 1) + PL_FF_OUT;
  RC(o.norm_bwd(t->xe[2 * Le], t->o_eln, t->dhE, nullptr, dcur, G, Me, SITE_ENC + SITE_FIN));
  for (int l = Le - 1; l >= 0; --l) {
    const EncOff& e = t->enc[l];
    o.after_site = SITE_ENC + 16 * l + PL_SELF_OUT;
    RC(ff_bwd<T>(o, t->xe[2 * l + 1], dcur, dnext, G, e.ln1, e.wi, e.wo, t->h1e[l], t->abe[l], t->mide[l], Me, SITE_ENC + 16 * l));
    std::swap(dcur, dnext);
    o.after_site = l > 0 ? SITE_ENC + 16 * (l - 1) + PL_FF_OUT : -2;
    RC(attn_self_bwd<T>(o, t->xe[2 * l], dcur, dnext, G, e.ln0, e.qkv, e.o, t->h0e[l], t->qkve[l], t->Pe[l], t->aoe[l], B, S, t->ebucket, t->o_erb,
                        l == Le - 1 ? 0 : 1, SITE_ENC + 16 * l, t->kte[l], t->lse_e[l], t->etab));
    std::swap(dcur, dnext);
  }
  // The tail of the pass is a set of reductions that do not depend on one another: the grouped weight-gradient launch (0.35 ms,
  // seven rounds of workgroups) and a handful of small ones — conditioning-embedding rows, the norm-weight column sums, the
  // relative-position-bias reductions (~90 us as a chain).  The small ones go to the side stream and run beside the big launch.
  static const bool tail_side = [] { const char* v = getenv("M2M_TRAIN_TAIL_SIDE"); return !(v && v[0] == '0'); }();
  const bool fork = o.group && st_side && tail_side;
  hipStream_t small = fork ? st_side : st;
  if (fork) {
    M2M_CHECK_HIP(hipEventRecord(t->ev_ready, st));
    M2M_CHECK_HIP(hipStreamWaitEvent(st_side, t->ev_ready, 0));
  }
  // conditioning embeddings: rows 0 .. n_cond-1 of every clip's encoder input (ref: music2midi/input.py:57-59)
  for (int i = 0; i < t->n_cond; ++i) {
    const bool dr = o.dropping(SITE_ENC + SITE_EMB);
    hipLaunchKernelGGL(embed_bwd_kernel<4>, dim3(t->cond_rows[i]), dim3(256), embed_bwd_smem(B, 4), small, cond_idx, B, t->n_cond, i, dcur, (int64_t)S, (int64_t)i,
                       G + t->o_cond[i], d, 0, t->cond_rows[i], dr ? o.key(SITE_ENC + SITE_EMB) : DropKey{nullptr, 0}, dr ? t->drop_thresh : 0u,
                       t->drop_scale);
  }
  M2M_CHECK_HIP(hipGetLastError());
  RC(o.join_side());
  o.st = small;
  rc = o.flush_norms();
  if (rc == M2M_OK) rc = o.flush_bias();
  o.st = st;
  if (rc != M2M_OK) return rc;
  if (fork) M2M_CHECK_HIP(hipEventRecord(t->ev_free[0], st_side));
  RC(o.flush_group());
  if (fork) M2M_CHECK_HIP(hipStreamWaitEvent(st, t->ev_free[0], 0));
  return M2M_OK;
}

}  // namespace

// ------------------------------------------------------------------ C ABI ---
namespace { void drop_graph(m2m_trainer* t); }

extern "C" int m2m_trainer_create(const m2m_t5_geometry* geom, int n_cond, const int* cond_rows, int precision, int max_batch,
                                  int max_enc_len, int max_dec_len, m2m_trainer** out) {
  M2M_REQUIRE(geom && out && (n_cond == 0 || cond_rows), "m2m_trainer_create: null argument");
  M2M_REQUIRE(precision == M2M_PREC_FP32 || precision == M2M_PREC_BF16 || precision == M2M_PREC_FP8, "m2m_trainer_create: bad precision %d", precision);
  const bool fp8 = precision == M2M_PREC_FP8;
  if (fp8) {
    precision = M2M_PREC_BF16;       // storage type and every non-projection product are the bf16 mode's
    M2M_REQUIRE(geom->d_model % 128 == 0 && geom->d_ff % 128 == 0 && (geom->num_heads * geom->d_kv) % 128 == 0,
                "m2m_trainer_create: fp8 mode needs d_model, d_ff and num_heads*d_kv to be multiples of 128 (MX blocks of 32 in 128-byte rows)");
  }
  M2M_REQUIRE(geom->d_kv == DK, "m2m_trainer_create: d_kv=%d unsupported (64 only)", geom->d_kv);
  M2M_REQUIRE(geom->d_model % 64 == 0 && geom->d_model <= 512 && geom->d_ff % 8 == 0, "m2m_trainer_create: d_model must be a multiple of 64 (<= 512), d_ff of 8");
  M2M_REQUIRE(n_cond >= 0 && n_cond <= 8 && max_batch >= 1 && max_enc_len > n_cond && max_dec_len >= 1, "m2m_trainer_create: bad sizes");
  // embed_bwd_kernel keeps the pass's whole id list in LDS (embed_bwd_smem: 4 bytes per label position + 16 KiB against the 158 KiB opt-in)
  M2M_REQUIRE(embed_bwd_smem(max_batch * max_dec_len) <= (size_t)158 * 1024,
              "m2m_trainer_create: max_batch * max_dec_len = %d label positions per pass exceed the %d the shared-embedding gradient kernel "
              "lists in LDS; use a smaller dataloader batch or shorter label sequences (the reference trains 16 x <= ~360)",
              max_batch * max_dec_len, (int)((158 * 1024 - EMB_NW_SHARED * 256 * 4) / 4));
  m2m_trainer* t = new m2m_trainer();
  t->g = *geom; t->precision = precision; t->inner = geom->num_heads * geom->d_kv; t->n_cond = n_cond;
  t->es = precision == M2M_PREC_BF16 ? 2 : 4;
  t->fp8 = fp8;
  // Gradient operands: e4m3 by default — with a scale per 32 elements the range of e5m2 is not needed, and its third
  // mantissa bit halves the noise every backward product adds (measured on the full model, per-tensor gradient cosine
  // against the fp32 oracle: e5m2 median 0.931 / min 0.908; e4m3: see tests/test_train_gpu.py).  M2M_FP8_GRAD=e5m2 selects e5m2.
  t->grad_fmt = (getenv("M2M_FP8_GRAD") && strcmp(getenv("M2M_FP8_GRAD"), "e5m2") == 0) ? 1 : 0;
  if (const char* parts = getenv("M2M_FP8_PARTS")) {
    t->fp8_fwd = strstr(parts, "fwd") != nullptr; t->fp8_dx = strstr(parts, "dx") != nullptr; t->fp8_dw = strstr(parts, "dw") != nullptr;
  }
  t->max_batch = max_batch; t->max_enc = max_enc_len; t->max_dec = max_dec_len;
  t->cond_rows.assign(cond_rows, cond_rows + n_cond);
  build_layout(t);
  int rc = build_arena(t);
  if (rc == M2M_OK) rc = build_optimizer(t);
  if (rc != M2M_OK) { m2m_trainer_destroy(t); return rc; }
  // streams / events of the step (M2M_TRAIN_SIDE=0: everything on the caller's stream; M2M_TRAIN_GRAPH=0: no graph replay)
  { const char* v = getenv("M2M_TRAIN_SIDE"); t->use_side = !(v && v[0] == '0'); }
  { const char* v = getenv("M2M_TRAIN_GRAPH"); t->use_graph = !(v && v[0] == '0'); }
  { const char* v = getenv("M2M_TRAIN_DW_GROUP"); t->use_group = !(v && v[0] == '0'); }
  if (t->use_side) {
    hipError_t e = hipStreamCreateWithFlags(&t->s_main, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&t->s_side, hipStreamNonBlocking);
    hipEvent_t* evs[] = {&t->ev_in, &t->ev_out, &t->ev_ready, &t->ev_free[0], &t->ev_free[1], &t->ev_mid};
    for (hipEvent_t* ev : evs)
      if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e != hipSuccess) { set_error("m2m_trainer_create: stream / event creation failed: %s", hipGetErrorString(e)); m2m_trainer_destroy(t); return M2M_ERR_HIP; }
  }
  *out = t;
  return M2M_OK;
}

extern "C" void m2m_trainer_destroy(m2m_trainer* t) {
  if (!t) return;
  (void)hipDeviceSynchronize();
  drop_graph(t);
  hipEvent_t evs[] = {t->ev_in, t->ev_out, t->ev_ready, t->ev_free[0], t->ev_free[1], t->ev_mid};
  for (hipEvent_t ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  if (t->s_main) (void)hipStreamDestroy(t->s_main);
  if (t->s_side) (void)hipStreamDestroy(t->s_side);
  if (t->arena) (void)hipFree(t->arena);
  if (t->af_mem) (void)hipFree(t->af_mem);
  delete t;
}

extern "C" int64_t m2m_trainer_num_params(const m2m_trainer* t) { return t ? t->n_floats : (int64_t)M2M_ERR_INVALID; }
extern "C" int m2m_trainer_num_tensors(const m2m_trainer* t) { return t ? (int)t->tensors.size() : M2M_ERR_INVALID; }
extern "C" int64_t m2m_trainer_workspace_bytes(const m2m_trainer* t) { return t ? t->arena_bytes : (int64_t)M2M_ERR_INVALID; }

extern "C" int m2m_trainer_tensor_info(const m2m_trainer* t, int index, m2m_tensor_info* out) {
  M2M_REQUIRE(t && out, "m2m_trainer_tensor_info: null argument");
  M2M_REQUIRE(index >= 0 && index < (int)t->tensors.size(), "m2m_trainer_tensor_info: index %d out of range", index);
  const TensorDesc& d = t->tensors[index];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, d.name.c_str(), sizeof(out->name) - 1);
  out->offset = d.off; out->rows = d.rows; out->cols = d.cols;
  return M2M_OK;
}

namespace {

// relative-position bucket tables of (S, L) on the device (host-built, uploaded synchronously: never inside a capture)
int ensure_tables(m2m_trainer* t, int S, int L, hipStream_t st) {
  const m2m_t5_geometry& g = t->g;
  if (t->tab_S != S) {
    const std::vector<int> eb = bucket_table(g, S, S, true);
    M2M_CHECK_HIP(hipMemcpyAsync(t->ebucket, eb.data(), eb.size() * 4, hipMemcpyHostToDevice, st));
    M2M_CHECK_HIP(hipStreamSynchronize(st));   // eb is a stack object
    t->tab_S = S;
  }
  if (t->tab_L != L) {
    const std::vector<int> db = bucket_table(g, L, L, false);
    M2M_CHECK_HIP(hipMemcpyAsync(t->dbucket, db.data(), db.size() * 4, hipMemcpyHostToDevice, st));
    M2M_CHECK_HIP(hipStreamSynchronize(st));
    t->tab_L = L;
  }
  return M2M_OK;
}

int run_pass(m2m_trainer* t, const float* P, const float* x, const int64_t* cond, const int64_t* labels, int B, int S, int L, float* loss, float* G,
             float* logits, hipStream_t st, hipStream_t side, const std::function<int()>* at_split = nullptr) {
  return t->precision == M2M_PREC_BF16 ? forward_backward_t<bf16_t>(t, P, x, cond, labels, B, S, L, loss, G, logits, st, side, at_split)
                                       : forward_backward_t<float>(t, P, x, cond, labels, B, S, L, loss, G, logits, st, side, at_split);
}

void drop_slot_graphs(m2m_trainer::GraphSlot& s) {
  if (s.gexec) { (void)hipGraphExecDestroy(s.gexec); s.gexec = nullptr; }
  if (s.gexec2) { (void)hipGraphExecDestroy(s.gexec2); s.gexec2 = nullptr; }
}
void drop_graph(m2m_trainer* t) {
  for (auto& s : t->slots) { drop_slot_graphs(s); s.valid = false; s.calls = 0; }
}
// the slot of `key`: its own if the key is cached, else the least recently used one, recycled (tables stay: the next direct pass
// compares against their host image and uploads what differs — behind a stream sync, so no kept graph of THIS slot is in flight)
int slot_for(m2m_trainer* t, const m2m_trainer::GraphKey& key) {
  int lru = 0;
  for (int i = 0; i < m2m_trainer::N_SLOTS; ++i) {
    if (t->slots[i].valid && t->slots[i].key == key) return i;
    if (!t->slots[i].valid) { if (t->slots[lru].valid) lru = i; }
    else if (t->slots[lru].valid && t->slots[i].last_use < t->slots[lru].last_use) lru = i;
  }
  m2m_trainer::GraphSlot& s = t->slots[lru];
  drop_slot_graphs(s);
  s.key = key; s.valid = true; s.calls = 0;
  return lru;
}

}  // namespace

// One forward (+ backward) pass.  Three ways to issue the same ~600 kernels, chosen here:
//  * directly on the caller's stream (forward only, M2M_TRAIN_GRAPH=0 / M2M_TRAIN_SIDE=0, or the first call of a shape);
//  * on the trainer's own two streams: the weight-gradient products of a sub-layer run beside the next sub-layer's
//    gradient chain (they fill the CUs the small dX products leave idle);
//  * as ONE captured HIP graph of those two streams, replayed from the second call with the same buffers and shapes on:
//    inputs are staged into trainer-owned buffers first, so the graph never holds a caller pointer other than the flat
//    parameter / gradient buffers (part of its key), and the dropout key advances on the device.
extern "C" int m2m_train_forward_backward(m2m_trainer* t, const float* params_dev, const float* enc_inputs_dev, const int64_t* cond_idx_dev,
                                          const int64_t* labels_dev, int B, int S, int Ld, float* loss_out_dev, float* grads_dev,
                                          float* logits_out_dev, void* stream) {
  M2M_REQUIRE(t && params_dev && enc_inputs_dev && labels_dev && loss_out_dev, "m2m_train_forward_backward: null argument");
  M2M_REQUIRE(t->n_cond == 0 || cond_idx_dev, "m2m_train_forward_backward: cond_idx_dev is null");
  M2M_REQUIRE(B >= 1 && B <= t->max_batch && S > t->n_cond && S <= t->max_enc && Ld >= 1 && Ld <= t->max_dec,
              "m2m_train_forward_backward: (B=%d, S=%d, Ld=%d) outside the trainer's (%d, %d, %d)", B, S, Ld, t->max_batch, t->max_enc, t->max_dec);
  hipStream_t caller = (hipStream_t)stream;
  int rc = ensure_tables(t, S, Ld, caller);
  if (rc != M2M_OK) return rc;
  const bool two = grads_dev && t->use_side && t->s_main && t->s_side;
  // split pass: the stream that carries the decoder-side gradient all-reduce waits for ev_mid, recorded where those gradients are final
  const bool split = grads_dev && t->sync_stream && t->ev_mid && t->use_group && !(t->fp8 && t->fp8_dw);
  hipStream_t work = two ? t->s_main : caller;
  const std::function<int()> release = [t, work]() -> int {
    M2M_CHECK_HIP(hipEventRecord(t->ev_mid, work));
    M2M_CHECK_HIP(hipStreamWaitEvent(t->sync_stream, t->ev_mid, 0));
    return M2M_OK;
  };
  // A sync stream is set but this pass is NOT split (per-product weight gradients: M2M_TRAIN_DW_GROUP=0, fp8 weight gradients):
  // whoever enqueues the early all-reduce on the sync stream must still find FINISHED gradients, so the stream is released at
  // the END of the pass instead of half-way (no overlap, no race)
  const bool release_at_end = grads_dev && t->sync_stream && t->ev_mid && !split;
  if (!two) {
    t->cur_slot = m2m_trainer::N_SLOTS;
    rc = run_pass(t, params_dev, enc_inputs_dev, cond_idx_dev, labels_dev, B, S, Ld, loss_out_dev, grads_dev, logits_out_dev, caller, nullptr,
                  split ? &release : nullptr);
    if (rc == M2M_OK && release_at_end) rc = release();
    return rc;
  }

  // ---- stage the inputs (stream-ordered behind whatever produced them), then hand over to the trainer's streams ----
  const m2m_t5_geometry& g = t->g;
  // (one launch instead of three copy dispatches of ~5 us each on the caller's stream: they sit between two steps)
  {
    const int64_t n4 = (int64_t)B * S * g.d_model / 4, nl = (int64_t)B * Ld, nc = (int64_t)B * t->n_cond;
    hipLaunchKernelGGL(stage_inputs_kernel, dim3(grid_1d(n4 + nl + nc)), dim3(256), 0, caller, reinterpret_cast<const float4*>(enc_inputs_dev),
                       reinterpret_cast<float4*>(t->xe[0]), n4, labels_dev, reinterpret_cast<int64_t*>(t->labels_buf), nl, cond_idx_dev,
                       reinterpret_cast<int64_t*>(t->cond_buf), nc);
    M2M_CHECK_HIP(hipGetLastError());
  }
  M2M_CHECK_HIP(hipEventRecord(t->ev_in, caller));
  M2M_CHECK_HIP(hipStreamWaitEvent(t->s_main, t->ev_in, 0));

  const m2m_trainer::GraphKey key{params_dev, grads_dev, B, S, Ld, t->drop_thresh, t->drop_seed, split};
  const int si = slot_for(t, key);
  m2m_trainer::GraphSlot& slot = t->slots[si];
  slot.calls += 1;
  slot.last_use = ++t->tick;
  t->cur_slot = si;
  if (t->use_graph && slot.calls >= 2 && !slot.gexec) {                    // second call with this key: capture
    // a split pass becomes TWO graphs: the capture is closed and reopened where the decoder-side gradients are final, and the
    // replay records ev_mid between the two launches
    hipGraphExec_t first = nullptr;
    int n_nodes = 0;
    const auto close_capture = [t, &n_nodes](hipGraphExec_t* out) -> int {
      hipGraph_t graph = nullptr;
      const hipError_t ce = hipStreamEndCapture(t->s_main, &graph);
      if (ce != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        set_error("m2m_train_forward_backward: graph capture failed: %s", hipGetErrorString(ce));
        return M2M_ERR_HIP;
      }
      size_t nn = 0;
      if (hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess) n_nodes += (int)nn;
      const hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ie != hipSuccess) { *out = nullptr; set_error("m2m_train_forward_backward: hipGraphInstantiate: %s", hipGetErrorString(ie)); return M2M_ERR_HIP; }
      return M2M_OK;
    };
    const std::function<int()> cut = [t, &first, &close_capture]() -> int {
      const int r = close_capture(&first);
      if (r != M2M_OK) return r;
      M2M_CHECK_HIP(hipStreamBeginCapture(t->s_main, hipStreamCaptureModeThreadLocal));
      return M2M_OK;
    };
    M2M_CHECK_HIP(hipStreamBeginCapture(t->s_main, hipStreamCaptureModeThreadLocal));
    rc = run_pass(t, params_dev, t->xe[0], t->cond_buf, t->labels_buf, B, S, Ld, t->loss_dev, grads_dev, nullptr, t->s_main, t->s_side,
                  split ? &cut : nullptr);
    hipGraphExec_t last = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(t->s_main, &cs);
    int rc2 = M2M_OK;
    if (cs != hipStreamCaptureStatusNone) rc2 = close_capture(&last);       // (a failed cut leaves no capture open)
    if (rc != M2M_OK || rc2 != M2M_OK || (split && !first)) {
      if (first) (void)hipGraphExecDestroy(first);
      if (last) (void)hipGraphExecDestroy(last);
      if (rc == M2M_OK && rc2 == M2M_OK) { set_error("m2m_train_forward_backward: the split pass never reached its split point"); rc = M2M_ERR_INVALID; }
      slot.valid = false;                                                    // the next call starts this key over
      return rc != M2M_OK ? rc : rc2;
    }
    if (split) { slot.gexec = first; slot.gexec2 = last; } else { slot.gexec = last; }
    slot.nodes = n_nodes;
  }
  // A replayed graph that is not split is launched straight on the CALLER's stream: the inputs were staged there, the loss copy and the
  // optimizer follow there, so neither hand-over event (ev_in / ev_out: a cross-stream dependency on each side of every step) is needed.
  // M2M_TRAIN_GRAPH_CALLER=0: on the trainer's own stream, as the directly issued and the split passes run.
  static const bool graph_on_caller = [] { const char* v = getenv("M2M_TRAIN_GRAPH_CALLER"); return !(v && v[0] == '0'); }();
  if (slot.gexec && !split && !release_at_end && graph_on_caller) {
    M2M_CHECK_HIP(hipGraphLaunch(slot.gexec, caller));
    M2M_CHECK_HIP(hipMemcpyAsync(loss_out_dev, t->loss_dev, 4, hipMemcpyDeviceToDevice, caller));
    if (logits_out_dev)
      M2M_CHECK_HIP(hipMemcpyAsync(logits_out_dev, t->logits, (size_t)B * Ld * g.vocab_size * 4, hipMemcpyDeviceToDevice, caller));
    return M2M_OK;
  }
  if (slot.gexec) {
    M2M_CHECK_HIP(hipGraphLaunch(slot.gexec, t->s_main));
    if (split) {
      rc = release();
      if (rc != M2M_OK) return rc;
      M2M_CHECK_HIP(hipGraphLaunch(slot.gexec2, t->s_main));
    }
  } else {
    rc = run_pass(t, params_dev, t->xe[0], t->cond_buf, t->labels_buf, B, S, Ld, t->loss_dev, grads_dev, nullptr, t->s_main, t->s_side,
                  split ? &release : nullptr);
    if (rc != M2M_OK) { slot.valid = false; return rc; }
  }
  if (release_at_end) { rc = release(); if (rc != M2M_OK) return rc; }
  M2M_CHECK_HIP(hipEventRecord(t->ev_out, t->s_main));
  M2M_CHECK_HIP(hipStreamWaitEvent(caller, t->ev_out, 0));
  M2M_CHECK_HIP(hipMemcpyAsync(loss_out_dev, t->loss_dev, 4, hipMemcpyDeviceToDevice, caller));
  if (logits_out_dev)
    M2M_CHECK_HIP(hipMemcpyAsync(logits_out_dev, t->logits, (size_t)B * Ld * g.vocab_size * 4, hipMemcpyDeviceToDevice, caller));
  return M2M_OK;
}

extern "C" int m2m_trainer_set_dropout(m2m_trainer* t, float p, uint64_t seed) {
  M2M_REQUIRE(t && p >= 0.f && p < 1.f, "m2m_trainer_set_dropout: p must be in [0, 1)");
  t->drop_p = p;
  t->drop_thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
  t->drop_scale = 1.0f / (1.0f - p);
  t->drop_seed = seed;
  M2M_CHECK_HIP(hipDeviceSynchronize());                               // nothing of an earlier pass still reads the counter
  M2M_CHECK_HIP(hipMemset(t->step_ctr_dev, 0, 8));                     // the mask sequence restarts
  return M2M_OK;
}

// Data-parallel overlap.  With a sync stream set, every forward+backward call issues the backward pass in two parts and makes
// `stream` wait (an event, no host sync) for the point where the gradients in the two "early" ranges are final; work the caller
// then enqueues on `stream` — the all-reduce of those ranges — runs beside the encoder-side backward.  nullptr switches it off.
extern "C" int m2m_trainer_set_sync_stream(m2m_trainer* t, void* stream) {
  M2M_REQUIRE(t, "m2m_trainer_set_sync_stream: null trainer");
  if (stream && !t->ev_mid) {                                            // (a trainer built without its own streams has no events yet)
    M2M_CHECK_HIP(hipEventCreateWithFlags(&t->ev_mid, hipEventDisableTiming));
  }
  t->sync_stream = (hipStream_t)stream;
  return M2M_OK;
}
// out[0..3] = {offset, count, offset, count} (floats of the flat gradient buffer): shared embedding + lm_head, and the decoder blocks
extern "C" int m2m_trainer_early_grad_ranges(const m2m_trainer* t, int64_t* out) {
  M2M_REQUIRE(t && out, "m2m_trainer_early_grad_ranges: null argument");
  out[0] = 0; out[1] = t->o_erb;
  out[2] = t->dec_begin; out[3] = t->dec_end - t->dec_begin;
  return M2M_OK;
}

extern "C" int m2m_trainer_graph_nodes(const m2m_trainer* t) {
  if (!t) return M2M_ERR_INVALID;
  const m2m_trainer::GraphSlot& s = t->slots[t->cur_slot];
  return (s.valid && s.gexec) ? s.nodes : 0;
}

extern "C" int m2m_adafactor_step(m2m_trainer* t, float* params_dev, const float* grads_dev, void* stream) {
  M2M_REQUIRE(t && params_dev && grads_dev, "m2m_adafactor_step: null argument");
  t->step += 1;
  return launch_adafactor(t->af, params_dev, grads_dev, t->step, (hipStream_t)stream);
}

extern "C" int m2m_adafactor_get_step(const m2m_trainer* t) { return t ? t->step : M2M_ERR_INVALID; }
extern "C" int64_t m2m_adafactor_state_floats(const m2m_trainer* t) { return t ? t->af.state_floats : (int64_t)M2M_ERR_INVALID; }

extern "C" int m2m_adafactor_state_export(const m2m_trainer* t, float* state_out_dev, void* stream) {
  M2M_REQUIRE(t && state_out_dev, "m2m_adafactor_state_export: null argument");
  M2M_CHECK_HIP(hipMemcpyAsync(state_out_dev, t->af.state, (size_t)t->af.state_floats * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return M2M_OK;
}
extern "C" int m2m_adafactor_state_import(m2m_trainer* t, const float* state_in_dev, int step, void* stream) {
  M2M_REQUIRE(t && state_in_dev && step >= 0, "m2m_adafactor_state_import: bad argument");
  M2M_CHECK_HIP(hipMemcpyAsync(t->af.state, state_in_dev, (size_t)t->af.state_floats * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  t->step = step;
  return M2M_OK;
}

#ifdef M2M_ST_STAMP
// diagnostic builds only (not declared in the public header): the stripe kernel's last phase stamps, [variant][phase]
extern "C" int m2m_debug_stripe_stamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(m2m::g_st_stamp), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
#endif
