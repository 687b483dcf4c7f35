// Row-complete products of the training step (bf16): a workgroup owns 32 WHOLE rows of a [M, d_model] output, so the row-wise
// operation that follows the product in the network runs in its epilogue instead of in a launch of its own.
//   forward   x_out = x_in + dropout(A . W^T)  and  h = T(rmsnorm(x_out) * w_next)        (ref: music2midi/model.py:32-38 -> hf:
//             modeling_t5.py T5LayerSelfAttention / CrossAttention / FF: hidden + dropout(branch), then the NEXT sub-layer's
//             T5LayerNorm, modeling_t5.py:59-72) — the residual product and the rmsnorm_kernel launch behind it, in one.
// Why it pays although only ceil(M / 32) = 131 workgroups run (16 clips): the two launches it replaces are latency chains of
// 9.6 + 5.2 us (K = 512) / 13.9 + 5.2 us (K = 1 152), and what bounds THIS form is the weight matrix streaming through every
// workgroup — measured with tools/shared_stream.hip: 131 workgroups pulling the same 416 KB / 936 KB from L2 take 4.1 / 7.7 us
// launch to launch (120-130 GB/s per CU, 16 TB/s in total), with 12 MFMAs per wave and 64-deep k-step riding along.
// Main loop: operand tiles global -> LDS by DMA (global_load_lds_dwordx4, swizzled source addresses, as gemm_kernel's ring), a ring
// of three 64-deep stages with two requested ahead, counted s_waitcnt vmcnt + one s_barrier per step.  Epilogue: the accumulators go to LDS once, are re-read row-major (8 threads per row, 16-byte
// pieces), so the dropout hash is one per four elements, x_out / h leave in whole 128 / 64-byte row pieces and the row statistic is
// a sum over 8 neighbouring lanes.
#include "mma.h"
#include "t5.h"
#include "train.h"

namespace m2m {

namespace {

template <int N> __device__ inline void rg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void rg_glds16(const bf16_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int RG_BM = 32;

// Ring of 64-deep stages in gemm_kernel's LDS layout (128-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7): a DMA wave
// instruction moves 8 whole 128-byte lines): three stages at N <= 384 (2 x 52 KB in flight per CU), two at N = 512.  (Measured on the
// way here, K = 512 / 1 152 at 16 clips, tools/rowgemm_bench.py: ONE 64-deep stage in flight 17 / 30 us — a memory round trip per
// step; five 32-deep stages of 64-byte rows 16.6 / 30.2 us — half lines, 65 GB/s per CU, and fragment reads issued one MFMA ahead.)
constexpr int rg_stages(int nt) { return nt == 4 ? 2 : 3; }

// NT: 32-column tiles per wave (N = 128 NT)
template <int NT>
__global__ __launch_bounds__(256) void rowgemm_norm_kernel(RowGemmArgs g) {
  constexpr int N = 128 * NT, RG_NS = rg_stages(NT);
  constexpr int ROWS = RG_BM + N;                     // tile rows of a stage: A [32][64], then W [N][64]
  constexpr int STAGE = ROWS * 64;                    // elements
  constexpr int PER = ROWS / 32;                      // DMA instructions per wave and stage (8 rows x 128 B each, four waves)
  constexpr int EP = N + 4;                           // pitch (floats) of the epilogue tile
  static_assert(ROWS % 32 == 0 && (RG_NS - 1) * PER <= 63, "every wave issues the same count; the vmcnt field counts 63 loads");
  extern __shared__ __align__(1024) unsigned char rg_smem[];
  bf16_t* AB = reinterpret_cast<bf16_t*>(rg_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * RG_BM;
  const int K = g.K, nk = K / 64;

  // the residual rows in the epilogue's layout: thread = (row tid >> 3, 16-byte pieces (tid & 7) + 8 j): requested before the k loop
  const int erow = tid >> 3, ec = tid & 7;
  const int grow = min(m0 + erow, g.M - 1);
  float4 xin[4 * NT];                                 // (N / 4 = 32 NT pieces per row, 8 threads: 4 NT pieces per thread)
  {
    const float* xr = g.resid + (int64_t)grow * N;
#pragma unroll
    for (int j = 0; j < 4 * NT; ++j) xin[j] = *reinterpret_cast<const float4*>(xr + 4 * (ec + 8 * j));
  }

  // DMA sources: wave instruction q = wave + 4 u of a stage covers tile rows 8 q .. 8 q + 7; lane -> (row 8 q + (lane >> 3), slot lane & 7)
  const int lrow = lane >> 3, slot = lane & 7;
  const bf16_t* src[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int row = (wave + 4 * u) * 8 + lrow;        // tile row
    const int chunk = slot ^ ((row >> 1) & 7);
    if (row < RG_BM) src[u] = g.A + (int64_t)min(m0 + row, g.M - 1) * K + chunk * 8;
    else src[u] = g.W + (int64_t)(row - RG_BM) * K + chunk * 8;
  }
  typedef __attribute__((address_space(3))) bf16_t* lds_ptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)AB;
  const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int kt) {
    const unsigned base = lds0 + 2u * (unsigned)((kt % RG_NS) * STAGE) + wave_u * 1024u;
#pragma unroll
    for (int u = 0; u < PER; ++u) rg_glds16(src[u] + (int64_t)kt * 64, base + (unsigned)u * 4096u);
  };
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = zero_acc();
  if (!(g.dbg & 2)) {
#pragma unroll
    for (int s0 = 0; s0 < RG_NS - 1; ++s0)
      if (s0 < nk) issue(s0);
  }
  const int sw = (r >> 1) & 7;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed when at most `ahead` younger stages of this wave are outstanding (vmcnt is in order; the residual rows are older)
    const int ahead = min(nk, kt + RG_NS - 1) - kt - 1;
    if (g.dbg & 2) {}
    else if (RG_NS > 2 && ahead >= 1) rg_wait_vmcnt<(RG_NS > 2 ? PER : 0)>();
    else rg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                     // ... everybody's pieces have; and every wave is past its reads of stage kt - 1
    asm volatile("" ::: "memory");
    if (kt + RG_NS - 1 < nk && !(g.dbg & 2)) issue(kt + RG_NS - 1);   // into the slot stage kt - 1 was read from
    if (g.dbg & 1) continue;
    const bf16_t* As = AB + (kt % RG_NS) * STAGE;
    const bf16_t* Ws = As + (RG_BM + wave * NT * 32) * 64;
    // all fragments of two 16-deep substeps are requested before their MFMAs: one wave per SIMD has nobody to hide an LDS round trip per product
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      Frag<bf16_t> fa[2], fb[2][NT];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int co = ((2 * (2 * sp + q) + h) ^ sw) << 3;
        fa[q] = load_frag(As + r * 64 + co);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[q][j] = load_frag(Ws + (j * 32 + r) * 64 + co);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma16(acc[j], fa[q], fb[q][j]);
    }
  }
  if (g.dbg & 4) return;
  __syncthreads();                                    // the ring is free: it becomes the fp32 tile [32][EP]
  float* Et = reinterpret_cast<float*>(rg_smem);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) Et[acc_row(e, lane) * EP + (wave * NT + j) * 32 + r] = acc[j][e];
  __syncthreads();
  const uint64_t dkey = g.thresh ? drop_site_key(g.dk) : 0ull;
  const bool live = m0 + erow < g.M;
  float4 x[4 * NT];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4 * NT; ++j) {
    const int col = 4 * (ec + 8 * j);
    float4 v = *reinterpret_cast<const float4*>(Et + erow * EP + col);
    const int64_t at = (int64_t)grow * N + col;
    if (g.thresh) {
      const uint32_t kb = drop_keep4(dkey, at, g.thresh);
      v = make_float4((kb & 1u) ? v.x * g.scale : 0.f, (kb & 2u) ? v.y * g.scale : 0.f, (kb & 4u) ? v.z * g.scale : 0.f, (kb & 8u) ? v.w * g.scale : 0.f);
    }
    const float4 xi = xin[j];
    const float4 o = make_float4(xi.x + v.x, xi.y + v.y, xi.z + v.z, xi.w + v.w);
    x[j] = o;
    if (live) *reinterpret_cast<float4*>(g.x_out + at) = o;
    ss += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
  }
  ss = group_sum<8>(ss);                              // the 8 threads of a row are 8 neighbouring lanes
  const float rstd = rsqrtf(ss / (float)N + g.eps);
  if (g.h_out && live) {
#pragma unroll
    for (int j = 0; j < 4 * NT; ++j) {
      const int col = 4 * (ec + 8 * j);
      const float4 w = *reinterpret_cast<const float4*>(g.norm_w + col);
      const uint2 pk = make_uint2(pack2_bf16(x[j].x * rstd * w.x, x[j].y * rstd * w.y), pack2_bf16(x[j].z * rstd * w.z, x[j].w * rstd * w.w));
      *reinterpret_cast<uint2*>(g.h_out + (int64_t)grow * N + col) = pk;
    }
  }
}

template <int NT>
int launch_nt(const RowGemmArgs& a, hipStream_t st) {
  constexpr int N = 128 * NT;
  const size_t ring = (size_t)rg_stages(NT) * (RG_BM + N) * 64 * sizeof(bf16_t), tile = (size_t)RG_BM * (N + 4) * sizeof(float);
  const size_t smem = ring > tile ? ring : tile;
  M2M_OPT_IN_LDS((rowgemm_norm_kernel<NT>), 158 * 1024);
  hipLaunchKernelGGL((rowgemm_norm_kernel<NT>), dim3((unsigned)ceil_div(a.M, RG_BM)), dim3(256), smem, st, a);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

}  // namespace

bool rowgemm_norm_ok(int M, int N, int K) { return M >= 1 && (N == 128 || N == 256 || N == 384 || N == 512) && K >= 64 && K % 64 == 0; }

int launch_rowgemm_norm(const RowGemmArgs& a, hipStream_t st) {
  M2M_REQUIRE(rowgemm_norm_ok(a.M, a.N, a.K) && a.A && a.W && a.resid && a.x_out, "rowgemm_norm: unsupported shape %d x %d x %d", a.M, a.N, a.K);
  switch (a.N / 128) {
    case 1: return launch_nt<1>(a, st);
    case 2: return launch_nt<2>(a, st);
    case 3: return launch_nt<3>(a, st);
    default: return launch_nt<4>(a, st);
  }
}

}  // namespace m2m

// ---------------------------------------------------------------------------------------------------- test utility (C ABI)
extern "C" int m2m_rowgemm_norm_bf16(const uint16_t* A, const uint16_t* W, const float* resid, const float* norm_w, int M, int N, int K, float eps,
                                     float drop_p, const uint64_t* step_key_dev, uint64_t site_salt, float* x_out, uint16_t* h_out, int dbg, void* stream) {
  using namespace m2m;
  M2M_REQUIRE(A && W && resid && x_out && (step_key_dev || drop_p <= 0.f), "m2m_rowgemm_norm_bf16: null argument");
  RowGemmArgs a{};
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.M = M; a.N = N; a.K = K; a.resid = resid; a.x_out = x_out; a.norm_w = norm_w; a.h_out = (bf16_t*)h_out;
  a.eps = eps; a.dk = DropKey{step_key_dev, site_salt}; a.thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
  a.scale = 1.0f / (1.0f - drop_p); a.dbg = dbg;
  return launch_rowgemm_norm(a, (hipStream_t)stream);
}
