// Decoder-side kernels: the KV-cached greedy step.  Replaces the per-token loop of HF
// generate() that ref: music2midi/transformer.py:44 drives (hf: generation/utils.py:2876-2937,
// one T5 decoder forward hf: models/t5/modeling_t5.py:448-509,1031-1047 per token, ~60 vendor
// kernel launches and a host sync each).  Here a step is a fixed sequence of kernels that read
// the step index from device memory, so the whole loop replays captured hipGraphs with no
// host round trip; finished-row bookkeeping (pad after EOS) lives in device memory.
//
// Clips are independent, so a batch is decoded as several independent CHAINS ("views": a
// contiguous range of clips with its own step counter, stream and graph).  One chain's kernels
// are latency-bound (12..72 workgroups each); running several chains side by side fills the
// 256 CUs and overlaps one chain's HBM-bound attention with another chain's projections.
//
//   dec_attn_kernel  one (clip, head) per 1024-thread workgroup: RMSNorm + that head's projection
//                    (self: q,k,v + cache append; cross: q) fused in front of the attention, which
//                    streams K then V straight from HBM to registers (16 B per lane per load, no
//                    LDS staging: each byte is used once), fp32 softmax, shuffle + LDS reduction.
//   dec_ff_kernel    the whole gated feed-forward sub-layer: RMSNorm + MFMA up projection of a 32-column
//                    slice + gated GELU (LDS) + MFMA down projection, added into the residual stream.
//   dec_gemm_kernel  lm_head: RMSNorm + skinny projection, one 16x16 MFMA tile per workgroup, waves split K.
//   dec_head_kernel  argmax / EOS+pad bookkeeping / next-token embedding / step counter.
#include "mma.h"
#include "t5.h"

#include <stdlib.h>

namespace m2m {

bool decode_headless();

// ---- optional in-kernel wall-clock stamps (diagnostic builds only: -DM2M_STAMPS) ----
// s_memrealtime runs at a constant 100 MHz and is the same clock on every CU, so stamps from
// different kernels can be laid on one timeline.  Block 0 / thread 0 of every decode kernel logs
// {kernel id, phase, ticks}.  Never compiled into the product library.
#ifdef M2M_STAMPS
__device__ unsigned long long g_stamps[1 << 18];
__device__ unsigned int g_stamp_n;
// phase 0 claims 8 consecutive slots with ONE returning atomic (its round trip is paid at kernel
// entry, before anything is timed); later phases are fire-and-forget stores into the claimed slots.
#define M2M_STAMP(kid, phase)                                                                   \
  do {                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                               \
      static_assert((phase) < 8, "phase");                                                      \
      if ((phase) == 0) m2m_stamp_base = atomicAdd(&g_stamp_n, 8u);                             \
      const unsigned long long tk = __builtin_amdgcn_s_memrealtime();                           \
      if (m2m_stamp_base + 8 <= (1u << 18))                                                     \
        g_stamps[m2m_stamp_base + (phase)] = ((unsigned long long)(kid) << 56) | ((unsigned long long)(phase) << 48) | (tk & 0xFFFFFFFFFFFFull); \
    }                                                                                           \
  } while (0)
#define M2M_STAMP_DECL unsigned int m2m_stamp_base = 0; (void)m2m_stamp_base;
#else
#define M2M_STAMP(kid, phase) do {} while (0)
#define M2M_STAMP_DECL
#endif

// ---- decoder residual stream: int64 fixed point (scale 2^30) ----
// The output projections of the attention sub-layers are accumulated per head straight from the
// attention kernels (8 workgroups per clip add into the same row).  Float atomics would make the
// sum depend on arrival order; integer adds are associative, so the result is bit-reproducible
// whatever the order, and exact (2^-30 resolution, +-8.6e9 range) where fp32 would round.
//
// The stream lives in THREE buffers that rotate (A -> B -> C -> A): an attention kernel READS the
// complete row from one buffer and ACCUMULATES residual + per-head projections into the next one,
// which the kernel before it left zeroed, and zeroes the third.  A workgroup that starts late
// (more workgroups than CUs) therefore never reads a row that a finished sibling has already
// added to, which a single read-modify buffer would allow.
typedef long long xq_t;
// float -> fixed point, round to nearest even: (double)v * 2^30 is exact, adding 1.5 * 2^52 leaves the rounded
// integer in the low mantissa bits (two's complement), so the conversion is cvt + mul + add + a 64-bit subtract
// instead of the ~14-instruction float -> int64 sequence.  Valid for |v| < 2^21.
__device__ inline xq_t xq_fix(float v) {
  const double m = (double)v * 1073741824.0 + 6755399441055744.0;
  return __double_as_longlong(m) - 0x4338000000000000ll;
}
// The same conversion with the range guarded: a value the magic-number add cannot represent (|v| >= 2^21, Inf,
// NaN - a corrupt checkpoint, a diverged fine-tune) would otherwise become an arbitrary integer and the decoder
// would go on emitting plausible-looking ids where the reference produces NaN logits.  Such a value is clamped
// (NaN -> 0) and a sticky flag is raised in the chain's DecState; m2m_generate_greedy / m2m_decode_forced return
// M2M_ERR_RANGE when they find it.  One compare per conversion; the store happens only on the failure path.
constexpr float XQ_LIMIT = 2097152.0f;   // 2^21
__device__ inline xq_t xq_fix_guarded(float v, DecState* st) {
  if (!(fabsf(v) < XQ_LIMIT)) {          // also true for NaN
    st->overflow = 1;
    v = (v != v) ? 0.f : copysignf(XQ_LIMIT - 1.0f, v);
  }
  return xq_fix(v);
}
// via double: int64 -> f64 is 4 instructions (two 32-bit converts + fma) against ~12 for the correctly
// rounded int64 -> f32 sequence, and the result is the same single rounding while |q| < 2^53 (|x| < 8.4e6)
__device__ inline float xq_flt(xq_t q) { return (float)((double)q * (1.0 / 1073741824.0)); }
__device__ inline float4 xq_load4(const xq_t* p) {   // p 16-byte aligned
  const longlong2 a = *reinterpret_cast<const longlong2*>(p);
  const longlong2 b = *reinterpret_cast<const longlong2*>(p + 2);
  return make_float4(xq_flt(a.x), xq_flt(a.y), xq_flt(b.x), xq_flt(b.y));
}

// ---- headless greedy loop: the arg-max travels as a packed 64-bit key ----
// The lm_head workgroups reduce their 16 columns per row and atomicMax a key into keys[row]; the NEXT step's layer-0
// self-attention workgroups decode it (token id -> embedding row = their input), the layer-0 cross-attention kernel
// does the per-row bookkeeping (token matrix, finished flag, unfinished count) and clears the key, the layer-0
// feed-forward kernel turns "no row unfinished" into done.  That removes dec_head_kernel (a kernel boundary + two
// dependent memory round trips per step) from the greedy chain; teacher forcing keeps the head kernel.
// key = (order-preserving image of the fp32 logit) << 32 | (0xFFFFFFFF - column): the largest key is the largest logit,
// ties go to the lowest column (torch.argmax); 0 = "no key".
__device__ inline unsigned long long amax_key(float v, int col) {
  unsigned int u = __float_as_uint(v + 0.0f);                      // -0 -> +0: they must tie
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)col);
}
__device__ inline int amax_key_token(unsigned long long key, int fin, int V, int pad_id) {
  int tok = (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
  if (fin || tok < 0 || tok >= V) tok = pad_id;                    // finished rows keep emitting pad (hf generation/utils.py:2929)
  return tok;
}
// every kernel of a step: the chain is over when done is set OR the step counter has reached the step budget
__device__ inline int chain_done(const DecState* st) { return st->done | (st->t >= st->max_steps); }

// ===================================================== skinny projection ====
// lm_head: logits[B, V] = RMSNorm(x) . W^T (untied, no d_model**-0.5 scaling: transformers 4.34 semantics).
struct DecGemmArgs {
  const xq_t* x;         // [rows, K] fixed-point residual stream
  int ldx;
  const float* ln_w;     // [K] RMSNorm weight
  float eps;
  const void* W;         // [Npad, K] T
  int K, N, B;
  DecState* state;
  float* out;            // [B, ldo] logits
  int ldo;
  unsigned long long* keys;   // headless greedy loop: [B] arg-max keys (the logits themselves are not stored)
};

// One 16-row x 16-column output tile per workgroup (MFMA 16x16x32 / 16x16x4), NW waves split K.
// A CU fetches only ~25 GB/s from beyond its L2 and every kernel starts cold for what the previous
// one wrote, so a skinny projection's time IS the bytes one workgroup pulls: small tiles spread W (and
// the rows of x) over many CUs.  Every global load (weights, activations, norm weights) is issued
// before the first use, so a launch is one memory round trip; RMSNorm statistics are reduced across
// waves through LDS while the weight loads are in flight; the cross-wave sum has a fixed order.
//
// NS (32-wide k-steps per wave) is a template parameter: a runtime bound would put every load of
// the unrolled batch behind its own branch + s_waitcnt (cdna_hip_programming.md, "three .s-level
// traps" item c).  blockDim.x = 64 * NW with NW = K / (32 * NS) <= 12.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int DG_MAXW = 12;

__device__ inline void mma32_16(f32x4_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                acc, 0, 0, 0);
}
// fp32: lane (r, g) holds k = 8g + j, j = 0..7; MFMA j sums the four lane groups -> k in {j, 8+j, 16+j, 24+j}
__device__ inline void mma32_16(f32x4_t& acc, const Frag<float>& a, const Frag<float>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
}

template <typename T, int NS, bool HEADLESS>
__global__ __launch_bounds__(64 * DG_MAXW) void dec_gemm_kernel(DecGemmArgs a) {
  M2M_STAMP_DECL
  __shared__ float ss_s[DG_MAXW][16];
  __shared__ float red[DG_MAXW][16 * 17];
  M2M_STAMP(2, 0);
  // headless: "live" comes from done and the STABLE copy of t (this kernel advances t itself at its end)
  const int t_cur = HEADLESS ? a.state->t_copy : 0;
  const int done = HEADLESS ? (a.state->done | (t_cur >= a.state->max_steps)) : chain_done(a.state);   // consumed only before the stores
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * 16;
  const int K = a.K;
  const T* W = reinterpret_cast<const T*>(a.W);
  const int kbeg = wave * (32 * NS) + 8 * g;
  const bool row_ok = (b0 + r) < a.B;
  const int arow = b0 + (row_ok ? r : 0);           // padding rows read row b0 and are never stored
  const T* wr = W + (int64_t)(n0 + r) * K + kbeg;
  // epilogue coordinates of this thread (threads 0..255 own one output each)
  const int orow = (tid >> 4) & 15, ocol = tid & 15;
  const int ob = b0 + orow, on = n0 + ocol;

  Frag<T> wf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) wf[s] = load_frag(wr + 32 * s);
  asm volatile("" ::"s"(done));   // materialise the flag now, under the vector loads (else it is sunk to the epilogue)

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  {
    const xq_t* xr = a.x + (int64_t)arow * a.ldx + kbeg;
    float4 x0[NS], x1[NS], g0[NS], g1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      x0[s] = xq_load4(xr + 32 * s);
      x1[s] = xq_load4(xr + 32 * s + 4);
      g0[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 32 * s);
      g1[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 32 * s + 4);
    }
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      ss += x0[s].x * x0[s].x + x0[s].y * x0[s].y + x0[s].z * x0[s].z + x0[s].w * x0[s].w +
            x1[s].x * x1[s].x + x1[s].y * x1[s].y + x1[s].z * x1[s].z + x1[s].w * x1[s].w;
    ss += lane_xor<16>(ss);
    ss += lane_xor<32>(ss);
    if (g == 0) ss_s[wave][r] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < nw; ++w) tot += ss_s[w][r];
    const float rs = row_ok ? rsqrtf(tot / (float)K + a.eps) : 0.f;   // rs = 0 zeroes the padding rows
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float xv[8] = {g0[s].x * (x0[s].x * rs), g0[s].y * (x0[s].y * rs), g0[s].z * (x0[s].z * rs),
                           g0[s].w * (x0[s].w * rs), g1[s].x * (x1[s].x * rs), g1[s].y * (x1[s].y * rs),
                           g1[s].z * (x1[s].z * rs), g1[s].w * (x1[s].w * rs)};
      const Frag<T> fa = pack_frag<T>(xv);
      mma32_16(acc, fa, wf[s]);
    }
  }
  // ---- cross-wave reduction (fixed order: deterministic).  acc[i]: row 4g + i, column r ----
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][(4 * g + i) * 17 + r] = acc[i];
  M2M_STAMP(2, 1);
  __syncthreads();
  if constexpr (HEADLESS) {
    if (done || tid >= 256) return;                     // (wave-uniform: waves 0-3 hold the 256 outputs)
    const int idx = orow * 17 + ocol;
    float v = red[0][idx];
    for (int w = 1; w < nw; ++w) v += red[w][idx];
    const bool ok = ob < a.B && on < a.N;
    if (ok && !(fabsf(v) <= 3.0e38f)) a.state->overflow = 1;        // non-finite logit: the reference would emit NaN, never a token
    unsigned long long key = ok ? amax_key(v, on) : 0ull;
    // max over the tile's 16 columns = the 16 lanes of this row's group (xor 8, 4, 2, 1 stay inside it)
    auto kmax = [&](unsigned long long o) { key = o > key ? o : key; };
    unsigned int lo = (unsigned int)key, hi = (unsigned int)(key >> 32);
#define M2M_KEY_STEP(M)                                                                                     \
    {                                                                                                       \
      const unsigned int olo = __builtin_bit_cast(unsigned int, lane_xor<M>(__builtin_bit_cast(float, lo)));  \
      const unsigned int ohi = __builtin_bit_cast(unsigned int, lane_xor<M>(__builtin_bit_cast(float, hi)));  \
      kmax(((unsigned long long)ohi << 32) | olo);                                                          \
      lo = (unsigned int)key; hi = (unsigned int)(key >> 32);                                               \
    }
    M2M_KEY_STEP(8) M2M_KEY_STEP(4) M2M_KEY_STEP(2) M2M_KEY_STEP(1)
#undef M2M_KEY_STEP
    if (ocol == 0 && ob < a.B) atomicMax(a.keys + ob, key);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) a.state->t = t_cur + 1;    // the step is over: advance the position
    M2M_STAMP(2, 2);
  } else {
    if (done || tid >= 256 || ob >= a.B || on >= a.N) return;
    const int idx = orow * 17 + ocol;
    float v = red[0][idx];
    for (int w = 1; w < nw; ++w) v += red[w][idx];
    a.out[(int64_t)ob * a.ldo + on] = v;
    M2M_STAMP(2, 2);
  }
}

// waves x steps decomposition of the reduction length: K = 32 * NS * NW
static bool dec_gemm_shape(int K, int* ns, int* nw) {
  switch (K) {
    case 128: *ns = 1; *nw = 4; return true;
    case 256: *ns = 2; *nw = 4; return true;
    case 384: *ns = 2; *nw = 6; return true;
    case 512: *ns = 2; *nw = 8; return true;
    default: return false;
  }
}

template <typename T>
static int launch_dec_gemm_t(const DecGemmArgs& a, hipStream_t st) {
  int ns = 0, nw = 0;
  if (!dec_gemm_shape(a.K, &ns, &nw)) {
    set_error("dec_gemm: K=%d not supported by the decode projections (128/256/384/512)", a.K);
    return M2M_ERR_INVALID;
  }
  dim3 grid((unsigned)ceil_div(a.N, 16), (unsigned)ceil_div(a.B, 16));
  dim3 block((unsigned)(64 * nw));
  if (a.keys) {
    if (ns == 1) hipLaunchKernelGGL((dec_gemm_kernel<T, 1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((dec_gemm_kernel<T, 2, true>), grid, block, 0, st, a);
  } else {
    if (ns == 1) hipLaunchKernelGGL((dec_gemm_kernel<T, 1, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((dec_gemm_kernel<T, 2, false>), grid, block, 0, st, a);
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

static int launch_dec_gemm(int precision, const DecGemmArgs& a, hipStream_t st) {
  return precision == M2M_PREC_BF16 ? launch_dec_gemm_t<bf16_t>(a, st) : launch_dec_gemm_t<float>(a, st);
}

// ===================================================== fused feed-forward ====
// One kernel per layer for  x += wo( gelu_new(wi_0 . n) * (wi_1 . n) ),  n = RMSNorm(x)
// (hf: modeling_t5.py T5DenseGatedActDense + T5LayerFF).  A workgroup owns 32 of the d_ff hidden
// columns for one 16-row block: it computes that slice of the gated activation (64 rows of the
// interleaved wi), keeps it in LDS, multiplies it with the matching 32 columns of wo and ADDS the
// [16 x d_model] partial result into the fixed-point residual stream (integer atomics: the sum does
// not depend on arrival order, see xq_t above).  The hidden activations never reach memory and the
// dependent kernel boundary between the up and the down projection is gone (26 -> 20 kernels/step).
// 32 columns per workgroup is the balance point measured with tools/atomic_rate.hip: d_ff/32 = 36
// adds per residual element cost ~2 us, 72 would cost ~5 us; wider slices put >100 KB of weights
// behind one CU's ~24..60 GB/s fetch path.
//
// Waves = KS = d_model / 64: wave ks multiplies k-slice ks of the normalised rows with all four
// n-tiles of the slice (a tile = 8 wi_0 rows + the matching 8 wi_1 rows, as repack.hip interleaves
// them), so the rows of x — 8 bytes per element, the largest operand — are fetched once per
// workgroup; in phase 2 wave w owns output columns [64 w, 64 w + 64).  One extra workgroup per row
// block adds the residual rows themselves and zeroes the third buffer of the rotation.  A CU pulls only ~25-30 GB/s
// from beyond its L2, so the kernel's time is the bytes one workgroup requests (x 49 KB + weights
// 74 KB at d_model 384); every global load is issued before the first use.
struct DecFfArgs {
  const xq_t* x;         // [B, d] residual stream after the attention sub-layers (complete)
  xq_t* x_out;           // [B, d] zero on entry: x + FF(x) is accumulated into it
  xq_t* x_zero;          // [B, d] third buffer of the rotation, left zeroed
  const float* ln_w;
  float eps;
  const void* Wi;        // [2 * d_ff, d] T, 16-row groups: 8 rows of wi_0 then the same 8 of wi_1
  const void* Wo;        // [d, d_ff] T
  int d, d_ff, B;
  DecState* state;
  int check_done;        // headless greedy loop, layer 0: the carry workgroup of row block 0 turns "no row unfinished" into done
};

// FF_R = residual rows per workgroup (<= 16, the MFMA tile height), a template parameter chosen per chain.  Small chains: 8 halves
// the fixed-point rows a workgroup pulls (they were just written, so they come from HBM, not from a cache: tools/l2_persist.hip) at
// the price of reading the weight slices twice from the Infinity Cache: 247.3 -> 244.6 ms per batch at 2 x 16 clips; 4 rows: 259.5.
// Large chains (round 6): 16 rows halve the workgroups and the weight-slice reads of a launch (decode_ff_rows).  A row's arithmetic
// does not depend on its tile (MFMA rows are independent): ids are bit-identical whichever is taken.
constexpr int FF_C = 32;        // hidden columns per workgroup
constexpr int FF_HP = FF_C + 8; // LDS row pitch of the activation slice (elements; keeps 16-byte alignment)

// diagnostic builds only (-DM2M_FF_ABL=mask, never the product): 1 = the slice workgroups add nothing (x passes through), 2 = they read no rows of x,
// 4 (multi-slice form) = the later slices re-use the first slice's weights
#ifndef M2M_FF_ABL
#define M2M_FF_ABL 0
#endif
template <typename T, int KS, int FF_R>
__global__ __launch_bounds__(64 * KS) __attribute__((amdgpu_waves_per_eu(1, 2))) void dec_ff_kernel(DecFfArgs a) {
  M2M_STAMP_DECL
  __shared__ float ss_s[KS][16];
  __shared__ float red[KS][4][16 * 17];
  __shared__ __align__(16) T hs[16 * FF_HP];
  M2M_STAMP(4, 0);
  const int done = chain_done(a.state);
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  // XCD-aware order (workgroup id % 8 picks the XCD, each has its own L2): the row blocks that share a weight
  // slice run on the same XCD, so the slice is fetched into that L2 once instead of once per row block
  // (PMC: 14.2 MB fetched per launch for 2.75 MB of unique operands with the row-major order)
  const int nrb = (a.B + FF_R - 1) / FF_R, nsl = a.d_ff / FF_C + 1;      // row blocks; weight slices + the carry
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int c = xcd + 8 * (slot / nrb), b0 = (slot - (slot / nrb) * nrb) * FF_R;
  const int K = a.d;
  if (c >= nsl) return;   // padding workgroups (uniform)
  if (c == nsl - 1) {
    // the extra workgroup of each row block carries the residual itself into x_out and zeroes the
    // third buffer (blockDim.x == d_model: one column per thread), so the slice workgroups issue no
    // conditional loads (a branch around a load costs them a full s_waitcnt)
    xq_t v[FF_R];
#pragma unroll
    for (int i = 0; i < FF_R; ++i) v[i] = a.x[(int64_t)min(b0 + i, a.B - 1) * K + tid];
    if (done) return;
    if (a.check_done && b0 == 0 && tid == 0) {
      // every row's layer-0 cross workgroup has counted itself in (the kernel before this one): none unfinished = the
      // batch is complete.  The token fed THIS step (position t) was the last one: t + 1 valid columns.
      if (a.state->n_unfinished == 0) { a.state->done = 1; a.state->out_len = a.state->t + 1; }
      a.state->n_unfinished = 0;
    }
#pragma unroll
    for (int i = 0; i < FF_R; ++i) {
      if (b0 + i < a.B) {
        const int64_t at = (int64_t)(b0 + i) * K + tid;
        atomicAdd(reinterpret_cast<unsigned long long*>(a.x_out + at), (unsigned long long)v[i]);
        a.x_zero[at] = 0;
      }
    }
    return;
  }
  const int kbeg = ks * 64 + 8 * g;
  const bool row_ok = r < FF_R && (b0 + r) < a.B;
  const int arow = b0 + (row_ok ? r : 0);            // padding rows read row b0; their activations are zeroed
  const T* Wi = reinterpret_cast<const T*>(a.Wi);
  const T* Wo = reinterpret_cast<const T*>(a.Wo);

  const xq_t* xr = a.x + (int64_t)arow * K + kbeg;
  longlong2 xraw[2][4];   // raw fixed-point rows: converted only after every load has been issued
  float4 g0[2], g1[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) xraw[s][q] = (M2M_FF_ABL & 2) ? make_longlong2(1 << 28, 1 << 27) : *reinterpret_cast<const longlong2*>(xr + 32 * s + 2 * q);
    g0[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 32 * s);
    g1[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 32 * s + 4);
  }
  Frag<T> wf[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) wf[j][s] = load_frag(Wi + (int64_t)(2 * FF_C * c + 16 * j + r) * K + kbeg + 32 * s);
  // phase-2 operands: this wave's four output n-tiles of wo, columns [32 c, 32 c + 32) of d_ff
  Frag<T> wo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wo[j] = load_frag(Wo + (int64_t)(16 * (4 * ks + j) + r) * a.d_ff + FF_C * c + 8 * g);
  // every load is in flight; the loop-state flag is forced here (two dependent scalar loads that the
  // compiler would otherwise sink to the first use, in front of the barrier on the critical path)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"s"(done));
  float4 x0[2], x1[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    x0[s] = make_float4(xq_flt(xraw[s][0].x), xq_flt(xraw[s][0].y), xq_flt(xraw[s][1].x), xq_flt(xraw[s][1].y));
    x1[s] = make_float4(xq_flt(xraw[s][2].x), xq_flt(xraw[s][2].y), xq_flt(xraw[s][3].x), xq_flt(xraw[s][3].y));
  }

  // ---- phase 1: RMSNorm + up projection of this slice ----
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
    ss += x0[s].x * x0[s].x + x0[s].y * x0[s].y + x0[s].z * x0[s].z + x0[s].w * x0[s].w +
          x1[s].x * x1[s].x + x1[s].y * x1[s].y + x1[s].z * x1[s].z + x1[s].w * x1[s].w;
  ss += lane_xor<16>(ss);
  ss += lane_xor<32>(ss);
  if (g == 0) ss_s[ks][r] = ss;
  __syncthreads();
  M2M_STAMP(4, 3);
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < KS; ++w) tot += ss_s[w][r];
  const float rs = row_ok ? rsqrtf(tot / (float)K + a.eps) : 0.f;   // rs = 0 zeroes the padding rows
  f32x4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float xv[8] = {g0[s].x * (x0[s].x * rs), g0[s].y * (x0[s].y * rs), g0[s].z * (x0[s].z * rs),
                         g0[s].w * (x0[s].w * rs), g1[s].x * (x1[s].x * rs), g1[s].y * (x1[s].y * rs),
                         g1[s].z * (x1[s].z * rs), g1[s].w * (x1[s].w * rs)};
    const Frag<T> fa = pack_frag<T>(xv);
#pragma unroll
    for (int j = 0; j < 4; ++j) mma32_16(acc[j], fa, wf[j][s]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[ks][j][(4 * g + i) * 17 + r] = acc[j][i];
  M2M_STAMP(4, 1);
  __syncthreads();
  if (done) return;   // uniform
  // gated activation: column cc of the slice lives in tile cc / 8 at columns (cc % 8) [wi_0] and (cc % 8) + 8 [wi_1];
  // k-slices are summed in a fixed order
  for (int idx = tid; idx < FF_R * FF_C; idx += 64 * KS) {   // rows FF_R..15 of the tile are padding and stay unread
    const int row = idx / FF_C, cc = idx % FF_C;
    const int t = cc >> 3, q = row * 17 + (cc & 7);
    float v0 = red[0][t][q], v1 = red[0][t][q + 8];
#pragma unroll
    for (int w = 1; w < KS; ++w) { v0 += red[w][t][q]; v1 += red[w][t][q + 8]; }
    hs[row * FF_HP + cc] = from_f32<T>(gelu_new_t<T>(v0) * v1);
  }
  __syncthreads();
  M2M_STAMP(4, 5);

  // ---- phase 2: [16 x 32] activations x [32 x d_model] slice of wo, added into the residual rows ----
  const Frag<T> fh = load_frag(hs + r * FF_HP + 8 * g);
  f32x4_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    mma32_16(o[j], fh, wo[j]);
  }
  if (M2M_FF_ABL & 1) return;
  if constexpr (FF_R <= 8) {
    // Rows 8..15 of the MFMA tile are padding, i.e. lanes g >= 2 hold nothing to store.  They take over half
    // of their partner's rows (lane ^ 32 <-> g ^ 2) so that every lane issues 2 adds per tile instead of
    // half the lanes issuing 4: lane (r, g) stores tile rows 4 (g & 1) + 2 (g >> 1) + {0, 1}.
    const int rl = 4 * (g & 1) + 2 * (g >> 1);
    xq_t* const p0 = a.x_out + (int64_t)(b0 + rl) * K + 64 * ks + r;
    const bool ok0 = rl < FF_R && b0 + rl < a.B, ok1 = rl + 1 < FF_R && b0 + rl + 1 < a.B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x2 = lane_xor<32>(o[j][2]), x3 = lane_xor<32>(o[j][3]);
      const float v0 = g < 2 ? o[j][0] : x2, v1 = g < 2 ? o[j][1] : x3;
      if (ok0) atomicAdd(reinterpret_cast<unsigned long long*>(p0 + 16 * j), (unsigned long long)xq_fix_guarded(v0, a.state));
      if (ok1) atomicAdd(reinterpret_cast<unsigned long long*>(p0 + 16 * j + K), (unsigned long long)xq_fix_guarded(v1, a.state));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 16 * (4 * ks + j) + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = b0 + 4 * g + i;
        if (4 * g + i < FF_R && row < a.B)
          atomicAdd(reinterpret_cast<unsigned long long*>(a.x_out + (int64_t)row * K + col), (unsigned long long)xq_fix_guarded(o[j][i], a.state));
      }
    }
  }
  M2M_STAMP(4, 2);
}

template <typename T, int FF_R>
static int launch_dec_ff_t(const DecFfArgs& a, hipStream_t st) {
  const int nsl = a.d_ff / FF_C + 1;                                          // + 1: the residual-carry workgroup
  dim3 grid((unsigned)(ceil_div(nsl, 8) * 8 * ceil_div(a.B, FF_R)));
  switch (a.d / 64) {
    case 2: hipLaunchKernelGGL((dec_ff_kernel<T, 2, FF_R>), grid, dim3(128), 0, st, a); break;
    case 4: hipLaunchKernelGGL((dec_ff_kernel<T, 4, FF_R>), grid, dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((dec_ff_kernel<T, 6, FF_R>), grid, dim3(384), 0, st, a); break;
    case 8: hipLaunchKernelGGL((dec_ff_kernel<T, 8, FF_R>), grid, dim3(512), 0, st, a); break;
    default: set_error("dec_ff: d_model=%d not supported (128/256/384/512)", a.d); return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// ---- the same sub-layer for LARGE chains: NSL consecutive hidden slices per workgroup (round 6) ----
// At 64 clips per chain the kernel above is bound by its atomics: 36 slice workgroups + the carry add into every residual element
// (885 K 64-bit atomics per launch; ablation build -DM2M_FF_ABL=1: 11.9 -> 7.6 us of kernel time, tools/r6_ablate.sh).  Here a
// workgroup walks NSL consecutive slices of its row block with the SAME normalised rows (read and normalised once), converts each
// slice's partial output to fixed point exactly as the first kernel does and sums the NSL integers in registers: ONE atomic per
// element and workgroup.  Integer adds are associative, so the residual stream - and every id - is bit-identical to the first form
// (tests/test_t5_gpu.py::test_multi_clip_attention_and_wide_ff_tiles_are_bit_identical); atomics and row reads per launch / NSL.
// The next slice's up-projection weights are requested as soon as this slice's MFMAs have consumed theirs, its down-projection
// weights once this slice's have been used.  FF_R <= 8 only (the lane pairing of the 8-row form).
template <typename T, int KS, int NSL>
__global__ __launch_bounds__(64 * KS) __attribute__((amdgpu_waves_per_eu(1, 2))) void dec_ff_multi_kernel(DecFfArgs a) {
  constexpr int FF_R = 8;
  __shared__ float ss_s[KS][16];
  __shared__ float red[KS][4][16 * 17];
  __shared__ __align__(16) T hs[16 * FF_HP];
  const int done = chain_done(a.state);
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  // XCD-aware order as in the first kernel: the row blocks that share a group of slices run on the same XCD
  const int nrb = (a.B + FF_R - 1) / FF_R, ngr = a.d_ff / (FF_C * NSL) + 1;   // row blocks; slice groups + the carry
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int c = xcd + 8 * (slot / nrb), b0 = (slot - (slot / nrb) * nrb) * FF_R;
  const int K = a.d;
  if (c >= ngr) return;   // padding workgroups (uniform)
  if (c == ngr - 1) {     // the carry workgroup of the row block: as in the first kernel
    xq_t v[FF_R];
#pragma unroll
    for (int i = 0; i < FF_R; ++i) v[i] = a.x[(int64_t)min(b0 + i, a.B - 1) * K + tid];
    if (done) return;
    if (a.check_done && b0 == 0 && tid == 0) {
      if (a.state->n_unfinished == 0) { a.state->done = 1; a.state->out_len = a.state->t + 1; }
      a.state->n_unfinished = 0;
    }
#pragma unroll
    for (int i = 0; i < FF_R; ++i) {
      if (b0 + i < a.B) {
        const int64_t at = (int64_t)(b0 + i) * K + tid;
        atomicAdd(reinterpret_cast<unsigned long long*>(a.x_out + at), (unsigned long long)v[i]);
        a.x_zero[at] = 0;
      }
    }
    return;
  }
  const int kbeg = ks * 64 + 8 * g;
  const bool row_ok = r < FF_R && (b0 + r) < a.B;
  const int arow = b0 + (row_ok ? r : 0);
  const T* Wi = reinterpret_cast<const T*>(a.Wi);
  const T* Wo = reinterpret_cast<const T*>(a.Wo);
  const int s0 = c * NSL;                                   // first slice of this workgroup

  const xq_t* xr = a.x + (int64_t)arow * K + kbeg;
  longlong2 xraw[2][4];
  float4 g0[2], g1[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) xraw[s][q] = (M2M_FF_ABL & 2) ? make_longlong2(1 << 28, 1 << 27) : *reinterpret_cast<const longlong2*>(xr + 32 * s + 2 * q);
    g0[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 32 * s);
    g1[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 32 * s + 4);
  }
  // NB register sets of slice weights (slice i in set i % NB, the slice NB ahead requested into the set just consumed).  The ablation
  // build prices the later slices' fetches at 2.9 of the kernel's 12.2 us with NB = 1 (-DM2M_FF_ABL=4: the later slices re-use the
  // first slice's weights), but TWO sets in flight measured SLOWER on the same box (tools/native_mc_sweep.py, us per step): 2 x 64
  // clips 351.7 / 352.7 against 344-348, with 2 slices 365.7 against 349-353, 2 x 32 clips 301.1 against 291.9 - as with the K/V
  // prefetch, bytes requested earlier delay everybody's latency-critical rows.  NB = 1.
#ifndef M2M_FF_MULTI_SETS
#define M2M_FF_MULTI_SETS 1
#endif
  constexpr int NB = (NSL > 1 && sizeof(T) == 2) ? M2M_FF_MULTI_SETS : 1;
  Frag<T> wf[NB][4][2];
  Frag<T> wo[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) wf[b][j][s] = load_frag(Wi + (int64_t)(2 * FF_C * (s0 + b) + 16 * j + r) * K + kbeg + 32 * s);
#pragma unroll
    for (int j = 0; j < 4; ++j) wo[b][j] = load_frag(Wo + (int64_t)(16 * (4 * ks + j) + r) * a.d_ff + FF_C * (s0 + b) + 8 * g);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"s"(done));
  float4 x0[2], x1[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    x0[s] = make_float4(xq_flt(xraw[s][0].x), xq_flt(xraw[s][0].y), xq_flt(xraw[s][1].x), xq_flt(xraw[s][1].y));
    x1[s] = make_float4(xq_flt(xraw[s][2].x), xq_flt(xraw[s][2].y), xq_flt(xraw[s][3].x), xq_flt(xraw[s][3].y));
  }
  // ---- RMSNorm once: the normalised fragments serve every slice ----
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
    ss += x0[s].x * x0[s].x + x0[s].y * x0[s].y + x0[s].z * x0[s].z + x0[s].w * x0[s].w +
          x1[s].x * x1[s].x + x1[s].y * x1[s].y + x1[s].z * x1[s].z + x1[s].w * x1[s].w;
  ss += lane_xor<16>(ss);
  ss += lane_xor<32>(ss);
  if (g == 0) ss_s[ks][r] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < KS; ++w) tot += ss_s[w][r];
  const float rs = row_ok ? rsqrtf(tot / (float)K + a.eps) : 0.f;
  Frag<T> fa[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float xv[8] = {g0[s].x * (x0[s].x * rs), g0[s].y * (x0[s].y * rs), g0[s].z * (x0[s].z * rs),
                         g0[s].w * (x0[s].w * rs), g1[s].x * (x1[s].x * rs), g1[s].y * (x1[s].y * rs),
                         g1[s].z * (x1[s].z * rs), g1[s].w * (x1[s].w * rs)};
    fa[s] = pack_frag<T>(xv);
  }
  // lane (r, g) stores tile rows 4 (g & 1) + 2 (g >> 1) + {0, 1} (the 8-row form's lane pairing)
  const int rl = 4 * (g & 1) + 2 * (g >> 1);
  xq_t sum0[4], sum1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sum0[j] = 0; sum1[j] = 0; }

#pragma unroll
  for (int i = 0; i < NSL; ++i) {
    // ---- phase 1: up projection of slice s0 + i ----
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma32_16(acc[j], fa[s], wf[i % NB][j][s]);
    if (i + NB < NSL && !(M2M_FF_ABL & 4)) {   // the up-projection weights of the slice NB ahead, into the set just consumed
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[i % NB][j][s] = load_frag(Wi + (int64_t)(2 * FF_C * (s0 + i + NB) + 16 * j + r) * K + kbeg + 32 * s);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[ks][j][(4 * g + q) * 17 + r] = acc[j][q];
    __syncthreads();
    if (done) return;   // uniform
    for (int idx = tid; idx < FF_R * FF_C; idx += 64 * KS) {
      const int row = idx / FF_C, cc = idx % FF_C;
      const int t = cc >> 3, q = row * 17 + (cc & 7);
      float v0 = red[0][t][q], v1 = red[0][t][q + 8];
#pragma unroll
      for (int w = 1; w < KS; ++w) { v0 += red[w][t][q]; v1 += red[w][t][q + 8]; }
      hs[row * FF_HP + cc] = from_f32<T>(gelu_new_t<T>(v0) * v1);
    }
    __syncthreads();
    // ---- phase 2: down projection of the slice, converted to fixed point per slice, summed as integers ----
    const Frag<T> fh = load_frag(hs + r * FF_HP + 8 * g);
    f32x4_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      mma32_16(o[j], fh, wo[i % NB][j]);
    }
    if (i + NB < NSL && !(M2M_FF_ABL & 4)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wo[i % NB][j] = load_frag(Wo + (int64_t)(16 * (4 * ks + j) + r) * a.d_ff + FF_C * (s0 + i + NB) + 8 * g);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x2 = lane_xor<32>(o[j][2]), x3 = lane_xor<32>(o[j][3]);
      const float v0 = g < 2 ? o[j][0] : x2, v1 = g < 2 ? o[j][1] : x3;
      sum0[j] += xq_fix_guarded(v0, a.state);
      sum1[j] += xq_fix_guarded(v1, a.state);
    }
  }
  if (M2M_FF_ABL & 1) return;
  xq_t* const p0 = a.x_out + (int64_t)(b0 + rl) * K + 64 * ks + r;
  const bool ok0 = rl < FF_R && b0 + rl < a.B, ok1 = rl + 1 < FF_R && b0 + rl + 1 < a.B;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (ok0) atomicAdd(reinterpret_cast<unsigned long long*>(p0 + 16 * j), (unsigned long long)sum0[j]);
    if (ok1) atomicAdd(reinterpret_cast<unsigned long long*>(p0 + 16 * j + K), (unsigned long long)sum1[j]);
  }
}

template <typename T, int NSL>
static int launch_dec_ff_multi_t(const DecFfArgs& a, hipStream_t st) {
  const int ngr = a.d_ff / (FF_C * NSL) + 1;
  dim3 grid((unsigned)(ceil_div(ngr, 8) * 8 * ceil_div(a.B, 8)));
  switch (a.d / 64) {
    case 2: hipLaunchKernelGGL((dec_ff_multi_kernel<T, 2, NSL>), grid, dim3(128), 0, st, a); break;
    case 4: hipLaunchKernelGGL((dec_ff_multi_kernel<T, 4, NSL>), grid, dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((dec_ff_multi_kernel<T, 6, NSL>), grid, dim3(384), 0, st, a); break;
    case 8: hipLaunchKernelGGL((dec_ff_multi_kernel<T, 8, NSL>), grid, dim3(512), 0, st, a); break;
    default: set_error("dec_ff: d_model=%d not supported (128/256/384/512)", a.d); return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

static int launch_dec_ff(int precision, const DecFfArgs& a, int rows, int slices, hipStream_t st) {
  if (a.d % 64 != 0 || a.d_ff % FF_C != 0) {
    set_error("dec_ff: d_model=%d must be a multiple of 64 and d_ff=%d of %d", a.d, a.d_ff, FF_C);
    return M2M_ERR_INVALID;
  }
  const bool bf = precision == M2M_PREC_BF16;
  if (slices > 1 && (a.d_ff / FF_C) % slices == 0) {        // several hidden slices per workgroup (large chains)
    if (slices == 2) return bf ? launch_dec_ff_multi_t<bf16_t, 2>(a, st) : launch_dec_ff_multi_t<float, 2>(a, st);
    if (slices == 4) return bf ? launch_dec_ff_multi_t<bf16_t, 4>(a, st) : launch_dec_ff_multi_t<float, 4>(a, st);
  }
  if (rows == 16) return bf ? launch_dec_ff_t<bf16_t, 16>(a, st) : launch_dec_ff_t<float, 16>(a, st);
  return bf ? launch_dec_ff_t<bf16_t, 8>(a, st) : launch_dec_ff_t<float, 8>(a, st);
}

// ======================================================= decode attention ====
// Fused per-(clip, head) kernel:  RMSNorm(x[b]) -> this head's projection (self: q,k,v + cache
// append at slot t; cross: q) -> single-pass ("online") softmax attention over the cached keys ->
// o[b, head] -> this head's slice of the output projection, ADDED into the residual row (integer
// adds: order-independent).  Every byte of K/V is used once, so it goes HBM -> registers (16 B per lane) with no
// LDS staging, and because the softmax is online there is no workgroup-wide reduction between
// reading K and reading V: both are requested together at kernel start and stream continuously
// while the norm and the projection run; the only reductions are at the very end.
struct DecAttnArgs {
  const xq_t* x;         // [B, d] fixed-point residual stream, complete: read by the norm (never written here)
  xq_t* x_out;           // [B, d] zero on entry: residual + every head's output projection are ADDED into it
  xq_t* x_zero;          // [B, d] third buffer of the rotation: zeroed for the kernel after next
  const float* ln_w;     // [d] RMSNorm weight of this sub-layer
  float eps;
  int d;                 // d_model
  const void* Wp;        // self: wqkv [3*inner, d] ; cross: wcq [inner, d]        (T)
  void* Kc;              // self: K cache [B][H][kv_stride][64] (slot t is written) ; cross: cross K (read only)
  void* Vc;
  int kv_stride;         // keys allocated per (b,h): Lmax (self) or S (cross)
  int n_keys;            // cross: S
  int self_len_override; // bench only: pretend t = self_len_override - 1
  const float* bias;     // self: [H][Lmax] by n = q_pos - k_pos
  int bias_stride;
  const void* Wo;        // [d, inner] T output projection of this sub-layer (this head uses columns [64h, 64h+64))
  int H, inner;
  DecState* state;
  // headless greedy loop, layer 0 only (null / 0 elsewhere)
  const float* emb;            // self: [V, d] token embedding — the input row is emb[token decoded from keys[b]]
  unsigned long long* keys;    // [B] arg-max keys of the previous step (self: read; cross with `book`: read, then cleared)
  int* finished;               // [B]
  int64_t* tokens;             // [session rows, max_len]: the WHOLE token matrix; row of slot b = tok_row[b]
  const int* tok_row;          // [B] clip (token-matrix row) decoded in slot b of this chain: identity until live rows are re-packed
  int max_len, V, pad_id, eos_id;
  int book;                    // cross: head 0's workgroup of every row does the row's bookkeeping at its end
  // Finished-row early-out (greedy loop): fin_skip[b * fin_stride] != 0 -> row b has emitted EOS; its tokens are pad whatever
  // is computed (hf generation/utils.py:2929), so its workgroups stop re-requesting K/V after the entry prefetch.  Never null:
  // "never skip" is a pointer to DecState::zero with stride 0 (an unconditional scalar load: a branch around it would split
  // the prologue's block of loads, see the kernel).
  const int* fin_skip;
  int fin_stride;
};

// K/V rows are read once per step.  When the per-step K/V working set is larger than the 256 MB
// Infinity Cache (540 MB at B = 32) they are loaded NON-TEMPORAL, so they do not evict what the
// latency-bound kernels re-read every step (30 MB of weights, the residual rows): 290 -> 271 ms per
// batch at B = 32.  When it fits (B <= 8) the default policy keeps K/V itself cache-resident across
// steps, which is faster (B = 1: 196 vs 204 us per step) — NT is a launch-time template choice.
typedef unsigned int m2m_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, typename V> __device__ inline V kv_load(const V* p) {
  if constexpr (NT) return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const m2m_u32x4*>(p)));
  else return *p;
}
#define M2M_KV_LOAD(p) kv_load<NT>(p)

// softmax exponential of a non-positive argument.  fp32 (parity) mode: accurate expf (~10 instructions).
// bf16 mode: one multiply + v_exp_f32 (relative error ~|x| * 6e-8, far below bf16's 4e-3); every key costs
// one of these in all 16 waves, so it is visible in the stream phase (216.8 -> 215.2 ms).
template <typename T> __device__ inline float m2m_exp(float x) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  else return expf(x);
}

// acc += <8 (bf16) / 4 (fp32) weights of one 16-byte chunk, the matching LDS-resident inputs>.
// bf16: the inputs are kept PACKED in LDS (they are rounded to bf16 anyway) and each pair goes through
// v_dot2c_f32_bf16 - 4 instructions per chunk instead of 8 converts + 8 FMAs; the per-head projections
// are VALU-bound (16 waves on 4 SIMDs), so this is their time.  fp32 (parity mode): plain FMA chain.
typedef __bf16 m2m_bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(m2m_bf16x2, a), __builtin_bit_cast(m2m_bf16x2, b), c, false);
}
template <typename T> __device__ inline float chunk_dot(const Vec16<T>& w, const float* in_lds, int chunk, float acc);
template <> __device__ inline float chunk_dot<bf16_t>(const Vec16<bf16_t>& w, const float* in_lds, int chunk, float acc) {
  const uint4 x = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(in_lds) + chunk * 4);
  acc = dot2_bf16(w.v.x, x.x, acc);
  acc = dot2_bf16(w.v.y, x.y, acc);
  acc = dot2_bf16(w.v.z, x.z, acc);
  return dot2_bf16(w.v.w, x.w, acc);
}
template <> __device__ inline float chunk_dot<float>(const Vec16<float>& w, const float* in_lds, int chunk, float acc) {
  const float* hp = in_lds + chunk * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = fmaf(hp[e], w.get(e), acc);
  return acc;
}
// store element i of an LDS input vector in the layout chunk_dot reads (bf16: packed halves; fp32: floats)
template <typename T> __device__ inline void put_in(float* in_lds, int i, float v) {
  if constexpr (sizeof(T) == 2) reinterpret_cast<bf16_t*>(in_lds)[i] = from_f32<bf16_t>(v);
  else in_lds[i] = v;
}

template <typename T, bool SELF, bool NT, bool FETCH = false>
__global__ __launch_bounds__(1024) void dec_attn_kernel(DecAttnArgs a) {
  static_assert(!FETCH || SELF, "only the layer-0 self-attention fetches its input row from the embedding table");
  constexpr int E = 16 / sizeof(T);      // elements per 16-byte chunk: 8 (bf16) / 4 (fp32)
  constexpr int LPR = DK / E;            // lanes per key row (a "group"): 8 / 16
  constexpr int KPW = 64 / LPR;          // keys per wave-load: 8 / 4
  constexpr int KPB = 16 * KPW;          // keys per block round: 128 / 64
  constexpr int NOUT = DK;               // the query projection (self and cross alike): 64 outputs
  constexpr int LPO = 16;                // lanes per projection output
  // self only: the k,v rows this step appends are projected LATER (their weights stream in under
  // the K/V cache reads and they are needed only for the final key), so the prologue is as short
  // as the cross kernel's: 128 outputs x 8 lanes
  constexpr int LPO2 = 8;
  constexpr int WMAX2 = 6;               // k/v weight chunks per lane held in registers (d_model 384, bf16)
  // K/V rounds (one K row + one V row per lane = 32 KB per workgroup) kept in flight by the rolling
  // prefetch.  The first PF rounds are requested at kernel entry, right behind the prologue's own
  // loads (the x row first: loads retire in order), so the stream is already running while the norm
  // and the query projection execute.  Two rounds is the measured optimum (B = 32, ms per batch:
  // PF 1 / 2 / 3 = 219.4 / 217.4 / 229.2): deeper windows flood the fabric queues ahead of the
  // latency-critical x / weight loads of workgroups that start a little later, and the whole 220 KB
  // stream at once stalls the issuing waves (prologue done at ~9 us).
#ifndef M2M_DA_PF_CROSS
#define M2M_DA_PF_CROSS 2
#endif
#ifndef M2M_DA_PF_SELF
#define M2M_DA_PF_SELF 2
#endif
#ifndef M2M_DA_REQUEST_FIRST
#define M2M_DA_REQUEST_FIRST 0
#endif
  // fp32 (parity) mode, cross-attention: 4 rounds.  Its per-workgroup stream is 442 KB at S = 864 against a ~3 us prologue, and the
  // same-box A/B (tools/r6_fp32_sweep.sh, 32 x S = 864, us per step incl. encoder) reads 345.1 / 345.5 / 345.9 with 2 rounds, 345.5
  // with 3, 340.9 with 4; 3 rounds in the self-attention: 353.4.
#ifndef M2M_DA_PF_CROSS_F32
#define M2M_DA_PF_CROSS_F32 4
#endif
  constexpr int PF = SELF ? M2M_DA_PF_SELF : (sizeof(T) == 4 ? M2M_DA_PF_CROSS_F32 : M2M_DA_PF_CROSS);
  constexpr int WMAX = 3;                // q weight chunks per lane held in registers (d_model 384, bf16)
  using V16 = decltype(Vec16<T>().v);
  extern __shared__ __align__(16) float hn[];   // [d] normalised input row (already rounded to T)
  __shared__ float redw[16], redl[16], redlf[16];
  __shared__ float redo[16][DK];
  __shared__ __align__(16) float redg[16 * (64 / LPR)][DK];   // per-group partial outputs (32 KB bf16 / 16 KB fp32)
  __shared__ __align__(16) float qs[DK];
  __shared__ __align__(16) float kn[DK];
  __shared__ __align__(16) float vn[DK];
  __shared__ __align__(16) float oh[DK];
  M2M_STAMP_DECL
  M2M_STAMP(6 + (SELF ? 1 : 0), 0);
  // loop state: requested now, first consumed AFTER the prologue loads below have been issued, so
  // its round trip overlaps theirs instead of preceding it (every kernel of the step is latency-bound:
  // at B = 1 a step still takes 196 us, i.e. 7.5 us per kernel with nothing to stream)
  const int st_t = a.state->t;
  const int st_done = a.state->done | (st_t >= a.state->max_steps);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = blockIdx.x, b = blockIdx.y;   // grid (H, B): linear id = 8 b + h as before (a head's clips on one XCD), no division
  // finished-row flag (scalar, requested with the loop state; first consumed after the prologue).  In the layer-0 cross kernel
  // head 0 of the row WRITES the flag at its end while its siblings may still be starting: they can disagree at the one step in
  // which the row finishes — harmless, that step's output is the first one forced to pad.
  const int row_fin = a.fin_skip[(int64_t)b * a.fin_stride];
  const int sub = lane % LPR;
  const int kslot = wave * KPW + lane / LPR;
  T* Kb = reinterpret_cast<T*>(a.Kc) + ((int64_t)b * a.H + hh) * a.kv_stride * DK;
  T* Vb = reinterpret_cast<T*>(a.Vc) + ((int64_t)b * a.H + hh) * a.kv_stride * DK;

  // ---- 0. requests for the prologue: x row, norm weights, this lane's projection weights.
  //         Every kernel starts with a cold L2 for data other XCDs produced, so these come from
  //         memory; if the 220 KB K/V stream of this workgroup were requested at the same time they
  //         would queue behind it in the fabric (measured: the norm then completes only after ~9 us).
  //         Only the first PF rounds of the stream follow them here. ----
  const int xc = min(tid * 4, a.d - 4);
  const int on_ = min(tid >> 1, a.d - 1), opart = tid & 1;
  // RAW fixed-point words: converted only after every request below has been issued (the conversion needs the
  // data, and in program order it would put the wait for x in front of the K/V prefetch)
  longlong2 xr0 = make_longlong2(0, 0), xr1 = make_longlong2(0, 0);
  xq_t xres = 0;
  unsigned long long kraw = 0;
  int fin0 = 0;
  if constexpr (FETCH) {          // headless layer 0: the row is an embedding row, chosen by the previous step's arg-max key
    kraw = a.keys[b];
    fin0 = a.finished[b];
  } else {
    xr0 = *reinterpret_cast<const longlong2*>(a.x + (int64_t)b * a.d + xc);
    xr1 = *reinterpret_cast<const longlong2*>(a.x + (int64_t)b * a.d + xc + 2);
    // head 0 also carries the residual row into x_out: its raw fixed-point value, requested now
    xres = a.x[(int64_t)b * a.d + on_];   // unconditional: a branch would split the block of loads
  }
  const float4 gv = *reinterpret_cast<const float4*>(a.ln_w + xc);
  __builtin_amdgcn_sched_barrier(0);               // the latency-critical row goes out FIRST (loads retire in order)
  const int po = min(tid / LPO, NOUT - 1), part = tid % LPO;
  const int which = po / DK, dd = po - which * DK;
  const T* wrow = reinterpret_cast<const T*>(a.Wp) + ((int64_t)which * a.inner + hh * DK + dd) * a.d;
  const int cnt = a.d / E / LPO;         // chunks per lane
  Vec16<T> w[WMAX];
#pragma unroll
  for (int u = 0; u < WMAX; ++u) w[u].v = *reinterpret_cast<const V16*>(wrow + (min(u, cnt - 1) * LPO + part) * E);
  // self: this head's relative-position bias row goes to LDS.  Read from global memory inside the key
  // loop it would sit behind the next round's K/V request in the wave's in-order vmcnt queue: the
  // compiler then has to wait vmcnt(0) every round, and the stream runs at one round per memory latency.
  float* const biasl = hn + a.d;                      // [kv_stride] (self only)
  constexpr int BPT = 2;                               // bias entries per thread held in registers (Lmax <= 2048; more: loop below)
  float bv[SELF ? BPT : 1];
  if (SELF) {
#pragma unroll
    for (int u = 0; u < BPT; ++u) bv[u] = a.bias[(int64_t)hh * a.bias_stride + min(tid + 1024 * u, a.bias_stride - 1)];
  }

  // No early exit on st_done: a branch here makes the compiler sink the weight loads above below
  // it, behind the wait for x (one more serial round trip).  A finished chain runs at most the rest
  // of its graph with every store suppressed.
  const int t = SELF ? (a.self_len_override > 0 ? a.self_len_override - 1 : st_t) : 0;
  const int n_prev = SELF ? t : a.n_keys;            // keys that are read from memory
  const int last = max(n_prev - 1, 0);

  // ---- 1. RMSNorm of x[b] -> hn (rounded to the GEMM-input type T) ----
  Vec16<T> kv[PF], vv[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {   // clamped addresses, never predicated
    const int64_t off = (int64_t)min(kslot + u * KPB, last) * DK + sub * E;
    kv[u].v = M2M_KV_LOAD(reinterpret_cast<const V16*>(Kb + off));
    vv[u].v = M2M_KV_LOAD(reinterpret_cast<const V16*>(Vb + off));
  }
  // everything above is in flight before anything waits: without the fence the scheduler places the wait for x
  // (and its int64 -> float conversion) ahead of the weight loads and of the K/V prefetch (the ISA showed the
  // prefetch going out only after x had arrived, ~2 us into the kernel)
  __builtin_amdgcn_sched_barrier(0);
  // Only the waves that own a piece of the row (d / 4 threads: 2 of the 16 at d_model 384) convert it and
  // reduce; the others just post a zero, so the owners do not share their SIMD's issue slots with 3 waves of
  // duplicate work (waves are spread round-robin over the 4 SIMDs).
  const bool own_wave = wave * 256 < a.d;           // wave-uniform
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
  float eres = 0.f;
  float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (FETCH) {          // second (dependent) round trip of the headless prologue: the embedding row
    const int tok = amax_key_token(kraw, fin0, a.V, a.pad_id);
    const float* er = a.emb + (int64_t)tok * a.d;
    e4 = *reinterpret_cast<const float4*>(er + xc);
    eres = er[on_];
  }
  {
    const bool own = tid * 4 < a.d;
    if (own_wave) {
      if constexpr (FETCH) xv = e4;
      else xv = make_float4(xq_flt(xr0.x), xq_flt(xr0.y), xq_flt(xr1.x), xq_flt(xr1.y));
      float ss = own ? (xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w) : 0.f;
      ss = wave_sum(ss);
      if (lane == 0) redw[wave] = ss;
    } else if (lane == 0) {
      redw[wave] = 0.f;
    }
    __syncthreads();
    M2M_STAMP(6 + (SELF ? 1 : 0), 4);
    if (own) {
      float tot = 0.f;
#pragma unroll
      for (int wv = 0; wv < 16; ++wv) tot += redw[wv];
      const float rs = rsqrtf(tot / (float)a.d + a.eps);
      put_in<T>(hn, xc + 0, gv.x * (xv.x * rs));
      put_in<T>(hn, xc + 1, gv.y * (xv.y * rs));
      put_in<T>(hn, xc + 2, gv.z * (xv.z * rs));
      put_in<T>(hn, xc + 3, gv.w * (xv.w * rs));
    }
    __syncthreads();
    M2M_STAMP(6 + (SELF ? 1 : 0), 5);
  }

  // ---- 2. this head's projection: NOUT outputs, LPO lanes each, 16-byte chunks strided over lanes ----
  {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < WMAX; ++u) {
      if (u < cnt) {
        acc = chunk_dot<T>(w[u], hn, u * LPO + part, acc);
      }
    }
    for (int i = WMAX; i < cnt; i += 4) {   // d_model / dtype combinations beyond the register budget (fp32)
      Vec16<T> w2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w2[u].v = *reinterpret_cast<const V16*>(wrow + (min(i + u, cnt - 1) * LPO + part) * E);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u < cnt) {
          acc = chunk_dot<T>(w2[u], hn, (i + u) * LPO + part, acc);
        }
      }
    }
    acc = group_sum<LPO>(acc);
#ifndef M2M_STAMPS_MERGE
    M2M_STAMP(6 + (SELF ? 1 : 0), 6);
#endif
    if (part == 0 && tid < NOUT * LPO) qs[dd] = acc;      // q stays fp32
  }
  if (SELF) {
#pragma unroll
    for (int u = 0; u < BPT; ++u)
      if (tid + 1024 * u < a.bias_stride) biasl[tid + 1024 * u] = bv[u];
    for (int n = tid + 1024 * BPT; n < a.bias_stride; n += 1024) biasl[n] = a.bias[(int64_t)hh * a.bias_stride + n];
  }
  __syncthreads();
  M2M_STAMP(6 + (SELF ? 1 : 0), 1);

  // this head's slice of the output projection (2 lanes per output column, 32 inputs each):
  // requested now so it arrives under the K/V stream
  constexpr int OCH = 32 / E;            // 16-byte chunks per lane: 4 (bf16) / 8 (fp32)
  const T* worow = reinterpret_cast<const T*>(a.Wo) + (int64_t)on_ * a.inner + hh * DK + opart * 32;
  Vec16<T> wo[OCH];
#pragma unroll
  for (int u = 0; u < OCH; ++u) wo[u].v = *reinterpret_cast<const V16*>(worow + u * E);
  // self: weights of the k,v rows this step appends (128 outputs x 8 lanes), also under the stream
  const int o2 = tid >> 3, part2 = tid & 7;
  const int which2 = 1 + (o2 >> 6), dd2 = o2 & 63;
  const T* wrow2 = reinterpret_cast<const T*>(a.Wp) + ((int64_t)which2 * a.inner + hh * DK + dd2) * a.d;
  const int cnt2 = a.d / E / LPO2;
  Vec16<T> wkv[SELF ? WMAX2 : 1];
  if (SELF) {
#pragma unroll
    for (int u = 0; u < WMAX2; ++u) wkv[u].v = *reinterpret_cast<const V16*>(wrow2 + (min(u, cnt2 - 1) * LPO2 + part2) * E);
  }

  // ---- 3. single-pass attention: each group of LPR lanes walks its keys with a running
  //         (max, sum, weighted-V) triple; no workgroup-wide step until the end ----
  // A finished row walks NO cached keys: the two rounds requested at entry are dropped, nothing is re-requested (32 KB instead
  // of the whole stream for this workgroup); the self kernel still appends and visits this step's own key, the cross kernel's
  // empty softmax gives a zero head output below.  Nothing of a finished row is observable: its tokens are pad.
  const int n_live = row_fin ? 0 : n_prev;
  float qv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) qv[e] = qs[sub * E + e];
  float m_run = -1e30f, l_run = 0.f;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  // one exponential per key: of (alpha, p) = (exp(m_run - m_new), exp(s - m_new)) one is always exp(0) = 1
  auto visit = [&](float s, const float (&vrow)[E]) {
    const bool up = s > m_run;
    const float m_new = up ? s : m_run;
    const float ex = m2m_exp<T>(up ? m_run - s : s - m_run);
    const float alpha = up ? ex : 1.f;
    const float p = up ? 1.f : ex;
    l_run = fmaf(l_run, alpha, p);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = fmaf(acc[e], alpha, p * vrow[e]);
    m_run = m_new;
  };
  // One round: consume slot u (scores, online-softmax update), optionally re-request the round PF ahead
  // into the same slot.  REISSUE is a compile-time choice per loop: in the main loop every re-request
  // exists, so the loads are unconditional and the compiler can count them (s_waitcnt vmcnt(N) keeps the
  // next round in flight while this one is consumed); a runtime "is there a round PF ahead" test around
  // the loads makes it fall back to vmcnt(0) at every round, i.e. batches instead of a rolling window.
  auto round = [&](const int k0, const int u, Vec16<T>& ks, Vec16<T>& vs, const int reissue /*0 no, 1 yes, 2 if it exists*/) {
    const int key = k0 + kslot + u * KPB;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s = fmaf(qv[e], ks.get(e), s);
    s = group_sum<LPR>(s);                                                   // every lane of the group gets the sum
#if M2M_DA_REQUEST_FIRST       // rounds 1-5: the slot re-requested BEFORE its visit (see below)
    float vrow[E];
#pragma unroll
    for (int e = 0; e < E; ++e) vrow[e] = vs.get(e);
    if (reissue == 1 || (reissue == 2 && k0 + (u + PF) * KPB < n_live)) {   // workgroup-uniform, no lane is predicated
      const int64_t off = (int64_t)min(key + PF * KPB, last) * DK + sub * E;
      ks.v = M2M_KV_LOAD(reinterpret_cast<const V16*>(Kb + off));
      vs.v = M2M_KV_LOAD(reinterpret_cast<const V16*>(Vb + off));
    }
    if (key < n_live) {   // VALU-only predicate
      if (SELF) s += biasl[t - key];
      visit(s, vrow);
    }
#else
    // The slot is re-requested AFTER it has been consumed (round 6).  With the request in front of the visit the compiler sinks V's
    // unpacking into the predicated block, keeps the packed V alive past the request, loads into a SECOND register set and copies
    // it back behind s_waitcnt vmcnt(0) at the end of EVERY iteration (ISA of rounds 1-5: vmcnt(2), 4 v_mov, vmcnt(0), 4 v_mov at
    // the back edge): the "rolling window" drained once per iteration - two rounds per memory round trip and CU.
    if (key < n_live) {   // VALU-only predicate
      float vrow[E];
#pragma unroll
      for (int e = 0; e < E; ++e) vrow[e] = vs.get(e);
      if (SELF) s += biasl[t - key];
      visit(s, vrow);
    }
    if (reissue == 1 || (reissue == 2 && k0 + (u + PF) * KPB < n_live)) {   // workgroup-uniform, no lane is predicated
      const int64_t off = (int64_t)min(key + PF * KPB, last) * DK + sub * E;
      ks.v = M2M_KV_LOAD(reinterpret_cast<const V16*>(Kb + off));
      vs.v = M2M_KV_LOAD(reinterpret_cast<const V16*>(Vb + off));
    }
#endif
  };
  int k0 = 0;
  // main loop: all PF re-requests of the iteration exist (the last one targets round k0 / KPB + 2 PF - 1)
  for (; k0 + (2 * PF - 1) * KPB < n_live; k0 += PF * KPB) {
#pragma unroll
    for (int u = 0; u < PF; ++u) round(k0, u, kv[u], vv[u], 1);
  }
  // tail (at most two iterations): past the end nothing is requested, so the merge barrier below does
  // not wait for a useless round trip
  for (; k0 < n_live; k0 += PF * KPB) {
#pragma unroll
    for (int u = 0; u < PF; ++u) round(k0, u, kv[u], vv[u], 2);
  }
  if (SELF) {
    // project, round to T, append to the cache and publish through LDS the k,v rows of this step
    float acc2 = 0.f;
#pragma unroll
    for (int u = 0; u < WMAX2; ++u) {
      if (u < cnt2) {
        acc2 = chunk_dot<T>(wkv[u], hn, u * LPO2 + part2, acc2);
      }
    }
    for (int i = WMAX2; i < cnt2; i += 4) {   // d_model / dtype combinations beyond the register budget (fp32)
      Vec16<T> w2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w2[u].v = *reinterpret_cast<const V16*>(wrow2 + (min(i + u, cnt2 - 1) * LPO2 + part2) * E);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u < cnt2) {
          acc2 = chunk_dot<T>(w2[u], hn, (i + u) * LPO2 + part2, acc2);
        }
      }
    }
    acc2 = group_sum<LPO2>(acc2);
    if (part2 == 0) {
      const T r = from_f32<T>(acc2);                       // k, v are stored (and used) rounded to T
      const int64_t slot = (int64_t)t * DK + dd2;
      if (which2 == 1) { kn[dd2] = to_f32(r); if (!st_done) Kb[slot] = r; }
      else             { vn[dd2] = to_f32(r); if (!st_done) Vb[slot] = r; }
    }
    __syncthreads();
  }
  if (SELF && wave == 0 && lane < LPR) {   // the key/value appended this step (relative position 0): group 0
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s = fmaf(qv[e], kn[sub * E + e], s);
    s = group_sum<LPR>(s);
    s += biasl[0];
    float vrow[E];
#pragma unroll
    for (int e = 0; e < E; ++e) vrow[e] = vn[sub * E + e];
    visit(s, vrow);
  }
  M2M_STAMP(6 + (SELF ? 1 : 0), 2);

  // ---- 4. merge the 128 key groups (16 waves x 64 / LPR groups) ----
  // Each group's partial output goes to LDS scaled to its WAVE's maximum (one LDS write per lane
  // instead of 3 cross-lane exchanges per accumulator register: the exchanges were the bulk of the
  // tail's VALU time with 16 waves on 4 SIMDs); waves are then summed group by group, and the 16 wave
  // results are brought to the global maximum by the 64 threads that finish the row.
  {
    constexpr int GPW = 64 / LPR;                    // groups per wave: 8 (bf16) / 4 (fp32)
    const float mw = wave_max(m_run);
    const float scale = m2m_exp<T>(m_run - mw);          // groups that saw no key have m_run = -1e30 -> 0 (or 1 if the whole wave saw none: l = acc = 0)
    float lsum = (sub == 0) ? l_run * scale : 0.f;   // l is replicated over a group's lanes: count it once
    lsum = wave_sum(lsum);
    float* gp = &redg[wave * GPW + lane / LPR][sub * E];
#pragma unroll
    for (int e = 0; e < E; e += 4)
      *reinterpret_cast<float4*>(gp + e) = make_float4(acc[e] * scale, acc[e + 1] * scale, acc[e + 2] * scale, acc[e + 3] * scale);
    if (lane == 0) { redw[wave] = mw; redl[wave] = lsum; }
    __syncthreads();
#ifdef M2M_STAMPS_MERGE
    M2M_STAMP(6 + (SELF ? 1 : 0), 6);
#endif
    {
      // every wave brings ITS partial sums to the global maximum here (one exponential per wave, in parallel),
      // so the single wave that finishes the row below only adds: the 16 exponentials it used to evaluate were
      // ~0.4 us of serial VALU time per launch
      float M = redw[lane & 15];
      M = fmaxf(M, lane_xor<8>(M)); M = fmaxf(M, lane_xor<4>(M)); M = fmaxf(M, lane_xor<2>(M)); M = fmaxf(M, lane_xor<1>(M));
      const float fw = m2m_exp<T>(redw[wave] - M);
      float sw = redg[wave * GPW][lane];             // thread (wave, lane = dim): this wave's groups, fixed order
#pragma unroll
      for (int j = 1; j < GPW; ++j) sw += redg[wave * GPW + j][lane];
      redo[wave][lane] = sw * fw;
      if (lane == 0) redlf[wave] = redl[wave] * fw;
    }
    __syncthreads();
    if (tid < DK) {
      float s = 0.f, L = 0.f;
#pragma unroll
      for (int wv = 0; wv < 16; ++wv) {
        s += redo[wv][tid];
        L += redlf[wv];
      }
      put_in<T>(oh, tid, L > 0.f ? s / L : 0.f);     // the projection input is rounded to T, as every GEMM input (L = 0: a finished row's empty cross softmax)
    }
    __syncthreads();
    M2M_STAMP(6 + (SELF ? 1 : 0), 7);
  }
  // ---- 5. output projection of this head, accumulated into the residual row ----
  {
    float accp = 0.f;
#pragma unroll
    for (int u = 0; u < OCH; ++u) {
      accp = chunk_dot<T>(wo[u], oh, opart * OCH + u, accp);
    }
    accp += lane_xor<1>(accp);
    if (opart == 0 && tid < 2 * a.d && !st_done) {
      xq_t add = xq_fix_guarded(accp, a.state);
      if (hh == 0) add += FETCH ? xq_fix_guarded(eres, a.state) : xres;   // head 0 also carries the residual itself
      atomicAdd(reinterpret_cast<unsigned long long*>(a.x_out + (int64_t)b * a.d + on_), (unsigned long long)add);
      if (hh == a.H - 1) a.x_zero[(int64_t)b * a.d + on_] = 0;
    }
  }
  if (!SELF && a.book && hh == 0 && tid == 0) {
    // headless bookkeeping of row b, after the layer-0 self-attention kernel (all 8 head workgroups of the row) has consumed
    // the key: the token fed at this step goes into the token matrix, EOS finishes the row, the key is cleared for this
    // step's lm_head.  (hf generation/utils.py:2925-2937.)
    if (!st_done) {
      const int fin = a.finished[b];
      const int next = amax_key_token(a.keys[b], fin, a.V, a.pad_id);
      if (st_t < a.max_len) a.tokens[(int64_t)a.tok_row[b] * a.max_len + st_t] = next;
      const int nf = fin | (next == a.eos_id);
      a.finished[b] = nf;
      if (!nf) atomicAdd(&a.state->n_unfinished, 1);
      a.keys[b] = 0ull;
    }
    if (b == 0) a.state->t_copy = st_t;      // ALWAYS (also in the no-op steps after the end): the lm_head kernel's "live" test reads it
  }
  M2M_STAMP(6 + (SELF ? 1 : 0), 3);
}

// ---- the same kernel for LARGE chains: C clips of one head per workgroup (round 6) ----
// At the reference's own chunk (128 three-second segments, S = 190: ref music2midi/model.py:115-135, config.yaml inference.batch_size) a
// chain launches 512 (clip, head) workgroups per attention kernel, one per CU at a time (16 waves x ~124 VGPRs), and each of them is the
// same latency chain as at 16 clips: the row its predecessor wrote (~2.4 us), then its head's projection weights from L2 - 192 KB
// (self) / 96 KB (cross) against a K/V stream of 131 KB (t = 512) / 49 KB: 1 024 workgroups pull more bytes from L2 as weights than
// from HBM as K/V, and the regime is throughput-bound (12 attention launches x 2 rounds of workgroups per chain and step).  Here a
// workgroup owns one head of C consecutive clips: ONE row round trip for all of them (wave pair c normalises clip c's row), the
// head's weights fetched ONCE into the same registers and applied to the C rows (self: q AND this step's k, v rows up front, so
// the 24 weight registers are free again when the streams start), one bias row in LDS, then the clips' K/V streams walked one
// after the other by all 16 waves as ONE continuous stream: a clip's rounds are padded to a multiple of the prefetch depth, and
// its last rounds re-request into the NEXT live clip, so MC_PF rounds are in flight through every merge - and the C output
// projections at the end.
// Per row the arithmetic is the first kernel's, operation for operation (same lane -> element maps, same reduction trees, same
// key -> lane-group partition, same rounding points): ids and logits are bit-identical whichever form a chain takes
// (tests/test_t5_gpu.py::test_multi_clip_attention_and_wide_ff_tiles_are_bit_identical).  Chosen per chain by its clip count.
// Prefetch depth (rounds in flight per workgroup).  Same-box A/B at 128 x S = 190, two chains of 64 clips, C = 2 (tools/r6_mc_ab.sh,
// us per decode step): 2 rounds 370.0, 3 rounds 379.5, 4 rounds 394.0, 6 rounds 457.5 - as in the small-chain regime (the first
// kernel's PF note) more bytes in flight per CU only lengthen every workgroup's latency-critical loads: the launches of a chain
// start their streams together, the memory system is saturated in those bursts, and its loaded latency is bytes in flight / rate.
#ifndef M2M_MC_PF_SELF
#define M2M_MC_PF_SELF 2
#endif
#ifndef M2M_MC_PF_CROSS
#define M2M_MC_PF_CROSS 2
#endif
// diagnostic builds only (-DM2M_MC_ABL=mask, never the product): 1 = no output-projection atomics, 2 = no K/V stream
#ifndef M2M_MC_ABL
#define M2M_MC_ABL 0
#endif
// Waves per SIMD the kernel is compiled for: 4 = the 128 registers of a 16-wave workgroup alone on its CU; 8 = 64 registers, TWO
// workgroups per CU.  Same-box A/B (tools/r6_mc_ab.sh, 128 x S = 190, us per step): two chains of 64 clips, C = 4: 364.2 / 364.9 at
// 4 waves against 409.4 / 411.4 at 8; C = 2: 379.2 / 382.5 against 384.8 / 387.5; only ONE chain of 128 clips with C = 2 (512
// workgroups per launch) prefers 8 (411-417 against 437-440).  Co-resident workgroups of two chains stream at the same time, and
// more streams in flight cost more than the latency phases they hide - the same finding as the prefetch depth.
#ifndef M2M_MC_WAVES
#define M2M_MC_WAVES 4
#endif
// CIF = clips in flight: register-slot sets, clip c in set c % CIF, a clip's last rounds handed to the next walking clip OF ITS SET.
// 1: one stream, clip after clip (the product).  2 (cross-attention, M2M_MC_CIF=2): measured slower, see launch_dec_attn_mc_t.
template <typename T, bool SELF, bool NT, bool FETCH, int C, int CIF = 1>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(M2M_MC_WAVES, M2M_MC_WAVES))) void dec_attn_mc_kernel(DecAttnArgs a, int nb) {
  static_assert(CIF == 1 || (CIF == 2 && !SELF && C % 2 == 0), "two clips in flight: cross-attention, an even clip count");
  static_assert(!FETCH || SELF, "only the layer-0 self-attention fetches its input row from the embedding table");
  static_assert(C >= 2 && C <= 8, "a wave pair normalises one clip's row: at most 8 clips per 16-wave workgroup");
  constexpr int E = 16 / sizeof(T);
  constexpr int LPR = DK / E;
  constexpr int KPW = 64 / LPR;
  constexpr int KPB = 16 * KPW;
  constexpr int NOUT = DK;
  constexpr int LPO = 16;
  constexpr int LPO2 = 8;
  constexpr int WMAX2 = 6;
  constexpr int PF = SELF ? M2M_MC_PF_SELF : M2M_MC_PF_CROSS;
  constexpr int WMAX = 3;
  constexpr int GPW = 64 / LPR;
  using V16 = decltype(Vec16<T>().v);
  extern __shared__ __align__(16) float hn[];   // [C][d] normalised input rows (already rounded to T), then (self) the bias row
  __shared__ float redw[16], redl[16], redlf[16];
  __shared__ float redo[16][DK];
  __shared__ __align__(16) float redg[16 * GPW][DK];
  __shared__ __align__(16) float qs[C][DK];
  __shared__ __align__(16) float kn[SELF ? C : 1][DK];
  __shared__ __align__(16) float vn[SELF ? C : 1][DK];
  __shared__ __align__(16) float oh[C][DK];
  __shared__ __align__(16) xq_t xrow[C][512];    // the fixed-point input rows (d_model <= 512): head 0 adds them into x_out at the end
  const int st_t = a.state->t;
  const int st_done = a.state->done | (st_t >= a.state->max_steps);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = blockIdx.x, b0 = blockIdx.y * C;      // linear id = 8 * block + h: a head's workgroups on one XCD, as in the first kernel
  // finished-row flags of the C clips (scalar loads, requested with the loop state) as a bit mask; clips past the chain's end are
  // clamped onto its last clip for every load and never stored
  unsigned finmask = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) finmask |= (a.fin_skip[(int64_t)min(b0 + c, nb - 1) * a.fin_stride] != 0 ? 1u : 0u) << c;
  const int sub = lane % LPR;
  const int kslot = wave * KPW + lane / LPR;
  const int64_t kv_clip = (int64_t)a.H * a.kv_stride * DK;                       // elements between two clips' blocks
  T* const Kb0 = reinterpret_cast<T*>(a.Kc) + ((int64_t)b0 * a.H + hh) * a.kv_stride * DK;
  T* const Vb0 = reinterpret_cast<T*>(a.Vc) + ((int64_t)b0 * a.H + hh) * a.kv_stride * DK;

  // ---- 0. requests for the prologue.  Wave pair cl = wave / 2 owns clip cl's row: wave 2 cl holds its elements 0..255, wave
  //         2 cl + 1 the rest - the lane -> element map and the reduction tree of the first kernel's waves 0 and 1 ----
  const int cl = wave >> 1, tin = (wave & 1) * 64 + lane;
  const int xc = min(tin * 4, a.d - 4);
  const int bx = min(b0 + min(cl, C - 1), nb - 1);
  const bool own_wave = cl < C && (wave & 1) * 256 < a.d;      // wave-uniform
  const bool own = cl < C && tin * 4 < a.d;
  const int on_ = min(tid >> 1, a.d - 1), opart = tid & 1;
  longlong2 xr0 = make_longlong2(0, 0), xr1 = make_longlong2(0, 0);
  unsigned long long kraw = 0;
  int fin0 = 0;
  if constexpr (FETCH) {
    kraw = a.keys[bx];
    fin0 = a.finished[bx];
  } else {
    xr0 = *reinterpret_cast<const longlong2*>(a.x + (int64_t)bx * a.d + xc);
    xr1 = *reinterpret_cast<const longlong2*>(a.x + (int64_t)bx * a.d + xc + 2);
  }
  const float4 gv = *reinterpret_cast<const float4*>(a.ln_w + xc);
  __builtin_amdgcn_sched_barrier(0);               // the latency-critical rows go out FIRST (loads retire in order)
  const int po = min(tid / LPO, NOUT - 1), part = tid % LPO;
  const int which = po / DK, dd = po - which * DK;
  const T* wrow = reinterpret_cast<const T*>(a.Wp) + ((int64_t)which * a.inner + hh * DK + dd) * a.d;
  const int cnt = a.d / E / LPO;
  Vec16<T> w[WMAX];
#pragma unroll
  for (int u = 0; u < WMAX; ++u) w[u].v = *reinterpret_cast<const V16*>(wrow + (min(u, cnt - 1) * LPO + part) * E);
  // self: the weights of the k,v rows this step appends (128 outputs x 8 lanes) are part of the prologue here
  const int o2 = tid >> 3, part2 = tid & 7;
  const int which2 = 1 + (o2 >> 6), dd2 = o2 & 63;
  const T* wrow2 = reinterpret_cast<const T*>(a.Wp) + ((int64_t)which2 * a.inner + hh * DK + dd2) * a.d;
  const int cnt2 = a.d / E / LPO2;
  float* const biasl = hn + C * a.d;                   // [kv_stride] (self only): ONE bias row serves the C clips
  constexpr int BPT = 2;
  float bv[SELF ? BPT : 1];
  if (SELF) {
#pragma unroll
    for (int u = 0; u < BPT; ++u) bv[u] = a.bias[(int64_t)hh * a.bias_stride + min(tid + 1024 * u, a.bias_stride - 1)];
  }
  const int t = SELF ? (a.self_len_override > 0 ? a.self_len_override - 1 : st_t) : 0;
  const int n_prev = SELF ? t : a.n_keys;
  const int last = max(n_prev - 1, 0);
  // a clip's rounds, padded to a multiple of the prefetch depth (the padding rounds re-read its last key row and visit nothing), so
  // that a clip always starts in register slot 0 and its last PF rounds can re-request into the next clip slot for slot
  const int rpc = ((n_prev + KPB - 1) / KPB + PF - 1) / PF * PF;
  // clips that walk keys at all: live rows of this chain with at least one cached key
  unsigned walk = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) walk |= ((n_prev > 0 && !((finmask >> c) & 1u) && b0 + c < nb) ? 1u : 0u) << c;
  if (M2M_MC_ABL & 2) walk = 0;

  // The stream goes through buffer loads: one descriptor per operand for this workgroup's C clips (wave-uniform: built from the
  // kernel arguments and blockIdx only), the clip as the scalar offset, a 32-bit byte offset per lane - one address register per
  // slot serves K and V (with flat 64-bit addresses the compiler kept a register PAIR per load and slot alive across the loop:
  // 2 x 2 x PF registers, which is what limited the depth of the window).
  const unsigned clip_bytes = (unsigned)(kv_clip * (int64_t)sizeof(T));
  const unsigned span_bytes = (unsigned)(min(C, nb - b0) - 1) * clip_bytes + (unsigned)(a.kv_stride * DK * (int)sizeof(T));
  const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(Kb0, 0, (int)span_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(Vb0, 0, (int)span_bytes, 0x00020000);
  auto kvload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(V16, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, NT ? 2 : 0));   // aux 2 = nt
  };
  // the first walking clip's first PF rounds, right behind the prologue's own loads (clamped addresses, never predicated; when
  // no clip walks this is clip 0's last row once more - dropped)
  constexpr unsigned SETBITS = CIF == 1 ? 0xFFFFFFFFu : 0x55555555u;      // the clips of set 0
  Vec16<T> kv[CIF][PF], vv[CIF][PF];
#pragma unroll
  for (int st = 0; st < CIF; ++st) {
    const unsigned mine = walk & (SETBITS << st);
    const int c_first = mine ? __builtin_ctz(mine) : st;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const unsigned off = (unsigned)((min(kslot + u * KPB, last) * DK + sub * E) * (int)sizeof(T));
      kv[st][u].v = kvload(rK, off, (unsigned)c_first * clip_bytes);
      vv[st][u].v = kvload(rV, off, (unsigned)c_first * clip_bytes);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- 1. RMSNorm of the C rows -> hn[c] (rounded to the GEMM-input type T) ----
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (FETCH) {
    const int tok = amax_key_token(kraw, fin0, a.V, a.pad_id);
    e4 = *reinterpret_cast<const float4*>(a.emb + (int64_t)tok * a.d + xc);
  }
  if (own_wave) {
    if constexpr (FETCH) xv = e4;
    else xv = make_float4(xq_flt(xr0.x), xq_flt(xr0.y), xq_flt(xr1.x), xq_flt(xr1.y));
    float ss = own ? (xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w) : 0.f;
    ss = wave_sum(ss);
    if (lane == 0) redw[wave] = ss;
  } else if (lane == 0) {
    redw[wave] = 0.f;
  }
  __syncthreads();
  if (own && hh == 0) {     // (hh: uniform) the raw row for head 0's residual carry; an embedding row is converted as the first kernel does
    if constexpr (FETCH) {
      if (!st_done) {
        xr0 = make_longlong2(xq_fix_guarded(xv.x, a.state), xq_fix_guarded(xv.y, a.state));
        xr1 = make_longlong2(xq_fix_guarded(xv.z, a.state), xq_fix_guarded(xv.w, a.state));
      }
    }
    *reinterpret_cast<longlong2*>(&xrow[cl][xc]) = xr0;
    *reinterpret_cast<longlong2*>(&xrow[cl][xc + 2]) = xr1;
  }
  if (own) {
    // the first kernel sums its 16 wave slots in order, 14 of them zeros: 0 + r0 + r1 (+ 0 ...) - the same two adds
    float tot = 0.f;
    tot += redw[2 * cl];
    tot += redw[2 * cl + 1];
    const float rs = rsqrtf(tot / (float)a.d + a.eps);
    float* const hc = hn + cl * a.d;
    put_in<T>(hc, xc + 0, gv.x * (xv.x * rs));
    put_in<T>(hc, xc + 1, gv.y * (xv.y * rs));
    put_in<T>(hc, xc + 2, gv.z * (xv.z * rs));
    put_in<T>(hc, xc + 3, gv.w * (xv.w * rs));
  }
  __syncthreads();

  // ---- 2. this head's projections of the C rows: the weights are in registers once ----
  {
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
    for (int u = 0; u < WMAX; ++u) {
      if (u < cnt) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = chunk_dot<T>(w[u], hn + c * a.d, u * LPO + part, acc[c]);
      }
    }
    for (int i = WMAX; i < cnt; i += 4) {   // d_model / dtype combinations beyond the register budget (fp32)
      Vec16<T> w2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w2[u].v = *reinterpret_cast<const V16*>(wrow + (min(i + u, cnt - 1) * LPO + part) * E);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u < cnt) {
#pragma unroll
          for (int c = 0; c < C; ++c) acc[c] = chunk_dot<T>(w2[u], hn + c * a.d, (i + u) * LPO + part, acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float q = group_sum<LPO>(acc[c]);
      if (part == 0 && tid < NOUT * LPO) qs[c][dd] = q;      // q stays fp32
    }
  }
  if (SELF) {
    // the k,v rows of this step: projected, rounded to T, appended to the cache and kept in LDS for the clips' own-key visits.
    // Their weights are requested only now, into the registers the q weights have left (one more L2 round trip in this workgroup's
    // chain, which the co-resident workgroup fills: the kernel is held to 64 registers so that TWO workgroups share a CU)
    Vec16<T> wkv[WMAX2];
#pragma unroll
    for (int u = 0; u < WMAX2; ++u) wkv[u].v = *reinterpret_cast<const V16*>(wrow2 + (min(u, cnt2 - 1) * LPO2 + part2) * E);
    float acc2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc2[c] = 0.f;
#pragma unroll
    for (int u = 0; u < WMAX2; ++u) {
      if (u < cnt2) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc2[c] = chunk_dot<T>(wkv[u], hn + c * a.d, u * LPO2 + part2, acc2[c]);
      }
    }
    for (int i = WMAX2; i < cnt2; i += 4) {
      Vec16<T> w2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w2[u].v = *reinterpret_cast<const V16*>(wrow2 + (min(i + u, cnt2 - 1) * LPO2 + part2) * E);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i + u < cnt2) {
#pragma unroll
          for (int c = 0; c < C; ++c) acc2[c] = chunk_dot<T>(w2[u], hn + c * a.d, (i + u) * LPO2 + part2, acc2[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float s2 = group_sum<LPO2>(acc2[c]);
      if (part2 == 0) {
        const T r = from_f32<T>(s2);                       // k, v are stored (and used) rounded to T
        const int64_t slot = (int64_t)c * kv_clip + (int64_t)t * DK + dd2;
        const bool st_ok = !st_done && b0 + c < nb;
        if (which2 == 1) { kn[c][dd2] = to_f32(r); if (st_ok) Kb0[slot] = r; }
        else             { vn[c][dd2] = to_f32(r); if (st_ok) Vb0[slot] = r; }
      }
    }
#pragma unroll
    for (int u = 0; u < BPT; ++u)
      if (tid + 1024 * u < a.bias_stride) biasl[tid + 1024 * u] = bv[u];
    for (int n = tid + 1024 * BPT; n < a.bias_stride; n += 1024) biasl[n] = a.bias[(int64_t)hh * a.bias_stride + n];
  }
  __syncthreads();

  // ---- 3. the clips' streams, one after the other ----
#pragma nounroll
  for (int c0 = 0; c0 < C; c0 += CIF) {
#pragma unroll
  for (int st = 0; st < CIF; ++st) {
    const int c = c0 + st;
    const bool walks = (walk >> c) & 1u;                                          // uniform
    const unsigned cbase = (unsigned)c * clip_bytes;      // this clip inside the descriptors (an invalid tail clip walks nothing)
    // per-clip OPAQUE copies of the lane's place in a round: derived from threadIdx the hand-over offsets below are loop invariants
    // that the compiler keeps from the prologue, spills (64 registers), and reloads in the middle of the stream - and a scratch
    // reload waits vmcnt(0), i.e. for every round in flight
    int kslot_c = kslot, sub_c = sub;
    asm volatile("" : "+v"(kslot_c), "+v"(sub_c));
    const int n_live = walks ? n_prev : 0;
    float qv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) qv[e] = qs[c][sub * E + e];
    float m_run = -1e30f, l_run = 0.f;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    auto visit = [&](float s, const float (&vrow)[E]) {
      const bool up = s > m_run;
      const float m_new = up ? s : m_run;
      const float ex = m2m_exp<T>(up ? m_run - s : s - m_run);
      const float alpha = up ? ex : 1.f;
      const float p = up ? 1.f : ex;
      l_run = fmaf(l_run, alpha, p);
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = fmaf(acc[e], alpha, p * vrow[e]);
      m_run = m_new;
    };
    // consume slot u (round k0 / KPB + u of this clip); then re-request into it: mode 1 = this clip's round PF ahead, mode 2 = round u
    // of the clip `nxt` elements further on (the next walking clip), mode 0 = nothing.  The mode is a compile-time constant per loop,
    // so every load is unconditional and counted (s_waitcnt vmcnt(N) keeps the other slots in flight).
    auto round = [&](const int k0, const int u, Vec16<T>& ks, Vec16<T>& vs, const int mode, const unsigned nxt) {
      const int key = k0 + kslot_c + u * KPB;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s = fmaf(qv[e], ks.get(e), s);
      s = group_sum<LPR>(s);
      if (key < n_live) {   // VALU-only predicate
        float vrow[E];
#pragma unroll
        for (int e = 0; e < E; ++e) vrow[e] = vs.get(e);
        if (SELF) s += biasl[t - key];
        visit(s, vrow);
      }
      // the slot is re-requested AFTER it has been consumed: with the request in front of the visit (the first kernel's order) the
      // compiler sinks V's unpacking into the predicated block, keeps the packed V alive past the request, loads into a SECOND
      // register set and copies it back behind a vmcnt(0) at the end of every iteration (seen in the ISA at PF = 4: the window
      // became batches)
      if (mode == 1) {
        const unsigned off = (unsigned)((min(key + PF * KPB, last) * DK + sub_c * E) * (int)sizeof(T));
        ks.v = kvload(rK, off, cbase);
        vs.v = kvload(rV, off, cbase);
      } else if (mode == 2) {
        // (nxt < 2^31: a clip's byte offset goes into the scalar offset; otherwise it is the out-of-range marker and goes into the
        // lane offset, the only part the range check looks at)
        const unsigned off = (unsigned)((min(kslot_c + u * KPB, last) * DK + sub_c * E) * (int)sizeof(T));
        const bool oob = nxt >= 0x7F000000u;                 // uniform
        ks.v = kvload(rK, oob ? nxt : off, oob ? 0u : nxt);
        vs.v = kvload(rV, oob ? nxt : off, oob ? 0u : nxt);
      }
    };
    if (walks) {
      const int kend = rpc * KPB;
      int k0 = 0;
      for (; k0 + PF * KPB < kend; k0 += PF * KPB) {           // every re-request lies inside this clip's (padded) rounds
#pragma unroll
        for (int u = 0; u < PF; ++u) round(k0, u, kv[st][u], vv[st][u], 1, 0);
      }
      // the clip's last PF rounds hand their slots to the next walking clip, so the stream runs through the merge below.  After the
      // LAST walking clip the same loads are issued with a lane offset beyond the descriptor's range: the hardware's range check
      // drops them (zeros, no memory access) - one code path, so the slots stay in ONE register set (with a second, load-free path
      // for the last clip the register allocator kept a second set of 8 PF registers for the handed-over rounds and copied it back)
      const unsigned rest = (walk >> (c + CIF)) & SETBITS;                     // later walking clips of this set
      const unsigned nxt = rest ? cbase + (unsigned)(__builtin_ctz(rest) + CIF) * clip_bytes : 0x7F000000u;
#pragma unroll
      for (int u = 0; u < PF; ++u) round(k0, u, kv[st][u], vv[st][u], 2, nxt);
    }
    if (SELF && wave == 0 && lane < LPR) {   // the key/value appended this step (relative position 0): group 0
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s = fmaf(qv[e], kn[c][sub * E + e], s);
      s = group_sum<LPR>(s);
      s += biasl[0];
      float vrow[E];
#pragma unroll
      for (int e = 0; e < E; ++e) vrow[e] = vn[c][sub * E + e];
      visit(s, vrow);
    }
    // ---- merge of the 128 key groups (as in the first kernel) ----
    {
      const float mw = wave_max(m_run);
      const float scale = m2m_exp<T>(m_run - mw);
      float lsum = (sub == 0) ? l_run * scale : 0.f;
      lsum = wave_sum(lsum);
      float* gp = &redg[wave * GPW + lane / LPR][sub * E];
#pragma unroll
      for (int e = 0; e < E; e += 4)
        *reinterpret_cast<float4*>(gp + e) = make_float4(acc[e] * scale, acc[e + 1] * scale, acc[e + 2] * scale, acc[e + 3] * scale);
      if (lane == 0) { redw[wave] = mw; redl[wave] = lsum; }
      __syncthreads();
      {
        float M = redw[lane & 15];
        M = fmaxf(M, lane_xor<8>(M)); M = fmaxf(M, lane_xor<4>(M)); M = fmaxf(M, lane_xor<2>(M)); M = fmaxf(M, lane_xor<1>(M));
        const float fw = m2m_exp<T>(redw[wave] - M);
        float sw = redg[wave * GPW][lane];
#pragma unroll
        for (int j = 1; j < GPW; ++j) sw += redg[wave * GPW + j][lane];
        redo[wave][lane] = sw * fw;
        if (lane == 0) redlf[wave] = redl[wave] * fw;
      }
      __syncthreads();
      if (tid < DK) {
        float s = 0.f, L = 0.f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) {
          s += redo[wv][tid];
          L += redlf[wv];
        }
        put_in<T>(oh[c], tid, L > 0.f ? s / L : 0.f);
      }
      __syncthreads();
    }
  }   // clips of a trip (sets)
  }

  // ---- 4. output projections of this head, accumulated into the C residual rows (the slice of wo is requested only here: 16 / 32
  //         registers that would otherwise be held through every stream) ----
  constexpr int OCH = 32 / E;
  const T* worow = reinterpret_cast<const T*>(a.Wo) + (int64_t)on_ * a.inner + hh * DK + opart * 32;
  Vec16<T> wo[OCH];
#pragma unroll
  for (int u = 0; u < OCH; ++u) wo[u].v = *reinterpret_cast<const V16*>(worow + u * E);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int bc = min(b0 + c, nb - 1);
    float accp = 0.f;
#pragma unroll
    for (int u = 0; u < OCH; ++u) accp = chunk_dot<T>(wo[u], oh[c], opart * OCH + u, accp);
    accp += lane_xor<1>(accp);
    if (opart == 0 && tid < 2 * a.d && !st_done && b0 + c < nb) {
      xq_t add = xq_fix_guarded(accp, a.state);
      if (hh == 0) add += xrow[c][on_];                 // head 0 also carries the residual itself (kept in LDS since the prologue)
      if (!(M2M_MC_ABL & 1) || hh == 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(a.x_out + (int64_t)bc * a.d + on_), (unsigned long long)add);
      if (hh == a.H - 1) a.x_zero[(int64_t)bc * a.d + on_] = 0;
    }
  }
  if (!SELF && a.book && hh == 0 && tid == 0) {
    // headless bookkeeping of the C rows (see the first kernel)
    for (int c = 0; c < C && b0 + c < nb; ++c) {
      const int b = b0 + c;
      if (!st_done) {
        const int fin = a.finished[b];
        const int next = amax_key_token(a.keys[b], fin, a.V, a.pad_id);
        if (st_t < a.max_len) a.tokens[(int64_t)a.tok_row[b] * a.max_len + st_t] = next;
        const int nf = fin | (next == a.eos_id);
        a.finished[b] = nf;
        if (!nf) atomicAdd(&a.state->n_unfinished, 1);
        a.keys[b] = 0ull;
      }
      if (b == 0) a.state->t_copy = st_t;
    }
  }
}

template <typename T>
static void launch_dec_attn_t(bool self, bool nt, const DecAttnArgs& a, dim3 grid, size_t smem, hipStream_t st) {
  if (self && a.emb) {
    if (nt) hipLaunchKernelGGL((dec_attn_kernel<T, true, true, true>), grid, dim3(1024), smem, st, a);
    else hipLaunchKernelGGL((dec_attn_kernel<T, true, false, true>), grid, dim3(1024), smem, st, a);
  } else if (self) {
    if (nt) hipLaunchKernelGGL((dec_attn_kernel<T, true, true>), grid, dim3(1024), smem, st, a);
    else hipLaunchKernelGGL((dec_attn_kernel<T, true, false>), grid, dim3(1024), smem, st, a);
  } else {
    if (nt) hipLaunchKernelGGL((dec_attn_kernel<T, false, true>), grid, dim3(1024), smem, st, a);
    else hipLaunchKernelGGL((dec_attn_kernel<T, false, false>), grid, dim3(1024), smem, st, a);
  }
}

// several clips per workgroup (dec_attn_mc_kernel): C = 2 or 4
template <typename T, int C>
static void launch_dec_attn_mc_t(bool self, bool nt, const DecAttnArgs& a, int nb, size_t smem, hipStream_t st) {
  const dim3 grid((unsigned)a.H, (unsigned)ceil_div(nb, C));
  // M2M_MC_CIF=2 (diagnostic): two clips in flight in the cross-attention.  At S = 190 a clip's whole stream is one round trip, so the
  // kernel is C dependent round trips long and two register-slot sets overlap them - on paper 14 -> ~10.5 us per launch; measured on
  // one box (tools/native_mc_sweep.py, us per step, one / two in flight): 2 x 64 clips bf16 340.6 / 338.9 against 351.4 / 350.9,
  // fp32 577.2 against 591.9, 2 x 32 clips 220.4 against 226.6, 2 x 64 at S = 864 480.9 against 485.0.  One in flight.
  static const int cif_env = [] { const char* v = getenv("M2M_MC_CIF"); return v ? atoi(v) : -1; }();
  const bool two_in_flight = !self && cif_env == 2;
  if (self && a.emb) {
    if (nt) hipLaunchKernelGGL((dec_attn_mc_kernel<T, true, true, true, C>), grid, dim3(1024), smem, st, a, nb);
    else hipLaunchKernelGGL((dec_attn_mc_kernel<T, true, false, true, C>), grid, dim3(1024), smem, st, a, nb);
  } else if (self) {
    if (nt) hipLaunchKernelGGL((dec_attn_mc_kernel<T, true, true, false, C>), grid, dim3(1024), smem, st, a, nb);
    else hipLaunchKernelGGL((dec_attn_mc_kernel<T, true, false, false, C>), grid, dim3(1024), smem, st, a, nb);
  } else if (two_in_flight) {     // short cross streams (a clip's keys fit the prefetch window): two clips in flight
    if (nt) hipLaunchKernelGGL((dec_attn_mc_kernel<T, false, true, false, C, 2>), grid, dim3(1024), smem, st, a, nb);
    else hipLaunchKernelGGL((dec_attn_mc_kernel<T, false, false, false, C, 2>), grid, dim3(1024), smem, st, a, nb);
  } else {
    if (nt) hipLaunchKernelGGL((dec_attn_mc_kernel<T, false, true, false, C>), grid, dim3(1024), smem, st, a, nb);
    else hipLaunchKernelGGL((dec_attn_mc_kernel<T, false, false, false, C>), grid, dim3(1024), smem, st, a, nb);
  }
}

static int launch_dec_attn(int precision, bool self, bool nt, DecAttnArgs a, int B, int clips, hipStream_t st) {
  const size_t smem = ((size_t)a.d * (size_t)clips + (self ? (size_t)a.bias_stride : 0)) * sizeof(float);   // hn rows + (self) the bias row
  M2M_REQUIRE(smem <= 24 * 1024, "decode attention: max_dec_len=%d too long for the LDS bias row (<= %d)", a.bias_stride,
              (24 * 1024 - a.d * 4 * clips) / 4);
  const bool bf = precision == M2M_PREC_BF16;
  if (clips == 1) {
    dim3 grid((unsigned)a.H, (unsigned)B);
    if (bf) launch_dec_attn_t<bf16_t>(self, nt, a, grid, smem, st);
    else launch_dec_attn_t<float>(self, nt, a, grid, smem, st);
  } else if (clips == 2) {
    if (bf) launch_dec_attn_mc_t<bf16_t, 2>(self, nt, a, B, smem, st);
    else launch_dec_attn_mc_t<float, 2>(self, nt, a, B, smem, st);
  } else if (clips == 4) {
    if (bf) launch_dec_attn_mc_t<bf16_t, 4>(self, nt, a, B, smem, st);
    else launch_dec_attn_mc_t<float, 4>(self, nt, a, B, smem, st);
  } else {
    set_error("decode attention: %d clips per workgroup not instantiated (1, 2, 4)", clips);
    return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// ================================================================== head ====
struct DecHeadArgs {
  const float* logits;     // [B, ldl]
  int ldl, V, B, d;
  const float* shared;     // [V, d] embedding
  xq_t* x;                 // [B, d] next-step input (fixed-point residual stream, buffer A)
  xq_t* x_zero;            // [B, d] buffer B: zeroed by the init kernel (later by every FFN down projection)
  int64_t* tokens;         // [session rows, max_len] generated ids (col 0 = start): the whole matrix, row of slot b = tok_row[b]
  int* tok_row;            // [B] (chain view) clip decoded in slot b; written by the init kernel (identity: row0 + b)
  int row0;                // first slot of this chain in the session
  int max_len;
  int* finished;           // [B]
  unsigned long long* keys;// [B] (headless greedy loop; null otherwise)
  DecState* state;
  int pad_id, eos_id;
  // teacher forcing (nullptr for greedy)
  const int64_t* forced;   // [B, Ld] decoder input ids
  int Ld;
  float* logits_out;       // [B, Ld, V]
};

// 32 lanes per clip: with B <= 32 every row is handled concurrently (one pass of independent loads,
// then one dependent embedding fetch), so the kernel is two memory round trips long.
__global__ __launch_bounds__(1024) void dec_head_kernel(DecHeadArgs a) {
  __shared__ int s_unfinished;
  DecState* stp = a.state;
  M2M_STAMP_DECL
  M2M_STAMP(8, 0);
  const int t = stp->t;
  const int st_done = stp->done | (t >= stp->max_steps);
  const int tid = threadIdx.x, l32 = tid & 31, grp = tid >> 5;
  // first batch of logits of this lane group's first row: requested before the loop state is consumed
  constexpr int HK = 16;   // V <= 32 * HK columns are covered by the register batch (vocab 400 -> 13 used)
  float lg0[HK];
  {
    const float* lg = a.logits + (int64_t)min(grp, a.B - 1) * a.ldl;
#pragma unroll
    for (int j = 0; j < HK; ++j) lg0[j] = lg[min(l32 + 32 * j, a.V - 1)];
  }
  // no early exit on st_done (a branch here makes the compiler hoist the loop-state load in front of the logits
  // requests: two serial round trips; the state was written by this kernel one step ago, so it comes from HBM):
  // a finished chain runs the rest of its graph with every store below suppressed
  __builtin_amdgcn_sched_barrier(0);
  const bool live = !st_done;
  if (tid == 0) s_unfinished = 0;
  __syncthreads();
  for (int b = grp; b < a.B; b += 32) {
    const float* lg = a.logits + (int64_t)b * a.ldl;
    const int fin = a.forced ? 0 : a.finished[b];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    bool bad = false;      // a non-finite logit (NaN weights ...): the reference would emit NaN logits, never a silent token
    if (b == grp) {
#pragma unroll
      for (int j = 0; j < HK; ++j) {
        const int v = l32 + 32 * j;
        const float x = lg0[j];
        if (v < a.V) bad |= !(fabsf(x) <= 3.0e38f);
        if (v < a.V && (x > best || (x == best && v < bi))) { best = x; bi = v; }
      }
    }
    for (int v = l32 + (b == grp ? 32 * HK : 0); v < a.V; v += 32) {
      const float x = lg[v];
      bad |= !(fabsf(x) <= 3.0e38f);
      if (x > best || (x == best && v < bi)) { best = x; bi = v; }
    }
    if (bad && live) stp->overflow = 1;
    auto take = [&](float ob, int oi) {
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    };
    // butterfly inside the 32-lane half (xor 16, 8, 4, 2, 1), LDS-free exchanges
    take(lane_xor<16>(best), __builtin_bit_cast(int, lane_xor<16>(__builtin_bit_cast(float, bi))));
    take(lane_xor<8>(best), __builtin_bit_cast(int, lane_xor<8>(__builtin_bit_cast(float, bi))));
    take(lane_xor<4>(best), __builtin_bit_cast(int, lane_xor<4>(__builtin_bit_cast(float, bi))));
    take(lane_xor<2>(best), __builtin_bit_cast(int, lane_xor<2>(__builtin_bit_cast(float, bi))));
    take(lane_xor<1>(best), __builtin_bit_cast(int, lane_xor<1>(__builtin_bit_cast(float, bi))));
    int next;
    if (a.forced) {
      if (a.logits_out && live)
        for (int v = l32; v < a.V; v += 32) a.logits_out[((int64_t)b * a.Ld + t) * a.V + v] = lg[v];
      next = (t + 1 < a.Ld) ? (int)a.forced[(int64_t)b * a.Ld + t + 1] : a.pad_id;
    } else {
      // hf generation/utils.py:2925-2937: argmax; finished rows emit pad; EOS finishes a row
      next = fin ? a.pad_id : (bi == 0x7fffffff ? 0 : bi);
      if (l32 == 0 && live) {
        if (t + 1 < a.max_len) a.tokens[(int64_t)a.tok_row[b] * a.max_len + t + 1] = next;
        const int nf = fin | (next == a.eos_id);
        a.finished[b] = nf;
        if (!nf) atomicAdd(&s_unfinished, 1);
      }
    }
    if (next < 0 || next >= a.V) next = a.pad_id;
    const float* emb = a.shared + (int64_t)next * a.d;
    for (int c = l32 * 4; c < a.d; c += 128) {
      const float4 e4 = *reinterpret_cast<const float4*>(emb + c);
      xq_t* xp = a.x + (int64_t)b * a.d + c;
      if (live) {
        *reinterpret_cast<longlong2*>(xp) = make_longlong2(xq_fix_guarded(e4.x, stp), xq_fix_guarded(e4.y, stp));
        *reinterpret_cast<longlong2*>(xp + 2) = make_longlong2(xq_fix_guarded(e4.z, stp), xq_fix_guarded(e4.w, stp));
      }
    }
  }
  __syncthreads();
  if (tid == 0 && live) {
    const int nt = t + 1;
    stp->t = nt;
    if (a.forced) {
      if (nt >= a.Ld) stp->done = 1;
    } else {
      stp->n_unfinished = s_unfinished;
      if (s_unfinished == 0 || nt >= stp->max_steps) { stp->done = 1; stp->out_len = nt + 1; }
    }
  }
  M2M_STAMP(8, 2);
}

__global__ void dec_init_kernel(DecHeadArgs a, int start_id, int max_steps) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nthreads = gridDim.x * blockDim.x;
  if (tid == 0) {
    a.state->t = 0; a.state->done = (max_steps <= 0) ? 1 : 0; a.state->out_len = 1;
    a.state->n_unfinished = a.keys ? 0 : a.B; a.state->max_steps = max_steps; a.state->t_copy = 0;
  }
  for (int b = tid; b < a.B; b += nthreads) {
    a.finished[b] = 0;
    a.tok_row[b] = a.row0 + b;
    if (a.keys) a.keys[b] = amax_key(0.f, start_id);         // headless: step 0 "decodes" the start token
  }
  if (!a.forced) {
    int64_t* tk = a.tokens + (int64_t)a.row0 * a.max_len;
    for (int i = tid; i < a.B * a.max_len; i += nthreads) tk[i] = (i % a.max_len == 0) ? start_id : a.pad_id;
  }
  for (int i = tid; i < a.B * a.d; i += nthreads) {
    const int b = i / a.d, c = i - b * a.d;
    int tok = a.forced ? (int)a.forced[(int64_t)b * a.Ld] : start_id;
    if (tok < 0 || tok >= a.V) tok = a.pad_id;
    a.x[i] = xq_fix_guarded(a.shared[(int64_t)tok * a.d + c], a.state);
    a.x_zero[i] = 0;
  }
}

// after the loop: the last lm_head's keys are still pending (no later step consumed them) unless the chain ended on EOS
__global__ void dec_final_kernel(DecHeadArgs a) {
  DecState* st = a.state;
  const int t = st->t;
  if (st->done || !a.keys) return;
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    const int fin = a.finished[b];
    const int next = amax_key_token(a.keys[b], fin, a.V, a.pad_id);
    if (t < a.max_len) a.tokens[(int64_t)a.tok_row[b] * a.max_len + t] = next;
    a.finished[b] = fin | (next == a.eos_id);
  }
  __syncthreads();
  if (threadIdx.x == 0) { st->done = 1; st->out_len = t + 1; }
}

// M2M_HEADLESS=0 keeps dec_head_kernel in the greedy loop (the round-1 step: 20 kernels); default: folded away (19 kernels)
bool decode_headless() {
  static const bool on = [] { const char* v = getenv("M2M_HEADLESS"); return !(v && v[0] == '0'); }();
  return on;
}

// ============================================================ live-row re-packing ====
// Clips end at different steps (a real checkpoint ends a 3 s segment after tens to hundreds of its 1 024 tokens).  The finished-row
// early-out above stops a finished row's K/V stream, but its workgroups still launch and the chain still pays its latency floor for
// every slot.  At a host poll the decode loop (t5_api.hip) therefore RE-PACKS the live rows into the first slots and relaunches
// smaller chains: every per-clip buffer has the clip index outermost, so a row is moved by copying its planes — self K/V up to the
// current position, cross K/V, the pending arg-max key, the residual row — from a slot behind the packed range into the slot of a
// finished row in front of it (source and destination sets are disjoint: one launch, no ordering).  Token rows do NOT move: slot ->
// clip is the tok_row table.  Nothing of this changes an id: a row's arithmetic never depended on its slot.
struct MoveArgs {
  int n;                        // moves in this launch (<= 64)
  short src[64], dst[64];
  unsigned char* self_k; unsigned char* self_v; unsigned char* cross_kv;
  int64_t self_row, self_layer;     // bytes between rows / layers of the self caches ([L][maxB][H][Lmax][64])
  int64_t self_head; int self_len_bytes;   // bytes between heads, bytes of the t valid keys of one head
  int H;
  int64_t cross_row, cross_plane;   // bytes of one row's [H][S][64] block, bytes between (layer, k|v) planes ([L][2][B][H][S][64])
  int L;
  unsigned long long* keys; int* finished; int* tok_row;
  xq_t* x0; int d;                  // residual buffer A
};

__global__ __launch_bounds__(256) void dec_move_rows_kernel(MoveArgs a) {
  const int mv = blockIdx.x, plane = blockIdx.y, part = blockIdx.z, nparts = gridDim.z;
  const int src = a.src[mv], dst = a.dst[mv];
  const int tid = threadIdx.x;
  if (plane < 2 * a.L) {                              // self K (even) / V (odd) of layer plane / 2: H pieces of self_len_bytes
    unsigned char* base = (plane & 1) ? a.self_v : a.self_k;
    const int64_t lo = (int64_t)(plane >> 1) * a.self_layer;
    const int per_head = a.self_len_bytes / 16;       // 16-byte units (64 elements * es is a multiple of 16)
    const int total = a.H * per_head;
    for (int i = part * 256 + tid; i < total; i += nparts * 256) {
      const int hh = i / per_head, u = i - hh * per_head;
      const int64_t off = lo + (int64_t)hh * a.self_head + (int64_t)u * 16;
      *reinterpret_cast<uint4*>(base + off + (int64_t)dst * a.self_row) = *reinterpret_cast<const uint4*>(base + off + (int64_t)src * a.self_row);
    }
  } else if (plane < 4 * a.L) {                       // cross K/V plane (layer, k|v): one contiguous block per row
    const int64_t lo = (int64_t)(plane - 2 * a.L) * a.cross_plane;
    const int64_t total = a.cross_row / 16;
    for (int64_t i = part * 256 + tid; i < total; i += nparts * 256)
      *reinterpret_cast<uint4*>(a.cross_kv + lo + (int64_t)dst * a.cross_row + i * 16) =
          *reinterpret_cast<const uint4*>(a.cross_kv + lo + (int64_t)src * a.cross_row + i * 16);
  } else if (part == 0) {                             // the row's small state
    for (int c = tid; c < a.d; c += 256) a.x0[(int64_t)dst * a.d + c] = a.x0[(int64_t)src * a.d + c];
    if (tid == 0) { a.keys[dst] = a.keys[src]; a.finished[dst] = a.finished[src]; a.tok_row[dst] = a.tok_row[src]; }
  }
}

// src[i] -> dst[i] for i < n (slots of the session; the sets are disjoint); t = keys already cached per row
int decode_move_rows(m2m_session* s, const int* src, const int* dst, int n, int t, hipStream_t st) {
  const m2m_model* m = s->m;
  const m2m_t5_geometry& g = m->g;
  const int64_t es = (int64_t)m->esize;
  for (int at = 0; at < n; at += 64) {
    MoveArgs a{};
    a.n = n - at < 64 ? n - at : 64;
    for (int i = 0; i < a.n; ++i) { a.src[i] = (short)src[at + i]; a.dst[i] = (short)dst[at + i]; }
    a.self_k = (unsigned char*)s->self_k; a.self_v = (unsigned char*)s->self_v; a.cross_kv = (unsigned char*)s->cross_kv;
    a.self_head = (int64_t)s->max_dec * DK * es; a.self_row = (int64_t)g.num_heads * a.self_head;
    a.self_layer = (int64_t)s->max_batch * a.self_row; a.self_len_bytes = (int)((int64_t)t * DK * es);
    a.H = g.num_heads; a.L = g.num_decoder_layers;
    a.cross_row = (int64_t)g.num_heads * s->S * DK * es; a.cross_plane = (int64_t)s->B * a.cross_row;
    a.keys = s->keys; a.finished = s->finished; a.tok_row = s->tok_row;
    a.x0 = reinterpret_cast<xq_t*>(s->x_dec); a.d = g.d_model;
    hipLaunchKernelGGL(dec_move_rows_kernel, dim3((unsigned)a.n, (unsigned)(4 * a.L + 1), 8), dim3(256), 0, st, a);
    M2M_CHECK_HIP(hipGetLastError());
  }
  return M2M_OK;
}

// ============================================================ step driver ====
static xq_t* xbuf(m2m_session* s, const DecView& v, int which);
// All per-clip buffers are [B][...] with the clip index outermost, so a view is a pointer offset.
static DecHeadArgs head_args(m2m_session* s, const DecView& v, bool forced, float* logits_out, int Ld) {
  const m2m_model* m = s->m;
  DecHeadArgs h{};
  h.logits = s->logits + (int64_t)v.b0 * m->vocab_pad; h.ldl = m->vocab_pad; h.V = m->g.vocab_size; h.B = v.nb;
  h.d = m->g.d_model; h.shared = m->shared; h.x = xbuf(s, v, 0); h.x_zero = xbuf(s, v, 1);
  h.tokens = s->tokens; h.tok_row = s->tok_row + v.b0; h.row0 = v.b0; h.max_len = s->max_dec;
  h.finished = s->finished + v.b0; h.state = v.state; h.pad_id = m->g.pad_token_id; h.eos_id = m->g.eos_token_id;
  h.keys = (!forced && decode_headless()) ? s->keys + v.b0 : nullptr;
  h.forced = forced ? s->forced_ids + (int64_t)v.b0 * Ld : nullptr; h.Ld = Ld;
  h.logits_out = logits_out ? logits_out + (int64_t)v.b0 * Ld * m->g.vocab_size : nullptr;
  return h;
}

int decode_finalize(m2m_session* s, const DecView& v, hipStream_t st) {
  if (!decode_headless()) return M2M_OK;
  DecHeadArgs h = head_args(s, v, false, nullptr, 0);
  hipLaunchKernelGGL(dec_final_kernel, dim3(1), dim3(256), 0, st, h);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

int decode_init(m2m_session* s, const DecView& v, int max_steps, bool forced, hipStream_t st) {
  DecHeadArgs h = head_args(s, v, forced, nullptr, forced ? max_steps : 0);
  // the sticky range flag is cleared here, not by the init kernel: that kernel's own embedding conversions may raise it
  M2M_CHECK_HIP(hipMemsetAsync(v.state, 0, sizeof(DecState), st));
  hipLaunchKernelGGL(dec_init_kernel, dim3(32), dim3(256), 0, st, h, s->m->g.decoder_start_token_id, max_steps);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

static size_t kv_layer_elems(const m2m_session* s, int len) {
  return (size_t)s->max_batch * s->m->g.num_heads * len * DK;
}

// rotation of the residual-stream buffers inside a layer: A (x_dec[0]) holds the layer input
static xq_t* xbuf(m2m_session* s, const DecView& v, int which) {
  const size_t stride = (size_t)ceil_div(s->max_batch, 32) * 32 * s->m->g.d_model;
  return reinterpret_cast<xq_t*>(s->x_dec) + (size_t)which * stride + (size_t)v.b0 * s->m->g.d_model;
}

// M2M_FINISHED_SKIP=0: finished rows keep streaming their K/V (the behaviour before round 4; bench.py's ragged_eos "before" leg)
bool decode_finished_skip_on() {
  const char* v = getenv("M2M_FINISHED_SKIP");      // read per launch (graphs bake it at capture: the bench re-creates the session)
  return !(v && v[0] == '0');
}

// Clips of one head per attention workgroup for a chain of nb clips: the multi-clip form (dec_attn_mc_kernel) from the chain sizes at
// which the launches stop being single latency chains and become rounds of workgroups (measured: tools/native_chain_sweep.py, DESIGN
// 4.3); s->attn_clips > 0 forces a value (M2M_DA_CLIPS, latched when the session is created; tests and A/B runs).
int decode_attn_clips(const m2m_session* s, int nb) {
  if (s->attn_clips > 0) return s->attn_clips;
  // same-box sweeps (tools/native_mc_sweep.py, us per step, C = 1 / 2 / 4): 2 x 16 clips, S = 864: 195 / 238 / -; 2 x 24: 264 / 266 / 363;
  // 2 x 32: 329 / 300 / 386; 2 x 64, S = 864: 595 / 522 / 512; 2 x 64, S = 190 (the reference's chunk): 446 / 383 / 362;
  // S = 190: 2 x 32: 256 / 226 / 289; 2 x 40: - / 276 / 299; 2 x 48: - / 316 / 313
  return nb >= 48 ? 4 : (nb >= 32 ? 2 : 1);
}

// residual rows per feed-forward workgroup for a chain of nb clips (M2M_DEC_FF_ROWS forces 8 or 16, latched per session).  16 rows
// measured no different from 8 at 2 x 64 clips (365.5 / 366.3 us per step): 8 everywhere.
int decode_ff_rows(const m2m_session* s, int nb) {
  (void)nb;
  return s->ff_rows > 0 ? s->ff_rows : 8;
}
// hidden slices per feed-forward workgroup (dec_ff_multi_kernel from 2; M2M_DEC_FF_SLICES forces 1, 2 or 4, latched per session)
int decode_ff_slices(const m2m_session* s, int nb) {
  if (s->ff_slices > 0) return s->ff_slices;
  // same-box sweeps (tools/native_mc_sweep.py, us per step, 1 / 2 / 4 slices): 2 x 16 clips 196 / 204 / -; 2 x 32: 301 / 292 / 312;
  // 2 x 48, S = 190: - / 313 / 326; 2 x 64 (the reference's chunk): 362 / 349-353 / 345-347
  return nb >= 56 ? 4 : (nb >= 32 ? 2 : 1);
}

int decode_launch_attn(m2m_session* s, const DecView& v, bool self, int layer, int self_len, hipStream_t st, bool headless, bool skip_finished) {
  const m2m_model* m = s->m;
  const m2m_t5_geometry& g = m->g;
  const size_t es = m->esize;
  const int H = g.num_heads;
  const DecLayerPacked& L = m->dec[layer];
  DecAttnArgs a{};
  // K/V working set of one decode step (all layers, cross + self at the session's maximum length)
  const double kv_step_bytes = (double)g.num_decoder_layers * 2.0 * s->B * m->inner * ((double)s->S + s->max_dec) * (double)es;
  bool nt = kv_step_bytes > 200e6;   // beyond what the 256 MB Infinity Cache can keep between steps
  if (nt && !self) {
    // ... but as many whole layers of cross K/V as fit a 180 MB budget keep the default policy and stay
    // cache-resident from step to step (they are re-read every step; the rest streams past them
    // non-temporally).  B = 32: 3 of 6 layers, 271.0 -> 266.6 ms per batch.
    static const int forced = [] { const char* v = getenv("M2M_KV_RESIDENT_LAYERS"); return v ? atoi(v) : -1; }();
    const double per_layer = 2.0 * s->B * m->inner * (double)s->S * (double)es;
    const int resident = forced >= 0 ? forced : (int)(180e6 / per_layer);
    if (layer < resident) nt = false;
  }
  int clips = decode_attn_clips(s, v.nb);
  if (self && s->attn_clips_self > 0) clips = s->attn_clips_self;      // diagnostic: another width for the self-attention launches
  // self: A -> B (zero C); cross: B -> C (zero A)
  a.x = xbuf(s, v, self ? 0 : 1); a.x_out = xbuf(s, v, self ? 1 : 2); a.x_zero = xbuf(s, v, self ? 2 : 0);
  a.eps = g.layer_norm_eps; a.d = g.d_model;
  a.H = H; a.inner = m->inner; a.state = v.state;
  if (skip_finished && decode_finished_skip_on()) { a.fin_skip = s->finished + v.b0; a.fin_stride = 1; }
  else { a.fin_skip = &v.state->zero; a.fin_stride = 0; }
  if (headless && layer == 0) {            // the greedy loop without the head kernel: layer 0 takes over its work
    a.keys = s->keys + v.b0; a.finished = s->finished + v.b0; a.tokens = s->tokens; a.tok_row = s->tok_row + v.b0;
    a.max_len = s->max_dec; a.V = g.vocab_size; a.pad_id = g.pad_token_id; a.eos_id = g.eos_token_id;
    if (self) a.emb = m->shared; else a.book = 1;
  }
  if (self) {
    const size_t off = ((size_t)layer * kv_layer_elems(s, s->max_dec) + (size_t)v.b0 * H * s->max_dec * DK) * es;
    a.ln_w = L.ln0; a.Wp = L.wqkv; a.Wo = L.wo;
    a.Kc = (unsigned char*)s->self_k + off;
    a.Vc = (unsigned char*)s->self_v + off;
    a.kv_stride = s->max_dec; a.n_keys = 0; a.self_len_override = self_len;
    a.bias = s->dec_bias_tab; a.bias_stride = s->max_dec;
    return launch_dec_attn(m->precision, true, nt, a, v.nb, clips, st);
  }
  // cross K/V: [L][2][B][H][S][64] with B, S = the encoded problem
  const size_t per = (size_t)s->B * H * s->S * DK;
  const size_t voff = (size_t)v.b0 * H * s->S * DK;
  a.ln_w = L.ln1; a.Wp = L.wcq; a.Wo = L.wco;
  a.Kc = (unsigned char*)s->cross_kv + (((size_t)layer * 2 + 0) * per + voff) * es;
  a.Vc = (unsigned char*)s->cross_kv + (((size_t)layer * 2 + 1) * per + voff) * es;
  a.kv_stride = s->S; a.n_keys = s->S; a.self_len_override = 0; a.bias = nullptr; a.bias_stride = 0;
  return launch_dec_attn(m->precision, false, nt, a, v.nb, clips, st);
}

int decode_launch_step(m2m_session* s, const DecView& v, bool forced, float* logits_out, int Ld, hipStream_t st) {
  const m2m_model* m = s->m;
  const m2m_t5_geometry& g = m->g;
  const int P = m->precision;
  xq_t* xA = xbuf(s, v, 0);
  xq_t* xB = xbuf(s, v, 1);
  xq_t* xC = xbuf(s, v, 2);
  int rc;
  const bool headless = !forced && decode_headless();
  for (int l = 0; l < g.num_decoder_layers; ++l) {
    const DecLayerPacked& L = m->dec[l];
    // 1. RMSNorm + per-head QKV projection + KV-cache append + causal self-attention + per-head
    //    output projection accumulated into the residual stream (one kernel)
    if ((rc = decode_launch_attn(s, v, true, l, 0, st, headless, !forced))) return rc;
    // 2. the same for cross-attention over the S encoder positions (query projection only)
    if ((rc = decode_launch_attn(s, v, false, l, 0, st, headless, !forced))) return rc;
    // 3. feed-forward sub-layer, one kernel: reads C (the stream after both attention sub-layers),
    //    accumulates C + FF(C) into A (zeroed by the cross-attention kernel) and leaves B zeroed for
    //    the next layer's self-attention
    DecFfArgs f{};
    f.x = xC; f.x_out = xA; f.x_zero = xB; f.ln_w = L.ln2; f.eps = g.layer_norm_eps;
    f.Wi = L.wi; f.Wo = L.wo_ff; f.d = g.d_model; f.d_ff = g.d_ff; f.B = v.nb; f.state = v.state;
    f.check_done = (headless && l == 0) ? 1 : 0;
    if ((rc = launch_dec_ff(P, f, decode_ff_rows(s, v.nb), decode_ff_slices(s, v.nb), st))) return rc;
  }
  // final RMSNorm + lm_head (untied, no d_model**-0.5 scaling: transformers 4.34 semantics)
  DecGemmArgs a{};
  a.eps = g.layer_norm_eps; a.B = v.nb; a.state = v.state;
  a.x = xA; a.ldx = g.d_model; a.ln_w = m->dec_final_ln; a.W = m->lm_head; a.K = g.d_model; a.N = g.vocab_size;
  a.out = s->logits + (int64_t)v.b0 * m->vocab_pad; a.ldo = m->vocab_pad;
  a.keys = headless ? s->keys + v.b0 : nullptr;
  if ((rc = launch_dec_gemm(P, a, st))) return rc;
  if (headless) return M2M_OK;             // the arg-max key is consumed by the next step's layer 0 (or by decode_finalize)
  DecHeadArgs h = head_args(s, v, forced, logits_out, Ld);
  hipLaunchKernelGGL(dec_head_kernel, dim3(1), dim3(1024), 0, st, h);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

}  // namespace m2m

#ifdef M2M_STAMPS
// diagnostic builds only: copy the stamp log to the host and reset it (not declared in the public header)
extern "C" int m2m_debug_read_stamps(unsigned long long* out_host, int max_n) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(m2m::g_stamp_n), sizeof(n)) != hipSuccess) return -1;
  if ((int)n > max_n) n = (unsigned)max_n;
  if (n > (1u << 18)) n = 1u << 18;
  if (n && hipMemcpyFromSymbol(out_host, HIP_SYMBOL(m2m::g_stamps), (size_t)n * 8) != hipSuccess) return -1;
  unsigned int z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(m2m::g_stamp_n), &z, sizeof(z));
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(m2m::g_stamps)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) << 18);
  return (int)n;
}
#endif
