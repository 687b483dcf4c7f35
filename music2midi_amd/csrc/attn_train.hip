// Whole-head attention of the training step (bf16), forward and backward, for the short sequences a training batch has
// (ref: music2midi/model.py:32-38 -> hf: modeling_t5.py:159-170 T5Attention: scores = Q K^T (no 1/sqrt(d)) + relative-position
// bias, softmax in fp32, dropout on the probabilities, context = P~ V; its backward is torch autograd's).
//
// Round 2 materialised the probabilities: per attention layer a K|V transpose launch, a "stripe" kernel per 32 queries writing P
// (and reading it back in the backward pass together with a re-emitted dropped copy and dS, 17.6 MB each at 16 clips), and a batched
// dV | dK product — 27-34 us forward and 47-53 us backward per layer, 1.4 ms of a 4.5 ms step, with matrix cores 4-6 % busy.
// Here ONE workgroup family per (clip, head) keeps the head's operands in LDS and never writes a probability:
//   forward   stage K and V (global_load_lds, swizzled), a wave per 32-query block walks the key blocks with an ONLINE softmax in
//             the S^T = K Q^T orientation (a lane owns one query: row statistics are in-lane + one lane^32 exchange), the dropped
//             probabilities go from the accumulator STRAIGHT into the next product as its B operand — O^T = V^T P~^T, V^T fragments
//             by transposing LDS reads (ds_read_b64_tr_b16) in the accumulator's own key order — and only O and the row
//             log-sum-exp leave the kernel;
//   backward  recomputes P = exp(S + bias - lse) twice, once per orientation, so that both reductions stay inside a wave:
//             pass A, a wave per 32-KEY block (S = Q K^T, a lane owns one key): dV^T += dO^T P~, dK^T += Q^T dS with P~ / dS from
//             the accumulators as B operands and dO^T / Q^T by transposing reads; pass B, a wave per 32-QUERY block (S^T again):
//             dQ^T += K^T dS^T.  delta = rowsum(dO o O) replaces sum_k P dP~ (the flash-attention identity, dropout included).
// The dropout masks are the step's counter-based hash with the SAME element index as before ((clip*H + head) * Sq + query) * ldp
// + key, so the oracle regenerates them; the relative-position-bias gradient leaves pass B as per-query-block diagonal sums in the
// layout the stripe kernel used (bias_stripes_sum_kernel / bias_bucket_kernel are unchanged).  fp32 (parity) mode, fp32 storage
// and sequences beyond AH_MAX_S keep the stripe path.
#include "mma.h"
#include "t5.h"
#include "train.h"

#include <algorithm>

namespace m2m {

constexpr int AH_MAX_S = 288;                // rows an LDS image holds (9 blocks of 32): 36 KB per [S, 64] bf16 operand, two images + tables per workgroup, two workgroups per CU

// LDS image of a [rows, 64] bf16 operand: 128-byte rows, no padding; 16-byte chunk c of row r sits at slot c ^ ah_g(r).  The
// natural fragment reads (lane = row, 16 lanes cover (r & 1, slot) = all 16 bank groups) and the transposing reads (four consecutive
// rows of an aligned group x 64 bytes: rows 0-1 land in one 64-byte half of their 128-byte bank half, rows 2-3 in the other) are
// both conflict-free with it.
__device__ inline int ah_g(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ inline int ah_off(int row, int chunk) { return row * 64 + ((chunk ^ ah_g(row)) << 3); }

typedef short ah_v4s __attribute__((ext_vector_type(4)));
// Fragment of X^T for one 32x32x16 step: rows = 32 columns [32 db, +32) of the LDS image X, k = the 16 image rows
// {p0 + 4h + (0..3)} u {p0 + 8 + 4h + (0..3)} (h = lane >> 5) — the order in which an accumulator lane holds its rows, so that a
// probability tile feeds the next product from the registers it was computed in.
__device__ inline Frag<bf16_t> ah_tr_frag(const bf16_t* X, int p0, int db, int lane) {
  typedef ah_v4s __attribute__((address_space(3))) * lds_v4s;
  const int li = lane & 15, q = li >> 2, pp = li & 3, gq = lane >> 4, h = lane >> 5;
  const int c = 4 * db + 2 * (gq & 1) + (pp >> 1), sub = (pp & 1) * 4;
  const int r0 = p0 + 4 * h + q, r1 = r0 + 8;
  const ah_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(X + ah_off(r0, c) + sub));
  const ah_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(X + ah_off(r1, c) + sub));
  Frag<bf16_t> f;
  f.v = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
  return f;
}
// natural fragment: row `row` of the image, k = d in [16 s + 8 h, +8)
__device__ inline Frag<bf16_t> ah_nat_frag(const bf16_t* X, int row, int s, int h) { return load_frag(X + ah_off(row, 2 * s + h)); }

// one 16-byte global -> LDS copy per lane (LDS address = wave-uniform base + 16 * lane); inline asm: see gemm_kernel's note on why
__device__ inline void ah_glds16(const bf16_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// stage rows [0, rows_p) of a (pos, d) operand into the LDS image X (rows >= n_valid repeat the last valid row: finite, and masked)
__device__ inline void ah_stage(bf16_t* X, const bf16_t* G, int64_t ld, int n_valid, int rows_p, int wave, int lane, int n_waves) {
  typedef __attribute__((address_space(3))) bf16_t* lds_ptr_t;
  const unsigned base = (unsigned)(uintptr_t)(lds_ptr_t)X;
  const int lrow = lane >> 3, slot = lane & 7;
  for (int g8 = wave; g8 * 8 < rows_p; g8 += n_waves) {                      // 8 rows (1 KiB) per wave instruction
    const int row = g8 * 8 + lrow;
    const int c = slot ^ ah_g(row);
    ah_glds16(G + (int64_t)min(row, n_valid - 1) * ld + c * 8, (unsigned)__builtin_amdgcn_readfirstlane((int)(base + (unsigned)g8 * 1024u)));
  }
}

struct HeadAttnArgs {
  // (position, d) of clip b, head h at ptr + b * sXb + h * 64 + position * ldx
  const bf16_t *Q, *K, *V;
  int64_t ldq, ldk, ldv, sQb, sKb, sVb;
  bf16_t* O;                   // forward out, backward in
  int64_t ldo, sOb;
  float* lse;                  // [B*H][Sq]: forward out, backward in
  const bf16_t* dO;            // backward in (layout of O)
  bf16_t *dQ, *dK, *dV;        // backward out
  int64_t lddq, lddk, lddv, sdQb, sdKb, sdVb;
  const float* bias_tab;       // [H][tab_stride] by (key - query + tab_center), or null
  int tab_stride, tab_center;
  float* diag_part;            // backward, self-attention with bias: [B*H][ceil(Sq/32)][Sk + 31] diagonal sums of dS, or null
  int H, Sq, Sk, causal, ldp;  // ldp: row pitch of the dropout element index (round-up-8 of Sk, as the stored P had)
  DropKey dk;
  uint32_t thresh;
  float scale;
};

__device__ inline float ah_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ------------------------------------------------------------------------------------------------------------ forward
template <bool CAUSAL, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_head_fwd_kernel(HeadAttnArgs a) {
  extern __shared__ __align__(1024) unsigned char ah_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hh = bh - b * a.H;
  const int nk = (a.Sk + 31) >> 5, Skp = nk * 32;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(ah_smem);
  bf16_t* Vs = Ks + Skp * 64;
  float* bias_s = reinterpret_cast<float*>(Vs + Skp * 64);
  const bf16_t* Kg = a.K + b * a.sKb + hh * 64;
  const bf16_t* Vg = a.V + b * a.sVb + hh * 64;
  ah_stage(Ks, Kg, a.ldk, a.Sk, Skp, wave, lane, 4);
  ah_stage(Vs, Vg, a.ldv, a.Sk, Skp, wave, lane, 4);
  if (BIAS)
    for (int i = threadIdx.x; i < a.tab_stride + 32; i += 256) bias_s[i] = i < a.tab_stride ? a.bias_tab[(int64_t)hh * a.tab_stride + i] : 0.f;
  // this wave's query block
  const int qi = 4 * blockIdx.x + wave;
  const bool active = qi * 32 < a.Sq;
  const int q = qi * 32 + r, qc = min(q, a.Sq - 1);
  Frag<bf16_t> qf[4];
  {
    const bf16_t* qrow = a.Q + b * a.sQb + hh * 64 + (int64_t)qc * a.ldq + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = load_frag(qrow + 16 * s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 oacc[2] = {zero_acc(), zero_acc()};
  float m = -1e30f, l = 0.f;
  if (active) {
    const uint64_t key = DROP ? drop_site_key(a.dk) : 0ull;
    const int64_t prow = ((int64_t)bh * a.Sq + qc) * a.ldp;
    const float* bt = bias_s + a.tab_center - qc;
    const int jend = CAUSAL ? min(nk, qi + 1) : nk;          // causal: tiles past the query block's diagonal hold no key <= q
    for (int j = 0; j < jend; ++j) {
      f32x16 acc = zero_acc();
#pragma unroll
      for (int s = 0; s < 4; ++s) mma16(acc, ah_nat_frag(Ks, 32 * j + r, s, h), qf[s]);      // S^T: rows = keys, cols = queries
      float tm = -1e30f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * h;
        float v = acc[e];
        if (BIAS) v += bt[k];
        v = (k < a.Sk && (!CAUSAL || k <= qc)) ? v : -1e30f;
        acc[e] = v;
        tm = fmaxf(tm, v);
      }
      tm = fmaxf(tm, lane_xor<32>(tm));
      const float mn = fmaxf(m, tm);
      const float alpha = ah_exp(m - mn);
      float ts = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[e] = ah_exp(acc[e] - mn); ts += acc[e]; }      // masked entries: exp(-1e30 - mn) = 0
      ts += lane_xor<32>(ts);
      l = l * alpha + ts;
      m = mn;
#pragma unroll
      for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
      if (DROP) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t kb = drop_keep4(key, prow + 32 * j + 8 * g + 4 * h, a.thresh);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * g + e] = ((kb >> e) & 1u) ? acc[4 * g + e] * a.scale : 0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float pv[8] = {acc[8 * s], acc[8 * s + 1], acc[8 * s + 2], acc[8 * s + 3], acc[8 * s + 4], acc[8 * s + 5], acc[8 * s + 6], acc[8 * s + 7]};
        const Frag<bf16_t> pf = pack_frag<bf16_t>(pv);
#pragma unroll
        for (int db = 0; db < 2; ++db) mma16(oacc[db], ah_tr_frag(Vs, 32 * j + 16 * s, db, lane), pf);      // O^T += V^T P~^T
      }
    }
  }
  __syncthreads();                                                   // every wave is done with K / V: their LDS becomes the output staging
  if (!active) return;
  const float inv = 1.0f / l;
  constexpr int OP = 72;                                             // staging pitch (elements)
  bf16_t* Os = Ks + wave * 32 * OP;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2 pk = make_uint2(pack2_bf16(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv), pack2_bf16(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv));
      *reinterpret_cast<uint2*>(Os + r * OP + 32 * db + 8 * g + 4 * h) = pk;
    }
  if (h == 0 && q < a.Sq) a.lse[(int64_t)bh * a.Sq + q] = m + logf(l);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bf16_t* Og = a.O + b * a.sOb + hh * 64;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = lane + 64 * u, rl = idx >> 3, ch = idx & 7;
    if (qi * 32 + rl < a.Sq)
      *reinterpret_cast<uint4*>(Og + (int64_t)(qi * 32 + rl) * a.ldo + ch * 8) = *reinterpret_cast<const uint4*>(Os + rl * OP + ch * 8);
  }
}

size_t attn_head_fwd_smem(int Sk, bool bias, int tab_stride) {
  const size_t skp = (size_t)((Sk + 31) / 32) * 32;
  const size_t kv = 2 * skp * 64 * sizeof(bf16_t), bias_b = bias ? ((size_t)(tab_stride + 32) * 4 + 15) / 16 * 16 : 0;
  const size_t out_stage = (size_t)4 * 32 * 72 * sizeof(bf16_t);          // the four waves' output tiles re-use the K / V (/ bias) bytes
  return kv + bias_b > out_stage ? kv + bias_b : out_stage;
}

int launch_attn_head_fwd(const HeadAttnArgs& a, int nB, hipStream_t st) {
  M2M_REQUIRE(a.Sq >= 1 && a.Sk >= 1 && a.Sq <= AH_MAX_S && a.Sk <= AH_MAX_S, "attn_head: sequence lengths (%d, %d) beyond %d", a.Sq, a.Sk, AH_MAX_S);
  M2M_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.sQb % 8 == 0 && a.sKb % 8 == 0 && a.sVb % 8 == 0 && a.sOb % 8 == 0,
              "attn_head: operand strides must keep 16-byte alignment");
  const bool bias = a.bias_tab != nullptr, drop = a.thresh != 0;
  const size_t smem = attn_head_fwd_smem(a.Sk, bias, a.tab_stride);
  M2M_REQUIRE(smem <= 78 * 1024, "attn_head: %zu bytes of LDS", smem);
  dim3 grid((unsigned)ceil_div(ceil_div(a.Sq, 32), 4), (unsigned)(nB * a.H));
#define M2M_AH_FWD(C_, B_, D_)                                                                   \
  do {                                                                                           \
    M2M_OPT_IN_LDS((attn_head_fwd_kernel<C_, B_, D_>), 158 * 1024);                              \
    hipLaunchKernelGGL((attn_head_fwd_kernel<C_, B_, D_>), grid, dim3(256), smem, st, a);        \
  } while (0)
  if (a.causal) {
    if (bias) { if (drop) M2M_AH_FWD(true, true, true); else M2M_AH_FWD(true, true, false); }
    else { if (drop) M2M_AH_FWD(true, false, true); else M2M_AH_FWD(true, false, false); }
  } else {
    if (bias) { if (drop) M2M_AH_FWD(false, true, true); else M2M_AH_FWD(false, true, false); }
    else { if (drop) M2M_AH_FWD(false, false, true); else M2M_AH_FWD(false, false, false); }
  }
#undef M2M_AH_FWD
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}


// ------------------------------------------------------------------------------------------------------------ backward
// a 32 x 64 tile held as acc[db][e] (row = lane & 31, d = 32 db + (e & 3) + 8 (e >> 2) + 4 h) -> LDS tile [32][AH_TP] bf16
constexpr int AH_TP = 72;
__device__ inline void ah_tile_to_lds(bf16_t* Ts, const f32x16 (&acc)[2], int r, int h, float mul) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<uint2*>(Ts + r * AH_TP + 32 * db + 8 * g + 4 * h) =
          make_uint2(pack2_bf16(acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul), pack2_bf16(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul));
}
// the wave's staged tile -> rows [row0, row0 + 32) n (< n_rows) of a (pos, d) operand, 16 bytes per lane and instruction (8 lanes = one 128-byte row)
__device__ inline void ah_tile_to_global(const bf16_t* Ts, bf16_t* G, int64_t ld, int row0, int n_rows, int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = lane + 64 * u, rl = idx >> 3, ch = idx & 7;
    if (row0 + rl < n_rows) *reinterpret_cast<uint4*>(G + (int64_t)(row0 + rl) * ld + ch * 8) = *reinterpret_cast<const uint4*>(Ts + rl * AH_TP + ch * 8);
  }
}
template <int X> __device__ inline uint32_t ah_quad_bcast(uint32_t v) {      // value of lane X of this lane's quad
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, X * 0x55, 0xF, 0xF, true);
}

template <bool CAUSAL, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_head_bwd_kernel(HeadAttnArgs a) {
  extern __shared__ __align__(1024) unsigned char ah_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hh = bh - b * a.H;
  const int nq = (a.Sq + 31) >> 5, nk = (a.Sk + 31) >> 5, Sqp = nq * 32, Skp = nk * 32;
  const int Sp = max(max(Sqp, Skp), 288);                       // (the output staging needs 36 KB of the second image whatever the lengths)
  bf16_t* Xa = reinterpret_cast<bf16_t*>(ah_smem);             // pass A: Q image; pass B: K image
  bf16_t* Xb = Xa + Sp * 64;                                   // pass A: dO image; then per-wave staging / scratch
  float* lse_s = reinterpret_cast<float*>(Xb + Sp * 64);       // [Sqp]: +1e30 beyond Sq (=> P = 0 there)
  float* delta_s = lse_s + Sqp;                                // [Sqp]
  float* bias_s = delta_s + Sqp;                               // [tab_stride + 32]
  const bf16_t* Qg = a.Q + b * a.sQb + hh * 64;
  const bf16_t* Kg = a.K + b * a.sKb + hh * 64;
  const bf16_t* Vg = a.V + b * a.sVb + hh * 64;
  const bf16_t* Og = a.O + b * a.sOb + hh * 64;
  const bf16_t* dOg = a.dO + b * a.sOb + hh * 64;
  const int blk = 4 * blockIdx.x + wave;                       // this wave's key block (pass A) and query block (pass B)
  const uint64_t key = DROP ? drop_site_key(a.dk) : 0ull;

  // ---- phase 0: Q, dO -> LDS; delta = rowsum(dO o O), lse, bias row; this wave's K / V block -> registers
  ah_stage(Xa, Qg, a.ldq, a.Sq, Sqp, wave, lane, 4);
  ah_stage(Xb, dOg, a.ldo, a.Sq, Sqp, wave, lane, 4);
  for (int q = threadIdx.x; q < Sqp; q += 256) {
    float dsum = 0.f, lv = 1e30f;
    if (q < a.Sq) {
      const uint4* pd = reinterpret_cast<const uint4*>(dOg + (int64_t)q * a.ldo);
      const uint4* po = reinterpret_cast<const uint4*>(Og + (int64_t)q * a.ldo);
      uint4 vd[8], vo[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { vd[c] = pd[c]; vo[c] = po[c]; }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t wd[4] = {vd[c].x, vd[c].y, vd[c].z, vd[c].w}, wo[4] = {vo[c].x, vo[c].y, vo[c].z, vo[c].w};
#pragma unroll
        for (int w = 0; w < 4; ++w)
          dsum += __uint_as_float(wd[w] << 16) * __uint_as_float(wo[w] << 16) + __uint_as_float(wd[w] & 0xFFFF0000u) * __uint_as_float(wo[w] & 0xFFFF0000u);
      }
      lv = a.lse[(int64_t)bh * a.Sq + q];
    }
    delta_s[q] = dsum;
    lse_s[q] = lv;
  }
  if (BIAS)
    for (int i = threadIdx.x; i < a.tab_stride + 32; i += 256) bias_s[i] = i < a.tab_stride ? a.bias_tab[(int64_t)hh * a.tab_stride + i] : 0.f;
  const bool actA = blk * 32 < a.Sk;
  Frag<bf16_t> kf[4], vf[4];
  {
    const int krow = min(blk * 32 + r, a.Sk - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = load_frag(Kg + (int64_t)krow * a.ldk + 16 * s + 8 * h);
      vf[s] = load_frag(Vg + (int64_t)krow * a.ldv + 16 * s + 8 * h);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- pass A: this wave's 32 keys against every query block.  S = Q_i K_j^T: accumulator rows = queries, a lane owns ONE key.
  f32x16 dvT[2] = {zero_acc(), zero_acc()}, dkT[2] = {zero_acc(), zero_acc()};
  if (actA) {
    const int kcol = blk * 32 + r;                              // this lane's key
    const bool kvalid = kcol < a.Sk;
    const int t4 = lane & 3;
    for (int i = CAUSAL ? blk : 0; i < nq; ++i) {
      f32x16 sacc = zero_acc(), pacc = zero_acc();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma16(sacc, ah_nat_frag(Xa, 32 * i + r, s, h), kf[s]);
        mma16(pacc, ah_nat_frag(Xb, 32 * i + r, s, h), vf[s]);
      }
      // dropout keep bits: the four keys 4c .. 4c+3 of a query share one hash, i.e. the four lanes of a quad; lane t of the quad hashes
      // for the accumulator rows e = 4 g + t and the quad reads each other's 4-bit masks through DPP
      uint32_t kown[4] = {0xFu, 0xFu, 0xFu, 0xFu};
      if (DROP) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qe = min(32 * i + t4 + 8 * g + 4 * h, a.Sq - 1);
          kown[g] = drop_keep4(key, ((int64_t)bh * a.Sq + qe) * a.ldp + (kcol & ~3), a.thresh);
        }
      }
      float ptv[16], dsv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 32 * i + 8 * g + 4 * h);
        const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 32 * i + 8 * g + 4 * h);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
        const uint32_t kq[4] = {ah_quad_bcast<0>(kown[g]), ah_quad_bcast<1>(kown[g]), ah_quad_bcast<2>(kown[g]), ah_quad_bcast<3>(kown[g])};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = 4 * g + u, q = 32 * i + u + 8 * g + 4 * h;
          float v = sacc[e];
          if (BIAS) v += bias_s[kcol - q + a.tab_center];        // (index stays inside the padded row: |kcol - q| < tab_stride for the rows that count)
          const bool valid = kvalid && (!CAUSAL || kcol <= q);
          const float p = valid ? ah_exp(v - lq[u]) : 0.f;
          const bool keep = !DROP || ((kq[u] >> t4) & 1u);
          ptv[e] = keep ? p * a.scale : 0.f;
          const float dpt = keep ? pacc[e] * a.scale : 0.f;
          dsv[e] = p * (dpt - dq[u]);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float pv[8] = {ptv[8 * s], ptv[8 * s + 1], ptv[8 * s + 2], ptv[8 * s + 3], ptv[8 * s + 4], ptv[8 * s + 5], ptv[8 * s + 6], ptv[8 * s + 7]};
        const float dv[8] = {dsv[8 * s], dsv[8 * s + 1], dsv[8 * s + 2], dsv[8 * s + 3], dsv[8 * s + 4], dsv[8 * s + 5], dsv[8 * s + 6], dsv[8 * s + 7]};
        const Frag<bf16_t> pf = pack_frag<bf16_t>(pv), df = pack_frag<bf16_t>(dv);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          mma16(dvT[db], ah_tr_frag(Xb, 32 * i + 16 * s, db, lane), pf);      // dV^T += dO^T P~
          mma16(dkT[db], ah_tr_frag(Xa, 32 * i + 16 * s, db, lane), df);      // dK^T += Q^T dS
        }
      }
    }
  }
  __syncthreads();                                               // every wave is done with the Q / dO images

  // ---- between the passes: K -> LDS (pass B reads it both ways); dV / dK tiles out through the second image; Q_i, dO_i -> registers
  ah_stage(Xa, Kg, a.ldk, a.Sk, Skp, wave, lane, 4);
  bf16_t* Tw = Xb + wave * (2 * 32 * AH_TP);                     // this wave's 9 KB of the second image
  if (actA) {
    ah_tile_to_lds(Tw, dvT, r, h, 1.0f);
    ah_tile_to_lds(Tw + 32 * AH_TP, dkT, r, h, 1.0f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ah_tile_to_global(Tw, a.dV + b * a.sdVb + hh * 64, a.lddv, blk * 32, a.Sk, lane);
    ah_tile_to_global(Tw + 32 * AH_TP, a.dK + b * a.sdKb + hh * 64, a.lddk, blk * 32, a.Sk, lane);
  }
  const bool actB = blk * 32 < a.Sq;
  const int q = blk * 32 + r, qc = min(q, a.Sq - 1);
  Frag<bf16_t> qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = load_frag(Qg + (int64_t)qc * a.ldq + 16 * s + 8 * h);
    dof[s] = load_frag(dOg + (int64_t)qc * a.ldo + 16 * s + 8 * h);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                               // K image complete; the tile stores above have read their LDS

  // ---- pass B: this wave's 32 queries against every key block.  S^T = K_j Q_i^T: a lane owns ONE query.
  if (!actB) return;
  f32x16 dqT[2] = {zero_acc(), zero_acc()};
  const float lse_q = q < a.Sq ? lse_s[q] : 1e30f, delta_q = delta_s[min(q, Sqp - 1)];
  float* tile_s = reinterpret_cast<float*>(Tw);                  // [32][33] fp32: the dS tile for the diagonal sums (bias gradient)
  float* diag_s = tile_s + 32 * 33;                              // [Sk + 31 (+ padding)]
  const int dl = a.Sk + 31;
  const bool want_diag = a.diag_part != nullptr;
  if (want_diag) {
    for (int x = lane; x < dl + 33; x += 64) diag_s[x] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const int64_t prow = ((int64_t)bh * a.Sq + qc) * a.ldp;
  const float* bt = bias_s + a.tab_center - qc;
  const int jend = CAUSAL ? min(nk, blk + 1) : nk;
  for (int j = 0; j < jend; ++j) {
    f32x16 sacc = zero_acc(), pacc = zero_acc();
    {
      const int vrow = min(32 * j + r, a.Sk - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma16(sacc, ah_nat_frag(Xa, 32 * j + r, s, h), qf[s]);
        mma16(pacc, load_frag(Vg + (int64_t)vrow * a.ldv + 16 * s + 8 * h), dof[s]);
      }
    }
    float dsv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k0 = 32 * j + 8 * g + 4 * h;
      const uint32_t kb = DROP ? drop_keep4(key, prow + k0, a.thresh) : 0xFu;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 4 * g + u, k = k0 + u;
        float v = sacc[e];
        if (BIAS) v += bt[k];
        const bool valid = k < a.Sk && (!CAUSAL || k <= qc);
        const float p = valid ? ah_exp(v - lse_q) : 0.f;
        const float dpt = (!DROP || ((kb >> u) & 1u)) ? pacc[e] * a.scale : 0.f;
        dsv[e] = p * (dpt - delta_q);
      }
    }
    Frag<bf16_t> df[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float dv[8] = {dsv[8 * s], dsv[8 * s + 1], dsv[8 * s + 2], dsv[8 * s + 3], dsv[8 * s + 4], dsv[8 * s + 5], dsv[8 * s + 6], dsv[8 * s + 7]};
      df[s] = pack_frag<bf16_t>(dv);
#pragma unroll
      for (int db = 0; db < 2; ++db) mma16(dqT[db], ah_tr_frag(Xa, 32 * j + 16 * s, db, lane), df[s]);      // dQ^T += K^T dS^T
    }
    if (want_diag) {
      // relative-position-bias gradient, stage 1: sums of this query block's dS along the diagonals key - local row = x - 31 (the layout
      // the stripe kernel left for bias_stripes_sum_kernel).  The tile goes to LDS as [row][key] in the bf16 values the product used, lane
      // x' sums diagonal x' - 31 of the tile in row order and adds it to entry 32 j + x' of the wave's running row.
#pragma unroll
      for (int e = 0; e < 16; ++e) tile_s[r * 33 + (e & 3) + 8 * (e >> 2) + 4 * h] = to_f32(from_f32<bf16_t>(dsv[e]));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < 63) {
        const int dd = lane - 31;
        float acc = 0.f;
        for (int rr = max(0, -dd); rr <= min(31, 31 - dd); ++rr) acc += tile_s[rr * 33 + rr + dd];
        diag_s[32 * j + lane] += acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (want_diag) {
    float* out = a.diag_part + ((int64_t)bh * nq + blk) * dl;
    for (int x = lane; x < dl; x += 64) out[x] = diag_s[x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  ah_tile_to_lds(Tw, dqT, r, h, 1.0f);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  ah_tile_to_global(Tw, a.dQ + b * a.sdQb + hh * 64, a.lddq, blk * 32, a.Sq, lane);
}

size_t attn_head_bwd_smem(int Sq, int Sk, bool bias, int tab_stride) {
  const size_t sqp = (size_t)((Sq + 31) / 32) * 32, skp = (size_t)((Sk + 31) / 32) * 32;
  const size_t sp = std::max<size_t>(std::max(sqp, skp), 288);
  return 2 * sp * 64 * sizeof(bf16_t) + 2 * sqp * 4 + (bias ? ((size_t)(tab_stride + 32) * 4 + 15) / 16 * 16 : 0);
}

int launch_attn_head_bwd(const HeadAttnArgs& a, int nB, hipStream_t st) {
  M2M_REQUIRE(a.Sq >= 1 && a.Sk >= 1 && a.Sq <= AH_MAX_S && a.Sk <= AH_MAX_S, "attn_head: sequence lengths (%d, %d) beyond %d", a.Sq, a.Sk, AH_MAX_S);
  M2M_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.lddq % 8 == 0 && a.lddk % 8 == 0 && a.lddv % 8 == 0 &&
                  a.sQb % 8 == 0 && a.sKb % 8 == 0 && a.sVb % 8 == 0 && a.sOb % 8 == 0 && a.sdQb % 8 == 0 && a.sdKb % 8 == 0 && a.sdVb % 8 == 0,
              "attn_head: operand strides must keep 16-byte alignment");
  const bool bias = a.bias_tab != nullptr, drop = a.thresh != 0;
  const size_t smem = attn_head_bwd_smem(a.Sq, a.Sk, bias, a.tab_stride);
  M2M_REQUIRE(smem <= 80 * 1024, "attn_head: %zu bytes of LDS", smem);
  const int nq = ceil_div(a.Sq, 32), nk = ceil_div(a.Sk, 32);
  dim3 grid((unsigned)ceil_div(std::max(nq, nk), 4), (unsigned)(nB * a.H));
#define M2M_AH_BWD(C_, B_, D_)                                                                   \
  do {                                                                                           \
    M2M_OPT_IN_LDS((attn_head_bwd_kernel<C_, B_, D_>), 158 * 1024);                              \
    hipLaunchKernelGGL((attn_head_bwd_kernel<C_, B_, D_>), grid, dim3(256), smem, st, a);        \
  } while (0)
  if (a.causal) {
    if (bias) { if (drop) M2M_AH_BWD(true, true, true); else M2M_AH_BWD(true, true, false); }
    else { if (drop) M2M_AH_BWD(true, false, true); else M2M_AH_BWD(true, false, false); }
  } else {
    if (bias) { if (drop) M2M_AH_BWD(false, true, true); else M2M_AH_BWD(false, true, false); }
    else { if (drop) M2M_AH_BWD(false, false, true); else M2M_AH_BWD(false, false, false); }
  }
#undef M2M_AH_BWD
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

}  // namespace m2m

// ---------------------------------------------------------------------------------------------------- test utilities (C ABI)
using namespace m2m;

namespace {
// one 64-bit word on the device holding the dropout step key of a test call
struct StepWord {
  uint64_t* dev = nullptr;
  int set(uint64_t v, hipStream_t st) {
    if (!dev) M2M_CHECK_HIP(hipMalloc((void**)&dev, 8));
    M2M_CHECK_HIP(hipMemcpyAsync(dev, &v, 8, hipMemcpyHostToDevice, st));
    M2M_CHECK_HIP(hipStreamSynchronize(st));
    return M2M_OK;
  }
};
}  // namespace

extern "C" int m2m_attn_head_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, const float* bias_tab, int B, int H, int Sq, int Sk,
                                      int causal, float drop_p, uint64_t step_key, uint64_t site_salt, uint16_t* out, float* lse, void* stream) {
  M2M_REQUIRE(q && k && v && out && lse && B >= 1 && H >= 1, "m2m_attn_head_fwd_bf16: bad argument");
  static StepWord word;
  hipStream_t st = (hipStream_t)stream;
  int rc = word.set(step_key, st);
  if (rc != M2M_OK) return rc;
  HeadAttnArgs a{};
  const int inner = H * 64;
  a.Q = (const bf16_t*)q; a.K = (const bf16_t*)k; a.V = (const bf16_t*)v; a.O = (bf16_t*)out; a.lse = lse;
  a.ldq = a.ldk = a.ldv = a.ldo = inner; a.sQb = (int64_t)Sq * inner; a.sKb = a.sVb = (int64_t)Sk * inner; a.sOb = (int64_t)Sq * inner;
  a.bias_tab = bias_tab; a.tab_stride = Sq + Sk - 1; a.tab_center = Sq - 1;
  a.H = H; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.ldp = (Sk + 7) / 8 * 8;
  a.dk = DropKey{word.dev, site_salt}; a.thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u; a.scale = 1.0f / (1.0f - drop_p);
  return launch_attn_head_fwd(a, B, st);
}

extern "C" int m2m_attn_head_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out, const float* lse, const uint16_t* d_out,
                                      const float* bias_tab, int B, int H, int Sq, int Sk, int causal, float drop_p, uint64_t step_key, uint64_t site_salt,
                                      uint16_t* dq, uint16_t* dk, uint16_t* dv, float* diag_part, void* stream) {
  M2M_REQUIRE(q && k && v && out && lse && d_out && dq && dk && dv && B >= 1 && H >= 1, "m2m_attn_head_bwd_bf16: bad argument");
  static StepWord word;
  hipStream_t st = (hipStream_t)stream;
  int rc = word.set(step_key, st);
  if (rc != M2M_OK) return rc;
  HeadAttnArgs a{};
  const int inner = H * 64;
  a.Q = (const bf16_t*)q; a.K = (const bf16_t*)k; a.V = (const bf16_t*)v; a.O = (bf16_t*)const_cast<uint16_t*>(out); a.lse = const_cast<float*>(lse);
  a.dO = (const bf16_t*)d_out; a.dQ = (bf16_t*)dq; a.dK = (bf16_t*)dk; a.dV = (bf16_t*)dv;
  a.ldq = a.ldk = a.ldv = a.ldo = a.lddq = a.lddk = a.lddv = inner;
  a.sQb = a.sOb = a.sdQb = (int64_t)Sq * inner; a.sKb = a.sVb = a.sdKb = a.sdVb = (int64_t)Sk * inner;
  a.bias_tab = bias_tab; a.tab_stride = Sq + Sk - 1; a.tab_center = Sq - 1; a.diag_part = diag_part;
  a.H = H; a.Sq = Sq; a.Sk = Sk; a.causal = causal; a.ldp = (Sk + 7) / 8 * 8;
  a.dk = DropKey{word.dev, site_salt}; a.thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u; a.scale = 1.0f / (1.0f - drop_p);
  return launch_attn_head_bwd(a, B, st);
}
