// Whole-head attention of the training step (bf16), forward and backward, for the short sequences a training batch has
// (ref: music2midi/model.py:32-38 -> hf: modeling_t5.py:159-170 T5Attention: scores = Q K^T (no 1/sqrt(d)) + relative-position
// bias, softmax in fp32, dropout on the probabilities, context = P~ V; its backward is torch autograd's).
//
// Round 2 materialised the probabilities: per attention layer a K|V transpose launch, a "stripe" kernel per 32 queries writing P
// (and reading it back in the backward pass together with a re-emitted dropped copy and dS, 17.6 MB each at 16 clips), and a batched
// dV | dK product — 27-34 us forward and 47-53 us backward per layer, 1.4 ms of a 4.5 ms step, with matrix cores 4-6 % busy.
// Here ONE workgroup family per (clip, head) keeps the head's operands in LDS and never writes a probability:
//   forward   stage K and V (global_load_lds, swizzled); a query block of 32 is walked by TWO waves, each over half of the key tiles
//             (their (m, l, O) meet in LDS at the end), with an ONLINE softmax in the S^T = K Q^T orientation (a lane owns one query: row
//             statistics are in-lane + one lane^32 exchange), in log2 units and with a lazy running maximum; the probabilities go from the
//             accumulator STRAIGHT into the next product as its B operand — O^T = V^T P^T, V^T fragments by transposing LDS reads
//             (ds_read_b64_tr_b16) in the accumulator's own key order; the dropout scale is applied once, to O.  Out: O, the row
//             log-sum-exp and — with dropout — ONE keep-bit word per (query, 32 keys);
//   backward  recomputes P = exp2(S log2e + bias - lse) twice, once per orientation, so that both reductions stay inside a wave:
//             pass A, a wave per 32-KEY block (S = Q K^T, a lane owns one key): dV^T += dO^T P~, dK^T += Q^T dS with P~ / dS from
//             the accumulators as B operands and dO^T / Q^T by transposing reads; pass B, a wave per 32-QUERY block (S^T again):
//             dQ^T += K^T dS^T.  delta = rowsum(dO o O) replaces sum_k P dP~ (the flash-attention identity, dropout included).
//             Both passes read the forward pass's keep-bit words instead of hashing again.
// The dropout masks are the step's counter-based hash with the SAME element index as the stored-probability path
// ((clip*H + head) * Sq + query) * ldp + key, hashed ONCE (forward), so the oracle regenerates them; the relative-position-bias
// gradient leaves pass B as per-query-block diagonal sums — summed on the matrix core from a skewed copy of the dS tile — in the layout
// the stripe kernel used (bias_stripes_sum_kernel / bias_bucket_kernel are unchanged).  fp32 (parity) mode, fp32 storage and sequences
// beyond AH_MAX_S keep the stripe path.  DESIGN_HISTORY.md 4.8 has the measurements, including what was tried and removed.
#include "mma.h"
#include "t5.h"
#include "train.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace m2m {


// LDS image of a [rows, 64] bf16 operand: 128-byte rows, no padding; 16-byte chunk c of row r sits at slot c ^ ah_g(r).  The
// natural fragment reads (lane = row, 16 lanes cover (r & 1, slot) = all 16 bank groups) and the transposing reads (four consecutive
// rows of an aligned group x 64 bytes: rows 0-1 land in one 64-byte half of their 128-byte bank half, rows 2-3 in the other) are
// both conflict-free with it.
__device__ inline int ah_g(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ inline int ah_off(int row, int chunk) { return row * 64 + ((chunk ^ ah_g(row)) << 3); }

typedef short ah_v4s __attribute__((ext_vector_type(4)));
// The swizzle of a row depends on its low five bits only, so inside a 32-row tile every fragment address is (tile base) + (a lane
// constant): the loops below keep these constants in registers and add the tile base — the index arithmetic of ah_off / ah_tr_frag per
// fragment and tile was a quarter of the VALU instructions of a tile.
// natural fragment s: row (lane & 31) of the tile, k = d in [16 s + 8 h, +8)           (h = lane >> 5)
// transposing fragment (s, db) = a fragment of X^T: rows = the image's columns [32 db, +32), k = the 16 image rows
//   {16 s + 4 h + (0..3)} u {16 s + 8 + 4 h + (0..3)} — the order in which an accumulator lane holds ITS rows, so a probability tile
//   feeds the next product from the registers it was computed in (two ds_read_b64_tr_b16, each four rows x 64 bytes)
struct AhLaneOffs {
  int nat[4];        // natural fragment s of tile row (lane & 31)
  int tr[2][2];      // transposing fragment: [db][first / second group of four rows], + 16 s * 64 for substep s
};
__device__ inline AhLaneOffs ah_lane_offs(int lane) {
  AhLaneOffs o;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 4; ++s) o.nat[s] = ah_off(r, 2 * s + h);
  const int li = lane & 15, q = li >> 2, pp = li & 3, gq = lane >> 4;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int c = 4 * db + 2 * (gq & 1) + (pp >> 1), sub = (pp & 1) * 4;
    o.tr[db][0] = ah_off(4 * h + q, c) + sub;
    o.tr[db][1] = ah_off(4 * h + q + 8, c) + sub;
  }
  return o;
}
__device__ inline Frag<bf16_t> ah_tr_frag_at(const bf16_t* T, const AhLaneOffs& o, int s, int db) {      // T: the tile's first row in its image
  typedef ah_v4s __attribute__((address_space(3))) * lds_v4s;
  const ah_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(T + 16 * 64 * s + o.tr[db][0]));
  const ah_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(T + 16 * 64 * s + o.tr[db][1]));
  Frag<bf16_t> f;
  f.v = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
  return f;
}
constexpr int AH_IMG = 288 * 64;      // elements of one LDS image (AH_MAX_S rows): images sit at compile-time distances from one another

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


// all ones when bit `bit` of w is set, else zero (v_bfe_i32): a dropout keep bit as an AND mask on the value's bits
__device__ inline uint32_t ah_bit_mask(uint32_t w, int bit) { return (uint32_t)((int32_t)(w << (31 - bit)) >> 31); }
constexpr float AH_LOG2E = 1.4426950408889634f;
constexpr float AH_LAZY = 6.0f;

// ------------------------------------------------------------------------------------------------------------ forward
template <bool CAUSAL, bool BIAS, bool DROP>
__global__ __launch_bounds__(512, 2) void attn_head_fwd_kernel(HeadAttnArgs a) {
  extern __shared__ __align__(1024) unsigned char ah_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hh = bh - b * a.H;
  const int nk = (a.Sk + 31) >> 5, Skp = nk * 32, Sqp = ((a.Sq + 31) >> 5) * 32;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(ah_smem);
  bf16_t* Vs = Ks + AH_IMG;
  float* bias_s = reinterpret_cast<float*>(Vs + AH_IMG);
  const bf16_t* Kg = a.K + b * a.sKb + hh * 64;
  const bf16_t* Vg = a.V + b * a.sVb + hh * 64;
  // A query block is walked by `split` waves, each over its share of the key tiles; their (m, l, O) meet in LDS at the end.  A kernel of
  // this family cannot end before one wave has walked its tiles (12 us for 9 tiles with a quarter of the heads), and at 16 clips there
  // are only 1.1 waves per SIMD to begin with: two half-length walks per block halve that chain and give every SIMD a second wave to
  // switch to.  nw = query blocks per workgroup, blockDim = 64 nw split.
  const int nwaves = blockDim.x >> 6, split = a.key_split, nw = nwaves / split;
  ah_stage(Ks, Kg, a.ldk, a.Sk, Skp, wave, lane, nwaves);
  ah_stage(Vs, Vg, a.ldv, a.Sk, Skp, wave, lane, nwaves);
  // (round 5, measured and dropped: this loop compiles to load -> s_waitcnt vmcnt(0) -> store per iteration, two or three round trips;
  // four entries per thread requested before the K / V transfers instead: 21.0 / 13.8 against 20.7 / 13.2 us — no gain at these sizes)
  if (BIAS)
    for (int i = threadIdx.x; i < a.tab_stride + 32; i += blockDim.x) bias_s[i] = i < a.tab_stride ? a.bias_tab[(int64_t)hh * a.tab_stride + i] * AH_LOG2E : 0.f;
  // this wave's query block and its part of the key tiles
  const int part = wave / nw, wq = wave - part * nw;
  const int qi = nw * blockIdx.x + wq;
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
  // The softmax runs in log2 units (scores and bias scaled by log2 e: a probability is one multiply-add, one subtract and one exp2) with
  // a LAZY running maximum: the accumulators are rescaled only when some query's tile maximum exceeds its running one by more than
  // 2^AH_LAZY — until then p = exp2(s - m) may reach 2^AH_LAZY instead of 1, harmless in fp32 sums and bf16 operands, and m + log2(l)
  // is the same log-sum-exp.  The dropout scale 1 / (1 - p) is applied once, to O.
  f32x16 oacc[2] = {zero_acc(), zero_acc()};
  float m = -1e30f, l = 0.f;                                  // l: this lane's half of the row sum (its 16 keys per tile); halves meet at the end
  if (active) {
    const uint64_t key = DROP ? drop_site_key(a.dk) : 0ull;
    const int64_t prow = ((int64_t)bh * a.Sq + qc) * a.ldp;
    const float* bt = bias_s + a.tab_center - qc + 4 * h;
    const int jend = CAUSAL ? min(nk, qi + 1) : nk;          // causal: tiles past the query block's diagonal hold no key <= q
    const bool ragged = (a.Sk & 31) != 0;
    const AhLaneOffs lo = ah_lane_offs(lane);
    auto tile = [&](int j, auto masked) {
      constexpr bool MASK = decltype(masked)::value;          // the last key block (keys beyond Sk) and the causal diagonal
      f32x16 acc = zero_acc();
      const bf16_t* Kt = Ks + 32 * 64 * j;
#pragma unroll
      for (int s = 0; s < 4; ++s) mma16(acc, load_frag(Kt + lo.nat[s]), qf[s]);      // S^T: rows = keys, cols = queries
      float tm = -1e30f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = (e & 3) + 8 * (e >> 2);                // key 32 j + kk + 4 h
        float v = BIAS ? fmaf(acc[e], AH_LOG2E, bt[32 * j + kk]) : acc[e] * AH_LOG2E;
        if (MASK) {
          const int k = 32 * j + kk + 4 * h;
          v = (k < a.Sk && (!CAUSAL || k <= qc)) ? v : -1e30f;
        }
        acc[e] = v;
        tm = fmaxf(tm, v);
      }
      tm = fmaxf(tm, lane_xor<32>(tm));
      if (__builtin_amdgcn_ballot_w64(tm > m + AH_LAZY) != 0ull) {
        const float mn = fmaxf(m, tm);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        l *= alpha;
        m = mn;
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
      }
      float ts = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[e] = __builtin_amdgcn_exp2f(acc[e] - m); ts += acc[e]; }      // masked entries: exp2(-1e30 - m) = 0
      l += ts;
      if (DROP) {
        uint32_t w = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t kb = drop_keep4(key, prow + 32 * j + 8 * g + 4 * h, a.thresh);
          w |= kb << (8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * g + e] = __uint_as_float(__float_as_uint(acc[4 * g + e]) & ah_bit_mask(kb, e));
        }
        w <<= 4 * h;                                           // bit k of the word = key 32 j + k of this lane's query
        w |= __float_as_uint(lane_xor<32>(__uint_as_float(w)));
        if (h == 0) a.keep_bits[((int64_t)bh * nk + j) * Sqp + qi * 32 + r] = w;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float pv[8] = {acc[8 * s], acc[8 * s + 1], acc[8 * s + 2], acc[8 * s + 3], acc[8 * s + 4], acc[8 * s + 5], acc[8 * s + 6], acc[8 * s + 7]};
        const Frag<bf16_t> pf = pack_frag<bf16_t>(pv);
#pragma unroll
        for (int db = 0; db < 2; ++db) mma16(oacc[db], ah_tr_frag_at(Kt + AH_IMG, lo, s, db), pf);      // O^T += V^T P^T
      }
    };
    const int per = (jend + split - 1) / split;              // tiles [part * per, (part + 1) * per) n [0, jend)
    const int j0 = part * per, j1 = min(jend, j0 + per);
    for (int j = j0; j < j1; ++j) {
      if ((CAUSAL && j == qi) || (ragged && j == nk - 1)) tile(j, std::true_type{});
      else tile(j, std::false_type{});
    }
    l += lane_xor<32>(l);
  }
  __syncthreads();                                                   // every wave is done with K / V: their LDS becomes merge / output staging
  constexpr int OP = 72;                                             // staging pitch (elements)
  if (split > 1) {
    // parts 1.. leave (m, l, O^T accumulators) in LDS — [part - 1][query block][lane][34 floats], behind the output staging — and part 0 folds them in
    float* Mg = reinterpret_cast<float*>(Ks + 4 * 32 * OP);
    if (part > 0 && active) {
      float* dst = Mg + ((size_t)((part - 1) * nw + wq) * 64 + lane) * 36;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(dst + 16 * db + 4 * g) = make_float4(oacc[db][4 * g], oacc[db][4 * g + 1], oacc[db][4 * g + 2], oacc[db][4 * g + 3]);
      dst[32] = m; dst[33] = l;
    }
    __syncthreads();
    if (part > 0) return;
    if (active) {
      for (int p = 1; p < split; ++p) {
        const float* src = Mg + ((size_t)((p - 1) * nw + wq) * 64 + lane) * 36;
        const float m1 = src[32], l1 = src[33];
        const float mn = fmaxf(m, m1);
        const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);      // (a part without tiles: m1 = -1e30, l1 = 0, O = 0)
        l = l * a0 + l1 * a1;
        m = mn;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 o1 = *reinterpret_cast<const float4*>(src + 16 * db + 4 * g);
            oacc[db][4 * g] = oacc[db][4 * g] * a0 + o1.x * a1; oacc[db][4 * g + 1] = oacc[db][4 * g + 1] * a0 + o1.y * a1;
            oacc[db][4 * g + 2] = oacc[db][4 * g + 2] * a0 + o1.z * a1; oacc[db][4 * g + 3] = oacc[db][4 * g + 3] * a0 + o1.w * a1;
          }
      }
    }
  }
  if (!active) return;
  const float inv = (DROP ? a.scale : 1.0f) / l;
  bf16_t* Os = Ks + wq * 32 * OP;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2 pk = make_uint2(pack2_bf16(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv), pack2_bf16(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv));
      *reinterpret_cast<uint2*>(Os + r * OP + 32 * db + 8 * g + 4 * h) = pk;
    }
  if (h == 0 && q < a.Sq) a.lse[(int64_t)bh * a.Sq + q] = (m + __log2f(l)) * 0.6931471805599453f;
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

// waves per workgroup for `blocks` 32-row blocks: every workgroup stages the whole head, so idle waves in the last one are pure cost
// (9 blocks: 3 x 3 waves instead of 4 + 4 + 1)
static int ah_waves(int blocks) {
  static const int forced = [] { const char* v = getenv("M2M_AH_WAVES"); return v ? atoi(v) : 0; }();
  if (forced >= 1 && forced <= 4) return forced;
  if (blocks <= 4) return blocks;
  return (ceil_div(blocks, 3) * 3 - blocks) < (ceil_div(blocks, 4) * 4 - blocks) ? 3 : 4;
}

size_t attn_head_fwd_smem(int Sk, bool bias, int tab_stride) {
  (void)Sk;
  const size_t kv = 2 * (size_t)AH_IMG * sizeof(bf16_t), bias_b = bias ? ((size_t)(tab_stride + 32) * 4 + 15) / 16 * 16 : 0;
  const size_t out_stage = (size_t)4 * 32 * 72 * sizeof(bf16_t) + (size_t)4 * 64 * 36 * sizeof(float);      // output tiles + (split - 1) nw <= 4 merge records re-use the K / V (/ bias) bytes
  return kv + bias_b > out_stage ? kv + bias_b : out_stage;
}

int launch_attn_head_fwd(const HeadAttnArgs& a, int nB, hipStream_t st) {
  M2M_REQUIRE(a.Sq >= 1 && a.Sk >= 1 && a.Sq <= AH_MAX_S && a.Sk <= AH_MAX_S, "attn_head: sequence lengths (%d, %d) beyond %d", a.Sq, a.Sk, AH_MAX_S);
  M2M_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.sQb % 8 == 0 && a.sKb % 8 == 0 && a.sVb % 8 == 0 && a.sOb % 8 == 0,
              "attn_head: operand strides must keep 16-byte alignment");
  const bool bias = a.bias_tab != nullptr, drop = a.thresh != 0;
  M2M_REQUIRE(!drop || a.keep_bits, "attn_head: dropout needs the keep-bit buffer");
  const size_t smem = attn_head_fwd_smem(a.Sk, bias, a.tab_stride);
  M2M_REQUIRE(smem <= 78 * 1024, "attn_head: %zu bytes of LDS", smem);
  // two waves per query block (halves of the key tiles) when the head has at least two key tiles; M2M_AH_SPLIT=1 / 2 / 3 forces the count
  static const int forced_split = [] { const char* v = getenv("M2M_AH_SPLIT"); return v ? atoi(v) : 0; }();
  const int nkt = ceil_div(a.Sk, 32);
  const int split = std::max(1, std::min(forced_split >= 1 && forced_split <= 3 ? forced_split : 2, nkt));
  int nw = ah_waves(ceil_div(a.Sq, 32));
  if (nw * split > 8) nw = 8 / split;
  if (split > 1 && nw > 3 && ceil_div(a.Sq, 32) % 4 != 0) nw = 3;      // (six waves, two workgroups per CU; eight when the blocks come in fours)
  HeadAttnArgs a2 = a;
  a2.key_split = split;
  dim3 grid((unsigned)ceil_div(ceil_div(a.Sq, 32), nw), (unsigned)(nB * a.H));
#define M2M_AH_FWD(C_, B_, D_)                                                                           \
  do {                                                                                                   \
    M2M_OPT_IN_LDS((attn_head_fwd_kernel<C_, B_, D_>), 158 * 1024);                                      \
    hipLaunchKernelGGL((attn_head_fwd_kernel<C_, B_, D_>), grid, dim3(64 * nw * split), smem, st, a2);   \
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

constexpr int AH_SKEW_P = 40;      // pitch (elements) of the skewed dS tile of the bias gradient: 80-byte rows keep its 16-byte column reads conflict-free

template <bool CAUSAL, bool BIAS, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_head_bwd_kernel(HeadAttnArgs a) {
  extern __shared__ __align__(1024) unsigned char ah_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hh = bh - b * a.H;
  const int nq = (a.Sq + 31) >> 5, nk = (a.Sk + 31) >> 5, Sqp = nq * 32, Skp = nk * 32;
  bf16_t* Xa = reinterpret_cast<bf16_t*>(ah_smem);             // pass A: Q image; pass B: K image
  bf16_t* Xb = Xa + AH_IMG;                                    // pass A: dO image; then per-wave staging / scratch
  // Row statistics and bias rows are staged in the units the element loop wants, so that a probability costs a subtract, a
  // multiply-add and an exp2:  P~ = P / (1 - p) = exp2(S log2e + bias log2e - (lse log2e - log2 scale)),  dS = P~ (keep . dP - delta / scale)
  float* lse_s = reinterpret_cast<float*>(Xb + AH_IMG);        // [Sqp]: lse log2e - log2 scale; +1e30 beyond Sq (=> P = 0 there)
  float* delta_s = lse_s + Sqp;                                // [Sqp]: rowsum(dO o O) / scale
  float* bias_s = delta_s + Sqp;                               // [tab_stride + 32]: bias log2e by (key - query + tab_center), zeros behind
  float* biasr_s = bias_s + a.tab_stride + 32;                 // the same row reversed (pass A walks queries upwards for a fixed key), zeros behind
  const bf16_t* Qg = a.Q + b * a.sQb + hh * 64;
  const bf16_t* Kg = a.K + b * a.sKb + hh * 64;
  const bf16_t* Vg = a.V + b * a.sVb + hh * 64;
  const bf16_t* Og = a.O + b * a.sOb + hh * 64;
  const bf16_t* dOg = a.dO + b * a.sOb + hh * 64;
  const int nw = blockDim.x >> 6;
  const int blk = nw * blockIdx.x + wave;                      // this wave's key block (pass A) and query block (pass B)
  const float inv_scale = DROP ? 1.0f / a.scale : 1.0f, lg_scale = DROP ? __log2f(a.scale) : 0.f;

  // ---- phase 0: Q, dO -> LDS; delta = rowsum(dO o O), lse, bias rows; this wave's K / V block -> registers
  ah_stage(Xa, Qg, a.ldq, a.Sq, Sqp, wave, lane, nw);
  ah_stage(Xb, dOg, a.ldo, a.Sq, Sqp, wave, lane, nw);
  for (int q = threadIdx.x; q < Sqp; q += blockDim.x) {
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
      lv = a.lse[(int64_t)bh * a.Sq + q] * AH_LOG2E - lg_scale;
    }
    delta_s[q] = dsum * inv_scale;
    lse_s[q] = lv;
  }
  if (BIAS)
    for (int i = threadIdx.x; i < a.tab_stride + 32; i += blockDim.x) {
      bias_s[i] = i < a.tab_stride ? a.bias_tab[(int64_t)hh * a.tab_stride + i] * AH_LOG2E : 0.f;
      biasr_s[i] = i < a.tab_stride ? a.bias_tab[(int64_t)hh * a.tab_stride + (a.tab_stride - 1 - i)] * AH_LOG2E : 0.f;
    }
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
  // Keys beyond Sk (clamped copies of the last row) need no masking here: they only reach columns of dV^T / dK^T that are never stored.
  f32x16 dvT[2] = {zero_acc(), zero_acc()}, dkT[2] = {zero_acc(), zero_acc()};
  const AhLaneOffs lo = ah_lane_offs(lane);
  if (actA) {
    const int kcol = blk * 32 + r;                              // this lane's key
    const uint32_t* kbits = DROP ? a.keep_bits + ((int64_t)bh * nk + blk) * Sqp + 4 * h : nullptr;
    const float* brow = biasr_s + (a.tab_stride - 1 - a.tab_center - kcol) + 4 * h;      // brow[q - 4 h] = bias log2e of (kcol - q)
    const float* lrow = lse_s + 4 * h;
    const float* drow = delta_s + 4 * h;
    const int i0 = CAUSAL ? blk : 0;
    uint4 kw[4];
    if (DROP) {
#pragma unroll
      for (int g = 0; g < 4; ++g) kw[g] = *reinterpret_cast<const uint4*>(kbits + 32 * i0 + 8 * g);
    }
    auto tile = [&](int i, auto on_diag) {
      constexpr bool DIAG = decltype(on_diag)::value;           // the causal diagonal tile: keys past the query are masked
      f32x16 sacc = zero_acc(), pacc = zero_acc();
      const bf16_t* Qt = Xa + 32 * 64 * i;                      // (the dO tile sits AH_IMG behind it)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma16(sacc, load_frag(Qt + lo.nat[s]), kf[s]);
        mma16(pacc, load_frag(Qt + AH_IMG + lo.nat[s]), vf[s]);
      }
      float ptv[16], dsv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = *reinterpret_cast<const float4*>(lrow + 32 * i + 8 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(drow + 32 * i + 8 * g);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
        const uint32_t kq[4] = {kw[g].x, kw[g].y, kw[g].z, kw[g].w};      // the keep words of queries 32 i + 8 g + 4 h + (0..3) against this key block
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = 4 * g + u;
          const float t = BIAS ? brow[32 * i + 8 * g + u] - lq[u] : -lq[u];
          float arg = fmaf(sacc[e], AH_LOG2E, t);
          if (DIAG) arg = (r <= u + 8 * g + 4 * h) ? arg : -1e30f;
          const float ps = __builtin_amdgcn_exp2f(arg);         // P / (1 - p)
          if (DROP) {
            const uint32_t mk = (uint32_t)__builtin_amdgcn_sbfe((int)kq[u], (unsigned)r, 1u);
            ptv[e] = __uint_as_float(__float_as_uint(ps) & mk);
            dsv[e] = ps * (__uint_as_float(__float_as_uint(pacc[e]) & mk) - dq[u]);
          } else {
            ptv[e] = ps;
            dsv[e] = ps * (pacc[e] - dq[u]);
          }
        }
      }
      if (DROP && i + 1 < nq) {                                 // the next tile's keep words: in flight behind the eight products below
#pragma unroll
        for (int g = 0; g < 4; ++g) kw[g] = *reinterpret_cast<const uint4*>(kbits + 32 * (i + 1) + 8 * g);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float pv[8] = {ptv[8 * s], ptv[8 * s + 1], ptv[8 * s + 2], ptv[8 * s + 3], ptv[8 * s + 4], ptv[8 * s + 5], ptv[8 * s + 6], ptv[8 * s + 7]};
        const float dv[8] = {dsv[8 * s], dsv[8 * s + 1], dsv[8 * s + 2], dsv[8 * s + 3], dsv[8 * s + 4], dsv[8 * s + 5], dsv[8 * s + 6], dsv[8 * s + 7]};
        const Frag<bf16_t> pf = pack_frag<bf16_t>(pv), df = pack_frag<bf16_t>(dv);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          mma16(dvT[db], ah_tr_frag_at(Qt + AH_IMG, lo, s, db), pf);      // dV^T += dO^T P~
          mma16(dkT[db], ah_tr_frag_at(Qt, lo, s, db), df);               // dK^T += Q^T dS
        }
      }
    };
    int i = i0;
    if (CAUSAL) { tile(i, std::true_type{}); ++i; }
    for (; i < nq; ++i) tile(i, std::false_type{});
  }
  __syncthreads();                                               // every wave is done with the Q / dO images

  // ---- between the passes: K -> LDS (pass B reads it both ways); dV / dK tiles out through the second image; Q_i, dO_i -> registers
  ah_stage(Xa, Kg, a.ldk, a.Sk, Skp, wave, lane, nw);
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
  const float lse_q = lse_s[blk * 32 + r], delta_q = delta_s[blk * 32 + r];      // (rows beyond Sq: lse = 1e30 => P = 0)
  // Relative-position-bias gradient, stage 1: sums of this query block's dS along the diagonals key - local row = x - 31 (the layout
  // bias_stripes_sum_kernel reads).  The tile is written SKEWED — dS[row][key] at [x = key - row + 31][row], bf16 as the product used it
  // — so a diagonal becomes a row of 32, and its sum a product with a vector of ones on the otherwise idle matrix core; the upper
  // half of a tile's 63 diagonals continues in the next tile's lower half and rides there as the accumulator input.
  const bool want_diag = a.diag_part != nullptr;
  bf16_t* skew = Tw;                                            // [64][AH_SKEW_P]; entries outside a row's 32-wide band stay zero
  const int dl = a.Sk + 31;
  float* dout = want_diag ? a.diag_part + ((int64_t)bh * nq + blk) * dl : nullptr;
  if (want_diag) {
    for (int x = lane; x < 64 * AH_SKEW_P / 8; x += 64) reinterpret_cast<uint4*>(skew)[x] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  uint16_t* skew_w = reinterpret_cast<uint16_t*>(skew) + (31 + 4 * h - r) * AH_SKEW_P + r;       // + key-in-tile * AH_SKEW_P
  const bf16_t* skew_r = skew + r * AH_SKEW_P + 8 * h;                                              // + 32 AH_SKEW_P nb + 16 s
  Frag<bf16_t> ones;
  ones.v = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
  f32x16 carry = zero_acc();
  const float* bt = bias_s + a.tab_center - qc;
  const int jend = CAUSAL ? min(nk, blk + 1) : nk;
  const uint32_t* kbq = DROP ? a.keep_bits + (int64_t)bh * nk * Sqp + blk * 32 + r : nullptr;
  uint32_t kw_next = DROP ? kbq[0] : 0u;
  Frag<bf16_t> vfr[4];
  {
    const int vrow = min(r, a.Sk - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) vfr[s] = load_frag(Vg + (int64_t)vrow * a.ldv + 16 * s + 8 * h);
  }
  auto tileB = [&](int j, auto masked, auto with_diag) {
    constexpr bool MASK = decltype(masked)::value;              // the last key block (keys beyond Sk) and the causal diagonal
    constexpr bool DIAG = decltype(with_diag)::value;           // == want_diag, as a compile-time fact of the loop the tile runs in
    f32x16 sacc = zero_acc(), pacc = zero_acc();
    const uint32_t kwq = kw_next >> (4 * h);                    // this query's keep bits against key block j
    const bf16_t* Kt = Xa + 32 * 64 * j;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      mma16(sacc, load_frag(Kt + lo.nat[s]), qf[s]);
      mma16(pacc, vfr[s], dof[s]);
    }
    if (j + 1 < jend) {                                         // the next tile's V rows and keep word: in flight behind this tile's element work
      const int vrow = min(32 * (j + 1) + r, a.Sk - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) vfr[s] = load_frag(Vg + (int64_t)vrow * a.ldv + 16 * s + 8 * h);
      if (DROP) kw_next = kbq[(int64_t)(j + 1) * Sqp];
    }
    float dsv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 4 * g + u, kk = 8 * g + u;               // key 32 j + kk + 4 h
        const float t = BIAS ? bt[32 * j + 4 * h + kk] - lse_q : -lse_q;
        float arg = fmaf(sacc[e], AH_LOG2E, t);
        if (MASK) {
          const int k = 32 * j + 4 * h + kk;
          arg = (k < a.Sk && (!CAUSAL || k <= qc)) ? arg : -1e30f;
        }
        const float ps = __builtin_amdgcn_exp2f(arg);
        const float dpt = DROP ? __uint_as_float(__float_as_uint(pacc[e]) & (uint32_t)__builtin_amdgcn_sbfe((int)kwq, (unsigned)kk, 1u)) : pacc[e];
        dsv[e] = ps * (dpt - delta_q);
      }
    }
    Frag<bf16_t> df[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float dv[8] = {dsv[8 * s], dsv[8 * s + 1], dsv[8 * s + 2], dsv[8 * s + 3], dsv[8 * s + 4], dsv[8 * s + 5], dsv[8 * s + 6], dsv[8 * s + 7]};
      df[s] = pack_frag<bf16_t>(dv);
#pragma unroll
      for (int db = 0; db < 2; ++db) mma16(dqT[db], ah_tr_frag_at(Kt, lo, s, db), df[s]);      // dQ^T += K^T dS^T
    }
    if (DIAG && (!CAUSAL || want_diag)) {                       // (causal: one loop with the run-time test — two loops there cost a spill)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t w[4] = {df[s].v.x, df[s].v.y, df[s].v.z, df[s].v.w};
#pragma unroll
        for (int m = 0; m < 4; ++m) {                          // elements e = 8 s + 2 m, + 1: keys kk = 8 (2 s + (m >> 1)) + 2 (m & 1), + 1
          const int kk = 8 * (2 * s + (m >> 1)) + 2 * (m & 1);
          skew_w[kk * AH_SKEW_P] = (uint16_t)(w[m] & 0xFFFFu);
          skew_w[(kk + 1) * AH_SKEW_P] = (uint16_t)(w[m] >> 16);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x16 c0 = carry, c1 = zero_acc();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        mma16(c0, ones, load_frag(skew_r + 16 * s));
        mma16(c1, ones, load_frag(skew_r + 32 * AH_SKEW_P + 16 * s));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // Stored by BOTH half-waves (the upper one repeats the lower one's value and address), in a loop compiled for want_diag: behind
      // `if (want_diag)` / `if (h == 0)` the store sat in a branch, the compiler could not tell whether it had been issued, and the
      // next tile's first use of a prefetched register (the keep word, the V rows) waited vmcnt(0) — for this store's
      // acknowledgement, a memory write round trip in every tile of the bias variants (found in the ISA; vector-memory
      // operations retire in order, stores included).  Now the wait is vmcnt(1).
      // (The causal kernel keeps the branchy form: it is shorter-lived per tile and measured 26.6 us so against 26.9 - 27.5.)
      if constexpr (CAUSAL) {
        if (h == 0) dout[32 * j + r] = c0[0];                  // (32 j + 31 <= Sk + 30: always inside the row)
      } else {
        const float drow = c0[0];
        dout[32 * j + r] = h == 0 ? drow : lane_xor<32>(drow);
      }
      carry = c1;
    }
  };
  const bool ragged = (a.Sk & 31) != 0;
  if (CAUSAL || want_diag) {
    for (int j = 0; j < jend; ++j) {
      if ((CAUSAL && j == blk) || (ragged && j == nk - 1)) tileB(j, std::true_type{}, std::true_type{});
      else tileB(j, std::false_type{}, std::true_type{});
    }
  } else {
    for (int j = 0; j < jend; ++j) {
      if ((CAUSAL && j == blk) || (ragged && j == nk - 1)) tileB(j, std::true_type{}, std::false_type{});
      else tileB(j, std::false_type{}, std::false_type{});
    }
  }
  if (want_diag) {
    // the upper diagonals of the last tile visited, then zeros for what a causal block never reaches
    if (h == 0) {
      if (32 * jend + r < dl) dout[32 * jend + r] = r < 31 ? carry[0] : 0.f;
      for (int x = 32 * (jend + 1) + r; x < dl; x += 32) dout[x] = 0.f;
    }
  }
  ah_tile_to_lds(Tw, dqT, r, h, 1.0f);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  ah_tile_to_global(Tw, a.dQ + b * a.sdQb + hh * 64, a.lddq, blk * 32, a.Sq, lane);
}

size_t attn_head_bwd_smem(int Sq, int Sk, bool bias, int tab_stride) {
  const size_t sqp = (size_t)((Sq + 31) / 32) * 32;
  (void)Sk;
  return 2 * (size_t)AH_IMG * sizeof(bf16_t) + 2 * sqp * 4 + (bias ? ((size_t)2 * (tab_stride + 32) * 4 + 15) / 16 * 16 : 0);
}

int launch_attn_head_bwd(const HeadAttnArgs& a, int nB, hipStream_t st) {
  M2M_REQUIRE(a.Sq >= 1 && a.Sk >= 1 && a.Sq <= AH_MAX_S && a.Sk <= AH_MAX_S, "attn_head: sequence lengths (%d, %d) beyond %d", a.Sq, a.Sk, AH_MAX_S);
  M2M_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.lddq % 8 == 0 && a.lddk % 8 == 0 && a.lddv % 8 == 0 &&
                  a.sQb % 8 == 0 && a.sKb % 8 == 0 && a.sVb % 8 == 0 && a.sOb % 8 == 0 && a.sdQb % 8 == 0 && a.sdKb % 8 == 0 && a.sdVb % 8 == 0,
              "attn_head: operand strides must keep 16-byte alignment");
  const bool bias = a.bias_tab != nullptr, drop = a.thresh != 0;
  M2M_REQUIRE(!drop || a.keep_bits, "attn_head: dropout needs the keep-bit buffer");
  const size_t smem = attn_head_bwd_smem(a.Sq, a.Sk, bias, a.tab_stride);
  M2M_REQUIRE(smem <= 80 * 1024, "attn_head: %zu bytes of LDS", smem);
  const int nq = ceil_div(a.Sq, 32), nk = ceil_div(a.Sk, 32);
  const int nw = ah_waves(std::max(nq, nk));
  dim3 grid((unsigned)ceil_div(std::max(nq, nk), nw), (unsigned)(nB * a.H));
#define M2M_AH_BWD(C_, B_, D_)                                                                   \
  do {                                                                                           \
    M2M_OPT_IN_LDS((attn_head_bwd_kernel<C_, B_, D_>), 158 * 1024);                              \
    hipLaunchKernelGGL((attn_head_bwd_kernel<C_, B_, D_>), grid, dim3(64 * nw), smem, st, a);    \
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
  uint64_t last = 0;
  int set(uint64_t v, hipStream_t st) {
    if (dev && v == last) return M2M_OK;                    // (repeated calls with one key — tools/attn_head_bench.py — stay asynchronous)
    if (!dev) M2M_CHECK_HIP(hipMalloc((void**)&dev, 8));
    M2M_CHECK_HIP(hipStreamSynchronize(st));
    M2M_CHECK_HIP(hipMemcpyAsync(dev, &v, 8, hipMemcpyHostToDevice, st));
    M2M_CHECK_HIP(hipStreamSynchronize(st));
    last = v;
    return M2M_OK;
  }
};
}  // namespace

extern "C" int m2m_attn_head_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, const float* bias_tab, int B, int H, int Sq, int Sk,
                                      int causal, float drop_p, uint64_t step_key, uint64_t site_salt, uint16_t* out, float* lse, uint32_t* keep_bits,
                                      void* stream) {
  M2M_REQUIRE(q && k && v && out && lse && B >= 1 && H >= 1 && (keep_bits || drop_p <= 0.f), "m2m_attn_head_fwd_bf16: bad argument");
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
  a.dk = DropKey{word.dev, site_salt}; a.thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u; a.scale = 1.0f / (1.0f - drop_p); a.keep_bits = keep_bits;
  return launch_attn_head_fwd(a, B, st);
}

extern "C" int m2m_attn_head_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out, const float* lse, const uint16_t* d_out,
                                      const float* bias_tab, int B, int H, int Sq, int Sk, int causal, float drop_p, uint64_t step_key, uint64_t site_salt,
                                      const uint32_t* keep_bits, uint16_t* dq, uint16_t* dk, uint16_t* dv, float* diag_part, void* stream) {
  M2M_REQUIRE(q && k && v && out && lse && d_out && dq && dk && dv && B >= 1 && H >= 1 && (keep_bits || drop_p <= 0.f), "m2m_attn_head_bwd_bf16: bad argument");
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
  a.dk = DropKey{word.dev, site_salt}; a.thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u; a.scale = 1.0f / (1.0f - drop_p); a.keep_bits = const_cast<uint32_t*>(keep_bits);
  return launch_attn_head_bwd(a, B, st);
}
