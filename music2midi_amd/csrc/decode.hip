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
//   dec_gemm_kernel  skinny projection  out[<=32 rows, 32 cols] per workgroup, 8 waves split K,
//                    MFMA 32x32 tiles; optional fused RMSNorm on the input rows; epilogues:
//                    QKV (+KV-cache append), plain, residual-add, gated-GELU, logits.
//   dec_attn_kernel  one (clip, head) per 1024-thread workgroup streams K then V straight from
//                    HBM to registers (16 B per lane per load, no LDS staging: each byte is
//                    used once), fp32 softmax, shuffle + LDS reduction.  HBM-bound.
//   dec_head_kernel  argmax / EOS+pad bookkeeping / next-token embedding / step counter.
#include "mma.h"
#include "t5.h"

namespace m2m {

// ===================================================== skinny projection ====
enum { DEPI_QKV = 0, DEPI_PLAIN = 1, DEPI_RESID = 2, DEPI_GATED = 3 };

struct DecGemmArgs {
  const void* x;         // [rows, K] input activations: fp32 (normed epilogues) or T (DEPI_RESID)
  int ldx;
  const float* ln_w;     // [K] RMSNorm weight (QKV / PLAIN / GATED), unused for RESID
  float eps;
  const void* W;         // [Npad, K] T
  int K, N, B;
  const DecState* state;
  // outputs
  void* out;             // PLAIN: float [B, N]; RESID: float x_res [B, N] (+=); GATED: T [B, N/2]; QKV: float q [B, inner]
  int ldo;
  void* kcache;          // QKV: [B][H][Lmax][64] T for this layer
  void* vcache;
  int H, Lmax, inner;
};

// 512 threads = 8 waves, each owning K/8 of the reduction (<= 9 macro steps of 16).  Every global
// load of the workgroup (weights, activations, norm weights, loop state) is issued before the
// first use, so a launch costs about one memory round trip instead of one per k-step; the
// RMSNorm statistics are reduced across waves through LDS while the weight loads are in flight.
constexpr int DG_WAVES = 8;
constexpr int DG_MAXS = 9;   // K <= 8 * 9 * 16 = 1152

template <typename T, int EPI>
__global__ __launch_bounds__(512) void dec_gemm_kernel(DecGemmArgs a) {
  constexpr bool NORM = (EPI != DEPI_RESID);
  __shared__ float ss_s[DG_WAVES][32];
  __shared__ float red[DG_WAVES][32 * 33];
  const int done = a.state->done;   // consumed only before the stores
  const int t = a.state->t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int b0 = blockIdx.y * 32;
  const int K = a.K;
  const T* W = reinterpret_cast<const T*>(a.W);
  const int ks = K / DG_WAVES;          // multiple of 16
  const int ns = ks / 16;               // <= DG_MAXS
  const int kbeg = wave * ks + 8 * h;
  const bool row_ok = (b0 + r) < a.B;
  const T* wr = W + (int64_t)(n0 + r) * K + kbeg;

  Frag<T> wf[DG_MAXS];
#pragma unroll
  for (int s = 0; s < DG_MAXS; ++s)
    if (s < ns) wf[s] = load_frag(wr + 16 * s);

  f32x16 acc = zero_acc();
  if constexpr (NORM) {
    const float* xr = reinterpret_cast<const float*>(a.x) + (int64_t)(b0 + (row_ok ? r : 0)) * a.ldx + kbeg;
    float4 x0[DG_MAXS], x1[DG_MAXS], g0[DG_MAXS], g1[DG_MAXS];
#pragma unroll
    for (int s = 0; s < DG_MAXS; ++s) {
      if (s < ns) {
        if (row_ok) {
          x0[s] = *reinterpret_cast<const float4*>(xr + 16 * s);
          x1[s] = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
        } else {
          x0[s] = make_float4(0.f, 0.f, 0.f, 0.f);
          x1[s] = x0[s];
        }
        g0[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 16 * s);
        g1[s] = *reinterpret_cast<const float4*>(a.ln_w + kbeg + 16 * s + 4);
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < DG_MAXS; ++s)
      if (s < ns)
        ss += x0[s].x * x0[s].x + x0[s].y * x0[s].y + x0[s].z * x0[s].z + x0[s].w * x0[s].w +
              x1[s].x * x1[s].x + x1[s].y * x1[s].y + x1[s].z * x1[s].z + x1[s].w * x1[s].w;
    ss += __shfl_xor(ss, 32, 64);
    if (h == 0) ss_s[wave][r] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < DG_WAVES; ++w) tot += ss_s[w][r];
    const float rs = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
    for (int s = 0; s < DG_MAXS; ++s) {
      if (s < ns) {
        const float xv[8] = {g0[s].x * (x0[s].x * rs), g0[s].y * (x0[s].y * rs), g0[s].z * (x0[s].z * rs),
                             g0[s].w * (x0[s].w * rs), g1[s].x * (x1[s].x * rs), g1[s].y * (x1[s].y * rs),
                             g1[s].z * (x1[s].z * rs), g1[s].w * (x1[s].w * rs)};
        const Frag<T> fa = pack_frag<T>(xv);
        mma16(acc, fa, wf[s]);
      }
    }
  } else {
    const T* xr = reinterpret_cast<const T*>(a.x) + (int64_t)(b0 + (row_ok ? r : 0)) * a.ldx + kbeg;
    Frag<T> xf[DG_MAXS];
#pragma unroll
    for (int s = 0; s < DG_MAXS; ++s)
      if (s < ns) xf[s] = row_ok ? load_frag(xr + 16 * s) : zero_frag<T>();
#pragma unroll
    for (int s = 0; s < DG_MAXS; ++s)
      if (s < ns) mma16(acc, xf[s], wf[s]);
  }
  // ---- cross-wave reduction (fixed order: deterministic) ----
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][acc_row(i, lane) * 33 + r] = acc[i];
  __syncthreads();
  if (done) return;

  auto rsum = [&](int idx) {
    float v = red[0][idx];
#pragma unroll
    for (int w = 1; w < DG_WAVES; ++w) v += red[w][idx];
    return v;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * 512;
    const int row = e >> 5, col = e & 31;
    const int b = b0 + row;
    if (b >= a.B) continue;
    const int idx = row * 33 + col;
    if constexpr (EPI == DEPI_GATED) {
      // tile columns: [16 of wi_0 | the matching 16 of wi_1]
      if (col < 16) {
        const float v0 = rsum(idx), v1 = rsum(idx + 16);
        const int oc = (n0 >> 1) + col;
        if (2 * oc < a.N) reinterpret_cast<T*>(a.out)[(int64_t)b * a.ldo + oc] = from_f32<T>(gelu_new(v0) * v1);
      }
    } else {
      const int n = n0 + col;
      if (n >= a.N) continue;
      const float v = rsum(idx);
      float* outf = reinterpret_cast<float*>(a.out);
      if constexpr (EPI == DEPI_PLAIN) {
        outf[(int64_t)b * a.ldo + n] = v;
      } else if constexpr (EPI == DEPI_RESID) {
        outf[(int64_t)b * a.ldo + n] += v;
      } else {  // DEPI_QKV
        const int which = n / a.inner, rem = n - which * a.inner;
        if (which == 0) {
          outf[(int64_t)b * a.ldo + rem] = v;
        } else {
          const int hh = rem / DK, dd = rem - hh * DK;
          T* cache = reinterpret_cast<T*>(which == 1 ? a.kcache : a.vcache);
          cache[(((int64_t)b * a.H + hh) * a.Lmax + t) * DK + dd] = from_f32<T>(v);
        }
      }
    }
  }
}

template <typename T>
static int launch_dec_gemm_t(int epi, const DecGemmArgs& a, hipStream_t st) {
  const int npad = ceil_div(a.N, 32) * 32;
  dim3 grid((unsigned)(npad / 32), (unsigned)ceil_div(a.B, 32));
  switch (epi) {
    case DEPI_QKV: hipLaunchKernelGGL((dec_gemm_kernel<T, DEPI_QKV>), grid, dim3(512), 0, st, a); break;
    case DEPI_PLAIN: hipLaunchKernelGGL((dec_gemm_kernel<T, DEPI_PLAIN>), grid, dim3(512), 0, st, a); break;
    case DEPI_RESID: hipLaunchKernelGGL((dec_gemm_kernel<T, DEPI_RESID>), grid, dim3(512), 0, st, a); break;
    case DEPI_GATED: hipLaunchKernelGGL((dec_gemm_kernel<T, DEPI_GATED>), grid, dim3(512), 0, st, a); break;
    default: return M2M_ERR_INVALID;
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

static int launch_dec_gemm(int precision, int epi, const DecGemmArgs& a, hipStream_t st) {
  return precision == M2M_PREC_BF16 ? launch_dec_gemm_t<bf16_t>(epi, a, st) : launch_dec_gemm_t<float>(epi, a, st);
}

// ======================================================= decode attention ====
struct DecAttnArgs {
  const float* q;        // [B, inner] fp32
  const void* K;         // [B][H][kv_stride][64] T
  const void* V;
  int kv_stride;         // keys allocated per (b,h): Lmax (self) or S (cross)
  int n_keys;            // cross: S ; self: ignored (t+1 from state unless self_len_override > 0)
  int self_len_override; // bench only
  const float* bias;     // self: [H][Lmax] by n = q_pos - k_pos ; cross: nullptr
  int bias_stride;
  void* out;             // [B, inner] T (input of the output projection)
  int H, inner;
  const DecState* state;
  int is_self;
};

template <typename T>
__global__ __launch_bounds__(1024) void dec_attn_kernel(DecAttnArgs a) {
  constexpr int E = 16 / sizeof(T);      // elements per 16-byte chunk: 8 (bf16) / 4 (fp32)
  constexpr int LPR = DK / E;            // lanes per key row: 8 / 16
  constexpr int KPW = 64 / LPR;          // keys per wave-load: 8 / 4
  constexpr int KPB = 16 * KPW;          // keys per block round: 128 / 64
  extern __shared__ __align__(16) float sm[];
  float* sc = sm;                         // [n_keys] scores -> probabilities
  __shared__ float redw[16];
  __shared__ float redo[16][DK];
  __shared__ float bcast[2];
  if (a.state->done) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.H, hh = blockIdx.x - b * a.H;
  const int t = a.state->t;
  const int n_keys = a.is_self ? (a.self_len_override > 0 ? a.self_len_override : t + 1) : a.n_keys;
  const int qpos = a.is_self ? n_keys - 1 : 0;
  const int sub = lane % LPR;
  const int kslot = wave * KPW + lane / LPR;
  const T* Kb = reinterpret_cast<const T*>(a.K) + ((int64_t)b * a.H + hh) * a.kv_stride * DK + sub * E;
  const T* Vb = reinterpret_cast<const T*>(a.V) + ((int64_t)b * a.H + hh) * a.kv_stride * DK + sub * E;

  float qv[E];
  {
    const float* qp = a.q + (int64_t)b * a.inner + hh * DK + sub * E;
#pragma unroll
    for (int e = 0; e < E; ++e) qv[e] = qp[e];
  }

  // ---- phase 1: scores ----
  for (int k0 = kslot; k0 < n_keys; k0 += 4 * KPB) {
    Vec16<T> kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + u * KPB;
      if (key < n_keys) kv[u].v = *reinterpret_cast<const decltype(kv[u].v)*>(Kb + (int64_t)key * DK);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + u * KPB;
      float d = 0.f;
      if (key < n_keys) {
#pragma unroll
        for (int e = 0; e < E; ++e) d = fmaf(qv[e], kv[u].get(e), d);
      }
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
      if (sub == 0 && key < n_keys) {
        if (a.bias) d += a.bias[(int64_t)hh * a.bias_stride + (qpos - key)];
        sc[key] = d;
      }
    }
  }
  __syncthreads();
  // ---- phase 2: softmax statistics ----
  float mx = -1e30f;
  for (int k = tid; k < n_keys; k += 1024) mx = fmaxf(mx, sc[k]);
  mx = wave_max(mx);
  if (lane == 0) redw[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = redw[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, redw[w]);
    bcast[0] = m;
  }
  __syncthreads();
  mx = bcast[0];
  float sum = 0.f;
  for (int k = tid; k < n_keys; k += 1024) {
    const float p = expf(sc[k] - mx);
    sc[k] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __syncthreads();  // everyone has read bcast[0]/redw before they are rewritten
  if (lane == 0) redw[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += redw[w];
    bcast[1] = s;
  }
  // ---- phase 3: P.V ----
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int k0 = kslot; k0 < n_keys; k0 += 4 * KPB) {
    Vec16<T> vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + u * KPB;
      if (key < n_keys) vv[u].v = *reinterpret_cast<const decltype(vv[u].v)*>(Vb + (int64_t)key * DK);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + u * KPB;
      if (key < n_keys) {
        const float p = sc[key];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = fmaf(p, vv[u].get(e), acc[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane < LPR) {
#pragma unroll
    for (int e = 0; e < E; ++e) redo[wave][lane * E + e] = acc[e];
  }
  __syncthreads();
  if (tid < DK) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += redo[w][tid];
    reinterpret_cast<T*>(a.out)[(int64_t)b * a.inner + hh * DK + tid] = from_f32<T>(s / bcast[1]);
  }
}

static int launch_dec_attn(int precision, const DecAttnArgs& a, int B, int max_keys, hipStream_t st) {
  const size_t smem = (size_t)max_keys * sizeof(float);
  dim3 grid((unsigned)(B * a.H));
  if (precision == M2M_PREC_BF16)
    hipLaunchKernelGGL(dec_attn_kernel<bf16_t>, grid, dim3(1024), smem, st, a);
  else
    hipLaunchKernelGGL(dec_attn_kernel<float>, grid, dim3(1024), smem, st, a);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// ================================================================== head ====
struct DecHeadArgs {
  const float* logits;     // [B, ldl]
  int ldl, V, B, d;
  const float* shared;     // [V, d] embedding
  float* x;                // [B, d] next-step input
  int64_t* tokens;         // [B, max_len] generated ids (col 0 = start)
  int max_len;
  int* finished;           // [B]
  DecState* state;
  int pad_id, eos_id;
  // teacher forcing (nullptr for greedy)
  const int64_t* forced;   // [B, Ld] decoder input ids
  int Ld;
  float* logits_out;       // [B, Ld, V]
};

__global__ __launch_bounds__(1024) void dec_head_kernel(DecHeadArgs a) {
  __shared__ int s_unfinished;
  DecState* stp = a.state;
  if (stp->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = stp->t;
  if (tid == 0) s_unfinished = 0;
  __syncthreads();
  for (int b = wave; b < a.B; b += 16) {
    const float* lg = a.logits + (int64_t)b * a.ldl;
    const int fin = a.forced ? 0 : a.finished[b];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < a.V; v += 64) {
      const float x = lg[v];
      if (x > best || (x == best && v < bi)) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    int next;
    if (a.forced) {
      if (a.logits_out)
        for (int v = lane; v < a.V; v += 64) a.logits_out[((int64_t)b * a.Ld + t) * a.V + v] = lg[v];
      next = (t + 1 < a.Ld) ? (int)a.forced[(int64_t)b * a.Ld + t + 1] : a.pad_id;
    } else {
      // hf generation/utils.py:2925-2937: argmax; finished rows emit pad; EOS finishes a row
      next = fin ? a.pad_id : (bi == 0x7fffffff ? 0 : bi);
      if (lane == 0) {
        if (t + 1 < a.max_len) a.tokens[(int64_t)b * a.max_len + t + 1] = next;
        const int nf = fin | (next == a.eos_id);
        a.finished[b] = nf;
        if (!nf) atomicAdd(&s_unfinished, 1);
      }
    }
    if (next < 0 || next >= a.V) next = a.pad_id;
    const float* emb = a.shared + (int64_t)next * a.d;
    for (int c = lane; c < a.d; c += 64) a.x[(int64_t)b * a.d + c] = emb[c];
  }
  __syncthreads();
  if (tid == 0) {
    const int nt = t + 1;
    stp->t = nt;
    if (a.forced) {
      if (nt >= a.Ld) stp->done = 1;
    } else {
      stp->n_unfinished = s_unfinished;
      if (s_unfinished == 0 || nt >= stp->max_steps) { stp->done = 1; stp->out_len = nt + 1; }
    }
  }
}

__global__ void dec_init_kernel(DecHeadArgs a, int start_id, int max_steps) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nthreads = gridDim.x * blockDim.x;
  if (tid == 0) {
    a.state->t = 0; a.state->done = (max_steps <= 0) ? 1 : 0; a.state->out_len = 1;
    a.state->n_unfinished = a.B; a.state->max_steps = max_steps;
  }
  for (int b = tid; b < a.B; b += nthreads) a.finished[b] = 0;
  if (!a.forced)
    for (int i = tid; i < a.B * a.max_len; i += nthreads) a.tokens[i] = (i % a.max_len == 0) ? start_id : a.pad_id;
  for (int i = tid; i < a.B * a.d; i += nthreads) {
    const int b = i / a.d, c = i - b * a.d;
    int tok = a.forced ? (int)a.forced[(int64_t)b * a.Ld] : start_id;
    if (tok < 0 || tok >= a.V) tok = a.pad_id;
    a.x[i] = a.shared[(int64_t)tok * a.d + c];
  }
}

// ============================================================ step driver ====
// All per-clip buffers are [B][...] with the clip index outermost, so a view is a pointer offset.
static DecHeadArgs head_args(m2m_session* s, const DecView& v, bool forced, float* logits_out, int Ld) {
  const m2m_model* m = s->m;
  DecHeadArgs h{};
  h.logits = s->logits + (int64_t)v.b0 * m->vocab_pad; h.ldl = m->vocab_pad; h.V = m->g.vocab_size; h.B = v.nb;
  h.d = m->g.d_model; h.shared = m->shared; h.x = s->x_dec + (int64_t)v.b0 * m->g.d_model;
  h.tokens = s->tokens + (int64_t)v.b0 * s->max_dec; h.max_len = s->max_dec;
  h.finished = s->finished + v.b0; h.state = v.state; h.pad_id = m->g.pad_token_id; h.eos_id = m->g.eos_token_id;
  h.forced = forced ? s->forced_ids + (int64_t)v.b0 * Ld : nullptr; h.Ld = Ld;
  h.logits_out = logits_out ? logits_out + (int64_t)v.b0 * Ld * m->g.vocab_size : nullptr;
  return h;
}

int decode_init(m2m_session* s, const DecView& v, int max_steps, bool forced, hipStream_t st) {
  DecHeadArgs h = head_args(s, v, forced, nullptr, forced ? max_steps : 0);
  hipLaunchKernelGGL(dec_init_kernel, dim3(32), dim3(256), 0, st, h, s->m->g.decoder_start_token_id, max_steps);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

static size_t kv_layer_elems(const m2m_session* s, int len) {
  return (size_t)s->max_batch * s->m->g.num_heads * len * DK;
}

int decode_launch_attn(m2m_session* s, const DecView& v, bool self, int layer, int self_len, hipStream_t st) {
  const m2m_model* m = s->m;
  const size_t es = m->esize;
  const int H = m->g.num_heads;
  DecAttnArgs a{};
  a.q = s->q_dec + (int64_t)v.b0 * m->inner;
  a.out = (unsigned char*)s->o_dec + (size_t)v.b0 * m->inner * es;
  a.H = H; a.inner = m->inner; a.state = v.state;
  if (self) {
    const size_t off = ((size_t)layer * kv_layer_elems(s, s->max_dec) + (size_t)v.b0 * H * s->max_dec * DK) * es;
    a.K = (const unsigned char*)s->self_k + off;
    a.V = (const unsigned char*)s->self_v + off;
    a.kv_stride = s->max_dec; a.n_keys = 0; a.self_len_override = self_len;
    a.bias = s->dec_bias_tab; a.bias_stride = s->max_dec; a.is_self = 1;
    return launch_dec_attn(m->precision, a, v.nb, s->max_dec, st);
  }
  // cross K/V: [L][2][B][H][S][64] with B, S = the encoded problem
  const size_t per = (size_t)s->B * H * s->S * DK;
  const size_t voff = (size_t)v.b0 * H * s->S * DK;
  a.K = (const unsigned char*)s->cross_kv + (((size_t)layer * 2 + 0) * per + voff) * es;
  a.V = (const unsigned char*)s->cross_kv + (((size_t)layer * 2 + 1) * per + voff) * es;
  a.kv_stride = s->S; a.n_keys = s->S; a.self_len_override = 0; a.bias = nullptr; a.bias_stride = 0; a.is_self = 0;
  return launch_dec_attn(m->precision, a, v.nb, s->S, st);
}

int decode_launch_step(m2m_session* s, const DecView& v, bool forced, float* logits_out, int Ld, hipStream_t st) {
  const m2m_model* m = s->m;
  const m2m_t5_geometry& g = m->g;
  const int P = m->precision;
  const size_t es = m->esize;
  const int H = g.num_heads;
  float* x = s->x_dec + (int64_t)v.b0 * g.d_model;
  float* q = s->q_dec + (int64_t)v.b0 * m->inner;
  unsigned char* o = (unsigned char*)s->o_dec + (size_t)v.b0 * m->inner * es;
  unsigned char* gg = (unsigned char*)s->g_dec + (size_t)v.b0 * g.d_ff * es;
  int rc;
  for (int l = 0; l < g.num_decoder_layers; ++l) {
    const DecLayerPacked& L = m->dec[l];
    DecGemmArgs a{};
    a.eps = g.layer_norm_eps; a.B = v.nb; a.state = v.state; a.H = H; a.Lmax = s->max_dec; a.inner = m->inner;
    // 1. RMSNorm + fused QKV projection, K/V appended to the cache at slot t
    a.x = x; a.ldx = g.d_model; a.ln_w = L.ln0; a.W = L.wqkv; a.K = g.d_model; a.N = 3 * m->inner;
    a.out = q; a.ldo = m->inner;
    const size_t koff = ((size_t)l * kv_layer_elems(s, s->max_dec) + (size_t)v.b0 * H * s->max_dec * DK) * es;
    a.kcache = (unsigned char*)s->self_k + koff;
    a.vcache = (unsigned char*)s->self_v + koff;
    if ((rc = launch_dec_gemm(P, DEPI_QKV, a, st))) return rc;
    // 2. causal self-attention over t+1 cached keys
    if ((rc = decode_launch_attn(s, v, true, l, 0, st))) return rc;
    // 3. output projection + residual
    a.x = o; a.ldx = m->inner; a.ln_w = nullptr; a.W = L.wo; a.K = m->inner; a.N = g.d_model;
    a.out = x; a.ldo = g.d_model;
    if ((rc = launch_dec_gemm(P, DEPI_RESID, a, st))) return rc;
    // 4. RMSNorm + cross-attention query projection
    a.x = x; a.ldx = g.d_model; a.ln_w = L.ln1; a.W = L.wcq; a.K = g.d_model; a.N = m->inner;
    a.out = q; a.ldo = m->inner;
    if ((rc = launch_dec_gemm(P, DEPI_PLAIN, a, st))) return rc;
    // 5. cross-attention over the S encoder positions (K/V projected once in m2m_encode)
    if ((rc = decode_launch_attn(s, v, false, l, 0, st))) return rc;
    // 6. output projection + residual
    a.x = o; a.ldx = m->inner; a.ln_w = nullptr; a.W = L.wco; a.K = m->inner; a.N = g.d_model;
    a.out = x; a.ldo = g.d_model;
    if ((rc = launch_dec_gemm(P, DEPI_RESID, a, st))) return rc;
    // 7. RMSNorm + gated-GELU up projection
    a.x = x; a.ldx = g.d_model; a.ln_w = L.ln2; a.W = L.wi; a.K = g.d_model; a.N = 2 * g.d_ff;
    a.out = gg; a.ldo = g.d_ff;
    if ((rc = launch_dec_gemm(P, DEPI_GATED, a, st))) return rc;
    // 8. down projection + residual
    a.x = gg; a.ldx = g.d_ff; a.ln_w = nullptr; a.W = L.wo_ff; a.K = g.d_ff; a.N = g.d_model;
    a.out = x; a.ldo = g.d_model;
    if ((rc = launch_dec_gemm(P, DEPI_RESID, a, st))) return rc;
  }
  // final RMSNorm + lm_head (untied, no d_model**-0.5 scaling: transformers 4.34 semantics)
  DecGemmArgs a{};
  a.eps = g.layer_norm_eps; a.B = v.nb; a.state = v.state; a.H = H; a.Lmax = s->max_dec; a.inner = m->inner;
  a.x = x; a.ldx = g.d_model; a.ln_w = m->dec_final_ln; a.W = m->lm_head; a.K = g.d_model; a.N = g.vocab_size;
  a.out = s->logits + (int64_t)v.b0 * m->vocab_pad; a.ldo = m->vocab_pad;
  if ((rc = launch_dec_gemm(P, DEPI_PLAIN, a, st))) return rc;
  DecHeadArgs h = head_args(s, v, forced, logits_out, Ld);
  hipLaunchKernelGGL(dec_head_kernel, dim3(1), dim3(1024), 0, st, h);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

}  // namespace m2m
