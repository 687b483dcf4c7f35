// Internal structures of the T5 encoder/decoder path (not part of the C ABI).
#pragma once

#include "common.h"

#include <vector>

namespace m2m {

constexpr int DK = 64;  // d_kv: the attention kernels are specialised for 64 (HF default the reference uses)

// ------------------------------------------------------------------ model ---
struct EncLayerPacked {
  const float* ln0;   // [d]
  const void* wqkv;   // [3*inner, d]      rows: q | k | v
  const void* wo;     // [d, inner]
  const float* ln1;   // [d]
  const void* wi;     // [2*dff, d]        64-row chunks: 32 rows of wi_0 then the matching 32 rows of wi_1
  const void* wo_ff;  // [d, dff]
};

struct DecLayerPacked {
  const float* ln0;
  const void* wqkv;   // [3*inner, d]
  const void* wo;     // [d, inner]
  const float* ln1;
  const void* wcq;    // [inner, d]
  const void* wco;    // [d, inner]
  const float* ln2;
  const void* wi;     // [2*dff, d]        16-row chunks: 8 rows of wi_0 then the matching 8 rows of wi_1
  const void* wo_ff;  // [d, dff]
};

}  // namespace m2m

struct m2m_model {
  m2m_t5_geometry g;
  int precision;
  int inner;            // num_heads * d_kv
  int vocab_pad;        // vocab rounded up to 16 (lm_head rows beyond vocab are zero)
  size_t esize;         // sizeof storage element (4 or 2)
  void* blob = nullptr; // one device allocation holding every packed tensor
  int64_t blob_bytes = 0;
  std::vector<m2m::EncLayerPacked> enc;
  std::vector<m2m::DecLayerPacked> dec;
  const float* enc_final_ln = nullptr;
  const float* dec_final_ln = nullptr;
  const float* shared = nullptr;     // [V, d] fp32 (feeds the fp32 residual stream)
  const void* lm_head = nullptr;     // [vocab_pad, d]
  const void* wckv = nullptr;        // [L_dec * 2 * inner, d]  per layer: ck rows then cv rows
  std::vector<float> enc_rel_bias_host;  // [num_buckets, H]
  std::vector<float> dec_rel_bias_host;  // [num_buckets, H]
};

namespace m2m {

// device-resident decode loop state
struct DecState {
  int t;             // position of the token being fed this step (0-based)
  int done;          // every row has emitted EOS
  int out_len;       // valid columns of the token matrix once done
  int n_unfinished;
  int max_steps;     // steps after which the loop must stop (max_length - 1)
  int overflow;      // sticky: a value outside the fixed-point residual range (|v| >= 2^21, Inf, NaN) was produced
  int t_copy;        // headless greedy loop: t of the current step, rewritten every step by the layer-0 cross kernel (the lm_head
                     // kernel, which ADVANCES t at its end, reads this stable copy instead of t itself)
  int zero;          // always 0 (cleared with the rest of the state, never written): what a "never skip" row flag points at
};

}  // namespace m2m

namespace m2m {
// Environment switches of the encoder-side kernel families (enc_kernels.hip), read ONCE when a session is created: they choose
// between kernels that are not all bit-identical (the two attention forms differ in the last bits), so a session keeps the forms
// it started with whatever the environment does later, and no launch calls getenv (not safe against a concurrent setenv).
// The public entry points of a session put theirs in scope (EncSwitchScope); launches outside any scope (tools, the trainer's
// products) read the environment themselves, as before.
struct EncSwitches {
  int norm_gemm = 0;            // M2M_NORM_GEMM: 0 by size, 1 never ("0"), 2 always ("force")
  unsigned norm_gemm_skip = 0;  // "e<digits>": not for the listed epilogue ids (diagnostic)
  int resid_panel = 0;          // M2M_RESID_PANEL: as norm_gemm
  int attn_wide = 1;            // M2M_ATTN_WIDE=0: the first attention kernel in the bf16 mode as well
  int min_blocks = 160;         // M2M_NORM_GEMM_MIN_BLOCKS: row blocks from which the row-panel kernels run
  int norm_gemm_hout = 0;       // M2M_NORM_GEMM_HOUT (diagnostic)
  int norm_gemm_split = 1;      // M2M_NORM_GEMM_SPLIT=0: below min_blocks the two-kernel path instead of the column-group form (round 6)
};
EncSwitches read_enc_switches();
extern thread_local const EncSwitches* tl_enc_switches;
struct EncSwitchScope {
  const EncSwitches* prev;
  explicit EncSwitchScope(const EncSwitches* s) : prev(tl_enc_switches) { tl_enc_switches = s; }
  ~EncSwitchScope() { tl_enc_switches = prev; }
};
inline EncSwitches enc_switches_now() { return tl_enc_switches ? *tl_enc_switches : read_enc_switches(); }

constexpr int MAX_GROUPS = 8;

// A contiguous range of clips decoded as one independent chain (own step counter/stream/graph).
struct DecView {
  int b0, nb;
  DecState* state;          // device
};

struct DecGroup {
  DecView view{0, 0, nullptr};
  DecState* state_host = nullptr;   // pinned
  hipStream_t stream = nullptr;
  hipEvent_t ev_done = nullptr;
  hipGraphExec_t graph_exec = nullptr;           // the graph of the current view (an entry of `graphs`)
  // captured graphs by (B, S, b0, nb, steps per graph, finished-row skip on / off) — all baked into the launches.  Re-packing the live
  // rows changes (b0, nb) several times per batch, and the next batch starts from the full views again: a small cache instead of
  // a re-capture (~1 ms per 8-step graph) at every change
  struct GraphEntry { int key[6]; hipGraph_t graph; hipGraphExec_t exec; unsigned long long used; };
  std::vector<GraphEntry> graphs;
  unsigned long long graph_clock = 0;
};
}  // namespace m2m

struct m2m_session {
  const m2m_model* m;
  int max_batch, max_enc, max_dec;
  unsigned char* ws;       // caller-owned workspace
  int64_t ws_bytes;
  // carved buffers
  float* x_enc;            // [B*S, d] fp32 residual stream
  void* h_enc;             // [B*S, d] T normalised activations
  void* qkv_enc;           // [3][B][H][S][64] T  (the V third is unused: V is written transposed)
  void* vt_enc;            // [B][H][64][Sp] T, Sp = S rounded up to 64
  void* attn_enc;          // [B*S, inner] T
  void* mid_enc;           // [B*S, dff] T
  float* enc_bias_tab;     // [H][2*max_enc-1]
  int enc_bias_far, dec_bias_far;   // AttnArgs::bias_far of the two tables (0 when the table never becomes constant inside its range)
  float* dec_bias_tab;     // [H][max_dec]
  float* dec_bias_full_tab;// [H][2*max_dec-1] the same bias by (key - query) + max_dec - 1 (batched causal pass)
  void* cross_vt;          // [B][H][64][Sp] T scratch: one layer's cross V transposed (batched pass)
  void* cross_kv;          // [L][2][B][H][S][64] T
  void* self_k;            // [L][B][H][max_dec][64] T
  void* self_v;
  void* x_dec;             // [32-row padded B, d] int64 fixed-point residual stream of the decoder (decode.hip xq_t)
  float* logits;           // [B, vocab_pad]
  int64_t* tokens;         // [B, max_dec]
  int* finished;           // [B]
  int* tok_row;            // [B] slot -> clip (row of `tokens`) of the greedy loop: identity until live rows are re-packed (decode.hip)
  unsigned long long* keys;// [B] headless greedy loop: packed (logit, vocabulary index) arg-max keys of the previous step
  m2m::DecState* states;   // device [MAX_GROUPS]
  int64_t* forced_ids;     // [B, max_dec]
  // current problem
  int B = 0, S = 0;
  m2m::EncSwitches enc_sw;  // encoder-side kernel switches, latched at creation
  int attn_clips = 0;      // decode attention: clips per workgroup forced by M2M_DA_CLIPS when the session was created (0: by chain size)
  int attn_clips_self = 0; // M2M_DA_CLIPS_SELF (diagnostic): clips per self-attention workgroup when it should differ from the cross kernels
  int ff_rows = 0;         // decode feed-forward: residual rows per workgroup forced by M2M_DEC_FF_ROWS (0: by chain size)
  int ff_slices = 0;       // decode feed-forward: hidden slices per workgroup forced by M2M_DEC_FF_SLICES (0: by chain size)
  int repacks = 0, rows_moved = 0;   // live-row re-packings / rows moved by them in the last m2m_generate_greedy
  bool encoded = false;
  // decode chains
  hipEvent_t ev_in = nullptr;
  m2m::DecGroup groups[m2m::MAX_GROUPS];
};

namespace m2m {

// -------------------------------------------------------- launch helpers ---
// encoder-side (enc_kernels.hip)
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_GATED = 2, EPI_HEADS = 3, EPI_STORE_F32 = 4, EPI_GATED16 = 5, EPI_GATED_TRAIN = 6, EPI_GATED_BWD = 7 };

struct GemmArgs {
  const void* A;     // [M, K] T row-major
  const void* W;     // [N, K] T row-major (HF [out, in])
  int M, N, K;
  void* out;         // EPI_STORE / EPI_GATED / EPI_HEADS: T ; EPI_RESID: float (in-place +=)
  int ldo;           // leading dimension of out (STORE/RESID/GATED)
  // EPI_HEADS: out[(which*B + b)*H + h][s][64], which = n / inner
  int Bsz, S, H, inner;
  // EPI_HEADS, optional: the projection `vt_which` (V of the encoder) is written TRANSPOSED instead,
  // vt_out[(b*H + h)*64 + d][s] with row pitch Sp, so the attention kernel stages V^T with 16-byte copies
  int vt_which;      // -1: none
  void* vt_out;
  int Sp;
  int ng;            // column tiles per group of the XCD-aware tile order (set by launch_gemm)
  // EPI_RESID extras (training path; zero = the inference behaviour out += acc):
  const float* resid;      // out = resid + acc instead of out += acc (same indexing as out)
  uint32_t drop_thresh;    // dropout on acc before the add: keep(i) * drop_scale * acc, i = row * ldo + col (common.h drop_keep)
  float drop_scale;
  uint64_t drop_key;       // the site's salt: key = splitmix64(*drop_step + drop_key) (common.h DropKey)
  const uint64_t* drop_step;
  // EPI_GATED_TRAIN (training forward of the gated feed-forward, bf16, weights row-interleaved in 32-row chunks like EPI_GATED): out
  // = mid [M][N/2] = dropout(gelu_new(a) * b) as EPI_GATED writes it, AND the gate pair itself for the backward pass,
  // ab_out [M][N] = [a | b] in the MASTER column order (column c of wi_0 at c, of wi_1 at N/2 + c); dropout via the drop_* fields
  // with element index row * (N/2) + c.  gemm_takes_gated_train() says whether a shape takes this path.
  void* ab_out;
  // EPI_GATED_BWD (training backward of the gated feed-forward, bf16, small tiles): the product is dmid = dy . Wo [M][N = d_ff]; the
  // epilogue turns each tile straight into the gate pair's gradient, out = dab [M][2N] = [dmid~ * b * gelu'(a) | dmid~ * gelu(a)]
  // (ldo = 2N) with dmid~ = dropout mask of the forward applied to bf16(dmid) and a | b read from ab_out [M][2N]; dmid itself is
  // never stored.
  // launch_norm_gemm (the product's A operand is RMSNorm(nx) over the K = d_model wide fp32 residual rows; `A` is then only the scratch the
  // two-kernel fallback normalises into): nx [M][K] fp32, nw [K] the norm weight, neps; h_out (optional) [M][K] T receives the
  // normalised rows as well (the training pass keeps them for the weight gradients).
  const float* nx;
  const float* nw;
  float neps;
  void* h_out;
};

int launch_gemm(int precision, int epi, const GemmArgs& a, hipStream_t st);
// out = epilogue(RMSNorm(a.nx; a.nw, a.neps) . W^T): one kernel in the bf16 mode (row panel normalised once into LDS and kept there for
// every column tile), rmsnorm_kernel into a.A + launch_gemm otherwise (fp32 parity mode, K beyond the LDS panel, M2M_NORM_GEMM=0).
// Epilogues: EPI_HEADS, EPI_GATED, EPI_GATED16, EPI_STORE_F32.  Bit-identical to the two-kernel path (tests/test_t5_gpu.py).
int launch_norm_gemm(int precision, int epi, const GemmArgs& a, hipStream_t st);
bool gemm_takes_gated_train(int precision, int M, int N, int K);
bool gemm_takes_gated_bwd(int precision, int M, int N, int K);
int launch_rmsnorm(int precision, const float* x, const float* w, void* out, int M, int d, float eps, hipStream_t st);
// flash attention of the encoder and of the batched (teacher-forced) decoder pass
struct AttnArgs {
  const void* Q;          // [B*H][Sq][64] T
  const void* K;          // [B*H][Sk][64] T
  const void* Vt;         // [B*H][64][Sp] T (V transposed, row pitch Sp >= Sk rounded up to the 64-key tile)
  int Sp;
  const float* bias_tab;  // [H][tab_stride] relative-position bias by (key - query) + tab_center, or nullptr (cross-attention)
  int tab_stride, tab_center;
  void* out;              // [B*Sq][H*64] T
  int B, H, Sq, Sk;
  int bias_far;           // the table is constant from (key - query) <= -bias_far down and from >= +bias_far up (T5's last bucket of
                          // either direction); 0 = not known.  attn_wide_kernel reads no table for tiles that far from their queries
};
int launch_attn(int precision, const AttnArgs& a, bool causal, hipStream_t st);
int launch_transpose_v(int precision, const void* v, void* vt, int BH, int S, int Sp, hipStream_t st);
int launch_embed_rows(const int64_t* ids, const float* table, float* x, int M, int d, int V, int pad_id, hipStream_t st);
int launch_final_norm_f32(const float* x, const float* w, float* out_f32, void* out_T, int precision, int M, int d,
                          float eps, hipStream_t st);

// repack (repack.hip)
int launch_convert(int precision, const float* src, void* dst, int64_t n, hipStream_t st);
int launch_interleave(int precision, const float* wi0, const float* wi1, void* dst, int dff, int d, int half,
                      hipStream_t st);
int launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st);
int launch_fill_zero(void* dst, int64_t bytes, hipStream_t st);
int launch_checksum(const void* buf, int64_t bytes, unsigned long long* acc_dev, hipStream_t st);

// decoder-side (decode.hip)
int decode_init(m2m_session* s, const DecView& v, int max_steps, bool forced, hipStream_t st);
int decode_launch_step(m2m_session* s, const DecView& v, bool forced, float* logits_out, int Ld, hipStream_t st);
bool decode_finished_skip_on();
int decode_launch_attn(m2m_session* s, const DecView& v, bool self, int layer, int self_len, hipStream_t st, bool headless = false,
                       bool skip_finished = false);
int decode_move_rows(m2m_session* s, const int* src, const int* dst, int n, int t, hipStream_t st);   // live-row re-packing (decode.hip)
int decode_finalize(m2m_session* s, const DecView& v, hipStream_t st);   // headless greedy loop: write the last token, close the chain
bool decode_headless();
int decode_attn_clips(const m2m_session* s, int nb);
int decode_ff_rows(const m2m_session* s, int nb);
int decode_ff_slices(const m2m_session* s, int nb);

}  // namespace m2m
