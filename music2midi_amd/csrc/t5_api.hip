// C ABI of the T5 encoder/decoder path: model (repacked weights), session (workspace +
// captured decode-step graph), encode, greedy generate, teacher-forced decode, bench hooks.
#include "t5.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace m2m;

// ---------------------------------------------------------------- buckets ---
// T5 relative-position bucket (hf: models/t5/modeling_t5.py:217-262).  HF evaluates the log
// branch in float32 and truncates; the same float32 expression is used here and the result
// is pinned against the integer table of SURVEY.md §8a-A6 by tests/test_host_logic.py.
extern "C" int m2m_rel_bucket(int rel, int bidirectional, int num_buckets, int max_distance) {
  int bucket = 0;
  int nb = num_buckets;
  int n;
  if (bidirectional) {
    nb /= 2;
    if (rel > 0) bucket += nb;
    n = rel < 0 ? -rel : rel;
  } else {
    n = rel < 0 ? -rel : 0;
  }
  const int max_exact = nb / 2;
  if (n < max_exact) return bucket + n;
  const float ratio = logf((float)n / (float)max_exact) / (float)log((double)max_distance / (double)max_exact);
  int large = max_exact + (int)(ratio * (float)(nb - max_exact));
  if (large > nb - 1) large = nb - 1;
  return bucket + large;
}

// ------------------------------------------------------------------ model ---
extern "C" int m2m_model_create(const m2m_t5_geometry* geom, const m2m_t5_weights* w, int precision, void* stream,
                                m2m_model** out) {
  M2M_REQUIRE(geom && w && out, "m2m_model_create: null argument");
  M2M_REQUIRE(precision == M2M_PREC_FP32 || precision == M2M_PREC_BF16, "m2m_model_create: bad precision %d", precision);
  const m2m_t5_geometry& g = *geom;
  M2M_REQUIRE(g.d_kv == DK, "m2m_model_create: d_kv=%d unsupported (attention kernels are specialised for 64)", g.d_kv);
  M2M_REQUIRE(g.d_model == 128 || g.d_model == 256 || g.d_model == 384 || g.d_model == 512,
              "m2m_model_create: d_model=%d must be 128, 256, 384 or 512 (the reduction lengths the decode kernels are "
              "instantiated for; the fused decode attention projects with 2 lanes per output column of a 1024-thread workgroup)",
              g.d_model);
  M2M_REQUIRE(g.d_ff >= 64 && g.d_ff % 64 == 0, "m2m_model_create: d_ff=%d must be a multiple of 64", g.d_ff);
  M2M_REQUIRE((g.num_heads * g.d_kv) % 128 == 0 && g.num_heads * g.d_kv <= 1152, "m2m_model_create: num_heads*d_kv=%d must be a multiple of 128, <= 1152", g.num_heads * g.d_kv);
  M2M_REQUIRE(g.num_heads >= 1 && g.num_layers >= 1 && g.num_decoder_layers >= 1 && g.vocab_size >= 2,
              "m2m_model_create: bad geometry");
  M2M_REQUIRE(g.num_buckets >= 4 && g.num_buckets % 2 == 0 && g.max_distance > g.num_buckets / 2,
              "m2m_model_create: bad relative-attention geometry");
  M2M_REQUIRE(w->shared && w->lm_head && w->enc_rel_bias && w->dec_rel_bias && w->enc_final_ln && w->dec_final_ln &&
                  w->enc && w->dec, "m2m_model_create: null weight pointer");
  hipStream_t st = (hipStream_t)stream;
  const int d = g.d_model, dff = g.d_ff, inner = g.num_heads * g.d_kv, V = g.vocab_size;
  const int Le = g.num_layers, Ld = g.num_decoder_layers;
  const size_t es = precision == M2M_PREC_BF16 ? 2 : 4;
  const int vocab_pad = ceil_div(V, 16) * 16;

  // ---- carve one blob ----
  int64_t off = 0;
  auto take = [&](int64_t bytes) { int64_t o = off; off = align_up(off + bytes, 256); return o; };
  struct EncOff { int64_t ln0, wqkv, wo, ln1, wi, wo_ff; };
  struct DecOff { int64_t ln0, wqkv, wo, ln1, wcq, wco, ln2, wi, wo_ff; };
  std::vector<EncOff> eo(Le);
  std::vector<DecOff> dof(Ld);
  for (int l = 0; l < Le; ++l) {
    eo[l].ln0 = take(d * 4); eo[l].wqkv = take((int64_t)3 * inner * d * es); eo[l].wo = take((int64_t)d * inner * es);
    eo[l].ln1 = take(d * 4); eo[l].wi = take((int64_t)2 * dff * d * es); eo[l].wo_ff = take((int64_t)d * dff * es);
  }
  for (int l = 0; l < Ld; ++l) {
    dof[l].ln0 = take(d * 4); dof[l].wqkv = take((int64_t)3 * inner * d * es); dof[l].wo = take((int64_t)d * inner * es);
    dof[l].ln1 = take(d * 4); dof[l].wcq = take((int64_t)inner * d * es); dof[l].wco = take((int64_t)d * inner * es);
    dof[l].ln2 = take(d * 4); dof[l].wi = take((int64_t)2 * dff * d * es); dof[l].wo_ff = take((int64_t)d * dff * es);
  }
  const int64_t o_eln = take(d * 4), o_dln = take(d * 4);
  const int64_t o_shared = take((int64_t)V * d * 4);
  const int64_t o_head = take((int64_t)vocab_pad * d * es);
  const int64_t o_ckv = take((int64_t)Ld * 2 * inner * d * es);

  m2m_model* m = new m2m_model();
  m->g = g; m->precision = precision; m->inner = inner; m->vocab_pad = vocab_pad; m->esize = es; m->blob_bytes = off;
  hipError_t e = hipMalloc(&m->blob, (size_t)off);
  if (e != hipSuccess) {
    set_error("m2m_model_create: hipMalloc(%lld) failed: %s", (long long)off, hipGetErrorString(e));
    delete m;
    return M2M_ERR_NOMEM;
  }
  unsigned char* base = (unsigned char*)m->blob;
  int rc = M2M_OK;
  // the alignment gaps between the packed tensors are part of what m2m_model_checksum covers: they must not be allocator garbage
  if (hipMemsetAsync(m->blob, 0, (size_t)off, st) != hipSuccess) { set_error("m2m_model_create: hipMemsetAsync failed"); rc = M2M_ERR_HIP; }
  auto conv = [&](const float* src, int64_t o, int64_t n) {
    if (rc == M2M_OK && !src) { set_error("m2m_model_create: null layer weight pointer"); rc = M2M_ERR_INVALID; }
    if (rc == M2M_OK) rc = launch_convert(precision, src, base + o, n, st);
  };
  auto copyf = [&](const float* src, int64_t o, int64_t n) {
    if (rc == M2M_OK && !src) { set_error("m2m_model_create: null layer weight pointer"); rc = M2M_ERR_INVALID; }
    if (rc == M2M_OK) rc = launch_copy_f32(src, (float*)(base + o), n, st);
  };
  m->enc.resize(Le); m->dec.resize(Ld);
  for (int l = 0; l < Le && rc == M2M_OK; ++l) {
    const m2m_enc_layer_weights& s = w->enc[l];
    copyf(s.ln0, eo[l].ln0, d); copyf(s.ln1, eo[l].ln1, d);
    conv(s.q, eo[l].wqkv, (int64_t)inner * d);
    conv(s.k, eo[l].wqkv + (int64_t)inner * d * es, (int64_t)inner * d);
    conv(s.v, eo[l].wqkv + (int64_t)2 * inner * d * es, (int64_t)inner * d);
    conv(s.o, eo[l].wo, (int64_t)d * inner);
    if (rc == M2M_OK && !(s.wi0 && s.wi1)) { set_error("m2m_model_create: null wi pointer"); rc = M2M_ERR_INVALID; }
    if (rc == M2M_OK) rc = launch_interleave(precision, s.wi0, s.wi1, base + eo[l].wi, dff, d, 32, st);
    conv(s.wo, eo[l].wo_ff, (int64_t)d * dff);
    m->enc[l] = {(const float*)(base + eo[l].ln0), base + eo[l].wqkv, base + eo[l].wo, (const float*)(base + eo[l].ln1),
                 base + eo[l].wi, base + eo[l].wo_ff};
  }
  for (int l = 0; l < Ld && rc == M2M_OK; ++l) {
    const m2m_dec_layer_weights& s = w->dec[l];
    copyf(s.ln0, dof[l].ln0, d); copyf(s.ln1, dof[l].ln1, d); copyf(s.ln2, dof[l].ln2, d);
    conv(s.q, dof[l].wqkv, (int64_t)inner * d);
    conv(s.k, dof[l].wqkv + (int64_t)inner * d * es, (int64_t)inner * d);
    conv(s.v, dof[l].wqkv + (int64_t)2 * inner * d * es, (int64_t)inner * d);
    conv(s.o, dof[l].wo, (int64_t)d * inner);
    conv(s.cq, dof[l].wcq, (int64_t)inner * d);
    conv(s.co, dof[l].wco, (int64_t)d * inner);
    if (rc == M2M_OK && !(s.wi0 && s.wi1)) { set_error("m2m_model_create: null wi pointer"); rc = M2M_ERR_INVALID; }
    if (rc == M2M_OK) rc = launch_interleave(precision, s.wi0, s.wi1, base + dof[l].wi, dff, d, 8, st);
    conv(s.wo, dof[l].wo_ff, (int64_t)d * dff);
    conv(s.ck, o_ckv + (int64_t)(l * 2 + 0) * inner * d * es, (int64_t)inner * d);
    conv(s.cv, o_ckv + (int64_t)(l * 2 + 1) * inner * d * es, (int64_t)inner * d);
    m->dec[l] = {(const float*)(base + dof[l].ln0), base + dof[l].wqkv, base + dof[l].wo,
                 (const float*)(base + dof[l].ln1), base + dof[l].wcq, base + dof[l].wco,
                 (const float*)(base + dof[l].ln2), base + dof[l].wi, base + dof[l].wo_ff};
  }
  copyf(w->enc_final_ln, o_eln, d); copyf(w->dec_final_ln, o_dln, d);
  copyf(w->shared, o_shared, (int64_t)V * d);
  if (rc == M2M_OK) rc = launch_fill_zero(base + o_head, (int64_t)vocab_pad * d * es, st);
  conv(w->lm_head, o_head, (int64_t)V * d);
  m->enc_final_ln = (const float*)(base + o_eln); m->dec_final_ln = (const float*)(base + o_dln);
  m->shared = (const float*)(base + o_shared); m->lm_head = base + o_head; m->wckv = base + o_ckv;
  m->enc_rel_bias_host.resize((size_t)g.num_buckets * g.num_heads);
  m->dec_rel_bias_host.resize((size_t)g.num_buckets * g.num_heads);
  hipError_t he = hipSuccess;
  if (rc == M2M_OK) he = hipMemcpyAsync(m->enc_rel_bias_host.data(), w->enc_rel_bias, m->enc_rel_bias_host.size() * 4, hipMemcpyDeviceToHost, st);
  if (rc == M2M_OK && he == hipSuccess) he = hipMemcpyAsync(m->dec_rel_bias_host.data(), w->dec_rel_bias, m->dec_rel_bias_host.size() * 4, hipMemcpyDeviceToHost, st);
  if (rc == M2M_OK && he == hipSuccess) he = hipStreamSynchronize(st);
  if (rc == M2M_OK && he != hipSuccess) { set_error("m2m_model_create: %s", hipGetErrorString(he)); rc = M2M_ERR_HIP; }
  if (rc != M2M_OK) {
    (void)hipFree(m->blob);
    delete m;
    return rc;
  }
  *out = m;
  return M2M_OK;
}

extern "C" void m2m_model_destroy(m2m_model* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}
extern "C" int m2m_model_precision(const m2m_model* m) { return m ? m->precision : M2M_ERR_INVALID; }
extern "C" int64_t m2m_model_param_bytes(const m2m_model* m) { return m ? m->blob_bytes : (int64_t)M2M_ERR_INVALID; }

extern "C" int m2m_model_checksum(const m2m_model* m, uint64_t* out_host, void* stream) {
  M2M_REQUIRE(m && out_host, "m2m_model_checksum: null argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* acc = nullptr;
  M2M_CHECK_HIP(hipMalloc((void**)&acc, sizeof(unsigned long long)));       // once per run, never on the hot path
  unsigned long long host = 0;
  hipError_t e = hipMemsetAsync(acc, 0, sizeof(unsigned long long), st);
  int rc = M2M_OK;
  if (e == hipSuccess) rc = launch_checksum(m->blob, m->blob_bytes, acc, st);
  if (e == hipSuccess && rc == M2M_OK) e = hipMemcpyAsync(&host, acc, sizeof(host), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && rc == M2M_OK) e = hipStreamSynchronize(st);
  (void)hipFree(acc);
  if (rc != M2M_OK) return rc;
  if (e != hipSuccess) { set_error("m2m_model_checksum: %s", hipGetErrorString(e)); return M2M_ERR_HIP; }
  *out_host = (uint64_t)host;
  return M2M_OK;
}

// ---------------------------------------------------------------- session ---
namespace {
struct WsLayout {
  int64_t x_enc, h_enc, qkv_enc, vt_enc, attn_enc, mid_enc, enc_bias, dec_bias, dec_bias_full, cross_vt, cross_kv, self_k, self_v;
  int64_t x_dec, logits, tokens, finished, tok_row, keys, state, forced, total;
};

WsLayout ws_layout(const m2m_model* m, int B, int S, int L) {
  const m2m_t5_geometry& g = m->g;
  // the activation buffers serve the encoder (S rows per clip) and the batched teacher-forced decoder pass (L rows)
  const int64_t es = (int64_t)m->esize, M = (int64_t)B * S, Ma = (int64_t)B * (S > L ? S : L), Bp = (int64_t)ceil_div(B, 32) * 32;
  const int64_t Spa = (int64_t)ceil_div(S > L ? S : L, 64) * 64;
  WsLayout w{};
  int64_t off = 0;
  auto take = [&](int64_t bytes) { int64_t o = off; off = align_up(off + bytes, 256); return o; };
  w.x_enc = take(Ma * g.d_model * 4);
  w.h_enc = take(Ma * g.d_model * es);
  w.qkv_enc = take(3 * Ma * m->inner * es);
  w.vt_enc = take((int64_t)B * m->inner * Spa * es);
  w.attn_enc = take(Ma * m->inner * es);
  w.mid_enc = take(Ma * g.d_ff * es);
  w.enc_bias = take((int64_t)g.num_heads * (2 * S - 1) * 4);
  w.dec_bias = take((int64_t)g.num_heads * L * 4);
  w.dec_bias_full = take((int64_t)g.num_heads * (2 * L - 1) * 4);
  w.cross_vt = take((int64_t)B * m->inner * (ceil_div(S, 64) * 64) * es);
  w.cross_kv = take((int64_t)g.num_decoder_layers * 2 * M * m->inner * es);
  w.self_k = take((int64_t)g.num_decoder_layers * B * m->inner * L * es);
  w.self_v = take((int64_t)g.num_decoder_layers * B * m->inner * L * es);
  w.x_dec = take(3 * Bp * g.d_model * 8);   // int64 fixed-point residual stream, 3 rotating buffers
  w.logits = take(Bp * m->vocab_pad * 4);
  w.tokens = take((int64_t)B * L * 8);
  w.finished = take((int64_t)B * 4);
  w.tok_row = take((int64_t)B * 4);
  w.keys = take((int64_t)B * 8);
  w.state = take(sizeof(DecState) * MAX_GROUPS);
  w.forced = take((int64_t)B * L * 8);
  w.total = off;
  return w;
}
}  // namespace

extern "C" int64_t m2m_session_workspace_bytes(const m2m_model* m, int max_batch, int max_enc_len, int max_dec_len) {
  if (!m || max_batch < 1 || max_enc_len < 1 || max_dec_len < 1) {
    set_error("m2m_session_workspace_bytes: bad argument");
    return M2M_ERR_INVALID;
  }
  return ws_layout(m, max_batch, max_enc_len, max_dec_len).total;
}

extern "C" int m2m_session_create(const m2m_model* m, int max_batch, int max_enc_len, int max_dec_len,
                                  void* workspace_dev, int64_t workspace_bytes, m2m_session** out) {
  M2M_REQUIRE(m && workspace_dev && out, "m2m_session_create: null argument");
  M2M_REQUIRE(max_batch >= 1 && max_enc_len >= 1 && max_dec_len >= 1, "m2m_session_create: bad geometry");
  M2M_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, "m2m_session_create: workspace must be 256-byte aligned");
  const WsLayout w = ws_layout(m, max_batch, max_enc_len, max_dec_len);
  if (workspace_bytes < w.total) {
    set_error("m2m_session_create: workspace %lld bytes < required %lld", (long long)workspace_bytes, (long long)w.total);
    return M2M_ERR_NOMEM;
  }
  M2M_REQUIRE((size_t)(2 * max_enc_len - 1) * 4 + 40 * 1024 <= 150 * 1024,
              "m2m_session_create: max_enc_len=%d too long (attention bias table must fit LDS)", max_enc_len);
  M2M_REQUIRE((size_t)max_enc_len * 4 <= 60 * 1024 && (size_t)max_dec_len * 4 <= 60 * 1024,
              "m2m_session_create: sequence too long for the decode attention score buffer");
  // switches of the decode path are read ONCE per session (graphs bake them in; getenv is not safe against a concurrent setenv)
  const char* da_env = getenv("M2M_DA_CLIPS");
  const int da_clips = (da_env && da_env[0]) ? atoi(da_env) : 0;
  M2M_REQUIRE(da_clips == 0 || da_clips == 1 || da_clips == 2 || da_clips == 4,
              "M2M_DA_CLIPS=%d: clips per decode-attention workgroup must be 0 (by chain size), 1, 2 or 4", da_clips);
  const char* ff_env = getenv("M2M_DEC_FF_ROWS");
  const int ff_rows = (ff_env && ff_env[0]) ? atoi(ff_env) : 0;
  M2M_REQUIRE(ff_rows == 0 || ff_rows == 8 || ff_rows == 16, "M2M_DEC_FF_ROWS=%d: rows per decode feed-forward workgroup must be 0 (by chain size), 8 or 16", ff_rows);
  const char* fs_env = getenv("M2M_DEC_FF_SLICES");
  const int ff_slices = (fs_env && fs_env[0]) ? atoi(fs_env) : 0;
  M2M_REQUIRE(ff_slices == 0 || ff_slices == 1 || ff_slices == 2 || ff_slices == 4,
              "M2M_DEC_FF_SLICES=%d: hidden slices per decode feed-forward workgroup must be 0 (by chain size), 1, 2 or 4", ff_slices);
  m2m_session* s = new m2m_session();
  s->m = m; s->max_batch = max_batch; s->max_enc = max_enc_len; s->max_dec = max_dec_len;
  s->ws = (unsigned char*)workspace_dev; s->ws_bytes = workspace_bytes;
  s->attn_clips = da_clips; s->ff_rows = ff_rows; s->ff_slices = ff_slices;
  s->enc_sw = read_enc_switches();
  { const char* v = getenv("M2M_DA_CLIPS_SELF"); const int c = (v && v[0]) ? atoi(v) : 0; s->attn_clips_self = (c == 1 || c == 2 || c == 4) ? c : 0; }
  unsigned char* b = s->ws;
  s->x_enc = (float*)(b + w.x_enc); s->h_enc = b + w.h_enc; s->qkv_enc = b + w.qkv_enc; s->vt_enc = b + w.vt_enc; s->attn_enc = b + w.attn_enc;
  s->mid_enc = b + w.mid_enc; s->enc_bias_tab = (float*)(b + w.enc_bias); s->dec_bias_tab = (float*)(b + w.dec_bias);
  s->cross_kv = b + w.cross_kv; s->self_k = b + w.self_k; s->self_v = b + w.self_v;
  s->dec_bias_full_tab = (float*)(b + w.dec_bias_full); s->cross_vt = b + w.cross_vt;
  s->x_dec = (b + w.x_dec);
  s->logits = (float*)(b + w.logits); s->tokens = (int64_t*)(b + w.tokens);
  s->finished = (int*)(b + w.finished); s->tok_row = (int*)(b + w.tok_row); s->keys = (unsigned long long*)(b + w.keys); s->states = (DecState*)(b + w.state); s->forced_ids = (int64_t*)(b + w.forced);

  // relative-position bias tables (fp32), built on the host from the bucket function
  const m2m_t5_geometry& g = m->g;
  const int H = g.num_heads, S = max_enc_len, L = max_dec_len;
  std::vector<float> et((size_t)H * (2 * S - 1)), dt((size_t)H * L);
  for (int rel = -(S - 1); rel <= S - 1; ++rel) {
    const int bk = m2m_rel_bucket(rel, 1, g.num_buckets, g.max_distance);
    for (int h = 0; h < H; ++h) et[(size_t)h * (2 * S - 1) + rel + S - 1] = m->enc_rel_bias_host[(size_t)bk * H + h];
  }
  for (int n = 0; n < L; ++n) {
    const int bk = m2m_rel_bucket(-n, 0, g.num_buckets, g.max_distance);
    for (int h = 0; h < H; ++h) dt[(size_t)h * L + n] = m->dec_rel_bias_host[(size_t)bk * H + h];
  }
  // distance from which on a table holds one value per side (the last bucket): the smallest D with bucket(-n) == bucket(-(len - 1))
  // and bucket(+n) == bucket(len - 1) for every n >= D; 0 if the table is too short to get there
  auto far_of = [&](int len, int bidir) {
    int d = len - 1;
    while (d > 0 && m2m_rel_bucket(-(d - 1), bidir, g.num_buckets, g.max_distance) == m2m_rel_bucket(-(len - 1), bidir, g.num_buckets, g.max_distance) &&
           m2m_rel_bucket(d - 1, bidir, g.num_buckets, g.max_distance) == m2m_rel_bucket(len - 1, bidir, g.num_buckets, g.max_distance)) --d;
    return d >= len - 1 ? 0 : d;
  };
  s->enc_bias_far = far_of(S, 1);
  s->dec_bias_far = far_of(L, 0);      // (the future half of the causal table is masked, never read by a wide step)
  // the same decoder bias as a (key - query) table for the batched causal pass: entry L-1-n = bias(n), n = q - k >= 0;
  // the upper half (future keys) is masked by the kernel and stays 0
  std::vector<float> df((size_t)H * (2 * L - 1), 0.f);
  for (int h = 0; h < H; ++h)
    for (int n = 0; n < L; ++n) df[(size_t)h * (2 * L - 1) + (L - 1 - n)] = dt[(size_t)h * L + n];
  hipError_t e = hipMemcpy(s->enc_bias_tab, et.data(), et.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(s->dec_bias_tab, dt.data(), dt.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(s->dec_bias_full_tab, df.data(), df.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(s->states, 0, sizeof(DecState) * MAX_GROUPS);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming);
  for (int i = 0; i < MAX_GROUPS && e == hipSuccess; ++i) {
    DecGroup& gr = s->groups[i];
    gr.view.state = s->states + i;
    // M2M_CHAIN_CU_MASK=<n> (diagnostic, round 6): chain i may only use the CUs whose index within their XCD satisfies
    // (cu * n / 32) % n == i % n, i.e. n disjoint slices of every XCD - to see whether two chains of 128-workgroup launches already
    // land side by side on their own (they do: DESIGN 4.3)
    const char* cm = getenv("M2M_CHAIN_CU_MASK");
    const int nmask = (cm && cm[0]) ? atoi(cm) : 0;
    if (nmask >= 2 && nmask <= 8) {
      uint32_t mask[8];                                   // 256 CUs = 8 XCDs x 32; bit index = the runtime's CU numbering
      const int mode = getenv("M2M_CHAIN_CU_MASK_MODE") ? atoi(getenv("M2M_CHAIN_CU_MASK_MODE")) : 0;
      for (int w = 0; w < 8; ++w) {
        mask[w] = 0;
        for (int b = 0; b < 32; ++b) {
          const int cu = w * 32 + b;
          // mode 0: contiguous slices of each 32-CU group; mode 1: interleaved CUs
          const int owner = mode == 0 ? ((b * nmask) / 32) % nmask : cu % nmask;
          if (owner == i % nmask) mask[w] |= 1u << b;
        }
      }
      e = hipExtStreamCreateWithCUMask(&gr.stream, 8, mask);
    } else {
      e = hipStreamCreateWithFlags(&gr.stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&gr.ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&gr.state_host, sizeof(DecState), hipHostMallocDefault);
  }
  if (e != hipSuccess) {
    set_error("m2m_session_create: %s", hipGetErrorString(e));
    m2m_session_destroy(s);
    return M2M_ERR_HIP;
  }
  *out = s;
  return M2M_OK;
}

extern "C" void m2m_session_destroy(m2m_session* s) {
  if (!s) return;
  for (int i = 0; i < MAX_GROUPS; ++i) {
    DecGroup& gr = s->groups[i];
    for (auto& ge : gr.graphs) {
      if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
      if (ge.graph) (void)hipGraphDestroy(ge.graph);
    }
    gr.graphs.clear();
    if (gr.ev_done) (void)hipEventDestroy(gr.ev_done);
    if (gr.stream) (void)hipStreamDestroy(gr.stream);
    if (gr.state_host) (void)hipHostFree(gr.state_host);
  }
  if (s->ev_in) (void)hipEventDestroy(s->ev_in);
  delete s;
}

// ----------------------------------------------------------------- encode ---
extern "C" int m2m_encode(m2m_session* s, const float* inputs_embeds_dev, int B, int S, float* enc_out_dev, void* stream) {
  M2M_REQUIRE(s && inputs_embeds_dev, "m2m_encode: null argument");
  M2M_REQUIRE(B >= 1 && B <= s->max_batch, "m2m_encode: batch %d exceeds session max_batch %d", B, s->max_batch);
  M2M_REQUIRE(S >= 1 && S <= s->max_enc, "m2m_encode: S=%d exceeds session max_enc_len %d", S, s->max_enc);
  const m2m_model* m = s->m;
  const m2m_t5_geometry& g = m->g;
  const int P = m->precision;
  hipStream_t st = (hipStream_t)stream;
  const int M = B * S, d = g.d_model;
  const EncSwitchScope sw_scope(&s->enc_sw);
  s->encoded = false;
  M2M_CHECK_HIP(hipMemcpyAsync(s->x_enc, inputs_embeds_dev, (size_t)M * d * 4, hipMemcpyDeviceToDevice, st));
  int rc;
  for (int l = 0; l < g.num_layers; ++l) {
    const EncLayerPacked& L = m->enc[l];
    // RMSNorm + QKV projection: one kernel in the bf16 mode (launch_norm_gemm), norm into h_enc + GEMM otherwise
    GemmArgs a{};
    const int Sp = ceil_div(S, 64) * 64;
    a.A = s->h_enc; a.W = L.wqkv; a.M = M; a.N = 3 * m->inner; a.K = d; a.out = s->qkv_enc;
    a.Bsz = B; a.S = S; a.H = g.num_heads; a.inner = m->inner;
    a.vt_which = 2; a.vt_out = s->vt_enc; a.Sp = Sp;
    a.nx = s->x_enc; a.nw = L.ln0; a.neps = g.layer_norm_eps;
    if ((rc = launch_norm_gemm(P, EPI_HEADS, a, st))) return rc;
    AttnArgs at{};
    at.Q = s->qkv_enc; at.K = (const unsigned char*)s->qkv_enc + (size_t)M * m->inner * m->esize; at.Vt = s->vt_enc; at.Sp = Sp;
    at.bias_tab = s->enc_bias_tab; at.tab_stride = 2 * s->max_enc - 1; at.tab_center = s->max_enc - 1; at.bias_far = s->enc_bias_far;
    at.out = s->attn_enc; at.B = B; at.H = g.num_heads; at.Sq = S; at.Sk = S;
    if ((rc = launch_attn(P, at, false, st))) return rc;
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->attn_enc; a.W = L.wo; a.M = M; a.N = d; a.K = m->inner; a.out = s->x_enc; a.ldo = d;
    if ((rc = launch_gemm(P, EPI_RESID, a, st))) return rc;
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->h_enc; a.W = L.wi; a.M = M; a.N = 2 * g.d_ff; a.K = d; a.out = s->mid_enc; a.ldo = g.d_ff;
    a.nx = s->x_enc; a.nw = L.ln1; a.neps = g.layer_norm_eps;
    if ((rc = launch_norm_gemm(P, EPI_GATED, a, st))) return rc;
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->mid_enc; a.W = L.wo_ff; a.M = M; a.N = d; a.K = g.d_ff; a.out = s->x_enc; a.ldo = d;
    if ((rc = launch_gemm(P, EPI_RESID, a, st))) return rc;
  }
  // final norm -> GEMM input (T) and, if asked, the fp32 encoder states
  // cross-attention K/V of every decoder layer in one GEMM, written in decode layout
  GemmArgs a{};
  a.vt_which = -1;
  a.A = s->h_enc; a.W = m->wckv; a.M = M; a.N = g.num_decoder_layers * 2 * m->inner; a.K = d; a.out = s->cross_kv;
  a.Bsz = B; a.S = S; a.H = g.num_heads; a.inner = m->inner;
  if (enc_out_dev) {        // the caller wants the fp32 encoder states too: the norm runs as its own kernel (both outputs), then the product
    if ((rc = launch_final_norm_f32(s->x_enc, m->enc_final_ln, enc_out_dev, s->h_enc, P, M, d, g.layer_norm_eps, st))) return rc;
    if ((rc = launch_gemm(P, EPI_HEADS, a, st))) return rc;
  } else {                  // generate(): the final norm is the cross-K/V product's own prologue
    a.nx = s->x_enc; a.nw = m->enc_final_ln; a.neps = g.layer_norm_eps;
    if ((rc = launch_norm_gemm(P, EPI_HEADS, a, st))) return rc;
  }
  s->B = B; s->S = S; s->encoded = true;
  return M2M_OK;
}

// ----------------------------------------------------------------- decode ---
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && v[0]) ? atoi(v) : dflt;
}

static bool use_graph() { return env_int("M2M_NO_GRAPH", 0) != 1; }

// Split the B encoded clips into independent chains: M2M_GROUP_ROWS clips per chain (default 32:
// one 32-row MFMA tile of clips per chain), at most MAX_GROUPS chains.  Measured at B = 32 on
// MI355X: 1 chain 322 ms, 2 chains 315 ms, 4 chains 666 ms (dispatch-bound) - see DESIGN_HISTORY.md 4.4.
static int plan_groups(m2m_session* s, int rows = -1) {       // rows: the packed slots to decode (default: the whole encoded batch)
  // Default: TWO chains once there are enough clips to split (B >= 24), one otherwise.  Two graph chains on two
  // streams overlap one chain's latency phases (prologue, merge tail, feed-forward, lm_head / head) with the other's
  // K/V stream: B = 32 -> 2 x 16 clips is +3.7 % over one chain of 32 (tools/chain_sweep.py); more than two chains do
  // not help (the dependent-dispatch rate of the command processor becomes the limit: 4 x 8 = 1 x 32, 8 x 4 is 4x slower).
  const int nrows = rows < 0 ? s->B : rows;
  const int rows_env = env_int("M2M_GROUP_ROWS", 0);
  int G = rows_env > 0 ? ceil_div(nrows, rows_env) : (nrows >= 24 ? 2 : 1);
  if (G < 1) G = 1;
  if (G > MAX_GROUPS) G = MAX_GROUPS;
  const int base = nrows / G, extra = nrows % G;
  int b0 = 0;
  for (int i = 0; i < G; ++i) {
    s->groups[i].view.b0 = b0;
    s->groups[i].view.nb = base + (i < extra ? 1 : 0);
    b0 += s->groups[i].view.nb;
  }
  return G;
}

// One graph = `steps` consecutive decode steps of one chain (kernels read the step index from
// device memory, so the same graph replays for every position; steps past the end are no-ops).
static int ensure_graph(m2m_session* s, DecGroup& gr, int steps) {
  const int key[6] = {s->B, s->S, gr.view.b0, gr.view.nb, steps, decode_finished_skip_on() ? 1 : 0};
  ++gr.graph_clock;
  for (auto& ge : gr.graphs)
    if (memcmp(key, ge.key, sizeof(key)) == 0) { ge.used = gr.graph_clock; gr.graph_exec = ge.exec; return M2M_OK; }
  constexpr size_t MAX_CACHED = 12;
  if (gr.graphs.size() >= MAX_CACHED) {                      // evict the least recently used entry
    size_t lru = 0;
    for (size_t i = 1; i < gr.graphs.size(); ++i) if (gr.graphs[i].used < gr.graphs[lru].used) lru = i;
    (void)hipGraphExecDestroy(gr.graphs[lru].exec);
    (void)hipGraphDestroy(gr.graphs[lru].graph);
    gr.graphs.erase(gr.graphs.begin() + (long)lru);
  }
  gr.graph_exec = nullptr;
  M2M_CHECK_HIP(hipStreamBeginCapture(gr.stream, hipStreamCaptureModeThreadLocal));
  int rc = M2M_OK;
  for (int i = 0; i < steps && rc == M2M_OK; ++i) rc = decode_launch_step(s, gr.view, false, nullptr, 0, gr.stream);
  hipGraph_t gph = nullptr;
  hipError_t e = hipStreamEndCapture(gr.stream, &gph);
  if (rc != M2M_OK) { if (gph) (void)hipGraphDestroy(gph); return rc; }
  if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return M2M_ERR_HIP; }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, gph, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(gph); set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return M2M_ERR_HIP; }
  DecGroup::GraphEntry ge{};
  memcpy(ge.key, key, sizeof(key));
  ge.graph = gph; ge.exec = ex; ge.used = gr.graph_clock;
  gr.graphs.push_back(ge);
  gr.graph_exec = ex;
  return M2M_OK;
}

// Nothing of a failed call may still be running when it returns: the caller owns the workspace and may free or
// regrow it at once.  Errors are rare, so the blanket synchronisation of every chain stream costs nothing.
static void quiesce(m2m_session* s, hipStream_t caller) {
  for (int i = 0; i < MAX_GROUPS; ++i)
    if (s->groups[i].stream) (void)hipStreamSynchronize(s->groups[i].stream);
  (void)hipStreamSynchronize(caller);
  (void)hipGetLastError();
}

static int generate_greedy_impl(m2m_session* s, int max_length, int64_t* tokens_out_dev, int* out_len_host, hipStream_t caller);

// why a call that needs the encoded state finds none: never encoded, or the last greedy decode re-packed its live rows over it
static const char* encode_missing(const m2m_session* s) {
  return s->rows_moved > 0 ? "re-encode: the last m2m_generate_greedy re-packed its live rows over the encoded state (it consumes the encode "
                             "whenever m2m_session_repack_stats reports rows moved)"
                           : "call m2m_encode first";
}

extern "C" int m2m_generate_greedy(m2m_session* s, int max_length, int64_t* tokens_out_dev, int* out_len_host, void* stream) {
  M2M_REQUIRE(s && tokens_out_dev && out_len_host, "m2m_generate_greedy: null argument");
  if (!s->encoded) { set_error("m2m_generate_greedy: %s", encode_missing(s)); return M2M_ERR_STATE; }
  M2M_REQUIRE(max_length >= 1 && max_length <= s->max_dec, "m2m_generate_greedy: max_length %d outside [1, %d]", max_length, s->max_dec);
  const int rc = generate_greedy_impl(s, max_length, tokens_out_dev, out_len_host, (hipStream_t)stream);
  if (rc != M2M_OK) quiesce(s, (hipStream_t)stream);
  return rc;
}

static int generate_greedy_impl(m2m_session* s, int max_length, int64_t* tokens_out_dev, int* out_len_host, hipStream_t caller) {
  const int steps = max_length - 1;
  int G = plan_groups(s);
  const bool graph = use_graph();
  const int U = env_int("M2M_GRAPH_STEPS", 8) < 1 ? 1 : env_int("M2M_GRAPH_STEPS", 8);   // decode steps per graph
  // Live-row re-packing at the host polls (decode.hip "live-row re-packing"): once a quarter of the packed rows have emitted EOS the
  // live ones are moved into the first slots and smaller chains take over.  M2M_COMPACT=0: rows keep their slots (round 4).
  const bool compact = env_int("M2M_COMPACT", 1) != 0;
  int rc;
  // order every chain after whatever the caller enqueued (encode ran on the caller's stream)
  M2M_CHECK_HIP(hipEventRecord(s->ev_in, caller));
  for (int i = 0; i < G; ++i) {
    DecGroup& gr = s->groups[i];
    M2M_CHECK_HIP(hipStreamWaitEvent(gr.stream, s->ev_in, 0));
    if ((rc = decode_init(s, gr.view, steps, false, gr.stream))) return rc;
    if (graph && (rc = ensure_graph(s, gr, U))) return rc;
    gr.state_host->done = (steps == 0);
  }
  // Launch round-robin over the chains in chunks; after each chunk fetch the loop states so a
  // batch whose rows have all emitted EOS stops early (the kernels themselves turn into
  // no-ops once their chain's state.done is set).
  const int CHUNK = 64;
  int launched = 0;
  int cur_rows = s->B;                 // packed slots still being decoded
  s->repacks = 0; s->rows_moved = 0;
  int out_len = 1;                     // longest finished chain so far (chains retired by a re-packing included)
  bool range_error = false;
  bool all_done = steps == 0;
  std::vector<int> fin_host, mv_src, mv_dst;
  while (!all_done && launched < steps) {
    // the first polls come sooner (after 16, 32, 64, 96, 128 steps, then every 64): many rows of a real batch end within their first
    // tens of tokens, and a poll costs one drained pipeline (~a step) while every step before a re-packing costs the full batch
    const int chunk = compact ? (launched < 32 ? 16 : (launched < 128 ? 32 : CHUNK)) : CHUNK;
    const int n = steps - launched < chunk ? steps - launched : chunk;
    for (int k = 0; k < n; k += (graph ? U : 1)) {
      for (int i = 0; i < G; ++i) {
        DecGroup& gr = s->groups[i];
        if (gr.state_host->done) continue;
        if (graph) M2M_CHECK_HIP(hipGraphLaunch(gr.graph_exec, gr.stream));
        else if ((rc = decode_launch_step(s, gr.view, false, nullptr, 0, gr.stream))) return rc;
      }
    }
    launched += ceil_div(n, graph ? U : 1) * (graph ? U : 1);
    all_done = true;
    for (int i = 0; i < G; ++i) {
      DecGroup& gr = s->groups[i];
      if (gr.state_host->done) continue;
      M2M_CHECK_HIP(hipMemcpyAsync(gr.state_host, gr.view.state, sizeof(DecState), hipMemcpyDeviceToHost, gr.stream));
    }
    for (int i = 0; i < G; ++i) {
      DecGroup& gr = s->groups[i];
      M2M_CHECK_HIP(hipStreamSynchronize(gr.stream));
      if (!gr.state_host->done) all_done = false;
    }
    // ---- re-pack the live rows (every chain is idle here) ----
    if (compact && !all_done && steps - launched >= CHUNK && cur_rows >= 2 && cur_rows < 32000) {
      fin_host.resize((size_t)cur_rows);
      M2M_CHECK_HIP(hipMemcpy(fin_host.data(), s->finished, (size_t)cur_rows * sizeof(int), hipMemcpyDeviceToHost));
      int live = 0;
      for (int b = 0; b < cur_rows; ++b) live += fin_host[(size_t)b] ? 0 : 1;
      if (live >= 1 && (cur_rows - live) * 4 >= cur_rows) {
        int t_cur = -1;
        for (int i = 0; i < G; ++i) {                             // retire the current chains: keep what they report
          const DecState& hs = *s->groups[i].state_host;
          range_error |= hs.overflow != 0;
          if (hs.done) { if (hs.out_len > out_len) out_len = hs.out_len; }
          else t_cur = hs.t;
        }
        M2M_REQUIRE(t_cur >= 0, "m2m_generate_greedy: no running chain at a re-packing point");
        mv_src.clear(); mv_dst.clear();
        for (int hole = 0, tail = live; hole < live; ++hole) {    // finished slots in front take the live rows from behind the packed range
          if (!fin_host[(size_t)hole]) continue;
          while (tail < cur_rows && fin_host[(size_t)tail]) ++tail;
          mv_src.push_back(tail++); mv_dst.push_back(hole);
        }
        hipStream_t st0 = s->groups[0].stream;
        // Moving a row OVERWRITES the finished clip in its destination slot (cross K/V, self K/V, residual row): from here on the
        // session no longer holds the encode of clips 0..B-1 in clip order, so this generate CONSUMES it.  A later
        // m2m_decode_forced / m2m_generate_greedy / m2m_bench_kernel without a new m2m_encode returns M2M_ERR_STATE instead of
        // decoding permuted / duplicated clips.
        if (!mv_src.empty()) s->encoded = false;
        if (!mv_src.empty() && (rc = decode_move_rows(s, mv_src.data(), mv_dst.data(), (int)mv_src.size(), t_cur, st0))) return rc;
        M2M_CHECK_HIP(hipStreamSynchronize(st0));
        s->repacks += 1; s->rows_moved += (int)mv_src.size();
        cur_rows = live;
        G = plan_groups(s, cur_rows);
        for (int i = 0; i < G; ++i) {
          DecGroup& gr = s->groups[i];
          DecState hs{};
          hs.t = t_cur; hs.t_copy = t_cur; hs.done = 0; hs.out_len = 1; hs.n_unfinished = 0; hs.max_steps = steps;
          *gr.state_host = hs;
          M2M_CHECK_HIP(hipMemcpyAsync(gr.view.state, gr.state_host, sizeof(DecState), hipMemcpyHostToDevice, gr.stream));
          if (graph && (rc = ensure_graph(s, gr, U))) return rc;
        }
      }
    }
  }
  // headless loop: the last step's arg-max is still a pending key (no later step consumed it)
  if (steps > 0)
    for (int i = 0; i < G; ++i)
      if ((rc = decode_finalize(s, s->groups[i].view, s->groups[i].stream))) return rc;
  // valid length = the longest chain (one process decoding the whole batch stops when EVERY row has finished)
  for (int i = 0; i < G; ++i) {
    DecGroup& gr = s->groups[i];
    M2M_CHECK_HIP(hipMemcpyAsync(gr.state_host, gr.view.state, sizeof(DecState), hipMemcpyDeviceToHost, gr.stream));
    M2M_CHECK_HIP(hipStreamSynchronize(gr.stream));
    const int l = steps == 0 ? 1 : gr.state_host->out_len;
    if (l > out_len) out_len = l;
    range_error |= gr.state_host->overflow != 0;
  }
  // pack [B, max_dec] -> caller's [B, max_length] on the caller's stream (all chains are idle now)
  M2M_CHECK_HIP(hipMemcpy2DAsync(tokens_out_dev, (size_t)max_length * 8, s->tokens, (size_t)s->max_dec * 8,
                                 (size_t)max_length * 8, (size_t)s->B, hipMemcpyDeviceToDevice, caller));
  M2M_CHECK_HIP(hipStreamSynchronize(caller));
  *out_len_host = out_len;
  if (range_error) {
    set_error("m2m_generate_greedy: a decoder activation left the fixed-point residual range (|x| >= 2^21) or was not finite; "
              "the fp32 reference would produce Inf/NaN logits here - token ids are not valid (check the checkpoint)");
    return M2M_ERR_RANGE;
  }
  return M2M_OK;
}

// Teacher-forced decoder pass over all Ld positions at once (hf: modeling_t5.py:448-509 per block, :898-1066 wrapper):
// the encoder's MFMA GEMMs and flash attention applied to the decoder weights, causal + relative-position bias in
// the self-attention, the cached cross K/V (V transposed once per layer) in the cross-attention.  Same arithmetic as
// Ld KV-cached decode steps (M2M_FORWARD=step runs those instead), ~1/50 of the time at Ld = 1024.
static int forward_batched(m2m_session* s, const int64_t* ids, int Ld, float* logits_out, hipStream_t st) {
  const m2m_model* m = s->m;
  const m2m_t5_geometry& g = m->g;
  const int P = m->precision, B = s->B, S = s->S, H = g.num_heads, d = g.d_model;
  const size_t es = m->esize;
  const int M = B * Ld, Lp = ceil_div(Ld, 64) * 64, Sp = ceil_div(S, 64) * 64;
  int rc;
  if ((rc = launch_embed_rows(ids, m->shared, s->x_enc, M, d, g.vocab_size, g.pad_token_id, st))) return rc;
  unsigned char* qkv = (unsigned char*)s->qkv_enc;
  const size_t per_enc = (size_t)s->B * H * S * DK;                       // elements of one cross K (or V) block
  for (int l = 0; l < g.num_decoder_layers; ++l) {
    const DecLayerPacked& L = m->dec[l];
    // --- causal self-attention
    GemmArgs a{};
    a.A = s->h_enc; a.W = L.wqkv; a.M = M; a.N = 3 * m->inner; a.K = d; a.out = qkv;
    a.Bsz = B; a.S = Ld; a.H = H; a.inner = m->inner; a.vt_which = 2; a.vt_out = s->vt_enc; a.Sp = Lp;
    a.nx = s->x_enc; a.nw = L.ln0; a.neps = g.layer_norm_eps;
    if ((rc = launch_norm_gemm(P, EPI_HEADS, a, st))) return rc;
    AttnArgs at{};
    at.Q = qkv; at.K = qkv + (size_t)M * m->inner * es; at.Vt = s->vt_enc; at.Sp = Lp;
    at.bias_tab = s->dec_bias_full_tab; at.tab_stride = 2 * s->max_dec - 1; at.tab_center = s->max_dec - 1; at.bias_far = s->dec_bias_far;
    at.out = s->attn_enc; at.B = B; at.H = H; at.Sq = Ld; at.Sk = Ld;
    if ((rc = launch_attn(P, at, true, st))) return rc;
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->attn_enc; a.W = L.wo; a.M = M; a.N = d; a.K = m->inner; a.out = s->x_enc; a.ldo = d;
    if ((rc = launch_gemm(P, EPI_RESID, a, st))) return rc;
    // --- cross-attention over the cached encoder K/V
    a = GemmArgs{};
    a.A = s->h_enc; a.W = L.wcq; a.M = M; a.N = m->inner; a.K = d; a.out = qkv;
    a.Bsz = B; a.S = Ld; a.H = H; a.inner = m->inner; a.vt_which = -1;
    a.nx = s->x_enc; a.nw = L.ln1; a.neps = g.layer_norm_eps;
    if ((rc = launch_norm_gemm(P, EPI_HEADS, a, st))) return rc;
    const unsigned char* ck = (const unsigned char*)s->cross_kv + ((size_t)l * 2 + 0) * per_enc * es;
    const unsigned char* cv = (const unsigned char*)s->cross_kv + ((size_t)l * 2 + 1) * per_enc * es;
    if ((rc = launch_transpose_v(P, cv, s->cross_vt, B * H, S, Sp, st))) return rc;
    at = AttnArgs{};
    at.Q = qkv; at.K = ck; at.Vt = s->cross_vt; at.Sp = Sp; at.bias_tab = nullptr;
    at.out = s->attn_enc; at.B = B; at.H = H; at.Sq = Ld; at.Sk = S;
    if ((rc = launch_attn(P, at, false, st))) return rc;
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->attn_enc; a.W = L.wco; a.M = M; a.N = d; a.K = m->inner; a.out = s->x_enc; a.ldo = d;
    if ((rc = launch_gemm(P, EPI_RESID, a, st))) return rc;
    // --- gated feed-forward (the decoder's 16-row interleave of wi)
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->h_enc; a.W = L.wi; a.M = M; a.N = 2 * g.d_ff; a.K = d; a.out = s->mid_enc; a.ldo = g.d_ff;
    a.nx = s->x_enc; a.nw = L.ln2; a.neps = g.layer_norm_eps;
    if ((rc = launch_norm_gemm(P, EPI_GATED16, a, st))) return rc;
    a = GemmArgs{}; a.vt_which = -1;
    a.A = s->mid_enc; a.W = L.wo_ff; a.M = M; a.N = d; a.K = g.d_ff; a.out = s->x_enc; a.ldo = d;
    if ((rc = launch_gemm(P, EPI_RESID, a, st))) return rc;
  }
  GemmArgs a{};
  a.vt_which = -1;
  a.A = s->h_enc; a.W = m->lm_head; a.M = M; a.N = g.vocab_size; a.K = d; a.out = logits_out; a.ldo = g.vocab_size;
  a.nx = s->x_enc; a.nw = m->dec_final_ln; a.neps = g.layer_norm_eps;
  return launch_norm_gemm(P, EPI_STORE_F32, a, st);
}

extern "C" int m2m_session_repack_stats(const m2m_session* s, int* repacks_out, int* rows_moved_out) {
  M2M_REQUIRE(s && repacks_out && rows_moved_out, "m2m_session_repack_stats: null argument");
  *repacks_out = s->repacks; *rows_moved_out = s->rows_moved;
  return M2M_OK;
}

extern "C" int m2m_decode_forced(m2m_session* s, const int64_t* dec_input_ids_dev, int Ld, float* logits_out_dev, void* stream) {
  M2M_REQUIRE(s && dec_input_ids_dev && logits_out_dev, "m2m_decode_forced: null argument");
  if (!s->encoded) { set_error("m2m_decode_forced: %s", encode_missing(s)); return M2M_ERR_STATE; }
  M2M_REQUIRE(Ld >= 1 && Ld <= s->max_dec, "m2m_decode_forced: Ld %d outside [1, %d]", Ld, s->max_dec);
  hipStream_t st = (hipStream_t)stream;
  const EncSwitchScope sw_scope(&s->enc_sw);
  M2M_CHECK_HIP(hipMemcpyAsync(s->forced_ids, dec_input_ids_dev, (size_t)s->B * Ld * 8, hipMemcpyDeviceToDevice, st));
  const char* fwd = getenv("M2M_FORWARD");          // "step": Ld KV-cached decode steps instead (read per call: tests toggle it)
  const bool stepwise = fwd && strcmp(fwd, "step") == 0;
  if (!stepwise) return forward_batched(s, s->forced_ids, Ld, logits_out_dev, st);
  const DecView all{0, s->B, s->states};
  int rc;
  if ((rc = decode_init(s, all, Ld, true, st))) return rc;
  for (int t = 0; t < Ld; ++t)
    if ((rc = decode_launch_step(s, all, true, logits_out_dev, Ld, st))) return rc;
  DecState hs{};
  M2M_CHECK_HIP(hipMemcpyAsync(&hs, all.state, sizeof(hs), hipMemcpyDeviceToHost, st));
  M2M_CHECK_HIP(hipStreamSynchronize(st));
  if (hs.overflow) {
    set_error("m2m_decode_forced: a decoder activation left the fixed-point residual range (|x| >= 2^21) or was not finite");
    return M2M_ERR_RANGE;
  }
  return M2M_OK;
}

// ------------------------------------------------------------------ bench ---
extern "C" int m2m_bench_kernel(m2m_session* s, int which, int self_len, int iters, float* avg_us_host,
                                int64_t* bytes_host, void* stream) {
  M2M_REQUIRE(s && avg_us_host && bytes_host && iters >= 1, "m2m_bench_kernel: bad argument");
  if (!s->encoded) { set_error("m2m_bench_kernel: %s", encode_missing(s)); return M2M_ERR_STATE; }
  M2M_REQUIRE(self_len >= 1 && self_len <= s->max_dec, "m2m_bench_kernel: self_len out of range");
  M2M_REQUIRE(which == M2M_KERNEL_DEC_CROSS_ATTN || which == M2M_KERNEL_DEC_SELF_ATTN || which == M2M_KERNEL_DEC_STEP,
              "m2m_bench_kernel: unknown kernel id %d", which);
  const m2m_model* m = s->m;
  const int Ld = m->g.num_decoder_layers;
  hipStream_t caller = (hipStream_t)stream;
  const int G = plan_groups(s);
  M2M_CHECK_HIP(hipEventRecord(s->ev_in, caller));
  for (int i = 0; i < G; ++i) M2M_CHECK_HIP(hipStreamWaitEvent(s->groups[i].stream, s->ev_in, 0));
  int rc;
  // every chain sees a live loop in its (self_len)-th step
  DecState hs{}; hs.t = self_len - 1; hs.t_copy = self_len - 1; hs.done = 0; hs.out_len = 1; hs.n_unfinished = 0; hs.max_steps = s->max_dec;
  {   // slot -> clip table of the greedy loop: identity (a fresh session has none, a re-packed generate leaves a permutation)
    std::vector<int> ident((size_t)s->B);
    for (int b = 0; b < s->B; ++b) ident[(size_t)b] = b;
    M2M_CHECK_HIP(hipMemcpy(s->tok_row, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  for (int i = 0; i < G; ++i) {
    const DecGroup& gr = s->groups[i];
    M2M_CHECK_HIP(hipMemcpyAsync(gr.view.state, &hs, sizeof(hs), hipMemcpyHostToDevice, gr.stream));
    // ... with every row live: after a generate whose rows emitted EOS the finished-row early-out would skip those rows' K/V
    // streams in the timed launches while bytes_host below counts them in full
    M2M_CHECK_HIP(hipMemsetAsync(s->finished + gr.view.b0, 0, (size_t)gr.view.nb * sizeof(int), gr.stream));
  }
  for (int i = 0; i < G; ++i) M2M_CHECK_HIP(hipStreamSynchronize(s->groups[i].stream));
  hipEvent_t e0, e1;
  M2M_CHECK_HIP(hipEventCreate(&e0));
  M2M_CHECK_HIP(hipEventCreate(&e1));
  const int same_layer = env_int("M2M_BENCH_SAME_LAYER", 0);   // diagnostic: K/V working set small enough for the Infinity Cache
  // the kernel exactly as the decode loop launches it: every chain launches ITS clips on ITS stream (the chains run
  // side by side), back to back, cycling through the decoder layers so the K/V working set is the real loop's.
  // One "launch" of the figures below = the co-scheduled launches of all G chains = all B clips.
  auto run = [&](int n) -> int {
    for (int i = 0; i < n; ++i) {
      const int layer = same_layer ? 0 : i % Ld;
      for (int c = 0; c < G; ++c) {
        const DecGroup& gr = s->groups[c];
        if (which == M2M_KERNEL_DEC_CROSS_ATTN) { if ((rc = decode_launch_attn(s, gr.view, false, layer, 0, gr.stream, decode_headless()))) return rc; }
        else if (which == M2M_KERNEL_DEC_SELF_ATTN) { if ((rc = decode_launch_attn(s, gr.view, true, layer, self_len, gr.stream, decode_headless()))) return rc; }
      }
    }
    return M2M_OK;
  };
  auto fork = [&](hipEvent_t ev) -> int {
    M2M_CHECK_HIP(hipEventRecord(ev, caller));
    for (int c = 0; c < G; ++c) M2M_CHECK_HIP(hipStreamWaitEvent(s->groups[c].stream, ev, 0));
    return M2M_OK;
  };
  auto join = [&](hipEvent_t ev) -> int {
    for (int c = 0; c < G; ++c) {
      M2M_CHECK_HIP(hipEventRecord(s->groups[c].ev_done, s->groups[c].stream));
      M2M_CHECK_HIP(hipStreamWaitEvent(caller, s->groups[c].ev_done, 0));
    }
    M2M_CHECK_HIP(hipEventRecord(ev, caller));
    return M2M_OK;
  };
  float ms = 0.f;
  if (which != M2M_KERNEL_DEC_STEP) {
    if ((rc = run(Ld))) return rc;  // warm-up
    if ((rc = fork(e0))) return rc;
    if ((rc = run(iters))) return rc;
    if ((rc = join(e1))) return rc;
    M2M_CHECK_HIP(hipEventSynchronize(e1));
    M2M_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  } else {
    // the whole decode step as the loop runs it: all chains concurrently, each replaying its graph
    // (t advances by `iters` from self_len - 1, so self-attention lengths are the real ones)
    M2M_REQUIRE(self_len - 1 + iters < s->max_dec, "m2m_bench_kernel: self_len + iters exceeds max_dec_len");
    const int U = 1;
    for (int i = 0; i < G; ++i) if (use_graph() && (rc = ensure_graph(s, s->groups[i], U))) return rc;
    M2M_CHECK_HIP(hipEventRecord(e0, caller));
    for (int i = 0; i < G; ++i) M2M_CHECK_HIP(hipStreamWaitEvent(s->groups[i].stream, e0, 0));
    for (int k = 0; k < iters; ++k)
      for (int i = 0; i < G; ++i) {
        DecGroup& gr = s->groups[i];
        if (use_graph()) M2M_CHECK_HIP(hipGraphLaunch(gr.graph_exec, gr.stream));
        else if ((rc = decode_launch_step(s, gr.view, false, nullptr, 0, gr.stream))) return rc;
      }
    for (int i = 0; i < G; ++i) {
      M2M_CHECK_HIP(hipEventRecord(s->groups[i].ev_done, s->groups[i].stream));
      M2M_CHECK_HIP(hipStreamWaitEvent(caller, s->groups[i].ev_done, 0));
    }
    M2M_CHECK_HIP(hipEventRecord(e1, caller));
    M2M_CHECK_HIP(hipEventSynchronize(e1));
    M2M_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us_host = ms * 1000.0f / (float)iters;
  const int64_t qo = (int64_t)s->B * m->inner * (4 + (int64_t)m->esize);
  const int64_t kv_cross = (int64_t)s->B * m->inner * s->S * 2 * (int64_t)m->esize;
  const int64_t kv_self = (int64_t)s->B * m->inner * self_len * 2 * (int64_t)m->esize;
  if (which == M2M_KERNEL_DEC_CROSS_ATTN) *bytes_host = kv_cross + qo;
  else if (which == M2M_KERNEL_DEC_SELF_ATTN) *bytes_host = kv_self + qo;
  else *bytes_host = (int64_t)Ld * (kv_cross + kv_self + 2 * qo);  // + weights: added by the caller
  for (int i = 0; i < G; ++i) M2M_CHECK_HIP(hipStreamSynchronize(s->groups[i].stream));
  return M2M_OK;
}
