/*
 * music2midi_amd — C ABI of the MI355X (gfx950) Music2MIDI inference hot path.
 *
 * The reference (ytinyui/music2midi) is pure Python with no FFI of its own; its
 * hot path is two Python call sites into third-party libraries:
 *
 *   ref: music2midi/input.py:33-41      LogMelSpectrogram.forward  (torchaudio MelSpectrogram)
 *   ref: music2midi/input.py:50-59      Conditioning.forward       (2 embedding rows prepended)
 *   ref: music2midi/transformer.py:28-39 T5Transformer.forward     (HF T5 teacher-forced forward)
 *   ref: music2midi/transformer.py:41-45 T5Transformer.generate    (HF T5 encoder + greedy generate)
 *
 * Each entry point below names the call site it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative M2M_ERR_* code; it never
 *    throws or aborts.  m2m_last_error() returns a thread-local message.
 *  - "dev" pointers are device (HBM) addresses, "host" pointers are host
 *    addresses.  The caller owns every input, output and workspace buffer; the
 *    library owns only what *_create() returns (plans, repacked weights,
 *    sessions) until the matching *_destroy().
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    All work is enqueued asynchronously on it unless stated otherwise.
 *  - No function allocates or frees device memory except the *_create /
 *    *_destroy pairs.
 */
#ifndef MUSIC2MIDI_AMD_H
#define MUSIC2MIDI_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M2M_ABI_VERSION 1

enum {
  M2M_OK = 0,
  M2M_ERR_INVALID = -1,     /* bad argument / unsupported geometry */
  M2M_ERR_HIP = -2,         /* a HIP runtime call failed */
  M2M_ERR_NOMEM = -3,       /* workspace too small / allocation failed */
  M2M_ERR_STATE = -4,       /* call order violated (e.g. generate before encode) */
  M2M_ERR_RANGE = -5        /* the decoder produced a value outside its fixed-point residual range (|x| >= 2^21) or a
                               non-finite one (corrupt checkpoint, diverged fine-tune): the token ids are NOT valid */
};

enum {
  M2M_PREC_FP32 = 0,        /* fp32 weights / KV / GEMM inputs (f32 MFMA): parity mode */
  M2M_PREC_BF16 = 1,        /* bf16 weights / KV / GEMM inputs, fp32 accumulate: throughput mode */
  M2M_PREC_FP8 = 2          /* m2m_trainer_create only: as BF16, but the dense projection products run on block-scaled OCP FP8
                               (MXFP8, e4m3 elements, 32 per E8M0 scale): forward and dX by default, the weight gradients too with
                               M2M_FP8_PARTS=fwd,dx,dw; M2M_FP8_GRAD=e5m2 for e5m2 gradient operands — BASELINE configs[4] */
};

int m2m_abi_version(void);
const char* m2m_last_error(void);

/* Number of visible HIP devices (0 when none; never fails). */
int m2m_device_count(void);

/* ------------------------------------------------------------------------- *
 * Frontend: framed STFT -> |X|^2 -> mel filterbank -> log(clamp(., 1e-6))
 * replaces ref: music2midi/input.py:25-41 (torchaudio MelSpectrogram + log).
 * ------------------------------------------------------------------------- */
typedef struct m2m_frontend m2m_frontend;

typedef struct {
  int n_fft;                /* must be 2048 (ref: config.yaml:12) */
  int hop_length;           /* 1..n_fft/2 (reference: 256) */
  int n_freqs;              /* n_fft/2 + 1 */
  int n_mels;               /* = d_model (ref: music2midi/transformer.py:20) */
  const float* window_host; /* [n_fft] analysis window (Hann, periodic) */
  const float* fb_host;     /* [n_freqs, n_mels] row-major mel filterbank (dense) */
} m2m_frontend_desc;

int  m2m_frontend_create(const m2m_frontend_desc* desc, m2m_frontend** out);
void m2m_frontend_destroy(m2m_frontend* fe);
/* frames = 1 + n_samples / hop (center=True), or a negative error. */
int  m2m_frontend_num_frames(const m2m_frontend* fe, int n_samples);
/* non-zero taps the sparse filterbank keeps (diagnostic). */
int  m2m_frontend_fb_nnz(const m2m_frontend* fe);

/*
 * wav_dev  [B, T] fp32 row-major.
 * out_dev  row f of clip b is written at out_dev + b*out_batch_stride + (row_offset + f)*n_mels
 *          (fp32).  With row_offset = n_cond and out_batch_stride = (n_cond+frames)*n_mels the
 *          kernel writes straight into the encoder input, leaving rows [0, n_cond) for
 *          m2m_cond_rows_f32 — the reference's torch.cat copy (input.py:59) never happens.
 * T >= n_fft/2 + 1 (reflect padding needs it, as torch.stft does).
 */
int m2m_logmel_f32(const m2m_frontend* fe, const float* wav_dev, int B, int T,
                   float* out_dev, int64_t out_batch_stride, int row_offset, void* stream);

/*
 * Conditioning rows, replaces ref: music2midi/input.py:57-59.
 * tables_dev_host: host array of n_tables device pointers, table i is [n_i, n_dim] fp32.
 * table_rows_host: host array with n_i (indices are range-checked on device: an
 *                  out-of-range index writes NaNs into that row rather than faulting).
 * idx_dev [B, n_tables] int64.  Row i of clip b lands at out_dev + b*out_batch_stride + i*n_dim.
 */
int m2m_cond_rows_f32(const float* const* tables_dev_host, const int* table_rows_host, int n_tables,
                      int n_dim, const int64_t* idx_dev, int B, float* out_dev,
                      int64_t out_batch_stride, void* stream);

/* ------------------------------------------------------------------------- *
 * T5 encoder-decoder weights, replaces what ref: music2midi/transformer.py:14-16
 * builds (T5Config + T5ForConditionalGeneration) once a state dict is loaded.
 * ------------------------------------------------------------------------- */
typedef struct {
  int d_model, d_ff, num_layers, num_decoder_layers, num_heads, d_kv;
  int vocab_size, num_buckets, max_distance;
  int pad_token_id, eos_token_id, decoder_start_token_id;
  float layer_norm_eps;
} m2m_t5_geometry;

/* All pointers: DEVICE, fp32, HuggingFace layout ([out_features, in_features]). */
typedef struct {
  const float *ln0, *q, *k, *v, *o;             /* layer.0.layer_norm, SelfAttention.{q,k,v,o} */
  const float *ln1, *wi0, *wi1, *wo;            /* layer.1.layer_norm, DenseReluDense.{wi_0,wi_1,wo} */
} m2m_enc_layer_weights;

typedef struct {
  const float *ln0, *q, *k, *v, *o;             /* layer.0: self attention */
  const float *ln1, *cq, *ck, *cv, *co;         /* layer.1: EncDecAttention.{q,k,v,o} */
  const float *ln2, *wi0, *wi1, *wo;            /* layer.2: DenseReluDense */
} m2m_dec_layer_weights;

typedef struct {
  const float* shared;            /* [V, d_model] token embedding */
  const float* lm_head;           /* [V, d_model] separate tensor (transformers 4.34 untied head) */
  const float* enc_rel_bias;      /* [num_buckets, H] encoder.block.0 relative_attention_bias */
  const float* dec_rel_bias;      /* [num_buckets, H] decoder.block.0 relative_attention_bias */
  const float* enc_final_ln;      /* [d_model] */
  const float* dec_final_ln;      /* [d_model] */
  const m2m_enc_layer_weights* enc;  /* host array [num_layers] */
  const m2m_dec_layer_weights* dec;  /* host array [num_decoder_layers] */
} m2m_t5_weights;

typedef struct m2m_model m2m_model;

/* Repacks the weights into kernel layouts (device-side kernels on `stream`,
 * synchronised before returning); the source tensors may be freed afterwards. */
int  m2m_model_create(const m2m_t5_geometry* geom, const m2m_t5_weights* w, int precision,
                      void* stream, m2m_model** out);
void m2m_model_destroy(m2m_model* m);
int  m2m_model_precision(const m2m_model* m);
int64_t m2m_model_param_bytes(const m2m_model* m);   /* bytes of repacked weights held */
/* 64-bit position-weighted checksum of the repacked device weights (sum of word_i * (2 i + 1) mod 2^64 over the
 * whole packed blob), synchronised before returning.  After the one-time weight broadcast of a multi-GPU run
 * (ref: train.py:40-41 strategy="ddp") every rank's value must be equal: the ranks all-reduce MIN and
 * MAX of it and fail loudly on a difference instead of decoding from diverged replicas. */
int  m2m_model_checksum(const m2m_model* m, uint64_t* out_host, void* stream);

/* T5 relative-position bucket (hf: models/t5/modeling_t5.py:217-262), host-side,
 * exported so the integer table can be tested without a GPU. rel = key_pos - query_pos. */
int m2m_rel_bucket(int rel, int bidirectional, int num_buckets, int max_distance);

/* ------------------------------------------------------------------------- *
 * Session: workspace + captured decode-step graph for up to (max_batch,
 * max_enc_len, max_dec_len).  One session per concurrent call; a session is
 * not re-entrant.  The workspace is caller-owned device memory.
 * ------------------------------------------------------------------------- */
typedef struct m2m_session m2m_session;

int64_t m2m_session_workspace_bytes(const m2m_model* m, int max_batch, int max_enc_len, int max_dec_len);
int  m2m_session_create(const m2m_model* m, int max_batch, int max_enc_len, int max_dec_len,
                        void* workspace_dev, int64_t workspace_bytes, m2m_session** out);
void m2m_session_destroy(m2m_session* s);

/*
 * Encoder stack + cross-attention K/V projection for every decoder layer,
 * replaces the encoder half of ref: music2midi/transformer.py:44 (HF generate
 * runs the encoder once, hf: generation/utils.py:809-848) and of :35-37.
 * inputs_embeds_dev [B, S, d_model] fp32 (cond rows + log-mel rows).
 * enc_out_dev: optional [B, S, d_model] fp32 copy of the final encoder states (NULL to skip).
 */
int m2m_encode(m2m_session* s, const float* inputs_embeds_dev, int B, int S, float* enc_out_dev, void* stream);

/*
 * KV-cached greedy decode, replaces the decode half of
 * ref: music2midi/transformer.py:44 (hf: generation/utils.py:2783-2973, do_sample=False):
 * start token decoder_start_token_id, argmax, rows that emitted EOS keep emitting
 * pad, stop when every row has finished or the length reaches max_length.
 * tokens_out_dev [B, max_length] int64 (columns >= *out_len_host are pad).
 * *out_len_host: number of valid columns L <= max_length.  This call synchronises `stream`.
 * Returns M2M_ERR_RANGE (tokens are still written) when an activation left the decoder's fixed-point range or
 * was not finite - where the fp32 reference would have produced Inf/NaN logits.  On every error return all
 * library-owned streams have been synchronised, so the caller may free or reuse the workspace at once.
 *
 * Session state afterwards: m2m_encode -> any number of m2m_decode_forced / m2m_generate_greedy / m2m_bench_kernel calls on the same
 * encode is legal (HF's forward and generate share one encoder pass the same way, ref: music2midi/transformer.py:28-45) - EXCEPT
 * that a greedy decode which re-packed its live rows (m2m_session_repack_stats reports rows_moved > 0, see below) has overwritten
 * finished clips' cross K/V with live ones and thereby CONSUMED the encode: the next call that needs it returns M2M_ERR_STATE
 * ("re-encode: ...") until m2m_encode runs again.  The token ids of the call itself are unaffected.
 */
int m2m_generate_greedy(m2m_session* s, int max_length, int64_t* tokens_out_dev, int* out_len_host, void* stream);

/*
 * Rows end at different steps (ref: music2midi/model.py:115-135 decodes chunks of inference.batch_size = 128 three-second
 * segments to max_length 1024; a trained checkpoint ends a segment after tens to hundreds of tokens).  Once a quarter of the
 * rows still being decoded have emitted EOS, m2m_generate_greedy re-packs the live rows into the first slots of the batch at its
 * next host poll and goes on with smaller launches; ids do not depend on it.  The host polls after 16, 32, 64, 96 and 128 steps and
 * then every 64 (without the re-packing: every 64); each poll drains every chain, and a poll at which a re-packing is possible (a
 * chain still running, >= 64 steps left, >= 2 rows) also reads the finished flags back synchronously - so M2M_COMPACT=1 adds four
 * early polls (about a decode step each) even to a batch that never emits EOS.  This returns how often rows were re-packed in the
 * last call and how many were moved; rows_moved > 0 means that call consumed the session's encode (see m2m_generate_greedy).
 * M2M_COMPACT=0 in the environment disables the re-packing.
 */
int m2m_session_repack_stats(const m2m_session* s, int* repacks_out, int* rows_moved_out);

/*
 * Teacher-forced decoder pass, replaces the decoder half of
 * ref: music2midi/transformer.py:35-37 (logits only; the loss is a host-side reduction).
 * dec_input_ids_dev [B, Ld] int64 (= shift_right(labels)), logits_out_dev [B, Ld, V] fp32.
 */
int m2m_decode_forced(m2m_session* s, const int64_t* dec_input_ids_dev, int Ld, float* logits_out_dev, void* stream);

/* ------------------------------------------------------------------------- *
 * Training step: what ref: music2midi/model.py:27-43 drives through Lightning —
 * `self.model(inputs).loss` (ref: music2midi/transformer.py:28-39: HF T5 forward with labels),
 * `loss.backward()`, and `Adafactor(self.parameters(), warmup_init=True).step()` with
 * `AdafactorSchedule` (relative step, no external learning rate).
 *
 * Parameters and gradients live in two caller-owned flat fp32 device buffers of
 * m2m_trainer_num_params() floats; m2m_trainer_tensor_info() gives every tensor's state-dict key
 * (relative to the LightningModule's `model.` prefix), shape and offset, so the host framework can
 * expose views of the same memory as its parameters.  q/k/v (and wi_0/wi_1, cross k/v) of a layer
 * are adjacent so that each fused projection is one matrix.  The library owns the activations,
 * the bf16 copy of the weights (throughput mode) and the optimizer state.
 * ------------------------------------------------------------------------- */
typedef struct m2m_trainer m2m_trainer;

typedef struct {
  char name[160];           /* e.g. "transformer.encoder.block.0.layer.0.SelfAttention.q.weight" */
  int64_t offset;           /* floats into the flat buffers */
  int rows, cols;           /* 1-D tensors: rows = length, cols = 0 */
} m2m_tensor_info;

/* n_cond / cond_rows_host: the conditioning embedding tables (ref: music2midi/input.py:45-55), trainable.
 * max_*: largest batch, encoder length (cond rows + frames) and label length of a step.
 * A trainer is used from one host thread at a time; it owns its activations (m2m_trainer_workspace_bytes: 2.3 GB at 16 clips of
 * 3 s — every sub-layer keeps its operands until the grouped weight-gradient launch), two streams and a captured graph. */
int  m2m_trainer_create(const m2m_t5_geometry* geom, int n_cond, const int* cond_rows_host, int precision,
                        int max_batch, int max_enc_len, int max_dec_len, m2m_trainer** out);
void m2m_trainer_destroy(m2m_trainer* t);
int64_t m2m_trainer_num_params(const m2m_trainer* t);
int  m2m_trainer_num_tensors(const m2m_trainer* t);
int  m2m_trainer_tensor_info(const m2m_trainer* t, int index, m2m_tensor_info* out);
int64_t m2m_trainer_workspace_bytes(const m2m_trainer* t);   /* device bytes the trainer holds (activations etc.) */

/*
 * Teacher-forced forward + backward.
 * enc_inputs_dev [B, S, d_model] fp32: log-mel rows at [n_cond, S) as m2m_logmel_f32 writes them; rows
 *                [0, n_cond) are overwritten from the CURRENT conditioning tables (they are trainable).
 * cond_idx_dev   [B, n_cond] int64.        labels_dev [B, Ld] int64, -100 = ignored
 *                (decoder inputs = shift_right(labels), hf: modeling_t5.py:618-637).
 * loss_out_dev   fp32[1]: mean cross entropy over the non-ignored labels.
 * grads_dev      flat fp32, OVERWRITTEN with d loss / d params; NULL = forward only.
 * logits_out_dev optional [B, Ld, V] fp32.
 * Deterministic: every reduction has a fixed order, two calls give bit-identical results.
 * Streams: with gradients the pass runs on the trainer's own streams — stream-ordered behind everything already on `stream`
 * (the inputs are copied into trainer-owned buffers first) and `stream` waits for its end, so for the caller it behaves like
 * work on `stream`.  From the second call with the same (params_dev, grads_dev, B, S, Ld, dropout) on, the pass is replayed as
 * one captured HIP graph (M2M_TRAIN_GRAPH=0 disables that), one graph PER SHAPE: up to 8 keys are kept, least recently used
 * first out, so batches whose label length changes and comes back replay their own graph; params_dev / grads_dev must stay
 * valid while the trainer lives.  A batch without any scored label (all -100) gives loss = NaN and zero gradients, as torch's
 * CrossEntropyLoss does.
 */
int m2m_train_forward_backward(m2m_trainer* t, const float* params_dev, const float* enc_inputs_dev,
                               const int64_t* cond_idx_dev, const int64_t* labels_dev, int B, int S, int Ld,
                               float* loss_out_dev, float* grads_dev, float* logits_out_dev, void* stream);

/* Dropout of the teacher-forced pass (hf T5Config.dropout_rate, 0.1 in the reference's config; active because
 * ref: train.py:33 puts the module in train() mode): on the embeddings, the attention probabilities, every
 * residual branch, the gated activation and the final norms, as hf: modeling_t5.py places them.  Masks come from a
 * counter-based hash of (seed, forward_backward call index since this call, site, element) and are regenerated in the
 * backward pass.  p = 0 (the default after create) switches it off. */
int m2m_trainer_set_dropout(m2m_trainer* t, float p, uint64_t seed);

/* Data-parallel training (ref: train.py:40-41 — pl.Trainer(strategy="ddp"): the gradient all-reduce overlapped with backward).
 * With a sync stream set, m2m_train_forward_backward issues the backward pass in two parts — decoder side, then encoder side —
 * and makes `sync_stream` wait (a device-side event, no host synchronisation) for the point where the gradients of the two
 * "early" ranges are final and will not be written again in this call: the shared embedding + lm_head at the front of the flat
 * buffer, and the decoder blocks.  The all-reduce of those ranges, enqueued on `sync_stream` right after the call returns, runs
 * beside the encoder-side backward; the rest of the buffer is final when the caller's own stream continues.  The gradients are
 * bit-identical to the unsplit pass.  nullptr switches the split off.
 * m2m_trainer_early_grad_ranges: out[0..3] = {offset, count, offset, count} in floats of the flat gradient buffer. */
int m2m_trainer_set_sync_stream(m2m_trainer* t, void* sync_stream);
int m2m_trainer_early_grad_ranges(const m2m_trainer* t, int64_t* out);
/* Measurement hook (bench.py train_configs4.roofline): nodes (kernel launches + copies) of the captured graph(s) of the shape
 * most recently passed to m2m_train_forward_backward = launches per training step; 0 while that shape has not been captured. */
int m2m_trainer_graph_nodes(const m2m_trainer* t);

/* One transformers.optimization.Adafactor step with the reference's settings (lr=None, eps=(1e-30, 1e-3),
 * clip_threshold=1.0, decay_rate=-0.8, beta1=None, weight_decay=0, scale_parameter, relative_step,
 * warmup_init): params_dev is updated in place from grads_dev.  The step counter and the factored second
 * moments are library-owned; export/import them to checkpoint a run. */
int m2m_adafactor_step(m2m_trainer* t, float* params_dev, const float* grads_dev, void* stream);
int m2m_adafactor_get_step(const m2m_trainer* t);
int64_t m2m_adafactor_state_floats(const m2m_trainer* t);
int m2m_adafactor_state_export(const m2m_trainer* t, float* state_out_dev, void* stream);
int m2m_adafactor_state_import(m2m_trainer* t, const float* state_in_dev, int step, void* stream);

/* One MXFP8 product, C[M,N] = A[M,K] . B[N,K]^T (fp32 in and out, device pointers, row-major): both operands are quantised
 * to OCP block-scaled FP8 — 32 elements along K per power-of-two E8M0 scale, e4m3 elements (e5m2 for A when a_is_e5m2, the
 * gradient format) — and multiplied on gfx950's scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4).  This is the arithmetic of
 * the fp8 training mode (M2M_PREC_FP8 of m2m_trainer_create), exposed so it can be checked on its own.  A test utility:
 * unlike the hot-path entry points it allocates and frees its own device scratch, and it synchronises `stream`. */
int m2m_mx8_matmul_f32(const float* a_dev, const float* b_dev, int M, int N, int K, int a_is_e5m2, float* c_dev, void* stream);
/* The same with A in bf16, as the training step feeds it (K a multiple of 8): fused != 0 quantises A inside the product's operand
 * staging (one launch per product, what the fp8 mode runs), 0 through the separate row quantiser; bit-identical by construction
 * and by test.  Test utility like the above. */
int m2m_mx8_matmul_bf16a(const uint16_t* a_bf16_dev, const float* b_dev, int M, int N, int K, int a_is_e5m2, int fused, float* c_dev,
                         void* stream);

/* The training step's whole-head attention kernels on their own (csrc/attn_train.hip; hf: modeling_t5.py:159-170 T5Attention and its
 * autograd backward): q [B, Sq, H*64], k / v [B, Sk, H*64] bf16, row-major; bias_tab [H][Sq + Sk - 1] fp32 by (key - query + Sq - 1)
 * or NULL; no 1/sqrt(d) scaling; dropout on the probabilities with the step's counter-based hash (element index
 * ((b*H + h)*Sq + query) * round_up_8(Sk) + key, key = splitmix64(step_key + site_salt)).  Forward: out [B, Sq, H*64] bf16 and the
 * row log-sum-exp lse [B*H][Sq] fp32.  Backward: from q, k, v, out, lse and d_out the gradients dq / dk / dv (layouts of q / k / v) and,
 * with a bias, diag_part [B*H][ceil(Sq/32)][Sk + 31] = per-query-block sums of dS along the diagonals key - local row = x - 31.
 * keep_bits [B*H][ceil(Sk/32)][round_up_32(Sq)] uint32 (needed when drop_p > 0): the forward pass writes one word per (key block, query) —
 * bit k = probability (query, 32 * block + k) is kept — and the backward pass reads them instead of hashing again.
 * Test utilities like m2m_mx8_matmul_f32 (they own one device word for the step key and synchronise `stream` to set it). */
int m2m_attn_head_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, const float* bias_tab, int B, int H, int Sq, int Sk,
                           int causal, float drop_p, uint64_t step_key, uint64_t site_salt, uint16_t* out, float* lse, uint32_t* keep_bits,
                           void* stream);
int m2m_attn_head_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out, const float* lse, const uint16_t* d_out,
                           const float* bias_tab, int B, int H, int Sq, int Sk, int causal, float drop_p, uint64_t step_key, uint64_t site_salt,
                           const uint32_t* keep_bits, uint16_t* dq, uint16_t* dk, uint16_t* dv, float* diag_part, void* stream);

/* ------------------------------------------------------------------------- *
 * Measurement hooks (bench.py): time one kernel of the decode step in isolation
 * with hipEvents on `stream`, cycling through all decoder layers so the working
 * set matches the real loop.  Requires a prior m2m_encode on the session.
 * ------------------------------------------------------------------------- */
enum {
  M2M_KERNEL_DEC_CROSS_ATTN = 0,
  M2M_KERNEL_DEC_SELF_ATTN = 1,
  M2M_KERNEL_DEC_STEP = 2        /* the whole captured step graph */
};
/* self_len: number of cached self-attention keys to assume (1..max_dec_len).
 * avg_us_host: mean duration of one launch in microseconds.
 * bytes_host: algorithmic HBM bytes one launch moves (K/V streamed + q read + o written). */
int m2m_bench_kernel(m2m_session* s, int which, int self_len, int iters, float* avg_us_host,
                     int64_t* bytes_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUSIC2MIDI_AMD_H */
