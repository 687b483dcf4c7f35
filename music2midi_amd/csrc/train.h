// Internal structures of the training path (train.hip); not part of the C ABI.
#pragma once

#include "common.h"

namespace m2m {

enum { TG_STORE_T = 0, TG_STORE_F32 = 1, TG_ACC_F32 = 2, TG_RESID_F32 = 3 };

// C[z][M,N] (op)= alpha * A[z][M,K] . B[z][N,K]^T ; z = (b1, b2)
struct BGemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* R;          // TG_RESID_F32: C = R + acc (same indexing as C)
  int M, N, K;
  int64_t lda, ldb, ldc;   // elements between consecutive rows AS STORED
  int a_kmajor, b_kmajor;  // 0: element (m, k) at [m*ld + k]; 1: at [k*ld + m]
  int nb1, nb2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  float alpha;
  // split-K (nb1 == nb2 == 1 only): blockIdx.z takes k in [z*kchunk, (z+1)*kchunk); partial tiles go to Cpart[z][M][N]
  // (fp32) and splitk_reduce sums them into C in a fixed order.  ksplit <= 1: off.
  int ksplit, kchunk;
  float* Cpart;
  // TG_RESID_F32 only: dropout on the product before the residual add, C = R + keep(i) * drop_scale * acc  (drop_thresh == 0: off)
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t drop_key;           // the site's salt: key = splitmix64(*drop_step + drop_key) (common.h DropKey)
  const uint64_t* drop_step;
  // pair mode (A2 != null): a SECOND product of the same shape in the same launch (blockIdx.z >= nb1*nb2): same strides for A and
  // C, its own B layout — dV = P~^T dO and dK = dS^T Q of an attention layer are one launch
  const void *A2, *B2;
  // XCD-contiguous workgroup order (set by launch_bgemm): 1-D launch of xcd_total = nx * ny * nz logical workgroups; 0: plain 3-D grid
  int xcd_total, xcd_nx, xcd_ny;
  void* C2;
  int64_t ldb2, sB1_2, sB2_2;
};


int launch_bgemm(int precision, int epi, const BGemmArgs& g, hipStream_t st);

// MXFP8 product (mx8.hip): C[M,N] (epi)= A[M,K] . B[N,K]^T, block-scaled fp8 operands (32 elements along K per E8M0 scale)
struct MxGemmArgs {
  const uint8_t *A, *B;      // fp8 bytes, [M][lda] / [N][ldb]; lda, ldb multiples of 128, zero-padded past K
  const uint8_t *sA, *sB;    // E8M0 scales, [M][lda/32] / [N][ldb/32]
  void* C;
  const float* R;            // TG_RESID_F32
  int M, N, K;
  int64_t lda, ldb, ldc;
  int ksplit, kchunk;
  float* Cpart;
  uint32_t drop_thresh;
  float drop_scale;
  uint64_t drop_key;           // salt, as in BGemmArgs
  const uint64_t* drop_step;
  // launch_mxgemm_q: A is given as bf16 [M][ld_src] (columns >= Kvalid count as zero) and quantised while it is staged
  const void* Asrc;
  int64_t ld_src;
  int Kvalid;
  // XCD-contiguous workgroup order (set by the launchers of the _q / _p kernels): 1-D launch of xcd_total = nx * ny logical
  // workgroups, column tile fastest, so the column tiles of a row block share its rows in ONE XCD's L2; 0: plain 2-D grid
  int xcd_total, xcd_nx;
};
int launch_mxgemm(int fmt_a, int fmt_b, int epi, const MxGemmArgs& g, hipStream_t st);      // fmt: 0 = e4m3, 1 = e5m2
int launch_mxgemm_q(int fmt_a, int epi, const MxGemmArgs& g, hipStream_t st);               // A quantised in the product's own staging
int launch_mxq_rows(int src_kind, const void* src, int64_t ld_s, uint8_t* q, uint8_t* sc, int R, int C, int Cp, int fmt, hipStream_t st);
int launch_mxq_cols(int src_kind, const void* src, int64_t ld_s, uint8_t* qt, uint8_t* sc, int R, int C, int Rp, int fmt, hipStream_t st);

// Whole-head attention of the training step (attn_train.hip): forward with the row log-sum-exp, two-pass backward
struct HeadAttnArgs {
  // (position, d) of clip b, head h at ptr + b * sXb + h * 64 + position * ldx   (bf16 storage)
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
  int key_split;               // forward: waves per query block (set by the launcher)
  uint32_t* keep_bits;         // with dropout: [B*H][ceil(Sk/32)][round_up_32(Sq)] words, bit k of word (key block j, query q) = probability (q, 32 j + k)
                               // is kept.  The forward pass hashes once and writes them; both backward orientations read them instead of hashing again.
};
constexpr int AH_MAX_S = 288;  // rows an LDS image holds (9 blocks of 32): 36 KB per [S, 64] bf16 operand, two images + tables per workgroup, two workgroups per CU
int launch_attn_head_fwd(const HeadAttnArgs& a, int nB, hipStream_t st);
int launch_attn_head_bwd(const HeadAttnArgs& a, int nB, hipStream_t st);

// Adafactor plan (device tables built once per trainer)
struct AfTensor {
  int64_t offset;          // into the flat parameter / gradient buffers
  int rows, cols;          // vectors: rows = 1
  int row_off;             // into rowsum / rfac
  int cfac_off;            // into cfac
  int64_t col_off;         // into colpart (nblocks * cols floats)
  int64_t state_off;       // into the optimizer state (matrix: R[rows] | C[cols]; vector: V[cols])
  int block0, nblocks;
};
struct AfBlock {
  int tensor, row0;
  int64_t col_off;
};
struct AfPlan {
  int n_tensors = 0, n_blocks = 0;
  AfTensor* tensors = nullptr;
  AfBlock* blocks = nullptr;
  float *rowsum = nullptr, *colpart = nullptr, *blk_a = nullptr, *blk_b = nullptr, *state = nullptr, *rfac = nullptr,
        *cfac = nullptr, *tstat = nullptr;
  int64_t state_floats = 0;
};
int launch_adafactor(const AfPlan& pl, float* P, const float* G, int step, hipStream_t st);

}  // namespace m2m
