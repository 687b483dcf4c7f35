// Where does a k-step of the weight-gradient tile go?  The k loop of dw_tile_tr (128 x 128 tile, BK = 64, k-major operands copied
// straight to LDS, fragments by ds_read_b64_tr_b16, 16 MFMA 32x32x16 per wave and step) with its parts switched off one by one:
//   MODE 0: everything          1: no global loads (LDS written from constant registers)      2: no LDS writes / barriers either
//   MODE 3: MFMA only (fragments from registers)        4: everything, two k-steps of loads in flight (second register set)
// 1850 tiles x 65 steps like the 16-clip training step.  hipcc --offload-arch=gfx950 -O3 tools/kloop_probe.hip -o /tmp/kloop && /tmp/kloop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int P = 160, BK = 64;

__device__ inline uint4 trfrag(const short* p) {
  typedef v4s __attribute__((address_space(3))) * lp;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * P));
  return make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y, __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
}

template <int MODE>
__global__ __launch_bounds__(256) void kloop(const short* __restrict__ A, const short* __restrict__ B, int64_t lda, int64_t ldb, int K, int tn2, float* __restrict__ out, int n1t) {
  __shared__ __align__(16) short AB[2 * BK * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int t1 = (blockIdx.x / tn2) % (int)(lda / 128 > 0 ? min((int)(lda / 128), n1t) : 1), t2 = blockIdx.x % tn2;
  const int op = tid >> 7, c = tid & 127, ch = c & 15, rw = c >> 4;
  const int64_t ld = op ? ldb : lda;
  const short* src = (op ? B + t2 * 128 : A + t1 * 128) + ch * 8;
  short* dst = AB + op * (BK * P) + rw * P + ch * 8;
  uint4 r[8], r2[8];
  auto gload = [&](uint4 (&x)[8], int k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = (MODE == 0 || MODE == 4) ? *reinterpret_cast<const uint4*>(src + (int64_t)min(k0 + rw + 8 * u, K - 1) * ld) : make_uint4(k0, u, tid, 1);
  };
  const int li = lane & 15, q = li >> 2, pp = li & 3, gq = lane >> 4, h = lane >> 5;
  const short* fa = AB + (8 * h + q) * P + wm * 64 + 16 * (gq & 1) + 4 * pp;
  const short* fb = AB + BK * P + (8 * h + q) * P + wn * 64 + 16 * (gq & 1) + 4 * pp;
  f32x16 acc[2][2] = {};
  gload(r, 0);
  if (MODE == 4) gload(r2, BK);
  uint4 fr = make_uint4(tid, 2, 3, 4);
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (MODE <= 1 || MODE == 4) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<uint4*>(dst + 8 * u * P) = r[u];
      __syncthreads();
      if (MODE == 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = r2[u];
        if (k0 + 2 * BK < K) gload(r2, k0 + 2 * BK);
      } else if (k0 + BK < K) gload(r, k0 + BK);
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (MODE == 3) { a[i] = fr; b[i] = fr; fr.x += 1; }
        else { a[i] = trfrag(fa + 16 * s * P + 32 * i); b[i] = trfrag(fb + 16 * s * P + 32 * i); }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  if (sum == 12345.678f) out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE>
static int run(const short* A, const short* B, int K, float* out, const char* what, int N1 = 1152 * 8, int n1t = 72, int N2 = 384) {
  const int tn2 = N2 / 128, tiles = 1848 / tn2 * tn2;       // one wide dY against X: 616 x 3 tiles
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kloop<MODE>, dim3(tiles), dim3(256), 0, 0, A, B, (int64_t)N1, (int64_t)N2, K, tn2, out, n1t);
    hipEventRecord(e1, 0);
    CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  const double fl = (double)tiles * 128 * 128 * K * 2;
  printf("mode %d %-58s %7.1f us  %6.0f TFLOP/s  (%.2f us per k-step and tile-slot of a CU)\n", MODE, what, best * 1e3, fl / best * 1e-9, best * 1e3 / ((double)tiles / 256 * (K / BK)));
  return 0;
}

int main() {
  const int K = 4160, N1 = 1152 * 8, N2 = 384;
  short *A, *B; float* out;
  CK(hipMalloc(&A, (size_t)K * N1 * 2)); CK(hipMalloc(&B, (size_t)K * 1024 * 2)); CK(hipMalloc(&out, 1848 * 256 * 4));   // B: the widest X of the runs below
  CK(hipMemset(A, 0, (size_t)K * N1 * 2)); CK(hipMemset(B, 0, (size_t)K * 1024 * 2));
  if (run<0>(A, B, K, out, "full k loop, all-zero operands")) return 1;
  {   // realistic operand bits (the matrix cores' clock depends on the data): bf16 values ~N(0,1)
    const size_t na = (size_t)K * N1, nb = (size_t)K * 1024;
    short* h = (short*)malloc(na * 2);
    unsigned x = 12345u;
    for (size_t i = 0; i < na; ++i) { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (short)(u >> 16); }
    CK(hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h, nb * 2, hipMemcpyHostToDevice));
    free(h);
  }
  if (run<0>(A, B, K, out, "full k loop")) return 1;
  // leading dimensions of the real operands: d(gate pair) 2048, dqkv 1536, ao 512, h 384, mid 1024 (the A image is re-read modulo its width)
  if (run<0>(A, B, K, out, "full, lda 2048 (16 column tiles re-used)", 2048, 16)) return 1;
  if (run<0>(A, B, K, out, "full, lda 2048 + 64", 2112, 16)) return 1;
  if (run<0>(A, B, K, out, "full, lda 1536", 1536, 12)) return 1;
  if (run<0>(A, B, K, out, "full, lda 512, ldb 1024", 512, 4, 1024)) return 1;
  if (run<0>(A, B, K, out, "full, lda 9216, 16 column tiles re-used", 9216, 16)) return 1;
  if (run<4>(A, B, K, out, "full, two k-steps of loads in flight")) return 1;
  if (run<1>(A, B, K, out, "no global loads")) return 1;
  if (run<2>(A, B, K, out, "no global loads, no LDS writes, no barriers")) return 1;
  if (run<3>(A, B, K, out, "MFMA only")) return 1;
  return 0;
}
