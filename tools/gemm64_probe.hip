// What bounds the small-tile NT product (gemm_kernel TF = 1: 64 x 64 tile, BK = 128, 4 waves) at the training step's dX shape
// M = 4096, N = 384, K = 2048 (16.7 us in the step)?  The same loop with its parts switched:
//   MODE 0: as the product does it (register prefetch one step ahead, store, two barriers per step)
//   MODE 1: two LDS buffers, ONE barrier per step          MODE 2: no global loads        MODE 3: MFMA + fragment reads only
//   hipcc --offload-arch=gfx950 -O3 tools/gemm64_probe.hip -o /tmp/g64 && /tmp/g64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BK = 128, PITCH = BK + 8;

template <int MODE>
__global__ __launch_bounds__(256) void g64(const short* __restrict__ A, const short* __restrict__ W, int M, int N, int K, float* __restrict__ out) {
  extern __shared__ __align__(16) short lds[];
  constexpr int NBUF = MODE == 1 ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int ntn = N / 64, mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
  const int m0 = mt * 64, n0 = nt * 64;
  // named registers: arrays across the k loop end up in scratch (hipcc 7.2)
#define CH(i) const int c##i = tid + 256 * (i); const short* ap##i = A + (int64_t)(m0 + c##i / 16) * K + (c##i % 16) * 8; \
              const short* bp##i = W + (int64_t)(n0 + c##i / 16) * K + (c##i % 16) * 8; const int so##i = (c##i / 16) * PITCH + (c##i % 16) * 8; uint4 ra##i, rb##i;
  CH(0) CH(1) CH(2) CH(3)
#define LD(i, k0) if (MODE <= 1) { ra##i = *reinterpret_cast<const uint4*>(ap##i + (k0)); rb##i = *reinterpret_cast<const uint4*>(bp##i + (k0)); } \
                  else { ra##i = make_uint4((k0), i, tid, 1); rb##i = ra##i; }
#define GLOAD(k0) { LD(0, k0) LD(1, k0) LD(2, k0) LD(3, k0) }
#define ST(i) *reinterpret_cast<uint4*>(As + so##i) = ra##i; *reinterpret_cast<uint4*>(Bs + so##i) = rb##i;
  f32x16 acc = {};
  const int nk = K / BK;
  GLOAD(0)
  for (int kt = 0; kt < nk; ++kt) {
    short* As = lds + (NBUF == 2 ? (kt & 1) * (128 * PITCH) : 0);
    short* Bs = As + 64 * PITCH;
    if (MODE != 3) {
      if (NBUF == 1) __syncthreads();
      ST(0) ST(1) ST(2) ST(3)
      __syncthreads();
      if (kt + 1 < nk) GLOAD((kt + 1) * BK)
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const uint4 fa = *reinterpret_cast<const uint4*>(As + (wm * 32 + r) * PITCH + s * 16 + 8 * h);
      const uint4 fb = *reinterpret_cast<const uint4*>(Bs + (wn * 32 + r) * PITCH + s * 16 + 8 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * 32 + r;
    out[(int64_t)row * N + col] = acc[e];
  }
}

template <int MODE>
static int run(const short* A, const short* W, int M, int N, int K, float* out, const char* what) {
  const int tiles = (M / 64) * (N / 64);
  const size_t smem = (size_t)(MODE == 1 ? 2 : 1) * 128 * PITCH * 2;
  CK(hipFuncSetAttribute((const void*)g64<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(g64<MODE>, dim3(tiles), dim3(256), smem, 0, A, W, M, N, K, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  const double us = best * 1e3 / 20, fl = 2.0 * M * N * K;
  printf("M %d N %d K %d  mode %d %-44s %6.2f us per launch (back to back)  %5.0f TFLOP/s\n", M, N, K, MODE, what, us, fl / us * 1e-6);
  return 0;
}

int main() {
  const int M = 4096, NMAX = 2048, KMAX = 2048;
  short *A, *W; float* out;
  CK(hipMalloc(&A, (size_t)M * KMAX * 2)); CK(hipMalloc(&W, (size_t)NMAX * KMAX * 2)); CK(hipMalloc(&out, (size_t)M * NMAX * 4));
  {
    const size_t n = (size_t)M * KMAX; short* h = (short*)malloc(n * 2); unsigned x = 1u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; const float f = ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (short)(u >> 16); }
    CK(hipMemcpy(A, h, n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h, (size_t)NMAX * KMAX * 2, hipMemcpyHostToDevice)); free(h);
  }
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int N = cfg == 2 ? 1536 : 384, K = cfg == 0 ? 2048 : 384;
    if (cfg == 1) { /* N = 384, K = 384: the residual / dO products */ }
    if (run<0>(A, W, M, N, K, out, "as the product (two barriers per step)")) return 1;
    if (run<1>(A, W, M, N, K, out, "two LDS buffers, one barrier per step")) return 1;
    if (run<2>(A, W, M, N, K, out, "no global loads")) return 1;
    if (run<3>(A, W, M, N, K, out, "MFMA + fragment reads only")) return 1;
  }
  return 0;
}
