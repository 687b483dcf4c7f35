// Weight repacking kernels: fp32 HuggingFace-layout tensors -> kernel layouts in the
// storage type of the precision mode (fp32 or bf16).  Run once in m2m_model_create.
#include "t5.h"

namespace m2m {

template <typename T>
__global__ void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = from_f32<T>(src[i]);
}

// dst rows: for chunk c: [c*2*half, +half) = wi0 rows [c*half, +half); next half rows = wi1 rows [c*half, +half)
template <typename T>
__global__ void interleave_kernel(const float* __restrict__ wi0, const float* __restrict__ wi1,
                                  T* __restrict__ dst, int dff, int d, int half) {
  const int64_t n = (int64_t)2 * dff * d;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int row = (int)(i / d), col = (int)(i - (int64_t)row * d);
    const int c = row / (2 * half), r = row - c * 2 * half;
    const float* src = (r < half) ? wi0 : wi1;
    const int srow = c * half + (r < half ? r : r - half);
    dst[i] = from_f32<T>(src[(int64_t)srow * d + col]);
  }
}

static inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int launch_convert(int precision, const float* src, void* dst, int64_t n, hipStream_t st) {
  if (precision == M2M_PREC_BF16)
    hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, src, (bf16_t*)dst, n);
  else
    hipLaunchKernelGGL(convert_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, src, (float*)dst, n);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

int launch_interleave(int precision, const float* wi0, const float* wi1, void* dst, int dff, int d, int half,
                      hipStream_t st) {
  const int64_t n = (int64_t)2 * dff * d;
  if (precision == M2M_PREC_BF16)
    hipLaunchKernelGGL(interleave_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, wi0, wi1, (bf16_t*)dst, dff, d, half);
  else
    hipLaunchKernelGGL(interleave_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, wi0, wi1, (float*)dst, dff, d, half);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

// 64-bit position-weighted checksum of a device buffer (32-bit words w_i): sum of w_i * (2 i + 1) mod 2^64.  Integer adds in any
// order give the same value, so it is reproducible; the odd multipliers make it sensitive to WHERE a word sits, not only to the
// multiset of words.  Used after the multi-GPU weight broadcast: every rank's repacked weights must be the same bytes.
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ w, int64_t n, unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long part[4];
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += (unsigned long long)w[i] * (unsigned long long)(2 * i + 1);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

int launch_checksum(const void* buf, int64_t bytes, unsigned long long* acc_dev, hipStream_t st) {
  const int64_t n = bytes / 4;
  hipLaunchKernelGGL(checksum_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const uint32_t*)buf, n, acc_dev);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

int launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st) {
  M2M_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return M2M_OK;
}

int launch_fill_zero(void* dst, int64_t bytes, hipStream_t st) {
  M2M_CHECK_HIP(hipMemsetAsync(dst, 0, (size_t)bytes, st));
  return M2M_OK;
}

}  // namespace m2m
