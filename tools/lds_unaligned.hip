// Does the LDS serve 8- / 16-byte reads from addresses that are only 4-byte aligned on gfx950 (ROCm 7.2), and at what cost?
// (The mel stage reads 4 consecutive power bins from a per-lane, arbitrarily aligned start: one ds_read_b128 instead of four b32.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(float* out, unsigned long long* cyc, int mode) {
  __shared__ __align__(16) float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x;
  const unsigned base = (unsigned)(size_t)buf;     // LDS byte address (low 32 bits of the generic pointer's LDS offset are not this!)
  (void)base;
  float4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
    // per-lane start: 6 floats apart (like the widest filter group), + off floats of misalignment
    const int start = (lane * 6 + (mode & 3) + it) & 2047;
    const float* p = buf + start;
    float4 v;
    if (mode < 4) {          // four scalar reads
      v = make_float4(p[0], p[1], p[2], p[3]);
    } else {                 // one 16-byte read at a 4-byte-aligned address
      const unsigned addr = (unsigned)(size_t)(p) ;   // generic -> LDS offset: use the builtin below instead
      (void)addr;
      const __attribute__((address_space(3))) float* lp = (const __attribute__((address_space(3))) float*)(p);
      unsigned a32 = (unsigned)(unsigned long long)lp;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a32));
    }
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + 2 * acc.y + 3 * acc.z + 5 * acc.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
  for (int mode = 0; mode < 8; ++mode) {
    hipMemset(out, 0, 256);
    probe<<<1, 64>>>(out, cyc, mode);
    hipError_t e = hipDeviceSynchronize();
    float h[64]; unsigned long long c;
    hipMemcpy(h, out, 256, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s, misalignment %d floats): %s lane0 %.1f lane1 %.1f lane63 %.1f, %.1f cycles per read group\n", mode, mode < 4 ? "4 x b32" : "1 x b128", mode & 3,
           hipGetErrorString(e), h[0], h[1], h[63], (double)c / 256);
  }
  return 0;
}
