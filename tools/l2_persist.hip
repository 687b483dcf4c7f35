// Microbenchmark: does data one kernel pulled into an XCD's L2 survive the kernel boundary?
// Kernel A (same grid, so the same workgroup -> XCD mapping) touches a buffer; kernel B then loads it
// and times the round trip per workgroup with the 100 MHz wall clock.  Cases:
//   cold     B alone after a big flush kernel
//   after A  flush, A touches, B reads            (hit only if L2 contents survive the boundary)
//   twice    B reads the same lines a second time inside one kernel (a true L2 hit, for scale)
//   hipcc --offload-arch=gfx950 -O3 tools/l2_persist.hip -o /tmp/l2_persist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int LINES = 384;   // 128-byte lines per workgroup (48 KB: one head's query weights)

__global__ void flush(const uint4* big, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += big[i].x;
  if (acc == 0x12345) sink[0] = acc;
}
__global__ void touch(const unsigned* buf, unsigned* sink) {
  const unsigned* p = buf + (size_t)blockIdx.x * LINES * 32;
  unsigned acc = 0;
  for (int l = threadIdx.x; l < LINES; l += blockDim.x) acc += p[l * 32];
  if (acc == 0x12345) sink[0] = acc;
}
// every 8-byte word of each workgroup's region gets one device-scope atomic add / one plain store
__global__ void atomic_touch(unsigned long long* buf) {
  unsigned long long* p = buf + (size_t)blockIdx.x * LINES * 16;
  for (int i = threadIdx.x; i < LINES * 16; i += blockDim.x) atomicAdd(p + i, 1ull);
}
__global__ void store_touch(unsigned long long* buf) {
  unsigned long long* p = buf + (size_t)blockIdx.x * LINES * 16;
  for (int i = threadIdx.x; i < LINES * 16; i += blockDim.x) p[i] = i;
}
// read every line first (pulls it into the Infinity Cache), then update it with atomics / stores in the same kernel
__global__ void touch_then_atomic(unsigned long long* buf, unsigned* sink, int use_store) {
  unsigned long long* p = buf + (size_t)blockIdx.x * LINES * 16;
  unsigned long long acc = 0;
  for (int l = threadIdx.x; l < LINES; l += blockDim.x) acc += p[l * 16];
  if (acc == 0x12345) sink[0] = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < LINES * 16; i += blockDim.x) {
    if (use_store) p[i] = i; else atomicAdd(p + i, 1ull);
  }
}
// out[3*wg + 0] = first-read time, [1] = second-read time (10 ns ticks)
__global__ void timed_read(const unsigned* buf, unsigned* sink, unsigned* out) {
  const unsigned* p = buf + (size_t)blockIdx.x * LINES * 32;
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int l = threadIdx.x; l < LINES; l += blockDim.x) acc += p[l * 32];
  if (acc == 0x12345) sink[0] = acc;
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  for (int l = threadIdx.x; l < LINES; l += blockDim.x) acc += p[l * 32 + 1];
  if (acc == 0x12345) sink[1] = acc;
  __syncthreads();
  const unsigned long long t2 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (unsigned)(t1 - t0); out[2 * blockIdx.x + 1] = (unsigned)(t2 - t1); }
}

int main() {
  const int G = 256;
  unsigned *buf, *sink, *out; uint4* big;
  const size_t big_n = (512u << 20) / 16;
  CK(hipMalloc(&buf, (size_t)(G + 1) * LINES * 128)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, G * 8)); CK(hipMalloc(&big, big_n * 16));
  CK(hipMemset(buf, 0, (size_t)(G + 1) * LINES * 128)); CK(hipMemset(big, 0, big_n * 16));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<unsigned> h(G * 2);
  auto report = [&](const char* name) {
    hipMemcpyAsync(h.data(), out, G * 8, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
    std::vector<unsigned> a, b;
    for (int i = 0; i < G; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-28s first read: median %4u0 ns  p90 %4u0 ns    second read: median %4u0 ns\n", name, a[G / 2], a[G * 9 / 10], b[G / 2]);
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, st, big, big_n, sink);
    hipLaunchKernelGGL(timed_read, dim3(G), dim3(384), 0, st, buf, sink, out);
    report("cold (after 512 MB flush)");
    hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, st, big, big_n, sink);
    hipLaunchKernelGGL(touch, dim3(G), dim3(384), 0, st, buf, sink);
    hipLaunchKernelGGL(timed_read, dim3(G), dim3(384), 0, st, buf, sink, out);
    report("after touch kernel");
    hipLaunchKernelGGL(timed_read, dim3(G), dim3(384), 0, st, buf, sink, out);
    report("after a previous timed_read");
    hipLaunchKernelGGL(atomic_touch, dim3(G), dim3(384), 0, st, (unsigned long long*)buf + LINES * 16);
    hipLaunchKernelGGL(timed_read, dim3(G - 1), dim3(384), 0, st, buf, sink, out);
    report("after atomics by other XCD");
    hipLaunchKernelGGL(store_touch, dim3(G), dim3(384), 0, st, (unsigned long long*)buf + LINES * 16);
    hipLaunchKernelGGL(timed_read, dim3(G - 1), dim3(384), 0, st, buf, sink, out);
    report("after stores by other XCD");
    hipLaunchKernelGGL(atomic_touch, dim3(G), dim3(384), 0, st, (unsigned long long*)buf);
    hipLaunchKernelGGL(timed_read, dim3(G), dim3(384), 0, st, buf, sink, out);
    report("after atomics by same XCD");
    hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, st, big, big_n, sink);
    hipLaunchKernelGGL(touch_then_atomic, dim3(G), dim3(384), 0, st, (unsigned long long*)buf + LINES * 16, sink, 0);
    hipLaunchKernelGGL(timed_read, dim3(G - 1), dim3(384), 0, st, buf, sink, out);
    report("read+atomics by other XCD");
    hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, st, big, big_n, sink);
    hipLaunchKernelGGL(touch_then_atomic, dim3(G), dim3(384), 0, st, (unsigned long long*)buf + LINES * 16, sink, 1);
    hipLaunchKernelGGL(timed_read, dim3(G - 1), dim3(384), 0, st, buf, sink, out);
    report("read+stores by other XCD");
    hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, st, big, big_n, sink);
    hipLaunchKernelGGL(atomic_touch, dim3(G), dim3(384), 0, st, (unsigned long long*)buf + LINES * 16);
    hipLaunchKernelGGL(timed_read, dim3(G - 1), dim3(384), 0, st, buf, sink, out);
    report("flush, atomics, read");
    // small buffer that fits the Infinity Cache but was last used by another XCD mapping: shift the grid by one
    hipLaunchKernelGGL(touch, dim3(G), dim3(384), 0, st, buf + LINES * 32, sink);
    hipLaunchKernelGGL(timed_read, dim3(G - 1), dim3(384), 0, st, buf, sink, out);
    report("after touch by other XCD");
  }
  return 0;
}
