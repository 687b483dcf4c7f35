// Microbenchmark: cost of a per-row barrier between the H workgroups of a clip inside ONE kernel
// (candidate replacement for the kernel boundary between self- and cross-attention): each workgroup
// adds into its row with device-scope atomics, fences, bumps the row counter and polls it until all
// H siblings have arrived, then reads the row back.  Reports the time per round with the 100 MHz clock.
//   hipcc --offload-arch=gfx950 -O3 tools/row_barrier.hip -o /tmp/row_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int D = 384, ROUNDS = 16;

// grid = B * H workgroups of 1024 threads; counters [ROUNDS][B] zero on entry
__global__ __launch_bounds__(1024) void rounds(unsigned long long* rows, unsigned* counters, int B, int H, unsigned* out,
                                               unsigned long long* sink, int xcd_local, int fence_mode, int sleep_mode) {
  // xcd_local: put the H workgroups of a row on one XCD (workgroup id -> XCD is id % 8)
  int b, hh;
  if (xcd_local) { const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3; b = (j / H) * 8 + xcd; hh = j % H; }
  else { b = blockIdx.x / H; hh = blockIdx.x % H; }
  if (b >= B) return;
  __shared__ unsigned timeout;
  if (threadIdx.x == 0) timeout = 0;
  unsigned long long acc = 0;
  const unsigned long long t0 = wall_clock64();
  unsigned long long t_arrive = 0, t_pass = 0, t_read = 0;
  for (int r = 0; r < ROUNDS; ++r) {
    unsigned long long* row = rows + ((size_t)r * B + b) * D;
    if (threadIdx.x < D) atomicAdd(row + threadIdx.x, (unsigned long long)(hh + 1));
    if (fence_mode == 0) __threadfence();                                // every thread: full agent-scope fence
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // every thread: its atomics are acknowledged
    __syncthreads();
    const unsigned long long ta = wall_clock64();
    if (threadIdx.x == 0) {
      unsigned* c = counters + (size_t)r * B + b;
      if (fence_mode == 2) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else atomicAdd(c, 1u);
      int spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)H) {
        if (sleep_mode) __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) { timeout = 1; break; }   // bounded: never hang the GPU
      }
    }
    __syncthreads();
    const unsigned long long tp = wall_clock64();
    if (threadIdx.x < D) acc += __hip_atomic_load(row + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned long long tr = wall_clock64();
    t_arrive += ta; t_pass += tp; t_read += tr;
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x < D && acc != (unsigned long long)ROUNDS * H * (H + 1) / 2) sink[0] = acc;   // all rows must be complete
  if (threadIdx.x == 0) {
    unsigned* o = out + 4 * blockIdx.x;
    o[0] = (unsigned)(t1 - t0); o[1] = (unsigned)(t_pass - t_arrive); o[2] = (unsigned)(t_read - t_pass); o[3] = timeout;
  }
}

int main() {
  const int B = 32, H = 8, G = B * H;
  unsigned long long *rows, *sink; unsigned *counters, *out;
  CK(hipMalloc(&rows, (size_t)ROUNDS * B * D * 8)); CK(hipMalloc(&counters, ROUNDS * B * 4)); CK(hipMalloc(&out, G * 16)); CK(hipMalloc(&sink, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<unsigned> h(G * 4);
  for (int mode = 0; mode < 2; ++mode)
   for (int fm = 0; fm < 3; ++fm)
    for (int sm = 0; sm < 2; ++sm) {
      CK(hipMemsetAsync(rows, 0, (size_t)ROUNDS * B * D * 8, st)); CK(hipMemsetAsync(counters, 0, ROUNDS * B * 4, st)); CK(hipMemsetAsync(sink, 0, 64, st));
      hipLaunchKernelGGL(rounds, dim3(G), dim3(1024), 0, st, rows, counters, B, H, out, sink, mode, fm, sm);
      CK(hipMemcpyAsync(h.data(), out, G * 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      unsigned long long bad = 0; CK(hipMemcpy(&bad, sink, 8, hipMemcpyDeviceToHost));
      std::vector<double> tot, wait, rd; int timeouts = 0;
      for (int i = 0; i < G; ++i) { tot.push_back(h[4 * i] * 10.0 / ROUNDS); wait.push_back(h[4 * i + 1] * 10.0 / ROUNDS); rd.push_back(h[4 * i + 2] * 10.0 / ROUNDS); timeouts += h[4 * i + 3]; }
      std::sort(tot.begin(), tot.end()); std::sort(wait.begin(), wait.end()); std::sort(rd.begin(), rd.end());
      printf("fence %d sleep %d  ", fm, sm);
      printf("%s  per round: total median %.0f ns (max %.0f)   arrive->pass median %.0f ns (max %.0f)   read-back median %.0f ns   timeouts %d  bad %llu\n",
             mode ? "rows on one XCD " : "rows across XCDs", tot[G / 2], tot[G - 1], wait[G / 2], wait[G - 1], rd[G / 2], timeouts, bad);
    }
  return 0;
}
