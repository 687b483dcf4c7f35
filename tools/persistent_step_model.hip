// Traffic-and-synchronisation MODEL of a persistent, XCD-local decode step (DESIGN_HISTORY.md 4.5: "the one restructuring the numbers do
// not rule out"): no arithmetic of the model, only what bounds it — every workgroup streams the bytes its (clip, head) role would
// stream (projection weights shared per head through the XCD's L2, the self / cross K/V of its clip, its feed-forward weight slice)
// and hands 384-float partial rows over through the XCD's L2 exactly as tools/xcd_barrier.hip measured it (plain stores, a flag
// word per member, sc1 polls and reads, fixed-order sum).  256 workgroups x 1024 threads, one per CU, 32 per XCD = 4 clips x 8
// heads; per layer: self-attention phase (8-member hand-over), cross-attention phase (8-member), feed-forward phase (publish the
// rows XCD-wide, 32-member hand-over of 4 x 384 partials).  The time per step is a LOWER bound of such a kernel (the softmax /
// projection VALU work and the lm_head are left out) — to be compared with the 199 us of the kernel-per-sub-layer product path.
//   hipcc --offload-arch=gfx950 -O3 tools/persistent_step_model.hip -o /tmp/psm && /tmp/psm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 384, H = 8, L = 6, CLIPS_PER_XCD = 4, NXCD = 8, B = CLIPS_PER_XCD * NXCD;
constexpr int SPIN_CAP = 1 << 14;

struct Args {
  const u32x4* w_self;    // [L][H][12544]      196 KB per (layer, head): q, k, v and the output-projection slice
  const u32x4* w_cross;   // [L][H][6272]        98 KB
  const u32x4* w_ff;      // [L][32][5312]       83 KB per (layer, slice)
  const u32x4* kv_self;   // [L][B][H][kv_self_v]   self K/V of (layer, clip, head) at step t (131 KB at t = 512)
  const u32x4* kv_cross;  // [L][B][H][14080]   220 KB
  int kv_self_v;
  unsigned* claim;        // [NXCD]
  unsigned* flags;        // [B][16]: 8 member flags (attention phases) ; [NXCD][64]: 32 member flags (feed-forward), separate arrays
  unsigned* flags_x;      // [NXCD][64]
  float* part;            // [2][B][H][D]          attention partial rows (two buffers by phase parity)
  float* part_ff;         // [2][NXCD][32][CLIPS_PER_XCD][D]
  float* xrows;           // [2][B][D]           rows published for the feed-forward phase
  int steps, prefetch, nt_layers;   // prefetch: request the next phase's weights before waiting for the flags; nt_layers: cross layers streamed non-temporally
  int* err;
  unsigned* sink;
};

__device__ inline unsigned ld_sc1_u32(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline float4 ld_sc1_f4(const float* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// stream n_v 16-byte vectors with the whole workgroup, two rounds per wave in flight (the decode kernels' window)
template <bool NT>
__device__ inline void stream(const u32x4* p, int n_v, u32x4& acc) {
  int i = threadIdx.x;
  u32x4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
  if (i < n_v) r0 = NT ? __builtin_nontemporal_load(p + i) : p[i];
  if (i + 1024 < n_v) r1 = NT ? __builtin_nontemporal_load(p + i + 1024) : p[i + 1024];
  for (i += 2048; i - 2048 < n_v; i += 2048) {
    acc ^= r0;
    r0 = (i < n_v) ? (NT ? __builtin_nontemporal_load(p + i) : p[i]) : u32x4{0, 0, 0, 0};
    acc ^= r1;
    r1 = (i + 1024 < n_v) ? (NT ? __builtin_nontemporal_load(p + i + 1024) : p[i + 1024]) : u32x4{0, 0, 0, 0};
  }
  acc ^= r0; acc ^= r1;
}

__global__ __launch_bounds__(1024) void persistent_step(Args a) {
  __shared__ unsigned s_xcd, s_slot;
  __shared__ int s_dead;
  __shared__ float xrow[CLIPS_PER_XCD][D];
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s_xcd = xcc & 7;
    s_slot = __hip_atomic_fetch_add(a.claim + (xcc & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dead = 0;
  }
  __syncthreads();
  const unsigned xcd = s_xcd, slot = s_slot;
  if (slot >= 32) { if (threadIdx.x == 0) a.err[2] = 1; return; }
  const int cl = slot >> 3, head = slot & 7, clip = xcd * CLIPS_PER_XCD + cl;
  u32x4 acc = {0, 0, 0, 0};
  unsigned phase = 0;
  auto wait_flags = [&](const unsigned* fl, int n, unsigned ph) {
    if (threadIdx.x == 0 && !s_dead) {
      int spins = 0;
      for (int m = 0; m < n; ++m)
        while (ld_sc1_u32(fl + m) < ph && ++spins < SPIN_CAP) {}
      if (spins >= SPIN_CAP) { s_dead = 1; a.err[0] = 1; }
    }
    __syncthreads();
  };
  for (int step = 0; step < a.steps; ++step) {
    for (int l = 0; l < L; ++l) {
      // ---- attention phases: self (kind 0), cross (kind 1) ----
      for (int kind = 0; kind < 2; ++kind) {
        ++phase;
        const u32x4* w = kind ? a.w_cross + ((size_t)l * H + head) * 6272 : a.w_self + ((size_t)l * H + head) * 12544;
        const int wv = kind ? 6272 : 12544;
        const u32x4* kv = kind ? a.kv_cross + (((size_t)l * B + clip) * H + head) * 14080
                               : a.kv_self + (((size_t)l * B + clip) * H + head) * a.kv_self_v;
        const int kvv = kind ? 14080 : a.kv_self_v;
        if (!a.prefetch) stream<false>(w, wv, acc);                 // weights: through L2, shared by the 4 clips of this XCD
        if (kind && l < a.nt_layers) stream<true>(kv, kvv, acc); else stream<false>(kv, kvv, acc);
        // partial row of this head -> siblings of the clip
        float* pr = a.part + ((size_t)(phase & 1) * B * H + (size_t)clip * H + head) * D;
        if (threadIdx.x < D) pr[threadIdx.x] = (float)((acc.x ^ threadIdx.x) & 7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* fl = a.flags + clip * 16;
        if (threadIdx.x == 0) fl[head] = phase;
        // with prefetch: the NEXT phase's weights are requested while the flags travel (they do not depend on the row)
        if (a.prefetch) {
          const int nk = kind ^ 1, nl = kind ? l : l;      // next attention phase of this layer, or the feed-forward slice
          if (kind == 0) stream<false>(a.w_cross + ((size_t)nl * H + head) * 6272, 6272, acc);
          else stream<false>(a.w_ff + ((size_t)l * 32 + slot) * 5312, 5312, acc);
          (void)nk;
        }
        wait_flags(fl, 8, phase);
        if (threadIdx.x < D / 4) {
          const float* pb = a.part + ((size_t)(phase & 1) * B * H + (size_t)clip * H) * D + threadIdx.x * 4;
          float4 v[8];
          for (int m = 0; m < 8; ++m) v[m] = ld_sc1_f4(pb + (size_t)m * D);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          float4 s = v[0];
          for (int m = 1; m < 8; ++m) { s.x += v[m].x; s.y += v[m].y; s.z += v[m].z; s.w += v[m].w; }
          reinterpret_cast<float4*>(&xrow[cl][0])[threadIdx.x] = s;
        }
        __syncthreads();
      }
      // ---- feed-forward phase: rows of the XCD's 4 clips to every workgroup, 32 slices, 32-member hand-over ----
      ++phase;
      if (head == 0 && threadIdx.x < D) a.xrows[((size_t)(phase & 1) * B + clip) * D + threadIdx.x] = xrow[cl][threadIdx.x];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned* fx = a.flags_x + xcd * 64;
      if (head == 0 && threadIdx.x == 0) fx[cl] = phase;
      if (!a.prefetch) stream<false>(a.w_ff + ((size_t)l * 32 + slot) * 5312, 5312, acc);
      wait_flags(fx, CLIPS_PER_XCD, phase);
      if (threadIdx.x < CLIPS_PER_XCD * D / 4) {
        const int c = threadIdx.x / (D / 4), j = threadIdx.x % (D / 4);
        const float4 v = ld_sc1_f4(a.xrows + ((size_t)(phase & 1) * B + xcd * CLIPS_PER_XCD + c) * D + j * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        reinterpret_cast<float4*>(&xrow[c][0])[j] = v;
      }
      __syncthreads();
      float* pf = a.part_ff + (((size_t)(phase & 1) * NXCD + xcd) * 32 + slot) * CLIPS_PER_XCD * D;
      for (int i = threadIdx.x; i < CLIPS_PER_XCD * D; i += 1024) pf[i] = xrow[i / D][i % D] * 0.03125f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned* ff = a.flags_x + xcd * 64 + 16;
      if (threadIdx.x == 0) ff[slot] = phase;
      if (a.prefetch && l + 1 < L) stream<false>(a.w_self + ((size_t)(l + 1) * H + head) * 12544, 12544, acc);
      else if (a.prefetch) stream<false>(a.w_self + (size_t)head * 12544, 12544, acc);
      wait_flags(ff, 32, phase);
      if (threadIdx.x < D / 4) {
        const float* pb = a.part_ff + ((size_t)(phase & 1) * NXCD + xcd) * 32 * CLIPS_PER_XCD * D + (size_t)cl * D + threadIdx.x * 4;
        float4 s = {0, 0, 0, 0};
        for (int m0 = 0; m0 < 32; m0 += 8) {
          float4 v[8];
          for (int m = 0; m < 8; ++m) v[m] = ld_sc1_f4(pb + (size_t)(m0 + m) * CLIPS_PER_XCD * D);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          for (int m = 0; m < 8; ++m) { s.x += v[m].x; s.y += v[m].y; s.z += v[m].z; s.w += v[m].w; }
        }
        reinterpret_cast<float4*>(&xrow[cl][0])[threadIdx.x] = s;
      }
      __syncthreads();
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.sink[0] = acc.x + (unsigned)xrow[0][0];
}

int main() {
  const int kv_self_v = 8192;                                   // 131 KB: the self K/V of one (clip, head) at t = 512
  const size_t n_ws = (size_t)L * H * 12544, n_wc = (size_t)L * H * 6272, n_wf = (size_t)L * 32 * 5312;
  const size_t n_ks = (size_t)L * B * H * kv_self_v, n_kc = (size_t)L * B * H * 14080;
  u32x4 *ws, *wc, *wf, *ks, *kc; unsigned *claim, *flags, *flags_x, *sink; float *part, *part_ff, *xrows; int* err;
  CK(hipMalloc(&ws, n_ws * 16)); CK(hipMalloc(&wc, n_wc * 16)); CK(hipMalloc(&wf, n_wf * 16)); CK(hipMalloc(&ks, n_ks * 16)); CK(hipMalloc(&kc, n_kc * 16));
  CK(hipMemset(ws, 1, n_ws * 16)); CK(hipMemset(wc, 1, n_wc * 16)); CK(hipMemset(wf, 1, n_wf * 16)); CK(hipMemset(ks, 1, n_ks * 16)); CK(hipMemset(kc, 1, n_kc * 16));
  CK(hipMalloc(&claim, 256)); CK(hipMalloc(&flags, B * 16 * 4)); CK(hipMalloc(&flags_x, NXCD * 64 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&err, 64));
  CK(hipMalloc(&part, (size_t)2 * B * H * D * 4)); CK(hipMalloc(&part_ff, (size_t)2 * NXCD * 32 * CLIPS_PER_XCD * D * 4)); CK(hipMalloc(&xrows, (size_t)2 * B * D * 4));
  printf("bytes per step: weights (per XCD) %.1f MB x 8, self K/V %.1f MB, cross K/V %.1f MB\n", (n_ws + n_wc + n_wf) * 16 / 1e6, n_ks * 16 / 1e6, n_kc * 16 / 1e6);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int prefetch = 0; prefetch < 2; ++prefetch)
    for (int nt : {0, 3, 6}) {
      float us[2]; int herr[3] = {0, 0, 0};
      for (int k = 0; k < 2; ++k) {
        Args a{ws, wc, wf, ks, kc, kv_self_v, claim, flags, flags_x, part, part_ff, xrows, k ? 72 : 8, prefetch, nt, err, sink};
        CK(hipMemsetAsync(claim, 0, 256, st)); CK(hipMemsetAsync(flags, 0, B * 16 * 4, st)); CK(hipMemsetAsync(flags_x, 0, NXCD * 64 * 4, st));
        CK(hipMemsetAsync(err, 0, 64, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persistent_step, dim3(256), dim3(1024), 0, st, a);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[k] = ms * 1000.f;
        int e3[3]; CK(hipMemcpy(e3, err, 12, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) herr[i] |= e3[i];
      }
      printf("prefetch %d  non-temporal cross layers %d: %7.1f us per step (18 hand-overs)%s%s\n", prefetch, nt, (us[1] - us[0]) / 64.f,
             herr[0] ? "  SPIN-CAP" : "", herr[2] ? "  SLOT-OVERFLOW" : "");
      fflush(stdout);
    }
  return 0;
}
