// Microbenchmark: can the 8 head-workgroups of a clip hand their partial rows over THROUGH THEIR XCD'S L2, without the
// memory-side atomics / kernel boundary the decode step pays today (DESIGN_HISTORY.md 4.3: ~2 us gap + 2.4 us to read the row back)?
// Workgroups find their XCD from HW_REG_XCC_ID, claim a (group, member) slot of that XCD, and every phase:
//   write a 384-float partial row to part[group][member] -> wait for the stores -> write flag[group][member] = phase ->
//   poll the 8 flags of the group -> read the 8 partial rows and add them in a fixed order (checked against the exact sum).
// The store / load cache-policy bits are the experiment: which combination is coherent between CUs of one XCD, and what does
// a phase cost, idle and with every workgroup streaming `stream_kb` of K/V-like data per phase.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier.hip -o gpurun_out/xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SPIN_CAP = 1 << 13;   // every poll loop ends; a workgroup that hits the cap stops waiting for good
constexpr int D = 384, MEMBERS = 8, GROUPS_PER_XCD = 4, NXCD = 8;

struct Args {
  unsigned* claim;      // [NXCD] slot counters (agent-scope atomics, once per launch)
  unsigned* flags;      // [NXCD * GROUPS_PER_XCD][MEMBERS] (padded to 64 B per group... 8 words = 32 B; pitch 16 words)
  float* part;          // [groups][MEMBERS][D]
  const float4* stream; // K/V-like stream, stream_kb per workgroup per phase, rotating over a big buffer
  size_t stream_f4;     // float4 elements in the stream buffer
  int stream_kb;
  int phases, smode, lmode;
  int* err;             // [0] spin cap hit, [1] wrong sum seen, [2] slot overflow (more than 32 workgroups on an XCD)
  unsigned* xcd_hist;   // [NXCD]
  float* sink;
};

template <int L> __device__ inline unsigned ld_u32(const unsigned* p) {
  unsigned v;
  if (L == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 4) asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 5) asm volatile("buffer_inv sc1\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int L> __device__ inline float ld_f32(const float* p) {
  float v;
  if (L == 0 || L == 4 || L == 5) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int S> __device__ inline void st_f32(float* p, float v) {
  if (S == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (S == 1) asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if (S == 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (S == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int S> __device__ inline void st_u32(unsigned* p, unsigned v) { st_f32<S>(reinterpret_cast<float*>(p), __uint_as_float(v)); }

template <int S, int L>
__global__ __launch_bounds__(256) void persistent(Args a) {
  __shared__ unsigned s_slot, s_xcd;
  __shared__ int s_dead;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    s_xcd = xcc;
    s_slot = __hip_atomic_fetch_add(a.claim + (xcc & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(a.xcd_hist + (xcc & 7), 1u);
    s_dead = 0;
  }
  __syncthreads();
  const unsigned xcd = s_xcd & 7, slot = s_slot;
  if (slot >= GROUPS_PER_XCD * MEMBERS) { if (threadIdx.x == 0) a.err[2] = 1; return; }   // whole workgroup leaves; its group will hit the cap
  const unsigned group = xcd * GROUPS_PER_XCD + slot / MEMBERS, member = slot % MEMBERS;
  unsigned* gflags = a.flags + group * 16;
  float* gpart = a.part + (size_t)group * MEMBERS * D;
  float acc = 0.f;
  float4 sacc = make_float4(0, 0, 0, 0);
  const size_t per_wg_f4 = (size_t)a.stream_kb * 1024 / 16;
  for (int ph = 1; ph <= a.phases; ++ph) {
    // K/V-like stream: non-temporal 16-byte loads, issued first so the barrier traffic queues behind / among them
    if (per_wg_f4) {
      size_t base = (((size_t)blockIdx.x * a.phases + ph) * per_wg_f4) % (a.stream_f4 - per_wg_f4);
      for (size_t i = threadIdx.x; i < per_wg_f4; i += blockDim.x) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.stream + base + i));
        sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
      }
    }
    // publish this member's partial row: small integers (every partial sum below is exact in fp32)
    for (int c = threadIdx.x; c < D; c += blockDim.x) st_f32<S>(gpart + member * D + c, (float)(ph * 8 + member + (c & 3)));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      st_u32<S>(gflags + member, (unsigned)ph);
      if (!s_dead) {
        int n = 0;
        for (int m = 0; m < MEMBERS; ++m)
          while (ld_u32<L>(gflags + m) < (unsigned)ph && ++n < SPIN_CAP) {}
        if (n >= SPIN_CAP) { s_dead = 1; a.err[0] = 1; }
      }
    }
    __syncthreads();
    if (L == 4 && threadIdx.x % 64 == 0) asm volatile("buffer_inv sc0" ::: "memory");
    if (L == 5 && threadIdx.x % 64 == 0) asm volatile("buffer_inv sc1" ::: "memory");
    // read the 8 partial rows back, fixed order
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      float v[MEMBERS];
      for (int m = 0; m < MEMBERS; ++m) v[m] = ld_f32<L>(gpart + m * D + c);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float s = 0.f;
      for (int m = 0; m < MEMBERS; ++m) s += v[m];
      const float want = (float)(ph * 64 + 28 + 8 * (c & 3));
      if (s != want && !s_dead) a.err[1] = 1;
      acc += s;
    }
    __syncthreads();   // nobody overwrites its partial row before every thread of this workgroup has read ... (other
                       // workgroups are held off by the NEXT phase's flags: a member only rewrites its row after passing this phase's barrier,
                       // and siblings read the rows right after it; two row buffers by phase parity make that safe)
    gpart = a.part + (size_t)(group + (ph & 1) * NXCD * GROUPS_PER_XCD) * MEMBERS * D;
  }
  if (acc + sacc.x + sacc.y + sacc.z + sacc.w == 12345.678f) a.sink[0] = acc;
}

typedef void (*kern_t)(Args);
template <int S> kern_t pick_l(int l) {
  switch (l) { case 0: return persistent<S, 0>; case 1: return persistent<S, 1>; case 2: return persistent<S, 2>;
               case 3: return persistent<S, 3>; case 4: return persistent<S, 4>; default: return persistent<S, 5>; }
}
kern_t pick(int s, int l) { switch (s) { case 0: return pick_l<0>(l); case 1: return pick_l<1>(l); case 2: return pick_l<2>(l); default: return pick_l<3>(l); } }

int main() {
  unsigned *claim, *flags, *hist; float *part, *sink; float4* stream; int* err;
  const size_t stream_bytes = (size_t)1 << 30;
  CK(hipMalloc(&claim, 256)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&hist, 256));
  CK(hipMalloc(&part, (size_t)2 * NXCD * GROUPS_PER_XCD * MEMBERS * D * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&err, 64));
  CK(hipMalloc(&stream, stream_bytes)); CK(hipMemset(stream, 0, stream_bytes));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* sn[] = {"st", "st.sc0", "st.sc1", "st.sc0sc1"};
  const char* ln[] = {"ld", "ld.sc0", "ld.sc1", "ld.sc0sc1", "inv.sc0+ld", "inv.sc1+ld"};
  const int G = NXCD * GROUPS_PER_XCD * MEMBERS;
  for (int kb : {0, 64, 220})
    for (int s = 0; s < 4; ++s)
      for (int l = 0; l < 6; ++l) {
        float us[2]; int herr[3] = {0, 0, 0}; unsigned hh[8];
        for (int k = 0; k < 2; ++k) {
          Args a{claim, flags, part, stream, stream_bytes / 16, kb, k ? 1100 : 100, s, l, err, hist, sink};
          CK(hipMemsetAsync(claim, 0, 256, st)); CK(hipMemsetAsync(flags, 0, 4096, st)); CK(hipMemsetAsync(hist, 0, 256, st));
          CK(hipMemsetAsync(err, 0, 64, st));
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(pick(s, l), dim3(G), dim3(256), 0, st, a);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[k] = ms * 1000.f;
          int e3[3]; CK(hipMemcpy(e3, err, 12, hipMemcpyDeviceToHost));
          for (int i = 0; i < 3; ++i) herr[i] |= e3[i];
          CK(hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost));
        }
        printf("stream %3d KB/WG  %-10s %-11s: %7.3f us per phase%s%s%s   xcd hist %u %u %u %u %u %u %u %u\n", kb, sn[s], ln[l], (us[1] - us[0]) / 1000.f,
               herr[0] ? "  SPIN-CAP" : "", herr[1] ? "  STALE-DATA" : "", herr[2] ? "  SLOT-OVERFLOW" : "", hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7]);
        fflush(stdout);
      }
  return 0;
}
