// Issue-rate probe for the log-mel kernel's instruction mix (MI355X): cycles per wave-instruction on one SIMD for
// v_fma_f32, v_pk_fma_f32, v_pk_add_f32 (with op_sel / neg modifiers), v_mov_b32 dpp (quad_perm, row_ror), v_permlane32_swap,
// at 1, 2, 3 and 4 waves per SIMD.  Also checks the SEMANTICS of op_sel / op_sel_hi / neg_lo / neg_hi on v_pk_add_f32 and
// v_pk_fma_f32 with 64-bit f32 pairs (not in the guides: measured here).   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned long long* cyc) {
  v2f a[8], b = {1.0001f, 0.9999f}, c = {1e-9f, -1e-9f};
  for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, (float)i};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(c.y)); }
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
      if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) { asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x)); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].y)); }
      if (OP == 6) { asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x)); asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i].y)); }
      if (OP == 7) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[i].y)); }
      if (OP == 8) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y)); }
      if (OP == 9) { asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(a[i])); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void sem_kernel(float* out) {
  v2f a = {1.f, 2.f}, b = {10.f, 20.f}, c = {100.f, 200.f}, r;
  int o = 0;
  // plain
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;                                            // 11 22
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;              // expect a.x+b.y, a.y+b.x = 21 12
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;                              // expect a.x-b.x, a.y+b.y = -9 22
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;                              // expect 11 -18
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y; // a + (-i) b = (a.x+b.y, a.y-b.x) = 21 -8
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y; // a + i b = (a.x-b.y, a.y+b.x) = -19 12
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;              // a * b.x broadcast = 10 20
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;              // a * b.y broadcast = 20 40
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); out[o++] = r.x; out[o++] = r.y; // (-a.y*b.y + c.x, a.x*b.y + c.y) = 60 220
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); out[o++] = r.x; out[o++] = r.y; // same via neg on src0
  asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b)); out[o++] = r.x; out[o++] = r.y;                                // expect a.y, b.x = 2 10
}

template <int OP> static double run(int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
  rate_kernel<OP><<<blocks, 256>>>(out, 10, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate_kernel<OP><<<blocks, 256>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
  hipFree(out); hipFree(cyc);
  const int per_it = (OP == 0 || OP == 5 || OP == 6 || OP == 8) ? 16 : 8;
  // s_memtime ticks at 100 MHz on this part: report wall-clock based cycles at 2.4 GHz as well
  const double inst = (double)iters * per_it;
  printf("  %d waves/SIMD: %.2f ticks(100MHz)/inst per wave -> SIMD throughput %.2f ns per wave-inst (wall %.3f ms)\n", waves_per_simd, mean / inst,
         ms * 1e6 / (inst * waves_per_simd), ms);
  return ms;
}

int main() {
  float* o; hipMalloc(&o, 256);
  sem_kernel<<<1, 1>>>(o);
  float h[32]; hipMemcpy(h, o, 22 * 4, hipMemcpyDeviceToHost);
  const char* names[] = {"pk_add plain (11 22)", "pk_add op_sel swap src1 (21 12)", "pk_add neg_lo src1 (-9 22)", "pk_add neg_hi src1 (11 -18)",
                         "a + (-i)b (21 -8)", "a + (i)b (-19 12)", "pk_mul bcast b.x (10 20)", "pk_mul bcast b.y (20 40)", "fma (-a.y b.y + c.x, a.x b.y + c.y) (60 220)",
                         "same, neg on src0 (60 220)", "pk_mov (a.y, b.x) (2 10)"};
  for (int i = 0; i < 11; ++i) printf("SEM %-50s -> %g %g\n", names[i], h[2 * i], h[2 * i + 1]);
  const char* ops[] = {"2x v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_add_f32 op_sel+neg", "v_pk_mul_f32", "2x v_mov_b32_dpp quad_perm", "v_add_dpp quad_perm + row_ror",
                       "v_permlane32_swap", "v_add_f32 + v_mul_f32", "v_pk_mov_b32"};
  const int iters = 20000;
#define RUN(OP) printf("%s\n", ops[OP]); for (int w = 1; w <= 4; ++w) run<OP>(w, iters);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
  return 0;
}
