// Fused log-mel frontend for gfx950: reflect-pad + framing + Hann window + real FFT-2048
// + |X|^2 + sparse mel filterbank + clamp(1e-6) + log, one kernel, power spectrum never
// leaves the CU.  Replaces ref: music2midi/input.py:25-41 (torchaudio MelSpectrogram, i.e.
// torch.stft -> |.|^2 -> dense [1025 x n_mels] matmul -> transpose -> clamp -> log).
//
// Work decomposition
//   grid  = (ceil(frames / FR), B);   block = 256 threads = 4 waves.
//   A workgroup stages the (FR-1)*hop + 2048 padded samples its FR frames cover in LDS once
//   (frames overlap 8x at hop 256, so the waveform is re-used from LDS, not re-read), then
//   each WAVE transforms one frame at a time:
//     real FFT-2048 = complex FFT-1024 of z[n] = x[2n] + i x[2n+1], 1024 = 16 x 16 x 4:
//       radix-16 in registers (lane l holds n = 64*n1 + l)      -> twiddle W1024^(l*k1)
//       transpose through LDS (per-wave buffer, conflict-free pitch 68)
//       radix-16 in registers (lane = (k1, m2))                  -> twiddle W64^(m2*q1)
//       radix-4 ACROSS the 4 lanes of a quad with two DPP quad_perm exchanges (no LDS)
//     split/post-process pairs (k, 1024-k) into the 1025 real-FFT power bins in LDS.
//   The 4 frames' power spectra are then reduced by all 256 threads through the sparse
//   (contiguous-tap) mel filterbank, clamped, logged, and stored coalesced (n_mels floats/row).
//
// Roofline: algorithmic HBM bytes/clip = 4*T + 4*frames*n_mels; arithmetic ~56.8 MFLOP/clip
// (~26 flop/B, at the fp32-vector ridge) — see DESIGN.md.
#include "common.h"

#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace m2m {

constexpr int NFFT = 2048;
constexpr int HALF = 1024;            // complex FFT length
constexpr int FE_THREADS = 256;
constexpr int FE_WAVES = 4;
constexpr int T_PITCH = 68;           // float2 per k1 row of the transpose buffer
constexpr int WAVE_C2 = 16 * T_PITCH; // float2 per wave (transpose buffer, aliased by Z)

struct FrontendDev {
  const float* window;    // [2048]
  const float2* tw1024;   // [1024]  exp(-2 pi i k / 1024)
  const float2* tw2048;   // [1024]  exp(-2 pi i k / 2048)
  const int* fb_start;    // [n_mels] first frequency bin with a tap
  const float* fb_wpad;   // [sum_j gq[j]][64][4]: taps 4 q .. 4 q + 3 of filter 64 j + lane at ((goff[j] + q) * 64 + lane) * 4; zero beyond the filter's own taps
  int gq[8];              // 4-tap chunks walked for the 64 filters of group j (= ceil(widest filter of the group / 4)): wave-uniform trip counts
  int n_wpad;             // floats in fb_wpad
  int n_mels;
  int hop;
};

__device__ inline float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ inline float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i : (x + iy)(-i) = y - ix
__device__ inline float2 mul_negi(float2 a) { return make_float2(a.y, -a.x); }

// forward 4-point DFT, natural order in and out
__device__ inline void fft4(float2& x0, float2& x1, float2& x2, float2& x3) {
  float2 s02 = cadd(x0, x2), d02 = csub(x0, x2);
  float2 s13 = cadd(x1, x3), d13 = mul_negi(csub(x1, x3));  // -i (x1 - x3)
  x0 = cadd(s02, s13);
  x1 = cadd(d02, d13);
  x2 = csub(s02, s13);
  x3 = csub(d02, d13);
}

// forward 16-point DFT in registers: n = 4 n1 + n2, k = k1 + 4 k2.
__device__ inline void fft16(float2 (&z)[16]) {
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
  float2 a[16];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    float2 y0 = z[n2], y1 = z[4 + n2], y2 = z[8 + n2], y3 = z[12 + n2];
    fft4(y0, y1, y2, y3);
    a[n2 * 4 + 0] = y0; a[n2 * 4 + 1] = y1; a[n2 * 4 + 2] = y2; a[n2 * 4 + 3] = y3;
  }
  // twiddles W16^(n2*k1)
  a[1 * 4 + 1] = cmul(a[1 * 4 + 1], make_float2(C1, -S1));   // W^1
  a[1 * 4 + 2] = cmul(a[1 * 4 + 2], make_float2(R, -R));     // W^2
  a[1 * 4 + 3] = cmul(a[1 * 4 + 3], make_float2(S1, -C1));   // W^3
  a[2 * 4 + 1] = cmul(a[2 * 4 + 1], make_float2(R, -R));     // W^2
  a[2 * 4 + 2] = mul_negi(a[2 * 4 + 2]);                     // W^4 = -i
  a[2 * 4 + 3] = cmul(a[2 * 4 + 3], make_float2(-R, -R));    // W^6
  a[3 * 4 + 1] = cmul(a[3 * 4 + 1], make_float2(S1, -C1));   // W^3
  a[3 * 4 + 2] = cmul(a[3 * 4 + 2], make_float2(-R, -R));    // W^6
  a[3 * 4 + 3] = cmul(a[3 * 4 + 3], make_float2(-C1, S1));   // W^9
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float2 y0 = a[0 * 4 + k1], y1 = a[1 * 4 + k1], y2 = a[2 * 4 + k1], y3 = a[3 * 4 + k1];
    fft4(y0, y1, y2, y3);
    z[k1] = y0; z[k1 + 4] = y1; z[k1 + 8] = y2; z[k1 + 12] = y3;
  }
}

template <int CTRL>
__device__ inline float dpp_quad(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;  // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;  // quad_perm [2,3,0,1]

__device__ inline int zidx(int k) { return k + 4 * (k >> 8); }  // bank-spread layout of Z[0..1023]

constexpr int FE_MAXJ = 8;          // mel filters per lane (n_mels <= 512)
constexpr int FE_FBW_LDS = 6144;    // padded tap table kept in LDS when it has at most this many floats (24 KB), else read from memory

#ifndef M2M_FE_WGS_PER_CU
#define M2M_FE_WGS_PER_CU 2
#endif
template <bool TAPS_LDS>      // the padded tap table in LDS (<= FE_FBW_LDS floats: every configuration the reference uses) or read from memory
__global__ __launch_bounds__(FE_THREADS, M2M_FE_WGS_PER_CU) void logmel_kernel(
    const float* __restrict__ wav, int T, int F, FrontendDev fe, float* __restrict__ out,
    int64_t out_bstride, int row_offset, int FR, int NCH) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hop = fe.hop;
  const int span_max = (FR - 1) * hop + NFFT;

  // LDS: samples | per-wave FFT scratch (transpose buffer, then Z, then the 1025 power bins) | filter taps
  float* samples = reinterpret_cast<float*>(smem_raw);
  float2* cbase = reinterpret_cast<float2*>(smem_raw + (size_t)((span_max + 3) & ~3) * sizeof(float));
  float2* cbuf = cbase + wave * WAVE_C2;
  float* pbuf = reinterpret_cast<float*>(cbuf);          // aliases cbuf: written only after every Z read of the wave
  float2* tw2048_s = cbase + FE_WAVES * WAVE_C2;         // [1024] split/post-process twiddles, one copy per workgroup (8 KB)
  float* fbw_s = reinterpret_cast<float*>(tw2048_s + HALF);
  const int nnz = fe.n_wpad;
  constexpr bool fbw_in_lds = TAPS_LDS;

  const int b = blockIdx.y;
  if (fbw_in_lds)
    for (int i = tid; i < nnz; i += FE_THREADS) fbw_s[i] = fe.fb_wpad[i];
  for (int i = tid; i < HALF; i += FE_THREADS) tw2048_s[i] = fe.tw2048[i];   // visible after the first chunk's barrier

  // ---- per-lane constants, loaded once ----
  float2 win[16];   // window[2n], window[2n+1] for n = 64*n1 + lane
  float2 tw_a[16];  // W1024^(lane*k1)
  float2 tw_b[16];  // W64^(m2*q1) = W1024^(16*m2*q1)
  const int k1_lane = lane >> 2, m2 = lane & 3;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    win[j] = *reinterpret_cast<const float2*>(fe.window + 2 * (64 * j + lane));
    tw_a[j] = fe.tw1024[lane * j];
    tw_b[j] = fe.tw1024[16 * m2 * j];
  }
  // this lane's mel filters: m = lane + 64 j (first tap bin; the taps come from the padded per-group table)
  const int n_mels = fe.n_mels;
  int fb_s[FE_MAXJ];
#pragma unroll
  for (int j = 0; j < FE_MAXJ; ++j) fb_s[j] = fe.fb_start[min(lane + 64 * j, n_mels - 1)];

  const float sg2 = (m2 & 2) ? -1.f : 1.f, sg1 = (m2 & 1) ? -1.f : 1.f;
  const float rc = (m2 == 3) ? 0.f : 1.f, rs_ = (m2 == 3) ? 1.f : 0.f;
  // Every table load above is complete before the frame loops start: the only vector-memory operations inside them are the
  // output stores, and nothing may wait for THOSE (a lazily placed counted wait for a table register turns, from the second
  // frame on, into a wait for the previous frame's stores to be acknowledged by memory: ~1 us per frame, found in the ISA).
  __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0) only

  // A workgroup walks NCH consecutive chunks of FR frames: the ~110 table loads per lane above and the launch are
  // paid once per NCH * FR frames instead of once per FR.
  for (int ch = 0; ch < NCH; ++ch) {
  const int f0 = (blockIdx.x * NCH + ch) * FR;
  if (f0 >= F) break;                          // uniform
  const int nfr = min(FR, F - f0);
  const int span = (nfr - 1) * hop + NFFT;
  __syncthreads();                             // every wave is done with the previous chunk's samples
  // ---- stage the padded samples (reflect, as torch.stft center=True pad_mode="reflect") ----
  {
    const float* w = wav + (int64_t)b * T;
    const int base = f0 * hop - NFFT / 2;
    if (base >= 0 && base + span <= T && (span & 3) == 0 && ((reinterpret_cast<uintptr_t>(w + base) & 15) == 0)) {
      // interior chunk: no reflection, 16-byte loads, all of a thread's loads in flight at once (one round trip)
      const float4* src = reinterpret_cast<const float4*>(w + base);
      float4* dst = reinterpret_cast<float4*>(samples);
      for (int i = tid; i < span / 4; i += FE_THREADS) dst[i] = src[i];
    } else {
      for (int i = tid; i < span; i += FE_THREADS) {
        int j = base + i;
        if (j < 0) j = -j;
        else if (j >= T) j = 2 * (T - 1) - j;
        samples[i] = w[j];
      }
    }
  }
  __syncthreads();

  // Each WAVE now runs on its own: frame -> FFT -> power bins -> mel -> store, with wave-level
  // barriers only, so the four waves of the workgroup drift apart and overlap each other's phases.
  for (int fl = wave; fl < nfr; fl += FE_WAVES) {
    const float* fs = samples + fl * hop;   // 8-byte aligned: hop is even (checked on the host)
    float2 z[16];
#ifdef M2M_FE_SKIP_FFT     // diagnostic builds only: time everything but the transform
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) z[n1] = *reinterpret_cast<const float2*>(fs + 2 * (64 * n1 + lane));
    {
      const int q2 = ((m2 & 1) << 1) | (m2 >> 1);
#pragma unroll
      for (int q1 = 0; q1 < 16; ++q1) cbuf[zidx(k1_lane + 16 * q1 + 256 * q2)] = z[q1];
    }
#else
    // stage 1: lane holds n = 64*n1 + lane
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const float2 sv = *reinterpret_cast<const float2*>(fs + 2 * (64 * n1 + lane));
      z[n1] = make_float2(sv.x * win[n1].x, sv.y * win[n1].y);
    }
    fft16(z);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      float2 v = (k1 == 0) ? z[0] : cmul(z[k1], tw_a[k1]);
      cbuf[k1 * T_PITCH + lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // stage 2: lane = (k1, m2) holds l = 4*m1 + m2
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) z[m1] = cbuf[k1_lane * T_PITCH + 4 * m1 + m2];
    __builtin_amdgcn_wave_barrier();
    fft16(z);
#pragma unroll
    for (int q1 = 1; q1 < 16; ++q1) z[q1] = cmul(z[q1], tw_b[q1]);
    // stage 3: radix-4 across the quad (m2 = lane&3) by two DPP exchanges.
    //   after: lane m2 holds Y[q1 + 16*q2] with q2 = bitrev2(m2).
    //   Branch-free: v <- p + s*v with a per-lane sign (lanes 0,1: v+p; lanes 2,3: p-v), then lane 3
    //   multiplies by -i, written as a rotation with per-lane (c, sn) = (1,0) or (0,1).
#pragma unroll
    for (int q1 = 0; q1 < 16; ++q1) {
      float2 v = z[q1];
      float2 p = make_float2(dpp_quad<DPP_XOR2>(v.x), dpp_quad<DPP_XOR2>(v.y));
      v = make_float2(fmaf(sg2, v.x, p.x), fmaf(sg2, v.y, p.y));
      v = make_float2(fmaf(rc, v.x, rs_ * v.y), fmaf(rc, v.y, -rs_ * v.x));   // lane 3: (x,y) -> (y,-x)
      p = make_float2(dpp_quad<DPP_XOR1>(v.x), dpp_quad<DPP_XOR1>(v.y));
      z[q1] = make_float2(fmaf(sg1, v.x, p.x), fmaf(sg1, v.y, p.y));
    }
    {
      const int q2 = ((m2 & 1) << 1) | (m2 >> 1);
#pragma unroll
      for (int q1 = 0; q1 < 16; ++q1) cbuf[zidx(k1_lane + 16 * q1 + 256 * q2)] = z[q1];
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // post-process pairs (k, 1024-k): X[k] = E + W2048^k O, X[1024-k] = conj(E - W2048^k O).
    // All Z values are read into registers first: the power bins overwrite the Z image.
    float2 zk[8], zn[8], tw_p[8];   // tw_p: W2048^(lane + 64 i), lane-contiguous reads of the workgroup's copy (were 16 registers
#pragma unroll                      // per lane for the whole kernel: with them the kernel spilled, and a spill reload inside the
    for (int i = 0; i < 8; ++i) {   // frame loop is a vmcnt(0) wait behind the frame's output stores)
      const int k = lane + 64 * i;
      zk[i] = cbuf[zidx(k)];
      zn[i] = cbuf[zidx((HALF - k) & (HALF - 1))];
      tw_p[i] = tw2048_s[k];
    }
    const float2 z512 = cbuf[zidx(512)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = lane + 64 * i;
      const float2 c = make_float2(zn[i].x, -zn[i].y);  // conj
      const float2 e = make_float2(0.5f * (zk[i].x + c.x), 0.5f * (zk[i].y + c.y));
      const float2 d = make_float2(0.5f * (zk[i].x - c.x), 0.5f * (zk[i].y - c.y));
      const float2 o = make_float2(d.y, -d.x);  // d / i
      const float2 t = cmul(tw_p[i], o);
      const float2 xp = cadd(e, t), xm = csub(e, t);
      pbuf[k] = xp.x * xp.x + xp.y * xp.y;
      pbuf[HALF - k] = xm.x * xm.x + xm.y * xm.y;
    }
    if (lane == 0) pbuf[512] = z512.x * z512.x + z512.y * z512.y;  // k = 512: E = Re Z, O = Im Z, W2048^512 = -i
    if (lane < 48) pbuf[1025 + lane] = 0.f;                          // the padded taps of the last filters read (and ignore) these
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- mel filterbank + clamp + log, this wave's frame.  The 64 filters of a group walk the SAME number of 4-tap chunks
    //      (the widest filter of the group, rounded up; the table is zero-padded), so the trip counts are wave-uniform scalars:
    //      no exec masking.  A chunk is one 16-byte weight read (lane-contiguous: conflict-free) + four power-bin reads, and
    //      the NEXT chunk's reads are issued before this chunk's multiply-adds, so a group costs one LDS round trip plus its
    //      arithmetic instead of one round trip per tap (the first form: 39 dependent round trips per frame).
    {
      float* orow = out + (int64_t)b * out_bstride + (int64_t)(row_offset + f0 + fl) * n_mels;
      int wrow = 0;
      float res[FE_MAXJ];
#pragma unroll
      for (int j = 0; j < FE_MAXJ; ++j) {
        res[j] = 0.f;
        if (64 * j >= n_mels) break;                               // uniform
#ifdef M2M_FE_SKIP_MEL     // diagnostic builds only
        const int nq = 1;
#else
        const int nq = fe.gq[j];
#endif
        const float* p = pbuf + fb_s[j];
        // (two call sites so that each keeps its address space: a pointer that may be LDS or global compiles to flat loads)
        auto group = [&](const float4* fw) {
          float4 w = fw[0];
          float p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
          float acc = 0.f;
          for (int q = 1; q < nq; ++q) {
            const float4 wn = fw[q * 64];
            const float n0 = p[4 * q], n1 = p[4 * q + 1], n2 = p[4 * q + 2], n3 = p[4 * q + 3];
            acc = fmaf(p0, w.x, acc); acc = fmaf(p1, w.y, acc); acc = fmaf(p2, w.z, acc); acc = fmaf(p3, w.w, acc);
            w = wn; p0 = n0; p1 = n1; p2 = n2; p3 = n3;
          }
          acc = fmaf(p0, w.x, acc); acc = fmaf(p1, w.y, acc); acc = fmaf(p2, w.z, acc); acc = fmaf(p3, w.w, acc);
          return acc;
        };
        float acc;
        if constexpr (TAPS_LDS) acc = group(reinterpret_cast<const float4*>(fbw_s) + wrow * 64 + lane);
        else acc = group(reinterpret_cast<const float4*>(fe.fb_wpad) + wrow * 64 + lane);
        wrow += nq;
        res[j] = acc;
      }
      // clamp(min=1e-6).log(): the floor is the correctly rounded fp32 ln(1e-6f), so silent (zero-padded) regions are
      // bit-identical to the reference's constant.  All groups' logarithms and stores together, after the last LDS read.
#pragma unroll
      for (int j = 0; j < FE_MAXJ; ++j) {
        if (64 * j >= n_mels) break;                               // uniform
        const int m = lane + 64 * j;
        const float acc = res[j];
#ifdef M2M_FE_SKIP_LOG      // diagnostic builds only
        if (m < n_mels) orow[m] = acc;
#elif defined(M2M_FE_SKIP_STORE)
        if (m < n_mels && acc == 12345.678f) orow[m] = acc;
#else
        if (m < n_mels) orow[m] = (acc > 1e-6f) ? logf(acc) : -13.815510749816895f;
#endif
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the next frame's transpose writes reuse this buffer
  }
  }  // chunks
}


// ---------------------------------------------------------------------------------------------------------------------
// Second form of the same transform (round 4): ONE workgroup of WAVES waves per CU, a frame per wave per chunk.
//
// What the first form above cannot have is occupancy: its window / twiddle tables sit in 112 VGPRs per lane and its
// 16 x 68 complex transpose buffer takes 8.7 KB of LDS per wave: two waves per SIMD.  tools/valu_rate.hip measures what that
// costs on this machine before any latency is counted: a SIMD issues one plain vector instruction every 2.9 cycles with two
// resident waves and every 1.96 with four (packed-f32: 5.0 -> 3.3).  Here
//   * every table lives ONCE in LDS per workgroup (window pairs, stage twiddles in lane order, split twiddles, tap table, filter
//     starts: 34 KB), read lane-contiguously (ds_read_b64) right where it is used;
//   * the two exchanges go through LDS one PLANE at a time (real parts, then imaginary parts, through the same 16 x 68 floats):
//     4.25 KB per wave instead of 8.7, and the 1 025 power bins alias it; the pitch / padding make every access conflict-free;
//   * 16 waves x <= 128 VGPRs = four waves per SIMD;
//   * a workgroup walks NCH consecutive chunks of its clip with the NEXT chunk's samples in flight (requested into registers when
//     a chunk starts, written to the other sample buffer when it ends: one barrier per chunk, no wave waits for memory).
// LDS (hop 256, 16 waves): 2 x 23.0 KB samples + 16 x 4.25 KB planes + 34 KB tables = 148 KB.
#ifndef M2M_FE2_SKIP        // diagnostic builds only (timing of the kernel with a part removed; results are wrong):
#define M2M_FE2_SKIP 0      // 1 mel taps, 2 exchange 1, 4 exchange 2, 8 the two radix-16 passes, 16 table reads, 32 logf, 64 stores, 128 DPP stage
#endif
// Lanes of a wave exchange data through LDS here.  The hardware executes one wave's LDS operations in order, so no instruction is
// needed between a write phase and the dependent read phase (or a read phase and the writes that overwrite what it read) — but
// the COMPILER must not move a ds_read above the ds_write of another lane's value: a wavefront-scope release fence plus
// wave_barrier emits nothing and pins the order (the first form above does the same at every exchange).
#ifdef M2M_FE2_NOFENCE      // diagnostic builds only (the round-4 kernel, for the same-box A/B of the fences' cost)
#define V2_LANES_SYNC() do {} while (0)
#else
#define V2_LANES_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
constexpr int V2_PITCH = 68;                 // floats per k1 row of a transpose plane
constexpr int V2_SCR = 16 * V2_PITCH;        // floats of per-wave scratch (planes, then the 1 025 + 48 power bins)
constexpr int V2_NPRE = 6;                   // prefetch registers per thread (samples of the next chunk: 5 888 / 1 024 threads at hop 256)
__device__ inline int zidx8(int k) { return k + 8 * (k >> 8); }   // plane layout of Z[0..1023]: lanes (k1, q2) hit 32 different banks

struct V2Layout {      // float offsets into dynamic LDS (host and device agree through this one function)
  int span_pad, samples1, scr, win2, twa, twb, twp, fbs, fbw, total;
};
__host__ __device__ inline V2Layout v2_layout(int waves, int hop, int n_wpad) {
  V2Layout l;
  l.span_pad = (((waves - 1) * hop + NFFT) + 3) & ~3;
  l.samples1 = l.span_pad;
  l.scr = 2 * l.span_pad;
  l.win2 = l.scr + waves * V2_SCR;
  l.twa = l.win2 + 2 * HALF;        // float2 [1024]
  l.twb = l.twa + 2 * 16 * 64;      // float2 [16][64]
  l.twp = l.twb + 2 * 16 * 4;       // float2 [16][4]
  l.fbs = l.twp + 2 * 512;          // float2 [512]
  l.fbw = l.fbs + 512;              // int [512]
  l.total = l.fbw + ((n_wpad + 3) & ~3);
  return l;
}

// ln(x) for x > 1e-6: the hardware log2 (1 ulp) times ln 2 — absolute error <= |log2 x| 2^-23 ln 2 + one rounding < 2e-6 over the
// 20 octaves above the clamp, against the 1e-4 the frontend is held to; the accurate logf is ~20 instructions per value.
__device__ inline float fe_log(float x) {
#ifdef M2M_FE_ACCURATE_LOG
  return logf(x);
#else
  return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
#endif
}

template <int WAVES, int NJ>     // NJ: groups of 64 mel filters (6 for the reference's 384; 8 = the general form, n_mels <= 512)
__global__ __launch_bounds__(64 * WAVES) void logmel_v2_kernel(const float* __restrict__ wav, int T, int F, FrontendDev fe,
                                                               float* __restrict__ out, int64_t out_bstride, int row_offset, int NCH) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int NT = 64 * WAVES, FR = WAVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = fe.hop, n_mels = fe.n_mels;
  const V2Layout L = v2_layout(WAVES, hop, fe.n_wpad);
  float* const lds = reinterpret_cast<float*>(smem_raw);
  float* const scr = lds + L.scr + wave * V2_SCR;
  const float2* const win2_s = reinterpret_cast<const float2*>(lds + L.win2);
  const float2* const twa_s = reinterpret_cast<const float2*>(lds + L.twa);
  const float2* const twb_s = reinterpret_cast<const float2*>(lds + L.twb);
  const float2* const twp_s = reinterpret_cast<const float2*>(lds + L.twp);
  const int* const fbs_s = reinterpret_cast<const int*>(lds + L.fbs);
  const float* const fbw_s = lds + L.fbw;
  const int b = blockIdx.y;
  const float* const w = wav + (int64_t)b * T;
  const int m2 = lane & 3;

  // samples of chunk f0 -> registers (reflect padding as torch.stft center=True, pad_mode="reflect"); element i = tid + NT * u
  auto fetch = [&](int f0, float (&pre)[V2_NPRE]) {
    const int nfr = min(FR, F - f0), span = (nfr - 1) * hop + NFFT, base = f0 * hop - NFFT / 2;
#pragma unroll
    for (int u = 0; u < V2_NPRE; ++u) {
      const int i = tid + NT * u;
      int j = base + min(i, span - 1);
      if (j < 0) j = -j;
      else if (j >= T) j = 2 * (T - 1) - j;
      pre[u] = w[j];
    }
  };
  auto deposit = [&](float* dst, const float (&pre)[V2_NPRE]) {
#pragma unroll
    for (int u = 0; u < V2_NPRE; ++u) {
      const int i = tid + NT * u;
      if (i < L.span_pad) dst[i] = pre[u];
    }
  };

  int f0 = blockIdx.x * NCH * FR;
  if (f0 >= F) return;                                           // uniform
  float pre[V2_NPRE];
  fetch(f0, pre);
  // ---- tables, once per workgroup ----
  {
    float2* win2_w = reinterpret_cast<float2*>(lds + L.win2);
    float2* twa_w = reinterpret_cast<float2*>(lds + L.twa);
    float2* twb_w = reinterpret_cast<float2*>(lds + L.twb);
    float2* twp_w = reinterpret_cast<float2*>(lds + L.twp);
    int* fbs_w = reinterpret_cast<int*>(lds + L.fbs);
    float* fbw_w = lds + L.fbw;
    for (int i = tid; i < HALF; i += NT) {
      win2_w[i] = *reinterpret_cast<const float2*>(fe.window + 2 * i);
      twa_w[i] = fe.tw1024[(i & 63) * (i >> 6)];                 // [k1][lane] = W1024^(lane k1)
    }
    for (int i = tid; i < 64; i += NT) twb_w[i] = fe.tw1024[16 * (i & 3) * (i >> 2)];   // [q1][m2] = W64^(m2 q1)
    for (int i = tid; i < 512; i += NT) {
      twp_w[i] = fe.tw2048[i];
      fbs_w[i] = fe.fb_start[min(i, n_mels - 1)];
    }
    for (int i = tid; i < fe.n_wpad; i += NT) fbw_w[i] = fe.fb_wpad[i];
  }
  deposit(lds, pre);
  __syncthreads();

  const float sg2 = (m2 & 2) ? -1.f : 1.f, sg1 = (m2 & 1) ? -1.f : 1.f;
  const float rc = (m2 == 3) ? 0.f : 1.f, rs_ = (m2 == 3) ? 1.f : 0.f;

  int fstart[NJ];                                                // this lane's filter starts: NJ registers instead of NJ LDS reads per frame
#pragma unroll
  for (int j = 0; j < NJ; ++j) fstart[j] = fbs_s[lane + 64 * j];
  for (int ch = 0; ch < NCH; ++ch, f0 += FR) {
    if (f0 >= F) break;                                          // uniform
    const float* const samples = lds + ((ch & 1) ? L.samples1 : 0);
    const bool more = (ch + 1 < NCH) && (f0 + FR < F);           // uniform
    if (more) fetch(f0 + FR, pre);                               // in flight under this chunk's transform
    const int fl = wave;
    if (f0 + fl < F) {                                           // wave-uniform
      // An opaque per-frame copy of the lane id: every per-lane LDS address below is ONE base register plus compile-time offsets,
      // formed inside the frame.  Derived from threadIdx they are loop invariants: the compiler hoists ~40 of them (and the filter
      // starts it finds in LDS) out of the chunk loop, runs out of the 128 registers four waves per SIMD allow, and reloads its
      // spills in every frame — a vmcnt(0) wait behind the previous frame's output stores each time.
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int k1l = ln >> 2, m2l = ln & 3, q2l = ((m2l & 1) << 1) | (m2l >> 1);
      const float2* const sp = reinterpret_cast<const float2*>(samples + fl * hop) + ln;   // 8-byte aligned: hop is even (checked on the host)
      const float2* const wp = win2_s + ln;
      const float2* const tap = twa_s + ln;
      const float2* const tbp = twb_s + m2l;
      float* const sw = scr + ln;                                // plane rows: [k1][lane]
      const float* const sr = scr + k1l * V2_PITCH + m2l;        // lane (k1, m2) reads l = 4 m1 + m2 of its row
      float* const zw = scr + k1l + 264 * q2l;                   // zidx8(k1 + 16 q1 + 256 q2) = k1 + 264 q2 + 16 q1
      const float* const zr = scr + ln;                          // zidx8(lane + 64 i) = lane + 64 i + 8 (i >> 2)
      const float* const znr = scr + (1048 - 456) - ln;          // zidx8(1024 - lane - 64 i) = 1048 - lane - 64 i - 8 (i >> 2)  (lane >= 1)
      float2 z[16];
      // stage 1: lane holds n = 64 n1 + lane
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const float2 sv = sp[64 * n1];
        const float2 wv = (M2M_FE2_SKIP & 16) ? make_float2(0.5f, 0.25f) : wp[64 * n1];
        z[n1] = make_float2(sv.x * wv.x, sv.y * wv.y);
      }
      if (!(M2M_FE2_SKIP & 8)) fft16(z);
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], (M2M_FE2_SKIP & 16) ? make_float2(0.6f, 0.8f) : tap[64 * k1]);
      // exchange 1, one plane at a time: lane (k1, m2) takes l = 4 m1 + m2 of row k1
      if (!(M2M_FE2_SKIP & 2)) {
      float zre[16];
      V2_LANES_SYNC();                                             // the previous frame's mel reads are done with the scratch
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) sw[k1 * V2_PITCH] = z[k1].x;
      V2_LANES_SYNC();
#pragma unroll
      for (int m1 = 0; m1 < 16; ++m1) zre[m1] = sr[4 * m1];
      V2_LANES_SYNC();
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) sw[k1 * V2_PITCH] = z[k1].y;
      V2_LANES_SYNC();
#pragma unroll
      for (int m1 = 0; m1 < 16; ++m1) z[m1] = make_float2(zre[m1], sr[4 * m1]);
      V2_LANES_SYNC();
      }
      if (!(M2M_FE2_SKIP & 8)) fft16(z);
#pragma unroll
      for (int q1 = 1; q1 < 16; ++q1) z[q1] = cmul(z[q1], (M2M_FE2_SKIP & 16) ? make_float2(0.6f, 0.8f) : tbp[4 * q1]);
      // stage 3: radix-4 across the quad (m2 = lane & 3) by two DPP exchanges; after it lane m2 holds Y[q1 + 16 q2], q2 = bitrev2(m2)
#pragma unroll
      for (int q1 = 0; q1 < ((M2M_FE2_SKIP & 128) ? 0 : 16); ++q1) {
        float2 v = z[q1];
        float2 pp = make_float2(dpp_quad<DPP_XOR2>(v.x), dpp_quad<DPP_XOR2>(v.y));
        v = make_float2(fmaf(sg2, v.x, pp.x), fmaf(sg2, v.y, pp.y));
        v = make_float2(fmaf(rc, v.x, rs_ * v.y), fmaf(rc, v.y, -rs_ * v.x));   // lane 3: (x, y) -> (y, -x)
        pp = make_float2(dpp_quad<DPP_XOR1>(v.x), dpp_quad<DPP_XOR1>(v.y));
        z[q1] = make_float2(fmaf(sg1, v.x, pp.x), fmaf(sg1, v.y, pp.y));
      }
      // exchange 2 (Z image -> pairs (k, 1024 - k), k = lane + 64 i), again by planes.  Lane 0's partners are irregular: k = 0
      // pairs with itself, k = 256 with Z[768], which sits behind one more padding step than the other lanes' partners.
      float2 zk[8], zn[8];
      // Z[512] and Z[768] are register 0 of lanes 1 and 3 (k = k1 + 16 q1 + 256 q2, q2 = bitrev2(m2)): lane 0, the only lane that
      // needs them, takes them with v_readlane instead of four more LDS reads per frame
      const float z512x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[0].x), 1));
      const float z512y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[0].y), 1));
      const float z768x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[0].x), 3));
      const float z768y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[0].y), 3));
      if (M2M_FE2_SKIP & 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { zk[i] = z[i]; zn[i] = z[8 + i]; }
      } else {
#pragma unroll
      for (int q1 = 0; q1 < 16; ++q1) zw[16 * q1] = z[q1].x;
      V2_LANES_SYNC();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        zk[i].x = zr[64 * i + 8 * (i >> 2)];
        zn[i].x = znr[456 - 64 * i - 8 * (i >> 2)];
      }
      V2_LANES_SYNC();
#pragma unroll
      for (int q1 = 0; q1 < 16; ++q1) zw[16 * q1] = z[q1].y;
      V2_LANES_SYNC();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        zk[i].y = zr[64 * i + 8 * (i >> 2)];
        zn[i].y = znr[456 - 64 * i - 8 * (i >> 2)];
      }
      V2_LANES_SYNC();                                             // every lane has its pairs: the power bins may overwrite the planes
      }
      if (ln == 0) { zn[0] = zk[0]; zn[4] = make_float2(z768x, z768y); }
      // split / post-process: X[k] = E + W2048^k O, X[1024 - k] = conj(E - W2048^k O); the power bins overwrite the planes
      // (the LDS executes a wave's operations in order and V2_LANES_SYNC pins that order for the compiler)
      float* const pk = scr + ln;                                  // bins lane + 64 i
      float* const pn = scr + 1024 - ln;                           // bins 1024 - lane - 64 i
      const float2* const tpp = twp_s + ln;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float2 c = make_float2(zn[i].x, -zn[i].y);  // conj
        const float2 e = make_float2(0.5f * (zk[i].x + c.x), 0.5f * (zk[i].y + c.y));
        const float2 d = make_float2(0.5f * (zk[i].x - c.x), 0.5f * (zk[i].y - c.y));
        const float2 o = make_float2(d.y, -d.x);  // d / i
        const float2 t = cmul((M2M_FE2_SKIP & 16) ? make_float2(0.6f, 0.8f) : tpp[64 * i], o);
        const float2 xp = cadd(e, t), xm = csub(e, t);
        pk[64 * i] = xp.x * xp.x + xp.y * xp.y;
        pn[-64 * i] = xm.x * xm.x + xm.y * xm.y;
      }
      if (ln == 0) scr[512] = z512x * z512x + z512y * z512y;      // k = 512: E = Re Z, O = Im Z, W2048^512 = -i
      if (ln < 48) scr[1025 + ln] = 0.f;                           // the padded taps of the last filters read (and ignore) these
      V2_LANES_SYNC();                                             // the mel taps below read bins other lanes wrote
      // mel filterbank (4-tap chunks: one aligned 16-byte weight read + four power-bin reads; the next chunk is requested before
      // this chunk's multiply-adds) + clamp + log.  The power-bin reads are dword reads from per-lane starts a few bins apart:
      // they collide on the 32 banks a b32 read sees (272 LDS cycles per frame for 48 reads = every bank conflict the kernel has,
      // SQ_LDS_BANK_CONFLICT).  The obvious remedy was built and measured: ONE ds_read_b128 per chunk from the 4-byte-aligned start
      // (gfx950 serves it — tools/lds_unaligned.hip — hipcc splits such a load, so it was issued by inline asm; 12 reads on 64
      // banks, ~100 cycles by the bank model): results identical, kernel 158 -> 190 us.  A misaligned wide LDS read is replayed
      // (a single wave sees 192 cycles per read against 101 for the four dword reads).  Dword reads stay.
      float* orow = out + (int64_t)b * out_bstride + (int64_t)(row_offset + f0 + fl) * n_mels + ln;
      float res[NJ];
      {
        const float4* fw = reinterpret_cast<const float4*>(fbw_s) + ln;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          res[j] = 0.f;
          if (64 * j >= n_mels) break;                             // uniform
          const int nq = (M2M_FE2_SKIP & 1) ? 1 : fe.gq[j];
          const float* p = scr + fstart[j];
          float4 wq = fw[0];
          float p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
          float acc = 0.f;
          for (int q = 1; q < nq; ++q) {
            const float4 wn = fw[q * 64];
            const float n0 = p[4 * q], n1 = p[4 * q + 1], n2 = p[4 * q + 2], n3 = p[4 * q + 3];
            acc = fmaf(p0, wq.x, acc); acc = fmaf(p1, wq.y, acc); acc = fmaf(p2, wq.z, acc); acc = fmaf(p3, wq.w, acc);
            wq = wn; p0 = n0; p1 = n1; p2 = n2; p3 = n3;
          }
          acc = fmaf(p0, wq.x, acc); acc = fmaf(p1, wq.y, acc); acc = fmaf(p2, wq.z, acc); acc = fmaf(p3, wq.w, acc);
          fw += nq * 64;
          res[j] = acc;
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (64 * j >= n_mels) break;                               // uniform
        // clamp(min=1e-6).log(): the floor is the correctly rounded fp32 ln(1e-6f), so silence is bit-identical to the reference's constant
        if (M2M_FE2_SKIP & 64) { if (ln + 64 * j < n_mels && res[j] == 12345.678f) orow[64 * j] = res[j]; }
        else if (M2M_FE2_SKIP & 32) { if (ln + 64 * j < n_mels) orow[64 * j] = res[j]; }
        else if (ln + 64 * j < n_mels) orow[64 * j] = (res[j] > 1e-6f) ? fe_log(res[j]) : -13.815510749816895f;
      }
    }
    if (more) {
      deposit(lds + ((ch & 1) ? 0 : L.samples1), pre);           // the buffer chunk ch - 1 used: every wave left it a barrier ago
      __syncthreads();
    }
  }
}

struct CondArgs {
  const float* tables[8];
  int rows[8];
};

__global__ void cond_rows_kernel(CondArgs a, int n_tables, int n_dim, const int64_t* __restrict__ idx,
                                 float* __restrict__ out, int64_t out_bstride) {
  const int b = blockIdx.x / n_tables;
  const int i = blockIdx.x - b * n_tables;
  const int64_t id = idx[(int64_t)b * n_tables + i];
  const bool ok = id >= 0 && id < a.rows[i];
  const float* src = a.tables[i] + (ok ? id : 0) * n_dim;
  float* dst = out + (int64_t)b * out_bstride + (int64_t)i * n_dim;
  for (int d = threadIdx.x; d < n_dim; d += blockDim.x) dst[d] = ok ? src[d] : __builtin_nanf("");
}

}  // namespace m2m

// ------------------------------------------------------------------ C ABI ---
using namespace m2m;

struct m2m_frontend {
  int n_fft, hop, n_freqs, n_mels, nnz, n_wpad;
  void* dev_blob = nullptr;  // one allocation holding every table
  FrontendDev dev;
};

extern "C" int m2m_frontend_create(const m2m_frontend_desc* d, m2m_frontend** out) {
  M2M_REQUIRE(d && out, "m2m_frontend_create: null argument");
  M2M_REQUIRE(d->n_fft == NFFT, "m2m_frontend_create: n_fft=%d unsupported (kernel is specialised for 2048, ref config.yaml:12)", d->n_fft);
  M2M_REQUIRE(d->n_freqs == NFFT / 2 + 1, "m2m_frontend_create: n_freqs must be n_fft/2+1");
  M2M_REQUIRE(d->hop_length >= 2 && d->hop_length <= NFFT / 2 && d->hop_length % 2 == 0,
              "m2m_frontend_create: hop_length=%d must be even and in [2, n_fft/2] (the kernel reads sample pairs)", d->hop_length);
  M2M_REQUIRE(d->n_mels <= 64 * 8, "m2m_frontend_create: n_mels=%d > 512 unsupported", d->n_mels);
  M2M_REQUIRE(d->n_mels >= 1 && d->window_host && d->fb_host, "m2m_frontend_create: bad n_mels / null tables");

  const int n_mels = d->n_mels, n_freqs = d->n_freqs;
  std::vector<int> start(n_mels), count(n_mels);
  int nnz_true = 0;
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < n_freqs; ++k)
      if (d->fb_host[(size_t)k * n_mels + m] != 0.0f) { if (lo < 0) lo = k; hi = k; nnz_true++; }
    start[m] = lo < 0 ? 0 : lo;
    count[m] = lo < 0 ? 0 : hi - lo + 1;
  }
  // per group of 64 filters: the widest filter's tap count in 4-tap chunks; the table holds taps 4 q .. 4 q + 3 of filter
  // 64 j + lane at ((goff[j] + q) * 64 + lane) * 4, zero where a tap is past the filter's own (interior zeros stay zeros)
  int gq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, goff[9] = {0};
  for (int m = 0; m < n_mels; ++m) gq[m / 64] = std::max(gq[m / 64], (count[m] + 3) / 4);
  for (int j = 0; j < 8; ++j) { if (64 * j < n_mels) gq[j] = std::max(gq[j], 1); goff[j + 1] = goff[j] + gq[j]; }
  M2M_REQUIRE(4 * *std::max_element(gq, gq + 8) <= 48,
              "m2m_frontend_create: a mel filter spans more than 48 frequency bins (the kernel keeps 48 spare power bins for the padded taps)");
  std::vector<float> w((size_t)goff[8] * 256, 0.0f);
  for (int m = 0; m < n_mels; ++m)
    for (int c = 0; c < count[m]; ++c)
      w[(((size_t)goff[m / 64] + c / 4) * 64 + (m % 64)) * 4 + (c % 4)] = d->fb_host[(size_t)(start[m] + c) * n_mels + m];
  std::vector<float2> tw1(HALF), tw2(HALF);
  for (int k = 0; k < HALF; ++k) {
    double a1 = -2.0 * M_PI * k / 1024.0, a2 = -2.0 * M_PI * k / 2048.0;
    tw1[k] = make_float2((float)cos(a1), (float)sin(a1));
    tw2[k] = make_float2((float)cos(a2), (float)sin(a2));
  }
  // one blob: window | tw1024 | tw2048 | start | padded weights
  size_t o_win = 0;
  size_t o_tw1 = o_win + NFFT * sizeof(float);
  size_t o_tw2 = o_tw1 + HALF * sizeof(float2);
  size_t o_st = o_tw2 + HALF * sizeof(float2);
  size_t o_w = (size_t)align_up((int64_t)(o_st + (size_t)n_mels * sizeof(int)), 16);
  size_t total = o_w + (w.size() + 4) * sizeof(float);
  std::vector<unsigned char> host(total, 0);
  memcpy(host.data() + o_win, d->window_host, NFFT * sizeof(float));
  memcpy(host.data() + o_tw1, tw1.data(), HALF * sizeof(float2));
  memcpy(host.data() + o_tw2, tw2.data(), HALF * sizeof(float2));
  memcpy(host.data() + o_st, start.data(), (size_t)n_mels * sizeof(int));
  if (!w.empty()) memcpy(host.data() + o_w, w.data(), w.size() * sizeof(float));

  m2m_frontend* fe = new m2m_frontend();
  fe->n_fft = d->n_fft; fe->hop = d->hop_length; fe->n_freqs = n_freqs; fe->n_mels = n_mels;
  fe->nnz = nnz_true;
  fe->n_wpad = (int)w.size();
  hipError_t e = hipMalloc(&fe->dev_blob, total);
  if (e == hipSuccess) e = hipMemcpy(fe->dev_blob, host.data(), total, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("m2m_frontend_create: device allocation/copy failed: %s", hipGetErrorString(e));
    if (fe->dev_blob) (void)hipFree(fe->dev_blob);
    delete fe;
    return M2M_ERR_HIP;
  }
  unsigned char* base = (unsigned char*)fe->dev_blob;
  fe->dev.window = (const float*)(base + o_win);
  fe->dev.tw1024 = (const float2*)(base + o_tw1);
  fe->dev.tw2048 = (const float2*)(base + o_tw2);
  fe->dev.fb_start = (const int*)(base + o_st);
  fe->dev.fb_wpad = (const float*)(base + o_w);
  for (int j = 0; j < 8; ++j) fe->dev.gq[j] = gq[j];
  fe->dev.n_wpad = (int)w.size();
  fe->dev.n_mels = n_mels;
  fe->dev.hop = d->hop_length;
  *out = fe;
  return M2M_OK;
}

extern "C" void m2m_frontend_destroy(m2m_frontend* fe) {
  if (!fe) return;
  if (fe->dev_blob) (void)hipFree(fe->dev_blob);
  delete fe;
}

extern "C" int m2m_frontend_num_frames(const m2m_frontend* fe, int n_samples) {
  M2M_REQUIRE(fe, "m2m_frontend_num_frames: null frontend");
  M2M_REQUIRE(n_samples >= fe->n_fft / 2 + 1,
              "m2m_frontend_num_frames: n_samples=%d too short for reflect padding (need > n_fft/2 = %d)",
              n_samples, fe->n_fft / 2);
  return 1 + n_samples / fe->hop;
}

extern "C" int m2m_frontend_fb_nnz(const m2m_frontend* fe) { return fe ? fe->nnz : M2M_ERR_INVALID; }

static size_t frontend_smem_bytes(int FR, int hop, int nnz) {
  size_t span = (size_t)(FR - 1) * hop + NFFT;
  span = (span + 3) & ~(size_t)3;
#ifdef M2M_FE_TAPS_GLOBAL
  nnz = FE_FBW_LDS + 1;
#endif
  return span * sizeof(float) + (size_t)FE_WAVES * WAVE_C2 * sizeof(float2) + (size_t)HALF * sizeof(float2) +
         (nnz <= FE_FBW_LDS ? (size_t)nnz * sizeof(float) : 0);
}

extern "C" int m2m_logmel_f32(const m2m_frontend* fe, const float* wav_dev, int B, int T, float* out_dev,
                              int64_t out_batch_stride, int row_offset, void* stream) {
  M2M_REQUIRE(fe && wav_dev && out_dev, "m2m_logmel_f32: null argument");
  M2M_REQUIRE(B >= 1 && B <= 65535, "m2m_logmel_f32: batch %d out of range", B);
  M2M_REQUIRE(T >= fe->n_fft / 2 + 1, "m2m_logmel_f32: T=%d too short for reflect padding (need > %d)", T, fe->n_fft / 2);
  M2M_REQUIRE(row_offset >= 0, "m2m_logmel_f32: negative row_offset");
  const int F = 1 + T / fe->hop;
  M2M_REQUIRE(out_batch_stride >= (int64_t)(row_offset + F) * fe->n_mels,
              "m2m_logmel_f32: out_batch_stride %lld smaller than (row_offset+frames)*n_mels", (long long)out_batch_stride);
  // Second form (one 16-wave workgroup per CU, tables and half-plane exchanges in LDS, four waves per SIMD) whenever its LDS image
  // fits: hop <= 272 (15 hop + 2048 samples in 6 prefetch registers per thread) and the tap table inside 160 KB, i.e. every configuration
  // the reference uses (hop 256, 384 mels) and e.g. hop 128.  Longer hops, and M2M_FE_V2=0, run the first form.
  {
    constexpr int WAVES = 16;
    const V2Layout L = v2_layout(WAVES, fe->hop, fe->n_wpad);
    static const bool v2_on = [] { const char* v = getenv("M2M_FE_V2"); return !(v && v[0] == '0'); }();
    if (v2_on && (size_t)L.total * sizeof(float) <= 160 * 1024 && L.span_pad <= V2_NPRE * 64 * WAVES && fe->n_mels <= 512) {
      const int cpc = ceil_div(F, WAVES);                              // chunks per clip
      static const int n_cu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
      int per_clip = (n_cu + B / 2) / B;                               // workgroups per clip: about one per CU in total
      per_clip = per_clip < 1 ? 1 : (per_clip > cpc ? cpc : per_clip);
      int NCH = ceil_div(cpc, per_clip);
      if (const char* v = getenv("M2M_FE_CHUNKS")) NCH = atoi(v) > 0 ? atoi(v) : NCH;
      dim3 grid((unsigned)ceil_div(cpc, NCH), (unsigned)B);
      if (fe->n_mels <= 384) {
        M2M_OPT_IN_LDS((logmel_v2_kernel<WAVES, 6>), 160 * 1024);
        hipLaunchKernelGGL((logmel_v2_kernel<WAVES, 6>), grid, dim3(64 * WAVES), (size_t)L.total * sizeof(float), (hipStream_t)stream, wav_dev, T, F,
                           fe->dev, out_dev, out_batch_stride, row_offset, NCH);
      } else {
        M2M_OPT_IN_LDS((logmel_v2_kernel<WAVES, 8>), 160 * 1024);
        hipLaunchKernelGGL((logmel_v2_kernel<WAVES, 8>), grid, dim3(64 * WAVES), (size_t)L.total * sizeof(float), (hipStream_t)stream, wav_dev, T, F,
                           fe->dev, out_dev, out_batch_stride, row_offset, NCH);
      }
      M2M_CHECK_HIP(hipGetLastError());
      return M2M_OK;
    }
  }
  // 16 frames per workgroup: waveform re-read factor 1.44 at hop 256, two workgroups per CU.
  int FR = getenv("M2M_FE_FR") ? atoi(getenv("M2M_FE_FR")) : 16;
  while (FR > 4 && frontend_smem_bytes(FR, fe->hop, fe->n_wpad) > 80 * 1024) FR -= 4;      // two workgroups per CU (160 KB)
  const size_t smem = frontend_smem_bytes(FR, fe->hop, fe->n_wpad);
  // chunks of FR frames per workgroup.  Measured on MI355X (B = 64 / 32, us per launch): 1 chunk 199.9 / 108.3,
  // 2 chunks 196.0 / 112.5, 3 chunks 208.1 / 136.9: the table loads are not what bounds the kernel, so one chunk
  // (most workgroups, best balance) unless a launch has thousands of workgroups to spare.
  int NCH = ((int64_t)ceil_div(F, FR * 2) * B >= 1536) ? 2 : 1;
  if (const char* v = getenv("M2M_FE_CHUNKS")) NCH = atoi(v) > 0 ? atoi(v) : NCH;
  dim3 grid((unsigned)ceil_div(F, FR * NCH), (unsigned)B);
#ifdef M2M_FE_TAPS_GLOBAL       // diagnostic builds only
  const bool taps_lds = false;
#else
  const bool taps_lds = fe->n_wpad <= FE_FBW_LDS;
#endif
  if (taps_lds) {
    M2M_OPT_IN_LDS(logmel_kernel<true>, 160 * 1024);
    hipLaunchKernelGGL(logmel_kernel<true>, grid, dim3(FE_THREADS), smem, (hipStream_t)stream, wav_dev, T, F, fe->dev,
                       out_dev, out_batch_stride, row_offset, FR, NCH);
  } else {
    M2M_OPT_IN_LDS(logmel_kernel<false>, 160 * 1024);
    hipLaunchKernelGGL(logmel_kernel<false>, grid, dim3(FE_THREADS), smem, (hipStream_t)stream, wav_dev, T, F, fe->dev,
                       out_dev, out_batch_stride, row_offset, FR, NCH);
  }
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}

extern "C" int m2m_cond_rows_f32(const float* const* tables_dev_host, const int* table_rows_host, int n_tables,
                                 int n_dim, const int64_t* idx_dev, int B, float* out_dev,
                                 int64_t out_batch_stride, void* stream) {
  M2M_REQUIRE(tables_dev_host && table_rows_host && idx_dev && out_dev, "m2m_cond_rows_f32: null argument");
  M2M_REQUIRE(n_tables >= 1 && n_tables <= 8, "m2m_cond_rows_f32: n_tables %d out of range (1..8)", n_tables);
  M2M_REQUIRE(B >= 1 && n_dim >= 1, "m2m_cond_rows_f32: bad B / n_dim");
  CondArgs a{};
  for (int i = 0; i < n_tables; ++i) {
    M2M_REQUIRE(tables_dev_host[i] && table_rows_host[i] >= 1, "m2m_cond_rows_f32: table %d is null/empty", i);
    a.tables[i] = tables_dev_host[i];
    a.rows[i] = table_rows_host[i];
  }
  hipLaunchKernelGGL(cond_rows_kernel, dim3((unsigned)(B * n_tables)), dim3(128), 0, (hipStream_t)stream, a,
                     n_tables, n_dim, idx_dev, out_dev, out_batch_stride);
  M2M_CHECK_HIP(hipGetLastError());
  return M2M_OK;
}
