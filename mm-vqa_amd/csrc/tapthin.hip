// Visual-token tap of a shallow feature map (models/image_encoding.py:53-62: conv1x1 C -> hidden, activation, global
// average pool) when the contraction is only C <= 64 channels deep: the stem taps of ResNet (C = 64, 112x112) and
// EfficientNetV2 (C = 24).
//
// As an implicit GEMM this is ONE K-tile per 64x64 output tile: 37 632 workgroups whose time is their setup and their
// epilogue (round-2 trace: 385 us isolated / 560-630 us inside the step for 9.9-19.7 GFLOP), and the activation of
// 154 M outputs is the real work.  Here a wave keeps its weight fragments (64 output features x C) in registers for
// its whole life, walks a contiguous range of 32-pixel row tiles with the activations loaded straight into MFMA
// operand registers (no LDS: any permutation of k is fine as long as both operands use it), applies the activation to
// the accumulators in place and carries the per-image column sums in registers; one atomic per (image, feature) and
// wave at the end.
#include "common.h"
#include "kernels.h"

namespace {

inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// KQ = C / 8 float4 per lane and operand; PRO: the feature map is relu(x * sc + sh) (BatchNorm + ReLU of the stem)
template <int KQ, bool PRO>
__global__ __launch_bounds__(256) void tap_thin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, const float* __restrict__ W,
                                                           float* __restrict__ out, int M, int N, int HW, int act,
                                                           int tiles_per_wave) {
  constexpr int C = KQ * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 64;
  // k order of both operands: lane half lh owns k = 8q + 4lh + i (q < KQ, i < 4)
  float bf[2][KQ * 4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = n0 + b * 32 + li;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      f32x4 w = {0, 0, 0, 0};
      if (n < N) w = *reinterpret_cast<const f32x4*>(W + (size_t)n * C + 8 * q + 4 * lh);
#pragma unroll
      for (int i = 0; i < 4; ++i) bf[b][4 * q + i] = w[i];
    }
  }
  float scv[PRO ? KQ * 4 : 1], shv[PRO ? KQ * 4 : 1];
  if constexpr (PRO) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + 8 * q + 4 * lh);
      const f32x4 h4 = *reinterpret_cast<const f32x4*>(sh + 8 * q + 4 * lh);
#pragma unroll
      for (int i = 0; i < 4; ++i) { scv[4 * q + i] = s4[i]; shv[4 * q + i] = h4[i]; }
    }
  }
  const int ntiles = M / 32;
  const int t0 = (blockIdx.y * 4 + wave) * tiles_per_wave;
  const int t1 = t0 + tiles_per_wave < ntiles ? t0 + tiles_per_wave : ntiles;
  if (t0 >= t1) return;
  const float inv_hw = 1.0f / (float)HW;
  float cs[2] = {0.f, 0.f};
  int cur = (t0 * 32) / HW;
  auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float tot = cs[b] + __shfl_xor(cs[b], 32, 64);
      const int n = n0 + b * 32 + li;
      if (lh == 0 && n < N) atomicAdd(&out[(size_t)cur * N + n], tot * inv_hw);
      cs[b] = 0.f;
    }
  };
  f32x4 a4[KQ], nx[KQ];
  {
    const float* src = x + (size_t)(t0 * 32 + li) * C + 4 * lh;
#pragma unroll
    for (int q = 0; q < KQ; ++q) a4[q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
  }
  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) {   // next tile's rows on their way while this one is multiplied and activated
      const float* src = x + (size_t)((t + 1) * 32 + li) * C + 4 * lh;
#pragma unroll
      for (int q = 0; q < KQ; ++q) nx[q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
    }
    const int img = (t * 32) / HW;
    if (img != cur) { flush(); cur = img; }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = a4[q][i];
        if constexpr (PRO) a = fmaxf(a * scv[4 * q + i] + shv[4 * q + i], 0.f);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[0][4 * q + i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[1][4 * q + i], acc1, 0, 0, 0);
      }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0 += act_fwd(act, acc0[e]); s1 += act_fwd(act, acc1[e]); }
    cs[0] += s0; cs[1] += s1;
#pragma unroll
    for (int q = 0; q < KQ; ++q) a4[q] = nx[q];
  }
  flush();
}

// Backward recompute of the same tap: du[pix][n] = dv[img][n] / HW * act'(u[pix][n]) with u recomputed as above (the
// 616 MB pre-activation map is never kept).  The accumulator tile goes through a wave-private LDS tile so that the
// stores are whole 256-byte row segments (the MFMA layout's own dword pattern reached 0.4 TB/s on this map).
template <int KQ, bool PRO>
__global__ __launch_bounds__(256) void tap_thin_bwd_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, const float* __restrict__ W,
                                                           const float* __restrict__ dv, float* __restrict__ du, int M,
                                                           int N, int HW, int act, int tiles_per_wave) {
  constexpr int C = KQ * 8, LDT = 68;
  __shared__ __attribute__((aligned(16))) float tile_s[4][32 * LDT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 64;
  float* tl = tile_s[wave];
  float bf[2][KQ * 4];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = n0 + b * 32 + li;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      f32x4 w = {0, 0, 0, 0};
      if (n < N) w = *reinterpret_cast<const f32x4*>(W + (size_t)n * C + 8 * q + 4 * lh);
#pragma unroll
      for (int i = 0; i < 4; ++i) bf[b][4 * q + i] = w[i];
    }
  }
  float scv[PRO ? KQ * 4 : 1], shv[PRO ? KQ * 4 : 1];
  if constexpr (PRO) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + 8 * q + 4 * lh);
      const f32x4 h4 = *reinterpret_cast<const f32x4*>(sh + 8 * q + 4 * lh);
#pragma unroll
      for (int i = 0; i < 4; ++i) { scv[4 * q + i] = s4[i]; shv[4 * q + i] = h4[i]; }
    }
  }
  const int ntiles = M / 32;
  const int t0 = (blockIdx.y * 4 + wave) * tiles_per_wave;
  const int t1 = t0 + tiles_per_wave < ntiles ? t0 + tiles_per_wave : ntiles;
  if (t0 >= t1) return;
  const float inv_hw = 1.0f / (float)HW;
  int cur = -1;
  float g0 = 0.f, g1 = 0.f;
  f32x4 a4[KQ], nx[KQ];
  {
    const float* src = x + (size_t)(t0 * 32 + li) * C + 4 * lh;
#pragma unroll
    for (int q = 0; q < KQ; ++q) a4[q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
  }
  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) {
      const float* src = x + (size_t)((t + 1) * 32 + li) * C + 4 * lh;
#pragma unroll
      for (int q = 0; q < KQ; ++q) nx[q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
    }
    const int img = (t * 32) / HW;
    if (img != cur) {
      cur = img;
      g0 = (n0 + li < N) ? dv[(size_t)img * N + n0 + li] * inv_hw : 0.f;
      g1 = (n0 + 32 + li < N) ? dv[(size_t)img * N + n0 + 32 + li] * inv_hw : 0.f;
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = a4[q][i];
        if constexpr (PRO) a = fmaxf(a * scv[4 * q + i] + shv[4 * q + i], 0.f);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[0][4 * q + i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[1][4 * q + i], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
      tl[row * LDT + li] = g0 * act_bwd(act, acc0[e]);
      tl[row * LDT + 32 + li] = g1 * act_bwd(act, acc1[e]);
    }
    // rows of 64 floats = 16 float4 chunks: 512 chunks over 64 lanes (same wave wrote them: LDS is in order per wave)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i, row = c >> 4, c4 = (c & 15) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tl + row * LDT + c4);
      const int col = n0 + c4;
      float* dst = du + (size_t)(t * 32 + row) * N + col;
      if (col + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (col + j < N) dst[j] = v[j];
      }
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) a4[q] = nx[q];
  }
}

}  // namespace

bool k_tap_thin_ok(long M, int N, int C, int HW) {
  return (C == 24 || C == 64) && M % 32 == 0 && HW % 32 == 0 && N % 4 == 0 && M / 32 >= 1024;
}

// out[img][n] += mean_hw act(sum_c x'[pix][c] W[n][c]),  x' = x or relu(x*sc+sh) (sc non-null); out is zeroed by the caller
int k_tap_thin_fwd(hipStream_t st, const float* x, const float* sc, const float* sh, const float* W, float* out, long M,
                   int N, int C, int HW, int act) {
  if (!k_tap_thin_ok(M, N, C, HW)) return mmvqa_set_error(MMVQA_ERR_ARG, "tap_thin_fwd: M=%ld N=%d C=%d HW=%d", M, N, C, HW);
  const int ntiles = (int)(M / 32);
  const int gx = cdiv_i(N, 64);
  // about 4 waves per SIMD over the chip, every wave a contiguous run of row tiles (few image changes per wave)
  int gy = cdiv_i(4096, gx);
  if (gy * 4 > ntiles) gy = cdiv_i(ntiles, 4);
  const int tpw = cdiv_i(ntiles, (long)gy * 4);
  gy = cdiv_i(ntiles, (long)tpw * 4);
  const dim3 grid(gx, gy);
#define GO(KQ_)                                                                                                          \
  do {                                                                                                                   \
    if (sc)                                                                                                              \
      hipLaunchKernelGGL((tap_thin_fwd_kernel<KQ_, true>), grid, dim3(256), 0, st, x, sc, sh, W, out, (int)M, N, HW, act, tpw); \
    else                                                                                                                 \
      hipLaunchKernelGGL((tap_thin_fwd_kernel<KQ_, false>), grid, dim3(256), 0, st, x, sc, sh, W, out, (int)M, N, HW, act, tpw); \
  } while (0)
  if (C == 24) GO(3); else GO(8);
#undef GO
  HIP_CHECK_RET(hipGetLastError());
  return MMVQA_OK;
}

// du[pix][n] = dv[img][n] / HW * act'(sum_c x'[pix][c] W[n][c])   (N a multiple of 4)
int k_tap_thin_bwd(hipStream_t st, const float* x, const float* sc, const float* sh, const float* W, const float* dv,
                   float* du, long M, int N, int C, int HW, int act) {
  if (!k_tap_thin_ok(M, N, C, HW)) return mmvqa_set_error(MMVQA_ERR_ARG, "tap_thin_bwd: M=%ld N=%d C=%d HW=%d", M, N, C, HW);
  const int ntiles = (int)(M / 32);
  const int gx = cdiv_i(N, 64);
  int gy = cdiv_i(4096, gx);
  if (gy * 4 > ntiles) gy = cdiv_i(ntiles, 4);
  const int tpw = cdiv_i(ntiles, (long)gy * 4);
  gy = cdiv_i(ntiles, (long)tpw * 4);
  const dim3 grid(gx, gy);
#define GO(KQ_)                                                                                                          \
  do {                                                                                                                   \
    if (sc)                                                                                                              \
      hipLaunchKernelGGL((tap_thin_bwd_kernel<KQ_, true>), grid, dim3(256), 0, st, x, sc, sh, W, dv, du, (int)M, N, HW, act, tpw); \
    else                                                                                                                 \
      hipLaunchKernelGGL((tap_thin_bwd_kernel<KQ_, false>), grid, dim3(256), 0, st, x, sc, sh, W, dv, du, (int)M, N, HW, act, tpw); \
  } while (0)
  if (C == 24) GO(3); else GO(8);
#undef GO
  HIP_CHECK_RET(hipGetLastError());
  return MMVQA_OK;
}
