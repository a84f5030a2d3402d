// Shared declarations for the gfx950 kernels of the MMBERT hot path.
// Everything here is CDNA4-only: 64-wide wavefronts, fp32-input MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmvqa.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SERF = 3, ACT_SILU = 4, ACT_SIGMOID = 5 };
// A/B operand prologues (applied when a tile is written to LDS)
enum { PRO_NONE = 0, PRO_AFFINE_RELU = 1, PRO_DZ = 2, PRO_AFFINE = 3, PRO_AFFINE_SILU = 4, PRO_SILU_GATE = 5 };
// epilogue special modes
enum { EPI_PLAIN = 0, EPI_TAP_FWD = 1, EPI_TAP_BWD = 2 };
// contraction kinds of the implicit-GEMM family (MMVQA_KIND_* in the ABI)
enum { KIND_FWD = 0, KIND_DGRAD = 1, KIND_WGRAD = 2 };


// One descriptor drives the three implicit-GEMM kernels (forward / dgrad / wgrad); it is part of
// the C ABI (include/mmvqa.h).  Geometry ("g_") describes the gathered operand: rows of the row
// space are pixels (n, oy, ox) over [g_OH, g_OW]; the source tensor is NHWC over [g_SH, g_SW] with
// g_Cs channels per tap and ld floats per pixel.
//   a_pro/b_pro : PRO_* prologue applied when a tile is written to LDS (coefficients per channel)
//   A2          : second tensor of PRO_DZ (BatchNorm backward: dz = A*c0 + A2*c1 + c2)
//   b_tapstride : dgrad: floats between taps inside one output-channel row of W
//   g_nchw      : stem: source is NCHW [n][g_Cs][g_SH][g_SW], K index = (kh*KW+kw)*g_Cs + c
//   epilogue order: +bias -> Cpre store -> *act'(Pre) -> act -> dropout -> +R -> ReLU mask (Mk)
//                   -> store/atomicAdd -> column statistics (double, MMVQA_STAT_SLOTS replicas)
typedef mmvqa_gemm_desc GemmParams;
typedef mmvqa_attn_desc AttnParams;

// --------------------------------------------------------------------------- device math
// SERF(x) = x * erf(softplus(min(x,50)))  (models/serf.py:23-24).  softplus >= 0, so erf is only needed
// on [0, inf): odd Taylor polynomial below 0.5 (no cancellation for tiny arguments), Abramowitz-Stegun
// 7.1.28 above (|err| < 3e-7); exp/log on the hardware transcendental unit.  Measured against fp64:
// max abs error 9e-7 forward / 1e-6 derivative (torch's own fp32 result: 2e-7); ~4x fewer
// instructions than libm's erff/log1pf/expf, which matters in the tap epilogues (154 M evaluations/step).
__device__ __forceinline__ float erf_pos(float s) {
  float t = 1.0f + s * (0.0705230784f + s * (0.0422820123f + s * (0.0092705272f +
            s * (0.0001520143f + s * (0.0002765672f + s * 0.0000430638f)))));
  t = t * t; t = t * t; t = t * t; t = t * t;
  const float big = 1.0f - 1.0f / t;
  const float s2 = s * s;
  float pl = -7.5757575757e-4f;                       // -1/1320
  pl = pl * s2 + 4.6296296296e-3f;                    //  1/216
  pl = pl * s2 - 2.3809523810e-2f;                    // -1/42
  pl = pl * s2 + 0.1f;
  pl = pl * s2 - 0.33333333333f;
  pl = pl * s2 + 1.0f;
  const float small = 1.12837916709551257390f * s * pl;
  return s < 0.5f ? small : big;
}
__device__ __forceinline__ void serf_parts(float x, float& e, float& sp) {
  const float xc = x < 50.f ? x : 50.f;
  e = __expf(xc);
  sp = e < 0.01f ? e * (1.0f - e * (0.5f - e * 0.33333333333f)) : __logf(1.0f + e);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float dsilu_f(float x) { const float s = sigmoid_f(x); return s * (1.0f + x * (1.0f - s)); }

__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case ACT_SILU: return silu_f(x);
    case ACT_SIGMOID: return sigmoid_f(x);
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_GELU: return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    case ACT_SERF: {
      float e, sp;
      serf_parts(x, e, sp);
      return x * erf_pos(sp);
    }
    default: return x;
  }
}

__device__ __forceinline__ float act_bwd(int act, float x) {
  switch (act) {
    case ACT_SILU: return dsilu_f(x);
    case ACT_SIGMOID: { const float s = sigmoid_f(x); return s * (1.0f - s); }
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_GELU: {
      float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case ACT_SERF: {
      // d/dx [x erf(sp(min(x,50)))]; the clamp kills the inner derivative above 50
      float e, sp;
      serf_parts(x, e, sp);
      const float er = erf_pos(sp);
      if (x > 50.f) return er;
      const float sig = e / (1.0f + e);
      return er + x * 1.12837916709551257390f * __expf(-sp * sp) * sig;
    }
    default: return 1.f;
  }
}

// counter-based uniform in [0,1): (seed, index) -> float; identical in forward and backward
__device__ __forceinline__ float rng_uniform(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  x += seed * 0x85ebca6bu;
  x ^= x >> 13; x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// --------------------------------------------------------------------------- BatchNorm coefficients from raw sums
// (mmvqa_bn_fold in the ABI).  The arithmetic is that of bn_coef_fwd_kernel / bn_coef_bwd_kernel (elementwise.hip):
// sums in double over the replicas, torchvision BatchNorm2d semantics (biased variance for the batch, unbiased for the
// running estimate, momentum applied `reps` times at once).
typedef mmvqa_bn_fold BnFold;
typedef double f64x2 __attribute__((ext_vector_type(2)));

// sums of channel c over the replicas in use: every load is issued before the first add (one memory round trip)
__device__ __forceinline__ void bn_fold_sums(const double* __restrict__ stat, int slots, int C, int c, double& s0,
                                             double& s1) {
  f64x2 v[MMVQA_STAT_SLOTS];
#pragma unroll
  for (int k = 0; k < MMVQA_STAT_SLOTS; ++k) {
    v[k] = f64x2{0.0, 0.0};
    if (k < slots) v[k] = *reinterpret_cast<const f64x2*>(stat + ((size_t)k * C + c) * 2);
  }
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int k = 0; k < MMVQA_STAT_SLOTS; ++k) { s0 += v[k][0]; s1 += v[k][1]; }
}

// forward: (sum z, sum z^2) -> y = z*scale + shift; `pub`: this thread writes channel c's published values
__device__ __forceinline__ void bn_fold_fwd(const BnFold& f, int C, int c, bool pub, float& scale, float& shift) {
  double s, ss;
  bn_fold_sums(f.stat, f.slots, C, c, s, ss);
  const double mu = s / f.count;
  double var = ss / f.count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mu;
  const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  scale = f.gamma[c] * invstd;
  shift = f.beta[c] - mean * scale;
  if (pub) {
    f.out0[c] = scale; f.out1[c] = shift; f.out2[c] = mean; f.out3[c] = invstd;
    if (f.run_mean) {
      const double unb = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
      f.run_mean[c] = (float)(f.keep * (double)f.run_mean[c] + (1.0 - f.keep) * mu);
      f.run_var[c] = (float)(f.keep * (double)f.run_var[c] + (1.0 - f.keep) * unb);
    }
  }
}

// scale / shift of the NCH consecutive channels from c0 that a workgroup of an elementwise kernel works on, into LDS:
// from the coefficient arrays, or folded from the raw sums (thread t < NCH folds channel c0 + t once for the workgroup --
// every thread folding its own quad cost 16 replicas x 16 B x 4 channels of L2 reads per THREAD and lost to the separate
// coefficient launch).  `pub`: this workgroup also publishes.  All threads call it; ends with a barrier.
template <int NCH>
__device__ __forceinline__ void bn_coef_block_fwd(const float* __restrict__ s, const float* __restrict__ b, const BnFold& f,
                                                  int C, int c0, bool pub, float (*cf)[NCH]) {
  const int t = threadIdx.x;
  if (t < NCH) {
    float a = 0.f, d = 0.f;
    const int c = c0 + t;
    if (c < C) {
      if (f.stat) bn_fold_fwd(f, C, c, pub, a, d);
      else { a = s[c]; d = b[c]; }
    }
    cf[0][t] = a; cf[1][t] = d;
  }
  __syncthreads();
}

// backward: (sum g, sum g*xhat) -> dz = P*g + Q*z + R ; published: P, Q, R and dgamma += sum g*xhat, dbeta += sum g
__device__ __forceinline__ void bn_fold_bwd(const BnFold& f, int C, int c, bool pub, float& P, float& Q, float& R) {
  double sg, sgx;
  bn_fold_sums(f.stat, f.slots, C, c, sg, sgx);
  const double is = (double)f.invstd[c];
  const double p = (double)f.gamma[c] * is;
  const double c1 = sg / f.count, c2 = sgx / f.count;
  P = (float)p;
  Q = (float)(-p * c2 * is);
  R = (float)(p * (c2 * is * (double)f.mean[c] - c1));
  if (pub) {
    f.out0[c] = P; f.out1[c] = Q; f.out2[c] = R;
    f.dgamma[c] += (float)sgx;
    f.dbeta[c] += (float)sg;
  }
}

// the same for the backward coefficients (dz = P*g + Q*z + R) of a workgroup's NCH channels; ends with a barrier
template <int NCH>
__device__ __forceinline__ void bn_coef_block_bwd(const float* __restrict__ P, const float* __restrict__ Q,
                                                  const float* __restrict__ R, const BnFold& f, int C, int c0, bool pub,
                                                  float (*cf)[NCH]) {
  const int t = threadIdx.x;
  if (t < NCH) {
    float p = 0.f, q = 0.f, r = 0.f;
    const int c = c0 + t;
    if (c < C) {
      if (f.stat) bn_fold_bwd(f, C, c, pub, p, q, r);
      else { p = P[c]; q = Q[c]; r = R[c]; }
    }
    cf[0][t] = p; cf[1][t] = q; cf[2][t] = r;
  }
  __syncthreads();
}

// --------------------------------------------------------------------------- host side
int mmvqa_set_error(int code, const char* fmt, ...);
#define MMVQA_OK 0
#define MMVQA_ERR_ARG -1
#define MMVQA_ERR_HIP -2
#define MMVQA_ERR_STATE -3

#define HIP_CHECK_RET(expr)                                                               \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return mmvqa_set_error(MMVQA_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(_e), __FILE__, __LINE__);                  \
  } while (0)

#define KERNEL_CHECK_RET() HIP_CHECK_RET(hipGetLastError())
