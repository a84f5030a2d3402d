// HBM-bound primitives of the MMBERT path (gfx950): BatchNorm coefficient kernels, block-end
// add+ReLU, max-pool, LayerNorm, embedding gather, mean-pool, vocab log-softmax/NLL, ASL,
// SupCon, L2-normalise, Adam.  All use 16-byte accesses along the contiguous axis and
// 64-lane wave reductions; per-channel sums are kept in double and spread over
// MMVQA_STAT_SLOTS replicas to avoid same-address atomic serialisation.
#include <cstring>

#include "common.h"

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// =========================================================================== BatchNorm coefficients
// torchvision BatchNorm2d as the reference drives it (models/image_encoding.py:72-86):
// train mode -> batch statistics (biased var) + running-stat update repeated `reps` times
// with the same statistics; eval mode -> running statistics.
// One 16-lane group per channel: lane k reads replica k of the (sum, sum of squares) pair with one 16-byte load
// and the group folds them with four shuffles (a thread per channel walking the 16 replicas serially made the
// kernel 4.8 us of pure load latency, 310 times per step on the critical path between two convolutions).
static_assert(MMVQA_STAT_SLOTS == 16, "one lane per statistics replica");
__device__ __forceinline__ void stat_fold16(const double* __restrict__ stat, int C, int c, bool live, double& s0,
                                            double& s1) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const int sl = threadIdx.x & 15;
  f64x2 v = {0.0, 0.0};
  if (live) v = *reinterpret_cast<const f64x2*>(stat + ((size_t)sl * C + c) * 2);
  s0 = v[0]; s1 = v[1];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
}
__global__ __launch_bounds__(256) void bn_coef_fwd_kernel(const double* __restrict__ stat, int C, double count, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ run_mean, float* __restrict__ run_var,
                                   long long* __restrict__ nbt, double keep, int reps, int training,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (blockIdx.x == 0 && threadIdx.x == 0 && training && nbt) *nbt += reps;
  double s = 0.0, ss = 0.0;
  if (training) stat_fold16(stat, C, c, c < C, s, ss);
  if (c >= C || (threadIdx.x & 15) != 0) return;
  float mean, invstd;
  if (training) {
    double mu = s / count;
    double var = ss / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
      double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
      run_mean[c] = (float)(keep * (double)run_mean[c] + (1.0 - keep) * mu);
      run_var[c] = (float)(keep * (double)run_var[c] + (1.0 - keep) * unb);
    }
  } else {
    mean = run_mean[c];
    invstd = 1.0f / sqrtf(run_var[c] + eps);
  }
  float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  mean_out[c] = mean;
  invstd_out[c] = invstd;
}

// dz = P*g + Q*z + R  with g = dL/d(bn output); also dgamma += sum g*xhat, dbeta += sum g
__global__ __launch_bounds__(256) void bn_coef_bwd_kernel(const double* __restrict__ stat, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, int training,
                                   float* __restrict__ P, float* __restrict__ Q, float* __restrict__ R,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  double sg, sgx;
  stat_fold16(stat, C, c, c < C, sg, sgx);
  if (c >= C || (threadIdx.x & 15) != 0) return;
  double p = (double)gamma[c] * (double)invstd[c];
  if (training) {
    double c1 = sg / count, c2 = sgx / count;
    P[c] = (float)p;
    Q[c] = (float)(-p * c2 * (double)invstd[c]);
    R[c] = (float)(p * (c2 * (double)invstd[c] * (double)mean[c] - c1));
  } else {  // eval-mode BN is a fixed affine map
    P[c] = (float)p; Q[c] = 0.f; R[c] = 0.f;
  }
  dgamma[c] += (float)sgx;
  dbeta[c] += (float)sg;
}

// =========================================================================== block end: relu(bn3(z3) + identity)
__global__ void bn_add_relu_kernel(const float* __restrict__ z, const float* __restrict__ s,
                                   const float* __restrict__ b, const float* __restrict__ idn,
                                   const float* __restrict__ ids, const float* __restrict__ idb,
                                   float* __restrict__ out, long n4, int C4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    int c = (int)(i % C4) * 4;
    f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
    f32x4 sv = *reinterpret_cast<const f32x4*>(s + c), bv = *reinterpret_cast<const f32x4*>(b + c);
    f32x4 iv = reinterpret_cast<const f32x4*>(idn)[i];
    if (ids) {
      f32x4 s2 = *reinterpret_cast<const f32x4*>(ids + c), b2 = *reinterpret_cast<const f32x4*>(idb + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) iv[j] = iv[j] * s2[j] + b2[j];
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { float t = zv[j] * sv[j] + bv[j] + iv[j]; o[j] = t > 0.f ? t : 0.f; }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// =========================================================================== max-pool 3x3 s2 p1 on relu(bn(z))
__global__ void maxpool_fwd_kernel(const float* __restrict__ z, const float* __restrict__ s,
                                   const float* __restrict__ b, float* __restrict__ out,
                                   unsigned char* __restrict__ idx, int N, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  long total = (long)N * OH * OW * C4;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    int c = (int)(i % C4) * 4;
    long pix = i / C4;
    int ox = (int)(pix % OW); long t = pix / OW;
    int oy = (int)(t % OH); int n = (int)(t / OH);
    f32x4 sv = *reinterpret_cast<const f32x4*>(s + c), bv = *reinterpret_cast<const f32x4*>(b + c);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int y = oy * 2 - 1 + kh;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int x = ox * 2 - 1 + kw;
        if (x < 0 || x >= W) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(z + ((size_t)(n * H + y) * W + x) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = v[j] * sv[j] + bv[j];
          a = a > 0.f ? a : 0.f;
          if (a > best[j]) { best[j] = a; bi[j] = kh * 3 + kw; }
        }
      }
    }
    reinterpret_cast<f32x4*>(out)[i] = best;
    uchar4 u; u.x = bi[0]; u.y = bi[1]; u.z = bi[2]; u.w = bi[3];
    reinterpret_cast<uchar4*>(idx)[i] = u;
  }
}

// g0 = (maxpool_bwd(gp) + extra) * [relu(bn(z)) > 0]; accumulates BN-backward sums of bn(z).
// blockDim = 256, C4 must divide 256 so that a thread keeps its channel quad across the grid-stride loop.
__global__ void maxpool_bwd_kernel(const float* __restrict__ gp, const unsigned char* __restrict__ idx,
                                   const float* __restrict__ extra, const float* __restrict__ z,
                                   const float* __restrict__ s, const float* __restrict__ b,
                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                   float* __restrict__ g0, double* __restrict__ stat,
                                   int N, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  long total = (long)N * H * W * C4;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  const int c = (int)(i % C4) * 4;
  f32x4 sv = *reinterpret_cast<const f32x4*>(s + c), bv = *reinterpret_cast<const f32x4*>(b + c);
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
  double sg[4] = {0, 0, 0, 0}, sgx[4] = {0, 0, 0, 0};
  for (; i < total; i += stride) {
    long pix = i / C4;
    int x = (int)(pix % W); long t = pix / W;
    int y = (int)(t % H); int n = (int)(t / H);
    f32x4 g = extra ? reinterpret_cast<const f32x4*>(extra)[i] : f32x4{0, 0, 0, 0};
    int oy0 = y >> 1, ox0 = x >> 1;  // windows oy with oy*2-1 <= y <= oy*2+1
    for (int oy = oy0; oy <= oy0 + 1; ++oy) {
      int kh = y - (oy * 2 - 1);
      if (kh < 0 || kh > 2 || oy >= OH) continue;
      for (int ox = ox0; ox <= ox0 + 1; ++ox) {
        int kw = x - (ox * 2 - 1);
        if (kw < 0 || kw > 2 || ox >= OW) continue;
        size_t o = ((size_t)(n * OH + oy) * OW + ox) * C + c;
        uchar4 u = *reinterpret_cast<const uchar4*>(idx + o);
        f32x4 gv = *reinterpret_cast<const f32x4*>(gp + o);
        int code = kh * 3 + kw;
        if (u.x == code) g[0] += gv[0];
        if (u.y == code) g[1] += gv[1];
        if (u.z == code) g[2] += gv[2];
        if (u.w == code) g[3] += gv[3];
      }
    }
    f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = zv[j] * sv[j] + bv[j];
      if (!(a > 0.f)) g[j] = 0.f;
      sg[j] += (double)g[j];
      sgx[j] += (double)(g[j] * ((zv[j] - mu[j]) * is[j]));
    }
    reinterpret_cast<f32x4*>(g0)[i] = g;
  }
  // block reduction over the threads that share a channel quad (tid % C4)
  __shared__ double red[256 * 8];
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = sg[j]; red[tid * 8 + 4 + j] = sgx[j]; }
  __syncthreads();
  if (tid < C4) {
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = tid; t < 256; t += C4)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += red[t * 8 + j];
    int slot = blockIdx.x & (MMVQA_STAT_SLOTS - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double* d = stat + ((size_t)slot * C + tid * 4 + j) * 2;
      atomicAdd(d, a[j]);
      atomicAdd(d + 1, a[4 + j]);
    }
  }
}

// =========================================================================== LayerNorm (one wave per row)
#define LN_MAXV 4  // H <= 1024
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ y, float* __restrict__ sum_out,
                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                     int rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int H4 = H / 4;
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    v[j] = f32x4{0, 0, 0, 0};
    if (q < H4) {
      v[j] = reinterpret_cast<const f32x4*>(x + (size_t)row * H)[q];
      if (res) {
        f32x4 r = reinterpret_cast<const f32x4*>(res + (size_t)row * H)[q];
        v[j] += r;
        if (sum_out) reinterpret_cast<f32x4*>(sum_out + (size_t)row * H)[q] = v[j];
      }
      s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
  }
  float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    if (q < H4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[j][e] - mean; ss += d * d; }
    }
  }
  float var = wave_sum(ss) / (float)H;
  float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    if (q < H4) {
      f32x4 g = reinterpret_cast<const f32x4*>(gamma)[q], b = reinterpret_cast<const f32x4*>(beta)[q], o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + b[e];
      reinterpret_cast<f32x4*>(y + (size_t)row * H)[q] = o;
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dx = rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)) (+ dres); dgamma/dbeta accumulated with
// float atomics after a per-workgroup reduction through LDS.
__global__ void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                     float* __restrict__ dx, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int rows, int H, int rows_per_wave) {
  extern __shared__ float lds[];  // [2][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int H4 = H / 4;
  for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  f32x4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) { ag[j] = f32x4{0, 0, 0, 0}; ab[j] = f32x4{0, 0, 0, 0}; }
  const int row0 = (blockIdx.x * nw + wave) * rows_per_wave;
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int row = row0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[LN_MAXV], dg[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      int q = lane + 64 * j;
      xh[j] = f32x4{0, 0, 0, 0}; dg[j] = f32x4{0, 0, 0, 0};
      if (q < H4) {
        f32x4 xv = reinterpret_cast<const f32x4*>(x + (size_t)row * H)[q];
        f32x4 d = reinterpret_cast<const f32x4*>(dy + (size_t)row * H)[q];
        f32x4 g = reinterpret_cast<const f32x4*>(gamma)[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[j][e] = (xv[e] - mu) * rs;
          dg[j][e] = d[e] * g[e];
          s1 += dg[j][e];
          s2 += dg[j][e] * xh[j][e];
          ag[j][e] += d[e] * xh[j][e];
          ab[j][e] += d[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      int q = lane + 64 * j;
      if (q < H4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (dg[j][e] - s1 - xh[j][e] * s2);
        if (dres) o += reinterpret_cast<const f32x4*>(dres + (size_t)row * H)[q];
        reinterpret_cast<f32x4*>(dx + (size_t)row * H)[q] = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    if (q < H4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(&lds[q * 4 + e], ag[j][e]);
        atomicAdd(&lds[H + q * 4 + e], ab[j][e]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    atomicAdd(&dgamma[i], lds[i]);
    atomicAdd(&dbeta[i], lds[H + i]);
  }
}

// =========================================================================== embeddings + visual-token overwrite
// HF BertEmbeddings (call site models/mmbert.py:63) + prepare_input overwrite (mmbert.py:64-66):
// row (b,t): t < num_vis -> vis[t][b][:]; else dropout(LN(word[id]+type[seg]+pos[t])).
__global__ void embed_fwd_kernel(const long long* __restrict__ ids, const long long* __restrict__ seg,
                                 const float* __restrict__ word, const float* __restrict__ pos,
                                 const float* __restrict__ type, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ vis,
                                 float* __restrict__ out, float* __restrict__ xhat_out,
                                 float* __restrict__ rstd_out, int B, int T, int H, int num_vis,
                                 float eps, float drop_p, uint32_t seed) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int b = row / T, t = row - b * T;
  const int H4 = H / 4;
  if (t < num_vis) {
    for (int q = lane; q < H4; q += 64)
      reinterpret_cast<f32x4*>(out + (size_t)row * H)[q] =
          reinterpret_cast<const f32x4*>(vis + ((size_t)t * B + b) * H)[q];
    return;
  }
  const long long id = ids[row], sg = seg[row];
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    v[j] = f32x4{0, 0, 0, 0};
    if (q < H4) {
      f32x4 w = reinterpret_cast<const f32x4*>(word + (size_t)id * H)[q];
      f32x4 ty = reinterpret_cast<const f32x4*>(type + (size_t)sg * H)[q];
      f32x4 p = reinterpret_cast<const f32x4*>(pos + (size_t)t * H)[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = (w[e] + ty[e]) + p[e];
      s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
  }
  float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    if (q < H4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[j][e] - mean; ss += d * d; }
    }
  }
  float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)H + eps);
  const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    if (q < H4) {
      f32x4 g = reinterpret_cast<const f32x4*>(gamma)[q], be = reinterpret_cast<const f32x4*>(beta)[q], o, xh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[e] = (v[j][e] - mean) * rstd;
        o[e] = xh[e] * g[e] + be[e];
        if (drop_p > 0.f) {
          float u = rng_uniform(seed, (uint32_t)row * (uint32_t)H + q * 4 + e);
          o[e] = (u >= drop_p) ? o[e] * ks : 0.f;
        }
      }
      reinterpret_cast<f32x4*>(out + (size_t)row * H)[q] = o;
      if (xhat_out) reinterpret_cast<f32x4*>(xhat_out + (size_t)row * H)[q] = xh;
    }
  }
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
}

// word_embeddings has padding_idx = 0 (HF BertConfig.pad_token_id): row 0 never receives gradient.
__global__ void embed_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ ids,
                                 const long long* __restrict__ seg, const float* __restrict__ xhat,
                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                 float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype,
                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dvis,
                                 int B, int T, int H, int num_vis, float drop_p, uint32_t seed, int pad_idx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int b = row / T, t = row - b * T;
  const int H4 = H / 4;
  if (t < num_vis) {
    for (int q = lane; q < H4; q += 64)
      reinterpret_cast<f32x4*>(dvis + ((size_t)t * B + b) * H)[q] =
          reinterpret_cast<const f32x4*>(dout + (size_t)row * H)[q];
    return;
  }
  const long long id = ids[row], sg = seg[row];
  const float rs = rstd[row];
  const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  f32x4 xh[LN_MAXV], dg[LN_MAXV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    xh[j] = f32x4{0, 0, 0, 0}; dg[j] = f32x4{0, 0, 0, 0};
    if (q < H4) {
      f32x4 d = reinterpret_cast<const f32x4*>(dout + (size_t)row * H)[q];
      xh[j] = reinterpret_cast<const f32x4*>(xhat + (size_t)row * H)[q];
      f32x4 g = reinterpret_cast<const f32x4*>(gamma)[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (drop_p > 0.f) {
          float u = rng_uniform(seed, (uint32_t)row * (uint32_t)H + q * 4 + e);
          d[e] = (u >= drop_p) ? d[e] * ks : 0.f;
        }
        atomicAdd(&dgamma[q * 4 + e], d[e] * xh[j][e]);
        atomicAdd(&dbeta[q * 4 + e], d[e]);
        dg[j][e] = d[e] * g[e];
        s1 += dg[j][e];
        s2 += dg[j][e] * xh[j][e];
      }
    }
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    int q = lane + 64 * j;
    if (q < H4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float de = rs * (dg[j][e] - s1 - xh[j][e] * s2);
        if (id != pad_idx) atomicAdd(&dword[(size_t)id * H + q * 4 + e], de);
        atomicAdd(&dtype[(size_t)sg * H + q * 4 + e], de);
        atomicAdd(&dpos[(size_t)t * H + q * 4 + e], de);
      }
    }
  }
}

// =========================================================================== mean pooling (models/mmbert.py:169-172)
__global__ void meanpool_fwd_kernel(const float* __restrict__ h, const long long* __restrict__ mask,
                                    float* __restrict__ out, int B, int T, int H) {
  int b = blockIdx.x;
  float cnt = 0.f;
  for (int t = 0; t < T; ++t) cnt += (float)mask[b * T + t];
  float denom = fmaxf(cnt, 1e-9f);
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += h[((size_t)b * T + t) * H + c] * (float)mask[b * T + t];
    out[(size_t)b * H + c] = s / denom;
  }
}
// dh (+)= dout * mask / denom
__global__ void meanpool_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ mask,
                                    float* __restrict__ dh, int B, int T, int H, int accumulate) {
  int b = blockIdx.x;
  float cnt = 0.f;
  for (int t = 0; t < T; ++t) cnt += (float)mask[b * T + t];
  float denom = fmaxf(cnt, 1e-9f);
  for (int i = threadIdx.x; i < T * H; i += blockDim.x) {
    int t = i / H, c = i - t * H;
    float v = dout[(size_t)b * H + c] * (float)mask[b * T + t] / denom;
    size_t o = ((size_t)b * T + t) * H + c;
    dh[o] = accumulate ? dh[o] + v : v;
  }
}

// =========================================================================== vocab log-softmax + NLL + argmax
// pretrain/roco_utils.py:235-236,257-265.  One workgroup per row.  Writes row loss
// (lse - logit[target]), first-index argmax, and (optionally) dlogits = (softmax - onehot) * gscale.
__global__ void lsm_nll_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target,
                               float* __restrict__ row_loss, float* __restrict__ row_lse,
                               long long* __restrict__ pred,
                               float* __restrict__ dlogits, int dld, const float* __restrict__ gscale_ptr,
                               float gscale_mul, int V) {
  __shared__ float red_m[4], red_s[4];
  __shared__ int red_i[4];
  const int row = blockIdx.x;
  const float* x = logits + (size_t)row * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY; int mi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    float v = x[i];
    if (v > m) { m = v; mi = i; }
  }
  // wave argmax (first index on ties)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(m, o, 64); int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  if (lane == 0) { red_m[wave] = m; red_i[wave] = mi; }
  __syncthreads();
  m = red_m[0]; mi = red_i[0];
  for (int w = 1; w < 4; ++w) if (red_m[w] > m || (red_m[w] == m && red_i[w] < mi)) { m = red_m[w]; mi = red_i[w]; }
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) s += expf(x[i] - m);
  s = wave_sum(s);
  if (lane == 0) red_s[wave] = s;
  __syncthreads();
  s = red_s[0] + red_s[1] + red_s[2] + red_s[3];
  const float lse = m + logf(s);
  const long long tg = target[row];
  if (threadIdx.x == 0) {
    row_loss[row] = lse - x[tg];
    if (row_lse) row_lse[row] = lse;
    pred[row] = mi;
  }
  if (dlogits) {
    const float gs = (gscale_ptr ? *gscale_ptr : 1.0f) * gscale_mul;
    float* d = dlogits + (size_t)row * dld;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
      float pr = expf(x[i] - lse);
      d[i] = (pr - (i == tg ? 1.f : 0.f)) * gs;
    }
  }
}

// ---- HBM-speed form of the same reduction (rows 16-byte aligned, ld >= round_up(V, 4)).
// One 1024-thread workgroup per row (2 rows per CU in flight = full wave occupancy), ONE pass over the row with
// float4 loads, LSM_U of them in flight per thread; a thread keeps a running (max, first index, sum) and rescales
// its sum once per chunk, not per element.  Algorithmic bytes: rows * V * 4 read once (62.5 MB at B*T=512, V=30522).
#define LSM_U 4
__device__ __forceinline__ void lsm_combine(float& m, int& mi, float& s, float om, int oi, float os) {
  const float M = fmaxf(m, om);
  const float sa = (m == -INFINITY) ? 0.f : s * expf(m - M);
  const float sb = (om == -INFINITY) ? 0.f : os * expf(om - M);
  if (om > m || (om == m && oi < mi)) mi = oi;
  m = M; s = sa + sb;
}
__global__ void __launch_bounds__(1024) lsm_stats_kernel(const float* __restrict__ logits, int ld,
                                                         const long long* __restrict__ target,
                                                         float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                         long long* __restrict__ pred, int V) {
  __shared__ float red_m[16], red_s[16];
  __shared__ int red_i[16];
  const int row = blockIdx.x;
  const float* x = logits + (size_t)row * ld;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const int nvec = (V + 3) >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY, s = 0.f; int mi = 0x7fffffff;
  for (int j0 = threadIdx.x; j0 < nvec; j0 += 1024 * LSM_U) {
    f32x4 v[LSM_U];
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) {
      const int j = j0 + u * 1024;
      v[u] = (j < nvec) ? x4[j] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    float cm = m;
#pragma unroll
    for (int u = 0; u < LSM_U; ++u) {
      const int base = (j0 + u * 1024) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (base + e >= V) v[u][e] = -INFINITY;            // the pad columns of the last vector
        if (v[u][e] > cm) { cm = v[u][e]; mi = base + e; }   // indices increase within a thread: first index wins
      }
    }
    if (cm > -INFINITY) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < LSM_U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += __expf(v[u][e] - cm);   // hardware exp2 path (rel. error ~1e-6, averaged over V terms); exp(-inf) = 0 for masked slots
      s = (m == -INFINITY ? 0.f : s * expf(m - cm)) + acc;
      m = cm;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    lsm_combine(m, mi, s, om, oi, os);
  }
  if (lane == 0) { red_m[wave] = m; red_i[wave] = mi; red_s[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) lsm_combine(m, mi, s, red_m[w], red_i[w], red_s[w]);
    const float lse = m + logf(s);
    row_lse[row] = lse;
    row_loss[row] = lse - x[target[row]];
    pred[row] = mi;
  }
}
// dlogits = (softmax - onehot) * gs from the saved row log-sums: one streaming read + one streaming write
// (2 * rows * V * 4 bytes); blockIdx.y walks the row in pieces of 256 * LSM_U float4; pad columns are zeroed.
__global__ void __launch_bounds__(256) lsm_grad_kernel(const float* __restrict__ logits, int ld,
                                                       const long long* __restrict__ target,
                                                       const float* __restrict__ row_lse, float* __restrict__ dlogits,
                                                       int dld, const float* __restrict__ gscale_ptr, float gscale_mul,
                                                       int V) {
  const int row = blockIdx.x;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(logits + (size_t)row * ld);
  f32x4* d4 = reinterpret_cast<f32x4*>(dlogits + (size_t)row * dld);
  const int nvec = (V + 3) >> 2;
  const float lse = row_lse[row];
  const int tg = (int)target[row];
  const float gs = (gscale_ptr ? *gscale_ptr : 1.0f) * gscale_mul;
  const int j0 = blockIdx.y * 256 * LSM_U + threadIdx.x;
  f32x4 v[LSM_U];
#pragma unroll
  for (int u = 0; u < LSM_U; ++u) {
    const int j = j0 + u * 256;
    if (j < nvec) v[u] = x4[j];
  }
#pragma unroll
  for (int u = 0; u < LSM_U; ++u) {
    const int j = j0 + u * 256;
    if (j >= nvec) continue;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = j * 4 + e;
      o[e] = i < V ? (__expf(v[u][e] - lse) - (i == tg ? 1.f : 0.f)) * gs : 0.f;
    }
    d4[j] = o;
  }
}

// loss = mean(row_loss); n_masked = #(target>0); n_correct = #(target>0 && pred==target)
__global__ void mlm_reduce_kernel(const float* __restrict__ row_loss, const long long* __restrict__ pred,
                                  const long long* __restrict__ target, int rows, float* __restrict__ out3) {
  float s = 0.f, nm = 0.f, nc = 0.f;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) {
    s += row_loss[i];
    if (target[i] > 0) { nm += 1.f; if (pred[i] == target[i]) nc += 1.f; }
  }
  __shared__ float red[3][4];
  s = wave_sum(s); nm = wave_sum(nm); nc = wave_sum(nc);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = nm; red[2][wave] = nc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out3[0] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)rows;
    out3[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out3[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  }
}

// =========================================================================== ASL single-label (models/asl_singlelabel.py:23-53)
// One workgroup per sample; loss_b = -sum_j t~_j w_j lp_j with w_target = (1-p_t)^gp... (gamma_pos on the
// target, gamma_neg elsewhere); writes per-sample loss and dlogits * gscale / B.
__global__ void asl_kernel(const float* __restrict__ logits, int ld, const long long* __restrict__ target,
                           float* __restrict__ row_loss, float* __restrict__ dlogits, int dld, int C,
                           float gpos, float gneg, float eps, float gscale) {
  __shared__ float red[4];
  __shared__ float bc;
  const int row = blockIdx.x;
  const float* x = logits + (size_t)row * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tg = (int)target[row];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < C; i += blockDim.x) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += blockDim.x) s += expf(x[i] - m);
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float lse = m + logf(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  // loss and G_j = dL/d lp_j ; dlogit_k = G_k - p_k * sum_j G_j
  float loss = 0.f, gsum = 0.f;
  const float tsm_o = eps / (float)C;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    float lp = x[i] - lse, p = expf(lp);
    bool is_t = (i == tg);
    float ts = (is_t ? (1.f - eps) : 0.f) + tsm_o;
    float base = is_t ? (1.f - p) : p;      // 1 - xs_pos*t - xs_neg*anti
    float gam = is_t ? gpos : gneg;
    float w = (gam == 0.f) ? 1.f : powf(base, gam);
    loss += -ts * w * lp;
    // dw/dlp = gam * base^(gam-1) * dbase/dlp ; dbase/dlp = (is_t ? -p : p)
    float dw = (gam == 0.f) ? 0.f : gam * powf(base, gam - 1.f) * (is_t ? -p : p);
    float G = -ts * (w + lp * dw);
    gsum += G;
    if (dlogits) dlogits[(size_t)row * dld + i] = G;  // finished below
  }
  loss = wave_sum(loss); gsum = wave_sum(gsum);
  if (lane == 0) red[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) row_loss[row] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (lane == 0) red[wave] = gsum;
  __syncthreads();
  if (threadIdx.x == 0) bc = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (dlogits) {
    const float gs = bc;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
      float p = expf(x[i] - lse);
      size_t o = (size_t)row * dld + i;
      dlogits[o] = (dlogits[o] - p * gs) * gscale;
    }
  }
}

// =========================================================================== L2 normalise rows (F.normalize, eps 1e-12)
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm,
                                  int rows, int D) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) { float v = x[(size_t)row * D + i]; s += v * v; }
  float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int i = lane; i < D; i += 64) y[(size_t)row * D + i] = x[(size_t)row * D + i] / n;
  if (lane == 0) nrm[row] = n;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                  const float* __restrict__ nrm, float* __restrict__ dx, int rows, int D) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += dy[(size_t)row * D + i] * y[(size_t)row * D + i];
  s = wave_sum(s);
  float n = nrm[row];
  for (int i = lane; i < D; i += 64)
    dx[(size_t)row * D + i] = (dy[(size_t)row * D + i] - y[(size_t)row * D + i] * s) / n;
}

// =========================================================================== SupCon / SimCLR (models/SupConLoss/loss.py:21-98)
// features f [R=2N, D] ordered view-major (cat(unbind(features,1)), loss.py:57): positive of r is (r+N) mod 2N.
// Tiled over row blocks so that the all-gathered view set of a data-parallel job (R = 2N*world, SURVEY 8(e)
// collective 2) runs at any size: a workgroup owns SC_ROWS anchor rows and walks the contrast rows in tiles of
// 64 staged in LDS (lane = contrast row, so a score never leaves its register); nothing of size R*R exists.
//   pass 1 (supcon_rows_kernel): z = f f^T / T (loss.py:70-72), row max over ALL columns incl. the diagonal
//     (:74-75), log sum of exp over the columns != row (:88-89) -> lse[r], row_loss[r] = -(z[r,pos] - lse[r]) (:92-95)
//   pass 2 (supcon_grad_kernel): dL/dz[a][b] = c (p_a[b] - [b = pos a]) with p_a = softmax over b != a; z and the
//     positive map are symmetric, so df[a] = c/T * sum_b (p_a[b] + p_b[a] - 2 [b = pos a]) f[b]: scores recomputed.
//   pass 3 (supcon_reduce_kernel): loss = (T/T_base) * mean_r row_loss[r] (:95-96), fixed summation order.
#define SC_ROWS 16
#define SC_COLS 64
#define SC_MAXD 256
__device__ __forceinline__ void sc_stage(float* dst, int dst_ld, const float* __restrict__ f, int row0, int nrows,
                                         int R, int D) {
  for (int i = threadIdx.x; i < nrows * D; i += blockDim.x) {
    const int r = i / D, k = i - r * D;
    dst[r * dst_ld + k] = (row0 + r < R) ? f[(size_t)(row0 + r) * D + k] : 0.f;
  }
}
__global__ void __launch_bounds__(256) supcon_rows_kernel(const float* __restrict__ f, float* __restrict__ lse,
                                                          float* __restrict__ row_loss, int N, int D, float temp) {
  extern __shared__ float sm[];
  const int R = 2 * N, ldb = D + 1;
  float* fa = sm;                      // [SC_ROWS][D]   anchors of this workgroup
  float* fb = sm + SC_ROWS * D;        // [SC_COLS][D+1] contrast tile (odd stride: lane = row reads conflict-free)
  const int a0 = blockIdx.x * SC_ROWS, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sc_stage(fa, D, f, a0, SC_ROWS, R, D);
  float m[4], s[4], zp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; s[r] = 0.f; zp[r] = 0.f; }
  for (int b0 = 0; b0 < R; b0 += SC_COLS) {
    __syncthreads();
    sc_stage(fb, ldb, f, b0, SC_COLS, R, D);
    __syncthreads();
    const int b = b0 + lane;
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < D; ++k) {
      const float vb = fb[lane * ldb + k];
#pragma unroll
      for (int r = 0; r < 4; ++r) dot[r] += fa[(wave * 4 + r) * D + k] * vb;
    }
    if (b < R) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + wave * 4 + r;
        const float z = dot[r] / temp;
        const float mn = fmaxf(m[r], z);                 // the diagonal takes part in the max, not in the sum
        s[r] = s[r] * __expf(m[r] - mn) + (b != a ? __expf(z - mn) : 0.f);
        m[r] = mn;
        if (b == (a + N) % R) zp[r] = z;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int a = a0 + wave * 4 + r;
    const float M = wave_max(m[r]);
    const float S = wave_sum(m[r] == -INFINITY ? 0.f : s[r] * __expf(m[r] - M));
    const float Z = wave_sum(zp[r]);
    if (lane == 0 && a < R) {
      const float l = M + logf(S);
      lse[a] = l;
      row_loss[a] = -(Z - l);
    }
  }
}
__global__ void __launch_bounds__(256) supcon_grad_kernel(const float* __restrict__ f, const float* __restrict__ lse,
                                                          float* __restrict__ df, int N, int D, float temp,
                                                          float cscale) {
  extern __shared__ float sm[];
  const int R = 2 * N, ldb = D + 1;
  float* fa = sm;                               // [SC_ROWS][D]
  float* fb = fa + SC_ROWS * D;                 // [SC_COLS][D+1]
  float* w = fb + SC_COLS * ldb;                // [SC_ROWS][SC_COLS] gradient weights of the current tile
  const int a0 = blockIdx.x * SC_ROWS, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sc_stage(fa, D, f, a0, SC_ROWS, R, D);
  float lse_a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int a = a0 + wave * 4 + r; lse_a[r] = a < R ? lse[a] : 0.f; }
  float acc[SC_ROWS * SC_MAXD / 256];
#pragma unroll
  for (int j = 0; j < SC_ROWS * SC_MAXD / 256; ++j) acc[j] = 0.f;
  for (int b0 = 0; b0 < R; b0 += SC_COLS) {
    __syncthreads();
    sc_stage(fb, ldb, f, b0, SC_COLS, R, D);
    __syncthreads();
    const int b = b0 + lane;
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < D; ++k) {
      const float vb = fb[lane * ldb + k];
#pragma unroll
      for (int r = 0; r < 4; ++r) dot[r] += fa[(wave * 4 + r) * D + k] * vb;
    }
    const float lse_b = b < R ? lse[b] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = a0 + wave * 4 + r;
      float g = 0.f;
      if (b < R && a < R && b != a) {
        const float z = dot[r] / temp;
        g = __expf(z - lse_a[r]) + __expf(z - lse_b) - (b == (a + N) % R ? 2.f : 0.f);
      }
      w[(wave * 4 + r) * SC_COLS + lane] = g;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SC_ROWS * SC_MAXD / 256; ++j) {
      const int i = threadIdx.x + j * 256;
      if (i < SC_ROWS * D) {
        const int r = i / D, k = i - r * D;
        float t = 0.f;
        for (int c = 0; c < SC_COLS; ++c) t += w[r * SC_COLS + c] * fb[c * ldb + k];
        acc[j] += t;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < SC_ROWS * SC_MAXD / 256; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < SC_ROWS * D) {
      const int r = i / D, k = i - r * D;
      if (a0 + r < R) df[(size_t)(a0 + r) * D + k] = acc[j] * cscale;
    }
  }
}
__global__ void supcon_reduce_kernel(const float* __restrict__ row_loss, float* __restrict__ loss_out, int R,
                                     float coef) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < R; i += blockDim.x) s += row_loss[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss_out = coef * (red[0] + red[1] + red[2] + red[3]) / (float)R;
}

// =========================================================================== Adam (torch defaults; roco_train.py:90)
template <int U>
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n4, float step_size, float b1, float omb1, float b2,
                            float omb2, float eps, float bc2_sqrt, float gscale, int zero_grad) {
  // 32 bytes of HBM traffic per parameter and nothing else: two float4 of each array in flight per thread
  // (one per iteration left the pass at 2.4 TB/s), m and v streamed past the caches (they are not touched
  // again before the next optimizer step; the parameters are: the forward re-reads them)
  const long stride = (long)gridDim.x * blockDim.x;
  const float inv_bc2 = 1.0f / bc2_sqrt;
  auto upd = [&](f32x4& pv, const f32x4& gv, f32x4& mv, f32x4& vv) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gscale;
      mv[e] = mv[e] + (gg - mv[e]) * omb1;           // exp_avg.lerp_(grad, 1 - beta1)
      vv[e] = vv[e] * b2 + gg * gg * omb2;           // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] = pv[e] - step_size * (mv[e] / denom);
    }
  };
  (void)inv_bc2;
  f32x4* P4 = reinterpret_cast<f32x4*>(p); f32x4* G4 = reinterpret_cast<f32x4*>(g);
  f32x4* M4 = reinterpret_cast<f32x4*>(m); f32x4* V4 = reinterpret_cast<f32x4*>(v);
  const f32x4 zero = {0, 0, 0, 0};
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      pv[u] = P4[j]; gv[u] = G4[j]; mv[u] = __builtin_nontemporal_load(M4 + j); vv[u] = __builtin_nontemporal_load(V4 + j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) upd(pv[u], gv[u], mv[u], vv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      P4[j] = pv[u]; __builtin_nontemporal_store(mv[u], M4 + j); __builtin_nontemporal_store(vv[u], V4 + j);
      if (zero_grad) G4[j] = zero;
    }
  }
  for (; i < n4; i += stride) {
    f32x4 p0 = P4[i], g0 = G4[i], m0 = M4[i], v0 = V4[i];
    upd(p0, g0, m0, v0);
    P4[i] = p0; M4[i] = m0; V4[i] = v0;
    if (zero_grad) G4[i] = zero;
  }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] += a * x[i];
}

// out[c] += sum_rows x[row][c]   (bias gradients of the outermost linears)
__global__ void colsum_kernel(const float* __restrict__ x, int ld, int rows, int cols, float* __restrict__ out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  int r0 = blockIdx.y * 64, r1 = min(rows, r0 + 64);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += x[(size_t)r * ld + c];
  atomicAdd(&out[c], s);
}

// dropout applied in place on a dense tensor (used where no GEMM epilogue is available)
__global__ void dropout_kernel(float* __restrict__ x, long n, float p, uint32_t seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  const float ks = 1.0f / (1.0f - p);
  for (; i < n; i += stride) x[i] = (rng_uniform(seed, (uint32_t)i) >= p) ? x[i] * ks : 0.f;
}

// y = dropout(x) with the same (seed, linear index) stream as the GEMM epilogue; used on the
// backward side of a residual branch: d(branch) = mask * dy / (1-p) while dy itself flows on.
__global__ void dropout_copy_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                                    uint32_t seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  const float ks = 1.0f / (1.0f - p);
  for (; i < n; i += stride) y[i] = (rng_uniform(seed, (uint32_t)i) >= p) ? x[i] * ks : 0.f;
}

// =========================================================================== host launchers
static inline int grid_for(long n, int block = 256, int cap = 2048) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int k_bn_coef_fwd(hipStream_t st, const double* stat, int C, double count, float eps, const float* gamma,
                  const float* beta, float* run_mean, float* run_var, long long* nbt, float momentum, int reps,
                  int training, float* scale, float* shift, float* mean, float* invstd) {
  hipLaunchKernelGGL(bn_coef_fwd_kernel, dim3(cdiv_i(C, 16)), dim3(256), 0, st, stat, C, count, eps, gamma, beta,
                     run_mean, run_var, nbt, pow(1.0 - (double)momentum, (double)reps), reps, training, scale, shift,
                     mean, invstd);   // (1 - momentum)^reps on the host: a double pow() per thread was most of the kernel's math
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

// Block end with the coefficients of bn3 (and of the downsample BatchNorm) folded from their raw sums (mmvqa_bn_fold):
// no coefficient launch between conv3 and this kernel.  One workgroup = 64 channels x `rpw` rows; 64 (+64) threads fold
// one channel each into LDS; the workgroups of row block 0 publish scale / shift / mean / invstd / running statistics.
// out = POST(PRE(bn(z)) + idn'),  idn' = bn_d(idn) | idn | nothing:  ResNet Bottleneck <NONE, RELU>, the EfficientNet
// blocks <SILU, NONE> / <NONE, NONE> (models/image_encoding.py:72-86, 89-115)
template <int PRE, int POST>
__global__ __launch_bounds__(256) void bn_add_relu_fold_kernel(const float* __restrict__ z, const BnFold f3,
                                                               const float* __restrict__ idn, const BnFold fd, int has_d,
                                                               float* __restrict__ out, long rows, int C, int rpw) {
  __shared__ float cs[4][64];
  const int tid = threadIdx.x, c0 = blockIdx.y * 64;
  const bool pub = blockIdx.x == 0;
  if (tid < 64) {
    float a = 0.f, b = 0.f;
    if (c0 + tid < C) bn_fold_fwd(f3, C, c0 + tid, pub && f3.publish, a, b);
    cs[0][tid] = a; cs[1][tid] = b;
  } else if (tid < 128) {
    float a = 1.f, b = 0.f;
    if (has_d && c0 + tid - 64 < C) bn_fold_fwd(fd, C, c0 + tid - 64, pub && fd.publish, a, b);
    cs[2][tid - 64] = a; cs[3][tid - 64] = b;
  }
  if (pub && blockIdx.y == 0) {
    if (tid == 0 && f3.publish && f3.nbt) *f3.nbt += f3.reps;
    if (tid == 64 && has_d && fd.publish && fd.nbt) *fd.nbt += fd.reps;
  }
  __syncthreads();
  const int q = tid & 15, rl = tid >> 4;
  const int c = c0 + q * 4;
  if (c >= C) return;
  const f32x4 s3 = *reinterpret_cast<const f32x4*>(&cs[0][q * 4]), b3 = *reinterpret_cast<const f32x4*>(&cs[1][q * 4]);
  const f32x4 sd = *reinterpret_cast<const f32x4*>(&cs[2][q * 4]), bd = *reinterpret_cast<const f32x4*>(&cs[3][q * 4]);
  const long r0 = (long)blockIdx.x * rpw, r1 = r0 + rpw < rows ? r0 + rpw : rows;
  const f32x4 zero = {0, 0, 0, 0};
  for (long r = r0 + rl; r < r1; r += 32) {
    const long i0 = r * C + c, i1 = (r + 16) * C + c;
    const bool two = r + 16 < r1;
    const f32x4 z0 = *reinterpret_cast<const f32x4*>(z + i0), d0 = idn ? *reinterpret_cast<const f32x4*>(idn + i0) : zero;
    f32x4 z1 = z0, d1 = d0;
    if (two) { z1 = *reinterpret_cast<const f32x4*>(z + i1); if (idn) d1 = *reinterpret_cast<const f32x4*>(idn + i1); }
    f32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t0 = act_fwd(PRE, z0[j] * s3[j] + b3[j]) + (d0[j] * sd[j] + bd[j]);
      const float t1 = act_fwd(PRE, z1[j] * s3[j] + b3[j]) + (d1[j] * sd[j] + bd[j]);
      o0[j] = act_fwd(POST, t0);
      o1[j] = act_fwd(POST, t1);
    }
    *reinterpret_cast<f32x4*>(out + i0) = o0;
    if (two) *reinterpret_cast<f32x4*>(out + i1) = o1;
  }
}

int k_bn_act_add_fold(hipStream_t st, const float* z, const mmvqa_bn_fold* f3, int pre_act, const float* idn,
                      const mmvqa_bn_fold* fd, int post_act, float* out, long rows, int C) {
  if (C % 4 != 0 || !f3 || !f3->stat || f3->bwd || (fd && (!fd->stat || fd->bwd || !idn)))
    return mmvqa_set_error(MMVQA_ERR_ARG, "bn_act_add_fold: C=%d must be a multiple of 4 and the folds forward ones", C);
  const int cb = cdiv_i(C, 64);
  // rows per workgroup: every workgroup pays the fold of its 64 channels (~1.5 us of setup) before it streams its rows, so
  // few, long workgroups; MMVQA_BAR_WGS = target number of workgroups (A/B)
  static const long target = getenv("MMVQA_BAR_WGS") ? atol(getenv("MMVQA_BAR_WGS")) : 4096;
  int rpw = 32;
  while ((long)cdiv_i(rows, rpw) * cb > target && rpw < 1024) rpw *= 2;
  mmvqa_bn_fold none;
  memset(&none, 0, sizeof(none));
  const dim3 grid(cdiv_i(rows, rpw), cb);
#define BAR_GO(PRE_, POST_)                                                                                            \
  hipLaunchKernelGGL((bn_add_relu_fold_kernel<PRE_, POST_>), grid, dim3(256), 0, st, z, *f3, idn, fd ? *fd : none, fd ? 1 : 0, \
                     out, rows, C, rpw)
  if (pre_act == ACT_NONE && post_act == ACT_RELU) BAR_GO(ACT_NONE, ACT_RELU);
  else if (pre_act == ACT_SILU && post_act == ACT_NONE) BAR_GO(ACT_SILU, ACT_NONE);
  else if (pre_act == ACT_NONE && post_act == ACT_NONE) BAR_GO(ACT_NONE, ACT_NONE);
  else return mmvqa_set_error(MMVQA_ERR_ARG, "bn_act_add_fold: activation pair (%d, %d) not built", pre_act, post_act);
#undef BAR_GO
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_bn_add_relu_fold(hipStream_t st, const float* z, const mmvqa_bn_fold* f3, const float* idn, const mmvqa_bn_fold* fd,
                       float* out, long rows, int C) {
  if (!idn) return mmvqa_set_error(MMVQA_ERR_ARG, "bn_add_relu_fold: no identity tensor");
  return k_bn_act_add_fold(st, z, f3, ACT_NONE, idn, fd, ACT_RELU, out, rows, C);
}

int k_bn_coef_fwd_keep(hipStream_t st, const double* stat, int C, double count, float eps, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, long long* nbt, double keep, int reps,
                       int training, float* scale, float* shift, float* mean, float* invstd) {
  hipLaunchKernelGGL(bn_coef_fwd_kernel, dim3(cdiv_i(C, 16)), dim3(256), 0, st, stat, C, count, eps, gamma, beta,
                     run_mean, run_var, nbt, keep, reps, training, scale, shift, mean, invstd);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_bn_coef_bwd(hipStream_t st, const double* stat, int C, double count, const float* gamma, const float* mean,
                  const float* invstd, int training, float* P, float* Q, float* R, float* dgamma, float* dbeta) {
  hipLaunchKernelGGL(bn_coef_bwd_kernel, dim3(cdiv_i(C, 16)), dim3(256), 0, st, stat, C, count, gamma, mean,
                     invstd, training, P, Q, R, dgamma, dbeta);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_bn_add_relu(hipStream_t st, const float* z, const float* s, const float* b, const float* idn,
                  const float* ids, const float* idb, float* out, long rows, int C) {
  long n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_add_relu_kernel, dim3(grid_for(n4)), dim3(256), 0, st, z, s, b, idn, ids, idb, out, n4,
                     C / 4);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_maxpool_fwd(hipStream_t st, const float* z, const float* s, const float* b, float* out, unsigned char* idx,
                  int N, int H, int W, int C, int OH, int OW) {
  long total = (long)N * OH * OW * C / 4;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, z, s, b, out, idx, N, H, W, C,
                     OH, OW);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_maxpool_bwd(hipStream_t st, const float* gp, const unsigned char* idx, const float* extra, const float* z,
                  const float* s, const float* b, const float* mean, const float* invstd, float* g0, double* stat,
                  int N, int H, int W, int C, int OH, int OW) {
  if (C % 4 != 0 || 256 % (C / 4) != 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "maxpool_bwd: C/4 must divide 256 (C=%d)", C);
  long total = (long)N * H * W * C / 4;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 256, 1024)), dim3(256), 0, st, gp, idx, extra, z, s,
                     b, mean, invstd, g0, stat, N, H, W, C, OH, OW);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_layernorm_fwd(hipStream_t st, const float* x, const float* res, const float* gamma, const float* beta,
                    float* y, float* sum_out, float* mean, float* rstd, int rows, int H, float eps) {
  if (H % 4 != 0 || H > 256 * LN_MAXV) return mmvqa_set_error(MMVQA_ERR_ARG, "layernorm: H=%d unsupported", H);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv_i(rows, 4)), dim3(256), 0, st, x, res, gamma, beta, y,
                     sum_out, mean, rstd, rows, H, eps);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_layernorm_bwd(hipStream_t st, const float* dy, const float* x, const float* gamma, const float* mean,
                    const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int H) {
  if (H % 4 != 0 || H > 256 * LN_MAXV) return mmvqa_set_error(MMVQA_ERR_ARG, "layernorm: H=%d unsupported", H);
  int rpw = rows >= 2048 ? 4 : (rows >= 512 ? 2 : 1);
  int grid = cdiv_i(rows, 4 * rpw);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(256), 2 * H * sizeof(float), st, dy, x, gamma, mean,
                     rstd, dres, dx, dgamma, dbeta, rows, H, rpw);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_embed_fwd(hipStream_t st, const long long* ids, const long long* seg, const float* word, const float* pos,
                const float* type, const float* gamma, const float* beta, const float* vis, float* out,
                float* xhat, float* rstd, int B, int T, int H, int num_vis, float eps, float drop_p,
                uint32_t seed) {
  if (H % 4 != 0 || H > 256 * LN_MAXV) return mmvqa_set_error(MMVQA_ERR_ARG, "embed: H=%d unsupported", H);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv_i((long)B * T, 4)), dim3(256), 0, st, ids, seg, word, pos, type,
                     gamma, beta, vis, out, xhat, rstd, B, T, H, num_vis, eps, drop_p, seed);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_embed_bwd(hipStream_t st, const float* dout, const long long* ids, const long long* seg, const float* xhat,
                const float* rstd, const float* gamma, float* dword, float* dpos, float* dtype, float* dgamma,
                float* dbeta, float* dvis, int B, int T, int H, int num_vis, float drop_p, uint32_t seed,
                int pad_idx) {
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv_i((long)B * T, 4)), dim3(256), 0, st, dout, ids, seg, xhat, rstd,
                     gamma, dword, dpos, dtype, dgamma, dbeta, dvis, B, T, H, num_vis, drop_p, seed, pad_idx);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_meanpool_fwd(hipStream_t st, const float* h, const long long* mask, float* out, int B, int T, int H) {
  hipLaunchKernelGGL(meanpool_fwd_kernel, dim3(B), dim3(256), 0, st, h, mask, out, B, T, H);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}
int k_meanpool_bwd(hipStream_t st, const float* dout, const long long* mask, float* dh, int B, int T, int H,
                   int accumulate) {
  hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(B), dim3(256), 0, st, dout, mask, dh, B, T, H, accumulate);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

static bool lsm_fast_ok(const float* logits, int ld, int V) {
  return ((uintptr_t)logits & 15) == 0 && (ld & 3) == 0 && ld >= ((V + 3) & ~3);
}
int k_lsm_grad(hipStream_t st, const float* logits, int ld, const long long* target, const float* row_lse,
               float* dlogits, int dld, const float* gscale_ptr, float gscale_mul, int rows, int V) {
  if (!lsm_fast_ok(logits, ld, V) || !lsm_fast_ok(dlogits, dld, V))
    return mmvqa_set_error(MMVQA_ERR_ARG, "mlm_grad: logits/dlogits rows must be 16-byte aligned with ld >= round_up(V,4)");
  hipLaunchKernelGGL(lsm_grad_kernel, dim3(rows, cdiv_i((V + 3) >> 2, 256 * LSM_U)), dim3(256), 0, st, logits, ld, target,
                     row_lse, dlogits, dld, gscale_ptr, gscale_mul, V);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}
int k_lsm_nll(hipStream_t st, const float* logits, int ld, const long long* target, float* row_loss, float* row_lse,
              long long* pred, float* dlogits, int dld, const float* gscale_ptr, float gscale_mul, int rows, int V,
              float* out3) {
  if (row_lse && lsm_fast_ok(logits, ld, V) && (!dlogits || lsm_fast_ok(dlogits, dld, V))) {
    hipLaunchKernelGGL(lsm_stats_kernel, dim3(rows), dim3(1024), 0, st, logits, ld, target, row_loss, row_lse, pred, V);
    KERNEL_CHECK_RET();
    if (dlogits) {
      int r = k_lsm_grad(st, logits, ld, target, row_lse, dlogits, dld, gscale_ptr, gscale_mul, rows, V);
      if (r != MMVQA_OK) return r;
    }
  } else {   // unaligned rows: scalar three-pass form
    hipLaunchKernelGGL(lsm_nll_kernel, dim3(rows), dim3(256), 0, st, logits, ld, target, row_loss, row_lse, pred,
                       dlogits, dld, gscale_ptr, gscale_mul, V);
    KERNEL_CHECK_RET();
  }
  if (out3) {
    hipLaunchKernelGGL(mlm_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, pred, target, rows, out3);
    KERNEL_CHECK_RET();
  }
  return MMVQA_OK;
}

int k_asl(hipStream_t st, const float* logits, int ld, const long long* target, float* row_loss, float* dlogits,
          int dld, int rows, int C, float gpos, float gneg, float eps, float gscale) {
  hipLaunchKernelGGL(asl_kernel, dim3(rows), dim3(256), 0, st, logits, ld, target, row_loss, dlogits, dld, C, gpos,
                     gneg, eps, gscale);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_l2norm_fwd(hipStream_t st, const float* x, float* y, float* nrm, int rows, int D) {
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv_i(rows, 4)), dim3(256), 0, st, x, y, nrm, rows, D);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}
int k_l2norm_bwd(hipStream_t st, const float* dy, const float* y, const float* nrm, float* dx, int rows, int D) {
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv_i(rows, 4)), dim3(256), 0, st, dy, y, nrm, dx, rows, D);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_supcon(hipStream_t st, const float* f, float* loss, float* df, float* ws, int N, int D, float temp,
             float base_temp, float gscale) {
  const int R = 2 * N;
  if (N < 1 || D < 1 || D > SC_MAXD) return mmvqa_set_error(MMVQA_ERR_ARG, "supcon: N=%d D=%d (1 <= D <= %d)", N, D, SC_MAXD);
  if (!ws) return mmvqa_set_error(MMVQA_ERR_ARG, "supcon: workspace of 4*N floats required (lse, row losses)");
  float* lse = ws;
  float* row_loss = ws + R;
  const int blocks = cdiv_i(R, SC_ROWS);
  const size_t sm1 = ((size_t)SC_ROWS * D + (size_t)SC_COLS * (D + 1)) * sizeof(float);
  const size_t sm2 = sm1 + (size_t)SC_ROWS * SC_COLS * sizeof(float);
  if (sm2 > 48 * 1024) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)supcon_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)supcon_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2));
  }
  hipLaunchKernelGGL(supcon_rows_kernel, dim3(blocks), dim3(256), sm1, st, f, lse, row_loss, N, D, temp);
  KERNEL_CHECK_RET();
  const float coef = temp / base_temp;
  if (df) {
    hipLaunchKernelGGL(supcon_grad_kernel, dim3(blocks), dim3(256), sm2, st, f, lse, df, N, D, temp,
                       coef / (float)R / temp * gscale);
    KERNEL_CHECK_RET();
  }
  hipLaunchKernelGGL(supcon_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, loss, R, coef);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_adam(hipStream_t st, float* p, float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
           int step, float gscale, int zero_grad) {
  if (n % 4 != 0) return mmvqa_set_error(MMVQA_ERR_ARG, "adam: n must be a multiple of 4");
  // bias corrections and (1 - beta) in double, as torch.optim.Adam computes them from Python floats
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  // MMVQA_ADAM_WGS: cap on the workgroups of one launch (A/B: a ranged launch beside the backward pass takes wave slots
  // and bandwidth from the GEMMs it shares the chip with)
  const char* cap_s = getenv("MMVQA_ADAM_WGS");
  const int cap = cap_s ? atoi(cap_s) : 16384;   // (tools/adam_bench.py: 0.80 ms at 4096, 0.705 ms = 6.0 TB/s at 16384, 133 M parameters)
  // MMVQA_ADAM_UNROLL: float4 of each array in flight per thread (A/B; default 2: one per iteration left the pass at 2.4 TB/s)
  static const int unroll = getenv("MMVQA_ADAM_UNROLL") ? atoi(getenv("MMVQA_ADAM_UNROLL")) : 2;
#define ADAM_GO(U_)                                                                                                       \
  hipLaunchKernelGGL(adam_kernel<U_>, dim3(grid_for(n / 4, 256, cap > 0 ? cap : 16384)), dim3(256), 0, st, p, g, m, v, n / 4, \
                     (float)(lr / bc1), (float)b1, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps,            \
                     (float)sqrt(bc2), gscale, zero_grad)
  if (unroll >= 4) ADAM_GO(4); else if (unroll == 3) ADAM_GO(3); else if (unroll == 1) ADAM_GO(1); else ADAM_GO(2);
#undef ADAM_GO
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_axpy(hipStream_t st, float* y, const float* x, float a, long n) {
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, st, y, x, a, n);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_colsum(hipStream_t st, const float* x, int ld, int rows, int cols, float* out) {
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv_i(cols, 256), cdiv_i(rows, 64)), dim3(256), 0, st, x, ld, rows, cols,
                     out);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_dropout(hipStream_t st, float* x, long n, float p, uint32_t seed) {
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, p, seed);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

// per output pixel: which filter taps read inside the image (the padding pattern of nn.Conv2d)
__global__ void pixmask_kernel(int* out, int total, int OH, int OW, int SH, int SW, int KH, int KW, int stride, int pad) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= total) return;
  const int rem = m % (OH * OW), oy = rem / OW, ox = rem - oy * OW;
  const int y0 = oy * stride - pad, x0 = ox * stride - pad;
  unsigned mk = 0;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw)
      if ((unsigned)(y0 + kh) < (unsigned)SH && (unsigned)(x0 + kw) < (unsigned)SW) mk |= 1u << (kh * KW + kw);
  out[m] = (int)mk;
}

int k_pixmask(hipStream_t st, int* out, int N, int OH, int OW, int SH, int SW, int KH, int KW, int stride, int pad) {
  if (KH * KW > 32) return mmvqa_set_error(MMVQA_ERR_ARG, "pixmask: at most 32 taps");
  const int total = N * OH * OW;
  if (total <= 0) return MMVQA_OK;
  hipLaunchKernelGGL(pixmask_kernel, dim3((total + 255) / 256), dim3(256), 0, st, out, total, OH, OW, SH, SW, KH, KW, stride, pad);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_dropout_copy(hipStream_t st, const float* x, float* y, long n, float p, uint32_t seed) {
  hipLaunchKernelGGL(dropout_copy_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n, p, seed);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

// =========================================================================== EfficientNetV2 pieces
// (timm tf_efficientnetv2_m as the reference instantiates it, models/image_encoding.py:15,26,100-115)

// out = post( pre(z*s+b) + idn' ), idn' = idn | iact(idn*is+ib) | nothing
__global__ void bn_act_add_kernel(const float* __restrict__ z, const float* __restrict__ s,
                                  const float* __restrict__ b, int pre_act, const float* __restrict__ idn,
                                  const float* __restrict__ ids, const float* __restrict__ idb, int idn_act,
                                  int post_act, float* __restrict__ out, long n4, int C4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    int c = (int)(i % C4) * 4;
    f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
    f32x4 sv = *reinterpret_cast<const f32x4*>(s + c), bv = *reinterpret_cast<const f32x4*>(b + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_fwd(pre_act, zv[j] * sv[j] + bv[j]);
    if (idn) {
      f32x4 iv = reinterpret_cast<const f32x4*>(idn)[i];
      if (ids) {
        f32x4 s2 = *reinterpret_cast<const f32x4*>(ids + c), b2 = *reinterpret_cast<const f32x4*>(idb + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) iv[j] = act_fwd(idn_act, iv[j] * s2[j] + b2[j]);
      }
      o += iv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_fwd(post_act, o[j]);
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// Depthwise 3x3 (groups = C), NHWC.  Workgroup = 32 pixel lanes x 8 channel quads (32 channels):
// the per-channel statistics are reduced over the 32 pixel lanes in LDS, then one atomic per channel.
//   forward : z2 = dw(silu(z1*s1+b1)) ; statistics of z2
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ z1, const float* __restrict__ s1,
                                                        const float* __restrict__ b1, const BnFold f1, const float* __restrict__ w,
                                                        float* __restrict__ z2, double* __restrict__ stat, int N,
                                                        int H, int W, int C, int OH, int OW, int stride, int pad) {
  __shared__ double red[256 * 8];
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.y * 32 + q * 4;
  const long npix = (long)N * OH * OW;
  __shared__ __attribute__((aligned(16))) float cf[2][32];
  bn_coef_block_fwd<32>(s1, b1, f1, C, blockIdx.y * 32, f1.publish && blockIdx.x == 0, cf);
  const f32x4 sv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), bv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]);
  if (f1.stat && f1.publish && f1.nbt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *f1.nbt += f1.reps;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[t][j] = w[(size_t)(c + j) * 9 + t];
  double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
  for (long pix = (long)blockIdx.x * 32 + pl; pix < npix; pix += (long)gridDim.x * 32) {
    int ox = (int)(pix % OW); long t = pix / OW;
    int oy = (int)(t % OH), n = (int)(t / OH);
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int y = oy * stride - pad + kh;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int x = ox * stride - pad + kw;
        if (x < 0 || x >= W) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(z1 + ((size_t)(n * H + y) * W + x) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += silu_f(v[j] * sv[j] + bv[j]) * wv[kh * 3 + kw][j];
      }
    }
    *reinterpret_cast<f32x4*>(z2 + (size_t)pix * C + c) = acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sa[j] += (double)acc[j]; sb[j] += (double)acc[j] * (double)acc[j]; }
  }
  if (stat) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = sa[j]; red[threadIdx.x * 8 + 4 + j] = sb[j]; }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int qq = threadIdx.x >> 2, j = threadIdx.x & 3;
      double a0 = 0, a1 = 0;
      for (int r = 0; r < 32; ++r) { a0 += red[(r * 8 + qq) * 8 + j]; a1 += red[(r * 8 + qq) * 8 + 4 + j]; }
      const int slot = blockIdx.x & (MMVQA_STAT_SLOTS - 1);
      double* d = stat + ((size_t)slot * C + blockIdx.y * 32 + qq * 4 + j) * 2;
      atomicAdd(d, a0);
      atomicAdd(d + 1, a1);
    }
  }
}

//   backward (data): da1 = dw^T(dz2), dz2 = P*g + Q*z2 + R ; du1 = da1 * silu'(z1*s1+b1) -> g1 ; BN1-backward sums
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(
    const float* __restrict__ g2, const float* __restrict__ z2, const float* __restrict__ P, const float* __restrict__ Q,
    const float* __restrict__ R, const float* __restrict__ w, const float* __restrict__ z1, const float* __restrict__ s1,
    const float* __restrict__ b1, const float* __restrict__ mean1, const float* __restrict__ invstd1,
    float* __restrict__ g1, double* __restrict__ stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad,
    const BnFold f2) {
  __shared__ double red[256 * 8];
  __shared__ __attribute__((aligned(16))) float cf[3][32];
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.y * 32 + q * 4;
  const long npix = (long)N * H * W;
  bn_coef_block_bwd<32>(P, Q, R, f2, C, blockIdx.y * 32, f2.publish && blockIdx.x == 0, cf);
  f32x4 sv = *reinterpret_cast<const f32x4*>(s1 + c), bv = *reinterpret_cast<const f32x4*>(b1 + c);
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean1 + c), is = *reinterpret_cast<const f32x4*>(invstd1 + c);
  f32x4 Pv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), Qv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]),
        Rv = *reinterpret_cast<const f32x4*>(&cf[2][q * 4]);
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[t][j] = w[(size_t)(c + j) * 9 + t];
  double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
  for (long pix = (long)blockIdx.x * 32 + pl; pix < npix; pix += (long)gridDim.x * 32) {
    int x = (int)(pix % W); long t = pix / W;
    int y = (int)(t % H), n = (int)(t / H);
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int ty = y + pad - kh;
      if (ty < 0 || ty % stride) continue;
      int oy = ty / stride;
      if (oy >= OH) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int tx = x + pad - kw;
        if (tx < 0 || tx % stride) continue;
        int ox = tx / stride;
        if (ox >= OW) continue;
        size_t o = ((size_t)(n * OH + oy) * OW + ox) * C + c;
        f32x4 gv = *reinterpret_cast<const f32x4*>(g2 + o), zv = *reinterpret_cast<const f32x4*>(z2 + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += (gv[j] * Pv[j] + zv[j] * Qv[j] + Rv[j]) * wv[kh * 3 + kw][j];
      }
    }
    f32x4 z = *reinterpret_cast<const f32x4*>(z1 + (size_t)pix * C + c), o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o4[j] = acc[j] * dsilu_f(z[j] * sv[j] + bv[j]);
      sa[j] += (double)o4[j];
      sb[j] += (double)(o4[j] * ((z[j] - mu[j]) * is[j]));
    }
    *reinterpret_cast<f32x4*>(g1 + (size_t)pix * C + c) = o4;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = sa[j]; red[threadIdx.x * 8 + 4 + j] = sb[j]; }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int qq = threadIdx.x >> 2, j = threadIdx.x & 3;
    double a0 = 0, a1 = 0;
    for (int r = 0; r < 32; ++r) { a0 += red[(r * 8 + qq) * 8 + j]; a1 += red[(r * 8 + qq) * 8 + 4 + j]; }
    const int slot = blockIdx.x & (MMVQA_STAT_SLOTS - 1);
    double* d = stat + ((size_t)slot * C + blockIdx.y * 32 + qq * 4 + j) * 2;
    atomicAdd(d, a0);
    atomicAdd(d + 1, a1);
  }
}

//   backward (weight): dW[c][tap] += sum_pix dz2[pix,c] * silu(z1*s1+b1)[pix@tap, c]
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(
    const float* __restrict__ g2, const float* __restrict__ z2, const float* __restrict__ P, const float* __restrict__ Q,
    const float* __restrict__ R, const float* __restrict__ z1, const float* __restrict__ s1, const float* __restrict__ b1,
    float* __restrict__ dw, int N, int H, int W, int C, int OH, int OW, int stride, int pad, const BnFold f2) {
  __shared__ float red[32 * 8 * 36];
  __shared__ __attribute__((aligned(16))) float cf[3][32];
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.y * 32 + q * 4;
  const long npix = (long)N * OH * OW;
  bn_coef_block_bwd<32>(P, Q, R, f2, C, blockIdx.y * 32, false, cf);   // (the data gradient publishes)
  f32x4 sv = *reinterpret_cast<const f32x4*>(s1 + c), bv = *reinterpret_cast<const f32x4*>(b1 + c);
  f32x4 Pv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), Qv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]),
        Rv = *reinterpret_cast<const f32x4*>(&cf[2][q * 4]);
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0, 0, 0, 0};
  for (long pix = (long)blockIdx.x * 32 + pl; pix < npix; pix += (long)gridDim.x * 32) {
    int ox = (int)(pix % OW); long t = pix / OW;
    int oy = (int)(t % OH), n = (int)(t / OH);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g2 + (size_t)pix * C + c), zv = *reinterpret_cast<const f32x4*>(z2 + (size_t)pix * C + c), dz;
#pragma unroll
    for (int j = 0; j < 4; ++j) dz[j] = gv[j] * Pv[j] + zv[j] * Qv[j] + Rv[j];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int y = oy * stride - pad + kh;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int x = ox * stride - pad + kw;
        if (x < 0 || x >= W) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(z1 + ((size_t)(n * H + y) * W + x) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[kh * 3 + kw][j] += dz[j] * silu_f(v[j] * sv[j] + bv[j]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(pl * 8 + q) * 36 + j * 9 + t] = acc[t][j];
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 36; i += 256) {   // (quad, channel-in-quad * 9 + tap)
    const int qq = i / 36, e = i - qq * 36;
    float a0 = 0.f;
    for (int r = 0; r < 32; ++r) a0 += red[(r * 8 + qq) * 36 + e];
    atomicAdd(&dw[(size_t)(blockIdx.y * 32 + qq * 4) * 9 + e], a0);
  }
}

// ---- image-tiled forms of the three depthwise kernels (round 2).  The pixel-strided forms above evaluate
// silu(bn(z1)) (forward, weight gradient) and P*g + Q*z2 + R (both gradients) once per TAP: nine transcendental
// evaluations per element.  Here a workgroup owns one image x 32 channels, stages the activated input (or dz2) of the
// WHOLE map in LDS once (14x14x32 floats = 25 KB; 28x28 = 100 KB) and takes the nine taps from there.  Used whenever the
// map fits (every depthwise layer of tf_efficientnetv2_m at 224x224: 28x28 -> 14x14, 14x14, 14x14 -> 7x7, 7x7).
#define DW_TILE_MAX_BYTES (150 * 1024)
__device__ __forceinline__ void dw_reduce_stats(double* red, const double (&sa)[4], const double (&sb)[4],
                                                double* __restrict__ stat, int C, int cbase, int slot) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = sa[j]; red[threadIdx.x * 8 + 4 + j] = sb[j]; }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int qq = threadIdx.x >> 2, j = threadIdx.x & 3;
    double a0 = 0, a1 = 0;
    for (int r = 0; r < 32; ++r) { a0 += red[(r * 8 + qq) * 8 + j]; a1 += red[(r * 8 + qq) * 8 + 4 + j]; }
    double* d = stat + ((size_t)slot * C + cbase + qq * 4 + j) * 2;
    atomicAdd(d, a0);
    atomicAdd(d + 1, a1);
  }
}

__global__ __launch_bounds__(256) void dwconv_fwd_tile_kernel(const float* __restrict__ z1, const float* __restrict__ s1,
                                                             const float* __restrict__ b1, const BnFold f1, const float* __restrict__ w,
                                                             float* __restrict__ z2, double* __restrict__ stat, int H,
                                                             int W, int C, int OH, int OW, int stride, int pad) {
  extern __shared__ __attribute__((aligned(16))) float dwsm[];
  f32x4* act = reinterpret_cast<f32x4*>(dwsm);                              // [H*W][8]
  double* red = reinterpret_cast<double*>(dwsm + (size_t)H * W * 32);       // [256][8]
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int n = blockIdx.y, cb = blockIdx.x * 32, c = cb + q * 4;
  __shared__ __attribute__((aligned(16))) float cf[2][32];   // (folded: the image-0 workgroup of a channel block publishes)
  bn_coef_block_fwd<32>(s1, b1, f1, C, cb, f1.publish && n == 0, cf);
  const f32x4 sv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), bv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]);
  if (f1.stat && f1.publish && f1.nbt && blockIdx.x == 0 && n == 0 && threadIdx.x == 0) *f1.nbt += f1.reps;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[t][j] = w[(size_t)(c + j) * 9 + t];
  const float* zin = z1 + (size_t)n * H * W * C + c;
  for (int p = pl; p < H * W; p += 32) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(zin + (size_t)p * C);
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = silu_f(v[j] * sv[j] + bv[j]);
    act[p * 8 + q] = a;
  }
  __syncthreads();
  double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
  float* zout = z2 + (size_t)n * OH * OW * C + c;
  for (int op = pl; op < OH * OW; op += 32) {
    const int oy = op / OW, ox = op - oy * OW;
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int y = oy * stride - pad + kh;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int x = ox * stride - pad + kw;
        if (x < 0 || x >= W) continue;
        const f32x4 a = act[(y * W + x) * 8 + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += a[j] * wv[kh * 3 + kw][j];
      }
    }
    *reinterpret_cast<f32x4*>(zout + (size_t)op * C) = acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sa[j] += (double)acc[j]; sb[j] += (double)acc[j] * (double)acc[j]; }
  }
  if (stat) dw_reduce_stats(red, sa, sb, stat, C, cb, n & (MMVQA_STAT_SLOTS - 1));
}

__global__ __launch_bounds__(256) void dwconv_bwd_data_tile_kernel(
    const float* __restrict__ g2, const float* __restrict__ z2, const float* __restrict__ P, const float* __restrict__ Q,
    const float* __restrict__ R, const float* __restrict__ w, const float* __restrict__ z1, const float* __restrict__ s1,
    const float* __restrict__ b1, const float* __restrict__ mean1, const float* __restrict__ invstd1,
    float* __restrict__ g1, double* __restrict__ stat, int H, int W, int C, int OH, int OW, int stride, int pad,
    const BnFold f2) {
  extern __shared__ __attribute__((aligned(16))) float dwsm[];
  __shared__ __attribute__((aligned(16))) float cf[3][32];
  f32x4* dzs = reinterpret_cast<f32x4*>(dwsm);                              // [OH*OW][8]  dz2 = P*g2 + Q*z2 + R
  double* red = reinterpret_cast<double*>(dwsm + (size_t)OH * OW * 32);
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int n = blockIdx.y, cb = blockIdx.x * 32, c = cb + q * 4;
  bn_coef_block_bwd<32>(P, Q, R, f2, C, cb, f2.publish && n == 0, cf);   // (folded: the image-0 workgroup of a channel block publishes)
  const f32x4 sv = *reinterpret_cast<const f32x4*>(s1 + c), bv = *reinterpret_cast<const f32x4*>(b1 + c);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean1 + c), is = *reinterpret_cast<const f32x4*>(invstd1 + c);
  const f32x4 Pv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), Qv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]),
              Rv = *reinterpret_cast<const f32x4*>(&cf[2][q * 4]);
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[t][j] = w[(size_t)(c + j) * 9 + t];
  const size_t ob = (size_t)n * OH * OW * C + c;
  for (int op = pl; op < OH * OW; op += 32) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g2 + ob + (size_t)op * C), zv = *reinterpret_cast<const f32x4*>(z2 + ob + (size_t)op * C);
    f32x4 d;
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = gv[j] * Pv[j] + zv[j] * Qv[j] + Rv[j];
    dzs[op * 8 + q] = d;
  }
  __syncthreads();
  double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
  const size_t ib = (size_t)n * H * W * C + c;
  for (int p = pl; p < H * W; p += 32) {
    const int y = p / W, x = p - y * W;
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ty = y + pad - kh;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tx = x + pad - kw;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= OW) continue;
        const f32x4 d = dzs[(oy * OW + ox) * 8 + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += d[j] * wv[kh * 3 + kw][j];
      }
    }
    const f32x4 z = *reinterpret_cast<const f32x4*>(z1 + ib + (size_t)p * C);
    f32x4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o4[j] = acc[j] * dsilu_f(z[j] * sv[j] + bv[j]);
      sa[j] += (double)o4[j];
      sb[j] += (double)(o4[j] * ((z[j] - mu[j]) * is[j]));
    }
    *reinterpret_cast<f32x4*>(g1 + ib + (size_t)p * C) = o4;
  }
  dw_reduce_stats(red, sa, sb, stat, C, cb, n & (MMVQA_STAT_SLOTS - 1));
}

__global__ __launch_bounds__(256) void dwconv_bwd_weight_tile_kernel(
    const float* __restrict__ g2, const float* __restrict__ z2, const float* __restrict__ P, const float* __restrict__ Q,
    const float* __restrict__ R, const float* __restrict__ z1, const float* __restrict__ s1, const float* __restrict__ b1,
    float* __restrict__ dw, int H, int W, int C, int OH, int OW, int stride, int pad, const BnFold f2) {
  extern __shared__ __attribute__((aligned(16))) float dwsm[];
  __shared__ __attribute__((aligned(16))) float cf[3][32];
  f32x4* act = reinterpret_cast<f32x4*>(dwsm);                              // [H*W][8]
  f32x4* dzs = act + (size_t)H * W * 8;                                     // [OH*OW][8]
  float* red = dwsm + ((size_t)H * W + (size_t)OH * OW) * 32;               // [32*8][36]
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int n = blockIdx.y, cb = blockIdx.x * 32, c = cb + q * 4;
  bn_coef_block_bwd<32>(P, Q, R, f2, C, cb, false, cf);   // (the data gradient publishes)
  const f32x4 sv = *reinterpret_cast<const f32x4*>(s1 + c), bv = *reinterpret_cast<const f32x4*>(b1 + c);
  const f32x4 Pv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), Qv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]),
              Rv = *reinterpret_cast<const f32x4*>(&cf[2][q * 4]);
  const size_t ib = (size_t)n * H * W * C + c, ob = (size_t)n * OH * OW * C + c;
  for (int p = pl; p < H * W; p += 32) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(z1 + ib + (size_t)p * C);
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = silu_f(v[j] * sv[j] + bv[j]);
    act[p * 8 + q] = a;
  }
  for (int op = pl; op < OH * OW; op += 32) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g2 + ob + (size_t)op * C), zv = *reinterpret_cast<const f32x4*>(z2 + ob + (size_t)op * C);
    f32x4 d;
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = gv[j] * Pv[j] + zv[j] * Qv[j] + Rv[j];
    dzs[op * 8 + q] = d;
  }
  __syncthreads();
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0, 0, 0, 0};
  for (int op = pl; op < OH * OW; op += 32) {
    const int oy = op / OW, ox = op - oy * OW;
    const f32x4 dz = dzs[op * 8 + q];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int y = oy * stride - pad + kh;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int x = ox * stride - pad + kw;
        if (x < 0 || x >= W) continue;
        const f32x4 a = act[(y * W + x) * 8 + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[kh * 3 + kw][j] += dz[j] * a[j];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(pl * 8 + q) * 36 + j * 9 + t] = acc[t][j];
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 36; i += 256) {   // (quad, channel-in-quad * 9 + tap)
    const int qq = i / 36, e = i - qq * 36;
    float a0 = 0.f;
    for (int r = 0; r < 32; ++r) a0 += red[(r * 8 + qq) * 36 + e];
    atomicAdd(&dw[(size_t)(cb + qq * 4) * 9 + e], a0);
  }
}

// squeeze: pool[n][c] = mean_hw silu(z*s+b)
// Workgroup = 16 channel quads (64 channels) x 16 pixel lanes of ONE image: every pixel row of the chunk is a
// 256-byte segment, the HW pixels are spread over the 16 lanes and reduced through LDS (one thread per image and
// channel quad walking all HW pixels serially left the chip at 80 waves and 46 us per call: round-2 profile).
__global__ __launch_bounds__(256) void se_pool_kernel(const float* __restrict__ z, const float* __restrict__ s,
                                                      const float* __restrict__ b, const BnFold f, float* __restrict__ pool, int HW,
                                                      int C) {
  __shared__ f32x4 red[16][17];
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int n = blockIdx.y, c = blockIdx.x * 64 + q * 4;
  f32x4 acc = {0, 0, 0, 0};
  if (f.stat && f.publish && f.nbt && blockIdx.x == 0 && n == 0 && threadIdx.x == 0) *f.nbt += f.reps;
  __shared__ __attribute__((aligned(16))) float cf[2][64];   // (folded: the image-0 workgroup of a channel block publishes)
  bn_coef_block_fwd<64>(s, b, f, C, blockIdx.x * 64, f.publish && n == 0, cf);
  if (c < C) {
    const f32x4 sv = *reinterpret_cast<const f32x4*>(&cf[0][q * 4]), bv = *reinterpret_cast<const f32x4*>(&cf[1][q * 4]);
    for (int p = pl; p < HW; p += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(z + ((size_t)n * HW + p) * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += silu_f(v[j] * sv[j] + bv[j]);
    }
  }
  red[pl][q] = acc;
  __syncthreads();
  if (pl == 0 && c < C) {
#pragma unroll
    for (int r = 1; r < 16; ++r) acc += red[r][q];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] /= (float)HW;
    *reinterpret_cast<f32x4*>(pool + (size_t)n * C + c) = acc;
  }
}

// dgate[n][c] = sum_hw t[pix,c] * silu(z*s+b)[pix,c]   (t = gradient wrt the gated activation); same shape
__global__ __launch_bounds__(256) void se_dgate_kernel(const float* __restrict__ t, const float* __restrict__ z,
                                                       const float* __restrict__ s, const float* __restrict__ b,
                                                       float* __restrict__ dgate, int HW, int C,
                                                       float* __restrict__ zero, int nzero) {
  __shared__ f32x4 red[16][17];
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int n = blockIdx.y, c = blockIdx.x * 64 + q * 4;
  if (zero && blockIdx.x == 0 && blockIdx.y == 0)     // clears the scratch the next kernel of the stream accumulates into
    for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0.f;
  f32x4 acc = {0, 0, 0, 0};
  if (c < C) {
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + c), bv = *reinterpret_cast<const f32x4*>(b + c);
    for (int p = pl; p < HW; p += 16) {
      const size_t o = ((size_t)n * HW + p) * C + c;
      const f32x4 v = *reinterpret_cast<const f32x4*>(z + o), tv = *reinterpret_cast<const f32x4*>(t + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += tv[j] * silu_f(v[j] * sv[j] + bv[j]);
    }
  }
  red[pl][q] = acc;
  __syncthreads();
  if (pl == 0 && c < C) {
#pragma unroll
    for (int r = 1; r < 16; ++r) acc += red[r][q];
    *reinterpret_cast<f32x4*>(dgate + (size_t)n * C + c) = acc;
  }
}

// du = (t * gate[n][c] + add[n][c] / HW) * act'(z*s+b) -> out ; BatchNorm-backward sums of bn(z)
// (gate / add nullable).  Same workgroup shape as the depthwise kernels.
__global__ __launch_bounds__(256) void act_bwd_stats_kernel(
    const float* __restrict__ t, const float* __restrict__ gate, const float* __restrict__ add,
    const float* __restrict__ z, const float* __restrict__ s, const float* __restrict__ b,
    const float* __restrict__ mean, const float* __restrict__ invstd, int act, float* __restrict__ out,
    double* __restrict__ stat, long npix, int HW, int C) {
  __shared__ double red[256 * 8];
  const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.y * 32 + q * 4;
  const bool cv = c < C;              // C is a multiple of 4, not necessarily of 32 (24-channel stem/stage 0)
  const int cc = cv ? c : 0;
  f32x4 sv = *reinterpret_cast<const f32x4*>(s + cc), bv = *reinterpret_cast<const f32x4*>(b + cc);
  f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cc), is = *reinterpret_cast<const f32x4*>(invstd + cc);
  const float inv_hw = 1.0f / (float)HW;
  double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
  if (cv) {
    for (long pix = (long)blockIdx.x * 32 + pl; pix < npix; pix += (long)gridDim.x * 32) {
      const long n = pix / HW;
      f32x4 tv = *reinterpret_cast<const f32x4*>(t + (size_t)pix * C + c);
      f32x4 zv = *reinterpret_cast<const f32x4*>(z + (size_t)pix * C + c), o;
      if (gate) {
        f32x4 gv = *reinterpret_cast<const f32x4*>(gate + (size_t)n * C + c), av = *reinterpret_cast<const f32x4*>(add + (size_t)n * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) tv[j] = tv[j] * gv[j] + av[j] * inv_hw;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = tv[j] * act_bwd(act, zv[j] * sv[j] + bv[j]);
        sa[j] += (double)o[j];
        sb[j] += (double)(o[j] * ((zv[j] - mu[j]) * is[j]));
      }
      *reinterpret_cast<f32x4*>(out + (size_t)pix * C + c) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = sa[j]; red[threadIdx.x * 8 + 4 + j] = sb[j]; }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int qq = threadIdx.x >> 2, j = threadIdx.x & 3;
    const int ch = blockIdx.y * 32 + qq * 4 + j;
    if (ch < C) {
      double a0 = 0, a1 = 0;
      for (int r = 0; r < 32; ++r) { a0 += red[(r * 8 + qq) * 8 + j]; a1 += red[(r * 8 + qq) * 8 + 4 + j]; }
      const int slot = blockIdx.x & (MMVQA_STAT_SLOTS - 1);
      double* d = stat + ((size_t)slot * C + ch) * 2;
      atomicAdd(d, a0);
      atomicAdd(d + 1, a1);
    }
  }
}

// y = x * act'(pre)   (squeeze-excite gate backward on [B, C] tensors)
__global__ void mul_dact_kernel(const float* __restrict__ x, const float* __restrict__ pre, int act,
                                float* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * act_bwd(act, pre[i]);
}

static inline int pix_grid(long npix, int cgroups) {
  long g = (npix + 31) / 32;
  long cap = 4096 / (cgroups > 0 ? cgroups : 1);
  if (cap < 8) cap = 8;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

int k_bn_act_add(hipStream_t st, const float* z, const float* s, const float* b, int pre_act, const float* idn,
                 const float* ids, const float* idb, int idn_act, int post_act, float* out, long rows, int C) {
  long n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_act_add_kernel, dim3(grid_for(n4)), dim3(256), 0, st, z, s, b, pre_act, idn, ids, idb,
                     idn_act, post_act, out, n4, C / 4);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

static bool dw_pixel_form() {   // A/B switch: the pixel-strided kernels even where a map fits in LDS
  static const bool v = getenv("MMVQA_DW_PIXEL") != nullptr;
  return v;
}

int k_dwconv_fwd(hipStream_t st, const float* z1, const float* s1, const float* b1, const float* w, float* z2,
                 double* stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold) {
  if (C % 32) return mmvqa_set_error(MMVQA_ERR_ARG, "dwconv: C=%d must be a multiple of 32", C);
  if (fold && (!fold->stat || fold->bwd)) return mmvqa_set_error(MMVQA_ERR_ARG, "dwconv: the fold must be a forward one");
  mmvqa_bn_fold f1;
  if (fold) f1 = *fold; else memset(&f1, 0, sizeof(f1));
  const size_t sm = (size_t)H * W * 128 + 256 * 8 * sizeof(double);
  if (sm <= DW_TILE_MAX_BYTES && !dw_pixel_form()) {
    // (every launch: the attribute is per device and the call is cheap -- a once-per-process flag left a second device without it)
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)dwconv_fwd_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DW_TILE_MAX_BYTES));
    hipLaunchKernelGGL(dwconv_fwd_tile_kernel, dim3(C / 32, N), dim3(256), sm, st, z1, s1, b1, f1, w, z2, stat, H, W, C, OH, OW,
                       stride, pad);
    KERNEL_CHECK_RET();
    return MMVQA_OK;
  }
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(pix_grid((long)N * OH * OW, C / 32), C / 32), dim3(256), 0, st, z1, s1, b1, f1,
                     w, z2, stat, N, H, W, C, OH, OW, stride, pad);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_dwconv_bwd_data(hipStream_t st, const float* g2, const float* z2, const float* P, const float* Q,
                      const float* R, const float* w, const float* z1, const float* s1, const float* b1,
                      const float* mean1, const float* invstd1, float* g1, double* stat, int N, int H, int W, int C,
                      int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold) {
  if (C % 32) return mmvqa_set_error(MMVQA_ERR_ARG, "dwconv: C=%d must be a multiple of 32", C);
  if (fold && (!fold->stat || !fold->bwd)) return mmvqa_set_error(MMVQA_ERR_ARG, "dwconv_bwd_data: the fold must be a backward one");
  mmvqa_bn_fold f2;
  if (fold) f2 = *fold; else memset(&f2, 0, sizeof(f2));
  const size_t sm = (size_t)OH * OW * 128 + 256 * 8 * sizeof(double);
  if (sm <= DW_TILE_MAX_BYTES && !dw_pixel_form()) {
    // (every launch: the attribute is per device and the call is cheap -- a once-per-process flag left a second device without it)
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)dwconv_bwd_data_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DW_TILE_MAX_BYTES));
    hipLaunchKernelGGL(dwconv_bwd_data_tile_kernel, dim3(C / 32, N), dim3(256), sm, st, g2, z2, P, Q, R, w, z1, s1, b1,
                       mean1, invstd1, g1, stat, H, W, C, OH, OW, stride, pad, f2);
    KERNEL_CHECK_RET();
    return MMVQA_OK;
  }
  hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(pix_grid((long)N * H * W, C / 32), C / 32), dim3(256), 0, st, g2, z2,
                     P, Q, R, w, z1, s1, b1, mean1, invstd1, g1, stat, N, H, W, C, OH, OW, stride, pad, f2);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_dwconv_bwd_weight(hipStream_t st, const float* g2, const float* z2, const float* P, const float* Q,
                        const float* R, const float* z1, const float* s1, const float* b1, float* dw, int N, int H,
                        int W, int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold) {
  if (C % 32) return mmvqa_set_error(MMVQA_ERR_ARG, "dwconv: C=%d must be a multiple of 32", C);
  if (fold && (!fold->stat || !fold->bwd)) return mmvqa_set_error(MMVQA_ERR_ARG, "dwconv_bwd_weight: the fold must be a backward one");
  mmvqa_bn_fold f2;
  if (fold) f2 = *fold; else memset(&f2, 0, sizeof(f2));
  const size_t sm = ((size_t)H * W + (size_t)OH * OW) * 128 + 32 * 8 * 36 * sizeof(float);
  if (sm <= DW_TILE_MAX_BYTES && !dw_pixel_form()) {
    // (every launch: the attribute is per device and the call is cheap -- a once-per-process flag left a second device without it)
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)dwconv_bwd_weight_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DW_TILE_MAX_BYTES));
    hipLaunchKernelGGL(dwconv_bwd_weight_tile_kernel, dim3(C / 32, N), dim3(256), sm, st, g2, z2, P, Q, R, z1, s1, b1, dw, H, W,
                       C, OH, OW, stride, pad, f2);
    KERNEL_CHECK_RET();
    return MMVQA_OK;
  }
  int g = pix_grid((long)N * OH * OW, C / 32);
  if (g > 64) g = 64;   // every workgroup ends with 288 atomics per 32 channels
  hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(g, C / 32), dim3(256), 0, st, g2, z2, P, Q, R, z1, s1, b1, dw, N,
                     H, W, C, OH, OW, stride, pad, f2);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_se_pool(hipStream_t st, const float* z, const float* s, const float* b, float* pool, int N, int HW, int C,
              const mmvqa_bn_fold* fold) {
  if (fold && (!fold->stat || fold->bwd || C % 4)) return mmvqa_set_error(MMVQA_ERR_ARG, "se_pool: the fold must be a forward one and C a multiple of 4");
  mmvqa_bn_fold f;
  if (fold) f = *fold; else memset(&f, 0, sizeof(f));
  hipLaunchKernelGGL(se_pool_kernel, dim3(cdiv_i(C, 64), N), dim3(256), 0, st, z, s, b, f, pool, HW, C);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_se_dgate(hipStream_t st, const float* t, const float* z, const float* s, const float* b, float* dgate, int N,
               int HW, int C, float* zero, int nzero) {
  hipLaunchKernelGGL(se_dgate_kernel, dim3(cdiv_i(C, 64), N), dim3(256), 0, st, t, z, s, b, dgate, HW, C, zero, nzero);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_act_bwd_stats(hipStream_t st, const float* t, const float* gate, const float* add, const float* z,
                    const float* s, const float* b, const float* mean, const float* invstd, int act, float* out,
                    double* stat, long npix, int HW, int C) {
  if (C % 4) return mmvqa_set_error(MMVQA_ERR_ARG, "act_bwd_stats: C=%d must be a multiple of 4", C);
  const int cg = (C + 31) / 32;
  hipLaunchKernelGGL(act_bwd_stats_kernel, dim3(pix_grid(npix, cg), cg), dim3(256), 0, st, t, gate, add, z, s, b,
                     mean, invstd, act, out, stat, npix, HW, C);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int k_mul_dact(hipStream_t st, const float* x, const float* pre, int act, float* y, long n) {
  hipLaunchKernelGGL(mul_dact_kernel, dim3(cdiv_i(n, 256)), dim3(256), 0, st, x, pre, act, y, n);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}
