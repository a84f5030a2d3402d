// Squeeze-excite gate of EfficientNetV2's MBConv blocks (timm SqueezeExcite under
// models/image_encoding.py:89-115): the two fully connected layers on the pooled [B, mid] tensor, forward and backward.
//
//   r    = silu(pool W_r^T + b_r)          [B, rd]      W_r [rd, mid]
//   gate = sigmoid(r W_e^T + b_e)          [B, mid]     W_e [mid, rd]
//
// B is the per-GPU batch (16..64): as GEMMs these are 16-row problems with one or two 64x64 output tiles and up to 96
// K-tiles walked by a single workgroup (round-2 profile of config 3: 264 such launches, 4.3 ms of a 29 ms step, the
// longest 60 us each).  Here the work is spread over the OTHER axis: one wave (or workgroup) per output feature in the
// forward pass, one workgroup per 64 channels of `mid` in the two backward kernels, which also produce the weight and
// bias gradients in the same pass (every weight / bias gradient has one owner; only the [B, rd] contraction over `mid`
// crosses workgroups, with fp32 atomics).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SE_MT = 16;    // batch rows handled per pass
constexpr int SE_CH = 64;    // channels of `mid` per workgroup in the backward kernels
constexpr int SE_MAXRD = 128;
inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------------- forward
// y[m][n] = act(sum_k x[m][k] W[n][k] + b[n]); pre[m][n] keeps the pre-activation (nullable).
// WPN waves share one output feature n (its K range is split over them); 4 / WPN features per workgroup.
template <int WPN>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, int x_ld,
                                                         const float* __restrict__ W, const float* __restrict__ b,
                                                         int act, float* __restrict__ pre, float* __restrict__ y,
                                                         int M, int N, int K) {
  constexpr int NPW = 4 / WPN;                 // features per workgroup
  constexpr int LPN = WPN * 64;                // lanes per feature
  __shared__ float red[NPW][SE_MT][LPN + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nl = wave / WPN, kpart = wave % WPN;
  const int n = blockIdx.x * NPW + nl;
  const int m0 = blockIdx.y * SE_MT;
  const int u = kpart * 64 + lane;             // lane index inside the feature's group
  float acc[SE_MT];
#pragma unroll
  for (int m = 0; m < SE_MT; ++m) acc[m] = 0.f;
  if (n < N) {
    const float* w = W + (size_t)n * K;
    // rows past M repeat row M-1 (discarded at the end): no branch sits between the loads of one k step
    const float* xr[SE_MT];
#pragma unroll
    for (int m = 0; m < SE_MT; ++m) xr[m] = x + (size_t)(m0 + m < M ? m0 + m : M - 1) * x_ld;
    if ((K & 3) == 0 && (x_ld & 3) == 0) {
#pragma unroll 2
      for (int k = u * 4; k < K; k += LPN * 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
        f32x4 xv[SE_MT];
#pragma unroll
        for (int m = 0; m < SE_MT; ++m) xv[m] = *reinterpret_cast<const f32x4*>(xr[m] + k);
#pragma unroll
        for (int m = 0; m < SE_MT; ++m)
          acc[m] += wv[0] * xv[m][0] + wv[1] * xv[m][1] + wv[2] * xv[m][2] + wv[3] * xv[m][3];
      }
    } else {
#pragma unroll 2
      for (int k = u; k < K; k += LPN) {
        const float wv = w[k];
        float xv[SE_MT];
#pragma unroll
        for (int m = 0; m < SE_MT; ++m) xv[m] = xr[m][k];
#pragma unroll
        for (int m = 0; m < SE_MT; ++m) acc[m] += wv * xv[m];
      }
    }
  }
#pragma unroll
  for (int m = 0; m < SE_MT; ++m) red[nl][m][u] = acc[m];
  __syncthreads();
  // LPN lanes of the group: 16 rows x (LPN / 16) partial sums of 16 values each, then a shuffle tree
  constexpr int PARTS = LPN / 16;              // 4 or 16 consecutive lanes per row
  const int m = u / PARTS, part = u % PARTS;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[nl][m][part * 16 + i];
#pragma unroll
  for (int o = PARTS / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (part == 0 && n < N && m0 + m < M) {
    s += b ? b[n] : 0.f;
    if (pre) pre[(size_t)(m0 + m) * N + n] = s;
    y[(size_t)(m0 + m) * N + n] = act_fwd(act, s);
  }
}

// ------------------------------------------------------------------------------------------------------- backward
// Both backward kernels work on 16 batch rows at a time with every operand of the pass in LDS (rows padded to a
// multiple of four floats, so that the 4x4 / 1x4 register tiles below read them with ds_read_b128).
constexpr int SE_RS = SE_MAXRD + 4;   // row stride of [.][rd] tiles
constexpr int SE_CS = SE_CH + 4;      // row stride of [.][64] tiles

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// acc[a][b] += sum_m P[m][pa + a] * Q[m][qb + b]  (4x4 outer-product tile over the 16 rows of the pass)
__device__ __forceinline__ void outer16(const float* P, int p_ld, int pa, const float* Q, int q_ld, int qb,
                                        float (&acc)[4][4]) {
#pragma unroll
  for (int m = 0; m < SE_MT; ++m) {
    const f32x4 pv = lds4(P + m * p_ld + pa), qv = lds4(Q + m * q_ld + qb);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += pv[a] * qv[b];
  }
}

// One workgroup = 64 channels n of `mid` (rows of W_e):
//   dgpre[m][n] = dgate[m][n] * sigmoid'(gpre[m][n])
//   dW_e[n][j] += sum_m dgpre[m][n] r[m][j] ;  db_e[n] += sum_m dgpre[m][n]
//   drraw[m][j] += sum_{n in chunk} dgpre[m][n] W_e[n][j]              (atomics into the zeroed [B, rd] scratch)
template <bool VEC>   // VEC: rd and mid are multiples of 4 -> every global access of the pass is a 16-byte one, all issued up front
__global__ __launch_bounds__(256) void se_bwd_a_kernel(const float* __restrict__ dgate, const float* __restrict__ gpre,
                                                       const float* __restrict__ r, const float* __restrict__ We,
                                                       float* __restrict__ dWe, float* __restrict__ dbe,
                                                       float* __restrict__ drraw, int B, int mid, int rd) {
  __shared__ __attribute__((aligned(16))) float We_s[SE_CH * SE_RS];    // [n][j]
  __shared__ __attribute__((aligned(16))) float r_s[SE_MT * SE_RS];     // [m][j]
  __shared__ __attribute__((aligned(16))) float dg_s[SE_MT * SE_CS];    // [m][n]
  __shared__ __attribute__((aligned(16))) float dgT_s[SE_CH * SE_MT];   // [n][m]
  const int tid = threadIdx.x, n0 = blockIdx.x * SE_CH;
  const int rd4 = (rd + 3) & ~3;
  constexpr int NWQ = SE_CH * SE_MAXRD / 4 / 256;   // 8 quads of the W_e tile per thread at most
  f32x4 wq[NWQ];
  if constexpr (VEC) {   // 64 rows of W_e are one contiguous run of 64*rd floats
    const float* src = We + (size_t)n0 * rd;
    const int nv = (mid - n0 < SE_CH ? mid - n0 : SE_CH) * rd;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = (tid + q * 256) * 4;
      wq[q] = (e < nv) ? *reinterpret_cast<const f32x4*>(src + e) : f32x4{0, 0, 0, 0};
    }
  } else {
    const float* src = We + (size_t)n0 * rd;
    const int nv = (mid - n0 < SE_CH ? mid - n0 : SE_CH) * rd;
#pragma unroll 8
    for (int i = tid; i < SE_CH * rd; i += 256) {
      const float v = i < nv ? src[i] : 0.f;
      const int nl = i / rd;
      We_s[nl * SE_RS + i - nl * rd] = v;
    }
    if (rd4 != rd)
      for (int i = tid; i < SE_CH * 4; i += 256) We_s[(i >> 2) * SE_RS + rd + (i & 3)] = 0.f;   // (rd + 3 < SE_RS)
  }
  // weight-gradient tile of this thread: channels ng..ng+3, features jh + 64 q .. +3
  const int ng = (tid & 15) * 4, jh = (tid >> 4) * 4;
  float dw[2][4][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dw[q][a][b] = 0.f;
  float db = 0.f;
  for (int m0 = 0; m0 < B; m0 += SE_MT) {
    if constexpr (VEC) {
      const int m = tid >> 4, nl = (tid & 15) * 4;
      const bool ok = m0 + m < B && n0 + nl < mid;
      const size_t o = ok ? (size_t)(m0 + m) * mid + n0 + nl : 0;
      f32x4 dv = *reinterpret_cast<const f32x4*>(dgate + o);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gpre + o);
      const float* rsrc = r + (size_t)m0 * rd;
      const int nvr = (B - m0 < SE_MT ? B - m0 : SE_MT) * rd;
      f32x4 rq[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (tid + q * 256) * 4;
        rq[q] = (e < nvr) ? *reinterpret_cast<const f32x4*>(rsrc + e) : f32x4{0, 0, 0, 0};
      }
      if (m0 > 0) __syncthreads();
      if (m0 == 0) {
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
          const int e = (tid + q * 256) * 4;
          if (e < SE_CH * rd) {
            const int wn = e / rd;
            *reinterpret_cast<f32x4*>(We_s + wn * SE_RS + e - wn * rd) = wq[q];
          }
        }
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const float v = ok ? dv[y] * act_bwd(ACT_SIGMOID, gv[y]) : 0.f;
        dv[y] = v;
        dgT_s[(nl + y) * SE_MT + m] = v;
      }
      *reinterpret_cast<f32x4*>(dg_s + m * SE_CS + nl) = dv;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (tid + q * 256) * 4;
        if (e < SE_MT * rd) {
          const int rm = e / rd;
          *reinterpret_cast<f32x4*>(r_s + rm * SE_RS + e - rm * rd) = rq[q];
        }
      }
    } else {
      __syncthreads();
      float dv[SE_MT * SE_CH / 256], gv[SE_MT * SE_CH / 256];
#pragma unroll
      for (int q = 0; q < SE_MT * SE_CH / 256; ++q) {
        const int i = tid + q * 256, m = i / SE_CH, nl = i - m * SE_CH;
        const bool ok = m0 + m < B && n0 + nl < mid;
        const size_t o = ok ? (size_t)(m0 + m) * mid + n0 + nl : 0;
        dv[q] = dgate[o]; gv[q] = gpre[o];
        if (!ok) dv[q] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < SE_MT * SE_CH / 256; ++q) {
        const int i = tid + q * 256, m = i / SE_CH, nl = i - m * SE_CH;
        const float v = dv[q] * act_bwd(ACT_SIGMOID, gv[q]);
        dg_s[m * SE_CS + nl] = v;
        dgT_s[nl * SE_MT + m] = v;
      }
      const float* src = r + (size_t)m0 * rd;
      const int nv = (B - m0 < SE_MT ? B - m0 : SE_MT) * rd;
#pragma unroll 8
      for (int i = tid; i < SE_MT * rd; i += 256) {
        const float v = i < nv ? src[i] : 0.f;
        const int m = i / rd;
        r_s[m * SE_RS + i - m * rd] = v;
      }
      if (rd4 != rd)
        for (int i = tid; i < SE_MT * 4; i += 256) r_s[(i >> 2) * SE_RS + rd + (i & 3)] = 0.f;
    }
    __syncthreads();
    if (jh < rd4) outer16(dg_s, SE_CS, ng, r_s, SE_RS, jh, dw[0]);
    if (jh + 64 < rd4) outer16(dg_s, SE_CS, ng, r_s, SE_RS, jh + 64, dw[1]);
    if (tid < SE_CH) {
#pragma unroll
      for (int q = 0; q < SE_MT / 4; ++q) {
        const f32x4 v = lds4(dgT_s + tid * SE_MT + q * 4);
        db += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
    // drraw tile: rows mq..mq+3, features jq..jq+3 (4 x rd4/4 tiles)
    for (int t = tid; t < 4 * (rd4 >> 2); t += 256) {
      const int mq = (t & 3) * 4, jq = (t >> 2) * 4;
      float a[4][4];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) a[x][y] = 0.f;
#pragma unroll 4
      for (int nl = 0; nl < SE_CH; ++nl) {
        const f32x4 dv = lds4(dgT_s + nl * SE_MT + mq), wv = lds4(We_s + nl * SE_RS + jq);
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) a[x][y] += dv[x] * wv[y];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          if (m0 + mq + x < B && jq + y < rd) atomicAdd(&drraw[(size_t)(m0 + mq + x) * rd + jq + y], a[x][y]);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n = n0 + ng + a, jj = jh + 64 * q + b;
        if (n < mid && jj < rd) atomicAdd(&dWe[(size_t)n * rd + jj], dw[q][a][b]);   // one owner: no-return add, no round trip
      }
  if (tid < SE_CH && n0 + tid < mid) atomicAdd(&dbe[n0 + tid], db);
}

// One workgroup = 64 channels k of `mid` (columns of W_r):
//   drpre[m][j] = drraw[m][j] * silu'(rpre[m][j])                          (every workgroup, 16 x rd values)
//   db_r[j] += sum_m drpre[m][j]                                            (workgroup 0)
//   dW_r[j][k] += sum_m drpre[m][j] pool[m][k] ;  dpool[m][k] = sum_j drpre[m][j] W_r[j][k]
template <bool VEC>
__global__ __launch_bounds__(256) void se_bwd_b_kernel(const float* __restrict__ drraw,
                                                       const float* __restrict__ rpre, const float* __restrict__ pool,
                                                       const float* __restrict__ Wr, float* __restrict__ dWr,
                                                       float* __restrict__ dbr, float* __restrict__ dpool, int B,
                                                       int mid, int rd) {
  __shared__ __attribute__((aligned(16))) float Wr_s[SE_MAXRD * SE_CS];   // [j][k]
  __shared__ __attribute__((aligned(16))) float dr_s[SE_MT * SE_RS];      // [m][j]
  __shared__ __attribute__((aligned(16))) float pl_s[SE_MT * SE_CS];      // [m][k]
  const int tid = threadIdx.x, k0 = blockIdx.x * SE_CH;
  const int rd4 = (rd + 3) & ~3;
  constexpr int NWQ = SE_CH * SE_MAXRD / 4 / 256;
  f32x4 wq[NWQ];
  if constexpr (VEC) {
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int i4 = tid + q * 256, j = i4 >> 4, kl = (i4 & 15) * 4;
      wq[q] = (j < rd && k0 + kl < mid) ? *reinterpret_cast<const f32x4*>(Wr + (size_t)j * mid + k0 + kl)
                                        : f32x4{0, 0, 0, 0};
    }
  } else {
#pragma unroll 8
    for (int i = tid; i < rd * SE_CH; i += 256) {
      const int j = i / SE_CH, kl = i - j * SE_CH;
      Wr_s[j * SE_CS + kl] = (k0 + kl < mid) ? Wr[(size_t)j * mid + k0 + kl] : 0.f;
    }
  }
  // weight-gradient tile of this thread: features jh + 64 q .. +3, channels kg..kg+3
  const int kg = (tid & 15) * 4, jh = (tid >> 4) * 4;
  float dw[2][4][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dw[q][a][b] = 0.f;
  float db = 0.f;
  for (int m0 = 0; m0 < B; m0 += SE_MT) {
    if constexpr (VEC) {
      const size_t o0 = (size_t)m0 * rd;
      const int nv = (B - m0 < SE_MT ? B - m0 : SE_MT) * rd;
      f32x4 aq[2], pq[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (tid + q * 256) * 4;
        const bool ok = e < nv;
        aq[q] = *reinterpret_cast<const f32x4*>(drraw + (ok ? o0 + e : 0));
        pq[q] = *reinterpret_cast<const f32x4*>(rpre + (ok ? o0 + e : 0));
      }
      const int m = tid >> 4, kl = (tid & 15) * 4;
      const f32x4 pv = (m0 + m < B && k0 + kl < mid) ? *reinterpret_cast<const f32x4*>(pool + (size_t)(m0 + m) * mid + k0 + kl)
                                                     : f32x4{0, 0, 0, 0};
      if (m0 > 0) __syncthreads();
      if (m0 == 0) {
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
          const int i4 = tid + q * 256, j = i4 >> 4, wk = (i4 & 15) * 4;
          if (j < rd) *reinterpret_cast<f32x4*>(Wr_s + j * SE_CS + wk) = wq[q];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (tid + q * 256) * 4;
        if (e < SE_MT * rd) {
          f32x4 v;
#pragma unroll
          for (int y = 0; y < 4; ++y) v[y] = (e < nv) ? aq[q][y] * act_bwd(ACT_SILU, pq[q][y]) : 0.f;
          const int rm = e / rd;
          *reinterpret_cast<f32x4*>(dr_s + rm * SE_RS + e - rm * rd) = v;
        }
      }
      *reinterpret_cast<f32x4*>(pl_s + m * SE_CS + kl) = pv;
    } else {
      __syncthreads();
      const size_t o0 = (size_t)m0 * rd;
      const int nv = (B - m0 < SE_MT ? B - m0 : SE_MT) * rd;
#pragma unroll 8
      for (int i = tid; i < SE_MT * rd; i += 256) {
        const bool ok = i < nv;
        const float a = drraw[ok ? o0 + i : 0], pr = rpre[ok ? o0 + i : 0];
        const int m = i / rd;
        dr_s[m * SE_RS + i - m * rd] = ok ? a * act_bwd(ACT_SILU, pr) : 0.f;
      }
      if (rd4 != rd)
        for (int i = tid; i < SE_MT * 4; i += 256) dr_s[(i >> 2) * SE_RS + rd + (i & 3)] = 0.f;
#pragma unroll
      for (int q = 0; q < SE_MT * SE_CH / 256; ++q) {
        const int i = tid + q * 256, m = i / SE_CH, kl = i - m * SE_CH;
        pl_s[m * SE_CS + kl] = (m0 + m < B && k0 + kl < mid) ? pool[(size_t)(m0 + m) * mid + k0 + kl] : 0.f;
      }
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < rd) {
#pragma unroll
      for (int m = 0; m < SE_MT; ++m) db += dr_s[m * SE_RS + tid];
    }
    if (jh < rd4) outer16(dr_s, SE_RS, jh, pl_s, SE_CS, kg, dw[0]);
    if (jh + 64 < rd4) outer16(dr_s, SE_RS, jh + 64, pl_s, SE_CS, kg, dw[1]);
    {   // dpool: row m, channels kq..kq+3
      const int m = tid >> 4, kq = (tid & 15) * 4;
      f32x4 a = {0, 0, 0, 0};
#pragma unroll 4
      for (int j = 0; j < rd; ++j) {
        const float dv = dr_s[m * SE_RS + j];
        const f32x4 wv = lds4(Wr_s + j * SE_CS + kq);
        a += dv * wv;
      }
#pragma unroll
      for (int y = 0; y < 4; ++y)
        if (m0 + m < B && k0 + kq + y < mid) dpool[(size_t)(m0 + m) * mid + k0 + kq + y] = a[y];
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int jj = jh + 64 * q + a, k = k0 + kg + b;
        if (jj < rd && k < mid) atomicAdd(&dWr[(size_t)jj * mid + k], dw[q][a][b]);
      }
  if (blockIdx.x == 0 && tid < rd) atomicAdd(&dbr[tid], db);
}

}  // namespace

int k_skinny_fwd(hipStream_t st, const float* x, int x_ld, const float* W, const float* b, int act, float* pre,
                 float* y, int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return mmvqa_set_error(MMVQA_ERR_ARG, "skinny_fwd: M=%d N=%d K=%d", M, N, K);
  const int my = cdiv_i(M, SE_MT);
  if (K >= 512)
    hipLaunchKernelGGL(skinny_fwd_kernel<4>, dim3(N, my), dim3(256), 0, st, x, x_ld, W, b, act, pre, y, M, N, K);
  else
    hipLaunchKernelGGL(skinny_fwd_kernel<1>, dim3(cdiv_i(N, 4), my), dim3(256), 0, st, x, x_ld, W, b, act, pre, y, M,
                       N, K);
  HIP_CHECK_RET(hipGetLastError());
  return MMVQA_OK;
}

bool k_se_fc_bwd_ok(int rd) { return rd > 0 && rd <= SE_MAXRD; }

size_t k_se_fc_bwd_scratch_floats(int B, int mid, int rd) { (void)mid; return (size_t)B * rd; }

// gradients of both squeeze-excite layers: accumulates dW_e, db_e, dW_r, db_r; writes dpool[B, mid].
// scratch: B*rd floats, zero on entry (scratch_is_zero: the caller's previous kernel cleared it; else a memset is queued)
int k_se_fc_bwd(hipStream_t st, const float* dgate, const float* gpre, const float* r, const float* rpre,
                const float* pool, const float* We, const float* Wr, float* dWe, float* dbe, float* dWr, float* dbr,
                float* dpool, float* scratch, int scratch_is_zero, int B, int mid, int rd) {
  if (rd > SE_MAXRD || rd <= 0 || B <= 0 || mid <= 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "se_fc_bwd: rd=%d (max %d) B=%d mid=%d", rd, SE_MAXRD, B, mid);
  const int nch = cdiv_i(mid, SE_CH);
  if (!scratch_is_zero) HIP_CHECK_RET(hipMemsetAsync(scratch, 0, sizeof(float) * (size_t)B * rd, st));
  if ((rd & 3) == 0 && (mid & 3) == 0) {
    hipLaunchKernelGGL(se_bwd_a_kernel<true>, dim3(nch), dim3(256), 0, st, dgate, gpre, r, We, dWe, dbe, scratch, B, mid, rd);
    hipLaunchKernelGGL(se_bwd_b_kernel<true>, dim3(nch), dim3(256), 0, st, scratch, rpre, pool, Wr, dWr, dbr, dpool, B,
                       mid, rd);
  } else {
    hipLaunchKernelGGL(se_bwd_a_kernel<false>, dim3(nch), dim3(256), 0, st, dgate, gpre, r, We, dWe, dbe, scratch, B, mid, rd);
    hipLaunchKernelGGL(se_bwd_b_kernel<false>, dim3(nch), dim3(256), 0, st, scratch, rpre, pool, Wr, dWr, dbr, dpool, B,
                       mid, rd);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMVQA_OK;
}
