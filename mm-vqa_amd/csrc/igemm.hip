// Implicit-GEMM kernel family on the gfx950 fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// One template covers the three contractions of a convolution / linear layer in NHWC:
//   KIND_FWD   C[pix, co]      = sum_{tap,ci} X[pix@tap, ci] * W[co, tap, ci]
//   KIND_DGRAD C[pix_in, ci]   = sum_{tap,co} dZ[pix_out(pix_in,tap), co] * W[co, tap, ci]
//   KIND_WGRAD C[co, (tap,ci)] = sum_{pix} dZ[pix, co] * X[pix@tap, ci]
// (models/image_encoding.py:53-86 convs via torchvision, models/transformer.py:13-15,45-48
//  and models/mmbert.py:133-137 linears are all instances.)
//
// Tiling: 256 threads = 4 waves (2x2); workgroup tile BMxBN, wave tile (BM/2)x(BN/2) made of
// 32x32 MFMA tiles; BK = 32, LDS double-buffered, one barrier per K-tile.  Operands whose
// contraction index is contiguous in memory sit in LDS as [row][k] (stride 36 floats, read
// as ds_read_b128: conflict-free, see MI355X_MICROARCH LDS table); operands whose row index
// is contiguous sit as [k][row] and are read with ds_read_b32.  Within an 8-deep k-group lane
// half h feeds k = 4h+j at MFMA step j for BOTH operands, so the permuted k order is consistent.
//
// Prologues (BatchNorm apply + ReLU, BatchNorm backward) run when a tile is written to LDS;
// the epilogue fuses bias / activation / dropout / residual / ReLU-mask / per-channel
// statistics so that BatchNorm never needs its own pass over a feature map.
#include "common.h"

#define BK 32
#define LDK 36  // row stride (floats) of a [row][k] LDS tile

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ f32x4 apply_pro(int pro, f32x4 v, f32x4 v2, f32x4 c0, f32x4 c1, f32x4 c2) {
  f32x4 r;
  if (pro == PRO_AFFINE_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { float t = v[j] * c0[j] + c1[j]; r[j] = t > 0.f ? t : 0.f; }
  } else if (pro == PRO_DZ) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = v[j] * c0[j] + v2[j] * c1[j] + c2[j];
  } else if (pro == PRO_AFFINE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = v[j] * c0[j] + c1[j];
  } else {
    r = v;
  }
  return r;
}

template <int BM, int BN, int KIND, bool NCHW>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmParams p) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int NA = BM / 32, NB = BN / 32;  // float4 chunks per thread per K-tile
  constexpr bool A_ROWK = (KIND != KIND_WGRAD);
  constexpr bool B_ROWK = (KIND == KIND_FWD);
  constexpr int LDA_KM = BM + 4, LDB_KM = BN + 4;
  constexpr int A_TILE = A_ROWK ? BM * LDK : BK * LDA_KM;
  constexpr int B_TILE = B_ROWK ? BN * LDK : BK * LDB_KM;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][A_TILE]
  float* Bs = smem + 2 * A_TILE;    // [2][B_TILE]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // K range of this split
  const int nkt_total = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.z * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt_total) kt_end = nkt_total;
  const int nkt = kt_end - kt_begin;

  const int OHW = p.g_OH * p.g_OW;
  const int taps = p.g_KH * p.g_KW;

  // ------------------------------------------------------------------ loader state
  // A, row-major gather (FWD / DGRAD): per-thread rows (tid>>3)+32r, k-quad tid&7
  int a_pix[NA], a_y0[NA], a_x0[NA];
  // A, k-major plain (WGRAD): chunk (krow, x4)
  constexpr int A_X4 = BM / 4;   // float4 per k-row
  const int akm_x4 = tid % A_X4, akm_k0 = tid / A_X4;
  constexpr int A_KSTEP = 256 / A_X4;
  f32x4 ac0 = {1, 1, 1, 1}, ac1 = {0, 0, 0, 0}, ac2 = {0, 0, 0, 0};
  if constexpr (A_ROWK) {
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      int row = m0 + (tid >> 3) + 32 * r;
      if (row < p.M) {
        int n = row / OHW, rem = row - n * OHW;
        int oy = rem / p.g_OW, ox = rem - oy * p.g_OW;
        a_pix[r] = n * p.g_SH * p.g_SW;
        if (KIND == KIND_FWD) { a_y0[r] = oy * p.g_stride - p.g_pad; a_x0[r] = ox * p.g_stride - p.g_pad; }
        else { a_y0[r] = oy + p.g_pad; a_x0[r] = ox + p.g_pad; }
        if (NCHW) a_pix[r] = n;  // image index; channel planes are resolved per element
      } else {
        a_pix[r] = 0; a_y0[r] = -(1 << 24); a_x0[r] = -(1 << 24);
      }
    }
  } else {
    // channel = output row index i: loop invariant
    int i = m0 + akm_x4 * 4;
    if (p.a_pro != PRO_NONE && i < p.M) {
      ac0 = ld4(p.a_c0 + i); ac1 = ld4(p.a_c1 + i);
      if (p.a_pro == PRO_DZ) ac2 = ld4(p.a_c2 + i);
    }
  }

  // B loaders
  constexpr int B_X4 = BN / 4;
  const int bkm_x4 = tid % B_X4, bkm_k0 = tid / B_X4;
  constexpr int B_KSTEP = 256 / B_X4;
  f32x4 bc0 = {1, 1, 1, 1}, bc1 = {0, 0, 0, 0};
  int b_kh = 0, b_kw = 0, b_ci = 0;   // WGRAD gather: tap/channel of this thread's 4 columns
  int b_dy[4], b_dx[4], b_cc[4];      // NCHW WGRAD: per-element tap decode
  bool b_colvalid = true;
  if constexpr (KIND == KIND_WGRAD) {
    int nn = n0 + bkm_x4 * 4;
    b_colvalid = nn < p.N;
    if constexpr (!NCHW) {
      int tap = nn / p.g_Cs;
      b_ci = nn - tap * p.g_Cs;
      b_kh = tap / p.g_KW; b_kw = tap - b_kh * p.g_KW;
      if (p.b_pro != PRO_NONE && b_colvalid) { bc0 = ld4(p.b_c0 + b_ci); bc1 = ld4(p.b_c1 + b_ci); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kk = nn + j;
        int tap = kk / p.g_Cs;
        b_cc[j] = kk - tap * p.g_Cs;
        b_dy[j] = tap / p.g_KW; b_dx[j] = tap - b_dy[j] * p.g_KW;
        if (kk >= p.N) b_dy[j] = -(1 << 24);
      }
    }
  }

  f32x4 ra[NA], ra2[NA], rb[NB];
  unsigned a_ok = 0, b_ok = 0;  // bit r: chunk r holds real data (prologue applies)

  auto load_tile = [&](int kt) {
    const int kbase = kt * BK;
    // ---------------- A
    if constexpr (A_ROWK) {
      const int k = kbase + (tid & 7) * 4;
      a_ok = 0;
      if constexpr (!NCHW) {
        int tap = 0, c = k;
        if (taps > 1) { tap = k / p.g_Cs; c = k - tap * p.g_Cs; }
        int kh = 0, kw = tap;
        if (p.g_KW > 1 && taps > 1) { kh = tap / p.g_KW; kw = tap - kh * p.g_KW; }
        else if (taps > 1) { kh = tap; kw = 0; }
        const bool kvalid = k < p.K;
        if (p.a_pro != PRO_NONE && kvalid) {
          ac0 = ld4(p.a_c0 + c); ac1 = ld4(p.a_c1 + c);
          if (p.a_pro == PRO_DZ) ac2 = ld4(p.a_c2 + c);
        }
#pragma unroll
        for (int r = 0; r < NA; ++r) {
          int sy, sx; bool ok = kvalid;
          if (KIND == KIND_FWD) { sy = a_y0[r] + kh; sx = a_x0[r] + kw; }
          else {
            int ty = a_y0[r] - kh, tx = a_x0[r] - kw;
            if (p.g_stride == 1) { sy = ty; sx = tx; }
            else if (p.g_stride == 2) { ok = ok && (((ty | tx) & 1) == 0); sy = ty >> 1; sx = tx >> 1; }
            else { sy = ty / p.g_stride; sx = tx / p.g_stride;
                   ok = ok && (sy * p.g_stride == ty) && (sx * p.g_stride == tx); }
          }
          ok = ok && sy >= 0 && sy < p.g_SH && sx >= 0 && sx < p.g_SW;
          f32x4 v = {0, 0, 0, 0}, v2 = {0, 0, 0, 0};
          if (ok) {
            size_t off = (size_t)(a_pix[r] + sy * p.g_SW + sx) * p.a_ld + c;
            v = ld4(p.A + off);
            if (KIND != KIND_FWD && p.a_pro == PRO_DZ) v2 = ld4(p.A2 + off);
            a_ok |= 1u << r;
          }
          ra[r] = v; ra2[r] = v2;
        }
      } else {
        // stem: NCHW source, scalar gather, K index = (kh*KW+kw)*Cs + c
        int ekh[4], ekw[4], ec[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int kk = k + j;
          int tap = kk / p.g_Cs;
          ec[j] = kk - tap * p.g_Cs;
          ekh[j] = tap / p.g_KW; ekw[j] = tap - ekh[j] * p.g_KW;
          if (kk >= p.K) ekh[j] = -(1 << 24);
        }
#pragma unroll
        for (int r = 0; r < NA; ++r) {
          f32x4 v = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int sy = a_y0[r] + ekh[j], sx = a_x0[r] + ekw[j];
            if (sy >= 0 && sy < p.g_SH && sx >= 0 && sx < p.g_SW)
              v[j] = p.A[((size_t)(a_pix[r] * p.g_Cs + ec[j]) * p.g_SH + sy) * p.g_SW + sx];
          }
          ra[r] = v;
        }
      }
    } else {
      // WGRAD A': element (k = pixel, i = channel) at A[k*a_ld + i]
      a_ok = 0;
      const int i = m0 + akm_x4 * 4;
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        int k = kbase + akm_k0 + A_KSTEP * r;
        f32x4 v = {0, 0, 0, 0}, v2 = {0, 0, 0, 0};
        if (k < p.K && i < p.M) {
          size_t off = (size_t)k * p.a_ld + i;
          v = ld4(p.A + off);
          if (p.a_pro == PRO_DZ) v2 = ld4(p.A2 + off);
          a_ok |= 1u << r;
        }
        ra[r] = v; ra2[r] = v2;
      }
    }
    // ---------------- B
    if constexpr (KIND == KIND_FWD) {
      const int k = kbase + (tid & 7) * 4;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        int n = n0 + (tid >> 3) + 32 * r;
        f32x4 v = {0, 0, 0, 0};
        if (n < p.N) {
          const float* src = p.B + (size_t)n * p.b_ld + k;
          if constexpr (!NCHW) {
            if (k < p.K) v = ld4(src);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k + j < p.K) v[j] = src[j];
          }
        }
        rb[r] = v;
      }
    } else if constexpr (KIND == KIND_DGRAD) {
      const int n = n0 + bkm_x4 * 4;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        int k = kbase + bkm_k0 + B_KSTEP * r;
        f32x4 v = {0, 0, 0, 0};
        if (k < p.K && n < p.N) {
          int tap = 0, co = k;
          if (taps > 1) { tap = k / p.g_Cs; co = k - tap * p.g_Cs; }
          v = ld4(p.B + (size_t)co * p.b_ld + (size_t)tap * p.b_tapstride + n);
        }
        rb[r] = v;
      }
    } else {
      b_ok = 0;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        int m = kbase + bkm_k0 + B_KSTEP * r;
        f32x4 v = {0, 0, 0, 0};
        if (m < p.K && b_colvalid) {
          int n = m / OHW, rem = m - n * OHW;
          int oy = rem / p.g_OW, ox = rem - oy * p.g_OW;
          if constexpr (!NCHW) {
            int sy = oy * p.g_stride - p.g_pad + b_kh, sx = ox * p.g_stride - p.g_pad + b_kw;
            if (sy >= 0 && sy < p.g_SH && sx >= 0 && sx < p.g_SW) {
              v = ld4(p.B + (size_t)((n * p.g_SH + sy) * p.g_SW + sx) * p.b_ld + b_ci);
              b_ok |= 1u << r;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              int sy = oy * p.g_stride - p.g_pad + b_dy[j], sx = ox * p.g_stride - p.g_pad + b_dx[j];
              if (sy >= 0 && sy < p.g_SH && sx >= 0 && sx < p.g_SW)
                v[j] = p.B[((size_t)(n * p.g_Cs + b_cc[j]) * p.g_SH + sy) * p.g_SW + sx];
            }
          }
        }
        rb[r] = v;
      }
    }
  };

  auto store_tile = [&](int buf, int kt) {
    float* as = As + buf * A_TILE;
    float* bs = Bs + buf * B_TILE;
    if constexpr (A_ROWK) {
      const int kq = (tid & 7) * 4;
      const int k = kt * BK + kq;
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        f32x4 v = ra[r];
        if (!NCHW && p.a_pro != PRO_NONE && ((a_ok >> r) & 1)) v = apply_pro(p.a_pro, v, ra2[r], ac0, ac1, ac2);
        if (!NCHW) {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j >= p.K) v[j] = 0.f;
        }
        *reinterpret_cast<f32x4*>(&as[((tid >> 3) + 32 * r) * LDK + kq]) = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < NA; ++r) {
        f32x4 v = ra[r];
        if (p.a_pro != PRO_NONE && ((a_ok >> r) & 1)) {
          v = apply_pro(p.a_pro, v, ra2[r], ac0, ac1, ac2);
          const int i = m0 + akm_x4 * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) if (i + j >= p.M) v[j] = 0.f;
        }
        *reinterpret_cast<f32x4*>(&as[(akm_k0 + A_KSTEP * r) * LDA_KM + akm_x4 * 4]) = v;
      }
    }
    if constexpr (B_ROWK) {
      const int kq = (tid & 7) * 4;
#pragma unroll
      for (int r = 0; r < NB; ++r)
        *reinterpret_cast<f32x4*>(&bs[((tid >> 3) + 32 * r) * LDK + kq]) = rb[r];
    } else {
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        f32x4 v = rb[r];
        if (KIND == KIND_WGRAD && !NCHW && p.b_pro != PRO_NONE && ((b_ok >> r) & 1))
          v = apply_pro(p.b_pro, v, v, bc0, bc1, bc1);
        *reinterpret_cast<f32x4*>(&bs[(bkm_k0 + B_KSTEP * r) * LDB_KM + bkm_x4 * 4]) = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nkt > 0) {
    load_tile(kt_begin);
    store_tile(0, kt_begin);
  }
  __syncthreads();

  for (int t = 0; t < nkt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nkt) load_tile(kt_begin + t + 1);
    const float* as = As + buf * A_TILE;
    const float* bs = Bs + buf * B_TILE;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_ROWK) {
          fa[i] = *reinterpret_cast<const f32x4*>(&as[(wm0 + i * 32 + li) * LDK + kg * 8 + lh * 4]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fa[i][j] = as[(kg * 8 + lh * 4 + j) * LDA_KM + wm0 + i * 32 + li];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if constexpr (B_ROWK) {
          fb[i] = *reinterpret_cast<const f32x4*>(&bs[(wn0 + i * 32 + li) * LDK + kg * 8 + lh * 4]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[i][j] = bs[(kg * 8 + lh * 4 + j) * LDB_KM + wn0 + i * 32 + li];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    }
    if (t + 1 < nkt) store_tile(buf ^ 1, kt_begin + t + 1);
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  const float inv_hw = p.tap_HW > 0 ? 1.0f / (float)p.tap_HW : 0.f;
  const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.f;
  const int slot = (blockIdx.y + blockIdx.x * 7 + blockIdx.z * 3) & (MMVQA_STAT_SLOTS - 1);
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = n0 + wn0 + b * 32 + li;
    const bool cvalid = col < p.N;
    float bias = (p.bias && cvalid) ? p.bias[col] : 0.f;
    float mks = 1.f, mkb = 0.f, mu1 = 0.f, is1 = 0.f, mu2 = 0.f, is2 = 0.f;
    if (cvalid) {
      if (p.Mk && p.mk_s) { mks = p.mk_s[col]; mkb = p.mk_b[col]; }
      if (p.stat1 && p.stat_bwd) { mu1 = p.mean1[col]; is1 = p.invstd1[col]; }
      if (p.stat2) { mu2 = p.mean2[col]; is2 = p.invstd2[col]; }
    }
    double s_a = 0.0, s_b = 0.0, s_c = 0.0;
    float cs = 0.f;
    int tap_b = -1; float tap_acc = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= p.M || !cvalid) continue;
        float v = acc[a][b][e] + bias;
        if (p.Cpre) p.Cpre[(size_t)row * p.c_ld + col] = v;
        if (p.epi_mode == EPI_TAP_FWD) {
          int bi = row / p.tap_HW;
          float w = act_fwd(p.act, v) * inv_hw;
          if (bi != tap_b) {
            if (tap_b >= 0) atomicAdd(&p.tap_out[(size_t)tap_b * p.N + col], tap_acc);
            tap_b = bi; tap_acc = 0.f;
          }
          tap_acc += w;
          continue;
        }
        if (p.epi_mode == EPI_TAP_BWD) {
          int bi = row / p.tap_HW;
          v = p.tap_dv[(size_t)bi * p.N + col] * inv_hw * act_bwd(p.act, v);
        } else {
          if (p.dact) v *= act_bwd(p.dact, p.Pre[(size_t)row * p.pre_ld + col]);
          v = act_fwd(p.act, v);
          if (p.drop_p > 0.f) {
            float u = rng_uniform(p.drop_seed, (uint32_t)row * (uint32_t)p.N + (uint32_t)col);
            v = (u >= p.drop_p) ? v * keep_scale : 0.f;
          }
          if (p.R) v += p.R[(size_t)row * p.r_ld + col];
        }
        if (p.Mk) {
          float mval = p.Mk[(size_t)row * p.mk_ld + col] * mks + mkb;
          if (!(mval > 0.f)) v = 0.f;
        }
        if (p.c_atomic) atomicAdd(&p.C[(size_t)row * p.c_ld + col], v);
        else p.C[(size_t)row * p.c_ld + col] = v;
        if (p.stat1) {
          s_a += (double)v;
          if (p.stat_bwd) {
            s_b += (double)(v * ((p.Z1[(size_t)row * p.z1_ld + col] - mu1) * is1));
            if (p.stat2) s_c += (double)(v * ((p.Z2[(size_t)row * p.z2_ld + col] - mu2) * is2));
          } else {
            s_b += (double)v * (double)v;
          }
        }
        if (p.colsum) cs += v;
      }
    }
    if (p.epi_mode == EPI_TAP_FWD && tap_b >= 0) atomicAdd(&p.tap_out[(size_t)tap_b * p.N + col], tap_acc);
    if (p.stat1) {
      s_a += __shfl_xor(s_a, 32, 64);
      s_b += __shfl_xor(s_b, 32, 64);
      if (p.stat2) s_c += __shfl_xor(s_c, 32, 64);
      if (lh == 0 && cvalid) {
        double* d1 = p.stat1 + ((size_t)slot * p.N + col) * 2;
        atomicAdd(d1, s_a);
        atomicAdd(d1 + 1, s_b);
        if (p.stat2) {
          double* d2 = p.stat2 + ((size_t)slot * p.N + col) * 2;
          atomicAdd(d2, s_a);
          atomicAdd(d2 + 1, s_c);
        }
      }
    }
    if (p.colsum) {
      cs += __shfl_xor(cs, 32, 64);
      if (lh == 0 && cvalid) atomicAdd(&p.colsum[col], cs);
    }
  }
}

// --------------------------------------------------------------------------- host launch
template <int BM, int BN, int KIND, bool NCHW>
static int launch_cfg(const GemmParams& p, hipStream_t stream) {
  constexpr bool A_ROWK = (KIND != KIND_WGRAD);
  constexpr bool B_ROWK = (KIND == KIND_FWD);
  constexpr int A_TILE = A_ROWK ? BM * LDK : BK * (BM + 4);
  constexpr int B_TILE = B_ROWK ? BN * LDK : BK * (BN + 4);
  constexpr size_t smem = (size_t)(2 * A_TILE + 2 * B_TILE) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, KIND, NCHW>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.splitk);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, KIND, NCHW>), grid, dim3(256), smem, stream, p);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// tile: 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x128
int mmvqa_launch_igemm(GemmParams p, int kind, int nchw, int tile, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MMVQA_OK;
  if (p.g_KH <= 0) { p.g_KH = p.g_KW = 1; p.g_stride = 1; p.g_pad = 0; }
  const int nkt = cdiv(p.K, BK);
  if (tile == 0) {
    // enough workgroups to fill 256 CUs (2 resident per CU) before growing the tile
    long t128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
    long t12864 = (long)cdiv(p.M, 128) * cdiv(p.N, 64);
    if (t128 >= 384) tile = 1;
    else if (t12864 >= 384 || (p.N <= 64 && p.M >= 4096)) tile = 2;
    else tile = 3;
    if (nchw) tile = (kind == KIND_FWD) ? 2 : 3;
  }
  int bm = (tile == 1 || tile == 2) ? 128 : 64;
  int bn = (tile == 1 || tile == 4) ? 128 : 64;
  if (p.splitk <= 0) {
    p.splitk = 1;
    if (kind == KIND_WGRAD) {
      long tiles = (long)cdiv(p.M, bm) * cdiv(p.N, bn);
      int want = (int)((512 + tiles - 1) / tiles);
      int maxs = nkt / 4 > 0 ? nkt / 4 : 1;  // at least 4 K-tiles per split
      p.splitk = want < maxs ? want : maxs;
      if (p.splitk < 1) p.splitk = 1;
      if (p.splitk > 1) p.c_atomic = 1;
    }
  }
  p.ktiles_per_split = cdiv(nkt, p.splitk);
  p.splitk = cdiv(nkt, p.ktiles_per_split);
  if (p.splitk > 1 && !p.c_atomic)
    return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: split-K needs an accumulating epilogue");
#define GO(BM_, BN_)                                                                   \
  do {                                                                                 \
    if (nchw) {                                                                        \
      if (kind == KIND_FWD) return launch_cfg<128, 64, KIND_FWD, true>(p, stream);     \
      if (kind == KIND_WGRAD) return launch_cfg<64, 64, KIND_WGRAD, true>(p, stream);  \
      return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: no NCHW dgrad");                   \
    }                                                                                  \
    if (kind == KIND_FWD) return launch_cfg<BM_, BN_, KIND_FWD, false>(p, stream);     \
    if (kind == KIND_DGRAD) return launch_cfg<BM_, BN_, KIND_DGRAD, false>(p, stream); \
    return launch_cfg<BM_, BN_, KIND_WGRAD, false>(p, stream);                         \
  } while (0)
  if (nchw) { GO(128, 64); }
  switch (tile) {
    case 1: GO(128, 128);
    case 2: GO(128, 64);
    case 4: GO(64, 128);
    default: GO(64, 64);
  }
#undef GO
  return MMVQA_OK;
}
