// Fused self-attention for T <= 128 on the gfx950 fp32-input MFMA; no LDS, no score tensor in HBM
// other than the probabilities kept for backward.
//
// One wave owns (batch b, head, 32-query tile).  Scores are produced TRANSPOSED
// (S^T[j][i] = K_j . Q_i: keys on accumulator registers, queries on lanes) so that
//  * softmax over keys is an in-register reduction plus one cross-half exchange, and
//  * the probability tile is directly the A operand of the P.V MFMA (its k index = key sits on
//    the registers/lane-half exactly as v_mfma_f32_32x32x2_f32 wants it; see
//    cdna_hip_programming.md "An accumulator tile as the next MFMA's operand").
//
// Covers both encoders of the reference:
//   BertLayer  (models/transformer.py:19-30): S/sqrt(d) - 10000(1-mask[key]); softmax; dropout; .V
//   RealFormer (models/realformer.py:30-45) : S/sqrt(d) + prev - 10000(1-mask[QUERY]); prev<-S; softmax; .V
// fp32 op order follows the reference: (q.k)/sqrt(d), then +prev, then -10000*(1-m).
#include "common.h"

template <int D2, int VW>
__device__ __forceinline__ void load_half_row(const float* p, float (&r)[D2]) {
  if constexpr (VW == 4) {
#pragma unroll
    for (int s = 0; s < D2; s += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(p + s);
      r[s] = t[0]; r[s + 1] = t[1]; r[s + 2] = t[2]; r[s + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int s = 0; s < D2; ++s) r[s] = p[s];
  }
}

__device__ __forceinline__ int erow(int e, int lh) { return (e & 3) + 8 * (e >> 2) + 4 * lh; }

// --------------------------------------------------------------------------- forward
template <int D, int NJ>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const AttnParams p) {
  constexpr int D2 = D / 2, VW = (D2 % 4 == 0) ? 4 : 1, ND = (D + 31) / 32;
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const int it = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int T = p.T;
  const int qi = it * 32 + li;
  const bool qvalid = qi < T;
  const size_t hb = (size_t)head * p.head_stride;

  float qf[D2];
  if (qvalid) load_half_row<D2, VW>(p.q + (size_t)(b * T + qi) * p.row_stride + hb + lh * D2, qf);
  else {
#pragma unroll
    for (int s = 0; s < D2; ++s) qf[s] = 0.f;
  }
  const float qmask = (qvalid && p.mask_on_query) ? (float)p.mask[b * T + qi] : 1.f;

  // One 32-key tile (T <= 32, the bench's shape): every global load of the kernel is issued here, before the first
  // store.  The stores below (prev_out, probs, out) may alias the inputs as far as the compiler knows, so loads placed
  // after them each waited for a full memory round trip: 20 us for 64 MFMAs of work (round-2 profile).
  constexpr bool HOIST = (NJ == 1);
  float h_mk[HOIST ? 16 : 1], h_pv[HOIST ? 16 : 1], h_v[HOIST ? ND * 16 : 1];
  if constexpr (HOIST) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = erow(e, lh);
      const bool ok = j < T && qvalid;
      h_mk[e] = (ok && !p.mask_on_query) ? (float)p.mask[b * T + j] : 1.f;
      h_pv[e] = (ok && p.prev_in) ? p.prev_in[((size_t)(b * T + qi) * T + j) * p.heads + head] : 0.f;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const int dd = dt * 32 + li;
        h_v[dt * 16 + e] = (j < T && dd < D) ? p.v[(size_t)(b * T + j) * p.row_stride + hb + dd] : 0.f;
      }
    }
  }

  f32x16 sc[NJ];
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
    float kf[D2];
    const int kj = jt * 32 + li;
    if (kj < T) load_half_row<D2, VW>(p.k + (size_t)(b * T + kj) * p.row_stride + hb + lh * D2, kf);
    else {
#pragma unroll
      for (int s = 0; s < D2; ++s) kf[s] = 0.f;
    }
    f32x16 a;
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
    for (int s = 0; s < D2; ++s) a = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], a, 0, 0, 0);
    sc[jt] = a;
  }

  // scale, residual, mask; row max
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = jt * 32 + erow(e, lh);
      float s = -INFINITY;
      if (j < T && qvalid) {
        s = sc[jt][e] / p.sqrt_d;
        const size_t po = ((size_t)(b * T + qi) * T + j) * p.heads + head;
        float mval;
        if constexpr (HOIST) {
          if (p.prev_in) s = s + h_pv[e];
          mval = p.mask_on_query ? qmask : h_mk[e];
        } else {
          if (p.prev_in) s = s + p.prev_in[po];
          mval = p.mask_on_query ? qmask : (float)p.mask[b * T + j];
        }
        s = s - 10000.0f * (1.0f - mval);
        if (p.prev_out) p.prev_out[po] = s;
      }
      sc[jt][e] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float ex = (sc[jt][e] == -INFINITY) ? 0.f : expf(sc[jt][e] - mx);
      sc[jt][e] = ex;
      sum += ex;
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = qvalid ? 1.0f / sum : 0.f;
  const float ks = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.f;
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = jt * 32 + erow(e, lh);
      float pr = sc[jt][e] * inv;
      if (p.probs && j < T && qvalid) p.probs[((size_t)(b * p.heads + head) * T + j) * T + qi] = pr;
      if (p.drop_p > 0.f) {
        float u = rng_uniform(p.seed, (uint32_t)(((b * p.heads + head) * T + qi) * T + j));
        pr = (u >= p.drop_p) ? pr * ks : 0.f;
      }
      sc[jt][e] = pr;
    }
  }

  // O[i][d] = sum_j P[i][j] V[j][d]: A = P^T accumulator (k = key on registers/half), B = V rows
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) {
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    const int dd = dt * 32 + li;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jt * 32 + erow(e, lh);
        float vv;
        if constexpr (HOIST) vv = h_v[dt * 16 + e];
        else vv = (j < T && dd < D) ? p.v[(size_t)(b * T + j) * p.row_stride + hb + dd] : 0.f;
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[jt][e], vv, o, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = it * 32 + erow(e, lh);
      if (i < T && dd < D)
        p.out[(size_t)(b * T + i) * p.out_row_stride + (size_t)head * p.out_head_stride + dd] = o[e];
    }
  }
}

// --------------------------------------------------------------------------- backward
// One wave per (b, head).  Pass A (queries on lanes): dP^T, delta_i = sum_j dP.P, dS^T -> dQ.
// Pass B (keys on lanes): dP, dS -> dK, dV.  Both passes recompute their tiles from q/k/v/dout,
// so nothing but P (and the dropout seed) is carried from forward.
template <int D, int NJ>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const AttnParams p) {
  constexpr int D2 = D / 2, VW = (D2 % 4 == 0) ? 4 : 1, ND = (D + 31) / 32;
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int T = p.T;
  const size_t hb = (size_t)head * p.head_stride;
  const size_t ohb = (size_t)head * p.out_head_stride;
  const float ks = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.f;
  const size_t pbase = (size_t)(b * p.heads + head) * T * T;

  float delta[NJ];  // delta[it] for query it*32+li
  // One 32 x 32 tile (T <= 32): every global load of both passes is issued here, before the first store (see the
  // forward kernel: the dq / dprev stores would otherwise fence the loads of pass B behind them, one round trip each)
  constexpr bool HOIST = (NJ == 1);
  float h_pa[HOIST ? 16 : 1], h_pb[HOIST ? 16 : 1], h_da[HOIST ? 16 : 1], h_db[HOIST ? 16 : 1];
  float h_k[HOIST ? ND * 16 : 1], h_q[HOIST ? ND * 16 : 1], h_do[HOIST ? ND * 16 : 1];
  if constexpr (HOIST) {
    const bool lv = li < T;   // this lane's query (pass A) / key (pass B) exists
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = erow(e, lh);   // the key of pass A, the query of pass B
      const bool ok = r < T && lv;
      h_pa[e] = ok ? p.probs[pbase + (size_t)r * T + li] : 0.f;
      h_pb[e] = ok ? p.probs[pbase + (size_t)li * T + r] : 0.f;
      h_da[e] = (ok && p.dprev_in) ? p.dprev_in[((size_t)(b * T + li) * T + r) * p.heads + head] : 0.f;
      h_db[e] = (ok && p.dprev_in) ? p.dprev_in[((size_t)(b * T + r) * T + li) * p.heads + head] : 0.f;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const int dd = dt * 32 + li;
        const bool okd = r < T && dd < D;
        h_k[dt * 16 + e] = okd ? p.k[(size_t)(b * T + r) * p.row_stride + hb + dd] : 0.f;
        h_q[dt * 16 + e] = okd ? p.q[(size_t)(b * T + r) * p.row_stride + hb + dd] : 0.f;
        h_do[dt * 16 + e] = okd ? p.dout[(size_t)(b * T + r) * p.out_row_stride + ohb + dd] : 0.f;
      }
    }
  }
  // ------------------------------ pass A
#pragma unroll
  for (int it = 0; it < NJ; ++it) {
    const int qi = it * 32 + li;
    const bool qvalid = qi < T;
    float dof[D2];
    if (qvalid) load_half_row<D2, VW>(p.dout + (size_t)(b * T + qi) * p.out_row_stride + ohb + lh * D2, dof);
    else {
#pragma unroll
      for (int s = 0; s < D2; ++s) dof[s] = 0.f;
    }
    f32x16 ds[NJ];
    float dl = 0.f;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
      float vf[D2];
      const int kj = jt * 32 + li;
      if (kj < T) load_half_row<D2, VW>(p.v + (size_t)(b * T + kj) * p.row_stride + hb + lh * D2, vf);
      else {
#pragma unroll
        for (int s = 0; s < D2; ++s) vf[s] = 0.f;
      }
      f32x16 a;
#pragma unroll
      for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
      for (int s = 0; s < D2; ++s) a = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[s], dof[s], a, 0, 0, 0);
      // a[e] = dP'[i=qi][j(e)]  (grad wrt post-dropout probabilities)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jt * 32 + erow(e, lh);
        float pr = 0.f, dp = 0.f;
        if (j < T && qvalid) {
          if constexpr (HOIST) pr = h_pa[e];
          else pr = p.probs[pbase + (size_t)j * T + qi];
          dp = a[e];
          if (p.drop_p > 0.f) {
            float u = rng_uniform(p.seed, (uint32_t)(((b * p.heads + head) * T + qi) * T + j));
            dp = (u >= p.drop_p) ? dp * ks : 0.f;
          }
        }
        dl += dp * pr;
        ds[jt][e] = pr * dp;  // dS = P*dP - P*delta once delta is known (P is re-read: tiny, cache-resident)
      }
    }
    dl += __shfl_xor(dl, 32, 64);
    delta[it] = dl;
    // dS^T[j][i] = P*dP - P*delta (+ dprev) ; then dQ += dS . K / sqrt(d)
    f32x16 dq[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jt * 32 + erow(e, lh);
        float g = 0.f;
        if (j < T && qvalid) {
          float pr;
          if constexpr (HOIST) pr = h_pa[e];
          else pr = p.probs[pbase + (size_t)j * T + qi];
          g = ds[jt][e] - pr * dl;
          const size_t po = ((size_t)(b * T + qi) * T + j) * p.heads + head;
          if constexpr (HOIST) { if (p.dprev_in) g += h_da[e]; }
          else { if (p.dprev_in) g += p.dprev_in[po]; }
          if (p.dprev_out) p.dprev_out[po] = g;
          g = g / p.sqrt_d;
        }
        ds[jt][e] = g;
      }
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const int dd = dt * 32 + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = jt * 32 + erow(e, lh);
          float kv;
          if constexpr (HOIST) kv = h_k[dt * 16 + e];
          else kv = (j < T && dd < D) ? p.k[(size_t)(b * T + j) * p.row_stride + hb + dd] : 0.f;
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[jt][e], kv, dq[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const int dd = dt * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = it * 32 + erow(e, lh);
        if (i < T && dd < D) p.dq[(size_t)(b * T + i) * p.row_stride + hb + dd] = dq[dt][e];
      }
    }
  }

  // ------------------------------ pass B (keys on lanes)
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
    const int kj = jt * 32 + li;
    const bool kvalid = kj < T;
    float vf[D2];
    if (kvalid) load_half_row<D2, VW>(p.v + (size_t)(b * T + kj) * p.row_stride + hb + lh * D2, vf);
    else {
#pragma unroll
      for (int s = 0; s < D2; ++s) vf[s] = 0.f;
    }
    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
#pragma unroll
    for (int it = 0; it < NJ; ++it) {
      float dof[D2];
      const int qi_l = it * 32 + li;
      if (qi_l < T) load_half_row<D2, VW>(p.dout + (size_t)(b * T + qi_l) * p.out_row_stride + ohb + lh * D2, dof);
      else {
#pragma unroll
        for (int s = 0; s < D2; ++s) dof[s] = 0.f;
      }
      // a[e] = dP'[i(e)][j=kj] = dO_i . V_j   (A = dO rows -> registers carry i, B = V rows -> lanes carry j)
      f32x16 a;
#pragma unroll
      for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
      for (int s = 0; s < D2; ++s) a = __builtin_amdgcn_mfma_f32_32x32x2f32(dof[s], vf[s], a, 0, 0, 0);
      f32x16 pd, sd;  // P' (post-dropout) and dS, element e <-> query i(e), lane <-> key kj
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = it * 32 + erow(e, lh);
        const float dli = __shfl(delta[it], erow(e, lh), 64);  // delta of query i lives in lane (i & 31)
        float pr = 0.f, prd = 0.f, g = 0.f;
        if (i < T && kvalid) {
          if constexpr (HOIST) pr = h_pb[e];
          else pr = p.probs[pbase + (size_t)kj * T + i];
          float dp = a[e];
          prd = pr;
          if (p.drop_p > 0.f) {
            float u = rng_uniform(p.seed, (uint32_t)(((b * p.heads + head) * T + i) * T + kj));
            bool keep = u >= p.drop_p;
            dp = keep ? dp * ks : 0.f;
            prd = keep ? pr * ks : 0.f;
          }
          g = pr * (dp - dli);
          if constexpr (HOIST) { if (p.dprev_in) g += h_db[e]; }
          else { if (p.dprev_in) g += p.dprev_in[((size_t)(b * T + i) * T + kj) * p.heads + head]; }
          g = g / p.sqrt_d;
        }
        pd[e] = prd; sd[e] = g;
      }
      // dK[j][d] += sum_i dS[i][j] Q[i][d] ; dV[j][d] += sum_i P'[i][j] dO[i][d]
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const int dd = dt * 32 + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = it * 32 + erow(e, lh);
          const bool ok = (i < T && dd < D);
          float qv, dov;
          if constexpr (HOIST) { qv = h_q[dt * 16 + e]; dov = h_do[dt * 16 + e]; }
          else {
            qv = ok ? p.q[(size_t)(b * T + i) * p.row_stride + hb + dd] : 0.f;
            dov = ok ? p.dout[(size_t)(b * T + i) * p.out_row_stride + ohb + dd] : 0.f;
          }
          // A operand must carry k = i on registers/half with the OUTPUT row (key) on lanes:
          // sd/pd have lane <-> key, register <-> query, i.e. A^T; the MFMA wants A[row=key][k=query].
          // v_mfma A operand: lane l supplies A[row = l&31][k = l>>5]  -> row = key (lane) OK, k = query half.
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[e], qv, dk[dt], 0, 0, 0);
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd[e], dov, dv[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const int dd = dt * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jt * 32 + erow(e, lh);
        if (j < T && dd < D) {
          p.dk[(size_t)(b * T + j) * p.row_stride + hb + dd] = dk[dt][e];
          p.dv[(size_t)(b * T + j) * p.row_stride + hb + dd] = dv[dt][e];
        }
      }
    }
  }
}

// --------------------------------------------------------------------------- host
template <int D>
static int attn_dispatch(const AttnParams& p, int bwd, hipStream_t st) {
  const int nj = (p.T + 31) / 32;
  dim3 gf(nj, p.heads, p.B), gb(1, p.heads, p.B);
#define L(NJ_)                                                                               \
  do {                                                                                       \
    if (bwd) hipLaunchKernelGGL((attn_bwd_kernel<D, NJ_>), gb, dim3(64), 0, st, p);          \
    else hipLaunchKernelGGL((attn_fwd_kernel<D, NJ_>), gf, dim3(64), 0, st, p);              \
  } while (0)
  switch (nj) {
    case 1: L(1); break;
    case 2: L(2); break;
    case 3: L(3); break;
    case 4: L(4); break;
    default: return mmvqa_set_error(MMVQA_ERR_ARG, "attention: T=%d > 128 unsupported", p.T);
  }
#undef L
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int mmvqa_launch_attention(const AttnParams& p, int head_dim, int bwd, hipStream_t st) {
  switch (head_dim) {
    case 8: return attn_dispatch<8>(p, bwd, st);
    case 12: return attn_dispatch<12>(p, bwd, st);
    case 64: return attn_dispatch<64>(p, bwd, st);
    case 96: return attn_dispatch<96>(p, bwd, st);
    default: return mmvqa_set_error(MMVQA_ERR_ARG, "attention: head_dim=%d unsupported (8,12,64,96)", head_dim);
  }
}
