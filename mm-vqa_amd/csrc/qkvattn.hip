// Fused QKV projection + self-attention of one BertLayer, forward (models/transformer.py:19-30), for the shapes the
// reference trains with: sequence length T <= 32, head dimension 64.
//
// One workgroup owns one (sample b, head h): it computes the 32 x 192 block [q_h | k_h | v_h] = xn[b] (32 x H) times the
// 192 rows of the fused [3H, H] projection weight that belong to head h (+ bias), keeps the block in LDS, writes it to
// the q/k/v tensor the backward pass reads, and runs the attention of that (b, h) on it: S^T = K Q^T / sqrt(d)
// - 10000 (1 - mask[key]), softmax over keys, dropout, O = P V.  No kernel boundary and no HBM round trip between the
// projection and the attention: the two launches this replaces spent most of their time on launch latency and on
// filling 256 CUs with 288 small tiles / 192 single waves (SURVEY.md section 7 "hard part 1").
//
// The projection runs on v_mfma_f32_16x16x4_f32 (exact fp32, same rate as the 32x32x2 form): 32 x 192 outputs are
// 2 x 12 tiles of 16 x 16, six per column strip of 48, so the four SIMDs of the CU carry equal shares.  Eight waves:
// waves 0-3 and 4-7 own the same four strips and split every K tile of 32 between them (one 16-deep k-group each), so
// that each SIMD has two waves to hide LDS / barrier latency behind the other's MFMAs (one wave per SIMD measured
// 38 us per launch against a 15.4 us matrix-pipe bound); the second group's partial sums meet the first's through LDS
// once, after the loop.  LDS double buffered ([row][k], stride 36: 16-byte fragment reads), the next tile's global loads
// in flight behind the MFMAs.  Measured in isolation (tools/qkvattn_bench.py, MMVQA_QA_DBG; B 16, T 32, 12 heads;
// the two launches this replaces: 32.0 + 10.5 us):
//   one wave per SIMD, K tiles of 32                      37.9 us  (K loop 25.6, attention part 7.2, rest 7.8 -- the mask
//                                                                   was loaded inside the attention part: one exposed round trip)
//   two waves per SIMD (this form), K tiles of 32         K loop 22.9, attention part 4.8 with the mask loaded up front
//   K tiles of 64, rows padded to 68 / XOR-swizzled       43.6 / 39.2 us  (K loop 30.5 / 26.0)
//   two wave groups staggered (one writes LDS and issues loads before its MFMAs, the other after), loads two tiles ahead
//                                                         37.5 us  (K loop 24.7): no gain over lockstep here
// The loop runs at ~0.68 of its matrix-pipe time (1 536 MFMA cycles per SIMD and K tile); 192 workgroups use 192 CUs.
// Within a 16-deep k-group lane group g = lane / 16 feeds k = 4g + j at MFMA step j for BOTH operands (the same
// permuted-k trick as igemm.hip).  The attention part is the register-resident scheme of attention.hip (transposed
// scores on v_mfma_f32_32x32x2_f32, probabilities reused as the A operand of P V) reading q, k, v from LDS; waves 0
// and 1 each produce one 32-column half of the context.  Dropout uses the same counter-based stream (seed, element) as
// attn_fwd_kernel, so attn_bwd_kernel regenerates the mask unchanged.
#include "kernels.h"

namespace {

constexpr int QA_BK = 32, QA_LDK = QA_BK + 4, QA_ROWS = 32, QA_COLS = 192, QA_LDS = QA_COLS + 4;
constexpr int QA_A_TILE = QA_ROWS * QA_LDK, QA_B_TILE = QA_COLS * QA_LDK;

struct QkvAttnArgs {
  const float* xn; const float* W; const float* bias; const long long* mask;
  float* qkv; float* probs; float* ctx;
  int B, T, H, heads;
  float sqrt_d, drop_p;
  uint32_t seed;
  int dbg;   // timing experiments only (MMVQA_QA_DBG): 1 = no attention part, 2 = one K tile only, 4 = no q/k/v store
};

__device__ __forceinline__ int erow32(int e, int lh) { return (e & 3) + 8 * (e >> 2) + 4 * lh; }

__global__ __launch_bounds__(512) void qkv_attn_fwd_kernel(const QkvAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][32][36]
  float* Bs = smem + 2 * QA_A_TILE;      // [2][192][36]
  float* S = Bs;                         // after the K loop: [32][196] = q | k | v of this (b, h)
  float* Pt = Bs + QA_ROWS * QA_LDS;     // after the K loop: [4 strips][64 lanes][24] partial sums of waves 4-7
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ks = tid >> 8;
  const int head = blockIdx.x, b = blockIdx.y;
  const int T = p.T, H = p.H;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- loader state per K tile: threads 0..255 one float4 of A, every thread three of B
  const int lrow = tid >> 3, lkq = (tid & 7) * 4;     // lrow 0..63
  const bool a_mine = lrow < QA_ROWS, a_ok = a_mine && lrow < T;
  const float* a_src = p.xn + (size_t)(b * T + (a_ok ? lrow : 0)) * H + lkq;
  const float* b_src[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int c = lrow + 64 * r;                      // local column 0..191: part (q, k, v) x 64
    const int wrow = (c >> 6) * H + head * 64 + (c & 63);
    b_src[r] = p.W + (size_t)wrow * H + lkq;
  }
  // the key mask of this sample: needed by the attention part only, loaded here so that its memory round trip is over by then
  float mk[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = erow32(e, lane >> 5);
    mk[e] = (j < T) ? (float)p.mask[b * T + j] : 1.f;
  }
  f32x4 ra = {0, 0, 0, 0}, rb[3];
  auto gload = [&](int k0) __attribute__((always_inline)) {
    if (a_ok) ra = *reinterpret_cast<const f32x4*>(a_src + k0);
#pragma unroll
    for (int r = 0; r < 3; ++r) rb[r] = *reinterpret_cast<const f32x4*>(b_src[r] + k0);
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    if (a_mine) *reinterpret_cast<f32x4*>(&As[buf * QA_A_TILE + lrow * QA_LDK + lkq]) = ra;
#pragma unroll
    for (int r = 0; r < 3; ++r) *reinterpret_cast<f32x4*>(&Bs[buf * QA_B_TILE + (lrow + 64 * r) * QA_LDK + lkq]) = rb[r];
  };

  f32x4 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  const int cw = wave * 48;
  const int nkt = (p.dbg & 2) ? 1 : H / QA_BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    const float* as = As + buf * QA_A_TILE;
    const float* bs = Bs + buf * QA_B_TILE;
    if (kt + 1 < nkt) gload((kt + 1) * QA_BK);   // in flight behind this tile's MFMAs
    {
      const int kg = ks;                        // this wave group's 16-deep half of the K tile
      f32x4 fa[2], fb[3];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const f32x4*>(&as[(mt * 16 + r16) * QA_LDK + kg * 16 + g * 4]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) fb[nt] = *reinterpret_cast<const f32x4*>(&bs[(cw + nt * 16 + r16) * QA_LDK + kg * 16 + g * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 3; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][j], fb[nt][j], acc[mt][nt], 0, 0, 0);
    }
    if (kt + 1 < nkt) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- the second wave group hands its partial sums to the first (same strip, same lane, same element)
  if (ks == 1) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
        *reinterpret_cast<f32x4*>(&Pt[((wave * 64 + lane) * 6 + mt * 3 + nt) * 4]) = acc[mt][nt];
  }
  __syncthreads();
  if (ks == 0) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
      acc[mt][nt] += *reinterpret_cast<const f32x4*>(&Pt[((wave * 64 + lane) * 6 + mt * 3 + nt) * 4]);

  // ---- + bias -> LDS block S[token][q | k | v] and the q/k/v tensor of the backward pass
  // accumulator element i of lane (g, r16): row = mt*16 + 4g + i, column = cw + nt*16 + r16
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int c = cw + nt * 16 + r16;
    const int part = c >> 6, within = c & 63;
    const int gcol = part * H + head * 64 + within;
    const float bv = p.bias ? p.bias[gcol] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = mt * 16 + 4 * g + i;
        const float v = acc[mt][nt][i] + bv;
        S[row * QA_LDS + c] = v;
        if (row < T && !(p.dbg & 4)) p.qkv[(size_t)(b * T + row) * 3 * H + gcol] = v;
      }
  }
  }
  __syncthreads();
  if (ks == 1 || wave >= 2 || (p.dbg & 1)) return;

  // ---- attention of (b, head): wave 0 -> context columns 0..31, wave 1 -> 32..63 (both form the probabilities)
  const int li = lane & 31, lh = lane >> 5;
  const int dt = wave;
  const bool qvalid = li < T;
  float qf[32], kf[32];
#pragma unroll
  for (int s = 0; s < 32; s += 4) {
    const f32x4 tq = *reinterpret_cast<const f32x4*>(&S[li * QA_LDS + lh * 32 + s]);
    const f32x4 tk = *reinterpret_cast<const f32x4*>(&S[li * QA_LDS + 64 + lh * 32 + s]);
    qf[s] = tq[0]; qf[s + 1] = tq[1]; qf[s + 2] = tq[2]; qf[s + 3] = tq[3];
    kf[s] = tk[0]; kf[s + 1] = tk[1]; kf[s + 2] = tk[2]; kf[s + 3] = tk[3];
  }
  float vv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) vv[e] = S[erow32(e, lh) * QA_LDS + 128 + dt * 32 + li];
  f32x16 sc;
#pragma unroll
  for (int e = 0; e < 16; ++e) sc[e] = 0.f;
#pragma unroll
  for (int s = 0; s < 32; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sc, 0, 0, 0);   // S^T[key][query]

  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = erow32(e, lh);
    float s = -INFINITY;
    if (j < T && qvalid) {
      s = sc[e] * 0.125f;                       // (q.k)/sqrt(64): a power of two, the product IS the quotient; then the mask term
      s = s - 10000.0f * (1.0f - mk[e]);
    }
    sc[e] = s;
    mx = fmaxf(mx, s);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float ex = (sc[e] == -INFINITY) ? 0.f : expf(sc[e] - mx);
    sc[e] = ex;
    sum += ex;
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = qvalid ? 1.0f / sum : 0.f;
  const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = erow32(e, lh);
    float pr = sc[e] * inv;
    if (dt == 0 && p.probs && j < T && qvalid) p.probs[((size_t)(b * p.heads + head) * T + j) * T + li] = pr;
    if (p.drop_p > 0.f) {
      const float u = rng_uniform(p.seed, (uint32_t)(((b * p.heads + head) * T + li) * T + j));
      pr = (u >= p.drop_p) ? pr * keep_scale : 0.f;
    }
    sc[e] = pr;
  }
  f32x16 o;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[e], vv[e], o, 0, 0, 0);   // O[query][d]
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = erow32(e, lh);
    if (i < T) p.ctx[(size_t)(b * T + i) * H + head * 64 + dt * 32 + li] = o[e];
  }
}

}  // namespace

bool k_qkv_attn_fwd_ok(int T, int H, int heads) { return T >= 1 && T <= 32 && heads >= 1 && heads * 64 == H && H % QA_BK == 0; }

int k_qkv_attn_fwd(hipStream_t st, const float* xn, const float* W, const float* bias, const long long* mask, float* qkv,
                   float* probs, float* ctx, int B, int T, int H, int heads, float drop_p, uint32_t seed) {
  if (!k_qkv_attn_fwd_ok(T, H, heads))
    return mmvqa_set_error(MMVQA_ERR_ARG, "qkv_attn_fwd: needs T <= 32 and head dimension 64 (T=%d H=%d heads=%d)", T, H, heads);
  if (!xn || !W || !mask || !qkv || !probs || !ctx) return mmvqa_set_error(MMVQA_ERR_ARG, "qkv_attn_fwd: null operand");
  static const int dbg = getenv("MMVQA_QA_DBG") ? atoi(getenv("MMVQA_QA_DBG")) : 0;
  QkvAttnArgs a{xn, W, bias, mask, qkv, probs, ctx, B, T, H, heads, 8.0f, drop_p, seed, dbg};
  const size_t smem = (size_t)(2 * QA_A_TILE + 2 * QA_B_TILE) * sizeof(float);   // 64.5 KB (the q|k|v block reuses the B tiles)
  static_assert(QA_ROWS * QA_LDS + 4 * 64 * 24 <= 2 * QA_B_TILE, "q|k|v block + the second wave group's partial sums must fit in the B tiles");
  HIP_CHECK_RET(hipFuncSetAttribute((const void*)qkv_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(qkv_attn_fwd_kernel, dim3(heads, B), dim3(512), smem, st, a);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}
