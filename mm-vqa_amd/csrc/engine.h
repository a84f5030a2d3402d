// Host runtime of the MMBERT hot path: parameter table, workspace plan and the launch sequences of
// Model.forward / backward (models/mmbert.py:129-167).  Pure C++ over the HIP launchers in kernels.h.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

struct TensorSpec {
  std::string name;
  int kind;  // 0 param, 1 float buffer, 2 int64 buffer
  int ndim;
  long long shape[4];
  long long offset;
  int channels_last;
};

struct BNRef {
  int C = 0, reps = 1;
  int slots = MMVQA_STAT_SLOTS;   // replicas of the per-channel sums its producers spread their atomics over
  float eps = 1e-5f;
  long long gamma = 0, beta = 0, rmean = 0, rvar = 0, nbt = 0;
  // plan (offsets into workspace, floats; stats in doubles from ws_d)
  size_t stat_f = 0, stat_b = 0;  // double offsets
  size_t scale = 0, shift = 0, mean = 0, invstd = 0, P = 0, Q = 0, R = 0;
  double count = 0;
};

struct ConvRef {
  long long w = 0;
  int Cin = 0, Cout = 0, KH = 1, stride = 1, pad = 0;
};

struct BlockRef {  // torchvision Bottleneck
  ConvRef c1, c2, c3, cd;
  BNRef b1, b2, b3, bd;
  bool has_ds = false;
  int N = 0, H = 0, W = 0, OH = 0, OW = 0;   // input / output spatial dims
  size_t z1 = 0, z2 = 0, z3 = 0, zd = 0, out = 0;  // workspace offsets
};

struct LinRef {
  long long w = 0, b = -1;
  int in = 0, out = 0;
};

struct LNRef {
  long long g = 0, b = 0;
};

// timm EfficientNetV2 block: ConvBnAct (cn) | EdgeResidual / Fused-MBConv (er) | InvertedResidual / MBConv (ir)
struct EffBlock {
  int type = 0, cin = 0, cout = 0, mid = 0, rd = 0, stride = 1;
  bool skip = false;
  ConvRef c_a;        // cn: conv 3x3 | er: conv_exp 3x3 | ir: conv_pw 1x1
  BNRef b_a;
  long long dw_w = 0;  // ir: depthwise 3x3 weight [mid][3][3]
  BNRef b_dw;         // ir: bn2
  LinRef se_r, se_e;  // ir: squeeze-excite conv_reduce / conv_expand (1x1 with bias)
  ConvRef c_p;        // er / ir: conv_pwl 1x1
  BNRef b_p;          // er: bn2 | ir: bn3
  int N = 0, H = 0, W = 0, OH = 0, OW = 0, pad = 0;
  size_t za = 0, zdw = 0, zp = 0, out = 0, pool = 0, rpre = 0, r = 0, gpre = 0, gate = 0;
  int feature = -1;   // index of the tap fed by this block's output, or -1
};

struct TapRef {
  long long w = 0;
  int C = 0;
  int HW = 0;
  long M = 0;
};

struct BertLayerRef {
  LinRef qkv, proj, fc1, fc2;
  // workspace
  size_t xn1, mean1, rstd1, qkvo, probs, ctx, y, xn2, mean2, rstd2, pre1, h1, z;
};

struct RFLayerRef {
  LinRef kqv, proj, ff0, ff2;
  LNRef ln1, ln2;
  size_t kqvo, probs, prev, res, s1, x1, mean1, rstd1, pre, hact, s2, x2, mean2, rstd2;
};

// launch classes of the profiler: igemm_kernel | attention | everything else without matrix work | matrix work OUTSIDE
// igemm_kernel (the register-resident stem tap of tapthin.hip, the squeeze-excite fully connected layers of se.hip)
enum { PROF_IGEMM = 0, PROF_ATTN = 1, PROF_OTHER = 2, PROF_MATRIX = 3, PROF_NCLS = 4 };
// profiler regions (per-block rooflines): which part of the step a launch belongs to
enum { REG_BACKBONE = 0, REG_TAP = 1, REG_QKV = 2, REG_ATTN = 3, REG_ENC = 4, REG_HEAD = 5, REG_EMBED = 6,
       REG_BNCOEF = 7, REG_QKV_ATTN = 8 /* the fused projection + attention launch (qkvattn.hip) */, REG_N = 9 };

// HBM-bound kernels the profiler reports one by one (algorithmic bytes / measured time against the 8 TB/s roofline)
enum { HB_NONE = 0, HB_BN_ADD_RELU, HB_MAXPOOL_FWD, HB_MAXPOOL_BWD, HB_LAYERNORM_FWD, HB_LAYERNORM_BWD, HB_DROPOUT_COPY,
       HB_BN_ACT_ADD, HB_DWCONV_FWD, HB_DWCONV_BWD_DATA, HB_DWCONV_BWD_WEIGHT, HB_SE_POOL, HB_SE_DGATE, HB_ACT_BWD_STATS,
       HB_TAP_THIN_FWD, HB_TAP_THIN_BWD, HB_N };

constexpr size_t SK_WS_FLOATS = (size_t)8 << 20;   // 32 MB: 8 splits of a 224-tile (64x64) product
constexpr int SK_CNT_N = 16384;                    // tiles a persistent launch may have (one arrival ticket each)

struct mmvqa_engine {
  mmvqa_model_desc d;
  std::vector<TensorSpec> specs;
  long long n_params = 0, n_bufs = 0, n_nbt = 0;

  // ---- parameters
  ConvRef stem_conv;
  BNRef stem_bn;
  std::vector<BlockRef> blocks;
  std::vector<EffBlock> eff;     // EfficientNetV2 body (cnn == 1)
  size_t eff_a0 = 0;             // materialised stem activation silu(bn1(conv_stem))
  size_t sk_ws[3] = {0, 0, 0};   // split-K partial-tile scratch per stream (igemm sk_ws): caller's, side, tap stream
  size_t eff_gA[2] = {0, 0}, eff_gB[2] = {0, 0}, eff_separt = 0, eff_se[6];   // backward scratch ([pixels, mid], two of each: blocks alternate,
                                                                            // the side stream still reads block i's while block i-1 writes), squeeze-excite temporaries
  int layer_end[4];          // index of last block of layer1..4
  TapRef taps[5];            // order of the reference's return tuple: conv2(l4),conv3(l3),conv4(l2),conv5(l1),conv7(stem)
  long long emb_word = 0, emb_pos = 0, emb_type = 0;
  LNRef emb_ln;
  LNRef norm1, norm2;        // BertLayer (norm2 never used: quirk 2)
  std::vector<BertLayerRef> bert;
  std::vector<RFLayerRef> rf;
  LinRef fc1, cls0, cls2, head0, head2;
  LNRef cls_ln;

  // ---- plan
  int B = 0, T = 0, IH = 0, IW = 0;
  bool planned = false, bound = false;
  size_t ws_floats = 0;
  int SH = 0, SW = 0, PH = 0, PW = 0;  // stem output dims / pooled dims
  size_t z0 = 0, p0 = 0, pool_idx = 0, vis = 0, dvis = 0, du = 0, tapgrad[5], gbuf[3], g1buf = 0, g2buf = 0,
         dstmp = 0, statzone = 0, statzone_floats = 0;
  size_t emb_out = 0, emb_xhat = 0, emb_rstd = 0;
  size_t enc_out_final = 0;
  size_t hd_upre = 0, hd_u = 0, hd_c0 = 0, hd_c1 = 0, hd_mean = 0, hd_rstd = 0, hd_pool = 0;
  size_t sc_pool = 0, sc_pre = 0, sc_a = 0, sc_f = 0, sc_nrm = 0, sc_y = 0;
  size_t t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_big = 0, t_dprev[2];  // backward scratch
  // ---- bound pointers
  float *params = nullptr, *grads = nullptr, *bufs = nullptr, *ws = nullptr;
  long long* nbt = nullptr;
  // ---- step state
  const long long *ids = nullptr, *seg = nullptr, *mask = nullptr;
  const float* img = nullptr;
  int training = 0;
  uint32_t seed = 0;
  float* logits = nullptr;
  int logits_ld = 0;
  float* feat = nullptr;
  // ---- per-shape kernel configuration
  IgemmTuner tuner;
  // ---- second stream: weight-gradient GEMMs and tap backward run beside the data-gradient chain
  hipStream_t side = nullptr;
  hipStream_t side2 = nullptr;   // the tap backward (needed late, long launches): a stream of its own so that it does not sit in
                                 // front of the weight gradients the dependency chain waits for
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  int use_side = 1;
  int bn_fold = 1;   // ResNet, training: BatchNorm coefficients are folded inside the consuming launches (mmvqa_bn_fold)
  // ---- per-geometry tap-validity tables of the 3x3 weight gradients (mmvqa_gemm_desc.pixmask): planned into the
  // workspace, ALL built on the caller's stream at the start of the first forward after every bind -- before either
  // stream can launch a weight gradient (the workspace is the caller's: see mmvqa_engine_bind in mmvqa.h)
  struct PixGeom { int N, OH, OW, H, W, KH, stride, pad; size_t off; };
  std::map<std::string, PixGeom> pixmask_off;
  bool ws_ready = false;                        // persistent workspace state (tables, tickets) is in place
  size_t sk_cnt[3] = {0, 0, 0};                 // arrival tickets of ticketed / persistent launches (one set per stream)
  // ---- gradient-ready notifications (data-parallel overlap): called on the host right after the kernels that
  // complete grads[lo, hi) have been enqueued and the main stream has been ordered behind them
  void (*grad_cb)(void* user, long long lo, long long hi) = nullptr;
  void* grad_cb_user = nullptr;
  long long enc_lo = 0, emb_hi = 0;
  // ---- profiling
  int prof_on = 0;
  int prof_reg = REG_BACKBONE;
  struct ProfRec { hipEvent_t a, b; int cls, reg; double flops; int tag; double bytes; };
  long long tag_launch[HB_N];
  double tag_ms[HB_N], tag_bytes[HB_N];
  std::vector<ProfRec> prof;
  long long prof_launch[PROF_NCLS];
  double prof_ms[PROF_NCLS], prof_flops[PROF_NCLS];
  long long reg_launch[REG_N][PROF_NCLS];
  double reg_ms[REG_N][PROF_NCLS], reg_flops[REG_N][PROF_NCLS];
};
