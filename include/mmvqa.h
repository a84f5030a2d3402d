/* mmvqa.h -- C ABI of the MI355X-native MMBERT hot path (libmmvqa_hip.so).
 *
 * The reference (DannielSilva/MM-VQA) has no FFI: its boundary is the Python object
 * protocol  Model(args) / model(img, ids, seg, mask) / state_dict  (models/mmbert.py:129-167,
 * pretrain/roco_utils.py:214-247, vqamed2019/utils.py:633-666).  This header is the native side of
 * that boundary: plain pointers + sizes + a hipStream_t, no torch types.  Every entry point
 *   - borrows raw DEVICE pointers (never allocates or frees caller memory),
 *   - enqueues on the given stream and returns without synchronising,
 *   - returns 0 on success or a negative code; mmvqa_last_error() gives the message.
 * INTEGRATION.md shows the ctypes binding the Python shim uses.
 *
 * Layout conventions: activations are NHWC ("channels last") fp32; conv weights are
 * [Cout][KH][KW][Cin]; linear weights [out][in]; ids/masks/labels int64.
 */
#ifndef MMVQA_H
#define MMVQA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mmvqa_stream_t; /* hipStream_t */

#define MMVQA_ACT_NONE 0
#define MMVQA_ACT_RELU 1
#define MMVQA_ACT_GELU 2 /* models/transformer.py:7-8 */
#define MMVQA_ACT_SERF 3 /* models/serf.py:23-24 */
#define MMVQA_ACT_SILU 4 /* timm tf_efficientnetv2_m activations */
#define MMVQA_ACT_SIGMOID 5 /* squeeze-excite gate */

#define MMVQA_KIND_FWD 0   /* C[pix,co]      = sum X[pix@tap,ci] W[co,tap,ci]      */
#define MMVQA_KIND_DGRAD 1 /* C[pix_in,ci]   = sum dZ[pix_out,co] W[co,tap,ci]     */
#define MMVQA_KIND_WGRAD 2 /* C[co,(tap,ci)] = sum dZ[pix,co] X[pix@tap,ci]        */

#define MMVQA_STAT_SLOTS 16

/* BatchNorm coefficients folded inside the CONSUMING launch (training mode).
 * The launches that produce a BatchNorm's input accumulate its per-channel sums into `stat`
 * ([MMVQA_STAT_SLOTS][C][2] doubles, the first `slots` replicas in use: forward sum z / sum z^2, backward
 * sum g / sum g*xhat).  A launch whose operand prologue applies that BatchNorm can derive the coefficients
 * from the sums in its own setup phase, so that no coefficient launch (mmvqa_bn_coef_fwd / _bwd) sits between
 * producer and consumer on the dependency chain.  With `publish` one workgroup of the launch also writes what
 * the separate launch would have written: coefficients for later launches, running statistics with torch's
 * momentum rule applied `reps` times (models/image_encoding.py:72-86 runs the backbone prefix 5 times),
 * num_batches_tracked, and -- backward -- the gamma / beta gradients.  stat == NULL: no folding. */
typedef struct mmvqa_bn_fold {
  const double* stat;
  int slots;          /* replicas of the sums that are in use (power of two, <= MMVQA_STAT_SLOTS) */
  int bwd;            /* 0: -> scale, shift (and mean, invstd) ; 1: -> P, Q, R of dz = P*g + Q*z + R */
  int publish;
  int reps;
  double count;       /* elements per channel */
  double keep;        /* (1 - momentum)^reps */
  float eps;
  int reserved;
  const float* gamma;
  const float* beta;    /* forward */
  const float* mean;    /* backward: the forward pass's batch mean / 1/sqrt(var+eps) */
  const float* invstd;
  float* out0;          /* publish: forward scale | backward P */
  float* out1;          /*          forward shift | backward Q */
  float* out2;          /*          forward mean  | backward R */
  float* out3;          /*          forward invstd */
  float* run_mean;
  float* run_var;
  long long* nbt;
  float* dgamma;        /* backward publish: += sum g*xhat */
  float* dbeta;         /*                   += sum g */
} mmvqa_bn_fold;

/* Descriptor of one implicit-GEMM launch (see mm-vqa_amd/csrc/igemm.hip for the field semantics).
 * Replaces: torch.nn.Conv2d / nn.Linear forward+backward as used by models/image_encoding.py:53-86,
 * models/transformer.py:13-15,45-48, models/realformer.py:13-27, models/mmbert.py:133-148. */
typedef struct mmvqa_gemm_desc {
  int M, N, K;
  int splitk;
  int ktiles_per_split;
  const float* A;
  const float* A2;
  const float* a_c0;
  const float* a_c1;
  const float* a_c2;
  int a_pro;
  int a_ld;
  const float* B;
  const float* b_c0;
  const float* b_c1;
  int b_pro;
  int b_ld;
  int b_tapstride;
  int g_SH, g_SW, g_Cs, g_OH, g_OW, g_KH, g_KW, g_stride, g_pad;
  int g_nchw;
  float* C;
  int c_ld;
  int c_atomic;
  float* Cpre;
  const float* bias;
  int act;
  int dact;
  const float* Pre;
  int pre_ld;
  float drop_p;
  uint32_t drop_seed;
  const float* R;
  int r_ld;
  int epi_mode;
  int tap_HW;
  float* tap_out;
  const float* tap_dv;
  const float* Mk;
  int mk_ld;
  const float* mk_s;
  const float* mk_b;
  double* stat1;
  int stat_bwd;
  const float* Z1;
  int z1_ld;
  const float* mean1;
  const float* invstd1;
  double* stat2;
  const float* Z2;
  int z2_ld;
  const float* mean2;
  const float* invstd2;
  float* colsum;
  const float* gate; /* PRO_SILU_GATE: [image][g_Cs] squeeze-excite gates; image = pixel / gate_hw */
  int gate_hw;
  int mk_mode;       /* 0: ReLU mask (Mk*s+b > 0), 1: multiply by SiLU'(Mk*s+b) */
  const int* pixmask; /* optional, KIND_WGRAD of a stride-1 "same" convolution: per output pixel the bit mask of filter
                         taps that fall inside the image (mmvqa_pixmask); enables the uniform-tap weight-gradient loaders */
  float* sk_ws;       /* optional, KIND_FWD / KIND_DGRAD: scratch of sk_ws_floats floats.  With it a launch that has few
                         output tiles and a long contraction may split K over workgroups (splitk, or the launcher's /
                         tuner's choice when splitk <= 0): partial tiles go to the scratch and a second launch sums them
                         and applies the whole epilogue.  Needs splitk * M * N <= sk_ws_floats; one stream at a time. */
  long long sk_ws_floats;
  mmvqa_bn_fold a_fold; /* optional: the coefficients of the A prologue (a_pro AFFINE_RELU: scale/shift; DZ: P/Q/R) come
                           from raw sums instead of a_c0 / a_c1 / a_c2 (which may then be NULL) */
  int stat_slots;       /* replicas the statistics epilogue (stat1 / stat2) spreads its sums over; 0 = MMVQA_STAT_SLOTS */
  int persist;          /* > 0: persistent ("stream-K") launch of that many workgroups, each walking an equal share of the
                           launch's K-tile iterations; tiles cut over several workgroups are completed by the workgroup
                           that arrives last.  Plain-epilogue products only; an accumulating product (c_atomic) needs
                           nothing else, any other needs sk_ws (partial tiles) and sk_cnt (tickets).  0 = one workgroup
                           per tile (and split).  The launcher falls back to that form when a condition is not met. */
  unsigned int* sk_cnt; /* persist: sk_cnt_n arrival tickets, ZERO before the first launch; every launch leaves them zero */
  int sk_cnt_n;
  int reserved0;
} mmvqa_gemm_desc;

/* Fused attention (models/transformer.py:19-30 and models/realformer.py:30-45). */
typedef struct mmvqa_attn_desc {
  const float* q;
  const float* k;
  const float* v;
  int row_stride, head_stride;
  float* out;
  int out_row_stride, out_head_stride;
  const long long* mask;
  int mask_on_query;
  const float* prev_in;
  float* prev_out;
  float* probs;
  int B, T, heads;
  float sqrt_d;
  float drop_p;
  uint32_t seed;
  const float* dout;
  float* dq;
  float* dk;
  float* dv;
  const float* dprev_in;
  float* dprev_out;
} mmvqa_attn_desc;

/* Architecture of one Model(args) instance: the option surface of pretrain/roco_train.py:23-60,
 * vqamed2019/train.py:32-79 that the hot path reads (SURVEY.md section 5 "Config"). */
typedef struct mmvqa_model_desc {
  int cnn;              /* 0 = resnet (torchvision layout), 1 = tf_efficientnetv2_m (timm features_only) */
  int resnet_layers[4]; /* (3,8,36,3) = resnet152 */
  int resnet_width;     /* 64 */
  int effnet_depth_div; /* 1 = full depth; >1 divides stage repeats (tests) */
  int encoder;          /* 0 = transformer (BertLayer pre-LN), 1 = realformer */
  int hidden, heads, n_layers;
  int emb_vocab, max_pos, type_vocab;
  int num_vis;
  int head_kind;        /* 0 = roco (per-token MLM logits), 1 = VQA-Med (mean-pooled logits) */
  int n_classes;        /* width of classifier[2] */
  int supcon, feat_dim;
  int use_relu;
  float p_drop;         /* --hidden_dropout_prob (BertLayer attention + residual dropouts) */
  float p_emb_drop;     /* BertEmbeddings dropout (0.1) */
  float p_rf_drop;      /* RealFormer dp1/dp2 (0.1) */
} mmvqa_model_desc;

/* gradient-ready notification of mmvqa_engine_set_grad_callback: grads[lo, hi) (float offsets) are final */
typedef void (*mmvqa_grad_cb)(void* user, long long lo, long long hi);

typedef struct mmvqa_engine mmvqa_engine;

/* ---- library ---------------------------------------------------------------------------- */
int mmvqa_version(void);
const char* mmvqa_last_error(void);
size_t mmvqa_sizeof_gemm_desc(void);
size_t mmvqa_sizeof_attn_desc(void);
size_t mmvqa_sizeof_model_desc(void);

/* ---- op level (each is what one torch op of the reference's hot path lowers to) ------------ */
/* kind: MMVQA_KIND_*; nchw: 1 only for the 7x7 stem (NCHW image source); tile: 0 auto */
int mmvqa_igemm(const mmvqa_gemm_desc* d, int kind, int nchw, int tile, mmvqa_stream_t s);
int mmvqa_attention(const mmvqa_attn_desc* d, int head_dim, int backward, mmvqa_stream_t s);
/* BertLayer forward, projection and attention in ONE launch (models/transformer.py:19-30: proj_q / proj_k / proj_v,
 * split heads, scores / sqrt(d) - 10000 (1 - mask), softmax, dropout, @ v, merge heads) for T <= 32 and head dimension 64
 * (hidden = 64 * heads): xn [B*T, hidden] -> qkv [B*T, 3*hidden] (q | k | v, kept for the backward pass),
 * probs [B, heads, T(key), T(query)] (before dropout), ctx [B*T, hidden].  W = the three projection weights back to back
 * [3*hidden, hidden], bias [3*hidden] or NULL.  Same dropout stream as mmvqa_attention with the same seed. */
int mmvqa_qkv_attention_fwd(mmvqa_stream_t s, const float* xn, const float* W, const float* bias, const long long* mask,
                            float* qkv, float* probs, float* ctx, int B, int T, int hidden, int heads, float drop_p,
                            uint32_t seed);

/* BatchNorm2d (train: batch stats + running update repeated `reps` times; eval: running stats) */
int mmvqa_bn_coef_fwd(mmvqa_stream_t s, const double* stat, int C, double count, float eps, const float* gamma,
                      const float* beta, float* run_mean, float* run_var, long long* nbt, float momentum, int reps,
                      int training, float* scale, float* shift, float* mean, float* invstd);
int mmvqa_bn_coef_bwd(mmvqa_stream_t s, const double* stat, int C, double count, const float* gamma,
                      const float* mean, const float* invstd, int training, float* P, float* Q, float* R,
                      float* dgamma, float* dbeta);
int mmvqa_bn_add_relu(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, const float* idn,
                      const float* id_sc, const float* id_sh, float* out, long rows, int C);
/* the same block end, relu(bn3(z) + [bn_d](idn)), with the BatchNorm coefficients folded from their raw sums inside the
 * launch (forward mmvqa_bn_fold; fd == NULL: the identity branch is added as it is) -- no coefficient launch between
 * conv3 and the block end (models/image_encoding.py:72-86 via torchvision Bottleneck.forward) */
int mmvqa_bn_add_relu_fold(mmvqa_stream_t s, const float* z, const mmvqa_bn_fold* f3, const float* idn,
                           const mmvqa_bn_fold* fd, float* out, long rows, int C);
int mmvqa_maxpool_fwd(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, float* out,
                      unsigned char* idx, int N, int H, int W, int C, int OH, int OW);
int mmvqa_maxpool_bwd(mmvqa_stream_t s, const float* gp, const unsigned char* idx, const float* extra,
                      const float* z, const float* sc, const float* sh, const float* mean, const float* invstd,
                      float* g0, double* stat, int N, int H, int W, int C, int OH, int OW);
/* LayerNorm over the last axis, optional residual input (y = LN(x + res)) */
int mmvqa_layernorm_fwd(mmvqa_stream_t s, const float* x, const float* res, const float* gamma, const float* beta,
                        float* y, float* sum_out, float* mean, float* rstd, int rows, int H, float eps);
int mmvqa_layernorm_bwd(mmvqa_stream_t s, const float* dy, const float* x, const float* gamma, const float* mean,
                        const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta, int rows,
                        int H);
/* HF BertEmbeddings + visual-token overwrite (models/mmbert.py:60-67); vis is [num_vis][B][H] */
int mmvqa_embed_fwd(mmvqa_stream_t s, const long long* ids, const long long* seg, const float* word,
                    const float* pos, const float* type, const float* gamma, const float* beta, const float* vis,
                    float* out, float* xhat, float* rstd, int B, int T, int H, int num_vis, float eps,
                    float drop_p, uint32_t seed);
int mmvqa_embed_bwd(mmvqa_stream_t s, const float* dout, const long long* ids, const long long* seg,
                    const float* xhat, const float* rstd, const float* gamma, float* dword, float* dpos,
                    float* dtype, float* dgamma, float* dbeta, float* dvis, int B, int T, int H, int num_vis,
                    float drop_p, uint32_t seed, int pad_idx);
int mmvqa_meanpool_fwd(mmvqa_stream_t s, const float* h, const long long* mask, float* out, int B, int T, int H);
int mmvqa_meanpool_bwd(mmvqa_stream_t s, const float* dout, const long long* mask, float* dh, int B, int T, int H,
                       int accumulate);
/* log_softmax + NLLLoss() + masked argmax accuracy (pretrain/roco_utils.py:235-236,257-265).
 * out3 = {mean loss, #target>0, #correct}; row_lse (nullable) = log sum exp of each row, kept for mmvqa_mlm_grad;
 * dlogits (nullable) = (softmax - onehot) * (*gscale_ptr) * gscale_mul.  Rows that are 16-byte aligned with
 * ld >= round_up(V,4) take the single-pass float4 kernels (one read of the logits); others a scalar form. */
int mmvqa_mlm_loss(mmvqa_stream_t s, const float* logits, int ld, const long long* target, float* row_loss,
                   float* row_lse, long long* pred, float* dlogits, int dld, const float* gscale_ptr, float gscale_mul,
                   int rows, int V, float* out3);
/* backward of the above from the saved row_lse: dlogits = (exp(logits - lse) - onehot) * (*gscale_ptr) * gscale_mul
 * (gscale_ptr: device scalar = the upstream gradient of the mean loss, nullable); pad columns are zeroed.
 * Requires aligned rows (see above). */
int mmvqa_mlm_grad(mmvqa_stream_t s, const float* logits, int ld, const long long* target, const float* row_lse,
                   float* dlogits, int dld, const float* gscale_ptr, float gscale_mul, int rows, int V);
/* ASLSingleLabel (models/asl_singlelabel.py:23-53): per-sample losses + dlogits*gscale */
int mmvqa_asl_loss(mmvqa_stream_t s, const float* logits, int ld, const long long* target, float* row_loss,
                   float* dlogits, int dld, int rows, int C, float gamma_pos, float gamma_neg, float eps,
                   float gscale);
int mmvqa_l2norm_fwd(mmvqa_stream_t s, const float* x, float* y, float* nrm, int rows, int D);
int mmvqa_l2norm_bwd(mmvqa_stream_t s, const float* dy, const float* y, const float* nrm, float* dx, int rows,
                     int D);
/* SupConLoss.forward(features) without labels/mask = SimCLR (models/SupConLoss/loss.py:21-98);
 * f is [2N][D] view-major (loss.py:57), D <= 256; df (nullable) = dloss/df * gscale; ws = 4*N floats of scratch
 * (row log-sums and row losses).  Tiled over row blocks: any N (the all-gathered view set 2N*world of a
 * data-parallel job, SURVEY 8(e) collective 2). */
int mmvqa_supcon_loss(mmvqa_stream_t s, const float* f, float* loss, float* df, float* ws, int N, int D, float temp,
                      float base_temp, float gscale);
/* ---- EfficientNetV2 (timm tf_efficientnetv2_m as models/image_encoding.py:15,26,100-115 instantiates it) pieces,
 * NHWC fp32; sc/sh = BatchNorm scale/shift of the producing conv (applied on load), stat = [16][C][2] doubles.
 * depthwise 3x3 (MBConv conv_dw): z2 = dw(silu(z1*s1+b1)), statistics of z2; TF "SAME" padding via pad (begin) */
int mmvqa_dwconv_fwd(mmvqa_stream_t s, const float* z1, const float* s1, const float* b1, const float* w, float* z2,
                     double* stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad);
/* g1 = dw^T(P*g2 + Q*z2 + R) * silu'(z1*s1+b1); BatchNorm-backward sums of bn1 (sum g1, sum g1*xhat1) */
int mmvqa_dwconv_bwd_data(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                          const float* R, const float* w, const float* z1, const float* s1, const float* b1,
                          const float* mean1, const float* invstd1, float* g1, double* stat, int N, int H, int W, int C,
                          int OH, int OW, int stride, int pad);
/* dw[c][tap] += sum_pix (P*g2 + Q*z2 + R)[pix,c] * silu(z1*s1+b1)[pix@tap,c] */
int mmvqa_dwconv_bwd_weight(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                            const float* R, const float* z1, const float* s1, const float* b1, float* dw, int N, int H,
                            int W, int C, int OH, int OW, int stride, int pad);
/* squeeze-excite: pool[n][c] = mean_hw silu(z*sc+sh); dgate[n][c] = sum_hw t * silu(z*sc+sh) */
int mmvqa_se_pool(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, float* pool, int N, int HW, int C);
int mmvqa_se_dgate(mmvqa_stream_t s, const float* t, const float* z, const float* sc, const float* sh, float* dgate,
                   int N, int HW, int C);
/* The same launches with the BatchNorm coefficients of their operand folded from the producer's raw sums inside the
 * launch (mmvqa_bn_fold; round 3): the FIRST consumer of a train-mode BatchNorm needs no coefficient launch in front.
 * Each workgroup derives the coefficients of its 32 / 64 channels once into LDS; with fold->publish the image-0
 * workgroup of a channel block writes what the coefficient launch would have (scale / shift / mean / invstd / running
 * statistics with the k-fold rule / batch counter; backward: P, Q, R and dgamma, dbeta) for the later consumers.
 * forward folds: (s1, b1) / (sc, sh) are ignored; backward folds (bwd_data publishes, bwd_weight never): (P, Q, R) are.
 * fold == NULL: identical to the plain entry.  (timm MBConv: conv_pw -> bn1 -> SiLU -> conv_dw -> bn2 -> SiLU -> SE) */
int mmvqa_dwconv_fwd_fold(mmvqa_stream_t s, const float* z1, const float* s1, const float* b1, const float* w, float* z2,
                          double* stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad,
                          const mmvqa_bn_fold* fold);
int mmvqa_dwconv_bwd_data_fold(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                               const float* R, const float* w, const float* z1, const float* s1, const float* b1,
                               const float* mean1, const float* invstd1, float* g1, double* stat, int N, int H, int W,
                               int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold);
int mmvqa_dwconv_bwd_weight_fold(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                                 const float* R, const float* z1, const float* s1, const float* b1, float* dw, int N,
                                 int H, int W, int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold);
int mmvqa_se_pool_fold(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, float* pool, int N, int HW,
                       int C, const mmvqa_bn_fold* fold);
/* visual-token tap of a feature map with few channels (models/image_encoding.py:53-62 at the stem: conv1x1 C -> N,
 * activation, global average pool): out[img][n] += mean_hw act(sum_c x'[pix][c] W[n][c]) with x' = x, or
 * relu(x*sc+sh) when sc/sh are given; x [M, C] NHWC rows, M = images*HW.  C in {24, 64}, M and HW multiples of 32,
 * M >= 32768 (mmvqa_tap_thin_ok); `out` [images, N] must be zero on entry. */
int mmvqa_tap_thin_ok(long M, int N, int C, int HW);
int mmvqa_tap_thin_fwd(mmvqa_stream_t s, const float* x, const float* sc, const float* sh, const float* W, float* out,
                       long M, int N, int C, int HW, int act);
/* backward recompute of the same tap: du[pix][n] = dv[img][n] / HW * act'(sum_c x'[pix][c] W[n][c]); du [M, N] */
int mmvqa_tap_thin_bwd(mmvqa_stream_t s, const float* x, const float* sc, const float* sh, const float* W,
                       const float* dv, float* du, long M, int N, int C, int HW, int act);
/* squeeze-excite fully connected layers (timm SqueezeExcite conv_reduce / conv_expand on the pooled [B, mid] tensor):
 *   rpre = pool Wr^T + br, r = silu(rpre)   [B, rd]   Wr [rd, mid]
 *   gpre = r We^T + be,    gate = sigmoid(gpre) [B, mid]   We [mid, rd]
 * backward: from dgate [B, mid] accumulates dWe, dbe, dWr, dbr (+=) and writes dpool [B, mid]; `scratch` holds
 * mmvqa_se_fc_bwd_scratch_floats(B, mid, rd) floats; rd <= 128. */
int mmvqa_se_fc_fwd(mmvqa_stream_t s, const float* pool, const float* Wr, const float* br, const float* We,
                    const float* be, float* rpre, float* r, float* gpre, float* gate, int B, int mid, int rd);
size_t mmvqa_se_fc_bwd_scratch_floats(int B, int mid, int rd);
int mmvqa_se_fc_bwd(mmvqa_stream_t s, const float* dgate, const float* gpre, const float* r, const float* rpre,
                    const float* pool, const float* We, const float* Wr, float* dWe, float* dbe, float* dWr, float* dbr,
                    float* dpool, float* scratch, int B, int mid, int rd);
/* out = (t * gate[n][c] + add[n][c] / HW) * act'(z*sc+sh) (gate/add nullable); BatchNorm-backward sums into stat */
int mmvqa_act_bwd_stats(mmvqa_stream_t s, const float* t, const float* gate, const float* add, const float* z,
                        const float* sc, const float* sh, const float* mean, const float* invstd, int act, float* out,
                        double* stat, long npix, int HW, int C);
/* block end: out = post(pre(z*sc+sh) + idn'), idn' = idn | iact(idn*ids+idb) | nothing */
int mmvqa_bn_act_add(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, int pre_act, const float* idn,
                     const float* ids, const float* idb, int idn_act, int post_act, float* out, long rows, int C);
/* the block end with folded coefficients: out = post(pre(bn(z)) + [bn_d](idn)); idn / fd nullable; (pre, post) in
 * {(NONE, RELU): ResNet Bottleneck = mmvqa_bn_add_relu_fold, (SILU, NONE) and (NONE, NONE): EfficientNetV2 blocks} */
int mmvqa_bn_act_add_fold(mmvqa_stream_t s, const float* z, const mmvqa_bn_fold* f3, int pre_act, const float* idn,
                          const mmvqa_bn_fold* fd, int post_act, float* out, long rows, int C);
/* torch.optim.Adam defaults over a flat buffer; g is scaled by gscale first and zeroed when zero_grad != 0 */
int mmvqa_adam(mmvqa_stream_t s, float* p, float* g, float* m, float* v, long n, double lr, double b1, double b2,
               double eps, int step, float gscale, int zero_grad);
int mmvqa_axpy(mmvqa_stream_t s, float* y, const float* x, float a, long n);
int mmvqa_colsum(mmvqa_stream_t s, const float* x, int ld, int rows, int cols, float* out);
int mmvqa_dropout(mmvqa_stream_t s, float* x, long n, float p, uint32_t seed);
/* out[n*OH*OW + oy*OW + ox] = bit (kh*KW + kw) set iff input pixel (oy*stride - pad + kh, ox*stride - pad + kw) lies
 * inside the SH x SW image: the zero-padding pattern of torch.nn.Conv2d (torchvision Bottleneck.conv2), as consumed
 * by mmvqa_gemm_desc.pixmask */
int mmvqa_pixmask(mmvqa_stream_t s, int* out, int N, int OH, int OW, int SH, int SW, int KH, int KW, int stride,
                  int pad);

/* ---- engine level: the whole Model.forward / backward (models/mmbert.py:150-167) ------------ */
int mmvqa_engine_create(const mmvqa_model_desc* desc, mmvqa_engine** out);
void mmvqa_engine_destroy(mmvqa_engine* e);
/* parameter / buffer table: names follow the reference state_dict.
 * kind: 0 = trainable fp32 (offset into params/grads), 1 = fp32 buffer (offset into bufs),
 *       2 = int64 buffer (offset into nbt).  shape is the LOGICAL torch shape; channels_last != 0
 *       means the physical order is [d0][d2][d3][d1]. */
int mmvqa_engine_num_tensors(const mmvqa_engine* e);
int mmvqa_engine_tensor_info(const mmvqa_engine* e, int i, char* name, int name_cap, int* kind, int* ndim,
                             long long shape[4], long long* offset, int* channels_last);
long long mmvqa_engine_param_floats(const mmvqa_engine* e);
long long mmvqa_engine_buf_floats(const mmvqa_engine* e);
long long mmvqa_engine_nbt_count(const mmvqa_engine* e);
/* plan for a batch geometry; returns workspace bytes needed (0 on error) */
size_t mmvqa_engine_plan(mmvqa_engine* e, int B, int T, int img_h, int img_w);
/* Borrowed buffers of the following forward / backward calls.  Part of the WORKSPACE holds state that must survive
 * between calls (tap-validity tables of the 3x3 weight gradients, the zeroed arrival tickets of persistent GEMM
 * launches): the first forward after a bind puts it in place on that call's stream; the caller must not write to the
 * workspace while it is bound (re-bind after any such write). */
int mmvqa_engine_bind(mmvqa_engine* e, float* params, float* grads, float* bufs, long long* nbt, void* workspace,
                      size_t workspace_bytes);
/* img fp32 NCHW [B,3,h,w]; ids/seg/mask int64 [B,T]; logits [rows][ld] (rows = B*T or B);
 * feat [B][feat_dim] or NULL.  training != 0: batch-stat BN + dropout with `seed`. */
int mmvqa_engine_forward(mmvqa_engine* e, mmvqa_stream_t s, const float* img, const long long* ids,
                         const long long* seg, const long long* mask, float* logits, int logits_ld, float* feat,
                         int training, uint32_t seed);
/* accumulates into grads; dlogits same layout as logits; dfeat nullable */
int mmvqa_engine_backward(mmvqa_engine* e, mmvqa_stream_t s, const float* dlogits, int dlogits_ld,
                          const float* dfeat);
/* data-parallel overlap: cb(user, lo, hi) is called on the HOST thread inside mmvqa_engine_backward as soon as
 * the kernels completing grads[lo, hi) (float offsets) are enqueued and the given stream is ordered behind them, so
 * the caller can start the all-reduce of that range on its communication stream while backward continues.
 * The ranges of one backward partition [0, param_floats) exactly.  NULL disables. */
int mmvqa_engine_set_grad_callback(mmvqa_engine* e, mmvqa_grad_cb cb, void* user);
/* per-shape kernel configuration: while enabled, every implicit-GEMM shape met for the first time in
 * forward/backward is timed over its candidate (tile, split-K) set and the fastest is kept for later
 * calls.  A pass run with tuning enabled is a throw-away pass (outputs/statistics are garbage).
 * Returns the number of tuned shapes so far (>= 0) or a negative error.  enable = 2 changes nothing and returns how many
 * of the tuned shapes run in the persistent form (mmvqa_gemm_desc.persist). */
int mmvqa_engine_tune(mmvqa_engine* e, int enable);
/* per-kernel-class timing (HIP events on the launch stream) of the NEXT forward+backward:
 * enable, run, then read back {n_launches, total_ms, algorithmic_flops} per class.
 * enable = 1: as the step normally runs (weight-gradient GEMMs on the second stream beside the data-gradient chain, so
 * a launch's time includes sharing the chip); enable = 2: everything on one stream; 0: off (both streams again) */
int mmvqa_engine_profile(mmvqa_engine* e, int enable);
int mmvqa_engine_profile_read(mmvqa_engine* e, int cls, long long* launches, double* ms, double* flops);
/* the same split by region of the step (SURVEY 8(d) per-block figures): 0 CNN backbone, 1 tap 1x1 convs
 * (image_encoding.py:53-62), 2 fused QKV / kqv projection (transformer.py:13-20, realformer.py:32-33) fwd+dgrad+wgrad,
 * 3 attention core (transformer.py:22-27, realformer.py:34-44), 4 rest of the encoder, 5 heads (mmbert.py:133-137,
 * 154-166), 6 embeddings, 7 BatchNorm coefficient kernels, 8 the fused QKV projection + attention launch (class 1, its
 * FLOPs = projection + attention) */
int mmvqa_engine_profile_read_region(mmvqa_engine* e, int region, int cls, long long* launches, double* ms,
                                     double* flops);
/* the HBM-bound kernels of the profiled step one by one: launches, total ms and ALGORITHMIC bytes (every tensor the
 * kernel has to read or write, once, fp32) -- bytes / ms against the 8 TB/s roofline.  kernel: 1 bn_add_relu (block end),
 * 2 / 3 maxpool fwd / bwd, 4 / 5 layernorm fwd / bwd, 6 dropout_copy, 7 bn_act_add, 8 / 9 / 10 dwconv fwd / bwd_data /
 * bwd_weight, 11 se_pool, 12 se_dgate, 13 act_bwd_stats, 14 / 15 tap_thin fwd / bwd */
int mmvqa_engine_profile_read_hbm(mmvqa_engine* e, int kernel, long long* launches, double* ms, double* bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Device input pipeline: the torchvision/PIL transforms of pretrain/roco_train.py:98-112 and
 * vqamed2019/train.py:179-200 on decoded uint8 RGB images in HBM, bit-exact with Pillow's arithmetic.
 * One job = PIL crop(box) -> resize((rw, rh), BILINEAR) -> keep the out_w x out_h window at (ox, oy):
 *   Resize(224)+CenterCrop(224): box = whole image, (rw, rh) = resized size, (ox, oy) = crop offset
 *   RandomResizedCrop          : box = sampled crop, (rw, rh) = (224, 224), (ox, oy) = (0, 0)
 * All pointers are device pointers; coefficient tables come from mmvqa_resample_coeffs (host) copied to the device. */
typedef struct mmvqa_resample_job {
  const unsigned char* src; int sh, sw, spitch;   /* source image, HWC uint8, row pitch in bytes */
  int bx, by, bw, bh;                             /* box of the source that is resized */
  int rw, rh;                                     /* size the box is resized to */
  int ox, oy;                                     /* window of the resized image that is produced */
  int ty0, tyn;                                   /* box rows [ty0, ty0+tyn) the vertical pass of the window reads */
  unsigned char* tmp;                             /* scratch [tyn][out_w][3] */
  unsigned char* dst; int dpitch;                 /* output [out_h][out_w][3] */
  const int* hb; const int* hk; int hks;          /* horizontal bounds [rw][2] / coefficients [rw][hks] */
  const int* vb; const int* vk; int vks;          /* vertical   bounds [rh][2] / coefficients [rh][vks] */
} mmvqa_resample_job;
size_t mmvqa_sizeof_resample_job(void);
/* HOST function: Pillow's bilinear resampling coefficients (Resample.c precompute_coeffs + normalize_coeffs_8bpc) of
 * resizing source range [in0, in1) of an axis of in_size pixels to out_size; returns taps per output (ksize);
 * with bounds == NULL only the size is returned */
int mmvqa_resample_coeffs(int in_size, double in0, double in1, int out_size, int* bounds, int* kk, int ksize_cap);
int mmvqa_aug_resample(mmvqa_stream_t s, const mmvqa_resample_job* jobs_dev, int njobs, int max_tmp_rows, int out_h,
                       int out_w);
/* Image.rotate(angle, NEAREST, expand=False, fillcolor=0) on [B][H][W][3]: fix_dev[b][6] = the 16.16 fixed-point
 * affine coefficients a0..a5 of Geometry.c affine_fixed (computed by the host side from the angle) */
int mmvqa_aug_rotate(mmvqa_stream_t s, const unsigned char* src, unsigned char* dst, const int* fix_dev, int B, int H,
                     int W);
/* one ColorJitter round in place: image b applies op_dev[b] (0 brightness, 1 contrast, 2 saturation, 3 hue with
 * factor = the uint8 hue shift, < 0 none) with factor_dev[b]; lsum_dev = B uint64 of scratch (L sums for contrast) */
int mmvqa_aug_jitter_round(mmvqa_stream_t s, unsigned char* img, const int* op_dev, const float* factor_dev,
                           unsigned long long* lsum_dev, int B, int npix);
/* ToTensor + Normalize: uint8 [B][npix][3] -> fp32 [B][3][npix], (x/255 - mean)/std; mean3/std3 are HOST pointers */
int mmvqa_aug_to_tensor(mmvqa_stream_t s, const unsigned char* img, float* out, int B, int npix, const float* mean3,
                        const float* std3);

#ifdef __cplusplus
}
#endif
#endif /* MMVQA_H */
