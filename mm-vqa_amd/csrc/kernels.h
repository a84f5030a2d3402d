// Host-side launchers of the HIP kernels (defined in igemm.hip / attention.hip / elementwise.hip).
#pragma once
#include "common.h"

#include <string>
#include <unordered_map>
#include <utility>

// per-shape (tile, split-K) choices of the implicit-GEMM launcher; see igemm.hip
struct IgemmChoice { int tile, splitk, persist; };   // persist: workgroups of the persistent ("stream-K") form, 0 = off
struct IgemmTuner {
  bool tuning = false;
  std::unordered_map<std::string, IgemmChoice> table;
};
void mmvqa_set_tuner(IgemmTuner* t);
int mmvqa_launch_igemm(GemmParams p, int kind, int nchw, int tile, hipStream_t stream);
int mmvqa_launch_attention(const AttnParams& p, int head_dim, int bwd, hipStream_t st);
// qkvattn.hip: fused QKV projection + self-attention of a BertLayer (forward), T <= 32, head dimension 64
bool k_qkv_attn_fwd_ok(int T, int H, int heads);
int k_qkv_attn_fwd(hipStream_t st, const float* xn, const float* W, const float* bias, const long long* mask, float* qkv,
                   float* probs, float* ctx, int B, int T, int H, int heads, float drop_p, uint32_t seed);

int k_bn_coef_fwd(hipStream_t st, const double* stat, int C, double count, float eps, const float* gamma,
                  const float* beta, float* run_mean, float* run_var, long long* nbt, float momentum, int reps,
                  int training, float* scale, float* shift, float* mean, float* invstd);
// same with (1 - momentum)^reps given directly (mmvqa_bn_fold carries it that way)
int k_bn_coef_fwd_keep(hipStream_t st, const double* stat, int C, double count, float eps, const float* gamma,
                       const float* beta, float* run_mean, float* run_var, long long* nbt, double keep, int reps,
                       int training, float* scale, float* shift, float* mean, float* invstd);
int k_bn_coef_bwd(hipStream_t st, const double* stat, int C, double count, const float* gamma, const float* mean,
                  const float* invstd, int training, float* P, float* Q, float* R, float* dgamma, float* dbeta);
int k_bn_add_relu(hipStream_t st, const float* z, const float* s, const float* b, const float* idn,
                  const float* ids, const float* idb, float* out, long rows, int C);
// relu(bn3(z) + [bnd](idn)) with the BatchNorm coefficients folded from raw sums inside the kernel (fd == NULL: plain identity)
// out = post(pre(bn(z)) + [bn_d](idn)) with the BatchNorm coefficients folded from raw sums in the kernel (idn may be null)
int k_bn_act_add_fold(hipStream_t st, const float* z, const mmvqa_bn_fold* f3, int pre_act, const float* idn,
                      const mmvqa_bn_fold* fd, int post_act, float* out, long rows, int C);
int k_bn_add_relu_fold(hipStream_t st, const float* z, const mmvqa_bn_fold* f3, const float* idn, const mmvqa_bn_fold* fd,
                       float* out, long rows, int C);
int k_maxpool_fwd(hipStream_t st, const float* z, const float* s, const float* b, float* out, unsigned char* idx,
                  int N, int H, int W, int C, int OH, int OW);
int k_maxpool_bwd(hipStream_t st, const float* gp, const unsigned char* idx, const float* extra, const float* z,
                  const float* s, const float* b, const float* mean, const float* invstd, float* g0, double* stat,
                  int N, int H, int W, int C, int OH, int OW);
int k_layernorm_fwd(hipStream_t st, const float* x, const float* res, const float* gamma, const float* beta,
                    float* y, float* sum_out, float* mean, float* rstd, int rows, int H, float eps);
int k_layernorm_bwd(hipStream_t st, const float* dy, const float* x, const float* gamma, const float* mean,
                    const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta, int rows, int H);
int k_embed_fwd(hipStream_t st, const long long* ids, const long long* seg, const float* word, const float* pos,
                const float* type, const float* gamma, const float* beta, const float* vis, float* out,
                float* xhat, float* rstd, int B, int T, int H, int num_vis, float eps, float drop_p, uint32_t seed);
int k_embed_bwd(hipStream_t st, const float* dout, const long long* ids, const long long* seg, const float* xhat,
                const float* rstd, const float* gamma, float* dword, float* dpos, float* dtype, float* dgamma,
                float* dbeta, float* dvis, int B, int T, int H, int num_vis, float drop_p, uint32_t seed,
                int pad_idx);
int k_meanpool_fwd(hipStream_t st, const float* h, const long long* mask, float* out, int B, int T, int H);
int k_meanpool_bwd(hipStream_t st, const float* dout, const long long* mask, float* dh, int B, int T, int H,
                   int accumulate);
int k_lsm_nll(hipStream_t st, const float* logits, int ld, const long long* target, float* row_loss, float* row_lse,
              long long* pred, float* dlogits, int dld, const float* gscale_ptr, float gscale_mul, int rows, int V,
              float* out3);
int k_lsm_grad(hipStream_t st, const float* logits, int ld, const long long* target, const float* row_lse,
               float* dlogits, int dld, const float* gscale_ptr, float gscale_mul, int rows, int V);
int k_asl(hipStream_t st, const float* logits, int ld, const long long* target, float* row_loss, float* dlogits,
          int dld, int rows, int C, float gpos, float gneg, float eps, float gscale);
int k_l2norm_fwd(hipStream_t st, const float* x, float* y, float* nrm, int rows, int D);
int k_l2norm_bwd(hipStream_t st, const float* dy, const float* y, const float* nrm, float* dx, int rows, int D);
int k_supcon(hipStream_t st, const float* f, float* loss, float* df, float* ws, int N, int D, float temp,
             float base_temp, float gscale);
int k_adam(hipStream_t st, float* p, float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
           int step, float gscale, int zero_grad);
int k_axpy(hipStream_t st, float* y, const float* x, float a, long n);
int k_colsum(hipStream_t st, const float* x, int ld, int rows, int cols, float* out);
int k_dropout(hipStream_t st, float* x, long n, float p, uint32_t seed);
int k_pixmask(hipStream_t st, int* out, int N, int OH, int OW, int SH, int SW, int KH, int KW, int stride, int pad);
int k_dropout_copy(hipStream_t st, const float* x, float* y, long n, float p, uint32_t seed);

// EfficientNetV2 pieces (elementwise.hip)
int k_bn_act_add(hipStream_t st, const float* z, const float* s, const float* b, int pre_act, const float* idn,
                 const float* ids, const float* idb, int idn_act, int post_act, float* out, long rows, int C);
int k_dwconv_fwd(hipStream_t st, const float* z1, const float* s1, const float* b1, const float* w, float* z2,
                 double* stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold = nullptr);
int k_dwconv_bwd_data(hipStream_t st, const float* g2, const float* z2, const float* P, const float* Q,
                      const float* R, const float* w, const float* z1, const float* s1, const float* b1,
                      const float* mean1, const float* invstd1, float* g1, double* stat, int N, int H, int W, int C,
                      int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold = nullptr);
int k_dwconv_bwd_weight(hipStream_t st, const float* g2, const float* z2, const float* P, const float* Q,
                        const float* R, const float* z1, const float* s1, const float* b1, float* dw, int N, int H,
                        int W, int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold = nullptr);
int k_se_pool(hipStream_t st, const float* z, const float* s, const float* b, float* pool, int N, int HW, int C,
              const mmvqa_bn_fold* fold = nullptr);
int k_se_dgate(hipStream_t st, const float* t, const float* z, const float* s, const float* b, float* dgate, int N,
               int HW, int C, float* zero = nullptr, int nzero = 0);
int k_act_bwd_stats(hipStream_t st, const float* t, const float* gate, const float* add, const float* z,
                    const float* s, const float* b, const float* mean, const float* invstd, int act, float* out,
                    double* stat, long npix, int HW, int C);
int k_mul_dact(hipStream_t st, const float* x, const float* pre, int act, float* y, long n);
// tapthin.hip: visual-token tap of a feature map with few channels (stem taps)
bool k_tap_thin_ok(long M, int N, int C, int HW);
int k_tap_thin_fwd(hipStream_t st, const float* x, const float* sc, const float* sh, const float* W, float* out, long M,
                   int N, int C, int HW, int act);
int k_tap_thin_bwd(hipStream_t st, const float* x, const float* sc, const float* sh, const float* W, const float* dv,
                   float* du, long M, int N, int C, int HW, int act);
// se.hip: the squeeze-excite fully connected layers (16-row problems) without the GEMM tile machinery
int k_skinny_fwd(hipStream_t st, const float* x, int x_ld, const float* W, const float* b, int act, float* pre,
                 float* y, int M, int N, int K);
size_t k_se_fc_bwd_scratch_floats(int B, int mid, int rd);
bool k_se_fc_bwd_ok(int rd);   // the backward kernels keep an rd-wide tile in LDS: widths beyond their limit are refused at plan time
int k_se_fc_bwd(hipStream_t st, const float* dgate, const float* gpre, const float* r, const float* rpre,
                const float* pool, const float* We, const float* Wr, float* dWe, float* dbe, float* dWr, float* dbr,
                float* dpool, float* scratch, int scratch_is_zero, int B, int mid, int rd);
