// extern "C" surface of libmmvqa_hip.so (declared in include/mmvqa.h).
#include <cstring>

#include "engine.h"

const char* mmvqa_get_error();
size_t engine_plan(mmvqa_engine* e, int B, int T, int IH, int IW);
int engine_forward(mmvqa_engine* e, hipStream_t st, const float* img, const long long* ids, const long long* seg,
                   const long long* mask, float* logits, int logits_ld, float* feat, int training, uint32_t seed);
int engine_backward(mmvqa_engine* e, hipStream_t st, const float* dlogits, int dl_ld, const float* dfeat);
int engine_create(const mmvqa_model_desc* desc, mmvqa_engine** out);
int engine_profile_collect(mmvqa_engine* e);

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int mmvqa_version(void) { return 100; }
const char* mmvqa_last_error(void) { return mmvqa_get_error(); }
size_t mmvqa_sizeof_gemm_desc(void) { return sizeof(mmvqa_gemm_desc); }
size_t mmvqa_sizeof_attn_desc(void) { return sizeof(mmvqa_attn_desc); }
size_t mmvqa_sizeof_model_desc(void) { return sizeof(mmvqa_model_desc); }

static int check_gemm(const mmvqa_gemm_desc* d, int kind, int nchw) {
  if (!d) return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: null descriptor");
  if (kind < 0 || kind > 2) return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: kind=%d", kind);
  if (!d->A || !d->B || (!d->C && d->epi_mode != EPI_TAP_FWD))
    return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: null operand");
  if (!nchw) {
    const int taps = d->g_KH * d->g_KW;
    if ((d->a_ld & 3) || (d->b_ld & 3) || (taps > 1 && (d->g_Cs & 3)))
      return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: a_ld=%d b_ld=%d g_Cs=%d must be multiples of 4", d->a_ld,
                             d->b_ld, d->g_Cs);
    if (kind != KIND_FWD && (d->N & 3) && kind == KIND_DGRAD)
      return mmvqa_set_error(MMVQA_ERR_ARG, "igemm dgrad: N=%d must be a multiple of 4", d->N);
  }
  if (d->a_pro == PRO_DZ && (!d->A2 || (!d->a_fold.stat && (!d->a_c0 || !d->a_c1 || !d->a_c2))))
    return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: PRO_DZ needs A2 and three coefficient arrays (or a_fold)");
  if (d->g_SH <= 0 || d->g_SW <= 0 || d->g_OH <= 0 || d->g_OW <= 0 || d->g_Cs <= 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "igemm: gather geometry not set");
  return MMVQA_OK;
}

int mmvqa_igemm(const mmvqa_gemm_desc* d, int kind, int nchw, int tile, mmvqa_stream_t s) {
  int r = check_gemm(d, kind, nchw);
  if (r != MMVQA_OK) return r;
  return mmvqa_launch_igemm(*d, kind, nchw, tile, ST(s));
}

int mmvqa_attention(const mmvqa_attn_desc* d, int head_dim, int backward, mmvqa_stream_t s) {
  if (!d || !d->q || !d->k || !d->v || !d->mask || !d->probs)
    return mmvqa_set_error(MMVQA_ERR_ARG, "attention: null operand");
  if ((d->row_stride & 3) || (d->head_stride & 3) || (d->out_row_stride & 3) || (d->out_head_stride & 3))
    return mmvqa_set_error(MMVQA_ERR_ARG, "attention: strides must be multiples of 4");
  if (backward && (!d->dout || !d->dq || !d->dk || !d->dv))
    return mmvqa_set_error(MMVQA_ERR_ARG, "attention backward: null gradient pointer");
  if (!backward && !d->out) return mmvqa_set_error(MMVQA_ERR_ARG, "attention: null output");
  return mmvqa_launch_attention(*d, head_dim, backward, ST(s));
}

int mmvqa_qkv_attention_fwd(mmvqa_stream_t s, const float* xn, const float* W, const float* bias, const long long* mask,
                            float* qkv, float* probs, float* ctx, int B, int T, int hidden, int heads, float drop_p,
                            uint32_t seed) {
  if (B <= 0 || drop_p < 0.f || drop_p >= 1.f) return mmvqa_set_error(MMVQA_ERR_ARG, "qkv_attention_fwd: B=%d drop_p=%g", B, (double)drop_p);
  return k_qkv_attn_fwd(ST(s), xn, W, bias, mask, qkv, probs, ctx, B, T, hidden, heads, drop_p, seed);
}
int mmvqa_bn_coef_fwd(mmvqa_stream_t s, const double* stat, int C, double count, float eps, const float* gamma,
                      const float* beta, float* run_mean, float* run_var, long long* nbt, float momentum, int reps,
                      int training, float* scale, float* shift, float* mean, float* invstd) {
  return k_bn_coef_fwd(ST(s), stat, C, count, eps, gamma, beta, run_mean, run_var, nbt, momentum, reps, training,
                       scale, shift, mean, invstd);
}
int mmvqa_bn_coef_bwd(mmvqa_stream_t s, const double* stat, int C, double count, const float* gamma,
                      const float* mean, const float* invstd, int training, float* P, float* Q, float* R,
                      float* dgamma, float* dbeta) {
  return k_bn_coef_bwd(ST(s), stat, C, count, gamma, mean, invstd, training, P, Q, R, dgamma, dbeta);
}
int mmvqa_bn_add_relu_fold(mmvqa_stream_t s, const float* z, const mmvqa_bn_fold* f3, const float* idn,
                           const mmvqa_bn_fold* fd, float* out, long rows, int C) {
  if (!z || !idn || !out || !f3) return mmvqa_set_error(MMVQA_ERR_ARG, "bn_add_relu_fold: null operand");
  return k_bn_add_relu_fold(ST(s), z, f3, idn, fd, out, rows, C);
}
int mmvqa_bn_add_relu(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, const float* idn,
                      const float* id_sc, const float* id_sh, float* out, long rows, int C) {
  if (C & 3) return mmvqa_set_error(MMVQA_ERR_ARG, "bn_add_relu: C %% 4 != 0");
  return k_bn_add_relu(ST(s), z, sc, sh, idn, id_sc, id_sh, out, rows, C);
}
int mmvqa_maxpool_fwd(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, float* out,
                      unsigned char* idx, int N, int H, int W, int C, int OH, int OW) {
  if (C & 3) return mmvqa_set_error(MMVQA_ERR_ARG, "maxpool: C %% 4 != 0");
  return k_maxpool_fwd(ST(s), z, sc, sh, out, idx, N, H, W, C, OH, OW);
}
int mmvqa_maxpool_bwd(mmvqa_stream_t s, const float* gp, const unsigned char* idx, const float* extra,
                      const float* z, const float* sc, const float* sh, const float* mean, const float* invstd,
                      float* g0, double* stat, int N, int H, int W, int C, int OH, int OW) {
  return k_maxpool_bwd(ST(s), gp, idx, extra, z, sc, sh, mean, invstd, g0, stat, N, H, W, C, OH, OW);
}
int mmvqa_layernorm_fwd(mmvqa_stream_t s, const float* x, const float* res, const float* gamma, const float* beta,
                        float* y, float* sum_out, float* mean, float* rstd, int rows, int H, float eps) {
  return k_layernorm_fwd(ST(s), x, res, gamma, beta, y, sum_out, mean, rstd, rows, H, eps);
}
int mmvqa_layernorm_bwd(mmvqa_stream_t s, const float* dy, const float* x, const float* gamma, const float* mean,
                        const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta, int rows,
                        int H) {
  return k_layernorm_bwd(ST(s), dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, H);
}
int mmvqa_embed_fwd(mmvqa_stream_t s, const long long* ids, const long long* seg, const float* word,
                    const float* pos, const float* type, const float* gamma, const float* beta, const float* vis,
                    float* out, float* xhat, float* rstd, int B, int T, int H, int num_vis, float eps,
                    float drop_p, uint32_t seed) {
  return k_embed_fwd(ST(s), ids, seg, word, pos, type, gamma, beta, vis, out, xhat, rstd, B, T, H, num_vis, eps,
                     drop_p, seed);
}
int mmvqa_embed_bwd(mmvqa_stream_t s, const float* dout, const long long* ids, const long long* seg,
                    const float* xhat, const float* rstd, const float* gamma, float* dword, float* dpos,
                    float* dtype, float* dgamma, float* dbeta, float* dvis, int B, int T, int H, int num_vis,
                    float drop_p, uint32_t seed, int pad_idx) {
  return k_embed_bwd(ST(s), dout, ids, seg, xhat, rstd, gamma, dword, dpos, dtype, dgamma, dbeta, dvis, B, T, H,
                     num_vis, drop_p, seed, pad_idx);
}
int mmvqa_meanpool_fwd(mmvqa_stream_t s, const float* h, const long long* mask, float* out, int B, int T, int H) {
  return k_meanpool_fwd(ST(s), h, mask, out, B, T, H);
}
int mmvqa_meanpool_bwd(mmvqa_stream_t s, const float* dout, const long long* mask, float* dh, int B, int T, int H,
                       int accumulate) {
  return k_meanpool_bwd(ST(s), dout, mask, dh, B, T, H, accumulate);
}
int mmvqa_mlm_loss(mmvqa_stream_t s, const float* logits, int ld, const long long* target, float* row_loss,
                   float* row_lse, long long* pred, float* dlogits, int dld, const float* gscale_ptr, float gscale_mul,
                   int rows, int V, float* out3) {
  return k_lsm_nll(ST(s), logits, ld, target, row_loss, row_lse, pred, dlogits, dld, gscale_ptr, gscale_mul, rows, V, out3);
}
int mmvqa_mlm_grad(mmvqa_stream_t s, const float* logits, int ld, const long long* target, const float* row_lse,
                   float* dlogits, int dld, const float* gscale_ptr, float gscale_mul, int rows, int V) {
  return k_lsm_grad(ST(s), logits, ld, target, row_lse, dlogits, dld, gscale_ptr, gscale_mul, rows, V);
}
int mmvqa_asl_loss(mmvqa_stream_t s, const float* logits, int ld, const long long* target, float* row_loss,
                   float* dlogits, int dld, int rows, int C, float gamma_pos, float gamma_neg, float eps,
                   float gscale) {
  return k_asl(ST(s), logits, ld, target, row_loss, dlogits, dld, rows, C, gamma_pos, gamma_neg, eps, gscale);
}
int mmvqa_l2norm_fwd(mmvqa_stream_t s, const float* x, float* y, float* nrm, int rows, int D) {
  return k_l2norm_fwd(ST(s), x, y, nrm, rows, D);
}
int mmvqa_l2norm_bwd(mmvqa_stream_t s, const float* dy, const float* y, const float* nrm, float* dx, int rows,
                     int D) {
  return k_l2norm_bwd(ST(s), dy, y, nrm, dx, rows, D);
}
int mmvqa_supcon_loss(mmvqa_stream_t s, const float* f, float* loss, float* df, float* ws, int N, int D, float temp,
                      float base_temp, float gscale) {
  return k_supcon(ST(s), f, loss, df, ws, N, D, temp, base_temp, gscale);
}
int mmvqa_dwconv_fwd(mmvqa_stream_t s, const float* z1, const float* s1, const float* b1, const float* w, float* z2,
                     double* stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad) {
  return k_dwconv_fwd(ST(s), z1, s1, b1, w, z2, stat, N, H, W, C, OH, OW, stride, pad);
}
int mmvqa_dwconv_bwd_data(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                          const float* R, const float* w, const float* z1, const float* s1, const float* b1,
                          const float* mean1, const float* invstd1, float* g1, double* stat, int N, int H, int W, int C,
                          int OH, int OW, int stride, int pad) {
  return k_dwconv_bwd_data(ST(s), g2, z2, P, Q, R, w, z1, s1, b1, mean1, invstd1, g1, stat, N, H, W, C, OH, OW, stride, pad);
}
int mmvqa_dwconv_bwd_weight(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                            const float* R, const float* z1, const float* s1, const float* b1, float* dw, int N, int H,
                            int W, int C, int OH, int OW, int stride, int pad) {
  return k_dwconv_bwd_weight(ST(s), g2, z2, P, Q, R, z1, s1, b1, dw, N, H, W, C, OH, OW, stride, pad);
}
int mmvqa_se_pool(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, float* pool, int N, int HW, int C) {
  return k_se_pool(ST(s), z, sc, sh, pool, N, HW, C);
}
int mmvqa_dwconv_fwd_fold(mmvqa_stream_t s, const float* z1, const float* s1, const float* b1, const float* w, float* z2,
                          double* stat, int N, int H, int W, int C, int OH, int OW, int stride, int pad,
                          const mmvqa_bn_fold* fold) {
  return k_dwconv_fwd(ST(s), z1, s1, b1, w, z2, stat, N, H, W, C, OH, OW, stride, pad, fold);
}
int mmvqa_dwconv_bwd_data_fold(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                               const float* R, const float* w, const float* z1, const float* s1, const float* b1,
                               const float* mean1, const float* invstd1, float* g1, double* stat, int N, int H, int W,
                               int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold) {
  return k_dwconv_bwd_data(ST(s), g2, z2, P, Q, R, w, z1, s1, b1, mean1, invstd1, g1, stat, N, H, W, C, OH, OW, stride, pad, fold);
}
int mmvqa_dwconv_bwd_weight_fold(mmvqa_stream_t s, const float* g2, const float* z2, const float* P, const float* Q,
                                 const float* R, const float* z1, const float* s1, const float* b1, float* dw, int N,
                                 int H, int W, int C, int OH, int OW, int stride, int pad, const mmvqa_bn_fold* fold) {
  return k_dwconv_bwd_weight(ST(s), g2, z2, P, Q, R, z1, s1, b1, dw, N, H, W, C, OH, OW, stride, pad, fold);
}
int mmvqa_se_pool_fold(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, float* pool, int N, int HW,
                       int C, const mmvqa_bn_fold* fold) {
  return k_se_pool(ST(s), z, sc, sh, pool, N, HW, C, fold);
}
int mmvqa_bn_act_add_fold(mmvqa_stream_t s, const float* z, const mmvqa_bn_fold* f3, int pre_act, const float* idn,
                          const mmvqa_bn_fold* fd, int post_act, float* out, long rows, int C) {
  return k_bn_act_add_fold(ST(s), z, f3, pre_act, idn, fd, post_act, out, rows, C);
}
int mmvqa_se_dgate(mmvqa_stream_t s, const float* t, const float* z, const float* sc, const float* sh, float* dgate,
                   int N, int HW, int C) {
  return k_se_dgate(ST(s), t, z, sc, sh, dgate, N, HW, C);
}
int mmvqa_tap_thin_ok(long M, int N, int C, int HW) { return k_tap_thin_ok(M, N, C, HW) ? 1 : 0; }
int mmvqa_tap_thin_fwd(mmvqa_stream_t s, const float* x, const float* sc, const float* sh, const float* W, float* out,
                       long M, int N, int C, int HW, int act) {
  return k_tap_thin_fwd(ST(s), x, sc, sh, W, out, M, N, C, HW, act);
}
int mmvqa_tap_thin_bwd(mmvqa_stream_t s, const float* x, const float* sc, const float* sh, const float* W,
                       const float* dv, float* du, long M, int N, int C, int HW, int act) {
  return k_tap_thin_bwd(ST(s), x, sc, sh, W, dv, du, M, N, C, HW, act);
}
int mmvqa_se_fc_fwd(mmvqa_stream_t s, const float* pool, const float* Wr, const float* br, const float* We,
                    const float* be, float* rpre, float* r, float* gpre, float* gate, int B, int mid, int rd) {
  int rc = k_skinny_fwd(ST(s), pool, mid, Wr, br, ACT_SILU, rpre, r, B, rd, mid);
  if (rc != MMVQA_OK) return rc;
  return k_skinny_fwd(ST(s), r, rd, We, be, ACT_SIGMOID, gpre, gate, B, mid, rd);
}
size_t mmvqa_se_fc_bwd_scratch_floats(int B, int mid, int rd) { return k_se_fc_bwd_scratch_floats(B, mid, rd); }
int mmvqa_se_fc_bwd(mmvqa_stream_t s, const float* dgate, const float* gpre, const float* r, const float* rpre,
                    const float* pool, const float* We, const float* Wr, float* dWe, float* dbe, float* dWr, float* dbr,
                    float* dpool, float* scratch, int B, int mid, int rd) {
  return k_se_fc_bwd(ST(s), dgate, gpre, r, rpre, pool, We, Wr, dWe, dbe, dWr, dbr, dpool, scratch, 0, B, mid, rd);
}
int mmvqa_act_bwd_stats(mmvqa_stream_t s, const float* t, const float* gate, const float* add, const float* z,
                        const float* sc, const float* sh, const float* mean, const float* invstd, int act, float* out,
                        double* stat, long npix, int HW, int C) {
  return k_act_bwd_stats(ST(s), t, gate, add, z, sc, sh, mean, invstd, act, out, stat, npix, HW, C);
}
int mmvqa_bn_act_add(mmvqa_stream_t s, const float* z, const float* sc, const float* sh, int pre_act, const float* idn,
                     const float* ids, const float* idb, int idn_act, int post_act, float* out, long rows, int C) {
  return k_bn_act_add(ST(s), z, sc, sh, pre_act, idn, ids, idb, idn_act, post_act, out, rows, C);
}
int mmvqa_adam(mmvqa_stream_t s, float* p, float* g, float* m, float* v, long n, double lr, double b1, double b2,
               double eps, int step, float gscale, int zero_grad) {
  return k_adam(ST(s), p, g, m, v, n, lr, b1, b2, eps, step, gscale, zero_grad);
}
int mmvqa_axpy(mmvqa_stream_t s, float* y, const float* x, float a, long n) { return k_axpy(ST(s), y, x, a, n); }
int mmvqa_colsum(mmvqa_stream_t s, const float* x, int ld, int rows, int cols, float* out) {
  return k_colsum(ST(s), x, ld, rows, cols, out);
}
int mmvqa_dropout(mmvqa_stream_t s, float* x, long n, float p, uint32_t seed) {
  return k_dropout(ST(s), x, n, p, seed);
}

int mmvqa_pixmask(mmvqa_stream_t s, int* out, int N, int OH, int OW, int SH, int SW, int KH, int KW, int stride,
                  int pad) {
  if (!out) return mmvqa_set_error(MMVQA_ERR_ARG, "pixmask: null output");
  return k_pixmask(ST(s), out, N, OH, OW, SH, SW, KH, KW, stride, pad);
}

// ---------------------------------------------------------------------------------- engine
int mmvqa_engine_create(const mmvqa_model_desc* desc, mmvqa_engine** out) { return engine_create(desc, out); }
void mmvqa_engine_destroy(mmvqa_engine* e) {
  if (!e) return;
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->side2) (void)hipStreamDestroy(e->side2);
  delete e;
}
int mmvqa_engine_num_tensors(const mmvqa_engine* e) { return e ? (int)e->specs.size() : 0; }
int mmvqa_engine_tensor_info(const mmvqa_engine* e, int i, char* name, int name_cap, int* kind, int* ndim,
                             long long shape[4], long long* offset, int* channels_last) {
  if (!e || i < 0 || i >= (int)e->specs.size()) return mmvqa_set_error(MMVQA_ERR_ARG, "tensor_info: bad index");
  const TensorSpec& s = e->specs[i];
  if (name && name_cap > 0) {
    strncpy(name, s.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (kind) *kind = s.kind;
  if (ndim) *ndim = s.ndim;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = s.shape[k];
  if (offset) *offset = s.offset;
  if (channels_last) *channels_last = s.channels_last;
  return MMVQA_OK;
}
long long mmvqa_engine_param_floats(const mmvqa_engine* e) { return e ? e->n_params : 0; }
long long mmvqa_engine_buf_floats(const mmvqa_engine* e) { return e ? e->n_bufs : 0; }
long long mmvqa_engine_nbt_count(const mmvqa_engine* e) { return e ? e->n_nbt : 0; }
size_t mmvqa_engine_plan(mmvqa_engine* e, int B, int T, int img_h, int img_w) {
  if (!e) { mmvqa_set_error(MMVQA_ERR_ARG, "plan: null engine"); return 0; }
  return engine_plan(e, B, T, img_h, img_w);
}
int mmvqa_engine_bind(mmvqa_engine* e, float* params, float* grads, float* bufs, long long* nbt, void* workspace,
                      size_t workspace_bytes) {
  if (!e || !e->planned) return mmvqa_set_error(MMVQA_ERR_STATE, "bind: plan first");
  if (!params || !grads || !bufs || !nbt || !workspace) return mmvqa_set_error(MMVQA_ERR_ARG, "bind: null pointer");
  if (workspace_bytes < e->ws_floats * sizeof(float))
    return mmvqa_set_error(MMVQA_ERR_ARG, "bind: workspace too small (%zu < %zu)", workspace_bytes,
                           e->ws_floats * sizeof(float));
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)bufs | (uintptr_t)workspace) & 15)
    return mmvqa_set_error(MMVQA_ERR_ARG, "bind: buffers must be 16-byte aligned");
  e->params = params; e->grads = grads; e->bufs = bufs; e->nbt = nbt;
  e->ws = reinterpret_cast<float*>(workspace);
  e->ws_ready = false;        // tap-validity tables and tickets live in the (new) workspace: put in place by the next forward
  e->bound = true;
  e->img = nullptr;
  return MMVQA_OK;
}
int mmvqa_engine_forward(mmvqa_engine* e, mmvqa_stream_t s, const float* img, const long long* ids,
                         const long long* seg, const long long* mask, float* logits, int logits_ld, float* feat,
                         int training, uint32_t seed) {
  if (!e || !img || !ids || !seg || !mask || !logits) return mmvqa_set_error(MMVQA_ERR_ARG, "forward: null pointer");
  return engine_forward(e, ST(s), img, ids, seg, mask, logits, logits_ld, feat, training, seed);
}
int mmvqa_engine_backward(mmvqa_engine* e, mmvqa_stream_t s, const float* dlogits, int dlogits_ld,
                          const float* dfeat) {
  if (!e || !dlogits) return mmvqa_set_error(MMVQA_ERR_ARG, "backward: null pointer");
  return engine_backward(e, ST(s), dlogits, dlogits_ld, dfeat);
}
int mmvqa_engine_set_grad_callback(mmvqa_engine* e, mmvqa_grad_cb cb, void* user) {
  if (!e) return mmvqa_set_error(MMVQA_ERR_ARG, "set_grad_callback: null engine");
  e->grad_cb = cb;
  e->grad_cb_user = user;
  return MMVQA_OK;
}
int mmvqa_engine_tune(mmvqa_engine* e, int enable) {
  if (!e) return mmvqa_set_error(MMVQA_ERR_ARG, "tune: null engine");
  if (enable == 2) {   // query: how many shapes run in the persistent form
    int n = 0;
    for (const auto& kv : e->tuner.table) n += kv.second.persist > 0;
    return n;
  }
  e->tuner.tuning = enable != 0;
  return (int)e->tuner.table.size();
}
int mmvqa_engine_profile(mmvqa_engine* e, int enable) {
  if (!e) return mmvqa_set_error(MMVQA_ERR_ARG, "profile: null engine");
  e->prof_on = enable ? 1 : 0;
  e->use_side = enable == 2 ? 0 : 1;   // 2: one stream only, every launch has the chip to itself
  if (enable) {
    memset(e->prof_launch, 0, sizeof(e->prof_launch));
    memset(e->prof_ms, 0, sizeof(e->prof_ms));
    memset(e->prof_flops, 0, sizeof(e->prof_flops));
    memset(e->reg_launch, 0, sizeof(e->reg_launch));
    memset(e->reg_ms, 0, sizeof(e->reg_ms));
    memset(e->reg_flops, 0, sizeof(e->reg_flops));
    memset(e->tag_launch, 0, sizeof(e->tag_launch));
    memset(e->tag_ms, 0, sizeof(e->tag_ms));
    memset(e->tag_bytes, 0, sizeof(e->tag_bytes));
  }
  return MMVQA_OK;
}
int mmvqa_engine_profile_read_region(mmvqa_engine* e, int region, int cls, long long* launches, double* ms,
                                     double* flops) {
  if (!e || cls < 0 || cls >= PROF_NCLS || region < 0 || region >= REG_N)
    return mmvqa_set_error(MMVQA_ERR_ARG, "profile_read_region: bad region/class");
  int r = engine_profile_collect(e);
  if (r != MMVQA_OK) return r;
  if (launches) *launches = e->reg_launch[region][cls];
  if (ms) *ms = e->reg_ms[region][cls];
  if (flops) *flops = e->reg_flops[region][cls];
  return MMVQA_OK;
}
int mmvqa_engine_profile_read_hbm(mmvqa_engine* e, int kernel, long long* launches, double* ms, double* bytes) {
  if (!e || kernel <= HB_NONE || kernel >= HB_N) return mmvqa_set_error(MMVQA_ERR_ARG, "profile_read_hbm: bad kernel id");
  int r = engine_profile_collect(e);
  if (r != MMVQA_OK) return r;
  if (launches) *launches = e->tag_launch[kernel];
  if (ms) *ms = e->tag_ms[kernel];
  if (bytes) *bytes = e->tag_bytes[kernel];
  return MMVQA_OK;
}
int mmvqa_engine_profile_read(mmvqa_engine* e, int cls, long long* launches, double* ms, double* flops) {
  if (!e || cls < 0 || cls >= PROF_NCLS) return mmvqa_set_error(MMVQA_ERR_ARG, "profile_read: bad class");
  int r = engine_profile_collect(e);
  if (r != MMVQA_OK) return r;
  if (launches) *launches = e->prof_launch[cls];
  if (ms) *ms = e->prof_ms[cls];
  if (flops) *flops = e->prof_flops[cls];
  return MMVQA_OK;
}

}  // extern "C"
