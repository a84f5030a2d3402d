// Launch sequences of the MMBERT hot path (see engine.h).  Reference call stack: SURVEY.md 3.1-3.4.
//
//   forward : ResNet stem -> [tap] -> maxpool -> bottlenecks (+tap at every layer end)
//             -> BertEmbeddings + visual-token overwrite -> 4x BertLayer | 4x ResEncoderBlock
//             -> fc1/SERF/classifier (per token or mean-pooled) [+ SupCon head]
//   backward: the exact reverse, every BatchNorm/ReLU/residual backward folded into GEMM
//             prologues/epilogues (igemm.hip), weight gradients accumulated into `grads`.
#include "engine.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

// --------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
int mmvqa_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
const char* mmvqa_get_error() { return g_err; }

#define TRY(x)                    \
  do {                            \
    int _r = (x);                 \
    if (_r != MMVQA_OK) return _r; \
  } while (0)

// --------------------------------------------------------------------------- parameter table
namespace {

struct Builder {
  mmvqa_engine* e;
  long long align4(long long x) { return (x + 3) & ~3LL; }
  long long add(const std::string& name, int kind, std::initializer_list<long long> shape, int cl = 0) {
    TensorSpec s;
    s.name = name; s.kind = kind; s.ndim = (int)shape.size(); s.channels_last = cl;
    long long n = 1; int i = 0;
    for (auto v : shape) { s.shape[i++] = v; n *= v; }
    for (; i < 4; ++i) s.shape[i] = 1;
    long long& cur = kind == 0 ? e->n_params : (kind == 1 ? e->n_bufs : e->n_nbt);
    if (kind != 2) cur = align4(cur);
    s.offset = cur;
    cur += n;
    e->specs.push_back(s);
    return s.offset;
  }
  ConvRef conv(const std::string& name, int cin, int cout, int k, int stride, int pad) {
    ConvRef c; c.Cin = cin; c.Cout = cout; c.KH = k; c.stride = stride; c.pad = pad;
    c.w = add(name + ".weight", 0, {cout, cin, k, k}, 1);
    return c;
  }
  BNRef bn(const std::string& name, int C, int reps) {
    BNRef b; b.C = C; b.reps = reps;
    b.gamma = add(name + ".weight", 0, {C});
    b.beta = add(name + ".bias", 0, {C});
    b.rmean = add(name + ".running_mean", 1, {C});
    b.rvar = add(name + ".running_var", 1, {C});
    b.nbt = add(name + ".num_batches_tracked", 2, {});
    return b;
  }
  LinRef lin(const std::string& name, int in, int out, bool bias) {
    LinRef l; l.in = in; l.out = out;
    l.w = add(name + ".weight", 0, {out, in});
    l.b = bias ? add(name + ".bias", 0, {out}) : -1;
    return l;
  }
  LNRef ln(const std::string& name, int H) {
    LNRef l;
    l.g = add(name + ".weight", 0, {H});
    l.b = add(name + ".bias", 0, {H});
    return l;
  }
};

int build_tables(mmvqa_engine* e) {
  const mmvqa_model_desc& d = e->d;
  Builder b{e};
  const int H = d.hidden;
  // models/mmbert.py:52-56 -- HF BertEmbeddings
  const std::string be = "transformer.bert_embedding.";
  e->emb_word = b.add(be + "word_embeddings.weight", 0, {d.emb_vocab, H});
  e->emb_pos = b.add(be + "position_embeddings.weight", 0, {d.max_pos, H});
  e->emb_type = b.add(be + "token_type_embeddings.weight", 0, {d.type_vocab, H});
  e->emb_ln = b.ln(be + "LayerNorm", H);
  e->emb_hi = b.align4(e->n_params);
  // models/image_encoding.py:43-62 -- backbone + tap convs
  const std::string rm = "transformer.trans.model.";
  const int w = d.resnet_width;
  if (d.cnn == 1) {
    // timm tf_efficientnetv2_m(features_only=True) -- architecture: SURVEY.md Appendix B (wiring unpinned)
    struct St { int type, rep, stride, exp, out; bool se; };
    const St arch[7] = {{0, 3, 1, 1, 24, false}, {1, 5, 2, 4, 48, false}, {1, 5, 2, 4, 80, false}, {2, 7, 2, 4, 160, true},
                        {2, 14, 1, 6, 176, true}, {2, 18, 2, 6, 304, true}, {2, 5, 1, 6, 512, true}};
    const int feat_of_stage[7] = {0, 1, 2, -1, 3, -1, 4};
    const int div = d.effnet_depth_div > 0 ? d.effnet_depth_div : 1;
    e->stem_conv = b.conv(rm + "conv_stem", 3, 24, 3, 2, 0);
    e->stem_bn = b.bn(rm + "bn1", 24, 1);
    e->stem_bn.eps = 1e-3f;
    int cin = 24;
    for (int sI = 0; sI < 7; ++sI) {
      const St& a = arch[sI];
      const int reps = std::max(1, (a.rep + div - 1) / div);
      for (int k = 0; k < reps; ++k) {
        EffBlock blk;
        const std::string p = rm + "blocks." + std::to_string(sI) + "." + std::to_string(k) + ".";
        blk.type = a.type; blk.cin = cin; blk.cout = a.out; blk.stride = k == 0 ? a.stride : 1;
        blk.skip = blk.stride == 1 && cin == a.out;
        blk.mid = cin * a.exp;
        if (a.type == 0) {
          blk.c_a = b.conv(p + "conv", cin, a.out, 3, blk.stride, 0);
          blk.b_a = b.bn(p + "bn1", a.out, 1);
        } else if (a.type == 1) {
          blk.c_a = b.conv(p + "conv_exp", cin, blk.mid, 3, blk.stride, 0);
          blk.b_a = b.bn(p + "bn1", blk.mid, 1);
          blk.c_p = b.conv(p + "conv_pwl", blk.mid, a.out, 1, 1, 0);
          blk.b_p = b.bn(p + "bn2", a.out, 1);
        } else {
          blk.rd = (int)std::lround(cin * 0.25);
          blk.c_a = b.conv(p + "conv_pw", cin, blk.mid, 1, 1, 0);
          blk.b_a = b.bn(p + "bn1", blk.mid, 1);
          blk.dw_w = b.add(p + "conv_dw.weight", 0, {blk.mid, 1, 3, 3}, 1);
          blk.b_dw = b.bn(p + "bn2", blk.mid, 1);
          blk.se_r.in = blk.mid; blk.se_r.out = blk.rd;
          blk.se_r.w = b.add(p + "se.conv_reduce.weight", 0, {blk.rd, blk.mid, 1, 1}, 1);
          blk.se_r.b = b.add(p + "se.conv_reduce.bias", 0, {blk.rd});
          blk.se_e.in = blk.rd; blk.se_e.out = blk.mid;
          blk.se_e.w = b.add(p + "se.conv_expand.weight", 0, {blk.mid, blk.rd, 1, 1}, 1);
          blk.se_e.b = b.add(p + "se.conv_expand.bias", 0, {blk.mid});
          blk.c_p = b.conv(p + "conv_pwl", blk.mid, a.out, 1, 1, 0);
          blk.b_p = b.bn(p + "bn3", a.out, 1);
        }
        blk.b_a.eps = blk.b_dw.eps = blk.b_p.eps = 1e-3f;
        blk.feature = (k == reps - 1) ? feat_of_stage[sI] : -1;
        cin = a.out;
        e->eff.push_back(blk);
      }
    }
    const char* tapn[5] = {"conv2", "conv3", "conv4", "conv5", "conv7"};
    const int tapc[5] = {24, 48, 80, 176, 512};
    for (int k = 0; k < 5; ++k) {
      e->taps[k].C = tapc[k];
      e->taps[k].w = b.add(std::string("transformer.trans.") + tapn[k] + ".weight", 0, {H, tapc[k], 1, 1}, 1);
    }
  } else if (d.cnn == 0) {
  e->stem_conv = b.conv(rm + "conv1", 3, w, 7, 2, 3);
  e->stem_bn = b.bn(rm + "bn1", w, 5);
  int inpl = w;
  for (int l = 0; l < 4; ++l) {
    const int planes = w << l, stride = l == 0 ? 1 : 2, reps = 4 - l;
    for (int k = 0; k < d.resnet_layers[l]; ++k) {
      BlockRef blk;
      const std::string p = rm + "layer" + std::to_string(l + 1) + "." + std::to_string(k) + ".";
      const int s = k == 0 ? stride : 1;
      blk.c1 = b.conv(p + "conv1", inpl, planes, 1, 1, 0);
      blk.b1 = b.bn(p + "bn1", planes, reps);
      blk.c2 = b.conv(p + "conv2", planes, planes, 3, s, 1);
      blk.b2 = b.bn(p + "bn2", planes, reps);
      blk.c3 = b.conv(p + "conv3", planes, planes * 4, 1, 1, 0);
      blk.b3 = b.bn(p + "bn3", planes * 4, reps);
      blk.has_ds = (k == 0);
      if (blk.has_ds) {
        blk.cd = b.conv(p + "downsample.0", inpl, planes * 4, 1, s, 0);
        blk.bd = b.bn(p + "downsample.1", planes * 4, reps);
      }
      inpl = planes * 4;
      e->blocks.push_back(blk);
    }
    e->layer_end[l] = (int)e->blocks.size() - 1;
  }
  b.add(rm + "fc.weight", 0, {1000, (long long)w * 32});  // exists in the state_dict, never used (quirk 2)
  b.add(rm + "fc.bias", 0, {1000});
  const char* tapn[5] = {"conv2", "conv3", "conv4", "conv5", "conv7"};
  const int tapc[5] = {w * 32, w * 16, w * 8, w * 4, w};
  for (int k = 0; k < 5; ++k) {
    e->taps[k].C = tapc[k];
    e->taps[k].w = b.add(std::string("transformer.trans.") + tapn[k] + ".weight", 0, {H, tapc[k], 1, 1}, 1);
  }
  } else {
    return mmvqa_set_error(MMVQA_ERR_ARG, "engine: unknown cnn=%d", d.cnn);
  }
  // encoder
  e->enc_lo = b.align4(e->n_params);
  if (d.encoder == 0) {
    const std::string bl = "transformer.blocks.";
    e->norm1 = b.ln(bl + "norm1", H);
    e->norm2 = b.ln(bl + "norm2", H);
    e->bert.resize(d.n_layers);
    for (int i = 0; i < d.n_layers; ++i) {
      BertLayerRef& L = e->bert[i];
      const std::string a = bl + "attention." + std::to_string(i) + ".";
      // q,k,v weights (and biases) are laid out back to back: one fused [3H,H] projection
      L.qkv.in = H; L.qkv.out = 3 * H;
      L.qkv.w = b.add(a + "proj_q.weight", 0, {H, H});
      long long kw = b.add(a + "proj_k.weight", 0, {H, H});
      long long vw = b.add(a + "proj_v.weight", 0, {H, H});
      L.qkv.b = b.add(a + "proj_q.bias", 0, {H});
      long long kb = b.add(a + "proj_k.bias", 0, {H});
      long long vb = b.add(a + "proj_v.bias", 0, {H});
      if (kw != L.qkv.w + (long long)H * H || vw != kw + (long long)H * H || kb != L.qkv.b + H || vb != kb + H)
        return mmvqa_set_error(MMVQA_ERR_STATE, "engine: qkv parameters are not contiguous (H=%d)", H);
    }
    for (int i = 0; i < d.n_layers; ++i) e->bert[i].proj = b.lin(bl + "proj." + std::to_string(i), H, H, true);
    for (int i = 0; i < d.n_layers; ++i) {
      const std::string f = bl + "feedforward." + std::to_string(i) + ".";
      e->bert[i].fc1 = b.lin(f + "fc1", H, 4 * H, true);
      e->bert[i].fc2 = b.lin(f + "fc2", 4 * H, H, true);
    }
  } else {
    e->rf.resize(d.n_layers);
    const int es = H / 8;
    for (int i = 0; i < d.n_layers; ++i) {
      RFLayerRef& L = e->rf[i];
      const std::string m = "transformer.mains." + std::to_string(i) + ".";
      L.kqv = b.lin(m + "kqv", es, 3 * es, false);
      L.proj = b.lin(m + "proj", H, H, false);
      L.ln1 = b.ln(m + "ln1", H);
      L.ln2 = b.ln(m + "ln2", H);
      L.ff0 = b.lin(m + "ff.0", H, 4 * H, true);
      L.ff2 = b.lin(m + "ff.2", 4 * H, H, true);
    }
  }
  // heads (models/mmbert.py:133-148)
  e->fc1 = b.lin("fc1", H, H, true);
  e->cls0 = b.lin("classifier.0", H, H, true);
  e->cls_ln = b.ln("classifier.1", H);
  e->cls2 = b.lin("classifier.2", H, d.n_classes, true);
  if (d.supcon) {
    e->head0 = b.lin("head.0", H, H, true);
    e->head2 = b.lin("head.2", H, d.feat_dim, true);
  }
  e->n_params = b.align4(e->n_params);
  e->n_bufs = b.align4(e->n_bufs);
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- workspace plan
struct Arena {
  size_t cur = 0;
  size_t f(size_t n) {  // floats, 64-float (256 B) aligned
    cur = (cur + 63) & ~(size_t)63;
    size_t o = cur;
    cur += n;
    return o;
  }
};

// replicas of a folded BatchNorm's sums: every consumer workgroup reads slots x C x 16 bytes in its setup phase, every
// producer workgroup adds one value per channel and sum to ONE replica -- few replicas for the consumers' sake, enough
// of them that a replica does not collect more than a few dozen same-address atomics
int fold_slots(double rows) {
  static const int forced = getenv("MMVQA_BN_SLOTS") ? atoi(getenv("MMVQA_BN_SLOTS")) : 0;
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
  const double row_tiles = rows / 64.0;
  return row_tiles >= 32.0 ? 4 : (row_tiles >= 8.0 ? 2 : 1);
}

void plan_bn(Arena& a, BNRef& bn, double count) {
  bn.count = count;
  const size_t C = bn.C;
  bn.scale = a.f(C); bn.shift = a.f(C); bn.mean = a.f(C); bn.invstd = a.f(C);
  bn.P = a.f(C); bn.Q = a.f(C); bn.R = a.f(C);
}

}  // namespace

static inline int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

size_t engine_plan(mmvqa_engine* e, int B, int T, int IH, int IW) {
  const mmvqa_model_desc& d = e->d;
  if (B <= 0 || T <= 0 || T > 128 || T > d.max_pos || T <= d.num_vis) {
    mmvqa_set_error(MMVQA_ERR_ARG, "plan: bad B=%d T=%d (max_pos %d, num_vis %d, T<=128)", B, T, d.max_pos, d.num_vis);
    return 0;
  }
  e->B = B; e->T = T; e->IH = IH; e->IW = IW;
  e->pixmask_off.clear(); e->ws_ready = false;
  Arena a;
  const int H = d.hidden;
  const size_t M = (size_t)B * T;
  // ---- backbone
  size_t max_io = 0, max_mid = 0, max_tapM = 0;
  if (d.cnn == 1) {
    // TensorFlow SAME padding: out = ceil(in/s); the odd padding element goes to the end, so the leading pad is
    // total/2 and the trailing one is implied by the bounds checks of the gather
    auto same = [](int in, int k, int s2, int* out) {
      *out = (in + s2 - 1) / s2;
      int total = std::max((*out - 1) * s2 + k - in, 0);
      return total / 2;
    };
    int ph = same(IH, 3, 2, &e->SH), pw = same(IW, 3, 2, &e->SW);
    if (ph != pw) { mmvqa_set_error(MMVQA_ERR_ARG, "plan: image %dx%d needs different SAME pads per axis", IH, IW); return 0; }
    e->stem_conv.pad = ph;
    const size_t M0 = (size_t)B * e->SH * e->SW;
    e->z0 = a.f(M0 * 24);
    e->eff_a0 = a.f(M0 * 24);
    plan_bn(a, e->stem_bn, (double)M0);
    max_io = M0 * 24;
    int h = e->SH, wd = e->SW;
    size_t max_se = 0, max_separt = 0;
    for (auto& blk : e->eff) {
      blk.N = B; blk.H = h; blk.W = wd;
      int p1 = same(h, 3, blk.stride, &blk.OH), p2 = same(wd, 3, blk.stride, &blk.OW);
      if (p1 != p2) { mmvqa_set_error(MMVQA_ERR_ARG, "plan: map %dx%d needs different SAME pads per axis", h, wd); return 0; }
      blk.pad = p1;
      const size_t Min = (size_t)B * h * wd, Mout = (size_t)B * blk.OH * blk.OW;
      if (blk.type == 0) {
        blk.c_a.pad = p1;
        blk.za = a.f(Mout * blk.cout); plan_bn(a, blk.b_a, (double)Mout);
      } else if (blk.type == 1) {
        blk.c_a.pad = p1;
        blk.za = a.f(Mout * blk.mid); plan_bn(a, blk.b_a, (double)Mout);
        blk.zp = a.f(Mout * blk.cout); plan_bn(a, blk.b_p, (double)Mout);
        max_mid = std::max(max_mid, Mout * blk.mid);
      } else {
        blk.za = a.f(Min * blk.mid); plan_bn(a, blk.b_a, (double)Min);
        blk.zdw = a.f(Mout * blk.mid); plan_bn(a, blk.b_dw, (double)Mout);
        blk.zp = a.f(Mout * blk.cout); plan_bn(a, blk.b_p, (double)Mout);
        blk.pool = a.f((size_t)B * blk.mid); blk.gpre = a.f((size_t)B * blk.mid); blk.gate = a.f((size_t)B * blk.mid);
        blk.rpre = a.f((size_t)B * blk.rd); blk.r = a.f((size_t)B * blk.rd);
        max_mid = std::max(max_mid, std::max(Min, Mout) * blk.mid);
        max_se = std::max(max_se, (size_t)B * blk.mid);
        max_separt = std::max(max_separt, k_se_fc_bwd_scratch_floats(B, blk.mid, blk.rd));
        if (!k_se_fc_bwd_ok(blk.rd)) {   // refuse here, not at the first backward
          mmvqa_set_error(MMVQA_ERR_ARG, "plan: squeeze-excite reduce width %d exceeds what se_fc_bwd supports", blk.rd);
          return 0;
        }
      }
      blk.out = a.f(Mout * blk.cout);
      max_io = std::max(max_io, std::max(Min * blk.cin, Mout * blk.cout));
      if (blk.feature >= 0) { e->taps[blk.feature].HW = blk.OH * blk.OW; e->taps[blk.feature].M = (long)Mout; }
      h = blk.OH; wd = blk.OW;
    }
    for (int k = 0; k < 2; ++k) { e->eff_gA[k] = a.f(max_mid); e->eff_gB[k] = a.f(max_mid); }
    for (int i = 0; i < 6; ++i) e->eff_se[i] = a.f(max_se + 64);
    e->eff_separt = a.f(max_separt + 64);
    for (int k = 0; k < 5; ++k) {
      max_tapM = std::max(max_tapM, (size_t)e->taps[k].M);
      e->tapgrad[k] = a.f((size_t)e->taps[k].M * e->taps[k].C);
    }
  } else {
  e->SH = conv_out(IH, 7, 2, 3); e->SW = conv_out(IW, 7, 2, 3);
  e->PH = conv_out(e->SH, 3, 2, 1); e->PW = conv_out(e->SW, 3, 2, 1);
  const int w = d.resnet_width;
  const size_t M0 = (size_t)B * e->SH * e->SW;
  e->z0 = a.f(M0 * w);
  plan_bn(a, e->stem_bn, (double)M0);
  e->p0 = a.f((size_t)B * e->PH * e->PW * w);
  e->pool_idx = a.f(((size_t)B * e->PH * e->PW * w + 3) / 4);
  int h = e->PH, wd = e->PW;
  max_io = M0 * w; max_tapM = M0;
  for (auto& blk : e->blocks) {
    blk.N = B; blk.H = h; blk.W = wd;
    blk.OH = conv_out(h, 3, blk.c2.stride, 1); blk.OW = conv_out(wd, 3, blk.c2.stride, 1);
    const size_t Min = (size_t)B * h * wd, Mout = (size_t)B * blk.OH * blk.OW;
    blk.z1 = a.f(Min * blk.c1.Cout);
    if (blk.c2.stride == 1) {   // per-pixel tap-validity table of the 3x3 weight gradient: one per geometry, in the workspace
      char key[96];
      snprintf(key, sizeof(key), "%d,%d,%d,%d,%d", B, blk.OH, blk.OW, blk.c2.KH, blk.c2.pad);
      if (!e->pixmask_off.count(key))
        e->pixmask_off[key] = mmvqa_engine::PixGeom{B, blk.OH, blk.OW, h, wd, blk.c2.KH, blk.c2.stride, blk.c2.pad, a.f(Mout)};
    }
    blk.z2 = a.f(Mout * blk.c2.Cout);
    blk.z3 = a.f(Mout * blk.c3.Cout);
    blk.out = a.f(Mout * blk.c3.Cout);
    plan_bn(a, blk.b1, (double)Min);
    plan_bn(a, blk.b2, (double)Mout);
    plan_bn(a, blk.b3, (double)Mout);
    if (blk.has_ds) { blk.zd = a.f(Mout * blk.cd.Cout); plan_bn(a, blk.bd, (double)Mout); }
    if (e->bn_fold) {
      blk.b1.slots = fold_slots((double)Min);
      blk.b2.slots = blk.b3.slots = blk.bd.slots = fold_slots((double)Mout);
    }
    max_io = std::max(max_io, std::max(Min * blk.c1.Cin, Mout * blk.c3.Cout));
    max_mid = std::max(max_mid, std::max(Min * blk.c1.Cout, Mout * blk.c2.Cout));
    h = blk.OH; wd = blk.OW;
  }
  // taps: k=0..3 on the outputs of layer4..layer1, k=4 on the stem
  for (int k = 0; k < 4; ++k) {
    const BlockRef& blk = e->blocks[e->layer_end[3 - k]];
    e->taps[k].HW = blk.OH * blk.OW;
    e->taps[k].M = (long)B * blk.OH * blk.OW;
  }
  e->taps[4].HW = e->SH * e->SW;
  e->taps[4].M = (long)M0;
  for (int k = 0; k < 5; ++k) {
    max_tapM = std::max(max_tapM, (size_t)e->taps[k].M);
    e->tapgrad[k] = a.f((size_t)e->taps[k].M * e->taps[k].C);
  }
  }
  e->vis = a.f((size_t)5 * B * H);
  for (int i = 0; i < 3; ++i) e->sk_ws[i] = a.f(SK_WS_FLOATS);   // split-K partial tiles (caller's / side / tap stream)
  for (int i = 0; i < 3; ++i) e->sk_cnt[i] = a.f(SK_CNT_N);      // arrival tickets of ticketed / persistent launches
  e->dvis = a.f((size_t)5 * B * H);
  e->du = a.f(max_tapM * H);
  for (int i = 0; i < 3; ++i) e->gbuf[i] = a.f(max_io);
  e->g1buf = a.f(max_mid);
  e->g2buf = a.f(max_mid);
  e->dstmp = a.f(max_io);
  // ---- embeddings / encoder
  e->emb_out = a.f(M * H); e->emb_xhat = a.f(M * H); e->emb_rstd = a.f(M);
  if (d.encoder == 0) {
    for (auto& L : e->bert) {
      L.xn1 = a.f(M * H); L.mean1 = a.f(M); L.rstd1 = a.f(M);
      L.qkvo = a.f(M * 3 * H); L.probs = a.f((size_t)B * d.heads * T * T); L.ctx = a.f(M * H);
      L.y = a.f(M * H); L.xn2 = a.f(M * H); L.mean2 = a.f(M); L.rstd2 = a.f(M);
      L.pre1 = a.f(M * 4 * H); L.h1 = a.f(M * 4 * H); L.z = a.f(M * H);
    }
  } else {
    for (auto& L : e->rf) {
      L.kqvo = a.f(M * 3 * H); L.probs = a.f((size_t)B * 8 * T * T); L.prev = a.f((size_t)B * T * T * 8);
      L.res = a.f(M * H); L.s1 = a.f(M * H); L.x1 = a.f(M * H); L.mean1 = a.f(M); L.rstd1 = a.f(M);
      L.pre = a.f(M * 4 * H); L.hact = a.f(M * 4 * H); L.s2 = a.f(M * H); L.x2 = a.f(M * H);
      L.mean2 = a.f(M); L.rstd2 = a.f(M);
    }
  }
  // ---- heads
  const size_t HM = d.head_kind == 0 ? M : (size_t)B;
  e->hd_pool = a.f((size_t)B * H);
  e->hd_upre = a.f(HM * H); e->hd_u = a.f(HM * H); e->hd_c0 = a.f(HM * H); e->hd_c1 = a.f(HM * H);
  e->hd_mean = a.f(HM); e->hd_rstd = a.f(HM);
  if (d.supcon) {
    e->sc_pool = a.f((size_t)B * H); e->sc_pre = a.f((size_t)B * H); e->sc_a = a.f((size_t)B * H);
    e->sc_f = a.f((size_t)B * d.feat_dim); e->sc_nrm = a.f(B); e->sc_y = a.f((size_t)B * d.feat_dim);
  }
  // ---- backward scratch
  e->t_a = a.f(M * H); e->t_b = a.f(M * H); e->t_c = a.f(M * H); e->t_d = a.f(M * H);
  e->t_big = a.f(M * 4 * H);
  e->t_dprev[0] = a.f((size_t)B * T * T * 8); e->t_dprev[1] = a.f((size_t)B * T * T * 8);
  // ---- BatchNorm statistic zones (doubles): [fwd of every BN][bwd of every BN]
  e->statzone = a.f(0);
  size_t sd = 0;  // in doubles
  auto take = [&](BNRef& bn, bool bwd) {
    (bwd ? bn.stat_b : bn.stat_f) = sd;
    sd += (size_t)MMVQA_STAT_SLOTS * bn.C * 2;
  };
  for (int pass = 0; pass < 2; ++pass) {
    const bool bw = pass == 1;
    take(e->stem_bn, bw);
    for (auto& blk : e->blocks) { take(blk.b1, bw); take(blk.b2, bw); take(blk.b3, bw); if (blk.has_ds) take(blk.bd, bw); }
    for (auto& blk : e->eff) { take(blk.b_a, bw); if (blk.type == 2) take(blk.b_dw, bw); if (blk.type != 0) take(blk.b_p, bw); }
  }
  const size_t half = sd / 2;
  if (sd != 2 * half) { mmvqa_set_error(MMVQA_ERR_STATE, "plan: stat zone mismatch"); return 0; }
  e->statzone_floats = sd * 2;
  a.f(e->statzone_floats);
  e->ws_floats = (a.cur + 63) & ~(size_t)63;
  e->planned = true;
  e->bound = false;
  return e->ws_floats * sizeof(float);
}

// --------------------------------------------------------------------------- launch helpers
#define WS(off) (e->ws + (off))
#define PRM(off) (e->params + (off))
#define GRD(off) (e->grads + (off))
static inline double* stat_ptr(mmvqa_engine* e, size_t doff) {
  return reinterpret_cast<double*>(e->ws + e->statzone) + doff;
}

static int prof_begin(mmvqa_engine* e, hipStream_t st, int cls, double flops, int tag = HB_NONE, double bytes = 0.0) {
  if (!e->prof_on) return MMVQA_OK;
  mmvqa_engine::ProfRec r;
  HIP_CHECK_RET(hipEventCreate(&r.a));
  HIP_CHECK_RET(hipEventCreate(&r.b));
  r.cls = cls; r.reg = e->prof_reg; r.flops = flops; r.tag = tag; r.bytes = bytes;
  HIP_CHECK_RET(hipEventRecord(r.a, st));
  e->prof.push_back(r);
  return MMVQA_OK;
}
static int prof_end(mmvqa_engine* e, hipStream_t st) {
  if (!e->prof_on) return MMVQA_OK;
  HIP_CHECK_RET(hipEventRecord(e->prof.back().b, st));
  return MMVQA_OK;
}
struct RegScope {   // tags the launches of a scope with a profiler region
  mmvqa_engine* e; int old;
  RegScope(mmvqa_engine* e_, int r) : e(e_), old(e_->prof_reg) { e->prof_reg = r; }
  ~RegScope() { e->prof_reg = old; }
};
#define REG(r) RegScope _reg_scope(e, r)
// an HBM-bound launch: `bytes` = the tensors it has to read and write once (algorithmic bytes, fp32)
#define RUNB(tag, bytes, call)                             \
  do {                                                     \
    TRY(prof_begin(e, st, PROF_OTHER, 0, tag, (double)(bytes))); \
    TRY(call);                                             \
    TRY(prof_end(e, st));                                  \
  } while (0)
#define RUN(cls, flops, call)          \
  do {                                 \
    TRY(prof_begin(e, st, cls, flops)); \
    TRY(call);                         \
    TRY(prof_end(e, st));              \
  } while (0)

static GemmParams gp_linear_geom() {
  GemmParams g;
  memset(&g, 0, sizeof(g));
  g.g_SH = g.g_SW = g.g_OH = g.g_OW = 1;
  g.g_KH = g.g_KW = 1; g.g_stride = 1; g.g_pad = 0;
  return g;
}

// scratch for K split over workgroups (igemm.hip: partial tiles + finishing launch); one per stream
static void set_sk(mmvqa_engine* e, hipStream_t st, GemmParams& g) {
  static const bool off = getenv("MMVQA_NO_SK_WS") != nullptr;   // A/B switch: no K split of forward / data-gradient products
  if (off) return;
  const int which = (e->side && st == e->side) ? 1 : ((e->side2 && st == e->side2) ? 2 : 0);
  g.sk_ws = WS(e->sk_ws[which]);
  g.sk_ws_floats = (long long)SK_WS_FLOATS;
  g.sk_cnt = reinterpret_cast<unsigned int*>(WS(e->sk_cnt[which]));
  g.sk_cnt_n = SK_CNT_N;
}

struct EpiOpt {
  const float* R = nullptr; int r_ld = 0;
  const float* Mk = nullptr; int mk_ld = 0; const float* mk_s = nullptr; const float* mk_b = nullptr; int mk_mode = 0;
  BNRef* st1 = nullptr; const float* Z1 = nullptr;
  BNRef* st2 = nullptr; const float* Z2 = nullptr;
};

static void apply_epi(mmvqa_engine* e, GemmParams& g, const EpiOpt& o) {
  g.R = o.R; g.r_ld = o.r_ld;
  g.Mk = o.Mk; g.mk_ld = o.mk_ld; g.mk_s = o.mk_s; g.mk_b = o.mk_b; g.mk_mode = o.mk_mode;
  if (o.st1) {
    g.stat1 = stat_ptr(e, o.st1->stat_b); g.stat_bwd = 1; g.stat_slots = o.st1->slots;
    g.Z1 = o.Z1; g.z1_ld = o.st1->C; g.mean1 = WS(o.st1->mean); g.invstd1 = WS(o.st1->invstd);
  }
  if (o.st2) {
    g.stat2 = stat_ptr(e, o.st2->stat_b);
    g.Z2 = o.Z2; g.z2_ld = o.st2->C; g.mean2 = WS(o.st2->mean); g.invstd2 = WS(o.st2->invstd);
  }
}

// y[M,out] = act(x[M,in] W^T + b) (+dropout) (+R)
static int lin_fwd(mmvqa_engine* e, hipStream_t st, const float* x, int x_ld, long M, const LinRef& L, float* y,
                   int y_ld, int act, float* pre, float drop_p, uint32_t seed, const float* R, int r_ld) {
  GemmParams g = gp_linear_geom();
  g.M = (int)M; g.N = L.out; g.K = L.in;
  g.A = x; g.a_ld = x_ld; g.g_Cs = L.in;
  g.B = PRM(L.w); g.b_ld = L.in;
  g.C = y; g.c_ld = y_ld; g.Cpre = pre;
  g.bias = L.b >= 0 ? PRM(L.b) : nullptr;
  g.act = act; g.drop_p = drop_p; g.drop_seed = seed; g.R = R; g.r_ld = r_ld;
  set_sk(e, st, g);
  RUN(PROF_IGEMM, 2.0 * M * L.out * L.in, mmvqa_launch_igemm(g, KIND_FWD, 0, 0, st));
  return MMVQA_OK;
}

// dx[M,in] = dy[M,out] W  (optionally * act'(Pre), + R, column sums for the upstream bias)
static int lin_dgrad(mmvqa_engine* e, hipStream_t st, const float* dy, int dy_ld, long M, const LinRef& L, float* dx,
                     int dx_ld, int dact, const float* Pre, int pre_ld, float* colsum, const float* R, int r_ld) {
  GemmParams g = gp_linear_geom();
  g.M = (int)M; g.N = L.in; g.K = L.out;
  g.A = dy; g.a_ld = dy_ld; g.g_Cs = L.out;
  g.B = PRM(L.w); g.b_ld = L.in; g.b_tapstride = 0;
  g.C = dx; g.c_ld = dx_ld;
  g.dact = dact; g.Pre = Pre; g.pre_ld = pre_ld; g.colsum = colsum; g.R = R; g.r_ld = r_ld;
  // Few output tiles and a long contraction (the vocabulary-sized decoder: 96 tiles x 477 K-tiles): split K over
  // workgroups and accumulate with atomics into the zeroed output.  Only for a plain epilogue.
  const long tiles = ((M + 63) / 64) * ((L.in + 63) / 64);
  // (the vocabulary-sized contraction, L.out = 30522, is faster through the scratch + finishing launch that set_sk
  // enables below: heads 0.94 -> 0.90 ms per step; the QKV-sized one, 2304, is faster this way: 0.37 vs 0.40 ms)
  if (!dact && !colsum && !R && tiles < 256 && L.out >= 1536 && L.out <= 8192 && dx_ld == L.in) {
    int sk = (int)((512 + tiles - 1) / tiles);
    const int maxs = (L.out / 64) / 8;
    if (sk > maxs) sk = maxs;
    if (sk > 1) {
      HIP_CHECK_RET(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)M * dx_ld, st));
      g.c_atomic = 1; g.splitk = sk;
    }
  }
  set_sk(e, st, g);
  RUN(PROF_IGEMM, 2.0 * M * L.out * L.in, mmvqa_launch_igemm(g, KIND_DGRAD, 0, 0, st));
  return MMVQA_OK;
}

// dW[out,in] += dy^T x ; db[out] += colsum(dy)
static int lin_wgrad(mmvqa_engine* e, hipStream_t st, const float* dy, int dy_ld, const float* x, int x_ld, long M,
                     const LinRef& L, bool bias_from_colsum) {
  GemmParams g = gp_linear_geom();
  g.M = L.out; g.N = L.in; g.K = (int)M;
  g.A = dy; g.a_ld = dy_ld;
  g.B = x; g.b_ld = x_ld; g.g_Cs = L.in;
  g.C = GRD(L.w); g.c_ld = L.in; g.c_atomic = 1;
  RUN(PROF_IGEMM, 2.0 * M * L.out * L.in, mmvqa_launch_igemm(g, KIND_WGRAD, 0, 0, st));
  if (bias_from_colsum && L.b >= 0) RUN(PROF_OTHER, 0, k_colsum(st, dy, dy_ld, (int)M, L.out, GRD(L.b)));
  return MMVQA_OK;
}

static int ln_fwd(mmvqa_engine* e, hipStream_t st, const float* x, const LNRef& ln, float* y, float* mean,
                  float* rstd, long rows, float eps) {
  RUNB(HB_LAYERNORM_FWD, 8.0 * rows * e->d.hidden, k_layernorm_fwd(st, x, nullptr, PRM(ln.g), PRM(ln.b), y, nullptr, mean, rstd, (int)rows,
                                     e->d.hidden, eps));
  return MMVQA_OK;
}
static int ln_bwd(mmvqa_engine* e, hipStream_t st, const float* dy, const float* x, const LNRef& ln,
                  const float* mean, const float* rstd, const float* dres, float* dx, long rows) {
  RUNB(HB_LAYERNORM_BWD, (dres ? 16.0 : 12.0) * rows * e->d.hidden, k_layernorm_bwd(st, dy, x, PRM(ln.g), mean, rstd, dres, dx, GRD(ln.g), GRD(ln.b), (int)rows,
                                     e->d.hidden));
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- convolution helpers
static void conv_geom(GemmParams& g, const ConvRef& c, int H, int W, int OH, int OW) {
  g.g_KH = g.g_KW = c.KH; g.g_stride = c.stride; g.g_pad = c.pad;
  (void)H; (void)W; (void)OH; (void)OW;
}

static int bn_coef_fwd(mmvqa_engine* e, hipStream_t st, BNRef& bn) {
  REG(REG_BNCOEF);
  RUN(PROF_OTHER, 0,
      k_bn_coef_fwd(st, stat_ptr(e, bn.stat_f), bn.C, bn.count, bn.eps, PRM(bn.gamma), PRM(bn.beta),
                    e->bufs + bn.rmean, e->bufs + bn.rvar, e->nbt + bn.nbt, 0.1f, bn.reps, e->training,
                    WS(bn.scale), WS(bn.shift), WS(bn.mean), WS(bn.invstd)));
  return MMVQA_OK;
}
static int bn_coef_bwd(mmvqa_engine* e, hipStream_t st, BNRef& bn) {
  REG(REG_BNCOEF);
  RUN(PROF_OTHER, 0,
      k_bn_coef_bwd(st, stat_ptr(e, bn.stat_b), bn.C, bn.count, PRM(bn.gamma), WS(bn.mean), WS(bn.invstd),
                    e->training, WS(bn.P), WS(bn.Q), WS(bn.R), GRD(bn.gamma), GRD(bn.beta)));
  return MMVQA_OK;
}

// The ResNet path in training mode folds the BatchNorm coefficients inside the launches that consume them
// (mmvqa_bn_fold: no coefficient launch between a convolution and its consumer on the dependency chain).
static inline bool folding(const mmvqa_engine* e) { return e->bn_fold && e->training && e->d.cnn == 0; }
// The EfficientNet path folds where the FIRST consumer of a BatchNorm is an elementwise kernel (block end, depthwise
// convolution, squeeze-excite pooling: forward) -- later consumers read what that launch published.
static inline bool eff_folding(const mmvqa_engine* e) { return e->bn_fold && e->training && e->d.cnn != 0; }

static void set_fold_fwd(mmvqa_engine* e, mmvqa_bn_fold& f, BNRef& bn) {
  memset(&f, 0, sizeof(f));
  f.stat = stat_ptr(e, bn.stat_f); f.slots = bn.slots; f.bwd = 0; f.publish = 1; f.reps = bn.reps;
  f.count = bn.count; f.keep = pow(1.0 - (double)0.1f, (double)bn.reps); f.eps = bn.eps;
  f.gamma = PRM(bn.gamma); f.beta = PRM(bn.beta);
  f.out0 = WS(bn.scale); f.out1 = WS(bn.shift); f.out2 = WS(bn.mean); f.out3 = WS(bn.invstd);
  f.run_mean = e->bufs + bn.rmean; f.run_var = e->bufs + bn.rvar; f.nbt = e->nbt + bn.nbt;
}
static void set_fold_bwd(mmvqa_engine* e, mmvqa_bn_fold& f, const BNRef& bn, bool publish) {
  memset(&f, 0, sizeof(f));
  f.stat = stat_ptr(e, bn.stat_b); f.slots = bn.slots; f.bwd = 1; f.publish = publish ? 1 : 0; f.reps = bn.reps;
  f.count = bn.count; f.keep = 1.0; f.eps = bn.eps;
  f.gamma = PRM(bn.gamma); f.mean = WS(bn.mean); f.invstd = WS(bn.invstd);
  f.out0 = WS(bn.P); f.out1 = WS(bn.Q); f.out2 = WS(bn.R);
  f.dgamma = GRD(bn.gamma); f.dbeta = GRD(bn.beta);
}

// z = conv(x) with x optionally = relu(bn_in(x_raw)); accumulates the batch statistics of z
static int conv_fwd(mmvqa_engine* e, hipStream_t st, const ConvRef& c, const float* x, const BNRef* bn_in, int N,
                    int H, int W, int OH, int OW, float* z, BNRef& bn_out) {
  GemmParams g;
  memset(&g, 0, sizeof(g));
  g.M = N * OH * OW; g.N = c.Cout; g.K = c.KH * c.KH * c.Cin;
  g.A = x; g.a_ld = c.Cin;
  if (bn_in) {
    g.a_pro = PRO_AFFINE_RELU; g.a_c0 = WS(bn_in->scale); g.a_c1 = WS(bn_in->shift);
    if (folding(e)) set_fold_fwd(e, g.a_fold, *const_cast<BNRef*>(bn_in));   // this launch is bn_in's first consumer
  }
  g.g_SH = H; g.g_SW = W; g.g_Cs = c.Cin; g.g_OH = OH; g.g_OW = OW;
  conv_geom(g, c, H, W, OH, OW);
  g.B = PRM(c.w); g.b_ld = g.K;
  g.C = z; g.c_ld = c.Cout;
  if (e->training) { g.stat1 = stat_ptr(e, bn_out.stat_f); g.stat_bwd = 0; g.stat_slots = bn_out.slots; }
  set_sk(e, st, g);
  RUN(PROF_IGEMM, 2.0 * g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_FWD, 0, 0, st));
  if (folding(e)) return MMVQA_OK;   // bn_out's coefficients: folded by its consumer (next convolution / block end)
  return bn_coef_fwd(e, st, bn_out);
}

// dW += dz^T * x_gathered with dz = P*G + Q*z + R (BatchNorm backward of bn_out), x optionally relu(bn_in(.))
static int conv_wgrad(mmvqa_engine* e, hipStream_t st, const ConvRef& c, const float* G, const float* z,
                      const BNRef& bn_out, const float* x, const BNRef* bn_in, int N, int H, int W, int OH, int OW) {
  GemmParams g;
  memset(&g, 0, sizeof(g));
  g.M = c.Cout; g.N = c.KH * c.KH * c.Cin; g.K = N * OH * OW;
  g.A = G; g.A2 = z; g.a_ld = c.Cout; g.a_pro = PRO_DZ;
  g.a_c0 = WS(bn_out.P); g.a_c1 = WS(bn_out.Q); g.a_c2 = WS(bn_out.R);
  if (folding(e)) set_fold_bwd(e, g.a_fold, bn_out, false);   // (the data gradient of the same BatchNorm publishes)
  g.B = x; g.b_ld = c.Cin;
  if (bn_in) { g.b_pro = PRO_AFFINE_RELU; g.b_c0 = WS(bn_in->scale); g.b_c1 = WS(bn_in->shift); }
  g.g_SH = H; g.g_SW = W; g.g_Cs = c.Cin; g.g_OH = OH; g.g_OW = OW;
  conv_geom(g, c, H, W, OH, OW);
  g.C = GRD(c.w); g.c_ld = g.N; g.c_atomic = 1;
  if (c.KH > 1 && c.stride == 1 && OH == H && OW == W) {
    // stride-1 "same" convolution: hand the kernel the per-pixel tap-validity table (one per geometry)
    char key[96];
    snprintf(key, sizeof(key), "%d,%d,%d,%d,%d", N, OH, OW, c.KH, c.pad);
    auto it = e->pixmask_off.find(key);
    if (it != e->pixmask_off.end()) g.pixmask = reinterpret_cast<int*>(WS(it->second.off));   // built by prepare_workspace()
  }
  RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_WGRAD, 0, 0, st));
  return MMVQA_OK;
}

// dx[N,H,W,Cin] = conv_transpose(dz, W) with dz as above; epilogue options in `o`
static int conv_dgrad(mmvqa_engine* e, hipStream_t st, const ConvRef& c, const float* G, const float* z,
                      const BNRef& bn_out, int N, int H, int W, int OH, int OW, float* dx, const EpiOpt& o) {
  GemmParams g;
  memset(&g, 0, sizeof(g));
  g.M = N * H * W; g.N = c.Cin; g.K = c.KH * c.KH * c.Cout;
  g.A = G; g.A2 = z; g.a_ld = c.Cout; g.a_pro = PRO_DZ;
  g.a_c0 = WS(bn_out.P); g.a_c1 = WS(bn_out.Q); g.a_c2 = WS(bn_out.R);
  if (folding(e)) set_fold_bwd(e, g.a_fold, bn_out, true);
  g.g_SH = OH; g.g_SW = OW; g.g_Cs = c.Cout; g.g_OH = H; g.g_OW = W;
  conv_geom(g, c, H, W, OH, OW);
  g.B = PRM(c.w); g.b_ld = c.KH * c.KH * c.Cin; g.b_tapstride = c.Cin;
  g.C = dx; g.c_ld = c.Cin;
  apply_epi(e, g, o);
  set_sk(e, st, g);
  RUN(PROF_IGEMM, 2.0 * (double)N * OH * OW * c.Cout * c.KH * c.KH * c.Cin,
      mmvqa_launch_igemm(g, KIND_DGRAD, 0, 0, st));
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- taps (models/image_encoding.py:53-62,72-86)
static int tap_act(const mmvqa_engine* e) { return e->d.use_relu ? ACT_RELU : ACT_SERF; }

static int tap_fwd(mmvqa_engine* e, hipStream_t st, int k, const float* fmap, const BNRef* bn_in) {
  REG(REG_TAP);
  const TapRef& t = e->taps[k];
  static const bool thin_off = getenv("MMVQA_NO_TAP_THIN") != nullptr;   // A/B switch
  // (C = 64 with BatchNorm+ReLU on load needs 256 VGPRs there and only ties the GEMM kernel: measured, round 2)
  if (!thin_off && t.C <= 32 && k_tap_thin_ok(t.M, e->d.hidden, t.C, t.HW)) {
    // few channels, huge map (EfficientNet's stem tap): weights stay in registers, no tile machinery (tapthin.hip)
    TRY(prof_begin(e, st, PROF_MATRIX, 2.0 * (double)t.M * e->d.hidden * t.C, HB_TAP_THIN_FWD, 4.0 * (double)t.M * t.C));
    TRY(k_tap_thin_fwd(st, fmap, bn_in ? WS(bn_in->scale) : nullptr, bn_in ? WS(bn_in->shift) : nullptr,
                                      PRM(t.w), WS(e->vis) + (size_t)k * e->B * e->d.hidden, t.M, e->d.hidden, t.C, t.HW,
                                      tap_act(e)));
    TRY(prof_end(e, st));
    return MMVQA_OK;
  }
  GemmParams g = gp_linear_geom();
  g.M = (int)t.M; g.N = e->d.hidden; g.K = t.C;
  g.A = fmap; g.a_ld = t.C; g.g_Cs = t.C;
  if (bn_in) { g.a_pro = PRO_AFFINE_RELU; g.a_c0 = WS(bn_in->scale); g.a_c1 = WS(bn_in->shift); }
  g.B = PRM(t.w); g.b_ld = t.C;
  g.epi_mode = EPI_TAP_FWD; g.act = tap_act(e); g.tap_HW = t.HW;
  g.tap_out = WS(e->vis) + (size_t)k * e->B * e->d.hidden;
  g.C = g.tap_out; g.c_ld = g.N;  // unused by this epilogue
  set_sk(e, st, g);
  RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_FWD, 0, 0, st));
  return MMVQA_OK;
}

// backward of one tap: du (recompute), dW_tap, and the gradient wrt the feature map (T_k or masked G)
static int tap_bwd(mmvqa_engine* e, hipStream_t st, int k, const float* fmap, const BNRef* bn_in, float* dfmap,
                   const EpiOpt& o) {
  REG(REG_TAP);
  const TapRef& t = e->taps[k];
  const int Hd = e->d.hidden;
  GemmParams g = gp_linear_geom();
  g.M = (int)t.M; g.N = Hd; g.K = t.C;
  g.A = fmap; g.a_ld = t.C; g.g_Cs = t.C;
  if (bn_in) { g.a_pro = PRO_AFFINE_RELU; g.a_c0 = WS(bn_in->scale); g.a_c1 = WS(bn_in->shift); }
  g.B = PRM(t.w); g.b_ld = t.C;
  g.epi_mode = EPI_TAP_BWD; g.act = tap_act(e); g.tap_HW = t.HW;
  g.tap_dv = WS(e->dvis) + (size_t)k * e->B * Hd;
  g.C = WS(e->du); g.c_ld = Hd;
  set_sk(e, st, g);
  static const bool thin_off = getenv("MMVQA_NO_TAP_THIN") != nullptr;
  if (!thin_off && t.C <= 32 && k_tap_thin_ok(t.M, Hd, t.C, t.HW))   // EfficientNet stem tap: recompute in registers (tapthin.hip)
  {
    TRY(prof_begin(e, st, PROF_MATRIX, 2.0 * (double)t.M * Hd * t.C, HB_TAP_THIN_BWD, 4.0 * (double)t.M * (t.C + Hd)));
    TRY(k_tap_thin_bwd(st, fmap, bn_in ? WS(bn_in->scale) : nullptr, bn_in ? WS(bn_in->shift) : nullptr,
                       PRM(t.w), g.tap_dv, WS(e->du), t.M, Hd, t.C, t.HW, tap_act(e)));
    TRY(prof_end(e, st));
  } else
    RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_FWD, 0, 0, st));
  // dW_tap[Hd][C] += du^T fmap
  GemmParams w = gp_linear_geom();
  w.M = Hd; w.N = t.C; w.K = (int)t.M;
  w.A = WS(e->du); w.a_ld = Hd;
  w.B = fmap; w.b_ld = t.C; w.g_Cs = t.C;
  if (bn_in) { w.b_pro = PRO_AFFINE_RELU; w.b_c0 = WS(bn_in->scale); w.b_c1 = WS(bn_in->shift); }
  w.C = GRD(t.w); w.c_ld = t.C; w.c_atomic = 1;
  RUN(PROF_IGEMM, 2.0 * (double)w.M * w.N * w.K, mmvqa_launch_igemm(w, KIND_WGRAD, 0, 0, st));
  // dfmap[M][C] = du W_tap
  GemmParams dg = gp_linear_geom();
  dg.M = (int)t.M; dg.N = t.C; dg.K = Hd;
  dg.A = WS(e->du); dg.a_ld = Hd; dg.g_Cs = Hd;
  dg.B = PRM(t.w); dg.b_ld = t.C;
  dg.C = dfmap; dg.c_ld = t.C;
  apply_epi(e, dg, o);
  set_sk(e, st, dg);
  RUN(PROF_IGEMM, 2.0 * (double)dg.M * dg.N * dg.K, mmvqa_launch_igemm(dg, KIND_DGRAD, 0, 0, st));
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- side stream (backward overlap)
// The data-gradient chain (dgrad -> BN coefficients -> dgrad ...) is the critical path of the backward pass
// and, on the 14x14 / 7x7 layers, fills well under one workgroup per CU; the weight-gradient GEMMs and
// the tap backward only feed the optimizer, so they run on a second HIP stream and fill the idle CUs.
// Ordering is by events: fork() makes the side stream wait for everything enqueued so far on the main
// stream, mark() returns an event for the side work enqueued so far, need() makes the main stream wait
// for such an event before it overwrites a buffer the side work reads.
struct SideCtx {
  mmvqa_engine* e;
  hipStream_t st, sd;
  bool on;
  SideCtx(mmvqa_engine* e_, hipStream_t st_) : e(e_), st(st_) {
    on = e->use_side && !e->tuner.tuning;   // (the per-launch profiler records its events on the launching stream)
    if (on && !e->side) {
      // the side stream carries work with slack (weight gradients, taps, downsample branches): lowest priority, so
      // that the dependency chain on the caller's stream is dispatched first whenever both have kernels ready
      int least = 0, greatest = 0;
      static const bool prio_off = getenv("MMVQA_SIDE_PRIO_OFF") != nullptr;   // A/B switch
      if (!prio_off && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        on = hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, least) == hipSuccess;
      else
        on = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess;
    }
    sd = on ? e->side : st;
  }
  // third stream for the tap backward: four long launch groups (2.4 ms per config-2 step) that are needed only when the
  // chain reaches their layer; queued on `sd` they sat in front of the first weight gradients, whose completion the chain
  // waits for before it reuses a gradient buffer.  MMVQA_TAP_STREAM_OFF=1 puts them back on `sd` (A/B switch): 24.46 ms
  // against 24.15 ms per config-2 step.  (More launches share the chip this way and each launch's own duration stretches:
  // the bench line's per-launch figure `roofline.frac` reads lower on the faster step; `roofline.step_level` and
  // `roofline.single_stream` do not have that blind spot.  DESIGN 7.5.)
  hipStream_t tap_stream() {
    static const bool want = getenv("MMVQA_TAP_STREAM_OFF") == nullptr;
    if (!on || !want) return sd;
    if (!e->side2) {
      int least = 0, greatest = 0;
      bool ok;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        ok = hipStreamCreateWithPriority(&e->side2, hipStreamNonBlocking, least) == hipSuccess;
      else
        ok = hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking) == hipSuccess;
      if (!ok) { e->side2 = nullptr; return sd; }
    }
    return e->side2;
  }
  void fork_to(hipStream_t s) {   // s starts behind everything queued on the caller's stream so far
    if (!on || s == st) return;
    hipEvent_t ev = next_event();
    if (!ev) return;
    (void)hipEventRecord(ev, st);
    (void)hipStreamWaitEvent(s, ev, 0);
  }
  hipEvent_t mark_on(hipStream_t s) {
    if (!on) return nullptr;
    hipEvent_t ev = next_event();
    if (ev) (void)hipEventRecord(ev, s);
    return ev;
  }
  hipEvent_t next_event() {
    if (e->ev_next == e->ev_pool.size()) {
      hipEvent_t ev;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
      e->ev_pool.push_back(ev);
    }
    return e->ev_pool[e->ev_next++];
  }
  void fork() {
    if (!on) return;
    hipEvent_t ev = next_event();
    if (!ev) return;
    (void)hipEventRecord(ev, st);
    (void)hipStreamWaitEvent(sd, ev, 0);
  }
  hipEvent_t mark() {
    if (!on) return nullptr;
    hipEvent_t ev = next_event();
    if (ev) (void)hipEventRecord(ev, sd);
    return ev;
  }
  void need(hipEvent_t ev) {
    if (on && ev) (void)hipStreamWaitEvent(st, ev, 0);
  }
};

// Weight gradients of the linear layers (heads, encoder) on the side stream: the 512-row products of the encoder fill
// 96-384 of 256 CUs x 2 slots, so a weight gradient beside the data gradient it shares its input with is nearly free.
// SideReads remembers, per scratch buffer, the event of the last side launch that READS it; the chain calls write(ptr)
// before it overwrites that buffer.  MMVQA_ENC_SIDE_OFF=1: everything on the caller's stream (A/B switch).
struct SideReads {
  SideCtx& sc;
  bool on;
  std::vector<std::pair<const void*, hipEvent_t>> rd;
  explicit SideReads(SideCtx& s) : sc(s) {
    static const bool off = getenv("MMVQA_ENC_SIDE_OFF") != nullptr;
    on = sc.on && !off;
  }
  void read(const void* p) {
    hipEvent_t ev = sc.mark();
    for (auto& kv : rd) if (kv.first == p) { kv.second = ev; return; }
    rd.emplace_back(p, ev);
  }
  void write(const void* p) {
    for (auto& kv : rd) if (kv.first == p && kv.second) { sc.need(kv.second); kv.second = nullptr; }
  }
  void join() { if (on) sc.need(sc.mark()); }
};
// dW += dy^T x (+ bias gradient) beside the chain: starts behind everything queued on the caller's stream so far
static int lin_wgrad_side(mmvqa_engine* e, SideReads& sr, hipStream_t st, const float* dy, int dy_ld, const float* x, int x_ld, long M,
                          const LinRef& L, bool bias_from_colsum) {
  if (!sr.on) return lin_wgrad(e, st, dy, dy_ld, x, x_ld, M, L, bias_from_colsum);
  sr.sc.fork();
  TRY(lin_wgrad(e, sr.sc.sd, dy, dy_ld, x, x_ld, M, L, bias_from_colsum));
  sr.read(dy);
  return MMVQA_OK;
}

static void notify(mmvqa_engine* e, SideCtx& sc, long long lo, long long hi) {
  if (!e->grad_cb || hi <= lo) return;
  sc.need(sc.mark());   // weight gradients of the range may still be queued on the side stream
  e->grad_cb(e->grad_cb_user, lo, hi);
}

// --------------------------------------------------------------------------- ResNet forward / backward
static int resnet_forward(mmvqa_engine* e, hipStream_t st) {
  const mmvqa_model_desc& d = e->d;
  const int B = e->B, w = d.resnet_width;
  if (e->training)
    HIP_CHECK_RET(hipMemsetAsync(stat_ptr(e, 0), 0, e->statzone_floats / 2 * sizeof(float), st));
  HIP_CHECK_RET(hipMemsetAsync(WS(e->vis), 0, (size_t)5 * B * d.hidden * sizeof(float), st));
  {  // stem 7x7/2 on the NCHW image
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = B * e->SH * e->SW; g.N = w; g.K = 147;
    g.A = e->img; g.g_nchw = 1;
    g.g_SH = e->IH; g.g_SW = e->IW; g.g_Cs = 3; g.g_OH = e->SH; g.g_OW = e->SW;
    g.g_KH = g.g_KW = 7; g.g_stride = 2; g.g_pad = 3;
    g.B = PRM(e->stem_conv.w); g.b_ld = 147;
    g.C = WS(e->z0); g.c_ld = w;
    if (e->training) g.stat1 = stat_ptr(e, e->stem_bn.stat_f);
    RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_FWD, 1, 0, st));
    TRY(bn_coef_fwd(e, st, e->stem_bn));
  }
  // The five taps (1x1 conv -> act -> global average pool) and the downsample convolutions do not feed the
  // bottleneck chain: they run on the side stream beside it (most of the chain's launches fill 196 of 256 CUs).
  SideCtx sc(e, st);
  // A tap is queued on the side stream BEHIND the next block's downsample convolution (MMVQA_TAP_FIRST=1: in front of it, as
  // before round 3's last change): the block end of that block waits for the downsample branch, and a 0.1-0.6 ms tap in
  // front of it stalled the chain at every layer start (the stem tap alone is 634 us against 220 us of layer1.0's
  // three convolutions).
  static const bool tap_first = getenv("MMVQA_TAP_FIRST") != nullptr;
  int pend_tap = -1;
  const float* pend_x = nullptr;
  const BNRef* pend_bn = nullptr;
  if (tap_first) {
    sc.fork();
    TRY(tap_fwd(e, sc.sd, 4, WS(e->z0), &e->stem_bn));
  } else {
    pend_tap = 4; pend_x = WS(e->z0); pend_bn = &e->stem_bn;
  }
  RUNB(HB_MAXPOOL_FWD, 4.0 * B * e->SH * e->SW * w + 5.0 * B * e->PH * e->PW * w, k_maxpool_fwd(st, WS(e->z0), WS(e->stem_bn.scale), WS(e->stem_bn.shift), WS(e->p0),
                                   reinterpret_cast<unsigned char*>(WS(e->pool_idx)), B, e->SH, e->SW, w, e->PH,
                                   e->PW));
  const float* x = WS(e->p0);
  int layer = 0;
  for (size_t i = 0; i < e->blocks.size(); ++i) {
    BlockRef& b = e->blocks[i];
    hipEvent_t ev_ds = nullptr;
    if (b.has_ds) {
      sc.fork();
      TRY(conv_fwd(e, sc.sd, b.cd, x, nullptr, B, b.H, b.W, b.OH, b.OW, WS(b.zd), b.bd));
      ev_ds = sc.mark();
    }
    if (pend_tap >= 0) {   // the tap on the previous layer's output (or the stem's), behind this block's downsample branch
      if (!b.has_ds) sc.fork();
      TRY(tap_fwd(e, sc.sd, pend_tap, pend_x, pend_bn));
      pend_tap = -1;
    }
    TRY(conv_fwd(e, st, b.c1, x, nullptr, B, b.H, b.W, b.H, b.W, WS(b.z1), b.b1));
    TRY(conv_fwd(e, st, b.c2, WS(b.z1), &b.b1, B, b.H, b.W, b.OH, b.OW, WS(b.z2), b.b2));
    TRY(conv_fwd(e, st, b.c3, WS(b.z2), &b.b2, B, b.OH, b.OW, b.OH, b.OW, WS(b.z3), b.b3));
    const long rows = (long)B * b.OH * b.OW;
    if (folding(e)) {
      mmvqa_bn_fold f3, fd;
      set_fold_fwd(e, f3, b.b3);
      if (b.has_ds) { sc.need(ev_ds); set_fold_fwd(e, fd, b.bd); }
      RUNB(HB_BN_ADD_RELU, 12.0 * rows * b.c3.Cout, k_bn_add_relu_fold(st, WS(b.z3), &f3, b.has_ds ? WS(b.zd) : x, b.has_ds ? &fd : nullptr,
                                            WS(b.out), rows, b.c3.Cout));
    } else if (b.has_ds) {
      sc.need(ev_ds);
      RUNB(HB_BN_ADD_RELU, 12.0 * rows * b.c3.Cout, k_bn_add_relu(st, WS(b.z3), WS(b.b3.scale), WS(b.b3.shift), WS(b.zd), WS(b.bd.scale),
                                       WS(b.bd.shift), WS(b.out), rows, b.c3.Cout));
    } else {
      RUNB(HB_BN_ADD_RELU, 12.0 * rows * b.c3.Cout, k_bn_add_relu(st, WS(b.z3), WS(b.b3.scale), WS(b.b3.shift), x, nullptr, nullptr,
                                       WS(b.out), rows, b.c3.Cout));
    }
    x = WS(b.out);
    if ((int)i == e->layer_end[layer]) {
      if (tap_first) {
        sc.fork();
        TRY(tap_fwd(e, sc.sd, 3 - layer, x, nullptr));
      } else {
        pend_tap = 3 - layer; pend_x = x; pend_bn = nullptr;
      }
      ++layer;
    }
  }
  if (pend_tap >= 0) {   // the tap on the last layer's output
    sc.fork();
    TRY(tap_fwd(e, sc.sd, pend_tap, pend_x, pend_bn));
  }
  sc.need(sc.mark());   // join: the embedding reads the visual tokens
  return MMVQA_OK;
}

static int resnet_backward(mmvqa_engine* e, hipStream_t st) {
  const int B = e->B, w = e->d.resnet_width;
  HIP_CHECK_RET(hipMemsetAsync(stat_ptr(e, 0) + e->statzone_floats / 4, 0,
                               e->statzone_floats / 2 * sizeof(float), st));
  SideCtx sc(e, st);
  hipStream_t sd = sc.sd;
  const int nb = (int)e->blocks.size();
  int cur = 0;
  {  // the layer4 tap starts the main chain
    BlockRef& last = e->blocks[nb - 1];
    EpiOpt o;
    o.Mk = WS(last.out); o.mk_ld = last.c3.Cout;
    o.st1 = &last.b3; o.Z1 = WS(last.z3);
    if (last.has_ds) { o.st2 = &last.bd; o.Z2 = WS(last.zd); }
    TRY(tap_bwd(e, st, 0, WS(last.out), nullptr, WS(e->gbuf[cur]), o));
  }
  // taps on layer1..3 and the stem produce side gradients T_k that are only needed when the chain reaches
  // their layer: side stream (they share the `du` scratch with the tap above, hence the fork after it)
  hipEvent_t ev_tap[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t ts = sc.tap_stream();
  sc.fork_to(ts);
  for (int k = 1; k <= 3; ++k) {
    const BlockRef& blk = e->blocks[e->layer_end[3 - k]];
    TRY(tap_bwd(e, ts, k, WS(blk.out), nullptr, WS(e->tapgrad[k]), EpiOpt()));
    ev_tap[k] = sc.mark_on(ts);
  }
  TRY(tap_bwd(e, ts, 4, WS(e->z0), &e->stem_bn, WS(e->tapgrad[4]), EpiOpt()));
  ev_tap[4] = sc.mark_on(ts);
  hipEvent_t ev_g1 = nullptr, ev_g2 = nullptr, ev_prevG = nullptr;   // side readers of g1buf / g2buf / previous G
  long long hi_mark = e->enc_lo;    // gradients in [hi_mark, end) were announced by the caller; fc + taps sit below
  for (int i = nb - 1; i >= 0; --i) {
    BlockRef& b = e->blocks[i];
    const float* G = WS(e->gbuf[cur]);
    const float* x = i == 0 ? WS(e->p0) : WS(e->blocks[i - 1].out);
    float* Gprev = WS(e->gbuf[cur ^ 1]);
    // side gradient arriving at this block's input from a tap (only where the previous block ends a layer)
    const float* extra = nullptr;
    hipEvent_t ev_extra = nullptr;
    for (int l = 0; l < 3; ++l)
      if (i - 1 == e->layer_end[l]) { extra = WS(e->tapgrad[3 - l]); ev_extra = ev_tap[3 - l]; }
    hipEvent_t evG = nullptr;   // side readers of this block's G
    // conv3 / bn3
    if (!folding(e)) TRY(bn_coef_bwd(e, st, b.b3));
    sc.fork();
    TRY(conv_wgrad(e, sd, b.c3, G, WS(b.z3), b.b3, WS(b.z2), &b.b2, B, b.OH, b.OW, b.OH, b.OW));
    evG = sc.mark();
    {
      EpiOpt o;
      o.Mk = WS(b.z2); o.mk_ld = b.c2.Cout; o.mk_s = WS(b.b2.scale); o.mk_b = WS(b.b2.shift);
      o.st1 = &b.b2; o.Z1 = WS(b.z2);
      sc.need(ev_g2);
      TRY(conv_dgrad(e, st, b.c3, G, WS(b.z3), b.b3, B, b.OH, b.OW, b.OH, b.OW, WS(e->g2buf), o));
    }
    // conv2 / bn2
    if (!folding(e)) TRY(bn_coef_bwd(e, st, b.b2));
    sc.fork();
    TRY(conv_wgrad(e, sd, b.c2, WS(e->g2buf), WS(b.z2), b.b2, WS(b.z1), &b.b1, B, b.H, b.W, b.OH, b.OW));
    ev_g2 = sc.mark();
    {
      EpiOpt o;
      o.Mk = WS(b.z1); o.mk_ld = b.c1.Cout; o.mk_s = WS(b.b1.scale); o.mk_b = WS(b.b1.shift);
      o.st1 = &b.b1; o.Z1 = WS(b.z1);
      sc.need(ev_g1);
      TRY(conv_dgrad(e, st, b.c2, WS(e->g2buf), WS(b.z2), b.b2, B, b.H, b.W, b.OH, b.OW, WS(e->g1buf), o));
    }
    // conv1 / bn1
    if (!folding(e)) TRY(bn_coef_bwd(e, st, b.b1));
    sc.fork();
    TRY(conv_wgrad(e, sd, b.c1, WS(e->g1buf), WS(b.z1), b.b1, x, nullptr, B, b.H, b.W, b.H, b.W));
    ev_g1 = sc.mark();
    // gradient wrt the block input: conv1 path + identity/downsample path (+ tap side gradient)
    EpiOpt o;
    if (i > 0) {
      BlockRef& pb = e->blocks[i - 1];
      o.Mk = x; o.mk_ld = b.c1.Cin;
      o.st1 = &pb.b3; o.Z1 = WS(pb.z3);
      if (pb.has_ds) { o.st2 = &pb.bd; o.Z2 = WS(pb.zd); }
    }
    if (b.has_ds) {
      if (!folding(e)) TRY(bn_coef_bwd(e, st, b.bd));
      sc.fork();
      TRY(conv_wgrad(e, sd, b.cd, G, WS(b.zd), b.bd, x, nullptr, B, b.H, b.W, b.OH, b.OW));
      evG = sc.mark();
      EpiOpt od;
      od.R = extra; od.r_ld = b.cd.Cin;
      sc.need(ev_extra);
      TRY(conv_dgrad(e, st, b.cd, G, WS(b.zd), b.bd, B, b.H, b.W, b.OH, b.OW, WS(e->dstmp), od));
      o.R = WS(e->dstmp); o.r_ld = b.c1.Cin;
    } else {
      if (extra) return mmvqa_set_error(MMVQA_ERR_STATE, "resnet_backward: tap gradient without downsample");
      o.R = G; o.r_ld = b.c1.Cin;
    }
    sc.need(ev_prevG);   // the buffer written next was the G of the block before: its side readers must be done
    TRY(conv_dgrad(e, st, b.c1, WS(e->g1buf), WS(b.z1), b.b1, B, b.H, b.W, b.H, b.W, Gprev, o));
    ev_prevG = evG;
    cur ^= 1;
    // announce finished gradient ranges every ~12 blocks and at layer starts (large, few all-reduces)
    bool layer_start = false;
    for (int l = 0; l < 3; ++l) layer_start |= (i - 1 == e->layer_end[l]);
    if (i > 0 && (layer_start || (nb - i) % 12 == 0)) { notify(e, sc, b.c1.w, hi_mark); hi_mark = b.c1.w; }
  }
  // stem: max-pool backward + stem-tap gradient + ReLU mask + BN statistics, then the 7x7 weight gradient
  float* g0 = WS(e->gbuf[cur ^ 1]);
  sc.need(ev_prevG);
  sc.need(ev_tap[4]);
  RUNB(HB_MAXPOOL_BWD, 12.0 * B * e->SH * e->SW * w + 5.0 * B * e->PH * e->PW * w,
      k_maxpool_bwd(st, WS(e->gbuf[cur]), reinterpret_cast<unsigned char*>(WS(e->pool_idx)), WS(e->tapgrad[4]),
                    WS(e->z0), WS(e->stem_bn.scale), WS(e->stem_bn.shift), WS(e->stem_bn.mean),
                    WS(e->stem_bn.invstd), g0, stat_ptr(e, e->stem_bn.stat_b), B, e->SH, e->SW, w, e->PH, e->PW));
  TRY(bn_coef_bwd(e, st, e->stem_bn));
  {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = w; g.N = 147; g.K = B * e->SH * e->SW;
    g.A = g0; g.A2 = WS(e->z0); g.a_ld = w; g.a_pro = PRO_DZ;
    g.a_c0 = WS(e->stem_bn.P); g.a_c1 = WS(e->stem_bn.Q); g.a_c2 = WS(e->stem_bn.R);
    g.B = e->img; g.g_nchw = 1;
    g.g_SH = e->IH; g.g_SW = e->IW; g.g_Cs = 3; g.g_OH = e->SH; g.g_OW = e->SW;
    g.g_KH = g.g_KW = 7; g.g_stride = 2; g.g_pad = 3;
    g.C = GRD(e->stem_conv.w); g.c_ld = 147; g.c_atomic = 1;
    RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_WGRAD, 1, 0, st));
  }
  sc.need(sc.mark());   // join: everything after the backward (all-reduce, Adam) sees the side stream's gradients
  notify(e, sc, e->emb_hi, hi_mark);
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- EfficientNetV2 forward / backward
// conv (1x1 / 3x3 SAME) whose input is silu(bn_in(x_raw)) [* squeeze-excite gate]; statistics of the output
static int eff_conv_fwd(mmvqa_engine* e, hipStream_t st, const ConvRef& c, const float* x, const BNRef* bn_in,
                        const float* gate, int N, int H, int W, int OH, int OW, float* z, BNRef& bn_out, bool coef = true) {
  GemmParams g;
  memset(&g, 0, sizeof(g));
  g.M = N * OH * OW; g.N = c.Cout; g.K = c.KH * c.KH * c.Cin;
  g.A = x; g.a_ld = c.Cin;
  if (bn_in) {
    g.a_pro = gate ? PRO_SILU_GATE : PRO_AFFINE_SILU;
    g.a_c0 = WS(bn_in->scale); g.a_c1 = WS(bn_in->shift);
    g.gate = gate; g.gate_hw = OH * OW;
  }
  g.g_SH = H; g.g_SW = W; g.g_Cs = c.Cin; g.g_OH = OH; g.g_OW = OW;
  g.g_KH = g.g_KW = c.KH; g.g_stride = c.stride; g.g_pad = c.pad;
  g.B = PRM(c.w); g.b_ld = g.K;
  g.C = z; g.c_ld = c.Cout;
  if (e->training) { g.stat1 = stat_ptr(e, bn_out.stat_f); g.stat_bwd = 0; }
  set_sk(e, st, g);
  RUN(PROF_IGEMM, 2.0 * g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_FWD, 0, 0, st));
  return coef ? bn_coef_fwd(e, st, bn_out) : MMVQA_OK;   // (!coef: the consumer folds the coefficients itself)
}

static int eff_conv_wgrad(mmvqa_engine* e, hipStream_t st, const ConvRef& c, const float* G, const float* z,
                          const BNRef& bn_out, const float* x, const BNRef* bn_in, const float* gate, int N, int H,
                          int W, int OH, int OW) {
  GemmParams g;
  memset(&g, 0, sizeof(g));
  g.M = c.Cout; g.N = c.KH * c.KH * c.Cin; g.K = N * OH * OW;
  g.A = G; g.A2 = z; g.a_ld = c.Cout; g.a_pro = PRO_DZ;
  g.a_c0 = WS(bn_out.P); g.a_c1 = WS(bn_out.Q); g.a_c2 = WS(bn_out.R);
  g.B = x; g.b_ld = c.Cin;
  if (bn_in) {
    g.b_pro = gate ? PRO_SILU_GATE : PRO_AFFINE_SILU;
    g.b_c0 = WS(bn_in->scale); g.b_c1 = WS(bn_in->shift);
    g.gate = gate; g.gate_hw = OH * OW;
  }
  g.g_SH = H; g.g_SW = W; g.g_Cs = c.Cin; g.g_OH = OH; g.g_OW = OW;
  g.g_KH = g.g_KW = c.KH; g.g_stride = c.stride; g.g_pad = c.pad;
  g.C = GRD(c.w); g.c_ld = g.N; g.c_atomic = 1;
  RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_WGRAD, 0, 0, st));
  return MMVQA_OK;
}

static int effnet_forward(mmvqa_engine* e, hipStream_t st) {
  const mmvqa_model_desc& d = e->d;
  const int B = e->B;
  if (e->training)
    HIP_CHECK_RET(hipMemsetAsync(stat_ptr(e, 0), 0, e->statzone_floats / 2 * sizeof(float), st));
  HIP_CHECK_RET(hipMemsetAsync(WS(e->vis), 0, (size_t)5 * B * d.hidden * sizeof(float), st));
  const long M0 = (long)B * e->SH * e->SW;
  const bool fold = eff_folding(e);
  // out = act(bn(z)) + idn: with folding, this launch derives bn's coefficients from the raw sums and publishes them
  auto block_end = [&](BNRef& bn, const float* z, int act, const float* idn, float* out, long rows, int C, bool have_coef) -> int {
    if (fold && !have_coef) {
      mmvqa_bn_fold f;
      set_fold_fwd(e, f, bn);
      RUNB(HB_BN_ACT_ADD, (idn ? 12.0 : 8.0) * rows * C, k_bn_act_add_fold(st, z, &f, act, idn, nullptr, ACT_NONE, out, rows, C));
    } else {
      if (!have_coef) TRY(bn_coef_fwd(e, st, bn));
      RUNB(HB_BN_ACT_ADD, (idn ? 12.0 : 8.0) * rows * C, k_bn_act_add(st, z, WS(bn.scale), WS(bn.shift), act, idn, nullptr, nullptr, 0,
                                      ACT_NONE, out, rows, C));
    }
    return MMVQA_OK;
  };
  {  // conv_stem 3x3/2 SAME on the NCHW image
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = (int)M0; g.N = 24; g.K = 27;
    g.A = e->img; g.g_nchw = 1;
    g.g_SH = e->IH; g.g_SW = e->IW; g.g_Cs = 3; g.g_OH = e->SH; g.g_OW = e->SW;
    g.g_KH = g.g_KW = 3; g.g_stride = 2; g.g_pad = e->stem_conv.pad;
    g.B = PRM(e->stem_conv.w); g.b_ld = 27;
    g.C = WS(e->z0); g.c_ld = 24;
    if (e->training) g.stat1 = stat_ptr(e, e->stem_bn.stat_f);
    RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_FWD, 1, 0, st));
    TRY(block_end(e->stem_bn, WS(e->z0), ACT_SILU, nullptr, WS(e->eff_a0), M0, 24, false));
  }
  const float* x = WS(e->eff_a0);
  SideCtx sc(e, st);
  for (auto& b : e->eff) {
    const long Mout = (long)B * b.OH * b.OW;
    const float* idn = b.skip ? x : nullptr;
    if (b.type == 0) {
      TRY(eff_conv_fwd(e, st, b.c_a, x, nullptr, nullptr, B, b.H, b.W, b.OH, b.OW, WS(b.za), b.b_a, false));
      TRY(block_end(b.b_a, WS(b.za), ACT_SILU, idn, WS(b.out), Mout, b.cout, false));
    } else if (b.type == 1) {
      TRY(eff_conv_fwd(e, st, b.c_a, x, nullptr, nullptr, B, b.H, b.W, b.OH, b.OW, WS(b.za), b.b_a));
      TRY(eff_conv_fwd(e, st, b.c_p, WS(b.za), &b.b_a, nullptr, B, b.OH, b.OW, b.OH, b.OW, WS(b.zp), b.b_p, false));
      TRY(block_end(b.b_p, WS(b.zp), ACT_NONE, idn, WS(b.out), Mout, b.cout, false));
    } else {
      mmvqa_bn_fold fa, fdw;
      if (fold) { set_fold_fwd(e, fa, b.b_a); set_fold_fwd(e, fdw, b.b_dw); }
      TRY(eff_conv_fwd(e, st, b.c_a, x, nullptr, nullptr, B, b.H, b.W, b.H, b.W, WS(b.za), b.b_a, !fold));
      RUNB(HB_DWCONV_FWD, 4.0 * ((double)B * b.H * b.W + (double)Mout) * b.mid, k_dwconv_fwd(st, WS(b.za), WS(b.b_a.scale), WS(b.b_a.shift), PRM(b.dw_w), WS(b.zdw),
                                      e->training ? stat_ptr(e, b.b_dw.stat_f) : nullptr, B, b.H, b.W, b.mid, b.OH, b.OW,
                                      b.stride, b.pad, fold ? &fa : nullptr));
      if (!fold) TRY(bn_coef_fwd(e, st, b.b_dw));
      // squeeze-excite: gate = sigmoid(W_e silu(W_r mean_hw(a2) + b_r) + b_e)
      RUNB(HB_SE_POOL, 4.0 * Mout * b.mid, k_se_pool(st, WS(b.zdw), WS(b.b_dw.scale), WS(b.b_dw.shift), WS(b.pool), B, b.OH * b.OW, b.mid,
                                      fold ? &fdw : nullptr));
      RUN(PROF_MATRIX, 2.0 * B * b.mid * b.rd, k_skinny_fwd(st, WS(b.pool), b.mid, PRM(b.se_r.w), PRM(b.se_r.b), ACT_SILU, WS(b.rpre), WS(b.r),
                                      B, b.rd, b.mid));
      RUN(PROF_MATRIX, 2.0 * B * b.mid * b.rd, k_skinny_fwd(st, WS(b.r), b.rd, PRM(b.se_e.w), PRM(b.se_e.b), ACT_SIGMOID, WS(b.gpre),
                                      WS(b.gate), B, b.mid, b.rd));
      TRY(eff_conv_fwd(e, st, b.c_p, WS(b.zdw), &b.b_dw, WS(b.gate), B, b.OH, b.OW, b.OH, b.OW, WS(b.zp), b.b_p, false));
      TRY(block_end(b.b_p, WS(b.zp), ACT_NONE, idn, WS(b.out), Mout, b.cout, false));
    }
    x = WS(b.out);
    // the taps feed only the encoder: side stream, beside the chain of (mostly sub-chip-sized) launches
    if (b.feature >= 0) { sc.fork(); TRY(tap_fwd(e, sc.sd, b.feature, x, nullptr)); }
  }
  sc.need(sc.mark());   // join: the embedding reads the visual tokens
  return MMVQA_OK;
}

static int effnet_backward(mmvqa_engine* e, hipStream_t st) {
  const int B = e->B;
  HIP_CHECK_RET(hipMemsetAsync(stat_ptr(e, 0) + e->statzone_floats / 4, 0,
                               e->statzone_floats / 2 * sizeof(float), st));
  const int nb = (int)e->eff.size();
  // BatchNorm-backward coefficients folded where both consumers are elementwise kernels (the depthwise BatchNorm).  Folding
  // in the convolutions' weight / data gradients as the ResNet path does was measured and dropped here: these BatchNorms
  // keep 16 replicas of their sums and have up to 3072 channels, every workgroup of a data gradient folds all of them, and
  // the step got 0.2 ms slower than with the coefficient launches (config 3, same box).
  const bool fold = eff_folding(e);
  // Weight gradients (convolutions, depthwise, taps) only feed the optimizer: they run on the side stream beside the
  // data-gradient chain, whose launches are mostly smaller than the chip.  A side launch starts after everything the
  // caller's stream had queued when it was issued (fork) and leaves an event behind; the chain waits for that event before
  // it overwrites a buffer the side launch reads (G alternates between the two gbuf, gA / gB between two copies by block
  // parity, so the wait is for the launch of two blocks ago).
  SideCtx scx(e, st);
  hipStream_t sd = scx.sd;
  hipEvent_t ev_G[2] = {nullptr, nullptr}, ev_gA[2] = {nullptr, nullptr}, ev_gB[2] = {nullptr, nullptr};
  hipEvent_t ev_tap[4] = {nullptr, nullptr, nullptr, nullptr};
  long long hi_mark = e->enc_lo;
  // a block whose last op is a BatchNorm (er / ir) expects the BN-backward sums of that BN with its gradient
  auto producer_opts = [&](int i, EpiOpt& o) {
    if (i < 0) return;
    EffBlock& pb = e->eff[i];
    if (pb.type != 0) { o.st1 = &pb.b_p; o.Z1 = WS(pb.zp); }
  };
  int cur = 0;
  {
    EpiOpt o;
    producer_opts(nb - 1, o);
    TRY(tap_bwd(e, st, 4, WS(e->eff[nb - 1].out), nullptr, WS(e->gbuf[cur]), o));
  }
  // the other taps produce side gradients that are needed when the chain reaches their block (they share the `du`
  // scratch with the tap above, hence the fork after it)
  scx.fork();
  for (auto& b : e->eff)
    if (b.feature >= 0 && b.feature < 4) {
      TRY(tap_bwd(e, sd, b.feature, WS(b.out), nullptr, WS(e->tapgrad[b.feature]), EpiOpt()));
      ev_tap[b.feature] = scx.mark();
    }
  // (a stream of their own for these taps, as the ResNet path has, measured neutral here: 22.74 / 22.76 / 22.84 vs 22.67 / 22.75 / 22.72 ms)
  for (int i = nb - 1; i >= 0; --i) {
    EffBlock& b = e->eff[i];
    const float* G = WS(e->gbuf[cur]);
    const float* x = i == 0 ? WS(e->eff_a0) : WS(e->eff[i - 1].out);
    float* Gprev = WS(e->gbuf[cur ^ 1]);
    const float* extra = (i > 0 && e->eff[i - 1].feature >= 0) ? WS(e->tapgrad[e->eff[i - 1].feature]) : nullptr;
    if (extra && b.skip) return mmvqa_set_error(MMVQA_ERR_STATE, "effnet_backward: tap gradient into a skip block");
    const long Min = (long)B * b.H * b.W, Mout = (long)B * b.OH * b.OW;
    const int pi = i & 1;
    float* gA = WS(e->eff_gA[pi]);
    float* gB = WS(e->eff_gB[pi]);
    const float* dz_first = nullptr;   // gradient tensor feeding the block's first convolution
    if (b.type == 0) {
      // out = silu(bn(za)) + x
      scx.need(ev_gA[pi]);
      RUNB(HB_ACT_BWD_STATS, 12.0 * Mout * b.cout, k_act_bwd_stats(st, G, nullptr, nullptr, WS(b.za), WS(b.b_a.scale), WS(b.b_a.shift),
                                         WS(b.b_a.mean), WS(b.b_a.invstd), ACT_SILU, gA, stat_ptr(e, b.b_a.stat_b), Mout,
                                         b.OH * b.OW, b.cout));
      dz_first = gA;
    } else if (b.type == 1) {
      TRY(bn_coef_bwd(e, st, b.b_p));
      scx.fork();
      TRY(eff_conv_wgrad(e, sd, b.c_p, G, WS(b.zp), b.b_p, WS(b.za), &b.b_a, nullptr, B, b.OH, b.OW, b.OH, b.OW));
      ev_G[cur] = scx.mark();
      scx.need(ev_gA[pi]);
      EpiOpt o;
      o.Mk = WS(b.za); o.mk_ld = b.mid; o.mk_s = WS(b.b_a.scale); o.mk_b = WS(b.b_a.shift); o.mk_mode = 1;
      o.st1 = &b.b_a; o.Z1 = WS(b.za);
      TRY(conv_dgrad(e, st, b.c_p, G, WS(b.zp), b.b_p, B, b.OH, b.OW, b.OH, b.OW, gA, o));
      dz_first = gA;
    } else {
      TRY(bn_coef_bwd(e, st, b.b_p));
      scx.fork();
      TRY(eff_conv_wgrad(e, sd, b.c_p, G, WS(b.zp), b.b_p, WS(b.zdw), &b.b_dw, WS(b.gate), B, b.OH, b.OW, b.OH, b.OW));
      ev_G[cur] = scx.mark();
      scx.need(ev_gA[pi]);
      TRY(conv_dgrad(e, st, b.c_p, G, WS(b.zp), b.b_p, B, b.OH, b.OW, b.OH, b.OW, gA, EpiOpt()));   // t = d(a2*gate)
      // squeeze-excite backward
      float* dgate = WS(e->eff_se[0]); float* dpool = WS(e->eff_se[3]);
      RUNB(HB_SE_DGATE, 8.0 * Mout * b.mid, k_se_dgate(st, gA, WS(b.zdw), WS(b.b_dw.scale), WS(b.b_dw.shift), dgate, B, b.OH * b.OW, b.mid,
                                    WS(e->eff_separt), B * b.rd));
      RUN(PROF_MATRIX, 8.0 * B * b.mid * b.rd, k_se_fc_bwd(st, dgate, WS(b.gpre), WS(b.r), WS(b.rpre), WS(b.pool), PRM(b.se_e.w), PRM(b.se_r.w),
                                     GRD(b.se_e.w), GRD(b.se_e.b), GRD(b.se_r.w), GRD(b.se_r.b), dpool,
                                     WS(e->eff_separt), 1, B, b.mid, b.rd));
      // du2 = (t*gate + dpool/HW) * silu'(bn2(zdw)); BN2 sums
      scx.need(ev_gB[pi]);
      RUNB(HB_ACT_BWD_STATS, 12.0 * Mout * b.mid, k_act_bwd_stats(st, gA, WS(b.gate), dpool, WS(b.zdw), WS(b.b_dw.scale), WS(b.b_dw.shift),
                                         WS(b.b_dw.mean), WS(b.b_dw.invstd), ACT_SILU, gB, stat_ptr(e, b.b_dw.stat_b), Mout,
                                         b.OH * b.OW, b.mid));
      mmvqa_bn_fold fdw_w, fdw_d;
      if (fold) { set_fold_bwd(e, fdw_w, b.b_dw, false); set_fold_bwd(e, fdw_d, b.b_dw, true); }
      else TRY(bn_coef_bwd(e, st, b.b_dw));
      scx.fork();
      {
        hipStream_t st = sd;   // (RUNB times / launches on `st`)
        RUNB(HB_DWCONV_BWD_WEIGHT, 4.0 * (2.0 * Mout + (double)Min) * b.mid, k_dwconv_bwd_weight(st, gB, WS(b.zdw), WS(b.b_dw.P), WS(b.b_dw.Q), WS(b.b_dw.R), WS(b.za),
                                             WS(b.b_a.scale), WS(b.b_a.shift), GRD(b.dw_w), B, b.H, b.W, b.mid, b.OH, b.OW,
                                             b.stride, b.pad, fold ? &fdw_w : nullptr));
      }
      ev_gB[pi] = scx.mark();
      RUNB(HB_DWCONV_BWD_DATA, 4.0 * (2.0 * Mout + 2.0 * (double)Min) * b.mid, k_dwconv_bwd_data(st, gB, WS(b.zdw), WS(b.b_dw.P), WS(b.b_dw.Q), WS(b.b_dw.R), PRM(b.dw_w),
                                           WS(b.za), WS(b.b_a.scale), WS(b.b_a.shift), WS(b.b_a.mean), WS(b.b_a.invstd), gA,
                                           stat_ptr(e, b.b_a.stat_b), B, b.H, b.W, b.mid, b.OH, b.OW, b.stride, b.pad, fold ? &fdw_d : nullptr));
      dz_first = gA;
    }
    // first convolution of the block: weight gradient, then the gradient wrt the block input
    TRY(bn_coef_bwd(e, st, b.b_a));
    const int fo_h = b.type == 2 ? b.H : b.OH, fo_w = b.type == 2 ? b.W : b.OW;   // its output resolution
    scx.fork();
    TRY(eff_conv_wgrad(e, sd, b.c_a, dz_first, WS(b.za), b.b_a, x, nullptr, nullptr, B, b.H, b.W, fo_h, fo_w));
    ev_gA[pi] = scx.mark();
    EpiOpt o;
    producer_opts(i - 1, o);
    if (b.skip) { o.R = G; o.r_ld = b.cin; }
    else if (extra) { o.R = extra; o.r_ld = b.cin; scx.need(ev_tap[e->eff[i - 1].feature]); }
    scx.need(ev_G[cur ^ 1]);   // the buffer written next was the G of the block before: its side reader must be done
    TRY(conv_dgrad(e, st, b.c_a, dz_first, WS(b.za), b.b_a, B, b.H, b.W, fo_h, fo_w, Gprev, o));
    (void)Min;
    cur ^= 1;
    if (i > 0 && (nb - i) % 12 == 0) { notify(e, scx, b.c_a.w, hi_mark); hi_mark = b.c_a.w; }
  }
  // stem: du0 = G(a0) * silu'(bn1(z0)); BN sums; 3x3 weight gradient on the NCHW image
  float* g0 = WS(e->gbuf[cur ^ 1]);
  const long M0 = (long)B * e->SH * e->SW;
  scx.need(ev_G[cur ^ 1]);
  RUNB(HB_ACT_BWD_STATS, 12.0 * M0 * 24, k_act_bwd_stats(st, WS(e->gbuf[cur]), nullptr, nullptr, WS(e->z0), WS(e->stem_bn.scale),
                                     WS(e->stem_bn.shift), WS(e->stem_bn.mean), WS(e->stem_bn.invstd), ACT_SILU, g0,
                                     stat_ptr(e, e->stem_bn.stat_b), M0, e->SH * e->SW, 24));
  TRY(bn_coef_bwd(e, st, e->stem_bn));
  {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = 24; g.N = 27; g.K = (int)M0;
    g.A = g0; g.A2 = WS(e->z0); g.a_ld = 24; g.a_pro = PRO_DZ;
    g.a_c0 = WS(e->stem_bn.P); g.a_c1 = WS(e->stem_bn.Q); g.a_c2 = WS(e->stem_bn.R);
    g.B = e->img; g.g_nchw = 1;
    g.g_SH = e->IH; g.g_SW = e->IW; g.g_Cs = 3; g.g_OH = e->SH; g.g_OW = e->SW;
    g.g_KH = g.g_KW = 3; g.g_stride = 2; g.g_pad = e->stem_conv.pad;
    g.C = GRD(e->stem_conv.w); g.c_ld = 27; g.c_atomic = 1;
    RUN(PROF_IGEMM, 2.0 * (double)g.M * g.N * g.K, mmvqa_launch_igemm(g, KIND_WGRAD, 1, 0, st));
  }
  scx.need(scx.mark());   // join: everything after the backward (all-reduce, Adam) sees the side stream's gradients
  notify(e, scx, e->emb_hi, hi_mark);
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- encoders
static inline uint32_t site_seed(const mmvqa_engine* e, int layer, int site) {
  return e->seed + 0x9E3779B9u * (uint32_t)(layer * 8 + site + 1);
}

static int attn_call(mmvqa_engine* e, hipStream_t st, AttnParams& a, int head_dim, int bwd) {
  REG(REG_ATTN);
  const double fl = (bwd ? 10.0 : 4.0) * (double)e->B * a.heads * e->T * e->T * head_dim;
  RUN(PROF_ATTN, fl, mmvqa_launch_attention(a, head_dim, bwd, st));
  return MMVQA_OK;
}

// models/transformer.py:75-86 (pre-LN, norm1 for both sub-layers, shared by all layers)
static int bert_forward(mmvqa_engine* e, hipStream_t st, const float* x_in, const float** x_out) {
  REG(REG_ENC);
  const mmvqa_model_desc& d = e->d;
  const int H = d.hidden;
  const long M = (long)e->B * e->T;
  const float p = e->training ? d.p_drop : 0.f;
  const float* x = x_in;
  for (int i = 0; i < d.n_layers; ++i) {
    BertLayerRef& L = e->bert[i];
    TRY(ln_fwd(e, st, x, e->norm1, WS(L.xn1), WS(L.mean1), WS(L.rstd1), M, 1e-12f));
    static const bool fused_off = getenv("MMVQA_NO_FUSED_QKV") != nullptr;   // A/B switch
    if (!fused_off && k_qkv_attn_fwd_ok(e->T, H, d.heads)) {
      // projection + attention of every (sample, head) in one launch (qkvattn.hip): a profiler region of its own (the
      // north-star block = QKV products + attention, whichever launches carry them: bench.py adds the three regions)
      REG(REG_QKV_ATTN);
      RUN(PROF_ATTN, 2.0 * M * 3 * H * H + 4.0 * e->B * d.heads * (double)e->T * e->T * (H / d.heads),
          k_qkv_attn_fwd(st, WS(L.xn1), PRM(L.qkv.w), PRM(L.qkv.b), e->mask, WS(L.qkvo), WS(L.probs), WS(L.ctx), e->B, e->T,
                         H, d.heads, p, site_seed(e, i, 0)));
    } else {
    { REG(REG_QKV); TRY(lin_fwd(e, st, WS(L.xn1), H, M, L.qkv, WS(L.qkvo), 3 * H, ACT_NONE, nullptr, 0.f, 0, nullptr, 0)); }
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = WS(L.qkvo); a.k = a.q + H; a.v = a.q + 2 * H;
    a.row_stride = 3 * H; a.head_stride = H / d.heads;
    a.out = WS(L.ctx); a.out_row_stride = H; a.out_head_stride = H / d.heads;
    a.mask = e->mask; a.mask_on_query = 0; a.probs = WS(L.probs);
    a.B = e->B; a.T = e->T; a.heads = d.heads; a.sqrt_d = sqrtf((float)(H / d.heads));
    a.drop_p = p; a.seed = site_seed(e, i, 0);
    TRY(attn_call(e, st, a, H / d.heads, 0));
    }
    TRY(lin_fwd(e, st, WS(L.ctx), H, M, L.proj, WS(L.y), H, ACT_NONE, nullptr, p, site_seed(e, i, 1), x, H));
    TRY(ln_fwd(e, st, WS(L.y), e->norm1, WS(L.xn2), WS(L.mean2), WS(L.rstd2), M, 1e-12f));
    TRY(lin_fwd(e, st, WS(L.xn2), H, M, L.fc1, WS(L.h1), 4 * H, ACT_GELU, WS(L.pre1), 0.f, 0, nullptr, 0));
    TRY(lin_fwd(e, st, WS(L.h1), 4 * H, M, L.fc2, WS(L.z), H, ACT_NONE, nullptr, p, site_seed(e, i, 2), WS(L.y), H));
    x = WS(L.z);
  }
  *x_out = x;
  return MMVQA_OK;
}

// dz (in t_a) -> dx (left in t_a)
static int bert_backward(mmvqa_engine* e, hipStream_t st, const float* x_in, SideReads& sr) {
  REG(REG_ENC);
  const mmvqa_model_desc& d = e->d;
  const int H = d.hidden;
  const long M = (long)e->B * e->T;
  const float p = e->training ? d.p_drop : 0.f;
  float* dz = WS(e->t_a);
  for (int i = d.n_layers - 1; i >= 0; --i) {
    BertLayerRef& L = e->bert[i];
    const float* x = i == 0 ? x_in : WS(e->bert[i - 1].z);
    // FFN branch: z = y + drop(fc2(gelu(fc1(norm1(y)))))
    const float* dzd = dz;
    if (p > 0.f) {
      sr.write(WS(e->t_b));
      RUNB(HB_DROPOUT_COPY, 8.0 * M * H, k_dropout_copy(st, dz, WS(e->t_b), M * H, p, site_seed(e, i, 2)));
      dzd = WS(e->t_b);
    }
    TRY(lin_wgrad_side(e, sr, st, dzd, H, WS(L.h1), 4 * H, M, L.fc2, true));
    sr.write(WS(e->t_big));
    TRY(lin_dgrad(e, st, dzd, H, M, L.fc2, WS(e->t_big), 4 * H, ACT_GELU, WS(L.pre1), 4 * H, GRD(L.fc1.b), nullptr, 0));
    TRY(lin_wgrad_side(e, sr, st, WS(e->t_big), 4 * H, WS(L.xn2), H, M, L.fc1, false));
    sr.write(WS(e->t_c));
    TRY(lin_dgrad(e, st, WS(e->t_big), 4 * H, M, L.fc1, WS(e->t_c), H, 0, nullptr, 0, nullptr, nullptr, 0));
    // dy = LN'(dxn2) + dz
    sr.write(WS(e->t_d));
    TRY(ln_bwd(e, st, WS(e->t_c), WS(L.y), e->norm1, WS(L.mean2), WS(L.rstd2), dz, WS(e->t_d), M));
    float* dy = WS(e->t_d);
    // attention branch: y = x + drop(proj(attn(norm1(x))))
    const float* dyd = dy;
    if (p > 0.f) {
      sr.write(WS(e->t_b));
      RUNB(HB_DROPOUT_COPY, 8.0 * M * H, k_dropout_copy(st, dy, WS(e->t_b), M * H, p, site_seed(e, i, 1)));
      dyd = WS(e->t_b);
    }
    TRY(lin_wgrad_side(e, sr, st, dyd, H, WS(L.ctx), H, M, L.proj, true));
    sr.write(WS(e->t_c));
    TRY(lin_dgrad(e, st, dyd, H, M, L.proj, WS(e->t_c), H, 0, nullptr, 0, nullptr, nullptr, 0));  // dctx
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = WS(L.qkvo); a.k = a.q + H; a.v = a.q + 2 * H;
    a.row_stride = 3 * H; a.head_stride = H / d.heads;
    a.out_row_stride = H; a.out_head_stride = H / d.heads;
    a.mask = e->mask; a.mask_on_query = 0; a.probs = WS(L.probs);
    a.B = e->B; a.T = e->T; a.heads = d.heads; a.sqrt_d = sqrtf((float)(H / d.heads));
    a.drop_p = p; a.seed = site_seed(e, i, 0);
    a.dout = WS(e->t_c);
    float* dqkv = WS(e->t_big);
    a.dq = dqkv; a.dk = dqkv + H; a.dv = dqkv + 2 * H;
    sr.write(dqkv);
    TRY(attn_call(e, st, a, H / d.heads, 1));
    { REG(REG_QKV);
      TRY(lin_wgrad_side(e, sr, st, dqkv, 3 * H, WS(L.xn1), H, M, L.qkv, true));
      sr.write(WS(e->t_c));
      TRY(lin_dgrad(e, st, dqkv, 3 * H, M, L.qkv, WS(e->t_c), H, 0, nullptr, 0, nullptr, nullptr, 0)); }  // dxn1
    sr.write(dz);
    TRY(ln_bwd(e, st, WS(e->t_c), x, e->norm1, WS(L.mean1), WS(L.rstd1), dy, dz, M));
  }
  return MMVQA_OK;
}

// models/realformer.py:30-51 (post-LN, shared per-head kqv, residual scores, query-axis mask)
static int rf_forward(mmvqa_engine* e, hipStream_t st, const float* x_in, const float** x_out) {
  REG(REG_ENC);
  const mmvqa_model_desc& d = e->d;
  const int H = d.hidden, es = H / 8;
  const long M = (long)e->B * e->T;
  const float p = e->training ? d.p_rf_drop : 0.f;
  const float* x = x_in;
  for (int i = 0; i < d.n_layers; ++i) {
    RFLayerRef& L = e->rf[i];
    { REG(REG_QKV); TRY(lin_fwd(e, st, x, es, M * 8, L.kqv, WS(L.kqvo), 3 * es, ACT_NONE, nullptr, 0.f, 0, nullptr, 0)); }
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.k = WS(L.kqvo); a.q = a.k + es; a.v = a.k + 2 * es;
    a.row_stride = 8 * 3 * es; a.head_stride = 3 * es;
    a.out = WS(L.res); a.out_row_stride = H; a.out_head_stride = es;
    a.mask = e->mask; a.mask_on_query = 1;
    a.prev_in = i > 0 ? WS(e->rf[i - 1].prev) : nullptr;
    a.prev_out = WS(L.prev); a.probs = WS(L.probs);
    a.B = e->B; a.T = e->T; a.heads = 8; a.sqrt_d = sqrtf((float)es);
    TRY(attn_call(e, st, a, es, 0));
    // s1 = x + drop(proj(res)); x1 = ln1(s1)
    TRY(lin_fwd(e, st, WS(L.res), H, M, L.proj, WS(L.s1), H, ACT_NONE, nullptr, p, site_seed(e, i, 1), x, H));
    TRY(ln_fwd(e, st, WS(L.s1), L.ln1, WS(L.x1), WS(L.mean1), WS(L.rstd1), M, 1e-5f));
    TRY(lin_fwd(e, st, WS(L.x1), H, M, L.ff0, WS(L.hact), 4 * H, ACT_SERF, WS(L.pre), 0.f, 0, nullptr, 0));
    TRY(lin_fwd(e, st, WS(L.hact), 4 * H, M, L.ff2, WS(L.s2), H, ACT_NONE, nullptr, p, site_seed(e, i, 2), WS(L.x1), H));
    TRY(ln_fwd(e, st, WS(L.s2), L.ln2, WS(L.x2), WS(L.mean2), WS(L.rstd2), M, 1e-5f));
    x = WS(L.x2);
  }
  *x_out = x;
  return MMVQA_OK;
}

static int rf_backward(mmvqa_engine* e, hipStream_t st, const float* x_in, SideReads& sr) {
  REG(REG_ENC);
  const mmvqa_model_desc& d = e->d;
  const int H = d.hidden, es = H / 8;
  const long M = (long)e->B * e->T;
  const float p = e->training ? d.p_rf_drop : 0.f;
  float* dx2 = WS(e->t_a);
  for (int i = d.n_layers - 1; i >= 0; --i) {
    RFLayerRef& L = e->rf[i];
    const float* x = i == 0 ? x_in : WS(e->rf[i - 1].x2);
    // x2 = ln2(s2), s2 = x1 + drop(ff2(serf(ff0(x1))))
    sr.write(WS(e->t_d));
    TRY(ln_bwd(e, st, dx2, WS(L.s2), L.ln2, WS(L.mean2), WS(L.rstd2), nullptr, WS(e->t_d), M));
    float* ds2 = WS(e->t_d);
    const float* dff = ds2;
    if (p > 0.f) {
      sr.write(WS(e->t_b));
      RUNB(HB_DROPOUT_COPY, 8.0 * M * H, k_dropout_copy(st, ds2, WS(e->t_b), M * H, p, site_seed(e, i, 2)));
      dff = WS(e->t_b);
    }
    TRY(lin_wgrad_side(e, sr, st, dff, H, WS(L.hact), 4 * H, M, L.ff2, true));
    sr.write(WS(e->t_big));
    TRY(lin_dgrad(e, st, dff, H, M, L.ff2, WS(e->t_big), 4 * H, ACT_SERF, WS(L.pre), 4 * H, GRD(L.ff0.b), nullptr, 0));
    TRY(lin_wgrad_side(e, sr, st, WS(e->t_big), 4 * H, WS(L.x1), H, M, L.ff0, false));
    sr.write(WS(e->t_c));
    TRY(lin_dgrad(e, st, WS(e->t_big), 4 * H, M, L.ff0, WS(e->t_c), H, 0, nullptr, 0, nullptr, ds2, H));  // dx1 total
    // x1 = ln1(s1), s1 = x + drop(proj(res))
    sr.write(WS(e->t_d));
    TRY(ln_bwd(e, st, WS(e->t_c), WS(L.s1), L.ln1, WS(L.mean1), WS(L.rstd1), nullptr, WS(e->t_d), M));
    float* ds1 = WS(e->t_d);
    const float* dr = ds1;
    if (p > 0.f) {
      sr.write(WS(e->t_b));
      RUNB(HB_DROPOUT_COPY, 8.0 * M * H, k_dropout_copy(st, ds1, WS(e->t_b), M * H, p, site_seed(e, i, 1)));
      dr = WS(e->t_b);
    }
    TRY(lin_wgrad_side(e, sr, st, dr, H, WS(L.res), H, M, L.proj, false));
    sr.write(WS(e->t_c));
    TRY(lin_dgrad(e, st, dr, H, M, L.proj, WS(e->t_c), H, 0, nullptr, 0, nullptr, nullptr, 0));  // dres
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.k = WS(L.kqvo); a.q = a.k + es; a.v = a.k + 2 * es;
    a.row_stride = 8 * 3 * es; a.head_stride = 3 * es;
    a.out_row_stride = H; a.out_head_stride = es;
    a.mask = e->mask; a.mask_on_query = 1; a.probs = WS(L.probs);
    a.B = e->B; a.T = e->T; a.heads = 8; a.sqrt_d = sqrtf((float)es);
    a.dout = WS(e->t_c);
    float* dkqv = WS(e->t_big);
    a.dk = dkqv; a.dq = dkqv + es; a.dv = dkqv + 2 * es;
    a.dprev_in = (i < d.n_layers - 1) ? WS(e->t_dprev[(i + 1) & 1]) : nullptr;
    a.dprev_out = i > 0 ? WS(e->t_dprev[i & 1]) : nullptr;
    sr.write(dkqv);
    TRY(attn_call(e, st, a, es, 1));
    { REG(REG_QKV);
      TRY(lin_wgrad_side(e, sr, st, dkqv, 3 * es, x, es, M * 8, L.kqv, false));
      sr.write(dx2);
      TRY(lin_dgrad(e, st, dkqv, 3 * es, M * 8, L.kqv, dx2, es, 0, nullptr, 0, nullptr, ds1, es)); }  // dx total
  }
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- heads (models/mmbert.py:150-167)
static int heads_forward(mmvqa_engine* e, hipStream_t st, const float* h) {
  REG(REG_HEAD);
  const mmvqa_model_desc& d = e->d;
  const int H = d.hidden;
  const long M = (long)e->B * e->T;
  const float* hin = h;
  long HM = M;
  if (d.head_kind == 1) {
    RUN(PROF_OTHER, 0, k_meanpool_fwd(st, h, e->mask, WS(e->hd_pool), e->B, e->T, H));
    hin = WS(e->hd_pool);
    HM = e->B;
  }
  TRY(lin_fwd(e, st, hin, H, HM, e->fc1, WS(e->hd_u), H, ACT_SERF, WS(e->hd_upre), 0.f, 0, nullptr, 0));
  TRY(lin_fwd(e, st, WS(e->hd_u), H, HM, e->cls0, WS(e->hd_c0), H, ACT_NONE, nullptr, 0.f, 0, nullptr, 0));
  TRY(ln_fwd(e, st, WS(e->hd_c0), e->cls_ln, WS(e->hd_c1), WS(e->hd_mean), WS(e->hd_rstd), HM, 1e-12f));
  TRY(lin_fwd(e, st, WS(e->hd_c1), H, HM, e->cls2, e->logits, e->logits_ld, ACT_NONE, nullptr, 0.f, 0, nullptr, 0));
  if (d.supcon && e->feat) {
    RUN(PROF_OTHER, 0, k_meanpool_fwd(st, h, e->mask, WS(e->sc_pool), e->B, e->T, H));
    TRY(lin_fwd(e, st, WS(e->sc_pool), H, e->B, e->head0, WS(e->sc_a), H, ACT_SERF, WS(e->sc_pre), 0.f, 0, nullptr, 0));
    TRY(lin_fwd(e, st, WS(e->sc_a), H, e->B, e->head2, WS(e->sc_f), d.feat_dim, ACT_NONE, nullptr, 0.f, 0, nullptr, 0));
    // the normalised features stay in the engine's workspace for the backward pass: the caller may drop its
    // `feat` tensor before backward (supcon_utils.py:283-284 rebinds it to split_feat(feat))
    RUN(PROF_OTHER, 0, k_l2norm_fwd(st, WS(e->sc_f), WS(e->sc_y), WS(e->sc_nrm), e->B, d.feat_dim));
    HIP_CHECK_RET(hipMemcpyAsync(e->feat, WS(e->sc_y), (size_t)e->B * d.feat_dim * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
  }
  return MMVQA_OK;
}

// produces dh (gradient wrt the encoder output) in t_a
static int heads_backward(mmvqa_engine* e, hipStream_t st, const float* h, const float* dlogits, int dl_ld,
                          const float* dfeat, SideReads& sr) {
  REG(REG_HEAD);
  const mmvqa_model_desc& d = e->d;
  const int H = d.hidden;
  const long M = (long)e->B * e->T;
  const long HM = d.head_kind == 1 ? e->B : M;
  const float* hin = d.head_kind == 1 ? WS(e->hd_pool) : h;
  {  // classifier[2]: logits = c1 W^T + b  (beside the data gradient of the same layer: both read dlogits, nobody writes it)
    hipStream_t st_main = st;
    if (sr.on) sr.sc.fork();
    hipStream_t st = sr.on ? sr.sc.sd : st_main;
    GemmParams g = gp_linear_geom();
    g.M = d.n_classes; g.N = H; g.K = (int)HM;
    g.A = dlogits; g.a_ld = dl_ld;
    g.B = WS(e->hd_c1); g.b_ld = H; g.g_Cs = H;
    g.C = GRD(e->cls2.w); g.c_ld = H; g.c_atomic = 1;
    RUN(PROF_IGEMM, 2.0 * HM * d.n_classes * H, mmvqa_launch_igemm(g, KIND_WGRAD, 0, 0, st));
    RUN(PROF_OTHER, 0, k_colsum(st, dlogits, dl_ld, (int)HM, d.n_classes, GRD(e->cls2.b)));
  }
  TRY(lin_dgrad(e, st, dlogits, dl_ld, HM, e->cls2, WS(e->t_b), H, 0, nullptr, 0, nullptr, nullptr, 0));  // dc1
  TRY(ln_bwd(e, st, WS(e->t_b), WS(e->hd_c0), e->cls_ln, WS(e->hd_mean), WS(e->hd_rstd), nullptr, WS(e->t_c), HM));
  TRY(lin_wgrad_side(e, sr, st, WS(e->t_c), H, WS(e->hd_u), H, HM, e->cls0, true));
  TRY(lin_dgrad(e, st, WS(e->t_c), H, HM, e->cls0, WS(e->t_b), H, ACT_SERF, WS(e->hd_upre), H, GRD(e->fc1.b), nullptr, 0));
  TRY(lin_wgrad_side(e, sr, st, WS(e->t_b), H, hin, H, HM, e->fc1, false));
  float* dh = WS(e->t_a);
  if (d.head_kind == 1) {
    sr.write(WS(e->t_c));
    TRY(lin_dgrad(e, st, WS(e->t_b), H, HM, e->fc1, WS(e->t_c), H, 0, nullptr, 0, nullptr, nullptr, 0));
    RUN(PROF_OTHER, 0, k_meanpool_bwd(st, WS(e->t_c), e->mask, dh, e->B, e->T, H, 0));
  } else {
    TRY(lin_dgrad(e, st, WS(e->t_b), H, HM, e->fc1, dh, H, 0, nullptr, 0, nullptr, nullptr, 0));
  }
  if (d.supcon && dfeat) {
    sr.write(WS(e->t_b)); sr.write(WS(e->t_c));
    float* df = WS(e->t_b);  // [B][feat_dim]
    RUN(PROF_OTHER, 0, k_l2norm_bwd(st, dfeat, WS(e->sc_y), WS(e->sc_nrm), df, e->B, d.feat_dim));
    TRY(lin_wgrad(e, st, df, d.feat_dim, WS(e->sc_a), H, e->B, e->head2, true));
    TRY(lin_dgrad(e, st, df, d.feat_dim, e->B, e->head2, WS(e->t_c), H, ACT_SERF, WS(e->sc_pre), H, GRD(e->head0.b), nullptr, 0));
    TRY(lin_wgrad(e, st, WS(e->t_c), H, WS(e->sc_pool), H, e->B, e->head0, false));
    TRY(lin_dgrad(e, st, WS(e->t_c), H, e->B, e->head0, WS(e->t_d), H, 0, nullptr, 0, nullptr, nullptr, 0));
    RUN(PROF_OTHER, 0, k_meanpool_bwd(st, WS(e->t_d), e->mask, dh, e->B, e->T, H, 1));
  }
  return MMVQA_OK;
}

// --------------------------------------------------------------------------- whole model
// State that lives in the caller's workspace ACROSS calls (mmvqa.h, mmvqa_engine_bind): the tap-validity tables of the
// 3x3 weight gradients and the (zero) arrival tickets of the persistent launches.  Put in place on the caller's stream
// by the first forward after a bind, i.e. before the side stream exists for that workspace.
static int prepare_workspace(mmvqa_engine* e, hipStream_t st) {
  if (e->ws_ready) return MMVQA_OK;
  for (auto& kv : e->pixmask_off) {
    const mmvqa_engine::PixGeom& q = kv.second;
    TRY(k_pixmask(st, reinterpret_cast<int*>(WS(q.off)), q.N, q.OH, q.OW, q.H, q.W, q.KH, q.KH, q.stride, q.pad));
  }
  for (int i = 0; i < 3; ++i) HIP_CHECK_RET(hipMemsetAsync(WS(e->sk_cnt[i]), 0, SK_CNT_N * sizeof(unsigned int), st));
  e->ws_ready = true;
  return MMVQA_OK;
}

int engine_forward(mmvqa_engine* e, hipStream_t st, const float* img, const long long* ids, const long long* seg,
                   const long long* mask, float* logits, int logits_ld, float* feat, int training, uint32_t seed) {
  if (!e->planned || !e->bound) return mmvqa_set_error(MMVQA_ERR_STATE, "engine_forward: plan/bind first");
  TRY(prepare_workspace(e, st));
  const mmvqa_model_desc& d = e->d;
  if (logits_ld < d.n_classes || (logits_ld & 3))
    return mmvqa_set_error(MMVQA_ERR_ARG, "engine_forward: logits_ld=%d must be >= n_classes and %%4==0", logits_ld);
  e->img = img; e->ids = ids; e->seg = seg; e->mask = mask;
  e->logits = logits; e->logits_ld = logits_ld; e->feat = feat;
  e->training = training; e->seed = seed;
  struct TunerScope { TunerScope(IgemmTuner* t) { mmvqa_set_tuner(t); } ~TunerScope() { mmvqa_set_tuner(nullptr); } } ts(&e->tuner);
  e->ev_next = 0;
  if (d.cnn == 1) TRY(effnet_forward(e, st)); else TRY(resnet_forward(e, st));
  const float pe = training ? d.p_emb_drop : 0.f;
  e->prof_reg = REG_EMBED;
  RUN(PROF_OTHER, 0,
      k_embed_fwd(st, ids, seg, PRM(e->emb_word), PRM(e->emb_pos), PRM(e->emb_type), PRM(e->emb_ln.g),
                  PRM(e->emb_ln.b), WS(e->vis), WS(e->emb_out), WS(e->emb_xhat), WS(e->emb_rstd), e->B, e->T,
                  d.hidden, d.num_vis, 1e-12f, pe, site_seed(e, 100, 0)));
  e->prof_reg = REG_BACKBONE;
  const float* h = nullptr;
  if (d.encoder == 0) TRY(bert_forward(e, st, WS(e->emb_out), &h));
  else TRY(rf_forward(e, st, WS(e->emb_out), &h));
  e->enc_out_final = (size_t)(h - e->ws);
  return heads_forward(e, st, h);
}

int engine_backward(mmvqa_engine* e, hipStream_t st, const float* dlogits, int dl_ld, const float* dfeat) {
  if (!e->planned || !e->bound || !e->img) return mmvqa_set_error(MMVQA_ERR_STATE, "engine_backward: run forward first");
  const mmvqa_model_desc& d = e->d;
  struct TunerScope { TunerScope(IgemmTuner* t) { mmvqa_set_tuner(t); } ~TunerScope() { mmvqa_set_tuner(nullptr); } } ts(&e->tuner);
  e->ev_next = 0;
  const float* h = WS(e->enc_out_final);
  SideCtx sc_enc(e, st);
  SideReads sr(sc_enc);
  TRY(heads_backward(e, st, h, dlogits, dl_ld, dfeat, sr));
  if (d.encoder == 0) TRY(bert_backward(e, st, WS(e->emb_out), sr));
  else TRY(rf_backward(e, st, WS(e->emb_out), sr));
  if (e->grad_cb) {
    sr.join();   // the weight gradients of the range may still be queued on the side stream
    e->grad_cb(e->grad_cb_user, e->enc_lo, e->n_params);   // heads + encoder gradients are final
  }
  const float pe = e->training ? d.p_emb_drop : 0.f;
  e->prof_reg = REG_EMBED;
  RUN(PROF_OTHER, 0,
      k_embed_bwd(st, WS(e->t_a), e->ids, e->seg, WS(e->emb_xhat), WS(e->emb_rstd), PRM(e->emb_ln.g),
                  GRD(e->emb_word), GRD(e->emb_pos), GRD(e->emb_type), GRD(e->emb_ln.g), GRD(e->emb_ln.b),
                  WS(e->dvis), e->B, e->T, d.hidden, d.num_vis, pe, site_seed(e, 100, 0), 0));
  e->prof_reg = REG_BACKBONE;
  if (e->grad_cb) e->grad_cb(e->grad_cb_user, 0, e->emb_hi);            // embedding tables + LayerNorm
  return d.cnn == 1 ? effnet_backward(e, st) : resnet_backward(e, st);
}

int engine_create(const mmvqa_model_desc* desc, mmvqa_engine** out) {
  if (!desc || !out) return mmvqa_set_error(MMVQA_ERR_ARG, "engine_create: null argument");
  const mmvqa_model_desc& d = *desc;
  if (d.hidden % 8 != 0 || d.hidden > 1024 || d.n_layers < 1 || d.num_vis != 5)
    return mmvqa_set_error(MMVQA_ERR_ARG, "engine_create: hidden=%d n_layers=%d num_vis=%d unsupported", d.hidden,
                           d.n_layers, d.num_vis);
  if (d.encoder == 0 && (d.heads < 1 || d.hidden % d.heads != 0))
    return mmvqa_set_error(MMVQA_ERR_ARG, "engine_create: heads=%d does not divide hidden=%d", d.heads, d.hidden);
  if (d.cnn == 0 && d.resnet_width % 8 != 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "engine_create: resnet_width=%d must be a multiple of 8", d.resnet_width);
  mmvqa_engine* e = new mmvqa_engine();
  e->d = d;
  if (getenv("MMVQA_NO_SIDE_STREAM")) e->use_side = 0;   // A/B switch: everything on the caller's stream
  if (getenv("MMVQA_NO_BN_FOLD")) e->bn_fold = 0;        // A/B switch: separate BatchNorm coefficient launches everywhere
  memset(e->prof_launch, 0, sizeof(e->prof_launch));
  memset(e->prof_ms, 0, sizeof(e->prof_ms));
  memset(e->prof_flops, 0, sizeof(e->prof_flops));
  memset(e->reg_launch, 0, sizeof(e->reg_launch));
  memset(e->reg_ms, 0, sizeof(e->reg_ms));
  memset(e->reg_flops, 0, sizeof(e->reg_flops));
  int r = build_tables(e);
  if (r != MMVQA_OK) { delete e; return r; }
  *out = e;
  return MMVQA_OK;
}

int engine_profile_collect(mmvqa_engine* e) {
  for (auto& r : e->prof) {
    HIP_CHECK_RET(hipEventSynchronize(r.b));
    float ms = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&ms, r.a, r.b));
    e->prof_launch[r.cls] += 1;
    e->prof_ms[r.cls] += ms;
    e->prof_flops[r.cls] += r.flops;
    e->reg_launch[r.reg][r.cls] += 1;
    e->reg_ms[r.reg][r.cls] += ms;
    e->reg_flops[r.reg][r.cls] += r.flops;
    if (r.tag > HB_NONE && r.tag < HB_N) { e->tag_launch[r.tag] += 1; e->tag_ms[r.tag] += ms; e->tag_bytes[r.tag] += r.bytes; }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  e->prof.clear();
  return MMVQA_OK;
}
