// Device input pipeline: the image transforms of the reference's data layer
//   pretrain/roco_train.py:98-112, vqamed2019/train.py:179-200
//     Resize(224) -> CenterCrop(224) -> RandomResizedCrop -> RandomRotation -> ColorJitter -> ToTensor -> Normalize
// on decoded uint8 RGB images resident in HBM.  The reference runs them per sample on the host through
// torchvision's PIL backend, i.e. the pixel arithmetic is Pillow's C code; these kernels restate that arithmetic
// bit for bit (byte / integer work: the bar is bit-exact, tests/test_augment.py compares with Pillow itself):
//   resample   ImagingResample (src/libImaging/Resample.c): separable convolution with the bilinear (triangle)
//              filter whose support grows with the down-scale factor (anti-aliasing), coefficients normalised
//              and quantised to 22 fractional bits, 32-bit integer accumulation, horizontal pass then vertical
//   rotate     Image.rotate(..., NEAREST) -> ImagingTransformAffine -> affine_fixed (src/libImaging/Geometry.c):
//              16.16 fixed-point inverse mapping, black fill
//   colour     ImageEnhance.{Brightness, Contrast, Color} = ImagingBlend with a degenerate image (Blend.c: fp32
//              interpolate, truncate / clip), L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (Convert.c),
//              hue = RGB -> HSV -> H += shift (uint8 wrap) -> RGB (Convert.c rgb2hsv_row / hsv2rgb, double maths)
//   to_tensor  ToTensor (x / 255) + Normalize ((x - mean) / std) into fp32 NCHW
// All of it is HBM-bound byte work: one thread per output pixel (3 channels), rows contiguous across lanes.
// This file is compiled with -ffp-contract=off: Pillow's results depend on separately rounded multiplies and adds.
#include <math.h>

#include "common.h"

typedef unsigned char u8;

// --------------------------------------------------------------------------- coefficients (host, Resample.c precompute_coeffs)
// Bilinear filter, support 1.  bounds[2*i] = first source index, bounds[2*i+1] = tap count of output i;
// kk[i*ksize + k] = round(w_k * 2^22).  Returns ksize (taps allocated per output), or a negative error.
extern "C" int mmvqa_resample_coeffs(int in_size, double in0, double in1, int out_size, int* bounds, int* kk,
                                     int ksize_cap) {
  if (in_size <= 0 || out_size <= 0 || !(in1 > in0)) return mmvqa_set_error(MMVQA_ERR_ARG, "resample_coeffs: bad sizes");
  const double scale = (in1 - in0) / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (!bounds || !kk) return ksize;                       // size query
  if (ksize > ksize_cap) return mmvqa_set_error(MMVQA_ERR_ARG, "resample_coeffs: ksize %d > cap %d", ksize, ksize_cap);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double w[64];
    if (ksize > 64) return mmvqa_set_error(MMVQA_ERR_ARG, "resample_coeffs: down-scale factor too large (ksize %d)", ksize);
    int x = 0;
    for (; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double v = a < 1.0 ? 1.0 - a : 0.0;
      w[x] = v;
      ww += v;
    }
    for (x = 0; x < xmax; ++x)
      if (ww != 0.0) w[x] /= ww;
    for (; x < ksize; ++x) w[x] = 0.0;
    for (x = 0; x < ksize; ++x)
      kk[xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << 22)) : (int)(0.5 + w[x] * (1 << 22));
    bounds[xx * 2] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}

// --------------------------------------------------------------------------- resample
__device__ __forceinline__ u8 clip8_22(int v) {
  v >>= 22;
  return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass of job blockIdx.z: tmp[y - ty0][x][c] for y in [ty0, ty0 + tyn), x in [0, out_w)
__global__ __launch_bounds__(256) void resample_h_kernel(const mmvqa_resample_job* __restrict__ jobs, int out_w) {
  const mmvqa_resample_job j = jobs[blockIdx.z];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), yl = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= out_w || yl >= j.tyn) return;
  const int xo = j.ox + x;                           // column of the resized image
  const int xmin = j.hb[xo * 2], xn = j.hb[xo * 2 + 1];
  const int* k = j.hk + (size_t)xo * j.hks;
  const u8* row = j.src + (size_t)(j.by + j.ty0 + yl) * j.spitch + (size_t)(j.bx + xmin) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int t = 0; t < xn; ++t) {
    const int c = k[t];
    s0 += (int)row[t * 3] * c; s1 += (int)row[t * 3 + 1] * c; s2 += (int)row[t * 3 + 2] * c;
  }
  u8* o = j.tmp + ((size_t)yl * out_w + x) * 3;
  o[0] = clip8_22(s0); o[1] = clip8_22(s1); o[2] = clip8_22(s2);
}

// vertical pass: dst[y][x][c] for y in [0, out_h), x in [0, out_w)
__global__ __launch_bounds__(256) void resample_v_kernel(const mmvqa_resample_job* __restrict__ jobs, int out_h,
                                                         int out_w) {
  const mmvqa_resample_job j = jobs[blockIdx.z];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= out_w || y >= out_h) return;
  const int yo = j.oy + y;
  const int ymin = j.vb[yo * 2], yn = j.vb[yo * 2 + 1];
  const int* k = j.vk + (size_t)yo * j.vks;
  const u8* col = j.tmp + ((size_t)(ymin - j.ty0) * out_w + x) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int t = 0; t < yn; ++t) {
    const int c = k[t];
    const u8* p = col + (size_t)t * out_w * 3;
    s0 += (int)p[0] * c; s1 += (int)p[1] * c; s2 += (int)p[2] * c;
  }
  u8* o = j.dst + (size_t)y * j.dpitch + (size_t)x * 3;
  o[0] = clip8_22(s0); o[1] = clip8_22(s1); o[2] = clip8_22(s2);
}

// --------------------------------------------------------------------------- rotate (affine_fixed, nearest)
// fix[b][6] = {a0, a1, a2, a3, a4, a5} in 16.16 fixed point (host: FIX(v) = floor-or-truncate(v * 65536 + 0.5))
__global__ __launch_bounds__(256) void rotate_nearest_kernel(const u8* __restrict__ src, u8* __restrict__ dst,
                                                             const int* __restrict__ fix, int H, int W) {
  const int b = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int* a = fix + b * 6;
  const int xx = a[2] + a[1] * y + a[0] * x, yy = a[5] + a[4] * y + a[3] * x;   // (wraps like the C int arithmetic)
  const int xin = xx >> 16, yin = yy >> 16;
  const size_t img = (size_t)b * H * W * 3;
  u8* o = dst + img + ((size_t)y * W + x) * 3;
  if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
    const u8* p = src + img + ((size_t)yin * W + xin) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  } else {
    o[0] = 0; o[1] = 0; o[2] = 0;
  }
}

// --------------------------------------------------------------------------- colour jitter
__device__ __forceinline__ int lum(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// sums[b] += sum of L over the pixels of image b (exact integer), for ImageStat.Stat(convert("L")).mean
__global__ __launch_bounds__(256) void lsum_kernel(const u8* __restrict__ img, unsigned long long* __restrict__ sums,
                                                   int npix) {
  const int b = blockIdx.y;
  const u8* p = img + (size_t)b * npix * 3;
  unsigned int s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) s += (unsigned)lum(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
  __shared__ unsigned int red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[b], (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}

// ImagingBlend(degenerate, image, alpha) for one byte
__device__ __forceinline__ u8 blend1(int deg, int v, float alpha, bool inside) {
  const float t = (float)deg + alpha * (float)(v - deg);     // separately rounded (no contraction in this file)
  if (inside) return (u8)(int)t;
  if (t <= 0.0f) return 0;
  if (t >= 255.0f) return 255;
  return (u8)(int)t;
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
  float h;
  if (r == maxc) h = bc - gc;
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
  uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
  us = is < 0 ? 0 : (is > 255 ? 255 : is);
}
__device__ __forceinline__ int round_half_away(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5)); }
__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) { r = g = b = v; return; }
  const double hf = (double)h * 6.0 / 255.0;
  const int i = (int)floor(hf);
  const float f = (float)(hf - (double)i);
  const float fs = (float)((double)s / 255.0);
  int p = round_half_away((double)v * (1.0 - (double)fs));
  int q = round_half_away((double)v * (1.0 - (double)fs * (double)f));
  int t = round_half_away((double)v * (1.0 - (double)fs * (1.0 - (double)f)));
  p = p < 0 ? 0 : (p > 255 ? 255 : p); q = q < 0 ? 0 : (q > 255 ? 255 : q); t = t < 0 ? 0 : (t > 255 ? 255 : t);
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// One round of ColorJitter: image b applies op[b] (0 brightness, 1 contrast, 2 saturation, 3 hue, <0 nothing) with
// factor[b]; contrast uses mean[b] = int(Lsum / npix + 0.5) of the image as it stands (lsum_kernel just before).
__global__ __launch_bounds__(256) void jitter_kernel(u8* __restrict__ img, const int* __restrict__ op,
                                                     const float* __restrict__ factor,
                                                     const unsigned long long* __restrict__ lsums, int npix) {
  const int b = blockIdx.y, o = op[b];
  if (o < 0) return;
  const float f = factor[b];
  const bool inside = f >= 0.0f && f <= 1.0f;
  u8* p = img + (size_t)b * npix * 3;
  int mean = 0, shift = 0;
  if (o == 1) mean = (int)((double)lsums[b] / (double)npix + 0.5);
  if (o == 3) shift = ((int)f) & 0xFF;                  // the host passes np.uint8(hue_factor * 255) itself (integer-valued)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    int r = p[i * 3], g = p[i * 3 + 1], bb = p[i * 3 + 2];
    if (o == 0) {
      r = blend1(0, r, f, inside); g = blend1(0, g, f, inside); bb = blend1(0, bb, f, inside);
    } else if (o == 1) {
      r = blend1(mean, r, f, inside); g = blend1(mean, g, f, inside); bb = blend1(mean, bb, f, inside);
    } else if (o == 2) {
      const int L = lum(r, g, bb);
      r = blend1(L, r, f, inside); g = blend1(L, g, f, inside); bb = blend1(L, bb, f, inside);
    } else {
      int h, s, v;
      rgb2hsv(r, g, bb, h, s, v);
      h = (h + shift) & 0xFF;
      hsv2rgb(h, s, v, r, g, bb);
    }
    p[i * 3] = (u8)r; p[i * 3 + 1] = (u8)g; p[i * 3 + 2] = (u8)bb;
  }
}

// --------------------------------------------------------------------------- ToTensor + Normalize -> fp32 NCHW
__global__ __launch_bounds__(256) void to_tensor_kernel(const u8* __restrict__ img, float* __restrict__ out, int npix,
                                                        float m0, float m1, float m2, float s0, float s1, float s2) {
  const int b = blockIdx.y;
  const u8* p = img + (size_t)b * npix * 3;
  float* o = out + (size_t)b * npix * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    o[i] = ((float)p[i * 3] / 255.0f - m0) / s0;
    o[npix + i] = ((float)p[i * 3 + 1] / 255.0f - m1) / s1;
    o[2 * npix + i] = ((float)p[i * 3 + 2] / 255.0f - m2) / s2;
  }
}

// --------------------------------------------------------------------------- C ABI
extern "C" {

size_t mmvqa_sizeof_resample_job(void) { return sizeof(mmvqa_resample_job); }

int mmvqa_aug_resample(mmvqa_stream_t s, const mmvqa_resample_job* jobs_dev, int njobs, int max_tmp_rows, int out_h,
                       int out_w) {
  if (!jobs_dev || njobs <= 0 || out_h <= 0 || out_w <= 0 || max_tmp_rows <= 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "aug_resample: bad arguments");
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(resample_h_kernel, dim3((out_w + 63) / 64, (max_tmp_rows + 3) / 4, njobs), dim3(256), 0, st, jobs_dev, out_w);
  KERNEL_CHECK_RET();
  hipLaunchKernelGGL(resample_v_kernel, dim3((out_w + 63) / 64, (out_h + 3) / 4, njobs), dim3(256), 0, st, jobs_dev, out_h, out_w);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int mmvqa_aug_rotate(mmvqa_stream_t s, const unsigned char* src, unsigned char* dst, const int* fix_dev, int B, int H,
                     int W) {
  if (!src || !dst || src == dst || !fix_dev || B <= 0 || H <= 0 || W <= 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "aug_rotate: bad arguments (src and dst must differ)");
  hipLaunchKernelGGL(rotate_nearest_kernel, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(256), 0, (hipStream_t)s, src, dst, fix_dev, H, W);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int mmvqa_aug_jitter_round(mmvqa_stream_t s, unsigned char* img, const int* op_dev, const float* factor_dev,
                           unsigned long long* lsum_dev, int B, int npix) {
  if (!img || !op_dev || !factor_dev || !lsum_dev || B <= 0 || npix <= 0)
    return mmvqa_set_error(MMVQA_ERR_ARG, "aug_jitter_round: bad arguments");
  hipStream_t st = (hipStream_t)s;
  HIP_CHECK_RET(hipMemsetAsync(lsum_dev, 0, sizeof(unsigned long long) * (size_t)B, st));
  const int gx = (npix + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(lsum_kernel, dim3(gx, B), dim3(256), 0, st, img, lsum_dev, npix);
  KERNEL_CHECK_RET();
  hipLaunchKernelGGL(jitter_kernel, dim3(gx, B), dim3(256), 0, st, img, op_dev, factor_dev, lsum_dev, npix);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

int mmvqa_aug_to_tensor(mmvqa_stream_t s, const unsigned char* img, float* out, int B, int npix, const float* mean3,
                        const float* std3) {
  if (!img || !out || !mean3 || !std3 || B <= 0 || npix <= 0) return mmvqa_set_error(MMVQA_ERR_ARG, "aug_to_tensor: bad arguments");
  const int gx = (npix + 256 * 4 - 1) / (256 * 4);
  hipLaunchKernelGGL(to_tensor_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)s, img, out, npix, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2]);
  KERNEL_CHECK_RET();
  return MMVQA_OK;
}

}  // extern "C"
