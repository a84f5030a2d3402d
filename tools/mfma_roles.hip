// Should the loads / prologue math / LDS writes of a K-tile live in the MFMA waves or in partner waves?
// One "period" = one K-tile of the 64x64x64 fp32 tile: 32 dependent MFMAs + 16 ds_read_b128 per computing wave,
// and per loading wave 10 buffer_load_dwordx4 (L2-resident), 48 VALU, 8 ds_write_b128; one barrier per period.
//   mode 0: 4 waves, each does both        mode 1: 8 waves, waves 0-3 compute, waves 4-7 load
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_roles.hip -o /tmp/roles && /tmp/roles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(MODE ? 512 : 256) void k(const float* src, float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 8704];
  const int tid = threadIdx.x, ltid = tid & 255;
  const bool compute = MODE == 0 || tid < 256, load = MODE == 0 || tid >= 256;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0;
  float x = tid * 1e-3f + 0.5f, y = 1.0f - tid * 1e-4f;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 22, 0x00020000);
  const int voff = ltid * 16 + (blockIdx.x & 63) * 4096;
  f32x4 st[10], st2[10];
  for (int i = 0; i < 10; ++i) st[i] = st2[i] = f32x4{x, y, x, y};
  float* wp = lds + ltid * 4;
  const unsigned rp = (unsigned)(size_t)(lds + (ltid & 63) * 4);
  f32x4 q0, q1;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int soff = 0;
  for (int it = 0; it < iters; ++it) {
    if (compute) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(rp));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q1) : "v"(rp));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
        if (MODE == 0 && g == 0) {
#pragma unroll
          for (int i = 0; i < 10; ++i)
            st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + i * 262144, soff, 0));
        }
        if (MODE == 0 && g >= 1 && g <= 4) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            f32x4 v = st[(g - 1) * 2 + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], x, y); v[e] = fmaxf(v[e], 0.f); }
            *reinterpret_cast<f32x4*>(wp + ((g - 1) * 2 + c) * 1024) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE == 1 && load) {
      // two register stages: loads of period it+1 go to one while the other (loaded a period ago) is written to LDS
      auto half = [&](f32x4 (&ld)[10], f32x4 (&wr)[10]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; ++i)
          ld[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + i * 262144, soff, 0));
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          f32x4 v = wr[c];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = __builtin_fmaf(v[e], x, y); v[e] = fmaxf(v[e], 0.f); }
          *reinterpret_cast<f32x4*>(wp + c * 1024) = v;
        }
      };
      if (it & 1) half(st, st2); else half(st2, st);
    }
    soff = (soff + 4096) & 0x3FFFF;
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = q0[0] + q1[1];
  for (int i = 0; i < 10; ++i) s += st[i][0] + st2[i][1];
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const float* src, float* out, unsigned long long* st, int blocks) {
  const int iters = 1000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(MODE ? 512 : 256), 0, 0, src, out, st, iters);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(MODE ? 512 : 256), 0, 0, src, out, st, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("mode %d blocks %3d : %.0f cycles per K-tile period (2048 = MFMA bound)\n", MODE, blocks, (double)h[blocks / 2] / iters);
}

int main() {
  float *src, *out; unsigned long long* st;
  hipMalloc(&src, 1 << 23); hipMemset(src, 0, 1 << 23); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&st, 512 * 8);
  for (int blocks : {196, 256, 512}) { run<0>(src, out, st, blocks); run<1>(src, out, st, blocks); }
  return 0;
}
