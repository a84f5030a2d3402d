// Does side work issued between the MFMAs of one wave's dependent chain hide behind them?
// Per iteration: 8 x { v_mfma_f32_32x32x2_f32 (same accumulator) ; N independent instructions of one kind }.
// Kinds: VALU (v_fma_f32), LDS read (ds_read_b128), LDS write (ds_write_b128).  One wave per SIMD.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int N, int NT = 256>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[NT * 4 * 4];

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk0 = {1.5f, 0.5f}, pk1 = pk0;
  int sa = iters, sb = 3;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0;
  float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-4f;
  float v0 = x, v1 = y, v2 = x + y, v3 = x - y;
  f32x4 q0 = {x, y, x, y}, q1 = q0, q2 = q0, q3 = q0;
  float* lp = lds + threadIdx.x * 4;
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(lp + i * 1024) = q0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if constexpr (KIND == 0) {
          if ((n & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(x), "v"(y));
          if ((n & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(x), "v"(y));
          if ((n & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(x), "v"(y));
          if ((n & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(x), "v"(y));
        } else if constexpr (KIND == 1) {
          if ((n & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"((unsigned)(size_t)lp));
          if ((n & 3) == 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q1) : "v"((unsigned)(size_t)lp));
          if ((n & 3) == 2) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(q2) : "v"((unsigned)(size_t)lp));
          if ((n & 3) == 3) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(q3) : "v"((unsigned)(size_t)lp));
        } else if constexpr (KIND == 2) {
          asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)lp), "v"(q0) : "memory");
        } else if constexpr (KIND == 3) {
          asm volatile("s_add_i32 %0, %0, %1" : "+s"(sa) : "s"(sb) : "scc");
        } else if constexpr (KIND == 4) {
          if ((n & 1) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk0) : "v"(pk1));
          if ((n & 1) == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk1) : "v"(pk0));
        } else if constexpr (KIND == 5) {
          if ((n & 1) == 0) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(x), "v"(y));
          if ((n & 1) == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v1) : "v"(x));
        } else if constexpr (KIND == 6) {
          if ((n & 1) == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q0) : "v"(out + (threadIdx.x & 63) * 4) : "memory");
          if ((n & 1) == 1) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q1) : "v"(out + 256 + (threadIdx.x & 63) * 4) : "memory");
        }
      }
    }
    if constexpr (KIND == 1 || KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (KIND == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v1 + v2 + v3 + q0[0] + q1[1] + q2[2] + q3[3] + pk0[0] + pk1[1] + (float)sa;
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int KIND, int N, int NT = 256>
void run(const char* name, float* out, unsigned long long* st) {
  const int blocks = 256, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, N, NT>), dim3(blocks), dim3(NT), 0, 0, out, st, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, N, NT>), dim3(blocks), dim3(NT), 0, 0, out, st, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-9s N=%2d waves/SIMD %d : %.1f cycles per MFMA slot per wave\n", name, N, NT / 256, (double)h[blocks / 2] / (iters * 8.0));
  printf("          kernel %.3f ms -> %.1f TFLOP/s of MFMA work\n", ms, (double)blocks * (NT / 64) * iters * 8.0 * 4096.0 / ms / 1e9);
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&st, 256 * 8);
  run<0, 0>("valu", out, st); run<0, 4>("valu", out, st); run<0, 8>("valu", out, st); run<0, 12>("valu", out, st);
  run<0, 16>("valu", out, st); run<0, 24>("valu", out, st);
  run<1, 1>("ds_read", out, st); run<1, 2>("ds_read", out, st); run<1, 4>("ds_read", out, st);
  run<2, 1>("ds_write", out, st); run<2, 2>("ds_write", out, st);
  run<3, 4>("salu", out, st); run<3, 16>("salu", out, st);
  run<4, 4>("pk_fma", out, st); run<4, 8>("pk_fma", out, st);
  run<5, 8>("med3/cnd", out, st);
  run<6, 1>("gload", out, st); run<6, 2>("gload", out, st);
  run<0, 0, 512>("valu", out, st); run<0, 8, 512>("valu", out, st); run<0, 16, 512>("valu", out, st);
  run<1, 2, 512>("ds_read", out, st);
  return 0;
}
