// Calibration: sustained clock and issue rate of v_mfma_f32_32x32x2_f32 under load.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters, int nacc) {
  f32x16 a0, a1, a2, a3;
  for (int e = 0; e < 16; ++e) { a0[e] = 0; a1[e] = 0; a2[e] = 0; a3[e] = 0; }
  float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (nacc == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* st;
  const int maxb = 1024;
  hipMalloc(&out, maxb * 256 * 4); hipMalloc(&st, maxb * 16);
  for (int nacc : {1, 4}) for (int blocks : {64, 196, 256, 512}) {
    int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {   // sustained load
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, iters, nacc); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) {
        std::vector<unsigned long long> h(blocks * 2); hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<double> ghz, cyc;
        for (int b = 0; b < blocks; ++b) { ghz.push_back((double)h[2*b] / (double)h[2*b+1] * 0.1); cyc.push_back((double)h[2*b] / (iters * 16.0)); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        double flops = (double)blocks * 4 * iters * 16.0 * 4096.0;
        printf("nacc=%d blocks=%3d: %.3f ms  %.1f TFLOP/s  clock(median) %.2f GHz  cycles/MFMA(median) %.1f\n", nacc, blocks, ms,
               flops / ms / 1e9, ghz[blocks / 2], cyc[blocks / 2]);
      }
    }
  }
  return 0;
}
