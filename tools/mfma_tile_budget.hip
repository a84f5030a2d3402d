// Cycle budget of one K-tile of the 64x64x64 fp32 tile as the igemm kernel runs it (4 waves, one per SIMD,
// every wave does everything): 32 dependent MFMAs + 16 ds_read_b128, plus per wave and K-tile
//   NL buffer_load_dwordx4 (L2-resident, burst in one k-group or one per 3 MFMAs),  NV VALU,  NW ds_write_b128.
// Components are switched on one at a time to see what each really costs beside the fp32 MFMA.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_tile_budget.hip -o /tmp/budget && /tmp/budget
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LOADS: 0 none, 1 burst of NL after the first MFMA group, 2 one load every third MFMA; WAITLD: wait for the loads of the
// previous period in the middle of the period (as the LDS-write phase does); NV VALU per period; NW ds_write per period
template <int LOADS, int NL, int WAITLD, int NV, int NW, int BAR>
__global__ __launch_bounds__(256) void k(const float* src, float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 8704];
  const int tid = threadIdx.x;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0;
  float x = tid * 1e-3f + 0.5f, y = 1.0f - tid * 1e-4f;
  float v0 = x, v1 = y, v2 = x + y, v3 = x - y;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 22, 0x00020000);
  const int voff = tid * 16 + (blockIdx.x & 63) * 4096;
  f32x4 st[12];
  for (int i = 0; i < 12; ++i) st[i] = f32x4{x, y, x, y};
  const unsigned wp = (unsigned)(size_t)(lds + tid * 4);
  const unsigned rp = (unsigned)(size_t)(lds + (tid & 63) * 4);
  f32x4 q0, q1;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int soff = 0;
  for (int it = 0; it < iters; ++it) {
    int li = 0, vi = 0, wi = 0;
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) {
      if ((sl & 3) == 0) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(rp));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q1) : "v"(rp));
      }
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
      // loads
      const bool ld_here = LOADS == 1 ? (sl >= 1 && sl <= 3) : (LOADS == 2 ? (sl % 3 == 1) : false);
      if (ld_here) {
        const int cnt = LOADS == 1 ? (NL + 2) / 3 : 1;
#pragma unroll
        for (int c = 0; c < cnt; ++c)
          if (li < NL) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[li]) : "v"(voff + li * 262144), "s"(r), "s"(soff) : "memory");
            ++li;
          }
      }
      if (WAITLD && sl == 12) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL > 4 ? 4 : 0) : "memory");
      // VALU + LDS writes spread over slots 12..27
      if (sl >= 12 && sl < 28) {
#pragma unroll
        for (int c = 0; c < (NV + 15) / 16; ++c)
          if (vi < NV) {
            if ((vi & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(x), "v"(y));
            if ((vi & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(x), "v"(y));
            if ((vi & 3) == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v2) : "v"(x));
            if ((vi & 3) == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v3) : "v"(y));
            ++vi;
          }
        if ((sl & 1) == 0 && wi < NW) {
          asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(wp), "v"(st[wi]), "n"(0) : "memory");
          ++wi;
        }
      }
      if (BAR && sl == 27) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    soff = (soff + 4096) & 0x3FFFF;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = q0[0] + q1[1] + v0 + v1 + v2 + v3;
  for (int i = 0; i < 12; ++i) s += st[i][0];
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int LOADS, int NL, int WAITLD, int NV, int NW, int BAR>
void run(const char* what, const float* src, float* out, unsigned long long* st) {
  const int iters = 1000, blocks = 196;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<LOADS, NL, WAITLD, NV, NW, BAR>), dim3(blocks), dim3(256), 0, 0, src, out, st, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-58s %5.0f cycles per K-tile (+%4.0f over the 2048 of the MFMAs)\n", what, (double)h[blocks / 2] / iters, (double)h[blocks / 2] / iters - 2048);
}

int main() {
  float *src, *out; unsigned long long* st;
  hipMalloc(&src, 1 << 23); hipMemset(src, 0, 1 << 23); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&st, 512 * 8);
  run<0, 0, 0, 0, 0, 0>("MFMA + 16 ds_read", src, out, st);
  run<0, 0, 0, 0, 0, 1>("+ barrier", src, out, st);
  run<0, 0, 0, 48, 0, 1>("+ barrier + 48 VALU", src, out, st);
  run<0, 0, 0, 0, 8, 1>("+ barrier + 8 ds_write", src, out, st);
  run<0, 0, 0, 48, 8, 1>("+ barrier + 48 VALU + 8 ds_write", src, out, st);
  run<1, 10, 0, 0, 0, 1>("+ barrier + 10 loads (burst), never waited", src, out, st);
  run<2, 10, 0, 0, 0, 1>("+ barrier + 10 loads (spread), never waited", src, out, st);
  run<1, 10, 1, 0, 0, 1>("+ barrier + 10 loads (burst) + mid-period wait", src, out, st);
  run<2, 10, 1, 0, 0, 1>("+ barrier + 10 loads (spread) + mid-period wait", src, out, st);
  run<2, 10, 1, 48, 8, 1>("everything, loads spread", src, out, st);
  run<1, 10, 1, 48, 8, 1>("everything, loads burst", src, out, st);
  run<2, 5, 1, 48, 8, 1>("everything, 5 loads spread", src, out, st);
  run<2, 10, 1, 24, 8, 1>("everything, 24 VALU", src, out, st);
  return 0;
}
