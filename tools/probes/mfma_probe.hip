// MFMA issue-rate probe for gfx950: what fraction of the fp32 MFMA peak does a loop of v_mfma_f32_32x32x2_f32 reach with
// 1 or 2 waves per SIMD, alone and with the side traffic k_wn_layer's K loop carries (LDS reads, global loads, barriers)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE bit 0: two ds_read_b128 per 32 MFMAs; bit 1: four global_load_dwordx4 per 32 MFMAs (L2-resident 1 MiB image);
// bit 2: one __syncthreads per 256 MFMAs; bit 3: A operands come from the loaded registers (data dependence on the loads);
// bit 4: two loads instead of four; bit 5: the loads are spread between the MFMA clusters (one per 8 MFMAs) instead of
// issued together; bit 6: the loads are LDS-DMA (global_load_lds_dwordx4) + ds_read_b128 back instead of loads to VGPRs
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_probe(const float4* __restrict__ w, float* __restrict__ out, int iters) {
  __shared__ float4 lds[4096];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float4(1e-3f * i, 0.5f, 0.25f, 0.125f);
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float4 a[4] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(.1f, .2f, .3f, .4f), make_float4(.5f, .6f, .7f, .8f), make_float4(.9f, 1.f, 1.1f, 1.2f)};
  float4 an[4];
  float4 b0 = lds[lane], b1 = lds[64 + lane];
  const float4* wp = w + (size_t)wv * 256 + lane;
  constexpr int NLD = (MODE & 16) ? 2 : 4;
  __shared__ float4 dma[8 * 4 * 64];
  for (int it = 0; it < iters; ++it) {
    if ((MODE & 2) && !(MODE & 32) && !(MODE & 64)) {
#pragma unroll
      for (int r = 0; r < NLD; ++r) an[r] = wp[(size_t)((it * 4 + r) & 1023) * 64];
    }
    if (MODE & 64) {
#pragma unroll
      for (int r = 0; r < NLD; ++r)
        __builtin_amdgcn_global_load_lds(wp + (size_t)((it * 4 + r) & 1023) * 64 - lane + lane, (__attribute__((address_space(3))) void*)(dma + (wv * 4 + r) * 64), 16, 0, 0);
#pragma unroll
      for (int r = 0; r < NLD; ++r) an[r] = dma[(wv * 4 + r) * 64 + lane];
    }
    float4 c0 = b0, c1 = b1;
    if (MODE & 1) { c0 = lds[(it & 31) * 128 + lane]; c1 = lds[(it & 31) * 128 + 64 + lane]; }
    __builtin_amdgcn_sched_barrier(0);
    const float bs[4][2] = {{c0.x, c1.x}, {c0.y, c1.y}, {c0.z, c1.z}, {c0.w, c1.w}};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const float av = s == 0 ? a[rb].x : s == 1 ? a[rb].y : s == 2 ? a[rb].z : a[rb].w;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb * 2 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bs[s][cb], acc[rb * 2 + cb], 0, 0, 0);
        if ((MODE & 32) && rb == 3 && s < NLD) {
          an[s] = wp[(size_t)((it * 4 + s) & 1023) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    if (MODE & 2) {
      if (MODE & 8) { for (int r = 0; r < NLD; ++r) a[r] = an[r]; }
      else asm volatile("" ::"v"(an[0].x), "v"(an[1].x), "v"(an[NLD - 1].x));
    }
    if ((MODE & 4) && (it & 7) == 7) __syncthreads();
  }
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) sum += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int MODE, int WAVES>
int run(const char* what, int blocks_per_cu, const float4* w, float* out) {
  const int iters = 4000, ncu = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = ncu * blocks_per_cu;
  k_probe<MODE, WAVES><<<grid, WAVES * 64>>>(w, out, 100);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_probe<MODE, WAVES><<<grid, WAVES * 64>>>(w, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * WAVES * iters * 32 * 4096.0;
  printf("%-64s waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s = %.3f of 157.3\n", what, WAVES * blocks_per_cu / 4, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  return 0;
}

int main() {
  float4* w; float* out;
  CK(hipMalloc(&w, (size_t)1024 * 64 * 16 + 4 * 256 * 16));
  CK(hipMemset(w, 0, (size_t)1024 * 64 * 16 + 4 * 256 * 16));
  CK(hipMalloc(&out, (size_t)256 * 2 * 512 * 4));
  run<0, 4>("MFMA only, 4-wave workgroups, 1 per CU", 1, w, out);
  run<0, 4>("MFMA only, 4-wave workgroups, 2 per CU", 2, w, out);
  run<0, 8>("MFMA only, 8-wave workgroups, 1 per CU", 1, w, out);
  run<1, 4>("+ 2 ds_read_b128 / 32 MFMA, 2 per CU", 2, w, out);
  run<2, 4>("+ 4 global_load_dwordx4 / 32 MFMA (unused), 2 per CU", 2, w, out);
  run<10, 4>("+ 4 global_load_dwordx4 / 32 MFMA feeding A, 2 per CU", 2, w, out);
  run<11, 4>("+ LDS reads + global loads feeding A, 2 per CU", 2, w, out);
  run<15, 4>("+ LDS + global + barrier / 256 MFMA, 2 per CU", 2, w, out);
  run<15, 4>("+ LDS + global + barrier / 256 MFMA, 1 per CU", 1, w, out);
  run<15, 8>("+ LDS + global + barrier, 8-wave workgroups, 1 per CU", 1, w, out);
  run<10 + 16, 4>("2 global loads / 32 MFMA feeding A, 2 per CU", 2, w, out);
  run<10 + 16, 8>("2 global loads / 32 MFMA feeding A, 8-wave, 1 per CU", 1, w, out);
  run<10 + 32, 4>("4 global loads spread (1 per 8 MFMA) feeding A, 2 per CU", 2, w, out);
  run<10 + 16 + 32, 4>("2 global loads spread feeding A, 2 per CU", 2, w, out);
  run<10 + 64, 4>("4 LDS-DMA loads + 4 ds_read_b128 feeding A, 2 per CU", 2, w, out);
  run<10 + 64 + 16, 4>("2 LDS-DMA loads + 2 ds_read_b128 feeding A, 2 per CU", 2, w, out);
  return 0;
}
