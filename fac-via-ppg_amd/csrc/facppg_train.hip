// Training-direction kernels of the WaveGlow flow edges (src/waveglow/glow.py:82-102, 208-250): the c x c
// channel-mixing conv of Invertible1x1Conv (c <= 8) forward / data gradient (the same kernel with the
// transposed matrix) and its weight gradient.  These are HBM-streaming kernels: one position's c channels are
// c strided 4-byte-per-lane rows, a lane owns 4 consecutive positions (16-byte accesses), the c x c matrix
// sits in SGPRs/registers.
#include <algorithm>
#include <vector>

#include "facppg_common.h"

namespace facppg {
namespace {

// out[b][i][l] = sum_j W[i][j] * z[b][j][l]   (TRANS: W[j][i])
template <int C, bool TRANS>
__global__ __launch_bounds__(256) void k_conv1x1(const float* __restrict__ W, const float* __restrict__ z, float* __restrict__ out,
                                                 int L) {
  const int b = blockIdx.y;
  const size_t base = (size_t)b * C * L;
  float w[C][C];
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) w[i][j] = TRANS ? W[j * C + i] : W[i * C + j];
  const int l0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (l0 >= L) return;
  if (l0 + 4 <= L && (L & 3) == 0) {
    float4 x[C];
#pragma unroll
    for (int j = 0; j < C; ++j) x[j] = *(const float4*)(z + base + (size_t)j * L + l0);
#pragma unroll
    for (int i = 0; i < C; ++i) {
      float4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < C; ++j) {
        a.x = fmaf(w[i][j], x[j].x, a.x); a.y = fmaf(w[i][j], x[j].y, a.y);
        a.z = fmaf(w[i][j], x[j].z, a.z); a.w = fmaf(w[i][j], x[j].w, a.w);
      }
      *(float4*)(out + base + (size_t)i * L + l0) = a;
    }
  } else {
    for (int l = l0; l < min(l0 + 4, L); ++l) {
      float x[C];
#pragma unroll
      for (int j = 0; j < C; ++j) x[j] = z[base + (size_t)j * L + l];
#pragma unroll
      for (int i = 0; i < C; ++i) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) a = fmaf(w[i][j], x[j], a);
        out[base + (size_t)i * L + l] = a;
      }
    }
  }
}

// dW[i][j] = sum_{b,l} dout[b][i][l] * z[b][j][l]: per-workgroup partial sums in a fixed order (wave shuffle tree,
// then the waves' partials in LDS order), then a second pass sums the workgroups in index order -- deterministic.
template <int C>
__global__ __launch_bounds__(256) void k_conv1x1_wgrad_part(const float* __restrict__ dout, const float* __restrict__ z,
                                                            float* __restrict__ part, int B, int L) {
  float acc[C][C];
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) acc[i][j] = 0.f;
  const size_t n = (size_t)B * L;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const size_t b = p / L, l = p - b * L;
    float d[C], x[C];
#pragma unroll
    for (int i = 0; i < C; ++i) { d[i] = dout[(b * C + i) * L + l]; x[i] = z[(b * C + i) * L + l]; }
#pragma unroll
    for (int i = 0; i < C; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j) acc[i][j] = fmaf(d[i], x[j], acc[i][j]);
  }
  __shared__ float red[4][C * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      float v = acc[i][j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][i * C + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < C * C)
    part[(size_t)blockIdx.x * C * C + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[i] = sum of the partials of output i (n <= 64 outputs, the c x c mixing matrix): 16 groups of threads walk the partials on four
// interleaved chains each and meet in LDS in group order -- a fixed order (bit-reproducible).  (One thread per output walking all
// 512 partials as one dependent chain cost 14 us per launch at batch 12, 12 launches per step.)
__global__ __launch_bounds__(1024) void k_sum_parts(const float* __restrict__ part, float* __restrict__ out, int n_parts, int n) {
  __shared__ float red[16][64];
  const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int p0 = n_parts * q / 16, p1 = n_parts * (q + 1) / 16;
  float v4[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n)
    for (int p = p0; p < p1; p += 4)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u < p1) v4[u] += part[(size_t)(p + u) * n + i];
  red[q][i] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
  __syncthreads();
  if (q || i >= n) return;
  float v = red[0][i];
#pragma unroll
  for (int k = 1; k < 16; ++k) v += red[k][i];
  out[i] = v;
}

// ---- weight normalisation of EVERY weight-normed conv of the model in one launch (torch.nn.utils.weight_norm, dim 0:
// w[r][:] = g[r] * v[r][:] / ||v[r][:]||, glow.py:118-146).  One wavefront per output row; the table gives, per tensor,
// its pointers, row length and first global row.
struct WnTableEntry { const float* v; const float* g; float* w; float* norm; long row0; int rows, len; };
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int wn_find(const WnTableEntry* __restrict__ tab, int n, long row) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].row0 <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// Rows of <= 1024 floats whose length and start are multiples of 4 (every conv of the WN stacks: 768 / 640 / 256 / 1024) are read ONCE,
// as float4 pieces kept in registers between the norm and the scaling (round 5: one 4-byte request at a time per lane, rows read
// twice, left these two launches at 2.3 - 3 TB/s: 102 / 107 us per flow group).
constexpr int WN_MAXV = 4;     // float4 pieces per lane: rows up to 1024 floats
__device__ __forceinline__ bool wn_vec_row(const WnTableEntry& e, const void* a, const void* b) {
  return e.len % 4 == 0 && e.len <= 256 * WN_MAXV && (((size_t)a | (size_t)b) & 15) == 0;
}
__global__ __launch_bounds__(256) void k_weight_norm_fwd(const WnTableEntry* __restrict__ tab, int n, long total_rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int lane = threadIdx.x & 63;
  const WnTableEntry e = tab[wn_find(tab, n, row)];
  const int r = (int)(row - e.row0);
  const float* v = e.v + (size_t)r * e.len;
  float* w = e.w + (size_t)r * e.len;
  if (wn_vec_row(e, v, w)) {
    float4 x[WN_MAXV];
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < WN_MAXV; ++k) {
      const int i = 4 * lane + 256 * k;
      x[k] = i < e.len ? *reinterpret_cast<const float4*>(v + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < WN_MAXV; ++k) ss = fmaf(x[k].x, x[k].x, fmaf(x[k].y, x[k].y, fmaf(x[k].z, x[k].z, fmaf(x[k].w, x[k].w, ss))));
    const float norm = sqrtf(wave_sum(ss));
    const float sc = e.g[r] / norm;
#pragma unroll
    for (int k = 0; k < WN_MAXV; ++k) {
      const int i = 4 * lane + 256 * k;
      if (i < e.len) *reinterpret_cast<float4*>(w + i) = make_float4(x[k].x * sc, x[k].y * sc, x[k].z * sc, x[k].w * sc);
    }
    if (lane == 0) e.norm[r] = norm;
    return;
  }
  float ss = 0.0f;
  for (int i = lane; i < e.len; i += 64) { const float x = v[i]; ss = fmaf(x, x, ss); }
  const float norm = sqrtf(wave_sum(ss));
  const float sc = e.g[r] / norm;
  for (int i = lane; i < e.len; i += 64) w[i] = v[i] * sc;
  if (lane == 0) e.norm[r] = norm;
}
// backward: dg[r] = <dw, v> / ||v||;  dv = (g / ||v||) * (dw - v * <dw, v> / ||v||^2).  Table: v, g, w := dw (in), norm (in);
// outputs dv, dg in a second table of the same order (fields v := dv, g := dg).
__global__ __launch_bounds__(256) void k_weight_norm_bwd(const WnTableEntry* __restrict__ tab, const WnTableEntry* __restrict__ out, int n,
                                                         long total_rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int lane = threadIdx.x & 63;
  const int t = wn_find(tab, n, row);
  const WnTableEntry e = tab[t];
  const int r = (int)(row - e.row0);
  const float* v = e.v + (size_t)r * e.len;
  const float* dw = e.w + (size_t)r * e.len;
  float* dv = const_cast<float*>(out[t].v) + (size_t)r * e.len;
  const float norm = e.norm[r], gr = e.g[r];
  if (wn_vec_row(e, v, dw) && (((size_t)dv) & 15) == 0) {
    float4 xv[WN_MAXV], xd[WN_MAXV];
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < WN_MAXV; ++k) {
      const int i = 4 * lane + 256 * k;
      const bool in = i < e.len;
      xv[k] = in ? *reinterpret_cast<const float4*>(v + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      xd[k] = in ? *reinterpret_cast<const float4*>(dw + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < WN_MAXV; ++k) dot = fmaf(xd[k].x, xv[k].x, fmaf(xd[k].y, xv[k].y, fmaf(xd[k].z, xv[k].z, fmaf(xd[k].w, xv[k].w, dot))));
    dot = wave_sum(dot);
    const float a = gr / norm, bq = gr * dot / (norm * norm * norm);
#pragma unroll
    for (int k = 0; k < WN_MAXV; ++k) {
      const int i = 4 * lane + 256 * k;
      if (i < e.len)
        *reinterpret_cast<float4*>(dv + i) = make_float4(a * xd[k].x - bq * xv[k].x, a * xd[k].y - bq * xv[k].y, a * xd[k].z - bq * xv[k].z, a * xd[k].w - bq * xv[k].w);
    }
    if (lane == 0) const_cast<float*>(out[t].g)[r] = dot / norm;
    return;
  }
  float dot = 0.0f;
  for (int i = lane; i < e.len; i += 64) dot = fmaf(dw[i], v[i], dot);
  dot = wave_sum(dot);
  const float a = gr / norm, bq = gr * dot / (norm * norm * norm);
  for (int i = lane; i < e.len; i += 64) dv[i] = a * dw[i] - bq * v[i];
  if (lane == 0) const_cast<float*>(out[t].g)[r] = dot / norm;
}

template <int C>
int conv1x1_c(const float* W, const float* z, float* out, int B, int L, bool trans, hipStream_t s) {
  const dim3 grid((L + 1023) / 1024, B);
  if (trans) k_conv1x1<C, true><<<grid, 256, 0, s>>>(W, z, out, L);
  else k_conv1x1<C, false><<<grid, 256, 0, s>>>(W, z, out, L);
  return FACPPG_OK;
}

constexpr int kWgradParts = 512;

// affine coupling of a flow in the training direction (glow.py:240-245): y = cat(x0, exp(log_s) * x1 + b) with
// x = [x0 | x1] (h channels each), wn = [b | log_s]; 4 consecutive positions per thread (<= 8 channels: a few MB per step)
__global__ __launch_bounds__(256) void k_affine_fwd(const float* __restrict__ x, const float* __restrict__ wn, float* __restrict__ y, int h, int L) {
  const int b = blockIdx.y;
  const size_t base = (size_t)b * 2 * h * L;
  const int l0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  for (int l = l0; l < min(l0 + 4, L); ++l)
    for (int j = 0; j < h; ++j) {
      y[base + (size_t)j * L + l] = x[base + (size_t)j * L + l];
      y[base + (size_t)(h + j) * L + l] = __expf(wn[base + (size_t)(h + j) * L + l]) * x[base + (size_t)(h + j) * L + l] + wn[base + (size_t)j * L + l];
    }
}
// dx = [dy0 | dy1 * exp(log_s)], dwn = [dy1 | dy1 * exp(log_s) * x1]
__global__ __launch_bounds__(256) void k_affine_bwd(const float* __restrict__ x, const float* __restrict__ wn, const float* __restrict__ dy,
                                                    float* __restrict__ dx, float* __restrict__ dwn, int h, int L) {
  const int b = blockIdx.y;
  const size_t base = (size_t)b * 2 * h * L;
  const int l0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  for (int l = l0; l < min(l0 + 4, L); ++l)
    for (int j = 0; j < h; ++j) {
      const size_t o0 = base + (size_t)j * L + l, o1 = base + (size_t)(h + j) * L + l;
      const float e = __expf(wn[o1]), d1 = dy[o1];
      dx[o0] = dy[o0];
      dx[o1] = d1 * e;
      dwn[o0] = d1;
      dwn[o1] = d1 * e * x[o1];
    }
}


// Sums of a handful of strided fp32 segments in ONE launch each stage (WaveGlowLoss, glow.py:43-59: sum(z*z), and the sum of
// every flow's log_s, which lives in the upper half of that flow's WN output): segment e = `outer` runs of `inner` contiguous
// floats, `outer_stride` apart; `square` sums v*v.  Stage 1: workgroup (slice, segment) adds its slice in double, threads in a
// fixed stride pattern, lanes meet in LDS in a fixed tree; stage 2 adds the slices in index order -- bit-reproducible.
constexpr int SUM_SLICES = 64;
struct SumSeg { const float* p; long outer_stride; int outer, inner, square, pad; };
struct SumArgs { SumSeg seg[FACPPG_MAX_SUM_SEGMENTS]; double* part; float* out; };
__global__ __launch_bounds__(256) void k_seg_sum_part(SumArgs a) {
  const SumSeg& sg = a.seg[blockIdx.y];
  const long total = (long)sg.outer * sg.inner;
  const long i0 = total * blockIdx.x / SUM_SLICES, i1 = total * (blockIdx.x + 1) / SUM_SLICES;
  double v = 0.0;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const long o = i / sg.inner;
    const float x = sg.p[o * sg.outer_stride + (i - o * sg.inner)];
    v += sg.square ? (double)x * x : (double)x;
  }
  __shared__ double red[256];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.part[blockIdx.y * SUM_SLICES + blockIdx.x] = red[0];
}
__global__ void k_seg_sum_final(const double* __restrict__ part, float* __restrict__ out, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double v = 0.0;
  for (int s = 0; s < SUM_SLICES; ++s) v += part[e * SUM_SLICES + s];
  out[e] = (float)v;
}

}  // namespace
}  // namespace facppg

using namespace facppg;

// log det W and W^-T of one small mixing matrix (c <= 8): facppg::logdet_wave (facppg_common.h), one wave per matrix
__global__ __launch_bounds__(64) void k_logdet(const float* __restrict__ W, int c, float* __restrict__ logdet, float* __restrict__ winv_t) {
  logdet_wave(W, c, logdet, winv_t);
}

extern "C" int facppg_logdet(const float* w_dev, int c, float* logdet_dev, float* winv_t_dev, void* stream) {
  FACPPG_REQUIRE(w_dev && logdet_dev && winv_t_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(c >= 1 && c <= 8, FACPPG_EUNSUPPORTED, "mixing matrices are at most 8 x 8 (got %d)", c);
  k_logdet<<<1, 64, 0, (hipStream_t)stream>>>(w_dev, c, logdet_dev, winv_t_dev);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_conv1x1(const float* w_dev, const float* z_dev, float* out_dev, int B, int c, int L, int transpose_w,
                              void* stream) {
  FACPPG_REQUIRE(w_dev && z_dev && out_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && L > 0 && B <= 65535, FACPPG_EINVAL, "bad B/L");
  FACPPG_REQUIRE(z_dev != out_dev, FACPPG_EINVAL, "facppg_conv1x1 is not in-place");
  hipStream_t s = (hipStream_t)stream;
  switch (c) {
    case 2: conv1x1_c<2>(w_dev, z_dev, out_dev, B, L, transpose_w != 0, s); break;
    case 4: conv1x1_c<4>(w_dev, z_dev, out_dev, B, L, transpose_w != 0, s); break;
    case 6: conv1x1_c<6>(w_dev, z_dev, out_dev, B, L, transpose_w != 0, s); break;
    case 8: conv1x1_c<8>(w_dev, z_dev, out_dev, B, L, transpose_w != 0, s); break;
    default: FACPPG_REQUIRE(false, FACPPG_EUNSUPPORTED, "channel count %d (built: 2, 4, 6, 8)", c);
  }
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" size_t facppg_conv1x1_wgrad_workspace_bytes(int c) { return (size_t)kWgradParts * c * c * 4; }

extern "C" int facppg_conv1x1_wgrad(const float* dout_dev, const float* z_dev, float* dw_dev, int B, int c, int L,
                                    void* workspace_dev, size_t workspace_bytes, void* stream) {
  FACPPG_REQUIRE(dout_dev && z_dev && dw_dev && workspace_dev, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(B > 0 && L > 0, FACPPG_EINVAL, "bad B/L");
  FACPPG_REQUIRE(workspace_bytes >= facppg_conv1x1_wgrad_workspace_bytes(c), FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu",
                 workspace_bytes, facppg_conv1x1_wgrad_workspace_bytes(c));
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace_dev;
  const int parts = (int)std::min<size_t>(kWgradParts, ((size_t)B * L + 255) / 256);
  switch (c) {
    case 2: k_conv1x1_wgrad_part<2><<<parts, 256, 0, s>>>(dout_dev, z_dev, part, B, L); break;
    case 4: k_conv1x1_wgrad_part<4><<<parts, 256, 0, s>>>(dout_dev, z_dev, part, B, L); break;
    case 6: k_conv1x1_wgrad_part<6><<<parts, 256, 0, s>>>(dout_dev, z_dev, part, B, L); break;
    case 8: k_conv1x1_wgrad_part<8><<<parts, 256, 0, s>>>(dout_dev, z_dev, part, B, L); break;
    default: FACPPG_REQUIRE(false, FACPPG_EUNSUPPORTED, "channel count %d (built: 2, 4, 6, 8)", c);
  }
  k_sum_parts<<<1, 1024, 0, s>>>(part, dw_dev, parts, c * c);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_affine_forward(const float* x_dev, const float* wn_out_dev, float* y_dev, int B, int h, int L, void* stream) {
  FACPPG_REQUIRE(x_dev && wn_out_dev && y_dev && B > 0 && h > 0 && L > 0 && B <= 65535, FACPPG_EINVAL, "bad argument");
  k_affine_fwd<<<dim3((L + 1023) / 1024, B), 256, 0, (hipStream_t)stream>>>(x_dev, wn_out_dev, y_dev, h, L);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_affine_backward(const float* x_dev, const float* wn_out_dev, const float* dy_dev, float* dx_dev, float* dwn_out_dev, int B,
                                      int h, int L, void* stream) {
  FACPPG_REQUIRE(x_dev && wn_out_dev && dy_dev && dx_dev && dwn_out_dev && B > 0 && h > 0 && L > 0 && B <= 65535, FACPPG_EINVAL, "bad argument");
  k_affine_bwd<<<dim3((L + 1023) / 1024, B), 256, 0, (hipStream_t)stream>>>(x_dev, wn_out_dev, dy_dev, dx_dev, dwn_out_dev, h, L);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

extern "C" int facppg_segment_sums(const facppg_sum_segment* segs, int n, void* workspace_dev, size_t workspace_bytes, float* out_dev,
                                   void* stream) {
  FACPPG_REQUIRE(segs && out_dev && workspace_dev && n > 0 && n <= FACPPG_MAX_SUM_SEGMENTS, FACPPG_EINVAL, "bad argument (1..%d segments)",
                 FACPPG_MAX_SUM_SEGMENTS);
  FACPPG_REQUIRE(workspace_bytes >= (size_t)n * SUM_SLICES * sizeof(double), FACPPG_EINVAL, "workspace of %zu bytes, need %zu", workspace_bytes,
                 (size_t)n * SUM_SLICES * sizeof(double));
  SumArgs a;
  for (int e = 0; e < n; ++e) {
    FACPPG_REQUIRE(segs[e].data_dev && segs[e].outer > 0 && segs[e].inner > 0, FACPPG_EINVAL, "segment %d is empty", e);
    a.seg[e] = SumSeg{segs[e].data_dev, segs[e].outer_stride, segs[e].outer, segs[e].inner, segs[e].square, 0};
  }
  a.part = (double*)workspace_dev; a.out = out_dev;
  k_seg_sum_part<<<dim3(SUM_SLICES, n), 256, 0, (hipStream_t)stream>>>(a);
  k_seg_sum_final<<<1, 64, 0, (hipStream_t)stream>>>(a.part, out_dev, n);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// table entries are 6 x 8 bytes: {v, g, w, norm, row0, (rows | len << 32)} as the host packs them (see facppg.h)
extern "C" int facppg_weight_norm_forward(const void* table_dev, int n_tensors, long total_rows, void* stream) {
  FACPPG_REQUIRE(table_dev && n_tensors > 0 && total_rows > 0, FACPPG_EINVAL, "bad argument");
  k_weight_norm_fwd<<<(unsigned)((total_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>((const WnTableEntry*)table_dev, n_tensors, total_rows);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
extern "C" int facppg_weight_norm_backward(const void* table_dev, const void* out_table_dev, int n_tensors, long total_rows, void* stream) {
  FACPPG_REQUIRE(table_dev && out_table_dev && n_tensors > 0 && total_rows > 0, FACPPG_EINVAL, "bad argument");
  k_weight_norm_bwd<<<(unsigned)((total_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>((const WnTableEntry*)table_dev, (const WnTableEntry*)out_table_dev,
                                                                                      n_tensors, total_rows);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Adam over ALL parameters of the model in one launch (script/train_waveglow.py:83,134: torch.optim.Adam(...).step()).
// torch's fused multi-tensor Adam walks the 938 parameters in 27 launches of a chunked kernel at ~1.9 TB/s (1.27 ms of
// a 13 ms step); this one is a single streaming pass over a pointer table at the HBM rate: 16 B read + 12 B written per
// element, 16-byte accesses, one 4096-element chunk per workgroup.  Same arithmetic as torch._fused_adam_
// (ATen/native/cuda/fused_adam_utils.cuh, ADAM mode ORIGINAL, amsgrad off): bias corrections formed in double from the
// step count, lerp for the first moment, sqrt(v) / sqrt(bc2) + eps.
// ------------------------------------------------------------------------------------------------------------
namespace facppg {
namespace {
struct AdamTensor { float* p; const float* g; float* m; float* v; long n; };
static_assert(sizeof(AdamTensor) == 40, "table layout is part of the ABI (facppg.h)");
constexpr int ADAM_CHUNK = 4096;

__global__ void k_adam_tick(float* step) { *step += 1.0f; }

__global__ __launch_bounds__(256) void k_adam(const AdamTensor* __restrict__ tens, const int2* __restrict__ chunks, const float* __restrict__ step_ptr,
                                              float lr, double beta1, double beta2, float eps, float weight_decay) {
  const int2 ck = chunks[blockIdx.x];
  const AdamTensor t = tens[ck.x];
  const double step = (double)*step_ptr;
  const float bc1 = (float)(1.0 - pow(beta1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, step));
  const float step_size = lr / bc1, b2 = (float)beta2, w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
  const long base = (long)ck.y * ADAM_CHUNK;
  const long n = t.n - base < ADAM_CHUNK ? t.n - base : ADAM_CHUNK;
  auto update = [&](float& p, float g, float& m, float& v) {
    if (weight_decay != 0.0f) g = fmaf(weight_decay, p, g);
    m = fmaf(w1, g - m, m);                       // lerp(m, g, 1 - beta1), weight < 0.5 branch
    v = b2 * v + w2 * g * g;
    p -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
  };
  const bool vec = (((size_t)t.p | (size_t)t.g | (size_t)t.m | (size_t)t.v) & 15) == 0;
  if (vec) {
    float4* p4 = reinterpret_cast<float4*>(t.p + base);
    const float4* g4 = reinterpret_cast<const float4*>(t.g + base);
    float4* m4 = reinterpret_cast<float4*>(t.m + base);
    float4* v4 = reinterpret_cast<float4*>(t.v + base);
    const long n4 = n / 4;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 p = p4[i], m = m4[i], v = v4[i];
      const float4 g = g4[i];
      update(p.x, g.x, m.x, v.x); update(p.y, g.y, m.y, v.y); update(p.z, g.z, m.z, v.z); update(p.w, g.w, m.w, v.w);
      p4[i] = p; m4[i] = m; v4[i] = v;
    }
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) update(t.p[base + i], t.g[base + i], t.m[base + i], t.v[base + i]);
  } else {
    for (long i = threadIdx.x; i < n; i += 256) update(t.p[base + i], t.g[base + i], t.m[base + i], t.v[base + i]);
  }
}
}  // namespace
}  // namespace facppg

extern "C" int facppg_adam_chunk_elems(void) { return facppg::ADAM_CHUNK; }

extern "C" int facppg_adam_step(const void* table_dev, int n_tensors, const int32_t* chunks_dev, int n_chunks, float* step_dev, float lr,
                                double beta1, double beta2, float eps, float weight_decay, void* stream) {
  using namespace facppg;
  FACPPG_REQUIRE(table_dev && chunks_dev && step_dev && n_tensors > 0 && n_chunks > 0, FACPPG_EINVAL, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  k_adam_tick<<<1, 1, 0, s>>>(step_dev);
  k_adam<<<(unsigned)n_chunks, 256, 0, s>>>((const AdamTensor*)table_dev, (const int2*)chunks_dev, step_dev, lr, beta1, beta2, eps, weight_decay);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}

namespace facppg {
namespace {

// ------------------------------------------------------------------------------------------
// Weight and bias gradients of one flow's WN stack in fp32 (the autograd backward of glow.py:154-175 with respect to the
// parameters; until round 4 these were torch.bmm(...).sum(0) calls, i.e. rocBLAS): every weight gradient is an NT product
// over the positions of two channel-major saved tensors,
//     out[m][k] = sum_b sum_{n < L} A[b][m][n] * X[b][k][n]      (X optionally the product of two tensors: the gate output)
// k_wgrad_f32 forms a 128 x 64 tile of one such product per workgroup on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32;
// wave w: rows 32w .. 32w+31, two 32-column blocks), all products of a flow in ONE launch (blockIdx.z = problem).  The
// reduction runs over 32 positions per stage: both operand tiles are read along their contiguous position axis (16 B per
// lane), written to LDS as [n/4][row][4] with the four positions of a group in the order (0, 2, 1, 3), so that a lane's two
// MFMA K-steps of a group (n = 4g + kh and 4g + 2 + kh) are ONE 8-byte LDS read; two LDS buffers, loads of the next stage in
// registers under the MFMAs of this one.  The sum over the batch runs inside the workgroup (fixed order: deterministic, no
// partial buffers).  k_rowsum_f32 forms the bias gradients (row sums over batch and positions) in a fixed order.
// ------------------------------------------------------------------------------------------
struct WgradF32Prob {
  const float* A;    // [B][>= M][lda]   (row m at A + b * a_bs + m * lda)
  const float* X;    // [B][>= K][ldx]
  const float* X2;   // optional: X is taken as X[k][n] * X2[k][n]
  float* out;        // element (m, k) at out[m * so_m + k * so_k]
  long a_bs, x_bs;
  int lda, ldx, M, K, so_m, so_k;
};
struct RowSumF32 {
  const float* src;  // [B][rows][ld]
  float* out;        // [rows]
  float* out2;       // optional second copy (the in and cond biases share their gradient)
  long bs;
  int ld, rows, first;   // first: index of this group's first row in the launch
};

typedef float f4ua __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment

constexpr int WG_TM = 128, WG_TK = 64, WG_TN = 32;

__global__ __launch_bounds__(256) void k_wgrad_f32(const WgradF32Prob* __restrict__ probs, int B, int L) {
  __shared__ __attribute__((aligned(16))) float As[2][WG_TN / 4][WG_TM][4];
  __shared__ __attribute__((aligned(16))) float Xs[2][WG_TN / 4][WG_TK][4];
  const WgradF32Prob pr = probs[blockIdx.z];
  const int m0 = blockIdx.y * WG_TM, k0 = blockIdx.x * WG_TK;
  if (m0 >= pr.M || k0 >= pr.K) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int g = tid & 7, r0 = tid >> 3;   // staging: 8 threads cover 32 positions of a row, 32 rows per pass
  f32x16 acc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
  float4 ra[4], rx[2];
  const int nchunk = (L + WG_TN - 1) / WG_TN, nst = B * nchunk;
  auto load = [&](int st) __attribute__((always_inline)) {
    const int b = st / nchunk, n = (st - b * nchunk) * WG_TN + 4 * g;
    // positions >= L add nothing (and may never have been written: 0 * NaN = NaN, so BOTH operands are zeroed there);
    // the last group of a row is read element by element -- a row of a0 / dout ends where the tensor ends
    auto row4 = [&](const float* p) __attribute__((always_inline)) {
      if (n + 4 <= L) {
        const f4ua q = *reinterpret_cast<const f4ua*>(p + n);
        return make_float4(q.x, q.y, q.z, q.w);
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < L) v.x = p[n];
      if (n + 1 < L) v.y = p[n + 1];
      if (n + 2 < L) v.z = p[n + 2];
      return v;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + r0 + 32 * j;
      ra[j] = m < pr.M ? row4(pr.A + (size_t)b * pr.a_bs + (size_t)m * pr.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + r0 + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < pr.K) {
        const size_t o = (size_t)b * pr.x_bs + (size_t)k * pr.ldx;
        v = row4(pr.X + o);
        if (pr.X2) {
          const float4 q2 = row4(pr.X2 + o);
          v.x *= q2.x; v.y *= q2.y; v.z *= q2.z; v.w *= q2.w;
        }
      }
      rx[j] = v;
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&As[buf][g][r0 + 32 * j][0]) = make_float4(ra[j].x, ra[j].z, ra[j].y, ra[j].w);
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&Xs[buf][g][r0 + 32 * j][0]) = make_float4(rx[j].x, rx[j].z, rx[j].y, rx[j].w);
  };
  load(0);
  stash(0);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) load(st + 1);
#pragma unroll
    for (int gg = 0; gg < WG_TN / 4; ++gg) {
      const float2 a = *reinterpret_cast<const float2*>(&As[buf][gg][32 * w + li][2 * kh]);
      const float2 x0 = *reinterpret_cast<const float2*>(&Xs[buf][gg][li][2 * kh]);
      const float2 x1 = *reinterpret_cast<const float2*>(&Xs[buf][gg][32 + li][2 * kh]);
      acc[0] = mfma32x32x2(a.x, x0.x, acc[0]);
      acc[1] = mfma32x32x2(a.x, x1.x, acc[1]);
      acc[0] = mfma32x32x2(a.y, x0.y, acc[0]);
      acc[1] = mfma32x32x2(a.y, x1.y, acc[1]);
    }
    if (st + 1 < nst) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int k = k0 + 32 * cb + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * w + 8 * (r >> 2) + (r & 3) + 4 * kh;
      if (m < pr.M && k < pr.K) pr.out[(size_t)m * pr.so_m + (size_t)k * pr.so_k] = acc[cb][r];
    }
  }
}

// one workgroup per row: sum over batch and positions, each thread a strided share in index order, then the wave's
// shuffle tree and the four waves in LDS order
__global__ __launch_bounds__(256) void k_rowsum_f32(const RowSumF32* __restrict__ groups, int n_groups, int B, int L) {
  __shared__ float part[4];
  int gi = 0;
  while (gi + 1 < n_groups && (int)blockIdx.x >= groups[gi + 1].first) ++gi;
  const RowSumF32 gr = groups[gi];
  const int row = blockIdx.x - gr.first;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* p = gr.src + (size_t)b * gr.bs + (size_t)row * gr.ld;
    for (int n = threadIdx.x; n < L; n += 256) s += p[n];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = ((part[0] + part[1]) + part[2]) + part[3];
    gr.out[row] = t;
    if (gr.out2) gr.out2[row] = t;
  }
}


// The tables of facppg_wn_weight_grads, built on the device (one thread; ~75 entries): see there.
struct WnGradTableArgs {
  const float *a0, *spect, *h_all, *ts_all, *skip, *dout, *dpre_all, *dh_all, *dskip;
  facppg_wn_grads g;
  int n_in, n_layers, B, L;
};
__global__ void k_wn_grad_tables(WnGradTableArgs t, WgradF32Prob* __restrict__ probs, RowSumF32* __restrict__ rows) {
  constexpr int CH = 256, NC = 640, HALO_ = 128;
  const int n_in = t.n_in, n_layers = t.n_layers, B = t.B, L = t.L;
  const int Lr = round_up(L, 64), Lp = HALO_ + Lr + HALO_;
  const size_t dh_sz = (size_t)B * CH * Lr, dp_sz = (size_t)B * 2 * CH * Lr, h_sz = (size_t)B * CH * Lp;
  const facppg_wn_grads* g = &t.g;
  int np = 0, nr = 0, first = 0;
  auto prob = [&](const float* A, long a_bs, int lda, int M, const float* X, const float* X2, long x_bs, int ldx, int K, float* out,
                  int so_m, int so_k) {
    WgradF32Prob q;
    q.A = A; q.X = X; q.X2 = X2; q.out = out; q.a_bs = a_bs; q.x_bs = x_bs; q.lda = lda; q.ldx = ldx; q.M = M; q.K = K; q.so_m = so_m; q.so_k = so_k;
    probs[np++] = q;
  };
  auto rowsum = [&](const float* src, long bs, int ld, int n, float* out, float* out2) {
    RowSumF32 r;
    r.src = src; r.out = out; r.out2 = out2; r.bs = bs; r.ld = ld; r.rows = n; r.first = first;
    first += n;
    rows[nr++] = r;
  };
  const float *a0_dev = t.a0, *spect_pad_dev = t.spect, *h_all_dev = t.h_all, *ts_all_dev = t.ts_all, *skip_dev = t.skip, *dout_dev = t.dout,
              *dpre_all_dev = t.dpre_all, *dh_all_dev = t.dh_all, *dskip_dev = t.dskip;
  // start conv (glow.py:130-131): d start.w[m][c] = sum dh_0[m][n] a0[c][n]
  prob(dh_all_dev, (long)CH * Lr, Lr, CH, a0_dev, nullptr, (long)n_in * L, L, n_in, g->start_w, n_in, 1);
  rowsum(dh_all_dev, (long)CH * Lr, Lr, CH, g->start_b, nullptr);
  for (int i = 0; i < n_layers; ++i) {
    const bool last = i == n_layers - 1;
    const float* dpre = dpre_all_dev + (size_t)i * dp_sz;
    const float* ts = ts_all_dev + (size_t)i * dp_sz;
    const float* h = h_all_dev + (size_t)i * h_sz;
    const int d = 1 << i;
    for (int tap = 0; tap < 3; ++tap)   // dilated conv (glow.py:137-143): d in.w[m][c][tap] = sum dpre[m][n] h_i[c][n + (tap-1) d]
      prob(dpre, (long)2 * CH * Lr, Lr, 2 * CH, h + HALO_ + (tap - 1) * d, nullptr, (long)CH * Lp, Lp, CH, g->in_w[i] + tap, 3 * CH, 3);
    // conditioning conv (glow.py:145-147)
    prob(dpre, (long)2 * CH * Lr, Lr, 2 * CH, spect_pad_dev, nullptr, (long)NC * Lr, Lr, NC, g->cond_w[i], NC, 1);
    rowsum(dpre, (long)2 * CH * Lr, Lr, 2 * CH, g->in_b[i], g->cond_b[i]);
    // res_skip conv on the gate output tanh * sigmoid (glow.py:150-158, 166-173): res rows take dh_{i+1}, skip rows dskip
    const float* dres = dh_all_dev + (size_t)(i + 1) * dh_sz;
    if (!last) {
      prob(dres, (long)CH * Lr, Lr, CH, ts, ts + (size_t)CH * Lr, (long)2 * CH * Lr, Lr, CH, g->rs_w[i], CH, 1);
      rowsum(dres, (long)CH * Lr, Lr, CH, g->rs_b[i], nullptr);
    }
    float* skw = g->rs_w[i] + (last ? 0 : (size_t)CH * CH);
    prob(dskip_dev, (long)CH * Lr, Lr, CH, ts, ts + (size_t)CH * Lr, (long)2 * CH * Lr, Lr, CH, skw, CH, 1);
    rowsum(dskip_dev, (long)CH * Lr, Lr, CH, g->rs_b[i] + (last ? 0 : CH), nullptr);
  }
  // end conv (glow.py:160-164): d end.w[m][c] = sum dout[m][n] skip[c][n]
  prob(dout_dev, (long)2 * n_in * L, L, 2 * n_in, skip_dev, nullptr, (long)CH * Lr, Lr, CH, g->end_w, CH, 1);
  rowsum(dout_dev, (long)2 * n_in * L, L, 2 * n_in, g->end_b, nullptr);
}
}  // namespace
}  // namespace facppg

extern "C" size_t facppg_wn_weight_grads_workspace_bytes(int n_layers) {
  return (size_t)(6 * n_layers + 2) * sizeof(facppg::WgradF32Prob) + (size_t)(3 * n_layers + 2) * sizeof(facppg::RowSumF32) + 512;
}

// Lr = round_up(L, 64), Lp = 128 + Lr + 128 as in facppg_wn_forward_save / facppg_wn_backward_data, whose tensors these are.
extern "C" int facppg_wn_weight_grads(int n_in, int n_layers, const float* a0_dev, const float* spect_pad_dev, const float* h_all_dev,
                                      const float* ts_all_dev, const float* skip_dev, const float* dout_dev,
                                      const float* dpre_all_dev, const float* dh_all_dev, const float* dskip_dev, int B, int L,
                                      const facppg_wn_grads* g, void* ws_, size_t ws_bytes, void* stream_) {
  using namespace facppg;
  FACPPG_REQUIRE(a0_dev && spect_pad_dev && h_all_dev && ts_all_dev && skip_dev && dout_dev && dpre_all_dev && dh_all_dev && dskip_dev &&
                     g && ws_, FACPPG_EINVAL, "NULL argument");
  FACPPG_REQUIRE(n_in >= 1 && n_in <= 4 && n_layers >= 1 && n_layers <= 8 && B > 0 && L > 0, FACPPG_EINVAL,
                 "n_in %d, n_layers %d, B %d, L %d out of range", n_in, n_layers, B, L);
  FACPPG_REQUIRE(ws_bytes >= facppg_wn_weight_grads_workspace_bytes(n_layers), FACPPG_EWORKSPACE, "workspace has %zu bytes, need %zu",
                 ws_bytes, facppg_wn_weight_grads_workspace_bytes(n_layers));
  hipStream_t s = (hipStream_t)stream_;
  constexpr int CH = 256, NC = 640;
  // The problem / row-sum tables are built ON THE DEVICE by a one-thread launch from this call's arguments (k_wn_grad_tables):
  // nothing is copied from host memory, so the call neither stalls the host behind the stream (a copy from pageable memory
  // waits for the stream to reach it -- 12 times per backward pass) nor, under stream capture, leaves the graph a copy node
  // that re-reads a host buffer freed on return.
  WnGradTableArgs t;
  t.a0 = a0_dev; t.spect = spect_pad_dev; t.h_all = h_all_dev; t.ts_all = ts_all_dev; t.skip = skip_dev; t.dout = dout_dev;
  t.dpre_all = dpre_all_dev; t.dh_all = dh_all_dev; t.dskip = dskip_dev; t.g = *g; t.n_in = n_in; t.n_layers = n_layers; t.B = B; t.L = L;
  const int n_probs = 6 * n_layers + 1, n_rows = 3 * n_layers + 1;
  const int first = CH + n_layers * 2 * CH + (2 * n_layers - 1) * CH + 2 * n_in;   // rows the row-sum launch walks
  char* ws = (char*)ws_;
  WgradF32Prob* probs_dev = (WgradF32Prob*)ws;
  RowSumF32* rows_dev = (RowSumF32*)(ws + round_up((int)(n_probs * sizeof(WgradF32Prob)), 256));
  k_wn_grad_tables<<<1, 1, 0, s>>>(t, probs_dev, rows_dev);
  k_wgrad_f32<<<dim3((NC + WG_TK - 1) / WG_TK, (2 * CH + WG_TM - 1) / WG_TM, (unsigned)n_probs), 256, 0, s>>>(probs_dev, B, L);
  k_rowsum_f32<<<first, 256, 0, s>>>(rows_dev, n_rows, B, L);
  FACPPG_HIP_CHECK(hipGetLastError());
  return FACPPG_OK;
}
