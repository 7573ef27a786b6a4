// Training-direction kernels of the WaveGlow flow edges (src/waveglow/glow.py:82-102, 208-250): the c x c
// channel-mixing conv of Invertible1x1Conv (c <= 8) forward / data gradient (the same kernel with the
// transposed matrix) and its weight gradient.  These are HBM-streaming kernels: one position's c channels are
// c strided 4-byte-per-lane rows, a lane owns 4 consecutive positions (16-byte accesses), the c x c matrix
// sits in SGPRs/registers.
#include <algorithm>

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

__global__ void k_sum_parts(const float* __restrict__ part, float* __restrict__ out, int n_parts, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int p = 0; p < n_parts; ++p) v += part[(size_t)p * n + i];
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
__global__ __launch_bounds__(256) void k_weight_norm_fwd(const WnTableEntry* __restrict__ tab, int n, long total_rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int lane = threadIdx.x & 63;
  const WnTableEntry e = tab[wn_find(tab, n, row)];
  const int r = (int)(row - e.row0);
  const float* v = e.v + (size_t)r * e.len;
  float ss = 0.0f;
  for (int i = lane; i < e.len; i += 64) { const float x = v[i]; ss = fmaf(x, x, ss); }
  const float norm = sqrtf(wave_sum(ss));
  const float sc = e.g[r] / norm;
  float* w = e.w + (size_t)r * e.len;
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
  float dot = 0.0f;
  for (int i = lane; i < e.len; i += 64) dot = fmaf(dw[i], v[i], dot);
  dot = wave_sum(dot);
  const float norm = e.norm[r], gr = e.g[r];
  const float a = gr / norm, bq = gr * dot / (norm * norm * norm);
  float* dv = const_cast<float*>(out[t].v) + (size_t)r * e.len;
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

// log det W and W^-T of one small mixing matrix (c <= 8) by LU with partial pivoting, one thread: what torch.logdet and its
// backward do through rocSOLVER in ~22 tiny launches per flow and direction (glow.py:100: log_det_W = B * L * logdet(W)).
// det <= 0 follows torch.logdet: NaN for a negative determinant, -inf for a singular matrix.
__global__ __launch_bounds__(64) void k_logdet(const float* __restrict__ W, int c, float* __restrict__ logdet, float* __restrict__ winv_t) {
  // one wave; lane (i, j) owns element [i][j] of the matrix and of the accumulating inverse (Gauss-Jordan on [A | I] with
  // partial pivoting).  (A one-thread version with the two 8 x 8 arrays in scratch took 72 us per call.)
  __shared__ float A[8][8], Iv[8][8];
  __shared__ int s_piv, s_singular;
  __shared__ float s_sign, s_log;
  const int tid = threadIdx.x, i = tid >> 3, j = tid & 7;
  const bool in = i < c && j < c;
  A[i][j] = in ? W[i * c + j] : (i == j ? 1.0f : 0.0f);
  Iv[i][j] = i == j ? 1.0f : 0.0f;
  if (tid == 0) { s_sign = 1.0f; s_log = 0.0f; s_singular = 0; }
  __syncthreads();
  for (int k = 0; k < c; ++k) {
    if (tid == 0) {
      int piv = k;
      float best = fabsf(A[k][k]);
      for (int r = k + 1; r < c; ++r)
        if (fabsf(A[r][k]) > best) { best = fabsf(A[r][k]); piv = r; }
      s_piv = piv;
      if (best == 0.0f) s_singular = 1;
    }
    __syncthreads();
    if (s_singular) break;
    const int piv = s_piv;
    if (piv != k && i == k) {   // row k's lanes exchange rows k and piv
      float t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t;
      t = Iv[k][j]; Iv[k][j] = Iv[piv][j]; Iv[piv][j] = t;
    }
    __syncthreads();
    const float d = A[k][k];
    __syncthreads();
    if (tid == 0) {
      if (piv != k) s_sign = -s_sign;
      if (d < 0.0f) s_sign = -s_sign;
      s_log += logf(fabsf(d));
    }
    if (i == k) { const float r = 1.0f / d; A[k][j] *= r; Iv[k][j] *= r; }
    __syncthreads();
    const float f = A[i][k], akj = A[k][j], ikj = Iv[k][j];
    __syncthreads();
    if (i != k) { A[i][j] = fmaf(-f, akj, A[i][j]); Iv[i][j] = fmaf(-f, ikj, Iv[i][j]); }
    __syncthreads();
  }
  if (s_singular) {
    if (tid == 0) *logdet = -INFINITY;
    if (in) winv_t[i * c + j] = NAN;
    return;
  }
  if (tid == 0) *logdet = s_sign > 0.0f ? s_log : NAN;
  if (in) winv_t[i * c + j] = Iv[j][i];
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
  k_sum_parts<<<1, 64, 0, s>>>(part, dw_dev, parts, c * c);
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
