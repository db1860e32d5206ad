// Row-wise fused ops on [N, D] fp32 tables in HBM (one pass, one launch), for the models of the torch.sparse.mm
// family that re-weight every propagated layer.
//
//   * LayerGCN (Model/LayerGCN.py:125-127):  w = cosine_similarity(y, e, dim=-1);  out = einsum('a,ab->ab', w, y)
//     The reference runs this as ~12 elementwise / reduction launches per layer forward and ~25 backward; here it
//     is one launch each way, placed right after the SpMM that produced y.
//
// Layout: a group of LPR lanes owns one row (float4 per lane per chunk, like the SpMM kernel), NG = 64 / LPR rows
// per wave; the three row reductions are butterfly shuffles inside the group, in a fixed order.  HBM-bound:
// forward reads y, e and writes out (12 B per element), backward reads g, y, e and writes g_y, g_e (20 B).
#include "common.h"

namespace chaorec {

constexpr float kCosEps = 1e-8f;   // F.cosine_similarity's eps (clamp on each norm)

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// out = w * y with w = <y, e> / (max(|y|, eps) * max(|e|, eps));  w_out[r] = w (optional)
template <int LPR, int CPL>
__global__ __launch_bounds__(256) void row_cosine_scale_fwd_kernel(const float4 *__restrict__ y,
                                                                   const float4 *__restrict__ e,
                                                                   float4 *__restrict__ out, float *__restrict__ w_out,
                                                                   int64_t n_rows, int D4) {
  constexpr int NG = kWave / LPR;
  const int lane = threadIdx.x & 63, li = lane % LPR;
  const int64_t r = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NG + lane / LPR;
  const bool ok = r < n_rows;
  float4 yv[CPL], ev[CPL];
  float dot = 0.f, ny = 0.f, ne = 0.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = li + q * LPR;
    yv[q] = ev[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && c < D4) {
      yv[q] = y[(size_t)r * D4 + c];
      ev[q] = e[(size_t)r * D4 + c];
    }
    dot += yv[q].x * ev[q].x + yv[q].y * ev[q].y + yv[q].z * ev[q].z + yv[q].w * ev[q].w;
    ny += yv[q].x * yv[q].x + yv[q].y * yv[q].y + yv[q].z * yv[q].z + yv[q].w * yv[q].w;
    ne += ev[q].x * ev[q].x + ev[q].y * ev[q].y + ev[q].z * ev[q].z + ev[q].w * ev[q].w;
  }
  dot = group_sum<LPR>(dot);
  ny = group_sum<LPR>(ny);
  ne = group_sum<LPR>(ne);
  const float w = dot / (fmaxf(sqrtf(ny), kCosEps) * fmaxf(sqrtf(ne), kCosEps));
  if (!ok) return;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = li + q * LPR;
    if (c < D4) out[(size_t)r * D4 + c] = make_float4(w * yv[q].x, w * yv[q].y, w * yv[q].z, w * yv[q].w);
  }
  if (w_out && li == 0) w_out[r] = w;
}

// With a = max(|y|, eps), b = max(|e|, eps), w = <y,e>/(a b), s = <g, y>:
//   g_y = w g + s (e/(a b) - [|y| > eps] w y / a^2)       g_e = s (y/(a b) - [|e| > eps] w e / b^2)
template <int LPR, int CPL>
__global__ __launch_bounds__(256) void row_cosine_scale_bwd_kernel(const float4 *__restrict__ g,
                                                                   const float4 *__restrict__ y,
                                                                   const float4 *__restrict__ e,
                                                                   float4 *__restrict__ gy, float4 *__restrict__ ge,
                                                                   int64_t n_rows, int D4) {
  constexpr int NG = kWave / LPR;
  const int lane = threadIdx.x & 63, li = lane % LPR;
  const int64_t r = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NG + lane / LPR;
  const bool ok = r < n_rows;
  float4 yv[CPL], ev[CPL], gv[CPL];
  float dot = 0.f, ny = 0.f, ne = 0.f, s = 0.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = li + q * LPR;
    yv[q] = ev[q] = gv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && c < D4) {
      yv[q] = y[(size_t)r * D4 + c];
      ev[q] = e[(size_t)r * D4 + c];
      gv[q] = g[(size_t)r * D4 + c];
    }
    dot += yv[q].x * ev[q].x + yv[q].y * ev[q].y + yv[q].z * ev[q].z + yv[q].w * ev[q].w;
    ny += yv[q].x * yv[q].x + yv[q].y * yv[q].y + yv[q].z * yv[q].z + yv[q].w * yv[q].w;
    ne += ev[q].x * ev[q].x + ev[q].y * ev[q].y + ev[q].z * ev[q].z + ev[q].w * ev[q].w;
    s += gv[q].x * yv[q].x + gv[q].y * yv[q].y + gv[q].z * yv[q].z + gv[q].w * yv[q].w;
  }
  dot = group_sum<LPR>(dot);
  ny = group_sum<LPR>(ny);
  ne = group_sum<LPR>(ne);
  s = group_sum<LPR>(s);
  if (!ok) return;
  const float nys = sqrtf(ny), nes = sqrtf(ne);
  const float a = fmaxf(nys, kCosEps), b = fmaxf(nes, kCosEps);
  const float iab = 1.f / (a * b);
  const float w = dot * iab;
  const float ky = nys > kCosEps ? w / (a * a) : 0.f;
  const float ke = nes > kCosEps ? w / (b * b) : 0.f;
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int c = li + q * LPR;
    if (c >= D4) continue;
    float4 oy, oe;
    oy.x = w * gv[q].x + s * (ev[q].x * iab - ky * yv[q].x);
    oy.y = w * gv[q].y + s * (ev[q].y * iab - ky * yv[q].y);
    oy.z = w * gv[q].z + s * (ev[q].z * iab - ky * yv[q].z);
    oy.w = w * gv[q].w + s * (ev[q].w * iab - ky * yv[q].w);
    oe.x = s * (yv[q].x * iab - ke * ev[q].x);
    oe.y = s * (yv[q].y * iab - ke * ev[q].y);
    oe.z = s * (yv[q].z * iab - ke * ev[q].z);
    oe.w = s * (yv[q].w * iab - ke * ev[q].w);
    gy[(size_t)r * D4 + c] = oy;
    ge[(size_t)r * D4 + c] = oe;
  }
}

template <int LPR, int CPL>
static int launch_row_cosine(const float *g, const float *y, const float *e, float *o1, float *o2, float *w,
                             int64_t n_rows, int D4, hipStream_t st) {
  constexpr int NG = kWave / LPR;
  const int64_t waves = (n_rows + NG - 1) / NG;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  if (!g)
    hipLaunchKernelGGL((row_cosine_scale_fwd_kernel<LPR, CPL>), dim3(blocks), dim3(256), 0, st, (const float4 *)y,
                       (const float4 *)e, (float4 *)o1, w, n_rows, D4);
  else
    hipLaunchKernelGGL((row_cosine_scale_bwd_kernel<LPR, CPL>), dim3(blocks), dim3(256), 0, st, (const float4 *)g,
                       (const float4 *)y, (const float4 *)e, (float4 *)o1, (float4 *)o2, n_rows, D4);
  return check_launch("row_cosine_scale");
}

static int dispatch_row_cosine(const float *g, const float *y, const float *e, float *o1, float *o2, float *w,
                               int64_t n_rows, int32_t D, hipStream_t st) {
  const int D4 = D / 4;
#define CHAOREC_RC_ARGS g, y, e, o1, o2, w, n_rows, D4, st
  if (D4 <= 1) return launch_row_cosine<1, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 2) return launch_row_cosine<2, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 4) return launch_row_cosine<4, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 8) return launch_row_cosine<8, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 16) return launch_row_cosine<16, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 32) return launch_row_cosine<32, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 64) return launch_row_cosine<64, 1>(CHAOREC_RC_ARGS);
  if (D4 <= 128) return launch_row_cosine<64, 2>(CHAOREC_RC_ARGS);
  return launch_row_cosine<64, 4>(CHAOREC_RC_ARGS);
#undef CHAOREC_RC_ARGS
}

// ---- deterministic reductions without semaphores or memset nodes ------------------------------------------------
// torch's multi-block reductions (x.sum(0) over many rows, .mean() over millions of elements) clear a semaphore buffer
// with cudaMemsetAsync; inside a captured hipGraph on this stack that memset node does not replay, and from the second
// replay on the reduction returns stale or garbage values.  These two-pass kernels need neither: pass 1
// writes one partial per (row chunk, column) / per block, pass 2 adds the partials in a fixed order.
// rows per pass-1 block of the column sum: 128 for narrow tables (a [60 k, 64] sum is one column block wide, so 512-row
// chunks give 118 workgroups on 256 CUs: 17.8 us, 128-row chunks 13.1 us), 512 for wide ones (768 columns: 32 vs 40 us)
__host__ __device__ constexpr int red_chunk_rows(int64_t N) { return N <= 128 ? 128 : 512; }
constexpr int kRedMaxBlocks = 1024;    // pass-1 blocks of the scalar sum

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, int64_t M, int64_t N,
                                                             int64_t ldx, float *__restrict__ part, int chunk_rows) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * chunk_rows, r1 = min(M, r0 + chunk_rows);
  float v = 0.f;
  if (col < N)
    for (int64_t r = r0 + q; r < r1; r += 4) v += x[r * ldx + col];
  red[q][c] = v;
  __syncthreads();
  if (q == 0 && col < N) part[(int64_t)blockIdx.y * N + col] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// float4 form (N and ldx multiples of 4, 16-byte aligned base): 16 column quads x 16 row lanes per block, four
// independent accumulators per thread so that the row loads overlap
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float4 *__restrict__ x4, int64_t M, int64_t N4,
                                                              int64_t ld4, float4 *__restrict__ part4, int chunk_rows) {
  __shared__ float4 red[16][16];
  const int cq = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int64_t col = (int64_t)blockIdx.x * 16 + cq;
  const int64_t r0 = (int64_t)blockIdx.y * chunk_rows, r1 = min(M, r0 + chunk_rows);
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N4) {
    int64_t r = r0 + q;
    for (; r + 48 < r1; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = x4[(r + 16 * u) * ld4 + col];
        a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
      }
    }
    for (; r < r1; r += 16) {
      const float4 v = x4[r * ld4 + col];
      a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
  }
  float4 t;
  t.x = (a[0].x + a[1].x) + (a[2].x + a[3].x);
  t.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
  t.z = (a[0].z + a[1].z) + (a[2].z + a[3].z);
  t.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
  red[q][cq] = t;
  __syncthreads();
  if (q == 0 && col < N4) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 o = red[k][cq];
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    part4[(int64_t)blockIdx.y * N4 + col] = t;
  }
}

// pass 2: 64 columns per block, the chunks of a column over kFinalWaves waves (each with four independent accumulators, so
// the loads of a wave are in flight together: one thread walking 118 chunk rows of a 60 k-row sum alone took 27 us, four
// waves 9.5 us -- 24 times per MMGCN step), combined in a fixed order
constexpr int kFinalWaves = 16;
__global__ __launch_bounds__(64 * kFinalWaves) void colsum_final_kernel(const float *__restrict__ part, int64_t chunks,
                                                                        int64_t N, float *__restrict__ out) {
  __shared__ float red[kFinalWaves][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + tx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < N) {
    int64_t k = ty;
    for (; k + 3 * kFinalWaves < chunks; k += 4 * kFinalWaves) {
      a0 += part[k * N + col];
      a1 += part[(k + kFinalWaves) * N + col];
      a2 += part[(k + 2 * kFinalWaves) * N + col];
      a3 += part[(k + 3 * kFinalWaves) * N + col];
    }
    for (; k < chunks; k += kFinalWaves) a0 += part[k * N + col];
  }
  red[ty][tx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ty == 0 && col < N) {
    float v = red[0][tx];
#pragma unroll
    for (int w = 1; w < kFinalWaves; ++w) v += red[w][tx];
    out[col] = v;
  }
}

__global__ __launch_bounds__(256) void sum_partial_kernel(const float *__restrict__ x, int64_t n,
                                                          float *__restrict__ part) {
  __shared__ float red[256];
  float v = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v += x[i];
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sum_final_kernel(const float *__restrict__ part, int blocks, float scale,
                                                        float *__restrict__ out) {
  __shared__ float red[256];
  float v = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) v += part[i];
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// ---- NGCF's elementwise backward (Model/NGCF.py:60-84 restated as out = leaky_0.2(s W1^T + (s * x) W2^T)) --------------
// g = leaky_relu'(y) * gy  (one launch instead of compare + multiply + where)
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float4 *__restrict__ y, const float4 *__restrict__ gy,
                                                        float slope, float4 *__restrict__ g, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = y[i], b = gy[i];
    g[i] = make_float4(a.x > 0.f ? b.x : b.x * slope, a.y > 0.f ? b.y : b.y * slope, a.z > 0.f ? b.z : b.z * slope,
                       a.w > 0.f ? b.w : b.w * slope);
  }
}
// ---- MMGCN's layer tail (Model/MMGCN.py:102-131: h = leaky_relu(conv(x)); u = leaky_relu(linear(x)) + id_embedding;
// x' = leaky_relu(g_layer(cat(h, u)))) --------------------------------------------------------------------------------
// out[r] = [leaky(s[r]) | u[r] + id[r]]: the activation of the propagated half, the id residual of the other half and the
// concatenation in ONE pass (torch: leaky_relu + add + cat = 3 launches, 10 row passes instead of 5).  id may be NULL.
__global__ __launch_bounds__(256) void leaky_cat_add_kernel(const float4 *__restrict__ s, const float4 *__restrict__ u,
                                                            const float4 *__restrict__ id, float4 *__restrict__ out,
                                                            int64_t n_rows, int d1q, int d2q, float slope) {
  const int tq = d1q + d2q;
  const int64_t total = n_rows * tq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / tq;
    const int c = (int)(i - r * tq);
    float4 v;
    if (c < d1q) {
      const float4 a = s[r * d1q + c];
      v = make_float4(a.x > 0.f ? a.x : a.x * slope, a.y > 0.f ? a.y : a.y * slope, a.z > 0.f ? a.z : a.z * slope,
                      a.w > 0.f ? a.w : a.w * slope);
    } else {
      v = u[r * d2q + (c - d1q)];
      if (id) {
        const float4 b = id[r * d2q + (c - d1q)];
        v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
      }
    }
    out[i] = v;
  }
}
// backwards: the gradient of the concatenation [N, D1 + D2] -> gs = g[:, :D1] * leaky'(h) (h = the stored first half of
// the concatenation: leaky(s) > 0 <=> s > 0), gu = g[:, D1:] * leaky'(uy) (uy = the activated Linear output the residual
// was added to), both contiguous (torch: two strided copies + two activation backwards).  gid (optional) = g[:, D1:].
__global__ __launch_bounds__(256) void leaky_split_bwd_kernel(const float4 *__restrict__ g, const float4 *__restrict__ cat,
                                                              const float4 *__restrict__ uy, float4 *__restrict__ gs,
                                                              float4 *__restrict__ gu, float4 *__restrict__ gid,
                                                              int64_t n_rows, int d1q, int d2q, float slope) {
  const int tq = d1q + d2q;
  const int64_t total = n_rows * tq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / tq;
    const int c = (int)(i - r * tq);
    const float4 b = g[i];
    if (c < d1q) {
      const float4 a = cat[i];
      gs[r * d1q + c] = make_float4(a.x > 0.f ? b.x : b.x * slope, a.y > 0.f ? b.y : b.y * slope,
                                    a.z > 0.f ? b.z : b.z * slope, a.w > 0.f ? b.w : b.w * slope);
    } else {
      const int64_t j = r * d2q + (c - d1q);
      const float4 a = uy[j];
      gu[j] = make_float4(a.x > 0.f ? b.x : b.x * slope, a.y > 0.f ? b.y : b.y * slope, a.z > 0.f ? b.z : b.z * slope,
                          a.w > 0.f ? b.w : b.w * slope);
      if (gid) gid[j] = b;
    }
  }
}

// ---- F.normalize(cat(a, b), p=2, dim=1) (Model/MMGCN.py:99-100: x = F.normalize(cat(preference, features))) ---------
// One wave per row: y = x / max(|x|, eps), the row norm kept for the backward.  Rows [0, na) come from `a`, the rest from
// `b` (the concatenation is never materialised).  torch: cat + norm + clamp + expand + div = 4 launches forward and 8
// backward over [N, 256]; here one each.
__global__ __launch_bounds__(256) void normalize_rows_fwd_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                                 int64_t na, int64_t n_rows, int dq, float eps,
                                                                 float4 *__restrict__ y, float *__restrict__ norm) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += (int64_t)gridDim.x * 4) {
    const float4 *src = r < na ? a + r * dq : b + (r - na) * dq;
    float ss = 0.f;
    for (int c = lane; c < dq; c += 64) {
      const float4 v = src[c];
      ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float nr = sqrtf(ss), d = fmaxf(nr, eps);
    for (int c = lane; c < dq; c += 64) {
      const float4 v = src[c];
      y[r * dq + c] = make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
    }
    if (lane == 0) norm[r] = nr;
  }
}
// gx = (gy - y <gy, y>) / |x|  where |x| >= eps (the clamp passes the gradient there), gy / eps below it; rows < r_skip
// are not written (the caller does not need the gradient of `a`: MMGCN's preference is no Parameter, Q2)
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float4 *__restrict__ gy, const float4 *__restrict__ y,
                                                                 const float *__restrict__ norm, int64_t r_skip,
                                                                 int64_t n_rows, int dq, float eps,
                                                                 float4 *__restrict__ gx) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = r_skip + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += (int64_t)gridDim.x * 4) {
    const float nr = norm[r];
    float dot = 0.f;
    if (nr >= eps) {
      for (int c = lane; c < dq; c += 64) {
        const float4 g = gy[r * dq + c], v = y[r * dq + c];
        dot += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    }
    const float d = fmaxf(nr, eps);
    for (int c = lane; c < dq; c += 64) {
      const float4 g = gy[r * dq + c], v = y[r * dq + c];
      gx[r * dq + c] = make_float4((g.x - v.x * dot) / d, (g.y - v.y * dot) / d, (g.z - v.z * dot) / d, (g.w - v.w * dot) / d);
    }
  }
}

// the product t = s * x backwards, and the sum with s's other gradient:  gs += gt * x;  gx = gt * s
// (separately rounded product and sum: what torch's mul + add give)
__global__ __launch_bounds__(256) void mul_pair_bwd_kernel(const float4 *__restrict__ gt, const float4 *__restrict__ s,
                                                           const float4 *__restrict__ x, float4 *__restrict__ gs,
                                                           float4 *__restrict__ gx, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 t = gt[i], a = s[i], b = x[i], c = gs[i];
    gs[i] = make_float4(__fadd_rn(c.x, __fmul_rn(t.x, b.x)), __fadd_rn(c.y, __fmul_rn(t.y, b.y)),
                        __fadd_rn(c.z, __fmul_rn(t.z, b.z)), __fadd_rn(c.w, __fmul_rn(t.w, b.w)));
    gx[i] = make_float4(t.x * a.x, t.y * a.y, t.z * a.z, t.w * a.w);
  }
}

}  // namespace chaorec

using namespace chaorec;

extern "C" size_t chaorec_reduce_workspace_bytes(int64_t M, int64_t N) {
  const int cr = red_chunk_rows(N);
  const int64_t chunks = (M + cr - 1) / cr;
  const size_t col = (size_t)(chunks > 0 ? chunks : 1) * (size_t)(N > 0 ? N : 1) * sizeof(float);
  const size_t sc = (size_t)kRedMaxBlocks * sizeof(float);
  return col > sc ? col : sc;
}

extern "C" int chaorec_colsum_f32(const float *x, int64_t M, int64_t N, int64_t ldx, float *out, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  if (!x || !out || !workspace) return fail(CHAOREC_E_INVALID, "colsum: NULL argument");
  if (M < 0 || N <= 0 || ldx < N) return fail(CHAOREC_E_INVALID, "colsum: M=%lld N=%lld ldx=%lld", (long long)M, (long long)N, (long long)ldx);
  if (workspace_bytes < chaorec_reduce_workspace_bytes(M, N))
    return fail(CHAOREC_E_WORKSPACE, "colsum: workspace %zu < %zu", workspace_bytes, chaorec_reduce_workspace_bytes(M, N));
  hipStream_t st = (hipStream_t)stream;
  const int cr = red_chunk_rows(N);
  const int64_t chunks = (M + cr - 1) / cr;
  if (chunks > 65535) return fail(CHAOREC_E_INVALID, "colsum: M=%lld too large", (long long)M);
  float *part = (float *)workspace;
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(part) & 15) == 0);
  if (chunks > 0 && vec)
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3((unsigned)((N / 4 + 15) / 16), (unsigned)chunks), dim3(256), 0, st,
                       (const float4 *)x, M, N / 4, ldx / 4, (float4 *)part, cr);
  else if (chunks > 0)
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, st, x, M, N,
                       ldx, part, cr);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64 * kFinalWaves), 0, st, part, chunks, N, out);
  return check_launch("colsum");
}

extern "C" int chaorec_sum_f32(const float *x, int64_t n, float scale, float *out, void *workspace,
                               size_t workspace_bytes, void *stream) {
  if (!x || !out || !workspace) return fail(CHAOREC_E_INVALID, "sum: NULL argument");
  if (n < 0) return fail(CHAOREC_E_INVALID, "sum: n=%lld", (long long)n);
  if (workspace_bytes < (size_t)kRedMaxBlocks * sizeof(float)) return fail(CHAOREC_E_WORKSPACE, "sum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((n + 4095) / 4096);
  if (blocks < 1) blocks = 1;
  if (blocks > kRedMaxBlocks) blocks = kRedMaxBlocks;
  float *part = (float *)workspace;
  hipLaunchKernelGGL(sum_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n, part);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, part, blocks, scale, out);
  return check_launch("sum");
}

extern "C" int chaorec_row_cosine_scale_fwd_f32(const float *y, const float *e, float *out, float *w_out,
                                                int64_t n_rows, int32_t D, void *stream) {
  if (!y || !e || !out) return fail(CHAOREC_E_INVALID, "row_cosine_scale_fwd: NULL argument");
  if (n_rows < 0 || D < 4 || D > 1024 || (D & 3))
    return fail(CHAOREC_E_INVALID, "row_cosine_scale_fwd: n_rows=%lld D=%d (D: multiple of 4 in [4,1024])", (long long)n_rows, D);
  if (n_rows == 0) return CHAOREC_OK;
  return dispatch_row_cosine(nullptr, y, e, out, nullptr, w_out, n_rows, D, (hipStream_t)stream);
}

extern "C" int chaorec_row_cosine_scale_bwd_f32(const float *grad_out, const float *y, const float *e, float *grad_y,
                                                float *grad_e, int64_t n_rows, int32_t D, void *stream) {
  if (!grad_out || !y || !e || !grad_y || !grad_e) return fail(CHAOREC_E_INVALID, "row_cosine_scale_bwd: NULL argument");
  if (n_rows < 0 || D < 4 || D > 1024 || (D & 3))
    return fail(CHAOREC_E_INVALID, "row_cosine_scale_bwd: n_rows=%lld D=%d (D: multiple of 4 in [4,1024])", (long long)n_rows, D);
  if (n_rows == 0) return CHAOREC_OK;
  return dispatch_row_cosine(grad_out, y, e, grad_y, grad_e, nullptr, n_rows, D, (hipStream_t)stream);
}

// out = w t_0 + w t_1 + ... + w t_{k-1}, accumulated in that order with separately rounded products and sums: the
// association of LightGCN's layer mean (Model/LightGCN.py:86-93: final = 0 + w x_0 + w x_1 + ...), i.e. of
// chaorec_spmm_csr_mean_f32's epilogue.  Pointers by value in the kernel argument (capturable like any argument).
struct MeanTerms {
  const float4 *t[8];
  int n;
};
__global__ __launch_bounds__(256) void rows_mean_kernel(const MeanTerms T, float w, float4 *__restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = mul_rn4(w, T.t[0][i]);
    for (int k = 1; k < T.n; ++k) {
      const float4 v = mul_rn4(w, T.t[k][i]);
      a = make_float4(add_rn(a.x, v.x), add_rn(a.y, v.y), add_rn(a.z, v.z), add_rn(a.w, v.w));
    }
    out[i] = a;
  }
}

extern "C" int chaorec_rows_mean_f32(const float *const *terms, int32_t n_terms, float w, float *out, int64_t n,
                                     void *stream) {
  if (!terms || !out) return fail(CHAOREC_E_INVALID, "rows_mean: null pointer");
  if (n_terms < 1 || n_terms > 8) return fail(CHAOREC_E_INVALID, "rows_mean: n_terms=%d must be in [1, 8]", n_terms);
  if (n < 0 || n % 4) return fail(CHAOREC_E_INVALID, "rows_mean: n=%lld must be a non-negative multiple of 4", (long long)n);
  if (n == 0) return CHAOREC_OK;
  MeanTerms T;
  T.n = n_terms;
  for (int k = 0; k < 8; ++k) T.t[k] = (const float4 *)(k < n_terms ? terms[k] : terms[0]);
  const int64_t n4 = n / 4;
  const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 4096);
  rows_mean_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(T, w, (float4 *)out, n4);
  return check_launch("rows_mean");
}

extern "C" int chaorec_leaky_bwd_f32(const float *y, const float *grad_out, float slope, float *grad_in, int64_t n,
                                     void *stream) {
  if (!y || !grad_out || !grad_in) return fail(CHAOREC_E_INVALID, "leaky_bwd: null pointer");
  if (n < 0 || n % 4) return fail(CHAOREC_E_INVALID, "leaky_bwd: n=%lld must be a non-negative multiple of 4", (long long)n);
  if (n == 0) return CHAOREC_OK;
  const int64_t n4 = n / 4;
  const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 4096);
  leaky_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4 *)y, (const float4 *)grad_out, slope,
                                                           (float4 *)grad_in, n4);
  return check_launch("leaky_bwd");
}

extern "C" int chaorec_mul_pair_bwd_f32(const float *grad_t, const float *s, const float *x, float *grad_s, float *grad_x,
                                        int64_t n, void *stream) {
  if (!grad_t || !s || !x || !grad_s || !grad_x) return fail(CHAOREC_E_INVALID, "mul_pair_bwd: null pointer");
  if (n < 0 || n % 4) return fail(CHAOREC_E_INVALID, "mul_pair_bwd: n=%lld must be a non-negative multiple of 4", (long long)n);
  if (n == 0) return CHAOREC_OK;
  const int64_t n4 = n / 4;
  const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 4096);
  mul_pair_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4 *)grad_t, (const float4 *)s, (const float4 *)x,
                                                              (float4 *)grad_s, (float4 *)grad_x, n4);
  return check_launch("mul_pair_bwd");
}

// ---- edge scores over a CSR's entries (sampled dense-dense product) -----------------------------------------------------
// out[k] = <a[row_k], b[col_k]> for every stored entry k.  HBM / L2-gather bound: two D-float rows per entry.  A group of
// LPE lanes takes one entry (float4 per lane and pass, LPE = the power of two covering D / 4, at most 64), 64 / LPE entries
// per wave at a time; the partial sums meet through xor-shuffles inside the group.  Entries are walked in storage order,
// so the a-row of consecutive entries is the same row (one L2 line set), the b-rows are the gathers.
template <int LPE>
__global__ __launch_bounds__(256) void edge_dot_kernel(const int32_t *__restrict__ entry_row, const int32_t *__restrict__ col,
                                                       const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                       float *__restrict__ out, int64_t nnz, int d4) {
  constexpr int EPW = 64 / LPE;                       // entries per wave and round
  const int lane = threadIdx.x & 63, sub = lane % LPE, slot = lane / LPE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t k0 = wave * EPW; k0 < nnz; k0 += n_waves * EPW) {
    const int64_t k = k0 + slot;
    float acc = 0.f;
    if (k < nnz) {
      const float4 *ar = a + (int64_t)entry_row[k] * d4, *br = b + (int64_t)col[k] * d4;
      for (int q = sub; q < d4; q += LPE) {
        const float4 x = ar[q], y = br[q];
        acc = fmaf(x.x, y.x, acc);
        acc = fmaf(x.y, y.y, acc);
        acc = fmaf(x.z, y.z, acc);
        acc = fmaf(x.w, y.w, acc);
      }
    }
#pragma unroll
    for (int j = LPE / 2; j > 0; j >>= 1) acc += __shfl_xor(acc, j, 64);
    if (sub == 0 && k < nnz) out[k] = acc;
  }
}

extern "C" int chaorec_edge_dot_f32(const int32_t *entry_row, const int32_t *col, const float *a, const float *b, float *out,
                                    int64_t nnz, int32_t D, void *stream) {
  if (!entry_row || !col || !a || !b || !out) return fail(CHAOREC_E_INVALID, "edge_dot: null pointer");
  if (nnz < 0 || D < 4 || (D & 3)) return fail(CHAOREC_E_INVALID, "edge_dot: nnz=%lld D=%d (a multiple of 4)", (long long)nnz, D);
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15)
    return fail(CHAOREC_E_INVALID, "edge_dot: a and b must be 16-byte aligned (rows are read as float4)");
  if (nnz == 0) return CHAOREC_OK;
  const int d4 = D / 4;
  int lpe = 1;
  while (lpe < d4 && lpe < 64) lpe <<= 1;
  const int64_t waves = (nnz + (64 / lpe) - 1) / (64 / lpe);
  const unsigned blocks = (unsigned)std::min<int64_t>((waves + 3) / 4, 16384);
  hipStream_t st = (hipStream_t)stream;
  const float4 *a4 = (const float4 *)a, *b4 = (const float4 *)b;
  switch (lpe) {
    case 1: edge_dot_kernel<1><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
    case 2: edge_dot_kernel<2><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
    case 4: edge_dot_kernel<4><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
    case 8: edge_dot_kernel<8><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
    case 16: edge_dot_kernel<16><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
    case 32: edge_dot_kernel<32><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
    default: edge_dot_kernel<64><<<blocks, 256, 0, st>>>(entry_row, col, a4, b4, out, nnz, d4); break;
  }
  return check_launch("edge_dot");
}

extern "C" int chaorec_leaky_cat_add_f32(const float *s, const float *u, const float *id, float *out, int64_t n_rows,
                                         int32_t d1, int32_t d2, float slope, void *stream) {
  if (!s || !u || !out) return fail(CHAOREC_E_INVALID, "leaky_cat_add: null pointer");
  if (n_rows < 0 || d1 < 4 || d2 < 4 || (d1 & 3) || (d2 & 3))
    return fail(CHAOREC_E_INVALID, "leaky_cat_add: n_rows=%lld d1=%d d2=%d (multiples of 4)", (long long)n_rows, d1, d2);
  if (n_rows == 0) return CHAOREC_OK;
  const int64_t total = n_rows * ((d1 + d2) / 4);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
  leaky_cat_add_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4 *)s, (const float4 *)u, (const float4 *)id,
                                                               (float4 *)out, n_rows, d1 / 4, d2 / 4, slope);
  return check_launch("leaky_cat_add");
}

extern "C" int chaorec_leaky_split_bwd_f32(const float *grad_cat, const float *cat, const float *uy, float *grad_s,
                                           float *grad_u, float *grad_id, int64_t n_rows, int32_t d1, int32_t d2,
                                           float slope, void *stream) {
  if (!grad_cat || !cat || !uy || !grad_s || !grad_u) return fail(CHAOREC_E_INVALID, "leaky_split_bwd: null pointer");
  if (n_rows < 0 || d1 < 4 || d2 < 4 || (d1 & 3) || (d2 & 3))
    return fail(CHAOREC_E_INVALID, "leaky_split_bwd: n_rows=%lld d1=%d d2=%d (multiples of 4)", (long long)n_rows, d1, d2);
  if (n_rows == 0) return CHAOREC_OK;
  const int64_t total = n_rows * ((d1 + d2) / 4);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
  leaky_split_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4 *)grad_cat, (const float4 *)cat,
                                                                 (const float4 *)uy, (float4 *)grad_s, (float4 *)grad_u,
                                                                 (float4 *)grad_id, n_rows, d1 / 4, d2 / 4, slope);
  return check_launch("leaky_split_bwd");
}

extern "C" int chaorec_normalize_rows_fwd_f32(const float *a, const float *b, int64_t rows_a, int64_t n_rows, int32_t D,
                                              float eps, float *y, float *norm, void *stream) {
  if (!a || !y || !norm || (rows_a < n_rows && !b)) return fail(CHAOREC_E_INVALID, "normalize_rows_fwd: null pointer");
  if (n_rows < 0 || rows_a < 0 || rows_a > n_rows || D < 4 || (D & 3))
    return fail(CHAOREC_E_INVALID, "normalize_rows_fwd: rows_a=%lld n_rows=%lld D=%d", (long long)rows_a, (long long)n_rows, D);
  if (n_rows == 0) return CHAOREC_OK;
  const unsigned blocks = (unsigned)std::min<int64_t>((n_rows + 3) / 4, 16384);
  normalize_rows_fwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4 *)a, (const float4 *)b, rows_a, n_rows,
                                                                    D / 4, eps, (float4 *)y, norm);
  return check_launch("normalize_rows_fwd");
}

extern "C" int chaorec_normalize_rows_bwd_f32(const float *grad_y, const float *y, const float *norm, int64_t skip_rows,
                                              int64_t n_rows, int32_t D, float eps, float *grad_x, void *stream) {
  if (!grad_y || !y || !norm || !grad_x) return fail(CHAOREC_E_INVALID, "normalize_rows_bwd: null pointer");
  if (n_rows < 0 || skip_rows < 0 || skip_rows > n_rows || D < 4 || (D & 3))
    return fail(CHAOREC_E_INVALID, "normalize_rows_bwd: skip=%lld n_rows=%lld D=%d", (long long)skip_rows, (long long)n_rows, D);
  if (n_rows == skip_rows) return CHAOREC_OK;
  const unsigned blocks = (unsigned)std::min<int64_t>((n_rows - skip_rows + 3) / 4, 16384);
  normalize_rows_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4 *)grad_y, (const float4 *)y, norm, skip_rows,
                                                                    n_rows, D / 4, eps, (float4 *)grad_x);
  return check_launch("normalize_rows_bwd");
}
