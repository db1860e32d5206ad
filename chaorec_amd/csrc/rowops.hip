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

}  // namespace chaorec

using namespace chaorec;

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
