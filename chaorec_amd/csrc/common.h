// Shared helpers for libchaorec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/chaorec_hip.h"

namespace chaorec {

// Per-thread last-error text returned by chaorec_last_error().
inline char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CHAOREC_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return CHAOREC_OK;
}

constexpr int kWave = 64;  // CDNA wavefront width

// Separately rounded fp32 product and sum: the compiler must not contract these into an FMA,
// the reference's message()/scatter_add_ pair rounds twice (Model/LightGCN.py:40-43).
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }

__device__ __forceinline__ float4 mul_rn4(float s, float4 v) {
  return make_float4(mul_rn(s, v.x), mul_rn(s, v.y), mul_rn(s, v.z), mul_rn(s, v.w));
}
__device__ __forceinline__ float4 add_rn4(float4 a, float4 b) {
  return make_float4(add_rn(a.x, b.x), add_rn(a.y, b.y), add_rn(a.z, b.z), add_rn(a.w, b.w));
}

// splitmix64 finaliser: the counter-based generator behind the negative sampler (bpr.hip) and the edge dropout
// (graph_dropout.hip).  Stateless: a draw depends only on its counters, never on launch geometry.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// torch._single_tensor_adam on one element (main.py:397 defaults): exp_avg.lerp_(grad, 1-b1);
// exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom.
// Shared by adam_step_kernel and the SpMM kernel's Adam epilogue: same expression, same rounding.
struct AdamConsts {
  float lr, b1, b2, eps, wd;
  float omb1, omb2;     // 1 - beta, taken in DOUBLE on the host and then rounded, as torch does with its python floats:
                        // (float)(1.0 - 0.999) = 0.001f, whereas 1.0f - 0.999f = 0.00099998713f (1.3e-5 off in every
                        // exp_avg_sq increment)
};
__host__ __device__ inline AdamConsts make_adam_consts(float lr, float b1, float b2, float eps, float wd) {
  return AdamConsts{lr, b1, b2, eps, wd, (float)(1.0 - (double)b1), (float)(1.0 - (double)b2)};
}
__device__ __forceinline__ void adam_update(float &pi, float gi, float &mi, float &vi, const AdamConsts &c, float bc1,
                                            float bc2_sqrt) {
  if (c.wd != 0.f) gi = gi + c.wd * pi;
  mi = mi + (gi - mi) * c.omb1;
  vi = vi * c.b2 + c.omb2 * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + c.eps;
  pi = pi - (c.lr / bc1) * (mi / denom);
}
// bias corrections in double, as torch does with python floats
__device__ __forceinline__ void adam_bias_corrections(int step, float b1, float b2, float &bc1, float &bc2_sqrt) {
  bc1 = (float)(1.0 - pow((double)b1, (double)step));
  bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
}

__device__ __forceinline__ float wave_sum(float v) {
  // fixed butterfly order -> identical result on every lane and every run
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// max over the wave of a value that is UNIFORM within each group of LPR lanes (a row's entry count), every lane active: one
// v_readlane per group and scalar maxima -- the butterfly above is six LDS swizzles and as many VALU maxima, per wave, in
// launches that are bound by instruction issue (profiles/r06_gated_occupancy.txt)
template <int LPR>
__device__ __forceinline__ int group_uniform_max_i32(int v) {
  if constexpr (LPR < 16) return wave_max_i32(v);       // (many small groups: the butterfly is shorter)
  int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int g = 1; g < 64 / LPR; ++g) {
    const int o = __builtin_amdgcn_readlane(v, g * LPR);
    m = o > m ? o : m;
  }
  return m;
}


// lanes per output element of gemm_reduce_slabs_kernel (block = 32 elements x lanes): many slabs of a SMALL output (a
// 64 x 64 weight gradient over 256 slabs) want 32 lanes -- the launch is a latency chain of splits / lanes loads --, a large
// output with few slabs (768 x 772 over 25) has parallelism enough and would only idle the extra lanes
inline int reduce_lanes(int splits, int64_t mn) {
  int lanes = 8;
  while (lanes < 32 && splits >= 8 * lanes && mn * lanes <= (int64_t)1 << 20) lanes *= 2;
  return lanes;
}

}  // namespace chaorec
