// Dense fp32 GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32): exact fp32 products and a
// k-ascending fp32 fmaf chain per output element, bit-identical to oracle_gemm().
//
// Replaces nn.Linear forward/backward on the modality features and MMGCN's per-layer Linears
// (Model/FREEDOM.py:59-60,209,212; Model/MMGCN.py:40,97,102-131; BasicGCN.py:40).
//
// Block tile 128 x 64 x 16, 4 waves, each wave a 32 x 64 strip (two 32x32 accumulators).
// Operand tiles are staged k-major in LDS ([k][row] with a +1 pad), so the MFMA fragment read
// "lane (r, h) -> element [k = 2s + h][r]" walks consecutive banks.
#include "common.h"

namespace chaorec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 64, BK = 16;

__global__ __launch_bounds__(256) void gemm_f32_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
    const float *__restrict__ bias, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
    int64_t ldc, int transA, int transB, int accumulate, int act) {
  __shared__ float As[BK][BM + 1];
  __shared__ float Bs[BK][BN + 1];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  for (int64_t k0 = 0; k0 < K; k0 += BK) {
    // stage A tile (BM x BK) -> As[k][m]; keep the stored-contiguous dimension on adjacent threads
    if (!transA) {
#pragma unroll
      for (int p = 0; p < (BM * BK) / 256; ++p) {
        const int kk = t & (BK - 1), mm = (t >> 4) + p * 16;
        const int64_t gm = m0 + mm, gk = k0 + kk;
        As[kk][mm] = (gm < M && gk < K) ? A[gm * lda + gk] : 0.f;
      }
    } else {
#pragma unroll
      for (int p = 0; p < (BM * BK) / 256; ++p) {
        const int mm = t & (BM - 1), kk = (t >> 7) + p * 2;
        const int64_t gm = m0 + mm, gk = k0 + kk;
        As[kk][mm] = (gm < M && gk < K) ? A[gk * lda + gm] : 0.f;
      }
    }
    // stage B tile (BK x BN) -> Bs[k][n]
    if (!transB) {
#pragma unroll
      for (int p = 0; p < (BN * BK) / 256; ++p) {
        const int nn = t & (BN - 1), kk = (t >> 6) + p * 4;
        const int64_t gn = n0 + nn, gk = k0 + kk;
        Bs[kk][nn] = (gn < N && gk < K) ? B[gk * ldb + gn] : 0.f;
      }
    } else {
#pragma unroll
      for (int p = 0; p < (BN * BK) / 256; ++p) {
        const int kk = t & (BK - 1), nn = (t >> 4) + p * 16;
        const int64_t gn = n0 + nn, gk = k0 + kk;
        Bs[kk][nn] = (gn < N && gk < K) ? B[gn * ldb + gk] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float a = As[2 * s + h][wave * 32 + r];
      const float b0 = Bs[2 * s + h][r];
      const float b1 = Bs[2 * s + h][32 + r];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int64_t gn = n0 + half * 32 + r;
    if (gn >= N) continue;
    const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const int64_t gm = m0 + wave * 32 + row;
      if (gm >= M) continue;
      float v = half ? acc1[reg] : acc0[reg];
      if (bias) v = v + bv;
      if (accumulate) v = C[gm * ldc + gn] + v;
      if (act == 1) v = v > 0.f ? v : v * 0.01f;
      C[gm * ldc + gn] = v;
    }
  }
}

__global__ __launch_bounds__(256) void adam_step_kernel(float *__restrict__ p,
                                                        const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v,
                                                        int64_t n, float lr, float b1, float b2,
                                                        float eps, float wd, int step_host,
                                                        const int32_t *__restrict__ step_dev) {
  // bias corrections in double, as torch does with python floats; once per block.  The step count may
  // live in device memory so a captured hipGraph replays with the right correction every time.
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const int step = step_dev ? step_dev[0] : step_host;
    bc[0] = (float)(1.0 - pow((double)b1, (double)step));
    bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)step));
  }
  __syncthreads();
  const float bc1 = bc[0], bc2_sqrt = bc[1];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    // torch._single_tensor_adam: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int chaorec_gemm_f32(const float *A, const float *B, float *C, const float *bias,
                                int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                int64_t ldc, int32_t transA, int32_t transB, int32_t accumulate,
                                int32_t act, void *stream) {
  if (!A || !B || !C) return fail(CHAOREC_E_INVALID, "gemm: NULL argument");
  if (M < 0 || N < 0 || K < 0) return fail(CHAOREC_E_INVALID, "gemm: negative size");
  if (act < 0 || act > 1) return fail(CHAOREC_E_INVALID, "gemm: act %d", act);
  if (M == 0 || N == 0) return CHAOREC_OK;
  const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, bias, M, N, K,
                     lda, ldb, ldc, transA, transB, accumulate, act);
  return check_launch("gemm_f32_kernel");
}

extern "C" int chaorec_adam_step_f32(float *param, const float *grad, float *exp_avg,
                                     float *exp_avg_sq, int64_t n, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, int32_t step,
                                     const int32_t *step_dev, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(CHAOREC_E_INVALID, "adam: NULL argument");
  if (n < 0 || (!step_dev && step < 1)) return fail(CHAOREC_E_INVALID, "adam: n=%lld step=%d", (long long)n, step);
  if (n == 0) return CHAOREC_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048 * 4) blocks = 2048 * 4;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, step_dev);
  return check_launch("adam_step_kernel");
}

extern "C" int chaorec_abi_version(void) { return CHAOREC_ABI_VERSION; }
extern "C" const char *chaorec_last_error(void) { return chaorec::err_buf(); }
