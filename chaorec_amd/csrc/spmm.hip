// CSR SpMM over the normalised user-item Laplacian:  y = alpha * (A x) [+ beta z], fp32.
//
// Replaces the reference's propagate (index_select + norm*x_j + scatter_add,
// Model/LightGCN.py:40-43, BasicGCN.py:48-53) and torch.sparse.mm (Model/FREEDOM.py:168,174).
//
// HBM-bound gather: 2 flop per 4 gathered bytes, so no MFMA here.  Layout choices:
//   * one LPR-lane group per destination row, each lane owning one float4 of the row
//     (D=64 -> 16 lanes x 16 B = one 256-B row per group, 4 rows per wave64), so every
//     gathered source row is read as whole 128-B lines with dwordx4 loads;
//   * the group loads LPR (col,val) pairs with ONE coalesced load and broadcasts them with
//     lane shuffles instead of every lane re-reading the same index;
//   * 4 source rows are in flight per group before the first add (4 x 4 KiB per wave),
//     the adds themselves stay in CSR order with separately rounded product and sum, so the
//     result is bit-identical to a sequential scatter_add_ in the reference's edge order.
#include "common.h"

namespace chaorec {

template <int LPR, int CPL>
__global__ __launch_bounds__(256) void spmm_csr_ordered_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, float *__restrict__ y,
    int64_t n_rows, int D4, float alpha, const float *__restrict__ z, float beta,
    float *acc, const float *__restrict__ acc_init, float acc_w) {
  constexpr int RPW = kWave / LPR;  // destination rows per wave
  constexpr int UNR = 4;            // gathered rows in flight per group
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;
  const int li = lane % LPR;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t r = wave * RPW + sub;
  const bool row_ok = r < n_rows;

  int64_t e0 = 0, e1 = 0;
  if (row_ok) {
    e0 = rowptr[r];
    e1 = rowptr[r + 1];
  }
  const int deg = (int)(e1 - e0);
  const int dmax = wave_max_i32(deg);  // wave-uniform trip counts keep every shuffle fully active

  float4 sum[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) sum[q] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);

  for (int base = 0; base < dmax; base += LPR) {
    // one coalesced (col,val) load per group, broadcast below
    int c = 0;
    float v = 0.f;
    if (base + li < deg) {
      c = col[e0 + base + li];
      v = val[e0 + base + li];
    }
    const int n = min(LPR, deg - base);          // this group's entries in the block (may be <= 0)
    const int nmax = min(LPR, dmax - base);      // wave-uniform
    for (int j = 0; j < nmax; j += UNR) {
      int cj[UNR];
      float vj[UNR];
      bool p[UNR];
      float4 xv[UNR][CPL];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int src = sub * LPR + ((j + u) & (LPR - 1));
        cj[u] = __shfl(c, src, 64);
        vj[u] = __shfl(v, src, 64);
        p[u] = (j + u) < n;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int chunk = li + q * LPR;
          xv[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p[u] && chunk < D4) xv[u][q] = x4[(size_t)cj[u] * (size_t)D4 + chunk];
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (p[u]) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) sum[q] = add_rn4(sum[q], mul_rn4(vj[u], xv[u][q]));
        }
      }
    }
  }

  if (!row_ok) return;
  const float4 *z4 = reinterpret_cast<const float4 *>(z);
  const float4 *init4 = reinterpret_cast<const float4 *>(acc_init);
  float4 *y4 = reinterpret_cast<float4 *>(y);
  float4 *acc4 = reinterpret_cast<float4 *>(acc);
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int chunk = li + q * LPR;
    if (chunk >= D4) continue;
    const size_t o = (size_t)r * (size_t)D4 + chunk;
    float4 s = mul_rn4(alpha, sum[q]);
    if (z) s = add_rn4(s, mul_rn4(beta, z4[o]));
    if (y) y4[o] = s;
    if (acc) {
      const float4 a0 = acc_init ? mul_rn4(acc_w, init4[o]) : acc4[o];
      acc4[o] = add_rn4(a0, mul_rn4(acc_w, s));
    }
  }
}

template <int LPR, int CPL>
static int launch_spmm(const int64_t *rowptr, const int32_t *col, const float *val, const float *x,
                       float *y, int64_t n_rows, int D4, float alpha, const float *z, float beta,
                       float *acc, const float *acc_init, float acc_w, hipStream_t st) {
  constexpr int RPW = kWave / LPR;
  const int64_t waves = (n_rows + RPW - 1) / RPW;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return fail(CHAOREC_E_INVALID, "spmm: grid too large");
  hipLaunchKernelGGL((spmm_csr_ordered_kernel<LPR, CPL>), dim3((unsigned)blocks), dim3(256), 0, st,
                     rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w);
  return check_launch("spmm_csr_ordered_kernel");
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int chaorec_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                    const float *x, float *y, int64_t n_rows, int64_t n_cols,
                                    int32_t D, float alpha, const float *z, float beta, float *acc,
                                    const float *acc_init, float acc_w, int32_t mode, void *stream) {
  if (!rowptr || !x || (!y && !acc)) return fail(CHAOREC_E_INVALID, "spmm: NULL rowptr/x or no output");
  if (n_rows < 0 || n_cols < 0) return fail(CHAOREC_E_INVALID, "spmm: negative size");
  if (D < 4 || D > 1024 || (D & 3)) return fail(CHAOREC_E_INVALID, "spmm: D=%d must be a multiple of 4 in [4,1024]", D);
  if (mode != 0) return fail(CHAOREC_E_INVALID, "spmm: unknown mode %d", mode);
  if (acc_init && !acc) return fail(CHAOREC_E_INVALID, "spmm: acc_init without acc");
  if (n_rows == 0) return CHAOREC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int D4 = D / 4;
#define CHAOREC_SPMM_ARGS rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, st
  if (D4 <= 1) return launch_spmm<1, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 2) return launch_spmm<2, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 4) return launch_spmm<4, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 8) return launch_spmm<8, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 16) return launch_spmm<16, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 32) return launch_spmm<32, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 64) return launch_spmm<64, 1>(CHAOREC_SPMM_ARGS);
  if (D4 <= 128) return launch_spmm<64, 2>(CHAOREC_SPMM_ARGS);
  if (D4 <= 192) return launch_spmm<64, 3>(CHAOREC_SPMM_ARGS);
  return launch_spmm<64, 4>(CHAOREC_SPMM_ARGS);
#undef CHAOREC_SPMM_ARGS
}
