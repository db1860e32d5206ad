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

__device__ __forceinline__ float4 shfl4(float4 v, int src) {
  return make_float4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
}

#ifndef CHAOREC_SPMM_UNR
#define CHAOREC_SPMM_UNR 8
#endif
#ifndef CHAOREC_SPMM_LONG_T
#define CHAOREC_SPMM_LONG_T (4 * CHAOREC_SPMM_UNR)
#endif
#ifndef CHAOREC_SPMM_MINW
#define CHAOREC_SPMM_MINW 1
#endif

template <int LPR, int CPL>
__global__ __launch_bounds__(256, CHAOREC_SPMM_MINW) void spmm_csr_ordered_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const float *__restrict__ x, float *__restrict__ y,
    int64_t n_rows, int D4, float alpha, const float *__restrict__ z, float beta,
    float *acc, const float *__restrict__ acc_init, float acc_w,
    const int32_t *__restrict__ group_order, int64_t n_groups) {
  constexpr int NG = kWave / LPR;   // lane groups = destination rows per wave
  constexpr int UNR = CHAOREC_SPMM_UNR;  // gathered rows in flight per group (short-row phase)
  constexpr int UNR2 = (kWave / NG) < 16 ? (kWave / NG) : 16;  // per group in the long-row phase
  constexpr int LONG_T = CHAOREC_SPMM_LONG_T;  // rows above this are walked by the whole wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;
  const int li = lane % LPR;
  const int64_t wslot = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // longest-first schedule: the host sorts the NG-row groups by their heaviest row so the long
  // rows start at t=0 instead of stretching the tail (the output location is unchanged).
  // Surplus waves of the last block stay (they take part in the block barriers below) with no rows.
  const int64_t wave = wslot < n_groups ? (group_order ? (int64_t)group_order[wslot] : wslot) : n_groups;
  const int64_t r = wave * NG + sub;
  const bool row_ok = r < n_rows;

  int64_t e0 = 0, e1 = 0;
  if (row_ok) {
    e0 = rowptr[r];
    e1 = rowptr[r + 1];
  }
  const int deg = (int)(e1 - e0);
  const bool is_long = (NG > 1) && deg > LONG_T;
  const int deg1 = is_long ? 0 : deg;
  const int dmax = wave_max_i32(deg1);  // wave-uniform trip counts keep every shuffle fully active

  float4 sum[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) sum[q] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);

  // ---- phase 1: every group walks its own (short) row, UNR source rows in flight -----------
  for (int base = 0; base < dmax; base += LPR) {
    int c = 0;
    float v = 0.f;
    if (base + li < deg1) {  // one coalesced (col,val) load per group, broadcast below
      c = col[e0 + base + li];
      v = val[e0 + base + li];
    }
    const int n = min(LPR, deg1 - base);
    const int nmax = min(LPR, dmax - base);
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

  // ---- phase 2: long rows, one at a time, gathered by ALL groups ---------------------------
  // The sum stays the sequential CSR-order sum: the groups only share the LOADS.  A long row's
  // time is set by how many source rows one wave keeps in flight, so for D = 64 / 128 the walk is
  // software-pipelined: two register buffers of HALF = 8*NG source rows ping-pong (one in flight
  // while the other is reduced), (col,val) blocks of 64 entries are fetched a block ahead, and
  // the ordered reduction goes through an 8 KiB LDS tile (every lane reads entry e of its own
  // float4 column, a broadcast read) instead of NG x 4 lane shuffles per entry.
  if constexpr (NG == 2 || NG == 4) {
    // Block-cooperative, still sequential: the 4 waves of the workgroup take the row's 32-entry chunks
    // round-robin.  Each wave gathers its chunk and parks the products in its own 8 KiB LDS tile while
    // the previous chunks are being summed; the running sum is then handed from chunk to chunk through
    // LDS (`carry`, ordered by the `seq` ticket), so only the adds -- ~8 cycles per entry -- are serial,
    // not the memory latency.  The additions happen in entry order: bit-identical to one lane walking
    // the row alone.
    constexpr int UH = 8;
    constexpr int HALF = NG * UH;       // entries per chunk (32 for D=64, 16 for D=128)
    constexpr int FPL = LPR * 4 / 64;   // features per lane in the ordered sum (1 or 2)
    constexpr int MAXL = 4 * NG;        // rows per block
    __shared__ float4 tile_all[4][HALF * LPR];
    __shared__ float carry[64 * FPL];
    __shared__ float4 long_sum[MAXL][LPR];
    __shared__ long long long_e0[MAXL];
    __shared__ int long_n[MAXL];
    __shared__ int n_long_s, seq_s;
    const int wv = threadIdx.x >> 6;
    float4 *tile = tile_all[wv];
    const float *tilef = reinterpret_cast<const float *>(tile);
    if (threadIdx.x == 0) n_long_s = 0;
    __syncthreads();
    int my_slot = -1;
    if (is_long && li == 0) {
      my_slot = atomicAdd(&n_long_s, 1);
      long_e0[my_slot] = e0;
      long_n[my_slot] = deg;
    }
    my_slot = __shfl(my_slot, sub * LPR, 64);
    __syncthreads();
    const int nl = n_long_s;  // block-uniform
    for (int t = 0; t < nl; ++t) {
      const int n = long_n[t];
      const int64_t le0 = long_e0[t];
      const int nchunks = (n + HALF - 1) / HALF;
      if (threadIdx.x == 0) seq_s = 0;
      __syncthreads();
      float4 xv[UH];
      float vv[UH];
      auto gather = [&](int k) {
        int c = 0;
        float v = 0.f;
        if (lane < HALF && k * HALF + lane < n) {
          c = col[le0 + k * HALF + lane];
          v = val[le0 + k * HALF + lane];
        }
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int idx = u * NG + sub;
          const int cj = __shfl(c, idx, 64);
          vv[u] = __shfl(v, idx, 64);
          xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k * HALF + idx < n && li < D4) xv[u] = x4[(size_t)cj * (size_t)D4 + li];
        }
      };
      if (wv < nchunks) gather(wv);
      for (int k = wv; k < nchunks; k += 4) {
#pragma unroll
        for (int u = 0; u < UH; ++u) tile[(u * NG + sub) * LPR + li] = mul_rn4(vv[u], xv[u]);
        if (k + 4 < nchunks) gather(k + 4);  // next chunk's loads fly while we wait for the carry
        while (__hip_atomic_load(&seq_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != k)
          __builtin_amdgcn_s_sleep(1);
        float a[FPL];
#pragma unroll
        for (int f = 0; f < FPL; ++f) a[f] = k == 0 ? 0.f : carry[lane * FPL + f];
        const int cntv = min(HALF, n - k * HALF);
#pragma unroll 8
        for (int e = 0; e < cntv; ++e) {
#pragma unroll
          for (int f = 0; f < FPL; ++f) a[f] = add_rn(a[f], tilef[e * (LPR * 4) + lane * FPL + f]);
        }
        float *dst = (k == nchunks - 1) ? reinterpret_cast<float *>(long_sum[t]) : carry;
#pragma unroll
        for (int f = 0; f < FPL; ++f) dst[lane * FPL + f] = a[f];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&seq_s, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
    }
    if (my_slot >= 0) sum[0] = long_sum[my_slot][li];
  } else if constexpr (NG > 1) {
    unsigned long long lm = __ballot(is_long && li == 0);
    while (lm) {
      const int gl = (int)(__builtin_ctzll(lm) / LPR);
      lm &= lm - 1;
      const int owner = gl * LPR;
      const int n = __shfl(deg, owner, 64);
      const int64_t le0 = ((int64_t)__shfl((int)(e0 >> 32), owner, 64) << 32) |
                          (int64_t)(unsigned int)__shfl((int)(e0 & 0xffffffffll), owner, 64);
      float4 a[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int base = 0; base < n; base += NG * UNR2) {
        int c = 0;
        float v = 0.f;
        if (lane < NG * UNR2 && base + lane < n) {
          c = col[le0 + base + lane];
          v = val[le0 + base + lane];
        }
        float vv[UNR2];
        float4 xv[UNR2][CPL];
#pragma unroll
        for (int u = 0; u < UNR2; ++u) {
          const int idx = u * NG + sub;
          const int cj = __shfl(c, idx, 64);
          vv[u] = __shfl(v, idx, 64);
          const bool p = base + idx < n;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const int chunk = li + q * LPR;
            xv[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p && chunk < D4) xv[u][q] = x4[(size_t)cj * (size_t)D4 + chunk];
          }
        }
#pragma unroll
        for (int u = 0; u < UNR2; ++u) {
          float4 t[CPL];
#pragma unroll
          for (int q = 0; q < CPL; ++q) t[q] = mul_rn4(vv[u], xv[u][q]);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const bool p = base + u * NG + g < n;  // wave-uniform
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
              const float4 tg = shfl4(t[q], g * LPR + li);
              if (p) a[q] = add_rn4(a[q], tg);
            }
          }
        }
      }
      if (sub == gl) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) sum[q] = a[q];
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
                       float *acc, const float *acc_init, float acc_w, const int32_t *group_order,
                       hipStream_t st) {
  constexpr int RPW = kWave / LPR;
  const int64_t waves = (n_rows + RPW - 1) / RPW;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return fail(CHAOREC_E_INVALID, "spmm: grid too large");
  hipLaunchKernelGGL((spmm_csr_ordered_kernel<LPR, CPL>), dim3((unsigned)blocks), dim3(256), 0, st,
                     rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, group_order,
                     waves);
  return check_launch("spmm_csr_ordered_kernel");
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int chaorec_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                    const float *x, float *y, int64_t n_rows, int64_t n_cols,
                                    int32_t D, float alpha, const float *z, float beta, float *acc,
                                    const float *acc_init, float acc_w, const int32_t *group_order,
                                    int32_t mode, void *stream) {
  if (!rowptr || !x || (!y && !acc)) return fail(CHAOREC_E_INVALID, "spmm: NULL rowptr/x or no output");
  if (n_rows < 0 || n_cols < 0) return fail(CHAOREC_E_INVALID, "spmm: negative size");
  if (D < 4 || D > 1024 || (D & 3)) return fail(CHAOREC_E_INVALID, "spmm: D=%d must be a multiple of 4 in [4,1024]", D);
  if (mode != 0) return fail(CHAOREC_E_INVALID, "spmm: unknown mode %d", mode);
  if (acc_init && !acc) return fail(CHAOREC_E_INVALID, "spmm: acc_init without acc");
  if (n_rows == 0) return CHAOREC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int D4 = D / 4;
#define CHAOREC_SPMM_ARGS rowptr, col, val, x, y, n_rows, D4, alpha, z, beta, acc, acc_init, acc_w, group_order, st
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

extern "C" int chaorec_spmm_rows_per_wave(int32_t D) {
  if (D < 4 || (D & 3)) return 0;
  int lpr = 1;
  while (lpr < D / 4 && lpr < 64) lpr <<= 1;
  return 64 / lpr;
}
