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
  if (wslot >= n_groups) return;  // wave-uniform
  // longest-first schedule: the host sorts the NG-row groups by their heaviest row so the long
  // rows start at t=0 instead of stretching the tail (the output location is unchanged)
  const int64_t wave = group_order ? (int64_t)group_order[wslot] : wslot;
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
    constexpr int UH = 8;
    constexpr int HALF = NG * UH;     // entries per half-step
    constexpr int HPB = 64 / HALF;    // half-steps per 64-entry (col,val) block
    __shared__ float4 red_all[4][HALF * LPR];
    float4 *red = red_all[threadIdx.x >> 6];
    unsigned long long lm = __ballot(is_long && li == 0);
    while (lm) {
      const int gl = (int)(__builtin_ctzll(lm) / LPR);
      lm &= lm - 1;
      const int owner = gl * LPR;
      const int n = __shfl(deg, owner, 64);
      const int64_t le0 = ((int64_t)__shfl((int)(e0 >> 32), owner, 64) << 32) |
                          (int64_t)(unsigned int)__shfl((int)(e0 & 0xffffffffll), owner, 64);
      const int nh = (n + HALF - 1) / HALF;
      // ordered sum, one (or two) FEATURES per lane: entry e of the LDS tile is D consecutive
      // floats, lane l adds floats [l*FPL, l*FPL+FPL) -- one v_add per entry instead of four
      constexpr int FPL = LPR * 4 / 64;
      float a[FPL];
#pragma unroll
      for (int f = 0; f < FPL; ++f) a[f] = 0.f;
      const float *redf = reinterpret_cast<const float *>(red);
      int bcur = 0;
      int c0 = 0, c1 = 0;
      float v0 = 0.f, v1 = 0.f;
      if (lane < n) {
        c0 = col[le0 + lane];
        v0 = val[le0 + lane];
      }
      if (64 + lane < n) {
        c1 = col[le0 + 64 + lane];
        v1 = val[le0 + 64 + lane];
      }
      float4 xa[UH], xb[UH];
      float va[UH], vb[UH];
      auto gather = [&](int h, float4(&xv)[UH], float(&vv)[UH]) {
        const int blk = h / HPB;
        const int off = (h % HPB) * HALF;
        const int cs = (blk == bcur) ? c0 : c1;
        const float vs = (blk == bcur) ? v0 : v1;
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int idx = off + u * NG + sub;
          const int cj = __shfl(cs, idx, 64);
          vv[u] = __shfl(vs, idx, 64);
          xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (h * HALF + u * NG + sub < n && li < D4) xv[u] = x4[(size_t)cj * (size_t)D4 + li];
        }
      };
      auto reduce = [&](int h, float4(&xv)[UH], float(&vv)[UH]) {
#pragma unroll
        for (int u = 0; u < UH; ++u) red[(u * NG + sub) * LPR + li] = mul_rn4(vv[u], xv[u]);
        __builtin_amdgcn_wave_barrier();
        const int cntv = min(HALF, n - h * HALF);
#pragma unroll 8
        for (int e = 0; e < cntv; ++e) {
#pragma unroll
          for (int f = 0; f < FPL; ++f) a[f] = add_rn(a[f], redf[e * (LPR * 4) + lane * FPL + f]);
        }
        __builtin_amdgcn_wave_barrier();
      };
      gather(0, xa, va);
      if (1 < nh) gather(1, xb, vb);
      for (int h = 0; h < nh; h += 2) {
        reduce(h, xa, va);
        if (h + 2 < nh) gather(h + 2, xa, va);
        if (h + 1 < nh) reduce(h + 1, xb, vb);
        if (h + 3 < nh) gather(h + 3, xb, vb);
        if ((h + 2) / HPB > bcur) {  // every half of block bcur has been gathered: rotate
          ++bcur;
          c0 = c1;
          v0 = v1;
          c1 = 0;
          v1 = 0.f;
          const int nb = (bcur + 1) * 64 + lane;
          if (nb < n) {
            c1 = col[le0 + nb];
            v1 = val[le0 + nb];
          }
        }
      }
      // back to the float4-per-lane row layout of the owning group
      float *redw = reinterpret_cast<float *>(red);
#pragma unroll
      for (int f = 0; f < FPL; ++f) redw[lane * FPL + f] = a[f];
      __builtin_amdgcn_wave_barrier();
      const float4 arow = red[li];
      __builtin_amdgcn_wave_barrier();
      if (sub == gl) sum[0] = arow;
    }
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
