// Dense fp32 GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32): exact fp32 products and a
// k-ascending fp32 fmaf chain per output element, bit-identical to oracle_gemm().
//
// Replaces nn.Linear forward/backward on the modality features and MMGCN's per-layer Linears
// (Model/FREEDOM.py:59-60,209,212; Model/MMGCN.py:40,97,102-131; BasicGCN.py:40).
//
#include "common.h"

namespace chaorec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Block tile BM x BN x BK = 128 x {64,128} x 16, 4 waves.  BN=128: waves 2x2, each a 64x64 patch (2x2 MFMA
// 32x32 accumulators); BN=64: waves 4x1, each 32x64 (1x2).  Operand tiles are staged k-major in LDS
// ([k][row], rows padded by 4 floats), double-buffered: the next tile is fetched global -> registers (float4
// along whichever dimension is contiguous in memory) while the current one feeds the MFMAs.  A k-step of the
// f32 MFMA consumes k = 2s (lanes 0-31) and 2s+1 (lanes 32-63): the per-element sum is the k-ascending fmaf
// chain of oracle_gemm_f32 -- also across split-K slabs only up to the order of the final slab sum.
constexpr int BK = 16, PAD = 4;
// Short reductions (K of a 64-wide Linear) over many rows are latency-bound with 16-deep tiles: a workgroup would go
// global -> registers -> LDS -> barrier four times for 2 us of MFMA work.  The <.., BKT = 64, NBUF = 1> form stages a
// 64-deep tile per barrier pair (every load of the tile in flight at once, one LDS buffer: 49 KB for 128 x 64).

// <BM, BN> in {(128,128): waves 2x2 of 64x64; (128,64): waves 4x1 of 32x64; (64,128): waves 2x2 of 32x64; (64,64):
// waves 2x2 of 32x32 -- the 64-row tiles for outputs with <= 64 rows (weight gradients of 64-wide layers), where a
// 128-row tile would be half padding}
template <int BM, int BN, int BKT = BK, int NBUF = 2>
__global__ __launch_bounds__(256) void gemm_f32_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
    const float *__restrict__ bias, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
    int64_t ldc, int transA, int transB, int accumulate, int act, int64_t k_per_split,
    float *__restrict__ slabs) {
  constexpr int WM = (BM == 128 && BN == 128) ? 2 : 1;   // 32-row accumulator blocks per wave
  constexpr int WN = (BM == 64 && BN == 64) ? 1 : 2;     // 32-col accumulator blocks per wave
  constexpr int BK = BKT;                 // (shadows the default tile depth inside this kernel)
  // k-rows of a deep tile are skewed by 4 floats per 16 rows (SK): the transposing stash below writes, per wave,
  // 4 tile rows x 16 k-quads, and k-quads 16 rows apart would otherwise land in the same LDS banks
  constexpr int SKW = BK > 16 ? 12 : 0;
#define SK(k) (4 * (((k) >> 4) & 3))
  constexpr int KQ = BK / 4;              // float4 per tile row along k
  __shared__ float As[NBUF][BK][BM + PAD + SKW];
  __shared__ float Bs[NBUF][BK][BN + PAD + SKW];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  // waves: (128,64) 4x1 of 32x64; (64,64) 2x2 of 32x32; (128,128) 2x2 of 64x64; (64,128) 2x2 of 32x64
  const int wrow = (BM == 128 && BN == 64) ? wave * 32 : (wave >> 1) * (BM / 2);
  const int wcol = (BM == 128 && BN == 64) ? 0 : (wave & 1) * (BN / 2);
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  const int64_t kb = (int64_t)blockIdx.z * k_per_split;
  const int64_t ke = min(K, kb + k_per_split);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // which float4 of the tile this thread fetches; the vector runs along the memory-contiguous dimension
  const bool a_vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool b_vec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  constexpr int A_V = BM * BK / 4 / 256;  // float4 per thread per A tile (2)
  constexpr int B_V = BN * BK / 4 / 256;  // (2 for BN=128, 1 for BN=64)
  float4 ra[A_V], rb[B_V];

  // Interior tiles (the whole BM x BK / BN x BK window inside the matrix, 16-B aligned rows) take a branch with
  // unconditional float4 loads; only edge tiles run the element-guarded path.  The test is block-uniform.
  const bool a_rows_in = m0 + BM <= M, b_rows_in = n0 + BN <= N;
  auto fetch_a = [&](int64_t k0) {
    if (a_vec && a_rows_in && k0 + BK <= ke) {
#pragma unroll
      for (int p = 0; p < A_V; ++p) {
        const int v = t + p * 256;
        if (!transA) {
          const int mm = v / KQ, k4 = (v % KQ) * 4;
          ra[p] = *reinterpret_cast<const float4 *>(A + (m0 + mm) * lda + k0 + k4);
        } else {
          const int kk = v / (BM / 4), m4 = (v % (BM / 4)) * 4;
          ra[p] = *reinterpret_cast<const float4 *>(A + (k0 + kk) * lda + m0 + m4);
        }
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < A_V; ++p) {
      const int v = t + p * 256;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!transA) {  // A[m][k]: vector along k
        const int mm = v / KQ, k4 = (v % KQ) * 4;
        const int64_t gm = m0 + mm, gk = k0 + k4;
        if (gm < M) {
          const float *src = A + gm * lda + gk;
          if (a_vec && gk + 3 < ke) x = *reinterpret_cast<const float4 *>(src);
          else {
            if (gk + 0 < ke) x.x = src[0];
            if (gk + 1 < ke) x.y = src[1];
            if (gk + 2 < ke) x.z = src[2];
            if (gk + 3 < ke) x.w = src[3];
          }
        }
      } else {        // A[k][m]: vector along m
        const int kk = v / (BM / 4), m4 = (v % (BM / 4)) * 4;
        const int64_t gk = k0 + kk, gm = m0 + m4;
        if (gk < ke) {
          const float *src = A + gk * lda + gm;
          if (a_vec && gm + 3 < M) x = *reinterpret_cast<const float4 *>(src);
          else {
            if (gm + 0 < M) x.x = src[0];
            if (gm + 1 < M) x.y = src[1];
            if (gm + 2 < M) x.z = src[2];
            if (gm + 3 < M) x.w = src[3];
          }
        }
      }
      ra[p] = x;
    }
  };
  auto fetch_b = [&](int64_t k0) {
    if (b_vec && b_rows_in && k0 + BK <= ke) {
#pragma unroll
      for (int p = 0; p < B_V; ++p) {
        const int v = t + p * 256;
        if (transB) {
          const int nn = v / KQ, k4 = (v % KQ) * 4;
          rb[p] = *reinterpret_cast<const float4 *>(B + (n0 + nn) * ldb + k0 + k4);
        } else {
          const int kk = v / (BN / 4), n4 = (v % (BN / 4)) * 4;
          rb[p] = *reinterpret_cast<const float4 *>(B + (k0 + kk) * ldb + n0 + n4);
        }
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < B_V; ++p) {
      const int v = t + p * 256;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (transB) {   // B stored [n][k]: vector along k
        const int nn = v / KQ, k4 = (v % KQ) * 4;
        const int64_t gn = n0 + nn, gk = k0 + k4;
        if (gn < N) {
          const float *src = B + gn * ldb + gk;
          if (b_vec && gk + 3 < ke) x = *reinterpret_cast<const float4 *>(src);
          else {
            if (gk + 0 < ke) x.x = src[0];
            if (gk + 1 < ke) x.y = src[1];
            if (gk + 2 < ke) x.z = src[2];
            if (gk + 3 < ke) x.w = src[3];
          }
        }
      } else {        // B stored [k][n]: vector along n
        const int kk = v / (BN / 4), n4 = (v % (BN / 4)) * 4;
        const int64_t gk = k0 + kk, gn = n0 + n4;
        if (gk < ke) {
          const float *src = B + gk * ldb + gn;
          if (b_vec && gn + 3 < N) x = *reinterpret_cast<const float4 *>(src);
          else {
            if (gn + 0 < N) x.x = src[0];
            if (gn + 1 < N) x.y = src[1];
            if (gn + 2 < N) x.z = src[2];
            if (gn + 3 < N) x.w = src[3];
          }
        }
      }
      rb[p] = x;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int p = 0; p < A_V; ++p) {
      const int v = t + p * 256;
      if (!transA) {
        const int mm = v / KQ, k4 = (v % KQ) * 4;
        float *d = &As[buf][k4][mm + SK(k4)];           // k4 .. k4+3 share a skew (k4 is a multiple of 4)
        d[0 * (BM + PAD + SKW)] = ra[p].x;
        d[1 * (BM + PAD + SKW)] = ra[p].y;
        d[2 * (BM + PAD + SKW)] = ra[p].z;
        d[3 * (BM + PAD + SKW)] = ra[p].w;
      } else {
        const int kk = v / (BM / 4), m4 = (v % (BM / 4)) * 4;
        *reinterpret_cast<float4 *>(&As[buf][kk][m4 + SK(kk)]) = ra[p];
      }
    }
#pragma unroll
    for (int p = 0; p < B_V; ++p) {
      const int v = t + p * 256;
      if (transB) {
        const int nn = v / KQ, k4 = (v % KQ) * 4;
        float *d = &Bs[buf][k4][nn + SK(k4)];
        d[0 * (BN + PAD + SKW)] = rb[p].x;
        d[1 * (BN + PAD + SKW)] = rb[p].y;
        d[2 * (BN + PAD + SKW)] = rb[p].z;
        d[3 * (BN + PAD + SKW)] = rb[p].w;
      } else {
        const int kk = v / (BN / 4), n4 = (v % (BN / 4)) * 4;
        *reinterpret_cast<float4 *>(&Bs[buf][kk][n4 + SK(kk)]) = rb[p];
      }
    }
  };

  int buf = 0;
  if (kb < ke) {
    fetch_a(kb);
    fetch_b(kb);
    stash(0);
  }
  __syncthreads();
  for (int64_t k0 = kb; k0 < ke; k0 += BK) {
    const bool more = k0 + BK < ke;
    if (more) {
      fetch_a(k0 + BK);
      fetch_b(k0 + BK);
    }
    // all fragments of the tile first (8 k-pairs x (WM + WN) LDS reads in flight), then the MFMA chain: the reads
    // of a k-pair are not waited on right in front of its MFMAs
    float af[BK / 2][WM], bf[BK / 2][WN];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
#pragma unroll
      for (int i = 0; i < WM; ++i) af[s][i] = As[buf][2 * s + h][SK(2 * s) + wrow + 32 * i + r];
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[s][j] = Bs[buf][2 * s + h][SK(2 * s) + wcol + 32 * j + r];
    }
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
    }
    if (NBUF == 2) {
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    } else {
      __syncthreads();          // every wave is done reading the single buffer
      if (more) stash(0);
      __syncthreads();
    }
  }

  // epilogue: C/D layout col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float *out = slabs ? slabs + (size_t)blockIdx.z * (size_t)M * (size_t)N : C;
  const int64_t ldo = slabs ? N : ldc;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int64_t gn = n0 + wcol + 32 * j + r;
      if (gn >= N) continue;
      const float bv = (bias && !slabs) ? bias[gn] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int64_t gm = m0 + wrow + 32 * i + row;
        if (gm >= M) continue;
        float v = acc[i][j][reg];
        if (!slabs) {
          if (bias) v = v + bv;
          if (accumulate) v = out[gm * ldo + gn] + v;
          if (act == 1) v = v > 0.f ? v : v * 0.01f;
          if (act == 2) v = v > 0.f ? v : v * 0.2f;
        }
        out[gm * ldo + gn] = v;
      }
    }
  }
}

#undef SK

// split-K second pass: C = sum_z slab[z] (+ bias) (+ C) (+ act), in a FIXED order: kRedLanes lanes per output element take
// the slabs z = l, l + kRedLanes, ... in ascending order (each 32-thread group reads 32 consecutive elements of a slab:
// coalesced), then the partial sums are added in lane order.  Deterministic.  The output is small -- e.g. 64 x 64 -- and
// there are up to 256 slabs: with 8 lanes per element a lane walked 32 dependent-issue loads and the launch took 12 us, 27
// times per MMGCN step; 32 lanes (1024-thread blocks) walk 8.
constexpr int kRedLanes = 32;   // at most; the launch picks blockDim = 32 * lanes (reduce_lanes)
__global__ __launch_bounds__(32 * kRedLanes) void gemm_reduce_slabs_kernel(const float *__restrict__ slabs, int splits,
                                                                          float *__restrict__ C, const float *__restrict__ bias,
                                                                          int64_t M, int64_t N, int64_t ldc, int accumulate,
                                                                          int act, float *__restrict__ C2, int64_t ldc2,
                                                                          int64_t row_split) {   // rows m >= row_split -> C2
  __shared__ float part[kRedLanes][32];
  const int lanes = blockDim.x >> 5;
  const int e = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  const size_t mn = (size_t)M * (size_t)N;
  float v = 0.f;
  if (i < (int64_t)mn) {
    int z = l;
    float v1 = 0.f;
    for (; z + lanes < splits; z += 2 * lanes) {     // two loads in flight per lane
      const float a = slabs[(size_t)z * mn + i], b = slabs[(size_t)(z + lanes) * mn + i];
      v = v + a;
      v1 = v1 + b;
    }
    if (z < splits) v = v + slabs[(size_t)z * mn + i];
    v = v + v1;
  }
  part[l][e] = v;
  __syncthreads();
  if (l != 0 || i >= (int64_t)mn) return;
  for (int k = 1; k < lanes; ++k) v = v + part[k][e];
  const int64_t m = i / N, n = i % N;
  if (bias) v = v + bias[n];
  float *out = m >= row_split ? C2 + (m - row_split) * ldc2 + n : C + m * ldc + n;
  if (accumulate) v = *out + v;
  if (act == 1) v = v > 0.f ? v : v * 0.01f;
  if (act == 2) v = v > 0.f ? v : v * 0.2f;
  *out = v;
}

struct GemmPlan {
  int bm;
  int bn;
  int bk;
  int splits;
  int64_t k_per_split;
};

static GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K) {
  GemmPlan p;
  p.bn = N <= 64 ? 64 : 128;
  p.bm = M <= 64 ? 64 : 128;      // (64-row outputs -- weight gradients of 64-wide layers -- would be half padding)
  // tall 64-wide products with a short reduction (Linear 64 -> 64 over all graph nodes): one k-tile per workgroup, so
  // a workgroup's life is one load -> MFMA -> store chain; 128-row tiles give 1.3 workgroups per CU at 44 k rows (two
  // rounds of that chain), 64-row tiles 2.7, all resident at once: 18.3 -> 14.2 us (no gain once 64-row tiles exceed
  // three per CU as well: 60 k rows)
  if (p.bn == 64 && K <= 512 && M > 64 && (M + 63) / 64 <= 768) p.bm = 64;
  const int64_t tiles = ((M + p.bm - 1) / p.bm) * ((N + p.bn - 1) / p.bn);
  p.splits = 1;
  // few output tiles and a long reduction (weight gradients: K = number of graph nodes): split K so that the
  // chip is filled, partial sums go to slabs and are added in a fixed order
  // (from K = 2048 on: the weight gradient of a projection over the 2 B gathered rows of a batch, ops.linear_rows,
  // is [64, 4096] from a reduction of 2048 -- 32 tiles on 256 CUs without the split)
  if (tiles < 128 && K >= 2048) {
    int64_t s = (1024 + tiles - 1) / tiles;     // ~4 workgroups per CU
    const int64_t min_k = K >= 4096 ? 256 : 128;
    if (s > K / min_k) s = K / min_k;           // at least 16 (8 below K = 4096) k-tiles per workgroup
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    p.splits = (int)s;
  }
  // 64-wide outputs with a reduction of at most a few hundred (Linear 64->64 / 128->64 over all graph nodes, and the
  // k-slices of their weight gradients): deep tiles, see the note at BK
  p.bk = (p.bn == 64 && (p.splits > 1 || K <= 512)) ? 64 : BK;
  int64_t per = (K + p.splits - 1) / p.splits;
  per = (per + p.bk - 1) / p.bk * p.bk;
  p.k_per_split = per < p.bk ? p.bk : per;
  p.splits = (int)((K + p.k_per_split - 1) / p.k_per_split);
  if (p.splits < 1) p.splits = 1;
  return p;
}

__global__ __launch_bounds__(256) void adam_step_kernel(float *__restrict__ p,
                                                        const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v,
                                                        int64_t n, float lr, float b1, float b2,
                                                        float eps, float wd, int step_host,
                                                        const int32_t *__restrict__ step_dev, int vec4) {
  // bias corrections in double, as torch does with python floats; once per block.  The step count may
  // live in device memory so a captured hipGraph replays with the right correction every time.
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const int step = step_dev ? step_dev[0] : step_host;
    adam_bias_corrections(step, b1, b2, bc[0], bc[1]);
  }
  __syncthreads();
  const float bc1 = bc[0], bc2_sqrt = bc[1];
  const AdamConsts ac = make_adam_consts(lr, b1, b2, eps, wd);
  auto upd = [&](float &pi, float gi, float &mi, float &vi) { adam_update(pi, gi, mi, vi, ac, bc1, bc2_sqrt); };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec4) {   // 16-B accesses: the update streams 7 arrays, it is pure bandwidth
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += stride) {
      float4 pp = p4[i], mm = m4[i], vv = v4[i];
      const float4 gg = g4[i];
      upd(pp.x, gg.x, mm.x, vv.x);
      upd(pp.y, gg.y, mm.y, vv.y);
      upd(pp.z, gg.z, mm.z, vv.z);
      upd(pp.w, gg.w, mm.w, vv.w);
      m4[i] = mm;
      v4[i] = vv;
      p4[i] = pp;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {   // tail
      float pi = p[i], mi = m[i], vi = v[i];
      upd(pi, g[i], mi, vi);
      m[i] = mi;
      v[i] = vi;
      p[i] = pi;
    }
  } else {
    for (int64_t i = tid; i < n; i += stride) {
      float pi = p[i], mi = m[i], vi = v[i];
      upd(pi, g[i], mi, vi);
      m[i] = mi;
      v[i] = vi;
      p[i] = pi;
    }
  }
}

}  // namespace chaorec

using namespace chaorec;

extern "C" size_t chaorec_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmPlan p = plan_gemm(M, N, K);
  return p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int chaorec_gemm_f32(const float *A, const float *B, float *C, const float *bias,
                                int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                int64_t ldc, int32_t transA, int32_t transB, int32_t accumulate,
                                int32_t act, void *workspace, size_t workspace_bytes, void *stream) {
  if (!A || !B || !C) return fail(CHAOREC_E_INVALID, "gemm: NULL argument");
  if (M < 0 || N < 0 || K < 0) return fail(CHAOREC_E_INVALID, "gemm: negative size");
  if (act < 0 || act > 2) return fail(CHAOREC_E_INVALID, "gemm: act %d", act);
  if (M == 0 || N == 0) return CHAOREC_OK;
  GemmPlan p = plan_gemm(M, N, K > 0 ? K : 1);
  if (K == 0) p.splits = 1;
  const size_t need = p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "gemm: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float *slabs = p.splits > 1 ? (float *)workspace : nullptr;
  const dim3 grid((unsigned)((M + p.bm - 1) / p.bm), (unsigned)((N + p.bn - 1) / p.bn), (unsigned)p.splits);
  if (p.bn == 64 && p.bm == 64 && p.bk == 64)
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64, 64, 1>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc,
                       transA, transB, accumulate, act, p.k_per_split, slabs);
  else if (p.bn == 64 && p.bm == 64)
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, transA,
                       transB, accumulate, act, p.k_per_split, slabs);
  else if (p.bn == 64 && p.bk == 64)
    hipLaunchKernelGGL((gemm_f32_kernel<128, 64, 64, 1>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc,
                       transA, transB, accumulate, act, p.k_per_split, slabs);
  else if (p.bn == 64)
    hipLaunchKernelGGL((gemm_f32_kernel<128, 64>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, transA,
                       transB, accumulate, act, p.k_per_split, slabs);
  else if (p.bm == 64)
    hipLaunchKernelGGL((gemm_f32_kernel<64, 128>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, transA,
                       transB, accumulate, act, p.k_per_split, slabs);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<128, 128>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, transA,
                       transB, accumulate, act, p.k_per_split, slabs);
  int rc = check_launch("gemm_f32_kernel");
  if (rc || p.splits == 1) return rc;
  hipLaunchKernelGGL(gemm_reduce_slabs_kernel, dim3((unsigned)((M * N + 31) / 32)), dim3(32 * reduce_lanes(p.splits, M * N)), 0, st, slabs,
                     p.splits, C, bias, M, N, ldc, accumulate, act, (float *)nullptr, (int64_t)0, (int64_t)INT64_MAX);
  return check_launch("gemm_reduce_slabs_kernel");
}

extern "C" int chaorec_adam_step_f32(float *param, const float *grad, float *exp_avg,
                                     float *exp_avg_sq, int64_t n, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, int32_t step,
                                     const int32_t *step_dev, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(CHAOREC_E_INVALID, "adam: NULL argument");
  if (n < 0 || (!step_dev && step < 1)) return fail(CHAOREC_E_INVALID, "adam: n=%lld step=%d", (long long)n, step);
  if (n == 0) return CHAOREC_OK;
  const int vec4 = ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0) ? 1 : 0;
  int64_t blocks = ((vec4 ? (n + 3) / 4 : n) + 255) / 256;
  if (blocks > 2048) blocks = 2048;   // a few waves per SIMD; every block pays two double pow() for the bias corrections
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, step_dev, vec4);
  return check_launch("adam_step_kernel");
}

extern "C" int chaorec_abi_version(void) { return CHAOREC_ABI_VERSION; }
extern "C" const char *chaorec_last_error(void) { return chaorec::err_buf(); }
