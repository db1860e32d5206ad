// All-items scoring + history mask + top-K without materialising the [U, I] score matrix.
//
// Replaces gene_ranklist's torch.matmul + python mask loop + torch.topk
// (Model/LightGCN.py:147-155, Model/FREEDOM.py:230-238, Model/MMGCN.py:220-230).
//
// precision 0 (this file): exact fp32 on the f32 MFMA pipe, v_mfma_f32_32x32x2_f32.
//   One wave64 = 32 users x a stream of 32-item tiles.  Items are the MFMA A rows, users the
//   B columns, so the 32x32 accumulator puts ONE user on each lane (column = lane & 31) with
//   16 of the tile's items in its registers: selection needs no cross-lane traffic.
//   The users' fragment (D/2 floats per lane) stays in registers for the whole stream; item
//   fragments come straight from global/L2 (the item table is a few MB and shared by every
//   wave), prefetched one tile ahead of the 32*D MFMA cycles that consume them.
//   Selection: per-user threshold tau (current K-th best) in a register; a score that beats
//   tau is appended to the user's 128-entry LDS candidate list with a slot computed from
//   popcounts (no atomics); a list that could overflow is pruned back to K by a wave-wide
//   register bitonic sort of 64-bit keys (score bits | inverted item index), which also
//   yields the new tau.  Ties: the key orders equal scores by ascending item index.
#include "common.h"
#include <limits.h>

namespace chaorec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCand = 128;            // candidate capacity per user (>= K + 32)
constexpr int kCandStride = kCand + 1;  // 64-bit entries; odd stride spreads users over banks
constexpr int kMaxK = 64;

__device__ __forceinline__ uint32_t f32_to_ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_to_f32(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t item) {
  return ((uint64_t)f32_to_ord(score) << 32) | (uint64_t)(0xFFFFFFFFu - item);
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl(lo, src, 64);
  hi = __shfl(hi, src, 64);
  return ((uint64_t)hi << 32) | lo;
}

// Wave-wide descending bitonic sort of 128 keys: element i lives in lane (i & 63), e0 for
// i < 64 and e1 for i >= 64.  Afterwards e0 holds ranks 0..63, e1 ranks 64..127.
__device__ __forceinline__ void sort128_desc(uint64_t &e0, uint64_t &e1, int lane) {
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {
        // only at k == 128: partner is the other register of the same lane, block is descending
        const uint64_t hi = e0 > e1 ? e0 : e1, lo = e0 > e1 ? e1 : e0;
        e0 = hi;
        e1 = lo;
      } else {
        const uint64_t p0 = shfl_xor_u64(e0, j), p1 = shfl_xor_u64(e1, j);
        const bool lower = (lane & j) == 0;
        const bool desc0 = (k == 128) ? true : ((lane & k) == 0);
        const bool desc1 = (k == 128) ? true : (k == 64 ? false : ((lane & k) == 0));
        const uint64_t mx0 = e0 > p0 ? e0 : p0, mn0 = e0 > p0 ? p0 : e0;
        const uint64_t mx1 = e1 > p1 ? e1 : p1, mn1 = e1 > p1 ? p1 : e1;
        e0 = (lower == desc0) ? mx0 : mn0;
        e1 = (lower == desc1) ? mx1 : mn1;
      }
    }
  }
}

template <int D>
__device__ __forceinline__ void load_item_frag(float (&a)[D / 2], const float *__restrict__ item_emb,
                                               int64_t j, int64_t i_end, int h) {
  if (j < i_end) {
    const float4 *src = reinterpret_cast<const float4 *>(item_emb + (size_t)j * D + h * (D / 2));
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
      const float4 v = src[q];
      a[4 * q + 0] = v.x;
      a[4 * q + 1] = v.y;
      a[4 * q + 2] = v.z;
      a[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < D / 2; ++s) a[s] = 0.f;
  }
}

// D > 0: K-dim known at compile time, the users' fragment lives in registers for the whole stream.
// D == 0 ("stream"): K-dim = Dk at run time (multiple of 64; the kNN build over 384/4096-wide
// modality features): both operands are re-read per 64-wide k-chunk, B from L1/L2.
template <int D>
__global__ __launch_bounds__(64) void score_topk_f32_kernel(
    const float *__restrict__ user_emb, const float *__restrict__ item_emb, int64_t n_users,
    int64_t n_items, int Dk, const int64_t *__restrict__ hist_rowptr, const int32_t *__restrict__ hist_col,
    float mask_value, int K, int64_t id_offset, int64_t *__restrict__ out_idx,
    float *__restrict__ out_val, uint64_t *__restrict__ partial, int splits,
    int64_t items_per_split) {
  __shared__ uint64_t cand[32 * kCandStride];
  const int lane = threadIdx.x;
  const int ur = lane & 31;
  const int h = lane >> 5;
  const int64_t u = (int64_t)blockIdx.x * 32 + ur;
  const bool u_ok = u < n_users;
  const int split = blockIdx.y;
  const int64_t i_begin = (int64_t)split * items_per_split;
  const int64_t i_end = min(n_items, i_begin + items_per_split);

  constexpr int DR = D > 0 ? D : 64;  // register fragment width
  // users' B fragment: lane (ur, h) holds k = h*D/2 + s
  float bu[DR / 2];
  if (D == 0) {
  } else if (u_ok) {
    const float4 *src = reinterpret_cast<const float4 *>(user_emb + (size_t)u * DR + h * (DR / 2));
#pragma unroll
    for (int q = 0; q < DR / 8; ++q) {
      const float4 v = src[q];
      bu[4 * q + 0] = v.x;
      bu[4 * q + 1] = v.y;
      bu[4 * q + 2] = v.z;
      bu[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < DR / 2; ++s) bu[s] = 0.f;
  }

  // history cursor: first interacted item >= i_begin
  int64_t hp = 0, hend = 0;
  int64_t hnext = LLONG_MAX;
  if (hist_rowptr && u_ok) {
    int64_t lo = hist_rowptr[u];
    hend = hist_rowptr[u + 1];
    int64_t hi = hend;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)hist_col[mid] < i_begin) lo = mid + 1; else hi = mid;
    }
    hp = lo;
    if (hp < hend) hnext = hist_col[hp];
  }

  float tau = -INFINITY;
  int cnt = 0;  // identical on both lanes of a user
  uint64_t *my = cand + ur * kCandStride;

  float a_cur[DR / 2], a_nxt[DR / 2];
  if (D > 0) load_item_frag<DR>(a_cur, item_emb, i_begin + ur, i_end, h);

  for (int64_t j0 = i_begin; j0 < i_end; j0 += 32) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (D > 0) {
      load_item_frag<DR>(a_nxt, item_emb, j0 + 32 + ur, i_end, h);
#pragma unroll
      for (int s = 0; s < DR / 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], bu[s], acc, 0, 0, 0);
    } else {
      const int64_t j = j0 + ur;
      for (int kc = 0; kc < Dk; kc += 64) {
        float4 av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          bv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < i_end) av[q] = reinterpret_cast<const float4 *>(item_emb + (size_t)j * Dk + kc + h * 32)[q];
          if (u_ok) bv[q] = reinterpret_cast<const float4 *>(user_emb + (size_t)u * Dk + kc + h * 32)[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
        }
      }
    }

    // prune the lists that a full tile could overflow (wave-uniform loop over such users)
    {
      const bool need = cnt > kCand - 32;
      uint32_t m = (uint32_t)(__ballot(need) & 0xFFFFFFFFull);
      while (m) {
        const int t = __builtin_ctz(m);
        m &= m - 1;
        const int ct = __shfl(cnt, t, 64);
        uint64_t *buf = cand + t * kCandStride;
        uint64_t e0 = lane < ct ? buf[lane] : 0ull;
        uint64_t e1 = lane + 64 < ct ? buf[lane + 64] : 0ull;
        sort128_desc(e0, e1, lane);
        if (lane < K) buf[lane] = e0;
        const uint64_t kth = shfl_u64(e0, K - 1);
        if (ur == t) {
          cnt = K;  // ct > 96 >= K here
          tau = ord_to_f32((uint32_t)(kth >> 32));
        }
      }
    }

    // history mask bits for this tile
    uint32_t mbits = 0;
    while (hnext < j0 + 32) {
      mbits |= 1u << (uint32_t)(hnext - j0);
      ++hp;
      hnext = hp < hend ? (int64_t)hist_col[hp] : LLONG_MAX;
    }

    // qualify: strictly better than the current K-th (later equal scores lose: lowest index first)
    uint32_t qbits = 0;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float s = acc[reg];
      if ((mbits >> off) & 1u) s = mask_value;
      acc[reg] = s;
      const bool q = u_ok && (j0 + off < i_end) && (s > tau || cnt < K);
      qbits |= q ? (1u << reg) : 0u;
    }
    if (__any(qbits != 0)) {
      const int nq = __popc(qbits);
      const int nqp = __shfl_xor(nq, 32, 64);
      int slot = cnt + (h ? nqp : 0);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        if ((qbits >> reg) & 1u) {
          const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          my[slot++] = make_key(acc[reg], (uint32_t)(j0 + off - 0));
        }
      }
      cnt += nq + nqp;
    }

    if (D > 0) {
#pragma unroll
      for (int s = 0; s < DR / 2; ++s) a_cur[s] = a_nxt[s];
    }
  }

  // final ordering of every user's list
  for (int t = 0; t < 32; ++t) {
    const int ct = __shfl(cnt, t, 64);
    const int64_t ut = (int64_t)blockIdx.x * 32 + t;
    if (ut >= n_users) break;
    uint64_t *buf = cand + t * kCandStride;
    uint64_t e0 = lane < ct ? buf[lane] : 0ull;
    uint64_t e1 = lane + 64 < ct ? buf[lane + 64] : 0ull;
    sort128_desc(e0, e1, lane);
    if (lane < K) {
      if (splits > 1) {
        partial[((size_t)split * n_users + ut) * K + lane] = e0;
      } else {
        const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e0 & 0xFFFFFFFFull);
        out_idx[(size_t)ut * K + lane] = (int64_t)item + id_offset;
        out_val[(size_t)ut * K + lane] = ord_to_f32((uint32_t)(e0 >> 32));
      }
    }
  }
}

// Merge the per-split top-K lists: one wave per user, running best-64 in e0.
__global__ __launch_bounds__(64) void score_topk_merge_kernel(const uint64_t *__restrict__ partial,
                                                              int64_t n_users, int K, int splits,
                                                              int64_t id_offset,
                                                              int64_t *__restrict__ out_idx,
                                                              float *__restrict__ out_val) {
  const int lane = threadIdx.x;
  const int64_t u = blockIdx.x;
  uint64_t e0 = 0ull;
  for (int s = 0; s < splits; ++s) {
    uint64_t e1 = lane < K ? partial[((size_t)s * n_users + u) * K + lane] : 0ull;
    sort128_desc(e0, e1, lane);
  }
  if (lane < K) {
    const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e0 & 0xFFFFFFFFull);
    out_idx[(size_t)u * K + lane] = (int64_t)item + id_offset;
    out_val[(size_t)u * K + lane] = ord_to_f32((uint32_t)(e0 >> 32));
  }
}

static void plan_splits(int64_t n_users, int64_t n_items, int K, int *splits, int64_t *per_split) {
  const int64_t groups = (n_users + 31) / 32;
  int64_t s = (2048 + groups - 1) / groups;  // aim for >= 2 waves per SIMD over 256 CUs
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  int64_t per = (n_items + s - 1) / s;
  per = (per + 31) / 32 * 32;
  const int64_t min_per = ((int64_t)(K > 256 ? K : 256) + 31) / 32 * 32;
  if (per < min_per) per = min_per;
  s = (n_items + per - 1) / per;
  if (s < 1) s = 1;
  *splits = (int)s;
  *per_split = per;
}

}  // namespace chaorec

using namespace chaorec;

extern "C" size_t chaorec_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int32_t K) {
  if (n_users <= 0 || n_items <= 0 || K <= 0) return 0;
  int splits;
  int64_t per;
  plan_splits(n_users, n_items, K, &splits, &per);
  if (splits <= 1) return 0;
  return (size_t)splits * (size_t)n_users * (size_t)K * sizeof(uint64_t);
}

extern "C" int chaorec_score_topk_f32(const float *user_emb, const float *item_emb, int64_t n_users,
                                      int64_t n_items, int32_t D, const int64_t *hist_rowptr,
                                      const int32_t *hist_col, float mask_value, int32_t K,
                                      int64_t id_offset, int64_t *out_idx, float *out_val,
                                      void *workspace, size_t workspace_bytes, int32_t precision,
                                      void *stream) {
  if (!user_emb || !item_emb || !out_idx || !out_val) return fail(CHAOREC_E_INVALID, "score_topk: NULL argument");
  if (n_users < 0 || n_items <= 0) return fail(CHAOREC_E_INVALID, "score_topk: bad sizes");
  if (K < 1 || K > kMaxK) return fail(CHAOREC_E_INVALID, "score_topk: K=%d must be in [1,%d]", K, kMaxK);
  if (n_items < K) return fail(CHAOREC_E_INVALID, "score_topk: n_items=%lld < K=%d (torch.topk would raise)", (long long)n_items, K);
  if (n_items > 0xFFFFFFF0ll) return fail(CHAOREC_E_INVALID, "score_topk: n_items too large");
  if (precision != 0) return fail(CHAOREC_E_INVALID, "score_topk: precision %d not built", precision);
  if (hist_rowptr && !hist_col) return fail(CHAOREC_E_INVALID, "score_topk: hist_rowptr without hist_col");
  if (n_users == 0) return CHAOREC_OK;
  int splits;
  int64_t per;
  plan_splits(n_users, n_items, K, &splits, &per);
  const size_t need = splits > 1 ? (size_t)splits * (size_t)n_users * (size_t)K * sizeof(uint64_t) : 0;
  if (need > workspace_bytes || (need && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "score_topk: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((n_users + 31) / 32), (unsigned)splits);
  uint64_t *partial = (uint64_t *)workspace;
#define CHAOREC_ST_ARGS user_emb, item_emb, n_users, n_items, (int)D, hist_rowptr, hist_col, mask_value, K, \
                        id_offset, out_idx, out_val, partial, splits, per
  switch (D) {
    case 8: hipLaunchKernelGGL(score_topk_f32_kernel<8>, grid, dim3(64), 0, st, CHAOREC_ST_ARGS); break;
    case 16: hipLaunchKernelGGL(score_topk_f32_kernel<16>, grid, dim3(64), 0, st, CHAOREC_ST_ARGS); break;
    case 32: hipLaunchKernelGGL(score_topk_f32_kernel<32>, grid, dim3(64), 0, st, CHAOREC_ST_ARGS); break;
    case 64: hipLaunchKernelGGL(score_topk_f32_kernel<64>, grid, dim3(64), 0, st, CHAOREC_ST_ARGS); break;
    case 128: hipLaunchKernelGGL(score_topk_f32_kernel<128>, grid, dim3(64), 0, st, CHAOREC_ST_ARGS); break;
    default:
      if (D > 128 && (D % 64) == 0) {
        hipLaunchKernelGGL(score_topk_f32_kernel<0>, grid, dim3(64), 0, st, CHAOREC_ST_ARGS);
        break;
      }
      return fail(CHAOREC_E_INVALID, "score_topk: D=%d not in {8,16,32,64,128} and not a multiple of 64 above 128", D);
  }
#undef CHAOREC_ST_ARGS
  int rc = check_launch("score_topk_f32_kernel");
  if (rc) return rc;
  if (splits > 1) {
    hipLaunchKernelGGL(score_topk_merge_kernel, dim3((unsigned)n_users), dim3(64), 0, st, partial,
                       n_users, K, splits, id_offset, out_idx, out_val);
    rc = check_launch("score_topk_merge_kernel");
  }
  return rc;
}
