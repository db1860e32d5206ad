// All-items scoring + history mask + top-K without materialising the [U, I] score matrix.
//
// Replaces gene_ranklist's torch.matmul + python mask loop + torch.topk
// (Model/LightGCN.py:147-155, Model/FREEDOM.py:230-238, Model/MMGCN.py:220-230).
//
// Three routes to the same result -- the exact top-K of the fp32 scores defined by the k-ordered fmaf chain of
// v_mfma_f32_32x32x2_f32 (oracle_score_dot), ties to the lowest index:
//   precision 0  bf16-MFMA prefilter + exact fp32 re-score (score_prefilter.hpp) where it applies, else route 2
//   precision 2  fp32 MFMA sweep with sampled thresholds: score_candidates_kernel + score_select_kernel (this file)
//   precision 1  one unthresholded fp32 pass: score_topk_f32_kernel (this file; also the streamed-K kNN build)
// Common geometry of the fp32 kernels: one wave64 = 32 users x a stream of 32-item tiles.  Items are the MFMA A
// rows, users the B columns, so the 32x32 accumulator puts ONE user on each lane (column = lane & 31) with 16 of
// the tile's items in its registers: selection needs no cross-lane traffic.  The users' fragment (D/2 floats per
// lane) stays in registers for the whole stream; item fragments come straight from global/L2, prefetched one tile
// ahead of the 32*D MFMA cycles that consume them.  score_topk_f32_kernel keeps a per-user threshold (current K-th
// best) in a register; a score that beats it is appended to the lane's LDS candidate list; a list that could
// overflow is pruned back to K by a wave-wide register bitonic sort of 64-bit keys (score bits | inverted item
// index), which also yields the new threshold.  Ties: the key orders equal scores by ascending item index.
#include "common.h"
#include <limits.h>
#include <mutex>
#include <unordered_map>

namespace chaorec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHalf = 64;               // candidate capacity per LANE (a user has two lanes = 128)
constexpr int kHalfStride = kHalf + 1;  // 64-bit entries; odd stride spreads lanes over banks
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

// The same network over NB independent (e0,e1) pairs at once: a lone sort is a chain of 27 dependent
// cross-lane shuffles (~100 cycles of latency each); interleaving four of them fills that latency.
template <int NB>
__device__ __forceinline__ void sort128_desc_batch(uint64_t (&e0)[NB], uint64_t (&e1)[NB], int lane) {
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const uint64_t hi = e0[b] > e1[b] ? e0[b] : e1[b], lo = e0[b] > e1[b] ? e1[b] : e0[b];
          e0[b] = hi;
          e1[b] = lo;
        }
      } else {
        uint64_t p0[NB], p1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          p0[b] = shfl_xor_u64(e0[b], j);
          p1[b] = shfl_xor_u64(e1[b], j);
        }
        const bool lower = (lane & j) == 0;
        const bool desc0 = (k == 128) ? true : ((lane & k) == 0);
        const bool desc1 = (k == 128) ? true : (k == 64 ? false : ((lane & k) == 0));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          // keep the larger iff (this lane holds the lower index) == (block is descending)
          e0[b] = ((e0[b] > p0[b]) == (lower == desc0)) ? e0[b] : p0[b];
          e1[b] = ((e1[b] > p1[b]) == (lower == desc1)) ? e1[b] : p1[b];
        }
      }
    }
  }
}

// ---- item packing ---------------------------------------------------------------------------
// The A fragment of lane (r, h) for tile t is D/2 consecutive floats of item row 32t + r: read in
// place that is 64 different 128-B lines per load instruction.  pack_items_kernel rewrites the table
// once per call into the kernel's own order, P[(t * D/8 + q) * 64 + lane] (float4), so every fragment
// load is one fully coalesced 1 KiB wave access; rows past n_items are zero.
__global__ __launch_bounds__(256) void pack_items_kernel(const float *__restrict__ item_emb,
                                                         float4 *__restrict__ packed, int64_t n_items,
                                                         int D, int64_t n_tiles) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the packed table
  const int Q = D / 8;
  if (i >= n_tiles * Q * 64) return;
  const int lane = (int)(i & 63);
  const int64_t tq = i >> 6;
  const int q = (int)(tq % Q);
  const int64_t t = tq / Q;
  const int64_t j = t * 32 + (lane & 31);
  const int h = lane >> 5;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n_items) v = reinterpret_cast<const float4 *>(item_emb + (size_t)j * D + h * (D / 2))[q];
  packed[i] = v;
}

template <int D>
__device__ __forceinline__ void load_item_frag(float (&a)[D / 2], const float *__restrict__ item_emb,
                                               const float4 *__restrict__ packed, int64_t t, int64_t n_items,
                                               int lane) {
  if (packed) {
    const float4 *src = packed + (size_t)t * (D / 8) * 64 + lane;
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
      const float4 v = src[q * 64];
      a[4 * q + 0] = v.x;
      a[4 * q + 1] = v.y;
      a[4 * q + 2] = v.z;
      a[4 * q + 3] = v.w;
    }
    return;
  }
  const int64_t j = t * 32 + (lane & 31);
  const int h = lane >> 5;
  if (j < n_items) {
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

enum { kModeMain = 0, kModeSample = 1, kModeFallback = 2 };

struct ScoreArgs {
  const float *user_emb;
  const float *item_emb;
  const float4 *packed;        // packed item table or NULL
  int64_t n_users, n_items;
  int Dk;                      // run-time K-dim of the streamed variant
  const int64_t *hist_rowptr;
  const int32_t *hist_col;
  float mask_value;
  int K;
  int64_t id_offset;
  int64_t *out_idx;
  float *out_val;
  uint64_t *partial;           // [splits][n_users][K] keys when splits > 1
  int splits;
  int64_t tiles_per_split;
  int mode;
  int tile_stride;             // sampling passes: every tile_stride-th tile
  int max_tiles;               // kModeSample: stop after this many sampled tiles (0 = no limit)
  float *tau;                  // kModeSample: out; kModeMain: in (may be NULL)
  int *certify;                // kModeMain, splits == 1: out per-user "threshold was too high" flags
  const int *fail;             // kModeFallback: per-user flags
  // compact user set (the prefilter route's exact fallback at large item counts): row u of the launch is user
  // user_map[u] of the tables, and only the first min(*n_users_dev, n_users) rows exist.  NULL = identity / n_users.
  const int *user_map;
  const int *n_users_dev;
};

// D > 0: K-dim known at compile time, the users' fragment lives in registers for the whole stream.
// D == 0 ("stream"): K-dim = Dk at run time (multiple of 64; the kNN build over 384/4096-wide
// modality features): both operands are re-read per 64-wide k-chunk, B from L1/L2.
//
// Selection is the expensive part (the MFMA chain of a 32x32 tile is 32*D cycles; a naive
// compare/append pass over the 16 scores a lane holds was 3x that), so the per-tile VALU work is kept
// to ~10 instructions per score: each lane owns its OWN 64-entry candidate list (no slot arithmetic
// between the two lanes of a user), a score is appended under one exec-masked compare, the history
// mask and the range check only run on tiles that need them (wave-uniform branches).
template <int D>
__global__ __launch_bounds__(64) void score_topk_f32_kernel(const ScoreArgs A) {
  __shared__ uint64_t cand[64 * kHalfStride];
  const int lane = threadIdx.x;
  const int ur = lane & 31;
  const int h = lane >> 5;
  const int64_t n_users_eff = A.n_users_dev ? min((int64_t)*A.n_users_dev, A.n_users) : A.n_users;
  if ((int64_t)blockIdx.x * 32 >= n_users_eff) return;       // (only a compact launch has empty groups)
  const int64_t uc = (int64_t)blockIdx.x * 32 + ur;          // row of this launch
  const bool u_ok = uc < n_users_eff;
  const int64_t u = (A.user_map && u_ok) ? (int64_t)A.user_map[uc] : uc;   // row of the tables
  const int K = A.K;
  const uint32_t n_items = (uint32_t)A.n_items;
  const int n_tiles = (int)((A.n_items + 31) / 32);
  const int split = blockIdx.y;
  int t_begin = (int)((int64_t)split * A.tiles_per_split);
  int t_end = (int)min((int64_t)n_tiles, (int64_t)t_begin + A.tiles_per_split);
  int t_stride = 1;
  if (A.mode == kModeSample) {
    t_begin = 0;
    t_end = n_tiles;
    t_stride = A.tile_stride;
    if (A.max_tiles > 0) t_end = min(t_end, A.max_tiles * t_stride);
  } else if (A.mode == kModeFallback) {
    // only the groups that hold a user whose thresholded pass came up short re-run, unthresholded
    // (same item splits as the main sweep, merged afterwards for the flagged users only)
    if (!__any(u_ok && A.fail[u] != 0)) return;
  }

  constexpr int DR = D > 0 ? D : 64;  // register fragment width
  // users' B fragment: lane (ur, h) holds k = h*D/2 + s
  float bu[DR / 2];
  if (D == 0) {
  } else if (u_ok) {
    const float4 *src = reinterpret_cast<const float4 *>(A.user_emb + (size_t)u * DR + h * (DR / 2));
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

  // history cursor: first interacted item >= first item of the range
  int64_t hp = 0, hend = 0;
  uint32_t hnext = 0xFFFFFFFFu;
  if (A.hist_rowptr && u_ok) {
    int64_t lo = A.hist_rowptr[u];
    hend = A.hist_rowptr[u + 1];
    int64_t hi = hend;
    const uint32_t first = (uint32_t)t_begin * 32u;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint32_t)A.hist_col[mid] < first) lo = mid + 1; else hi = mid;
    }
    hp = lo;
    if (hp < hend) hnext = (uint32_t)A.hist_col[hp];
  }

  // tau0: a lower bound of the user's K-th best score from the sampling pass (exclusive: s > tau0
  // keeps every score >= the sampled rank value because the sampler stores the next float below it)
  float tau0 = -INFINITY;
  if (A.mode == kModeMain && A.tau && u_ok) tau0 = A.tau[u];
  float tau = u_ok ? tau0 : INFINITY;  // padding users never qualify
  int cnt = 0;                         // entries in THIS lane's list
  uint64_t *my = cand + lane * kHalfStride;

  // prune up to NB users' half-list pairs back to their K best (wave-wide register bitonic sorts,
  // NB of them interleaved); tus[b] < 0 = unused slot
  constexpr int NB = 4;
  auto prune_batch = [&](const int (&tus)[NB]) {
    uint64_t e0[NB], e1[NB];
    int tot[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int tu = tus[b] < 0 ? 0 : tus[b];
      const int c0 = tus[b] < 0 ? 0 : __shfl(cnt, tu, 64), c1 = tus[b] < 0 ? 0 : __shfl(cnt, tu + 32, 64);
      e0[b] = lane < c0 ? cand[tu * kHalfStride + lane] : 0ull;
      e1[b] = lane < c1 ? cand[(tu + 32) * kHalfStride + lane] : 0ull;
      tot[b] = c0 + c1;
    }
    sort128_desc_batch<NB>(e0, e1, lane);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (tus[b] < 0) continue;  // wave-uniform
      const int tu = tus[b];
      const int keep = min(tot[b], K);
      if (lane < keep) cand[(tu + 32 * (lane & 1)) * kHalfStride + (lane >> 1)] = e0[b];
      const uint64_t kth = shfl_u64(e0[b], K - 1);
      if (ur == tu) {
        cnt = h ? keep / 2 : (keep + 1) / 2;
        if (tot[b] >= K) tau = fmaxf(tau, ord_to_f32((uint32_t)(kth >> 32)));
      }
    }
  };

  auto select = [&](f32x16 &acc, int t) {
    const uint32_t j0 = (uint32_t)t * 32u;
    // lists that a full tile could overflow (wave-uniform loop over such users)
    {
      // (the shuffle must run with every lane active: no short-circuit in front of it)
      const int mine = cnt > kHalf - 16 ? 1 : 0;
      const int partner = __shfl_xor(mine, 32, 64);
      const bool need = (mine | partner) != 0;
      uint32_t m = (uint32_t)(__ballot(need) & 0xFFFFFFFFull);
      while (m) {
        int tus[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          tus[b] = m ? __builtin_ctz(m) : -1;
          m &= m - 1;
        }
        prune_batch(tus);
      }
    }
    // history mask for this tile (entries of skipped tiles are stepped over)
    uint32_t mbits = 0;
    while (hnext < j0 + 32u) {
      if (hnext >= j0) mbits |= 1u << (hnext - j0);
      ++hp;
      hnext = hp < hend ? (uint32_t)A.hist_col[hp] : 0xFFFFFFFFu;
    }
    if (__any(mbits != 0)) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if ((mbits >> off) & 1u) acc[reg] = A.mask_value;
      }
    }
    if (j0 + 32u > n_items) {  // last, partial tile: rows past n_items never qualify
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (j0 + off >= n_items) acc[reg] = -INFINITY;
      }
    }
    // strictly better than the current K-th (later equal scores lose: lowest index first)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      if (acc[reg] > tau) {
        const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        my[cnt++] = make_key(acc[reg], j0 + off);
      }
    }
  };

  if (D > 0) {
    float a0[DR / 2], a1[DR / 2];
    int t = t_begin;
    if (t < t_end) load_item_frag<DR>(a0, A.item_emb, A.packed, t, A.n_items, lane);
    while (t < t_end) {
      {
        const int tn = t + t_stride < t_end ? t + t_stride : t;
        load_item_frag<DR>(a1, A.item_emb, A.packed, tn, A.n_items, lane);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < DR / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bu[s], acc, 0, 0, 0);
        select(acc, t);
        t += t_stride;
      }
      if (t >= t_end) break;
      {
        const int tn = t + t_stride < t_end ? t + t_stride : t;
        load_item_frag<DR>(a0, A.item_emb, A.packed, tn, A.n_items, lane);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < DR / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bu[s], acc, 0, 0, 0);
        select(acc, t);
        t += t_stride;
      }
    }
  } else {
    for (int t = t_begin; t < t_end; t += t_stride) {
      const int64_t j = (int64_t)t * 32 + ur;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      for (int kc = 0; kc < A.Dk; kc += 64) {
        float4 av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          bv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < A.n_items) av[q] = reinterpret_cast<const float4 *>(A.item_emb + (size_t)j * A.Dk + kc + h * 32)[q];
          if (u_ok) bv[q] = reinterpret_cast<const float4 *>(A.user_emb + (size_t)u * A.Dk + kc + h * 32)[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
        }
      }
      select(acc, t);
    }
  }

  // final ordering of every user's list, NB users per pass
  for (int base = 0; base < 32; base += NB) {
    if ((int64_t)blockIdx.x * 32 + base >= A.n_users) break;
    uint64_t e0[NB], e1[NB];
    int tot[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int tu = base + b;
      const int c0 = __shfl(cnt, tu, 64), c1 = __shfl(cnt, tu + 32, 64);
      e0[b] = lane < c0 ? cand[tu * kHalfStride + lane] : 0ull;
      e1[b] = lane < c1 ? cand[(tu + 32) * kHalfStride + lane] : 0ull;
      tot[b] = c0 + c1;
    }
    sort128_desc_batch<NB>(e0, e1, lane);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t ut = (int64_t)blockIdx.x * 32 + base + b;
      if (ut >= n_users_eff) continue;
      const uint64_t kth = shfl_u64(e0[b], K - 1);
      const float kth_val = (tot[b] >= K) ? ord_to_f32((uint32_t)(kth >> 32)) : -INFINITY;
      if (A.mode == kModeSample) {
        // K-th best of the sample, one float below (so `s > tau0` keeps equal scores); -inf if the
        // sample holds fewer than K valid items
        if (lane == 0) A.tau[ut] = (tot[b] >= K) ? nextafterf(kth_val, -INFINITY) : -INFINITY;
        continue;
      }
      if (A.mode == kModeFallback && A.fail[ut] == 0) continue;
      if (A.mode == kModeMain && A.certify && A.splits == 1 && lane == 0) {
        // exact iff at least K scores lie above the sampled threshold, i.e. the K-th best does
        const float t0 = A.tau ? A.tau[ut] : -INFINITY;
        A.certify[ut] = ((tot[b] >= K && kth_val > t0) || t0 == -INFINITY) ? 0 : 1;
      }
      if (lane < K) {
        if (A.splits > 1) {
          A.partial[((size_t)split * A.n_users + ut) * K + lane] = e0[b];
        } else {
          const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e0[b] & 0xFFFFFFFFull);
          const size_t uo = A.user_map ? (size_t)A.user_map[ut] : (size_t)ut;
          A.out_idx[uo * K + lane] = (int64_t)item + A.id_offset;
          A.out_val[uo * K + lane] = ord_to_f32((uint32_t)(e0[b] >> 32));
        }
      }
    }
  }
}

// Merge the per-split top-K lists: one wave per user, running best-64 in e0.  With a sampled
// threshold in play it also certifies the result: the merged K-th best must lie above tau0 (= at
// least K scores passed the threshold), else the user is flagged for the unthresholded fallback.
__global__ __launch_bounds__(64) void score_topk_merge_kernel(const uint64_t *__restrict__ partial,
                                                              int64_t n_users, int K, int splits,
                                                              int64_t id_offset,
                                                              int64_t *__restrict__ out_idx,
                                                              float *__restrict__ out_val,
                                                              const float *__restrict__ tau,
                                                              int *__restrict__ fail,
                                                              const int *__restrict__ only_if,
                                                              const int *__restrict__ user_map,
                                                              const int *__restrict__ n_users_dev,
                                                              float *__restrict__ hint_out = nullptr) {
  const int lane = threadIdx.x;
  const int64_t u = blockIdx.x;
  if (only_if && only_if[u] == 0) return;
  if (n_users_dev && u >= *n_users_dev) return;
  uint64_t e0 = 0ull;
  for (int s = 0; s < splits; ++s) {
    uint64_t e1 = lane < K ? partial[((size_t)s * n_users + u) * K + lane] : 0ull;
    sort128_desc(e0, e1, lane);
  }
  if (fail) {
    const uint64_t kth = shfl_u64(e0, K - 1);
    if (lane == 0) {
      const float t0 = tau[u];
      const bool ok = t0 == -INFINITY || (kth != 0ull && ord_to_f32((uint32_t)(kth >> 32)) > t0);
      fail[u] = ok ? 0 : 1;
    }
  }
  const size_t uo = user_map ? (size_t)user_map[u] : (size_t)u;
  if (lane < K) {
    const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e0 & 0xFFFFFFFFull);
    out_idx[uo * K + lane] = (int64_t)item + id_offset;
    out_val[uo * K + lane] = ord_to_f32((uint32_t)(e0 >> 32));
  }
  if (hint_out) {
    // (the grouped fallback of the prefilter route: these users' thresholds cut too close -- the next call's is taken as
    //  the exact per-user route takes it, score_exact_user_kernel: ranks 32 and 64 extrapolated as far again below rank 64.
    //  Without it the user kept the threshold that had just failed, and failed again at every carried-threshold call.)
    const uint32_t o31 = (uint32_t)__shfl((int)(uint32_t)(e0 >> 32), 31, 64), o63 = (uint32_t)__shfl((int)(uint32_t)(e0 >> 32), 63, 64);
    const uint64_t ek = shfl_u64(e0, K - 1);
    float t = nextafterf(ord_to_f32((uint32_t)(ek >> 32)), -INFINITY);
    if (shfl_u64(e0, 63) != 0ull) {
      const float s31 = ord_to_f32(o31), s63 = ord_to_f32(o63);
      t = fminf(t, s63 - (s31 - s63));
    }
    if (lane == 0) hint_out[uo] = t;
  }
}

// ---- thresholded candidate pass (the fast main pass) ---------------------------------------
// With a per-user threshold tau0 in hand (sampling pass) the full sweep needs no selection at all:
// a score above tau0 is appended to the lane's own list in GLOBAL memory, [split][user][half][kCandCap]
// 64-bit keys, and the exact top-K is taken afterwards from the ~6K survivors per user
// (score_select_kernel).  No LDS lists, no prunes, no sorts here, so two waves fit per SIMD and the
// compare/append work of one hides under the other's MFMA chain.  Per tile: 16 compares build a hit mask;
// the tile's scores are parked in a 4 KiB LDS scratch ([reg][lane], conflict-free both ways) so a lane can
// fetch "its" hit by a run-time register index; a short wave-uniform loop drains the hits.
constexpr int kCandCap = 96;  // keys per (split, user, half), <= 128; expected ~6K / (2 * splits)

#ifndef CHAOREC_CAND_MINW
#define CHAOREC_CAND_MINW 1
#endif
template <int D>
__global__ __launch_bounds__(64, CHAOREC_CAND_MINW) void score_candidates_kernel(const ScoreArgs A, uint64_t *__restrict__ cand_buf,
                                                              int *__restrict__ cand_cnt) {
  __shared__ float scratch[16 * 64];
  const int lane = threadIdx.x;
  const int ur = lane & 31;
  const int h = lane >> 5;
  const int64_t u = (int64_t)blockIdx.x * 32 + ur;
  const bool u_ok = u < A.n_users;
  const uint32_t n_items = (uint32_t)A.n_items;
  const int n_tiles = (int)((A.n_items + 31) / 32);
  const int split = blockIdx.y;
  // tile index = i * stride for i in this split's range of the (sampled) tile list
  const int t_stride = A.tile_stride > 0 ? A.tile_stride : 1;
  int n_samp = (n_tiles + t_stride - 1) / t_stride;
  if (A.max_tiles > 0) n_samp = min(n_samp, A.max_tiles);
  const int i_begin = (int)((int64_t)split * A.tiles_per_split);
  const int i_end = (int)min((int64_t)n_samp, (int64_t)i_begin + A.tiles_per_split);
  const int t_begin = i_begin * t_stride;
  const int t_end = min(n_tiles, (i_end - 1) * t_stride + 1);

  float bu[D / 2];
  if (u_ok) {
    const float4 *src = reinterpret_cast<const float4 *>(A.user_emb + (size_t)u * D + h * (D / 2));
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
      const float4 v = src[q];
      bu[4 * q + 0] = v.x;
      bu[4 * q + 1] = v.y;
      bu[4 * q + 2] = v.z;
      bu[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < D / 2; ++s) bu[s] = 0.f;
  }

  int64_t hp = 0, hend = 0;
  uint32_t hnext = 0xFFFFFFFFu;
  if (A.hist_rowptr && u_ok) {
    int64_t lo = A.hist_rowptr[u];
    hend = A.hist_rowptr[u + 1];
    int64_t hi = hend;
    const uint32_t first = (uint32_t)t_begin * 32u;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint32_t)A.hist_col[mid] < first) lo = mid + 1; else hi = mid;
    }
    hp = lo;
    if (hp < hend) hnext = (uint32_t)A.hist_col[hp];
  }

  float tau0 = u_ok ? (A.tau ? A.tau[u] : -INFINITY) : INFINITY;   // only ever raised (overflow prune below)
  int cnt = 0;
  uint64_t *mine = cand_buf + (((size_t)split * A.n_users + (u_ok ? u : 0)) * 2 + h) * kCandCap;
  const int K = A.K;

  auto select = [&](f32x16 &acc, int t) {
    const uint32_t j0 = (uint32_t)t * 32u;
    // A list that the next tile could overflow (scores bunched in this item range, or a loose tau0) is cut
    // back to the lane's own K best and the lane's threshold raised to its K-th: still exact, because the K
    // entries kept all beat anything rejected later.  Rare; wave-wide bitonic sort of the list.
    {
      unsigned long long m = __ballot(cnt > kCandCap - 16);
      if (m) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // our own global stores, read back below
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        while (m) {
          const int src = __builtin_ctzll(m);
          m &= m - 1;
          const int c = __shfl(cnt, src, 64);
          const unsigned long long base = ((unsigned long long)__shfl((int)((unsigned long long)mine >> 32), src, 64) << 32) |
                                          (unsigned int)__shfl((int)((unsigned long long)mine & 0xffffffffull), src, 64);
          uint64_t *lst = (uint64_t *)base;
          uint64_t e0 = lane < c ? lst[lane] : 0ull;
          uint64_t e1 = lane + 64 < c ? lst[lane + 64] : 0ull;
          sort128_desc(e0, e1, lane);
          if (lane < K) lst[lane] = e0;
          const uint64_t kth = shfl_u64(e0, K - 1);
          if (lane == src) {
            cnt = K;
            tau0 = fmaxf(tau0, ord_to_f32((uint32_t)(kth >> 32)));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      }
    }
    uint32_t mbits = 0;
    while (hnext < j0 + 32u) {
      if (hnext >= j0) mbits |= 1u << (hnext - j0);
      ++hp;
      hnext = hp < hend ? (uint32_t)A.hist_col[hp] : 0xFFFFFFFFu;
    }
    if (__any(mbits != 0)) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if ((mbits >> off) & 1u) acc[reg] = A.mask_value;
      }
    }
    if (j0 + 32u > n_items) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (j0 + off >= n_items) acc[reg] = -INFINITY;
      }
    }
    uint32_t qbits = 0;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) qbits |= (acc[reg] > tau0) ? (1u << reg) : 0u;
    if (__any(qbits != 0)) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) scratch[reg * 64 + lane] = acc[reg];
      __builtin_amdgcn_wave_barrier();
      while (__any(qbits != 0)) {
        if (qbits) {
          const int reg = __ffs(qbits) - 1;
          qbits &= qbits - 1;
          const float sc = scratch[reg * 64 + lane];
          const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          mine[cnt++] = make_key(sc, j0 + off);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  };

  float a0[D / 2], a1[D / 2];
  int t = t_begin;
  if (t < t_end) load_item_frag<D>(a0, A.item_emb, A.packed, t, A.n_items, lane);
  while (t < t_end) {
    {
      const int tn = t + t_stride < t_end ? t + t_stride : t;
      load_item_frag<D>(a1, A.item_emb, A.packed, tn, A.n_items, lane);
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < D / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bu[s], acc, 0, 0, 0);
      select(acc, t);
      t += t_stride;
    }
    if (t >= t_end) break;
    {
      const int tn = t + t_stride < t_end ? t + t_stride : t;
      load_item_frag<D>(a0, A.item_emb, A.packed, tn, A.n_items, lane);
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < D / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bu[s], acc, 0, 0, 0);
      select(acc, t);
      t += t_stride;
    }
  }
  if (u_ok) cand_cnt[((size_t)split * A.n_users + u) * 2 + h] = cnt;
}

// Exact top-K of one user's candidate lists (one wave per user).  The ~6K candidates are held 8 per lane; the
// K-th largest score is found by a 32-step bitwise search with wave ballots and scalar popcounts (no LDS
// traffic, no sort), the survivors (score >= that value: K plus ties) are compacted through LDS and ONE
// register bitonic sort orders them.  Certifies the threshold: at least K candidates in total <=> the K best
// scores all lie above tau0 <=> the answer is exact; otherwise the user is flagged for the unthresholded
// fallback.  tau_out != NULL: sampling stage -- write the K-th best (one float below; the incoming threshold if
// there are fewer than K candidates) as the user's threshold and nothing else.
constexpr int kSelNR = 8;  // candidates per lane held in registers (512 per user), more -> streaming path

__global__ __launch_bounds__(64) void score_select_kernel(const uint64_t *__restrict__ cand_buf,
                                                          const int *__restrict__ cand_cnt, int64_t n_users,
                                                          int K, int splits, int64_t id_offset,
                                                          const float *__restrict__ tau,
                                                          int64_t *__restrict__ out_idx,
                                                          float *__restrict__ out_val, int *__restrict__ fail,
                                                          float *__restrict__ tau_out) {
  __shared__ uint64_t stage[64 * kSelNR];
  const int lane = threadIdx.x;
  const int64_t u = blockIdx.x;
  const int n_lists = 2 * splits;
  int total = 0;
  for (int l = 0; l < n_lists; ++l) total += cand_cnt[((size_t)(l >> 1) * n_users + u) * 2 + (l & 1)];

  uint64_t e0 = 0ull, e1 = 0ull;
  bool sorted = false;
  if (total <= 64 * kSelNR) {
    // stage every list contiguously in LDS, then 8 keys per lane in registers
    int base = 0;
    for (int l = 0; l < n_lists; ++l) {
      const size_t li = ((size_t)(l >> 1) * n_users + u) * 2 + (l & 1);
      const int c = cand_cnt[li];
      const uint64_t *src = cand_buf + li * kCandCap;
      for (int j = lane; j < c; j += 64) stage[base + j] = src[j];
      base += c;
    }
    __builtin_amdgcn_wave_barrier();
    uint64_t k[kSelNR];
#pragma unroll
    for (int r = 0; r < kSelNR; ++r) k[r] = (lane + 64 * r < total) ? stage[lane + 64 * r] : 0ull;
    __builtin_amdgcn_wave_barrier();
    int count_ge = total;
    uint32_t T = 0;
    if (total > 128) {
      // K-th largest 32-bit score key (ties counted): bitwise search, counts via ballot + scalar popcount
      uint32_t prefix = 0;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t candv = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int r = 0; r < kSelNR; ++r) c += __popcll(__ballot((uint32_t)(k[r] >> 32) >= candv && k[r] != 0ull));
        if (c >= K) prefix = candv;
      }
      T = prefix;
      count_ge = 0;
#pragma unroll
      for (int r = 0; r < kSelNR; ++r) count_ge += __popcll(__ballot((uint32_t)(k[r] >> 32) >= T && k[r] != 0ull));
    }
    if (count_ge <= 128) {
      // compact the survivors into the first count_ge LDS slots (wave prefix popcounts), one sort orders them
      int base2 = 0;
#pragma unroll
      for (int r = 0; r < kSelNR; ++r) {
        const bool keep = k[r] != 0ull && (uint32_t)(k[r] >> 32) >= T;
        const unsigned long long m = __ballot(keep);
        const int pos = base2 + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (keep) stage[pos] = k[r];
        base2 += __popcll(m);
      }
      __builtin_amdgcn_wave_barrier();
      e0 = lane < count_ge ? stage[lane] : 0ull;
      e1 = lane + 64 < count_ge ? stage[lane + 64] : 0ull;
      sort128_desc(e0, e1, lane);
      sorted = true;
    }
  }
  if (!sorted) {
    // streaming path (more than 512 candidates, or more than 128 tied at the K-th score): running best-64 in
    // e0, 64 new keys per bitonic pass
    e0 = 0ull;
    e1 = 0ull;
    int fill = 0;
    for (int l = 0; l < n_lists; ++l) {
      const size_t li = ((size_t)(l >> 1) * n_users + u) * 2 + (l & 1);
      const int c = cand_cnt[li];
      const uint64_t *src = cand_buf + li * kCandCap;
      int done = 0;
      while (done < c) {
        const int take = min(c - done, 64 - fill);
        if (lane >= fill && lane < fill + take) e1 = src[done + lane - fill];
        fill += take;
        done += take;
        if (fill == 64) {
          sort128_desc(e0, e1, lane);
          e1 = 0ull;
          fill = 0;
        }
      }
    }
    if (fill > 0) sort128_desc(e0, e1, lane);
  }
  if (tau_out) {
    const uint64_t kth = shfl_u64(e0, K - 1);
    if (lane == 0) {
      // fewer than K candidates passed the incoming threshold: that threshold is already tight enough
      const float inc = tau ? tau[u] : -INFINITY;
      tau_out[u] = total >= K ? nextafterf(ord_to_f32((uint32_t)(kth >> 32)), -INFINITY) : inc;
    }
    return;
  }
  if (lane == 0) fail[u] = (total < K) ? 1 : 0;
  if (lane < K) {
    const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e0 & 0xFFFFFFFFull);
    out_idx[(size_t)u * K + lane] = (int64_t)item + id_offset;
    out_val[(size_t)u * K + lane] = ord_to_f32((uint32_t)(e0 >> 32));
  }
}

#include "score_prefilter.hpp"

// ---- launch plan -----------------------------------------------------------------------------
struct ScorePlan {
  int splits;
  int64_t tiles_per_split;
  bool pack;        // packed item table (register-resident variants only)
  bool sample;      // sampled per-user threshold + certification + fallback
  int sample_rank;  // r: tau0 = r-th best of the sample
  int tile_stride;
  // bf16 prefilter path (score_prefilter.hpp)
  bool prefilter;
  int pf_ub, pf_splits, pf_sample_stride, pf_sample_splits, pf_sample_rank;
  bool pf_sample_long;
  size_t off_pf_retry, off_pf_wide, off_pf_ncand, off_pf_fb, off_pf_fbdone, off_pf_fbpart, off_pf_inorm, pf_zero_bytes;
  bool pf_reth;                // pass C: raised thresholds for overflowing users (long item ranges)
  size_t off_pf_reth, off_pf_rt2;
  bool pf_cls;                 // norm-sorted packed table, the bound without its MFMA (score_prefilter.hpp "norm classes")
  int pf_cls_blocks;           // blocks of the counting sort (kClsChunk items each)
  size_t off_pf_perm, off_pf_inv, off_pf_tbound, off_pf_key, off_pf_bhist, off_pf_bmax, off_pf_segb, off_pf_spacked;
  int64_t pf_sample_tiles;     // tiles of the sampler's own table (every pf_sample_stride-th item of the sorted order)
  bool pf_group_fb;            // large item ranges: the first kPfFbGroupCap queued users share f32 MFMA sweeps
  int pf_group_fb_splits;
  size_t off_pf_fbgroup;
  size_t off_pf_packed, off_pf_scalars, off_pf_tau, off_pf_theta, off_pf_cand, off_pf_cnt;
  size_t off_packed, off_tau, off_tau1, off_fail, off_partial, off_cand, off_cnt, total;
};

#ifndef CHAOREC_PF_UB64
#define CHAOREC_PF_UB64 3
#endif
#ifndef CHAOREC_PF_UB128
#define CHAOREC_PF_UB128 2
#endif
// Workgroup slots of the sweep kernel THAT IS LAUNCHED on this device (workgroups per CU x CUs), queried once; without a
// device (the CPU-side workspace query) the gfx950 defaults.
static int sweep_wave_slots(int D, bool cls) {
  static int cached[4] = {0, 0, 0, 0};
  int &c = cached[(D == 64 ? 0 : 1) + (cls ? 2 : 0)];
  if (c) return c;
  int per_cu = 0, cus = 0, dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e == hipSuccess) {
    if (D == 64 && !cls) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, score_sweep_bf16_kernel<64, CHAOREC_PF_UB64, false>, 64 * kSweepWaves, 0);
    else if (D == 64) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, score_sweep_bf16_kernel<64, CHAOREC_PF_UB64, true>, 64 * kSweepWaves, 0);
    else if (!cls) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, score_sweep_bf16_kernel<128, CHAOREC_PF_UB128, false>, 64 * kSweepWaves, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, score_sweep_bf16_kernel<128, CHAOREC_PF_UB128, true>, 64 * kSweepWaves, 0);
  }
  if (e != hipSuccess || per_cu <= 0 || cus <= 0) {
    (void)hipGetLastError();
    return 1024;   // (4 workgroups x 256 CUs) not cached: a later call with a device asks again
  }
  c = per_cu * cus;      // workgroup slots (kSweepWaves waves each)
  return c;
}

// When the packed table is norm-sorted.  The layout takes one MFMA in nine (five) out of the sweep, ~6 % of its time, and
// costs (i) a sort in front of the pack (~0.1 ms + 0.6 ns per item: one workgroup scans the class histogram), (ii) a re-scale
// of the users' fragments per run of norm classes and walk (~3 us; a table whose norms span several octaves has 10 - 30
// runs).  So it pays for LONG walks of MANY users: from 524 288 items on (>= 1 024 tiles per workgroup walk) and
// 2 U I D >= 3e13 flops per call -- measured: 2 M items x 625 k users -11 % on the sweep, 262 144 x 131 072 -5 % on level
// norms and a loss on log-normal ones, 120 k x 29 k +50 % (tools/score_mid_sizes.py).  CHAOREC_PF_CLS_MIN_ITEMS overrides
// both conditions (tests run the sorted path on small tables with it; 0 switches it off); read per call.
static bool use_sorted_table(int64_t n_users, int64_t n_items, int D) {
  if (n_items > (int64_t)INT32_MAX - 64) return false;      // (positions of the sorted order are int32)
  const char *e = std::getenv("CHAOREC_PF_CLS_MIN_ITEMS");
  if (e && *e) {
    const int64_t v = std::atoll(e);
    return v > 0 && n_items >= v;
  }
  return n_items >= 524288 && (double)n_users * (double)n_items * (double)D >= 1.5e13;
}

static int env_int(const char *name, int dflt) {
  const char *e = std::getenv(name);
  return e && *e ? std::atoi(e) : dflt;
}

// The layout a workspace was FILLED with.  use_sorted_table() reads the environment per call, and a workspace's offsets
// (perm, inv, tile_bound, the scalars ...) depend on its answer: a BACK-phase call or a statistics query that re-planned
// under another environment than the FRONT / whole call that filled the workspace would read stale bytes as counters and
// lists.  So the filling call records its decision per workspace address (host side, bounded), and the calls
// that only READ a workspace take it from there.
static std::mutex g_layout_mu;
static std::unordered_map<const void *, int> g_layout_cls;
static void remember_layout(const void *ws, bool cls) {
  std::lock_guard<std::mutex> lk(g_layout_mu);
  if (g_layout_cls.size() > 4096) g_layout_cls.clear();
  g_layout_cls[ws] = cls ? 1 : 0;
}
static int recall_layout(const void *ws) {       // -1: unknown workspace
  std::lock_guard<std::mutex> lk(g_layout_mu);
  auto it = g_layout_cls.find(ws);
  return it == g_layout_cls.end() ? -1 : it->second;
}

// force_cls: -1 = decide (use_sorted_table), 0 / 1 = the layout a workspace already holds
static ScorePlan plan_score(int64_t n_users, int64_t n_items, int K, int D, int force_cls = -1) {
  ScorePlan p;
  const int64_t n_tiles = (n_items + 31) / 32;
  const int64_t groups = (n_users + 31) / 32;
  int64_t s = (2048 + groups - 1) / groups;  // aim for >= 2 waves per SIMD over 256 CUs
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  int64_t per = (n_tiles + s - 1) / s;
  const int64_t min_per = ((int64_t)(K > 256 ? K : 256) + 31) / 32;
  if (per < min_per) per = min_per;
  s = (n_tiles + per - 1) / per;
  if (s < 1) s = 1;
  p.splits = (int)s;
  p.tiles_per_split = per;
  const bool reg_variant = D <= 128;
  p.pack = reg_variant && n_users >= 64;
  // Sampling: tau0 = r-th best of every `stride`-th tile.  Expected items above tau0 over the full
  // range = r * stride; r * stride ~ 6K keeps P(fewer than K) below 1e-5 per user (relative spread
  // ~ 1/sqrt(r)), and those users are caught by the certification and re-run without a threshold.
  p.sample_rank = 32;
  p.tile_stride = (int)((6 * (int64_t)K + p.sample_rank - 1) / p.sample_rank);
  if (p.tile_stride < 2) p.tile_stride = 2;
  const int64_t sample_tiles = n_tiles / p.tile_stride;
  p.sample = reg_variant && K <= 64 && sample_tiles * 32 >= 8 * p.sample_rank && n_tiles >= 4 * p.tile_stride;
  if (p.sample) {
    // the candidate sweep keeps ~6K scores per user in 2*splits lists of kCandCap: aim for <= 40 expected per
    // list so the overflow prune stays a rarity
    int64_t want = (6 * (int64_t)K + 79) / 80;
    if (want > p.splits && n_tiles / want >= 8) {
      p.splits = (int)want;
      p.tiles_per_split = (n_tiles + want - 1) / want;
      p.splits = (int)((n_tiles + p.tiles_per_split - 1) / p.tiles_per_split);
    }
  }
  // bf16 prefilter + exact re-score: D in {64, 128}, enough tiles for the sampler's statistics
  p.prefilter = (D == 64 || D == 128) && K <= 64 && n_tiles >= 128 && n_tiles <= (int64_t)65535 * kPfMaxSplits;
  p.pf_ub = D == 64 ? CHAOREC_PF_UB64 : CHAOREC_PF_UB128;
  // very long item ranges: every 8th tile still samples >= 64 k items per user, and the sampler streams the packed
  // table once per 32 users
  p.pf_sample_stride = n_tiles > 32768 ? 16 : (n_tiles > 16384 ? 8 : 4);
  // a sampler wave's share of the sample fits its 24-slot lists up to ~16 k items; longer ranges take the
  // streaming-top-r instantiation (32 slots, fewer and longer waves)
  p.pf_sample_long = n_tiles > 512;
  p.pf_cls = p.prefilter && p.pf_sample_long && (force_cls >= 0 ? force_cls != 0 : use_sorted_table(n_users, n_items, D));
  // The sorted layout's sampler takes every stride-th ITEM of the sorted order, strata dealt evenly to its waves: a
  // systematic sample, stratified by norm -- half the sample gives the spread a sample of whole tiles had (config-5 shard,
  // propagated tables: every 32nd item at rank 4 = 0 of 1.25 M users with fewer than K candidates, 782 through pass C, call
  // 555 -> 534 ms against every 16th at rank 7; rank 3: 62 users on the exact routes).  CHAOREC_PF_CLS_STRIDE / _RANK override.
  if (p.pf_cls) p.pf_sample_stride = std::max(1, env_int("CHAOREC_PF_CLS_STRIDE", 2 * p.pf_sample_stride));
  p.pf_cls_blocks = (int)((n_tiles * 32 + kClsChunk - 1) / kClsChunk);
  {
    const int64_t ublocks = (groups + p.pf_ub * kSweepWaves - 1) / (p.pf_ub * kSweepWaves);   // workgroups per split
    // one full round of 2 waves per SIMD (2048 slots) when the user blocks allow it: a second, partly filled
    // round costs a whole wave time
    int64_t sp = sweep_wave_slots(D, p.pf_cls) / ublocks;
    // ~200 candidates per user (~300 with the coarser sample of very long item ranges) over 2 * splits lists of
    // kPfCap = 64 entries: keep the lists short
    const int64_t sp_min = n_tiles > 16384 ? 10 : 6;
    if (sp < sp_min) sp = sp_min;
    if (sp > kPfMaxSplits) sp = kPfMaxSplits;
    while (sp < kPfMaxSplits && (n_tiles + sp - 1) / sp > 65535) ++sp;   // (an entry holds a 16-bit tile sequence number)
    if (sp > n_tiles / 16) sp = n_tiles / 16;
    if (sp < 1) sp = 1;
    p.pf_splits = (int)sp;
  }
  p.pf_sample_splits = p.pf_sample_long ? 3 : 4;
  #ifndef CHAOREC_PF_RANK
#define CHAOREC_PF_RANK 8
#endif
  p.pf_sample_rank = p.pf_sample_long ? (p.pf_sample_stride >= 16 ? 7 : (p.pf_sample_stride == 8 ? 10 : 14)) : CHAOREC_PF_RANK;
  if (p.pf_cls) p.pf_sample_rank = std::max(1, env_int("CHAOREC_PF_CLS_RANK", p.pf_sample_stride >= 32 ? 4 : (p.pf_sample_stride >= 16 ? 6 : 8)));
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  p.off_pf_packed = take(p.prefilter ? (size_t)n_tiles * 64 * (size_t)(D / 16 + 1) * 16 : 0);
  p.off_pf_inorm = take(p.prefilter ? (size_t)n_tiles * 32 * 4 : 0);
  p.off_pf_perm = take(p.pf_cls ? (size_t)n_tiles * 32 * 4 : 0);
  p.off_pf_inv = take(p.pf_cls ? (size_t)n_items * 4 : 0);
  p.off_pf_tbound = take(p.pf_cls ? (size_t)n_tiles * 4 : 0);
  p.off_pf_key = take(p.pf_cls ? (size_t)n_tiles * 32 : 0);
  p.off_pf_bhist = take(p.pf_cls ? (size_t)256 * p.pf_cls_blocks * 4 : 0);
  p.off_pf_bmax = take(p.pf_cls ? (size_t)p.pf_cls_blocks * 4 : 0);
  p.off_pf_segb = take(p.pf_cls ? (size_t)256 * 4 : 0);
  p.pf_sample_tiles = (n_tiles + p.pf_sample_stride - 1) / p.pf_sample_stride;
  p.off_pf_spacked = take(p.pf_cls ? (size_t)p.pf_sample_tiles * 64 * (size_t)(D / 16) * 16 : 0);
  // scalars | tau_sum | fb_done sit back to back: cleared together by the pack launch
  p.off_pf_scalars = take(p.prefilter ? 256 : 0);
  p.off_pf_tau = take(p.prefilter ? (size_t)n_users * 4 : 0);
  p.off_pf_fbdone = take(p.prefilter ? (size_t)n_users * 4 : 0);
  p.pf_zero_bytes = o - p.off_pf_scalars;
  p.off_pf_retry = take(p.prefilter ? (size_t)n_users * 4 : 0);
  p.off_pf_wide = take(p.prefilter ? (size_t)n_users * 4 : 0);
  p.off_pf_ncand = take(p.prefilter ? (size_t)n_users * 4 : 0);
  p.off_pf_fb = take(p.prefilter ? (size_t)n_users * 4 : 0);
  // The exact route costs a pass over the whole item table per user: from 128 k items on, a user whose lists overflowed
  // (threshold too low) is first given a raised threshold and one more compact pass (pass C, score_rethreshold_kernel)
  p.pf_reth = p.prefilter && n_items >= 131072;
  p.off_pf_reth = take(p.pf_reth ? (size_t)n_users * 4 : 0);
  p.off_pf_rt2 = take(p.pf_reth ? (size_t)n_users * 4 : 0);
  p.off_pf_fbpart = take(p.prefilter ? (size_t)n_users * kExSlices * kMaxK * 8 : 0);
  // The per-user exact route streams the whole item table once per queued user (1 GB per user at 2 M x 128): past
  // 128 k items the queued users are instead swept 32 at a time on the f32 MFMA pipe (the route-1 kernel over a
  // compact user set), the item range cut into enough splits to fill the chip with 1..16 user groups
  p.pf_group_fb = p.prefilter && n_items >= 131072;
  p.pf_group_fb_splits = (int)std::min<int64_t>(256, std::max<int64_t>(1, n_tiles / 64));
  p.off_pf_fbgroup = take(p.pf_group_fb ? (size_t)p.pf_group_fb_splits * kPfFbGroupCap * (size_t)K * 8 : 0);
  p.off_pf_theta = take(p.prefilter ? (size_t)n_users * 4 : 0);
  // (sized for the most splits any device plan uses, so that the CPU-side query and the device plan agree)
  p.off_pf_cand = take(p.prefilter ? (size_t)kPfMaxSplits * (size_t)n_users * 2 * kPfCap * 4 : 0);
  p.off_pf_cnt = take(p.prefilter ? (size_t)kPfMaxSplits * (size_t)n_users * 2 * 4 : 0);
  p.off_packed = take(p.pack ? (size_t)n_tiles * 32 * (size_t)D * 4 : 0);
  p.off_tau = take(p.sample ? (size_t)n_users * 4 : 0);
  p.off_tau1 = take(p.sample ? (size_t)n_users * 4 : 0);
  p.off_fail = take((p.sample || p.prefilter) ? (size_t)n_users * 4 : 0);
  p.off_partial = take(p.splits > 1 ? (size_t)p.splits * (size_t)n_users * (size_t)K * 8 : 0);
  p.off_cand = take(p.sample ? (size_t)p.splits * (size_t)n_users * 2 * kCandCap * 8 : 0);
  p.off_cnt = take(p.sample ? (size_t)p.splits * (size_t)n_users * 2 * 4 : 0);
  p.total = o;
  return p;
}

template <int D>
static void launch_score(const ScoreArgs &a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL(score_topk_f32_kernel<D>, grid, dim3(64), 0, st, a);
}

static int dispatch_score(int D, const ScoreArgs &a, dim3 grid, hipStream_t st) {
  switch (D) {
    case 8: launch_score<8>(a, grid, st); break;
    case 16: launch_score<16>(a, grid, st); break;
    case 32: launch_score<32>(a, grid, st); break;
    case 64: launch_score<64>(a, grid, st); break;
    case 128: launch_score<128>(a, grid, st); break;
    default:
      if (D > 128 && (D % 64) == 0) {
        launch_score<0>(a, grid, st);
        break;
      }
      return fail(CHAOREC_E_INVALID, "score_topk: D=%d not in {8,16,32,64,128} and not a multiple of 64 above 128", D);
  }
  return check_launch("score_topk_f32_kernel");
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int chaorec_score_topk_f32(const float *user_emb, const float *item_emb, int64_t n_users,
                                      int64_t n_items, int32_t D, const int64_t *hist_rowptr,
                                      const int32_t *hist_col, float mask_value, int32_t K,
                                      int64_t id_offset, int64_t *out_idx, float *out_val,
                                      void *workspace, size_t workspace_bytes, int32_t precision,
                                      void *stream);

extern "C" size_t chaorec_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int32_t K, int32_t D) {
  if (n_users <= 0 || n_items <= 0 || K <= 0 || D <= 0) return 0;
  return plan_score(n_users, n_items, K, D).total;
}

extern "C" int chaorec_score_topk_stats(const void *workspace, int64_t n_users, int64_t n_items, int32_t K, int32_t D,
                                        uint64_t *out10, void *stream) {
  uint64_t *out9 = out10;
  if (!workspace || !out9) return fail(CHAOREC_E_INVALID, "score_topk_stats: NULL argument");
  if (n_users <= 0 || n_items <= 0 || K <= 0 || D <= 0) return fail(CHAOREC_E_INVALID, "score_topk_stats: bad sizes");
  const ScorePlan p = plan_score(n_users, n_items, K, D, recall_layout(workspace));   // (the layout the last call left there)
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out9, 0, 10 * sizeof(uint64_t), st) != hipSuccess) return fail(CHAOREC_E_LAUNCH, "stats: memset");
  if (!p.prefilter) return CHAOREC_OK;   // all zeros: the call did not take the prefilter route
  const char *ws = (const char *)workspace;
  hipLaunchKernelGGL(score_prefilter_stats_kernel, dim3((unsigned)((n_users + 255) / 256)), dim3(256), 0, st,
                     (const int *)(ws + p.off_fail), (const int *)(ws + p.off_pf_cnt),
                     (const int *)(ws + p.off_pf_ncand), n_users, p.pf_splits, (unsigned long long *)out9,
                     p.pf_reth ? (const int *)(ws + p.off_pf_scalars + 96) : (const int *)nullptr);
  return check_launch("score_prefilter_stats_kernel");
}

static int score_topk_impl(const float *user_emb, const float *item_emb, int64_t n_users,
                           int64_t n_items, int32_t D, const int64_t *hist_rowptr,
                           const int32_t *hist_col, float mask_value, int32_t K,
                           int64_t id_offset, int64_t *out_idx, float *out_val,
                           void *workspace, size_t workspace_bytes, int32_t precision,
                           const float *hint_in, float *hint_out, int32_t hint_rank, int32_t flags,
                           int32_t *counters_out, void *stream) {
  if (!user_emb || !item_emb || !out_idx || !out_val) return fail(CHAOREC_E_INVALID, "score_topk: NULL argument");
  if (n_users < 0 || n_items <= 0) return fail(CHAOREC_E_INVALID, "score_topk: bad sizes");
  if (K < 1 || K > kMaxK) return fail(CHAOREC_E_INVALID, "score_topk: K=%d must be in [1,%d]", K, kMaxK);
  if (n_items < K) return fail(CHAOREC_E_INVALID, "score_topk: n_items=%lld < K=%d (torch.topk would raise)", (long long)n_items, K);
  if (n_items > 0xFFFFFFF0ll) return fail(CHAOREC_E_INVALID, "score_topk: n_items too large");
  if (precision < 0 || precision > 2) return fail(CHAOREC_E_INVALID, "score_topk: precision %d not built", precision);
  if (hist_rowptr && !hist_col) return fail(CHAOREC_E_INVALID, "score_topk: hist_rowptr without hist_col");
  if (!((D == 8 || D == 16 || D == 32 || D == 64 || D == 128) || (D > 128 && D % 64 == 0)))
    return fail(CHAOREC_E_INVALID, "score_topk: D=%d not in {8,16,32,64,128} and not a multiple of 64 above 128", D);
  if (n_users == 0) return CHAOREC_OK;
  // a BACK-only call continues what a FRONT call began in this workspace: same layout, whatever the environment says now
  const bool back_only = (flags & CHAOREC_SCORE_BACK) && !(flags & CHAOREC_SCORE_FRONT);
  ScorePlan p = plan_score(n_users, n_items, K, D, back_only ? recall_layout(workspace) : -1);
  if (!back_only && workspace) remember_layout(workspace, p.pf_cls);
  if (precision == 1) p.sample = false;  // precision 1: single exact pass, no sampled threshold (A/B + tests)
  if (precision != 0) p.prefilter = false;  // precision 2: fp32 sweep with sampled thresholds (the pre-bf16 path)
  if (p.total > workspace_bytes || (p.total && !workspace))
    return fail(CHAOREC_E_WORKSPACE, "score_topk: workspace %zu < %zu", workspace_bytes, p.total);
  if (reinterpret_cast<uintptr_t>(workspace) & 255)
    return fail(CHAOREC_E_INVALID, "score_topk: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char *ws = (char *)workspace;
  const int64_t n_tiles = (n_items + 31) / 32;
  const unsigned groups = (unsigned)((n_users + 31) / 32);

  ScoreArgs a;
  a.user_emb = user_emb;
  a.item_emb = item_emb;
  a.packed = nullptr;
  a.n_users = n_users;
  a.n_items = n_items;
  a.Dk = D;
  a.hist_rowptr = hist_rowptr;
  a.hist_col = hist_col;
  a.mask_value = mask_value;
  a.K = K;
  a.id_offset = id_offset;
  a.out_idx = out_idx;
  a.out_val = out_val;
  a.partial = p.splits > 1 ? (uint64_t *)(ws + p.off_partial) : nullptr;
  a.splits = p.splits;
  a.tiles_per_split = p.tiles_per_split;
  a.mode = kModeMain;
  a.tile_stride = 1;
  a.max_tiles = 0;
  a.tau = nullptr;
  a.certify = nullptr;
  a.fail = nullptr;
  a.user_map = nullptr;
  a.n_users_dev = nullptr;

  int rc;
  // phases of one call split over two calls (same arguments, same workspace; a caller that ranks several user ranges can
  // run range k's back phase beside range k + 1's front phase on another stream): neither flag = the whole call
  const bool do_front = !(flags & CHAOREC_SCORE_BACK) || (flags & CHAOREC_SCORE_FRONT);
  const bool do_back = !(flags & CHAOREC_SCORE_FRONT) || (flags & CHAOREC_SCORE_BACK);
  if (p.prefilter) {
    PrefArgs P;
    P.user_emb = user_emb;
    P.item_emb = item_emb;
    P.packed = (const uint4 *)(ws + p.off_pf_packed);
    P.n_users = n_users;
    P.n_items = n_items;
    P.hist_rowptr = hist_rowptr;
    P.hist_col = hist_col;
    P.mask_value = mask_value;
    P.K = K;
    P.id_offset = id_offset;
    P.item_norm = (float *)(ws + p.off_pf_inorm);
    P.perm = p.pf_cls ? (const int32_t *)(ws + p.off_pf_perm) : nullptr;
    P.inv = p.pf_cls ? (const int32_t *)(ws + p.off_pf_inv) : nullptr;
    P.tile_bound = p.pf_cls ? (const float *)(ws + p.off_pf_tbound) : nullptr;
    P.sample_packed = p.pf_cls ? (const uint4 *)(ws + p.off_pf_spacked) : nullptr;
    P.n_sample_tiles = (int)p.pf_sample_tiles;
    P.sample_phase = p.pf_sample_stride / 2;
    P.retry_cnt = nullptr;
    P.retry_list = (int *)(ws + p.off_pf_retry);
    P.wide_cnt = nullptr;
    P.wide_list = (int *)(ws + p.off_pf_wide);
    P.n_cand = (int *)(ws + p.off_pf_ncand);
    P.fb_cnt = (int *)(ws + p.off_pf_scalars + 128);
    P.fb_list = (int *)(ws + p.off_pf_fb);
    P.fb_done = (int *)(ws + p.off_pf_fbdone);
    P.fb_partial = (uint64_t *)(ws + p.off_pf_fbpart);
    P.tau_sum = (float *)(ws + p.off_pf_tau);
    P.theta = (float *)(ws + p.off_pf_theta);
    P.cand = (uint32_t *)(ws + p.off_pf_cand);
    P.cand_cnt = (int *)(ws + p.off_pf_cnt);
    P.splits = p.pf_splits;
    P.xcd_group = env_int("CHAOREC_SWEEP_XCD", 1);
    P.sample_stride = p.pf_sample_stride;
    P.sample_splits = p.pf_sample_splits;
    P.sample_rank = p.pf_sample_rank;
    P.out_idx = out_idx;
    P.out_val = out_val;
    P.user_map = nullptr;
    P.n_active = nullptr;
    P.min_active = 0;
    P.small_retry = -1;
    P.retry_list_cnt = (const int *)(ws + p.off_pf_scalars + 64);
    P.counters_out = counters_out;
    P.hint_in = nullptr;
    P.hint_out = hint_out;
    P.hint_rank = hint_rank > K ? (hint_rank > 128 ? 128 : hint_rank) : K;
    int *failf = (int *)(ws + p.off_fail);
    P.fail = failf;
    P.reth_cnt = nullptr;
    P.reth_list = p.pf_reth ? (int *)(ws + p.off_pf_reth) : nullptr;
    P.rt2_cnt = p.pf_reth ? (int *)(ws + p.off_pf_scalars + 96) : nullptr;
    P.rt2_list = p.pf_reth ? (int *)(ws + p.off_pf_rt2) : nullptr;
    int *reth_cnt = (int *)(ws + p.off_pf_scalars + 32);
    int *retry_cnt = (int *)(ws + p.off_pf_scalars + 64);
    const int64_t nfrag = n_tiles * (D / 16) * 64;
    if (do_front && p.pf_cls) {
      // norm classes -> stable counting sort -> the pack of the sorted table (which also clears the call's counters)
      const int NB = p.pf_cls_blocks;
      const int64_t n_pad = n_tiles * 32;
      const int min_seg = (int)std::max<int64_t>(1, n_items / kClsSegments);
      hipLaunchKernelGGL(score_norm_class_kernel, dim3((unsigned)NB), dim3(256), 0, st, item_emb, n_items, (int)D, n_pad,
                         (uint8_t *)(ws + p.off_pf_key), (int *)(ws + p.off_pf_bhist), NB, (int *)(ws + p.off_pf_bmax));
      // (runs a walk can afford: one per ~128 tiles of a workgroup's walk, score_class_scan_kernel)
      const int max_runs = (int)std::min<int64_t>(kClsSegments, std::max<int64_t>(8, n_tiles / p.pf_splits / 128));
      hipLaunchKernelGGL(score_class_scan_kernel, dim3(1), dim3(1024), 0, st, (int *)(ws + p.off_pf_bhist), NB,
                         (const int *)(ws + p.off_pf_bmax), (float *)(ws + p.off_pf_segb), n_pad, min_seg, max_runs);
      hipLaunchKernelGGL(score_class_scatter_kernel, dim3((unsigned)NB), dim3(64), 0, st, (const uint8_t *)(ws + p.off_pf_key),
                         (const int *)(ws + p.off_pf_bhist), NB, n_pad, n_items, (int32_t *)(ws + p.off_pf_perm),
                         (int32_t *)(ws + p.off_pf_inv), (const float *)(ws + p.off_pf_segb), (float *)(ws + p.off_pf_tbound));
      const int64_t nfrag_s = (n_tiles + p.pf_sample_tiles) * (D / 16) * 64;
      hipLaunchKernelGGL(pack_items_bf16_sorted_kernel, dim3((unsigned)((nfrag_s + 255) / 256)), dim3(256), 0, st, item_emb,
                         (uint4 *)(ws + p.off_pf_packed), n_items, (int)D, n_tiles, P.perm,
                         (uint4 *)(ws + p.off_pf_spacked), p.pf_sample_tiles, p.pf_sample_stride, P.sample_phase,
                         (uint4 *)(ws + p.off_pf_scalars), (int64_t)(p.pf_zero_bytes / 16));
      rc = check_launch("pack_items_bf16_sorted_kernel");
      if (rc) return rc;
    } else if (do_front) {
      hipLaunchKernelGGL(pack_items_bf16_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, item_emb,
                         (uint4 *)(ws + p.off_pf_packed), n_items, (int)D, n_tiles, P.item_norm,
                         (uint4 *)(ws + p.off_pf_scalars), (int64_t)(p.pf_zero_bytes / 16));
      rc = check_launch("pack_items_bf16_kernel");
      if (rc) return rc;
    }
    const dim3 gs(groups, (unsigned)p.pf_sample_splits);
    const dim3 gs4((groups + 3) / 4, (unsigned)p.pf_sample_splits);
    const dim3 gw((unsigned)((groups + p.pf_ub * kSweepWaves - 1) / (p.pf_ub * kSweepWaves)), (unsigned)p.pf_splits);
    const unsigned sel_all = (unsigned)n_users;    // (a fixed grid walking the users is slower: 4 096 workgroups +10 %, 16 384 the same)
    const unsigned sel_queue = (unsigned)std::min<int64_t>(n_users, 8192);    // a pass over a device-side queue
    // (coarse samples of very long item ranges put a large part of the users above the narrow selection's 512 candidates:
    //  the wide selection is then a main pass, not a tail -- 256 one-wave workgroups took 106 ms per 1.1 M users at 2 M items)
    const unsigned sel_wide = p.pf_sample_stride >= 8 ? sel_queue : 256u;
    auto sweep = [&](const PrefArgs &A) {
      if (p.pf_cls) {
        if (D == 64) hipLaunchKernelGGL((score_sweep_bf16_kernel<64, CHAOREC_PF_UB64, true>), gw, dim3(64 * kSweepWaves), 0, st, A);
        else hipLaunchKernelGGL((score_sweep_bf16_kernel<128, CHAOREC_PF_UB128, true>), gw, dim3(64 * kSweepWaves), 0, st, A);
      } else {
        if (D == 64) hipLaunchKernelGGL((score_sweep_bf16_kernel<64, CHAOREC_PF_UB64, false>), gw, dim3(64 * kSweepWaves), 0, st, A);
        else hipLaunchKernelGGL((score_sweep_bf16_kernel<128, CHAOREC_PF_UB128, false>), gw, dim3(64 * kSweepWaves), 0, st, A);
      }
    };
    auto select = [&](const PrefArgs &A, unsigned grid) {
      if (D == 64) hipLaunchKernelGGL((score_select_kernel_pf<64, kPfMaxCand>), dim3(grid), dim3(64), 0, st, A);
      else hipLaunchKernelGGL((score_select_kernel_pf<128, kPfMaxCand>), dim3(grid), dim3(64), 0, st, A);
    };
    auto select_wide = [&](const PrefArgs &A, unsigned grid) {
      if (D == 64) hipLaunchKernelGGL((score_select_kernel_pf<64, kPfMaxCandWide>), dim3(grid), dim3(64), 0, st, A);
      else hipLaunchKernelGGL((score_select_kernel_pf<128, kPfMaxCandWide>), dim3(grid), dim3(64), 0, st, A);
    };
    auto sample = [&](const PrefArgs &A) {
      if (D == 64) {
        if (p.pf_cls) hipLaunchKernelGGL((score_sample_bf16_kernel<64, 32, true, 4, true>), gs4, dim3(256), 0, st, A);
        else if (p.pf_sample_long) hipLaunchKernelGGL((score_sample_bf16_kernel<64, 32, true, 4, false>), gs4, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((score_sample_bf16_kernel<64, 24, false, 1, false>), gs, dim3(64), 0, st, A);
      } else {
        if (p.pf_cls) hipLaunchKernelGGL((score_sample_bf16_kernel<128, 32, true, 4, true>), gs4, dim3(256), 0, st, A);
        else if (p.pf_sample_long) hipLaunchKernelGGL((score_sample_bf16_kernel<128, 32, true, 4, false>), gs4, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((score_sample_bf16_kernel<128, 24, false, 1, false>), gs, dim3(64), 0, st, A);
      }
    };
    // Pass A (carried thresholds) certifies nearly everybody in steady state.  What is left goes to pass B (sampled
    // thresholds over the device-side queue) -- unless it is a handful: then the exact per-user route is cheaper than
    // three more launches whose critical path is a whole item sweep by one wave, and pass B returns at once.  With
    // CHAOREC_SCORE_LIGHT the caller (who has seen the previous call's queue lengths) asks for no pass B at all.
    // Phases (CHAOREC_SCORE_FRONT / _BACK): the front is everything up to and including the call's FIRST whole sweep.
    const bool light = hint_in && (flags & CHAOREC_SCORE_LIGHT) && !p.pf_group_fb;
    const int small_queue = p.pf_group_fb ? -1 : 16;        // (the exact route costs ~6 us per user, a retry pass ~120 us)   // (very long item ranges: the per-user exact route streams the table per user)
    int *wide_cnt = (int *)(ws + p.off_pf_scalars + 192);
    if (hint_in) {
      PrefArgs A = P;            // pass A: the carried thresholds
      A.hint_in = hint_in;
      A.retry_cnt = retry_cnt;
      if (do_front) sweep(A);
      if (do_back) select(A, sel_all);
      P.small_retry = light ? INT_MAX : small_queue;
    }
    if (!light) {
      PrefArgs B = P;            // pass B: sampled thresholds; over the users pass A queued, or over everybody
      if (hint_in) {
        B.user_map = P.retry_list;
        B.n_active = retry_cnt;
        B.min_active = small_queue;
      }
      B.wide_cnt = wide_cnt;
      if (p.pf_reth) B.reth_cnt = reth_cnt;
      if (hint_in ? do_back : do_front) {
        sample(B);
        sweep(B);
      }
      if (do_back) {
        select(B, hint_in ? sel_queue : sel_all);
        // the users with more candidates than the narrow selection holds (coarse samples of very long item ranges)
        PrefArgs W = P;
        W.user_map = P.wide_list;
        W.n_active = wide_cnt;
        if (p.pf_reth) W.reth_cnt = reth_cnt;
        select_wide(W, sel_wide);
        if (p.pf_reth) {
          // pass C: lists that overflowed / more candidates than the wide selection holds = a threshold that was too low.
          // A raised one from the exact scores of the candidates the lists did keep, one more compact sweep, the wide
          // selection; what that cannot certify either goes to the exact routes below
          PrefArgs R = P;
          R.reth_cnt = reth_cnt;
          if (D == 64) hipLaunchKernelGGL(score_rethreshold_kernel<64>, dim3(1024), dim3(64), 0, st, R);
          else hipLaunchKernelGGL(score_rethreshold_kernel<128>, dim3(1024), dim3(64), 0, st, R);
          PrefArgs C = P;
          C.user_map = P.rt2_list;
          C.n_active = P.rt2_cnt;
          sweep(C);
          select_wide(C, 1024u);
        }
      }
      P.wide_cnt = wide_cnt;     // (for the counters the exact-route launch reports)
    }
    rc = check_launch("score prefilter kernels");
    if (rc) return rc;
    if (!do_back) return CHAOREC_OK;
    // uncertified users (list overflow, fewer than K above the threshold, band wider than the re-score slots) were
    // queued on the device.  Large item ranges: the first kPfFbGroupCap of them as a compact user set through the
    // unthresholded f32 MFMA sweep (32 users share each pass over the items), results written to their own rows.
    P.fb_skip = 0;
    if (p.pf_group_fb) {
      ScoreArgs f = a;
      f.n_users = kPfFbGroupCap;
      f.user_map = P.fb_list;
      f.n_users_dev = P.fb_cnt;
      f.splits = p.pf_group_fb_splits;
      f.tiles_per_split = (n_tiles + f.splits - 1) / f.splits;
      f.partial = (uint64_t *)(ws + p.off_pf_fbgroup);
      rc = dispatch_score(D, f, dim3(kPfFbGroupCap / 32, (unsigned)f.splits), st);
      if (rc) return rc;
      hipLaunchKernelGGL(score_topk_merge_kernel, dim3(kPfFbGroupCap), dim3(64), 0, st, f.partial,
                         (int64_t)kPfFbGroupCap, K, f.splits, id_offset, out_idx, out_val, (const float *)nullptr,
                         (int *)nullptr, (const int *)nullptr, (const int *)P.fb_list, (const int *)P.fb_cnt, hint_out);
      rc = check_launch("score_topk_merge_kernel (grouped fallback)");
      if (rc) return rc;
      P.fb_skip = kPfFbGroupCap;
    }
    // ... and exact fp32 scores of all items for each remaining one, one block per (user, item slice)
    // The queue is nearly always EMPTY (steady sports calls: no user at all) or a handful of users; the launch walks it
    // with a grid stride, so the grid only bounds how many (user, slice) pairs run at once.  Short item ranges: 16 users
    // x kExSlices -- 512 workgroups of 1024 threads (one per CU at a time: two dispatch rounds) cost 19 us on an empty
    // queue, as much as ranking 3 users.  Long ranges keep the wide grid: their queue is what 512 grouped users left over.
    const unsigned ex_grid = p.pf_group_fb ? (D == 64 ? 512u : 1024u) : 16u * kExSlices;
    if (D == 64) hipLaunchKernelGGL(score_exact_user_kernel<64>, dim3(ex_grid), dim3(ex_threads<64>()), 0, st, P);
    else hipLaunchKernelGGL(score_exact_user_kernel<128>, dim3(ex_grid), dim3(ex_threads<128>()), 0, st, P);
    return check_launch("score_exact_user_kernel");
  }
  if (!do_back) return CHAOREC_OK;   // (the other routes have no front phase: the back-phase call does everything)
  if (p.pack) {
    float4 *packed = (float4 *)(ws + p.off_packed);
    const int64_t n4 = n_tiles * (D / 8) * 64;
    hipLaunchKernelGGL(pack_items_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, item_emb,
                       packed, n_items, (int)D, n_tiles);
    rc = check_launch("pack_items_kernel");
    if (rc) return rc;
    a.packed = packed;
  }
  if (p.sample) {
    // sampled threshold -> sort-free candidate sweep -> exact selection + certification -> fallback
    float *tau = (float *)(ws + p.off_tau);
    int *failf = (int *)(ws + p.off_fail);
    uint64_t *cand_buf = (uint64_t *)(ws + p.off_cand);
    int *cand_cnt = (int *)(ws + p.off_cnt);
    // Sampling: two sort-free candidate sweeps + two selections instead of one prune-heavy streaming top-r:
    //  S1  candidate sweep of the first 4 sampled tiles with no threshold (128 scores per user);
    //      tau1 = their 6th best
    //  S2  candidate sweep of ALL sampled tiles against tau1 (~5 % pass), item range in 4 splits;
    //      tau0 = the r-th best of those
    float *tau1 = (float *)(ws + p.off_tau1);
    auto launch_cand = [&](const ScoreArgs &x, unsigned sp) -> int {
      const dim3 g(groups, sp);
      switch (D) {
        case 8: hipLaunchKernelGGL(score_candidates_kernel<8>, g, dim3(64), 0, st, x, cand_buf, cand_cnt); break;
        case 16: hipLaunchKernelGGL(score_candidates_kernel<16>, g, dim3(64), 0, st, x, cand_buf, cand_cnt); break;
        case 32: hipLaunchKernelGGL(score_candidates_kernel<32>, g, dim3(64), 0, st, x, cand_buf, cand_cnt); break;
        case 64: hipLaunchKernelGGL(score_candidates_kernel<64>, g, dim3(64), 0, st, x, cand_buf, cand_cnt); break;
        default: hipLaunchKernelGGL(score_candidates_kernel<128>, g, dim3(64), 0, st, x, cand_buf, cand_cnt); break;
      }
      return check_launch("score_candidates_kernel");
    };
    ScoreArgs s1 = a;
    s1.K = 6;
    s1.tile_stride = p.tile_stride;
    s1.max_tiles = 4;
    s1.tiles_per_split = 4;
    s1.tau = nullptr;
    rc = launch_cand(s1, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(score_select_kernel, dim3((unsigned)n_users), dim3(64), 0, st, cand_buf, cand_cnt, n_users, 6,
                       1, id_offset, (const float *)nullptr, out_idx, out_val, failf, tau1);
    rc = check_launch("score_select_kernel(S1)");
    if (rc) return rc;
    const int n_samp = (int)((n_tiles + p.tile_stride - 1) / p.tile_stride);
    const int samp_splits = (n_samp >= 16 && p.splits >= 4) ? 4 : 1;  // (the candidate region holds p.splits lists)
    ScoreArgs s2 = a;
    s2.K = p.sample_rank;
    s2.tile_stride = p.tile_stride;
    s2.tiles_per_split = (n_samp + samp_splits - 1) / samp_splits;
    s2.tau = tau1;
    rc = launch_cand(s2, (unsigned)samp_splits);
    if (rc) return rc;
    hipLaunchKernelGGL(score_select_kernel, dim3((unsigned)n_users), dim3(64), 0, st, cand_buf, cand_cnt, n_users,
                       p.sample_rank, samp_splits, id_offset, (const float *)tau1, out_idx, out_val, failf, tau);
    rc = check_launch("score_select_kernel(S2)");
    if (rc) return rc;
    a.tau = tau;
    rc = launch_cand(a, (unsigned)p.splits);
    if (rc) return rc;
    hipLaunchKernelGGL(score_select_kernel, dim3((unsigned)n_users), dim3(64), 0, st, cand_buf, cand_cnt, n_users,
                       K, p.splits, id_offset, (const float *)tau, out_idx, out_val, failf, (float *)nullptr);
    rc = check_launch("score_select_kernel");
    if (rc) return rc;
    // users with fewer than K scores above their sampled threshold (expected: a handful per million) re-run
    // unthresholded: their groups sweep the same item splits with the LDS-list kernel, then a merge
    ScoreArgs f = a;
    f.mode = kModeFallback;
    f.tau = nullptr;
    f.certify = nullptr;
    f.fail = failf;
    rc = dispatch_score(D, f, dim3(groups, (unsigned)p.splits), st);
    if (rc) return rc;
    if (p.splits > 1) {
      hipLaunchKernelGGL(score_topk_merge_kernel, dim3((unsigned)n_users), dim3(64), 0, st, a.partial, n_users, K,
                         p.splits, id_offset, out_idx, out_val, (const float *)nullptr, (int *)nullptr,
                         (const int *)failf, (const int *)nullptr, (const int *)nullptr);
      rc = check_launch("score_topk_merge_kernel");
    }
    return rc;
  }
  rc = dispatch_score(D, a, dim3(groups, (unsigned)p.splits), st);
  if (rc) return rc;
  if (p.splits > 1) {
    hipLaunchKernelGGL(score_topk_merge_kernel, dim3((unsigned)n_users), dim3(64), 0, st, a.partial, n_users,
                       K, p.splits, id_offset, out_idx, out_val, (const float *)nullptr, (int *)nullptr,
                       (const int *)nullptr, (const int *)nullptr, (const int *)nullptr);
    rc = check_launch("score_topk_merge_kernel");
    if (rc) return rc;
  }
  return CHAOREC_OK;
}

extern "C" int chaorec_score_topk_f32(const float *user_emb, const float *item_emb, int64_t n_users,
                                      int64_t n_items, int32_t D, const int64_t *hist_rowptr,
                                      const int32_t *hist_col, float mask_value, int32_t K,
                                      int64_t id_offset, int64_t *out_idx, float *out_val,
                                      void *workspace, size_t workspace_bytes, int32_t precision,
                                      void *stream) {
  return score_topk_impl(user_emb, item_emb, n_users, n_items, D, hist_rowptr, hist_col, mask_value, K, id_offset, out_idx,
                         out_val, workspace, workspace_bytes, precision, nullptr, nullptr, 0, 0, nullptr, stream);
}

extern "C" int chaorec_score_topk_hinted_f32(const float *user_emb, const float *item_emb, int64_t n_users,
                                             int64_t n_items, int32_t D, const int64_t *hist_rowptr,
                                             const int32_t *hist_col, float mask_value, int32_t K,
                                             int64_t id_offset, int64_t *out_idx, float *out_val,
                                             void *workspace, size_t workspace_bytes, const float *hint_in,
                                             float *hint_out, int32_t hint_rank, int32_t flags,
                                             int32_t *counters_out, void *stream) {
  return score_topk_impl(user_emb, item_emb, n_users, n_items, D, hist_rowptr, hist_col, mask_value, K, id_offset, out_idx,
                         out_val, workspace, workspace_bytes, 0, hint_in, hint_out, hint_rank, flags, counters_out, stream);
}
