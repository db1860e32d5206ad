// Block-joint selection for the bf16 prefilter (included by score_topk.hip after score_prefilter.hpp, inside namespace
// chaorec): the exact re-score of a user block's candidates on the f32 MFMA pipe.
//
// Round 2's selection ran one wave per USER: it expanded the user's sweep lists, gathered every candidate's 256-B row
// (0.8 GB of L2 gathers per sports call, 4 lanes per row) and walked the k-ordered fmaf chain on the VALU -- 80 us, bound
// by VALU issue and row gathers.  But the candidates of the 32 users of a sweep block overlap heavily (a trained table's
// top-100 slots concentrate on a few hundred items: tools/cand_popularity.py, tools/union_size.py), and
// v_mfma_f32_32x32x2_f32 IS the scoring contract's chain (oracle_score_dot).  So:
//
//   sweep    (score_sweep_bf16_kernel<D, UB, true>) no longer appends per-lane (tile, mask) entries to global lists:
//            per tile it ORs the 16-bit hit masks of a block's 32 users (5 DPP steps) and keeps the block's UNION
//            bitmap of its tiles in LDS, written out coalesced, 64 tiles at a time: [ublock][split][chunk][64] words.
//   select   (score_select_block_kernel<D>) one 4-wave workgroup per user block (all blocks resident at once):
//            1. the union bitmap -> a compact item list in LDS (popcounts, one block-wide prefix sum);
//            2. the list in tiles of 32 items: the wave that owns a tile gathers the 32 rows ONCE for all 32 users and
//               runs D/2 f32 MFMAs -- exact scores of 32 items x 32 users, the chain of every other route;
//               a score above the user's threshold T_u becomes a 64-bit key in the user's list (global memory, written
//               and read back by the same workgroup: L2-resident; LDS lists limited the occupancy).  Every item with
//               s > T_u is in the user's candidate set (score_prefilter.hpp's bound), hence in the union: the list
//               holds ALL of them;
//            3. per user: history members dropped, keys ranked by the bitonic networks of score_prefilter.hpp -- four
//               users' networks interleaved per wave, the chains are latency-bound --, the certification (>= K keys,
//               i.e. the K-th best > T_u), top-K out, next threshold out.
//   Items outside a user's own candidate set that sit in the union only cost MFMA time; their scores are <= T_u and never
//   become keys.  Any T_u is legal, as before; users that cannot be certified (fewer than K keys, more keys than the
//   list holds) are queued for the retry pass / the exact route exactly as the per-user selection queued them.
#pragma once

constexpr int kBsWaves = 4;            // waves per workgroup (one user block); 4 workgroups per CU, every block resident at once
constexpr int kBsList = 2048;          // union items held in LDS at a time (a longer union is processed in rounds)
#ifndef CHAOREC_BS_CAP
#define CHAOREC_BS_CAP 256
#endif
constexpr int kBsCap = CHAOREC_BS_CAP; // keys per user (global memory, L2-resident: written and read by the same workgroup)
#ifndef CHAOREC_BS_RANK_BATCH
#define CHAOREC_BS_RANK_BATCH 2
#endif
constexpr int kBsRankBatch = CHAOREC_BS_RANK_BATCH;   // users whose sorting networks one wave runs interleaved

// Words of the union bitmap per user block for any split count <= kPfMaxSplits (the workspace is sized for the worst case)
__host__ __device__ inline int64_t bs_words_per_block_max(int64_t n_tiles) { return n_tiles + 64 * (int64_t)kPfMaxSplits; }
__host__ __device__ inline int bs_chunks(int64_t n_tiles, int splits) {
  const int64_t n_mine = (n_tiles + splits - 1) / splits;     // tiles of split 0 (the longest)
  return (int)((n_mine + 63) / 64);
}

// ---- the sorting networks of score_prefilter.hpp over NB independent key sets at once ------------------------------
// One user's ranking is a chain of ~100 dependent compare-exchange stages, each waiting ~100 cycles for its two
// ds_bpermute: a wave that ranks its users one after the other spends its time in that latency (50 us of the first
// version's 125).  NB networks side by side issue NB times the permutes per wait.
template <int NB>
__device__ __forceinline__ void compare_exchange_b(uint64_t (&e)[NB], int addr, uint64_t keepmax) {
  uint64_t p[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) p[b] = permute64(e[b], addr);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const uint64_t gt = __ballot(p[b] > e[b]);
    e[b] = lane_select64(e[b], p[b], ~(gt ^ keepmax));
  }
}
template <int NB, int LOGJ>
__device__ __forceinline__ void merge_stages_b(uint64_t (&e)[NB], const PermAddr &pa) {
  compare_exchange_b<NB>(e, pa.template x<LOGJ>(), keepmax_mask(64, 1 << LOGJ, true));
  if constexpr (LOGJ > 0) merge_stages_b<NB, LOGJ - 1>(e, pa);
}
template <int NB, int LOGK, int LOGJ>
__device__ __forceinline__ void sort_stages_b(uint64_t (&e)[NB], const PermAddr &pa) {
  compare_exchange_b<NB>(e, pa.template x<LOGJ>(), keepmax_mask(1 << LOGK, 1 << LOGJ, LOGK == 6));
  if constexpr (LOGJ > 0) sort_stages_b<NB, LOGK, LOGJ - 1>(e, pa);
  else if constexpr (LOGK < 6) sort_stages_b<NB, LOGK + 1, LOGK>(e, pa);
}
template <int NB>
__device__ __forceinline__ void merge128_keys_b(uint64_t (&e0)[NB], uint64_t (&e1)[NB], const PermAddr &pa) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const uint64_t r = permute64(e1[b], pa.rev);
    const uint64_t gt = __ballot(r > e0[b]);
    const uint64_t hi = lane_select64(e0[b], r, gt), lo = lane_select64(r, e0[b], gt);
    e0[b] = hi;
    e1[b] = lo;
  }
  merge_stages_b<NB, 5>(e0, pa);
  merge_stages_b<NB, 5>(e1, pa);
}
// block number k (the same for the whole batch) of 64 more keys per set (0 = none) into the descending top-128 (e0, e1)
template <int NB>
__device__ __forceinline__ void take_block_keys_b(uint64_t (&e0)[NB], uint64_t (&e1)[NB], int k, uint64_t (&cur)[NB],
                                                  const PermAddr &pa) {
  sort_stages_b<NB, 1, 0>(cur, pa);                    // descending
  if (k == 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) e0[b] = cur[b];
  } else if (k == 1) {
#pragma unroll
    for (int b = 0; b < NB; ++b) e1[b] = cur[b];
    merge128_keys_b<NB>(e0, e1, pa);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {                     // descending vs ascending: the lane-wise maxima are the 64 largest, bitonic
      const uint64_t r = permute64(cur[b], pa.rev);
      e1[b] = lane_select64(e1[b], r, __ballot(r > e1[b]));
    }
    merge_stages_b<NB, 5>(e1, pa);
    merge128_keys_b<NB>(e0, e1, pa);
  }
}

template <int D>
__global__ __launch_bounds__(64 * kBsWaves, D <= 64 ? 4 : 2) void score_select_block_kernel(const PrefArgs P) {
  __shared__ uint32_t list_s[kBsList];
  __shared__ int kcnt_s[32];
  __shared__ uint32_t hist_all[kBsWaves][kBsRankBatch][kPfSelHist];
  __shared__ int wsum_s[kBsWaves];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ur = lane & 31, h = lane >> 5;
  const int cap = P.key_cap;
  const int K = P.K;
  const int S = P.splits;
  const int n_tiles = (int)((P.n_items + 31) / 32);
  const int n_chunks = P.bm_chunks;
  const int W = S * n_chunks * 64;                      // bitmap words of one user block
  const int wpt = (W + 64 * kBsWaves - 1) / (64 * kBsWaves);
  const int64_t n_act = P.n_active ? (int64_t)*P.n_active : P.n_users;
  if (n_act <= P.min_active) return;
  const PermAddr pa = perm_addr(lane);
#ifdef CHAOREC_BS_EXP      // stage cuts, switched at run time by hint_rank >= 1000 (tools/bs_variants.py; never in the product build)
  const int bs_exp = P.hint_rank >= 1000 ? P.hint_rank - 1000 : 0;
  if (bs_exp == 1) return;                               // the launch alone
#endif

#pragma unroll 1
  for (int64_t ublock = blockIdx.x; ublock * 32 < n_act; ublock += gridDim.x) {
    const int64_t uc = ublock * 32 + ur;
    const bool u_ok = uc < n_act;
    const int64_t u = (P.user_map && u_ok) ? (int64_t)P.user_map[uc] : uc;
    // users' B fragment: lane (ur, h) holds k = h * D/2 + s, the operand order of every f32 MFMA route
    float bu[D / 2];
    if (u_ok) {
      const float4 *src = reinterpret_cast<const float4 *>(P.user_emb + (size_t)u * D + h * (D / 2));
#pragma unroll
      for (int q = 0; q < D / 8; ++q) {
        const float4 v = src[q];
        bu[4 * q + 0] = v.x, bu[4 * q + 1] = v.y, bu[4 * q + 2] = v.z, bu[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < D / 2; ++s) bu[s] = 0.f;
    }
    const float theta = u_ok ? P.theta[u] : INFINITY;   // padding users never qualify
    uint64_t *keys_u = P.keys + (size_t)uc * cap;       // this lane's user's key list (rows of the launch)
    if (tid < 32) kcnt_s[tid] = 0;

    // ---- 1. union bitmap -> item list -----------------------------------------------------------------------
    // Word w of the block = (split w / (64 n_chunks), tile sequence number w % (64 n_chunks)); raw bit layout:
    // bit p (0..15) of half hh <=> accumulator register 15 - p of the lanes with h = hh <=> row (reg & 3) + 8 (reg >> 2) + 4 hh.
    const uint32_t *bm = P.bitmap + (size_t)ublock * (size_t)W;
    auto word_at = [&](int w, uint32_t &j0) -> uint32_t {
      const int split = w / (64 * n_chunks), seq = w % (64 * n_chunks);
      const int t = split + seq * S;
      if (w >= W || t >= n_tiles) return 0u;
      j0 = (uint32_t)t * 32u;
      uint32_t raw = bm[w];
      if (raw != 0u && j0 + 32u > (uint32_t)P.n_items) {   // the table's last tile: the sweep does not mask the rows past its end
#pragma unroll 1
        for (int b = 0; b < 32; ++b) {
          const int reg = 15 - (b & 15);
          if (j0 + (uint32_t)((reg & 3) + 8 * (reg >> 2) + 4 * (b >> 4)) >= (uint32_t)P.n_items) raw &= ~(1u << b);
        }
      }
      return raw;
    };
    int mine_cnt = 0;
    for (int k = 0; k < wpt; ++k) {
      uint32_t j0 = 0;
      mine_cnt += __popc(word_at(tid * wpt + k, j0));
    }
    const int incl = wave_scan_add(mine_cnt);
    if (lane == 63) wsum_s[wv] = incl;
    __syncthreads();
    int before = incl - mine_cnt, M = 0;
#pragma unroll
    for (int w2 = 0; w2 < kBsWaves; ++w2) {
      const int s = wsum_s[w2];
      before += w2 < wv ? s : 0;
      M += s;
    }
#ifdef CHAOREC_BS_EXP
    if (bs_exp == 2) {                                   // ... + the union's popcounts and prefix sum
      if (tid == 0) P.n_cand[u] = M;
      continue;
    }
#endif

#pragma unroll 1
    for (int lo = 0; lo < M; lo += kBsList) {
      const int n_here = min(M - lo, kBsList);
      // this thread's items whose position in the union falls into [lo, lo + kBsList)
      {
        int pos = before;
        for (int k = 0; k < wpt && pos < lo + kBsList; ++k) {
          uint32_t j0 = 0;
          uint32_t raw = word_at(tid * wpt + k, j0);
          while (raw) {
            const int b = __ffs(raw) - 1;
            raw &= raw - 1;
            if (pos >= lo && pos < lo + kBsList) {
              const int reg = 15 - (b & 15);
              list_s[pos - lo] = j0 + (uint32_t)((reg & 3) + 8 * (reg >> 2) + 4 * (b >> 4));
            }
            ++pos;
          }
        }
      }
      __syncthreads();

      // ---- 2. exact scores of the list x the block's 32 users on the f32 MFMA pipe ------------------------------
      // The next tile's rows are requested as soon as this tile's MFMAs are issued: they travel under the epilogue.
      float a[D / 2];
      auto load_tile = [&](int t) __attribute__((always_inline)) {
        const int idx = t * 32 + ur;
        const uint32_t item = list_s[idx < n_here ? idx : 0];
        const float4 *src = reinterpret_cast<const float4 *>(P.item_emb + (size_t)item * D + h * (D / 2));
#pragma unroll
        for (int q = 0; q < D / 8; ++q) {
          const float4 v = src[q];
          a[4 * q + 0] = v.x, a[4 * q + 1] = v.y, a[4 * q + 2] = v.z, a[4 * q + 3] = v.w;
        }
      };
      if (wv * 32 < n_here) load_tile(wv);
#pragma unroll 1
      for (int t = wv; t * 32 < n_here; t += kBsWaves) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < D / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bu[s], acc, 0, 0, 0);
        if ((t + kBsWaves) * 32 < n_here) load_tile(t + kBsWaves);
        // bit (15 - reg) <=> score > T_u: the sign of T_u - s (strict: an equal score is not above the threshold)
        uint32_t bits = 0;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(theta - acc[reg]), 31);
        if (t * 32 + 32 > n_here) {                         // the list's last, partial tile (wave-uniform)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            if (t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h >= n_here) bits &= ~(1u << (15 - reg));
        }
        if (__any(bits != 0u)) {
          int slot = 0;
          if (bits) slot = atomicAdd(&kcnt_s[ur], __popc(bits));
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            if ((bits >> (15 - reg)) & 1u) {
              if (slot < cap) keys_u[slot] = make_key(acc[reg], list_s[t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h]);
              ++slot;
            }
          }
        }
      }
      __syncthreads();
    }
#ifdef CHAOREC_BS_EXP
    if (bs_exp == 3) {                                   // ... + the item list and the MFMA re-score, no ranking
      if (tid < 32) P.n_cand[u] = kcnt_s[tid];
      __syncthreads();
      continue;
    }
#endif

    // ---- 3. per user: ranking, certification, outputs (the tail of select_user), kBsRankBatch users at a time ---------
    // (the key lists were written by this workgroup's waves before the barrier above: visible through the CU's
    //  write-through L1 / L2)
    constexpr int NB = kBsRankBatch;
#pragma unroll 1
    for (int q0 = 0; q0 < 32 / kBsWaves; q0 += NB) {
      const int ur0 = wv * (32 / kBsWaves) + q0;            // this wave's users: ur0 .. ur0 + NB - 1 (a contiguous run)
      if (ublock * 32 + ur0 >= n_act) break;                // wave-uniform
      int64_t u2[NB], hb[NB];
      int total[NB], n[NB], deg[NB], why[NB];
      bool hist_lds[NB];
      int kmax = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int64_t uc2 = ublock * 32 + ur0 + b;
        const bool ok2 = uc2 < n_act;
        u2[b] = ok2 ? (P.user_map ? (int64_t)P.user_map[uc2] : uc2) : -1;
        total[b] = ok2 ? kcnt_s[ur0 + b] : 0;
        why[b] = total[b] > cap ? 3 : 0;
        n[b] = why[b] ? 0 : total[b];
        hb[b] = 0;
        deg[b] = 0;
        if (ok2 && P.hist_rowptr) {
          hb[b] = P.hist_rowptr[u2[b]];
          deg[b] = (int)(P.hist_rowptr[u2[b] + 1] - hb[b]);
        }
        hist_lds[b] = deg[b] <= kPfSelHist;
        if (hist_lds[b]) {
#pragma unroll 1
          for (int i = lane; i < deg[b]; i += 64) hist_all[wv][b][i] = (uint32_t)P.hist_col[hb[b] + i];
        }
        kmax = max(kmax, (n[b] + 63) >> 6);
      }
      __builtin_amdgcn_wave_barrier();
      auto in_hist = [&](int b, uint32_t item) -> bool {
        int lo2 = 0, hi2 = deg[b];
        while (lo2 < hi2) {
          const int mid = (lo2 + hi2) >> 1;
          const uint32_t hv = hist_lds[b] ? hist_all[wv][b][mid] : (uint32_t)P.hist_col[hb[b] + mid];
          if (hv < item) lo2 = mid + 1; else hi2 = mid;
        }
        return lo2 < deg[b] && (hist_lds[b] ? hist_all[wv][b][lo2] : (uint32_t)P.hist_col[hb[b] + lo2]) == item;
      };
      const uint32_t mord = f32_to_ord(P.mask_value);
      uint64_t e0[NB], e1[NB];
      int valid[NB], above[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) e0[b] = e1[b] = 0ull, valid[b] = above[b] = 0;
#pragma unroll 1
      for (int k = 0; k < kmax; ++k) {
        uint64_t cur[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int i = k * 64 + lane;
          cur[b] = i < n[b] ? P.keys[(size_t)(ublock * 32 + ur0 + b) * cap + i] : 0ull;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          // (a history member leaves the keys: the sweep and the re-score ran unmasked)
          if (cur[b] != 0ull && deg[b] > 0 && in_hist(b, 0xFFFFFFFFu - (uint32_t)(cur[b] & 0xFFFFFFFFull))) cur[b] = 0ull;
          valid[b] += __popcll(__ballot(cur[b] != 0ull));
          above[b] += __popcll(__ballot(cur[b] != 0ull && (uint32_t)(cur[b] >> 32) > mord));
        }
        take_block_keys_b<NB>(e0, e1, k, cur, pa);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (u2[b] < 0) continue;                            // wave-uniform
        uint64_t f0 = e0[b], f1 = e1[b];
        int w = why[b];
        int n_keys = valid[b];
        if (w == 0) {
          // the user's history at mask_value joins where it can reach the top-K (quirk Q7), as in select_user
          if (deg[b] > 0 && above[b] < K) {   // wave-uniform, rare
            int blocks = kmax;
#pragma unroll 1
            for (int i0 = 0; i0 < deg[b]; i0 += 64) {
              const int i = i0 + lane;
              take_block_keys(f0, f1, blocks, i < deg[b] ? make_key(P.mask_value, hist_lds[b] ? hist_all[wv][b][i] : (uint32_t)P.hist_col[hb[b] + i]) : 0ull, pa);
            }
            n_keys = valid[b] + deg[b];
          }
          if (n_keys < K) w = 2;
        }
        if (w == 0) {
          const float theta2 = P.theta[u2[b]];
          const uint64_t kth = key_of_rank(f0, f1, K - 1);   // K <= 64
          if (kth == 0ull || !(ord_to_f32((uint32_t)(kth >> 32)) > theta2)) w = 4;
          if (w == 0 && P.hint_out) {
            // next call's threshold: one float below the exact score of rank `want` (>= K); extrapolated where fewer
            // keys than that lie above the current threshold (select_user has the reasoning)
            const int want = min(max(P.hint_rank, K), 128);
            const uint64_t hk = key_of_rank(f0, f1, min(want, n_keys) - 1);
            float tn = nextafterf(ord_to_f32((uint32_t)(hk >> 32)), -INFINITY);
            if (n_keys < want) {
              const float s_k = ord_to_f32((uint32_t)(kth >> 32)), s_last = ord_to_f32((uint32_t)(hk >> 32));
              const float s_top = ord_to_f32((uint32_t)(key_of_rank(f0, f1, 0) >> 32));
              const float slope = n_keys > K ? (s_k - s_last) / (float)(n_keys - K) : (s_top - s_k) / (float)max(K - 1, 1);
              tn -= slope * (float)(want - n_keys);
            }
            if (lane == 0) P.hint_out[u2[b]] = tn;
          }
        }
        if (lane == 0) {
          P.fail[u2[b]] = w;
          P.n_cand[u2[b]] = total[b];
          if (w != 0) {
            if (P.retry_cnt) P.retry_list[atomicAdd(P.retry_cnt, 1)] = (int)u2[b];
            else P.fb_list[atomicAdd(P.fb_cnt, 1)] = (int)u2[b];
          }
        }
        if (w == 0 && lane < K) {
          const uint32_t item = 0xFFFFFFFFu - (uint32_t)(f0 & 0xFFFFFFFFull);
          P.out_idx[(size_t)u2[b] * K + lane] = (int64_t)item + P.id_offset;
          P.out_val[(size_t)u2[b] * K + lane] = ord_to_f32((uint32_t)(f0 >> 32));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                        // the next block of this workgroup reuses the LDS
  }
}
