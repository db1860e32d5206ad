// bf16-MFMA prefilter for the all-items scoring (included by score_topk.hip, inside namespace chaorec).
//
// The exact fp32 sweep costs 32*D MFMA cycles per 32x32 tile (v_mfma_f32_32x32x2_f32); the bf16 pipe does the same
// tile in 2*D cycles (v_mfma_f32_32x32x16_bf16).  So the [U, I] sweep runs in bf16 and only decides WHICH items can
// be in a user's top-K; those are re-scored in exact fp32 (the same k-ascending fmaf chain as the f32 MFMA kernel /
// oracle_score_dot) and ranked on those values.  The result is bit-identical to the fp32 path:
//
//   |s~(u,j) - s(u,j)| <= e_uj := c * ||u||_2 * ||i_j||_2,   c = 1.05 * 2^-8
//     (two RNE bf16 roundings, 2^-9 relative each, + fp32 accumulation; Cauchy-Schwarz on sum_d |u_d||i_d|, PER ITEM:
//      a table with a few outsized item norms does not widen the band of every other item).
//   With e~_uj >= e_uj (bf16-rounded-up factors) v_j = s~_j + e~_uj is an upper bound of the exact score s_j.
//   The sweep keeps C = { j : v_j > T_u }: every item with s_j > T_u is in C.  The selection computes the EXACT score
//   of every member of C (history members carry mask_value instead) and takes the top-K of those.  If the K-th best
//   of them is > T_u, every item outside C has s_j <= v_j <= T_u < K-th best: it is strictly beaten by K items, so it
//   is not in the top-K whatever the tie order.  T_u may be ANY value: a good one keeps C small, a bad one fails the
//   certification and the user is retried / handed to the exact route; it never changes the result.
//   The threshold rides on the MFMA: the item side of an extra k-step carries (1, ||i_j||), the user side
//   (T_u, -c||u||) in the sweep -- with the users' fragments negated the accumulator is T_u - v_j, a hit is its sign
//   bit -- and (0, -c||u||) in the sampler (accumulator = the lower bound w_j = s~_j - e~_uj).
//
// Long item ranges of many users (score_topk.hip: use_sorted_table) run on a NORM-SORTED packed table instead, where the bound rides on the users' operand
// scale and costs no MFMA at all: section "norm classes" below.  Everything in this header up to there describes the table
// in its own order (sports: 15 k items); the certification argument is the same for both.
//
// Where T_u comes from:
//   hint     the caller's per-user thresholds from the PREVIOUS call on (nearly) the same tables -- the exact score
//            of rank hint_rank (> K) then.  One training epoch moves the scores little: ~1.7 K candidates per user
//            instead of ~3.5 K, and no sampling pass at all.  (chaorec_score_topk_hinted_f32)
//   sample   per user a threshold from every 4th tile (4 sample splits): the first call, and the RETRY pass for the
//            users whose hint failed (scores moved too much: fewer than K above T_u, or more than the selection holds).
//
// Pipeline (all on one stream, no host round trip; every pass after the first works on a device-side queue):
//   pack      items -> bf16 MFMA A-fragments (one coalesced 1 KiB wave load per k-step) + the bound fragment
//   pass A    (hint given)  sweep(T = hint) -> select: certified users are done, the others are queued
//   pass B    sample -> sweep(T = sample) -> select over the queue (all users when there is no hint)
//   tail      users pass B could not certify: exact fp32 scores of all items, per user (or grouped f32 MFMA sweeps for
//             very long item ranges)
//   sweep     one wave = UB x 32 users (fragments in registers, negated), tiles interleaved over the splits, item
//             fragments staged once per workgroup through LDS.  Per tile and user block: 16 v_alignbit collect the
//             sign bits; a lane with a hit appends ONE 32-bit entry (tile sequence number << 16 | hit bits) to its own
//             list -- no scores are stored, no LDS parking, no drain loop.
//   select    one wave per user: entries -> item ids (popcount / prefix sums), history members dropped, exact fp32
//             score per candidate (one candidate per lane, user row by scalar loads), register bitonic sort,
//             certification, top-K out, next call's hint out.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

union Frag16 {
  uint4 u;
  bf16x8 v;
};

__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
}
// smallest bf16-representable value >= x, for x >= 0 (as a float)
__device__ __forceinline__ float bf16_ceil_pos(float x) {
  const uint32_t b = __float_as_uint(x);
  return __uint_as_float(((b >> 16) + ((b & 0xFFFFu) ? 1u : 0u)) << 16);
}
__device__ __forceinline__ uint32_t bf16_pack2(float lo, float hi) {
  return bf16_rne_bits(lo) | (bf16_rne_bits(hi) << 16);
}
__device__ __forceinline__ uint4 bf16_pack8(const float4 a, const float4 b) {
  return make_uint4(bf16_pack2(a.x, a.y), bf16_pack2(a.z, a.w), bf16_pack2(b.x, b.y), bf16_pack2(b.z, b.w));
}

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ int dpp_or0(int v) {   // lanes without a source (or outside ROW_MASK) read 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, BOUND);
}

constexpr int kPfFbGroupCap = 512;   // queued users the grouped f32 fallback takes (16 groups of 32)
constexpr float kBf16ErrCoef = 1.05f / 256.0f;
constexpr int kPfCap = 64;           // 32-bit entries per (split, user, half) list of the sweep
constexpr int kPfMaxCand = 512;      // candidates per user the selection holds (8 per lane) ...
constexpr int kPfMaxCandWide = 1024; // ... and its second instantiation for the few users with more (long item ranges)
constexpr int kPfMaxSplits = 16;

struct PrefArgs {
  const float *user_emb;
  const float *item_emb;
  const uint4 *packed;          // [n_tiles][D/16 + 1][64] bf16x8 fragments
  int64_t n_users, n_items;
  const int64_t *hist_rowptr;
  const int32_t *hist_col;
  float mask_value;
  int K;
  int64_t id_offset;
  float *item_norm;             // [n_tiles * 32] ||i_j||_2 rounded UP to a bf16 value (0 for padding rows)
  float *tau_sum;               // [U] sampled threshold: mean over the sample splits (atomicAdd of est / splits)
  float *theta;                 // [U] the threshold the sweep used (written by sweep split 0)
  uint32_t *cand;               // [splits][U][2][kPfCap] entries: (tile sequence number << 16) | hit bits
  int *cand_cnt;                // [splits][U][2]
  int *n_cand;                  // [U] candidates the selection expanded (statistics)
  int splits;                   // sweep splits (tiles interleaved)
  int xcd_group;                // sweep: the workgroups of a split on ONE XCD (see score_sweep_bf16_kernel)
  int sample_stride;            // every sample_stride-th tile is sampled ...
  int sample_splits;            // ... dealt round-robin to this many sampler waves per user block
  int sample_rank;              // r_l: per-lane rank
  int64_t *out_idx;
  float *out_val;
  int *fail;
  // compact passes: row i of the launch is user user_map[i], only the first *n_active rows exist (NULL = identity / all)
  const int *user_map;
  const int *n_active;
  int min_active;               // a compact pass with at most this many rows does nothing (the exact route takes them)
  const float *hint_in;         // [U] thresholds carried from the previous call (pass A), else NULL: tau_sum
  float *hint_out;              // [U] thresholds for the next call, or NULL
  int hint_rank;                // the rank (> K, <= 128) whose exact score becomes the next threshold
  int *retry_cnt;               // users pass A could not certify -> pass B (NULL: uncertified users go to the exact route)
  int *retry_list;              // [U]
  int *wide_cnt;                // users with more candidates than the narrow selection holds -> the wide one
  int *wide_list;               // [U]
  int small_retry;              // exact route: also take the pass-A queue when it holds at most this many users
  const int *retry_list_cnt;    // (the queue's length for that; retry_cnt itself is NULL outside pass A)
  int *counters_out;            // [4] (optional) retry / exact-route / wide queue lengths of this call, for the caller
  int *fb_cnt;                  // users queued for the exact per-user route
  int *fb_list;                 // [U]
  int *fb_done;                 // [U] slices finished per queued user (zeroed per call)
  uint64_t *fb_partial;         // [U][kExSlices][kMaxK] per-slice best keys
  int fb_skip;                  // queue entries below this index were ranked by the grouped f32 sweep
  // pass C (long item ranges): a user whose lists overflowed / who has more candidates than the wide selection holds is
  // not ranked exactly at once (the exact route streams the whole item table per user) but gets a RAISED threshold taken
  // from the exact scores of the candidates its lists did keep (score_rethreshold_kernel) and one more compact pass
  int *reth_cnt;                // users queued for a raised threshold (NULL: such users go to the exact route)
  int *reth_list;               // [U]
  int *rt2_cnt;                 // users that got one -> compact sweep + wide selection (pass C)
  int *rt2_list;                // [U]
  // norm-sorted tiles (long item ranges, see "norm classes" below): the packed table holds the items in DESCENDING order of
  // their norm class, no bound fragment; NULL = the table's own order with the per-item bound on an extra MFMA k-step
  const int32_t *perm;          // [n_tiles * 32] packed position -> item (positions >= n_items: padding rows)
  const int32_t *inv;           // [n_items] item -> packed position
  const float *tile_bound;      // [n_tiles] N_t >= the norm of every item of tile t and of every later tile
  // ... and the sampler's own table: every sample_stride-th ITEM of the sorted order (not every sample_stride-th tile: in a
  // sorted table the items that can reach a user's top-K sit in the first few tiles, a sample of whole tiles sees all or
  // none of them), dealt to the sample tiles round-robin: sampled item q (sorted position q * sample_stride + sample_phase)
  // is stratum q / n_sample_tiles of tile q % n_sample_tiles -- EVERY tile is a systematic sample of the whole norm range,
  // so is any subset of tiles a sampler wave takes.  Stratum s of tile k sits in the row that accumulator register s / 2
  // of the user's lane (s + k) % 2 holds (row (i & 3) + 8 (i >> 2) + 4 h for i = s / 2, h = (s + k) % 2): over the tiles
  // both lanes of a user see every stratum equally often (the sampler pools the two lanes' lists and needs them alike:
  // with the top strata always in one lane its list alone decided the estimate, 0.7 of the rank aimed at)
  const uint4 *sample_packed;   // [n_sample_tiles][D/16][64]
  int n_sample_tiles, sample_phase;
};

// ---- pack ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_items_bf16_kernel(const float *__restrict__ item_emb,
                                                              uint4 *__restrict__ packed, int64_t n_items, int D,
                                                              int64_t n_tiles, float *__restrict__ item_norm,
                                                              uint4 *__restrict__ zero, int64_t zero_n) {
  // per tile Q = D/16 data fragments + ONE bound fragment: lane (r, h=0) holds k=0: 1.0, k=1: ||i_j|| (rounded up)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-element data fragment
  // the per-call counters (scalars | tau_sum | fb_done) are cleared by this launch too: no memset in the call, so
  // it can sit in a captured hipGraph (memset nodes were seen not to replay, see graph_dropout.hip)
  for (int64_t z = i; z < zero_n; z += (int64_t)gridDim.x * blockDim.x) zero[z] = make_uint4(0u, 0u, 0u, 0u);
  const int Q = D / 16;
  if (i >= n_tiles * Q * 64) return;
  const int lane = (int)(i & 63);
  const int64_t tq = i >> 6;
  const int q = (int)(tq % Q);
  const int64_t t = tq / Q;
  const int64_t j = t * 32 + (lane & 31);
  const int h = lane >> 5;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (j < n_items) {
    const float4 *src = reinterpret_cast<const float4 *>(item_emb + (size_t)j * D + 16 * q + 8 * h);
    v = bf16_pack8(src[0], src[1]);
  }
  packed[(t * (Q + 1) + q) * 64 + lane] = v;
  if (q == 0) {
    uint4 b = make_uint4(0u, 0u, 0u, 0u);
    if (h == 0) {
      float n2 = 0.f;
      if (j < n_items) {
        const float4 *row = reinterpret_cast<const float4 *>(item_emb + (size_t)j * D);
        for (int d = 0; d < D / 4; ++d) {
          const float4 x = row[d];
          n2 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
      }
      const float nu = bf16_ceil_pos(sqrtf(n2) * 1.0001f);   // (the fp32 sum / sqrt may round down: 1e-4 covers it)
      item_norm[j] = nu;
      b.x = 0x3F80u | ((__float_as_uint(nu) >> 16) << 16);
    }
    packed[(t * (Q + 1) + Q) * 64 + lane] = b;
  }
}

// ---- norm classes: the bound without its MFMA (long item ranges) ---------------------------------------------------
// The sweep's time is linear in its MFMA count, and one MFMA in nine (D = 128; one in five at D = 64) only
// adds the rank-1 term T_u - c ||u|| ||i_j||.  With ONE norm N for a whole run of tiles the test s~_j + c ||u|| N > T_u is
// s~_j > theta_u = T_u - c ||u|| N, and scaling the user's fragments by 1 / |theta_u| BEFORE the bf16 rounding turns it
// into s~'_j > +-1 -- 1.0 is an inline constant of the MFMA's C operand: no bound fragment, no extra k-step.  One N for the
// whole table (the largest norm) widens every item's band to the largest item's: +33 % candidates on propagated tables
// whose norms fall with the degree.  So the packed table is SORTED by norm class, descending -- a class =
// the norm's exponent and three mantissa bits: norms within 12.5 % of each other -- and the sweep re-scales its users'
// fragments whenever the walk enters a tile with another bound (tile_bound[t], non-increasing in t; classes with few
// items are merged so that a walk re-scales a few dozen times at most).  Candidate positions of the packed table go
// through perm[] in the selection; everything downstream (exact re-score, history, ranking keys) sees item ids.
//
//   |s~'(u,j) - a s(u,j)| <= c a ||u|| ||i_j||,  c = 1.05 * 2^-8  (two RNE bf16 roundings of 2^-9 each on a u_d and i_d:
//     2^-8 (1 + 2^-10) sum |a u_d i_d| <= 1.001 * 2^-8 a ||u|| ||i_j||; the fp32 rounding of a u_d (2^-24) and the MFMA's
//     fp32 accumulation ((D + 1) 2^-24 of the terms' absolute sum) disappear in c's 5 % of slack)
//   s_j > T_u  =>  a s_j > a T_u  =>  s~'_j > a (T_u - c ||u|| N) - d = a theta_u - d,  d <= (D + 1) 2^-24 (the
//     accumulation's rounding next to the constant 1); a = (1 + 2^-14) / theta_u makes that > 1 (theta_u > 0: fragments
//     negated, accumulator 1 - s~', a hit is its SIGN BIT); theta_u < 0: a = (1 - 2^-14) / |theta_u|, accumulator 1 + s~',
//     a hit is a CLEAR sign bit (one xor with the lane's flip mask).
// Sorting: a stable counting sort on the class (norm_class -> scan -> scatter), three small launches in front of the pack.
constexpr int kClsBase = 107 * 8;          // class 0 holds every norm below 2^-20, class 255 every norm from 2^11 * 1.875 on
constexpr int kClsChunk = 1024;            // items per block of the counting sort
constexpr int kClsSegments = 32;           // classes are merged into runs of at least n_items / kClsSegments items ...
constexpr int kClsRunSpan = 4;             // ... that span at most this many classes (half an octave) ...
constexpr int kClsRunSpanMax = 16;         // ... or, where the table would have more runs than its walks afford, up to two octaves

__device__ __forceinline__ int norm_class(float nu) {            // nu >= 0 (NaN / inf: the top class)
  const int k = (int)(__float_as_uint(nu) >> 20) - kClsBase;
  return min(max(k, 0), 255);
}
__device__ __forceinline__ float class_edge(int k) {             // > every norm of the classes <= k (k < 255)
  return __uint_as_float((uint32_t)(k + kClsBase + 1) << 20);
}

// key[j] = 255 - class of item j (ascending keys = descending norms; padding rows: norm 0, the last key, and -- their
// indices being the largest -- the last positions of the stable sort), per-block class histograms laid out class-major
// (a plain exclusive scan over the array gives every (class, block) its first position), per-block largest norm.
// 8 lanes per row: 128 contiguous bytes per load instruction and group.
__global__ __launch_bounds__(256) void score_norm_class_kernel(const float *__restrict__ item_emb, int64_t n_items, int D,
                                                               int64_t n_pad, uint8_t *__restrict__ key,
                                                               int *__restrict__ blockhist, int NB, int *__restrict__ blockmax) {
  __shared__ int hist[256];
  __shared__ int smax;
  const int tid = threadIdx.x;
  hist[tid] = 0;
  if (tid == 0) smax = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kClsChunk;
  const int g = tid >> 3, l = tid & 7;
#pragma unroll 1
  for (int pass = 0; pass < kClsChunk / 32; ++pass) {
    const int64_t j = base + pass * 32 + g;
    float n2 = 0.f;
    if (j < n_items) {
      const float4 *row = reinterpret_cast<const float4 *>(item_emb + (size_t)j * D);
      for (int d = l; d < D / 4; d += 8) {
        const float4 x = row[d];
        n2 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      }
    }
    n2 += __shfl_xor(n2, 1, 64);
    n2 += __shfl_xor(n2, 2, 64);
    n2 += __shfl_xor(n2, 4, 64);
    if (l == 0 && j < n_pad) {
      const float nu = j < n_items ? bf16_ceil_pos(sqrtf(n2) * 1.0001f) : 0.f;   // (the fp32 sum / sqrt may round down: 1e-4 covers it)
      const int kd = 255 - norm_class(nu);
      key[j] = (uint8_t)kd;
      atomicAdd(&hist[kd], 1);
      atomicMax(&smax, (int)(__float_as_uint(nu) & 0x7FFFFFFFu));
    }
  }
  __syncthreads();
  blockhist[(size_t)tid * NB + blockIdx.x] = hist[tid];
  if (tid == 0) blockmax[blockIdx.x] = smax;
}

// One workgroup: exclusive scan of blockhist[256 * NB] in place; seg_bound[kd] = the bound of the run of classes key kd
// belongs to.  Runs are formed from the largest norms down; a run is closed once it holds min_seg items.  The run that
// starts at key 0 (the open-ended top class) takes the table's largest norm itself.
__global__ __launch_bounds__(1024) void score_class_scan_kernel(int *__restrict__ blockhist, int NB, const int *__restrict__ blockmax,
                                                                float *__restrict__ seg_bound, int64_t n_pad, int min_seg,
                                                                int max_runs) {
  __shared__ int part[1024];
  __shared__ int cstart[257];
  __shared__ int gmax_s;
  const int tid = threadIdx.x;
  const int total = 256 * NB;
  const int per = (total + 1023) / 1024;
  const int lo = min(tid * per, total), hi = min(lo + per, total);
  int sum = 0;
#pragma unroll 8
  for (int i = lo; i < hi; ++i) sum += blockhist[i];      // (unrolled: eight loads in flight, not one round trip per counter)
  part[tid] = sum;
  if (tid == 0) gmax_s = 0;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan of the 1024 partial sums
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
#pragma unroll 8
  for (int i = lo; i < hi; ++i) {
    const int c = blockhist[i];
    blockhist[i] = run;
    run += c;
  }
  int m = 0;
  for (int i = tid; i < NB; i += 1024) m = max(m, blockmax[i]);
  atomicMax(&gmax_s, m);
  __syncthreads();
  if (tid < 256) cstart[tid] = blockhist[(size_t)tid * NB];
  if (tid == 0) cstart[256] = (int)n_pad;
  __syncthreads();
  if (tid == 0) {
    // A run also ends after `span` classes: a few outsized rows at the top of the table must not lend their bound to
    // thousands of ordinary ones.  span starts at kClsRunSpan (half an octave) and doubles, up to kClsRunSpanMax (two
    // octaves), while the table has more than max_runs runs: a re-scale costs a walk ~3 us (row, norm and threshold through
    // L2).  Runs of up to two octaves widen the band of their low-norm members 4x at most -- harmless; a run that reaches
    // from an outlier cluster 30x above the table down into it makes every ordinary item a candidate (measured: all users
    // overflow), so the span stops there and a table that still has more runs simply pays for them (the host only takes
    // this layout for long walks: plan_score).
    auto walk = [&](int span, bool write) -> int {
      float cur = 0.f;
      int acc = 0, first = 0, runs = 0;
      for (int kd = 0; kd < 256; ++kd) {
        const int cnt = cstart[kd + 1] - cstart[kd];
        if (acc > 0 && kd - first >= span) acc = 0;
        if (cnt > 0 && acc == 0) {
          cur = kd == 0 ? __int_as_float(gmax_s) : class_edge(255 - kd);
          first = kd;
          ++runs;
        }
        if (write) seg_bound[kd] = cur;
        acc += cnt;
        if (acc >= min_seg) acc = 0;
      }
      return runs;
    };
    int span = kClsRunSpan;
    while (span < kClsRunSpanMax && walk(span, false) > max_runs) span *= 2;
    walk(span, true);
  }
}

// One wave per kClsChunk items: position = first position of (class, block) + the number of earlier items of the block
// in the same class (stable: deterministic, lowest index first).  Per 64 items one round per distinct class among them.
__global__ __launch_bounds__(64) void score_class_scatter_kernel(const uint8_t *__restrict__ key, const int *__restrict__ offs, int NB,
                                                                 int64_t n_pad, int64_t n_items, int32_t *__restrict__ perm,
                                                                 int32_t *__restrict__ inv, const float *__restrict__ seg_bound,
                                                                 float *__restrict__ tile_bound) {
  __shared__ int cnt[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) cnt[i] = offs[(size_t)i * NB + blockIdx.x];
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * kClsChunk;
#pragma unroll 1
  for (int ch = 0; ch < kClsChunk / 64; ++ch) {
    const int64_t j = base + ch * 64 + lane;
    const bool valid = j < n_pad;
    const int kd = valid ? (int)key[j] : 256;
    int pos = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {                                               // wave-uniform
      const int leader = __ffsll((long long)todo) - 1;
      const int k0 = __shfl(kd, leader, 64);
      const unsigned long long m = __ballot(kd == k0);
      const int before = __popcll(m & ((1ull << lane) - 1ull));
      if (kd == k0) pos = cnt[k0] + before;
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) cnt[k0] += __popcll(m);
      __builtin_amdgcn_wave_barrier();
      todo &= ~m;
    }
    if (valid) {
      perm[pos] = (int32_t)j;
      if (j < n_items) inv[j] = pos;
      if ((pos & 31) == 0) tile_bound[pos >> 5] = seg_bound[kd];
    }
  }
}

// the pack of a norm-sorted table: row r of tile t is item perm[32 t + r]; D / 16 data fragments per tile, no bound
// fragment.  The same launch writes the sampler's table behind it (its tiles count on from n_tiles): row r of sample
// tile k is stratum s = 2 ((r & 3) + 4 (r >> 3)) + (((r >> 2) + k) & 1) (PrefArgs), the item at sorted position
// (s * n_stiles + k) * s_stride + s_phase (past the table: a zero row).
__global__ __launch_bounds__(256) void pack_items_bf16_sorted_kernel(const float *__restrict__ item_emb,
                                                                     uint4 *__restrict__ packed, int64_t n_items, int D,
                                                                     int64_t n_tiles, const int32_t *__restrict__ perm,
                                                                     uint4 *__restrict__ s_packed, int64_t n_stiles, int s_stride,
                                                                     int s_phase, uint4 *__restrict__ zero, int64_t zero_n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-element data fragment
  for (int64_t z = i; z < zero_n; z += (int64_t)gridDim.x * blockDim.x) zero[z] = make_uint4(0u, 0u, 0u, 0u);   // (as pack_items_bf16_kernel)
  const int Q = D / 16;
  if (i >= (n_tiles + n_stiles) * Q * 64) return;
  const int lane = (int)(i & 63);
  const int64_t tq = i >> 6;
  const int q = (int)(tq % Q);
  int64_t t = tq / Q;
  int64_t pos = t * 32 + (lane & 31);
  uint4 *dst = packed;
  if (t >= n_tiles) {
    t -= n_tiles;
    const int r = lane & 31;
    const int stratum = 2 * ((r & 3) + 4 * (r >> 3)) + (((r >> 2) + (int)t) & 1);
    pos = ((int64_t)stratum * n_stiles + t) * s_stride + s_phase;
    dst = s_packed;
  }
  const int64_t j = pos < n_tiles * 32 ? (int64_t)perm[pos] : n_items;
  const int h = lane >> 5;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (j < n_items) {
    const float4 *src = reinterpret_cast<const float4 *>(item_emb + (size_t)j * D + 16 * q + 8 * h);
    v = bf16_pack8(src[0], src[1]);
  }
  dst[(t * Q + q) * 64 + lane] = v;
}

// this lane's half of ||u||^2 (lane h of a user's two lanes: the elements 16 q + 8 h .. + 7 of every k-step)
template <int D>
__device__ __forceinline__ float user_half_norm2(const float *__restrict__ urow, int h) {
  float n2 = 0.f;
  if (urow) {
#pragma unroll
    for (int q = 0; q < D / 16; ++q) {
      const float4 *src = reinterpret_cast<const float4 *>(urow + 16 * q + 8 * h);
      const float4 a = src[0], b = src[1];
      n2 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    }
  }
  return n2;
}
// the lane's fragments of scale * u, rounded to bf16 (urow == NULL: zero fragments)
template <int D>
__device__ __forceinline__ void load_user_frags_scaled(bf16x8 (&bu)[D / 16], const float *__restrict__ urow, int h, float scale) {
#pragma unroll
  for (int q = 0; q < D / 16; ++q) {
    Frag16 f;
    f.u = make_uint4(0u, 0u, 0u, 0u);
    if (urow) {
      const float4 *src = reinterpret_cast<const float4 *>(urow + 16 * q + 8 * h);
      float4 a = src[0], b = src[1];
      a.x *= scale, a.y *= scale, a.z *= scale, a.w *= scale;
      b.x *= scale, b.y *= scale, b.z *= scale, b.w *= scale;
      f.u = bf16_pack8(a, b);
    }
    bu[q] = f.v;
  }
}
// where the accumulator of user block b starts: a constant of its own per block -- 1.0, 0.5, 2.0 are all inline constants of
// the MFMA's C operand; shared by two chains the compiler keeps the splat in 16 registers and rebuilds it every tile
template <int UB>
__device__ __forceinline__ constexpr float sweep_start(int b) {
  static_assert(UB <= 3, "one inline constant per user block");
  return b == 0 ? 1.0f : (b == 1 ? 0.5f : 2.0f);
}
// The signed scale of a user's fragments (negative = negated) and its flip mask, for threshold th, nu = ||u|| and item
// norms <= N.  Everything is rounded towards MORE hits.
__device__ __forceinline__ float sweep_user_scale(float th, float nu, float N, uint32_t &flip) {
  flip = 0u;
  if (th == INFINITY) return 0.f;                          // padding / +inf: zero fragments, accumulator 1: never a hit
  const float e_u = bf16_ceil_pos(kBf16ErrCoef * nu + 1e-30f) * N * 1.00001f;   // >= c ||u|| N
  const float theta = th - e_u - 1e-6f * fabsf(th);
  const float mag = fabsf(theta);
  if (mag >= 1e-30f && mag <= 1e30f && nu <= 1e30f * mag && nu >= 1e-20f * mag) {   // the scaled row stays in bf16's normal range
    const float inv = 1.0f / mag;
    if (theta > 0.f) return -inv * (1.0f + 0x1p-14f);      // s~ > theta  =>  s~' > 1:  a >= 1 / theta
    flip = 0xFFFFu;
    return inv * (1.0f - 0x1p-14f);                        // s~ > theta  =>  s~' > -1: a <= 1 / |theta|
  }
  // no usable scale.  A row too short to reach a positive theta (|s~| <= 1.01 nu N < theta) has no hit at all; otherwise
  // (NaN / -inf thresholds, theta ~ 0, a row that would leave the range) every item is a candidate: the lists overflow and
  // the user fails over, as with a NaN threshold
  if (theta > 0.f && mag <= 1e30f && N <= 1e19f && nu < 1e-20f * mag) return 0.f;
  flip = 0xFFFFu;
  return 0.f;
}

template <int D>
__device__ __forceinline__ void load_user_frags(bf16x8 (&bu)[D / 16], float &norm2_half, const float *__restrict__ user_emb,
                                                int64_t u, bool ok, int h) {
  norm2_half = 0.f;
#pragma unroll
  for (int q = 0; q < D / 16; ++q) {
    Frag16 f;
    f.u = make_uint4(0u, 0u, 0u, 0u);
    if (ok) {
      const float4 *src = reinterpret_cast<const float4 *>(user_emb + (size_t)u * D + 16 * q + 8 * h);
      const float4 a = src[0], b = src[1];
      f.u = bf16_pack8(a, b);
      norm2_half += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    }
    bu[q] = f.v;
  }
}

// FR fragments per tile: D / 16 data fragments, then (tables in their own order only) the bound fragment (1, ||i_j||)
template <int FR>
__device__ __forceinline__ void load_item_frags_bf16(uint4 (&a)[FR], const uint4 *__restrict__ packed, int64_t t, int lane) {
  const uint4 *src = packed + (size_t)t * FR * 64 + lane;
#pragma unroll
  for (int q = 0; q < FR; ++q) a[q] = src[q * 64];
}

template <int D, int FR>
__device__ __forceinline__ f32x16 tile_scores_bf16(const uint4 (&a)[FR], const bf16x8 (&bu)[D / 16]) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int q = 0; q < D / 16; ++q) {
    Frag16 f;
    f.u = a[q];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v, bu[q], acc, 0, 0, 0);
  }
  return acc;
}

// history cursor of one user over an ascending tile walk.  Two entries are kept in registers ahead of the walk,
// so stepping over an entry never waits on the load it issues (a vmcnt wait here also drains the item-fragment
// prefetch and every pending list store: it was half of the sweep's wall time).
struct HistCursor {
  int64_t hp, hend;       // hp: index of the entry after h1
  uint32_t h0, h1;        // next two entries, 0xFFFFFFFF = none
  __device__ __forceinline__ void init(const int64_t *rowptr, const int32_t *col, int64_t u, bool ok) {
    hp = hend = 0;
    h0 = h1 = 0xFFFFFFFFu;
    if (rowptr && ok) {
      hp = rowptr[u];
      hend = rowptr[u + 1];
      if (hp < hend) h0 = (uint32_t)col[hp];
      if (hp + 1 < hend) h1 = (uint32_t)col[hp + 1];
      hp += 2;
    }
  }
  // bit `off` set <=> item j0 + off is in the history; entries of skipped tiles are stepped over
  __device__ __forceinline__ uint32_t tile_bits(const int32_t *col, uint32_t j0) {
    uint32_t mbits = 0;
    while (h0 < j0 + 32u) {
      if (h0 >= j0) mbits |= 1u << (h0 - j0);
      h0 = h1;
      h1 = hp < hend ? (uint32_t)col[hp] : 0xFFFFFFFFu;
      ++hp;
    }
    return mbits;
  }
};

__device__ __forceinline__ void apply_mask_and_range(f32x16 &acc, uint32_t mbits, uint32_t j0, uint32_t n_items, int h,
                                                     float mask_value) {
  if (__any(mbits != 0)) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if ((mbits >> off) & 1u) acc[reg] = mask_value;
    }
  }
  if (j0 + 32u > n_items) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int off = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (j0 + off >= n_items) acc[reg] = -INFINITY;
    }
  }
}

// Streaming top-r of the sampler: every lane tightens its own threshold to a value that keeps `keep` of its list
// entries and compacts the list (register copy of the list: one LDS round trip, no dependent LDS loops).
template <int CAP>
__device__ __forceinline__ void sample_prune(float *lst, int lane, int &cnt, float &tl, int keep) {
  float e[CAP];
#pragma unroll
  for (int i = 0; i < CAP; ++i) e[i] = i < cnt ? lst[i * 64 + lane] : -INFINITY;
  if (cnt > keep) {
    float lo2 = tl, hi2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < CAP; ++i) hi2 = fmaxf(hi2, e[i]);
    for (int it = 0; it < 8; ++it) {
      const float mid = lo2 + (hi2 - lo2) * 0.5f;
      int c = 0;
#pragma unroll
      for (int i = 0; i < CAP; ++i) c += e[i] >= mid ? 1 : 0;
      if (c >= keep) lo2 = mid; else hi2 = mid;
    }
    int j = 0;
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (e[i] >= lo2) lst[(j++) * 64 + lane] = e[i];   // (-inf padding never passes: lo2 > -inf)
    cnt = j;
    tl = fmaxf(tl, nextafterf(lo2, -INFINITY));
  }
}

// ---- sample --------------------------------------------------------------------------------------------------
// grid (user blocks of 32, sample_splits).  Sample tile i of this wave is tile (i * sample_splits + split) * stride.
// Each wave estimates, per user, a score with about r sampled scores at or above it (r = sample_rank, pooled
// over the user's two lanes); the user's threshold is the mean of the sample splits' estimates.
// CAP / PRUNE: short item ranges (a wave's sample fits the list) run without the streaming-top-r code -- it doubles
// the kernel's size and costs more than it saves there; long ranges need it (see sample_prune).
// WG waves per workgroup (one user block each, same tiles, loosely in step through a barrier every 4 tiles): for
// item tables beyond L2 the waves of a workgroup then find each other's fragment loads in the CU's L1 instead of
// each streaming the table from HBM / Infinity Cache.  WG = 1 for tables that sit in L2 anyway.
// CLS: the packed table is norm-sorted (P.perm / P.inv, no bound fragment; long ranges only, hence PRUNE)
template <int D, int CAP, bool PRUNE, int WG, bool CLS = false>
__global__ __launch_bounds__(64 * WG) void score_sample_bf16_kernel(const PrefArgs P) {
  static_assert(!CLS || PRUNE, "sorted tables carry no bound fragment: the sample is taken on the plain bf16 scores");
  constexpr int FR = D / 16 + (CLS ? 0 : 1);
  __shared__ float lst_all[WG][CAP * 64];  // per wave [slot][lane]: conflict-free for lane-local walks
  __shared__ float park_all[WG][16 * 64];
  __shared__ int64_t rp_all[WG][33];
  __shared__ int hs_all[WG][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *lst = lst_all[wv], *park = park_all[wv];
  int64_t *rp_s = rp_all[wv];
  int *hs_s = hs_all[wv];
  const int ur = lane & 31, h = lane >> 5;
  const int64_t ublock = (int64_t)blockIdx.x * WG + wv;
  const int64_t n_act = P.n_active ? (int64_t)*P.n_active : P.n_users;
  if ((int64_t)blockIdx.x * WG * 32 >= n_act || n_act <= P.min_active) return;    // (only a compact pass has empty workgroups)
  const int64_t uc = ublock * 32 + ur;                    // row of this launch
  const bool u_ok = uc < n_act;
  const int64_t u = (P.user_map && u_ok) ? (int64_t)P.user_map[uc] : uc;   // row of the tables
  // What is sampled.  Tables in their own order: every sample_stride-th TILE of the packed table.  Norm-sorted tables
  // (CLS): every tile of the sampler's own table, which holds every sample_stride-th ITEM of the sorted order (PrefArgs).
  // Either way a wave sees 1 / (sample_stride * sample_splits) of the items.
  const uint4 *const tbl = CLS ? P.sample_packed : P.packed;
  const int tile_stride = CLS ? 1 : P.sample_stride;
  // (CLS: sampled items in all -- sample tile k holds the rows r with r * n_sample_tiles + k < n_items)
  const uint32_t n_items = CLS ? (uint32_t)((P.n_items - P.sample_phase + P.sample_stride - 1) / P.sample_stride) : (uint32_t)P.n_items;
  const int n_tiles = CLS ? P.n_sample_tiles : (int)((P.n_items + 31) / 32);
  // rows of tile t past the table get -inf (CLS: the strata s with s * n_tiles + t >= n_items; register reg of lane h
  // holds stratum 2 reg + (h + t) % 2)
  // (s * n_tiles + t <= n_items - 1  <=>  s <= (n_items - 1) / n_tiles - (t > (n_items - 1) % n_tiles): one division per wave)
  const int strata_q = CLS && n_items > 0 ? (int)((n_items - 1u) / (uint32_t)n_tiles) : -1;
  const int strata_r = CLS && n_items > 0 ? (int)((n_items - 1u) % (uint32_t)n_tiles) : 0;
  auto mask_range = [&](f32x16 &acc, int t) __attribute__((always_inline)) {
    if constexpr (CLS) {
      const int strata = min(strata_q + 1 - (t > strata_r ? 1 : 0), 32);
      if (strata < 32) {                                              // wave-uniform
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (2 * reg + ((h + t) & 1) >= strata) acc[reg] = -INFINITY;
      }
    } else {
      apply_mask_and_range(acc, 0u, (uint32_t)t * 32u, n_items, h, P.mask_value);
    }
  };
  const int split = blockIdx.y;
  const int step = tile_stride * P.sample_splits;
  const int t_first = split * tile_stride;
  // tile of the walk that holds history item cj, or -1 (CLS: the item's sorted position has to be a sampled one)
  auto hist_tile = [&](uint32_t cj) __attribute__((always_inline)) -> int {
    if constexpr (CLS) {
      const uint32_t pos = (uint32_t)P.inv[cj];
      if (pos % (uint32_t)P.sample_stride != (uint32_t)P.sample_phase) return -1;
      return (int)((pos / (uint32_t)P.sample_stride) % (uint32_t)n_tiles);
    } else {
      return (int)(cj >> 5);
    }
  };
  // The wave's sample: tiles t_first + i * step, i < n_samp, in that order (prefetches past the end wrap round to a tile
  // seen before)
  const int n_samp = t_first < n_tiles ? (n_tiles - t_first + step - 1) / step : 0;
  constexpr int pm = 1;
  auto walk_next = [&](int &i) __attribute__((always_inline)) {
    i += pm;
    if (i >= n_samp) i -= n_samp;
  };
  auto walk_tile = [&](int i) __attribute__((always_inline)) -> int { return t_first + i * step; };

  bf16x8 bu[D / 16];
  float n2;
  load_user_frags<D>(bu, n2, P.user_emb, u, u_ok, h);
  n2 += __shfl_xor(n2, 32, 64);
  // What the threshold is estimated on.  Short item ranges (no PRUNE: a few thousand tiles, rank target ~2.5 K): the lower
  // bounds w_j = s~_j - e~_uj -- the user side of the bound k-step is (0, -c||u||) against the items' (1, ||i_j||) -- whose
  // margin keeps users from ending with fewer than K candidates (trained sports tables, cold call: 1 user on the exact
  // route against 5 with the plain scores).  Long ranges (PRUNE; rank target >= 6 K): the bf16 scores s~_j themselves, one
  // MFMA per tile less -- there the lower bound put TWO error bands of candidates above the threshold (the sampler's and
  // the sweep's) where one is needed: configs[4] whole 462 -> 403 candidates per user, 6115 -> 778 users with overflowing
  // lists, the call 5.01 -> 4.83 s.
  Frag16 fw;
  fw.u = make_uint4(h == 0 ? (((__float_as_uint(bf16_ceil_pos(kBf16ErrCoef * sqrtf(n2) + 1e-30f)) >> 16) | 0x8000u) << 16) : 0u,
                    0u, 0u, 0u);
  const bf16x8 bw = fw.v;
  auto tile_w = [&](const uint4 (&a)[FR]) __attribute__((always_inline)) -> f32x16 {
    f32x16 acc = tile_scores_bf16<D, FR>(a, bu);
    if constexpr (PRUNE) {
      return acc;
    } else {
      Frag16 fa;
      fa.u = a[D / 16];
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, bw, acc, 0, 0, 0);
    }
  };

  // The sample is taken on RAW scores (no history mask: a cursor over the history costs a dependent load in the
  // tile loop).  Interacted items usually score high, so the rank is shifted by h_s = the number of the user's
  // history items inside this wave's sampled tiles: the (r + h_s)-th best raw score is at most the r-th best
  // masked one whenever mask_value does not matter, and any value is a valid threshold anyway (certification).
  // (counted by the whole wave over the block's contiguous CSR range: a per-lane walk of the own row is a chain of
  //  dependent loads as long as the heaviest user's history)
  int h_s = 0;
  if (P.hist_rowptr && P.user_map) {
    // compact pass: the block's users are scattered over the CSR; every lane walks its own user's row (both lanes of
    // a user do the same walk: the pass is small, simplicity over speed)
    if (u_ok) {
      for (int64_t e = P.hist_rowptr[u]; e < P.hist_rowptr[u + 1]; ++e) {
        const int tt = hist_tile((uint32_t)P.hist_col[e]);
        if (tt >= t_first && (tt - t_first) % step == 0) ++h_s;
      }
    }
  } else if (P.hist_rowptr) {
    const int64_t ub = ublock * 32;
    if (lane < 33) rp_s[lane] = P.hist_rowptr[min(ub + lane, P.n_users)];
    if (lane < 32) hs_s[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t e0 = rp_s[0], e1 = rp_s[32];
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const int tt = hist_tile((uint32_t)P.hist_col[e]);
      if (tt >= t_first && (tt - t_first) % step == 0) {
        int lo = 0, hi = 31;   // owner row: last r with rp_s[r] <= e
#pragma unroll
        for (int st = 0; st < 5; ++st) {
          const int mid = (lo + hi + 1) >> 1;
          if (rp_s[mid] <= e) lo = mid; else hi = mid - 1;
        }
        atomicAdd(&hs_s[lo], 1);
      }
    }
    __builtin_amdgcn_wave_barrier();
    h_s = hs_s[ur];
  }

  // phase 1: the first 8 sampled tiles (128 scores per lane), lane-local top-4 by median-of-3 insertion; tau1 = the
  // larger of the two lanes' 4th best: at least 4 of the user's 256 scores reach it (expected: the top ~2.5 %)
  float tau1;
  float b0 = -INFINITY, b1 = -INFINITY, b2 = -INFINITY, b3 = -INFINITY;
  int wi = 0, visited = 0;        // sample index of the next visit, visits so far
  {
    uint4 a[FR], an[FR];
    if (n_samp > 0) load_item_frags_bf16<FR>(a, tbl, walk_tile(wi), lane);
    for (int g = 0; g < 8 && visited < n_samp; ++g) {
      const int t = walk_tile(wi);
      walk_next(wi);
      ++visited;
      load_item_frags_bf16<FR>(an, tbl, walk_tile(wi), lane);   // (after the last visit: a tile seen before, never consumed)
      f32x16 acc = tile_w(a);
      mask_range(acc, t);
#pragma unroll
      for (int q = 0; q < FR; ++q) a[q] = an[q];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float x = acc[reg];
        b3 = __builtin_amdgcn_fmed3f(b2, b3, x);
        b2 = __builtin_amdgcn_fmed3f(b1, b2, x);
        b1 = __builtin_amdgcn_fmed3f(b0, b1, x);
        b0 = fmaxf(b0, x);
      }
    }
    tau1 = fmaxf(b3, __shfl_xor(b3, 32, 64));
  }

  // phase 2: every sampled tile; scores above the lane's threshold go to its LDS list.  Same compact hit handling as
  // the sweep: bit mask per lane, scores parked in LDS, one drain loop.  When a list could overflow on the next tile,
  // every lane tightens its own threshold to a value that keeps `keep` of its entries (lane-local bisection) and
  // compacts: a streaming top-r, so the sample may be arbitrarily long (millions of items) with 32 slots per lane.
  int cnt = 0;
  const int r = P.sample_rank + h_s;
  const int keep = min(r, CAP / 4);   // well below the trigger level: a prune buys room for many tiles
  float tl = nextafterf(tau1, -INFINITY);   // strict compare below keeps scores == tau1
  // the phase-1 tiles are not visited again: every score of theirs that reaches tau1 is one of the lane's top 4
  // (tau1 >= the lane's own 4th best), so the list starts with those
  if (b0 >= tau1) lst[(cnt++) * 64 + lane] = b0;
  if (b1 >= tau1) lst[(cnt++) * 64 + lane] = b1;
  if (b2 >= tau1) lst[(cnt++) * 64 + lane] = b2;
  if (b3 >= tau1) lst[(cnt++) * 64 + lane] = b3;
  {
    // item fragments three tiles ahead: one tile of this kernel is short (one user block), a single tile of
    // lookahead does not cover the L2 latency
    uint4 ring[3][FR];
    int pi = wi;                  // sample index the next prefetch takes: three visits ahead
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (n_samp > 0) load_item_frags_bf16<FR>(ring[i], tbl, walk_tile(pi), lane);
      walk_next(pi);
    }
    auto consume = [&](const uint4 (&a)[FR], int tt) __attribute__((always_inline)) {
      f32x16 acc = tile_w(a);
      mask_range(acc, tt);
      uint32_t qbits = 0;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) qbits = __builtin_amdgcn_alignbit(qbits, __float_as_uint(tl - acc[reg]), 31);
      if (__any(qbits != 0)) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) park[reg * 64 + lane] = acc[reg];
        __builtin_amdgcn_wave_barrier();
        while (__any(qbits != 0)) {
          if (qbits) {
            const int bit = 31 - __clz(qbits);
            qbits &= ~(1u << bit);
            const float sc = park[(15 - bit) * 64 + lane];
            if (cnt < CAP) lst[(cnt++) * 64 + lane] = sc;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (PRUNE && __any(cnt > CAP - 16)) sample_prune<CAP>(lst, lane, cnt, tl, keep);
      }
    };
    int since_sync = 0;
    while (visited < n_samp) {
      if (WG > 1 && ++since_sync == 2) {   // (every 6 tiles: keeps the workgroup's waves within L1 reach of each other)
        since_sync = 0;
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (visited < n_samp) {
          uint4 cur[FR];
#pragma unroll
          for (int q = 0; q < FR; ++q) cur[q] = ring[i][q];
          const int t = walk_tile(wi);
          walk_next(wi);
          ++visited;
          load_item_frags_bf16<FR>(ring[i], tbl, walk_tile(pi), lane);   // (past the end: a tile seen before, never consumed)
          walk_next(pi);
          consume(cur, t);
        }
      }
    }
  }

  // phase 3: bisection on the value, counts pooled over the user's two lanes: the largest of 2^8 grid values in
  // [lo, max] with at least r list entries at or above it (any such value is a valid lower bound of the r-th best).
  // lo starts at the larger of the two lanes' thresholds: above it both lists are complete.
  float lo = fmaxf(tl, __shfl_xor(tl, 32, 64));
  int pooled = 0;
  for (int i = 0; i < cnt; ++i) pooled += lst[i * 64 + lane] >= lo ? 1 : 0;
  pooled += __shfl_xor(pooled, 32, 64);
  if (pooled >= r && lo > -INFINITY) {
    float hi = -INFINITY;
    for (int i = 0; i < cnt; ++i) hi = fmaxf(hi, lst[i * 64 + lane]);
    hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
    const int nmax = max(cnt, __shfl_xor(cnt, 32, 64));
    for (int it = 0; it < 8; ++it) {
      const float mid = lo + (hi - lo) * 0.5f;
      int c = 0;
      for (int i = 0; i < nmax; ++i) c += (i < cnt && lst[i * 64 + lane] >= mid) ? 1 : 0;
      c += __shfl_xor(c, 32, 64);
      if (c >= r) lo = mid; else hi = mid;
    }
  }
  // one float below, so that the sweep's strict compare keeps equal scores; the user's threshold is the mean of
  // the splits' estimates (two commutative float adds: deterministic)
  if (u_ok && h == 0) atomicAdd(P.tau_sum + u, nextafterf(lo, -INFINITY) * (1.0f / (float)P.sample_splits));
}

// ---- sweep ---------------------------------------------------------------------------------------------------
// No history handling here: an interacted item is swept like any other; the selection drops history members from the
// candidates and ranks the user's history with mask_value where that can matter.
//
// The threshold compare rides on the MFMA: the users' fragments are stored NEGATED and an extra k-step multiplies the
// items' (1, ||i_j||) with the users' (T_u, -c||u||) (bf16, T_u rounded toward -inf, the norms rounded up), so the
// accumulator holds T_u - v_j, v_j = s~_j + e~_uj the upper bound of the score, and a hit is its SIGN BIT: one
// v_alignbit per accumulator register builds the lane's 16-bit hit mask.  A lane with a hit appends one 32-bit entry
// (this split's tile sequence number << 16 | mask) to its own global list; the exact scores are computed by the
// selection, so nothing else leaves the sweep: ~24 VALU instructions per tile and user block next to 5 MFMAs.
__device__ __forceinline__ float bf16_floor(float x) {  // largest bf16-representable value <= x
  const uint32_t b = __float_as_uint(x);
  uint32_t t = b & 0xFFFF0000u;
  if ((b & 0x80000000u) && (b & 0xFFFFu)) t += 0x10000u;
  return __uint_as_float(t);
}

// A workgroup is kSweepWaves waves = kSweepWaves x UB x 32 users that walk the SAME tiles: every item fragment is
// fetched from L2/HBM once per workgroup into an LDS stage (kSweepStage tiles per barrier, double-buffered) and read
// from there by all its waves -- with one wave per workgroup the sweep streamed the whole packed item table once per
// 64 users (1.1 GB per sports sweep, 11 TB at 1.25 M users x 2 M items).
constexpr int kSweepWaves = 4;
#ifndef CHAOREC_SWEEP_STAGE
#define CHAOREC_SWEEP_STAGE 2
#endif
constexpr int kSweepStage = CHAOREC_SWEEP_STAGE;
#ifndef CHAOREC_SWEEP_PARK
#define CHAOREC_SWEEP_PARK 8
#endif
constexpr int kSweepPark = CHAOREC_SWEEP_PARK;    // list entries per lane and user block parked in LDS (0: store each at once)
static_assert(kSweepPark % 4 == 0 && kSweepPark <= kPfCap, "whole 16-byte stores inside a list");

// CLS: norm-sorted table ("norm classes" above): D / 16 MFMAs per tile and user block, the accumulators start at an inline
// constant, the users' fragments are re-scaled when the walk reaches a tile with another bound.
template <int D, int UB, bool CLS = false>
__global__ __launch_bounds__(64 * kSweepWaves) void score_sweep_bf16_kernel(const PrefArgs P) {
  constexpr int FR = D / 16 + (CLS ? 0 : 1);           // fragments (1 KiB each) per tile
  __shared__ uint4 stage[2][kSweepStage][FR * 64];
  // A lane's first kSweepPark list entries wait here ([entry][lane]: conflict-free) and leave as 16-byte stores after the
  // sweep.  Every entry used to be its own 4-byte store into its own cache line -- 2.9 M separate L2 write requests per
  // sports sweep (~100 entries per user): 14 of the kernel's 77 us (stage cut: hits counted but never stored).  A list
  // gets ~5 entries per split: now one or two requests instead of five.
  __shared__ uint32_t park_s[kSweepWaves][UB][kSweepPark > 0 ? kSweepPark : 1][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ur = lane & 31, h = lane >> 5;
  const int64_t n_act = P.n_active ? (int64_t)*P.n_active : P.n_users;
  // Which (user workgroup, split) this workgroup is.  In dispatch order (user workgroup fastest) every XCD gets an eighth of the
  // user workgroups of EVERY split and so streams the whole packed table through its L2; with xcd_group the dispatch ids are
  // regrouped (d -> (d % 8) * (n / 8) + d / 8, bijective form) so that an XCD holds a contiguous run of the split-major list:
  // an eighth of the splits, an eighth of the table.  The dispatcher's round-robin is observed, not promised: speed only.
  int64_t bx = blockIdx.x;
  int split_ = blockIdx.y;
  if (P.xcd_group) {
    const int64_t gx = gridDim.x, nwg = gx * (int64_t)gridDim.y, q = nwg >> 3, rr = nwg & 7;
    int64_t wid = bx + gx * split_;
    const int64_t x = wid & 7;
    wid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (wid >> 3);
    split_ = (int)(wid / gx);
    bx = wid - (int64_t)split_ * gx;
  }
  if (bx * kSweepWaves * UB * 32 >= n_act || n_act <= P.min_active) return;   // (only a compact pass has empty workgroups)
  const int64_t ublock0 = (bx * kSweepWaves + wv) * UB;
  const uint32_t n_items = (uint32_t)P.n_items;
  const int n_tiles = (int)((P.n_items + 31) / 32);
  const int split = split_;
  const int splits = P.splits;
  const float *thr_src = P.hint_in ? P.hint_in : P.tau_sum;

  bf16x8 bu[UB][D / 16], bth[UB];
  int cnt[UB];
  uint32_t *mine[UB];
  uint32_t flip[UB];                     // (CLS) the lane's flip mask of the current scale
#pragma unroll
  for (int b = 0; b < UB; ++b) {
    const int64_t uc = (ublock0 + b) * 32 + ur;
    const bool ok = uc < n_act;
    const int64_t u = (P.user_map && ok) ? (int64_t)P.user_map[uc] : uc;
    cnt[b] = 0;
    mine[b] = P.cand + (((size_t)split * P.n_users + (ok ? u : 0)) * 2 + h) * kPfCap;
    if constexpr (CLS) {
      if (ok && split == 0 && h == 0) {
        const float th = thr_src[u];
        P.theta[u] = (th == th) ? th : -INFINITY;
      }
      flip[b] = 0u;
#pragma unroll
      for (int q = 0; q < D / 16; ++q) {    // (scaled at the first tile: rescale below)
        Frag16 f;
        f.u = make_uint4(0u, 0u, 0u, 0u);
        bu[b][q] = f.v;
      }
      bth[b] = bu[b][0];
    } else {
    float n2;
    load_user_frags<D>(bu[b], n2, P.user_emb, u, ok, h);
#pragma unroll
    for (int q = 0; q < D / 16; ++q) {   // negate: exact, and RNE is symmetric, so the chain gives -s~ bit for bit
      Frag16 f;
      f.v = bu[b][q];
      f.u.x ^= 0x80008000u;
      f.u.y ^= 0x80008000u;
      f.u.z ^= 0x80008000u;
      f.u.w ^= 0x80008000u;
      bu[b][q] = f.v;
    }
    n2 += __shfl_xor(n2, 32, 64);
    const float cu = bf16_ceil_pos(kBf16ErrCoef * sqrtf(n2) + 1e-30f);   // e~_uj = cu * item_norm[j]
    float th = INFINITY;  // padding users never qualify
    if (ok) {
      th = thr_src[u];
      th = (th == th) ? bf16_floor(th) : -INFINITY;    // (a NaN threshold = none: every item is a candidate, the user fails over)
      if (split == 0 && h == 0) P.theta[u] = th;
    }
    // user side of the bound k-step: (T_u, -c||u||) against the items' (1, ||i_j||)
    Frag16 ft;
    ft.u = make_uint4(h == 0 ? ((__float_as_uint(th) >> 16) | (((__float_as_uint(cu) >> 16) | 0x8000u) << 16)) : 0u, 0u, 0u, 0u);
    bth[b] = ft.v;
    }
  }
  // (CLS) the users' fragments for item norms <= N: scale = +-1 / |T_u - c ||u|| N| (sweep_user_scale), times the block's
  // start constant.  Row, norm and threshold are read again (L2), not kept: a walk re-scales a few dozen times, and the
  // registers they would take cost the kernel its third wave per SIMD.
  auto rescale = [&](float N) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < UB; ++b) {
      const int64_t uc = (ublock0 + b) * 32 + ur;
      const bool ok = uc < n_act;
      const int64_t u = (P.user_map && ok) ? (int64_t)P.user_map[uc] : uc;
      const float *urow = ok ? P.user_emb + (size_t)u * D : nullptr;
      float n2 = user_half_norm2<D>(urow, h);
      n2 += __shfl_xor(n2, 32, 64);
      float th = INFINITY;  // padding users never qualify
      if (ok) {
        th = thr_src[u];
        if (!(th == th)) th = -INFINITY;    // (a NaN threshold = none: every item is a candidate, the user fails over)
      }
      const float sc = sweep_start<UB>(b) * sweep_user_scale(th, sqrtf(n2), N, flip[b]);
      load_user_frags_scaled<D>(bu[b], urow, h, sc);
    }
  };
  int cur_nb = 0;                        // bits of the bound the fragments are scaled for

  auto consume = [&](const uint4 (&a)[FR], int t, int seq) __attribute__((always_inline)) {
    const uint32_t j0 = (uint32_t)t * 32u;
    // all UB accumulation chains first, k-step major: UB independent MFMAs between two dependent ones
    f32x16 accs[UB];
#pragma unroll
    for (int b = 0; b < UB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) accs[b][i] = CLS ? sweep_start<UB>(b) : 0.f;
#pragma unroll
    for (int q = 0; q < D / 16; ++q) {
      Frag16 f;
      f.u = a[q];
#pragma unroll
      for (int b = 0; b < UB; ++b) accs[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v, bu[b][q], accs[b], 0, 0, 0);  // -s~ | C -+ s~'
    }
    if constexpr (!CLS) {
      Frag16 fa;
      fa.u = a[D / 16];
#pragma unroll
      for (int b = 0; b < UB; ++b)
        accs[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, bth[b], accs[b], 0, 0, 0);  // T_u - (s~ + e~) = T_u - v_j
    }
#pragma unroll
    for (int b = 0; b < UB; ++b) {
      f32x16 &acc = accs[b];
      // (rows past the table are zero vectors with a zero norm: they qualify when T_u < 0; the selection drops them --
      //  masking them here put 42 more instructions into every tile's block for 31 rows of the whole table)
      // bit (15 - reg) <=> v_j > T_u: the accumulator's sign bit (CLS: ^ flip)
      uint32_t qbits = 0;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) qbits = __builtin_amdgcn_alignbit(qbits, __float_as_uint(acc[reg]), 31);
      if constexpr (CLS) qbits ^= flip[b];
      if (qbits) {
        // past kPfCap entries are counted, not stored: the selection sees the overflow and flags the user
        const uint32_t entry = ((uint32_t)seq << 16) | qbits;
        if constexpr (kSweepPark > 0) {
          if (cnt[b] < kSweepPark) park_s[wv][b][cnt[b]][lane] = entry;
          else if (cnt[b] < kPfCap) mine[b][cnt[b]] = entry;
        } else {
          if (cnt[b] < kPfCap) mine[b][cnt[b]] = entry;
        }
        ++cnt[b];
      }
    }
  };

  // stage s = tiles split + (kSweepStage * s + i) * splits, i < kSweepStage.  Global -> registers for stage s + 1
  // while stage s is consumed from LDS; one barrier per stage.
  constexpr int PER = (kSweepStage * FR * 64 + 64 * kSweepWaves - 1) / (64 * kSweepWaves);   // uint4 per thread and stage
  uint4 pre[PER];
  float pre_nb = 0.f;
  auto fetch = [&](int s0) __attribute__((always_inline)) {
    // (CLS) the bound of the stage's first tile (it exists) = of all its tiles: non-increasing in t.  Every lane loads
    // it with the stage's fragments (one address); it is looked at after the wait for those
    if constexpr (CLS) pre_nb = P.tile_bound[split + kSweepStage * s0 * splits];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int e = threadIdx.x + p * 64 * kSweepWaves;          // element of the stage: tile slot i, fragment word w
      const int i = e / (FR * 64), w = e % (FR * 64);
      int tt = split + (kSweepStage * s0 + i) * splits;
      if (tt >= n_tiles) tt = n_tiles - 1;                        // (a slot past the end is loaded, never consumed)
      pre[p] = make_uint4(0u, 0u, 0u, 0u);
      if (i < kSweepStage) pre[p] = P.packed[(size_t)tt * (FR * 64) + w];
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int e = threadIdx.x + p * 64 * kSweepWaves;
      if (e < kSweepStage * FR * 64) (&stage[buf][0][0])[e] = pre[p];
    }
  };
  const int n_mine = split < n_tiles ? (n_tiles - split + splits - 1) / splits : 0;   // tiles of this split
  const int n_stages = (n_mine + kSweepStage - 1) / kSweepStage;
  int next_nb = 0;                       // (CLS) bits of the bound of the stage that was staged last
  if (n_stages > 0) {
    fetch(0);
    stash(0);
    if constexpr (CLS) next_nb = __builtin_amdgcn_readfirstlane(__float_as_int(pre_nb));
  }
  __syncthreads();
  // Outer loop (CLS; otherwise one pass): one turn per run of stages with the same bound -- the users' fragments are
  // (re-)scaled at its top, where nothing of the tile loop is live, and are loop constants of the inner loop.
  int s0 = 0;
  while (s0 < n_stages) {                                         // block-uniform
    if constexpr (CLS) {
      cur_nb = next_nb;
      rescale(__int_as_float(cur_nb));
    }
    for (;;) {
      const int buf = s0 & 1;
      if (s0 + 1 < n_stages) fetch(s0 + 1);
#pragma unroll
      for (int i = 0; i < kSweepStage; ++i) {
        const int seq = kSweepStage * s0 + i;
        const int t = split + seq * splits;
        if (t < n_tiles) {                                        // block-uniform
          uint4 a[FR];
#pragma unroll
          for (int q = 0; q < FR; ++q) a[q] = stage[buf][i][q * 64 + lane];
          consume(a, t, seq);
        }
      }
      if (s0 + 1 < n_stages) {
        stash(buf ^ 1);
        if constexpr (CLS) next_nb = __builtin_amdgcn_readfirstlane(__float_as_int(pre_nb));
      }
      __syncthreads();
      ++s0;
      if (s0 >= n_stages) break;
      if constexpr (CLS) {
        if (next_nb != cur_nb) break;                             // the walk enters another run of classes
      }
    }
  }
#pragma unroll
  for (int b = 0; b < UB; ++b) {
    if constexpr (kSweepPark > 0) {
      // the parked entries: a lane reads back its own column (its wave's LDS operations are ordered: no barrier) and
      // writes four entries per store; slots past the count receive stale words the selection never reads
#pragma unroll
      for (int c0 = 0; c0 < kSweepPark; c0 += 4) {
        if (c0 < cnt[b]) {
          uint4 q;
          q.x = park_s[wv][b][c0][lane], q.y = park_s[wv][b][c0 + 1][lane];
          q.z = park_s[wv][b][c0 + 2][lane], q.w = park_s[wv][b][c0 + 3][lane];
          *reinterpret_cast<uint4 *>(mine[b] + c0) = q;
        }
      }
    }
    const int64_t uc = (ublock0 + b) * 32 + ur;
    if (uc < n_act) {
      const int64_t u = P.user_map ? (int64_t)P.user_map[uc] : uc;
      P.cand_cnt[((size_t)split * P.n_users + u) * 2 + h] = cnt[b];
    }
  }
}

// ---- select + exact re-score -----------------------------------------------------------------------------------
// exact fp32 score of (u, item): the chain of the f32 MFMA kernel for D <= 128 (k = s and D/2 + s alternate)
template <int D>
__device__ __forceinline__ float exact_score(const float *__restrict__ urow, const float *__restrict__ irow) {
  float acc = 0.f;
  const float4 *i0 = reinterpret_cast<const float4 *>(irow), *i1 = reinterpret_cast<const float4 *>(irow + D / 2);
  const float4 *u0 = reinterpret_cast<const float4 *>(urow), *u1 = reinterpret_cast<const float4 *>(urow + D / 2);
#pragma unroll
  for (int q = 0; q < D / 8; ++q) {
    const float4 a = i0[q], b = i1[q], x = u0[q], y = u1[q];
    acc = __fmaf_rn(x.x, a.x, acc);
    acc = __fmaf_rn(y.x, b.x, acc);
    acc = __fmaf_rn(x.y, a.y, acc);
    acc = __fmaf_rn(y.y, b.y, acc);
    acc = __fmaf_rn(x.z, a.z, acc);
    acc = __fmaf_rn(y.z, b.z, acc);
    acc = __fmaf_rn(x.w, a.w, acc);
    acc = __fmaf_rn(y.w, b.w, acc);
  }
  return acc;
}

// Exact scores of 64 CONSECUTIVE items [i0, i0 + 64) of one user, lane j -> item i0 + j (items >= n_items: garbage,
// rows clamped into the table), fetched by the wave together: 4 lanes per row, 64 consecutive bytes per instruction
// and group -- whole sectors, 16 rows per instruction.  With a lane per row every load instruction asks for 64
// different sectors and uses 16 bytes of each; a 1024-thread block of the exact route then spends ~35 us waiting for
// its 2048 rows, the rest of its work ~10.  Lane l of a group holds floats [16 t + 4 l, 16 t + 4 l + 4) of both row
// halves (ua, ub: the user's), the operands of chain steps 32 t + 8 l .. + 7; the accumulator walks round the
// group's lanes D/32 times in the k-ascending order of the f32 MFMA kernel, every lane running every segment.
__device__ __forceinline__ float dpp_quad_rot(float v) {   // lane i <- lane (i - 1) % 4 of its quad
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93 /* quad_perm:[3,0,1,2] */, 0xF, 0xF, false));
}
template <int D>
__device__ __forceinline__ float exact_scores_64(const float *__restrict__ item_emb, int64_t i0, int64_t n_items,
                                                 const float (&ua)[D / 8], const float (&ub)[D / 8], int lane) {
  constexpr int G = 4, CPI = 16, SEG = D / 8;
  const int l = lane % G, g = lane / G;
  auto load_rows = [&](float (&a)[SEG], float (&b)[SEG], int pass) __attribute__((always_inline)) {
    const int64_t item = min(i0 + pass * CPI + g, n_items - 1);
    const float *row = item_emb + (size_t)item * D + 4 * l;
#pragma unroll
    for (int i = 0; i < SEG; i += 4) {
      const float4 x = reinterpret_cast<const float4 *>(row + 4 * i)[0];
      const float4 y = reinterpret_cast<const float4 *>(row + D / 2 + 4 * i)[0];
      a[i] = x.x, a[i + 1] = x.y, a[i + 2] = x.z, a[i + 3] = x.w;
      b[i] = y.x, b[i + 1] = y.y, b[i + 2] = y.z, b[i + 3] = y.w;
    }
  };
  float sc = 0.f;
  float a[SEG], b[SEG];
  load_rows(a, b, 0);
#pragma unroll
  for (int it = 0; it < G; ++it) {     // pass `it`: items i0 + it * 16 + g
    float ca[SEG], cb[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) ca[i] = a[i], cb[i] = b[i];
    if (it + 1 < G) load_rows(a, b, it + 1);
    float acc = 0.f;
#pragma unroll
    for (int seg = 0; seg < SEG; ++seg) {
      float t = acc;
#pragma unroll
      for (int i = 4 * (seg / G); i < 4 * (seg / G) + 4; ++i) {
        t = __fmaf_rn(ua[i], ca[i], t);
        t = __fmaf_rn(ub[i], cb[i], t);
      }
      if (l == seg % G) acc = t;
      if (seg + 1 < SEG) {
        const float nx = dpp_quad_rot(acc);
        if (l == (seg + 1) % G) acc = nx;
      }
    }
    // the group's last lane holds the score of item i0 + it * 16 + g: route it to lane it * 16 + g
    const float moved = __shfl(acc, (lane % CPI) * G + G - 1, 64);
    if (lane / CPI == it) sc = moved;
  }
  return sc;
}

// The `rank`-th largest 64-bit key (keys unique, 0 = none; at least `rank` non-zero keys): bitwise search with wave
// ballots and scalar popcounts, 32 steps on the score word, up to 32 more on the index word when a tie straddles the rank.
template <int NR>
__device__ __forceinline__ uint64_t kth_largest_key(const uint64_t (&k)[NR], int rank) {
  uint32_t T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t candv = T | (1u << bit);
    int c = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) c += __popcll(__ballot(k[r] != 0ull && (uint32_t)(k[r] >> 32) >= candv));
    if (c >= rank) T = candv;
  }
  int gt = 0, eq = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    gt += __popcll(__ballot(k[r] != 0ull && (uint32_t)(k[r] >> 32) > T));
    eq += __popcll(__ballot(k[r] != 0ull && (uint32_t)(k[r] >> 32) == T));
  }
  const int need = rank - gt;      // >= 1
  uint32_t L = 0;
  if (eq > need) {
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t candv = L | (1u << bit);
      int c = 0;
#pragma unroll
      for (int r = 0; r < NR; ++r)
        c += __popcll(__ballot(k[r] != 0ull && (uint32_t)(k[r] >> 32) == T && (uint32_t)k[r] >= candv));
      if (c >= need) L = candv;
    }
  }
  return ((uint64_t)T << 32) | (uint64_t)L;
}

// Wave-wide descending bitonic sort of 64 keys, one per lane.
__device__ __forceinline__ void sort64_desc(uint64_t &e, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint64_t p = shfl_xor_u64(e, j);
      const bool lower = (lane & j) == 0;
      const bool desc = (k == 64) ? true : ((lane & k) == 0);
      const uint64_t mx = e > p ? e : p, mn = e > p ? p : e;
      e = (lower == desc) ? mx : mn;
    }
  }
}

constexpr int kPfHistLds = 1024;            // longest history the exact per-user route keeps in LDS
constexpr int kPfSelHist = 128;             // ... and the selection (longer histories are searched in global memory)

// The selection is bound by instruction issue (profiles/r02_c_score_pmc.txt: its VALU pipes are busy for the whole
// kernel), so everything below is written for few instructions per user:
//  * prefix sums and owner look-ups are DPP scans (6 instructions), not shuffles through LDS or binary searches;
//  * the exact re-score fetches rows by 4 lanes per candidate and keeps the next rows in flight under the chain;
//  * the ranking is a bitonic network with the per-stage lane masks as scalar constants: compare, two selects.

// inclusive scans over the 64 lanes: shifts by 1, 2, 4, 8 inside the rows of 16, then lane 15 of row 0 / 2 into
// row 1 / 3 and lane 31 into rows 2 and 3
__device__ __forceinline__ int wave_scan_add(int x) {
  x += dpp_or0<0x111, 0xF, true>(x);
  x += dpp_or0<0x112, 0xF, true>(x);
  x += dpp_or0<0x114, 0xF, true>(x);
  x += dpp_or0<0x118, 0xF, true>(x);
  x += dpp_or0<0x142, 0xA, false>(x);
  x += dpp_or0<0x143, 0xC, false>(x);
  return x;
}
__device__ __forceinline__ int wave_scan_max(int x) {   // x >= 0
  x = max(x, dpp_or0<0x111, 0xF, true>(x));
  x = max(x, dpp_or0<0x112, 0xF, true>(x));
  x = max(x, dpp_or0<0x114, 0xF, true>(x));
  x = max(x, dpp_or0<0x118, 0xF, true>(x));
  x = max(x, dpp_or0<0x142, 0xA, false>(x));
  x = max(x, dpp_or0<0x143, 0xC, false>(x));
  return x;
}
__device__ __forceinline__ float dpp_prev_lane(float v) {   // lane i <- lane i - 1 (within its row of 16 lanes)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, false));
}

// ---- bitonic networks over one key per lane, lane masks as scalar constants ---------------------------------------
__device__ __forceinline__ uint32_t lane_select(uint32_t a, uint32_t b, uint64_t m) {   // bit of the lane set in m ? b : a
  uint32_t r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}
__device__ __forceinline__ uint64_t lane_select64(uint64_t a, uint64_t b, uint64_t m) {
  return ((uint64_t)lane_select((uint32_t)(a >> 32), (uint32_t)(b >> 32), m) << 32) | lane_select((uint32_t)a, (uint32_t)b, m);
}
struct PermAddr {   // byte addresses of ds_bpermute: lane ^ 1, 2, 4, 8, 16, 32 and 63 - lane
  int x0, x1, x2, x3, x4, x5, rev;
  template <int I>
  __device__ __forceinline__ int x() const {
    if constexpr (I == 0) return x0;
    else if constexpr (I == 1) return x1;
    else if constexpr (I == 2) return x2;
    else if constexpr (I == 3) return x3;
    else if constexpr (I == 4) return x4;
    else return x5;
  }
};
__device__ __forceinline__ PermAddr perm_addr(int lane) {
  PermAddr a;
  a.x0 = (lane ^ 1) << 2, a.x1 = (lane ^ 2) << 2, a.x2 = (lane ^ 4) << 2, a.x3 = (lane ^ 8) << 2;
  a.x4 = (lane ^ 16) << 2, a.x5 = (lane ^ 32) << 2, a.rev = (63 - lane) << 2;
  return a;
}
__device__ __forceinline__ uint64_t permute64(uint64_t v, int addr) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// lanes of `keepmax` keep the larger of (e, p), the others the smaller: one compare, one scalar xnor, two selects
__device__ __forceinline__ void compare_exchange(uint64_t &e, uint64_t p, uint64_t keepmax) {
  const uint64_t gt = __ballot(p > e);
  e = lane_select64(e, p, ~(gt ^ keepmax));
}
constexpr uint64_t keepmax_mask(int k, int j, bool all_desc) {   // stage (k, j) of a descending bitonic sort
  uint64_t m = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const bool lower = (lane & j) == 0, desc = all_desc || (lane & k) == 0;
    if (lower == desc) m |= 1ull << lane;
  }
  return m;
}
template <int LOGJ>
__device__ __forceinline__ void merge_stages(uint64_t &e, const PermAddr &pa) {   // bitonic -> descending, strides 2^LOGJ .. 1
  compare_exchange(e, permute64(e, pa.template x<LOGJ>()), keepmax_mask(64, 1 << LOGJ, true));
  if constexpr (LOGJ > 0) merge_stages<LOGJ - 1>(e, pa);
}
template <int LOGK, int LOGJ>
__device__ __forceinline__ void sort_stages(uint64_t &e, const PermAddr &pa) {
  compare_exchange(e, permute64(e, pa.template x<LOGJ>()), keepmax_mask(1 << LOGK, 1 << LOGJ, LOGK == 6));
  if constexpr (LOGJ > 0) sort_stages<LOGK, LOGJ - 1>(e, pa);
  else if constexpr (LOGK < 6) sort_stages<LOGK + 1, LOGK>(e, pa);
}
__device__ __forceinline__ void sort64_keys(uint64_t &e, const PermAddr &pa) { sort_stages<1, 0>(e, pa); }   // descending
// two descending registers -> the 128 keys in descending order over (e0, e1)
__device__ __forceinline__ void merge128_keys(uint64_t &e0, uint64_t &e1, const PermAddr &pa) {
  const uint64_t r = permute64(e1, pa.rev);
  const uint64_t gt = __ballot(r > e0);
  const uint64_t hi = lane_select64(e0, r, gt), lo = lane_select64(r, e0, gt);
  e0 = hi;
  e1 = lo;
  merge_stages<5>(e0, pa);
  merge_stages<5>(e1, pa);
}
// 64 more keys (any order) into the descending top-128 (e0, e1); what falls out of the 128 is dropped
__device__ __forceinline__ void merge_block_keys(uint64_t &e0, uint64_t &e1, uint64_t b, const PermAddr &pa) {
  sort64_keys(b, pa);
  const uint64_t r = permute64(b, pa.rev);
  e1 = lane_select64(e1, r, __ballot(r > e1));   // descending vs ascending: the lane-wise maxima are the 64 largest, bitonic
  merge_stages<5>(e1, pa);
  merge128_keys(e0, e1, pa);
}

// 64 more keys (0 = none) into the descending top-128 (e0, e1); `blocks` = how many were taken before
__device__ __forceinline__ void take_block_keys(uint64_t &e0, uint64_t &e1, int &blocks, uint64_t cur, const PermAddr &pa) {
  if (blocks == 0) {
    sort64_keys(cur, pa);
    e0 = cur;
  } else if (blocks == 1) {
    sort64_keys(cur, pa);
    uint64_t a = e0, b = cur;
    merge128_keys(a, b, pa);
    e0 = a, e1 = b;
  } else {
    uint64_t a = e0, b = e1;
    merge_block_keys(a, b, cur, pa);
    e0 = a, e1 = b;
  }
  ++blocks;
}
__device__ __forceinline__ uint64_t key_of_rank(uint64_t e0, uint64_t e1, int rank) {   // rank < 128, wave-uniform
  const uint32_t lo0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e0, rank & 63);
  const uint32_t hi0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e0 >> 32), rank & 63);
  const uint32_t lo1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e1, rank & 63);
  const uint32_t hi1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e1 >> 32), rank & 63);
  return rank < 64 ? (((uint64_t)hi0 << 32) | lo0) : (((uint64_t)hi1 << 32) | lo1);
}

// Exact scores of the 64 candidates cand_s[base .. base + 64) (lane j -> candidate base + j; beyond n_cand: 0).
// What the re-score costs is ADDRESSES as much as flops: a lane that loads its own candidate's row puts 64 different
// cache lines into every load instruction, and the CU's address path retires about one line per clock.  So a row is
// fetched by G = 4 lanes per candidate (16 candidates per load instruction), lane l of a group holding floats
// [S l, S l + S) and [D/2 + S l, D/2 + S l + S) of the row, S = D/8 -- the operands of chain steps 2 S l .. 2 S l + 2 S - 1
// (the chain of the f32 MFMA kernel alternates between the two halves of the row; ua, ub: the user's).  The chain stays
// sequential: the accumulator walks through the group's lanes, one DPP move per 2 S fmas, every lane executing every
// segment (the result of the lane whose turn it is is kept); the next 16 candidates' rows fly under it.
// (Tried, each bit-exact: a pipeline in which every lane's fma is a useful one, 4 lanes per row and 8 lanes per row
//  with whole-line fetches, 2 to 12 row pieces in flight per lane: 112 and 170 us for this stage against 44 us.)
#ifndef CHAOREC_SEL_G
#define CHAOREC_SEL_G 4
#endif
constexpr int kSelG = CHAOREC_SEL_G;      // lanes per candidate row in the selection's re-score (2: half the chain instructions
                                          // per candidate, twice the row registers)
template <int D>
__device__ __forceinline__ float chain_block(const float *__restrict__ item_emb, const uint32_t *cand_s, int base, int n_cand,
                                             const float (&ua)[D / (2 * kSelG)], const float (&ub)[D / (2 * kSelG)], int lane) {
  constexpr int G = kSelG, CPI = 64 / G, SEG = D / (2 * G);
  const int l = lane % G, g = lane / G;
  auto load_rows = [&](float (&a)[SEG], float (&b)[SEG], int c) __attribute__((always_inline)) {
    const float *row = item_emb + (size_t)cand_s[c] * D;
#pragma unroll
    for (int i = 0; i < SEG; i += 4) {
      const float4 x = reinterpret_cast<const float4 *>(row + SEG * l)[i / 4];
      const float4 y = reinterpret_cast<const float4 *>(row + D / 2 + SEG * l)[i / 4];
      a[i] = x.x, a[i + 1] = x.y, a[i + 2] = x.z, a[i + 3] = x.w;
      b[i] = y.x, b[i + 1] = y.y, b[i + 2] = y.z, b[i + 3] = y.w;
    }
  };
  float sc = 0.f;
  float a[SEG], b[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) a[i] = b[i] = 0.f;
  if (base + g < n_cand) load_rows(a, b, base + g);
  for (int it = 0; it < G; ++it) {     // pass `it`: candidates base + it * CPI + g
    if (base + it * CPI >= n_cand) break;                      // wave-uniform
    float ca[SEG], cb[SEG];
#pragma unroll
    for (int i = 0; i < SEG; ++i) ca[i] = a[i], cb[i] = b[i];
    const int cn = base + (it + 1) * CPI + g;                  // next pass's rows fly under this pass's chain
    if (it + 1 < G && cn < n_cand) load_rows(a, b, cn);
    float acc = 0.f;
#pragma unroll
    for (int seg = 0; seg < G; ++seg) {
      float t = acc;
#pragma unroll
      for (int i = 0; i < SEG; ++i) {
        t = __fmaf_rn(ua[i], ca[i], t);
        t = __fmaf_rn(ub[i], cb[i], t);
      }
      if (l == seg) acc = t;
      if (seg + 1 < G) {
        const float nx = dpp_prev_lane(acc);
        if (l == seg + 1) acc = nx;
      }
    }
    // the group's last lane holds the score of candidate base + it * CPI + g: route it to lane it * CPI + g
    const float moved = __shfl(acc, (lane % CPI) * G + G - 1, 64);
    if (lane / CPI == it) sc = moved;
  }
  return sc;
}

// reason codes (non-zero = not certified): 1 a sweep list overflowed, 2 fewer than K candidates, 3 more candidates
// than the selection holds, 4 the K-th best exact score does not clear the sweep threshold
template <int D, int MAXC>
__device__ __forceinline__ void select_user(const PrefArgs &P, const int64_t u, int lane, int2 *meta_s, uint32_t *cand_s,
                                            uint32_t *hist_s, float *score_s) {
  const int K = P.K;
  const int n_lists = 2 * P.splits;  // <= 32
  const PermAddr pa = perm_addr(lane);

  // this lane's list of the sweep: (split lane / 2, half lane & 1)
  const int listidx = (int)(((int64_t)(lane >> 1) * P.n_users + u) * 2 + (lane & 1));
  int c = 0;
  if (lane < n_lists) c = P.cand_cnt[listidx];
  // this lane's segments of the user's row (chain_block)
  float ua[D / (2 * kSelG)], ub[D / (2 * kSelG)];
  {
    const float *urow = P.user_emb + (size_t)u * D + (D / (2 * kSelG)) * (lane % kSelG);
#pragma unroll
    for (int i = 0; i < D / (2 * kSelG); i += 4) {
      const float4 x = reinterpret_cast<const float4 *>(urow)[i / 4], y = reinterpret_cast<const float4 *>(urow + D / 2)[i / 4];
      ua[i] = x.x, ua[i + 1] = x.y, ua[i + 2] = x.z, ua[i + 3] = x.w;
      ub[i] = y.x, ub[i + 1] = y.y, ub[i + 2] = y.z, ub[i + 3] = y.w;
    }
  }
  int64_t hb = 0, he = 0;
  if (P.hist_rowptr) {
    hb = P.hist_rowptr[u];
    he = P.hist_rowptr[u + 1];
  }
  const int deg = (int)(he - hb);
  const bool hist_lds = deg <= kPfSelHist;
  const float theta = P.theta[u];
  const bool overflow = __any(c > kPfCap);
  const int incl = wave_scan_add(c);
  const int total = __builtin_amdgcn_readlane(incl, 63);   // entries (each holds >= 1 candidate)
  int why = overflow ? 1 : (total > MAXC ? 3 : 0);
  int n_cand = 0;
  if (why == 0) {
    // entry e of the concatenated lists belongs to the last list that starts at or before it: the lists mark their
    // first entry with their number + 1, a running maximum over the entries spreads it
    int *owner_s = reinterpret_cast<int *>(score_s);
#pragma unroll 1
    for (int i = lane; i < total; i += 64) owner_s[i] = 0;
    meta_s[lane] = make_int2(incl - c, listidx);
    if (hist_lds) {
#pragma unroll 1
      for (int i = lane; i < deg; i += 64) hist_s[i] = (uint32_t)P.hist_col[hb + i];
    }
    __builtin_amdgcn_wave_barrier();
    if (c > 0) owner_s[incl - c] = lane + 1;
    __builtin_amdgcn_wave_barrier();
    int owner_carry = 0;
#pragma unroll 1
    for (int base = 0; base < total; base += 64) {
      const int e = base + lane;
      int own = e < total ? owner_s[e] : 0;
      own = max(wave_scan_max(own), owner_carry);
      owner_carry = __builtin_amdgcn_readlane(own, 63);
      uint32_t bits = 0, j0 = 0;
      if (e < total) {
        const int list = own - 1;
        const int2 m = meta_s[list];
        const uint32_t raw = P.cand[(size_t)m.y * kPfCap + (e - m.x)];
        bits = raw & 0xFFFFu;
        j0 = ((uint32_t)(list >> 1) + (raw >> 16) * (uint32_t)P.splits) * 32u + 4u * (uint32_t)(list & 1);
      }
      if (__any(bits != 0u && j0 + 32u > (uint32_t)P.n_items)) {   // the table's last tile (a scalar branch: rarely taken):
#pragma unroll 1
        for (int reg = 0; reg < 16; ++reg)                          // the sweep does not mask the rows past the table's end
          if (j0 + (uint32_t)((reg & 3) + 8 * (reg >> 2)) >= (uint32_t)P.n_items) bits &= ~(1u << (15 - reg));
      }
      const int pc = __popc(bits);
      const int ex = wave_scan_add(pc);
      int slot = n_cand + ex - pc;
#pragma unroll 1
      while (__any(bits != 0u)) {
        if (bits) {
          const int bit = 31 - __clz(bits);
          bits &= ~(1u << bit);
          const int reg = 15 - bit;
          if (slot < MAXC) cand_s[slot] = j0 + (uint32_t)((reg & 3) + 8 * (reg >> 2));
          ++slot;
        }
      }
      n_cand += __builtin_amdgcn_readlane(ex, 63);
    }
    if (n_cand > MAXC) why = 3;
    __builtin_amdgcn_wave_barrier();
    if (P.perm && why == 0) {        // norm-sorted table: positions of the packed table -> items (all < n_items, see above)
#pragma unroll 1
      for (int i = lane; i < n_cand; i += 64) cand_s[i] = (uint32_t)P.perm[cand_s[i]];
      __builtin_amdgcn_wave_barrier();
    }
  }
  uint64_t e0 = 0ull, e1 = 0ull;      // the 128 best keys, descending over (e0, e1)
  if (why == 0) {
    auto in_hist = [&](uint32_t item) -> bool {
      int lo = 0, hi = deg;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t hv = hist_lds ? hist_s[mid] : (uint32_t)P.hist_col[hb + mid];
        if (hv < item) lo = mid + 1; else hi = mid;
      }
      return lo < deg && (hist_lds ? hist_s[lo] : (uint32_t)P.hist_col[hb + lo]) == item;
    };
    const uint32_t mord = f32_to_ord(P.mask_value);
    int valid = 0, above = 0, blocks = 0;
#pragma unroll 1
    for (int base = 0; base < n_cand; base += 64) {
      const float sc = chain_block<D>(P.item_emb, cand_s, base, n_cand, ua, ub, lane);
      const int idx = base + lane;
      uint64_t cur = 0ull;
      if (idx < n_cand) {
        const uint32_t item = cand_s[idx];
        // (a history member leaves the candidates: the sweep ran unmasked)
        if (deg == 0 || !in_hist(item)) cur = make_key(sc, item);
      }
      valid += __popcll(__ballot(cur != 0ull));
      above += __popcll(__ballot(cur != 0ull && (uint32_t)(cur >> 32) > mord));
      take_block_keys(e0, e1, blocks, cur, pa);
    }
    int n_keys = valid;
    // The masked row restricted to what can matter = the candidates + the user's history at mask_value.  The history
    // only joins when mask_value can reach the top-K (the reference's 1e-6 / 1e-5 does when the real scores are tiny
    // or negative: quirk Q7), i.e. when fewer than K candidates beat it.
    if (deg > 0 && above < K) {   // wave-uniform
#pragma unroll 1
      for (int i0 = 0; i0 < deg; i0 += 64) {
        const int i = i0 + lane;
        take_block_keys(e0, e1, blocks, i < deg ? make_key(P.mask_value, hist_lds ? hist_s[i] : (uint32_t)P.hist_col[hb + i]) : 0ull, pa);
      }
      n_keys = valid + deg;
    }
    if (n_keys < K) why = 2;
    if (why == 0) {
      // certification: the K-th best exact score must clear the threshold the sweep used
      const uint64_t kth = key_of_rank(e0, e1, K - 1);   // K <= 64
      if (kth == 0ull || !(ord_to_f32((uint32_t)(kth >> 32)) > theta)) why = 4;
      if (why == 0 && P.hint_out) {
        // next call's threshold: one float below the exact score of rank `want` (>= K): equal scores stay candidates
        const int want = min(max(P.hint_rank, K), 128);
        const uint64_t hk = key_of_rank(e0, e1, min(want, n_keys) - 1);
        float t = nextafterf(ord_to_f32((uint32_t)(hk >> 32)), -INFINITY);
        if (n_keys < want) {
          // Fewer candidates than the rank the threshold is taken at: the user's scores sank below the carried
          // threshold since it was set.  The lowest candidate is then (about) that same threshold again, and a user
          // that keeps sinking ends on the exact route (30 us for the call, however few users take it).  Take the
          // threshold where rank `want` would be if the scores went on falling as they do between rank K and the
          // last candidate.  Any value is legal.
          const float s_k = ord_to_f32((uint32_t)(kth >> 32)), s_last = ord_to_f32((uint32_t)(hk >> 32));
          const float s_top = ord_to_f32((uint32_t)(key_of_rank(e0, e1, 0) >> 32));
          const float slope = n_keys > K ? (s_k - s_last) / (float)(n_keys - K) : (s_top - s_k) / (float)max(K - 1, 1);
          t -= slope * (float)(want - n_keys);
        }
        if (lane == 0) P.hint_out[u] = t;
      }
    }
  }
  if (lane == 0) {
    P.fail[u] = why;
    P.n_cand[u] = n_cand;
    if (why != 0) {
      // not certifiable from this threshold: retry with a sampled one (pass A); more candidates than this
      // instantiation holds (but no list overflow): the wide one; else the exact per-user route
      if (P.retry_cnt) P.retry_list[atomicAdd(P.retry_cnt, 1)] = (int)u;
      else if (why == 3 && MAXC < kPfMaxCandWide && P.wide_cnt && n_cand <= kPfMaxCandWide && total <= kPfMaxCandWide)
        P.wide_list[atomicAdd(P.wide_cnt, 1)] = (int)u;
      else if ((why == 1 || why == 3) && P.reth_cnt) P.reth_list[atomicAdd(P.reth_cnt, 1)] = (int)u;   // threshold too low: pass C
      else P.fb_list[atomicAdd(P.fb_cnt, 1)] = (int)u;
    }
  }
  if (why == 0 && lane < K) {
    const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e0 & 0xFFFFFFFFull);
    P.out_idx[(size_t)u * K + lane] = (int64_t)item + P.id_offset;
    P.out_val[(size_t)u * K + lane] = ord_to_f32((uint32_t)(e0 >> 32));
  }
}

// One wave per user (a fixed grid walking the rows of the pass).
#ifndef CHAOREC_SEL_WAVES
#define CHAOREC_SEL_WAVES 4
#endif
template <int D, int MAXC>
__global__ __launch_bounds__(64, D > 64 ? 3 : (MAXC <= 512 ? CHAOREC_SEL_WAVES : 3)) void score_select_kernel_pf(const PrefArgs P) {
  __shared__ int2 meta_s[64];
  __shared__ uint32_t cand_s[MAXC];
  __shared__ uint32_t hist_s[kPfSelHist];
  __shared__ float score_s[MAXC];
  const int64_t n_act = P.n_active ? (int64_t)*P.n_active : P.n_users;
  if (n_act <= P.min_active) return;
#pragma unroll 1
  for (int64_t i = blockIdx.x; i < n_act; i += gridDim.x) {
    const int64_t u = P.user_map ? (int64_t)P.user_map[i] : i;
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));       // (keeps the lane-dependent constants of the networks out of the loop preheader)
    select_user<D, MAXC>(P, u, lane, meta_s, cand_s, hist_s, score_s);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- pass C: a raised threshold for the users whose candidate lists overflowed --------------------------------------
// One wave per queued user (reth_list).  A list of the sweep holds its FIRST kPfCap entries, the rest were counted; the walk
// below expands what is stored -- at most kPfMaxCandWide entries and candidates, in list order --, scores those candidates
// exactly and takes the new threshold one float below the exact score of rank r among the non-history ones.  r is chosen
// so that about kPfRethTarget items of the user's WHOLE candidate set are expected above it (the walked candidates are
// taken as a sample of that set: the splits' tiles are interleaved over the item range).  r >= K and the r scored items
// themselves lie above the new threshold, so the pass that follows (compact sweep + wide selection) certifies the user
// unless its lists overflow again; whoever cannot get such a threshold is queued for the exact route as before.
// Any threshold is legal: a poor estimate costs a second failure, never a wrong result.
constexpr int kPfRethTarget = 256;

template <int D>
__device__ __forceinline__ void rethreshold_user(const PrefArgs &P, const int64_t u, int lane, int2 *meta_s, uint32_t *cand_s,
                                                 uint32_t *hist_s, float *score_s) {
  constexpr int MAXC = kPfMaxCandWide;
  const int K = P.K;
  const int n_lists = 2 * P.splits;  // <= 32
  const PermAddr pa = perm_addr(lane);
  const int listidx = (int)(((int64_t)(lane >> 1) * P.n_users + u) * 2 + (lane & 1));
  int c_all = 0;
  if (lane < n_lists) c_all = P.cand_cnt[listidx];
  const int c = min(c_all, kPfCap);                              // entries this list holds
  float ua[D / (2 * kSelG)], ub[D / (2 * kSelG)];
  {
    const float *urow = P.user_emb + (size_t)u * D + (D / (2 * kSelG)) * (lane % kSelG);
#pragma unroll
    for (int i = 0; i < D / (2 * kSelG); i += 4) {
      const float4 x = reinterpret_cast<const float4 *>(urow)[i / 4], y = reinterpret_cast<const float4 *>(urow + D / 2)[i / 4];
      ua[i] = x.x, ua[i + 1] = x.y, ua[i + 2] = x.z, ua[i + 3] = x.w;
      ub[i] = y.x, ub[i + 1] = y.y, ub[i + 2] = y.z, ub[i + 3] = y.w;
    }
  }
  int64_t hb = 0, he = 0;
  if (P.hist_rowptr) {
    hb = P.hist_rowptr[u];
    he = P.hist_rowptr[u + 1];
  }
  const int deg = (int)(he - hb);
  const bool hist_lds = deg <= kPfSelHist;
  const int incl = wave_scan_add(c);
  const int stored = __builtin_amdgcn_readlane(incl, 63);
  const int counted = __builtin_amdgcn_readlane(wave_scan_add(c_all), 63);
  const int used = min(stored, MAXC);                            // entries walked (the owner table has MAXC slots)
  int *owner_s = reinterpret_cast<int *>(score_s);
#pragma unroll 1
  for (int i = lane; i < used; i += 64) owner_s[i] = 0;
  meta_s[lane] = make_int2(incl - c, listidx);
  if (hist_lds) {
#pragma unroll 1
    for (int i = lane; i < deg; i += 64) hist_s[i] = (uint32_t)P.hist_col[hb + i];
  }
  __builtin_amdgcn_wave_barrier();
  if (c > 0 && incl - c < used) owner_s[incl - c] = lane + 1;
  __builtin_amdgcn_wave_barrier();
  int n_raw = 0, owner_carry = 0;
#pragma unroll 1
  for (int base = 0; base < used; base += 64) {
    const int e = base + lane;
    int own = e < used ? owner_s[e] : 0;
    own = max(wave_scan_max(own), owner_carry);
    owner_carry = __builtin_amdgcn_readlane(own, 63);
    uint32_t bits = 0, j0 = 0;
    if (e < used) {
      const int list = own - 1;
      const int2 m = meta_s[list];
      const uint32_t raw = P.cand[(size_t)m.y * kPfCap + (e - m.x)];
      bits = raw & 0xFFFFu;
      j0 = ((uint32_t)(list >> 1) + (raw >> 16) * (uint32_t)P.splits) * 32u + 4u * (uint32_t)(list & 1);
    }
    if (__any(bits != 0u && j0 + 32u > (uint32_t)P.n_items)) {
#pragma unroll 1
      for (int reg = 0; reg < 16; ++reg)
        if (j0 + (uint32_t)((reg & 3) + 8 * (reg >> 2)) >= (uint32_t)P.n_items) bits &= ~(1u << (15 - reg));
    }
    const int pc = __popc(bits);
    const int ex = wave_scan_add(pc);
    int slot = n_raw + ex - pc;
#pragma unroll 1
    while (__any(bits != 0u)) {
      if (bits) {
        const int bit = 31 - __clz(bits);
        bits &= ~(1u << bit);
        const int reg = 15 - bit;
        if (slot < MAXC) cand_s[slot] = j0 + (uint32_t)((reg & 3) + 8 * (reg >> 2));
        ++slot;
      }
    }
    n_raw += __builtin_amdgcn_readlane(ex, 63);
  }
  const int n_cand = min(n_raw, MAXC);
  __builtin_amdgcn_wave_barrier();
  if (P.perm) {
#pragma unroll 1
    for (int i = lane; i < n_cand; i += 64) cand_s[i] = (uint32_t)P.perm[cand_s[i]];
    __builtin_amdgcn_wave_barrier();
  }
  auto in_hist = [&](uint32_t item) -> bool {
    int lo = 0, hi = deg;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const uint32_t hv = hist_lds ? hist_s[mid] : (uint32_t)P.hist_col[hb + mid];
      if (hv < item) lo = mid + 1; else hi = mid;
    }
    return lo < deg && (hist_lds ? hist_s[lo] : (uint32_t)P.hist_col[hb + lo]) == item;
  };
  uint64_t e0 = 0ull, e1 = 0ull;
  int valid = 0, blocks = 0;
#pragma unroll 1
  for (int base = 0; base < n_cand; base += 64) {
    const float sc = chain_block<D>(P.item_emb, cand_s, base, n_cand, ua, ub, lane);
    const int idx = base + lane;
    uint64_t cur = 0ull;
    if (idx < n_cand) {
      const uint32_t item = cand_s[idx];
      if (deg == 0 || !in_hist(item)) cur = make_key(sc, item);
    }
    valid += __popcll(__ballot(cur != 0ull));
    take_block_keys(e0, e1, blocks, cur, pa);
  }
  bool queued = false;
  if (valid >= K && n_cand > 0) {   // wave-uniform
    // candidates of the whole set per walked one: counted / walked entries, and the walk's own cut at MAXC candidates
    const float scale = ((float)counted / (float)used) * ((float)n_raw / (float)n_cand);
    int r = (int)ceilf((float)kPfRethTarget / scale);
    r = min(min(max(r, K + 6), 128), valid);
    if ((float)r * scale <= 0.9f * (float)kPfMaxCandWide) {
      const uint64_t rk = key_of_rank(e0, e1, r - 1);
      if (lane == 0) {
        P.tau_sum[u] = nextafterf(ord_to_f32((uint32_t)(rk >> 32)), -INFINITY);
        P.rt2_list[atomicAdd(P.rt2_cnt, 1)] = (int)u;
      }
      queued = true;
    }
  }
  if (!queued && lane == 0) P.fb_list[atomicAdd(P.fb_cnt, 1)] = (int)u;
}

template <int D>
__global__ __launch_bounds__(64, 3) void score_rethreshold_kernel(const PrefArgs P) {
  __shared__ int2 meta_s[64];
  __shared__ uint32_t cand_s[kPfMaxCandWide];
  __shared__ uint32_t hist_s[kPfSelHist];
  __shared__ float score_s[kPfMaxCandWide];
  const int64_t n_act = (int64_t)*P.reth_cnt;
#pragma unroll 1
  for (int64_t i = blockIdx.x; i < n_act; i += gridDim.x) {
    const int64_t u = (int64_t)P.reth_list[i];
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    rethreshold_user<D>(P, u, lane, meta_s, cand_s, hist_s, score_s);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- exact per-user route for the users the prefilter could not certify ------------------------------------------
// kExSlices 1024-thread blocks per queued user (fixed grid walking the device-side queue), each taking a contiguous
// slice of the items.  All scores are computed with the exact fp32 chain on the VALU (exact_score); every wave keeps
// its own K best 64-bit keys (score, lowest index first) in registers, updated per round by a barrier-free
// ballot/popcount radix select; wave 0 merges the block's 16 lists, the last slice to finish merges the slices'
// lists and sorts the final K.  Cost is per user, independent of how many users need it (a 32-user group of the
// fp32 MFMA sweep costs ~0.5 ms of latency).
constexpr int kExThreads = 1024;
// D = 128 keeps twice the row pieces per lane: at 1024 threads (128 VGPRs) the kernel spilled 18 registers; 512-thread
// blocks get 256
template <int D>
constexpr int ex_threads() { return D <= 64 ? kExThreads : kExThreads / 2; }
constexpr int kExPer = 2;                         // keys per lane and round
constexpr int kExSlices = 8;                      // blocks per user: one CU's vector-memory path cannot stream the
                                                  // item table (a thread per row = 64 cache lines per load) fast enough

// The K largest of one wave's keys (NK per lane, 0 = none) plus its running best `bk` (one key per lane): radix
// select with ballots and scalar popcounts -- 32 steps on the score word, 32 more on the index word only when a tie
// straddles the K-th place -- no barriers.  Returns the lane's new running-best key (unordered over the lanes, 0 =
// empty).  Keys are unique, so exactly min(K, #keys) keys are >= the threshold found.  `stage` = 64 LDS slots of
// this wave.
template <int NK>
__device__ __forceinline__ uint64_t wave_select_topk(const uint64_t (&key)[NK], uint64_t bk, int K, uint64_t *stage) {
  const int lane = threadIdx.x & 63;
  int have = __popcll(__ballot(bk != 0ull));
#pragma unroll
  for (int j = 0; j < NK; ++j) have += __popcll(__ballot(key[j] != 0ull));
  const int want = min(K, have);
  uint32_t T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = T | (1u << bit);
    int c = __popcll(__ballot(bk != 0ull && (uint32_t)(bk >> 32) >= cand));
#pragma unroll
    for (int j = 0; j < NK; ++j) c += __popcll(__ballot(key[j] != 0ull && (uint32_t)(key[j] >> 32) >= cand));
    if (c >= want) T = cand;
  }
  int gt = __popcll(__ballot(bk != 0ull && (uint32_t)(bk >> 32) > T));
  int eq = __popcll(__ballot(bk != 0ull && (uint32_t)(bk >> 32) == T));
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    gt += __popcll(__ballot(key[j] != 0ull && (uint32_t)(key[j] >> 32) > T));
    eq += __popcll(__ballot(key[j] != 0ull && (uint32_t)(key[j] >> 32) == T));
  }
  const int need = want - gt;
  uint32_t L = 0;
  if (eq > need) {   // the `need` lowest indices among the ties = the largest inverted-index words
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = L | (1u << bit);
      int c = __popcll(__ballot(bk != 0ull && (uint32_t)(bk >> 32) == T && (uint32_t)bk >= cand));
#pragma unroll
      for (int j = 0; j < NK; ++j)
        c += __popcll(__ballot(key[j] != 0ull && (uint32_t)(key[j] >> 32) == T && (uint32_t)key[j] >= cand));
      if (c >= need) L = cand;
    }
  }
  const uint64_t thr = ((uint64_t)T << 32) | (uint64_t)L;
  // compact the winners into the wave's stage, lane i takes slot i
  stage[lane] = 0ull;
  __builtin_amdgcn_wave_barrier();
  int base = 0;
  {
    const bool w = bk != 0ull && bk >= thr;
    const unsigned long long m = __ballot(w);
    if (w) stage[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = bk;
    base = __popcll(m);
  }
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const bool w = key[j] != 0ull && key[j] >= thr;
    const unsigned long long m = __ballot(w);
    if (w) stage[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = key[j];
    base += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  const uint64_t out = stage[lane];
  __builtin_amdgcn_wave_barrier();
  return out;
}

// a: 64 keys in descending order, b: 64 keys in descending order (one per lane, 0 = none) -> the 64 largest of
// both, descending: b reversed against a gives a bitonic sequence of the lane-wise maxima, six merge stages order it.
__device__ __forceinline__ uint64_t merge_top64(uint64_t a, uint64_t b, int lane) {
  const uint64_t r = shfl_u64(b, 63 - lane);
  uint64_t m = a > r ? a : r;
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const uint64_t p = shfl_xor_u64(m, j);
    const uint64_t mx = m > p ? m : p, mn = m > p ? p : m;
    m = (lane & j) == 0 ? mx : mn;
  }
  return m;
}

template <int D>
__global__ __launch_bounds__(ex_threads<D>()) void score_exact_user_kernel(const PrefArgs P) {
  constexpr int kExThreads = ex_threads<D>();      // (shadows the namespace constant: this instantiation's block size)
  constexpr int NW = kExThreads / 64;
  __shared__ uint64_t stage[NW][64];         // per-wave staging / the waves' lists for the block merge
  __shared__ uint32_t hist_s[kPfHistLds];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = P.K;
  const int n_fb = *P.fb_cnt;
  // the queue: fb_list[fb_skip .. n_fb), then -- when pass A left only a handful of users uncertified and no retry pass
  // ran for them -- that retry queue as well (queue positions n_fb ..)
  int n_rt = P.retry_list ? *(P.retry_list_cnt) : 0;
  if (n_rt > P.small_retry) n_rt = 0;
  if (P.counters_out && blockIdx.x == 0 && tid == 0) {
    P.counters_out[0] = P.retry_list ? *(P.retry_list_cnt) : 0;
    P.counters_out[1] = n_fb;
    P.counters_out[2] = P.wide_cnt ? *P.wide_cnt : 0;
    P.counters_out[3] = n_rt;
  }
  const int64_t per_slice = (P.n_items + kExSlices - 1) / kExSlices;
  for (int w = blockIdx.x + P.fb_skip * kExSlices; w < (n_fb + n_rt) * kExSlices; w += gridDim.x) {
    const int qi = w / kExSlices, slice = w % kExSlices;
    const int64_t u = qi < n_fb ? P.fb_list[qi] : P.retry_list[qi - n_fb];
    const float *urow = P.user_emb + (size_t)u * D;
    int64_t hb = 0, he = 0;
    if (P.hist_rowptr) {
      hb = P.hist_rowptr[u];
      he = P.hist_rowptr[u + 1];
    }
    const int deg = (int)(he - hb);
    const bool hist_lds = deg <= kPfHistLds;
    __syncthreads();
    // (the history is only needed for the mask AFTER a round's scores: its loads -- two dependent round trips behind the
    //  user's id -- travel under the item rows' instead of in front of them)
    constexpr int HR = kPfHistLds / kExThreads;     // history entries per thread
    uint32_t hreg[HR];
#pragma unroll
    for (int q = 0; q < HR; ++q) hreg[q] = (hist_lds && tid + q * kExThreads < deg) ? (uint32_t)P.hist_col[hb + tid + q * kExThreads] : 0u;
    bool hist_ready = false;
    // this block's contiguous slice of the items, kExThreads * kExPer per round; every wave keeps its own K best
    const int64_t i_begin = (int64_t)slice * per_slice, i_end = min(P.n_items, i_begin + per_slice);
    // this lane's pieces of the user's row (exact_scores_64)
    float ua[D / 8], ub[D / 8];
#pragma unroll
    for (int i = 0; i < D / 8; ++i) {
      ua[i] = urow[16 * (i / 4) + 4 * (lane & 3) + (i & 3)];
      ub[i] = urow[D / 2 + 16 * (i / 4) + 4 * (lane & 3) + (i & 3)];
    }
    uint64_t bk = 0ull;
    for (int64_t c0 = i_begin; c0 < i_end; c0 += (int64_t)kExThreads * kExPer) {
      uint64_t key[kExPer];
      float sxs[kExPer];
#pragma unroll
      for (int j = 0; j < kExPer; ++j) {
        const int64_t it0 = c0 + 64 * wave + (int64_t)kExThreads * j;      // the wave's 64 consecutive items
        sxs[j] = 0.f;
        if (it0 < i_end) sxs[j] = exact_scores_64<D>(P.item_emb, it0, P.n_items, ua, ub, lane);   // wave-uniform
      }
      if (!hist_ready) {          // block-uniform: every thread's first round
#pragma unroll
        for (int q = 0; q < HR; ++q)
          if (tid + q * kExThreads < deg) hist_s[tid + q * kExThreads] = hreg[q];
        __syncthreads();
        hist_ready = true;
      }
#pragma unroll
      for (int j = 0; j < kExPer; ++j) {
        const int64_t it0 = c0 + 64 * wave + (int64_t)kExThreads * j;
        const int64_t it = it0 + lane;
        key[j] = 0ull;
        if (it0 < i_end) {
          const float sx = sxs[j];
          if (it < i_end) {
            const uint32_t item = (uint32_t)it;
            int lo = 0, hi = deg;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              const uint32_t hv = hist_lds ? hist_s[mid] : (uint32_t)P.hist_col[hb + mid];
              if (hv < item) lo = mid + 1; else hi = mid;
            }
            const bool masked = lo < deg && (hist_lds ? hist_s[lo] : (uint32_t)P.hist_col[hb + lo]) == item;
            key[j] = make_key(masked ? P.mask_value : sx, item);
          }
        }
      }
      bk = wave_select_topk<kExPer>(key, bk, kMaxK, stage[wave]);   // (the 64 best: the next call's threshold wants more than K)
    }
    // block merge: every wave orders its 64 best, then a tree of pairwise merges through LDS (log2 NW levels of one
    // reverse + six stages each).  One wave picking the 64 best of all NW lists by a bitwise search (32 steps x NW + 1
    // ballots) was 20 of the route's 48 us.
    sort64_desc(bk, lane);
    stage[wave][lane] = bk;
    __syncthreads();
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
      uint64_t mine = 0ull;
      if (wave < half) mine = merge_top64(stage[wave][lane], stage[wave + half][lane], lane);
      __syncthreads();
      if (wave < half) stage[wave][lane] = mine;
      __syncthreads();
    }
    uint64_t *part = P.fb_partial + (size_t)qi * kExSlices * kMaxK;
    if (wave == 0) {
      part[slice * kMaxK + lane] = stage[0][lane];   // kMaxK == 64; descending
      __threadfence();
      if (lane == 0) last = atomicAdd(P.fb_done + qi, 1) == kExSlices - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last && wave == 0) {   // the last slice to arrive merges the user's kExSlices lists
      __threadfence();
      uint64_t sl[kExSlices];                                    // the slices' lists are in descending order; all loads
#pragma unroll                                                   // in flight together, not one round trip per merge
      for (int j = 0; j < kExSlices; ++j) sl[j] = __builtin_nontemporal_load(part + j * kMaxK + lane);
      uint64_t e = sl[0];
#pragma unroll
      for (int j = 1; j < kExSlices; ++j) e = merge_top64(e, sl[j], lane);
      if (lane < K) {
        const uint32_t item = 0xFFFFFFFFu - (uint32_t)(e & 0xFFFFFFFFull);
        P.out_idx[(size_t)u * K + lane] = (int64_t)item + P.id_offset;
        P.out_val[(size_t)u * K + lane] = ord_to_f32((uint32_t)(e >> 32));
      }
      // A user on this route had a threshold that cut too close: its next one is taken well below what a certified
      // user gets (the score of rank 2 K): the scores of ranks 32 and 64 extrapolated as far again below rank 64 --
      // about rank 100 on a level score curve, lower where the curve is steep.  (With "one float below the K-th best"
      // these users failed again at the next call, whichever way their scores moved, and the queue grew from call to
      // call.)  Any value is legal.
      if (P.hint_out) {
        const uint32_t o31 = (uint32_t)__shfl((int)(uint32_t)(e >> 32), 31, 64), o63 = (uint32_t)__shfl((int)(uint32_t)(e >> 32), 63, 64);
        const uint64_t ek = shfl_u64(e, K - 1);
        float t = nextafterf(ord_to_f32((uint32_t)(ek >> 32)), -INFINITY);
        if (shfl_u64(e, 63) != 0ull) {
          const float s31 = ord_to_f32(o31), s63 = ord_to_f32(o63);
          t = fminf(t, s63 - (s31 - s63));
        }
        if (lane == 0) P.hint_out[u] = t;
      }
    }
  }
}

// ---- statistics of the last prefilter call (monitoring / tuning) ------------------------------------------------
// out[0] users sent to the exact route, out[1] candidates in total, out[2] longest sweep list (entries), out[3] users,
// out[4..8] uncertified users of the LAST selection pass by reason code 1..5 (select_user), out[9] users of pass C
__global__ __launch_bounds__(256) void score_prefilter_stats_kernel(const int *__restrict__ fail,
                                                                    const int *__restrict__ cand_cnt,
                                                                    const int *__restrict__ n_cand, int64_t n_users,
                                                                    int splits, unsigned long long *__restrict__ out,
                                                                    const int *__restrict__ rt2_cnt) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u == 0 && rt2_cnt) out[9] = (unsigned long long)*rt2_cnt;
  if (u >= n_users) return;
  int mx = 0;
  if (cand_cnt) {
    for (int l = 0; l < 2 * splits; ++l) mx = max(mx, cand_cnt[((size_t)(l >> 1) * n_users + u) * 2 + (l & 1)]);
  } else {
    mx = n_cand[u];          // block-joint selection: no per-lane lists; the longest key list instead
  }
  if (fail[u]) {
    atomicAdd(out + 0, 1ull);
    atomicAdd(out + 3 + min(fail[u], 5), 1ull);
  }
  atomicAdd(out + 1, (unsigned long long)n_cand[u]);
  atomicMax(out + 2, (unsigned long long)mx);
  atomicAdd(out + 3, 1ull);
}
