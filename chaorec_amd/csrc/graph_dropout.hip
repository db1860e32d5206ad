// Per-step edge dropout on a structure-static CSR in HBM.
//
//   * NGCF (Model/NGCF.py:38-58): every conv call drops each DIRECTED edge independently with probability p
//     (torch_geometric.utils.dropout_adj), appends the self loops, recounts degree(row) on what is left and
//     normalises norm_e = deg^-1/2[row] * deg^-1/2[col].  The reference rebuilds edge lists for that; here the CSR
//     structure (self loops included, built once) never changes and only its VALUE array is rewritten: a dropped
//     entry gets 0, a kept one the renormalised weight.  Two launches: integer degree count (one atomic per row segment of a wave), then the values of
//     A and of A^T (same structure, entry k <-> transpose_entry[k]) for the backward pass.
//   * FREEDOM (Model/FREEDOM.py:143-162): degree-sensitive pruning keeps a weighted sample WITHOUT replacement of
//     the training edges (torch.multinomial, limited to 2^24 categories).  Here: exponential-race keys
//     key_e = -log(u_e) / w_e, the k smallest win (the same law as sequential draws without replacement), found by
//     an exact 64-bit radix select with integer histograms -> deterministic for a given (seed, step), any edge count.
//
// Both are HBM-streaming integer/byte work: one coalesced pass per launch, no LDS tiles beyond the histograms.
#include "common.h"

namespace chaorec {

// uniform in [0,1) with 24 random bits (what torch.rand yields for fp32), from the stateless counter generator
__device__ __forceinline__ float edge_uniform(uint64_t seed, uint64_t step, uint64_t salt, uint64_t k) {
  const uint64_t h = mix64(seed ^ mix64(step ^ mix64((salt << 48) ^ k)));
  return (float)(uint32_t)(h >> 40) * 5.9604644775390625e-08f;  // 2^-24, exact
}

__device__ __forceinline__ bool edge_kept(const int32_t *erow, const int32_t *col, const uint8_t *keep_in, int64_t k,
                                          float p, uint64_t seed, uint64_t step, uint64_t salt) {
  if (erow[k] == col[k]) return true;             // self loops are appended AFTER the dropout: never dropped
  if (keep_in) return keep_in[k] != 0;
  return edge_uniform(seed, step, salt, (uint64_t)k) >= p;   // dropout_adj: mask = rand(E) >= p
}

// degree(row, ...) counts SOURCES (edge_index[0]).  Entry k of the destination-major CSR has source col[k]; the
// structure is symmetric, so the entries whose source is n are the reversed edges of row n's entries:
//   deg[n] = #{k in row n : keep(transpose_entry[k])}
// One thread per entry; the entries of a row are consecutive, so a wave first adds up each row segment it holds
// (two ballots + a popcount) and issues ONE atomic per segment -- a popular item costs ~deg/64 atomics, not deg.
__global__ __launch_bounds__(256) void edge_dropout_degree_kernel(
    const int32_t *__restrict__ erow, const int32_t *__restrict__ col, const int32_t *__restrict__ tentry,
    const uint8_t *__restrict__ keep_in, int64_t nnz, float p, uint64_t seed, uint64_t step,
    const int64_t *__restrict__ step_dev, uint64_t salt, int32_t *__restrict__ deg) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (step_dev) step += (uint64_t)step_dev[0];
  const bool in = k < nnz;
  const int row = in ? erow[k] : -1;
  const bool kept = in && edge_kept(erow, col, keep_in, tentry[k], p, seed, step, salt);
  const int prev = __shfl_up(row, 1, 64);
  const bool head = in && (lane == 0 || prev != row);
  const uint64_t kept_m = __ballot(kept), head_m = __ballot(head);
  if (head) {
    const uint64_t later = lane == 63 ? 0ull : (head_m >> (lane + 1));
    const int end = later ? lane + 1 + __builtin_ctzll(later) : 64;             // next segment's first lane
    const uint64_t seg = (end == 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
    const int cnt = __popcll(kept_m & seg);
    if (cnt) atomicAdd(deg + row, cnt);
  }
}

__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t *__restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(256) void edge_dropout_norm_kernel(
    const int32_t *__restrict__ erow, const int32_t *__restrict__ col, const int32_t *__restrict__ tentry,
    const uint8_t *__restrict__ keep_in, int64_t nnz, float p, uint64_t seed, uint64_t step,
    const int64_t *__restrict__ step_dev, uint64_t salt, const int32_t *__restrict__ deg,
    float *__restrict__ val, float *__restrict__ val_t) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  if (step_dev) step += (uint64_t)step_dev[0];
  const float ds = 1.0f / sqrtf((float)deg[col[k]]);    // deg.pow(-0.5): correctly rounded sqrt, then division
  const float dd = 1.0f / sqrtf((float)deg[erow[k]]);
  const float w = mul_rn(ds, dd);
  val[k] = edge_kept(erow, col, keep_in, k, p, seed, step, salt) ? w : 0.f;
  // A^T in the SAME (rowptr, col) structure: its entry k is A's entry tentry[k] (the reversed edge)
  val_t[k] = edge_kept(erow, col, keep_in, tentry[k], p, seed, step, salt) ? w : 0.f;
}

// ---- weighted sampling without replacement ----------------------------------------------------------------
// 64-bit key of entry e: high 32 bits = fp32 bits of -log(u)/w (positive -> unsigned order == float order), low 32
// bits = fresh hash bits that order entries whose fp32 keys coincide.  w <= 0 -> never selected.
__device__ __forceinline__ uint64_t race_key_of(float we, int64_t e, uint64_t seed, uint64_t step) {
  const uint64_t h = mix64(seed ^ mix64(step ^ mix64(0x5A3Bull << 48 ^ (uint64_t)e)));
  if (!(we > 0.f)) return ~0ull;
  const float u = ((float)(uint32_t)(h >> 40) + 0.5f) * 5.9604644775390625e-08f;   // (0,1): log finite
  const float key = fabsf(logf(u)) / we;          // >= +0: unsigned order of the bits == float order
  return ((uint64_t)__float_as_uint(key) << 32) | (uint32_t)h;
}
__device__ __forceinline__ uint64_t race_key(const float *__restrict__ w, int64_t e, uint64_t seed, uint64_t step) {
  return race_key_of(w[e], e, seed, step);
}

constexpr int kRaceBits = 11, kRaceBins = 1 << kRaceBits, kRacePasses = 6;   // 6 x 11 >= 64

// state[0] = prefix found so far (the high bits of the k-th smallest key), state[1] = how many keys with that
// prefix are still wanted.  Pass j histograms digit j (from the top) of the keys that match the prefix.
__global__ __launch_bounds__(256) void race_histogram_kernel(const float *__restrict__ w, int64_t n, uint64_t seed,
                                                             uint64_t step, const int64_t *__restrict__ step_dev,
                                                             int pass, const uint64_t *__restrict__ state,
                                                             uint32_t *__restrict__ hist) {
  __shared__ uint32_t h_s[kRaceBins];
  for (int i = threadIdx.x; i < kRaceBins; i += blockDim.x) h_s[i] = 0;
  __syncthreads();
  if (step_dev) step += (uint64_t)step_dev[0];
  const int shift = 64 - kRaceBits * (pass + 1);           // may be negative on the last pass (66 > 64)
  const uint64_t prefix = state[0];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = race_key(w, e, seed, step);
    const int hs = shift + kRaceBits;                        // bits above this digit
    const bool match = pass == 0 || (hs >= 64 ? true : (key >> hs) == prefix);
    if (match) {
      const uint32_t digit = (uint32_t)(shift >= 0 ? (key >> shift) : (key << -shift)) & (kRaceBins - 1);
      atomicAdd(&h_s[digit], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRaceBins; i += blockDim.x)
    if (h_s[i]) atomicAdd(hist + i, h_s[i]);
}

// one wave: find the bin the k-th smallest key falls into.  Lane l owns bins [32 l, 32 l + 32): lane totals, an
// exclusive wave scan, then the owning lane walks its 32 bins.
__global__ __launch_bounds__(64) void race_pick_kernel(int pass, uint64_t *__restrict__ state,
                                                       uint32_t *__restrict__ hist) {
  constexpr int PER = kRaceBins / 64;
  const int lane = threadIdx.x;
  uint32_t mine[PER];
  uint64_t tot = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    mine[j] = hist[lane * PER + j];
    tot += mine[j];
  }
  uint64_t incl = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  const uint64_t excl = incl - tot;
  const uint64_t want = state[1], prefix = pass == 0 ? 0 : state[0];
  // the owner: the first lane whose inclusive count reaches `want` (the last lane if none does: bin 2047, as before)
  const bool owner = (excl < want && incl >= want) || (lane == 63 && incl < want);
  if (owner) {
    uint64_t run = excl;
    int j = 0;
    for (; j < PER - 1; ++j) {
      if (run + mine[j] >= want) break;
      run += mine[j];
    }
    const int b = lane * PER + j;
    const int shift = 64 - kRaceBits * (pass + 1);
    // last pass: only the top (64 - 55) = 9 bits of the digit are real key bits (the digit was shifted LEFT)
    state[0] = shift >= 0 ? ((prefix << kRaceBits) | (uint64_t)b)
                          : ((prefix << (kRaceBits + shift)) | ((uint64_t)b >> -shift));
    state[1] = want - run;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) hist[lane * PER + j] = 0;   // ready for the next pass
}

__global__ __launch_bounds__(256) void race_keep_kernel(const float *__restrict__ w, int64_t n, uint64_t seed,
                                                        uint64_t step, const int64_t *__restrict__ step_dev,
                                                        const uint64_t *__restrict__ state,
                                                        uint8_t *__restrict__ keep, uint64_t *__restrict__ keys_out) {
  if (step_dev) step += (uint64_t)step_dev[0];
  const uint64_t kth = state[0];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = race_key(w, e, seed, step);
    keep[e] = key <= kth ? 1 : 0;
    if (keys_out) keys_out[e] = key;
  }
}

// keys only, entry j numbered ids[j] (a rank's share of an edge list keeps the numbers of the whole list: the keys,
// and with them the kept set, do not depend on how the edges are spread over the ranks)
__global__ __launch_bounds__(256) void race_keys_kernel(const float *__restrict__ w, const int64_t *__restrict__ ids,
                                                        int64_t n, uint64_t seed, uint64_t step,
                                                        const int64_t *__restrict__ step_dev,
                                                        uint64_t *__restrict__ keys_out) {
  if (step_dev) step += (uint64_t)step_dev[0];
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
    keys_out[j] = race_key_of(w[j], ids ? ids[j] : j, seed, step);
}

__global__ void race_init_kernel(uint64_t *state, uint32_t *hist, int64_t k) {
  for (int i = threadIdx.x; i < kRaceBins; i += blockDim.x) hist[i] = 0;
  if (threadIdx.x == 0) {
    state[0] = 0;
    state[1] = (uint64_t)k;
  }
}

}  // namespace chaorec

using namespace chaorec;

extern "C" int chaorec_edge_dropout_norm(const int32_t *entry_row, const int32_t *col, const int32_t *transpose_entry,
                                         int64_t nnz, int64_t n_nodes, float p, uint64_t seed, uint64_t step,
                                         const int64_t *step_dev, uint32_t salt, const uint8_t *keep_in,
                                         int32_t *deg_ws, float *val, float *val_t, void *stream) {
  if (!entry_row || !col || !transpose_entry || !deg_ws || !val || !val_t)
    return fail(CHAOREC_E_INVALID, "edge_dropout_norm: null pointer");
  if (nnz < 0 || n_nodes <= 0 || !(p >= 0.f && p < 1.f) || salt >= 65536u)
    return fail(CHAOREC_E_INVALID, "edge_dropout_norm: nnz=%lld n_nodes=%lld p=%g salt=%u", (long long)nnz,
                (long long)n_nodes, (double)p, salt);
  hipStream_t st = (hipStream_t)stream;
  // (a kernel, not hipMemsetAsync: this call sits inside captured training steps, and a memset NODE of this size
  // was observed not to take effect on replay -- the degrees then accumulate from step to step)
  zero_i32_kernel<<<(unsigned)((n_nodes + 255) / 256), 256, 0, st>>>(deg_ws, n_nodes);
  if (nnz == 0) return check_launch("edge_dropout_norm");
  const unsigned blocks = (unsigned)((nnz + 255) / 256);
  edge_dropout_degree_kernel<<<blocks, 256, 0, st>>>(entry_row, col, transpose_entry, keep_in, nnz, p, seed, step,
                                                     step_dev, salt, deg_ws);
  edge_dropout_norm_kernel<<<blocks, 256, 0, st>>>(entry_row, col, transpose_entry, keep_in, nnz, p, seed, step,
                                                   step_dev, salt, deg_ws, val, val_t);
  return check_launch("edge_dropout_norm");
}

extern "C" size_t chaorec_weighted_sample_workspace_bytes(void) {
  return 2 * sizeof(uint64_t) + kRaceBins * sizeof(uint32_t);
}

extern "C" int chaorec_weighted_sample_keep(const float *weights, int64_t n, int64_t k, uint64_t seed, uint64_t step,
                                            const int64_t *step_dev, void *workspace, size_t workspace_bytes,
                                            uint8_t *keep, uint64_t *keys_out, void *stream) {
  if (!weights || !keep || !workspace) return fail(CHAOREC_E_INVALID, "weighted_sample_keep: null pointer");
  if (n <= 0 || k <= 0 || k > n) return fail(CHAOREC_E_INVALID, "weighted_sample_keep: n=%lld k=%lld", (long long)n, (long long)k);
  if (workspace_bytes < chaorec_weighted_sample_workspace_bytes())
    return fail(CHAOREC_E_WORKSPACE, "weighted_sample_keep: workspace %zu < %zu", workspace_bytes,
                chaorec_weighted_sample_workspace_bytes());
  hipStream_t st = (hipStream_t)stream;
  uint64_t *state = (uint64_t *)workspace;
  uint32_t *hist = (uint32_t *)(state + 2);
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  race_init_kernel<<<1, 256, 0, st>>>(state, hist, k);
  for (int pass = 0; pass < kRacePasses; ++pass) {
    race_histogram_kernel<<<blocks, 256, 0, st>>>(weights, n, seed, step, step_dev, pass, state, hist);
    race_pick_kernel<<<1, 64, 0, st>>>(pass, state, hist);
  }
  race_keep_kernel<<<blocks, 256, 0, st>>>(weights, n, seed, step, step_dev, state, keep, keys_out);
  return check_launch("weighted_sample_keep");
}

extern "C" int chaorec_weighted_sample_keys(const float *weights, const int64_t *ids, int64_t n, uint64_t seed,
                                            uint64_t step, const int64_t *step_dev, uint64_t *keys_out, void *stream) {
  if (!weights || !keys_out) return fail(CHAOREC_E_INVALID, "weighted_sample_keys: null pointer");
  if (n < 0) return fail(CHAOREC_E_INVALID, "weighted_sample_keys: n=%lld", (long long)n);
  if (n == 0) return CHAOREC_OK;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  race_keys_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(weights, ids, n, seed, step, step_dev, keys_out);
  return check_launch("weighted_sample_keys");
}
