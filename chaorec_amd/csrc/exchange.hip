// The item-partial exchange of a user-sharded GCN layer ("a direct reduce-scatter + all-gather that sends 1/8 of
// the buffer to each of the 7 peers simultaneously uses all links ... or a hand-written peer-to-peer RS+AG") as two plain
// kernels over IPC-mapped peer buffers -- capturable in a hipGraph like any launch, which RCCL's all-to-all is not on this
// stack.  Every rank keeps two "mailboxes" that its peers have mapped (hipIpcOpenMemHandle through torch's
// storage sharing): P, a copy of its [rows_pad, D] partial, and R, one row block long, for the block it reduces.  Between the
// phases the ranks meet at a barrier (a one-element RCCL all-reduce on the stream: a kernel boundary on every rank, so what a
// peer wrote before it is visible after it):
//     P <- my partial | barrier | pull_sum: R = sum over ranks of THEIR P's rows of my block, read over xGMI, rank order
//     0..W-1 (each block crosses each link once) | barrier | pull_gather: every rank's R into my buffer.
// Two barriers are enough for any sequence of exchanges: a rank rewrites P only after the second barrier of the previous
// exchange (every peer has finished pull_sum by then) and R only after the first barrier of this one (every peer has
// finished the previous pull_gather by then).
// The sum of a block is formed by exactly one rank, in a fixed order: all ranks end with the same bits.
#include "common.h"

namespace chaorec {

constexpr int kMaxPeers = 16;
struct PeerPtrs {
  float4 *p[kMaxPeers];
  int world;
};

// out[i] = ((p_0[off + i] + p_1[off + i]) + ...) for i < n4 (float4 units)
__global__ __launch_bounds__(256) void pull_sum_kernel(const PeerPtrs P, int64_t off4, int64_t n4, float4 *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = P.p[0][off4 + i];
    for (int r = 1; r < P.world; ++r) {
      const float4 v = P.p[r][off4 + i];
      a = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
    }
    out[i] = a;
  }
}

// out[r * block4 + i] = p_r[i]: block r comes from rank r's result mailbox (one block long)
__global__ __launch_bounds__(256) void pull_gather_kernel(const PeerPtrs P, int64_t block4, float4 *__restrict__ out) {
  const int64_t total = block4 * P.world;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / block4);
    out[i] = P.p[r][i - (int64_t)r * block4];
  }
}

// ---- the same exchange for a buffer that is non-zero in a FRONTIER's rows only (the row-sparse backward / light forward of a
// user shard: the gradient seed has <= 2 B world non-zero rows, a frontier partial ~1e4-1e5 of 2 M) ----------------------------
// `bits`: a bitmap over the buffer's rows, IDENTICAL on every rank (the union of the ranks' frontiers), a superset of the
// rows that are non-zero anywhere.  Only flagged rows are copied into the mailbox, summed and gathered; the other rows of the
// caller's buffer are not touched (they hold zeros on every rank, and zeros is what their sum is).  One wave per bitmap
// word; an empty word is a load and an exit.
__device__ __forceinline__ bool next_row(uint32_t &word, int64_t wi, int64_t &r) {
  if (!word) return false;
  const int b = __builtin_ctz(word);
  word &= word - 1;
  r = wi * 32 + b;
  return true;
}

__global__ __launch_bounds__(256) void rows_copy_kernel(float4 *__restrict__ dst, const float4 *__restrict__ src, int64_t n_rows,
                                                        int D4, const uint32_t *__restrict__ bits, int64_t n_words) {
  const int lane = threadIdx.x & 63;
  const int64_t wi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n_words) return;
  uint32_t word = bits[wi];
  int64_t r;
  while (next_row(word, wi, r)) {
    if (r >= n_rows) break;
    for (int c = lane; c < D4; c += 64) dst[(size_t)r * D4 + c] = src[(size_t)r * D4 + c];
  }
}

// R[r - row0] = ((p_0[r] + p_1[r]) + ...) for the flagged rows r of [row0, row0 + n_block)
__global__ __launch_bounds__(256) void pull_sum_rows_kernel(const PeerPtrs P, int64_t row0, int64_t n_block, int D4,
                                                            const uint32_t *__restrict__ bits, int64_t n_words,
                                                            float4 *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wi = (row0 >> 5) + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n_words || wi * 32 >= row0 + n_block) return;
  uint32_t word = bits[wi];
  int64_t r;
  while (next_row(word, wi, r)) {
    if (r < row0) continue;
    if (r >= row0 + n_block) break;
    for (int c = lane; c < D4; c += 64) {
      float4 a = P.p[0][(size_t)r * D4 + c];
      for (int q = 1; q < P.world; ++q) {
        const float4 v = P.p[q][(size_t)r * D4 + c];
        a = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
      }
      out[(size_t)(r - row0) * D4 + c] = a;
    }
  }
}

// out[r] = R_owner[r - owner * n_block] for every flagged row r (owner = r / n_block)
__global__ __launch_bounds__(256) void pull_gather_rows_kernel(const PeerPtrs P, int64_t n_block, int D4,
                                                               const uint32_t *__restrict__ bits, int64_t n_words,
                                                               float4 *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n_words) return;
  uint32_t word = bits[wi];
  int64_t r;
  while (next_row(word, wi, r)) {
    const int owner = (int)(r / n_block);
    if (owner >= P.world) break;
    for (int c = lane; c < D4; c += 64) out[(size_t)r * D4 + c] = P.p[owner][(size_t)(r - (int64_t)owner * n_block) * D4 + c];
  }
}

// ---- the compact form of a frontier exchange, for collectives that cannot skip rows (RCCL) ----------------------------------
// The bitmap is the same on every rank, so "flagged row number k in bitmap order" names the same row everywhere: the flagged rows
// are packed into a [cap, D] buffer in that order (unused tail rows zeroed), the SMALL buffer is all-reduced, and the sums are
// written back.  Only for frontiers with a static bound on their size (the batch items of all ranks: <= 2 B world rows) -- a
// captured step cannot choose its collective's size on the device.
// prefix[w] = number of flagged rows in words [0, w); prefix[n_words] = their total.  One workgroup per 1024 words, no
// hand-over between workgroups: workgroup b counts the words in front of its own (coalesced reads of a bitmap that sits in L2:
// 250 KB at 2 M rows, ~8 MB of reads over all workgroups) and scans its own 1024.  (One workgroup walking the whole bitmap,
// 61 words per thread, took 96 us per call at 2 M rows -- four calls per sharded light step.)
__global__ __launch_bounds__(1024) void bits_prefix_kernel(const uint32_t *__restrict__ bits, int64_t n_rows, int64_t n_words,
                                                           int32_t *__restrict__ prefix) {
  __shared__ int part[1024];
  __shared__ int wave_sum[16];
  const int t = threadIdx.x;
  auto count_at = [&](int64_t w) {
    uint32_t v = bits[w];
    const int64_t left = n_rows - w * 32;               // (bits past the last row are not rows)
    if (left < 32) v &= left <= 0 ? 0u : ((1u << left) - 1u);
    return __popc(v);
  };
  const int64_t w_first = (int64_t)blockIdx.x * 1024;
  int before = 0;                                        // flagged rows in words [0, w_first)
  for (int64_t w = t; w < w_first; w += 1024) before += count_at(w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if ((t & 63) == 0) wave_sum[t >> 6] = before;
  const int64_t w = w_first + t;
  const int mine = w < n_words ? count_at(w) : 0;
  part[t] = mine;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) base += wave_sum[k];
  for (int off = 1; off < 1024; off <<= 1) {            // inclusive scan of the workgroup's 1024 counts
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (w < n_words) prefix[w] = base + part[t] - mine;
  if (w == n_words - 1) prefix[n_words] = base + part[t];
}

// pack: compact[k] = src[row k of the bitmap] (k < cap), rows [total, cap) of compact zeroed; unpack: the inverse copy
template <bool PACK>
__global__ __launch_bounds__(256) void frontier_move_kernel(float4 *__restrict__ table, int64_t n_rows, int D4,
                                                            const uint32_t *__restrict__ bits, int64_t n_words,
                                                            const int32_t *__restrict__ prefix, float4 *__restrict__ compact,
                                                            int64_t cap) {
  const int lane = threadIdx.x & 63;
  const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot < n_words) {
    uint32_t word = bits[slot];
    const uint32_t all = word;
    int64_t r;
    while (next_row(word, slot, r)) {
      if (r >= n_rows) break;
      const int64_t k = prefix[slot] + __popc(all & ((1u << (r & 31)) - 1u));
      if (k >= cap) break;
      for (int c = lane; c < D4; c += 64) {
        if (PACK) compact[(size_t)k * D4 + c] = table[(size_t)r * D4 + c];
        else table[(size_t)r * D4 + c] = compact[(size_t)k * D4 + c];
      }
    }
  } else if (PACK) {                                     // the waves past the bitmap zero the unused tail of the compact buffer
    const int64_t total = prefix[n_words];
    const int64_t k = total + (slot - n_words);
    if (k < cap)
      for (int c = lane; c < D4; c += 64) compact[(size_t)k * D4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace chaorec

using namespace chaorec;

static int fill_peers(PeerPtrs &P, const void *const *peers, int32_t world, const char *who) {
  if (!peers || world < 1 || world > kMaxPeers) return fail(CHAOREC_E_INVALID, "%s: world=%d must be in [1, %d]", who, world, kMaxPeers);
  P.world = world;
  for (int r = 0; r < kMaxPeers; ++r) {
    const void *q = peers[r < world ? r : 0];
    if (!q || (reinterpret_cast<uintptr_t>(q) & 15)) return fail(CHAOREC_E_INVALID, "%s: peer %d NULL or not 16-byte aligned", who, r);
    P.p[r] = (float4 *)const_cast<void *>(q);
  }
  return CHAOREC_OK;
}

extern "C" int chaorec_exchange_pull_sum_f32(const void *const *peers, int32_t world, int64_t offset, int64_t n, float *out,
                                             void *stream) {
  if (!out || offset < 0 || n < 0 || (offset & 3) || (n & 3) || (reinterpret_cast<uintptr_t>(out) & 15))
    return fail(CHAOREC_E_INVALID, "exchange_pull_sum: offset=%lld n=%lld must be multiples of 4, out 16-byte aligned", (long long)offset, (long long)n);
  PeerPtrs P;
  int rc = fill_peers(P, peers, world, "exchange_pull_sum");
  if (rc || n == 0) return rc;
  const int64_t n4 = n / 4;
  const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 4096);
  hipLaunchKernelGGL(pull_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, offset / 4, n4, (float4 *)out);
  return check_launch("pull_sum_kernel");
}

extern "C" int chaorec_exchange_pull_gather_f32(const void *const *peers, int32_t world, int64_t block, float *out,
                                                void *stream) {
  if (!out || block < 0 || (block & 3) || (reinterpret_cast<uintptr_t>(out) & 15))
    return fail(CHAOREC_E_INVALID, "exchange_pull_gather: block=%lld must be a multiple of 4, out 16-byte aligned", (long long)block);
  PeerPtrs P;
  int rc = fill_peers(P, peers, world, "exchange_pull_gather");
  if (rc || block == 0) return rc;
  const int64_t total4 = block / 4 * world;
  const unsigned blocks = (unsigned)std::min<int64_t>((total4 + 255) / 256, 4096);
  hipLaunchKernelGGL(pull_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, block / 4, (float4 *)out);
  return check_launch("pull_gather_kernel");
}

static int rows_args(const void *a, const void *bits, int64_t n_rows, int32_t D, const char *who) {
  if (!a || !bits) return fail(CHAOREC_E_INVALID, "%s: NULL argument", who);
  if (n_rows <= 0 || D <= 0 || (D & 3) || (reinterpret_cast<uintptr_t>(a) & 15))
    return fail(CHAOREC_E_INVALID, "%s: n_rows=%lld D=%d (a multiple of 4), 16-byte aligned buffers", who, (long long)n_rows, D);
  return CHAOREC_OK;
}

extern "C" int chaorec_rows_copy_by_bits_f32(float *dst, const float *src, int64_t n_rows, int32_t D, const uint32_t *bits,
                                             void *stream) {
  int rc = rows_args(dst, bits, n_rows, D, "rows_copy_by_bits");
  if (rc) return rc;
  if (!src || (reinterpret_cast<uintptr_t>(src) & 15)) return fail(CHAOREC_E_INVALID, "rows_copy_by_bits: src NULL or unaligned");
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL(rows_copy_kernel, dim3((unsigned)((n_words + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (float4 *)dst,
                     (const float4 *)src, n_rows, D / 4, bits, n_words);
  return check_launch("rows_copy_kernel");
}

extern "C" int chaorec_exchange_pull_sum_rows_f32(const void *const *peers, int32_t world, int64_t row0, int64_t n_block,
                                                  int64_t n_rows, int32_t D, const uint32_t *bits, float *out, void *stream) {
  int rc = rows_args(out, bits, n_rows, D, "exchange_pull_sum_rows");
  if (rc) return rc;
  if (row0 < 0 || n_block <= 0) return fail(CHAOREC_E_INVALID, "exchange_pull_sum_rows: row0=%lld n_block=%lld", (long long)row0, (long long)n_block);
  PeerPtrs P;
  rc = fill_peers(P, peers, world, "exchange_pull_sum_rows");
  if (rc) return rc;
  const int64_t n_words = (n_rows + 31) / 32;                    // (bits over the rows that can be flagged at all: pad rows never are)
  const int64_t span = (row0 + n_block + 31) / 32 - row0 / 32;   // words that overlap my block
  hipLaunchKernelGGL(pull_sum_rows_kernel, dim3((unsigned)((span + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, row0, n_block,
                     D / 4, bits, n_words, (float4 *)out);
  return check_launch("pull_sum_rows_kernel");
}

extern "C" int chaorec_exchange_pull_gather_rows_f32(const void *const *peers, int32_t world, int64_t n_block, int64_t n_rows,
                                                     int32_t D, const uint32_t *bits, float *out, void *stream) {
  int rc = rows_args(out, bits, n_rows, D, "exchange_pull_gather_rows");
  if (rc) return rc;
  if (n_block <= 0) return fail(CHAOREC_E_INVALID, "exchange_pull_gather_rows: n_block=%lld", (long long)n_block);
  PeerPtrs P;
  rc = fill_peers(P, peers, world, "exchange_pull_gather_rows");
  if (rc) return rc;
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL(pull_gather_rows_kernel, dim3((unsigned)((n_words + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, n_block,
                     D / 4, bits, n_words, (float4 *)out);
  return check_launch("pull_gather_rows_kernel");
}

extern "C" int chaorec_frontier_pack_f32(const float *src, int64_t n_rows, int32_t D, const uint32_t *bits, int32_t *prefix,
                                         float *compact, int64_t cap, void *stream) {
  int rc = rows_args(compact, bits, n_rows, D, "frontier_pack");
  if (rc) return rc;
  if (!src || !prefix || cap <= 0 || (reinterpret_cast<uintptr_t>(src) & 15))
    return fail(CHAOREC_E_INVALID, "frontier_pack: NULL / unaligned argument or cap=%lld", (long long)cap);
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL(bits_prefix_kernel, dim3((unsigned)((n_words + 1023) / 1024)), dim3(1024), 0, (hipStream_t)stream, bits, n_rows, n_words,
                     prefix);
  rc = check_launch("bits_prefix_kernel");
  if (rc) return rc;
  const int64_t slots = n_words + cap;                    // (a wave per bitmap word + a wave per possible tail row)
  hipLaunchKernelGGL((frontier_move_kernel<true>), dim3((unsigned)((slots + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (float4 *)const_cast<float *>(src), n_rows, D / 4, bits, n_words, prefix, (float4 *)compact, cap);
  return check_launch("frontier_move_kernel<pack>");
}

extern "C" int chaorec_frontier_unpack_f32(float *dst, int64_t n_rows, int32_t D, const uint32_t *bits, const int32_t *prefix,
                                           const float *compact, int64_t cap, void *stream) {
  int rc = rows_args(dst, bits, n_rows, D, "frontier_unpack");
  if (rc) return rc;
  if (!compact || !prefix || cap <= 0 || (reinterpret_cast<uintptr_t>(compact) & 15))
    return fail(CHAOREC_E_INVALID, "frontier_unpack: NULL / unaligned argument or cap=%lld", (long long)cap);
  const int64_t n_words = (n_rows + 31) / 32;
  hipLaunchKernelGGL((frontier_move_kernel<false>), dim3((unsigned)((n_words + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (float4 *)dst, n_rows, D / 4, bits, n_words, prefix, (float4 *)const_cast<float *>(compact), cap);
  return check_launch("frontier_move_kernel<unpack>");
}
