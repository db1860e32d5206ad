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
