// What does a device-wide barrier inside ONE persistent launch cost against the kernel boundaries it would replace?
// (DESIGN: "one persistent launch per direction walking all L layers" for the cache-resident sports step.)
//
// Workload shape = the sports SpMM: 361 255 output rows of 256 B, 16 rows per 256-thread workgroup -> 22 579 row groups,
// every phase WRITES all rows (12 MB of dirty lines, like a layer's output) and READS 8 pseudo-random rows of the
// previous phase's output per row (so a phase really depends on the one before it, through L2 / Infinity Cache).
//   A  `phases` dependent launches of that grid (what the step does today), back to back on one stream;
//   B  one persistent launch, 4 workgroups per CU, grid-stride over the row groups, a device-wide barrier between the
//      phases: lane-0 agent-scope release fence -> counter -> spin on an sc1 load with s_sleep -> agent-scope acquire
//      (MI355X_MICROARCH.md "barrier-counter");
//   C  as B with the XCD-hierarchical form (per-XCC counter, the XCD's last arriver goes to the top counter).
// Prints microseconds per `phases`-sequence.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRows = 361255, kRowF4 = 16;   // 256-B rows as 16 float4
constexpr int kGroups = (kRows + 15) / 16;

__device__ __forceinline__ void phase_rows(const float4 *__restrict__ src, float4 *__restrict__ dst, int group, int phase) {
  const int lane16 = threadIdx.x & 15, r = group * 16 + (threadIdx.x >> 4);
  if (r >= kRows) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned h = (unsigned)r * 2654435761u + (unsigned)phase * 40503u;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h = h * 1664525u + 1013904223u;
    const float4 v = src[(size_t)(h % kRows) * kRowF4 + lane16];
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  dst[(size_t)r * kRowF4 + lane16] = make_float4(acc.x * 0.125f, acc.y * 0.125f, acc.z * 0.125f, acc.w * 0.125f);
}

__global__ __launch_bounds__(256) void one_phase(const float4 *src, float4 *dst, int phase) { phase_rows(src, dst, blockIdx.x, phase); }

__device__ __forceinline__ void grid_barrier_flat(unsigned *counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
      __builtin_amdgcn_s_sleep(2);       // (bounded: a protocol error ends in wrong data, not in a hung GPU)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// per-XCC counters [8] (each on its own 128-B line), top counter, per-XCC generation words
__device__ __forceinline__ void grid_barrier_xcd(unsigned *xcc_cnt, unsigned *top, unsigned *gen, unsigned per_xcc_target,
                                                 unsigned top_target, unsigned phase_no, unsigned xcc) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned mine = __hip_atomic_fetch_add(xcc_cnt + 32 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mine + 1 == per_xcc_target * phase_no) {            // this XCD's last arriver goes to the top
      __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < top_target; ++spin)
        __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(gen + 32 * xcc, phase_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(gen + 32 * xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase_no; ++spin)
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <int MODE>   // 1 flat, 2 XCD-hierarchical, 0 no barrier at all (wrong results: the floor of the persistent form)
__global__ __launch_bounds__(256) void persistent(float4 *a, float4 *b, int phases, unsigned *sync, unsigned epoch,
                                                  unsigned *xcc_census) {
  unsigned xcc = 0;
  if (MODE == 2) {
    xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;   // HW_REG_XCC_ID, bits [3:0]
  }
  float4 *src = a, *dst = b;
  for (int p = 0; p < phases; ++p) {
    for (int g = blockIdx.x; g < kGroups; g += gridDim.x) phase_rows(src, dst, g, p);
    if (MODE == 1) grid_barrier_flat(sync, (epoch * phases + p + 1) * gridDim.x);
    if (MODE == 2) {
      // census of workgroups per XCC (first launch only decides the targets; here: read from xcc_census)
      grid_barrier_xcd(sync + 64, sync, sync + 64 + 8 * 32, xcc_census[xcc], (epoch * phases + p + 1) * 8u,
                       epoch * phases + p + 1, xcc);
    }
    float4 *t = src;
    src = dst;
    dst = t;
  }
}

__global__ void census(unsigned *xcc_census) {
  if (threadIdx.x == 0) atomicAdd(xcc_census + (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF), 1u);
}

int main() {
  const int phases = 6, reps = 200;
  float4 *a, *b;
  unsigned *sync, *cens;
  CK(hipMalloc(&a, (size_t)kRows * 256));
  CK(hipMalloc(&b, (size_t)kRows * 256));
  CK(hipMalloc(&sync, 4096 * 4));
  CK(hipMalloc(&cens, 16 * 4));
  CK(hipMemset(a, 0, (size_t)kRows * 256));
  CK(hipMemset(sync, 0, 4096 * 4));
  CK(hipMemset(cens, 0, 64));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  // A: dependent launches
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
      for (int p = 0; p < phases; ++p) one_phase<<<kGroups, 256, 0, st>>>(p & 1 ? b : a, p & 1 ? a : b, p);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("A  %d dependent launches (grid %d x 256)              %8.2f us per sequence\n", phases, kGroups, ms * 1e3 / reps);
  int per_cu = 0, cus = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent<1>, 256, 0));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  for (int wg_per_cu : {1, 2, 4}) {
    if (wg_per_cu > per_cu) continue;
    const int grid = wg_per_cu * cus;
    // the XCC census for this grid (round-robin placement is observed, not promised: count it)
    CK(hipMemsetAsync(cens, 0, 64, st));
    census<<<grid, 256, 0, st>>>(cens);
    CK(hipStreamSynchronize(st));
    unsigned hc[16];
    CK(hipMemcpy(hc, cens, 64, hipMemcpyDeviceToHost));
    for (int mode : {0, 1, 2}) {
      CK(hipMemsetAsync(sync, 0, 4096 * 4, st));
      for (int w = 0; w < 2; ++w) {
        CK(hipMemsetAsync(sync, 0, 4096 * 4, st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) {
          if (mode == 0) persistent<0><<<grid, 256, 0, st>>>(a, b, phases, sync, (unsigned)r, cens);
          if (mode == 1) persistent<1><<<grid, 256, 0, st>>>(a, b, phases, sync, (unsigned)r, cens);
          if (mode == 2) persistent<2><<<grid, 256, 0, st>>>(a, b, phases, sync, (unsigned)r, cens);
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("%s  persistent, %d WG/CU (grid %4d), %-28s %8.2f us per sequence   (XCC census %u %u %u %u %u %u %u %u)\n",
             mode == 0 ? "B0" : (mode == 1 ? "B " : "C "), wg_per_cu, grid,
             mode == 0 ? "NO barrier (floor, wrong)" : (mode == 1 ? "flat counter barrier" : "XCD-hierarchical barrier"),
             ms * 1e3 / reps, hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
    }
  }
  return 0;
}
