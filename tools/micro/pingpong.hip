// Latency of one cross-workgroup hand-off through a tagged 64-bit word (agent-scope store -> agent-scope polling load), as a
// function of where the two workgroups run:   hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o tools/micro/build/pingpong
// Workgroups are dispatched round-robin over the 8 XCDs (workgroup i -> XCD i % 8); each XCD has its own L2.  The persistent
// PCG (csrc/pcg_persist.hip) pays ~4 such hand-offs per iteration: this is the number that bounds it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__device__ __forceinline__ void st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void pingpong(u64* word, int a, int b, int iters, unsigned* xcc) {
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[blockIdx.x == (unsigned)a ? 0 : 1] = id;
  const bool first = (int)blockIdx.x == a;
  for (int it = 0; it < iters; ++it) {
    if (first) {
      st(word, 2ull * it + 1);
      long spin = 0;
      while (ld(word) != 2ull * it + 2 && ++spin < (1L << 24)) {}
    } else {
      long spin = 0;
      while (ld(word) != 2ull * it + 1 && ++spin < (1L << 24)) {}
      st(word, 2ull * it + 2);
    }
  }
}
// all-gather of one tagged word per workgroup: every workgroup publishes, then polls everybody's (64 lanes, rows strided)
__global__ void allgather(u64* tab, int iters, float* sink) {
  float acc = 0;
  for (int it = 1; it <= iters; ++it) {
    if (threadIdx.x == 0) st(tab + (size_t)(it & 1) * 1024 + blockIdx.x, ((u64)it << 32) | blockIdx.x);
    for (int r = threadIdx.x; r < (int)gridDim.x; r += 64) {
      u64 v;
      long spin = 0;
      do { v = ld(tab + (size_t)(it & 1) * 1024 + r); } while ((unsigned)(v >> 32) != (unsigned)it && ++spin < (1L << 22));
      acc += (float)(unsigned)v;
    }
    __syncthreads();
  }
  if (acc < 0) sink[0] = acc;
}
int main() {
  u64* word; unsigned* xcc; float* sink; u64* tab;
  hipMalloc(&word, 4096); hipMalloc(&xcc, 64); hipMalloc(&sink, 4); hipMalloc(&tab, 2 * 1024 * 8);
  const int iters = 2000;
  for (int b : {1, 2, 4, 7, 8, 16, 64, 128}) {
    hipMemset(word, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pingpong<<<256, 64>>>(word, 0, b, 10, xcc); hipDeviceSynchronize(); hipMemset(word, 0, 4096);
    hipEventRecord(e0); pingpong<<<256, 64>>>(word, 0, b, iters, xcc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2]; hipMemcpy(h, xcc, 8, hipMemcpyDeviceToHost);
    printf("pingpong wg 0 <-> wg %3d  (XCC_ID %u <-> %u): %.3f us per round trip (2 hand-offs)\n", b, h[0] & 0xf, h[1] & 0xf, ms * 1e3 / iters);
  }
  for (int grid : {8, 32, 64, 128, 256}) {
    hipMemset(tab, 0, 2 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); allgather<<<grid, 64>>>(tab, 1000, sink); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("all-gather of one tagged word per workgroup, grid %3d: %.3f us per exchange\n", grid, ms * 1e3 / 1000);
  }
  return 0;
}
