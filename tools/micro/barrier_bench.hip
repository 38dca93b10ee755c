// Cost of the grid barrier of csrc/gridsync.h as a function of the grid size (hipcc tools/micro/barrier_bench.hip -o /tmp/bb).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pypose_amd/csrc/gridsync.h"
__global__ void k2(unsigned* bar, int iters, float* sink) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    pplie::grid_rendezvous(bar);
    v += 1.0f;
  }
  if (v < 0) sink[0] = v;
}
__global__ void k(unsigned* bar, int iters, float* sink) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    pplie::grid_barrier(bar, bar + 1);
    v += 1.0f;
  }
  if (v < 0) sink[0] = v;
}
int main() {
  unsigned* bar; float* sink;
  hipMalloc(&bar, 8192); hipMemset(bar, 0, 8192); hipMalloc(&sink, 4);
  for (int block : {256, 1024})
    for (int grid : {8, 16, 32, 64, 128, 256}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      const int iters = 200;
      k<<<grid, block>>>(bar, 10, sink); hipDeviceSynchronize();
      hipEventRecord(a); k<<<grid, block>>>(bar, iters, sink); hipEventRecord(b); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("block %4d grid %3d: %.2f us per fenced barrier", block, grid, ms * 1e3 / iters);
      k2<<<grid, block>>>(bar, 10, sink); hipDeviceSynchronize();
      hipEventRecord(a); k2<<<grid, block>>>(bar, iters, sink); hipEventRecord(b); hipDeviceSynchronize();
      hipEventElapsedTime(&ms, a, b);
      printf(", %.2f us per two-level rendezvous without cache maintenance\n", ms * 1e3 / iters);
    }
  return 0;
}
