// Time from the host's launch call to the first kernel's first store becoming visible, and the time four ~10 us kernels take end to end:
// four direct launches on one stream against one hipGraphLaunch of the same four kernels (the captured pose-graph LM trial is four kernels).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/launch_latency.hip -o /tmp/ll && /tmp/ll
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(volatile unsigned* flag, unsigned v, int at_start, long spin_clocks) {
  if (at_start && threadIdx.x == 0 && blockIdx.x == 0) { *flag = v; __threadfence_system(); }
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_clocks) {}
  if (!at_start && threadIdx.x == 0 && blockIdx.x == 0) { *flag = v; __threadfence_system(); }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned* flag;
  CK(hipHostMalloc(&flag, 64, hipHostMallocMapped));
  *flag = 0;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const long spin = 1000;                       // 100 MHz clock: 10 us per kernel
  unsigned* dflag = flag;
  for (int where = 0; where < 2; ++where) {     // 0: the FIRST kernel's start reports; 1: the LAST kernel's end
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 4; ++k) {
      const bool mark = where == 0 ? k == 0 : k == 3;
      hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, mark ? dflag : dflag + 8, 0u, where == 0, spin);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    // (the captured kernels write 0: replays signal by a separate counter kernel argument being fixed; use a per-iteration reset instead)
    for (int mode = 0; mode < 2; ++mode) {      // 0 direct, 1 graph
      std::vector<double> lat, call;
      for (unsigned it = 1; it <= 300; ++it) {
        *flag = 0xffffffffu;
        CK(hipStreamSynchronize(st));
        const double t0 = now();
        if (mode == 0) {
          for (int k = 0; k < 4; ++k) {
            const bool mark = where == 0 ? k == 0 : k == 3;
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, mark ? dflag : dflag + 8, 0u, where == 0, spin);
          }
        } else {
          CK(hipGraphLaunch(ge, st));
        }
        const double t1 = now();
        while (*(volatile unsigned*)flag != 0u) {}
        const double t2 = now();
        if (it > 50) { lat.push_back(t2 - t0); call.push_back(t1 - t0); }
      }
      std::sort(lat.begin(), lat.end()); std::sort(call.begin(), call.end());
      printf("%s  %s: host call %.1f us, until the mark %.1f us (median of %zu)\n", where == 0 ? "first kernel starts" : "fourth kernel ends ",
             mode == 0 ? "4 direct launches" : "hipGraphLaunch    ", call[call.size() / 2], lat[lat.size() / 2], lat.size());
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
