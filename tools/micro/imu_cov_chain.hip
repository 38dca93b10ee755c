// What does the IMU covariance kernel cost when its bytes are free?  ("next" item 5: "commit a microbenchmark that
// reproduces the kernel's dependency chain at its instruction mix".)
//   python tools/build_micro.py imu_cov_chain        (hipcc with the library's flags, -I pypose_amd/csrc)
//   tools/micro/build/imu_cov_chain [B F reps]       -> one JSON line
// Runs csrc/scan.hip's imu_cov_seg_kernel<float, 2, 4, ISO> itself -- the library's instantiation on real streams of
// [B, F] inputs, then the SAME instantiation with in_mask = 63 (sequence b reads the input streams of sequence b & 63: one scalar AND
// on the row address; 2.9 MB of inputs for all B sequences, cache-resident after the first touch, instead of 185 MB from memory;
// identical instructions, registers, walks, scans, LDS slots, accumulators, final reduction and store).  The second time is
// the floor of every variant that only re-arranges this kernel's memory traffic (reading the inputs once in a kernel fused with the
// integration, staging through LDS differently, ...): the work itself -- two walks of four steps per lane around one cross-lane scan
// per 256 steps, at three waves per SIMD -- takes that long.
#include "../../pypose_amd/csrc/scan.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float run(int64_t in_mask, const float* dt, const float* gy, const float* ac, const float* ro, const float* init_cov, const float* gc, const float* acov,
                 float* cov, int64_t B, int64_t F, int reps) {
  constexpr int WAVES = 2;
  const int64_t blocks = (B + WAVES - 1) / WAVES;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < reps + 2; ++r) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((pplie::imu_cov_seg_kernel<float, WAVES, 4, true>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, 0, dt, gy, ac, ro, ro,
                       (const float*)nullptr, init_cov, gc, (int64_t)0, (int64_t)0, acov, (int64_t)0, (int64_t)0, 0.f, 0.f, 9.81f, cov, B, F, in_mask);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    if (r >= 2 && ms < best) best = ms;
  }
  return best * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096, F = argc > 2 ? atoll(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const size_t n = (size_t)B * F;
  std::vector<float> h_dt(n, 0.005f), h_gy(n * 3), h_ac(n * 3), h_ro(n * 4);
  unsigned s = 12345u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) { h_gy[i * 3 + k] = 0.2f * rnd(); h_ac[i * 3 + k] = rnd() + (k == 2 ? 9.8f : 0.f); }
    float q[4] = {rnd(), rnd(), rnd(), 1.f + rnd()};
    const float nn = 1.f / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; ++k) h_ro[i * 4 + k] = q[k] * nn;
  }
  std::vector<float> h_ic((size_t)B * 81, 0.f);
  for (int64_t b = 0; b < B; ++b) for (int k = 0; k < 9; ++k) h_ic[(size_t)b * 81 + k * 10] = 1e-6f;
  const float h_gc[3] = {1e-4f, 1e-4f, 1e-4f}, h_ac3[3] = {1e-3f, 1e-3f, 1e-3f};
  float *dt, *gy, *ac, *ro, *ic, *gc, *acov, *cov, *cov2;
  CK(hipMalloc(&dt, n * 4)); CK(hipMalloc(&gy, n * 12)); CK(hipMalloc(&ac, n * 12)); CK(hipMalloc(&ro, n * 16));
  CK(hipMalloc(&ic, (size_t)B * 81 * 4)); CK(hipMalloc(&gc, 12)); CK(hipMalloc(&acov, 12)); CK(hipMalloc(&cov, (size_t)B * 81 * 4));
  CK(hipMalloc(&cov2, (size_t)B * 81 * 4));
  CK(hipMemcpy(dt, h_dt.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(gy, h_gy.data(), n * 12, hipMemcpyHostToDevice));
  CK(hipMemcpy(ac, h_ac.data(), n * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(ro, h_ro.data(), n * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(ic, h_ic.data(), (size_t)B * 81 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(gc, h_gc, 12, hipMemcpyHostToDevice)); CK(hipMemcpy(acov, h_ac3, 12, hipMemcpyHostToDevice));
  const float us_real = run(-1, dt, gy, ac, ro, ic, gc, acov, cov, B, F, reps);
  const float us_synth = run(63, dt, gy, ac, ro, ic, gc, acov, cov2, B, F, reps);
  CK(hipDeviceSynchronize());
  std::vector<float> out((size_t)B * 81), out2((size_t)B * 81);
  CK(hipMemcpy(out.data(), cov, out.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(out2.data(), cov2, out2.size() * 4, hipMemcpyDeviceToHost));
  double c1 = 0, c2 = 0;
  for (size_t i = 0; i < out.size(); ++i) { c1 += std::fabs(out[i]); c2 += std::fabs(out2[i]); }
  printf("{\"B\": %lld, \"F\": %lld, \"kernel\": \"imu_cov_seg_kernel<float, 2, 4, ISO>\", \"us_real_inputs\": %.1f, \"us_cache_resident_inputs\": %.1f, "
         "\"bytes_read_real\": %.0f, \"checksum_real\": %.6e, \"checksum_masked\": %.6e}\n",
         (long long)B, (long long)F, us_real, us_synth, (double)n * 44.0, c1, c2);
  return 0;
}
