// Issue rate of plain wave64 VALU instructions on one SIMD of gfx950, as a function of the waves resident on it:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/build/valu_rate
// Every wave runs ITER x 16 independent instructions of one kind (inline asm, nothing for the compiler to fold); the grid is
// 256 CUs x 4 SIMDs x k waves.  Prints cycles per wave64 instruction per SIMD at the 2.4 GHz peak clock and the clock-free
// ratio against v_pk_fma_f32 -- the figure bench.py's `bound: "valu"` rooflines are priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND> __global__ void __launch_bounds__(256) rate(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
  const float m = 0.999f, c = 0.001f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
  const f2 pm = {m, m}, pc = {c, c};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {          // v_fma_f32, 16 independent chains
      asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                   "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                   "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                   "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                     "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m), "v"(c));
    } else if (KIND == 1) {   // v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n"
                   "v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n"
                   "v_mul_f32 %8, %8, %16\n v_mul_f32 %9, %9, %16\n v_mul_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n"
                   "v_mul_f32 %12, %12, %16\n v_mul_f32 %13, %13, %16\n v_mul_f32 %14, %14, %16\n v_mul_f32 %15, %15, %16\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                     "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m));
    } else if (KIND == 2) {   // v_pk_fma_f32: two fp32 FMAs per lane per instruction
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
    } else if (KIND == 3) {   // v_mov_b32 with a DPP row shift (what the wave scans are made of)
      asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %10, %11 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %11, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %13, %14 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %14, %15 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %15, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                     "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    } else if (KIND == 4) {   // v_rcp_f32 (transcendental pipe)
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n"
                   "v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n v_rcp_f32 %8, %8\n v_rcp_f32 %9, %9\n v_rcp_f32 %10, %10\n v_rcp_f32 %11, %11\n"
                   "v_rcp_f32 %12, %12\n v_rcp_f32 %13, %13\n v_rcp_f32 %14, %14\n v_rcp_f32 %15, %15\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                     "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    } else {                  // one DEPENDENT chain of v_fma_f32: the latency a lone wave sees
      asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  s += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
  if (s == 12345.678f) out[0] = s;
}
template <int KIND> double run(int waves_per_simd, int iters, float* out) {
  const int cus = 256, blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD; k blocks per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters / 8, 1.0f);
  hipDeviceSynchronize();
  double best = 1e30;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // wave-instructions issued on ONE SIMD = waves_per_simd * iters * 16; cycles at 2.4 GHz
  return best * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 16);
}
int main() {
  float* out;
  hipMalloc(&out, 64);
  const int iters = 1 << 16;
  const char* names[] = {"v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_mov_b32_dpp row_shr:1", "v_rcp_f32", "v_fma_f32 (one dependent chain)"};
  printf("{\"unit\": \"cycles at 2.4 GHz per wave64 instruction on one SIMD\", \"rows\": [\n");
  for (int k = 0; k < 6; ++k) {
    printf(" {\"instr\": \"%s\"", names[k]);
    for (int w : {1, 2, 4, 8}) {
      double c = k == 0 ? run<0>(w, iters, out) : k == 1 ? run<1>(w, iters, out) : k == 2 ? run<2>(w, iters, out)
               : k == 3 ? run<3>(w, iters, out) : k == 4 ? run<4>(w, iters, out) : run<5>(w, iters, out);
      printf(", \"waves_%d\": %.3f", w, c);
    }
    printf("}%s\n", k < 5 ? "," : "");
  }
  printf("]}\n");
  return 0;
}
