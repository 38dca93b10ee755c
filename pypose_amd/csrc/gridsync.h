// gridsync.h -- a grid-wide barrier for persistent kernels whose workgroups are all resident.
#pragma once
#include <hip/hip_runtime.h>

namespace pplie {

// Sense-reversing barrier on two words (count, generation), zeroed once by the caller and left at rest (count = 0) by
// every complete use.  Every workgroup of the launch must be resident (grid <= what the device holds at once) or the
// barrier cannot complete; the spin is bounded (~1 s) so that a stranded workgroup ends the kernel instead of wedging
// the GPU.  Memory: the __syncthreads() before the arrival drains each wave's stores (vmcnt(0)), the agent-scope fence
// writes this XCD's L2 back, the fence after the wait invalidates this CU's L1 / the XCD's non-local L2 lines -- so data
// written before the barrier by ANY workgroup is visible to plain loads after it (MI355X_MICROARCH.md, inter-workgroup
// visibility).
__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    if (atomicAdd(count, 1u) == gridDim.x - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (long spin = 0; spin < (1L << 24) && __hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g; ++spin)
        __builtin_amdgcn_s_sleep(2);
    }
    __threadfence();
  }
  __syncthreads();
}

// The same rendezvous WITHOUT the cache maintenance: no fence, so neither this XCD's L2 is written back nor its
// non-local lines invalidated (an agent-scope acquire drops every cached line of the kernel's read-only operands --
// measured: each PCG iteration then re-fetched its 11 MB of blocks from memory).  Data that crosses workgroups must go
// through agent-scope (sc1) stores and loads instead (xwg_store / xwg_load below); everything else keeps its cache lines.
// Arrival is two-level (groups of 8 workgroups on their own 128-byte-apart counters, then one counter for the groups):
// same-address atomics serialise at ~80 ns each, 64 flat arrivals measured 5.3 us.  bar: PPLIE_GRID_BAR_WORDS uint32,
// zeroed once.
constexpr int kBarWords = 64 + 32 * 32;       // = PPLIE_GRID_BAR_WORDS: [0] group arrivals, [32] generation, [64 + 32 g] group g
__device__ __forceinline__ void grid_rendezvous(unsigned* bar) {
  __syncthreads();                            // (drains this workgroup's outstanding stores: s_waitcnt vmcnt(0))
  if (threadIdx.x == 0) {
    unsigned* gen = bar + 32;
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned ngroups = (gridDim.x + 7) >> 3, grp = blockIdx.x >> 3;
    const unsigned members = grp + 1 < ngroups ? 8u : gridDim.x - 8u * (ngroups - 1);
    unsigned* gc = bar + 64 + 32 * (grp & 31);
    bool release = false;
    if (__hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
      __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        release = true;
      }
    }
    if (release) {
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (long spin = 0; spin < (1L << 24) && __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g; ++spin)
        __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
template <class T> __device__ __forceinline__ void xwg_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ T xwg_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


// ---- reductions across workgroups without a separate barrier ---------------------------------------------------------
// Every workgroup publishes its partial sums as 64-bit words { sequence number | 32 value bits } (one agent-scope store
// each: tag and payload can never be seen apart) and then collects everybody's: a row whose tags all equal the expected
// sequence number is complete, so the poll IS the barrier -- one memory round trip instead of three (arrival counter,
// release flag, table read).  Rows are added in a fixed order, identically in every workgroup.  Phase A and phase B use
// separate tables: a workgroup can only overwrite its phase-A row after everyone published phase B, i.e. after everyone
// finished reading phase A.
typedef unsigned long long u64;
template <class T> __device__ __forceinline__ void put_tagged(u64* row, int q, T v, unsigned seq) {
  constexpr int NW = sizeof(T) / 4;
  unsigned w[NW];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (int k = 0; k < NW; ++k) xwg_store(row + q * NW + k, ((u64)seq << 32) | (u64)w[k]);
}
// sums of Q quantities over `rows` rows (first wave polls; result broadcast to the workgroup); false on a timeout
template <class T, int Q> __device__ __forceinline__ bool gather_tagged(const u64* tab, int rows, unsigned seq, T out[Q], T* sh) {
  constexpr int NW = sizeof(T) / 4, RW = 4 * NW;            // words per table row (4 quantities reserved)
  __shared__ int ok_sh;
  if (threadIdx.x < 64) {
    T a[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) a[q] = T(0);
    bool ok = true;
    for (int i = threadIdx.x; i < rows; i += 64) {
      u64 w[Q * NW];
      bool got = false;
      for (long spin = 0; spin < (1L << 22) && !got; ++spin) {
        got = true;
#pragma unroll
        for (int k = 0; k < Q * NW; ++k) {
          w[k] = xwg_load(tab + (size_t)i * RW + k);
          got = got && (unsigned)(w[k] >> 32) == seq;
        }
      }
      ok = ok && got;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        unsigned bits[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) bits[k] = (unsigned)w[q * NW + k];
        T v;
        __builtin_memcpy(&v, bits, sizeof(T));
        a[q] += v;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    ok = __all(ok);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Q; ++q) sh[q] = a[q];
      ok_sh = ok ? 1 : 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q) out[q] = sh[q];
  const bool ok = ok_sh != 0;
  __syncthreads();
  return ok;
}
}  // namespace pplie
