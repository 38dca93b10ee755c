// rowmap.h -- the one kernel shell behind every batched Lie op.
//
// All ops on the hot path are row maps: up to three AoS inputs [N, W_in] -> up to two AoS
// outputs [N, W_out] with W in {3..9} scalars.  Row pitches are 12..36 B (fp32), so a lane
// that owns a row cannot issue aligned 16 B global accesses.  The shell therefore moves
// whole TILE x W slabs between HBM and LDS with dwordx4 accesses (the slab of a tile is
// contiguous in memory), and lanes pick their rows out of LDS (odd pitches are bank-conflict
// free for ds_read_b32; even pitches use b64/b128 accesses).  HBM sees exactly the
// algorithmic bytes, once, fully coalesced; the arithmetic (lie_math.h) stays in registers.
//
//   HBM --dwordx4--> LDS(in slabs) --row/lane--> VGPR --Op::apply--> LDS(out slabs) --dwordx4--> HBM
//
// Grid: one workgroup per tile (a launch fills all 256 CUs / 8 XCDs many times over at the
// sizes this library targets; there is no inter-tile reuse, so no XCD-specific tile mapping is
// needed -- each tile is touched by exactly one workgroup, consecutive tiles land on
// consecutive XCDs and stream disjoint HBM pages).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lie_math.h"

namespace pplie {

typedef int __attribute__((ext_vector_type(4))) raw16;
typedef int __attribute__((ext_vector_type(2))) raw8;

// error codes returned through the C ABI
enum { PPLIE_OK = 0, PPLIE_EBADARG = -1, PPLIE_ELAUNCH = -2, PPLIE_ECAPACITY = -3 };

template <int W, class T> __device__ __forceinline__ void row_ld(const T* __restrict__ p, T* r) {
  if constexpr ((W * sizeof(T)) % 16 == 0) {
    constexpr int NV = W * sizeof(T) / 16;
    raw16 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = reinterpret_cast<const raw16*>(p)[i];
    __builtin_memcpy(r, v, sizeof(v));
  } else if constexpr ((W * sizeof(T)) % 8 == 0) {
    constexpr int NV = W * sizeof(T) / 8;
    raw8 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = reinterpret_cast<const raw8*>(p)[i];
    __builtin_memcpy(r, v, sizeof(v));
  } else {
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = p[i];
  }
}
template <int W, class T> __device__ __forceinline__ void row_st(T* __restrict__ p, const T* r) {
  if constexpr ((W * sizeof(T)) % 16 == 0) {
    constexpr int NV = W * sizeof(T) / 16;
    raw16 v[NV];
    __builtin_memcpy(v, r, sizeof(v));
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<raw16*>(p)[i] = v[i];
  } else if constexpr ((W * sizeof(T)) % 8 == 0) {
    constexpr int NV = W * sizeof(T) / 8;
    raw8 v[NV];
    __builtin_memcpy(v, r, sizeof(v));
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<raw8*>(p)[i] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < W; ++i) p[i] = r[i];
  }
}

// slab copy HBM -> LDS.  NELEM = elements of a full tile slab (a multiple of 16 B).
template <class T, int BLOCK, int NELEM, bool VEC>
__device__ __forceinline__ void slab_g2s(const T* __restrict__ g, T* __restrict__ s, int valid, bool full) {
  const int tid = threadIdx.x;
  if (VEC && full) {
    constexpr int NV = NELEM * (int)sizeof(T) / 16;
    constexpr int PER = (NV + BLOCK - 1) / BLOCK;
    raw16 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int idx = tid + k * BLOCK;
      if (NV % BLOCK == 0 || idx < NV) v[k] = __builtin_nontemporal_load(reinterpret_cast<const raw16*>(g) + idx);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int idx = tid + k * BLOCK;
      if (NV % BLOCK == 0 || idx < NV) reinterpret_cast<raw16*>(s)[idx] = v[k];
    }
  } else {
    for (int i = tid; i < valid; i += BLOCK) s[i] = g[i];
  }
}
template <class T, int BLOCK, int NELEM, bool VEC>
__device__ __forceinline__ void slab_s2g(const T* __restrict__ s, T* __restrict__ g, int valid, bool full) {
  const int tid = threadIdx.x;
  if (VEC && full) {
    constexpr int NV = NELEM * (int)sizeof(T) / 16;
    constexpr int PER = (NV + BLOCK - 1) / BLOCK;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int idx = tid + k * BLOCK;
      if (NV % BLOCK == 0 || idx < NV)
        __builtin_nontemporal_store(reinterpret_cast<const raw16*>(s)[idx], reinterpret_cast<raw16*>(g) + idx);
    }
  } else {
    for (int i = tid; i < valid; i += BLOCK) g[i] = s[i];
  }
}

// sum over a 256-thread workgroup (valid in thread 0); every thread must call it
template <class T> __device__ __forceinline__ T block_sum(T v) {
  __shared__ T part[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = v;
  __syncthreads();
  T s = T(0);
  if (threadIdx.x == 0) s = part[0] + part[1] + part[2] + part[3];
  __syncthreads();
  return s;   // valid in thread 0
}

template <int W> struct AtLeast1 { enum { v = W > 0 ? W : 1 }; };

// Op concept: enum {IW0,IW1,IW2,OW0,OW1} (0 = unused) and
//   static __device__ void apply(const T* a, const T* b, const T* c, T* o, T* p)
// or, for ops with a launch-wide scalar parameter (Prm != NoParam),
//   static __device__ void apply(const T* a, const T* b, const T* c, T* o, T* p, Prm prm)
struct NoParam {};
template <class A, class B> struct SameType { static constexpr bool v = false; };
template <class A> struct SameType<A, A> { static constexpr bool v = true; };

// GB ("cotangent broadcast"): the LAST input is ONE row shared by all n rows -- what autograd hands the first backward node
// of `op(x).sum().backward()` (a stride-0 expansion of one scalar).  That row is read once per workgroup into registers; no slab,
// no HBM traffic for it: materialising it instead costs W x 4 bytes written + read per row (15 % of the Exp-Log chain's traffic).
template <class T, class Op, int RPT, int BLOCK, bool VEC, bool ROLL = false, class Prm = NoParam, bool GB = false>
__global__ void __launch_bounds__(BLOCK)
rowmap_lds_kernel(const T* i0, const T* i1, const T* i2, T* o0, T* o1, int64_t n, Prm prm = Prm()) {
  // (no __restrict__: an output may be the buffer of an input -- the optimizer's in-place retraction -- and every tile is
  //  in LDS before any of its rows is written back)
  constexpr int TILE = RPT * BLOCK;
  constexpr int OW0 = Op::OW0, OW1 = Op::OW1;
  constexpr int LASTIN = Op::IW2 > 0 ? 2 : (Op::IW1 > 0 ? 1 : 0);
  // slab widths: the broadcast input has none
  constexpr int IW0 = (GB && LASTIN == 0) ? 0 : Op::IW0, IW1 = (GB && LASTIN == 1) ? 0 : Op::IW1, IW2 = (GB && LASTIN == 2) ? 0 : Op::IW2;
  constexpr int GW = LASTIN == 2 ? Op::IW2 : (LASTIN == 1 ? Op::IW1 : Op::IW0);
  constexpr int OFF_I1 = TILE * IW0, OFF_I2 = OFF_I1 + TILE * IW1, OFF_O0 = OFF_I2 + TILE * IW2,
                OFF_O1 = OFF_O0 + TILE * OW0, TOTAL = OFF_O1 + TILE * OW1;
  T gb[AtLeast1<GB ? GW : 0>::v];
  if constexpr (GB) {
    const T* gp = LASTIN == 2 ? i2 : (LASTIN == 1 ? i1 : i0);
#pragma unroll
    for (int k = 0; k < GW; ++k) gb[k] = gp[k];
  }
  __shared__ __attribute__((aligned(16))) T lds[TOTAL];
  T* s_i0 = lds;
  T* s_i1 = lds + OFF_I1;
  T* s_i2 = lds + OFF_I2;
  T* s_o0 = lds + OFF_O0;
  T* s_o1 = lds + OFF_O1;

  const int64_t ntiles = (n + TILE - 1) / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * TILE;
    const int64_t left = n - row0;
    const bool full = left >= TILE;
    const int rows = full ? TILE : (int)left;

    if constexpr (IW0 > 0) slab_g2s<T, BLOCK, TILE * IW0, VEC>(i0 + row0 * IW0, s_i0, rows * IW0, full);
    if constexpr (IW1 > 0) slab_g2s<T, BLOCK, TILE * IW1, VEC>(i1 + row0 * IW1, s_i1, rows * IW1, full);
    if constexpr (IW2 > 0) slab_g2s<T, BLOCK, TILE * IW2, VEC>(i2 + row0 * IW2, s_i2, rows * IW2, full);
    __syncthreads();

    auto do_row = [&](int r) {
      const int row = threadIdx.x + r * BLOCK;
      if (row < rows) {
        T a[AtLeast1<Op::IW0>::v], b[AtLeast1<Op::IW1>::v], c[AtLeast1<Op::IW2>::v], p[AtLeast1<OW0>::v], q[AtLeast1<OW1>::v];
        if constexpr (IW0 > 0) row_ld<IW0>(s_i0 + row * IW0, a);
        if constexpr (IW1 > 0) row_ld<IW1>(s_i1 + row * IW1, b);
        if constexpr (IW2 > 0) row_ld<IW2>(s_i2 + row * IW2, c);
        if constexpr (GB) {
          T* dst = LASTIN == 2 ? c : (LASTIN == 1 ? b : a);
#pragma unroll
          for (int k = 0; k < GW; ++k) dst[k] = gb[k];
        }
        if constexpr (SameType<Prm, NoParam>::v) Op::apply(a, b, c, p, q);
        else Op::apply(a, b, c, p, q, prm);
        row_st<OW0>(s_o0 + row * OW0, p);
        if constexpr (OW1 > 0) row_st<OW1>(s_o1 + row * OW1, q);
      }
    };
    if constexpr (ROLL) {      // rows one after the other: register footprint of a single row
#pragma unroll 1
      for (int r = 0; r < RPT; ++r) do_row(r);
    } else {
#pragma unroll
      for (int r = 0; r < RPT; ++r) do_row(r);
    }
    __syncthreads();

    slab_s2g<T, BLOCK, TILE * OW0, VEC>(s_o0, o0 + row0 * OW0, rows * OW0, full);
    if constexpr (OW1 > 0) slab_s2g<T, BLOCK, TILE * OW1, VEC>(s_o1, o1 + row0 * OW1, rows * OW1, full);
    // no third barrier: the next tile's g2s only writes the input slabs, which every lane
    // finished reading before the barrier above; the output slabs are rewritten only after
    // the next tile's first barrier, which every lane reaches after its s2g reads.
  }
}

// comparison variant without LDS staging (each lane reads/writes its own row from global)
template <class T, class Op, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
rowmap_direct_kernel(const T* __restrict__ i0, const T* __restrict__ i1, const T* __restrict__ i2,
                     T* __restrict__ o0, T* __restrict__ o1, int64_t n) {
  constexpr int IW0 = Op::IW0, IW1 = Op::IW1, IW2 = Op::IW2, OW0 = Op::OW0, OW1 = Op::OW1;
  for (int64_t row = (int64_t)blockIdx.x * BLOCK + threadIdx.x; row < n; row += (int64_t)gridDim.x * BLOCK) {
    T a[AtLeast1<IW0>::v], b[AtLeast1<IW1>::v], c[AtLeast1<IW2>::v], p[AtLeast1<OW0>::v], q[AtLeast1<OW1>::v];
#pragma unroll
    for (int k = 0; k < IW0; ++k) a[k] = i0[row * IW0 + k];
    if constexpr (IW1 > 0) {
#pragma unroll
      for (int k = 0; k < IW1; ++k) b[k] = i1[row * IW1 + k];
    }
    if constexpr (IW2 > 0) {
#pragma unroll
      for (int k = 0; k < IW2; ++k) c[k] = i2[row * IW2 + k];
    }
    Op::apply(a, b, c, p, q);
#pragma unroll
    for (int k = 0; k < OW0; ++k) o0[row * OW0 + k] = p[k];
    if constexpr (OW1 > 0) {
#pragma unroll
      for (int k = 0; k < OW1; ++k) o1[row * OW1 + k] = q[k];
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// One tile per workgroup (no grid-stride cap) measured fastest on MI355X at 10M rows
// (profiles/r01/tune_rowmap_first.json: 6.3 TB/s uncapped vs 5.3-5.7 TB/s capped at 2048/4096);
// the grid-stride loop only engages beyond 2^30 tiles.
constexpr int kGridCap = 1 << 30;

template <class T, class Op, int RPT = 2, int BLOCK = 256, bool ROLL = false, class Prm = NoParam, bool GB = false>
int launch_rowmap(const void* i0, const void* i1, const void* i2, void* o0, void* o1, int64_t n, void* stream,
                  int grid_cap = kGridCap, Prm prm = Prm()) {
  if (n < 0) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  if (!i0 || !o0 || (Op::IW1 > 0 && !i1) || (Op::IW2 > 0 && !i2) || (Op::OW1 > 0 && !o1)) return PPLIE_EBADARG;
  constexpr int TILE = RPT * BLOCK;
  int64_t ntiles = (n + TILE - 1) / TILE;
  int grid = (int)(ntiles < grid_cap ? ntiles : grid_cap);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int LASTIN = Op::IW2 > 0 ? 2 : (Op::IW1 > 0 ? 1 : 0);           // (GB: the broadcast row is read element by element)
  bool vec = ((GB && LASTIN == 0) || aligned16(i0)) && aligned16(o0) && (Op::IW1 == 0 || (GB && LASTIN == 1) || aligned16(i1)) &&
             (Op::IW2 == 0 || (GB && LASTIN == 2) || aligned16(i2)) && (Op::OW1 == 0 || aligned16(o1));
  const T* a = static_cast<const T*>(i0);
  const T* b = static_cast<const T*>(i1);
  const T* c = static_cast<const T*>(i2);
  T* p = static_cast<T*>(o0);
  T* q = static_cast<T*>(o1);
  if (vec)
    hipLaunchKernelGGL((rowmap_lds_kernel<T, Op, RPT, BLOCK, true, ROLL, Prm, GB>), dim3(grid), dim3(BLOCK), 0, st, a, b, c, p, q, n, prm);
  else
    hipLaunchKernelGGL((rowmap_lds_kernel<T, Op, RPT, BLOCK, false, ROLL, Prm, GB>), dim3(grid), dim3(BLOCK), 0, st, a, b, c, p, q, n, prm);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

template <class T, class Op, int BLOCK = 256>
int launch_rowmap_direct(const void* i0, const void* i1, const void* i2, void* o0, void* o1, int64_t n, void* stream,
                         int grid_cap = kGridCap) {
  if (n < 0) return PPLIE_EBADARG;
  if (n == 0) return PPLIE_OK;
  int64_t nb = (n + BLOCK - 1) / BLOCK;
  int grid = (int)(nb < grid_cap ? nb : grid_cap);
  hipLaunchKernelGGL((rowmap_direct_kernel<T, Op, BLOCK>), dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     static_cast<const T*>(i0), static_cast<const T*>(i1), static_cast<const T*>(i2), static_cast<T*>(o0),
                     static_cast<T*>(o1), n);
  return hipGetLastError() == hipSuccess ? PPLIE_OK : PPLIE_ELAUNCH;
}

// ---- op functor generators -------------------------------------------------------------
#define PPLIE_OP_1_1(NAME, FN, I0, O0)                                                      \
  template <class T> struct NAME {                                                          \
    enum { IW0 = I0, IW1 = 0, IW2 = 0, OW0 = O0, OW1 = 0 };                                 \
    static PP_HD void apply(const T* a, const T*, const T*, T* o, T*) { FN<T>(a, o); }      \
  };
#define PPLIE_OP_2_1(NAME, FN, I0, I1, O0)                                                  \
  template <class T> struct NAME {                                                          \
    enum { IW0 = I0, IW1 = I1, IW2 = 0, OW0 = O0, OW1 = 0 };                                \
    static PP_HD void apply(const T* a, const T* b, const T*, T* o, T*) { FN<T>(a, b, o); } \
  };
#define PPLIE_OP_2_2(NAME, FN, I0, I1, O0, O1)                                                  \
  template <class T> struct NAME {                                                              \
    enum { IW0 = I0, IW1 = I1, IW2 = 0, OW0 = O0, OW1 = O1 };                                   \
    static PP_HD void apply(const T* a, const T* b, const T*, T* o, T* p) { FN<T>(a, b, o, p); } \
  };
#define PPLIE_OP_3_2(NAME, FN, I0, I1, I2, O0, O1)                                                    \
  template <class T> struct NAME {                                                                    \
    enum { IW0 = I0, IW1 = I1, IW2 = I2, OW0 = O0, OW1 = O1 };                                        \
    static PP_HD void apply(const T* a, const T* b, const T* c, T* o, T* p) { FN<T>(a, b, c, o, p); } \
  };

// Tile shape of an op: rows per lane per tile (tile = RPT x 256 rows).  Default: 2 rows per lane
// while the LDS slabs stay <= 28 KB per workgroup, else 1 (multi-slab backward kernels would drop
// to 2 workgroups per CU); fp64 always 1.  Ops measured faster at another shape on MI355X
// (profiles/r01/tune_*.json) specialise TileOf.
template <class Op> struct TileOf {
  static constexpr int sumw = Op::IW0 + Op::IW1 + Op::IW2 + Op::OW0 + Op::OW1;
  static constexpr int rpt32 = sumw <= 14 ? 2 : 1;
  static constexpr int rpt64 = 1;
  static constexpr int block32 = 256;          // workgroup size of the fp32 build
  static constexpr bool roll32 = false;        // rows of a lane one after the other (register footprint of a single row)
};
#define PPLIE_TILE(OP, RPT32) \
  template <> struct TileOf<OP<float>> { static constexpr int rpt32 = RPT32; static constexpr int rpt64 = 1; \
                                         static constexpr int block32 = 256; static constexpr bool roll32 = false; };
// full shape: rows per lane, workgroup size, rolled rows (tools/tune_general.py measures the candidates at 10 M rows)
#define PPLIE_TILE_EX(OP, RPT32, BLOCK32, ROLL32) \
  template <> struct TileOf<OP<float>> { static constexpr int rpt32 = RPT32; static constexpr int rpt64 = 1; \
                                         static constexpr int block32 = BLOCK32; static constexpr bool roll32 = ROLL32; };

// fp64 shape of an op (its own trait: the fp32 specialisations above stay as they are).  Default 256 x 1 unrolled; ops measured
// faster at another shape at 10 M rows (tools/tune_general.py --f64, profiles/r05) specialise it with PPLIE_TILE64.
template <class Op> struct TileOf64 {
  static constexpr int rpt = 1;
  static constexpr int block = 256;
  static constexpr bool roll = false;
};
#define PPLIE_TILE64(OP, RPT64, BLOCK64, ROLL64) \
  template <> struct TileOf64<OP<double>> { static constexpr int rpt = RPT64; static constexpr int block = BLOCK64; \
                                            static constexpr bool roll = ROLL64; };

// C-ABI export of one op in both precisions (uniform signature, see include/pplie.h).
#define PPLIE_EXPORT(SYM, OP)                                                                                    \
  extern "C" int SYM##_f32(const void* i0, const void* i1, const void* i2, void* o0, void* o1, int64_t n,        \
                           void* stream) {                                                                       \
    return pplie::launch_rowmap<float, OP<float>, pplie::TileOf<OP<float>>::rpt32, pplie::TileOf<OP<float>>::block32,  \
                                pplie::TileOf<OP<float>>::roll32>(i0, i1, i2, o0, o1, n, stream);                    \
  }                                                                                                              \
  extern "C" int SYM##_f64(const void* i0, const void* i1, const void* i2, void* o0, void* o1, int64_t n,        \
                           void* stream) {                                                                       \
    return pplie::launch_rowmap<double, OP<double>, pplie::TileOf64<OP<double>>::rpt, pplie::TileOf64<OP<double>>::block, \
                                pplie::TileOf64<OP<double>>::roll>(i0, i1, i2, o0, o1, n, stream);                   \
  }

// the same op with its LAST input broadcast from one row (rowmap_lds_kernel GB): SYM_gb_f32 / SYM_gb_f64
#define PPLIE_EXPORT_GB(SYM, OP)                                                                                 \
  extern "C" int SYM##_gb_f32(const void* i0, const void* i1, const void* i2, void* o0, void* o1, int64_t n,     \
                              void* stream) {                                                                    \
    return pplie::launch_rowmap<float, OP<float>, pplie::TileOf<OP<float>>::rpt32, pplie::TileOf<OP<float>>::block32,  \
                                pplie::TileOf<OP<float>>::roll32, pplie::NoParam, true>(i0, i1, i2, o0, o1, n, stream); \
  }                                                                                                              \
  extern "C" int SYM##_gb_f64(const void* i0, const void* i1, const void* i2, void* o0, void* o1, int64_t n,     \
                              void* stream) {                                                                    \
    return pplie::launch_rowmap<double, OP<double>, pplie::TileOf64<OP<double>>::rpt, pplie::TileOf64<OP<double>>::block, \
                                pplie::TileOf64<OP<double>>::roll, pplie::NoParam, true>(i0, i1, i2, o0, o1, n, stream); \
  }

}  // namespace pplie
