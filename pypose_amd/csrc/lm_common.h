// lm_common.h -- what every device-resident Levenberg-Marquardt step shares (csrc/lm_step.hip: the InvNet program;
// csrc/lm_generic.hip: any r = Log / Act of L * P^{+-1} * R): the loop state, the in-kernel decision (strategy update + accept /
// reject, optimizer.py:673-678, strategy.py:143-151, 260-274) and the fixed-order reduction of the partial sums.
#pragma once
#include "rowmap.h"
#include "gridsync.h"

namespace pplie {

constexpr int kStepPartials = 4096;     // = PPLIE_LM_TRIAL_PARTIALS
enum { ST_DAMPING = 0, ST_RADIUS, ST_DOWN, ST_SCALE, ST_LAST, ST_LOSS, ST_REJECTS, ST_DONE, ST_FAILED, ST_TRIALS, ST_QUALITY,
       // StopOnPlateau evaluated on the device (scheduler.py:130-160; cfg.plateau_max_steps > 0): steps taken, consecutive steps that
       // lowered the loss by less than `decreasing`, and the verdict -- once it is 1 every later launch of the step returns at once
       ST_PL_STEPS, ST_PL_COUNT, ST_PL_STOP,
       ST_SIZE = 16 };                  // = PPLIE_LM_STATE
enum { LM_CONSTANT = 0, LM_ADAPTIVE = 1, LM_TRUSTREGION = 2 };
enum { LMF_HOST_STATE = 1, LMF_NO_LOSS = 2, LMF_OCC4 = 4 /* tuning: the trial kernel built for 4 waves / SIMD */ };

struct LmCfg {     // = pplie_lm_cfg in include/pplie.h
  double high, low, up, factor, smin, smax, sdown;    // strategy constants (param group / strategy object)
  double dmin, dmax;                                  // clamp of the diagonal of J^T J (LM(min=, max=))
  double host_damping, host_down;                     // param-group values, used when flags & LMF_HOST_STATE
  int strategy, reject, flags, grid_cap;               // grid_cap: workgroups of the trial kernel (0: 4096)
  int plateau_patience, plateau_max_steps;             // StopOnPlateau(steps=, patience=) on the device; max_steps = 0: off
  double plateau_decreasing;                           // StopOnPlateau(decreasing=)
  unsigned long long plateau_flag;                     // address of a HOST-PINNED double (or 0): the step that stops the run stores its
                                                       // step count there with system scope -- the host stops enqueuing without a sync
};

// ---- decision (optimizer.py:673-678 + strategy.py:143-151, 260-274) ---------------------------------------------------
// Reads the loop state from `in`, returns the new one in `o` (the caller decides who stores it: several workgroups may
// evaluate the same decision redundantly, from the same numbers in the same order, and only one of them writes).
__device__ __forceinline__ void lm_decide(const double* in, double* o, const LmCfg& cfg, bool first, double v_new, double v_old,
                                          double v_jj, double v_jr) {
  double damping, radius, down, scale, last, rejects;
  if (first) {
    const bool host = cfg.flags & LMF_HOST_STATE;
    damping = host ? cfg.host_damping : in[ST_DAMPING];
    down = host ? cfg.host_down : in[ST_DOWN];
    scale = 1.0 + damping;
    last = (cfg.flags & LMF_NO_LOSS) ? v_old : in[ST_LOSS];
    rejects = 0.0;
  } else {
    damping = in[ST_DAMPING];
    down = in[ST_DOWN];
    scale = in[ST_SCALE] * (1.0 + damping);
    last = in[ST_LAST];
    rejects = in[ST_REJECTS];
  }
  radius = 1.0 / damping;
  double loss = v_new;
  const bool failed = !(v_new == v_new);           // NaN: a non-positive pivot in some problem's Cholesky (solver.py:214)
  const double quality = (last - loss) / -(v_jj + 2.0 * v_jr);
  // a failed factorisation: the reference's solver raises and the step is abandoned BEFORE strategy.update
  // (optimizer.py:667-670) -- damping, radius and down stay what they were
  if (!failed) {
    if (cfg.strategy == LM_ADAPTIVE) {
      if (quality > cfg.high) damping *= down;
      else if (!(quality > cfg.low)) damping *= cfg.up;
      damping = fmax(cfg.smin, fmin(damping, cfg.smax));
    } else if (cfg.strategy == LM_TRUSTREGION) {
      if (quality > cfg.high) { radius *= cfg.up; down = cfg.sdown; }
      else if (quality > cfg.low) { down = cfg.sdown; }
      else { radius *= down; down *= cfg.factor; }
      down = fmax(cfg.smin, fmin(down, cfg.smax));
      radius = fmax(cfg.smin, fmin(radius, cfg.smax));
      damping = 1.0 / radius;
    }
  }
  double done = 1.0;
  if (!failed && last < loss && rejects < (double)cfg.reject) {     // reject the step
    loss = last;
    rejects += 1.0;
    done = 0.0;
  }
  if (failed) loss = last;
  o[ST_DAMPING] = damping; o[ST_RADIUS] = radius; o[ST_DOWN] = down; o[ST_SCALE] = scale;
  o[ST_LAST] = last; o[ST_LOSS] = loss; o[ST_REJECTS] = rejects; o[ST_DONE] = done;
  o[ST_FAILED] = failed ? 1.0 : 0.0;
  o[ST_TRIALS] = first ? 1.0 : in[ST_TRIALS] + 1.0;
  o[ST_QUALITY] = quality;
  // the scheduler's rules on the step that has just ENDED (accepted, out of retries, or abandoned): scheduler.py:136-160 reads
  // optimizer.last / .loss / .reject_count right after optimizer.step()
  double pl_steps = in[ST_PL_STEPS], pl_count = in[ST_PL_COUNT], pl_stop = in[ST_PL_STOP];
  if (cfg.plateau_max_steps > 0 && (done != 0.0 || failed)) {
    pl_steps += 1.0;
    pl_count = (last - loss) < cfg.plateau_decreasing ? pl_count + 1.0 : 0.0;
    if (pl_steps >= (double)cfg.plateau_max_steps || pl_count >= (double)cfg.plateau_patience || rejects > 0.0) pl_stop = 1.0;
  }
  o[ST_PL_STEPS] = pl_steps; o[ST_PL_COUNT] = pl_count; o[ST_PL_STOP] = pl_stop;
}
template <class T> __device__ __forceinline__ void lm_store_state(const double* o, double* st, T* loss_out, T* last_out,
                                                                  unsigned long long plateau_flag = 0ull) {
#pragma unroll
  for (int i = 0; i <= ST_PL_STOP; ++i) __hip_atomic_store(st + i, o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (plateau_flag && o[ST_PL_STOP] != 0.0)
    __hip_atomic_store(reinterpret_cast<double*>(plateau_flag), o[ST_PL_STEPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (loss_out) *loss_out = (T)o[ST_LOSS];
  if (last_out) *last_out = (T)o[ST_LAST];
}
__device__ __forceinline__ double ld_state(const double* st, int i) {
  return __hip_atomic_load(st + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the workgroup (valid in thread 0); every thread must call it
template <class T, int BLOCK> __device__ __forceinline__ T wg_sum(T v) {
  __shared__ T part[BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  T s = T(0);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += part[w];
  }
  __syncthreads();
  return s;
}

// fixed-order sum of the first `rows` partial rows by one workgroup (bit-reproducible from run to run, and the same
// bits in every workgroup that evaluates it); valid in thread 0.  FRESH: the rows were written by an earlier launch
// (plain 16-byte loads); otherwise by other workgroups of THIS launch, whose stores this CU's L1 never sees:
// agent-scope loads.
template <class T, int BLOCK, bool FRESH>
__device__ __forceinline__ void lm_reduce_partials(const T* partials, int rows, double out[4]) {
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < rows; i += BLOCK) {
    if (FRESH) {
      T r[4];
      row_ld<4>(partials + (size_t)i * 4, r);
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += (double)r[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        a[k] += (double)__hip_atomic_load(partials + (size_t)i * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the four sums with ONE pair of barriers (this sits on the critical path of every LM step)
  __shared__ double part4[4][BLOCK / 64];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = a[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) part4[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < BLOCK / 64; ++w) sum += part4[k][w];
      out[k] = sum;
    }
  }
  __syncthreads();
}

// Workgroups of the finish kernel.  Its grid barrier needs all of them resident: 64 single-workgroup CUs are there on any
// part this library targets (MI355X: 256 CUs); few enough for the redundant reductions of the common path to stay cheap,
// and the retries they run are rare.
constexpr int kFinishGrid = 64;

}  // namespace pplie
