// lm_kernels.hip -- the Levenberg-Marquardt inner loop as gfx950 kernels (fp64).
//
// Replaces what ceres::Solve does per iteration for the reference's problem
// (src/solve/optimizer.cpp:54-61): residual/Jacobian evaluation, normal-equation
// assembly, the linear solve's elimination steps, candidate evaluation and the
// trust-region bookkeeping.  Nothing of J is ever written to HBM: every pass
// recomputes the 2x6 / 2x3 blocks from the (L2-resident) parameters, so an
// iteration streams the 24-byte observation records a handful of times instead
// of writing and re-reading 144 bytes of Jacobian per observation.
//
// All kernels are batched over windows: a workgroup belongs to exactly one
// window and returns at once when that window has terminated.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "ba_device.hpp"
#include "chol_tiles.hpp"
#include "hipba_internal.hpp"

namespace hipba {

// ------------------------------------------------------------------------------------------
// reductions (fixed order => bitwise reproducible sums)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// sums N values over the workgroup; the result is valid in every thread of wave 0
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* lds /* N * (blockDim/64) */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) lds[wave * N + i] = v[i];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += lds[w * N + i];
      v[i] = s;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ double block_max(double v, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = wave_max(v);
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double m = lds[0];
  for (int w = 1; w < nw; ++w) m = fmax(m, lds[w]);
  __syncthreads();
  return m;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// One observation, fully corrected the way Ceres hands it to the linear solver: loss
// corrector (sqrt(rho1) on residual and Jacobian), then Jacobi column scaling.
// Returns the cost contribution 1/2 rho(|r|^2).
template <bool WITH_JP>
__device__ __forceinline__ double eval_obs(const CameraK& K, const LmParams& lm, const double* pose, const double* p,
                                           double u, double v, const double* sc /*6 or null*/,
                                           const double* sp /*3 or null*/, double* r, double* Jc, double* Jp,
                                           double* loss_factor = nullptr) {
  double jp_local[6];
  double* jp = WITH_JP ? Jp : jp_local;
  residual_jacobian(K, pose, p, u, v, r, Jc, jp);
  const double s = r[0] * r[0] + r[1] * r[1];
  double rho0 = s;
  if (lm.loss == HIPBA_LOSS_HUBER) {
    double f;
    huber_correct(lm.loss_scale, s, &rho0, &f);
    if (loss_factor) *loss_factor = f;
    r[0] *= f; r[1] *= f;
#pragma unroll
    for (int i = 0; i < 12; ++i) Jc[i] *= f;
    if (WITH_JP) {
#pragma unroll
      for (int i = 0; i < 6; ++i) jp[i] *= f;
    }
  }
  if (sc) {
#pragma unroll
    for (int i = 0; i < 6; ++i) { Jc[i] *= sc[i]; Jc[6 + i] *= sc[i]; }
  }
  if (WITH_JP && sp) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { jp[i] *= sp[i]; jp[3 + i] *= sp[i]; }
  }
  return 0.5 * rho0;
}

// ------------------------------------------------------------------------------------------
// cost pass: 1/2 sum rho(|r|^2) per block (k_residual)
//   which = 0: at x, 1: at the candidate.  blk_off selects the block range (active or CC).
// ------------------------------------------------------------------------------------------
// A workgroup takes kCostGroup consecutive blocks, lane t the t-th observation of each: the index loads of all of
// them are issued before the first gather and the gathers before the first multiply, so a wave has four
// observations' memory round trips in flight instead of one (the pass is bound by those round trips, not by
// arithmetic or bandwidth).  Partial sums stay one entry per block, summed in the same fixed order.
constexpr int kCostGroup = 4;
__global__ __launch_bounds__(kBlock) void k_cost(DevP P, LmParams lm, int blk_off, int blk_end, int which, int ignore_active) {
  __shared__ double red[2 * kCostGroup * 4];
  const int b0 = blk_off + blockIdx.x * kCostGroup;
  const int t = threadIdx.x;
  BlockDesc d[kCostGroup];
  bool on[kCostGroup];
  int sel[kCostGroup];
  bool any = false;
#pragma unroll
  for (int k = 0; k < kCostGroup; ++k) {
    on[k] = b0 + k < blk_end;
    d[k] = P.blocks[on[k] ? b0 + k : b0];
  }
#pragma unroll
  for (int k = 0; k < kCostGroup; ++k) {
    const WinState* st = P.ws + d[k].window;
    if (on[k] && !ignore_active && !st->active) on[k] = false;
    if (on[k] && which == 1 && st->solve_failed) on[k] = false;
    sel[k] = st->sel ^ which;
    any = any || on[k];
  }
  if (!any) return;
  int ip[kCostGroup], iq[kCostGroup];
  double2 uv[kCostGroup];
  bool live[kCostGroup];
#pragma unroll
  for (int k = 0; k < kCostGroup; ++k) {
    live[k] = on[k] && t < d[k].n_obs;
    const long long o = d[k].obs_begin + (live[k] ? t : 0);
    ip[k] = P.obs_pose[o];
    iq[k] = P.obs_dp[o];
    uv[k] = reinterpret_cast<const double2*>(P.obs_uv)[o];
  }
  double pose[kCostGroup][7], pt[kCostGroup][3];
#pragma unroll
  for (int k = 0; k < kCostGroup; ++k) {
    const double* ps = P.poses[sel[k]] + 7ll * ip[k];
    const double* p = P.points[sel[k]] + 3ll * iq[k];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[k][i] = ps[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pt[k][i] = p[i];
  }
  double acc[2 * kCostGroup];
#pragma unroll
  for (int k = 0; k < kCostGroup; ++k) {
    acc[2 * k] = 0.0; acc[2 * k + 1] = 0.0;
    if (live[k]) {
      const CameraK K = P.wd[d[k].window].K;
      double r[2];
      residual_only(K, pose[k], pt[k], uv[k].x, uv[k].y, r);
      const double s = r[0] * r[0] + r[1] * r[1];
      double rho0 = s;
      if (lm.loss == HIPBA_LOSS_HUBER) {
        double f;
        huber_correct(lm.loss_scale, s, &rho0, &f);
      }
      acc[2 * k] = 0.5 * rho0;
      acc[2 * k + 1] = isfinite(rho0) ? 0.0 : 1.0;
    }
  }
  block_sum<2 * kCostGroup>(acc, red);
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < kCostGroup; ++k) {
      if (!on[k]) continue;
      P.partial[(long long)(b0 + k) * kPartialStride + 0] = acc[2 * k];
      P.partial[(long long)(b0 + k) * kPartialStride + 1] = acc[2 * k + 1];
    }
  }
}

// ------------------------------------------------------------------------------------------
// reduced system initialisation, one thread per variable pose:
//   S(c,c) = U_c + D_c^2,  b_c = g_c,  and the pose part of |x - Plus(x, -g)|.
// S must have been zeroed on the stream before this kernel.
// ------------------------------------------------------------------------------------------
// ATOMIC: S and b of the window were zeroed earlier and other workgroups of the same launch are subtracting their
// Schur products from them already: the diagonal block and the gradient are added, not stored (packed S only).
template <bool ATOMIC = false>
__device__ __forceinline__ void init_reduced_pose(const DevP& P, const LmParams& lm, int g, int owner) {
  const int c = P.slot_pose[g];
  const int w = P.pose_window[c];
  WinState* st = P.ws + w;
  if (!st->active) return;
  const WinDesc wd = P.wd[w];
  const int slot = g - wd.cam_base;
  double* Sb = P.S + wd.S_off + 36ll * tri_index(slot, slot, wd.ncam);
  const double* U = P.U + 36ll * g;
  const double inv_radius = 1.0 / st->radius;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double v = U[6 * i + j];
      if (i == j) {
        double dg = fmin(fmax(v, lm.min_lm_diagonal), lm.max_lm_diagonal);
        v += dg * inv_radius;
      }
      if (P.dense_S) {
        // lower triangle only; in a sharded solve one rank contributes U + D^2 (the matrix was zeroed before)
        if (j >= i && owner) P.dense_S[(long long)(6 * slot + i) * P.dense_ld + 6 * slot + j] = v;
      } else if (ATOMIC) {
        atomic_add_f64(Sb + 6 * i + j, v);
      } else if (owner) {
        Sb[6 * i + j] = v;  // sharded solves: one rank contributes U + D^2 and g_c
      }
    }
  }
  double neg[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double gs = P.gc[6ll * g + i];
    if (ATOMIC) atomic_add_f64(P.b + 6ll * g + i, gs);
    else P.b[6ll * g + i] = owner ? gs : 0.0;
    neg[i] = -gs / P.scale_c[6ll * g + i];
  }
  // gradient norms are defined through the retraction: |x - Plus(x, -g)|
  const double* x = P.poses[st->sel] + 7ll * c;
  double xp[7], xx[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) xx[i] = x[i];
  se3_plus(xx, neg, xp);
  double m = 0.0, s2 = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const double dlt = xx[i] - xp[i];
    m = fmax(m, fabs(dlt));
    s2 += dlt * dlt;
  }
  // stash per pose in yc (overwritten by the solver afterwards): [0]=max, [1]=sum of squares
  P.yc[6ll * g + 0] = m;
  P.yc[6ll * g + 1] = s2;
  if (P.cam_tab) {
    // what k_schur_gather needs of a pose, once per iteration instead of once per product
    double* ct = P.cam_tab + 18ll * g;
    double Rt[9];
    quat_to_Rt(xx, Rt);
#pragma unroll
    for (int i = 0; i < 9; ++i) ct[i] = Rt[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ct[9 + i] = Rt[3 * i] * xx[4] + Rt[3 * i + 1] * xx[5] + Rt[3 * i + 2] * xx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ct[12 + i] = P.scale_c[6ll * g + i];
  }
}

__global__ void k_init_reduced(DevP P, LmParams lm, int owner) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.n_act_poses) return;
  init_reduced_pose(P, lm, g, owner);
}

// ------------------------------------------------------------------------------------------
// camera blocks: U_c = sum Jc^T Jc (6x6), g_c = sum Jc^T r over the camera's observations,
// one workgroup per variable pose, walking its camera-major observation list.
// ------------------------------------------------------------------------------------------
// fuse_init: thread 0 goes on to do k_init_reduced's work for this pose (S must have been zeroed before this
// launch then): one launch less per iteration.
// ATOMIC_INIT: the same with S and b added to instead of stored (see init_reduced_pose<true>).
template <bool ATOMIC_INIT>
__device__ __forceinline__ void camera_blocks_body(const DevP& P, const LmParams& lm, int g, int use_scale, int fuse_init,
                                                   double* red) {
  const int c = P.slot_pose[g];
  const int w = P.pose_window[c];
  const WinState* st = P.ws + w;
  if (!st->active) return;
  const int sel = st->sel;
  const CameraK K = P.wd[w].K;
  double pose[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) pose[i] = P.poses[sel][7ll * c + i];
  double sc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) sc[i] = use_scale ? P.scale_c[6ll * g + i] : 1.0;
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0;
  const long long o1 = P.cm_start[g + 1];
  for (long long o = P.cm_start[g] + threadIdx.x; o < o1; o += kBlock) {
    const double* p = P.points[sel] + 3ll * P.cm_dp[o];
    const double2 uv = reinterpret_cast<const double2*>(P.cm_uv)[o];
    double r[2], Jc[12];
    eval_obs<false>(K, lm, pose, p, uv.x, uv.y, sc, nullptr, r, Jc, nullptr);
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = i; j < 6; ++j) { acc[k] += Jc[i] * Jc[j] + Jc[6 + i] * Jc[6 + j]; ++k; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
  }
  block_sum<27>(acc, red);
  if (threadIdx.x == 0) {
    double* U = P.U + 36ll * g;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = i; j < 6; ++j) { U[6 * i + j] = acc[k]; U[6 * j + i] = acc[k]; ++k; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) P.gc[6ll * g + i] = acc[21 + i];
    if (ATOMIC_INIT) init_reduced_pose<true>(P, lm, g, 1);
    else if (fuse_init) init_reduced_pose(P, lm, g, 1);
  }
}

__global__ __launch_bounds__(kBlock) void k_camera_blocks(DevP P, LmParams lm, int use_scale, int fuse_init) {
  __shared__ double red[27 * 4];
  camera_blocks_body<false>(P, lm, blockIdx.x, use_scale, fuse_init, red);
}

// Jacobi scaling of the pose columns, computed once at iteration 0 from the unscaled blocks
// (Ceres: scale = 1 / (1 + sqrt(squared column norm))), then folded into U and g.
__device__ __forceinline__ void make_scale_c_pose(const DevP& P, int g, int jacobi) {
  double* U = P.U + 36ll * g;
  double s[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    s[i] = jacobi ? 1.0 / (1.0 + sqrt(U[7 * i])) : 1.0;
    P.scale_c[6ll * g + i] = s[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) U[6 * i + j] *= s[i] * s[j];
    P.gc[6ll * g + i] *= s[i];
  }
}

__global__ void k_make_scale_c(DevP P, int jacobi) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.n_act_poses) return;
  make_scale_c_pose(P, g, jacobi);
}

// the same, and the thread goes on to initialise the reduced system for the first pass (k_init_reduced's work: it
// needs this pose's scaled block only); S and b must hold zeros
__global__ void k_scale_init_reduced(DevP P, LmParams lm, int jacobi) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.n_act_poses) return;
  make_scale_c_pose(P, g, jacobi);
  init_reduced_pose(P, lm, g, 1);
}

// Jacobi scaling of the landmark columns (iteration 0): unscaled squared column norms per landmark.
__device__ __forceinline__ void point_colnorm_body(const DevP& P, const LmParams& lm, int blk, double* sh) {
  const BlockDesc d = P.blocks[blk];
  const WinState* st = P.ws + d.window;
  if (!st->active || d.pad) return;  // (a piece of a long landmark: k_long_point's)
  const int sel = st->sel;
  const CameraK K = P.wd[d.window].K;
  const int t = threadIdx.x;
  int pp = -1;
  if (t < d.n_obs) {
    const long long o = d.obs_begin + t;
    pp = P.obs_dp[o];
    const double* pose = P.poses[sel] + 7ll * P.obs_pose[o];
    const double* p = P.points[sel] + 3ll * pp;
    const double2 uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
    double r[2], Jc[12], Jp[6];
    eval_obs<true>(K, lm, pose, p, uv.x, uv.y, nullptr, nullptr, r, Jc, Jp);
#pragma unroll
    for (int i = 0; i < 3; ++i) sh[3 * t + i] = Jp[i] * Jp[i] + Jp[3 + i] * Jp[3 + i];
  }
  __syncthreads();
  if (t < d.n_obs) {
    const int s0 = (int)(P.pp_obs_start[pp] - d.obs_begin);
    if (t == s0) {
      const int s1 = (int)(P.pp_obs_start[pp + 1] - d.obs_begin);
      double n[3] = {0.0, 0.0, 0.0};
      for (int j = s0; j < s1; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) n[i] += sh[3 * j + i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) P.scale_p[3ll * pp + i] = lm.jacobi_scaling ? 1.0 / (1.0 + sqrt(n[i])) : 1.0;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_point_colnorm(DevP P, LmParams lm) {
  __shared__ double sh[kBlock * 3];
  point_colnorm_body(P, lm, blockIdx.x, sh);
}

// iteration 0 of small windows: the unscaled camera blocks (the first P.n_act_poses workgroups) and the landmarks'
// column norms (the others) have nothing to do with each other: one launch
__global__ __launch_bounds__(kBlock) void k_first_blocks(DevP P, LmParams lm) {
  __shared__ double sh[kBlock * 3];
  if ((int)blockIdx.x < P.n_act_poses) camera_blocks_body<false>(P, lm, blockIdx.x, 0, 0, sh);
  else point_colnorm_body(P, lm, (int)blockIdx.x - P.n_act_poses, sh);
}


// ------------------------------------------------------------------------------------------
// landmark elimination (k_schur_point + k_schur_camera of the design):
// per landmark V = sum Jp^T Jp + D^2 (3x3), g_p, V^-1; per observation W = Jc^T Jp (6x3),
// Y = W V^-1; reduced system S(ci,cj) -= Y_i W_j^T, b(ci) -= Y_i g_p.
//
// Phase 1: one lane per observation evaluates r, Jc, Jp and stages its V/g_p contribution
//          in LDS.  Phase 2: every lane of a landmark's segment sums the segment (the same
//          LDS words are broadcast to all of them), inverts V and stages W, Y.  Phase 3:
//          groups of 12 lanes walk the block's precomputed pair list; a group owns one 6x6
//          block at a time and its three atomic wave-instructions each cover 96 contiguous
//          bytes of that block (S is stored block-major for exactly this reason).
// LDS_ACC: the window's whole packed S and b fit in LDS; accumulate there and flush once.
// ------------------------------------------------------------------------------------------
#ifdef HIPBA_PHASE_TIMING
__device__ unsigned long long g_ticks[8 * 4096];
#endif

// 1/sqrt(x): hardware estimate + two Newton steps (full double precision, no division)
__device__ __forceinline__ double rsqrt_newton(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}

constexpr int kPairGroup = 12;
constexpr int kPairGroups = kBlock / kPairGroup;  // 21
// LDS doubles per lane for Z = W C (6x3), C the lower Cholesky factor of V^-1: Y_i W_j^T = W_i V^-1 W_j^T = Z_i Z_j^T,
// so one 18-word factor per observation serves both sides of a product (half the staging of Y and W, which is what
// lets two workgroups share a CU); odd, so lanes fall on different banks
constexpr int kYW = 19;

// Pair products of one block: a lane group takes whole runs (same destination), sums each in registers
// and adds it to the reduced matrix once.  WIDE: twelve lanes per product, three outputs each, the three
// atomics of a group covering 96 contiguous bytes of the destination block (for S in HBM).  Otherwise one
// lane per run (for S in LDS, where the phase is bound by its LDS reads: 36 reads per 108 multiply-adds instead of the
// 27 per 54 of two lanes sharing a product).
template <bool WIDE>
__device__ __forceinline__ void pair_products(const unsigned short* pairs, const RunDesc* runs, int n_runs,
                                              const double* sYW, double* Sbase, int t) {
  if (WIDE) {
    const int grp = t / kPairGroup, k = t - grp * kPairGroup;
    if (grp >= kPairGroups) return;
    const int bcol = k % 6, a0 = k / 6;
    for (int r = grp; r < n_runs; r += kPairGroups) {
      const RunDesc rd = runs[r];
      const int off = rd.first_count >> 8, cnt = rd.first_count & 0xff;
      unsigned short pr[kRunMax];
#pragma unroll
      for (int q = 0; q < kRunMax; ++q) pr[q] = pairs[off + q];
      double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < kRunMax; ++q) {
        if (q < cnt) {
          const int i = pr[q] & 0xff, j = pr[q] >> 8;
          const double* Yi = sYW + kYW * i;
          const double* Wj = sYW + kYW * j + 3 * bcol;
          const double w0 = Wj[0], w1 = Wj[1], w2 = Wj[2];
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int a = a0 + 2 * m;
            acc[m] += Yi[3 * a] * w0 + Yi[3 * a + 1] * w1 + Yi[3 * a + 2] * w2;
          }
        }
      }
      double* dst = Sbase + 36ll * rd.dest + bcol;
#pragma unroll
      for (int m = 0; m < 3; ++m) atomic_add_f64(dst + 6 * (a0 + 2 * m), -acc[m]);
    }
  } else {
    for (int r = t; r < n_runs; r += kBlock) {
      const RunDesc rd = runs[r];
      const int off = rd.first_count >> 8, cnt = rd.first_count & 0xff;
      unsigned short pr[kRunMax];
#pragma unroll
      for (int q = 0; q < kRunMax; ++q) pr[q] = pairs[off + q];
      double acc[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
      for (int q = 0; q < kRunMax; ++q) {
        if (q < cnt) {
          const int i = pr[q] & 0xff, j = pr[q] >> 8;
          const double* Yi = sYW + kYW * i;
          const double* Wj = sYW + kYW * j;
          double wj[18];
#pragma unroll
          for (int ii = 0; ii < 18; ++ii) wj[ii] = Wj[ii];
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            const double y0 = Yi[3 * a], y1 = Yi[3 * a + 1], y2 = Yi[3 * a + 2];
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[6 * a + c] += y0 * wj[3 * c] + y1 * wj[3 * c + 1] + y2 * wj[3 * c + 2];
          }
        }
      }
      double* dst = Sbase + 36ll * rd.dest;
#pragma unroll
      for (int a = 0; a < 36; ++a) atomic_add_f64(dst + a, -acc[a]);
    }
  }
}

// MODE 0: fp64 atomics straight into S, one workgroup per block.
// MODE 2 (gathered): only V^-1, g_p and V^-1 g_p are formed here and left in the landmark's record;
//   k_schur_gather then lets every block of S sum its own products from a list of observation pairs,
//   recomputing each from the record and the two poses (no atomics on S, fixed order).
// MODE 1 / 3 (windows of at most kSmallMaxN/6 poses): `parts` workgroups per window, each walking every
//   parts-th block of the window; b is accumulated in LDS and flushed once per workgroup, and so is the
//   whole packed S when it fits (MODE 1; MODE 3 sends S to HBM with atomics).  With S in LDS a lane forms whole
//   products (one run each) instead of sharing one with eleven others: a third of the LDS bytes read per multiply-add.
// MODE 4 (small windows, owned destinations): no pair or run lists at all.  A lane owns one 6x6 block (s_i, s_j) of the
//   window's packed S for the whole launch and keeps it in 36 registers; for every landmark of a block it looks up
//   whether poses s_i and s_j both observe it (one broadcast LDS read: the landmark's mask of observing slots and
//   the lane of its first variable observation -- a segment is sorted by slot, so the observation of slot s sits
//   popcount(mask below s) lanes further) and adds Z_i Z_j^T.  Nothing is added atomically inside the loop; the
//   registers go to S once at the end.  Windows of few poses (at most 128 blocks in S) split a block's landmarks
//   over several lane groups per destination and sum the groups in LDS first; windows of more than 22 poses (more
//   than 256 blocks) are covered by several workgroups per part, each repeating the cheap phases 1 and 2 and owning
//   its 256 destinations.  Needs: no pose observing a landmark twice, at most 32 poses (the mask).
constexpr int kOwnedSplitBlocks = 128;  // windows of at most this many blocks in S: several lane groups per block

template <int MODE>
__device__ __forceinline__ void point_schur_body(const DevP& P, const LmParams& lm, int parts, const int bid,
                                                 const int dgroups = 1) {
  constexpr bool OWNED = MODE == 4;
  constexpr bool SMALL = MODE == 1 || MODE == 3 || OWNED;
  constexpr bool LDS_S = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int kStage = MODE == 2 ? 9 : kYW;  // gathered mode stages only the phase-1 words
  double* sYW = smem;                         // [kBlock][kYW]: Z (18 words); phase 1 aliases 9 words/lane
  int* sslot = reinterpret_cast<int*>(smem + kBlock * kStage);  // [kBlock] local slot or -1
  double* red = smem + kBlock * kStage + kBlock / 2;            // [8]
  unsigned short* sPairs = reinterpret_cast<unsigned short*>(red + 8);   // [kLdsPairs] (not in staged mode)
  RunDesc* sRuns = reinterpret_cast<RunDesc*>(sPairs + kLdsPairs);      // [kLdsRuns]
  uint2* sLm = reinterpret_cast<uint2*>(red + 8);                       // OWNED (instead of the lists): [kBlock] (mask, first lane)
  double* accB = OWNED ? reinterpret_cast<double*>(sLm + kBlock)
                       : reinterpret_cast<double*>(sRuns + kLdsRuns);   // SMALL: b, then (LDS_S, OWNED) packed S
  double* accS = accB + kSmallMaxN;

  const int t = threadIdx.x;
  HIPBA_TICK(0);
  int blk = bid, blk_end = bid + 1, blk_step = 1, w, dg = 0;
  if (SMALL) {
    const int wp = OWNED ? bid / dgroups : bid;
    if (OWNED) dg = bid - wp * dgroups;
    w = wp / parts;
    const int part = wp - w * parts;
    blk = P.wd[w].pm_blk_begin + part;
    blk_end = P.wd[w].pm_blk_end;
    blk_step = parts;
    if (blk >= blk_end) return;
  } else {
    w = P.blocks[blk].window;
  }
  const WinState* st = P.ws + w;
  if (!st->active) return;
  const WinDesc wd = P.wd[w];
  const int sel = st->sel;
  const double inv_radius = 1.0 / st->radius;
  // owned destinations: this lane's block (own_i, own_j) of S, and which of the own_k lane groups of that block it is
  const int ndest = wd.ncam * (wd.ncam + 1) / 2;
  int own_i = 0, own_j = 0, own_sub = 0, own_k = 1, own_dest = 0;
  bool own = false;
  if (OWNED) {
    if (dg > 0 && dg * kBlock >= ndest) return;  // (a window of fewer poses than the launch was sized for)
    if (ndest == 0) { own = false; }  // (every pose constant: the landmark quantities are still formed)
    else if (ndest <= kOwnedSplitBlocks) { own_k = kBlock / ndest; own_sub = t / ndest; own_dest = t - own_sub * ndest; own = own_sub < own_k; }
    else { own_dest = dg * kBlock + t; own = own_dest < ndest; }
    if (own) {
      int rem = own_dest;
      while (rem >= wd.ncam - own_i) { rem -= wd.ncam - own_i; ++own_i; }
      own_j = own_i + rem;
    }
  }
  double own_acc[OWNED ? 36 : 1];
#pragma unroll
  for (int i = 0; i < (OWNED ? 36 : 1); ++i) own_acc[i] = 0.0;
  // what goes to S from LDS at the end: the whole packed matrix (MODE 1), this workgroup's destinations (MODE 4)
  const int own_base = OWNED ? dg * kBlock : 0;
  const int nS = LDS_S ? ndest * 36 : OWNED ? max(0, min(ndest - own_base, kBlock)) * 36 : 0;
  const int nS_zero = (LDS_S || own_k > 1) ? nS : 0;
  if (SMALL) {
    for (int i = t; i < 6 * wd.ncam; i += kBlock) accB[i] = 0.0;
    for (int i = t; i < nS_zero; i += kBlock) accS[i] = 0.0;
  }
  double gmax = 0.0, g2 = 0.0;
  const int first_blk = blk;

  for (; blk < blk_end; blk += blk_step) {
    const BlockDesc d = P.blocks[blk];
    if (MODE == 2 && d.pad) return;  // a piece of a long landmark (gathered mode only): k_long_point's
    if (SMALL) __syncthreads();  // accumulators zeroed / previous block's products read
    HIPBA_TICK(1);
    int pp0 = 0, n_lm = 0;
    if (OWNED) {
      pp0 = P.obs_dp[d.obs_begin];
      n_lm = P.obs_dp[d.obs_begin + d.n_obs - 1] - pp0 + 1;  // (consecutive landmarks, at least one observation each)
      sLm[t] = make_uint2(0u, 0xffffffffu);
    }
    // the block's pair and run lists go to LDS when they fit; the loads are issued here, ahead of the
    // observation work, and land in LDS together with Y and W
    const bool lists_in_lds = MODE != 2 && !OWNED && d.n_pairs <= kLdsPairs && d.n_runs <= kLdsRuns;
    // (native vector types, named one by one: an array of HIP's uint4 class is not promoted to registers -- it sat in
    // scratch, and every one of these loads waited for its own round trip to memory before the next was issued)
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u pv0 = {0, 0, 0, 0}, pv1 = {0, 0, 0, 0}, rv0 = {0, 0, 0, 0}, rv1 = {0, 0, 0, 0};
    if (lists_in_lds) {
      const v4u* gp = reinterpret_cast<const v4u*>(P.pairs + d.pair_begin);
      const v4u* gr = reinterpret_cast<const v4u*>(P.runs + d.run_begin);
      const int npv = (d.n_pairs + 7) >> 3, nrv = (d.n_runs + 1) >> 1;
      if (t < npv) pv0 = gp[t];
      if (t + kBlock < npv) pv1 = gp[t + kBlock];
      if (t < nrv) rv0 = gr[t];
      if (t + kBlock < nrv) rv1 = gr[t + kBlock];
    }
    double r[2], Jc[12], Jp[6];
    int pp = -1, slot = -1;
    const bool live = t < d.n_obs;
    if (live) {
      const long long o = d.obs_begin + t;
      pp = P.obs_dp[o];
      const int c = P.obs_pose[o];
      const int g = P.pose_gslot[c];
      slot = g < 0 ? -1 : g - wd.cam_base;
      const double* pose = P.poses[sel] + 7ll * c;
      const double* p = P.points[sel] + 3ll * pp;
      const double2 uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
      double lf = 1.0;
      eval_obs<true>(wd.K, lm, pose, p, uv.x, uv.y, g < 0 ? nullptr : P.scale_c + 6ll * g, P.scale_p + 3ll * pp, r,
                     Jc, Jp, &lf);
      if (MODE == 2 && lm.loss != HIPBA_LOSS_NONE) P.fw[o] = lf;
      double* cw = sYW + 9 * t;
      cw[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3];
      cw[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4];
      cw[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
      cw[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4];
      cw[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5];
      cw[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
      cw[6] = Jp[0] * r[0] + Jp[3] * r[1];
      cw[7] = Jp[1] * r[0] + Jp[4] * r[1];
      cw[8] = Jp[2] * r[0] + Jp[5] * r[1];
    }
    __syncthreads();
    HIPBA_TICK(2);

    double Z[18];
    if (live) {
      const int s0 = (int)(P.pp_obs_start[pp] - d.obs_begin);
      const int s1 = (int)(P.pp_obs_start[pp + 1] - d.obs_begin);
      double V[6] = {0, 0, 0, 0, 0, 0}, gpv[3] = {0, 0, 0};
      for (int j = s0; j < s1; ++j) {
        const double* cw = sYW + 9 * j;
#pragma unroll
        for (int i = 0; i < 6; ++i) V[i] += cw[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) gpv[i] += cw[6 + i];
      }
      // Levenberg-Marquardt diagonal: clamp(column norm^2) / radius
      V[0] += fmin(fmax(V[0], lm.min_lm_diagonal), lm.max_lm_diagonal) * inv_radius;
      V[3] += fmin(fmax(V[3], lm.min_lm_diagonal), lm.max_lm_diagonal) * inv_radius;
      V[5] += fmin(fmax(V[5], lm.min_lm_diagonal), lm.max_lm_diagonal) * inv_radius;
      double Vi[6];
      inv3_sym(V, Vi);
      if (t == s0 && dg == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) P.Vinv[6ll * pp + i] = Vi[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          P.gp[3ll * pp + i] = gpv[i];
          const double sp = P.scale_p[3ll * pp + i];
          const double gu = gpv[i] / sp;
          gmax = fmax(gmax, fabs(gu));
          g2 += gu * gu;
        }
        if (MODE == 2) {
          double* rec = P.prec + 16ll * pp;
          const double* p = P.points[sel] + 3ll * pp;
#pragma unroll
          for (int i = 0; i < 3; ++i) { rec[i] = p[i]; rec[3 + i] = P.scale_p[3ll * pp + i]; }
#pragma unroll
          for (int i = 0; i < 6; ++i) rec[6 + i] = Vi[i];
          // scale_p o (V^-1 g_p) rides in the spare words: k_schur_gather forms the reduced right-hand side from it
          rec[12] = P.scale_p[3ll * pp] * (Vi[0] * gpv[0] + Vi[1] * gpv[1] + Vi[2] * gpv[2]);
          rec[13] = P.scale_p[3ll * pp + 1] * (Vi[1] * gpv[0] + Vi[3] * gpv[1] + Vi[4] * gpv[2]);
          rec[14] = P.scale_p[3ll * pp + 2] * (Vi[2] * gpv[0] + Vi[4] * gpv[1] + Vi[5] * gpv[2]);
          rec[15] = 0.0;  // the whole line is staged through LDS by the gather: keep it finite
        }
      }
      if (MODE != 2 && slot >= 0) {
        // V^-1 = C C^T (C lower triangular; V^-1 is positive definite with V), Z = W C with W = Jc^T Jp
        double c00, c10, c20, c11, c21, c22;
        {
          const double r0 = rsqrt_newton(Vi[0]);
          c00 = Vi[0] * r0; c10 = Vi[1] * r0; c20 = Vi[2] * r0;
          const double d1 = Vi[3] - c10 * c10;
          const double r1 = rsqrt_newton(d1);
          c11 = d1 * r1; c21 = (Vi[4] - c20 * c10) * r1;
          const double d2 = Vi[5] - c20 * c20 - c21 * c21;
          c22 = d2 * rsqrt_newton(d2);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double w0 = Jc[i] * Jp[0] + Jc[6 + i] * Jp[3];
          const double w1 = Jc[i] * Jp[1] + Jc[6 + i] * Jp[4];
          const double w2 = Jc[i] * Jp[2] + Jc[6 + i] * Jp[5];
          Z[3 * i + 0] = w0 * c00 + w1 * c10 + w2 * c20;
          Z[3 * i + 1] = w1 * c11 + w2 * c21;
          Z[3 * i + 2] = w2 * c22;
        }
        // b_c -= Y g_p = Z (C^T g_p) (gathered mode leaves this to k_schur_gather: one writer per pose instead of atomics)
        if (MODE != 2 && dg == 0) {
          const double u0 = c00 * gpv[0] + c10 * gpv[1] + c20 * gpv[2], u1 = c11 * gpv[1] + c21 * gpv[2], u2 = c22 * gpv[2];
          double* bdst = SMALL ? accB + 6 * slot : P.b + 6ll * (wd.cam_base + slot);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double v = Z[3 * i] * u0 + Z[3 * i + 1] * u1 + Z[3 * i + 2] * u2;
            atomic_add_f64(bdst + i, -v);
          }
        }
      }
    }
    __syncthreads();  // every lane has finished reading the phase-1 words
    HIPBA_TICK(3);
    if (lists_in_lds) {
      reinterpret_cast<v4u*>(sPairs)[t] = pv0;
      reinterpret_cast<v4u*>(sPairs)[t + kBlock] = pv1;
      reinterpret_cast<v4u*>(sRuns)[t] = rv0;
      reinterpret_cast<v4u*>(sRuns)[t + kBlock] = rv1;
    }
    if (MODE != 2 && live) {
      sslot[t] = slot;
      if (slot >= 0) {
        double* z = sYW + kYW * t;
#pragma unroll
        for (int i = 0; i < 18; ++i) z[i] = Z[i];
        if (OWNED) {
          atomicOr(&sLm[pp - pp0].x, 1u << slot);
          atomicMin(&sLm[pp - pp0].y, (unsigned)t);
        }
      }
    }
    __syncthreads();
    HIPBA_TICK(4);
    if (OWNED && own) {
      const unsigned bit_i = 1u << own_i, bit_j = 1u << own_j;
      for (int l = own_sub; l < n_lm; l += own_k) {
        const uint2 m = sLm[l];
        if ((m.x & bit_i) && (m.x & bit_j)) {
          const double* Zi = sYW + kYW * (m.y + __popc(m.x & (bit_i - 1)));
          const double* Zj = sYW + kYW * (m.y + __popc(m.x & (bit_j - 1)));
          double zj[18];
#pragma unroll
          for (int ii = 0; ii < 18; ++ii) zj[ii] = Zj[ii];
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            const double y0 = Zi[3 * a], y1 = Zi[3 * a + 1], y2 = Zi[3 * a + 2];
#pragma unroll
            for (int c = 0; c < 6; ++c) own_acc[6 * a + c] -= y0 * zj[3 * c] + y1 * zj[3 * c + 1] + y2 * zj[3 * c + 2];
          }
        }
      }
    }

    // phase 3: pair products (staged mode: every product to its own slot; otherwise run by run)
    if (MODE != 2 && !OWNED) {
      double* Sbase = LDS_S ? accS : P.S + wd.S_off;
      if (lists_in_lds) pair_products<!LDS_S>(sPairs, sRuns, d.n_runs, sYW, Sbase, t);
      else pair_products<!LDS_S>(P.pairs + d.pair_begin, P.runs + d.run_begin, d.n_runs, sYW, Sbase, t);
    }
  }

  HIPBA_TICK(5);
  // landmark part of the gradient norms: one entry per block; a workgroup that walked several blocks
  // reports under its first one and zeroes the others
  gmax = block_max(gmax, red);
  double gs[1] = {g2};
  block_sum<1>(gs, red);
  if (t == 0 && dg == 0) {
    P.gpart[2ll * first_blk + 0] = gmax;
    P.gpart[2ll * first_blk + 1] = gs[0];
    if (SMALL) {
      for (int bb = first_blk + blk_step; bb < blk_end; bb += blk_step) {
        P.gpart[2ll * bb + 0] = 0.0;
        P.gpart[2ll * bb + 1] = 0.0;
      }
    }
  }
  HIPBA_TICK(6);
  if (OWNED && own) {
    // the lane's block of S goes to LDS (several lane groups per block: summed there), so that the sweep below adds
    // it to S with neighbouring lanes on neighbouring words -- a lane adding its own 36 words would make every
    // one of them a memory request of its own
    double* dst = accS + 36 * (own_dest - own_base);
    if (own_k > 1) {
#pragma unroll
      for (int a = 0; a < 36; ++a)
        if (own_acc[a] != 0.0) atomic_add_f64(dst + a, own_acc[a]);
    } else {
#pragma unroll
      for (int a = 0; a < 36; ++a) dst[a] = own_acc[a];
    }
  }
  if (SMALL) {
    __syncthreads();
    double* Sg = P.S + wd.S_off + 36ll * own_base;
    for (int i = t; i < nS; i += kBlock) {
      const double v = accS[i];
      if (v != 0.0) atomic_add_f64(Sg + i, v);
    }
    double* bg = P.b + 6ll * wd.cam_base;
    for (int i = t; i < 6 * wd.ncam; i += kBlock) {
      const double v = accB[i];
      if (v != 0.0) atomic_add_f64(bg + i, v);
    }
  }
  HIPBA_TICK(7);
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_point_schur(DevP P, LmParams lm, int parts) {
  point_schur_body<MODE>(P, lm, parts, blockIdx.x);
}

// Small windows, second pass on: the camera blocks and the elimination of the landmarks in ONE launch.  The first
// P.n_act_poses workgroups do k_camera_blocks' work for one pose each and ADD U + D^2 and g to the window's S and b;
// the others do k_point_schur<MODE>'s.  S and b must hold zeros when the launch starts (k_window_control of the
// pass before leaves them so).  Both halves are a few dozen workgroups of a few microseconds' dependent work: side
// by side on the chip they cost the longer of the two, and one launch boundary less.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_small_linearize(DevP P, LmParams lm, int parts, int dgroups) {
  static_assert(MODE == 1 || MODE == 3 || MODE == 4, "small-window modes only");
  if ((int)blockIdx.x < P.n_act_poses) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    camera_blocks_body<true>(P, lm, blockIdx.x, 1, 1, smem);
  } else {
    point_schur_body<MODE>(P, lm, parts, (int)blockIdx.x - P.n_act_poses, dgroups);
  }
}

__global__ __launch_bounds__(kBlock) void k_point_schur_owned(DevP P, LmParams lm, int parts, int dgroups) {
  point_schur_body<4>(P, lm, parts, blockIdx.x, dgroups);
}

// gathered mode: S(block d) -= sum over its observation pairs of Y_i W_j^T.  A wave owns a chunk of the
// destination-ordered pair list (whole destinations) and takes it 64 pairs at a time, one pair per lane:
//   the 64 landmark records (one aligned 128-byte line each) are loaded line by line, eight lanes to a line, and
//   handed to their lanes through LDS; a lane rebuilds the two observations' Jacobians from the record and the
//   two rows of the camera table, and forms the product in the factored form
//     Y_i W_j^T = Jc_i^T [ (Jp_i V^-1 Jp_j^T) Jc_j ]            (2x2 core, then 2x6, then 6x6);
//   the 64 products go to LDS transposed, and lanes 0-35 (lane e = 6 r + c owns entry (r, c)) sum them
//   destination by destination in list order.
// HBM traffic per product: 8 bytes of list and one 128-byte record (the records of a million landmarks are
// 128 MB and largely stay in the last-level cache); nothing per observation.  Off-diagonal blocks were zeroed
// and have this one writer: plain stores; diagonal blocks already hold U + D^2: one atomic per entry.
// Fixed summation order: S is reproducible run to run.
constexpr int kGatherRow = 65;                // LDS doubles per row of 64 products (odd: rows on different banks)
constexpr int kGatherRec = 17;                // LDS doubles per staged landmark record
constexpr int kGatherLds = 36 * kGatherRow + 16;  // doubles per wave (the records, 64 x 17, use the same words first)

#define HIPBA_WAVE_SYNC()                                  \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// Jacobians of one observation from the camera table row (R^T, R^T t, column scales) and the landmark: the pose
// columns scaled, the landmark columns not yet
__device__ __forceinline__ void gather_side(const CameraK& K, const double* cam, const double* p, double* Jc, double* Jp) {
  double Rt[9], pc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rt[i] = cam[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) pc[i] = Rt[3 * i] * p[0] + Rt[3 * i + 1] * p[1] + Rt[3 * i + 2] * p[2] - cam[9 + i];
  projection_jacobians(K, Rt, pc, Jc, Jp);
#pragma unroll
  for (int i = 0; i < 6; ++i) { const double s = cam[12 + i]; Jc[i] *= s; Jc[6 + i] *= s; }
}

__global__ __launch_bounds__(kBlock) void k_schur_gather(DevP P, LmParams lm) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (!P.ws[0].active) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunk = blockIdx.x * (kBlock / 64) + wave;
  if (chunk >= P.n_chunks) return;  // whole waves leave; nothing below synchronises across waves
  double* sT = smem + wave * kGatherLds;
  const CameraK K = P.wd[0].K;
  const int ncam = P.wd[0].ncam;
  const double* cam_tab = P.cam_tab + 18ll * P.wd[0].cam_base;
  double* S = P.S + P.wd[0].S_off;
  const bool robust = lm.loss != HIPBA_LOSS_NONE;
  const int e = lane < 36 ? lane : 0;
  const long long q0 = P.chunk_start[2 * chunk], q1 = P.chunk_start[2 * chunk + 1];
  // the pad column of every row and the words after the last row are read (and multiplied by zero) by the sums
  if (lane < 36) sT[lane * kGatherRow + 64] = 0.0;
  if (lane < 16) sT[36 * kGatherRow + lane] = 0.0;
  int cur = -1, cur_diag = 0;  // destination being summed by lanes 0-35 (wave-uniform)
  int cur_a = 0, cur_b = 0;    // its row and column slot
  double acc = 0.0;
  double* const dense = P.dense_S;
  const int er = e / 6, ec = e % 6;
  // word of the right-hand side a lane sums in a diagonal destination (see the products below), or -1
  const int e_rhs = (lane < 36 && er >= 1 && er <= 3 && ec < er) ? er * (er - 1) / 2 + ec : -1;
  double* const bw = P.b + 6ll * P.wd[0].cam_base;
  auto flush = [&]() {
    if (cur >= 0 && lane < 36) {
      // (a diagonal block is symmetric: the packed matrix, which keeps all of it, gets the words that carry the
      // right-hand side from their mirror images)
      double mirror = 0.0;
      if (cur_diag && !dense) mirror = __shfl(acc, 6 * ec + er);  // (wave-uniform condition)
      if (cur_diag && e_rhs >= 0) {
        bw[6 * cur_a + e_rhs] -= acc;  // this wave is the only writer of the pose's six words
        if (!dense) atomic_add_f64(S + 36ll * cur + e, -mirror);
      } else if (dense) {
        // straight into the dense lower triangle: consecutive destinations of a row slot are consecutive 48-byte
        // pieces of the same six columns
        double* dst = dense + (long long)(6 * cur_a + er) * P.dense_ld + 6 * cur_b + ec;
        if (!cur_diag) *dst = -acc;
        else if (ec >= er) atomic_add_f64(dst, -acc);
      } else {
        double* dst = S + 36ll * cur + e;
        if (cur_diag) atomic_add_f64(dst, -acc);
        else *dst = -acc;
      }
    }
  };
  // software pipeline: the list entries run two batches ahead of the arithmetic and the landmark records one,
  // so a batch's HBM latency is covered by the previous batch's products
  auto load_entry = [&](long long q) {
    const long long at = q + lane;
    return P.pair_pt[at < q1 ? at : q0];
  };
  auto load_records = [&](const uint2& ent, double2* rv) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const unsigned ppu = __shfl(ent.x, 8 * m + (lane >> 3)) & 0x7fffffffu;  // (bit 31: a pair of an observation with itself)
      rv[m] = reinterpret_cast<const double2*>(P.prec + 16ll * ppu)[lane & 7];
    }
  };
  uint2 pe = load_entry(q0), pe_next = load_entry(q0 + 64);
  double2 rv[8];
  load_records(pe, rv);
  for (long long q = q0; q < q1; q += 64) {
    const int nb = (int)(q1 - q < 64 ? q1 - q : 64);
    const bool live = lane < nb;
    // this batch's camera rows first (loads return in order), then the prefetches
    const int a = (int)(pe.y >> 16), b = (int)(pe.y & 0xffffu);
    double camA[18], camB[18];
    {
      const double2* ca = reinterpret_cast<const double2*>(cam_tab + 18ll * a);
      const double2* cb = reinterpret_cast<const double2*>(cam_tab + 18ll * b);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double2 x = ca[i], y = cb[i];
        camA[2 * i] = x.x; camA[2 * i + 1] = x.y; camB[2 * i] = y.x; camB[2 * i + 1] = y.y;
      }
    }
    double wgt = 1.0, wgt_rhs = 1.0;
    if (robust) {
      const uint2 po = P.pair_obs[q + (live ? lane : 0)];
      const double fa = P.fw[po.x];
      wgt = fa * P.fw[po.y];
      wgt *= wgt;
      wgt_rhs = fa * fa;
    }
    const uint2 pe_next2 = load_entry(q + 128);
    double2 rv_next[8];
    load_records(pe_next, rv_next);
    HIPBA_WAVE_SYNC();  // the previous batch's sums have been read
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      double* w = sT + (8 * m + (lane >> 3)) * kGatherRec + 2 * (lane & 7);
      w[0] = rv[m].x; w[1] = rv[m].y;
    }
    HIPBA_WAVE_SYNC();
    double rec[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) rec[k] = sT[lane * kGatherRec + k];
    HIPBA_WAVE_SYNC();  // records are in registers; the words are free for the products
    double JcA[12], JpA[6], JcB[12], JpB[6];
    const bool self = (pe.x >> 31) != 0u;
    gather_side(K, camA, rec, JcA, JpA);
    if (__all(self)) {
      // a batch inside a diagonal block: every pair is an observation with itself, both sides are the same
#pragma unroll
      for (int i = 0; i < 12; ++i) JcB[i] = JcA[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) JpB[i] = JpA[i];
    } else {
      gather_side(K, camB, rec, JcB, JpB);
    }
    const int diag = a == b ? 1 : 0;
    // A pair of an observation with itself also carries that observation's term of the reduced right-hand side,
    // Jc^T Jp (V^-1 g_p) with scale_p o (V^-1 g_p) from the record and the landmark columns still unscaled.  Only
    // the upper triangle of a diagonal block is kept, so six of the words below its diagonal carry these instead
    // and are summed with the block, in the same fixed order, by the same single writer.
    double bq[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (__any(self)) {
      const double ws = self ? wgt_rhs : 0.0;  // two different observations by one pose: the block only
      const double m0 = (JpA[0] * rec[12] + JpA[1] * rec[13] + JpA[2] * rec[14]) * ws;
      const double m1 = (JpA[3] * rec[12] + JpA[4] * rec[13] + JpA[5] * rec[14]) * ws;
#pragma unroll
      for (int k = 0; k < 6; ++k) bq[k] = JcA[k] * m0 + JcA[6 + k] * m1;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double sp = rec[3 + i];
      JpA[i] *= sp; JpA[3 + i] *= sp; JpB[i] *= sp; JpB[3 + i] *= sp;
    }
    const double* Vi = rec + 6;
    double M[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double c0 = JpA[3 * k] * Vi[0] + JpA[3 * k + 1] * Vi[1] + JpA[3 * k + 2] * Vi[2];
      const double c1 = JpA[3 * k] * Vi[1] + JpA[3 * k + 1] * Vi[3] + JpA[3 * k + 2] * Vi[4];
      const double c2 = JpA[3 * k] * Vi[2] + JpA[3 * k + 1] * Vi[4] + JpA[3 * k + 2] * Vi[5];
#pragma unroll
      for (int l = 0; l < 2; ++l) M[2 * k + l] = (c0 * JpB[3 * l] + c1 * JpB[3 * l + 1] + c2 * JpB[3 * l + 2]) * wgt;
    }
    double T[12];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int c = 0; c < 6; ++c) T[6 * k + c] = M[2 * k] * JcB[c] + M[2 * k + 1] * JcB[6 + c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double val = JcA[r] * T[c] + JcA[6 + r] * T[6 + c];
        // rows 1-3 below the diagonal: (1,0) (2,0) (2,1) (3,0) (3,1) (3,2) -> right-hand side words 0..5
        if (r >= 1 && r <= 3 && c < r && diag) val = bq[r * (r - 1) / 2 + c];
        sT[(6 * r + c) * kGatherRow + lane] = val;
      }
    }
    const int dest = live ? (int)tri_index(a, b, ncam) : -1;
    HIPBA_WAVE_SYNC();
    // the batch's runs of one destination, from a ballot of the run heads.  Lanes 0-35 sum a run in list order,
    // sixteen columns per step with the next run's first sixteen already on their way from LDS; columns
    // past a run's end are read and multiplied by zero (every word read is a finite product or a zeroed pad)
    const int prev_dest = __shfl_up(dest, 1);
    unsigned long long heads = __ballot(live && lane != 0 && dest != prev_dest);
    const double* row = sT + e * kGatherRow;
    double v[16], vn[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k] = row[k]; vn[k] = 0.0; }
    int u0 = 0;
    for (;;) {
      const int u1 = heads ? __builtin_ctzll(heads) : nb;
      heads &= heads - 1;
      if (u1 < nb) {
#pragma unroll
        for (int k = 0; k < 16; ++k) vn[k] = row[u1 + k];
      }
      const int du = __builtin_amdgcn_readlane(dest, u0);
      if (du != cur) {
        flush();
        cur = du;
        cur_diag = __builtin_amdgcn_readlane(diag, u0);
        cur_a = __builtin_amdgcn_readlane(a, u0);
        cur_b = __builtin_amdgcn_readlane(b, u0);
        acc = 0.0;
      }
      const int len = u1 - u0;
      acc += v[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) acc = fma(v[k], k < len ? 1.0 : 0.0, acc);
      for (int u = u0 + 16; u < u1; u += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = row[u + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(v[k], u + k < u1 ? 1.0 : 0.0, acc);
      }
      if (u1 >= nb) break;
      u0 = u1;
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = vn[k];
    }
    pe = pe_next;
    pe_next = pe_next2;
#pragma unroll
    for (int m = 0; m < 8; ++m) rv[m] = rv_next[m];
  }
  flush();
}

// gathered mode, off-diagonal blocks: ONE LANE per block of S.  The pairs of a block sit together in the
// destination-ordered list and there are a few dozen of them (two poses share the landmarks both happened to be
// drawn for), so a lane walks its block's pairs one after the other with the block in 36 registers: no transposition
// through LDS, no segmented sums, no synchronisation at all, and the two poses' rows of the camera table are loaded
// once per block instead of once per pair.  The column scales of the two poses are the same for every pair of a block:
// they multiply the sum once at the end.  The loads run ahead of the arithmetic -- the list entry (and a robust
// loss's two observation numbers) two pairs ahead, the landmark's record (and the two weights) one pair ahead.
// Blocks come longest first, 64 neighbours of the sorted list to a wave, so the lanes of a wave finish together.
// One writer per block, pairs summed in list order: reproducible run to run.  Diagonal blocks (thousands of pairs
// each, and the right-hand side) stay with k_schur_gather.
__global__ __launch_bounds__(kBlock) void k_schur_gather_lanes(DevP P, LmParams lm) {
  if (!P.ws[0].active) return;
  const int n_jobs = P.n_lane_jobs;
  const int gid = blockIdx.x * kBlock + threadIdx.x;
  const uint4 job = P.lane_jobs[gid < n_jobs ? gid : n_jobs - 1];
  const int len = gid < n_jobs ? (int)job.y : 0;
  const long long q0 = job.x;
  const int a = (int)(job.z >> 16), b = (int)(job.z & 0xffffu);
  const CameraK K = P.wd[0].K;
  const double* cam_tab = P.cam_tab + 18ll * P.wd[0].cam_base;
  const bool robust = lm.loss != HIPBA_LOSS_NONE;
  typedef double v2d __attribute__((ext_vector_type(2)));
  double RA[12], RB[12];  // R^T (9) and R^T t (3) of the two poses, for all of the block's pairs
  {
    const v2d* ca = reinterpret_cast<const v2d*>(cam_tab + 18ll * a);
    const v2d* cb = reinterpret_cast<const v2d*>(cam_tab + 18ll * b);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const v2d x = ca[i], y = cb[i];
      RA[2 * i] = x.x; RA[2 * i + 1] = x.y; RB[2 * i] = y.x; RB[2 * i + 1] = y.y;
    }
  }
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  // (an index past the block's last pair reads its first pair again: loaded, never used)
  auto entry = [&](int it) { return P.pair_pt[q0 + (it < len ? it : 0)].x & 0x7fffffffu; };
  auto obs_of = [&](int it) { return P.pair_obs[q0 + (it < len ? it : 0)]; };
  auto record = [&](unsigned pp, v2d* r) {
    const v2d* src = reinterpret_cast<const v2d*>(P.prec + 16ll * pp);
#pragma unroll
    for (int m = 0; m < 6; ++m) r[m] = src[m];  // p (3), scale_p (3), V^-1 (6): the first twelve words
  };
  v2d rec_next[6];
  record(entry(0), rec_next);
  unsigned ent_next = entry(1);
  uint2 po_next = make_uint2(0u, 0u);
  double fa_next = 1.0, fb_next = 1.0;
  if (robust) {
    const uint2 po = obs_of(0);
    fa_next = P.fw[po.x]; fb_next = P.fw[po.y];
    po_next = obs_of(1);
  }
  for (int it = 0; __any(it < len); ++it) {
    double rec[12];
#pragma unroll
    for (int m = 0; m < 6; ++m) { rec[2 * m] = rec_next[m].x; rec[2 * m + 1] = rec_next[m].y; }
    double wgt = 1.0;
    if (robust) {
      wgt = fa_next * fb_next;
      wgt *= wgt;
      fa_next = P.fw[po_next.x]; fb_next = P.fw[po_next.y];
      po_next = obs_of(it + 2);
    }
    record(ent_next, rec_next);
    ent_next = entry(it + 2);
    if (it < len) {
      double pc[3], JcA[12], JpA[6], JcB[12], JpB[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) pc[i] = RB[3 * i] * rec[0] + RB[3 * i + 1] * rec[1] + RB[3 * i + 2] * rec[2] - RB[9 + i];
      projection_jacobians_sparse(K, RB, pc, JcB, JpB);
#pragma unroll
      for (int i = 0; i < 3; ++i) pc[i] = RA[3 * i] * rec[0] + RA[3 * i + 1] * rec[1] + RA[3 * i + 2] * rec[2] - RA[9 + i];
      projection_jacobians_sparse(K, RA, pc, JcA, JpA);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double sp = rec[3 + i];
        JpA[i] *= sp; JpA[3 + i] *= sp; JpB[i] *= sp; JpB[3 + i] *= sp;
      }
      const double* Vi = rec + 6;
      double T[12];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double c0 = JpA[3 * k] * Vi[0] + JpA[3 * k + 1] * Vi[1] + JpA[3 * k + 2] * Vi[2];
        const double c1 = JpA[3 * k] * Vi[1] + JpA[3 * k + 1] * Vi[3] + JpA[3 * k + 2] * Vi[4];
        const double c2 = JpA[3 * k] * Vi[2] + JpA[3 * k + 1] * Vi[4] + JpA[3 * k + 2] * Vi[5];
        const double m0 = (c0 * JpB[0] + c1 * JpB[1] + c2 * JpB[2]) * wgt;
        const double m1 = (c0 * JpB[3] + c1 * JpB[4] + c2 * JpB[5]) * wgt;
#pragma unroll
        for (int c = 0; c < 6; ++c) T[6 * k + c] = m0 * JcB[c] + m1 * JcB[6 + c];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[6 * r + c] = fma(JcA[6 + r], T[6 + c], fma(JcA[r], T[c], acc[6 * r + c]));
    }
  }
  if (len <= 0) return;
  double sa[6], sb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { sa[i] = cam_tab[18ll * a + 12 + i]; sb[i] = cam_tab[18ll * b + 12 + i]; }
  if (P.dense_S) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double* dst = P.dense_S + (long long)(6 * a + r) * P.dense_ld + 6 * b;
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[c] = -(sa[r] * sb[c]) * acc[6 * r + c];
    }
  } else {
    double* dst = P.S + P.wd[0].S_off + 36ll * tri_index(a, b, P.wd[0].ncam);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[6 * r + c] = -(sa[r] * sb[c]) * acc[6 * r + c];
  }
}

// ------------------------------------------------------------------------------------------
// back-substitution + model cost change + candidate landmarks, all active blocks:
//   y_p = V^-1 (g_p - sum W^T y_c),  step = -y,  delta = step * scale,
//   model_cost_change = -sum m (r + m/2),  m = Jc step_c + Jp step_p   (Ceres'
//   TrustRegionMinimizer::ComputeTrustRegionStep on the scaled Jacobian).
// partial[b] = { sum m (r + m/2), |delta_p|^2, |x_p|^2 }
// WITH_COST: the block's share of the cost at the candidate point as well, into the second bank of partial[] (what
//   k_cost(which = 1) writes there, bit for bit: the same lanes hold the same observations and the sums take the same
//   order).  The candidate poses must exist already (the solver launch or k_pose_update); the candidate landmark of an
//   observation is formed in this very block.  Saves a pass over the observations and a launch per iteration.
// ------------------------------------------------------------------------------------------
template <bool WITH_COST>
__global__ __launch_bounds__(kBlock) void k_backsub(DevP P, LmParams lm, double* cost_partial) {
  __shared__ double sh[kBlock * 3];
  __shared__ double red[5 * 4];
  const int b = blockIdx.x;
  const BlockDesc d = P.blocks[b];
  const int w = d.window;
  const WinState* st = P.ws + w;
  if (!st->active || st->solve_failed) return;
  if (d.kind == 0 && d.pad) return;  // a piece of a long landmark: k_long_point's
  const WinDesc wd = P.wd[w];
  const int sel = st->sel;
  const int t = threadIdx.x;
  const bool live = t < d.n_obs;
  double r[2] = {0, 0}, Jc[12], Jp[6];
  double mc[2] = {0, 0};  // Jc * step_c
  int pp = -1, c = 0, s0 = 0;
  double2 uv = make_double2(0.0, 0.0);
  bool has_c = false;
  if (live) {
    const long long o = d.obs_begin + t;
    pp = P.obs_dp[o];
    c = P.obs_pose[o];
    const int g = P.pose_gslot[c];
    has_c = g >= 0;
    const double* pose = P.poses[sel] + 7ll * c;
    const double* p = P.points[sel] + 3ll * pp;
    uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
    const bool pm = d.kind == 0;
    eval_obs<true>(wd.K, lm, pose, p, uv.x, uv.y, has_c ? P.scale_c + 6ll * g : nullptr,
                   pm ? P.scale_p + 3ll * pp : nullptr, r, Jc, Jp);
    if (has_c) {
      const double* y = P.yc + 6ll * g;
#pragma unroll
      for (int i = 0; i < 6; ++i) { mc[0] -= Jc[i] * y[i]; mc[1] -= Jc[6 + i] * y[i]; }
    }
  }
  double acc[WITH_COST ? 5 : 3];
#pragma unroll
  for (int i = 0; i < (WITH_COST ? 5 : 3); ++i) acc[i] = 0.0;
  double xn[3] = {0.0, 0.0, 0.0};  // the observation's landmark at the candidate point
  if (d.kind == 0) {
    if (live) {
      // W^T y_c = Jp^T (Jc y_c) = -Jp^T mc
#pragma unroll
      for (int i = 0; i < 3; ++i) sh[3 * t + i] = -(Jp[i] * mc[0] + Jp[3 + i] * mc[1]);
    }
    __syncthreads();
    if (live) {
      s0 = (int)(P.pp_obs_start[pp] - d.obs_begin);
      const int s1 = (int)(P.pp_obs_start[pp + 1] - d.obs_begin);
      double tt[3] = {P.gp[3ll * pp], P.gp[3ll * pp + 1], P.gp[3ll * pp + 2]};
      for (int j = s0; j < s1; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) tt[i] -= sh[3 * j + i];
      }
      const double* Vi = P.Vinv + 6ll * pp;
      double sp[3];  // step_p = -y_p
      sp[0] = -(Vi[0] * tt[0] + Vi[1] * tt[1] + Vi[2] * tt[2]);
      sp[1] = -(Vi[1] * tt[0] + Vi[3] * tt[1] + Vi[4] * tt[2]);
      sp[2] = -(Vi[2] * tt[0] + Vi[4] * tt[1] + Vi[5] * tt[2]);
      const double m0 = mc[0] + Jp[0] * sp[0] + Jp[1] * sp[1] + Jp[2] * sp[2];
      const double m1 = mc[1] + Jp[3] * sp[0] + Jp[4] * sp[1] + Jp[5] * sp[2];
      acc[0] = m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
      if (t == s0) {
        const double* x = P.points[sel] + 3ll * pp;
        double* xc = P.points[sel ^ 1] + 3ll * pp;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double dl = sp[i] * P.scale_p[3ll * pp + i];
          xn[i] = x[i] + dl;
          xc[i] = xn[i];
          const double df = x[i] - xn[i];
          acc[1] += df * df;
          acc[2] += x[i] * x[i];
        }
      }
    }
    if (WITH_COST) {
      // the landmark's first lane hands the candidate to the others through the words its sums came from
      __syncthreads();
      if (live && t == s0) { sh[3 * t] = xn[0]; sh[3 * t + 1] = xn[1]; sh[3 * t + 2] = xn[2]; }
      __syncthreads();
      if (live) { xn[0] = sh[3 * s0]; xn[1] = sh[3 * s0 + 1]; xn[2] = sh[3 * s0 + 2]; }
    }
  } else if (live) {
    acc[0] = mc[0] * (r[0] + mc[0] / 2.0) + mc[1] * (r[1] + mc[1] / 2.0);
    if (WITH_COST) {
      const double* x = P.points[sel ^ 1] + 3ll * pp;  // a constant landmark: the same in both banks
      xn[0] = x[0]; xn[1] = x[1]; xn[2] = x[2];
    }
  }
  if (WITH_COST && live) {
    double pose[7];
    const double* ps = P.poses[sel ^ 1] + 7ll * c;
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = ps[i];
    double rc[2];
    residual_only(wd.K, pose, xn, uv.x, uv.y, rc);
    const double s = rc[0] * rc[0] + rc[1] * rc[1];
    double rho0 = s;
    if (lm.loss == HIPBA_LOSS_HUBER) {
      double f;
      huber_correct(lm.loss_scale, s, &rho0, &f);
    }
    acc[3] = 0.5 * rho0;
    acc[4] = isfinite(rho0) ? 0.0 : 1.0;
  }
  block_sum<(WITH_COST ? 5 : 3)>(acc, red);
  if (t == 0) {
    double* pr = P.partial + (long long)b * kPartialStride;
    pr[0] = acc[0]; pr[1] = acc[1]; pr[2] = acc[2];
    if (WITH_COST) {
      cost_partial[(long long)b * kPartialStride + 0] = acc[3];
      cost_partial[(long long)b * kPartialStride + 1] = acc[4];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Landmarks with more observations than a block holds (LongDesc): one workgroup per landmark walks all of its
// observations, 256 at a time, and does for it what the block kernels do for the others.
//   PHASE 0  Jacobi column scaling (k_point_colnorm)
//   PHASE 1  V, g_p, V^-1, the landmark's record, robust weights (k_point_schur<2>: gathered elimination only,
//            the pair products themselves come from the global pair list)
//   PHASE 2  back-substitution, candidate landmark, model-cost-change terms (k_backsub)
// Sums run over the lanes' strided partial sums in a fixed order.
// ------------------------------------------------------------------------------------------
template <int PHASE>
__global__ __launch_bounds__(kBlock) void k_long_point(DevP P, LmParams lm) {
  __shared__ double red[9 * 4];
  __shared__ double bc[4];
  const LongDesc L = P.longs[blockIdx.x];
  const BlockDesc d0 = P.blocks[L.blk_begin];
  const int w = d0.window;
  const WinState* st = P.ws + w;
  if (!st->active) return;
  if (PHASE == 2 && st->solve_failed) return;
  const WinDesc wd = P.wd[w];
  const int sel = st->sel;
  const int t = threadIdx.x, pp = L.pp;
  const double p[3] = {P.points[sel][3ll * pp], P.points[sel][3ll * pp + 1], P.points[sel][3ll * pp + 2]};
  if (PHASE == 0) {
    double n[3] = {0.0, 0.0, 0.0};
    for (int k = t; k < L.n_obs; k += kBlock) {
      const long long o = L.obs_begin + k;
      const double2 uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
      double r[2], Jc[12], Jp[6];
      eval_obs<true>(wd.K, lm, P.poses[sel] + 7ll * P.obs_pose[o], p, uv.x, uv.y, nullptr, nullptr, r, Jc, Jp);
#pragma unroll
      for (int i = 0; i < 3; ++i) n[i] += Jp[i] * Jp[i] + Jp[3 + i] * Jp[3 + i];
    }
    block_sum<3>(n, red);
    if (t == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) P.scale_p[3ll * pp + i] = lm.jacobi_scaling ? 1.0 / (1.0 + sqrt(n[i])) : 1.0;
    }
    return;
  }
  const double sp[3] = {P.scale_p[3ll * pp], P.scale_p[3ll * pp + 1], P.scale_p[3ll * pp + 2]};
  if (PHASE == 1) {
    double a[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = 0.0;
    for (int k = t; k < L.n_obs; k += kBlock) {
      const long long o = L.obs_begin + k;
      const int g = P.pose_gslot[P.obs_pose[o]];
      const double2 uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
      double r[2], Jc[12], Jp[6], lf = 1.0;
      eval_obs<true>(wd.K, lm, P.poses[sel] + 7ll * P.obs_pose[o], p, uv.x, uv.y, g < 0 ? nullptr : P.scale_c + 6ll * g, sp,
                     r, Jc, Jp, &lf);
      if (lm.loss != HIPBA_LOSS_NONE && P.fw) P.fw[o] = lf;
      a[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
      a[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
      a[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
      a[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
      a[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
      a[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
      a[6] += Jp[0] * r[0] + Jp[3] * r[1];
      a[7] += Jp[1] * r[0] + Jp[4] * r[1];
      a[8] += Jp[2] * r[0] + Jp[5] * r[1];
    }
    block_sum<9>(a, red);
    if (t == 0) {
      const double inv_radius = 1.0 / st->radius;
      double V[6] = {a[0], a[1], a[2], a[3], a[4], a[5]};
      V[0] += fmin(fmax(V[0], lm.min_lm_diagonal), lm.max_lm_diagonal) * inv_radius;
      V[3] += fmin(fmax(V[3], lm.min_lm_diagonal), lm.max_lm_diagonal) * inv_radius;
      V[5] += fmin(fmax(V[5], lm.min_lm_diagonal), lm.max_lm_diagonal) * inv_radius;
      double Vi[6];
      inv3_sym(V, Vi);
      double gmax = 0.0, g2 = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) P.Vinv[6ll * pp + i] = Vi[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        P.gp[3ll * pp + i] = a[6 + i];
        const double gu = a[6 + i] / sp[i];
        gmax = fmax(gmax, fabs(gu));
        g2 += gu * gu;
      }
      // the landmark's share of the gradient norms is reported under its first block
      for (int b = 0; b < L.n_blk; ++b) {
        P.gpart[2ll * (L.blk_begin + b)] = b == 0 ? gmax : 0.0;
        P.gpart[2ll * (L.blk_begin + b) + 1] = b == 0 ? g2 : 0.0;
      }
      if (P.prec) {
        double* rec = P.prec + 16ll * pp;
#pragma unroll
        for (int i = 0; i < 3; ++i) { rec[i] = p[i]; rec[3 + i] = sp[i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) rec[6 + i] = Vi[i];
        rec[12] = sp[0] * (Vi[0] * a[6] + Vi[1] * a[7] + Vi[2] * a[8]);
        rec[13] = sp[1] * (Vi[1] * a[6] + Vi[3] * a[7] + Vi[4] * a[8]);
        rec[14] = sp[2] * (Vi[2] * a[6] + Vi[4] * a[7] + Vi[5] * a[8]);
        rec[15] = 0.0;
      }
    }
    return;
  }
  // PHASE 2: W^T y_c summed over the landmark, then the per-observation model terms with the landmark's step
  double wy[3] = {0.0, 0.0, 0.0};
  for (int k = t; k < L.n_obs; k += kBlock) {
    const long long o = L.obs_begin + k;
    const int g = P.pose_gslot[P.obs_pose[o]];
    if (g < 0) continue;
    const double2 uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
    double r[2], Jc[12], Jp[6];
    eval_obs<true>(wd.K, lm, P.poses[sel] + 7ll * P.obs_pose[o], p, uv.x, uv.y, P.scale_c + 6ll * g, sp, r, Jc, Jp);
    double mc0 = 0.0, mc1 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { mc0 -= Jc[i] * P.yc[6ll * g + i]; mc1 -= Jc[6 + i] * P.yc[6ll * g + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) wy[i] += -(Jp[i] * mc0 + Jp[3 + i] * mc1);
  }
  block_sum<3>(wy, red);
  if (t == 0) { bc[0] = wy[0]; bc[1] = wy[1]; bc[2] = wy[2]; }
  __syncthreads();
  const double tt[3] = {P.gp[3ll * pp] - bc[0], P.gp[3ll * pp + 1] - bc[1], P.gp[3ll * pp + 2] - bc[2]};
  const double* Vi = P.Vinv + 6ll * pp;
  double stp[3];  // step_p = -y_p
  stp[0] = -(Vi[0] * tt[0] + Vi[1] * tt[1] + Vi[2] * tt[2]);
  stp[1] = -(Vi[1] * tt[0] + Vi[3] * tt[1] + Vi[4] * tt[2]);
  stp[2] = -(Vi[2] * tt[0] + Vi[4] * tt[1] + Vi[5] * tt[2]);
  double acc[3] = {0.0, 0.0, 0.0};
  for (int k = t; k < L.n_obs; k += kBlock) {
    const long long o = L.obs_begin + k;
    const int g = P.pose_gslot[P.obs_pose[o]];
    const double2 uv = reinterpret_cast<const double2*>(P.obs_uv)[o];
    double r[2], Jc[12], Jp[6];
    eval_obs<true>(wd.K, lm, P.poses[sel] + 7ll * P.obs_pose[o], p, uv.x, uv.y, g < 0 ? nullptr : P.scale_c + 6ll * g, sp, r,
                   Jc, Jp);
    double m0 = Jp[0] * stp[0] + Jp[1] * stp[1] + Jp[2] * stp[2];
    double m1 = Jp[3] * stp[0] + Jp[4] * stp[1] + Jp[5] * stp[2];
    if (g >= 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) { m0 -= Jc[i] * P.yc[6ll * g + i]; m1 -= Jc[6 + i] * P.yc[6ll * g + i]; }
    }
    acc[0] += m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
  }
  if (t == 0) {
    double* xc = P.points[sel ^ 1] + 3ll * pp;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double xn = p[i] + stp[i] * sp[i];
      xc[i] = xn;
      const double df = p[i] - xn;
      acc[1] += df * df;
      acc[2] += p[i] * p[i];
    }
  }
  block_sum<3>(acc, red);
  if (t == 0) {
    for (int b = 0; b < L.n_blk; ++b) {
      double* pr = P.partial + (long long)(L.blk_begin + b) * kPartialStride;
      pr[0] = b == 0 ? acc[0] : 0.0; pr[1] = b == 0 ? acc[1] : 0.0; pr[2] = b == 0 ? acc[2] : 0.0;
    }
  }
}

// candidate poses T * exp(delta_c) (Sophus::Manifold<SE3>::Plus) and their norm terms,
// one workgroup per window
// candidate poses x+ = x * exp(-y_c o scale_c) of window w and the pose part of |x+ - x|^2, |x|^2, by the calling
// workgroup (any size that is a multiple of 64; red: two words per wave)
__device__ __forceinline__ void pose_update_window(const DevP& P, int w, double* red) {
  WinState* st = P.ws + w;
  const WinDesc wd = P.wd[w];
  const int sel = st->sel;
  double acc[2] = {0.0, 0.0};
  for (int c = wd.pose_begin + threadIdx.x; c < wd.pose_end; c += blockDim.x) {
    const int g = P.pose_gslot[c];
    if (g < 0) continue;
    double x[7], dl[6], xn[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = P.poses[sel][7ll * c + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) dl[i] = -P.yc[6ll * g + i] * P.scale_c[6ll * g + i];
    se3_plus(x, dl, xn);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      P.poses[sel ^ 1][7ll * c + i] = xn[i];
      const double df = x[i] - xn[i];
      acc[0] += df * df;
      acc[1] += x[i] * x[i];
    }
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) { st->pose_step2 = acc[0]; st->pose_x2 = acc[1]; }
}

__global__ __launch_bounds__(kWideBlock) void k_pose_update(DevP P) {
  __shared__ double red[2 * (kWideBlock / 64)];
  const WinState* st = P.ws + blockIdx.x;
  if (!st->active || st->solve_failed) return;
  pose_update_window(P, blockIdx.x, red);
}

// ------------------------------------------------------------------------------------------
// per-window reductions + the trust-region controller (one workgroup per window)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void log_iteration(const DevP& P, const WinDesc& wd, const hipba_iteration& it) {
  if (P.log && it.iteration < P.log_cap) P.log[wd.log_base + it.iteration] = it;
}

// Per-window sums can be completed across ranks (observation-sharded solves): every
// window kernel below runs in one of three modes
//   kFused   reduce the per-block partials and decide, in one launch (single GPU)
//   kReduce  reduce only, leaving the local sums in P.wsum for the caller's all-reduce
//   kDecide  decide from the (summed) values in P.wsum
constexpr int kWsum = 8;  // doubles of P.wsum per window

// iteration 0: initial cost, fixed cost, state reset
// zero_S: the window's packed reduced matrix and right-hand side are zeroed for the first pass (small windows: no
// memset queued by the host)
__global__ __launch_bounds__(kBlock) void k_window_init(DevP P, LmParams lm, int mode, int zero_S) {
  __shared__ double red[3 * 4];
  const int w = blockIdx.x;
  const WinDesc wd = P.wd[w];
  WinState* st = P.ws + w;
  if (zero_S) {
    double2* Sw = reinterpret_cast<double2*>(P.S + wd.S_off);
    const long long n2 = 18ll * ((long long)wd.ncam * (wd.ncam + 1) / 2);
    for (long long i = threadIdx.x; i < n2; i += kBlock) Sw[i] = make_double2(0.0, 0.0);
    double2* bw = reinterpret_cast<double2*>(P.b + 6ll * wd.cam_base);
    for (int i = threadIdx.x; i < 3 * wd.ncam; i += kBlock) bw[i] = make_double2(0.0, 0.0);
  }
  double acc[3] = {0.0, 0.0, 0.0};
  if (mode != kDecide) {
    for (int b = wd.pm_blk_begin + threadIdx.x; b < wd.pm_blk_end; b += kBlock) {
      acc[0] += P.partial[(long long)b * kPartialStride];
      acc[2] += P.partial[(long long)b * kPartialStride + 1];
    }
    for (int b = wd.vc_blk_begin + threadIdx.x; b < wd.vc_blk_end; b += kBlock) {
      acc[0] += P.partial[(long long)b * kPartialStride];
      acc[2] += P.partial[(long long)b * kPartialStride + 1];
    }
    for (int b = wd.cc_blk_begin + threadIdx.x; b < wd.cc_blk_end; b += kBlock) {
      acc[1] += P.partial[(long long)b * kPartialStride];
    }
    block_sum<3>(acc, red);
  }
  if (threadIdx.x != 0) return;
  if (mode == kReduce) {
    for (int i = 0; i < 3; ++i) P.wsum[kWsum * w + i] = acc[i];
    return;
  }
  if (mode == kDecide) {
    for (int i = 0; i < 3; ++i) acc[i] = P.wsum[kWsum * w + i];
  }
  const int sel = st->sel;
  WinState s;
  s.sel = sel;
  s.termination = HIPBA_NO_CONVERGENCE;
  s.iteration = 0;
  s.n_invalid = 0;
  s.any_success = 0;
  s.num_successful = 1;  // Ceres counts iteration 0 as a successful step
  s.num_unsuccessful = 0;
  s.solve_failed = 0;
  s.last_success_log = 0;
  s.lin_iters = 0;
  s.num_linear_solves = 0;
  s.handoff_timeouts = 0;
  s.pad_ = 0;
  s.radius = lm.initial_radius;
  s.decrease_factor = 2.0;
  s.x_cost = acc[0];
  s.fixed_cost = acc[1];
  s.initial_cost = acc[0] + acc[1];
  s.gmax = 0.0; s.gnorm = 0.0;
  s.pose_step2 = s.pose_x2 = s.pose_gmax = s.pose_g2 = 0.0;
  s.active = 1;
  const bool nothing = (wd.ncam == 0) && (wd.pp_end == wd.pp_begin) && !P.sharded;
  if (nothing) { s.active = 0; s.termination = HIPBA_CONVERGENCE; }
  if (acc[2] > 0.0 || !isfinite(acc[0])) { s.active = 0; s.termination = HIPBA_FAILURE; }
  hipba_iteration it;
  it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.linear_solver_iterations = 0;
  it.cost = s.initial_cost; it.cost_change = 0.0; it.gradient_max_norm = 0.0; it.gradient_norm = 0.0;
  it.step_norm = 0.0; it.relative_decrease = 0.0; it.trust_region_radius = s.radius;
  log_iteration(P, wd, it);
  if (s.active && lm.max_num_iterations <= 0) { s.active = 0; s.termination = HIPBA_NO_CONVERGENCE; }
  *st = s;
}

// gradient norms of the fresh linearisation + gradient-tolerance test (Ceres checks it when an
// iteration that moved x is finalised; the norms only exist once the blocks have been rebuilt).
// wsum slots: [0] landmark max (max-reduced across ranks), [1] landmark sum of squares.
// In two halves, so that a caller can put loads of its own between the words this test reads and the reductions
// (four barriers) that follow: window_gradient_partials returns the thread's share, window_gradient_finish does the rest.
struct GradientPartials { double gm, g2, pm, p2; };
__device__ __forceinline__ GradientPartials window_gradient_partials(const DevP& P, const WinDesc& wd, int mode) {
  GradientPartials v{0.0, 0.0, 0.0, 0.0};
  if (mode != kDecide) {
#pragma unroll 8
    for (int b = wd.pm_blk_begin + threadIdx.x; b < wd.pm_blk_end; b += blockDim.x) {
      const double2 gp = reinterpret_cast<const double2*>(P.gpart)[b];  // (max, sum of squares) of block b
      v.gm = fmax(v.gm, gp.x);
      v.g2 += gp.y;
    }
  }
  if (mode != kReduce) {
    for (int g = wd.cam_base + threadIdx.x; g < wd.cam_base + wd.ncam; g += blockDim.x) {
      v.pm = fmax(v.pm, P.yc[6ll * g]);
      v.p2 += P.yc[6ll * g + 1];
    }
  }
  return v;
}
__device__ __forceinline__ void window_gradient_finish(const DevP& P, const LmParams& lm, const WinDesc& wd, WinState* st,
                                                       double* red, int mode, GradientPartials v) {
  const int w = (int)(st - P.ws);
  double gm = v.gm;
  double g2[1] = {v.g2};
  if (mode != kDecide) {
    gm = block_max(gm, red);
    block_sum<1>(g2, red);
    if (mode == kReduce) {
      if (threadIdx.x == 0) { P.wsum[kWsum * w] = gm; P.wsum[kWsum * w + 1] = g2[0]; }
      return;
    }
  }
  double pm = v.pm;
  double p2[1] = {v.p2};
  pm = block_max(pm, red);
  block_sum<1>(p2, red);
  if (threadIdx.x == 0) {
    if (mode == kDecide) { gm = P.wsum[kWsum * w]; g2[0] = P.wsum[kWsum * w + 1]; }
    gm = fmax(gm, pm);
    st->gmax = gm;
    st->gnorm = sqrt(g2[0] + p2[0]);
    if (P.log && st->last_success_log >= 0 && st->last_success_log < P.log_cap) {
      hipba_iteration* e = P.log + wd.log_base + st->last_success_log;
      e->gradient_max_norm = gm;
      e->gradient_norm = st->gnorm;
    }
    if (st->last_success_log == st->iteration && gm <= lm.gradient_tolerance) {
      st->active = 0;
      st->termination = HIPBA_CONVERGENCE;
    }
  }
  __syncthreads();
}
__device__ void window_gradient_check(const DevP& P, const LmParams& lm, const WinDesc& wd, WinState* st, double* red,
                                      int mode) {
  window_gradient_finish(P, lm, wd, st, red, mode, window_gradient_partials(P, wd, mode));
}

// end of an iteration: reduce the per-block partials and run Ceres' acceptance logic
// (TrustRegionMinimizer::Minimize, LevenbergMarquardtStrategy::StepAccepted/Rejected/IsInvalid)
__device__ __forceinline__ void window_control_body(const DevP& P, const LmParams& lm, int mode) {
  __shared__ double red[5 * (kWideBlock / 64)];
  const int w = blockIdx.x;
  WinState* st = P.ws + w;
  if (!st->active) return;
  const WinDesc wd = P.wd[w];
  const bool failed = st->solve_failed != 0;
  // acc: 0 model term, 1 point step^2, 2 point x^2, 3 candidate cost, 4 bad flag
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  // k_backsub writes the first bank of P.partial, the candidate cost pass the second
  if (!failed && mode != kDecide) {
    const long long nb = (long long)P.n_blk_pm + P.n_blk_vc + P.n_blk_cc;
    const double* p2 = P.partial + nb * kPartialStride;  // second bank: candidate cost
    // (unrolled: a large window has tens of thousands of blocks, and the loads of one trip do not depend on the last)
#pragma unroll 8
    for (int b = wd.pm_blk_begin + threadIdx.x; b < wd.pm_blk_end; b += blockDim.x) {
      const double2* pr = reinterpret_cast<const double2*>(P.partial + (long long)b * kPartialStride);
      const double2 pa = pr[0], pb = pr[1];
      const double2 pc = *reinterpret_cast<const double2*>(p2 + (long long)b * kPartialStride);
      acc[0] += pa.x; acc[1] += pa.y; acc[2] += pb.x;
      acc[3] += pc.x; acc[4] += pc.y;
    }
    for (int b = wd.vc_blk_begin + threadIdx.x; b < wd.vc_blk_end; b += blockDim.x) {
      const double* pr = P.partial + (long long)b * kPartialStride;
      acc[0] += pr[0];
      acc[3] += p2[(long long)b * kPartialStride]; acc[4] += p2[(long long)b * kPartialStride + 1];
    }
  }
  if (mode != kDecide) block_sum<5>(acc, red);
  if (threadIdx.x != 0) return;
  if (mode == kReduce) {
    for (int i = 0; i < 5; ++i) P.wsum[kWsum * w + i] = acc[i];
    return;
  }
  if (mode == kDecide) {
    for (int i = 0; i < 5; ++i) acc[i] = P.wsum[kWsum * w + i];
  }

  WinState s = *st;
  s.num_linear_solves++;
  const int iter = s.iteration + 1;
  hipba_iteration it;
  it.iteration = iter; it.linear_solver_iterations = s.lin_iters;
  it.gradient_max_norm = s.gmax; it.gradient_norm = s.gnorm;
  const double model_cost_change = -acc[0];
  const bool valid = !failed && (model_cost_change > 0.0);
  bool logged = true;
  if (!valid) {
    // HandleInvalidStep
    s.n_invalid++;
    if (s.n_invalid >= lm.max_invalid) {
      s.active = 0; s.termination = HIPBA_FAILURE;
      *st = s;
      return;
    }
    s.radius *= 0.5;
    it.step_is_valid = 0; it.step_is_successful = 0;
    it.cost = s.x_cost + s.fixed_cost; it.cost_change = 0.0; it.step_norm = 0.0; it.relative_decrease = 0.0;
    s.num_unsuccessful++;
  } else {
    s.n_invalid = 0;
    it.step_is_valid = 1;
    const double step_norm = sqrt(acc[1] + s.pose_step2);
    const double x_norm = sqrt(acc[2] + s.pose_x2);
    it.step_norm = step_norm;
    double cand = acc[3];
    if (acc[4] > 0.0 || !isfinite(cand)) cand = DBL_MAX;
    // TrustRegionMinimizer::ParameterToleranceReached: on every valid step (the option delays it until a step
    // has been accepted, which upstream Ceres does not do)
    if ((s.any_success || !lm.param_tol_needs_success) &&
        step_norm <= lm.parameter_tolerance * (x_norm + lm.parameter_tolerance)) {
      s.active = 0; s.termination = HIPBA_CONVERGENCE;
      *st = s;
      return;
    }
    // FunctionToleranceReached
    it.cost_change = s.x_cost - cand;
    if (fabs(it.cost_change) <= lm.function_tolerance * s.x_cost) {
      s.active = 0; s.termination = HIPBA_CONVERGENCE;
      *st = s;
      return;
    }
    it.relative_decrease = (cand >= DBL_MAX) ? -DBL_MAX : it.cost_change / model_cost_change;
    if (it.relative_decrease > lm.min_relative_decrease) {
      // HandleSuccessfulStep + StepAccepted
      s.sel ^= 1;
      s.x_cost = cand;
      s.any_success = 1;
      const double q = 2.0 * it.relative_decrease - 1.0;
      s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
      s.radius = fmin(lm.max_radius, s.radius);
      s.decrease_factor = 2.0;
      it.step_is_successful = 1;
      it.cost = s.x_cost + s.fixed_cost;
      s.num_successful++;
      s.last_success_log = iter;
    } else {
      // StepRejected
      s.radius = s.radius / s.decrease_factor;
      s.decrease_factor *= 2.0;
      it.step_is_successful = 0;
      it.cost = cand + s.fixed_cost;
      s.num_unsuccessful++;
    }
  }
  it.trust_region_radius = s.radius;
  if (logged) log_iteration(P, wd, it);
  s.iteration = iter;
  s.solve_failed = 0;
  // FinalizeIterationAndCheckIfMinimizerCanContinue (the gradient test runs with the next linearisation)
  if (iter >= lm.max_num_iterations) { s.active = 0; s.termination = HIPBA_NO_CONVERGENCE; }
  else if (s.radius <= lm.min_radius) { s.active = 0; s.termination = HIPBA_CONVERGENCE; }
  *st = s;
}

// count_out (one-window solves): the "windows still active" word the host polls, written here instead of by a
// launch of k_count_active
// zero_S: a window that goes on to another iteration has its packed reduced matrix zeroed here, so the host does
// not queue a memset (a launch of its own on the iteration's critical path) before every pass
// count_tag: OR-ed into that word (the pass number shifted left by one when the host polls the word itself instead of
// an event behind this launch: it then needs to tell this pass's word from an older one); the store is pushed out to
// the host at once
__global__ __launch_bounds__(kWideBlock) void k_window_control(DevP P, LmParams lm, int mode, int* count_out, int zero_S,
                                                             int count_tag) {
  window_control_body(P, lm, mode);
  if (count_out && threadIdx.x == 0) {
    __hip_atomic_store(count_out, count_tag | (P.ws[blockIdx.x].active ? 1 : 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!zero_S) return;
  __syncthreads();  // the decision thread 0 has just written
  const WinState* st = P.ws + blockIdx.x;
  if (!*reinterpret_cast<const volatile int*>(&st->active)) return;
  const WinDesc wd = P.wd[blockIdx.x];
  double2* Sw = reinterpret_cast<double2*>(P.S + wd.S_off);
  const long long n2 = 18ll * ((long long)wd.ncam * (wd.ncam + 1) / 2);
  for (long long i = threadIdx.x; i < n2; i += blockDim.x) Sw[i] = make_double2(0.0, 0.0);
  // (and its right-hand side: k_small_linearize adds to both)
  double2* bw = reinterpret_cast<double2*>(P.b + 6ll * wd.cam_base);
  for (int i = threadIdx.x; i < 3 * wd.ncam; i += blockDim.x) bw[i] = make_double2(0.0, 0.0);
}

// ------------------------------------------------------------------------------------------
// reduced solve for small windows: gradient check, then Cholesky of the packed S in LDS
// (one workgroup per window, n = 6 * ncam <= kSmallMaxN)
// ------------------------------------------------------------------------------------------

constexpr int kSolveThreads = 512;

// index of entry (a, c), a <= c, in a row-major packed upper triangle of order 6
#define HIPBA_UIDX(a, c) ((a) * 6 - (a) * ((a) - 1) / 2 + ((c) - (a)))

// Cholesky factor R (upper, R^T R = D) of the 6x6 block at D, replaced in place by R^-1 (upper
// triangle only; the words below the diagonal are left alone).  One lane.  Returns false on a
// non-positive or non-finite pivot.
__device__ __forceinline__ bool factor_invert_6x6(double* D) {
  double U[21], dinv[6];
  {
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) U[q++] = D[6 * a + c];
    }
  }
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double dg = U[HIPBA_UIDX(a, a)];
#pragma unroll
    for (int mm = 0; mm < a; ++mm) dg -= U[HIPBA_UIDX(mm, a)] * U[HIPBA_UIDX(mm, a)];
    if (!(dg > 0.0) || !isfinite(dg)) ok = false;
    // 1/sqrt: hardware estimate + two Newton steps (full double precision, no division)
    double inv = __builtin_amdgcn_rsq(dg);
    inv = inv * (1.5 - 0.5 * dg * inv * inv);
    inv = inv * (1.5 - 0.5 * dg * inv * inv);
    dinv[a] = inv;
#pragma unroll
    for (int c = a + 1; c < 6; ++c) {
      double v = U[HIPBA_UIDX(a, c)];
#pragma unroll
      for (int mm = 0; mm < a; ++mm) v -= U[HIPBA_UIDX(mm, a)] * U[HIPBA_UIDX(mm, c)];
      U[HIPBA_UIDX(a, c)] = v * inv;
    }
  }
  // X = R^-1: X(a,a) = 1/R(a,a), X(a,c) = -(sum_{a<=m<c} X(a,m) R(m,c)) / R(c,c)
  double X[21];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    X[HIPBA_UIDX(a, a)] = dinv[a];
#pragma unroll
    for (int c = a + 1; c < 6; ++c) {
      double v = 0.0;
#pragma unroll
      for (int mm = a; mm < c; ++mm) v += X[HIPBA_UIDX(a, mm)] * U[HIPBA_UIDX(mm, c)];
      X[HIPBA_UIDX(a, c)] = -v * dinv[c];
    }
  }
  {
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) D[6 * a + c] = X[q++];
    }
  }
  return ok;
}

// The same for a whole wave (all 64 lanes call it): lane r < 6 carries row r of the lower factor L = R^T, the pivots
// and multipliers travel as wave-uniform scalars, and lanes 16-21 start from the columns of the identity and execute
// the very same scalings and multiply-subtracts, which leaves column r of L^-1 (= row r of R^-1) in lane 16 + r:
// the inverse costs no instruction of its own and the dependency chain is one column per step, not one term.
__device__ __forceinline__ bool factor_invert_6x6_wave(double* D, int lane) {
  const int r = lane & 15;
  const bool inverse_half = (lane & 16) != 0;
  const bool row_lane = r < 6;
  double x[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    // row r of L starts as row r of the symmetric block, read from the upper triangle: D(r, c) = D(c, r), c <= r
    const double a = (row_lane && c <= r) ? D[6 * c + r] : 0.0;
    x[c] = inverse_half ? (c == r ? 1.0 : 0.0) : a;
  }
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x[c]), c), hi = __builtin_amdgcn_readlane(__double2hiint(x[c]), c);
    const double d = __hiloint2double(hi, lo);
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    x[c] *= inv;
#pragma unroll
    for (int kc = c + 1; kc < 6; ++kc) {
      const int mlo = __builtin_amdgcn_readlane(__double2loint(x[c]), kc), mhi = __builtin_amdgcn_readlane(__double2hiint(x[c]), kc);
      x[kc] -= x[c] * __hiloint2double(mhi, mlo);  // L(kc, c)
    }
  }
  if (ok && inverse_half && row_lane && lane < 32) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i >= r) D[6 * r + i] = x[i];  // R^-1(r, i) = L^-1(i, r)
  }
  return ok;
}

// fuse_pose_update: the workgroup goes on to form the window's candidate poses (k_pose_update's work) from the
// solution it has just written: one launch less per iteration.
__global__ __launch_bounds__(kSolveThreads) void k_window_solve_small(DevP P, LmParams lm, int fuse_pose_update) {
  // all LDS comes from the dynamic region so that its base stays 16-byte aligned
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* red = smem;                                  // [8]
  int& fail = *reinterpret_cast<int*>(smem + 8);
  const int w = blockIdx.x;
  WinState* st = P.ws + w;
  if (!st->active) return;
  HIPBA_TICK(0);
  const WinDesc wd = P.wd[w];
  window_gradient_check(P, lm, wd, st, red, kFused);
  HIPBA_TICK(1);
  if (!st->active) return;
  const int nb = wd.ncam;
  if (nb == 0) return;
  // S = R^T R by 6x6 blocks, in the packed upper block order S already has in HBM (the load is a
  // straight copy).  Blocks are padded to kSmallBlk doubles: an odd stride, so lanes that work on
  // neighbouring blocks hit different LDS banks.  A factored diagonal block is kept as its inverse.
  double* Rm = smem + 16;
  const int t = threadIdx.x;
  const int nblk = nb * (nb + 1) / 2;
  double* zb = Rm + (size_t)nblk * kSmallBlk;          // right-hand side, becomes z = R^-T b
  const double* Sg = P.S + wd.S_off;
  for (int e0 = t; e0 < nblk * 36; e0 += 8 * kSolveThreads) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kSolveThreads;
      v[u] = e < nblk * 36 ? Sg[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kSolveThreads;
      if (e < nblk * 36) {
        const int blk = e / 36;
        Rm[blk * kSmallBlk + (e - blk * 36)] = v[u];
      }
    }
  }
  for (int i = t; i < 6 * nb; i += kSolveThreads) zb[i] = P.b[6ll * wd.cam_base + i];
  if (t == 0) fail = 0;
  __syncthreads();
  if (t == 0 && !factor_invert_6x6(Rm)) fail = 1;
  __syncthreads();
  HIPBA_TICK(2);
  HIPBA_CYCLES(5);

  for (int k = 0; k < nb; ++k) {
    const int m = nb - k - 1;                           // block columns to the right of the diagonal
    const int rowk = (int)tri_index(k, k, nb);          // blocks (k, k..nb-1) are contiguous from here
    // block row k: R(k,j) = R(k,k)^-T S(k,j), one lane per column; one more lane for the right-hand side
    if (t < 6 * m + 1) {
      const double* X = Rm + rowk * kSmallBlk;          // R(k,k)^-1, upper
      const bool rhs = t == 6 * m;
      const int jj = t / 6, c = t - 6 * jj;
      double* col = rhs ? zb + 6 * k : Rm + (rowk + 1 + jj) * kSmallBlk + c;
      const int stride = rhs ? 1 : 6;
      double x[6], y[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) x[a] = col[stride * a];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double v = 0.0;
#pragma unroll
        for (int mm = 0; mm <= a; ++mm) v += X[6 * mm + a] * x[mm];
        y[a] = v;
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) col[stride * a] = y[a];
    }
    __syncthreads();
    // trailing update S(i,j) -= R(k,i)^T R(k,j), k < i <= j.  Lane 0 takes the next diagonal block whole and
    // factors it on the spot (that chain runs beside the other lanes' updates); the other lanes take half
    // blocks (3 rows x 6 columns), pairs enumerated row by row so that a wave's lanes share R(k,i) and walk
    // consecutive blocks; the last m items update the right-hand side.
    const int npair = m * (m + 1) / 2;
    const int nitem = 2 * npair + m;
    if (t < 64) {
      // wave 0: the next diagonal block, the critical path of the step -- issued first.  36 lanes take one entry of
      // the rank-6 update each (the upper triangle is the one kept), then the wave factors and inverts the block
      // with a lane per row (factor_invert_6x6_wave).  The rest of the wave's time is idle on purpose: update items
      // here would run after this chain, not beside it.
      if (m > 0) {
        __builtin_amdgcn_s_setprio(3);
        const double* Ri = Rm + (rowk + 1) * kSmallBlk;
        double* T = Rm + (int)tri_index(k + 1, k + 1, nb) * kSmallBlk;
        const int a = t / 6, c = t - 6 * a;
        if (t < 36 && a <= c) {
          double v = T[6 * a + c];
#pragma unroll
          for (int mm = 0; mm < 6; ++mm) v -= Ri[6 * mm + a] * Ri[6 * mm + c];
          T[6 * a + c] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!factor_invert_6x6_wave(T, t) && t == 0) fail = 1;
        __builtin_amdgcn_s_setprio(0);
      }
    } else if (t >= 64) {
      // (the rest of wave 0 idles: it would run this loop after lane 0's chain, not beside it)
      for (int e = 2 + (t - 64); e < nitem; e += kSolveThreads - 64) {
        if (e < 2 * npair) {
          const int p = e >> 1, h = e & 1;
          // row ii of an m-row upper triangle starts at ii*m - ii(ii-1)/2
          const float fm = 2.0f * m + 1.0f;
          int ii = (int)((fm - sqrtf(fm * fm - 8.0f * p)) * 0.5f);
          ii = max(0, min(ii, m - 1));
          while (ii > 0 && ii * m - ii * (ii - 1) / 2 > p) --ii;
          while ((ii + 1) * m - (ii + 1) * ii / 2 <= p) ++ii;
          const int jj = ii + (p - (ii * m - ii * (ii - 1) / 2));
          const double* Ri = Rm + (rowk + 1 + ii) * kSmallBlk + 3 * h;
          const double* Rj = Rm + (rowk + 1 + jj) * kSmallBlk;
          double* T = Rm + (int)tri_index(k + 1 + ii, k + 1 + jj, nb) * kSmallBlk + 18 * h;
          double acc[18];
#pragma unroll
          for (int i = 0; i < 18; ++i) acc[i] = T[i];
#pragma unroll
          for (int mm = 0; mm < 6; ++mm) {
            double rj[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) rj[c] = Rj[6 * mm + c];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const double ri = Ri[6 * mm + a];
#pragma unroll
              for (int c = 0; c < 6; ++c) acc[6 * a + c] -= ri * rj[c];
            }
          }
#pragma unroll
          for (int i = 0; i < 18; ++i) T[i] = acc[i];
        } else {
          const int ii = e - 2 * npair;
          const double* Ri = Rm + (rowk + 1 + ii) * kSmallBlk;
          double* z = zb + 6 * (k + 1 + ii);
          double acc[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) acc[a] = z[a];
#pragma unroll
          for (int mm = 0; mm < 6; ++mm) {
            const double zk = zb[6 * k + mm];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a] -= Ri[6 * mm + a] * zk;
          }
#pragma unroll
          for (int a = 0; a < 6; ++a) z[a] = acc[a];
        }
      }
    }
    __syncthreads();
  }
  HIPBA_TICK(3);
  HIPBA_CYCLES(6);
  if (fail) {
    if (t == 0) st->solve_failed = 1;
    return;
  }
  // backward sweep R y = z, last block row first: every lane forms y(k) = R(k,k)^-1 z(k) itself, then one
  // lane per row above subtracts that block column's contribution
  bool ok = true;
  for (int k = nb - 1; k >= 0; --k) {
    // only the waves that have a row to update (or the six lanes that store y) form y: the others would
    // just take issue slots from them
    if ((t & ~63) < (6 * k > 6 ? 6 * k : 6)) {
    const double* X = Rm + (int)tri_index(k, k, nb) * kSmallBlk;
    double zk[6], y[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) zk[a] = zb[6 * k + a];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double v = 0.0;
#pragma unroll
      for (int c = a; c < 6; ++c) v += X[6 * a + c] * zk[c];
      y[a] = v;
    }
    if (t < 6) {
      const double v = t == 0 ? y[0] : t == 1 ? y[1] : t == 2 ? y[2] : t == 3 ? y[3] : t == 4 ? y[4] : y[5];
      P.yc[6ll * (wd.cam_base + k) + t] = v;
      ok = ok && isfinite(v);
    }
    for (int e = t; e < 6 * k; e += kSolveThreads) {
      const int i = e / 6, a = e - 6 * i;
      const double* B = Rm + (int)tri_index(i, k, nb) * kSmallBlk + 6 * a;
      double v = zb[e];
#pragma unroll
      for (int c = 0; c < 6; ++c) v -= B[c] * y[c];
      zb[e] = v;
    }
    }
    __syncthreads();
  }
  HIPBA_TICK(4);
  if (t < 64) {
    const unsigned long long okm = __ballot(ok);
    if (t == 0) {
      fail = okm != ~0ull;
      if (fail) st->solve_failed = 1;
    }
  }
  if (!fuse_pose_update) return;
  __syncthreads();  // the solution (global) and the verdict (LDS) are visible to the whole workgroup
  if (fail) return;
  pose_update_window(P, w, Rm);  // the matrix words are free: two sums per wave go there
}
#undef HIPBA_UIDX

// The same solve for the larger of the small windows (a 30-keyframe window: n = 174), on the matrix cores.  S is held
// in LDS as 16x16 tiles (chol_tiles.hpp), padded with an identity to a multiple of 16, and factored right-looking by
// tile columns: wave 0 factors and inverts the diagonal tile (tile_cholesky_inverse), the panel below it is one
// matrix-core product per tile with that inverse, and the trailing tiles are updated by the other seven waves while
// wave 0 already updates and factors the next diagonal tile.  Forward and backward substitution are 16-wide
// matrix-vector products by wave 0 with the tiles' inverses.  The 6x6-block version above walks 29 block columns of
// ~5.6 k cycles each for such a window; this one 11 tile columns of ~7 k.
constexpr int kTileSolveMinN = 48;  // windows from this size on take this kernel (smaller ones: the 6x6-block kernel above)
// table_mode (P.tile_src): 0 the source offsets of the tile image are computed, 1 computed and written to the table,
// 2 read from it (the offsets depend on the number of poses only; computing them is two divisions by six and a
// triangular index per word, 39 times per lane -- 7 us of the kernel's 12 us before the factorisation).
__global__ __launch_bounds__(kSolveThreads) void k_window_solve_tiles(DevP P, LmParams lm, int fuse_pose_update, int table_mode) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* red = smem;                                  // [8]
  int& fail = *reinterpret_cast<int*>(smem + 8);
  const int w = blockIdx.x;
  WinState* st = P.ws + w;
  if (!st->active) return;
  HIPBA_TICK(0);
  const WinDesc wd = P.wd[w];
  const int nb = wd.ncam;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* Sg = P.S + wd.S_off;
  // Packed upper 6x6 blocks (row-major inside a block) -> lower 16x16 tiles in LDS, walked in the order of the
  // DESTINATION: half the workgroup per tile, a lane per word of it, so the LDS stores are linear and all a lane
  // computes is where its word comes from: entry (r, c), r >= c, of S lies in block (c / 6, r / 6) -- at (c % 6, r % 6)
  // of an off-diagonal block, at (r % 6, c % 6) of a diagonal one (whose entries on and below the diagonal are the
  // ones used).  Rows n and beyond (all in the last tile row) are the identity.  Every word -- at most 39 per lane
  // for 192 unknowns -- is requested HERE, before the gradient test, whose reductions and barriers run while they
  // travel.  (Walked in the order of the source, with each word dropped at its place, this took 8 us after a 4 us
  // test: 32 scattered stores per lane behind a data-dependent walk of the block rows.)
  constexpr int kPre = (kSmallMaxN / TB) * (kSmallMaxN / TB + 1) / 4;  // tiles of the largest window / 2
  const int n = 6 * nb, nt = (n + TB - 1) / TB, ntl = nt * (nt + 1) / 2;
  // (the gradient test's own few words first: the memory pipe returns in order, so its reductions need not wait for
  // the matrix)
  const GradientPartials gpv = window_gradient_partials(P, wd, kFused);
  const int half = t >> 8, ent = t & 255, ep = ent & 15, eq = ent >> 4;
  double pre[kPre];
  {
    int ti = 0, tj = half;  // tile (ti, tj) of this half's first tile: linear index `half`
    while (tj > ti) { tj -= ti + 1; ++ti; }
    // source of a word: its offset in the packed matrix, or -1 (one: the diagonal of the padding) or -2 (zero)
    if (table_mode == 2) {
      int src[kPre];
#pragma unroll
      for (int k = 0; k < kPre; ++k) src[k] = 2 * k + half < ntl ? P.tile_src[(2 * k + half) * kTile + ent] : -2;
#pragma unroll
      for (int k = 0; k < kPre; ++k) pre[k] = src[k] >= 0 ? Sg[src[k]] : (src[k] == -1 ? 1.0 : 0.0);
    } else {
#pragma unroll
      for (int k = 0; k < kPre; ++k) {
        int src = -2;
        if (2 * k + half < ntl) {
          const int r = TB * ti + ep, cc = TB * tj + eq;
          if (r >= n) src = r == cc ? -1 : -2;
          else if (r >= cc) {
            const int bi = cc / 6, bj = r / 6;
            const int blk = bi * nb - bi * (bi - 1) / 2 + (bj - bi);
            src = 36 * blk + (bi == bj ? 6 * (r - 6 * bi) + (cc - 6 * bi) : 6 * (cc - 6 * bi) + (r - 6 * bj));
          }
          if (table_mode == 1) P.tile_src[(2 * k + half) * kTile + ent] = src;
        }
        pre[k] = src >= 0 ? Sg[src] : (src == -1 ? 1.0 : 0.0);
        tj += 2;
        while (tj > ti) { tj -= ti + 1; ++ti; }
      }
    }
  }
  HIPBA_TICK(7);
  window_gradient_finish(P, lm, wd, st, red, kFused, gpv);
  HIPBA_TICK(1);
  if (!st->active) return;
  if (nb == 0) return;
  double* Lt = smem + 16;
  double* zb = Lt + (size_t)ntl * kTile;               // right-hand side, then z = L^-1 b, then y
#pragma unroll
  for (int k = 0; k < kPre; ++k)
    if (2 * k + half < ntl) Lt[(2 * k + half) * kTile + ent] = pre[k];
  for (int i = t; i < TB * nt; i += kSolveThreads) zb[i] = i < n ? P.b[6ll * wd.cam_base + i] : 0.0;
  if (t == 0) fail = 0;
  __syncthreads();
  HIPBA_TICK(2);
  HIPBA_CYCLES(5);
  if (wave == 0 && !tile_cholesky_inverse(Lt, nullptr, 0, lane) && lane == 0) fail = 1;
  __syncthreads();
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // sum over the four lanes of a quad, in every lane: two DPP quad permutes per 32-bit half (no LDS round trip)
  auto quad_xor = [](double v, int ctrl_is_xor2) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = ctrl_is_xor2 ? __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false)
                                 : __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false);
    const int phi = ctrl_is_xor2 ? __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false)
                                 : __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
    return __hiloint2double(phi, plo);
  };
  auto groups_sum = [&](double v) {
    v += quad_xor(v, 0);
    v += quad_xor(v, 1);
    return v;
  };
  // Forward substitution inside the factor loop (zb holds b on entry and z = L^-1 b at the end).  In step j -- the
  // panel of column j is final, and so was that of column j - 1 a step earlier:
  //   one wave (forward_chain): b_j -= L(j, j-1) z_(j-1), the last update b_j lacks, then z_j = L(j,j)^-1 b_j;
  //   the updating waves (forward_rows), beside their tiles: b_i -= L(i, j-1) z_(j-1) for the rows i > j, a row per wave.
  // Every block of b has one writer per step, and z_(j-1) is not written in step j.
  // lane (p = lane >> 2, g = lane & 3): (T v)(p), the quad's four partial sums combined in every lane
  auto tile_times = [&](const double* T, const double* v) {
    const int p = lane >> 2, g = lane & 3;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += T[(4 * g + q) * TB + p] * v[4 * g + q];
    return groups_sum(acc);
  };
  auto forward_chain = [&](int j) {
    const int p = lane >> 2, g = lane & 3;
    if (j > 0) {
      const double r = tile_times(Lt + ltile(j, j - 1), zb + TB * (j - 1));
      const double cur = zb[TB * j + p];
      wave_sync();
      if (g == 0) zb[TB * j + p] = cur - r;
      wave_sync();
    }
    const double zj = tile_times(Lt + ltile(j, j), zb + TB * j);   // (the diagonal tile holds its inverse)
    wave_sync();                                                     // every lane has read b_j
    if (g == 0) zb[TB * j + p] = zj;
  };
  auto forward_rows = [&](int j, int first, int step) {
    const int p = lane >> 2, g = lane & 3;
    for (int i = j + 1 + first; i < nt; i += step) {
      const double r = tile_times(Lt + ltile(i, j - 1), zb + TB * (j - 1));
      if (g == 0) zb[TB * i + p] -= r;
    }
  };
#ifdef HIPBA_PHASE_TIMING
  long long tq[4] = {0, 0, 0, 0}, tprev = clock64();
#define HIPBA_LAP(i) do { const long long now_ = clock64(); tq[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define HIPBA_LAP(i)
#endif
  for (int j = 0; j < nt; ++j) {
    if (fail) break;
    // panel below the diagonal tile: L(i,j) = A(i,j) L(j,j)^-T (the result replaces the tile)
    {
      const double* Dinv = Lt + ltile(j, j);
      for (int i = j + 1 + wave; i < nt; i += kSolveThreads / 64) {
        double* T = Lt + ltile(i, j);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        acc = mfma_tile<false>(Dinv, T, acc, lane);
        tile_store_t(T, acc, lane);
      }
    }
    HIPBA_LAP(0);
    __syncthreads();
    HIPBA_LAP(1);
    // Updates, LEFT-looking: in step j the tiles of column j + 1 below the diagonal receive the products of ALL columns
    // 0 .. j at once (one accumulator chain per tile, the tile read and written once), and the diagonal tile j + 2
    // those of columns 0 .. j as well, so that wave 0 finds it lacking only column j + 1 a step later.  Applied
    // right-looking -- column j to the whole trailing block at once -- the same 220 products of a 30-keyframe window came
    // 54 in the first step and one in the last, and the first steps waited for the updating waves; this way a step has
    // at most 30, and the longest job of every step is wave 0's: the next diagonal tile, updated with column j and
    // factored at once, at raised priority.  Wave 4 shares wave 0's SIMD -- and with it the fp64 pipe, which vector and
    // matrix-core instructions both use: it takes no tiles, only the forward substitution's chain.
    if (wave == 0) {
      if (j + 1 < nt) {
        __builtin_amdgcn_s_setprio(3);
        double* C = Lt + ltile(j + 1, j + 1);
        v4d acc = tile_load_t(C, lane);
        acc = mfma_tile<true>(Lt + ltile(j + 1, j), Lt + ltile(j + 1, j), acc, lane);
        tile_store_t(C, acc, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!tile_cholesky_inverse(C, nullptr, 0, lane) && lane == 0) fail = 1;
        __builtin_amdgcn_s_setprio(0);
      }
    } else if (wave == 4) {
      forward_chain(j);
    } else {
      constexpr int kUpd = kSolveThreads / 64 - 2;
      const int widx = wave < 4 ? wave - 1 : wave - 2;  // this wave's place among the updating ones
      if (j > 0) forward_rows(j, widx, kUpd);
      const int n_col = nt - j - 2 > 0 ? nt - j - 2 : 0;       // tiles (j + 2 .. nt - 1, j + 1)
      const int n_jobs = n_col + (j + 2 < nt ? 1 : 0);          // and the diagonal tile (j + 2, j + 2)
      // (the jobs of a step are as long as the step's number: the first one goes to a different wave every step)
      for (int idx = (widx + kUpd - j % kUpd) % kUpd; idx < n_jobs; idx += kUpd) {
        const bool dg = idx == n_col;
        const int r = dg ? j + 2 : j + 2 + idx, c = dg ? j + 2 : j + 1;
        double* C = Lt + ltile(r, c);
        v4d acc = tile_load_t(C, lane);
        // the operands of column k + 1 are requested before the products of column k are issued
        double a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int o = (kk * 4 + (lane >> 4)) * TB + (lane & 15);
          a[kk] = -Lt[ltile(c, 0) + o]; b[kk] = Lt[ltile(r, 0) + o];
        }
        for (int k = 0; k <= j; ++k) {
          double an[4], bn[4];
          const int kn = k < j ? k + 1 : k;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int o = (kk * 4 + (lane >> 4)) * TB + (lane & 15);
            an[kk] = -Lt[ltile(c, kn) + o]; bn[kk] = Lt[ltile(r, kn) + o];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) { a[kk] = an[kk]; b[kk] = bn[kk]; }
        }
        tile_store_t(C, acc, lane);
      }
    }
    HIPBA_LAP(2);
    __syncthreads();
    HIPBA_LAP(3);
  }
#ifdef HIPBA_PHASE_TIMING
  if (t == 0 || t == 64 || t == 256) for (int i = 0; i < 4; ++i) g_ticks[8 * (1 + (t >> 6)) + i] = (unsigned long long)tq[i];
#endif
  HIPBA_TICK(3);
  HIPBA_CYCLES(6);
  if (fail) {
    if (t == 0) st->solve_failed = 1;
    return;
  }

  // L^T y = z by wave 0 (z = L^-1 b was formed inside the factor loop): lane (p = lane >> 2, g = lane & 3) sums four
  // columns of a 16x16 matrix-vector product, and the four partial sums -- neighbouring lanes -- meet through two quad
  // permutes.  The diagonal tiles hold their inverses.
  if (wave == 0) {
    const int p = lane >> 2, g = lane & 3;
    double* vb = zb + TB * nt;  // 16 words: the vector a diagonal tile's inverse is applied to
    for (int i = nt - 1; i >= 0; --i) {
      double acc = 0.0;
#pragma unroll 4
      for (int k = i + 1; k < nt; ++k) {
        const double* T = Lt + ltile(k, i);                 // (L(k,i)^T y_k)(p) = sum_q L(k,i)(q, p) y_k(q)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += T[p * TB + 4 * g + q] * zb[TB * k + 4 * g + q];
      }
      const double v = zb[TB * i + p] - groups_sum(acc);
      if (g == 0) vb[p] = v;
      wave_sync();
      const double* D = Lt + ltile(i, i);
      double yi = 0.0;
      // y(p) = sum_q Linv(q, p) v(q)
#pragma unroll
      for (int q = 0; q < 4; ++q) yi += D[p * TB + 4 * g + q] * vb[4 * g + q];
      yi = groups_sum(yi);
      if (g == 0) zb[TB * i + p] = yi;
      wave_sync();
    }
  }
  __syncthreads();
  HIPBA_TICK(4);
  bool ok = true;
  for (int i = t; i < n; i += kSolveThreads) {
    const double v = zb[i];
    P.yc[6ll * wd.cam_base + i] = v;
    ok = ok && isfinite(v);
  }
  if (!ok) fail = 1;  // (several lanes may write the same value)
  __syncthreads();
  if (fail) {
    if (t == 0) st->solve_failed = 1;
    return;
  }
  if (!fuse_pose_update) return;
  pose_update_window(P, w, Lt);  // the matrix words are free: two sums per wave go there
}

// gradient check alone, for windows whose reduced system is solved by the large-matrix path
__global__ __launch_bounds__(kWideBlock) void k_window_gradient(DevP P, LmParams lm, int mode) {
  __shared__ double red[kWideBlock / 64];
  const int w = blockIdx.x;
  WinState* st = P.ws + w;
  if (!st->active) return;
  window_gradient_check(P, lm, P.wd[w], st, red, mode);
}

// ------------------------------------------------------------------------------------------
// ceres::Problem::Evaluate twin: one thread per observation, residual + Jacobians to HBM
// (k_residual_jacobian of the design; also the kernel measured against the 184 B/obs figure)
// ------------------------------------------------------------------------------------------
// An observation whose pose or landmark index is out of range reads nothing and yields NaN outputs (and so a NaN
// cost): callers that hand over device-resident index arrays cannot have them checked on the host.
__global__ __launch_bounds__(kBlock) void k_residual_jacobian(CameraK K, LmParams lm, int apply_loss, long long n_obs,
                                                              int n_poses, int n_points,
                                                              const double* __restrict__ poses,
                                                              const double* __restrict__ points,
                                                              const int* __restrict__ obs_pose,
                                                              const int* __restrict__ obs_point,
                                                              const double* __restrict__ obs_uv, double* res,
                                                              double* jac_pose, double* jac_point, double* cost_part) {
  __shared__ double red[4];
  // the 12 + 6 Jacobian doubles of a lane are turned around in LDS (odd pitches: no bank conflicts) so that
  // the block's 24 KB + 12 KB leave as contiguous 16-byte-per-lane streaming stores
  __shared__ __attribute__((aligned(16))) double sJ[kBlock * 13];
  const int t = threadIdx.x;
  const long long o0 = (long long)blockIdx.x * kBlock, o = o0 + t;
  const int nvalid = (int)(n_obs - o0 < kBlock ? n_obs - o0 : kBlock);
  double c[1] = {0.0};
  double r[2] = {0.0, 0.0}, Jc[12], Jp[6];
  if (o < n_obs) {
    const double2 uv = reinterpret_cast<const double2*>(obs_uv)[o];
    LmParams l2 = lm;
    if (!apply_loss) l2.loss = HIPBA_LOSS_NONE;
    const int ip = obs_pose[o], iq = obs_point[o];
    if (ip >= 0 && ip < n_poses && iq >= 0 && iq < n_points) {
      c[0] = eval_obs<true>(K, l2, poses + 7ll * ip, points + 3ll * iq, uv.x, uv.y, nullptr, nullptr, r, Jc, Jp);
    } else {
      const double bad = __builtin_nan("");
      c[0] = r[0] = r[1] = bad;
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = bad;
#pragma unroll
      for (int i = 0; i < 6; ++i) Jp[i] = bad;
    }
    if (res) reinterpret_cast<double2*>(res)[o] = make_double2(r[0], r[1]);
  }
  if (jac_pose) {
    if (o < n_obs) {
#pragma unroll
      for (int i = 0; i < 12; ++i) sJ[13 * t + i] = Jc[i];
    }
    __syncthreads();
    // pairs of doubles: 16 bytes per lane, streaming (the Jacobian is not read again by this kernel)
    double* dst = jac_pose + 12 * o0;
    for (int idx = t; idx < 6 * nvalid; idx += kBlock) {
      const int l = idx / 6, k = 2 * (idx - 6 * l);
      __builtin_nontemporal_store(sJ[13 * l + k], dst + 2 * idx);
      __builtin_nontemporal_store(sJ[13 * l + k + 1], dst + 2 * idx + 1);
    }
    __syncthreads();
  }
  if (jac_point) {
    if (o < n_obs) {
#pragma unroll
      for (int i = 0; i < 6; ++i) sJ[7 * t + i] = Jp[i];
    }
    __syncthreads();
    double* dst = jac_point + 6 * o0;
    for (int idx = t; idx < 3 * nvalid; idx += kBlock) {
      const int l = idx / 3, k = 2 * (idx - 3 * l);
      __builtin_nontemporal_store(sJ[7 * l + k], dst + 2 * idx);
      __builtin_nontemporal_store(sJ[7 * l + k + 1], dst + 2 * idx + 1);
    }
  }
  block_sum<1>(c, red);
  if (threadIdx.x == 0 && cost_part) cost_part[blockIdx.x] = c[0];
}

__global__ void k_sum_serial(const double* v, long long n, double* out) {
  // single workgroup, fixed order
  __shared__ double red[4];
  double a[1] = {0.0};
  for (long long i = threadIdx.x; i < n; i += kBlock) a[0] += v[i];
  block_sum<1>(a, red);
  if (threadIdx.x == 0) *out = a[0];
}

__global__ void k_count_active(DevP P, int* out) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int w = threadIdx.x; w < P.n_windows; w += blockDim.x) c += P.ws[w].active ? 1 : 0;
  if (c) atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) *out = cnt;
}

// ------------------------------------------------------------------------------------------
// launch wrappers (called from the host driver)
// ------------------------------------------------------------------------------------------
// LDS of k_point_schur: Y/W staging, slots, reduction words; outside staged mode the block's pair and run
// lists; the small-window modes add b and, up to kLdsSMaxCam poses, the packed S
// the owned-destination mode: Z staging, slots, reduction words, the landmark table, b, and the workgroup's 256
// blocks of S
size_t schur_owned_lds_bytes(int small_ncam) {
  const size_t ndest = (size_t)small_ncam * (small_ncam + 1) / 2;
  return (size_t)kBlock * kYW * 8 + (size_t)kBlock * 4 + 8 * 8 + (size_t)kBlock * 8 + (size_t)kSmallMaxN * 8 +
         std::min<size_t>(ndest, kBlock) * 36 * 8;
}
// workgroups per part in that mode: 256 destinations each
static int owned_dest_groups(int small_ncam) {
  const int ndest = small_ncam * (small_ncam + 1) / 2;
  return (ndest + kBlock - 1) / kBlock;
}

size_t schur_lds_bytes(int small_ncam, bool staged) {
  size_t base = (size_t)kBlock * (staged ? 9 : kYW) * 8 + (size_t)kBlock * 4 + 8 * 8;
  if (!staged) base += (size_t)kLdsPairs * 2 + (size_t)kLdsRuns * sizeof(RunDesc);
  if (small_ncam > 0) {
    base += (size_t)kSmallMaxN * 8;
    if (small_ncam <= kLdsSMaxCam) base += (size_t)small_ncam * (small_ncam + 1) / 2 * 36 * 8;
  }
  return base;
}

// Dynamic-LDS limits of the kernels that ask for more than the default 64 KB.  The attribute belongs to the function
// ON THE CURRENT DEVICE: called by hipba_create for every handle, with the handle's device current.
hipError_t configure_lm_kernels() {
  const void* big[] = {reinterpret_cast<const void*>(&k_point_schur<0>), reinterpret_cast<const void*>(&k_point_schur<1>),
                       reinterpret_cast<const void*>(&k_point_schur<2>), reinterpret_cast<const void*>(&k_point_schur<3>),
                       reinterpret_cast<const void*>(&k_small_linearize<1>), reinterpret_cast<const void*>(&k_small_linearize<3>),
                       reinterpret_cast<const void*>(&k_small_linearize<4>), reinterpret_cast<const void*>(&k_point_schur_owned)};
  for (const void* f : big) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_gather),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kBlock / 64) * kGatherLds * 8));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_window_solve_tiles), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024 - 256);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_window_solve_small),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
}

hipError_t launch_cost(const DevP& P, const LmParams& lm, int blk_off, int n_blk, int which, int ignore_active,
                       bool second_bank, hipStream_t s) {
  if (n_blk <= 0) return hipSuccess;
  DevP Q = P;
  if (second_bank) Q.partial = P.partial + ((long long)P.n_blk_pm + P.n_blk_vc + P.n_blk_cc) * kPartialStride;
  hipLaunchKernelGGL(k_cost, dim3((n_blk + kCostGroup - 1) / kCostGroup), dim3(kBlock), 0, s, Q, lm, blk_off,
                     blk_off + n_blk, which, ignore_active);
  return hipGetLastError();
}

hipError_t launch_camera_blocks(const DevP& P, const LmParams& lm, int use_scale, int fuse_init, hipStream_t s) {
  if (P.n_act_poses <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_camera_blocks, dim3(P.n_act_poses), dim3(kBlock), 0, s, P, lm, use_scale, fuse_init);
  return hipGetLastError();
}

hipError_t launch_make_scale_c(const DevP& P, int jacobi, hipStream_t s) {
  if (P.n_act_poses <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_make_scale_c, dim3((P.n_act_poses + 255) / 256), dim3(256), 0, s, P, jacobi);
  return hipGetLastError();
}

// iteration 0 of small windows in two launches instead of four (see k_first_blocks, k_scale_init_reduced)
hipError_t launch_first_blocks(const DevP& P, const LmParams& lm, hipStream_t s) {
  const unsigned grid = (unsigned)std::max(P.n_act_poses, 0) + (unsigned)std::max(P.n_blk_pm, 0);
  if (grid == 0 || P.n_long > 0) return grid == 0 ? hipSuccess : hipErrorInvalidValue;
  hipLaunchKernelGGL(k_first_blocks, dim3(grid), dim3(kBlock), 0, s, P, lm);
  return hipGetLastError();
}
hipError_t launch_scale_init_reduced(const DevP& P, const LmParams& lm, int jacobi, hipStream_t s) {
  if (P.n_act_poses <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scale_init_reduced, dim3((P.n_act_poses + 255) / 256), dim3(256), 0, s, P, lm, jacobi);
  return hipGetLastError();
}

hipError_t launch_point_colnorm(const DevP& P, const LmParams& lm, hipStream_t s) {
  if (P.n_blk_pm <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_point_colnorm, dim3(P.n_blk_pm), dim3(kBlock), 0, s, P, lm);
  if (P.n_long > 0) hipLaunchKernelGGL(k_long_point<0>, dim3(P.n_long), dim3(kBlock), 0, s, P, lm);
  return hipGetLastError();
}

hipError_t launch_init_reduced(const DevP& P, const LmParams& lm, int owner, hipStream_t s) {
  if (P.n_act_poses <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_init_reduced, dim3((P.n_act_poses + 255) / 256), dim3(256), 0, s, P, lm, owner);
  return hipGetLastError();
}

hipError_t launch_point_schur(const DevP& P, const LmParams& lm, int small_ncam, int parts, hipStream_t s) {
  if (P.n_blk_pm <= 0) return hipSuccess;
  if (P.prec) {
    hipLaunchKernelGGL(k_point_schur<2>, dim3(P.n_blk_pm), dim3(kBlock), schur_lds_bytes(0, true), s, P, lm, 1);
    if (P.n_long > 0) hipLaunchKernelGGL(k_long_point<1>, dim3(P.n_long), dim3(kBlock), 0, s, P, lm);
    const size_t gather_lds = (size_t)(kBlock / 64) * kGatherLds * 8;
    const unsigned grid = (unsigned)((P.n_chunks + kBlock / 64 - 1) / (kBlock / 64));
    if (grid) hipLaunchKernelGGL(k_schur_gather, dim3(grid), dim3(kBlock), gather_lds, s, P, lm);
    if (P.n_lane_jobs > 0)
      hipLaunchKernelGGL(k_schur_gather_lanes, dim3((unsigned)((P.n_lane_jobs + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, P, lm);
  } else if (small_ncam > 0 && P.dest_owned) {
    if (parts < 1) parts = 1;
    const int dgroups = owned_dest_groups(small_ncam);
    hipLaunchKernelGGL(k_point_schur_owned, dim3((unsigned)P.n_windows * (unsigned)parts * (unsigned)dgroups), dim3(kBlock),
                       schur_owned_lds_bytes(small_ncam), s, P, lm, parts, dgroups);
  } else if (small_ncam > 0) {
    if (parts < 1) parts = 1;
    const size_t lds = schur_lds_bytes(small_ncam, false);
    const unsigned grid = (unsigned)P.n_windows * (unsigned)parts;
    if (small_ncam <= kLdsSMaxCam)
      hipLaunchKernelGGL(k_point_schur<1>, dim3(grid), dim3(kBlock), lds, s, P, lm, parts);
    else
      hipLaunchKernelGGL(k_point_schur<3>, dim3(grid), dim3(kBlock), lds, s, P, lm, parts);
  } else {
    hipLaunchKernelGGL(k_point_schur<0>, dim3(P.n_blk_pm), dim3(kBlock), schur_lds_bytes(0, false), s, P, lm, 1);
  }
  return hipGetLastError();
}

// camera blocks (with the reduced system's initialisation) and landmark elimination of small windows in one launch;
// S and b hold zeros (see k_small_linearize)
hipError_t launch_small_linearize(const DevP& P, const LmParams& lm, int small_ncam, int parts, hipStream_t s) {
  if (small_ncam <= 0 || P.n_act_poses <= 0 || P.prec) return hipErrorInvalidValue;
  if (parts < 1) parts = 1;
  const int dgroups = P.dest_owned ? owned_dest_groups(small_ncam) : 1;
  // (the camera half needs 27 * 4 doubles of the LDS)
  const size_t lds = P.dest_owned ? schur_owned_lds_bytes(small_ncam) : schur_lds_bytes(small_ncam, false);
  const unsigned grid = (unsigned)P.n_act_poses + (P.n_blk_pm > 0 ? (unsigned)P.n_windows * (unsigned)parts * (unsigned)dgroups : 0u);
  if (P.dest_owned)
    hipLaunchKernelGGL(k_small_linearize<4>, dim3(grid), dim3(kBlock), lds, s, P, lm, parts, dgroups);
  else if (small_ncam <= kLdsSMaxCam)
    hipLaunchKernelGGL(k_small_linearize<1>, dim3(grid), dim3(kBlock), lds, s, P, lm, parts, 1);
  else
    hipLaunchKernelGGL(k_small_linearize<3>, dim3(grid), dim3(kBlock), lds, s, P, lm, parts, 1);
  return hipGetLastError();
}

hipError_t launch_window_solve_small(const DevP& P, const LmParams& lm, int max_n, int fuse_pose_update, int table_mode,
                                     hipStream_t s) {
  // development switch: HIPBA_SMALL_SOLVER_BLOCKS=1 keeps every window on the 6x6-block kernel (A/B timing)
  static const bool blocks_only = getenv("HIPBA_SMALL_SOLVER_BLOCKS") != nullptr;
  if (max_n >= kTileSolveMinN && !blocks_only) {
    const size_t nt = (size_t)(max_n + TB - 1) / TB;
    const size_t lds = (16 + nt * (nt + 1) / 2 * kTile + TB * nt + TB) * 8;
    hipLaunchKernelGGL(k_window_solve_tiles, dim3(P.n_windows), dim3(kSolveThreads), lds, s, P, lm, fuse_pose_update,
                       P.tile_src && P.n_windows == 1 ? table_mode : 0);
    return hipGetLastError();
  }
  const size_t nb = (size_t)(max_n + 5) / 6;
  const size_t lds = (16 + nb * (nb + 1) / 2 * kSmallBlk + 6 * nb) * 8;
  hipLaunchKernelGGL(k_window_solve_small, dim3(P.n_windows), dim3(kSolveThreads), lds, s, P, lm, fuse_pose_update);
  return hipGetLastError();
}

// Workgroup size of the one-workgroup-per-window kernels (gradient norms, candidate poses, step control): a lone large
// window has tens of thousands of per-block partial sums and thousands of poses behind one workgroup, so it gets 16
// waves instead of 4.  Batches and small windows keep 4, and with them the order of their sums.
static unsigned window_threads(const DevP& P) {
  return (P.n_windows == 1 && (long long)P.n_blk_pm + P.n_blk_vc >= 2048) ? kWideBlock : kBlock;
}

hipError_t launch_window_gradient(const DevP& P, const LmParams& lm, int mode, hipStream_t s) {
  hipLaunchKernelGGL(k_window_gradient, dim3(P.n_windows), dim3(window_threads(P)), 0, s, P, lm, mode);
  return hipGetLastError();
}

bool backsub_can_carry_cost(const DevP& P) { return P.n_long == 0; }

hipError_t launch_backsub(const DevP& P, const LmParams& lm, bool with_cost, hipStream_t s) {
  const int nb = P.n_blk_pm + P.n_blk_vc;
  if (nb <= 0) return hipSuccess;
  if (with_cost) {
    // (the second bank of partial[]: where launch_cost(..., second_bank) puts the candidate cost)
    double* bank2 = P.partial + ((long long)P.n_blk_pm + P.n_blk_vc + P.n_blk_cc) * kPartialStride;
    hipLaunchKernelGGL(k_backsub<true>, dim3(nb), dim3(kBlock), 0, s, P, lm, bank2);
  } else {
    hipLaunchKernelGGL(k_backsub<false>, dim3(nb), dim3(kBlock), 0, s, P, lm, (double*)nullptr);
  }
  if (P.n_long > 0) hipLaunchKernelGGL(k_long_point<2>, dim3(P.n_long), dim3(kBlock), 0, s, P, lm);
  return hipGetLastError();
}

hipError_t launch_pose_update(const DevP& P, hipStream_t s) {
  hipLaunchKernelGGL(k_pose_update, dim3(P.n_windows), dim3(window_threads(P)), 0, s, P);
  return hipGetLastError();
}

hipError_t launch_window_init(const DevP& P, const LmParams& lm, int mode, int zero_S, hipStream_t s) {
  hipLaunchKernelGGL(k_window_init, dim3(P.n_windows), dim3(kBlock), 0, s, P, lm, mode, zero_S);
  return hipGetLastError();
}

hipError_t launch_window_control(const DevP& P, const LmParams& lm, int mode, int* count_out, int zero_S, int count_tag,
                                 hipStream_t s) {
  hipLaunchKernelGGL(k_window_control, dim3(P.n_windows), dim3(window_threads(P)), 0, s, P, lm, mode, count_out, zero_S, count_tag);
  return hipGetLastError();
}

hipError_t launch_count_active(const DevP& P, int* out, hipStream_t s) {
  hipLaunchKernelGGL(k_count_active, dim3(1), dim3(256), 0, s, P, out);
  return hipGetLastError();
}

hipError_t launch_residual_jacobian(const CameraK& K, const LmParams& lm, int apply_loss, long long n_obs,
                                    int n_poses, int n_points, const double* poses, const double* points, const int* obs_pose,
                                    const int* obs_point, const double* obs_uv, double* res, double* jc, double* jp,
                                    double* cost_part, double* cost_out, hipStream_t s) {
  if (n_obs <= 0) return hipSuccess;
  const long long nb = (n_obs + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_residual_jacobian, dim3((unsigned)nb), dim3(kBlock), 0, s, K, lm, apply_loss, n_obs, n_poses,
                     n_points, poses, points, obs_pose, obs_point, obs_uv, res, jc, jp, cost_part);
  if (cost_part && cost_out) hipLaunchKernelGGL(k_sum_serial, dim3(1), dim3(kBlock), 0, s, cost_part, nb, cost_out);
  return hipGetLastError();
}

}  // namespace hipba

#ifdef HIPBA_PHASE_TIMING
extern "C" int hipba_debug_ticks(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipba::g_ticks), sizeof(unsigned long long) * n);
}
#endif
