// ba_device.hpp -- device-side data layout of a resident bundle-adjustment problem.
//
// One resident problem holds n_windows independent windows (one for hipba_solve, many
// for hipba_solve_batched).  Everything a kernel needs travels in a DevP passed by
// value; the per-window Levenberg-Marquardt state lives in device memory (WinState) so
// that acceptance, trust-region update and termination run on the GPU.
//
// Observation classes, by which of the two parameter blocks is variable
// (reference: SetParameterBlockConstant at src/solve/optimizer.cpp:30,33):
//   PM section  variable landmark (pose variable or constant), sorted by
//               (window, landmark, pose): a landmark's observations are one contiguous
//               segment, and segments never straddle a 256-observation block.
//   VC section  variable pose, constant landmark.
//   CC section  both constant: evaluated once, Ceres' "fixed cost".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hipba.h"
#include "ba_math.hpp"

namespace hipba {

constexpr int kBlock = 256;        // threads per workgroup in the observation kernels
constexpr int kWideBlock = 1024;   // threads of a per-window kernel when the window is a lone large one
constexpr int kPartialStride = 4;  // doubles of per-block partial results
constexpr int kSmallMaxN = 192;    // largest reduced system solved inside one workgroup's LDS
constexpr int kLdsSMaxCam = 21;    // most poses for which the packed reduced matrix is accumulated in LDS
constexpr int kSmallBlk = 37;      // LDS stride of a 6x6 block in that solver (odd: neighbouring blocks fall on different banks)

// modes of the per-window kernels (see lm_kernels.hip)
enum { kFused = 0, kReduce = 1, kDecide = 2 };

struct BlockDesc {
  long long obs_begin;   // first observation (index into obs_* arrays)
  long long pair_begin;  // first entry of this block's pair list (a multiple of 8)
  long long run_begin;   // first entry of this block's run list (a multiple of 2)
  int n_obs;             // <= kBlock
  int n_pairs;
  int n_runs;
  int window;
  int kind;              // 0 = PM, 1 = VC, 2 = CC
  int pad;               // PM: 0, or 1 + the index of the LongDesc whose landmark this block is a piece of
};

// A landmark with more observations than one block holds (the reference adds any number of residual blocks per
// landmark, src/solve/optimizer.cpp:37-51): its observations fill a run of blocks of their own, and one workgroup
// (k_long_point) walks them all for the sums the other landmarks form inside one block.
struct LongDesc {
  long long obs_begin;
  int n_obs;
  int pp;          // device landmark index
  int blk_begin;   // its blocks: blk_begin .. blk_begin + n_blk - 1
  int n_blk;
};

// a run: consecutive entries of a block's pair list that go to the same 6x6 block of the reduced matrix
struct RunDesc {
  unsigned int first_count;  // (offset of the first pair in the block's list << 8) | number of pairs
  unsigned int dest;         // packed upper block-triangular index within the window
};
constexpr int kRunMax = 8;                        // pairs per run
constexpr unsigned short kPairPad = 0x00ff;       // filler entry of a pair list (lane 255 before lane 0: never a pair)
constexpr int kLdsPairs = 4096, kLdsRuns = 1024;  // a block's lists are staged in LDS up to these sizes

struct WinDesc {
  CameraK K;
  int pose_begin, pose_end;    // device pose range of this window
  int cam_base, ncam;          // first global slot and number of variable observed poses
  long long S_off;             // offset (doubles) of this window's packed reduced matrix
  int pm_blk_begin, pm_blk_end;
  int vc_blk_begin, vc_blk_end;
  int cc_blk_begin, cc_blk_end;
  int pp_begin, pp_end;        // variable observed landmarks of this window
  int log_base;                // first entry of this window in the iteration log
  int pad;
};

struct WinState {
  int sel;              // which parameter buffer holds x
  int active;           // still iterating
  int termination;      // HIPBA_CONVERGENCE / NO_CONVERGENCE / FAILURE
  int iteration;        // index of the last logged iteration
  int n_invalid;        // consecutive invalid steps
  int any_success;
  int num_successful;
  int num_unsuccessful;
  int solve_failed;     // set by the reduced solver of the current iteration
  int last_success_log; // log entry whose gradient norms are filled by the next linearisation
  int lin_iters;
  int num_linear_solves; // trust-region steps computed, the one a convergence test ends on included
  int handoff_timeouts;  // sticky: a bounded wait between workgroups of the reduced solver ran out (reported as an error)
  int pad_;
  double radius;
  double decrease_factor;
  double x_cost;
  double initial_cost;
  double fixed_cost;
  double gmax, gnorm;
  // scratch written by per-window kernels of the current iteration
  double pose_step2, pose_x2, pose_gmax, pose_g2;
};

struct LmParams {
  int max_num_iterations;
  int loss;
  double loss_scale;
  int jacobi_scaling;
  int max_invalid;
  double max_radius, min_radius, min_relative_decrease;
  double min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius;
  int param_tol_needs_success;  // hipba_options::parameter_tolerance_needs_success
};

struct DevP {
  int n_poses, n_points, n_windows;
  int n_act_poses;     // global slots
  int n_pm_points;
  long long n_pm_obs, n_act_obs, n_all_obs;
  int n_blk_pm, n_blk_vc, n_blk_cc;  // blocks are stored PM | VC | CC
  // parameters, double buffered (x and candidate); constant / unobserved blocks identical in both
  double* poses[2];
  double* points[2];
  // per pose
  const int* pose_window;
  const int* pose_gslot;   // global slot or -1
  // per global slot
  const int* slot_pose;
  const long long* cm_start;  // [n_act_poses+1] camera-major observation lists
  const int* cm_dp;
  const double* cm_uv;        // [2*]
  // observations (PM | VC | CC)
  const int* obs_pose;
  const int* obs_dp;
  const double* obs_uv;       // [2*]
  const long long* pp_obs_start;  // [n_pm_points+1]
  const BlockDesc* blocks;
  const LongDesc* longs;          // landmarks longer than a block (gathered elimination only)
  int n_long;
  const unsigned short* pairs;    // low byte = lane i, high byte = lane j, slot_i <= slot_j
  const RunDesc* runs;            // pair lists cut into runs of one destination (absent in staged mode)
  // gathered accumulation of the reduced matrix (one large window): every 6x6 block of S sums the products of
  // its observation pairs, listed in destination order; a product is recomputed from the landmark's record
  // and the two poses, so nothing per observation is kept in HBM.  Null when fp64 atomics are used instead.
  const uint2* pair_pt;           // [n_pairs] (landmark | self pair << 31, (row slot << 16) | column slot)
  const uint2* pair_obs;          // [n_pairs] (observation i, observation j): where a robust loss reads its weights
  const long long* chunk_start;   // [2 * n_chunks] (begin, end) pair ranges cut at destination boundaries, one wave
                                  // each, longest first
  int n_chunks;
  const uint4* lane_jobs;         // [n_lane_jobs] off-diagonal blocks of S that one LANE sums (k_schur_gather_lanes):
                                  // (first pair, pairs, (row slot << 16) | column slot, 0), longest first
  int n_lane_jobs;
  double* prec;                   // [16 * n_pm_points] landmark record, one 128-byte line: p (3), scale_p (3),
                                  // V^-1 (6), scale_p o (V^-1 g_p) (3), one unused
  double* fw;                     // [n_pm_obs] sqrt(rho') of every observation (robust loss only)
  double* cam_tab;                // [18 * n_act_poses] R^T (9), R^T t (3), scale_c (6) of every variable pose
  const WinDesc* wd;
  WinState* ws;
  // work arrays
  double* U;        // [36*n_act_poses] scaled camera blocks
  double* gc;       // [6*n_act_poses]  scaled camera gradient
  double* scale_c;  // [6*n_act_poses]
  double* b;        // [6*n_act_poses]  reduced right-hand side
  double* yc;       // [6*n_act_poses]  reduced solution
  double* Vinv;     // [6*n_pm_points]  (V + D^2)^-1, symmetric packing
  double* gp;       // [3*n_pm_points]  scaled landmark gradient
  double* scale_p;  // [3*n_pm_points]
  double* S;        // packed upper block-triangular reduced matrices, 36 doubles per 6x6 block
  // Set for a solve whose every block of S is written in full each iteration and whose reduced system goes to the
  // dense factorisation: the blocks are then written straight into that solver's column-major lower triangle
  // (element (r, c) of block (i, j), i <= j, is row 6 j + c of column 6 i + r) and S is not used at all.
  double* dense_S;
  int dense_ld;
  double* partial;  // [2 banks][kPartialStride * n_blocks]
  double* gpart;    // [2 * n_blk_pm] landmark gradient max / sum of squares per PM block
  double* wsum;     // [8 * n_windows] per-window sums handed to the all-reduce of sharded solves
  int* tile_src;    // one small window of 48 unknowns and more: where every word of the in-LDS tile image of S comes
                    // from in the packed S (k_window_solve_tiles writes it in a solve's first pass, reads it after)
  int sharded;      // 1 when this rank holds only a shard of the window's landmarks
  int dest_owned;   // small windows: the elimination lets every lane own one block of S and find its products itself
                    // (no pair or run lists were built: see point_schur_body, MODE 4)
  hipba_iteration* log;
  int log_cap;      // entries per window
};

// packed upper block-triangular index of block (r, c), r <= c, for n block rows
__host__ __device__ __forceinline__ long long tri_index(int r, int c, int n) {
  return (long long)r * n - ((long long)r * (r - 1)) / 2 + (c - r);
}

}  // namespace hipba
