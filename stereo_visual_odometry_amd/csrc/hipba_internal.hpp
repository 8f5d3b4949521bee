// hipba_internal.hpp -- host-side declarations shared by the translation units of libhipba.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "ba_device.hpp"

namespace hipba {

// ---- kernel launch wrappers (lm_kernels.hip) ---------------------------------------------------
hipError_t configure_lm_kernels();     // per device, from hipba_create
hipError_t configure_dense_kernels();  // (dense_solver.hip)
size_t schur_lds_bytes(int small_ncam, bool staged);
hipError_t launch_cost(const DevP& P, const LmParams& lm, int blk_off, int n_blk, int which, int ignore_active,
                       bool second_bank, hipStream_t s);
hipError_t launch_camera_blocks(const DevP& P, const LmParams& lm, int use_scale, int fuse_init, hipStream_t s);
hipError_t launch_make_scale_c(const DevP& P, int jacobi, hipStream_t s);
hipError_t launch_point_colnorm(const DevP& P, const LmParams& lm, hipStream_t s);
hipError_t launch_init_reduced(const DevP& P, const LmParams& lm, int owner, hipStream_t s);
hipError_t launch_point_schur(const DevP& P, const LmParams& lm, int small_ncam, int parts, hipStream_t s);
hipError_t launch_small_linearize(const DevP& P, const LmParams& lm, int small_ncam, int parts, hipStream_t s);
hipError_t launch_window_solve_small(const DevP& P, const LmParams& lm, int max_n, int fuse_pose_update, int table_mode,
                                     hipStream_t s);
hipError_t launch_window_gradient(const DevP& P, const LmParams& lm, int mode, hipStream_t s);
// with_cost: the pass also evaluates the cost at the candidate point (no launch_cost(which = 1) after it); the
// candidate poses must have been formed before, and the problem must have no landmark that spans several blocks
bool backsub_can_carry_cost(const DevP& P);
hipError_t launch_backsub(const DevP& P, const LmParams& lm, bool with_cost, hipStream_t s);
hipError_t launch_pose_update(const DevP& P, hipStream_t s);
hipError_t launch_window_init(const DevP& P, const LmParams& lm, int mode, int zero_S, hipStream_t s);
hipError_t launch_first_blocks(const DevP& P, const LmParams& lm, hipStream_t s);
hipError_t launch_scale_init_reduced(const DevP& P, const LmParams& lm, int jacobi, hipStream_t s);
hipError_t launch_window_control(const DevP& P, const LmParams& lm, int mode, int* count_out, int zero_S, int count_tag,
                                 hipStream_t s);
hipError_t launch_count_active(const DevP& P, int* out, hipStream_t s);
hipError_t launch_residual_jacobian(const CameraK& K, const LmParams& lm, int apply_loss, long long n_obs,
                                    int n_poses, int n_points, const double* poses, const double* points, const int* obs_pose,
                                    const int* obs_point, const double* obs_uv, double* res, double* jc, double* jp,
                                    double* cost_part, double* cost_out, hipStream_t s);

// ---- stereo triangulation (stereo_kernels.hip): every pointer is device memory; pose may be null ----
hipError_t launch_triangulate_stereo(const double* d_pl, const double* d_pr, long long n, const float* d_left,
                                     const float* d_right, const double* d_pose, double* d_out, hipStream_t s);

// out[i] = poses[i] * exp(deltas[i]) for n poses (7 / 6 / 7 doubles each); device memory
hipError_t launch_se3_plus(long long n, const double* d_poses, const double* d_deltas, double* d_out, hipStream_t s);

// PnP hypotheses by random sample consensus: n_hyp poses from four-match samples (Gauss-Newton from d_prior), their inlier
// counts, then d_out = the best pose [7] + its count [1] and d_mask [n]; every pointer is device memory
hipError_t launch_pnp_ransac(const CameraK& K, long long n, const double* d_pts, const double* d_uv, const double* d_prior,
                             int closed_form, int n_hyp, unsigned long long seed, double reprojection_error,
                             double confidence, int* d_count, double* d_poses, double* d_out, unsigned char* d_mask,
                             hipStream_t s);

// ---- reduced-system solvers for one large window (dense_solver.hip) ----------------------------
struct BigWork {
  int n = 0;           // 6 * ncam
  int npad = 0;        // n rounded up to the factorisation panel width
  int ld = 0;          // leading dimension of A
  double* A = nullptr;     // [(npad + NB)^2] column-major, lower triangle holds S then its Cholesky factor;
                           // row npad carries b^T, then z^T = (L^-1 b)^T
  double* Linv = nullptr;  // [npad/NB][NB*NB] inverses of the diagonal blocks of L
  double* LinvT = nullptr; // the same blocks transposed (backward substitution reads them by rows)
  double* rhs = nullptr;   // [npad]
  double* pcg = nullptr;   // PCG vectors and scalars
  double* Minv = nullptr;  // [36 * ncam] block-Jacobi preconditioner
  int* first = nullptr;    // [npad/NB + 1] first tile column of every tile row of the factor's envelope
  unsigned* sync = nullptr;  // arrival counter of the in-launch hand-offs of one factorisation; [1]: the CU of the
                             // workgroup that factors the next diagonal block
  unsigned* flags = nullptr; // [2 (npad/NB + 1)] "y_k is ready" words of the backward sweep, then one tile counter per
                             // panel for the streamed band updates
  int2* tiles = nullptr;     // (tile row, tile column) of the lower-triangle tiles inside the envelope (sharded solves)
  size_t bytes() const;
};
constexpr int kCholNB = 128;
size_t big_work_bytes(int n, int ncam);
void big_work_bind(BigWork& W, int n, int ncam, void* base);
// A sharded solve may hand the factorisation the other ranks' part of the matrix while it runs: the packed tiles
// (W.tiles, ordered by tile column) are summed over the ranks in chunks of whole tile columns on a second stream, and
// the factorisation adds a chunk to its matrix just before the first panel that needs one of its columns.  That is
// exact because every step of a right-looking factorisation before that point only ever ADDS to those columns.
constexpr int kMaxFeedChunks = 16;
struct TileFeed {
  int n_chunks = 0;
  int first_col[kMaxFeedChunks] = {};       // first tile column of chunk c (ascending; chunk 0 starts at column 0)
  int tile_begin[kMaxFeedChunks + 1] = {};  // tiles [tile_begin[c], tile_begin[c + 1]) of W.tiles / of the buffer
  hipEvent_t ready[kMaxFeedChunks] = {};    // recorded behind chunk c's all-reduce
  const double* packed = nullptr;           // the buffer the chunks arrive in
};
// Cholesky: S (packed, P.S) y = b (P.b) -> P.yc; marks ws.solve_failed when S is not positive definite
// h_first: the envelope (host copy of W.first, camera_order.hpp), or null for a dense factorisation
hipError_t big_solve_cholesky(const DevP& P, const BigWork& W, const int* h_first, hipStream_t s,
                              const TileFeed* feed = nullptr);
// zeroes the lower triangle of W.A (enveloped: only inside the envelope in W.first): for solves whose elimination
// writes the blocks of S straight into it (DevP::dense_S)
hipError_t big_zero_lower(const DevP& P, const BigWork& W, bool enveloped, hipStream_t s);
// moves tiles [tile0, tile0 + n_tiles) of W.tiles between the dense matrix and a packed buffer (sharded solves'
// exchange): kTilesUnpack buffer -> matrix, kTilesPack matrix -> buffer, kTilesTake matrix -> buffer and zeroes in
// the matrix, kTilesAdd matrix += buffer (the last two: on and below the diagonal of the leading n x n part only)
enum { kTilesUnpack = 0, kTilesPack = 1, kTilesTake = 2, kTilesAdd = 3 };
hipError_t big_env_tiles(const DevP& P, const BigWork& W, int tile0, int n_tiles, double* buf, int mode, hipStream_t s);
// block-Jacobi preconditioned conjugate gradients on the packed S
hipError_t big_solve_pcg(const DevP& P, const BigWork& W, int max_iter, double tol, hipStream_t s);

}  // namespace hipba

// development aid (-DHIPBA_PHASE_TIMING): 100 MHz timestamps of workgroup phases, read back by hipba_debug_ticks
#ifdef HIPBA_PHASE_TIMING
namespace hipba { extern __device__ unsigned long long g_ticks[8 * 4096]; }
#define HIPBA_TICK(i) \
  if (threadIdx.x == 0 && blockIdx.x < 4096) hipba::g_ticks[8 * blockIdx.x + (i)] = wall_clock64()
#define HIPBA_CYCLES(i) \
  if (threadIdx.x == 0 && blockIdx.x < 4096) hipba::g_ticks[8 * blockIdx.x + (i)] = clock64()
#define HIPBA_CYCLES_AT(slot, i) \
  if (threadIdx.x == 0 && (slot) < 4096) hipba::g_ticks[8 * (slot) + (i)] = clock64()
#else
#define HIPBA_TICK(i)
#define HIPBA_CYCLES(i)
#define HIPBA_CYCLES_AT(slot, i)
#endif
