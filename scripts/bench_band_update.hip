// Stand-alone bench of the K = 128 band update, C(i,j) -= L(i,p) L(j,p)^T on 64x64 tiles of a lower-triangular band,
// with the inner product engine as a template parameter.  Same streaming structure as k_update_stream
// (dense_solver.hip) without the fused factorisation.
// hipcc -O3 --offload-arch=gfx950 scripts/bench_band_update.hip -o scripts/dev/bench_band_update
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int NB = 128, KC = 16;
static int g_lds = 76000;

template <int CTRL>
__device__ __forceinline__ double rot_groups(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ v4d mfma_rot(double a, double b, v4d acc) {
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(rot_groups<0x124>(a), b, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(rot_groups<0x128>(a), b, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(rot_groups<0x12c>(a), b, acc[3], 0, 0, 0);
  return acc;
}
__device__ __forceinline__ double rot_to_tile(v4d acc, int r, int lane) {
  const int s = (((lane >> 2) & 3) - r) & 3;
  return s == 0 ? acc[0] : (s == 1 ? acc[1] : (s == 2 ? acc[2] : acc[3]));
}

// MODE 0: 16x16x4 instruction; 1: 4x4x4 + DPP rotations; 2: 4x4x4, rotated copies read from LDS;
// 3: as 1 with all of a stage's LDS reads issued before its first product
template <int MODE, int FLAGS = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_update(double* Amat, int ld, int i0, int kcol,
                                                                                          int row_strips, int col_strips) {
  constexpr int TS = 64, LDS_ = TS + 16, PT = TS / 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*sA)[KC * LDS_] = reinterpret_cast<double (*)[KC * LDS_]>(smem);
  double (*sB)[KC * LDS_] = reinterpret_cast<double (*)[KC * LDS_]>(smem + 2 * KC * LDS_);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lcol = t >> 5, lsub = (t & 31) * 2;
  const int total = col_strips * row_strips - col_strips * (col_strips - 1) / 2;
  const int stride = (int)gridDim.x;
  int n = blockIdx.x;
  if (n >= total) return;
  const double* pan = Amat + (long long)kcol * ld;
#define STREAM_DECODE(n_, row0_, col0_)                                     \
  {                                                                         \
    int y_ = 0, len_ = row_strips, m_ = (n_);                               \
    while (m_ >= len_) { m_ -= len_; --len_; ++y_; }                        \
    const int x_ = y_ + m_;                                                 \
    row0_ = i0 * NB + x_ * TS;                                              \
    col0_ = i0 * NB + y_ * TS;                                              \
  }
  int row0, col0, row0n = 0, col0n = 0;
  STREAM_DECODE(n, row0, col0);
  v2d ra[4][PT / 2], rb[4][PT / 2];
  if (FLAGS & 1) { for (int i = 0; i < 4; ++i) for (int m = 0; m < PT / 2; ++m) { ra[i][m] = (v2d){1e-3 * t, 2e-3}; rb[i][m] = (v2d){3e-3, 1e-3 * t}; } }
#define STREAM_GLOAD(slot, chunk, r0_, c0_)                                                                    \
  if (!(FLAGS & 1)) {                                                                                          \
    _Pragma("unroll") for (int m = 0; m < PT / 2; ++m) {                                                       \
      const long long o_ = (long long)((chunk)*KC + 8 * m + lcol) * ld + lsub;                                 \
      ra[slot][m] = *reinterpret_cast<const v2d*>(pan + o_ + (r0_));                                           \
      rb[slot][m] = *reinterpret_cast<const v2d*>(pan + o_ + (c0_));                                           \
    }                                                                                                          \
  }
  STREAM_GLOAD(0, 0, row0, col0); STREAM_GLOAD(1, 1, row0, col0); STREAM_GLOAD(2, 2, row0, col0); STREAM_GLOAD(3, 3, row0, col0);
  // lane's source lane group for the LDS-rotated reads of MODE 2: group g reads the rows of group (g - s) & 3
  const int l15 = lane & 15;
  const int rot1 = ((l15 - 4) & 15), rot2 = ((l15 - 8) & 15), rot3 = ((l15 - 12) & 15);
  for (;;) {
    const int n_next = n + stride;
    const bool has_next = n_next < total;
    if (has_next) STREAM_DECODE(n_next, row0n, col0n);
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    double cold[2][2][4];
    double* Cp = Amat + (long long)col0 * ld + row0;
#define STREAM_STAGE(c, AFTER_SYNC)                                                                               \
  {                                                                                                               \
    constexpr int buf = (c) & 1, slot = (c) & 3;                                                                  \
    {                                                                                                             \
      _Pragma("unroll") for (int m = 0; m < PT / 2; ++m) {                                                        \
        *reinterpret_cast<v2d*>(&sA[buf][(8 * m + lcol) * LDS_ + lsub]) = ra[slot][m];                            \
        *reinterpret_cast<v2d*>(&sB[buf][(8 * m + lcol) * LDS_ + lsub]) = rb[slot][m];                            \
      }                                                                                                           \
    }                                                                                                             \
    __syncthreads();                                                                                              \
    AFTER_SYNC                                                                                                    \
    if (MODE == 3) {                                                                                              \
      double fbs[4][2], fas[4][2];                                                                                \
      _Pragma("unroll") for (int kk = 0; kk < KC / 4; ++kk) {                                                     \
        const int kr = kk * 4 + (lane >> 4);                                                                      \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) fbs[kk][m] = sA[buf][kr * LDS_ + l15 + wr * 32 + m * 16];   \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) fas[kk][m] = sB[buf][kr * LDS_ + l15 + wc * 32 + m * 16];   \
      }                                                                                                           \
      _Pragma("unroll") for (int kk = 0; kk < KC / 4; ++kk)                                                       \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                               \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) acc[a][b] = mfma_rot(fas[kk][a], fbs[kk][b], acc[a][b]);      \
    } else {                                                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < KC / 4; ++kk) {                                                       \
      const int kr = kk * 4 + (lane >> 4);                                                                        \
      double fb[2];                                                                                               \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) fb[m] = sA[buf][kr * LDS_ + l15 + wr * 32 + m * 16];          \
      _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                             \
        const double fa = sB[buf][kr * LDS_ + l15 + wc * 32 + a * 16];                                            \
        if (MODE == 4) {                                                                                          \
          acc[a][0][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb[0] + fb[1], acc[a][0][0], 0, 0, 0);            \
        } else if (MODE == 5) {                                                                                   \
          _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)          \
            acc[a][b][s_] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb[b], acc[a][b][s_], 0, 0, 0);                \
        } else if (MODE == 0) {                                                                                   \
          _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                           \
              acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[b], acc[a][b], 0, 0, 0);                    \
        } else if (MODE == 1) {                                                                                   \
          _Pragma("unroll") for (int b = 0; b < 2; ++b) acc[a][b] = mfma_rot(fa, fb[b], acc[a][b]);               \
        } else {                                                                                                  \
          const double f1 = sB[buf][kr * LDS_ + rot1 + wc * 32 + a * 16];                                         \
          const double f2 = sB[buf][kr * LDS_ + rot2 + wc * 32 + a * 16];                                         \
          const double f3 = sB[buf][kr * LDS_ + rot3 + wc * 32 + a * 16];                                         \
          _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                         \
            acc[a][b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, fb[b], acc[a][b][0], 0, 0, 0);                  \
            acc[a][b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(f1, fb[b], acc[a][b][1], 0, 0, 0);                  \
            acc[a][b][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(f2, fb[b], acc[a][b][2], 0, 0, 0);                  \
            acc[a][b][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(f3, fb[b], acc[a][b][3], 0, 0, 0);                  \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    }                                                                                                             \
  }
    STREAM_STAGE(0, STREAM_GLOAD(0, 4, row0, col0);)
    STREAM_STAGE(1, STREAM_GLOAD(1, 5, row0, col0);)
    STREAM_STAGE(2, STREAM_GLOAD(2, 6, row0, col0);)
    STREAM_STAGE(3, STREAM_GLOAD(3, 7, row0, col0);
                 _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)
                 _Pragma("unroll") for (int r = 0; r < 4; ++r)
                     cold[a][b][r] = (FLAGS & 2) ? 0.0 : Cp[(long long)(wc * 32 + a * 16 + (lane >> 4) + 4 * r) * ld + wr * 32 + b * 16 + (lane & 15)];)
    STREAM_STAGE(4, if (has_next) STREAM_GLOAD(0, 0, row0n, col0n);)
    STREAM_STAGE(5, if (has_next) STREAM_GLOAD(1, 1, row0n, col0n);)
    STREAM_STAGE(6, if (has_next) STREAM_GLOAD(2, 2, row0n, col0n);)
    STREAM_STAGE(7, if (has_next) STREAM_GLOAD(3, 3, row0n, col0n);)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(FLAGS & 2) || (cold[a][b][r] - (MODE == 0 ? acc[a][b][r] : rot_to_tile(acc[a][b], r, lane))) == 12345.678)
          Cp[(long long)(wc * 32 + a * 16 + (lane >> 4) + 4 * r) * ld + wr * 32 + b * 16 + (lane & 15)] =
              cold[a][b][r] - (MODE == 0 ? acc[a][b][r] : rot_to_tile(acc[a][b], r, lane));
    if (!has_next) break;
    n = n_next; row0 = row0n; col0 = col0n;
  }
}

// ---- no LDS, no barriers: every wave feeds the matrix cores straight from global memory --------------------------
// A wave owns a (16 RB rows) x (16 RA columns) piece of the workgroup's tile (2 x 2 waves).  The operands of one
// k-step of four are loaded in the instruction's own layout (lane l: row l & 15 of a strip, column k0 + (l >> 4) of
// the panel: four 128-byte runs per load), D k-steps ahead of their use, across tile boundaries.
template <int RA, int RB, int D>
__global__ __launch_bounds__(256) void k_update_direct(double* Amat, int ld, int i0, int kcol, int row_strips, int col_strips) {
  constexpr int TR = 32 * RB, TC = 32 * RA;          // workgroup tile
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int total = 0;  // (strips counted in workgroup tiles; column strip y starts at the tile row that holds its diagonal)
  for (int y = 0; y < col_strips; ++y) total += row_strips - y * TC / TR;
  const int stride = (int)gridDim.x;
  int n = blockIdx.x;
  if (n >= total) return;
  const double* pan = Amat + (long long)kcol * ld + (long long)(lane >> 4) * ld + (lane & 15);
  auto decode = [&](int n_, int& row0_, int& col0_) {
    int y_ = 0, m_ = n_;
    for (;;) { const int len_ = row_strips - y_ * TC / TR; if (m_ < len_) break; m_ -= len_; ++y_; }
    const int x_ = y_ * TC / TR + m_;
    row0_ = i0 * NB + x_ * TR + wr * 16 * RB;
    col0_ = i0 * NB + y_ * TC + wc * 16 * RA;
  };
  int row0, col0;
  decode(n, row0, col0);
  double fa[D][RA], fb[D][RB];
  const double* pa = pan + col0;   // C columns: first operand (rotated)
  const double* pb = pan + row0;   // C rows: second operand
#define DIRECT_LOAD(slot)                                                        \
  {                                                                              \
    _Pragma("unroll") for (int m = 0; m < RA; ++m) fa[slot][m] = pa[16 * m];     \
    _Pragma("unroll") for (int m = 0; m < RB; ++m) fb[slot][m] = pb[16 * m];     \
    pa += 4 * ld; pb += 4 * ld;                                                  \
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) DIRECT_LOAD(sl);
  for (;;) {
    const int n_next = n + stride;
    const bool has_next = n_next < total;
    int row0n = row0, col0n = col0;
    if (has_next) decode(n_next, row0n, col0n);
    v4d acc[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
      for (int b = 0; b < RB; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    double* Cp = Amat + (long long)col0 * ld + row0;
    for (int it = 0; it < 32 / D; ++it) {
      if (it == 32 / D - 1) {  // the loads issued from here on belong to the next tile
        pa = pan + col0n; pb = pan + row0n;
      }
#pragma unroll
      for (int sl = 0; sl < D; ++sl) {
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
          for (int b = 0; b < RB; ++b) acc[a][b] = mfma_rot(fa[sl][a], fb[sl][b], acc[a][b]);
        if (it < 32 / D - 1 || has_next) DIRECT_LOAD(sl);
      }
    }
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double* c = Cp + (long long)(a * 16 + (lane >> 4) + 4 * r) * ld + b * 16 + (lane & 15);
          *c = *c - rot_to_tile(acc[a][b], r, lane);
        }
    if (!has_next) break;
    n = n_next; row0 = row0n; col0 = col0n;
  }
#undef DIRECT_LOAD
}

// ---- LDS-staged, software-pipelined: the operands of k-step kk + 1 travel from LDS to registers while the products
// of k-step kk are issued; the stage's barrier sits in front of the last k-step's products, with the next stage's
// first operands requested right behind it.  A wave owns 16 columns x 64 rows of the 64x64 tile (one rotated operand
// strip: 6 DPP moves per 16 products).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_update_pipe(double* Amat, int ld, int i0, int kcol,
                                                                                               int row_strips, int col_strips) {
  constexpr int TS = 64, LDS_ = TS + 16, PT = TS / 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*sA)[KC * LDS_] = reinterpret_cast<double (*)[KC * LDS_]>(smem);
  double (*sB)[KC * LDS_] = reinterpret_cast<double (*)[KC * LDS_]>(smem + 2 * KC * LDS_);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lcol = t >> 5, lsub = (t & 31) * 2;
  const int total = col_strips * row_strips - col_strips * (col_strips - 1) / 2;
  const int stride = (int)gridDim.x;
  int n = blockIdx.x;
  if (n >= total) return;
  const double* pan = Amat + (long long)kcol * ld;
#define PIPE_DECODE(n_, row0_, col0_)                                       \
  {                                                                         \
    int y_ = 0, len_ = row_strips, m_ = (n_);                               \
    while (m_ >= len_) { m_ -= len_; --len_; ++y_; }                        \
    const int x_ = y_ + m_;                                                 \
    row0_ = i0 * NB + x_ * TS;                                              \
    col0_ = i0 * NB + y_ * TS;                                              \
  }
  int row0, col0, row0n = 0, col0n = 0;
  PIPE_DECODE(n, row0, col0);
  v2d ra[4][PT / 2], rb[4][PT / 2];
#define PIPE_GLOAD(slot, chunk, r0_, c0_)                                                                      \
  {                                                                                                            \
    _Pragma("unroll") for (int m = 0; m < PT / 2; ++m) {                                                       \
      const long long o_ = (long long)((chunk)*KC + 8 * m + lcol) * ld + lsub;                                 \
      ra[slot][m] = *reinterpret_cast<const v2d*>(pan + o_ + (r0_));                                           \
      rb[slot][m] = *reinterpret_cast<const v2d*>(pan + o_ + (c0_));                                           \
    }                                                                                                          \
  }
#define PIPE_LWRITE(buf, slot)                                                                                 \
  {                                                                                                            \
    _Pragma("unroll") for (int m = 0; m < PT / 2; ++m) {                                                       \
      *reinterpret_cast<v2d*>(&sA[buf][(8 * m + lcol) * LDS_ + lsub]) = ra[slot][m];                           \
      *reinterpret_cast<v2d*>(&sB[buf][(8 * m + lcol) * LDS_ + lsub]) = rb[slot][m];                           \
    }                                                                                                          \
  }
  // operands of one k-step: fa = 16 columns of the wave (first operand, rotated), fb[4] = the 64 rows
#define PIPE_LREAD(buf, kk, fa_, fb_)                                                                          \
  {                                                                                                            \
    const int kr_ = (kk) * 4 + (lane >> 4);                                                                    \
    fa_[0] = sB[buf][kr_ * LDS_ + (lane & 15) + wave * 16];                                                    \
    if (MODE == 2) {                                                                                           \
      fa_[1] = sB[buf][kr_ * LDS_ + ((lane - 4) & 15) + wave * 16];                                            \
      fa_[2] = sB[buf][kr_ * LDS_ + ((lane - 8) & 15) + wave * 16];                                            \
      fa_[3] = sB[buf][kr_ * LDS_ + ((lane - 12) & 15) + wave * 16];                                           \
    }                                                                                                          \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) fb_[m] = sA[buf][kr_ * LDS_ + (lane & 15) + m * 16];         \
  }
#define PIPE_MATH(fa_, fb_)                                                                                    \
  {                                                                                                            \
    if (MODE == 0) {                                                                                           \
      _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa_[0], fb_[b], acc[b], 0, 0, 0); \
    } else {                                                                                                   \
      const double f1_ = MODE == 2 ? fa_[1] : rot_groups<0x124>(fa_[0]);                                       \
      const double f2_ = MODE == 2 ? fa_[2] : rot_groups<0x128>(fa_[0]);                                       \
      const double f3_ = MODE == 2 ? fa_[3] : rot_groups<0x12c>(fa_[0]);                                       \
      _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                          \
        acc[b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa_[0], fb_[b], acc[b][0], 0, 0, 0);                    \
        acc[b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(f1_, fb_[b], acc[b][1], 0, 0, 0);                       \
        acc[b][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(f2_, fb_[b], acc[b][2], 0, 0, 0);                       \
        acc[b][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(f3_, fb_[b], acc[b][3], 0, 0, 0);                       \
      }                                                                                                        \
    }                                                                                                          \
  }
  PIPE_GLOAD(0, 0, row0, col0); PIPE_GLOAD(1, 1, row0, col0); PIPE_GLOAD(2, 2, row0, col0); PIPE_GLOAD(3, 3, row0, col0);
  double fa0[4], fb0[4], fa1[4], fb1[4];
  PIPE_LWRITE(0, 0);
  PIPE_GLOAD(0, 4, row0, col0);
  __syncthreads();
  PIPE_LREAD(0, 0, fa0, fb0);
  for (;;) {
    const int n_next = n + stride;
    const bool has_next = n_next < total;
    if (has_next) PIPE_DECODE(n_next, row0n, col0n);
    v4d acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (v4d){0.0, 0.0, 0.0, 0.0};
    double cold[4][4];
    double* Cp = Amat + (long long)col0 * ld + row0;
    // stage c: LDS buffer c & 1 holds it; the staging slot (c + 1) & 3 holds stage c + 1, which goes to the other buffer
    // now; the slot is then reloaded with stage c + 5 (GLOAD5: literal statement, may address the next tile)
#define PIPE_STAGE(c, WRITE_NEXT, GLOAD5)                                                                      \
  {                                                                                                            \
    constexpr int buf = (c) & 1;                                                                               \
    WRITE_NEXT                                                                                                 \
    GLOAD5                                                                                                     \
    PIPE_LREAD(buf, 1, fa1, fb1);                                                                              \
    PIPE_MATH(fa0, fb0);                                                                                       \
    PIPE_LREAD(buf, 2, fa0, fb0);                                                                              \
    PIPE_MATH(fa1, fb1);                                                                                       \
    PIPE_LREAD(buf, 3, fa1, fb1);                                                                              \
    PIPE_MATH(fa0, fb0);                                                                                       \
    __syncthreads();                                                                                           \
    PIPE_LREAD(buf ^ 1, 0, fa0, fb0);                                                                          \
    PIPE_MATH(fa1, fb1);                                                                                       \
  }
    PIPE_STAGE(0, PIPE_LWRITE(1, 1);, PIPE_GLOAD(1, 5, row0, col0);)
    PIPE_STAGE(1, PIPE_LWRITE(0, 2);, PIPE_GLOAD(2, 6, row0, col0);)
    PIPE_STAGE(2, PIPE_LWRITE(1, 3);, PIPE_GLOAD(3, 7, row0, col0);)
    PIPE_STAGE(3, PIPE_LWRITE(0, 0);, if (has_next) PIPE_GLOAD(0, 0, row0n, col0n);
               _Pragma("unroll") for (int b = 0; b < 4; ++b) _Pragma("unroll") for (int r = 0; r < 4; ++r)
                   cold[b][r] = Cp[(long long)(wave * 16 + (lane >> 4) + 4 * r) * ld + b * 16 + (lane & 15)];)
    PIPE_STAGE(4, PIPE_LWRITE(1, 1);, if (has_next) PIPE_GLOAD(1, 1, row0n, col0n);)
    PIPE_STAGE(5, PIPE_LWRITE(0, 2);, if (has_next) PIPE_GLOAD(2, 2, row0n, col0n);)
    PIPE_STAGE(6, PIPE_LWRITE(1, 3);, if (has_next) PIPE_GLOAD(3, 3, row0n, col0n);)
    PIPE_STAGE(7, if (has_next) PIPE_LWRITE(0, 0);, if (has_next) PIPE_GLOAD(0, 4, row0n, col0n);)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cp[(long long)(wave * 16 + (lane >> 4) + 4 * r) * ld + b * 16 + (lane & 15)] =
            cold[b][r] - (MODE == 0 ? acc[b][r] : rot_to_tile(acc[b], r, lane));
    if (!has_next) break;
    n = n_next; row0 = row0n; col0 = col0n;
  }
#undef PIPE_STAGE
}

template <int MODE>
double run_pipe(double* d, const std::vector<double>& init, std::vector<double>& out, int ld, int rows, int grid, int reps) {
  const int row_strips = 2 * rows, col_strips = 2 * rows;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_pipe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds);
  hipMemcpy(d, init.data(), init.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_update_pipe<MODE>, dim3(grid), dim3(256), g_lds, 0, d, ld, 1, 0, row_strips, col_strips);
  hipMemcpy(out.data(), d, init.size() * 8, hipMemcpyDeviceToHost);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(k_update_pipe<MODE>, dim3(grid), dim3(256), g_lds, 0, d, ld, 1, 0, row_strips, col_strips);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 1e3 * ms / reps;
}

template <int RA, int RB, int D>
double run_direct(double* d, const std::vector<double>& init, std::vector<double>& out, int ld, int rows, int grid, int reps) {
  const int col_strips = rows * NB / (32 * RA), row_strips = rows * NB / (32 * RB);
  hipMemcpy(d, init.data(), init.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_update_direct<RA, RB, D>), dim3(grid), dim3(256), 0, 0, d, ld, 1, 0, row_strips, col_strips);
  hipMemcpy(out.data(), d, init.size() * 8, hipMemcpyDeviceToHost);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((k_update_direct<RA, RB, D>), dim3(grid), dim3(256), 0, 0, d, ld, 1, 0, row_strips, col_strips);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 1e3 * ms / reps;
}

template <int MODE, int FLAGS = 0>
double run(double* d, const std::vector<double>& init, std::vector<double>& out, int ld, int rows, int grid, int reps) {
  const int row_strips = 2 * rows, col_strips = 2 * rows;  // the whole lower triangle below panel 0
  const size_t lds = 4 * KC * (64 + 16) * 8 + 64;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update<MODE, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds);
  hipMemcpy(d, init.data(), init.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_update<MODE, FLAGS>), dim3(grid), dim3(256), g_lds, 0, d, ld, 1, 0, row_strips, col_strips);
  hipMemcpy(out.data(), d, init.size() * 8, hipMemcpyDeviceToHost);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((k_update<MODE, FLAGS>), dim3(grid), dim3(256), g_lds, 0, d, ld, 1, 0, row_strips, col_strips);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  (void)lds;
  return 1e3 * ms / reps;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 35;   // 128-row tile rows below the panel
  const int n = (rows + 1) * NB, ld = n + 16;
  std::vector<double> init((size_t)ld * n), o0(init.size()), o1(init.size());
  srand(1);
  for (auto& v : init) v = (rand() % 2001 - 1000) * 1e-3;
  double* d; hipMalloc(&d, init.size() * 8);
  const int tiles = (2 * rows) * (2 * rows + 1) / 2;
  const double gflop = tiles * 64.0 * 64.0 * 128.0 * 2.0 * 1e-9;
  const int grid = 510;
  const double t0 = run<0>(d, init, o0, ld, rows, grid, 20);
  printf("tiles %d  (%.2f GFLOP per launch), grid %d\n", tiles, gflop, grid);
  printf("16x16x4              : %7.2f us  %5.1f TFLOP/s\n", t0, gflop / t0 * 1e3 * 1e-3 * 1e3 / 1e3);
  auto check = [&](const char* name, double us, const std::vector<double>& o) {
    double worst = 0.0;
    for (size_t i = 0; i < o.size(); ++i)
      if ((int)(i % ld) >= (int)(i / ld)) worst = fmax(worst, fabs(o[i] - o0[i]));  // on and below the diagonal
    printf("%-21s: %7.2f us  %5.1f TFLOP/s   max |diff| vs 16x16x4 %.2e\n", name, us, gflop / us * 1e3, worst);
  };
  double t;
  t = run<1>(d, init, o1, ld, rows, grid, 20); check("4x4x4 + dpp", t, o1);
  t = run<2>(d, init, o1, ld, rows, grid, 20); check("4x4x4, lds rotations", t, o1);
  t = run<3>(d, init, o1, ld, rows, grid, 20); check("4x4x4 + dpp, prefetch", t, o1);
  for (int lds : {76000, 50000, 41100}) {
    g_lds = lds;
    printf("-- dynamic LDS request %d bytes\n", lds);
    t = run<0>(d, init, o1, ld, rows, grid, 20); check("16x16x4", t, o1);
    t = run<1>(d, init, o1, ld, rows, grid, 20); check("4x4x4 + dpp", t, o1);
    t = run_pipe<0>(d, init, o1, ld, rows, grid, 20); check("pipelined 16x16x4", t, o1);
    t = run_pipe<1>(d, init, o1, ld, rows, grid, 20); check("pipelined 4x4x4 + dpp", t, o1);
  }
  g_lds = 76000;
  t = run_pipe<0>(d, init, o1, ld, rows, grid, 20); check("pipelined 16x16x4", t, o1);
  t = run_pipe<1>(d, init, o1, ld, rows, grid, 20); check("pipelined 4x4x4 + dpp", t, o1);
  t = run_pipe<2>(d, init, o1, ld, rows, grid, 20); check("pipelined 4x4x4, lds rot", t, o1);
  for (int g : {510}) {
    char nm[64];
    snprintf(nm, sizeof nm, "direct 32x32 D4, grid %d", g);
    t = run_direct<2, 2, 4>(d, init, o1, ld, rows, g, 20); check(nm, t, o1);
    snprintf(nm, sizeof nm, "direct 32x32 D8, grid %d", g);
    t = run_direct<2, 2, 8>(d, init, o1, ld, rows, g, 20); check(nm, t, o1);
  }
  for (int g : {510}) {
    char nm[64];
    snprintf(nm, sizeof nm, "direct 64rx32c D4, grid %d", g);
    t = run_direct<2, 4, 4>(d, init, o1, ld, rows, g, 20); check(nm, t, o1);
  }
  t = run<5>(d, init, o1, ld, rows, grid, 20); check("4x4x4, no rotations (wrong)", t, o1);
  t = run<4>(d, init, o1, ld, rows, grid, 20); check("data path only (wrong)", t, o1);
  t = run<4, 1>(d, init, o1, ld, rows, grid, 20); check("data path, no operand loads", t, o1);
  t = run<4, 2>(d, init, o1, ld, rows, grid, 20); check("data path, no C traffic", t, o1);
  t = run<4, 3>(d, init, o1, ld, rows, grid, 20); check("data path, LDS + barriers only", t, o1);
  t = run<1, 3>(d, init, o1, ld, rows, grid, 20); check("4x4x4 + dpp, LDS only", t, o1);
  t = run<0, 3>(d, init, o1, ld, rows, grid, 20); check("16x16x4, LDS only", t, o1);
  t = run<1, 1>(d, init, o1, ld, rows, grid, 20); check("4x4x4 + dpp, no operand loads", t, o1);
  t = run<1, 2>(d, init, o1, ld, rows, grid, 20); check("4x4x4 + dpp, no C traffic", t, o1);
  return 0;
}
