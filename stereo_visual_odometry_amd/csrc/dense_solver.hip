// dense_solver.hip -- reduced camera system of ONE large window (6*ncam > 192 unknowns).
//
// The reference hands the whole damped normal-equation system to Ceres'
// SPARSE_NORMAL_CHOLESKY (src/solve/optimizer.cpp:55).  Here the landmarks have already
// been eliminated (lm_kernels.hip), leaving S y = b with S dense-ish and symmetric
// positive definite.  Two solvers:
//   * blocked right-looking Cholesky, fp64, trailing update and panel solve on the
//     matrix cores (v_mfma_f64_16x16x4_f64) -- exact, used when the camera count makes S a
//     true dense contraction;
//   * block-Jacobi preconditioned conjugate gradients straight on the packed block storage.
//
// Storage: P.S holds the upper block triangle, 6x6 blocks of 36 contiguous doubles.
// The factorisation works on a column-major lower-triangular copy A (leading dimension
// npad, a multiple of the 128-wide panel), padded with an identity block.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "chol_tiles.hpp"
#include "hipba_internal.hpp"

namespace hipba {

constexpr int NB = kCholNB;  // 128
typedef double v2d __attribute__((ext_vector_type(2)));  // (arrays of HIP's double2 class are not promoted to registers)

// ---- workspace -----------------------------------------------------------------------------------
static size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr int kPcgScalars = 16;
// leading dimension of the dense matrix: the padded order, one extra tile row for the right-hand side, and a
// skew of kLdSkew doubles so that consecutive columns do not start on the same memory channel
constexpr int kLdSkew = 16;
static size_t dense_ld(size_t npad) { return npad + NB + kLdSkew; }

size_t big_work_bytes(int n, int ncam) {
  const size_t npad = rup((size_t)n, NB);
  size_t b = 0;
  b += rup(dense_ld(npad) * (npad + NB) * 8, 256);  // A, with the extra tile row that carries the right-hand side
  b += 2 * rup((npad / NB) * NB * NB * 8, 256);  // Linv, LinvT
  b += rup(2 * npad * 8, 256);               // rhs, z
  b += rup((6 * npad + kPcgScalars + 3 * (size_t)ncam + 64) * 8, 256);  // pcg vectors + scalars + partials
  b += rup((size_t)36 * ncam * 8, 256);      // Minv
  b += rup((npad / NB + 1) * sizeof(int), 256);  // first
  b += 256;                                      // sync
  b += rup((npad / NB) * (npad / NB + 1) / 2 * sizeof(int2), 256);  // tiles
  b += rup(2 * (npad / NB + 1) * sizeof(unsigned), 256);            // flags, tile counters
  return b + 1024;
}

void big_work_bind(BigWork& W, int n, int ncam, void* base) {
  const size_t npad = rup((size_t)n, NB);
  char* p = static_cast<char*>(base);
  W.n = n; W.npad = (int)npad; W.ld = (int)dense_ld(npad);
  W.A = reinterpret_cast<double*>(p); p += rup(dense_ld(npad) * (npad + NB) * 8, 256);
  W.Linv = reinterpret_cast<double*>(p); p += rup((npad / NB) * NB * NB * 8, 256);
  W.LinvT = reinterpret_cast<double*>(p); p += rup((npad / NB) * NB * NB * 8, 256);
  W.rhs = reinterpret_cast<double*>(p); p += rup(2 * npad * 8, 256);
  W.pcg = reinterpret_cast<double*>(p); p += rup((6 * npad + kPcgScalars + 3 * (size_t)ncam + 64) * 8, 256);
  W.Minv = reinterpret_cast<double*>(p); p += rup((size_t)36 * ncam * 8, 256);
  W.first = reinterpret_cast<int*>(p); p += rup((npad / NB + 1) * sizeof(int), 256);
  W.sync = reinterpret_cast<unsigned*>(p); p += 256;
  W.tiles = reinterpret_cast<int2*>(p); p += rup((npad / NB) * (npad / NB + 1) / 2 * sizeof(int2), 256);
  W.flags = reinterpret_cast<unsigned*>(p);
}

// ---- packed blocks -> dense column-major lower ----------------------------------------------------
__global__ void k_unpack_lower(DevP P, double* A, int ld) {
  if (!P.ws[0].active) return;
  const int ncam = P.wd[0].ncam;
  const int bi = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = ncam - bi;
  if (e >= nb * 36) return;
  const int bj = bi + e / 36, ac = e % 36, a = ac / 6, c = ac % 6;
  const double v = P.S[P.wd[0].S_off + 36ll * tri_index(bi, bi, ncam) + e];
  const int row = 6 * bj + c, col = 6 * bi + a;
  if (row >= col) A[(long long)col * ld + row] = v;
}

// Zeroes the lower triangle of the leading n x n part (one workgroup per column): what a memset of the packed
// matrix is to the unpacking path when the elimination writes its blocks straight into A but leaves some untouched.
// With an envelope (first != null) a column is only zeroed down to the last tile row that reaches it: nothing below
// is ever written or read.
__global__ __launch_bounds__(256) void k_zero_lower(DevP P, double* A, int n, int ld, const int* first, int nt) {
  if (!P.ws[0].active) return;
  const int col = blockIdx.x;
  int end = n;
  if (first) {
    int i = nt - 1;
    while (i > col / NB && first[i] > col / NB) --i;
    end = min(n, (i + 1) * NB);
  }
  double* c = A + (long long)col * ld;
  const int r0 = col & ~1;  // 16-byte aligned start: ld is even; the one word above the diagonal is free to zero
  for (int r = r0 + 2 * threadIdx.x; r < end; r += 512) *reinterpret_cast<double2*>(c + r) = make_double2(0.0, 0.0);
}
hipError_t big_zero_lower(const DevP& P, const BigWork& W, bool enveloped, hipStream_t s) {
  hipLaunchKernelGGL(k_zero_lower, dim3(W.n), dim3(256), 0, s, P, W.A, W.n, W.ld, enveloped ? (const int*)W.first : nullptr,
                     W.npad / NB);
  return hipGetLastError();
}

// Sharded solves exchange the reduced matrix as the 128x128 tiles inside its envelope, packed one after another.
// One workgroup per tile.  kTilesPack / kTilesUnpack copy whole tiles (matrix -> buffer before a one-piece
// all-reduce, buffer -> matrix after it).  kTilesTake / kTilesAdd are the pipelined exchange's pair: the rank's own
// part moves to the buffer and leaves zeros behind, the factorisation starts on those, and the sum over the ranks is
// added column chunk by column chunk (TileFeed).  Those two touch only what belongs to the matrix -- on and below
// the diagonal, inside the leading n x n part -- because the rest of a tile (padding rows, the transposed factor
// above the diagonal of a diagonal tile) is not the elimination's to sum.
__global__ __launch_bounds__(256) void k_env_tiles(DevP P, double* A, int n, int ld, const int2* tiles, int tile0, double* buf,
                                                   int mode) {
  if (!P.ws[0].active) return;
  const int2 tl = tiles[tile0 + blockIdx.x];  // (tile row, tile column)
  double* a = A + (long long)tl.y * NB * ld + (long long)tl.x * NB;
  double* b = buf + (long long)(tile0 + blockIdx.x) * NB * NB;
  const int r = (threadIdx.x & 63) * 2;
  const int row = tl.x * NB + r;
  for (int c = threadIdx.x >> 6; c < NB; c += 4) {
    double2* pa = reinterpret_cast<double2*>(a + (long long)c * ld + r);
    double2* pb = reinterpret_cast<double2*>(b + c * NB + r);
    if (mode == kTilesPack) {
      *pb = *pa;
    } else if (mode == kTilesUnpack) {
      *pa = *pb;
    } else {
      const int col = tl.y * NB + c;
      const bool in0 = row >= col && row < n && col < n, in1 = row + 1 >= col && row + 1 < n && col < n;
      if (mode == kTilesTake) {
        double2 v = *pa;
        if (!in0) v.x = 0.0;
        if (!in1) v.y = 0.0;
        *pb = v;
        if (in0 || in1) {
          double2 z = *pa;  // what is not the matrix's stays as it is
          if (in0) z.x = 0.0;
          if (in1) z.y = 0.0;
          *pa = z;
        }
      } else if (in0 || in1) {
        double2 v = *pa;
        const double2 w = *pb;
        if (in0) v.x += w.x;
        if (in1) v.y += w.y;
        *pa = v;
      }
    }
  }
}
hipError_t big_env_tiles(const DevP& P, const BigWork& W, int tile0, int n_tiles, double* buf, int mode, hipStream_t s) {
  if (n_tiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_env_tiles, dim3(n_tiles), dim3(256), 0, s, P, W.A, W.n, W.ld, (const int2*)W.tiles, tile0, buf, mode);
  return hipGetLastError();
}

// rows beyond n: identity on the padded diagonal, and b^T as row npad -- the first row of one extra tile row.
// Factorising [S; b^T] row-wise like any other row block leaves z^T = (L^-1 b)^T in that row: the forward
// substitution costs no launches of its own.
// "not written yet" in the solution vector of the one-launch backward sweep (a NaN no arithmetic produces: the
// sweep stores the canonical quiet NaN for any NaN of its own)
constexpr unsigned long long kNotYet = 0x7ff8dead7ff8deadull;

// A workgroup takes kPadCols columns and walks their rows [n, ld) with the row index fastest, so that a wave writes
// runs of consecutive words (one thread per column wrote ld * 8 bytes apart: 34 us at 12000 columns against 8).
constexpr int kPadCols = 16;
__global__ __launch_bounds__(256) void k_pad_init(DevP P, double* A, int n, int npad, int ld, unsigned* sync, unsigned* flags,
                                                  double* y_not_yet) {
  if (!P.ws[0].active) return;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < 3) sync[gid] = 0u;  // hand-off words of this factorisation's launches (see k_update_stream)
  // the tile counters of the band updates (behind one word per panel that older sweeps used)
  if (gid < 2 * (npad / NB + 1)) flags[gid] = 0u;
  if (gid < npad && y_not_yet) reinterpret_cast<unsigned long long*>(y_not_yet)[gid] = kNotYet;
  const int col0 = blockIdx.x * kPadCols;
  const int span = ld - n;
  const double* b = P.b + 6ll * P.wd[0].cam_base;
  for (int idx = threadIdx.x; idx < kPadCols * span; idx += blockDim.x) {
    const int col = col0 + idx / span, row = n + idx % span;
    if (col >= npad) break;
    double v = (row == col) ? 1.0 : 0.0;
    if (row == npad && col < n) v = b[col];
    A[(long long)col * ld + row] = v;
  }
}

// z = row npad of the factored matrix
__global__ void k_extract_z(DevP P, const double* A, int npad, int ld, double* z) {
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col < npad) z[col] = A[(long long)col * ld + npad];
}

// ---- diagonal block: Cholesky + explicit inverse of the factor, one workgroup ------------------------
// The 128x128 block is held in LDS as 16x16 tiles (column-major inside a tile, so that the 16x4
// operand slab of v_mfma_f64_16x16x4_f64 is 64 consecutive doubles).  Right-looking over the 8 tile
// columns: wave 0 factors and inverts the 16x16 diagonal tile with scalar code, the panel below it and
// the trailing tiles are updated on the matrix cores.  L^-1 is then built tile column by tile column
// (block forward substitution), two tile columns per wave.  The workgroup that factors block k can ride in the
// launch that updates the rest of the matrix with the previous panel(s): see potrf_block and the FUSE notes below.
constexpr int NTL = NB / TB;                      // 8 tile rows (tile primitives: chol_tiles.hpp)

// LDS of one panel factorisation: 36 lower tiles, the failure word (and 1 KB to spare)
constexpr size_t kPotrfLds = (size_t)(NTL * (NTL + 1) / 2) * kTile * 8 + 64 + NB * 8;

// Factorisation of diagonal block k by the calling workgroup (256 threads, kPotrfLds bytes at smem).
// UPD: the block is first brought up to date with panel columns [kcol0, kcol0 + K) of its own block row,
//   A(k,k) -= L(k, cols) L(k, cols)^T, 64 columns per stage through LDS (the tile area doubles as the stage),
//   so that the workgroup can run beside the kernel that applies the same update to the rest of the matrix.
template <bool UPD>
__device__ __forceinline__ void potrf_block(const DevP& P, double* A, int ld, int k, double* Linv_all, double* LinvT_all,
                                            double* smem, int kcol0, int K) {
  double* Lt = smem;                                   // 36 tiles; the diagonal ones hold their inverses at the end
  int& fail = *reinterpret_cast<int*>(smem + (NTL * (NTL + 1) / 2) * kTile);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double* Akk = A + (long long)k * NB * ld + (long long)k * NB;
  HIPBA_CYCLES_AT(k, 0);
  if (UPD) {
    // wave w owns tile rows w and 7 - w of the lower triangle: nine tiles, accumulated (negated, transposed:
    // register r of lane l is element (l & 15, (l >> 4) + 4 r)) over all stages.  The products are bound by the
    // one CU's matrix cores (about 12 us for K = 128); variants with operand reuse or software pipelining of
    // the LDS reads only added instructions and measured slower.
    v4d acc[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[e] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* pan = A + (long long)kcol0 * ld + (long long)k * NB;  // rows of block k, first panel column
    const int q = t >> 4, pr = t & 15;
    double stage[32];  // thread t carries element (pr, q) of each of the 8 x 4 stage tiles
#define HIPBA_STAGE_LOAD(s_)                                                                     \
  _Pragma("unroll") for (int i = 0; i < NTL; ++i) _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) \
      stage[4 * i + kt] = pan[(long long)(64 * (s_) + TB * kt + q) * ld + TB * i + pr];
    const int nstage = K / 64;
    HIPBA_STAGE_LOAD(0);
    for (int sidx = 0; sidx < nstage; ++sidx) {
      if (sidx) __syncthreads();  // the previous stage's products are done with the tiles
#pragma unroll
      for (int i = 0; i < 32; ++i) Lt[i * kTile + t] = stage[i];
      __syncthreads();
      if (sidx + 1 < nstage) HIPBA_STAGE_LOAD(sidx + 1);
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const int i = e <= wave ? wave : NTL - 1 - wave, j = e <= wave ? e : e - wave - 1;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[e] = mfma_tile<true>(Lt + (4 * j + kt) * kTile, Lt + (4 * i + kt) * kTile, acc[e], lane);
      }
    }
#undef HIPBA_STAGE_LOAD
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const int i = e <= wave ? wave : NTL - 1 - wave, j = e <= wave ? e : e - wave - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = (lane >> 4) + 4 * r, row = lane & 15;
        Lt[ltile(i, j) + col * TB + row] = Akk[(long long)(TB * j + col) * ld + TB * i + row] + acc[e][r];
      }
    }
  } else {
    // load the lower tiles: thread t carries element (row t & 15, column t >> 4) of every tile, so the 36
    // loads are independent and each wave reads 128-byte runs
    const int q = t >> 4, pr = t & 15;
#pragma unroll
    for (int bi = 0; bi < NTL; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) Lt[ltile(bi, bj) + t] = Akk[(long long)(TB * bj + q) * ld + TB * bi + pr];
  }
  if (t == 0) fail = 0;
  __syncthreads();
  HIPBA_CYCLES_AT(k, 1);

  // (a) Cholesky of the 16x16 diagonal tile jd by one wave (tile_cholesky_inverse): the factor goes straight to
  // global memory; LDS keeps the tile's inverse, which is what the panel below, the block inverse and the panel
  // solve's caller need.
  auto diag_tile = [&](int jd) {
    if (!tile_cholesky_inverse(Lt + ltile(jd, jd), Akk + (long long)(TB * jd) * ld + TB * jd, ld, lane) && lane == 0) fail = 1;
  };
  if (wave == 0) diag_tile(0);
  __syncthreads();
  for (int j = 0; j < NTL; ++j) {
    if (j == 2) HIPBA_CYCLES_AT(k, 5);
    if (fail) break;
    // (b) panel below the diagonal tile: L(i,j) = A(i,j) L(j,j)^-T, one matrix-core tile product each with the
    // tile's inverse (the result replaces the tile: every lane has read its operands before the first store)
    {
      const double* Dinv = Lt + ltile(j, j);
      for (int i = j + 1 + wave; i < NTL; i += 4) {
        double* T = Lt + ltile(i, j);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        acc = mfma_tile<false>(Dinv, T, acc, lane);
        tile_store_t(T, acc, lane);
      }
    }
    __syncthreads();
    if (j == 2) HIPBA_CYCLES_AT(k, 6);
    // (c) trailing tiles: A(i2,j2) -= L(i2,j) L(j2,j)^T, again held transposed in the accumulator.  Wave 0 takes
    // the next diagonal tile alone and factors it at once (the longest single-wave job of a step) while the
    // other three waves share the remaining tiles.
    if (wave == 0) {
      if (j + 1 < NTL) {
        double* C = Lt + ltile(j + 1, j + 1);
        v4d acc = tile_load_t(C, lane);
        acc = mfma_tile<true>(Lt + ltile(j + 1, j), Lt + ltile(j + 1, j), acc, lane);
        tile_store_t(C, acc, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        diag_tile(j + 1);
      }
    } else {
      int e = 0;
      for (int i2 = j + 2; i2 < NTL; ++i2) {
        for (int j2 = j + 1; j2 <= i2; ++j2, ++e) {
          if (e % 3 != wave - 1) continue;
          double* C = Lt + ltile(i2, j2);
          v4d acc = tile_load_t(C, lane);
          acc = mfma_tile<true>(Lt + ltile(j2, j), Lt + ltile(i2, j), acc, lane);
          tile_store_t(C, acc, lane);
        }
      }
    }
    __syncthreads();
    if (j == 2) HIPBA_CYCLES_AT(k, 7);
  }
  HIPBA_CYCLES_AT(k, 2);
  if (fail) {
    if (t == 0) P.ws[0].solve_failed = 1;
    return;
  }
  // factor back to global memory (the diagonal tiles went out as they were factored)
  {
    const int q = t >> 4, pr = t & 15;
#pragma unroll
    for (int bi = 0; bi < NTL; ++bi)
#pragma unroll
      for (int bj = 0; bj < bi; ++bj) Akk[(long long)(TB * bj + q) * ld + TB * bi + pr] = Lt[ltile(bi, bj) + t];
  }
  __syncthreads();
  HIPBA_CYCLES_AT(k, 3);
  // (d) X = L^-1 by tile columns; wave w takes columns w and 7 - w.  The tiles of the current column stay in
  // registers: an accumulator tile D(p, q) sits in exactly the lanes and registers the B operand of the next
  // product wants (k = p, n = q), so X(tt, j) feeds y += L(i, tt) X(tt, j) and Y feeds X(i, j) = -Dinv(i) Y
  // without a trip through LDS.
  double* Linv = Linv_all + (long long)k * NB * NB;
  double* LinvT = LinvT_all + (long long)k * NB * NB;
  for (int pass = 0; pass < 2; ++pass) {
    const int j = pass == 0 ? wave : NTL - 1 - wave;
    // tiles above the diagonal are zero
    for (int i = 0; i < j; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
        Linv[(long long)(TB * j + (lane & 15)) * NB + TB * i + (lane >> 4) + 4 * r] = 0.0;
        LinvT[(long long)(TB * i + (lane >> 4) + 4 * r) * NB + TB * j + (lane & 15)] = 0.0;
      }
    }
    v4d xs[NTL];  // X(j + rel, j), rel = 0 .. 7 - j
    xs[0] = tile_load(Lt + ltile(j, j), lane);
    // the same tile transposed, for the store of L^-1 itself: with the two operands of the last product swapped the
    // matrix core delivers X^T, whose lanes hold consecutive ROWS of X (128-byte runs instead of 8-byte stores)
    v4d xt = tile_load_t(Lt + ltile(j, j), lane);
#pragma unroll
    for (int rel = 0; rel < NTL; ++rel) {
      const int i = j + rel;
      if (i < NTL) {
        if (rel > 0) {
          v4d y = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int rt = 0; rt < rel; ++rt) {
            const double* M = Lt + ltile(i, j + rt);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              y = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(kk * 4 + (lane >> 4)) * TB + (lane & 15)], xs[rt][kk], y, 0, 0, 0);
          }
          v4d x = (v4d){0.0, 0.0, 0.0, 0.0};
          xt = (v4d){0.0, 0.0, 0.0, 0.0};
          const double* Di = Lt + ltile(i, i);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double d = -Di[(kk * 4 + (lane >> 4)) * TB + (lane & 15)];
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(d, y[kk], x, 0, 0, 0);
            xt = __builtin_amdgcn_mfma_f64_16x16x4f64(y[kk], d, xt, 0, 0, 0);
          }
          xs[rel] = x;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          Linv[(long long)(TB * j + (lane >> 4) + 4 * r) * NB + TB * i + (lane & 15)] = xt[r];
          LinvT[(long long)(TB * i + (lane >> 4) + 4 * r) * NB + TB * j + (lane & 15)] = xs[rel][r];
        }
      }
    }
  }
  HIPBA_CYCLES_AT(k, 4);
}

__global__ __launch_bounds__(256) void k_potrf_inv(DevP P, double* A, int ld, int k, double* Linv_all,
                                                   double* LinvT_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  potrf_block<false>(P, A, ld, k, Linv_all, LinvT_all, smem, 0, 0);
}

// ---- fp64 matrix-core GEMM, C(i,j) -= sum_k A(i,k) B(j,k), column-major, 128x128 tiles ----------------
// Trailing update A(i,j) -= sum over panel columns [kcol, kcol + K) of L(i,k) L(j,k): a 1-D grid over the tiles
// (i0 + r, j0 + c), c <= r < m, plus row m in full.
//
// The product is computed transposed (D[p][q] = C(i0+q, j0+p)) so that the 16 lanes sharing an
// accumulator register hold 16 consecutive rows of one column: stores are 128-byte runs.
constexpr int KC = 16;            // k-depth of one LDS stage
constexpr int LDT = 128 + 16;     // LDS row pitch (doubles): the two k-groups of a half-wave land 32 banks apart

// FUSE (fuse_linv != null): workgroup 0, which owns the diagonal tile (i0, j0), brings that tile up to
// date itself and goes on to factor it (potrf_block): the next panel factorisation, one workgroup's work on the
// critical path, runs beside this launch's other tiles instead of after them.
constexpr size_t kGemmNtLds = (size_t)4 * KC * LDT * 8;
// ENVELOPE: first[i] is the first tile column of tile row i that can be non-zero (monotone in i): of the
// panel columns [kcol, kcol + K) a tile in row i needs only those from first[i] * NB on -- the factor is zero before
// them.  Tile row m of the grid is the right-hand side (tile row rhs_tile of the matrix), which is dense: its tile in
// column j needs the panel columns from first[j] * NB on.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_nt(DevP P, double* Amat, int ld, int i0, int j0, int kcol, int K,
                                                 int m, double* fuse_linv, double* fuse_linvt, const int* first,
                                                 int rhs_tile) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*sA)[KC * LDT] = reinterpret_cast<double (*)[KC * LDT]>(smem);                 // i-tile, two stages
  double (*sB)[KC * LDT] = reinterpret_cast<double (*)[KC * LDT]>(smem + 2 * KC * LDT);  // j-tile
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  if (fuse_linv && blockIdx.x == 0) {
    const int skip = first ? max(0, first[i0] * NB - kcol) : 0;
    potrf_block<true>(P, Amat, ld, i0, fuse_linv, fuse_linvt, smem, kcol + skip, K - skip);
    return;
  }
  int ti, tj;
  {
    // One workgroup per tile of the lower triangle of an m x m tile grid (row by row), then one per tile of an
    // extra full row m (the right-hand side): a 1-D grid without idle workgroups.  Workgroups go to the eight
    // XCDs round-robin by index, so every XCD gets the same number of tiles whatever m is (a 2-D grid with
    // early exits above the diagonal left some XCDs up to 17 % more tiles than others for odd widths).
    const int id = blockIdx.x, tri = m * (m + 1) / 2;
    int r, c;
    if (id < tri) {
      r = (int)((sqrtf(8.0f * id + 1.0f) - 1.0f) * 0.5f);
      while (r * (r + 1) / 2 > id) --r;
      while ((r + 1) * (r + 2) / 2 <= id) ++r;
      c = id - r * (r + 1) / 2;
    } else {
      r = m; c = id - tri;
    }
    ti = r == m ? rhs_tile : i0 + r; tj = j0 + c;
    if (first) {
      // (the right-hand side's row is dense, but the other factor of its products is tile row tj of the matrix: the
      // columns before that row's envelope are not part of the factor -- nothing ever writes them, not even zeros)
      const int skip = first[r == m ? tj : ti] * NB - kcol;
      if (skip > 0) { kcol += skip; K -= skip; }
      if (K <= 0) return;
    }
  }
  const double* Ap = Amat + (long long)kcol * ld + (long long)ti * NB;   // rows of tile i, panel columns
  const double* Bp = Amat + (long long)kcol * ld + (long long)tj * NB;   // rows of tile j
  const int ldb = ld;
  double* Cp = Amat + (long long)tj * NB * ld + (long long)ti * NB;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // wave's 64x64 quadrant: rows (i) wr, cols (j) wc
  const int lcol = t >> 4, lrow = (t & 15) * 8;  // staging: column of the chunk, first of 8 rows

  v4d acc[4][4];  // [j sub-tile][i sub-tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GLOAD(chunk)                                                                        \
  {                                                                                         \
    const double2* a_ = reinterpret_cast<const double2*>(Ap + (long long)((chunk)*KC + lcol) * ld + lrow);  \
    const double2* b_ = reinterpret_cast<const double2*>(Bp + (long long)((chunk)*KC + lcol) * ldb + lrow); \
    ra0 = a_[0]; ra1 = a_[1]; ra2 = a_[2]; ra3 = a_[3];                                     \
    rb0 = b_[0]; rb1 = b_[1]; rb2 = b_[2]; rb3 = b_[3];                                     \
  }
#define LSTORE(buf)                                                                         \
  {                                                                                         \
    double2* a_ = reinterpret_cast<double2*>(&sA[buf][lcol * LDT + lrow]);                  \
    double2* b_ = reinterpret_cast<double2*>(&sB[buf][lcol * LDT + lrow]);                  \
    a_[0] = ra0; a_[1] = ra1; a_[2] = ra2; a_[3] = ra3;                                     \
    b_[0] = rb0; b_[1] = rb1; b_[2] = rb2; b_[3] = rb3;                                     \
  }
  const int nchunk = K / KC;
  GLOAD(0);
  LSTORE(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) GLOAD(c + 1);
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int krow = (kk * 4 + (lane >> 4)) * LDT + (lane & 15);
      double fa[4], fb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        fa[m] = sB[buf][krow + wc * 64 + m * 16];  // MFMA a-operand: rows of the j-tile
        fb[m] = sA[buf][krow + wr * 64 + m * 16];  // MFMA b-operand: rows of the i-tile
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (c + 1 < nchunk) LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef GLOAD
#undef LSTORE
  // D[p][q]: q = lane & 15 (row i within sub-tile), p = (lane >> 4) + 4 * reg (column j within sub-tile)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = wc * 64 + a * 16 + (lane >> 4) + 4 * r;
        const int row = wr * 64 + b * 16 + (lane & 15);
        Cp[(long long)col * ld + row] -= acc[a][b][r];
      }
    }
  }
}

// ---- the K = 128 steps of the panel chain ------------------------------------------------------------------
// Panel solve L(i,k) = A(i,k) Linv(k)^T, in place, one workgroup per 32 rows of the panel's envelope (the last one
// takes the right-hand side's row).  The launch is short (a hundred-odd strips, 0.5 MFLOP each) and sits on the
// critical path between two panel factorisations; with one LDS stage in flight it was bound by eight back-to-back
// load latencies, so the k-loop is fully unrolled and all eight stages of global loads are in flight from the start.
// The inverse is lower triangular: the 16 columns of sub-tile ct only need the stages c <= ct; the waves take the
// sub-tiles {0, 3, 4, 7} and {1, 2, 5, 6}, 18 stage products each instead of 32.
constexpr int kTrsmRows = 32;  // rows per workgroup: 32 -> twice the workgroups of 64, all stages in flight
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_panel_solve(
    DevP P, double* Amat, int ld, int i0, int kcol, const double* Linv, int strips, int rhs_row0) {
  constexpr int TI = kTrsmRows, TJ = NB;
  constexpr int LDA = TI + 16, LDB = TJ + 16;
  constexpr int PA = TI / 16, PB = TJ / 16;    // doubles per thread and stage
  constexpr int MJ = TJ / 32;                  // 16x16 sub-tiles per wave (one row of them)
  constexpr int NCH = NB / KC;                 // 8 stages, all in flight
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*sA)[KC * LDA] = reinterpret_cast<double (*)[KC * LDA]>(smem);
  double (*sB)[KC * LDB] = reinterpret_cast<double (*)[KC * LDB]>(smem + 2 * KC * LDA);
  // the row strips of the panel's envelope, then the strip of the right-hand side's tile row
  const int row0 = (int)blockIdx.x == strips - 1 ? rhs_row0 : i0 * NB + blockIdx.x * TI;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
#define K128_CTILE(a) (2 * (a) + (((a) ^ wc) & 1))
  const int lcol = t >> 4, lsub = t & 15;
  const double* Ap = Amat + (long long)kcol * ld + row0;
  double* Cp = Amat + (long long)kcol * ld + row0;

  v2d ra[NCH][PA / 2], rb[NCH][PB / 2];
#define K128_GLOAD(chunk)                                                                                         \
  {                                                                                                               \
    const v2d* a_ = reinterpret_cast<const v2d*>(Ap + (long long)((chunk)*KC + lcol) * ld + lsub * PA);           \
    const v2d* b_ = reinterpret_cast<const v2d*>(Linv + (long long)((chunk)*KC + lcol) * NB + lsub * PB);         \
    _Pragma("unroll") for (int m = 0; m < PA / 2; ++m) ra[chunk][m] = a_[m];                                      \
    _Pragma("unroll") for (int m = 0; m < PB / 2; ++m) rb[chunk][m] = b_[m];                                      \
  }
  // (stage indices are written out as literals: with a loop counter the staging registers ended up in scratch)
  K128_GLOAD(0) K128_GLOAD(1) K128_GLOAD(2) K128_GLOAD(3) K128_GLOAD(4) K128_GLOAD(5) K128_GLOAD(6) K128_GLOAD(7)
  const bool live = P.ws[0].active && !P.ws[0].solve_failed;  // uniform; tested after the loads are on their way

  v4d acc[MJ];
#pragma unroll
  for (int a = 0; a < MJ; ++a) acc[a] = (v4d){0.0, 0.0, 0.0, 0.0};
  if (!live) return;

#define K128_STAGE(c)                                                                                             \
  {                                                                                                               \
    constexpr int buf = (c) & 1;                                                                                  \
    {                                                                                                             \
      v2d* a_ = reinterpret_cast<v2d*>(&sA[buf][lcol * LDA + lsub * PA]);                                         \
      v2d* b_ = reinterpret_cast<v2d*>(&sB[buf][lcol * LDB + lsub * PB]);                                         \
      _Pragma("unroll") for (int m = 0; m < PA / 2; ++m) a_[m] = ra[c][m];                                        \
      _Pragma("unroll") for (int m = 0; m < PB / 2; ++m) b_[m] = rb[c][m];                                        \
    }                                                                                                             \
    __syncthreads(); /* stage c visible; everybody is past the reads of stage c - 2, which shared this buffer */  \
    _Pragma("unroll") for (int kk = 0; kk < KC / 4; ++kk) {                                                       \
      const int kr = kk * 4 + (lane >> 4);                                                                        \
      const double fb = sA[buf][kr * LDA + (lane & 15) + wr * (TI / 2)];                                          \
      _Pragma("unroll") for (int a = 0; a < MJ; ++a) {                                                            \
        if ((c) <= K128_CTILE(a)) {                                                                               \
          const double fa = sB[buf][kr * LDB + (lane & 15) + 16 * K128_CTILE(a)];                                 \
          acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[a], 0, 0, 0);                                 \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }
  K128_STAGE(0) K128_STAGE(1) K128_STAGE(2) K128_STAGE(3) K128_STAGE(4) K128_STAGE(5) K128_STAGE(6) K128_STAGE(7)
  static_assert(NCH == 8, "the stages are written out");
#undef K128_STAGE
#undef K128_GLOAD
#pragma unroll
  for (int a = 0; a < MJ; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      Cp[(long long)(16 * K128_CTILE(a) + (lane >> 4) + 4 * r) * ld + wr * (TI / 2) + (lane & 15)] = acc[a][r];
  // L^T goes to the (unused) upper triangle so that the backward substitution reads columns too.  In the
  // accumulators consecutive lanes hold consecutive ROWS of L; each wave turns its RW x 64 block around in LDS,
  // 32 columns at a time, so that consecutive lanes write consecutive columns (256-byte runs, not 8-byte ones).
  constexpr int TP = 33;                      // staging pitch (doubles): odd, no bank conflicts on the way back
  constexpr int RW = TI / 2;                  // rows of a wave's block
  __syncthreads();                            // every wave is done reading the last stage
  double* stg = &sB[0][0] + wave * RW * TP;   // 4 x 4.2 KB inside sB
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg[(lane & 15) * TP + a2 * 16 + (lane >> 4) + 4 * r] = acc[2 * half + a2][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // RW rows x 32 columns: lanes 0-31 take the columns of row rr, lanes 32-63 those of row rr + 1
#pragma unroll
    for (int rr = 0; rr < RW; rr += 2) {
      const int rl = rr + (lane >> 5), cl = lane & 31;
      Amat[(long long)(row0 + wr * RW + rl) * ld + kcol + 16 * K128_CTILE(2 * half + (cl >> 4)) + (cl & 15)] = stg[rl * TP + cl];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#undef K128_CTILE
}

// ---- the K = 128 update of a whole band, streamed ---------------------------------------------------------------
// A(i,j) -= L(i,p) L(j,p)^T on the 64x64 tiles of the row strips [0, row_strips) x column strips [0, col_strips)
// below panel p that lie on or below the diagonal (the last row strip is the right-hand side's), numbered column by
// column.  One launch of at most as many workgroups as the chip holds at two per CU; workgroup b takes tiles b,
// b + (grid - 1), ... and keeps its loads running across the tile boundary: four stages of operands are always in
// flight, those of the next tile are issued while the last stages of this one are multiplied, and the destination
// tile is fetched half a tile ahead of its read-modify-write.
// Workgroup 0 takes tile 0 -- tile (0,0) of the next diagonal block -- waits for the block's other two tiles (numbers
// 1 and row_strips; hand-off through sync[0]) and factors the block (potrf_block): the next panel factorisation, one
// workgroup's work on the critical path, runs beside this launch's other tiles.  It runs 1.5x slower when another
// workgroup multiplies on the same CU (8 tile steps: 87k cycles instead of 59k), so workgroup 0 posts where it runs
// (sync[1]) and a workgroup that finds itself on that CU posts its index (sync[2]) and leaves at once; whoever
// finishes its own tiles first -- the tiles rarely divide evenly -- draws the leaver's from a counter.  (Drawing
// every tile from a counter was tried: a returning atomic among the loads makes the compiler wait for all of them,
// once per tile.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_update_stream(
    DevP P, double* Amat, int ld, int i0, int kcol, double* fuse_linv, double* fuse_linvt, int row_strips,
    int col_strips, int rhs_row0, unsigned* orphan_draw, unsigned* sync, unsigned sync_target) {
  constexpr int TS = 64, LDS_ = TS + 16, PT = TS / 16, NCH = NB / KC;
  static_assert(NCH == 8 && PT == 4, "the stages are written out");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*sA)[KC * LDS_] = reinterpret_cast<double (*)[KC * LDS_]>(smem);
  double (*sB)[KC * LDS_] = reinterpret_cast<double (*)[KC * LDS_]>(smem + 2 * KC * LDS_);
  int* nxt = reinterpret_cast<int*>(smem + 4 * KC * LDS_);
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lcol = t >> 5, lsub = (t & 31) * 2;  // staging: column (and column + 8) of a stage, first of two rows
  const int total = col_strips * row_strips - col_strips * (col_strips - 1) / 2;
  const bool factoring = blockIdx.x == 0;
  // the CU this workgroup runs on, tagged with the launch (the words are not reset between launches)
  const unsigned tag = sync_target << 12;
  const unsigned here = tag | (__smid() & 0xfffu);
  const int stride = (int)gridDim.x - 1;
  int n = blockIdx.x;
  // sync[2], the leaver's word: tag | kPosted | index once a workgroup has left, | kPassed once a workgroup has gone
  // past the point where it looks for a leaver's tiles.  A workgroup may only leave while nobody has passed (one
  // compare-and-swap decides), so whoever passes afterwards -- everybody else -- sees the posting.
  constexpr unsigned kPosted = 1u << 10, kPassed = 1u << 11;
  if (factoring) {
    if (t == 0) {
      __hip_atomic_store(sync + 2, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 1, here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    if (t == 0) {
      unsigned leave = 0;
      // (relaxed on purpose: an acquire here would have every workgroup of the launch invalidate its L2; if sync[2]'s
      // initial value is not visible yet the swap fails and the workgroup simply stays)
      if (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == here) {
        unsigned expect = tag;
        leave = __hip_atomic_compare_exchange_strong(sync + 2, &expect, tag | kPosted | blockIdx.x, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      nxt[0] = (int)leave;
    }
    __syncthreads();
    const bool leave = nxt[0] != 0;
    __syncthreads();  // everybody has read the word before it is written again
    if (leave) return;
  }
  if (n >= total) return;
  const double* pan = Amat + (long long)kcol * ld;
#define STREAM_DECODE(n_, row0_, col0_)                                     \
  {                                                                         \
    int y_ = 0, len_ = row_strips, m_ = (n_);                               \
    while (m_ >= len_) { m_ -= len_; --len_; ++y_; }                        \
    const int x_ = y_ + m_;                                                 \
    row0_ = x_ == row_strips - 1 ? rhs_row0 : i0 * NB + x_ * TS;            \
    col0_ = i0 * NB + y_ * TS;                                              \
  }
  int row0, col0, row0n = 0, col0n = 0;
  STREAM_DECODE(n, row0, col0);
  v2d ra[4][PT / 2], rb[4][PT / 2];
  // staging: half-wave h of the workgroup (8 of them) takes columns h and h + 8 of a 16-column stage, lane l of the
  // half-wave rows 2l and 2l + 1: every load instruction of a wave reads two whole 512-byte columns
#define STREAM_GLOAD(slot, chunk, r0_, c0_)                                                                    \
  {                                                                                                            \
    _Pragma("unroll") for (int m = 0; m < PT / 2; ++m) {                                                       \
      const long long o_ = (long long)((chunk)*KC + 8 * m + lcol) * ld + lsub;                                 \
      ra[slot][m] = *reinterpret_cast<const v2d*>(pan + o_ + (r0_));                                           \
      rb[slot][m] = *reinterpret_cast<const v2d*>(pan + o_ + (c0_));                                           \
    }                                                                                                          \
  }
  STREAM_GLOAD(0, 0, row0, col0); STREAM_GLOAD(1, 1, row0, col0); STREAM_GLOAD(2, 2, row0, col0); STREAM_GLOAD(3, 3, row0, col0);
  bool own = true;  // still on its own tiles (then: on a leaver's)
  for (;;) {
    const int n_next = n + stride;
    const bool has_next = own && !factoring && n_next < total;
    if (has_next) STREAM_DECODE(n_next, row0n, col0n);
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    double cold[2][2][4];
    double* Cp = Amat + (long long)col0 * ld + row0;
    // (stage indices are literals: with a loop counter the staging registers ended up in scratch)
#define STREAM_STAGE(c, AFTER_SYNC)                                                                  \
  {                                                                                                               \
    constexpr int buf = (c) & 1, slot = (c) & 3;                                                                  \
    {                                                                                                             \
      _Pragma("unroll") for (int m = 0; m < PT / 2; ++m) {                                                        \
        *reinterpret_cast<v2d*>(&sA[buf][(8 * m + lcol) * LDS_ + lsub]) = ra[slot][m];                            \
        *reinterpret_cast<v2d*>(&sB[buf][(8 * m + lcol) * LDS_ + lsub]) = rb[slot][m];                            \
      }                                                                                                           \
    }                                                                                                             \
    __syncthreads(); /* stage c visible; everybody is past the reads of the stage before last (same buffer) */    \
    AFTER_SYNC                                                                                                    \
    _Pragma("unroll") for (int kk = 0; kk < KC / 4; ++kk) {                                                       \
      const int kr = kk * 4 + (lane >> 4);                                                                        \
      double fb[2];                                                                                               \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) fb[m] = sA[buf][kr * LDS_ + (lane & 15) + wr * 32 + m * 16];  \
      _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                             \
        const double fa = sB[buf][kr * LDS_ + (lane & 15) + wc * 32 + a * 16];                                    \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                             \
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[b], acc[a][b], 0, 0, 0);                      \
      }                                                                                                           \
    }                                                                                                             \
  }
    STREAM_STAGE(0, STREAM_GLOAD(0, 4, row0, col0);)
    STREAM_STAGE(1, STREAM_GLOAD(1, 5, row0, col0);)
    STREAM_STAGE(2, STREAM_GLOAD(2, 6, row0, col0);)
    STREAM_STAGE(3, STREAM_GLOAD(3, 7, row0, col0);
                 _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)
                 _Pragma("unroll") for (int r = 0; r < 4; ++r)
                     cold[a][b][r] = Cp[(long long)(wc * 32 + a * 16 + (lane >> 4) + 4 * r) * ld + wr * 32 + b * 16 + (lane & 15)];)
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
          Cp[(long long)(wc * 32 + a * 16 + (lane >> 4) + 4 * r) * ld + wr * 32 + b * 16 + (lane & 15)] = cold[a][b][r] - acc[a][b][r];
    // Hand-off of the diagonal block (i0, i0).  Producers: every storing wave drains its stores, the workgroup
    // meets, one lane releases at agent scope and counts itself in.
    if (n == 1 || n == row_strips) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (has_next) { n = n_next; row0 = row0n; col0 = col0n; continue; }
    if (factoring) break;
    // own tiles done: a leaver's, if there is one (its tiles are leaver, leaver + stride, ...)
    own = false;
    __syncthreads();  // the last stage's reads are over: the word below shares nothing with them, but keep it simple
    if (t == 0) {
      int take = total;
      const unsigned o = __hip_atomic_fetch_or(sync + 2, kPassed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((o & ~0xfffu) == tag && (o & kPosted)) {
        const unsigned k = __hip_atomic_fetch_add(orphan_draw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long cand = (long long)(o & (kPosted - 1)) + (long long)k * stride;
        if (cand < total) take = (int)cand;
      }
      nxt[0] = take;
    }
    __syncthreads();
    n = nxt[0];
    __syncthreads();
    if (n >= total) break;
    STREAM_DECODE(n, row0, col0);
    STREAM_GLOAD(0, 0, row0, col0); STREAM_GLOAD(1, 1, row0, col0); STREAM_GLOAD(2, 2, row0, col0); STREAM_GLOAD(3, 3, row0, col0);
  }
#undef STREAM_STAGE
#undef STREAM_GLOAD
#undef STREAM_DECODE
  if (!factoring) return;
  HIPBA_CYCLES_AT(2048 + i0, 1);
  // Consumer: drains its own stores, one lane polls the counter (bounded), acquires at agent scope (which also drops
  // this CU's L1 lines of the block), the workgroup meets, and the factorisation reads the block from memory.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* ok = reinterpret_cast<int*>(smem);
  if (t == 0) {
    int good = 0;
    for (int spin = 0; spin < (1 << 24); ++spin) {
      if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= sync_target) { good = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *ok = good;
    if (!good) { P.ws[0].solve_failed = 1; P.ws[0].handoff_timeouts = 1; }  // never seen: the launch's other workgroups always run
  }
  __syncthreads();
  const int good = *ok;
  HIPBA_CYCLES_AT(2048 + i0, 2);
  __syncthreads();  // everybody has read the word before the factorisation reuses the LDS
  if (good) potrf_block<false>(P, Amat, ld, i0, fuse_linv, fuse_linvt, smem, 0, 0);
}

// ---- backward substitution with the blocked factor (the forward one happens inside the factorisation) ----
// y_k = Linv_k^T z_k ; z_j -= L(k,j)^T y_k for every row block j < k.  L(k,j)^T is read from the upper triangle
// (written by the panel solve), so consecutive threads read consecutive addresses.  Step k is one launch of k
// workgroups (row block j each); the one that owns block k - 1 goes on to form y_(k-1) from its finished z, so
// the next launch finds its y ready and no workgroup repeats that product.  (Walking eight panels inside one
// workgroup was measured too: a single CU streams the factor at ~10 GB/s, 370 us per group instead of 110.)
constexpr int kTrsvThreads = 512;

// y_k = Linv_k^T zk (zk in LDS), by the calling workgroup; result to yk (LDS) and y (global)
__device__ __forceinline__ void trsv_solve_block(const double* LinvT, const double* zk, double (*part)[NB], double* yk,
                                                 double* yout) {
  const int t = threadIdx.x, r = t & (NB - 1), h = t >> 7;  // h = 0..3
  double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
  for (int c = h * 32; c < h * 32 + 32; c += 2) {
    s0 += LinvT[c * NB + r] * zk[c];
    s1 += LinvT[(c + 1) * NB + r] * zk[c + 1];
  }
  part[h][r] = s0 + s1;
  __syncthreads();
  if (t < NB) {
    const double v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    yk[t] = v;
    yout[t] = v;
  }
  __syncthreads();
}

// the bottom block: y_(nt-1) = Linv^T z
__global__ __launch_bounds__(kTrsvThreads) void k_trsv_first(DevP P, int k, const double* LinvT_all, const double* z,
                                                             double* y) {
  __shared__ double zk[NB];
  __shared__ double yk[NB];
  __shared__ double part[4][NB];
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  if (threadIdx.x < NB) zk[threadIdx.x] = z[k * NB + threadIdx.x];
  __syncthreads();
  trsv_solve_block(LinvT_all + (long long)k * NB * NB, zk, part, yk, y + k * NB);
}

__global__ __launch_bounds__(kTrsvThreads) void k_trsv_bwd(DevP P, const double* A, int ld, int k, int j_first,
                                                           const double* LinvT_all, double* z, double* y) {
  __shared__ double zk[NB];
  __shared__ double yk[NB];
  __shared__ double part[4][NB];
  const int t = threadIdx.x, r = t & (NB - 1), h = t >> 7;
  const int j = j_first + blockIdx.x;  // j_first .. k-1: the row blocks above k inside its envelope
  const bool last = j == k - 1;
  // The launch is one link of a chain of dependent launches, so every load that does not depend on another is
  // issued before anything waits: the tile of the factor, the inverse block the last workgroup needs, z and y.
  const double* Ujk = A + (long long)k * NB * ld + (long long)j * NB;  // rows of block j, columns of block k
  const double* Li = LinvT_all + (long long)(k - 1) * NB * NB;
  double u[32], li[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) u[i] = Ujk[(long long)(h * 32 + i) * ld + r];
  if (last) {
#pragma unroll
    for (int i = 0; i < 32; ++i) li[i] = Li[(h * 32 + i) * NB + r];
  }
  double zv = 0.0, yv = 0.0;
  if (t < NB) { zv = z[j * NB + t]; yv = y[k * NB + t]; }
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  if (t < NB) yk[t] = yv;
  __syncthreads();
  {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      s0 += u[i] * yk[h * 32 + i];
      s1 += u[i + 1] * yk[h * 32 + i + 1];
    }
    part[h][r] = s0 + s1;
  }
  __syncthreads();
  if (t < NB) {
    const double v = zv - ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t]));
    z[j * NB + t] = v;
    zk[t] = v;
  }
  if (!last) return;
  __syncthreads();
  {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      s0 += li[i] * zk[h * 32 + i];
      s1 += li[i + 1] * zk[h * 32 + i + 1];
    }
    part[h][r] = s0 + s1;
  }
  __syncthreads();
  if (t < NB) y[(k - 1) * NB + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

// The whole backward sweep in ONE launch: workgroup j (dispatched in the order the sweep needs them, last row
// block first) owns row block j.  It applies z_j -= L(k,j)^T y_k for the row blocks k below it inside the envelope,
// last one first, each as soon as y_k has been published, then forms y_j = Linv_j^T z_j and publishes it.  The
// tile of the next step is already in registers while the workgroup waits, so a link of the chain is one hand-off
// of 1 KB and two 128x128 matrix-vector products instead of a launch.
// Hand-off without flags or fences: y starts out as kNotYet in every word (k_pad_init); the owner stores its 128
// values with write-through (agent-scope atomic) stores, and a reader's 128 lanes poll their own words with
// agent-scope loads until each has seen a value -- one trip through memory per link instead of a drained store, a
// flag and a second read behind it.  Every workgroup is resident at once (the host uses this kernel for at most one
// workgroup per CU), and a workgroup only ever waits for lower-indexed ones.
__global__ __launch_bounds__(kTrsvThreads) void k_trsv_pipeline(DevP P, const double* A, int ld, int nt, int npad,
                                                                const int* first, const double* LinvT_all, double* y) {
  __shared__ double zj[NB];
  __shared__ double yk[NB];
  __shared__ double part[4][NB];
  __shared__ int s_last, s_ok;
  if (!P.ws[0].active || P.ws[0].solve_failed) return;  // the same answer in every workgroup: nobody is left waiting
  const int t = threadIdx.x, r = t & (NB - 1), h = t >> 7;
  const int j = nt - 1 - (int)blockIdx.x;
  if (t == 0) { s_last = j; s_ok = 1; }
  __syncthreads();
  // last row block below j whose envelope reaches back to column block j
  if (t < nt && t > j && (!first || first[t] <= j)) atomicMax(&s_last, t);
  if (t < NB) zj[t] = A[(long long)(j * NB + t) * ld + npad];  // z_j: the right-hand side's row of the factored matrix
  const double* Li = LinvT_all + (long long)j * NB * NB;
  // two register buffers: the tile in use and the one on its way; the inverse block Linv_j takes the second one's
  // place in the last step (and is all there is for the bottom block)
  double u[32], un[32];
#define TRSV_LINV_LOAD(dst) \
  { _Pragma("unroll") for (int i = 0; i < 32; ++i) dst[i] = Li[(h * 32 + i) * NB + r]; }
  __syncthreads();
  const int klast = s_last;
#define TRSV_TILE_LOAD(dst, k_)                                                                   \
  {                                                                                               \
    const double* U_ = A + (long long)(k_) * NB * ld + (long long)j * NB;                         \
    _Pragma("unroll") for (int i = 0; i < 32; ++i) dst[i] = U_[(long long)(h * 32 + i) * ld + r]; \
  }
  if (klast > j) TRSV_TILE_LOAD(u, klast) else TRSV_LINV_LOAD(u);
  for (int k = klast; k > j; --k) {
    if (k - 1 > j) TRSV_TILE_LOAD(un, k - 1) else TRSV_LINV_LOAD(un);  // on its way while this step waits and computes
    if (t < NB) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(y) + k * NB + t;
      unsigned long long bits = kNotYet;
      for (int spin = 0; spin < (1 << 22); ++spin) {
        bits = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits != kNotYet) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (bits == kNotYet) s_ok = 0;
      yk[t] = __longlong_as_double((long long)bits);
    }
    __syncthreads();
    if (!s_ok) break;
    {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        s0 += u[i] * yk[h * 32 + i];
        s1 += u[i + 1] * yk[h * 32 + i + 1];
      }
      part[h][r] = s0 + s1;
    }
    __syncthreads();
    if (t < NB) zj[t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
#pragma unroll
    for (int i = 0; i < 32; ++i) u[i] = un[i];
  }
#undef TRSV_TILE_LOAD
#undef TRSV_LINV_LOAD
  __syncthreads();
  if (!s_ok) {  // never seen: a producer of this launch did not arrive
    if (t == 0) { P.ws[0].solve_failed = 1; P.ws[0].handoff_timeouts = 1; }
    return;
  }
  {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      s0 += u[i] * zj[h * 32 + i];  // (u holds Linv_j^T by now)
      s1 += u[i + 1] * zj[h * 32 + i + 1];
    }
    part[h][r] = s0 + s1;
  }
  __syncthreads();
  if (t < NB) {
    double v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    if (v != v) v = __longlong_as_double(0x7ff8000000000000ll);  // never kNotYet (k_store_solution reports the NaN)
    __hip_atomic_store(y + j * NB + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void k_store_solution(DevP P, const double* y, int n) {
  if (!P.ws[0].active || P.ws[0].solve_failed) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = y[i];
  P.yc[6ll * P.wd[0].cam_base + i] = v;
  if (!isfinite(v)) P.ws[0].solve_failed = 1;
}

constexpr size_t kLdsTrsm = (size_t)(2 * KC * (kTrsmRows + 16) + 2 * KC * (NB + 16)) * 8;
constexpr size_t kLdsK128 = (size_t)(4 * KC * (64 + 16)) * 8;
constexpr size_t kLdsK128Fused = kLdsK128 > kPotrfLds ? kLdsK128 : kPotrfLds;
constexpr size_t kLdsNtFused = kGemmNtLds > kPotrfLds ? kGemmNtLds : kPotrfLds;

// Dynamic-LDS limits of the factorisation kernels.  The attribute belongs to the function ON THE CURRENT DEVICE:
// called by hipba_create for every handle, with the handle's device current.
hipError_t configure_dense_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_potrf_inv),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPotrfLds);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_stream), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsK128Fused);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsTrsm);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsNtFused);
  return e;
}

hipError_t big_solve_cholesky(const DevP& P, const BigWork& W, const int* h_first, hipStream_t s, const TileFeed* feed) {
  const int n = W.n, npad = W.npad, nt = npad / NB;
  const int ld = (int)dense_ld(npad);  // one extra tile row: b^T rides along as row npad (see k_pad_init)
  const int ncam = n / 6;
  // Envelope of the factor by tiles (camera_order.hpp): first[i], monotone, <= i - 1; last[k] = the last tile row
  // panel k reaches.  Without one the matrix is taken as dense.
  std::vector<int> first(nt, 0), last(nt, nt - 1);
  if (h_first) {
    for (int i = 0; i < nt; ++i) first[i] = h_first[i];
    for (int k = 0, i = 0; k < nt; ++k) {
      while (i + 1 < nt && first[i + 1] <= k) ++i;
      last[k] = i < k ? k : i;
    }
  }
  const int* d_first = h_first ? W.first : nullptr;
  if (!P.dense_S)  // otherwise the elimination has written the blocks where they belong
    hipLaunchKernelGGL(k_unpack_lower, dim3((ncam * 36 + 255) / 256, ncam), dim3(256), 0, s, P, W.A, ld);
  // compute units of the device this solve runs on (the stream's: the current one); asked per call, handles of
  // one process may sit on different devices
  int max_resident = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&max_resident, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      max_resident = 0;
  }
  const bool one_launch_sweep = nt <= max_resident;  // as one launch of nt workgroups that hand y_k to each other
  hipLaunchKernelGGL(k_pad_init, dim3((npad + kPadCols - 1) / kPadCols), dim3(256), 0, s, P, W.A, n, npad, ld, W.sync, W.flags,
                     one_launch_sweep ? W.rhs : (double*)nullptr);
  // Two-level right-looking factorisation.  Panels of 128 columns are grouped in blocks of kOuter
  // panels.  Inside a block a freshly factored panel only updates the block's remaining panel columns
  // (narrow updates, K = 128); the rest of the trailing matrix is updated once per block with
  // K = kOuter * 128, which divides the read-modify-write traffic of the trailing tiles by kOuter.
  // Look-ahead by fusion: the workgroup that owns the next diagonal block in an update launch factors it right
  // there, so only the very first panel factorisation is a launch of its own.  (Overlapping through a second
  // stream was measured and lost 5 %: the panel kernels queue behind 50 us trailing-update workgroups.)
  // Every step covers the tile rows of its panel's envelope plus tile row nt, the right-hand side.
  // Block size, dense matrix: with the panel factorisations hidden in the narrow updates, wide blocks win --
  // measured at 94 panels: 8 -> 18.8 ms, 16 -> 17.7, 24 -> 17.7, 32 -> 17.0, 47 -> 19.3 (30 and 34-40: 18.3-19.1).
  // With a narrow envelope the picture changes: a narrow update then touches few enough tiles to finish not long
  // after the panel factorisation it carries, and a wide update -- which hides nothing -- is pure extra time
  // (band of ~35 tiles, 94 panels: blocks of 32 -> 8.7 ms, 48 -> 8.4, one block -> 7.4, before the streamed update).
  // The block size is picked per problem from a model of the two launch kinds fitted to traces on MI355X: a narrow
  // launch takes max(47, 12 + 0.0198 tiles) us for its 64x64 tiles on or below the diagonal (47 us: the fused
  // factorisation), a wide one its multiply-adds at 65 TFLOP/s plus 15 us.
  static const int outer_override = getenv("HIPBA_CHOL_OUTER") ? atoi(getenv("HIPBA_CHOL_OUTER")) : 0;
  auto modelled_us = [&](int outer) {
    double us = 0.0;
    for (int p0 = 0; p0 < nt; p0 += outer) {
      const int p1 = p0 + outer < nt ? p0 + outer : nt;
      for (int p = p0; p + 1 < p1; ++p) {
        const int rows = last[p] - p, cols = (p1 - 1 < last[p] ? p1 - 1 : last[p]) - p;
        double tiles = 0.0;
        for (int y = 0; y < 2 * cols; ++y) tiles += 2 * rows + 1 - y;
        const double t = 12.0 + 0.0198 * tiles;
        us += t > 47.0 ? t : 47.0;
      }
      if (p1 < nt) {
        const int m = last[p1 - 1] - p1 + 1;
        double k_tiles = 0.0;  // 128-deep tile products
        for (int r = 0; r < m; ++r) {
          const int kb = first[p1 + r] > p0 ? first[p1 + r] : p0;
          if (p1 > kb) k_tiles += (double)(p1 - kb) * (r + 1);
        }
        k_tiles += (double)(p1 - p0) * m;  // the right-hand side's row
        us += 15.0 + k_tiles * (2.0 * NB * NB * NB) / 65e6;
      }
    }
    return us;
  };
  int kOuter = outer_override;
  if (kOuter <= 0) {
    kOuter = nt >= 64 ? 32 : (nt >= 24 ? 16 : 8);
    if (h_first) {
      double best = modelled_us(kOuter);
      for (int cand : {8, 16, 32, 48, 64, 96, nt}) {
        if (cand > nt) continue;
        const double t = modelled_us(cand);
        if (t < 0.97 * best) { best = t; kOuter = cand; }
      }
    }
  }
  // Pipelined exchange of a sharded solve: the launches of panel p factor diagonal block p + 1 as well, so every
  // chunk that holds a tile column <= p + 1 is waited for and added before them.
  int fed = 0;
  hipError_t feed_error = hipSuccess;
  auto feed_columns = [&](int upto) {
    while (feed && fed < feed->n_chunks && feed->first_col[fed] <= upto && feed_error == hipSuccess) {
      feed_error = hipStreamWaitEvent(s, feed->ready[fed], 0);
      if (feed_error == hipSuccess)
        feed_error = big_env_tiles(P, W, feed->tile_begin[fed], feed->tile_begin[fed + 1] - feed->tile_begin[fed],
                                   const_cast<double*>(feed->packed), kTilesAdd, s);
      ++fed;
    }
  };
  feed_columns(0);
  hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), kPotrfLds, s, P, W.A, ld, 0, W.Linv, W.LinvT);
  unsigned sync_arrivals = 0;  // two producers per fused band update
  const int resident = 2 * (max_resident > 0 ? max_resident : 256);  // two workgroups per CU
  for (int p0 = 0; p0 < nt; p0 += kOuter) {
    const int p1 = p0 + kOuter < nt ? p0 + kOuter : nt;
    for (int p = p0; p < p1; ++p) {
      feed_columns(p + 1);
      const int rows = last[p] - p;  // tile rows below the diagonal block that panel p reaches
      const int strips = (NB / kTrsmRows) * rows + 1;
      hipLaunchKernelGGL(k_panel_solve, dim3(strips), dim3(256), kLdsTrsm, s, P, W.A, ld, p + 1, p * NB,
                         (const double*)(W.Linv + (long long)p * NB * NB), strips, nt * NB);
      if (p + 1 < p1) {  // bring the block's remaining panel columns up to date with panel p (K = 128); factor p + 1
        const int cols = (p1 - 1 < last[p] ? p1 - 1 : last[p]) - p;
        const int row_strips = 2 * rows + 1, col_strips = 2 * cols;
        const int tiles = col_strips * row_strips - col_strips * (col_strips - 1) / 2;
        hipLaunchKernelGGL(k_update_stream, dim3(tiles < resident ? tiles : resident), dim3(256), kLdsK128Fused, s, P, W.A,
                           ld, p + 1, p * NB, W.Linv, W.LinvT, row_strips, col_strips, nt * NB, W.flags + (nt + 1) + p,
                           W.sync, (sync_arrivals += 2));
      }
    }
    if (p1 < nt) {  // trailing update with the whole block (K = kOuter * 128) inside its envelope; factor panel p1
      const int m = last[p1 - 1] - p1 + 1;
      hipLaunchKernelGGL(k_gemm_nt, dim3(m * (m + 1) / 2 + m), dim3(256), kLdsNtFused, s, P, W.A, ld, p1, p1,
                         p0 * NB, (p1 - p0) * NB, m, W.Linv, W.LinvT, d_first, nt);
    }
  }
  feed_columns(nt);  // (nothing left by now: the last column is due with panel nt - 2)
  if (feed_error != hipSuccess) return feed_error;
  double* z = W.rhs + npad;
  double* y = W.rhs;
  if (one_launch_sweep) {
    // (at most one workgroup per CU: all resident)
    hipLaunchKernelGGL(k_trsv_pipeline, dim3(nt), dim3(kTrsvThreads), 0, s, P, (const double*)W.A, ld, nt, npad, d_first,
                       (const double*)W.LinvT, y);
  } else {
    hipLaunchKernelGGL(k_extract_z, dim3((npad + 255) / 256), dim3(256), 0, s, P, (const double*)W.A, npad, ld, z);
    hipLaunchKernelGGL(k_trsv_first, dim3(1), dim3(kTrsvThreads), 0, s, P, nt - 1, (const double*)W.LinvT, (const double*)z, y);
    for (int k = nt - 1; k >= 1; --k) {
      const int j0 = first[k] < k ? first[k] : k - 1;
      hipLaunchKernelGGL(k_trsv_bwd, dim3(k - j0), dim3(kTrsvThreads), 0, s, P, (const double*)W.A, ld, k, j0,
                         (const double*)W.LinvT, z, y);
    }
  }
  hipLaunchKernelGGL(k_store_solution, dim3((n + 255) / 256), dim3(256), 0, s, P, y, n);
  return hipGetLastError();
}

// =====================================================================================================
// block-Jacobi preconditioned conjugate gradients on the packed upper block triangle
// =====================================================================================================
// scalars: 0 rz, 1 bnorm2, 2 done, 3 iterations, 4 rr
struct PcgView {
  double *x, *r, *z, *p, *q, *scal, *part_pq, *part_rz, *part_rr;
};
__host__ __device__ static PcgView pcg_view(const BigWork& W, int ncam) {
  PcgView v;
  const size_t np = (size_t)W.npad;
  v.x = W.pcg; v.r = v.x + np; v.z = v.r + np; v.p = v.z + np; v.q = v.p + np;
  v.scal = v.q + 2 * np;
  v.part_pq = v.scal + kPcgScalars;
  v.part_rz = v.part_pq + ncam;
  v.part_rr = v.part_rz + ncam;
  return v;
}

// 6x6 SPD inverse by Gauss-Jordan (no pivoting needed for SPD)
__device__ void inv6_spd(const double* Ain, double* out) {
  double a[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) { a[i] = Ain[i]; out[i] = (i % 7 == 0) ? 1.0 : 0.0; }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const double ip = 1.0 / a[7 * c];
#pragma unroll
    for (int j = 0; j < 6; ++j) { a[6 * c + j] *= ip; out[6 * c + j] *= ip; }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = a[6 * r + c];
#pragma unroll
      for (int j = 0; j < 6; ++j) { a[6 * r + j] -= f * a[6 * c + j]; out[6 * r + j] -= f * out[6 * c + j]; }
    }
  }
}

__global__ void k_pcg_init(DevP P, BigWork W, PcgView v) {
  if (!P.ws[0].active) return;
  const int ncam = P.wd[0].ncam;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncam) return;
  const double* D = P.S + P.wd[0].S_off + 36ll * tri_index(c, c, ncam);
  double Mi[36];
  inv6_spd(D, Mi);
  double rz = 0.0, bb = 0.0;
  double rr[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) rr[i] = P.b[6ll * (P.wd[0].cam_base + c) + i];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) z += Mi[6 * i + j] * rr[j];
    v.x[6 * c + i] = 0.0; v.r[6 * c + i] = rr[i]; v.z[6 * c + i] = z; v.p[6 * c + i] = z;
    rz += rr[i] * z; bb += rr[i] * rr[i];
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) W.Minv[36ll * c + i] = Mi[i];
  v.part_rz[c] = rz;
  v.part_rr[c] = bb;
}

__global__ __launch_bounds__(256) void k_pcg_init2(DevP P, PcgView v, int ncam) {
  __shared__ double red[2 * 4];
  if (!P.ws[0].active) return;
  double a[2] = {0.0, 0.0};
  for (int i = threadIdx.x; i < ncam; i += 256) { a[0] += v.part_rz[i]; a[1] += v.part_rr[i]; }
  // fixed-order block reduction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off);
  }
  if (lane == 0) { red[wave * 2] = a[0]; red[wave * 2 + 1] = a[1]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double rz = 0.0, bb = 0.0;
    for (int w = 0; w < 4; ++w) { rz += red[w * 2]; bb += red[w * 2 + 1]; }
    v.scal[0] = rz; v.scal[1] = bb; v.scal[2] = (bb == 0.0) ? 1.0 : 0.0; v.scal[3] = 0.0; v.scal[4] = bb;
  }
}

// q = S p for block row blockIdx.x, partial p.q
__global__ __launch_bounds__(256) void k_pcg_spmv(DevP P, PcgView v) {
  __shared__ double accU[7][36];
  __shared__ double accL[7][36];
  __shared__ double yrow[6];
  if (!P.ws[0].active || v.scal[2] != 0.0) return;
  const int ncam = P.wd[0].ncam;
  const int r = blockIdx.x;
  const int t = threadIdx.x;
  const int grp = t / 36, e = t % 36;
  const double* S = P.S + P.wd[0].S_off;
  double au = 0.0, al = 0.0;
  if (grp < 7) {
    const int ea = e / 6, eb = e % 6;
    for (int c = grp; c < ncam; c += 7) {
      if (c >= r) {
        const double bv = S[36ll * tri_index(r, c, ncam) + e];   // block (r,c)[ea][eb]
        au += bv * v.p[6 * c + eb];
      } else {
        const double bv = S[36ll * tri_index(c, r, ncam) + e];   // block (c,r)[ea][eb] = S(6c+ea, 6r+eb)
        al += bv * v.p[6 * c + ea];
      }
    }
    accU[grp][e] = au;
    accL[grp][e] = al;
  }
  __syncthreads();
  if (t < 6) {
    double s = 0.0;
    for (int g = 0; g < 7; ++g) {
      for (int b = 0; b < 6; ++b) s += accU[g][6 * t + b];   // upper: row ea = t
      for (int a = 0; a < 6; ++a) s += accL[g][6 * a + t];   // lower (transposed): column eb = t
    }
    v.q[6 * r + t] = s;
    yrow[t] = s * v.p[6 * r + t];
  }
  __syncthreads();
  if (t == 0) v.part_pq[r] = yrow[0] + yrow[1] + yrow[2] + yrow[3] + yrow[4] + yrow[5];
}

__device__ __forceinline__ double sum_fixed(const double* a, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += a[i];
  return s;
}

// x += alpha p ; r -= alpha q ; z = Minv r ; partials of r.z and r.r   (one thread per camera)
__global__ __launch_bounds__(256) void k_pcg_update(DevP P, BigWork W, PcgView v, int ncam) {
  __shared__ double s_alpha;
  if (!P.ws[0].active || v.scal[2] != 0.0) return;
  if (threadIdx.x == 0) {
    // every workgroup sums the same partials in the same order
    double pq = 0.0;
    for (int i = 0; i < ncam; ++i) pq += v.part_pq[i];
    s_alpha = v.scal[0] / pq;
  }
  __syncthreads();
  const double alpha = s_alpha;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncam) return;
  double rr[6];
  double rz = 0.0, r2 = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    v.x[6 * c + i] += alpha * v.p[6 * c + i];
    rr[i] = v.r[6 * c + i] - alpha * v.q[6 * c + i];
    v.r[6 * c + i] = rr[i];
    r2 += rr[i] * rr[i];
  }
  const double* Mi = W.Minv + 36ll * c;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) z += Mi[6 * i + j] * rr[j];
    v.z[6 * c + i] = z;
    rz += rr[i] * z;
  }
  v.part_rz[c] = rz;
  v.part_rr[c] = r2;
}

// beta = rz_new / rz ; p = z + beta p ; convergence test |r| <= tol |b|
__global__ __launch_bounds__(256) void k_pcg_dir(DevP P, PcgView v, int ncam, double tol2) {
  __shared__ double s_beta;
  __shared__ int s_done;
  if (!P.ws[0].active || v.scal[2] != 0.0) return;
  if (threadIdx.x == 0) {
    double rz = 0.0, rr = 0.0;
    for (int i = 0; i < ncam; ++i) { rz += v.part_rz[i]; rr += v.part_rr[i]; }
    s_beta = rz / v.scal[0];
    s_done = (rr <= tol2 * v.scal[1]) ? 1 : 0;
  }
  __syncthreads();
  const double beta = s_beta;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * ncam && !s_done) v.p[i] = v.z[i] + beta * v.p[i];
}

// scalars are updated by a single-thread kernel AFTER every workgroup of k_pcg_dir has read them
__global__ void k_pcg_commit(DevP P, PcgView v, int ncam, double tol2) {
  if (!P.ws[0].active || v.scal[2] != 0.0) return;
  double rz = 0.0, rr = 0.0;
  for (int i = 0; i < ncam; ++i) { rz += v.part_rz[i]; rr += v.part_rr[i]; }
  v.scal[0] = rz;
  v.scal[4] = rr;
  v.scal[3] += 1.0;
  if (rr <= tol2 * v.scal[1] || !isfinite(rr)) v.scal[2] = 1.0;
}

__global__ void k_pcg_finish(DevP P, PcgView v, int n) {
  if (!P.ws[0].active) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) P.ws[0].lin_iters = (int)v.scal[3];
  if (i >= n) return;
  const double x = v.x[i];
  P.yc[6ll * P.wd[0].cam_base + i] = x;
  if (!isfinite(x)) P.ws[0].solve_failed = 1;
}

hipError_t big_solve_pcg(const DevP& P, const BigWork& W, int max_iter, double tol, hipStream_t s) {
  const int n = W.n, ncam = n / 6;
  const PcgView v = pcg_view(W, ncam);
  const double tol2 = tol * tol;
  const int gc = (ncam + 255) / 256;
  hipLaunchKernelGGL(k_pcg_init, dim3(gc), dim3(256), 0, s, P, W, v);
  hipLaunchKernelGGL(k_pcg_init2, dim3(1), dim3(256), 0, s, P, v, ncam);
  double done = 0.0;
  for (int it = 0; it < max_iter; ++it) {
    hipLaunchKernelGGL(k_pcg_spmv, dim3(ncam), dim3(256), 0, s, P, v);
    hipLaunchKernelGGL(k_pcg_update, dim3(gc), dim3(256), 0, s, P, W, v, ncam);
    hipLaunchKernelGGL(k_pcg_dir, dim3((6 * ncam + 255) / 256), dim3(256), 0, s, P, v, ncam, tol2);
    hipLaunchKernelGGL(k_pcg_commit, dim3(1), dim3(1), 0, s, P, v, ncam, tol2);
    if ((it % 16) == 15) {
      hipError_t e = hipMemcpyAsync(&done, v.scal + 2, sizeof(double), hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return e;
      e = hipStreamSynchronize(s);
      if (e != hipSuccess) return e;
      if (done != 0.0) break;
    }
  }
  hipLaunchKernelGGL(k_pcg_finish, dim3((n + 255) / 256), dim3(256), 0, s, P, v, n);
  return hipGetLastError();
}

}  // namespace hipba
