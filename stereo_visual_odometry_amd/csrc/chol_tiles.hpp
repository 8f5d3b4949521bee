// chol_tiles.hpp -- 16x16 fp64 tiles in LDS for the matrix cores (v_mfma_f64_16x16x4_f64): the pieces the two
// in-LDS Cholesky factorisations share (the 128x128 diagonal blocks of the large reduced system, dense_solver.hip,
// and the whole reduced system of a keyframe window, lm_kernels.hip).
//
// A tile is column-major (element (p, q) at q * 16 + p), so that the 16x4 operand slab of one matrix-core product is
// 64 consecutive doubles.  The lower triangle of a matrix of tiles is packed row by row (ltile).
#pragma once
#include <hip/hip_runtime.h>

namespace hipba {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int TB = 16;
constexpr int kTile = TB * TB;                    // doubles per tile

__device__ __forceinline__ int ltile(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * kTile; }

// wave-uniform broadcast of lane `src`'s double (src is a compile-time constant after unrolling)
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// acc += M * N^T for two column-major 16x16 tiles (negate: acc -= ...)
template <bool NEG>
__device__ __forceinline__ v4d mfma_tile(const double* M, const double* N, v4d acc, int lane) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int o = (kk * 4 + (lane >> 4)) * TB + (lane & 15);
    const double a = NEG ? -M[o] : M[o];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, N[o], acc, 0, 0, 0);
  }
  return acc;
}
// accumulator element r of lane l is D(p, q), p = (l >> 4) + 4 r, q = l & 15
__device__ __forceinline__ v4d tile_load(const double* T, int lane) {
  v4d v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = T[(lane & 15) * TB + (lane >> 4) + 4 * r];
  return v;
}
__device__ __forceinline__ void tile_store(double* T, v4d v, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) T[(lane & 15) * TB + (lane >> 4) + 4 * r] = v[r];
}
__device__ __forceinline__ v4d tile_load_t(const double* T, int lane) {  // loads D^T
  v4d v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = T[((lane >> 4) + 4 * r) * TB + (lane & 15)];
  return v;
}
__device__ __forceinline__ void tile_store_t(double* T, v4d v, int lane) {  // stores D^T
#pragma unroll
  for (int r = 0; r < 4; ++r) T[((lane >> 4) + 4 * r) * TB + (lane & 15)] = v[r];
}

// Cholesky of the 16x16 tile at D (LDS) by one wave, in registers: lane r < 16 holds row r, the pivots and multipliers
// travel between lanes as wave-uniform scalars (v_readlane).  The tile's inverse comes out of the same instructions:
// lanes 16-31 start from the columns of the identity and go through the very same scalings and multiply-subtracts
// (forward substitution wants exactly the multipliers the factorisation broadcasts), so lane 16 + r ends up with
// column r of L^-1.  The inverse replaces the tile in LDS (zero above the diagonal); the factor goes to G (leading
// dimension ld, lower triangle only) unless G is null.  Returns false on a non-positive or non-finite pivot (nothing
// is written then).  All 64 lanes call it.
__device__ __forceinline__ bool tile_cholesky_inverse(double* D, double* G, int ld, int lane) {
  const int r = lane & 15;
  const bool inverse_half = (lane & 16) != 0;
  double x[TB];
#pragma unroll
  for (int c = 0; c < TB; ++c) {
    const double a = D[c * TB + r];
    x[c] = inverse_half ? (c == r ? 1.0 : 0.0) : a;
  }
  bool bad = false;
#pragma unroll
  for (int c = 0; c < TB; ++c) {
    const double d = bcast_lane(x[c], c);
    if (!(d > 0.0) || !isfinite(d)) bad = true;
    // 1/sqrt(d): hardware estimate + two Newton steps (full double precision, no division)
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    x[c] *= inv;
#pragma unroll
    for (int kc = c + 1; kc < TB; ++kc) x[kc] -= x[c] * bcast_lane(x[c], kc);  // lane kc < 16: L(kc, c)
  }
  if (bad) return false;
  if (lane < TB) {
    if (G) {
#pragma unroll
      for (int c = 0; c < TB; ++c)
        if (c <= r) G[(long long)c * ld + r] = x[c];
    }
  } else if (lane < 2 * TB) {
#pragma unroll
    for (int c = 0; c < TB; ++c) D[r * TB + c] = x[c];  // column r of the inverse (zero above the diagonal)
  }
  return true;
}

}  // namespace hipba
