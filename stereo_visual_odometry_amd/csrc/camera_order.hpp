// camera_order.hpp -- fill-reducing order of the poses in the reduced camera system (host code).
//
// The reference hands its normal equations to Ceres' SPARSE_NORMAL_CHOLESKY (src/solve/optimizer.cpp:55), which
// orders the columns before it factors so that the factor stays sparse.  The reduced camera matrix S of this
// engine is factored as a dense matrix by 128-wide panels; what an ordering can buy there is a narrow ENVELOPE: with
// the poses numbered so that co-visible poses are close together, every row of S (and of its Cholesky factor, which
// fills in only inside the envelope) starts late, and the factorisation skips the tiles before that start.  A
// sequence of keyframes is banded in its natural order already; a camera grid or an unordered collection is not.
//
// reverse_cuthill_mckee: the classic bandwidth-reducing order (breadth-first from a pseudo-peripheral node,
// neighbours by increasing degree, reversed).  tile_envelope: first tile column of every 128-row tile of S under an
// order, made monotone (a row tile never starts before the one above it) so that every panel's rows are one range.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace hipba {

// co-visibility of n poses as a symmetric bit matrix
struct PoseGraph {
  int n = 0;
  size_t words = 0;               // 64-bit words per row
  std::vector<uint64_t> bits;
  explicit PoseGraph(int n_) : n(n_), words(((size_t)n_ + 63) / 64), bits(((size_t)n_ + 63) / 64 * (size_t)n_, 0) {}
  void connect(int a, int b) {
    bits[(size_t)a * words + (b >> 6)] |= 1ull << (b & 63);
    bits[(size_t)b * words + (a >> 6)] |= 1ull << (a & 63);
  }
  bool has(int a, int b) const { return (bits[(size_t)a * words + (b >> 6)] >> (b & 63)) & 1ull; }
  // neighbours of a (a itself excluded), ascending
  void neighbours(int a, std::vector<int>& out) const {
    out.clear();
    const uint64_t* row = bits.data() + (size_t)a * words;
    for (size_t w = 0; w < words; ++w) {
      uint64_t m = row[w];
      while (m) {
        const int b = (int)(w * 64) + __builtin_ctzll(m);
        m &= m - 1;
        if (b != a) out.push_back(b);
      }
    }
  }
};

// order[k] = node placed at position k
inline std::vector<int> reverse_cuthill_mckee(const PoseGraph& g) {
  const int n = g.n;
  std::vector<int> start(n + 1, 0), adj;
  {
    std::vector<int> nb;
    for (int v = 0; v < n; ++v) {
      g.neighbours(v, nb);
      start[v + 1] = start[v] + (int)nb.size();
      adj.insert(adj.end(), nb.begin(), nb.end());
    }
  }
  auto degree = [&](int v) { return start[v + 1] - start[v]; };
  // neighbours by increasing degree (ties by index: deterministic)
  for (int v = 0; v < n; ++v)
    std::sort(adj.begin() + start[v], adj.begin() + start[v + 1], [&](int a, int b) {
      const int da = degree(a), db = degree(b);
      return da != db ? da < db : a < b;
    });
  std::vector<int> order;
  order.reserve(n);
  std::vector<int> level(n, -1), queue;
  std::vector<char> placed(n, 0);
  // breadth-first levels from r over the not yet placed nodes; returns the last level's lowest-degree node and depth
  auto bfs_far = [&](int r, int& depth) {
    std::fill(level.begin(), level.end(), -1);
    queue.clear();
    queue.push_back(r);
    level[r] = 0;
    for (size_t h = 0; h < queue.size(); ++h) {
      const int v = queue[h];
      for (int e = start[v]; e < start[v + 1]; ++e) {
        const int u = adj[e];
        if (level[u] < 0 && !placed[u]) { level[u] = level[v] + 1; queue.push_back(u); }
      }
    }
    depth = level[queue.back()];
    int best = queue.back();
    for (size_t h = queue.size(); h-- > 0 && level[queue[h]] == depth;)
      if (degree(queue[h]) < degree(best) || (degree(queue[h]) == degree(best) && queue[h] < best)) best = queue[h];
    return best;
  };
  std::vector<int> by_degree(n);
  std::iota(by_degree.begin(), by_degree.end(), 0);
  std::stable_sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return degree(a) < degree(b); });
  for (int seed : by_degree) {
    if (placed[seed]) continue;
    // pseudo-peripheral start of this component (George & Liu): walk to a far node until the depth stops growing
    int r = seed, depth = -1;
    for (int pass = 0; pass < 8; ++pass) {
      int d2;
      const int far = bfs_far(r, d2);
      if (d2 <= depth) break;
      depth = d2;
      r = far;
    }
    const size_t first = order.size();
    order.push_back(r);
    placed[r] = 1;
    for (size_t h = first; h < order.size(); ++h) {
      const int v = order[h];
      for (int e = start[v]; e < start[v + 1]; ++e) {
        const int u = adj[e];
        if (!placed[u]) { placed[u] = 1; order.push_back(u); }
      }
    }
  }
  std::reverse(order.begin(), order.end());
  return order;
}

// First tile column of every tile row of S (6 rows per pose, tiles of `tile` rows, n_tiles of them) when pose
// order[k] sits at position k; monotone non-decreasing, and never later than the tile just left of the diagonal (so
// that consecutive diagonal blocks always couple and no panel is empty).  cost (optional): multiply-adds of the
// tile-level factorisation inside that envelope, in units of tile^3.
inline std::vector<int> tile_envelope(const PoseGraph& g, const std::vector<int>& order, int tile, int n_tiles,
                                      double* cost = nullptr) {
  const int n = g.n;
  std::vector<int> pos(n);
  for (int k = 0; k < n; ++k) pos[order[k]] = k;
  std::vector<int> first(n_tiles);
  std::iota(first.begin(), first.end(), 0);  // padding rows: diagonal only
  std::vector<int> nb;
  for (int k = 0; k < n; ++k) {
    g.neighbours(order[k], nb);
    int lo = k;
    for (int u : nb) lo = std::min(lo, pos[u]);
    const int col_tile = 6 * lo / tile;
    for (int row : {6 * k, 6 * k + 5}) {
      const int t = row / tile;
      if (t < n_tiles) first[t] = std::min(first[t], col_tile);
    }
  }
  for (int i = 1; i < n_tiles; ++i) first[i] = std::min(first[i], i - 1);
  for (int i = n_tiles - 2; i >= 0; --i) first[i] = std::min(first[i], first[i + 1]);
  if (cost) {
    double c = 0.0;
    for (int i = 0; i < n_tiles; ++i) {
      const double w = i - first[i] + 1;
      c += w * (w - 1) / 2;
    }
    *cost = c;
  }
  return first;
}

}  // namespace hipba
