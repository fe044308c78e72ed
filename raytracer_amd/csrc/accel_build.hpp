// Host-side builder of the ORDERED walk's own tree (f32 fast path, large scenes).
//
// The reference's tree (hittable.rs:159-206: random axis, median split) is what the reference-order
// walk and parity level 1 use, bit for bit.  The ordered walk does not depend on the tree it walks:
// every leaf yields a candidate t that is independent of the bounds it is given, the winner is the
// smallest t, ties go to the leaf with the smaller DFS rank IN THE REFERENCE'S TREE (stored per leaf).
// So the fast path is free to walk a better tree over the same leaves: a binned-SAH tree built here
// from the leaves' verified-conservative world boxes.  On the 10^6-sphere scene it needs ~5x fewer box
// tests per ray than the random-axis tree (whose r = 1000 ground sphere inflates 20 ancestors).
//
// Layout: binary tree, one leaf reference per leaf, nodes in pre-order; a subtree with k leaves has
// k-1 nodes, so node indices are known before recursing and subtrees can be built by threads.
// Depth is capped (the LDS stack holds `max_depth` entries per lane): when the SAH split of a range
// would not leave enough depth for its larger side, the range is split at its median instead.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <future>
#include <vector>

#include "device/scene_view.hpp"

namespace rt {
namespace accel {

struct Box {
  float lo[3], hi[3];
  void clear() {
    for (int k = 0; k < 3; ++k) {
      lo[k] = INFINITY;
      hi[k] = -INFINITY;
    }
  }
  void grow(const float* b) {  // b = lo[3] hi[3]
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], b[k]);
      hi[k] = std::max(hi[k], b[3 + k]);
    }
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], b.lo[k]);
      hi[k] = std::max(hi[k], b.hi[k]);
    }
  }
  float half_area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.0f && dy >= 0.0f && dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

inline uint32_t ceil_log2(uint32_t n) {
  uint32_t d = 0;
  while ((1ull << d) < n) ++d;
  return d;
}

struct Builder {
  const float* leaf_box;      // [n][6] by reference-tree DFS rank
  const uint32_t* leaf_refs;  // [n]
  std::vector<uint32_t> order;  // permutation of leaf ranks, partitioned in place
  std::vector<Node2> nodes;     // n - 1
  static constexpr int BINS = 16;
  static constexpr uint32_t PARALLEL_MIN = 1u << 15;

  // Builds the subtree over order[b, e) (e - b >= 2) at node index `at`, returns its bounds.
  // `budget` = depth (number of node levels) this subtree may use.
  Box build(uint32_t b, uint32_t e, uint32_t at, uint32_t budget, int par_levels) {
    const uint32_t count = e - b;
    Box cb;  // centroid bounds (2 * centroid, to skip the halving)
    cb.clear();
    for (uint32_t i = b; i < e; ++i) {
      const float* bx = leaf_box + 6 * (size_t)order[i];
      for (int k = 0; k < 3; ++k) {
        const float c = bx[k] + bx[3 + k];
        cb.lo[k] = std::min(cb.lo[k], c);
        cb.hi[k] = std::max(cb.hi[k], c);
      }
    }
    uint32_t mid = 0;
    bool have_split = false;
    // a balanced subtree over m leaves needs ceil_log2(m) levels: the larger side must fit in budget - 1
    const uint32_t side_cap = budget >= 2 ? (budget - 1 >= 31 ? 0xFFFFFFFFu : (1u << (budget - 1))) : 1u;
    if (count > 2) {
      float best_cost = INFINITY;
      int best_axis = -1, best_bin = -1;
      for (int axis = 0; axis < 3; ++axis) {
        const float ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
        const float scale = (float)BINS / ext;
        Box bin_box[BINS];
        uint32_t bin_n[BINS];
        for (int i = 0; i < BINS; ++i) {
          bin_box[i].clear();
          bin_n[i] = 0;
        }
        for (uint32_t i = b; i < e; ++i) {
          const float* bx = leaf_box + 6 * (size_t)order[i];
          int bi = (int)((bx[axis] + bx[3 + axis] - cb.lo[axis]) * scale);
          bi = bi < 0 ? 0 : (bi >= BINS ? BINS - 1 : bi);
          bin_box[bi].grow(bx);
          ++bin_n[bi];
        }
        float right_area[BINS];
        uint32_t right_n[BINS];
        Box acc;
        acc.clear();
        uint32_t n_acc = 0;
        for (int i = BINS - 1; i > 0; --i) {
          acc.grow(bin_box[i]);
          n_acc += bin_n[i];
          right_area[i] = acc.half_area();
          right_n[i] = n_acc;
        }
        acc.clear();
        n_acc = 0;
        for (int i = 0; i < BINS - 1; ++i) {  // split after bin i
          acc.grow(bin_box[i]);
          n_acc += bin_n[i];
          const uint32_t nl = n_acc, nr = right_n[i + 1];
          if (nl == 0 || nr == 0) continue;
          if (std::max(nl, nr) > side_cap) continue;  // would overflow the depth budget
          const float cost = acc.half_area() * (float)nl + right_area[i + 1] * (float)nr;
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = axis;
            best_bin = i;
          }
        }
      }
      if (best_axis >= 0) {
        const float scale = (float)BINS / (cb.hi[best_axis] - cb.lo[best_axis]);
        const float lo = cb.lo[best_axis];
        auto first_right = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t r) {
          const float* bx = leaf_box + 6 * (size_t)r;
          int bi = (int)((bx[best_axis] + bx[3 + best_axis] - lo) * scale);
          bi = bi < 0 ? 0 : (bi >= BINS ? BINS - 1 : bi);
          return bi <= best_bin;
        });
        mid = (uint32_t)(first_right - order.begin());
        have_split = mid > b && mid < e;
      }
    }
    if (!have_split) {  // median along the widest centroid axis (also: count == 2, coincident centroids)
      int axis = 0;
      for (int k = 1; k < 3; ++k)
        if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
      mid = b + count / 2;
      std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) {
        const float cx = leaf_box[6 * (size_t)x + axis] + leaf_box[6 * (size_t)x + 3 + axis];
        const float cy = leaf_box[6 * (size_t)y + axis] + leaf_box[6 * (size_t)y + 3 + axis];
        return cx < cy || (cx == cy && x < y);
      });
    }
    const uint32_t nl = mid - b, nr = e - mid;
    const uint32_t left_at = at + 1, right_at = at + nl;  // a k-leaf subtree has k - 1 nodes
    Box lb, rb;
    auto side = [&](uint32_t sb, uint32_t se, uint32_t sat, uint32_t& ref) -> Box {
      if (se - sb == 1) {
        Box x;
        x.clear();
        x.grow(leaf_box + 6 * (size_t)order[sb]);
        ref = leaf_refs[order[sb]];
        return x;
      }
      ref = sat;
      return build(sb, se, sat, budget - 1, par_levels - 1);
    };
    uint32_t lref = 0, rref = 0;
    if (par_levels > 0 && count >= PARALLEL_MIN && nl >= 2 && nr >= 2) {
      auto fut = std::async(std::launch::async, [&] { return side(b, mid, left_at, lref); });
      rb = side(mid, e, right_at, rref);
      lb = fut.get();
    } else {
      lb = side(b, mid, left_at, lref);
      rb = side(mid, e, right_at, rref);
    }
    Node2& n = nodes[at];
    std::memset(&n, 0, sizeof(n));
    for (int k = 0; k < 3; ++k) {
      n.lbox[k] = lb.lo[k];
      n.lbox[3 + k] = lb.hi[k];
      n.rbox[k] = rb.lo[k];
      n.rbox[3 + k] = rb.hi[k];
    }
    n.left = lref;
    n.right = rref;
    lb.grow(rb);
    return lb;
  }
};

// Tree height in node levels (1 = a single node with two leaves).
inline uint32_t height_of(const std::vector<Node2>& nodes, uint32_t root) {
  // pre-order with children after parents: one reverse sweep
  std::vector<uint32_t> h(nodes.size(), 1);
  for (size_t i = nodes.size(); i-- > 0;) {
    uint32_t m = 0;
    for (uint32_t ref : {nodes[i].left, nodes[i].right})
      if (!(ref & REF_LEAF)) m = std::max(m, h[ref]);
    h[i] = m + 1;
  }
  return nodes.empty() ? 0 : h[root];
}

// leaf_box: [n][6] f32 boxes proven to bound each leaf (by DFS rank), leaf_refs: [n] leaf references.
// n >= 2.  Returns the nodes (root = node 0) and the height via *height.
inline std::vector<Node2> build_sah(const float* leaf_box, const uint32_t* leaf_refs, uint32_t n, uint32_t max_depth,
                                    uint32_t* height) {
  Builder bld;
  bld.leaf_box = leaf_box;
  bld.leaf_refs = leaf_refs;
  bld.order.resize(n);
  for (uint32_t i = 0; i < n; ++i) bld.order[i] = i;
  bld.nodes.resize(n - 1);
  const uint32_t budget = std::max(max_depth, ceil_log2(n));
  bld.build(0, n, 0, budget, 3);
  *height = height_of(bld.nodes, 0);
  return std::move(bld.nodes);
}

}  // namespace accel
}  // namespace rt
