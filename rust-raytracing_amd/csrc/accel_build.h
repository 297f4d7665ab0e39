// Host side: a tree of the library's own for the single-precision per-lane walk.
//
// The tree the caller hands over is the reference's (src/bvh.rs:19-76: a random axis per node, median split, boxes as the
// reference computes them, the same object inserted many times where the scene does that).  The double-precision kernels walk
// exactly that tree, because the order of its leaves decides ties and its boxes decide what a ray can reach.  The closest hit
// itself does not depend on the tree: any hierarchy of conservative boxes over the same objects finds the same surface, except
// that among exactly coincident surfaces another one may be reported.  The single-precision kernels (compared with the
// reference by tolerance, not bit for bit) therefore walk a tree built here:
//   * objects that are exact copies of one another (same wrapper, same shape, materials equal by value) are reduced to the copy
//     the reference's walk would report, the last one it visits - the `random` scene inserts its three large spheres 22 times;
//   * boxes are tight boxes of the shapes over the shutter interval (a negative radius does not invert them);
//   * the split is chosen by the surface-area heuristic over 16 bins per axis.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <thread>
#include <unordered_map>
#include <vector>

#include "context.h"

namespace rti {

struct Box3d {
  double mn[3], mx[3];
  void reset() { for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<double>::infinity(); mx[k] = -mn[k]; } }
  void grow(const Box3d& b) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); } }
  double area() const {
    double d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0;
    return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
  }
};

class FastTreeBuilder {
 public:
  explicit FastTreeBuilder(const rt_scene_desc& s) : s_(s) {}

  // The main tree once more, four children to a node, root first and level by level - filled by build() when no nested tree is
  // left (every one was dissolved).  `wide_stack_need`: entries the deepest walk can have on its stack.
  std::vector<WideTmp> wide;
  int wide_stack_need = 0;
  // Leaves left out of the four-wide form: objects as large as the scene (a fog that fills it, the ground sphere).  Nearly every
  // ray has to test them, and inside the tree each lane would reach them at a moment of its own; the walk tests them first, for
  // all its lanes at once (DeviceScene::pre), which also gives it a first `closest` to prune with.  Leaf words, at most kWidePre.
  std::vector<uint32_t> wide_pre;

  // `ref` are the links of the reference trees before reordering, `ref_root` / `ref_sub_root` their roots.
  // Fills `links` (top-level tree first, then the nested ones), `root` and `sub_root`.
  // `dissolved[o]` != 0: the nested tree held by object o is not built; its objects - which the caller has given o's own
  // Translate / RotateY - join the main tree as ordinary leaves.  Translate(RotateY(tree of shapes)) reports the closest of
  // Translate(RotateY(shape)) over its shapes, record for record (both wrappers keep t), so only tie order can differ.
  void build(const std::vector<TmpLink>& ref, int ref_root, const std::vector<int>& ref_sub_root, const std::vector<char>& dissolved,
             std::vector<TmpLink>& links, int& root, std::vector<int>& sub_root) {
    links.clear();
    sub_root.assign((size_t)s_.n_objects, LINK_END);
    nested_box_.assign((size_t)s_.n_objects, Box3d{});
    // nested trees first: their root boxes are the geometry boxes of the objects that hold them
    std::vector<std::vector<TmpLink>> nested_links;
    std::vector<int> nested_owner;
    for (int o = 0; o < s_.n_objects; ++o) {
      if (!(s_.objects[o].flags & RT_OBJ_SUB_BVH) || dissolved[(size_t)o]) continue;
      std::vector<int> objs = leaves_in_visiting_order(ref, ref_sub_root[(size_t)o], s_.n_objects);
      nested_links.emplace_back();
      nested_owner.push_back(o);
      Box3d rb;
      emit_tree(objs, (int)nested_links.size(), nested_links.back(), rb);
      nested_box_[(size_t)o] = rb;
    }
    std::vector<int> top;
    for (int o : leaves_in_visiting_order(ref, ref_root, s_.n_objects)) {
      if (!dissolved[(size_t)o]) { top.push_back(o); continue; }
      for (int child : leaves_in_visiting_order(ref, ref_sub_root[(size_t)o], s_.n_objects)) top.push_back(child);
    }
    Box3d unused;
    want_wide_ = nested_links.empty() && !std::getenv("RTAMD_NO_WIDE");
    emit_tree(top, 0, links, unused);
    root = 0;
    for (size_t k = 0; k < nested_links.size(); ++k) {
      const int base = (int)links.size();
      for (TmpLink l : nested_links[k]) {
        if (!(l.hit & LEAF_FLAG)) l.hit += (uint32_t)base;
        if (l.miss >= 0) l.miss += base;
        links.push_back(l);
      }
      sub_root[(size_t)nested_owner[k]] = base;
    }
  }

 private:
  const rt_scene_desc& s_;
  std::vector<Box3d> nested_box_;
  bool want_wide_ = false;
  std::vector<char> twice_;  // objects whose leaf the reference's tree visits twice and whose second visit matters (media)
  struct Node {
    Box3d box;
    int left = -1, right = -1, obj = -1;
    int n_links = 0;  // entries the subtree takes in the threaded list
  };

  // Objects of one reference tree, each once, ordered by the LAST time the reference's walk visits them.
  std::vector<int> leaves_in_visiting_order(const std::vector<TmpLink>& ref, int start, int n_objects) {
    std::vector<int> seq;
    if (twice_.empty()) twice_.assign((size_t)n_objects, 0);
    for (int i = start; i >= 0;) {
      const TmpLink& l = ref[(size_t)i];
      if (l.hit & LEAF_FLAG) {
        seq.push_back((int)(l.hit & LEAF_INDEX_MASK));
        if (l.hit & LEAF_SECOND) twice_[(size_t)(l.hit & LEAF_INDEX_MASK)] = 1;  // a medium the reference's tree visits twice
        i = l.miss;
      }
      else i = (int)l.hit;
    }
    std::vector<int> out;
    std::vector<char> seen((size_t)n_objects, 0);
    for (size_t k = seq.size(); k-- > 0;)
      if (!seen[(size_t)seq[k]]) { seen[(size_t)seq[k]] = 1; out.push_back(seq[k]); }
    std::reverse(out.begin(), out.end());
    return out;
  }

 public:
  // Everything that decides what a ray sees of an object, by value (compared bit for bit).
  struct Signature {
    double v[28];
    bool operator==(const Signature& o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
  };
  struct SignatureHash {
    size_t operator()(const Signature& g) const {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (double d : g.v) {
        uint64_t b;
        std::memcpy(&b, &d, 8);
        h = (h ^ b) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
      }
      return (size_t)h;
    }
  };
  Signature signature(int o) const {
    const rt_object& ob = s_.objects[o];
    Signature g{};
    int n = 0;
    auto push = [&](double x) { g.v[n++] = x; };
    push((double)ob.flags);
    for (int k = 0; k < 3; ++k) push(ob.offset[k]);
    push(ob.sin_theta); push(ob.cos_theta);
    const rt_shape& sh = s_.shapes[ob.geom];
    push((double)sh.tag);
    for (int k = 0; k < 9; ++k) push(sh.p[k]);
    const rt_material& m = s_.materials[sh.material];
    push((double)m.tag);
    for (int k = 0; k < 3; ++k) push(m.albedo[k]);
    push(m.fuzz); push(m.ir);
    if (m.texture >= 0 && m.texture < s_.n_textures) {
      const rt_texture& t = s_.textures[m.texture];
      push((double)t.tag);
      if (t.tag == RT_TEX_SOLID) for (int k = 0; k < 3; ++k) push(t.color[k]);
      else push((double)m.texture);  // other textures: the same record only
    } else {
      push(-1.0);
    }
    return g;  // 26 values at most; the rest stays zero
  }
  // Objects that may stand in for an exact copy of themselves: bare or wrapped shapes, but no medium (it draws random numbers per
  // visit), no nested tree, and no sphere of negative radius (what reaches it is decided by the boxes around each copy).
  bool copy_eligible(int o) const {
    const rt_object& ob = s_.objects[o];
    return !(ob.flags & (RT_OBJ_MEDIUM | RT_OBJ_SUB_BVH)) && !has_guard(o);
  }
  // out[o] = 1 for every object of `order` (objects in the order a walk visits them) that is an exact copy of an object coming
  // later in it.  `eligible(o)` says which objects take part.  Open-addressing table of (hash, last object with that signature).
  // `sort_pairs` (optional): sorts (hash, position) pairs by hash, stably - the device's radix sort for scenes of a million objects,
  // where filling a hash table from one thread is most of this search.
  std::function<bool(uint64_t*, uint32_t*, size_t)> sort_pairs;
  template <class Pred> std::vector<char> earlier_copies(const std::vector<int>& order, Pred eligible) const {
    size_t sort_min = 65536;  // RTAMD_SORT_MIN: another threshold (tests run the sorted search on small scenes)
    if (const char* e = std::getenv("RTAMD_SORT_MIN")) sort_min = (size_t)std::max(1, atoi(e));
    if (sort_pairs && order.size() >= sort_min) {
      std::vector<uint64_t> key(order.size());
      std::vector<uint32_t> pos(order.size());
      parallel_chunks(order.size(), [&](size_t, size_t lo, size_t hi) {
        for (size_t q = lo; q < hi; ++q) {
          // objects that take no part sort to the end (their own hash value could collide with that marker; they are skipped by position)
          key[q] = eligible(order[q]) ? ((uint64_t)SignatureHash()(signature(order[q])) >> 1) : ~0ull;
          pos[q] = (uint32_t)q;
        }
      });
      if (sort_pairs(key.data(), pos.data(), key.size())) {
        std::vector<char> out((size_t)s_.n_objects, 0);
        for (size_t a = 0; a < key.size();) {
          size_t b = a + 1;
          while (b < key.size() && key[b] == key[a]) ++b;
          if (key[a] != ~0ull && b - a > 1) {
            // equal hashes (positions ascending: the sort is stable): an object is an earlier copy when the LAST entry of the run
            // with its signature is another object (the walk may visit an object's own leaf more than once)
            for (size_t i = a; i < b; ++i) {
              const Signature g = signature(order[pos[i]]);
              size_t last = i;
              for (size_t j = i + 1; j < b; ++j)
                if (signature(order[pos[j]]) == g) last = j;
              if (order[pos[last]] != order[pos[i]]) out[(size_t)order[pos[i]]] = 1;
            }
          }
          a = b;
        }
        return out;
      }
    }
    size_t cap = 16;
    while (cap < 2 * order.size()) cap <<= 1;
    struct Slot { uint64_t hash; int obj; };
    std::vector<Slot> table(cap, Slot{0, -1});
    std::vector<int> slot_of((size_t)s_.n_objects, -1);
    // the hashes of all signatures first, on threads (most of the time of this search for a million objects); the table is then
    // filled in walk order by one thread
    std::vector<uint64_t> hash_of(order.size());
    {
      unsigned hw = std::thread::hardware_concurrency();
      if (const char* e = std::getenv("OMP_NUM_THREADS")) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)std::max(1, atoi(e)));
      const size_t n_threads = order.size() < 65536 ? 1 : std::max(1u, std::min(hw, 16u));
      std::vector<std::thread> pool;
      auto work = [&](size_t lo, size_t hi) {
        for (size_t q = lo; q < hi; ++q) hash_of[q] = eligible(order[q]) ? (uint64_t)SignatureHash()(signature(order[q])) : 0ull;
      };
      const size_t per = (order.size() + n_threads - 1) / n_threads;
      for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work, std::min(order.size(), t * per), std::min(order.size(), (t + 1) * per));
      work(0, std::min(order.size(), per));
      for (std::thread& t : pool) t.join();
    }
    for (size_t q = 0; q < order.size(); ++q) {
      const int o = order[q];
      if (!eligible(o)) continue;
      const Signature g = signature(o);
      const uint64_t h = hash_of[q];
      size_t k = (size_t)h & (cap - 1);
      while (table[k].obj >= 0 && !(table[k].hash == h && signature(table[k].obj) == g)) k = (k + 1) & (cap - 1);
      table[k] = Slot{h, o};  // a later copy takes the place of an earlier one
      slot_of[(size_t)o] = (int)k;
    }
    std::vector<char> out((size_t)s_.n_objects, 0);
    for (int o : order)
      if (eligible(o) && table[(size_t)slot_of[(size_t)o]].obj != o) out[(size_t)o] = 1;
    return out;
  }

  // For the device-side builder (gpu_build.hip), which takes one box and one leaf word per object: what this builder would give
  // the object as a leaf.  Only for scenes without nested trees, media and spheres of negative radius.
  Box3d leaf_box(int o) const { return object_box(o); }
  uint32_t word_of(int o) const { return leaf_word(o); }

 private:

  // A sphere of negative radius (the hollow-glass trick, src/scenes.rs:235-239) reports an inverted box (src/sphere.rs:75-81),
  // which barely widens its parent's box in the reference's tree: there the sphere is only ever tested by rays that pass the box
  // of the node holding it, and rays that miss that box see through the shell.  The tree built here reproduces that: such a
  // sphere stands behind a box entry of its own carrying the box of its node in the reference's tree.
  bool has_guard(int o) const {
    const rt_object& ob = s_.objects[o];
    if (ob.flags & (RT_OBJ_SUB_BVH | RT_OBJ_MEDIUM)) return false;
    const rt_shape& sh = s_.shapes[ob.geom];
    return (sh.tag == RT_SHAPE_SPHERE && sh.p[3] < 0.0) || (sh.tag == RT_SHAPE_MOVING_SPHERE && sh.p[6] < 0.0);
  }
  Box3d guard_box(int o) const {
    Box3d b;
    b.reset();
    for (int n = 0; n < s_.n_nodes; ++n)
      if (s_.nodes[n].left == ~o || s_.nodes[n].right == ~o) {
        Box3d nb;
        for (int k = 0; k < 3; ++k) { nb.mn[k] = s_.nodes[n].bmin[k]; nb.mx[k] = s_.nodes[n].bmax[k]; }
        b.grow(nb);
      }
    return b;
  }

  static Box3d sphere_box(const double c[3], double r) {
    r = std::fabs(r);
    Box3d b;
    for (int k = 0; k < 3; ++k) { b.mn[k] = c[k] - r; b.mx[k] = c[k] + r; }
    return b;
  }
  Box3d shape_box(const rt_shape& sh) const {
    const double pad = 1e-4;
    Box3d b;
    switch (sh.tag) {
      case RT_SHAPE_SPHERE: return sphere_box(sh.p, sh.p[3]);
      case RT_SHAPE_MOVING_SPHERE: {  // centre0 p[0..2], centre1 p[3..5], radius p[6], time0 p[7], time1 p[8]
        b.reset();
        // linear motion: the extremes of the time a ray can carry - the camera's shutter interval, and 0, which RotateY gives its
        // local ray (src/hittable.rs:214-223)
        for (double time : {std::min(0.0, s_.camera.shutter_open_time), std::max(1.0, s_.camera.shutter_close_time)}) {
          double f = (time - sh.p[7]) / (sh.p[8] - sh.p[7]);
          double c[3] = {sh.p[0] + f * (sh.p[3] - sh.p[0]), sh.p[1] + f * (sh.p[4] - sh.p[1]), sh.p[2] + f * (sh.p[5] - sh.p[2])};
          b.grow(sphere_box(c, sh.p[6]));
        }
        return b;
      }
      case RT_SHAPE_RECT_XY: b = Box3d{{sh.p[0], sh.p[2], sh.p[4] - pad}, {sh.p[1], sh.p[3], sh.p[4] + pad}}; return b;
      case RT_SHAPE_RECT_XZ: b = Box3d{{sh.p[0], sh.p[4] - pad, sh.p[2]}, {sh.p[1], sh.p[4] + pad, sh.p[3]}}; return b;
      case RT_SHAPE_RECT_YZ: b = Box3d{{sh.p[4] - pad, sh.p[0], sh.p[2]}, {sh.p[4] + pad, sh.p[1], sh.p[3]}}; return b;
      default: b = Box3d{{sh.p[0], sh.p[1], sh.p[2]}, {sh.p[3], sh.p[4], sh.p[5]}}; return b;
    }
  }
  Box3d object_box(int o) const {
    const rt_object& ob = s_.objects[o];
    Box3d g = (ob.flags & RT_OBJ_SUB_BVH) ? nested_box_[(size_t)o] : shape_box(s_.shapes[ob.geom]);
    if (ob.flags & RT_OBJ_ROTATE_Y) {  // world x = cos * x + sin * z, world z = -sin * x + cos * z (src/hittable.rs:229-232)
      Box3d r;
      r.reset();
      for (int i = 0; i < 8; ++i) {
        double x = (i & 1) ? g.mx[0] : g.mn[0], y = (i & 2) ? g.mx[1] : g.mn[1], z = (i & 4) ? g.mx[2] : g.mn[2];
        double c[3] = {ob.cos_theta * x + ob.sin_theta * z, y, -ob.sin_theta * x + ob.cos_theta * z};
        Box3d p{{c[0], c[1], c[2]}, {c[0], c[1], c[2]}};
        r.grow(p);
      }
      g = r;
    }
    if (ob.flags & RT_OBJ_TRANSLATE)
      for (int k = 0; k < 3; ++k) { g.mn[k] += ob.offset[k]; g.mx[k] += ob.offset[k]; }
    const double eps = 1e-6;  // the single-precision box test rounds: keep a margin of a few ulps at scene scale
    for (int k = 0; k < 3; ++k) {
      double m = eps * std::max({1.0, std::fabs(g.mn[k]), std::fabs(g.mx[k])});
      g.mn[k] -= m; g.mx[k] += m;
    }
    return g;
  }

  // A subtree over k objects has 2k - 1 nodes: the node of ids[lo, hi) sits at `me`, its first subtree right behind it and the
  // second one behind that, so disjoint subtrees write disjoint parts of `nodes` and the top levels can be split over threads.
  void build_node(std::vector<Node>& nodes, std::vector<int>& ids, const std::vector<Box3d>& boxes, int lo, int hi, int me, int spawn_depth) {
    Box3d bb, cb;
    bb.reset(); cb.reset();
    for (int k = lo; k < hi; ++k) {
      const Box3d& b = boxes[(size_t)ids[(size_t)k]];
      bb.grow(b);
      Box3d c;
      for (int a = 0; a < 3; ++a) c.mn[a] = c.mx[a] = 0.5 * (b.mn[a] + b.mx[a]);
      cb.grow(c);
    }
    nodes[(size_t)me].box = bb;
    if (hi - lo == 1) {
      const int o = ids[(size_t)lo];
      nodes[(size_t)me].obj = o;
      nodes[(size_t)me].n_links = 1 + (has_guard(o) ? 1 : 0) + ((!twice_.empty() && twice_[(size_t)o]) ? 1 : 0);
      return;
    }
    constexpr int kBins = 16;
    double best_cost = std::numeric_limits<double>::infinity();
    int best_axis = -1, best_split = -1;
    for (int axis = 0; axis < 3; ++axis) {
      const double ext = cb.mx[axis] - cb.mn[axis];
      if (!(ext > 0.0)) continue;
      Box3d bin_box[kBins];
      int bin_n[kBins] = {0};
      for (auto& b : bin_box) b.reset();
      const double scale = kBins / ext;
      for (int k = lo; k < hi; ++k) {
        const Box3d& b = boxes[(size_t)ids[(size_t)k]];
        int bi = std::min(kBins - 1, std::max(0, (int)((0.5 * (b.mn[axis] + b.mx[axis]) - cb.mn[axis]) * scale)));
        bin_box[bi].grow(b);
        bin_n[bi]++;
      }
      double right_area[kBins];
      int right_n[kBins];
      Box3d acc;
      acc.reset();
      int n = 0;
      for (int b = kBins - 1; b > 0; --b) { acc.grow(bin_box[b]); n += bin_n[b]; right_area[b] = acc.area(); right_n[b] = n; }
      acc.reset();
      n = 0;
      for (int b = 0; b + 1 < kBins; ++b) {
        acc.grow(bin_box[b]);
        n += bin_n[b];
        if (n == 0 || right_n[b + 1] == 0) continue;
        double cost = acc.area() * n + right_area[b + 1] * right_n[b + 1];
        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = b; }
      }
    }
    int mid;
    if (best_axis < 0) {
      mid = (lo + hi) / 2;  // all centroids coincide
    } else {
      const double ext = cb.mx[best_axis] - cb.mn[best_axis];
      const double scale = kBins / ext;
      auto it = std::partition(ids.begin() + lo, ids.begin() + hi, [&](int id) {
        const Box3d& b = boxes[(size_t)id];
        int bi = std::min(kBins - 1, std::max(0, (int)((0.5 * (b.mn[best_axis] + b.mx[best_axis]) - cb.mn[best_axis]) * scale)));
        return bi <= best_split;
      });
      mid = (int)(it - ids.begin());
      if (mid == lo || mid == hi) mid = (lo + hi) / 2;
    }
    int l = me + 1, r = me + 2 * (mid - lo);
    if (spawn_depth > 0 && hi - lo >= 8192) {
      std::thread other([&, r] { build_node(nodes, ids, boxes, mid, hi, r, spawn_depth - 1); });
      build_node(nodes, ids, boxes, lo, mid, l, spawn_depth - 1);
      other.join();
    } else {
      build_node(nodes, ids, boxes, lo, mid, l, 0);
      build_node(nodes, ids, boxes, mid, hi, r, 0);
    }
    nodes[(size_t)me].n_links = 1 + nodes[(size_t)l].n_links + nodes[(size_t)r].n_links;
    // The walk visits children in a fixed order, so the one nearer to the camera goes first: camera rays (a quarter to a half
    // of all rays) then find their hit early and skip what lies behind it; for the other rays the order is as good as any.
    if (!std::getenv("RTAMD_NO_CAMERA_ORDER")) {
      auto dist2 = [&](const Box3d& b) {
        double d = 0.0;
        for (int k = 0; k < 3; ++k) {
          const double c = s_.camera.origin[k];
          const double q = c < b.mn[k] ? b.mn[k] - c : (c > b.mx[k] ? c - b.mx[k] : 0.0);  // distance to the box, 0 inside
          d += q * q;
        }
        return d;
      };
      if (dist2(nodes[(size_t)r].box) < dist2(nodes[(size_t)l].box)) std::swap(l, r);
    }
    nodes[(size_t)me].left = l;
    nodes[(size_t)me].right = r;
  }

  // The binary tree of build_node() folded into nodes of up to four children: a node starts with the two children of its binary
  // node and, while it has room, replaces the child of largest surface area that is itself an inner node by that node's two
  // children.  A leaf keeps what the threaded list gives it: the object's own box, except that a sphere of negative radius
  // stands behind the box of its node in the reference's tree (see guard_box), and a medium the reference visits twice takes two
  // slots, one per visit (each draws its own free path; the nearer of the two hits wins whichever comes first).
  struct WideItem { Box3d box; int node; uint32_t word; };
  void wide_items(const std::vector<Node>& nodes, int n, std::vector<WideItem>& out) const {
    const Node& nd = nodes[(size_t)n];
    if (nd.obj < 0) { out.push_back(WideItem{nd.box, n, 0u}); return; }
    for (uint32_t word : wide_pre)
      if ((int)(word & LEAF_INDEX_MASK) == nd.obj) return;  // tested ahead of the walk
    const Box3d b = has_guard(nd.obj) ? guard_box(nd.obj) : nd.box;
    out.push_back(WideItem{b, -1, leaf_word(nd.obj)});
    if (!twice_.empty() && twice_[(size_t)nd.obj]) out.push_back(WideItem{b, -1, leaf_word(nd.obj) | LEAF_SECOND});
  }
  void build_wide(const std::vector<Node>& nodes) {
    wide.clear();
    wide_pre.clear();
    if (!std::getenv("RTAMD_NO_PRE_LEAVES") && nodes.size() > 16) {
      const double whole = nodes[0].box.area();
      for (const Node& nd : nodes) {
        if (nd.obj < 0 || has_guard(nd.obj) || !(nd.box.area() >= 0.5 * whole)) continue;
        const bool twice = !twice_.empty() && twice_[(size_t)nd.obj];
        if (wide_pre.size() + (twice ? 2u : 1u) > kWidePre) break;
        wide_pre.push_back(leaf_word(nd.obj));
        if (twice) wide_pre.push_back(leaf_word(nd.obj) | LEAF_SECOND);
      }
    }
    wide.emplace_back();
    std::vector<int> queue{0};  // binary node of wide node k
    std::vector<WideItem> items, grown;
    for (size_t k = 0; k < queue.size(); ++k) {
      const Node& nd = nodes[(size_t)queue[k]];
      items.clear();
      if (nd.obj >= 0) {
        wide_items(nodes, queue[k], items);  // a tree of one object
      } else {
        wide_items(nodes, nd.left, items);
        wide_items(nodes, nd.right, items);
        for (;;) {
          int pick = -1;
          double best = -1.0;
          auto slots = [&](int n) {  // slots the binary node n takes as a child: one, or two for a medium visited twice
            const Node& c = nodes[(size_t)n];
            return (c.obj >= 0 && !twice_.empty() && twice_[(size_t)c.obj]) ? 2u : 1u;
          };
          for (size_t j = 0; j < items.size(); ++j) {
            if (items[j].node < 0) continue;
            const Node& in = nodes[(size_t)items[j].node];
            if (items.size() - 1 + slots(in.left) + slots(in.right) > 4) continue;
            const double a = items[j].box.area();
            if (a > best) { best = a; pick = (int)j; }
          }
          if (pick < 0) break;
          grown.clear();
          wide_items(nodes, nodes[(size_t)items[(size_t)pick].node].left, grown);
          wide_items(nodes, nodes[(size_t)items[(size_t)pick].node].right, grown);
          items.erase(items.begin() + pick);
          items.insert(items.end(), grown.begin(), grown.end());
        }
      }
      WideTmp w{};
      for (int c = 0; c < 4; ++c) {
        w.child[c] = kWideNone;
        for (int a = 0; a < 3; ++a) w.lo[c][a] = w.hi[c][a] = std::numeric_limits<double>::infinity();
      }
      for (size_t j = 0; j < items.size(); ++j) {
        for (int a = 0; a < 3; ++a) { w.lo[j][a] = items[j].box.mn[a]; w.hi[j][a] = items[j].box.mx[a]; }
        if (items[j].node >= 0) {
          w.child[j] = (uint32_t)wide.size();
          wide.emplace_back();
          queue.push_back(items[j].node);
        } else {
          w.child[j] = items[j].word;
        }
      }
      wide[k] = w;
    }
    // stack entries the deepest walk can hold: a node with m children leaves m - 1 of them stacked while the walk is below it
    std::vector<int> need(wide.size(), 0);
    for (size_t k = wide.size(); k-- > 0;) {
      int m = 0, below = 0;
      for (int c = 0; c < 4; ++c) {
        if (wide[k].child[c] == kWideNone) continue;
        ++m;
        if (!(wide[k].child[c] & LEAF_FLAG)) below = std::max(below, need[(size_t)wide[k].child[c]]);
      }
      need[k] = std::max(0, m - 1) + below;
    }
    wide_stack_need = need[0];
  }

  uint32_t leaf_word(int obj) const {
    const uint32_t f = s_.objects[obj].flags;
    return LEAF_FLAG | ((f & 15u) == 0u ? LEAF_SIMPLE : 0u) | ((f & RT_OBJ_SUB_BVH) ? LEAF_NESTED : 0u) | (uint32_t)obj;
  }
  // Emits the subtree of `n` in visiting order; returns the index of its first link.
  int emit(const std::vector<Node>& nodes, int n, int next, int depth, int tree, std::vector<TmpLink>& out) const {
    const Node& nd = nodes[(size_t)n];
    TmpLink l{};
    l.depth = depth; l.tree = tree; l.miss = next;
    const int me = (int)out.size();
    if (nd.obj >= 0) {
      if (has_guard(nd.obj)) {  // see guard_box(): a box entry of its own in front of the leaf
        const Box3d g = guard_box(nd.obj);
        TmpLink b = l;
        for (int k = 0; k < 3; ++k) { b.bmin[k] = g.mn[k]; b.bmax[k] = g.mx[k]; }
        b.hit = (uint32_t)(me + 1);
        out.push_back(b);
        l.depth = depth + 1;
      }
      l.hit = leaf_word(nd.obj);
      if (!twice_.empty() && twice_[(size_t)nd.obj]) {  // both visits, the second right behind the first
        l.miss = (int)out.size() + 1;
        out.push_back(l);
        l.hit |= LEAF_SECOND;
        l.miss = next;
      }
      out.push_back(l);
      return me;
    }
    for (int k = 0; k < 3; ++k) { l.bmin[k] = nd.box.mn[k]; l.bmax[k] = nd.box.mx[k]; }
    out.push_back(l);
    // children in visiting order right behind their parent: first subtree, then second subtree
    const int left_first = me + 1;
    const int right_first = left_first + nodes[(size_t)nd.left].n_links;
    (void)emit(nodes, nd.left, right_first, depth + 1, tree, out);
    (void)emit(nodes, nd.right, next, depth + 1, tree, out);
    out[(size_t)me].hit = (uint32_t)left_first;
    return me;
  }

  void emit_tree(const std::vector<int>& objs_in_order, int tree, std::vector<TmpLink>& out, Box3d& root_box) {
    // copies: keep the last visited one
    const bool timing = std::getenv("RTAMD_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
      if (!timing) return;
      auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "  own tree %d: %-22s %.3f s\n", tree, what, std::chrono::duration<double>(now - t_last).count());
      t_last = now;
    };
    const std::vector<char> earlier = earlier_copies(objs_in_order, [&](int o) { return !(s_.objects[o].flags & (RT_OBJ_MEDIUM | RT_OBJ_SUB_BVH)); });
    std::vector<int> keep;
    for (int o : objs_in_order)
      if (s_.objects[o].flags & (RT_OBJ_MEDIUM | RT_OBJ_SUB_BVH)) keep.push_back(o);
    mark("copies");
    for (int o : objs_in_order) {
      const rt_object& ob = s_.objects[o];
      if (ob.flags & (RT_OBJ_MEDIUM | RT_OBJ_SUB_BVH)) continue;
      if (earlier[(size_t)o]) continue;
      if (has_guard(o)) {
        // alone in a one-object node, the inverted box IS the node's box, and the reference's box test (src/aabb.rs:30-46) fails
        // for every ray: the sphere is never tested there, so it is not part of this tree either
        const Box3d g = guard_box(o);
        if (g.mn[0] > g.mx[0] || g.mn[1] > g.mx[1] || g.mn[2] > g.mx[2]) continue;
      }
      keep.push_back(o);
    }
    if (keep.empty() && !objs_in_order.empty()) keep.push_back(objs_in_order.back());  // a tree needs a leaf (all objects unreachable: keep one)
    std::vector<Box3d> boxes((size_t)s_.n_objects);
    for (int o : keep) boxes[(size_t)o] = object_box(o);
    std::vector<Node> nodes(keep.size() * 2 - 1);
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)std::max(1, atoi(e)));
    int spawn_depth = 0;
    while ((1u << (spawn_depth + 1)) <= std::max(1u, hw) && spawn_depth < 6) ++spawn_depth;
    mark("boxes");
    build_node(nodes, keep, boxes, 0, (int)keep.size(), 0, spawn_depth);
    mark("split");
    root_box = nodes[0].box;
    if (tree == 0 && want_wide_) {
      build_wide(nodes);
      mark("four-wide form");
    }
    out.clear();
    out.reserve((size_t)nodes[0].n_links + 2);
    if (nodes[0].obj >= 0) {
      // a single object: give it a box entry so that the walk starts on a box, as every other tree does
      TmpLink b{};
      for (int k = 0; k < 3; ++k) { b.bmin[k] = nodes[0].box.mn[k]; b.bmax[k] = nodes[0].box.mx[k]; }
      b.hit = 1u; b.miss = LINK_END; b.depth = 0; b.tree = tree;
      out.push_back(b);
      TmpLink l{};
      l.hit = leaf_word(nodes[0].obj);
      l.miss = LINK_END; l.depth = 1; l.tree = tree;
      if (!twice_.empty() && twice_[(size_t)nodes[0].obj]) {
        l.miss = 2;
        out.push_back(l);
        l.hit |= LEAF_SECOND;
        l.miss = LINK_END;
      }
      out.push_back(l);
      return;
    }
    (void)emit(nodes, 0, LINK_END, 0, tree, out);
    mark("threaded list");
  }
};

}  // namespace rti
