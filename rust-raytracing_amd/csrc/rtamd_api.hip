// C ABI of librtamd.so (declared in include/rtamd.h): context, scene validation and upload, dispatch to the
// per-precision render code.  No kernels in this file.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "accel_build.h"
#include "context.h"

#include <limits>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace rti {
thread_local std::string g_err;
}
using namespace rti;

// ---- static scratch of rt_render_noalloc ----
// Everything that entry point needs on the device beyond the caller's buffers: 16 counter words (ray count, path cursor) and the
// RGB8 image between the colour conversion and the copy to the host.  A file-scope device array: every GPU the library runs on
// holds its own copy, at its own address, until the process ends.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress; the address that call returns (kept per device below) is what kernels, memsets and copies are given.
#define RT_NOALLOC_MAX_PIXELS (3840u * 2160u)
#define RT_NOALLOC_COUNTER_BYTES 256u
static __device__ __align__(256) unsigned char g_noalloc_scratch[RT_NOALLOC_COUNTER_BYTES + (size_t)RT_NOALLOC_MAX_PIXELS * 3u];
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_of_device[kMaxDevices] = {};  // device address of g_noalloc_scratch on each device, once looked up

// With `device` current: the address of this device's copy of the scratch array (looked up once per device).
int scratch_for_device(int device, void** out) {
  if (device < 0 || device >= kMaxDevices) return fail(RT_E_NO_DEVICE, "device index beyond the scratch table");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_of_device[device]) {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_noalloc_scratch));
    if (e != hipSuccess || !ptr) return fail(RT_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    g_scratch_of_device[device] = ptr;
  }
  *out = g_scratch_of_device[device];
  return RT_OK;
}
}  // namespace

namespace {

template <class R> void free_image(SceneImage<R>& im) {
  if (im.d_blob) (void)hipFree(im.d_blob);
  if (im.d_exact_boxes) (void)hipFree(im.d_exact_boxes);
  im = SceneImage<R>{};
}

void free_scene(rt_context* c) {
  free_image(c->sf);
  free_image(c->sd);
  free_image(c->sd_own);
  if (c->d_images) (void)hipFree(c->d_images);
  c->d_images = nullptr;
  for (uint8_t* p : c->d_image_bytes) (void)hipFree(p);
  c->d_image_bytes.clear();
  c->has_scene = false;
}

// ---- scene validation + threaded-link construction (host) ----
struct LinkBuilder {
  const rt_scene_desc& s;
  std::vector<TmpLink> links;
  std::vector<uint8_t> node_seen;
  std::string err;
  // splice[o] != 0: the nested tree held by object o is emitted in place of o's leaf, its boxes turned into world-space boxes
  // around what the holder's RotateY / Translate make of them (see rt_upload_scene)
  const std::vector<char>* splice = nullptr;
  // skip[o] != 0: object o is an exact copy of an object the walk visits later; its leaf is left out (see rt_upload_scene)
  const std::vector<char>* skip = nullptr;
  // objects in the order their leaves are emitted: a node emits its second subtree before its first (the first one's entries need
  // to know where the second begins), so this is the walk's visiting order backwards
  std::vector<int> leaves_emitted;
  explicit LinkBuilder(const rt_scene_desc& sc) : s(sc), node_seen((size_t)std::max(0, sc.n_nodes), 0) {
    links.reserve((size_t)std::max(0, sc.n_nodes) * 2 + 16);  // a node's entry and at most one leaf entry per child
    leaves_emitted.reserve((size_t)std::max(0, sc.n_nodes) + 16);
  }

  // Axis-aligned box around a box of holder `h`'s frame taken to world space (src/hittable.rs:166-197 for the rotation, 127-138
  // for the translation), with a margin that covers the rounding of these few operations many times over.
  void to_world_box(int h, double mn[3], double mx[3]) const {
    const rt_object& ob = s.objects[h];
    double lo[3] = {mn[0], mn[1], mn[2]}, hi[3] = {mx[0], mx[1], mx[2]};
    if (ob.flags & RT_OBJ_ROTATE_Y) {
      const double inf = std::numeric_limits<double>::infinity();
      double a[3] = {inf, inf, inf}, b[3] = {-inf, -inf, -inf};
      for (int k = 0; k < 8; ++k) {
        const double x = (k & 1) ? hi[0] : lo[0], y = (k & 2) ? hi[1] : lo[1], z = (k & 4) ? hi[2] : lo[2];
        const double w[3] = {ob.cos_theta * x + ob.sin_theta * z, y, -ob.sin_theta * x + ob.cos_theta * z};
        for (int q = 0; q < 3; ++q) { a[q] = std::min(a[q], w[q]); b[q] = std::max(b[q], w[q]); }
      }
      for (int q = 0; q < 3; ++q) { lo[q] = a[q]; hi[q] = b[q]; }
    }
    // A box that comes through unchanged, or moved by an offset that adds without rounding, stays exactly the box the reference
    // tests: where two boxes of a nested tree share a face (the ground blocks of final-scene), whether the second one is still
    // visited after a hit on that face hangs on an exact `t_max <= t_min`, and with it which of the two coincident sides is
    // reported.  Only a rotated box, or a sum that had to round, gets a margin.
    const bool rotated = (ob.flags & RT_OBJ_ROTATE_Y) != 0;
    for (int q = 0; q < 3; ++q) {
      double a = lo[q], b = hi[q];
      bool exact = !rotated;
      if (ob.flags & RT_OBJ_TRANSLATE) {
        a = lo[q] + ob.offset[q]; b = hi[q] + ob.offset[q];
        exact = exact && (a - ob.offset[q] == lo[q]) && (a - lo[q] == ob.offset[q]) && (b - ob.offset[q] == hi[q]) && (b - hi[q] == ob.offset[q]);
      }
      const double m = exact ? 0.0 : 1e-9 * std::max({1.0, std::fabs(a), std::fabs(b)});
      mn[q] = a - m; mx[q] = b + m;
    }
  }

  bool simple(int obj) const { return (s.objects[obj].flags & 15u) == 0u; }

  int leaf(int obj, int next, int depth, int tree) {
    TmpLink l{};
    l.hit = LEAF_FLAG | (simple(obj) ? LEAF_SIMPLE : 0u) | ((s.objects[obj].flags & RT_OBJ_SUB_BVH) ? LEAF_NESTED : 0u) | (uint32_t)obj;
    l.miss = next;
    l.depth = depth; l.tree = tree;
    links.push_back(l);
    if (tree == 0) leaves_emitted.push_back(obj);
    return (int)links.size() - 1;
  }
  int child(int c, int next, int depth, int tree, int holder = -1) {
    if (c >= 0) return node(c, next, depth, tree, holder);
    int obj = ~c;
    if (obj < 0 || obj >= s.n_objects) { err = "BVH leaf references an object out of range"; return LINK_END; }
    if ((tree != 0 || holder >= 0) && (s.objects[obj].flags & RT_OBJ_SUB_BVH)) { err = "nested BVH inside a nested BVH is not supported"; return LINK_END; }
    if (splice && (*splice)[(size_t)obj]) return node(s.objects[obj].geom, next, depth, tree, obj);  // the nested tree in place of its holder
    if (skip && (*skip)[(size_t)obj]) return next;  // no entry: whoever pointed here goes on with what follows
    return leaf(obj, next, depth, tree);
  }
  // Emits the subtree of `n`; returns the index of its first link.  `next` is where traversal continues afterwards.
  // `holder` >= 0: the subtree belongs to the nested tree of that object and is being spliced into the world's tree.
  int node(int n, int next, int depth, int tree, int holder = -1) {
    if (!err.empty()) return LINK_END;
    if (n < 0 || n >= s.n_nodes) { err = "BVH node index out of range"; return LINK_END; }
    if (node_seen[n]) { err = "BVH node referenced twice (nodes must form trees)"; return LINK_END; }
    if (depth > 512) { err = "BVH deeper than 512 levels"; return LINK_END; }
    node_seen[n] = 1;
    const rt_bvh_node& nd = s.nodes[n];
    // the children's records are read next (the second child's at once, the first child's after the second subtree)
    if (nd.right >= 0 && nd.right < s.n_nodes) __builtin_prefetch(&s.nodes[nd.right]);
    if (nd.left >= 0 && nd.left < s.n_nodes) __builtin_prefetch(&s.nodes[nd.left]);
    int me = (int)links.size();
    links.push_back(TmpLink{});
    int right_first, left_first;
    // BVHNode::new gives a one-object node the same child twice (src/bvh.rs:38); the second test of a deterministic
    // shape with t_max = its own t reproduces the first record, so it is dropped unless the object draws random numbers.
    bool twin = nd.left < 0 && nd.left == nd.right && !(s.objects[std::min(std::max(~nd.left, 0), s.n_objects - 1)].flags & RT_OBJ_MEDIUM);
    if (twin) {
      left_first = child(nd.left, next, depth + 1, tree, holder);
    } else {
      right_first = child(nd.right, next, depth + 1, tree, holder);
      // a medium in a one-object node: its second visit draws a random number of its own (src/constant_medium.rs:91)
      if (nd.left < 0 && nd.left == nd.right && right_first >= 0) links[(size_t)right_first].hit |= LEAF_SECOND;
      left_first = child(nd.left, right_first, depth + 1, tree, holder);
    }
    TmpLink& l = links[me];
    for (int k = 0; k < 3; ++k) { l.bmin[k] = nd.bmin[k]; l.bmax[k] = nd.bmax[k]; }
    if (holder >= 0 && !(l.bmin[0] > l.bmax[0] || l.bmin[1] > l.bmax[1] || l.bmin[2] > l.bmax[2])) to_world_box(holder, l.bmin, l.bmax);
    if (left_first < 0) {  // nothing left below this node (every leaf was an earlier copy) and nothing after it: a box no ray enters
      for (int k = 0; k < 3; ++k) { l.bmin[k] = 1.0; l.bmax[k] = -1.0; }
      left_first = me;
    }
    l.hit = (uint32_t)left_first;
    l.miss = next;
    l.depth = depth; l.tree = tree;
    return me;
  }
};

int validate(const rt_scene_desc& s, std::string& err) {
  auto bad = [&](const char* m) { err = m; return RT_E_INVALID; };
  if (s.n_nodes <= 0 || s.n_objects <= 0 || s.n_shapes <= 0 || s.n_materials <= 0) return bad("scene has no nodes/objects/shapes/materials");
  if (!s.nodes || !s.objects || !s.shapes || !s.materials) return bad("scene array pointer is NULL");
  if (s.n_textures > 0 && !s.textures) return bad("textures pointer is NULL");
  if (s.n_perlins > 0 && !s.perlins) return bad("perlins pointer is NULL");
  if (s.n_images > 0 && !s.images) return bad("images pointer is NULL");
  if ((uint32_t)s.n_objects > LEAF_INDEX_MASK) return bad("too many objects (at most 2^28 - 1)");
  if (s.root_node < 0 || s.root_node >= s.n_nodes) return bad("root_node out of range");
  for (int i = 0; i < s.n_textures; ++i) {
    const rt_texture& t = s.textures[i];
    switch (t.tag) {
      case RT_TEX_SOLID: break;
      case RT_TEX_CHECKER:
        if (t.even < 0 || t.even >= s.n_textures || t.odd < 0 || t.odd >= s.n_textures) return bad("checker sub-texture out of range");
        if (s.textures[t.even].tag == RT_TEX_CHECKER || s.textures[t.odd].tag == RT_TEX_CHECKER) {
          err = "checker of checker textures is not supported";
          return RT_E_UNSUPPORTED;
        }
        break;
      case RT_TEX_NOISE:
        if (t.perlin < 0 || t.perlin >= s.n_perlins) return bad("noise texture perlin index out of range");
        break;
      case RT_TEX_IMAGE:
        if (t.image >= s.n_images) return bad("image texture index out of range");
        break;
      default: return bad("unknown texture tag");
    }
  }
  for (int i = 0; i < s.n_images; ++i)
    if (s.images[i].width <= 0 || s.images[i].height <= 0 || !s.images[i].rgb) return bad("bad image");
  for (int i = 0; i < s.n_materials; ++i) {
    const rt_material& m = s.materials[i];
    if (m.tag < 0 || m.tag >= MAT_CLASSES) return bad("unknown material tag");
    bool needs_tex = m.tag == RT_MAT_LAMBERTIAN || m.tag == RT_MAT_DIFFUSE_LIGHT || m.tag == RT_MAT_ISOTROPIC;
    if (needs_tex && (m.texture < 0 || m.texture >= s.n_textures)) return bad("material texture out of range");
  }
  for (int i = 0; i < s.n_shapes; ++i) {
    const rt_shape& sh = s.shapes[i];
    if (sh.tag < RT_SHAPE_SPHERE || sh.tag > RT_SHAPE_BOX) return bad("unknown shape tag");
    if (sh.material < 0 || sh.material >= s.n_materials) return bad("shape material out of range");
  }
  for (int i = 0; i < s.n_objects; ++i) {
    const rt_object& o = s.objects[i];
    if (o.flags & ~(RT_OBJ_TRANSLATE | RT_OBJ_ROTATE_Y | RT_OBJ_MEDIUM | RT_OBJ_SUB_BVH)) return bad("unknown object flag");
    if (o.flags & RT_OBJ_SUB_BVH) {
      if (o.geom < 0 || o.geom >= s.n_nodes) return bad("object sub-BVH root out of range");
    } else if (o.geom < 0 || o.geom >= s.n_shapes) {
      return bad("object shape out of range");
    }
    if ((o.flags & RT_OBJ_MEDIUM) && (o.medium_material < 0 || o.medium_material >= s.n_materials)) return bad("medium material out of range");
  }
  return RT_OK;
}

// The radiance sums are added with hardware floating-point atomics, which act on ordinary device memory only: on host-mapped or
// managed (fine-grained) memory they are silently dropped and the image comes out black.  Such a buffer is refused.
int check_device_buffer(const void* p, const char* who) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(RT_E_INVALID, std::string(who) + ": accum_dev is not a device pointer");
  }
  if (at.type != hipMemoryTypeDevice)
    return fail(RT_E_INVALID, std::string(who) + ": accum_dev must be ordinary device memory (hipMalloc); host-mapped and managed memory drop the atomic adds");
  return RT_OK;
}

int check_render_params(const rt_context* c, const rt_render_params* p) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  if (!p) return fail(RT_E_INVALID, "params is NULL");
  if (!c->has_scene) return fail(RT_E_NO_SCENE, "no scene uploaded");
  if (p->width < 2 || p->height < 2) return fail(RT_E_INVALID, "width and height must be at least 2 (the sample loop divides by W-1 and H-1)");
  if ((uint64_t)p->width * p->height > 0x7FFFFFFFull) return fail(RT_E_INVALID, "image too large");
  if (p->sample_end < p->sample_begin) return fail(RT_E_INVALID, "sample_end < sample_begin");
  if (p->max_depth > kMaxDepth) return fail(RT_E_INVALID, "max_depth exceeds 4096");
  if ((p->precision & ~RT_F64_OWN_TREE) != RT_PRECISION_F32 && (p->precision & ~RT_F64_OWN_TREE) != RT_PRECISION_F64) return fail(RT_E_INVALID, "unknown precision");
  if (p->pipeline < RT_PIPELINE_AUTO || p->pipeline > RT_PIPELINE_PERSISTENT) return fail(RT_E_INVALID, "unknown pipeline");
  if (p->band_stride > 1u) {
    if (p->band_rows == 0u || (p->band_rows & (p->band_rows - 1u)) != 0u || p->band_rows > 32768u) return fail(RT_E_INVALID, "band_rows must be a power of two (1..32768)");
    if (p->band_offset >= p->band_stride) return fail(RT_E_INVALID, "band_offset must be below band_stride");
  }
  return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
const char* rt_version(void) { return "rtamd 0.1 (gfx950)"; }

int rt_create(rt_context** out, int device) {
  if (!out) return fail(RT_E_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(RT_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= n) return fail(RT_E_NO_DEVICE, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  auto* c = new rt_context;
  c->device = device;
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  auto cleanup = [&](int code, const std::string& m) { rt_destroy(c); return fail(code, m); };
  if (hipStreamCreate(&c->stream) != hipSuccess) return cleanup(RT_E_HIP, "hipStreamCreate failed");
  for (auto& ev : c->ev)
    if (hipEventCreate(&ev) != hipSuccess) return cleanup(RT_E_HIP, "hipEventCreate failed");
  if (hipMalloc((void**)&c->d_counters, sizeof(uint32_t) * kCounterWords) != hipSuccess) return cleanup(RT_E_HIP, "hipMalloc counters failed");
  if (hipMalloc((void**)&c->d_ray_count, sizeof(unsigned long long) * rti::kRayCountWords) != hipSuccess) return cleanup(RT_E_HIP, "hipMalloc ray counter failed");
  if (hipHostMalloc((void**)&c->h_rays, 64, hipHostMallocDefault) != hipSuccess) return cleanup(RT_E_HIP, "hipHostMalloc failed");
  // set-up of the allocation-free entry point: this device's address of the static scratch array, looked up with the device current
  if (int rc = scratch_for_device(device, &c->scratch)) { std::string m = g_err; return cleanup(rc, m); }
  *out = c;
  return RT_OK;
}

void rt_destroy(rt_context* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rt_comm_destroy(c);
  free_scene(c);
  if (c->ws) (void)hipFree(c->ws);
  if (c->accum) (void)hipFree(c->accum);
  if (c->d_rgb) (void)hipFree(c->d_rgb);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->d_ray_count) (void)hipFree(c->d_ray_count);
  if (c->h_rays) (void)hipHostFree(c->h_rays);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int rt_upload_scene(rt_context* c, const rt_scene_desc* s) {
  if (!c || !s) return fail(RT_E_INVALID, "context or scene is NULL");
  std::string err;
  int rc = validate(*s, err);
  if (rc) return fail(rc, "rt_upload_scene: " + err);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  free_scene(c);
  c->progress_active = false;  // a frame in progress belonged to the scene that is being replaced

  // RTAMD_TIMING=1: seconds spent in each stage of the upload, on stderr
  const bool timing = std::getenv("RTAMD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "rt_upload_scene: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  // threaded links: the top-level tree first, then every nested tree
  LinkBuilder lb(*s);
  int root = lb.node(s->root_node, LINK_END, 0, 0);
  std::vector<int> sub_root((size_t)s->n_objects, LINK_END);
  int tree = 1;
  for (int i = 0; i < s->n_objects && lb.err.empty(); ++i)
    if (s->objects[i].flags & RT_OBJ_SUB_BVH) sub_root[i] = lb.node(s->objects[i].geom, LINK_END, 0, tree++);
  if (!lb.err.empty()) return fail(lb.err.find("not supported") != std::string::npos ? RT_E_UNSUPPORTED : RT_E_INVALID, "rt_upload_scene: " + lb.err);
  mark("reference tree -> links");
  // Which preset covers this scene?
  uint32_t feat = 0;
  bool has_nested = false;
  for (int i = 0; i < s->n_objects; ++i) {
    const rt_object& o = s->objects[i];
    if (o.flags & (RT_OBJ_TRANSLATE | RT_OBJ_ROTATE_Y)) feat |= F_XFORM;
    if (o.flags & RT_OBJ_MEDIUM) feat |= F_MEDIUM;
    if (o.flags & RT_OBJ_SUB_BVH) { feat |= F_NESTED; has_nested = true; continue; }
    switch (s->shapes[o.geom].tag) {
      case RT_SHAPE_SPHERE: feat |= F_SPHERE; break;
      case RT_SHAPE_MOVING_SPHERE: feat |= F_MOVING; break;
      case RT_SHAPE_BOX: feat |= F_BOX; break;
      default: feat |= F_RECT; break;
    }
  }
  for (int i = 0; i < s->n_materials; ++i) {
    if (s->materials[i].tag == RT_MAT_METAL) feat |= F_METAL;
    if (s->materials[i].tag == RT_MAT_DIELECTRIC) feat |= F_DIELECTRIC;
    if (s->materials[i].tag == RT_MAT_ISOTROPIC) feat |= F_MEDIUM;
  }
  for (int i = 0; i < s->n_textures; ++i) {
    if (s->textures[i].tag == RT_TEX_CHECKER) feat |= F_CHECKER;
    if (s->textures[i].tag == RT_TEX_NOISE) feat |= F_NOISE;
    if (s->textures[i].tag == RT_TEX_IMAGE) feat |= F_IMAGE | F_UV;
  }
  if (s->background.kind == RT_BG_GRADIENT) feat |= F_GRADIENT;
  const char* force = std::getenv("RTAMD_FORCE_GENERIC");
  // The small-scene presets stage the whole image in LDS: it has to fit there in both precisions (a Perlin table alone is 11 KiB
  // in double precision), otherwise the scene takes a per-lane preset, which falls back to staging the top of the tree only.
  BlobLayout projected{};
  projected.n_links = (uint32_t)lb.links.size(); projected.n_objects = (uint32_t)s->n_objects; projected.n_materials = (uint32_t)s->n_materials;
  projected.n_textures = (uint32_t)std::max(1, s->n_textures); projected.n_perlins = (uint32_t)s->n_perlins; projected.flat_mode = 1u;
  layout_sections<double>(projected);
  bool inverted_box = false;  // a sphere of negative radius: what it is reachable through is decided by the tree (see accel_build.h)
  for (int i = 0; i < s->n_objects; ++i) {
    if (s->objects[i].flags & RT_OBJ_SUB_BVH) continue;
    const rt_shape& sh = s->shapes[s->objects[i].geom];
    inverted_box = inverted_box || (sh.tag == RT_SHAPE_SPHERE && sh.p[3] < 0.0) || (sh.tag == RT_SHAPE_MOVING_SPHERE && sh.p[6] < 0.0);
  }
  size_t n_leaf_links = 0;
  for (const TmpLink& l : lb.links) n_leaf_links += (l.hit & LEAF_FLAG) ? 1u : 0u;
  const bool small = n_leaf_links <= kFlatMaxLeaves && !has_nested && !inverted_box && !(force && force[0] == '1') && !std::getenv("RTAMD_NO_FLAT") &&
                     projected.total_bytes <= kLdsAllMax;
  int preset = PRESET_ALL_LDS;  // build_image demotes to PRESET_ALL_TOP when the image does not fit in LDS
  if (small && (feat & ~P_BOXES) == 0) preset = PRESET_BOXES;
  else if (small && (feat & ~P_SMOKE) == 0) preset = PRESET_SMOKE;
  else if (small && (feat & ~P_SPHERES) == 0) preset = PRESET_SPHERES;
  else if ((feat & ~P_BALLS) == 0 && !(force && force[0] == '1')) preset = PRESET_BALLS_LDS;
  const bool uniform = preset != PRESET_ALL_LDS && preset != PRESET_BALLS_LDS;

  // Storage order of the links.  Small scenes: visiting order (the wave-uniform walk steps through it).  Otherwise by
  // (nested?, depth): the first entries are the top levels of the main tree, which is what gets staged in LDS.
  std::string order_err;
  auto reorder = [&](const std::vector<TmpLink>& in, int& root_io, std::vector<int>& sub_root_io) {
    std::vector<int> order;
    if (uniform) {
      for (int i = root_io; i >= 0; i = (in[i].hit & LEAF_FLAG) ? in[i].miss : (int)in[i].hit) order.push_back(i);
      if (order.size() != in.size()) order_err = "rt_upload_scene: internal error while ordering links";
    }
    if (!uniform || !order_err.empty()) {
      // stable counting sort by (nested?, depth), on the host's threads: every thread counts the keys of its chunk, the counts
      // are turned into first positions chunk by chunk (so that equal keys keep their order), and every thread places its chunk
      order.resize(in.size());
      const size_t n_chunks = parallel_chunk_count(in.size());
      std::vector<int> chunk_max(n_chunks, 0);
      parallel_chunks(in.size(), [&](size_t t, size_t lo, size_t hi) {
        int m = 0;
        for (size_t i = lo; i < hi; ++i) m = std::max(m, in[i].depth);
        chunk_max[t] = m;
      });
      int max_depth = 0;
      for (int m : chunk_max) max_depth = std::max(max_depth, m);
      const size_t n_keys = 2 * (size_t)(max_depth + 1);
      auto key = [&](const TmpLink& l) { return (size_t)(l.tree != 0) * (size_t)(max_depth + 1) + (size_t)l.depth; };
      std::vector<std::vector<size_t>> count(n_chunks, std::vector<size_t>(n_keys, 0));
      parallel_chunks(in.size(), [&](size_t t, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) ++count[t][key(in[i])];
      });
      size_t at = 0;
      for (size_t k = 0; k < n_keys; ++k)
        for (size_t t = 0; t < n_chunks; ++t) { const size_t n_here = count[t][k]; count[t][k] = at; at += n_here; }
      parallel_chunks(in.size(), [&](size_t t, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) order[count[t][key(in[i])]++] = (int)i;
      });
    }
    std::vector<int> new_index(order.size());
    parallel_chunks(order.size(), [&](size_t, size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) new_index[order[i]] = (int)i;
    });
    std::vector<TmpLink> out(order.size());
    parallel_chunks(order.size(), [&](size_t, size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        TmpLink l = in[order[i]];
        if (!(l.hit & LEAF_FLAG)) l.hit = (uint32_t)new_index[l.hit];
        if (l.miss >= 0) l.miss = new_index[l.miss];
        out[i] = l;
      }
    });
    root_io = new_index[root_io];
    for (int& r : sub_root_io)
      if (r >= 0) r = new_index[r];
    return out;
  };
  // The single-precision per-lane walk gets a tree of the library's own (see accel_build.h) unless RTAMD_REFERENCE_TREE is set.
  std::vector<TmpLink> fast_links;
  int fast_root = LINK_END;
  std::vector<int> fast_sub_root;
  // There, a nested tree whose holder only translates / rotates it and whose objects are bare shapes is dissolved: the objects
  // take the holder's wrappers and become leaves of the main tree.  A scene left without nested trees runs the kernels that carry
  // no nested-walk code (half the registers).  RTAMD_KEEP_NESTED=1 turns this off.
  const bool own_tree = !uniform && !std::getenv("RTAMD_REFERENCE_TREE");
  // (copies of the caller's arrays only where something in them changes: scenes with nested trees)
  std::vector<rt_object> objects32;
  std::vector<rt_shape> shapes32;
  if (has_nested) { objects32.assign(s->objects, s->objects + s->n_objects); shapes32.assign(s->shapes, s->shapes + s->n_shapes); }
  std::vector<char> dissolved((size_t)s->n_objects, 0);
  int preset32 = preset;
  if (own_tree) {
    bool nested_left = false;
    for (int i = 0; i < s->n_objects; ++i) {
      const rt_object& holder = s->objects[i];
      if (!(holder.flags & RT_OBJ_SUB_BVH)) continue;
      std::vector<int> children;
      for (int k = sub_root[(size_t)i]; k >= 0;) {
        const TmpLink& l = lb.links[(size_t)k];
        if (l.hit & LEAF_FLAG) { children.push_back((int)(l.hit & LEAF_INDEX_MASK)); k = l.miss; }
        else k = (int)l.hit;
      }
      bool bare = !(holder.flags & RT_OBJ_MEDIUM) && !std::getenv("RTAMD_KEEP_NESTED");
      for (int ch : children) bare = bare && s->objects[ch].flags == 0u;
      if (!bare) { nested_left = true; continue; }
      dissolved[(size_t)i] = 1;
      for (int ch : children) {
        rt_object& o = objects32[(size_t)ch];
        // A sphere under [Translate(] RotateY is the sphere around the moved centre: same t, same point.  Translate, the outermost
        // wrapper, ends with set_face_normal against the world direction (src/hittable.rs:146-148), which undoes whatever
        // RotateY's mixed-frame call (src/hittable.rs:236-239) did to the normal; only the front_face flag can then differ, at
        // grazing incidence - it is read by Dielectric alone - and (u, v) turn with the sphere, so those materials keep the
        // wrappers.  Under a RotateY without a Translate the mixed-frame call decides the normal itself: wrappers kept.
        const rt_shape& sh = s->shapes[o.geom];
        const rt_material& m = s->materials[sh.material];
        const bool uv_free = m.texture < 0 || m.texture >= s->n_textures || s->textures[m.texture].tag != RT_TEX_IMAGE;
        const bool normal_restored = (holder.flags & RT_OBJ_TRANSLATE) != 0 || !(holder.flags & RT_OBJ_ROTATE_Y);
        if (sh.tag == RT_SHAPE_SPHERE && m.tag != RT_MAT_DIELECTRIC && uv_free && normal_restored) {
          shapes32.push_back(sh);
          rt_shape& moved = shapes32.back();
          const bool rot = (holder.flags & RT_OBJ_ROTATE_Y) != 0, tr = (holder.flags & RT_OBJ_TRANSLATE) != 0;
          const double cs = rot ? holder.cos_theta : 1.0, sn = rot ? holder.sin_theta : 0.0;
          moved.p[0] = cs * sh.p[0] + sn * sh.p[2] + (tr ? holder.offset[0] : 0.0);   // src/hittable.rs:229-232, 146
          moved.p[1] = sh.p[1] + (tr ? holder.offset[1] : 0.0);
          moved.p[2] = -sn * sh.p[0] + cs * sh.p[2] + (tr ? holder.offset[2] : 0.0);
          o.geom = (int32_t)shapes32.size() - 1;
          if (rot) {  // front_face still goes through the wrappers' set_face_normal calls (read by nothing but Dielectric, reported by rt_trace_batch)
            o.flags = OBJ_FRONT_CHAIN;
            o.sin_theta = holder.sin_theta; o.cos_theta = holder.cos_theta;
          }
          continue;
        }
        o.flags = holder.flags & (RT_OBJ_TRANSLATE | RT_OBJ_ROTATE_Y);
        for (int k = 0; k < 3; ++k) o.offset[k] = holder.offset[k];
        o.sin_theta = holder.sin_theta; o.cos_theta = holder.cos_theta;
      }
    }
    if (has_nested && !nested_left && preset == PRESET_ALL_LDS) preset32 = PRESET_FLATALL_LDS;  // build_image demotes to _TOP when too large
  }
  // Double precision, the caller's tree: a nested tree whose holder only translates / rotates it and whose objects are bare shapes
  // is spliced into the world's tree in place of its holder - same entries in the same order, so the same leaf is tested at the
  // same point of the walk with the same t_max.  Its objects take the holder's wrappers: Translate(RotateY(shape)) forms the
  // same local ray and takes the record back with the same operations as Translate(RotateY(tree)) does around the shape's hit,
  // bit for bit.  Its boxes become world-space boxes around the rotated and translated originals: larger, which can only add
  // leaves to the ones a ray gets to test (see BoxLink).  The image then has no nested tree and runs the kernels without
  // nested-walk code, whose lanes step through the tree instead of each finishing its own walk inside a turn (final-scene: 10
  // of 64 lanes were active there).  RTAMD_KEEP_NESTED=1 turns this off as well.
  std::vector<rt_object> objects64;
  if (has_nested) objects64.assign(s->objects, s->objects + s->n_objects);
  std::vector<char> splice((size_t)s->n_objects, 0);
  bool nested_left64 = false, spliced_any = false;
  if (!uniform && has_nested) {
    for (int i = 0; i < s->n_objects; ++i) {
      const rt_object& holder = s->objects[i];
      if (!(holder.flags & RT_OBJ_SUB_BVH)) continue;
      std::vector<int> children;
      for (int k = sub_root[(size_t)i]; k >= 0;) {
        const TmpLink& l = lb.links[(size_t)k];
        if (l.hit & LEAF_FLAG) { children.push_back((int)(l.hit & LEAF_INDEX_MASK)); k = l.miss; }
        else k = (int)l.hit;
      }
      bool bare = !(holder.flags & RT_OBJ_MEDIUM) && !std::getenv("RTAMD_KEEP_NESTED");
      for (int ch : children) bare = bare && s->objects[ch].flags == 0u;
      if (!bare) { nested_left64 = true; continue; }
      splice[(size_t)i] = 1;
      spliced_any = true;
      for (int ch : children) {
        rt_object& o = objects64[(size_t)ch];
        o.flags = holder.flags & (RT_OBJ_TRANSLATE | RT_OBJ_ROTATE_Y);
        for (int k = 0; k < 3; ++k) o.offset[k] = holder.offset[k];
        o.sin_theta = holder.sin_theta; o.cos_theta = holder.cos_theta;
      }
    }
  }
  // Exact copies (the `random` scene adds its three large spheres 22 times, src/scenes.rs:197-217): every copy returns the same
  // distance, a tie the reference gives to the copy its walk tests last.  The earlier copies' leaves are left out of the list -
  // 17 sphere tests per ray become 5 - and the answer is the one the reference gives.
  std::vector<char> skip((size_t)s->n_objects, 0);
  bool skipped_any = false;
  // One thing does change: a hit that does not write (u, v) - a moving sphere, a medium - reports whatever an earlier accepted hit
  // left in the reference's shared record, and with fewer accepted hits on the way that leftover can be another.  Only an image
  // texture reads (u, v): with one on such an object the copies stay.
  const bool stale_uv_read = reads_leftover_uv(*s);
  if (!uniform && !stale_uv_read && !std::getenv("RTAMD_KEEP_COPIES")) {
    mark("features, preset");
    FastTreeBuilder sig(*s);
    sig.sort_pairs = [&](uint64_t* k, uint32_t* v, size_t n) { return gpu_sort_pairs(c, k, v, n) == RT_OK; };
    // leaves of the world's tree in the order the walk visits them
    const std::vector<int> order(lb.leaves_emitted.rbegin(), lb.leaves_emitted.rend());
    skip = sig.earlier_copies(order, [&](int o) { return sig.copy_eligible(o); });
    for (char f : skip) skipped_any = skipped_any || f != 0;
    mark("f64 list: copies found");
  }
  rt_scene_desc scene64 = *s;
  if (has_nested) scene64.objects = objects64.data();
  std::vector<TmpLink> links64;
  int root64 = LINK_END;
  std::vector<int> sub_root64((size_t)s->n_objects, LINK_END);
  int preset64 = preset;
  const bool relinked = spliced_any || skipped_any;
  // Two different objects with the same geometry (same wrappers, same shape; different materials, or they would be copies of one
  // another and all but one left out above) tie on every ray that hits them, and the reference decides such ties by the order of its
  // walk: a scene that holds such a pair keeps the list walk of the caller's tree in double precision.  A medium never ties (its
  // distance is drawn).  64-bit hashes; a chance collision only costs the faster walk.
  bool coincident = false;
  if (!uniform) {
    std::vector<uint64_t> keys;
    keys.reserve((size_t)s->n_objects);
    for (int i = 0; i < s->n_objects; ++i) {
      const rt_object& o = scene64.objects[i];
      if ((o.flags & (RT_OBJ_SUB_BVH | RT_OBJ_MEDIUM)) || skip[(size_t)i]) continue;
      const rt_shape& sh = s->shapes[o.geom];
      uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)(uint32_t)sh.tag ^ ((uint64_t)o.flags << 8);
      auto mix = [&](double v) {
        uint64_t b;
        std::memcpy(&b, &v, 8);
        h = (h ^ b) * 0x100000001B3ull;
        h ^= h >> 29;
      };
      for (int k = 0; k < 9; ++k) mix(sh.p[k]);
      if (o.flags & RT_OBJ_TRANSLATE) for (int k = 0; k < 3; ++k) mix(o.offset[k]);
      if (o.flags & RT_OBJ_ROTATE_Y) { mix(o.sin_theta); mix(o.cos_theta); }
      keys.push_back(h);
    }
    std::vector<uint32_t> unused_vals;
    if (keys.size() >= 65536) unused_vals.resize(keys.size());
    if (unused_vals.empty() || gpu_sort_pairs(c, keys.data(), unused_vals.data(), keys.size()) != RT_OK) std::sort(keys.begin(), keys.end());
    coincident = std::adjacent_find(keys.begin(), keys.end()) != keys.end();
    if (coincident && timing) std::fprintf(stderr, "rt_upload_scene: double precision keeps the list walk of the caller's tree (coincident objects)\n");
  }
  if (relinked) {
    LinkBuilder lbs(scene64);
    lbs.splice = &splice;
    lbs.skip = &skip;
    root64 = lbs.node(s->root_node, LINK_END, 0, 0);
    int tree64 = 1;
    for (int i = 0; i < s->n_objects && lbs.err.empty(); ++i)
      if ((s->objects[i].flags & RT_OBJ_SUB_BVH) && !splice[(size_t)i]) sub_root64[(size_t)i] = lbs.node(s->objects[i].geom, LINK_END, 0, tree64++);
    if (!lbs.err.empty()) return fail(RT_E_INVALID, "rt_upload_scene: " + lbs.err);
    links64 = reorder(lbs.links, root64, sub_root64);
    if (has_nested && !nested_left64 && preset == PRESET_ALL_LDS) preset64 = PRESET_FLATALL_LDS;
    mark("f64 list: nested trees spliced, copies left out");
  }
  rt_scene_desc scene32 = *s;
  if (has_nested) {
    scene32.objects = objects32.data();
    scene32.shapes = shapes32.data();
    scene32.n_shapes = (int32_t)shapes32.size();
  }
  mark("preset, nested trees");
  // the own tree's four-wide form, handed to the images that walk the own tree when its walk's stack fits (device_scene.h: kWideStack)
  std::vector<WideTmp> wide_nodes;
  std::vector<uint32_t> wide_pre;  // leaves that image keeps outside the tree (FastTreeBuilder::wide_pre)
  // Scenes of 100 000 objects and more (RTAMD_GPU_BUILD_MIN sets another threshold) whose objects are all plain leaves - no nested
  // tree, no medium, no sphere of negative radius - get that tree from the device-side builder (gpu_build.hip): the host computes
  // one box and one leaf word per object on its threads and the rest happens on the GPU.  Such an image carries no threaded list of
  // the own tree (nothing walks it when the four-wide form exists); if the four-wide walk would need a deeper stack than the
  // kernels have, the host builder below takes over.
  struct DeviceTree {
    WideNode* nodes = nullptr;
    uint32_t n = 0;
    ~DeviceTree() { if (nodes) (void)hipFree(nodes); }
  } device_tree;
  size_t gpu_min = 100000;
  if (const char* e = std::getenv("RTAMD_GPU_BUILD_MIN")) gpu_min = (size_t)std::max(2, atoi(e));
  if (own_tree && !has_nested && !(feat & F_MEDIUM) && !inverted_box && (size_t)s->n_objects >= gpu_min && !std::getenv("RTAMD_NO_GPU_BUILD") &&
      !std::getenv("RTAMD_NO_WIDE")) {
    FastTreeBuilder ftb(scene32);
    std::vector<int> keep;
    keep.reserve((size_t)s->n_objects);
    for (int o = 0; o < s->n_objects; ++o)
      if (!skip[(size_t)o]) keep.push_back(o);
    const size_t n_keep = keep.size();
    std::vector<float> boxes(6 * n_keep);
    std::vector<uint32_t> words(n_keep);
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)std::max(1, atoi(e)));
    const size_t n_threads = n_keep < 65536 ? 1 : std::max(1u, std::min(hw, 16u));
    std::vector<Box3d> whole_of(n_threads), centres_of(n_threads);
    auto work = [&](size_t t, size_t lo, size_t hi) {
      Box3d whole, cen;
      whole.reset(); cen.reset();
      for (size_t q = lo; q < hi; ++q) {
        const Box3d b = ftb.leaf_box(keep[q]);
        for (int k = 0; k < 3; ++k) {
          float f = (float)b.mn[k];
          boxes[6 * q + k] = ((double)f > b.mn[k]) ? std::nextafterf(f, -INFINITY) : f;  // single-precision boxes contain the double ones
          f = (float)b.mx[k];
          boxes[6 * q + 3 + k] = ((double)f < b.mx[k]) ? std::nextafterf(f, INFINITY) : f;
        }
        words[q] = ftb.word_of(keep[q]);
        whole.grow(b);
        Box3d c1;
        for (int k = 0; k < 3; ++k) c1.mn[k] = c1.mx[k] = 0.5 * (b.mn[k] + b.mx[k]);
        cen.grow(c1);
      }
      whole_of[t] = whole; centres_of[t] = cen;
    };
    {
      std::vector<std::thread> pool;
      const size_t per = (n_keep + n_threads - 1) / n_threads;
      for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work, t, std::min(n_keep, t * per), std::min(n_keep, (t + 1) * per));
      work(0, 0, std::min(n_keep, per));
      for (std::thread& t : pool) t.join();
    }
    Box3d whole;
    whole.reset();
    for (const Box3d& b : whole_of) whole.grow(b);
    // objects as large as the scene stay outside the tree, tested ahead of the walk (FastTreeBuilder::wide_pre has the reason)
    std::vector<uint32_t> pre;
    std::vector<char> is_pre(n_keep, 0);
    if (!std::getenv("RTAMD_NO_PRE_LEAVES")) {
      const double whole_area = whole.area();
      for (size_t q = 0; q < n_keep && pre.size() < kWidePre; ++q) {
        Box3d b;
        for (int k = 0; k < 3; ++k) { b.mn[k] = boxes[6 * q + k]; b.mx[k] = boxes[6 * q + 3 + k]; }
        if (b.area() >= 0.5 * whole_area) { pre.push_back(words[q]); is_pre[q] = 1; }
      }
    }
    Box3d cen;
    cen.reset();
    if (pre.empty()) {
      for (const Box3d& b : centres_of) cen.grow(b);
    } else {  // compact the leaves that stay, and take the bounds of their centres
      size_t w = 0;
      for (size_t q = 0; q < n_keep; ++q) {
        if (is_pre[q]) continue;
        for (int k = 0; k < 6; ++k) boxes[6 * w + k] = boxes[6 * q + k];
        words[w] = words[q];
        Box3d c1;
        for (int k = 0; k < 3; ++k) c1.mn[k] = c1.mx[k] = 0.5 * ((double)boxes[6 * w + k] + (double)boxes[6 * w + 3 + k]);
        cen.grow(c1);
        ++w;
      }
      boxes.resize(6 * w); words.resize(w);
    }
    mark("own tree: boxes (host threads)");
    if (words.size() >= 2) {
      int need = 0;
      int brc = gpu_build_wide(c, boxes.data(), words.data(), words.size(), cen.mn, cen.mx, &device_tree.nodes, &device_tree.n, &need);
      if (brc != RT_OK) return brc;
      if (timing) std::fprintf(stderr, "rt_upload_scene: device-built four-wide tree of %u nodes, stack need %d (limit %u)\n", device_tree.n, need, kWideStack);
      if (need > (int)kWideStack) {  // the host builder's tree is shallower
        (void)hipFree(device_tree.nodes);
        device_tree.nodes = nullptr; device_tree.n = 0;
      } else {
        wide_pre = pre;
        // a list of one box and one leaf stands where the own tree's list would be
        TmpLink b{};
        for (int k = 0; k < 3; ++k) { b.bmin[k] = whole.mn[k]; b.bmax[k] = whole.mx[k]; }
        b.hit = 1u; b.miss = LINK_END; b.depth = 0; b.tree = 0;
        TmpLink l{};
        l.hit = words[0]; l.miss = LINK_END; l.depth = 1; l.tree = 0;
        fast_links = {b, l};
        fast_root = 0;
        fast_sub_root.assign((size_t)s->n_objects, LINK_END);
      }
      mark("own tree: built on the device");
    }
  }
  if (own_tree && !device_tree.nodes) {
    FastTreeBuilder ftb(scene32);
    ftb.build(lb.links, root, sub_root, dissolved, fast_links, fast_root, fast_sub_root);
    if (!ftb.wide.empty() && ftb.wide_stack_need <= (int)kWideStack) { wide_nodes.swap(ftb.wide); wide_pre.swap(ftb.wide_pre); }
    if (timing) std::fprintf(stderr, "rt_upload_scene: four-wide tree of %zu nodes, stack need %d (limit %u)\n", wide_nodes.size(), ftb.wide_stack_need, kWideStack);
    mark("own tree");
    fast_links = reorder(fast_links, fast_root, fast_sub_root);
    mark("own tree: storage order");
  }
  std::vector<TmpLink> links = reorder(lb.links, root, sub_root);
  mark("reference tree: storage order");
  if (!order_err.empty()) return fail(RT_E_INVALID, order_err);

  // images
  if (s->n_images > 0) {
    std::vector<DImage> host_images((size_t)s->n_images);
    for (int i = 0; i < s->n_images; ++i) {
      size_t bytes = (size_t)s->images[i].width * s->images[i].height * 3;
      uint8_t* d = nullptr;
      HIP_TRY(hipMalloc((void**)&d, bytes));
      c->d_image_bytes.push_back(d);
      HIP_TRY(hipMemcpy(d, s->images[i].rgb, bytes, hipMemcpyHostToDevice));
      host_images[i] = DImage{d, s->images[i].width, s->images[i].height};
    }
    HIP_TRY(hipMalloc((void**)&c->d_images, sizeof(DImage) * host_images.size()));
    HIP_TRY(hipMemcpy(c->d_images, host_images.data(), sizeof(DImage) * host_images.size(), hipMemcpyHostToDevice));
  }
  // Double precision, default image.  The closest hit does not depend on the tree that is walked, only on the leaf tests, unless two
  // different surfaces lie at exactly the same distance; so where the library has a four-wide tree of its own (every nested tree
  // dissolved, the walk's stack fits) the double-precision kernels walk that one as well, with the reference's leaf arithmetic on the
  // caller's objects (spliced nested trees keep their holders' wrappers: the same operations bit for bit), flag the rays on which
  // two leaves tie and have those traced by the list walk of the caller's tree, which the image carries too (F_EXACT, kernels.h).
  // Kept on the list walk: scenes in which a texture reads a leftover (u, v) (the leftover depends on the order of the tests),
  // scenes with coincident objects (every ray on them would tie), and RTAMD_F64_LIST=1.
  const bool have_wide_tree = !wide_nodes.empty() || device_tree.nodes != nullptr;
  const bool exact_own = own_tree && have_wide_tree && !stale_uv_read && !coincident && !nested_left64 && !std::getenv("RTAMD_F64_LIST");
  // RT_F64_OWN_TREE: the library's tree in double with its own cheaper leaf forms (boxes as slabs, re-centred spheres), compared with
  // the reference by tolerance; for sphere scenes that is the default image itself
  const bool want_own64 = own_tree && preset32 != PRESET_BALLS_LDS && !std::getenv("RTAMD_NO_F64_OWN_TREE");
  int rc_sd = RT_OK, rc_own = RT_OK;
  std::string err_sd, err_own;
  const int device = c->device;
  std::thread t_sd([&] {
    if (hipSetDevice(device) != hipSuccess) { rc_sd = RT_E_HIP; err_sd = "hipSetDevice failed"; return; }
    const std::vector<WideTmp>* w = exact_own ? &wide_nodes : nullptr;
    const std::vector<uint32_t>* p = exact_own ? &wide_pre : nullptr;
    const WideNode* dw = exact_own ? device_tree.nodes : nullptr;
    rc_sd = relinked ? build_image<double>(c, scene64, links64, root64, sub_root64, preset64, c->sd, w, p, exact_own, dw, device_tree.n)
                     : build_image<double>(c, *s, links, root, sub_root, preset, c->sd, w, p, exact_own, dw, device_tree.n);
    if (rc_sd != RT_OK) err_sd = g_err;
  });
  std::thread t_own([&] {
    if (!want_own64) return;
    if (hipSetDevice(device) != hipSuccess) { rc_own = RT_E_HIP; err_own = "hipSetDevice failed"; return; }
    rc_own = build_image<double>(c, scene32, fast_links, fast_root, fast_sub_root, preset32, c->sd_own, &wide_nodes, &wide_pre, false, device_tree.nodes, device_tree.n);
    if (rc_own != RT_OK) err_own = g_err;
  });
  rc = own_tree ? build_image<float>(c, scene32, fast_links, fast_root, fast_sub_root, preset32, c->sf, &wide_nodes, &wide_pre, false, device_tree.nodes, device_tree.n)
                : build_image<float>(c, *s, links, root, sub_root, preset, c->sf, nullptr, nullptr, false);
  t_sd.join();
  t_own.join();
  if (rc == RT_OK && rc_sd != RT_OK) rc = fail(rc_sd, err_sd);
  if (rc == RT_OK && rc_own != RT_OK) rc = fail(rc_own, err_own);
  mark("f32, f64, f64 own-tree images");
  if (rc != RT_OK) { free_scene(c); return rc; }
  c->has_scene = true;
  return RT_OK;
}

int rt_render_accum(rt_context* c, const rt_render_params* p, void* accum_dev, rt_stats* stats) {
  int rc = check_render_params(c, p);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if (accum_dev)
    if (int bad = check_device_buffer(accum_dev, "rt_render_accum")) return bad;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  // NULL = the context's own buffer, cleared (and possibly reallocated) by this call: a progressive frame that was being
  // refined in it is over, and its next rt_progress_step says so instead of resolving a picture of the wrong sums
  if (!accum_dev) c->progress_active = false;
  rt_render_params q = *p;
  q.precision = select_tree(c, p->precision);
  if (q.precision == RT_PRECISION_F64) return render_accum_impl<double>(c, q, accum_dev, stats);
  return render_accum_impl<float>(c, q, accum_dev, stats);
}

int rt_resolve(rt_context* c, const void* accum_dev, uint32_t width, uint32_t height, uint32_t spp, int precision, uint8_t* rgb8_out) {
  if (!c || !rgb8_out) return fail(RT_E_INVALID, "context or output is NULL");
  precision &= ~RT_F64_OWN_TREE;
  if (spp == 0) return fail(RT_E_INVALID, "samples_per_pixel is 0");
  if (precision != RT_PRECISION_F32 && precision != RT_PRECISION_F64) return fail(RT_E_INVALID, "unknown precision");
  HIP_TRY(hipSetDevice(c->device));
  const size_t n = (size_t)width * height;
  const size_t need = n * 4 * (precision == RT_PRECISION_F64 ? sizeof(double) : sizeof(float));
  if (!accum_dev) {
    if (!c->accum || c->accum_bytes < need) return fail(RT_E_INVALID, "the context holds no accumulation buffer of this size");
    accum_dev = c->accum;
  }
  // the RGB8 image between the conversion kernel and the copy to the host: kept by the context and grown on demand, so that a
  // render loop (one rt_resolve per frame) allocates nothing after its first frame
  if (c->rgb_bytes < n * 3) {
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    c->d_rgb = nullptr; c->rgb_bytes = 0;
    HIP_TRY(hipMalloc((void**)&c->d_rgb, n * 3));
    c->rgb_bytes = n * 3;
  }
  uint8_t* const d_rgb = c->d_rgb;
  int rrc = precision == RT_PRECISION_F64 ? resolve_impl<double>(c, accum_dev, n, spp, d_rgb) : resolve_impl<float>(c, accum_dev, n, spp, d_rgb);
  if (rrc) return rrc;
  hipError_t e = hipMemcpyAsync(rgb8_out, d_rgb, n * 3, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(RT_E_HIP, std::string("rt_resolve: ") + hipGetErrorString(e));
  return RT_OK;
}

int rt_read_accum(rt_context* c, const void* accum_dev, uint32_t width, uint32_t height, int precision, double* out) {
  if (!c || !out) return fail(RT_E_INVALID, "context or output is NULL");
  precision &= ~RT_F64_OWN_TREE;
  if (precision != RT_PRECISION_F32 && precision != RT_PRECISION_F64) return fail(RT_E_INVALID, "unknown precision");
  HIP_TRY(hipSetDevice(c->device));
  const size_t n = (size_t)width * height;
  const size_t esz = precision == RT_PRECISION_F64 ? sizeof(double) : sizeof(float);
  if (!accum_dev) {
    if (!c->accum || c->accum_bytes < n * 4 * esz) return fail(RT_E_INVALID, "the context holds no accumulation buffer of this size");
    accum_dev = c->accum;
  }
  std::vector<uint8_t> tmp(n * 4 * esz);
  HIP_TRY(hipMemcpyAsync(tmp.data(), accum_dev, tmp.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k)
      out[i * 3 + k] = precision == RT_PRECISION_F64 ? reinterpret_cast<const double*>(tmp.data())[i * 4 + k]
                                                      : (double)reinterpret_cast<const float*>(tmp.data())[i * 4 + k];
  return RT_OK;
}

int rt_render(rt_context* c, const rt_render_params* p, uint8_t* rgb8_out, double* accum_out, rt_stats* stats) {
  if (!rgb8_out && !accum_out) return fail(RT_E_INVALID, "no output buffer given");
  auto t0 = std::chrono::steady_clock::now();
  int rc = rt_render_accum(c, p, nullptr, stats);
  if (rc) return rc;
  if (p->sample_end == p->sample_begin && rgb8_out) return fail(RT_E_INVALID, "cannot resolve an image of zero samples");
  if (rgb8_out) rc = rt_resolve(c, nullptr, p->width, p->height, p->sample_end - p->sample_begin, p->precision, rgb8_out);
  if (rc == RT_OK && accum_out) rc = rt_read_accum(c, nullptr, p->width, p->height, p->precision, accum_out);
  if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int rt_progress_begin(rt_context* c, const rt_render_params* p) {
  int rc = check_render_params(c, p);
  if (rc) return rc;
  if (p->sample_end == p->sample_begin) return fail(RT_E_INVALID, "cannot resolve an image of zero samples");
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = (size_t)p->width * p->height * 4 * ((p->precision & ~RT_F64_OWN_TREE) == RT_PRECISION_F64 ? sizeof(double) : sizeof(float));
  rt_render_params none = *p;
  none.sample_end = none.sample_begin;  // renders nothing: allocates (if need be) and clears the context's own buffer
  rc = rt_render_accum(c, &none, nullptr, nullptr);
  if (rc) return rc;
  if (!c->accum || c->accum_bytes < bytes) return fail(RT_E_HIP, "rt_progress_begin: no accumulation buffer");
  c->progress = *p;
  c->progress_next = p->sample_begin;
  c->progress_active = true;
  return RT_OK;
}

int rt_progress_step(rt_context* c, uint32_t max_samples, uint8_t* rgb8_out, uint32_t* samples_done, int* finished, rt_stats* stats) {
  if (!c || !rgb8_out) return fail(RT_E_INVALID, "context or output is NULL");
  if (!c->progress_active)
    return fail(RT_E_INVALID, "rt_progress_step without a frame in progress (no rt_progress_begin, or a scene upload / another render into the context's own "
                              "buffer since then)");
  if (max_samples == 0) return fail(RT_E_INVALID, "max_samples is 0");
  if (!c->has_scene) return fail(RT_E_NO_SCENE, "no scene uploaded");
  const rt_render_params& f = c->progress;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (c->progress_next < f.sample_end) {
    rt_render_params slice = f;
    slice.sample_begin = c->progress_next;
    slice.sample_end = f.sample_end - c->progress_next > max_samples ? c->progress_next + max_samples : f.sample_end;
    HIP_TRY(hipSetDevice(c->device));
    // into the context's own buffer WITHOUT clearing it: hand its address over as a caller's buffer
    int rc = rt_render_accum(c, &slice, c->accum, stats);
    if (rc) return rc;
    c->progress_next = slice.sample_end;
  }
  const uint32_t done = c->progress_next - f.sample_begin;
  int rc = rt_resolve(c, nullptr, f.width, f.height, done, f.precision, rgb8_out);
  if (rc) return rc;
  if (samples_done) *samples_done = done;
  if (finished) *finished = c->progress_next >= f.sample_end ? 1 : 0;
  return RT_OK;
}

int rt_render_noalloc(rt_context* c, const rt_render_params* p, void* accum_dev, uint8_t* rgb8_out, rt_stats* stats) {
  int rc = check_render_params(c, p);
  if (rc) return rc;
  if (!accum_dev || !rgb8_out) return fail(RT_E_INVALID, "rt_render_noalloc: accum_dev and rgb8_out are required (this entry point owns no buffers)");
  if (p->pipeline != RT_PIPELINE_AUTO && p->pipeline != RT_PIPELINE_PERSISTENT)
    return fail(RT_E_UNSUPPORTED, "rt_render_noalloc: the queue pipelines need a workspace; only RT_PIPELINE_AUTO / RT_PIPELINE_PERSISTENT");
  const size_t n = (size_t)p->width * p->height;
  if (n > (size_t)RT_NOALLOC_MAX_PIXELS)
    return fail(RT_E_INVALID, "rt_render_noalloc: frame larger than the static scratch holds (at most 3840*2160 = 8294400 pixels)");
  if (p->sample_end == p->sample_begin) return fail(RT_E_INVALID, "cannot resolve an image of zero samples");
  if (!c->scratch) return fail(RT_E_HIP, "rt_render_noalloc: no scratch address for this device");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(c->device));
  if (int bad = check_device_buffer(accum_dev, "rt_render_noalloc")) return bad;  // a query, not an allocation
  if (stats) std::memset(stats, 0, sizeof(*stats));
  rt_render_params q = *p;
  q.precision = select_tree(c, p->precision);
  const bool f64 = q.precision == RT_PRECISION_F64;
  unsigned char* const scratch = static_cast<unsigned char*>(c->scratch);  // the device address, never the array's name
  unsigned long long* const counters = reinterpret_cast<unsigned long long*>(scratch);
  uint8_t* const d_rgb = scratch + RT_NOALLOC_COUNTER_BYTES;
  // the whole frame, as rt_render: sums start at zero; the counters are cleared inside (the array keeps the last call's contents)
  HIP_TRY(hipMemsetAsync(accum_dev, 0, n * 4 * (f64 ? sizeof(double) : sizeof(float)), c->stream));
  rc = f64 ? render_accum_impl<double>(c, q, accum_dev, stats, counters, c->h_rays) : render_accum_impl<float>(c, q, accum_dev, stats, counters, c->h_rays);
  if (rc) return rc;
  const uint32_t spp = p->sample_end - p->sample_begin;
  rc = f64 ? resolve_impl<double>(c, accum_dev, n, spp, d_rgb) : resolve_impl<float>(c, accum_dev, n, spp, d_rgb);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(rgb8_out, d_rgb, n * 3, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return RT_OK;
}

int rt_trace_batch(rt_context* c, const rt_ray* rays, size_t n, rt_hit* hits, int precision, uint64_t seed) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  if (!c->has_scene) return fail(RT_E_NO_SCENE, "no scene uploaded");
  if (n == 0) return RT_OK;
  if (!rays || !hits) return fail(RT_E_INVALID, "rays or hits is NULL");
  if (n > 0x7FFFFFFFull) return fail(RT_E_INVALID, "batch too large");
  const int mode = (precision & RT_TRACE_VISITS) ? 2 : ((precision & RT_TRACE_AS_RENDERED) ? 1 : 0);
  precision &= ~(RT_TRACE_AS_RENDERED | RT_TRACE_VISITS);
  precision = select_tree(c, precision);
  if (precision != RT_PRECISION_F32 && precision != RT_PRECISION_F64) return fail(RT_E_INVALID, "unknown precision");
  HIP_TRY(hipSetDevice(c->device));
  return precision == RT_PRECISION_F64 ? trace_batch_impl<double>(c, rays, n, hits, seed, mode)
                                       : trace_batch_impl<float>(c, rays, n, hits, seed, mode);
}

int rt_camera_rays(rt_context* c, uint32_t width, uint32_t height, const uint32_t* pixel, const uint32_t* sample, size_t n, uint64_t seed,
                   int precision, rt_ray* out) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  if (!c->has_scene) return fail(RT_E_NO_SCENE, "no scene uploaded");
  precision = select_tree(c, precision);
  if (n == 0) return RT_OK;
  if (!pixel || !sample || !out) return fail(RT_E_INVALID, "NULL argument");
  if (width < 2 || height < 2) return fail(RT_E_INVALID, "width and height must be at least 2");
  for (size_t i = 0; i < n; ++i)
    if (pixel[i] >= (uint64_t)width * height) return fail(RT_E_INVALID, "pixel index out of range");
  HIP_TRY(hipSetDevice(c->device));
  return precision == RT_PRECISION_F64 ? pairs_kernel_impl<double>(c, 0, width, height, 0, pixel, sample, n, seed, out, n * sizeof(rt_ray))
                                       : pairs_kernel_impl<float>(c, 0, width, height, 0, pixel, sample, n, seed, out, n * sizeof(rt_ray));
}

int rt_sample_radiance(rt_context* c, uint32_t width, uint32_t height, uint32_t max_depth, const uint32_t* pixel, const uint32_t* sample, size_t n,
                       uint64_t seed, int precision, double* out) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  if (!c->has_scene) return fail(RT_E_NO_SCENE, "no scene uploaded");
  precision = select_tree(c, precision);
  if (n == 0) return RT_OK;
  if (!pixel || !sample || !out) return fail(RT_E_INVALID, "NULL argument");
  if (width < 2 || height < 2) return fail(RT_E_INVALID, "width and height must be at least 2");
  for (size_t i = 0; i < n; ++i)
    if (pixel[i] >= (uint64_t)width * height) return fail(RT_E_INVALID, "pixel index out of range");
  HIP_TRY(hipSetDevice(c->device));
  return precision == RT_PRECISION_F64
             ? pairs_kernel_impl<double>(c, 1, width, height, max_depth, pixel, sample, n, seed, out, n * 4 * sizeof(double))
             : pairs_kernel_impl<float>(c, 1, width, height, max_depth, pixel, sample, n, seed, out, n * 4 * sizeof(double));
}

int rt_sample_path(rt_context* c, uint32_t width, uint32_t height, uint32_t max_depth, const uint32_t* pixel, const uint32_t* sample, size_t n,
                   uint64_t seed, int precision, uint32_t max_vertices, double* out) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  if (!c->has_scene) return fail(RT_E_NO_SCENE, "no scene uploaded");
  precision = select_tree(c, precision);
  if (n == 0) return RT_OK;
  if (!pixel || !sample || !out || max_vertices == 0) return fail(RT_E_INVALID, "NULL argument or max_vertices == 0");
  if (width < 2 || height < 2) return fail(RT_E_INVALID, "width and height must be at least 2");
  for (size_t i = 0; i < n; ++i)
    if (pixel[i] >= (uint64_t)width * height) return fail(RT_E_INVALID, "pixel index out of range");
  HIP_TRY(hipSetDevice(c->device));
  size_t bytes = n * (4 + (size_t)max_vertices * 12) * sizeof(double);
  return precision == RT_PRECISION_F64 ? pairs_kernel_impl<double>(c, 2, width, height, max_depth, pixel, sample, n, seed, out, bytes)
                                       : pairs_kernel_impl<float>(c, 2, width, height, max_depth, pixel, sample, n, seed, out, bytes);
}

}  // extern "C"
