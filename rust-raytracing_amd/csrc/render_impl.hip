// Host-side render loop and test entry points for both arithmetic types.  The kernels themselves are reached through
// the per-preset launcher tables of kernel_table.h.
#include <memory>
#include <limits>
#include "context.h"
#include "kernel_table.h"

namespace rti {

template <class R> KernelTable<R> table_for(int preset) {
  switch (preset) {
    case PRESET_BOXES: return kernel_table<R, PRESET_BOXES>();
    case PRESET_SMOKE: return kernel_table<R, PRESET_SMOKE>();
    case PRESET_SPHERES: return kernel_table<R, PRESET_SPHERES>();
    case PRESET_ALL_LDS: return kernel_table<R, PRESET_ALL_LDS>();
    case PRESET_BALLS_LDS: return kernel_table<R, PRESET_BALLS_LDS>();
    case PRESET_BALLS_TOP: return kernel_table<R, PRESET_BALLS_TOP>();
    case PRESET_FLATALL_LDS: return kernel_table<R, PRESET_FLATALL_LDS>();
    case PRESET_FLATALL_TOP: return kernel_table<R, PRESET_FLATALL_TOP>();
    case PRESET_BALLS_WIDE: return kernel_table<R, PRESET_BALLS_WIDE>();
    case PRESET_FLATALL_WIDE: return kernel_table<R, PRESET_FLATALL_WIDE>();
    case PRESET_BALLS_WIDE_EXACT:  // double precision only: a single-precision image never asks for them
      if constexpr (sizeof(R) == 8) return kernel_table<R, PRESET_BALLS_WIDE_EXACT>();
      else return kernel_table<R, PRESET_BALLS_WIDE>();
    case PRESET_FLATALL_WIDE_EXACT:
      if constexpr (sizeof(R) == 8) return kernel_table<R, PRESET_FLATALL_WIDE_EXACT>();
      else return kernel_table<R, PRESET_FLATALL_WIDE>();
    default: break;
  }
  return kernel_table<R, PRESET_ALL_TOP>();
}

static_assert(sizeof(RayIO) == sizeof(rt_ray), "rt_ray layout");
static_assert(sizeof(HitIO) == sizeof(rt_hit), "rt_hit layout");
static_assert(sizeof(DLink<float>) == 32 && sizeof(DLink<double>) == 64, "link size");
static_assert(sizeof(WideNode) == 128, "wide node size");


template <class R> R round_down(double x);
template <class R> R round_up(double x);
template <> inline double round_down<double>(double x) { return x; }
template <> inline double round_up<double>(double x) { return x; }
// single-precision boxes must still contain the double-precision ones
template <> inline float round_down<float>(double x) {
  float f = (float)x;
  return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f;
}
template <> inline float round_up<float>(double x) {
  float f = (float)x;
  return ((double)f < x) ? std::nextafterf(f, INFINITY) : f;
}


// Byte offsets of the blob's sections from the counts in `lay` (every section 16-byte aligned).
template <class R> void layout_sections(BlobLayout& lay) {
  uint32_t off = 0;
  lay.links_off = off; off = align16(off + lay.n_links * (uint32_t)sizeof(BoxLink));
  lay.shapes_off = off; off = align16(off + lay.n_objects * (uint32_t)sizeof(DShape<R>));
  lay.objects_off = off; off = align16(off + lay.n_objects * (uint32_t)sizeof(DObject<R>));
  lay.materials_off = off; off = align16(off + lay.n_materials * (uint32_t)sizeof(DMaterial<R>));
  lay.textures_off = off; off = align16(off + lay.n_textures * (uint32_t)sizeof(DTexture<R>));
  lay.perlins_off = off; off = align16(off + lay.n_perlins * (uint32_t)sizeof(DPerlin<R>));
  lay.flat_off = off; off = align16(off + (uint32_t)(std::max(1u, lay.n_flat) * sizeof(FlatLeaf<R>)));
  lay.frect_off = off; off = align16(off + (uint32_t)(kFlatMaxLeaves * sizeof(FlatRect)));
  lay.faces_off = off;
  if (lay.flat_mode) off = align16(off + (uint32_t)(6u * lay.n_objects * sizeof(FlatFace<R>)));
  lay.camera_off = off; off = align16(off + (uint32_t)sizeof(CameraRecord<R>));
  lay.wide_off = (off + 127u) & ~127u;
  if (lay.n_wide) off = lay.wide_off + lay.n_wide * (uint32_t)sizeof(WideNode);
  lay.total_bytes = off;
}

#ifdef RT_WIDE_Q8
// Measurement build: WideNode -> WideNodeQ, conservatively (a decoded box contains the original).
__global__ void k_quantize_wide(const WideNode* src, uint32_t n, WideNodeQ* dst) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const WideNode nd = src[k];
  WideNodeQ q;
  for (int a = 0; a < 3; ++a) {
    float lo = INFINITY, hi = -INFINITY;
    for (int ch = 0; ch < 4; ++ch)
      if (nd.child[ch] != kWideNone) { lo = fminf(lo, nd.lo[a][ch]); hi = fmaxf(hi, nd.hi[a][ch]); }
    if (!(hi >= lo)) { lo = 0.0f; hi = 0.0f; }
    float cell = (hi - lo) / 254.0f;
    if (!(cell > 0.0f)) cell = 1e-30f;
    q.org[a] = lo; q.cell[a] = cell;
    q.qlo[a] = 0u; q.qhi[a] = 0u;
    for (int ch = 0; ch < 4; ++ch) {
      uint32_t ql = 0u, qh = 255u;
      if (nd.child[ch] != kWideNone) {
        ql = (uint32_t)fminf(fmaxf(floorf((nd.lo[a][ch] - lo) / cell), 0.0f), 255.0f);
        while (ql > 0u && lo + (float)ql * cell > nd.lo[a][ch]) --ql;
        qh = (uint32_t)fminf(fmaxf(ceilf((nd.hi[a][ch] - lo) / cell), 0.0f), 255.0f);
        while (qh < 255u && lo + (float)qh * cell < nd.hi[a][ch]) ++qh;
      }
      q.qlo[a] |= ql << (8 * ch);
      q.qhi[a] |= qh << (8 * ch);
    }
  }
  for (int ch = 0; ch < 4; ++ch) q.child[ch] = nd.child[ch];
  dst[k] = q;
}
#endif

template <class R>
int build_image(rt_context* c, const rt_scene_desc& s, const std::vector<TmpLink>& links, int root_link, const std::vector<int>& sub_root_of_object,
                int preset, SceneImage<R>& im, const std::vector<WideTmp>* wide, const std::vector<uint32_t>* pre, bool exact,
                const WideNode* device_wide, uint32_t n_device_wide) {
  BlobLayout lay{};
  lay.n_links = (uint32_t)links.size();
  lay.n_objects = (uint32_t)s.n_objects;
  lay.n_materials = (uint32_t)s.n_materials;
  lay.n_textures = (uint32_t)std::max(1, s.n_textures);
  lay.n_perlins = (uint32_t)s.n_perlins;
  // leaf words in visiting order (the small-scene presets store their links in that order)
  std::vector<uint32_t> flat;
  const bool uniform_preset = preset == PRESET_BOXES || preset == PRESET_SMOKE || preset == PRESET_SPHERES;
  constexpr bool single = sizeof(R) == sizeof(float);
  if (uniform_preset) {
    for (const TmpLink& l : links)
      if (l.hit & LEAF_FLAG) flat.push_back(l.hit);
    // a tree may hold the same object more than once (a one-object tree tests it twice, src/bvh.rs:38): testing it again gives
    // the same distance, which the closed range accepts again, so only its last visit is kept (the second visit of a medium
    // carries LEAF_SECOND, is a different word, and stays)
    for (size_t k = 0; k < flat.size();) {
      bool again = false;
      for (size_t m = k + 1; m < flat.size(); ++m) again = again || flat[m] == flat[k];
      if (again) flat.erase(flat.begin() + (long)k); else ++k;
    }
    if (flat.size() > kFlatMaxLeaves) return fail(RT_E_INVALID, "internal error: a small-scene preset was chosen for a scene with too many leaves");
  }
  // single precision: plain rectangles leave the list of leaf words for per-axis lists of ready-made records
  std::vector<FlatRect> frects[3];
  std::vector<uint32_t> others;
  for (uint32_t word : flat) {
    const int obj = (int)(word & LEAF_INDEX_MASK);
    const rt_object& o = s.objects[obj];
    const rt_shape& sh = s.shapes[o.geom];
    const int axis = sh.tag == RT_SHAPE_RECT_YZ ? 0 : sh.tag == RT_SHAPE_RECT_XZ ? 1 : sh.tag == RT_SHAPE_RECT_XY ? 2 : -1;
    if (single && (word & LEAF_SIMPLE) && axis >= 0 && !getenv("RTAMD_NO_FLAT_RECTS")) {
      FlatRect r{};
      r.a0 = (float)sh.p[0]; r.a1 = (float)sh.p[1]; r.b0 = (float)sh.p[2]; r.b1 = (float)sh.p[3]; r.k = (float)sh.p[4];
      r.obj = obj;
      frects[axis].push_back(r);
    } else {
      others.push_back(word);
    }
  }
  lay.flat_mode = flat.empty() ? 0u : 1u;
  lay.n_flat = (uint32_t)others.size();
  for (int k = 0; k < 3; ++k) lay.n_frect[k] = (uint32_t)frects[k].size();
  layout_sections<R>(lay);
  // The four-wide form of the library's own tree replaces the threaded list where the image does not fit in LDS (there the list
  // is walked from LDS and is the faster of the two).  RTAMD_WIDE_ALWAYS=1: wherever a wide tree exists (measurement switch).
  // (a tree built on the device comes without a list: it is walked whatever the image's size)
  const bool have_wide = device_wide ? n_device_wide > 0 : (wide && !wide->empty());
  if (have_wide && (preset == PRESET_BALLS_LDS || preset == PRESET_FLATALL_LDS) &&
      (lay.total_bytes > kLdsAllMax || device_wide || getenv("RTAMD_WIDE_ALWAYS"))) {
    // `exact`: the four-wide tree stands in for the list in `links` (the caller's tree): the kernels with the reference's leaf
    // arithmetic and tie rule (F_EXACT)
    preset = preset == PRESET_BALLS_LDS ? (exact ? PRESET_BALLS_WIDE_EXACT : PRESET_BALLS_WIDE) : (exact ? PRESET_FLATALL_WIDE_EXACT : PRESET_FLATALL_WIDE);
    lay.n_wide = device_wide ? n_device_wide : (uint32_t)wide->size();
    layout_sections<R>(lay);
  } else if (device_wide) {
    return fail(RT_E_INVALID, "internal error: a tree built on the device for a preset that does not walk it");
  }
  // the host's part of the blob: all of it, or everything in front of a four-wide tree that is already on the device
  const bool wide_on_device = device_wide && lay.n_wide;
  struct RawBlob {  // zero-filled on the host's threads (a std::vector would fill a quarter of a gigabyte from one)
    std::unique_ptr<uint8_t[]> mem;
    size_t bytes;
    explicit RawBlob(size_t n) : mem(new uint8_t[n]), bytes(n) {
      uint8_t* base = mem.get();
      parallel_chunks(n, [base](size_t, size_t lo_, size_t hi_) { std::memset(base + lo_, 0, hi_ - lo_); }, (size_t)4 << 20);
    }
    uint8_t* data() { return mem.get(); }
    size_t size() const { return bytes; }
  } blob(wide_on_device ? lay.wide_off : lay.total_bytes);
  if (lay.n_wide && !wide_on_device) {
    WideNode* W = reinterpret_cast<WideNode*>(blob.data() + lay.wide_off);
    const float inf = std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < wide->size(); ++i) {
      const WideTmp& w = (*wide)[i];
      // a child takes part unless its slot is unused or its box is inverted along some axis (see the links below: no ray passes)
      bool used[4];
      for (int ch = 0; ch < 4; ++ch)
        used[ch] = w.child[ch] != kWideNone && !(w.lo[ch][0] > w.hi[ch][0] || w.lo[ch][1] > w.hi[ch][1] || w.lo[ch][2] > w.hi[ch][2]);
      WideNode& n = W[i];
      std::memset(&n, 0, sizeof(n));
      for (int ch = 0; ch < 4; ++ch) {
        n.child[ch] = used[ch] ? w.child[ch] : kWideNone;
        // "plain shape, no wrapper" is a property of this image's objects (the tree may have been built over another image's, in
        // which a sphere under wrappers was re-centred and became plain)
        if (used[ch] && (w.child[ch] & LEAF_FLAG))
          n.child[ch] = (w.child[ch] & ~LEAF_SIMPLE) | ((s.objects[w.child[ch] & LEAF_INDEX_MASK].flags & 15u) == 0u ? LEAF_SIMPLE : 0u);
        for (int k = 0; k < 3; ++k) {
          n.lo[k][ch] = used[ch] ? round_down<float>(w.lo[ch][k]) : inf;
          n.hi[k][ch] = used[ch] ? round_up<float>(w.hi[ch][k]) : inf;
        }
        // see the links below: a re-centred sphere's front_face rule is read from the object's flags (such spheres come with wrapped
        // objects, so the single-precision walk of their preset builds its records after the walk and reads the flags there)
        if (!single && used[ch] && (w.child[ch] & LEAF_FLAG) && (s.objects[w.child[ch] & LEAF_INDEX_MASK].flags & OBJ_FRONT_CHAIN)) n.child[ch] &= ~LEAF_SIMPLE;
      }
    }
  }
  if (lay.flat_mode) {
    FlatFace<R>* F = reinterpret_cast<FlatFace<R>*>(blob.data() + lay.faces_off);
    for (int i = 0; i < s.n_objects; ++i) {
      const rt_object& o = s.objects[i];
      if (o.flags & RT_OBJ_SUB_BVH) continue;
      const rt_shape& sh = s.shapes[o.geom];
      const bool box = sh.tag == RT_SHAPE_BOX;
      if (!box && sh.tag != RT_SHAPE_RECT_XY && sh.tag != RT_SHAPE_RECT_XZ && sh.tag != RT_SHAPE_RECT_YZ) continue;
      const bool rot = (o.flags & RT_OBJ_ROTATE_Y) != 0;
      const R cs = rot ? (R)o.cos_theta : (R)1, sn = rot ? (R)o.sin_theta : (R)0;
      for (int side = 0; side < (box ? 6 : 1); ++side) {
        // box sides: 0 z=hi, 1 z=lo, 2 y=hi, 3 y=lo, 4 x=hi, 5 x=lo (the reference's list order, src/box_struct.rs:22-71)
        const int axis = box ? (side < 2 ? 2 : side < 4 ? 1 : 0) : (sh.tag == RT_SHAPE_RECT_XY ? 2 : sh.tag == RT_SHAPE_RECT_XZ ? 1 : 0);
        static const int box_k[6] = {5, 2, 4, 1, 3, 0};
        FlatFace<R>& f = F[6 * i + side];
        f.n[0] = axis == 0 ? cs : axis == 2 ? sn : (R)0;   // RotateY of the axis vector (src/hittable.rs:229-239)
        f.n[1] = axis == 1 ? (R)1 : (R)0;
        f.n[2] = axis == 0 ? -sn : axis == 2 ? cs : (R)0;
        f.flags = (o.flags & 15u) | ((uint32_t)axis << 4);
        f.cos_theta = cs; f.sin_theta = sn;
        f.k = (R)(box ? sh.p[box_k[side]] : sh.p[4]);
      }
    }
  }
  {
    FlatLeaf<R>* fl = reinterpret_cast<FlatLeaf<R>*>(blob.data() + lay.flat_off);
    for (uint32_t word : others) {
      const rt_object& o = s.objects[word & LEAF_INDEX_MASK];
      FlatLeaf<R>& f = *fl++;
      f.word = word; f.tag = -1; f.obj_id = o.obj_id; f.flags = o.flags;
      if (!(o.flags & RT_OBJ_SUB_BVH)) {
        const rt_shape& sh = s.shapes[o.geom];
        f.tag = sh.tag;
        for (int k = 0; k < 9; ++k) f.p[k] = (R)sh.p[k];
      }
      for (int k = 0; k < 3; ++k) f.offset[k] = (o.flags & RT_OBJ_TRANSLATE) ? (R)o.offset[k] : (R)0;  // subtracted unconditionally on the device
      f.sin_theta = (R)o.sin_theta; f.cos_theta = (R)o.cos_theta; f.neg_inv_density = (R)o.neg_inv_density;
    }
  }
  {
    FlatRect* fr = reinterpret_cast<FlatRect*>(blob.data() + lay.frect_off);
    for (int k = 0; k < 3; ++k)
      for (const FlatRect& r : frects[k]) *fr++ = r;
  }
  auto* L = reinterpret_cast<BoxLink*>(blob.data() + lay.links_off);  // single-precision boxes in both images (see BoxLink)
  parallel_chunks(links.size(), [&](size_t, size_t lo_, size_t hi_) {
  for (size_t i = lo_; i < hi_; ++i) {
    for (int k = 0; k < 3; ++k) { L[i].bmin[k] = round_down<float>(links[i].bmin[k]); L[i].bmax[k] = round_up<float>(links[i].bmax[k]); }
    // A box that is inverted along some axis (the node of a lone sphere of negative radius, src/sphere.rs:75-81) fails the
    // reference's test for every ray (src/aabb.rs:38-45: t_max <= t_min on that axis).  It is stored as the point at +infinity,
    // which fails both forms of the test the kernels use - the reference's as written and the min/max form, which would
    // otherwise read an inverted box as the ordinary one.
    if (!(links[i].hit & LEAF_FLAG) && (links[i].bmin[0] > links[i].bmax[0] || links[i].bmin[1] > links[i].bmax[1] || links[i].bmin[2] > links[i].bmax[2]))
      for (int k = 0; k < 3; ++k) L[i].bmin[k] = L[i].bmax[k] = std::numeric_limits<float>::infinity();
    L[i].hit = links[i].hit;
    L[i].miss = links[i].miss;
    // the double-precision walk looks at an object's flags only when its leaf is not marked simple; a re-centred sphere's
    // front_face rule lives in those flags (the single-precision walk reads them when it writes the record)
    if (!single && (links[i].hit & LEAF_FLAG) && (s.objects[links[i].hit & LEAF_INDEX_MASK].flags & OBJ_FRONT_CHAIN)) L[i].hit &= ~LEAF_SIMPLE;
  }
  });
  auto* S = reinterpret_cast<DShape<R>*>(blob.data() + lay.shapes_off);
  auto* O = reinterpret_cast<DObject<R>*>(blob.data() + lay.objects_off);
  parallel_chunks((size_t)s.n_objects, [&](size_t, size_t lo_, size_t hi_) {
  for (size_t i = lo_; i < hi_; ++i) {
    const rt_object& o = s.objects[i];
    S[i].obj_id = o.obj_id; S[i].sub_id = o.sub_id;
    if (o.flags & RT_OBJ_SUB_BVH) {
      S[i].tag = -1; S[i].material = -1;
    } else {
      const rt_shape& sh = s.shapes[o.geom];
      S[i].tag = sh.tag; S[i].material = sh.material;
      for (int k = 0; k < 9; ++k) S[i].p[k] = (R)sh.p[k];
    }
    O[i].flags = o.flags;
    O[i].sub_root = sub_root_of_object[i];
    O[i].medium_material = o.medium_material;
    for (int k = 0; k < 3; ++k) O[i].offset[k] = (o.flags & RT_OBJ_TRANSLATE) ? (R)o.offset[k] : (R)0;  // subtracted unconditionally on the device
    O[i].sin_theta = (R)o.sin_theta; O[i].cos_theta = (R)o.cos_theta; O[i].neg_inv_density = (R)o.neg_inv_density;
  }
  });
  auto* M = reinterpret_cast<DMaterial<R>*>(blob.data() + lay.materials_off);
  for (int i = 0; i < s.n_materials; ++i) {
    M[i].tag = s.materials[i].tag; M[i].texture = s.materials[i].texture;
    for (int k = 0; k < 3; ++k) M[i].albedo[k] = (R)s.materials[i].albedo[k];
    M[i].fuzz = (R)s.materials[i].fuzz; M[i].ir = (R)s.materials[i].ir;
    // a solid colour is carried in the material itself (texture = ~index): the shading code then reads one record, not two
    const int tag = s.materials[i].tag, tx = s.materials[i].texture;
    if ((tag == RT_MAT_LAMBERTIAN || tag == RT_MAT_DIFFUSE_LIGHT || tag == RT_MAT_ISOTROPIC) && tx >= 0 && tx < s.n_textures &&
        s.textures[tx].tag == RT_TEX_SOLID) {
      for (int k = 0; k < 3; ++k) M[i].albedo[k] = (R)s.textures[tx].color[k];
      M[i].texture = ~tx;
    }
  }
  auto* T = reinterpret_cast<DTexture<R>*>(blob.data() + lay.textures_off);
  for (int i = 0; i < s.n_textures; ++i) {
    const rt_texture& t = s.textures[i];
    T[i].tag = t.tag; T[i].even = t.even; T[i].odd = t.odd; T[i].perlin = t.perlin; T[i].image = t.image;
    for (int k = 0; k < 3; ++k) T[i].color[k] = (R)t.color[k];
    T[i].scale = (R)t.scale;
  }
  auto* P = reinterpret_cast<DPerlin<R>*>(blob.data() + lay.perlins_off);
  for (int i = 0; i < s.n_perlins; ++i) {
    for (int j = 0; j < 256; ++j) {
      for (int k = 0; k < 3; ++k) P[i].ran[j][k] = (R)s.perlins[i].ran_vec[j][k];
      P[i].perm[0][j] = s.perlins[i].perm_x[j]; P[i].perm[1][j] = s.perlins[i].perm_y[j]; P[i].perm[2][j] = s.perlins[i].perm_z[j];
    }
  }
  {
    CameraRecord<R>& cr = *reinterpret_cast<CameraRecord<R>*>(blob.data() + lay.camera_off);
    const double* const parts[6] = {s.camera.origin, s.camera.lower_left_corner, s.camera.horizontal, s.camera.vertical, s.camera.u, s.camera.v};
    for (int v = 0; v < 6; ++v)
      for (int k = 0; k < 3; ++k) cr.f[3 * v + k] = (R)parts[v][k];
    cr.f[18] = (R)s.camera.lens_radius; cr.f[19] = (R)s.camera.shutter_open_time; cr.f[20] = (R)s.camera.shutter_close_time;
  }
  HIP_TRY(hipMalloc((void**)&im.d_blob, lay.total_bytes));
  HIP_TRY(hipMemcpy(im.d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
  if (wide_on_device) HIP_TRY(hipMemcpy(im.d_blob + lay.wide_off, device_wide, (size_t)lay.n_wide * sizeof(WideNode), hipMemcpyDeviceToDevice));
  im.device_built_tree = wide_on_device;
#ifdef RT_WIDE_Q8
  if (lay.n_wide && single) {  // the 64-byte nodes, packed, over the first half of the section
    WideNodeQ* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)lay.n_wide * sizeof(WideNodeQ)));
    hipLaunchKernelGGL(k_quantize_wide, dim3((lay.n_wide + 255u) / 256u), dim3(256), 0, c->stream, reinterpret_cast<const WideNode*>(im.d_blob + lay.wide_off), lay.n_wide, tmp);
    HIP_TRY(hipMemcpyAsync(im.d_blob + lay.wide_off, tmp, (size_t)lay.n_wide * sizeof(WideNodeQ), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(tmp));
  }
#endif
  if (preset == PRESET_ALL_LDS && lay.total_bytes > kLdsAllMax) preset = PRESET_ALL_TOP;  // the f64 image is larger than the f32 one
  if (preset == PRESET_BALLS_LDS && lay.total_bytes > kLdsAllMax) preset = PRESET_BALLS_TOP;
  if (preset == PRESET_FLATALL_LDS && lay.total_bytes > kLdsAllMax) preset = PRESET_FLATALL_TOP;
  im.preset = preset;
  im.table = table_for<R>(preset);
  DeviceScene<R>& d = im.dev;
  d.blob = im.d_blob;
  d.lay = lay;
  d.images = c->d_images;
  d.n_images = s.n_images;
  d.root_link = root_link;
  d.defer_ok = (reads_leftover_uv(s) || getenv("RTAMD_EAGER_RECORDS")) ? 0u : 1u;
  // The boxes as the caller gave them, beside their float form, where that form could change an answer: a scene with planar shapes
  // (two of them can coincide exactly, and which one the reference reports then hangs on `t_max <= t_min` in its box test,
  // src/aabb.rs:40) whose tree has a box coordinate that is not a float.  Every scene of the reference has float coordinates there.
  // The double-precision four-wide walk tests child boxes in single precision when no product of that test can overflow: the
  // root's child boxes contain every other box of the tree.
  d.narrow_ok = 0u;
  if (!single && lay.n_wide && !getenv("RTAMD_NO_NARROW_BOXES")) {
    WideNode root;
    HIP_TRY(hipMemcpy(&root, im.d_blob + lay.wide_off, sizeof(root), hipMemcpyDeviceToHost));
    bool ok = true;
    for (int ch = 0; ch < 4; ++ch)
      if (root.child[ch] != kWideNone)
        for (int k = 0; k < 3; ++k) ok = ok && std::fabs(root.lo[k][ch]) <= 0x1p40f && std::fabs(root.hi[k][ch]) <= 0x1p40f;
    d.narrow_ok = ok ? 1u : 0u;
  }
  d.exact_boxes = nullptr;
  if (!single && !links.empty() && !getenv("RTAMD_NO_EXACT_BOXES")) {
    bool planar = false;
    for (int i = 0; i < s.n_shapes && !planar; ++i) planar = s.shapes[i].tag >= RT_SHAPE_RECT_XY && s.shapes[i].tag <= RT_SHAPE_BOX;
    bool all_floats = true;
    if (planar)
      for (size_t i = 0; i < links.size() && all_floats; ++i)
        if (!(links[i].hit & LEAF_FLAG))
          for (int k = 0; k < 3; ++k) all_floats = all_floats && (double)(float)links[i].bmin[k] == links[i].bmin[k] && (double)(float)links[i].bmax[k] == links[i].bmax[k];
    if (planar && !all_floats) {
      std::vector<double> e(6 * links.size());
      for (size_t i = 0; i < links.size(); ++i) {
        const bool inverted = links[i].bmin[0] > links[i].bmax[0] || links[i].bmin[1] > links[i].bmax[1] || links[i].bmin[2] > links[i].bmax[2];
        for (int k = 0; k < 3; ++k) {  // (an inverted box: the point at +infinity, as in the links)
          e[6 * i + k] = inverted ? std::numeric_limits<double>::infinity() : links[i].bmin[k];
          e[6 * i + 3 + k] = inverted ? std::numeric_limits<double>::infinity() : links[i].bmax[k];
        }
      }
      HIP_TRY(hipMalloc((void**)&im.d_exact_boxes, e.size() * sizeof(double)));
      HIP_TRY(hipMemcpy(im.d_exact_boxes, e.data(), e.size() * sizeof(double), hipMemcpyHostToDevice));
      d.exact_boxes = im.d_exact_boxes;
    }
  }
  d.n_pre = 0;
  if (lay.n_wide && pre)
    for (uint32_t word : *pre) {
      if (d.n_pre >= kWidePre) return fail(RT_E_INVALID, "internal error: too many leaves kept outside the four-wide tree");
      // see the links above: a re-centred sphere's front_face rule is read from the object's flags
      const uint32_t of = s.objects[word & LEAF_INDEX_MASK].flags;
      const uint32_t w = (word & ~LEAF_SIMPLE) | ((of & 15u) == 0u ? LEAF_SIMPLE : 0u);
      d.pre[d.n_pre++] = (!single && (of & OBJ_FRONT_CHAIN)) ? (w & ~LEAF_SIMPLE) : w;
    }
  uint32_t top_bytes = kLdsTopMax;
  if (const char* e = getenv("RTAMD_TOP_LDS_BYTES")) top_bytes = (uint32_t)std::max(0, atoi(e));  // measurement switch
  d.lds_links = (preset == PRESET_ALL_TOP || preset == PRESET_BALLS_TOP || preset == PRESET_FLATALL_TOP) ? std::min<uint32_t>(lay.n_links, top_bytes / (uint32_t)sizeof(BoxLink)) : 0u;
  im.lds_bytes = im.table.lds_bytes(lay, d.lds_links);
  auto v3 = [](const double* p) { return V3<R>((R)p[0], (R)p[1], (R)p[2]); };
  d.camera.origin = v3(s.camera.origin); d.camera.lower_left_corner = v3(s.camera.lower_left_corner);
  d.camera.horizontal = v3(s.camera.horizontal); d.camera.vertical = v3(s.camera.vertical);
  d.camera.u = v3(s.camera.u); d.camera.v = v3(s.camera.v);
  d.camera.lens_radius = (R)s.camera.lens_radius; d.camera.t_open = (R)s.camera.shutter_open_time; d.camera.t_close = (R)s.camera.shutter_close_time;
  d.background.kind = s.background.kind; d.background.c0 = v3(s.background.c0); d.background.c1 = v3(s.background.c1);
  return RT_OK;
}


inline int ensure_accum(rt_context* c, size_t bytes) {
  if (c->accum_bytes >= bytes) return RT_OK;
  if (c->accum) (void)hipFree(c->accum);
  c->accum = nullptr; c->accum_bytes = 0;
  HIP_TRY(hipMalloc(&c->accum, bytes));
  c->accum_bytes = bytes;
  return RT_OK;
}
inline int ensure_ws(rt_context* c, size_t bytes) {
  if (c->ws_bytes >= bytes) return RT_OK;
  if (c->ws) (void)hipFree(c->ws);
  c->ws = nullptr; c->ws_bytes = 0;
  HIP_TRY(hipMalloc(&c->ws, bytes));
  c->ws_bytes = bytes;
  return RT_OK;
}

// `counters` / `host_rays`: the allocation-free entry point hands in its static scratch words and a page-locked host word; the
// ordinary entry points pass NULL and use the context's counter block and a stack variable.
template <class R> int render_accum_impl(rt_context* c, const rt_render_params& p, void* accum_dev, rt_stats* stats, unsigned long long* counters_override,
                                         unsigned long long* host_rays) {
  using R4 = typename Vec4<R>::type;
  SceneImage<R>& im = image_of<R>(c);
  const uint64_t frame_pixels = (uint64_t)p.width * p.height;
  // rows this call renders: all of them, or those of its own bands (the last band of the frame may be short)
  uint32_t own_rows = p.height, band_shift = 0;
  const bool banded = p.band_stride > 1u;
  if (banded) {
    while ((1u << band_shift) < p.band_rows) ++band_shift;
    own_rows = 0;
    for (uint32_t y0 = p.band_offset << band_shift; y0 < p.height; y0 += p.band_stride << band_shift) own_rows += std::min(p.band_rows, p.height - y0);
  }
  const uint64_t n_pixels = (uint64_t)p.width * own_rows;
  const uint32_t n_samples = p.sample_end - p.sample_begin;
  auto t_begin = std::chrono::steady_clock::now();
  R* accum = reinterpret_cast<R*>(accum_dev);
  if (!accum) {
    int rc = ensure_accum(c, frame_pixels * 4 * sizeof(R));
    if (rc) return rc;
    accum = reinterpret_cast<R*>(c->accum);
    HIP_TRY(hipMemsetAsync(accum, 0, frame_pixels * 4 * sizeof(R), c->stream));
  }
  if (p.pipeline == RT_PIPELINE_PERSISTENT || p.pipeline == RT_PIPELINE_AUTO) {
    // one launch for the whole sample range; no ray queues
    unsigned long long* const cnt = counters_override ? counters_override : c->d_ray_count;
    FrameParams fp{};
    fp.width = p.width; fp.height = p.height; fp.n_pixels = (uint32_t)n_pixels;
    fp.sample_base = p.sample_begin; fp.n_paths = 0; fp.max_depth = p.max_depth; fp.seed = p.seed;
    fp.n_seg = kSegments; fp.seg_cap = 0;
    // Chunks of up to 16 samples of one pixel per lane, summed in registers (FrameParams::chunk): for the small sphere scenes
    // under a bright sky without procedural textures - a globe or two, every path a bounce or two long and every one of them
    // bright - where one radiance atomic per path is two thirds of the frame time.  Everywhere else a unit is a single path.
    const bool bright_sky = im.dev.background.c0.x + im.dev.background.c0.y + im.dev.background.c0.z > (R)0;
    const bool cheap_paths = im.preset == PRESET_SPHERES && im.dev.lay.n_perlins == 0 && bright_sky;
    uint32_t chunk = cheap_paths ? std::min<uint32_t>(16u, std::max<uint32_t>(1u, n_samples)) : 1u;
    if (const char* e = getenv("RTAMD_CHUNK")) chunk = im.preset == PRESET_SPHERES ? (uint32_t)std::max(1, atoi(e)) : 1u;  // measurement switch
    fp.chunk = chunk;
    fp.n_samples = n_samples;
    fp.band_shift = band_shift; fp.band_stride = banded ? p.band_stride : 0u; fp.band_offset = banded ? p.band_offset : 0u;
    const unsigned long long total_paths = (unsigned long long)n_pixels * ((n_samples + chunk - 1) / chunk);  // units the path cursor counts
#ifdef RT_STAMPS
    if (counters_override) return fail(RT_E_UNSUPPORTED, "the diagnostic build has no allocation-free entry point");
    HIP_TRY(hipMemsetAsync(c->d_ray_count, 0, sizeof(unsigned long long) * kRayCountWords, c->stream));
#else
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * 16, c->stream));  // word 0 rays, word 8 path cursor
#endif
    const uint32_t bpc = resident_blocks(im.table.resident_blocks_per_cu(3, im.lds_bytes));
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->n_cus * bpc, (total_paths + kBlock - 1) / kBlock));
    {  // about 16 grabs per wave, but never under 256 paths: the cursor is one address, and at the path rate of the small
       // scenes grabs of 64 or 128 paths from 7000 waves pile up on it (measured: 1080p x 16 spp ran a third slower)
      const uint64_t waves = (uint64_t)grid * (kBlock / 64u);
      const char* e = getenv("RTAMD_GRABS_PER_WAVE");
      const uint64_t per = total_paths / (waves * (e ? (uint64_t)std::max(1, atoi(e)) : 16u));
      fp.path_grab = (uint32_t)std::min<uint64_t>(kPathGrab, std::max<uint64_t>(e ? 64u : 256u, per / 64u * 64u));
      // RTAMD_FIXED_GRAB=1: the grab of large frames (and with it the kernel instantiations that fold it as a constant) whatever
      // the frame size, so that those instantiations can be run on frames small enough to compare with a CPU evaluation
      if (getenv("RTAMD_FIXED_GRAB")) fp.path_grab = kPathGrab;
    }
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    uint32_t variant = 0;
    if (p.max_depth > 0 && total_paths > 0)
      variant = im.table.paths_persistent(grid, im.lds_bytes, c->stream, im.dev, fp, total_paths, cnt + 8, cnt, accum);
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipGetLastError());
    unsigned long long rays_stack[2] = {0, 0};
    unsigned long long* const rays_dst = host_rays ? host_rays : rays_stack;  // word 0 rays, word 1 rays traced a second time at a tie
    HIP_TRY(hipMemcpyAsync(rays_dst, cnt, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t retraced = rays_dst[1];  // rays on which two leaves tied, answered by the list walk in the reference's order
    const unsigned long long rays = rays_dst[0];
#ifdef RT_STAMPS
    {
      std::vector<unsigned long long> all(kRayCountWords);
      HIP_TRY(hipMemcpy(all.data(), c->d_ray_count, sizeof(unsigned long long) * kRayCountWords, hipMemcpyDeviceToHost));
      double st[8] = {0}, tot = 0;
      for (size_t w = 16; w + 8 <= kRayCountWords; w += 8)
        for (int k = 0; k < 8; ++k) st[k] += (double)all[w + k];
      for (int k = 0; k < 8; ++k) tot += st[k];
      if (im.preset >= PRESET_BALLS_LDS) {  // the presets that run k_paths_stepwise (the table of the persistent kernel is printed as well; ignore it there)
        const double cyc = st[0] + st[1] + st[2];
        fprintf(stderr, "[stamps] stepwise: wave cycles restart %.1f%% box steps %.1f%% leaf steps %.1f%%; per ray: %.1f box steps of %.1f lanes, %.2f leaf steps of %.1f lanes, "
                "restarts of %.1f idle lanes; %.0f cycles per 64 rays\n", 100 * st[0] / cyc, 100 * st[1] / cyc, 100 * st[2] / cyc,
                st[3] / (double)rays, st[3] / st[4], st[5] / (double)rays, st[5] / st[6], 0.0, cyc / ((double)rays / 64.0));
      }
      fprintf(stderr, "[stamps] persistent, wave cycles by phase (0 refill, 1 primary rays + trace rest, 2 scatter, 3 walk setup, 4 leaf loop, 5 finalize, 6 close ended paths, 7 Philox):");
      for (int k = 0; k < 8; ++k) fprintf(stderr, " [%d] %.1f%%", k, 100.0 * st[k] / tot);
      fprintf(stderr, "  %.0f cycles per wave iteration of %.1f rays\n", tot / ((double)rays / 64.0) , 64.0);
    }
#endif
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
    if (stats) {
      stats->rays = rays;
      stats->camera_rays = n_pixels * n_samples;
      stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      stats->bounce_kernel_ms = ms;
      stats->bounce_kernel_launches = 1;
      stats->gen_kernel_ms = 0.0;
      stats->passes = 1;
      stats->lds_bytes = im.lds_bytes;
      stats->scene_in_lds = (uint32_t)im.preset;
      stats->kernel_variant = variant | (im.device_built_tree ? RT_VARIANT_DEVICE_TREE : 0u) | (im.dev.narrow_ok ? RT_VARIANT_NARROW_BOXES : 0u);
      stats->rays_retraced = retraced;
    }
    return RT_OK;
  }
  if (counters_override) return fail(RT_E_UNSUPPORTED, "the allocation-free entry point renders with the persistent pipeline only");
  if (banded) return fail(RT_E_UNSUPPORTED, "row bands are rendered by the persistent pipeline only");
  uint32_t spp_pass = p.samples_per_pass ? p.samples_per_pass : (uint32_t)std::max<uint64_t>(1, kTargetPathsPerPass / n_pixels);
  spp_pass = std::max(1u, std::min(spp_pass, std::max(1u, n_samples)));
  if (n_pixels * spp_pass > 0x7FFFFFFFull) return fail(RT_E_INVALID, "width*height*samples_per_pass exceeds 2^31-1");
  // Segment capacity: a whole number of 64-entry chunks, and an ODD number of them, so that the byte distance between the segments
  // (and, with the pad below, between the arrays) is an odd multiple of 1 KiB: the 64 segment streams that advance in step then
  // spread over all memory channels instead of landing on the few that a power-of-two distance selects.
  uint32_t seg_chunks = (uint32_t)(((n_pixels * spp_pass + kSegments - 1) / kSegments + 63) / 64);
  seg_chunks |= 1u;
  const uint32_t seg_cap = seg_chunks * 64u;
  const uint64_t cap = (uint64_t)seg_cap * kSegments;
  const bool sorted = p.pipeline == RT_PIPELINE_SORTED;
  // workspace: two ray queues (3 arrays each) [+ hit queue (3 arrays) + class index lists]
  const size_t kArrayPad = 5u * 1024u;
  size_t q_bytes = (size_t)cap * sizeof(R4) + kArrayPad;
  size_t org_bytes = (((size_t)cap * 4 + 255) & ~(size_t)255) + kArrayPad;
  size_t ws_bytes = q_bytes * 6 + org_bytes * 2 + (sorted ? q_bytes * 3 + (size_t)cap * 4 * MAT_CLASSES : 0);
  int rc = ensure_ws(c, ws_bytes);
  if (rc) return rc;
  uint8_t* w = reinterpret_cast<uint8_t*>(c->ws);
  PathQueue<R> q[2];
  for (int k = 0; k < 2; ++k) {
    q[k].a = reinterpret_cast<R4*>(w); w += q_bytes;
    q[k].b = reinterpret_cast<R4*>(w); w += q_bytes;
    q[k].c = reinterpret_cast<R4*>(w); w += q_bytes;
  }
  for (int k = 0; k < 2; ++k) {
    q[k].org = reinterpret_cast<uint32_t*>(w); w += org_bytes;
  }
  HitQueue<R> hq{};
  uint32_t* class_index = nullptr;
  if (sorted) {
    hq.a = reinterpret_cast<R4*>(w); w += q_bytes;
    hq.b = reinterpret_cast<R4*>(w); w += q_bytes;
    hq.c = reinterpret_cast<R4*>(w); w += q_bytes;
    class_index = reinterpret_cast<uint32_t*>(w);
  }
  // counter block: queue counts, cursors of the first and of the second kernel of a bounce (each (max_depth + 2) * kSegments words,
  // zeroed together), then the class counters at a fixed offset
  const size_t per_array = (size_t)kSegments * (p.max_depth + 2);
  uint32_t* counters = c->d_counters;
  uint32_t* cursors_a = counters + per_array;
  uint32_t* cursors_b = cursors_a + per_array;
  uint32_t* class_counters = c->d_counters + (size_t)(kMaxDepth + 2) * kSegments * 3;
  #ifdef RT_STAMPS
  HIP_TRY(hipMemsetAsync(c->d_ray_count, 0, sizeof(unsigned long long) * kRayCountWords, c->stream));  // ray counter + diagnostic words
#else
  HIP_TRY(hipMemsetAsync(c->d_ray_count, 0, sizeof(unsigned long long) * 16, c->stream));
#endif

  double bounce_ms = 0.0, gen_ms = 0.0;
  uint64_t launches = 0, passes = 0;
  const size_t class_bytes = sizeof(uint32_t) * (size_t)kSegments * (size_t)(p.max_depth + 1) * 8;
  for (uint32_t s0 = p.sample_begin; s0 < p.sample_end; s0 += spp_pass) {
    uint32_t ns = std::min(spp_pass, p.sample_end - s0);
    FrameParams fp{};
    fp.width = p.width; fp.height = p.height; fp.n_pixels = (uint32_t)n_pixels;
    fp.sample_base = s0; fp.n_paths = (uint32_t)(n_pixels * ns); fp.max_depth = p.max_depth; fp.seed = p.seed;
    fp.n_seg = kSegments; fp.seg_cap = seg_cap;
    HIP_TRY(hipMemsetAsync(counters, 0, sizeof(uint32_t) * per_array * 3, c->stream));
    if (sorted) HIP_TRY(hipMemsetAsync(class_counters, 0, class_bytes, c->stream));
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    launch_gen_primary<R>(grid_for(c, fp.n_paths, 8), c->stream, im.dev, fp, q[0], counters);
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    const uint32_t grid = grid_for(c, fp.n_paths, resident_blocks(im.table.resident_blocks_per_cu(sorted ? 1 : 0, im.lds_bytes)));
    const uint32_t grid_shade = sorted ? grid_for(c, fp.n_paths, resident_blocks(im.table.resident_blocks_per_cu(2, im.lds_bytes))) : grid;
    for (uint32_t b = 0; b < p.max_depth; ++b) {
      PathQueue<R>& in = q[b & 1u];
      PathQueue<R>& out = q[(b + 1u) & 1u];
      if (!sorted) {
        im.table.bounce_fused(grid, im.lds_bytes, c->stream, im.dev, fp, in, out, counters, cursors_a, c->d_ray_count, accum, b);
        launches += 1;
      } else {
        uint32_t* cc = class_counters + (size_t)b * 8 * kSegments;
        im.table.trace(grid, im.lds_bytes, c->stream, im.dev, fp, in, hq, counters, cursors_a, c->d_ray_count, cc, class_index, (uint32_t)cap, accum, b);
        im.table.shade(grid_shade, im.lds_bytes, c->stream, im.dev, fp, in, hq, out, counters, cursors_b, cc, class_index, (uint32_t)cap, accum, b);
        launches += 2;
      }
    }
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    gen_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2]));
    bounce_ms += ms;
    ++passes;
  }
  unsigned long long rays = 0;
  HIP_TRY(hipMemcpyAsync(&rays, c->d_ray_count, sizeof(rays), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
#ifdef RT_STAMPS
  {
    std::vector<unsigned long long> all(kRayCountWords);
    HIP_TRY(hipMemcpy(all.data(), c->d_ray_count, sizeof(unsigned long long) * kRayCountWords, hipMemcpyDeviceToHost));
    unsigned long long st[9] = {0};
    for (size_t w = 16; w + 8 <= kRayCountWords; w += 8)
      for (int k = 0; k < 8; ++k) st[1 + k] += all[w + k];
    double tot = 0;
    for (int k = 1; k < 9; ++k) tot += (double)st[k];
    fprintf(stderr, "[stamps] wave cycles by phase (persistent kernel: 0 refill, 1 trace, 2 primary rays, 3 close ended paths, 4 Philox, 5 scatter, 6 loop):");
    for (int k = 1; k < 8; ++k) fprintf(stderr, " [%d] %.1f%%", k - 1, 100.0 * (double)st[k] / tot);
    fprintf(stderr, "  total %.3e cycles, %.0f per wave pass of 64 rays\n", tot, tot / ((double)rays / 64.0));
  }
#endif
  if (stats) {
    stats->rays = rays;
    stats->camera_rays = n_pixels * n_samples;
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    stats->bounce_kernel_ms = bounce_ms;
    stats->bounce_kernel_launches = launches;
    stats->gen_kernel_ms = gen_ms;
    stats->passes = passes;
    stats->lds_bytes = im.lds_bytes;
    stats->scene_in_lds = (uint32_t)im.preset;
  }
  return RT_OK;
}

template <class R> int trace_batch_impl(rt_context* c, const rt_ray* rays, size_t n, rt_hit* hits, uint64_t seed, int mode) {
  SceneImage<R>& im = image_of<R>(c);
  RayIO* d_rays = nullptr;
  HitIO* d_hits = nullptr;
  HIP_TRY(hipMalloc((void**)&d_rays, n * sizeof(RayIO)));
  hipError_t e = hipMalloc((void**)&d_hits, n * sizeof(HitIO));
  if (e != hipSuccess) { (void)hipFree(d_rays); return fail(RT_E_HIP, "hipMalloc hits failed"); }
  int rc = RT_OK;
  do {
    if (hipMemcpyAsync(d_rays, rays, n * sizeof(RayIO), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(RT_E_HIP, "H2D rays failed"); break; }
    uint32_t grid = grid_for(c, n, 8);
    im.table.first_hit(grid, im.lds_bytes, c->stream, im.dev, d_rays, (uint32_t)n, d_hits, seed, mode);
    if (hipGetLastError() != hipSuccess) { rc = fail(RT_E_HIP, "k_first_hit launch failed"); break; }
    if (hipMemcpyAsync(hits, d_hits, n * sizeof(HitIO), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(RT_E_HIP, "D2H hits failed"); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(RT_E_HIP, "stream sync failed after k_first_hit"); break; }
  } while (0);
  (void)hipFree(d_rays);
  (void)hipFree(d_hits);
  return rc;
}

template <class R>
int pairs_kernel_impl(rt_context* c, int which, uint32_t W, uint32_t H, uint32_t max_depth, const uint32_t* pixel, const uint32_t* sample, size_t n,
                      uint64_t seed, void* out, size_t out_bytes) {
  SceneImage<R>& im = image_of<R>(c);
  uint32_t *d_px = nullptr, *d_sm = nullptr;
  void* d_out = nullptr;
  int rc = RT_OK;
  do {
    if (hipMalloc((void**)&d_px, n * 4) != hipSuccess || hipMalloc((void**)&d_sm, n * 4) != hipSuccess || hipMalloc(&d_out, out_bytes) != hipSuccess) {
      rc = fail(RT_E_HIP, "hipMalloc failed");
      break;
    }
    if (hipMemcpyAsync(d_px, pixel, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_sm, sample, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      rc = fail(RT_E_HIP, "H2D failed");
      break;
    }
    if (which == 0) {
      launch_camera_rays<R>((uint32_t)((n + kBlock - 1) / kBlock), c->stream, im.dev, W, H, d_px, d_sm, (uint32_t)n, seed, reinterpret_cast<RayIO*>(d_out));
    } else {
      uint32_t grid = grid_for(c, n, 8);
      // which == 2: `out` receives path vertices (max_vertices * 12 doubles per pair) after the 4 radiance doubles of every pair
      uint32_t max_vertices = which == 2 ? (uint32_t)((out_bytes / sizeof(double) / n - 4) / 12) : 0u;
      double* d_path = which == 2 ? reinterpret_cast<double*>(d_out) + n * 4 : nullptr;
      if (which == 2 && hipMemsetAsync(d_out, 0, out_bytes, c->stream) != hipSuccess) { rc = fail(RT_E_HIP, "memset failed"); break; }
      im.table.sample_radiance(grid, im.lds_bytes, c->stream, im.dev, W, H, max_depth, d_px, d_sm, (uint32_t)n, seed, reinterpret_cast<double*>(d_out),
                               d_path, max_vertices);
    }
    if (hipGetLastError() != hipSuccess) { rc = fail(RT_E_HIP, "kernel launch failed"); break; }
    if (hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(RT_E_HIP, "D2H failed"); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(RT_E_HIP, "stream sync failed"); break; }
  } while (0);
  if (d_px) (void)hipFree(d_px);
  if (d_sm) (void)hipFree(d_sm);
  if (d_out) (void)hipFree(d_out);
  return rc;
}


template <class R> int resolve_impl(rt_context* c, const void* accum_dev, size_t n_pixels, uint32_t spp, uint8_t* d_rgb) {
  uint32_t grid = (uint32_t)((n_pixels + kBlock - 1) / kBlock);
  launch_resolve<R>(grid, c->stream, reinterpret_cast<const R*>(accum_dev), (uint32_t)n_pixels, spp, d_rgb);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

#define RTI_INSTANTIATE(R)                                                                                                                   \
  template void layout_sections<R>(BlobLayout&);                                                                                        \
  template int build_image<R>(rt_context*, const rt_scene_desc&, const std::vector<TmpLink>&, int, const std::vector<int>&, int, SceneImage<R>&, \
                              const std::vector<WideTmp>*, const std::vector<uint32_t>*, bool, const WideNode*, uint32_t);                     \
  template int render_accum_impl<R>(rt_context*, const rt_render_params&, void*, rt_stats*, unsigned long long*, unsigned long long*);                                              \
  template int trace_batch_impl<R>(rt_context*, const rt_ray*, size_t, rt_hit*, uint64_t, int);                                                   \
  template int pairs_kernel_impl<R>(rt_context*, int, uint32_t, uint32_t, uint32_t, const uint32_t*, const uint32_t*, size_t, uint64_t, void*, \
                                    size_t);                                                                                               \
  template int resolve_impl<R>(rt_context*, const void*, size_t, uint32_t, uint8_t*);


RTI_INSTANTIATE(float)
RTI_INSTANTIATE(double)

}  // namespace rti
