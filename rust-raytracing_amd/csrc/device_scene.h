// Device image of a flattened scene, templated on the arithmetic type.
//
// The host uploads one packed byte blob per precision:
//     [links][shapes][objects][materials][textures][perlins]        (every section 16-byte aligned)
// and the kernels read it through a SceneView whose pointers are either into global memory or
// into the workgroup's LDS copy.  `links` is the BVH of src/bvh.rs rewritten as a threaded list
// (hit link / miss link per entry): the reference always descends left then right with a shrinking
// t_max (src/bvh.rs:108-119), so a fixed visiting order with skip pointers reproduces it without a
// per-lane stack.  Links are sorted by tree depth, so the first K entries are the top levels of the
// tree and can be staged in LDS on their own when the whole scene does not fit.
#pragma once
#include "device_math.h"

namespace rtd {

constexpr int LINK_END = -1;
constexpr uint32_t LEAF_FLAG = 0x80000000u;    // hit word of a leaf entry
constexpr uint32_t LEAF_SIMPLE = 0x40000000u;  // object has no transform / medium / nested tree: test the shape directly
constexpr uint32_t LEAF_INDEX_MASK = 0x3FFFFFFFu;

template <class R> struct alignas(16) DLink {  // 32 B (float) / 64 B (double)
  R bmin[3];
  uint32_t hit;  // internal: next entry when the box is hit; leaf: LEAF_FLAG | flags | object index
  R bmax[3];
  int32_t miss;  // next entry when the box is missed / after a leaf was tested; LINK_END terminates
};

// tags equal the RT_SHAPE_* / RT_MAT_* / RT_TEX_* / RT_OBJ_* values of rtamd.h
enum { SH_SPHERE = 0, SH_MOVING = 1, SH_RECT_XY = 2, SH_RECT_XZ = 3, SH_RECT_YZ = 4, SH_BOX = 5 };
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_LIGHT = 3, MAT_ISOTROPIC = 4, MAT_CLASSES = 5 };
enum { TEX_SOLID = 0, TEX_CHECKER = 1, TEX_NOISE = 2, TEX_IMAGE = 3 };
enum : uint32_t { OBJ_TRANSLATE = 1u, OBJ_ROTATE_Y = 2u, OBJ_MEDIUM = 4u, OBJ_SUB_BVH = 8u };

template <class R> struct alignas(16) DShape {  // one per object, indexed by object
  int32_t tag, material, obj_id, sub_id;
  R p[10];
};
template <class R> struct alignas(16) DObject {  // indexed by object; read only for non-simple leaves
  uint32_t flags;
  int32_t sub_root;  // first link of the nested tree (OBJ_SUB_BVH)
  int32_t medium_material;
  int32_t pad;
  R offset[3];
  R sin_theta, cos_theta, neg_inv_density;
  R pad2[2];
};
template <class R> struct alignas(16) DMaterial {
  int32_t tag, texture;
  R albedo[3];
  R fuzz, ir;
  R pad;
};
template <class R> struct alignas(16) DTexture {
  int32_t tag, even, odd, perlin, image, pad;
  R color[3];
  R scale;
  R pad2[2];
};
template <class R> struct alignas(16) DPerlin {
  R ran[256][4];
  int32_t perm[3][256];
};
struct DImage {
  const uint8_t* rgb;
  int32_t width, height;
};

template <class R> struct DCamera {  // src/camera.rs:6-17
  V3<R> origin, lower_left_corner, horizontal, vertical, u, v;
  R lens_radius, t_open, t_close;
};
template <class R> struct DBackground {  // src/color.rs:35-39
  int32_t kind;
  V3<R> c0, c1;
};

// Section byte offsets inside the blob + counts.  Passed to kernels by value.
struct BlobLayout {
  uint32_t links_off, shapes_off, objects_off, materials_off, textures_off, perlins_off, total_bytes;
  uint32_t n_links, n_objects, n_materials, n_textures, n_perlins;
};

template <class R> struct DeviceScene {
  const uint8_t* blob;  // global memory
  BlobLayout lay;
  const DImage* images;  // global memory
  int32_t n_images;
  int32_t root_link;
  uint32_t lds_links;  // number of leading links staged in LDS when the blob is not staged whole
  DCamera<R> camera;
  DBackground<R> background;
};

// What the hit / shade routines read.  LDS_ALL: every pointer addresses LDS.  Otherwise pointers address
// global memory and `top` is the LDS copy of the first `n_top` links.
template <class R, bool LDS_ALL> struct SceneView {
  const DLink<R>* links;
  const DLink<R>* top;
  uint32_t n_top;
  const DShape<R>* shapes;
  const DObject<R>* objects;
  const DMaterial<R>* materials;
  const DTexture<R>* textures;
  const DPerlin<R>* perlins;
  const DImage* images;

  RT_DEV DLink<R> link(int i) const {
    if constexpr (LDS_ALL) {
      return links[i];
    } else {
      if ((uint32_t)i < n_top) return top[i];
      return links[i];
    }
  }
};

// Cooperative copy of `bytes` (multiple of 16) from global to LDS by the whole workgroup.
RT_DEV void stage_to_lds(void* lds, const void* g, uint32_t bytes) {
  const uint4* src = reinterpret_cast<const uint4*>(g);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (uint32_t i = threadIdx.x; i < bytes / 16u; i += blockDim.x) dst[i] = src[i];
}

template <class R, bool LDS_ALL> RT_DEV SceneView<R, LDS_ALL> make_view(const DeviceScene<R>& sc, uint8_t* lds) {
  SceneView<R, LDS_ALL> v;
  const uint8_t* base;
  if constexpr (LDS_ALL) {
    stage_to_lds(lds, sc.blob, sc.lay.total_bytes);
    base = lds;
    v.top = nullptr;
    v.n_top = 0;
  } else {
    stage_to_lds(lds, sc.blob + sc.lay.links_off, sc.lds_links * (uint32_t)sizeof(DLink<R>));
    base = sc.blob;
    v.top = reinterpret_cast<const DLink<R>*>(lds);
    v.n_top = sc.lds_links;
  }
  __syncthreads();
  v.links = reinterpret_cast<const DLink<R>*>(base + sc.lay.links_off);
  v.shapes = reinterpret_cast<const DShape<R>*>(base + sc.lay.shapes_off);
  v.objects = reinterpret_cast<const DObject<R>*>(base + sc.lay.objects_off);
  v.materials = reinterpret_cast<const DMaterial<R>*>(base + sc.lay.materials_off);
  v.textures = reinterpret_cast<const DTexture<R>*>(base + sc.lay.textures_off);
  v.perlins = reinterpret_cast<const DPerlin<R>*>(base + sc.lay.perlins_off);
  v.images = sc.images;
  return v;
}

}  // namespace rtd
