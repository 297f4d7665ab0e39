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
constexpr uint32_t LEAF_NESTED = 0x20000000u;  // object is a tree of its own (OBJ_SUB_BVH)
// Second visit of the object of a one-object node: BVHNode::new gives such a node the same child twice (src/bvh.rs:38) and
// BVHNode::hit calls it twice (src/bvh.rs:114-116).  A deterministic shape returns the same record again and the visit is dropped
// at upload; a ConstantMedium draws a fresh random number on every call (src/constant_medium.rs:91), so its second visit stays
// in the list and reads a draw slot of its own.
constexpr uint32_t LEAF_SECOND = 0x10000000u;
constexpr uint32_t LEAF_INDEX_MASK = 0x0FFFFFFFu;
// Draw slot of a medium's free-path sample on the MEDIA stream: the object's index, plus bit 28 on the second visit of a
// one-object node of the main tree and bit 29 on the second visit of one inside a nested tree.
RT_DEV uint32_t medium_slot(uint32_t obj_id, uint32_t visit) { return obj_id | (visit << 28); }
RT_DEV uint32_t visit_bit(uint32_t leaf_word, int level) { return (leaf_word & LEAF_SECOND) ? (1u << level) : 0u; }

template <class R> struct alignas(16) DLink {  // 32 B (float) / 64 B (double)
  R bmin[3];
  uint32_t hit;  // internal: next entry when the box is hit; leaf: LEAF_FLAG | flags | object index
  R bmax[3];
  int32_t miss;  // next entry when the box is missed / after a leaf was tested; LINK_END terminates
};
// The tree is stored with single-precision boxes in BOTH precisions: the double-precision image rounds its boxes outward to
// floats, the double-precision walk converts them back (exactly) and tests them in double.  A box only decides which leaves get
// tested; rounding it outward can add leaves to that set, never remove one (subtraction and multiplication round monotonically,
// so the slab distances of the larger box bracket those of the exact one), and a leaf the reference's own boxes would have kept a
// ray away from cannot be hit by that ray.  Half the bytes per tree entry for the walks that live on memory.
using BoxLink = DLink<float>;

// A node of the four-wide tree (WALK_LANE_WIDE): the boxes of up to four children and what each child is - the index of another
// node, or the leaf word of an object (LEAF_FLAG set).  One node is one 128-byte cache line, and a step of the walk is one
// fetch for four box tests where the threaded list makes four dependent ones.  Single-precision boxes in both images, like
// BoxLink's.  An unused slot carries the point at +infinity as its box (no ray passes the test) and kWideNone as its child.
constexpr uint32_t kWideNone = 0xFFFFFFFFu;  // also "no walk in progress" in the walk's state: not a leaf word any tree contains
struct alignas(128) WideNode {
  float lo[3][4];  // lo[axis][child]
  float hi[3][4];
  uint32_t child[4];
  uint32_t pad[4];
};
#ifdef RT_WIDE_Q8
// Measurement build: the single-precision images keep a node in 64 bytes - the corner and cell size of a 256-step grid per axis
// and the children's boxes as 8-bit grid coordinates (lower bounds rounded down, upper bounds up), four to a word.
struct alignas(64) WideNodeQ {
  float org[3], cell[3];
  uint32_t qlo[3], qhi[3];  // byte c of word a: child c, axis a
  uint32_t child[4];
};
static_assert(sizeof(WideNodeQ) == 64, "quantised wide node size");
#endif
// Entries of a lane's stack of the wide walk, one 32-bit child each.  The host refuses the walk (and falls back to the threaded
// list) for a tree whose worst case needs more: three pushes per level of the path from the root.  36 entries are 144 bytes a
// lane, which lets four waves per SIMD share a compute unit's 160 KiB of LDS - the walk lives on memory latency, and its speed
// follows the number of resident waves (measured on the million-sphere field: 5.7 Grays/s at two waves per SIMD, 7.3 at three,
// 8.1 at four).
#ifndef RT_WIDE_STACK
#define RT_WIDE_STACK 36
#endif
constexpr uint32_t kWideStack = RT_WIDE_STACK;
constexpr uint32_t kWidePre = 4;  // leaves a four-wide image may keep outside its tree (DeviceScene::pre)
constexpr uint32_t kWideStackBytesPerWave = kWideStack * 64u * 4u;

// tags equal the RT_SHAPE_* / RT_MAT_* / RT_TEX_* / RT_OBJ_* values of rtamd.h
enum { SH_SPHERE = 0, SH_MOVING = 1, SH_RECT_XY = 2, SH_RECT_XZ = 3, SH_RECT_YZ = 4, SH_BOX = 5 };
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_LIGHT = 3, MAT_ISOTROPIC = 4, MAT_CLASSES = 5 };
enum { TEX_SOLID = 0, TEX_CHECKER = 1, TEX_NOISE = 2, TEX_IMAGE = 3 };
enum : uint32_t { OBJ_TRANSLATE = 1u, OBJ_ROTATE_Y = 2u, OBJ_MEDIUM = 4u, OBJ_SUB_BVH = 8u,
                  // library-internal (never in rt_object as the caller passes it): a sphere that stood under Translate(RotateY(..)) and
                  // was re-centred at upload.  It is tested like a bare sphere; only its front_face flag goes through the chain of
                  // set_face_normal calls the wrappers would have made (sin_theta / cos_theta are the holder's).
                  OBJ_FRONT_CHAIN = 16u };

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
// The camera as a record of the scene blob (origin, lower_left_corner, horizontal, vertical, u, v, lens_radius, t_open, t_close
// in that order): the path kernels fetch it where a primary ray is made instead of holding 21 kernel arguments in scalar
// registers for the whole launch.
template <class R> struct alignas(16) CameraRecord {
  R f[sizeof(R) == 8 ? 22 : 24];
};
template <class R> struct DBackground {  // src/color.rs:35-39
  int32_t kind;
  V3<R> c0, c1;
};

// Section byte offsets inside the blob + counts.  Passed to kernels by value.
struct BlobLayout {
  uint32_t links_off, shapes_off, objects_off, materials_off, textures_off, perlins_off, flat_off, total_bytes;
  uint32_t n_links, n_objects, n_materials, n_textures, n_perlins;
  // Small scenes (flat_mode != 0): the tree's leaves as lists.  Single precision: plain rectangles, grouped by their constant
  // axis (n_frect[0] YZ, [1] XZ, [2] XY), as FlatRect records at frect_off; every other leaf as a FlatLeaf record, in visiting
  // order, at flat_off (n_flat of them).  Double precision: every leaf as a FlatLeaf record, in visiting order (no FlatRect records).
  uint32_t n_flat, flat_mode, frect_off, n_frect[3];
  uint32_t faces_off;  // FlatFace records, six per object (flat_mode only)
  uint32_t camera_off; // one CameraRecord
  uint32_t wide_off, n_wide;  // WideNode records (128-byte aligned), root first; n_wide = 0: the image has no four-wide tree
};

template <class R> struct DeviceScene {
  const uint8_t* blob;  // global memory
  BlobLayout lay;
  const DImage* images;  // global memory
  int32_t n_images;
  int32_t root_link;
  uint32_t lds_links;  // number of leading links staged in LDS when the blob is not staged whole
  uint32_t n_pre;      // four-wide walk: leaves tested ahead of the walk (objects as large as the scene), leaf words in `pre`
  uint32_t pre[4];
  uint32_t defer_ok;   // no texture of the scene reads the (u, v) a moving sphere or a medium inherits from an earlier candidate
  // Double precision, scenes with planar shapes whose tree boxes are not all floats: the boxes of the links as the caller gave them
  // (six doubles per link: min xyz, max xyz; global memory), else null.  Whether the second of two exactly coincident surfaces is
  // still visited hangs on an exact comparison in the reference's box test (src/aabb.rs:40); a box the stored float form lets a
  // ray into is tested once more in this form.
  const double* exact_boxes;
  uint32_t narrow_ok;  // four-wide tree: every box coordinate within +-2^40, so the double-precision walk may test the boxes in single precision (wide_keys)
  DCamera<R> camera;
  DBackground<R> background;
};

// ---------------------------------------------------------------------------------------------
struct alignas(16) FlatRect {  // a plain rectangle ready for the test: plane coordinate, in-plane bounds, object index
  float k, a0, a1, b0, b1;
  int32_t obj, pad0, pad1;
};

// One leaf of a small scene as the flat walk reads it: the leaf word, the shape and the holder's wrappers in one record, stored
// in visiting order, so that a step of the walk is one scalar fetch whose address depends on no earlier fetch (the leaf word, then
// the shape it names, then the object record would be three fetches in a row).
template <class R> struct alignas(16) FlatLeaf {  // 80 B (float) / 144 B (double)
  uint32_t word;
  int32_t tag, obj_id;
  uint32_t flags;
  R p[10];
  R offset[3], sin_theta, cos_theta, neg_inv_density;
};

// One planar side of a rectangle or box, as the record of a hit needs it (small scenes, flat walk): slot 6 * object + side,
// a rectangle's only side in slot 0.  `n` is the side's outward normal after the holder's RotateY; bits 0..3 of `flags` are the
// holder's OBJ_* flags, bits 4..5 the side's constant axis in the shape's own frame.
template <class R> struct alignas(16) FlatFace {  // 32 B (float) / 64 B (double)
  R n[3];
  R cos_theta, sin_theta, k;
  uint32_t flags;
  uint32_t pad[sizeof(R) == 4 ? 1 : 3];
};

// Kernel configuration: which parts of the scene grammar a kernel instantiation carries, and how it walks the tree.
// A scene is rendered by the smallest preset that covers its contents; code for absent features is not compiled
// in, which is what keeps the register count (and with it the waves per SIMD) of the common cases low.
// ---------------------------------------------------------------------------------------------
enum : uint32_t {
  F_SPHERE = 1u << 0, F_MOVING = 1u << 1, F_RECT = 1u << 2, F_BOX = 1u << 3,
  F_XFORM = 1u << 4,    // Translate / RotateY wrappers
  F_MEDIUM = 1u << 5,   // ConstantMedium wrapper + Isotropic phase function
  F_NESTED = 1u << 6,   // objects that are a BVH of their own
  F_UV = 1u << 7,       // some texture reads (u, v): records carry them
  F_CHECKER = 1u << 8, F_NOISE = 1u << 9, F_IMAGE = 1u << 10,
  F_METAL = 1u << 11, F_DIELECTRIC = 1u << 12,
  F_GRADIENT = 1u << 13,  // gradient background
  F_ALL = (1u << 14) - 1u,
  // Not a part of the scene grammar: the four-wide walk standing in for the reference's own walk in double precision.  Leaf tests
  // are the reference's expressions (a box is its six rectangles, with true quotients), and a ray on which two different leaves
  // report exactly the same distance - the one case in which the order of the tests decides what is reported - is flagged and
  // traced once more by the list walk in the reference's order.
  F_EXACT = 1u << 14,
};
// WALK_UNIFORM: the scene has few leaves (<= kFlatMaxLeaves).  Every wave steps through the list of leaves in the tree's visiting
//   order with a wave-uniform index, so shape and object data arrive through scalar loads and live in scalar registers; every
//   lane tests every leaf against its own shrinking t_max and no box is tested (see traverse_flat).
// WALK_LANE: every lane follows its own hit/miss links; the first entries (top tree levels) or the whole scene sit in LDS.
// WALK_LANE_WIDE: every lane walks the four-wide tree with a stack of its own in LDS, nearest child first; the scene stays in
//   global memory.  Only trees of the library's own take it (visiting order depends on the ray, so exact ties may resolve
//   differently from the reference's fixed order).
enum { WALK_UNIFORM = 0, WALK_LANE_LDS_ALL = 1, WALK_LANE_LDS_TOP = 2, WALK_LANE_WIDE = 3 };
constexpr uint32_t kFlatMaxLeaves = 32;

template <uint32_t FEATURES, int WALK> struct Cfg {
  static constexpr uint32_t F = FEATURES;
  static constexpr int walk = WALK;
  static constexpr bool has(uint32_t f) { return (FEATURES & f) != 0u; }
  // Small-scene presets: the double-precision kernels may replace an IEEE divide or square root by a form that differs from it
  // by an ulp or two (reciprocal multiply, refined hardware estimates).  Their scenes have no clusters of small specular spheres
  // that would amplify the difference, so paths still follow the reference arithmetic's to ~1e-13; the per-lane presets, whose
  // scenes do have such clusters, keep the IEEE forms everywhere.
#ifdef RT_STRICT_F64  // measurement build: true quotients and square roots everywhere (with -ffp-contract=off: the reference's plain IEEE arithmetic)
  static constexpr bool relaxed() { return false; }
#else
  static constexpr bool relaxed() { return WALK == 0; }
#endif
  // Per-lane walks that step through the tree (k_paths_stepwise, traverse_wide): keep the winning candidate as (t, leaf, side) and
  // build the record once, after the walk, where a record per accepted candidate is costly - boxes (six rectangle records),
  // wrappers, media (two boundary records per test); final-scene +17 % in single precision.  Sphere-only presets fill the record
  // as they go: there the deferred form measures 3-4 % slower.  Double precision on the caller's tree always fills the record as
  // it goes, as the reference does (a hit that writes no (u, v) must find the ones an earlier candidate left); on the library's
  // own tree (the four-wide walk) the order of the candidates is not the reference's anyway, and it defers like single precision.
  // The path kernel has a deferring variant for the caller's tree in double precision as well, launched when nothing in the scene
  // reads such a leftover (DeviceScene::defer_ok): every other field of the record is a function of the ray, the winning leaf and
  // its t alone.
  static constexpr bool costly_records() { return has(F_BOX) || has(F_MEDIUM) || has(F_XFORM); }
  template <class R> static constexpr bool defer_records() { return costly_records() && (sizeof(R) == sizeof(float) || WALK == 3); }
};
constexpr uint32_t P_BOXES = F_RECT | F_BOX | F_XFORM;                                  // cornell-box
constexpr uint32_t P_SMOKE = P_BOXES | F_MEDIUM;                                        // cornell-smoke
constexpr uint32_t P_SPHERES = F_SPHERE | F_RECT | F_UV | F_CHECKER | F_NOISE | F_IMAGE;  // two-spheres, perlin, earth, simple-light
using CfgBoxes = Cfg<P_BOXES, WALK_UNIFORM>;
using CfgSmoke = Cfg<P_SMOKE, WALK_UNIFORM>;
using CfgSpheres = Cfg<P_SPHERES, WALK_UNIFORM>;
constexpr uint32_t P_BALLS = F_SPHERE | F_MOVING | F_CHECKER | F_METAL | F_DIELECTRIC | F_GRADIENT;  // random, mega-spheres
using CfgAllLds = Cfg<F_ALL, WALK_LANE_LDS_ALL>;
using CfgAllTop = Cfg<F_ALL, WALK_LANE_LDS_TOP>;
// every feature except nested trees: single-precision images whose nested trees were dissolved into the main tree at upload
using CfgFlatAllLds = Cfg<F_ALL & ~F_NESTED, WALK_LANE_LDS_ALL>;
using CfgFlatAllTop = Cfg<F_ALL & ~F_NESTED, WALK_LANE_LDS_TOP>;
using CfgBallsLds = Cfg<P_BALLS, WALK_LANE_LDS_ALL>;
using CfgBallsTop = Cfg<P_BALLS, WALK_LANE_LDS_TOP>;
using CfgBallsWide = Cfg<P_BALLS, WALK_LANE_WIDE>;
using CfgFlatAllWide = Cfg<F_ALL & ~F_NESTED, WALK_LANE_WIDE>;
using CfgBallsWideExact = Cfg<P_BALLS | F_EXACT, WALK_LANE_WIDE>;
using CfgFlatAllWideExact = Cfg<(F_ALL & ~F_NESTED) | F_EXACT, WALK_LANE_WIDE>;
enum {
  PRESET_BOXES = 0, PRESET_SMOKE = 1, PRESET_SPHERES = 2, PRESET_ALL_LDS = 3, PRESET_ALL_TOP = 4, PRESET_BALLS_LDS = 5, PRESET_BALLS_TOP = 6,
  PRESET_FLATALL_LDS = 7, PRESET_FLATALL_TOP = 8,  // images whose nested trees were dissolved (the library's own tree)
  PRESET_BALLS_WIDE = 9, PRESET_FLATALL_WIDE = 10,  // the library's own tree in its four-wide form
  PRESET_BALLS_WIDE_EXACT = 11, PRESET_FLATALL_WIDE_EXACT = 12,  // ... with the reference's leaf arithmetic and tie rule (double precision)
  PRESET_COUNT = 13
};

// Read-only record at a wave-uniform address in global memory that no kernel writes (the scene blob): loaded through
// the constant address space, i.e. by scalar loads into scalar registers.
template <class T> RT_DEV T load_uniform(const T* p) {
  static_assert(sizeof(T) % 16 == 0, "record size must be a multiple of 16 bytes");
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) u32x4* const_ptr;
  const_ptr q = (const_ptr)(uintptr_t)p;
  union {
    T t;
    u32x4 v[sizeof(T) / 16];
  } u;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 16; ++k) u.v[k] = q[k];
  return u.t;
}

// Record in the workgroup's LDS, read with LDS instructions whatever the optimiser knows about the pointer.
template <class T> RT_DEV T load_lds(const T* p) {
  static_assert(sizeof(T) % 16 == 0, "record size must be a multiple of 16 bytes");
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) u32x4* lds_ptr;
  lds_ptr q = (lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)p;  // the low 32 bits of a generic LDS address are the LDS offset
  union {
    T t;
    u32x4 v[sizeof(T) / 16];
  } u;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 16; ++k) u.v[k] = q[k];
  return u.t;
}

// What the hit / shade routines read.  `links`, `shapes`, `objects` are read with wave-uniform indices in WALK_UNIFORM
// (global memory, scalar loads) and per lane otherwise; materials, textures and Perlin tables are always gathered per lane
// and sit in LDS unless only the top links fit there.
template <class R, class C> struct SceneView {
  const BoxLink* links;
  const BoxLink* top;  // WALK_LANE_LDS_TOP: LDS copy of the first n_top links
  uint32_t n_top;
  uint32_t n_links;
  const DShape<R>* shapes;
  const DObject<R>* objects;
  const FlatLeaf<R>* flat;         // leaves in visiting order (global memory, scalar loads), n_flat of them
  uint32_t n_flat;
  const FlatRect* frect;           // plain rectangles by axis (global memory, scalar loads)
  uint32_t n_frect[3];
  const FlatFace<R>* faces_lane;   // per-lane gathers (LDS)
  const DShape<R>* shapes_lane;    // the same records where a lane gathers them with an index of its own (LDS in WALK_UNIFORM)
  const DObject<R>* objects_lane;
  const DMaterial<R>* materials;
  const DTexture<R>* textures;
  const DPerlin<R>* perlins;
  const DImage* images;
  const CameraRecord<R>* camera_rec;  // global memory, scalar loads
  const WideNode* wide;               // WALK_LANE_WIDE: the four-wide tree (global memory)
  uint32_t* stack;                    // ... and the stacks of the lane's wave in LDS: child words, [entry][lane]
  uint32_t n_pre, pre[4];             // ... and the leaves tested ahead of the walk
  const double* exact_boxes;          // see DeviceScene::exact_boxes (null in nearly every scene)
  bool narrow_ok;                     // see DeviceScene::narrow_ok

  RT_DEV DCamera<R> camera() const { return camera_of(load_uniform(camera_rec)); }
  RT_DEV static DCamera<R> camera_of(const CameraRecord<R>& r) {
    DCamera<R> c;
    c.origin = V3<R>(r.f[0], r.f[1], r.f[2]); c.lower_left_corner = V3<R>(r.f[3], r.f[4], r.f[5]);
    c.horizontal = V3<R>(r.f[6], r.f[7], r.f[8]); c.vertical = V3<R>(r.f[9], r.f[10], r.f[11]);
    c.u = V3<R>(r.f[12], r.f[13], r.f[14]); c.v = V3<R>(r.f[15], r.f[16], r.f[17]);
    c.lens_radius = r.f[18]; c.t_open = r.f[19]; c.t_close = r.f[20];
    return c;
  }
  RT_DEV BoxLink link(int i) const {
    if constexpr (C::walk == WALK_LANE_LDS_TOP) {
      // the LDS copy is read through an LDS-typed pointer: a pointer that may be either LDS or global would turn every link
      // fetch of the walk into a flat load
      if ((uint32_t)i < n_top) return load_lds(top + i);
    }
    return links[i];
  }
  // Scene records addressed by an index that is the same in every lane (WALK_UNIFORM): fetched through the scalar
  // data path, once per wave, whatever the lanes' execution mask is.
  RT_DEV DShape<R> shape(int i) const {
    if constexpr (C::walk == WALK_UNIFORM) return load_uniform(shapes + i);
    else return shapes[i];
  }
  RT_DEV DObject<R> object(int i) const {
    if constexpr (C::walk == WALK_UNIFORM) return load_uniform(objects + i);
    else return objects[i];
  }
};

// Cooperative copy of `bytes` (multiple of 16) from global to LDS by the whole workgroup.
RT_DEV void stage_to_lds(void* lds, const void* g, uint32_t bytes) {
  const uint4* src = reinterpret_cast<const uint4*>(g);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (uint32_t i = threadIdx.x; i < bytes / 16u; i += blockDim.x) dst[i] = src[i];
}

// LDS bytes a workgroup needs for a configuration.
template <class R, class C> inline uint32_t lds_bytes_for(const BlobLayout& lay, uint32_t lds_links) {
  if (C::walk == WALK_UNIFORM) return lay.total_bytes;
  if (C::walk == WALK_LANE_LDS_ALL) return lay.total_bytes;
  if (C::walk == WALK_LANE_WIDE) return 4u * kWideStackBytesPerWave;  // every kernel of these presets runs 256 threads a workgroup
  return lds_links * (uint32_t)sizeof(BoxLink);
}

template <class R, class C> RT_DEV SceneView<R, C> make_view(const DeviceScene<R>& sc, uint8_t* lds) {
  SceneView<R, C> v;
  const uint8_t* g = sc.blob;
  const uint8_t* geom = g;   // base for links / shapes / objects
  const uint8_t* shade = g;  // base for materials / textures / perlins ...
  uint32_t shade_sub = 0;    // ... whose blob offsets are reduced by this much
  v.top = nullptr;
  v.n_top = 0;
  const uint8_t* lane = g;   // base for per-lane gathers of shapes / objects
  if constexpr (C::walk == WALK_UNIFORM) {
    stage_to_lds(lds, g, sc.lay.total_bytes);  // small by construction; links/shapes/objects are ALSO read from global by scalar loads
    shade = lds;
    lane = lds;
  } else if constexpr (C::walk == WALK_LANE_LDS_ALL) {
    stage_to_lds(lds, g, sc.lay.total_bytes);
    geom = lds;
    shade = lds;
    lane = lds;
  } else if constexpr (C::walk == WALK_LANE_WIDE) {
    // nothing staged: the LDS holds the stacks
  } else {
    stage_to_lds(lds, g + sc.lay.links_off, sc.lds_links * (uint32_t)sizeof(BoxLink));
    v.top = reinterpret_cast<const BoxLink*>(lds);
    v.n_top = sc.lds_links;
  }
  __syncthreads();
  v.wide = reinterpret_cast<const WideNode*>(g + sc.lay.wide_off);
  v.n_pre = sc.n_pre;
  for (int k = 0; k < 4; ++k) v.pre[k] = sc.pre[k];
  v.stack = reinterpret_cast<uint32_t*>(lds + (threadIdx.x >> 6) * kWideStackBytesPerWave);
  v.exact_boxes = sc.exact_boxes;
  v.narrow_ok = sc.narrow_ok != 0u;
  v.n_links = sc.lay.n_links;
  v.links = reinterpret_cast<const BoxLink*>(geom + sc.lay.links_off);
  v.shapes = reinterpret_cast<const DShape<R>*>(geom + sc.lay.shapes_off);
  v.objects = reinterpret_cast<const DObject<R>*>(geom + sc.lay.objects_off);
  v.shapes_lane = reinterpret_cast<const DShape<R>*>(lane + sc.lay.shapes_off);
  v.objects_lane = reinterpret_cast<const DObject<R>*>(lane + sc.lay.objects_off);
  v.faces_lane = reinterpret_cast<const FlatFace<R>*>(lane + sc.lay.faces_off);
  v.materials = reinterpret_cast<const DMaterial<R>*>(shade + (sc.lay.materials_off - shade_sub));
  v.textures = reinterpret_cast<const DTexture<R>*>(shade + (sc.lay.textures_off - shade_sub));
  v.perlins = reinterpret_cast<const DPerlin<R>*>(shade + (sc.lay.perlins_off - shade_sub));
  v.images = sc.images;
  v.flat = reinterpret_cast<const FlatLeaf<R>*>(g + sc.lay.flat_off);
  v.n_flat = sc.lay.n_flat;
  v.camera_rec = reinterpret_cast<const CameraRecord<R>*>(g + sc.lay.camera_off);
  v.frect = reinterpret_cast<const FlatRect*>(g + sc.lay.frect_off);
  for (int k = 0; k < 3; ++k) v.n_frect[k] = sc.lay.n_frect[k];
  return v;
}

}  // namespace rtd
