/*
 * rtamd.h — C ABI of the MI355X path-tracing library (librtamd.so).
 *
 * The reference (soikkea/rust-raytracing) has no FFI or plugin interface; the
 * narrowest seam shared by its CLI and GUI front ends is the blocking render
 * entry point
 *     fn render(scene: SceneConfig) -> Result<image::RgbImage, RecvError>      src/render.rs:179-207
 * fed by `SceneConfig` (src/scenes.rs:33-41) and producing W*H `(x, y, Rgb<u8>)`
 * pixels (src/render.rs:39,192-194).  This header is what a Rust `build.rs`
 * would bind to replace that function: the host flattens its `dyn Hittable`
 * tree into the plain arrays below, uploads them once, and calls rt_render().
 *
 * Conventions: every function returns 0 on success and a non-zero RT_E_* code
 * on failure; the message is available from rt_last_error() (thread local).
 * All pointers are plain host pointers unless a parameter is documented as a
 * device pointer.  No function falls back to a CPU path: without a usable HIP
 * device rt_create() fails.
 * A context serves one host thread at a time and drives one GPU (one process
 * or thread per GPU; rt_comm_init() / rt_reduce_accum() sum the GPUs'
 * accumulation buffers onto one of them).
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID 1      /* bad argument / malformed scene */
#define RT_E_NO_DEVICE 2    /* no HIP device, or device index out of range */
#define RT_E_HIP 3          /* a HIP runtime call failed */
#define RT_E_NO_SCENE 4     /* render/trace before rt_upload_scene */
#define RT_E_UNSUPPORTED 5  /* scene uses a composition the device code does not implement */

/* Arithmetic type of the device kernels.  The reference computes in f64.
 * RT_PRECISION_F64 reports what the reference's own walk reports.  The closest hit does not depend on the tree that is walked
 * unless two different surfaces lie at exactly the same distance, so on scenes too large for the small-scene kernels the library
 * walks a tree of its own with the reference's leaf arithmetic, notes the rays on which two leaves tie, and traces the paths of
 * such a ray a second time by the list walk of the caller's tree, in the reference's order (rt_stats.rays_retraced). */
#define RT_PRECISION_F32 0
#define RT_PRECISION_F64 1
/* RT_PRECISION_F64 | RT_F64_OWN_TREE: the same walk with cheaper leaf forms (a box as three slabs instead of six rectangles,
 * spheres of a dissolved nested tree re-centred) and without the second pass.  Images agree to rounding; what is given up is the
 * reference's rule for exactly coincident surfaces (the last one its walk tests wins).  Scenes that take a small-scene preset
 * have no such tree: the flag is ignored. */
#define RT_F64_OWN_TREE 0x400

/* Pipeline.  FUSED = per-bounce ray queues, one trace+shade kernel per bounce;
 * SORTED = per-bounce ray queues, trace kernel writing hit records binned by
 * material class + shade kernel; PERSISTENT = one launch per frame, every lane
 * follows whole paths and lanes whose path ended take the next unstarted
 * (pixel, sample) in place.  All three produce the same per-sample radiance. */
#define RT_PIPELINE_AUTO 0
#define RT_PIPELINE_FUSED 1
#define RT_PIPELINE_SORTED 2
#define RT_PIPELINE_PERSISTENT 3

/* ---- scene description (flattened `SceneConfig`, src/scenes.rs:33-41) ---- */

/* src/camera.rs:6-17, fields in declaration order (w is carried but unused by get_ray). */
typedef struct rt_camera {
  double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
  double lens_radius, shutter_open_time, shutter_close_time;
} rt_camera;

/* src/color.rs:35-39 */
#define RT_BG_SOLID 0    /* c0 */
#define RT_BG_GRADIENT 1 /* Gradient(start = c0, end = c1) */
typedef struct rt_background {
  int32_t kind, pad;
  double c0[3], c1[3];
} rt_background;

/* src/texture.rs */
#define RT_TEX_SOLID 0   /* color */
#define RT_TEX_CHECKER 1 /* even, odd = texture indices (must be non-checker) */
#define RT_TEX_NOISE 2   /* perlin = index into perlins[], scale */
#define RT_TEX_IMAGE 3   /* image = index into images[], or -1 for "file failed to load" (cyan) */
typedef struct rt_texture {
  int32_t tag, even, odd, perlin, image, pad;
  double color[3];
  double scale;
} rt_texture;

/* src/perlin.rs:6-11 */
typedef struct rt_perlin {
  double ran_vec[256][3];
  int32_t perm_x[256], perm_y[256], perm_z[256];
} rt_perlin;

/* src/texture.rs:87-92; rgb = height*width*3 bytes, row 0 on top */
typedef struct rt_image {
  int32_t width, height;
  const uint8_t* rgb;
} rt_image;

/* src/material.rs */
#define RT_MAT_LAMBERTIAN 0    /* texture */
#define RT_MAT_METAL 1         /* albedo, fuzz (already clamped to <= 1) */
#define RT_MAT_DIELECTRIC 2    /* ir */
#define RT_MAT_DIFFUSE_LIGHT 3 /* texture */
#define RT_MAT_ISOTROPIC 4     /* texture */
typedef struct rt_material {
  int32_t tag, texture;
  double albedo[3];
  double fuzz, ir;
} rt_material;

/* Leaf geometry.  p[]:
 *   SPHERE         cx cy cz r                          src/sphere.rs:9-13
 *   MOVING_SPHERE  c0x c0y c0z c1x c1y c1z r t0 t1     src/moving_sphere.rs:9-16
 *   RECT_XY        x0 x1 y0 y1 k                       src/aarect.rs:11-18
 *   RECT_XZ        x0 x1 z0 z1 k                       src/aarect.rs:19-26
 *   RECT_YZ        y0 y1 z0 z1 k                       src/aarect.rs:27-34
 *   BOX            minx miny minz maxx maxy maxz       src/box_struct.rs:12-16 (six rects, fixed order :22-71)
 */
#define RT_SHAPE_SPHERE 0
#define RT_SHAPE_MOVING_SPHERE 1
#define RT_SHAPE_RECT_XY 2
#define RT_SHAPE_RECT_XZ 3
#define RT_SHAPE_RECT_YZ 4
#define RT_SHAPE_BOX 5
typedef struct rt_shape {
  int32_t tag, material;
  double p[9];
} rt_shape;

/* One entry of a hittable list: [ConstantMedium(] [Translate(] [RotateY(] geometry [)))].
 * Exactly this nesting order is supported (it is the only one the reference's
 * scenes use: src/scenes.rs:364-365,406-427,548-553); geometry is a shape or a
 * nested BVH whose leaves are objects without a nested BVH of their own. */
#define RT_OBJ_TRANSLATE 1u /* src/hittable.rs:112-152, offset */
#define RT_OBJ_ROTATE_Y 2u  /* src/hittable.rs:154-243, sin_theta / cos_theta */
#define RT_OBJ_MEDIUM 4u    /* src/constant_medium.rs:12-34, neg_inv_density + phase material */
#define RT_OBJ_SUB_BVH 8u   /* geom is a node index of nodes[] instead of a shape index */
typedef struct rt_object {
  uint32_t flags;
  int32_t geom;
  int32_t medium_material;
  int32_t obj_id, sub_id; /* identity reported by rt_trace_batch: index in the world list / in the nested list (-1) */
  int32_t pad;
  double offset[3];
  double sin_theta, cos_theta;
  double neg_inv_density;
} rt_object;

/* src/bvh.rs:12-16.  A child >= 0 is a node index; a child < 0 is object ~child.
 * Traversal order is always left then right (src/bvh.rs:108-119). */
typedef struct rt_bvh_node {
  double bmin[3], bmax[3];
  int32_t left, right;
} rt_bvh_node;

typedef struct rt_scene_desc {
  rt_camera camera;
  rt_background background;
  int32_t root_node; /* BVHNode::new(&world.objects, 0, 1), src/render.rs:211 */
  int32_t n_nodes, n_objects, n_shapes, n_materials, n_textures, n_perlins, n_images;
  const rt_bvh_node* nodes;
  const rt_object* objects;
  const rt_shape* shapes;
  const rt_material* materials;
  const rt_texture* textures;
  const rt_perlin* perlins;
  const rt_image* images;
} rt_scene_desc;

/* ---- rendering ---- */

typedef struct rt_render_params {
  uint32_t width, height;            /* SceneConfig::image_size(), src/scenes.rs:135-138 */
  uint32_t sample_begin, sample_end; /* global sample indices this call renders (sample-range sharding) */
  uint32_t max_depth;                /* src/scenes.rs:59 */
  int32_t precision;                 /* RT_PRECISION_* */
  int32_t pipeline;                  /* RT_PIPELINE_* */
  uint32_t samples_per_pass;         /* queue pipelines: samples of every pixel traced together; 0 = choose */
  uint64_t seed;                     /* Philox counter words 2,3 */
  /* Row bands (image-tile sharding, the static row split of src/render.rs:218-224,247-271 made interleaved): the frame's rows are
   * cut into bands of `band_rows` rows (a power of two), band b belongs to the caller with b % band_stride == band_offset, and this
   * call renders the rows of its own bands only (the other rows of `accum_dev` are not touched).  band_stride 0 or 1 = every row.
   * Pixels keep their whole-frame index in the random streams, so the image does not depend on how the rows were shared out.
   * PERSISTENT / AUTO pipeline only. */
  uint32_t band_rows, band_stride, band_offset;
  uint32_t pad;
} rt_render_params;

typedef struct rt_stats {
  uint64_t rays;        /* closest-hit queries = ray_color calls with depth > 0 (src/render.rs:158) */
  uint64_t camera_rays; /* width*height*(sample_end-sample_begin) */
  double seconds;       /* host wall clock of the call, device idle on return */
  double bounce_kernel_ms;     /* HIP-event time of the path kernel (PERSISTENT) or summed over the bounce kernels */
  uint64_t bounce_kernel_launches;
  double gen_kernel_ms;        /* HIP-event time of the primary-ray kernels */
  uint64_t passes;
  uint32_t lds_bytes;          /* LDS used per workgroup by the bounce kernel */
  uint32_t scene_in_lds;       /* kernel preset the scene selected: 0 boxes, 1 smoke, 2 spheres, 3 all (scene in LDS), 4 all (top links in LDS), 5/6 spheres-only lane walk (LDS / top links), 7/8 all without nested trees (images whose nested trees were dissolved or spliced), 9/10 spheres-only / all without nested trees on the four-wide form of the library's own tree, 11/12 the same with the reference's leaf arithmetic and tie rule (the double-precision default) */
  uint32_t kernel_variant;     /* PERSISTENT pipeline: which instantiation of the path kernel ran, RT_VARIANT_* bits */
  uint32_t pad;
  uint64_t rays_retraced;      /* double precision on the library's own tree: closest-hit queries on which two different surfaces lay at
                                  exactly the same distance, answered a second time by the list walk in the reference's order
                                  (src/bvh.rs:108-119) */
} rt_stats;
#define RT_VARIANT_FIXED_GRAB 1u       /* paths handed out in constant grabs of 2048 per wave (large frames) */
#define RT_VARIANT_CHUNKED 2u          /* a lane sums up to 16 samples of one pixel before its one atomic add (cheap bright paths) */
#define RT_VARIANT_STEPWISE 4u         /* k_paths_stepwise (per-lane tree walks) instead of k_paths_persistent */
#define RT_VARIANT_DEFERRED_RECORDS 8u /* double precision on the caller's tree: hit record built once, after the walk */
#define RT_VARIANT_DEVICE_TREE 16u     /* the tree the kernel walked was built on the GPU at upload (scenes of 100 000 objects and more) */
#define RT_VARIANT_NARROW_BOXES 32u    /* double precision on the library's tree: child boxes tested in single precision, widened so that no box is missed */

typedef struct rt_context rt_context;

/* Create a context on HIP device `device`.  Replaces ThreadPool::new in render() (src/render.rs:184). */
int rt_create(rt_context** out, int device);
void rt_destroy(rt_context* ctx);

/* Copy the flattened scene to the device (both f32 and f64 images).  The caller may free `scene` afterwards.
 * Replaces the Arc-sharing of world/camera in render_with_threadpool (src/render.rs:211-216). */
int rt_upload_scene(rt_context* ctx, const rt_scene_desc* scene);

/* The sample loop (src/render.rs:224-237) for samples [sample_begin, sample_end) of every pixel:
 * adds linear radiance sums into `accum_dev`, a DEVICE buffer of width*height*4 floats (F32) or
 * doubles (F64), pixel-major, row 0 on top, channel 3 unused.  NULL = the context's own buffer
 * (zeroed at the start of the call).  Returns with the stream idle.  The buffer must be ordinary device memory (hipMalloc):
 * the sums are added with hardware floating-point atomics, which host-mapped and managed memory drop; such a buffer is refused
 * with RT_E_INVALID. */
int rt_render_accum(rt_context* ctx, const rt_render_params* params, void* accum_dev, rt_stats* stats);

/* color_to_pixel (src/color.rs:14-33) over an accumulation buffer (NULL = the context's own):
 * writes width*height*3 bytes, row 0 on top, to host memory. */
int rt_resolve(rt_context* ctx, const void* accum_dev, uint32_t width, uint32_t height, uint32_t samples_per_pixel,
               int precision, uint8_t* rgb8_out);

/* Copy an accumulation buffer (NULL = the context's own) to host as width*height*3 doubles. */
int rt_read_accum(rt_context* ctx, const void* accum_dev, uint32_t width, uint32_t height, int precision, double* out);

/* render() of src/render.rs:179-207: rt_render_accum into the context's buffer + rt_resolve.
 * `accum_out` (optional) receives the width*height*3 linear sums as doubles. */
int rt_render(rt_context* ctx, const rt_render_params* params, uint8_t* rgb8_out, double* accum_out, rt_stats* stats);

/* ---- progressive rendering: the polling side of ThreadedRenderer (src/render.rs:69-108) ----
 * The GUI of the reference starts a render, then drains finished pixels once per UI frame (check_progress, src/render.rs:84-108)
 * into an image it redraws.  Here the image refines as a whole: rt_progress_begin() fixes the frame and clears the context's
 * own accumulation buffer; every rt_progress_step() renders the next `max_samples` samples of every pixel (blocking for that
 * slice only - milliseconds) and writes the image of all samples so far to rgb8_out, so a caller polling once per UI frame shows
 * a picture that sharpens until *finished is set.  Sample streams are keyed by the global sample index: the last image is the
 * one rt_render() returns for the same parameters. */
int rt_progress_begin(rt_context* ctx, const rt_render_params* params);
int rt_progress_step(rt_context* ctx, uint32_t max_samples, uint8_t* rgb8_out /* width*height*3, row 0 on top */,
                     uint32_t* samples_done /* of sample_end - sample_begin, may be NULL */, int* finished /* may be NULL */,
                     rt_stats* stats /* of this step, may be NULL */);

/* ---- multi-GPU: one context per GPU (one process or thread each), sums reduced onto one rank over RCCL / xGMI ----
 * The reference partitions image rows over worker threads (src/render.rs:218-224) and gathers pixels through an mpsc channel
 * (src/render.rs:185,192-201).  Here every rank renders its slice of the SAMPLE range of the whole frame with rt_render_accum
 * (streams are keyed by the global sample index, so the image does not depend on the number of ranks beyond summation order),
 * then rt_reduce_accum sums the buffers onto `root`, which calls rt_resolve.  RCCL is opened at run time; without it these
 * calls return RT_E_UNSUPPORTED and everything else keeps working. */
typedef struct rt_comm_id { char bytes[128]; } rt_comm_id; /* = ncclUniqueId */
/* Rank 0 creates the id; the host hands its 128 bytes to the other ranks by whatever channel it has (file, pipe, MPI, ...). */
int rt_comm_unique_id(rt_comm_id* id);
/* Collective over the n_ranks contexts that share `id`: joins this context's GPU to the communicator (ncclCommInitRank). */
int rt_comm_init(rt_context* ctx, const rt_comm_id* id, int n_ranks, int rank);
int rt_comm_destroy(rt_context* ctx);
/* Collective: element-wise sum of the ranks' accumulation buffers (DEVICE, width*height*4 floats or doubles as rt_render_accum
 * fills them; NULL = the context's own) onto rank `root`.  One ncclReduce(sum) of width*height*3 values - the unused fourth
 * value of a pixel's slot does not travel - on the context's stream; returns with the stream idle.  The other ranks' buffers
 * are left unchanged. */
int rt_reduce_accum(rt_context* ctx, void* accum_dev, uint32_t width, uint32_t height, int precision, int root);

/* rt_render() for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc, hosts that must not
 * fragment device memory.  No counterpart in the reference (its render() allocates freely); it exists for those callers only.
 * The call allocates and frees nothing, on the device or through any allocating routine: the counters and the RGB8 staging
 * image live in a static device array inside the library (about 24 MiB on every GPU the library runs on, held until the process
 * ends), everything else is the caller's.  Allocation-free is all it promises: the call still synchronises the context's stream,
 * reads two events and copies the ray count and the image to the host, none of which may run inside a HIP graph capture.
 *   accum_dev  DEVICE buffer of width*height*4 floats (F32) or doubles (F64), required; zeroed by the call, receives the sums
 *   rgb8_out   host buffer of width*height*3 bytes, required (page-locked memory makes the copy one DMA; a pageable buffer goes
 *              through the runtime's own staging)
 * Limits: width*height <= 3840*2160 (RT_E_INVALID beyond), pipeline AUTO or PERSISTENT (RT_E_UNSUPPORTED otherwise).
 * Every call on a GPU uses that GPU's one scratch array: calls that overlap on the same GPU (two contexts, two threads) race. */
int rt_render_noalloc(rt_context* ctx, const rt_render_params* params, void* accum_dev, uint8_t* rgb8_out, rt_stats* stats);

/* ---- first-hit batches: world.hit(ray, tmin, tmax, rec) through the BVH (src/bvh.rs:108-119) ---- */
typedef struct rt_ray {
  double origin[3], direction[3], time, tmin, tmax;
  uint32_t pixel, sample, bounce, pad; /* key of the random stream ConstantMedium::hit draws from */
} rt_ray;
typedef struct rt_hit { /* HitRecord, src/hittable.rs:10-18 */
  int32_t hit, obj, sub, side, material, front_face;
  double t, p[3], normal[3], u, v;
} rt_hit;
/* `precision` may carry RT_TRACE_AS_RENDERED: the records are then built by exactly the code the render kernels run for this
 * scene, which fills (u, v) only when one of the scene's textures reads them (they are reported as 0 otherwise). */
#define RT_TRACE_AS_RENDERED 0x100
/* RT_TRACE_VISITS (implies AS_RENDERED): u and v of every record report instead the number of box entries and the number of
 * leaves the walk tested for the ray - the work a closest-hit query costs on the tree the kernels of `precision` walk. */
#define RT_TRACE_VISITS 0x200
int rt_trace_batch(rt_context* ctx, const rt_ray* rays, size_t n, rt_hit* hits, int precision, uint64_t seed);

/* Primary rays exactly as the sample loop builds them (src/render.rs:230-234 + src/camera.rs:62-72). */
int rt_camera_rays(rt_context* ctx, uint32_t width, uint32_t height, const uint32_t* pixel, const uint32_t* sample, size_t n,
                   uint64_t seed, int precision, rt_ray* out);

/* Per-sample radiance ray_color(ray, ..., max_depth) (src/render.rs:149-177) for chosen (pixel, sample) pairs:
 * `out` receives 4 doubles each = r, g, b, number of closest-hit queries the path made. */
int rt_sample_radiance(rt_context* ctx, uint32_t width, uint32_t height, uint32_t max_depth, const uint32_t* pixel,
                       const uint32_t* sample, size_t n, uint64_t seed, int precision, double* out);

/* The same paths with their vertices: `out` receives n*4 doubles as rt_sample_radiance writes them, followed by
 * n*max_vertices records of 12 doubles (ray origin[3], direction[3], t, obj, sub, material, front_face, hit) for the
 * first max_vertices closest-hit queries of every path; unused records are zero. */
int rt_sample_path(rt_context* ctx, uint32_t width, uint32_t height, uint32_t max_depth, const uint32_t* pixel,
                   const uint32_t* sample, size_t n, uint64_t seed, int precision, uint32_t max_vertices, double* out);

const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
