/*
 * rtamd_host.h — host-side mirror of the reference's scene / camera / render
 * driver layer, written in C++ because the reference's own toolchain (Rust) is
 * absent from this environment.  It produces the flattened rt_scene_desc that
 * librtamd.so consumes and drives the GPU render exactly where the reference
 * calls render().
 *
 *   Scene enum + SceneConfig::get_scene      src/scenes.rs:21-133
 *   Camera::new                              src/camera.rs:21-60
 *   BVHNode::new (tree shape, tie order)     src/bvh.rs:19-76
 *   RenderConfig / render_and_save           src/render.rs:27-37,278-284
 *   Cli {no_gui, output, scene}              src/cli.rs:7-21
 */
#ifndef RTAMD_HOST_H
#define RTAMD_HOST_H

#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* clap::ValueEnum order of src/scenes.rs:21-31; kebab-case names as the CLI spells them. */
#define RTH_SCENE_RANDOM 0
#define RTH_SCENE_TWO_SPHERES 1
#define RTH_SCENE_TWO_PERLIN_SPHERES 2
#define RTH_SCENE_EARTH 3
#define RTH_SCENE_SIMPLE_LIGHT 4
#define RTH_SCENE_CORNELL_BOX 5
#define RTH_SCENE_CORNELL_SMOKE 6
#define RTH_SCENE_FINAL_SCENE 7
#define RTH_SCENE_MEGA_SPHERES 8 /* not in the reference: random_scene on a (2*grid)^2 lattice */

typedef struct rth_scene_opts {
  int32_t scene;
  int32_t width;      /* 0 = the scene's constant (src/scenes.rs:56,91,100,109) */
  double aspect;      /* 0 = the scene's constant */
  int32_t spp;        /* 0 = the scene's constant */
  int32_t depth;      /* 0 = 50 (src/scenes.rs:59) */
  uint64_t aux_seed;  /* seeds every value the reference takes from the unseeded thread_rng() while building a scene */
  uint64_t bvh_seed;  /* seeds the split-axis choice of BVHNode::new (src/bvh.rs:26-27) */
  const uint8_t* earth_rgb; /* decoded earthmap.jpg (RGB8, row 0 on top) or NULL */
  int32_t earth_w, earth_h;
  int32_t mega_grid;  /* RTH_SCENE_MEGA_SPHERES only: a, b in [-grid, grid) */
  int32_t mega_view;  /* RTH_SCENE_MEGA_SPHERES only: 0 = camera at (13, 6, 3) looking across the field (a sliver of it fills the frame);
                         1 = camera 2.8 * grid above the centre looking straight down, the whole field in the frame */
  const char* earth_path; /* used when earth_rgb is NULL: binary PPM (P6) to load; NULL = "earthmap.ppm" in the CWD,
                             a missing file gives the reference's cyan fallback (src/texture.rs:106-110,131-133) */
} rth_scene_opts;

typedef struct rth_scene rth_scene;

rth_scene* rth_scene_create(const rth_scene_opts* opts);
void rth_scene_destroy(rth_scene* s);
const rt_scene_desc* rth_scene_desc(const rth_scene* s);
/* image_size() (src/scenes.rs:135-138), samples_per_pixel, max_depth */
void rth_scene_config(const rth_scene* s, uint32_t* width, uint32_t* height, uint32_t* spp, uint32_t* depth);
/* -1 if `name` is not one of the kebab-case scene names */
int rth_scene_from_name(const char* name);
const char* rth_scene_name(int scene);

/* image.save(path): encoder chosen from the extension (src/render.rs:273-276).  .ppm (P6), .png and .jpg / .jpeg (baseline JFIF, quality 75) are implemented. */
int rth_save_image(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);
/* binary PPM reader used for the pre-decoded earth map */
int rth_load_ppm(const char* path, uint8_t** rgb8_out, int32_t* width, int32_t* height);
void rth_free(void* p);

/* render_and_save (src/render.rs:278-284) on GPU `device`: builds the scene, renders, prints the reference's
 * "Done. / Time elapsed while rendering" lines to stderr and saves the image. */
typedef struct rth_render_config {
  rth_scene_opts scene;
  const char* output;
  int32_t device;
  int32_t precision; /* RT_PRECISION_* */
  int32_t pipeline;  /* RT_PIPELINE_* */
  int32_t quiet;
  uint64_t seed;
} rth_render_config;
int rth_render_and_save(const rth_render_config* cfg, rt_stats* stats_out);

/* One item of the stream that ThreadedRenderer::check_progress drains: type Pixel = (u32, u32, Rgb<u8>) (src/render.rs:39). */
typedef struct rth_pixel {
  uint32_t x, y; /* y counted from the top row, as the reference sends it (src/render.rs:239) */
  uint8_t rgb[3];
  uint8_t pad;
} rth_pixel;
/* A frame as that stream delivers it, for a host that sits where the GUI's check_progress loop sits (src/render.rs:84-108):
 * items first .. first + max_items of the frame's width * height pixels, in the order the reference's workers send theirs - the
 * row blocks of divide_into_ranges(height, workers) (src/render.rs:218,247-271) one after the other (the reference's run
 * concurrently, so its interleaving is not defined), inside a block from the row with the largest j down (src/render.rs:224: the
 * block's rows counted from the bottom of the image, in reverse), left to right inside a row.  `rgb8` is a frame as rt_render or
 * rt_progress_step writes it (row 0 on top); after a progress step every pixel has changed, so a GUI drains the frame again.
 * Returns the number of items written to `out` (0 past the end or on a bad argument). */
size_t rth_pixel_stream(const uint8_t* rgb8, uint32_t width, uint32_t height, uint32_t workers, size_t first, size_t max_items, rth_pixel* out);

const char* rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_HOST_H */
