/* The C ABI used from plain C11, with nothing in between: a hand-written scene of two rectangles (a grey floor and a light above
 * it) is uploaded and rendered at 16x16 on GPU 0, and the status codes, the statistics and the error paths are checked.
 *
 *   gcc -std=c11 -Wall -Wextra -Werror -pedantic -Iinclude tests/c_abi_smoke.c -Lrust-raytracing_amd/_build -lrtamd \
 *       -Wl,-rpath,$PWD/rust-raytracing_amd/_build -Wl,-rpath-link,/opt/rocm/lib -o c_abi_smoke
 *
 * `c_abi_smoke --no-gpu` stops after the calls that need no device (argument checks, the error string). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtamd.h"

#define CHECK(cond)                                                                          \
  do {                                                                                       \
    if (!(cond)) {                                                                           \
      fprintf(stderr, "c_abi_smoke: line %d: %s failed (%s)\n", __LINE__, #cond, rt_last_error()); \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int no_gpu = argc > 1 && strcmp(argv[1], "--no-gpu") == 0;
  CHECK(rt_version() != NULL && strlen(rt_version()) > 0);
  CHECK(rt_create(NULL, 0) == RT_E_INVALID);
  CHECK(strlen(rt_last_error()) > 0);
  CHECK(rt_upload_scene(NULL, NULL) == RT_E_INVALID);
  CHECK(rt_render(NULL, NULL, NULL, NULL, NULL) == RT_E_INVALID);
  if (no_gpu) {
    puts("c_abi_smoke ok (no device calls)");
    return 0;
  }

  /* src/camera.rs:21-60 for look_from (0, 1, -4), look_at (0, 1, 0), v_up (0, 1, 0), vfov 90, aspect 1, aperture 0, focus 1 */
  rt_scene_desc scene;
  memset(&scene, 0, sizeof scene);
  const rt_camera cam = {{0, 1, -4}, {-1, 0, -3}, {2, 0, 0}, {0, 2, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, -1}, 0.0, 0.0, 1.0};
  scene.camera = cam;
  scene.background.kind = RT_BG_SOLID;
  scene.background.c0[0] = 0.1; scene.background.c0[1] = 0.1; scene.background.c0[2] = 0.2;

  rt_texture textures[2];
  memset(textures, 0, sizeof textures);
  textures[0].tag = RT_TEX_SOLID; textures[0].color[0] = textures[0].color[1] = textures[0].color[2] = 0.7;
  textures[1].tag = RT_TEX_SOLID; textures[1].color[0] = textures[1].color[1] = textures[1].color[2] = 4.0;
  textures[0].even = textures[0].odd = textures[0].perlin = textures[0].image = -1;
  textures[1].even = textures[1].odd = textures[1].perlin = textures[1].image = -1;
  rt_material materials[2];
  memset(materials, 0, sizeof materials);
  materials[0].tag = RT_MAT_LAMBERTIAN; materials[0].texture = 0;
  materials[1].tag = RT_MAT_DIFFUSE_LIGHT; materials[1].texture = 1;
  rt_shape shapes[2];
  memset(shapes, 0, sizeof shapes);
  shapes[0].tag = RT_SHAPE_RECT_XZ; shapes[0].material = 0;  /* floor: x, z in [-5, 5] at y = 0 */
  shapes[0].p[0] = -5; shapes[0].p[1] = 5; shapes[0].p[2] = -5; shapes[0].p[3] = 5; shapes[0].p[4] = 0;
  shapes[1].tag = RT_SHAPE_RECT_XZ; shapes[1].material = 1;  /* light: x, z in [-1, 1] at y = 3 */
  shapes[1].p[0] = -1; shapes[1].p[1] = 1; shapes[1].p[2] = -1; shapes[1].p[3] = 1; shapes[1].p[4] = 3;
  rt_object objects[2];
  memset(objects, 0, sizeof objects);
  for (int i = 0; i < 2; ++i) {
    objects[i].geom = i; objects[i].medium_material = -1; objects[i].obj_id = i; objects[i].sub_id = -1;
    objects[i].cos_theta = 1.0;
  }
  /* BVHNode::new over the two objects (src/bvh.rs:19-76): one node, children are objects ~0 and ~1 */
  rt_bvh_node node = {{-5, -0.0001, -5}, {5, 3.0001, 5}, ~0, ~1};
  scene.root_node = 0;
  scene.n_nodes = 1; scene.nodes = &node;
  scene.n_objects = 2; scene.objects = objects;
  scene.n_shapes = 2; scene.shapes = shapes;
  scene.n_materials = 2; scene.materials = materials;
  scene.n_textures = 2; scene.textures = textures;

  rt_context* ctx = NULL;
  CHECK(rt_create(&ctx, 0) == RT_OK && ctx != NULL);
  rt_render_params p;
  memset(&p, 0, sizeof p);
  p.width = 16; p.height = 16; p.sample_begin = 0; p.sample_end = 8; p.max_depth = 10;
  p.precision = RT_PRECISION_F64; p.pipeline = RT_PIPELINE_AUTO; p.seed = 7;
  static uint8_t rgb[16 * 16 * 3];
  static double sums[16 * 16 * 3];
  rt_stats st;
  memset(&st, 0, sizeof st);
  CHECK(rt_render(ctx, &p, rgb, sums, &st) == RT_E_NO_SCENE);
  CHECK(rt_upload_scene(ctx, &scene) == RT_OK);
  CHECK(rt_render(ctx, &p, rgb, sums, &st) == RT_OK);
  CHECK(st.camera_rays == 16u * 16u * 8u);
  CHECK(st.rays >= st.camera_rays && st.rays <= st.camera_rays * 10u);
  /* the floor fills the lower half of the frame and is lit from above; the upper half sees the dark background */
  double top = 0, bottom = 0;
  for (int y = 0; y < 6; ++y)
    for (int x = 0; x < 16; ++x) top += sums[(y * 16 + x) * 3 + 1];
  for (int y = 10; y < 16; ++y)
    for (int x = 0; x < 16; ++x) bottom += sums[(y * 16 + x) * 3 + 1];
  CHECK(bottom > 0.0 && top > 0.0);
  int lit = 0;
  for (size_t i = 0; i < sizeof rgb; ++i) lit += rgb[i] != 0;
  CHECK(lit > 16 * 16);
  /* the same frame in single precision, through the split entry points and the context's own buffer */
  p.precision = RT_PRECISION_F32;
  CHECK(rt_render_accum(ctx, &p, NULL, &st) == RT_OK);
  CHECK(rt_resolve(ctx, NULL, 16, 16, 8, RT_PRECISION_F32, rgb) == RT_OK);
  /* argument checks with a scene in place */
  p.width = 1;
  CHECK(rt_render(ctx, &p, rgb, NULL, NULL) == RT_E_INVALID);
  p.width = 16; p.band_rows = 3; p.band_stride = 2; p.band_offset = 0;
  CHECK(rt_render(ctx, &p, rgb, NULL, NULL) == RT_E_INVALID);
  rt_destroy(ctx);
  puts("c_abi_smoke ok");
  return 0;
}
