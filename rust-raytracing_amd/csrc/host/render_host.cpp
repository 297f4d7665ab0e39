// RenderConfig::new + render_and_save (src/render.rs:27-37,278-284) with the GPU library standing where
// render() (src/render.rs:179-207) stands in the reference.
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "host_util.h"
#include "rtamd_host.h"

extern "C" int rth_render_and_save(const rth_render_config* cfg, rt_stats* stats_out) {
  if (!cfg || !cfg->output) { rth::g_err = "rth_render_and_save: bad argument"; return 1; }
  rth_scene* scene = rth_scene_create(&cfg->scene);  // RenderConfig::new -> SceneConfig::get_scene
  if (!scene) return 1;
  uint32_t w = 0, h = 0, spp = 0, depth = 0;
  rth_scene_config(scene, &w, &h, &spp, &depth);
  rt_context* ctx = nullptr;
  int rc = rt_create(&ctx, cfg->device);
  if (rc == RT_OK) rc = rt_upload_scene(ctx, rth_scene_desc(scene));
  std::vector<uint8_t> rgb((size_t)w * h * 3);
  rt_stats st{};
  if (rc == RT_OK) {
    rt_render_params p{};
    p.width = w; p.height = h; p.sample_begin = 0; p.sample_end = spp; p.max_depth = depth;
    p.precision = cfg->precision; p.pipeline = cfg->pipeline; p.samples_per_pass = 0; p.seed = cfg->seed;
    auto t0 = std::chrono::steady_clock::now();
    rc = rt_render(ctx, &p, rgb.data(), nullptr, &st);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc == RT_OK && !cfg->quiet) {
      std::fprintf(stderr, "\rScanlines remaining: 0 ");  // src/render.rs:198
      std::fprintf(stderr, "\nDone.\nTime elapsed while rendering: %.6fs", secs);  // src/render.rs:204
      std::fprintf(stderr, "\n[gpu] %.1f Mrays/s, %llu rays, %llu bounce-kernel launches\n", (double)st.rays / secs / 1e6,
                   (unsigned long long)st.rays, (unsigned long long)st.bounce_kernel_launches);
    }
  }
  if (rc != RT_OK) rth::g_err = rt_last_error();
  if (rc == RT_OK) {
    rc = rth_save_image(cfg->output, rgb.data(), w, h);  // save_image, src/render.rs:273-276
  }
  if (stats_out) *stats_out = st;
  if (ctx) rt_destroy(ctx);
  rth_scene_destroy(scene);
  return rc;
}
