// RenderConfig::new + render_and_save (src/render.rs:27-37,278-284) with the GPU library standing where
// render() (src/render.rs:179-207) stands in the reference.
#include <chrono>
#include <cstdio>
#include <string>
#include <utility>
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

// divide_into_ranges (src/render.rs:247-271): `rows` rows in `ranges` consecutive blocks, the first rows % ranges one row longer.
static std::vector<std::pair<uint32_t, uint32_t>> divide_into_ranges(uint32_t rows, uint32_t ranges) {
  std::vector<std::pair<uint32_t, uint32_t>> out;
  if (rows == 0) {
    out.emplace_back(0u, 0u);
  } else if (rows <= ranges) {
    for (uint32_t i = 0; i < rows; ++i) out.emplace_back(i, i + 1u);
  } else {
    const uint32_t group = rows / ranges;
    uint32_t remain = rows % ranges, end = 0;
    for (uint32_t k = 0; k < ranges; ++k) {
      const uint32_t start = end;
      end = start + group;
      if (remain > 0) { end += 1u; remain -= 1u; }
      out.emplace_back(start, end);
    }
  }
  return out;
}

extern "C" size_t rth_pixel_stream(const uint8_t* rgb8, uint32_t width, uint32_t height, uint32_t workers, size_t first, size_t max_items,
                                   rth_pixel* out) {
  if (!rgb8 || !out || width == 0 || height == 0 || workers == 0) return 0;
  const size_t total = (size_t)width * height;
  size_t written = 0, index = 0;
  for (const auto& block : divide_into_ranges(height, workers)) {
    const size_t in_block = (size_t)(block.second - block.first) * width;
    if (index + in_block <= first) { index += in_block; continue; }  // the whole block lies before `first`
    for (uint32_t j = block.second; j-- > block.first;) {  // for j in range.rev() (src/render.rs:224)
      const uint32_t y = height - 1u - j;                  // src/render.rs:239
      for (uint32_t i = 0; i < width; ++i, ++index) {
        if (index < first) continue;
        if (written == max_items || index >= total) return written;
        const uint8_t* px = rgb8 + ((size_t)y * width + i) * 3u;
        out[written++] = rth_pixel{i, y, {px[0], px[1], px[2]}, 0};
      }
    }
  }
  return written;
}

