// Command line front end: the three flags of src/cli.rs:7-21 with clap's relations
// (`--no-gui` requires --output and --scene; each of those requires --no-gui), dispatch as in
// src/main.rs:27-46.  The GUI branch (src/main.rs:48-55) does not exist here.  The reference has
// no flags for image size, samples or depth (they are per-scene constants); the extra flags below
// exist because the benchmark configurations are not reachable through the three original ones.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rtamd_host.h"

static void usage(const char* argv0) {
  std::fprintf(stderr,
               "Usage: %s --no-gui --output <FILE> --scene <SCENE> [options]\n\n"
               "Options:\n"
               "      --no-gui           Use without GUI\n"
               "  -o, --output <FILE>    Output file (.ppm, .png or .jpg)\n"
               "  -s, --scene <SCENE>    Scene to render [possible values: random, two-spheres, two-perlin-spheres, earth,\n"
               "                         simple-light, cornell-box, cornell-smoke, final-scene]\n"
               "      --width <N> --aspect <W/H> --spp <N> --depth <N>   override the scene's constants\n"
               "      --seed <N> --aux-seed <N> --bvh-seed <N>            random streams\n"
               "      --precision <f32|f64> --pipeline <auto|fused|sorted> --device <N> --earth <FILE.ppm>\n",
               argv0);
}

int main(int argc, char** argv) {
  bool no_gui = false;
  const char* output = nullptr;
  const char* scene_name = nullptr;
  rth_render_config cfg{};
  cfg.scene.aux_seed = 1;
  cfg.scene.bvh_seed = 7;
  cfg.precision = RT_PRECISION_F64;  // the reference's arithmetic; --precision f32 is the tolerance mode
  cfg.pipeline = RT_PIPELINE_AUTO;
  auto need = [&](int& i) -> const char* {
    if (i + 1 >= argc) { std::fprintf(stderr, "error: a value is required for '%s'\n", argv[i]); std::exit(2); }
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--no-gui") no_gui = true;
    else if (a == "-o" || a == "--output") output = need(i);
    else if (a == "-s" || a == "--scene") scene_name = need(i);
    else if (a == "--width") cfg.scene.width = std::atoi(need(i));
    else if (a == "--aspect") {
      const char* v = need(i);
      const char* slash = std::strchr(v, '/');
      cfg.scene.aspect = slash ? std::atof(v) / std::atof(slash + 1) : std::atof(v);
    }
    else if (a == "--spp") cfg.scene.spp = std::atoi(need(i));
    else if (a == "--depth") cfg.scene.depth = std::atoi(need(i));
    else if (a == "--seed") cfg.seed = std::strtoull(need(i), nullptr, 10);
    else if (a == "--aux-seed") cfg.scene.aux_seed = std::strtoull(need(i), nullptr, 10);
    else if (a == "--bvh-seed") cfg.scene.bvh_seed = std::strtoull(need(i), nullptr, 10);
    else if (a == "--device") cfg.device = std::atoi(need(i));
    else if (a == "--earth") cfg.scene.earth_path = need(i);
    else if (a == "--mega-grid") cfg.scene.mega_grid = std::atoi(need(i));
    else if (a == "--mega-view") cfg.scene.mega_view = std::atoi(need(i));
    else if (a == "--precision") cfg.precision = std::strcmp(need(i), "f32") == 0 ? RT_PRECISION_F32 : RT_PRECISION_F64;
    else if (a == "--pipeline") {
      const char* v = need(i);
      cfg.pipeline = !std::strcmp(v, "fused") ? RT_PIPELINE_FUSED : !std::strcmp(v, "sorted") ? RT_PIPELINE_SORTED : RT_PIPELINE_AUTO;
    }
    else if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
    else { std::fprintf(stderr, "error: unexpected argument '%s' found\n", argv[i]); usage(argv[0]); return 2; }
  }
  if (!no_gui) {
    if (output || scene_name) {  // cli.rs:15,19: `requires = "no_gui"`
      std::fprintf(stderr, "error: the following required arguments were not provided:\n  --no-gui\n");
      return 2;
    }
    std::fprintf(stderr, "error: this build has no GUI; run with --no-gui --output <FILE> --scene <SCENE>\n");
    return 2;
  }
  if (!output || !scene_name) {  // cli.rs:11: `requires_all = ["output", "scene"]`
    std::fprintf(stderr, "error: the following required arguments were not provided:\n%s%s", output ? "" : "  --output <FILE>\n",
                 scene_name ? "" : "  --scene <SCENE>\n");
    return 2;
  }
  cfg.scene.scene = rth_scene_from_name(scene_name);
  if (cfg.scene.scene < 0) {
    std::fprintf(stderr, "error: invalid value '%s' for '--scene <SCENE>'\n", scene_name);
    return 2;
  }
  cfg.output = output;
  if (rth_render_and_save(&cfg, nullptr) != 0) {  // main.rs:42-45
    std::fprintf(stderr, "Error: %s\n", rth_last_error());
    return 1;
  }
  return 0;
}
