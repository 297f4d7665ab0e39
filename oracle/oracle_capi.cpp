// CPU ORACLE — TEST INFRASTRUCTURE ONLY (see rt_oracle.hpp).
//
// Scene construction (scenes.rs), the per-pixel sample loop (render.rs:209-245)
// and a flat C API for ctypes.  Built by oracle/Makefile into
// oracle/_build/liboracle.so.  Never linked into the product library.
#include "rt_oracle.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>

namespace orc {

enum SceneKind { SC_RANDOM = 0, SC_TWO_SPHERES, SC_TWO_PERLIN, SC_EARTH, SC_SIMPLE_LIGHT, SC_CORNELL_BOX, SC_CORNELL_SMOKE, SC_FINAL, SC_MEGA_SPHERES };

template <class R> struct Scene {
  Camera<R> camera;
  HittableList<R> world;
  Background<R> background;
  uint32_t image_width = 0;
  double aspect_ratio = 1;
  uint32_t samples_per_pixel = 0, max_depth = 0;
  std::shared_ptr<BVHNode<R>> bvh;  // built by render_with_threadpool in the reference (render.rs:211)
  std::vector<std::shared_ptr<Material<R>>> materials;
  uint32_t width() const { return image_width; }
  uint32_t height() const { return (uint32_t)((double)image_width / aspect_ratio); }  // scenes.rs:135-138
};

struct SceneOpts {
  int kind;
  int width;          // 0 = scene default
  double aspect;      // 0 = scene default
  int spp;            // 0 = scene default
  int depth;          // 0 = scene default (50)
  uint64_t aux_seed;  // stands in for every thread_rng() draw made while building the scene
  uint64_t bvh_seed;  // stands in for thread_rng() in BVHNode::new
  const uint8_t* earth;  // decoded earthmap RGB8 or null (-> cyan, texture.rs:131-133)
  int earth_w, earth_h;
  int mega_grid;      // SC_MEGA_SPHERES: a, b in [-grid, grid)
  int mega_view;      // SC_MEGA_SPHERES: 1 = camera above the centre of the field, looking down
};

template <class R> struct Builder {
  Scene<R>& sc;
  Pcg32 aux;
  const SceneOpts& o;
  Builder(Scene<R>& s, const SceneOpts& opts) : sc(s), aux(Pcg32::seed_from_u64(opts.aux_seed)), o(opts) {}
  using MP = std::shared_ptr<Material<R>>;
  using TP = std::shared_ptr<Texture<R>>;
  template <class M> MP reg(std::shared_ptr<M> m) {
    m->id = (int)sc.materials.size();
    sc.materials.push_back(m);
    return m;
  }
  TP solid(V3<R> c) { return std::make_shared<SolidColor<R>>(c); }
  MP lambert(TP t) { auto m = std::make_shared<Lambertian<R>>(); m->albedo = t; return reg(m); }
  MP lambert(V3<R> c) { return lambert(solid(c)); }
  MP metal(V3<R> c, R f) { return reg(std::make_shared<Metal<R>>(c, f)); }
  MP glass(R ir) { return reg(std::make_shared<Dielectric<R>>(ir)); }
  MP light(V3<R> c) { auto m = std::make_shared<DiffuseLight<R>>(); m->emit = solid(c); return reg(m); }
  MP iso(V3<R> c) { auto m = std::make_shared<Isotropic<R>>(); m->albedo = solid(c); return reg(m); }
  TP checker(V3<R> even, V3<R> odd) {  // texture.rs:47-51
    auto t = std::make_shared<CheckerTexture<R>>();
    t->even = solid(even); t->odd = solid(odd);
    return t;
  }
  V3<R> aux_random() { double a = aux.gen_f64(), b = aux.gen_f64(), c = aux.gen_f64(); return V3<R>((R)a, (R)b, (R)c); }  // vec3.rs:19-22
  V3<R> aux_random_range(double lo, double hi) {  // vec3.rs:24-31
    double a = aux.gen_range_f64(lo, hi), b = aux.gen_range_f64(lo, hi), c = aux.gen_range_f64(lo, hi);
    return V3<R>((R)a, (R)b, (R)c);
  }
  TP noise(R scale) {  // texture.rs:71-76 + perlin.rs:14-30,74-93
    auto t = std::make_shared<NoiseTexture<R>>();
    t->scale = scale;
    for (int i = 0; i < 256; ++i) {
      V3<double> v;
      v[0] = aux.gen_range_f64(-1.0, 1.0); v[1] = aux.gen_range_f64(-1.0, 1.0); v[2] = aux.gen_range_f64(-1.0, 1.0);
      V3<double> uv = unit_vector(v);
      t->noise.ran_vec[i] = V3<R>((R)uv[0], (R)uv[1], (R)uv[2]);
    }
    for (int* perm : {t->noise.perm_x, t->noise.perm_y, t->noise.perm_z}) {
      for (int i = 0; i < 256; ++i) perm[i] = i;
      for (int i = 255; i >= 1; --i) {  // gen_range(0..i): exclusive upper bound (perlin.rs:89-92)
        int target = (int)aux.gen_range_u64_exclusive(0, (uint64_t)i);
        std::swap(perm[i], perm[target]);
      }
    }
    return t;
  }
  TP image() {  // texture.rs:97-125 (decode done by the caller)
    auto t = std::make_shared<ImageTexture<R>>();
    if (o.earth && o.earth_w > 0 && o.earth_h > 0) {
      t->width = (uint32_t)o.earth_w; t->height = (uint32_t)o.earth_h; t->bytes_per_scanline = 3u * t->width;
      t->data.assign(o.earth, o.earth + (size_t)3 * o.earth_w * o.earth_h);
    }
    return t;
  }
  HPtr<R> sphere(V3<R> c, R r, MP m) { return std::make_shared<Sphere<R>>(c, r, m); }

  void random_scene() {  // scenes.rs:141-247
    auto ground = lambert(checker(V3<R>(0.2, 0.3, 0.1), V3<R>(0.9, 0.9, 0.9)));
    sc.world.add(sphere(V3<R>(0, -1000, 0), 1000, ground));
    Pcg32 rng = Pcg32::seed_from_u64(2);
    for (int a = -11; a < 11; ++a) {
      for (int b = -11; b < 11; ++b) {
        double choose_mat = rng.gen_f64();
        double cx = (double)a + 0.9 * rng.gen_f64();
        double cz = (double)b + 0.9 * rng.gen_f64();
        V3<double> center(cx, 0.2, cz);
        if ((center - V3<double>(4.0, 0.2, 0.0)).length() > 0.9) {
          V3<R> c((R)cx, (R)0.2, (R)cz);
          if (choose_mat < 0.8) {
            V3<R> albedo = aux_random() * aux_random();
            auto m = lambert(albedo);
            double dy = rng.gen_range_f64(0.0, 0.5);
            V3<R> c2 = c + V3<R>(0, (R)dy, 0);
            sc.world.add(std::make_shared<MovingSphere<R>>(c, c2, (R)0, (R)1, (R)0.2, m));
          } else if (choose_mat < 0.95) {
            V3<R> albedo = aux_random_range(0.5, 1.0);
            double fuzz = rng.gen_range_f64(0.0, 0.5);
            sc.world.add(sphere(c, (R)0.2, metal(albedo, (R)fuzz)));
          } else {
            sc.world.add(sphere(c, (R)0.2, glass((R)1.5)));
          }
        }
      }
      // scenes.rs:197-217 — inside the `for a` loop: the three big spheres are added 22 times.
      sc.world.add(sphere(V3<R>(0, 1, 0), 1, glass((R)1.5)));
      sc.world.add(sphere(V3<R>(-4, 1, 0), 1, lambert(V3<R>(0.4, 0.2, 0.1))));
      sc.world.add(sphere(V3<R>(4, 1, 0), 1, metal(V3<R>(0.7, 0.6, 0.5), 0)));
    }
    auto material_center = lambert(V3<R>(0.1, 0.2, 0.5));
    auto material_left = glass((R)1.5);
    auto material_right = metal(V3<R>(0.8, 0.6, 0.2), 0);
    sc.world.add(sphere(V3<R>(0, 0, -1), (R)0.5, material_center));
    sc.world.add(sphere(V3<R>(-1, 0, -1), (R)0.5, material_left));
    sc.world.add(sphere(V3<R>(-1, 0, -1), (R)-0.45, material_left));
    sc.world.add(sphere(V3<R>(1, 0, -1), (R)0.5, material_right));
  }
  void two_spheres() {  // scenes.rs:249-271
    auto m = lambert(checker(V3<R>(0.2, 0.3, 0.1), V3<R>(0.9, 0.9, 0.9)));
    sc.world.add(sphere(V3<R>(0, -10, 0), 10, m));
    sc.world.add(sphere(V3<R>(0, 10, 0), 10, m));
  }
  void two_perlin_spheres() {  // scenes.rs:273-292
    auto m = lambert(noise(4));
    sc.world.add(sphere(V3<R>(0, -1000, 0), 1000, m));
    sc.world.add(sphere(V3<R>(0, 2, 0), 2, m));
  }
  void earth() {  // scenes.rs:294-308
    auto m = lambert(image());
    sc.world.add(sphere(V3<R>(0, 0, 0), 2, m));
  }
  void simple_light() {  // scenes.rs:310-339
    auto m = lambert(noise(4));
    sc.world.add(sphere(V3<R>(0, -1000, 0), 1000, m));
    sc.world.add(sphere(V3<R>(0, 2, 0), 2, m));
    auto dl = light(V3<R>(4, 4, 4));
    sc.world.add(std::make_shared<XYRect<R>>(3, 5, 1, 3, -2, dl));
    sc.world.add(sphere(V3<R>(0, 10, 0), 2, dl));
  }
  HPtr<R> rotated_box(V3<R> hi, R angle, V3<R> offset, MP m) {
    HPtr<R> b = std::make_shared<BoxShape<R>>(V3<R>(0, 0, 0), hi, m);
    b = std::make_shared<RotateY<R>>(b, angle);
    b = std::make_shared<Translate<R>>(b, offset);
    return b;
  }
  void cornell_box() {  // scenes.rs:341-381
    auto red = lambert(V3<R>(0.65, 0.05, 0.05));
    auto white = lambert(V3<R>(0.73, 0.73, 0.73));
    auto green = lambert(V3<R>(0.12, 0.45, 0.15));
    auto lt = light(V3<R>(15, 15, 15));
    sc.world.add(std::make_shared<YZRect<R>>(0, 555, 0, 555, 555, green));
    sc.world.add(std::make_shared<YZRect<R>>(0, 555, 0, 555, 0, red));
    sc.world.add(std::make_shared<XZRect<R>>(213, 343, 227, 332, 554, lt));
    sc.world.add(std::make_shared<XZRect<R>>(0, 555, 0, 555, 0, white));
    sc.world.add(std::make_shared<XZRect<R>>(0, 555, 0, 555, 555, white));
    sc.world.add(std::make_shared<XYRect<R>>(0, 555, 0, 555, 555, white));
    sc.world.add(rotated_box(V3<R>(165, 330, 165), 15, V3<R>(265, 0, 295), white));
    sc.world.add(rotated_box(V3<R>(168, 165, 165), -18, V3<R>(130, 0, 65), white));
  }
  void cornell_smoke() {  // scenes.rs:383-430
    auto red = lambert(V3<R>(0.65, 0.05, 0.05));
    auto white = lambert(V3<R>(0.73, 0.73, 0.73));
    auto green = lambert(V3<R>(0.12, 0.45, 0.15));
    auto lt = light(V3<R>(7, 7, 7));
    sc.world.add(std::make_shared<YZRect<R>>(0, 555, 0, 555, 555, green));
    sc.world.add(std::make_shared<YZRect<R>>(0, 555, 0, 555, 0, red));
    sc.world.add(std::make_shared<XZRect<R>>(113, 443, 127, 432, 554, lt));
    sc.world.add(std::make_shared<XZRect<R>>(0, 555, 0, 555, 555, white));
    sc.world.add(std::make_shared<XZRect<R>>(0, 555, 0, 555, 0, white));
    sc.world.add(std::make_shared<XYRect<R>>(0, 555, 0, 555, 555, white));
    auto box1 = rotated_box(V3<R>(165, 330, 165), 15, V3<R>(265, 0, 295), white);
    auto box2 = rotated_box(V3<R>(168, 165, 165), -18, V3<R>(130, 0, 65), white);
    sc.world.add(std::make_shared<ConstantMedium<R>>(box1, (R)0.01, iso(V3<R>(0, 0, 0))));
    sc.world.add(std::make_shared<ConstantMedium<R>>(box2, (R)0.01, iso(V3<R>(1, 1, 1))));
  }
  void final_scene(Pcg32& bvh_rng) {  // scenes.rs:432-556
    HittableList<R> boxes1;
    auto ground = lambert(V3<R>(0.48, 0.83, 0.53));
    Pcg32 rng = Pcg32::seed_from_u64(3);
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; ++i)
      for (int j = 0; j < boxes_per_side; ++j) {
        double w = 100.0;
        double x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
        double x1 = x0 + w, y1 = rng.gen_range_f64(1.0, 101.0), z1 = z0 + w;
        boxes1.add(std::make_shared<BoxShape<R>>(V3<R>((R)x0, (R)y0, (R)z0), V3<R>((R)x1, (R)y1, (R)z1), ground));
      }
    for (size_t i = 0; i < boxes1.objects.size(); ++i) boxes1.objects[i]->label(-1, (int)i);
    sc.world.add(std::make_shared<BVHNode<R>>(boxes1.objects, 0, boxes1.objects.size(), (R)0, (R)1, bvh_rng, 1));
    auto lt = light(V3<R>(7, 7, 7));
    sc.world.add(std::make_shared<XZRect<R>>(123, 423, 147, 412, 554, lt));
    V3<R> center1(400, 400, 400);
    V3<R> center2 = center1 + V3<R>(30, 0, 0);
    sc.world.add(std::make_shared<MovingSphere<R>>(center1, center2, (R)0, (R)1, (R)50, lambert(V3<R>(0.7, 0.3, 0.1))));
    auto dielectric = glass((R)1.5);
    sc.world.add(sphere(V3<R>(260, 150, 45), 50, dielectric));
    sc.world.add(sphere(V3<R>(0, 150, 145), 50, metal(V3<R>(0.8, 0.8, 0.9), 1)));
    auto boundary = sphere(V3<R>(360, 150, 145), 70, dielectric);
    sc.world.add(boundary);
    sc.world.add(std::make_shared<ConstantMedium<R>>(boundary, (R)0.2, iso(V3<R>(0.2, 0.4, 0.9))));
    auto boundary2 = sphere(V3<R>(0, 0, 0), 5000, dielectric);
    sc.world.add(std::make_shared<ConstantMedium<R>>(boundary2, (R)0.0001, iso(V3<R>(1, 1, 1))));
    sc.world.add(sphere(V3<R>(400, 200, 400), 100, lambert(image())));
    sc.world.add(sphere(V3<R>(220, 280, 300), 80, lambert(noise((R)0.1))));
    HittableList<R> boxes2;
    auto white = lambert(V3<R>(0.73, 0.73, 0.73));
    for (int i = 0; i < 1000; ++i) boxes2.add(sphere(aux_random_range(0.0, 165.0), 10, white));
    for (size_t i = 0; i < boxes2.objects.size(); ++i) boxes2.objects[i]->label(-1, (int)i);
    HPtr<R> h = std::make_shared<BVHNode<R>>(boxes2.objects, 0, boxes2.objects.size(), (R)0, (R)1, bvh_rng, 1);
    h = std::make_shared<RotateY<R>>(h, (R)15);
    sc.world.add(std::make_shared<Translate<R>>(h, V3<R>(-100, 270, 395)));
  }
  // Not in the reference: random_scene generalised to a (2*grid)^2 lattice; stresses BVH memory traffic.
  void mega_spheres(int grid) {
    auto ground = lambert(checker(V3<R>(0.2, 0.3, 0.1), V3<R>(0.9, 0.9, 0.9)));
    sc.world.add(sphere(V3<R>(0, -100000, 0), 100000, ground));
    Pcg32 rng = Pcg32::seed_from_u64(5);
    auto g = glass((R)1.5);
    for (int a = -grid; a < grid; ++a)
      for (int b = -grid; b < grid; ++b) {
        double choose_mat = rng.gen_f64();
        double cx = (double)a + 0.9 * rng.gen_f64();
        double cz = (double)b + 0.9 * rng.gen_f64();
        V3<R> c((R)cx, (R)0.2, (R)cz);
        if (choose_mat < 0.8) {
          V3<R> albedo = aux_random() * aux_random();
          double dy = rng.gen_range_f64(0.0, 0.5);
          sc.world.add(std::make_shared<MovingSphere<R>>(c, c + V3<R>(0, (R)dy, 0), (R)0, (R)1, (R)0.2, lambert(albedo)));
        } else if (choose_mat < 0.95) {
          V3<R> albedo = aux_random_range(0.5, 1.0);
          double fuzz = rng.gen_range_f64(0.0, 0.5);
          sc.world.add(sphere(c, (R)0.2, metal(albedo, (R)fuzz)));
        } else {
          sc.world.add(sphere(c, (R)0.2, g));
        }
      }
  }
};

template <class R> std::unique_ptr<Scene<R>> build_scene(const SceneOpts& o) {  // scenes.rs:44-133
  auto sp = std::make_unique<Scene<R>>();
  Scene<R>& sc = *sp;
  Builder<R> b(sc, o);
  Pcg32 bvh_rng = Pcg32::seed_from_u64(o.bvh_seed);
  V3<R> v_up(0, 1, 0);
  R v_fov = 20, aperture = 0, time0 = 0, time1 = 1, focus_dist = 10;
  V3<R> look_from(13, 2, 3), look_at(0, 0, 0);
  sc.background = Background<R>{0, V3<R>(), V3<R>()};
  sc.image_width = 400;
  sc.aspect_ratio = 16.0 / 9.0;
  sc.samples_per_pixel = 100;
  sc.max_depth = 50;
  switch (o.kind) {
    case SC_RANDOM:
      b.random_scene();
      sc.background = Background<R>{1, V3<R>(0.5, 0.7, 1.0), V3<R>(1, 1, 1)};
      aperture = (R)0.1;
      break;
    case SC_TWO_SPHERES: b.two_spheres(); sc.background = Background<R>{0, V3<R>(0.70, 0.80, 1.00), V3<R>()}; break;
    case SC_TWO_PERLIN: b.two_perlin_spheres(); sc.background = Background<R>{0, V3<R>(0.70, 0.80, 1.00), V3<R>()}; break;
    case SC_EARTH: b.earth(); sc.background = Background<R>{0, V3<R>(0.70, 0.80, 1.00), V3<R>()}; break;
    case SC_SIMPLE_LIGHT:
      b.simple_light();
      look_from = V3<R>(26, 3, 6); look_at = V3<R>(0, 2, 0);
      sc.samples_per_pixel = 400;
      break;
    case SC_CORNELL_BOX:
    case SC_CORNELL_SMOKE:
      if (o.kind == SC_CORNELL_BOX) b.cornell_box(); else b.cornell_smoke();
      look_from = V3<R>(278, 278, -800); look_at = V3<R>(278, 278, 0);
      v_fov = 40; sc.aspect_ratio = 1.0; sc.image_width = 600; sc.samples_per_pixel = 200;
      break;
    case SC_FINAL:
      b.final_scene(bvh_rng);
      look_from = V3<R>(478, 278, -600); look_at = V3<R>(278, 278, 0);
      v_fov = 40; sc.aspect_ratio = 1.0; sc.image_width = 800; sc.samples_per_pixel = 10000;
      break;
    case SC_MEGA_SPHERES:
      b.mega_spheres(o.mega_grid > 0 ? o.mega_grid : 500);
      sc.background = Background<R>{1, V3<R>(0.5, 0.7, 1.0), V3<R>(1, 1, 1)};
      look_from = V3<R>(13, 6, 3); v_fov = 40; aperture = (R)0.1;
      if (o.mega_view == 1) {
        const R h = (R)2.8 * (R)(o.mega_grid > 0 ? o.mega_grid : 500);
        look_from = V3<R>(0, h, 0); v_up = V3<R>(0, 0, -1); focus_dist = h;
      }
      sc.image_width = 3840; sc.samples_per_pixel = 1024;
      break;
    default: return nullptr;
  }
  if (o.width > 0) sc.image_width = (uint32_t)o.width;
  if (o.aspect > 0) sc.aspect_ratio = o.aspect;
  if (o.spp > 0) sc.samples_per_pixel = (uint32_t)o.spp;
  if (o.depth > 0) sc.max_depth = (uint32_t)o.depth;
  sc.camera = Camera<R>(look_from, look_at, v_up, v_fov, (R)sc.aspect_ratio, aperture, focus_dist, time0, time1);
  for (size_t i = 0; i < sc.world.objects.size(); ++i) sc.world.objects[i]->label((int)i, -1);
  // render.rs:211 — BVHNode::new(&scene.world.objects, 0.0, 1.0)
  sc.bvh = std::make_shared<BVHNode<R>>(sc.world.objects, 0, sc.world.objects.size(), (R)0, (R)1, bvh_rng);
  return sp;
}

// render.rs:247-271
inline std::vector<std::pair<uint32_t, uint32_t>> divide_into_ranges(uint32_t rows, uint32_t ranges) {
  std::vector<std::pair<uint32_t, uint32_t>> out;
  if (rows == 0) out.push_back({0, 0});
  else if (rows <= ranges) for (uint32_t i = 0; i < rows; ++i) out.push_back({i, i + 1});
  else {
    uint32_t group = rows / ranges, remain = rows % ranges, end = 0;
    for (uint32_t r = 0; r < ranges; ++r) {
      uint32_t start = end;
      end = start + group;
      if (remain > 0) { ++end; --remain; }
      out.push_back({start, end});
    }
  }
  return out;
}

// One camera sample: render.rs:230-236.  (i, j) with j counted from the bottom row.
template <class R>
V3<R> sample_radiance(const Scene<R>& sc, uint32_t W, uint32_t H, uint32_t i, uint32_t j, uint32_t s, uint64_t seed, Ray<R>* ray_out = nullptr) {
  uint32_t y = H - 1 - j;
  uint32_t pixel = y * W + i;
  Stream cam(seed, pixel, s, 0, STAGE_CAMERA);
  R u_rand = gen01_at<R>(cam, 0);
  R v_rand = gen01_at<R>(cam, 1);
  const uint32_t less = g_mutation == MUT_DIVIDE_BY_W_H ? 0u : 1u;
  R u = ((R)i + u_rand) / (R)(W - less);  // render.rs:232-233
  R v = ((R)j + v_rand) / (R)(H - less);
  Ray<R> ray = sc.camera.get_ray(u, v, cam);
  ++g_cnt.camera_rays;
  if (ray_out) { *ray_out = ray; return V3<R>(); }
  PathKey k{seed, pixel, s};
  return ray_color<R>(ray, sc.background, *sc.bvh, sc.max_depth, sc.max_depth, k);
}

// render.rs:209-245 with the reference's schedule: contiguous row blocks, one per thread.
template <class R>
void render_accum(const Scene<R>& sc, uint32_t s_begin, uint32_t s_end, uint64_t seed, int threads, double* accum, Counters* total,
                  uint32_t row_begin, uint32_t row_end) {
  uint32_t W = sc.width(), H = sc.height();
  if (row_end > H) row_end = H;
  auto ranges = divide_into_ranges(row_end - row_begin, (uint32_t)std::max(1, threads));
  std::vector<std::thread> pool;
  std::vector<Counters> cnts(ranges.size());
  for (size_t t = 0; t < ranges.size(); ++t) {
    pool.emplace_back([&, t]() {
      g_cnt = Counters{};
      for (uint32_t jj = ranges[t].second; jj-- > ranges[t].first;) {  // range.rev()
        uint32_t j = row_begin + jj;
        for (uint32_t i = 0; i < W; ++i) {
          V3<R> pixel_color;
          for (uint32_t s = s_begin; s < s_end; ++s) {
            V3<R> c = sample_radiance<R>(sc, W, H, i, j, s, seed);
            pixel_color = pixel_color + c;
          }
          uint32_t y = H - 1 - j;
          double* a = accum + ((size_t)y * W + i) * 3;
          a[0] += (double)pixel_color[0]; a[1] += (double)pixel_color[1]; a[2] += (double)pixel_color[2];
        }
      }
      cnts[t] = g_cnt;
    });
  }
  for (auto& th : pool) th.join();
  if (total) for (auto& c : cnts) total->add(c);
}

}  // namespace orc

// ===========================================================================
// C API (ctypes)
// ===========================================================================
using namespace orc;

struct orc_scene {
  std::unique_ptr<Scene<double>> d;
  std::unique_ptr<Scene<float>> f;
};

struct orc_ray {
  double o[3], d[3], time, tmin, tmax;
  uint32_t pixel, sample, bounce, pad;
};
struct orc_hit {
  int32_t hit, obj, sub, side, material, front_face;
  double t, p[3], n[3], u, v;
};

// `skip` (optional): per ray, the winner of an earlier query; that surface - and every exact copy of its object, by
// `class_of_obj` - is left out, so the query returns the runner-up.
template <class R>
static void first_hit_impl(const Scene<R>& sc, uint64_t seed, size_t n, const orc_ray* rays, orc_hit* hits, const orc_hit* skip = nullptr,
                           const int32_t* class_of_obj = nullptr) {
  for (size_t r = 0; r < n; ++r) {
    Ray<R> ray{V3<R>((R)rays[r].o[0], (R)rays[r].o[1], (R)rays[r].o[2]), V3<R>((R)rays[r].d[0], (R)rays[r].d[1], (R)rays[r].d[2]), (R)rays[r].time};
    Stream media(seed, rays[r].pixel, rays[r].sample, rays[r].bounce, STAGE_MEDIA);
    HitCtx ctx{&media};
    if (skip && class_of_obj && skip[r].hit && skip[r].obj >= 0) {
      ctx.class_of_obj = class_of_obj;
      ctx.skip_class = class_of_obj[skip[r].obj]; ctx.skip_sub = skip[r].sub; ctx.skip_side = skip[r].side;
    }
    HitRecord<R> rec;
    bool h = sc.bvh->hit(ray, (R)rays[r].tmin, (R)rays[r].tmax, rec, ctx);
    orc_hit& o = hits[r];
    std::memset(&o, 0, sizeof(o));
    o.hit = h ? 1 : 0;
    if (!h) { o.obj = o.sub = o.side = o.material = -1; continue; }
    o.obj = rec.obj; o.sub = rec.sub; o.side = rec.side; o.material = rec.material ? rec.material->id : -1;
    o.front_face = rec.front_face ? 1 : 0;
    o.t = (double)rec.t; o.u = (double)rec.u; o.v = (double)rec.v;
    for (int k = 0; k < 3; ++k) { o.p[k] = (double)rec.p[k]; o.n[k] = (double)rec.normal[k]; }
  }
}
extern "C" {

orc_scene* orc_scene_create(int kind, int width, double aspect, int spp, int depth, uint64_t aux_seed, uint64_t bvh_seed,
                            const uint8_t* earth, int earth_w, int earth_h, int mega_grid, int mega_view, int want_f32) {
  SceneOpts o{kind, width, aspect, spp, depth, aux_seed, bvh_seed, earth, earth_w, earth_h, mega_grid, mega_view};
  auto* s = new orc_scene;
  s->d = build_scene<double>(o);
  if (!s->d) { delete s; return nullptr; }
  if (want_f32) s->f = build_scene<float>(o);
  return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
void orc_scene_info(const orc_scene* s, int* w, int* h, int* spp, int* depth, int* n_objects, int* n_materials) {
  *w = (int)s->d->width(); *h = (int)s->d->height(); *spp = (int)s->d->samples_per_pixel; *depth = (int)s->d->max_depth;
  *n_objects = (int)s->d->world.objects.size(); *n_materials = (int)s->d->materials.size();
}
// Camera fields in declaration order (camera.rs:6-17): origin, llc, horizontal, vertical, u, v, w, lens_radius, t_open, t_close = 24 doubles.
void orc_scene_camera(const orc_scene* s, double* out) {
  const auto& c = s->d->camera;
  const V3<double>* vs[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
  for (int i = 0; i < 7; ++i) for (int k = 0; k < 3; ++k) out[i * 3 + k] = (*vs[i])[k];
  out[21] = c.lens_radius; out[22] = c.shutter_open; out[23] = c.shutter_close;
}
// Top-level object bounding boxes at (0,1): 6 doubles each (for host-flattener cross checks).
void orc_scene_object_boxes(const orc_scene* s, double* out) {
  const auto& objs = s->d->world.objects;
  for (size_t i = 0; i < objs.size(); ++i) {
    auto b = objs[i]->bounding_box(0, 1);
    for (int k = 0; k < 3; ++k) { out[i * 6 + k] = b ? b->minimum[k] : 0; out[i * 6 + 3 + k] = b ? b->maximum[k] : 0; }
  }
}
void orc_scene_root_box(const orc_scene* s, double* out) {
  for (int k = 0; k < 3; ++k) { out[k] = s->d->bvh->box.minimum[k]; out[3 + k] = s->d->bvh->box.maximum[k]; }
}

// Primary rays exactly as the sample loop builds them (render.rs:230-234): pixel = y*W + x with y from the top.
void orc_camera_rays(const orc_scene* s, uint64_t seed, size_t n, const uint32_t* pixel, const uint32_t* sample, orc_ray* out) {
  const auto& sc = *s->d;
  uint32_t W = sc.width(), H = sc.height();
  for (size_t r = 0; r < n; ++r) {
    uint32_t y = pixel[r] / W, i = pixel[r] % W, j = H - 1 - y;
    Ray<double> ray;
    sample_radiance<double>(sc, W, H, i, j, sample[r], seed, &ray);
    for (int k = 0; k < 3; ++k) { out[r].o[k] = ray.origin[k]; out[r].d[k] = ray.direction[k]; }
    out[r].time = ray.time; out[r].tmin = 0.001; out[r].tmax = std::numeric_limits<double>::infinity();
    out[r].pixel = pixel[r]; out[r].sample = sample[r]; out[r].bounce = 0; out[r].pad = 0;
  }
}

// world.hit(ray, tmin, tmax) through the BVH (bvh.rs:108-119).  precision: 0 = f32 study build, 1 = f64 (the oracle).
int orc_first_hit(const orc_scene* s, uint64_t seed, size_t n, const orc_ray* rays, orc_hit* hits, int precision) {
  if (precision == 1) first_hit_impl<double>(*s->d, seed, n, rays, hits);
  else { if (!s->f) return 1; first_hit_impl<float>(*s->f, seed, n, rays, hits); }
  return 0;
}

// The same query with the surface each ray hit first (`winners`, from orc_first_hit) left out, together with every object of the
// same class (class_of_obj[n_objects]: exact copies of an object share a class): the runner-up candidate, for the near-tie sets of
// the parity tests.  f64 only.
int orc_first_hit_runner_up(const orc_scene* s, uint64_t seed, size_t n, const orc_ray* rays, const orc_hit* winners, const int32_t* class_of_obj,
                            orc_hit* hits) {
  first_hit_impl<double>(*s->d, seed, n, rays, hits, winners, class_of_obj);
  return 0;
}

// Accumulate linear radiance sums for samples [s_begin, s_end) of rows [row_begin,row_end) (rows from the bottom, as j in render.rs:225).
// accum: H*W*3 doubles, row 0 = top.  counters: sizeof(Counters)/8 uint64 or null.  Returns seconds.
double orc_render(const orc_scene* s, uint32_t s_begin, uint32_t s_end, uint64_t seed, int threads, double* accum, uint64_t* counters,
                  uint32_t row_begin, uint32_t row_end, int precision) {
  Counters total;
  auto t0 = std::chrono::steady_clock::now();
  if (precision == 1) render_accum<double>(*s->d, s_begin, s_end, seed, threads, accum, &total, row_begin, row_end);
  else { if (!s->f) return -1.0; render_accum<float>(*s->f, s_begin, s_end, seed, threads, accum, &total, row_begin, row_end); }
  auto t1 = std::chrono::steady_clock::now();
  if (counters) std::memcpy(counters, &total, sizeof(Counters));
  return std::chrono::duration<double>(t1 - t0).count();
}
// 0 = the reference's behaviour; 1..3 = a deliberate fault (orc::Mutation) that the known-answer tests must notice
void orc_set_mutation(int m) { g_mutation = m; }
int orc_counters_len(void) { return (int)(sizeof(Counters) / sizeof(uint64_t)); }

// Per-sample radiance for selected (pixel, sample) pairs: 4 doubles each = r, g, b, closest-hit queries of the path.
void orc_sample_radiance(const orc_scene* s, uint64_t seed, size_t n, const uint32_t* pixel, const uint32_t* sample, double* out) {
  const auto& sc = *s->d;
  uint32_t W = sc.width(), H = sc.height();
  for (size_t r = 0; r < n; ++r) {
    uint32_t y = pixel[r] / W, i = pixel[r] % W, j = H - 1 - y;
    const uint64_t rays_before = g_cnt.rays;
    V3<double> c = sample_radiance<double>(sc, W, H, i, j, sample[r], seed);
    out[r * 4] = c[0]; out[r * 4 + 1] = c[1]; out[r * 4 + 2] = c[2]; out[r * 4 + 3] = (double)(g_cnt.rays - rays_before);
  }
}

// The closest-hit queries of one path: up to max_vertices records of 11 doubles (origin[3], direction[3], t, obj, sub, hit,
// front_face); returns how many the path made.
int orc_sample_path(const orc_scene* s, uint64_t seed, uint32_t pixel, uint32_t sample, int max_vertices, double* out) {
  const auto& sc = *s->d;
  uint32_t W = sc.width(), H = sc.height();
  std::vector<PathVertex> trace;
  g_path_trace = &trace;
  uint32_t y = pixel / W, i = pixel % W, j = H - 1 - y;
  (void)sample_radiance<double>(sc, W, H, i, j, sample, seed);
  g_path_trace = nullptr;
  for (int k = 0; k < max_vertices && k < (int)trace.size(); ++k) {
    const PathVertex& v = trace[k];
    double* o = out + (size_t)k * 11;
    for (int a = 0; a < 3; ++a) { o[a] = v.o[a]; o[3 + a] = v.d[a]; }
    o[6] = v.t; o[7] = v.obj; o[8] = v.sub; o[9] = v.hit; o[10] = v.front;
  }
  return (int)trace.size();
}

// color.rs:14-27 over a whole accumulation buffer.
void orc_resolve(const double* accum, size_t n_pixels, uint32_t spp, uint8_t* rgb) {
  for (size_t i = 0; i < n_pixels; ++i) color_to_rgb(accum + i * 3, spp, rgb + i * 3);
}

// ---- unit-level entry points for known-answer tests ----
int orc_aabb_hit(const double mn[3], const double mx[3], const double o[3], const double d[3], double tmin, double tmax) {
  Aabb<double> b{V3<double>(mn[0], mn[1], mn[2]), V3<double>(mx[0], mx[1], mx[2])};
  Ray<double> r{V3<double>(o[0], o[1], o[2]), V3<double>(d[0], d[1], d[2]), 0};
  return b.hit(r, tmin, tmax) ? 1 : 0;
}
void orc_sphere_uv(const double p[3], double* u, double* v) { get_sphere_uv<double>(V3<double>(p[0], p[1], p[2]), *u, *v); }
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32(ctr, key, out); }
void orc_philox_rounds(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) { philox4x32(ctr, key, out, rounds); }
int orc_philox_stream_rounds() { return kPhiloxRounds; }
void orc_stream_words(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stage, int n, uint64_t* out) {
  Stream s(seed, pixel, sample, bounce, (Stage)stage);
  for (int i = 0; i < n; ++i) out[i] = s.word((uint32_t)i);
}
void orc_pcg32_words(uint64_t seed, int n, uint32_t* out) {
  Pcg32 g = Pcg32::seed_from_u64(seed);
  for (int i = 0; i < n; ++i) out[i] = g.next_u32();
}
// Single-shape hits: kind 0 sphere(c, r), 1 xy-rect, 2 xz-rect, 3 yz-rect (a0,a1,b0,b1,k), 4 box(min,max)
int orc_shape_hit(int kind, const double* prm, const orc_ray* ray, orc_hit* out) {
  auto m = std::make_shared<Dielectric<double>>(1.5);
  m->id = 0;
  HPtr<double> h;
  switch (kind) {
    case 0: h = std::make_shared<Sphere<double>>(V3<double>(prm[0], prm[1], prm[2]), prm[3], m); break;
    case 1: h = std::make_shared<XYRect<double>>(prm[0], prm[1], prm[2], prm[3], prm[4], m); break;
    case 2: h = std::make_shared<XZRect<double>>(prm[0], prm[1], prm[2], prm[3], prm[4], m); break;
    case 3: h = std::make_shared<YZRect<double>>(prm[0], prm[1], prm[2], prm[3], prm[4], m); break;
    case 4: h = std::make_shared<BoxShape<double>>(V3<double>(prm[0], prm[1], prm[2]), V3<double>(prm[3], prm[4], prm[5]), m); break;
    default: return -1;
  }
  h->label(0, -1);
  Ray<double> r{V3<double>(ray->o[0], ray->o[1], ray->o[2]), V3<double>(ray->d[0], ray->d[1], ray->d[2]), ray->time};
  Stream media(0, 0, 0, 0, STAGE_MEDIA);
  HitCtx ctx{&media};
  HitRecord<double> rec;
  bool hit = h->hit(r, ray->tmin, ray->tmax, rec, ctx);
  std::memset(out, 0, sizeof(*out));
  out->hit = hit;
  if (hit) {
    out->obj = rec.obj; out->sub = rec.sub; out->side = rec.side; out->material = 0; out->front_face = rec.front_face;
    out->t = rec.t; out->u = rec.u; out->v = rec.v;
    for (int k = 0; k < 3; ++k) { out->p[k] = rec.p[k]; out->n[k] = rec.normal[k]; }
  }
  return hit ? 1 : 0;
}
void orc_reflect(const double v[3], const double n[3], double out[3]) {
  auto r = reflect(V3<double>(v[0], v[1], v[2]), V3<double>(n[0], n[1], n[2]));
  for (int k = 0; k < 3; ++k) out[k] = r[k];
}
void orc_refract(const double uv[3], const double n[3], double ratio, double out[3]) {
  auto r = refract(V3<double>(uv[0], uv[1], uv[2]), V3<double>(n[0], n[1], n[2]), ratio);
  for (int k = 0; k < 3; ++k) out[k] = r[k];
}
double orc_schlick(double cosine, double ref_idx) { return Dielectric<double>::reflectance(cosine, ref_idx); }
void orc_color_to_rgb(const double sum[3], uint32_t spp, uint8_t out[3]) { color_to_rgb(sum, spp, out); }
int orc_divide_into_ranges(uint32_t rows, uint32_t ranges, uint32_t* out, int cap) {
  auto r = divide_into_ranges(rows, ranges);
  for (size_t i = 0; i < r.size() && (int)i < cap; ++i) { out[2 * i] = r[i].first; out[2 * i + 1] = r[i].second; }
  return (int)r.size();
}
// Perlin noise / turbulence / texture value of the scene's first NoiseTexture-bearing material (two-perlin / simple-light: material 0).
int orc_material_texture_value(const orc_scene* s, int material, double u, double v, const double p[3], double out[3]) {
  if (material < 0 || material >= (int)s->d->materials.size()) return 1;
  const Material<double>* m = s->d->materials[material].get();
  V3<double> c;
  V3<double> pt(p[0], p[1], p[2]);
  if (auto* l = dynamic_cast<const Lambertian<double>*>(m)) c = l->albedo->value(u, v, pt);
  else if (auto* d = dynamic_cast<const DiffuseLight<double>*>(m)) c = d->emit->value(u, v, pt);
  else if (auto* i = dynamic_cast<const Isotropic<double>*>(m)) c = i->albedo->value(u, v, pt);
  else if (auto* mt = dynamic_cast<const Metal<double>*>(m)) c = mt->albedo;
  else c = V3<double>(1, 1, 1);
  for (int k = 0; k < 3; ++k) out[k] = c[k];
  return 0;
}

}  // extern "C"
