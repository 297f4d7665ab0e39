// Host-side scene construction: restates SceneConfig::get_scene (src/scenes.rs:44-133) and the
// eight scene functions (src/scenes.rs:141-556) directly into the flat arrays of rt_scene_desc,
// builds the BVH with the reference's algorithm (src/bvh.rs:19-76) and the camera with
// Camera::new (src/camera.rs:21-60).  Index-based throughout: there is no pointer tree here.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "host_util.h"
#include "rtamd_host.h"

namespace rth {

thread_local std::string g_err;

namespace {

struct D3 {
  double x, y, z;
};
inline D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 scale(double f, D3 v) { return {v.x * f, v.y * f, v.z * f}; }       // src/vec3.rs:200-206
inline D3 divide(D3 v, double f) { return scale(1.0 / f, v); }                 // src/vec3.rs:224-230
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double length(D3 v) { return std::sqrt(dot(v, v)); }
inline D3 unit(D3 v) { return divide(v, length(v)); }                          // src/vec3.rs:77-79
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Box3 {
  double mn[3], mx[3];
};
inline Box3 surrounding(const Box3& a, const Box3& b) {  // src/aabb.rs:48-61
  Box3 r;
  for (int k = 0; k < 3; ++k) {
    r.mn[k] = std::fmin(a.mn[k], b.mn[k]);
    r.mx[k] = std::fmax(a.mx[k], b.mx[k]);
  }
  return r;
}

}  // namespace

struct SceneData {
  rt_scene_desc desc{};
  std::vector<rt_bvh_node> nodes;
  std::vector<rt_object> objects;
  std::vector<rt_shape> shapes;
  std::vector<rt_material> materials;
  std::vector<rt_texture> textures;
  std::vector<rt_perlin> perlins;
  std::vector<rt_image> images;
  std::vector<std::vector<uint8_t>> image_bytes;
  uint32_t image_width = 400;
  double aspect_ratio = 16.0 / 9.0;
  uint32_t spp = 100, depth = 50;

  // ---- textures / materials (src/texture.rs, src/material.rs constructors) ----
  int solid(double r, double g, double b) {
    rt_texture t{};
    t.tag = RT_TEX_SOLID; t.even = t.odd = t.perlin = t.image = -1;
    t.color[0] = r; t.color[1] = g; t.color[2] = b;
    textures.push_back(t);
    return (int)textures.size() - 1;
  }
  int checker(const double even[3], const double odd[3]) {  // src/texture.rs:47-51
    int e = solid(even[0], even[1], even[2]);
    int o = solid(odd[0], odd[1], odd[2]);
    rt_texture t{};
    t.tag = RT_TEX_CHECKER; t.even = e; t.odd = o; t.perlin = t.image = -1;
    textures.push_back(t);
    return (int)textures.size() - 1;
  }
  int noise(double scale_, Pcg32& aux) {  // src/texture.rs:71-76, src/perlin.rs:14-30,74-93
    rt_perlin p{};
    for (int i = 0; i < 256; ++i) {
      D3 v;
      v.x = aux.range_f64(-1.0, 1.0); v.y = aux.range_f64(-1.0, 1.0); v.z = aux.range_f64(-1.0, 1.0);
      D3 u = unit(v);
      p.ran_vec[i][0] = u.x; p.ran_vec[i][1] = u.y; p.ran_vec[i][2] = u.z;
    }
    int32_t* perms[3] = {p.perm_x, p.perm_y, p.perm_z};
    for (int a = 0; a < 3; ++a) {
      for (int i = 0; i < 256; ++i) perms[a][i] = i;
      for (int i = 255; i >= 1; --i) {  // gen_range(0..i): i itself is never drawn (src/perlin.rs:89-92)
        int target = (int)aux.below_u64((uint64_t)i);
        std::swap(perms[a][i], perms[a][target]);
      }
    }
    perlins.push_back(p);
    rt_texture t{};
    t.tag = RT_TEX_NOISE; t.even = t.odd = t.image = -1; t.perlin = (int)perlins.size() - 1; t.scale = scale_;
    textures.push_back(t);
    return (int)textures.size() - 1;
  }
  int image_texture(const rth_scene_opts& o) {  // src/texture.rs:97-125 (the JPEG is pre-decoded)
    rt_texture t{};
    t.tag = RT_TEX_IMAGE; t.even = t.odd = t.perlin = -1; t.image = -1;
    std::vector<uint8_t> bytes;
    int w = 0, h = 0;
    if (o.earth_rgb && o.earth_w > 0 && o.earth_h > 0) {
      w = o.earth_w; h = o.earth_h;
      bytes.assign(o.earth_rgb, o.earth_rgb + (size_t)3 * w * h);
    } else {
      const char* path = o.earth_path ? o.earth_path : "earthmap.ppm";
      uint8_t* px = nullptr;
      int32_t pw = 0, ph = 0;
      if (rth_load_ppm(path, &px, &pw, &ph) == 0) {
        w = pw; h = ph;
        bytes.assign(px, px + (size_t)3 * w * h);
        rth_free(px);
      } else {
        std::fprintf(stderr, "ERROR: Could not load texture file %s\n", path);  // src/texture.rs:108
      }
    }
    if (!bytes.empty()) {
      image_bytes.push_back(std::move(bytes));
      rt_image im{};
      im.width = w; im.height = h; im.rgb = nullptr;  // pointers fixed up in finalize()
      images.push_back(im);
      t.image = (int)images.size() - 1;
    }
    textures.push_back(t);
    return (int)textures.size() - 1;
  }
  int material(int tag, int tex, double ar = 0, double ag = 0, double ab = 0, double fuzz = 0, double ir = 0) {
    rt_material m{};
    m.tag = tag; m.texture = tex; m.albedo[0] = ar; m.albedo[1] = ag; m.albedo[2] = ab; m.fuzz = fuzz; m.ir = ir;
    materials.push_back(m);
    return (int)materials.size() - 1;
  }
  int lambert_tex(int tex) { return material(RT_MAT_LAMBERTIAN, tex); }
  int lambert(double r, double g, double b) { return lambert_tex(solid(r, g, b)); }       // src/material.rs:36-39
  int metal(double r, double g, double b, double f) { return material(RT_MAT_METAL, -1, r, g, b, f < 1.0 ? f : 1.0); }  // :66-71
  int glass(double ir) { return material(RT_MAT_DIELECTRIC, -1, 0, 0, 0, 0, ir); }
  int light(double r, double g, double b) { return material(RT_MAT_DIFFUSE_LIGHT, solid(r, g, b)); }  // :153-156
  int isotropic(double r, double g, double b) { return material(RT_MAT_ISOTROPIC, solid(r, g, b)); }  // :180-183

  // ---- shapes / objects ----
  int shape(int tag, int mat, std::initializer_list<double> p) {
    rt_shape s{};
    s.tag = tag; s.material = mat;
    int i = 0;
    for (double v : p) s.p[i++] = v;
    shapes.push_back(s);
    return (int)shapes.size() - 1;
  }
  int object_of_shape(int shape_idx) {
    rt_object o{};
    o.flags = 0; o.geom = shape_idx; o.medium_material = -1; o.obj_id = -1; o.sub_id = -1;
    o.cos_theta = 1.0;
    objects.push_back(o);
    return (int)objects.size() - 1;
  }
  int sphere(D3 c, double r, int mat) { return object_of_shape(shape(RT_SHAPE_SPHERE, mat, {c.x, c.y, c.z, r})); }
  int moving_sphere(D3 c0, D3 c1, double t0, double t1, double r, int mat) {
    return object_of_shape(shape(RT_SHAPE_MOVING_SPHERE, mat, {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, r, t0, t1}));
  }
  int rect(int tag, double a0, double a1, double b0, double b1, double k, int mat) { return object_of_shape(shape(tag, mat, {a0, a1, b0, b1, k})); }
  int box(D3 lo, D3 hi, int mat) { return object_of_shape(shape(RT_SHAPE_BOX, mat, {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z})); }
  void rotate_y(int obj, double angle_degrees) {  // RotateY::new, src/hittable.rs:162-166
    double radians = angle_degrees * (3.14159265358979323846264338327950288 / 180.0);
    objects[obj].flags |= RT_OBJ_ROTATE_Y;
    objects[obj].sin_theta = std::sin(radians);
    objects[obj].cos_theta = std::cos(radians);
  }
  void translate(int obj, D3 off) {  // Translate::new, src/hittable.rs:118-123
    objects[obj].flags |= RT_OBJ_TRANSLATE;
    objects[obj].offset[0] = off.x; objects[obj].offset[1] = off.y; objects[obj].offset[2] = off.z;
  }
  void medium(int obj, double density, int phase_material) {  // ConstantMedium::new_from_color, src/constant_medium.rs:27-33
    objects[obj].flags |= RT_OBJ_MEDIUM;
    objects[obj].medium_material = phase_material;
    objects[obj].neg_inv_density = -1.0 / density;
  }
  int object_of_bvh(int root) {
    rt_object o{};
    o.flags = RT_OBJ_SUB_BVH; o.geom = root; o.medium_material = -1; o.obj_id = -1; o.sub_id = -1; o.cos_theta = 1.0;
    objects.push_back(o);
    return (int)objects.size() - 1;
  }

  // ---- bounding boxes (bounding_box() of every Hittable) ----
  D3 moving_center(const rt_shape& s, double time) const {  // src/moving_sphere.rs:37-40
    D3 c0{s.p[0], s.p[1], s.p[2]}, c1{s.p[3], s.p[4], s.p[5]};
    return add(c0, scale((time - s.p[7]) / (s.p[8] - s.p[7]), sub(c1, c0)));
  }
  Box3 shape_box(const rt_shape& s, double t0, double t1) const {
    Box3 b{};
    const double pad = 0.0001;
    switch (s.tag) {
      case RT_SHAPE_SPHERE:  // src/sphere.rs:75-81
        for (int k = 0; k < 3; ++k) { b.mn[k] = s.p[k] - s.p[3]; b.mx[k] = s.p[k] + s.p[3]; }
        break;
      case RT_SHAPE_MOVING_SPHERE: {  // src/moving_sphere.rs:74-85
        D3 a = moving_center(s, t0), c = moving_center(s, t1);
        double r = s.p[6];
        Box3 b0{{a.x - r, a.y - r, a.z - r}, {a.x + r, a.y + r, a.z + r}};
        Box3 b1{{c.x - r, c.y - r, c.z - r}, {c.x + r, c.y + r, c.z + r}};
        b = surrounding(b0, b1);
        break;
      }
      case RT_SHAPE_RECT_XY:  // src/aarect.rs:76-86
        b = Box3{{s.p[0], s.p[2], s.p[4] - pad}, {s.p[1], s.p[3], s.p[4] + pad}};
        break;
      case RT_SHAPE_RECT_XZ:  // src/aarect.rs:111-121
        b = Box3{{s.p[0], s.p[4] - pad, s.p[2]}, {s.p[1], s.p[4] + pad, s.p[3]}};
        break;
      case RT_SHAPE_RECT_YZ:  // src/aarect.rs:146-156
        b = Box3{{s.p[4] - pad, s.p[0], s.p[2]}, {s.p[4] + pad, s.p[1], s.p[3]}};
        break;
      case RT_SHAPE_BOX:  // src/box_struct.rs:82-84
        b = Box3{{s.p[0], s.p[1], s.p[2]}, {s.p[3], s.p[4], s.p[5]}};
        break;
    }
    return b;
  }
  Box3 object_box(int obj, double t0, double t1) const {
    const rt_object& o = objects[obj];
    auto geom_box = [&](double a, double c) {
      if (o.flags & RT_OBJ_SUB_BVH) {  // src/bvh.rs:121-123
        Box3 b;
        for (int k = 0; k < 3; ++k) { b.mn[k] = nodes[o.geom].bmin[k]; b.mx[k] = nodes[o.geom].bmax[k]; }
        return b;
      }
      return shape_box(shapes[o.geom], a, c);
    };
    Box3 b;
    if (o.flags & RT_OBJ_ROTATE_Y) {  // src/hittable.rs:166-197: extents of the 8 rotated corners of the child's box at (0,1)
      Box3 c = geom_box(0.0, 1.0);
      const double inf = std::numeric_limits<double>::infinity();
      for (int k = 0; k < 3; ++k) { b.mn[k] = inf; b.mx[k] = -inf; }
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          for (int k = 0; k < 2; ++k) {
            double x = (double)i * c.mx[0] + (double)(1 - i) * c.mn[0];
            double y = (double)j * c.mx[1] + (double)(1 - j) * c.mn[1];
            double z = (double)k * c.mx[2] + (double)(1 - k) * c.mn[2];
            double nx = o.cos_theta * x + o.sin_theta * z;
            double nz = -o.sin_theta * x + o.cos_theta * z;
            double t[3] = {nx, y, nz};
            for (int a = 0; a < 3; ++a) { b.mn[a] = std::fmin(b.mn[a], t[a]); b.mx[a] = std::fmax(b.mx[a], t[a]); }
          }
    } else {
      b = geom_box(t0, t1);
    }
    if (o.flags & RT_OBJ_TRANSLATE)  // src/hittable.rs:127-138
      for (int k = 0; k < 3; ++k) { b.mn[k] += o.offset[k]; b.mx[k] += o.offset[k]; }
    return b;  // ConstantMedium forwards to its boundary (src/constant_medium.rs:37-39)
  }

  // ---- BVHNode::new (src/bvh.rs:19-76).  Returns the node index. ----
  static bool total_less(double a, double b) {  // f64::total_cmp(..).is_lt() for the non-NaN values that occur
    if (a == b) return std::signbit(a) && !std::signbit(b);
    return a < b;
  }
  // The tree the reference would build, built faster.  The recursion of the reference draws a node's axis, sorts the node's
  // objects, then builds the left half completely before the right one: a node's number (and the position of its axis draw in
  // the random sequence) is its preorder index, and how many nodes a subtree has depends only on how many objects it holds.
  // So all axes are drawn up front, the sort keys (bounding_box(0,0).min per axis, src/bvh.rs:83-94) are computed once per object
  // instead of once per comparison, the halves are sorted in place (the copies the reference makes are the sorted halves), and
  // subtrees are handed to threads: every node still gets the axis, the order and the index it has in the reference.
  static size_t bvh_node_count(size_t n) { return n <= 2 ? 1 : 1 + bvh_node_count(n / 2) + bvh_node_count(n - n / 2); }
  struct BvhJob {
    std::vector<int> order;          // object indices; each node sorts its own span
    std::vector<uint8_t> axis;       // per node of this tree, preorder
    std::vector<double> key[3];      // per object index
    double time0, time1;
    int base;                        // index of the tree's first node in `nodes`
  };
  void build_span(BvhJob& job, size_t lo, size_t hi, int me, int spawn_depth) {
    const int axis = job.axis[(size_t)(me - job.base)];
    const std::vector<double>& key = job.key[axis];
    int* o = job.order.data();
    const size_t n = hi - lo;
    int left, right;
    if (n == 1) {
      left = right = ~o[lo];
    } else if (n == 2) {
      if (total_less(key[(size_t)o[lo]], key[(size_t)o[lo + 1]])) { left = ~o[lo]; right = ~o[lo + 1]; }
      else { left = ~o[lo + 1]; right = ~o[lo]; }
    } else {
      std::stable_sort(o + lo, o + hi, [&](int a, int b) { return total_less(key[(size_t)a], key[(size_t)b]); });
      const size_t mid = lo + n / 2;
      left = me + 1;
      right = me + 1 + (int)bvh_node_count(mid - lo);
      if (spawn_depth > 0 && n >= 4096) {
        std::thread other([&, right] { build_span(job, mid, hi, right, spawn_depth - 1); });
        build_span(job, lo, mid, left, spawn_depth - 1);
        other.join();
      } else {
        build_span(job, lo, mid, left, 0);
        build_span(job, mid, hi, right, 0);
      }
    }
    auto child_box = [&](int c) {
      if (c >= 0) {
        Box3 b;
        for (int k = 0; k < 3; ++k) { b.mn[k] = nodes[(size_t)c].bmin[k]; b.mx[k] = nodes[(size_t)c].bmax[k]; }
        return b;
      }
      return object_box(~c, job.time0, job.time1);
    };
    Box3 b = surrounding(child_box(left), child_box(right));
    rt_bvh_node& nd = nodes[(size_t)me];
    for (int k = 0; k < 3; ++k) { nd.bmin[k] = b.mn[k]; nd.bmax[k] = b.mx[k]; }
    nd.left = left; nd.right = right;
  }
  int build_bvh(std::vector<int> objs, double time0, double time1, Pcg32& axis_rng) {
    BvhJob job;
    job.time0 = time0; job.time1 = time1;
    job.base = (int)nodes.size();
    const size_t n_nodes = bvh_node_count(objs.size());
    job.axis.resize(n_nodes);
    for (size_t i = 0; i < n_nodes; ++i) job.axis[i] = (uint8_t)axis_rng.range_u32_inclusive(0, 2);
    for (int k = 0; k < 3; ++k) job.key[k].assign(objects.size(), 0.0);
    for (int o : objs) {
      const Box3 b = object_box(o, 0.0, 0.0);
      for (int k = 0; k < 3; ++k) job.key[k][(size_t)o] = b.mn[k];
    }
    job.order = std::move(objs);
    nodes.resize(nodes.size() + n_nodes);  // no reallocation while threads write their nodes
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)std::max(1, atoi(e)));
    int depth = 0;
    while ((1u << (depth + 1)) <= std::max(1u, hw) && depth < 6) ++depth;  // 2^depth threads at the deepest split
    build_span(job, 0, job.order.size(), job.base, depth);
    return job.base;
  }

  void set_camera(D3 look_from, D3 look_at, D3 v_up, double vfov_degrees, double aspect, double aperture, double focus_dist,
                  double t_open, double t_close) {  // src/camera.rs:21-60
    double theta = vfov_degrees * (3.14159265358979323846264338327950288 / 180.0);
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h;
    double viewport_width = aspect * viewport_height;
    D3 w = unit(sub(look_from, look_at));
    D3 u = unit(cross(v_up, w));
    D3 v = cross(w, u);
    D3 origin = look_from;
    D3 horizontal = scale(focus_dist * viewport_width, u);
    D3 vertical = scale(focus_dist * viewport_height, v);
    D3 llc = sub(sub(sub(origin, divide(horizontal, 2.0)), divide(vertical, 2.0)), scale(focus_dist, w));
    rt_camera& c = desc.camera;
    auto put = [](double* d, D3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    put(c.origin, origin); put(c.lower_left_corner, llc); put(c.horizontal, horizontal); put(c.vertical, vertical);
    put(c.u, u); put(c.v, v); put(c.w, w);
    c.lens_radius = aperture / 2.0;
    c.shutter_open_time = t_open;
    c.shutter_close_time = t_close;
  }

  void finalize() {
    for (size_t i = 0; i < images.size(); ++i) images[i].rgb = image_bytes[i].data();
    desc.n_nodes = (int)nodes.size(); desc.n_objects = (int)objects.size(); desc.n_shapes = (int)shapes.size();
    desc.n_materials = (int)materials.size(); desc.n_textures = (int)textures.size();
    desc.n_perlins = (int)perlins.size(); desc.n_images = (int)images.size();
    desc.nodes = nodes.data(); desc.objects = objects.data(); desc.shapes = shapes.data();
    desc.materials = materials.data(); desc.textures = textures.data();
    desc.perlins = perlins.empty() ? nullptr : perlins.data();
    desc.images = images.empty() ? nullptr : images.data();
  }
};

namespace {

struct WorldList {
  std::vector<int> objs;
  void add(int o) { objs.push_back(o); }
};

void random_scene(SceneData& s, WorldList& world, Pcg32& aux) {  // src/scenes.rs:141-247
  const double even[3] = {0.2, 0.3, 0.1}, odd[3] = {0.9, 0.9, 0.9};
  int ground = s.lambert_tex(s.checker(even, odd));
  world.add(s.sphere({0.0, -1000.0, 0.0}, 1000.0, ground));
  Pcg32 rng = Pcg32::seed_from_u64(2);
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      double choose_mat = rng.f64();
      D3 center{(double)a + 0.9 * rng.f64(), 0.2, 0.0};
      center.z = (double)b + 0.9 * rng.f64();
      if (length(sub(center, D3{4.0, 0.2, 0.0})) > 0.9) {
        if (choose_mat < 0.8) {
          double r0 = aux.f64(), g0 = aux.f64(), b0 = aux.f64();  // Color::random() * Color::random(), src/scenes.rs:172
          double r1 = aux.f64(), g1 = aux.f64(), b1 = aux.f64();
          int m = s.lambert(r0 * r1, g0 * g1, b0 * b1);
          D3 center2 = add(center, D3{0.0, rng.range_f64(0.0, 0.5), 0.0});
          world.add(s.moving_sphere(center, center2, 0.0, 1.0, 0.2, m));
        } else if (choose_mat < 0.95) {
          double r = aux.range_f64(0.5, 1.0), g = aux.range_f64(0.5, 1.0), bl = aux.range_f64(0.5, 1.0);  // src/scenes.rs:185
          double fuzz = rng.range_f64(0.0, 0.5);
          world.add(s.sphere(center, 0.2, s.metal(r, g, bl, fuzz)));
        } else {
          world.add(s.sphere(center, 0.2, s.glass(1.5)));
        }
      }
    }
    // The three large spheres sit inside the `for a` loop (src/scenes.rs:197-217): 22 copies of each.
    world.add(s.sphere({0.0, 1.0, 0.0}, 1.0, s.glass(1.5)));
    world.add(s.sphere({-4.0, 1.0, 0.0}, 1.0, s.lambert(0.4, 0.2, 0.1)));
    world.add(s.sphere({4.0, 1.0, 0.0}, 1.0, s.metal(0.7, 0.6, 0.5, 0.0)));
  }
  int material_center = s.lambert(0.1, 0.2, 0.5);
  int material_left = s.glass(1.5);
  int material_right = s.metal(0.8, 0.6, 0.2, 0.0);
  world.add(s.sphere({0.0, 0.0, -1.0}, 0.5, material_center));
  world.add(s.sphere({-1.0, 0.0, -1.0}, 0.5, material_left));
  world.add(s.sphere({-1.0, 0.0, -1.0}, -0.45, material_left));
  world.add(s.sphere({1.0, 0.0, -1.0}, 0.5, material_right));
}

void two_spheres(SceneData& s, WorldList& world) {  // src/scenes.rs:249-271
  const double even[3] = {0.2, 0.3, 0.1}, odd[3] = {0.9, 0.9, 0.9};
  int m = s.lambert_tex(s.checker(even, odd));
  world.add(s.sphere({0.0, -10.0, 0.0}, 10.0, m));
  world.add(s.sphere({0.0, 10.0, 0.0}, 10.0, m));
}

void two_perlin_spheres(SceneData& s, WorldList& world, Pcg32& aux) {  // src/scenes.rs:273-292
  int m = s.lambert_tex(s.noise(4.0, aux));
  world.add(s.sphere({0.0, -1000.0, 0.0}, 1000.0, m));
  world.add(s.sphere({0.0, 2.0, 0.0}, 2.0, m));
}

void earth(SceneData& s, WorldList& world, const rth_scene_opts& o) {  // src/scenes.rs:294-308
  int m = s.lambert_tex(s.image_texture(o));
  world.add(s.sphere({0.0, 0.0, 0.0}, 2.0, m));
}

void simple_light(SceneData& s, WorldList& world, Pcg32& aux) {  // src/scenes.rs:310-339
  int m = s.lambert_tex(s.noise(4.0, aux));
  world.add(s.sphere({0.0, -1000.0, 0.0}, 1000.0, m));
  world.add(s.sphere({0.0, 2.0, 0.0}, 2.0, m));
  int dl = s.light(4.0, 4.0, 4.0);
  world.add(s.rect(RT_SHAPE_RECT_XY, 3.0, 5.0, 1.0, 3.0, -2.0, dl));
  world.add(s.sphere({0.0, 10.0, 0.0}, 2.0, dl));
}

int rotated_box(SceneData& s, D3 hi, double angle, D3 off, int mat) {  // src/scenes.rs:358-365
  int b = s.box({0.0, 0.0, 0.0}, hi, mat);
  s.rotate_y(b, angle);
  s.translate(b, off);
  return b;
}

void cornell_walls(SceneData& s, WorldList& world, int red, int white, int green, int lt, const double light_rect[4], bool smoke_order) {
  world.add(s.rect(RT_SHAPE_RECT_YZ, 0.0, 555.0, 0.0, 555.0, 555.0, green));
  world.add(s.rect(RT_SHAPE_RECT_YZ, 0.0, 555.0, 0.0, 555.0, 0.0, red));
  world.add(s.rect(RT_SHAPE_RECT_XZ, light_rect[0], light_rect[1], light_rect[2], light_rect[3], 554.0, lt));
  // floor/ceiling order differs between the two scenes (src/scenes.rs:354-355 vs 396-397)
  world.add(s.rect(RT_SHAPE_RECT_XZ, 0.0, 555.0, 0.0, 555.0, smoke_order ? 555.0 : 0.0, white));
  world.add(s.rect(RT_SHAPE_RECT_XZ, 0.0, 555.0, 0.0, 555.0, smoke_order ? 0.0 : 555.0, white));
  world.add(s.rect(RT_SHAPE_RECT_XY, 0.0, 555.0, 0.0, 555.0, 555.0, white));
}

void cornell_box(SceneData& s, WorldList& world) {  // src/scenes.rs:341-381
  int red = s.lambert(0.65, 0.05, 0.05);
  int white = s.lambert(0.73, 0.73, 0.73);
  int green = s.lambert(0.12, 0.45, 0.15);
  int lt = s.light(15.0, 15.0, 15.0);
  const double lr[4] = {213.0, 343.0, 227.0, 332.0};
  cornell_walls(s, world, red, white, green, lt, lr, false);
  world.add(rotated_box(s, {165.0, 330.0, 165.0}, 15.0, {265.0, 0.0, 295.0}, white));
  world.add(rotated_box(s, {168.0, 165.0, 165.0}, -18.0, {130.0, 0.0, 65.0}, white));
}

void cornell_smoke(SceneData& s, WorldList& world) {  // src/scenes.rs:383-430
  int red = s.lambert(0.65, 0.05, 0.05);
  int white = s.lambert(0.73, 0.73, 0.73);
  int green = s.lambert(0.12, 0.45, 0.15);
  int lt = s.light(7.0, 7.0, 7.0);
  const double lr[4] = {113.0, 443.0, 127.0, 432.0};
  cornell_walls(s, world, red, white, green, lt, lr, true);
  int box1 = rotated_box(s, {165.0, 330.0, 165.0}, 15.0, {265.0, 0.0, 295.0}, white);
  int box2 = rotated_box(s, {168.0, 165.0, 165.0}, -18.0, {130.0, 0.0, 65.0}, white);
  s.medium(box1, 0.01, s.isotropic(0.0, 0.0, 0.0));
  s.medium(box2, 0.01, s.isotropic(1.0, 1.0, 1.0));
  world.add(box1);
  world.add(box2);
}

void final_scene(SceneData& s, WorldList& world, Pcg32& aux, Pcg32& bvh_rng, const rth_scene_opts& o) {  // src/scenes.rs:432-556
  std::vector<int> boxes1;
  int ground = s.lambert(0.48, 0.83, 0.53);
  Pcg32 rng = Pcg32::seed_from_u64(3);
  const int boxes_per_side = 20;
  for (int i = 0; i < boxes_per_side; ++i)
    for (int j = 0; j < boxes_per_side; ++j) {
      double w = 100.0;
      double x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
      double x1 = x0 + w, y1 = rng.range_f64(1.0, 101.0), z1 = z0 + w;
      int b = s.box({x0, y0, z0}, {x1, y1, z1}, ground);
      s.objects[b].sub_id = (int)boxes1.size();
      boxes1.push_back(b);
    }
  world.add(s.object_of_bvh(s.build_bvh(boxes1, 0.0, 1.0, bvh_rng)));
  world.add(s.rect(RT_SHAPE_RECT_XZ, 123.0, 423.0, 147.0, 412.0, 554.0, s.light(7.0, 7.0, 7.0)));
  D3 center1{400.0, 400.0, 400.0};
  D3 center2 = add(center1, D3{30.0, 0.0, 0.0});
  world.add(s.moving_sphere(center1, center2, 0.0, 1.0, 50.0, s.lambert(0.7, 0.3, 0.1)));
  int dielectric = s.glass(1.5);
  world.add(s.sphere({260.0, 150.0, 45.0}, 50.0, dielectric));
  world.add(s.sphere({0.0, 150.0, 145.0}, 50.0, s.metal(0.8, 0.8, 0.9, 1.0)));
  // boundary sphere is both an object and the boundary of a medium (src/scenes.rs:498-508): two objects over equal shapes
  world.add(s.sphere({360.0, 150.0, 145.0}, 70.0, dielectric));
  int fog1 = s.sphere({360.0, 150.0, 145.0}, 70.0, dielectric);
  s.medium(fog1, 0.2, s.isotropic(0.2, 0.4, 0.9));
  world.add(fog1);
  int fog2 = s.sphere({0.0, 0.0, 0.0}, 5000.0, dielectric);
  s.medium(fog2, 0.0001, s.isotropic(1.0, 1.0, 1.0));
  world.add(fog2);
  world.add(s.sphere({400.0, 200.0, 400.0}, 100.0, s.lambert_tex(s.image_texture(o))));
  world.add(s.sphere({220.0, 280.0, 300.0}, 80.0, s.lambert_tex(s.noise(0.1, aux))));
  std::vector<int> boxes2;
  int white = s.lambert(0.73, 0.73, 0.73);
  for (int j = 0; j < 1000; ++j) {
    double x = aux.range_f64(0.0, 165.0), y = aux.range_f64(0.0, 165.0), z = aux.range_f64(0.0, 165.0);  // Point3::random_range
    int sp = s.sphere({x, y, z}, 10.0, white);
    s.objects[sp].sub_id = j;
    boxes2.push_back(sp);
  }
  int cluster = s.object_of_bvh(s.build_bvh(boxes2, 0.0, 1.0, bvh_rng));
  s.rotate_y(cluster, 15.0);
  s.translate(cluster, {-100.0, 270.0, 395.0});
  world.add(cluster);
}

void mega_spheres(SceneData& s, WorldList& world, Pcg32& aux, int grid) {  // not in the reference
  const double even[3] = {0.2, 0.3, 0.1}, odd[3] = {0.9, 0.9, 0.9};
  world.add(s.sphere({0.0, -100000.0, 0.0}, 100000.0, s.lambert_tex(s.checker(even, odd))));
  Pcg32 rng = Pcg32::seed_from_u64(5);
  int g = s.glass(1.5);
  for (int a = -grid; a < grid; ++a)
    for (int b = -grid; b < grid; ++b) {
      double choose_mat = rng.f64();
      D3 center{(double)a + 0.9 * rng.f64(), 0.2, 0.0};
      center.z = (double)b + 0.9 * rng.f64();
      if (choose_mat < 0.8) {
        double r0 = aux.f64(), g0 = aux.f64(), b0 = aux.f64();
        double r1 = aux.f64(), g1 = aux.f64(), b1 = aux.f64();
        double dy = rng.range_f64(0.0, 0.5);
        world.add(s.moving_sphere(center, add(center, D3{0.0, dy, 0.0}), 0.0, 1.0, 0.2, s.lambert(r0 * r1, g0 * g1, b0 * b1)));
      } else if (choose_mat < 0.95) {
        double r = aux.range_f64(0.5, 1.0), gg = aux.range_f64(0.5, 1.0), bl = aux.range_f64(0.5, 1.0);
        double fuzz = rng.range_f64(0.0, 0.5);
        world.add(s.sphere(center, 0.2, s.metal(r, gg, bl, fuzz)));
      } else {
        world.add(s.sphere(center, 0.2, g));
      }
    }
}

}  // namespace

SceneData* build_scene(const rth_scene_opts& o) {  // src/scenes.rs:44-133
  auto* sp = new SceneData;
  SceneData& s = *sp;
  Pcg32 aux = Pcg32::seed_from_u64(o.aux_seed);
  Pcg32 bvh_rng = Pcg32::seed_from_u64(o.bvh_seed);
  WorldList world;
  D3 v_up{0.0, 1.0, 0.0};
  double v_fov = 20.0, aperture = 0.0, time0 = 0.0, time1 = 1.0, focus_dist = 10.0;
  D3 look_from{13.0, 2.0, 3.0}, look_at{0.0, 0.0, 0.0};
  rt_background& bg = s.desc.background;
  bg.kind = RT_BG_SOLID;
  auto solid_bg = [&](double r, double g, double b) { bg.kind = RT_BG_SOLID; bg.c0[0] = r; bg.c0[1] = g; bg.c0[2] = b; };
  auto gradient_bg = [&]() {  // Gradient(Color(0.5,0.7,1.0), Color(1,1,1)), src/scenes.rs:63-64
    bg.kind = RT_BG_GRADIENT;
    bg.c0[0] = 0.5; bg.c0[1] = 0.7; bg.c0[2] = 1.0;
    bg.c1[0] = 1.0; bg.c1[1] = 1.0; bg.c1[2] = 1.0;
  };
  switch (o.scene) {
    case RTH_SCENE_RANDOM: random_scene(s, world, aux); gradient_bg(); aperture = 0.1; break;
    case RTH_SCENE_TWO_SPHERES: two_spheres(s, world); solid_bg(0.70, 0.80, 1.00); break;
    case RTH_SCENE_TWO_PERLIN_SPHERES: two_perlin_spheres(s, world, aux); solid_bg(0.70, 0.80, 1.00); break;
    case RTH_SCENE_EARTH: earth(s, world, o); solid_bg(0.70, 0.80, 1.00); break;
    case RTH_SCENE_SIMPLE_LIGHT:
      simple_light(s, world, aux);
      look_from = {26.0, 3.0, 6.0}; look_at = {0.0, 2.0, 0.0};
      s.spp = 400;
      break;
    case RTH_SCENE_CORNELL_BOX:
    case RTH_SCENE_CORNELL_SMOKE:
      if (o.scene == RTH_SCENE_CORNELL_BOX) cornell_box(s, world); else cornell_smoke(s, world);
      look_from = {278.0, 278.0, -800.0}; look_at = {278.0, 278.0, 0.0};
      v_fov = 40.0; s.aspect_ratio = 1.0; s.image_width = 600; s.spp = 200;
      break;
    case RTH_SCENE_FINAL_SCENE:
      final_scene(s, world, aux, bvh_rng, o);
      look_from = {478.0, 278.0, -600.0}; look_at = {278.0, 278.0, 0.0};
      v_fov = 40.0; s.aspect_ratio = 1.0; s.image_width = 800; s.spp = 10000;
      break;
    case RTH_SCENE_MEGA_SPHERES:
      mega_spheres(s, world, aux, o.mega_grid > 0 ? o.mega_grid : 500);
      gradient_bg();
      look_from = {13.0, 6.0, 3.0}; v_fov = 40.0; aperture = 0.1;
      if (o.mega_view == 1) {  // from above: tan(20 deg) * 2.8 = 1.02 grids of half height, the field's 2 * grid square inside a 16:9 frame
        const double h = 2.8 * (double)(o.mega_grid > 0 ? o.mega_grid : 500);
        look_from = {0.0, h, 0.0}; v_up = {0.0, 0.0, -1.0}; focus_dist = h;
      }
      s.image_width = 3840; s.spp = 1024;
      break;
    default:
      g_err = "unknown scene id";
      delete sp;
      return nullptr;
  }
  if (o.width > 0) s.image_width = (uint32_t)o.width;
  if (o.aspect > 0) s.aspect_ratio = o.aspect;
  if (o.spp > 0) s.spp = (uint32_t)o.spp;
  if (o.depth > 0) s.depth = (uint32_t)o.depth;
  s.set_camera(look_from, look_at, v_up, v_fov, s.aspect_ratio, aperture, focus_dist, time0, time1);
  for (size_t i = 0; i < world.objs.size(); ++i) s.objects[world.objs[i]].obj_id = (int)i;
  // nested lists inherit the identity of the world entry that owns them
  for (size_t i = 0; i < world.objs.size(); ++i) {
    const rt_object& top = s.objects[world.objs[i]];
    if (!(top.flags & RT_OBJ_SUB_BVH)) continue;
    std::vector<int> stack{top.geom};
    while (!stack.empty()) {
      int n = stack.back();
      stack.pop_back();
      for (int c : {s.nodes[n].left, s.nodes[n].right}) {
        if (c >= 0) stack.push_back(c);
        else s.objects[~c].obj_id = (int)i;
      }
    }
  }
  // BVHNode::new(&scene.world.objects, 0.0, 1.0) — src/render.rs:211
  s.desc.root_node = s.build_bvh(world.objs, 0.0, 1.0, bvh_rng);
  s.finalize();
  return sp;
}

}  // namespace rth

struct rth_scene {
  rth::SceneData* d;
};

extern "C" {

rth_scene* rth_scene_create(const rth_scene_opts* opts) {
  if (!opts) { rth::g_err = "opts is NULL"; return nullptr; }
  rth::SceneData* d = rth::build_scene(*opts);
  if (!d) return nullptr;
  return new rth_scene{d};
}
void rth_scene_destroy(rth_scene* s) {
  if (!s) return;
  delete s->d;
  delete s;
}
const rt_scene_desc* rth_scene_desc(const rth_scene* s) { return s ? &s->d->desc : nullptr; }
void rth_scene_config(const rth_scene* s, uint32_t* width, uint32_t* height, uint32_t* spp, uint32_t* depth) {
  if (width) *width = s->d->image_width;
  if (height) *height = (uint32_t)((double)s->d->image_width / s->d->aspect_ratio);  // src/scenes.rs:135-138
  if (spp) *spp = s->d->spp;
  if (depth) *depth = s->d->depth;
}
static const char* const kSceneNames[] = {"random", "two-spheres", "two-perlin-spheres", "earth", "simple-light",
                                          "cornell-box", "cornell-smoke", "final-scene", "mega-spheres"};
int rth_scene_from_name(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < 9; ++i)
    if (std::strcmp(name, kSceneNames[i]) == 0) return i;
  return -1;
}
const char* rth_scene_name(int scene) { return (scene >= 0 && scene < 9) ? kSceneNames[scene] : nullptr; }
const char* rth_last_error(void) { return rth::g_err.c_str(); }

}  // extern "C"
