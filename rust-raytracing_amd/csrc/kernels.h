// Kernels of the wavefront path tracer (one lane per ray, 64-wide wavefronts, persistent grid-stride
// workgroups).  Replaces the worker closure of render_with_threadpool (src/render.rs:224-243) and the
// ray_color recursion (src/render.rs:149-177): the recursion `E + A * ray_color(scattered, depth - 1)` is
// unrolled into per-bounce ray queues carrying the running product of attenuations (`throughput`), and
// every term `throughput * E` is added to the pixel's accumulation slot as soon as it is known.
#pragma once
#include "device_shade.h"

namespace rtd {

template <class R> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

RT_DEV float bits_to_real(uint32_t lo, uint32_t, float) { return __uint_as_float(lo); }
RT_DEV double bits_to_real(uint32_t lo, uint32_t hi, double) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); }
RT_DEV uint32_t real_lo_bits(float x) { return __float_as_uint(x); }
RT_DEV uint32_t real_lo_bits(double x) { return (uint32_t)(uint64_t)__double_as_longlong(x); }

// Ray queue, structure of arrays of 4-vectors: a = (origin, time), b = (direction, pixel bits), c = (throughput, sample bits).
template <class R> struct PathQueue {
  using R4 = typename Vec4<R>::type;
  R4* a;
  R4* b;
  R4* c;
  uint32_t* org;  // Origin bits of every ray (read and written by the single-precision kernels only)
};

template <class R> struct PathState {
  Ray<R> ray;
  V3<R> thr;
  uint32_t pixel, sample;
  Origin org;
};

template <class R> RT_DEV PathState<R> load_path(const PathQueue<R>& q, uint32_t i) {
  typename Vec4<R>::type a = q.a[i], b = q.b[i], c = q.c[i];
  PathState<R> s;
  s.ray.o = V3<R>(a.x, a.y, a.z);
  s.ray.time = a.w;
  s.ray.d = V3<R>(b.x, b.y, b.z);
  s.pixel = real_lo_bits(b.w);
  s.thr = V3<R>(c.x, c.y, c.z);
  s.sample = real_lo_bits(c.w);
  if constexpr (Guard<R>::on) s.org = Origin(q.org[i]);
  return s;
}
template <class R> RT_DEV void store_path(const PathQueue<R>& q, uint32_t i, const PathState<R>& s) {
  typename Vec4<R>::type a, b, c;
  a.x = s.ray.o.x; a.y = s.ray.o.y; a.z = s.ray.o.z; a.w = s.ray.time;
  b.x = s.ray.d.x; b.y = s.ray.d.y; b.z = s.ray.d.z; b.w = bits_to_real(s.pixel, 0u, (R)0);
  c.x = s.thr.x; c.y = s.thr.y; c.z = s.thr.z; c.w = bits_to_real(s.sample, 0u, (R)0);
  q.a[i] = a; q.b[i] = b; q.c[i] = c;
  if constexpr (Guard<R>::on) q.org[i] = s.org.bits;
}

// Wave-level stream compaction: every lane with `keep` receives a unique slot; one atomic per wave.
RT_DEV uint32_t compact_slot(bool keep, uint32_t* counter) {
  unsigned long long mask = __ballot(keep);
  uint32_t lane_prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  uint32_t total = (uint32_t)__popcll(mask);
  uint32_t base = 0;
  if (total != 0u) {
    int leader = __ffsll((long long)mask) - 1;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(counter, total);
    base = (uint32_t)__shfl((int)base, leader, 64);
  }
  return base + lane_prefix;
}

template <class R> RT_DEV void add_radiance(R* accum, uint32_t pixel, V3<R> c) {
#ifdef RT_EXP_NOATOMIC  // diagnostic build only: what the radiance atomics cost (the image stays black)
  if (c.x == (R)-1.2345) accum[pixel] = c.y;
  return;
#endif
  if (c.x != (R)0 || c.y != (R)0 || c.z != (R)0) {
    R* px = accum + (size_t)pixel * 4u;
    atomicAdd(px + 0, c.x);
    atomicAdd(px + 1, c.y);
    atomicAdd(px + 2, c.z);
  }
}

// Diagnostic build (-DRT_STAMPS): every wave sums the shader-clock cycles between phase boundaries of the bounce loop (all
// outstanding memory operations are drained at each boundary so a phase owns its own latency) and adds them to a slot of its own
// (8 words per wave) behind the ray counter.  No stamp executes in the normal build.
#ifdef RT_STAMPS
#define RT_STAMP_INIT() unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long stamp_t = __builtin_amdgcn_s_memtime()
#define RT_STAMP(k)                                                        \
  do {                                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            \
    unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
    stamp_acc[k] += now_ - stamp_t;                                        \
    stamp_t = now_;                                                        \
  } while (0)
#define RT_STAMP_FLUSH(base)                                               \
  do {                                                                     \
    if ((threadIdx.x & 63u) == 0u) {                                       \
      unsigned long long* slot_ = (base) + 16 + (size_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u; \
      for (int k_ = 0; k_ < 8; ++k_) slot_[k_] += stamp_acc[k_];           \
    }                                                                      \
  } while (0)
#define RT_STAMP_ATTACH(stream) do { (stream).stamp_acc = stamp_acc; (stream).stamp_t = &stamp_t; } while (0)
#define RT_STAMP_COUNT(k, n) do { stamp_acc[k] += (unsigned long long)(n); } while (0)
#else
#define RT_STAMP_COUNT(k, n) do {} while (0)
#define RT_STAMP_ATTACH(stream) do {} while (0)
#define RT_STAMP_INIT() do {} while (0)
#define RT_STAMP(k) do {} while (0)
#define RT_STAMP_FLUSH(base) do {} while (0)
#endif

struct FrameParams {
  uint32_t width, height, n_pixels;
  uint32_t sample_base;  // first sample index of this pass
  uint32_t n_paths;      // n_pixels * samples in this pass
  uint32_t max_depth;
  uint64_t seed;
  // The ray queue is split into n_seg independent segments of seg_cap slots.  Workgroup b serves segment b % n_seg only:
  // it reads that segment's rays and appends the survivors to the same segment of the next queue, so every segment has a
  // counter of its own (one shared counter saturates at under 10^8 returning atomics per second, far below the rate at
  // which waves finish), and because workgroups b and b + 8 share an XCD a segment's rays stay in one XCD's L2.
  uint32_t n_seg, seg_cap;
  // Paths a wave of the path kernels takes from the global cursor at a time: kPathGrab on large frames, fewer when the frame has
  // only a few grabs per wave to hand out (the last grabs decide how evenly the frame ends).
  uint32_t path_grab;
  // k_paths_persistent: a lane takes `chunk` consecutive samples of ONE pixel, sums their radiance in registers and adds the
  // sum to the pixel's slot once (the unit the feeder hands out is then a chunk: unit u = chunk index u / n_pixels of pixel
  // u % n_pixels).  `n_samples` = samples of this launch.  Frames whose paths are short and all bright (a textured globe in
  // front of a sky) otherwise spend two thirds of their time in the radiance atomics.
  uint32_t chunk, n_samples;
  // Row bands (rt_render_params::band_*): band_stride > 1 means that the launch renders only the rows of the bands b with
  // b % band_stride == band_offset, bands being 1 << band_shift rows; n_pixels then counts the pixels of those rows and the feeder
  // numbers them row after row ("local" rows), while `height` stays the frame's.
  uint32_t band_shift, band_stride, band_offset;
};

struct Segment {
  uint32_t seg, origin;  // segment id of this workgroup, first slot of the segment
};
RT_DEV Segment segment_of_block(const FrameParams& fp) {
  Segment g;
  g.seg = blockIdx.x % fp.n_seg;
  g.origin = g.seg * fp.seg_cap;
  return g;
}

// Work distribution inside a segment: every wave takes the next 64 queue entries from the segment's cursor, so the split
// does not depend on how many workgroups the hardware keeps resident and waves with cheap rays simply take more chunks.
// The request for the following chunk is issued before the current one is processed and read when it is needed.
struct ChunkCursor {
  uint32_t* cursor;
  uint32_t pending;  // lane 0: first entry of the next chunk
  RT_DEV explicit ChunkCursor(uint32_t* c) : cursor(c), pending(0) { request(); }
  RT_DEV void request() {
    if ((threadIdx.x & 63u) == 0u) pending = atomicAdd(cursor, 64u);
  }
  RT_DEV uint32_t take() {  // wave-uniform
    uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)pending);
    return base;
  }
};

// Primary rays for one pass: path index = sample_in_pass * n_pixels + pixel, so adjacent lanes hold adjacent pixels.
template <class R>
__global__ __launch_bounds__(256) void k_gen_primary(DeviceScene<R> sc, FrameParams fp, PathQueue<R> out, uint32_t* counters) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < fp.n_paths; i += stride) {
    uint32_t s = i / fp.n_pixels;
    uint32_t pixel = i - s * fp.n_pixels;
    PathState<R> st;
    st.pixel = pixel;
    st.sample = fp.sample_base + s;
    st.ray = primary_ray<R>(sc.camera, fp.width, fp.height, pixel, st.sample, fp.seed);
    st.thr = V3<R>((R)1, (R)1, (R)1);
    st.org = Origin();
    store_path(out, i, st);
  }
  if (blockIdx.x == 0 && threadIdx.x < fp.n_seg) {  // path i sits in slot i: segment s starts with paths [s*seg_cap, (s+1)*seg_cap)
    uint32_t lo = threadIdx.x * fp.seg_cap;
    counters[threadIdx.x] = lo >= fp.n_paths ? 0u : (fp.n_paths - lo < fp.seg_cap ? fp.n_paths - lo : fp.seg_cap);
  }
}

// One bounce for every queued ray: closest hit, miss shader or emitted + scatter, compaction into the next queue.
template <class R, class C>
__global__ __launch_bounds__(256) void k_bounce_fused(DeviceScene<R> sc, FrameParams fp, PathQueue<R> in, PathQueue<R> out,
                                                      uint32_t* counters, uint32_t* cursors, unsigned long long* ray_count, R* accum,
                                                      uint32_t bounce) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Segment sg = segment_of_block(fp);
  const uint32_t n_in = counters[bounce * fp.n_seg + sg.seg];
  if (n_in == 0u) return;
  const SceneView<R, C> sv = make_view<R, C>(sc, lds);
  const bool last = bounce + 1u >= fp.max_depth;  // ray_color(depth == 0) contributes nothing (src/render.rs:151-153)
  uint32_t* const out_count = counters + (bounce + 1u) * fp.n_seg + sg.seg;
  ChunkCursor chunks(cursors + bounce * fp.n_seg + sg.seg);
  RT_STAMP_INIT();
  for (;;) {
    const uint32_t base = chunks.take();
    RT_STAMP(0);
    if (base >= n_in) break;
    const uint32_t j = base + (threadIdx.x & 63u);
    const uint32_t i = sg.origin + j;
    bool alive = false;
    PathState<R> st;
    if (j < n_in) st = load_path(in, i);
    chunks.request();  // behind the loads: memory operations complete in issue order, so the loads do not wait for this atomic
    if (base == 0u && (threadIdx.x & 63u) == 0u) atomicAdd(ray_count, (unsigned long long)n_in);
    RT_STAMP(1);
    bool hit = false;
    Rec<R> rec;
    if (j < n_in) {
      Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
      hit = closest_hit<R, C>(sv, sc.root_link, st.ray, (R)0.001, Lim<R>::inf(), rec, media, st.org);
    }
    RT_STAMP(2);
    V3<R> radiance;
    if (j < n_in) {
      if (!hit) {
        radiance = st.thr * background_color<R, C>(sc.background, st.ray.d);
      } else {
        Stream rng(fp.seed, st.pixel, st.sample, bounce, STAGE_SCATTER);
        V3<R> emitted, attenuation;
        Ray<R> scattered;
        alive = shade_hit<R, C>(sv, st.ray, rec, rng, emitted, attenuation, scattered);
        radiance = st.thr * emitted;
        if (alive) {
          st.thr = st.thr * attenuation;
          st.ray = scattered;
          st.org = Origin(rec.leaf, rec.side);
        }
      }
    }
    RT_STAMP(3);
    if (j < n_in) add_radiance(accum, st.pixel, radiance);
    RT_STAMP(4);
    if (!last) {
      uint32_t slot = sg.origin + compact_slot(alive, out_count);
      RT_STAMP(5);
      if (alive) store_path(out, slot, st);
      RT_STAMP(6);
    }
  }
  RT_STAMP_FLUSH(ray_count);
}

// ---- persistent pipeline: whole paths per lane, dead lanes refilled in place ----
// Every lane carries one path through all of its bounces.  At the top of each iteration the lanes whose path has ended take
// the next unstarted (pixel, sample) pairs - ranks by ballot / prefix count inside the wave, path indices in blocks of
// kPathGrab per wave from one global cursor - so the wave stays full until the frame runs out of paths.  There are no ray
// queues: nothing is written to memory between bounces except the radiance sums.  A path's random streams are keyed by
// (pixel, sample, bounce), so its result does not depend on which lane or iteration it ran in.
constexpr uint32_t kPathGrab = 2048;
// Blocks of candidates a scatter draw may look at before it is put off to the lane's next visit to the shading code: the
// persistent kernel comes back every turn, the stepwise kernel only at its next restart.
#ifndef RT_BLOCKS_PER_RESTART
#define RT_BLOCKS_PER_RESTART 3
#endif
constexpr uint32_t kBlocksPerTurn = 2, kBlocksPerRestart = RT_BLOCKS_PER_RESTART;
#ifndef RT_NESTED_WAVES_F64
#define RT_NESTED_WAVES_F64 3  // double precision, final-scene: 1.24 / 1.45 / 1.39 Grays/s at 2 / 3 / 4 waves per SIMD (fewer spills against fewer waves to hide the walk's loads)
#endif
#ifndef RT_NESTED_WAVES
#define RT_NESTED_WAVES 4
#endif


// Hands unstarted paths to lanes.  Wave-uniform state: the wave owns `left` consecutive paths starting at pixel `pixel` of
// sample `sample` (path index = sample * n_pixels + pixel, kept as the pair so that no lane ever divides).
template <class R> struct PathFeeder {
  uint32_t left = 0, pixel = 0, sample = 0;
  uint32_t row = 0, col = 0;  // of `pixel` (row counted from the top): carried along so that no lane has to split a pixel index
  bool exhausted = false;  // the global cursor has passed total_paths

  // Gives every lane with `!active` a new path while there are any; lanes that receive one become active with bounce = 0.
  // MAKE_RAY = false leaves st.ray for the caller to fill in.
  // FIXED_GRAB: the grab is kPathGrab (large frames), a constant the compiler folds, instead of fp.path_grab.
  // `px_row`, `px_col`: row and column of the pixel a lane receives.
  template <bool MAKE_RAY, bool FIXED_GRAB> RT_DEV void refill(const DeviceScene<R>& sc, const FrameParams& fp, unsigned long long total_paths, unsigned long long* path_cursor,
                     PathState<R>& st, uint32_t& bounce, bool& active, uint32_t& px_row, uint32_t& px_col) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long dead = __ballot(!active);
    while (dead != 0ull && !exhausted) {
      if (left == 0u) {
        unsigned long long b = 0;
        const uint32_t grab = FIXED_GRAB ? kPathGrab : fp.path_grab;
        if (lane == 0u) b = atomicAdd(path_cursor, (unsigned long long)grab);
        uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
        b = ((unsigned long long)hi << 32) | lo;
        if (b >= total_paths) { exhausted = true; break; }
        const unsigned long long e = b + grab < total_paths ? b + grab : total_paths;
        left = (uint32_t)(e - b);
        sample = (uint32_t)(b / fp.n_pixels);
        pixel = (uint32_t)(b - (unsigned long long)sample * fp.n_pixels);
        row = pixel / fp.width;
        col = pixel - row * fp.width;
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
      const uint32_t n_dead = (uint32_t)__popcll(dead);
      uint32_t n_take = left < n_dead ? left : n_dead;
      const uint32_t row_left = fp.n_pixels - pixel;  // do not run past the end of a sample in one step
      n_take = n_take < row_left ? n_take : row_left;
      if (!active && rank < n_take) {
        st.pixel = pixel + rank;
        if (fp.width >= 64u) {  // the lanes of one step span at most two rows
          const uint32_t c = col + rank;
          const bool wrap = c >= fp.width;
          px_col = wrap ? c - fp.width : c;
          px_row = row + (wrap ? 1u : 0u);
        } else {
          pixel_row_col(st.pixel, fp.width, px_row, px_col);
        }
        if (fp.band_stride > 1u) {  // the same for every lane: local row -> row of the frame; the pixel keeps its whole-frame index
          const uint32_t band = px_row >> fp.band_shift;
          px_row = ((band * fp.band_stride + fp.band_offset) << fp.band_shift) + (px_row - (band << fp.band_shift));
          st.pixel = px_row * fp.width + px_col;
        }
        st.sample = fp.sample_base + sample;
        if constexpr (MAKE_RAY) st.ray = primary_ray<R>(sc.camera, fp.width, fp.height, st.pixel, st.sample, fp.seed);
        st.thr = V3<R>((R)1, (R)1, (R)1);
        st.org = Origin();
        bounce = 0;
        active = true;
      }
      left -= n_take;
      pixel += n_take;
      col += n_take;
      while (col >= fp.width) { col -= fp.width; row += 1u; }
      if (pixel == fp.n_pixels) { pixel = 0u; sample += 1u; row = 0u; col = 0u; }
      dead = __ballot(!active);
    }
  }
};

// Presets with nested trees want 250 registers; asking for four waves per SIMD (128 registers, some spills) is the faster trade.
#ifndef RT_SMALL_WAVES_F64
#define RT_SMALL_WAVES_F64 1
#endif
#ifndef RT_NOISE_WAVES
#define RT_NOISE_WAVES 4  // single precision only: the Perlin code wants 136 registers; 128 (four waves per SIMD) is worth 7 % on the noise scenes
#endif
// CHUNKED: see FrameParams::chunk.  Carrying a chunk's sum costs three registers and a dozen instructions a turn (5-15 % on
// frames that do not need it), so the launcher picks this variant only for frames made of short, cheap, bright paths.
// Waves per SIMD asked of the compiler for a double-precision flat-walk kernel (a cap on its registers).  Measured per preset: the
// smoke kernel wants 150 registers and is 9 % faster squeezed into 128 (four waves); the chunked spheres kernel stands at the edge
// of 256 (two waves; one register more and it would run a single wave: earth 48 -> 29 Grays/s) and is told to stay below; the
// unchunked spheres kernel (Perlin noise, simple-light) is 5 % faster with all the registers it wants and one wave.
template <class C, bool CHUNKED> constexpr int flat_waves_f64() { return C::has(F_MEDIUM) ? 4 : (CHUNKED ? 2 : RT_SMALL_WAVES_F64); }
template <class R, class C, bool FIXED_GRAB, bool CHUNKED = false>
__global__ __launch_bounds__(256, C::has(F_NESTED) ? (sizeof(R) == 8 ? RT_NESTED_WAVES_F64 : RT_NESTED_WAVES) : (C::has(F_NOISE) && sizeof(R) == 4 ? RT_NOISE_WAVES : (sizeof(R) == 8 ? flat_waves_f64<C, CHUNKED>() : 1))) void k_paths_persistent(DeviceScene<R> sc, FrameParams fp, unsigned long long total_paths,
                                                          unsigned long long* path_cursor, unsigned long long* ray_count, R* accum) {
  extern __shared__ __align__(16) uint8_t lds[];
  const SceneView<R, C> sv = make_view<R, C>(sc, lds);
  const uint32_t lane = threadIdx.x & 63u;
  PathFeeder<R> feeder;
  bool active = false;   // the lane holds a path
  bool hit = false;      // ... whose last ray hit something: `rec` waits to be shaded
  uint32_t resume = 0;   // != 0: the scatter draw of that hit is unfinished and continues at this block of its stream
  uint32_t bounce = 0;
  unsigned long long n_rays = 0;
  PathState<R> st;
  Rec<R> rec;
  V3<R> sum((R)0, (R)0, (R)0);  // radiance of the samples of the lane's chunk so far
  uint32_t more = 0;            // samples of the chunk still to start after the current one
  uint32_t px_row = 0, px_col = 0;  // row / column of the lane's pixel, set when the lane receives a path
  RT_STAMP_INIT();
  // One turn: paths that end at their last hit are closed - their lanes go on with the next sample of their chunk or, the chunk
  // done, add its sum to the pixel and take a new chunk - then ONE evaluation of Philox serves the scatter draw of the continuing
  // lanes and the pixel jitter of the starting ones, and all lanes trace their next ray.
  for (;;) {
    RT_STAMP(0);
    bool to_scatter = false;
    bool absorbed = false;
    if (active) {
      bool ended = true;
      V3<R> term((R)0, (R)0, (R)0);
      if (!hit) {
        term = st.thr * background_color<R, C>(sc.background, st.ray.d);
      } else if (sv.materials[rec.mat].tag == MAT_LIGHT) {  // emits, never scatters (src/material.rs:148-167)
        term = st.thr * material_color<R, C>(sv, sv.materials[rec.mat], rec);
      } else if (bounce + 1u >= fp.max_depth) {  // ray_color(depth == 0) contributes nothing (src/render.rs:151-153)
      } else {
        to_scatter = true;
        ended = false;
      }
      if constexpr (CHUNKED) {
        if (ended) {
          sum = sum + term;
          if (more != 0u) {  // the next sample of the same pixel, in place
            more -= 1u;
            st.sample += 1u;
            st.thr = V3<R>((R)1, (R)1, (R)1);
            st.org = Origin();
            bounce = 0;
          } else {
            add_radiance(accum, st.pixel, sum);
            active = false;
          }
        }
      } else {
        if (ended) {
          add_radiance(accum, st.pixel, term);
          active = false;
        }
      }
    }
    RT_STAMP(6);
    if constexpr (CHUNKED) {
      const bool was_active = active;
      feeder.template refill<false, FIXED_GRAB>(sc, fp, total_paths, path_cursor, st, bounce, active, px_row, px_col);
      if (active && !was_active) {  // a new chunk: the feeder's "sample" is the chunk's index
        const uint32_t first = (st.sample - fp.sample_base) * fp.chunk;
        const uint32_t n = fp.n_samples - first < fp.chunk ? fp.n_samples - first : fp.chunk;
        st.sample = fp.sample_base + first;
        more = n - 1u;
        sum = V3<R>((R)0, (R)0, (R)0);
      }
    } else {
      // (the pixel's row and column are used within this turn only; the single-precision noise kernel, capped at 128 registers,
      // is told so - nothing carried from turn to turn - while the other instantiations measure better left alone)
      if constexpr (C::has(F_NOISE) && sizeof(R) == sizeof(float)) { px_row = 0u; px_col = 0u; }
      feeder.template refill<false, FIXED_GRAB>(sc, fp, total_paths, path_cursor, st, bounce, active, px_row, px_col);
    }
    RT_STAMP(0);
    if (__ballot(active) == 0ull) break;
    const PrimedStream rng(Stream(fp.seed, st.pixel, st.sample, bounce, to_scatter ? STAGE_SCATTER : STAGE_CAMERA), to_scatter ? resume : 0u, kBlocksPerTurn);
    RT_STAMP(7);
    if (to_scatter) {
      V3<R> emitted, attenuation;
      Ray<R> scattered;
      const bool alive = shade_hit<R, C>(sv, st.ray, rec, rng, emitted, attenuation, scattered);
      resume = rng.resume;
      if (resume == 0u) {  // otherwise the draw goes on next turn, from block `resume`, and the lane keeps its hit until then
        ++bounce;
        st.thr = st.thr * attenuation;
        st.ray = scattered;
        st.org = Origin(rec.leaf, rec.side);
        if constexpr (CHUNKED) absorbed = !alive;
        else active = alive;
      }
    }
    RT_STAMP(2);
    if (!to_scatter && active) st.ray = primary_ray_at<R, C::relaxed()>(sv.camera(), fp.width, fp.height, px_row, px_col, rng);
    RT_STAMP(1);
    // an absorbed path (a metal's scattered ray below the surface, src/material.rs:88-90) ends here: it sits out this trace as a
    // miss that adds nothing, and the next turn closes it like any other ended path
    const bool go = active && resume == 0u && !absorbed;
    if (absorbed) { hit = false; st.thr = V3<R>((R)0, (R)0, (R)0); }
    n_rays += (unsigned long long)__popcll(__ballot(go));
    if (go) {
      hit = false;
      rec = Rec<R>();
      Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
      RT_STAMP_ATTACH(media);
      hit = closest_hit<R, C>(sv, sc.root_link, st.ray, (R)0.001, Lim<R>::inf(), rec, media, st.org);
    }
    RT_STAMP(1);
  }
  RT_STAMP_FLUSH(ray_count);
  if (lane == 0u && n_rays != 0ull) atomicAdd(ray_count, n_rays);
}

// The same pipeline for scenes whose tree every lane walks on its own.  Walks differ in length by an order of magnitude from ray
// to ray, so waiting for the slowest lane of a wave before shading leaves most lanes idle.  Here the tree position is part of the
// lane's state: every turn of the outer loop advances all walking lanes by a bounded number of entries, and as soon as
// kRestartLanes lanes have finished their walk they are shaded, given their next ray (or a new path) and sent back into the tree,
// while the other lanes keep their place.
#ifndef RT_STEP_WAVES
#define RT_STEP_WAVES 1
#endif
// The stepwise kernel with every feature compiled in wants 195 registers (two waves per SIMD); this asks for more waves.
#ifndef RT_STEP_WAVES_FULL
#define RT_STEP_WAVES_FULL 4
#endif
#ifndef RT_STEP_WAVES_FULL_F64
#define RT_STEP_WAVES_FULL_F64 3  // double precision: final-scene on the library's tree 3.71 / 3.95 / 3.83 Grays/s at 4 / 3 / 2 waves per SIMD, on the caller's tree no difference
#endif
#ifndef RT_STEP_WAVES_EXACT_F64
#define RT_STEP_WAVES_EXACT_F64 3  // the reference's leaf arithmetic on the four-wide walk (six rectangles to a box, wrapped spheres) and the list walk for rays that tie; 2 / 3 / 4 waves (256 / 168 / 128 registers; 74 put aside at three, all of them in the restart block): final-scene 4.5 / 5.2 / 5.1 Grays/s
#endif
#ifndef RT_WIDE_WAVES_F32
#define RT_WIDE_WAVES_F32 0  // != 0: waves per SIMD asked for the single-precision four-wide kernels (measurement switch; needs RT_WIDE_STACK to match the LDS)
#endif
#ifndef RT_STEP_THREADS
#define RT_STEP_THREADS 256
#endif
constexpr uint32_t kStepThreads = RT_STEP_THREADS;
static_assert(RT_STEP_THREADS <= 256, "the four-wide presets size their LDS stacks for workgroups of four waves (lds_bytes_for)");
#ifndef RT_RESTART_LANES
#define RT_RESTART_LANES 32
#define RT_BOX_STEPS 4
#define RT_TURNS 4
#endif
#ifndef RT_LEAF_LANES
#define RT_LEAF_LANES 16
#endif
constexpr uint32_t kLeafLanes = RT_LEAF_LANES;
#ifndef RT_LEAF_LANES_COSTLY
#define RT_LEAF_LANES_COSTLY 24
#endif
constexpr uint32_t kLeafLanesCostly = RT_LEAF_LANES_COSTLY;
constexpr uint32_t kRestartLanes = RT_RESTART_LANES;
#ifndef RT_RESTART_LANES_WIDE
#define RT_RESTART_LANES_WIDE 40  // the four-wide walk: 32 / 40 / 48 idle lanes measure 9.5 / 9.8 / 9.8 Grays/s on the million-sphere field, 8.55 / 8.63 / 8.42 on final-scene
#endif
constexpr uint32_t kRestartLanesWide = RT_RESTART_LANES_WIDE;
constexpr int kBoxStepsPerTurn = RT_BOX_STEPS;
constexpr int kTurnsPerCheck = RT_TURNS;
#ifndef RT_WIDE_STEPS
#define RT_WIDE_STEPS 1
#endif
constexpr int kWideStepsPerTurn = RT_WIDE_STEPS;  // node steps of the four-wide walk between two looks at the waiting leaves

// The stepwise kernel's stamps.  -DRT_STAMPS: [0] restart blocks, [1] node steps, [2] leaf steps and the lane counts in [3]..[7].
// -DRT_STAMPS -DRT_STAMPS_RESTART: cycles only, the restart block cut up - [0] rays traced again at a tie, [1] records, [2] ended
// paths (background, lights, depth), [3] refill, [4] Philox + scatter + primary rays, [5] walk set-up and the objects kept outside
// the tree, [6] node steps, [7] leaf steps.
#ifdef RT_STAMPS_RESTART
#define SW_STAMP(k) RT_STAMP((k) + 5)
#define SW_RSTAMP(k) RT_STAMP(k)
#define SW_COUNT(k, n) do {} while (0)
#else
#define SW_STAMP(k) RT_STAMP(k)
#define SW_RSTAMP(k) do {} while (0)
#define SW_COUNT(k, n) RT_STAMP_COUNT(k, n)
#endif
#ifndef RT_STEP_WAVES_BALLS_F64
#define RT_STEP_WAVES_BALLS_F64 4  // the double-precision four-wide walk of sphere scenes: 127 to 143 registers as the compiler pleases; held at 128
#endif
// Waves per SIMD asked of the compiler for an instantiation of the stepwise kernel (a cap on its registers).
template <class R, class C> constexpr int step_waves() {
  if (RT_WIDE_WAVES_F32 && C::walk == WALK_LANE_WIDE && sizeof(R) == 4) return RT_WIDE_WAVES_F32;
  if (C::has(F_NESTED)) return RT_NESTED_WAVES;
  if (C::has(F_NOISE)) return sizeof(R) == 8 ? (C::has(F_EXACT) ? RT_STEP_WAVES_EXACT_F64 : RT_STEP_WAVES_FULL_F64) : RT_STEP_WAVES_FULL;
  if (C::walk == WALK_LANE_WIDE && sizeof(R) == 8) return RT_STEP_WAVES_BALLS_F64;
  return RT_STEP_WAVES;
}
template <class R, class C, bool FIXED_GRAB, bool DEFER_LIST = false>
__global__ __launch_bounds__(RT_STEP_THREADS, (step_waves<R, C>())) void k_paths_stepwise(DeviceScene<R> sc, FrameParams fp, unsigned long long total_paths,
                                                        unsigned long long* path_cursor, unsigned long long* ray_count, R* accum) {
  extern __shared__ __align__(16) uint8_t lds[];
  const SceneView<R, C> sv = make_view<R, C>(sc, lds);
  const uint32_t lane = threadIdx.x & 63u;
  PathFeeder<R> feeder;
  bool active = false;  // the lane holds a path
  uint32_t resume = 0;  // != 0: the scatter draw of the lane's hit is unfinished and continues at this block of its stream
  uint32_t bounce = 0;
  unsigned long long n_rays = 0;
  PathState<R> st;
  uint32_t px_row = 0, px_col = 0;  // row / column of the lane's pixel, set when the lane receives a path
  // walk state of the lane's current ray
  constexpr bool WIDE = C::walk == WALK_LANE_WIDE;
  int i = LINK_END;     // entry to look at next; LINK_END = no walk in progress.  Four-wide walk: 0 while `wl` holds a walk
  BoxLink lk{};         // entry i
  WideLane<R> wl;       // four-wide walk: the child to look at next and the lane's stack
  // Single precision, presets with boxes / wrappers / media (Cfg::defer_records): the winning candidate is kept as (closest, leaf,
  // side) while the walk lasts and the record is built once, when the lane is shaded (device_trace_deferred.h).  Double precision
  // on the caller's tree fills the record at every accepted candidate, as the reference does (a hit that writes no (u, v) must find
  // the ones an earlier candidate left), unless the launcher knows that nothing reads such a leftover (DEFER_LIST).
  constexpr bool DEFER = C::template defer_records<R>() || (DEFER_LIST && C::costly_records());
  // presets with boxes, wrappers or media have the longer leaf code and wait for more lanes before they run it (final-scene +2-3 %);
  // not so the double-precision four-wide walk, whose node steps cost as much as its leaves (final-scene f64 at 8 / 12 / 16 / 24 /
  // 32 lanes: 4.29 / 4.38 / 4.38 / 4.30 / 4.18 Grays/s)
  constexpr uint32_t leaf_threshold = C::costly_records() && !(WIDE && sizeof(R) == sizeof(double)) ? kLeafLanesCostly : kLeafLanes;
  int best_leaf = -1, best_side = -1;
  R closest = Lim<R>::inf();
  bool any = false;
  Rec<R> rec;
  LaneBoxTest<R> box(Ray<R>{});
  // the test of one leaf against the lane's ray
  // F_EXACT: two different leaves at exactly the same distance - the order of the tests would decide (see traverse_wide)
  constexpr bool EXACT = WIDE && C::has(F_EXACT) && sizeof(R) == sizeof(double);
  bool tie = false;
  unsigned long long retraced = 0;  // rays of the wave answered a second time by the list walk (the same in every lane)
  auto step_leaf = [&](uint32_t word, Stream& media) {
    if constexpr (DEFER) {
      RayInv<R> ri(st.ray.d);
      ri.inv = box.inv;
      R t;
      int side;
      if (leaf_t<R, C>(sv, word, st.ray, ri, (R)0.001, closest, media, st.org, t, side, visit_bit(word, 0))) {
        const int leaf = (int)(word & LEAF_INDEX_MASK);
        if constexpr (EXACT) tie = tie | (best_leaf >= 0 && best_leaf != leaf && t == closest);
        closest = t; best_leaf = leaf; best_side = side;
      }
    } else {
      const int prev_leaf = rec.leaf;
      const R prev_t = closest;
      if (leaf_hit<R, C, 0>(sv, word, st.ray, (R)0.001, closest, rec, media, st.org)) {
        if constexpr (EXACT) tie = tie | (any && rec.leaf >= 0 && prev_leaf >= 0 && prev_leaf != rec.leaf && rec.t == prev_t);
        closest = rec.t; any = true;
      }
    }
  };
  RT_STAMP_INIT();  // diagnostic build: cycles of [0] restart blocks, [1] box steps, [2] leaf steps; lanes taking part in [3] box steps,
                    // [5] leaf steps, [7] restarts, and the number of [4] box steps, [6] leaf steps (restarts: [7] / lanes per restart)
  for (;;) {
    const unsigned long long idle = __ballot(i < 0);
    if ((uint32_t)__popcll(idle) >= (WIDE ? kRestartLanesWide : kRestartLanes) || idle == ~0ull) {
      bool to_scatter = false;
      // Kernels that build the record after the walk use it inside this block only: a lane whose scatter draw goes on at the next
      // restart keeps (closest, leaf, side) - or its tie flag - and has the record built again then.  Cleared here, the record's
      // two dozen registers are not carried across the walk.
      if constexpr (DEFER) rec = Rec<R>();
      if constexpr (EXACT) {
        // A walk that met a tie - two different leaves at exactly the same distance - answers nothing by itself: which of the two
        // the reference reports hangs on the order of its walk and on its boxes (src/bvh.rs:108-119, src/aabb.rs:40).  The lane
        // traces the ray once more, by the list walk of the caller's tree in the reference's order, while the other lanes of the
        // wave wait: a few thousand rays of a frame's half billion (final-scene: rays that slip between two ground blocks), a
        // fraction of a percent of the waves' time.  Word 1 of the counter block counts them.
        const bool redo = i < 0 && active && tie;
        // (a lane in the middle of its scatter draw comes by again with its answer, no longer with the tie)
        retraced += (unsigned long long)__popcll(__ballot(redo));
        if (redo) {
          // the same leaf tests as the four-wide walk's, over the threaded list: (closest, leaf, side) of the reference's answer,
          // from which the record is built below like any other
          Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
          RayInv<R> ri(st.ray.d);
          ri.inv = box.inv;
          closest = Lim<R>::inf();
          best_leaf = -1; best_side = -1;
          int j = sc.root_link;
          while (j >= 0) {
            BoxLink e = sv.link(j);
            while (!(e.hit & LEAF_FLAG)) {
              j = box.hit(e, st.ray, (R)0.001, closest, sv.exact_boxes, j) ? (int)e.hit : e.miss;
              if (j < 0) break;
              e = sv.link(j);
            }
            if (j < 0) break;
            R t;
            int side;
            if (leaf_t<R, C>(sv, e.hit, st.ray, ri, (R)0.001, closest, media, st.org, t, side, visit_bit(e.hit, 0))) {
              closest = t; best_leaf = (int)(e.hit & LEAF_INDEX_MASK); best_side = side;
            }
            j = e.miss;
          }
          tie = false;
        }
      }
      if constexpr (DEFER) {
        if (i < 0 && active && best_leaf >= 0 && !tie) {  // the walk has ended on a hit: its record
          finalize_hit<R, C>(sv, st.ray, closest, best_leaf, best_side, rec);
          any = true;
        }
      }
      SW_RSTAMP(1);
      if (i < 0 && active) {
        if (!any) {
          add_radiance(accum, st.pixel, st.thr * background_color<R, C>(sc.background, st.ray.d));
          active = false;
        } else if (sv.materials[rec.mat].tag == MAT_LIGHT) {  // emits, never scatters (src/material.rs:148-167)
          add_radiance(accum, st.pixel, st.thr * material_color<R, C>(sv, sv.materials[rec.mat], rec));
          active = false;
        } else if (bounce + 1u >= fp.max_depth) {  // ray_color(depth == 0) contributes nothing (src/render.rs:151-153)
          active = false;
        } else {
          to_scatter = true;
        }
      }
      // used within this restart only; cleared so that nothing is carried across the walk (two registers the single-precision kernels
      // with every feature were spilling: final-scene f32 8.3 -> 8.65 Grays/s.  The flat-walk kernel is better off without: cleared
      // there, cornell-box f64 goes from 54.1 to 50.6)
      SW_RSTAMP(2);
      px_row = 0u; px_col = 0u;
      feeder.template refill<false, FIXED_GRAB>(sc, fp, total_paths, path_cursor, st, bounce, active, px_row, px_col);
      SW_RSTAMP(3);
      if (i < 0 && active) {  // one Philox pass: scatter draw of the continuing lanes, pixel jitter of the new ones
        const PrimedStream rng(Stream(fp.seed, st.pixel, st.sample, bounce, to_scatter ? STAGE_SCATTER : STAGE_CAMERA), to_scatter ? resume : 0u, kBlocksPerRestart);
        if (to_scatter) {
          V3<R> emitted, attenuation;
          Ray<R> scattered;
          const bool alive = shade_hit<R, C>(sv, st.ray, rec, rng, emitted, attenuation, scattered);
          resume = rng.resume;
          if (resume == 0u) {  // otherwise the draw goes on at the next restart, and the lane keeps its hit until then
            ++bounce;
            st.thr = st.thr * attenuation;
            st.ray = scattered;
            st.org = Origin(rec.leaf, rec.side);
            active = alive;
          }
        } else {
          st.ray = primary_ray_at<R, C::relaxed()>(sv.camera(), fp.width, fp.height, px_row, px_col, rng);
        }
      }
      SW_RSTAMP(4);
      const bool start = i < 0 && active && resume == 0u;
      n_rays += (unsigned long long)__popcll(__ballot(start));
      SW_COUNT(7, __popcll(__ballot(i < 0)));
      if (start) {
        box = LaneBoxTest<R>(st.ray, WIDE && sv.narrow_ok);
        if constexpr (WIDE) { wl.start(); i = 0; }
        else { i = sc.root_link; lk = sv.link(i); }
        closest = Lim<R>::inf();
        any = false;
        best_leaf = -1;
        tie = false;
        rec = Rec<R>();  // a fresh HitRecord per world.hit call (src/render.rs:157)
        if constexpr (WIDE) {  // objects as large as the scene, kept outside the tree: all starting lanes test them together
          Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
          for (uint32_t k = 0; k < sc.n_pre; ++k) step_leaf(sc.pre[k], media);
        }
      }
      if (__ballot(i >= 0 || resume != 0u) == 0ull) break;  // no walk in progress, no draw unfinished, no path left to start
    }
    SW_STAMP(0);
    if constexpr (WIDE) {
#pragma unroll 1
      for (int turn = 0; turn < kTurnsPerCheck; ++turn) {
#pragma unroll 1
        for (int k = 0; k < kWideStepsPerTurn; ++k) {
          const bool at_box = wl.at_box();
          if (__ballot(at_box) == 0ull) break;
          SW_COUNT(3, __popcll(__ballot(at_box)));
          SW_COUNT(4, 1);
          if (at_box) wl.box_step(sv.wide, sv.stack, box, st.ray, (R)0.001, closest);
        }
        SW_STAMP(1);
        const bool at_leaf = wl.at_leaf();
        const unsigned long long leaf_lanes = __ballot(at_leaf);
        if ((uint32_t)__popcll(leaf_lanes) < leaf_threshold && __ballot(wl.at_box()) != 0ull) continue;
        SW_COUNT(5, __popcll(leaf_lanes));
        SW_COUNT(6, 1);
        if (at_leaf) {
          Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
          step_leaf(wl.cur, media);
          wl.pop(sv.stack);
        }
        SW_STAMP(2);
      }
      if (wl.idle()) i = LINK_END;
      continue;
    }
#pragma unroll 1
    for (int turn = 0; turn < kTurnsPerCheck; ++turn) {
#pragma unroll 1
      for (int k = 0; k < kBoxStepsPerTurn; ++k) {
        const bool at_box = i >= 0 && !(lk.hit & LEAF_FLAG);
        if (__ballot(at_box) == 0ull) break;
        SW_COUNT(3, __popcll(__ballot(at_box)));
        SW_COUNT(4, 1);
        if (at_box) {
          i = box.hit(lk, st.ray, (R)0.001, closest) ? (int)lk.hit : lk.miss;
          if (i >= 0) lk = sv.link(i);
        }
      }
      // Leaves are tested when enough lanes stand on one (or no lane has a box to test): the leaf code is several times longer than
      // a box step, and run for the one or two lanes that happen to have arrived it would cost as much as for a full wave.
      SW_STAMP(1);
      const bool at_leaf = i >= 0 && (lk.hit & LEAF_FLAG);
      const unsigned long long leaf_lanes = __ballot(at_leaf);
      if ((uint32_t)__popcll(leaf_lanes) < leaf_threshold && __ballot(i >= 0 && !(lk.hit & LEAF_FLAG)) != 0ull) continue;
      SW_COUNT(5, __popcll(leaf_lanes));
      SW_COUNT(6, 1);
      if (at_leaf) {
        Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
        step_leaf(lk.hit, media);
        i = lk.miss;
        if (i >= 0) lk = sv.link(i);
      }
      SW_STAMP(2);
    }
  }
  RT_STAMP_FLUSH(ray_count);
  if (lane == 0u && n_rays != 0ull) atomicAdd(ray_count, n_rays);
  if constexpr (EXACT) {
    if (lane == 0u && retraced != 0ull) atomicAdd(ray_count + 1, retraced);
  }
}

// ---- material-sorted pipeline: trace kernel + shade kernel ----
template <class R> struct HitQueue {
  using R4 = typename Vec4<R>::type;
  R4* a;  // p, t
  R4* b;  // n, (front | mat << 1) bits
  R4* c;  // u, v, Origin bits of the surface hit, unused
};

// class_counters: 8 counters per segment for this bounce (MAT_CLASSES used); class_index[(cls * n_seg + seg) * seg_cap + k] = ray slot.
template <class R, class C>
__global__ __launch_bounds__(256) void k_trace(DeviceScene<R> sc, FrameParams fp, PathQueue<R> in, HitQueue<R> hits, uint32_t* counters,
                                               uint32_t* cursors, unsigned long long* ray_count, uint32_t* class_counters, uint32_t* class_index,
                                               uint32_t capacity, R* accum, uint32_t bounce) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Segment sg = segment_of_block(fp);
  const uint32_t n_in = counters[bounce * fp.n_seg + sg.seg];
  if (n_in == 0u) return;
  const SceneView<R, C> sv = make_view<R, C>(sc, lds);
  ChunkCursor chunks(cursors + bounce * fp.n_seg + sg.seg);
  for (;;) {
    const uint32_t base = chunks.take();
    if (base >= n_in) break;
    chunks.request();
    if (base == 0u && (threadIdx.x & 63u) == 0u) atomicAdd(ray_count, (unsigned long long)n_in);
    const uint32_t j = base + (threadIdx.x & 63u);
    const uint32_t i = sg.origin + j;
    int cls = -1;
    if (j < n_in) {
      PathState<R> st = load_path(in, i);
      Stream media(fp.seed, st.pixel, st.sample, bounce, STAGE_MEDIA);
      Rec<R> rec;
      if (!closest_hit<R, C>(sv, sc.root_link, st.ray, (R)0.001, Lim<R>::inf(), rec, media, st.org)) {
        add_radiance(accum, st.pixel, st.thr * background_color<R, C>(sc.background, st.ray.d));
      } else {
        rec.finish_uv();
        cls = sv.materials[rec.mat].tag;
        typename Vec4<R>::type a, b, c;
        a.x = rec.p.x; a.y = rec.p.y; a.z = rec.p.z; a.w = rec.t;
        b.x = rec.n.x; b.y = rec.n.y; b.z = rec.n.z; b.w = bits_to_real((uint32_t)rec.front | ((uint32_t)rec.mat << 1), 0u, (R)0);
        c.x = rec.u; c.y = rec.v; c.z = bits_to_real(Origin(rec.leaf, rec.side).bits, 0u, (R)0); c.w = (R)0;
        hits.a[i] = a; hits.b[i] = b; hits.c[i] = c;
      }
    }
#pragma unroll
    for (int k = 0; k < MAT_CLASSES; ++k) {
      bool mine = cls == k;
      if (__ballot(mine) != 0ull) {
        uint32_t slot = compact_slot(mine, class_counters + sg.seg * 8u + k);
        if (mine) class_index[((size_t)k * fp.n_seg + sg.seg) * fp.seg_cap + slot] = i;
      }
    }
  }
}

template <class R, class C>
__global__ __launch_bounds__(256) void k_shade(DeviceScene<R> sc, FrameParams fp, PathQueue<R> in, HitQueue<R> hits, PathQueue<R> out,
                                               uint32_t* counters, uint32_t* cursors, const uint32_t* class_counters, const uint32_t* class_index,
                                               uint32_t capacity, R* accum, uint32_t bounce) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Segment sg = segment_of_block(fp);
  uint32_t cls_end[MAT_CLASSES];
  uint32_t total = 0;
#pragma unroll
  for (int k = 0; k < MAT_CLASSES; ++k) {
    total += class_counters[sg.seg * 8u + k];
    cls_end[k] = total;
  }
  if (total == 0u) return;
  const SceneView<R, C> sv = make_view<R, C>(sc, lds);
  const bool last = bounce + 1u >= fp.max_depth;
  uint32_t* const out_count = counters + (bounce + 1u) * fp.n_seg + sg.seg;
  ChunkCursor chunks(cursors + bounce * fp.n_seg + sg.seg);
  for (;;) {
    const uint32_t base = chunks.take();
    if (base >= total) break;
    chunks.request();
    uint32_t w = base + (threadIdx.x & 63u);
    bool alive = false;
    PathState<R> st;
    if (w < total) {
      // work item w belongs to the first class whose cumulative end exceeds it
      uint32_t cls = 0, begin = 0;
#pragma unroll
      for (int k = 0; k < MAT_CLASSES - 1; ++k)
        if (w >= cls_end[k]) { cls = (uint32_t)k + 1u; begin = cls_end[k]; }
      uint32_t i = class_index[((size_t)cls * fp.n_seg + sg.seg) * fp.seg_cap + (w - begin)];
      st = load_path(in, i);
      typename Vec4<R>::type a = hits.a[i], b = hits.b[i], c = hits.c[i];
      Rec<R> rec;
      rec.p = V3<R>(a.x, a.y, a.z); rec.t = a.w;
      rec.n = V3<R>(b.x, b.y, b.z);
      uint32_t bits = real_lo_bits(b.w);
      rec.front = (int32_t)(bits & 1u); rec.mat = (int32_t)(bits >> 1);
      rec.u = c.x; rec.v = c.y; rec.uv_lazy = 0;
      const Origin hit_surface(real_lo_bits(c.z));
      Stream rng(fp.seed, st.pixel, st.sample, bounce, STAGE_SCATTER);
      V3<R> emitted, attenuation;
      Ray<R> scattered;
      alive = shade_hit<R, C>(sv, st.ray, rec, rng, emitted, attenuation, scattered);
      add_radiance(accum, st.pixel, st.thr * emitted);
      if (alive) {
        st.thr = st.thr * attenuation;
        st.ray = scattered;
        st.org = hit_surface;
      }
    }
    if (!last) {
      uint32_t slot = sg.origin + compact_slot(alive, out_count);
      if (alive) store_path(out, slot, st);
    }
  }
}

// ---- resolve: color_to_pixel over the accumulation buffer (src/color.rs:14-33), evaluated in f64 ----
template <class R> __global__ void k_resolve(const R* accum, uint32_t n_pixels, uint32_t spp, uint8_t* rgb) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  double scale = 1.0 / (double)spp;
  rgb[i * 3u + 0] = to_u8<double>((double)accum[(size_t)i * 4u + 0], scale);
  rgb[i * 3u + 1] = to_u8<double>((double)accum[(size_t)i * 4u + 1], scale);
  rgb[i * 3u + 2] = to_u8<double>((double)accum[(size_t)i * 4u + 2], scale);
}

// ---- test / inspection entry points ----
struct RayIO {  // = rt_ray
  double o[3], d[3], time, tmin, tmax;
  uint32_t pixel, sample, bounce, pad;
};
struct HitIO {  // = rt_hit
  int32_t hit, obj, sub, side, material, front_face;
  double t, p[3], n[3], u, v;
};

// MODE 0: the inspection entry reports (u, v) whatever the scene's textures read.  MODE 1 (as rendered): the kernel configuration
// the render kernels run, so that their record building is what gets inspected ((u, v) then stay 0 in scenes that never read
// them).  MODE 2: as rendered, and u, v report the numbers of box entries and of leaves the walk tested for the ray.
template <class R, class C, int MODE>
__global__ __launch_bounds__(256) void k_first_hit(DeviceScene<R> sc, const RayIO* rays, uint32_t n, HitIO* hits, uint64_t seed) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr bool AS_RENDERED = MODE != 0;
  using CU = Cfg<AS_RENDERED ? C::F : (C::F | F_UV), C::walk>;
  const SceneView<R, CU> sv = make_view<R, CU>(sc, lds);
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    RayIO r = rays[i];
    Ray<R> ray;
    ray.o = V3<R>((R)r.o[0], (R)r.o[1], (R)r.o[2]);
    ray.d = V3<R>((R)r.d[0], (R)r.d[1], (R)r.d[2]);
    ray.time = (R)r.time;
    Stream media(seed, r.pixel, r.sample, r.bounce, STAGE_MEDIA);
    Rec<R> rec;
    Visits tally;
    bool h;
    if constexpr (MODE == 0 && sizeof(R) == 8 && C::walk == WALK_LANE_WIDE) {
      // The plain query in double precision reports the record the reference's walk leaves behind, leftover (u, v) of a moving
      // sphere included, and those depend on the order of the tests: it walks the list, which every image carries and which keeps
      // the tree's own order.  The render kernels' walk is what the as-rendered query shows.
      h = traverse_lane<R, CU, 0>(sv, sc.root_link, ray, (R)r.tmin, (R)r.tmax, rec, media, Origin(), 0u, nullptr);
    } else {
      h = closest_hit<R, CU>(sv, sc.root_link, ray, (R)r.tmin, (R)r.tmax, rec, media, Origin(), MODE == 2 ? &tally : nullptr);
    }
    HitIO o;
    o.hit = h ? 1 : 0;
    if (h) {
      rec.finish_uv();
      o.obj = rec.obj; o.sub = rec.sub; o.side = rec.side; o.material = rec.mat; o.front_face = rec.front;
      o.t = (double)rec.t; o.u = (double)rec.u; o.v = (double)rec.v;
      o.p[0] = (double)rec.p.x; o.p[1] = (double)rec.p.y; o.p[2] = (double)rec.p.z;
      o.n[0] = (double)rec.n.x; o.n[1] = (double)rec.n.y; o.n[2] = (double)rec.n.z;
    } else {
      o.obj = o.sub = o.side = o.material = -1; o.front_face = 0;
      o.t = o.u = o.v = 0.0;
      o.p[0] = o.p[1] = o.p[2] = o.n[0] = o.n[1] = o.n[2] = 0.0;
    }
    if constexpr (MODE == 2) { o.u = (double)tally.boxes; o.v = (double)tally.leaves; }
    hits[i] = o;
  }
}

template <class R>
__global__ void k_camera_rays(DeviceScene<R> sc, uint32_t W, uint32_t H, const uint32_t* pixel, const uint32_t* sample, uint32_t n,
                              uint64_t seed, RayIO* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray<R> ray = primary_ray<R>(sc.camera, W, H, pixel[i], sample[i], seed);
  RayIO r;
  r.o[0] = (double)ray.o.x; r.o[1] = (double)ray.o.y; r.o[2] = (double)ray.o.z;
  r.d[0] = (double)ray.d.x; r.d[1] = (double)ray.d.y; r.d[2] = (double)ray.d.z;
  r.time = (double)ray.time; r.tmin = 0.001; r.tmax = __builtin_inf();
  r.pixel = pixel[i]; r.sample = sample[i]; r.bounce = 0; r.pad = 0;
  out[i] = r;
}

// Whole path in one thread (one lane follows one camera sample to termination).  Used to compare single samples
// with a CPU evaluation; the render path uses the queue kernels above.
template <class R, class C>
__global__ __launch_bounds__(256) void k_sample_radiance(DeviceScene<R> sc, uint32_t W, uint32_t H, uint32_t max_depth, const uint32_t* pixel,
                                                         const uint32_t* sample, uint32_t n, uint64_t seed, double* out, double* path_out,
                                                         uint32_t max_vertices) {
  extern __shared__ __align__(16) uint8_t lds[];
  const SceneView<R, C> sv = make_view<R, C>(sc, lds);
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t px = pixel[i], sm = sample[i];
    Ray<R> ray = primary_ray<R, C::relaxed()>(sc.camera, W, H, px, sm, seed);
    V3<R> thr((R)1, (R)1, (R)1), sum((R)0, (R)0, (R)0);
    uint32_t n_rays = 0;
    Origin org;
    for (uint32_t bounce = 0; bounce < max_depth; ++bounce) {
      ++n_rays;
      Stream media(seed, px, sm, bounce, STAGE_MEDIA);
      Rec<R> rec;
      bool hit = closest_hit<R, C>(sv, sc.root_link, ray, (R)0.001, Lim<R>::inf(), rec, media, org);
      if (path_out && bounce < max_vertices) {  // one vertex = origin(3) direction(3) t obj sub material front hit
        double* v = path_out + ((size_t)i * max_vertices + bounce) * 12u;
        v[0] = (double)ray.o.x; v[1] = (double)ray.o.y; v[2] = (double)ray.o.z;
        v[3] = (double)ray.d.x; v[4] = (double)ray.d.y; v[5] = (double)ray.d.z;
        v[6] = hit ? (double)rec.t : 0.0; v[7] = hit ? (double)rec.obj : -1.0; v[8] = hit ? (double)rec.sub : -1.0;
        v[9] = hit ? (double)rec.mat : -1.0; v[10] = hit ? (double)rec.front : 0.0; v[11] = hit ? 1.0 : 0.0;
      }
      if (!hit) {
        sum = sum + thr * background_color<R, C>(sc.background, ray.d);
        break;
      }
      Stream rng(seed, px, sm, bounce, STAGE_SCATTER);
      V3<R> emitted, attenuation;
      Ray<R> scattered;
      bool alive = shade_hit<R, C>(sv, ray, rec, rng, emitted, attenuation, scattered);
      sum = sum + thr * emitted;
      if (!alive) break;
      thr = thr * attenuation;
      ray = scattered;
      org = Origin(rec.leaf, rec.side);
    }
    out[(size_t)i * 4u + 0] = (double)sum.x;
    out[(size_t)i * 4u + 1] = (double)sum.y;
    out[(size_t)i * 4u + 2] = (double)sum.z;
    out[(size_t)i * 4u + 3] = (double)n_rays;
  }
}

}  // namespace rtd
