"""Headline benchmark: Mrays/s on cornell_box 1920x1080, 1024 samples per pixel, depth 50, on 1..8 MI355X.

One process per GPU.  A step is one complete render of the frame: every rank renders its share of the sample range
[rank*S/N, (rank+1)*S/N) into its own accumulation buffer (scene already resident in HBM), the buffers are summed onto
rank 0 with one reduce over RCCL, and rank 0 converts the sums to RGB8 and copies the image to the host.  Total work is
fixed as N grows ("strong" scaling).  Rank 0 prints one JSON line.

The reference computes in f64 (src/vec3.rs:5-10), so the top-level figures (`value`, `dtype`, `roofline`) are those of the
double-precision kernels.  The single-precision kernels are a tolerance mode: the same frame is timed with them right after,
and reported in the `f32` sub-record with its own roofline and the error of its image against the f64 image of the same run.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))

# Flop weights per event (adds and multiplies count 1; divide, sqrt and transcendentals count 1 each): the algorithmic
# cost of the reference's own arithmetic, applied to event counts of the oracle on a sample of the same workload.
FLOP_WEIGHTS = {
    "aabb": 15, "sphere_miss": 23, "sphere_hit": 54, "moving_miss": 47, "moving_hit": 78, "rect_miss": 8, "rect_hit": 27,
    "translate": 11, "rotate": 29, "medium": 12, "sc_lambert": 60, "sc_metal": 75, "sc_dielectric": 55, "sc_light": 0, "sc_iso": 40,
    "tx_solid": 0, "tx_checker": 8, "tx_noise": 7 * 180 + 7, "tx_image": 10, "bg": 22,
}
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # MI355X vector peaks
HBM_PEAK_GBS = 8000.0
QUEUE_BYTES_PER_RAY = {"f32": 2 * 52, "f64": 2 * 96}  # one queue entry read + one written per ray (fused pipeline)


def flops_per_ray(cnt):
    return sum(FLOP_WEIGHTS[k] * cnt[k] for k in FLOP_WEIGHTS) / max(1, cnt["rays"])


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """Cores this job may use: the affinity mask, capped by the pool's OMP_NUM_THREADS-style limit (a box's CPU share is far
    below the machine's core count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            n = min(n, int(v))
            break
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cornell-box")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--aspect", type=float, default=16.0 / 9.0)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--mega-grid", type=int, default=0, help="mega-spheres: side of the sphere grid (500 = one million spheres)")
    ap.add_argument("--mega-view", choices=["across", "above"], default="across",
                    help="mega-spheres: across = camera at (13, 6, 3), a sliver of the field in the frame; above = looking down on the whole field")
    ap.add_argument("--precision", choices=["f32", "f64", "both"], default="both",
                    help="both = f64 headline + f32 sub-record with its image error against the f64 frame")
    ap.add_argument("--pipeline", choices=["auto", "fused", "sorted", "persistent"], default="auto")
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=96, help="samples per pixel of the bounded CPU sample")
    ap.add_argument("--save", default="", help="rank 0 writes the last image of the headline precision here (.png/.ppm)")
    ap.add_argument("--reduce", choices=["lib", "torch"], default="lib",
                    help="lib = the library's own rt_reduce_accum (RCCL communicator created through rt_comm_init, its id broadcast by "
                         "torch.distributed); torch = torch.distributed.reduce.  lib falls back to torch when the communicator cannot be made")
    ap.add_argument("--dump-accum", default="", help="rank 0 writes the headline precision's summed radiance (H x W x 3 float64, .npy) here")
    ap.add_argument("--partition", choices=["samples", "bands"], default="samples",
                    help="how N ranks share a frame: samples = every rank renders all pixels for a slice of the sample range; bands = every rank "
                         "renders all samples for the rows of its bands (rt_render_params::band_*), bands dealt out in turn")
    ap.add_argument("--band-rows", type=int, default=16, help="--partition bands: rows per band (a power of two)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the multi-rank path on a box with fewer GPUs than ranks (sums travel through host memory)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rtamd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to time)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    pipe = {"auto": rtamd.PIPELINE_AUTO, "fused": rtamd.PIPELINE_FUSED, "sorted": rtamd.PIPELINE_SORTED,
            "persistent": rtamd.PIPELINE_PERSISTENT}[args.pipeline]
    earth = os.path.join(REPO, "rust-raytracing_amd", "assets", "earthmap.ppm")
    t_scene = time.perf_counter()
    scene = rtamd.Scene(args.scene, width=args.width, aspect=args.aspect, spp=args.spp, depth=args.depth, earth_path=earth,
                        mega_grid=args.mega_grid, mega_view=1 if args.mega_view == "above" else 0)
    t_scene = time.perf_counter() - t_scene
    W, H = scene.width, scene.height
    n_objects = int(scene.desc.n_objects)
    ctx = rtamd.Context(local_rank)
    t_upload = time.perf_counter()
    ctx.upload(scene)
    t_upload = time.perf_counter() - t_upload
    if args.partition == "samples":
        s0, s1, bands = rank * args.spp // world, (rank + 1) * args.spp // world, None
    else:
        s0, s1, bands = 0, args.spp, (args.band_rows, world, rank)
    # N > 1: the sums travel through the library's own communicator (rt_comm_init / rt_reduce_accum, RCCL over xGMI); the id that
    # rank 0 creates reaches the other ranks through torch.distributed, which also provides the barriers of the timing contract
    lib_reduce = False
    if dist is not None and args.backend == "nccl" and args.reduce == "lib":
        # Every rank first finds out, on its own, whether the library can open RCCL (rt_comm_unique_id does nothing else that can fail),
        # and the ranks agree on the answer BEFORE any of them enters rt_comm_init: that call is a collective, and a rank that went
        # in alone would wait there for ever.
        ok, cid = 1, None
        try:
            cid = rtamd.comm_unique_id()
        except Exception as e:  # noqa: BLE001 - any failure means: use torch's reduce instead, on every rank
            ok = 0
            print("rank %d: the library cannot open RCCL (%s); falling back to torch.distributed.reduce" % (rank, e), file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if bool(flag.item()):
            box = [cid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.comm_init(box[0], world, rank)  # all ranks are in here together; a failure now is an error, not a fallback
            lib_reduce = True
    rgb_host = torch.zeros((H, W, 3), dtype=torch.uint8).pin_memory().numpy() if rank == 0 else None  # page-locked: the copy is one DMA
    persistent = args.pipeline in ("auto", "persistent")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(name, steps, warmup):
        """W warm-up frames, then `steps` timed frames of the whole job in arithmetic `name`; returns the figures of the leg
        (on every rank; `accum` holds the reduced sums on rank 0)."""
        prec = {"f64": rtamd.F64, "f64_own_tree": rtamd.F64_OWN_TREE, "f32": rtamd.F32}[name]
        accum = torch.zeros(H * W * 4, dtype=torch.float32 if name == "f32" else torch.float64, device=dev)
        # channel 3 of a pixel's slot is unused: the reduce carries the three colour sums only
        packed = torch.zeros(H * W * 3, dtype=accum.dtype, device=dev) if dist is not None else None
        split = {"render_s": 0.0, "reduce_s": 0.0, "resolve_s": 0.0}

        def step(timed):
            t_a = time.perf_counter()
            accum.zero_()
            torch.cuda.synchronize()  # the library renders on its own stream
            st = ctx.render_accum(accum.data_ptr(), s0, s1, precision=prec, pipeline=pipe, samples_per_pass=args.samples_per_pass, bands=bands)
            t_b = time.perf_counter()
            if dist is not None and lib_reduce:
                ctx.reduce_accum(accum.data_ptr(), W, H, precision=prec, root=0)  # W*H*3 values, once per frame, stream idle on return
            elif dist is not None:
                packed.copy_(accum.view(-1, 4)[:, :3].reshape(-1))
                if args.backend == "nccl":
                    dist.reduce(packed, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI: W*H*3 values, once per frame
                else:
                    host = packed.cpu()
                    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                    if rank == 0:
                        packed.copy_(host)
                if rank == 0:
                    accum.view(-1, 4)[:, :3] = packed.view(-1, 3)
                torch.cuda.synchronize()
            t_c = time.perf_counter()
            rgb = ctx.resolve(accum.data_ptr(), W, H, args.spp, precision=prec, out=rgb_host) if rank == 0 else None
            t_d = time.perf_counter()
            if timed:
                split["render_s"] += t_b - t_a
                split["reduce_s"] += t_c - t_b
                split["resolve_s"] += t_d - t_c
            return st, rgb

        for _ in range(warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        stats = []
        rgb = None
        for _ in range(steps):
            st, rgb = step(True)
            stats.append(st)
        barrier()
        elapsed = time.perf_counter() - t0
        rays = float(sum(s["rays"] for s in stats))
        kernel_ms = float(sum(s["bounce_kernel_ms"] for s in stats))
        launches = float(sum(s["bounce_kernel_launches"] for s in stats))
        rays_all, elapsed_max = rays, elapsed
        # split of a step (ms) on this rank: kernel, render call, reduce, resolve + copy
        per_rank = [[round(kernel_ms / steps, 3), round(1e3 * split["render_s"] / steps, 3), round(1e3 * split["reduce_s"] / steps, 3),
                     round(1e3 * split["resolve_s"] / steps, 3)]]
        if dist is not None:
            red_dev = dev if args.backend == "nccl" else torch.device("cpu")
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_max = float(t.item())
            r = torch.tensor([rays], dtype=torch.float64, device=red_dev)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            rays_all = float(r.item())
            # rank 0 reports every rank's split
            mine = torch.tensor([kernel_ms / steps, 1e3 * split["render_s"] / steps, 1e3 * split["reduce_s"] / steps,
                                 1e3 * split["resolve_s"] / steps], dtype=torch.float64, device=red_dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [[round(float(x), 3) for x in a.tolist()] for a in allr]
        # per-lane walks of trees that do not fit on chip: what a closest-hit query costs on the tree THIS precision walks (the f64
        # kernels walk the reference's tree, the f32 kernels the library's own), measured on the device over the rays of 8192 paths
        walk = None
        if rank == 0 and stats and stats[-1]["scene_in_lds"] in (4, 6, 8, 9, 10, 11, 12) and n_objects >= 100_000:
            walk = ctx.walk_cost(prec, n_paths=8192, seed=1)
        return {"name": name, "walk": walk, "elapsed": elapsed_max, "rays": rays, "rays_all": rays_all, "kernel_ms": kernel_ms, "launches": launches,
                "stats": stats, "rgb": None if rgb is None else rgb.copy(), "accum": accum, "split": split, "per_rank": per_rank}

    scene_key = args.scene + ("-above" if args.scene == "mega-spheres" and args.mega_view == "above" else "")

    def roofline_of(leg, fpr):
        name = "f32" if leg["name"] == "f32" else "f64"  # the arithmetic; leg["name"] also tells the tree
        if fpr is None or leg["kernel_ms"] <= 0:
            return None
        stats = leg["stats"]
        achieved = fpr * leg["rays"] / (leg["kernel_ms"] * 1e-3) / 1e12
        peak = VALU_PEAK_TFLOPS[name]
        rays_per_sample = leg["rays_all"] / max(1, args.steps) / (W * H * args.spp)
        traffic = None
        issue = None
        try:  # counters of the same kernel on this frame, collected by tools/profile_round.sh (separate rocprofv3 passes)
            with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get("%s_%s_%s" % (scene_key, leg["name"], "persistent" if persistent else args.pipeline))
            if t is not None:  # bytes per frame at the profiled sample count -> per launch of this run
                traffic = t["bytes_per_frame"] * (args.spp / t["spp"]) * (leg["rays"] / max(1.0, leg["rays_all"])) / (leg["launches"] / max(1, args.steps))
                with open(os.path.join(REPO, "profiles", "%s.json" % t["from"])) as f:
                    k = json.load(f)["kernels"]
                k = k.get("k_paths_persistent") or k.get("k_paths_stepwise") or {}
                if persistent and "valu_insts_per_simd_cycle" in k:
                    issue = {"valu_insts_per_simd_cycle": k["valu_insts_per_simd_cycle"], "salu_insts_per_cu_cycle": k.get("salu_insts_per_cu_cycle"),
                             "lanes_active_of_64": k.get("lanes_active_per_valu_cycle"),
                             "source": "profiles/%s.json" % t["from"]}
        except (OSError, KeyError, ValueError):
            pass
        # bytes the algorithm has to move per ray: the queue pipelines read and write one queue entry per ray; the persistent
        # pipeline keeps paths in registers and only adds finished radiance terms to the pixel sums (4 values per path)
        qb = (4 * (8 if name == "f64" else 4)) / rays_per_sample if persistent else QUEUE_BYTES_PER_RAY[name]
        kernel = (("k_paths_stepwise" if stats[-1]["scene_in_lds"] in (5, 6, 7, 8, 9, 10, 11, 12) else "k_paths_persistent") if persistent
                  else ("k_bounce_fused" if args.pipeline == "fused" else "k_trace+k_shade"))
        secs = leg["kernel_ms"] * 1e-3
        valu = {"bound": "valu", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "flops_per_ray": fpr}
        # Scenes whose tree does not fit on chip (presets 4, 6, 8: only its top levels are staged in LDS; 9, 10: the four-wide walk,
        # nothing staged): the walk is priced against the memory system.  Algorithmic bytes: 32 per box tested (a four-wide node
        # is four boxes, 128 bytes) and per primitive tested - counted on the device, on the tree this precision's kernels walk,
        # over the rays of a sample of the frame's paths - plus the radiance sums.
        in_memory = leg["walk"] is not None
        bpr = (32.0 * (leg["walk"][0] + leg["walk"][1]) + qb) if in_memory else qb
        hbm = {"bound": "hbm", "achieved": bpr * leg["rays"] / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": bpr * leg["rays"] / secs / 1e9 / HBM_PEAK_GBS, "bytes_per_ray": bpr}
        if in_memory:
            # The lanes ask for `bpr` bytes per ray, and most of that is served by L2 and Infinity Cache; what reaches HBM is what the
            # counters of the same kernel on the same frame saw (FETCH_SIZE, corrected, + WRITE_SIZE), and that - a fraction of the
            # HBM peak that cannot exceed 1 - is what is priced here.  The request volume stays beside it.
            hbm["requested_bytes_per_ray"] = bpr
            hbm["requested_GBs"] = hbm["achieved"]
            if traffic is not None:
                per_ray = traffic * leg["launches"] / max(1.0, leg["rays"])
                hbm.update({"achieved": per_ray * leg["rays"] / secs / 1e9, "frac": per_ray * leg["rays"] / secs / 1e9 / HBM_PEAK_GBS,
                            "bytes_per_ray": per_ray, "source": "counter bytes of profiles/pmc_traffic.json scaled to this run's rays / this run's kernel time"})
            else:
                hbm.update({"achieved": None, "frac": None, "bytes_per_ray": None, "source": "no counter profile of this frame under profiles/"})
        # the walk of a tree in memory is priced against HBM only where the counters show HBM carrying a fifth of its peak or more;
        # below that the vector unit is the binding resource there too
        hbm_bound = in_memory and hbm["frac"] is not None and hbm["frac"] >= 0.2
        main, other = (hbm, valu) if hbm_bound else (valu, hbm)
        out = dict(main)
        if in_memory:
            out["walk"] = {"box_tests_per_ray": leg["walk"][0], "leaf_tests_per_ray": leg["walk"][1], "rays_sampled": leg["walk"][2]}
        out.update({
            "traffic": traffic, "kernel": kernel, "avg_launch_ms": leg["kernel_ms"] / max(1.0, leg["launches"]), "launches": leg["launches"],
            "issue": issue,
            "note": ("the tree (tens of MB) lives in HBM / L2 / Infinity Cache and every lane fetches its own nodes; hbm = bytes that reached "
                     "HBM by the counters against the 8 TB/s peak (requested_* = what the lanes asked the caches for); valu = the same "
                     "launches against the vector-ALU roofline" if in_memory else
                     "no dense contraction on this path (MFMA unused by design) and the scene is cache resident, so the vector ALU bounds it; "
                     "peak is the vector FMA peak of the dtype; hbm = the same launches against the HBM roofline"),
            ("valu" if hbm_bound else "hbm"): other,
        })
        return out

    head_name = "f32" if args.precision == "f32" else "f64"
    head = run_leg(head_name, args.steps, args.warmup)
    side = run_leg("f32", args.steps, args.warmup) if args.precision == "both" else None
    # scenes with boxes, wrappers or media on the library's four-wide tree have a second double-precision image with cheaper leaf
    # forms (RT_F64_OWN_TREE: boxes as slabs, re-centred spheres; compared with the reference by tolerance)
    own = None
    if args.precision == "both" and head["stats"] and head["stats"][-1]["scene_in_lds"] in (3, 4, 7, 8, 12):
        own = run_leg("f64_own_tree", args.steps, args.warmup)

    if rank == 0:
        headline = (args.scene, args.width, args.spp) == ("cornell-box", 1920, 1024)
        out = {
            "metric": "Mrays/s cornell_box 1080p 1024spp" if headline else "Mrays/s %s %dx%d %dspp" % (args.scene, W, H, args.spp),
            "value": head["rays_all"] / head["elapsed"] / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * head["elapsed"] / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": head_name,
            "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp, depth %d, %s, one reduce to rank 0" % (
                           scene_key, W, H, args.spp, args.depth,
                           "samples sharded over ranks" if args.partition == "samples" else "rows sharded over ranks in bands of %d" % args.band_rows),
                       "partition": args.partition if args.partition == "samples" else "bands of %d rows" % args.band_rows,
                       # what "f64" means here: IEEE double operations throughout; the small-scene kernels (flat walk) are compiled with
                       # -ffp-contract=fast (a*b+c may fuse into one rounding) and take reciprocals and the Lambertian unit vector by
                       # v_rcp / v_rsq + Newton steps (<= 11 / 19 ulp) where the reference divides and takes square roots
                       "arithmetic": {"f64": "IEEE double; fp-contract=fast and refined rcp/rsq (<= 19 ulp) in the flat-walk kernels, fp-contract as written "
                                             "and true quotients in the leaf tests of the four-wide walk", "f32": "tolerance mode"},
                       "pipeline": args.pipeline, "rays_per_step": head["rays_all"] / max(1, args.steps),
                       "rays_per_camera_sample": head["rays_all"] / max(1, args.steps) / (W * H * args.spp),
                       "lds_bytes": head["stats"][-1]["lds_bytes"], "scene_in_lds": head["stats"][-1]["scene_in_lds"],
                       "scene_build_s": t_scene, "upload_s": t_upload},
        }
        if world > 1:
            out["config"]["reduce"] = "rt_reduce_accum (RCCL)" if lib_reduce else ("torch.distributed.reduce (%s)" % args.backend)
        out["per_rank_ms"] = {"columns": ["kernel", "render_call", "reduce", "resolve_and_copy"], "ranks": head["per_rank"]}
        cpu = None
        cnt = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(REPO, "tests"))
            import oracle_lib  # the checker, used here only as the timed CPU baseline
            share = cpu_share()
            threads = max(1, share - 1)  # num_cpus - 1 worker threads (src/render.rs:184) of the cores this job may use
            earth_rgb = None
            if args.scene in ("earth", "final-scene"):
                earth_rgb = rtamd.load_ppm(earth)
            osc = oracle_lib.OracleScene(args.scene, width=args.width, aspect=args.aspect, spp=args.spp, depth=args.depth, earth=earth_rgb,
                                         mega_grid=args.mega_grid, mega_view=1 if args.mega_view == "above" else 0)
            _, cnt, secs = osc.render(0, args.cpu_spp, seed=0, threads=threads)
            cpu = {"value": cnt["rays"] / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": "same scene and frame, samples 0..%d of %d per pixel (%d rays), C++ f64 restatement of the reference with its "
                             "row-block schedule on num_cpus-1 threads; the Rust binary cannot be built here" % (args.cpu_spp, args.spp, cnt["rays"]),
                   "seconds": secs, "cpu_model": cpu_model(), "cpus_usable": share, "cpus_machine": os.cpu_count()}
            out["vs_cpu_baseline"] = out["value"] / cpu["value"]
        out["cpu_baseline"] = cpu
        # Roofline of the dominant kernel (the one launch of the path kernel, or the per-bounce kernels of the queue pipelines):
        # algorithmic flops per ray x rays / kernel time measured by the library with HIP events on its own stream.
        fpr = flops_per_ray(cnt) if cnt else None
        if fpr is None:
            try:
                with open(os.path.join(REPO, "profiles", "flops_per_ray.json")) as f:
                    fpr = json.load(f).get(args.scene)
            except OSError:
                fpr = None
        out["roofline"] = roofline_of(head, fpr)
        if side is not None:
            # error of the f32 frame against the f64 frame of this run, both at the full sample count: linear radiance per pixel
            # (sum / spp) clamped to [0, 1] as color_to_rgb does before its gamma (src/color.rs:20-25)
            a64 = (head["accum"].view(-1, 4)[:, :3] / args.spp).clamp(0.0, 1.0)
            a32 = (side["accum"].view(-1, 4)[:, :3].double() / args.spp).clamp(0.0, 1.0)
            m64, m32 = a64.mean(0), a32.mean(0)
            diff8 = np.abs(head["rgb"].astype(np.int16) - side["rgb"].astype(np.int16))
            out["f32"] = {
                "value": side["rays_all"] / side["elapsed"] / 1e6, "unit": "Mrays/s", "dtype": "f32",
                "ms_per_step": 1e3 * side["elapsed"] / max(1, args.steps), "steps": args.steps, "warmup": args.warmup,
                "roofline": roofline_of(side, fpr),
                "image_error_vs_f64": {
                    "samples_per_pixel": args.spp,
                    "mean_f64": [float(x) for x in m64], "mean_f32": [float(x) for x in m32],
                    "mean_error_rel": [float(x) for x in ((m32 - m64).abs() / m64.clamp_min(1e-12))],
                    "rmse": [float(x) for x in ((a32 - a64) ** 2).mean(0).sqrt()],
                    "rgb8_max_abs_diff": int(diff8.max()), "rgb8_frac_bytes_differing": float((diff8 > 0).mean()),
                    "rays_rel_diff": abs(side["rays_all"] - head["rays_all"]) / head["rays_all"],
                    "note": "per channel, linear radiance clamped to [0,1]; both frames use the same random streams, so the difference is "
                            "rounding plus the paths that part where single precision resolves an edge differently",
                },
            }
        if own is not None:
            b64 = (own["accum"].view(-1, 4)[:, :3] / args.spp).clamp(0.0, 1.0)
            a64 = (head["accum"].view(-1, 4)[:, :3] / args.spp).clamp(0.0, 1.0)
            d8 = np.abs(head["rgb"].astype(np.int16) - own["rgb"].astype(np.int16))
            out["f64_own_tree"] = {
                "value": own["rays_all"] / own["elapsed"] / 1e6, "unit": "Mrays/s", "dtype": "f64",
                "ms_per_step": 1e3 * own["elapsed"] / max(1, args.steps), "steps": args.steps, "warmup": args.warmup,
                "roofline": roofline_of(own, fpr),
                "image_error_vs_f64": {
                    "mean_error_rel": [float(x) for x in ((b64.mean(0) - a64.mean(0)).abs() / a64.mean(0).clamp_min(1e-12))],
                    "rmse": [float(x) for x in ((b64 - a64) ** 2).mean(0).sqrt()],
                    "rgb8_max_abs_diff": int(d8.max()), "rgb8_frac_bytes_differing": float((d8 > 0).mean()),
                    "rays_rel_diff": abs(own["rays_all"] - head["rays_all"]) / head["rays_all"]},
                "note": "RT_PRECISION_F64 | RT_F64_OWN_TREE: the library's tree with its cheaper leaf forms (a box as three slabs, spheres of a "
                        "dissolved nested tree re-centred) and without the second tracing of rays that meet an exact tie (rt_stats.rays_retraced)",
            }
        if side is not None:
            if True:
                out["f32"]["per_rank_ms"] = {"columns": ["kernel", "render_call", "reduce", "resolve_and_copy"], "ranks": side["per_rank"]}
        print(json.dumps(out), flush=True)
        if args.save and head["rgb"] is not None:
            rtamd.save_image(args.save, head["rgb"])
        if args.dump_accum:
            np.save(args.dump_accum, head["accum"].view(H, W, 4)[:, :, :3].double().cpu().numpy())
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
