#!/usr/bin/env python3
"""Benchmark of the render loop: one step = one full frame of the workload, image tiles split
over the ranks (one process per GPU), framebuffer assembled on rank 0 from an RCCL all-gather.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synthetic|scene7_4k|scene1|...]

Default workload: the largest configuration, the synthetic scene (1 M triangles + 10 k ellipsoids, 8 point lights,
3840x2160, maxdepth 2).  Prints one JSON line on rank 0:
* `value` = ray segments per second of the timed frames (default mode of the library: bounding hierarchy + exact tests);
* `roofline` = a BRUTE-FORCE leg on a centred crop of the same frame (RT_FLAG_NO_FILTER: every ray against every primitive,
  launches one after the other on one stream): ray-triangle pairs x 54 FLOP / HIP-event time of the closest-hit triangle
  scan launches, against the FP32 vector peak; the shadow scan beside it;
* `culled_mode` = what the default mode executed (box tests, pre-tests, full tests) during the timed frames;
* `cpu_baseline` = the CPU renderer on a small crop of the same frame (N = 1 only).
For N > 1 launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per ray-primitive test: FMA = 2, add/sub/mul/div/sqrt = 1, compare/select = 0
FLOP_TRI, FLOP_SPH, FLOP_ELL_CLOSEST, FLOP_ELL_SHADOW_POINT, FLOP_ELL_SHADOW_DIR = 54, 19, 95, 71, 62
PEAK_FP32_TFLOPS = 157.3   # MI355X FP32 vector peak: 256 CU x 4 SIMD x 32 lanes x 2 FLOP x 2.4 GHz
TILE_BYTES = 768
# centred crops (WxH): the brute-force leg, and the CPU baseline (sized for 10-30 s of CPU work on 16 threads).
# The brute-force crop has to be large enough for its launches to fill the chip many times over, or the last workgroups
# of a launch weigh on the figure: synthetic scene, closest-hit scan, 256x144: 0.387 of the peak, 512x288: 0.478,
# 768x432: 0.493, 1024x576: 0.495 (33 s and 59 s of device time for the last two).
BRUTE_CROP = {"synthetic": "768x432", "default": "1024x576"}
CPU_CROP = {"synthetic": "64x36", "default": "240x135"}
# how the generated scene departs from a plain reading of its specification (raytracer_amd/synth.py)
SYNTHETIC_NOTES = ("seeded numpy PCG64 streams (triangles seed 1, ellipsoids 2, materials 3); light colour .15 and diffuse/specular U[0.1,0.5] "
                   "so that no channel reaches 1.0 with 8 lights (the reference's 8-bit conversion is undefined above); lights on a ring of radius 25 at height 8")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synthetic")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--maxdepth", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-brute-force-leg", action="store_true")
    ap.add_argument("--chunk-tiles", type=int, default=0, help="tiles per chunk of ray state (0 = library default)")
    ap.add_argument("--brute-force", action="store_true", help="time the frames themselves with RT_FLAG_NO_FILTER (slow: for small sizes)")
    ap.add_argument("--plain-test", action="store_true", help="brute-force scans use RT_FLAG_PLAIN_TEST: the triangle test exactly as written (three IEEE divisions per pair)")
    ap.add_argument("--no-as-written", action="store_true", help="skip the part of the brute-force leg that runs the tests exactly as written on a quarter of the crop")
    ap.add_argument("--brute-crop", default="", help="centred crop of the brute-force leg (WxH)")
    ap.add_argument("--cpu-crop", default="", help="centred crop the CPU baseline renders (WxH)")
    return ap.parse_args()


def load_workload(api, name, width, height, maxdepth):
    """Returns (FileInfo, label, seconds the parser took)."""
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    if name == "synthetic":
        from raytracer_amd import synth
        path = synth.ensure_scene_file()        # generated on first use (not timed)
    else:
        path = os.path.join(scenes, (name.replace("_4k", "")) + ".test")
    t0 = time.perf_counter()
    fi = api.FileInfo(path)
    parse_s = time.perf_counter() - t0
    if name == "scene7_4k":
        fi.set_size(width or 3840, height or 2160)
    elif width and height:
        fi.set_size(width, height)
    label = "%s@%dx%d" % (name.replace("_4k", ""), fi.width, fi.height)
    return fi, label, parse_s


def centred(fi, crop):
    cw, ch = [int(v) for v in crop.split("x")]
    cw, ch = min(cw, fi.width), min(ch, fi.height)
    return dict(rect_x0=(fi.width - cw) // 2, rect_y0=(fi.height - ch) // 2, rect_w=cw, rect_h=ch), cw, ch


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # not under a launcher: one process drives all the GPUs (rt_init_devices / rt_render_multi_begin / _end); with one GPU that
    # path is taken only when RT_AMD_COMM_ALWAYS is set (it then exchanges through RCCL all the same: a rehearsal of the N-GPU step)
    single_process = world == 1 and (args.gpus > 1 or bool(os.environ.get("RT_AMD_COMM_ALWAYS")))
    if world != args.gpus and not single_process:
        args.gpus = world

    # The line of results is the only thing this program writes to its standard output: libraries that
    # print there (RCCL announces its version on rank 0) are sent to standard error instead.
    sys.stdout.flush()
    results_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, api, distributed

    # Rehearsal on a box with one GPU: RT_BENCH_REHEARSAL=1 puts every rank on device 0 and exchanges through gloo
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing, its frame check does.
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if single_process:
        return single_process_bench(args, api, abi, results_fd)
    exchange = "none"
    if world > 1:
        # torch.distributed is the launcher's rendezvous and the control plane only (barriers, the maximum of the ranks' times,
        # handing out the communicator's identifier): gloo, on the host.  The tile buffers travel through the library's own RCCL communicator.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    api.init(local_rank)
    if world > 1 and not rehearsal:
        ok, uid = 1, None
        if rank == 0:
            try:
                uid = api.comm_unique_id()
            except Exception as e:   # reported, never silent: the line of results names the exchange that ran
                sys.stderr.write("rank 0: RCCL identifier failed (%s)\n" % e)
        ids = [uid]
        dist.broadcast_object_list(ids, src=0)
        try:
            if ids[0] is None:
                raise RuntimeError("rank 0 could not create the identifier")
            api.comm_init_rank(ids[0], rank, world)
            exchange = "library RCCL communicator: grouped send/receive to rank 0 (rt_comm_gather_tiles)"
        except Exception as e:
            sys.stderr.write("rank %d: library communicator failed (%s)\n" % (rank, e))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            api.comm_destroy()
            nccl_group = dist.new_group(backend="nccl")
            exchange = "torch.distributed all_gather_into_tensor over RCCL (the library communicator could not be created)"
    elif world > 1:
        exchange = "gloo on host copies (rehearsal on one GPU)"

    if args.workload == "synthetic" and world > 1:
        # one rank writes the scene file, the others wait for it
        if rank == 0:
            from raytracer_amd import synth
            synth.ensure_scene_file()
        dist.barrier()
    use_library_comm = exchange.startswith("library")
    fi, label, parse_s = load_workload(api, args.workload, args.width, args.height, args.maxdepth)
    depth = fi.max_recursion_depth if args.maxdepth < 0 else args.maxdepth
    cam = fi.camera()
    t_setup = time.perf_counter()
    scene = api.Scene(fi.scene_desc)     # upload, re-packing on the GPU, bounding hierarchy (kd ordering on the host)
    scene_create_s = time.perf_counter() - t_setup
    flags = abi.RT_FLAG_NO_FILTER if args.brute_force else 0
    if args.plain_test:
        flags |= abi.RT_FLAG_PLAIN_TEST
    prm = fi.params(max_bounce_count=depth, tile_rank=rank, tile_world=world, flags=flags, chunk_tiles=args.chunk_tiles)
    stride = distributed.stride_tiles(prm, world)      # tiles per rank in the gathered buffer
    assert api.rank_tile_count(prm) == len(distributed.rank_tiles(prm, rank, world)) <= stride

    # Two tile buffers: while the ranks exchange the tiles of frame k (one RCCL all-gather over xGMI,
    # asynchronous on torch's stream), frame k+1 is already being traced into the other buffer.
    tiles = [torch.zeros(stride * TILE_BYTES, dtype=torch.uint8, device=device) for _ in range(2)]
    gathered = [torch.zeros(world * stride * TILE_BYTES, dtype=torch.uint8, device=device) if world > 1 else tiles[i] for i in range(2)]
    exchanged = [None, None]
    frame = torch.zeros(fi.height * fi.width * 3, dtype=torch.uint8, device=device) if rank == 0 else None
    torch.cuda.synchronize()      # the library writes these buffers from streams of its own

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(k):
        """Frame k: wait for its exchange, then rank 0 scatters the gathered tiles into the image."""
        if use_library_comm:
            api.comm_wait()
        elif exchanged[k % 2] is not None:
            exchanged[k % 2].synchronize()
        if rank == 0:
            api.assemble_tiles_device(gathered[k % 2].data_ptr(), stride, prm, frame.data_ptr())

    def run(n):
        stats = []
        for k in range(n):
            stats.append(scene.render_tiles_device(cam, prm, tiles[k % 2].data_ptr(), want_stats=True))   # returns with its streams idle
            if k:
                finish(k - 1)
            if use_library_comm:
                api.comm_gather_tiles(tiles[k % 2].data_ptr(), stride, gathered[k % 2].data_ptr())     # enqueued on the library's exchange stream
            elif world > 1 and rehearsal:
                host = [torch.zeros(stride * TILE_BYTES, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(host, tiles[k % 2].cpu())
                gathered[k % 2].copy_(torch.cat(host))
            elif world > 1:
                dist.all_gather_into_tensor(gathered[k % 2], tiles[k % 2], group=nccl_group)
                exchanged[k % 2] = torch.cuda.Event()
                exchanged[k % 2].record()
        if n:
            finish(n - 1)
        return stats

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    stats = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-rank work counters of the timed steps, summed over ranks
    fields = ["closest_rays", "shadow_rays", "samples", "culled", "hits", "shadow_blocked", "full_tests"]
    local = [float(sum(getattr(s, f) for s in stats)) for f in fields]
    local += [float(sum(s.closest_tests[k] for s in stats)) for k in range(3)]
    local += [float(sum(s.shadow_tests[k] for s in stats)) for k in range(3)]
    local += [float(sum(s.pair_pretests[k] for s in stats)) for k in range(2)]
    local += [float(sum(s.queued_pairs[k] for s in stats)) for k in range(2)]
    local += [float(sum(s.executed_flop[k] for s in stats)) for k in range(2)]
    local += [sum(s.closest_ms for s in stats), sum(s.shadow_ms for s in stats), sum(s.device_ms for s in stats)]
    local += [float(sum(s.closest_launches for s in stats)), float(sum(s.shadow_launches for s in stats))]
    acc = torch.tensor(local, dtype=torch.float64)
    mx = acc.clone()
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    tot = dict(zip(fields + ["ct_tri", "ct_sph", "ct_ell", "st_tri", "st_sph", "st_ell", "pp_closest", "pp_shadow", "q_closest", "q_shadow", "xf_closest", "xf_shadow", "closest_ms", "shadow_ms", "device_ms",
                             "closest_launches", "shadow_launches"], acc.tolist()))
    rank_max = dict(zip(["closest_ms", "shadow_ms", "device_ms"], mx.tolist()[19:22]))

    if rank == 0:
        rays = tot["closest_rays"] + tot["shadow_rays"]
        pairs = tot["ct_tri"] + tot["ct_sph"] + tot["ct_ell"] + tot["st_tri"] + tot["st_sph"] + tot["st_ell"]
        executed_tests = tot["pp_closest"] + tot["pp_shadow"] + tot["q_closest"] + tot["q_shadow"] + tot["full_tests"]
        out = {
            "metric": "Mrays/s",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.workload == "synthetic" else "reference scene file (tests/golden/scenes), size overridden",
            "config": {"workload": label, "maxdepth": depth, "samples_per_pixel": 16,
                       "primitives": {"triangles": int(fi.scene_desc.n_triangles), "spheres": int(fi.scene_desc.n_spheres),
                                      "ellipsoids": int(fi.scene_desc.n_ellipsoids)},
                       "lights": int(fi.scene_desc.n_point_lights) + int(fi.scene_desc.has_directional_light),
                       "parallelism": "tiles%%%d" % world, "exchange": exchange,
                       "mode": "brute force (RT_FLAG_NO_FILTER)" if args.brute_force else "bounding hierarchy + conservative pre-tests, exact test on the survivors"},
            "rays_per_step": rays / args.steps,
            "culled_samples_per_step": tot["culled"] / args.steps,
            "per_rank_max_ms": {k: v / args.steps for k, v in rank_max.items()},
            "setup_ms": {"parse_scene_file": parse_s * 1e3, "scene_create": scene_create_s * 1e3},   # rank 0, outside the timed steps
        }
        if args.workload == "synthetic":
            out["config"]["generator"] = SYNTHETIC_NOTES
        if not args.brute_force:
            out["culled_mode"] = {
                "what": "work of the timed frames: the hierarchy decides most ray-primitive pairs by bounds; executed = per-ray box tests + leaf pre-tests / tests + full triangle tests",
                "executed_tests_per_s": executed_tests / elapsed,
                "executed_tests_per_ray": executed_tests / max(rays, 1.0),
                "executed_tflops": (tot["xf_closest"] + tot["xf_shadow"]) / elapsed / 1e12 / world,
                "executed_frac_of_fp32_peak": (tot["xf_closest"] + tot["xf_shadow"]) / elapsed / 1e12 / world / PEAK_FP32_TFLOPS,
                "pairs_decided_by_bounds": 1.0 - (tot["q_closest"] + tot["q_shadow"]) / max(pairs, 1.0),
                "equivalent_ray_prim_pairs_per_s": pairs / elapsed,
                "equivalent_note": "rays x primitives a brute-force renderer would visit for these frames, per second of this renderer; not executed work",
                "full_triangle_tests_per_step": tot["full_tests"] / args.steps,
            }
            # the frame the ranks assembled in the last step against one rendered by this rank alone (outside the timed steps)
            alone = scene.render(cam, fi.params(max_bounce_count=depth, flags=flags))
            out["frame_identical_to_single_rank_render"] = bool((alone["rgb"].reshape(-1) == frame.cpu().numpy()).all())
        if not args.brute_force:
            out["output_ms"] = output_stage(api, scene, cam, prm, tiles[0], frame, fi, torch, world)
        if args.brute_force and world == 1:
            # the timed frames themselves were brute force: the table of the scans from their own launches (whole frames, not a crop)
            out["roofline"], out["brute_force"] = whole_frame_roofline(stats, fi, args, label, depth, elapsed)
            fast = scene.render(cam, fi.params(max_bounce_count=depth))
            out["frame_identical_to_default_mode"] = bool((fast["rgb"].reshape(-1) == frame.cpu().numpy()).all())
            out["counters_identical_to_default_mode"] = all(getattr(stats[-1], f) == getattr(fast["stats"], f) for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked"))
        elif not args.no_brute_force_leg:
            out["roofline"], out["brute_force"] = brute_force_leg(abi, fi, cam, scene, depth, args, label)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(api, abi, fi, cam, scene, depth, args.cpu_crop or CPU_CROP.get(args.workload, CPU_CROP["default"]))
        sys.stdout.flush()
        os.write(results_fd, (json.dumps(out) + "\n").encode())

    scene.close()
    if world > 1:
        dist.barrier()
        api.comm_destroy()
        dist.destroy_process_group()


def single_process_bench(args, api, abi, results_fd):
    """--gpus N without a launcher: this one process drives N GPUs through the C ABI alone (rt_init_devices, rt_scene_create
    replicating the scene, rt_render_multi_begin / _end: one persistent host thread per device, RCCL gather to device 0).
    A step is what a step of the launcher's path is: every device traces its share of frame k while the tiles of frame k - 1
    travel and are scattered into the image, which stays in the memory of device 0; nothing is copied to the host."""
    import torch
    n = args.gpus
    api.init_devices(n)
    fi, label, parse_s = load_workload(api, args.workload, args.width, args.height, args.maxdepth)
    depth = fi.max_recursion_depth if args.maxdepth < 0 else args.maxdepth
    cam = fi.camera()
    t0 = time.perf_counter()
    scene = api.Scene(fi.scene_desc)
    scene_create_s = time.perf_counter() - t0
    prm = fi.params(max_bounce_count=depth, chunk_tiles=args.chunk_tiles)
    images = [torch.zeros(fi.height * fi.width * 3, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()

    def run(steps):
        done, ticket = [], None
        for k in range(steps):
            now = api.render_multi_begin(scene, cam, prm, images[k % 2].data_ptr())
            if ticket is not None:
                done.append(api.render_multi_end(ticket, n))
            ticket = now
        if ticket is not None:
            done.append(api.render_multi_end(ticket, n))
        return done

    run(args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = run(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rays = float(sum(st.closest_rays + st.shadow_rays for stats, _ in frames for st in stats))
    per_device_ms = [sum(stats[d].device_ms for stats, _ in frames) / args.steps for d in range(n)]
    out = {"metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic" if args.workload == "synthetic" else "reference scene file (tests/golden/scenes), size overridden",
           "config": {"workload": label, "maxdepth": depth, "samples_per_pixel": 16, "parallelism": "tiles%%%d" % n,
                      "primitives": {"triangles": int(fi.scene_desc.n_triangles), "spheres": int(fi.scene_desc.n_spheres), "ellipsoids": int(fi.scene_desc.n_ellipsoids)},
                      "lights": int(fi.scene_desc.n_point_lights) + int(fi.scene_desc.has_directional_light),
                      "mode": "bounding hierarchy + conservative pre-tests, exact test on the survivors",
                      "exchange": ("REHEARSAL, %d device contexts on ONE GPU (RT_AMD_DEVICES_ON_ONE_GPU): gather by device-to-device copies; not a multi-GPU measurement" % n)
                                  if n > 1 and os.environ.get("RT_AMD_DEVICES_ON_ONE_GPU") else
                                  "one process, rt_render_multi_begin / _end: RCCL grouped send/receive to device 0 and scatter into the image on the exchange streams, "
                                  "behind the next frame; image left in device memory" if n > 1 or os.environ.get("RT_AMD_COMM_ALWAYS") else "one device: no exchange, scatter into the image only"},
           "rays_per_step": rays / max(args.steps, 1), "per_device_ms": per_device_ms,
           "exchange_ms": sum(ms for _, ms in frames) / max(args.steps, 1),
           "setup_ms": {"parse_scene_file": parse_s * 1e3, "scene_create_all_devices": scene_create_s * 1e3}}
    # the frame the devices assembled in the last step against one rendered by device 0 alone (outside the timed steps)
    alone = scene.render(cam, fi.params(max_bounce_count=depth))
    last = images[(args.steps - 1) % 2].cpu().numpy() if args.steps else None
    out["frame_identical_to_single_device_render"] = bool((alone["rgb"].reshape(-1) == last).all()) if last is not None else None
    scene.close()
    sys.stdout.flush()
    os.write(results_fd, (json.dumps(out) + "\n").encode())


def output_stage(api, scene, cam, prm, tiles, frame, fi, torch, world):
    """The stage after the path (Source/main.cpp:93): framebuffer to the host, PNG encoding and writing; alone, and behind the
    next frame (rt_write_png_begin / rt_write_png_end while this rank traces its share again).  Rank 0, outside the timed steps."""
    import tempfile
    t0 = time.perf_counter()
    host = frame.cpu().numpy().reshape(fi.height, fi.width, 3)
    d2h = time.perf_counter() - t0
    path = os.path.join(tempfile.gettempdir(), "rt_bench_frame_%d.png" % os.getpid())
    t0 = time.perf_counter()
    api.write_png(path, host)
    png = time.perf_counter() - t0
    size = os.path.getsize(path)
    t0 = time.perf_counter()
    scene.render_tiles_device(cam, prm, tiles.data_ptr(), want_stats=False)
    alone = time.perf_counter() - t0
    t0 = time.perf_counter()
    api.write_png_begin(path, host)
    scene.render_tiles_device(cam, prm, tiles.data_ptr(), want_stats=False)
    api.write_png_end()
    both = time.perf_counter() - t0
    os.remove(path)
    return {"framebuffer_to_host": d2h * 1e3, "png_encode_and_write": png * 1e3, "png_bytes": size, "raw_bytes": host.size,
            "frame_share_alone": alone * 1e3, "frame_share_with_png_in_background": both * 1e3,
            "note": "PNG: Sub filter, zlib level 3 in up to 16 pieces on host threads; pixels identical to the framebuffer (tests/test_integration_binding.py)"}


KINDS = ("triangle", "sphere", "ellipsoid")
CLOSEST_KERNELS = {False: ("k_closest_tri_brute<2,2>", "k_closest_sph_eye<2> (rays from the eye) / k_closest_sph<2>", "k_closest_ell_staged<2>"),
                   True: ("k_closest_tri_brute<1,0>", "k_closest_sph<2>", "k_closest_ell<2>")}
SHADOW_KERNELS = {False: ("k_shadow_tri_pretest<2> (long launches) / <1>", "k_shadow_brute<2,2>", "k_shadow_brute<2,1>"),
                  True: ("k_shadow_tri_brute<0>", "k_shadow_sph<2>", "k_shadow_ell<2>")}
# FLOP a lane executes per stage of the default form of the triangle scans (FMA = 2): the two dot products and a - o (13; d.n alone,
# 5, for rays of one origin, whose other parts are per triangle), the first pre-test (17), the second (8), the test as written (68)
STAGE_FLOP = {"front": 13, "front_one_origin": 5, "stage1": 17, "stage2": 8, "as_written": 68}


def kernel_table(st, fi, plain):
    """Per primitive kind and scan: counted ray-primitive pairs, HIP-event time of the launches, reference-equivalent FLOP
    (pairs x the FLOP of the reference's test with the ray-independent parts hoisted: 54 / 19 / 95, shadow ellipsoid 71 toward
    a point light and 62 toward the directional one, weighted by the number of lights of each sort) and the share of the FP32 peak."""
    n_point, n_dir = int(fi.scene_desc.n_point_lights), int(fi.scene_desc.has_directional_light)
    ell_shadow = (FLOP_ELL_SHADOW_POINT * n_point + FLOP_ELL_SHADOW_DIR * n_dir) / max(n_point + n_dir, 1)
    flop = {"closest": (FLOP_TRI, FLOP_SPH, FLOP_ELL_CLOSEST), "shadow": (FLOP_TRI, FLOP_SPH, ell_shadow)}
    rows, total_flop, total_ms, total_pairs = {}, 0.0, 0.0, 0.0
    for scan, pairs_of, ms_of, launches_of, names in (("closest", st.closest_tests, st.closest_kind_ms, st.closest_kind_launches, CLOSEST_KERNELS[plain]),
                                                       ("shadow", st.shadow_tests, st.shadow_kind_ms, st.shadow_kind_launches, SHADOW_KERNELS[plain])):
        for k, kind in enumerate(KINDS):
            pairs, ms, launches = float(pairs_of[k]), float(ms_of[k]), int(launches_of[k])
            if launches == 0:
                continue
            tflops = pairs * flop[scan][k] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            rows["%s_%s" % (scan, kind)] = {
                "kernel": names[k], "pairs": pairs, "flop_per_pair": flop[scan][k], "ms": ms, "launches": launches, "avg_launch_ms": ms / launches,
                "flop_per_launch": pairs * flop[scan][k] / launches, "pairs_per_s": pairs / (ms * 1e-3) if ms > 0 else 0.0,
                "achieved": tflops, "frac": tflops / PEAK_FP32_TFLOPS}
            total_flop += pairs * flop[scan][k]
            total_ms += ms
            total_pairs += pairs
    # what the triangle scans executed (default form): lanes x FLOP of the stages their wavefronts went through
    for scan, index, lanes in (("closest", 0, 64), ("shadow", 1, 64)):       # the library counts the steps in units of 64 pairs
        steps, stage1, stage2, exact = [float(v) for v in st.brute_wave_steps[index]]
        row = rows.get("%s_triangle" % scan)
        if row is None or steps == 0 or plain:
            continue
        # (rays of one origin pay 5 for the front instead of 13; launches are not told apart here, so this is an upper estimate)
        executed = lanes * (steps * STAGE_FLOP["front"] + stage1 * STAGE_FLOP["stage1"] + stage2 * STAGE_FLOP["stage2"] + exact * STAGE_FLOP["as_written"])
        row["wave_steps"] = {"all": steps, "past_the_sign_test": stage1, "second_pre_test": stage2, "test_as_written": exact}
        row["executed_flop_upper_estimate"] = executed
        row["executed_frac"] = executed / (row["ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if row["ms"] > 0 else 0.0
    tflops = total_flop / (total_ms * 1e-3) / 1e12 if total_ms > 0 else 0.0
    total = {"what": "sum over kinds and scans of pairs x FLOP of the kind / sum of the scans' kernel seconds", "pairs": total_pairs, "flop": total_flop, "kernel_ms": total_ms,
             "pairs_per_s": total_pairs / (total_ms * 1e-3) if total_ms > 0 else 0.0, "achieved": tflops, "frac": tflops / PEAK_FP32_TFLOPS}
    return rows, total


class SummedStats:
    """The work counters and launch times of several rt_stats, added up (what kernel_table reads)."""
    def __init__(self, stats):
        for name in ("closest_tests", "shadow_tests", "closest_kind_ms", "shadow_kind_ms", "closest_kind_launches", "shadow_kind_launches"):
            setattr(self, name, [sum(getattr(s, name)[k] for s in stats) for k in range(3)])
        self.brute_wave_steps = [[sum(s.brute_wave_steps[i][k] for s in stats) for k in range(4)] for i in range(2)]
        for name in ("closest_rays", "shadow_rays", "device_ms"):
            setattr(self, name, sum(getattr(s, name) for s in stats))


def whole_frame_roofline(stats, fi, args, label, depth, elapsed):
    """--brute-force: the `roofline` and `brute_force` objects from the launches of the timed frames themselves."""
    st = SummedStats(stats)
    rows, total = kernel_table(st, fi, args.plain_test)
    closest_rows = {k: v for k, v in rows.items() if k.startswith("closest_")}
    head_name = max(closest_rows, key=lambda k: closest_rows[k]["ms"]) if closest_rows else None
    out = {"bound": "valu_fp32", "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s"}
    out.update(dict(rows[head_name]) if head_name else {"kernel": None, "achieved": 0.0, "frac": 0.0})
    out.update({"traffic": None, "sample": "the %d timed frames of %s, whole, maxdepth %d: %d closest-hit rays, %d shadow rays" % (len(stats), label, depth, st.closest_rays, st.shadow_rays),
                "flop_basis": "reference-equivalent: counted pairs x FLOP of the reference's test; see `executed_frac` of the triangle scans for what was executed",
                "kernels": rows, "all_kinds": total, "dominant_by_time": max(rows, key=lambda k: rows[k]["ms"]) if rows else None})
    rays = float(st.closest_rays + st.shadow_rays)
    summary = {"what": "RT_FLAG_NO_FILTER on whole frames: every ray against every primitive, the mode BASELINE's largest configuration names",
               "mrays_s": rays / elapsed / 1e6, "tests_per_s": total["pairs"] / elapsed, "tests_per_kernel_s": total["pairs_per_s"], "frac_of_fp32_peak": total["frac"],
               "device_ms": st.device_ms, "rays": rays, "pairs": total["pairs"]}
    return out, summary


def brute_force_leg(abi, fi, cam, scene, depth, args, label):
    """Every ray against every primitive (RT_FLAG_NO_FILTER) on a centred crop of the same frame; one chunk, one stream,
    launches one after the other, each between two HIP events on the stream it is launched on (rt_stats.*_kind_ms).
    Returns the `roofline` object (headed by the closest-hit scan of the kind that takes the most closest-hit time, with every
    kind's closest-hit and shadow scan and their sum beside it) and the `brute_force` summary."""
    crop = args.brute_crop or BRUTE_CROP.get(args.workload, BRUTE_CROP["default"])
    rect, cw, ch = centred(fi, crop)
    flags = abi.RT_FLAG_NO_FILTER | (abi.RT_FLAG_PLAIN_TEST if args.plain_test else 0)
    prm = fi.params(max_bounce_count=depth, flags=flags, **rect)
    brute = scene.render(cam, prm)
    fast = scene.render(cam, fi.params(max_bounce_count=depth, **rect))
    st = brute["stats"]
    rows, total = kernel_table(st, fi, args.plain_test)
    closest_rows = {k: v for k, v in rows.items() if k.startswith("closest_")}
    head_name = max(closest_rows, key=lambda k: closest_rows[k]["ms"]) if closest_rows else None
    head = dict(rows[head_name]) if head_name else {"kernel": None, "achieved": 0.0, "frac": 0.0}
    measured = profile_figures(label, args.plain_test) if head_name == "closest_triangle" else {}
    out = {"bound": "valu_fp32", "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s"}
    out.update(head)
    out.update({
        "traffic": measured.get("closest_hbm_bytes_per_launch"),
        "valu_issue_utilisation": measured.get("closest_valu_issue_utilisation"),
        "traffic_source": (measured.get("source") or "profiles/traffic.json") + ": rocprofv3 PMC passes of this command run by the builder; copied, not measured in this run" if measured else None,
        "flop_basis": "reference-equivalent: counted pairs x FLOP of the reference's test (ray-independent parts hoisted). Every pair is visited; the "
                      + ("test is evaluated as written" if args.plain_test else
                         "pair is decided by a conservative pre-test where that is certain and by the test as written otherwise, so fewer operations are executed per "
                         "pair than are credited and `frac` can exceed 1 (results identical to the test as written; `executed_frac` of a triangle scan prices the "
                         "stages its wavefronts really went through, `as_written` below is the figure with every test evaluated as written)"),
        "sample": "centred %dx%d-pixel crop of %s, maxdepth %d: %d closest-hit rays, %d shadow rays" % (cw, ch, label, depth, st.closest_rays, st.shadow_rays),
        "kernels": rows,
        "all_kinds": total,
        "dominant_by_time": max(rows, key=lambda k: rows[k]["ms"]) if rows else None,
        "shadow_pairs_counted": "primitives staged while the ray was still open (triangle scan: a ray joins the list at the tile its workgroup has reached, goes once "
                                "round and leaves at the first refill after it is blocked)",
        "leg_device_ms": st.device_ms,
        "image_identical_to_default_mode": bool((brute["rgb"] == fast["rgb"]).all()),
        "counters_identical_to_default_mode": all(getattr(st, f) == getattr(fast["stats"], f) for f in ("closest_rays", "hits", "shadow_rays", "shadow_blocked")),
    })
    if "shadow_triangle" in rows:
        out["shadow_kernel"] = dict(rows["shadow_triangle"], traffic=measured.get("shadow_hbm_bytes_per_launch"), valu_issue_utilisation=measured.get("shadow_valu_issue_utilisation"))
    if not args.plain_test and not args.no_as_written:
        # the same scans with every test evaluated exactly as written, on a quarter of the crop (they take three to five times as long)
        small, sw, sh = centred(fi, "%dx%d" % (max(cw // 2, 16), max(ch // 2, 16)))
        plain = scene.render(cam, fi.params(max_bounce_count=depth, flags=abi.RT_FLAG_NO_FILTER | abi.RT_FLAG_PLAIN_TEST, **small))
        staged = scene.render(cam, fi.params(max_bounce_count=depth, flags=abi.RT_FLAG_NO_FILTER, **small))
        plain_rows, plain_total = kernel_table(plain["stats"], fi, True)
        out["as_written"] = {"sample": "centred %dx%d-pixel crop" % (sw, sh), "frac": {k: v["frac"] for k, v in plain_rows.items()}, "all_kinds_frac": plain_total["frac"],
                             "kernel_ms": plain_total["kernel_ms"], "image_identical_to_staged": bool((plain["rgb"] == staged["rgb"]).all())}
    rays = float(st.closest_rays + st.shadow_rays)
    summary = {"what": "RT_FLAG_NO_FILTER: every ray against every primitive, the mode BASELINE's largest configuration names; " + out["sample"],
               "mrays_s": rays / (st.device_ms * 1e-3) / 1e6 if st.device_ms > 0 else 0.0, "tests_per_s": total["pairs_per_s"],
               "frac_of_fp32_peak": total["frac"], "device_ms": st.device_ms, "rays": rays, "pairs": total["pairs"]}
    return out, summary


def profile_figures(label, plain):
    """Figures of the brute-force triangle scans from the committed rocprofv3 PMC passes (profiles/traffic.json), or {}."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        got = table.get(label + ("/brute-force-plain" if plain else "/brute-force"))
        if not isinstance(got, dict):
            return {}
        got = dict(got)
        got.setdefault("source", table.get("source"))
        return got
    except (OSError, ValueError):
        return {}


def cpu_baseline(api, abi, fi, cam, scene, depth, crop):
    """Time the CPU renderer on a centred crop of the same frame, on this box's host cores.

    The unmodified reference (oracle/_ref/librt_ref.so) when that library is present, else the scalar restatement in
    oracle/.  As many workers as this process has hardware threads (the reference starts hardware_concurrency() of them,
    Source/main.cpp:77-83), fed from one atomic counter like the reference's queue (Source/render_work_queue.cpp:59-80) with
    segments of four pixels, so that the workers finish together.  Ray counts for the crop come from the GPU path, whose
    counters are identical to the oracle's.
    """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import checkers
    rect, cw, ch = centred(fi, crop)
    prm = fi.params(max_bounce_count=depth, **rect)
    g = scene.render(cam, prm, want_ids=True)
    rays = g["stats"].closest_rays + g["stats"].shadow_rays
    pairs = float(sum(g["stats"].closest_tests[k] for k in range(3)))     # closest-hit pairs; the CPU's shadow scans stop at the first blocker and are not counted here
    threads = checkers.available_cpus()
    if checkers.have_ref():
        lib, prefix, kind = checkers.load_ref(), "ref_", "reference"
    else:
        lib, prefix, kind = checkers.load_oracle(), "rto_", "port"
    r = checkers.cpu_render(lib, prefix, fi.scene_desc, cam, prm, threads=threads)
    # A primary ray that hits a zero-padding lane of the last batch of 8 (a sphere of radius 0 at the origin: a ray exactly through the
    # origin) makes the reference read a material past the end of its vector, i.e. whatever the heap holds; such pixels are left out.
    import numpy as np
    ids = g["ids"]
    d = fi.scene_desc
    limit = np.choose(np.minimum(ids >> 30, 2), [int(d.n_triangles), int(d.n_spheres), int(d.n_ellipsoids)])
    padding = ((ids < 0xFFFFFFFE) & ((ids & 0x3FFFFFFF) >= limit)).any(axis=2)
    differing = (r["rgb"] != g["rgb"]).any(axis=2)
    same = bool(not (differing & ~padding).any())
    busy = r["worker_seconds"]
    return {"value": rays / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": threads, "cores_available": threads, "threads_used": threads,
            "cpu_count_reported_by_os": os.cpu_count(), "kind": kind, "cpu_model": cpu_model(),
            "sample": "centred %dx%d-pixel crop of the same frame (%d rays), %.1f s" % (cw, ch, rays, r["seconds"]),
            "seconds": r["seconds"], "closest_hit_tests_per_s": pairs / r["seconds"],
            "worker_seconds_min_max": [float(busy.min()), float(busy.max())], "workers_finish_within": float(1.0 - busy.min() / busy.max()) if busy.max() > 0 else None,
            "image_identical_to_gpu": same, "pixels_compared": int((~padding).sum()), "pixels_left_out_padding_lane_hit": int(padding.sum()),
            "pixels_differing": int((differing & ~padding).sum())}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
