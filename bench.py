#!/usr/bin/env python3
"""Benchmark of the render loop: one step = one full frame of the workload, image tiles split
over the ranks (one process per GPU), framebuffer assembled on rank 0 from an RCCL all-gather.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload scene7_4k|scene1|...|synthetic]

Prints one JSON line on rank 0: metric, value, unit, n_gpus, steps, warmup, ms_per_step, ..., `roofline` (closest-hit
scan kernels against the FP32 vector peak) and `cpu_baseline` (the CPU renderer on a crop of the same frame).  For N > 1 launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes as C
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="scene7_4k")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--maxdepth", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk-tiles", type=int, default=0, help="tiles per chunk of ray state (0 = library default)")
    ap.add_argument("--brute-force", action="store_true", help="RT_FLAG_NO_FILTER: every ray-primitive pair through the literal full test")
    ap.add_argument("--plain-test", action="store_true", help="with --brute-force: RT_FLAG_PLAIN_TEST, the triangle test exactly as written (three IEEE divisions per pair)")
    ap.add_argument("--cpu-crop", default="240x135", help="centred crop the CPU baseline renders (WxH)")
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


def flops_of(stats, n_point_lights_share=1.0):
    closest = stats.closest_tests[0] * FLOP_TRI + stats.closest_tests[1] * FLOP_SPH + stats.closest_tests[2] * FLOP_ELL_CLOSEST
    shadow = stats.shadow_tests[0] * FLOP_TRI + stats.shadow_tests[1] * FLOP_SPH + stats.shadow_tests[2] * FLOP_ELL_SHADOW_POINT
    return closest, shadow


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE=%d)" % (args.gpus, world))
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    api.init(local_rank)

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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(k):
        """Frame k: wait for its exchange, then rank 0 scatters the gathered tiles into the image."""
        if exchanged[k % 2] is not None:
            exchanged[k % 2].synchronize()
        if rank == 0:
            api.assemble_tiles_device(gathered[k % 2].data_ptr(), stride, prm, frame.data_ptr())

    def run(n):
        stats = []
        for k in range(n):
            stats.append(scene.render_tiles_device(cam, prm, tiles[k % 2].data_ptr(), want_stats=True))   # returns with its streams idle
            if k:
                finish(k - 1)
            if world > 1:
                dist.all_gather_into_tensor(gathered[k % 2], tiles[k % 2])
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
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
    acc = torch.tensor(local, dtype=torch.float64, device=device)
    mx = acc.clone()
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    tot = dict(zip(fields + ["ct_tri", "ct_sph", "ct_ell", "st_tri", "st_sph", "st_ell", "pp_closest", "pp_shadow", "q_closest", "q_shadow", "xf_closest", "xf_shadow", "closest_ms", "shadow_ms", "device_ms",
                             "closest_launches", "shadow_launches"], acc.tolist()))
    rank_max = dict(zip(["closest_ms", "shadow_ms", "device_ms"], mx.tolist()[19:22]))

    if rank == 0:
        rays = tot["closest_rays"] + tot["shadow_rays"]
        tests = tot["ct_tri"] + tot["ct_sph"] + tot["ct_ell"] + tot["st_tri"] + tot["st_sph"] + tot["st_ell"]
        closest_flop = tot["ct_tri"] * FLOP_TRI + tot["ct_sph"] * FLOP_SPH + tot["ct_ell"] * FLOP_ELL_CLOSEST
        shadow_flop = tot["st_tri"] * FLOP_TRI + tot["st_sph"] * FLOP_SPH + tot["st_ell"] * FLOP_ELL_SHADOW_POINT
        # dominant kernel: k_closest. Per-GPU figure: this rank-summed FLOP over rank-summed kernel time.
        closest_tflops = closest_flop / (tot["closest_ms"] * 1e-3) / 1e12 if tot["closest_ms"] > 0 else 0.0
        shadow_tflops = shadow_flop / (tot["shadow_ms"] * 1e-3) / 1e12 if tot["shadow_ms"] > 0 else 0.0
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
                       "parallelism": "tiles%%%d" % world},
            "ray_prim_tests_per_s": tests / elapsed,
            "rays_per_step": rays / args.steps,
            "tests_per_step": tests / args.steps,
            "culled_samples_per_step": tot["culled"] / args.steps,
            "pair_pretests_per_step": [tot["pp_closest"] / args.steps, tot["pp_shadow"] / args.steps],
            "queued_pairs_per_step": [tot["q_closest"] / args.steps, tot["q_shadow"] / args.steps],
            "full_tests_per_step": tot["full_tests"] / args.steps,
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "k_closest",
                "achieved": closest_tflops,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": closest_tflops / PEAK_FP32_TFLOPS,
                "traffic": measured_traffic(label, args.brute_force),
                "executed": None if args.brute_force else {
                    "tflops": tot["xf_closest"] / (tot["closest_ms"] * 1e-3) / 1e12 if tot["closest_ms"] > 0 else 0.0,
                    "frac": (tot["xf_closest"] / (tot["closest_ms"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS) if tot["closest_ms"] > 0 else 0.0,
                    "what": "FLOP of the per-ray tests the hierarchy scans executed (box tests 24, two-stage pre-test 40, full test 54 / 19 / 95) / the same kernel time"},
                "mode": "brute force (literal full test on every pair)" if args.brute_force else "bounding hierarchy + conservative pre-tests, full test on the survivors",
                "note": ROOFLINE_NOTE,
                "flop_per_launch": closest_flop / max(tot["closest_launches"], 1.0),
                "avg_launch_ms": tot["closest_ms"] / max(tot["closest_launches"], 1.0),
                "launches": int(tot["closest_launches"]),
                "shadow_kernel": {"kernel": "k_shadow", "achieved": shadow_tflops, "frac": shadow_tflops / PEAK_FP32_TFLOPS,
                                  "avg_launch_ms": tot["shadow_ms"] / max(tot["shadow_launches"], 1.0), "launches": int(tot["shadow_launches"])},
                "whole_step_tflops": (closest_flop + shadow_flop) / elapsed / 1e12 / world,
            },
            "per_rank_max_ms": {k: v / args.steps for k, v in rank_max.items()},
            "setup_ms": {"parse_scene_file": parse_s * 1e3, "scene_create": scene_create_s * 1e3},   # rank 0, outside the timed steps
        }
        if not args.brute_force:
            # the frame the ranks assembled in the last step against one rendered by this rank alone (outside the timed steps)
            alone = scene.render(cam, fi.params(max_bounce_count=depth, flags=flags))
            out["frame_identical_to_single_rank_render"] = bool((alone["rgb"].reshape(-1) == frame.cpu().numpy()).all())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(api, abi, fi, cam, scene, depth, args.cpu_crop)
        sys.stdout.flush()
        os.write(results_fd, (json.dumps(out) + "\n").encode())

    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


ROOFLINE_NOTE = ("achieved = ray-primitive tests x FLOP of the reference formulation (54 triangle, 19 sphere, 95 ellipsoid) / HIP-event time of the "
                 "closest-hit scan launches; in the default mode most tests are decided by bounds, so frac is above 1 and is not a utilisation: "
                 "executed VALU utilisation (PMC) is in profiles/")


def measured_traffic(label, brute_force):
    """HBM bytes per closest-hit scan launch of this workload from the committed rocprofv3 PMC pass (profiles/traffic.json), or None."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return table.get(label + ("/brute-force" if brute_force else ""))
    except (OSError, ValueError):
        return None


def cpu_baseline(api, abi, fi, cam, scene, depth, crop):
    """Time the CPU renderer on a centred crop of the same frame, on this box's host cores.

    The unmodified reference (oracle/_ref/librt_ref.so, multithreaded over scanlines like its
    work queue) when that library is present, else the scalar restatement in oracle/.  Ray counts
    for the crop come from the GPU path, whose counters are identical to the oracle's.
    """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import checkers
    cw, ch = [int(v) for v in crop.split("x")]
    cw, ch = min(cw, fi.width), min(ch, fi.height)
    prm = fi.params(max_bounce_count=depth, rect_x0=(fi.width - cw) // 2, rect_y0=(fi.height - ch) // 2, rect_w=cw, rect_h=ch)
    g = scene.render(cam, prm)
    rays = g["stats"].closest_rays + g["stats"].shadow_rays
    threads = min(os.cpu_count() or 1, 16)
    if checkers.have_ref():
        lib, prefix, kind = checkers.load_ref(), "ref_", "reference"
    else:
        lib, prefix, kind = checkers.load_oracle(), "rto_", "port"
    r = checkers.cpu_render(lib, prefix, fi.scene_desc, cam, prm, threads=threads)
    same = bool((r["rgb"] == g["rgb"]).all())
    return {"value": rays / r["seconds"] / 1e6, "unit": "Mrays/s", "cores": threads, "kind": kind,
            "sample": "centred %dx%d-pixel crop of the same frame (%d rays), %.1f s" % (cw, ch, rays, r["seconds"]),
            "seconds": r["seconds"], "image_identical_to_gpu": same}


if __name__ == "__main__":
    main()
