#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path: Mrays/s (primary + bounce closest-hit
queries, GPURaytracer.cu:247 calls) on BASELINE.json configs[1] (low-poly model, 1920x1080,
16 spp, depth 4) unless --workload says otherwise.

One "step" = one frame (one pass of the hot path over W*H*spp samples) per GPU.  With N > 1 the
ranks split frame indices (progressive accumulation): rank r renders frame
step*N + r of the replicated scene and the clamped frames are combined with one RCCL
reduce(SUM) per step, inside the timed region.  Per-GPU work is fixed => weak scaling.
`--mode tiles` instead splits ONE frame into interleaved row bands (strong scaling) and
all-gathers them.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and
`cpu_baseline` objects.  The CPU baseline is the oracle (reference kernel logic, brute force)
on a bounded centred crop of the same workload; it is a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector


def build_workload(name, args):
    from raytracer_amd import scenes
    if name == "c1":
        scene, cam, st = scenes.config1(args.width or 256, args.height or 256, args.spp or 1, args.depth or 4)
        label = "C1 cube+sphere (364 tris)"
    elif name == "c2":
        scene, cam, st = scenes.config2(args.width or 1920, args.height or 1080, args.spp or 16, args.depth or 4)
        label = "C2 synthetic low-poly aeroplane stand-in + 1024^2 albedo texture"
    else:
        fn = getattr(scenes, "config" + name[1:], None)
        if fn is None:
            raise SystemExit(f"unknown workload {name}")
        names = fn.__code__.co_varnames[:fn.__code__.co_argcount]
        given = {"width": args.width, "height": args.height, "spp": args.spp, "depth": args.depth}
        scene, cam, st = fn(**{k: v for k, v in given.items() if v and k in names})
        label = fn.__doc__.strip().split("\n")[0]
    return scene, cam, st, label


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--mode", default="frames", choices=["frames", "tiles"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--cpu-crop", type=int, default=448, help="side of the centred crop the CPU baseline renders")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--brute", action="store_true", help="reference-order brute force instead of the BVH")
    ap.add_argument("--pipeline", default="wavefront", choices=["wavefront", "persistent"],
                    help="BVH render pipeline: per-bounce trace/shade launches (default) or the single persistent kernel")
    ap.add_argument("--wf-lanes", type=int, default=0, help="chunks in flight (HIP streams) of the wavefront pipeline (0 = library default)")
    ap.add_argument("--builder", type=int, default=0, help="acceleration structure builder: 0 = PLOC (library default), 1 = radix tree")
    ap.add_argument("--wf-chunk", type=int, default=0, help="samples in flight per chunk of the wavefront pipeline (0 = library default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import capi, multigpu

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (see the task contract)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scene, cam, st, label = build_workload(args.workload, args)
    W, H = cam.width, cam.height
    flat = scene.flatten()

    ctx = capi.Context(local_rank)
    # a real (non-null) stream shared by torch and librt_amd: rt_set_stream(NULL) would mean the
    # context's own stream, and the legacy default stream's handle IS null
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_option(capi.OPT_TRAVERSAL, 1 if args.brute else (0 if args.pipeline == "wavefront" else 2))
    if args.wf_lanes:
        ctx.set_option(capi.OPT_WF_LANES, args.wf_lanes)
    if args.builder:
        ctx.set_option(capi.OPT_BUILDER, args.builder)
    if args.wf_chunk:
        ctx.set_option(capi.OPT_WF_CHUNK, args.wf_chunk)
    t0 = time.time()
    ctx.upload_scene(flat)
    upload_s = time.time() - t0
    ctx.set_camera_rays(cam.preTransformRays)
    fb_t = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    acc_t = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    fb = ctx.framebuffer_wrap(fb_t.data_ptr(), W, H)
    # frames mode: two buffers, so that the RCCL reduce of one frame runs while the next one renders
    fb_t2 = torch.zeros((H, W, 4), dtype=torch.float32, device=dev) if args.mode == "frames" else None
    fb2 = ctx.framebuffer_wrap(fb_t2.data_ptr(), W, H) if fb_t2 is not None else None
    fa = multigpu.FrameAccumulator(dist, world, rank, [fb_t, fb_t2], acc_t) if args.mode == "frames" else None

    if args.mode == "tiles":
        rects = multigpu.band_rects(W, H, rank, world)
    else:
        rects = [(0, 0, W, H)]

    def step(frame_base):
        st.progressiveFrameCount = 0
        if args.mode == "frames":
            st.frameCount = frame_base * world + rank
            b = fa.acquire(frame_base)
            ctx.render(cam, st, (fb, fb2)[b], rects[0])
            fa.submit(b)
        else:
            st.frameCount = frame_base
            for r in rects:
                ctx.render(cam, st, fb, r)
            if world > 1:
                multigpu.composite_tiles(dist, fb_t, W, H, rank, world)

    def barrier():
        if fa is not None:
            fa.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    ctx.set_option(capi.OPT_ACCUMULATE_STATS, 1)
    barrier()
    ctx.event_record(0)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if fa is not None:
        fa.drain()
    ctx.event_record(1)
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ctx.event_elapsed_ms(0, 1)
    s = ctx.stats()
    rays_local = int(s.rays)
    ctx.set_option(capi.OPT_ACCUMULATE_STATS, 0)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_local], dtype=torch.int64, device=dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())
    else:
        rays_total = rays_local

    out = None
    if rank == 0:
        launches = args.steps * len(rects)
        kernel_ms = gpu_ms / launches  # HIP events on the launch stream over the timed region
        node_b, tri_b, shade_b = 64.0, 48.0, 48.0 + 80.0 + 64.0
        tex_per_hit = sum(16.0 for m in flat.materials for k in ("texNormal", "texAlbedo", "texRoughness", "texMetal")
                          if m[k] >= 0) / max(1, len(flat.materials))
        samples_local = W * H * st.sampleCount * args.steps if args.mode == "frames" else sum(
            (r[2] - r[0]) * (r[3] - r[1]) for r in rects) * st.sampleCount * args.steps
        pix_local = samples_local / st.sampleCount
        # wavefront pipeline: per closest-hit query the path state crosses HBM between the launches
        # (trace: ray 32 + queue 4 in, hit 20 out; shade: ray 32 + hit 20 + state 32 + queue 4 in,
        # ray 32 + state 32 + queue 4 out) and per sample 16 + 1 bytes of result out and in again
        state_b = (32 + 4 + 20) + (32 + 20 + 32 + 4) + (32 + 32 + 4) if args.pipeline == "wavefront" and not args.brute else 0.0
        # a packet of 64 camera rays fetches a node once (scalar load) and tests it per ray
        node_fetches = s.node_visits - s.packet_node_tests + s.packet_node_fetches
        alg_bytes = (node_fetches * node_b + s.tri_tests * tri_b + s.hits * (shade_b + tex_per_hit)
                     + rays_local * state_b + samples_local * (64.0 + (34.0 if state_b else 0.0)) + pix_local * 16.0)
        alg_flops = s.node_visits * 44.0 + s.tri_tests * 45.0 + s.hits * 300.0 + samples_local * 60.0
        achieved_gbs = alg_bytes / launches / (kernel_ms * 1e-3) / 1e9
        achieved_tf = alg_flops / launches / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and not args.brute:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary+bounce closest-hit queries)",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.mode == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "spp": int(st.sampleCount),
                       "depth": int(st.bounceCount), "triangles": int(s.triangles), "objects": len(flat.objects),
                       "mode": args.mode, "traversal": "brute" if args.brute else "bvh", "pipeline": args.pipeline,
                       "rays_per_frame": rays_local / args.steps, "scene_upload_s": upload_s,
                       "bvh_build_ms": float(s.commit_ms)},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("all launches of one rt_render call (wf_gen, wf_trace + wf_shade per bounce level, wf_resolve, redo); "
                                    "wf_trace dominates, see profiles/") if args.pipeline == "wavefront" and not args.brute
                         else ("path_kernel + redo" if not args.brute else "render_kernel (brute force)"),
                         "kernel_ms": kernel_ms,
                         "bytes_per_ray": alg_bytes / max(1, rays_local), "nodes_per_ray": s.node_visits / max(1, rays_local),
                         "node_fetches_per_ray": node_fetches / max(1, rays_local),
                         "tris_per_ray": s.tri_tests / max(1, rays_local), "slow_path_frac": s.slow_path / max(1, rays_local),
                         "valu": {"achieved": achieved_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": achieved_tf / FP32_PEAK_TFLOPS, "flop_per_ray": alg_flops / max(1, rays_local)}},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg runs at N = 1 only
            from oracle import oracle_py as orc
            # bounded sample: the oracle is brute force (every ray tests every triangle of every object it does
            # not cull), so the crop shrinks with the scene to keep the CPU leg around 10-30 s
            budget = 2.2e10
            c = int(min(args.cpu_crop, W, H, max(8.0, (budget / (st.sampleCount * 1.8 * max(1, int(s.triangles)))) ** 0.5)))
            x0, y0 = (W - c) // 2, (H - c) // 2
            st.frameCount = args.warmup * (world if args.mode == "frames" else 1)
            st.progressiveFrameCount = 0
            threads = orc.lib().or_num_threads()
            tc = time.perf_counter()
            ref, _, nq = orc.render(flat, cam, st, rect=(x0, y0, x0 + c, y0 + c), want_ids=False)
            cpu_s = time.perf_counter() - tc
            # same frame on the GPU for a live parity figure on the crop
            ctx.render(cam, st, fb, (x0, y0, x0 + c, y0 + c))
            ctx.sync()
            g = fb_t[y0:y0 + c, x0:x0 + c, :3].cpu().numpy()
            err = float(np.abs(g - ref[y0:y0 + c, x0:x0 + c, :3]).max())
            out["cpu_baseline"] = {"value": nq / cpu_s / 1e6, "unit": "Mrays/s", "cores": int(threads), "kind": "port",
                                   "sample": f"centred {c}x{c} crop of frame {st.frameCount}, {st.sampleCount} spp, "
                                             f"depth {st.bounceCount}: {nq} rays in {cpu_s:.1f} s",
                                   "gpu_vs_oracle_max_abs_on_crop": err}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
