#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path: Mrays/s (primary + bounce closest-hit
queries, GPURaytracer.cu:247 calls) on BASELINE.json configs[1] (low-poly model, 1920x1080,
16 spp, depth 4) unless --workload says otherwise.

One "step" = one frame (one pass of the hot path over W*H*spp samples).
 * frames mode (default): with N > 1 the ranks split frame indices (progressive accumulation):
   rank r renders frame step*N + r of the replicated scene and the clamped frames are combined with one RCCL
   reduce(SUM) per step, inside the timed region.  Per-GPU work is fixed => weak scaling.
 * tiles mode: ONE frame per step, split into interleaved 16-row bands (multigpu.BAND_ROWS; rt_render_bands: a rank's
   bands are one pass of the pipeline), composited with one all_gather => strong scaling.  This is the partitioning
   BASELINE.json's north_star names for configs[3] (C4).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline`
objects, and under "also" the same measurement of C4 (1 M triangles, 4K: the north-star target config) --
at N = 1 as a whole frame, at N > 1 through the tile split.  The CPU baseline is the oracle (reference kernel
logic, brute force) on a bounded crop of the same workload, centred on the geometry; it is a reported baseline, not the target.

roofline follows SURVEY.md 8(d): algorithmic bytes per ray = node fetches x 64 B + triangle tests x 48 B +
hits x (12 + 96 + 16 per bound map) B + 16 B per pixel of output (+ 16 read when progressive); flops per ray =
node tests x 30 + triangle tests x 45 + hits x 300.  The wavefront pipeline's own path-state traffic is NOT
algorithmic and is reported beside it (state_bytes_per_ray).  `frac` is the larger of the HBM and FP32-vector
fractions, as 8(d) defines it; `bound` says what the counters say actually limits the kernels.
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
NODE_B, TRI_B = 64.0, 48.0                 # one four-wide BVH node; 3 x i32 + 3 x 3 x f32 per triangle test (8d)
HIT_B, MAP_B = 12.0 + 96.0, 16.0           # per shaded hit: 3 indices + 3 x (12 + 12 + 8) B attributes; 4 texels x 4 B per bound map
NODE_F, TRI_F, HIT_F = 30.0, 45.0, 300.0   # flop per node test / triangle test / shaded hit (8d)
# Path state crossing HBM between the launches of the wavefront pipeline (not algorithmic), in bytes:
#   a camera ray: direction 16 written by wf_gen, read by the packet kernel and by wf_shade; hit word 8 written and read
#   a bounce ray: 64 written by wf_shade (origin, direction, light, beta + slot), 32 of them read by wf_trace; hit word 8 written,
#                 and read by the dealing pass
#   a hit: t / barycentrics 16 written and read; at a bounce level also the 64 of its state and its hit word again (gathered)
#   a sample: result 17 written when the path ends and read by wf_resolve; written once more for every bounce ray made
#             (its what-if-it-misses result: misses of the bounce levels are never touched again)
def state_bytes(n_camera, n_rays, n_hits):
    n_bounce = n_rays - n_camera
    bounce_hits = n_hits * n_bounce / max(n_rays, 1.0)   # the counters do not split the hits by level: pro rata
    return (n_camera * (16 * 3 + 8 * 2) + n_bounce * (64 + 32 + 8 * 2) + n_hits * (16 * 2) + bounce_hits * (64 + 8)
            + 17.0 * (2 * n_camera + n_bounce))


def custom_size(args):
    return bool(args.width or args.height or args.spp or args.depth)


def build_workload(name, args):
    from raytracer_amd import scenes
    if name == "c1":
        scene, cam, st = scenes.config1(args.width or 256, args.height or 256, args.spp or 1, args.depth or 4)
        label = "C1 cube+sphere (364 tris)"
    elif name == "c2":
        scene, cam, st = scenes.config2(args.width or 1920, args.height or 1080, args.spp or 16, args.depth or 4)
        label = scenes.workload_label("c2")
    else:
        fn = getattr(scenes, "config" + name[1:], None)
        if fn is None:
            raise SystemExit(f"unknown workload {name}")
        names = fn.__code__.co_varnames[:fn.__code__.co_argcount]
        given = {"width": args.width, "height": args.height, "spp": args.spp, "depth": args.depth}
        scene, cam, st = fn(**{k: v for k, v in given.items() if v and k in names})
        label = scenes.workload_label(name) if hasattr(scenes, "workload_label") else fn.__doc__.strip().split("\n")[0]
    return scene, cam, st, label


class Job:
    """Shared state of one bench process: torch / RCCL, the context, its stream."""

    def __init__(self, args):
        self.t0 = time.time()
        import torch
        import torch.distributed as dist
        from raytracer_amd import capi, multigpu
        self.torch, self.dist, self.capi, self.multigpu, self.args = torch, dist, capi, multigpu, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus and self.world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (see the task contract)")
        # RT_BENCH_REHEARSE=1: every rank on GPU 0 with gloo collectives -- a rehearsal of the N > 1 code paths on a one-GPU
        # box (RCCL refuses two ranks on one device); its timings mean nothing
        self.rehearse = bool(os.environ.get("RT_BENCH_REHEARSE"))
        self.verbose = self.rehearse or bool(os.environ.get("RT_BENCH_VERBOSE"))
        if self.rehearse:
            self.local_rank = 0
            import faulthandler
            faulthandler.dump_traceback_later(int(os.environ.get("RT_BENCH_REHEARSE_LIMIT", "240")), exit=True)   # a hang names its line
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.ctx = capi.Context(self.local_rank)
        # a real (non-null) stream shared by torch and librt_amd: rt_set_stream(NULL) would mean the
        # context's own stream, and the legacy default stream's handle IS null
        self.stream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.stream)
        self.ctx.set_stream(self.stream.cuda_stream)
        # frames in flight: frame i is rendered on lane i % inflight -- a workspace lane of the library (RT_OPT_WF_LANE) and a
        # stream of its own -- so that one frame's last rays finish under the next frame's bulk
        self.inflight = max(1, min(2, args.inflight))
        self.lanes = [torch.cuda.Stream(device=self.dev) for _ in range(self.inflight)] if self.inflight > 1 else []
        self.ctx.set_option(capi.OPT_TRAVERSAL, 1 if args.brute else (0 if args.pipeline == "wavefront" else 2))
        if args.wf_lanes:
            self.ctx.set_option(capi.OPT_WF_LANES, args.wf_lanes)
        if args.builder:
            self.ctx.set_option(capi.OPT_BUILDER, args.builder)
        if args.wf_chunk:
            self.ctx.set_option(capi.OPT_WF_CHUNK, args.wf_chunk)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def note(job, msg):
    """Progress on stderr (RT_BENCH_VERBOSE=1 or the rehearsal): stdout carries the one JSON line only."""
    if job.verbose:
        print(f"[bench r{job.rank} {time.time() - job.t0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def measure(job, workload, mode, steps, warmup, cpu_baseline):
    """W untimed + K timed steps of one workload; returns the JSON object (rank 0) or None."""
    torch, dist, capi, multigpu, args, ctx = job.torch, job.dist, job.capi, job.multigpu, job.args, job.ctx
    rank, world, dev = job.rank, job.world, job.dev
    note(job, f"{workload} / {mode}: building the scene")
    scene, cam, st, label = build_workload(workload, args)
    W, H = cam.width, cam.height
    flat = scene.flatten()
    t0 = time.time()
    ctx.upload_scene(flat)
    upload_s = time.time() - t0
    ctx.set_camera_rays(cam.preTransformRays)
    note(job, f"{workload}: uploaded and committed in {upload_s:.1f} s")

    fa = tc = None
    if mode == "frames":
        # two buffers, so that the RCCL reduce of one frame runs while the next one renders
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        acc_t = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        fbs = [ctx.framebuffer_wrap(b.data_ptr(), W, H) for b in bufs]
        fa = multigpu.FrameAccumulator(dist, world, rank, bufs, acc_t, streams=job.lanes or None)
        frame_t = bufs[0]
    else:
        # two frames in flight: the all_gather of frame i runs on a side stream while frame i + 1 renders
        tc = multigpu.TileCompositor(torch, dist, W, H, rank, world, dev, buffers=2 if (world > 1 or job.lanes) else 1)   # a buffer per frame in flight
        fbs = [ctx.framebuffer_wrap(f.data_ptr(), W, H) for f in tc.frames]
        frame_t = tc.frames[0]

    def on_main():
        if job.lanes:
            ctx.set_option(capi.OPT_WF_LANE, 0)
            ctx.set_stream(job.stream.cuda_stream)

    def step(frame_base):
        if not job.lanes:
            return render_step(frame_base)
        lane = frame_base % len(job.lanes)
        ctx.set_option(capi.OPT_WF_LANE, lane)
        ctx.set_stream(job.lanes[lane].cuda_stream)
        with torch.cuda.stream(job.lanes[lane]):
            render_step(frame_base)

    def fork():     # the lanes start behind what the main stream holds
        for s in job.lanes:
            s.wait_stream(job.stream)

    def join():     # ... and the main stream goes on behind the lanes
        for s in job.lanes:
            job.stream.wait_stream(s)
        on_main()

    def render_step(frame_base):
        st.progressiveFrameCount = 0
        if mode == "frames":
            st.frameCount = frame_base * world + rank
            b = fa.acquire(frame_base)
            ctx.render(cam, st, fbs[b])
            fa.submit(b)
        else:
            st.frameCount = frame_base
            if world > 1:
                b = tc.acquire(frame_base)
                ctx.render(cam, st, fbs[b], bands=(multigpu.BAND_ROWS, rank, world))
                tc.submit(b)
            else:
                ctx.render(cam, st, fbs[frame_base % len(fbs)])

    def barrier():
        if fa is not None:
            fa.drain()
        if tc is not None:
            tc.drain()
        job.barrier()

    fork()
    for i in range(warmup):
        step(i)
    join()
    barrier()
    note(job, f"{workload}: warm-up done")
    ctx.set_option(capi.OPT_ACCUMULATE_STATS, 1)
    barrier()
    ctx.event_record(0)
    t0 = time.perf_counter()
    fork()
    for i in range(steps):
        step(warmup + i)
    if fa is not None:
        fa.drain()
    if tc is not None:
        tc.drain()
    join()
    ctx.event_record(1)
    barrier()
    elapsed = time.perf_counter() - t0
    note(job, f"{workload}: {steps} steps in {elapsed:.3f} s")
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
        wavefront = args.pipeline == "wavefront" and not args.brute
        kernel_ms = gpu_ms / steps  # HIP events on the launch stream over the timed region; one rt_render(_bands) per step
        maps_per_hit = sum(1.0 for m in flat.materials for k in ("texNormal", "texAlbedo", "texRoughness", "texMetal")
                           if m[k] >= 0) / max(1, len(flat.materials))
        pix_local = float(W) * H * steps / (world if mode == "tiles" else 1)
        # a packet of 64 camera rays fetches a node once (scalar load) and tests it per ray
        node_fetches = s.node_visits - s.packet_node_tests + s.packet_node_fetches
        alg_bytes = (node_fetches * NODE_B + s.tri_tests * TRI_B + s.hits * (HIT_B + MAP_B * maps_per_hit) + pix_local * 16.0)
        alg_flops = s.node_visits * NODE_F + s.tri_tests * TRI_F + s.hits * HIT_F
        hbm_gbs = alg_bytes / steps / (kernel_ms * 1e-3) / 1e9
        valu_tf = alg_flops / steps / (kernel_ms * 1e-3) / 1e12
        hbm_frac, valu_frac = hbm_gbs / HBM_PEAK_GBS, valu_tf / FP32_PEAK_TFLOPS
        by_hbm = hbm_frac >= valu_frac
        nr = max(1, rays_local)
        out = {
            "metric": "Mrays/s (primary+bounce closest-hit queries)",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if mode == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "spp": int(st.sampleCount),
                       "depth": int(st.bounceCount), "triangles": int(s.triangles), "objects": len(flat.objects),
                       "mode": mode, "traversal": "brute" if args.brute else "bvh", "pipeline": args.pipeline,
                       "frames_in_flight": max(1, len(job.lanes)),
                       "rays_per_frame": rays_total / steps / (world if mode == "frames" else 1), "scene_upload_s": upload_s,
                       "bvh_build_ms": float(s.commit_ms), "lib": ctx.L.rt_version().decode()},
            "roofline": {
                # what the counters show (profiles/): VALU issue slots 70-85 % full at partial lane
                # use in the trace kernels, L2 latency in wf_shade; HBM bandwidth is not the limit at these scene sizes
                "bound": "valu-issue at partial lane use + L2 latency (not HBM bandwidth; see profiles/)",
                "achieved": hbm_gbs if by_hbm else valu_tf, "peak": HBM_PEAK_GBS if by_hbm else FP32_PEAK_TFLOPS,
                "unit": "GB/s" if by_hbm else "TFLOP/s", "frac": max(hbm_frac, valu_frac),
                # HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) come from separate rocprofv3 --pmc passes of this same
                # command (tools/pmc.sh -> profiles/traffic.json, which names the build it was taken on); they cannot be
                # measured from inside this run, so the figure is filled in below only when that build is this one
                "traffic": None,
                "kernel": ("all launches of one rt_render call (wf_gen, wf_trace + wf_shade per bounce level, wf_resolve, redo)"
                           if wavefront else ("path_kernel + redo" if not args.brute else "render_kernel (brute force)")),
                "kernel_ms": kernel_ms,
                "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac, "bytes_per_ray": alg_bytes / nr},
                "valu": {"achieved": valu_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_frac, "flop_per_ray": alg_flops / nr},
                # the pipeline's own HBM traffic per query (state_bytes above): nominal, from the counts of this run
                "state_bytes_per_ray": (state_bytes(float(st.sampleCount) * pix_local, float(nr), float(s.hits)) / nr) if wavefront else 0.0,
                "nodes_per_ray": s.node_visits / nr, "node_fetches_per_ray": node_fetches / nr,
                "tris_per_ray": s.tri_tests / nr, "hits_per_ray": s.hits / nr, "slow_path_frac": s.slow_path / nr},
        }
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(workload)
                if tj and not args.brute and wavefront and not custom_size(args) and (world == 1 or mode == "frames"):   # the profile is of whole frames
                    same = tj.get("lib") == ctx.L.rt_version().decode()   # rt_version() carries a hash of the library's sources
                    out["roofline"]["traffic_profile"] = {"file": "profiles/traffic.json", "lib": tj.get("lib"), "same_build": same,
                                                          "hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch")}
                    if same and tj.get("hbm_bytes_per_launch"):
                        t = float(tj["hbm_bytes_per_launch"])
                        out["roofline"]["traffic"] = t
                        out["roofline"]["traffic_unit"] = "bytes per launch (one rt_render call), HBM side, from the PMC passes of this build"
                        out["roofline"]["traffic_over_algorithmic"] = t / (alg_bytes / steps)
                        # the share of that traffic that is the pipeline's own path state (nominal bytes per query x queries)
                        out["roofline"]["state_share"] = out["roofline"]["state_bytes_per_ray"] * nr / steps / t
                        out["roofline"]["traffic_hbm_frac"] = t / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            except Exception:
                pass
        if cpu_baseline:
            from oracle import oracle_py as orc
            # bounded sample: the oracle is brute force (every ray tests every triangle of every object it does
            # not cull), so the crop shrinks with the scene to keep the CPU leg around 10-30 s
            budget = 2.2e10
            c = int(min(args.cpu_crop, W, H, max(8.0, (budget / (st.sampleCount * 1.8 * max(1, int(s.triangles)))) ** 0.5)))
            # centred on the geometry (the centroid of the pixels that differ from the frame's corner, i.e. from the sky), so
            # that the CPU leg and the live parity figure are about hits, not about a patch of sky
            fr = frame_t[..., :3]
            mask = ((fr - fr[0, 0]).abs().amax(-1) > 1e-3)
            if bool(mask.any()):
                ys, xs = torch.nonzero(mask, as_tuple=True)
                cx, cy = int(xs.float().mean().item()), int(ys.float().mean().item())
            else:
                cx, cy = W // 2, H // 2
            x0, y0 = min(max(cx - c // 2, 0), W - c), min(max(cy - c // 2, 0), H - c)
            st.frameCount = warmup
            st.progressiveFrameCount = 0
            threads = orc.lib().or_num_threads()
            tcpu = time.perf_counter()
            ref, _, nq = orc.render(flat, cam, st, rect=(x0, y0, x0 + c, y0 + c), want_ids=False)
            cpu_s = time.perf_counter() - tcpu
            # same frame on the GPU for a live parity figure on the crop
            ctx.render(cam, st, fbs[0], (x0, y0, x0 + c, y0 + c))
            ctx.sync()
            g = frame_t[y0:y0 + c, x0:x0 + c, :3].cpu().numpy()
            err = float(np.abs(g - ref[y0:y0 + c, x0:x0 + c, :3]).max())
            out["cpu_baseline"] = {"value": nq / cpu_s / 1e6, "unit": "Mrays/s", "cores": int(threads), "kind": "port",
                                   "sample": f"{c}x{c} crop at ({x0}, {y0}), centred on the geometry, of frame {st.frameCount}, {st.sampleCount} spp, "
                                             f"depth {st.bounceCount}: {nq} rays in {cpu_s:.1f} s",
                                   "gpu_vs_oracle_max_abs_on_crop": err}
    for fb in fbs:
        ctx.framebuffer_destroy(fb)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--mode", default="", choices=["", "frames", "tiles"],
                    help="default: frames (weak scaling) for the headline workload; the C4 line under `also` uses tiles when N > 1")
    ap.add_argument("--also", default="c4", help="second workload measured in the same run and reported under `also` ('' = none)")
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
    ap.add_argument("--inflight", type=int, default=2, help="frames in flight per GPU (1 or 2): frame i is rendered on workspace lane and stream i %% inflight (RT_OPT_WF_LANE)")
    args = ap.parse_args()

    job = Job(args)
    mode = args.mode or "frames"
    custom = bool(args.width or args.height or args.spp or args.depth or args.brute)
    out = measure(job, args.workload, mode, args.steps, args.warmup, cpu_baseline=not args.no_cpu_baseline and job.world == 1)
    if args.also and args.also != args.workload and not custom:
        # the north-star target config: one GPU renders the whole frame, N > 1 split it by image tiles
        also = measure(job, args.also, "tiles", args.steps, args.warmup, cpu_baseline=False)
        if out is not None and also is not None:
            out["also"] = [{k: also[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "config", "roofline")}]
    if job.rank == 0:
        print(json.dumps(out), flush=True)
    if job.world > 1:
        job.dist.barrier()
        job.dist.destroy_process_group()
    job.ctx.close()


if __name__ == "__main__":
    main()
