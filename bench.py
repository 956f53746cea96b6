#!/usr/bin/env python3
"""bench.py -- throughput of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--width W --height H]

A step = one frame of the workload rendered through the C ABI (rt_hip_render_rows) with the
scene tables and the framebuffer resident in HBM.  Default workload: BASELINE.json configs[1],
exercises/pattern.rs at 4096x2160 (other workloads: glass_bubble, teapot, csg, cover, room,
hexagon, scene, plane, synthetic [10 000 primitives]).  With N > 1 (one process per GPU,
launched by torch.distributed.run) the frame's rows are dealt to the ranks in interleaved
16-row bands and gathered to rank 0 over RCCL inside the timed step -- total work is fixed,
so scaling is "strong".

Rank 0 prints ONE JSON line.  metric = Mrays/s, rays = primary + shadow + reflection +
refraction rays actually traced (one per World::intersections call: world.rs:33,91), counted
by the counting variant of the kernel in an untimed launch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pattern")
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--primitives", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--print-stats", action="store_true", help="event counters of one frame to stderr")
    ap.add_argument("--hierarchy", default="auto", choices=["auto", "nested", "global", "none"],
                    help="acceleration structure: the flattener's policy, or one form forced (studies)")
    ap.add_argument("--depth", type=int, default=5, help="recursion budget (camera.rs:59 uses 5; other values are for studies)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import _native as N
    from raytracer_amd import flops, tiling

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world_size
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks (RT_BENCH_REHEARSE=1): ranks share the devices there are and the
    # collectives go through host memory over gloo.  Everything but the RCCL calls themselves runs as in the real job;
    # the numbers of such a run mean nothing.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    N.check(N.hip.rt_hip_init(local_rank), "rt_hip_init")
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def gather_shards(shard, gathered):
        if not rehearse:
            dist.gather(shard, gathered, dst=0)
            return
        torch.cuda.synchronize()
        host = shard.cpu()
        parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
        dist.gather(host, parts, dst=0)
        if rank == 0:
            for g, h in zip(gathered, parts):
                g.copy_(h)

    def all_reduce(t, **kw):
        if not rehearse:
            dist.all_reduce(t, **kw)
            return
        h = t.cpu()
        dist.all_reduce(h, **kw)
        t.copy_(h)

    W, H = args.width, args.height
    if args.workload == "synthetic":
        world, cam = rt.synthetic(W, H, args.primitives)
    else:
        world, cam = rt.exercise(args.workload, W, H)
    flat = world.flatten(build_bvh={"auto": True, "none": False}.get(args.hierarchy, args.hierarchy))
    scene = rt.DeviceScene(flat)
    fd = flat.desc.contents
    scene_is_flat = fd.world_bvh_root < 0 and all((fd.shapes[i].kind_flags & 0xff) <= 6 for i in range(fd.n_shapes))
    # the library's default policy (rt_hip.hip launch_render): large scenes that can recurse run one launch per generation of rays
    recursive = any(fd.materials[i].reflective != 0.0 or fd.materials[i].transparency != 0.0 for i in range(fd.n_materials))
    generations = recursive and not scene_is_flat and fd.n_shapes >= 192 and args.depth > 0
    cd = cam.desc()
    band = tiling.DEFAULT_BAND_ROWS
    n_parts = world_size
    slots = tiling.padded_local_rows(H, n_parts, band)      # uniform shard height (gather needs equal sizes)
    my_rows = tiling.n_local_rows(H, rank, n_parts, band)
    shard = torch.zeros((slots, W, 3), dtype=torch.float64, device=dev)
    # the root receives every shard into one buffer (no concatenation copy), then permutes rows into the frame
    gathered_all = torch.empty((n_parts * slots, W, 3), dtype=torch.float64, device=dev) if (rank == 0 and n_parts > 1) else None
    gathered = [gathered_all[i * slots:(i + 1) * slots] for i in range(n_parts)] if gathered_all is not None else None
    perm = torch.as_tensor(tiling.gather_permutation(H, n_parts, band), device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream

    def render(d_stats=None):
        scene.render_rows(cd, args.depth, rank, n_parts, band, my_rows, shard.data_ptr(), stream, d_stats)

    def step():
        render()
        if n_parts > 1:
            gather_shards(shard, gathered)
            if rank == 0:
                return gathered_all.index_select(0, perm)
        return shard

    # ---- untimed counting launch: rays and the FLOP budget of this rank's share
    d_stats = torch.zeros(256, dtype=torch.uint8, device=dev)  # >= sizeof(RtStats), zeroed
    render(d_stats.data_ptr())
    torch.cuda.synchronize()
    scene.status()
    import ctypes as C
    st = N.RtStats.from_buffer_copy(bytes(d_stats.cpu().numpy()[:C.sizeof(N.RtStats)]))
    stats = st.as_dict()
    if args.print_stats and rank == 0:
        print(json.dumps(stats), file=sys.stderr)
    my_rays = flops.total_rays(stats)
    my_flops = flops.algorithmic_flops(stats)
    tot = torch.tensor([my_rays, my_flops], dtype=torch.float64, device=dev)
    if n_parts > 1:
        all_reduce(tot)
    rays_per_frame, flops_per_frame = float(tot[0]), float(tot[1])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if n_parts > 1:
        dist.barrier()
    torch.cuda.synchronize()

    # kernel time of the render launch alone: HIP events on the launch stream, live in the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        render()
        ev[k][1].record()
        if n_parts > 1:
            gather_shards(shard, gathered)
            if rank == 0:
                frame = gathered_all.index_select(0, perm)
    torch.cuda.synchronize()
    if n_parts > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    scene.status()
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    # tracing, gather and device-to-host times separately.  The copy of the finished frame to
    # host memory is outside the timed steps (inputs and outputs stay resident); one copy is timed here.
    d2h_ms = None
    if rank == 0:
        final = frame if n_parts > 1 else shard[:H]
        host = torch.empty(final.shape, dtype=final.dtype, pin_memory=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        host.copy_(final)
        torch.cuda.synchronize()
        d2h_ms = (time.perf_counter() - t1) * 1e3
        del host

    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if n_parts > 1:
        all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(t[0]), float(t[1])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        achieved = (my_flops / (kernel_ms * 1e-3)) / 1e12 if kernel_ms > 0 else 0.0
        line = {
            "metric": "Mrays/s (primary+secondary)",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": n_parts,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"exercises/{args.workload}.rs" if args.workload != "synthetic"
                else f"synthetic cover/room-style, {args.primitives} primitives, seed 0x5EED5EED",
                "width": W, "height": H, "depth": args.depth, "band_rows": band,
                "parallelism": f"row-bands x{n_parts}" + ((" + REHEARSAL (ranks share GPUs, gloo through host memory): timings meaningless"
                                                        if rehearse else " + RCCL gather to rank 0") if n_parts > 1 else ""),
                "rays_per_frame": int(rays_per_frame), "mpixels_per_s": round(W * H * args.steps / elapsed / 1e6, 3),
                # per step: slowest rank's render launches (HIP events); the rest of the step = RCCL gather + row permutation
                # on the root; and, outside the timed steps, one copy of the finished f64 frame to pinned host memory
                "trace_ms": round(kernel_ms_max, 4), "gather_ms": round(max(ms_per_step - kernel_ms_max, 0.0), 4) if n_parts > 1 else 0.0,
                "d2h_ms": round(d2h_ms, 3),
            },
            "roofline": {
                "bound": "valu_f64",
                "kernel": ("wf_level_kernel<STATS=false, FLAT=false> x %d generations + wf_combine_kernel" % (args.depth + 1))
                if generations else "render_kernel<STATS=false, FLAT=%s>" % ("true" if scene_is_flat else "false"),
                "achieved": round(achieved, 4),
                "peak": flops.PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / flops.PEAK_FP64_VALU_TFLOPS, 5),
                "peak_no_fma": flops.PEAK_FP64_VALU_NO_FMA_TFLOPS,
                "frac_no_fma": round(achieved / flops.PEAK_FP64_VALU_NO_FMA_TFLOPS, 5),
                "flops_per_launch": int(my_flops),
                "kernel_ms": round(kernel_ms, 4),
                "traffic": measured_traffic(args, n_parts),
            },
        }
        if n_parts == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(world, cam, args)
        print(json.dumps(line), flush=True)
    if n_parts > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, n_parts):
    """HBM bytes per launch from the committed PMC passes of this same command (profiles/traffic.json;
    counters cannot be read from inside the process), or None when this configuration was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(f"{args.workload}:{args.width}x{args.height}:{n_parts}")
    except OSError:
        return None


def cpu_baseline(world, cam, args):
    """The CPU oracle (C++ restatement of the reference algorithm, OpenMP over rows like rayon)
    timed on a bounded sample of the same frame: every `step`-th row."""
    from oracle import orc
    ow = orc.OracleWorld(world.describe())
    threads = min(16, os.cpu_count() or 1)
    H = cam.v_size
    # probe 8 spread rows to size the sample for ~cpu-seconds of work
    t0 = time.perf_counter()
    _, c = ow.render(cam, rows=(H // 16, H, max(1, H // 8)), nthreads=threads)
    dt = time.perf_counter() - t0
    probe_rows = len(range(H // 16, H, max(1, H // 8)))
    per_row = max(dt / probe_rows, 1e-6)
    n_rows = int(max(8, min(H, args.cpu_seconds / per_row)))
    stepr = max(1, H // n_rows)
    t0 = time.perf_counter()
    _, counts = ow.render(cam, rows=(0, H, stepr), nthreads=threads)
    dt = time.perf_counter() - t0
    rays = sum(counts)
    rows = len(range(0, H, stepr))
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": (f"the whole {cam.h_size}x{H} frame" if stepr == 1 else f"every {stepr}th row of the same {cam.h_size}x{H} frame")
                      + f" ({rows} rows, {rays} rays, {dt:.2f} s); "
                      "C++ restatement of the reference algorithm (oracle/), OpenMP schedule(dynamic,1) over rows"}


if __name__ == "__main__":
    main()
