#!/usr/bin/env python3
"""bench.py -- throughput of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--width W --height H]

A step = one frame of the workload rendered through the C ABI (rt_hip_render_rows) with the
scene tables and the framebuffer resident in HBM.  Default workload: the one BASELINE.json's
metric is quoted on ("4K frame at 1/2/4/8 MI355X"; north_star: "throughput on synthetic scenes"):
the 10 000-primitive synthetic scene of configs[4] at 4096x2160.  Other workloads by --workload:
pattern, glass_bubble, teapot, csg, cover, room, hexagon, scene, plane; --width/--height give the
7680x4320 frame.

With N > 1 there is one process per GPU.  `python bench.py --gpus N` typed as such starts the N
ranks itself (a child `python -m torch.distributed.run`; the parent touches no GPU and relays
rank 0's line); launched under torch.distributed.run it is a rank.  The frame's rows are dealt to
the ranks in interleaved 16-row bands and gathered to rank 0 over RCCL inside the timed steps
(--gather f64: the shard as rendered; rgb8: its 8-bit samples, rt_hip_quantize per shard); frame
k's gather runs beside frame k+1's trace (two shard buffers) unless --no-overlap.  Total work is
fixed, so scaling is "strong".

Rank 0 prints ONE JSON line.  metric = Mrays/s, rays = primary + shadow + reflection +
refraction rays actually traced (one per World::intersections call: world.rs:33,91), counted
by the counting variant of the kernels in an untimed launch.
"""
import argparse
import json
import os
import subprocess
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
    ap.add_argument("--workload", default="synthetic")
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--primitives", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--print-stats", action="store_true", help="event counters of one frame to stderr")
    ap.add_argument("--hierarchy", default="auto", choices=["auto", "nested", "global", "none"],
                    help="acceleration structure: the flattener's policy, or one form forced (studies)")
    ap.add_argument("--depth", type=int, default=5, help="recursion budget (camera.rs:59 uses 5; other values are for studies)")
    ap.add_argument("--wavefront", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="rt_hip_scene_set_wavefront mode (-1: the library's policy; studies)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each oracle sample")
    ap.add_argument("--gather", default="f64", choices=["f64", "rgb8"],
                    help="N > 1: what the ranks send to the root -- the f64 shard, or its 8-bit samples "
                         "(rt_hip_quantize per shard, canvas.rs:85-90: an eighth of the bytes)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: gather frame k before tracing frame k+1")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the ranks as fresh child processes.
    This process makes no GPU call (a GPU-initialised process must not be replaced or forked on this pool)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def host_threads():
    """Threads the CPU baseline may use: the CPUs this process may run on, within the harness's preset
    OMP_NUM_THREADS (the GPU box gives a one-GPU job a 16-CPU share of a much larger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)


def main():
    args = parse()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd import _native as N
    from raytracer_amd import flops, tiling

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    def gather_shards(send, gathered, async_op=False):
        if not rehearse:
            return dist.gather(send, gathered, dst=0, async_op=async_op)
        torch.cuda.synchronize()
        host = send.cpu()
        parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
        dist.gather(host, parts, dst=0)
        if rank == 0:
            for g, h in zip(gathered, parts):
                g.copy_(h)
        return None

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
    if args.wavefront >= 0:
        scene.set_wavefront(args.wavefront)
    cd = cam.desc()
    band = tiling.DEFAULT_BAND_ROWS
    n_parts = world_size
    slots = tiling.padded_local_rows(H, n_parts, band)      # uniform shard height (gather needs equal sizes)
    my_rows = tiling.n_local_rows(H, rank, n_parts, band)
    rgb8 = args.gather == "rgb8" and n_parts > 1
    overlap = n_parts > 1 and not args.no_overlap and not rehearse
    n_buf = 2 if overlap else 1
    shards = [torch.zeros((slots, W, 3), dtype=torch.float64, device=dev) for _ in range(n_buf)]
    sends = [torch.zeros((slots, W, 3), dtype=torch.uint8, device=dev) for _ in range(n_buf)] if rgb8 else shards
    # the root receives every shard into one buffer (no concatenation copy), then permutes rows into the frame
    root = rank == 0 and n_parts > 1
    gathered_all = [torch.empty((n_parts * slots, W, 3), dtype=sends[0].dtype, device=dev) for _ in range(n_buf)] if root else None
    gathered = [[g[i * slots:(i + 1) * slots] for i in range(n_parts)] for g in gathered_all] if root else [None] * n_buf
    perm = torch.as_tensor(tiling.gather_permutation(H, n_parts, band), device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream

    def render(buf=0, d_stats=None):
        scene.render_rows(cd, args.depth, rank, n_parts, band, my_rows, shards[buf].data_ptr(), stream, d_stats)
        if rgb8:
            N.check(N.hip.rt_hip_quantize(shards[buf].data_ptr(), shards[buf].numel(), sends[buf].data_ptr(), stream),
                    "rt_hip_quantize")

    frame = [None]

    def finish(buf, work):
        if work is not None:
            work.wait()
        if rank == 0:
            frame[0] = gathered_all[buf].index_select(0, perm)

    def run(steps, events=None):
        """`steps` frames; with two buffers the gather of frame k is in flight while frame k+1 is traced."""
        pending = None
        for k in range(steps):
            b = k % n_buf
            if events:
                events[k][0].record()
            render(b)
            if events:
                events[k][1].record()
            if n_parts == 1:
                continue
            if pending is not None:
                finish(*pending)
                pending = None
            work = gather_shards(sends[b], gathered[b], async_op=overlap)
            if overlap:
                pending = (b, work)
            else:
                finish(b, None)
        if pending is not None:
            finish(*pending)

    # ---- untimed counting launch: rays and the FLOP budget of this rank's share
    import ctypes as C
    d_stats = torch.zeros(256, dtype=torch.uint8, device=dev)  # >= sizeof(RtStats), zeroed
    render(0, d_stats.data_ptr())
    torch.cuda.synchronize()
    scene.status()
    st = N.RtStats.from_buffer_copy(bytes(d_stats.cpu().numpy()[:C.sizeof(N.RtStats)]))
    stats = st.as_dict()
    if args.print_stats and rank == 0:
        print(json.dumps(stats), file=sys.stderr)
    my_rays = flops.total_rays(stats)
    my_flops = flops.algorithmic_flops(stats)     # FP64 work only
    my_flops_f32 = flops.accel_f32_flops(stats)   # the f32 box tests of the 4-wide hierarchy, priced apart
    tot = torch.tensor([my_rays, my_flops], dtype=torch.float64, device=dev)
    if n_parts > 1:
        all_reduce(tot)
    rays_per_frame = float(tot[0])

    run(args.warmup)
    torch.cuda.synchronize()
    if n_parts > 1:
        dist.barrier()
    torch.cuda.synchronize()

    # kernel time of the render launches alone: HIP events on the launch stream, live in the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    run(args.steps, ev)
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
        final = frame[0] if n_parts > 1 else shards[(args.steps - 1) % n_buf][:H]
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
        achieved_f32 = (my_flops_f32 / (kernel_ms * 1e-3)) / 1e12 if kernel_ms > 0 else 0.0
        synthetic = args.workload == "synthetic"
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
            "data": "synthetic" if synthetic else f"the reference's own scene constants (exercises/{args.workload}.rs), no external data",
            "config": {
                "workload": f"exercises/{args.workload}.rs" if not synthetic
                else f"synthetic cover/room-style, {args.primitives} primitives, seed 0x5EED5EED (BASELINE.json configs[4])",
                "width": W, "height": H, "depth": args.depth, "band_rows": band,
                "parallelism": f"row-bands x{n_parts}" + ((" + REHEARSAL (ranks share GPUs, gloo through host memory): timings meaningless"
                                                        if rehearse else f" + RCCL gather ({args.gather}) to rank 0"
                                                        + (", overlapped with the next frame's trace" if overlap else "")) if n_parts > 1 else ""),
                "rays_per_frame": int(rays_per_frame), "mpixels_per_s": round(W * H * args.steps / elapsed / 1e6, 3),
                # per step: slowest rank's render launches (HIP events); the rest of the step = what the RCCL gather + row
                # permutation on the root add to it; and, outside the timed steps, one copy of the finished frame to pinned host memory
                "trace_ms": round(kernel_ms_max, 4), "gather_ms": round(max(ms_per_step - kernel_ms_max, 0.0), 4) if n_parts > 1 else 0.0,
                "d2h_ms": round(d2h_ms, 3),
            },
            "roofline": {
                "bound": "valu_f64",
                "kernel": scene.pipeline_name(args.depth),
                "achieved": round(achieved, 4),
                "peak": flops.PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / flops.PEAK_FP64_VALU_TFLOPS, 5),
                "peak_no_fma": flops.PEAK_FP64_VALU_NO_FMA_TFLOPS,
                "frac_no_fma": round(achieved / flops.PEAK_FP64_VALU_NO_FMA_TFLOPS, 5),
                "flops_per_launch": int(my_flops),
                # `achieved` / `frac` above are the FP64 work alone.  The f32 box tests of the 4-wide hierarchy
                # (acceleration only: RtStats.accel_f32_tests x 12) are priced apart, against the FP32 vector peak;
                # `frac_combined_r02` is rounds 1-2's figure (both numerators over the FP64 peak), kept for continuity.
                "flops_fp64": int(my_flops),
                "flops_f32_accel": int(my_flops_f32),
                "f32_accel": {"achieved": round(achieved_f32, 4), "peak": flops.PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(achieved_f32 / flops.PEAK_FP32_VALU_TFLOPS, 5)},
                "frac_combined_r02": round((achieved + achieved_f32) / flops.PEAK_FP64_VALU_TFLOPS, 5),
                "kernel_ms": round(kernel_ms, 4),
                "traffic": measured_traffic(args, n_parts)[0],
                "traffic_source": measured_traffic(args, n_parts)[1],
            },
        }
        if n_parts == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(world, cam, args)
        print(json.dumps(line), flush=True)
    if n_parts > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, n_parts):
    """HBM bytes per frame from the committed PMC passes of this same command (profiles/traffic.json;
    counters cannot be read from inside the process), or None when this configuration was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            table = json.load(f)
        key = f"{args.workload}:{args.width}x{args.height}:{n_parts}"
        if args.depth != 5 or args.wavefront not in (-1, 2) or (args.workload == "synthetic" and args.primitives != 10000):
            return None, None  # (not the command the counters were taken on)
        return table.get(key), table.get("_source", {}).get(key)
    except OSError:
        return None, None


def _time_oracle(ow, cam, threads, seconds):
    """Oracle render of every `step`-th row, sized for about `seconds` of wall time -> (rays, seconds, rows, step)."""
    H = cam.v_size
    t0 = time.perf_counter()
    probe = range(H // 16, H, max(1, H // 8))  # 8 spread rows to size the sample
    ow.render(cam, rows=(probe.start, H, probe.step), nthreads=threads)
    per_row = max((time.perf_counter() - t0) / len(probe), 1e-6)
    n_rows = int(max(8, min(H, seconds / per_row)))
    step = max(1, H // n_rows)
    best = None
    for _ in range(3 if seconds / 3.0 > per_row * len(range(0, H, step)) else 1):  # short samples: best of 3
        t0 = time.perf_counter()
        _, counts = ow.render(cam, rows=(0, H, step), nthreads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return sum(counts), best, len(range(0, H, step)), step


def cpu_baseline(world, cam, args):
    """The CPU oracle (C++ restatement of the reference algorithm, OpenMP schedule(dynamic,1) over rows like
    rayon's work stealing) timed on a bounded sample of the same frame: every `step`-th row.  Two builds, as
    BASELINE.md states: -O2 (the parity oracle itself) and -O3 -march=native compiled on this host."""
    from oracle import orc
    threads = host_threads()
    ow = orc.OracleWorld(world.describe())
    rays, dt, rows, step = _time_oracle(ow, cam, threads, args.cpu_seconds)
    H = cam.v_size
    out = {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "host_logical_cpus": os.cpu_count(),
           "sample": (f"the whole {cam.h_size}x{H} frame" if step == 1 else f"every {step}th row of the same {cam.h_size}x{H} frame")
                     + f" ({rows} rows, {rays} rays, {dt:.2f} s); "
                     "C++ restatement of the reference algorithm (oracle/, -O2 -ffp-contract=off), OpenMP schedule(dynamic,1) "
                     f"over rows on {threads} threads (this job's CPU share; the host shows {os.cpu_count()} logical CPUs)"}
    native = cpu_baseline_native(args, threads)
    if native is not None:
        out["value_O3_march_native"] = native
    return out


def cpu_baseline_native(args, threads):
    """The same sample with the oracle compiled `-O3 -march=native -ffp-contract=off` on THIS host, in a child
    process so that neither the compile nor an unexpected fault can take the bench line down.  None when that fails."""
    code = r'''
import json, os, subprocess, sys, tempfile
root, workload, W, H, prims, threads, seconds = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7])
sys.path.insert(0, root)
so = os.path.join(tempfile.gettempdir(), "librt_oracle_native_%d.so" % os.getuid())
subprocess.run(["g++", "-std=c++17", "-O3", "-march=native", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-shared",
                "-o", so, os.path.join(root, "oracle", "rt_oracle.cpp")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
os.environ["RT_ORACLE_LIB"] = so
import raytracer_amd as rt
from oracle import orc
import bench
world, cam = rt.synthetic(W, H, prims) if workload == "synthetic" else rt.exercise(workload, W, H)
ow = orc.OracleWorld(world.describe())
rays, dt, rows, step = bench._time_oracle(ow, cam, threads, seconds)
print(json.dumps(round(rays / dt / 1e6, 4)))
'''
    try:
        r = subprocess.run([sys.executable, "-c", code, ROOT, args.workload, str(args.width), str(args.height), str(args.primitives),
                            str(threads), str(args.cpu_seconds)], capture_output=True, text=True, timeout=240)
        return json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:
        return None


if __name__ == "__main__":
    main()
