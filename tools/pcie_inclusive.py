"""PCIe-inclusive rate of the host-buffer entry point (rt_hip_render): scene upload + kernel + D2H."""
import sys, time
sys.path.insert(0, '.')
import raytracer_amd as rt
for name in ("pattern", "glass_bubble"):
    world, cam = rt.exercise(name, 4096, 2160)
    cam.render(world)  # warm-up
    ts = []
    for _ in range(3):
        t = time.perf_counter(); img, st = cam.render(world, stats=True); ts.append(time.perf_counter() - t)
    rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
    t = min(ts)
    print(f"{name} 4096x2160 via rt_hip_render (host in/out, counting kernel): {t*1e3:.1f} ms wall, kernel {st.kernel_ms:.2f} ms, "
          f"{rays/t/1e6:.0f} Mrays/s PCIe-inclusive (212 MB framebuffer D2H to pageable memory)")
