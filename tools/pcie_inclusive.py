"""PCIe-inclusive rates of the host-buffer entry points: scene upload + kernel + D2H, as f64 colours
(rt_hip_render) and as 8-bit samples quantised on the device (rt_hip_render_rgb8), plus the host P3 writer."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
from oracle import orc

def best(fn, n=3):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t)
    return min(ts), out

# the exercises at 4K, then the bench's own scene (10 000 primitives: the tables are 10 MB, the flatten and the device
# hierarchy build are no longer small) at 4K and at 8K
cases = [("pattern", 4096, 2160), ("glass_bubble", 4096, 2160), ("synthetic", 4096, 2160), ("synthetic", 7680, 4320)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, W, H in cases:
    world, cam = rt.synthetic(W, H, 10000) if name == "synthetic" else rt.exercise(name, W, H)
    MB = W * H * 3 * 8 / 1e6
    cam.render(world)  # warm-up
    _, (img, st) = best(lambda: cam.render(world, stats=True), 1)  # the counting kernel: rays per frame
    rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
    t, _ = best(lambda: cam.render(world))
    print(f"{name} {W}x{H} via rt_hip_render (flatten + scene upload + kernel + {MB:.0f} MB f64 frame to the caller's pageable buffer): "
          f"{t*1e3:.1f} ms wall, {rays/t/1e6:.0f} Mrays/s PCIe-inclusive")
    # the C ABI call alone: tables already flattened, the caller's buffer already touched (as a Canvas's Vec is)
    import ctypes as C
    from raytracer_amd import _native as N
    flat = world.flatten()
    cd = cam.desc()
    buf = np.zeros((cam.v_size, cam.h_size, 3))
    buf8 = np.zeros((cam.v_size, cam.h_size, 3), dtype=np.uint8)
    def call():
        N.check(N.hip.rt_hip_render(flat.desc, C.byref(cd), 5, buf.ctypes.data_as(C.POINTER(C.c_double)), None), "rt_hip_render")
    def call8():
        N.check(N.hip.rt_hip_render_rgb8(flat.desc, C.byref(cd), 5, buf8.ctypes.data_as(C.POINTER(C.c_uint8)), None), "rt_hip_render_rgb8")
    call(); call8()
    tc, _ = best(call, 5)
    tc8, _ = best(call8, 5)
    assert np.array_equal(buf, img.pixels)
    print(f"  rt_hip_render alone (tables flattened, caller's buffer touched): {tc*1e3:.2f} ms f64, {tc8*1e3:.2f} ms rgb8")
    scene = rt.DeviceScene(flat)
    def res():
        N.check(N.hip.rt_hip_scene_render(scene._p, C.byref(cd), 5, buf.ctypes.data_as(C.POINTER(C.c_double)), None), "rt_hip_scene_render")
    def res8():
        N.check(N.hip.rt_hip_scene_render_rgb8(scene._p, C.byref(cd), 5, buf8.ctypes.data_as(C.POINTER(C.c_uint8)), None), "rt_hip_scene_render_rgb8")
    res(); res8()
    tr, _ = best(res, 5)
    tr8, _ = best(res8, 5)
    print(f"  rt_hip_scene_render on the resident scene: {tr*1e3:.2f} ms f64, {tr8*1e3:.2f} ms rgb8")
    scene.close()
    t8, rgb8 = best(lambda: cam.render_rgb8(world))
    print(f"{name} {W}x{H} via rt_hip_render_rgb8 (the same, quantised on the device: {MB/8:.1f} MB): {t8*1e3:.1f} ms wall, "
          f"{rays/t8/1e6:.0f} Mrays/s PCIe-inclusive")
    if name == "synthetic":  # where the time of a first call goes
        tfl, _ = best(lambda: world.flatten(), 3)
        def create():
            sc = rt.DeviceScene(flat); sc.close()
        tcr, _ = best(create, 3)
        print(f"  of which: flatten on the host {tfl*1e3:.1f} ms; rt_hip_scene_create + destroy (upload, device tables, hierarchy) {tcr*1e3:.1f} ms; "
              f"kernels {tr8*1e3:.2f} ms; f64 frame to the host {max(tr - tr8, 0)*1e3:.2f} ms more than the rgb8 one")
    if W > 4096:
        continue  # (the text writer below is measured at 4K)
    tq, q = best(lambda: img.quantized(), 1)
    assert np.array_equal(q, rgb8)
    tf, text = best(lambda: rt.Canvas.ppm_from_rgb8(rgb8))
    to, ref = best(lambda: orc.to_ppm(img.pixels), 1)
    assert text == ref
    print(f"  P3 text ({len(text)/1e6:.1f} MB): table-driven threaded writer {tf*1e3:.1f} ms; reference-style per-token "
          f"formatting of the f64 canvas (oracle port, 1 thread) {to*1e3:.1f} ms; host quantise alone {tq*1e3:.1f} ms")
    tp, _ = best(lambda: cam.render_ppm(world))
    print(f"  camera.render(&world).to_ppm() end to end: {tp*1e3:.1f} ms")
