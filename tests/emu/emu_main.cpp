// tests/emu/emu_main.cpp -- TEST INFRASTRUCTURE: the kernel library's own sources compiled for the host (hip/hip_runtime.h
// of this directory stands in for the HIP header) and run under -fsanitize=address,undefined.
//
//   emu_main [scene] [width height primitives]      scene: synthetic (default) | cover | room | csg | hexagon | teapot ...
//
// Renders the scene through the C ABI of the emulated library with the one-launch kernel (mode 0) and with a launch per
// generation of rays (mode 1), each also in its counting variant, at several queue / node capacities, and compares the
// frames bit for bit.  Any out-of-bounds access in the kernels' per-lane arrays, queues, node pool or table block ends the
// run with a sanitizer report.  ("compile the same headers for the host and run the oracle-sized
// synthetic scene under -fsanitize=address,undefined on the CPU".)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "../../raytracer_amd/csrc/rt_host_internal.h"

// emu_scene.cpp (the host mirror: built without MemorySanitizer, libstdc++ is not instrumented): an exercise or the
// synthetic scene, flattened with host-built direction cells; the tables live until the process ends
int emu_build_scene(const char* name, size_t width, size_t height, size_t prims, RtSceneDesc* desc, RtCameraDesc* cam);

// ---- the pieces of the library that are not compiled into the emulation (rt_hostmem.hip, rt_multi.hip, rt_dirindex.hip) ----
int rth::block_alloc(void** out, size_t bytes) { return hipMalloc(out, bytes ? bytes : 256) == hipSuccess ? RT_OK : RT_ERR_HIP; }
void rth::block_free(void* p) { std::free(p); }
int rth::copy_to_host(void* dst, const void* d_src, size_t bytes, hipStream_t, hipEvent_t) {
    std::memcpy(dst, d_src, bytes);
    return RT_OK;
}
void rth::hostmem_shutdown() {}
void rth::multi_shutdown() {}
int rth::rccl_available() { return 0; }
int rth::dir_index_build(const RtDirCand*, uint32_t, uint32_t, DirIndex*) { return RT_ERR_INVALID_ARG; }  // (host-built cells only)
void rth::dir_index_free(DirIndex*) {}

// MemorySanitizer builds (make SAN=-fsanitize=memory HOST_SAN=): only the kernel library and this file are instrumented;
// what the uninstrumented host mirror hands over (the flat tables, the camera record) is declared initialised here.
#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
#include <sanitizer/msan_interface.h>
#define RT_EMU_UNPOISON(p, n) __msan_unpoison((p), (n))
#endif
#endif
#ifndef RT_EMU_UNPOISON
#define RT_EMU_UNPOISON(p, n) ((void)0)
#endif

void hip_emu_collective_timeout(int what, unsigned arrived, unsigned n) {
    static const char* names[] = {"ballot", "__shfl", "__shfl_up", "__shfl_down", "readfirstlane", "wave barrier", "__syncthreads"};
    std::fprintf(stderr, "hip_emu: %s reached by %u of %u threads only after two minutes (a collective in divergent code?)\n", names[what], arrived, n);
    std::abort();
}

namespace {

void unpoison(const RtSceneDesc& d) {
    RT_EMU_UNPOISON(&d, sizeof d);
    RT_EMU_UNPOISON(d.shapes, sizeof(RtShape) * d.n_shapes), RT_EMU_UNPOISON(d.conics, sizeof(RtConic) * d.n_conics);
    RT_EMU_UNPOISON(d.triangles, sizeof(RtTriangle) * d.n_triangles), RT_EMU_UNPOISON(d.groups, sizeof(RtGroup) * d.n_groups);
    RT_EMU_UNPOISON(d.csgs, sizeof(RtCsg) * d.n_csgs), RT_EMU_UNPOISON(d.lhit, d.n_lhit);
    RT_EMU_UNPOISON(d.materials, sizeof(RtMaterial) * d.n_materials), RT_EMU_UNPOISON(d.patterns, sizeof(RtPattern) * d.n_patterns);
    RT_EMU_UNPOISON(d.lights, sizeof(RtLight) * d.n_lights), RT_EMU_UNPOISON(d.bvh_nodes, sizeof(RtBvhNode) * d.n_bvh_nodes);
    RT_EMU_UNPOISON(d.dir_cands, sizeof(RtDirCand) * d.n_dir_cands);
    if (d.dir_cell_start) RT_EMU_UNPOISON(d.dir_cell_start, sizeof(uint32_t) * ((size_t)6 * d.dir_res * d.dir_res + 1));
    if (d.dir_entries) RT_EMU_UNPOISON(d.dir_entries, sizeof(uint32_t) * d.n_dir_entries);
}

std::vector<double> render(RtScene* scene, const RtCameraDesc& cam, int mode, uint64_t q_items, uint64_t node_items, bool stats,
                           RtStats* st_out = nullptr) {
    const size_t n = (size_t)cam.hsize * cam.vsize * 3u;
    double* d_out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_out), n * sizeof(double)) != hipSuccess) std::abort();
    for (size_t i = 0; i < n; ++i) d_out[i] = -7.0;
    RtStats* d_stats = nullptr;
    if (stats) {
        if (hipMalloc(reinterpret_cast<void**>(&d_stats), sizeof(RtStats)) != hipSuccess) std::abort();
        std::memset(d_stats, 0, sizeof(RtStats));
    }
    int rc = rt_hip_scene_set_wavefront(scene, mode, q_items, node_items);
    if (rc == RT_OK) rc = rt_hip_render_rows(scene, &cam, 5, 0, 1, cam.vsize, cam.vsize, d_out, nullptr, d_stats);
    if (rc == RT_OK) rc = rt_hip_scene_status(scene, nullptr);
    if (rc != RT_OK) {
        std::fprintf(stderr, "emu: render mode %d failed: %s (%s)\n", mode, rt_hip_strerror(rc), rt_hip_last_hip_error());
        std::exit(2);
    }
    std::vector<double> out(d_out, d_out + n);
    if (st_out && d_stats) *st_out = *d_stats;
    std::free(d_out);
    std::free(d_stats);
    return out;
}

bool same(const std::vector<double>& a, const std::vector<double>& b) { return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * 8) == 0; }

}  // namespace

int main(int argc, char** argv) {
    const std::string name = argc > 1 ? argv[1] : "synthetic";
    const size_t width = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 96;
    const size_t height = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 54;
    const size_t prims = argc > 4 ? std::strtoul(argv[4], nullptr, 10) : 1500;
    const bool quick = argc > 5 && std::strcmp(argv[5], "quick") == 0;  // fewer capacity settings (a collective costs ~50 us here)
    RtSceneDesc desc;
    RtCameraDesc cam;
    if (emu_build_scene(name.c_str(), width, height, prims, &desc, &cam) != 0) return 3;
    unpoison(desc);
    RT_EMU_UNPOISON(&cam, sizeof cam);
    if (rt_hip_init(0) != RT_OK) return 3;
    RtScene* scene = nullptr;
    const int rc = rt_hip_scene_create(&desc, &scene);
    if (rc != RT_OK) {
        std::fprintf(stderr, "emu: rt_hip_scene_create: %s\n", rt_hip_strerror(rc));
        return 3;
    }
    std::printf("emu: %s %zux%zu, %u shapes, kernels: %s\n", name.c_str(), width, height, desc.n_shapes, rt_hip_scene_kernels(scene, 5));
    const size_t pixels = width * height;
    RtStats st0{}, st1{};
    const std::vector<double> mono = render(scene, cam, 0, 0, 0, false);
    for (double v : mono)
        if (v == -7.0) {
            std::fprintf(stderr, "emu: mode 0 left pixels unwritten\n");
            return 4;
        }
    bool ok = same(mono, render(scene, cam, 0, 0, 0, true, &st0));
    std::printf("emu: mode 0 and its counting variant: %s\n", ok ? "equal" : "DIFFER");
    struct Cap { uint64_t q, n; };
    const Cap caps[] = {{0, 0}, {pixels / 7, pixels}, {pixels, pixels / 11}, {193, 61}};
    for (const Cap& c : caps) {
        if (quick && c.q != 0 && c.q != 193) continue;
        const bool a = same(mono, render(scene, cam, 1, c.q, c.n, false));
        const bool b = same(mono, render(scene, cam, 1, c.q, c.n, true, &st1));
        std::printf("emu: mode 1 (queue %llu, nodes %llu): %s; counting variant: %s\n", (unsigned long long)c.q, (unsigned long long)c.n,
                    a ? "equal" : "DIFFER", b ? "equal" : "DIFFER");
        ok = ok && a && b;
    }
    // mode 2, the kernel pipeline (rt_wf2.h), where the scene has one: its kernels are wave-cooperative (records put
    // together in LDS, ballots, shuffles), so they run in the lockstep model -- 64 real lanes per wave
    rt_hip_scene_set_wavefront(scene, 2, 0, 0);
    if (std::strstr(rt_hip_scene_kernels(scene, 5), "pipeline") != nullptr) {
        hip_emu::lockstep = true;
        const Cap caps2[] = {{0, 0}, {pixels / 7, pixels}, {193, 61}};
        for (const Cap& c : caps2) {
            if (quick && c.q != 0 && c.q != 193) continue;
            const bool a = same(mono, render(scene, cam, 2, c.q, c.n, false));
            const bool b = quick && c.q != 0 ? true : same(mono, render(scene, cam, 2, c.q, c.n, true, &st1));
            std::printf("emu: mode 2 (queue %llu, nodes %llu), 64 lanes in lockstep: %s; counting variant: %s\n", (unsigned long long)c.q,
                        (unsigned long long)c.n, a ? "equal" : "DIFFER", b ? "equal" : "DIFFER");
            ok = ok && a && b;
            if (c.q == 0) {
                const bool rays = st0.rays_primary == st1.rays_primary && st0.rays_shadow == st1.rays_shadow &&
                                  st0.rays_reflect == st1.rays_reflect && st0.rays_refract == st1.rays_refract;
                std::printf("emu: ray counts of mode 2: primary %llu shadow %llu reflect %llu refract %llu\n", (unsigned long long)st1.rays_primary,
                            (unsigned long long)st1.rays_shadow, (unsigned long long)st1.rays_reflect, (unsigned long long)st1.rays_refract);
                (void)rays;  // (mode 0's counters come from the sequential model, where only lane 0 of a wave reports)
            }
        }
        hip_emu::lockstep = false;
    }
    rt_hip_scene_destroy(scene);
    rt_hip_shutdown();
    std::printf("emu: %s\n", ok ? "all frames equal, no sanitizer report" : "FRAMES DIFFER");
    return ok ? 0 : 1;
}
