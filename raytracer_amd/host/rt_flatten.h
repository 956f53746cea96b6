// Scene flattening: World (object tree) -> the flat record tables of
// include/rt_hip.h that the HIP kernels traverse.  Runs once per render on the
// host.  Everything ray-independent that the reference recomputes per ray is
// hoisted here with the reference's own arithmetic:
//   * the group gate box  = bounds.transform(self.transform)  (compound.rs:93-94)
//   * CSG membership      = left.includes(primitive)          (compound.rs:154)
#pragma once
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_scene.h"

namespace rt {

struct FlatScene {
    std::vector<RtShape> shapes;
    std::vector<RtConic> conics;
    std::vector<RtTriangle> triangles;
    std::vector<RtGroup> groups;
    std::vector<RtCsg> csgs;
    std::vector<uint8_t> lhit;
    std::vector<RtMaterial> materials;
    std::vector<RtPattern> patterns;
    std::vector<RtLight> lights;
    std::vector<RtBvhNode> bvh_nodes;
    int32_t world_bvh_root = -1;
    uint32_t dir_res = 0;  // direction index over truncated conics (see RtDirCand)
    std::vector<RtDirCand> dir_cands;
    std::vector<uint32_t> dir_cell_start, dir_entries;

    RtSceneDesc desc() const;  // pointers into the vectors above; valid while *this lives unchanged
};

struct FlattenOptions {
    bool build_bvh = true;        // acceleration structures over group children / world shapes
    // Which hierarchy: 0 = by policy; 1 = nested (one per large group level, entered through the group's
    // gate); 2 = global (ONE hierarchy over every primitive / outermost CSG node of the world with boxes in
    // world space; a leaf nested in groups is reached through its ancestors' transforms and gates)
    int hierarchy = 0;
    size_t bvh_min_children = 8;  // groups smaller than this are scanned linearly
    uint32_t dir_res = 256;       // cube-map resolution of the direction index
    size_t dir_device_min = 64;   // from this many candidates on, only the candidates are listed: the cells are built on the device
};

FlatScene flatten(const World& world, const FlattenOptions& opt = FlattenOptions());
RtCameraDesc camera_desc(const Camera& camera);

}  // namespace rt
