// compute_mesh_generation.hip -- boundary B1: a gfx950 code object that can stand in for the
// reference's PTX module "compute_mesh_generation" (src/cuda/mod.rs:68-90).
//
// Same extern "C" symbol names, same by-value kernarg layouts, same 1-D launch shape
// (ceil(n / BLOCK_SIZE) x BLOCK_SIZE, no dynamic LDS -- src/cuda/mod.rs:149-177, :226-250), same
// output layouts (8n Points with INFINITY, 5n Triangle slots with NaN/zero padding), so a host
// that swaps cudarc's load_ptx/get_func/launch for hipModuleLoadData / hipModuleGetFunction /
// hipModuleLaunchKernel needs no other change.  The scene is the reference's compiled-in sd_obj
// (cuda/modules/common.cu:222-226).
//
//   compute_refine_voxel_field_by_sdf                     compute_mesh_generation.cu:12-62
//   compute_surface_triangles_from_voxel_field_by_sdf     compute_mesh_generation.cu:64-120
//
// Because the caller fixes one thread per voxel, this module cannot share edge projections across
// workgroups the way libsdmesh.so's device-resident path does; it is the compatibility door, not
// the fast path.
#include <hip/hip_runtime.h>

#include "../../include/sdmesh/bindings.h"
#include "mesh_device.hpp"

#pragma clang fp contract(off)

using namespace sdm;

extern "C" __global__ void compute_refine_voxel_field_by_sdf(const VoxelField input_field, VoxelField output_field) {
    const unsigned int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= input_field.voxel_count) return;
    // (the reference's store to the by-value output_field.voxel_size at :25-27 is dead: not reproduced)
    Point cs;
    cs.x = input_field.voxel_size.x / 2.0f;
    cs.y = input_field.voxel_size.y / 2.0f;
    cs.z = input_field.voxel_size.z / 2.0f;
    const Point b = input_field.voxels[id];
    SceneParams sp;
    sp.p[0] = sp.p[1] = sp.p[2] = sp.p[3] = 0.0f;
    const uint32_t m = refine_child_mask<SCENE_SDOBJ>(b, cs, sp);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const unsigned int n_id = id * 8u + (unsigned int) c;
        if (n_id < output_field.voxel_count) {
            Point q;
            if (m & (1u << c)) {
                q.x = b.x + (float) (c >> 2) * cs.x;
                q.y = b.y + (float) ((c >> 1) & 1) * cs.y;
                q.z = b.z + (float) (c & 1) * cs.z;
            } else {
                q.x = q.y = q.z = __builtin_inff();
            }
            output_field.voxels[n_id] = q;
        }
    }
}

extern "C" __global__ void compute_surface_triangles_from_voxel_field_by_sdf(VoxelField field, Triangle *triangles) {
    const unsigned int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= field.voxel_count) return;
    SceneParams sp;
    sp.p[0] = sp.p[1] = sp.p[2] = sp.p[3] = 0.0f;
    voxel_triangles_direct<SCENE_SDOBJ>(field.voxels[id], field.voxel_size, sp, triangles + 5u * (size_t) id);
}
