// compute_render.hip -- boundary B1 for row N4: a gfx950 code object that can stand in for the reference's PTX module
// "compute_render" (src/cuda/mod.rs:68-90, launched at :377-400).
//
// Same extern "C" symbol, same by-value kernargs (RenderTexture, GlobalsBuffer, CameraBuffer), same launch shape
// (width * height / BLOCK_SIZE blocks of BLOCK_SIZE threads) and the same thread -> pixel map
// (render_texture_coord, cuda/modules/common.cu:180-212: every 32 consecutive threads cover a 4 x 8 pixel patch, a
// block 8 x 16 pixels).  The scene is the reference's compiled-in sd_obj.  Unlike sdmesh_render, this kernel has to
// evaluate tan(fov / 2) on the device (the ABI carries fov, not its tangent), so against the CPU oracle it is equal up
// to the last bit of that one value -- in practice the same image (tests/test_parity_gpu.py).
#include <hip/hip_runtime.h>

#include "../../include/sdmesh/bindings.h"
#include "render_device.hpp"

#pragma clang fp contract(off)

using namespace sdm;

extern "C" __global__ void compute_render(const RenderTexture render_texture, const GlobalsBuffer globals, const CameraBuffer camera) {
    // render_texture_coord (common.cu:180-212)
    const int WARP_H = 8, WARP_W = 4, BLOCK_WARP_H = 2, BLOCK_WARP_W = (BLOCK_SIZE / BLOCK_WARP_H) / 32;
    const int warp_local_id = (int) (threadIdx.x % 32u), warp_id = (int) (threadIdx.x / 32u);
    const int lx = warp_local_id % WARP_W + WARP_W * (warp_id % BLOCK_WARP_W);
    const int ly = warp_local_id / WARP_W + WARP_H * (warp_id / BLOCK_WARP_W);
    const int bcx = (int) render_texture.size[0] / (WARP_W * BLOCK_WARP_W);
    if (bcx <= 0) return;
    const int x = (WARP_W * BLOCK_WARP_W) * ((int) blockIdx.x % bcx) + lx;
    const int y = (WARP_H * BLOCK_WARP_H) * ((int) blockIdx.x / bcx) + ly;
    if ((unsigned) y >= render_texture.size[1] || (unsigned) x >= render_texture.size[0]) return;
    RenderParams r;
    for (int i = 0; i < 3; i++) { r.pos[i] = camera.position[i]; r.fwd[i] = camera.forward[i]; r.up[i] = camera.up[i]; r.right[i] = camera.right[i]; }
    r.fov_fac = tanf(camera.fov / 2);                                   // common.cu:86
    r.screen_x = globals.render_screen_size[0]; r.screen_y = globals.render_screen_size[1];
    r.gtex_x = (float) globals.render_texture_size[0]; r.gtex_y = (float) globals.render_texture_size[1];
    r.width = render_texture.size[0]; r.height = render_texture.size[1];
    SceneParams sp;
    sp.p[0] = sp.p[1] = sp.p[2] = sp.p[3] = 0.0f;
    sp.ninstr = 0;
    const uchar4 px = render_pixel<SCENE_SDOBJ>((uint32_t) x, (uint32_t) y, r, sp);
    Rgba o;
    o.r = px.x; o.g = px.y; o.b = px.z; o.a = px.w;
    render_texture.data[(size_t) y * render_texture.size[0] + (size_t) x] = o;      // index_2d (common.cu:33-36)
}
