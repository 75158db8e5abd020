// render_device.hpp -- row N4: the ray-march preview's per-pixel code, shared by libsdmesh.so (k_render) and the
// reference-ABI code object (compute_render.hip).
#pragma once
#include "sdf.hpp"

namespace sdm {

// Row N4: ray-march preview (cuda/modules/compute_render.cu:21-97, common.cu:13-215, ray_marching.cu:15-49,
// color.cu:7-22).  One thread per pixel, 8 x 8 pixel workgroups; same operation order as the reference, with
// tan(fov / 2) supplied by the host (one value per frame; device and host tanf are not bit-identical).
struct RenderParams {          // GlobalsBuffer / CameraBuffer fields the kernel needs, flattened by the host
    float pos[3], fwd[3], up[3], right[3];
    float fov_fac, screen_x, screen_y, gtex_x, gtex_y;
    uint32_t width, height;
};

template <int SCENE>
__device__ __forceinline__ float sd_scene(f3 p, const SceneParams &sp) {          // compute_render.cu:3-19
    float sd = 3.40282347E+38f;
    sd = gmin(sd, sdf<SCENE>(p.x, p.y, p.z, sp));
    sd = gmin(sd, sqrt_rn_normal(skeleton_cube5_min_d2(p.x, p.y, p.z)) - 0.05f);     // = sd_box_skeleton(p, 0, (5, 5, 5), 0.05), see sdf.hpp
    return sd;
}

__device__ __forceinline__ f3 pixel_dir(float px, float py, const RenderParams &r) {
    const float tw = (float) r.width, th = (float) r.height;
    const float nx = (px + 0.5f) / tw, ny = (py + 0.5f) / th;
    const float cx = (2.0f * nx - 1.0f) * (tw / th), cy = 1.0f - 2.0f * ny;
    const float width_factor = (r.screen_x / r.gtex_x) * (r.gtex_y / r.screen_y);
    const f3 fw = mk(r.fwd[0], r.fwd[1], r.fwd[2]), up = mk(r.up[0], r.up[1], r.up[2]), rt = mk(r.right[0], r.right[1], r.right[2]);
    return normalize3(add3(add3(fw, scale3(up, cy * r.fov_fac)), scale3(rt, cx * r.fov_fac * width_factor)));
}

__device__ __forceinline__ f3 aces_tone(f3 hdr) {                                     // color.cu:7-22
    const f3 v = mk(0.59719f * hdr.x + 0.35458f * hdr.y + 0.04823f * hdr.z, 0.07600f * hdr.x + 0.90834f * hdr.y + 0.01566f * hdr.z,
                    0.02840f * hdr.x + 0.13383f * hdr.y + 0.83777f * hdr.z);
    const f3 a = mk(v.x * (v.x + 0.0245786f) - 0.000090537f, v.y * (v.y + 0.0245786f) - 0.000090537f, v.z * (v.z + 0.0245786f) - 0.000090537f);
    const f3 b = mk(v.x * (0.983729f * v.x + 0.4329510f) + 0.238081f, v.y * (0.983729f * v.y + 0.4329510f) + 0.238081f,
                    v.z * (0.983729f * v.z + 0.4329510f) + 0.238081f);
    const f3 q = mk(a.x / b.x, a.y / b.y, a.z / b.z);
    const f3 o = mk(1.60475f * q.x + -0.53108f * q.y + -0.07367f * q.z, -0.10208f * q.x + 1.10813f * q.y + -0.00605f * q.z,
                    -0.00327f * q.x + -0.07276f * q.y + 1.07602f * q.z);
    return mk(gmin(gmax(o.x, 0.0f), 1.0f), gmin(gmax(o.y, 0.0f), 1.0f), gmin(gmax(o.z, 0.0f), 1.0f));
}

template <int SCENE>
__device__ __forceinline__ uchar4 render_pixel(uint32_t x, uint32_t y, const RenderParams &r, const SceneParams &sp) {
    const float SQI = 0.7071067811865475f;
    const f3 dir = pixel_dir((float) x, (float) y, r);
    const f3 b0 = pixel_dir((float) x - SQI, (float) y - SQI, r), b1 = pixel_dir((float) x - SQI, (float) y + SQI, r);
    const f3 b2 = pixel_dir((float) x + SQI, (float) y - SQI, r), b3 = pixel_dir((float) x + SQI, (float) y + SQI, r);
    const f3 d0 = sub3(dir, b0), d1 = sub3(dir, b1), d2 = sub3(dir, b2), d3 = sub3(dir, b3);
    const float cone = gmax(gmax(sqrt_rn(dot3(d0, d0)), sqrt_rn(dot3(d1, d1))), gmax(sqrt_rn(dot3(d2, d2)), sqrt_rn(dot3(d3, d3))));
    f3 pos = mk(r.pos[0], r.pos[1], r.pos[2]);
    float depth = 0.0f;
    int outcome = 1;                                                                  // StepLimit
    for (int steps = 0; steps < 256; steps++) {                                       // ray_marching.cu:15-49
        const float cd = cone * depth;
        const float d = sd_scene<SCENE>(pos, sp);
        if (d <= cd + 0.001f) { outcome = 0; break; }
        depth += (d - cd);
        pos = add3(pos, scale3(dir, d - cd));
        if (depth > 500.0f) { outcome = 2; break; }
    }
    f3 color = mk(0.0f, 0.0f, 0.0f);
    if (outcome == 0) {
        const float e1 = SDM_NORMAL_EPS, e2 = 2.0f * SDM_NORMAL_EPS;
#define SDM_F(dx, dy, dz) sd_scene<SCENE>(mk(pos.x + (dx), pos.y + (dy), pos.z + (dz)), sp)
        const float gx = (-SDM_F(e2, 0.0f, 0.0f) + 8.0f * SDM_F(e1, 0.0f, 0.0f) - 8.0f * SDM_F(-e1, 0.0f, 0.0f) + SDM_F(-e2, 0.0f, 0.0f));
        const float gy = (-SDM_F(0.0f, e2, 0.0f) + 8.0f * SDM_F(0.0f, e1, 0.0f) - 8.0f * SDM_F(0.0f, -e1, 0.0f) + SDM_F(0.0f, -e2, 0.0f));
        const float gz = (-SDM_F(0.0f, 0.0f, e2) + 8.0f * SDM_F(0.0f, 0.0f, e1) - 8.0f * SDM_F(0.0f, 0.0f, -e1) + SDM_F(0.0f, 0.0f, -e2));
#undef SDM_F
        const f3 n = normalize3(mk(gx, gy, gz));
        const f3 light = normalize3(mk(1.0f, 1.0f, 1.0f));
        const float t = (dot3(n, light) + 1.0f) / 2.0f;
        const f3 c1 = mk(19.0f / 255.0f, 9.0f / 255.0f, 130.0f / 255.0f), c2 = mk(240.0f / 255.0f, 103.0f / 255.0f, 24.0f / 255.0f);
        color = add3(scale3(c1, 1.0f - t), scale3(c2, t));
    } else if (outcome == 1) {
        color = mk(1.0f, 1.0f, 1.0f);
    }
    color = aces_tone(color);
    uchar4 px;
    px.x = (unsigned char) (gmin(gmax(color.x, 0.0f), 1.0f) * 255.0f);
    px.y = (unsigned char) (gmin(gmax(color.y, 0.0f), 1.0f) * 255.0f);
    px.z = (unsigned char) (gmin(gmax(color.z, 0.0f), 1.0f) * 255.0f);
    px.w = 0xFF;
    return px;
}


}  // namespace sdm
