// cuda_handler.hpp -- the reference's host interface for the mesh-generation path, as a compiled-language (C++17, header-only)
// mirror above the C ABI of libsdmesh.so.
//
// The reference's host side is Rust (src/cuda/mod.rs) and Rust is not in the build image, so the drop-in binding a maintainer
// would use lives under integration/ uncompiled.  This header is the same surface in the compiled language the image does have:
// same type and method names, same argument meaning, same ownership (host vectors owned by the caller's CudaVoxelField, device
// buffers owned by the handler, grow-only) and the same error behaviour -- the reference `.unwrap()`s every failure into a panic
// (src/cuda/mod.rs:52-100,149-177), here every failure throws sdmesh::Panic.  Nothing in here computes: each method is one call
// into the C ABI (include/sdmesh.h), which fails loudly without a gfx950 device -- there is no CPU fallback.
//
//   CudaHandler::CudaHandler()                          CudaHandler::new()                           src/cuda/mod.rs:49
//   CudaHandler::create_cuda_voxel_field()   (static)   CudaHandler::create_cuda_voxel_field()       :105
//   handler.refine_voxel_field(field)                   refine_voxel_field(&mut self, &mut field)    :124   in place: voxels <- kept children, size /= 2
//   handler.voxel_field_to_mesh(field) -> ObjData       voxel_field_to_mesh(&mut self, &field)       :204
//   handler.render(target, globals, camera)             render(&mut self, &mut [u8], globals, camera) :348
//   ~CudaHandler()                                      Drop
//   ObjData                                             obj::ObjData as mod.rs:303-326 fills it (position, normal, texture = [[0, 0]],
//                                                       one object "default" / one group "default", polygons (i, Some(0), Some(i)))
//   ObjData::save(path)                                 obj.save(path)                               src/renderer/mod.rs:204
#pragma once

#include <array>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sdmesh.h"

namespace sdmesh {

struct Panic : std::runtime_error {          // what a Rust `.unwrap()` on an Err becomes here
    int code;
    Panic(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

struct CudaVoxelField {                      // src/cuda/mod.rs:41-46
    std::vector<Point> voxels;
    Point voxel_size;
};

struct IndexTuple {                          // obj::IndexTuple(position, Option<texture>, Option<normal>): both options always Some here
    size_t position, texture, normal;
};
typedef std::array<IndexTuple, 3> SimplePolygon;

struct ObjData {                             // the fields of obj::ObjData the reference fills (src/cuda/mod.rs:303-326)
    std::vector<std::array<float, 3>> position;
    std::vector<std::array<float, 3>> normal;
    std::vector<std::array<float, 2>> texture{{0.0f, 0.0f}};
    std::string object_name = "default", group_name = "default";
    std::vector<SimplePolygon> polys;

    // Wavefront text as obj-rs writes it for this content: v / vt / vn lists, one object, one group, `f i/1/i j/1/j k/1/k`, 1-based.
    // Floats with 9 significant digits: loading the file restores every f32 bit for bit.
    void save(const std::string &path) const {
        FILE *f = std::fopen(path.c_str(), "w");
        if (!f) throw Panic(SDMESH_ERR_INVALID, "cannot open " + path);
        for (const auto &p : position) std::fprintf(f, "v %.9g %.9g %.9g\n", p[0], p[1], p[2]);
        for (const auto &t : texture) std::fprintf(f, "vt %.9g %.9g\n", t[0], t[1]);
        for (const auto &n : normal) std::fprintf(f, "vn %.9g %.9g %.9g\n", n[0], n[1], n[2]);
        std::fprintf(f, "o %s\ng %s\n", object_name.c_str(), group_name.c_str());
        for (const auto &t : polys)
            std::fprintf(f, "f %zu/%zu/%zu %zu/%zu/%zu %zu/%zu/%zu\n", t[0].position + 1, t[0].texture + 1, t[0].normal + 1, t[1].position + 1,
                         t[1].texture + 1, t[1].normal + 1, t[2].position + 1, t[2].texture + 1, t[2].normal + 1);
        std::fclose(f);
    }
};

class CudaHandler {
  public:
    CudaHandler() { unwrap(sdmesh_create(0, &h_), "sdmesh_create"); }                       // CudaDevice::new(0), mod.rs:52
    ~CudaHandler() { if (h_) sdmesh_destroy(h_); }
    CudaHandler(const CudaHandler &) = delete;
    CudaHandler &operator=(const CudaHandler &) = delete;

    static CudaVoxelField create_cuda_voxel_field() {                                        // mod.rs:105-122
        CudaVoxelField f;
        f.voxels.resize((size_t) MESH_GENERATION_INIT_FACTOR * MESH_GENERATION_INIT_FACTOR * MESH_GENERATION_INIT_FACTOR);
        const int rc = sdmesh_create_voxel_field(f.voxels.data(), &f.voxel_size);
        if (rc != SDMESH_OK) throw Panic(rc, "sdmesh_create_voxel_field failed");
        return f;
    }

    void refine_voxel_field(CudaVoxelField &field) {                                         // mod.rs:124-202
        if (field.voxels.empty()) return;                                                    // :137: nothing happens, not even the size halves
        std::vector<Point> out(8 * field.voxels.size());                                     // the kernel's 8 child slots per voxel (:141)
        size_t n = 0;
        Point size;
        unwrap(sdmesh_refine_voxel_field(h_, field.voxels.data(), field.voxels.size(), field.voxel_size, out.data(), out.size(), &n, &size),
               "sdmesh_refine_voxel_field");
        out.resize(n);                                                                       // `retain` of the finite children (:192-193)
        field.voxels.swap(out);
        field.voxel_size = size;
    }

    ObjData voxel_field_to_mesh(const CudaVoxelField &field) {                               // mod.rs:204-345
        ObjData obj;
        if (field.voxels.empty()) return obj;                                                // :327-344: the empty ObjData
        size_t nv = 0, nt = 0;
        unwrap(sdmesh_voxel_field_to_mesh(h_, field.voxels.data(), field.voxels.size(), field.voxel_size, nullptr, 0, &nv, nullptr, 0, &nt),
               "sdmesh_voxel_field_to_mesh");
        std::vector<Vertex> verts(nv);
        std::vector<IndexTriangle> tris(nt);
        unwrap(sdmesh_download_mesh(h_, verts.data(), tris.data()), "sdmesh_download_mesh");
        obj.position.resize(nv);
        obj.normal.resize(nv);
        for (size_t i = 0; i < nv; i++) {
            obj.position[i] = {verts[i].position.x, verts[i].position.y, verts[i].position.z};
            obj.normal[i] = {verts[i].normal.x, verts[i].normal.y, verts[i].normal.z};
        }
        obj.polys.resize(nt);
        for (size_t t = 0; t < nt; t++)
            for (int k = 0; k < 3; k++) obj.polys[t][k] = IndexTuple{tris[t].indices[k], 0, tris[t].indices[k]};     // (idx, Some(0), Some(idx)), :316-321
        return obj;
    }

    void render(uint8_t *target_rgba, size_t target_len, const GlobalsBuffer &globals, const CameraBuffer &camera) {   // mod.rs:348-400
        const size_t need = (size_t) globals.render_texture_size[0] * globals.render_texture_size[1] * 4;
        if (target_len < need) throw Panic(SDMESH_ERR_CAPACITY, "render target too small");
        unwrap(sdmesh_render(h_, &globals, &camera, reinterpret_cast<Rgba *>(target_rgba)), "sdmesh_render");
    }

    sdmesh_handle *raw() const { return h_; }                                                // the build's own extensions (sdmesh_regen ...) take this

  private:
    void unwrap(int rc, const char *what) const {
        if (rc != SDMESH_OK) throw Panic(rc, std::string(what) + ": " + (h_ ? sdmesh_last_error(h_) : "no handle"));
    }
    sdmesh_handle *h_ = nullptr;
};

}  // namespace sdmesh
