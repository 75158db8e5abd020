/*
 * sdmesh/bindings.h -- FFI structs and constants of the mesh-generation boundary.
 *
 * Layout-identical to the mesh-generation part of the reference's shared header
 * cuda/includes/bindings.h:7-10 (constants) and :43-75 (structs), which bindgen turns into
 * src/bindings/cuda.rs (build.rs:139-168).  Sizes/offsets are asserted below against the
 * table in SURVEY.md 8(b).  The render structs (GlobalsBuffer, CameraBuffer, Rgba, RenderTexture,
 * bindings.h:12-41) are declared too: they are the kernargs of the ray-march preview (row N4).
 * Field names, C types, declaration order and the three constants are pinned against the reference's
 * header by tests/test_reference_pins.py (fixture tests/golden/bindings_ref.json, made by
 * tools/pin_mc_tables.py from the reference's own file).
 */
#ifndef SDMESH_BINDINGS_H
#define SDMESH_BINDINGS_H

#include <stddef.h>
#include <stdint.h>

#define BLOCK_SIZE 128                    /* bindings.h:7  launch width the reference host uses */
#define MESH_GENERATION_INIT_FACTOR 32    /* bindings.h:9  level-0 cells per axis */
#define MESH_GENERATION_BB_SIZE 5.0f      /* bindings.h:10 edge of the bounding cube (+-2.5) */

struct Point {                            /* bindings.h:43-47 */
    float x;
    float y;
    float z;
};

struct VoxelField {                       /* bindings.h:51-55; passed BY VALUE in kernarg */
    struct Point voxel_size;
    struct Point *voxels;
    unsigned int voxel_count;
};

struct Vertex {                           /* bindings.h:57-60 */
    struct Point position;
    struct Point normal;
};

struct Triangle {                         /* bindings.h:62-64 */
    struct Vertex vertices[3];
};

struct IndexTriangle {                    /* bindings.h:66-68 (declared, unused by the reference) */
    unsigned int indices[3];
};

struct Mesh {                             /* bindings.h:70-75 (declared, unused by the reference) */
    struct Vertex *vertices;
    unsigned int vertex_count;

    struct IndexTriangle *triangles;
};

/* ---- ray-march preview (row N4), cuda/includes/bindings.h:12-41 ---- */
enum RayMarchHitOutcome { Collision, StepLimit, DepthLimit };

struct GlobalsBuffer {                    /* bindings.h:16-21 */
    unsigned long long tick;
    float time;
    unsigned int render_texture_size[2];
    float render_screen_size[2];
};

struct CameraBuffer {                     /* bindings.h:23-29 */
    float position[3];
    float forward[3];
    float up[3];
    float right[3];
    float fov;
};

struct Rgba {                             /* bindings.h:31-36 */
    unsigned char r;
    unsigned char g;
    unsigned char b;
    unsigned char a;
};

struct RenderTexture {                    /* bindings.h:38-41 */
    unsigned int size[2];
    struct Rgba *data;
};

#ifdef __cplusplus
#define SDMESH_SA(c, m) static_assert(c, m)
#else
#define SDMESH_SA(c, m) _Static_assert(c, m)
#endif
SDMESH_SA(sizeof(struct Point) == 12 && offsetof(struct Point, y) == 4 && offsetof(struct Point, z) == 8, "Point");
SDMESH_SA(sizeof(struct VoxelField) == 32 && offsetof(struct VoxelField, voxels) == 16 &&
          offsetof(struct VoxelField, voxel_count) == 24, "VoxelField");
SDMESH_SA(sizeof(struct Vertex) == 24 && offsetof(struct Vertex, normal) == 12, "Vertex");
SDMESH_SA(sizeof(struct Triangle) == 72, "Triangle");
SDMESH_SA(sizeof(struct IndexTriangle) == 12, "IndexTriangle");
SDMESH_SA(sizeof(struct Mesh) == 24 && offsetof(struct Mesh, vertex_count) == 8 &&
          offsetof(struct Mesh, triangles) == 16, "Mesh");
SDMESH_SA(sizeof(struct GlobalsBuffer) == 32 && offsetof(struct GlobalsBuffer, render_texture_size) == 12 &&
          offsetof(struct GlobalsBuffer, render_screen_size) == 20, "GlobalsBuffer");
SDMESH_SA(sizeof(struct CameraBuffer) == 52 && offsetof(struct CameraBuffer, fov) == 48, "CameraBuffer");
SDMESH_SA(sizeof(struct Rgba) == 4, "Rgba");
SDMESH_SA(sizeof(struct RenderTexture) == 16 && offsetof(struct RenderTexture, data) == 8, "RenderTexture");
#undef SDMESH_SA

#endif /* SDMESH_BINDINGS_H */
