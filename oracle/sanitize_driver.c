/* sanitize_driver.c -- TEST INFRASTRUCTURE.  The CPU restatement driven through one whole small regen (level-0 field -> refine ->
 * retain -> triangles -> weld) and a handful of point evaluations, built with -fsanitize=address,undefined by `make sanitize`
 * (tests/test_sanitizers.py): the GPU pool offers no device-side sanitizer, so the host-side code is what can be checked.
 * Prints the mesh counts and an FNV-1a-64 of the index buffer so that the test can also compare the sanitized build's RESULT
 * with the optimised library's. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } Point;
typedef struct { Point position, normal; } Vertex;
typedef struct { Vertex vertices[3]; } Triangle;

void orc_create_voxel_field(Point *voxels, Point *voxel_size);
void orc_refine_raw(int scene, const float *params, const Point *in, uint32_t n, Point voxel_size, Point *out);
size_t orc_retain_finite(Point *buf, size_t n);
void orc_triangles_raw(int scene, const float *params, const Point *vox, uint32_t n, Point voxel_size, Triangle *triangles, uint64_t *iter_hist);
size_t orc_weld(const Triangle *raw, size_t nslots, float *pos, float *nrm, uint32_t *idx, size_t *n_tris);
uint64_t orc_fnv1a64(const void *data, size_t n);
float orc_sdf(int scene, const float *params, float x, float y, float z);
void orc_set_threads(int n);

int main(int argc, char **argv) {
    const int scene = argc > 1 ? atoi(argv[1]) : 0, levels = argc > 2 ? atoi(argv[2]) : 1;
    const float params[4] = {0.25f, 0.0f, 0.0f, 0.0f};
    orc_set_threads(2);
    size_t n = 32u * 32u * 32u;
    Point *vox = (Point *) malloc(sizeof(Point) * n), size;
    orc_create_voxel_field(vox, &size);
    for (int l = 0; l < levels; l++) {
        Point *out = (Point *) malloc(sizeof(Point) * 8 * n);
        orc_refine_raw(scene, params, vox, (uint32_t) n, size, out);
        free(vox);
        vox = out;
        n = orc_retain_finite(vox, 8 * n);
        size.x /= 2.0f; size.y /= 2.0f; size.z /= 2.0f;
    }
    Triangle *raw = (Triangle *) malloc(sizeof(Triangle) * 5 * (n ? n : 1));
    orc_triangles_raw(scene, params, vox, (uint32_t) n, size, raw, NULL);
    const size_t nslots = 5 * n;
    float *pos = (float *) malloc(sizeof(float) * 9 * (nslots ? nslots : 1)), *nrm = (float *) malloc(sizeof(float) * 9 * (nslots ? nslots : 1));
    uint32_t *idx = (uint32_t *) malloc(sizeof(uint32_t) * 3 * (nslots ? nslots : 1));
    size_t nt = 0;
    const size_t nv = orc_weld(raw, nslots, pos, nrm, idx, &nt);
    float acc = 0.0f;
    for (int i = 0; i < 64; i++) acc += orc_sdf(scene, params, -2.5f + 0.078125f * (float) i, 0.3f, -0.2f);
    printf("%zu %zu %zu %016llx %a\n", n, nv, nt, (unsigned long long) orc_fnv1a64(idx, sizeof(uint32_t) * 3 * nt), (double) acc);
    free(vox); free(raw); free(pos); free(nrm); free(idx);
    return 0;
}
