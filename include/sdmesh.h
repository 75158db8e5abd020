/*
 * sdmesh.h -- C ABI of libsdmesh.so, the MI355X (gfx950) replacement for the mesh-generation
 * half of the reference's `CudaHandler` (src/cuda/mod.rs:25-346).
 *
 * Every entry point names the reference interface it replaces.  Plain pointers and sizes only;
 * all functions return 0 (SDMESH_OK) or an SDMESH_ERR_* code and never abort (the reference
 * `.unwrap()`s every driver call -- SURVEY.md section 5).  A handle is single-threaded, like the
 * reference handler that lives in a Bevy NonSend resource (src/renderer/mod.rs:159,233).
 *
 * Two families:
 *   host-buffer calls   1:1 with the reference methods: host vectors in, host vectors out,
 *                       fully synchronous (src/cuda/mod.rs:142-190, :219-261).
 *   device-resident     the whole level-0 -> refine^k -> mesh -> weld chain without leaving
 *                       HBM; results are read back (or handed to a renderer / RCCL) afterwards.
 */
#ifndef SDMESH_H
#define SDMESH_H

#include <stddef.h>
#include <stdint.h>

#include "sdmesh/bindings.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SDMESH_OK = 0,
    SDMESH_ERR_INVALID = 1,      /* bad argument */
    SDMESH_ERR_HIP = 2,          /* a HIP runtime call failed (sdmesh_last_error has the text) */
    SDMESH_ERR_NO_DEVICE = 3,    /* no usable gfx950 device */
    SDMESH_ERR_CAPACITY = 4,     /* caller-provided output buffer too small */
    SDMESH_ERR_WELD_RANGE = 5,   /* single-GPU calls never return it: vertices outside the packed +-10.48 unit key are welded with the
                                    wide key form, like the reference's [i64; 3] (src/cuda/mod.rs:268-273).  Only the multi-GPU forms
                                    that match vertices across ranks by the packed key return it (the distributed weld, the indexed
                                    z-slab gather -- whose raw form, option "shard_indexed" = 0, is the way out) */
    SDMESH_ERR_INTERNAL = 6,     /* invariant violated on the device (bug) */
    SDMESH_ERR_RCCL = 7,         /* librccl missing or a collective failed (sdmesh_last_error has the text); the communicator is
                                    unusable afterwards */
    SDMESH_ERR_PEER = 8          /* a collective call failed on ANOTHER rank (which returns its own error code): every rank of the
                                    communicator returns an error from the same call, nobody is left waiting; the communicator
                                    stays usable */
};

/* Scenes.  The reference hard-codes one SDF, sd_obj (cuda/modules/common.cu:222-226) = SDOBJ.
 * SPHERE is its a2 term alone; CSG3 and ANIM are build-defined. */
enum {
    SDMESH_SCENE_SDOBJ = 0,
    SDMESH_SCENE_SPHERE = 1,
    SDMESH_SCENE_CSG3 = 2,
    SDMESH_SCENE_ANIM = 3,       /* params[0] = y of the sphere centre (host passes 0.5*sin(t)) */
    /* 4 is taken by the runtime program (sdmesh_set_program) */
    SDMESH_SCENE_MANDELBULB = 5  /* the reference's sd_mandelbulb (cuda/includes/signed_distance.cu:29-53, never called there) in
                                  * sd_unit_mandelbulb's form (:55-57): sd_mandelbulb(p / params[0], params[1]) * params[0];
                                  * params[0] = scale (1 without params; the reference's unit bulb is 0.4), params[1] = time (the
                                  * power is 7 (1 + 0.001 time)).  acos / atan2 / pow / sin / cos / log are the library's own f64
                                  * evaluations rounded once to f32 (csrc/sdf.hpp): the same bits on host and device, within 1 ulp
                                  * of a correctly rounded libm, NOT the bits of CUDA's fast-math functions. */
};

/* Multi-GPU partition of the level-0 field. */
enum {
    SDMESH_PART_NONE = 0,
    SDMESH_PART_ZSLAB = 1,       /* level-0 z layers [32*part/nparts, 32*(part+1)/nparts) */
    SDMESH_PART_XSLAB = 2        /* level-0 x layers, same bounds: a contiguous range of the unpartitioned order */
};

typedef struct sdmesh_handle sdmesh_handle;

#define SDMESH_MAX_LEVELS 15

typedef struct sdmesh_regen_info {
    uint32_t levels;                                   /* refines performed */
    uint32_t voxel_count[SDMESH_MAX_LEVELS + 1];       /* active voxels entering level i */
    uint32_t unique_edges;                             /* projected vertices before the weld */
    uint32_t triangles;
    uint32_t vertices;                                 /* welded */
    struct Point voxel_size;                           /* of the final level */
    uint32_t status[8];                                /* device status block (all 0 = healthy) */
    uint32_t used_direct_path;                         /* 1 if the non-lattice fallback ran */
    float device_ms;                                   /* HIP-event time of the device chain; 0 for a planned (repeat) regen unless
                                                        * option "timing" = 1 (the event pair costs ~9 us of wall time per regen) */
} sdmesh_regen_info;

/* ---- lifetime ----------------------------------------------------------------------------- */

/* CudaHandler::new (src/cuda/mod.rs:49-103): picks the device, loads the kernels, creates the
 * grow-only device buffers (DynamicCudaSlice, :10-23).  The first handle a process makes on a device also runs
 * sdmesh_selfcheck_arithmetic there (strict build; about a second; SDMESH_SKIP_SELFCHECK=1 in the environment skips it) and the call
 * fails with SDMESH_ERR_INTERNAL on a part whose v_rsq_f32 / v_rcp_f32 do not make the lean square root / reciprocal exact. */
int sdmesh_create(int device, sdmesh_handle **out);
void sdmesh_destroy(sdmesh_handle *h);
const char *sdmesh_last_error(const sdmesh_handle *h);

/* The reference compiles its scene in; here it is a runtime choice. */
int sdmesh_set_scene(sdmesh_handle *h, int scene, const float *params, int nparams);
/* Runtime SDF programs (the reference has exactly one compiled-in function).  A program is a
 * linear CSG chain: acc = combine_i(acc, primitive_i(p)), i = 0..n-1, the first combine must be SET.
 *   SPHERE        a = {cx, cy, cz, r}                      length(p - c) - r
 *   BOX           a = {bpx, bpy, bpz, bsx, bsy, bsz}       sd_box            (cuda/includes/signed_distance.cu:86-91)
 *   BOX_SKELETON  a = {bp xyz, bs xyz, lw}                 sd_box_skeleton   (signed_distance.cu:93-113, bug included)
 *   CAPSULE       a = {b0 xyz, b1 xyz, lw}                 sd_line - lw      (signed_distance.cu:77-80)
 *   TORUS_Y       a = {cx, cy, cz, R, r}                   build-defined
 *   combine: SET, MIN (union), MAX (intersection), SMOOTH_MIN with radius k (signed_distance.cu:20-23), SUBTRACT
 * The reference scene is {BOX_SKELETON (0,0,0) (3,1,.5) .1 SET; SPHERE (0,0,0) 1 SMOOTH_MIN k=.5} and gives results
 * bit-identical to SDMESH_SCENE_SDOBJ. */
enum { SDMESH_OP_SPHERE = 1, SDMESH_OP_BOX = 2, SDMESH_OP_BOX_SKELETON = 3, SDMESH_OP_CAPSULE = 4, SDMESH_OP_TORUS_Y = 5,
       SDMESH_OP_RAY = 6 /* a[0..2] origin, a[3..5] direction, a[6] radius: sd_ray(p, bl, bd) - r, the unbounded ray (signed_distance.cu:61-63) */,
       SDMESH_OP_SIMPLE_BOX = 7 /* a[0..2] centre, a[3..5] size: sd_simple_box (:115-118) */,
       SDMESH_OP_SIMPLE_BOUNDING_BOX = 8 /* a[0..2] bb_min, a[3..5] bb_max: sd_simple_bounding_box (:120-131) */,
       SDMESH_OP_WRAP = 9 /* a[0..2] lower, a[3..5] higher: every LATER instruction sees wrap(p, lower, higher) (:9-18); combine / k unused */ };
enum { SDMESH_CB_SET = 0, SDMESH_CB_MIN = 1, SDMESH_CB_MAX = 2, SDMESH_CB_SMOOTH_MIN = 3, SDMESH_CB_SUBTRACT = 4 };
#define SDMESH_MAX_INSTR 16
typedef struct sdmesh_instr {
    uint32_t op;
    uint32_t combine;
    float k;
    float a[7];
} sdmesh_instr;
int sdmesh_set_program(sdmesh_handle *h, const sdmesh_instr *instrs, int n);
/* Time-varying programs ("time-varying CSG", BASELINE configs[4] as a program): per instruction seven rates da[i][0..7); the arguments the
 * kernels see are a[i][j] + time * da[i][j], evaluated on the host in f32 (one multiplication, one addition, no contraction) whenever the
 * time changes -- the device interpreter is unchanged and a frame costs one 700-byte uniform upload.  motion == NULL clears the rates.
 * sdmesh_set_program resets rates and time.  Sizes may change with the time: the capacity plan follows as for any scene change. */
int sdmesh_set_program_motion(sdmesh_handle *h, const float *motion /* n x 7 */, int n);
int sdmesh_set_program_time(sdmesh_handle *h, float time);

/* Run on a caller-owned hipStream_t, taken exactly as HIP reads it: NULL is the legacy default stream (which is what
 * torch.cuda.current_stream().cuda_stream is for torch's default stream).  hipGraph replay needs a capturable stream, so
 * on the NULL stream the "graph" option falls back to eager launches.  sdmesh_use_own_stream returns to the handle's own
 * (non-blocking) stream, the state after sdmesh_create. */
int sdmesh_set_stream(sdmesh_handle *h, void *hip_stream);
int sdmesh_use_own_stream(sdmesh_handle *h);
/* Options (all have measured defaults; the others exist for A/B runs and are covered by the parity tests):
 *   "direct" = 1        force the per-corner projection path (no edge sharing)
 *   "skip_weld" = 1     sdmesh_regen* stops after the orientation test (a multi-GPU rank exports
 *                       sdmesh_device_triangles and the root welds)
 *   "graph" = 1         replay the planned chain as a hipGraph (default 0: eager launches)
 *   "grid", "tgrid"     workgroups of the grid-stride kernels / of the per-tile kernels
 *   "project_lanes", "flip_lanes" = 0|1|4   lanes per vertex / triangle (0 = by planned item count)
 *   "project_mult", "flip_mult"             grid multipliers of those two kernels
 *   "emit_ci" = 0       run k_classify instead of taking the MC cases from the last refine level
 *   "fork" = 0|1|2      weld claim after / on a second stream beside / in one dispatch with the orientation test
 *   "ff_fork" = 0|1|2   weld-table clear inside k_create_field / on a second stream / as a memset node
 *   "col_refine" = 0    refine level by level from the level-0 field (default 1: levels 0 .. min(levels - 1, 3) are fused
 *                       per level-0 column, two launches in all)
 *   "refine2" = 1, "mask_chunk" = 8|16|32   refine variants
 *   "dense_q1" = 0      dense sample pass (P1) of the flat kernels in the persistent shape (default 1: one float4 store per thread,
 *                       one 4 KiB piece per workgroup -- the shape the HBM write stream is fastest in; volumes of < 2^32 samples)
 *   "dense_grid", "dense_nt", "dense_rows", "dense_contig"   launch shape of the dense sample pass (P1); dense_nt = 0 | 1 | 2:
 *                       nontemporal stores never / in the flat kernels and the plain row kernel (default) / in the aligned row kernel too
 *   "dense_aligned" = 0 sd_obj family rows through the plain row kernel (default 1: k_sample_dense_rows_aligned -- four same-phase
 *                       rows per wave, z terms hoisted, 16-byte aligned stores; needs a 16-byte aligned output, else falls back)
 *   "dense_xcd" = 0|1|2 an XCD's workgroups write one contiguous eighth of the output instead of every eighth piece: never / in the aligned row
 *                       kernel (default; 6-17 % there) / in every P1 kernel (no difference in the others)
 *   "dense_occ" = n     resident workgroups per CU of the aligned row kernel: 0 = by its shape (default), 1..8 = capped at n through an LDS
 *                       allotment the kernel never touches, < 0 = not capped
 *   "forget_plan" = 1   drop the capacity plan (the next regen discovers sizes again)
 *   "fast_math" = 0|1   confirms the numeric mode of the loaded library: libsdmesh.so is strict IEEE (bit-exact against the
 *                       CPU oracle), libsdmesh_fast.so is the same code built like the reference's `nvcc --use_fast_math`
 *                       (FMA contraction, approximate divide / sqrt, denormals flushed): link or load that file to opt in;
 *                       the call fails if the value does not match the build
 *   "claim" = 0         weld claims beside the orientation test instead of in the projection kernel's epilogue
 *   "weld_shift" = n    locality-preserving weld home slots (measured slower; default 0 = uniform)
 *   "col_segs" = n      z segments per level-0 column in the column kernels (1 / 2 / 4 / 8; default 1: measured fastest)
 *   "tf_threads" = n    workgroup size of k_tile_first when the orientation test runs inside it (128 / 256 / 512 / 768 / 1024; default 512)
 *   "project_batch", "project_grid", "project_prio"   A/B knobs of the projection kernel (dealing unit, grid, wave priorities;
 *                       measured, defaults are the fastest)
 *   "timing" = 1        bracket the planned chain with an event pair: sdmesh_regen_info.device_ms (default 0)
 *   "local_prefix" = 0  always launch the single-workgroup prefix kernels (default 1: up to 2 048 tiles the consumers'
 *                       workgroups prefix the ~1 000 tile counts themselves in LDS, two launches fewer)
 *   "publish_ctl" = 0   read the counters back with a device-to-host copy behind the chain (default 1: the chain's last
 *                       kernel writes them to the handle's pinned host block itself)
 *   "profile" = 1       time every stage of the planned chain with HIP events (sdmesh_profile_stages)
 *   "shard_indexed" = 0 sdmesh_sharded_regen in its raw form (unwelded slabs, triangle lists to the root, weld there); default 1: indexed form
 *   "shard_planned" = 0 every indexed sdmesh_sharded_regen step on the synchronous route (default 1: planned steps in steady state)
 *   "weld_wide" = 1     always weld with the unbounded key form (default 0: packed 3 x 21-bit keys, and the wide form
 *                       automatically from the first vertex that does not fit them) */
int sdmesh_set_option(sdmesh_handle *h, const char *name, int value);

/* ---- host-buffer API (B2) ----------------------------------------------------------------- */

/* CudaHandler::create_cuda_voxel_field (src/cuda/mod.rs:105-122).  voxels: 32^3 = 32768 Points,
 * x outer / z inner.  Pure host function, needs no handle and no GPU. */
int sdmesh_create_voxel_field(struct Point *voxels, struct Point *voxel_size);

/* CudaHandler::refine_voxel_field (src/cuda/mod.rs:124-202): kernel + order-preserving retain.
 * out needs room for up to 8*n Points (out_cap); *n_out receives the kept count, *size_out the
 * halved voxel size.  n == 0 leaves everything untouched, as the reference does (:137). */
int sdmesh_refine_voxel_field(sdmesh_handle *h, const struct Point *in, size_t n, struct Point voxel_size,
                              struct Point *out, size_t out_cap, size_t *n_out, struct Point *size_out);

/* The refine kernel's own output (compute_mesh_generation.cu:12-62): 8*n Points, INFINITY where a
 * child is not a border voxel. */
int sdmesh_refine_voxel_field_raw(sdmesh_handle *h, const struct Point *in, size_t n, struct Point voxel_size,
                                  struct Point *out8n);

/* The triangle kernel's own output (compute_mesh_generation.cu:64-120): 5*n Triangle slots,
 * unused slots = {NaN,NaN} in vertex 0 and zeros in vertices 1..2. */
int sdmesh_voxel_field_to_triangles_raw(sdmesh_handle *h, const struct Point *voxels, size_t n,
                                        struct Point voxel_size, struct Triangle *out5n);

/* CudaHandler::voxel_field_to_mesh (src/cuda/mod.rs:204-346) without the obj::ObjData wrapping:
 * welded vertices (first-seen position AND normal) and u32 index triples in emission order.
 * Worst case 15*n vertices / 5*n triangles; pass verts == NULL / tris == NULL to get the counts
 * only and fetch the data with sdmesh_download_mesh. */
int sdmesh_voxel_field_to_mesh(sdmesh_handle *h, const struct Point *voxels, size_t n, struct Point voxel_size,
                               struct Vertex *verts, size_t vert_cap, size_t *n_verts,
                               struct IndexTriangle *tris, size_t tri_cap, size_t *n_tris);

/* The host weld alone (src/cuda/mod.rs:263-296) on a raw slot list that lives on the HOST. */
int sdmesh_weld_triangles(sdmesh_handle *h, const struct Triangle *raw, size_t nslots, struct Vertex *verts,
                          size_t vert_cap, size_t *n_verts, struct IndexTriangle *tris, size_t tri_cap,
                          size_t *n_tris);

/* ---- device-resident API ------------------------------------------------------------------- */

/* One user-visible regen (src/renderer/mod.rs:162-225: create, `levels` x refine, mesh) entirely on
 * the device.  part/nparts/mode select this rank's share of the level-0 field. */
int sdmesh_regen(sdmesh_handle *h, int levels, sdmesh_regen_info *info);
int sdmesh_regen_part(sdmesh_handle *h, int levels, int mode, int part, int nparts, sdmesh_regen_info *info);

/* Split form for keeping several regens in flight (one handle each): _begin enqueues the planned device chain on the
 * handle's streams and returns without waiting (with no capacity plan yet it only records the request); _end waits,
 * validates and returns the counts -- or falls back to the synchronous path if a capacity was exceeded.  Most kernels of
 * the chain are latency-bound at low occupancy, so two handles in flight overlap well (bench.py).  The handle must not
 * be touched between the two calls. */
int sdmesh_regen_begin(sdmesh_handle *h, int levels, int mode, int part, int nparts);
int sdmesh_regen_end(sdmesh_handle *h, sdmesh_regen_info *info);

/* Borrow the results of the last regen / mesh call (device pointers owned by the handle, valid
 * until the next call).  Fills the reference's otherwise unused `struct Mesh` (bindings.h:70-75). */
int sdmesh_device_mesh(sdmesh_handle *h, struct Mesh *mesh, uint32_t *n_tris);
int sdmesh_device_voxels(sdmesh_handle *h, struct VoxelField *field);
/* Compact (un-welded) triangle list of the last regen in emission order, 72 B each: written into a
 * caller-provided DEVICE buffer (e.g. a torch tensor that RCCL then gathers). */
int sdmesh_device_triangles(sdmesh_handle *h, struct Triangle *d_out, size_t cap, size_t *n_tris);
/* Weld a compact/raw slot list that already lives on the DEVICE (root of the multi-GPU gather). */
int sdmesh_weld_device_triangles(sdmesh_handle *h, const struct Triangle *d_raw, size_t nslots,
                                 sdmesh_regen_info *info);

/* Triangles of the last regen per level-0 (x0, y0) column, index x0 * 32 + y0, written to a DEVICE array of
 * 32 * 32 u32.  With every rank's counts the z-slab lists can be merged back into the single-GPU order
 * (z0 is the least significant level-0 digit of the order key, so slabs interleave per column). */
int sdmesh_column_triangle_counts(sdmesh_handle *h, uint32_t *d_counts);

int sdmesh_download_mesh(sdmesh_handle *h, struct Vertex *verts, struct IndexTriangle *tris);
int sdmesh_download_voxels(sdmesh_handle *h, struct Point *voxels);

/* ---- one mesh over several GPUs (the reference is single-device, src/cuda/mod.rs:52) ------------- */

/* Slab boundaries of the partitioned regens, in level-0 layers along the slab axis (z for ZSLAB, x for XSLAB).  Default:
 * uniform, [32 p / n, 32 (p + 1) / n).  The reference scene occupies 14 of the 32 z layers, so uniform slabs leave half of
 * eight GPUs idle: sdmesh_balance_slabs meshes the scene once, unpartitioned, at `probe_levels` (<= 4), counts the active
 * voxels per layer and installs the contiguous split that minimises the heaviest slab.  Deterministic (every rank that
 * calls it with the same scene gets the same table); bounds_out (optional) receives the nparts + 1 boundaries.
 * sdmesh_set_slab_bounds installs a caller-made table (NULL: back to uniform).  Empty slabs are allowed. */
int sdmesh_set_slab_bounds(sdmesh_handle *h, int mode, int nparts, const uint32_t *bounds);
int sdmesh_balance_slabs(sdmesh_handle *h, int mode, int nparts, int probe_levels, uint32_t *bounds_out);

/* Order-restoring merge of `nranks` per-slab compact triangle lists (each from sdmesh_device_triangles of a
 * SDMESH_PART_ZSLAB regen) into the single-GPU emission order: d_lists = HOST array of nranks DEVICE pointers, d_counts =
 * DEVICE array of nranks x 1024 per-column counts (sdmesh_column_triangle_counts of every rank, rank-major).  A HIP
 * kernel; the merged list (device, cap triangles) is what sdmesh_weld_device_triangles then welds. */
int sdmesh_merge_by_column(sdmesh_handle *h, const struct Triangle *const *d_lists, const uint32_t *d_counts, int nranks,
                           struct Triangle *d_out, size_t cap, size_t *n_out);

/* RCCL communicator of the handle (one process per GPU).  Rank 0 makes the id, the host distributes the 128 bytes
 * (any side channel: the launcher's store, MPI, torch.distributed), every rank calls sdmesh_comm_init.  librccl.so.1 is
 * bound at run time, so hosts that never call these need no RCCL. */
typedef struct sdmesh_unique_id { char bytes[128]; } sdmesh_unique_id;      /* = ncclUniqueId */
int sdmesh_comm_unique_id(sdmesh_unique_id *out);
int sdmesh_comm_init(sdmesh_handle *h, const sdmesh_unique_id *id, int nranks, int rank);
int sdmesh_comm_destroy(sdmesh_handle *h);
/* ncclAllReduce(sum) of 1 over the communicator: how many ranks really took part */
int sdmesh_comm_ranks(sdmesh_handle *h, int *ranks_seen);

/* The caller's own collectives instead of RCCL (MPI, UCX, a host-staged test transport ...): the six operations the one-mesh steps
 * use, with RCCL's meaning.  Every buffer is a DEVICE pointer of this handle's GPU; `stream` is the hipStream_t the step runs on: an
 * operation takes effect after the work already enqueued there and before the work enqueued afterwards (a host-staged transport
 * synchronises the stream, moves the bytes and returns; a stream-aware one enqueues).  all_gather: `bytes` per rank, rank r's block at
 * recv + r * bytes, and send may BE that block (in place).  all_reduce_u32: `count` 32-bit unsigned words, op = SDMESH_REDUCE_*; send may
 * equal recv.  send / recv between group_start and group_end must make progress together (a root receives from every peer in one
 * group; the x-slab weld's neighbours send up and receive from below in one).  Return 0, or non-zero: the step then fails with
 * SDMESH_ERR_RCCL and the value in sdmesh_last_error.  sdmesh_sharded_regen and sdmesh_distributed_regen run unchanged over it, both
 * routes.  The struct is copied; `user` must outlive the communicator (sdmesh_comm_destroy). */
enum { SDMESH_REDUCE_SUM = 0, SDMESH_REDUCE_MAX = 1 };
typedef struct sdmesh_transport {
    void *user;
    int (*all_gather)(void *user, const void *send, void *recv, size_t bytes, void *stream);
    int (*all_reduce_u32)(void *user, const void *send, void *recv, size_t count, int op, void *stream);
    int (*send)(void *user, const void *buf, size_t bytes, int peer, void *stream);
    int (*recv)(void *user, void *buf, size_t bytes, int peer, void *stream);
    int (*group_start)(void *user);
    int (*group_end)(void *user);
} sdmesh_transport;
int sdmesh_comm_init_transport(sdmesh_handle *h, int nranks, int rank, const sdmesh_transport *transport);
/* Every operation of the handle's collective layer once (RCCL's or the caller's), this rank being its own peer where one is needed: grouped
 * send / recv of the element types and odd counts the steps use, an out-of-place all-gather, both reductions; *mismatches = words that did
 * not arrive as sent.  Collective (every rank calls it); a diagnostic for a new node or a new transport, never called by the steps. */
int sdmesh_comm_selftest(sdmesh_handle *h, uint64_t *mismatches);

/* ONE regen of ONE mesh over all ranks of the communicator, entirely behind this call.
 * Indexed form (default): rank r meshes AND welds level-0 z-slab r, ncclAllGather of the 1024 per-column triangle counts, the 1024
 * per-column first-seen vertex counts, sizes and status (8 KB), grouped ncclSend / ncclRecv of the slab's vertex and index buffers (24 B per vertex + 12 B per triangle) straight to
 * `root` (each peer on its own xGMI link); the root merges the buffers by (column, rank) segments WITHOUT welding again (only vertices
 * near a slab face are looked up, in one small hash).
 * Raw form (option "shard_indexed" = 0, also the way out for wide weld keys: the indexed form fails with SDMESH_ERR_WELD_RANGE then):
 * no weld on the ranks, the compact per-slab triangle lists (72 B per triangle) go to the root, order-restoring merge kernel and
 * first-seen weld of every corner there.  Every rank must use the same form (checked in the all-gather).
 * On the root *info describes the welded mesh (identical to a single-GPU sdmesh_regen, fetch it with sdmesh_device_mesh /
 * sdmesh_download_mesh; in the indexed form info->device_ms is the time of the root's merge, and the root's voxel list is not kept);
 * on the other ranks it describes their slab.  Collective: every rank of the communicator must call it with the same levels and root.
 * Failure is collective too: the all-gather carries a status word per rank and is entered even by a rank whose slab
 * failed, so when any rank fails (its regen, an allocation, a count mismatch) every rank returns from this call with an
 * error -- the failing one with its own code, the others with SDMESH_ERR_PEER -- and none is left waiting in a collective.
 * The root's receive buffers grow behind a second one-word agreement, entered only when its published capacity is too
 * small (never in steady state).  Only an SDMESH_ERR_RCCL / SDMESH_ERR_HIP from a collective itself is not agreed upon.
 *
 * Two routes (indexed form).  The first call -- and any call whose levels, root, scene, slab table or sizes no longer fit -- is
 * SYNCHRONOUS: the message sizes are read back from the all-gather (three host synchronisations per step).  It ends by committing a
 * plan: a slot size per rank (the rank's counts + 1/8 + 2048), the same on every rank because all derive it from the same gathered
 * records.  Later calls are PLANNED: the slab's chain, its count kernels, the all-gather, plan-sized messages into plan-sized slots
 * and the merge are enqueued blind and the call synchronises ONCE, at its end; the real counts ride in the gathered records, the
 * kernels read them on the device, every rank verifies them afterwards.  A planned step that did not fit (or in which a rank
 * reported a failure) is repeated synchronously by EVERY rank -- they all see the same records -- inside the same call.  Option
 * "shard_planned" = 0 keeps every step synchronous.  Options that change what a rank does in this call ("shard_indexed",
 * "shard_planned", the scene, the slab table) are part of the collective contract: set them on every rank alike. */
int sdmesh_sharded_regen(sdmesh_handle *h, int levels, int root, sdmesh_regen_info *info);
/* Which route the handle's last indexed sdmesh_sharded_regen / sdmesh_sharded_regen_rehearsal took: 1 planned, 2 synchronous,
 * 3 planned and then repeated synchronously, 0 none yet. */
int sdmesh_sharded_route(const sdmesh_handle *h);

/* The indexed form of sdmesh_sharded_regen (its default; option "shard_indexed" = 0 selects the raw-list form described above) for
 * `nranks` handles of ONE process on ONE device, with device-to-device copies in place of the RCCL messages: handle r meshes and
 * welds level-0 z-slab r; the root handle collects the slabs' vertex / index buffers in slots and merges them (csrc/zslab.hip:
 * no second weld, streaming passes only) into the single-GPU mesh, bit for bit, which becomes the root handle's
 * mesh (sdmesh_device_mesh / sdmesh_download_mesh).  Every kernel of the collective form runs, on both routes: the first call takes
 * the synchronous one and commits the plan, a repeat with the same arguments the planned one (slots with head-room, counts read on
 * the device, plan-sized copies).  Tests and single-GPU measurements of the root's share use it.  All handles need the same slab
 * bounds (sdmesh_set_slab_bounds / sdmesh_balance_slabs). */
int sdmesh_sharded_regen_rehearsal(sdmesh_handle **handles, int nranks, int levels, int root, sdmesh_regen_info *info);

/* The two halves of the indexed z-slab step for a caller that moves the bytes itself (the several-process rehearsal over gloo with all
 * ranks on one GPU, where RCCL refuses duplicate devices): exactly the kernels of sdmesh_sharded_regen's synchronous route.
 * sdmesh_zslab_local: this rank meshes and welds level-0 z-slab `part` of `nparts`; d_row (device, SDMESH_ZSLAB_ROW_WORDS u32) receives its
 * row: 1024 column triangle counts, 1024 column first-seen vertex counts, the record.  The slab's mesh is the handle's mesh
 * (sdmesh_device_mesh).  sdmesh_zslab_merge (root): d_rows = the nranks rows in rank order, d_verts[r] / d_tris[r] = rank r's vertex /
 * index buffer on THIS device; the merged single-GPU mesh becomes the handle's mesh.  Slab bounds must agree (sdmesh_set_slab_bounds). */
#define SDMESH_ZSLAB_ROW_WORDS 2056
int sdmesh_zslab_local(sdmesh_handle *h, int levels, int part, int nparts, uint32_t *d_row, sdmesh_regen_info *info);
int sdmesh_zslab_merge(sdmesh_handle *h, int nranks, int levels, const uint32_t *d_rows, const struct Vertex *const *d_verts,
                       const struct IndexTriangle *const *d_tris, sdmesh_regen_info *info);

/* Dense mode: the same regen through the dense (N+1)^3 lattice -- P1 sample, P2a sign bits, P2b
 * LDS-tiled classify with the hierarchical ancestor criterion (A2'), then the usual mesh stage.  O(N^3) work; it
 * exists as the cross-check that "dense == hierarchical" and to measure the two HBM passes, not as the fast path. */
int sdmesh_regen_dense(sdmesh_handle *h, int levels, sdmesh_regen_info *info);

/* P1: dense (N+1)^3 f32 sample of the scene SDF into a DEVICE buffer, x outer /
 * z inner, sample m at (float)m * (5/N) - 2.5.  *kernel_ms (optional) = HIP-event time. */
int sdmesh_sample_dense(sdmesh_handle *h, uint32_t N, float *d_out, float *kernel_ms);

/* P1 of one x-slab: the lattice planes x_begin <= x < x_end of the same volume into d_out[0 .. (x_end - x_begin) (N+1)^2), i.e. the
 * contiguous range [x_begin (N+1)^2, x_end (N+1)^2) of what sdmesh_sample_dense writes (x is the outermost index) -- bit for bit
 * the same samples, whatever the slab's phase against the 16-byte quads.  The dense pass shards over GPUs with no exchange at all:
 * rank r of R takes the planes [r (N+1) / R, (r+1) (N+1) / R) (bench.py --gpus N, leg `dense_p1_sharded`).  `iters` launches are
 * enqueued on the handle's stream; with avg_ms != NULL the call also synchronises and returns the HIP-event time per launch,
 * with avg_ms == NULL it only enqueues (the caller owns the synchronisation).  An empty slab is a no-op. */
int sdmesh_sample_dense_slab(sdmesh_handle *h, uint32_t N, uint32_t x_begin, uint32_t x_end, float *d_out, int iters, float *avg_ms);

/* Time `iters` back-to-back launches of one internal kernel with HIP events on the handle's stream
 * (bench.py's live roofline measurement).  which: 0 = P1 sample_dense(N) into d_buf; 1 = P2a sign bits of d_buf;
 * 2 = write ceiling (P1's store stream over the same 4 (N+1)^3 bytes with no arithmetic). */
int sdmesh_time_kernel(sdmesh_handle *h, int which, uint32_t N, float *d_buf, int iters, float *avg_ms);

/* Per-stage device times of the last planned regen, when option "profile" = 1 was set before it (HIP events between the
 * stages of the chain on the handle's stream; eager launches, never a graph): comma-separated stage names and one time
 * in ms per stage -- create_field, refine_mask / refine_emit per level, tile_dedupe, tile_prefix, project1|project4,
 * flip_claim1|flip_claim4, weld.  bench.py's live cross-check of the rocprofv3 kernel trace under profiles/. */
int sdmesh_profile_stages(sdmesh_handle *h, char *names, size_t names_cap, float *ms, int cap, int *n_out);
/* Newton rounds (vertex x round; each is 13 SDF evaluations) the projection kernel of that regen performed. */
int sdmesh_profile_rounds(sdmesh_handle *h, uint64_t *vertex_rounds);
/* Diagnostic (option "project_diag" = 1): per wave of the projection kernel four u64 {s_memtime at start, at end, vertex
 * rounds, HW_ID | XCC_ID << 32}; waves without work stay zero.  tools/project_timeline.py turns it into the lifetime histogram. */
int sdmesh_project_diag(sdmesh_handle *h, uint64_t *out, size_t cap_waves, size_t *n_waves);

/* Exhaustive self-check of the strict build's lean exact arithmetic on THIS device (csrc/sdf.hpp: correctly rounded sqrt,
 * 1/sqrt and x/3 from one Markstein step on the hardware's 1-ulp v_rsq_f32 / v_rcp_f32 estimates): every one of the 2^32
 * inputs of each function against hipcc's own correctly rounded expansion, about a second in all.  mismatches[i] for
 * i < min(n, SDMESH_SELFCHECKS): 0 sqrt_rn_normal, 1 sqrt_rn, 2 inv_length_rn, 3 third_rn, 4 the packed-pair sqrt.  All
 * zero = the build's results are the IEEE-754 results on this chip (the fast build returns its approximations' counts). */
#define SDMESH_SELFCHECKS 5
int sdmesh_selfcheck_arithmetic(sdmesh_handle *h, uint64_t *mismatches, int n);

/* ---- ray-march preview ------------------------------------------------ */

/* CudaHandler::render (src/cuda/mod.rs:348-420): sphere-traces the current scene (plus the bounding-cube wire frame of
 * compute_render.cu:8-16) into an RGBA8 image of globals->render_texture_size, kernel compute_render
 * (cuda/modules/compute_render.cu:21-97).  target: host buffer of 4 * width * height bytes, as the reference's
 * target_image.  tan(fov / 2) is evaluated once on the host (the reference evaluates the same value per pixel). */
int sdmesh_render(sdmesh_handle *h, const struct GlobalsBuffer *globals, const struct CameraBuffer *camera, struct Rgba *target);
/* Same, into a DEVICE buffer (a renderer that keeps the image on the GPU); *kernel_ms (optional) = HIP-event time. */
int sdmesh_render_device(sdmesh_handle *h, const struct GlobalsBuffer *globals, const struct CameraBuffer *camera,
                         struct Rgba *d_target, float *kernel_ms);

/* ONE regen of ONE mesh over all ranks of the communicator with a DISTRIBUTED weld (level-0 x-slabs; balance them with
 * sdmesh_balance_slabs(h, SDMESH_PART_XSLAB, nranks, ...) first).  x is the outermost digit of the reference's order
 * (src/cuda/mod.rs:105-122), so rank r's triangles are a contiguous range of the single-GPU list and the global first-seen
 * vertex order is "rank 0's vertices, then rank 1's that rank 0 does not hold, ...".  Every rank meshes AND welds its slab;
 * the exchange is two neighbour messages about the vertices within 2.5 voxels of a slab face (their weld keys up, then
 * their global numbers up) and two 4-byte all-gathers; nothing proportional to the mesh crosses a link.  The mesh stays
 * distributed: afterwards sdmesh_device_mesh / sdmesh_download_mesh give THIS rank's kept vertices and its triangles with
 * GLOBAL indices, *first_vertex = the global index of its first kept vertex.  The ranks' parts concatenated in rank order
 * are the single-GPU mesh, bit for bit.  Collective: every rank calls it with the same levels; failure is collective as in
 * sdmesh_sharded_regen (status words ride in the two all-gathers; a vertex outside the packed weld-key range on one rank
 * fails the call on all).  x-slabs must not be empty (sdmesh_set_slab_bounds rejects such a table). */
int sdmesh_distributed_regen(sdmesh_handle *h, int levels, sdmesh_regen_info *info, uint32_t *first_vertex);
/* Like sdmesh_sharded_regen, the call has two routes: synchronous (the band sizes are read back for the message lengths; commits the plan:
 * a message size per rank, the band + 1/8 + 2048) and planned (chain, band kernels, both all-gathers and plan-sized neighbour messages
 * enqueued blind, one host synchronisation; repeated synchronously by every rank if a band did not fit).  1 planned, 2 synchronous,
 * 3 planned then repeated, 0 none yet. */
int sdmesh_distributed_route(const sdmesh_handle *h);
/* The same step for nranks handles of one process on one device (copies instead of messages): the rehearsal used where
 * only one GPU exists.  infos / first_vertex: arrays of nranks (optional). */
int sdmesh_distributed_regen_rehearsal(sdmesh_handle **handles, int nranks, int levels, sdmesh_regen_info *infos, uint32_t *first_vertex);

#ifdef __cplusplus
}
#endif

#endif /* SDMESH_H */
