// kernels.hpp -- launcher declarations shared by kernels.hip and api.hip
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdmesh/bindings.h"
#include "sdf.hpp"

namespace sdm {

// words of the device status block (all zero == healthy)
enum StatusWord : int {
    ST_OVERFLOW = 0,        // a capacity was exceeded (bit says which); host grows buffers and re-runs
    ST_NOT_LATTICE = 1,     // input field is not a lattice field -> host takes the direct path
    ST_WELD_RANGE = 2,      // a vertex lies outside +-10.48 units (21-bit weld key)
    ST_NEWTON_CAPPED = 3,   // informational: closest_surface_point hit the 10 000-iteration cap
    ST_INTERNAL = 4,        // invariant violated
    ST_TILE_OVERFLOW = 5,   // a 256-voxel tile has more unique edges than the LDS hash holds -> host takes the direct path
    ST_NONFINITE = 6,       // a projected vertex has a non-finite x: the reference drops triangles whose first vertex is
                            // such a vertex (src/cuda/mod.rs:289) -> host takes the direct path, which applies that test
    ST_WORDS = 8
};

// Multi-GPU agreement records (every rank learns every rank's status in the all-gather that moves the sizes anyway):
//   z-slab gather   SHARD_ROW words per rank: 1024 column triangle counts, 1024 column first-seen VERTEX counts (indexed form; zeros in the
//                   raw form), then {status, triangles, two receive capacities of this rank as a root (raw form: received / merged
//                   triangles; indexed form: vertices / triangles), vertices, indexed form?, 0, 0}
static constexpr uint32_t SHARD_ROW = 2048 + 8;
static constexpr uint32_t SHARD_REC = 2048;        // first word of the record inside a row
static constexpr uint32_t DIST_REC = 8;            // {count, status, capacity for the lower neighbour's band, inject, plan signature, 0, 0, 0}

hipError_t launch_refine_mask(hipStream_t st, int grid, int scene, const SceneParams *sp, const Point *vox,
                              const uint32_t *n_ptr, Point cs, uint8_t *mask, uint32_t *tilecnt, int chunk = 32, uint32_t *signs = nullptr);
hipError_t launch_scan_tiles(hipStream_t st, uint32_t *cnt, const uint32_t *n_ptr, uint32_t items_per_tile,
                             uint32_t *total_out, uint32_t cap, uint32_t *status);
hipError_t launch_refine_emit(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point cs,
                              const uint8_t *mask, const uint32_t *tilecnt, Point *out, uint32_t cap, uint32_t *n_out,
                              uint32_t *status, const uint32_t *signs = nullptr, uint8_t *cubeidx = nullptr,
                              uint32_t *mtilecnt = nullptr, bool prescan = false);
hipError_t launch_sum_tiles(hipStream_t st, const uint32_t *tilecnt, const uint32_t *n_ptr, uint32_t *total_out);
hipError_t launch_col_refine(hipStream_t st, int scene, const SceneParams *sp, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1, int J,
                             uint8_t *g_list1, uint8_t *g_mask2, uint8_t *g_mask3, uint4 *colinfo, uint32_t *zero, uint32_t nzero,
                             void *ff, size_t ff_bytes, int segs_pref = 1 /* z segments per column (a power of two that divides z1 - z0 is used) */);
constexpr uint32_t SDM_COL_SEGS_MAX = 8;      // the per-column scratch arrays hold this many units per column
hipError_t launch_col_emit(hipStream_t st, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1, int J, const uint8_t *g_list1,
                           const uint8_t *g_mask2, const uint8_t *g_mask3, const uint4 *colinfo, Point *out, uint32_t cap,
                           uint32_t *level_ctr, uint32_t *status, int segs_pref = 1);
hipError_t launch_refine2_mask(hipStream_t st, int grid, int scene, const SceneParams *sp, const Point *vox, const uint32_t *n_ptr,
                               Point cs1, uint8_t *mask1, uint64_t *mask2, uint32_t *tilecnt, uint32_t *n_mid);
hipError_t launch_refine2_emit(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point cs1, const uint8_t *mask1,
                               const uint64_t *mask2, const uint32_t *tilecnt, Point *out, uint32_t cap, uint32_t *n_out,
                               uint32_t *status);
hipError_t launch_refine_expand_raw(hipStream_t st, int grid, const Point *vox, uint32_t n, Point cs, const uint8_t *mask,
                                    Point *out);
hipError_t launch_classify(hipStream_t st, int grid, int scene, const SceneParams *sp, const Point *vox,
                           const uint32_t *n_ptr, Point vs, uint8_t *cubeidx, uint32_t *tilecnt);
hipError_t launch_tile_dedupe(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point vs, const uint8_t *cubeidx,
                              const uint32_t *tilecnt, uint32_t *trioff, uint32_t *tri_e, Vertex *uverts, uint4 *tdesc,
                              uint32_t ucap, uint32_t tcap, uint32_t *ntri_out, uint32_t *ntiles_out, uint32_t *status, uint64_t *dbg = nullptr,
                              bool prescan = false,
                              uint32_t *colstart = nullptr /* z-slab gather: [c] = 1 + first triangle of level-0 column c where the column holds voxels (pre-zeroed) */,
                              uint32_t *coltile = nullptr /* with colstart: the tile of the column's first voxel */);
hipError_t launch_tile_prefix(hipStream_t st, const uint4 *tdesc, const uint32_t *ntiles_ptr, uint32_t *tpre, uint32_t *nu_out,
                              uint32_t *coarse);
hipError_t launch_project(hipStream_t st, int grid, int scene, const SceneParams *sp, Vertex *uverts, const uint4 *tdesc,
                          const uint32_t *tpre, const uint32_t *coarse, const uint32_t *ntiles_ptr, uint32_t *status, int lanes = 4,
                          uint32_t *prof = nullptr, uint64_t *diag = nullptr, uint64_t *wkeys = nullptr, uint32_t wsize = 0,
                          uint32_t *vslot = nullptr, uint32_t *wrep = nullptr, uint2 *wfm = nullptr, uint32_t wmode = 0 /* bit 1: claim here */,
                          uint32_t *nu_out = nullptr /* tpre == nullptr: workgroups prefix the tile counts themselves; total -> *nu_out */,
                          uint32_t batch_shift = 6 /* k_project1: runs of 2^n consecutive vertices are dealt round robin to the waves */);
constexpr uint32_t SDM_LOCAL_TILES = 2048;     // = LOCAL_TILES of kernels.hip: tile counts a workgroup prefixes for itself in LDS
hipError_t launch_render(hipStream_t st, int scene, const SceneParams *sp, const float *cam13, float fov_fac, float screen_x,
                         float screen_y, float gtex_x, float gtex_y, uint32_t width, uint32_t height, void *rgba);
hipError_t launch_flip_claim(hipStream_t st, int flip_grid, int claim_grid, int scene, const SceneParams *sp, const Vertex *uverts,
                            const uint32_t *tri_e, const uint32_t *ntri_ptr, uint8_t *flip, int lanes, const uint4 *tdesc,
                            const uint32_t *ntiles_ptr, uint64_t *wkeys, uint32_t wsize, uint32_t *vslot, uint32_t *status, uint32_t wmode,
                            uint32_t *wrep, uint2 *wfm);
hipError_t launch_flip(hipStream_t st, int grid, int scene, const SceneParams *sp, const Vertex *uverts, const uint32_t *tri_e,
                       const uint32_t *ntri_ptr, uint8_t *flip, int lanes = 4);
hipError_t launch_raw_desc(hipStream_t st, int grid, const uint32_t *ntri_ptr, uint4 *tdesc, uint8_t *flip, uint32_t *ntiles_out,
                           uint32_t *nu_out, uint32_t *tri_e = nullptr);
hipError_t launch_tile_claim(hipStream_t st, int grid, const Vertex *uverts, const uint4 *tdesc, const uint32_t *ntiles_ptr,
                             uint64_t *wkeys, uint32_t wsize, uint32_t *vslot, uint32_t *status, uint32_t wmode, uint32_t *wrep, uint2 *wfm);
hipError_t launch_merge_by_column(hipStream_t st, const Triangle *const *d_lists, const uint32_t *d_counts, uint32_t cstride, int nranks, uint32_t *d_plan,
                                  Triangle *d_out, uint32_t cap, uint32_t *total_out, uint32_t *status);
hipError_t launch_weld(hipStream_t st, int grid, const Vertex *uverts, const uint4 *tdesc, const uint32_t *ntiles_ptr,
                       uint32_t *tri_e, const uint8_t *flip, uint32_t *first_ord, uint2 *wfm, const uint32_t *wrep,
                       uint32_t *vslot, uint32_t *wv, uint32_t *torder, uint32_t *tfirst, uint32_t *tfpre,
                       Vertex *vertices, uint32_t vcap, uint32_t *nverts_out, uint32_t *indices, uint32_t *status,
                       int flip_scene = -1 /* >= 0: the orientation test runs inside k_tile_first */, const SceneParams *sp = nullptr,
                       int part = 0, const uint32_t *pub_src = nullptr /* control block -> host copy by k_tile_finish */, uint32_t *pub_dst = nullptr,
                       uint32_t pub_words = 0, uint32_t pub_ovf = 0, uint32_t pub_nvert = 0, int tf_threads = 256 /* workgroup size of k_tile_first */,
                       const uint32_t *colstart = nullptr, const uint32_t *coltile = nullptr /* as launch_tile_dedupe left them */, uint32_t *colvloc = nullptr /* [1024] scratch */,
                       uint32_t *colvstart = nullptr /* out: [c] = 1 + first welded vertex first seen in column c (0 where colstart is) */);
hipError_t launch_expand_compact(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *tri_e, const uint8_t *flip,
                                 const uint32_t *ntri_ptr, Triangle *out, uint32_t cap);
hipError_t launch_expand_raw(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *tri_e, const uint8_t *flip,
                             const uint8_t *cubeidx, const uint32_t *trioff, uint32_t n, Triangle *out);
hipError_t launch_triangles_direct(hipStream_t st, int scene, const SceneParams *sp, const Point *vox, uint32_t n, Point vs,
                                   Triangle *out);
// (x0, x1: the x planes [x0, x1) of the volume, written to out[0 ..) -- the whole volume is (0, N + 1))
hipError_t launch_sample_dense_q1(hipStream_t st, int scene, const SceneParams *sp, float *out, uint32_t N, bool nt, uint32_t x0, uint32_t x1,
                                  bool xcd = false /* an XCD's workgroups write one contiguous eighth */);   // one quad per thread (< 2^32 samples in the volume)
hipError_t launch_stream_fill(hipStream_t st, int grid, float *out, uint64_t total, float value, bool xcd = false);
hipError_t launch_sample_dense_rows(hipStream_t st, int grid, int scene, const SceneParams *sp, float *out, uint32_t N, bool nt, uint32_t x0, uint32_t x1,
                                    bool aligned_rows /* sd_obj family: whole-line stores + hoisted z terms (needs a 16-byte aligned out) */,
                                    bool nt_rows /* nontemporal stores in that kernel too (measured slower) */,
                                    int occ = 0 /* aligned kernel: workgroups per CU capped through its LDS allotment (0 = by row length, < 0 = no cap) */,
                                    bool xcd = false /* an XCD's workgroups take one contiguous eighth of the row groups */);
hipError_t launch_sample_dense(hipStream_t st, int grid, int scene, const SceneParams *sp, float *out, uint32_t N, bool nt,
                               bool contig, uint32_t x0, uint32_t x1);
hipError_t launch_create_field(hipStream_t st, Point *vox, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1, uint32_t *n_out,
                               uint32_t *zero = nullptr, uint32_t nzero = 0, void *ff = nullptr, size_t ff_bytes = 0);
hipError_t launch_raw_count(hipStream_t st, int grid, const Triangle *raw, const uint32_t *n_ptr, uint32_t *tilecnt);
hipError_t launch_raw_compact(hipStream_t st, int grid, const Triangle *raw, const uint32_t *n_ptr, const uint32_t *tileoff,
                              Triangle *out, uint32_t *tri_e, uint32_t cap);
hipError_t launch_column_counts(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, const uint8_t *cubeidx,
                                uint32_t *colcnt);
hipError_t launch_tile_tricount(hipStream_t st, int grid, const uint8_t *cubeidx, const uint32_t *n_ptr, uint32_t *tilecnt);
// multi-GPU distributed weld (x-slabs)
// ---- z-slab indexed gather (csrc/zslab.hip) -------------------------------------------------------------------------------
// The ranks' welded slab buffers lie in SLOTS of the root's gather buffers: vertex i of rank r at vbase[r] + i, triangle t at
// tbase[r] + t; vbase[r + 1] - vbase[r] is the slot's CAPACITY (a slot is as large as the message the root receives into it: the
// exact count on the synchronous path, the planned size on the planned one), the actual counts are in the gathered rows.
struct ZsRanks {
    uint32_t nranks;
    uint32_t vbase[33], tbase[33];      // first vertex / triangle of rank r's slot; [nranks] = the slots' total
    uint32_t blk;                       // slot entries per workgroup of k_zs_vertex_out / k_zs_index_out: 256 or 1024
    uint32_t tblk[33];                  // first `blk`-triangle block of rank r's slot in k_zs_index_out's grid; [nranks] = blocks in all
    float zlo[32], zhi[32];             // the slabs' z extents
    float zbot, ztop, band;             // the domain's z extent (a face on it is no slab face), half-width of the face bands
};
// words of a gathered row's record (row = 1024 column triangle counts | 1024 column first-seen vertex counts | record)
enum ZsRec : uint32_t { ZR_STATUS = 0, ZR_TRIS = 1, ZR_CAP_V = 2, ZR_CAP_T = 3, ZR_VERTS = 4, ZR_FORM = 5, ZR_INJECT = 6, ZR_SIG = 7 };
// words of the merge's own control block (device) and of what a step publishes to pinned host memory: [ZC_WORDS | R records of 8 words]
enum ZsCtl : uint32_t { ZC_TRIS = 0, ZC_VERTS = 1, ZC_NBAND = 2, ZC_BANDFLAGS = 3 /* 1: weld range, 64: band hash full */, ZC_PLANFLAGS = 4 /* 1: rows invalid or over a slot's capacity, 2: merged triangles over capacity */,
                     ZC_SEQ = 5, ZC_VFLAGS = 6 /* 4: merged vertices over capacity */, ZC_RANKBAD = 8 /* + r: the vertex counts of rank r's row do not add up */, ZC_WORDS = 40 };
// local phase, every rank: the row the chain published (column starts in triangles and in welded vertices: k_tile_dedupe, k_tile_finish)
// becomes counts, checked, with its record; `next_row`: the row of the next step's buffer, cleared here (the next chain publishes into it)
hipError_t launch_zs_local(hipStream_t st, uint32_t *row, uint32_t *next_row, const uint32_t *ctr_tris, const uint32_t *ctr_verts, const uint32_t *status,
                           uint32_t host_status, uint32_t cap_v, uint32_t cap_t, uint32_t inject, uint32_t sig);
// root: the merge.  counts = the gathered rows (cstride words each); zr = the slot table (host copy for the grid sizes, d_zr the same in device
// memory); the tables tplan / vplan [2 nseg], vstart / newcnt [nseg], per slot vertex segof / redirect / map; hash (hsize slots: 8-byte keys,
// then 4-byte minima) only with more than one rank; zctl = ZC_WORDS words; pub (pinned host, optional) receives zctl and the records
hipError_t launch_zs_merge(hipStream_t st, int grid, const Vertex *v_all, const uint32_t *idx_all, const uint32_t *counts, uint32_t cstride,
                           const ZsRanks &zr, const ZsRanks *d_zr, uint32_t *tplan, uint32_t *vplan, uint32_t *vstart, uint32_t *newcnt,
                           uint32_t *segof, uint32_t *redirect, uint32_t *map, uint32_t *blockdrop /* per zr.blk slot vertices */, uint32_t *thint /* per zr.blk slot triangles */,
                           uint64_t *hkeys, uint32_t hsize,
                           Vertex *v_out, uint32_t cap_verts, uint32_t *i_out, uint32_t cap_tris, uint32_t *zctl, uint32_t seq, uint32_t *pub);
// peers: the records (and a zeroed control block) to pinned host memory
hipError_t launch_zs_publish(hipStream_t st, const uint32_t *counts, uint32_t cstride, uint32_t nranks, const uint32_t *zctl, uint32_t seq, uint32_t *pub);
// distributed weld (x-slabs): every count is read from device memory (`*_ptr`), bounded by the capacity beside it
hipError_t launch_band_select(hipStream_t st, int grid, const Vertex *v, const uint32_t *v_ptr, uint32_t vcap, float up_thr, float lo_thr, bool has_up, bool has_lo,
                              uint32_t *up_ids, uint64_t *up_keys, uint32_t *lo_ids, uint32_t *ctr, uint32_t *dup_of, uint32_t *status);
hipError_t launch_dist_match(hipStream_t st, int grid, const uint64_t *prev_keys, const uint32_t *n_prev_ptr, uint32_t ncap, uint64_t *hkeys, uint32_t *hpos, uint32_t hsize,
                             const Vertex *v, const uint32_t *lo_ids, const uint32_t *n_lo_ptr, uint32_t *dup_of, uint32_t *status);
// what the planned step folds into the kernels that run anyway: this rank's record of the kept-count all-gather (written by the numbering
// kernel's workgroup 0) and the publication of both record tables to pinned host memory (by the finalize kernel's); dst / pub == nullptr: not
struct DistRecord { uint32_t *dst; const uint32_t *count; const uint32_t *status; uint32_t host_status, cap_prev, inject, sig; };
struct DistPublish { const uint32_t *nups; const uint32_t *counts; uint32_t nranks, seq; uint32_t *pub; };
hipError_t launch_dist_number(hipStream_t st, int grid, const uint32_t *dup_of, const uint32_t *v_ptr, uint32_t vcap, uint32_t *tilecnt,
                              uint32_t *n_keep_out, uint32_t *newidx, uint32_t *status, const DistRecord &rec);
hipError_t launch_dist_pack_up(hipStream_t st, int grid, const uint32_t *up_ids, const uint32_t *n_up_ptr, uint32_t ncap, const uint32_t *newidx, const uint32_t *counts,
                               uint32_t me, uint32_t *up_g);
hipError_t launch_dist_finalize(hipStream_t st, int grid, const Vertex *v, const uint32_t *v_ptr, uint32_t vcap, const uint32_t *t_ptr, uint32_t tcap,
                                const uint32_t *dup_of, const uint32_t *newidx, const uint32_t *prev_g, uint32_t pcap, const uint32_t *counts, uint32_t me,
                                uint32_t *indices, Vertex *out_v, const DistPublish &pb);
hipError_t launch_dist_record(hipStream_t st, uint32_t *dst, const uint32_t *count, const uint32_t *status, uint32_t host_status, uint32_t cap_prev,
                              uint32_t inject, uint32_t sig);
hipError_t launch_dist_publish(hipStream_t st, const uint32_t *nups, const uint32_t *counts, uint32_t nranks, uint32_t seq, uint32_t *pub /* 8 + 2 nranks DIST_REC words */);
hipError_t launch_sign_bits(hipStream_t st, int grid, const float *samples, uint64_t total, uint32_t *bits);
hipError_t launch_dense_classify(hipStream_t st, int grid, bool emit, const uint32_t *bits, uint32_t k, uint32_t *cellcnt,
                                 Point *vox, uint8_t *cubeidx, uint32_t cap);
hipError_t launch_set_u32(hipStream_t st, uint32_t *p, uint32_t v);
// exhaustive self-check of the lean exact arithmetic of sdf.hpp (which = 0..4, see kernels.hip); *bad += mismatches
hipError_t launch_selfcheck(hipStream_t st, int which, unsigned long long *bad);
// one 4-word record of a status-carrying all-gather: {*count or v0, v1 | device status flags (0x100 weld range, 0x200 internal, 0x400 overflow), v2, v3}
hipError_t launch_pack_record(hipStream_t st, uint32_t *dst, const uint32_t *count, const uint32_t *status, uint32_t v0, uint32_t v1, uint32_t v2,
                              uint32_t v3);

}  // namespace sdm
