// api.hip -- C ABI of libsdmesh.so (include/sdmesh.h): the host half of the reference's
// CudaHandler (src/cuda/mod.rs:25-346) re-designed around device-resident data.
//
// Memory model: every device buffer is grow-only and owned by the handle, like the reference's
// DynamicCudaSlice (src/cuda/mod.rs:10-23), but results stay in HBM until asked for.  The
// device chain of a regen is a fixed sequence of grid-stride launches that read their item
// counts from a device counter block, so once capacities are known there is exactly one host
// synchronisation per regen (the final read-back of counters + status).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include <rccl/rccl.h>        // types only: the functions are resolved with dlsym when a communicator is first asked for

#include "../../include/sdmesh.h"
#include "kernels.hpp"

using namespace sdm;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum {
    CTR_LEVEL0 = 0,     // .. CTR_LEVEL0 + SDMESH_MAX_LEVELS : voxel count entering level i
    CTR_NUNIQ = 16,
    CTR_NTRI = 17,
    CTR_NVERT = 18,
    CTR_NSLOTS = 19,
    CTR_SCRATCH = 20,
    CTR_NTILES = 21,
    CTR_DIST_NUP = 22,      // distributed weld: vertices in the upper / lower face band, vertices kept, vertex count
    CTR_DIST_NLO = 23,
    CTR_DIST_NKEEP = 24,
    CTR_DIST_V = 25,
    CTR_WORDS = 32,
    CTL_LVL0 = 64          // control block: [counters 32 | status 8 | pad | per-level tile counters ...]
};

}  // namespace

// RCCL entry points (librccl.so.1), bound at run time: a single-GPU host never loads the library, and inside a
// PyTorch-ROCm process the copy torch already loaded (same SONAME) is reused instead of a second one.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct sdmesh_handle {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t side = nullptr;                 // side stream: hash-table memset and weld claim overlap the main chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_ff = nullptr;
    int scene = SDMESH_SCENE_SDOBJ;
    SceneParams sp{};
    SceneParams *d_sp = nullptr;   // device copy of sp (kernels read the scene uniform through this pointer)
    SceneParams *h_sp = nullptr;   // pinned staging for it
    SdfInstr prog_base[SDM_MAX_INSTR] = {};        // the program as set; sp.prog = prog_base + prog_time * prog_motion (sdmesh_set_program_time)
    float prog_motion[SDM_MAX_INSTR][7] = {};
    float prog_time = 0.0f;
    bool sp_dirty = true;
    int grid = 1024;           // workgroups per grid-stride launch of the item-parallel kernels (4 per CU)
    int tgrid = 4096;          // workgroups of the per-tile kernels.  With 4 per CU a 512^3 field (1037 tiles on 1024
                               // workgroups) made every tile kernel take TWO tile latencies; the grid now covers the tile
                               // count up to the residency limit and beyond
    int force_direct = 0;
    int project_mult = 3;      // k_project grid = project_mult * grid: more, shorter waves tighten the tail (measured: -8 %)
    int flip_mult = 2;         // same knob for k_flip (measured: -3 %)
    // lanes per vertex / triangle in the projection / orientation kernels: 4 (one stencil offset per lane), 1 (the
    // whole Newton round in one lane: ~40 % fewer instructions per item, but 4x fewer items in flight, so it only pays
    // when there are enough items to keep every lane refilled) or 0 = by the planned item count (measured crossover
    // between 82 k and 323 k vertices: 256^3 0.158 vs 0.168 ms, 512^3 0.247 vs 0.234 ms, 1024^3 0.669 vs 0.561 ms)
    int project_lanes = 0;
    int flip_lanes = 0;
    int ff_fork = 0;           // weld-table clear: 0 = inside k_create_field, 1 = memset on the side stream during the refine (measured:
                               // the fork + join costs 8-12 us more than the whole clear), 2 = memset node on the main stream
    int mask_lanes = 0;        // k_refine_mask layout: 1 = one lane per parent, 32 = 32 lanes per parent, 0 = 1 for the compiled scenes
    int fork = 2;              // weld claim and orientation test: 2 = one dispatch, 1 = two streams, 0 = one after the other
    uint32_t tp_cap = 0;       // words of h->tpre before the coarse index
    uint32_t lane_hint = 0;    // planned pre-weld vertex count of the chain being enqueued (0 = unknown)
    int mask_chunk = 32;       // parents per workgroup pass in k_refine_mask (8, 16 or 32)
    int emit_ci = 1;           // planned path: the last refine level also writes the cube indices + tile triangle counts (no k_classify)
    int refine2 = 0;           // planned path: two refine levels per step (k_refine2_*): measured no faster, off by default
    int col_refine = 1;        // levels 0 .. min(levels - 1, 3) fused per level-0 column (k_col_refine + k_col_emit: 2 launches
                               // instead of 1 + 2 per level); 0 = level by level from the level-0 field
    DevBuf col_list1, col_mask2, col_mask3, colinfo;
    int skip_weld = 0;         // regen stops after the orientation test (multi-GPU ranks: the root welds)
    int weld_wide = 0;         // option: 1 = always weld with the unbounded [i64; 3] key form
    int fuse_flip = 1;         // with the claims in the projection kernel and the weld following, the orientation test runs inside
                               // k_tile_first (one launch less)
    bool flip_pending = false; // mesh_front left the orientation test to mesh_weld
    int claim_in_project = 1;  // the weld claims run in the projection kernel's epilogue (under its tail) instead of beside the
                               // orientation test; packed-key form and fork = 2 only
    int weld_shift = 0;        // option: coarse-cell shift of locality-preserving home slots (0 = uniform hashing, the default:
                               // locality makes the claim's atomics collide; -1 = shift from the voxel size)
    bool wide_active = false;  // the wide form is in use (forced, or a vertex did not fit the packed 3 x 21-bit key)
    bool welded = false;       // tri_e holds the final corner order
    int dense_grid = 16384;    // workgroups for k_sample_dense
    int dense_aligned = 1;     // option "dense_aligned": sd_obj family rows through k_sample_dense_rows_aligned (0 = the plain row kernel)
    int dense_nt = 1;          // nontemporal stores in k_sample_dense
    int dense_contig = 1;      // flat P1 kernel: every workgroup writes one contiguous run (measured best)
    int dense_q1 = 1;          // flat P1 kernel: one quad per thread, one 4 KiB piece per workgroup (the shape the write stream is fastest in)
    int dense_xcd = 1;         // option "dense_xcd": workgroups of one XCD write one contiguous eighth of the output: 1 = in the aligned row kernel (sd_obj
                               // family: 512^3 0.110 -> 0.104 ms, 1024^3 0.861 -> 0.805), 2 = in every P1 kernel (sphere / write ceiling: no difference), 0 = nowhere
    int dense_occ = 0;         // option "dense_occ": workgroups per CU of the aligned row kernel (0 = the launcher's choice, < 0 = no cap)
    int dense_rows = 2;        // P1 kernel: 0 = flat 16-B aligned quads, 1 = one wave per z-row, 2 = auto
    char err[512] = {0};
    uint64_t buf_gen = 0;      // bumped whenever a device buffer is (re)allocated: invalidates a captured graph
    int use_graph = 0;         // replay the planned regen as a hipGraph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_key[8] = {0};
    int graph_cur = 0;
    Point graph_vs{0, 0, 0};
    // regen_begin / regen_end: a planned chain that has been enqueued but not yet waited for
    bool pend_active = false, pend_async = false;
    int pend_levels = 0, pend_mode = 0, pend_part = 0, pend_nparts = 1;
    Point pend_vs{0, 0, 0};
    uint32_t pend_ncap = 0;
    bool pend_timed = false;

    // voxel lists (ping-pong), refine scratch
    DevBuf vox[2];
    int cur = 0;
    DevBuf render_tex;                             // RGBA8 target of the ray-march preview
    DevBuf mask, mask2x, signs, tilecnt, ctl;             // ctl: counters + status + pre-zeroed per-level tile counters
    uint32_t lvl_off[SDMESH_MAX_LEVELS + 2] = {0};
    // counters / status
    uint32_t *d_ctr = nullptr;     // CTR_WORDS
    uint32_t *d_status = nullptr;  // ST_WORDS
    uint32_t *h_pinned = nullptr;  // CTR_WORDS + ST_WORDS, pinned
    bool ctl_published = false;    // the chain just enqueued writes h_pinned itself (k_tile_finish): no read-back copy
    int publish_ctl = 1;           // option "publish_ctl": 0 = always read back with a copy
    int local_prefix = 1;          // option "local_prefix": 0 = always launch k_tile_prefix / k_scan_u32
    int col_segs = 1;              // option "col_segs": z segments per level-0 column in the column kernels (1 / 2 / 4 / 8); measured: more,
                                   // smaller units are slower (the kernels are bound by the fixed phase chain of a workgroup, not by the heaviest column)
    int tf_threads = 512;          // option "tf_threads": workgroup size of k_tile_first when it holds the orientation test (128 .. 1024)
    int project_prio = 0;          // option "project_prio" = 1: k_project1's waves lower their issue priority as they advance (s_setprio):
                                   // the waves of a SIMD then finish together instead of oldest first, at the same total time (measured)
    int project_grid = 0;          // option "project_grid": workgroups of the projection (0 = grid * project_mult, capped by the plan)
    int project_batch = 6;         // option "project_batch": log2 of k_project1's dealing unit (64 = one contiguous run per wave; shorter
                                   // runs even the waves out but every wave then pays the straggler rounds: measured slower, 48.7 -> 57 us)
    int inject_fail = 0;           // option "inject_fail" (tests of the multi-GPU agreement): 1 = this rank's local phase of the next one-mesh
                                   // call fails, 2 = its buffer growth after the first agreement fails; cleared by the call
    int download_streams = 1;      // option "download_streams" = 2: sdmesh_download_mesh copies the vertex and the index buffer on two streams (measured slower:
                                   // 12.7 MB to touched pageable memory 0.287 ms on one stream = 44 GB/s, the link's rate; 0.31-0.35 ms on two)
    int timing = 0;                // option "timing": 1 = an event pair around the planned chain fills info->device_ms; off by default:
                                   // the two marker packets cost ~9 us of wall time per regen (0.155 -> 0.146 ms at 512^3)
    // mesh stage
    DevBuf cubeidx, trioff, ff_arena, first_ord_buf, uverts, tri_e, vslot, tdesc, tfirst, tfpre, tpre, flip, wvidx, torder, wrep_buf, wfm_buf;
    DevBuf vertices, indices, raw, compact, staging, dense_samples, dense_bits, cellcnt;
    // capacities the device chain was laid out for
    uint32_t cap_vox = 0, ucap = 0, tcap = 0, wsize = 0;
    uint64_t *wkeys = nullptr;
    uint32_t *first_ord = nullptr;
    uint2 *wfm = nullptr;          // per representative vertex {its first-seen ordinal, minimum ordinal of the key's other vertices}

    // last results (host view)
    uint32_t n_vox = 0, n_tris = 0, n_verts = 0, n_unique = 0;
    Point voxel_size{0, 0, 0};
    bool mesh_valid = false, vox_valid = false, last_direct = false;

    // option "profile": HIP events between the stages of the planned chain (live per-stage device times for bench.py)
    int profile = 0;
    DevBuf prof_rounds;            // 256 counters x 64 B: Newton rounds of the projection kernel (profile mode)
    DevBuf proj_diag;              // option "project_diag": per-wave {start, end, rounds, hw id} of the projection kernel
    int project_diag = 0;
    uint32_t diag_waves = 0;
    uint64_t last_rounds = 0;
    std::vector<hipEvent_t> prof_ev;
    std::vector<std::string> prof_names;
    std::vector<float> prof_ms;
    size_t prof_n = 0;

    // multi-GPU: communicator + the buffers of the z-slab gather / merge
    ncclComm_t comm = nullptr;         // RCCL (sdmesh_comm_init) ...
    sdmesh_transport ext{};            // ... or the caller's collectives (sdmesh_comm_init_transport)
    bool comm_ext = false;
    bool has_comm() const { return comm != nullptr || comm_ext; }
    int comm_rank = 0, comm_nranks = 1;
    DevBuf shard_mine, shard_counts, shard_recv, shard_lists, shard_plan, shard_merged;
    // z-slab INDEXED gather: the ranks' welded buffers in SLOTS of zs_verts / zs_idx (root), the merge's tables, its outputs
    DevBuf zs_verts, zs_idx, zs_tables, zs_pervert, zs_hkeys, zs_vout, zs_iout, zs_ranks, zs_ctl, shard_counts2, zs_colscratch, zs_hints;
    ZsRanks *h_zs = nullptr;       // pinned staging copy of the merge's slot table
    ZsRanks zs_uploaded{};         // what zs_ranks holds (uploaded again only when the table changes: never in steady state)
    bool zs_uploaded_valid = false;
    uint32_t *h_zs_pub = nullptr;  // pinned: what a step publishes (ZC_WORDS control words, then the ranks' records)
    int shard_indexed = 1;         // option "shard_indexed": sdmesh_sharded_regen gathers welded per-slab vertex / index buffers (0: raw triangle lists + root weld)
    int shard_planned = 1;         // option "shard_planned": 0 = every indexed step takes the synchronous route (three host synchronisations)
    int zs_block = 0;              // option "zs_block": slot entries per workgroup of k_zs_vertex_out / k_zs_index_out (256, 1024; 0 = by the mesh's size)
    int shard_slack = 1;           // option "shard_slack": 0 = the plan's slots have no head-room (any growth makes the next planned step repeat: tests)
    uint32_t zs_band_plan = 0;     // face-band vertices of the last indexed merge (sizes the band hash of the next one)
    // The PLANNED indexed step (steady state: one host synchronisation).  The plan is COLLECTIVE state: every rank derives it from the
    // same gathered records at the end of a synchronous step and invalidates it on the same evidence, so all ranks always agree on
    // which route a call takes (message sizes are the plan's, not the step's).
    struct ZsPlan {
        bool valid = false;
        int levels = -1, root = -1, R = 0, scene = -1;
        uint32_t slab_sig = 0, sig = 0;
        uint32_t cap_v[32] = {0}, cap_t[32] = {0};      // slot = message size of rank r, in vertices / triangles
    } zsp;
    struct DistPlan {              // the x-slab distributed weld's plan, collective like ZsPlan: the message size of every rank's upper band
        bool valid = false;
        int levels = -1, R = 0, scene = -1;
        uint32_t slab_sig = 0, sig = 0;
        uint32_t cap_up[32] = {0};
    } dsp;
    uint32_t dist_last_route = 0;  // like zs_last_route
    uint32_t *h_dist_pub = nullptr;    // pinned: [seq .. | the two record tables] of a planned distributed step
    int zs_cur = 0;                // which of the two count buffers (shard_counts / shard_counts2) the next step uses
    uint32_t zs_seq = 0;           // step number, published with the step's results (a stale host copy is told apart)
    uint32_t zs_last_route = 0;    // 1 = the last indexed step ran planned, 2 = synchronous, 3 = planned, then repeated synchronously
    // hooks of a sharded local phase into the ordinary chain
    uint32_t *colcnt = nullptr;    // != nullptr: the chain publishes where every level-0 column starts into this (pre-zeroed) row: [c] in triangles
                                   // (k_tile_dedupe), [1024 + c] in welded vertices (k_tile_rank, k_tile_finish; scratch: zs_colscratch)
    Vertex *out_v = nullptr;       // != nullptr: the weld writes the mesh here (the root's slot of the gather buffers) instead of h->vertices / h->indices
    uint32_t *out_i = nullptr;
    uint32_t out_vcap = 0;
    // where the last mesh lies when it is not in h->vertices / h->indices (a merged or compacted mesh); nullptr = there
    Vertex *res_v = nullptr;
    uint32_t *res_i = nullptr;
    uint32_t *h_counts = nullptr;      // pinned: nranks x SHARD_ROW words (column counts + status record of every rank)
    size_t h_counts_words = 0;
    void **h_lists = nullptr;          // pinned: nranks device pointers

    // distributed weld (x-slabs): band lists, neighbour messages, match table, numbering
    DevBuf dist_upids, dist_upkeys, dist_loids, dist_prevkeys, dist_hkeys, dist_hpos, dist_dupof, dist_newidx, dist_tilecnt, dist_counts,
        dist_upg, dist_prevg, dist_verts, dist_nups;
    uint32_t dist_V = 0, dist_T = 0, dist_nup = 0, dist_nprev = 0;

    // slab boundaries of the partitioned regens (level-0 layers); slab_nparts == 0: uniform
    int slab_mode = 0, slab_nparts = 0;
    uint32_t slab_bounds[33] = {0};

    // capacity plan learnt from the last synchronous run (async path)
    bool plan_valid = false;
    int plan_scene = -1, plan_levels = -1, plan_mode = 0, plan_part = 0, plan_nparts = 1;
    uint32_t plan_counts[SDMESH_MAX_LEVELS + 1] = {0};
    uint32_t plan_uniq = 0;
    uint32_t plan_tris = 0;
    uint32_t tri_hint = 0;     // planned triangle count of the chain being enqueued (0 = unknown)
};

namespace {

int fail(sdmesh_handle *h, int code, const char *fmt, ...) {
    if (h) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(h->err, sizeof h->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(h, SDMESH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define SDCHK(expr)                  \
    do {                             \
        int r_ = (expr);             \
        if (r_ != SDMESH_OK) return r_; \
    } while (0)

int ensure(sdmesh_handle *h, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.p && b.cap >= bytes) return SDMESH_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIPCHK(hipMalloc(&b.p, bytes));
    b.cap = bytes;
    h->buf_gen++;
    return SDMESH_OK;
}

void release(DevBuf &b) {
    if (b.p) (void) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// Stage markers of option "profile": an event BEHIND each stage of the chain; stage time = gap to the previous marker.
int prof_mark(sdmesh_handle *h, const char *name) {
    if (!h->profile) return SDMESH_OK;
    if (h->prof_n >= h->prof_ev.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->prof_ev.push_back(e);
        h->prof_names.emplace_back();
    }
    h->prof_names[h->prof_n] = name;
    HIPCHK(hipEventRecord(h->prof_ev[h->prof_n], h->stream));
    h->prof_n++;
    return SDMESH_OK;
}

int prof_collect(sdmesh_handle *h) {      // after the stream has been synchronised
    h->prof_ms.assign(h->prof_n, 0.0f);
    for (size_t i = 1; i < h->prof_n; i++) HIPCHK(hipEventElapsedTime(&h->prof_ms[i], h->prof_ev[i - 1], h->prof_ev[i]));
    return SDMESH_OK;
}

inline uint32_t tiles(uint32_t n) { return (n + 255u) / 256u; }

// above this many 256-item tiles the ordered kernels get a single-workgroup scan of the tile counts in front instead of
// re-summing the counts before their tile in every workgroup (quadratic in the tile count; one extra ~5 us launch)
constexpr uint32_t PRESCAN_TILES = 2048;
inline size_t align256(size_t v) { return (v + 255) & ~(size_t) 255; }

Point half_size(Point p) { return Point{p.x / 2.0f, p.y / 2.0f, p.z / 2.0f}; }   // src/cuda/mod.rs:126-130

// Lay out the mesh-stage buffers for up to n_cap voxels / u_cap unique (pre-weld) vertices.
int layout_mesh(sdmesh_handle *h, uint32_t n_cap, uint32_t u_cap_hint) {
    if (n_cap < 64) n_cap = 64;
    const uint32_t tcap = 5u * n_cap;
    const uint32_t ucap = 3u * tcap;                  // vertex slots live in corner-ordinal space (sparse)
    const uint32_t uexp = std::max<uint32_t>(u_cap_hint, 2u * n_cap + 4096u);   // expected pre-weld vertices
    const uint32_t wsize = 2u * uexp + 64u;           // load factor <= 0.5; a full table raises ST_OVERFLOW
    SDCHK(ensure(h, h->cubeidx, n_cap));
    SDCHK(ensure(h, h->trioff, 4ull * n_cap));
    SDCHK(ensure(h, h->tilecnt, 4ull * std::max(tiles(n_cap), tiles(5u * n_cap)) + 64));
    SDCHK(ensure(h, h->uverts, sizeof(Vertex) * (size_t) ucap));
    SDCHK(ensure(h, h->first_ord_buf, 4ull * ucap));
    SDCHK(ensure(h, h->tri_e, 12ull * tcap));
    SDCHK(ensure(h, h->vslot, 4ull * ucap));
    SDCHK(ensure(h, h->tdesc, 16ull * (tiles(n_cap) + tiles(tcap) + 16)));
    SDCHK(ensure(h, h->tfirst, 4ull * (tiles(n_cap) + tiles(tcap) + 16)));
    h->tp_cap = tiles(n_cap) + tiles(tcap) + 16;          // tile prefix entries, then the coarse item -> tile index
    SDCHK(ensure(h, h->tpre, 4ull * ((size_t) h->tp_cap + (ucap >> 6) + 8)));
    SDCHK(ensure(h, h->tfpre, 4ull * (tiles(n_cap) + tiles(tcap) + 16)));
    SDCHK(ensure(h, h->torder, 4ull * ucap));
    SDCHK(ensure(h, h->flip, (size_t) tcap + 64));
    SDCHK(ensure(h, h->wvidx, 4ull * ucap));               // welded index per representative vertex (vertex-slot space)
    SDCHK(ensure(h, h->wrep_buf, 4ull * wsize));           // representative vertex per table slot (written by the claim's winner)
    SDCHK(ensure(h, h->wfm_buf, 8ull * ucap));             // records per representative vertex, initialised by the claim's winner
    SDCHK(ensure(h, h->vertices, sizeof(Vertex) * (size_t) uexp));
    SDCHK(ensure(h, h->indices, 12ull * tcap));
    // what must start as 0xFF per regen: the weld keys only (the per-key records are initialised by the claim's winners)
    SDCHK(ensure(h, h->ff_arena, align256(8ull * wsize)));
    h->wkeys = (uint64_t *) h->ff_arena.p;
    h->wfm = (uint2 *) h->wfm_buf.p;
    h->first_ord = (uint32_t *) h->first_ord_buf.p;
    h->wsize = wsize; h->ucap = ucap; h->tcap = tcap; h->cap_vox = n_cap;
    return SDMESH_OK;
}

size_t ff_bytes(const sdmesh_handle *h) {
    return 8ull * h->wsize;
}

// wide flag | coarse-cell shift << 8 for the weld claim: the coarse cell of the locality-preserving home slot spans about
// `weld_cell` voxels (default 8) in units of the 1e-5 weld quantum (option "weld_shift" overrides; 0 = uniform hashing)
uint32_t weld_mode(const sdmesh_handle *h, Point vs) {
    uint32_t shift;
    if (h->weld_shift >= 0) {
        shift = (uint32_t) h->weld_shift;
    } else {
        const float span = 8.0f * std::max(vs.x, std::max(vs.y, vs.z)) * 100000.0f;
        shift = 13;
        if (span > 1.0f && span < 1.0e6f) { shift = 0; while ((1u << (shift + 1)) <= (uint32_t) span) shift++; }
    }
    if (shift > 20) shift = 20;
    return (h->wide_active ? 1u : 0u) | (shift << 8);
}

// classify -> tile dedupe -> project -> flip for the voxel list d_vox (count at *n_ptr, at most cap_vox):
// leaves (uverts, tri_e, flip, trioff, tdesc) and the counters NUNIQ / NTRI / NTILES on the device
int mesh_front(sdmesh_handle *h, const Point *d_vox, const uint32_t *n_ptr, Point vs, bool zero_counters, bool ff_done = false,
               bool have_cubeidx = false, uint32_t *have_tilecnt = nullptr, bool weld_follows = false) {
    hipStream_t st = h->stream;
    // have_tilecnt: the last refine level already wrote cubeidx and the per-tile triangle counts (k_refine_emit)
    uint32_t *tilecnt = have_tilecnt ? have_tilecnt : (uint32_t *) h->tilecnt.p;
    if (!ff_done) HIPCHK(hipMemsetAsync(h->ff_arena.p, 0xFF, ff_bytes(h), st));
    else if (h->ff_fork == 1) HIPCHK(hipStreamWaitEvent(st, h->ev_ff, 0));   // table was cleared on the side stream during the refine
    if (zero_counters) HIPCHK(hipMemsetAsync(h->d_ctr + CTR_NUNIQ, 0, 4 * sizeof(uint32_t), st));
    if (zero_counters && h->colcnt) HIPCHK(hipMemsetAsync(h->colcnt, 0, 4096, st));      // (a mesh stage run again: its column counts start over)
    if (have_tilecnt) { /* nothing to launch */ }
    else if (have_cubeidx) HIPCHK(launch_tile_tricount(st, h->tgrid, (const uint8_t *) h->cubeidx.p, n_ptr, tilecnt));
    else HIPCHK(launch_classify(st, h->tgrid, h->scene, h->d_sp, d_vox, n_ptr, vs, (uint8_t *) h->cubeidx.p, tilecnt));
    uint64_t *d_tile_dbg = nullptr;
    if (h->project_diag == 2) {             // diagnostic: phase stamps of k_tile_dedupe instead of the projection timeline
        h->diag_waves = 2u * (tiles(h->cap_vox) + 16u);          // 8 stamps per tile = 2 "wave" records of 4 u64
        SDCHK(ensure(h, h->proj_diag, 32ull * h->diag_waves));
        HIPCHK(hipMemsetAsync(h->proj_diag.p, 0, 32ull * h->diag_waves, st));
        d_tile_dbg = (uint64_t *) h->proj_diag.p;
    }
    HIPCHK(launch_tile_dedupe(st, h->tgrid, d_vox, n_ptr, vs, (const uint8_t *) h->cubeidx.p, (const uint32_t *) tilecnt,
                              (uint32_t *) h->trioff.p, (uint32_t *) h->tri_e.p, (Vertex *) h->uverts.p, (uint4 *) h->tdesc.p,
                              h->ucap, h->tcap, h->d_ctr + CTR_NTRI, h->d_ctr + CTR_NTILES, h->d_status, d_tile_dbg,
                              tiles(h->cap_vox) > PRESCAN_TILES, h->colcnt, h->colcnt ? (uint32_t *) h->zs_colscratch.p : nullptr));
    SDCHK(prof_mark(h, "tile_dedupe"));
    // few tiles: the projection's workgroups prefix the tile counts themselves (no k_tile_prefix launch)
    const bool local_pre = h->local_prefix && tiles(h->cap_vox) <= SDM_LOCAL_TILES;
    uint32_t *coarse = (uint32_t *) h->tpre.p + h->tp_cap;
    if (!local_pre) {
        HIPCHK(launch_tile_prefix(st, (const uint4 *) h->tdesc.p, h->d_ctr + CTR_NTILES, (uint32_t *) h->tpre.p, h->d_ctr + CTR_NUNIQ, coarse));
        SDCHK(prof_mark(h, "tile_prefix"));
    }
    const int plan_lanes = h->lane_hint >= 160000u ? 1 : 4;
    const int p_lanes = h->project_lanes ? h->project_lanes : plan_lanes;
    const int f_lanes = h->flip_lanes ? h->flip_lanes : plan_lanes;
    // the table must have been cleared before the projection starts if its epilogue claims (always true on the planned
    // chain and whenever the memset above ran on this stream)
    // ... and only while a projection wave owns about one batch of vertices: with many batches per wave every claim would
    // wait for the wave's last batch and pile up at the end of the kernel (2048^3: projection 340 -> 498 us)
    // Larger meshes (1-lane kernel only): asynchronous claims, issued per finished vertex and looked at a round later.
    // projection grid: no more workgroups than the planned vertex count fills with one batch per wave (+3 %): workgroups
    // without work still pay the tile-prefix step before they can tell, and start behind the working ones
    int p_grid = h->project_grid > 0 ? h->project_grid : h->grid * h->project_mult;
    if (h->lane_hint) {
        const uint32_t per_wg = 4u * (64u / (uint32_t) p_lanes);
        p_grid = (int) std::min<uint32_t>((uint32_t) p_grid, (h->lane_hint + h->lane_hint / 32u) / per_wg + 2u);
    }
    const uint32_t proj_waves = (uint32_t) p_grid * 4u;
    const bool claim_ok = h->claim_in_project && h->fork == 2 && !h->wide_active && h->lane_hint != 0;
    // (round 4: asynchronous claims from the first batch a wave does not own alone -- 1024^3, 1.27 M vertices on 12 288 waves: 0.2961 -> 0.2909 ms;
    //  at 512^3 every wave owns one batch and the epilogue form is 6 us faster: 0.1202 vs 0.1274)
    const bool few_batches = h->lane_hint <= (64u / (uint32_t) p_lanes) * proj_waves;
    const bool async_claim = claim_ok && (!few_batches || h->claim_in_project == 3) && p_lanes == 1 && h->claim_in_project != 2;   // (3: asynchronous at every size, for A/B)
    const bool early_claim = claim_ok && (few_batches || async_claim);
    uint32_t *d_prof = nullptr;
    uint64_t *d_diag = nullptr;
    if (h->profile) {                       // (profile / diag runs are eager and may allocate: never on the captured path)
        SDCHK(ensure(h, h->prof_rounds, 256 * 64));
        HIPCHK(hipMemsetAsync(h->prof_rounds.p, 0, 256 * 64, st));
        d_prof = (uint32_t *) h->prof_rounds.p;
    }
    if (h->project_diag == 1) {
        h->diag_waves = proj_waves;
        SDCHK(ensure(h, h->proj_diag, 32ull * h->diag_waves));
        HIPCHK(hipMemsetAsync(h->proj_diag.p, 0, 32ull * h->diag_waves, st));
        d_diag = (uint64_t *) h->proj_diag.p;
    }
    HIPCHK(launch_project(st, p_grid, h->scene, h->d_sp, (Vertex *) h->uverts.p, (const uint4 *) h->tdesc.p,
                          local_pre ? nullptr : (const uint32_t *) h->tpre.p, coarse, h->d_ctr + CTR_NTILES, h->d_status, p_lanes, d_prof, d_diag,
                          h->wkeys, h->wsize, (uint32_t *) h->vslot.p, (uint32_t *) h->wrep_buf.p, h->wfm, weld_mode(h, vs) | (early_claim && !async_claim ? 2u : 0u) | (async_claim ? 4u : 0u) | (h->project_prio ? 8u : 0u),
                          h->d_ctr + CTR_NUNIQ, (uint32_t) h->project_batch));
    SDCHK(prof_mark(h, p_lanes == 1 ? "project1" : (p_lanes == 2 ? "project2" : "project4")));
    // fork: the weld claim (scattered CAS, latency-bound) runs beside the orientation test (VALU-bound)
    int f_grid = h->grid * (f_lanes == 1 ? 1 : h->flip_mult);
    if (h->tri_hint) {
        // Every lane (1-lane layout) or 4-lane group takes whole passes over the triangle list: with T = 530 536 on 1 024
        // workgroups 98 of the 4 096 waves ran a third pass and the kernel took three pass times instead of two.  Size the
        // grid so that the planned count divides into equal passes (the grid-stride loop still covers any actual count).
        const uint32_t per_wg = f_lanes == 1 ? 256u : 64u;
        const uint32_t passes = std::max<uint32_t>(1u, (h->tri_hint + per_wg * (uint32_t) f_grid / 2u) / (per_wg * (uint32_t) f_grid));
        f_grid = (int) std::min<uint32_t>(65535u, (h->tri_hint + passes * per_wg - 1u) / (passes * per_wg) + 1u);
    }
    h->flip_pending = false;
    if (early_claim && h->fuse_flip && weld_follows && (f_lanes == 1 || h->tri_hint >= 80000u || h->fuse_flip == 2)) {   // (measured: pays from ~100 k triangles)
        h->flip_pending = true;             // no dispatch here: k_tile_first tests the orientation of its tile's triangles itself
        return SDMESH_OK;
    }
    if (h->fork == 2) {
        // one dispatch: claim workgroups first (latency-bound, resident early), orientation-test workgroups behind them
        HIPCHK(launch_flip_claim(st, f_grid, early_claim ? 0 : h->tgrid, h->scene, h->d_sp, (const Vertex *) h->uverts.p, (const uint32_t *) h->tri_e.p,
                                 h->d_ctr + CTR_NTRI, (uint8_t *) h->flip.p, f_lanes, (const uint4 *) h->tdesc.p, h->d_ctr + CTR_NTILES,
                                 h->wkeys, h->wsize, (uint32_t *) h->vslot.p, h->d_status, weld_mode(h, vs), (uint32_t *) h->wrep_buf.p, h->wfm));
        SDCHK(prof_mark(h, f_lanes == 1 ? "flip_claim1" : "flip_claim4"));
        return SDMESH_OK;
    }
    hipStream_t cst = h->fork ? h->side : st;
    if (h->fork) {
        HIPCHK(hipEventRecord(h->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    }
    HIPCHK(launch_tile_claim(cst, h->tgrid, (const Vertex *) h->uverts.p, (const uint4 *) h->tdesc.p, h->d_ctr + CTR_NTILES,
                             h->wkeys, h->wsize, (uint32_t *) h->vslot.p, h->d_status, weld_mode(h, vs), (uint32_t *) h->wrep_buf.p, h->wfm));
    if (h->fork) HIPCHK(hipEventRecord(h->ev_join, h->side));
    HIPCHK(launch_flip(st, f_grid, h->scene, h->d_sp, (const Vertex *) h->uverts.p, (const uint32_t *) h->tri_e.p,
                       h->d_ctr + CTR_NTRI, (uint8_t *) h->flip.p, f_lanes));
    if (h->fork) HIPCHK(hipStreamWaitEvent(st, h->ev_join, 0));
    return SDMESH_OK;
}

// per-tile weld of (uverts, tri_e, flip) described by tdesc[0..*NTILES) -> vertices / indices
int mesh_weld(sdmesh_handle *h, const Vertex *uverts, uint32_t max_tiles, bool publish = false) {
    hipStream_t st = h->stream;
    const bool local_pre = h->local_prefix && max_tiles <= SDM_LOCAL_TILES;       // k_tile_finish prefixes the tile counts itself
    h->ctl_published = publish;             // the last kernel writes the control block to the pinned host copy itself
    h->res_v = nullptr; h->res_i = nullptr; // (whatever mesh a merge left elsewhere is superseded)
    const int flip_scene = h->flip_pending ? h->scene : -1;
    h->flip_pending = false;
    for (int part = h->profile ? 1 : 0; part <= (h->profile ? 2 : 0); part++) {        // profiling: a stage marker behind k_tile_first
        HIPCHK(launch_weld(st, h->tgrid, uverts, (const uint4 *) h->tdesc.p, h->d_ctr + CTR_NTILES, (uint32_t *) h->tri_e.p,
                           (const uint8_t *) h->flip.p, h->first_ord, h->wfm, (const uint32_t *) h->wrep_buf.p, (uint32_t *) h->vslot.p,
                           (uint32_t *) h->wvidx.p, (uint32_t *) h->torder.p, (uint32_t *) h->tfirst.p, local_pre ? nullptr : (uint32_t *) h->tfpre.p,
                           h->out_v ? h->out_v : (Vertex *) h->vertices.p, h->out_v ? h->out_vcap : (uint32_t) (h->vertices.cap / sizeof(Vertex)), h->d_ctr + CTR_NVERT,
                           h->out_v ? h->out_i : (uint32_t *) h->indices.p, h->d_status, flip_scene, h->d_sp, part,
                           publish ? h->d_ctr : nullptr, publish ? h->h_pinned : nullptr, CTR_WORDS + ST_WORDS, CTR_WORDS + ST_OVERFLOW, CTR_NVERT,
                           flip_scene >= 0 ? h->tf_threads : 256, h->colcnt, h->colcnt ? (const uint32_t *) h->zs_colscratch.p : nullptr,
                           h->colcnt ? (uint32_t *) h->zs_colscratch.p + 1024 : nullptr, h->colcnt ? h->colcnt + 1024 : nullptr));
        if (part == 1) SDCHK(prof_mark(h, flip_scene >= 0 ? "tile_first_flip" : "tile_first"));
    }
    SDCHK(prof_mark(h, h->profile ? "weld_rank_scan_finish" : "weld"));
    return SDMESH_OK;
}

// read counters + status back (the one synchronisation point of a device chain)
int read_back_enqueue(sdmesh_handle *h) {
    HIPCHK(hipMemcpyAsync(h->h_pinned, h->d_ctr, (CTR_WORDS + ST_WORDS) * 4, hipMemcpyDeviceToHost, h->stream));
    return SDMESH_OK;
}

int read_back(sdmesh_handle *h) {
    SDCHK(read_back_enqueue(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SDMESH_OK;
}

// The planned chain's last kernel writes the host copy of the control block itself (no read-back copy).  Should that kernel
// ever leave without publishing, the host must not mistake the previous regen's numbers for this one's: the last status
// word (never set on the device, so it arrives as 0 by either route) is poisoned before the chain is enqueued, and a chain
// that comes back with the poison still there is read back the ordinary way.
constexpr uint32_t PIN_SENTINEL = CTR_WORDS + ST_WORDS - 1;
void poison_pinned(sdmesh_handle *h) { h->h_pinned[PIN_SENTINEL] = 0xFFFFFFFFu; }
int unpoison_pinned(sdmesh_handle *h) {
    if (h->h_pinned[PIN_SENTINEL] == 0xFFFFFFFFu) SDCHK(read_back(h));
    return SDMESH_OK;
}

int status_to_error(sdmesh_handle *h) {
    const uint32_t *s = h->h_pinned + CTR_WORDS;
    if (s[ST_INTERNAL]) return fail(h, SDMESH_ERR_INTERNAL, "device invariant violated (status %u)", s[ST_INTERNAL]);
    if (s[ST_WELD_RANGE]) return fail(h, SDMESH_ERR_WELD_RANGE, "vertex outside the packed weld key range although the wide key form was requested");
    return SDMESH_OK;
}

// push the scene uniform to the device if it changed since the last upload
int sync_scene(sdmesh_handle *h) {
    if (!h->sp_dirty) return SDMESH_OK;
    *h->h_sp = h->sp;
    HIPCHK(hipMemcpyAsync(h->d_sp, h->h_sp, sizeof(SceneParams), hipMemcpyHostToDevice, h->stream));
    h->sp_dirty = false;
    return SDMESH_OK;
}

int zero_status(sdmesh_handle *h) {
    HIPCHK(hipMemsetAsync(h->d_status, 0, ST_WORDS * 4, h->stream));
    return SDMESH_OK;
}

// Control block = [counters | status | per-level tile counters]; grown on demand, counters preserved.
int ensure_ctl(sdmesh_handle *h, size_t lvl_words) {
    const size_t need = 4ull * (CTL_LVL0 + lvl_words + 64);
    if (h->ctl.cap >= need) return SDMESH_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, need * 2));
    HIPCHK(hipMemset(p, 0, need * 2));
    HIPCHK(hipMemcpy(p, h->ctl.p, 4ull * CTL_LVL0, hipMemcpyDeviceToDevice));
    HIPCHK(hipFree(h->ctl.p));
    h->ctl.p = p;
    h->ctl.cap = need * 2;
    h->buf_gen++;
    h->d_ctr = (uint32_t *) p;
    h->d_status = h->d_ctr + CTR_WORDS;
    return SDMESH_OK;
}

// one memset: counters, status and the first lvl_words per-level tile counters
int zero_ctl(sdmesh_handle *h, size_t lvl_words) {
    HIPCHK(hipMemsetAsync(h->ctl.p, 0, 4ull * (CTL_LVL0 + lvl_words), h->stream));
    return SDMESH_OK;
}

// Weld a raw / compact slot list on the device: ordered compaction of the valid slots, then the
// generic weld with every triangle corner as its own "unique vertex".
int weld_raw_device(sdmesh_handle *h, const Triangle *d_raw, uint32_t nslots, bool all_valid = false) {
    hipStream_t st = h->stream;
    if (nslots == 0) {
        h->n_tris = h->n_verts = 0;
        h->res_v = nullptr; h->res_i = nullptr;
        h->mesh_valid = true;
        return SDMESH_OK;
    }
    const uint32_t vcap = 3u * nslots;
    SDCHK(layout_mesh(h, (nslots + 4) / 5, vcap + 64));
    if (h->tcap < nslots) return fail(h, SDMESH_ERR_INTERNAL, "weld layout too small");
    HIPCHK(hipMemsetAsync(h->ff_arena.p, 0xFF, ff_bytes(h), st));
    const Vertex *corners;
    if (all_valid) {
        // a compact list (the merged slab lists of the multi-GPU root): every slot is a triangle, nothing to count or move
        HIPCHK(launch_set_u32(st, h->d_ctr + CTR_NTRI, nslots));
        HIPCHK(launch_raw_desc(st, h->grid, h->d_ctr + CTR_NTRI, (uint4 *) h->tdesc.p, (uint8_t *) h->flip.p, h->d_ctr + CTR_NTILES,
                               h->d_ctr + CTR_NUNIQ, (uint32_t *) h->tri_e.p));
        corners = (const Vertex *) d_raw;
    } else {
        SDCHK(ensure(h, h->compact, sizeof(Triangle) * (size_t) nslots));
        HIPCHK(launch_set_u32(st, h->d_ctr + CTR_NSLOTS, nslots));
        HIPCHK(launch_raw_count(st, h->tgrid, d_raw, h->d_ctr + CTR_NSLOTS, (uint32_t *) h->tilecnt.p));
        HIPCHK(launch_scan_tiles(st, (uint32_t *) h->tilecnt.p, h->d_ctr + CTR_NSLOTS, 256, h->d_ctr + CTR_NTRI, nslots, h->d_status));
        HIPCHK(launch_raw_compact(st, h->tgrid, d_raw, h->d_ctr + CTR_NSLOTS, (const uint32_t *) h->tilecnt.p,
                                  (Triangle *) h->compact.p, (uint32_t *) h->tri_e.p, nslots));
        HIPCHK(launch_raw_desc(st, h->grid, h->d_ctr + CTR_NTRI, (uint4 *) h->tdesc.p, (uint8_t *) h->flip.p, h->d_ctr + CTR_NTILES,
                               h->d_ctr + CTR_NUNIQ));
        corners = (const Vertex *) h->compact.p;
    }
    HIPCHK(launch_tile_claim(st, h->tgrid, corners, (const uint4 *) h->tdesc.p, h->d_ctr + CTR_NTILES,
                             h->wkeys, h->wsize, (uint32_t *) h->vslot.p, h->d_status, weld_mode(h, h->voxel_size), (uint32_t *) h->wrep_buf.p, h->wfm));
    SDCHK(mesh_weld(h, corners, (nslots + 255u) / 256u));
    return SDMESH_OK;
}

void fill_info(sdmesh_handle *h, sdmesh_regen_info *info, int levels, bool direct, float ms) {
    if (!info) return;
    memset(info, 0, sizeof *info);
    info->levels = (uint32_t) levels;
    for (int i = 0; i <= levels && i <= SDMESH_MAX_LEVELS; i++) info->voxel_count[i] = h->h_pinned[CTR_LEVEL0 + i];
    info->unique_edges = h->n_unique;
    info->triangles = h->n_tris;
    info->vertices = h->n_verts;
    info->voxel_size = h->voxel_size;
    memcpy(info->status, h->h_pinned + CTR_WORDS, sizeof info->status);
    info->used_direct_path = direct ? 1u : 0u;
    info->device_ms = ms;
}

// One refine level on the device: src list (count at ctr[L]) -> dst list (count to ctr[L+1]).
// cnt_region = this level's pre-zeroed tile counters.  If out_cap == 0 the output size is discovered
// with a host sync between the two kernels (exact allocation); otherwise nothing synchronises.
int refine_level(sdmesh_handle *h, int L, uint32_t n_cap_in, uint32_t out_cap, Point cs, uint32_t *cnt_region,
                 uint32_t *count_out, uint32_t *mesh_tilecnt = nullptr) {
    hipStream_t st = h->stream;
    const int src = h->cur, dst = 1 - h->cur;
    SDCHK(ensure(h, h->mask, n_cap_in));
    const uint32_t *n_in = h->d_ctr + CTR_LEVEL0 + L;
    uint32_t *n_out = h->d_ctr + CTR_LEVEL0 + L + 1;
    // 1 = one lane per parent (compiled scenes), else the 32-lane kernel with that parent chunk
    const int mask_mode = (h->mask_lanes == 1 || (h->mask_lanes == 0 && h->scene != SCENE_PROGRAM && h->scene != SCENE_BULB)) ? 1 : h->mask_chunk;
    uint32_t *signs = nullptr;
    if (mesh_tilecnt) {                 // last level of a planned chain: emit the cube indices too (buffers presized)
        SDCHK(ensure(h, h->signs, 4ull * n_cap_in));
        signs = (uint32_t *) h->signs.p;
    }
    HIPCHK(launch_refine_mask(st, h->tgrid, h->scene, h->d_sp, (const Point *) h->vox[src].p, n_in, cs, (uint8_t *) h->mask.p,
                              cnt_region, mask_mode, signs));
    SDCHK(prof_mark(h, "refine_mask"));
    uint32_t cap = out_cap;
    if (out_cap == 0) {
        HIPCHK(launch_sum_tiles(st, cnt_region, n_in, h->d_ctr + CTR_SCRATCH));
        uint32_t cnt = 0;
        HIPCHK(hipMemcpyAsync(&cnt, h->d_ctr + CTR_SCRATCH, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        cap = cnt;
        if (count_out) *count_out = cnt;
    }
    SDCHK(ensure(h, h->vox[dst], sizeof(Point) * (size_t) std::max<uint32_t>(cap, 64)));
    HIPCHK(launch_refine_emit(st, h->tgrid, (const Point *) h->vox[src].p, n_in, cs, (const uint8_t *) h->mask.p, cnt_region,
                              (Point *) h->vox[dst].p, cap, n_out, h->d_status, signs, signs ? (uint8_t *) h->cubeidx.p : nullptr,
                              mesh_tilecnt, tiles(n_cap_in) > PRESCAN_TILES));
    SDCHK(prof_mark(h, "refine_emit"));
    h->cur = dst;
    return SDMESH_OK;
}

uint32_t padded(uint32_t n) { return n + n / 4 + 4096; }

// levels the column kernels take care of (0 = level-by-level chain from the level-0 field)
int col_levels(const sdmesh_handle *h, int levels, uint32_t n0) {
    return (h->col_refine && !h->refine2 && levels >= 2 && n0) ? std::min(levels - 1, 3) : 0;
}

int ensure_col_scratch(sdmesh_handle *h, uint32_t ncols) {
    ncols *= SDM_COL_SEGS_MAX;              // units: a column may be cut into up to this many z segments
    SDCHK(ensure(h, h->col_list1, 256ull * ncols));
    SDCHK(ensure(h, h->col_mask2, 256ull * ncols));
    SDCHK(ensure(h, h->col_mask3, 2048ull * ncols));
    SDCHK(ensure(h, h->colinfo, 16ull * ncols));
    return SDMESH_OK;
}

// The planned, synchronisation-free device chain of one regen (capacities from the plan).  Every buffer has
// been sized by the caller, so nothing here allocates: the same code runs eagerly or under stream capture.
int enqueue_planned_chain(sdmesh_handle *h, int levels, const uint32_t *box, uint32_t n0, uint32_t lvl_words, Point *vs_out) {
    const uint32_t x0 = box[0], x1 = box[1], z0 = box[2], z1 = box[3];      // level-0 cell range of this part
    hipStream_t st = h->stream;
    // fork: the weld table is cleared on the side stream while the refine runs
    bool ff_in_kernel = h->ff_fork == 0;      // (the arena is allocated in 256-byte units, so rounding the clear up to 16 B stays inside it)
    if (h->ff_fork == 1) {
        HIPCHK(hipEventRecord(h->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        HIPCHK(hipMemsetAsync(h->ff_arena.p, 0xFF, ff_bytes(h), h->side));
        HIPCHK(hipEventRecord(h->ev_ff, h->side));
    } else if (h->ff_fork == 2 || n0 == 0) {
        HIPCHK(hipMemsetAsync(h->ff_arena.p, 0xFF, ff_bytes(h), st));
        ff_in_kernel = false;
    }
    Point vs;
    vs.x = vs.y = vs.z = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    h->prof_n = 0;
    SDCHK(prof_mark(h, "begin"));
    // (the control block is cleared by the first kernel itself: one graph node less in front of the chain)
    const int J = col_levels(h, levels, n0);
    uint32_t n_cap = n0;
    h->cur = 0;
    if (J) {
        // levels 0 .. J per level-0 column: two launches, whatever J is (the level-0 .. J-1 lists are never materialised)
        HIPCHK(launch_col_refine(st, h->scene, h->d_sp, x0, x1, z0, z1, J, (uint8_t *) h->col_list1.p, (uint8_t *) h->col_mask2.p,
                                 (uint8_t *) h->col_mask3.p, (uint4 *) h->colinfo.p, (uint32_t *) h->ctl.p, (uint32_t) (CTL_LVL0 + lvl_words),
                                 ff_in_kernel ? h->ff_arena.p : nullptr, (ff_bytes(h) + 15) & ~(size_t) 15, h->col_segs));
        SDCHK(prof_mark(h, "col_refine"));
        n_cap = padded(h->plan_counts[J]);
        HIPCHK(launch_col_emit(st, x0, x1, z0, z1, J, (const uint8_t *) h->col_list1.p, (const uint8_t *) h->col_mask2.p,
                               (const uint8_t *) h->col_mask3.p, (const uint4 *) h->colinfo.p, (Point *) h->vox[0].p, n_cap,
                               h->d_ctr + CTR_LEVEL0, h->d_status, h->col_segs));
        SDCHK(prof_mark(h, "col_emit"));
        for (int i = 0; i < J; i++) vs = half_size(vs);
    } else {
        if (n0) HIPCHK(launch_create_field(st, (Point *) h->vox[0].p, x0, x1, z0, z1, h->d_ctr + CTR_LEVEL0, (uint32_t *) h->ctl.p,
                                           (uint32_t) (CTL_LVL0 + lvl_words), ff_in_kernel ? h->ff_arena.p : nullptr, (ff_bytes(h) + 15) & ~(size_t) 15));
        else SDCHK(zero_ctl(h, lvl_words));
        SDCHK(prof_mark(h, "create_field"));
    }
    uint32_t *mesh_tc = nullptr;        // set when the last refine level also classifies (k_classify is then skipped)
    for (int L = J; L < levels;) {
        const Point cs = half_size(vs);
        if (h->refine2 && L + 2 <= levels) {
            // two levels in one step: the level L+1 list is never materialised, only counted
            const Point cs2 = half_size(cs);
            const uint32_t out_cap = padded(h->plan_counts[L + 2]);
            const int src = h->cur, dst = 1 - h->cur;
            uint32_t *cnt_region = h->d_ctr + CTL_LVL0 + h->lvl_off[L];
            HIPCHK(launch_refine2_mask(st, h->tgrid, h->scene, h->d_sp, (const Point *) h->vox[src].p, h->d_ctr + CTR_LEVEL0 + L, cs,
                                       (uint8_t *) h->mask.p, (uint64_t *) h->mask2x.p, cnt_region, h->d_ctr + CTR_LEVEL0 + L + 1));
            HIPCHK(launch_refine2_emit(st, h->tgrid, (const Point *) h->vox[src].p, h->d_ctr + CTR_LEVEL0 + L, cs,
                                       (const uint8_t *) h->mask.p, (const uint64_t *) h->mask2x.p, cnt_region,
                                       (Point *) h->vox[dst].p, out_cap, h->d_ctr + CTR_LEVEL0 + L + 2, h->d_status));
            h->cur = dst;
            vs = cs2;
            n_cap = out_cap;
            L += 2;
            continue;
        }
        const uint32_t out_cap = padded(h->plan_counts[L + 1]);
        if (h->emit_ci && L + 1 == levels) mesh_tc = h->d_ctr + CTL_LVL0 + h->lvl_off[levels];
        SDCHK(refine_level(h, L, std::max<uint32_t>(n_cap, 64), out_cap, cs, h->d_ctr + CTL_LVL0 + h->lvl_off[L], nullptr, mesh_tc));
        vs = cs;
        n_cap = out_cap;
        L++;
    }
    h->lane_hint = h->plan_uniq;
    h->tri_hint = h->plan_tris;
    const int rc_front = mesh_front(h, (const Point *) h->vox[h->cur].p, h->d_ctr + CTR_LEVEL0 + levels, vs, false, true, mesh_tc != nullptr, mesh_tc,
                                    !h->skip_weld);
    h->lane_hint = 0;
    h->tri_hint = 0;
    SDCHK(rc_front);
    h->ctl_published = false;
    if (!h->skip_weld) SDCHK(mesh_weld(h, (const Vertex *) h->uverts.p, tiles(h->cap_vox), h->publish_ctl != 0));
    *vs_out = vs;
    return SDMESH_OK;
}

// size every buffer the planned chain touches (idempotent; may allocate, so it runs before any capture)
int presize_planned_chain(sdmesh_handle *h, int levels, uint32_t n0, uint32_t zs_of_box, uint32_t *lvl_words_out) {
    const int J = col_levels(h, levels, n0);
    SDCHK(ensure(h, h->vox[0], sizeof(Point) * (size_t) std::max<uint32_t>(J ? padded(h->plan_counts[J]) : n0, 64)));
    if (J) SDCHK(ensure_col_scratch(h, n0 / std::max<uint32_t>(1u, zs_of_box)));
    SDCHK(layout_mesh(h, padded(h->plan_counts[levels]), padded(h->plan_uniq)));
    uint32_t lvl_words = 0, n_cap = J ? padded(h->plan_counts[J]) : n0;
    int cur = 0;
    for (int L = 0; L < J; L++) h->lvl_off[L] = 0;
    for (int L = J; L < levels;) {
        const int step = (h->refine2 && L + 2 <= levels) ? 2 : 1;
        h->lvl_off[L] = lvl_words;
        lvl_words += tiles(L == 0 ? n0 : padded(h->plan_counts[L])) + 16;       // (L == 0 only without the column kernels)
        SDCHK(ensure(h, h->mask, std::max<uint32_t>(n_cap, 64)));
        if (step == 2) SDCHK(ensure(h, h->mask2x, 8ull * std::max<uint32_t>(n_cap, 64)));
        if (step == 1 && L + 1 == levels && h->emit_ci) SDCHK(ensure(h, h->signs, 4ull * std::max<uint32_t>(n_cap, 64)));
        n_cap = padded(h->plan_counts[L + step]);
        SDCHK(ensure(h, h->vox[1 - cur], sizeof(Point) * (size_t) std::max<uint32_t>(n_cap, 64)));
        cur = 1 - cur;
        L += step;
    }
    h->lvl_off[levels] = lvl_words;                          // mesh-tile triangle counters (fused classification)
    lvl_words += tiles(padded(h->plan_counts[levels])) + 16;
    SDCHK(ensure_ctl(h, lvl_words));
    *lvl_words_out = lvl_words;
    return SDMESH_OK;
}

void drop_graph(sdmesh_handle *h) {
    if (h->graph_exec) (void) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void) hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
}

// phase 0: whole regen; 1: begin (enqueue the planned chain and return without waiting; without a plan only the request
// is recorded); 2: end (wait, validate, fall back to the synchronous path if a capacity was exceeded)
// SDMESH_SCENE_MANDELBULB is a FIELD scene: dense sampling and the preview take it, the mesh path does not -- the reference's Newton
// projection (signed_distance.cu:227-240: up to 10 000 steps of seven evaluations per vertex) does not converge on a distance ESTIMATE
// (measured with the oracle: more than half of the 32^3 level's vertices run past 16 steps, thousands to the limit; minutes per level)
int mesh_scene_ok(sdmesh_handle *h) {
    if (h->scene != SCENE_BULB) return SDMESH_OK;
    return fail(h, SDMESH_ERR_INVALID, "SDMESH_SCENE_MANDELBULB is a field scene (sdmesh_sample_dense*, sdmesh_render*): the reference's surface "
                                       "projection does not converge on a distance estimate, so it is not meshed");
}

int regen_impl(sdmesh_handle *h, int levels, int mode, int part, int nparts, sdmesh_regen_info *info, int phase = 0) {
    if (phase != 2) SDCHK(mesh_scene_ok(h));
    if (!h) return SDMESH_ERR_INVALID;
    if (phase == 2) {
        if (!h->pend_active) return fail(h, SDMESH_ERR_INVALID, "sdmesh_regen_end without sdmesh_regen_begin");
        levels = h->pend_levels; mode = h->pend_mode; part = h->pend_part; nparts = h->pend_nparts;
        h->pend_active = false;
    } else if (h->pend_active) {
        return fail(h, SDMESH_ERR_INVALID, "a regen is pending: call sdmesh_regen_end first");
    }
    if (levels < 0 || levels > SDMESH_MAX_LEVELS) return fail(h, SDMESH_ERR_INVALID, "levels must be 0..%d", SDMESH_MAX_LEVELS);
    if (mode != SDMESH_PART_NONE && mode != SDMESH_PART_ZSLAB && mode != SDMESH_PART_XSLAB) return fail(h, SDMESH_ERR_INVALID, "bad partition mode");
    if (mode == SDMESH_PART_NONE) { part = 0; nparts = 1; }
    if (nparts < 1 || nparts > 32 || part < 0 || part >= nparts) return fail(h, SDMESH_ERR_INVALID, "bad partition %d/%d", part, nparts);
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    hipStream_t st = h->stream;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR;
    uint32_t p0 = F * (uint32_t) part / (uint32_t) nparts, p1 = F * (uint32_t) (part + 1) / (uint32_t) nparts;
    if (mode != SDMESH_PART_NONE && h->slab_nparts == nparts && h->slab_mode == mode) { p0 = h->slab_bounds[part]; p1 = h->slab_bounds[part + 1]; }
    // level-0 cell box {x0, x1, z0, z1} of this part: z layers (ZSLAB) or x layers (XSLAB: x is the outermost digit of
    // the reference's order, so a part's voxel / triangle lists are contiguous ranges of the unpartitioned ones)
    const uint32_t box[4] = {mode == SDMESH_PART_XSLAB ? p0 : 0u, mode == SDMESH_PART_XSLAB ? p1 : F,
                             mode == SDMESH_PART_ZSLAB ? p0 : 0u, mode == SDMESH_PART_ZSLAB ? p1 : F};
    const uint32_t n0 = (box[1] - box[0]) * F * (box[3] - box[2]);

    if (h->weld_wide) h->wide_active = true;
    for (int attempt = 0; attempt < 3; attempt++) {
        const bool resumed = phase == 2 && attempt == 0 && h->pend_async;
        const bool async = resumed || (h->plan_valid && h->plan_scene == h->scene && h->plan_levels == levels && h->plan_mode == mode &&
                                       h->plan_part == part && h->plan_nparts == nparts && attempt == 0);
        h->mesh_valid = h->vox_valid = false;
        h->res_v = nullptr; h->res_i = nullptr;
        if (h->colcnt && attempt > 0) HIPCHK(hipMemsetAsync(h->colcnt, 0, 4096, st));     // (a sharded local phase: the column counts of the attempt before)
        Point vs;
        vs.x = vs.y = vs.z = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
        uint32_t n_cap = n0;
        bool direct = h->force_direct != 0;
        bool timed = true;                  // ev0 / ev1 bracket the device work of this attempt
        if (resumed) {
            // the planned chain (and its read-back) were enqueued by regen_begin: just wait for them
            vs = h->pend_vs;
            n_cap = h->pend_ncap;
            direct = false;
            timed = h->pend_timed;              // (latched by regen_begin: the events exist only if "timing" was on THEN)
            HIPCHK(hipStreamSynchronize(st));
            SDCHK(unpoison_pinned(h));
        } else if (phase == 1 && !(async && !direct)) {
            // no capacity plan yet: nothing can be enqueued blindly; regen_end will run the discovery path
            h->pend_active = true; h->pend_async = false;
            h->pend_levels = levels; h->pend_mode = mode; h->pend_part = part; h->pend_nparts = nparts;
            return SDMESH_OK;
        } else if (async && !direct) {
            h->cur = 0;
            // ---- planned path: no host round trip until the final read-back; optionally replayed as a hipGraph
            uint32_t lvl_words = 0;
            SDCHK(presize_planned_chain(h, levels, n0, box[3] - box[2], &lvl_words));
            n_cap = padded(h->plan_counts[levels]);
            poison_pinned(h);
            if (h->use_graph && st != nullptr && !h->profile && !h->project_diag && !h->colcnt && !h->out_v) {   // (a sharded local phase's hooks alternate: eager there)   // (the legacy null stream cannot be captured: eager launches there)
                uint64_t key[8] = {h->buf_gen, (uint64_t) h->scene, (uint64_t) levels, ((uint64_t) mode << 40) | ((uint64_t) part << 20) | (uint64_t) nparts,
                                   (uint64_t) h->skip_weld | ((uint64_t) h->wide_active << 1) | ((uint64_t) h->col_refine << 2) | ((uint64_t) h->claim_in_project << 3) | ((uint64_t) h->fuse_flip << 4) | ((uint64_t) h->publish_ctl << 6) | ((uint64_t) h->local_prefix << 7) | ((uint64_t) h->project_batch << 8) | ((uint64_t) h->project_grid << 12) | ((uint64_t) h->project_prio << 30) | ((uint64_t) h->tf_threads << 32) | ((uint64_t) h->col_segs << 44), (uint64_t) (uintptr_t) st, (uint64_t) h->grid | ((uint64_t) h->tgrid << 20) | ((uint64_t) h->project_mult << 40) | ((uint64_t) h->flip_mult << 48) | ((uint64_t) h->refine2 << 56) | ((uint64_t) h->emit_ci << 57) | ((uint64_t) h->project_lanes << 58) | ((uint64_t) h->flip_lanes << 61), 0};
                for (int i = 0; i <= levels; i++) key[7] = key[7] * 1000003ull + h->plan_counts[i];
                key[7] = key[7] * 1000003ull + h->plan_uniq;
                key[7] = key[7] * 1000003ull + h->plan_tris;
                if (!h->graph_exec || memcmp(key, h->graph_key, sizeof key) != 0) {
                    drop_graph(h);
                    HIPCHK(hipStreamSynchronize(st));
                    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                    int rc = enqueue_planned_chain(h, levels, box, n0, lvl_words, &vs);
                    if (rc == SDMESH_OK && !h->ctl_published) rc = read_back_enqueue(h);
                    hipGraph_t g = nullptr;
                    hipError_t ce = hipStreamEndCapture(st, &g);
                    if (rc != SDMESH_OK) { if (g) (void) hipGraphDestroy(g); return rc; }
                    if (ce != hipSuccess) return fail(h, SDMESH_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
                    h->graph = g;
                    HIPCHK(hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0));
                    memcpy(h->graph_key, key, sizeof key);
                    h->graph_cur = h->cur;
                    h->graph_vs = vs;
                }
                h->cur = h->graph_cur;
                vs = h->graph_vs;
                if (h->timing) HIPCHK(hipEventRecord(h->ev0, st));
                HIPCHK(hipGraphLaunch(h->graph_exec, st));
                if (h->timing) HIPCHK(hipEventRecord(h->ev1, st));
            } else {
                if (h->timing) HIPCHK(hipEventRecord(h->ev0, st));
                SDCHK(enqueue_planned_chain(h, levels, box, n0, lvl_words, &vs));
                if (h->timing) HIPCHK(hipEventRecord(h->ev1, st));
                if (!h->ctl_published) SDCHK(read_back_enqueue(h));
            }
            timed = h->timing != 0;
            if (phase == 1) {
                h->pend_active = true; h->pend_async = true;
                h->pend_levels = levels; h->pend_mode = mode; h->pend_part = part; h->pend_nparts = nparts;
                h->pend_vs = vs; h->pend_ncap = n_cap; h->pend_timed = timed;
                return SDMESH_OK;
            }
            HIPCHK(hipStreamSynchronize(st));
            SDCHK(unpoison_pinned(h));
            if (h->profile) SDCHK(prof_collect(h));
        } else {
            // ---- discovery path: sizes become known level by level (one host sync per level)
            h->cur = 0;
            SDCHK(ensure(h, h->vox[0], sizeof(Point) * (size_t) std::max<uint32_t>(n0, 64)));
            n_cap = n0;
            HIPCHK(hipEventRecord(h->ev0, st));
            SDCHK(zero_ctl(h, 0));
            const int J = col_levels(h, levels, n0);
            if (J) {
                // column kernels: lists first, then the totals (count-only emit + one host sync), then the level-J voxel list
                SDCHK(ensure_col_scratch(h, n0 / (box[3] - box[2])));
                HIPCHK(launch_col_refine(st, h->scene, h->d_sp, box[0], box[1], box[2], box[3], J, (uint8_t *) h->col_list1.p,
                                         (uint8_t *) h->col_mask2.p, (uint8_t *) h->col_mask3.p, (uint4 *) h->colinfo.p, nullptr, 0, nullptr, 0, h->col_segs));
                for (int pass = 0; pass < 2; pass++) {
                    HIPCHK(launch_col_emit(st, box[0], box[1], box[2], box[3], J, (const uint8_t *) h->col_list1.p, (const uint8_t *) h->col_mask2.p,
                                           (const uint8_t *) h->col_mask3.p, (const uint4 *) h->colinfo.p, pass ? (Point *) h->vox[0].p : nullptr,
                                           n_cap, h->d_ctr + CTR_LEVEL0, h->d_status, h->col_segs));
                    if (pass) break;
                    uint32_t cnt = 0;
                    HIPCHK(hipMemcpyAsync(&cnt, h->d_ctr + CTR_LEVEL0 + J, 4, hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    n_cap = cnt;
                    SDCHK(ensure(h, h->vox[0], sizeof(Point) * (size_t) std::max<uint32_t>(cnt, 64)));
                }
                for (int i = 0; i < J; i++) vs = half_size(vs);
            } else if (n0) {
                HIPCHK(launch_create_field(st, (Point *) h->vox[0].p, box[0], box[1], box[2], box[3], h->d_ctr + CTR_LEVEL0));
            }
            for (int L = J; L < levels; L++) {
                const Point cs = half_size(vs);
                uint32_t cnt = 0;
                h->lvl_off[L] = 0;
                SDCHK(ensure_ctl(h, tiles(n_cap) + 16));
                HIPCHK(hipMemsetAsync(h->d_ctr + CTL_LVL0, 0, 4ull * (tiles(n_cap) + 16), st));
                SDCHK(refine_level(h, L, std::max<uint32_t>(n_cap, 64), 0u, cs, h->d_ctr + CTL_LVL0, &cnt));
                vs = cs;
                n_cap = cnt;
            }
            if (!direct) {
                SDCHK(layout_mesh(h, n_cap, 0));
                SDCHK(mesh_front(h, (const Point *) h->vox[h->cur].p, h->d_ctr + CTR_LEVEL0 + levels, vs, false, false));
                if (!h->skip_weld) SDCHK(mesh_weld(h, (const Vertex *) h->uverts.p, tiles(h->cap_vox)));
            }
            HIPCHK(hipEventRecord(h->ev1, st));
            SDCHK(read_back(h));
        }
        const Point *d_vox = (const Point *) h->vox[h->cur].p;
        const uint32_t *n_ptr = h->d_ctr + CTR_LEVEL0 + levels;
        const uint32_t *c = h->h_pinned, *s = h->h_pinned + CTR_WORDS;
        if (s[ST_WELD_RANGE] && !h->wide_active) {
            // a vertex does not fit the packed key: the reference's [i64; 3] key has no range (src/cuda/mod.rs:268-273),
            // so run again with the wide key form (sticky for this handle)
            h->wide_active = true; h->plan_valid = false; drop_graph(h);
            continue;
        }
        if (!direct && s[ST_OVERFLOW]) {
            if (async) { h->plan_valid = false; drop_graph(h); continue; }   // capacities were too tight: redo synchronously
            // synchronous path: voxel buffers are exact, so only the unique-edge estimate can be too
            // small -- lay out for the observed count and run the mesh stage again
            h->plan_valid = false;
            SDCHK(layout_mesh(h, n_cap, c[CTR_NUNIQ] + 1024));
            SDCHK(zero_status(h));
            SDCHK(mesh_front(h, d_vox, n_ptr, vs, true));
            if (!h->skip_weld) SDCHK(mesh_weld(h, (const Vertex *) h->uverts.p, tiles(h->cap_vox)));
            HIPCHK(hipEventRecord(h->ev1, st));
            SDCHK(read_back(h));
            if (s[ST_OVERFLOW]) return fail(h, SDMESH_ERR_INTERNAL, "capacity overflow persists (status %u)", s[ST_OVERFLOW]);
        }
        // not expected for create+refine fields of the compiled scenes; handled anyway.  ST_NONFINITE: a projected vertex
        // with a non-finite x -- the reference drops a triangle whose (post-flip) first vertex is one (src/cuda/mod.rs:289),
        // which the per-slot direct path reproduces literally
        if (!direct && (s[ST_NOT_LATTICE] || s[ST_TILE_OVERFLOW] || s[ST_NONFINITE])) direct = true;
        const uint32_t n = c[CTR_LEVEL0 + levels];
        if (direct) {
            SDCHK(ensure(h, h->raw, sizeof(Triangle) * 5ull * std::max<uint32_t>(n, 1)));
            SDCHK(zero_status(h));
            HIPCHK(launch_triangles_direct(st, h->scene, h->d_sp, d_vox, n, vs, (Triangle *) h->raw.p));
            SDCHK(weld_raw_device(h, (const Triangle *) h->raw.p, 5u * n));
            HIPCHK(hipEventRecord(h->ev1, st));
            // keep the level counters: weld_raw_device only touches the mesh counters
            SDCHK(read_back(h));
            if (s[ST_WELD_RANGE] && !h->wide_active) {
                h->wide_active = true; h->plan_valid = false; drop_graph(h);
                SDCHK(zero_status(h));
                SDCHK(weld_raw_device(h, (const Triangle *) h->raw.p, 5u * n));
                HIPCHK(hipEventRecord(h->ev1, st));
                SDCHK(read_back(h));
            }
        }
        SDCHK(status_to_error(h));
        float ms = 0.0f;
        if (timed) HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->n_vox = n;
        h->n_unique = c[CTR_NUNIQ];
        h->n_tris = c[CTR_NTRI];
        h->n_verts = c[CTR_NVERT];
        h->voxel_size = vs;
        h->mesh_valid = h->vox_valid = true;
        h->last_direct = direct;
        h->welded = direct || !h->skip_weld;
        if (!direct) {
            h->plan_valid = true;
            h->plan_scene = h->scene; h->plan_levels = levels; h->plan_mode = mode; h->plan_part = part; h->plan_nparts = nparts;
            for (int i = 0; i <= levels; i++) h->plan_counts[i] = std::max(h->plan_counts[i] * (async ? 1u : 0u), c[CTR_LEVEL0 + i]);
            h->plan_uniq = std::max(async ? h->plan_uniq : 0u, c[CTR_NUNIQ]);
            h->plan_tris = std::max(async ? h->plan_tris : 0u, c[CTR_NTRI]);
        }
        fill_info(h, info, levels, direct, ms);
        return SDMESH_OK;
    }
    return fail(h, SDMESH_ERR_INTERNAL, "regen did not converge on a capacity plan");
}

// shared front half of the host-buffer mesh calls: upload, lattice path, fall back to direct
// when the field is not a lattice field (or when forced).  On return either (uverts, tri_e) are
// valid (direct == false) or h->raw holds the 5n raw slots (direct == true).
int host_mesh_front(sdmesh_handle *h, const Point *voxels, uint32_t n, Point vs, bool &direct, bool with_weld = false) {
    // with_weld: the weld is enqueued behind the front half and both are waited for together (one host synchronisation per attempt; the weld
    // of a front half that turns out to need another route is wasted work, which only a non-lattice field or a first call pays)
    hipStream_t st = h->stream;
    SDCHK(ensure(h, h->vox[0], sizeof(Point) * (size_t) n));
    h->cur = 0;
    h->vox_valid = false;
    h->mesh_valid = false;
    HIPCHK(hipMemcpyAsync(h->vox[0].p, voxels, sizeof(Point) * (size_t) n, hipMemcpyHostToDevice, st));
    SDCHK(zero_ctl(h, 0));
    HIPCHK(launch_set_u32(st, h->d_ctr + CTR_LEVEL0, n));
    direct = h->force_direct != 0;
    if (!direct) {
        SDCHK(layout_mesh(h, n, 0));
        for (int attempt = 0; attempt < 2; attempt++) {
            if (with_weld) {
                // size hints as the planned chain has them (grid sizes and kernel forms only: any count is handled): a refined field holds
                // about 1.2 pre-weld vertices and 2 triangles per voxel; with them the claims run in the projection kernel and the orientation
                // test inside k_tile_first, as on the device-resident path
                h->lane_hint = n + n / 4u; h->tri_hint = 2u * n;
            }
            const int frc = mesh_front(h, (const Point *) h->vox[0].p, h->d_ctr + CTR_LEVEL0, vs, true, false, false, nullptr, with_weld);
            h->lane_hint = 0; h->tri_hint = 0;
            SDCHK(frc);
            if (with_weld) SDCHK(mesh_weld(h, (const Vertex *) h->uverts.p, tiles(h->cap_vox)));
            SDCHK(read_back(h));
            const uint32_t *s = h->h_pinned + CTR_WORDS;
            if (s[ST_NOT_LATTICE] || s[ST_TILE_OVERFLOW] || s[ST_NONFINITE]) { direct = true; break; }
            if (!s[ST_OVERFLOW]) break;
            if (attempt == 1) return fail(h, SDMESH_ERR_INTERNAL, "capacity overflow persists");
            SDCHK(layout_mesh(h, n, h->h_pinned[CTR_NUNIQ] + 1024));
            SDCHK(zero_status(h));
        }
    }
    if (direct) {
        SDCHK(ensure(h, h->raw, sizeof(Triangle) * 5ull * n));
        SDCHK(zero_status(h));
        HIPCHK(launch_triangles_direct(st, h->scene, h->d_sp, (const Point *) h->vox[0].p, n, vs, (Triangle *) h->raw.p));
    }
    h->n_vox = n;
    h->voxel_size = vs;
    return SDMESH_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int sdmesh_create(int device, sdmesh_handle **out) {
    if (!out) return SDMESH_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return SDMESH_ERR_NO_DEVICE;
    sdmesh_handle *h = new sdmesh_handle();
    h->device = device;
    auto bail = [&](hipError_t e, const char *what) {
        fprintf(stderr, "sdmesh_create: %s failed: %s\n", what, hipGetErrorString(e));
        sdmesh_destroy(h);
        return SDMESH_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "sdmesh_create: device %d is %s; this library carries gfx950 code only\n", device, prop.gcnArchName);
        sdmesh_destroy(h);
        return SDMESH_ERR_NO_DEVICE;
    }
    h->grid = 4 * prop.multiProcessorCount;
    h->tgrid = 16 * prop.multiProcessorCount;   // measured: 512^3 (1037 tiles) -7 % vs 4/CU, 1024^3 (4150 tiles) best at 16/CU
    h->dense_grid = 48 * prop.multiProcessorCount;   // measured best for the write-only P1 stream (round-1 sweep over 4 .. 64 per CU)
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    h->stream = h->own_stream;
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&h->ev_ff, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&h->ctl.p, 4ull * (CTL_LVL0 + 65536))) != hipSuccess) return bail(e, "hipMalloc");
    h->ctl.cap = 4ull * (CTL_LVL0 + 65536);
    h->d_ctr = (uint32_t *) h->ctl.p;
    h->d_status = h->d_ctr + CTR_WORDS;
    if ((e = hipMemset(h->ctl.p, 0, h->ctl.cap)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipHostMalloc((void **) &h->h_pinned, (CTR_WORDS + ST_WORDS) * 4, hipHostMallocDefault)) != hipSuccess)
        return bail(e, "hipHostMalloc");
    memset(h->h_pinned, 0, (CTR_WORDS + ST_WORDS) * 4);
    if ((e = hipMalloc((void **) &h->d_sp, sizeof(SceneParams))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipHostMalloc((void **) &h->h_sp, sizeof(SceneParams), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
#if !defined(SDM_FAST_MATH)
    // The strict build's bit-exactness rests on a property of the silicon (one Markstein step on v_rsq_f32 / v_rcp_f32 rounds correctly:
    // csrc/sdf.hpp), proven by enumeration on the tested gfx950 parts.  Prove it on THIS device too, once per process and device (about a
    // second; SDMESH_SKIP_SELFCHECK=1 skips it): a part on which it does not hold must not silently produce other bits.
    {
        static bool checked[64] = {false};
        const char *skip = getenv("SDMESH_SKIP_SELFCHECK");
        if (device < 64 && !checked[device] && !(skip && skip[0] == '1')) {
            uint64_t bad[SDMESH_SELFCHECKS] = {0};
            const int rc = sdmesh_selfcheck_arithmetic(h, bad, SDMESH_SELFCHECKS);
            uint64_t sum = 0;
            for (int i = 0; i < SDMESH_SELFCHECKS; i++) sum += bad[i];
            if (rc != SDMESH_OK || sum != 0) {
                fprintf(stderr, "sdmesh_create: the lean exact arithmetic is NOT exact on device %d (mismatches %llu %llu %llu %llu %llu, rc %d): this build's "
                                "results would differ from IEEE-754; refusing\n", device, (unsigned long long) bad[0], (unsigned long long) bad[1],
                        (unsigned long long) bad[2], (unsigned long long) bad[3], (unsigned long long) bad[4], rc);
                sdmesh_destroy(h);
                return SDMESH_ERR_INTERNAL;
            }
            checked[device] = true;
        }
    }
#endif
    *out = h;
    return SDMESH_OK;
}

void sdmesh_destroy(sdmesh_handle *h) {
    if (!h) return;
    (void) hipSetDevice(h->device);
    if (h->stream) (void) hipStreamSynchronize(h->stream);
    drop_graph(h);
    DevBuf *bufs[] = {&h->render_tex, &h->vox[0], &h->vox[1], &h->mask, &h->mask2x, &h->signs, &h->tilecnt, &h->ctl, &h->cubeidx, &h->trioff, &h->ff_arena,
                      &h->first_ord_buf, &h->uverts, &h->tri_e, &h->vslot, &h->tdesc, &h->tfirst, &h->tfpre, &h->tpre, &h->flip, &h->wvidx, &h->torder, &h->wrep_buf, &h->wfm_buf,
                      &h->vertices, &h->indices, &h->raw, &h->compact, &h->staging, &h->dense_samples, &h->dense_bits,
                      &h->cellcnt, &h->dist_upids, &h->dist_upkeys, &h->dist_loids, &h->dist_prevkeys, &h->dist_hkeys, &h->dist_hpos, &h->dist_dupof,
                      &h->dist_newidx, &h->dist_tilecnt, &h->dist_counts, &h->dist_upg, &h->dist_prevg, &h->dist_verts, &h->dist_nups, &h->col_list1, &h->col_mask2, &h->col_mask3, &h->colinfo, &h->prof_rounds, &h->proj_diag, &h->shard_mine, &h->shard_counts, &h->shard_recv, &h->shard_lists, &h->shard_plan, &h->shard_merged, &h->zs_verts, &h->zs_idx, &h->zs_tables, &h->zs_pervert, &h->zs_hkeys, &h->zs_vout, &h->zs_iout, &h->zs_ranks, &h->zs_ctl, &h->shard_counts2, &h->zs_colscratch, &h->zs_hints};
    (void) sdmesh_comm_destroy(h);
    if (h->h_counts) (void) hipHostFree(h->h_counts);
    if (h->h_lists) (void) hipHostFree(h->h_lists);
    if (h->h_zs) (void) hipHostFree(h->h_zs);
    if (h->h_zs_pub) (void) hipHostFree(h->h_zs_pub);
    if (h->h_dist_pub) (void) hipHostFree(h->h_dist_pub);
    for (DevBuf *b : bufs) release(*b);
    if (h->h_pinned) (void) hipHostFree(h->h_pinned);
    if (h->d_sp) (void) hipFree(h->d_sp);
    if (h->h_sp) (void) hipHostFree(h->h_sp);
    if (h->ev0) (void) hipEventDestroy(h->ev0);
    if (h->ev1) (void) hipEventDestroy(h->ev1);
    if (h->side) { (void) hipStreamSynchronize(h->side); (void) hipStreamDestroy(h->side); }
    if (h->ev_fork) (void) hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void) hipEventDestroy(h->ev_join);
    if (h->ev_ff) (void) hipEventDestroy(h->ev_ff);
    for (hipEvent_t e : h->prof_ev) (void) hipEventDestroy(e);
    if (h->own_stream) (void) hipStreamDestroy(h->own_stream);
    delete h;
}

const char *sdmesh_last_error(const sdmesh_handle *h) { return h ? h->err : "null handle"; }

int sdmesh_set_scene(sdmesh_handle *h, int scene, const float *params, int nparams) {
    if (!h) return SDMESH_ERR_INVALID;
    if (scene < 0 || scene >= SCENE_COUNT || scene == SCENE_PROGRAM) return fail(h, SDMESH_ERR_INVALID, "unknown scene %d (programs: sdmesh_set_program)", scene);
    if (nparams < 0 || nparams > 4 || (nparams > 0 && !params)) return fail(h, SDMESH_ERR_INVALID, "bad scene params");
    h->scene = scene;
    h->sp.ninstr = 0;
    for (int i = 0; i < 4; i++) h->sp.p[i] = i < nparams ? params[i] : 0.0f;
    if (scene == SCENE_BULB) {
        if (nparams < 1) h->sp.p[0] = 1.0f;                    // scale (sd_unit_mandelbulb's 0.4 is params = {0.4})
        if (!(h->sp.p[0] > 0.0f)) return fail(h, SDMESH_ERR_INVALID, "SDMESH_SCENE_MANDELBULB: params[0] (scale) must be positive");
    }
    h->sp_dirty = true;
    return SDMESH_OK;
}

int sdmesh_set_program(sdmesh_handle *h, const sdmesh_instr *instrs, int n) {
    if (!h || !instrs) return SDMESH_ERR_INVALID;
    if (n < 1 || n > SDM_MAX_INSTR) return fail(h, SDMESH_ERR_INVALID, "a program has 1..%d instructions", SDM_MAX_INSTR);
    static_assert(sizeof(sdmesh_instr) == sizeof(SdfInstr), "sdmesh_instr layout");
    for (int i = 0; i < n; i++) {
        if (instrs[i].op < OP_SPHERE || instrs[i].op > OP_LAST || instrs[i].combine > CB_SUBTRACT)
            return fail(h, SDMESH_ERR_INVALID, "instruction %d: unknown op/combine", i);
    }
    int first = 0;
    while (first < n && instrs[first].op == OP_WRAP) first++;              // (wraps only move the point)
    if (first == n || instrs[first].combine != CB_SET) return fail(h, SDMESH_ERR_INVALID, "the first primitive must use SDMESH_CB_SET");
    h->scene = SCENE_PROGRAM;
    memset(&h->sp, 0, sizeof h->sp);
    h->sp.ninstr = (uint32_t) n;
    memcpy(h->sp.prog, instrs, sizeof(SdfInstr) * (size_t) n);
    memcpy(h->prog_base, instrs, sizeof(SdfInstr) * (size_t) n);
    memset(h->prog_motion, 0, sizeof h->prog_motion);
    h->prog_time = 0.0f;
    h->sp_dirty = true;
    h->plan_valid = false;
    return SDMESH_OK;
}

namespace {
// a[i][j] + time * da[i][j] in f32, the multiplication rounded before the addition (the oracle's Python helper does the same)
void apply_program_time(sdmesh_handle *h) {
    for (uint32_t i = 0; i < h->sp.ninstr; i++)
        for (int j = 0; j < 7; j++) {
            volatile float prod = h->prog_time * h->prog_motion[i][j];
            h->sp.prog[i].a[j] = h->prog_base[i].a[j] + prod;
        }
    h->sp_dirty = true;
}
}  // namespace

int sdmesh_set_program_motion(sdmesh_handle *h, const float *motion, int n) {
    if (!h) return SDMESH_ERR_INVALID;
    if (h->scene != SCENE_PROGRAM) return fail(h, SDMESH_ERR_INVALID, "no program is set (sdmesh_set_program)");
    if (motion && n != (int) h->sp.ninstr) return fail(h, SDMESH_ERR_INVALID, "the motion table needs one row of 7 rates per instruction (%u)", h->sp.ninstr);
    memset(h->prog_motion, 0, sizeof h->prog_motion);
    if (motion) memcpy(h->prog_motion, motion, sizeof(float) * 7 * (size_t) n);
    apply_program_time(h);
    return SDMESH_OK;
}

int sdmesh_set_program_time(sdmesh_handle *h, float time) {
    if (!h) return SDMESH_ERR_INVALID;
    if (h->scene != SCENE_PROGRAM) return fail(h, SDMESH_ERR_INVALID, "no program is set (sdmesh_set_program)");
    h->prog_time = time;
    apply_program_time(h);
    return SDMESH_OK;
}

int sdmesh_set_stream(sdmesh_handle *h, void *hip_stream) {
    if (!h) return SDMESH_ERR_INVALID;
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->stream = (hipStream_t) hip_stream;          // NULL = the legacy default stream, exactly as HIP itself reads it
    return SDMESH_OK;
}

int sdmesh_use_own_stream(sdmesh_handle *h) {
    if (!h) return SDMESH_ERR_INVALID;
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->stream = h->own_stream;
    return SDMESH_OK;
}

int sdmesh_set_option(sdmesh_handle *h, const char *name, int value) {
    if (!h || !name) return SDMESH_ERR_INVALID;
    if (!strcmp(name, "direct")) { h->force_direct = value ? 1 : 0; h->plan_valid = false; return SDMESH_OK; }
    if (!strcmp(name, "project_mult")) { if (value < 1 || value > 16) return fail(h, SDMESH_ERR_INVALID, "project_mult 1..16"); h->project_mult = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "project_lanes")) { if (value != 0 && value != 1 && value != 2 && value != 4) return fail(h, SDMESH_ERR_INVALID, "project_lanes 0|1|2|4"); h->project_lanes = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "flip_lanes")) { if (value != 0 && value != 1 && value != 4) return fail(h, SDMESH_ERR_INVALID, "flip_lanes 0|1|4"); h->flip_lanes = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "ff_fork")) { if (value < 0 || value > 2) return fail(h, SDMESH_ERR_INVALID, "ff_fork 0|1|2"); h->ff_fork = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "mask_lanes")) { if (value != 0 && value != 1 && value != 32) return fail(h, SDMESH_ERR_INVALID, "mask_lanes 0|1|32"); h->mask_lanes = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "fork")) { if (value < 0 || value > 2) return fail(h, SDMESH_ERR_INVALID, "fork 0|1|2"); h->fork = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "flip_mult")) { if (value < 1 || value > 16) return fail(h, SDMESH_ERR_INVALID, "flip_mult 1..16"); h->flip_mult = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "mask_chunk")) { if (value != 8 && value != 16 && value != 32) return fail(h, SDMESH_ERR_INVALID, "mask_chunk 8|16|32"); h->mask_chunk = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "refine2")) { h->refine2 = value ? 1 : 0; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "col_refine")) { h->col_refine = value ? 1 : 0; h->plan_valid = false; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "emit_ci")) { h->emit_ci = value ? 1 : 0; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "skip_weld")) { h->skip_weld = value ? 1 : 0; return SDMESH_OK; }
    if (!strcmp(name, "fast_math")) {
        // the numeric mode is a property of the BUILD (like the reference's nvcc --use_fast_math): libsdmesh.so is strict IEEE,
        // libsdmesh_fast.so contracts FMAs and takes approximate divide / sqrt.  The option only confirms which one is loaded.
#if defined(SDM_FAST_MATH)
        const int built = 1;
#else
        const int built = 0;
#endif
        if ((value ? 1 : 0) != built)
            return fail(h, SDMESH_ERR_INVALID, built ? "this is the fast-math build (libsdmesh_fast.so); load libsdmesh.so for strict IEEE"
                                                     : "this is the strict IEEE build (libsdmesh.so); load libsdmesh_fast.so for the fast numeric mode");
        return SDMESH_OK;
    }
    if (!strcmp(name, "col_segs")) { h->col_segs = value < 1 ? 1 : (value > (int) SDM_COL_SEGS_MAX ? (int) SDM_COL_SEGS_MAX : value); drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "tf_threads")) {
        if (value < 128 || value > 1024 || value % 64) return fail(h, SDMESH_ERR_INVALID, "tf_threads: a multiple of 64 in 128 .. 1024");
        h->tf_threads = value; drop_graph(h); return SDMESH_OK;
    }
    if (!strcmp(name, "inject_fail")) { h->inject_fail = value; return SDMESH_OK; }
    if (!strcmp(name, "shard_indexed")) { h->shard_indexed = value != 0; h->zsp.valid = false; return SDMESH_OK; }
    if (!strcmp(name, "colcnt_debug")) {      // diagnostic: plain regens accumulate column counts into a scratch row (the cost of that hook alone)
        if (value) { if (ensure(h, h->zs_ctl, 4ull * ZC_WORDS + 8192) != SDMESH_OK || ensure(h, h->zs_colscratch, 8192) != SDMESH_OK) return SDMESH_ERR_HIP; h->colcnt = (uint32_t *) h->zs_ctl.p + ZC_WORDS; }
        else h->colcnt = nullptr;
        return SDMESH_OK;
    }
    if (!strcmp(name, "zs_block")) { if (value != 0 && value != 256 && value != 1024) return fail(h, SDMESH_ERR_INVALID, "zs_block: 0, 256 or 1024"); h->zs_block = value; return SDMESH_OK; }
    if (!strcmp(name, "shard_slack")) { h->shard_slack = value != 0; h->zsp.valid = false; h->dsp.valid = false; return SDMESH_OK; }
    if (!strcmp(name, "shard_planned")) { h->shard_planned = value != 0; h->zsp.valid = false; h->dsp.valid = false; return SDMESH_OK; }   // (like every option of a collective call: the same on all ranks)
    if (!strcmp(name, "project_prio")) { h->project_prio = value != 0; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "project_grid")) { h->project_grid = value < 0 ? 0 : (value > 65535 ? 65535 : value); drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "project_batch")) { h->project_batch = value < 0 ? 0 : (value > 6 ? 6 : value); drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "timing")) { h->timing = value != 0; return SDMESH_OK; }
    if (!strcmp(name, "download_streams")) { h->download_streams = value == 1 ? 1 : 2; return SDMESH_OK; }
    if (!strcmp(name, "local_prefix")) { h->local_prefix = value != 0; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "publish_ctl")) { h->publish_ctl = value != 0; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "fuse_flip")) { h->fuse_flip = value < 0 ? 0 : (value > 2 ? 2 : value); drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "claim")) { h->claim_in_project = value < 0 ? 0 : (value > 3 ? 3 : value); drop_graph(h); return SDMESH_OK; }   // 2: never asynchronous, 3: always
    if (!strcmp(name, "weld_shift")) { if (value < -1 || value > 20) return fail(h, SDMESH_ERR_INVALID, "weld_shift -1..20"); h->weld_shift = value; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "profile")) { h->profile = value ? 1 : 0; h->prof_n = 0; h->prof_ms.clear(); return SDMESH_OK; }
    if (!strcmp(name, "project_diag")) { h->project_diag = value; return SDMESH_OK; }
    if (!strcmp(name, "weld_wide")) { h->weld_wide = value ? 1 : 0; h->wide_active = value != 0; drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "graph")) { h->use_graph = value ? 1 : 0; if (!value) drop_graph(h); return SDMESH_OK; }
    if (!strcmp(name, "tgrid")) {
        if (value < 1 || value > 65535) return fail(h, SDMESH_ERR_INVALID, "tgrid out of range");
        h->tgrid = value; drop_graph(h);
        return SDMESH_OK;
    }
    if (!strcmp(name, "grid")) {
        if (value < 1 || value > 65535) return fail(h, SDMESH_ERR_INVALID, "grid out of range");
        h->grid = value;
        return SDMESH_OK;
    }
    if (!strcmp(name, "dense_grid")) { h->dense_grid = value; return SDMESH_OK; }
    if (!strcmp(name, "dense_aligned")) { h->dense_aligned = value != 0; return SDMESH_OK; }
    if (!strcmp(name, "dense_nt")) { h->dense_nt = value < 0 ? 0 : (value > 2 ? 2 : value); return SDMESH_OK; }    // 2: also in the aligned row kernel
    if (!strcmp(name, "dense_rows")) { h->dense_rows = value; return SDMESH_OK; }
    if (!strcmp(name, "dense_occ")) { h->dense_occ = value; return SDMESH_OK; }
    if (!strcmp(name, "dense_xcd")) { h->dense_xcd = value < 0 ? 0 : (value > 2 ? 2 : value); return SDMESH_OK; }
    if (!strcmp(name, "dense_q1")) { h->dense_q1 = value ? 1 : 0; return SDMESH_OK; }
    if (!strcmp(name, "dense_contig")) { h->dense_contig = value ? 1 : 0; return SDMESH_OK; }
    if (!strcmp(name, "forget_plan")) { h->plan_valid = false; return SDMESH_OK; }
    return fail(h, SDMESH_ERR_INVALID, "unknown option %s", name);
}

int sdmesh_create_voxel_field(struct Point *voxels, struct Point *voxel_size) {
    if (!voxels || !voxel_size) return SDMESH_ERR_INVALID;
    const float SIZE = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    voxel_size->x = voxel_size->y = voxel_size->z = SIZE;
    size_t o = 0;
    for (int x = 0; x < MESH_GENERATION_INIT_FACTOR; x++)
        for (int y = 0; y < MESH_GENERATION_INIT_FACTOR; y++)
            for (int z = 0; z < MESH_GENERATION_INIT_FACTOR; z++, o++) {
                voxels[o].x = (float) x * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
                voxels[o].y = (float) y * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
                voxels[o].z = (float) z * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
            }
    return SDMESH_OK;
}

static int refine_host_common(sdmesh_handle *h, const Point *in, size_t n, Point voxel_size) {
    hipStream_t st = h->stream;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    SDCHK(ensure(h, h->vox[0], sizeof(Point) * n));
    SDCHK(ensure(h, h->mask, n));
    SDCHK(ensure_ctl(h, tiles((uint32_t) n) + 16));
    h->cur = 0;
    h->vox_valid = h->mesh_valid = false;
    HIPCHK(hipMemcpyAsync(h->vox[0].p, in, sizeof(Point) * n, hipMemcpyHostToDevice, st));
    SDCHK(zero_ctl(h, tiles((uint32_t) n) + 16));
    HIPCHK(launch_set_u32(st, h->d_ctr + CTR_LEVEL0, (uint32_t) n));
    HIPCHK(launch_refine_mask(st, h->tgrid, h->scene, h->d_sp, (const Point *) h->vox[0].p, h->d_ctr + CTR_LEVEL0, half_size(voxel_size),
                              (uint8_t *) h->mask.p, h->d_ctr + CTL_LVL0));
    return SDMESH_OK;
}

int sdmesh_refine_voxel_field(sdmesh_handle *h, const struct Point *in, size_t n, struct Point voxel_size,
                              struct Point *out, size_t out_cap, size_t *n_out, struct Point *size_out) {
    if (!h || !n_out || !size_out || (n && (!in || !out))) return SDMESH_ERR_INVALID;
    SDCHK(mesh_scene_ok(h));
    if (n == 0) {                      // src/cuda/mod.rs:137: nothing happens, not even the size halves
        *n_out = 0;
        *size_out = voxel_size;
        return SDMESH_OK;
    }
    if (n > 0x1FFFFFFFu) return fail(h, SDMESH_ERR_INVALID, "too many voxels");
    hipStream_t st = h->stream;
    SDCHK(refine_host_common(h, in, n, voxel_size));
    HIPCHK(launch_sum_tiles(st, h->d_ctr + CTL_LVL0, h->d_ctr + CTR_LEVEL0, h->d_ctr + CTR_SCRATCH));
    SDCHK(read_back(h));
    const uint32_t cnt = h->h_pinned[CTR_SCRATCH];
    if (cnt > out_cap) return fail(h, SDMESH_ERR_CAPACITY, "refine output needs %u points, caller gave %zu", cnt, out_cap);
    SDCHK(ensure(h, h->vox[1], sizeof(Point) * (size_t) std::max<uint32_t>(cnt, 64)));
    HIPCHK(launch_refine_emit(st, h->tgrid, (const Point *) h->vox[0].p, h->d_ctr + CTR_LEVEL0, half_size(voxel_size),
                              (const uint8_t *) h->mask.p, h->d_ctr + CTL_LVL0, (Point *) h->vox[1].p, cnt,
                              h->d_ctr + CTR_LEVEL0 + 1, h->d_status));
    if (cnt) HIPCHK(hipMemcpyAsync(out, h->vox[1].p, sizeof(Point) * (size_t) cnt, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    h->cur = 1;
    h->n_vox = cnt;
    h->voxel_size = half_size(voxel_size);
    h->vox_valid = true;
    *n_out = cnt;
    *size_out = h->voxel_size;
    return SDMESH_OK;
}

int sdmesh_refine_voxel_field_raw(sdmesh_handle *h, const struct Point *in, size_t n, struct Point voxel_size,
                                  struct Point *out8n) {
    if (!h || (n && (!in || !out8n))) return SDMESH_ERR_INVALID;
    SDCHK(mesh_scene_ok(h));
    if (n == 0) return SDMESH_OK;
    if (n > 0x0FFFFFFFu) return fail(h, SDMESH_ERR_INVALID, "too many voxels");
    hipStream_t st = h->stream;
    SDCHK(refine_host_common(h, in, n, voxel_size));
    SDCHK(ensure(h, h->vox[1], sizeof(Point) * 8 * n));
    HIPCHK(launch_refine_expand_raw(st, h->grid, (const Point *) h->vox[0].p, (uint32_t) n, half_size(voxel_size),
                                    (const uint8_t *) h->mask.p, (Point *) h->vox[1].p));
    HIPCHK(hipMemcpyAsync(out8n, h->vox[1].p, sizeof(Point) * 8 * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SDMESH_OK;
}

int sdmesh_voxel_field_to_triangles_raw(sdmesh_handle *h, const struct Point *voxels, size_t n, struct Point voxel_size,
                                        struct Triangle *out5n) {
    if (!h || (n && (!voxels || !out5n))) return SDMESH_ERR_INVALID;
    SDCHK(mesh_scene_ok(h));
    if (n == 0) return SDMESH_OK;
    if (n > 0x0FFFFFFFu) return fail(h, SDMESH_ERR_INVALID, "too many voxels");
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    hipStream_t st = h->stream;
    bool direct = false;
    SDCHK(host_mesh_front(h, voxels, (uint32_t) n, voxel_size, direct));
    SDCHK(ensure(h, h->raw, sizeof(Triangle) * 5ull * n));
    if (!direct)
        HIPCHK(launch_expand_raw(st, h->grid, (const Vertex *) h->uverts.p, (const uint32_t *) h->tri_e.p,
                                 (const uint8_t *) h->flip.p /* weld has not run: apply flips here */, (const uint8_t *) h->cubeidx.p, (const uint32_t *) h->trioff.p, (uint32_t) n, (Triangle *) h->raw.p));
    HIPCHK(hipMemcpyAsync(out5n, h->raw.p, sizeof(Triangle) * 5ull * n, hipMemcpyDeviceToHost, st));
    SDCHK(read_back(h));
    return status_to_error(h);
}

int sdmesh_voxel_field_to_mesh(sdmesh_handle *h, const struct Point *voxels, size_t n, struct Point voxel_size,
                               struct Vertex *verts, size_t vert_cap, size_t *n_verts, struct IndexTriangle *tris,
                               size_t tri_cap, size_t *n_tris) {
    if (!h || !n_verts || !n_tris || (n && !voxels)) return SDMESH_ERR_INVALID;
    SDCHK(mesh_scene_ok(h));
    *n_verts = *n_tris = 0;
    if (n == 0) {                                  // src/cuda/mod.rs:327-344: empty mesh
        h->n_tris = h->n_verts = 0;
        h->res_v = nullptr; h->res_i = nullptr;
        h->mesh_valid = true;
        return SDMESH_OK;
    }
    if (n > 0x0FFFFFFFu) return fail(h, SDMESH_ERR_INVALID, "too many voxels");
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    bool direct = false;
    if (h->weld_wide) h->wide_active = true;
    for (int pass = 0; pass < 2; pass++) {
        SDCHK(host_mesh_front(h, voxels, (uint32_t) n, voxel_size, direct, true));
        if (direct) { SDCHK(weld_raw_device(h, (const Triangle *) h->raw.p, 5u * (uint32_t) n)); SDCHK(read_back(h)); }
        if (!h->h_pinned[CTR_WORDS + ST_WELD_RANGE] || h->wide_active) break;
        h->wide_active = true;               // a vertex outside the packed key range: weld again with [i64; 3]-equivalent keys
    }
    SDCHK(status_to_error(h));
    h->n_unique = h->h_pinned[CTR_NUNIQ];
    h->n_tris = h->h_pinned[CTR_NTRI];
    h->n_verts = h->h_pinned[CTR_NVERT];
    h->mesh_valid = true;
    h->welded = true;
    h->last_direct = direct;
    *n_verts = h->n_verts;
    *n_tris = h->n_tris;
    if (verts || tris) {
        if ((verts && h->n_verts > vert_cap) || (tris && h->n_tris > tri_cap))
            return fail(h, SDMESH_ERR_CAPACITY, "mesh needs %u vertices / %u triangles", h->n_verts, h->n_tris);
        return sdmesh_download_mesh(h, verts, tris);
    }
    return SDMESH_OK;
}

int sdmesh_weld_triangles(sdmesh_handle *h, const struct Triangle *raw, size_t nslots, struct Vertex *verts, size_t vert_cap,
                          size_t *n_verts, struct IndexTriangle *tris, size_t tri_cap, size_t *n_tris) {
    if (!h || !n_verts || !n_tris || (nslots && !raw)) return SDMESH_ERR_INVALID;
    *n_verts = *n_tris = 0;
    if (nslots > 0x0FFFFFFFu) return fail(h, SDMESH_ERR_INVALID, "too many slots");
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    if (nslots) {
        SDCHK(ensure(h, h->staging, sizeof(Triangle) * nslots));
        HIPCHK(hipMemcpyAsync(h->staging.p, raw, sizeof(Triangle) * nslots, hipMemcpyHostToDevice, h->stream));
    }
    SDCHK(sdmesh_weld_device_triangles(h, (const Triangle *) h->staging.p, nslots, nullptr));
    *n_verts = h->n_verts;
    *n_tris = h->n_tris;
    if (verts || tris) {
        if ((verts && h->n_verts > vert_cap) || (tris && h->n_tris > tri_cap))
            return fail(h, SDMESH_ERR_CAPACITY, "mesh needs %u vertices / %u triangles", h->n_verts, h->n_tris);
        return sdmesh_download_mesh(h, verts, tris);
    }
    return SDMESH_OK;
}

static int weld_device_impl(sdmesh_handle *h, const struct Triangle *d_raw, size_t nslots, sdmesh_regen_info *info, bool all_valid);

int sdmesh_weld_device_triangles(sdmesh_handle *h, const struct Triangle *d_raw, size_t nslots, sdmesh_regen_info *info) {
    return weld_device_impl(h, d_raw, nslots, info, false);
}

static int weld_device_impl(sdmesh_handle *h, const struct Triangle *d_raw, size_t nslots, sdmesh_regen_info *info, bool all_valid) {
    if (!h || (nslots && !d_raw)) return SDMESH_ERR_INVALID;
    if (nslots > 0x0FFFFFFFu) return fail(h, SDMESH_ERR_INVALID, "too many slots");
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    h->mesh_valid = false;
    if (h->weld_wide) h->wide_active = true;
    for (int pass = 0; pass < 2; pass++) {
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        SDCHK(zero_status(h));
        HIPCHK(hipMemsetAsync(h->d_ctr + CTR_NUNIQ, 0, 4 * sizeof(uint32_t), h->stream));
        SDCHK(weld_raw_device(h, d_raw, (uint32_t) nslots, all_valid));
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        SDCHK(read_back(h));
        if (!h->h_pinned[CTR_WORDS + ST_WELD_RANGE] || h->wide_active) break;
        h->wide_active = true;               // src/cuda/mod.rs:268-273: [i64; 3] keys have no range
    }
    SDCHK(status_to_error(h));
    if (h->h_pinned[CTR_WORDS + ST_OVERFLOW]) return fail(h, SDMESH_ERR_INTERNAL, "weld overflow (status %u)", h->h_pinned[CTR_WORDS + ST_OVERFLOW]);
    h->n_unique = nslots ? h->h_pinned[CTR_NUNIQ] : 0;
    h->n_tris = nslots ? h->h_pinned[CTR_NTRI] : 0;
    h->n_verts = nslots ? h->h_pinned[CTR_NVERT] : 0;
    h->mesh_valid = true;
    h->welded = true;
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    fill_info(h, info, 0, true, ms);
    return SDMESH_OK;
}

int sdmesh_regen(sdmesh_handle *h, int levels, sdmesh_regen_info *info) {
    return regen_impl(h, levels, SDMESH_PART_NONE, 0, 1, info);
}

int sdmesh_regen_part(sdmesh_handle *h, int levels, int mode, int part, int nparts, sdmesh_regen_info *info) {
    return regen_impl(h, levels, mode, part, nparts, info);
}

int sdmesh_regen_begin(sdmesh_handle *h, int levels, int mode, int part, int nparts) {
    return regen_impl(h, levels, mode, part, nparts, nullptr, 1);
}

int sdmesh_regen_end(sdmesh_handle *h, sdmesh_regen_info *info) {
    return regen_impl(h, 0, 0, 0, 1, info, 2);
}

int sdmesh_device_mesh(sdmesh_handle *h, struct Mesh *mesh, uint32_t *n_tris) {
    if (!h || !mesh) return SDMESH_ERR_INVALID;
    if (!h->mesh_valid || !h->welded) return fail(h, SDMESH_ERR_INVALID, "no welded mesh on the device");
    mesh->vertices = h->res_v ? h->res_v : (Vertex *) h->vertices.p;
    mesh->vertex_count = h->n_verts;
    mesh->triangles = h->res_i ? (IndexTriangle *) h->res_i : (IndexTriangle *) h->indices.p;
    if (n_tris) *n_tris = h->n_tris;
    return SDMESH_OK;
}

int sdmesh_device_voxels(sdmesh_handle *h, struct VoxelField *field) {
    if (!h || !field) return SDMESH_ERR_INVALID;
    if (!h->vox_valid) return fail(h, SDMESH_ERR_INVALID, "no voxel field on the device");
    field->voxel_size = h->voxel_size;
    field->voxels = (Point *) h->vox[h->cur].p;
    field->voxel_count = h->n_vox;
    return SDMESH_OK;
}

int sdmesh_device_triangles(sdmesh_handle *h, struct Triangle *d_out, size_t cap, size_t *n_tris) {
    if (!h || !n_tris) return SDMESH_ERR_INVALID;
    if (!h->mesh_valid || !h->vox_valid) return fail(h, SDMESH_ERR_INVALID, "no regen result on the device");
    *n_tris = h->n_tris;
    if (!d_out) return SDMESH_OK;
    if (cap < h->n_tris) return fail(h, SDMESH_ERR_CAPACITY, "need room for %u triangles", h->n_tris);
    if (h->n_tris == 0) return SDMESH_OK;
    const bool have_compact = h->last_direct;   // the direct path left the compact list in h->compact
    if (have_compact) {
        HIPCHK(hipMemcpyAsync(d_out, h->compact.p, sizeof(Triangle) * (size_t) h->n_tris, hipMemcpyDeviceToDevice, h->stream));
    } else {
        HIPCHK(launch_expand_compact(h->stream, h->grid, (const Vertex *) h->uverts.p, (const uint32_t *) h->tri_e.p,
                                     h->welded ? nullptr : (const uint8_t *) h->flip.p /* the weld applies the flips to tri_e */, h->d_ctr + CTR_NTRI, d_out, (uint32_t) std::min<size_t>(cap, 0xFFFFFFFFu)));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return SDMESH_OK;
}

int sdmesh_column_triangle_counts(sdmesh_handle *h, uint32_t *d_counts) {
    if (!h || !d_counts) return SDMESH_ERR_INVALID;
    if (!h->mesh_valid || !h->vox_valid) return fail(h, SDMESH_ERR_INVALID, "no regen result on the device");
    if (h->last_direct) return fail(h, SDMESH_ERR_INVALID, "column counts need the lattice path");
    const size_t bytes = sizeof(uint32_t) * MESH_GENERATION_INIT_FACTOR * MESH_GENERATION_INIT_FACTOR;
    HIPCHK(hipMemsetAsync(d_counts, 0, bytes, h->stream));
    if (h->n_vox) {
        HIPCHK(launch_set_u32(h->stream, h->d_ctr + CTR_SCRATCH, h->n_vox));
        HIPCHK(launch_column_counts(h->stream, std::min(h->grid, (int) tiles(h->n_vox)), (const Point *) h->vox[h->cur].p,
                                    h->d_ctr + CTR_SCRATCH, (const uint8_t *) h->cubeidx.p, d_counts));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return SDMESH_OK;
}

int sdmesh_download_mesh(sdmesh_handle *h, struct Vertex *verts, struct IndexTriangle *tris) {
    if (!h) return SDMESH_ERR_INVALID;
    if (!h->mesh_valid) return fail(h, SDMESH_ERR_INVALID, "no mesh on the device");
    // (option "download_streams" = 2: the two buffers on two streams.  Measured slower than one after the other: one copy already runs at the link's rate)
    const bool both = verts && h->n_verts && tris && h->n_tris && h->side && h->stream != nullptr && h->download_streams == 2;
    if (both) {
        HIPCHK(hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        HIPCHK(hipMemcpyAsync(tris, h->res_i ? (const void *) h->res_i : h->indices.p, sizeof(IndexTriangle) * (size_t) h->n_tris, hipMemcpyDeviceToHost, h->side));
        HIPCHK(hipMemcpyAsync(verts, h->res_v ? (const void *) h->res_v : h->vertices.p, sizeof(Vertex) * (size_t) h->n_verts, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->side));
        HIPCHK(hipStreamSynchronize(h->stream));
        return SDMESH_OK;
    }
    if (verts && h->n_verts)
        HIPCHK(hipMemcpyAsync(verts, h->res_v ? (const void *) h->res_v : h->vertices.p, sizeof(Vertex) * (size_t) h->n_verts, hipMemcpyDeviceToHost, h->stream));
    if (tris && h->n_tris)
        HIPCHK(hipMemcpyAsync(tris, h->res_i ? (const void *) h->res_i : h->indices.p, sizeof(IndexTriangle) * (size_t) h->n_tris, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SDMESH_OK;
}

int sdmesh_download_voxels(sdmesh_handle *h, struct Point *voxels) {
    if (!h || !voxels) return SDMESH_ERR_INVALID;
    if (!h->vox_valid) return fail(h, SDMESH_ERR_INVALID, "no voxel field on the device");
    if (h->n_vox)
        HIPCHK(hipMemcpyAsync(voxels, h->vox[h->cur].p, sizeof(Point) * (size_t) h->n_vox, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SDMESH_OK;
}

static hipError_t launch_p1(sdmesh_handle *h, float *d_out, uint32_t N, uint32_t x0, uint32_t x1) {
    // planes [x0, x1) of the x-outer volume into d_out[0 ..); the whole volume is (0, N + 1)
    const int grid = h->dense_grid > 0 ? h->dense_grid : 8 * h->grid;
    // measured (round-1 sweep): the write-bound sphere wants whole 16-B aligned cache lines per store (flat
    // quads, 5.5 TB/s vs 3.9 with dword-aligned row stores); the VALU-bound scenes want the leaner row kernel
    bool rows = h->dense_rows == 2 ? (h->scene != SDMESH_SCENE_SPHERE) : (h->dense_rows != 0);
    // the flat kernels store 16-byte quads: a slab written into its place of a shared volume starts at x_begin (N + 1)^2 samples, which is
    // only dword-aligned when N + 1 is odd -> the row kernel there (4-byte aligned vector stores)
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) rows = true;
    if (!rows && h->dense_q1 && N >= 1u) return launch_sample_dense_q1(h->stream, h->scene, h->d_sp, d_out, N, h->dense_nt != 0, x0, x1, h->dense_xcd == 2);
    return rows ? launch_sample_dense_rows(h->stream, h->dense_grid, h->scene, h->d_sp, d_out, N, h->dense_nt != 0, x0, x1, h->dense_aligned != 0, h->dense_nt == 2, h->dense_occ, h->dense_xcd != 0)
                         : launch_sample_dense(h->stream, grid, h->scene, h->d_sp, d_out, N, h->dense_nt != 0, h->dense_contig != 0, x0, x1);
}
static hipError_t launch_p1(sdmesh_handle *h, float *d_out, uint32_t N) { return launch_p1(h, d_out, N, 0u, N + 1u); }

int sdmesh_regen_dense(sdmesh_handle *h, int levels, sdmesh_regen_info *info) {
    if (!h) return SDMESH_ERR_INVALID;
    if (levels < 0 || levels > 5) return fail(h, SDMESH_ERR_INVALID, "dense mode supports levels 0..5 (N <= 1024)");
    SDCHK(mesh_scene_ok(h));
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    hipStream_t st = h->stream;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR, N = F << levels;
    const uint64_t total = (uint64_t) (N + 1) * (N + 1) * (N + 1);
    h->mesh_valid = h->vox_valid = false;
    h->plan_valid = false;
    SDCHK(ensure(h, h->dense_samples, 4ull * total + 64));
    SDCHK(ensure(h, h->dense_bits, 4ull * (total / 32 + 8)));
    SDCHK(ensure(h, h->cellcnt, 4ull * (F * F * F + 16)));
    HIPCHK(hipEventRecord(h->ev0, st));
    SDCHK(zero_ctl(h, 0));
    HIPCHK(hipMemsetAsync(h->dense_bits.p, 0, 4ull * (total / 32 + 8), st));
    HIPCHK(launch_p1(h, (float *) h->dense_samples.p, N));
    HIPCHK(launch_sign_bits(st, h->dense_grid, (const float *) h->dense_samples.p, total, (uint32_t *) h->dense_bits.p));
    HIPCHK(launch_dense_classify(st, 8 * h->grid, false, (const uint32_t *) h->dense_bits.p, (uint32_t) levels,
                                 (uint32_t *) h->cellcnt.p, nullptr, nullptr, 0));
    HIPCHK(launch_set_u32(st, h->d_ctr + CTR_SCRATCH, F * F * F));
    HIPCHK(launch_scan_tiles(st, (uint32_t *) h->cellcnt.p, h->d_ctr + CTR_SCRATCH, 1, h->d_ctr + CTR_LEVEL0 + levels, 0xFFFFFFFFu,
                             h->d_status));
    uint32_t n = 0;
    HIPCHK(hipMemcpyAsync(&n, h->d_ctr + CTR_LEVEL0 + levels, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    SDCHK(ensure(h, h->vox[0], sizeof(Point) * (size_t) std::max<uint32_t>(n, 64)));
    SDCHK(layout_mesh(h, n, 0));
    h->cur = 0;
    HIPCHK(launch_dense_classify(st, 8 * h->grid, true, (const uint32_t *) h->dense_bits.p, (uint32_t) levels,
                                 (uint32_t *) h->cellcnt.p, (Point *) h->vox[0].p, (uint8_t *) h->cubeidx.p, n));
    Point vs;
    vs.x = vs.y = vs.z = (MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR) / (float) (1u << levels);
    for (int attempt = 0; attempt < 2; attempt++) {
        SDCHK(mesh_front(h, (const Point *) h->vox[0].p, h->d_ctr + CTR_LEVEL0 + levels, vs, true, false, true));
        SDCHK(mesh_weld(h, (const Vertex *) h->uverts.p, tiles(h->cap_vox)));
        HIPCHK(hipEventRecord(h->ev1, st));
        SDCHK(read_back(h));
        if (!h->h_pinned[CTR_WORDS + ST_OVERFLOW]) break;
        if (attempt == 1) return fail(h, SDMESH_ERR_INTERNAL, "capacity overflow persists");
        SDCHK(layout_mesh(h, n, h->h_pinned[CTR_NUNIQ] + 1024));
        SDCHK(zero_status(h));
    }
    SDCHK(status_to_error(h));
    if (h->h_pinned[CTR_WORDS + ST_NOT_LATTICE] || h->h_pinned[CTR_WORDS + ST_TILE_OVERFLOW])
        return fail(h, SDMESH_ERR_INTERNAL, "dense mode produced a non-lattice field");
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->n_vox = n;
    h->n_unique = h->h_pinned[CTR_NUNIQ];
    h->n_tris = h->h_pinned[CTR_NTRI];
    h->n_verts = h->h_pinned[CTR_NVERT];
    h->voxel_size = vs;
    h->mesh_valid = h->vox_valid = true;
    h->last_direct = false;
    h->welded = true;
    fill_info(h, info, levels, false, ms);
    if (info) { for (int i = 0; i < levels; i++) info->voxel_count[i] = 0; info->voxel_count[levels] = n; }
    return SDMESH_OK;
}

int sdmesh_sample_dense(sdmesh_handle *h, uint32_t N, float *d_out, float *kernel_ms) {
    if (!h || !d_out || N == 0 || N > 4096) return SDMESH_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(launch_p1(h, d_out, N));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, h->ev0, h->ev1));
    return SDMESH_OK;
}

int sdmesh_sample_dense_slab(sdmesh_handle *h, uint32_t N, uint32_t x_begin, uint32_t x_end, float *d_out, int iters, float *avg_ms) {
    if (!h || N == 0 || N > 4096 || iters < 1) return SDMESH_ERR_INVALID;
    if (x_begin > x_end || x_end > N + 1u) return fail(h, SDMESH_ERR_INVALID, "x planes [%u, %u) outside the lattice [0, %u)", x_begin, x_end, N + 1u);
    if (x_end > x_begin && !d_out) return SDMESH_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    const bool timed = avg_ms != nullptr;
    if (timed) HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < iters; i++) HIPCHK(launch_p1(h, d_out, N, x_begin, x_end));
    if (timed) {
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        *avg_ms = ms / (float) iters;
    }
    return SDMESH_OK;
}

int sdmesh_time_kernel(sdmesh_handle *h, int which, uint32_t N, float *d_buf, int iters, float *avg_ms) {
    if (!h || !avg_ms || iters < 1 || which < 0 || which > 2 || !d_buf || N == 0 || N > 4096) return SDMESH_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    if (which == 1) {     // P2a: sign bits of the dense lattice in d_buf (fills it first)
        const uint64_t total = (uint64_t) (N + 1) * (N + 1) * (N + 1);
        SDCHK(ensure(h, h->dense_bits, 4ull * (total / 32 + 8)));
        HIPCHK(launch_p1(h, d_buf, N));
        HIPCHK(launch_sign_bits(h->stream, h->dense_grid, d_buf, total, (uint32_t *) h->dense_bits.p));
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        for (int i = 0; i < iters; i++) HIPCHK(launch_sign_bits(h->stream, h->dense_grid, d_buf, total, (uint32_t *) h->dense_bits.p));
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms1 = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms1, h->ev0, h->ev1));
        *avg_ms = ms1 / (float) iters;
        return SDMESH_OK;
    }
    if (which == 2) {     // write ceiling: P1's store stream without the arithmetic
        const uint64_t total = (uint64_t) (N + 1) * (N + 1) * (N + 1);
        const int grid = h->dense_q1 ? 0 : (h->dense_grid > 0 ? h->dense_grid : 8 * h->grid);       // 0 = one quad per thread
        HIPCHK(launch_stream_fill(h->stream, grid, d_buf, total, 1.0f, h->dense_xcd == 2));
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        for (int i = 0; i < iters; i++) HIPCHK(launch_stream_fill(h->stream, grid, d_buf, total, 1.0f, h->dense_xcd == 2));
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms2 = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms2, h->ev0, h->ev1));
        *avg_ms = ms2 / (float) iters;
        return SDMESH_OK;
    }
    HIPCHK(launch_p1(h, d_buf, N));   // warm-up
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < iters; i++) HIPCHK(launch_p1(h, d_buf, N));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_ms = ms / (float) iters;
    return SDMESH_OK;
}


int sdmesh_render_device(sdmesh_handle *h, const GlobalsBuffer *globals, const CameraBuffer *camera, Rgba *d_target, float *kernel_ms) {
    if (!h || !globals || !camera || !d_target) return SDMESH_ERR_INVALID;
    const uint32_t w = globals->render_texture_size[0], ht = globals->render_texture_size[1];
    if (w == 0 || ht == 0 || w > 16384 || ht > 16384) return fail(h, SDMESH_ERR_INVALID, "render size %u x %u out of range", w, ht);
    HIPCHK(hipSetDevice(h->device));
    SDCHK(sync_scene(h));
    float cam13[12];
    for (int i = 0; i < 3; i++) { cam13[i] = camera->position[i]; cam13[3 + i] = camera->forward[i]; cam13[6 + i] = camera->up[i]; cam13[9 + i] = camera->right[i]; }
    const float fov_fac = tanf(camera->fov / 2);                 // common.cu:86, once per frame instead of once per pixel
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(launch_render(h->stream, h->scene, h->d_sp, cam13, fov_fac, globals->render_screen_size[0], globals->render_screen_size[1],
                         (float) globals->render_texture_size[0], (float) globals->render_texture_size[1], w, ht, d_target));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if (kernel_ms) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipEventElapsedTime(kernel_ms, h->ev0, h->ev1));
    }
    return SDMESH_OK;
}

int sdmesh_render(sdmesh_handle *h, const GlobalsBuffer *globals, const CameraBuffer *camera, Rgba *target) {
    if (!h || !globals || !camera || !target) return SDMESH_ERR_INVALID;
    const size_t bytes = 4ull * globals->render_texture_size[0] * globals->render_texture_size[1];
    HIPCHK(hipSetDevice(h->device));
    SDCHK(ensure(h, h->render_tex, std::max<size_t>(bytes, 256)));          // grow-only, like render_texture_buffer (mod.rs:364-366)
    SDCHK(sdmesh_render_device(h, globals, camera, (Rgba *) h->render_tex.p, nullptr));
    HIPCHK(hipMemcpyAsync(target, h->render_tex.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SDMESH_OK;
}

}  // extern "C"

extern "C" int sdmesh_set_slab_bounds(sdmesh_handle *h, int mode, int nparts, const uint32_t *bounds) {
    if (!h) return SDMESH_ERR_INVALID;
    if (!bounds) { h->slab_nparts = 0; h->plan_valid = false; drop_graph(h); return SDMESH_OK; }
    if ((mode != SDMESH_PART_ZSLAB && mode != SDMESH_PART_XSLAB) || nparts < 1 || nparts > 32) return fail(h, SDMESH_ERR_INVALID, "bad slab table");
    if (bounds[0] != 0 || bounds[nparts] != (uint32_t) MESH_GENERATION_INIT_FACTOR) return fail(h, SDMESH_ERR_INVALID, "slab bounds must run from 0 to %d", MESH_GENERATION_INIT_FACTOR);
    for (int i = 0; i < nparts; i++) {
        if (bounds[i + 1] < bounds[i]) return fail(h, SDMESH_ERR_INVALID, "slab bounds must not decrease");
        // the distributed weld unifies a rank's lower face with rank r - 1's upper face only: an empty x-slab would leave the
        // duplicates between its two neighbours un-unified (wrong vertex count and indices, silently)
        if (mode == SDMESH_PART_XSLAB && bounds[i + 1] == bounds[i]) return fail(h, SDMESH_ERR_INVALID, "x-slab %d is empty: every x-slab needs at least one level-0 layer", i);
    }
    h->slab_mode = mode; h->slab_nparts = nparts;
    memcpy(h->slab_bounds, bounds, sizeof(uint32_t) * (size_t) (nparts + 1));
    h->plan_valid = false;
    drop_graph(h);
    return SDMESH_OK;
}

extern "C" int sdmesh_balance_slabs(sdmesh_handle *h, int mode, int nparts, int probe_levels, uint32_t *bounds_out) {
    if (!h) return SDMESH_ERR_INVALID;
    if ((mode != SDMESH_PART_ZSLAB && mode != SDMESH_PART_XSLAB) || nparts < 1 || nparts > 32 || probe_levels < 0 || probe_levels > 4)
        return fail(h, SDMESH_ERR_INVALID, "bad arguments");
    const int F = MESH_GENERATION_INIT_FACTOR;
    // the surface content of every level-0 layer along the slab axis, from an unpartitioned coarse regen of the scene
    const int saved_skip = h->skip_weld;
    h->skip_weld = 1;
    sdmesh_regen_info info;
    const int rc = regen_impl(h, probe_levels, SDMESH_PART_NONE, 0, 1, &info);
    h->skip_weld = saved_skip;
    SDCHK(rc);
    std::vector<Point> vox(std::max<uint32_t>(h->n_vox, 1));
    SDCHK(sdmesh_download_voxels(h, vox.data()));
    double w[32] = {0};
    const float cell = MESH_GENERATION_BB_SIZE / (float) F;
    for (uint32_t i = 0; i < h->n_vox; i++) {
        const float c = mode == SDMESH_PART_XSLAB ? vox[i].x : vox[i].z;
        int l = (int) floorf((c + MESH_GENERATION_BB_SIZE / 2.0f) / cell);
        l = l < 0 ? 0 : (l >= F ? F - 1 : l);
        w[l] += 1.0;
    }
    double total = 0;
    for (int l = 0; l < F; l++) total += w[l];
    for (int l = 0; l < F; l++) w[l] += 0.002 * total / F + 1e-9;          // an empty layer still costs its level-0 cells: never free
    // contiguous partition of the 32 layers into nparts slabs minimising the heaviest slab (dynamic programme, 32 x 32 x nparts)
    double pre[33]; pre[0] = 0;
    for (int l = 0; l < F; l++) pre[l + 1] = pre[l] + w[l];
    double best[33][33]; int cut[33][33];                 // (locals: two handles may balance from two threads)
    for (int e = 0; e <= F; e++) { best[1][e] = e >= 1 ? pre[e] : 1e300; cut[1][e] = 0; }
    for (int k = 2; k <= nparts; k++)
        for (int e = 0; e <= F; e++) {
            best[k][e] = 1e300; cut[k][e] = k - 1;
            for (int m = k - 1; m < e; m++) {                         // every slab at least one layer thick
                const double v = std::max(best[k - 1][m], pre[e] - pre[m]);
                if (v < best[k][e]) { best[k][e] = v; cut[k][e] = m; }
            }
        }
    uint32_t b[33];
    b[nparts] = (uint32_t) F;
    for (int k = nparts, e = F; k >= 1; k--) { e = cut[k][e]; b[k - 1] = (uint32_t) e; }
    b[0] = 0;
    SDCHK(sdmesh_set_slab_bounds(h, mode, nparts, b));
    if (bounds_out) memcpy(bounds_out, b, sizeof(uint32_t) * (size_t) (nparts + 1));
    return SDMESH_OK;
}

extern "C" int sdmesh_selfcheck_arithmetic(sdmesh_handle *h, uint64_t *mismatches, int n) {
    if (!h || !mismatches || n < 1) return SDMESH_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(ensure(h, h->staging, 8ull * SDMESH_SELFCHECKS));
    HIPCHK(hipMemsetAsync(h->staging.p, 0, 8ull * SDMESH_SELFCHECKS, h->stream));
    for (int w = 0; w < SDMESH_SELFCHECKS && w < n; w++) HIPCHK(launch_selfcheck(h->stream, w, (unsigned long long *) h->staging.p + w));
    uint64_t host[SDMESH_SELFCHECKS] = {0};
    HIPCHK(hipMemcpyAsync(host, h->staging.p, sizeof host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int w = 0; w < n; w++) mismatches[w] = w < SDMESH_SELFCHECKS ? host[w] : 0;
    return SDMESH_OK;
}

extern "C" int sdmesh_profile_rounds(sdmesh_handle *h, uint64_t *vertex_rounds) {
    if (!h || !vertex_rounds) return SDMESH_ERR_INVALID;
    if (!h->prof_rounds.p) return fail(h, SDMESH_ERR_INVALID, "no profiled regen yet (option \"profile\")");
    uint32_t host[256 * 16];
    HIPCHK(hipMemcpy(host, h->prof_rounds.p, sizeof host, hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (int i = 0; i < 256; i++) sum += host[i * 16];
    *vertex_rounds = sum;
    return SDMESH_OK;
}

extern "C" int sdmesh_project_diag(sdmesh_handle *h, uint64_t *out, size_t cap_waves, size_t *n_waves) {
    if (!h || !n_waves) return SDMESH_ERR_INVALID;
    *n_waves = h->proj_diag.p ? h->diag_waves : 0;
    if (!out) return SDMESH_OK;
    if (cap_waves < *n_waves) return fail(h, SDMESH_ERR_CAPACITY, "need room for %zu waves", *n_waves);
    if (*n_waves) HIPCHK(hipMemcpy(out, h->proj_diag.p, 32ull * *n_waves, hipMemcpyDeviceToHost));
    return SDMESH_OK;
}

extern "C" int sdmesh_profile_stages(sdmesh_handle *h, char *names, size_t names_cap, float *ms, int cap, int *n_out) {
    if (!h || !n_out) return SDMESH_ERR_INVALID;
    const int n = h->prof_ms.size() > 1 ? (int) h->prof_ms.size() - 1 : 0;      // stage i = gap between markers i and i + 1
    *n_out = n;
    if (!names && !ms) return SDMESH_OK;
    if (cap < n) return fail(h, SDMESH_ERR_CAPACITY, "need room for %d stages", n);
    std::string joined;
    for (int i = 0; i < n; i++) {
        if (ms) ms[i] = h->prof_ms[(size_t) i + 1];
        if (i) joined += ',';
        joined += h->prof_names[(size_t) i + 1];
    }
    if (names) {
        if (joined.size() + 1 > names_cap) return fail(h, SDMESH_ERR_CAPACITY, "need %zu bytes for the stage names", joined.size() + 1);
        memcpy(names, joined.c_str(), joined.size() + 1);
    }
    return SDMESH_OK;
}

// ================================================================================================
// Multi-GPU: ONE mesh over the ranks of an RCCL communicator
// ================================================================================================
namespace {

RcclApi g_rccl;

const char *load_rccl() {
    if (g_rccl.lib) return nullptr;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy already in the process (PyTorch-ROCm)
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return "librccl.so.1 not found";
#define SDM_BIND(field, sym)                                                     \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, sym));    \
    if (!g_rccl.field) return "librccl.so.1 lacks " sym
    SDM_BIND(GetUniqueId, "ncclGetUniqueId");
    SDM_BIND(CommInitRank, "ncclCommInitRank");
    SDM_BIND(CommDestroy, "ncclCommDestroy");
    SDM_BIND(AllGather, "ncclAllGather");
    SDM_BIND(AllReduce, "ncclAllReduce");
    SDM_BIND(Send, "ncclSend");
    SDM_BIND(Recv, "ncclRecv");
    SDM_BIND(GroupStart, "ncclGroupStart");
    SDM_BIND(GroupEnd, "ncclGroupEnd");
    SDM_BIND(GetErrorString, "ncclGetErrorString");
#undef SDM_BIND
    g_rccl.lib = lib;
    return nullptr;
}

#define NCCLCHK(expr)                                                                                               \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess)                                                                                      \
            return fail(h, SDMESH_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// The handle's collectives: RCCL on its communicator, or the caller's transport.  Same semantics either way (ordered on `st` after the work
// already enqueued there and in front of what follows; all-gather / all-reduce may be in place; send / recv between GroupStart and
// GroupEnd progress together).  Counts are in elements of `type` (4- or 8-byte ones only).
inline size_t tp_bytes(size_t count, ncclDataType_t type) { return count * (type == ncclUint64 ? 8u : 4u); }
int tp_fail(sdmesh_handle *h, const char *what, int rc) { return fail(h, SDMESH_ERR_RCCL, "the caller's transport failed in %s (returned %d)", what, rc); }
int tp_all_gather(sdmesh_handle *h, const void *send, void *recv, size_t count, ncclDataType_t type, hipStream_t st) {
    if (h->comm_ext) { const int rc = h->ext.all_gather(h->ext.user, send, recv, tp_bytes(count, type), st); return rc ? tp_fail(h, "all_gather", rc) : SDMESH_OK; }
    NCCLCHK(g_rccl.AllGather(send, recv, count, type, h->comm, st));
    return SDMESH_OK;
}
int tp_all_reduce_u32(sdmesh_handle *h, const void *send, void *recv, size_t count, bool max_not_sum, hipStream_t st) {
    if (h->comm_ext) { const int rc = h->ext.all_reduce_u32(h->ext.user, send, recv, count, max_not_sum ? SDMESH_REDUCE_MAX : SDMESH_REDUCE_SUM, st); return rc ? tp_fail(h, "all_reduce_u32", rc) : SDMESH_OK; }
    NCCLCHK(g_rccl.AllReduce(send, recv, count, ncclUint32, max_not_sum ? ncclMax : ncclSum, h->comm, st));
    return SDMESH_OK;
}
int tp_send(sdmesh_handle *h, const void *buf, size_t count, ncclDataType_t type, int peer, hipStream_t st) {
    if (h->comm_ext) { const int rc = h->ext.send(h->ext.user, buf, tp_bytes(count, type), peer, st); return rc ? tp_fail(h, "send", rc) : SDMESH_OK; }
    NCCLCHK(g_rccl.Send(buf, count, type, peer, h->comm, st));
    return SDMESH_OK;
}
int tp_recv(sdmesh_handle *h, void *buf, size_t count, ncclDataType_t type, int peer, hipStream_t st) {
    if (h->comm_ext) { const int rc = h->ext.recv(h->ext.user, buf, tp_bytes(count, type), peer, st); return rc ? tp_fail(h, "recv", rc) : SDMESH_OK; }
    NCCLCHK(g_rccl.Recv(buf, count, type, peer, h->comm, st));
    return SDMESH_OK;
}
int tp_group(sdmesh_handle *h, bool start) {
    if (h->comm_ext) { const int rc = start ? h->ext.group_start(h->ext.user) : h->ext.group_end(h->ext.user); return rc ? tp_fail(h, start ? "group_start" : "group_end", rc) : SDMESH_OK; }
    if (start) NCCLCHK(g_rccl.GroupStart()); else NCCLCHK(g_rccl.GroupEnd());
    return SDMESH_OK;
}

int ensure_grow(sdmesh_handle *h, DevBuf &b, size_t bytes) {      // like ensure(), with 25 % head-room when it has to grow
    if (b.p && b.cap >= bytes) return SDMESH_OK;
    return ensure(h, b, bytes + bytes / 4 + 4096);
}

int ensure_shard_host(sdmesh_handle *h, int nranks) {
    const size_t words = (size_t) nranks * SHARD_ROW;
    if (h->h_counts_words >= words) return SDMESH_OK;
    if (h->h_counts) (void) hipHostFree(h->h_counts);
    if (h->h_lists) (void) hipHostFree(h->h_lists);
    h->h_counts = nullptr; h->h_lists = nullptr; h->h_counts_words = 0;
    HIPCHK(hipHostMalloc((void **) &h->h_counts, 4 * words, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **) &h->h_lists, sizeof(void *) * (size_t) nranks, hipHostMallocDefault));
    h->h_counts_words = words;
    return SDMESH_OK;
}

// merged list of `nranks` device lists (pointers in h->h_lists, pinned) by the per-rank column counts at d_counts (`cstride`
// words per rank)
int merge_lists(sdmesh_handle *h, const uint32_t *d_counts, uint32_t cstride, int nranks, Triangle *d_out, size_t cap) {
    SDCHK(ensure(h, h->shard_lists, sizeof(void *) * (size_t) nranks));
    SDCHK(ensure(h, h->shard_plan, 8ull * (size_t) nranks * 1024));
    HIPCHK(hipMemcpyAsync(h->shard_lists.p, h->h_lists, sizeof(void *) * (size_t) nranks, hipMemcpyHostToDevice, h->stream));
    HIPCHK(launch_merge_by_column(h->stream, (const Triangle *const *) h->shard_lists.p, d_counts, cstride, nranks, (uint32_t *) h->shard_plan.p,
                                  d_out, (uint32_t) std::min<size_t>(cap, 0xFFFFFFFFu), h->d_ctr + CTR_SCRATCH, h->d_status));
    return SDMESH_OK;
}

}  // namespace

extern "C" {

int sdmesh_comm_unique_id(sdmesh_unique_id *out) {
    if (!out) return SDMESH_ERR_INVALID;
    static_assert(sizeof(sdmesh_unique_id) == sizeof(ncclUniqueId), "sdmesh_unique_id must be an ncclUniqueId");
    if (load_rccl()) return SDMESH_ERR_RCCL;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return SDMESH_ERR_RCCL;
    memcpy(out->bytes, &id, sizeof id);
    return SDMESH_OK;
}

int sdmesh_comm_init(sdmesh_handle *h, const sdmesh_unique_id *id, int nranks, int rank) {
    if (!h || !id || nranks < 1 || nranks > 32 || rank < 0 || rank >= nranks) return SDMESH_ERR_INVALID;
    if (h->has_comm()) return fail(h, SDMESH_ERR_INVALID, "the handle already has a communicator");
    if (const char *why = load_rccl()) return fail(h, SDMESH_ERR_RCCL, "%s", why);
    HIPCHK(hipSetDevice(h->device));
    ncclUniqueId nid;
    memcpy(&nid, id->bytes, sizeof nid);
    NCCLCHK(g_rccl.CommInitRank(&h->comm, nranks, nid, rank));
    h->comm_rank = rank;
    h->comm_nranks = nranks;
    return SDMESH_OK;
}

int sdmesh_comm_init_transport(sdmesh_handle *h, int nranks, int rank, const sdmesh_transport *t) {
    if (!h || !t || nranks < 1 || nranks > 32 || rank < 0 || rank >= nranks) return SDMESH_ERR_INVALID;
    if (h->has_comm()) return fail(h, SDMESH_ERR_INVALID, "the handle already has a communicator");
    if (!t->all_gather || !t->all_reduce_u32 || !t->send || !t->recv || !t->group_start || !t->group_end)
        return fail(h, SDMESH_ERR_INVALID, "sdmesh_comm_init_transport: every entry of the transport must be set");
    h->ext = *t;
    h->comm_ext = true;
    h->comm_rank = rank;
    h->comm_nranks = nranks;
    return SDMESH_OK;
}

int sdmesh_comm_destroy(sdmesh_handle *h) {
    if (!h) return SDMESH_ERR_INVALID;
    if (h->comm_ext) {
        (void) hipStreamSynchronize(h->stream);
        h->comm_ext = false;
        h->ext = sdmesh_transport{};
    }
    if (h->comm) {
        (void) hipStreamSynchronize(h->stream);
        (void) g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->comm_rank = 0;
    h->comm_nranks = 1;
    return SDMESH_OK;
}

int sdmesh_comm_ranks(sdmesh_handle *h, int *ranks_seen) {
    if (!h || !ranks_seen) return SDMESH_ERR_INVALID;
    if (!h->has_comm()) return fail(h, SDMESH_ERR_INVALID, "no communicator (sdmesh_comm_init)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(launch_set_u32(h->stream, h->d_ctr + CTR_SCRATCH, 1u));
    SDCHK(tp_all_reduce_u32(h, h->d_ctr + CTR_SCRATCH, h->d_ctr + CTR_SCRATCH, 1, false, h->stream));
    uint32_t v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->d_ctr + CTR_SCRATCH, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *ranks_seen = (int) v;
    return SDMESH_OK;
}

int sdmesh_merge_by_column(sdmesh_handle *h, const struct Triangle *const *d_lists, const uint32_t *d_counts, int nranks,
                           struct Triangle *d_out, size_t cap, size_t *n_out) {
    if (!h || !d_lists || !d_counts || !n_out || nranks < 1 || nranks > 32 || (cap && !d_out)) return SDMESH_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(ensure_shard_host(h, nranks));
    for (int r = 0; r < nranks; r++) h->h_lists[r] = (void *) d_lists[r];
    SDCHK(zero_status(h));
    SDCHK(merge_lists(h, d_counts, 1024u, nranks, d_out, cap));
    SDCHK(read_back(h));
    *n_out = h->h_pinned[CTR_SCRATCH];
    if (h->h_pinned[CTR_WORDS + ST_OVERFLOW]) return fail(h, SDMESH_ERR_CAPACITY, "merged list needs room for %u triangles", h->h_pinned[CTR_SCRATCH]);
    return SDMESH_OK;
}

}  // extern "C"

namespace {

// Agreement before payload (both one-mesh entry points).  A rank that fails locally -- its slab regen, an allocation, a
// device-side invariant -- must not simply return: the others would wait for it in the next collective forever.  So every
// rank ALWAYS enters the all-gather that carries the sizes, with a status word in its record; after it every rank holds
// every rank's status and sizes, evaluates the same checks on the same numbers, and all of them leave the call at the same
// point: the failing rank with its own error, the others with SDMESH_ERR_PEER.  Buffers whose size is only known after
// that exchange are grown behind a second, one-word agreement, which runs only when some rank's published capacity is too
// small (every rank can tell from the records) -- never in steady state.  After an agreed failure the communicator is
// still usable; after an SDMESH_ERR_RCCL (a collective itself failed) it is not.
int peer_failure(sdmesh_handle *h, int R, int me, const uint32_t *rec, uint32_t stride, uint32_t status_word, int local_rc) {
    if (local_rc != SDMESH_OK) return local_rc;                              // (h->err already says why)
    for (int r = 0; r < R; r++)
        if (r != me && rec[(size_t) r * stride + status_word])
            return fail(h, SDMESH_ERR_PEER, "rank %d failed in this collective call (status 0x%x); every rank returns an error", r,
                        rec[(size_t) r * stride + status_word]);
    return SDMESH_OK;
}

// one-word agreement: max over the ranks of `flag` (0 = go)
int agree_flag(sdmesh_handle *h, uint32_t flag, uint32_t *out) {
    hipStream_t st = h->stream;
    HIPCHK(launch_set_u32(st, h->d_ctr + CTR_SCRATCH, flag));
    SDCHK(tp_all_reduce_u32(h, h->d_ctr + CTR_SCRATCH, h->d_ctr + CTR_SCRATCH, 1, true, st));
    HIPCHK(hipMemcpyAsync(h->h_pinned + CTR_SCRATCH, h->d_ctr + CTR_SCRATCH, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *out = h->h_pinned[CTR_SCRATCH];
    return SDMESH_OK;
}

uint32_t cap_items(const DevBuf &b, size_t item) { return (uint32_t) std::min<size_t>(b.p ? b.cap / item : 0, 0xFFFFFFFFu); }

}  // namespace

namespace {

// ---- z-slab INDEXED gather behind sdmesh_sharded_regen (kernels: csrc/zslab.hip) --------------------------------------------------
// Two routes through the same kernels:
//   synchronous  (first call, after any change, after a planned step that did not fit): slab regen with its own read-back, all-gather of
//                the rows, read-back of the rows (message sizes must be host values), exact-size messages into slots, merge, read-back:
//                three host synchronisations.  It ends by committing the PLAN: per-rank slot sizes with head-room, identical on every
//                rank because they all derive it from the same gathered records.
//   planned      (steady state): the slab's chain enqueued from its capacity plan, its two count kernels, the all-gather, plan-sized
//                messages into plan-sized slots, the merge -- all enqueued blind, ONE host synchronisation at the end.  The real counts
//                ride in the gathered records; the kernels read them on the device, the host verifies them afterwards.  A step whose counts
//                did not fit (or in which any rank reported a failure) is repeated on the synchronous route -- by every rank, because every
//                rank sees the same records.
constexpr int ZS_REDO = -1000;                     // (internal: the planned step asks for the synchronous route)
uint32_t zs_padded(const sdmesh_handle *h, uint32_t n) { return h->shard_slack ? n + n / 8u + 2048u : n + 1u; }   // (option "shard_slack" = 0: tests of the repeat)

uint32_t zs_slab_sig(const sdmesh_handle *h, int R) {
    uint32_t s = 2166136261u;
    const bool own = h->slab_nparts == R && h->slab_mode == SDMESH_PART_ZSLAB;
    for (int i = 0; own && i <= R; i++) s = (s ^ h->slab_bounds[i]) * 16777619u;
    return s ^ (own ? 0x9e3779b9u : 0u);
}

uint32_t *zs_counts(sdmesh_handle *h, int which) { return (uint32_t *) (which ? h->shard_counts2.p : h->shard_counts.p); }

// the fixed-size scratch of the exchange: two count buffers (rows of every rank; they alternate, the count areas of the idle one are
// cleared by the running step), the merge's control block, pinned landing zones
int zs_scratch(sdmesh_handle *h, int R) {
    const bool fresh = !h->shard_counts.p || !h->shard_counts2.p || h->shard_counts.cap < 4ull * (size_t) R * SHARD_ROW || h->shard_counts2.cap < 4ull * (size_t) R * SHARD_ROW;
    SDCHK(ensure(h, h->shard_counts, 4ull * (size_t) R * SHARD_ROW));
    SDCHK(ensure(h, h->shard_counts2, 4ull * (size_t) R * SHARD_ROW));
    if (fresh) h->zsp.valid = false;               // (only ever on a handle's first sharded call or after the communicator grew: the same call on every rank)
    SDCHK(ensure(h, h->zs_ctl, 4ull * ZC_WORDS + 8192));
    SDCHK(ensure(h, h->zs_colscratch, 8192));
    SDCHK(ensure(h, h->zs_ranks, sizeof(ZsRanks)));
    SDCHK(ensure(h, h->zs_tables, 4ull * 6 * (size_t) R * 1024));   // tplan [2 nseg] | vplan [2 nseg] | vstart [nseg] | newcnt [nseg]
    SDCHK(ensure_shard_host(h, R));
    if (!h->h_zs) HIPCHK(hipHostMalloc((void **) &h->h_zs, sizeof(ZsRanks), hipHostMallocDefault));
    if (!h->h_zs_pub) HIPCHK(hipHostMalloc((void **) &h->h_zs_pub, 4ull * (ZC_WORDS + 32 * 8), hipHostMallocDefault));
    return SDMESH_OK;
}

// everything the root's merge touches, for slots of vslots vertices / tslots triangles in all
int zs_root_ensure(sdmesh_handle *h, int R, uint64_t vslots, uint64_t tslots) {
    const size_t v = (size_t) std::max<uint64_t>(vslots, 64), t = (size_t) std::max<uint64_t>(tslots, 64);
    SDCHK(ensure_grow(h, h->zs_verts, sizeof(Vertex) * v));
    SDCHK(ensure_grow(h, h->zs_idx, sizeof(IndexTriangle) * t));
    SDCHK(ensure_grow(h, h->zs_pervert, 4ull * 3 * v + 4ull * (v / 256 + 2)));    // per slot vertex: segof | redirect | map; then the drop counts per 256 of them
    SDCHK(ensure_grow(h, h->zs_hints, 4ull * (t / 256 + (size_t) R + 2)));     // one word per 256 slot triangles (every rank's slot rounds up)
    SDCHK(ensure_grow(h, h->zs_vout, sizeof(Vertex) * v));
    SDCHK(ensure_grow(h, h->zs_iout, sizeof(IndexTriangle) * t));
    return SDMESH_OK;
}
uint32_t zs_root_capacity_v(const sdmesh_handle *h) {               // what this rank could take as a root without allocating
    const size_t a = cap_items(h->zs_verts, sizeof(Vertex)), b = h->zs_pervert.cap > 64 ? (h->zs_pervert.cap - 8) * 256 / (12 * 256 + 4) : 0, c = cap_items(h->zs_vout, sizeof(Vertex));
    return (uint32_t) std::min(a, std::min(b, c));
}
uint32_t zs_root_capacity_t(const sdmesh_handle *h) {
    const uint32_t by_hints = h->zs_hints.cap > 4ull * 40 ? (uint32_t) std::min<size_t>((h->zs_hints.cap / 4 - 36) * 256, 0xFFFFFFFFu) : 0u;
    return std::min(by_hints, std::min(cap_items(h->zs_idx, sizeof(IndexTriangle)), cap_items(h->zs_iout, sizeof(IndexTriangle))));
}

void zs_slot_table(const sdmesh_handle *h, int R, const uint32_t *cap_v, const uint32_t *cap_t, int levels, ZsRanks &zr) {
    memset(&zr, 0, sizeof zr);
    zr.nranks = (uint32_t) R;
    // entries per workgroup of the two streaming merge kernels: the big block from ~1 M triangles on (option "zs_block": 256 / 1024 / 0 = by size)
    uint64_t tsum = 0;
    for (int r = 0; r < R; r++) tsum += cap_t[r];
    zr.blk = h->zs_block ? (uint32_t) h->zs_block : (tsum >= 1000000ull ? 1024u : 256u);
    uint32_t vb = 0, tb = 0;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR;
    for (int r = 0; r < R; r++) {
        zr.vbase[r] = vb; zr.tbase[r] = tb;
        vb += cap_v[r]; tb += cap_t[r];
        zr.tblk[r + 1] = zr.tblk[r] + (cap_t[r] + zr.blk - 1u) / zr.blk;
        uint32_t lo = F * (uint32_t) r / (uint32_t) R, hi = F * (uint32_t) (r + 1) / (uint32_t) R;
        if (h->slab_nparts == R && h->slab_mode == SDMESH_PART_ZSLAB) { lo = h->slab_bounds[r]; hi = h->slab_bounds[r + 1]; }
        zr.zlo[r] = -MESH_GENERATION_BB_SIZE / 2.0f + MESH_GENERATION_BB_SIZE * (float) lo / (float) F;
        zr.zhi[r] = -MESH_GENERATION_BB_SIZE / 2.0f + MESH_GENERATION_BB_SIZE * (float) hi / (float) F;
    }
    zr.vbase[R] = vb; zr.tbase[R] = tb;
    zr.zbot = -MESH_GENERATION_BB_SIZE / 2.0f; zr.ztop = MESH_GENERATION_BB_SIZE / 2.0f;
    // a projected vertex stays within about a voxel of its lattice edge: 2.5 voxels of the finest level is a safe band
    zr.band = 2.5f * ((MESH_GENERATION_BB_SIZE / (float) F) / (float) (1u << levels));
}

// enqueue the merge of the slots described by zr (root).  Nothing here waits; the band hash is sized from the last merge's band.
int zs_merge_enqueue(sdmesh_handle *h, const ZsRanks &zr, const uint32_t *d_counts, bool whole_band, uint32_t seq) {
    hipStream_t st = h->stream;
    const uint32_t R = zr.nranks, V = zr.vbase[R], T = zr.tbase[R], nseg = R * 1024u;
    if (!h->zs_uploaded_valid || memcmp(&h->zs_uploaded, &zr, sizeof zr) != 0) {
        // (the pinned staging copy is rewritten only here; every step ends with a wait for the stream, so no copy of it is in flight)
        *h->h_zs = zr;
        HIPCHK(hipMemcpyAsync(h->zs_ranks.p, h->h_zs, sizeof(ZsRanks), hipMemcpyHostToDevice, st));
        h->zs_uploaded = zr; h->zs_uploaded_valid = true;
    }
    uint32_t hsize = 0;
    if (R > 1) {
        uint32_t band_cap = std::max<uint32_t>(V, 64u);
        if (!whole_band && h->zs_band_plan && (uint64_t) h->zs_band_plan + h->zs_band_plan / 4 + 1024 < V) band_cap = h->zs_band_plan + h->zs_band_plan / 4 + 1024;
        hsize = 2u * band_cap + 64u;
        SDCHK(ensure_grow(h, h->zs_hkeys, 12ull * hsize));                    // keys (8 B per slot), then the slots' smallest vertex ids (4 B)
    }
    uint32_t *tplan = (uint32_t *) h->zs_tables.p, *vplan = tplan + 2 * nseg, *vstart = vplan + 2 * nseg, *newcnt = vstart + nseg;
    const uint32_t vcap = std::max<uint32_t>(V, 64u);
    uint32_t *segof = (uint32_t *) h->zs_pervert.p, *redirect = segof + vcap, *map = redirect + vcap, *blockdrop = map + vcap;
    HIPCHK(launch_zs_merge(st, h->grid, (const Vertex *) h->zs_verts.p, (const uint32_t *) h->zs_idx.p, d_counts, SHARD_ROW, zr, (const ZsRanks *) h->zs_ranks.p,
                           tplan, vplan, vstart, newcnt, segof, redirect, map, blockdrop, (uint32_t *) h->zs_hints.p, (uint64_t *) h->zs_hkeys.p, hsize, (Vertex *) h->zs_vout.p,
                           (uint32_t) std::min<size_t>(h->zs_vout.cap / sizeof(Vertex), 0xFFFFFFFFu), (uint32_t *) h->zs_iout.p,
                           (uint32_t) std::min<size_t>(h->zs_iout.cap / sizeof(IndexTriangle), 0xFFFFFFFFu), (uint32_t *) h->zs_ctl.p, seq, h->h_zs_pub));
    (void) T;
    return SDMESH_OK;
}

// after the stream has been waited for: the merge's verdict (root).  May run the merge once more with a hash that holds every vertex.
int zs_merge_finish(sdmesh_handle *h, const ZsRanks &zr, const uint32_t *d_counts, uint32_t seq, int levels, Point voxel_size, sdmesh_regen_info *info) {
    for (int attempt = 0; attempt < 2; attempt++) {
        const uint32_t *pub = h->h_zs_pub;
        if (pub[ZC_SEQ] != seq) return fail(h, SDMESH_ERR_INTERNAL, "the indexed merge did not publish its results (step %u, found %u)", seq, pub[ZC_SEQ]);
        h->zs_band_plan = pub[ZC_NBAND];
        if (pub[ZC_BANDFLAGS] & 1u) return fail(h, SDMESH_ERR_WELD_RANGE, "the indexed gather needs vertices inside the packed key range (option shard_indexed = 0 welds on the root instead)");
        if (pub[ZC_BANDFLAGS] & 64u) {                                       // the band outgrew its hash -> once more with room for every vertex
            if (attempt == 1) return fail(h, SDMESH_ERR_INTERNAL, "the face-band hash overflowed although it had room for every vertex");
            SDCHK(zs_merge_enqueue(h, zr, d_counts, true, seq));
            HIPCHK(hipStreamSynchronize(h->stream));
            continue;
        }
        if (pub[ZC_PLANFLAGS] | pub[ZC_VFLAGS]) return fail(h, SDMESH_ERR_INTERNAL, "indexed merge: plan flags 0x%x (rows invalid or an output buffer too small)", pub[ZC_PLANFLAGS] | pub[ZC_VFLAGS]);
        for (uint32_t r = 0; r < zr.nranks; r++)
            if (pub[ZC_RANKBAD + r]) return fail(h, SDMESH_ERR_INTERNAL, "indexed merge: the first-seen vertex counts of rank %u do not add up to its vertices", r);
        h->res_v = (Vertex *) h->zs_vout.p;                                  // the merged mesh is the handle's mesh now
        h->res_i = (uint32_t *) h->zs_iout.p;
        h->n_tris = pub[ZC_TRIS];
        h->n_verts = pub[ZC_VERTS];
        h->n_unique = h->n_verts;
        h->mesh_valid = true;
        h->welded = true;
        h->last_direct = false;
        h->vox_valid = false;                                              // (the root's voxel list is its own slab's, not the mesh's)
        if (info) {
            fill_info(h, info, 0, false, 0.0f);
            info->levels = (uint32_t) levels;
            info->voxel_size = voxel_size;
        }
        return SDMESH_OK;
    }
    return SDMESH_ERR_INTERNAL;
}

// grouped messages of one step (no HIP call and no early return between GroupStart and GroupEnd: a group is always closed)
struct ZsMsg { void *p; size_t count; ncclDataType_t type; int peer; bool send; };
int zs_exchange(sdmesh_handle *h, const std::vector<ZsMsg> &msgs) {
    if (msgs.empty()) return SDMESH_OK;
    SDCHK(tp_group(h, true));
    int first = SDMESH_OK;
    std::string why;
    for (const ZsMsg &m : msgs) {
        first = m.send ? tp_send(h, m.p, m.count, m.type, m.peer, h->stream) : tp_recv(h, m.p, m.count, m.type, m.peer, h->stream);
        if (first != SDMESH_OK) { why = h->err; break; }
    }
    const int end = tp_group(h, false);
    if (first != SDMESH_OK) return fail(h, first, "%s", why.c_str());          // (the first failure is the one reported)
    return end;
}

// this rank's slab, meshed AND welded with the ordinary synchronous regen; the column triangle counts land in `row` (pre-zeroed) on the way
int zs_local_sync(sdmesh_handle *h, int levels, int part, int nparts, uint32_t *row, sdmesh_regen_info *slab) {
    SDCHK(ensure(h, h->zs_colscratch, 8192));
    const int saved_skip = h->skip_weld;
    h->skip_weld = 0;
    h->colcnt = row;
    const int rc = regen_impl(h, levels, SDMESH_PART_ZSLAB, part, nparts, slab);
    h->colcnt = nullptr;
    h->skip_weld = saved_skip;
    SDCHK(rc);
    if (h->last_direct || !h->welded) return fail(h, SDMESH_ERR_INTERNAL, "a slab took the direct path: no column counts");
    if (h->wide_active) return fail(h, SDMESH_ERR_WELD_RANGE, "the indexed gather needs the packed weld keys (option shard_indexed = 0 welds on the root instead)");
    return SDMESH_OK;
}

int zs_step_sync(sdmesh_handle *h, int levels, int root, int inject, sdmesh_regen_info *info) {
    const int R = h->comm_nranks, me = h->comm_rank;
    hipStream_t st = h->stream;
    h->zsp.valid = false;
    h->zs_cur = 0;
    uint32_t *d_counts = zs_counts(h, 0), *d_row = d_counts + (size_t) me * SHARD_ROW, *d_next = zs_counts(h, 1) + (size_t) me * SHARD_ROW;
    HIPCHK(hipMemsetAsync(d_row, 0, 8192, st));
    HIPCHK(hipMemsetAsync(d_next, 0, 8192, st));
    // 1. this rank's slab, welded locally; any failure in here becomes this rank's status word
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    const int lrc = inject == 1 ? fail(h, SDMESH_ERR_INTERNAL, "injected failure (option inject_fail)") : zs_local_sync(h, levels, me, R, d_row, &slab);
    const uint32_t t_mine = lrc == SDMESH_OK ? h->n_tris : 0u, v_mine = lrc == SDMESH_OK ? h->n_verts : 0u;
    // 2. the row's vertex counts and record, then the all-gather -- ALWAYS entered
    HIPCHK(launch_zs_local(st, d_row, nullptr, h->d_ctr + CTR_NTRI, h->d_ctr + CTR_NVERT, h->d_status, (uint32_t) lrc, zs_root_capacity_v(h), zs_root_capacity_t(h),
                           (uint32_t) inject, 0u));
    SDCHK(tp_all_gather(h, d_row, d_counts, SHARD_ROW, ncclUint32, st));                     // in place
    HIPCHK(hipMemcpyAsync(h->h_counts, d_counts, 4ull * (size_t) R * SHARD_ROW, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // 3. every rank, on the same numbers
    SDCHK(peer_failure(h, R, me, h->h_counts, SHARD_ROW, SHARD_REC + ZR_STATUS, lrc));
    uint32_t cap_v[32] = {0}, cap_t[32] = {0}, Vr[32] = {0}, Tr[32] = {0};
    uint64_t tslots = 0, vslots = 0;
    bool any_inject2 = false;
    for (int r = 0; r < R; r++) {
        const uint32_t *row = h->h_counts + (size_t) r * SHARD_ROW, *rec = row + SHARD_REC;
        if (rec[ZR_STATUS]) return fail(h, SDMESH_ERR_INTERNAL, "rank %d: its own row reports status 0x%x", r, rec[ZR_STATUS]);      // (me: the device-side checks of the row)
        if (rec[ZR_FORM] != 1u) return fail(h, SDMESH_ERR_INVALID, "rank %d runs the raw-list form of the z-slab gather (option shard_indexed differs between the ranks)", r);
        Tr[r] = rec[ZR_TRIS]; Vr[r] = rec[ZR_VERTS];
        cap_v[r] = zs_padded(h, Vr[r]); cap_t[r] = zs_padded(h, Tr[r]);
        tslots += cap_t[r]; vslots += cap_v[r];
        any_inject2 = any_inject2 || rec[ZR_INJECT] == 2u;
    }
    if (tslots > 0x0FFFFFFFull || vslots > 0x0FFFFFFFull) return fail(h, SDMESH_ERR_INVALID, "too many triangles for one mesh");
    // 4. the root's buffers (the slots are plan-sized already: the next step may be a planned one); grown behind a second agreement only if
    //    its published capacity is too small
    const uint32_t *root_rec = h->h_counts + (size_t) root * SHARD_ROW + SHARD_REC;
    if (vslots > root_rec[ZR_CAP_V] || tslots > root_rec[ZR_CAP_T] || root_rec[ZR_CAP_V] == 0u || any_inject2) {
        int grc = SDMESH_OK;
        if (me == root) grc = inject == 2 ? fail(h, SDMESH_ERR_HIP, "injected allocation failure (option inject_fail)") : zs_root_ensure(h, R, vslots, tslots);
        else if (inject == 2) grc = fail(h, SDMESH_ERR_HIP, "injected allocation failure (option inject_fail)");
        uint32_t worst = 0;
        SDCHK(agree_flag(h, grc == SDMESH_OK ? 0u : 1u, &worst));
        if (grc != SDMESH_OK) return grc;
        if (worst) return fail(h, SDMESH_ERR_PEER, "a rank (the root is rank %d) could not grow its buffers; every rank returns an error", root);
    }
    // 5. the plan of the next steps: the same on every rank (from here on all of them hold every buffer a planned step needs, or will
    //    find out before they enqueue anything)
    sdmesh_handle::ZsPlan &pl = h->zsp;
    pl.levels = levels; pl.root = root; pl.R = R; pl.scene = h->scene; pl.slab_sig = zs_slab_sig(h, R);
    uint32_t sig = 2166136261u;
    for (int r = 0; r < R; r++) { pl.cap_v[r] = cap_v[r]; pl.cap_t[r] = cap_t[r]; sig = ((sig ^ cap_v[r]) * 16777619u ^ cap_t[r]) * 16777619u; }
    pl.sig = sig | 1u;
    pl.valid = h->shard_planned != 0;
    h->zs_cur = 1;                                   // (buffer 1's row of mine was cleared above and nothing has touched it)
    ZsRanks zr;
    zs_slot_table(h, R, cap_v, cap_t, levels, zr);
    // 6. payload: exact sizes (they are host values here)
    std::vector<ZsMsg> msgs;
    if (me != root) {
        if (v_mine) msgs.push_back({h->vertices.p, (size_t) v_mine * 6, ncclFloat32, root, true});
        if (t_mine) msgs.push_back({h->indices.p, (size_t) t_mine * 3, ncclUint32, root, true});
        SDCHK(zs_exchange(h, msgs));
        HIPCHK(hipStreamSynchronize(st));
        if (info) *info = slab;
        return SDMESH_OK;
    }
    if (Vr[me]) HIPCHK(hipMemcpyAsync((Vertex *) h->zs_verts.p + zr.vbase[me], h->vertices.p, sizeof(Vertex) * (size_t) Vr[me], hipMemcpyDeviceToDevice, st));
    if (Tr[me]) HIPCHK(hipMemcpyAsync((IndexTriangle *) h->zs_idx.p + zr.tbase[me], h->indices.p, sizeof(IndexTriangle) * (size_t) Tr[me], hipMemcpyDeviceToDevice, st));
    for (int r = 0; r < R; r++) {
        if (r == me) continue;
        if (Vr[r]) msgs.push_back({(Vertex *) h->zs_verts.p + zr.vbase[r], (size_t) Vr[r] * 6, ncclFloat32, r, false});
        if (Tr[r]) msgs.push_back({(IndexTriangle *) h->zs_idx.p + zr.tbase[r], (size_t) Tr[r] * 3, ncclUint32, r, false});
    }
    SDCHK(zs_exchange(h, msgs));
    // 7. the merge (from here on nobody waits for the root any more: a failure is the root's alone)
    const uint32_t seq = ++h->zs_seq;
    HIPCHK(hipEventRecord(h->ev0, st));
    SDCHK(zs_merge_enqueue(h, zr, d_counts, false, seq));
    HIPCHK(hipEventRecord(h->ev1, st));
    HIPCHK(hipStreamSynchronize(st));
    SDCHK(zs_merge_finish(h, zr, d_counts, seq, levels, slab.voxel_size, info));
    if (info) HIPCHK(hipEventElapsedTime(&info->device_ms, h->ev0, h->ev1));          // the root's merge alone
    return SDMESH_OK;
}

// planned route, local part of rank `me` (handle h): buffers, the slab's chain from its capacity plan with the column counts on the way
// (a handle without a plan runs its regen synchronously here: still no exchange of host values between the ranks), the two count kernels.
// slot_v / slot_i != nullptr: the weld writes straight into that slot of the root's gather buffers (the root itself).  Returns what the host
// knows about the local phase when everything is enqueued; *pending = the chain's own verdict is still outstanding (zs_planned_local_end).
int zs_planned_local(sdmesh_handle *h, int levels, int me, int R, const sdmesh_handle::ZsPlan &pl, Vertex *slot_v, IndexTriangle *slot_i, int inject,
                     uint32_t *d_row, uint32_t *d_next, sdmesh_regen_info *slab) {
    hipStream_t st = h->stream;
    if (!slot_v) {            // peers send plan-sized messages out of their own mesh buffers (nothing may allocate once the chain is in flight)
        SDCHK(ensure(h, h->vertices, sizeof(Vertex) * (size_t) pl.cap_v[me]));
        SDCHK(ensure(h, h->indices, sizeof(IndexTriangle) * (size_t) pl.cap_t[me]));
    }
    SDCHK(ensure(h, h->zs_colscratch, 8192));
    const int saved_skip = h->skip_weld;
    h->skip_weld = 0;
    h->colcnt = d_row;
    if (slot_v) { h->out_v = slot_v; h->out_i = (uint32_t *) slot_i; h->out_vcap = pl.cap_v[me]; }
    int lrc = inject == 1 ? fail(h, SDMESH_ERR_INTERNAL, "injected failure (option inject_fail)") : regen_impl(h, levels, SDMESH_PART_ZSLAB, me, R, nullptr, 1);
    bool ended = false;
    if (lrc == SDMESH_OK && !h->pend_async) { lrc = regen_impl(h, 0, 0, 0, 1, slab, 2); ended = true; }      // no plan: the synchronous regen, now
    h->colcnt = nullptr;
    h->out_v = nullptr; h->out_i = nullptr; h->out_vcap = 0;
    h->skip_weld = saved_skip;
    if (lrc == SDMESH_OK && ended && (h->last_direct || !h->welded || h->wide_active)) lrc = fail(h, SDMESH_ERR_INTERNAL, "a slab left the packed-key lattice path");
    if (lrc != SDMESH_OK && h->pend_active) { sdmesh_regen_info tmp; (void) regen_impl(h, 0, 0, 0, 1, &tmp, 2); }
    HIPCHK(launch_zs_local(st, d_row, d_next, h->d_ctr + CTR_NTRI, h->d_ctr + CTR_NVERT, h->d_status, (uint32_t) lrc, zs_root_capacity_v(h), zs_root_capacity_t(h),
                           (uint32_t) inject, pl.sig));
    return lrc;
}

// after the step's wait: the chain's own verdict (a chain that overflowed its plan is repeated in there; its row said so already)
int zs_planned_local_end(sdmesh_handle *h, sdmesh_regen_info *slab) {
    if (!h->pend_active) return SDMESH_OK;
    return regen_impl(h, 0, 0, 0, 1, slab, 2);
}

// the records a planned step published: does every rank fit the plan?
bool zs_records_fit(const uint32_t *pub, int R, const sdmesh_handle::ZsPlan &pl) {
    for (int r = 0; r < R; r++) {
        const uint32_t *rec = pub + ZC_WORDS + 8 * r;
        if (rec[ZR_STATUS] != 0u || rec[ZR_FORM] != 1u || rec[ZR_VERTS] > pl.cap_v[r] || rec[ZR_TRIS] > pl.cap_t[r] || rec[ZR_INJECT] != 0u || rec[ZR_SIG] != pl.sig) return false;
    }
    return true;
}

int zs_step_planned(sdmesh_handle *h, int levels, int root, int inject, sdmesh_regen_info *info) {
    const int R = h->comm_nranks, me = h->comm_rank;
    hipStream_t st = h->stream;
    const sdmesh_handle::ZsPlan &pl = h->zsp;
    const int cur = h->zs_cur;
    uint32_t *d_counts = zs_counts(h, cur), *d_row = d_counts + (size_t) me * SHARD_ROW, *d_next = zs_counts(h, cur ^ 1) + (size_t) me * SHARD_ROW;
    ZsRanks zr;
    zs_slot_table(h, R, pl.cap_v, pl.cap_t, levels, zr);
    const bool is_root = me == root;
    if (is_root) SDCHK(zs_root_ensure(h, R, zr.vbase[R], zr.tbase[R]));
    // 1. the slab's chain + 2. vertex counts and record; 3. all-gather; 4. plan-sized payload; 5. merge / publish -- all enqueued blind
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    const int lrc = zs_planned_local(h, levels, me, R, pl, is_root ? (Vertex *) h->zs_verts.p + zr.vbase[me] : nullptr,
                                     is_root ? (IndexTriangle *) h->zs_idx.p + zr.tbase[me] : nullptr, inject, d_row, d_next, &slab);
    if (lrc == SDMESH_ERR_HIP) return lrc;                                  // (the enqueue itself failed: nothing sane can follow)
    const uint32_t seq = ++h->zs_seq;
    SDCHK(tp_all_gather(h, d_row, d_counts, SHARD_ROW, ncclUint32, st));                     // in place
    std::vector<ZsMsg> msgs;
    if (!is_root) {
        msgs.push_back({h->vertices.p, (size_t) pl.cap_v[me] * 6, ncclFloat32, root, true});
        msgs.push_back({h->indices.p, (size_t) pl.cap_t[me] * 3, ncclUint32, root, true});
        SDCHK(zs_exchange(h, msgs));
        HIPCHK(launch_zs_publish(st, d_counts, SHARD_ROW, (uint32_t) R, nullptr, seq, h->h_zs_pub));
    } else {
        for (int r = 0; r < R; r++) {
            if (r == me) continue;
            msgs.push_back({(Vertex *) h->zs_verts.p + zr.vbase[r], (size_t) pl.cap_v[r] * 6, ncclFloat32, r, false});
            msgs.push_back({(IndexTriangle *) h->zs_idx.p + zr.tbase[r], (size_t) pl.cap_t[r] * 3, ncclUint32, r, false});
        }
        SDCHK(zs_exchange(h, msgs));
        SDCHK(zs_merge_enqueue(h, zr, d_counts, false, seq));
    }
    // 6. the step's one host synchronisation
    HIPCHK(hipStreamSynchronize(st));
    h->zs_cur = cur ^ 1;
    const int erc = zs_planned_local_end(h, &slab);
    const uint32_t *pub = h->h_zs_pub;
    if (pub[ZC_SEQ] != seq) return fail(h, SDMESH_ERR_INTERNAL, "the planned z-slab step did not publish its records (step %u, found %u)", seq, pub[ZC_SEQ]);
    // (erc / lrc are this rank's alone, but they imply a non-zero status in this rank's record, which every rank sees: the decision is collective)
    if (erc != SDMESH_OK || lrc != SDMESH_OK || !zs_records_fit(pub, R, pl)) return ZS_REDO;
    if (!is_root) {
        if (info) *info = slab;
        return SDMESH_OK;
    }
    return zs_merge_finish(h, zr, d_counts, seq, levels, slab.voxel_size, info);
}

int sharded_regen_indexed(sdmesh_handle *h, int levels, int root, sdmesh_regen_info *info) {
    const int R = h->comm_nranks;
    const int inject = h->inject_fail;
    h->inject_fail = 0;
    SDCHK(zs_scratch(h, R));                                   // (a rank that cannot even get these few KB cannot tell anybody)
    const sdmesh_handle::ZsPlan &pl = h->zsp;
    if (h->shard_planned && pl.valid && pl.levels == levels && pl.root == root && pl.R == R && pl.scene == h->scene && pl.slab_sig == zs_slab_sig(h, R)) {
        const int rc = zs_step_planned(h, levels, root, inject, info);
        if (rc != ZS_REDO) { h->zs_last_route = 1; return rc; }
        h->zs_last_route = 3;
        h->mesh_valid = false;
        return zs_step_sync(h, levels, root, inject, info);
    }
    h->zs_last_route = 2;
    return zs_step_sync(h, levels, root, inject, info);
}

}  // namespace

extern "C" {

int sdmesh_sharded_route(const sdmesh_handle *h) { return h ? (int) h->zs_last_route : 0; }

// Every operation of the handle's collective layer once, with this rank as its own peer where a peer is needed: grouped send / recv of the
// three element types the steps use (odd counts, a megabyte), an out-of-place all-gather, both reductions.  *mismatches = words that did not
// arrive as sent (0 = the layer moves bytes as the steps assume).  Collective: every rank calls it.
int sdmesh_comm_selftest(sdmesh_handle *h, uint64_t *mismatches) {
    if (!h || !mismatches) return SDMESH_ERR_INVALID;
    if (!h->has_comm()) return fail(h, SDMESH_ERR_INVALID, "no communicator (sdmesh_comm_init)");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int R = h->comm_nranks, me = h->comm_rank;
    const size_t words = 1u << 18;                                          // 1 MiB of u32
    std::vector<uint32_t> src(words), dst(words);
    for (size_t i = 0; i < words; i++) src[i] = (uint32_t) (i * 2654435761u) ^ (uint32_t) (me + 1) * 0x9E3779B9u;
    uint32_t *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc((void **) &a, 4 * words));
    if (hipMalloc((void **) &b, 4 * words * (size_t) std::max(R, 1)) != hipSuccess) { (void) hipFree(a); return fail(h, SDMESH_ERR_HIP, "hipMalloc failed in the collective self-test"); }
    uint64_t bad = 0;
    int rc = SDMESH_OK;
    auto upload = [&]() { return hipMemcpyAsync(a, src.data(), 4 * words, hipMemcpyHostToDevice, st) == hipSuccess && hipMemsetAsync(b, 0xA5, 4 * words, st) == hipSuccess; };
    auto compare = [&](size_t n_words, size_t offset_words) {
        if (hipMemcpyAsync(dst.data(), b + offset_words, 4 * n_words, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = fail(h, SDMESH_ERR_HIP, "copy failed in the collective self-test"); return; }
        for (size_t i = 0; i < n_words; i++) bad += dst[i] != src[i] ? 1u : 0u;
    };
    const struct { ncclDataType_t type; size_t count, words; } cases[] = {{ncclUint32, words, words}, {ncclFloat32, 12345, 12345}, {ncclUint64, 777, 1554}, {ncclUint32, 1, 1}};
    for (const auto &c : cases) {
        if (rc != SDMESH_OK) break;
        if (!upload()) { rc = fail(h, SDMESH_ERR_HIP, "upload failed in the collective self-test"); break; }
        std::vector<ZsMsg> msgs;
        msgs.push_back({a, c.count, c.type, me, true});
        msgs.push_back({b, c.count, c.type, me, false});
        rc = zs_exchange(h, msgs);
        if (rc == SDMESH_OK) compare(c.words, 0);
    }
    if (rc == SDMESH_OK && upload()) {                                      // out of place: this rank's 64 words land in block `me`
        rc = tp_all_gather(h, a, b, 64, ncclUint32, st);
        if (rc == SDMESH_OK) compare(64, 64u * (size_t) me);
    }
    for (int mx = 0; mx < 2 && rc == SDMESH_OK; mx++) {
        const uint32_t mine = (uint32_t) me + 1u;
        if (hipMemcpyAsync(a, &mine, 4, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(h, SDMESH_ERR_HIP, "upload failed in the collective self-test"); break; }
        rc = tp_all_reduce_u32(h, a, b, 1, mx != 0, st);
        uint32_t got = 0;
        if (rc == SDMESH_OK && (hipMemcpyAsync(&got, b, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = fail(h, SDMESH_ERR_HIP, "copy failed in the collective self-test");
        if (rc == SDMESH_OK) bad += got != (mx ? (uint32_t) R : (uint32_t) (R * (R + 1) / 2)) ? 1u : 0u;
    }
    (void) hipStreamSynchronize(st);
    (void) hipFree(a); (void) hipFree(b);
    *mismatches = bad;
    return rc;
}


int sdmesh_sharded_regen(sdmesh_handle *h, int levels, int root, sdmesh_regen_info *info) {
    if (!h) return SDMESH_ERR_INVALID;
    // (argument errors are found before any communication: they must be the same on every rank, like the arguments)
    if (!h->has_comm()) return fail(h, SDMESH_ERR_INVALID, "no communicator (sdmesh_comm_init)");
    const int R = h->comm_nranks, me = h->comm_rank;
    if (root < 0 || root >= R) return fail(h, SDMESH_ERR_INVALID, "root %d outside the communicator", root);
    if (h->shard_indexed) return sharded_regen_indexed(h, levels, root, info);
    hipStream_t st = h->stream;
    const int inject = h->inject_fail;
    h->inject_fail = 0;
    // 0. the fixed-size scratch of the exchange itself.  (A rank that cannot even get these few KB cannot tell anybody.)
    SDCHK(ensure(h, h->shard_counts, 4ull * (size_t) R * SHARD_ROW));
    SDCHK(ensure(h, h->shard_lists, sizeof(void *) * (size_t) R));
    SDCHK(ensure(h, h->shard_plan, 8ull * (size_t) R * 1024));
    SDCHK(ensure_shard_host(h, R));
    uint32_t *d_counts = (uint32_t *) h->shard_counts.p, *d_row = d_counts + (size_t) me * SHARD_ROW;
    // 1. this rank's level-0 z-slab: refine^levels, marching cubes, projection, orientation; no weld; then its compact
    //    triangle list and its per-column counts.  Any failure in here becomes this rank's status word.
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    uint32_t n_mine = 0;
    auto local_phase = [&]() -> int {
        const int saved_skip = h->skip_weld;
        h->skip_weld = 1;
        const int rc = inject == 1 ? fail(h, SDMESH_ERR_INTERNAL, "injected failure (option inject_fail)") : regen_impl(h, levels, SDMESH_PART_ZSLAB, me, R, &slab);
        h->skip_weld = saved_skip;
        SDCHK(rc);
        if (h->last_direct) return fail(h, SDMESH_ERR_INTERNAL, "a slab took the direct path: no column counts");
        n_mine = h->n_tris;
        SDCHK(ensure_grow(h, h->shard_mine, sizeof(Triangle) * (size_t) std::max<uint32_t>(n_mine, 1)));
        if (n_mine)
            HIPCHK(launch_expand_compact(st, h->grid, (const Vertex *) h->uverts.p, (const uint32_t *) h->tri_e.p, (const uint8_t *) h->flip.p,
                                         h->d_ctr + CTR_NTRI, (Triangle *) h->shard_mine.p, n_mine));
        HIPCHK(hipMemsetAsync(d_row, 0, 8192, st));                        // (column triangle counts | vertex counts: none in the raw form)
        if (h->n_vox) {
            HIPCHK(launch_set_u32(st, h->d_ctr + CTR_SCRATCH, h->n_vox));
            HIPCHK(launch_column_counts(st, std::min(h->grid, (int) tiles(h->n_vox)), (const Point *) h->vox[h->cur].p,
                                        h->d_ctr + CTR_SCRATCH, (const uint8_t *) h->cubeidx.p, d_row));
        }
        return SDMESH_OK;
    };
    const int lrc = local_phase();
    if (lrc != SDMESH_OK) { n_mine = 0; HIPCHK(hipMemsetAsync(d_row, 0, 8192, st)); }
    // 2. the all-gather of the counts -- ALWAYS entered; record = {status, triangles, what this rank could receive / merge}
    HIPCHK(launch_pack_record(st, d_row + SHARD_REC, nullptr, nullptr, (uint32_t) lrc, n_mine, cap_items(h->shard_recv, sizeof(Triangle)),
                              cap_items(h->shard_merged, sizeof(Triangle))));
    // (k_pack_record writes {v0, v1 | status, v2, v3}: word 0 = status, word 1 = triangles here)
    HIPCHK(launch_pack_record(st, d_row + SHARD_REC + 4, nullptr, nullptr, 0u, 0u, (uint32_t) inject, 0u));     // {vertices: none, raw form, inject}
    SDCHK(tp_all_gather(h, d_row, d_counts, SHARD_ROW, ncclUint32, st));                     // in place
    HIPCHK(hipMemcpyAsync(h->h_counts, d_counts, 4ull * (size_t) R * SHARD_ROW, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // 3. every rank, on the same numbers: did anybody fail, do the counts add up, does the mesh fit one weld
    SDCHK(peer_failure(h, R, me, h->h_counts, SHARD_ROW, SHARD_REC, lrc));
    std::vector<uint64_t> tr((size_t) R, 0);
    uint64_t total = 0, foreign = 0;
    for (int r = 0; r < R; r++) {
        const uint32_t *row = h->h_counts + (size_t) r * SHARD_ROW;
        for (int c = 0; c < 1024; c++) tr[r] += row[c];
        if (tr[r] != row[SHARD_REC + 1])
            return fail(h, SDMESH_ERR_INTERNAL, "rank %d: column counts (%llu) disagree with the slab's triangle count (%u)", r, (unsigned long long) tr[r], row[SHARD_REC + 1]);
        if (row[SHARD_REC + 5] != 0u) return fail(h, SDMESH_ERR_INVALID, "rank %d runs the indexed form of the z-slab gather (option shard_indexed differs between the ranks)", r);
        total += tr[r];
        if (r != root) foreign += tr[r];
    }
    if (total > 0x0FFFFFFFull) return fail(h, SDMESH_ERR_INVALID, "too many triangles for one weld");
    // 4. the root's receive / merge buffers; grown behind a second agreement only if its published capacity is too small
    const uint32_t *root_row = h->h_counts + (size_t) root * SHARD_ROW;
    bool any_inject2 = false;                              // (an injected growth failure on ANY rank: every rank takes the same branch)
    for (int r = 0; r < R; r++) any_inject2 = any_inject2 || h->h_counts[(size_t) r * SHARD_ROW + SHARD_REC + ZR_INJECT] == 2u;
    if (foreign > root_row[SHARD_REC + 2] || total > root_row[SHARD_REC + 3] || any_inject2) {
        int grc = SDMESH_OK;
        if (inject == 2) grc = fail(h, SDMESH_ERR_HIP, "injected allocation failure (option inject_fail)");
        else if (me == root) {
            grc = ensure_grow(h, h->shard_recv, sizeof(Triangle) * (size_t) std::max<uint64_t>(foreign, 1));
            if (grc == SDMESH_OK) grc = ensure_grow(h, h->shard_merged, sizeof(Triangle) * (size_t) std::max<uint64_t>(total, 1));
        }
        uint32_t worst = 0;
        SDCHK(agree_flag(h, grc == SDMESH_OK ? 0u : 1u, &worst));
        if (grc != SDMESH_OK) return grc;
        if (worst) return fail(h, SDMESH_ERR_PEER, "a rank (the root is rank %d) could not grow its buffers; every rank returns an error", root);
    }
    if (me != root) {
        // 5a. peers: the slab's triangles go straight to the root (each peer on its own xGMI link)
        if (n_mine) SDCHK(tp_send(h, h->shard_mine.p, (size_t) n_mine * 18, ncclFloat32, root, st));
        HIPCHK(hipStreamSynchronize(st));
        if (info) *info = slab;
        return SDMESH_OK;
    }
    // 5b. root: grouped receives
    uint64_t off = 0;
    for (int r = 0; r < R; r++) {
        if (r == root) { h->h_lists[r] = h->shard_mine.p; continue; }
        h->h_lists[r] = (Triangle *) h->shard_recv.p + off;
        off += tr[r];
    }
    {
        std::vector<ZsMsg> msgs;                               // (grouped; the group is closed on every path)
        for (int r = 0; r < R; r++)
            if (r != root && tr[r]) msgs.push_back({h->h_lists[r], (size_t) tr[r] * 18, ncclFloat32, r, false});
        SDCHK(zs_exchange(h, msgs));
    }
    // 6. order-restoring merge (HIP kernel), then the first-seen weld of the merged list = the single-GPU mesh.  (From here
    //    on nobody waits for the root any more: a failure is the root's alone.)
    SDCHK(merge_lists(h, d_counts, SHARD_ROW, R, (Triangle *) h->shard_merged.p, (size_t) total));
    SDCHK(weld_device_impl(h, (const Triangle *) h->shard_merged.p, (size_t) total, info, true));      // (all slots valid: no compaction pass)
    if (info) {
        info->levels = (uint32_t) levels;
        info->voxel_size = slab.voxel_size;
    }
    return SDMESH_OK;
}

}  // extern "C"

namespace {

// the rehearsal's planned route: the planned step's kernels for every handle (slots with head-room, counts read on the device, the root's
// weld writing into its slot, the alternating count buffers), copies of the PLAN's sizes in place of the messages.  ZS_REDO like the real one.
int zs_rehearsal_planned(sdmesh_handle **hs, int R, int levels, int root, sdmesh_regen_info *info) {
    sdmesh_handle *H = hs[root], *h = H;
    const sdmesh_handle::ZsPlan &pl = H->zsp;
    const int cur = H->zs_cur;
    uint32_t *d_counts = zs_counts(H, cur), *d_nextbuf = zs_counts(H, cur ^ 1);
    ZsRanks zr;
    zs_slot_table(H, R, pl.cap_v, pl.cap_t, levels, zr);
    SDCHK(zs_root_ensure(H, R, zr.vbase[R], zr.tbase[R]));
    std::vector<sdmesh_regen_info> slab((size_t) R);
    std::vector<int> lrc((size_t) R, SDMESH_OK);
    for (int r = 0; r < R; r++) {
        h = hs[r];
        const int inject = h->inject_fail;
        h->inject_fail = 0;
        memset(&slab[(size_t) r], 0, sizeof(sdmesh_regen_info));
        lrc[(size_t) r] = zs_planned_local(h, levels, r, R, pl, r == root ? (Vertex *) H->zs_verts.p + zr.vbase[r] : nullptr,
                                           r == root ? (IndexTriangle *) H->zs_idx.p + zr.tbase[r] : nullptr, inject, d_counts + (size_t) r * SHARD_ROW,
                                           d_nextbuf + (size_t) r * SHARD_ROW, &slab[(size_t) r]);
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h = H;
    for (int r = 0; r < R; r++) {
        if (r == root) continue;
        HIPCHK(hipMemcpyAsync((Vertex *) H->zs_verts.p + zr.vbase[r], hs[r]->vertices.p, sizeof(Vertex) * (size_t) pl.cap_v[r], hipMemcpyDeviceToDevice, H->stream));
        HIPCHK(hipMemcpyAsync((IndexTriangle *) H->zs_idx.p + zr.tbase[r], hs[r]->indices.p, sizeof(IndexTriangle) * (size_t) pl.cap_t[r], hipMemcpyDeviceToDevice, H->stream));
    }
    const uint32_t seq = ++H->zs_seq;
    SDCHK(zs_merge_enqueue(H, zr, d_counts, false, seq));
    HIPCHK(hipStreamSynchronize(H->stream));
    H->zs_cur = cur ^ 1;
    bool redo = false;
    for (int r = 0; r < R; r++) redo = (zs_planned_local_end(hs[r], &slab[(size_t) r]) != SDMESH_OK) || lrc[(size_t) r] != SDMESH_OK || redo;
    const uint32_t *pub = H->h_zs_pub;
    if (pub[ZC_SEQ] != seq) return fail(H, SDMESH_ERR_INTERNAL, "the planned z-slab step did not publish its records (step %u, found %u)", seq, pub[ZC_SEQ]);
    if (redo || !zs_records_fit(pub, R, pl)) return ZS_REDO;
    return zs_merge_finish(H, zr, d_counts, seq, levels, slab[(size_t) root].voxel_size, info);
}

}  // namespace

extern "C" int sdmesh_sharded_regen_rehearsal(sdmesh_handle **hs, int nranks, int levels, int root, sdmesh_regen_info *info) {
    // The indexed z-slab gather for `nranks` handles of ONE process on ONE device, device-to-device copies in place of the RCCL
    // messages: every kernel of sdmesh_sharded_regen's indexed form on BOTH its routes -- the first call (and any call whose sizes,
    // root, scene or slabs differ from the plan the root handle holds) takes the synchronous route and commits the plan, a repeat
    // takes the planned one (tests and single-GPU measurements of the root's share)
    if (!hs || nranks < 1 || nranks > 32 || root < 0 || root >= nranks) return SDMESH_ERR_INVALID;
    for (int r = 0; r < nranks; r++)
        if (!hs[r] || hs[r]->device != hs[0]->device) return SDMESH_ERR_INVALID;
    sdmesh_handle *H = hs[root], *h = H;                   // (`h`: what the error macros report through)
    const int R = nranks;
    HIPCHK(hipSetDevice(H->device));
    SDCHK(zs_scratch(H, R));
    for (int r = 0; r < R; r++)
        if (hs[r]->slab_nparts != H->slab_nparts || (H->slab_nparts && (hs[r]->slab_mode != H->slab_mode || memcmp(hs[r]->slab_bounds, H->slab_bounds, sizeof(uint32_t) * (size_t) (H->slab_nparts + 1)))))
            return fail(H, SDMESH_ERR_INVALID, "handle %d has other slab bounds than the root", r);
    {
        const sdmesh_handle::ZsPlan &pl = H->zsp;
        bool planned = H->shard_planned && pl.valid && pl.levels == levels && pl.root == root && pl.R == R && pl.slab_sig == zs_slab_sig(H, R);
        for (int r = 0; planned && r < R; r++) planned = hs[r]->scene == pl.scene;
        if (planned) {
            const int rc = zs_rehearsal_planned(hs, R, levels, root, info);
            if (rc != ZS_REDO) { H->zs_last_route = 1; return rc; }
            H->zs_last_route = 3;
        } else {
            H->zs_last_route = 2;
        }
    }
    H->zsp.valid = false;
    uint32_t *d_counts = zs_counts(H, 0);
    HIPCHK(hipMemsetAsync(d_counts, 0, 4ull * (size_t) R * SHARD_ROW, H->stream));
    HIPCHK(hipMemsetAsync(zs_counts(H, 1), 0, 4ull * (size_t) R * SHARD_ROW, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    uint32_t Tr[32] = {0}, Vr[32] = {0}, cap_v[32] = {0}, cap_t[32] = {0};
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    uint64_t vslots = 0, tslots = 0;
    for (int r = 0; r < R; r++) {
        h = hs[r];
        const int inject = h->inject_fail;
        h->inject_fail = 0;
        uint32_t *row = d_counts + (size_t) r * SHARD_ROW;
        const int rc = inject == 1 ? fail(h, SDMESH_ERR_INTERNAL, "injected failure (option inject_fail)") : zs_local_sync(h, levels, r, R, row, &slab);
        if (rc != SDMESH_OK) { if (h != H) (void) fail(H, rc, "slab %d: %s", r, h->err); return rc; }
        HIPCHK(launch_zs_local(h->stream, row, nullptr, h->d_ctr + CTR_NTRI, h->d_ctr + CTR_NVERT, h->d_status, 0u, 0u, 0u, 0u, 0u));
        HIPCHK(hipStreamSynchronize(h->stream));
        Tr[r] = h->n_tris; Vr[r] = h->n_verts;
        cap_v[r] = zs_padded(H, Vr[r]); cap_t[r] = zs_padded(H, Tr[r]);
        vslots += cap_v[r]; tslots += cap_t[r];
    }
    h = H;
    if (vslots > 0x0FFFFFFFull || tslots > 0x0FFFFFFFull) return fail(h, SDMESH_ERR_INVALID, "too many vertices / triangles for one mesh");
    SDCHK(zs_root_ensure(H, R, vslots, tslots));
    sdmesh_handle::ZsPlan &pl = H->zsp;                    // the plan, committed as sdmesh_sharded_regen's synchronous route commits it
    pl.levels = levels; pl.root = root; pl.R = R; pl.scene = H->scene; pl.slab_sig = zs_slab_sig(H, R);
    uint32_t sig = 2166136261u;
    for (int r = 0; r < R; r++) { pl.cap_v[r] = cap_v[r]; pl.cap_t[r] = cap_t[r]; sig = ((sig ^ cap_v[r]) * 16777619u ^ cap_t[r]) * 16777619u; }
    pl.sig = sig | 1u;
    pl.valid = H->shard_planned != 0;
    H->zs_cur = 1;
    ZsRanks zr;
    zs_slot_table(H, R, cap_v, cap_t, levels, zr);
    for (int r = 0; r < R; r++) {
        if (Vr[r]) HIPCHK(hipMemcpyAsync((Vertex *) H->zs_verts.p + zr.vbase[r], hs[r]->vertices.p, sizeof(Vertex) * (size_t) Vr[r], hipMemcpyDeviceToDevice, H->stream));
        if (Tr[r]) HIPCHK(hipMemcpyAsync((IndexTriangle *) H->zs_idx.p + zr.tbase[r], hs[r]->indices.p, sizeof(IndexTriangle) * (size_t) Tr[r], hipMemcpyDeviceToDevice, H->stream));
    }
    const uint32_t seq = ++H->zs_seq;
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    SDCHK(zs_merge_enqueue(H, zr, d_counts, false, seq));
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    SDCHK(zs_merge_finish(H, zr, d_counts, seq, levels, slab.voxel_size, info));
    if (info) HIPCHK(hipEventElapsedTime(&info->device_ms, H->ev0, H->ev1));          // the root's merge alone
    return SDMESH_OK;
}

// The two halves of the indexed z-slab step as separate calls, for a caller that brings its own transport (the gloo-staged rehearsal
// with several processes on one GPU, partition.ShardedMesher): exactly the kernels of sdmesh_sharded_regen's synchronous route.
static_assert(SHARD_ROW == SDMESH_ZSLAB_ROW_WORDS, "sdmesh.h names the row size");
extern "C" int sdmesh_zslab_local(sdmesh_handle *h, int levels, int part, int nparts, uint32_t *d_row, sdmesh_regen_info *info) {
    if (!h || !d_row) return SDMESH_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemsetAsync(d_row, 0, 4ull * SHARD_ROW, h->stream));
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    SDCHK(zs_local_sync(h, levels, part, nparts, d_row, &slab));
    HIPCHK(launch_zs_local(h->stream, d_row, nullptr, h->d_ctr + CTR_NTRI, h->d_ctr + CTR_NVERT, h->d_status, 0u, 0u, 0u, 0u, 0u));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (info) *info = slab;
    return SDMESH_OK;
}

extern "C" int sdmesh_zslab_merge(sdmesh_handle *h, int nranks, int levels, const uint32_t *d_rows, const struct Vertex *const *d_verts,
                                  const struct IndexTriangle *const *d_tris, sdmesh_regen_info *info) {
    if (!h || !d_rows || !d_verts || !d_tris || nranks < 1 || nranks > 32 || levels < 0 || levels > SDMESH_MAX_LEVELS) return SDMESH_ERR_INVALID;
    const int R = nranks;
    HIPCHK(hipSetDevice(h->device));
    SDCHK(zs_scratch(h, R));
    h->zsp.valid = false;
    h->mesh_valid = false;
    uint32_t *d_counts = zs_counts(h, 0);
    HIPCHK(hipMemcpyAsync(d_counts, d_rows, 4ull * (size_t) R * SHARD_ROW, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_counts, d_counts, 4ull * (size_t) R * SHARD_ROW, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    uint32_t Tr[32] = {0}, Vr[32] = {0};
    uint64_t vtot = 0, ttot = 0;
    for (int r = 0; r < R; r++) {
        const uint32_t *rec = h->h_counts + (size_t) r * SHARD_ROW + SHARD_REC;
        if (rec[ZR_STATUS] || rec[ZR_FORM] != 1u) return fail(h, SDMESH_ERR_INVALID, "row %d is not a healthy row of the indexed form (status 0x%x)", r, rec[ZR_STATUS]);
        Tr[r] = rec[ZR_TRIS]; Vr[r] = rec[ZR_VERTS];
        if ((Vr[r] && !d_verts[r]) || (Tr[r] && !d_tris[r])) return SDMESH_ERR_INVALID;
        vtot += Vr[r]; ttot += Tr[r];
    }
    if (vtot > 0x0FFFFFFFull || ttot > 0x0FFFFFFFull) return fail(h, SDMESH_ERR_INVALID, "too many vertices / triangles for one mesh");
    SDCHK(zs_root_ensure(h, R, vtot, ttot));
    ZsRanks zr;
    zs_slot_table(h, R, Vr, Tr, levels, zr);              // (slots of exactly the slabs' sizes)
    for (int r = 0; r < R; r++) {
        if (Vr[r]) HIPCHK(hipMemcpyAsync((Vertex *) h->zs_verts.p + zr.vbase[r], d_verts[r], sizeof(Vertex) * (size_t) Vr[r], hipMemcpyDeviceToDevice, h->stream));
        if (Tr[r]) HIPCHK(hipMemcpyAsync((IndexTriangle *) h->zs_idx.p + zr.tbase[r], d_tris[r], sizeof(IndexTriangle) * (size_t) Tr[r], hipMemcpyDeviceToDevice, h->stream));
    }
    const uint32_t seq = ++h->zs_seq;
    SDCHK(zs_merge_enqueue(h, zr, d_counts, false, seq));
    HIPCHK(hipStreamSynchronize(h->stream));
    Point vs;
    vs.x = vs.y = vs.z = (MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR) / (float) (1u << levels);
    return zs_merge_finish(h, zr, d_counts, seq, levels, vs, info);
}

// ================================================================================================
// Multi-GPU: ONE mesh, x-slabs, DISTRIBUTED weld (the mesh stays distributed)
// ================================================================================================
namespace {

// Two routes through the same kernels, as in the z-slab gather (see there): SYNCHRONOUS (message lengths read back from the first all-gather;
// three host synchronisations; commits the plan) and PLANNED (the slab's chain, the band kernels, both all-gathers and plan-sized
// neighbour messages enqueued blind from a collective plan; one host synchronisation at the end; a step whose band sizes did not fit is
// repeated synchronously by every rank, because every rank sees the same records).
uint32_t dist_padded(const sdmesh_handle *h, uint32_t n) { return h->shard_slack ? n + n / 8u + 2048u : n + 1u; }

uint32_t dist_slab_sig(const sdmesh_handle *h, int R) {
    uint32_t s = 2166136261u;
    const bool own = h->slab_nparts == R && h->slab_mode == SDMESH_PART_XSLAB;
    for (int i = 0; own && i <= R; i++) s = (s ^ h->slab_bounds[i]) * 16777619u;
    return s ^ (own ? 0x9e3779b9u : 0u);
}

int dist_scratch(sdmesh_handle *h) {                       // the records of the two exchanges: DIST_REC words per rank, <= 32 ranks; their pinned landing zone
    SDCHK(ensure(h, h->dist_counts, 4ull * 32 * DIST_REC));
    SDCHK(ensure(h, h->dist_nups, 4ull * 32 * DIST_REC));
    if (!h->h_dist_pub) HIPCHK(hipHostMalloc((void **) &h->h_dist_pub, 4ull * (8 + 2 * 32 * DIST_REC), hipHostMallocDefault));
    return SDMESH_OK;
}

// band / numbering buffers for up to vcap vertices of this rank (a rank's own band is at most its vertices)
int dist_ensure_local(sdmesh_handle *h, size_t vcap) {
    vcap = std::max<size_t>(vcap, 64);
    SDCHK(ensure_grow(h, h->dist_upids, 4 * vcap));
    SDCHK(ensure_grow(h, h->dist_upkeys, 8 * vcap));
    SDCHK(ensure_grow(h, h->dist_loids, 4 * vcap));
    SDCHK(ensure_grow(h, h->dist_dupof, 4 * vcap));
    SDCHK(ensure_grow(h, h->dist_newidx, 4 * vcap));
    SDCHK(ensure_grow(h, h->dist_tilecnt, 4ull * (tiles((uint32_t) std::min<size_t>(vcap, 0xFFFFFF00u)) + 16)));
    SDCHK(ensure_grow(h, h->dist_upg, 4 * vcap));
    SDCHK(ensure(h, h->dist_verts, sizeof(Vertex) * vcap));
    return SDMESH_OK;
}
uint32_t dist_local_capacity(const sdmesh_handle *h) {         // vertices the buffers above hold
    const size_t c = std::min(std::min(std::min(h->dist_upids.cap / 4, h->dist_upkeys.cap / 8), std::min(h->dist_loids.cap / 4, h->dist_dupof.cap / 4)),
                              std::min(std::min(h->dist_newidx.cap / 4, h->dist_upg.cap / 4), std::min(h->dist_verts.cap / sizeof(Vertex), h->vertices.cap / sizeof(Vertex))));
    return (uint32_t) std::min<size_t>(c, 0xFFFFFFFFu);
}

// the face bands of this rank's welded slab (enqueued behind whatever produced h->vertices; V is read on the device)
int dist_bands(sdmesh_handle *h, int levels, int R, int me, bool keep_status) {
    hipStream_t st = h->stream;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR;
    uint32_t lo = F * (uint32_t) me / (uint32_t) R, hi = F * (uint32_t) (me + 1) / (uint32_t) R;
    if (h->slab_nparts == R && h->slab_mode == SDMESH_PART_XSLAB) { lo = h->slab_bounds[me]; hi = h->slab_bounds[me + 1]; }
    // a projected vertex stays within about a voxel of its lattice edge: 2.5 voxels of the finest level is a safe band
    const float band = 2.5f * ((MESH_GENERATION_BB_SIZE / (float) F) / (float) (1u << levels));
    const float x_lo = -MESH_GENERATION_BB_SIZE / 2.0f + MESH_GENERATION_BB_SIZE * (float) lo / (float) F;
    const float x_hi = -MESH_GENERATION_BB_SIZE / 2.0f + MESH_GENERATION_BB_SIZE * (float) hi / (float) F;
    HIPCHK(hipMemsetAsync(h->d_ctr + CTR_DIST_NUP, 0, 4 * sizeof(uint32_t), st));
    if (!keep_status) SDCHK(zero_status(h));               // (planned route: the chain's own flags must reach the record)
    HIPCHK(launch_band_select(st, h->grid, (const Vertex *) h->vertices.p, h->d_ctr + CTR_NVERT, dist_local_capacity(h), x_hi - band, x_lo + band, me + 1 < R, me > 0,
                              (uint32_t *) h->dist_upids.p, (uint64_t *) h->dist_upkeys.p, (uint32_t *) h->dist_loids.p,
                              h->d_ctr + CTR_DIST_NUP, (uint32_t *) h->dist_dupof.p, h->d_status));
    return SDMESH_OK;
}

// synchronous route, phase 1: this rank's slab, meshed AND welded locally with its own read-back; then the face bands (enqueued, not waited for)
int dist_phase_local(sdmesh_handle *h, int levels, int R, int me, sdmesh_regen_info *slab) {
    if (h->inject_fail == 1) { h->inject_fail = 0; return fail(h, SDMESH_ERR_INTERNAL, "injected failure (option inject_fail)"); }
    SDCHK(regen_impl(h, levels, SDMESH_PART_XSLAB, me, R, slab));
    if (!h->welded || h->last_direct) return fail(h, SDMESH_ERR_INTERNAL, "the distributed weld needs the locally welded lattice path");
    h->dist_V = h->n_verts; h->dist_T = h->n_tris;
    SDCHK(dist_ensure_local(h, h->vertices.cap / sizeof(Vertex)));
    return dist_bands(h, levels, R, me, false);
}

// what this rank can take from its lower neighbour without growing anything: band vertices
uint32_t dist_prev_capacity(const sdmesh_handle *h) {
    if (!h->dist_prevkeys.p || !h->dist_prevg.p || !h->dist_hkeys.p || !h->dist_hpos.p) return 0;
    const size_t a = h->dist_prevkeys.cap / 8, b = h->dist_prevg.cap / 4;
    const size_t c = h->dist_hkeys.cap / 8 > 64 ? (h->dist_hkeys.cap / 8 - 64) / 2 : 0, d = h->dist_hpos.cap / 4 > 64 ? (h->dist_hpos.cap / 4 - 64) / 2 : 0;
    return (uint32_t) std::min<size_t>(std::min(std::min(a, b), std::min(c, d)), 0xFFFFFFFFu);
}

int dist_grow_prev(sdmesh_handle *h, uint32_t n_prev) {
    if (h->inject_fail == 2) { h->inject_fail = 0; return fail(h, SDMESH_ERR_HIP, "injected allocation failure (option inject_fail)"); }   // (set on this rank)
    const uint32_t n = std::max<uint32_t>(n_prev, 1);
    SDCHK(ensure_grow(h, h->dist_prevkeys, 8ull * n));
    SDCHK(ensure_grow(h, h->dist_prevg, 4ull * n));
    SDCHK(ensure_grow(h, h->dist_hkeys, 8ull * (2ull * n + 64)));
    SDCHK(ensure_grow(h, h->dist_hpos, 4ull * (2ull * n + 64)));
    return SDMESH_OK;
}

// phase 2: match my lower band against the lower neighbour's upper-band keys (already in dist_prevkeys; their count at *n_prev_ptr, at most
// cap_prev), number what I keep
int dist_phase_match(sdmesh_handle *h, const uint32_t *n_prev_ptr, uint32_t cap_prev, const DistRecord &rec = DistRecord{}) {
    hipStream_t st = h->stream;
    if (cap_prev) {
        const uint32_t hsize = 2u * cap_prev + 64u;              // (dist_grow_prev made room)
        HIPCHK(hipMemsetAsync(h->dist_hkeys.p, 0xFF, 8ull * hsize, st));
        HIPCHK(launch_dist_match(st, h->grid, (const uint64_t *) h->dist_prevkeys.p, n_prev_ptr, cap_prev, (uint64_t *) h->dist_hkeys.p, (uint32_t *) h->dist_hpos.p,
                                 hsize, (const Vertex *) h->vertices.p, (const uint32_t *) h->dist_loids.p, h->d_ctr + CTR_DIST_NLO,
                                 (uint32_t *) h->dist_dupof.p, h->d_status));
    }
    HIPCHK(launch_dist_number(st, h->tgrid, (const uint32_t *) h->dist_dupof.p, h->d_ctr + CTR_NVERT, dist_local_capacity(h), (uint32_t *) h->dist_tilecnt.p,
                              h->d_ctr + CTR_DIST_NKEEP, (uint32_t *) h->dist_newidx.p, h->d_status, rec));
    return SDMESH_OK;
}

// phase 3: the global numbers of my upper band (dist_counts holds every rank's kept count; the band's size at *n_up_ptr, at most cap_up)
int dist_phase_pack(sdmesh_handle *h, int me, const uint32_t *n_up_ptr, uint32_t cap_up) {
    if (cap_up)
        HIPCHK(launch_dist_pack_up(h->stream, h->grid, (const uint32_t *) h->dist_upids.p, n_up_ptr, cap_up, (const uint32_t *) h->dist_newidx.p,
                                   (const uint32_t *) h->dist_counts.p, (uint32_t) me, (uint32_t *) h->dist_upg.p));
    return SDMESH_OK;
}

// phase 4: kept vertices compacted, indices global (dist_prevg holds the lower neighbour's numbers of its upper band)
int dist_phase_final(sdmesh_handle *h, int me, uint32_t cap_prev, const DistPublish &pb = DistPublish{}) {
    HIPCHK(launch_dist_finalize(h->stream, h->grid, (const Vertex *) h->vertices.p, h->d_ctr + CTR_NVERT, dist_local_capacity(h), h->d_ctr + CTR_NTRI,
                                cap_items(h->indices, sizeof(IndexTriangle)), (const uint32_t *) h->dist_dupof.p, (const uint32_t *) h->dist_newidx.p,
                                (const uint32_t *) h->dist_prevg.p, cap_prev, (const uint32_t *) h->dist_counts.p, (uint32_t) me, (uint32_t *) h->indices.p,
                                (Vertex *) h->dist_verts.p, pb));
    return SDMESH_OK;
}

int dist_finish_host(sdmesh_handle *h, int R, int me, const uint32_t *counts /* kept vertices per rank */, sdmesh_regen_info *info,
                     const sdmesh_regen_info &slab, uint32_t *first_vertex) {
    uint32_t off = 0;
    for (int r = 0; r < me; r++) off += counts[r];
    h->n_verts = counts[me];
    h->mesh_valid = true;
    // the compacted list is the handle's vertex buffer now, the indices were rewritten in place (set here, behind the end of the slab's own
    // regen, which clears the result pointers)
    h->res_v = (Vertex *) h->dist_verts.p;
    h->res_i = nullptr;
    if (info) { *info = slab; info->vertices = counts[me]; }
    if (first_vertex) *first_vertex = off;
    (void) R;
    return SDMESH_OK;
}

// the device status word of a rank's record -> this rank's return code
int dist_record_error(sdmesh_handle *h, int r, int me, uint32_t st) {
    if (!st) return SDMESH_OK;
    if (st == 0x100u) return fail(h, r == me ? SDMESH_ERR_WELD_RANGE : SDMESH_ERR_PEER, "rank %d: the distributed weld needs vertices inside the packed key range", r);
    return fail(h, r == me ? SDMESH_ERR_INTERNAL : SDMESH_ERR_PEER, "rank %d: device status 0x%x in the distributed weld", r, st);
}

int dist_step_sync(sdmesh_handle *h, int levels, sdmesh_regen_info *info, uint32_t *first_vertex) {
    const int R = h->comm_nranks, me = h->comm_rank;
    hipStream_t st = h->stream;
    h->dsp.valid = false;
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    const int inject = h->inject_fail;                         // (2 = "my buffer growth fails": every rank must learn it, or only this one enters the agreement below)
    const int lrc = dist_phase_local(h, levels, R, me, &slab);
    uint32_t counts[32] = {0};
    // (a one-rank communicator takes the same route: its collectives are degenerate, every phase still runs)
    // 1. band sizes of every rank (message lengths must be host values here), each with the rank's status and what it can take from its
    //    lower neighbour without growing -- ALWAYS entered
    uint32_t *d_nups = (uint32_t *) h->dist_nups.p, *d_counts = (uint32_t *) h->dist_counts.p;
    HIPCHK(launch_dist_record(st, d_nups + (size_t) me * DIST_REC, h->d_ctr + CTR_DIST_NUP, nullptr, (uint32_t) lrc, dist_prev_capacity(h), (uint32_t) inject, 0u));
    SDCHK(tp_all_gather(h, d_nups + (size_t) me * DIST_REC, d_nups, DIST_REC, ncclUint32, st));        // in place
    uint32_t *rec = h->h_counts;
    HIPCHK(hipMemcpyAsync(rec, d_nups, 4ull * (size_t) R * DIST_REC, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    SDCHK(peer_failure(h, R, me, rec, DIST_REC, 1, lrc));
    uint32_t nups[32] = {0}, cap_up[32] = {0};
    for (int r = 0; r < R; r++) { nups[r] = r + 1 < R ? rec[(size_t) r * DIST_REC] : 0u; cap_up[r] = r + 1 < R ? dist_padded(h, nups[r]) : 0u; }
    const uint32_t n_up = nups[me], n_prev = me > 0 ? nups[me - 1] : 0u;
    // 2. room for the neighbour's band (plan-sized already: the next step may be a planned one); grown behind a one-word agreement only if some
    //    rank's published capacity is too small
    bool any_grow = false;
    for (int r = 1; r < R; r++) any_grow = any_grow || cap_up[r - 1] > rec[(size_t) r * DIST_REC + 2];
    for (int r = 0; r < R; r++) any_grow = any_grow || rec[(size_t) r * DIST_REC + 2] == 0u;          // (a first call: nothing allocated yet)
    for (int r = 0; r < R; r++) any_grow = any_grow || rec[(size_t) r * DIST_REC + 3] == 2u;          // (an injected growth failure on ANY rank: the same decision everywhere)
    if (any_grow) {
        const int grc = dist_grow_prev(h, me > 0 ? cap_up[me - 1] : 0u);
        uint32_t worst = 0;
        SDCHK(agree_flag(h, grc == SDMESH_OK ? 0u : 1u, &worst));
        if (grc != SDMESH_OK) return grc;
        if (worst) return fail(h, SDMESH_ERR_PEER, "a rank could not grow its band buffers; every rank returns an error");
    }
    // 3. the plan of the next steps: the same on every rank
    sdmesh_handle::DistPlan &pl = h->dsp;
    pl.levels = levels; pl.R = R; pl.scene = h->scene; pl.slab_sig = dist_slab_sig(h, R);
    uint32_t sig = 2166136261u;
    for (int r = 0; r < R; r++) { pl.cap_up[r] = cap_up[r]; sig = (sig ^ cap_up[r]) * 16777619u; }
    pl.sig = sig | 1u;
    pl.valid = h->shard_planned != 0;
    {
        std::vector<ZsMsg> msgs;                               // (grouped; the group is closed on every path)
        if (n_up) msgs.push_back({h->dist_upkeys.p, n_up, ncclUint64, me + 1, true});
        if (n_prev) msgs.push_back({h->dist_prevkeys.p, n_prev, ncclUint64, me - 1, false});
        SDCHK(zs_exchange(h, msgs));
    }
    // (nothing below allocates: a failure from here on is a HIP / RCCL error, after which the communicator is gone anyway)
    SDCHK(dist_phase_match(h, me > 0 ? d_nups + (size_t) (me - 1) * DIST_REC : d_nups, n_prev));
    // 4. kept counts of every rank (stay on the device for the kernels) + each rank's device status: a vertex outside the
    //    packed key range on ONE rank fails the call on ALL of them
    HIPCHK(launch_dist_record(st, d_counts + (size_t) me * DIST_REC, h->d_ctr + CTR_DIST_NKEEP, h->d_status, 0u, 0u, 0u, 0u));
    SDCHK(tp_all_gather(h, d_counts + (size_t) me * DIST_REC, d_counts, DIST_REC, ncclUint32, st));
    SDCHK(dist_phase_pack(h, me, d_nups + (size_t) me * DIST_REC, n_up));
    {
        std::vector<ZsMsg> msgs;
        if (n_up) msgs.push_back({h->dist_upg.p, n_up, ncclUint32, me + 1, true});
        if (n_prev) msgs.push_back({h->dist_prevg.p, n_prev, ncclUint32, me - 1, false});
        SDCHK(zs_exchange(h, msgs));
    }
    SDCHK(dist_phase_final(h, me, n_prev));
    HIPCHK(hipMemcpyAsync(rec, d_counts, 4ull * (size_t) R * DIST_REC, hipMemcpyDeviceToHost, st));
    SDCHK(read_back(h));
    for (int r = 0; r < R; r++) {
        counts[r] = rec[(size_t) r * DIST_REC];
        SDCHK(dist_record_error(h, r, me, rec[(size_t) r * DIST_REC + 1]));
    }
    return dist_finish_host(h, R, me, counts, info, slab, first_vertex);
}

int dist_step_planned(sdmesh_handle *h, int levels, sdmesh_regen_info *info, uint32_t *first_vertex) {
    const int R = h->comm_nranks, me = h->comm_rank;
    hipStream_t st = h->stream;
    const sdmesh_handle::DistPlan &pl = h->dsp;
    const int inject = h->inject_fail;
    h->inject_fail = 0;
    const uint32_t cap_up = pl.cap_up[me], cap_prev = me > 0 ? pl.cap_up[me - 1] : 0u;
    // buffers first (nothing below may allocate once the chain is in flight)
    if (dist_prev_capacity(h) < cap_prev || (cap_prev && dist_prev_capacity(h) == 0u)) SDCHK(dist_grow_prev(h, cap_prev));
    // 1. the slab's chain from its capacity plan (a handle without one runs its regen synchronously here), then everything else, blind
    int lrc = inject == 1 ? fail(h, SDMESH_ERR_INTERNAL, "injected failure (option inject_fail)") : regen_impl(h, levels, SDMESH_PART_XSLAB, me, R, nullptr, 1);
    sdmesh_regen_info slab;
    memset(&slab, 0, sizeof slab);
    bool ended = false;
    if (lrc == SDMESH_OK && !h->pend_async) { lrc = regen_impl(h, 0, 0, 0, 1, &slab, 2); ended = true; }
    if (lrc == SDMESH_OK && ended && (h->last_direct || !h->welded || h->wide_active)) lrc = fail(h, SDMESH_ERR_INTERNAL, "a slab left the packed-key lattice path");
    if (lrc != SDMESH_OK && h->pend_active) { sdmesh_regen_info tmp; (void) regen_impl(h, 0, 0, 0, 1, &tmp, 2); }
    if (lrc == SDMESH_OK) lrc = dist_ensure_local(h, std::max<size_t>(h->vertices.cap / sizeof(Vertex), cap_up));     // (grow-only: allocates on the first planned step at most)
    uint32_t *d_nups = (uint32_t *) h->dist_nups.p, *d_counts = (uint32_t *) h->dist_counts.p;
    if (lrc == SDMESH_OK) lrc = dist_bands(h, levels, R, me, true);
    if (lrc == SDMESH_ERR_HIP) return lrc;
    const uint32_t seq = ++h->zs_seq;
    HIPCHK(launch_dist_record(st, d_nups + (size_t) me * DIST_REC, h->d_ctr + CTR_DIST_NUP, h->d_status, (uint32_t) lrc, dist_prev_capacity(h), (uint32_t) inject, pl.sig));
    SDCHK(tp_all_gather(h, d_nups + (size_t) me * DIST_REC, d_nups, DIST_REC, ncclUint32, st));        // in place
    {
        std::vector<ZsMsg> msgs;                               // plan-sized messages (the real counts are in the records)
        if (me + 1 < R && cap_up) msgs.push_back({h->dist_upkeys.p, cap_up, ncclUint64, me + 1, true});
        if (me > 0 && cap_prev) msgs.push_back({h->dist_prevkeys.p, cap_prev, ncclUint64, me - 1, false});
        SDCHK(zs_exchange(h, msgs));
    }
    // (this rank's record of the second all-gather and the publication of both tables ride in the numbering / finalize kernels; a rank whose
    //  local phase failed runs neither and uses the two one-thread kernels instead)
    if (lrc == SDMESH_OK) {
        const DistRecord rec = {d_counts + (size_t) me * DIST_REC, h->d_ctr + CTR_DIST_NKEEP, h->d_status, 0u, 0u, 0u, pl.sig};
        SDCHK(dist_phase_match(h, me > 0 ? d_nups + (size_t) (me - 1) * DIST_REC : d_nups, cap_prev, rec));
    } else {
        HIPCHK(launch_dist_record(st, d_counts + (size_t) me * DIST_REC, h->d_ctr + CTR_DIST_NKEEP, h->d_status, (uint32_t) lrc, 0u, 0u, pl.sig));
    }
    SDCHK(tp_all_gather(h, d_counts + (size_t) me * DIST_REC, d_counts, DIST_REC, ncclUint32, st));
    if (lrc == SDMESH_OK) SDCHK(dist_phase_pack(h, me, d_nups + (size_t) me * DIST_REC, me + 1 < R ? cap_up : 0u));
    {
        std::vector<ZsMsg> msgs;
        if (me + 1 < R && cap_up) msgs.push_back({h->dist_upg.p, cap_up, ncclUint32, me + 1, true});
        if (me > 0 && cap_prev) msgs.push_back({h->dist_prevg.p, cap_prev, ncclUint32, me - 1, false});
        SDCHK(zs_exchange(h, msgs));
    }
    if (lrc == SDMESH_OK) SDCHK(dist_phase_final(h, me, cap_prev, DistPublish{d_nups, d_counts, (uint32_t) R, seq, h->h_dist_pub}));
    else HIPCHK(launch_dist_publish(st, d_nups, d_counts, (uint32_t) R, seq, h->h_dist_pub));
    // the step's one host synchronisation
    HIPCHK(hipStreamSynchronize(st));
    int erc = SDMESH_OK;
    if (h->pend_active) erc = regen_impl(h, 0, 0, 0, 1, &slab, 2);           // the chain's own verdict (its record said so already if it overflowed)
    const uint32_t *pub = h->h_dist_pub;
    if (pub[0] != seq) return fail(h, SDMESH_ERR_INTERNAL, "the planned distributed step did not publish its records (step %u, found %u)", seq, pub[0]);
    const uint32_t *rn = pub + 8, *rc = pub + 8 + (size_t) R * DIST_REC;
    bool redo = erc != SDMESH_OK || lrc != SDMESH_OK;
    uint32_t counts[32] = {0};
    for (int r = 0; r < R; r++) {
        const uint32_t nup = r + 1 < R ? rn[(size_t) r * DIST_REC] : 0u;
        redo = redo || rn[(size_t) r * DIST_REC + 1] != 0u || rc[(size_t) r * DIST_REC + 1] != 0u || nup > pl.cap_up[r] || rn[(size_t) r * DIST_REC + 3] != 0u ||
               rn[(size_t) r * DIST_REC + 4] != pl.sig || rc[(size_t) r * DIST_REC + 4] != pl.sig;
        counts[r] = rc[(size_t) r * DIST_REC];
    }
    if (redo) return ZS_REDO;
    h->dist_V = h->n_verts; h->dist_T = h->n_tris;
    return dist_finish_host(h, R, me, counts, info, slab, first_vertex);
}

}  // namespace

extern "C" {

int sdmesh_distributed_regen(sdmesh_handle *h, int levels, sdmesh_regen_info *info, uint32_t *first_vertex) {
    if (!h) return SDMESH_ERR_INVALID;
    if (!h->has_comm()) return fail(h, SDMESH_ERR_INVALID, "no communicator (sdmesh_comm_init)");
    const int R = h->comm_nranks;
    SDCHK(dist_scratch(h));                                    // (before anything can fail on one rank only, see peer_failure)
    SDCHK(ensure_shard_host(h, R));                            // pinned landing zone of the records (>= 32 x DIST_REC words)
    const sdmesh_handle::DistPlan &pl = h->dsp;
    if (h->shard_planned && pl.valid && pl.levels == levels && pl.R == R && pl.scene == h->scene && pl.slab_sig == dist_slab_sig(h, R)) {
        const int inject = h->inject_fail;
        const int rc = dist_step_planned(h, levels, info, first_vertex);
        if (rc != ZS_REDO) { h->dist_last_route = 1; return rc; }
        h->dist_last_route = 3;
        h->inject_fail = inject;                               // (the repeat fails where the planned step was told to)
        h->mesh_valid = false;
        return dist_step_sync(h, levels, info, first_vertex);
    }
    h->dist_last_route = 2;
    return dist_step_sync(h, levels, info, first_vertex);
}

int sdmesh_distributed_route(const sdmesh_handle *h) { return h ? (int) h->dist_last_route : 0; }

// The same step for `nranks` handles of ONE process on one device, device-to-device copies standing in for the two
// neighbour messages and host arithmetic for the two all-gathers: every kernel and every phase of the distributed weld
// runs exactly as under RCCL -- the rehearsal the tests use on a one-GPU box (RCCL refuses two ranks on one device).
int sdmesh_distributed_regen_rehearsal(sdmesh_handle **hs, int nranks, int levels, sdmesh_regen_info *infos, uint32_t *first_vertex) {
    if (!hs || nranks < 1 || nranks > 32) return SDMESH_ERR_INVALID;
    for (int r = 0; r < nranks; r++) if (!hs[r]) return SDMESH_ERR_INVALID;
    const int R = nranks;
    std::vector<sdmesh_regen_info> slab((size_t) R);
    uint32_t nups[32] = {0}, counts[32] = {0};
    // (the agreement of the RCCL form, with the host standing in for the all-gather: every handle runs its local phase whatever
    //  the others did, then all of them fail together if one did)
    int worst = SDMESH_OK, worst_rank = -1;
    for (int r = 0; r < R; r++) {
        sdmesh_handle *h = hs[r];
        SDCHK(dist_scratch(h));
        const int lrc = dist_phase_local(h, levels, R, r, &slab[(size_t) r]);
        if (lrc != SDMESH_OK) { if (worst == SDMESH_OK) { worst = lrc; worst_rank = r; } continue; }
        HIPCHK(hipMemcpyAsync(&nups[r], h->d_ctr + CTR_DIST_NUP, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (worst != SDMESH_OK) {
        for (int r = 0; r < R; r++)
            if (r != worst_rank) (void) fail(hs[r], SDMESH_ERR_PEER, "rank %d failed in this collective call (status 0x%x); every rank returns an error", worst_rank, worst);
        return worst;
    }
    uint32_t nrecs[32 * DIST_REC] = {0};                  // (the band-size records every handle would hold after the first all-gather)
    for (int r = 0; r < R; r++) nrecs[(size_t) r * DIST_REC] = nups[r];
    for (int r = 0; r < R; r++) {
        sdmesh_handle *h = hs[r];
        h->dsp.valid = false;
        const uint32_t n_prev = r > 0 ? nups[r - 1] : 0u;
        if (n_prev > dist_prev_capacity(h) || dist_prev_capacity(h) == 0u) SDCHK(dist_grow_prev(h, n_prev));
        HIPCHK(hipMemcpyAsync(h->dist_nups.p, nrecs, 4ull * (size_t) R * DIST_REC, hipMemcpyHostToDevice, h->stream));
        if (n_prev) HIPCHK(hipMemcpyAsync(h->dist_prevkeys.p, hs[r - 1]->dist_upkeys.p, 8ull * n_prev, hipMemcpyDeviceToDevice, h->stream));
        SDCHK(dist_phase_match(h, (const uint32_t *) h->dist_nups.p + (size_t) (r > 0 ? r - 1 : 0) * DIST_REC, n_prev));
        HIPCHK(hipMemcpyAsync(&counts[r], h->d_ctr + CTR_DIST_NKEEP, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    uint32_t recs[32 * DIST_REC] = {0};
    for (int r = 0; r < R; r++) recs[(size_t) r * DIST_REC] = counts[r];
    for (int r = 0; r < R; r++) {
        sdmesh_handle *h = hs[r];
        HIPCHK(hipMemcpyAsync(h->dist_counts.p, recs, 4ull * (size_t) R * DIST_REC, hipMemcpyHostToDevice, h->stream));
        SDCHK(dist_phase_pack(h, r, (const uint32_t *) h->dist_nups.p + (size_t) r * DIST_REC, r + 1 < R ? nups[r] : 0u));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    for (int r = 0; r < R; r++) {
        sdmesh_handle *h = hs[r];
        const uint32_t n_prev = r > 0 ? nups[r - 1] : 0u;
        if (n_prev) HIPCHK(hipMemcpyAsync(h->dist_prevg.p, hs[r - 1]->dist_upg.p, 4ull * n_prev, hipMemcpyDeviceToDevice, h->stream));
        SDCHK(dist_phase_final(h, r, n_prev));
        SDCHK(read_back(h));
        if (h->h_pinned[CTR_WORDS + ST_WELD_RANGE]) return fail(h, SDMESH_ERR_WELD_RANGE, "the distributed weld needs vertices inside the packed key range");
        SDCHK(dist_finish_host(h, R, r, counts, infos ? &infos[r] : nullptr, slab[(size_t) r], first_vertex ? &first_vertex[r] : nullptr));
    }
    return SDMESH_OK;
}

}  // extern "C"
