"""Host-side mirror of the reference's `CudaHandler` / `CudaVoxelField` (src/cuda/mod.rs:25-346)
on top of the C ABI (include/sdmesh.h).  Same method names, argument meaning and behaviour:

    CudaHandler()                              CudaHandler::new                 mod.rs:49
    CudaHandler.create_cuda_voxel_field()      assoc. fn, no device needed      mod.rs:105
    handler.refine_voxel_field(field)          in place: voxels <- kept children, size <- size/2   mod.rs:124
    handler.voxel_field_to_mesh(field)         -> ObjData (positions, normals, index triples)       mod.rs:204

The reference panics on any driver failure; here failures raise SdMeshError.  There is no CPU
fallback: constructing a handler without a gfx950 GPU raises.
"""
import ctypes
import dataclasses
import math

import numpy as np

from . import _abi
from ._abi import MAX_LEVELS, PART_NONE, PART_XSLAB, PART_ZSLAB, SCENES, Mesh, Point, RegenInfo, VoxelField  # noqa: F401


class SdMeshError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"sdmesh error {code} ({_abi.ERRORS.get(code, '?')}): {text}")
        self.code = code


@dataclasses.dataclass
class CudaVoxelField:
    """`pub struct CudaVoxelField { pub voxels: Vec<Point>, pub voxel_size: Point }` (mod.rs:41-46).
    voxels: float32 [n, 3]; voxel_size: float32 [3]."""
    voxels: np.ndarray
    voxel_size: np.ndarray


@dataclasses.dataclass
class ObjData:
    """What the reference packs into obj::ObjData (mod.rs:303-326): `position`, `normal`, one texture
    coordinate [0, 0], and per polygon the IndexTuple (i, Some(0), Some(i)) -- position index ==
    normal index (asserted by obj_to_bevy_mesh, src/renderer/mod.rs:110-128)."""
    position: np.ndarray            # float32 [V, 3]
    normal: np.ndarray              # float32 [V, 3]
    indices: np.ndarray             # uint32  [T, 3]
    texture: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((1, 2), np.float32))

    def save(self, path):
        """Wavefront OBJ with the content the reference saves through obj-rs 0.10 (`obj.save`,
        src/renderer/mod.rs:204): every position as `v`, the single texture coordinate `vt 0 0`, every normal as
        `vn`, one object/group "default", and per triangle `f i/1/i j/1/j k/1/k` with 1-based indices, where the
        position index equals the normal index (src/cuda/mod.rs:316-321).  Floats are written with the shortest
        representation that round-trips f32, so loading the file restores the arrays bit for bit."""
        with open(path, "w") as f:
            for p in self.position:
                f.write("v %s %s %s\n" % tuple(_f32_str(c) for c in p))
            f.write("vt 0 0\n")
            for n in self.normal:
                f.write("vn %s %s %s\n" % tuple(_f32_str(c) for c in n))
            f.write("o default\ng default\n")
            for t in self.indices:
                a, b, c = int(t[0]) + 1, int(t[1]) + 1, int(t[2]) + 1
                f.write(f"f {a}/1/{a} {b}/1/{b} {c}/1/{c}\n")

    @staticmethod
    def load(path):
        """Inverse of save() (and a reader for obj files with the same v/vt/vn/f i/j/k layout)."""
        pos, nrm, idx = [], [], []
        with open(path) as f:
            for line in f:
                w = line.split()
                if not w:
                    continue
                if w[0] == "v":
                    pos.append([np.float32(x) for x in w[1:4]])
                elif w[0] == "vn":
                    nrm.append([np.float32(x) for x in w[1:4]])
                elif w[0] == "f":
                    tri = [c.split("/") for c in w[1:4]]
                    assert all(c[0] == c[2] for c in tri), "position index must equal normal index"
                    idx.append([int(c[0]) - 1 for c in tri])
        return ObjData(np.array(pos, np.float32).reshape(-1, 3), np.array(nrm, np.float32).reshape(-1, 3),
                       np.array(idx, np.uint32).reshape(-1, 3))


def _f32_str(x):
    """Shortest decimal string that parses back to the same f32."""
    x = np.float32(x)
    for prec in range(1, 10):
        t = "%.*g" % (prec, float(x))
        if np.float32(t) == x:
            return t
    return repr(float(x))


def _pt(a):
    return Point(float(a[0]), float(a[1]), float(a[2]))


class CudaHandler:
    def __init__(self, device=0, scene="sdobj", params=None, fast_math=False):
        """fast_math: use libsdmesh_fast.so, the opt-in build with the reference's own numeric mode (nvcc --use_fast_math):
        faster, within 1e-5 relative of the strict build, not bit-reproducible."""
        self._lib = _abi.load(fast_math)
        self.fast_math = bool(fast_math)
        hp = ctypes.c_void_p()
        rc = self._lib.sdmesh_create(int(device), ctypes.byref(hp))
        if rc != 0:
            raise SdMeshError(rc, "sdmesh_create failed (is a gfx950 GPU visible?)")
        self._h = hp
        self.comm = None
        self.device = int(device)
        self.set_option("fast_math", 1 if fast_math else 0)        # the library confirms its numeric mode
        self.set_scene(scene, params)

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdmesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SdMeshError(rc, self._lib.sdmesh_last_error(self._h).decode(errors="replace"))

    # -- configuration -------------------------------------------------------------------------
    def set_scene(self, scene, params=None):
        p = np.zeros(4, np.float32)
        n = 0
        if params is not None:
            a = np.asarray(params, np.float32).ravel()
            n = a.size
            p[:n] = a
        self.scene = scene
        self._chk(self._lib.sdmesh_set_scene(self._h, SCENES[scene] if isinstance(scene, str) else int(scene),
                                              p.ctypes.data, n))

    def set_program(self, program):
        """Runtime SDF program: [(op, args, combine[, k]), ...], e.g. the reference scene is
        [("box_skeleton", [0, 0, 0, 3, 1, .5, .1], "set"), ("sphere", [0, 0, 0, 1], "smooth_min", .5)]."""
        arr = _abi.pack_program(program)
        self.scene = "program"
        self._chk(self._lib.sdmesh_set_program(self._h, arr.ctypes.data, len(arr)))

    def set_program_motion(self, motion):
        """Per-instruction rates [n, 7]: the kernels see args + time * rates (evaluated on the host in f32).  None clears them."""
        if motion is None:
            self._chk(self._lib.sdmesh_set_program_motion(self._h, None, 0))
            return
        m = np.ascontiguousarray(motion, np.float32).reshape(-1, 7)
        self._chk(self._lib.sdmesh_set_program_motion(self._h, m.ctypes.data, m.shape[0]))

    def set_program_time(self, t):
        self._chk(self._lib.sdmesh_set_program_time(self._h, float(np.float32(t))))

    def set_animation_time(self, t):
        """sphere centre (0, 0.5*sin t, 0); sin is evaluated on the host."""
        self.set_scene("anim", [np.float32(0.5) * np.float32(math.sin(t))])

    def set_stream(self, hip_stream):
        """Run on a caller-owned hipStream_t handle (an int).  0 is the legacy default stream -- which is what
        torch.cuda.current_stream().cuda_stream is for torch's default stream; None returns to the handle's own stream."""
        if hip_stream is None:
            self._chk(self._lib.sdmesh_use_own_stream(self._h))
        else:
            self._chk(self._lib.sdmesh_set_stream(self._h, ctypes.c_void_p(int(hip_stream))))

    def set_option(self, name, value):
        self._chk(self._lib.sdmesh_set_option(self._h, name.encode(), int(value)))

    # -- the reference's methods ---------------------------------------------------------------
    @staticmethod
    def create_cuda_voxel_field():
        lib = _abi.load()
        vox = np.empty((_abi.MESH_GENERATION_INIT_FACTOR ** 3, 3), np.float32)
        size = np.empty(3, np.float32)
        rc = lib.sdmesh_create_voxel_field(vox.ctypes.data, size.ctypes.data)
        if rc != 0:
            raise SdMeshError(rc, "sdmesh_create_voxel_field")
        return CudaVoxelField(vox, size)

    def refine_voxel_field(self, field):
        vox = np.ascontiguousarray(field.voxels, np.float32).reshape(-1, 3)
        n = vox.shape[0]
        out = np.empty((max(8 * n, 1), 3), np.float32)
        n_out = ctypes.c_size_t(0)
        size_out = Point()
        self._chk(self._lib.sdmesh_refine_voxel_field(self._h, vox.ctypes.data, n, _pt(field.voxel_size),
                                                       out.ctypes.data, 8 * n, ctypes.byref(n_out),
                                                       ctypes.byref(size_out)))
        field.voxels = out[: n_out.value].copy()
        field.voxel_size = np.array([size_out.x, size_out.y, size_out.z], np.float32)

    def voxel_field_to_mesh(self, field):
        vox = np.ascontiguousarray(field.voxels, np.float32).reshape(-1, 3)
        nv, nt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._chk(self._lib.sdmesh_voxel_field_to_mesh(self._h, vox.ctypes.data, vox.shape[0], _pt(field.voxel_size),
                                                        None, 0, ctypes.byref(nv), None, 0, ctypes.byref(nt)))
        return self._download_obj(nv.value, nt.value)

    # -- kernel-level layouts ------------------------------------------------------------------
    def refine_voxel_field_raw(self, field):
        vox = np.ascontiguousarray(field.voxels, np.float32).reshape(-1, 3)
        out = np.empty((8 * vox.shape[0], 3), np.float32)
        self._chk(self._lib.sdmesh_refine_voxel_field_raw(self._h, vox.ctypes.data, vox.shape[0], _pt(field.voxel_size),
                                                           out.ctypes.data))
        return out

    def voxel_field_to_triangles_raw(self, field):
        vox = np.ascontiguousarray(field.voxels, np.float32).reshape(-1, 3)
        out = np.empty((5 * vox.shape[0], 3, 2, 3), np.float32)
        self._chk(self._lib.sdmesh_voxel_field_to_triangles_raw(self._h, vox.ctypes.data, vox.shape[0],
                                                                 _pt(field.voxel_size), out.ctypes.data))
        return out

    def weld_triangles(self, raw):
        raw = np.ascontiguousarray(raw, np.float32).reshape(-1, 3, 2, 3)
        nv, nt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._chk(self._lib.sdmesh_weld_triangles(self._h, raw.ctypes.data, raw.shape[0], None, 0, ctypes.byref(nv),
                                                   None, 0, ctypes.byref(nt)))
        return self._download_obj(nv.value, nt.value)

    # -- device-resident path ------------------------------------------------------------------
    def regen(self, levels, mode=PART_NONE, part=0, nparts=1):
        """create -> `levels` x refine -> mesh -> weld without leaving HBM.  Returns RegenInfo."""
        info = RegenInfo()
        self._chk(self._lib.sdmesh_regen_part(self._h, int(levels), int(mode), int(part), int(nparts),
                                               ctypes.byref(info)))
        return info

    def regen_begin(self, levels, mode=PART_NONE, part=0, nparts=1):
        """Enqueue a regen without waiting (pair with regen_end; keeps several handles in flight)."""
        self._chk(self._lib.sdmesh_regen_begin(self._h, int(levels), int(mode), int(part), int(nparts)))

    def regen_end(self):
        info = RegenInfo()
        self._chk(self._lib.sdmesh_regen_end(self._h, ctypes.byref(info)))
        return info

    def regen_dense(self, levels):
        """Same result as regen(levels) via the dense lattice (sample -> sign bits -> LDS-tiled classify)."""
        info = RegenInfo()
        self._chk(self._lib.sdmesh_regen_dense(self._h, int(levels), ctypes.byref(info)))
        return info

    def download_mesh(self, info=None):
        m = Mesh()
        nt = ctypes.c_uint32(0)
        self._chk(self._lib.sdmesh_device_mesh(self._h, ctypes.byref(m), ctypes.byref(nt)))
        return self._download_obj(m.vertex_count, nt.value)

    def download_voxels(self):
        f = VoxelField()
        self._chk(self._lib.sdmesh_device_voxels(self._h, ctypes.byref(f)))
        vox = np.empty((f.voxel_count, 3), np.float32)
        if f.voxel_count:
            self._chk(self._lib.sdmesh_download_voxels(self._h, vox.ctypes.data))
        return CudaVoxelField(vox, np.array([f.voxel_size.x, f.voxel_size.y, f.voxel_size.z], np.float32))

    def device_mesh(self):
        m = Mesh()
        nt = ctypes.c_uint32(0)
        self._chk(self._lib.sdmesh_device_mesh(self._h, ctypes.byref(m), ctypes.byref(nt)))
        return m, nt.value

    def render(self, width, height, position, forward, up, right, fov, screen=None, dev_ptr=None):
        """CudaHandler::render (src/cuda/mod.rs:348): ray-march preview of the current scene as an RGBA8 image
        [height, width, 4].  With dev_ptr the image is written to that device buffer and the kernel time (ms) returned."""
        from ._abi import CameraBuffer, GlobalsBuffer
        g = GlobalsBuffer()
        g.render_texture_size[0], g.render_texture_size[1] = int(width), int(height)
        sw, sh = screen or (float(width), float(height))
        g.render_screen_size[0], g.render_screen_size[1] = sw, sh
        c = CameraBuffer()
        for i in range(3):
            c.position[i], c.forward[i], c.up[i], c.right[i] = position[i], forward[i], up[i], right[i]
        c.fov = fov
        if dev_ptr is not None:
            ms = ctypes.c_float(0.0)
            self._chk(self._lib.sdmesh_render_device(self._h, ctypes.byref(g), ctypes.byref(c), ctypes.c_void_p(dev_ptr), ctypes.byref(ms)))
            return ms.value
        img = np.empty((int(height), int(width), 4), np.uint8)
        self._chk(self._lib.sdmesh_render(self._h, ctypes.byref(g), ctypes.byref(c), img.ctypes.data))
        return img

    def device_mesh_tensors(self, device):
        """Zero-copy torch views of the welded mesh in HBM: vertices [V, 6] float32 (position, normal) and triangles
        [T, 3] int32.  Valid until the next call that regenerates the mesh on this handle."""
        import torch
        m, nt = self.device_mesh()

        class _View:          # CUDA array interface (version 2) over a raw device pointer
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}

        nv = int(m.vertex_count)
        verts = (torch.as_tensor(_View(m.vertices, (nv, 6), "<f4"), device=device) if nv
                 else torch.empty((0, 6), dtype=torch.float32, device=device))
        tris = (torch.as_tensor(_View(m.triangles, (nt, 3), "<i4"), device=device) if nt
                else torch.empty((0, 3), dtype=torch.int32, device=device))
        return verts, tris

    def device_triangles_into(self, dev_ptr, cap):
        n = ctypes.c_size_t(0)
        self._chk(self._lib.sdmesh_device_triangles(self._h, ctypes.c_void_p(dev_ptr), cap, ctypes.byref(n)))
        return n.value

    def triangle_count(self):
        n = ctypes.c_size_t(0)
        self._chk(self._lib.sdmesh_device_triangles(self._h, None, 0, ctypes.byref(n)))
        return n.value

    def column_triangle_counts_into(self, dev_ptr):
        self._chk(self._lib.sdmesh_column_triangle_counts(self._h, ctypes.c_void_p(dev_ptr)))

    # -- one mesh over several GPUs -----------------------------------------------
    def balance_slabs(self, mode, nparts, probe_levels=3):
        """Install slab boundaries that give every slab about the same share of the surface (sdmesh_balance_slabs);
        returns the nparts + 1 level-0 layer boundaries."""
        b = (ctypes.c_uint32 * (int(nparts) + 1))()
        self._chk(self._lib.sdmesh_balance_slabs(self._h, int(mode), int(nparts), int(probe_levels), b))
        return [int(x) for x in b]

    def set_slab_bounds(self, mode, nparts, bounds):
        if bounds is None:
            self._chk(self._lib.sdmesh_set_slab_bounds(self._h, int(mode), int(nparts), None))
        else:
            b = (ctypes.c_uint32 * len(bounds))(*[int(x) for x in bounds])
            self._chk(self._lib.sdmesh_set_slab_bounds(self._h, int(mode), int(nparts), b))

    def merge_by_column(self, list_ptrs, counts_ptr, out_ptr, cap):
        """Order-restoring merge (HIP kernel): list_ptrs = device pointers of the ranks' compact triangle lists, counts_ptr
        = device pointer of the [R, 1024] uint32 per-column counts; returns the merged triangle count."""
        arr = (ctypes.c_void_p * len(list_ptrs))(*[int(p) for p in list_ptrs])
        n = ctypes.c_size_t(0)
        self._chk(self._lib.sdmesh_merge_by_column(self._h, arr, ctypes.c_void_p(counts_ptr), len(list_ptrs),
                                                    ctypes.c_void_p(out_ptr), int(cap), ctypes.byref(n)))
        return n.value

    @staticmethod
    def comm_unique_id():
        """128 bytes (an ncclUniqueId) made on rank 0 and handed to every rank's comm_init."""
        buf = ctypes.create_string_buffer(128)
        rc = _abi.load().sdmesh_comm_unique_id(buf)
        if rc != 0:
            raise SdMeshError(rc, "sdmesh_comm_unique_id (is librccl.so.1 loadable?)")
        return buf.raw

    def comm_init(self, unique_id, nranks, rank):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._lib.sdmesh_comm_init(self._h, buf, int(nranks), int(rank)))
        self.comm = (int(nranks), int(rank))

    def comm_init_transport(self, nranks, rank, transport):
        """The caller's collectives instead of RCCL (sdmesh_comm_init_transport): `transport` has all_gather(send, recv, nbytes, stream),
        all_reduce_u32(send, recv, count, op, stream), send(buf, nbytes, peer, stream), recv(buf, nbytes, peer, stream), group_start(),
        group_end() -- device addresses and the stream as ints.  An exception inside one of them fails the step (SDMESH_ERR_RCCL) and is
        kept in `transport_error`."""
        T = _abi.Transport
        self.transport_error = None

        def guarded(fn):
            def call(_user, *args):
                try:
                    fn(*[0 if a is None else a for a in args])
                    return 0
                except BaseException as e:                  # (an exception cannot cross the C frames)
                    self.transport_error = e
                    return 1
            return call
        t = T(None, T.AG(guarded(transport.all_gather)), T.AR(guarded(transport.all_reduce_u32)), T.SR(guarded(transport.send)),
              T.SR(guarded(transport.recv)), T.GR(guarded(transport.group_start)), T.GR(guarded(transport.group_end)))
        self._chk(self._lib.sdmesh_comm_init_transport(self._h, int(nranks), int(rank), ctypes.byref(t)))
        self._transport = (t, transport)                    # the callbacks live as long as the communicator
        self.comm = (int(nranks), int(rank))

    def comm_destroy(self):
        self._chk(self._lib.sdmesh_comm_destroy(self._h))
        self.comm = None
        self._transport = None

    def ensure_comm(self, dist, nranks, rank, staged_device=None):
        """The handle's RCCL communicator over the ranks of the current torch.distributed job (made once per handle; the
        process group only carries the 128-byte unique id).  With `staged_device`: the caller's collectives instead, host-staged through
        the process group itself (partition.StagedTransport)."""
        if getattr(self, "comm", None) == (int(nranks), int(rank)):
            return
        if getattr(self, "comm", None) is not None:
            self.comm_destroy()
        if staged_device is not None:
            from .partition import StagedTransport
            self.comm_init_transport(nranks, rank, StagedTransport(dist, staged_device))
            return
        uid = [CudaHandler.comm_unique_id() if rank == 0 else None]
        if nranks > 1:
            dist.broadcast_object_list(uid, src=0)
        self.comm_init(uid[0], nranks, rank)

    def comm_selftest(self):
        """Every operation of the collective layer once with this rank as its own peer (sdmesh_comm_selftest): words that did not arrive."""
        bad = ctypes.c_uint64(0)
        self._chk(self._lib.sdmesh_comm_selftest(self._h, ctypes.byref(bad)))
        return int(bad.value)

    def comm_ranks(self):
        n = ctypes.c_int(0)
        self._chk(self._lib.sdmesh_comm_ranks(self._h, ctypes.byref(n)))
        return n.value

    def sharded_regen(self, levels, root=0):
        """ONE mesh over all ranks of the handle's RCCL communicator (z-slabs, gather to root, merge, weld) in one call."""
        info = RegenInfo()
        self._chk(self._lib.sdmesh_sharded_regen(self._h, int(levels), int(root), ctypes.byref(info)))
        return info

    ZSLAB_ROW_WORDS = 2056

    def zslab_local(self, levels, part, nparts, row_dev_ptr):
        """This rank's half of the indexed z-slab step with a caller-provided transport: slab regen + weld, row -> device buffer."""
        info = RegenInfo()
        self._chk(self._lib.sdmesh_zslab_local(self._h, int(levels), int(part), int(nparts), ctypes.c_void_p(row_dev_ptr), ctypes.byref(info)))
        return info

    def zslab_merge(self, levels, rows_dev_ptr, vert_ptrs, tri_ptrs):
        """The root's half: rows of all ranks + their vertex / index buffers (device pointers on this GPU) -> the single-GPU mesh."""
        n = len(vert_ptrs)
        va = (ctypes.c_void_p * n)(*[ctypes.c_void_p(p) for p in vert_ptrs])
        ta = (ctypes.c_void_p * n)(*[ctypes.c_void_p(p) for p in tri_ptrs])
        info = RegenInfo()
        self._chk(self._lib.sdmesh_zslab_merge(self._h, n, int(levels), ctypes.c_void_p(rows_dev_ptr), va, ta, ctypes.byref(info)))
        return info

    def sharded_route(self):
        """Route of the last indexed sharded step on this handle: 1 planned, 2 synchronous, 3 planned then repeated synchronously."""
        return int(self._lib.sdmesh_sharded_route(self._h))

    def distributed_route(self):
        """Route of the last sdmesh_distributed_regen on this handle: 1 planned, 2 synchronous, 3 planned then repeated synchronously."""
        return int(self._lib.sdmesh_distributed_route(self._h))

    def distributed_regen(self, levels):
        """ONE mesh over all ranks of the handle's communicator with the distributed weld (x-slabs): returns (RegenInfo of
        this rank's part, global index of its first kept vertex); download_mesh() then gives the part with global indices."""
        info = RegenInfo()
        first = ctypes.c_uint32(0)
        self._chk(self._lib.sdmesh_distributed_regen(self._h, int(levels), ctypes.byref(info), ctypes.byref(first)))
        return info, int(first.value)

    @staticmethod
    def distributed_regen_rehearsal(handlers, levels):
        """The distributed weld for several handlers of THIS process on one GPU (copies instead of RCCL messages)."""
        n = len(handlers)
        arr = (ctypes.c_void_p * n)(*[h._h for h in handlers])
        infos = (RegenInfo * n)()
        first = (ctypes.c_uint32 * n)()
        rc = handlers[0]._lib.sdmesh_distributed_regen_rehearsal(arr, n, int(levels), infos, first)
        if rc != 0:
            msgs = [h._lib.sdmesh_last_error(h._h).decode(errors="replace") for h in handlers]
            raise SdMeshError(rc, " | ".join(m for m in msgs if m))
        return [infos[i] for i in range(n)], [int(first[i]) for i in range(n)]

    @staticmethod
    def sharded_regen_rehearsal(handlers, levels, root=0):
        """The indexed z-slab gather for several handlers of THIS process on one GPU (copies instead of RCCL messages): handler r meshes
        and welds z-slab r, handlers[root] merges the slabs' vertex / index buffers into the single-GPU mesh, which becomes ITS mesh."""
        n = len(handlers)
        arr = (ctypes.c_void_p * n)(*[h._h for h in handlers])
        info = RegenInfo()
        rc = handlers[0]._lib.sdmesh_sharded_regen_rehearsal(arr, n, int(levels), int(root), ctypes.byref(info))
        if rc != 0:
            msgs = [h._lib.sdmesh_last_error(h._h).decode(errors="replace") for h in handlers]
            raise SdMeshError(rc, " | ".join(m for m in msgs if m))
        return info

    def weld_device_triangles(self, dev_ptr, nslots):
        info = RegenInfo()
        self._chk(self._lib.sdmesh_weld_device_triangles(self._h, ctypes.c_void_p(dev_ptr), int(nslots),
                                                          ctypes.byref(info)))
        return info

    def profile_stages(self):
        """[(stage name, device ms)] of the last planned regen run with option "profile" = 1."""
        n = ctypes.c_int(0)
        self._chk(self._lib.sdmesh_profile_stages(self._h, None, 0, None, 0, ctypes.byref(n)))
        if n.value == 0:
            return []
        names = ctypes.create_string_buffer(64 * n.value)
        ms = (ctypes.c_float * n.value)()
        self._chk(self._lib.sdmesh_profile_stages(self._h, names, len(names), ms, n.value, ctypes.byref(n)))
        return list(zip(names.value.decode().split(","), [float(x) for x in ms]))

    def projection_rounds(self):
        """Newton rounds (vertex x round, 13 SDF evaluations each) of the projection kernel in the last profiled regen."""
        n = ctypes.c_uint64(0)
        self._chk(self._lib.sdmesh_profile_rounds(self._h, ctypes.byref(n)))
        return int(n.value)

    def selfcheck_arithmetic(self):
        """Mismatch counts of the library's lean exact arithmetic (sqrt_rn_normal, sqrt_rn, inv_length_rn, third_rn, packed sqrt)
        against hipcc's correctly rounded expansions over all 2^32 inputs each, on this device (about a second)."""
        out = (ctypes.c_uint64 * 5)()
        self._chk(self._lib.sdmesh_selfcheck_arithmetic(self._h, out, 5))
        return [int(v) for v in out]

    def project_diag(self):
        """uint64 [waves, 4]: {start, end, rounds, hw id} per wave of the projection kernel (option "project_diag")."""
        n = ctypes.c_size_t(0)
        self._chk(self._lib.sdmesh_project_diag(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, 4), np.uint64)
        if n.value:
            self._chk(self._lib.sdmesh_project_diag(self._h, out.ctypes.data, n.value, ctypes.byref(n)))
        return out

    def sample_dense(self, N, dev_ptr):
        ms = ctypes.c_float(0)
        self._chk(self._lib.sdmesh_sample_dense(self._h, int(N), ctypes.c_void_p(dev_ptr), ctypes.byref(ms)))
        return ms.value

    def sample_dense_slab(self, N, x_begin, x_end, dev_ptr, iters=1, timed=True):
        """P1 of the x planes [x_begin, x_end) into a device buffer of (x_end - x_begin) (N+1)^2 floats; ms per launch when timed,
        else the launches are only enqueued on the handle's stream (None)."""
        ms = ctypes.c_float(0)
        self._chk(self._lib.sdmesh_sample_dense_slab(self._h, int(N), int(x_begin), int(x_end), ctypes.c_void_p(dev_ptr), int(iters),
                                                     ctypes.byref(ms) if timed else None))
        return ms.value if timed else None

    def time_kernel(self, which, N, dev_ptr, iters):
        ms = ctypes.c_float(0)
        self._chk(self._lib.sdmesh_time_kernel(self._h, int(which), int(N), ctypes.c_void_p(dev_ptr), int(iters),
                                                ctypes.byref(ms)))
        return ms.value

    def _download_obj(self, nv, nt):
        verts = np.empty((max(nv, 1), 2, 3), np.float32)
        idx = np.empty((max(nt, 1), 3), np.uint32)
        if nv or nt:
            self._chk(self._lib.sdmesh_download_mesh(self._h, verts.ctypes.data, idx.ctypes.data))
        return ObjData(verts[:nv, 0].copy(), verts[:nv, 1].copy(), idx[:nt].copy())
