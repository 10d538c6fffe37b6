"""ctypes binding of libnislam_kcc_hip.so (C ABI: include/nislam_kcc.h).

Python is only the test / bench harness language here; the product is the shared library.  The
binding mirrors the reference's CorrelationFlow interface (include/correlation_flow.h:8-33):
``CorrelationFlow(cfg, H, W)``, ``ComputeIntermedium``, ``ComputePose``.  There is NO CPU fallback:
if the library or a HIP device is missing every entry point raises.

Array conventions: a reference (Eigen, column-major) rows x cols array is
a C-order numpy array of shape (cols, rows); u8 images are (rows, cols).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NIK_LIB") or os.path.join(_HERE, "libnislam_kcc_hip.so")   # NIK_LIB: tuning variants

NIK_OK = 0
NIK_ERR_INVALID_ARG, NIK_ERR_UNSUPPORTED_SIZE, NIK_ERR_INVALID_KERNEL = -1, -2, -3
NIK_ERR_HIP, NIK_ERR_CAPACITY, NIK_ERR_NOT_READY = -4, -5, -6

EXPORTS = ["nik_create", "nik_destroy", "nik_last_error", "nik_get_dims", "nik_stream", "nik_synchronize",
           "nik_intermedium_u8", "nik_intermedium_f32", "nik_intermedium_batch_dev", "nik_frame_export",
           "nik_frame_import", "nik_pose", "nik_pose_batch", "nik_track_batch_dev", "nik_match",
           "nik_dbg_fft", "nik_dbg_ifft", "nik_dbg_rotate", "nik_dbg_polar", "nik_dbg_response",
           "nik_profile_enable", "nik_profile_read", "nik_set_streams", "nik_set_chunk",
           "nik_match_topk", "nik_rgb_to_gray_dev", "nik_set_kzz_cache", "nik_camera_maps", "nik_set_undistort", "nik_undistort_dev", "nik_tracker_create", "nik_tracker_destroy", "nik_is_generic", "nik_host_fft_plan", "nik_tracker_push_dev", "nik_tracker_push_u8", "nik_tracker_push_host", "nik_tracker_prefetch_dev", "nik_upload_u8_async", "nik_upload_fence", "nik_upload_wait", "nik_upload_after_compute", "nik_group_rccl_library", "nik_dev_malloc", "nik_dev_free", "nik_tracker_keyframes",
           "nik_tracker_attach_map", "nik_tracker_loops", "nik_pose_graph_optimize", "nik_pose_graph_optimize_dev", "nik_pose_graph_linearize", "nik_pg_shard_create", "nik_pg_shard_destroy", "nik_pg_shard_device", "nik_pg_shard_cost_dev", "nik_group_pose_graph_cost", "nik_stitcher_create", "nik_stitcher_destroy", "nik_stitcher_insert_dev", "nik_stitcher_recompute",
           "nik_stitcher_cells", "nik_stitcher_read_cell", "nik_pose_batch_window", "nik_downsample_u8_dev", "nik_pyramid_create", "nik_pyramid_destroy",
           "nik_pyramid_levels", "nik_pyramid_track_dev", "nik_pyramid_track_dev_async", "nik_pyramid_synchronize", "nik_pyramid_last_error",
           "nik_downsample_u8_stream", "nik_downsample_pyr_u8_stream", "nik_stream_wait_ctx", "nik_ctx_wait_stream", "nik_set_call_depth", "nik_pose_batch_async", "nik_wait_results", "nik_set_lane_rotation", "nik_map_create", "nik_map_destroy", "nik_map_add_frame", "nik_map_size", "nik_map_candidates", "nik_map_find_loop",
           "nik_host_polar_plan", "nik_host_free", "nik_host_rot_terms", "nik_host_rot8_geom", "nik_device",
           "nik_set_residual_stats", "nik_residual_stats_dev", "nik_residual_stats", "nik_tracker_pending_loops", "nik_tracker_speculation", "nik_tracker_stats", "nik_tracker_guess_gap", "nik_tracker_poses", "nik_tracker_edges", "nik_tracker_optimizations", "nik_map_update_poses", "nik_pose_batch_chained", "nik_wait_for", "nik_downsample_u8_async", "nik_set_graphs",
           "nik_group_last_error", "nik_group_unique_id", "nik_group_create_rank", "nik_group_create_local", "nik_group_destroy",
           "nik_group_world", "nik_group_local_count", "nik_group_ctx", "nik_group_rank", "nik_rgb_to_gray_async", "nik_group_shard", "nik_group_pick_best", "nik_group_comm_ranks",
           "nik_group_allreduce_residual", "nik_group_residual_result", "nik_group_gather_best", "nik_group_track_batch",
           "nik_group_match"]


class NikConfig(C.Structure):
    """mirrors CFConfig (reference include/read_configs.h:15-25)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("lambda_", C.c_float), ("kernel", C.c_int),
                ("sigma", C.c_float), ("offset", C.c_float), ("power", C.c_int),
                ("rotation_divisor", C.c_int), ("rotation_channel", C.c_int)]


class NikPoseResult(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("info", C.c_double * 3),
                ("rot_row", C.c_int32), ("rot_col", C.c_int32),
                ("trans_row", C.c_int32 * 2), ("trans_col", C.c_int32 * 2),
                ("psr_rot", C.c_float), ("psr_trans", C.c_float * 2),
                ("degree_final", C.c_float), ("chosen", C.c_int32), ("n_hyp", C.c_int32)]

    def as_dict(self):
        return dict(pose=list(self.pose), info=list(self.info), rot_row=self.rot_row, rot_col=self.rot_col,
                    trans_row=list(self.trans_row), trans_col=list(self.trans_col), psr_rot=float(self.psr_rot),
                    psr_trans=[float(v) for v in self.psr_trans], degree_final=float(self.degree_final),
                    chosen=self.chosen, n_hyp=self.n_hyp)


class NikTrackerConfig(C.Structure):
    """camera intrinsics after undistortion + KeyframeSelectionConfig (reference camera.cc:20-75, read_configs.h:27-32)"""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("height", C.c_double),
                ("extrinsics", C.c_double * 9), ("max_distance", C.c_double), ("max_angle", C.c_double),
                ("lower_response_thr", C.c_double), ("upper_response_thr", C.c_double)]


class NikTrackOutput(C.Structure):
    _fields_ = [("frame_id", C.c_int32), ("inserted", C.c_int32), ("good_tracking", C.c_int32), ("key_frame_id", C.c_int32),
                ("slot", C.c_int32), ("response", C.c_double * 3), ("cf_pose", C.c_double * 3), ("robot_pose", C.c_double * 3),
                ("distance", C.c_double), ("optimized", C.c_int32), ("reserved_", C.c_int32)]

    def as_dict(self):
        return dict(frame_id=self.frame_id, inserted=bool(self.inserted), good_tracking=bool(self.good_tracking),
                    key_frame_id=self.key_frame_id, slot=self.slot, response=list(self.response), cf_pose=list(self.cf_pose),
                    robot_pose=list(self.robot_pose), distance=self.distance, optimized=bool(self.optimized))


class NikLoopConfig(C.Structure):
    """mirrors nik_loop_config (MapConfig + LoopClosureConfig, read_configs.h:34-44)"""
    _fields_ = [("grid_scale", C.c_double), ("to_find_loop", C.c_int32), ("frame_gap_thr", C.c_int32), ("distance_thr", C.c_double),
                ("position_response_thr", C.c_double), ("angle_response_thr", C.c_double)]


class NikLoopResult(C.Structure):
    _fields_ = [("found", C.c_int32), ("cur_frame_id", C.c_int32), ("loop_frame_id", C.c_int32), ("loop_slot", C.c_int32),
                ("n_candidates", C.c_int32), ("response", C.c_double * 3), ("relative_pose", C.c_double * 3)]

    def as_dict(self):
        return dict(found=bool(self.found), cur_frame_id=self.cur_frame_id, loop_frame_id=self.loop_frame_id, loop_slot=self.loop_slot,
                    n_candidates=self.n_candidates, response=list(self.response), relative_pose=list(self.relative_pose))


class NikStageStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("ms", C.c_double), ("launches", C.c_int64), ("bytes", C.c_double), ("bytes_design", C.c_double)]


class NikError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nislam_kcc error %d: %s" % (code, msg))
        self.code = code


def default_config(kernel=0, rotation_divisor=720, rotation_channel=480, power=3):
    """values of reference configs/config_ntu.yaml:6-17"""
    return NikConfig(width=640, height=480, lambda_=0.1, kernel=kernel, sigma=0.2, offset=0.1, power=power,
                     rotation_divisor=rotation_divisor, rotation_channel=rotation_channel)


_lib = None


def load():
    """Load the shared library (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NikError(NIK_ERR_HIP, "libnislam_kcc_hip.so not built -- run __graft_entry__.build()")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (soname libamdhip64.so.7).
        # If torch is going to be used in this process (device buffers, torch.distributed) it has to be
        # loaded FIRST so that our DT_NEEDED libamdhip64.so.7 binds to the copy already in the process;
        # loading /opt/rocm's runtime first and torch's second leaves torch without a device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        P, I = C.c_void_p, C.c_int
        L.nik_create.argtypes = [C.POINTER(NikConfig), I, I, I, I, I, C.POINTER(P)]
        L.nik_destroy.argtypes = [P]
        L.nik_destroy.restype = None
        L.nik_last_error.argtypes = [P]
        L.nik_last_error.restype = C.c_char_p
        L.nik_get_dims.argtypes = [P, C.POINTER(I * 6)]
        L.nik_stream.argtypes = [P]
        L.nik_stream.restype = P
        L.nik_synchronize.argtypes = [P]
        L.nik_set_streams.argtypes = [P, I]
        L.nik_set_chunk.argtypes = [P, I]
        L.nik_set_kzz_cache.argtypes = [P, I]
        L.nik_camera_maps.argtypes = [P, P, I, I, P, P, P]
        L.nik_tracker_attach_map.argtypes = [P, P, I]
        L.nik_tracker_loops.argtypes = [P, P, I, P]
        L.nik_tracker_pending_loops.argtypes = [P]
        L.nik_tracker_speculation.argtypes = [P, P]
        L.nik_tracker_guess_gap.argtypes = [P, I]
        L.nik_tracker_poses.argtypes = [P, P, P, I, P]
        L.nik_tracker_edges.argtypes = [P, P, P, I, P]
        L.nik_tracker_optimizations.argtypes = [P, P]
        L.nik_map_update_poses.argtypes = [P, I, P, P]
        L.nik_pose_graph_optimize.argtypes = [I, P, P, I, P, I, P]
        L.nik_pose_graph_optimize_dev.argtypes = [I, I, P, P, I, P, I, P]
        L.nik_pose_graph_linearize.argtypes = [I, I, P, P, I, P, P, P, P]
        L.nik_pg_shard_create.argtypes = [I, I, P, P, I, P, P]
        L.nik_pg_shard_destroy.argtypes = [P]
        L.nik_pg_shard_destroy.restype = None
        L.nik_pg_shard_device.argtypes = [P]
        L.nik_pg_shard_cost_dev.argtypes = [P, P, P, P]
        L.nik_group_pose_graph_cost.argtypes = [P, P, P, P]
        L.nik_stitcher_create.argtypes = [P, I, P]
        L.nik_stitcher_destroy.argtypes = [P]
        L.nik_stitcher_destroy.restype = None
        L.nik_stitcher_insert_dev.argtypes = [P, I, P, P]
        L.nik_stitcher_recompute.argtypes = [P, I, P, P]
        L.nik_stitcher_cells.argtypes = [P, P, I, P]
        L.nik_stitcher_read_cell.argtypes = [P, I, I, P, P]
        L.nik_pose_batch_window.argtypes = [P, I, P, P, P, I, P]
        L.nik_downsample_u8_dev.argtypes = [P, I, P, P]
        L.nik_pyramid_create.argtypes = [P, I, I, I, I, I, P]
        L.nik_pyramid_destroy.argtypes = [P]
        L.nik_pyramid_destroy.restype = None
        L.nik_pyramid_levels.argtypes = [P, P]
        L.nik_pyramid_track_dev.argtypes = [P, I, P, P, I, P]
        L.nik_pyramid_track_dev_async.argtypes = [P, I, P, P, I, P]
        L.nik_pyramid_synchronize.argtypes = [P]
        L.nik_downsample_u8_stream.argtypes = [P, I, P, P, P]
        L.nik_downsample_pyr_u8_stream.argtypes = [P, I, I, P, I, P, P, P]
        L.nik_stream_wait_ctx.argtypes = [P, P]
        L.nik_ctx_wait_stream.argtypes = [P, P]
        L.nik_set_call_depth.argtypes = [P, I]
        L.nik_pyramid_last_error.argtypes = [P, I]
        L.nik_pyramid_last_error.restype = C.c_char_p
        L.nik_map_create.argtypes = [P, P, P]
        L.nik_map_destroy.argtypes = [P]
        L.nik_map_destroy.restype = None
        L.nik_map_add_frame.argtypes = [P, I, I, P, P]
        L.nik_map_size.argtypes = [P]
        L.nik_map_candidates.argtypes = [P, I, P, P, I, P]
        L.nik_map_find_loop.argtypes = [P, I, P, P]
        L.nik_set_undistort.argtypes = [P, P, P]
        L.nik_undistort_dev.argtypes = [P, I, P, P]
        L.nik_intermedium_u8.argtypes = [P, P, I, I]
        L.nik_intermedium_f32.argtypes = [P, P, I]
        L.nik_intermedium_batch_dev.argtypes = [P, I, P, P]
        L.nik_frame_export.argtypes = [P, I, P, P, P]
        L.nik_frame_import.argtypes = [P, I, P, P, P]
        L.nik_pose.argtypes = [P, I, I, I, P, P, P]
        L.nik_pose_batch.argtypes = [P, I, P, P, I, P]
        L.nik_pose_batch_async.argtypes = [P, I, P, P, I, P]
        L.nik_wait_results.argtypes = [P, P, I]
        L.nik_set_lane_rotation.argtypes = [P, I]
        L.nik_track_batch_dev.argtypes = [P, I, P, P, P, I, P, I]
        L.nik_match.argtypes = [P, I, I, P, P, P, P]
        L.nik_match_topk.argtypes = [P, I, I, P, I, P, P, P]
        L.nik_rgb_to_gray_dev.argtypes = [P, I, P, I, P]
        L.nik_rgb_to_gray_async.argtypes = [P, I, P, I, P]
        L.nik_dbg_fft.argtypes = [P, I, P, P]
        L.nik_dbg_ifft.argtypes = [P, I, P, P]
        L.nik_dbg_rotate.argtypes = [P, I, I, P]
        L.nik_dbg_polar.argtypes = [P, P, P]
        L.nik_dbg_response.argtypes = [P, I, I, I, I, P]
        L.nik_profile_enable.argtypes = [P, I]
        L.nik_tracker_create.argtypes = [P, C.POINTER(NikTrackerConfig), C.POINTER(P)]
        L.nik_tracker_destroy.argtypes = [P]
        L.nik_tracker_destroy.restype = None
        L.nik_tracker_push_dev.argtypes = [P, I, P, P]
        L.nik_tracker_push_u8.argtypes = [P, P, I, P]
        L.nik_tracker_push_host.argtypes = [P, I, P, I, C.c_size_t, P]
        L.nik_tracker_prefetch_dev.argtypes = [P, I, P]
        L.nik_upload_u8_async.argtypes = [P, I, P, I, C.c_size_t, P]
        L.nik_upload_fence.argtypes = [P, I]
        L.nik_upload_wait.argtypes = [P]
        L.nik_upload_after_compute.argtypes = [P]
        L.nik_dev_malloc.argtypes = [P, C.c_size_t, P]
        L.nik_dev_free.argtypes = [P, P]
        L.nik_tracker_keyframes.argtypes = [P, P, I, P]
        L.nik_profile_read.argtypes = [P, P, I, P]
        L.nik_set_graphs.argtypes = [P, I]
        L.nik_set_residual_stats.argtypes = [P, I]
        L.nik_residual_stats.argtypes = [P, P]
        L.nik_residual_stats_dev.argtypes = [P, P, P]
        L.nik_device.argtypes = [P]
        L.nik_group_last_error.restype = C.c_char_p
        L.nik_group_last_error.argtypes = [P]
        L.nik_group_unique_id.argtypes = [P]
        L.nik_group_create_rank.argtypes = [P, I, I, P, P]
        L.nik_group_create_local.argtypes = [C.POINTER(NikConfig), I, I, I, I, I, P, P]
        L.nik_group_destroy.argtypes = [P]
        L.nik_group_destroy.restype = None
        L.nik_group_world.argtypes = [P]
        L.nik_group_local_count.argtypes = [P]
        L.nik_group_ctx.argtypes = [P, I]
        L.nik_group_ctx.restype = P
        L.nik_group_rank.argtypes = [P, I]
        L.nik_group_shard.argtypes = [I, I, I, P, P]
        L.nik_group_shard.restype = None
        L.nik_group_pick_best.argtypes = [P, I]
        L.nik_group_comm_ranks.argtypes = [P]
        L.nik_group_rccl_library.restype = C.c_char_p
        L.nik_group_rccl_library.argtypes = [C.POINTER(C.c_int)]
        L.nik_group_allreduce_residual.argtypes = [P, P]
        L.nik_group_residual_result.argtypes = [P, P]
        L.nik_group_gather_best.argtypes = [P, P, P, P, P]
        L.nik_group_track_batch.argtypes = [P, I, P, P, P, I, P]
        L.nik_group_match.argtypes = [P, P, I, P, P, P, P, P]
        L.nik_host_polar_plan.argtypes = [I, I, I, I, P, P, P, P, P]
        L.nik_host_free.argtypes = [P]
        L.nik_host_free.restype = None
        L.nik_host_rot_terms.argtypes = [I, I, C.c_float, P]
        L.nik_host_rot8_geom.argtypes = [I, P]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _i32(seq):
    return np.ascontiguousarray(np.asarray(seq, dtype=np.int32))


def host_polar_plan(H, W, PD, PC):
    """The polar gather plan nik_create builds for this geometry (host only, no GPU): dict of numpy arrays."""
    L = load()
    dims = np.zeros(8, np.int32)
    ch, sf, pts = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nch = C.c_int()
    rc = L.nik_host_polar_plan(H, W, PD, PC, _p(dims), C.byref(ch), C.byref(nch), C.byref(sf), C.byref(pts))
    if rc:
        raise RuntimeError("nik_host_polar_plan: %d %s" % (rc, L.nik_last_error(None).decode()))
    qs, nseg, tiles, lines, threads, rf, mf, lds = (int(v) for v in dims)
    def take(ptr, n, dt):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,)).copy()
        L.nik_host_free(ptr)
        return a
    out = dict(qs=qs, nseg=nseg, tiles=tiles, lines=lines, threads=threads, rf=rf, mf=mf, lds_bytes=lds,
               chunks=take(ch, max(nch.value, 1), C.c_uint32)[:nch.value],
               seg_first=take(sf, tiles * nseg + 1, C.c_int32),
               pts=take(pts, tiles * rf * lines * threads * 4, C.c_uint32).reshape(tiles, rf, lines * threads, 4))
    return out


def host_rot_terms(H, W, degree):
    out = np.zeros(2 * W + 2 * H, np.int32)
    rc = load().nik_host_rot_terms(H, W, C.c_float(degree), _p(out))
    if rc:
        raise RuntimeError("nik_host_rot_terms: %d" % rc)
    return out[:W], out[W:2 * W], out[2 * W:2 * W + H], out[2 * W + H:]


def host_rot8_geom(H):
    g = np.zeros(5, np.int32)
    rc = load().nik_host_rot8_geom(H, _p(g))
    if rc:
        raise RuntimeError("nik_host_rot8_geom: %d" % rc)
    return dict(band_rows=int(g[0]), bands=int(g[1]), box_rows=int(g[2]), pitch=int(g[3]), lds_bytes=int(g[4]))


def camera_maps(K, D, W, H):
    """Camera::Camera's map construction (camera.cc:46-47) on the host: K = (fx, cx, fy, cy), D = (k1, k2, p1, p2, k3)
    -> new_K (4,), map1 int16 (H, W, 2), map2 uint16 (H, W).  Needs the library but no GPU."""
    K = np.ascontiguousarray(K, np.float64); D = np.ascontiguousarray(D, np.float64)
    newK = np.empty(4, np.float64); m1 = np.empty((H, W, 2), np.int16); m2 = np.empty((H, W), np.uint16)
    rc = load().nik_camera_maps(_p(K), _p(D), int(W), int(H), _p(newK), _p(m1), _p(m2))
    if rc:
        raise NikError(rc, "nik_camera_maps: invalid intrinsics / size")
    return newK, m1, m2


class CorrelationFlow:
    """MI355X CorrelationFlow.  Frames live in device slots (the analogue of reference Frame's spectra)."""

    def __init__(self, cfg, image_height, image_width, max_batch=8, max_frames=64, device=0):
        self._L = load()
        self._ctx = C.c_void_p()
        rc = self._L.nik_create(C.byref(cfg), int(image_height), int(image_width), int(max_batch), int(max_frames),
                                int(device), C.byref(self._ctx))
        if rc:
            self._ctx = None
            raise NikError(rc, self._L.nik_last_error(None).decode())
        self.cfg, self.H, self.W = cfg, int(image_height), int(image_width)
        self.PD, self.PC = cfg.rotation_divisor, cfg.rotation_channel
        self.max_batch, self.max_frames = max_batch, max_frames

    @classmethod
    def borrow(cls, ctx, cfg, image_height, image_width, max_batch, max_frames):
        """a view of a context owned by someone else (a local nik_group): close() leaves it alone"""
        self = cls.__new__(cls)
        self._L = load()
        self._ctx = C.c_void_p(ctx)
        self._borrowed = True
        self.cfg, self.H, self.W = cfg, int(image_height), int(image_width)
        self.PD, self.PC = cfg.rotation_divisor, cfg.rotation_channel
        self.max_batch, self.max_frames = max_batch, max_frames
        return self

    def close(self):
        if getattr(self, "_ctx", None):
            if not getattr(self, "_borrowed", False):
                self._L.nik_destroy(self._ctx)
            self._ctx = None

    def set_graphs(self, max_pairs):
        self._chk(self._L.nik_set_graphs(self._ctx, int(max_pairs)))

    # ---- residual statistics of the latest batch, reduced on the device ---------------------
    def set_residual_stats(self, on=True):
        self._chk(self._L.nik_set_residual_stats(self._ctx, int(bool(on))))

    def residual_stats(self):
        out = np.zeros(4, np.float64)
        self._chk(self._L.nik_residual_stats(self._ctx, _p(out)))
        return out

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._ctx).decode())

    @property
    def stream(self):
        return self._L.nik_stream(self._ctx)

    def synchronize(self):
        self._chk(self._L.nik_synchronize(self._ctx))

    def set_kzz_cache(self, on=True):
        self._chk(self._L.nik_set_kzz_cache(self._ctx, int(bool(on))))

    def set_streams(self, n):
        return self._L.nik_set_streams(self._ctx, int(n))

    def set_chunk(self, pairs):
        """batched calls are cut into chunks of at most `pairs` pairs (0: one chunk per stream); returns the previous value"""
        return self._L.nik_set_chunk(self._ctx, int(pairs))

    # ---- camera undistortion (Camera::Camera maps / Camera::UndistortImage, camera.cc:45-47,92-93) ----
    def set_undistort(self, map1=None, map2=None):
        """Install (or with None remove) the CV_16SC2 / CV_16UC1 maps: u8 entry points then take raw camera frames."""
        if map1 is None:
            self._chk(self._L.nik_set_undistort(self._ctx, None, None))
            return
        map1 = np.ascontiguousarray(map1, np.int16); map2 = np.ascontiguousarray(map2, np.uint16)
        assert map1.shape == (self.H, self.W, 2) and map2.shape == (self.H, self.W)
        self._chk(self._L.nik_set_undistort(self._ctx, _p(map1), _p(map2)))

    def undistort_dev(self, d_raw_ptr, n, d_out_ptr):
        self._chk(self._L.nik_undistort_dev(self._ctx, int(n), C.c_void_p(int(d_raw_ptr)), C.c_void_p(int(d_out_ptr))))

    # ---- ComputeIntermedium ------------------------------------------------------------------
    def intermedium_u8(self, gray, dst):
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.H, self.W)
        self._chk(self._L.nik_intermedium_u8(self._ctx, _p(gray), self.W, int(dst)))

    def intermedium_f32(self, image, dst):
        image = np.ascontiguousarray(image, np.float32)
        assert image.shape == (self.W, self.H)
        self._chk(self._L.nik_intermedium_f32(self._ctx, _p(image), int(dst)))

    def intermedium_batch_dev(self, d_gray_ptr, n, dst):
        dst = _i32(dst)
        self._chk(self._L.nik_intermedium_batch_dev(self._ctx, int(n), C.c_void_p(int(d_gray_ptr)), _p(dst)))

    def ComputeIntermedium(self, image, dst=0):
        """reference signature ComputeIntermedium(image, fft_result&, fft_polar&): returns the two spectra."""
        self.intermedium_f32(image, dst)
        _, f, p = self.frame_export(dst, image=False)
        return f, p

    def frame_export(self, f, image=True, spectra=True):
        img = np.empty((self.W, self.H), np.float32) if image else None
        fr = np.empty((self.W, self.H // 2 + 1), np.complex64) if spectra else None
        fp = np.empty((self.PC, self.PD // 2 + 1), np.complex64) if spectra else None
        self._chk(self._L.nik_frame_export(self._ctx, int(f), _p(img), _p(fr), _p(fp)))
        return img, fr, fp

    def frame_import(self, f, image=None, fft_result=None, fft_polar=None):
        image = None if image is None else np.ascontiguousarray(image, np.float32)
        fft_result = None if fft_result is None else np.ascontiguousarray(fft_result, np.complex64)
        fft_polar = None if fft_polar is None else np.ascontiguousarray(fft_polar, np.complex64)
        self._chk(self._L.nik_frame_import(self._ctx, int(f), _p(image), _p(fft_result), _p(fft_polar)))

    # ---- ComputePose -------------------------------------------------------------------------
    def pose(self, key, cur, not_large_rotation=True):
        pose, info, res = np.zeros(3), np.zeros(3), NikPoseResult()
        self._chk(self._L.nik_pose(self._ctx, int(key), int(cur), int(bool(not_large_rotation)), _p(pose), _p(info),
                                   C.addressof(res)))
        return pose, info, res.as_dict()

    def ComputePose(self, last_fft_result, image, last_fft_polar, fft_polar, not_large_rotation, key_slot=0, cur_slot=1):
        """reference signature: spectra/image passed by value (host arrays in the reference layouts)."""
        self.frame_import(key_slot, None, last_fft_result, last_fft_polar)
        # the current frame's fft_result is not an input of ComputePose; import a zero spectrum placeholder
        self.frame_import(cur_slot, image, np.zeros((self.W, self.H // 2 + 1), np.complex64), fft_polar)
        # the key slot needs an image flag too (never read by ComputePose)
        self.frame_import(key_slot, np.zeros((self.W, self.H), np.float32), None, None)
        return self.pose(key_slot, cur_slot, not_large_rotation)

    def pose_batch(self, keys, curs, not_large_rotation=True):
        keys, curs = _i32(keys), _i32(curs)
        n = len(keys)
        res = (NikPoseResult * n)()
        self._chk(self._L.nik_pose_batch(self._ctx, n, _p(keys), _p(curs), int(bool(not_large_rotation)),
                                         C.cast(res, C.c_void_p)))
        return [r.as_dict() for r in res]

    def pose_batch_async(self, keys, curs, not_large_rotation=True, res=None):
        """nik_pose_batch_async over stored frames: returns the (NikPoseResult * n) array, final after wait_results(res) or
        synchronize(); keep it alive until then"""
        n = len(keys)
        res = res if res is not None else (NikPoseResult * n)()
        k, c = _i32(keys), _i32(curs)
        self._chk(self._L.nik_pose_batch_async(self._ctx, n, _p(k), _p(c), int(bool(not_large_rotation)), C.cast(res, C.c_void_p)))
        keep = self.__dict__.setdefault("_inflight", [])
        keep.append(res)
        del keep[:-16]
        return res

    def wait_results(self, res):
        """nik_wait_results: the results of ONE asynchronous batch (later batches keep running)"""
        self._chk(self._L.nik_wait_results(self._ctx, C.cast(res, C.c_void_p), len(res)))

    def set_lane_rotation(self, on=True):
        self._chk(self._L.nik_set_lane_rotation(self._ctx, int(bool(on))))

    def track_batch_dev(self, d_gray_ptr, keys, cur_dst, not_large_rotation=True, sync=True, res=None):
        keys, cur_dst = _i32(keys), _i32(cur_dst)
        n = len(keys)
        if res is None:
            res = (NikPoseResult * n)()
        if not sync:
            # an asynchronous call finalises its results into `res` when a later call of the lane retires it (or at synchronize()):
            # the array must outlive the caller's interest in it -- keep the latest ones alive here
            keep = self.__dict__.setdefault("_inflight", [])
            keep.append(res)
            del keep[:-16]
        self._chk(self._L.nik_track_batch_dev(self._ctx, n, C.c_void_p(int(d_gray_ptr)), _p(keys), _p(cur_dst),
                                              int(bool(not_large_rotation)), C.cast(res, C.c_void_p), int(bool(sync))))
        return res

    def pose_batch_window(self, keys, curs, centers, radius):
        """ComputePose (small-rotation mode) with both arg-max searches restricted to windows; centers: (n, 4) int32."""
        keys, curs = _i32(keys), _i32(curs)
        n = len(keys)
        centers = np.ascontiguousarray(centers, np.int32).reshape(n, 4)
        res = (NikPoseResult * n)()
        self._chk(self._L.nik_pose_batch_window(self._ctx, n, _p(keys), _p(curs), _p(centers), int(radius), C.cast(res, C.c_void_p)))
        return res

    def downsample_u8_dev(self, d_in_ptr, n, d_out_ptr):
        self._chk(self._L.nik_downsample_u8_dev(self._ctx, int(n), C.c_void_p(int(d_in_ptr)), C.c_void_p(int(d_out_ptr))))

    def downsample_pyr_u8(self, steps, na, d_a_ptr, nb, d_b_ptr, out_ptrs, stream=None):
        """`steps` (1..3) pyramid levels in one launch (nik_downsample_pyr_u8_stream); out_ptrs: device pointers of the levels;
        returns the C return code (NIK_ERR_UNSUPPORTED_SIZE for unaligned sizes / pointers)"""
        outs = (C.c_void_p * 3)(*([int(x) for x in out_ptrs] + [0] * (3 - len(out_ptrs))))
        return self._L.nik_downsample_pyr_u8_stream(self._ctx, int(steps), int(na), C.c_void_p(int(d_a_ptr)), int(nb),
                                                    C.c_void_p(int(d_b_ptr)) if d_b_ptr else None, outs, C.c_void_p(int(stream)) if stream else None)

    def match(self, query, cands, raw=False):
        """raw: hand back the ctypes result array as it is (timing loops: converting thousands of results to dicts costs
        more than the search)"""
        cands = _i32(cands)
        n = len(cands)
        res = (NikPoseResult * max(n, 1))()
        best, best_res = C.c_int(-1), NikPoseResult()
        self._chk(self._L.nik_match(self._ctx, int(query), n, _p(cands), C.addressof(best), C.cast(res, C.c_void_p),
                                    C.addressof(best_res)))
        if raw:
            return best.value, res, best_res
        return best.value, [res[i].as_dict() for i in range(n)], best_res.as_dict()

    def match_topk(self, query, cands, k):
        cands = _i32(cands)
        n = len(cands)
        best, best_res = C.c_int(-1), NikPoseResult()
        short = np.full(max(1, min(k, n)), -1, np.int32)
        self._chk(self._L.nik_match_topk(self._ctx, int(query), n, _p(cands), int(k), C.addressof(best),
                                         C.addressof(best_res), _p(short)))
        return best.value, best_res.as_dict(), short.tolist()

    def rgb_to_gray_dev(self, d_rgb_ptr, n, d_gray_ptr, bgr=False):
        self._chk(self._L.nik_rgb_to_gray_dev(self._ctx, int(n), C.c_void_p(int(d_rgb_ptr)), int(bool(bgr)),
                                              C.c_void_p(int(d_gray_ptr))))

    def rgb_to_gray_async(self, d_rgb_ptr, n, d_gray_ptr, bgr=False):
        self._chk(self._L.nik_rgb_to_gray_async(self._ctx, int(n), C.c_void_p(int(d_rgb_ptr)), int(bool(bgr)),
                                                C.c_void_p(int(d_gray_ptr))))

    # ---- measurement -------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self._L.nik_profile_enable(self._ctx, int(bool(on))))

    def profile_read(self):
        out = (NikStageStat * 64)()
        n = C.c_int(0)
        self._chk(self._L.nik_profile_read(self._ctx, C.cast(out, C.c_void_p), 64, C.addressof(n)))
        return [dict(name=out[i].name.decode(), ms=out[i].ms, launches=out[i].launches, bytes=out[i].bytes, bytes_design=out[i].bytes_design)
                for i in range(min(n.value, 64))]

    # ---- debug taps --------------------------------------------------------------------------
    def dbg_fft(self, x, which=0):
        x = np.ascontiguousarray(x, np.float32)
        cols, rows = x.shape
        out = np.empty((cols, rows // 2 + 1), np.complex64)
        self._chk(self._L.nik_dbg_fft(self._ctx, int(which), _p(x), _p(out)))
        return out

    def dbg_ifft(self, xf, which=0):
        xf = np.ascontiguousarray(xf, np.complex64)
        cols, hr = xf.shape
        out = np.empty((cols, (hr - 1) * 2), np.float32)
        self._chk(self._L.nik_dbg_ifft(self._ctx, int(which), _p(xf), _p(out)))
        return out

    def dbg_rotate(self, slot, degree2):
        out = np.empty((self.W, self.H), np.float32)
        self._chk(self._L.nik_dbg_rotate(self._ctx, int(slot), int(degree2), _p(out)))
        return out

    def dbg_polar(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty((self.PC, self.PD), np.float32)
        self._chk(self._L.nik_dbg_polar(self._ctx, _p(x), _p(out)))
        return out

    def dbg_response(self, which, key, cur, degree2=0):
        """g = IFFT(G) of EstimateTrans: which 0 -> rotation surface (PC, PD), which 1 -> translation surface (W, H) of
        cur's image de-rotated by degree2/2 degrees"""
        out = np.empty((self.PC, self.PD) if which == 0 else (self.W, self.H), np.float32)
        self._chk(self._L.nik_dbg_response(self._ctx, int(which), int(key), int(cur), int(degree2), _p(out)))
        return out


class Group:
    """nik_group: the contexts of several GPUs plus the RCCL communicator for the residual all-reduce and the
    best-candidate all-gather.  Group.local(...) drives every GPU from this process; Group.rank(cf, rank, world, id) is the
    one-process-per-GPU form (id = Group.unique_id() of rank 0, broadcast by the launcher)."""

    def __init__(self, handle, flows):
        self._L = load()
        self._g = handle
        self.flows = flows

    @staticmethod
    def unique_id():
        buf = np.zeros(128, np.uint8)
        L = load()
        rc = L.nik_group_unique_id(_p(buf))
        if rc:
            raise NikError(rc, L.nik_group_last_error(None).decode())
        return buf

    @classmethod
    def local(cls, cfg, image_height, image_width, max_batch=8, max_frames=64, devices=(0,)):
        L = load()
        g = C.c_void_p()
        dv = _i32(devices)
        rc = L.nik_group_create_local(C.byref(cfg), int(image_height), int(image_width), int(max_batch), int(max_frames), len(dv), _p(dv), C.byref(g))
        if rc:
            raise NikError(rc, L.nik_group_last_error(None).decode())
        flows = [CorrelationFlow.borrow(L.nik_group_ctx(g, i), cfg, image_height, image_width, max_batch, max_frames) for i in range(len(dv))]
        return cls(g, flows)

    @classmethod
    def rank(cls, flow, rank, world, uid=None):
        L = load()
        g = C.c_void_p()
        rc = L.nik_group_create_rank(flow._ctx, int(rank), int(world), _p(uid) if uid is not None else None, C.byref(g))
        if rc:
            raise NikError(rc, L.nik_group_last_error(None).decode())
        return cls(g, [flow])

    def _chk(self, rc):
        if rc:
            raise NikError(rc, self._L.nik_group_last_error(self._g).decode())

    @property
    def world(self):
        return self._L.nik_group_world(self._g)

    @staticmethod
    def shard(n, world, rank):
        b, e = C.c_int(), C.c_int()
        load().nik_group_shard(int(n), int(world), int(rank), C.byref(b), C.byref(e))
        return b.value, e.value

    @staticmethod
    def pick_best(records):
        """records: [world][8] doubles (score, global index, pose x3, info x3) -> winning rank by the reference's rule (-1: none)"""
        r = np.ascontiguousarray(records, np.float64).reshape(-1, 8)
        return load().nik_group_pick_best(_p(r), int(r.shape[0]))

    @staticmethod
    def rccl_library():
        """(path of the RCCL the library bound or None, True if it is the copy the host process -- PyTorch -- had already loaded)"""
        sh = C.c_int(0)
        p = load().nik_group_rccl_library(C.byref(sh))
        return (p.decode() if p else None), bool(sh.value)

    def comm_ranks(self):
        """ranks the RCCL communicator spans (ncclCommCount); 0 = no RCCL in use"""
        return self._L.nik_group_comm_ranks(self._g)

    def pose_graph_cost(self, shards, poses=None):
        """0.5 sum |r|^2 of a pose graph whose constraints are sharded over the group's GPUs: reduced per device, one double
        all-reduced (shards: one PgShard per local member)"""
        arr = (C.c_void_p * len(shards))(*[s._s for s in shards])
        x = None if poses is None else np.ascontiguousarray(np.array(poses, np.float64))
        cost = C.c_double(0)
        self._chk(self._L.nik_group_pose_graph_cost(self._g, arr, _p(x) if x is not None else None, C.addressof(cost)))
        return cost.value

    def allreduce_residual(self, wait=True):
        out = np.zeros(4, np.float64)
        self._chk(self._L.nik_group_allreduce_residual(self._g, _p(out) if wait else None))
        return out if wait else None

    def residual_result(self):
        out = np.zeros(4, np.float64)
        self._chk(self._L.nik_group_residual_result(self._g, _p(out)))
        return out

    def gather_best(self, global_index, local_best):
        gi = _i32(global_index)
        arr = (NikPoseResult * len(gi))(*local_best)
        best = C.c_int(-1)
        res = NikPoseResult()
        self._chk(self._L.nik_group_gather_best(self._g, _p(gi), C.addressof(arr), C.byref(best), C.addressof(res)))
        return best.value, (res.as_dict() if best.value >= 0 else None)

    def track_batch(self, gray, keys, cur_dst, not_large_rotation=True):
        gray = np.ascontiguousarray(gray, np.uint8)
        n = gray.shape[0]
        k, d = _i32(keys), _i32(cur_dst)
        res = (NikPoseResult * n)()
        self._chk(self._L.nik_group_track_batch(self._g, n, _p(gray), _p(k), _p(d), int(bool(not_large_rotation)), C.addressof(res)))
        return [r.as_dict() for r in res]

    def match(self, query, query_slot, cands_per_member):
        query = np.ascontiguousarray(query, np.uint8)
        arrs = [_i32(c) for c in cands_per_member]
        nc = _i32([len(a) for a in arrs])
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        bm, bl = C.c_int(-1), C.c_int(-1)
        res = NikPoseResult()
        self._chk(self._L.nik_group_match(self._g, _p(query), int(query_slot), _p(nc), ptrs, C.byref(bm), C.byref(bl), C.addressof(res)))
        return bm.value, bl.value, (res.as_dict() if bm.value >= 0 else None)

    def close(self):
        if getattr(self, "_g", None):
            for f in self.flows:
                if getattr(f, "_borrowed", False):
                    f._ctx = None
            self._L.nik_group_destroy(self._g)
            self._g = None


def tracker_config(fx=600.0, fy=600.0, cx=320.0, cy=240.0, height=0.1, max_distance=0.4, max_angle=0.052359877,
                   lower_response_thr=30.0, upper_response_thr=90.0):
    """defaults: SURVEY.md 8(d) synthetic camera + reference configs/config_ntu.yaml:19-23"""
    cfg = NikTrackerConfig(fx=fx, fy=fy, cx=cx, cy=cy, height=height, max_distance=max_distance, max_angle=max_angle,
                           lower_response_thr=lower_response_thr, upper_response_thr=upper_response_thr)
    for i, v in enumerate([1, 0, 0, 0, 1, 0, 0, 0, 1]):
        cfg.extrinsics[i] = float(v)
    return cfg


def tracker_guess_gap(gaps):
    """the next keyframe gap nik_tracker_push_dev would guess after `gaps` (oldest first)"""
    g = _i32(gaps)
    return int(load().nik_tracker_guess_gap(_p(g), len(g)))


class Tracker:
    """the tracking subset of the reference MapBuilder (map_builder.cc:30-138) on top of a CorrelationFlow context"""

    def __init__(self, flow, cfg):
        self._flow, self._L = flow, flow._L
        self._t = C.c_void_p()
        rc = self._L.nik_tracker_create(flow._ctx, C.byref(cfg), C.byref(self._t))
        if rc:
            raise NikError(rc, "nik_tracker_create failed")

    def close(self):
        if getattr(self, "_t", None):
            self._L.nik_tracker_destroy(self._t)
            self._t = None

    __del__ = close

    def push_dev(self, d_gray_ptr, n):
        out = (NikTrackOutput * n)()
        rc = self._L.nik_tracker_push_dev(self._t, int(n), C.c_void_p(int(d_gray_ptr)), C.cast(out, C.c_void_p))
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode())
        return [o.as_dict() for o in out]

    def push_dev_into(self, d_gray_ptr, n, out, offset=0):
        """nik_tracker_push_dev writing into out[offset : offset + n] of a caller-owned (NikTrackOutput * N)() array -- what a C
        caller does; no per-frame Python objects (as_dict on 2048 outputs costs as much as 20 % of the sequence workload)"""
        rc = self._L.nik_tracker_push_dev(self._t, int(n), C.c_void_p(int(d_gray_ptr)), C.c_void_p(C.addressof(out) + int(offset) * C.sizeof(NikTrackOutput)))
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode())

    def stats(self):
        """nik_tracker_stats: dict of the look-ahead diagnostics"""
        out = (C.c_long * 8)()
        self._L.nik_tracker_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._L.nik_tracker_stats(self._t, out)
        return dict(guesses_held=out[0], guesses_failed=out[1], batches=out[2], pairs_enqueued=out[3], pairs_consumed=out[4], pairs_in_flight_behind_failed_guesses=out[5])

    def prefetch_dev(self, d_gray_ptr, n):
        """start ComputeIntermedium of the window that will be pushed NEXT (same pointer, same n): nik_tracker_prefetch_dev"""
        rc = self._L.nik_tracker_prefetch_dev(self._t, int(n), C.c_void_p(int(d_gray_ptr)))
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode())

    def push_host(self, frames, ptr=None, raw=False):
        """n host frames [n][H][W] u8 (numpy, or the address `ptr` of pinned memory holding them): windows of max_batch frames,
        the next window uploaded while the current one is registered (nik_tracker_push_host).  raw: return the ctypes output
        array itself (what a C caller holds) instead of one dict per frame"""
        n, H, W = frames.shape
        out = (NikTrackOutput * n)()
        src = C.c_void_p(int(ptr)) if ptr is not None else _p(np.ascontiguousarray(frames, np.uint8))
        rc = self._L.nik_tracker_push_host(self._t, int(n), src, int(W), int(H) * int(W), C.cast(out, C.c_void_p))
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode())
        return out if raw else [o.as_dict() for o in out]

    def push_u8(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        out = NikTrackOutput()
        rc = self._L.nik_tracker_push_u8(self._t, _p(gray), gray.shape[1], C.addressof(out))
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode())
        return out.as_dict()

    def attach_map(self, kmap, to_find_loop=True):
        rc = self._L.nik_tracker_attach_map(self._t, kmap._m if kmap is not None else None, int(bool(to_find_loop)))
        if rc:
            raise NikError(rc, "nik_tracker_attach_map: attach before the first frame")
        self._map = kmap                                   # keep it alive

    def loops(self):
        n = C.c_int(0)
        self._L.nik_tracker_loops(self._t, None, 0, C.addressof(n))
        out = (NikLoopResult * max(n.value, 1))()
        self._L.nik_tracker_loops(self._t, C.cast(out, C.c_void_p), n.value, C.addressof(n))
        return [out[i].as_dict() for i in range(n.value)]

    def pending_loops(self):
        return self._L.nik_tracker_pending_loops(self._t)

    def speculation(self):
        """[key-frame guesses that held, guesses that failed, batched pose calls] of push_dev so far"""
        out = (C.c_long * 3)()
        self._L.nik_tracker_speculation(self._t, out)
        return list(out)

    def poses(self):
        """robot poses of all keyframes: (ids, (n, 3) array) as last written by the tracker / the optimiser"""
        n = C.c_int(0)
        self._L.nik_tracker_poses(self._t, None, None, 0, C.addressof(n))
        ids = np.zeros(max(n.value, 1), np.int32); poses = np.zeros((max(n.value, 1), 3), np.float64)
        self._L.nik_tracker_poses(self._t, _p(ids), _p(poses), n.value, C.addressof(n))
        return ids[: n.value].tolist(), poses[: n.value]

    def edges(self):
        """Map::_edges as OptimizeMap feeds them to the solver: list of (id_begin, id_end, x, y, yaw, information, type)"""
        n = C.c_int(0)
        self._L.nik_tracker_edges(self._t, None, None, 0, C.addressof(n))
        cons = (NikPgConstraint * max(n.value, 1))(); types = np.zeros(max(n.value, 1), np.int32)
        self._L.nik_tracker_edges(self._t, C.cast(cons, C.c_void_p), _p(types), n.value, C.addressof(n))
        return [(cons[i].id_begin, cons[i].id_end, cons[i].x, cons[i].y, cons[i].yaw_radians, np.array(cons[i].information).reshape(3, 3), int(types[i]))
                for i in range(n.value)]

    def optimizations(self):
        sm = NikPgSummary()
        k = self._L.nik_tracker_optimizations(self._t, C.addressof(sm))
        return k, dict(termination=sm.termination, iterations=sm.iterations, successful_steps=sm.successful_steps,
                       initial_cost=sm.initial_cost, final_cost=sm.final_cost, inexact_solves=sm.inexact_solves)

    def keyframes(self):
        slots = np.zeros(self._flow.max_frames, np.int32)
        n = C.c_int(0)
        self._L.nik_tracker_keyframes(self._t, _p(slots), len(slots), C.addressof(n))
        return slots[: n.value].tolist()


class NikPgConstraint(C.Structure):
    """Constraint2d (include/optimization_2d/types.h:80-96)"""
    _fields_ = [("id_begin", C.c_int32), ("id_end", C.c_int32), ("x", C.c_double), ("y", C.c_double), ("yaw_radians", C.c_double),
                ("information", C.c_double * 9)]


class NikPgSummary(C.Structure):
    _fields_ = [("termination", C.c_int32), ("iterations", C.c_int32), ("successful_steps", C.c_int32), ("inexact_solves", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]


def _pg_constraints(constraints):
    cons = (NikPgConstraint * max(len(constraints), 1))()
    for k, (a, b, x, y, yaw, info) in enumerate(constraints):
        cons[k].id_begin, cons[k].id_end, cons[k].x, cons[k].y, cons[k].yaw_radians = int(a), int(b), x, y, yaw
        cons[k].information[:] = list(np.asarray(info, np.float64).reshape(9))
    return cons


def pose_graph_linearize(ids, poses, constraints, device=-1):
    """cost, gradient (n, 3) and J^T J diagonal blocks (n, 3, 3) at `poses`; device < 0: host, else that HIP device"""
    ids = np.ascontiguousarray(ids, np.int32)
    x = np.ascontiguousarray(np.array(poses, np.float64).reshape(len(ids), 3))
    cons = _pg_constraints(constraints)
    cost = C.c_double(0)
    g = np.zeros((len(ids), 3)); d = np.zeros((len(ids), 3, 3))
    rc = load().nik_pose_graph_linearize(int(device), len(ids), _p(ids), _p(x), len(constraints), C.cast(cons, C.c_void_p), C.addressof(cost), _p(g), _p(d))
    if rc:
        raise NikError(rc, "nik_pose_graph_linearize failed")
    return cost.value, g, d


class PgShard:
    """a shard of a pose graph's constraints resident on one GPU (nik_pg_shard): operand of Group.pose_graph_cost"""

    def __init__(self, device, ids, poses, constraints):
        ids = np.ascontiguousarray(ids, np.int32)
        x = np.ascontiguousarray(np.array(poses, np.float64).reshape(len(ids), 3))
        self._cons = _pg_constraints(constraints)
        self._s = C.c_void_p()
        rc = load().nik_pg_shard_create(int(device), len(ids), _p(ids), _p(x), len(constraints), C.cast(self._cons, C.c_void_p), C.byref(self._s))
        if rc:
            raise NikError(rc, "nik_pg_shard_create failed")

    def close(self):
        if getattr(self, "_s", None):
            load().nik_pg_shard_destroy(self._s)
            self._s = None

    __del__ = close


def pose_graph_optimize(ids, poses, constraints, max_iterations=300, device=-1):
    """MapBuilder::OptimizeMap's solve (pose_graph_2d.cc).  ids: frame ids (0 is held fixed); poses: (n, 3) x, y, yaw;
    constraints: iterable of (id_begin, id_end, x, y, yaw, information 3x3).  Returns (optimised poses, summary dict).
    device < 0: host only (needs the library but no GPU); else residuals / normal equations on that HIP device."""
    ids = np.ascontiguousarray(ids, np.int32)
    out = np.array(poses, np.float64).reshape(len(ids), 3).copy()
    cons = _pg_constraints(constraints)
    sm = NikPgSummary()
    if device >= 0:
        rc = load().nik_pose_graph_optimize_dev(int(device), len(ids), _p(ids), _p(out), len(constraints), C.cast(cons, C.c_void_p), int(max_iterations),
                                                C.addressof(sm))
    else:
        rc = load().nik_pose_graph_optimize(len(ids), _p(ids), _p(out), len(constraints), C.cast(cons, C.c_void_p), int(max_iterations),
                                            C.addressof(sm))
    if rc:
        raise NikError(rc, "nik_pose_graph_optimize: unknown pose id / no pose 0 / information not positive definite")
    return out, dict(termination=sm.termination, iterations=sm.iterations, successful_steps=sm.successful_steps,
                     initial_cost=sm.initial_cost, final_cost=sm.final_cost, inexact_solves=sm.inexact_solves)


class Stitcher:
    """MapStitcher (map_stitcher.cc): occupancy map of key frames on the device"""

    def __init__(self, flow, cell_size=1000):
        self._flow, self._L, self.cell_size = flow, flow._L, cell_size
        self._s = C.c_void_p()
        rc = self._L.nik_stitcher_create(flow._ctx, int(cell_size), C.byref(self._s))
        if rc:
            raise NikError(rc, "nik_stitcher_create failed")

    def close(self):
        if getattr(self, "_s", None):
            self._L.nik_stitcher_destroy(self._s)
            self._s = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode() or "stitcher call failed")

    def insert_dev(self, frame_id, d_image_ptr, image_pose):
        pose = np.ascontiguousarray(image_pose, np.float64)
        self._chk(self._L.nik_stitcher_insert_dev(self._s, int(frame_id), C.c_void_p(int(d_image_ptr)), _p(pose)))

    def recompute(self, frame_ids, image_poses):
        ids = np.ascontiguousarray(frame_ids, np.int32); poses = np.ascontiguousarray(image_poses, np.float64).reshape(len(ids), 3)
        self._chk(self._L.nik_stitcher_recompute(self._s, len(ids), _p(ids), _p(poses)))

    def cells(self):
        n = C.c_int(0)
        self._L.nik_stitcher_cells(self._s, None, 0, C.addressof(n))
        locs = np.zeros((max(n.value, 1), 2), np.int32)
        self._L.nik_stitcher_cells(self._s, _p(locs), n.value, C.addressof(n))
        return [tuple(int(v) for v in locs[i]) for i in range(n.value)]

    def read_cell(self, cx, cy):
        d = np.empty((self.cell_size, self.cell_size), np.int32); w = np.empty_like(d)
        self._chk(self._L.nik_stitcher_read_cell(self._s, int(cx), int(cy), _p(d), _p(w)))
        return d, w


class Pyramid:
    """coarse-to-fine registration over an image pyramid (BASELINE config 3; no reference counterpart)"""

    def __init__(self, cfg, H, W, levels=4, max_batch=32, device=0):
        self._L = load()
        self._p = C.c_void_p()
        self.levels, self.max_batch = levels, max_batch
        rc = self._L.nik_pyramid_create(C.byref(cfg), H, W, levels, max_batch, device, C.byref(self._p))
        if rc:
            raise NikError(rc, "nik_pyramid_create failed (unsupported level geometry?)")
        d = np.zeros((levels, 4), np.int32)
        self._L.nik_pyramid_levels(self._p, _p(d))
        self.dims = d.tolist()                      # [level] -> [H, W, PD, PC]

    def close(self):
        if getattr(self, "_p", None):
            self._L.nik_pyramid_destroy(self._p)
            self._p = None

    __del__ = close

    def track_dev(self, d_key_ptr, d_cur_ptr, n, radius=4):
        res = (NikPoseResult * (self.levels * n))()
        rc = self._L.nik_pyramid_track_dev(self._p, int(n), C.c_void_p(int(d_key_ptr)), C.c_void_p(int(d_cur_ptr)), int(radius),
                                           C.cast(res, C.c_void_p))
        if rc:
            raise NikError(rc, "; ".join(self._L.nik_pyramid_last_error(self._p, l).decode() for l in range(self.levels)))
        return [[res[l * n + i].as_dict() for i in range(n)] for l in range(self.levels)]

    def _raise(self, rc):
        raise NikError(rc, "; ".join(self._L.nik_pyramid_last_error(self._p, l).decode() for l in range(self.levels)))

    def track_dev_async(self, d_key_ptr, d_cur_ptr, n, radius=4, res=None):
        """enqueue one batch without waiting; `res` (NikPoseResult * (levels * n)) is final after synchronize().  Successive
        batches pipeline: the frames and `res` of a batch must stay alive until then."""
        if res is None:
            res = (NikPoseResult * (self.levels * n))()
        rc = self._L.nik_pyramid_track_dev_async(self._p, int(n), C.c_void_p(int(d_key_ptr)), C.c_void_p(int(d_cur_ptr)), int(radius),
                                                 C.cast(res, C.c_void_p))
        if rc:
            self._raise(rc)
        return res

    def synchronize(self):
        rc = self._L.nik_pyramid_synchronize(self._p)
        if rc:
            self._raise(rc)

    def as_lists(self, res, n):
        return [[res[l * n + i].as_dict() for i in range(n)] for l in range(self.levels)]


def loop_config(grid_scale=0.1, to_find_loop=True, frame_gap_thr=100, distance_thr=5.0, position_response_thr=60.0,
                angle_response_thr=60.0):
    """defaults: /root/reference/configs/config_ntu.yaml:24-33"""
    return NikLoopConfig(grid_scale, int(to_find_loop), int(frame_gap_thr), distance_thr, position_response_thr, angle_response_thr)


class KeyframeMap:
    """Map + LoopClosure candidate management (map.cc, loop_closure.cc).  flow=None: candidate queries only (no GPU)."""

    def __init__(self, flow, cfg):
        self._flow, self._L = flow, load()
        self._m = C.c_void_p()
        rc = self._L.nik_map_create(flow._ctx if flow is not None else None, C.byref(cfg), C.byref(self._m))
        if rc:
            raise NikError(rc, "nik_map_create failed")

    def close(self):
        if getattr(self, "_m", None):
            self._L.nik_map_destroy(self._m)
            self._m = None

    __del__ = close

    def __len__(self):
        return self._L.nik_map_size(self._m)

    def add_frame(self, frame_id, slot, pose, distance=None):
        pose = np.ascontiguousarray(pose, np.float64)
        d = None if distance is None else np.array([distance], np.float64)
        rc = self._L.nik_map_add_frame(self._m, int(frame_id), int(slot), _p(pose), _p(d))
        if rc:
            raise NikError(rc, "nik_map_add_frame: duplicate frame id / bad argument")

    def candidates(self, cur_frame_id, prior_pose=None):
        ids = np.zeros(max(len(self), 1), np.int32)
        n = C.c_int(0)
        pp = None if prior_pose is None else np.ascontiguousarray(prior_pose, np.float64)
        rc = self._L.nik_map_candidates(self._m, int(cur_frame_id), _p(pp), _p(ids), len(ids), C.addressof(n))
        if rc:
            raise NikError(rc, "nik_map_candidates: unknown frame id")
        return ids[: n.value].tolist()

    def find_loop(self, cur_frame_id, prior_pose=None):
        out = NikLoopResult()
        pp = None if prior_pose is None else np.ascontiguousarray(prior_pose, np.float64)
        rc = self._L.nik_map_find_loop(self._m, int(cur_frame_id), _p(pp), C.addressof(out))
        if rc:
            raise NikError(rc, self._L.nik_last_error(self._flow._ctx).decode() if self._flow is not None else "no context")
        return out.as_dict()
