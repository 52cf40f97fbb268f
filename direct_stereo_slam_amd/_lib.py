"""ctypes binding of the C ABI (include/dsm_hotpath.h) -- the only way the Python host side reaches
the device.  There is no CPU fallback: if the HIP library is missing or no GPU is present the
failure is loud (ImportError / DsmError)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSM_HOTPATH_LIB: developer override used for A/B builds of the same sources (e.g. other compiler flags)
LIB_PATH = os.environ.get("DSM_HOTPATH_LIB") or os.path.join(_HERE, "lib", "libdsm_hotpath.so")
MAX_LEVELS = 6
ABI_VERSION = 4  # DSM_ABI_VERSION of the header the structures below mirror

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_int64_p = C.POINTER(C.c_int64)
NO_CANDIDATE = 0x7FFFFFFFFFFFFFFF


class DsmError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t),
        ("huber_th", C.c_float),
        ("coarse_cutoff_th", C.c_float),
        ("scale_xi_rot", C.c_float),
        ("scale_xi_trans", C.c_float),
        ("scale_a", C.c_float),
        ("scale_b", C.c_float),
        ("affine_opt_mode_a", C.c_float),
        ("affine_opt_mode_b", C.c_float),
        ("lambda_extrapolation_limit", C.c_float),
        ("max_iterations", C.c_int * MAX_LEVELS),
        ("adaptive_schedule", C.c_int),
        ("persistent_coarse", C.c_int),
        ("fuse_lm", C.c_int),
        ("work_queue", C.c_int),
        ("speculate", C.c_int),
        ("compact_tail", C.c_int),
        ("fixed_schedule", C.c_int),
        ("chunk_geometry", C.c_int),
        ("frame_check", C.c_int),
        ("frame_grad_tol", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("evals", C.c_int64 * MAX_LEVELS),
        ("launches", C.c_int64 * MAX_LEVELS),
        ("algorithmic_bytes", C.c_int64),
        ("eval_kernel_ms", C.c_double * MAX_LEVELS),
        ("eval_kernel_union_ms", C.c_double * MAX_LEVELS),
        ("eval_dispatches", C.c_int64 * MAX_LEVELS),
        ("total_ms", C.c_double),
        ("polls", C.c_int64),
        ("coarse_launches", C.c_int64),
        ("queue_blocks", C.c_int64),
        ("queue_items", C.c_int64),
        ("queue_kernel_ms", C.c_double),
        ("evals_residual_only", C.c_int64 * MAX_LEVELS),
    ]


class StreamResult(C.Structure):
    _fields_ = [
        ("ticket", C.c_uint64),
        ("kind", C.c_int),
        ("good", C.c_int),
        ("status", C.c_int),
        ("passes", C.c_int),
        ("pose", C.c_double * 7),
        ("aff", C.c_double * 2),
        ("last_residuals", C.c_double * MAX_LEVELS),
        ("flow", C.c_double * 3),
        ("scale", C.c_float),
        ("err", C.c_float),
        ("evals", C.c_int64 * MAX_LEVELS),
    ]


class RefJob(C.Structure):
    _fields_ = [
        ("t", C.c_void_p), ("frame_owner", C.c_void_p), ("slot", C.c_int), ("ref_frame_id", C.c_int),
        ("ref_aff_a", C.c_double), ("ref_aff_b", C.c_double), ("ref_exposure", C.c_float), ("npts", C.c_int),
        ("pu", c_float_p), ("pv", c_float_p), ("pidepth", c_float_p), ("pweight", c_float_p), ("n_out", c_int_p),
    ]


class LoopJob(C.Structure):
    _fields_ = [
        ("n_kf", C.c_int), ("kf_ids", c_int_p), ("kf_pose_wc", c_double_p), ("cur_cw", c_double_p),
        ("n_pts", C.c_int), ("pt_kf_id", c_int_p), ("pt_xyz", c_double_p),
        ("kf_keep", c_int_p), ("n_out", c_int_p), ("sel_idx", c_int_p), ("pts_spherical", c_double_p),
        ("ringkey", c_float_p), ("sig_idx", c_int_p), ("sig_val", c_double_p), ("n_sig", c_int_p), ("tfm_pca_rig", c_double_p),
    ]


# every symbol include/dsm_hotpath.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_pp_f = C.POINTER(c_float_p)
_pp_i = C.POINTER(c_int_p)
_pp_d = C.POINTER(c_double_p)
# transports of dsm_ringdb_merge_topk_with: (user, d_buf, count, hip_stream) / (user, d_send, d_recv, count, hip_stream)
ALLREDUCE_MIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
MERGE_ALGOS = {"allreduce_min": 0, "allgather": 1}
SYMBOLS = {
    "dsm_last_error": (C.c_char_p, []),
    "dsm_abi_version": (C.c_int, []),
    "dsm_tracker_upload_intensity": (C.c_int, [_vp, C.c_int, _pp_f, C.c_float]),
    "dsm_params_default": (None, [C.POINTER(Params)]),
    "dsm_params_default_sized": (C.c_int, [C.POINTER(Params), C.c_size_t]),
    "dsm_stream_create": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "dsm_stream_destroy": (C.c_int, [_vp]),
    "dsm_stream_submit_track": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_double_p, c_double_p, C.c_int, c_double_p, C.POINTER(C.c_uint64)]),
    "dsm_stream_submit_scale": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_float_p, C.c_int, C.POINTER(C.c_uint64)]),
    "dsm_stream_advance": (C.c_int, [_vp]),
    "dsm_stream_drain": (C.c_int, [_vp]),
    "dsm_stream_sync": (C.c_int, [_vp]),
    "dsm_stream_set_pipelined": (C.c_int, [_vp, C.c_int]),
    "dsm_stream_results": (C.c_int, [_vp, C.c_int, C.POINTER(StreamResult), c_int_p]),
    "dsm_stream_counts": (C.c_int, [_vp, c_int_p, c_int_p, c_int_p]),
    "dsm_stream_set_quantile": (C.c_int, [_vp, C.c_int, C.c_double]),
    "dsm_stream_set_engine": (C.c_int, [_vp, C.c_int, C.c_int]),
    "dsm_stream_set_chain": (C.c_int, [_vp, C.c_int]),
    "dsm_stream_set_rounds": (C.c_int, [_vp, C.c_int, c_int_p]),
    "dsm_stream_get_stats": (C.c_int, [_vp, C.POINTER(Stats), C.POINTER(Stats)]),
    "dsm_stream_get_schedule": (C.c_int, [_vp, C.c_int, c_int_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dsm_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "dsm_context_destroy": (C.c_int, [_vp]),
    "dsm_context_sync": (C.c_int, [_vp]),
    "dsm_context_set_timing": (C.c_int, [_vp, C.c_int]),
    "dsm_context_set_streams": (C.c_int, [_vp, C.c_int]),
    "dsm_context_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "dsm_context_stream": (_vp, [_vp]),
    "dsm_context_stream_queues": (C.c_int, [_vp, c_int_p, c_int_p]),
    "dsm_diag_read_bandwidth": (C.c_int, [_vp, C.c_size_t, C.c_int, c_double_p]),
    "dsm_diag_read_bandwidth_chunked": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.c_int, c_double_p]),
    "dsm_diag_xwg_litmus": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dsm_tracker_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, c_double_p, c_float_p, C.POINTER(Params), C.POINTER(_vp)]),
    "dsm_tracker_destroy": (C.c_int, [_vp]),
    "dsm_tracker_make_k": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    "dsm_tracker_set_ref": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_float, c_int_p, _pp_f, _pp_f, _pp_f, _pp_f]),
    "dsm_set_refs_from_points": (C.c_int, [_vp, C.c_int, C.POINTER(RefJob)]),
    "dsm_tracker_set_ref_from_points": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_int_p]),
    "dsm_tracker_scale_depth": (C.c_int, [_vp, C.c_float]),
    "dsm_tracker_get_template": (C.c_int, [_vp, C.c_int, c_int_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "dsm_tracker_upload_frame": (C.c_int, [_vp, C.c_int, _pp_f, C.c_float]),
    "dsm_tracker_upload_image": (C.c_int, [_vp, C.c_int, c_float_p, C.c_float]),
    "dsm_upload_images": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_int_p, C.POINTER(_vp), c_float_p, C.c_int, C.c_size_t]),
    "dsm_upload_images_async": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_int_p, C.POINTER(_vp), c_float_p, C.c_int, C.c_size_t]),
    "dsm_upload_images_enqueue": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_int_p, C.POINTER(_vp), c_float_p, C.c_int, C.c_size_t]),
    "dsm_upload_wait": (C.c_int, [_vp]),
    "dsm_frames_advance": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_int_p]),
    "dsm_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "dsm_host_free": (C.c_int, [_vp]),
    "dsm_tracker_get_frame": (C.c_int, [_vp, C.c_int, C.c_int, c_float_p]),
    "dsm_tracker_calc_res_pose": (C.c_int, [_vp, C.c_int, c_double_p, c_double_p, C.c_float, c_double_p, c_double_p, c_double_p, c_int_p]),
    "dsm_tracker_calc_res_scale": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, c_double_p, c_float_p, c_float_p, c_int_p]),
    "dsm_tracker_track": (C.c_int, [_vp, c_double_p, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]),
    "dsm_tracker_optimize_scale": (C.c_int, [_vp, c_float_p, C.c_int, c_float_p]),
    "dsm_tracker_optimize_scale_guesses": (C.c_int, [_vp, C.c_int, c_float_p, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]),
    "dsm_track_and_scale_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_double_p, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p,
                                           C.c_int, C.POINTER(_vp), c_float_p, c_float_p]),
    "dsm_context_get_stats2": (C.c_int, [_vp, C.POINTER(Stats)]),
    "dsm_track_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_double_p, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]),
    "dsm_optimize_scale_batch": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), c_float_p, C.c_int, c_float_p]),
    "dsm_tracker_ref_frame_id": (C.c_int, [_vp]),
    "dsm_reduction_geometry": (C.c_int, [_vp, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p]),
    "dsm_pose_estimator_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(_vp)]),
    "dsm_pose_estimator_destroy": (C.c_int, [_vp]),
    "dsm_pose_estimator_estimate": (C.c_int, [_vp, C.c_int, c_double_p, _pp_f, C.c_float, _pp_f, C.c_float, c_float_p, C.c_int, c_double_p, c_float_p, c_int_p]),
    "dsm_ringdb_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_float, c_float_p, C.c_int64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "dsm_ringdb_destroy": (C.c_int, [_vp]),
    "dsm_ringdb_size": (C.c_int64, [_vp]),
    "dsm_ringdb_query_then_enqueue": (C.c_int, [_vp, c_float_p, c_int_p, c_int_p]),
    "dsm_ringdb_add_points": (C.c_int, [_vp, c_float_p, C.c_int64]),
    "dsm_ringdb_enqueue": (C.c_int, [_vp, c_float_p]),
    "dsm_ringdb_knn_packed": (C.c_int, [_vp, c_float_p, C.c_int, _vp]),
    "dsm_ringdb_knn_packed_dev": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "dsm_ringdb_knn_packed_host": (C.c_int, [_vp, c_float_p, C.c_int, c_int64_p]),
    "dsm_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "dsm_comm_create": (C.c_int, [_vp, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(_vp)]),
    "dsm_comm_destroy": (C.c_int, [_vp]),
    "dsm_comm_rank": (C.c_int, [_vp]),
    "dsm_comm_size": (C.c_int, [_vp]),
    "dsm_ringdb_merge_topk": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "dsm_ringdb_merge_topk_with": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, ALLREDUCE_MIN_FN, ALLGATHER_FN, _vp]),
    "dsm_ringdb_attach_comm": (C.c_int, [_vp, _vp]),
    "dsm_ringdb_attach_transport": (C.c_int, [_vp, C.c_int, ALLREDUCE_MIN_FN, ALLGATHER_FN, _vp]),
    "dsm_scancontext_generate": (C.c_int, [c_double_p, C.c_int, C.c_double, C.c_int, C.c_int, c_float_p, c_int_p, c_double_p, c_int_p, c_double_p]),
    "dsm_generate_spherical_points": (C.c_int, [C.c_int, c_int_p, c_double_p, c_double_p, C.c_double, C.c_int, c_int_p, c_double_p, c_int_p, c_int_p, c_int_p, c_double_p]),
    "dsm_loop_descriptors_batch": (C.c_int, [_vp, C.c_int, C.POINTER(LoopJob), C.c_double, C.c_int, C.c_int]),
    "dsm_loop_detect_batch": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(LoopJob), C.c_double, C.c_int, C.c_int, c_int_p, c_int_p]),
    "dsm_write_trajectory": (C.c_int, [C.c_char_p, C.c_int, c_int_p, c_double_p]),
    "dsm_make_coarse_depth_l0": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, _pp_f, c_int_p, _pp_f, _pp_f, _pp_f, _pp_f]),
    "dsm_sc_distance": (C.c_float, [c_int_p, c_double_p, C.c_int, c_int_p, c_double_p, C.c_int, C.c_int]),
    "dsm_search_sc": (C.c_int, [c_int_p, c_double_p, C.c_int, C.c_int, c_int_p, _pp_i, _pp_d, c_int_p, C.c_int, c_int_p, c_float_p]),
}

_lib = None


def load():
    """Load libdsm_hotpath.so (built by __graft_entry__.build()).  Raises ImportError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: this image's PyTorch bundles its own libamdhip64.so (same SONAME
    # libamdhip64.so.7 as /opt/rocm's).  If torch is imported AFTER this library, the loader maps a
    # second copy of the runtime; imported before, ours resolves to the already loaded one.  So when
    # torch is installed, make sure it is loaded first (bench.py and the RCCL merge use it anyway).
    import sys

    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: plain /opt/rocm runtime
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        except AttributeError:
            if os.environ.get("DSM_HOTPATH_LIB") and os.environ.get("DSM_HOTPATH_LIB_OLDER_BUILD"):
                continue  # developer A/B against a build of an earlier commit (same ABI version, fewer entry points)
            raise
        fn.restype = res
        fn.argtypes = args
    # the struct layouts above are those of ABI version ABI_VERSION (include/dsm_hotpath.h: DSM_ABI_VERSION)
    have = L.dsm_abi_version()
    if have != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} implements ABI version {have}, this binding was written for {ABI_VERSION}: rebuild the library")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().dsm_last_error()
        raise DsmError(f"dsm error {rc}: {msg.decode() if msg else ''}")
