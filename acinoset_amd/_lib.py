"""ctypes binding of libacinoset_hip.so (the C ABI in include/acinoset_hip.h).

The library is built in-tree by ``build()`` (hipcc, gfx950 only).  There is NO CPU fallback:
``lib()`` raises if the shared object is missing or cannot be loaded, and every wrapper raises
``RuntimeError`` carrying ``acino_last_error_string()`` on a non-zero status.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libacinoset_hip.so")
BUILD_ID_SOURCE = "camera_kernels.hip"      # defines acino_build_id()
SOURCES = ["camera_kernels.hip", "fte_assemble.hip", "bcr.hip", "seplevel.hip", "chunk.hip", "fte_api.hip", "sba.hip", "ekf.hip", "skel_fte.hip"]
HEADERS = ["common.hpp", "fte_kernels.hpp", "bcr.hpp", "bcr_dev.hpp", "seplevel.hpp", "chunk.hpp", "trio80.hpp", "dense80.hpp", "cheetah_fk.hpp", os.path.join("..", "..", "include", "acinoset_hip.h")]

ABI_VERSION = 3          # ACINO_ABI_VERSION of include/acinoset_hip.h
N_ACTIVE = 25
N_STATES = 45
N_MARKERS = 20
CAM_STRIDE = 24
PINHOLE_STRIDE = 32
BS = 80
SEP_DOUBLES = 2 * BS * BS + BS


class FteParams(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_cams", C.c_int32), ("n_global", C.c_int64), ("n_offset", C.c_int64),
                ("pin_left", C.c_int32), ("pin_right", C.c_int32), ("dlc_thresh", C.c_double),
                ("inv_r_meas", C.c_double), ("redesc_a", C.c_double), ("redesc_b", C.c_double),
                ("redesc_c", C.c_double), ("q_w", C.c_double * N_ACTIVE), ("lo", C.c_double * N_ACTIVE),
                ("hi", C.c_double * N_ACTIVE), ("lam0", C.c_double), ("ftol", C.c_double), ("xtol", C.c_double),
                ("gtol", C.c_double), ("lam_max", C.c_double), ("clamp_lambda", C.c_int32), ("shared_gpu", C.c_int32),
                ("clip_len", C.c_int64), ("precision", C.c_int32), ("bcr_levels", C.c_int32), ("trunc_tol", C.c_double),
                ("own_first", C.c_int32), ("own_count", C.c_int32),
                ("chunk_nodes", C.c_int32), ("refine_sweeps", C.c_int32)]


class FteState(C.Structure):
    _fields_ = [("cost", C.c_double), ("cost_trial", C.c_double), ("lam", C.c_double), ("nu", C.c_double),
                ("gain", C.c_double), ("pred", C.c_double), ("step_inf", C.c_double), ("gnorm_inf", C.c_double),
                ("iter", C.c_int32), ("accepted", C.c_int32), ("status", C.c_int32), ("cur", C.c_int32),
                ("n_behind", C.c_int32), ("last_accept", C.c_int32), ("pad0", C.c_int32), ("pad1", C.c_int32),
                ("trunc_eps", C.c_double)]

    def as_dict(self):
        names = {0: "running", 1: "ftol", 2: "xtol", 3: "gtol", 4: "lambda_overflow", 5: "numeric", 6: "sync_timeout",
                 7: "truncation"}
        d = {f: getattr(self, f) for f, _ in self._fields_ if not f.startswith("pad")}
        d["status_name"] = names.get(self.status, "?")
        return d


class EkfParams(C.Structure):
    _fields_ = [("n_frames", C.c_int64), ("n_seq", C.c_int32), ("n_cams", C.c_int32), ("fps", C.c_double),
                ("dlc_thresh", C.c_double), ("cam_width", C.c_double), ("smoother_pivoting", C.c_int32),
                ("reserved", C.c_int32)]


class SkelOp(C.Structure):
    _fields_ = [("child", C.c_int32), ("parent", C.c_int32), ("angle", C.c_int32), ("flags", C.c_int32),
                ("off", C.c_double * 3)]


class SkelFteParams(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_cams", C.c_int32), ("n_pose", C.c_int32), ("n_ops", C.c_int32),
                ("n_angles", C.c_int32), ("n_active", C.c_int32), ("max_iter", C.c_int32), ("pad0", C.c_int32),
                ("h", C.c_double), ("model_weight", C.c_double), ("l1_eps", C.c_double), ("lam0", C.c_double),
                ("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double), ("lam_max", C.c_double)]


class SkelFteInfo(C.Structure):
    _fields_ = [("cost_initial", C.c_double), ("cost_final", C.c_double), ("gnorm_inf", C.c_double), ("lam", C.c_double),
                ("iterations", C.c_int32), ("accepted", C.c_int32), ("status", C.c_int32), ("pad0", C.c_int32)]

    def as_dict(self):
        names = {0: "max_iter", 1: "ftol", 2: "xtol", 3: "gtol", 4: "lambda_overflow", 5: "numeric"}
        d = {f: getattr(self, f) for f, _ in self._fields_ if not f.startswith("pad")}
        d["status_name"] = names.get(self.status, "?")
        return d


class SbaParams(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("optimize_cameras", C.c_int32), ("n_points", C.c_int64), ("n_obs", C.c_int64),
                ("f_scale", C.c_double), ("lam0", C.c_double), ("ftol", C.c_double), ("gtol", C.c_double),
                ("max_iter", C.c_int32), ("camera_model", C.c_int32), ("precision", C.c_int32), ("pad0", C.c_int32)]


class SbaInfo(C.Structure):
    _fields_ = [("cost_initial", C.c_double), ("cost_final", C.c_double), ("gnorm_inf", C.c_double), ("lam", C.c_double),
                ("iterations", C.c_int32), ("accepted", C.c_int32), ("status", C.c_int32), ("pad0", C.c_int32)]

    def as_dict(self):
        names = {0: "max_iter", 1: "ftol", 3: "gtol", 4: "lambda_overflow", 5: "numeric"}
        d = {f: getattr(self, f) for f, _ in self._fields_ if not f.startswith("pad")}
        d["status_name"] = names.get(self.status, "?")
        return d


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_D = C.c_double
_Z = C.c_size_t

# acino_reduce_fn: int (*)(void* user, double* d_buf, int64_t n, int op, void* stream)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)

# name -> (restype, argtypes); every symbol include/acinoset_hip.h declares
SIGNATURES = {
    "acino_last_error_string": (C.c_char_p, []),
    "acino_abi_version": (_I, []),
    "acino_build_id": (C.c_char_p, []),
    "acino_triangulate_pairs_pinhole": (_I, [_P, _L, _I, _I, _D, _P, _P, _P, _P, _P]),
    "acino_device_count": (_I, []),
    "acino_undistort_fisheye": (_I, [_P, _L, _P, _P, _I, _D, _P]),
    "acino_triangulate_fisheye": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "acino_triangulate_pinhole": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "acino_project_fisheye": (_I, [_P, _L, _P, _P, _P]),
    "acino_project_pinhole": (_I, [_P, _L, _P, _P, _P]),
    "acino_triangulate_pairs": (_I, [_P, _L, _I, _I, _D, _P, _P, _P, _P, _P]),
    "acino_fte_triangulation_init_scratch_bytes": (_Z, [_L]),
    "acino_fte_triangulation_init": (_I, [_P, _L, _I, _P, _I, _I, _P, _Z, _P, _P]),
    "acino_reproject_residuals": (_I, [_P, _P, _L, _I, _I, _D, _P, _P, _P, _P]),
    "acino_triangulate_reproject": (_I, [_P, _L, _I, _I, _D, _P, _P, _P, _P, _P, _P, _P]),
    "acino_cheetah_fk": (_I, [_P, _L, _P, _P]),
    "acino_fk_active": (_I, [_P, _L, _P, _P]),
    "acino_sizeof_fte_params": (_Z, []),
    "acino_sizeof_fte_state": (_Z, []),
    "acino_fte_workspace_bytes": (_Z, [C.POINTER(FteParams)]),
    "acino_fte_create": (_I, [C.POINTER(_P), C.POINTER(FteParams), _P, _P, _P, _Z, _P]),
    "acino_fte_destroy": (_I, [_P]),
    "acino_fte_plan": (_I, [C.POINTER(FteParams), C.POINTER(C.c_int32)]),
    "acino_fte_set_x": (_I, [_P, _P, _P]),
    "acino_fte_step": (_I, [_P, _P]),
    "acino_fte_enable_graph": (_I, [_P, _I]),
    "acino_fte_solve": (_I, [_P, _I, C.POINTER(FteState), _P]),
    "acino_fte_get_state": (_I, [_P, C.POINTER(FteState), _P]),
    "acino_fte_get_result": (_I, [_P, _D, _P, _P, _P, _P, _P]),
    "acino_fte_copy_frames": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "acino_fte_set_precision": (_I, [_P, _I]),
    "acino_fte_reevaluate": (_I, [_P, _P]),
    "acino_fte_cost": (_I, [_P, _P, _P, _P]),
    "acino_fte_get_grad_hess": (_I, [_P, _P, _P, _P]),
    "acino_fte_derivatives": (_I, [_P, _L, _D, _P, _P, _P]),
    "acino_fte_load_x": (_I, [_P, _P, _P]),
    "acino_fte_set_halo": (_I, [_P, _I, _P, _P, _P]),
    "acino_fte_eval": (_I, [_P, _I, _P]),
    "acino_fte_export_partials": (_I, [_P, _P, _P]),
    "acino_fte_control": (_I, [_P, _P, _I, _P]),
    "acino_fte_reduce_local": (_I, [_P, _P]),
    "acino_fte_export_separators": (_I, [_P, _P, _I, _I, _P]),
    "acino_sep_scratch_bytes": (_Z, [_I]),
    "acino_solve_separators": (_I, [_P, _I, _P, _P, _Z, _P]),
    "acino_fte_backsub_local": (_I, [_P, _P, _I, _I, _P]),
    "acino_fte_trial": (_I, [_P, _P]),
    "acino_fte_export_edges": (_I, [_P, _I, _P, _P]),
    "acino_fte_graphs_active": (_I, [_P]),
    "acino_fte_shard_reduce": (_I, [_P, _P, _I, _I, _P]),
    "acino_fte_shard_solve": (_I, [_P, _P, _P, _P, _Z, _P, _I, _I, _P]),
    "acino_fte_shard_eval": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "acino_fte_shard_control": (_I, [_P, _P, _I, _I, _P]),
    "acino_fte_profile_begin": (_I, [_P]),
    "acino_fte_debug_read": (_I, [_P, _I, _P, _L, _P]),
    "acino_fte_debug_stamps": (_I, [_P, _P]),
    "acino_fte_profile_end": (_I, [_P, _P, _P, _P, _P]),
    "acino_sizeof_sba_params": (_Z, []),
    "acino_sizeof_sba_info": (_Z, []),
    "acino_sba_workspace_bytes": (_Z, [_I, _L, _L]),
    "acino_sba_solve": (_I, [C.POINTER(SbaParams), _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _P, C.POINTER(SbaInfo), _P]),
    "acino_sba_solve_sharded": (_I, [C.POINTER(SbaParams), _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _P, C.POINTER(SbaInfo),
                                     REDUCE_FN, _P, _P]),
    "acino_sizeof_ekf_params": (_Z, []),
    "acino_ekf_workspace_bytes": (_Z, [_L, _I]),
    "acino_ekf_run": (_I, [C.POINTER(EkfParams), _P, _P, _P, _P, _Z, _P, _P, _P, _P]),
    "acino_skeleton_fk": (_I, [_P, _L, _I, _I, C.POINTER(SkelOp), _I, _P, _P]),
    "acino_sizeof_skel_fte_params": (_Z, []),
    "acino_sizeof_skel_fte_info": (_Z, []),
    "acino_skel_fte_workspace_bytes": (_Z, [C.POINTER(SkelFteParams)]),
    "acino_skel_fte_workspace_bytes_batch": (_Z, [C.POINTER(SkelFteParams), _I]),
    "acino_skel_fte_solve": (_I, [C.POINTER(SkelFteParams), C.POINTER(SkelOp), C.POINTER(C.c_int32), _P, _P, _P, _P, _P, _P, _P,
                                  _P, _Z, C.POINTER(SkelFteInfo), _P]),
    "acino_skel_fte_solve_batch": (_I, [C.POINTER(SkelFteParams), _I, C.POINTER(SkelOp), C.POINTER(C.c_int32), _P, _P, _P, _P, _P,
                                        _P, _P, _P, _Z, C.POINTER(SkelFteInfo), _P]),
    "acino_selftest_mfma": (_I, [_P, _P, _I, _P, _P]),
    "acino_debug_poison_lds": (_I, [_I, _I, _P]),
    "acino_debug_level_split": (_I, [_I, _P]),
}

_lib = None


def source_hash():
    """sha256 over every HIP source and header (name + content): the identity of a build."""
    import hashlib
    h = hashlib.sha256()
    for rel in sorted(SOURCES + HEADERS):
        path = os.path.join(_CSRC, rel)
        h.update(os.path.basename(rel).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_id():
    """Build id embedded in the shared object on disk, read from the file (dlopen caches a path once loaded, so a
    rebuilt library would keep answering with the old id inside one process).  None when absent / unstamped."""
    if not os.path.exists(SO_PATH):
        return None
    with open(SO_PATH, "rb") as f:
        blob = f.read()
    i = blob.find(b"ACINO_BUILD_ID=")
    if i < 0:
        return None
    return blob[i + 15:i + 31].decode("ascii", "replace")


def _needs_build():
    # identity, not mtimes: the .so travels with the tree (gpurun snapshot), a stale binary must not pass build()
    return built_id() != source_hash()


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into acinoset_amd/libacinoset_hip.so (hipcc cross-compiles
    without a GPU).  Every source becomes one object under csrc/_obj/ keyed by the hash of its text and all
    headers (unchanged files are not recompiled), the objects compile in parallel, and the source hash is
    embedded as ``acino_build_id()``.  Raises on any compiler error."""
    if not force and not _needs_build():
        return SO_PATH
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    sid = source_hash()
    hdr = hashlib.sha256()
    for rel in sorted(HEADERS):
        with open(os.path.join(_CSRC, rel), "rb") as f:
            hdr.update(f.read())
    objdir = os.path.join(_CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG"]
    jobs = []
    for src in SOURCES:
        with open(os.path.join(_CSRC, src), "rb") as f:
            key = hashlib.sha256(hdr.digest() + f.read() + " ".join(flags).encode()).hexdigest()[:16]
        extra = [f'-DACINO_BUILD_ID="{sid}"'] if src == BUILD_ID_SOURCE else []
        if extra:
            key = hashlib.sha256((key + sid).encode()).hexdigest()[:16]
        obj = os.path.join(objdir, f"{os.path.splitext(src)[0]}.{key}.o")
        jobs.append((src, obj, [hipcc] + flags + extra + ["-c", os.path.join(_CSRC, src), "-o", obj]))

    def run(job):
        src, obj, cmd = job
        if os.path.exists(obj) and not force:
            return None
        for old in os.listdir(objdir):                       # one object per source is kept
            if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".o"):
                os.unlink(os.path.join(objdir, old))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or res.returncode != 0:
            print(" ".join(cmd))
            print(res.stdout)
            print(res.stderr)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + res.stderr[-4000:])
        return src

    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        list(ex.map(run, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [j[1] for j in jobs] + ["-o", SO_PATH + ".tmp"]
    res = subprocess.run(link, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(link))
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed linking libacinoset_hip.so:\n" + res.stderr[-4000:])
    os.replace(SO_PATH + ".tmp", SO_PATH)
    got = built_id()
    if got != sid:
        raise RuntimeError(f"built library reports build id {got}, sources hash to {sid}")
    return SO_PATH


def lib():
    """The loaded library; raises RuntimeError when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(acinoset_amd has no CPU fallback)")
    try:
        handle = C.CDLL(SO_PATH)
    except OSError as exc:
        raise RuntimeError(f"cannot load {SO_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)   # AttributeError = ABI mismatch, let it surface
        fn.restype = res
        fn.argtypes = args
    if handle.acino_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libacinoset_hip.so reports ABI version {handle.acino_abi_version()}, this binding is written for "
                           f"{ABI_VERSION} (stale build?)")
    if handle.acino_sizeof_fte_params() != C.sizeof(FteParams) or handle.acino_sizeof_fte_state() != C.sizeof(FteState):
        raise RuntimeError("libacinoset_hip.so struct layout differs from the Python binding (stale build?)")
    if handle.acino_sizeof_sba_params() != C.sizeof(SbaParams) or handle.acino_sizeof_sba_info() != C.sizeof(SbaInfo):
        raise RuntimeError("libacinoset_hip.so SBA struct layout differs from the Python binding (stale build?)")
    if (handle.acino_sizeof_skel_fte_params() != C.sizeof(SkelFteParams) or
            handle.acino_sizeof_skel_fte_info() != C.sizeof(SkelFteInfo)):
        raise RuntimeError("libacinoset_hip.so skeleton-FTE struct layout differs from the Python binding (stale build?)")
    if handle.acino_sizeof_ekf_params() != C.sizeof(EkfParams):
        raise RuntimeError("libacinoset_hip.so EKF struct layout differs from the Python binding (stale build?)")
    _lib = handle
    return _lib


def check(status):
    if status != 0:
        msg = lib().acino_last_error_string().decode("utf-8", "replace")
        codes = {-1: ValueError}
        raise codes.get(status, RuntimeError)(f"libacinoset_hip status {status}: {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("acinoset_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    lib()


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous float64/uint8 CUDA tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())
