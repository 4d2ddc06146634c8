"""ctypes binding of libsmx_hip.so (the C-ABI declared in include/smx.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing the
import fails loudly with the build command; if no HIP device is usable every compute call raises
SmxError (SMX_E_HIP).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# The product library.  Diagnostic builds of the same C-ABI (tools/v3_stamps.sh: in-kernel stamps) are
# loaded only when BOTH SMX_LIB_PATH and SMX_ALLOW_LIB_OVERRIDE=1 are set, so that one stray variable
# cannot make the tests run against a different binary.
SO_PATH = os.path.join(_PKG, "_build", "libsmx_hip.so")
if os.environ.get("SMX_LIB_PATH"):
    if os.environ.get("SMX_ALLOW_LIB_OVERRIDE") != "1":
        raise ImportError("SMX_LIB_PATH is set but SMX_ALLOW_LIB_OVERRIDE=1 is not: refusing to load a "
                          "non-product build of libsmx_hip.so")
    SO_PATH = os.environ["SMX_LIB_PATH"]
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "smx.h")


class SmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"smx error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """smx_params (SystemIncludes.h:7-24 of the reference as runtime values)."""
    _fields_ = [("r_w", C.c_double), ("g_w", C.c_double), ("b_w", C.c_double),
                ("alpha", C.c_double), ("th_color", C.c_int), ("th_grad", C.c_int),
                ("radius", C.c_int), ("eps", C.c_double), ("d_lr", C.c_int)]


class StageMs(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("upload", "guidance", "aggregation", "wta", "finish", "download", "total")] + \
               [("calls", C.c_int), ("dropped", C.c_int)]


class PairOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "best_l", "best_r", "dmap_l", "dmap_r", "mean_l", "mean_r", "occlusion", "filled",
        "cost_l", "cost_r", "agg_l", "agg_r")]


def build(verbose=False):
    """Compile libsmx_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_PKG, "csrc")] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return SO_PATH


_vp, _i, _i64, _f, _sz, _u64, _u32 = (C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t,
                                      C.c_uint64, C.c_uint32)
_PP = C.POINTER(Params)

# name -> (restype, argtypes).  Mirrors include/smx.h one to one (tests/test_capi.py checks it).
SIGNATURES = {
    "smx_default_params": (None, [_PP]),
    "smx_last_error": (C.c_char_p, []),
    "smx_version": (C.c_char_p, []),
    "smx_device_count": (_i, []),
    "smx_rgb_to_grayscale": (_i, [_PP, _vp, _i64, _i, _vp]),
    "smx_compute_cost": (_i, [_PP, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "smx_integral": (_i, [_vp, _vp, _i, _i]),
    "smx_compute_guided_filter": (_i, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "smx_detect_occlusion": (_i, [_PP, _vp, _vp, _i, _i, _i]),
    "smx_fill_occlusion": (_i, [_vp, _i, _i, _f]),
    "smx_filter": (_i, [_PP, _vp, _i, _i, _vp, _vp]),
    "smx_dev_filter": (_i, [_PP, _vp, _i, _i, _vp, _vp, _vp]),
    "smx_stereo_pair": (_i, [_PP, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(PairOut)]),
    "smx_create": (_i, [_PP, _i, _i, _i, C.POINTER(_vp)]),
    "smx_ctx_stereo_pair": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(PairOut)]),
    "smx_destroy": (_i, [_vp]),
    "smx_ctx_set_agg_path": (_i, [_vp, _i]),
    "smx_ctx_stereo_pair_async": (_i, [_vp, _vp, _vp, _i, _i]),
    "smx_ctx_wait": (_i, [_vp, C.POINTER(PairOut), C.POINTER(PairOut)]),
    "smx_dev_rgb_to_grayscale": (_i, [_PP, _vp, _i64, _i, _vp, _vp]),
    "smx_dev_cost_volume": (_i, [_PP, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "smx_dev_integral": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "smx_agg_workspace_bytes": (_sz, [_i, _i, _i]),
    "smx_agg_workspace_bytes_for": (_sz, [_PP, _i, _i, _i]),
    "smx_dev_aggregate_wta": (_i, [_PP, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "smx_dev_aggregate_wta_pair": (_i, [_PP, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "smx_dev_aggregate_wta_pair_cost": (_i, [_PP, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "smx_dev_agg_status": (_i, [_vp]),
    "smx_dev_agg_fallback": (_i, [_vp, C.POINTER(_i)]),
    "smx_dev_init_keys": (_i, [_vp, _i64, _vp]),
    "smx_dev_apply_keys": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "smx_dev_init_wta": (_i, [_vp, _vp, _i64, _vp]),
    "smx_dev_detect_occlusion": (_i, [_PP, _vp, _vp, _i, _i, _i, _vp]),
    "smx_dev_fill_occlusion": (_i, [_vp, _i, _i, _f, _vp]),
    "smx_dev_finish_pair": (_i, [_PP, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "smx_pack_key": (_i64, [_f, _u32]),
    "smx_unpack_key": (None, [_i64, C.POINTER(_f), C.POINTER(_u32)]),
    "smx_set_agg_path": (_i, [_i]),
    "smx_last_agg_path": (_i, []),
    "smx_set_max_slices_per_launch": (_i, [_i]),
    "smx_set_keys_fresh": (_i, [_i]),
    "smx_last_agg_chunk": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "smx_agg_geometry": (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "smx_set_timing": (_i, [_i]),
    "smx_last_agg_ms": (_i, [C.POINTER(_f), C.POINTER(_i)]),
    "smx_stage_times": (_i, [C.POINTER(StageMs)]),
}

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME
    libamdhip64.so.7).  Two HIP runtimes in one process cannot both own the GPU, and device buffers,
    streams and RCCL come from torch, so when torch is installed its runtime is mapped first and
    libsmx_hip.so (NEEDED libamdhip64.so.7) binds to that same copy.  Without torch (e.g. the C++
    host build) the system runtime under /opt/rocm is used.  SMX_HIP_RUNTIME=system skips this."""
    if os.environ.get("SMX_HIP_RUNTIME", "") == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return None
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass
    return None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C stereo_matching_cuda_amd/csrc` (needs /opt/rocm/bin/hipcc). "
                "There is no CPU fallback.")
        _preload_torch_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SmxError(rc, lib().smx_last_error().decode("utf-8", "replace"))


def default_params():
    p = Params()
    lib().smx_default_params(C.byref(p))
    return p
