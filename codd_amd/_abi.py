"""ctypes binding of libcodd_hip.so (include/codd_hip.h).

The product path has NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes as C
import os

from . import build as _build

c_float_p = C.POINTER(C.c_float)


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ctot", C.c_int), ("coff", C.c_int)]


class XsView(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("c8", C.c_int), ("hp", C.c_int), ("wp", C.c_int), ("bt", C.c_int),
                ("bl", C.c_int), ("o8", C.c_int), ("terms", C.c_int)]


class ConvParams(C.Structure):
    _fields_ = [
        ("in0", View), ("in1", View), ("C0", C.c_int), ("C1", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int),
        ("wpacked", C.c_void_p), ("bias", C.c_void_p),
        ("res1", View), ("res2", View), ("post", View),
        ("out", C.c_void_p), ("out_ctot", C.c_int), ("out_coff", C.c_int),
        ("Cout", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int), ("sy", C.c_int), ("sx", C.c_int),
        ("pad_t", C.c_int), ("pad_l", C.c_int), ("dil_y", C.c_int), ("dil_x", C.c_int),
        ("act", C.c_int), ("store_mode", C.c_int), ("mb", C.c_int), ("npb", C.c_int), ("nw", C.c_int), ("ck", C.c_int), ("layout", C.c_int),
        ("terms", C.c_int), ("pgw", C.c_int), ("cgw", C.c_int),
        ("xs", C.c_void_p), ("xs_c8", C.c_int), ("xs_hp", C.c_int), ("xs_wp", C.c_int),
        ("xs_bt", C.c_int), ("xs_bl", C.c_int), ("xs_o8", C.c_int),
        ("xso", C.c_void_p), ("xso_c8", C.c_int), ("xso_hp", C.c_int), ("xso_wp", C.c_int), ("xso_bt", C.c_int),
        ("xso_bl", C.c_int), ("xso_o8", C.c_int), ("xso_terms", C.c_int), ("ksplit", C.c_int), ("dil2", C.c_int), ("gate", C.c_int),
    ]


ACT = dict(none=0, lrelu=1, relu=2, sigmoid=3, tanh=4, mish=5, relu_ch0=6)




class RollParams(C.Structure):
    _fields_ = [("in0", View), ("in1", View), ("C0", C.c_int), ("C1", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("mode", C.c_int),
                ("wA", C.c_void_p), ("bA", C.c_void_p), ("wB", C.c_void_p), ("bB", C.c_void_p),
                ("actA", C.c_int), ("actB", C.c_int), ("residual", C.c_int),
                ("out", C.c_void_p), ("out_ctot", C.c_int), ("out_coff", C.c_int), ("cout_store", C.c_int),
                ("rh", C.c_int)]


_i, _f, _p, _ll = C.c_int, C.c_float, C.c_void_p, C.c_longlong

# name -> (restype, argtypes); must list every function declared in include/codd_hip.h
SIGNATURES = {
    "codd_abi_version": (_i, []),
    "codd_conv2d": (_i, [C.POINTER(ConvParams), _p]),
    "codd_conv2d_check": (_i, [C.POINTER(ConvParams)]),
    "codd_conv2d_multi": (_i, [C.POINTER(ConvParams), _i, _p]),
    "codd_roll_packed_size": (_ll, [_i, _i, _i]),
    "codd_roll_pack_weights": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "codd_conv_roll": (_i, [C.POINTER(RollParams), _p]),
    "codd_conv2d_packed_size": (_ll, [_i] * 6),
    "codd_conv2d_pack_weights": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "codd_tile_costvol_argmin": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _i, _i, _i, _p]),
    "codd_tile_warp_cost": (_i, [_p, _p, _i, _i, _i, _i, View, View, _i, _p, _p, _p]),
    "codd_hyp_upsample": (_i, [View, _i, _i, _i, _f, _p, _i, _i, _p]),
    "codd_hyp_select": (_i, [_p, View, View, _i, _i, _i, _p, _i, _i, _p]),
    "codd_instnorm": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p]),
    "codd_instnorm_xs": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p, XsView, _p]),
    "codd_conv2d_pack_weights_ex": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _ll, _ll, _f, _p]),
    "codd_allpairs_corr": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "codd_allpairs_corr_scratch": (_ll, [_i, _i, _i, _i]),
    "codd_avgpool2": (_i, [_p, _i, _i, _i, _p, _p]),
    "codd_induced_flow": (_i, [_p, _p, _i, _i, _i, _f, _f, _f, _f, _p, _p]),
    "codd_context_split": (_i, [_p, _i, _i, _p, _p, _p]),
    "codd_se3_identity": (_i, [_p, _ll, _p]),
    "codd_corr_lookup": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "codd_raft_geometry": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p]),
    "codd_se3_gn_step": (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f, _p, _p]),
    "codd_se3_gn_scratch": (_ll, [_i, _i, _i, _i]),
    "codd_set_option": (_i, [_i, _i]),
    "codd_get_option": (_i, [_i]),
    "codd_se3_gn_step_heads": (_i, [_p, XsView, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _i, _f, _f, _p, _p, _p]),
    "codd_cvx_upsample": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "codd_cvx_upsample_se3_weight": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "codd_disp_to_depth": (_i, [_p, _ll, _f, _p, _p]),
    "codd_subsample": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "codd_splat": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f,
                        _p, _p, _p, _p]),
    "codd_splat_scratch": (_ll, [_i, _i, _i, _f]),
    "codd_resize_bilinear": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p]),
    "codd_resize_bilinear_add": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p, _p]),
    "codd_add_relu": (_i, [_p, _p, _ll, _i, _p, _p]),
    "codd_copy_many": (_i, [_p, _p, _p, _i, _p]),
    "codd_timestamp": (_i, [_p, _p]),
    "codd_gru_gate_zr": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "codd_gru_gate_q": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "codd_gru_gate_zr_xs": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, XsView, _p]),
    "codd_gru_gate_q_xs": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, XsView, _p]),
    "codd_fusion_cues_lr": (_i, [_p] * 6 + [_i] * 7 + [_p, _p, _i, _i, _p]),
    "codd_fusion_cues_fr": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "codd_fusion_forget": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "codd_disp_metrics": (_i, [_p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p]),
    "codd_raft_geometry_lookup": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p]),
    "codd_raft_geometry_lookup_xs": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _p, XsView, XsView, _p]),
    "codd_conv2d_packed_size_quad": (C.c_longlong, [_i, _i, _i, _i, _i, _i]),
    "codd_conv2d_pack_weights_quad": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "codd_split_bf16_bytes": (_ll, [_i] * 5),
    "codd_split_bf16": (_i, [View, _i, View, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "codd_conv2d_packed_bytes_bf16": (_ll, [_i] * 7),
    "codd_conv2d_pack_weights_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _ll, _ll, _f, _p]),
    "codd_fusion_select": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "codd_gt_motion": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "codd_tepe_metrics": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p]),
    "codd_sceneflow_metrics": (_i, [_p] * 6 + [_i] * 5 + [_f] * 7 + [_p, _p, _p]),
    "codd_preprocess": (_i, [_p, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _i, _p, _p]),
    "codd_fusion_blend": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
}

ABI_VERSION = 12  # CODD_ABI_VERSION of include/codd_hip.h
_lib = None
LOADED = None  # path of the library this process loaded
MISSING = []


class CoddHipError(RuntimeError):
    pass


def lib_path():
    """The library this process loads.  CODD_LIB_AB (dev, same-lease A/B of two builds: tools/ab_lib.sh) names a PREBUILT
    libcodd_hip.so that is loaded instead of the in-tree one WITHOUT the missing-or-stale rebuild check -- only its ABI
    version is compared -- so its use is announced on stderr and bench.py records the path in its JSON line."""
    ab = os.environ.get("CODD_LIB_AB")
    if ab:
        import sys
        print(f"codd_amd: WARNING: CODD_LIB_AB is set -- loading the prebuilt {ab} instead of the in-tree library; it is NOT "
              f"checked against the sources of this tree", file=sys.stderr, flush=True)
        return ab
    return _build.LIB


def load():
    """Load (building if the sources are newer) libcodd_hip.so; raises if unavailable."""
    global _lib, LOADED
    if _lib is not None:
        return _lib
    path = lib_path()
    # One process at a time decides / rebuilds (torchrun starts every rank at once on a fresh checkout); a rebuild
    # that is needed but fails is an error -- a stale library is never loaded silently against newer sources.  The
    # lock file is only opened when a build is needed, so a current library loads from a read-only install.
    if path == _build.LIB and (not os.path.exists(path) or _build.needs_build()):
        import fcntl
        try:
            lock = open(os.path.join(os.path.dirname(path), ".build.lock"), "w")
        except OSError as e:
            raise CoddHipError(f"libcodd_hip.so is missing or older than its sources and {os.path.dirname(path)} is "
                               f"not writable: {e}") from e
        with lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not os.path.exists(path) or _build.needs_build():  # (another rank may have built it meanwhile)
                    try:
                        _build.build(verbose=False)
                    except Exception as e:  # pragma: no cover
                        raise CoddHipError(f"libcodd_hip.so is missing or older than its sources and the rebuild "
                                           f"failed: {e}") from e
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # recorded; tests/test_abi.py asserts this list is empty, and calling the symbol raises
            MISSING.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if lib.codd_abi_version() != ABI_VERSION:
        raise CoddHipError("ABI version mismatch")
    _lib, LOADED = lib, path
    return lib


OPTIONS = dict(gn_q4=0, gn_builder=1)  # CODD_OPT_* of include/codd_hip.h


def set_option(name, value):
    """codd_set_option: the library's only process-wide state (defaults = the shipped configuration).  Returns the previous value."""
    rc = load().codd_set_option(OPTIONS[name], int(value))
    if rc < 0:
        raise CoddHipError(f"codd_set_option({name}, {value}) rejected")
    return rc


def check(rc, what):
    if rc != 0:
        raise CoddHipError(f"{what} failed with code {rc}")
