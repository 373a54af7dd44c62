"""ctypes binding of libgeoa3_hip.so (the C ABI declared in include/geoa3_hip.h).

There is NO fallback: if the shared object is missing or a symbol is absent the import of the
product path raises.  Build it with ``python -m geoa3_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GEOA3_LIB_PATH: a variant build of the same sources for an A/B run of tools/ (python -m geoa3_amd.build --variant DIR ...)
LIB_PATH = os.environ.get("GEOA3_LIB_PATH") or os.path.join(_HERE, "lib", "libgeoa3_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
vp = C.c_void_p


class GeoArgs(C.Structure):
    """struct geoa3_geo_args"""
    _fields_ = [("adv", vp), ("ori", vp), ("normal_ori", vp), ("kappa_ori", vp), ("d_ao", vp), ("i_ao", vp),
                ("d_oa", vp), ("i_oa", vp), ("knn_adv", vp), ("dkappa", vp),
                ("B", C.c_int32), ("N", C.c_int32), ("k", C.c_int32), ("Nr", C.c_int32), ("dis_type", C.c_int32),
                ("single_side", C.c_int32), ("w_dis", C.c_float), ("w_hd", C.c_float), ("w_curv", C.c_float),
                ("dis_loss", vp), ("hd_loss", vp), ("curv_loss", vp), ("constrain", vp), ("kappa_adv", vp),
                ("grad", vp), ("deterministic", C.c_int32), ("scratch", vp)]


class TnetWeights(C.Structure):
    """struct geoa3_tnet_weights"""
    _fields_ = [("K", C.c_int32)] + [(n, vp) for n in (
        "w1", "b1", "w2", "b2", "w3", "b3", "w3p", "w2t", "f1", "fb1", "f2", "fb2", "f3", "fb3",
        "f1t", "f2t", "f3t", "w3h")] + [("w3h_unscale", C.c_float), ("w2h", vp), ("w2h_unscale", C.c_float), ("w3h16", vp),
                                      ("w2th", vp), ("w2th_unscale", C.c_float)]


class PointNetWeights(C.Structure):
    """struct geoa3_pointnet_weights"""
    _fields_ = [("classes", C.c_int32), ("t3", TnetWeights), ("t64", TnetWeights)] + [(n, vp) for n in (
        "w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "w5", "b5", "w5p", "w4t", "w3t", "w2t",
        "f1", "fb1", "f2", "fb2", "f3", "fb3", "f1t", "f2t", "f3t", "w5h")] + [("w5h_unscale", C.c_float), ("w4h", vp),
                                                                                 ("w4h_unscale", C.c_float), ("w5h16", vp),
                                                                                 ("w4th", vp), ("w4th_unscale", C.c_float),
                                                                                 ("flags", C.c_int32)]


class Sa1Weights(C.Structure):
    """struct geoa3_sa1_weights"""
    _fields_ = [(n, vp) for n in ("w1", "b1", "w2", "b2", "w3", "b3")]


class Pn2SsgWeights(C.Structure):
    """struct geoa3_pn2ssg_weights"""
    _fields_ = [("classes", C.c_int32), ("sa1", Sa1Weights)] + [(n, vp) for n in (
        "sa2_wx", "sa2_wf", "sa2_b0", "sa2_wft", "sa2_w1", "sa2_b1", "sa2_w1t", "sa2_w2", "sa2_b2", "sa2_w2t",
        "sa3_wx", "sa3_wf", "sa3_b0", "sa3_wft", "sa3_w1", "sa3_b1", "sa3_w1t", "sa3_w2", "sa3_b2", "sa3_w2t",
        "f1", "fb1", "f1t", "f2", "fb2", "f2t", "f3", "fb3", "f3t", "images", "side")] + [("flags", C.c_int32)]


class AttackState(C.Structure):
    """struct geoa3_attack_state"""
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("classes", C.c_int32), ("targeted", C.c_int32),
                ("cls_loss_type", C.c_int32), ("confidence", C.c_float), ("inv_global_batch", C.c_float),
                ("gt", vp), ("target", vp), ("scale_const", vp), ("lower_bound", vp), ("upper_bound", vp),
                ("best_loss", vp), ("best_attack", vp), ("best_step", vp), ("best_bs", vp),
                ("iter_best_loss", vp), ("iter_best_score", vp), ("prev_constrain", vp), ("label", vp),
                ("cls_loss", vp), ("loss_n", vp), ("loss_hist", vp), ("last_label", vp)]


# name -> (restype, argtypes); every symbol include/geoa3_hip.h declares
SIGNATURES = {
    "geoa3_version": (C.c_int, []),
    "geoa3_strerror": (C.c_char_p, [C.c_int]),
    "geoa3_nn1_pair": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "geoa3_grid_nn1_pair": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "geoa3_knn": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "geoa3_knn_self_scratch_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "geoa3_knn_self": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "geoa3_kappa": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_geo_loss_grad": (C.c_int, [C.POINTER(GeoArgs), vp]),
    "geoa3_pointnet_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "geoa3_pointnet_forward": (C.c_int, [C.POINTER(PointNetWeights), vp, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_pointnet_backward": (C.c_int, [C.POINTER(PointNetWeights), vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_attack_head": (C.c_int, [C.POINTER(AttackState), vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "geoa3_attack_head_vote": (C.c_int, [C.POINTER(AttackState), vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]),
    "geoa3_attack_head_classify": (C.c_int, [C.POINTER(AttackState), vp, vp, C.c_int, vp, vp, vp]),
    "geoa3_attack_head_finish": (C.c_int, [C.POINTER(AttackState), vp, vp, vp, C.c_int, C.c_int, vp]),
    "geoa3_attack_update": (C.c_int, [C.POINTER(AttackState), vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float,
                                      C.c_float, C.c_float, vp]),
    "geoa3_attack_partial_step": (C.c_int, [C.POINTER(AttackState), vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                            C.c_float, C.c_float, C.c_float, C.c_int, vp]),
    "geoa3_attack_project": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "geoa3_attack_binary_update": (C.c_int, [C.POINTER(AttackState), vp]),
    "geoa3_attack_begin_search_step": (C.c_int, [C.POINTER(AttackState), vp, vp, vp, vp, vp, vp, vp]),
    "geoa3_pn2_furthest_point_sampling": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_pn2_gather_points": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_pn2_gather_points_grad": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_pn2_ball_query": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp]),
    "geoa3_pn2_ball_query_ex": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, C.c_int, vp]),
    "geoa3_pn2_furthest_point_sampling_ex": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "geoa3_pn2_group_points": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_pn2_group_points_grad": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_pn2_bias_relu": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_long, vp]),
    "geoa3_pn2_relu_grad": (C.c_int, [vp, vp, vp, C.c_long, vp]),
    "geoa3_pn2_bias_relu_max": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_long, C.c_int, vp, vp, vp]),
    "geoa3_pn2_bias_relu_max_grad": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_long, C.c_int, vp, vp]),
    "geoa3_pn2_sa1_forward": (C.c_int, [vp, vp, vp, C.POINTER(Sa1Weights), C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_pn2_sa1_backward": (C.c_int, [vp, vp, vp, C.POINTER(Sa1Weights), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                         vp, vp, vp]),
    "geoa3_pn2_sa1_scratch_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "geoa3_pn2ssg_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "geoa3_pn2ssg_images_bytes": (C.c_int64, []),
    "geoa3_pn2ssg_pack_images": (C.c_int, [C.POINTER(Pn2SsgWeights), vp, vp]),
    "geoa3_side_queue_create": (vp, []),
    "geoa3_side_queue_destroy": (None, [vp]),
    "geoa3_pn2ssg_forward": (C.c_int, [C.POINTER(Pn2SsgWeights), vp, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_pn2ssg_backward": (C.c_int, [C.POINTER(Pn2SsgWeights), vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_fps_sample": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "geoa3_knn_normal": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_local_frames": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_perp_jitter": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]),
    "geoa3_smoothness": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_sor_statistic": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_sor_select": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp]),
    "geoa3_debug_wide_bwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "geoa3_debug_wide_fwd": (C.c_int, [vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_pn2_group_points_grad_sums": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "geoa3_pn2_group_shift_relu": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "geoa3_pn2_shift_relu": (C.c_int, [vp, vp, C.c_long, C.c_int, vp]),
    "geoa3_pn2_shift_relu_grad": (C.c_int, [vp, vp, vp, vp, C.c_long, C.c_int, vp]),
    "geoa3_conv1x1_max64": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_long, C.c_int, C.c_int, vp]),
    "geoa3_conv1x1_onehot64": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_long, C.c_int, C.c_int, vp]),
    "geoa3_conv1x1": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, vp]),
    "geoa3_debug_pointnet_workspace_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                                                        C.POINTER(C.c_int64), C.c_int]),
    "geoa3_debug_grid_nn1_pair": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, vp]),
    "geoa3_debug_fc": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "geoa3_debug_conv_cm": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "geoa3_profile_enable": (C.c_int, [C.c_int]),
    "geoa3_profile_select": (C.c_int, [C.c_uint]),
    "geoa3_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.c_int]),
}

_lib = None


ENOSUPPORT = -3   # GEOA3_ENOSUPPORT
PN2_CONTRACT = 1   # GEOA3_PN2_CONTRACT
ABI_VERSION = 600  # GEOA3_ABI_VERSION of include/geoa3_hip.h this file mirrors (tests/test_abi.py holds the two together)


class Geoa3Error(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Geoa3Error("%s is missing: build it with `python -m geoa3_amd.build` "
                         "(the product path has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    have = lib.geoa3_version()
    if have != ABI_VERSION:
        raise Geoa3Error("%s was built for ABI version %d, this binding mirrors version %d of include/geoa3_hip.h: rebuild "
                         "it (`python -m geoa3_amd.build --force`)" % (LIB_PATH, have, ABI_VERSION))
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().geoa3_strerror(code).decode()
        raise Geoa3Error("%s failed: %s (%d)" % (what or "geoa3 call", msg, code))
