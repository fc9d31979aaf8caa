"""ctypes binding of include/dca_hip.h.  There is NO fallback: if libdca_hip.so is missing or does not
export every symbol the header declares, importing the ops raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdca_hip.so")

_p, _i, _l, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double

ABI_VERSION = 19  # == DCA_ABI_VERSION of include/dca_hip.h

# name -> (restype, argtypes); mirrors include/dca_hip.h one to one
SIGNATURES = {
    "dca_abi_version": (_i, []),
    "dca_gwc_volume_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dca_gwc_volume_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dca_concat_volume_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_concat_volume_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_cost_volume_fwd": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dca_softargmin_fwd": (_i, [_p, _p, _i, _i, _l, _i, _p]),
    "dca_softargmin_bwd": (_i, [_p, _p, _p, _i, _i, _l, _i, _p]),
    "dca_up_softargmin_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_up_softargmin_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_conv3d_prep_weight": (_i, [_p, _p] + [_i] * 9 + [_p]),
    "dca_conv3d_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f] + [_i] * 16 + [_p]),
    "dca_conv1_x3_weight_bytes": (_l, [_i]),
    "dca_conv1_x3_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_conv1_x3_forward": (_i, [_p] * 8 + [_f] + [_i] * 6 + [_l, _p]),
    "dca_conv3d_prep_many": (_i, [_p, _i, _p]),
    "dca_conv3d_x3_weight_bytes": (_l, [_i, _i]),
    "dca_conv3d_x3_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dca_conv3d_x3_forward": (_i, [_p] * 7 + [_f] + [_i] * 6 + [_p]),
    "dca_conv3d_x3_stats_chunks": (_l, [_i] * 5),
    "dca_conv3d_x3_forward_stats": (_i, [_p] * 4 + [_i] * 6 + [_p]),
    "dca_bn_finalize_centered": (_i, [_p, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _p, _i, _i, _p]),
    "dca_conv1_x3_stats_chunks": (_l, [_i, _l]),
    "dca_conv1_x3_forward_stats": (_i, [_p] * 5 + [_i] * 6 + [_l, _p]),
    "dca_deconv3d_x3_stats_chunks": (_l, [_i] * 4),
    "dca_deconv3d_x3_forward_stats": (_i, [_p] * 4 + [_i] * 6 + [_p]),
    "dca_deconv3d_x3_forward": (_i, [_p] * 7 + [_f] + [_i] * 6 + [_p]),
    "dca_conv3d_wgrad_workspace": (_l, [_i] * 8),
    "dca_conv3d_wgrad_x3_workspace": (_l, [_i] * 6),
    "dca_conv3d_wgrad_x3": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_l, _l, _p]),
    "dca_cmax_f32": (_i, [_p, _i, _i, _l, _p, _p]),
    "dca_conv3d_x2_weight_bytes": (_l, [_i, _i]),
    "dca_conv3d_x2_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "dca_conv3d_x2_forward": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _f, _p] + [_i] * 6 + [_p]),
    "dca_conv3d_x2_stats_chunks": (_l, [_i] * 5),
    "dca_conv3d_x2_forward_stats": (_i, [_p, _i, _p, _p, _p, _p] + [_i] * 6 + [_p]),
    "dca_conv3d_s2x2_weight_bytes": (_l, [_i, _i]),
    "dca_conv3d_s2x2_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "dca_conv3d_s2x2_out_slots": (_l, [_i] * 5),
    "dca_conv3d_s2x2_forward": (_i, [_p, _p, _p, _p, _p, _p, _f, _p, _p] + [_i] * 6 + [_p]),
    "dca_conv3d_wgrad_x2_workspace": (_l, [_i] * 6),
    "dca_conv3d_wgrad_s2_x2_workspace": (_l, [_i] * 6),
    "dca_conv3d_wgrad_s2_x2": (_i, [_p] * 6 + [_i] * 6 + [_l, _l, _p]),
    "dca_conv3d_wgrad_x2": (_i, [_p, _i, _p, _p, _i, _p, _p, _p] + [_i] * 6 + [_l, _l, _p]),
    "dca_conv3d_wgrad": (_i, [_p, _p, _p, _p] + [_i] * 11 + [_l, _l, _p]),
    "dca_conv3d_c1_gather": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dca_conv3d_c1_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_conv3d_c1_expand": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dca_conv3d_c1_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_bn_num_chunks": (_i, [_i, _l]),
    "dca_bn_pack_chunks": (_i, [_i, _l]),
    "dca_cmax_exps": (_i, [_p, _i, _i, _p, _p]),
    "dca_bn_apply_pack": (_i, [_p, _p, _p, _p, _i, _i, _l, _f, _p, _p, _p, _p, _p, _p]),
    "dca_bn_backward_pack": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _f, _i, _p]),
    "dca_bn_stats": (_i, [_p, _p, _i, _i, _l, _p]),
    "dca_bn_finalize": (_i, [_p, _i, _d, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _i, _p, _i, _i, _p]),
    "dca_bn_apply": (_i, [_p, _p, _p, _p, _p, _i, _i, _l, _f, _p, _p, _p]),
    "dca_bn_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _l, _f, _i, _p, _p]),
    "dca_avgpool3d_fwd": (_i, [_p, _p, _l, _i, _i, _i, _p]),
    "dca_avgpool3d_bwd": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _p]),
    "dca_trilinear_fwd": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "dca_trilinear_bwd": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "dca_context_inject_fwd": (_i, [_p] * 8 + [_i, _i, _i, _l, _p]),
    "dca_context_inject_bwd": (_i, [_p] * 12 + [_i, _i, _i, _l, _p]),
    "dca_disp_attention_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _l, _p]),
    "dca_disp_attention_bwd": (_i, [_p] * 7 + [_i, _i, _i, _l, _p]),
    "dca_conv3d_lp_weight_bytes": (_l, [_i, _i]),
    "dca_conv3d_lp_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "dca_conv3d_lp_forward": (_i, [_p] * 7 + [_f] + [_i] * 9 + [_p]),
    "dca_conv1_lp_weight_bytes": (_l, [_i, _i]),
    "dca_conv1_lp_prep_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dca_conv1_lp_forward": (_i, [_p] * 8 + [_f, _i, _i, _i, _i, _l, _i, _i, _p]),
    "dca_conv3d_forward_mixed": (_i, [_p] * 7 + [_f] + [_i] * 12 + [_p]),
    "dca_conv3d_s2_lp_weight_bytes": (_l, [_i]),
    "dca_conv3d_s2_lp_prep_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "dca_conv3d_s2_lp_forward": (_i, [_p] * 5 + [_f] + [_i] * 7 + [_p]),
    "dca_deconv3d_lp_forward": (_i, [_p] * 7 + [_f] + [_i] * 7 + [_p]),
    "dca_avgpool3d_lp_fwd": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "dca_trilinear_up2_lp_fwd": (_i, [_p, _p, _l, _i, _i, _i, _i, _p]),
    "dca_convex_up4_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "dca_convex_up4_bwd": (_i, [_p] * 6 + [_i, _i, _i, _p]),
    "dca_focal_loss_workspace": (_l, [_i, _i, _l]),
    "dca_focal_loss_fwd": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _l, _f, _p]),
    "dca_focal_loss_bwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _l, _f, _p]),
}

_lib = None


def load():
    """Loads the HIP library and binds every C-ABI symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the DCANet hot path has no CPU/PyTorch fallback. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype, fn.argtypes = res, args
    got = lib.dca_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} was built from another version of include/dca_hip.h (ABI {got}, this package "
                           f"binds ABI {ABI_VERSION}): rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
    _lib = lib
    return lib
