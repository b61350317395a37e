"""ctypes binding of libfcosdet_hip.so (include/fcosdet.h).  Loading fails loudly; there is no fallback."""
from __future__ import annotations

import ctypes as C
import os

FD_MAX_SEG = 8
ACT_NONE, ACT_RELU, ACT_SILU, ACT_EXP, ACT_SIGMOID = 0, 1, 2, 3, 4
CONV_GENERIC, CONV_STEM = 0, 1
PREC_F32, PREC_F16X3 = 0, 1
TILES = {1: (128, 128), 2: (128, 64), 3: (64, 128), 4: (64, 64), 5: (128, 32), 6: (128, 96),
         7: (128, 128), 8: (128, 64), 9: (64, 128),   # 7-9: single-LDS-buffer variants
         10: (256, 128), 11: (256, 128),             # 8-wave tile (11: single LDS buffer)
         12: (128, 96),                              # 128x96, single LDS buffer
         13: (128, 128),                             # 128x128 with the 3x3 input patch staged in LDS (3x3 stride-1 'same' only)
         15: (64, 64)}                               # wave-autonomous 64x64 tiles, one wave per workgroup (1x1 stride-1 layers, needs w_frag)
PATCH_TILE = 13
WAVE_TILE = 15
NARROW_TILE = 17    # 3x3 stride-1 pad-1 convs with Cout <= 8 on the vector unit (fd_conv_narrow.hip; ops.pack_conv_weight_narrow); not a member of TILES
F16K64_TILE = 18    # FD_PREC_F16 on K-tiles of 64 channels (fd_conv_f16.hip; ops.pack_conv_weight_f16k64); not a member of TILES
WINO4_TILE = 16     # Winograd F(4x4, 3x3) kernel (own weight packing: ops.pack_conv_weight_wino4); not a member of TILES
PREC_F32, PREC_F16X3, PREC_F16 = 0, 1, 2     # include/fcosdet.h FD_PREC_*
WINO_TILE = 14      # Winograd F(2x2, 3x3) kernel (own weight packing: ops.pack_conv_weight_wino); not a member of TILES

_HERE = os.path.dirname(os.path.abspath(__file__))
# FD_LIB: another build of the SAME library (development A/B runs: tools/gpu_ab.sh keeps its variants outside the package and never overwrites the product
# library).  It must export every entry point below -- there is still no fallback of any kind.
LIB_PATH = os.environ.get("FD_LIB") or os.path.join(_HERE, "csrc", "libfcosdet_hip.so")


class FdError(RuntimeError):
    """Error of the HIP path.  `rc` is the library's numeric return code (FD_E_INVAL -1, FD_E_UNSUPPORTED -2, FD_E_LAUNCH -3) when the
    error came out of a C-ABI call, else None (host-side contract violations)."""

    def __init__(self, msg: str = "", rc=None):
        super().__init__(msg)
        self.rc = rc


E_INVAL, E_UNSUPPORTED, E_LAUNCH = -1, -2, -3


class Segs(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("batch", C.c_int32), ("H", C.c_int32 * FD_MAX_SEG), ("W", C.c_int32 * FD_MAX_SEG),
                ("m_start", C.c_int32 * (FD_MAX_SEG + 1))]

    @staticmethod
    def make(batch: int, hw) -> "Segs":
        s = Segs()
        s.nseg, s.batch = len(hw), batch
        m = 0
        for i, (h, w) in enumerate(hw):
            s.H[i], s.W[i], s.m_start[i] = h, w, m
            m += batch * h * w
        for i in range(len(hw), FD_MAX_SEG + 1):
            s.m_start[i] = m
        return s

    @property
    def rows(self) -> int:
        return self.m_start[self.nseg]

    def level_hw(self):
        return [(self.H[i], self.W[i]) for i in range(self.nseg)]


class B2BParams(C.Structure):
    """fd_b2b_params (include/fcosdet.h): two 1x1 convs back to back in one launch."""
    _fields_ = [("x", C.c_void_p), ("w1_frag", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
                ("w2_frag", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p), ("z", C.c_void_p),
                ("x_cs", C.c_int32), ("x_co", C.c_int32), ("res_cs", C.c_int32), ("res_co", C.c_int32), ("y_cs", C.c_int32), ("y_co", C.c_int32),
                ("z_cs", C.c_int32), ("z_co", C.c_int32),
                ("K1", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32), ("act1", C.c_int32), ("act2", C.c_int32), ("reserved0", C.c_int32),
                ("rows", C.c_int64)]


class ConvParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("res", C.c_void_p),
                ("y", C.c_void_p),
                ("x_cs", C.c_int32), ("x_co", C.c_int32), ("res_cs", C.c_int32), ("res_co", C.c_int32),
                ("y_cs", C.c_int32), ("y_co", C.c_int32),
                ("Cin", C.c_int32), ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("dil", C.c_int32),
                ("act", C.c_int32), ("act_c0", C.c_int32), ("mode", C.c_int32), ("tile", C.c_int32), ("tag", C.c_int32), ("ksplit", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("precision", C.c_int32), ("res_mode", C.c_int32),
                ("seg_param", C.c_float * FD_MAX_SEG), ("segs", Segs),
                ("out_H", C.c_int32), ("out_W", C.c_int32), ("sc_sy", C.c_int32), ("sc_sx", C.c_int32), ("sc_oy", C.c_int32),
                ("sc_ox", C.c_int32), ("sc_H", C.c_int32), ("sc_W", C.c_int32),
                ("gate", C.c_void_p), ("gate_cs", C.c_int32), ("reserved0", C.c_int32), ("w_frag", C.c_void_p),
                ("gn_stats", C.c_void_p), ("gn_groups", C.c_int32), ("gate_act", C.c_int32), ("gate_b", C.c_void_p),
                ("x2", C.c_void_p), ("x2_cs", C.c_int32), ("x2_co", C.c_int32), ("x2_Cin", C.c_int32), ("x2_stride", C.c_int32), ("x2_H", C.c_int32),
                ("x2_W", C.c_int32), ("wg_first", C.c_int32), ("wg_count", C.c_int32), ("sk_wgs", C.c_int32), ("io_f16", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("out", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("mode", C.c_int32), ("reserved", C.c_int32)]


class WgradParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p),
                ("x_cs", C.c_int32), ("x_co", C.c_int32), ("dy_cs", C.c_int32), ("dy_co", C.c_int32),
                ("Cin", C.c_int32), ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("dil", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("nsplit", C.c_int32), ("layout", C.c_int32),
                ("scale", C.c_void_p), ("segs", Segs), ("precision", C.c_int32), ("io_f16", C.c_int32)]


_lib = None

_P, _I, _F, _D, _L = C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_int64
_SIGS = {
    "fd_version": (_I, []),
    "fd_last_error": (C.c_char_p, []),
    "fd_conv2d_nhwc_f32": (_I, [C.POINTER(ConvParams), _P]),
    "fd_conv_narrow_nco": (_I, [_I]),
    "fd_conv_workgroups": (_I, [C.POINTER(ConvParams)]),
    "fd_conv_workgroups_live": (_I, [C.POINTER(ConvParams)]),
    "fd_conv1x1_b2b_f32": (_I, [C.POINTER(B2BParams), _P]),
    "fd_conv_workspace_bytes": (_L, [_L, _I, _I]),
    "fd_conv_sk_workspace_bytes": (_L, [_I]),
    "fd_mbconv_pool_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "fd_mbconv_expand_dw_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fd_se_gate_from_pool": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "fd_conv_wgrad_workspace_bytes": (_L, [_L, _I, _I, _I, _I]),
    "fd_pack_conv_weight_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fd_pack_conv_weights_batch_f32": (_I, [_P, _I, _L, _P]),
    "fd_wino4_weight_bytes": (_L, [_I, _I]),
    "fd_wino4_pack_weights_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fd_conv_weight_wave_bytes": (_L, [_I, _I]),
    "fd_pack_conv_weight_wave_f32": (_I, [_P, _P, _I, _I, _P]),
    "fd_wino_weight_bytes": (_L, [_I, _I]),
    "fd_wino_pack_weights_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fd_conv2d_bwd_weight_f32": (_I, [C.POINTER(WgradParams), _P]),
    "fd_stem7x7_nhwc4": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fd_stem7x7_pool_nhwc4": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fd_stem7x7_nchw3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fd_nchw3_to_nhwc4": (_I, [_P, _P, _I, _I, _I, _P]),
    "fd_nhwc_to_nchw": (_I, [_P, _I, _I, _P, _I, _I, _I, _P]),
    "fd_preprocess_u8_nhwc4": (_I, [_P, _P, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P]),
    "fd_boxes_rescale_xywh": (_I, [_P, _L, _F, _P]),
    "fd_maxpool_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fd_upsample2x_add_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fd_dwconv3x3_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(Segs), _P]),
    "fd_dwconv2d_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fd_dwconv_dilated_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(Segs), _P]),
    "fd_stem_conv_nhwc4": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fd_collate_u8_nhwc4": (_I, [_P, _P, _P, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P]),
    "fd_dwconv3x3_wgrad_workspace_bytes": (_L, [C.POINTER(Segs), _I]),
    "fd_dwconv3x3_bwd_weight_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _P, _I, C.POINTER(Segs), _P, _P]),
    "fd_groupnorm_workspace_bytes": (_L, [C.POINTER(Segs), _I]),
    "fd_groupnorm_act_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _I, C.POINTER(Segs), _P, _P]),
    "fd_groupnorm_from_rowstats": (_I, [_P, _I, _I, _F, _P, _P, C.POINTER(Segs), _P, _P, _P]),
    "fd_groupnorm_apply_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _I, C.POINTER(Segs), _P, _P]),
    "fd_groupnorm_stats_nhwc": (_I, [_P, _I, _I, _I, _I, _F, _P, _P, C.POINTER(Segs), _P, _P, _P]),
    "fd_coef_apply_nhwc": (_I, [_P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I, C.POINTER(Segs), _P]),
    "fd_dwconv3x3_gn_nhwc": (_I, [_P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, C.POINTER(Segs), _P]),
    "fd_groupnorm_bwd_workspace_bytes": (_L, [C.POINTER(Segs), _I]),
    "fd_groupnorm_act_bwd_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, C.POINTER(Segs), _P, _P,
                                       _P]),
    "fd_se_workspace_bytes": (_L, [_I, _I, _I]),
    "fd_se_scale_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "fd_se_bwd_workspace_bytes": (_L, [_I, _I, _I]),
    "fd_se_scale_bwd_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "fd_act_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _L, _I, _I, _F, _P]),
    "fd_act_bwd_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _L, _I, _I, _F, _P]),
    "fd_act_bwd_nhwc_h": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _L, _I, _I, _F, _P]),
    "fd_maxpool_bwd_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fd_upsample2x_bwd_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fd_batchnorm_update_running": (_I, [_P, _L, _I, _F, _F, _P, _P, _P]),
    "fd_batchnorm_update_running_dev": (_I, [_P, _P, _I, _F, _F, _P, _P, _P]),
    "fd_batchnorm_sync_fwd_nhwc": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _L, _I, _F, _I, _I, _P, _D, _P, _P]),
    "fd_batchnorm_sync_bwd_nhwc": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _L, _I, _F, _I, _I, _P, _D, _P, _P, _P]),
    "fd_fcos_decode": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, C.POINTER(Segs), C.POINTER(_I), _P, _P, _P, _P]),
    "fd_topk_workspace_bytes": (_L, [_I, _I, _I]),
    "fd_fcos_topk": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "fd_nms_workspace_bytes": (_L, [_I, _I]),
    "fd_batched_nms": (_I, [_P, _P, _P, _I, _I, _F, _D, _P, _P, _P, _P, _P, _P, _P]),
    "fd_box_nms_plus1": (_I, [_P, _P, _P, _I, _I, _F, _I, _P, _P, _P]),
    "fd_pairwise_iou": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "fd_clip_boxes": (_I, [_P, _L, _I, _I, _P]),
    "fd_pack_detections": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "fd_unpack_detections": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "fd_ltrb_iou_loss_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "fd_ltrb_iou_loss_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "fd_focal_workspace_bytes": (_L, [_I]),
    "fd_focal_loss_fwd": (_I, [_P, _P, _I, _I, _I, _F, _F, _P, _P, _P]),
    "fd_focal_loss_bwd": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _P, _P]),
    "fd_bce_logits_loss_fwd": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "fd_bce_logits_loss_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "fd_fcos_gen_targets": (_I, [_P, _P, _I, C.POINTER(Segs), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _F, _P, _P, _P, _P]),
}
EXPORTS = tuple(_SIGS)


def lib() -> C.CDLL:
    """Load libfcosdet_hip.so once.  Raises FdError when it has not been built (python __graft_entry__.py / make)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch bundles its own libamdhip64: import it first so this library binds to the SAME HIP runtime
        # (two runtimes in one process => "no ROCm-capable device" / foreign stream handles)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = lib().fd_last_error().decode(errors="replace")
        raise FdError(f"{what or 'libfcosdet_hip'} failed ({code}): {msg}", rc=int(code))
