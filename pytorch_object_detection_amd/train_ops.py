"""Autograd bindings of the HIP kernels for the training forward (reference train.py:175-181).

Everything here works on the library's own layout: NHWC rows, i.e. 2-D tensors [rows, C] described by a segment table
(`Segs`: one segment per pyramid level).  An NCHW-shaped channels-last tensor IS such a buffer (zero-copy view), and the
five FCOS levels concatenated along the row axis are one buffer too, so a shared-weight head runs one launch per layer
over the whole pyramid exactly as at inference.

  conv_rows / conv_bn_act   Conv2d -> frozen BatchNorm2d -> (+ residual) -> ReLU as ONE launch of fd_conv2d_nhwc_f32
                            (BN folded into the epilogue).  Backward: ReLU mask on the incoming gradient (one
                            elementwise pass from the saved output), data gradient = the same conv kernel on dY with
                            flipped / transposed weights (stride-1 layers), weight gradient = fd_conv2d_bwd_weight_f32.
  dw_rows                   depthwise 3x3: fd_dwconv3x3_nhwc forward and data gradient, fd_dwconv3x3_bwd_weight_nhwc.
  groupnorm_rows            GroupNorm + ReLU / SiLU: fd_groupnorm_act_nhwc / fd_groupnorm_act_bwd_nhwc.

What the kernels do not cover falls back to stock PyTorch-ROCm ops on the GPU: the data gradient of strided layers,
dense layers with Cin % 32 != 0 (the 7x7 stem when it is trainable), BatchNorm in training mode or with trainable
affine parameters, and SiLU after a BatchNorm (its derivative needs the pre-activation, which the fused epilogue does
not keep).  Narrow outputs (class / centre-ness / box logits) are padded to 32 channels by `conv_rows(pad_out=True)`.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from ._lib import FdError, Segs
from .ops import ACT_NONE, ACT_RELU, ACT_SILU, Rows

# Under torch.autocast (the reference trains with AMP when cfg['model']['amp'] is set, train.py:175) the HIP nodes keep
# computing in fp32: inputs are cast up on entry, autocast is off inside, gradients come back in fp32.
import functools as _functools


def _fwd32(fwd):
    """torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32) without its recursive argument walk: under autocast the top-level floating-point CUDA
    tensors are cast to fp32 and the forward runs with autocast off (custom_bwd then runs the backward the same way); otherwise the forward runs as it is.  The nodes'
    arguments are flat (tensors, Segs tables, ints, tuples of fp32 constants), and torch's generic walk over them -- ~1 100 _cast calls per training step -- was 2 ms of
    the 20 ms of host work that bound the AMP step (tools/train_host_time.py)."""
    @_functools.wraps(fwd)
    def wrapper(ctx, *args):
        ctx._dtype = torch.get_autocast_dtype("cuda")
        ctx._fwd_used_autocast = False
        if torch.is_autocast_enabled("cuda"):
            args = tuple(a.float() if (isinstance(a, torch.Tensor) and a.is_cuda and a.is_floating_point() and a.dtype is not torch.float32) else a for a in args)
            with torch.autocast("cuda", enabled=False):
                return fwd(ctx, *args)
        return fwd(ctx, *args)
    return wrapper


_bwd = torch.amp.custom_bwd(device_type="cuda")
_fwd_keep = torch.amp.custom_fwd(device_type="cuda")      # nodes that handle f16 / fp32 inputs themselves (AMP_F16_STORE): no cast on entry

AMP_F16 = os.environ.get("FD_AMP_F16", "1") != "0"      # "0": the HIP nodes compute in fp32 under torch.autocast too (wider than the reference)
# Under autocast the reference's convolutions read and write fp16 TENSORS (train.py:175-181): with FD_AMP_F16_STORE (default on) the nodes that are built for it keep
# their activations -- and the gradients flowing back through them -- in HBM as f16 (fd_conv_params.io_f16: the FD_PREC_F16 kernels fetch / store f16 without a
# conversion pass; GradScaler keeps the gradients in f16's range exactly as it does for the reference).  Today: the ResNet bottlenecks (_BottleneckRows); the other
# nodes take fp32 and torch.amp's cast on entry converts at the boundary.  "0": fp32 maps everywhere (round 3's AMP step).
AMP_F16_STORE = os.environ.get("FD_AMP_F16_STORE", "1") != "0"


def amp_prec() -> int:
    """Conv arithmetic of the HIP nodes for the current autocast state: under torch.autocast(float16) -- how the reference trains
    (train.py:33 amp_enabled, :175-181 autocast + GradScaler) -- dense convolutions (forward, data gradient, weight gradient) run with
    f16 operands and fp32 accumulation (FD_PREC_F16: what autocast's fp16 convolution computes); GroupNorm / BatchNorm statistics,
    depthwise convs, SE, activations and the losses stay fp32, as autocast keeps normalisation and loss ops.  Otherwise exact fp32."""
    if AMP_F16 and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16:
        return _lib.PREC_F16
    return _lib.PREC_F32


AMP_K64 = os.environ.get("FD_AMP_K64", "1") != "0"      # "0": AMP convs stay on the single-plane f16 instantiations of the fp32 kernel (K-tiles of 32 channels)


def amp_pack(prec: int, K: int, Cout: int = 4):
    """PACKS.get's `f16` for a dense conv of reduction width K under the current arithmetic: False (fp32), True (the (hi, lo) f16 pair format of the fp32 kernel's f16
    instantiations) or 2 (FD_TILE_F16K64: K-tiles of 64 channels, fd_conv_f16.hip -- whenever the layer's widths allow)."""
    if not prec:
        return False
    return 2 if (AMP_K64 and ops.f16k64_ok(K, Cout)) else True


_STOCK = os.environ.get("FD_TRAIN_STOCK_CONV") == "1"   # diagnostic: route every layer to the stock ops (timing comparisons)
STATS = {"cl_copies": 0, "stock_fallbacks": 0}           # activation-sized layout copies made on entry / stock-op fallbacks taken (both should stay 0)
# FD_STRICT=1 (the test suite's default, tests/conftest.py): a layer of the TRAINING forward / backward that the HIP kernels do not cover
# raises FdError instead of silently running on stock PyTorch-ROCm ops (MIOpen convolutions, native batch / group norm).  Off by
# default so that exotic user configurations (a trainable 7x7 stem, odd widths) still train; the explicit diagnostic mode
# FD_TRAIN_STOCK_CONV=1 is exempt.
STRICT = os.environ.get("FD_STRICT", "0") == "1"


def stock_fallback(what: str) -> None:
    """Called on every path that is about to run a conv / norm of the training step on stock ops."""
    STATS["stock_fallbacks"] = STATS.get("stock_fallbacks", 0) + 1
    if STRICT and not _STOCK:
        raise FdError(f"FD_STRICT=1: {what} is not covered by the HIP kernels and would run on stock PyTorch-ROCm ops")


# ----------------------------------------------------------------------------------------------- layout helpers
def to_rows(t: torch.Tensor) -> torch.Tensor:
    """NCHW-shaped tensor -> [B*H*W, C] rows (a view when `t` is channels-last; autograd-tracked either way)."""
    if not t.is_contiguous(memory_format=torch.channels_last):
        STATS["cl_copies"] += 1
        t = t.contiguous(memory_format=torch.channels_last)
    B, Cc, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, Cc)


def from_rows(r: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """[B*H*W, C] rows -> NCHW-shaped channels-last view."""
    return r.view(B, H, W, r.shape[1]).permute(0, 3, 1, 2)


def pyramid_rows(maps: Sequence[torch.Tensor]):
    """Five NCHW maps -> one [sum B*H*W, C] rows buffer (level-major, the inference layout) + its segment table."""
    B = maps[0].shape[0]
    segs = Segs.make(B, [(t.shape[2], t.shape[3]) for t in maps])
    return torch.cat([to_rows(t) for t in maps], 0), segs


def pyramid_split(r: torch.Tensor, segs: Segs) -> List[torch.Tensor]:
    """Rows buffer -> list of NCHW-shaped channels-last views, one per level."""
    out = []
    for i, (h, w) in enumerate(segs.level_hw()):
        out.append(from_rows(r[segs.m_start[i]:segs.m_start[i + 1]], segs.batch, h, w))
    return out


def _r(t: torch.Tensor) -> Rows:
    return Rows(t)


def _rv(t: torch.Tensor) -> Optional[Rows]:
    """Rows of a 2-D tensor that may be a channel-slice VIEW of a contiguous rows buffer (x[:, a:b]); None if it is neither."""
    if t.is_contiguous():
        return Rows(t)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        cs = t.stride(0)
        co = t.storage_offset() % cs
        if co + t.shape[1] <= cs and t.storage_offset() - co + t.shape[0] * cs <= t.untyped_storage().nbytes() // 4:
            base = t.as_strided((t.shape[0], cs), (cs, 1), t.storage_offset() - co)
            return Rows(base, co, t.shape[1])
    return None


def _pad_of(m: nn.Conv2d) -> int:
    if isinstance(m.padding, str):
        return m.dilation[0] * (m.kernel_size[0] - 1) // 2
    return m.padding[0]


def _square(m: nn.Conv2d) -> bool:
    return (m.kernel_size[0] == m.kernel_size[1] and m.stride[0] == m.stride[1] and m.dilation[0] == m.dilation[1]
            and (isinstance(m.padding, str) or m.padding[0] == m.padding[1]) and m.padding_mode == "zeros")


def _f32(x: torch.Tensor) -> bool:
    """fp32 data, or any float data while autocast is on (the nodes cast their inputs up to fp32 then)."""
    return x.dtype == torch.float32 or (x.is_floating_point() and torch.is_autocast_enabled())


def _dense_ok(m: nn.Conv2d, x: torch.Tensor, pad_out: bool = False) -> bool:
    return (m.groups == 1 and m.in_channels % 32 == 0 and (pad_out or m.out_channels % 4 == 0) and _square(m)
            and _f32(x) and m.weight.dtype == torch.float32)


def _dw_ok(m: nn.Conv2d, x: torch.Tensor) -> bool:
    c4 = m.in_channels // 4
    return (m.groups == m.in_channels == m.out_channels and m.in_channels % 4 == 0 and m.kernel_size == (3, 3)
            and m.stride == (1, 1) and m.dilation == (1, 1) and _pad_of(m) == 1 and m.padding_mode == "zeros"
            and m.bias is None and ((c4 < 256 and 256 % c4 == 0) or c4 % 256 == 0) and _f32(x))


def _gn_ok(gn: nn.Module, x: torch.Tensor) -> bool:
    if not isinstance(gn, nn.GroupNorm) or not gn.affine or not _f32(x):
        return False
    Cc = gn.num_channels
    return Cc % 4 == 0 and Cc <= 1024 and 256 % (Cc // 4) == 0 and Cc % gn.num_groups == 0


BN_TYPES = (nn.BatchNorm2d, nn.SyncBatchNorm)     # SyncBatchNorm.convert_sync_batchnorm(model) (train.py:103) swaps the class, not the tensors


def bn_is_frozen(bn: Optional[nn.Module]) -> bool:
    return (isinstance(bn, BN_TYPES) and not bn.training and bn.track_running_stats and bn.affine
            and not bn.weight.requires_grad and not bn.bias.requires_grad)


def _bn_fold(bn: nn.BatchNorm2d):
    """(scale, shift) of a frozen BatchNorm2d, cached on the module until any of its tensors is written to."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr())
    hit = getattr(bn, "_fd_fold", None)
    if hit is None or hit[0] != key:
        hit = (key, ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
        bn._fd_fold = hit
    return hit[1]


def _act_id(act) -> int:
    if act is None or isinstance(act, nn.Identity):
        return ACT_NONE
    if isinstance(act, nn.ReLU):
        return ACT_RELU
    if isinstance(act, nn.SiLU):
        return ACT_SILU
    if isinstance(act, int):
        return act
    raise FdError(f"unsupported activation module {type(act).__name__}")


# ------------------------------------------------------------------------------------------- dense convolution
class _PackCache:
    """Packed conv weights of the model's PARAMETERS, kept across steps and refreshed by ONE launch per step.

    A train step needs every weight in the conv kernel's layout twice (forward, and flipped / transposed / BN-scaled for the
    data gradient) and the optimizer changes them all every step: ~150 small packing launches.  Entries are keyed by the
    parameter's storage (data_ptr, shape) and guarded by its version counter, so a stale entry is never used: a miss or a
    version mismatch packs on the spot (always correct), `refresh()` -- called at the top of every training forward --
    re-packs every known entry with fd_pack_conv_weights_batch_f32 (unconditionally: updates made through `.data` do not
    move the version counter) and records the versions it packed.  Code that calls the layer functions of this module
    directly AND updates weights through `.data` must call PACKS.refresh() itself before the next forward."""

    def __init__(self):
        self.entries: dict = {}     # key -> [weakref(param), scale, dgrad, out, version]
        self.params: dict = {}      # data_ptr -> weakref(param)
        self.table = None           # (signature, device tensor of fd_pack_job, max_elems)

    @staticmethod
    def _key(w, scale, dgrad, wino=0, f16=False):       # wino: 0 = direct kernel's packing, 1 = Winograd F(2x2, 3x3), 2 = Winograd F(4x4, 3x3); f16: False / True (hi, lo pair) / 2 (K-tile-64 f16)
        return (w.data_ptr(), tuple(w.shape), bool(dgrad), scale.data_ptr() if (scale is not None and dgrad) else 0,
                (16 if f16 == 2 else 2 if f16 else 0) + (1 if int(wino) == 1 else 0) + (8 if int(wino) == 2 else 0))

    @staticmethod
    def _pack_now(w, scale, dgrad, wino, f16=False):
        if not wino:
            if f16 == 2:       # FD_TILE_F16K64's operand (fd_conv_f16.hip)
                out = ops.pack_conv_weight_f16k64(w, scale, dgrad)
                out._fd_prec, out._fd_k64, out._fd_cout = _lib.PREC_F16, True, (w.shape[1] if dgrad else w.shape[0])
                return out
            out = ops.pack_conv_weight_hip(w, scale, dgrad, f16)
            out._fd_prec = _lib.PREC_F16 if f16 else _lib.PREC_F32
            return out
        out = ops.pack_conv_weight_wino4(w, scale, dgrad) if int(wino) == 2 else ops.pack_conv_weight_wino(w, scale, dgrad)
        out._fd_wino = (w.shape[1] if dgrad else w.shape[0])       # marks the Winograd packing for _conv_launch (value = output channels)
        out._fd_wino_tile = _lib.WINO4_TILE if int(wino) == 2 else _lib.WINO_TILE
        return out

    def get(self, w: torch.Tensor, scale: Optional[torch.Tensor] = None, dgrad: bool = False, wino: int = 0, f16: bool = False) -> torch.Tensor:
        """Packed weights for the conv kernel: the direct kernel's layout, or (3x3 stride-1 'same' layers) the Winograd packing -- wino=1: F(2x2, 3x3)
        of fd_conv_wino.hip, wino=2: F(4x4, 3x3) of fd_conv_wino4.hip; dgrad=True = the flipped / transposed / BN-scaled weights of the data gradient."""
        import weakref
        if isinstance(w, nn.Parameter) and w.is_contiguous():
            self.params[w.data_ptr()] = weakref.ref(w)
        ref = self.params.get(w.data_ptr())
        owner = ref() if ref is not None else None
        if owner is None or owner.data_ptr() != w.data_ptr() or owner.shape != w.shape or not w.is_contiguous():
            return self._pack_now(w, scale, dgrad, wino, f16)             # a temporary (merged / padded weights): pack now
        key = self._key(w, scale, dgrad, wino, f16)
        e = self.entries.get(key)
        if e is not None and e[0]() is not owner:       # the address was recycled by another parameter: drop the old entry
            e = None
        if e is None:
            out = self._pack_now(w, scale, dgrad, wino, f16)
            self.entries[key] = [ref, scale, dgrad, out, w._version]
            self.table = None
            return out
        if e[4] != w._version:                                            # changed since the last refresh: re-pack in place
            co, ci, kh, kw = w.shape
            sp = scale.data_ptr() if (scale is not None and dgrad) else None
            if int(wino) == 2:
                ops.check(_lib.lib().fd_wino4_pack_weights_f32(w.data_ptr(), sp, e[3].data_ptr(), co, ci, 1 if dgrad else 0, ops._stream()),
                          "fd_wino4_pack_weights_f32")
            elif wino:
                ops.check(_lib.lib().fd_wino_pack_weights_f32(w.data_ptr(), sp, e[3].data_ptr(), co, ci, 1 if dgrad else 0, ops._stream()),
                          "fd_wino_pack_weights_f32")
            else:
                ops.check(_lib.lib().fd_pack_conv_weight_f32(w.data_ptr(), sp, e[3].data_ptr(), co, ci, kh, kw, (1 if dgrad else 0) | (16 if f16 == 2 else 4 if f16 else 0),
                                                             ops._stream()), "fd_pack_conv_weight_f32")
            e[4] = w._version
        return e[3]

    def refresh(self) -> None:
        import ctypes as C
        live = {}
        newest: dict = {}        # (param address, shape) -> newest data-gradient entry: a re-folded frozen BN (load_state_dict) makes
        for key, e in self.entries.items():      # a new scale tensor and a new entry; the superseded one is dropped here
            if key[2]:
                newest[(key[0], key[1], key[4])] = key
        for key, e in self.entries.items():
            p = e[0]()
            if p is not None and p.data_ptr() == key[0] and tuple(p.shape) == key[1] and (not key[2] or newest[(key[0], key[1], key[4])] == key):
                live[key] = e
        self.params = {a: r for a, r in self.params.items() if r() is not None}
        if len(live) != len(self.entries):
            self.entries, self.table = live, None
        if not live:
            return
        # unconditional: an update made through `.data` does not move the version counter, a re-pack is one short launch
        if self.table is None:
            jobs = (_lib.PackJob * len(live))()
            mx = 0
            for j, (key, e) in zip(jobs, live.items()):
                co, ci, kh, kw = key[1]
                j.w, j.scale, j.out = key[0], (e[1].data_ptr() if (e[1] is not None and e[2]) else None), e[3].data_ptr()
                j.Cout, j.Cin, j.KH, j.KW, j.mode = co, ci, kh, kw, ((8 if key[4] & 8 else 2 if key[4] & 1 else 0) + (1 if e[2] else 0)
                                                                     + (4 if key[4] & 2 else 0) + (16 if key[4] & 16 else 0))
                mx = max(mx, co * ci * kh * kw)
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).clone()
            dev = next(iter(live.values()))[3].device
            self.table = (raw.to(dev), len(live), mx)
        tab, n, mx = self.table
        ops.check(_lib.lib().fd_pack_conv_weights_batch_f32(tab.data_ptr(), n, mx, ops._stream()), "fd_pack_conv_weights_batch_f32")
        for e in live.values():
            e[4] = e[0]()._version


PACKS = _PackCache()


def _wino(Cin: int, Cout: int, k: int, stride: int, pad: int, dil: int, segs: Segs, prec: int = 0) -> int:
    """3x3 stride-1 'same' layers run on a Winograd kernel -- forward (Cin -> Cout) and, with the channel roles swapped, data
    gradient -- unless FD_WINOGRAD=0 (the switch of the inference plans) or the map is so small that the direct kernel's split-K
    wins (ops.wino_preferred, the rule of the inference plans).  Returns PACKS.get's `wino`: 0 = direct, 1 = F(2x2, 3x3), 2 = F(4x4, 3x3)
    (ops.wino4_choice, the rule of the inference plans)."""
    from . import engine
    if prec != 0 or not engine.WINOGRAD:
        return 0
    if ops.wino4_ok(Cin, Cout, k, stride, pad, dil) and ops.wino4_choice(segs, Cin, Cout, dil, allow_split=False)[0]:      # (no workspace here)
        return 2
    return 1 if (ops.wino_ok(Cin, Cout, k, stride, pad, dil) and ops.wino_preferred(segs, Cin, Cout, dil)) else 0

_TILE_CACHE: dict = {}


def _conv_launch(x: torch.Tensor, segs: Segs, w_packed: torch.Tensor, y: torch.Tensor, *, k, stride, pad, dil, scale=None,
                 shift=None, res: Optional[torch.Tensor] = None, act=ACT_NONE, res_mask: bool = False) -> None:
    """y = act(conv(x, w) * scale + shift + res) on contiguous rows buffers; w_packed from ops.pack_conv_weight_hip.
    The block tile comes from the same table / heuristic / FD_AUTOTUNE timing as the inference plans (ops.autotune_conv),
    remembered per shape for the process."""
    wino_cout = getattr(w_packed, "_fd_wino", 0)
    if wino_cout:        # the Winograd packing (PACKS.get(..., wino=True)): fd_conv_wino.hip, no tile choice
        ops.conv_call(_r(x), segs, w_packed, _r(y), Cin=x.shape[1], Cout=wino_cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale,
                      shift=shift, res=_r(res) if res is not None else None, act=act, res_mask=res_mask, tile=getattr(w_packed, "_fd_wino_tile", _lib.WINO_TILE))()
        return
    if getattr(w_packed, "_fd_k64", False):      # AMP: f16 operands on K-tiles of 64 channels (the library picks the block tile)
        ops.conv_call(_r(x), segs, w_packed, _r(y), Cin=x.shape[1], Cout=w_packed._fd_cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift,
                      res=_r(res) if res is not None else None, act=act, res_mask=res_mask, precision=_lib.PREC_F16, tile=_lib.F16K64_TILE)()
        return
    Cin, Cout = x.shape[1], w_packed.shape[0]
    out_rows = y.shape[0]
    prec = getattr(w_packed, "_fd_prec", 0)
    KT = (Cin // 32) * k * k
    hw = "+".join(f"{h}x{w}" for h, w in segs.level_hw())
    key = f"B{segs.batch}|{hw}|{Cin}>{Cout}|k{k}s{stride}p{pad}d{dil}|res{int(res is not None)}|xcs{Cin}|ycs{Cout}"   # (mask / add: same cost)
    if prec:
        # f16 operands (AMP): own table entries ("f16|" keys, timed by `FD_AUTOTUNE=1 FD_AMP=1 python tools/tune_train.py`); a miss takes the
        # library's tile choice (the fp32 heuristic would name single-buffer tiles the f16 kernel is not built for), or is timed with FD_AUTOTUNE=1
        key = "f16|" + key
        code = _TILE_CACHE.get(key)
        if code is None:
            code = ops._tune_table().get(key)
            if code is None and ops._TUNE_MODE != "0":
                nb = _lib.lib().fd_conv_workspace_bytes(out_rows, Cout, ops.KSPLIT_MAX)
                ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device) if 0 < nb <= 256 * 1024 * 1024 else None
                probe = ops.conv_call(_r(x), segs, w_packed, _r(y), Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift,
                                      res=_r(res) if res is not None else None, act=act, workspace=ws, res_mask=res_mask, precision=prec)
                code = ops.autotune_conv(probe, key, out_rows, Cout, KT)
            code = _TILE_CACHE[key] = int(code or 0)
        ws = None
        if (code >> 8) > 1:
            ws = torch.empty(_lib.lib().fd_conv_workspace_bytes(out_rows, Cout, code >> 8) // 4, dtype=torch.float32, device=x.device)
        ops.conv_call(_r(x), segs, w_packed, _r(y), Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift,
                      res=_r(res) if res is not None else None, act=act, res_mask=res_mask, precision=prec, tile=code & 0xFF,
                      ksplit=max(1, code >> 8), workspace=ws)()
        return
    code = _TILE_CACHE.get(key)
    ws = None
    if code is None or (code >> 8) > 1:
        nb = _lib.lib().fd_conv_workspace_bytes(out_rows, Cout, ops.KSPLIT_MAX)
        if 0 < nb <= 256 * 1024 * 1024:
            ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
    call = ops.conv_call(_r(x), segs, w_packed, _r(y), Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale,
                         shift=shift, res=_r(res) if res is not None else None, act=act, workspace=ws, res_mask=res_mask)
    if code is None:
        code = _TILE_CACHE[key] = ops.autotune_conv(call, key, out_rows, Cout, KT)
    p = call.params
    p.tile, p.ksplit = code & 0xFF, max(1, code >> 8)
    if p.ksplit > 1 and ws is None:
        p.ksplit = 1
    call()


def _strided_dgrad(g: torch.Tensor, weight: torch.Tensor, scale: Optional[torch.Tensor], segs: Segs, k: int, stride: int, pad: int,
                   res: Optional[torch.Tensor] = None, res_mask: bool = False, prec: int = 0) -> Optional[torch.Tensor]:
    """dX rows of a strided single-level conv from dY rows `g` on the HIP conv kernel (None: geometry not covered)."""
    if _STOCK or segs.nseg != 1:
        return None
    B, (H, W) = segs.batch, segs.level_hw()[0]
    Cin = weight.shape[1]
    empty_class = any(T == 0 for _, T, _ in ops.strided_dgrad_classes(k, stride, pad))
    gx = (torch.zeros if empty_class else torch.empty)(B * H * W, Cin, dtype=g.dtype, device=g.device)      # (f16 maps under AMP_F16_STORE: the gradient's own type)
    ok = ops.conv_dgrad_strided(_r(g), weight, scale, _r(gx), B, H, W, k, stride, pad, res=_r(res) if res is not None else None,
                                res_mask=res_mask, precision=prec)
    if not ok:
        return None
    if res_mask and empty_class and res is not None:
        raise FdError("strided dgrad: a ReLU mask over a class-sparse gradient is not built")
    return gx


class _ConvRows(torch.autograd.Function):
    """y = act(conv(x, w) * scale + shift + residual) on rows; scale is a constant (frozen BN), shift may need a gradient."""

    @staticmethod
    @_fwd_keep
    def forward(ctx, x, weight, scale, shift, residual, segs, stride, pad, dil, act, prec=0, out_f16=False):
        # (no cast on entry: under AMP with AMP_F16_STORE the FD_PREC_F16 kernels read an f16 or an fp32 map as it is; out_f16 -- set by the caller for an edge whose
        # consumers are conv nodes too -- stores the output as f16.  Otherwise everything is fp32, as with torch.amp's cast_inputs.)
        if act not in (ACT_NONE, ACT_RELU):
            raise FdError("_ConvRows differentiates ACT_NONE / ACT_RELU epilogues only (use act_rows for SiLU: it keeps the pre-activation)")
        f16_io = bool(prec) and AMP_F16_STORE
        if not f16_io or x.dtype not in (torch.float16, torch.float32):
            x = x.float()
            residual = residual.float() if residual is not None else None
        x = x.contiguous()
        Cout, _, k, _ = weight.shape
        so = ops.conv_out_segs(segs, k, stride, pad, dil)
        y = torch.empty(so.rows, Cout, dtype=torch.float16 if (f16_io and out_f16) else torch.float32, device=x.device)
        _conv_launch(x, segs, PACKS.get(weight, wino=_wino(x.shape[1], Cout, k, stride, pad, dil, segs, prec), f16=amp_pack(prec, x.shape[1], Cout)), y, k=k, stride=stride,
                     pad=pad, dil=dil, scale=scale.float() if scale is not None else None, shift=shift.detach().float().contiguous() if shift is not None else None,
                     res=residual.contiguous() if residual is not None else None, act=act)
        ctx.res_dtype = residual.dtype if residual is not None else None
        ctx.save_for_backward(x, weight, scale, y if act == ACT_RELU else None)
        ctx.geom = (segs, so, stride, pad, dil, act, prec)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        segs, so, stride, pad, dil, act, prec = ctx.geom
        g = resolve_pending(gy).contiguous()
        if act == ACT_RELU:
            g = relu_mask(g if g.dtype == y.dtype else g.to(y.dtype), y)
        Cin = x.shape[1]
        Cout, _, k, _ = weight.shape
        gx = gw = gshift = gres = None
        if ctx.needs_input_grad[4]:
            gres = g if g.dtype == ctx.res_dtype else g.to(ctx.res_dtype)
        if ctx.needs_input_grad[0]:
            if stride == 1 and Cout % 32 == 0:
                gx = torch.empty_like(x)
                _conv_launch(g, so, PACKS.get(weight, scale, dgrad=True, wino=_wino(Cout, Cin, k, stride, dil * (k - 1) - pad, dil, so, prec), f16=amp_pack(prec, Cout, Cin)),
                             gx, k=k, stride=1, pad=dil * (k - 1) - pad, dil=dil)
            elif segs.nseg == 1 and stride > 1 and dil == 1 and (gx := _strided_dgrad(g, weight, scale, segs, k, stride, pad, prec=prec)) is not None:
                pass                                   # strided layer: one exact-FLOP launch per parity class (ops.conv_dgrad_strided)
            elif segs.nseg == 1:  # what is left (narrow Cout, dilated + strided): stock op for the data gradient
                stock_fallback(f"the data gradient of a {k}x{k} stride-{stride} conv with Cout={Cout}")
                weff = weight.detach() if scale is None else weight.detach() * scale.view(-1, 1, 1, 1)
                B, (H, W), (Ho, Wo) = segs.batch, segs.level_hw()[0], so.level_hw()[0]
                gx4 = torch.ops.aten.convolution_backward(from_rows(g, B, Ho, Wo), from_rows(x, B, H, W), weff, None,
                                                          [stride, stride], [pad, pad], [dil, dil], False, [0, 0], 1,
                                                          [True, False, False])[0]
                gx = to_rows(gx4)
            else:
                raise FdError("data gradient of a strided / narrow conv over a pyramid is not supported (pad Cout to 32)")
        if ctx.needs_input_grad[1]:
            gw = ops.conv_wgrad(_r(x), _r(g), segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, scale=scale,
                                oihw=True, precision=prec)
        if ctx.needs_input_grad[3]:
            gshift = g.sum(dim=0, dtype=torch.float32)
        return gx, gw, None, gshift, gres, None, None, None, None, None, None, None


def conv_rows(m: nn.Conv2d, x: torch.Tensor, segs: Segs, bn: Optional[nn.Module] = None, act: int = ACT_NONE,
              residual: Optional[torch.Tensor] = None, pad_out: bool = False, out_f16: bool = False) -> torch.Tensor:
    """HIP conv layer on a rows buffer (must be covered: check with `covered(m, bn, x)`).  With pad_out the output
    channels are zero-padded to a multiple of 32 inside (so the data gradient runs on the HIP kernel) and sliced back.
    out_f16: under AMP (and AMP_F16_STORE) the output map is stored as f16 -- for edges whose consumers are conv nodes (they read f16 maps directly)."""
    scale = shift = None
    if bn is not None:
        scale, shift = _bn_fold(bn)
    w, b = m.weight, m.bias
    Cout = w.shape[0]
    padn = (-Cout) % 32 if pad_out else 0
    if padn:
        w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, padn))
        b = F.pad(b, (0, padn)) if b is not None else None
        if scale is not None:
            scale, shift = F.pad(scale, (0, padn), value=1.0), F.pad(shift, (0, padn))
    if b is not None:
        shift = b if scale is None else b * scale + shift
    y = _ConvRows.apply(x, w, scale, shift, residual, segs, m.stride[0], _pad_of(m), m.dilation[0], act, amp_prec(), out_f16)
    return y[:, :Cout] if padn else y


class MergedConv:
    """Two convs with the same geometry and input (reg_pred 4 + cnt_logits 1) presented to conv_rows as one layer."""

    def __init__(self, a: nn.Conv2d, b: nn.Conv2d):
        self.weight = torch.cat((a.weight, b.weight), 0)
        self.bias = torch.cat((a.bias, b.bias), 0) if a.bias is not None else None
        self.stride, self.padding, self.dilation, self.kernel_size = a.stride, a.padding, a.dilation, a.kernel_size


class _BottleneckRows(torch.autograd.Function):
    """A whole ResNet bottleneck with frozen BatchNorm as one autograd node (torchvision Bottleneck.forward):
         y1 = relu(bn1(conv1 x));  y2 = relu(bn2(conv2 y1));  out = relu(bn3(conv3 y2) + (downsample(x) | x))
    Forward: the same 3-4 fused conv launches as separate nodes would issue.  Backward, written out by hand so that the
    elementwise work rides in the conv epilogues: one mask pass for the block output, then every data gradient is masked
    by the ReLU of the layer it flows into INSIDE the conv that produces it (res_mode 1) and the identity gradient is added
    in conv1's data-gradient epilogue (res_mode 0): per block 1 elementwise pass instead of 3 masks + 1 add."""

    @staticmethod
    @_fwd_keep
    def forward(ctx, x, w1, w2, w3, wd, c1, c2, c3, cd, segs, stride, prec=0):
        # (no cast on entry: under AMP the node takes an f16 map as it is -- the previous bottleneck's output -- or an fp32 one, and keeps its own maps in f16)
        h = bool(prec)
        st = torch.float16 if (h and AMP_F16_STORE) else torch.float32          # storage type of this node's activations and gradients
        x = x.contiguous() if (x.dtype == st or (h and AMP_F16_STORE and x.dtype == torch.float32)) else x.to(st).contiguous()
        dev = x.device
        so = ops.conv_out_segs(segs, 3, stride, 1, 1)
        P, C4 = w1.shape[0], w3.shape[0]
        y1 = torch.empty(segs.rows, P, dtype=st, device=dev)
        y2 = torch.empty(so.rows, P, dtype=st, device=dev)
        out = torch.empty(so.rows, C4, dtype=st, device=dev)
        _conv_launch(x, segs, PACKS.get(w1, f16=amp_pack(prec, w1.shape[1], w1.shape[0])), y1, k=1, stride=1, pad=0, dil=1, scale=c1[0], shift=c1[1], act=ACT_RELU)
        _conv_launch(y1, segs, PACKS.get(w2, wino=_wino(P, P, 3, stride, 1, 1, segs, prec), f16=amp_pack(prec, P, P)), y2, k=3, stride=stride, pad=1, dil=1, scale=c2[0],
                     shift=c2[1], act=ACT_RELU)
        if wd is not None:
            idt = torch.empty(so.rows, C4, dtype=st, device=dev)
            _conv_launch(x, segs, PACKS.get(wd, f16=amp_pack(prec, wd.shape[1], wd.shape[0])), idt, k=1, stride=stride, pad=0, dil=1, scale=cd[0], shift=cd[1])
        else:
            idt = x
        _conv_launch(y2, so, PACKS.get(w3, f16=amp_pack(prec, w3.shape[1], w3.shape[0])), out, k=1, stride=1, pad=0, dil=1, scale=c3[0], shift=c3[1], res=idt,
                     act=ACT_RELU)
        ctx.save_for_backward(x, y1, y2, out, w1, w2, w3, wd, c1[0], c2[0], c3[0], cd[0] if wd is not None else None)
        ctx.geom = (segs, so, stride, prec)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, gout):
        x, y1, y2, out, w1, w2, w3, wd, s1, s2, s3, sd = ctx.saved_tensors
        segs, so, stride, prec = ctx.geom
        h = bool(prec)
        need_x = ctx.needs_input_grad[0]
        P, Cin, C4 = w1.shape[0], w1.shape[1], w3.shape[0]
        g = relu_mask(gout.contiguous() if gout.dtype == out.dtype else gout.to(out.dtype), out)      # the one elementwise pass of the block
        gw1 = gw2 = gw3 = gwd = gx = None
        wg = lambda xx, gg, sg, Ci, Co, k, st, pad, sc: ops.conv_wgrad(_r(xx), _r(gg), sg, Cin=Ci, Cout=Co, k=k, stride=st,  # noqa: E731
                                                                     pad=pad, dil=1, scale=sc, oihw=True, precision=prec)
        if ctx.needs_input_grad[3]:
            gw3 = wg(y2, g, so, P, C4, 1, 1, 0, s3)
        g2 = torch.empty_like(y2)                                                    # d/d(conv2 output), ReLU-masked in the epilogue
        _conv_launch(g, so, PACKS.get(w3, s3, dgrad=True, f16=amp_pack(prec, w3.shape[0], w3.shape[1])), g2, k=1, stride=1, pad=0, dil=1, res=y2, res_mask=True)
        if ctx.needs_input_grad[2]:
            gw2 = wg(y1, g2, segs, P, P, 3, stride, 1, s2)
        if stride == 1:
            g1 = torch.empty_like(y1)
            _conv_launch(g2, so, PACKS.get(w2, s2, dgrad=True, wino=_wino(w2.shape[0], w2.shape[1], 3, 1, 1, 1, so, prec), f16=amp_pack(prec, w2.shape[0], w2.shape[1])), g1, k=3, stride=1, pad=1,
                         dil=1, res=y1, res_mask=True)
        elif (g1 := _strided_dgrad(g2, w2, s2, segs, 3, stride, 1, res=y1, res_mask=True, prec=prec)) is not None:
            pass    # strided 3x3: four parity-class launches on the conv kernel, ReLU mask of y1 applied in their epilogues
        else:   # strided 3x3: stock data gradient, masked separately
            stock_fallback("the data gradient of a bottleneck's strided 3x3 conv")
            B, (H, W), (Ho, Wo) = segs.batch, segs.level_hw()[0], so.level_hw()[0]
            g1 = torch.ops.aten.convolution_backward(from_rows(g2, B, Ho, Wo), from_rows(y1, B, H, W), w2.detach() * s2.view(-1, 1, 1, 1),
                                                     None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            g1 = relu_mask(to_rows(g1).contiguous(), y1)
        if ctx.needs_input_grad[1]:
            gw1 = wg(x, g1, segs, Cin, P, 1, 1, 0, s1)
        if wd is not None and ctx.needs_input_grad[4]:
            gwd = wg(x, g, segs, Cin, C4, 1, stride, 0, sd)
        if need_x:
            if wd is None:
                gid = g                                                             # identity path
            elif stride == 1:
                gid = torch.empty_like(x)
                _conv_launch(g, so, PACKS.get(wd, sd, dgrad=True, f16=amp_pack(prec, wd.shape[0], wd.shape[1])), gid, k=1, stride=1, pad=0, dil=1)
            elif (gid := _strided_dgrad(g, wd, sd, segs, 1, stride, 0, prec=prec)) is not None:
                pass    # 1x1 stride-2 downsample: the (0, 0) parity class is a plain GEMM scattered into a zeroed dX
            else:
                stock_fallback("the data gradient of a bottleneck's strided downsample conv")
                B, (H, W), (Ho, Wo) = segs.batch, segs.level_hw()[0], so.level_hw()[0]
                gid = to_rows(torch.ops.aten.convolution_backward(from_rows(g, B, Ho, Wo), from_rows(x, B, H, W),
                                                                   wd.detach() * sd.view(-1, 1, 1, 1), None, [stride, stride], [0, 0],
                                                                   [1, 1], False, [0, 0], 1, [True, False, False])[0])
            gx = torch.empty_like(x)                                                # conv1's data gradient + the identity gradient
            _conv_launch(g1, segs, PACKS.get(w1, s1, dgrad=True, f16=amp_pack(prec, w1.shape[0], w1.shape[1])), gx, k=1, stride=1, pad=0, dil=1, res=gid)
        return gx, gw1, gw2, gw3, gwd, None, None, None, None, None, None, None


def bottleneck(blk: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """torchvision-style Bottleneck (conv1/bn1, conv2/bn2, conv3/bn3, optional downsample) on an NCHW-shaped tensor: one
    fused autograd node when every BatchNorm is frozen and the shapes are covered, else layer by layer (conv_bn_act)."""
    _need_cuda(x)
    ds = blk.downsample
    bns = [blk.bn1, blk.bn2, blk.bn3] + ([ds[1]] if ds is not None else [])
    convs = [blk.conv1, blk.conv2, blk.conv3] + ([ds[0]] if ds is not None else [])
    ok = (not _STOCK and all(bn_is_frozen(b) for b in bns) and all(c.bias is None and _dense_ok(c, x) and c.out_channels % 32 == 0
                                                                   for c in convs)
          and blk.conv1.kernel_size == (1, 1) and blk.conv3.kernel_size == (1, 1) and blk.conv2.kernel_size == (3, 3)
          and blk.conv2.padding == (1, 1) and blk.conv2.dilation == (1, 1) and blk.conv1.stride == (1, 1) and blk.conv3.stride == (1, 1)
          and (ds is None or (ds[0].kernel_size == (1, 1) and ds[0].stride == blk.conv2.stride)))
    if not ok:
        idt = x if ds is None else conv_bn_act(ds[0], ds[1], x, ACT_NONE)
        y = conv_bn_act(blk.conv1, blk.bn1, x, ACT_RELU)
        y = conv_bn_act(blk.conv2, blk.bn2, y, ACT_RELU)
        return conv_bn_act(blk.conv3, blk.bn3, y, ACT_RELU, residual=idt)
    B, _, H, W = x.shape
    s = blk.conv2.stride[0]
    out = _BottleneckRows.apply(to_rows(x), blk.conv1.weight, blk.conv2.weight, blk.conv3.weight, ds[0].weight if ds is not None else None,
                                _bn_fold(blk.bn1), _bn_fold(blk.bn2), _bn_fold(blk.bn3), _bn_fold(ds[1]) if ds is not None else (None, None),
                                Segs.make(B, [(H, W)]), s, amp_prec())
    return from_rows(out, B, (H - 1) // s + 1, (W - 1) // s + 1)


# --------------------------------------------------------------------------------------------- depthwise 3x3
class _DwRows(torch.autograd.Function):
    """Depthwise 3x3 (stride 1, pad 1, no bias): y = act(dw(x, w) * scale + shift), scale / shift constants."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, weight, scale, shift, segs, act):
        if act not in (ACT_NONE, ACT_RELU):
            raise FdError("_DwRows differentiates ACT_NONE / ACT_RELU epilogues only (use act_rows for SiLU)")
        x = x.contiguous()
        y = torch.empty_like(x)
        ops.dwconv3x3(_r(x), ops.pack_dw_weight(weight), _r(y), segs, scale, shift, act)
        ctx.save_for_backward(x, weight, scale, y if act == ACT_RELU else None)
        ctx.geom = (segs, act)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        segs, act = ctx.geom
        g = resolve_pending(gy).contiguous()
        if act == ACT_RELU:
            g = relu_mask(g, y)
        Cc = x.shape[1]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            weff = weight.detach() if scale is None else weight.detach() * scale.view(-1, 1, 1, 1)
            gx = torch.empty_like(x)
            ops.dwconv3x3(_r(g), ops.pack_dw_weight(weff.flip(2, 3)), _r(gx), segs)
        if ctx.needs_input_grad[1]:
            gw = ops.dwconv3x3_wgrad(_r(x), _r(g), segs, scale, torch_layout=True)
        return gx, gw, None, None, None, None


def dw_rows(m: nn.Conv2d, x: torch.Tensor, segs: Segs, bn: Optional[nn.Module] = None, act: int = ACT_NONE) -> torch.Tensor:
    scale = shift = None
    if bn is not None:
        scale, shift = _bn_fold(bn)
    return _DwRows.apply(x, m.weight, scale, shift, segs, act)


# --------------------------------------------------------------------------------------- GroupNorm + activation
class _GroupNormRows(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, gamma, beta, segs, G, eps, act):
        x = x.contiguous()
        y = torch.empty_like(x)
        ws = ops.groupnorm_workspace(segs, G, x.device)
        gm, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        ops.groupnorm_act(_r(x), gm, bt, _r(y), segs, G, act, ws, eps)
        ctx.save_for_backward(x, gm, bt, ws)
        ctx.geom = (segs, G, eps, act)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, gm, bt, ws = ctx.saved_tensors
        segs, G, eps, act = ctx.geom
        g = gy.contiguous()
        gx = torch.empty_like(x)
        dgamma, dbeta = ops.groupnorm_act_bwd(_r(x), _r(g), gm, bt, _r(gx), segs, G, act, ws, eps)
        return gx, dgamma, dbeta, None, None, None, None


def groupnorm_rows(gn: nn.GroupNorm, x: torch.Tensor, segs: Segs, act=ACT_NONE) -> torch.Tensor:
    """act(GroupNorm(x)) per (level, image) on a rows buffer: two HIP launches forward, three backward."""
    return _GroupNormRows.apply(x, gn.weight, gn.bias, segs, gn.num_groups, gn.eps, _act_id(act))


# ------------------------------------------------------------------- elementwise / pooling / SE / BatchNorm(train) nodes
def relu_mask(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """g * [y > 0] on contiguous rows (the ReLU backward from the saved OUTPUT), one HIP launch."""
    out = torch.empty_like(g)
    ops.act_bwd(_r(y), _r(g), _r(out), ACT_RELU)
    return out


class _ActRows(torch.autograd.Function):
    """y = act(x) on rows with the INPUT saved: SiLU after a BatchNorm (HISFcos.py:100,112) needs the pre-activation."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, act):
        xr = _rv(x)                           # (a channel slice of a wider rows buffer is read in place)
        if xr is None:
            x = x.contiguous()
            xr = _r(x)
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        ops.act(xr, _r(y), act)
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        gx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        ops.act_bwd(_rv(x), _r(gy.contiguous()), _r(gx), ctx.act)
        return gx, None


def act_rows(x: torch.Tensor, act: int) -> torch.Tensor:
    return x if act == ACT_NONE else _ActRows.apply(x, act)


class _PoolAddRows(torch.autograd.Function):
    """y = max_pool2d(x, k, s, pad) (+ add) on single-level rows (FPN down_sample + torch.add, HISFcos.py:131-136,168-177)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, add, geom):
        B, H, W, k, s, pad = geom
        x = x.contiguous()
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        y = torch.empty(B * Ho * Wo, x.shape[1], dtype=torch.float32, device=x.device)
        ops.maxpool(_r(x), _r(y), B, H, W, k, s, pad, add=_r(add.contiguous()) if add is not None else None)
        ctx.save_for_backward(x)
        ctx.geom = geom
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        B, H, W, k, s, pad = ctx.geom
        g = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            ops.maxpool_bwd(_r(x), _r(g), _r(gx), B, H, W, k, s, pad)
        return gx, (g if ctx.needs_input_grad[1] else None), None


class _UpAddRows(torch.autograd.Function):
    """y = nearest_upsample_x2(x) + lat on single-level rows (HISFcos.py:155-165)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, lat, geom):
        B, H, W = geom                       # geometry of the LOW-resolution input x
        y = torch.empty_like(lat)
        ops.upsample2x_add(_r(x.contiguous()), _r(lat.contiguous()), _r(y), B, H, W)
        ctx.geom = geom
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        B, H, W = ctx.geom
        g = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B * H * W, g.shape[1], dtype=torch.float32, device=g.device)
            ops.upsample2x_bwd(_r(g), _r(gx), B, H, W)
        return gx, (g if ctx.needs_input_grad[1] else None), None


class _SERows(torch.autograd.Function):
    """SEBlock (modules.py:107-121) on rows: y = x * sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2), forward and backward on HIP."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, w1, b1, w2, b2, B, HW):
        x = x.contiguous()
        Cc, Cr = x.shape[1], w1.shape[0]
        y = torch.empty_like(x)
        ws = ops.se_workspace(B, HW, Cc, x.device)
        w1f, w2f = w1.detach().reshape(Cr, Cc).contiguous(), w2.detach().reshape(Cc, Cr).contiguous()
        b1f, b2f = b1.detach().contiguous(), b2.detach().contiguous()
        ops.se_scale(_r(x), w1f, b1f, w2f, b2f, _r(y), B, HW, Cr, ws)
        ctx.save_for_backward(x, w1f, b1f, w2f, b2f, ws)
        ctx.geom = (B, HW, Cr, tuple(w1.shape), tuple(w2.shape))
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, w1f, b1f, w2f, b2f, ws = ctx.saved_tensors
        B, HW, Cr, s1, s2 = ctx.geom
        gx = torch.empty_like(x)
        dw1, db1, dw2, db2 = ops.se_scale_bwd(_r(x), _r(gy.contiguous()), w1f, b1f, w2f, b2f, _r(gx), B, HW, Cr, ws)
        return gx, dw1.view(s1), db1, dw2.view(s2), db2, None, None


def se_rows(se: nn.Module, x: torch.Tensor, B: int, HW: int) -> torch.Tensor:
    ex = se.excitation
    return _SERows.apply(x, ex[0].weight, ex[0].bias, ex[2].weight, ex[2].bias, B, HW)


def _se_ok(se: nn.Module, x: torch.Tensor) -> bool:
    ex = getattr(se, "excitation", None)
    return (not _STOCK and ex is not None and len(ex) == 4 and isinstance(ex[1], nn.SiLU) and isinstance(ex[3], nn.Sigmoid)
            and ex[0].bias is not None and ex[2].bias is not None and ex[0].in_channels % 4 == 0 and ex[0].in_channels <= 4096
            and ex[0].out_channels <= 1024 and _f32(x))


class _BatchNormTrainRows(torch.autograd.Function):
    """nn.BatchNorm2d in TRAINING mode + ReLU / SiLU on rows.  Batch statistics per channel over all rows are GroupNorm
    statistics of ONE image with G = C groups, so forward / backward are the GroupNorm kernels (fp64 partial sums in fixed
    order); the running statistics are updated like nn.BatchNorm2d does (momentum, unbiased variance)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, gamma, beta, rmean, rvar, momentum, eps, act):
        x = x.contiguous()
        rows, Cc = x.shape
        segs = Segs.make(1, [(rows, 1)])
        y = torch.empty_like(x)
        ws = ops.groupnorm_workspace(segs, Cc, x.device)
        gm, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        ops.groupnorm_act(_r(x), gm, bt, _r(y), segs, Cc, act, ws, eps)
        if rmean is not None and momentum is not None:
            ops.batchnorm_update_running(ws, rows, Cc, momentum, eps, rmean, rvar)
        ctx.save_for_backward(x, gm, bt, ws)
        ctx.geom = (segs, eps, act)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, gm, bt, ws = ctx.saved_tensors
        segs, eps, act = ctx.geom
        gx = torch.empty_like(x)
        dgamma, dbeta = ops.groupnorm_act_bwd(_r(x), _r(gy.contiguous()), gm, bt, _r(gx), segs, x.shape[1], act, ws, eps)
        return gx, dgamma, dbeta, None, None, None, None, None


# ---- nn.SyncBatchNorm: statistics collectives OFF the critical path (round 4) ----
# A layer's statistics cannot be batched with another layer's (data dependence), but the all-reduce need not be WAITED for where it is issued:
#   forward   syncbn_begin() runs phase 1 (this rank's fp64 sums) and issues the all-reduce with async_op=True; the caller enqueues whatever does not
#             depend on the normalised tensor (HisBlock: conv2(x) beside bn1, the SE branch beside bn2; the FPN laterals beside the previous block),
#             then syncbn_finish() waits and runs phase 2.
#   backward  the node issues the all-reduce of (sum dz, sum dz * xhat) asynchronously too and -- when its caller has promised that the ONLY consumer of
#             its input gradient is one of this module's conv nodes (`defer`) -- returns that gradient UNFINISHED with a pending entry; the consumer
#             resolves it (wait + phase 2, same stream) on entry.  Autograd runs the nodes created between begin() and finish() in between (bn1's
#             all-reduce flies under conv2's data / weight gradient kernels, bn2's under the SE backward).
# With RCCL the collective runs on its own stream and wait() is a stream dependency; with gloo (tests) wait() blocks the host after the independent
# launches were enqueued.  SYNC_TRACE records, per collective, how many HIP launches were enqueued between issue and wait (tests assert on it).
import collections as _collections
SYNC_TRACE = _collections.deque(maxlen=4096)      # (bounded: a long multi-rank run appends ~200 entries per step; only tests read it)
# Unfinished input gradients of deferred SyncBatchNorm backward nodes: data_ptr -> (autograd graph-task id of the backward pass that deferred it, closure that
# finishes it).  The task id ties an entry to ITS pass: what a pass that raised before its flush left behind is never applied to a later pass's gradient that
# happens to reuse the address, and is dropped when the next pass defers its first gradient.
_PENDING: dict = {}


def _task_id() -> int:
    return torch._C._current_graph_task_id()


def resolve_pending(g: torch.Tensor) -> torch.Tensor:
    """Called by the conv nodes' backward on their incoming gradient: finishes it if a SyncBatchNorm node of THIS backward pass deferred its second phase."""
    if _PENDING:
        e = _PENDING.get(g.data_ptr())
        if e is not None and e[0] == _task_id():
            del _PENDING[g.data_ptr()]
            e[1]()
    return g


def flush_pending() -> None:
    """Finish every gradient this pass deferred and nobody resolved (safety net, queued at the end of EVERY backward pass that deferred one); entries of other
    (failed) passes are dropped unfinished."""
    tid = _task_id()
    while _PENDING:
        _, (t, fin) = _PENDING.popitem()
        if t == tid or tid < 0:          # (tid < 0: the engine runs the callback outside the task -- finishing a leftover writes its own, dead tensor: harmless)
            fin()


class _SyncHandle:
    __slots__ = ("bn", "x", "act", "group", "buf", "ws", "work", "launches_at_issue", "defer")


def syncbn_begin(bn: nn.Module, x: torch.Tensor, act: int = ACT_NONE, defer_backward: bool = False) -> "_SyncHandle":
    """Phase 1 of nn.SyncBatchNorm's forward on rows + the asynchronous all-reduce of the [2C + 1] fp64 buffer (sums and row count)."""
    import torch.distributed as dist
    h = _SyncHandle()
    h.bn, h.act, h.defer = bn, act, defer_backward
    h.group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    with torch.no_grad():
        xd = x.detach().float().contiguous()
        rows, Cc = xd.shape
        segs = Segs.make(1, [(rows, 1)])
        h.ws = ops.groupnorm_workspace(segs, Cc, xd.device)
        h.buf = torch.empty(2 * Cc + 1, dtype=torch.float64, device=xd.device)
        fn = _lib.lib().fd_batchnorm_sync_fwd_nhwc
        ops.check(fn(xd.data_ptr(), Cc, 0, None, None, None, 0, 0, rows, Cc, bn.eps, act, 1, h.buf.data_ptr(), 0.0, h.ws.data_ptr(), ops._stream()),
                  "fd_batchnorm_sync_fwd_nhwc (stats)")
        h.buf[2 * Cc] = float(rows)                               # this rank's row count rides behind the 2C sums (a device-side fill, no sync)
        h.work = dist.all_reduce(h.buf, group=h.group, async_op=True)       # THE forward collective of this layer (C3), not waited for here
    h.x = x
    h.launches_at_issue = ops.LAUNCHES[0]
    return h


def syncbn_finish(h: "_SyncHandle") -> torch.Tensor:
    bn = h.bn
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _SyncBatchNormApply.apply(h.x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, h.act, h)


class _SyncBatchNormApply(torch.autograd.Function):
    """Second half of nn.SyncBatchNorm in TRAINING mode + ReLU / SiLU on rows (train.py:101-103 converts the model after the DDP wrap): waits for the
    all-reduce syncbn_begin issued, normalises with the GLOBAL statistics, updates the running statistics with the global row count -- read ON THE
    DEVICE from the all-reduced buffer: ranks whose batches are padded to different H x W hold different row counts (dataset/voc.py:141-171).  Like
    torch's SyncBatchNorm the affine gradients come from the local sums (DDP averages them)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, gamma, beta, rmean, rvar, momentum, eps, act, h):
        import ctypes as C
        x = x.contiguous()
        rows, Cc = x.shape
        y = torch.empty_like(x)
        gm, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        SYNC_TRACE.append(("fwd", ops.LAUNCHES[0] - h.launches_at_issue))
        h.work.wait()
        fn, st = _lib.lib().fd_batchnorm_sync_fwd_nhwc, ops._stream()
        ops.check(fn(x.data_ptr(), Cc, 0, gm.data_ptr(), bt.data_ptr(), y.data_ptr(), Cc, 0, rows, Cc, eps, act, 2, h.buf.data_ptr(),
                     C.c_double(-1.0), h.ws.data_ptr(), st), "fd_batchnorm_sync_fwd_nhwc (apply)")
        if rmean is not None and momentum is not None:
            ops.batchnorm_update_running_dev(h.ws, h.buf[2 * Cc:], Cc, momentum, eps, rmean, rvar)
        ctx.save_for_backward(x, gm, bt, h.ws, h.buf)
        ctx.geom = (eps, act, h.group, h.defer)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        import ctypes as C
        import torch.distributed as dist
        x, gm, bt, ws, buf = ctx.saved_tensors
        eps, act, group, defer = ctx.geom
        rows, Cc = x.shape
        g = gy.contiguous()
        gx = torch.empty_like(x)
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
        segs = Segs.make(1, [(rows, 1)])
        nb = _lib.lib().fd_groupnorm_bwd_workspace_bytes(C.byref(segs), Cc)
        bws = torch.empty(nb // 8, dtype=torch.float64, device=x.device)
        sums = torch.empty(2 * Cc + 1, dtype=torch.float64, device=x.device)       # [sum dz | sum dz * xhat | global row count]
        fn = _lib.lib().fd_batchnorm_sync_bwd_nhwc
        ops.check(fn(x.data_ptr(), Cc, 0, g.data_ptr(), Cc, 0, gm.data_ptr(), bt.data_ptr(), None, 0, 0, dgamma.data_ptr(), dbeta.data_ptr(),
                     rows, Cc, eps, act, 1, sums.data_ptr(), 0.0, ws.data_ptr(), bws.data_ptr(), ops._stream()), "fd_batchnorm_sync_bwd_nhwc (sums)")
        sums[2 * Cc:].copy_(buf[2 * Cc:])                       # the forward's all-reduced row count, device to device (its own slot: not all-reduced again)
        work = dist.all_reduce(sums[:2 * Cc], group=group, async_op=True)     # THE backward collective of this layer
        at_issue = ops.LAUNCHES[0]

        def finish(x=x, g=g, gm=gm, bt=bt, gx=gx, sums=sums, ws=ws, work=work):
            SYNC_TRACE.append(("bwd", ops.LAUNCHES[0] - at_issue))
            work.wait()
            ops.check(fn(x.data_ptr(), Cc, 0, g.data_ptr(), Cc, 0, gm.data_ptr(), bt.data_ptr(), gx.data_ptr(), Cc, 0, None, None,
                         rows, Cc, eps, act, 2, sums.data_ptr(), C.c_double(-1.0), ws.data_ptr(), None, ops._stream()), "fd_batchnorm_sync_bwd_nhwc (apply)")

        if defer:
            # the ONLY consumer of gx is one of this module's conv nodes (conv_norm_begin built the chain): it finishes gx on entry (resolve_pending);
            # the nodes autograd runs before it are enqueued under the collective.  flush_pending at the end of the pass is the safety net.
            tid = _task_id()
            for k in [k for k, e in _PENDING.items() if e[0] != tid]:      # leftovers of a pass that raised before its flush
                del _PENDING[k]
            torch.autograd.Variable._execution_engine.queue_callback(flush_pending)      # always: a stale entry must not suppress this pass's flush
            _PENDING[gx.data_ptr()] = (tid, finish)
        else:
            finish()
        return gx, dgamma, dbeta, None, None, None, None, None, None


def _bn_train_ok(bn: nn.Module, x: torch.Tensor) -> bool:
    Cc = getattr(bn, "num_features", 0)
    return (not _STOCK and isinstance(bn, BN_TYPES) and bn.training and bn.affine and bn.track_running_stats
            and bn.momentum is not None and Cc % 4 == 0 and Cc <= 1024 and 256 % (Cc // 4) == 0 and _f32(x))


def batchnorm_train_rows(bn: nn.Module, x: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:
    """act(bn(x)) with batch statistics on rows; bn.running_mean / running_var / num_batches_tracked are updated in place like
    nn.BatchNorm2d does (check with _bn_train_ok).  An nn.SyncBatchNorm in an initialised process group of more than one rank
    takes its statistics over all ranks (_SyncBatchNormTrainRows: one all-reduce forward, one backward)."""
    if _is_sync(bn):
        return syncbn_finish(syncbn_begin(bn, x, act))          # one-shot form: nothing enqueued between issue and wait
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BatchNormTrainRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act)


def _is_sync(bn: nn.Module) -> bool:
    """An nn.SyncBatchNorm in an initialised process group of more than one rank (its statistics span all ranks)."""
    import torch.distributed as dist
    if isinstance(bn, nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        group = bn.process_group if bn.process_group is not None else dist.group.WORLD
        return dist.get_world_size(group) > 1
    return False


class NormHandle:
    """What conv_norm_begin returns: either the finished tensor (`y`) or a SyncBatchNorm whose all-reduce is in flight (`sync`)."""
    __slots__ = ("y", "sync")

    def __init__(self, y=None, sync=None):
        self.y, self.sync = y, sync


def conv_norm_begin(m: nn.Conv2d, bn: Optional[nn.Module], x: torch.Tensor, segs: Segs, act: int = ACT_NONE) -> Optional["NormHandle"]:
    """conv_norm_act_rows in two halves: everything up to and including the ISSUE of a SyncBatchNorm's statistics all-reduce.  The caller enqueues work
    that does not need the normalised tensor, then calls conv_norm_finish.  (Any other layer kind is simply computed here.)"""
    if bn is not None and not bn_is_frozen(bn) and _is_sync(bn):
        dense, dw = _dense_ok(m, x), _dw_ok(m, x)
        if _STOCK or not (dense or dw) or not _bn_train_ok(bn, x) or (dw and m.bias is not None):
            return None
        pre = conv_rows(m, x, segs, None, ACT_NONE) if dense else dw_rows(m, x, segs, None, ACT_NONE)
        # the conv node just created is the only consumer of the BatchNorm's input gradient: its backward resolves the deferred second phase
        return NormHandle(sync=syncbn_begin(bn, pre, act, defer_backward=True))
    y = conv_norm_act_rows(m, bn, x, segs, act)
    return None if y is None else NormHandle(y=y)


def conv_norm_finish(h: Optional["NormHandle"]) -> Optional[torch.Tensor]:
    if h is None:
        return None
    return h.y if h.sync is None else syncbn_finish(h.sync)


def conv_norm_act_rows(m: nn.Conv2d, bn: Optional[nn.Module], x: torch.Tensor, segs: Segs, act: int = ACT_NONE) -> Optional[torch.Tensor]:
    """act(bn(m(x))) on single-level rows with every piece on the HIP kernels, whatever mode `bn` is in: a frozen BatchNorm folds
    into the conv epilogue (ReLU fused, SiLU one extra launch that keeps the pre-activation), a BatchNorm in training mode runs
    on batch statistics (batchnorm_train_rows).  Returns None when a piece is not covered (the caller falls back)."""
    dense, dw = _dense_ok(m, x), _dw_ok(m, x)
    if _STOCK or not (dense or dw):
        return None
    conv = conv_rows if dense else (lambda mm, xx, sg, b=None, a=ACT_NONE: dw_rows(mm, xx, sg, b, a))
    if bn is None or bn_is_frozen(bn):
        if act in (ACT_NONE, ACT_RELU):
            return conv(m, x, segs, bn, act)
        return act_rows(conv(m, x, segs, bn, ACT_NONE), act)
    if not _bn_train_ok(bn, x) or (dw and m.bias is not None):
        return None
    return batchnorm_train_rows(bn, conv(m, x, segs, None, ACT_NONE), act)


# ------------------------------------------------------------------------------------ NCHW-shaped conveniences
def _need_cuda(x: torch.Tensor) -> None:
    if not x.is_cuda:
        raise FdError("pytorch_object_detection_amd runs on the GPU only; there is no CPU fallback (got a CPU tensor)")


def covered(m: nn.Conv2d, bn: Optional[nn.Module], x: torch.Tensor, pad_out: bool = False) -> bool:
    return not _STOCK and (bn is None or bn_is_frozen(bn)) and (_dense_ok(m, x, pad_out) or _dw_ok(m, x))


def conv_bn_act(m: nn.Conv2d, bn: Optional[nn.Module], x: torch.Tensor, act: int = ACT_NONE,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(bn(m(x)) + residual) on NCHW-shaped tensors, act in {ACT_NONE, ACT_RELU}; one HIP launch when covered."""
    _need_cuda(x)
    if covered(m, bn, x) and not (m.groups != 1 and residual is not None):
        B, _, H, W = x.shape
        segs = Segs.make(B, [(H, W)])
        if m.groups == 1:
            y = conv_rows(m, to_rows(x), segs, bn, act, to_rows(residual) if residual is not None else None)
            k, s, p, d = m.kernel_size[0], m.stride[0], _pad_of(m), m.dilation[0]
            return from_rows(y, B, (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1)
        return from_rows(dw_rows(m, to_rows(x), segs, bn, act), B, H, W)
    if not _STOCK and bn is not None and not bn_is_frozen(bn) and covered(m, None, x):
        stock_fallback(f"{type(bn).__name__}({getattr(bn, 'num_features', '?')}) in training mode behind a conv")
        y = bn(conv_bn_act(m, None, x))                     # BN in training mode: conv on HIP, statistics stock
    else:                                                   # documented stock-op fallbacks of the TRAINING forward (module docstring)
        stock_fallback(f"conv {tuple(m.weight.shape)} stride {m.stride[0]} groups {m.groups}" + (f" + {type(bn).__name__}" if bn is not None else ""))
        y = nn.Conv2d.forward(m, x) if bn is None else bn(nn.Conv2d.forward(m, x))
    if residual is not None:
        y = y + residual
    return F.relu(y) if act == ACT_RELU else y


def conv2d(m: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """m(x) with the HIP kernels where they apply."""
    return conv_bn_act(m, None, x)


def stem_frozen(trunk: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """maxpool(relu(bn1(conv1(x)))) of a frozen ResNet stem on the inference kernels (no autograd graph)."""
    B, _, H, W = x.shape
    dev = x.device
    with torch.no_grad():
        x4 = torch.empty(B * H * W, 4, dtype=torch.float32, device=dev)
        ops.nchw3_to_nhwc4(x.float().contiguous(), x4)
        sc, sf = _bn_fold(trunk.bn1)
        s_in = Segs.make(B, [(H, W)])
        s1 = ops.conv_out_segs(s_in, 7, 2, 3, 1)
        H1, W1 = s1.H[0], s1.W[0]
        y1 = torch.empty(s1.rows, 64, dtype=torch.float32, device=dev)
        from . import engine
        if engine.STEM_KERNEL:
            ops.stem7x7(Rows(x4), ops.pack_stem7_weight(trunk.conv1.weight), Rows(y1), B, H, W, sc, sf, ACT_RELU)
        else:
            ops.conv_call(Rows(x4), s_in, ops.pack_stem_weight(trunk.conv1.weight), Rows(y1), Cin=4, Cout=64, k=7, stride=2,
                          pad=3, scale=sc, shift=sf, act=ACT_RELU, stem=True)()
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        y2 = torch.empty(B * H2 * W2, 64, dtype=torch.float32, device=dev)
        ops.maxpool(Rows(y1), Rows(y2), B, H1, W1, 3, 2, 1)
    return from_rows(y2, B, H2, W2)


def stem_is_frozen(trunk: nn.Module, x: torch.Tensor) -> bool:
    return (not _STOCK and bn_is_frozen(trunk.bn1) and not trunk.conv1.weight.requires_grad and not x.requires_grad
            and _f32(x) and tuple(trunk.conv1.weight.shape) == (64, 3, 7, 7))
