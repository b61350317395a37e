"""Tensor-level wrappers over the C-ABI (include/fcosdet.h).  Every function enqueues HIP kernels on torch's
current stream and returns immediately; tensors must be CUDA (ROCm) fp32 and are never copied to the host.

`Rows` describes an NHWC activation as rows x channels with a channel stride / offset, so concatenations are
written in place (torch.cat in HISFcos.py:107,111 disappears) and slices are read without copies.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_EXP, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, ConvParams, FdError, Segs, check  # noqa: F401


CONV_LOG = None     # a list: conv_call appends (closure, description) of every launch it builds (tools/time_train_convs.py replays them); None = off
LAUNCHES = [0]      # C-ABI launches enqueued by this process (every wrapper asks for the stream once per launch): train_ops.SYNC_TRACE differences it


def _stream() -> int:
    LAUNCHES[0] += 1
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise FdError("pytorch_object_detection_amd runs on the GPU only (got a CPU tensor); there is no CPU "
                          "fallback — use the reference or oracle/ for CPU runs")


class Rows:
    """Channel view [rows, C] of a [rows, cs] buffer starting at channel `co`: fp32, or f16 (AMP activations stored as f16: the FD_PREC_F16 conv
    launches and the f16 forms of the elementwise kernels read / write them directly; cs / co stay in elements)."""
    __slots__ = ("buf", "cs", "co", "C")

    def __init__(self, buf: torch.Tensor, co: int = 0, C_: Optional[int] = None):
        assert buf.dim() == 2 and buf.dtype in (torch.float32, torch.float16) and buf.is_contiguous()
        _need_gpu(buf)
        self.buf, self.cs, self.co = buf, buf.shape[1], co
        self.C = buf.shape[1] - co if C_ is None else C_
        assert 0 <= co and co + self.C <= self.cs

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    @property
    def rows(self) -> int:
        return self.buf.shape[0]

    @property
    def f16(self) -> bool:
        return self.buf.dtype == torch.float16

    def slice(self, co: int, C_: int) -> "Rows":
        return Rows(self.buf, self.co + co, C_)

    def tensor(self) -> torch.Tensor:
        return self.buf[:, self.co:self.co + self.C]


def new_rows(rows: int, C_: int, device) -> Rows:
    return Rows(torch.empty(rows, C_, dtype=torch.float32, device=device))


# ---------------------------------------------------------------------------------------------------- weights
def _pad_cin32(w: torch.Tensor) -> torch.Tensor:
    """Zero input channels up to the next multiple of 32 (the kernel masks the matching activation reads: Cin % 4 == 0)."""
    i = w.shape[1]
    if i % 4:
        raise FdError(f"conv weights need Cin % 4 == 0 (got {i})")
    return torch.nn.functional.pad(w.detach(), (0, 0, 0, 0, 0, (-i) % 32)) if i % 32 else w


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW -> [Cout][Cin/32][KH][KW][32] contiguous: K runs (32-channel chunk, tap, channel-in-chunk), the order the
    conv kernel walks its K-tiles in (taps of one chunk adjacent -> shifted input re-reads stay in L1/L2)."""
    w = _pad_cin32(w)
    o, i, kh, kw = w.shape
    return w.detach().float().reshape(o, i // 32, 32, kh, kw).permute(0, 1, 3, 4, 2).contiguous()


def pack_conv_weight_f16x3(w: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 -> [Cout][Cin/32][KH][KW][2][32] f16: per K-tile of 32 the hi plane then the lo plane, where
    w = hi + lo * 2^-11 (hi = fp16(w) round-to-nearest, lo = fp16((w - hi) * 2^11)): FD_PREC_F16X3 operand format."""
    w = _pad_cin32(w)
    o, i, kh, kw = w.shape
    t = w.detach().float().reshape(o, i // 32, 32, kh, kw).permute(0, 1, 3, 4, 2)      # [O, I/32, KH, KW, 32]
    hi = t.half()
    lo = ((t - hi.float()) * 2048.0).half()
    return torch.stack([hi, lo], dim=-2).contiguous()                                   # [O, I/32, KH, KW, 2, 32]


def wave_ok(Cin: int, Cout: int, k: int, stride: int, pad: int) -> bool:
    """Layers FD_TILE_WAVE64 (fd_conv_wave.hip: wave-autonomous 64 x 64 tiles) covers: GEMM-addressed, Cin and Cout multiples of 32."""
    return k == 1 and stride == 1 and pad == 0 and Cin % 32 == 0 and Cout % 32 == 0


def pack_conv_weight_wave(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 1, 1] fp32 -> the MFMA-fragment-order operand of FD_TILE_WAVE64 (fd_pack_conv_weight_wave_f32), one HIP launch."""
    w = w.detach().float().contiguous()
    _need_gpu(w)
    co, ci = w.shape[0], w.shape[1]
    nb = _lib.lib().fd_conv_weight_wave_bytes(co, ci)
    if nb < 0 or w.numel() != co * ci:
        raise FdError(f"pack_conv_weight_wave: needs a 1x1 filter bank with Cin % 32 == 0 (got {tuple(w.shape)})")
    out = torch.empty(nb // 4, dtype=torch.float32, device=w.device)
    check(_lib.lib().fd_pack_conv_weight_wave_f32(w.data_ptr(), out.data_ptr(), co, ci, _stream()), "fd_pack_conv_weight_wave_f32")
    return out


def pack_stem_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout,3,7,7] -> [Cout][7][8][4], zero at kw=7 and c=3 (FD_CONV_STEM)."""
    co = w.shape[0]
    p = torch.zeros(co, 7, 8, 4, dtype=torch.float32, device=w.device)
    p[:, :, :7, :3] = w.detach().permute(0, 2, 3, 1).float()
    return p.contiguous()


def pack_stem7_weight(w: torch.Tensor) -> torch.Tensor:
    """[64,3,7,7] -> [7 filter rows][22][64]: k = 3 * column + channel, k = 21 zero (fd_stem7x7_nhwc4)."""
    if tuple(w.shape) != (64, 3, 7, 7):
        raise FdError(f"the stem kernel takes a [64, 3, 7, 7] filter bank (got {tuple(w.shape)})")
    p = torch.zeros(7, 22, 64, dtype=torch.float32, device=w.device)
    p[:, :21, :] = w.detach().float().permute(2, 3, 1, 0).reshape(7, 21, 64)
    return p.contiguous()


def stem7x7(x4: Rows, w722: torch.Tensor, y: Rows, N: int, H: int, W: int, scale=None, shift=None, act: int = ACT_NONE) -> None:
    """ResNet stem 7x7 s2 p3 (3 -> 64) + scale / shift + act on the [N][H][W][4] image rows: its own LDS-staged MFMA kernel."""
    _need_gpu(w722, scale, shift)
    if x4.cs != 4 or x4.co != 0 or y.C != 64:
        raise FdError("stem7x7: input must be the [rows][4] image buffer, output a 64-channel view")
    check(_lib.lib().fd_stem7x7_nhwc4(x4.ptr, w722.data_ptr(), scale.data_ptr() if scale is not None else None,
                                      shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, N, H, W, act, _stream()),
          "fd_stem7x7_nhwc4")


def stem7x7_pool(x4: Rows, w722: torch.Tensor, y: Rows, N: int, H: int, W: int, scale=None, shift=None) -> None:
    """ResNet stem 7x7 s2 p3 (3 -> 64) + scale / shift + ReLU + max-pool 3x3 s2 p1 in one call (fd_stem7x7_pool_nhwc4): y = the pooled map; the
    64-channel stride-2 map is never written."""
    _need_gpu(w722, scale, shift)
    if x4.cs != 4 or x4.co != 0 or y.C != 64:
        raise FdError("stem7x7_pool: input must be the [rows][4] image buffer, output a 64-channel view")
    check(_lib.lib().fd_stem7x7_pool_nhwc4(x4.ptr, w722.data_ptr(), scale.data_ptr() if scale is not None else None,
                                           shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, N, H, W, _stream()),
          "fd_stem7x7_pool_nhwc4")


def stem7x7_nchw(x: torch.Tensor, w722: torch.Tensor, y: Rows, scale=None, shift=None, act: int = ACT_NONE, pool: bool = False) -> None:
    """The stem (pool: + ReLU + max-pool 3x3 s2 p1) straight from the reference's fp32 [N, 3, H, W] input tensor (fd_stem7x7_nchw3): the planes are
    read by the patch loader, no [N][H][W][4] copy of the batch is made."""
    _need_gpu(x, w722, scale, shift)
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_contiguous() or y.C != 64:
        raise FdError("stem7x7_nchw: input must be a contiguous fp32 [N, 3, H, W] tensor, output a 64-channel view")
    N, _, H, W = x.shape
    check(_lib.lib().fd_stem7x7_nchw3(x.data_ptr(), w722.data_ptr(), scale.data_ptr() if scale is not None else None,
                                      shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, N, H, W, act, 1 if pool else 0, _stream()),
          "fd_stem7x7_nchw3")


def pack_dw_weight(w: torch.Tensor) -> torch.Tensor:
    """[C,1,3,3] -> [9][C]."""
    return w.detach().reshape(w.shape[0], 9).t().contiguous().float()


def pack_dwk_weight(w: torch.Tensor) -> torch.Tensor:
    """[C,1,K,K] -> [K*K][C] (fd_dwconv2d_nhwc)."""
    return w.detach().reshape(w.shape[0], -1).t().contiguous().float()


def pack_stem3_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout,3,K,K] -> [K*K][4][Cout] (fd_stem_conv_nhwc4); input channel 3 is zero."""
    co, ci, kh, kw = w.shape
    p = torch.zeros(kh * kw, 4, co, dtype=torch.float32, device=w.device)
    p[:, :ci, :] = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, ci, co)
    return p.contiguous()


def fold_bn(weight, bias, mean, var, eps: float = 1e-5, conv_bias: Optional[torch.Tensor] = None):
    """Frozen BatchNorm2d -> per-channel (scale, shift); a preceding conv bias is absorbed into the shift."""
    scale = (weight.detach().double() / torch.sqrt(var.detach().double() + eps))
    shift = bias.detach().double() - mean.detach().double() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


# ---------------------------------------------------------------------------------------------------- conv
def conv_call(x: Rows, segs: Segs, w_packed: torch.Tensor, y: Rows, *, Cin: int, Cout: int, k: int, stride: int = 1,
              pad: int = 0, dil: int = 1, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
              res: Optional[Rows] = None, act: int = ACT_NONE, act_c0: int = 0,
              seg_param: Optional[Sequence[float]] = None, stem: bool = False, tile: int = 0,
              tag: int = 0, precision: int = 0, ksplit: int = 1,
              workspace: Optional[torch.Tensor] = None, res_mask: bool = False, kw: Optional[int] = None,
              out_hw: Optional[Tuple[int, int]] = None, scatter: Optional[Tuple[int, int, int, int, int, int]] = None,
              gate: Optional[torch.Tensor] = None, w_frag: Optional[torch.Tensor] = None, gn_stats: Optional[torch.Tensor] = None,
              gn_groups: int = 0, gate_b: Optional[torch.Tensor] = None, gate_act: int = ACT_NONE,
              x2: Optional[Rows] = None, x2_stride: int = 1, x2_hw: Optional[Tuple[int, int]] = None, res_up: bool = False,
              sk_wgs: int = 0) -> Callable[[], None]:
    """Build the argument block once; the returned closure launches fd_conv2d_nhwc_f32 on the current stream.
    w_frag: the same weights in FD_TILE_WAVE64's fragment order (pack_conv_weight_wave), which makes that tile selectable."""
    _need_gpu(w_packed, scale, shift)
    p = ConvParams()
    p.x, p.w, p.y = x.ptr, w_packed.data_ptr(), y.ptr
    p.scale = scale.data_ptr() if scale is not None else None
    p.shift = shift.data_ptr() if shift is not None else None
    p.res = res.ptr if res is not None else None
    p.x_cs, p.x_co, p.y_cs, p.y_co = x.cs, x.co, y.cs, y.co
    if res is not None:
        p.res_cs, p.res_co = res.cs, res.co
    p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad, p.dil = Cin, Cout, k, (k if kw is None else kw), stride, pad, dil
    if out_hw is not None:
        p.out_H, p.out_W = out_hw
    if scatter is not None:           # (sy, sx, oy, ox, H, W): output pixel (i, j) -> (sy*i + oy, sx*j + ox) of an [N, H, W] map
        p.sc_sy, p.sc_sx, p.sc_oy, p.sc_ox, p.sc_H, p.sc_W = scatter
    p.act, p.act_c0, p.mode = act, act_c0, (_lib.CONV_STEM if stem else _lib.CONV_GENERIC)
    p.tile, p.tag, p.precision, p.ksplit = tile, tag, precision, ksplit
    p.res_mode = 1 if (res_mask and res is not None) else 0
    if res_up:        # `res` = the coarser level [batch, H / 2, W / 2]: added after the activation at (i / 2, j / 2) (fd_conv_params.res_mode 2)
        if res is None or res_mask or segs.nseg != 1 or res.rows != segs.batch * (segs.H[0] // 2) * (segs.W[0] // 2):
            raise FdError("conv_call: res_up needs a single-level conv and a residual of batch * (H / 2) * (W / 2) rows, no mask")
        p.res_mode = 2
    if gate is not None:             # [levels * batch, >= Cin] fp32: per-(level, image, input channel) gate applied in the loader (1x1 convs)
        _need_gpu(gate, gate_b)
        if gate.dim() != 2 or gate.shape[0] != segs.batch * segs.nseg or gate.stride(1) != 1 or gate.dtype != torch.float32:
            raise FdError("conv gate must be a [levels * batch, C] fp32 tensor with unit channel stride")
        p.gate, p.gate_cs = gate.data_ptr(), gate.stride(0)
        if gate_b is not None:       # x' = gate_act(x * gate + gate_b): the preceding GroupNorm's affine + activation (groupnorm_from_rowstats coef)
            if gate_b.shape != gate.shape or gate_b.stride() != gate.stride() or gate_b.dtype != torch.float32:
                raise FdError("conv gate_b must have gate's shape and strides")
            p.gate_b, p.gate_act = gate_b.data_ptr(), gate_act
    if gn_stats is not None:         # [rows, gn_groups, 2] fp32: row-group (sum, sum of squares) of the stored output (GroupNorm fused into the producer)
        _need_gpu(gn_stats)
        if gn_stats.dtype != torch.float32 or not gn_stats.is_contiguous() or gn_stats.numel() < y.rows * gn_groups * 2:
            raise FdError("conv gn_stats must be a contiguous fp32 buffer of rows x gn_groups x 2")
        p.gn_stats, p.gn_groups = gn_stats.data_ptr(), gn_groups
    if w_frag is not None:
        _need_gpu(w_frag)
        p.w_frag = w_frag.data_ptr()
    if x2 is not None:               # K-concatenated second source (1x1 layers): x2 [batch, x2_hw[0], x2_hw[1]] sampled with x2_stride; w covers Cin + x2.C channels
        p.x2, p.x2_cs, p.x2_co, p.x2_Cin, p.x2_stride = x2.ptr, x2.cs, x2.co, x2.C, x2_stride
        p.x2_H, p.x2_W = x2_hw
    if workspace is not None:
        _need_gpu(workspace)
        p.workspace, p.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    io = (1 if x.f16 else 0) | (2 if y.f16 else 0) | (4 if (res is not None and res.f16) else 0)
    if io:                           # f16 activation maps (AMP): the FD_PREC_F16 kernels read / write them directly (fd_conv_params.io_f16)
        if precision != _lib.PREC_F16:
            raise FdError("conv_call: f16 input / output / residual maps need precision = FD_PREC_F16")
        p.io_f16 = io
    if sk_wgs:                       # FD_TILE_WINOGRAD4 as a persistent stream-K grid of sk_wgs workgroups; workspace = sk_workspace(sk_wgs) (zeroed flags)
        p.sk_wgs = sk_wgs
    if seg_param is not None:
        for i, v in enumerate(seg_param):
            p.seg_param[i] = float(v)
    p.segs = segs
    fn = _lib.lib().fd_conv2d_nhwc_f32
    ref = C.byref(p)
    keep = (x, w_packed, y, scale, shift, res, p, workspace, gate, w_frag, gn_stats, gate_b, x2)

    def run(_keep=keep):
        check(fn(ref, _stream()), "fd_conv2d_nhwc_f32")

    run.params = p  # type: ignore[attr-defined]   (autotuning rewrites p.tile in place)
    if CONV_LOG is not None:
        eb = lambda r: 2 if r.f16 else 4
        rows_in = sum(segs.batch * segs.H[i] * segs.W[i] for i in range(segs.nseg))
        CONV_LOG.append((run, dict(Cin=Cin, Cout=Cout, k=k, stride=stride, dil=dil, rows_in=rows_in, rows_out=y.rows, io=io, precision=precision, res=res is not None,
                                   bytes=rows_in * Cin * eb(x) + y.rows * Cout * eb(y) + (y.rows * Cout * eb(res) if res is not None and not res_up else 0)
                                   + w_packed.numel() * w_packed.element_size(), flops=2 * y.rows * Cout * Cin * k * (k if kw is None else kw))))
    return run


def sk_workspace(sk_wgs: int, device) -> torch.Tensor:
    """Workspace of the persistent stream-K form of an F(4x4) launch (fd_conv_params.sk_wgs): flags (zero, and left zero by every launch) + one 128 KB slot per
    workgroup.  One per concurrently running launch; launches on one stream may share it."""
    n = _lib.lib().fd_conv_sk_workspace_bytes(sk_wgs)
    if n < 0:
        raise FdError(f"fd_conv_sk_workspace_bytes({sk_wgs}): sk_wgs must be a multiple of 8 in 8 .. 1024")
    ws = torch.empty((n + 3) // 4, dtype=torch.float32, device=device)
    ws[:2048].zero_()            # the fixed 8 KB header
    return ws


# F(4x4) launches as a persistent grid with a work queue per XCD (fd_conv_params.sk_wgs): FD_W4_SK=0 keeps every layer on the plain launch
W4_SK = os.environ.get("FD_W4_SK", "1") != "0"


def conv_sk(run, sk_wgs: int, ws: torch.Tensor) -> Callable[[], None]:
    """conv_call()'s F(4x4) launch in its persistent form: sk_wgs workgroups, workspace ws = sk_workspace(>= sk_wgs)."""
    q = _lib.ConvParams()
    C.memmove(C.byref(q), C.byref(run.params), C.sizeof(q))
    q.sk_wgs, q.wg_first, q.wg_count = sk_wgs, 0, 0
    q.workspace, q.workspace_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    fn = _lib.lib().fd_conv2d_nhwc_f32
    ref = C.byref(q)

    def run_sk(_keep=(run, q, ws)):
        check(fn(ref, _stream()), "fd_conv2d_nhwc_f32")

    run_sk.params = q  # type: ignore[attr-defined]
    return run_sk


def wino4_sk_choice(run, key: str, ws: torch.Tensor, reps: int = 5) -> int:
    """Plain launch (0) or the persistent form with 240 / 256 workgroups for one F(4x4) layer: the committed table first ("w4sk|" keys of
    tuned/gfx950_tiles.json, measured on MI355X); a miss is timed on the spot under FD_AUTOTUNE, else stays plain.  The persistent form wins where the
    plain grid ends in a mostly idle round of workgroups and the chunk loop is long enough to carry the pieces' fixed cost (cls_logits, the dilated
    HisBlock conv4, the head tower); elsewhere it costs ~2 % (an extra barrier and a queue claim per item)."""
    if not W4_SK:
        return 0
    table = _tune_table()
    key = "w4sk|" + key
    if _TUNE_MODE != "force" and key in table:
        return int(table[key])
    if _TUNE_MODE == "0":
        return 0
    best, best_t = 0, float("inf")
    trace = os.environ.get("FD_W4_SK_TRACE")
    for wgs in (0, 256, 240):
        call = conv_sk(run, wgs, ws) if wgs else run
        if trace:
            p_ = run.params
            print(f"[w4sk] {key} sk_wgs {wgs} x_cs {p_.x_cs} x_co {p_.x_co} y_cs {p_.y_cs} y_co {p_.y_co} res {bool(p_.res)} act {p_.act} tag {p_.tag} ksplit {p_.ksplit} "
                  f"wg_count {p_.wg_count}", file=sys.stderr, flush=True)
            call()
            torch.cuda.synchronize()
            print("[w4sk]   ok", file=sys.stderr, flush=True)
        try:
            call()
        except FdError as e:
            if e.rc != _lib.E_UNSUPPORTED:
                raise
            continue
        t = min(_time_launches(call, reps, False) for _ in range(3))
        if t < best_t * 0.98:
            best, best_t = wgs, t
    table[key] = best
    return best


def conv_workgroups(run) -> Tuple[int, int]:
    """(workgroups, non-empty workgroups) of the launch a conv_call() callable describes -- or of its slice, if it is one (FD_TILE_WINOGRAD4 only)."""
    n = _lib.lib().fd_conv_workgroups(C.byref(run.params))
    live = _lib.lib().fd_conv_workgroups_live(C.byref(run.params))
    if n < 0 or live < 0:
        raise FdError("fd_conv_workgroups: " + _lib.lib().fd_last_error().decode(errors="replace"), int(min(n, live)))
    return (run.params.wg_count or n), live


def conv_wg_slice(run, first: int, count: int, tag: Optional[int] = None) -> Callable[[], None]:
    """The workgroups [first, first + count) of conv_call()'s launch as a launch of their own (fd_conv_params.wg_first / wg_count; FD_TILE_WINOGRAD4, first % 8 == 0).
    The slices of a partition of the grid together compute the layer; `tag` overrides the kernel tag (1 = the instantiation rocprof lists as the head tower)."""
    q = _lib.ConvParams()
    C.memmove(C.byref(q), C.byref(run.params), C.sizeof(q))
    q.wg_first, q.wg_count = first, count
    if tag is not None:
        q.tag = tag
    fn = _lib.lib().fd_conv2d_nhwc_f32
    ref = C.byref(q)

    def run_slice(_keep=(run, q)):
        check(fn(ref, _stream()), "fd_conv2d_nhwc_f32")

    run_slice.params = q  # type: ignore[attr-defined]
    return run_slice


def b2b_ok(K1: int, N1: int, N2: int) -> bool:
    """Shapes fd_conv1x1_b2b_f32 covers (two 1x1 stride-1 convs back to back: a bottleneck's conv3 and the next block's conv1)."""
    return K1 % 32 == 0 and N1 % 64 == 0 and N2 in (64, 128)


def conv_b2b_call(x: Rows, w1_frag: torch.Tensor, y: Rows, w2_frag: torch.Tensor, z: Rows, *, K1: int, N1: int, N2: int,
                  scale1: Optional[torch.Tensor] = None, shift1: Optional[torch.Tensor] = None, res: Optional[Rows] = None, act1: int = ACT_NONE,
                  scale2: Optional[torch.Tensor] = None, shift2: Optional[torch.Tensor] = None, act2: int = ACT_NONE) -> Callable[[], None]:
    """y = act1(x . W1^T * scale1 + shift1 + res); z = act2(y . W2^T * scale2 + shift2) in ONE launch (fd_conv1x1_b2b_f32): y is written (it is the next
    residual) but never read back.  w1_frag / w2_frag = pack_conv_weight_wave of the [N1, K1] / [N2, N1] filter banks."""
    _need_gpu(w1_frag, w2_frag, scale1, shift1, scale2, shift2)
    p = _lib.B2BParams()
    p.x, p.w1_frag, p.y, p.w2_frag, p.z = x.ptr, w1_frag.data_ptr(), y.ptr, w2_frag.data_ptr(), z.ptr
    p.scale1 = scale1.data_ptr() if scale1 is not None else None
    p.shift1 = shift1.data_ptr() if shift1 is not None else None
    p.scale2 = scale2.data_ptr() if scale2 is not None else None
    p.shift2 = shift2.data_ptr() if shift2 is not None else None
    p.res = res.ptr if res is not None else None
    p.x_cs, p.x_co, p.y_cs, p.y_co, p.z_cs, p.z_co = x.cs, x.co, y.cs, y.co, z.cs, z.co
    if res is not None:
        p.res_cs, p.res_co = res.cs, res.co
    p.K1, p.N1, p.N2, p.act1, p.act2, p.rows = K1, N1, N2, act1, act2, y.rows
    fn = _lib.lib().fd_conv1x1_b2b_f32
    ref = C.byref(p)
    keep = (x, w1_frag, y, w2_frag, z, scale1, shift1, scale2, shift2, res, p)

    def run(_keep=keep):
        check(fn(ref, _stream()), "fd_conv1x1_b2b_f32")

    run.params = p  # type: ignore[attr-defined]
    return run


_TUNE_CACHE: dict = {}
_TUNE_FILE = os.environ.get("FD_TILE_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_tiles.json")   # (FD_TILE_TABLE: A/B of two tables)
# FD_AUTOTUNE: "0" (default) = committed table, misses use the shape heuristic (no first-call latency);
#              "1" = time the misses on the spot; "force" = re-time everything (bench.py --save-tuning writes the table)
_TUNE_MODE = os.environ.get("FD_AUTOTUNE", "0")
_TUNE_LOADED = False


def _tune_table() -> dict:
    global _TUNE_LOADED
    if not _TUNE_LOADED:
        _TUNE_LOADED = True
        try:
            with open(_TUNE_FILE) as f:
                _TUNE_CACHE.update(json.load(f))
        except Exception:
            pass
    return _TUNE_CACHE


def save_tune_table() -> None:
    os.makedirs(os.path.dirname(_TUNE_FILE), exist_ok=True)
    with open(_TUNE_FILE, "w") as f:
        json.dump(dict(sorted(_TUNE_CACHE.items())), f, indent=0)


KSPLIT_MAX = 8
import re as _re
_BKEY = _re.compile(r"^(f16x3\|)?B(\d+)(\|.*)$")


def heuristic_conv(M: int, Cout: int, KT: int, have_ws: bool) -> int:
    """Tile | ksplit << 8 for a shape that is not in the tuned table, distilled from the table: single-LDS-buffer
    tiles (3-4 blocks/CU) whenever they still give >= 2 workgroups per CU, otherwise 64x64 tiles with the K loop
    split so that about 2-4 workgroups per CU exist."""
    if Cout <= 32:
        return 5
    if Cout <= 64:
        tiles, bm, bn = 4, 64, 64
    elif Cout <= 96:
        return 12
    else:
        def nblk(bm_, bn_):
            return -(-M // bm_) * -(-Cout // bn_)
        if nblk(128, 128) >= 768:
            return 7
        if nblk(64, 128) >= 512:
            return 9
        tiles, bm, bn = 4, 64, 64
    n = -(-M // bm) * -(-Cout // bn)
    ks = 1
    if have_ws:
        while ks < KSPLIT_MAX and n * ks < 512 and KT >= 8 * ks:
            ks *= 2
    return tiles | ((ks if ks > 1 else 0) << 8)


_PAIR_STREAMS: list = []


def _time_launches(run: Callable[[], None], reps: int, pair: bool) -> float:
    """Device milliseconds per launch.  pair=True: the launch runs beside a twin of itself on a second stream (what a layer
    meets under pipeline.TwoLanePipeline, where the other batch's kernels fill its tail): per-launch share of the pair."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not pair:
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    if not _PAIR_STREAMS:
        _PAIR_STREAMS.extend([torch.cuda.Stream(), torch.cuda.Stream()])
    cur = torch.cuda.current_stream()
    e0.record()
    for st in _PAIR_STREAMS:
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for _ in range(reps):
                run()
    for st in _PAIR_STREAMS:
        cur.wait_stream(st)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (2 * reps)


def autotune_conv(run: Callable[[], None], key: str, M: int, Cout: int, KT: int, reps: int = 3, pair: bool = False) -> int:
    """Block-tile (+ split-K factor) choice for one conv launch.  Looked up in the committed table
    (tuned/gfx950_tiles.json, measured on MI355X) first; a miss is timed on the spot under every sensible tile — and,
    for maps with few tiles, K split 2/4/8 ways — (best of 3 batches of `reps` launches) and remembered for the
    process (bench.py --save-tuning writes the table back).  Returns tile | ksplit << 8 (tile 0 = heuristic)."""
    p = run.params  # type: ignore[attr-defined]
    table = _tune_table()

    def apply(code: int) -> int:
        """Install a table / heuristic / timed code on the launch and return the code ACTUALLY applied (what plan.tiles records)."""
        tile, ks = code & 0xFF, max(1, code >> 8)
        if tile == _lib.WAVE_TILE and not p.w_frag:
            # the key does not say whether the caller packed the weights in MFMA fragment order (train_ops._conv_launch never does, plans built
            # with FD_WAVE_TILE=0 neither): the wave-autonomous tile is then not available -- fall back to the library's heuristic tile
            tile, ks = 0, 1
        if ks > 1 and (not p.workspace or p.gn_stats):        # (split-K needs the scratch; a row-statistics epilogue has no combine launch)
            ks = 1
        p.tile, p.ksplit = tile, ks
        return tile | ((ks if ks > 1 else 0) << 8)

    base = key
    if pair:                         # tiles chosen for throughput beside another batch: own table entries, serial ones as fallback
        key = "pair|" + key
    if _TUNE_MODE != "force" and key in table:
        return apply(int(table[key]))
    if _TUNE_MODE == "0":
        if base in table:
            return apply(int(table[base]))
        # the table is keyed by exact shape incl. batch: for another batch size take the measured choice of the nearest larger
        # (else nearest smaller) batch of the same layer geometry, when its tile count says the same regime; else the heuristic
        m = _BKEY.match(base)
        if m:
            pre, b0, rest = m.group(1) or "", int(m.group(2)), m.group(3)
            near = []
            for k2 in table:
                m2 = _BKEY.match(k2)
                if m2 and (m2.group(1) or "") == pre and m2.group(3) == rest:
                    b2 = int(m2.group(2))
                    near.append((abs(b2 - b0) + (0.5 if b2 < b0 else 0.0), b2, k2))
            if near:
                _, b2, k2 = min(near)
                code = int(table[k2])
                if 0.5 <= b0 / b2 <= 2.0 and (code >> 8) <= 1:        # (split-K factors depend on the tile count: not transferred)
                    return apply(code)
        return apply(heuristic_conv(M, Cout, KT, bool(p.workspace)))
    cands = [(0, 1)]
    # FD_TILE_128x128_PATCH only exists for 3x3 stride-1 'same' convs whose (128-row tile + halo) patch fits its LDS budget, without split-K
    wmax = max(p.segs.W[i] for i in range(p.segs.nseg))
    patch_ok = (p.KH == 3 and p.KW == 3 and p.stride == 1 and p.pad == p.dil and p.out_H <= 0 and p.sc_H <= 0
                and 128 + 2 * p.dil * (wmax + 1) <= 320)
    wave_tile_ok = bool(p.w_frag) and wave_ok(p.Cin, p.Cout, p.KH, p.stride, p.pad) and p.KW == 1 and p.act_c0 % 32 == 0
    for tid, (bm, bn) in _lib.TILES.items():
        if (tid == _lib.PATCH_TILE and not patch_ok) or (tid == _lib.WAVE_TILE and not wave_tile_ok):
            continue
        padded = -(-Cout // bn) * bn
        if padded <= max(32, int(Cout * 1.34)) and not (bn == 32 and Cout > 32):
            cands.append((tid, 1))
            ntile = -(-M // bm) * -(-Cout // bn)
            if p.workspace and tid not in (_lib.PATCH_TILE, _lib.WAVE_TILE):
                for ks in (2, 4, 8):
                    if ks <= KSPLIT_MAX and KT >= 4 * ks and ntile * ks <= 2048 and ntile < 1024:
                        cands.append((tid, ks))
    best, best_t = 0, float("inf")
    for tid, ks in cands:
        apply(tid | (ks << 8))
        try:
            run()  # warm
        except FdError as e:          # a tile the library has no kernel for on this layer (FD_E_UNSUPPORTED): not a candidate
            if e.rc != _lib.E_UNSUPPORTED:
                raise
            continue
        t = float("inf")
        for _ in range(3):
            t = min(t, _time_launches(run, reps, pair))
        if t < best_t * 0.985:  # an earlier (simpler) candidate wins near-ties
            best, best_t = tid | ((ks if ks > 1 else 0) << 8), t
    table[key] = best
    return apply(best)


def conv_wgrad(x: Rows, dy: Rows, segs_in: Segs, *, Cin: int, Cout: int, k: int, stride: int = 1, pad: int = 0,
               dil: int = 1, nsplit: int = 0, scale: Optional[torch.Tensor] = None, oihw: bool = False, precision: int = 0) -> torch.Tensor:
    """Weight gradient of conv(x) w.r.t. its weights given dy (rows in output geometry): [Cout, k, k, Cin] (OHWI), or
    torch's [Cout, Cin, k, k] with oihw=True; `scale` [Cout] multiplies it per output channel (folded frozen BN).
    precision = FD_PREC_F16 (AMP): operands rounded to f16, fp32 accumulation (fd_conv_wgrad_params.precision)."""
    out_rows = conv_out_segs(segs_in, k, stride, pad, dil).rows
    dev = x.buf.device
    dw = torch.empty((Cout, Cin, k, k) if oihw else (Cout, k, k, Cin), dtype=torch.float32, device=dev)
    nb = _lib.lib().fd_conv_wgrad_workspace_bytes(out_rows, Cin, Cout, k, k) if nsplit <= 0 else (nsplit + 8) * dw.numel() * 4
    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=dev)
    p = _lib.WgradParams()
    p.nsplit = max(nsplit, 0)
    p.layout = 1 if oihw else 0
    p.scale = scale.data_ptr() if scale is not None else None
    p.x, p.dy, p.dw = x.ptr, dy.ptr, dw.data_ptr()
    p.x_cs, p.x_co, p.dy_cs, p.dy_co = x.cs, x.co, dy.cs, dy.co
    p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad, p.dil = Cin, Cout, k, k, stride, pad, dil
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    p.segs = segs_in
    p.precision = precision
    p.io_f16 = (1 if x.f16 else 0) | (2 if dy.f16 else 0)       # (f16 operand maps: FD_PREC_F16 -- the library rejects anything else)
    check(_lib.lib().fd_conv2d_bwd_weight_f32(C.byref(p), _stream()), "fd_conv2d_bwd_weight_f32")
    if CONV_LOG is not None:
        def rerun(_keep=(x, dy, dw, ws, scale, p)):
            check(_lib.lib().fd_conv2d_bwd_weight_f32(C.byref(p), _stream()), "fd_conv2d_bwd_weight_f32")
        rerun.params = p  # type: ignore[attr-defined]
        CONV_LOG.append((rerun, dict(kind="wgrad", Cin=Cin, Cout=Cout, k=k, stride=stride, dil=dil, rows_in=segs_in.rows, rows_out=out_rows, io=p.io_f16, precision=precision,
                                     res=False, bytes=segs_in.rows * Cin * (2 if x.f16 else 4) + out_rows * Cout * (2 if dy.f16 else 4) + dw.numel() * 4,
                                     flops=2 * out_rows * Cout * Cin * k * k)))
    return dw


def pack_conv_weight_hip(w: torch.Tensor, scale: Optional[torch.Tensor] = None, dgrad: bool = False, f16: bool = False) -> torch.Tensor:
    """pack_conv_weight (dgrad=False) or dgrad_weight with a per-output-channel scale (dgrad=True) as one HIP launch; f16=True: the
    (hi, lo) f16 operand format of FD_PREC_F16 / FD_PREC_F16X3 (same byte size, returned as an fp32-typed buffer)."""
    w = w.detach().contiguous()
    co, ci, kh, kw = w.shape
    out = torch.empty((ci, co // 32, kh, kw, 32) if dgrad else (co, ci // 32, kh, kw, 32), dtype=torch.float32, device=w.device)
    check(_lib.lib().fd_pack_conv_weight_f32(w.data_ptr(), scale.data_ptr() if (scale is not None and dgrad) else None,
                                             out.data_ptr(), co, ci, kh, kw, (1 if dgrad else 0) | (4 if f16 else 0), _stream()),
          "fd_pack_conv_weight_f32")
    return out


F16K64 = os.environ.get("FD_AMP_K64", "1") != "0"       # "0": AMP convs stay on the K-tile-32 f16 instantiations of the fp32 kernel


def f16k64_ok(Cin: int, Cout: int) -> bool:
    """Shapes FD_TILE_F16K64 covers (reduction width a multiple of 64, 4-channel aligned output)."""
    return Cin % 64 == 0 and Cout % 4 == 0


def pack_conv_weight_f16k64(w: torch.Tensor, scale: Optional[torch.Tensor] = None, dgrad: bool = False) -> torch.Tensor:
    """OIHW fp32 -> FD_TILE_F16K64's operand: f16 [N][K/64][KH][KW][64] (dgrad=True: the flipped / transposed / per-Cout scaled weights of the data-gradient conv),
    one HIP launch (fd_pack_conv_weight_f32 mode | 16); returned as an fp32-typed buffer of half the element count."""
    w = w.detach().contiguous()
    co, ci, kh, kw = w.shape
    out = torch.empty(co * ci * kh * kw // 2, dtype=torch.float32, device=w.device)
    check(_lib.lib().fd_pack_conv_weight_f32(w.data_ptr(), scale.data_ptr() if (scale is not None and dgrad) else None,
                                             out.data_ptr(), co, ci, kh, kw, (1 if dgrad else 0) | 16, _stream()), "fd_pack_conv_weight_f32")
    return out


def pack_conv_weight_wino(w: torch.Tensor, scale: Optional[torch.Tensor] = None, dgrad: bool = False) -> torch.Tensor:
    """OIHW [Cout, Cin, 3, 3] -> the Winograd F(2x2, 3x3) operand of FD_TILE_WINOGRAD: U = G g G^T per (cout, cin), packed
    [ceil(N/32)][K/8][16][32][8] (N = Cout, K = Cin; dgrad=True: the flipped / transposed weights of the data-gradient conv,
    N = Cin, K = Cout, times an optional per-Cout scale).  One HIP launch (fd_wino_pack_weights_f32)."""
    w = w.detach().float().contiguous()
    _need_gpu(w, scale)
    co, ci, kh, kw = w.shape
    if kh != 3 or kw != 3:
        raise FdError("Winograd weights need a 3x3 filter")
    n, k = (ci, co) if dgrad else (co, ci)
    nbytes = _lib.lib().fd_wino_weight_bytes(n, k)
    if nbytes < 0:
        raise FdError(f"Winograd weights need a reduction width that is a multiple of 8 (got {k})")
    out = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    check(_lib.lib().fd_wino_pack_weights_f32(w.data_ptr(), scale.data_ptr() if (scale is not None and dgrad) else None,
                                              out.data_ptr(), co, ci, 1 if dgrad else 0, _stream()), "fd_wino_pack_weights_f32")
    return out


def pack_conv_weight_wino4(w: torch.Tensor, scale: Optional[torch.Tensor] = None, dgrad: bool = False) -> torch.Tensor:
    """OIHW [Cout, Cin, 3, 3] -> the Winograd F(4x4, 3x3) operand of FD_TILE_WINOGRAD4 (fd_wino4_pack_weights_f32), one HIP launch;
    dgrad=True: the flipped / transposed weights of the data-gradient conv (N = Cin, K = Cout) times an optional per-Cout scale."""
    w = w.detach().float().contiguous()
    _need_gpu(w, scale)
    co, ci, kh, kw = w.shape
    if kh != 3 or kw != 3:
        raise FdError("Winograd weights need a 3x3 filter")
    n, k = (ci, co) if dgrad else (co, ci)
    nbytes = _lib.lib().fd_wino4_weight_bytes(n, k)
    if nbytes < 0:
        raise FdError(f"Winograd weights need a reduction width that is a multiple of 8 (got {k})")
    out = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    check(_lib.lib().fd_wino4_pack_weights_f32(w.data_ptr(), scale.data_ptr() if (scale is not None and dgrad) else None,
                                               out.data_ptr(), co, ci, 1 if dgrad else 0, _stream()), "fd_wino4_pack_weights_f32")
    return out


NARROW = os.environ.get("FD_NARROW", "1") != "0"     # "0": layers of <= 8 output channels stay on the Winograd / direct MFMA kernels
NARROW_MIN_TILES = 128     # below this many 16 x 16 tiles (batch-1 plans) the MFMA kernels' split-K forms are the lower-latency choice


def narrow_ok(Cin: int, Cout: int, k: int, stride: int, pad: int, dil: int) -> bool:
    """Shapes FD_TILE_NARROW covers (fd_conv_narrow.hip: the vector-unit kernel for 3x3 convs of <= 8 output channels)."""
    return k == 3 and stride == 1 and pad == 1 and dil == 1 and 1 <= Cout <= 8 and Cin % 16 == 0


def narrow_tiles(segs: Segs) -> int:
    return sum(segs.batch * -(-h // 16) * -(-w // 16) for h, w in segs.level_hw())


def pack_conv_weight_narrow(w: torch.Tensor) -> torch.Tensor:
    """OIHW [Cout <= 8, Cin, 3, 3] -> [Cin / 16][3 r][4 quads][3 q][4 k][8 couts] (FD_TILE_NARROW; zero filters past Cout): one (chunk, filter row,
    channel quad, filter column) step is a [4 channels][8 couts] block that the kernel keeps in LDS and reads one row of per lane."""
    w = w.detach().float()
    co, ci, kh, kw = w.shape
    if kh != 3 or kw != 3 or not 1 <= co <= 8 or ci % 16:
        raise FdError(f"narrow-conv weights need a 3x3 filter bank with Cout <= 8 and Cin % 16 == 0 (got {tuple(w.shape)})")
    if co < 8:
        w = torch.cat([w, torch.zeros(8 - co, ci, 3, 3, dtype=w.dtype, device=w.device)], 0)
    # (co, ch, c4, k, r, q) -> (ch, r, c4, q, k, co)
    return w.reshape(8, ci // 16, 4, 4, 3, 3).permute(1, 4, 2, 5, 3, 0).contiguous()


def wino4_ok(Cin: int, Cout: int, k: int, stride: int, pad: int, dil: int) -> bool:
    """Shapes FD_TILE_WINOGRAD4 covers."""
    return k == 3 and stride == 1 and pad == dil and dil in (1, 2) and Cin % 8 == 0 and Cout % 4 == 0


# FD_WINOGRAD4: "1" (default) = layers the F(4x4, 3x3) kernel covers run on it where wino4_choice's cost model says it beats F(2x2, 3x3);
# "0" = never (the round-2 plans); "force" = wherever it applies (tests exercise the kernel inside whole models this way)
WINO4_MODE = os.environ.get("FD_WINOGRAD4", "1")
_W4_FIXED_US = 12.0       # the cost model's per-workgroup prologue + epilogue term, microseconds (fitted; 8 moved more layers onto F(4x4) and lost: profiles/r03y_layer_times_w4_fixed8.tsv)


def wino4_tiles(segs: Segs, dil: int = 1) -> int:
    """4x4 output tiles FD_TILE_WINOGRAD4 enumerates: per level, image and dilation parity class ceil(ceil(H/dil)/4) x ceil(ceil(W/dil)/4)."""
    return sum(segs.batch * dil * dil * (-(-(-(-h // dil)) // 4)) * (-(-(-(-w // dil)) // 4)) for h, w in segs.level_hw())


def wino4_choice(segs: Segs, Cin: int, Cout: int, dil: int = 1, allow_split: bool = True) -> Tuple[bool, int]:
    """(F(4x4, 3x3) instead of F(2x2, 3x3) / the direct kernel?, its split-K factor).  One F(4x4) workgroup owns 32 tiles x 64 couts and a whole CU
    (146 KB of LDS, 2 x 256-register waves per SIMD), so its time goes in ROUNDS of 256 workgroups:
      t_w4(ks) = ceil(ks * workgroups / 256) * ((1.0 + 2.0 * live) us * Cin / 8 / ks + 12 us)  [+ 10 us for the combine launch, ks > 1]
    live = the fraction of the workgroups' 32-cout blocks below Cout (the waves of a dead block skip their MFMAs).  Fitted to the in-plan step
    times of profiles/r03y_layer_times_w4*.tsv at batch 16: head tower 0.94 ms in 9 rounds, layer3.conv2 0.118 in 1, layer2.conv2 0.150 in 2,
    layer1.conv2 0.179 in 4 -- against 0.182 on F(2x2): the break-even case.  Split-K (layer4's 20 x 20 maps: 104 workgroups for 256 CUs) halves
    the chunk loop per workgroup where that fills the chip; there is no row-statistics epilogue (add_conv keeps those launches on F(2x2))."""
    if WINO4_MODE == "0":
        return False, 1
    if WINO4_MODE == "force":
        return True, 1
    wgs = -(-wino4_tiles(segs, dil) // 32) * -(-Cout // 64)
    live = -(-Cout // 32) / (2.0 * -(-Cout // 64))
    nc = Cin // 8
    best_t, best_ks = None, 1
    for ks in ((1, 2, 4) if allow_split else (1,)):
        if ks > 1 and nc < 8 * ks:
            break
        t = -(-wgs * ks // 256) * ((1.0 + 2.0 * live) * nc / ks + _W4_FIXED_US) + (10.0 if ks > 1 else 0.0)
        if best_t is None or t < best_t * 0.9:
            best_t, best_ks = t, ks
    return best_t < 0.95 * _wino_times(segs, Cin, Cout, dil, allow_split)[0], best_ks


def wino_ok(Cin: int, Cout: int, k: int, stride: int, pad: int, dil: int) -> bool:
    """Shapes FD_TILE_WINOGRAD covers (the output / residual views must also be 16-byte addressable)."""
    return k == 3 and stride == 1 and pad == dil and dil in (1, 2) and Cin % 8 == 0 and Cout % 4 == 0


def wino_tiles(segs: Segs, dil: int) -> int:
    """2x2 output tiles the Winograd kernel enumerates: per level, image and dilation parity class ceil(ceil(H/dil)/2) x ceil(ceil(W/dil)/2)."""
    return sum(segs.batch * dil * dil * ((-(-h // dil) + 1) // 2) * ((-(-w // dil) + 1) // 2) for h, w in segs.level_hw())


# FD_WINOGRAD=force: every layer the Winograd kernel covers runs on it, whatever its size (tests exercise the kernel on small maps this way)
WINO_FORCE = os.environ.get("FD_WINOGRAD", "1") == "force"


def wino_choice(segs: Segs, Cin: int, Cout: int, dil: int, allow_split: bool = True) -> Tuple[bool, int]:
    """(use the Winograd kernel?, its split-K factor) for a 3x3 stride-1 layer both kernels cover.  Without split-K one Winograd
    workgroup walks all Cin (1.5 us per 8 channels), so a map with few tiles is latency-bound on it while the direct kernel splits K over
    workgroups (floor ~21 us, two launches); wide maps are throughput-bound and Winograd's 2.25x fewer MFMAs win.  A small cost model
    fitted to MI355X measurements (batch 1 .. 16 at 512^2 / 640^2: profiles/r02z_layer_times.tsv, r02z_bench_latency_b*.json):
      t_wino(ks) = (1.5 * Cin / 8 / ks + 3) us * max(1, ks * workgroups / (0.55 * resident slots))  [+ 7 us for the combine launch, ks > 1]
      t_direct   = max(21 us [narrow Cout <= 96: no split-K there, 1.9 us per K-tile], FLOPs / 95 TFLOP/s)"""
    if WINO_FORCE:
        return True, 1
    best_t, best_ks, t_d = _wino_times(segs, Cin, Cout, dil, allow_split)[1:]
    return best_t < t_d, best_ks


def _wino_times(segs: Segs, Cin: int, Cout: int, dil: int, allow_split: bool):
    """The cost model of wino_choice: (min(t_wino, t_direct), t_wino at its best split, that split, t_direct), microseconds."""
    T = wino_tiles(segs, dil)
    mt = -(-T // 32)
    nch = 4 if (Cout % 128 == 0 and Cout >= 256 and mt * (Cout // 128) >= 192) else 2      # as fd_launch_conv_wino chooses
    wgs = mt * -(-Cout // (32 * nch))
    slots = 256 * (2 if nch == 2 else 1)
    nc = Cin // 8
    best_t, best_ks = None, 1
    for ks in ((1, 2, 4, 8) if allow_split else (1,)):
        if ks > 1 and nc < 4 * ks:
            break
        t = (1.5 * nc / ks + 3.0) * max(1.0, ks * wgs / (0.55 * slots)) + (7.0 if ks > 1 else 0.0)
        if best_t is None or t < best_t * 0.9:          # (a split has to pay for its workspace traffic)
            best_t, best_ks = t, ks
    flops = 2.0 * segs.rows * max(Cout, 32) * Cin * 9
    floor = 1.9 * (-(-Cin // 32) * 9) if Cout <= 96 else 21.0
    t_d = max(floor, flops / 95e12 * 1e6)
    return min(best_t, t_d), best_t, best_ks, t_d


def wino_preferred(segs: Segs, Cin: int, Cout: int, dil: int) -> bool:
    """wino_choice without split-K (callers that have no workspace: the training nodes)."""
    return wino_choice(segs, Cin, Cout, dil, allow_split=False)[0]


def strided_dgrad_classes(k: int, stride: int, pad: int):
    """Parity classes of the data gradient of a k x k conv with `stride`: for input rows h = stride*i + a only the taps
    r = r0 + stride*t contribute, and dY row = i + c - t.  Returns per class a: (r0, T taps, c) with r0 = (a + pad) % stride,
    T = number of taps, c = (a + pad - r0) // stride; T == 0: the class is identically zero."""
    out = []
    for a in range(stride):
        r0 = (a + pad) % stride
        T = len(range(r0, k, stride))
        out.append((r0, T, (a + pad - r0) // stride))
    return out


def conv_dgrad_strided(dy: Rows, w: torch.Tensor, scale: Optional[torch.Tensor], dx: Rows, N: int, H: int, W: int, k: int,
                       stride: int, pad: int, res: Optional[Rows] = None, res_mask: bool = False, precision: int = 0) -> bool:
    """dX of y = conv(x, w, stride >= 2, pad, dilation 1) on the MFMA conv kernel, exact FLOPs: one stride-1 launch per parity
    class (h % stride, w % stride) over dY with that class's taps, outputs interleaved straight into dX (fd_conv_params out_H /
    sc_*).  dx must be zero-initialised when some class has no tap (1x1 stride 2).  `scale` [Cout]: a folded frozen BatchNorm;
    res / res_mask as in conv_call (dX geometry).  Returns False (nothing launched) for a geometry it does not cover."""
    Cout, Cin = w.shape[0], w.shape[1]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    cls = strided_dgrad_classes(k, stride, pad)
    if Cout % 32 or Cin % 4 or any(T and (T - 1 - c) != 0 for _, T, c in cls):
        return False                      # (the class convs of k = 3 / pad 1 and k = 1 / pad 0 need no padding; others are not built)
    segs = Segs.make(N, [(Ho, Wo)])
    wd = w.detach()
    for a, (r0, Ta, _) in enumerate(cls):
        Ia = len(range(a, H, stride))
        for b, (q0, Tb, _) in enumerate(cls):
            Jb = len(range(b, W, stride))
            if Ta == 0 or Tb == 0 or Ia == 0 or Jb == 0:
                continue
            sub = wd[:, :, r0::stride, q0::stride].contiguous()                      # [Cout, Cin, Ta, Tb]
            k64 = bool(precision) and F16K64 and f16k64_ok(Cout, Cin)               # AMP: the class convs on FD_TILE_F16K64 where the widths allow
            out = torch.empty(Cin * Cout * Ta * Tb // (2 if k64 else 1), dtype=torch.float32, device=w.device)
            check(_lib.lib().fd_pack_conv_weight_f32(sub.data_ptr(), scale.data_ptr() if scale is not None else None, out.data_ptr(),
                                                     Cout, Cin, Ta, Tb, 1 | (16 if k64 else 4 if precision else 0), _stream()), "fd_pack_conv_weight_f32")
            conv_call(dy, segs, out, dx, Cin=Cout, Cout=Cin, k=Ta, kw=Tb, stride=1, pad=0, res=res, res_mask=res_mask,
                      out_hw=(Ia, Jb), scatter=(stride, stride, a, b, H, W), precision=precision, tile=_lib.F16K64_TILE if k64 else 0)()
    return True


def dgrad_weight(w: torch.Tensor) -> torch.Tensor:
    """Packed weights of the stride-1 data-gradient conv: w'[ci][co][r][q] = w[co][ci][K-1-r][K-1-q]."""
    return pack_conv_weight(w.detach().flip(2, 3).transpose(0, 1).contiguous())


def conv_out_segs(segs: Segs, k: int, stride: int, pad: int, dil: int) -> Segs:
    hw = [((h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1)
          for h, w in segs.level_hw()]
    return Segs.make(segs.batch, hw)


# ---------------------------------------------------------------------------------------------------- layer ops
def nchw3_to_nhwc4(x: torch.Tensor, y: torch.Tensor) -> None:
    _need_gpu(x, y)
    N, c, H, W = x.shape
    assert c == 3 and x.is_contiguous() and x.dtype == torch.float32
    check(_lib.lib().fd_nchw3_to_nhwc4(x.data_ptr(), y.data_ptr(), N, H, W, _stream()), "fd_nchw3_to_nhwc4")


def preprocess_u8(x: torch.Tensor, y: torch.Tensor, mean, std) -> None:
    """uint8 [N,H,W,3] (resized + zero padded) -> normalised fp32 [N*H*W, 4] (stem input)."""
    _need_gpu(x, y)
    N, H, W, c = x.shape
    assert c == 3 and x.dtype == torch.uint8 and x.is_contiguous()
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s_ = (C.c_float * 3)(*[float(v) for v in std])
    check(_lib.lib().fd_preprocess_u8_nhwc4(x.data_ptr(), y.data_ptr(), N, H, W, m, s_, _stream()), "fd_preprocess_u8_nhwc4")


def boxes_rescale_xywh_(boxes: torch.Tensor, scale: float) -> torch.Tensor:
    """In place: boxes /= scale; (x1, y1, x2, y2) -> (x, y, w, h)  (Test_coco.py:147-151)."""
    _need_gpu(boxes)
    assert boxes.is_contiguous() and boxes.shape[-1] == 4 and boxes.dtype == torch.float32
    check(_lib.lib().fd_boxes_rescale_xywh(boxes.data_ptr(), boxes.numel() // 4, float(scale), _stream()), "fd_boxes_rescale_xywh")
    return boxes


def nhwc_to_nchw(x: Rows, N: int, HW: int, out: torch.Tensor) -> None:
    check(_lib.lib().fd_nhwc_to_nchw(x.ptr, x.cs, x.co, out.data_ptr(), N, HW, x.C, _stream()), "fd_nhwc_to_nchw")


def maxpool(x: Rows, y: Rows, N: int, H: int, W: int, k: int, s: int, pad: int, add: Optional[Rows] = None) -> None:
    a = (add.ptr, add.cs, add.co) if add is not None else (None, 0, 0)
    check(_lib.lib().fd_maxpool_nhwc(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, a[0], a[1], a[2], N, H, W, x.C, k, s, pad,
                                     _stream()), "fd_maxpool_nhwc")


def upsample2x_add(x: Rows, lat: Rows, y: Rows, N: int, H: int, W: int) -> None:
    check(_lib.lib().fd_upsample2x_add_nhwc(x.ptr, x.cs, x.co, lat.ptr, lat.cs, lat.co, y.ptr, y.cs, y.co, N, H, W, x.C,
                                            _stream()), "fd_upsample2x_add_nhwc")


def dwconv3x3(x: Rows, w9c: torch.Tensor, y: Rows, segs: Segs, scale=None, shift=None, act: int = ACT_NONE) -> None:
    check(_lib.lib().fd_dwconv3x3_nhwc(x.ptr, x.cs, x.co, w9c.data_ptr(),
                                       scale.data_ptr() if scale is not None else None,
                                       shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, x.C, act,
                                       C.byref(segs), _stream()), "fd_dwconv3x3_nhwc")


def dwconv2d(x: Rows, wkc: torch.Tensor, y: Rows, N: int, H: int, W: int, K: int, stride: int, pad_top: int, pad_left: int,
             Ho: int, Wo: int, scale=None, shift=None, act: int = ACT_NONE) -> None:
    """Depthwise K x K, stride 1 / 2, asymmetric zero padding (EfficientNet MBConv), + scale / shift + act."""
    check(_lib.lib().fd_dwconv2d_nhwc(x.ptr, x.cs, x.co, wkc.data_ptr(), scale.data_ptr() if scale is not None else None,
                                      shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, N, H, W, x.C, K, stride,
                                      pad_top, pad_left, Ho, Wo, act, _stream()), "fd_dwconv2d_nhwc")


def mbconv_fused_ok(Cin: int, mid: int, K: int, stride: int) -> bool:
    """Shapes fd_mbconv_expand_dw_nhwc covers (the input patch of a tile and two expanded tiles must fit LDS: Cin <= 48)."""
    return K in (3, 5) and stride in (1, 2) and 8 <= Cin <= 48 and Cin % 8 == 0 and mid % 4 == 0


def pack_mbconv_expand_weight(w: torch.Tensor) -> torch.Tensor:
    """[mid, Cin, 1, 1] expand weights -> fd_mbconv_expand_dw_nhwc's fragment order [ceil(mid / 32)][Cin / 8][2][32][4]: element (cb, g, h, l, jj) =
    w[32 cb + l][h * Cin / 2 + 4 g + jj] (a lane's four consecutive K steps are one 16-byte LDS read)."""
    mid, cin = w.shape[0], w.shape[1]
    wp = torch.nn.functional.pad(w.detach().float().reshape(mid, cin), (0, 0, 0, (-mid) % 32))
    return wp.view(-1, 32, 2, cin // 8, 4).permute(0, 3, 2, 1, 4).contiguous()


def mbconv_pool_buffer(N: int, Ho: int, Wo: int, mid: int, K: int, stride: int, device) -> Tuple[torch.Tensor, int]:
    """(per-tile pooling partials buffer, tiles per image) of mbconv_expand_dw."""
    nb = _lib.lib().fd_mbconv_pool_bytes(N, Ho, Wo, mid, K, stride)
    if nb < 0:
        raise FdError("fd_mbconv_pool_bytes: bad arguments")
    return torch.empty(nb // 4, dtype=torch.float32, device=device), nb // 4 // (N * mid)


def mbconv_expand_dw(x: Rows, we_frag: torch.Tensor, sc0, sf0, wkc: torch.Tensor, sc1, sf1, y: Rows, pool: torch.Tensor, N: int, H: int, W: int, K: int, stride: int,
                     pad_top: int, pad_left: int, Ho: int, Wo: int) -> None:
    """MBConv expand 1x1 + BN + swish -> depthwise K x K + BN + swish in one launch (the expanded map stays on chip) + the SE pooling's per-tile partial sums."""
    check(_lib.lib().fd_mbconv_expand_dw_nhwc(x.ptr, x.cs, x.co, we_frag.data_ptr(), sc0.data_ptr(), sf0.data_ptr(), wkc.data_ptr(), sc1.data_ptr(), sf1.data_ptr(),
                                              y.ptr, y.cs, y.co, pool.data_ptr(), N, H, W, x.C, y.C, K, stride, pad_top, pad_left, Ho, Wo, _stream()),
          "fd_mbconv_expand_dw_nhwc")


def se_gate_from_pool(pool: torch.Tensor, T: int, w1, b1, w2, b2, N: int, HW: int, C_: int, Cr: int, ws: torch.Tensor) -> torch.Tensor:
    """The SE gates from mbconv_expand_dw's per-tile partial sums (no pass over the map); returns the [N, C] gate view into `ws` like se_gate."""
    check(_lib.lib().fd_se_gate_from_pool(pool.data_ptr(), T, w1.data_ptr(), b1.data_ptr() if b1 is not None else None, w2.data_ptr(),
                                          b2.data_ptr() if b2 is not None else None, N, HW, C_, Cr, ws.data_ptr(), _stream()), "fd_se_gate_from_pool")
    return se_gate_view(ws, N, HW, C_)


def dwconv_dilated(x: Rows, wkc: torch.Tensor, y: Rows, segs: Segs, K: int, dil: int, scale=None, shift=None, act: int = ACT_NONE) -> None:
    """Dilated depthwise K x K, stride 1, 'same' padding, over a pyramid (MNBlock.DilatedDepthWiseConv + folded BN); w [K*K][C]."""
    _need_gpu(wkc, scale, shift)
    check(_lib.lib().fd_dwconv_dilated_nhwc(x.ptr, x.cs, x.co, wkc.data_ptr(), scale.data_ptr() if scale is not None else None,
                                            shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, x.C, K, dil, act,
                                            C.byref(segs), _stream()), "fd_dwconv_dilated_nhwc")


def stem_conv3(x4: torch.Tensor, w: torch.Tensor, y: Rows, N: int, H: int, W: int, K: int, stride: int, pad_top: int,
               pad_left: int, Ho: int, Wo: int, scale=None, shift=None, act: int = ACT_NONE) -> None:
    """3-channel stem conv on the [N*H*W, 4] image layout (w from pack_stem3_weight)."""
    check(_lib.lib().fd_stem_conv_nhwc4(x4.data_ptr(), w.data_ptr(), scale.data_ptr() if scale is not None else None,
                                        shift.data_ptr() if shift is not None else None, y.ptr, y.cs, y.co, N, H, W, y.C, K, stride,
                                        pad_top, pad_left, Ho, Wo, act, _stream()), "fd_stem_conv_nhwc4")


def collate_u8(images: Sequence[torch.Tensor], H: int, W: int, mean, std, out: Optional[torch.Tensor] = None):
    """Resized uint8 [h_n, w_n, 3] CUDA images of different sizes -> one normalised [N*H*W, 4] fp32 batch
    (dataset/voc.py:128-132,141-156 on the device).  Returns (batch rows tensor, keep-alive tuple)."""
    N = len(images)
    dev = images[0].device
    for t in images:
        _need_gpu(t)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3 or not t.is_contiguous() or t.shape[0] > H or t.shape[1] > W:
            raise FdError("collate_u8: images must be contiguous CUDA uint8 [h, w, 3] with h <= H and w <= W")
    ptrs = torch.tensor([t.data_ptr() for t in images], dtype=torch.int64).to(dev)
    hw = torch.tensor([[t.shape[0], t.shape[1]] for t in images], dtype=torch.int32).to(dev)
    if out is None:
        out = torch.empty(N * H * W, 4, dtype=torch.float32, device=dev)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s_ = (C.c_float * 3)(*[float(v) for v in std])
    check(_lib.lib().fd_collate_u8_nhwc4(ptrs.data_ptr(), hw.data_ptr(), out.data_ptr(), N, H, W, m, s_, _stream()),
          "fd_collate_u8_nhwc4")
    return out, (ptrs, hw, tuple(images))


def dwconv3x3_wgrad(x: Rows, dy: Rows, segs: Segs, scale: Optional[torch.Tensor] = None, torch_layout: bool = False) -> torch.Tensor:
    """Weight gradient of the depthwise 3x3 conv (stride 1, pad 1) given dy: [9][C], or [C][1][3][3] with torch_layout."""
    dev = x.buf.device
    nb = _lib.lib().fd_dwconv3x3_wgrad_workspace_bytes(C.byref(segs), x.C)
    if nb < 0:
        raise FdError("fd_dwconv3x3_wgrad_workspace_bytes: bad arguments")
    ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
    dw = torch.empty((x.C, 1, 3, 3) if torch_layout else (9, x.C), dtype=torch.float32, device=dev)
    check(_lib.lib().fd_dwconv3x3_bwd_weight_nhwc(x.ptr, x.cs, x.co, dy.ptr, dy.cs, dy.co, dw.data_ptr(), x.C,
                                                  scale.data_ptr() if scale is not None else None, 1 if torch_layout else 0,
                                                  C.byref(segs), ws.data_ptr(), _stream()), "fd_dwconv3x3_bwd_weight_nhwc")
    return dw


def groupnorm_workspace(segs: Segs, G: int, device) -> torch.Tensor:
    n = _lib.lib().fd_groupnorm_workspace_bytes(C.byref(segs), G)
    if n < 0:
        raise FdError("fd_groupnorm_workspace_bytes: bad arguments")
    return torch.empty(n // 8, dtype=torch.float64, device=device)


def groupnorm_act(x: Rows, gamma: torch.Tensor, beta: torch.Tensor, y: Rows, segs: Segs, G: int, act: int,
                  ws: torch.Tensor, eps: float = 1e-5) -> None:
    check(_lib.lib().fd_groupnorm_act_nhwc(x.ptr, x.cs, x.co, gamma.data_ptr(), beta.data_ptr(), y.ptr, y.cs, y.co, x.C,
                                           G, eps, act, C.byref(segs), ws.data_ptr(), _stream()), "fd_groupnorm_act_nhwc")


def groupnorm_from_rowstats(rowstats: torch.Tensor, Cc: int, G: int, eps: float, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                            segs: Segs, ws: torch.Tensor, coef: Optional[torch.Tensor] = None) -> None:
    """Row-group sums left by a producer (conv_call(gn_stats=...), dwconv3x3_gn) -> GroupNorm statistics per (level, image, group) in `ws`
    (groupnorm_workspace layout) and, with `coef` [levels * batch, 2, C], the affine (rstd * gamma, beta - mean * rstd * gamma) a consumer
    applies in its loader."""
    _need_gpu(rowstats, ws, coef, gamma, beta)
    check(_lib.lib().fd_groupnorm_from_rowstats(rowstats.data_ptr(), Cc, G, eps, gamma.data_ptr() if gamma is not None else None,
                                                beta.data_ptr() if beta is not None else None, C.byref(segs), ws.data_ptr(),
                                                coef.data_ptr() if coef is not None else None, _stream()), "fd_groupnorm_from_rowstats")


def groupnorm_apply(x: Rows, gamma: torch.Tensor, beta: torch.Tensor, y: Rows, segs: Segs, G: int, act: int, ws: torch.Tensor, eps: float = 1e-5) -> None:
    """y = act(GroupNorm(x)) from statistics already in `ws` (groupnorm_from_rowstats / an earlier groupnorm_act): one pass over the map."""
    check(_lib.lib().fd_groupnorm_apply_nhwc(x.ptr, x.cs, x.co, gamma.data_ptr(), beta.data_ptr(), y.ptr, y.cs, y.co, x.C, G, eps, act,
                                             C.byref(segs), ws.data_ptr(), _stream()), "fd_groupnorm_apply_nhwc")


def groupnorm_stats(x: Rows, gamma: torch.Tensor, beta: torch.Tensor, segs: Segs, G: int, ws: torch.Tensor, eps: float = 1e-5,
                    coef: Optional[torch.Tensor] = None) -> None:
    """GroupNorm statistics of x per (level, image, group) into `ws` (groupnorm_workspace layout) and, with `coef` [levels * batch, 2, C], the
    affine (rstd * gamma, beta - mean * rstd * gamma) for consumers that normalise themselves (conv_call(gate_b=...), coef_apply)."""
    _need_gpu(ws, coef, gamma, beta)
    check(_lib.lib().fd_groupnorm_stats_nhwc(x.ptr, x.cs, x.co, x.C, G, eps, gamma.data_ptr(), beta.data_ptr(), C.byref(segs), ws.data_ptr(),
                                             coef.data_ptr() if coef is not None else None, _stream()), "fd_groupnorm_stats_nhwc")


def coef_apply(x: Rows, coef_a: torch.Tensor, coef_b: torch.Tensor, y: Rows, segs: Segs, act: int) -> None:
    """y = act(x * coef_a[img] + coef_b[img]) over x's channel view: coef_a / coef_b are [levels * batch, x.C] views (unit channel stride, same row
    stride) into groupnorm_from_rowstats' coef -- the normalise pass of a channel SLICE of a jointly reduced map."""
    _need_gpu(coef_a, coef_b)
    if (coef_a.dim() != 2 or coef_a.shape != (segs.batch * segs.nseg, x.C) or coef_b.shape != coef_a.shape or coef_a.stride() != coef_b.stride()
            or coef_a.stride(1) != 1 or coef_a.dtype != torch.float32 or coef_b.dtype != torch.float32):
        raise FdError("coef_apply: coef_a / coef_b must be [levels * batch, C] fp32 views with unit channel stride and equal row strides")
    check(_lib.lib().fd_coef_apply_nhwc(x.ptr, x.cs, x.co, coef_a.data_ptr(), coef_b.data_ptr(), coef_a.stride(0), y.ptr, y.cs, y.co, x.C, act,
                                        C.byref(segs), _stream()), "fd_coef_apply_nhwc")


def dwconv3x3_gn(x: Rows, w9c: torch.Tensor, y: Rows, segs: Segs, in_coef: Optional[torch.Tensor], in_act: int,
                 gn_stats: Optional[torch.Tensor], gn_groups: int) -> None:
    """Depthwise 3x3 (stride 1, pad 1, no bias) that reads its input as in_act(x * a + b) (in_coef [levels * batch, 2, C] from
    groupnorm_from_rowstats; the zero padding applies to the normalised map) and leaves the row-group sums of its output in gn_stats."""
    _need_gpu(w9c, in_coef, gn_stats)
    check(_lib.lib().fd_dwconv3x3_gn_nhwc(x.ptr, x.cs, x.co, w9c.data_ptr(), in_coef.data_ptr() if in_coef is not None else None, in_act,
                                          y.ptr, y.cs, y.co, x.C, gn_stats.data_ptr() if gn_stats is not None else None, gn_groups,
                                          C.byref(segs), _stream()), "fd_dwconv3x3_gn_nhwc")


def groupnorm_act_bwd(x: Rows, dy: Rows, gamma: torch.Tensor, beta: torch.Tensor, dx: Rows, segs: Segs, G: int, act: int,
                      fwd_ws: torch.Tensor, eps: float = 1e-5):
    """Backward of groupnorm_act (fwd_ws = the forward call's workspace, untouched since).  Returns (dgamma, dbeta)."""
    dev = x.buf.device
    n = _lib.lib().fd_groupnorm_bwd_workspace_bytes(C.byref(segs), x.C)
    if n < 0:
        raise FdError("fd_groupnorm_bwd_workspace_bytes: bad arguments")
    ws = torch.empty(n // 8, dtype=torch.float64, device=dev)
    dgamma = torch.empty(x.C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(x.C, dtype=torch.float32, device=dev)
    check(_lib.lib().fd_groupnorm_act_bwd_nhwc(x.ptr, x.cs, x.co, dy.ptr, dy.cs, dy.co, gamma.data_ptr(), beta.data_ptr(),
                                               dx.ptr, dx.cs, dx.co, dgamma.data_ptr(), dbeta.data_ptr(), x.C, G, eps, act,
                                               C.byref(segs), fwd_ws.data_ptr(), ws.data_ptr(), _stream()),
          "fd_groupnorm_act_bwd_nhwc")
    return dgamma, dbeta


def se_workspace(N: int, HW: int, C_: int, device) -> torch.Tensor:
    n = _lib.lib().fd_se_workspace_bytes(N, HW, C_)
    return torch.empty((n + 7) // 8, dtype=torch.float64, device=device)


def se_scale(x: Rows, w1, b1, w2, b2, y: Rows, N: int, HW: int, Cr: int, ws: torch.Tensor) -> None:
    check(_lib.lib().fd_se_scale_nhwc(x.ptr, x.cs, x.co, w1.data_ptr(), b1.data_ptr() if b1 is not None else None,
                                      w2.data_ptr(), b2.data_ptr() if b2 is not None else None, y.ptr, y.cs, y.co, N, HW,
                                      x.C, Cr, ws.data_ptr(), _stream()), "fd_se_scale_nhwc")


def se_gate(x: Rows, w1, b1, w2, b2, N: int, HW: int, Cr: int, ws: torch.Tensor) -> torch.Tensor:
    """Squeeze-excitation gates only: sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2) as an [N, C] fp32 view into `ws` (se_workspace), for a
    consumer that applies them itself (conv_call(..., gate=...): the MBConv project conv multiplies them in on its way to LDS)."""
    check(_lib.lib().fd_se_scale_nhwc(x.ptr, x.cs, x.co, w1.data_ptr(), b1.data_ptr() if b1 is not None else None,
                                      w2.data_ptr(), b2.data_ptr() if b2 is not None else None, None, 0, 0, N, HW,
                                      x.C, Cr, ws.data_ptr(), _stream()), "fd_se_scale_nhwc")
    return se_gate_view(ws, N, HW, x.C)


def se_gate_view(ws: torch.Tensor, N: int, HW: int, C_: int) -> torch.Tensor:
    """The [N, C] gate array inside an se_workspace buffer (the last N * C floats of its fd_se_workspace_bytes)."""
    nbytes = _lib.lib().fd_se_workspace_bytes(N, HW, C_)
    return ws.view(torch.float32)[nbytes // 4 - N * C_: nbytes // 4].view(N, C_)


def se_scale_bwd(x: Rows, dy: Rows, w1, b1, w2, b2, dx: Rows, N: int, HW: int, Cr: int, fwd_ws: torch.Tensor):
    """Backward of se_scale: writes dx, returns (dw1 [Cr,C], db1 [Cr], dw2 [C,Cr], db2 [C])."""
    dev = x.buf.device
    Cc = x.C
    n = _lib.lib().fd_se_bwd_workspace_bytes(N, HW, Cc)
    ws = torch.empty((n + 7) // 8, dtype=torch.float64, device=dev)
    dw1 = torch.empty(Cr, Cc, dtype=torch.float32, device=dev)
    db1 = torch.empty(Cr, dtype=torch.float32, device=dev)
    dw2 = torch.empty(Cc, Cr, dtype=torch.float32, device=dev)
    db2 = torch.empty(Cc, dtype=torch.float32, device=dev)
    check(_lib.lib().fd_se_scale_bwd_nhwc(x.ptr, x.cs, x.co, dy.ptr, dy.cs, dy.co, w1.data_ptr(), b1.data_ptr() if b1 is not None else None,
                                          w2.data_ptr(), b2.data_ptr() if b2 is not None else None, dx.ptr, dx.cs, dx.co, dw1.data_ptr(),
                                          db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), N, HW, Cc, Cr, fwd_ws.data_ptr(), ws.data_ptr(),
                                          _stream()), "fd_se_scale_bwd_nhwc")
    return dw1, db1, dw2, db2


def act(x: Rows, y: Rows, act_id: int, param: float = 0.0) -> None:
    check(_lib.lib().fd_act_nhwc(x.ptr, x.cs, x.co, y.ptr, y.cs, y.co, x.rows, x.C, act_id, float(param), _stream()), "fd_act_nhwc")


def act_bwd(x: Rows, dy: Rows, dx: Rows, act_id: int, param: float = 0.0) -> None:
    if x.f16 or dy.f16 or dx.f16:          # f16 maps (AMP): all three, or none
        if not (x.f16 and dy.f16 and dx.f16):
            raise FdError("act_bwd: x, dy and dx must all be f16 maps or all fp32")
        check(_lib.lib().fd_act_bwd_nhwc_h(x.ptr, x.cs, x.co, dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, x.rows, x.C, act_id, float(param),
                                           _stream()), "fd_act_bwd_nhwc_h")
        return
    check(_lib.lib().fd_act_bwd_nhwc(x.ptr, x.cs, x.co, dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, x.rows, x.C, act_id, float(param),
                                     _stream()), "fd_act_bwd_nhwc")


def maxpool_bwd(x: Rows, dy: Rows, dx: Rows, N: int, H: int, W: int, k: int, s: int, pad: int) -> None:
    check(_lib.lib().fd_maxpool_bwd_nhwc(x.ptr, x.cs, x.co, dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, N, H, W, x.C, k, s, pad,
                                         _stream()), "fd_maxpool_bwd_nhwc")


def upsample2x_bwd(dy: Rows, dx: Rows, N: int, H: int, W: int) -> None:
    """dx [N,H,W] = 2x2 block sums of dy [N,2H,2W]."""
    check(_lib.lib().fd_upsample2x_bwd_nhwc(dy.ptr, dy.cs, dy.co, dx.ptr, dx.cs, dx.co, N, H, W, dx.C, _stream()), "fd_upsample2x_bwd_nhwc")


def batchnorm_update_running(gn_ws: torch.Tensor, rows: int, Cc: int, momentum: float, eps: float, running_mean: torch.Tensor,
                             running_var: torch.Tensor) -> None:
    _need_gpu(gn_ws, running_mean, running_var)
    check(_lib.lib().fd_batchnorm_update_running(gn_ws.data_ptr(), rows, Cc, float(momentum), float(eps), running_mean.data_ptr(),
                                                 running_var.data_ptr(), _stream()), "fd_batchnorm_update_running")


def batchnorm_update_running_dev(gn_ws: torch.Tensor, count: torch.Tensor, Cc: int, momentum: float, eps: float, running_mean: torch.Tensor,
                                 running_var: torch.Tensor) -> None:
    """As batchnorm_update_running with the row count in device memory (`count`: one float64 element, e.g. the tail of SyncBatchNorm's
    all-reduced buffer): no host round trip."""
    _need_gpu(gn_ws, count, running_mean, running_var)
    assert count.dtype == torch.float64 and count.numel() == 1
    check(_lib.lib().fd_batchnorm_update_running_dev(gn_ws.data_ptr(), count.data_ptr(), Cc, float(momentum), float(eps), running_mean.data_ptr(),
                                                     running_var.data_ptr(), _stream()), "fd_batchnorm_update_running_dev")


# ---------------------------------------------------------------------------------------------------- post-process
def fcos_decode(cls: Rows, cnt: Rows, reg: Rows, segs: Segs, strides: Sequence[int]):
    """-> scores [N,L] f32, classes [N,L] i32, boxes [N,L,4] f32 (levels concatenated per image)."""
    N = segs.batch
    L = sum(h * w for h, w in segs.level_hw())
    dev = cls.buf.device
    scores = torch.empty(N, L, dtype=torch.float32, device=dev)
    classes = torch.empty(N, L, dtype=torch.int32, device=dev)
    boxes = torch.empty(N, L, 4, dtype=torch.float32, device=dev)
    st = (C.c_int32 * len(strides))(*strides)
    check(_lib.lib().fd_fcos_decode(cls.ptr, cls.cs, cls.co, cnt.ptr, cnt.cs, cnt.co, reg.ptr, reg.cs, reg.co, cls.C,
                                    C.byref(segs), st, scores.data_ptr(), classes.data_ptr(), boxes.data_ptr(), _stream()),
          "fd_fcos_decode")
    return scores, classes, boxes


def fcos_topk(scores: torch.Tensor, classes: torch.Tensor, boxes: torch.Tensor, K: int, want_idx: bool = False):
    _need_gpu(scores, classes, boxes)
    N, L = scores.shape
    dev = scores.device
    ts = torch.empty(N, K, dtype=torch.float32, device=dev)
    tc = torch.empty(N, K, dtype=torch.int64, device=dev)
    tb = torch.empty(N, K, 4, dtype=torch.float32, device=dev)
    ti = torch.empty(N, K, dtype=torch.int32, device=dev) if want_idx else None
    ws = None
    if K > 1024:        # (FCOSHead(max_detection_box > 1024): the candidate list is sorted in a global scratch)
        nb = _lib.lib().fd_topk_workspace_bytes(N, L, K)
        if nb < 0:
            raise FdError(f"fd_topk_workspace_bytes: bad arguments N={N} L={L} K={K}")
        ws = torch.empty(nb // 8, dtype=torch.int64, device=dev)
    check(_lib.lib().fd_fcos_topk(scores.data_ptr(), classes.data_ptr(), boxes.data_ptr(), N, L, K, ts.data_ptr(),
                                  tc.data_ptr(), tb.data_ptr(), ti.data_ptr() if want_idx else None,
                                  ws.data_ptr() if ws is not None else None, _stream()),
          "fd_fcos_topk")
    return (ts, tc, tb, ti) if want_idx else (ts, tc, tb)


_NMS_WS: dict = {}


def _nms_workspace(N: int, K: int, dev) -> torch.Tensor:
    """Per-(device, stream) cached scratch for the suppression bitmask (stream order makes reuse safe)."""
    key = (str(dev), _stream())
    need = _lib.lib().fd_nms_workspace_bytes(N, K)
    if need < 0:
        raise FdError(f"fd_nms_workspace_bytes: unsupported N={N} K={K} (K <= 262144)")
    ws = _NMS_WS.get(key)
    if ws is None or ws.numel() * 8 < need:
        ws = torch.empty(need // 8, dtype=torch.int64, device=dev)
        _NMS_WS[key] = ws
    return ws


def batched_nms(scores: torch.Tensor, classes: torch.Tensor, boxes: torch.Tensor, score_thr: float, iou_thr: float):
    """Padded outputs [N,K] + keep_idx [N,K] (int32, -1 padded) + counts [N] (int32)."""
    _need_gpu(scores, classes, boxes)
    N, K = scores.shape
    assert classes.dtype == torch.int64 and scores.is_contiguous() and classes.is_contiguous() and boxes.is_contiguous()
    dev = scores.device
    os_ = torch.empty(N, K, dtype=torch.float32, device=dev)
    oc = torch.empty(N, K, dtype=torch.int64, device=dev)
    ob = torch.empty(N, K, 4, dtype=torch.float32, device=dev)
    keep = torch.empty(N, K, dtype=torch.int32, device=dev)
    counts = torch.empty(N, dtype=torch.int32, device=dev)
    ws = _nms_workspace(N, K, dev)
    check(_lib.lib().fd_batched_nms(scores.data_ptr(), classes.data_ptr(), boxes.data_ptr(), N, K, score_thr, iou_thr,
                                    os_.data_ptr(), oc.data_ptr(), ob.data_ptr(), keep.data_ptr(), counts.data_ptr(),
                                    ws.data_ptr(), _stream()), "fd_batched_nms")
    return os_, oc, ob, keep, counts


def box_nms_plus1(boxes: torch.Tensor, scores: torch.Tensor, thr: float = 0.5, mode: str = "union",
                  n_valid: Optional[torch.Tensor] = None):
    _need_gpu(boxes, scores)
    N, K = scores.shape
    keep = torch.empty(N, K, dtype=torch.int32, device=scores.device)
    counts = torch.empty(N, dtype=torch.int32, device=scores.device)
    check(_lib.lib().fd_box_nms_plus1(boxes.data_ptr(), scores.data_ptr(), n_valid.data_ptr() if n_valid is not None else None,
                                      N, K, thr, {"union": 0, "min": 1}[mode], keep.data_ptr(), counts.data_ptr(), _stream()),
          "fd_box_nms_plus1")
    return keep, counts


def pairwise_iou(a: torch.Tensor, b: torch.Tensor, plus_one: bool) -> torch.Tensor:
    _need_gpu(a, b)
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    check(_lib.lib().fd_pairwise_iou(a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], int(plus_one), out.data_ptr(),
                                     _stream()), "fd_pairwise_iou")
    return out


def clip_boxes_(boxes: torch.Tensor, img_h: int, img_w: int) -> torch.Tensor:
    _need_gpu(boxes)
    assert boxes.is_contiguous() and boxes.shape[-1] == 4 and boxes.dtype == torch.float32
    check(_lib.lib().fd_clip_boxes(boxes.data_ptr(), boxes.numel() // 4, img_h, img_w, _stream()), "fd_clip_boxes")
    return boxes


def pack_detections(scores: torch.Tensor, classes: torch.Tensor, boxes: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """Padded detections of B images -> one [B, K+1, 6] fp32 message (row 0 = count; then x1, y1, x2, y2, score, class)."""
    _need_gpu(scores, classes, boxes, counts)
    B, K = scores.shape
    assert classes.dtype == torch.int64 and counts.dtype == torch.int32
    rec = torch.empty(B, K + 1, 6, dtype=torch.float32, device=scores.device)
    check(_lib.lib().fd_pack_detections(scores.contiguous().data_ptr(), classes.contiguous().data_ptr(), boxes.contiguous().data_ptr(),
                                        counts.contiguous().data_ptr(), B, K, rec.data_ptr(), _stream()), "fd_pack_detections")
    return rec


def unpack_detections(rec: torch.Tensor):
    """Inverse of pack_detections on [B, K+1, 6] records -> (scores [B,K], classes [B,K] int64, boxes [B,K,4], counts [B] int32)."""
    _need_gpu(rec)
    B, K1, six = rec.shape
    assert six == 6 and rec.is_contiguous() and rec.dtype == torch.float32
    K, dev = K1 - 1, rec.device
    s = torch.empty(B, K, dtype=torch.float32, device=dev)
    c = torch.empty(B, K, dtype=torch.int64, device=dev)
    b = torch.empty(B, K, 4, dtype=torch.float32, device=dev)
    n = torch.empty(B, dtype=torch.int32, device=dev)
    check(_lib.lib().fd_unpack_detections(rec.data_ptr(), B, K, s.data_ptr(), c.data_ptr(), b.data_ptr(), n.data_ptr(), _stream()),
          "fd_unpack_detections")
    return s, c, b, n


# ---------------------------------------------------------------------------------------------------- loss
def ltrb_loss_fwd(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor, mode: int):
    _need_gpu(pred, target, mask)
    B, L, _ = pred.shape
    loss = torch.empty(B, dtype=torch.float32, device=pred.device)
    npos = torch.empty(B, dtype=torch.int32, device=pred.device)
    check(_lib.lib().fd_ltrb_iou_loss_fwd(pred.data_ptr(), target.data_ptr(), mask.data_ptr(), B, L, mode, loss.data_ptr(),
                                          npos.data_ptr(), _stream()), "fd_ltrb_iou_loss_fwd")
    return loss, npos


def ltrb_loss_bwd(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor, gscale: torch.Tensor, mode: int):
    B, L, _ = pred.shape
    grad = torch.empty_like(pred)
    check(_lib.lib().fd_ltrb_iou_loss_bwd(pred.data_ptr(), target.data_ptr(), mask.data_ptr(), gscale.data_ptr(), B, L, mode,
                                          grad.data_ptr(), _stream()), "fd_ltrb_iou_loss_bwd")
    return grad


def focal_loss_fwd(logits: torch.Tensor, labels: torch.Tensor, alpha: float = 0.25, gamma: float = 2.0) -> torch.Tensor:
    """logits [B,L,C] f32, labels [B,L] int64 -> per-image summed focal loss [B]."""
    _need_gpu(logits, labels)
    B, L, Cn = logits.shape
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    ws = torch.empty(_lib.lib().fd_focal_workspace_bytes(B) // 8, dtype=torch.float64, device=logits.device)
    check(_lib.lib().fd_focal_loss_fwd(logits.data_ptr(), labels.data_ptr(), B, L, Cn, alpha, gamma, loss.data_ptr(),
                                       ws.data_ptr(), _stream()), "fd_focal_loss_fwd")
    return loss


def focal_loss_bwd(logits, labels, gscale, alpha: float = 0.25, gamma: float = 2.0) -> torch.Tensor:
    B, L, Cn = logits.shape
    grad = torch.empty_like(logits)
    check(_lib.lib().fd_focal_loss_bwd(logits.data_ptr(), labels.data_ptr(), gscale.data_ptr(), B, L, Cn, alpha, gamma,
                                       grad.data_ptr(), _stream()), "fd_focal_loss_bwd")
    return grad


def bce_loss_fwd(x: torch.Tensor, target: torch.Tensor, mask: torch.Tensor):
    _need_gpu(x, target, mask)
    B, L = x.shape
    loss = torch.empty(B, dtype=torch.float32, device=x.device)
    npos = torch.empty(B, dtype=torch.int32, device=x.device)
    check(_lib.lib().fd_bce_logits_loss_fwd(x.data_ptr(), target.data_ptr(), mask.data_ptr(), B, L, loss.data_ptr(),
                                            npos.data_ptr(), _stream()), "fd_bce_logits_loss_fwd")
    return loss, npos


def bce_loss_bwd(x, target, mask, gscale) -> torch.Tensor:
    B, L = x.shape
    grad = torch.empty_like(x)
    check(_lib.lib().fd_bce_logits_loss_bwd(x.data_ptr(), target.data_ptr(), mask.data_ptr(), gscale.data_ptr(), B, L,
                                            grad.data_ptr(), _stream()), "fd_bce_logits_loss_bwd")
    return grad


def fcos_gen_targets(gt_boxes: torch.Tensor, labels: torch.Tensor, level_hw, strides, ranges, radius: float = 1.5):
    """gt_boxes [B,M,4] f32, labels [B,M] int64 -> (cls [B,L,1] int64, cnt [B,L,1] f32, reg [B,L,4] f32)."""
    _need_gpu(gt_boxes, labels)
    B, M = labels.shape
    segs = Segs.make(B, list(level_hw))
    L = sum(h * w for h, w in level_hw)
    dev = gt_boxes.device
    cls = torch.empty(B, L, 1, dtype=torch.int64, device=dev)
    cnt = torch.empty(B, L, 1, dtype=torch.float32, device=dev)
    reg = torch.empty(B, L, 4, dtype=torch.float32, device=dev)
    n = len(level_hw)
    st = (C.c_int32 * n)(*[int(s) for s in strides])
    lo = (C.c_int32 * n)(*[int(r[0]) for r in ranges])
    hi = (C.c_int32 * n)(*[int(r[1]) for r in ranges])
    gb = gt_boxes.contiguous().float()
    lb = labels.contiguous().to(torch.int64)
    check(_lib.lib().fd_fcos_gen_targets(gb.data_ptr(), lb.data_ptr(), M, C.byref(segs), st, lo, hi, radius, cls.data_ptr(),
                                         cnt.data_ptr(), reg.data_ptr(), _stream()), "fd_fcos_gen_targets")
    return cls, cnt, reg
