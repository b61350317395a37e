"""Inference engine: compiles the reference-shaped module tree (parameter containers) into a flat list of HIP
launches over pre-allocated NHWC buffers.

 * frozen BatchNorm (HISFcos.py:57-68) is folded into the conv epilogue (scale, shift);
 * torch.cat (HISFcos.py:107,111) disappears: producers write channel slices of the consumer's input buffer;
 * a HisBlock's conv1 (+bn1, SiLU) and conv2 (same input) run as one 2*half-wide conv (activation on the upper half only);
 * the 5 pyramid levels live in ONE rows buffer, so the shared-weight head (HISFcos.py:215-229) is one grouped
   launch per layer instead of five;
 * cls_conv / reg_conv (same input) run as one 512-wide conv, their GroupNorm(32,256) pair as GroupNorm(64,512);
   cnt_logits / reg_pred (same input) as one 5-wide conv with exp(scale_i * x) on channels 1..4.
A plan is specific to (batch, H, W) and cached by the owning module.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from . import _lib
from ._lib import ACT_EXP, ACT_NONE, ACT_RELU, ACT_SILU, FdError, Segs
from .ops import Rows


import os as _os

# Conv arithmetic: 'f32' = exact fp32 MFMA (default, parity baseline); 'f16x3' = opt-in split-f16 products for every
# conv except the 7x7 stem (K = 147 is too short to matter).  See include/fcosdet.h FD_PREC_*.
CONV_PRECISION = _os.environ.get("FD_CONV_PRECISION", "f32")
AUTOTUNE = True   # per-conv block-tile lookup / timing at plan-build time (see ops.autotune_conv, FD_AUTOTUNE)
# FD_WINOGRAD: "1" (default) = 3x3 stride-1 'same' convs (dilation 1 / 2, Cin % 8 == 0) of an exact-fp32 plan run on the Winograd
# F(2x2, 3x3) kernel (fd_conv_wino.hip: 2.25x fewer MFMAs, still fp32 arithmetic) where the map is large enough for it to win
# (ops.wino_preferred: it has no split-K); "force" = wherever it applies; "0" = every conv on the direct implicit-GEMM kernel
WINOGRAD = _os.environ.get("FD_WINOGRAD", "1") != "0"
SE_GATE_IN_PROJECT = _os.environ.get("FD_SE_GATE_FUSED", "1") != "0"     # MBConv: SE gate applied by the project conv's loader ("0": a scaling pass)
GN_FUSED_TOWER = _os.environ.get("FD_GN_FUSED_TOWER", "0") == "1"   # "1": the tower's statistics from its Winograd epilogue too (measured neutral, costs the tower launch 5 %)
# "1": a head-tower F(4x4) launch whose grid is no multiple of the CU count runs as whole rounds of workgroups + a tail launch ("head.tower3x3.tail", after the mark):
# TwoLanePipeline releases the other lane when the whole rounds are done, so the tail round (60 % of the chip idle at 16 x 640 x 640) has company
TOWER_TAIL_SPLIT = _os.environ.get("FD_TOWER_TAIL_SPLIT", "1") != "0"
# "1": an MBConv block's expand conv and depthwise conv run as ONE launch where the shapes allow (fd_mbconv_expand_dw_nhwc: Cin <= 48, k in {3, 5}); "0": separate launches
MBCONV_FUSED = _os.environ.get("FD_MBCONV_FUSED", "1") != "0"
FPN_UP_FUSED = _os.environ.get("FD_FPN_UP_FUSED", "1") != "0"   # "0": the top-down path's x2 upsample + add as its own pass over the finer map (else in the lateral conv's epilogue)
TOWER_GN_SPLIT = _os.environ.get("FD_TOWER_GN_SPLIT", "1") != "0"   # "0": the tower's GroupNorm normalises both halves in its own pass (else the box half in the narrow predictor's loader)
GN_FUSED = _os.environ.get("FD_GN_FUSED", "1") != "0"       # "0": HISFCOSHead's GroupNorms as three-pass launches (statistics / finalise / normalise)
WAVE_TILE = _os.environ.get("FD_WAVE_TILE", "1") != "0"       # "0": the 1x1 layers never see FD_TILE_WAVE64 (wave-autonomous tiles, fd_conv_wave.hip)
B2B_MIN_ROWS = int(_os.environ.get("FD_B2B_MIN_ROWS", str(64 * 1024)))   # fewer rows than 1 024 waves of 64: the seam stays two workgroup-tiled launches
DUAL_DS = _os.environ.get("FD_DUAL_DS", "1") != "0"      # "0": a block's downsample conv as its own launch, its output read back as conv3's residual
# FD_B2B: the trunk layers (digits) whose conv3 -> next-block conv1 seams run as ONE back-to-back launch (fd_conv1x1_b2b_f32: the 4 * planes wide map is
# written once and never read back); "" = none
B2B_LAYERS = _os.environ.get("FD_B2B", "1")
STEM_NCHW = _os.environ.get("FD_STEM_NCHW", "1") != "0"         # "0": an fp32 NCHW input batch is first copied to the [N][H][W][4] layout (fd_nchw3_to_nhwc4) instead of being read by the stem's loader
STEM_POOL = _os.environ.get("FD_STEM_POOL", "1") != "0"         # "0": the stem's 3x3 s2 max-pool as its own launch (the 64-channel stride-2 map written and read back)
STEM_KERNEL = _os.environ.get("FD_STEM_KERNEL", "1") != "0"     # "0": the ResNet stem through the generic conv kernel's FD_CONV_STEM mode


class PRows(Rows):
    """Rows carved from a pooled flat buffer."""
    __slots__ = ("_flat",)


class Pool:
    """Plan-time buffer pool: activations whose lifetime ended are reused by later layers (fewer live bytes in
    HBM / Infinity Cache).  Safe because the plan runs in stream order: a buffer is only handed out again after
    every step that reads it has been planned."""

    def __init__(self, device):
        self.device = device
        self.free: List[torch.Tensor] = []
        self.total = 0

    def get(self, rows: int, C: int) -> PRows:
        need = rows * C
        best = None
        for i, t in enumerate(self.free):
            if t.numel() >= need and (best is None or t.numel() < self.free[best].numel()):
                best = i
        if best is None:
            flat = torch.empty(need, dtype=torch.float32, device=self.device)
            self.total += need * 4
        else:
            flat = self.free.pop(best)
        r = PRows(flat[:need].view(rows, C))
        r._flat = flat
        return r

    def put(self, r: PRows) -> None:
        self.free.append(r._flat)


class Plan:
    def __init__(self, device, precision: Optional[str] = None, pair_tuned: bool = False):
        self.device = device
        self.steps: List[Callable[[], None]] = []
        self.names: List[str] = []
        self.pool = Pool(device)
        self.keep: List[object] = []       # tensors that must outlive the plan (packed weights, workspaces)
        self.flops = 0                     # algorithmic conv FLOPs (2*MACs) of one run
        self.step_flops: Dict[int, int] = {}
        self.step_info: Dict[int, dict] = {}   # conv steps: kernel size / stride / which kernel family ran it (bench.py's family rooflines)
        self.marks: Dict[str, Tuple[int, int]] = {}
        self.autotune = AUTOTUNE        # time block-tile candidates per conv at plan-build time
        self.precision = precision or CONV_PRECISION
        if self.precision not in ("f32", "f16x3", "mixed"):
            raise FdError(f"unknown conv precision '{self.precision}' (f32 | f16x3 | mixed)")
        self.tiles: Dict[str, int] = {}
        # 'mixed' (opt-in): the 3x3 stride-1 layers stay on the exact-fp32 Winograd kernel, the GEMM-addressed 1x1 layers -- 48 % of the step,
        # short K, epilogue-heavy -- take the split-f16 products (three f16 MFMAs per fp32 product = 3/16 of the matrix time, same 1e-4 bar)
        self.winograd = WINOGRAD and self.precision in ("f32", "mixed")
        self.pair_tuned = pair_tuned       # block tiles picked for throughput beside a second batch (pipeline.TwoLanePipeline)

    def add(self, name: str, fn: Callable[[], None]) -> None:
        self.steps.append(fn)
        self.names.append(name)

    graph = None        # torch.cuda.CUDAGraph of one run() (capture_graph): the ~190 launches of a forward replayed as ONE hipGraphLaunch

    def capture_graph(self, warmup: int = 2) -> None:
        """Capture one run() of the plan into a HIP graph (torch.cuda.CUDAGraph); later run() calls without `events` replay it.
        For the launch-bound shapes (the reference's own: batch 1, 512 x 512, test.py:202-223 -- ~190 launches of ~10 us each).  The
        steps read plan-owned buffers at fixed addresses; the caller's image is staged into `static_in` (same shape / dtype as the
        tensor in image_ref at capture time) by run(), so any input tensor works afterwards.  Inputs whose ADDRESSES the steps bake in
        per call (the mixed-aspect `collate` list) cannot be captured."""
        if self.graph is not None:
            return
        if getattr(self, "input_mode", None) == "collate":
            raise FdError("capture_graph: a 'collate' plan rebuilds its pointer table on every run and cannot be captured")
        src = self.image_ref[0]
        self.static_in = torch.empty_like(src).copy_(src)
        self.image_ref[0] = self.static_in
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                for st in self.steps:
                    st()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for st in self.steps:
                st()
        self.graph = g

    def run(self, events=None) -> None:
        """Launch every step on the current stream.  `events` maps a mark name to an (start, end) pair of
        torch.cuda.Event recorded around that mark's launches (used by bench.py for per-kernel timing)."""
        if self.graph is not None and not events:
            src = self.image_ref[0]
            if src is not self.static_in:
                self.static_in.copy_(src)
                self.image_ref[0] = self.static_in
            self.graph.replay()
            return
        if not events:
            for s in self.steps:
                s()
            return
        i = 0
        for lo, hi, (e0, e1) in sorted((self.marks[k][0], self.marks[k][1], ev) for k, ev in events.items()):
            self.run_range(i, lo)
            e0.record()
            self.run_range(lo, hi)
            e1.record()
            i = hi
        self.run_range(i, len(self.steps))

    def run_range(self, lo: int, hi: int) -> None:
        for s in self.steps[lo:hi]:
            s()


def _dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ conv helpers
def padded_input(plan: Plan, rows: int, C: int):
    """Plan-owned staging buffer for a caller-supplied map of C channels: (Rows the convs read, Rows the copy-in writes).
    The conv kernel reads any C % 4 == 0 (a partial 32-channel chunk is masked in its loader: EfficientNet's 48 / 136);
    other widths are rounded up to a multiple of 4 with zero channels that are never written (dedicated, not pooled)."""
    Cp = (C + 3) // 4 * 4
    if Cp == C:
        r = plan.pool.get(rows, C)
        return r, r
    buf = torch.zeros(rows, Cp, dtype=torch.float32, device=plan.device)
    plan.keep.append(buf)
    return Rows(buf), Rows(buf, 0, C)


def add_conv(plan: Plan, name: str, x: Rows, segs: Segs, conv: torch.nn.Conv2d, y: Rows, *, bn=None, act=ACT_NONE,
             res: Optional[Rows] = None, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
             Cout: Optional[int] = None, act_c0: int = 0, seg_param=None, tag: int = 0, fold=None,
             gate: Optional[torch.Tensor] = None, gate_b: Optional[torch.Tensor] = None, gate_act: int = ACT_NONE,
             gn_stats: Optional[torch.Tensor] = None, gn_groups: int = 0,
             x2: Optional[Rows] = None, x2_stride: int = 1, x2_hw: Optional[Tuple[int, int]] = None, res_up: bool = False) -> Segs:
    """conv (+folded BN / bias) (+res) (+act).  `weight`/`bias` override conv's own (fused multi-conv launches);
    `fold` = (scale, shift) overrides the epilogue constants altogether (convs with different BN / bias merged by hand)."""
    dev = plan.device
    if getattr(plan, "tail_steps", None):
        raise FdError("add_conv: a tail launch is still pending (_flush_tail_steps right after closing the mark of a tag = 1 conv)")
    w = conv.weight if weight is None else weight
    b = (conv.bias if bias is None else bias)
    k, stride, pad, dil = conv.kernel_size[0], conv.stride[0], conv.padding, conv.dilation[0]
    if isinstance(pad, str):  # 'same'
        pad = dil * (k - 1) // 2
    else:
        pad = pad[0]
    Cin, co = w.shape[1], (w.shape[0] if Cout is None else Cout)
    if res_up and (plan.precision != "f32" or k != 1 or stride != 1 or pad != 0 or segs.nseg != 1 or res is None or gate is not None or gn_stats is not None or x2 is not None):
        raise FdError("add_conv: res_up (a half-resolution addend behind the activation) needs an exact-fp32 1x1 stride-1 single-level conv with a residual")
    if x2 is not None and (plan.precision != "f32" or k != 1 or stride != 1 or segs.nseg != 1 or (Cin - x2.C) % 32 or x2.C % 32 or gate is not None or gn_stats is not None):
        raise FdError("add_conv: a K-concatenated second input needs an exact-fp32 1x1 stride-1 single-level conv with 32-aligned channel counts")
    if Cin % 4 and x.co == 0 and x.C == x.cs and x.C % 4 == 0 and 0 < x.C - Cin < 4:
        # input width that is not a multiple of 4: the caller staged the map into a zero-padded buffer (padded_input),
        # the weights get matching zero input channels
        w = torch.nn.functional.pad(_dev(w, dev).detach(), (0, 0, 0, 0, 0, x.C - Cin))
        Cin = x.C
    split = plan.precision == "f16x3" or (plan.precision == "mixed" and k == 1 and stride == 1 and gate is None and gn_stats is None
                                           and Cin % 32 == 0)
    # <= 8 output channels (the centre-ness / box predictor): the vector-unit kernel, where the map has enough tiles to fill the chip
    # (a gate there is the preceding GroupNorm's affine: gate + gate_b, applied to the patch)
    narrow = (ops.NARROW and plan.precision in ("f32", "mixed") and res is None and (gate is None or gate_b is not None) and gn_stats is None and w.shape[0] == co
              and ops.narrow_ok(Cin, co, k, stride, pad, dil) and ops.narrow_tiles(segs) >= ops.NARROW_MIN_TILES)
    if narrow:
        split = False
    wino, wino_ks = False, 1
    if (not narrow and plan.winograd and ops.wino_ok(Cin, co, k, stride, pad, dil) and y.cs % 4 == 0 and y.co % 4 == 0 and
            (res is None or (res.cs % 4 == 0 and res.co % 4 == 0))):
        wino, wino_ks = ops.wino_choice(segs, Cin, co, dil)
        wino = wino or tag == 1                   # (the head tower -- the roofline kernel -- always runs the Winograd kernel)
    # F(4x4, 3x3) where its cost model beats F(2x2, 3x3) / the direct kernel: exact-fp32 plans only, no gate / statistics epilogue
    wino4, w4_ks = False, 1
    if (not narrow and plan.winograd and gate is None and gn_stats is None and ops.wino4_ok(Cin, co, k, stride, pad, dil) and y.cs % 4 == 0 and y.co % 4 == 0
            and (res is None or (res.cs % 4 == 0 and res.co % 4 == 0))):
        wino4, w4_ks = ops.wino4_choice(segs, Cin, co, dil)
    if narrow:
        wp = ops.pack_conv_weight_narrow(_dev(w, dev))
    elif wino4:
        wino, wino_ks = True, w4_ks
        wp = ops.pack_conv_weight_wino4(_dev(w, dev))
    elif wino:
        wp = ops.pack_conv_weight_wino(_dev(w, dev))
    else:
        wp = ops.pack_conv_weight_f16x3(_dev(w, dev)) if split else ops.pack_conv_weight(_dev(w, dev))
    # GEMM-addressed fp32 layers also get their weights in MFMA fragment order: the wave-autonomous tile (FD_TILE_WAVE64) becomes selectable
    wfrag = None
    if (not wino and not narrow and not split and x2 is None and not res_up and WAVE_TILE and gate is None and ops.wave_ok(Cin, co, k, stride, pad) and act_c0 % 32 == 0
            and act in (ACT_NONE, ACT_RELU, ACT_SILU) and y.cs % 4 == 0 and y.co % 4 == 0 and (res is None or (res.cs % 4 == 0 and res.co % 4 == 0))):
        if w.shape[0] == co and w.shape[1] == Cin:
            wfrag = ops.pack_conv_weight_wave(_dev(w, dev))
    scale = shift = None
    if fold is not None:
        scale, shift = fold
    elif bn is not None:
        scale, shift = ops.fold_bn(_dev(bn.weight, dev), _dev(bn.bias, dev), _dev(bn.running_mean, dev),
                                   _dev(bn.running_var, dev), bn.eps, _dev(b, dev) if b is not None else None)
    elif b is not None:
        shift = _dev(b, dev)
    plan.keep += [wp, scale, shift, wfrag]
    out = ops.conv_out_segs(segs, k, stride, pad, dil)
    # split-K scratch: taken from the pool and handed straight back (stream order makes the sharing safe)
    ws_rows = ops.KSPLIT_MAX * out.rows
    if narrow:
        ws = None
    elif gn_stats is not None:
        if wino and wino_ks > 1:
            raise FdError("add_conv: gn_stats with a split-K Winograd launch (the caller checks ops.wino_choice first)")
        ws = None                  # (row-group statistics come out of the conv's own epilogue: no split-K, whose combine launch has none)
    elif wino:
        ws = plan.pool.get(wino_ks * out.rows, (co + 3) & ~3) if wino_ks > 1 else None
    elif x2 is not None:
        ws = None                  # (no split-K form)
    else:
        ws = plan.pool.get(ws_rows, (co + 3) & ~3) if (plan.autotune and ws_rows * ((co + 3) & ~3) <= 64 * 1024 * 1024) else None
    call = ops.conv_call(x, segs, wp, y, Cin=Cin - (x2.C if x2 is not None else 0), Cout=co, k=k, stride=stride, pad=pad, dil=dil, scale=scale,
                         shift=shift, res=res, act=act, act_c0=act_c0, seg_param=seg_param, tag=tag,
                         precision=1 if split else 0, workspace=ws.buf if ws is not None else None,
                         tile=_lib.NARROW_TILE if narrow else ((_lib.WINO4_TILE if wino4 else _lib.WINO_TILE) if wino else 0), ksplit=wino_ks if wino else 1,
                         gate=gate, w_frag=wfrag,
                         gate_b=gate_b, gate_act=gate_act, gn_stats=gn_stats, gn_groups=gn_groups, x2=x2, x2_stride=x2_stride, x2_hw=x2_hw, res_up=res_up)
    tail = None
    sk = 0
    if wino4 and wino_ks <= 1 and ops.W4_SK:
        # the persistent form of the F(4x4) launch (work queue per XCD, the last partial round of items cut into pieces) where it was measured faster
        if getattr(plan, "sk_ws", None) is None:
            plan.sk_ws = ops.sk_workspace(256, dev)         # one per plan: the launches of a plan are stream-ordered; every lane of the pipeline has its own plan
        hw_ = "+".join(f"{h}x{w}" for h, w in segs.level_hw())
        sk = ops.wino4_sk_choice(call, f"B{segs.batch}|{hw_}|{Cin}>{co}|d{dil}|res{int(res is not None)}", plan.sk_ws)
        if sk:
            call = ops.conv_sk(call, sk, plan.sk_ws)
            plan.sk_of = getattr(plan, "sk_of", {})
            plan.sk_of[name] = sk
    if tag == 1 and wino4 and wino_ks <= 1 and TOWER_TAIL_SPLIT and not sk:
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        total, live = ops.conv_workgroups(call)
        full = total // ncu * ncu          # (one workgroup of that kernel owns a CU; ncu % 8 == 0: the slice starts on an XCD boundary)
        if ncu % 8 == 0 and 4 * ncu <= full < total:
            main = ops.conv_wg_slice(call, 0, full)
            tail = ops.conv_wg_slice(call, full, total - full, tag=0)
            tail_share = ops.conv_workgroups(tail)[1] / live
            call = main
    plan.add(name, call)
    if tail is not None:
        plan.tail_of = getattr(plan, "tail_of", {})
        plan.tail_of[name] = {"workgroups": total, "main": full, "live": live, "main_share": 1.0 - tail_share}
    if ws is not None:
        plan.pool.put(ws)
    if narrow:
        plan.tiles[name] = _lib.NARROW_TILE
    elif wino:
        plan.tiles[name] = (_lib.WINO4_TILE if wino4 else _lib.WINO_TILE) | ((wino_ks if wino_ks > 1 else 0) << 8)
    elif gate is not None:
        plan.tiles[name] = 0                 # (the library picks the tile of a gated conv)
    elif plan.autotune:
        hw = "+".join(f"{h}x{w}" for h, w in segs.level_hw())
        # (res_up: the quarter-size addend costs next to nothing -- the plain layer's measured tile, mapped below onto the tiles that form is built for)
        key = f"B{segs.batch}|{hw}|{Cin}>{co}|k{k}s{stride}p{pad}d{dil}|res{int(res is not None and not res_up)}|xcs{x.cs}|ycs{y.cs}"
        if split:
            key = "f16x3|" + key
        if x2 is not None:
            key += f"|x2s{x2_stride}c{x2.C}"
        if res_up and key not in ops._tune_table():
            # a table MISS of a res_up layer is timed with res_mode 2 set, i.e. over the RUP tiles only: that restricted winner gets a key of its own and never
            # lands under the plain layer's key (which the unfused FD_FPN_UP_FUSED=0 lateral of the same shape reads).  A plain-key HIT is still mapped below.
            key += "|rup"
        plan.tiles[name] = ops.autotune_conv(call, key, out.rows, co, -(-Cin // 32) * k * k, pair=plan.pair_tuned and tag != 1)
        if res_up:
            t = {1: 8, 2: 8, 3: 9, 4: 4, 7: 8, 8: 8, 9: 9}.get(plan.tiles[name] & 0xFF, 9)
            call.params.tile, call.params.ksplit = t, 1
            plan.tiles[name] = t
        if x2 is not None:         # the dual-source loader is built for the tiles the bottleneck expansions use: map the choice onto them, no split-K
            t = {1: 7, 2: 8, 3: 9, 4: 4, 7: 7, 8: 8, 9: 9}.get(plan.tiles[name] & 0xFF, 8)
            call.params.tile, call.params.ksplit = t, 1
            plan.tiles[name] = t
        if gn_stats is not None and (plan.tiles[name] & 0xFF) not in (0, 2, 3, 4, 8, 9, _lib.WAVE_TILE):
            call.params.tile = plan.tiles[name] = 8      # the statistics epilogue exists for the one- / two-sub-tile tiles (and WAVE64 / Winograd)
    plan.flops += 2 * out.rows * co * Cin * k * k
    info = {"k": k, "stride": stride, "dil": dil, "Cin": Cin, "Cout": co, "rows": out.rows,
            "family": "narrow3x3 (vector unit)" if narrow else ("winograd3x3" if wino else ("1x1" if k == 1 else f"direct{k}x{k}")),
            # multiplies saved on the matrix pipe: F(4x4,3x3) 36 per 16 outputs, F(2x2,3x3) 16 per 4, direct 9 per output
            "mfma_div": 4.0 if wino4 else (2.25 if wino else 1.0)}
    if tail is not None:       # the layer as two launches: FLOPs split by the share of non-empty workgroups
        main_flops = int(round(2 * out.rows * co * Cin * k * k * (1.0 - tail_share)))
        plan.step_flops[len(plan.steps) - 1] = main_flops
        plan.step_info[len(plan.steps) - 1] = info
        plan.tail_steps = getattr(plan, "tail_steps", [])
        plan.tail_steps.append((name + ".tail", tail, 2 * out.rows * co * Cin * k * k - main_flops, dict(info)))
        return out
    plan.step_flops[len(plan.steps) - 1] = 2 * out.rows * co * Cin * k * k
    plan.step_info[len(plan.steps) - 1] = info
    return out


# ------------------------------------------------------------------------------------------------ plan input
def add_input(plan: Plan, x4: Rows, batch: int, H: int, W: int, image_ref: List) -> None:
    """First step of a detector plan: fill the stem's [N][H][W][4] input from whatever the caller hands over
    (plan.input_mode): None = fp32 NCHW (the reference's tensor, dataset/voc.py:141-173); 'u8' = one uint8 [N,H,W,3] batch,
    normalised on the device; 'collate' = a list of resized uint8 [h_n, w_n, 3] images of different sizes, padded to the
    batch canvas and normalised in one launch (voc.py:128-132,141-156).  (mean, std) = plan.input_u8."""
    mode = getattr(plan, "input_mode", None) or ("u8" if getattr(plan, "input_u8", None) else None)
    if mode == "u8":
        mean, std = plan.input_u8
        plan.add("input.preprocess_u8", lambda: ops.preprocess_u8(image_ref[0], x4.buf, mean, std))
    elif mode == "collate":
        mean, std = plan.input_u8
        hold = [None]

        def run():
            _, hold[0] = ops.collate_u8(image_ref[0], H, W, mean, std, out=x4.buf)   # `hold`: pointer table alive until the next run
        plan.add("input.collate_u8", run)
    else:
        plan.add("input.nchw3_to_nhwc4", lambda: ops.nchw3_to_nhwc4(image_ref[0], x4.buf))


# ------------------------------------------------------------------------------------------------ ResNet-50 trunk
def build_resnet50(plan: Plan, trunk, batch: int, H: int, W: int, image_ref: List[torch.Tensor]):
    """torchvision-style ResNet-50 v1.5 trunk -> (C3, C4, C5) as (Rows, Segs).  `trunk` has conv1, bn1, layer1..4."""
    dev, pool = plan.device, plan.pool
    stem_own = STEM_KERNEL and plan.precision in ("f32", "mixed") and tuple(trunk.conv1.weight.shape) == (64, 3, 7, 7)
    mode = getattr(plan, "input_mode", None) or ("u8" if getattr(plan, "input_u8", None) else None)
    # the reference's fp32 NCHW batch is read by the stem's own patch loader (fd_stem7x7_nchw3): no [N][H][W][4] copy, no conversion launch
    nchw_in = STEM_NCHW and stem_own and mode is None
    x4 = None
    if not nchw_in:
        x4 = pool.get(batch * H * W, 4)
        add_input(plan, x4, batch, H, W, image_ref)
    s_in = Segs.make(batch, [(H, W)])
    # stem 7x7 s2 + BN + ReLU
    sc, sf = ops.fold_bn(_dev(trunk.bn1.weight, dev), _dev(trunk.bn1.bias, dev), _dev(trunk.bn1.running_mean, dev),
                         _dev(trunk.bn1.running_var, dev), trunk.bn1.eps)
    s1 = ops.conv_out_segs(s_in, 7, 2, 3, 1)
    H1, W1 = s1.H[0], s1.W[0]
    H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
    s2 = Segs.make(batch, [(H2, W2)])
    fused_pool = STEM_POOL and stem_own

    def nchw_image() -> torch.Tensor:
        x = image_ref[0]
        if tuple(x.shape) != (batch, 3, H, W):
            raise FdError(f"plan input: expected a [{batch}, 3, {H}, {W}] batch, got {tuple(x.shape)}")
        return x

    if fused_pool:
        # conv1 + bn1 + relu + maxpool as ONE launch (fd_stem7x7_pool_nhwc4): the 64-channel stride-2 map is never written (resnet50.py:68-80)
        wp = ops.pack_stem7_weight(_dev(trunk.conv1.weight, dev))
        y2 = pool.get(s2.rows, 64)
        if nchw_in:
            plan.add("backbone.conv1+maxpool", lambda: ops.stem7x7_nchw(nchw_image(), wp, y2, sc, sf, pool=True))
        else:
            plan.add("backbone.conv1+maxpool", lambda: ops.stem7x7_pool(x4, wp, y2, batch, H, W, sc, sf))
        plan.keep += [wp, sc, sf]
        plan.flops += 2 * s1.rows * 64 * 147
        plan.step_flops[len(plan.steps) - 1] = 2 * s1.rows * 64 * 147
        if x4 is not None:
            pool.put(x4)
    y1 = pool.get(s1.rows, 64) if not fused_pool else None
    if fused_pool:
        pass
    elif stem_own:
        wp = ops.pack_stem7_weight(_dev(trunk.conv1.weight, dev))      # the dedicated stem kernel (fd_stem.hip): patch + filters staged in LDS
        if nchw_in:
            plan.add("backbone.conv1", lambda: ops.stem7x7_nchw(nchw_image(), wp, y1, sc, sf, ACT_RELU))
        else:
            plan.add("backbone.conv1", lambda: ops.stem7x7(x4, wp, y1, batch, H, W, sc, sf, ACT_RELU))
    else:
        wp = ops.pack_stem_weight(_dev(trunk.conv1.weight, dev))
        plan.add("backbone.conv1", ops.conv_call(x4, s_in, wp, y1, Cin=4, Cout=64, k=7, stride=2, pad=3, scale=sc, shift=sf,
                                                 act=ACT_RELU, stem=True))
    if not fused_pool:
        plan.keep += [wp, sc, sf]
        plan.flops += 2 * s1.rows * 64 * 147
        plan.step_flops[len(plan.steps) - 1] = 2 * s1.rows * 64 * 147
        if x4 is not None:
            pool.put(x4)
        y2 = pool.get(s2.rows, 64)
        plan.add("backbone.maxpool", lambda: ops.maxpool(y1, y2, batch, H1, W1, 3, 2, 1))
        pool.put(y1)
    x, sx = y2, s2
    feats = []
    blocks_all = [(li, bi, blk) for li in (1, 2, 3, 4) for bi, blk in enumerate(getattr(trunk, f"layer{li}"))]
    o1_ready = None            # the NEXT block's conv1 output when the previous block's conv3 launch has already produced it (back-to-back GEMMs)
    for gi, (li, bi, blk) in enumerate(blocks_all):
        nm = f"backbone.layer{li}.{bi}"
        planes = blk.conv1.weight.shape[0]
        so = ops.conv_out_segs(sx, 3, blk.conv2.stride[0], 1, 1)
        ds = blk.downsample
        if o1_ready is not None:
            o1, o1_ready = o1_ready, None       # produced by the previous block's conv3 launch
        else:
            o1 = pool.get(sx.rows, planes)
            add_conv(plan, nm + ".conv1", x, sx, blk.conv1, o1, bn=blk.bn1, act=ACT_RELU)
        o2 = pool.get(so.rows, planes)
        add_conv(plan, nm + ".conv2", o1, sx, blk.conv2, o2, bn=blk.bn2, act=ACT_RELU)
        pool.put(o1)
        dual = (DUAL_DS and ds is not None and plan.precision == "f32" and ds[0].kernel_size[0] == 1 and ds[0].bias is None and blk.conv3.bias is None
                and planes % 32 == 0 and x.C % 32 == 0 and x.cs % 4 == 0 and x.co % 4 == 0 and ds[0].stride[0] == blk.conv2.stride[0])
        if dual:
            # out = relu(bn3(conv3(o2)) + bn_d(downsample(x))) as ONE GEMM over K = planes + inplanes: the BatchNorm scales go into the two filter banks,
            # the shifts add up, the strided input of the downsample conv is the launch's second source -- the identity map is never materialised
            # (resnet50.py:68-80; 420 MB per 16 images in layer1, written and read back before)
            f3 = ops.fold_bn(_dev(blk.bn3.weight, dev), _dev(blk.bn3.bias, dev), _dev(blk.bn3.running_mean, dev), _dev(blk.bn3.running_var, dev), blk.bn3.eps)
            fdn = ops.fold_bn(_dev(ds[1].weight, dev), _dev(ds[1].bias, dev), _dev(ds[1].running_mean, dev), _dev(ds[1].running_var, dev), ds[1].eps)
            wcat = torch.cat([_dev(blk.conv3.weight, dev) * f3[0][:, None, None, None], _dev(ds[0].weight, dev) * fdn[0][:, None, None, None]], 1)
            out = pool.get(so.rows, 4 * planes)
            add_conv(plan, nm + ".conv3+downsample", o2, so, blk.conv3, out, weight=wcat, fold=(None, (f3[1] + fdn[1]).contiguous()), act=ACT_RELU,
                     x2=x, x2_stride=ds[0].stride[0], x2_hw=(sx.H[0], sx.W[0]))
            plan.step_info[len(plan.steps) - 1]["dual"] = (planes, x.C, ds[0].stride[0])
            pool.put(o2)
            if not (feats and feats[-1][0] is x):
                pool.put(x)
            x, sx = out, so
            if li >= 2 and bi == len(getattr(trunk, f"layer{li}")) - 1:
                feats.append((x, sx))
            continue
        if ds is not None:
            idt = pool.get(so.rows, 4 * planes)
            add_conv(plan, nm + ".downsample", x, sx, ds[0], idt, bn=ds[1])
        else:
            idt = x
        out = pool.get(so.rows, 4 * planes)
        nxt = blocks_all[gi + 1][2] if gi + 1 < len(blocks_all) else None
        n2 = nxt.conv1.weight.shape[0] if nxt is not None else 0
        # (the back-to-back kernel is wave-autonomous, 64 rows per wave: below ~4 waves per SIMD-quarter of the chip its workgroups leave most of every CU idle --
        #  at batch 1 x 512 x 512 the seam was 0.053 ms against 0.016 + 0.018 as two launches: B2B_MIN_ROWS)
        if (str(li) in B2B_LAYERS and so.rows >= B2B_MIN_ROWS and nxt is not None and plan.precision == "f32" and blk.conv3.bias is None and nxt.conv1.bias is None
                and nxt.conv1.kernel_size[0] == 1 and nxt.conv1.stride[0] == 1 and nxt.conv1.weight.shape[1] == 4 * planes
                and ops.b2b_ok(planes, 4 * planes, n2) and all(r.cs % 4 == 0 and r.co % 4 == 0 for r in (o2, out, idt))):
            # conv3 (+bn3 + identity, ReLU) and the NEXT block's conv1 (+bn1, ReLU) as one launch: `out` is written (next residual, downsample input)
            # but not read back for the conv1 GEMM (resnet50.py:68-80)
            o1_ready = pool.get(so.rows, n2)
            f3 = ops.fold_bn(_dev(blk.bn3.weight, dev), _dev(blk.bn3.bias, dev), _dev(blk.bn3.running_mean, dev), _dev(blk.bn3.running_var, dev), blk.bn3.eps)
            fn1 = ops.fold_bn(_dev(nxt.bn1.weight, dev), _dev(nxt.bn1.bias, dev), _dev(nxt.bn1.running_mean, dev), _dev(nxt.bn1.running_var, dev), nxt.bn1.eps)
            wa = ops.pack_conv_weight_wave(_dev(blk.conv3.weight, dev))
            wb = ops.pack_conv_weight_wave(_dev(nxt.conv1.weight, dev))
            plan.keep += [f3, fn1, wa, wb]
            nl, nb_ = blocks_all[gi + 1][0], blocks_all[gi + 1][1]
            plan.add(f"{nm}.conv3+layer{nl}.{nb_}.conv1",
                     ops.conv_b2b_call(o2, wa, out, wb, o1_ready, K1=planes, N1=4 * planes, N2=n2, scale1=f3[0], shift1=f3[1], res=idt, act1=ACT_RELU,
                                       scale2=fn1[0], shift2=fn1[1], act2=ACT_RELU))
            fl = 2 * so.rows * (4 * planes * planes + n2 * 4 * planes)
            plan.flops += fl
            plan.step_flops[len(plan.steps) - 1] = fl
            plan.step_info[len(plan.steps) - 1] = {"k": 1, "stride": 1, "dil": 1, "Cin": planes, "Cout": 4 * planes, "rows": so.rows, "family": "1x1",
                                                   "mfma_div": 1.0, "b2b": (4 * planes, n2)}
        else:
            add_conv(plan, nm + ".conv3", o2, so, blk.conv3, out, bn=blk.bn3, act=ACT_RELU, res=idt)
        pool.put(o2)
        if idt is not x:
            pool.put(idt)
        if not (feats and feats[-1][0] is x):  # keep C3/C4/C5 alive
            pool.put(x)
        x, sx = out, so
        if li >= 2 and bi == len(getattr(trunk, f"layer{li}")) - 1:
            feats.append((x, sx))
    return feats  # [(C3, segs), (C4, segs), (C5, segs)]


# ------------------------------------------------------------------------------------------------ EfficientNet trunk
def build_efficientnet(plan: Plan, net, batch: int, H: int, W: int, image_ref: List[torch.Tensor], keep=(2, 3, 4)):
    """efficientnet_pytorch 0.7.1 `EfficientNet.extract_endpoints` (the call behind the reference's EfficientNetV1.forward,
    model/backbone/efficientnetv1.py:24-26) -> [(Rows, Segs)] for reduction_1..reduction_5; endpoints not in `keep` are
    returned to the pool as soon as the next block has consumed them (their entry is None).  `net` is the parameter
    container model/backbone/efficientnetv1._EfficientNet.

    Per MBConv block: expand 1x1 + BN + swish (MFMA conv kernel; widths that are not multiples of 32 are masked in its
    loader) -> depthwise k x k stride s with the block's static TF-"SAME" padding + BN + swish (fd_dwconv2d_nhwc) ->
    squeeze-excitation in place (fd_se_scale_nhwc) -> project 1x1 + BN (+ block input when stride 1 and cin == cout)."""
    dev, pool = plan.device, plan.pool
    x4 = pool.get(batch * H * W, 4)
    add_input(plan, x4, batch, H, W, image_ref)

    def out_hw(h: int, w: int, k: int, s: int, pad) -> Tuple[int, int]:
        return (h + pad[0] + pad[1] - k) // s + 1, (w + pad[0] + pad[1] - k) // s + 1

    def bn_fold(bn):
        sc, sf = ops.fold_bn(_dev(bn.weight, dev), _dev(bn.bias, dev), _dev(bn.running_mean, dev), _dev(bn.running_var, dev), bn.eps)
        plan.keep += [sc, sf]
        return sc, sf

    # stem: 3x3 stride 2 + BN + swish on the [N][H][W][4] image
    stem = net._conv_stem
    C0 = stem.weight.shape[0]
    h, w = out_hw(H, W, 3, 2, net.stem_pad)
    ws = ops.pack_stem3_weight(_dev(stem.weight, dev))
    sc0, sf0 = bn_fold(net._bn0)
    plan.keep.append(ws)
    x = pool.get(batch * h * w, C0)
    plan.add("backbone._conv_stem", lambda x=x, h=h, w=w: ops.stem_conv3(x4.buf, ws, x, batch, H, W, 3, 2, net.stem_pad[0], net.stem_pad[0],
                                                                       h, w, sc0, sf0, ACT_SILU))
    plan.flops += 2 * batch * h * w * C0 * 27
    pool.put(x4)
    blocks = list(net._blocks)
    ends: List[Optional[Tuple[Rows, Segs]]] = []
    for bi, blk in enumerate(blocks):
        nm = f"backbone._blocks.{bi}"
        segs = Segs.make(batch, [(h, w)])
        inp = x
        mid = blk._depthwise_conv.weight.shape[0]
        ho, wo = out_hw(h, w, blk.kernel, blk.stride, blk.pad)
        if ho < 1 or wo < 1:
            raise FdError("EfficientNet: input too small")
        so = Segs.make(batch, [(ho, wo)])
        # expand 1x1 -> depthwise in ONE launch where the tile's input patch fits LDS (the early, high-resolution stages: the six-fold expanded map never reaches HBM,
        # and the SE pooling comes out of the same kernel as per-tile partial sums)
        # (k = 5 at stride 2 stays separate: a 6 x 6 output tile needs a 15 x 15 patch -- 1.56 x the expand GEMM and a depthwise stage that fills 56 % of its threads:
        #  0.70 ms fused against 0.60 ms for B3's block 5, profiles/r05_layer_times_fcos_b3_mbconv_fused.tsv)
        fused = (MBCONV_FUSED and blk.expand != 1 and plan.precision == "f32" and blk._expand_conv.bias is None and not (blk.kernel == 5 and blk.stride == 2)
                 and ops.mbconv_fused_ok(x.C, mid, blk.kernel, blk.stride) and x.C == blk._expand_conv.weight.shape[1])
        d = pool.get(so.rows, mid)
        wd = ops.pack_dwk_weight(_dev(blk._depthwise_conv.weight, dev))
        sc, sf = bn_fold(blk._bn1)
        pool_part = None
        if fused:
            wef = ops.pack_mbconv_expand_weight(_dev(blk._expand_conv.weight, dev))
            sc_e, sf_e = bn_fold(blk._bn0)
            pool_part, ntile = ops.mbconv_pool_buffer(batch, ho, wo, mid, blk.kernel, blk.stride, dev)
            plan.keep += [wef, pool_part]
            plan.add(nm + "._expand+depthwise", lambda x=x, d=d, wef=wef, sc_e=sc_e, sf_e=sf_e, wd=wd, sc=sc, sf=sf, pp=pool_part, h=h, w=w, ho=ho, wo=wo, blk=blk:
                     ops.mbconv_expand_dw(x, wef, sc_e, sf_e, wd, sc, sf, d, pp, batch, h, w, blk.kernel, blk.stride, blk.pad[0], blk.pad[0], ho, wo))
            fl = 2 * segs.rows * mid * x.C
            plan.flops += fl + 2 * so.rows * mid * blk.kernel * blk.kernel
            plan.step_flops[len(plan.steps) - 1] = fl
            plan.step_info[len(plan.steps) - 1] = {"k": 1, "stride": 1, "dil": 1, "Cin": x.C, "Cout": mid, "rows": segs.rows,
                                                   "family": "mbconv expand 1x1 + depthwise (one launch)", "mfma_div": 1.0}
        else:
            if blk.expand != 1:
                e = pool.get(segs.rows, mid)
                add_conv(plan, nm + "._expand_conv", x, segs, blk._expand_conv, e, bn=blk._bn0, act=ACT_SILU)
            else:
                e = x
            plan.add(nm + "._depthwise_conv", lambda e=e, d=d, wd=wd, sc=sc, sf=sf, h=h, w=w, ho=ho, wo=wo, blk=blk:
                     ops.dwconv2d(e, wd, d, batch, h, w, blk.kernel, blk.stride, blk.pad[0], blk.pad[0], ho, wo, sc, sf, ACT_SILU))
            plan.flops += 2 * so.rows * mid * blk.kernel * blk.kernel
            if e is not x:
                pool.put(e)
        w1 = _dev(blk._se_reduce.weight, dev).reshape(blk._se_reduce.weight.shape[0], -1).contiguous()
        b1 = _dev(blk._se_reduce.bias, dev)
        w2 = _dev(blk._se_expand.weight, dev).reshape(mid, -1).contiguous()
        b2 = _dev(blk._se_expand.bias, dev)
        sews = ops.se_workspace(batch, ho * wo, mid, dev)
        plan.keep += [wd, w1, b1, w2, b2, sews]
        out = pool.get(so.rows, blk.cout)
        if SE_GATE_IN_PROJECT and plan.precision in ("f32", "mixed"):
            # the gate is multiplied in by the project conv's loader: no scaling pass (a read and a write of the expanded map) at all
            if pool_part is not None:       # ... and its pooling came out of the fused expand + depthwise launch: no pass over the map at all
                plan.add(nm + "._se", lambda pp=pool_part, nt=ntile, w1=w1, b1=b1, w2=w2, b2=b2, sews=sews, hw=ho * wo, mid=mid:
                         ops.se_gate_from_pool(pp, nt, w1, b1, w2, b2, batch, hw, mid, w1.shape[0], sews))
            else:
                plan.add(nm + "._se", lambda d=d, w1=w1, b1=b1, w2=w2, b2=b2, sews=sews, hw=ho * wo:
                         ops.se_gate(d, w1, b1, w2, b2, batch, hw, w1.shape[0], sews))
            add_conv(plan, nm + "._project_conv", d, so, blk._project_conv, out, bn=blk._bn2, res=inp if blk.skip else None,
                     gate=ops.se_gate_view(sews, batch, ho * wo, mid))
        else:
            plan.add(nm + "._se", lambda d=d, w1=w1, b1=b1, w2=w2, b2=b2, sews=sews, hw=ho * wo:
                     ops.se_scale(d, w1, b1, w2, b2, d, batch, hw, w1.shape[0], sews))
            add_conv(plan, nm + "._project_conv", d, so, blk._project_conv, out, bn=blk._bn2, res=inp if blk.skip else None)
        pool.put(d)
        # extract_endpoints: the activation in front of every resolution drop is an endpoint (and the last block's output)
        if ho < h and len(ends) in keep:
            ends.append((inp, segs))
        else:
            if ho < h:
                ends.append(None)
            pool.put(inp)
        x, h, w = out, ho, wo
    ends.append((x, Segs.make(batch, [(h, w)])))
    if len(ends) != 5:
        raise FdError(f"EfficientNet: expected 5 endpoints, got {len(ends)} (input too small?)")
    return ends


# ------------------------------------------------------------------------------------------------ HISFCOS FPN
def _his_block(plan: Plan, name: str, blk, x: Rows, segs: Segs, out: Rows) -> None:
    """HisBlock (HISFcos.py:95-112); `out` is a [M, feature] view (a level slice of the pyramid buffer)."""
    dev, pool = plan.device, plan.pool
    M = segs.rows
    half = blk.conv1.weight.shape[0]
    N, HW = segs.batch, segs.H[0] * segs.W[0]
    cat1 = pool.get(M, 2 * half)
    cat2 = pool.get(M, 2 * half)
    # conv1 (+bn1, SiLU) and conv2 (bias only) read the same map: ONE 2*half-wide launch writing [x2 | x1] into cat2
    # (the activation applies to channels >= half).  x1 lives in cat2's upper half until conv3 overwrites it with y, so
    # conv4 sees [x2 | y] and gets its input channels swapped to match (the reference order is [y | x2], HISFcos.py:111).
    s1, t1 = ops.fold_bn(_dev(blk.bn1.weight, dev), _dev(blk.bn1.bias, dev), _dev(blk.bn1.running_mean, dev),
                         _dev(blk.bn1.running_var, dev), blk.bn1.eps, _dev(blk.conv1.bias, dev) if blk.conv1.bias is not None else None)
    bias2 = _dev(blk.conv2.bias, dev) if blk.conv2.bias is not None else torch.zeros(half, device=dev)
    w12 = torch.cat([blk.conv2.weight.detach(), blk.conv1.weight.detach()], 0)
    fold = (torch.cat([torch.ones(half, device=dev), s1]).contiguous(), torch.cat([bias2, t1]).contiguous())
    add_conv(plan, name + ".conv1+2", x, segs, blk.conv1, cat2, weight=w12, Cout=2 * half, act=ACT_SILU, act_c0=half, fold=fold)
    x1 = cat2.slice(half, half)
    wd = ops.pack_dw_weight(_dev(blk.conv1_1.weight, dev))
    sc, sf = ops.fold_bn(_dev(blk.bn2.weight, dev), _dev(blk.bn2.bias, dev), _dev(blk.bn2.running_mean, dev),
                         _dev(blk.bn2.running_var, dev), blk.bn2.eps)
    u = cat1.slice(0, half)
    se = blk.conv1_2.excitation
    w1 = _dev(se[0].weight, dev).reshape(se[0].weight.shape[0], -1).contiguous()
    b1 = _dev(se[0].bias, dev)
    w2 = _dev(se[2].weight, dev).reshape(se[2].weight.shape[0], -1).contiguous()
    b2 = _dev(se[2].bias, dev)
    ws = ops.se_workspace(N, HW, half, dev)
    v = cat1.slice(half, half)
    cr = w1.shape[0]
    # (the depthwise and the squeeze-excitation branch are independent -- HISFcos.py:100-104 -- but forking the second onto another stream costs the batch-1 plan
    #  0.15 ms in cross-stream events for seven pairs: measured)
    plan.add(name + ".conv1_1", lambda: ops.dwconv3x3(x1, wd, u, segs, sc, sf, ACT_RELU))
    plan.add(name + ".conv1_2", lambda: ops.se_scale(x1, w1, b1, w2, b2, v, N, HW, cr, ws))
    plan.keep += [wd, sc, sf, w1, b1, w2, b2, ws]
    add_conv(plan, name + ".conv3", cat1, segs, blk.conv3, cat2.slice(half, half), bn=blk.bn3, act=ACT_RELU)
    w4 = blk.conv4.weight.detach()
    add_conv(plan, name + ".conv4", cat2, segs, blk.conv4, out, bn=blk.bn4, act=ACT_SILU,
             weight=torch.cat([w4[:, half:], w4[:, :half]], 1))
    pool.put(cat1); pool.put(cat2)


def build_his_fpn(plan: Plan, fpn, feats):
    """HalfInvertedStageFPN.forward (HISFcos.py:147-179) -> pyramid (Rows, Segs) with levels P3..P7."""
    pool = plan.pool
    (c3, s3), (c4, s4), (c5, s5) = feats
    B = s3.batch
    F = fpn.tf1.weight.shape[0]
    h5, w5 = s5.H[0], s5.W[0]
    hw = [(s3.H[0], s3.W[0]), (s4.H[0], s4.W[0]), (h5, w5), (h5 // 2, w5 // 2), (h5 // 4, w5 // 4)]
    if hw[4][0] < 1 or hw[4][1] < 1:
        raise FdError("HISFCOS FPN: input too small for the stride-128 level")
    if (s4.H[0], s4.W[0]) != (2 * h5, 2 * w5) or (s3.H[0], s3.W[0]) != (4 * h5, 4 * w5):
        raise FdError("HISFCOS FPN: H and W must be multiples of 32 (x2 upsample-adds must line up, HISFcos.py:155-165)")
    pyr_segs = Segs.make(B, hw)
    pyr = pool.get(pyr_segs.rows, F)

    lv = [Rows(pyr.buf[pyr_segs.m_start[i]:pyr_segs.m_start[i + 1]]) for i in range(5)]

    def level(i: int) -> Rows:
        return lv[i]

    seg = [Segs.make(B, [hw[i]]) for i in range(5)]
    a = pool.get(s5.rows, F)
    add_conv(plan, "fpn.tf1", c5, s5, fpn.tf1, a, bn=fpn.gn1, act=ACT_RELU)
    x4 = pool.get(seg[3].rows, F)
    x5 = pool.get(seg[4].rows, F)
    plan.add("fpn.down_sample1", lambda: ops.maxpool(a, x4, B, hw[2][0], hw[2][1], 2, 2, 0))
    plan.add("fpn.down_sample2", lambda: ops.maxpool(x4, x5, B, hw[3][0], hw[3][1], 2, 2, 0))
    t3 = pool.get(s5.rows, F)
    _his_block(plan, "fpn.HisBlock1", fpn.HisBlock1, a, s5, t3)
    pool.put(a)
    l4 = pool.get(s4.rows, F)
    up_fused = FPN_UP_FUSED and plan.precision == "f32" and all(h % 2 == 0 and w % 2 == 0 for h, w in hw[:2]) and fpn.tf2.weight.shape[1] % 32 == 0 and fpn.tf3.weight.shape[1] % 32 == 0
    if up_fused:       # relu(gn2(tf2(c4))) + Up_sample1(t3) in one launch: the coarser level is read at (i / 2, j / 2) by the lateral's epilogue (HISFcos.py:155-159)
        add_conv(plan, "fpn.tf2+up1_add", c4, s4, fpn.tf2, l4, bn=fpn.gn2, act=ACT_RELU, res=t3, res_up=True)
    else:
        add_conv(plan, "fpn.tf2", c4, s4, fpn.tf2, l4, bn=fpn.gn2, act=ACT_RELU)
        plan.add("fpn.up1_add", lambda: ops.upsample2x_add(t3, l4, l4, B, hw[2][0], hw[2][1]))
    t4 = pool.get(s4.rows, F)
    _his_block(plan, "fpn.HisBlock2", fpn.HisBlock2, l4, s4, t4)
    pool.put(l4)
    l3 = pool.get(s3.rows, F)
    if up_fused:
        add_conv(plan, "fpn.tf3+up2_add", c3, s3, fpn.tf3, l3, bn=fpn.gn2, act=ACT_RELU, res=t4, res_up=True)  # gn2 again: HISFcos.py:163
    else:
        add_conv(plan, "fpn.tf3", c3, s3, fpn.tf3, l3, bn=fpn.gn2, act=ACT_RELU)  # gn2 again: HISFcos.py:163
        plan.add("fpn.up2_add", lambda: ops.upsample2x_add(t4, l3, l3, B, hw[1][0], hw[1][1]))
    _his_block(plan, "fpn.HisBlock3", fpn.HisBlock3, l3, s3, level(0))
    pool.put(l3)
    i4 = pool.get(s4.rows, F)
    plan.add("fpn.down3_add", lambda: ops.maxpool(level(0), i4, B, hw[0][0], hw[0][1], 2, 2, 0, add=t4))
    _his_block(plan, "fpn.HisBlock4", fpn.HisBlock4, i4, s4, level(1))
    pool.put(i4); pool.put(t4)
    i5 = pool.get(s5.rows, F)
    plan.add("fpn.down4_add", lambda: ops.maxpool(level(1), i5, B, hw[1][0], hw[1][1], 2, 2, 0, add=t3))
    _his_block(plan, "fpn.HisBlock5", fpn.HisBlock5, i5, s5, level(2))
    pool.put(i5); pool.put(t3)
    i6 = pool.get(seg[3].rows, F)
    plan.add("fpn.down5_add", lambda: ops.maxpool(level(2), i6, B, hw[2][0], hw[2][1], 2, 2, 0, add=x4))
    _his_block(plan, "fpn.HisBlock6", fpn.HisBlock6, i6, seg[3], level(3))
    pool.put(i6); pool.put(x4)
    i7 = pool.get(seg[4].rows, F)
    plan.add("fpn.down6_add", lambda: ops.maxpool(level(3), i7, B, hw[3][0], hw[3][1], 2, 2, 0, add=x5))
    _his_block(plan, "fpn.HisBlock7", fpn.HisBlock7, i7, seg[4], level(4))
    pool.put(i7); pool.put(x5)
    return pyr, pyr_segs


# ------------------------------------------------------------------------------------------------ heads
def _narrow_predictor(plan: Plan, head, segs: Segs, F: int) -> bool:
    """Does the 1 + 4-wide centre-ness / box predictor of this plan run on the vector-unit kernel (FD_TILE_NARROW)?"""
    rp = head.reg_pred
    rp_pad = rp.dilation[0] * (rp.kernel_size[0] - 1) // 2 if isinstance(rp.padding, str) else rp.padding[0]
    return bool(ops.NARROW and plan.precision in ("f32", "mixed") and ops.narrow_ok(F, 5, rp.kernel_size[0], rp.stride[0], rp_pad, rp.dilation[0])
                and ops.narrow_tiles(segs) >= ops.NARROW_MIN_TILES)


def _out_convs(plan: Plan, head, tower: Rows, segs: Segs, F: int, ncls: int, reg_gate=None):
    """cls_logits on tower[:, :F]; cnt_logits + reg_pred (+ScaleExp) on tower[:, F:] as one 5-wide conv.  reg_gate = (a, b, act): tower[:, F:] is
    still the RAW tower output and the narrow predictor applies its GroupNorm affine + activation in its loader (build_his_head)."""
    dev, pool = plan.device, plan.pool
    M = segs.rows
    cls = pool.get(M, ncls if ncls % 4 == 0 else ncls + (4 - ncls % 4))
    add_conv(plan, "head.cls_logits", tower.slice(0, F), segs, head.cls_logits, cls.slice(0, ncls))
    cr = pool.get(M, 8)
    w = torch.cat([head.cnt_logits.weight.detach(), head.reg_pred.weight.detach()], 0)
    b = torch.cat([head.cnt_logits.bias.detach(), head.reg_pred.bias.detach()], 0)
    scales = [float(s.scale.detach().reshape(-1)[0]) for s in head.scale_exp][:segs.nseg]
    rp = head.reg_pred
    rp_pad = rp.dilation[0] * (rp.kernel_size[0] - 1) // 2 if isinstance(rp.padding, str) else rp.padding[0]
    narrow = _narrow_predictor(plan, head, segs, F)
    if reg_gate is not None and not narrow:
        raise FdError("_out_convs: a GroupNorm folded into the box predictor's loader needs the narrow kernel")
    if not narrow and plan.winograd and ops.wino_ok(F, 8, rp.kernel_size[0], rp.stride[0], rp_pad, rp.dilation[0]) and ops.wino_choice(segs, F, 8, rp.dilation[0])[0]:
        # the Winograd kernel writes whole float4s: three zero filters fill the 8-wide buffer (channels 5..7 hold exp(0) = 1, never read);
        # 0.27 -> 0.17 ms against the direct kernel's 128 x 32 tile at Cout = 5
        w = torch.cat([w, torch.zeros(3, *w.shape[1:], dtype=w.dtype, device=w.device)], 0)
        b = torch.cat([b, torch.zeros(3, dtype=b.dtype, device=b.device)], 0)
        add_conv(plan, "head.cnt_reg", tower.slice(F, F), segs, head.reg_pred, cr.slice(0, 8), weight=w, bias=b, Cout=8,
                 act=ACT_EXP, act_c0=1, seg_param=scales)
        plan.flops -= 2 * segs.rows * 3 * F * 9          # (count the 5 real filters only)
        plan.step_flops[len(plan.steps) - 1] -= 2 * segs.rows * 3 * F * 9
    else:
        g = {} if reg_gate is None else {"gate": reg_gate[0], "gate_b": reg_gate[1], "gate_act": reg_gate[2]}
        add_conv(plan, "head.cnt_reg", tower.slice(F, F), segs, head.reg_pred, cr.slice(0, 5), weight=w, bias=b, Cout=5,
                 act=ACT_EXP, act_c0=1, seg_param=scales, **g)
    return cls.slice(0, ncls), cr.slice(0, 1), cr.slice(1, 4)


def _flush_tail_steps(plan: Plan) -> None:
    """The tail launches add_conv set aside (TOWER_TAIL_SPLIT) become steps of their own -- called right AFTER the caller closed its mark, so that the mark
    (= what TwoLanePipeline keeps exclusive and bench.py times as the roofline launch) covers the whole rounds only."""
    for nm, fn, fl, info in getattr(plan, "tail_steps", []):
        plan.add(nm, fn)
        plan.step_flops[len(plan.steps) - 1] = fl
        plan.step_info[len(plan.steps) - 1] = info
        plan.tiles[nm] = _lib.WINO4_TILE
    plan.tail_steps = []


def _fused_gn(plan: Plan, name: str, x: Rows, segs: Segs, gns, act: int) -> None:
    """k GroupNorm(32, F) over k adjacent F-channel slices == one GroupNorm(32k, kF) with concatenated affine."""
    dev = plan.device
    gamma = torch.cat([_dev(g.weight, dev) for g in gns]).contiguous()
    beta = torch.cat([_dev(g.bias, dev) for g in gns]).contiguous()
    G = sum(g.num_groups for g in gns)
    ws = ops.groupnorm_workspace(segs, G, dev)
    eps = gns[0].eps
    plan.keep += [gamma, beta, ws]
    plan.add(name, lambda: ops.groupnorm_act(x, gamma, beta, x, segs, G, act, ws, eps))


def _gn_fusable(plan: Plan, gns, Cc: int) -> bool:
    """GroupNorm(s) over Cc channels whose statistics the producer's epilogue can emit and fd_groupnorm_from_rowstats reduce."""
    G = sum(g.num_groups for g in gns)
    cg = Cc // G if G and Cc % G == 0 else 0
    return (GN_FUSED and plan.precision in ("f32", "mixed") and all(g.num_channels // g.num_groups == cg for g in gns) and cg in (4, 8, 16, 32)
            and Cc % 32 == 0 and 256 % (2 * G) == 0 and Cc <= 1024 and 256 % (Cc // 4) == 0 and ((Cc // 4) % 64 == 0 or 64 % (Cc // 4) == 0))


def _gn_from_rowstats(plan: Plan, name: str, rgs, segs: Segs, gns, Cc: int, want_coef: bool):
    """The reduction step behind a producer that left row-group sums in `rgs`: -> (workspace, gamma, beta, G, eps, coef | None)."""
    dev = plan.device
    gamma = torch.cat([_dev(g.weight, dev) for g in gns]).contiguous()
    beta = torch.cat([_dev(g.bias, dev) for g in gns]).contiguous()
    G, eps = sum(g.num_groups for g in gns), gns[0].eps
    ws = ops.groupnorm_workspace(segs, G, dev)
    coef = torch.empty(segs.nseg * segs.batch, 2, Cc, dtype=torch.float32, device=dev) if want_coef else None
    plan.keep += [gamma, beta, ws, coef]
    plan.add(name, lambda: ops.groupnorm_from_rowstats(rgs.buf, Cc, G, eps, gamma, beta, segs, ws, coef))
    return ws, gamma, beta, G, eps, coef


def build_his_head(plan: Plan, head, pyr: Rows, segs: Segs):
    """HISFCOSHead.forward (HISFcos.py:211-229), all levels in one launch per layer.

    GroupNorm is folded into its neighbours (FD_GN_FUSED=1, exact-fp32 plans): the producing conv's epilogue leaves per-row group sums
    (fd_conv_params.gn_stats), a small reduction turns them into per-(level, image) statistics, and the CONSUMER normalises on its way in
    -- dw1 reads ReLU(GN1(.)) in its window loads, pw2 reads SiLU(GN2(.)) in its operand loader -- so the pre-block is
    pw1 | dw1 | pw2 over HBM (4 passes over a 512-channel map instead of 10).  The tower's GroupNorm keeps its one normalise pass
    (its consumers are the Winograd predictors) but loses its statistics pass."""
    dev, pool = plan.device, plan.pool
    M = segs.rows
    F = head.pw1.weight.shape[1]
    ncls = head.cls_logits.weight.shape[0]
    wd = ops.pack_dw_weight(_dev(head.dw1.weight, dev))
    plan.keep.append(wd)
    pre_fused = _gn_fusable(plan, [head.gn1], 2 * F) and _gn_fusable(plan, [head.gn2], 2 * F) and head.dw1.bias is None
    h1 = pool.get(M, 2 * F)
    if pre_fused:
        G1, G2 = head.gn1.num_groups, head.gn2.num_groups
        rgs1 = pool.get(M, 2 * G1)
        add_conv(plan, "head.pw1", pyr, segs, head.pw1, h1, gn_stats=rgs1.buf, gn_groups=G1)
        coef1 = _gn_from_rowstats(plan, "head.gn1.stats", rgs1, segs, [head.gn1], 2 * F, True)[5]
        pool.put(rgs1)
        h2 = pool.get(M, 2 * F)
        rgs2 = pool.get(M, 2 * G2)
        plan.add("head.dw1", lambda: ops.dwconv3x3_gn(h1, wd, h2, segs, coef1, ACT_RELU, rgs2.buf, G2))
        coef2 = _gn_from_rowstats(plan, "head.gn2.stats", rgs2, segs, [head.gn2], 2 * F, True)[5]
        pool.put(rgs2); pool.put(h1)
        z = pool.get(M, F)
        add_conv(plan, "head.pw2", h2, segs, head.pw2, z, res=pyr, gate=coef2[:, 0], gate_b=coef2[:, 1], gate_act=ACT_SILU)
    else:
        add_conv(plan, "head.pw1", pyr, segs, head.pw1, h1)
        _fused_gn(plan, "head.gn1", h1, segs, [head.gn1], ACT_RELU)
        h2 = pool.get(M, 2 * F)
        plan.add("head.dw1", lambda: ops.dwconv3x3(h1, wd, h2, segs, None, None, ACT_NONE))
        _fused_gn(plan, "head.gn2", h2, segs, [head.gn2], ACT_SILU)
        pool.put(h1)
        z = pool.get(M, F)
        add_conv(plan, "head.pw2", h2, segs, head.pw2, z, res=pyr)
    pool.put(h2)
    tower = pool.get(M, 2 * F)
    w = torch.cat([head.cls_conv[0].weight.detach(), head.reg_conv[0].weight.detach()], 0)
    tgn = [head.cls_conv[1], head.reg_conv[1]]
    t_fused = (GN_FUSED_TOWER and _gn_fusable(plan, tgn, 2 * F) and plan.winograd and ops.wino_ok(F, 2 * F, 3, 1, 1, 1) and ops.wino_choice(segs, F, 2 * F, 1)[1] == 1)
    mark = len(plan.steps)
    if t_fused:
        Gt = sum(g.num_groups for g in tgn)
        rgs3 = pool.get(M, 2 * Gt)
        add_conv(plan, "head.tower3x3", z, segs, head.cls_conv[0], tower, weight=w, Cout=2 * F, tag=1, gn_stats=rgs3.buf, gn_groups=Gt)
        plan.marks["head.tower3x3"] = (mark, len(plan.steps))
        _flush_tail_steps(plan)
        reg_in_loader = TOWER_GN_SPLIT and _narrow_predictor(plan, head, segs, F)
        ws, gamma, beta, G, eps, coef = _gn_from_rowstats(plan, "head.tower_gn.stats", rgs3, segs, tgn, 2 * F, reg_in_loader)
        pool.put(rgs3)
        if reg_in_loader:
            # the box half of the tower is normalised inside the narrow predictor's patch loader; only the class half (whose consumer is the
            # F(4x4) kernel: no registers for an affine in its loader) keeps a normalise pass -- over half the bytes
            tc = tower.slice(0, F)
            plan.add("head.tower_gn", lambda: ops.coef_apply(tc, coef[:, 0, :F], coef[:, 1, :F], tc, segs, ACT_RELU))
            pool.put(z)
            return _out_convs(plan, head, tower, segs, F, ncls, reg_gate=(coef[:, 0, F:], coef[:, 1, F:], ACT_RELU))
        plan.add("head.tower_gn", lambda: ops.groupnorm_apply(tower, gamma, beta, tower, segs, G, ACT_RELU, ws, eps))
    else:
        add_conv(plan, "head.tower3x3", z, segs, head.cls_conv[0], tower, weight=w, Cout=2 * F, tag=1)
        plan.marks["head.tower3x3"] = (mark, len(plan.steps))
        _flush_tail_steps(plan)
        if TOWER_GN_SPLIT and _narrow_predictor(plan, head, segs, F) and 256 % (F // 4) == 0:
            # statistics of both halves in one pass; the class half keeps a normalise pass (over half the bytes), the box half is normalised inside
            # the narrow predictor's patch loader (fd_conv_params.gate_b on FD_TILE_NARROW): same arithmetic, bit-identical outputs
            gamma = torch.cat([_dev(g.weight, dev) for g in tgn]).contiguous()
            beta = torch.cat([_dev(g.bias, dev) for g in tgn]).contiguous()
            G, eps = sum(g.num_groups for g in tgn), tgn[0].eps
            ws = ops.groupnorm_workspace(segs, G, dev)
            coef = torch.empty(segs.nseg * segs.batch, 2, 2 * F, dtype=torch.float32, device=dev)
            plan.keep += [gamma, beta, ws, coef]
            plan.add("head.tower_gn.stats", lambda: ops.groupnorm_stats(tower, gamma, beta, segs, G, ws, eps, coef))
            tc = tower.slice(0, F)
            plan.add("head.tower_gn", lambda: ops.coef_apply(tc, coef[:, 0, :F], coef[:, 1, :F], tc, segs, ACT_RELU))
            pool.put(z)
            return _out_convs(plan, head, tower, segs, F, ncls, reg_gate=(coef[:, 0, F:], coef[:, 1, F:], ACT_RELU))
        _fused_gn(plan, "head.tower_gn", tower, segs, tgn, ACT_RELU)
    pool.put(z)
    return _out_convs(plan, head, tower, segs, F, ncls)


# ------------------------------------------------------------------------------------------------ FCOS baseline
def build_fcos_fpn(plan: Plan, fpn, feats):
    """FeaturePyramidNetwork.forward (Fcos.py:77-91); the returned P6 is the rectified map (in-place ReLU, Fcos.py:90)."""
    pool = plan.pool
    (c3, s3), (c4, s4), (c5, s5) = feats
    B = s3.batch
    F = fpn.P5.weight.shape[0]
    h5, w5 = s5.H[0], s5.W[0]
    h6, w6 = (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1
    h7, w7 = (h6 - 1) // 2 + 1, (w6 - 1) // 2 + 1
    hw = [(s3.H[0], s3.W[0]), (s4.H[0], s4.W[0]), (h5, w5), (h6, w6), (h7, w7)]
    if (s4.H[0], s4.W[0]) != (2 * h5, 2 * w5) or (s3.H[0], s3.W[0]) != (4 * h5, 4 * w5):
        raise FdError("FCOS FPN: H and W must be multiples of 32")
    pyr_segs = Segs.make(B, hw)
    pyr = pool.get(pyr_segs.rows, F)

    lv = [Rows(pyr.buf[pyr_segs.m_start[i]:pyr_segs.m_start[i + 1]]) for i in range(5)]

    def level(i: int) -> Rows:
        return lv[i]

    seg = [Segs.make(B, [hw[i]]) for i in range(5)]
    p5 = pool.get(s5.rows, F)
    add_conv(plan, "FPN.P5", c5, s5, fpn.P5, p5)
    p4 = pool.get(s4.rows, F)
    add_conv(plan, "FPN.P4", c4, s4, fpn.P4, p4)
    plan.add("FPN.P5_Up_add", lambda: ops.upsample2x_add(p5, p4, p4, B, h5, w5))
    add_conv(plan, "FPN.P4_c1", p4, s4, fpn.P4_c1, level(1))
    pool.put(p4)
    p3 = pool.get(s3.rows, F)
    add_conv(plan, "FPN.P3", c3, s3, fpn.P3, p3)
    plan.add("FPN.P4_Up_add", lambda: ops.upsample2x_add(level(1), p3, p3, B, hw[1][0], hw[1][1]))
    add_conv(plan, "FPN.P3_c1", p3, s3, fpn.P3_c1, level(0))
    pool.put(p3)
    add_conv(plan, "FPN.P5_c1", p5, s5, fpn.P5_c1, level(2))
    pool.put(p5)
    add_conv(plan, "FPN.P6_c1", level(2), seg[2], fpn.P6_c1, level(3), act=ACT_RELU)
    add_conv(plan, "FPN.P7_c1", level(3), seg[3], fpn.P7_c1, level(4))
    return pyr, pyr_segs


def build_fcos_head(plan: Plan, head, pyr: Rows, segs: Segs):
    """HeadFCOS.forward (Fcos.py:120-133): 4 x (3x3, GN, ReLU) per branch; layer 0 of both branches is one launch."""
    pool = plan.pool
    M = segs.rows
    F = head.cls_logits.weight.shape[1]
    ncls = head.cls_logits.weight.shape[0]
    cur = pool.get(M, 2 * F)
    w = torch.cat([head.cls_branch[0].weight.detach(), head.reg_branch[0].weight.detach()], 0)
    mark = len(plan.steps)
    add_conv(plan, "head.tower0", pyr, segs, head.cls_branch[0], cur, weight=w, Cout=2 * F)
    plan.marks["head.tower3x3"] = (mark, len(plan.steps))
    _flush_tail_steps(plan)
    _fused_gn(plan, "head.tower0_gn", cur, segs, [head.cls_branch[1], head.reg_branch[1]], ACT_RELU)
    for k in (1, 2, 3):
        nxt = pool.get(M, 2 * F)
        add_conv(plan, f"head.cls_branch.{3 * k}", cur.slice(0, F), segs, head.cls_branch[3 * k], nxt.slice(0, F))
        add_conv(plan, f"head.reg_branch.{3 * k}", cur.slice(F, F), segs, head.reg_branch[3 * k], nxt.slice(F, F))
        _fused_gn(plan, f"head.tower{k}_gn", nxt, segs, [head.cls_branch[3 * k + 1], head.reg_branch[3 * k + 1]], ACT_RELU)
        pool.put(cur)
        cur = nxt
    return _out_convs(plan, head, cur, segs, F, ncls)


# ------------------------------------------------------------------------------------------------ MNFCOS
def add_mn_block(plan: Plan, name: str, blk, x: Rows, segs: Segs, out: Rows) -> None:
    """MNBlock.forward (modules.py:209-216): out = x + PW2(SiLU(PW1(BN(dilated depthwise k x k (x))))) -- the depthwise conv with its
    frozen BatchNorm folded in (fd_dwconv_dilated_nhwc, 'same' padding: the repaired behaviour for k = 5 / 7), the two 1x1 convs on
    the MFMA kernel with SiLU and the residual add in their epilogues.  Works on one level or a whole pyramid (shared weights)."""
    dev, pool = plan.device, plan.pool
    dw = blk.DilatedDepthWiseConv
    C, K, dil = dw.weight.shape[0], dw.kernel_size[0], dw.dilation[0]
    wd = ops.pack_dwk_weight(_dev(dw.weight, dev))
    sc, sf = ops.fold_bn(_dev(blk.BN.weight, dev), _dev(blk.BN.bias, dev), _dev(blk.BN.running_mean, dev), _dev(blk.BN.running_var, dev),
                         blk.BN.eps)
    plan.keep += [wd, sc, sf]
    t = pool.get(segs.rows, C)
    plan.add(name + ".DilatedDepthWiseConv", lambda: ops.dwconv_dilated(x, wd, t, segs, K, dil, sc, sf, ACT_NONE))
    h = pool.get(segs.rows, blk.PW1.weight.shape[0])
    add_conv(plan, name + ".PW1", t, segs, blk.PW1, h, act=ACT_SILU)
    pool.put(t)
    add_conv(plan, name + ".PW2", h, segs, blk.PW2, out, res=x)
    pool.put(h)


def build_mn_fpn(plan: Plan, fpn, feats):
    """LieghtWeightFeaturePyramid_old.forward (MNFcos.py:239-256): 1x1 laterals (+bias), MNBlocks, nearest x2 upsample + add, 2x2 max-pool."""
    pool = plan.pool
    (c3, s3), (c4, s4), (c5, s5) = feats
    B = s3.batch
    F = fpn.C5PW.weight.shape[0]
    h5, w5 = s5.H[0], s5.W[0]
    hw = [(s3.H[0], s3.W[0]), (s4.H[0], s4.W[0]), (h5, w5), (h5 // 2, w5 // 2), (h5 // 4, w5 // 4)]
    if hw[1] != (2 * h5, 2 * w5) or hw[0] != (4 * h5, 4 * w5) or hw[4][0] < 1 or hw[4][1] < 1:
        raise FdError("MNFCOS FPN: H and W must be multiples of 32 and at least 128")
    pyr_segs = Segs.make(B, hw)
    pyr = pool.get(pyr_segs.rows, F)
    lv = [Rows(pyr.buf[pyr_segs.m_start[i]:pyr_segs.m_start[i + 1]]) for i in range(5)]
    seg = [Segs.make(B, [hw[i]]) for i in range(5)]
    l5 = pool.get(s5.rows, F)
    add_conv(plan, "FeaturePyramidNetwork.C5PW", c5, s5, fpn.C5PW, l5)
    add_mn_block(plan, "FeaturePyramidNetwork.MNB5", fpn.MNB5, l5, seg[2], lv[2])
    pool.put(l5)
    l4 = pool.get(s4.rows, F)
    add_conv(plan, "FeaturePyramidNetwork.C4PW", c4, s4, fpn.C4PW, l4)
    plan.add("FeaturePyramidNetwork.up1_add", lambda: ops.upsample2x_add(lv[2], l4, l4, B, hw[2][0], hw[2][1]))
    add_mn_block(plan, "FeaturePyramidNetwork.MNB4", fpn.MNB4, l4, seg[1], lv[1])
    pool.put(l4)
    l3 = pool.get(s3.rows, F)
    add_conv(plan, "FeaturePyramidNetwork.C3PW", c3, s3, fpn.C3PW, l3)
    plan.add("FeaturePyramidNetwork.up2_add", lambda: ops.upsample2x_add(lv[1], l3, l3, B, hw[1][0], hw[1][1]))
    add_mn_block(plan, "FeaturePyramidNetwork.MNB3", fpn.MNB3, l3, seg[0], lv[0])
    pool.put(l3)
    d6 = pool.get(seg[3].rows, F)
    plan.add("FeaturePyramidNetwork.DownSample_1", lambda: ops.maxpool(lv[2], d6, B, hw[2][0], hw[2][1], 2, 2, 0))
    add_mn_block(plan, "FeaturePyramidNetwork.MNB6", fpn.MNB6, d6, seg[3], lv[3])
    pool.put(d6)
    d7 = pool.get(seg[4].rows, F)
    plan.add("FeaturePyramidNetwork.DownSample_2", lambda: ops.maxpool(lv[3], d7, B, hw[3][0], hw[3][1], 2, 2, 0))
    add_mn_block(plan, "FeaturePyramidNetwork.MNB7", fpn.MNB7, d7, seg[4], lv[4])
    pool.put(d7)
    return pyr, pyr_segs


def build_mn_head(plan: Plan, head, pyr: Rows, segs: Segs):
    """MNHeadFCOS.forward (MNFcos.py:285-297) over the whole pyramid per launch (the weights are shared over levels): two MNBlocks, the
    cls / reg 3x3 convs as one 2F-wide conv + one GroupNorm(64, 2F) + SiLU, the 1x1 predictors (cnt + reg merged, ScaleExp fused)."""
    pool = plan.pool
    M = segs.rows
    F = head.cls_logits.weight.shape[1]
    ncls = head.cls_logits.weight.shape[0]
    b1 = pool.get(M, F)
    add_mn_block(plan, "head.block1", head.block1, pyr, segs, b1)
    b2 = pool.get(M, F)
    add_mn_block(plan, "head.block2", head.block2, b1, segs, b2)
    pool.put(b1)
    tower = pool.get(M, 2 * F)
    w = torch.cat([head.cls_conv[0].weight.detach(), head.reg_conv[0].weight.detach()], 0)
    mark = len(plan.steps)
    add_conv(plan, "head.tower3x3", b2, segs, head.cls_conv[0], tower, weight=w, Cout=2 * F, tag=1)
    plan.marks["head.tower3x3"] = (mark, len(plan.steps))
    _flush_tail_steps(plan)
    _fused_gn(plan, "head.tower_gn", tower, segs, [head.cls_conv[1], head.reg_conv[1]], ACT_SILU)
    pool.put(b2)
    return _out_convs(plan, head, tower, segs, F, ncls)


# ------------------------------------------------------------------------------------------------ post-processing inside the plan
def add_postprocess(plan: Plan, outs, segs: Segs, strides, score_thr: float, iou_thr: float, max_box: int, img_hw, clip: bool = True):
    """FCOSHead (head.py:52-102: decode, top-k, score threshold, per-class NMS) + ClipBoxes (head.py:152-162) as plan steps on preallocated
    buffers -- no Python-side allocation between the launches, so the whole detection (model + post-process) is one launch sequence that
    Plan.capture_graph turns into ONE graph launch (the latency path).  -> (scores [B,K], classes [B,K] int64, boxes [B,K,4], counts [B] int32),
    padded like FCOSHead.detect_padded; the tensors are plan-owned and overwritten by the next run."""
    import ctypes as C
    dev = plan.device
    cls, cnt, reg = outs
    nlev = min(len(strides), segs.nseg)
    if nlev != segs.nseg:
        raise FdError("add_postprocess: one stride per pyramid level expected")
    N = segs.batch
    L = sum(h * w for h, w in segs.level_hw())
    K = min(int(max_box), L)
    lib = _lib.lib()
    st = (C.c_int32 * nlev)(*[int(v) for v in strides][:nlev])
    scores = torch.empty(N, L, dtype=torch.float32, device=dev)
    classes = torch.empty(N, L, dtype=torch.int32, device=dev)
    boxes = torch.empty(N, L, 4, dtype=torch.float32, device=dev)
    ts = torch.empty(N, K, dtype=torch.float32, device=dev)
    tc = torch.empty(N, K, dtype=torch.int64, device=dev)
    tb = torch.empty(N, K, 4, dtype=torch.float32, device=dev)
    os_ = torch.empty(N, K, dtype=torch.float32, device=dev)
    oc = torch.empty(N, K, dtype=torch.int64, device=dev)
    ob = torch.empty(N, K, 4, dtype=torch.float32, device=dev)
    keep = torch.empty(N, K, dtype=torch.int32, device=dev)
    counts = torch.empty(N, dtype=torch.int32, device=dev)
    nb_t, nb_n = lib.fd_topk_workspace_bytes(N, L, K), lib.fd_nms_workspace_bytes(N, K)
    if nb_t < 0 or nb_n < 0:
        raise FdError(f"add_postprocess: unsupported N={N} L={L} K={K}")
    ws_t = torch.empty(max(nb_t // 8, 1), dtype=torch.int64, device=dev)
    ws_n = torch.empty(max(nb_n // 8, 1), dtype=torch.int64, device=dev)
    plan.keep += [scores, classes, boxes, ts, tc, tb, os_, oc, ob, keep, counts, ws_t, ws_n, st, segs]
    ncls = cls.C
    H, W = img_hw

    def decode():
        ops.check(lib.fd_fcos_decode(cls.ptr, cls.cs, cls.co, cnt.ptr, cnt.cs, cnt.co, reg.ptr, reg.cs, reg.co, ncls, C.byref(segs), st,
                                     scores.data_ptr(), classes.data_ptr(), boxes.data_ptr(), ops._stream()), "fd_fcos_decode")

    def topk():
        ops.check(lib.fd_fcos_topk(scores.data_ptr(), classes.data_ptr(), boxes.data_ptr(), N, L, K, ts.data_ptr(), tc.data_ptr(), tb.data_ptr(),
                                   None, ws_t.data_ptr() if nb_t > 0 else None, ops._stream()), "fd_fcos_topk")

    def nms():
        ops.check(lib.fd_batched_nms(ts.data_ptr(), tc.data_ptr(), tb.data_ptr(), N, K, float(score_thr), float(iou_thr), os_.data_ptr(), oc.data_ptr(),
                                     ob.data_ptr(), keep.data_ptr(), counts.data_ptr(), ws_n.data_ptr(), ops._stream()), "fd_batched_nms")

    plan.add("post.decode", decode)
    plan.add("post.topk", topk)
    plan.add("post.nms", nms)
    if clip:
        plan.add("post.clip", lambda: ops.check(lib.fd_clip_boxes(ob.data_ptr(), N * K, int(H), int(W), ops._stream()), "fd_clip_boxes"))
    plan.dets = (os_, oc, ob, counts)
    return plan.dets


# ------------------------------------------------------------------------------------------------ views
def level_views(rows: Rows, segs: Segs) -> List[torch.Tensor]:
    """Per-level NCHW-shaped tensors (channels-last memory: zero-copy views of the pyramid rows buffer)."""
    outs = []
    t = rows.tensor()
    for i in range(segs.nseg):
        h, w = segs.H[i], segs.W[i]
        outs.append(t[segs.m_start[i]:segs.m_start[i + 1]].view(segs.batch, h, w, rows.C).permute(0, 3, 1, 2))
    return outs


def rows_from_nchw(x: torch.Tensor) -> Tuple[Rows, Segs]:
    """NCHW tensor -> NHWC rows (zero-copy when x already is a channels-last view of a rows buffer)."""
    B, Cc, H, W = x.shape
    xp = x.permute(0, 2, 3, 1)
    if not xp.is_contiguous():
        xp = xp.contiguous()
    return Rows(xp.reshape(B * H * W, Cc)), Segs.make(B, [(H, W)])
