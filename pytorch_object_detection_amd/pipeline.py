"""Two batches in flight on two HIP streams ("lanes"), for throughput.

A detector forward is ~190 dependent launches; every launch ends in a tail during which part of the chip idles (mid-size
layers fill 256 CUs for one or two rounds of workgroups), and the next launch of the same batch cannot start before it.
A second, independent batch on another stream fills those tails: consecutive `submit()` calls alternate between two plan
instances (own activation buffers) on two streams.

The ONE kernel that gains nothing from company is the head tower (4 264 full tiles, MFMA-bound): it is kept exclusive, which
also keeps its HIP-event / rocprof duration a clean roofline measurement.  The interleaving is static, by cross-stream events:
a step is cut into  P1 | P2 | T | S3  (trunk + FPN + head pre-block in two parts, the tower launch, everything after it incl.
post-processing), with time(S3) + time(P1) ~ time(P2); per submitted step s on lane L (other lane O):

    L: P1(s)                      O: wait P1(s) ; T(s-1) ; S3(s-1)      <- T(s-1) runs alone: L is parked behind it
    L: wait T(s-1) ; P2(s)        ... which overlaps O's S3(s-1) and, from the next submit, O's P1(s+1)

Results of step s are complete once `submit` of step s+1 (or `drain()`) has been enqueued; every step's outputs live in its
lane's buffers until that lane's next step starts.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from ._lib import FdError


class TwoLanePipeline:
    def __init__(self, model, post: Callable, mark: str = "head.tower3x3", pair_tuned: bool = False):
        """model: a detector with plan_for(x, slot=); post(out, x, tag) -> result (enqueued right after the model on the lane's
        stream; `tag` is whatever submit() was given for that step)."""
        self.model, self.post, self.mark, self.pair_tuned = model, post, mark, pair_tuned
        self.streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        self.plans: List[Optional[object]] = [None, None]
        self.cut = None
        self.tick = 0
        self.pending = None          # (lane, x, tower_events, tag) of the step whose T / S3 are still to be enqueued
        self.last_result = None
        self.last_done = None        # event recorded behind the post-processing of the last finished step (on its lane)

    def _setup(self, x: torch.Tensor) -> None:
        # own plan instances (slots 1, 2; slot 0 stays the plain single-stream plan).  pair_tuned: take the block tiles from
        # the "pair|" entries of the tuning table (timed beside a twin launch on a second stream, ops.autotune_conv(pair=True));
        # measured on MI355X they do not beat the tiles timed alone (881 vs 900 img/s), so the default is off
        self.model._plan_pair_tuned = self.pair_tuned
        try:
            self.plans = [self.model.plan_for(x, slot=k) for k in (1, 2)]
        finally:
            self.model._plan_pair_tuned = False
        p = self.plans[0]
        if self.mark not in p.marks:
            raise FdError(f"plan has no '{self.mark}' mark")
        lo, hi = p.marks[self.mark]
        # per-step device times of one run -> cut the pre-tower part where time(P2) ~ time(P1) + time(S3)
        p.image_ref[0] = x
        p.run()
        n = len(p.steps)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i, st in enumerate(p.steps):
            st()
            ev[i + 1].record()
        torch.cuda.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        pre, s3 = sum(t[:lo]), sum(t[hi:]) + 0.4          # (+ post-processing, ~0.4 ms)
        target, acc, cut = max(0.0, (pre - s3) / 2), 0.0, 0
        while cut < lo and acc + t[cut] <= target:
            acc += t[cut]
            cut += 1
        self.cut, self.lo, self.hi = cut, lo, hi
        self._times = t

    def calibrate(self, x: torch.Tensor, steps: int = 8, shifts=(-0.2, -0.1, 0.0, 0.1, 0.2), rounds: int = 2) -> int:
        """The cut between P1 and P2 comes from single-stream step times; beside another batch the balance moves a little.
        Time `steps` pipelined steps for a few cuts around it (fractions of the pre-tower time), `rounds` times over (a single short timing
        scattered by +- 1.5 % and picked a different cut from run to run), and keep the cut whose best time is lowest."""
        if self.plans[0] is None:
            self._setup(x)
        t, lo = self._times, self.lo
        pre = sum(t[:lo])
        base = sum(t[:self.cut])
        cuts = []
        for sh in shifts:
            target, acc, cut = max(0.0, base + sh * pre), 0.0, 0
            while cut < lo and acc + t[cut] <= target:
                acc += t[cut]
                cut += 1
            if cut not in cuts:
                cuts.append(cut)
        best_of = {c: float("inf") for c in cuts}
        for _ in range(max(1, rounds)):
            for cut in cuts:
                self.cut = cut
                for _ in range(2):
                    self.submit(x)
                self.drain()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    self.submit(x)
                self.drain()
                e1.record()
                e1.synchronize()
                best_of[cut] = min(best_of[cut], e0.elapsed_time(e1) / steps)
        self.cut = min(cuts, key=lambda c: best_of[c])
        self.calibration = dict(best_of)
        return self.cut

    def _finish(self, wait_for: Optional[torch.cuda.Event]):
        """Enqueue T and S3 (+ post) of the pending step on its lane; T waits for `wait_for` (the other lane's P1)."""
        lane, x, tev, tag = self.pending
        plan, st = self.plans[lane], self.streams[lane]
        if wait_for is not None:
            st.wait_event(wait_for)
        with torch.cuda.stream(st):
            if tev is not None:
                tev[0].record()
            plan.run_range(self.lo, self.hi)
            if tev is not None:
                tev[1].record()
            t_end = torch.cuda.Event()
            t_end.record()
            plan.run_range(self.hi, len(plan.steps))
            self.last_result = self.post(self.model.outputs_of(plan), x, tag)
            self.last_done = torch.cuda.Event()
            self.last_done.record()
        self.pending = None
        return t_end

    def _hand_over(self, result):
        """Results are allocated on a lane's stream and consumed on the caller's: order the caller's stream behind the lane that
        produced them and tell the caching allocator about the second stream, so that a block the caller has dropped is not handed
        to the lane's next launch while a kernel on the caller's stream still reads it."""
        cur = torch.cuda.current_stream()
        if getattr(self, "last_done", None) is not None:
            cur.wait_event(self.last_done)

        def walk(o):
            if isinstance(o, torch.Tensor):
                if o.is_cuda:
                    o.record_stream(cur)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
        walk(result)
        return result

    def submit(self, x: torch.Tensor, tower_events=None, tag=None):
        """Enqueue one step (model + post) on the next lane; returns the result of the PREVIOUS step (None at first)."""
        if self.plans[0] is None:
            self._setup(x)
            cur = torch.cuda.current_stream()
            for st in self.streams:
                st.wait_stream(cur)
        lane = self.tick & 1
        self.tick += 1
        plan, st = self.plans[lane], self.streams[lane]
        plan.image_ref[0] = x
        with torch.cuda.stream(st):
            plan.run_range(0, self.cut)
            p1 = torch.cuda.Event()
            p1.record()
        prev = None
        if self.pending is not None:
            t_end = self._finish(p1)
            prev = self._hand_over(self.last_result)
            st.wait_event(t_end)
        with torch.cuda.stream(st):
            plan.run_range(self.cut, self.lo)
        self.pending = (lane, x, tower_events, tag)
        return prev

    def drain(self):
        """Enqueue what is left of the last submitted step, make the caller's stream wait for both lanes; returns its result."""
        if self.pending is not None:
            self._finish(None)
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)
        return self._hand_over(self.last_result)
