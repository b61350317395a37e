"""A whole training step -- forward, loss, backward, optimizer (and GradScaler under AMP) -- captured ONCE as a HIP graph and replayed.

The reference's loop (train.py:160-190) enqueues every step from Python; on the HIP path one HISFCOS-R50 step is ~1 050 short launches, and under AMP their 16 ms of
kernels sit behind 15 - 20 ms of Python / autograd / ctypes work: the step is bound by the host.  Every node of the HIP training path is capture-safe (no host
synchronisation, no data-dependent shape: tests/test_train_gpu.py asserts the first; weight re-packing is one launch over a cached job table), so the step can be
recorded with torch.cuda.graph and replayed with one call: 20.4 -> 17.1 ms under AMP (tools/train_graph_time.py).

Constraints (those of torch.cuda.graphs): static shapes; inputs are copied into the tensors captured at construction; nothing in `step_fn` may synchronise or read a
tensor on the host; the optimizer must not synchronise either (torch.optim.SGD(fused=True) / capturable Adam; under GradScaler the fused optimizers take found_inf on
the device); a learning-rate schedule must write a TENSOR lr in place (a Python float is baked into the recorded launch).  One process per GPU as everywhere else; a
DistributedDataParallel step is not captured here (its bucket hooks and RCCL launches are left to the eager path).
"""
from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, step_fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor], warmup: int = 3):
        """step_fn(*inputs) runs one full training step and returns a tensor (the loss); it is run `warmup` times eagerly on a side stream (allocator, autotuned
        choices, lazily built tables), then captured.  `example_inputs` become the static input tensors of the graph."""
        if not example_inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("GraphedStep: example_inputs must be CUDA tensors")
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                step_fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = step_fn(*self.static_inputs)

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        """Copy the batch into the captured tensors, replay the step; returns the captured output tensor (overwritten by the next call)."""
        if len(inputs) != len(self.static_inputs):
            raise ValueError(f"GraphedStep: expected {len(self.static_inputs)} inputs")
        for s, t in zip(self.static_inputs, inputs):
            if s.shape != t.shape or s.dtype != t.dtype:
                raise ValueError(f"GraphedStep: input {tuple(t.shape)} {t.dtype} does not match the captured {tuple(s.shape)} {s.dtype}")
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.static_out
