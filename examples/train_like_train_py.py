#!/usr/bin/env python3
"""The reference's train.py:97-181 loop on the MI355X path, with synthetic VOC-shaped batches.

    python examples/train_like_train_py.py [--steps 20] [--amp]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_like_train_py.py   # DDP / RCCL

Same objects and call order as the reference: HalfInvertedStageFCOS([512, 1024, 2048], 20, 256), FCOSGenTargets,
FCOSLoss('giou'), SGD, DistributedDataParallel(find_unused_parameters=True), autocast + GradScaler, linear warm-up.
In model.train() the forward is an autograd graph of HIP kernels (train_ops.py); target assignment and the losses are HIP
kernels; under autocast the dense convolutions run on the f16 MFMA with fp32 accumulation (forward, data and weight gradients:
the reference's AMP arithmetic), normalisation and losses in fp32; under DDP the FPN's BatchNorms are SyncBatchNorm on the HIP statistics path.
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd.model.loss import FCOSLoss  # noqa: E402
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets  # noqa: E402
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--amp", action="store_true")
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    torch.manual_seed(0)
    model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev)                               # train.py:97
    gen_target = FCOSGenTargets(strides=[8, 16, 32, 64, 128],
                                limit_range=[[-1, 64], [64, 128], [128, 256], [256, 512], [512, 999999]])   # train.py:98
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)       # train.py:101
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)                                 # train.py:103
    LR_INIT, WARMUP_STEPS = 1e-3, 501
    optimizer = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=LR_INIT, momentum=0.9, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda", enabled=args.amp)                                         # train.py:133
    criterion = FCOSLoss("giou")

    g = torch.Generator(device=dev).manual_seed(100 + rank)      # synthetic batches are drawn on the device (no dataloader here)
    model.train()
    t0 = None
    for step in range(1, args.steps + 1):
        imgs = torch.randn(args.batch, 3, 512, 512, generator=g, device=dev)                            # collate_fn output shape
        c = torch.rand(args.batch, 8, 2, generator=g, device=dev) * 400 + 50
        s = torch.rand(args.batch, 8, 2, generator=g, device=dev) * 150 + 20
        targets = torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511)
        classes = torch.randint(1, 21, (args.batch, 8), generator=g, device=dev)
        if step < WARMUP_STEPS:                                                                     # train.py:161-164
            for group in optimizer.param_groups:
                group["lr"] = float(step / WARMUP_STEPS * LR_INIT)
        optimizer.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):                         # train.py:175-179
            outputs = model(imgs)
            target = gen_target([outputs, targets, classes])
            losses = criterion([outputs, target])
            loss = losses[-1]
        scaler.scale(loss.mean()).backward()
        scaler.step(optimizer)
        scaler.update()
        if step == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if rank == 0 and (step % 5 == 0 or step == 1):
            print(f"step {step:4d}  cls {float(losses[0]):.4f}  cnt {float(losses[1]):.4f}  reg {float(losses[2]):.4f}  "
                  f"total {float(losses[3]):.4f}  mem {torch.cuda.memory_reserved() / 1e9:.2f} GB")
    torch.cuda.synchronize()
    if rank == 0 and t0 is not None and args.steps > 3:
        dt = (time.perf_counter() - t0) / (args.steps - 3)
        print(f"{dt * 1e3:.1f} ms/step, {args.batch * world / dt:.1f} img/s over {world} GPU(s)")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
