#!/usr/bin/env python3
"""The reference's Test_coco.py:120-190 inference loop on the MI355X path, with synthetic images.

    python examples/infer_like_test_coco.py [--images 4] [--precision f32|f16x3]

Per image (batch 1, as the reference evaluates): uint8 HWC image resized on the host to (min side 800, max side 1333)
and zero padded to a multiple of 32 (dataset/coco.py preprocess) -> model -> FCOSHead(0.05, 0.6, 1000) -> ClipBoxes ->
boxes /= scale, xyxy -> xywh.  Only the host-side resize is numpy; everything from the padded uint8 image on is HIP.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops  # noqa: E402
from pytorch_object_detection_amd.bulider import Builder, load_config  # noqa: E402
from pytorch_object_detection_amd.model.modules.head import ClipBoxes, FCOSHead  # noqa: E402


def resize_pad(img: np.ndarray, min_side=800, max_side=1333):
    """dataset/coco.py preprocess_img_boxes geometry (nearest-neighbour here; the reference uses cv2.resize)."""
    h, w, _ = img.shape
    scale = min_side / min(h, w)
    if max(h, w) * scale > max_side:
        scale = max_side / max(h, w)
    nw, nh = int(scale * w), int(scale * h)
    ys = (np.arange(nh) / scale).astype(int).clip(0, h - 1)
    xs = (np.arange(nw) / scale).astype(int).clip(0, w - 1)
    out = np.zeros((nh + 32 - nh % 32, nw + 32 - nw % 32, 3), np.uint8)
    out[:nh, :nw] = img[ys][:, xs]
    return out, scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = load_config()                      # COCO / HISFCOS, 80 classes
    model = Builder(cfg).model_build().eval().to(dev)
    model.conv_precision = args.precision
    head = FCOSHead(0.05, 0.6, 1000, cfg["HISFCOS"]["stride"])
    clip = ClipBoxes()
    rng = np.random.default_rng(0)
    results = []
    for i in range(args.images):
        h, w = [(480, 640), (640, 480), (427, 640), (500, 375)][i % 4]
        img, scale = resize_pad(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        x = torch.from_numpy(img)[None].to(dev)                      # uint8 [1, H, W, 3]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(x)                                                # normalisation happens in the first kernel
        scores, labels, boxes = head(out)
        boxes = clip(torch.empty(1, 3, img.shape[0], img.shape[1]), boxes.contiguous())
        boxes = ops.boxes_rescale_xywh_(boxes, scale)                # Test_coco.py:147-151
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        for b, s, l in zip(boxes[0].tolist(), scores[0].tolist(), labels[0].tolist()):
            if s < 0.05:
                break
            results.append({"image_id": i, "category_id": int(l), "score": float(s), "bbox": b})
        print(f"image {i}: {h}x{w} -> padded {img.shape[0]}x{img.shape[1]}, {scores.shape[1]} detections, {dt * 1e3:.1f} ms")
    print(f"{len(results)} result records (COCO json layout)")


if __name__ == "__main__":
    main()
