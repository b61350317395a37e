"""Cfg5 diagnostic: FCOS FPN + head + FCOSHead post-process on EfficientNet-B3-shaped endpoints
(48 / 136 / 384 channels at strides 8 / 16 / 32) of a 16-image mixed-aspect batch padded to 832x1344, features resident
in HBM.  The B3 trunk itself is third-party arithmetic (efficientnet_pytorch 0.7.1) and is not part of this number."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd.model.modules.head import ClipBoxes, FCOSHead
from pytorch_object_detection_amd.model.od.Fcos import FeaturePyramidNetwork, HeadFCOS
dev = "cuda:0"
torch.manual_seed(0)
B, H, W = 16, 832, 1344
fpn = FeaturePyramidNetwork([384, 136, 48], 256).eval().to(dev)
head = HeadFCOS(256, 80, 0.01).eval().to(dev)
post, clip = FCOSHead(0.05, 0.6, 1000, [8, 16, 32, 64, 128]), ClipBoxes()
feats = [torch.randn(B, c, H // s, W // s, device=dev) for c, s in ((48, 8), (136, 16), (384, 32))]
img = torch.empty(B, 3, H, W, device=dev)
def step():
    out = head(fpn(feats))
    s, c, b, n = post.detect_padded(out)
    return clip(img, b), n
for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
el = (time.perf_counter() - t) / n
print(f"cfg5 FPN+head+post-process on B3-shaped features, {B} x {H}x{W}: {el * 1e3:.2f} ms/step -> {B / el:.1f} img/s")
