"""Time the conv tiles of the training step's shapes (forward + data-gradient launches, B=16 512x512, HISFCOS and FCOS)
and merge them into tuned/gfx950_tiles.json.  usage (GPU box): FD_AUTOTUNE=1 python tools/tune_train.py
FD_AMP=1: the same step under torch.autocast(float16) -- times the f16-operand launches ("f16|" keys)."""
import os, sys
os.environ.setdefault("FD_AUTOTUNE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd.model.loss import FCOSLoss
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
from pytorch_object_detection_amd.model.od import FCOS, HalfInvertedStageFCOS
dev = "cuda:0"
B = 16
before = len(ops._tune_table())
for mk in (lambda: HalfInvertedStageFCOS([512, 1024, 2048], 20, 256), lambda: FCOS([2048, 1024, 512], 20, 256)):
    torch.manual_seed(0)
    model = mk().to(dev).train()
    x = torch.randn(B, 3, 512, 512, device=dev)
    c = torch.rand(B, 8, 2, device=dev) * 400 + 50
    s = torch.rand(B, 8, 2, device=dev) * 150 + 20
    gt = torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511)
    labels = torch.randint(1, 21, (B, 8), device=dev)
    gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
    with torch.autocast("cuda", dtype=torch.float16, enabled=os.environ.get("FD_AMP") == "1"):
        out = model(x)
        loss = FCOSLoss("giou")([out, gen([out, gt, labels])])[-1]
    loss.backward()
    torch.cuda.synchronize()
    del model, out
print("table entries:", before, "->", len(ops._tune_table()))
ops.save_tune_table()
import shutil
os.makedirs("gpurun_out", exist_ok=True)
shutil.copy(ops._TUNE_FILE, "gpurun_out/gfx950_tiles.json")
