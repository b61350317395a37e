import torch, sys, os
sys.path.insert(0, os.getcwd())
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd._lib import Segs
dev="cuda:0"
strides=[8,16,32,64,128]
def timed(fn, reps=50):
    for _ in range(5): out = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3, out
for B in (1, 16, 16, 4):
    segs = Segs.make(B, [(640//s, 640//s) for s in strides])
    g = torch.Generator().manual_seed(3)
    cls = ops.Rows((torch.randn(segs.rows, 80, generator=g)*2-3).to(dev))
    cnt = ops.Rows(torch.randn(segs.rows, 4, generator=g).to(dev), 0, 1)
    reg = ops.Rows((torch.rand(segs.rows, 4, generator=g)*64).to(dev))
    t, (sc, cl, bx) = timed(lambda: ops.fcos_decode(cls, cnt, reg, segs, strides))
    t2, _ = timed(lambda: ops.fcos_topk(sc, cl, bx, 1000))
    print(B, "decode us", round(t,2), "topk us", round(t2,2), "scores", float(sc.min()), float(sc.max()), int(torch.unique(sc).numel()), "nan", int(torch.isnan(sc).sum()))
    # 20-class variant
    cls20 = ops.Rows((torch.randn(segs.rows, 20, generator=g)*2-3).to(dev))
    t3, _ = timed(lambda: ops.fcos_decode(cls20, cnt, reg, segs, strides))
    print("   C=20 decode us", round(t3,2))
