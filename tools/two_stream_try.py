"""Experiment: splitting the 16-image batch into sub-batch plans on several HIP streams to fill the CU gaps of
mid-size layers (kept as a record)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from pytorch_object_detection_amd.model.modules.head import FCOSHead, ClipBoxes
dev = "cuda:0"
x = torch.randn(16, 3, 640, 640, device=dev)
head = FCOSHead(0.05, 0.6, 1000, [8, 16, 32, 64, 128]); clip = ClipBoxes()
prec = os.environ.get("FD_CONV_PRECISION", "f32")

def make(nchunk, nstream):
    models = [build_model(80, 0).to(dev) for _ in range(nchunk)]
    xs = [c.contiguous() for c in x.chunk(nchunk)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for i, (m, xi) in enumerate(zip(models, xs)):
            st = streams[i % nstream]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                s, c, b, n = head.detect_padded(m(xi))
                outs.append((s, c, clip(xi, b), n))
        for st in streams:
            cur.wait_stream(st)
        return [torch.cat([o[k] for o in outs]) for k in range(4)]
    return run

def base():
    m = build_model(80, 0).to(dev)
    def run():
        s, c, b, n = head.detect_padded(m(x))
        return s, c, clip(x, b), n
    return run

def pipelined(nslot):
    """successive B=16 steps alternate between `nslot` plan instances on their own streams (two batches in flight)."""
    models = [build_model(80, 0).to(dev) for _ in range(nslot)]
    streams = [torch.cuda.Stream() for _ in range(nslot)]
    state = {"i": 0, "last": None}
    def run():
        k = state["i"] % nslot
        state["i"] += 1
        cur = torch.cuda.current_stream()
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            s, c, b, n = head.detect_padded(models[k](x))
            out = (s, c, clip(x, b), n)
        state["last"] = out
        return out
    def finish():
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    run.finish = finish
    return run

for name, fn in (("1 plan B=16", base()), ("pipelined B=16 x 2 slots", pipelined(2)), ("pipelined B=16 x 3 slots", pipelined(3)), ("2 x B=8, 2 streams", make(2, 2)), ("4 x B=4, 2 streams", make(4, 2)),
                 ("4 x B=4, 4 streams", make(4, 4)), ("2 x B=8, 1 stream", make(2, 1))):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        r = fn()
    if hasattr(fn, "finish"):
        fn.finish()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t) / 10
    print(f"[{prec}] {name}: {el*1e3:.2f} ms/16 images -> {16/el:.1f} img/s (model + head + clip); kept {r[3][:3].tolist()}")
