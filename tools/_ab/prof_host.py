import os, sys, cProfile, pstats, io
os.environ["FD_AMP"] = "1"
sys.argv = ["x"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import runpy, torch
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train_host_time.py")).read()
src = src.split("for _ in range(3):\n    step()")[0]
g = {"__name__": "prof", "__file__": os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train_host_time.py")}
exec(compile(src, g["__file__"], "exec"), g)
step = g["step"]
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
