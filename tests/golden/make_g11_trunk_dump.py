"""Parse the layer table the reference RECORDED for its ResNet-50 trunk (a torchinfo summary of HISFCOS at 1x3x512x512 pasted into
/root/reference/Result/proposed, lines 12-184) into tests/golden/g11_trunk_dump.npz.

The trunk's arithmetic is torchvision's (third-party, absent from the reference tree, version unpinned) -- this table is the only
thing the reference itself holds about it: every layer's input / output shape, kernel shape, parameter count and multiply-adds.
The fixture is DATA (integers from a results file); no source text is stored.  Run here only (the reference does not travel):

    python tests/golden/make_g11_trunk_dump.py
"""
import os
import re

import numpy as np

SRC = "/root/reference/Result/proposed"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g11_trunk_dump.npz")
KIND = {"Conv2d": 0, "BatchNorm2d": 1, "ReLU": 2, "MaxPool2d": 3, "Sequential": 4, "Bottleneck": 5}
ROW = re.compile(r"(\w+): (\d+)-(\d+)\s+(\[[\d, ]+\]|--)\s+(\[[\d, ]+\]|--)\s+(\(?[\d,]+\)?|--|\(recursive\))\s+(\[[\d, ]+\]|--)\s+([\d,]+|--)\s*$")


def ints(s, n):
    v = [int(t) for t in re.findall(r"\d+", s)] if s != "--" else []
    return (v + [0] * n)[:n]


def main():
    lines = open(SRC).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if "ResNet50: 1-1" in ln)
    end = next(i for i, ln in enumerate(lines) if i > start and re.search(r"├─\w+: 1-\d+", ln))
    assert (start + 1, end) == (12, 184), (start + 1, end)          # the line range SURVEY / DESIGN cite (1-based, end exclusive + 1)
    kind, depth, in_shape, out_shape, params, frozen, kshape, madds = [], [], [], [], [], [], [], []
    for ln in lines[start + 1:end]:
        m = ROW.search(ln)
        assert m, ln
        name, d, _, i_s, o_s, p, k_s, ma = m.groups()
        kind.append(KIND[name]); depth.append(int(d))
        in_shape.append(ints(i_s, 4)); out_shape.append(ints(o_s, 4))
        params.append(int(re.sub(r"[(),]", "", p)) if p not in ("--", "(recursive)") else 0)
        frozen.append(1 if p.startswith("(") else 0)
        kshape.append(ints(k_s, 4)); madds.append(int(ma.replace(",", "")) if ma != "--" else 0)
    text = "\n".join(lines)
    totals = {
        "hisfcos_total_params_20cls": int(re.search(r"Total params: ([\d,]+)\s*\nTrainable params: 32,378,366", text).group(1).replace(",", "")),
    }
    # totals the reference states in its sources (read as data): trunk 23 508 032 (model/backbone/resnet50.py:45),
    # FPN 7 648 224 and head 1 507 358 at 20 classes (model/od/HISFcos.py:247-248)
    r50 = open("/root/reference/model/backbone/resnet50.py").read()
    his = open("/root/reference/model/od/HISFcos.py").read()
    totals["trunk_params"] = int(re.search(r"Total params: ([\d,]+)", r50).group(1).replace(",", ""))
    totals["fpn_params"] = int(re.search(r"\(([\d,]+)\)\s*\n\s*#\s*8, 207", his).group(1).replace(",", ""))
    totals["head_params_20cls"] = int(re.search(r"8, 207, 372, 496 ([\d,]+)", his).group(1).replace(",", ""))
    np.savez_compressed(OUT, kind=np.array(kind, np.int32), depth=np.array(depth, np.int32), in_shape=np.array(in_shape, np.int64),
                        out_shape=np.array(out_shape, np.int64), params=np.array(params, np.int64), frozen=np.array(frozen, np.int32),
                        kernel_shape=np.array(kshape, np.int64), mult_adds=np.array(madds, np.int64),
                        kind_names=np.array(sorted(KIND, key=KIND.get)),
                        **{k: np.array(v, np.int64) for k, v in totals.items()})
    print(f"{OUT}: {len(kind)} rows, {sum(k == 0 for k in kind)} convs, conv+bn params {sum(p for p, k in zip(params, kind) if k in (0, 1))}, totals {totals}")


if __name__ == "__main__":
    main()
