"""Builder(cfg).model_build() / .opt_build(model): the factory the reference's scripts go through (bulider.py:10-43,
module name spelled as there), written table-driven.  Behaviour differs from the reference where the reference is
broken: 'HISFCOS' resolves (the reference looks up a module `od.proposed` that does not exist), an unknown model or
optimizer raises NotImplementedError (the reference raises the NotImplemented *constant*), and the Adam family gets
weight_decay as weight_decay (the reference passes it positionally into `betas`).  Only trainable parameters are handed
to the optimizer (frozen BatchNorm / stage-1 weights carry requires_grad=False, HISFcos.py:57-68)."""
from __future__ import annotations

from typing import Callable, Dict

import torch
import torch.nn as nn

from .utill.utills import load_config  # noqa: F401  (re-exported: `from bulider import Builder, load_config`)


def _detectors() -> Dict[str, Callable[..., nn.Module]]:
    from .model import od
    return {"FCOS": od.Fcos.FCOS, "HISFCOS": od.HISFcos.HalfInvertedStageFCOS, "MNFCOS": od.MNFcos.MNFCOS}


_OPTIMIZERS: Dict[str, Callable[..., torch.optim.Optimizer]] = {
    "SGD": lambda params, o: torch.optim.SGD(params, lr=o["lr"], momentum=o["momentum"], weight_decay=o["weight_decay"]),
    "Adam": lambda params, o: torch.optim.Adam(params, lr=o["lr"], weight_decay=o["weight_decay"]),
    "AdamW": lambda params, o: torch.optim.AdamW(params, lr=o["lr"], weight_decay=o["weight_decay"]),
    "RAdam": lambda params, o: torch.optim.RAdam(params, lr=o["lr"], weight_decay=o["weight_decay"]),
}


class Builder:
    """cfg: the dict load_config() returns (keys 'model', 'dataset_setting', and one block per detector name)."""

    def __init__(self, cfg: dict):
        self.config = cfg
        self.model = cfg["model"]["name"]

    def _block(self) -> dict:
        try:
            return self.config[self.model]
        except KeyError:
            raise NotImplementedError(f"no '{self.model}' block in the dataset config") from None

    def model_build(self) -> nn.Module:
        block = self._block()
        ctor = _detectors().get(self.model)
        if ctor is None:
            raise NotImplementedError(f"model '{self.model}' is outside the MI355X hot path (FCOS, HISFCOS, MNFCOS)")
        return ctor(block["CannelofBackbone"], self.config["dataset_setting"]["class_num"], block["channel"])

    def opt_build(self, model: nn.Module) -> torch.optim.Optimizer:
        opt = self._block()["optimizer"]
        make = _OPTIMIZERS.get(opt["name"])
        if make is None:
            raise NotImplementedError(f"optimizer '{opt['name']}'")
        return make([p for p in model.parameters() if p.requires_grad], opt)
