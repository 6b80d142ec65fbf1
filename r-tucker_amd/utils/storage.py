"""Histories and checkpoints of a training run -- the surface of ``src/utils/storage.py`` (``Losses``,
``Metric``, ``Metrics``, ``StateDict.save/load``; ``train.py:130-159,252``) with a ``load`` that can read
what ``save`` wrote (the reference's cannot: Appendix A3) and the optimizer / scheduler state included."""
from __future__ import annotations

import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional

import torch


@dataclass
class Losses:
    train: List[float] = field(default_factory=list)
    test: List[float] = field(default_factory=list)
    val: List[float] = field(default_factory=list)
    norms: List[float] = field(default_factory=list)

    def update(self, train_loss=None, train_norm=None, val_loss=None, test_loss=None):
        for hist, x in ((self.train, train_loss), (self.norms, train_norm), (self.val, val_loss), (self.test, test_loss)):
            hist.append(x)

    def merge(self, other: "Losses"):
        for name in ("train", "test", "val", "norms"):
            getattr(self, name).extend(getattr(other, name))


@dataclass
class Metric:
    test: List[float] = field(default_factory=list)
    val: List[float] = field(default_factory=list)

    def __getitem__(self, split):
        return getattr(self, split)


_KEYS = {"mrr": "mrr", "hits_1": "hits@1", "hits_3": "hits@3", "hits_10": "hits@10"}


@dataclass
class Metrics:
    mrr: Metric = field(default_factory=Metric)
    hits_1: Metric = field(default_factory=Metric)
    hits_3: Metric = field(default_factory=Metric)
    hits_10: Metric = field(default_factory=Metric)

    def update(self, metrics_dict: dict, type: str):
        for attr, key in _KEYS.items():
            getattr(self, attr)[type].append(metrics_dict[key])

    def merge(self, other: "Metrics"):
        for attr in _KEYS:
            for split in ("val", "test"):
                getattr(self, attr)[split].extend(getattr(other, attr)[split])


@dataclass
class StateDict:
    model: dict
    losses: Losses
    metrics: Metrics
    last_epoch: int
    optimizer: Optional[dict] = None
    scheduler: Optional[dict] = None

    def save(self, dir, name, add_epoch=True):
        os.makedirs(dir, exist_ok=True)
        path = os.path.join(dir, name + (f"_{self.last_epoch}" if add_epoch else "") + ".pth")
        # plain containers only: loads under torch's weights_only default
        torch.save({"model": self.model, "losses": asdict(self.losses), "metrics": asdict(self.metrics),
                    "last_epoch": self.last_epoch, "scheduler": self.scheduler}, path)
        return path

    @classmethod
    def load(cls, name, **kwargs):
        raw = torch.load(name if name.endswith(".pth") else f"{name}.pth", **kwargs)
        losses = raw["losses"] if isinstance(raw["losses"], Losses) else Losses(**raw["losses"])
        m = raw["metrics"]
        metrics = m if isinstance(m, Metrics) else Metrics(**{k: Metric(**v) for k, v in m.items()})
        return cls(raw["model"], losses, metrics, raw["last_epoch"], raw.get("optimizer"), raw.get("scheduler"))
