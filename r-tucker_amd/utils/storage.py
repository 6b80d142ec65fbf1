"""Histories and checkpoints of a training run -- the surface of ``src/utils/storage.py`` (``Losses``,
``Metric``, ``Metrics``, ``StateDict.save/load``; ``train.py:130-159,252``) with a ``load`` that can read
what ``save`` wrote (the reference's cannot: Appendix A3) and the scheduler state included.

The three history containers are generated from their field lists: a history is a record of named lists
(one entry per epoch) that can be appended to by keyword and concatenated with another record."""
from __future__ import annotations

import os
from dataclasses import asdict, dataclass, field, fields, is_dataclass, make_dataclass
from typing import List, Optional

import torch

SPLITS = ("test", "val")
METRIC_KEYS = {"mrr": "mrr", "hits_1": "hits@1", "hits_3": "hits@3", "hits_10": "hits@10"}   # attribute -> dict key


def _extend(dst, src):
    """Concatenate every list of the record ``src`` onto the same-named list of ``dst`` (recursively)."""
    for f in fields(dst):
        mine, theirs = getattr(dst, f.name), getattr(src, f.name)
        if is_dataclass(mine):
            _extend(mine, theirs)
        else:
            mine.extend(theirs)


def _record(name, names, factory, **namespace):
    namespace.setdefault("merge", _extend)
    return make_dataclass(name, [(n, List[float] if factory is list else object, field(default_factory=factory))
                                 for n in names], namespace=namespace)


def _losses_update(self, train_loss=None, train_norm=None, val_loss=None, test_loss=None):
    for hist, x in ((self.train, train_loss), (self.norms, train_norm), (self.val, val_loss), (self.test, test_loss)):
        hist.append(x)


def _metrics_update(self, metrics_dict: dict, type: str):
    for attr, key in METRIC_KEYS.items():
        getattr(self, attr)[type].append(metrics_dict[key])


Losses = _record("Losses", ("train",) + SPLITS + ("norms",), list, update=_losses_update)
Metric = _record("Metric", SPLITS, list, __getitem__=lambda self, split: getattr(self, split))
Metrics = _record("Metrics", tuple(METRIC_KEYS), Metric, update=_metrics_update)
for _c in (Losses, Metric, Metrics):        # (picklable under their own names)
    _c.__module__ = __name__


@dataclass
class StateDict:
    model: dict
    losses: "Losses"
    metrics: "Metrics"
    last_epoch: int
    optimizer: Optional[dict] = None
    scheduler: Optional[dict] = None

    def save(self, dir, name, add_epoch=True):
        os.makedirs(dir, exist_ok=True)
        stem = f"{name}_{self.last_epoch}" if add_epoch else name
        path = os.path.join(dir, stem + ".pth")
        # plain containers only: loads under torch's weights_only default
        payload = dict(model=self.model, losses=asdict(self.losses), metrics=asdict(self.metrics),
                       last_epoch=self.last_epoch, scheduler=self.scheduler)
        torch.save(payload, path)
        return path

    @classmethod
    def load(cls, name, **kwargs):
        raw = torch.load(name if name.endswith(".pth") else name + ".pth", **kwargs)
        losses, metrics = raw["losses"], raw["metrics"]
        if isinstance(losses, dict):
            losses = Losses(**losses)
        if isinstance(metrics, dict):
            metrics = Metrics(**{attr: Metric(**hist) for attr, hist in metrics.items()})
        return cls(raw["model"], losses, metrics, raw["last_epoch"], raw.get("optimizer"), raw.get("scheduler"))
