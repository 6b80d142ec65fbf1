"""Knowledge-graph triples -> integer ids, query lists and filter targets.

Host-side counterpart of the reference's ``src/data/Data.py`` and
``src/data/Dataset.py`` (same constructor arguments, attributes and item
semantics, so ``train.py`` can use them unchanged), re-implemented around numpy
arrays and a CSR (subject, relation) -> objects index so that whole batches of
dense targets are produced at once instead of one python-built row per item.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset


class Data:
    """Reads ``train/valid/test.txt`` (whitespace separated ``s r o`` per line),
    optionally appends the reverse triple ``(o, r + "_reverse", s)`` after each
    split's own triples, and builds sorted entity / relation vocabularies.
    Mirrors ``Data.__init__`` (Data.py:5-19) incl. the relation order
    train-relations + unseen valid + unseen test."""

    def __init__(self, data_dir="data/FB15k-237/", reverse=False):
        for split in ("train", "valid", "test"):
            rows = self.load_data(data_dir, split + ".txt", reverse)
            setattr(self, split + "_data", rows)                       # train_data / valid_data / test_data
            setattr(self, split + "_relations", self.get_relations(rows))
        self.data = self.train_data + self.valid_data + self.test_data
        self.entities = self.get_entities(self.data)
        # relation ids: the train split's relations first, then those only valid / only test introduce
        # (a relation new to both valid and test appears twice, as in the reference's list)
        seen = set(self.train_relations)
        self.relations = list(self.train_relations)
        for extra in (self.valid_relations, self.test_relations):
            self.relations += [x for x in extra if x not in seen]

    @staticmethod
    def load_data(data_dir, file="train.txt", reverse=False):
        with open(os.path.join(data_dir, file), "r") as f:
            rows = [line.split() for line in f.read().strip().split("\n")]
        if reverse:
            rows = rows + [[o, r + "_reverse", s] for s, r, o in rows]
        return rows

    @staticmethod
    def get_relations(data):
        return sorted({d[1] for d in data})

    @staticmethod
    def get_entities(data):
        ents = {d[0] for d in data}
        ents.update(d[2] for d in data)
        return sorted(ents)


class KG_dataset(Dataset):
    """Items are ``(features, targets)``: train mode -> features ``(s, r)`` for each
    distinct pair, targets = label-smoothed multi-hot over objects seen in THIS split;
    ``test_set=True`` -> features ``(s, r, o)`` per triple, targets = multi-hot over
    objects of ``(s, r)`` in ALL splits, no smoothing (Dataset.py:9-53)."""

    def __init__(self, data, dataset, label_smoothing=None, test_set=False):
        self.data = dataset
        self.entity_index = {e: i for i, e in enumerate(data.entities)}
        self.relation_index = {r: i for i, r in enumerate(data.relations)}
        self.data_index = self._ids(dataset)
        self.n_ent, self.n_rel = len(self.entity_index), len(self.relation_index)
        own_pairs, own_ptr, own_obj = self._csr(self.data_index)
        self.entity_relation_pairs = [tuple(p) for p in own_pairs.tolist()]
        if test_set:
            self._pairs, self._ptr, self._obj = self._csr(self._ids(data.data))
        else:
            self._pairs, self._ptr, self._obj = own_pairs, own_ptr, own_obj
        self._pair_slot = {tuple(p): i for i, p in enumerate(self._pairs.tolist())}
        self.size = (len(self.entity_relation_pairs), self.n_ent, self.n_rel)
        self.test_set = test_set
        self.label_smoothing = 0 if test_set else (label_smoothing or 0)
        self._features = (np.asarray(self.data_index, dtype=np.int64).reshape(-1, 3) if test_set
                          else own_pairs.astype(np.int64))

    def _ids(self, triples):
        ei, ri = self.entity_index, self.relation_index
        return [(ei[s], ri[r], ei[o]) for s, r, o in triples]

    @staticmethod
    def _csr(index_triples):
        """distinct (s, r) pairs in first-appearance order (the reference's dict order),
        with their objects in appearance order."""
        t = np.asarray(index_triples, dtype=np.int64).reshape(-1, 3)
        key = t[:, 0] * (t[:, 1].max() + 1 if len(t) else 1) + t[:, 1]
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")          # unique keys by first appearance
        rank_of = np.empty_like(order)
        rank_of[order] = np.arange(len(order))
        slot = rank_of[inv]                                # pair slot of every triple
        by_slot = np.argsort(slot, kind="stable")
        counts = np.bincount(slot, minlength=len(uniq))
        ptr = np.concatenate([[0], np.cumsum(counts)])
        pairs = t[first[order]][:, :2]
        return pairs, ptr, t[by_slot, 2]

    def __len__(self):
        return len(self.data_index) if self.test_set else self.size[0]

    def objects_of(self, s, r):
        i = self._pair_slot[(int(s), int(r))]
        return self._obj[self._ptr[i]:self._ptr[i + 1]]

    @property
    def features(self):
        """All item features as one int64 array: (len, 3) in test mode, (len, 2) in train mode."""
        return self._features

    def dense_targets(self, item_ids, device=None, dtype=torch.float32):
        """Targets of a whole batch at once: (len(item_ids), n_ent)."""
        item_ids = np.asarray(item_ids, dtype=np.int64)
        f = self._features[item_ids]
        slots = np.fromiter((self._pair_slot[(int(s), int(r))] for s, r in f[:, :2]), dtype=np.int64,
                            count=len(f))
        lens = self._ptr[slots + 1] - self._ptr[slots]
        rows = np.repeat(np.arange(len(f)), lens)
        cols = np.concatenate([self._obj[self._ptr[s]:self._ptr[s + 1]] for s in slots]) if len(f) else np.zeros(0, np.int64)
        t = torch.zeros((len(f), self.n_ent), dtype=dtype, device=device)
        if len(rows):
            t[torch.as_tensor(rows, device=device), torch.as_tensor(cols, device=device)] = 1
        if self.label_smoothing > 0:
            t = (1 - self.label_smoothing) * t + self.label_smoothing / self.n_ent
        return t

    def __getitem__(self, idx):
        f = self._features[idx]
        return torch.as_tensor(f), self.dense_targets([idx])[0].float()
