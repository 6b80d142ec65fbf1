"""One Riemannian optimizer step -- ``optimizer.fit(loss_fn, x_k); optimizer.step()`` of the reference's training
loop (``train.py:76-85``) -- captured into ONE HIP graph and replayed per batch.

A step at the WN18RR recipe is ~700 small launches (scores + loss + backward at doubled rank in the HIP kernels,
tall-skinny Gram products, a few dozen 200 x 200 float64 factorizations, the truncated HOSVD of a 20 x 400 x 400
core): eager it is bound by the host's launch rate (57 ms in round 2, 110-140 ms with the host eigensolver), and
every step does the same thing at the same addresses with different ids.  So the step is written without a host
synchronisation (``smalllinalg.py``, ``tucker._round_tangent_step``), the optimizer keeps its state in
persistent buffers (``optim._ManifoldOptimizer._keep``), and what changes per batch enters through device
memory: the batch's item ids (one ``copy_`` into a static buffer before the replay), the learning rate and the
regularisation coefficient (device scalars refreshed once per epoch).  The running sums of loss and gradient
norm that ``train_one_epoch`` reports (``train.py:84-85``) are accumulated inside the graph.

``torch.cuda.graphs`` is HIP graphs on ROCm; our C-ABI launches take the capturing stream's handle like any
other kernel.  Before the capture two steps run eagerly (lazy initialisation: code objects, rocBLAS handles,
workspaces, the optimizer's state buffers) -- they are real steps of the epoch, nothing is run twice.
"""
from __future__ import annotations

import os

import torch

from . import tucker as _tucker

ENABLED = os.environ.get("R_TUCKER_AMD_GRAPH", "0") == "1"     # opt-in: see IN_FLIGHT below and DESIGN.md section 8
EAGER_STEPS = 2
# Replays the host may have in flight.  Re-launching this ~700-node graph while its PREVIOUS launch is still executing
# produced NaN parameters on this stack (ROCm 7.2, torch 2.10): within 40 replays at the WN18RR shape with two in
# flight (tools/opt_step_timing.py), after ~2 000 in two training runs -- while the same steps eagerly, replayed with
# one in flight (the host waits for replay n - 1 before it launches replay n), or replayed with torch's own Cholesky
# kernels in place of the one long single-workgroup kernel, gave the eager numbers.  The wait
# costs ~0.4 ms of a 22 ms step.
IN_FLIGHT = int(os.environ.get("R_TUCKER_AMD_GRAPH_IN_FLIGHT", "1"))


class CapturedTrainStep:
    def __init__(self, model, optimizer, flt, batch_size: int, label_smoothing: float, extract_tensor, batch_loss_fn):
        self.model, self.opt, self.flt = model, optimizer, flt
        self.B = int(batch_size)
        self.ls = float(label_smoothing)
        self._extract, self._loss_fn = extract_tensor, batch_loss_fn
        dev = flt.device
        self.dev = dev
        self.ids = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.reg = torch.zeros((), dtype=torch.float32, device=dev)
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        self.gnorm_sum = torch.zeros((), dtype=torch.float32, device=dev)
        self.graph = None
        self.eager_done = 0
        self.replays = 0
        self._events = []

    # one step on the ids currently in self.ids (runs eagerly or under capture: same code)
    def _body(self):
        f = self.flt.features[self.ids]
        loss_fn = self._loss_fn(self.model, f[:, 0].contiguous(), f[:, 1].contiguous(), self.flt, self.ids, self.ls, self.reg)
        x_k = self._extract(self.model)
        gn = self.opt.fit(loss_fn, x_k)
        self.opt.step()
        self.loss_sum += self.opt.loss.detach().to(torch.float32)
        self.gnorm_sum += gn.detach().to(torch.float32)

    def begin_epoch(self, regularization_coeff: float):
        self.reg.fill_(float(regularization_coeff))
        self.opt.refresh_lr()
        self.loss_sum.zero_()
        self.gnorm_sum.zero_()

    def run(self, ids: torch.Tensor):
        if ids.numel() != self.B:
            raise RuntimeError(f"captured step is for batches of {self.B} items, got {ids.numel()}")
        self.ids.copy_(ids)
        if self.graph is not None:
            self._replay()
            return
        if self.eager_done < EAGER_STEPS or not ENABLED or not getattr(self.opt, "capturable", False):
            self._body()
            self.eager_done += 1
            return
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self.graph = g              # (the capture itself executed nothing: replay it for this batch)
        self._replay()

    def _replay(self):
        if IN_FLIGHT > 0 and len(self._events) >= IN_FLIGHT:
            self._events.pop(0).synchronize()
        self.graph.replay()
        self.replays += 1
        if IN_FLIGHT > 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._events.append(ev)

    def drop_graph(self):
        """Forget the capture (the next steps run eagerly, two of them before a new capture if still enabled)."""
        self.graph = None
        self._events = []
        self.eager_done = 0

    def totals(self):
        """(sum of losses, sum of gradient norms) over the steps since ``begin_epoch`` -- one synchronisation."""
        return float(self.loss_sum), float(self.gnorm_sum)
