"""EXPERIMENT, not part of the product package (moved out of r-tucker_amd/ in round 4).

One Riemannian optimizer step -- ``optimizer.fit(loss_fn, x_k); optimizer.step()`` of the reference's training
loop (``train.py:76-85``) -- captured into ONE HIP graph and replayed per batch.  Opt in from a tool with
``graphstep.install()`` (it replaces ``r_tucker_amd.driver.TRAIN_STEP_FACTORY``); ``r_tucker_amd.driver`` itself
runs the step eagerly.

Why it is out: on this stack (ROCm 7.2, torch 2.10) ``hipGraphLaunch`` of the ~470-node capture onto a stream that
still has work queued is not ordered behind that work -- replays overlap with the previous replay (or just with the
ids copy queued in front of them): 9 ms steps where the kernels need 15, loss sums off in the 6th digit, NaN
parameters within 40-2000 replays.  The capture is a single chain and the launch is on the capturing stream; an
event recorded behind the previous replay and waited for on the stream does not order it either; launching onto a
DRAINED stream does, and so does ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` (the runtime's path that replays pre-built AQL
packets off): the cause sits in that runtime path, not in the captured work, and the only host-side ordering that
holds is a stream drain per replay -- which makes the replay pointless for a step that is bound by its kernels
(15.0 ms replayed against 15.2 ms eager).  Records: profiles/r03_graph_wait_modes.log, r03_step_graph_topology.txt,
tools/graph_event_probe.py, tools/graph_nan_bisect.py.

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


ENABLED = os.environ.get("R_TUCKER_AMD_GRAPH", "0") == "1"     # opt-in: see WAIT below and DESIGN.md section 8


def install():
    """Make ``r_tucker_amd.driver.train_one_epoch`` build its per-batch step through this module."""
    from r_tucker_amd import driver
    driver.TRAIN_STEP_FACTORY = CapturedTrainStep
EAGER_STEPS = 2
# How the host orders replay n behind what is already queued.  On this stack (ROCm 7.2, torch 2.10) a launch of this
# ~470-node graph onto a stream that still has work queued -- the previous replay, or just the copy of the next batch's
# ids behind it -- was seen to run concurrently with that work: steps of 9 ms where the kernels alone need 15, loss sums
# that differ from the eager ones in the 6th digit and NaN parameters within 40 replays (sometimes only after ~2 000;
# tools/opt_step_timing.py, tools/graph_event_probe.py).  The capture is a single chain (466 nodes,
# 465 edges, one root: ``describe_graph``), waiting for an event recorded after the previous replay is not enough (the
# ids copy is already queued behind it), DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hides it.  Launched onto a DRAINED stream the
# replays are bit-identical to the eager steps, so "stream" -- hipStreamSynchronize after the ids copy, before the launch
# -- is the default; it costs nothing once a step is bound by its kernels (15.0 ms replayed, 15.2 ms eager).
#   stream | device : drain the stream / the device before each launch
#   event           : wait for an event recorded after replay n - IN_FLIGHT (kept for the experiment; unsafe here)
WAIT = os.environ.get("R_TUCKER_AMD_GRAPH_WAIT", "stream")
IN_FLIGHT = int(os.environ.get("R_TUCKER_AMD_GRAPH_IN_FLIGHT", "1"))
DUMP = os.environ.get("R_TUCKER_AMD_GRAPH_DUMP")               # path: write the captured graph's topology there


def describe_graph(raw_graph: int) -> str:
    """Topology of a captured HIP graph (``hipGraphGetNodes`` / ``hipGraphGetEdges`` through ctypes): node count by
    type, edge count, roots, and every node with more than one predecessor or successor.  A capture from one stream
    must be a single chain (edges = nodes - 1, one root, no forks)."""
    import ctypes
    from collections import Counter
    hip = ctypes.CDLL("libamdhip64.so")
    gh = ctypes.c_void_p(raw_graph)
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(gh, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(gh, nodes, ctypes.byref(n)) == 0
    types = []
    for nd in nodes:
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t)) == 0
        types.append(t.value)
    e = ctypes.c_size_t(0)
    assert hip.hipGraphGetEdges(gh, None, None, ctypes.byref(e)) == 0
    src, dst = (ctypes.c_void_p * max(1, e.value))(), (ctypes.c_void_p * max(1, e.value))()
    if e.value:
        assert hip.hipGraphGetEdges(gh, src, dst, ctypes.byref(e)) == 0
    outdeg, indeg = Counter(src[i] for i in range(e.value)), Counter(dst[i] for i in range(e.value))
    names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
             10: "mem_alloc", 11: "mem_free"}
    by_type = Counter(names.get(t, str(t)) for t in types)
    roots = [nd for nd in nodes if indeg.get(nd, 0) == 0]
    forks = [(i, names.get(types[i], types[i]), indeg.get(nd, 0), outdeg.get(nd, 0)) for i, nd in enumerate(nodes)
             if indeg.get(nd, 0) > 1 or outdeg.get(nd, 0) > 1]
    return (f"nodes {n.value} by type {dict(by_type)}\nedges {e.value}\nroots {len(roots)}\n"
            f"nodes with in- or out-degree > 1 (index, type, in, out): {forks}\n")


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
        g = torch.cuda.CUDAGraph(keep_graph=True) if DUMP else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        if DUMP:
            with open(DUMP, "w") as f:
                f.write(describe_graph(g.raw_cuda_graph()))
        self.graph = g              # (the capture itself executed nothing: replay it for this batch)
        self._replay()

    def _replay(self):
        if WAIT == "device":
            torch.cuda.synchronize(self.dev)
        elif WAIT == "stream":
            torch.cuda.current_stream(self.dev).synchronize()
        elif WAIT != "event" or IN_FLIGHT != 1:
            raise RuntimeError("R_TUCKER_AMD_GRAPH_WAIT must be stream | device | event (event: IN_FLIGHT = 1; known to be unsafe)")
        elif IN_FLIGHT > 0 and len(self._events) >= IN_FLIGHT:
            self._events.pop(0).synchronize()
        self.graph.replay()
        self.replays += 1
        if WAIT == "event" and IN_FLIGHT > 0:
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
