#!/usr/bin/env python3
"""EXPERIMENT (tools/graphstep.py): the HIP-graph replay of the optimizer step against the same steps run eagerly
from the same start -- same batches, same learning-rate / regulariser schedule -> same parameters (with
graphstep.WAIT = "stream": every replay launched onto a drained stream).  Was tests/test_gpu_driver.py::
test_captured_step_equals_eager_steps while the module was part of the package.  Run on a GPU box:
    python tools/graph_capture_equals_eager.py"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(tmp_path):
    """The HIP-graph replay of the optimizer step (graphstep.CapturedTrainStep) against the same steps run
    eagerly from the same start: same batches, same learning-rate / regulariser schedule -> same parameters."""
    import r_tucker_amd as rt
    from r_tucker_amd import driver
    import graphstep
    graphstep.install()
    from r_tucker_amd.data import Data, KG_dataset
    from r_tucker_amd.model.asymmetric.optim import RSGDwithMomentum
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
    flt = rt.DeviceFilter(train_set, "cuda")
    rank = (6, 24, 24)

    def run(enabled):
        graphstep.ENABLED = enabled
        torch.manual_seed(5)
        model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), rank)
        model.init()
        model.cuda()
        params = torch.nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
        opt = RSGDwithMomentum(params, rank, 50.0, 0.8)
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
        gen = torch.Generator(device="cuda").manual_seed(11)
        losses = []
        for epoch in range(2):
            step = driver._captured_step(model, opt, flt, 512, 0.1)
            step.begin_epoch(1e-9 * (epoch + 1))
            for b in range(5):
                step.run(torch.randint(0, flt.features.shape[0], (512,), device="cuda", generator=gen))
            losses.append(step.totals())
            sched.step()
        replays = step.replays
        return [p.detach().clone() for p in params], losses, replays

    was, dump = graphstep.ENABLED, graphstep.DUMP
    try:
        eager, le, n0 = run(False)
        graphstep.DUMP = os.path.join(tmp_path, "step_graph.txt")          # also write the capture's topology
        graph, lg, n1 = run(True)
    finally:
        graphstep.ENABLED, graphstep.DUMP = was, dump
    assert n0 == 0 and n1 == 10 - graphstep.EAGER_STEPS
    topo = open(os.path.join(tmp_path, "step_graph.txt")).read()
    assert "roots 1\n" in topo and topo.rstrip().endswith(": []"), topo      # one chain: nothing may run beside anything
    for a, b in zip(eager, graph):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())
    for (a0, a1), (b0, b1) in zip(le, lg):
        assert abs(a0 - b0) <= 1e-5 * abs(a0) and abs(a1 - b1) <= 1e-4 * abs(a1)

    print("captured step == eager steps over", n1, "replays")


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        main(d)
