"""Host-side utilities of the driver surface (no GPU): coefficient schedules and run histories / checkpoints
(reference behaviour: src/utils/regularization.py, src/utils/storage.py)."""
import pickle

import pytest
import torch

from r_tucker_amd.utils.regularization import (CyclicDecreasingPolicy, RegularizationCoeffPolicy, SimpleDecreasingPolicy,
                                               SimpleIncreasingPolicy)
from r_tucker_amd.utils.storage import Losses, Metric, Metrics, StateDict


def test_decreasing_policies_follow_the_reference_arithmetic():
    # README recipe: "exp" 1e-4 -> 3e-9 in 350 steps; the value is moved while it is above the final one, so it
    # overshoots once
    p = SimpleDecreasingPolicy(1e-4, 350, 3e-9, "exp")
    v = [p.step() for _ in range(360)]
    ratio = (3e-9 / 1e-4) ** (1 / 350)
    assert v[0] == pytest.approx(1e-4 * ratio, rel=1e-12)
    assert v[349] > 3e-9 and v[350] == pytest.approx(v[349] * ratio, rel=1e-12) and v[350] < 3e-9
    assert v[351:] == [v[350]] * 9 and p.cur_step == 360
    # the reference's default: linear 1e-11 -> 1e-16 in 300 steps
    p = SimpleDecreasingPolicy(1e-11, 300, 1e-16)
    v = [p.step() for _ in range(305)]
    assert v[0] == pytest.approx(1e-11 - (1e-11 - 1e-16) / 300, rel=1e-12)
    assert all(a > b for a, b in zip(v[:299], v[1:300])) and v[300] == v[304] <= 1e-16 * (1 + 1e-6)
    # "cos": a cosine ramp over the step COUNT (the reference passes the value where an index is expected)
    p = SimpleDecreasingPolicy(1.0, 4, 0.0, "cos")
    assert [round(p.step(), 4) for _ in range(6)] == [0.8536, 0.5, 0.1464, 0.0, 0.0, 0.0]
    p = SimpleDecreasingPolicy(1.0, 4, 0.5, "const")
    assert [p.step() for _ in range(3)] == [1.0, 1.0, 1.0]
    with pytest.raises(NotImplementedError):
        SimpleDecreasingPolicy(1.0, 4, 0.5, "sqrt")


def test_other_policies():
    p = SimpleIncreasingPolicy(1.0, 4, 2.0)
    assert [p.step() for _ in range(6)] == [1.25, 1.5, 1.75, 2.0, 2.0, 2.0]
    with pytest.raises(NotImplementedError):
        SimpleIncreasingPolicy(1.0, 4, 2.0, "exp")
    p = CyclicDecreasingPolicy(1.0, 2, 0.0)
    assert [p.step() for _ in range(6)] == [0.5, 0.0, 0.5, 0.0, 0.5, 0.0]
    p = RegularizationCoeffPolicy(3.0, 10)
    assert [p.step() for _ in range(3)] == [3.0, 3.0, 3.0] and p.cur_step == 3


def test_histories_and_checkpoint_round_trip(tmp_path):
    a, b = Losses(), Losses()
    a.update(train_loss=1.0, train_norm=2.0, val_loss=3.0, test_loss=4.0)
    b.update(5.0, 6.0, 7.0, 8.0)                       # positional order of the reference: train, norm, val, test
    a.merge(b)
    assert (a.train, a.norms, a.val, a.test) == ([1.0, 5.0], [2.0, 6.0], [3.0, 7.0], [4.0, 8.0])
    m, m2 = Metrics(), Metrics()
    m.update({"mrr": .1, "hits@1": .2, "hits@3": .3, "hits@10": .4}, "val")
    m2.update({"mrr": .5, "hits@1": .6, "hits@3": .7, "hits@10": .8}, "test")
    m.merge(m2)
    assert m.mrr["val"] == [.1] and m.mrr.test == [.5] and m.hits_10["test"] == [.8] and isinstance(m.hits_3, Metric)
    assert pickle.loads(pickle.dumps(m)) == m
    sd = StateDict({"core": torch.arange(3.0)}, a, m, 7, None, {"last_epoch": 7})
    path = sd.save(str(tmp_path), "rk_200")
    assert path.endswith("rk_200_7.pth")
    back = StateDict.load(path[:-4])                   # the reference passes the name without the suffix
    assert back.losses == a and back.metrics == m and back.last_epoch == 7 and back.scheduler == {"last_epoch": 7}
    assert torch.equal(back.model["core"], torch.arange(3.0))
    assert sd.save(str(tmp_path), "snapshot", add_epoch=False).endswith("snapshot.pth")
