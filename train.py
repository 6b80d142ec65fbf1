#!/usr/bin/env python3
"""R-TuckER training on MI355X -- the command line of the reference's ``train.py`` (``train.py:170-252``):

    python train.py --mode asymmetric --optim rsgd --seed 322 --data data/WN18RR/ [--config wn18rr_readme]

Same flags (``--mode --seed --nw --device --optim --data --tune/--notune``), same ``Config`` attributes
(``configs/base_config.py``), same loop (``r_tucker_amd.driver``: Riemannian ``fit`` / ``step`` per batch,
validation + test evaluation per epoch, ``checkpoints/snapshot.pth`` and ``rk_<rank>_<epoch>.pth``), and the same
final report.  Differences, all documented in DESIGN.md: scoring, loss, backward and ranking run in the HIP
kernels of this package (``--device`` must be a GPU: there is no CPU path); ``--nw`` is accepted and unused (no
DataLoader workers: the split's target lists live on the GPU as a CSR); wandb is optional
(``LogConfig.use_wandb``); ``--config`` selects a named configuration (``base`` = the reference's shipped
defaults, ``wn18rr_readme`` / ``fb15k237_readme`` = the recipes of its README); ``--epochs`` / ``--max-batches``
shorten a run for smoke tests.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_scheduler(opt, cfg):
    tc = cfg.train_cfg
    if tc.scheduler == "onecycle":      # what train.py:213-215 hard-codes
        return torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=600, total_steps=tc.num_epoches,
                                                   pct_start=min(1.0, 100 / tc.num_epoches), div_factor=5.5,
                                                   cycle_momentum=False, anneal_strategy="linear")
    if tc.scheduler == "exp":           # README.md:36-41: lr, lr_decay
        return torch.optim.lr_scheduler.ExponentialLR(opt, gamma=tc.scheduler_step)
    raise ValueError(f"unknown scheduler {tc.scheduler!r}")


def _coerce(target, name, raw, item):
    """Value of ``--set SECTION.FIELD=RAW`` by the dataclass field's ANNOTATED type (a float field whose default is
    the int literal 2000 still takes 0.5 or 1e-2); JSON for tuples / bools / None, the raw string as a last resort."""
    import dataclasses
    import typing
    hints = typing.get_type_hints(type(target)) if dataclasses.is_dataclass(target) else {}
    want = hints.get(name)
    try:
        if want is float:
            return float(raw)
        if want is int:
            return int(raw)
        if want is str:
            return raw
        val = json.loads(raw)
        return tuple(val) if isinstance(val, list) else val
    except (ValueError, TypeError):
        if want in (float, int):
            raise SystemExit(f"--set {item}: {name} expects a {want.__name__}, got {raw!r}")
        return raw


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--mode", type=str, help="Model type", required=True)
    parser.add_argument("--seed", type=int, help="Random seed", default=20)
    parser.add_argument("--nw", type=int, help="Num workers", default=6)
    parser.add_argument("--device", type=str, help="Device", default="cuda")
    parser.add_argument("--optim", type=str, help="Optimizer", default="rsgd")
    parser.add_argument("--data", type=str, help="Dataset path", default="data/FB15k-237/")
    tune_parser = parser.add_mutually_exclusive_group(required=False)
    tune_parser.add_argument("--tune", dest="tune", action="store_true", help="Use rank tunning")
    tune_parser.add_argument("--notune", dest="tune", action="store_false", help="Do not use rank tunning")
    parser.set_defaults(tune=False)
    parser.add_argument("--config", default="base", help="named configuration (configs/base_config.py: NAMED_CONFIGS)")
    parser.add_argument("--epochs", type=int, default=None, help="override train_cfg.num_epoches")
    parser.add_argument("--max-batches", type=int, default=None, help="cap the batches per epoch (smoke runs)")
    parser.add_argument("--rank", type=int, nargs=3, default=None, help="override model_cfg.manifold_rank")
    parser.add_argument("--checkpoint-path", default=None, help="override train_cfg.checkpoint_path")
    parser.add_argument("--set", dest="overrides", action="append", default=[], metavar="SECTION.FIELD=VALUE",
                        help="override a configuration field, e.g. --set train_cfg.learning_rate=50 (repeatable)")
    args = parser.parse_args(argv)
    if args.mode not in ("symmetric", "asymmetric"):
        raise SystemExit("--mode must be symmetric or asymmetric")
    if not args.device.startswith("cuda"):
        raise SystemExit("this build scores on MI355X only (--device cuda[:i]); for CPU runs use the reference implementation")

    import r_tucker_amd as rt
    from configs.base_config import NAMED_CONFIGS
    from r_tucker_amd import driver
    from r_tucker_amd.data import Data, KG_dataset
    from r_tucker_amd.utils.regularization import SimpleDecreasingPolicy
    from r_tucker_amd.utils.storage import StateDict
    from r_tucker_amd.utils.utils import set_random_seed

    data = Data(args.data, reverse=True)
    set_random_seed(args.seed)
    rt.set_backend("pytorch")
    cfg = NAMED_CONFIGS[args.config]()
    if args.epochs is not None:
        cfg.train_cfg.num_epoches = args.epochs
    if args.rank is not None:
        cfg.model_cfg.manifold_rank = tuple(args.rank)
    for item in args.overrides:
        key, _, raw = item.partition("=")
        section, _, name = key.partition(".")
        try:
            target = getattr(cfg, section)
            getattr(target, name)
        except AttributeError:
            raise SystemExit(f"--set {item}: no field {section}.{name} in the configuration")
        setattr(target, name, _coerce(target, name, raw, item))
    tc = cfg.train_cfg
    if args.checkpoint_path is not None:
        tc.checkpoint_path = args.checkpoint_path

    Model = rt.SymmetricR_TuckER if args.mode == "symmetric" else rt.AsymmetricR_TuckER
    model = Model((len(data.entities), len(data.relations)), cfg.model_cfg.manifold_rank, device=args.device)
    model_state_dict = None
    if cfg.model_cfg.use_pretrained:
        cfg.state_dict = StateDict.load(cfg.model_cfg.pretrained_path)
        model_state_dict = cfg.state_dict.model
    model.init(model_state_dict)
    model.to(args.device)

    opt = driver.define_optimizer(model, cfg, args.mode, args.optim)
    scheduler = build_scheduler(opt, cfg)
    regulizer = SimpleDecreasingPolicy(tc.base_regularization_coeff, tc.num_regularizer_decreasing_steps,
                                       tc.final_regularization_coeff, tc.coeff_adjusting_policy)
    train_set = KG_dataset(data, data.train_data, label_smoothing=tc.label_smoothig)
    val_set = KG_dataset(data, data.valid_data, test_set=True)
    test_set = KG_dataset(data, data.test_data, test_set=True)

    run = None
    if cfg.log_cfg.use_wandb:
        import wandb
        run = wandb.init(project=cfg.log_cfg.project_name, entity=cfg.log_cfg.entity_name, name=cfg.log_cfg.run_name,
                         dir=cfg.log_cfg.log_dir, config={"model": cfg.model_cfg.to_dict(), "train_params": tc.to_dict(),
                                                          "tune": cfg.tune_cfg.to_dict()})
    final_state = driver.train(model, opt, train_set, val_set, test_set, cfg, regulizer=regulizer, scheduler=scheduler,
                               log=lambda rec: print(json.dumps(rec) if isinstance(rec, dict) else rec, flush=True),
                               wandb_run=run, max_batches_per_epoch=args.max_batches)
    if run is not None:
        run.finish()
    print("Final loss value:", final_state.losses.test[-1], sep="\t")
    print("Final mrr value:", final_state.metrics.mrr.test[-1], sep="\t")
    print("Final hits@1 value:", final_state.metrics.hits_1.test[-1], sep="\t")
    print("Final hits@3 value:", final_state.metrics.hits_3.test[-1], sep="\t")
    print("Final hits@10 value:", final_state.metrics.hits_10.test[-1], sep="\t")
    final_state.save(tc.checkpoint_path, f"rk_{model.rank[1]}_final", add_epoch=False)
    return final_state


if __name__ == "__main__":
    main()
