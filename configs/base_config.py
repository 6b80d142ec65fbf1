"""Hyper-parameters of a run -- the surface of the reference's ``configs/base_config.py`` (``Config(None)`` at
``train.py:198``; ``cfg.train_cfg.*``, ``cfg.model_cfg.*``, ``cfg.log_cfg.*``, ``cfg.tune_cfg.*`` with the same
attribute names, the reference's own misspellings ``num_epoches`` / ``label_smoothig`` included), as real
dataclass fields so a run can override them, plus the two recipes of the reference's README as named
configurations (``README.md:36-45``): the shipped defaults do NOT encode them.

New fields (not in the reference): ``TrainConfig.scheduler`` -- ``"onecycle"`` is what ``train.py:213-215``
hard-codes (OneCycleLR, max_lr 600, ignoring ``learning_rate``), ``"exp"`` is the README's ``lr`` /
``lr_decay`` column (``learning_rate`` decayed by ``scheduler_step`` per epoch); ``LogConfig.use_wandb``.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import Any, Optional, Tuple


class _Dictable:
    def to_dict(self):
        return asdict(self)


@dataclass
class TrainConfig(_Dictable):
    train_batch_size: int = 512
    eval_batch_size: int = 512

    num_epoches: int = 500
    momentum_beta: float = 0.8
    label_smoothig: float = 0.1
    learning_rate: float = 2000
    scheduler_step: float = 0.995
    scheduler: str = "onecycle"

    base_regularization_coeff: float = 1e-11
    final_regularization_coeff: float = 1e-16
    coeff_adjusting_policy: str = "linear"
    num_regularizer_decreasing_steps: int = 300

    checkpoint_path: str = "checkpoints/"


@dataclass
class TuneConfig(_Dictable):
    num_tunning_runs: int = 5
    num_run_epochs: int = 30
    relation_rank_inc: int = 0
    entity_rank_inc: int = 1


@dataclass
class ModelConfig(_Dictable):
    manifold_rank: Tuple[int, int, int] = (200, 100, 100)
    use_pretrained: bool = False
    pretrained_path: str = "./checkpoints/rk_20_903"


@dataclass
class LogConfig(_Dictable):
    project_name: str = "R_TuckER"
    entity_name: str = "johan_ddc_team"
    run_name: str = "R-TuckER on MI355X"
    log_dir: str = "wandb_logs"
    watch_log_freq: int = 500
    watch_log: str = "all"
    use_wandb: bool = False          # wandb is optional here (the reference imports it unconditionally, Appendix A8)


@dataclass
class Config:
    state_dict: Optional[Any] = None      # a utils.storage.StateDict when resuming (train.py:206-209)
    train_cfg: TrainConfig = field(default_factory=TrainConfig)
    model_cfg: ModelConfig = field(default_factory=ModelConfig)
    log_cfg: LogConfig = field(default_factory=LogConfig)
    tune_cfg: TuneConfig = field(default_factory=TuneConfig)


def _readme(rank, reg_finish, reg_steps) -> Config:
    cfg = Config(None)
    cfg.model_cfg.manifold_rank = rank
    t = cfg.train_cfg
    t.train_batch_size = t.eval_batch_size = 512
    t.learning_rate, t.scheduler_step, t.scheduler = 2000, 0.9981, "exp"
    t.coeff_adjusting_policy, t.base_regularization_coeff, t.final_regularization_coeff = "exp", 1e-4, reg_finish
    t.num_regularizer_decreasing_steps = reg_steps
    t.momentum_beta, t.label_smoothig, t.num_epoches = 0.8, 0.1, 1450
    return cfg


def wn18rr_readme_config() -> Config:
    """README.md:40  -- rank (rel, ent) = (10, 200), run with ``--mode asymmetric --optim rsgd --seed 322``."""
    return _readme((10, 200, 200), 3e-9, 350)


def fb15k237_readme_config() -> Config:
    """README.md:41  -- rank (rel, ent) = (200, 20)."""
    return _readme((200, 20, 20), 1e-10, 100)


NAMED_CONFIGS = {"base": lambda: Config(None), "wn18rr_readme": wn18rr_readme_config, "fb15k237_readme": fb15k237_readme_config}
