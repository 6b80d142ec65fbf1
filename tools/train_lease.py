#!/usr/bin/env python3
"""The README recipe of the reference (README.md:36-45: WN18RR, rank (10,200,200), RSGD with momentum 0.8, lr 2000
decayed .9981 / epoch, regulariser "exp" 1e-4 -> 3e-9 over 350 epochs, label smoothing 0.1, 1450 epochs) run in
time-boxed LEASES that resume from a checkpoint -- a GPU box is granted for at most 20 minutes at a time.

    python tools/train_lease.py --tag full --compress 1 --minutes 17 [--resume ckpt.npz]

``--compress k`` divides the three schedules by one factor (epochs and regulariser steps / k, lr decay ** k), the
form VERDICT r02 item 4 asks for when the whole recipe does not fit; k = 1 is the recipe itself.  Every epoch
appends one JSON line to ``gpurun_out/train_<tag>.log`` (train loss, gradient norm, validation metrics; test
metrics every ``--test-every`` epochs and at the end).  The checkpoint holds the parameters with the low byte of
every fp32 dropped (3 bytes per value: 51 MB instead of 67 -- what a lease may hand back is capped at 64 MiB;
relative error 2^-16, the factors are re-orthonormalised on load), the epoch, and the schedule state.  The
momentum is NOT carried over a lease boundary (131 MB; it rebuilds within a few steps at beta = 0.8).
The shuffle of epoch e is seeded by (seed, e): a run cut into leases sees the same batches as one that is not.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pack24(t: torch.Tensor) -> np.ndarray:
    u = t.detach().cpu().contiguous().view(torch.int32).numpy().view(np.uint32)
    u = ((u.astype(np.uint64) + 0x80) >> 8).astype(np.uint32)           # round to 24 bits
    u = np.minimum(u, 0xFFFFFF)
    b = np.empty(u.shape + (3,), dtype=np.uint8)
    b[..., 0], b[..., 1], b[..., 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
    return b


def unpack24(b: np.ndarray) -> torch.Tensor:
    u = (b[..., 0].astype(np.uint32) | (b[..., 1].astype(np.uint32) << 8) | (b[..., 2].astype(np.uint32) << 16)) << 8
    return torch.from_numpy(u.view(np.int32).copy()).view(torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="readme")
    ap.add_argument("--compress", type=float, default=1.0)
    ap.add_argument("--minutes", type=float, default=17.0, help="stop (and checkpoint) after this much wall time")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--seed", type=int, default=322)
    ap.add_argument("--test-every", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--max-epochs", type=int, default=None, help="stop after this many epochs of this lease (tests)")
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--reg-init", type=float, default=None)
    ap.add_argument("--reg-final", type=float, default=None)
    ap.add_argument("--const-lr", type=float, default=None, help="hold the learning rate at this value (diagnostic continuation)")
    ap.add_argument("--extra-epochs", type=int, default=0, help="run this many epochs past the schedule's end (with --const-lr)")
    ap.add_argument("--save-tag", default=None, help="write log / checkpoint under this tag instead of --tag")
    ap.add_argument("--debug-from-epoch", type=int, default=None,
                    help="from this epoch on, check every intermediate of the optimizer step for non-finite values (slow) "
                         "and stop with the name of the first one")
    ap.add_argument("--onecycle", action="store_true",
                    help="the scheduler the reference's train.py hard-codes (train.py:213-215): OneCycleLR(max_lr=600, "
                         "total_steps=epochs, pct_start=100/epochs, div_factor=5.5, linear), stepped once per epoch")
    ap.add_argument("--debug-to-epoch", type=int, default=None, help="last epoch of --debug-from-epoch's checks")
    ap.add_argument("--epochs", type=int, default=None, help="schedule length (default: the README recipe's 1450)")
    ap.add_argument("--lr-decay", type=float, default=None, help="per-epoch lr factor (default: the README recipe's .9981)")
    ap.add_argument("--reg-steps", type=int, default=None)
    ap.add_argument("--reg-policy", default=None, choices=["exp", "linear"])
    ap.add_argument("--variant-norm", default="frobenius", choices=["frobenius", "coordinate"],
                    help="what TangentVector.norm() measures (riemannian.EXPERIMENT; DESIGN.md section 8)")
    ap.add_argument("--variant-reg-excluded", action="store_true",
                    help="normalise the step by the DATA term's gradient norm only (riemannian.EXPERIMENT)")
    args = ap.parse_args()
    t_start = time.time()

    import r_tucker_amd as rt
    from configs.base_config import wn18rr_readme_config
    from r_tucker_amd import driver, tucker
    from r_tucker_amd import riemannian as _riem
    _riem.EXPERIMENT["norm"] = args.variant_norm
    _riem.EXPERIMENT["reg_in_norm"] = not args.variant_reg_excluded
    from r_tucker_amd.data import Data, KG_dataset
    from r_tucker_amd.utils.regularization import SimpleDecreasingPolicy

    k = args.compress
    cfg = wn18rr_readme_config()
    tc = cfg.train_cfg
    n_epochs = int(math.ceil((args.epochs or tc.num_epoches) / k)) + args.extra_epochs
    reg_steps = max(1, int(round((args.reg_steps or tc.num_regularizer_decreasing_steps) / k)))
    gamma = (args.lr_decay or tc.scheduler_step) ** k
    lr0 = args.lr if args.lr is not None else tc.learning_rate
    reg0 = args.reg_init if args.reg_init is not None else tc.base_regularization_coeff
    reg1 = args.reg_final if args.reg_final is not None else tc.final_regularization_coeff
    rank = cfg.model_cfg.manifold_rank
    dev = torch.device("cuda")

    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), rank)
    model.init()
    epoch0 = 0
    reg_state = None
    if args.resume:
        z = np.load(args.resume, allow_pickle=False)
        with torch.no_grad():
            model.core.copy_(torch.from_numpy(z["core"]))
            model.R.weight.copy_(torch.from_numpy(z["R"]))
            for name, w in (("S", model.S.weight), ("O", model.O.weight)):
                f = unpack24(z[name]).double()
                q, r_ = torch.linalg.qr(f)                                  # repair what the 24-bit rounding did
                w.copy_((q * torch.sign(torch.diagonal(r_))).float())
        epoch0 = int(z["epoch"])
        reg_state = (float(z["reg_val"]), int(z["reg_cur_step"]), int(z["reg_moves"]))
        assert float(z["compress"]) == k, "resume with the compress factor the run was started with"
    model.to(dev)

    opt = driver.define_optimizer(model, cfg, "asymmetric", "rsgd")
    def lr_at(e):
        return args.const_lr if args.const_lr is not None else lr0 * gamma ** e

    sched = None
    if args.onecycle:
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=600, total_steps=n_epochs, pct_start=100 / n_epochs, div_factor=5.5,
                                                    cycle_momentum=False, anneal_strategy="linear")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")              # (stepping the schedule before the optimizer: a resumed lease)
            for _ in range(epoch0):
                sched.step()

        def lr_at(e):                                    # noqa: F811  (only used by the non-finite restore path)
            return opt.param_groups[0]["lr"]
    else:
        for g in opt.param_groups:
            g["lr"] = lr_at(epoch0)
    regulizer = SimpleDecreasingPolicy(reg0, reg_steps, reg1, args.reg_policy or tc.coeff_adjusting_policy)
    if reg_state is not None:
        regulizer.val, regulizer.cur_step, regulizer._moves = reg_state
    train_set = KG_dataset(data, data.train_data, label_smoothing=tc.label_smoothig)
    val_set = KG_dataset(data, data.valid_data, test_set=True)
    test_set = KG_dataset(data, data.test_data, test_set=True)
    train_flt = rt.DeviceFilter(train_set, dev)
    val_flt, test_flt = rt.DeviceFilter(val_set, dev), rt.DeviceFilter(test_set, dev)

    os.makedirs(args.out, exist_ok=True)
    out_tag = args.save_tag or args.tag
    log_path = os.path.join(args.out, f"train_{out_tag}.log")
    ckpt_path = os.path.join(args.out, f"ckpt_{out_tag}.npz")

    def log(rec):
        with open(log_path, "a") as f:
            f.write(json.dumps(rec) + "\n")
        print(json.dumps(rec), flush=True)

    def save(epoch):
        sd = {k_: v.detach() for k_, v in model.state_dict().items()}
        np.savez(ckpt_path, core=sd["core"].cpu().numpy(), R=sd["R.weight"].cpu().numpy(), S=pack24(sd["S.weight"]),
                 O=pack24(sd["O.weight"]), epoch=epoch, reg_val=regulizer.val, reg_cur_step=regulizer.cur_step,
                 reg_moves=regulizer._moves, compress=k)

    log({"event": "lease_start", "tag": args.tag, "compress": k, "epochs_total": n_epochs, "reg_steps": reg_steps,
         "lr_decay": gamma, "lr0": lr0, "reg": [reg0, reg1], "resume_epoch": epoch0, "device": torch.cuda.get_device_name(0)})
    budget = args.minutes * 60.0
    epoch = epoch0
    epoch_times = []
    import graphstep                                    # tools/graphstep.py (experiment; same directory)
    if graphstep.ENABLED:                               # R_TUCKER_AMD_GRAPH=1: the step replayed from a HIP graph
        graphstep.install()

    def snapshot():
        return [p.detach().clone() for p in opt.param_groups[0]["params"]], (regulizer.val, regulizer.cur_step, regulizer._moves)

    good = snapshot()
    while epoch < n_epochs:
        if args.max_epochs is not None and epoch - epoch0 >= args.max_epochs:
            break
        per = (sum(epoch_times[-5:]) / len(epoch_times[-5:])) if epoch_times else 30.0
        if time.time() - t_start + 1.5 * per + 20.0 > budget:
            break
        epoch += 1
        te = time.time()
        coeff = regulizer.step()
        torch.manual_seed(args.seed * 100003 + epoch)
        lr = opt.param_groups[0]["lr"]
        if args.debug_from_epoch is not None:
            from r_tucker_amd import optim as _optim
            tucker._CHECK = _optim.CHECK_FINITE = (args.debug_from_epoch <= epoch <= (args.debug_to_epoch or 10 ** 9))
        try:
            train_loss, gnorm = driver.train_one_epoch(model, opt, train_flt, tc.train_batch_size, tc.label_smoothig,
                                                       regularization_coeff=coeff)
        except FloatingPointError as e:
            sv = [torch.linalg.svdvals(model.core.detach().double().movedim(m, 0).reshape(model.core.shape[m], -1)) for m in range(3)]
            log({"event": "first_non_finite_intermediate", "epoch": epoch, "what": str(e),
                 "core_singular_values_first_last": [[float(v[0]), float(v[-3]), float(v[-2]), float(v[-1])] for v in sv]})
            with torch.no_grad():
                for p, g in zip(opt.param_groups[0]["params"], good[0]):
                    p.copy_(g)
            epoch -= 1
            break
        t_train = time.time() - te
        vm, vl = driver.evaluate(model, val_set, tc.eval_batch_size, val_flt)
        rec = {"epoch": epoch, "train_loss": train_loss, "grad_norm": gnorm, "lr": lr, "reg_coeff": coeff,
               "core_norm": float(model.core.detach().norm()), "val_loss": float(vl), "epoch_time": t_train}
        rec.update({f"val_{k_}": v for k_, v in vm.items()})
        if epoch % args.test_every == 0 or epoch == n_epochs:
            tm, tl = driver.evaluate(model, test_set, tc.eval_batch_size, test_flt)
            rec.update({f"test_{k_}": v for k_, v in tm.items()})
            rec["test_loss"] = float(tl)
        health = tucker.read_health(dev)
        if health:
            rec["retraction_health"] = max(health.values())
        log(rec)
        epoch_times.append(time.time() - te)
        # the finiteness check comes BEFORE the schedule moves: a restored epoch is replayed at its own learning rate
        ok = math.isfinite(train_loss) and all(bool(torch.isfinite(p).all()) for p in opt.param_groups[0]["params"])
        if ok:
            if sched is not None:
                if epoch < n_epochs:
                    sched.step()
            else:
                for g in opt.param_groups:
                    g["lr"] = lr_at(epoch)
        if not ok:
            # restore the last good epoch and go on WITHOUT the HIP graph (launches onto a busy stream)
            log({"event": "non_finite_state", "epoch": epoch, "graph_was_enabled": graphstep.ENABLED, "action": "restore + eager"})
            if not graphstep.ENABLED:                   # eager steps produced it: a real numerical problem, stop
                with torch.no_grad():                   # (the checkpoint keeps the last good epoch)
                    for p, g in zip(opt.param_groups[0]["params"], good[0]):
                        p.copy_(g)
                regulizer.val, regulizer.cur_step, regulizer._moves = good[1]
                epoch -= 1
                break
            with torch.no_grad():
                for p, g in zip(opt.param_groups[0]["params"], good[0]):
                    p.copy_(g)
            regulizer.val, regulizer.cur_step, regulizer._moves = good[1]
            opt._prev = None
            graphstep.ENABLED = False
            cap = getattr(opt, "_rtk_captured", None)
            if cap is not None and hasattr(cap[1], "drop_graph"):
                cap[1].drop_graph()
            epoch -= 1
            if sched is None:
                for g in opt.param_groups:
                    g["lr"] = lr_at(epoch)
            continue
        good = snapshot()
    save(epoch)
    log({"event": "lease_end", "epoch": epoch, "done": epoch >= n_epochs, "wall_s": time.time() - t_start,
         "checkpoint_bytes": os.path.getsize(ckpt_path)})


if __name__ == "__main__":
    main()
