#!/usr/bin/env python3
"""bench.py -- 1-vs-N triples scored / s of the HIP scoring path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started without a torch.distributed environment, `--gpus N` launches the second form
itself (N fresh child processes, one rank per GPU over RCCL, before this process has touched a GPU) and exits
with the children's status; rank 0's JSON line is the only thing on stdout.

A "step" = one pass of the whole hot path (query vectors + 1-vs-all scores + sigmoid)
over one batch of synthetic (h, r) queries, operands already resident in HBM.
N = 1 workload: BASELINE.json configs[1] -- WN18RR asymmetric rank (10,200,200),
batch 512, fp32, 40 943 entities, 22 relations.  N > 1: the entity matrix O is
row-sharded over the ranks and the per-shard score blocks are all-gathered with RCCL
(north_star); every rank scores the same batch against its shard ("strong" scaling).
For a large relation rank (a > 32) stage 1 is split over the ranks too (each contracts
ceil(B/N) queries, one all-gather of the B x c vectors).

The headline (`--tables per-batch`, the default since round 4) runs ALL of the path in every step, relation
tables G x_0 R[r] included -- what the reference does per batch and what the CPU leg is timed on.
Parameters are fixed while a split is evaluated (train.py:94-125 scores every batch of valid / test with
the same extract_tensor(model)), so an evaluation pass can build the tables -- a function of the
parameters only -- once: `--tables cached` rebuilds them every EVAL_BATCHES steps inside the timed region;
the N = 1 line carries that figure as the extra key `cached_tables_ms_per_step`.

The K timed steps are launched as ONE HIP graph in a short single-GPU run (`--launch auto`: K <= 256; captured before
the timed region, launched once inside it) and one C-ABI call after the other otherwise; `config.launch` says which,
`eager_ms_per_step` is the eager loop of the same K steps timed right behind the graph.

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline`
for the dominant kernel (the score kernel: `kernel_ms` = its own begin -> end on HIP events
handed to the launch, rtk_timer_* of the C ABI, on steps right behind the timed region;
`stream_bracket_ms` = HIP events recorded on its stream around the launch inside the timed
region) and `cpu_baseline` (the oracle's restatement of the reference
op sequence timed on the host cores; N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import datetime
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_ent, n_rel, batch, rank, dtype); the entity matrix is row-sharded over the ranks of the run
    "wn18rr_asym_r10x200_b512_f32": (40943, 22, 512, (10, 200, 200), "f32"),          # BASELINE configs[1]: the bench line
    # parity-test / measurement cases (selectable; not the default line)
    "fb15k237_sym_r200x200_b2048_bf16": (14541, 474, 2048, (200, 200, 200), "bf16"),  # configs[2]
    "fb15k_asym_r200x200_b512_f32": (14951, 2690, 512, (200, 200, 200), "f32"),       # configs[3], B = 512
    "fb15k_asym_r200x200_b2048_f32": (14951, 2690, 2048, (200, 200, 200), "f32"),     # configs[3], B = 2048
    "synthetic1m_r256x512_b8192_bf16": (1_000_000, 1000, 8192, (256, 512, 512), "bf16"),  # configs[4]: 1 M entities / world
    # one GPU's share of configs[4] (1 M entities over 8 GPUs): shard-local measurement on a single GPU
    "synthetic1m_shard125k_r256x512_b8192_bf16": (125000, 1000, 8192, (256, 512, 512), "bf16"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
EVAL_BATCHES = 16      # --tables cached: the relation tables are rebuilt every this many steps (one "evaluation pass")


def score_kernel_name(n_loc, c):
    """Which fp32 split-fp16 score kernel the library's dispatcher takes for this shape (csrc/rtk_score_cg.hip,
    csrc/rtk_score_split.hip; RTK_SCORE_KERNEL overrides it)."""
    forced = os.environ.get("RTK_SCORE_KERNEL", "")
    if forced.startswith("v3") or c > 208 or c % 4:
        return "score_split_kernel"
    if forced.startswith("ws"):
        return "score_ws_kernel"
    G = -(-n_loc // 32)
    on_cg = forced.startswith("cg") or 576 <= G <= 1280      # one set per workgroup, at least two groups in it
    return "score_cg_kernel" if on_cg else "score_ws_kernel"


def cpu_baseline(gen, n_ent, n_rel, B, rank, pool, budget_s=15.0):
    """The oracle (CPU restatement of the reference's five torch ops) on the host cores.
    torch's intra-op thread count is swept (the box exposes more hardware threads than the
    cgroup grants; oversubscription makes the small GEMMs slower) and the best is reported."""
    from oracle import score_oracle as orc
    core, R, S, O = [torch.from_numpy(x) for x in gen.make_params(n_ent, n_rel, rank, 322)]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({t for t in (avail, 64, 32, 16, 8, 1) if t <= avail})       # (1 thread: SURVEY.md section 8d asks for both)
    best = None
    tried = []
    with torch.no_grad():
        for th in cands:
            torch.set_num_threads(th)
            for i in range(2):
                orc.score_ref(core, R, S, O, pool[i][0], pool[i][1])
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < budget_s / len(cands):
                h, r = pool[n % len(pool)]
                orc.score_ref(core, R, S, O, h, r)
                n += 1
            dt = time.perf_counter() - t0
            tried.append((th, n * B / dt))
            if best is None or n * B / dt > best[1]:
                best = (th, n * B / dt, n, dt)
    th, qps, n, dt = best
    one = [q for t, q in tried if t == 1]
    return {"value": qps, "unit": "queries/s", "cores": th, "kind": "port",
            "one_thread_value": one[0] if one else None,
            "sample": f"{n} batches of {B} queries ({dt:.1f} s) of the same workload, torch {torch.__version__} CPU fp32, "
                      f"best of thread counts {[(a, round(b)) for a, b in tried]} ({avail} hardware threads visible); "
                      f"oracle/score_oracle.py::score_ref; the CPU leg runs all five ops for every batch (no table caching): "
                      f"the like-for-like GPU figure is per_batch_ms_per_step"}


def surface_timings(rt, core, R, S, O, sym, pool, n_ent, n_rel, trank, B, workload, dev, n_calls=400):
    """Wall time per call of the reference-protocol surface on the bench operands: `model(h, r)(T)` in eval mode under
    no_grad (relation tables cached by the model, like the reference's evaluate() which scores every batch with one
    extract_tensor(model)) with the out-of-range-id check deferred to the caller's sync point (what evaluate() and the
    training driver do) and with the default per-call check (one stream synchronisation per call); and, for the
    WN18RR workload, rt.evaluate() over the real test split."""
    out = {}
    Model = rt.SymmetricR_TuckER if sym else rt.AsymmetricR_TuckER
    model = Model((n_ent, n_rel), tuple(trank))
    with torch.no_grad():
        model.core.data = core
        model.R.weight.data = R
        if sym:
            model.E.weight.data = S
        else:
            model.S.weight.data, model.O.weight.data = S, O
    model.eval()
    T = (rt.SFTucker(model.core.data, [model.R.weight], num_shared_factors=2, shared_factor=model.E.weight) if sym
         else rt.Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight]))
    with torch.no_grad():
        for mode, key in (("deferred", "closure_ms_per_step"), ("strict", "closure_strict_check_ms_per_step")):
            with rt.index_check(mode):
                for i in range(30):
                    model(*pool[i % len(pool)])(T)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(n_calls):
                    model(*pool[i % len(pool)])(T)
                torch.cuda.synchronize(dev)
                out[key] = (time.perf_counter() - t0) / n_calls * 1e3
        rt.check_device_errors(dev)
    data_dir = os.path.join(ROOT, "data", "WN18RR")
    if workload.startswith("wn18rr") and os.path.exists(os.path.join(data_dir, "test.txt")):
        from r_tucker_amd.data import Data, KG_dataset
        data = Data(data_dir + "/", reverse=True)
        if len(data.entities) == n_ent and len(data.relations) == n_rel:
            test_set = KG_dataset(data, data.test_data, test_set=True)
            for _ in range(3):
                rt.evaluate(model, test_set, batch_size=B)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            n_pass = 10
            for _ in range(n_pass):
                rt.evaluate(model, test_set, batch_size=B)
            torch.cuda.synchronize(dev)
            per_pass = (time.perf_counter() - t0) / n_pass * 1e3
            nb = -(-len(test_set) // B)
            out["evaluate_ms_per_pass"] = per_pass
            out["evaluate_ms_per_batch"] = per_pass / nb
            out["evaluate_batches_per_pass"] = nb
    return out


def launch_children(argv, n):
    """`python bench.py --gpus N` with no torch.distributed environment: start N ranks under torch.distributed.run
    (fresh processes: this one has not initialised a GPU and never will) and hand back their exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the hosts only support dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(args):
    """Hidden `--launcher-selftest` (CPU, gloo): what the children of `launch_children` do up to the first
    collective -- rendezvous, one all-reduce, ONE JSON line from rank 0, exit status -- without a GPU."""
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=2))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "sum_of_ranks_plus_one": float(t.item())}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="wn18rr_asym_r10x200_b512_f32", choices=sorted(WORKLOADS))
    ap.add_argument("--exact", action="store_true", help="exact-fp32 MFMA score kernel instead of split-fp16")
    ap.add_argument("--sigmoid", default=None, choices=["fast", "exact"], help="logistic of the fused epilogue (default: package default)")
    ap.add_argument("--out-dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 workloads: dtype of the score matrix (bf16 = what the reference's bf16 model returns; halves the exchange)")
    ap.add_argument("--tables", choices=["cached", "per-batch"], default="per-batch",
                    help="relation tables: rebuilt in every step (default: the reference's per-batch semantics) or once per "
                         "evaluation pass of %d batches" % EVAL_BATCHES)
    ap.add_argument("--stage1", choices=["auto", "replicated", "split", "relation"], default="auto",
                    help="multi-GPU: query vectors computed by every rank, batch-split + all-gather, or split by relation id + "
                         "all-reduce (auto: relation split for relation rank > 32 with cached tables, else batch split)")
    ap.add_argument("--emulate-ranks", type=int, default=8,
                    help="N = 1, the one-GPU share of configs[4]: also time ONE rank's step of an N-rank run (stage 1 for the "
                         "relations of rank 0 only, pack, score the shard; no collective) -> emulated_per_gpu_ms_per_step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                    help="how the K timed steps reach the GPU: one HIP graph of the K steps captured beforehand and launched "
                         "once inside the timed region, or one call into the C ABI after the other.  auto = graph for a "
                         "single-GPU run of at most 256 steps (a 20-step region is 1 ms of GPU work: the host's launch "
                         "pipeline filling and the event brackets are several per cent of it), eager otherwise; the eager "
                         "figure of the same K steps is reported beside it (eager_ms_per_step)")
    ap.add_argument("--prewarm-ms", type=float, default=1000.0,
                    help="untimed clock ramp before the --warmup steps: the same steps for this long (a 20-step run is "
                         "over before the GPU leaves its idle clocks); reported as config.prewarm_ms")
    ap.add_argument("--no-overlap", action="store_true",
                    help="multi-GPU: run the all-gather of step i before step i+1 starts (default: it overlaps the next step's kernels)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the multi-GPU code path (process group, collectives) with however many ranks there are, even one")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_children(sys.argv[1:], args.gpus))
    if args.launcher_selftest:
        return launcher_selftest(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if args.force_dist and "RANK" not in os.environ:     # one-rank rehearsal started without torchrun
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=3))

    import r_tucker_amd as rt  # noqa: F401
    from r_tucker_amd import _lib, synthetic as gen
    lib = _lib.load()
    from r_tucker_amd import ops as _ops
    sig_mode = args.sigmoid or _ops.DEFAULT_SIGMOID
    sflags = _lib.RTK_SCORE_SIGMOID | (_lib.RTK_SCORE_SIGMOID_FAST if sig_mode == "fast" else 0)

    n_ent, n_rel, B, trank, dtype = WORKLOADS[args.workload]
    a, b, c = trank
    sym = "_sym_" in args.workload
    bf16 = dtype == "bf16"
    n_loc = -(-n_ent // world)
    lo = min(rank * n_loc, n_ent)
    hi = min(lo + n_loc, n_ent)
    if n_ent * trank[1] > 50_000_000:      # big synthetic shapes: generate on the device
        gd = torch.Generator(device=dev).manual_seed(322)
        core = torch.randn(trank, generator=gd, device=dev) * (3.0 / float(np.sqrt(a * b * c)))
        R = torch.randn((n_rel, a), generator=gd, device=dev)
        S = torch.randn((n_ent, b), generator=gd, device=dev)
        O = S if sym else torch.randn((n_ent, c), generator=gd, device=dev)
    else:
        core, R, S, O = [torch.from_numpy(x).to(dev) for x in gen.make_params(n_ent, n_rel, trank, 322, shared=sym)]
    if bf16:
        core, R, S = [x.to(torch.bfloat16) for x in (core, R, S)]
        O = S if sym else O.to(torch.bfloat16)
    dcode = _lib.RTK_BF16 if bf16 else _lib.RTK_F32
    esz = 2 if bf16 else 4
    pool_cpu = [tuple(torch.from_numpy(x) for x in gen.make_queries(n_ent, n_rel, B, 1000 + i)) for i in range(64)]
    pool = [(h.to(dev), r.to(dev)) for h, r in pool_cpu]

    # entity shard of this rank (row block of O); N = 1: the whole matrix.  The subject-lookup matrix S
    # stays replicated.
    if world == 1:
        O_loc = O
    else:
        O_loc = torch.zeros((n_loc, c), dtype=O.dtype, device=dev)
        O_loc[: hi - lo] = O[lo:hi]                   # last shard zero-padded to equal size
        if not sym:
            del O
    # rank p's (B, n_loc) block is slot p of the gather buffer: the kernel writes its block in place;
    # rows start on 128-byte boundaries (r_tucker_amd.ops.ROW_ALIGN, the layout score_1vN allocates;
    # R_TUCKER_AMD_ROW_ALIGN=1 gives the dense (B, n_loc) layout)
    obf = args.out_dtype == "bf16"
    if obf and (not bf16 or args.exact or sig_mode != "fast"):
        raise SystemExit("--out-dtype bf16 needs a bf16 workload and the fast logistic")
    if bf16 and args.exact:
        raise SystemExit("--exact is an fp32 kernel")
    osz = 2 if obf else 4
    ra = _ops.ROW_ALIGN * (4 // osz) if _ops.ROW_ALIGN > 1 else 1
    pitch = -(-n_loc // ra) * ra
    # two gather buffers: the all-gather of step i (RCCL's own stream) overlaps the kernels of step i+1
    n_buf = 2 if (use_dist and not args.no_overlap) else 1
    gathered_all = [torch.empty((world, B, pitch), dtype=torch.bfloat16 if obf else torch.float32, device=dev)
                    for _ in range(n_buf)]
    if obf:
        sflags |= _lib.RTK_SCORE_OUT_BF16
    gathered, out = gathered_all[0], gathered_all[0][rank]
    pending = [None] * n_buf

    # everything runs on ONE side stream (the legacy default stream cannot be captured into a HIP graph: --launch graph)
    torch.cuda.synchronize(dev)             # (the operands were created on the default stream)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    cached = args.tables == "cached"
    split1 = world > 1 and (args.stage1 in ("split", "relation") or (args.stage1 == "auto" and a > 32))
    by_rel = split1 and cached and args.stage1 in ("relation", "auto")
    ftp_fn = lib.rtk_query_vectors_from_tables_part_bf16 if bf16 else lib.rtk_query_vectors_from_tables_part_f32
    ws_bytes = max(lib.rtk_workspace_bytes(dcode, B, n_rel, a, b, c), lib.rtk_from_tables_workspace_bytes(B, n_rel))
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    qp = torch.empty(lib.rtk_packed_query_bytes(dcode, B, c), dtype=torch.uint8, device=dev)
    qv_fn = lib.rtk_query_vectors_bf16 if bf16 else lib.rtk_query_vectors_f32
    ft_fn = lib.rtk_query_vectors_from_tables_bf16 if bf16 else lib.rtk_query_vectors_from_tables_f32
    tb_fn = lib.rtk_relation_tables_bf16 if bf16 else lib.rtk_relation_tables_f32
    sp_fn = lib.rtk_score_packed_bf16 if bf16 else lib.rtk_score_packed_f32
    need_v = args.exact or split1 or (world == 1 and not bf16)     # (the extra exact-fp32 leg of the N = 1 line)
    v = torch.empty((B, c), dtype=torch.float32, device=dev) if need_v else None
    tables = tws = None
    if cached or (world == 1 and not use_dist):
        tables = torch.empty((n_rel, b, c), dtype=torch.float32, device=dev)
        tws = torch.zeros(max(256, lib.rtk_relation_tables_workspace_bytes(dcode, n_rel, a, b, c)), dtype=torch.uint8, device=dev)
    # stage 1 split over the ranks: this rank's slice of the batch, the gathered (B_loc * world, c) vectors
    B_loc = -(-B // world)
    qlo, qhi = min(rank * B_loc, B), min((rank + 1) * B_loc, B)
    v_all = torch.zeros((max(world * B_loc, B), c), dtype=torch.float32, device=dev) if split1 else None

    def stage1(h, r, cached=cached, want_v=args.exact):
        """query vectors of the batch -> packed planes in qp (and/or fp32 v)"""
        if by_rel:
            # stage 1 split by relation id: this rank's relations only, rows of the others stay zero; one all-reduce
            vr = v_all[:B]
            vr.zero_()
            _lib.check(ftp_fn(tables.data_ptr(), n_rel, b, c, S.data_ptr(), n_ent, r.data_ptr(), h.data_ptr(), B, rank, world,
                              vr.data_ptr(), ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors_from_tables_part")
            dist.all_reduce(vr)
            if not args.exact:
                _lib.check(lib.rtk_pack_query_vectors(vr.data_ptr(), B, c, dcode, qp.data_ptr(), sp), "rtk_pack_query_vectors")
            return vr
        if split1:
            nq = qhi - qlo
            mine = v_all[rank * B_loc:(rank + 1) * B_loc]
            if nq > 0:
                hp, rp = h.data_ptr() + 8 * qlo, r.data_ptr() + 8 * qlo
                if cached:
                    _lib.check(ft_fn(tables.data_ptr(), n_rel, b, c, S.data_ptr(), n_ent, rp, hp, nq, mine.data_ptr(), None,
                                     ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors_from_tables")
                else:
                    _lib.check(qv_fn(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_ent, rp, hp, nq,
                                     mine.data_ptr(), None, ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors")
            dist.all_gather_into_tensor(v_all.view(-1), mine.reshape(-1))
            if not args.exact:
                _lib.check(lib.rtk_pack_query_vectors(v_all.data_ptr(), B, c, dcode, qp.data_ptr(), sp), "rtk_pack_query_vectors")
            return v_all
        vo = v.data_ptr() if want_v else None
        qo = None if args.exact else qp.data_ptr()
        if cached:
            _lib.check(ft_fn(tables.data_ptr(), n_rel, b, c, S.data_ptr(), n_ent, r.data_ptr(), h.data_ptr(), B, vo, qo,
                             ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors_from_tables")
        else:
            _lib.check(qv_fn(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_ent, r.data_ptr(), h.data_ptr(), B,
                             vo, qo, ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors")
        return v

    def step_local(i, ev=None, out=out, cached=cached, exact=args.exact, timer=None):
        h, r = pool[i % len(pool)]
        if cached and i % EVAL_BATCHES == 0:     # a new "evaluation pass": the parameters may have changed
            _lib.check(tb_fn(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, tables.data_ptr(), tws.data_ptr(), tws.numel(), sp),
                       "rtk_relation_tables")
        vv = stage1(h, r, cached, exact)
        if ev:
            ev[0].record(stream)
        if exact:
            _lib.check(lib.rtk_score_f32(vv.data_ptr(), B, c, O_loc.data_ptr(), n_loc, out.data_ptr(), pitch,
                                         _lib.RTK_SCORE_SIGMOID, sp), "rtk_score_f32")
        else:
            if timer is not None:                # the kernel's own begin / end (rtk_timer_*)
                _lib.check(lib.rtk_timer_arm(timer), "rtk_timer_arm")
            _lib.check(sp_fn(qp.data_ptr(), B, c, O_loc.data_ptr(), n_loc, out.data_ptr(), pitch, sflags, sp),
                       "rtk_score_packed")
        if ev:
            ev[1].record(stream)

    def step(i, ev=None, timer=None):
        k = i % n_buf
        if pending[k] is not None:          # the gather that last used this buffer must be done before it is rewritten
            pending[k].wait()
            pending[k] = None
        g = gathered_all[k]
        step_local(i, ev, g[rank], timer=timer)
        if use_dist:
            # in place: the input is this rank's slot of the output
            if n_buf == 1:
                dist.all_gather_into_tensor(g.view(-1), g[rank].view(-1))
            else:
                pending[k] = dist.all_gather_into_tensor(g.view(-1), g[rank].view(-1), async_op=True)

    def drain():
        for k in range(n_buf):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def barrier():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.prewarm_ms > 0:     # untimed: bring clocks, TLBs and the code objects to their steady state
        t_end = time.perf_counter() + args.prewarm_ms * 1e-3
        i = 0
        while True:
            for _ in range(16):
                step(i)
                i += 1
            go = time.perf_counter() < t_end
            if use_dist:          # one decision for all ranks: nobody is left alone in a collective
                more = torch.tensor([float(go)], device=dev)
                dist.all_reduce(more, op=dist.ReduceOp.MIN)
                go = more.item() != 0
            else:
                torch.cuda.synchronize(dev)
            if not go:
                break
        barrier()
    for i in range(args.warmup):
        step(i)
    # HIP events bracket the score kernel on its stream on every 8th timed step (an event pair
    # costs a few us of stream time; sampling keeps the timed region representative); short runs
    # (the driver's 20 steps) sample every 4th step: five brackets -- `roofline.kernel_ms` no longer rests on them
    # (the kernel timer behind the timed region does), they are the line's `stream_bracket_ms`
    every = 8 if args.steps >= 200 else 4
    events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) if i % every == 0 else None
              for i in range(args.steps)]
    use_graph = args.launch == "graph" or (args.launch == "auto" and world == 1 and not use_dist and args.steps <= 256)
    if use_graph and use_dist:
        raise SystemExit("--launch graph: single-GPU runs only (the all-gather is not captured)")
    eager_dt = None
    if use_graph:
        # The K steps as ONE HIP graph (the C ABI only enqueues on the given stream: capturable, tests/test_gpu_parity.py),
        # captured and replayed once untimed, then launched once -- onto a drained stream -- between the two barriers.
        # Same kernels, same operands, same order as the eager loop below, which is timed too.
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=stream):
                for i in range(args.steps):
                    step(args.warmup + i)
            graph.replay()
        except Exception as e:               # (--launch auto must not cost the line: fall back to the eager loop)
            if args.launch == "graph":
                raise
            print(f"bench.py: HIP graph capture failed ({e!r}); timing the eager loop", file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize(dev)
    if use_graph:
        barrier()
        t0 = time.perf_counter()
        graph.replay()
        barrier()
        dt = time.perf_counter() - t0
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, events[i])
        barrier()
        eager_dt = time.perf_counter() - t0
    else:
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, events[i])
        barrier()
        dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # out-of-range ids are flagged by the kernels (sticky word in the workspace); synthetic ids never are
    flag = torch.zeros(1, dtype=torch.int32)
    flag.copy_(ws[:4].view(torch.int32))
    assert int(flag) & 1 == 0, "device error word set"

    events = [e for e in events if e is not None]
    # bracket: two events recorded on the stream around the launch (event records + dispatch gap + kernel);
    # kern_ms: the kernel's own begin -> end as the runtime stamps it on the timer's events (hipExtLaunchKernelGGL) --
    # the duration a rocprofv3 kernel trace of the same run reports (include/rtucker_hip.h, rtk_timer_*)
    bracket_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    # kern_ms: the kernel's own begin -> end as the runtime stamps it on a timer's events (rtk_timer_*,
    # hipExtLaunchKernelGGL; include/rtucker_hip.h) -- the duration a rocprofv3 kernel trace of the same run reports.
    # Taken on 16 of 64 further steps right behind the timed region, not inside it: a timed launch costs the stream ~10 us.
    kern_ms, timed = bracket_ms, False
    if not args.exact:
        tms = []
        for _ in range(16):
            tms.append(C.c_void_p())
            _lib.check(lib.rtk_timer_create(C.byref(tms[-1])), "rtk_timer_create")
        for k in range(64):                 # one stream of steps, every fourth launch timed, read after the last
            step(args.warmup + args.steps + k, timer=tms[k // 4] if k % 4 == 3 else None)
        barrier()
        ms, ks = C.c_float(), []
        for tm in tms:
            if lib.rtk_timer_elapsed_ms(tm, C.byref(ms)) == 0:
                ks.append(ms.value)
            lib.rtk_timer_destroy(tm)
        timed = len(ks) == 16               # (a kernel without the timed launch -- RTK_SCORE_KERNEL=v3 -- leaves the bracket)
        if timed:
            kern_ms = float(np.mean(ks))
    # N = 1: the two figures the headline leaves out -- the step with the relation tables
    # rebuilt for EVERY batch (the reference's per-batch semantics), and the exact-fp32 MFMA score kernel
    extras = {}
    if world == 1 and not use_dist and not args.exact:
        n_x = max(10, min(args.steps, 200))
        if cached:
            for i in range(10):
                step_local(i, cached=False)
            barrier()
            t1 = time.perf_counter()
            for i in range(n_x):
                step_local(i, cached=False)
            barrier()
            extras["per_batch_ms_per_step"] = (time.perf_counter() - t1) / n_x * 1e3
        else:
            # the evaluation-pass form: tables of all relations built once per EVAL_BATCHES steps (inside the timed loop)
            for i in range(EVAL_BATCHES):
                step_local(i, cached=True)
            barrier()
            n_c = -(-n_x // EVAL_BATCHES) * EVAL_BATCHES
            t1 = time.perf_counter()
            for i in range(n_c):
                step_local(i, cached=True)
            barrier()
            extras["cached_tables_ms_per_step"] = (time.perf_counter() - t1) / n_c * 1e3
        # the score kernel launched back to back (no stage 1 between, ONE event pair around 64 launches): the bracket of
        # a single launch carries the event pair's own stream time and the dispatch gap in front of the kernel; this
        # figure is the one that a rocprofv3 kernel trace of the same run reports as the kernel's average duration
        if True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(2):
                e0.record(stream)
                for _ in range(64):
                    _lib.check(sp_fn(qp.data_ptr(), B, c, O_loc.data_ptr(), n_loc, out.data_ptr(), pitch, sflags, sp),
                               "rtk_score_packed")
                e1.record(stream)
                barrier()
            extras["score_kernel_back_to_back_ms"] = e0.elapsed_time(e1) / 64
        # the drop-in surface: the eval-mode closure `model(h, r)(T)` of the reference's protocol on the
        # same operands and batches, and `evaluate()` over the WN18RR test split (13 batches of 512; train.py:94-125)
        try:
            extras.update(surface_timings(rt, core, R, S, O, sym, pool, n_ent, n_rel, trank, B, args.workload, dev))
        except Exception as e:       # the headline line must survive a failure of this extra leg
            extras["surface_error"] = repr(e)
        if "shard" in args.workload and tables is not None and args.emulate_ranks > 1:
            _lib.check(tb_fn(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, tables.data_ptr(), tws.data_ptr(), tws.numel(), sp),
                       "rtk_relation_tables")
            # ONE rank's step of an --emulate-ranks-rank run on this GPU's entity shard: stage 1 for the relations of rank 0
            # (rtk_query_vectors_from_tables_part_*), pack, score.  The all-reduce of the B x c vectors (16 MB at configs[4])
            # that a real run adds is NOT in it: an emulation of the per-GPU compute, not a multi-GPU measurement.
            # (The packing and the score kernel run on the COMPLETE vectors of the batch, computed beforehand -- what the
            # all-reduce would have delivered: seven eighths of the rows left at zero would let the score kernel run on zero
            # operands, which the chip clocks ~10 % faster.)
            ve = torch.zeros((B, c), dtype=torch.float32, device=dev)
            ftp = lib.rtk_query_vectors_from_tables_part_bf16 if bf16 else lib.rtk_query_vectors_from_tables_part_f32
            v_full = []
            for i in range(len(pool)):
                h, r = pool[i]
                vf = torch.empty((B, c), dtype=torch.float32, device=dev)
                _lib.check(ft_fn(tables.data_ptr(), n_rel, b, c, S.data_ptr(), n_ent, r.data_ptr(), h.data_ptr(), B, vf.data_ptr(), None,
                                 ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors_from_tables")
                v_full.append(vf)

            def step_emulated(i):
                h, r = pool[i % len(pool)]
                ve.zero_()
                _lib.check(ftp(tables.data_ptr(), n_rel, b, c, S.data_ptr(), n_ent, r.data_ptr(), h.data_ptr(), B, 0,
                               args.emulate_ranks, ve.data_ptr(), ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors_from_tables_part")
                _lib.check(lib.rtk_pack_query_vectors(v_full[i % len(pool)].data_ptr(), B, c, dcode, qp.data_ptr(), sp), "rtk_pack_query_vectors")
                _lib.check(sp_fn(qp.data_ptr(), B, c, O_loc.data_ptr(), n_loc, out.data_ptr(), pitch, sflags, sp), "rtk_score_packed")

            n_e = max(10, min(args.steps, 100))
            for i in range(5):
                step_emulated(i)
            barrier()
            t1 = time.perf_counter()
            for i in range(n_e):
                step_emulated(i)
            barrier()
            extras["emulated_per_gpu_ms_per_step"] = (time.perf_counter() - t1) / n_e * 1e3
            extras["emulated_ranks"] = args.emulate_ranks
        if not bf16 and c <= 512:
            ex = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(12)]
            for e2 in ex:
                step_local(0, e2, cached=cached, exact=True)
            barrier()
            extras["exact_f32_kernel_ms"] = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ex[2:]]))
    gather_ms = None
    if use_dist:      # the exchange on its own (in the timed loop it runs beside the next step's kernels)
        ge = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(10)]
        for e0, e1 in ge:
            e0.record(stream)
            dist.all_gather_into_tensor(gathered.view(-1), out.view(-1))
            e1.record(stream)
        barrier()
        gather_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ge]))
    # algorithmic bytes of ONE score-kernel launch: read the O shard once, write the scores once,
    # read the query vectors once (SURVEY.md 8d formula restricted to this kernel)
    alg_bytes = n_loc * c * esz + B * n_loc * osz + B * c * esz
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # HBM bytes per launch: NOT measured by this run -- taken from the committed rocprofv3 PMC passes
    # (profiles/*_traffic.json, collected by tools/profile_round.sh) when they are for this workload
    traffic = traffic_source = None
    for name in ("r04_traffic.json", "r04_c5_traffic.json", "r03_traffic.json", "r03_c5_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tj = json.load(f)
            if (tj["workload"] == args.workload and world == 1 and not args.exact
                    and (bf16 or score_kernel_name(n_loc, c)[:8] in tj.get("kernel", score_kernel_name(n_loc, c)))):
                traffic, traffic_source = tj["traffic_bytes"], "profiles/" + name
                break
        except (OSError, KeyError, ValueError):
            pass
    result = {
        "metric": "1-vs-N triples scored/sec", "value": args.steps * B / dt, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": args.workload, "entities": n_ent, "relations": n_rel, "rank": list(trank),
                   "batch": B, "scores_per_query": n_ent, "score_row_pitch": pitch, "score_dtype": args.out_dtype,
                   "score_kernel": "exact_f32_mfma" if args.exact else ("bf16_mfma" if bf16 else "split_fp16_mfma"),
                   "sigmoid": "exact" if args.exact else sig_mode,
                   "prewarm_ms": args.prewarm_ms,
                   "launch": ("one HIP graph of the %d steps, launched once inside the timed region" % args.steps) if use_graph
                             else "eager: one C-ABI call after the other",
                   "relation_tables": (f"cached: rebuilt every {EVAL_BATCHES} steps inside the timed region" if cached
                                       else "rebuilt in every step"),
                   "stage1": ("split over ranks by relation id + all-reduce of the query vectors" if by_rel else
                              "batch split over ranks + all-gather of the query vectors" if split1 else "on every rank"),
                   "sharding": "none" if world == 1 else f"entity rows / {world} + RCCL all-gather"},
        "scores_per_s": args.steps * B * n_ent / dt,
        **({"eager_ms_per_step": eager_dt / args.steps * 1e3} if eager_dt is not None else {}),
        **extras,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": ("score_bf16_kernel" if bf16 else score_kernel_name(n_loc, c)) if not args.exact else "gemm_f32_kernel",
                     "kernel_ms": kern_ms, "kernel_ms_source": ("stream event bracket" if (args.exact or not timed) else
                                                                 "kernel begin/end events (rtk_timer_*, hipExtLaunchKernelGGL)"),
                     "stream_bracket_ms": bracket_ms, "algorithmic_bytes": alg_bytes},
    }
    if use_dist:
        # the exchange step: every rank receives (P-1) blocks of B*n_loc scores over xGMI
        # (7 links x ~153 GB/s per GPU, MI355X guide); reported next to the shard-local rate
        recv = (world - 1) * B * pitch * osz
        result["exchange"] = {"collective": "all_gather_into_tensor (RCCL, in place)", "ms": gather_ms,
                              "overlapped_with_next_step": n_buf == 2,
                              "bytes_received_per_gpu": recv, "achieved_GBps": recv / (gather_ms * 1e-3) / 1e9,
                              "xgmi_peak_GBps": 7 * 153.0, "frac": recv / (gather_ms * 1e-3) / 1e9 / (7 * 153.0),
                              "shard_local_queries_per_s": B / (kern_ms * 1e-3)}
    if use_dist:
        # The same scores consumed shard-locally: filtered rank of a queried object per
        # query with NO gather -- two all-reduces of B words instead of (P-1)*B*n_loc*4 bytes.
        try:
            if obf:
                raise RuntimeError("the ranking kernels read fp32 scores (run without --out-dtype bf16)")
            from r_tucker_amd.evaluation import rank_counts_block, target_scores_block
            n_valid = max(0, hi - lo)
            objs = [torch.randint(0, n_ent, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(7 + i))
                    for i in range(8)]

            def step_ranked(i):
                step_local(i)
                obj = objs[i % len(objs)]
                if n_valid > 0:
                    blk = out[:, :n_valid]
                    pt = target_scores_block(blk, obj, lo)
                else:                                  # a shard that is all padding owns no entity
                    pt = torch.full((B,), float("-inf"), dtype=torch.float32, device=dev)
                dist.all_reduce(pt, op=dist.ReduceOp.MAX)
                counts = (rank_counts_block(blk, obj, lo, pt) if n_valid > 0
                          else torch.zeros(B, dtype=torch.int32, device=dev))
                dist.all_reduce(counts, op=dist.ReduceOp.SUM)
                return counts

            nr = max(1, min(args.steps, 200))
            for i in range(min(args.warmup, 20)):
                step_ranked(i)
            barrier()
            t1 = time.perf_counter()
            for i in range(nr):
                step_ranked(args.warmup + i)
            barrier()
            dtr = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(dtr, op=dist.ReduceOp.MAX)
            dtr = float(dtr.item())
            result["ranked_without_gather"] = {
                "what": "stage 1 + shard-local scores + filtered rank counts; all-reduce(MAX) of B floats and "
                        "all-reduce(SUM) of B int32 per step, no score exchange",
                "queries_per_s": nr * B / dtr, "ms_per_step": dtr / nr * 1e3, "steps": nr,
                "collective_bytes_per_step": 8 * B}
        except Exception as e:       # the headline line above must survive a failure of this extra leg
            result["ranked_without_gather"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not bf16 and not args.force_dist:
        result["cpu_baseline"] = cpu_baseline(gen, n_ent, n_rel, B, trank, pool_cpu)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
