"""One exponent per GPU: shard a worktodo over the ranks of a torch.distributed job.

The squaring path does not span devices (one transform never leaves its GPU; the
reference itself is one process per device, `-d`, src/io/CliParser.cpp:198).  Ranks are independent
replicas; the only collective is the reduction of a 3 x int64 status word
{all Gerbicz checks passed (MIN), Gerbicz errors (SUM), iterations done (SUM)} plus an all_gather of
the per-exponent results -- RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.
"""
import os

import torch
import torch.distributed as dist

from . import prp


def run_sharded(worktodo_lines, make_engine, device="cpu", max_iters=None, checklevel=0, log=None):
    """Each rank runs its share of the worktodo (line i -> rank i mod world) with
    make_engine(exponent) -> engine; returns (results_of_all_ranks, status) on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = prp.shard_worktodo(worktodo_lines, rank, world)
    results = []
    ok, errors, iters = 1, 0, 0
    for mode, p in mine:
        eng = make_engine(p)
        try:
            r = prp.run_prp_or_ll(eng, p, mode, max_iters=max_iters, checklevel=checklevel, log=log)
        finally:
            eng.close()
        r["rank"] = rank
        results.append(r)
        errors += r["gerbicz_errors"]
        iters += r["iterations"]
    status = torch.tensor([ok, errors, iters], dtype=torch.int64, device=device)
    if world > 1:
        mn = status[:1].clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        sm = status[1:].clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        status = torch.cat([mn, sm])
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        results = [r for part in gathered for r in part]
    return results, {"all_ok": int(status[0]), "gerbicz_errors": int(status[1]), "iterations": int(status[2])}


def init_from_env(backend):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        dist.init_process_group(backend)
