"""One exponent per GPU: shard a worktodo over the ranks of a torch.distributed job.

The squaring path does not span devices (one transform never leaves its GPU; the
reference itself is one process per device, `-d`, src/io/CliParser.cpp:198,586, and walks its worktodo one
entry per process, src/modes/RunPrpOrLlMarin.cpp:727-751).  Ranks are independent replicas; the only
collective is the reduction of a 3 x int64 status word
{all Gerbicz checks passed (MIN), Gerbicz errors (SUM), iterations done (SUM)} -- at every Gerbicz-check boundary
when all entries share one check cadence (same number of boundaries on every rank: the BASELINE configs[4] case),
and always at exit -- plus an all_gather of the per-exponent results at exit.  RCCL over xGMI on the GPUs
(backend "nccl"), gloo in the CPU tests.

Command line (one process per GPU, rendezvous before any GPU call):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
      -m prmers_amd.launch --worktodo worktodo.txt [--per-gpu 2] [--results results.json.txt] [--max-iters N] [--fft crt:9]
"""
import argparse
import json
import math
import os
import signal
import sys
import threading

import torch
import torch.distributed as dist

from . import prp


def check_boundaries(entries, checklevel):
    """Gerbicz-Li checks each entry will perform (RunPrpOrLlMarin.cpp:258,338-409): a block boundary at every multiple
    of B = floor(sqrt(p)) of the remaining-iteration counter plus the last iteration; a check at every
    `checklevel`-th boundary and at the last one.  LL entries have none."""
    out = []
    for mode, p in entries:
        if mode != "prp":
            out.append(0)
            continue
        nb = (p - 1) // max(int(math.sqrt(p)), 1) + 1
        lvl = max(checklevel, 1)
        out.append(nb // lvl + (1 if nb % lvl else 0))
    return out


class StatusReducer:
    """all_reduce of {ok (MIN), errors (SUM), iterations (SUM)}; a no-op without a process group.  With a process group it always goes
    through the backend, also at world size 1 (one rank on one GPU: the same RCCL calls an eight-rank run issues)."""

    def __init__(self, device="cpu"):
        self.device = device
        self.on = dist.is_initialized()
        self.lock = threading.Lock()

    def reduce(self, ok, errors, iters):
        if not self.on:
            return {"all_ok": int(ok), "gerbicz_errors": int(errors), "iterations": int(iters)}
        with self.lock:
            mn = torch.tensor([int(ok)], dtype=torch.int64, device=self.device)
            sm = torch.tensor([int(errors), int(iters)], dtype=torch.int64, device=self.device)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            return {"all_ok": int(mn[0]), "gerbicz_errors": int(sm[0]), "iterations": int(sm[1])}


def run_sharded(worktodo_lines, make_engine, device="cpu", max_iters=None, checklevel=0, log=None, per_gpu=1,
                sync_checks="auto", on_status=None, should_stop=None, ckpt_dir=None, backup_interval_s=None, **run_kwargs):
    """Each rank runs its share of the worktodo (line i -> rank i mod world) with make_engine(exponent) -> engine;
    returns (results_of_all_ranks, status) on every rank.

    status["all_ok"] is 1 only if every rank finished every entry without an exception and without a failed
    Gerbicz-Li check.  The number of reductions a rank issues never depends on what happens to it: at check boundaries
    it reduces once per FIRST visit of a boundary (a failed check rolls the entry back and meets the same boundaries
    again: those visits do not reduce), and an entry that ends early -- exception, interrupt -- issues the reductions it
    still owes before the rank moves on, so the collectives of all ranks stay matched.
    per_gpu > 1 runs that many entries of the rank concurrently (threads, one engine and stream each).
    sync_checks: True / False / "auto" (reduce at check boundaries when every rank has the same number of them,
    per_gpu == 1 and no checkpoint directory is in play: a resumed entry has fewer boundaries left than a fresh one);
    on_status(status) is called on every rank after each reduction.
    should_stop(): polled by every entry before each iteration (SIGINT / SIGTERM in main): the entry checkpoints into
    ckpt_dir and returns, the rank skips its remaining entries, and the launcher still exits through the reductions."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    entries = [e for e in (prp.parse_worktodo_line(l) for l in worktodo_lines) if e]
    mine = entries[rank::world]
    reducer = StatusReducer(device)

    # boundary reductions only when every rank will reach the same number of them (else at exit only)
    per_rank_checks = [sum(check_boundaries(entries[r::world], checklevel)) for r in range(world)]
    lockstep = (sync_checks is True) or (sync_checks == "auto" and per_gpu == 1 and max_iters is None and checklevel > 0 and
                                         ckpt_dir is None and dist.is_initialized() and len(set(per_rank_checks)) == 1)
    state = {"ok": 1, "errors": 0, "iters": 0}
    state_lock = threading.Lock()
    results = []

    def one(mode, p):
        base_iters = [0]
        owed = check_boundaries([(mode, p)], checklevel)[0] if lockstep else 0   # boundary reductions of this entry
        visited = [0]                                                             # highest boundary already reduced

        def reduce_now():
            with state_lock:
                snap = dict(state)
            st = reducer.reduce(snap["ok"], snap["errors"], snap["iters"])
            if on_status:
                on_status(st)

        def on_check(passed, it):
            nonlocal owed
            with state_lock:
                if not passed:
                    state["ok"] = 0
                    state["errors"] += 1
                state["iters"] += max(it - base_iters[0], 0)
                base_iters[0] = max(it, base_iters[0])
            if lockstep and owed > 0 and it > visited[0]:
                visited[0] = it
                owed -= 1
                reduce_now()

        r = None
        try:
            eng = make_engine(p)
            try:
                kw = dict(run_kwargs)
                if ckpt_dir is not None:
                    kw.update(ckpt_path=prp.checkpoint_name(p, mode, ckpt_dir), backup_interval_s=backup_interval_s)
                r = prp.run_prp_or_ll(eng, p, mode, max_iters=max_iters, checklevel=checklevel, log=log, on_check=on_check,
                                      should_stop=should_stop, **kw)
            finally:
                eng.close()
        except Exception as exc:   # a failing entry must not strand the other ranks in a collective
            r = {"exponent": p, "mode": mode, "is_prime": False, "res64": "", "res2048": "", "iterations": base_iters[0], "gerbicz_checks": 0,
                 "gerbicz_errors": 0, "complete": False, "state": None, "error": "%s: %s" % (type(exc).__name__, exc)}
        r["rank"] = rank
        with state_lock:
            state["iters"] += max(r["iterations"] - base_iters[0], 0)
            if r.get("error") or (max_iters is None and not r["complete"] and not r.get("interrupted")):
                state["ok"] = 0
            results.append(r)
        while owed > 0:       # the entry ended before its last boundary: the peers still expect these reductions
            owed -= 1
            reduce_now()

    def skipped(mode, p):   # entries a rank does not start after an interrupt: they keep their place in the worktodo
        results.append({"exponent": p, "mode": mode, "is_prime": False, "res64": "", "res2048": "", "iterations": 0, "gerbicz_checks": 0,
                        "gerbicz_errors": 0, "complete": False, "state": None, "interrupted": True, "rank": rank})
        for _ in range(check_boundaries([(mode, p)], checklevel)[0] if lockstep else 0):
            reducer.reduce(state["ok"], state["errors"], state["iters"])

    if per_gpu <= 1:
        for mode, p in mine:
            if should_stop is not None and should_stop():
                skipped(mode, p)
            else:
                one(mode, p)
    else:
        queue = list(mine)
        qlock = threading.Lock()

        def worker():
            while True:
                with qlock:
                    if not queue:
                        return
                    mode, p = queue.pop(0)
                if should_stop is not None and should_stop():
                    with state_lock:
                        skipped(mode, p)
                else:
                    one(mode, p)
        threads = [threading.Thread(target=worker) for _ in range(min(per_gpu, max(len(mine), 1)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()

    status = reducer.reduce(state["ok"], state["errors"], state["iters"])   # always: the reduction at exit
    if on_status:
        on_status(status)
    if dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        results = [r for part in gathered for r in part]
    status["check_boundary_reductions"] = bool(lockstep)
    return results, status


def init_from_env(backend):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        dist.init_process_group(backend)


def visible_gpu_count():
    """GPUs this process may use, counted without initialising the HIP runtime: the KFD topology nodes that have SIMDs (CPU agents have
    none), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them is set."""
    import glob
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(props):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            ids = [x for x in val.split(",") if x.strip() != ""]
            n = min(n, len(ids)) if n else 0
            break
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m prmers_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--worktodo", required=True, help="worktodo.txt (PRP= / PRPDC= / Test= / DoubleCheck= lines)")
    ap.add_argument("--per-gpu", type=int, default=1, help="entries run concurrently on one GPU (two fill each other's launch gaps)")
    ap.add_argument("--results", default="results.json.txt", help="one PrimeNet-style JSON line per finished entry (rank 0 writes)")
    ap.add_argument("--max-iters", type=int, default=None, help="stop every entry after this many iterations (benchmarks)")
    ap.add_argument("--checklevel", type=int, default=0, help="Gerbicz-Li checks every this many block boundaries (0: the reference's automatic rule)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo")
    ap.add_argument("--fft", default=None, help="transform spec handed to every engine: a plan (m2=..,c=..) or crt[:odd][:words=N] for the "
                    "GF(M61^2) x GF(M31^2) family with a prime-factor axis (the reference's -fft); a bare `crt` picks the radix by the "
                    "reference's 1.30 / 1.60 size-ratio gates on THIS engine's capacity rule (~34 bits per word: README.md)")
    ap.add_argument("--ckpt-dir", default=None, help="directory of the per-exponent checkpoint files (resume on start, save every "
                    "--backup-interval seconds and on SIGINT / SIGTERM); default: no checkpoints")
    ap.add_argument("--backup-interval", type=float, default=300.0, help="seconds between checkpoints (the reference's -t)")
    ap.add_argument("--dry-run", action="store_true", help="print the shard of every rank and exit (no GPU, no process group)")
    args = ap.parse_args(argv)

    lines = open(args.worktodo).read().splitlines()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        entries = [e for e in (prp.parse_worktodo_line(l) for l in lines) if e]
        for r in range(world):
            print(json.dumps({"rank": r, "entries": [{"mode": m, "exponent": p} for m, p in entries[r::world]]}))
        return 0

    # rendezvous first: nothing below may touch the GPU before the process group exists -- so the devices are counted WITHOUT the HIP
    # runtime (torch.cuda.device_count() only avoids it when its amdsmi path is available): the KFD topology lists the GPU agents
    device = "cpu"
    ndev = visible_gpu_count()
    if ndev == 0:
        sys.stderr.write("error: no GPU visible (no KFD GPU node / empty HIP_VISIBLE_DEVICES): the MI355X engine has no CPU fallback\n")
        return 2
    gpu = local_rank % ndev                    # more ranks than GPUs: rank i runs on device i mod N (two tests fill each other's launch gaps)
    if world > 1 or os.environ.get("MI355_FORCE_PROCESS_GROUP") == "1":   # (the variable: a one-rank group, to rehearse the RCCL path on one GPU)
        if args.backend == "nccl":
            if int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) > ndev:
                sys.stderr.write("error: %d ranks on %d GPU(s): RCCL takes one rank per device; use --backend gloo for the status word\n" % (world, ndev))
                return 2
            torch.cuda.set_device(gpu)
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
            device = "cuda"
        else:
            dist.init_process_group(args.backend)
    from . import Engine   # the HIP engine; there is no CPU engine on the product path

    def log(msg):
        sys.stderr.write("[rank %d] %s\n" % (rank, msg))

    def on_status(st):
        if rank == 0:
            sys.stderr.write("[status] %s\n" % json.dumps(st))

    # SIGINT / SIGTERM: every running entry checkpoints at its next iteration and returns (RunPrpOrLlMarin.cpp:296-309);
    # the rank then goes through the same exit reductions as a finished one
    stop = threading.Event()
    for sig in (signal.SIGINT, signal.SIGTERM):
        signal.signal(sig, lambda *_: stop.set())
    try:
        results, status = run_sharded(lines, lambda p: Engine(p, prp.REGISTERS, device=gpu, plan=args.fft), device=device, max_iters=args.max_iters,
                                      checklevel=args.checklevel, log=log, per_gpu=args.per_gpu, on_status=on_status, should_stop=stop.is_set,
                                      ckpt_dir=args.ckpt_dir, backup_interval_s=args.backup_interval)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    if rank == 0:
        with open(args.results, "a") as f:
            for r in sorted(results, key=lambda r: r["exponent"]):
                if r.get("complete"):
                    f.write(prp.result_json(r, r.get("fft_length", 0)) + "\n")
        print(json.dumps({"status": status, "results": [{k: r[k] for k in ("exponent", "mode", "is_prime", "res64", "iterations", "complete", "interrupted", "rank") if k in r} |
                                                         ({"error": r["error"]} if r.get("error") else {}) for r in results]}))
    return 0 if status["all_ok"] == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
