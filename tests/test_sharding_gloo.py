"""world_size-2 run of the one-exponent-per-rank launcher on CPU (gloo), engines backed by the oracle."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKTODO = ["PRP=1,2,127,-1", "PRP=1,2,521,-1", "Test=607", "PRP=1,2,1001,-1"]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import orc
    from prmers_amd import launch, prp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res, status = launch.run_sharded(WORKTODO, lambda p: orc.OracleEngine(p, prp.REGISTERS), checklevel=1)
        q.put((rank, [(r["exponent"], r["mode"], r["is_prime"], r["rank"]) for r in res], status))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_a_worktodo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expected = sorted([(127, "prp", True, 0), (607, "ll", True, 0), (521, "prp", True, 1), (1001, "prp", False, 1)])
    for rank, res, status in out:
        assert sorted(res) == expected                      # every rank sees every result
        assert status["all_ok"] == 1 and status["gerbicz_errors"] == 0
        assert status["iterations"] == 127 + 521 + 605 + 1001
