"""world_size-2 runs of the one-exponent-per-rank launcher on CPU (gloo), engines backed by the oracle:
sharding, the status word, reductions at the Gerbicz-check boundaries, and a failing rank."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKTODO = ["PRP=1,2,127,-1", "PRP=1,2,521,-1", "Test=607", "PRP=1,2,1001,-1"]


def _worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import orc
    from prmers_amd import launch, prp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []
        if scenario == "plain":
            res, status = launch.run_sharded(WORKTODO, lambda p: orc.OracleEngine(p, prp.REGISTERS), checklevel=1)
        elif scenario == "lockstep":   # 521 and 523 have the same number of check boundaries: reduced at every one of them
            res, status = launch.run_sharded(["PRP=1,2,521,-1", "PRP=1,2,523,-1"], lambda p: orc.OracleEngine(p, prp.REGISTERS), checklevel=1,
                                             on_status=seen.append)
        elif scenario == "engine_fails":   # rank 1's engine cannot be created
            def make(p):
                if p == 521:
                    raise RuntimeError("no HIP device available")
                return orc.OracleEngine(p, prp.REGISTERS)
            res, status = launch.run_sharded(WORKTODO, make, checklevel=1)
        elif scenario == "check_fails":   # an injected error on one entry: caught by Gerbicz-Li, rolled back, and reported
            res, status = launch.run_sharded(["PRP=1,2,9941,-1", "PRP=1,2,127,-1"], lambda p: orc.OracleEngine(p, prp.REGISTERS),
                                             checklevel=1, erroriter=(9900 if rank == 0 else 0))
        elif scenario == "lockstep_error":   # a failed check on ONE rank under boundary reductions (rollback repeats its boundaries)
            res, status = launch.run_sharded(["PRP=1,2,521,-1", "PRP=1,2,523,-1"], lambda p: orc.OracleEngine(p, prp.REGISTERS), checklevel=1,
                                             on_status=seen.append, erroriter=(500 if rank == 0 else 0))
        elif scenario == "lockstep_raise":   # an engine that dies in the middle of rank 1's entry: rank 0 keeps reducing at its boundaries
            class Dying(orc.OracleEngine):
                count = 0

                def square_mul(self, r, a=1):
                    Dying.count += 1
                    if Dying.count > 300:
                        raise RuntimeError("HIP error: device lost")
                    return orc.OracleEngine.square_mul(self, r, a)
            res, status = launch.run_sharded(["PRP=1,2,521,-1", "PRP=1,2,523,-1"],
                                             lambda p: (Dying if p == 523 else orc.OracleEngine)(p, prp.REGISTERS), checklevel=1, on_status=seen.append)
        elif scenario == "interrupt":   # a stop request on both ranks: checkpoint, clean return; a second launch resumes and finishes
            import tempfile
            d = os.path.join(tempfile.gettempdir(), "mi355_gloo_ckpt_%d" % port)
            os.makedirs(d, exist_ok=True)
            polls = [0]

            def stop():
                polls[0] += 1
                return polls[0] > 12
            res1, st1 = launch.run_sharded(WORKTODO, lambda p: orc.OracleEngine(p, prp.REGISTERS), checklevel=1, should_stop=stop, ckpt_dir=d)
            assert st1["all_ok"] == 1 and all((not r["complete"]) for r in res1 if r["exponent"] != 127), (st1, res1)
            assert any(r.get("interrupted") for r in res1)
            res, status = launch.run_sharded(WORKTODO, lambda p: orc.OracleEngine(p, prp.REGISTERS), checklevel=1, ckpt_dir=d)
            status["first_run_iterations"] = st1["iterations"]
        q.put((rank, [(r["exponent"], r["mode"], r["is_prime"], r["rank"], r.get("error")) for r in res], status, len(seen)))
    finally:
        dist.destroy_process_group()


def _run(scenario, timeout=300):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def test_two_ranks_share_a_worktodo():
    expected = sorted([(127, "prp", True, 0, None), (607, "ll", True, 0, None), (521, "prp", True, 1, None), (1001, "prp", False, 1, None)])
    for rank, res, status, _ in _run("plain"):
        assert sorted(res) == expected                      # every rank sees every result
        assert status["all_ok"] == 1 and status["gerbicz_errors"] == 0
        assert status["iterations"] == 127 + 521 + 605 + 1001
        assert status["check_boundary_reductions"] is False   # the shares have different numbers of checks: exit only


def test_status_is_reduced_at_every_check_boundary_when_the_cadence_matches():
    for rank, res, status, nseen in _run("lockstep"):
        assert status["all_ok"] == 1 and status["check_boundary_reductions"] is True
        assert status["iterations"] == 521 + 523
        assert nseen == 24 + 1                               # 24 Gerbicz-Li checks each (B = 22) + the reduction at exit


def test_a_failing_rank_clears_all_ok_and_nobody_hangs():
    for rank, res, status, _ in _run("engine_fails"):
        assert status["all_ok"] == 0
        errs = [r for r in res if r[4]]
        assert len(errs) == 1 and errs[0][0] == 521 and "no HIP device" in errs[0][4]
        assert sorted(r[0] for r in res) == [127, 521, 607, 1001]   # the other entries still ran and were gathered


def test_a_failed_gerbicz_check_is_counted_and_clears_all_ok():
    for rank, res, status, _ in _run("check_fails"):
        assert status["gerbicz_errors"] == 1 and status["all_ok"] == 0
        assert (9941, "prp", True, 0, None) in res              # the run recovered from the last good state


def test_a_failed_check_on_one_rank_keeps_the_boundary_reductions_matched():
    for rank, res, status, nseen in _run("lockstep_error"):
        assert status["check_boundary_reductions"] is True and status["gerbicz_errors"] == 1 and status["all_ok"] == 0
        assert nseen == 24 + 1                               # same count as without the fault: re-visited boundaries do not reduce
        assert sorted((r[0], r[2]) for r in res) == [(521, True), (523, False)]


def test_an_entry_that_dies_mid_run_still_issues_the_reductions_it_owes():
    for rank, res, status, nseen in _run("lockstep_raise"):
        assert status["check_boundary_reductions"] is True and status["all_ok"] == 0 and nseen == 24 + 1
        errs = [r for r in res if r[4]]
        assert len(errs) == 1 and errs[0][0] == 523 and "device lost" in errs[0][4]
        assert (521, "prp", True, 0, None) in res


def test_interrupt_checkpoints_every_rank_and_a_second_launch_resumes():
    expected = sorted([(127, "prp", True, 0, None), (607, "ll", True, 0, None), (521, "prp", True, 1, None), (1001, "prp", False, 1, None)])
    for rank, res, status, _ in _run("interrupt"):
        assert sorted(res) == expected and status["all_ok"] == 1
        # the 13th poll of a rank stops it (one poll per entry + one per run of plain iterations, prp.py: up to the next Gerbicz-Li boundary,
        # 256 at most): rank 0 finished 127 and got part of the way into 607, rank 1 part of the way into 521 and never started 1001; the
        # second launch resumes both from their checkpoints (127 has no checkpoint after its last iteration: it runs again)
        total = 605 + 521 + 1001 + 127
        assert 127 < status["first_run_iterations"] < 127 + 605 + 521
        assert status["iterations"] == total     # every entry reports the iteration it ended on


def test_launcher_command_line_dry_run(tmp_path):
    wt = tmp_path / "worktodo.txt"
    wt.write_text("\n".join(WORKTODO) + "\n# comment\nPfactor=1,2,999,-1,70,2\n")
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "prmers_amd.launch", "--worktodo", str(wt), "--dry-run"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    shards = [json.loads(l) for l in out.stdout.splitlines()]
    assert [e["exponent"] for e in shards[0]["entries"]] == [127, 607] and [e["exponent"] for e in shards[1]["entries"]] == [521, 1001]


def _torchrun(tmp_path, *args, timeout=300):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_stub_entry.py"), *args]
    return subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="1"))


def test_launcher_refuses_to_start_without_a_gpu(tmp_path):
    """the launcher counts its devices without the HIP runtime (KFD topology) and fails with a clear message when none is
    visible, instead of creating engines on `local_rank % 0`."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    wt = tmp_path / "worktodo.txt"
    wt.write_text("PRP=1,2,127,-1\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "prmers_amd.launch", "--worktodo", str(wt), "--backend", "gloo"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode == 2 and "no GPU visible" in out.stderr, out.stdout + out.stderr
    from prmers_amd import launch
    assert launch.visible_gpu_count() == 0


def test_the_torchrun_command_line_of_the_launcher_end_to_end(tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... -m prmers_amd.launch --worktodo ...` with the
    engine replaced by the oracle (tests/dist_stub_entry.py): rendezvous, sharding, reductions, gather, the result file with real fft lengths"""
    wt = tmp_path / "worktodo.txt"
    wt.write_text("\n".join(WORKTODO) + "\n")
    out = _torchrun(tmp_path, "launch", "--worktodo", str(wt), "--backend", "gloo", "--checklevel", "1", "--results", str(tmp_path / "results.json.txt"))
    assert out.returncode == 0, out.stdout + out.stderr
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert summary["status"]["all_ok"] == 1 and summary["status"]["iterations"] == 127 + 521 + 605 + 1001
    assert sorted((r["exponent"], r["rank"], r["is_prime"]) for r in summary["results"]) == [(127, 0, True), (521, 1, True), (607, 0, True), (1001, 1, False)]
    lines = [json.loads(l) for l in (tmp_path / "results.json.txt").read_text().splitlines()]
    assert sorted(l["exponent"] for l in lines) == [127, 521, 607, 1001] and all(l["fft-length"] > 0 for l in lines)


def test_the_torchrun_command_line_of_the_bench_end_to_end(tmp_path):
    """`python -m torch.distributed.run ... bench.py --gpus 2 --steps K --warmup W` with a timing stub for the engine: one JSON line from
    rank 0, whole-job value, max-over-ranks time, one exponent per rank listed"""
    out = _torchrun(tmp_path, "bench", "--gpus", "2", "--steps", "50", "--warmup", "5", "--dist-backend", "gloo", "--preheat-seconds", "0")
    assert out.returncode == 0, out.stdout + out.stderr
    js = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1
    d = json.loads(js[0])
    assert d["n_gpus"] == 2 and d["steps"] == 50 and d["warmup"] == 5 and d["scaling"] == "weak" and d["unit"] == "iter/s"
    assert abs(d["value"] - 2 * 50 / (d["ms_per_step"] * 50e-3)) < 1e-2 * d["value"]
    ranks = d["config"]["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["exponent"] for r in ranks] == [136279841, 136279879]
    assert max(r["ms_per_step"] for r in ranks) <= d["ms_per_step"] + 1e-6
    assert d["config"]["status_reduction_backend"] == "gloo" and "cpu_baseline" not in d
