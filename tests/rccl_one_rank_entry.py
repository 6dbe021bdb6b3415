"""Child process of tests/test_gpu_rccl.py (not product code): a ONE-rank RCCL process group on cuda:0 -- created before anything else
touches the GPU, as prmers_amd/launch.py does -- then the sharded launcher on two small exponents with the status word reduced on the
device, i.e. the exact collectives an eight-rank run of BASELINE configs[4] issues (all_reduce MIN / SUM of int64 words at every
Gerbicz-Li boundary and at exit, all_gather_object of the results), over RCCL.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29517")
    try:
        dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))   # first GPU-touching call
        probe = torch.ones(3, dtype=torch.int64, device="cuda")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as exc:   # RCCL refuses a one-rank communicator on this box: say why, the parent skips with that reason
        print(json.dumps({"rccl_unavailable": "%s: %s" % (type(exc).__name__, exc)}))
        return 0
    from prmers_amd import Engine, launch, prp
    seen = []
    lines = ["PRP=1,2,9941,-1", "PRP=1,2,9949,-1"]
    try:
        results, status = launch.run_sharded(lines, lambda p: Engine(p, prp.REGISTERS, device=0), device="cuda", checklevel=1,
                                             on_status=seen.append)
    finally:
        dist.destroy_process_group()
    print(json.dumps({"backend": "nccl", "status": status, "reductions": len(seen),
                      "results": [{k: r[k] for k in ("exponent", "is_prime", "res64", "complete", "gerbicz_errors", "rank")} for r in results]}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
