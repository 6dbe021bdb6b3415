"""Test harness (not product code): runs prmers_amd.launch.main or bench.main under torchrun on CPU with the engine replaced --
launch: by the oracle-backed engine of tests/orc.py, bench: by a timing stub -- so that the command lines of DESIGN.md section 6
(rendezvous, sharding, the status reduction, the gathers, the JSON line) are exercised end to end without a GPU.
usage: python -m torch.distributed.run ... tests/dist_stub_entry.py launch|bench <arguments of that program>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None

import prmers_amd  # noqa: E402

which, argv = sys.argv[1], sys.argv[2:]
if which == "launch":
    import orc
    prmers_amd.Engine = lambda p, regs, device=0, plan=None: orc.OracleEngine(p, regs)
    from prmers_amd import launch
    launch.visible_gpu_count = lambda: 1   # (the stub engine needs no GPU; the real launcher refuses to start without one)
    sys.exit(launch.main(argv))

if which == "bench":
    class StubEngine:
        """the surface bench.py uses; a squaring 'takes' 20 us"""
        def __init__(self, p, regs, device=0, plan=None):
            self.p, self.n = p, 8388608

        def set_digits(self, r, d): pass
        def copy(self, a, b): pass
        def is_equal(self, a, b): return True
        def res64(self, r): return 1
        def close(self): pass

        def time_square_mul(self, reg, iters, a=1, sub=0, per_kernel=False):
            time.sleep(iters * 20e-6)
            k = {"k_front": 0.005, "k_middle": 0.01, "k_back": 0.005, "k_carry_fix": -1.0, "k_sub_small": -1.0, "event_overhead": 0.0}
            return iters * 0.02, (k if per_kernel else {})
    prmers_amd.Engine = StubEngine
    sys.argv = ["bench.py"] + argv
    import bench
    bench.main()
    sys.exit(0)
raise SystemExit("unknown program " + which)
