#!/usr/bin/env python3
"""Headline benchmark: Marin-path modular squaring x <- x^2 mod 2^p-1 at p ~ 136M on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one square_mul on every rank's own exponent (independent residues, one exponent per GPU,
no data-path collective: SURVEY.md 8e).  The residue vector is resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- dominant kernel: algorithmic bytes per launch (16*n: one read+write sweep of the
                  8-byte residue vector, SURVEY.md 8d: 48*n per squaring = 3 sweeps) / its average
                  duration from HIP events on the engine's stream (one event between kernels, minus the
                  measured cost of an event record: agrees with rocprofv3 --kernel-trace), against the
                  8 TB/s HBM peak
  roofline.valu -- the binding roof as a number: VALU instructions and issue cycles per wave of each kernel from the ISA walk
                  (tools/valu_model.py -> profiles/valu_model_latest.json: instruction counts of the built kernels, no timing in it),
                  predicted_us = waves per SIMD x cycles / the shader clock READ DURING THIS RUN (hwmon sclk sampled while a
                  batch of squarings runs), frac = predicted / measured per kernel and for the squaring; without a readable
                  clock the time prediction is left out (cycles only)
  preheat      -- untimed squarings before --warmup until the GPU clock has ramped (a cold box runs the
                  first tens of milliseconds slower; --steps/--warmup keep their meaning)
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm; the reference has no CPU
                  implementation of this path) timed on this box's host cores, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs[2] / metric: p = 136279841 (n = 2^23); one distinct prime exponent of the same
# transform size per GPU for the multi-GPU runs (worktodo-style sharding, configs[4])
EXPONENTS = [136279841, 136279879, 136279901, 136279919, 136279933, 136279967, 136279981, 136279987]
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0   # the guide's measured copy ceiling (SURVEY.md 8d asks for both)


def seeded_digits(p, n, seed):
    """digits uniform in [0, 2^width): distribution-equivalent to a mid-run residue."""
    import numpy as np
    j = np.arange(n + 1, dtype=np.uint64)
    ceil = (j * np.uint64(p) + np.uint64(n - 1)) // np.uint64(n)
    width = (ceil[1:] - ceil[:-1]).astype(np.uint64)
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 1 << 62, n, dtype=np.uint64) & ((np.uint64(1) << width) - np.uint64(1))
    return d | (width << np.uint64(32))


def _sclk_files(device_index):
    """hwmon frequency inputs labelled sclk of the GPU behind HIP device `device_index` (one per XCD on multi-die parts)."""
    import glob
    import torch
    roots = []
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        roots = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
    except Exception:
        roots = []
    if not roots:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        roots = cards[device_index:device_index + 1] if len(cards) > device_index else cards[:1]
    files = []
    for r in roots:
        for lab in sorted(glob.glob(os.path.join(r, "freq*_label"))):
            try:
                if open(lab).read().strip() == "sclk":
                    files.append(lab.replace("_label", "_input"))
            except OSError:
                pass
    return files


def measure_shader_clock(eng, device_index, iters=3000):
    """Shader clock (GHz) under THIS workload: the hwmon sclk inputs sampled from a second thread while `iters` squarings run
    (the ctypes call releases the GIL).  -> (GHz or None, description of the source)."""
    import threading
    files = _sclk_files(device_index)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            vals = []
            for f in files:
                try:
                    vals.append(float(open(f).read().strip()))   # Hz
                except (OSError, ValueError):
                    pass
            if vals:
                samples.append(sum(vals) / len(vals))
            time.sleep(0.005)
    if files:
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        eng.time_square_mul(0, iters)
        stop.set(); th.join()
        busy = [v for v in samples[len(samples) // 4:] if v > 0]   # (the first quarter may still see the idle clock)
        if busy:
            return sum(busy) / len(busy) / 1e9, "hwmon sclk, %d samples over %d squarings, %d sensor(s)" % (len(busy), iters, len(files))
    try:   # amdsmi through torch, sampled right after a batch (coarser: one reading)
        import torch
        eng.time_square_mul(0, iters // 4)
        mhz = float(torch.cuda.clock_rate(device_index))
        if mhz > 0:
            return mhz / 1e3, "torch.cuda.clock_rate after a batch of squarings"
    except Exception:
        pass
    return None, "no readable shader clock on this box (hwmon sclk and torch.cuda.clock_rate both unavailable)"


def cpu_baseline(p, sample_iters=100):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc   # the oracle is the checker / CPU baseline only (never on the product path)
    o = orc.Oracle(p, 1)
    o.set_digits(0, seeded_digits(p, o.n, 1))
    o.square_mul(0)   # warm-up
    t0 = time.perf_counter()
    for _ in range(sample_iters):
        o.square_mul(0)
    dt = time.perf_counter() - t0
    return {"value": round(sample_iters / dt, 4), "unit": "iter/s", "ms_per_iter": round(1e3 * dt / sample_iters, 2),
            "cores": int(orc.lib().orc_threads()), "kind": "port",
            "sample": "%d squarings at p=%d (n=%d) by oracle/oracle.c, OpenMP" % (sample_iters, p, o.n)}


def bench_crt(args):
    """BASELINE configs[3] taken literally: p = 205271257 at the radix-9 size 9 * 2^20 (or --odd 3: 3 * 2^21) on the second field
    family.  Same timing discipline as the headline line (preheat, warm-up, K steps between synchronisations), one GPU."""
    import numpy as np
    import torch
    from prmers_amd import CrtEngine
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--field crt is a single-GPU line")
    p = args.exponent or 205271257
    eng = CrtEngine(p, args.odd, args.words, reg_count=1)
    n = eng.n
    j = np.arange(n + 1, dtype=np.uint64)
    ceil = (j * np.uint64(p) + np.uint64(n - 1)) // np.uint64(n)
    width = (ceil[1:] - ceil[:-1]).astype(np.uint64)
    eng.set_digits(0, np.random.default_rng(1000).integers(0, 1 << 62, n, dtype=np.uint64) & ((np.uint64(1) << width) - np.uint64(1)))
    t_start = time.perf_counter()
    while args.preheat_seconds > 0 and time.perf_counter() - t_start < args.preheat_seconds:
        for _ in range(50):
            eng.square_mul(0, 1)
        eng.sync()
    for _ in range(max(1, args.warmup)):
        eng.square_mul(0, 1)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.square_mul(0, 1)
    eng.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _, kern = eng.time_square_mul(0, min(args.steps, 32))
    ms = 1e3 * elapsed / args.steps
    dom = max(kern, key=kern.get)
    bytes_iter = eng.algorithmic_bytes()
    traffic = None
    try:   # HBM bytes per squaring from the committed PMC passes of this command (tools/profile.sh, tools/make_traffic_json.py)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_crt_latest.json")))
        if tj.get("plan") == eng.describe():
            traffic = tj.get("per_squaring")
    except Exception:
        traffic = None
    out = {"metric": "squaring throughput at p=%d over GF(M61^2) x GF(M31^2), PFA radix-%d axis" % (p, args.odd), "value": round(args.steps / elapsed, 3),
           "unit": "iter/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u64 + u32 (GF(M61^2) x GF(M31^2))", "data": "synthetic",
           "config": {"workload": "square_mul x<-x^2 mod 2^p-1, p=%d, n=%d words (odd radix %d)" % (p, n, args.odd), "plan": eng.describe()},
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(bytes_iter / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(bytes_iter / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": "profiles/traffic_crt_latest.json (rocprofv3 --pmc passes of this command)" if traffic else None,
                        "algorithmic_bytes_per_iteration": bytes_iter, "stage_ms": {k: round(v, 5) for k, v in kern.items()},
                        "note": "whole iteration; stage slots: front, forward columns, rows + pointwise + inverse rows, inverse columns, back + carry (one kernel), range edges; DESIGN.md section 7 N1"}}
    print(json.dumps(out))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--exponent", type=int, default=0, help="override the exponent (testing)")
    ap.add_argument("--plan", type=str, default=None, help="plan override, e.g. m2=4096,c=4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--field", default="goldilocks", choices=["goldilocks", "crt"],
                    help="crt: the GF(M61^2) x GF(M31^2) engine with the prime-factor radix-3/9 axis (SURVEY.md 8f N1; one GPU, not the headline line)")
    ap.add_argument("--odd", type=int, default=9, help="--field crt: radix of the odd axis (1, 3, 9)")
    ap.add_argument("--words", type=int, default=0, help="--field crt: transform words (0: smallest admissible)")
    ap.add_argument("--preheat-seconds", type=float, default=2.0,
                    help="untimed clock ramp before the warm-up: at least this long and until two batches agree within 1 %% (0: off)")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="let the status reduction fall back to gloo when RCCL cannot start (otherwise that is an error)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo to rehearse")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1,
                    help="rehearsal only: put every rank on this device (implies a gloo status reduction)")
    args = ap.parse_args()

    import torch
    from prmers_amd import Engine

    if args.field == "crt":
        return bench_crt(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    rehearsal = args.all_ranks_on_device >= 0
    if rehearsal:
        local_rank, args.dist_backend = args.all_ranks_on_device, "gloo"
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            try:   # the collective is a 24-byte status word; if RCCL cannot come up, do not lose the measurement
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as exc:   # pragma: no cover - needs a broken fabric
                if not args.allow_gloo:
                    sys.stderr.write("[bench] RCCL unavailable (%s); pass --allow-gloo to reduce the status word over gloo instead\n" % exc)
                    raise SystemExit(3)
                sys.stderr.write("[bench] RCCL unavailable (%s); status reduction falls back to gloo (--allow-gloo)\n" % exc)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo")
                red_dev, args.dist_backend = "cpu", "gloo (RCCL init failed)"
        else:
            dist.init_process_group(args.dist_backend)

    p = args.exponent or EXPONENTS[rank % len(EXPONENTS)]
    eng = Engine(p, 2, device=local_rank, plan=args.plan)
    eng.set_digits(0, seeded_digits(p, eng.n, 1000 + rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    # clock ramp (untimed, not part of --warmup): batches of squarings until the rate has settled
    preheat_iters, preheat_ms = 0, 0.0
    if args.preheat_seconds > 0:
        last, t_start = None, time.perf_counter()
        while True:
            ms, _ = eng.time_square_mul(0, 500)
            preheat_iters += 500; preheat_ms += ms
            spent = time.perf_counter() - t_start
            settled = last is not None and abs(ms - last) <= 0.01 * last
            if (spent >= args.preheat_seconds and settled) or spent >= 4 * args.preheat_seconds:
                break
            last = ms
    eng.time_square_mul(0, max(1, args.warmup))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_ms, _ = eng.time_square_mul(0, args.steps)      # K squarings enqueued back to back, HIP events around them
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0

    status = torch.tensor([1, 0, args.steps], dtype=torch.int64, device=red_dev)   # ok, gerbicz errors, iterations
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    per_rank = [{"rank": rank, "device": local_rank, "exponent": p, "ms_per_step": round(1e3 * elapsed / args.steps, 5)}]
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(status, op=dist.ReduceOp.SUM)   # the only cross-GPU traffic: a 24-byte status word
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])    # what every rank ran and how long it took (reported, not timed)
        per_rank = gathered
    elapsed = float(tmax.item())

    # BASELINE configs[2] is the LL form of the same exponent (x^2 - 2): a short untimed-region measurement
    # of that variant, reported beside the metric (the subtraction rides on the next front sweep)
    ll_iters = min(args.steps, 500)
    ll_ms, _ = eng.time_square_mul(0, ll_iters, sub=2)
    # second, instrumented pass of the same loop: per-kernel durations from event pairs on the engine stream
    _, kern = eng.time_square_mul(0, min(args.steps, 64), per_kernel=True)
    # a Gerbicz-style check at this size: device-side canonical form + compare, and res64
    check = None
    if rank == 0:
        eng.copy(1, 0)
        eng.is_equal(0, 1); eng.res64(0)   # allocate the scratch, warm the kernels
        tc0 = time.perf_counter(); same = eng.is_equal(0, 1); tc1 = time.perf_counter(); eng.res64(0); tc2 = time.perf_counter()
        check = {"is_equal_ms": round(1e3 * (tc1 - tc0), 3), "res64_ms": round(1e3 * (tc2 - tc1), 3), "equal": bool(same),
                 "pcie_bytes_per_check": 16, "note": "strong carry, compare and res64 on the device (canon.hip)"}
    if rank == 0:
        n = eng.n
        event_overhead = kern.pop("event_overhead", 0.0)
        kern = {k: v for k, v in kern.items() if v >= 0}   # slots this path does not launch are reported as -1
        chain = {k: v for k, v in kern.items() if k != "k_sub_small"}
        dom = max(chain, key=chain.get)
        dom_ms = chain[dom]
        sweep_bytes = 16 * n
        achieved = sweep_bytes / (dom_ms * 1e-3) / 1e9
        ms_per_step = 1e3 * elapsed / args.steps
        traffic = None
        try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/profile.sh)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if tj.get("plan") == __import__("prmers_amd").resolve_plan(p, args.plan):
                traffic = tj.get(dom)
        except Exception:
            traffic = None
        valu = None
        try:   # the VALU issue model of the three kernels (ISA walk, tools/valu_model.py) against this run's kernel times and clock
            vm = json.load(open(os.path.join(ROOT, "profiles", "valu_model_latest.json")))
            if vm.get("plan") == __import__("prmers_amd").resolve_plan(p, args.plan):
                ghz, clock_source = measure_shader_clock(eng, local_rank)
                simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
                waves_per_simd = vm["waves_per_launch"] / simds
                ks = {}
                for k, v in vm["kernels"].items():
                    if k in chain:
                        pred = waves_per_simd * v["issue_cycles_per_wave"] / (ghz * 1e3) if ghz else None
                        ks[k] = {"insts": int(round(v["valu_insts_per_wave"] * vm["waves_per_launch"])), "issue_cycles_per_wave": v["issue_cycles_per_wave"],
                                 "predicted_us": round(pred, 2) if pred else None, "measured_us": round(chain[k] * 1e3, 2),
                                 "frac": round(pred / (chain[k] * 1e3), 4) if pred else None}
                pred = sum(v["predicted_us"] for v in ks.values()) if ghz else None
                valu = {"insts": sum(v["insts"] for v in ks.values()), "issue_cycles": round(sum(v["issue_cycles_per_wave"] for v in ks.values()), 1),
                        "predicted_us": round(pred, 2) if pred else None, "measured_us": round(ms_per_step * 1e3, 2),
                        "frac": round(pred / (ms_per_step * 1e3), 4) if pred else None,
                        "shader_clock_ghz": round(ghz, 4) if ghz else None, "shader_clock_source": clock_source, "simds": simds, "kernels": ks,
                        "note": "frac = predicted VALU issue time (instruction walk of the built kernels x measured issue cost per opcode, at the shader "
                                "clock read during this run) / measured time; the marginal-cost probe behind the steady-state figures is "
                                "profiles/r03_probe_timeline.md (a separate -DMI355_PROBE build, not part of this run)",
                        "source": "profiles/valu_model_latest.json (tools/valu_model.py) + this run's kernel times and clock"}
        except Exception as exc:
            valu = {"error": "valu model unavailable: %s" % exc}
        out = {
            "metric": "PRP squaring throughput at p~136M (Marin IBDWT, one exponent per GPU)",
            "value": round(world * args.steps / elapsed, 3),
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (GF(2^64-2^32+1))",
            "data": "synthetic",
            "config": {"workload": "square_mul x<-x^2 mod 2^p-1, p=%d, n=%d words" % (p, n),
                       "plan": __import__("prmers_amd").resolve_plan(p, args.plan),
                       "exponents": EXPONENTS[:world] if not args.exponent else [p],
                       "per_rank": per_rank,
                       "status_reduction_backend": args.dist_backend if world > 1 else None,
                       "parallelism": "replicas: one exponent per GPU, no data-path collective" +
                                      (" [REHEARSAL: all ranks on one device]" if rehearsal else "")},
            "event_ms_per_step": round(ev_ms / args.steps, 5),
            "ll_ms_per_step": round(ll_ms / ll_iters, 5),
            "preheat_iters": preheat_iters, "preheat_ms": round(preheat_ms, 1),
            "gerbicz_check": check,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4),
                         "binding_roof": "VALU issue (integer GF(P) arithmetic), see DESIGN.md section 5",
                         "valu": valu,
                         "algorithmic_bytes_per_launch": sweep_bytes,
                         "kernel_ms": {k: round(v, 5) for k, v in kern.items()},
                         "kernel_ms_sum": round(sum(chain.values()), 5), "event_record_ms_subtracted": round(event_overhead, 5),
                         "traffic_source": "profiles/traffic_latest.json (rocprofv3 --pmc passes of this command, tools/profile.sh)" if traffic else None,
                         "iteration": {"algorithmic_bytes": 48 * n,
                                       "achieved": round(48 * n / (ms_per_step * 1e-3) / 1e9, 1),
                                       "frac": round(48 * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
