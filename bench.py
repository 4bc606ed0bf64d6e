#!/usr/bin/env python3
"""Benchmark of the multigrid V-cycle on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one full V-cycle iteration (smooth, residual + norm, restrict, coarse
levels, prolong + correct) on the synthetic charge density of
multigrid_amd/problem.py, with phi and rho resident in HBM before the timed
region.  N = 1 runs the 513^3 five-level cycle (the configuration the headline
metric is quoted on); N > 1 (one rank per GPU under torch.distributed.run) runs
1025^3 slab-partitioned.  `value` is GDoF/s = fine-grid unknowns x V-cycles/s /
1e9, comparable across grid sizes; V-cycles/s is reported beside it.  Rank 0
prints ONE JSON line.

Besides the contract fields the line carries:
  roofline      the dominant kernel's algorithmic bytes / its HIP-event time, vs 8 TB/s
  cpu_baseline  the reference algorithm (lexicographic GS-SOR cycle; oracle/) timed
                single-threaded on this host on a bounded number of cycles
  solve         cycles and seconds to residual norm < 1e-6 from phi = 0
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# Algorithmic (compulsory) bytes per point of the level a kernel runs on, fp64.
KERNEL_BYTES_PER_POINT = {"smooth_fine": 16.0, "residual": 24.0, "restrict": 9.0,
                          "smooth_coarse": 16.0, "prolong": 9.0, "prolong_correct": 17.0}


# kernel names in the profile summaries: (prefix, suffix).  The fine sweep of the timed cycle is the
# single-pass kernel in its "correction applied while loading" form (last template argument 1).
# The last template argument of the sweeps is the short-form switch (false in the timed cycle), the one
# before it the on-load mode (1 = correction).  The residual instance ending in true> is the
# Gauss-Seidel solver's norm.
PROFILE_KERNEL = {"smooth_fine": ("k_sweep_rb<", ", 1, false>"), "residual": ("k_residual_restrict", "false>"),
                  "prolong_correct": ("k_prolong<true>", "")}


def profiled_traffic(kind):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 counter
    summary (profiles/*_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, FETCH_SIZE doubled for 16-byte-per-lane loads as the MI355X counter notes require).
    bench.py cannot run the profiler around itself; None when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")))
    match = PROFILE_KERNEL.get(kind)
    if not files or not match:
        return None, None
    try:
        prof = json.load(open(files[-1]))
        best = max((v.get("hbm_bytes_max_launch", 0.0) for k, v in prof["kernels"].items()
                    if k.startswith(match[0]) and k.endswith(match[1])), default=0.0)
        return (best or None), os.path.basename(files[-1])
    except Exception:
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=0, help="fine grid nodes per axis (default 513)")
    ap.add_argument("--depth", type=int, default=0, help="coarse grids (default: 5 at 513)")
    ap.add_argument("--cpu-cycles", type=int, default=3, help="cycles of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-solve", action="store_true", help="skip the time-to-1e-6 solve")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--no-fast", action="store_true", help="skip the extra leg that times MG_FLAG_FAST_SMOOTHER")
    ap.add_argument("--flags", type=int, default=0, help="MG_FLAG_* bits for the timed handle (profiling aid; "
                                                          "named in config.workload when set)")
    ap.add_argument("--no-single-gpu-compare", action="store_true",
                    help="N>1: skip timing the same workload on one GPU of the node")
    return ap.parse_args()


def cpu_baseline(n, depth, cycles):
    """The reference algorithm on one host core, same inputs, `cycles` iterations."""
    import numpy as np
    from oracle import mg_oracle as orc
    orc.build()
    note = ""
    if n > 513:
        # the oracle keeps phi, rho, R_0 and eps_0 whole in host memory (4 x n^3 x 8 B plus the
        # temporaries of the rho generator): fall back to the 513^3 sample when that does not fit
        need = 7.0 * 8.0 * float(n) ** 3
        try:
            import psutil
            avail = float(psutil.virtual_memory().available)
        except Exception:
            avail = 0.0
        if avail < 1.5 * need:
            note = f" (the {n}^3 arrays do not fit this host's free memory: 513^3 V5 sample instead)"
            n, depth = 513, 5
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    rho = orc.synthetic_rho(N)
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    t0 = time.perf_counter()
    s.solve(phi, rho, cycles, 0.0)
    dt = time.perf_counter() - t0
    per = dt / cycles
    return {"value": 1.0 / per, "unit": "V-cycles/s", "cores": 1, "kind": "port",
            "sample": f"{cycles} iterations of the lexicographic (reference-order) {n}^3 V{depth} cycle, "
                      f"gcc -O3 single thread, host has {os.cpu_count()} logical CPUs" + note,
            "s_per_cycle": per, "gdof_per_s": n ** 3 / per / 1e9}


def run_single(args):
    import numpy as np
    from multigrid_amd import libmg, problem

    libmg.load()
    if libmg.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    n = args.size or 513
    N = (n, n, n)
    depth = args.depth or problem.reference_depth_for(N)
    dh = problem.main_extents_dh(N)
    rho = problem.synthetic_rho(N)
    npts = float(n) ** 3

    h = libmg.Handle(N, dh, depth, device=0, flags=args.flags)
    h.set_rho(rho)
    h.set_phi(np.zeros(N))
    if args.warmup:
        h.vcycles(args.warmup)
    t0 = time.perf_counter()
    l2 = h.vcycles(args.steps)          # K iterations, no host sync inside; returns after a stream sync
    wall = time.perf_counter() - t0
    dev_ms = h.last_call_ms()
    ms_per_step = wall * 1e3 / args.steps
    vps = args.steps / wall

    # per-kernel HIP-event timing of the same cycle (separate pass: the events perturb the stream)
    h.set_profiling(True)
    h.reset_timers()
    h.vcycles(max(3, min(args.steps, 10)))
    timers = h.timers()
    h.set_profiling(False)
    kernels = {}
    for k, t in timers.items():
        if t["launches"]:
            # bytes the library books per category are those of the UNFUSED operators a launch stands
            # for (e.g. 32 B/pt per sweep + 17 for a folded correction): an operator-equivalent rate,
            # not a bandwidth -- the roofline object prices the dominant launch at its own bytes
            kernels[k] = {"launches": t["launches"], "avg_us": 1e3 * t["ms"] / t["launches"],
                          "operator_equivalent_GBps": t["bytes"] / (t["ms"] * 1e-3) / 1e9 if t["ms"] > 0 else None,
                          "share": None}
    tot = sum(t["ms"] for t in timers.values()) or 1.0
    for k in kernels:
        kernels[k]["share"] = timers[k]["ms"] / tot
    # dominant kernel: the single launch with the largest share of the cycle (the coarse-level
    # categories add up several kernels on several levels)
    single = [k for k in ("smooth_fine", "residual", "prolong_correct") if k in kernels]
    dom = max(single or kernels, key=lambda k: kernels[k]["share"])
    traffic, traffic_src = profiled_traffic(dom) if n == 513 else (None, None)
    # Algorithmic bytes of the dominant launch = its compulsory traffic with perfect on-chip reuse.
    # The fine sweep launch also applies the pending correction phi -= P(eps_1) while loading:
    # read phi 8 + rhs 8 + eps_1 1 + write phi 8 = 25 B/pt for what the unfused operators would
    # move as 32 (two colour passes) + 17 (prolongation + correction) = 49 B/pt; the second figure
    # is reported as `operator_equivalent`, never as `achieved`.
    fused_bytes = {"smooth_fine": 25.0, "residual": 17.0, "prolong_correct": 17.0}
    # the fine level's two launches at their own compulsory bytes (one launch per cycle each)
    for k, b in fused_bytes.items():
        if k in kernels and timers[k]["launches"] == max(3, min(args.steps, 10)):
            gbps = b * npts / (kernels[k]["avg_us"] * 1e-6) / 1e9
            kernels[k]["compulsory"] = {"bytes_per_point": b, "GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS}
    launch_us = kernels[dom]["avg_us"]
    if dom in fused_bytes and timers[dom]["launches"] == max(3, min(args.steps, 10)):
        alg = fused_bytes[dom] * npts
    else:
        alg = timers[dom]["bytes"] / timers[dom]["launches"]
    achieved = alg / (launch_us * 1e-6) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "avg_us": launch_us, "bytes_per_launch": alg,
                "operator_equivalent": {"bytes_per_launch": timers[dom]["bytes"] / timers[dom]["launches"],
                                        "GBps": kernels[dom]["operator_equivalent_GBps"],
                                        "note": "bytes the separate operators folded into this launch would move"}}

    solve = None
    if not args.no_solve:
        h.set_phi(np.zeros(N))
        h.reset_state()
        t0 = time.perf_counter()
        ok, cycles, l2s = h.solve(10000, args.tol)
        dt = time.perf_counter() - t0
        solve = {"tol": args.tol, "converged": ok, "cycles": cycles, "seconds": dt, "L2": l2s}

    # the rows either side of the V-cycle, on the same device-resident fields (not part of `value`):
    # E = -grad(phi) (read 8 + write 24 B/pt), one plain red-black SOR sweep of the Gauss-Seidel
    # solver (32 B/pt), and the host<->device transfers of one solve() through pageable memory
    extras = {}
    h.compute_ef(download=False)
    ef_ms = min(h.compute_ef(download=False) or h.last_call_ms() for _ in range(3))
    extras["compute_ef"] = {"ms": ef_ms, "GBps": 32.0 * npts / (ef_ms * 1e-3) / 1e9}
    h.gs_solve(2, 0.0)
    h.gs_solve(26, 0.0)
    extras["gs_26_sweeps_2_norms_ms"] = h.last_call_ms()
    # the sweep kernel without the folded correction (what every sweep after a level's first one is):
    # read u 8 + rhs 8 + write u 8 = 24 B/pt, HIP events around each launch
    h.set_profiling(True)
    h.reset_timers()
    for _ in range(6):
        h.gs_solve(1, 0.0)  # one sweep (and one norm) per call
    t = h.timers()["smooth_fine"]
    if t["launches"]:
        us = 1e3 * t["ms"] / t["launches"]
        extras["plain_sweep"] = {"avg_us": us, "GBps": 24.0 * npts / (us * 1e-6) / 1e9,
                                 "frac_of_hbm_peak": 24.0 * npts / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        # two sweeps per launch (what the Gauss-Seidel solver and levels with several sweeps run):
        # 25 sweeps = one single launch + 12 paired ones; 2 x 12 B/pt per sweep
        h.reset_timers()
        h.gs_solve(25, 0.0)
        t2 = h.timers()["smooth_fine"]
        if t2["launches"] == 13:
            us2 = (1e3 * t2["ms"] - us) / 12.0
            extras["paired_sweeps"] = {"avg_us_per_launch": us2, "GBps": 24.0 * npts / (us2 * 1e-6) / 1e9,
                                       "frac_of_hbm_peak": 24.0 * npts / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                       "what": "two red-black sweeps in one pass: read u 8 + rhs 8 + write u 8 B/pt per launch"}
    h.set_profiling(False)
    t0 = time.perf_counter()
    h.set_rho(rho)
    h.set_phi(np.zeros(N))
    buf = h.get_phi()
    dt = time.perf_counter() - t0
    extras["host_transfers"] = {"seconds": dt, "GBps": 3 * 8.0 * npts / dt / 1e9,
                                "what": "mg_set_rho + mg_set_phi + mg_get_phi, pageable host arrays"}
    del buf
    # What one solve() of the host adapter moves when the World's fields live in page-locked blocks
    # (mg_host_alloc; host/box_mg does so): rho up, phi up, phi down -- the adapter reads world.phi on
    # every call, as the reference does; with setAssumePhiUntouched(true) an unchanged phi stays on
    # the device and the solve moves rho up and phi down only.
    import ctypes as C
    lib = libmg.load()
    blocks = []
    try:
        for _ in range(2):
            ptr = C.c_void_p()
            libmg._check(lib.mg_host_alloc(C.byref(ptr), int(npts) * 8))
            blocks.append(ptr)
        pin = [np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_double)), shape=N) for b in blocks]
        pin[0][...] = rho
        h.set_rho(pin[0])
        h.get_phi(pin[1])
        t0 = time.perf_counter()
        h.set_rho(pin[0])
        h.set_phi(pin[1])
        h.get_phi(pin[1])
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        h.set_rho(pin[0])
        h.get_phi(pin[1])
        dt2 = time.perf_counter() - t0
        extras["host_transfers_pinned"] = {"seconds": dt, "GBps": 3 * 8.0 * npts / dt / 1e9,
                                           "what": "mg_set_rho + mg_set_phi + mg_get_phi, page-locked host arrays: "
                                                   "per-solve traffic of the host adapter (phi read on every call)",
                                           "phi_resident": {"seconds": dt2, "GBps": 2 * 8.0 * npts / dt2 / 1e9,
                                                            "what": "mg_set_rho + mg_get_phi: the adapter with "
                                                                    "setAssumePhiUntouched(true) and phi unchanged"}}
        if solve:
            extras["solve_wall_including_transfers_s"] = solve["seconds"] + dt
        del pin
    finally:
        for b in blocks:
            lib.mg_host_free(b)
    h.close()

    # The same cycle with the coarse levels' point update in its short form (MG_FLAG_FAST_SMOOTHER:
    # one chain of fused multiply-adds, equal to the reference's expression up to rounding; the fine
    # sweep, residual, restriction, prolongation and the norm stay exact).  Reported beside the
    # headline, never as it.
    hf = None if args.no_fast else libmg.Handle(N, dh, depth, device=0, flags=libmg.MG_FLAG_FAST_SMOOTHER)
    if hf is not None:
        hf.set_rho(rho)
        hf.set_phi(np.zeros(N))
        hf.vcycles(max(1, args.warmup))
        t0 = time.perf_counter()
        hf.vcycles(args.steps)
        wf = time.perf_counter() - t0
        hf.set_profiling(True)
        hf.reset_timers()
        hf.vcycles(5)
        tf = hf.timers()
        hf.set_profiling(False)
        fast = {"ms_per_step": wf * 1e3 / args.steps, "gdof_per_s": npts * args.steps / wf / 1e9,
                "kernels_avg_us": {k: 1e3 * t["ms"] / t["launches"] for k, t in tf.items() if t["launches"]},
                "what": "MG_FLAG_FAST_SMOOTHER: 8 instead of 22 fp64 operations per point update on the levels >= 1; "
                        "not bit-identical to the reference's expression (rounding), checked against the oracle's twin "
                        "of the same form"}
        if not args.no_solve:
            hf.set_phi(np.zeros(N))
            hf.reset_state()
            t0 = time.perf_counter()
            ok, cycles, l2s = hf.solve(10000, args.tol)
            fast["solve"] = {"converged": ok, "cycles": cycles, "seconds": time.perf_counter() - t0, "L2": l2s}
        hf.close()
        extras["fast_smoother"] = fast

    cpu = cpu_baseline(n, depth, args.cpu_cycles) if args.cpu_cycles > 0 else None

    out = {"metric": "GDoF/s", "value": npts * vps / 1e9, "unit": "GDoF/s (fine-grid unknowns x V-cycles/s / 1e9)",
           "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{n}^3 fp64 multigrid V{depth} cycle (sweeps {list(h.cfg.nu[:depth + 1])}, omega 1.4), "
                                  f"red-black GS-SOR, 1 GPU" + (f", MG_FLAG bits {args.flags}" if args.flags else ""),
                      "grid": n, "depth": depth},
           "vcycles_per_s": vps, "device_ms_per_step": dev_ms / args.steps, "last_L2": l2,
           "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu, "solve": solve,
           "next_rows": extras}
    if cpu:
        out["speedup_vs_cpu_per_cycle"] = cpu["s_per_cycle"] / (wall / args.steps)
    print(json.dumps(out))


def slab_profiled_traffic(n, world):
    """HBM bytes of the dominant launch on one rank's slab (the plain out-of-place sweep of the rank's
    fine planes) from the committed rocprofv3 counter summary of that share, measured with the rank
    alone on one GPU (tools/profile_slab.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    FETCH_SIZE doubled).  None where no summary of this configuration is committed."""
    name = {(1025, 8): "r03h_slab_mid", (1025, 4): "r03b_slab_w4", (1025, 2): "r03b_slab_w2",
            (2049, 8): "r03b_slab2049_mid"}.get((n, world))
    path = os.path.join(ROOT, "profiles", f"{name}_summary.json") if name else None
    if not path or not os.path.exists(path):
        return None, None
    try:
        prof = json.load(open(path))
        best = max((v.get("hbm_bytes_max_launch", 0.0) for k, v in prof["kernels"].items()
                    if k.startswith("k_sweep_rb<64, 32, 2, 512, 0")), default=0.0)
        return (best or None), os.path.basename(path)
    except Exception:
        return None, None


def sweep_roofline(libmg, n, planes, device):
    """HIP-event time of the dominant kernel on a slab of this rank's shape: one full red-black sweep
    (out of place) of `planes` owned planes of n x n nodes, launched on torch's current stream and
    timed with events on that stream.  Compulsory bytes: read u 8 + rhs 8 + write u 8 = 24 B/pt."""
    import ctypes as C
    import torch
    from multigrid_amd import problem
    pitch = (n + 1) & ~1
    dev = torch.device("cuda", device)
    # (random operands: all-zero arrays would send every quotient down the smoother's rare-operand path)
    u = torch.rand((planes + 4, n, pitch), dtype=torch.float64, device=dev)
    alt = torch.zeros_like(u)
    rhs = torch.rand((planes + 2, n, pitch), dtype=torch.float64, device=dev)
    lib = libmg.load()
    N = (n, n, n)
    coef = (C.c_double * 4)()
    libmg._check(lib.mg_level_coef((C.c_double * 3)(*problem.main_extents_dh(N)), 0, coef))
    # the slab sits in the middle of the grid: global planes [n/2 - 2, n/2 + planes + 2)
    i0 = max(2, n // 2 - planes // 2)
    arr = lambda t, first: libmg.MgDevArray(t.data_ptr(), t.shape[0], n, n, first)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    flag = C.c_int(0)

    def launch():
        libmg._check(lib.mg_dev_smooth_rb(stream, C.byref(arr(u, i0 - 2)), C.byref(arr(alt, i0 - 2)),
                                          C.byref(arr(rhs, i0 - 1)), (C.c_int64 * 3)(*N), coef, 1.4, 1,
                                          i0, i0 + planes, 0, 0, C.byref(flag)))
    launch()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 5
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        launch()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    nbytes = 24.0 * planes * n * n
    return {"bound": "hbm", "kernel": "smooth_fine (one full red-black sweep of this rank's slab)",
            "achieved": nbytes / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_us": us,
            "bytes_per_launch": nbytes,
            "operator_equivalent": {"bytes_per_launch": nbytes * 32.0 / 24.0,
                                    "GBps": nbytes * 32.0 / 24.0 / (us * 1e-6) / 1e9,
                                    "note": "bytes of the two colour passes this launch replaces"}}


def run_distributed(args):
    """N > 1: one rank per GPU (launched by torch.distributed.run), 1025^3 slab-partitioned, driven by
    the C++ rank driver inside libmg_hip (mg_dist_*): halo planes, the norm all-reduce and the
    all-gather of the first whole level go over RCCL.  torch.distributed (gloo) only carries the rendezvous: the RCCL
    unique id, the barriers around the timed region and the max over the ranks' times."""
    import torch  # before libmg_hip: both must share one HIP runtime
    import torch.distributed as dist
    import numpy as np
    from multigrid_amd import libmg, problem

    # MG_BENCH_FORCE_DIST=1 without a launcher: a world of one rank through the same code
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    world = int(os.environ["WORLD_SIZE"])
    # RCCL prints a version banner on stdout at communicator creation; keep stdout for the one
    # JSON line by pointing fd 1 at stderr until the result is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dist.init_process_group("gloo")
    libmg.load()
    n = args.size or 1025
    N = (n, n, n)
    depth = args.depth or problem.reference_depth_for(N)
    dh = problem.main_extents_dh(N)

    def make_rank():
        ids = [libmg.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        d = libmg.DistRank(N, dh, depth, rank, world, device=local, unique_id=ids[0])
        d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        return d

    d = make_rank()

    def fence():
        d.sync()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        d.vcycles(args.warmup)
    fence()
    t0 = time.perf_counter()
    l2 = d.vcycles(args.steps)   # K iterations enqueued back to back; returns after this rank's stream drained
    fence()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall = float(dt.item())
    vps = args.steps / wall
    npts = float(n) ** 3
    stats = d.stats()
    cut, own, deep = d.cut, d.own, d.deep_halos

    solve = None
    if not args.no_solve:
        d.close()
        d = make_rank()
        fence()
        t0 = time.perf_counter()
        ok, cycles, l2s = d.solve(10000, args.tol)
        fence()
        solve = {"tol": args.tol, "converged": ok, "cycles": cycles, "seconds": time.perf_counter() - t0, "L2": l2s}
    d.close()
    dist.barrier()

    roofline = sweep_roofline(libmg, n, own[1] - own[0], local)
    roofline["traffic"], roofline["traffic_source"] = slab_profiled_traffic(n, world)
    base1 = None
    if rank == 0 and not args.no_single_gpu_compare and n <= 1025:
        torch.cuda.empty_cache()
        h = libmg.Handle(N, dh, depth, device=local)
        h.set_rho(problem.synthetic_rho(N))
        h.set_phi(np.zeros(N))
        h.vcycles(2)
        t0 = time.perf_counter()
        k = max(3, min(args.steps, 10))
        h.vcycles(k)
        t1 = (time.perf_counter() - t0) / k
        h.close()
        base1 = {"workload": f"{n}^3 V{depth} on one GPU of this node, same run", "ms_per_step": t1 * 1e3,
                 "vcycles_per_s": 1.0 / t1, "gdof_per_s": npts / t1 / 1e9, "speedup": vps * t1}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # the reference-order CPU cycle on one host core of this node, after the group is gone (the
        # other ranks have nothing left to wait for): one iteration at this workload's size
        cpu = cpu_baseline(n, depth, max(1, min(args.cpu_cycles, 1 if n > 513 else 3))) if args.cpu_cycles > 0 else None
        out = {"metric": "GDoF/s", "value": npts * vps / 1e9, "unit": "GDoF/s (fine-grid unknowns x V-cycles/s / 1e9)",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{n}^3 fp64 multigrid V{depth} cycle, red-black GS-SOR, slab-partitioned over "
                                      f"{world} GPUs (halo planes over RCCL send/recv, levels >= {cut} "
                                      f"all-gathered and solved whole on every GPU"
                                      + (f", levels 1..{cut - 1} on deep halos: one exchange of R_l per cycle" if deep else "")
                                      + ")", "grid": n, "depth": depth, "driver": "libmg_hip mg_dist (C++)",
                          "deep_halos": deep,
                          "note": "N=1 runs the 513^3 headline configuration; N>=2 run 1025^3 (strong scaling among N>=2)"},
               "vcycles_per_s": vps, "last_L2": l2, "roofline": roofline, "cpu_baseline": cpu, "solve": solve,
               "single_gpu_same_workload": base1, "halo_bytes_sent_rank0": stats["bytes_sent"],
               "device_ms_per_step_rank0": stats["last_ms"] / args.steps}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)


def run_shared_gpu_rehearsal(args):
    """MG_BENCH_SHARE_GPU=1 python bench.py --gpus N on a box with ONE GPU: the N ranks are host
    threads inside the library and every slab lives on GPU 0 (mg_create with n_gpus = N; halo planes
    by device-to-device copies).  It rehearses the partitioned code path; its numbers say nothing
    about a multi-GPU node."""
    import numpy as np
    from multigrid_amd import libmg, problem
    libmg.load()
    n = args.size or 513
    N = (n, n, n)
    depth = args.depth or problem.reference_depth_for(N)
    h = libmg.Handle(N, problem.main_extents_dh(N), depth, devices=[0] * args.gpus)
    h.set_rho(problem.synthetic_rho(N))
    h.set_phi(np.zeros(N))
    if args.warmup:
        h.vcycles(args.warmup)
    t0 = time.perf_counter()
    l2 = h.vcycles(args.steps)
    wall = time.perf_counter() - t0
    print(json.dumps({"metric": "GDoF/s", "value": float(n) ** 3 * args.steps / wall / 1e9, "unit": "GDoF/s",
                      "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
                      "vs_baseline": None, "dtype": "f64", "data": "synthetic", "last_L2": l2,
                      "config": {"workload": f"REHEARSAL: {n}^3 V{depth} as {args.gpus} slabs sharing ONE GPU "
                                             f"(in-process transport); not a multi-GPU measurement"}}))
    h.close()


def run_distributed_python(args):
    """The same N > 1 run with the Python orchestration (multigrid_amd/dist.py over
    torch.distributed point-to-point ops) instead of the C++ rank driver: MG_BENCH_DRIVER=python."""
    import torch  # before libmg_hip: both must share one HIP runtime
    import torch.distributed as dist
    import numpy as np
    from multigrid_amd import dist as mgd, problem

    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    world = int(os.environ["WORLD_SIZE"])
    # RCCL prints a version banner on stdout at communicator creation; keep stdout for the one
    # JSON line by pointing fd 1 at stderr until the result is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    # MG_BENCH_BACKEND=gloo with MG_BENCH_SHARE_GPU=1 rehearses this code path with several ranks on
    # ONE GPU (halo planes staged through the host); the numbers of such a run mean nothing
    backend = os.environ.get("MG_BENCH_BACKEND", "nccl")
    if os.environ.get("MG_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    n = args.size or 1025
    N = (n, n, n)
    depth = args.depth or problem.reference_depth_for(N)
    dh = problem.main_extents_dh(N)
    ops = mgd.HipOps(local)
    comm = mgd.Comm()
    s = mgd.DistSolver(N, dh, depth, ops, comm)
    s.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
    s.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        s.vcycles(args.warmup)
    fence()
    t0 = time.perf_counter()
    l2 = s.vcycles(args.steps)
    fence()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    wall = float(dt.item())
    vps = args.steps / wall
    npts = float(n) ** 3

    # dominant kernel: the fused red-black sweep on this rank's fine slab, HIP events on the launch stream
    lo, hi = s.own(0, "phi")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 5
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        ops.smooth(s.phi, s.phi_alt, s.b, N, s.coef[0], s.omega, 1, lo, hi)  # result discarded: timing only
    ev[1].record()
    torch.cuda.synchronize()
    sweep_us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    # one out-of-place red-black sweep of the slab: read u 8 + rhs 8 + write u 8 = 24 B/pt compulsory
    # (the two colour passes it replaces would move 32)
    sweep_bytes = 24.0 * (hi - lo) * n * n
    roofline = {"bound": "hbm", "kernel": "smooth_fine (one full red-black sweep of this rank's slab)",
                "achieved": sweep_bytes / (sweep_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": sweep_bytes / (sweep_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "avg_us": sweep_us, "bytes_per_launch": sweep_bytes,
                "operator_equivalent": {"bytes_per_launch": sweep_bytes * 32.0 / 24.0,
                                        "GBps": sweep_bytes * 32.0 / 24.0 / (sweep_us * 1e-6) / 1e9,
                                        "note": "bytes of the two colour passes this launch replaces"}}

    solve = None
    if not args.no_solve:
        s2 = mgd.DistSolver(N, dh, depth, ops, comm)
        s2.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        s2.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        fence()
        t0 = time.perf_counter()
        ok, cycles, l2s = s2.solve(10000, args.tol)
        fence()
        solve = {"tol": args.tol, "converged": ok, "cycles": cycles, "seconds": time.perf_counter() - t0, "L2": l2s}
        del s2

    base1 = None
    if rank == 0 and not args.no_single_gpu_compare and n <= 1025:
        from multigrid_amd import libmg
        del s
        torch.cuda.empty_cache()
        h = libmg.Handle(N, dh, depth, device=local)
        h.set_rho(problem.synthetic_rho(N))
        h.set_phi(np.zeros(N))
        h.vcycles(2)
        t0 = time.perf_counter()
        k = max(3, min(args.steps, 10))
        h.vcycles(k)
        t1 = (time.perf_counter() - t0) / k
        h.close()
        base1 = {"workload": f"{n}^3 V{depth} on one GPU of this node, same run", "ms_per_step": t1 * 1e3,
                 "vcycles_per_s": 1.0 / t1, "gdof_per_s": npts / t1 / 1e9, "speedup": vps * t1}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # the reference-order CPU cycle on one host core of this node, after the group is gone (the
        # other ranks have nothing left to wait for): one iteration at this workload's size
        cpu = cpu_baseline(n, depth, max(1, min(args.cpu_cycles, 1 if n > 513 else 3))) if args.cpu_cycles > 0 else None
        out = {"metric": "GDoF/s", "value": npts * vps / 1e9, "unit": "GDoF/s (fine-grid unknowns x V-cycles/s / 1e9)",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{n}^3 fp64 multigrid V{depth} cycle, red-black GS-SOR, slab-partitioned over "
                                      f"{world} GPUs (halo planes over RCCL send/recv, levels >= {s_cut(N, depth, world)} "
                                      f"all-gathered and solved whole on every GPU)", "grid": n, "depth": depth,
                          "note": "N=1 runs the 513^3 headline configuration; N>=2 run 1025^3 (strong scaling among N>=2)"},
               "vcycles_per_s": vps, "last_L2": l2, "roofline": roofline, "cpu_baseline": cpu, "solve": solve,
               "single_gpu_same_workload": base1, "halo_bytes_sent_rank0": comm.bytes_sent}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)


def s_cut(N, depth, world):
    from multigrid_amd.dist import SlabPlan
    return SlabPlan(N, depth, world).cut


def visible_gpus():
    """GPUs this process could use, counted without initialising the HIP runtime (this process goes
    on to start the ranks as child programs and must not have touched a GPU itself)."""
    import torch
    return int(torch.cuda.device_count())


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` typed as is: this process has made no GPU call yet and makes none;
    it starts one rank per GPU under torch.distributed.run with the same arguments, relays rank 0's
    JSON line and returns the launcher's exit status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in p.stdout:
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 1
    if line is not None:
        print(line, flush=True)
    return rc


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus == 1 and world == 1 and not os.environ.get("MG_BENCH_FORCE_DIST"):
        run_single(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if os.environ.get("MG_BENCH_SHARE_GPU") and os.environ.get("MG_BENCH_DRIVER") != "python":
            run_shared_gpu_rehearsal(args)
            return
        have = visible_gpus()
        if args.gpus > have:
            raise SystemExit(f"bench.py: --gpus {args.gpus} asks for more GPUs than this machine shows ({have}); "
                             f"nothing was started (MG_BENCH_SHARE_GPU=1 rehearses the partitioned path on one GPU)")
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    if os.environ.get("MG_BENCH_DRIVER") == "python":
        run_distributed_python(args)
    else:
        run_distributed(args)


if __name__ == "__main__":
    main()
