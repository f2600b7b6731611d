#!/usr/bin/env python3
"""bench.py - Mpoints/s per ELBO+gradient step (BASELINE.json metric) on N GPUs of one node.

A "step" = one pass of the hot path over the synthetic batch: Phi pass over this rank's N-shard -> cross-workgroup reduce ->
one all-reduce(sum) of the packed band buffer (RCCL, only when N>1) -> banded ELBO + analytic gradient (replicated on every
rank).  Inputs are resident in HBM before the timed region.  Strong scaling: the BASELINE workload is N = 10M points in
total, sharded contiguously over the ranks.

`value` is the DEPENDENT-step figure (schedule C): theta of step i+1 is computed on the host from the result of step i that
the host has read (an optimiser step, /root/reference's experiments/snelson/example.py:31-32 semantics), so neither the M-side
launch nor the host forward pass of step i+1 can start before step i is finished; the only overlap is the theta-free Phi pass
of step i+1 under the M-side of step i.  The --steps block is timed R times (--repeats) inside the process: `value` /
`ms_per_step` are the median block, min / max ride along.  Extras: `one_step_at_a_time` (one stream, fixed theta, host runs
ahead: the device-side latency of a full step) and `independent_evaluations` (several steps with the SAME theta in flight: a
throughput of independent evaluations, not a step latency).

Prints ONE JSON line on rank 0.  python bench.py [--gpus N --steps K --warmup W]
With --gpus N > 1 and no torchrun environment (RANK unset) the script launches itself: the parent - before any GPU
call - starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` on this
file and exits with its return code; under torchrun (RANK set) it is the worker.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_POINT = 16           # x and y read once, fp64
PHI_KERNEL_NAMES = {6: "phi_sort_kernel<4, 6, 0, 1, 0> (Phi pass, algorithm 6: tile sort + register moments)",
                    62: "phi_sort_kernel<4, 4, 0, 1, 1> (Phi pass, algorithm 6 with the time-series front loop: per-wave run sums, no sort)",
                    5: "phi_moment_kernel<4, 2048, true> (Phi pass, algorithm 5)",
                    3: "phi_band_kernel (Phi pass, algorithm 3)", 1: "phi_band_kernel (Phi pass, algorithm 1)"}


def synth(N, seed=1234):
    """BASELINE.md synthetic inputs: x ~ U(1e-9, 1-1e-9) i.i.d. unsorted, y = sin(20x) + 0.1 eps, default_rng(1234)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(1e-9, 1 - 1e-9, N)
    y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
    return x, y


def measured_traffic(n_local, algo, series=False):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs,
    KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B streaming reads at 64 B).  The PMC passes
    cannot run inside this process; the committed summary profiles/r04_phi_traffic[_sorted].json (same kernel, same per-rank workload,
    taken with tools/collect_profiles.sh) is quoted when kernel and workload match, otherwise null.  It is a labelled constant
    from the committed profile, not a measurement of this run."""
    rel = "profiles/r04_phi_traffic%s.json" % ("_sorted" if series else "")
    path = os.path.join(ROOT, rel)
    try:
        d = json.load(open(path))
        if int(d["points_per_launch"]) == int(n_local) and int(d.get("phi_algorithm", -1)) == int(algo):
            return d["hbm_bytes_per_launch"], rel
    except Exception:
        pass
    return None, None


def cpu_baseline(x, y, M, theta, kind, gpu_stats=None, gpu_out=None):
    """The reference's CPU steps restated in numpy/scipy (oracle/, kind 'port'): piece polynomials -> csr_matrix ->
    Phi@y, Phi@Phi.T -> band (gpr.py:39-44) + one banded ELBO+gradient, single core, on (a prefix of) the SAME points the
    GPU processed.  When the sample is the whole workload the oracle's numbers double as the parity check of this very
    run: `parity` = statistics / ELBO / gradient of the timed GPU path against the oracle (fp64, reference elimination
    order) and against the oracle's long-double evaluation of the same recurrences."""
    from oracle import asvgp_oracle as O
    n = x.shape[0]
    bs = O.Basis(4, 0, 1, M)
    t0 = time.perf_counter()
    A, b, yy = O.sufficient_stats(bs, x.reshape(-1, 1), y.reshape(-1, 1))
    t1 = time.perf_counter()
    oe, og, _ = O.elbo_grad_1d(bs, kind, A, b, yy, n, *theta)
    t2 = time.perf_counter()
    base = dict(value=n / (t2 - t0) / 1e6, unit="Mpoints/s", cores=1, kind="port",
                sample="N=%d of the same synthetic workload (same seed, same points), M=%d: scipy CSR build + SpGEMM %.2fs, "
                       "python banded ELBO+grad %.2fs" % (n, M, t1 - t0, t2 - t1))
    parity = None
    if gpu_stats is not None:
        ref = np.concatenate([A.reshape(-1), b.reshape(-1), [yy]])
        ee, ge = O.elbo_grad_1d_extended(bs, kind, A, b, yy, n, *theta)
        g = np.asarray(gpu_out[1:4])
        parity = {"stats_max_abs_over_max": float(np.max(np.abs(gpu_stats - ref)) / np.max(np.abs(ref))),
                  "elbo_gpu": float(gpu_out[0]), "elbo_oracle_f64": oe, "elbo_oracle_long_double": ee,
                  "abs_elbo_vs_oracle": abs(float(gpu_out[0]) - oe), "abs_elbo_vs_long_double": abs(float(gpu_out[0]) - ee),
                  "rel_elbo_vs_long_double": abs(float(gpu_out[0]) - ee) / abs(ee),
                  "abs_oracle_vs_long_double": abs(oe - ee),
                  "grad_max_rel_vs_oracle": float(np.max(np.abs((g - og) / og))),
                  "grad_max_rel_vs_long_double": float(np.max(np.abs((g - ge) / ge))),
                  "gates": "stats 1e-12 of the largest entry; |ELBO - long double| <= 1e-9 |ELBO|; gradient rel 1e-6 (the fp64-oracle "
                           "differences are information: the fp64 reference order itself sits ~1e-8 |ELBO| from the long-double value)"}
    return base, parity


_CPU_WORKER = r"""
import sys, time, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import asvgp_oracle as O
N, M, C, w, srt = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
rng = np.random.default_rng(1234)
x = rng.uniform(1e-9, 1 - 1e-9, N)
y = np.sin(20 * x) + 0.1 * rng.standard_normal(N)
if srt:
    o = np.argsort(x); x, y = x[o], y[o]
lo, hi = N * w // C, N * (w + 1) // C
bs = O.Basis(4, 0, 1, M)
print("READY", flush=True)
sys.stdin.readline()
t0 = time.time()
A, b, yy = O.sufficient_stats(bs, x[lo:hi].reshape(-1, 1), y[lo:hi].reshape(-1, 1))
t1 = time.time()
print(json.dumps({"t0": t0, "t1": t1, "n": hi - lo}), flush=True)
"""


def cpu_baseline_all_cores(N, M, srt):
    """The same CPU steps on EVERY host core at once (SURVEY 8d asks for both figures): one process per core, each building the CSR
    design matrix and the band statistics of its contiguous N / C shard (the N-dependent work: what shards over ranks; the O(M) band
    algebra is not repeated per core).  Started before this process touches the GPU; the workers never do."""
    import subprocess
    C = os.cpu_count() or 1
    try:
        C = len(os.sched_getaffinity(0))
    except Exception:
        pass
    C = max(1, min(C, 64))
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, str(N), str(M), str(C), str(w), "1" if srt else "0"],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1"))
             for w in range(C)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "READY":
                raise RuntimeError("worker failed to start")
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
        for p in procs:
            p.wait(timeout=60)
        wall = max(r["t1"] for r in res) - min(r["t0"] for r in res)
        return {"value": N / wall / 1e6, "unit": "Mpoints/s", "cores": C, "seconds": wall,
                "what": "statistics pass only (CSR build + SpGEMM + band) of the whole workload, one single-threaded process per core on contiguous N / C shards"}
    except Exception as exc:
        for p in procs:
            try:
                p.kill()
            except Exception:
                pass
        return {"error": repr(exc)[:200], "cores": C}


def host_description():
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = None
    return {"cpu_model": model, "nproc": os.cpu_count(), "cores_available_to_this_process": aff}


def self_launch(argv, n):
    """Parent of an N-GPU run: no GPU call has happened in this process; the workers are torchrun children."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def spread(ms_list):
    a = np.sort(np.asarray(ms_list, dtype=np.float64))
    return {"median": float(np.median(a)), "min": float(a[0]), "max": float(a[-1]), "blocks": [round(float(v), 5) for v in ms_list]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=11, help="the --steps block is timed this many times; median / min / max are reported")
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--features", type=int, default=2048)
    ap.add_argument("--sorted", action="store_true", help="secondary case: time-series (sorted) inputs")
    ap.add_argument("--matern", type=int, default=32, choices=(12, 32, 52), help="kernel of the workload (BASELINE config 3 uses 52)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="points of the workload the CPU oracle is timed on (a prefix; "
                    "parity is reported when it covers the whole workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-cores CPU figure (one oracle process per host core)")
    ap.add_argument("--no-extras", action="store_true", help="skip dependent_steps_all_gpu and the emulated_shard extras")
    ap.add_argument("--band-algo", type=int, default=0, choices=(0, 1, 2, 3, 4), help="0 auto (matrix-core chains, planned prior), 1 sequential sweeps, "
                    "2 all-GPU block cyclic reduction, 3 planned prior chain (round-2 kernels), 4 matrix-core chains or error")
    ap.add_argument("--phi-algo", type=int, default=0, choices=(0, 1, 3, 5, 6))
    ap.add_argument("--in-flight", type=int, default=5, choices=tuple(range(0, 25)), help="extra `independent_evaluations`: steps in flight (0 = skip)")
    ap.add_argument("--chain-streams", type=int, default=2, help="M-side streams of the independent-evaluations extra")
    ap.add_argument("--phi-workgroups", type=int, default=240, help="Phi grid of the overlapped schedules (the chain workgroups need free CUs)")
    ap.add_argument("--no-three-sets", action="store_true", help="skip the extra dependent schedule with three buffer sets")
    ap.add_argument("--worker-forward", action="store_true", help="dependent schedules: the host forward pass of the Kuu chain on the handle's worker thread (asvgp_set_deferred_forward_pass(h, 2))")
    ap.add_argument("--value-launch-ahead", type=int, default=0, help="1: the two-set dependent schedule (`value`) with the ELBO launch enqueued ahead of theta")
    ap.add_argument("--reduce-stream", type=int, default=0, help="1: the Phi pass's cross-workgroup reduce on a third stream in the two-set dependent schedule")
    ap.add_argument("--no-launch-ahead", action="store_true", help="dependent schedules: launch the ELBO kernel only once theta is known (round 3's order) instead of ahead of it")
    ap.add_argument("--no-mirror", action="store_true", help="dependent schedule: read results through the stream (D2H copy + sync) instead of the pinned mirror")
    ap.add_argument("--kernel-events", type=int, default=5, help="HIP events around every n-th Phi kernel launch")
    ap.add_argument("--phase-events", type=int, default=25, help="one-at-a-time schedule: per-phase events on every n-th step (0 = never)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("ASVGP_BENCH_DRY"):
        # launcher rehearsal on a box without a GPU (tests/test_cabi_and_host.py): rendezvous, one all-reduce, one line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "ranks_seen": int(t.item())}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    cpu_all = None
    if world == 1 and not args.no_cpu_baseline and not args.no_cpu_all_cores:
        cpu_all = cpu_baseline_all_cores(args.points, args.features, args.sorted)   # (before this process touches the GPU)
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI ("nccl" is RCCL on ROCm); ASVGP_BENCH_BACKEND=gloo only for rehearsing the script on one GPU
        dist.init_process_group(os.environ.get("ASVGP_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    import asvgp_amd as A
    from asvgp_amd import _lib
    from asvgp_amd.dist import shard_bounds
    lib = _lib.get_lib()

    N, M = args.points, args.features
    K_steps, R = args.steps, max(1, args.repeats)
    theta0 = (1.0, 0.05, 0.01)
    x, y = synth(N)
    if args.sorted:
        o = np.argsort(x)
        x, y = x[o], y[o]
    lo, hi = shard_bounds(N, world, rank)
    n_local = hi - lo
    xd = torch.from_numpy(x[lo:hi].copy()).cuda().reshape(-1, 1)
    yd = torch.from_numpy(y[lo:hi].copy()).cuda().reshape(-1, 1)
    basis = A.B4Spline(0, 1, M)
    Kern = {12: A.Matern12, 32: A.Matern32, 52: A.Matern52}[args.matern]

    def new_model(n_total, overlapped, defer=True, n_points=None, prior_forward=0):
        xx, yy_ = (xd, yd) if n_points is None else (xd[:n_points], yd[:n_points])
        mm = A.GPR_1d((xx, yy_), Kern(variance=theta0[0], lengthscales=theta0[1]), basis)
        if prior_forward:
            mm._h.set_prior_forward(prior_forward)
        mm.likelihood.variance.assign(theta0[2])
        mm.num_data = n_total
        if args.band_algo:
            mm._h.set_band_algorithm(args.band_algo)
        if args.phi_algo:
            mm._h.set_phi_algorithm(args.phi_algo)
        if overlapped:
            mm._h.set_phi_workgroups(args.phi_workgroups)
            if defer:
                mm._h.set_phi_deferred_reduce(1)     # the N-side stream carries the streaming kernel only; the reduce rides with the M-side
        return mm

    def set_theta(mm, th):
        mm.kernel.variance.assign(th[0])
        mm.kernel.lengthscales.assign(th[1])
        mm.likelihood.variance.assign(th[2])

    def timed_blocks(block_fn, h, settle=None):
        """args.warmup untimed steps, then R blocks of EXACTLY K steps, each between barrier + synchronize pairs; max over ranks per block.
        HIP events around every n-th Phi launch of handle h inside the timed blocks."""
        block_fn(args.warmup)
        if settle:
            settle()
        torch.cuda.synchronize()
        lib.asvgp_profile_enable(h.ptr, args.kernel_events)
        ms = []
        for _ in range(R):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            block_fn(K_steps)
            if settle:
                settle()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = t.item()
            ms.append(dt / K_steps * 1e3)
        ms_sum, launches = ctypes.c_double(0), ctypes.c_int64(0)
        lib.asvgp_profile_read(h.ptr, ctypes.byref(ms_sum), ctypes.byref(launches))
        lib.asvgp_profile_enable(h.ptr, 0)
        return ms, ms_sum.value / max(launches.value, 1) * 1e3, int(launches.value)

    # ---- stream ceiling: a read-only pass over the same x, y with the Phi pass's launch shape, HIP events per launch
    sink = torch.zeros(8, dtype=torch.float64, device="cuda")
    st_main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    probe_us = []
    for i in range(23):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.asvgp_stream_probe(xd.data_ptr(), yd.data_ptr(), n_local, sink.data_ptr(), st_main), "stream_probe")
        e1.record()
        e1.synchronize()
        if i >= 3:
            probe_us.append(e0.elapsed_time(e1) * 1e3)
    ceiling_us = float(np.median(probe_us))
    ceiling_gbs = BYTES_PER_POINT * n_local / (ceiling_us * 1e-6) / 1e9

    # ---- schedule A: one step at a time on one stream, theta fixed (the host enqueues ahead): device-side latency of a full step
    model = new_model(N, overlapped=False)
    stats = model._stats
    marks = []
    n_marks = (args.warmup + K_steps * R) // max(args.phase_events, 1) + 2 if args.phase_events > 0 else 0
    ev_pool = [torch.cuda.Event(enable_timing=True) for _ in range(4 * n_marks)]   # (made before the timed region; sampled: a record costs stream time)
    ev = ev_pool.pop
    count = [0]

    def serial_block(k):
        for _ in range(k):
            rec = args.phase_events > 0 and count[0] % args.phase_events == 0 and len(ev_pool) >= 4
            count[0] += 1
            if rec:
                e0, e1, e2, e3 = ev(), ev(), ev(), ev()
                e0.record()
            model.phi_pass(allreduce=False)
            if rec:
                e1.record()
            if world > 1:
                dist.all_reduce(stats, op=dist.ReduceOp.SUM)
            if rec:
                e2.record()
            model._launch_elbo()
            if rec:
                e3.record()
                marks.append((e0, e1, e2, e3))

    ser_ms, kern_us, n_launches = timed_blocks(serial_block, model._h)
    model._check_pd()
    phi_algo_ran = model._h.phi_last_algorithm()
    mk = marks[1:] if len(marks) > 1 else marks          # (the first sampled step carries plan creation and allocations: excluded)
    t_phi = float(np.median([a.elapsed_time(b) for a, b, _, _ in mk]) * 1e3) if mk else 0.0
    t_comm = float(np.median([b.elapsed_time(c) for _, b, c, _ in mk]) * 1e3) if mk else 0.0
    t_band = float(np.median([c.elapsed_time(d) for _, _, c, d in mk]) * 1e3) if mk else 0.0
    phi_order_ran = model._h.phi_last_input_order() if model._h.phi_last_algorithm() == 6 else 0
    out4 = model._out.cpu().numpy()
    stats_host = stats.cpu().numpy()

    # ---- schedule C (`value`): dependent steps.  theta_{i+1} = theta_0 (1 + 1e-6 delta(result_i)) is computed on the host from the result of
    # step i that the host has read, so nothing of step i+1 that needs theta (ELBO launch, host forward pass) can start earlier.  The Phi
    # pass, its cross-workgroup reduce and the [all-reduce] need no theta and run on a second stream into another buffer set:
    #   two buffer sets (`value`): per step, in host order - Phi KERNEL of step i+1 (N stream; its buffer set was released by the result
    #     just read; no theta in it) -> theta_i -> ELBO + gradient launch of step i (M stream, behind the statistics event of its set; the
    #     call launches first and then runs the host's long-double forward pass: the Kuu workgroup waits for the table, the P chain does
    #     not; --worker-forward moves the pass to the handle's worker thread - measured: no gain in this order) -> reduce, [all-reduce],
    #     event of step i+1 (N stream) -> the host polls the pinned result mirror of step i.
    #   three buffer sets (extra `dependent_steps_phi_two_ahead`): ELBO launch of step i first, then the whole N side of step i+2.
    # Streams are made once and shared by every schedule of this process: HIP maps streams onto a handful of hardware queues round robin,
    # and a schedule whose N-side and M-side streams land on the SAME queue runs them in series (measured: the third or fourth schedule
    # of a process took 100-130 us per step instead of 72, its Phi kernel "70 us" - waiting behind the ELBO launch).
    stream_pool = {"n": torch.cuda.Stream(), "m": torch.cuda.Stream(priority=-1), "m2": torch.cuda.Stream(priority=-1), "r": torch.cuda.Stream()}

    def dependent_schedule(n_sets, n_points=None, prior_forward=0, launch_ahead_mode=False):
        lanes = [new_model(N, overlapped=True, defer=(n_sets == 2), n_points=n_points, prior_forward=prior_forward) for _ in range(n_sets)]
        if args.worker_forward and not args.no_mirror:
            for ln in lanes:
                ln._h.set_deferred_forward_pass(2)                 # the handle's worker thread runs the host forward pass (measured: no gain in this order)
        s_n, s_m = stream_pool["n"], stream_pool["m"]              # (made ONCE: see stream_pool)
        # --reduce-stream: the cross-workgroup reduce (+ all-reduce) of a set goes to a THIRD stream behind its Phi kernel, so that the Phi kernel
        # of the next step is not queued behind it (the reduce is 784 small workgroups that fit beside the Phi kernel's one workgroup per CU)
        s_r = stream_pool["r"] if (args.reduce_stream and n_sets == 2) else s_n
        ev_phi = [torch.cuda.Event() for _ in range(n_sets)]
        ev_stats = [torch.cuda.Event() for _ in range(n_sets)]
        ahead = n_sets - 1
        state = {"i": 0, "theta": theta0, "primed": False, "last": None, "t_a": 0.0, "t_b": 0.0, "t_poll": 0.0, "n": 0}

        def n_side_kernel(k):                                      # Phi pass (two sets: the streaming kernel alone, its reduce parked)
            _lib.set_stream(s_n)
            lanes[k].phi_pass(allreduce=False)
            if s_r is not s_n:
                ev_phi[k].record(s_n)

        def n_side_rest(k):                                        # reduce (if parked), [all-reduce], "statistics complete"
            if s_r is not s_n:
                s_r.wait_event(ev_phi[k])
            _lib.set_stream(s_r)
            lanes[k].phi_reduce()                                  # (a no-op when nothing is parked)
            if world > 1:
                with torch.cuda.stream(s_r):
                    dist.all_reduce(lanes[k]._stats, op=dist.ReduceOp.SUM)
            ev_stats[k].record(s_r)

        ahead_ok = launch_ahead_mode and not args.no_launch_ahead and not args.no_mirror and not prior_forward and not args.worker_forward
        if ahead_ok:
            for ln in lanes:
                ln._h.set_deferred_forward_pass(1)                 # publish_theta() fills the box only; publish_forward() runs the pass

        def launch_ahead(k):                                       # the NEXT evaluation's launch, enqueued before its theta exists
            if not ahead_ok:
                return None
            if not ev_stats[k].query():
                s_m.wait_event(ev_stats[k])
            _lib.set_stream(s_m)
            return lanes[k].launch_elbo_ahead()                    # (None: the matrix-core launch does not apply - the ordinary launch follows)

        def block(k):
            if not state["primed"]:
                for j in range(ahead):
                    n_side_kernel((state["i"] + j) % n_sets)
                    n_side_rest((state["i"] + j) % n_sets)
                state["primed"] = True
            if state.get("tok") is None:                            # (no launch is left waiting across a block boundary: the timed region
                state["tok"] = launch_ahead(state["i"] % n_sets)    #  ends with a device synchronisation)
            for it in range(k):
                i = state["i"]
                ln, nxt = lanes[i % n_sets], (i + ahead) % n_sets
                t0 = time.perf_counter()
                tok = state.get("tok")
                if tok is not None:
                    state["n_ahead"] = state.get("n_ahead", 0) + 1
                    ln.publish_theta(state["theta"])                # the launch is out already (resident, waiting): theta box FIRST (a few us) ...
                if n_sets == 2:
                    n_side_kernel(nxt)                              # ... then the Phi kernel of step i+1 (its buffer set was released by the result just read) ...
                if tok is not None:
                    ln._h.publish_forward()                         # ... then the host's long-double forward pass for the waiting Kuu workgroup (~19 us)
                if args.no_mirror:
                    set_theta(ln, state["theta"])
                if tok is None:
                    if not ev_stats[i % n_sets].query():            # (complete unless the N side is the slower one: then the launch waits in-stream)
                        s_m.wait_event(ev_stats[i % n_sets])
                    _lib.set_stream(s_m)
                    if args.no_mirror:
                        ln._launch_elbo()
                    else:
                        tok = ln.launch_elbo_host(state["theta"])   # (the trial point goes straight into the launch)
                t1 = time.perf_counter()
                if n_sets != 2:
                    n_side_kernel(nxt)
                n_side_rest(nxt)
                state["tok"] = launch_ahead((i + 1) % n_sets) if it + 1 < k else None   # step i+1's launch: behind step i's kernel and the statistics of its set
                t2 = time.perf_counter()
                if tok is None:
                    with torch.cuda.stream(s_m):
                        ln._check_pd(ln._launch_elbo)
                        r = ln._out[:4].tolist()
                else:
                    r = ln.read_elbo_host(tok)
                t3 = time.perf_counter()
                d = (r[0] * 1e3) % 1.0 - 0.5                       # any function of the host-read result; 1e-6 relative keeps the workload
                state["theta"] = tuple(t * (1.0 + 1e-6 * d) for t in theta0)
                state["last"] = r
                state["i"] = i + 1
                if k == K_steps:                                    # (timed blocks only: the warm-up carries one-time allocations)
                    state["t_a"] += t1 - t0; state["t_b"] += t2 - t1; state["t_poll"] += t3 - t2; state["n"] += 1
            _lib.set_stream(None)

        ms, kern_us, launches = timed_blocks(block, lanes[0]._h)
        n_acc = max(state["n"], 1)
        out = {"ms": ms, "kern_us": kern_us, "launches": launches, "last": state["last"], "launch_ahead": bool(ahead_ok and state.get("n_ahead", 0) > 0), "ahead_launches": state.get("n_ahead", 0),
               "fallbacks": sum(getattr(ln, "fused_launch_fallbacks", 0) for ln in lanes),
               "host_us": {("phi_kernel_enqueue_theta_and_elbo_launch" if n_sets == 2 else "theta_and_elbo_launch"): state["t_a"] / n_acc * 1e6,
                           ("reduce_and_event_enqueue" if n_sets == 2 else "n_side_enqueue"): state["t_b"] / n_acc * 1e6,
                           "wait_for_result": state["t_poll"] / n_acc * 1e6}}
        torch.cuda.synchronize()
        for ln in lanes:
            ln.close()                                             # (handles with their pinned rings and worker threads go now, not at garbage collection)
        del lanes
        import gc
        gc.collect()
        torch.cuda.synchronize()
        return out

    dep, dep_error, dep3, dep3_error = None, None, None, None
    for n_sets in ((2,) if args.no_three_sets else (2, 3)):
        try:
            res = dependent_schedule(n_sets, launch_ahead_mode=bool(args.value_launch_ahead) and n_sets == 2)
            if n_sets == 2:
                dep = res
            else:
                dep3 = res
        except Exception as exc:   # the contract line must come out: fall back to the one-at-a-time figures and say why
            if n_sets == 2:
                dep_error = repr(exc)[:300]
            else:
                dep3_error = repr(exc)[:300]
            _lib.set_stream(None)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    # ---- schedule B (extra): L steps with the SAME theta in flight - a throughput of independent evaluations.  N-side stream: Phi pass of
    # step i+1; M-side streams in turn: reduce, [all-reduce], ELBO launch of steps i, i-1.  Every step is a complete evaluation from the raw
    # points into its own buffers; a lane is reused once the host has seen its previous step finish.
    ind, ind_error = None, None
    if args.in_flight >= 2:
        try:
            lanes = [[new_model(N, overlapped=True), torch.cuda.Event(), torch.cuda.Event(), False] for _ in range(args.in_flight)]
            s_phi = stream_pool["n"]
            s_chains = [stream_pool["m"], stream_pool["m2"]][:max(1, min(2, args.chain_streams))]
            turn = [0]

            def independent_block(k):
                for _ in range(k):
                    j = turn[0]
                    turn[0] += 1
                    mm, ev_s, ev_done, used = lanes[j % len(lanes)]
                    if used:
                        ev_done.synchronize()            # host-side: the chains of step j-L have consumed this lane's buffers
                    with torch.cuda.stream(s_phi):
                        mm.phi_pass(allreduce=False)
                        ev_s.record(s_phi)
                    sc = s_chains[j % len(s_chains)]
                    with torch.cuda.stream(sc):
                        sc.wait_event(ev_s)
                        mm.phi_reduce()
                        if world > 1:
                            dist.all_reduce(mm._stats, op=dist.ReduceOp.SUM)
                        mm._launch_elbo()
                        ev_done.record(sc)
                    lanes[j % len(lanes)][3] = True

            ind_ms, _, _ = timed_blocks(independent_block, lanes[0][0]._h)
            outs = [ln[0]._out.cpu().numpy() for ln in lanes]
            for ln in lanes:
                ln[0]._check_pd()
            ind = {"ms": ind_ms, "max_rel_diff_vs_serial": float(max(np.max(np.abs(o[:4] - out4[:4]) / np.abs(out4[:4])) for o in outs))}
            del lanes
        except Exception as exc:
            ind, ind_error = None, repr(exc)[:300]
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    # ---- extras on the dependent schedule (VERDICT r3 #2a, #4): the same schedule with the Kuu chain's forward pass on the GPU in
    # double-double (no host arithmetic in the step), and with one rank's share of 2 / 4 / 8 ranks (the first N/2, N/4, N/8 points; world
    # size 1, no collective): what a rank of a strong-scaling run does per step before any collective latency.
    extras = {}
    if dep is not None and not args.no_extras and world == 1:   # (the extras are rows of the N = 1 line; a multi-rank run keeps to the contract's schedules)
        try:
            res = dependent_schedule(2, prior_forward=1)
            mg = new_model(N, overlapped=False, prior_forward=1)
            fixed = mg.elbo_and_grad().cpu().numpy()
            del mg
            extras["dependent_steps_all_gpu"] = {"res": res, "fixed_theta_out": [float(v) for v in fixed]}
        except Exception as exc:
            extras["dependent_steps_all_gpu_error"] = repr(exc)[:300]
            _lib.set_stream(None)
        try:
            extras["dependent_steps_launch_ahead"] = dependent_schedule(3, launch_ahead_mode=True)
        except Exception as exc:
            extras["dependent_steps_launch_ahead_error"] = repr(exc)[:300]
            _lib.set_stream(None)
        shards = {}
        for div in tuple(int(v) for v in os.environ.get("ASVGP_BENCH_SHARDS", "2,4,8").split(",")):
            try:
                npts = (n_local // div) & ~1
                res = dependent_schedule(2, n_points=npts, launch_ahead_mode=True)
                shards[str(div)] = {"res": res, "points": npts}
            except Exception as exc:
                shards[str(div)] = {"error": repr(exc)[:300]}
                _lib.set_stream(None)
        extras["emulated_shard"] = shards
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        # The Phi kernel on the OTHER input order SURVEY 8d names (a time series: the same points sorted by x; or, with --sorted, the
        # unsorted points): HIP events around the kernel alone, same handle mechanism as `roofline` - the driver line then carries both.
        if world == 1:
            try:
                if args.sorted:
                    perm = torch.randperm(n_local, device="cuda")
                else:
                    perm = torch.argsort(xd.reshape(-1))
                xo, yo = xd.reshape(-1)[perm].contiguous().reshape(-1, 1), yd.reshape(-1)[perm].contiguous().reshape(-1, 1)
                om = A.GPR_1d((xo, yo), Kern(variance=theta0[0], lengthscales=theta0[1]), basis)
                for _ in range(3):
                    om.phi_pass(allreduce=False)
                torch.cuda.synchronize()
                lib.asvgp_profile_enable(om._h.ptr, 1)
                for _ in range(20):
                    om.phi_pass(allreduce=False)
                torch.cuda.synchronize()
                ms_sum, launches = ctypes.c_double(0.0), ctypes.c_int64(0)
                lib.asvgp_profile_read(om._h.ptr, ctypes.byref(ms_sum), ctypes.byref(launches))
                lib.asvgp_profile_enable(om._h.ptr, 0)
                kus = ms_sum.value / max(launches.value, 1) * 1e3
                order_ran = om._h.phi_last_input_order() if hasattr(om._h, "phi_last_input_order") else 0
                ach = BYTES_PER_POINT * n_local / (kus * 1e-6) / 1e9 if kus > 0 else 0.0
                extras["phi_kernel_other_input_order"] = {"input": "unsorted (random permutation)" if args.sorted else "time series (the same points sorted by x)",
                                                          "kernel": PHI_KERNEL_NAMES.get(62 if order_ran == 2 else 6, "tile sort"), "kernel_us": kus, "launches": int(launches.value),
                                                          "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                                          "what": "HIP events around the Phi kernel alone (the handle's profile ring), 20 passes; the statistics are the same for any order (tests)"}
                om.close(); del om, xo, yo, perm
            except Exception as exc:
                extras["phi_kernel_other_input_order_error"] = repr(exc)[:300]
        # Two more rows of SURVEY section 8 measured in the same process (rank 0, world size 1 only): the streaming posterior over 10M test
        # points with this model's theta, and BASELINE configs[3]'s shape (2-D Kronecker, N = 1M, 128 x 128, B3) - Phi pass, bound, bound +
        # gradient.  Not `value`; they put the kernels' figures into the driver's own record.
        if world == 1:
            try:
                pm = new_model(N, overlapped=False)
                xs = torch.rand(10_000_000, dtype=torch.float64, device="cuda") * 0.998 + 0.001
                pm.predict_f_device(xs.reshape(-1, 1)); torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
                ev[0].record()
                for i in range(10):
                    pm.predict_f_device(xs.reshape(-1, 1)); ev[i + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(10))
                extras["posterior_10m_points"] = {"us_per_call": ts[5], "min_us": ts[0], "bytes_per_point": 24, "achieved_gbs": 24 * 1e7 / (ts[5] * 1e-6) / 1e9,
                                                  "frac_of_hbm_peak": 24 * 1e7 / (ts[5] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                  "what": "GPR_1d.predict_f_device on 10M i.i.d. test points (8 B in, 16 B out), posterior cached; HIP events around the calls, median of 10"}
                pm.close(); del pm, xs
            except Exception as exc:
                extras["posterior_10m_points_error"] = repr(exc)[:300]
            try:
                g = torch.Generator(device="cuda").manual_seed(4321)
                Xk = torch.rand((1_000_000, 2), dtype=torch.float64, device="cuda", generator=g) * (1 - 2e-6) + 1e-6
                yk = torch.sin(12 * Xk[:, :1]) * torch.cos(9 * Xk[:, 1:]) + 0.1 * torch.randn((1_000_000, 1), dtype=torch.float64, device="cuda", generator=g)
                Xk, yk = Xk.to(torch.float32), yk.to(torch.float32)           # (configs[3] names fp32 data: fp32 storage, fp64 arithmetic)
                km = A.GPR_kron((Xk, yk), [A.Matern32(variance=1.0, lengthscales=0.2), A.Matern32(variance=1.0, lengthscales=0.2)], [A.B3Spline(0, 1, 128), A.B3Spline(0, 1, 128)])
                km.likelihood.variance.assign(0.01)

                def med(f, n=5):
                    f(); torch.cuda.synchronize()
                    out = []
                    for _ in range(n):
                        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
                    return sorted(out)[n // 2]
                km.phi_pass(); torch.cuda.synchronize()
                evk = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
                evk[0].record()
                for i in range(10):
                    km.phi_pass(); evk[i + 1].record()
                torch.cuda.synchronize()
                tk = sorted(evk[i].elapsed_time(evk[i + 1]) * 1e3 for i in range(10))
                extras["kronecker_config4_shape"] = {"N": 1_000_000, "basis": "B3Spline 128 x 128 (M_tot = 16384, bandwidth 387)", "dtype": "f32 storage of (X, y) in the Phi pass, f64 arithmetic throughout",
                                                     "phi_pass_us": tk[5], "elbo_ms": med(lambda: km.elbo().item()) * 1e3,
                                                     "elbo_and_grad_ms": med(km.elbo_and_grad) * 1e3, "elbo": float(km.elbo().item()),
                                                     "factorisation": "two-sided" if km._twist_layout() else "one-sided",
                                                     "what": "phi_pass_us: HIP events around the pass (cell-sorted points: matrix-core cell sums + gather), median of 10; elbo / elbo_and_grad: wall clock per call incl. the host-side result read, median of 5 (tools/kron_probe.py measures the same)"}
                del km, Xk, yk
            except Exception as exc:
                extras["kronecker_config4_shape_error"] = repr(exc)[:300]
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    # Extra, N > 1 only: the one-at-a-time step with the BASELINE N on EVERY rank (weak scaling).  `value` stays the strong-scaling figure
    # the metric is quoted on; this field only shows what the replicated band chains cost in the other regime.
    weak = None
    if world > 1 and not os.environ.get("ASVGP_BENCH_NOWEAK"):
        try:
            xw, yw = synth(N, seed=1234 + rank)
            del model
            xd = torch.from_numpy(xw).cuda().reshape(-1, 1)
            yd = torch.from_numpy(yw).cuda().reshape(-1, 1)
            model = new_model(N * world, overlapped=False)
            stats = model._stats
            count[0] = 1                                  # (no phase events)
            pe, args.phase_events = args.phase_events, 0
            wk_ms, _, _ = timed_blocks(serial_block, model._h)
            args.phase_events = pe
            wk = spread(wk_ms)
            weak = {"value": N * world / (wk["median"] * 1e-3) / 1e6, "unit": "Mpoints/s", "points_per_rank": N,
                    "ms_per_step": wk, "scaling": "weak", "schedule": "one step at a time"}
        except Exception as exc:   # never let the extra measurement break the contract line
            weak = {"error": repr(exc)[:200]}

    if rank == 0:
        ser = spread(ser_ms)
        mp = lambda ms: N / (ms * 1e-3) / 1e6
        serial = {"ms_per_step": ser, "value": mp(ser["median"]), "unit": "Mpoints/s", "phi_kernel_us": kern_us,
                  "schedule": "one stream, theta fixed, the host enqueues ahead: Phi pass -> reduce -> [all-reduce] -> band chains + bound (one launch)"}
        if dep is not None:
            dsp = spread(dep["ms"])
            schedule = ("dependent steps: theta_{i+1} computed on the host from the host-read result of step i (pinned result mirror%s); "
                        "ELBO + gradient launch of step i on one stream, the theta-free part of step i+1 (Phi pass, reduce, [all-reduce]) on a "
                        "second stream under it (host order: Phi kernel of i+1, ELBO launch of i followed by the host's long-double forward pass for it, reduce of i+1); two buffer sets alternate; Phi grid %d workgroups"
                        % (" off: D2H copy + synchronise" if args.no_mirror else "", args.phi_workgroups))
        else:
            dsp = ser
            schedule = serial["schedule"] + " (dependent schedule failed: see dependent_schedule_error)"
        achieved = BYTES_PER_POINT * n_local / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        traffic, traffic_src = measured_traffic(n_local, phi_algo_ran, series=(phi_order_ran == 2))
        line = {
            "metric": "Mpoints/s per ELBO+grad step, N=10M 1D Matern-3/2 M=2048",
            "value": mp(dsp["median"]),
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": K_steps, "warmup": args.warmup,
            "ms_per_step": dsp["median"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "1D synthetic N=%d (U(0,1) i.i.d. %s), Matern-%d/2, B4Spline(0,1,M=%d) band k=4, theta=(1,0.05,0.01)"
                                   % (N, "sorted" if args.sorted else "unsorted", args.matern // 10, M),
                       "parallelism": "dp%d (contiguous N-shards, one all-reduce of the %d-double band buffer)" % (world, stats.numel()),
                       "points_per_rank": n_local, "schedule": schedule},
            "repeats": {"R": R, "steps_per_block": K_steps, "ms_per_step": dsp, "value_min": mp(dsp["max"]), "value_max": mp(dsp["min"]),
                        "note": "the --steps block timed R times in this process between barrier + synchronize pairs; value / ms_per_step = the median block"},
            "one_step_at_a_time": serial,
            "phases_us": {"phi_pass_and_reduce": t_phi, "band_allreduce": t_comm, "elbo_and_gradient_launch": t_band,
                          "samples": len(mk),
                          "note": "t_phi / t_comm / t_band of SURVEY 8d: HIP events in the one-step-at-a-time schedule, sampled every %d-th step; "
                                  "median, first sample (plan creation, allocations) excluded" % args.phase_events},
            "roofline": {"bound": "hbm", "kernel": PHI_KERNEL_NAMES.get(62 if (phi_algo_ran == 6 and phi_order_ran == 2) else phi_algo_ran, "Phi pass, algorithm %d" % phi_algo_ran), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "stream_ceiling": {"achieved": ceiling_gbs, "unit": "GB/s", "us": ceiling_us, "frac_of_peak": ceiling_gbs / HBM_PEAK_GBS,
                                            "kernel": "stream_probe_kernel: read-only 16 B/point over the same x, y, 256 x 1024 threads, non-temporal "
                                                      "16-byte loads (asvgp_stream_probe), median of 20 launches in this run"},
                         "frac_of_stream_ceiling": (achieved / ceiling_gbs) if ceiling_gbs > 0 else None,
                         "kernel_us": kern_us, "launches": n_launches, "algorithmic_bytes_per_launch": BYTES_PER_POINT * n_local,
                         "kernel_us_in_dependent_schedule": (dep["kern_us"] if dep else None),
                         "note": "HIP events around every %d-th launch inside the timed one-step-at-a-time blocks (the kernel alone on the device: its own "
                                 "duration, as in rocprofv3's kernel trace); in the dependent schedule the same events also contain the launch's wait "
                                 "for CUs next to the chain workgroups" % args.kernel_events},
            "elbo": float(out4[0]), "grad": [float(v) for v in out4[1:4]],
        }
        if dep is not None:
            line["dependent_schedule"] = {"host_us_per_step": dep["host_us"], "fused_launch_fallbacks": dep["fallbacks"],
                                          "last_result": dep["last"], "rel_diff_elbo_vs_fixed_theta": abs(dep["last"][0] - out4[0]) / abs(out4[0])}
        if dep_error is not None:
            line["dependent_schedule_error"] = dep_error
        if dep3 is not None:
            d3 = spread(dep3["ms"])
            line["dependent_steps_phi_two_ahead"] = {"value": mp(d3["median"]), "unit": "Mpoints/s", "ms_per_step": d3, "host_us_per_step": dep3["host_us"],
                                                     "fused_launch_fallbacks": dep3["fallbacks"],
                                                     "note": "the same dependent steps with THREE buffer sets: the theta-free part of step i+2 runs under the ELBO launch of "
                                                             "step i, so the N side never sits between two ELBO launches; reported beside `value`, which keeps to step i+1"}
        if dep3_error is not None:
            line["dependent_steps_phi_two_ahead_error"] = dep3_error
        if "dependent_steps_all_gpu" in extras:
            eg = extras["dependent_steps_all_gpu"]
            gsp = spread(eg["res"]["ms"])
            line["dependent_steps_all_gpu"] = {
                "value": mp(gsp["median"]), "unit": "Mpoints/s", "ms_per_step": gsp, "host_us_per_step": eg["res"]["host_us"],
                "fused_launch_fallbacks": eg["res"]["fallbacks"], "fixed_theta_elbo_and_grad": eg["fixed_theta_out"],
                "rel_diff_elbo_vs_default_path": abs(eg["fixed_theta_out"][0] - out4[0]) / abs(out4[0]),
                "note": "the `value` schedule with asvgp_set_prior_forward(1): the Kuu chain's forward (elimination) pass on the GPU in double-double "
                        "(prior_dd.hip) instead of the host's x87 long double - no host arithmetic inside the step"}
        if "dependent_steps_all_gpu_error" in extras:
            line["dependent_steps_all_gpu_error"] = extras["dependent_steps_all_gpu_error"]
        if "dependent_steps_launch_ahead" in extras:
            ea = extras["dependent_steps_launch_ahead"]
            asp = spread(ea["ms"])
            line["dependent_steps_launch_ahead"] = {
                "value": mp(asp["median"]), "unit": "Mpoints/s", "ms_per_step": asp, "host_us_per_step": ea["host_us"], "launch_ahead_used": ea["launch_ahead"],
                "fused_launch_fallbacks": ea["fallbacks"], "rel_diff_elbo_vs_fixed_theta": abs(ea["last"][0] - out4[0]) / abs(out4[0]),
                "note": "the same dependent steps (three buffer sets) with the ELBO + gradient kernel of step i+1 LAUNCHED AHEAD of its theta "
                        "(asvgp_elbo_grad_ahead_1d): it is resident, waiting on the handle's pinned theta box, when the host has read result i and "
                        "publishes theta_{i+1} (asvgp_elbo_publish_theta: box first, Phi enqueue, then the host forward pass) - the launch path and "
                        "dispatch latency leave the critical path; nothing that needs theta is computed earlier.  It pays where the M side bounds the "
                        "step (sorted input, small shards); on the unsorted headline the N side bounds and the extra host calls cost more than they save"}
        if "dependent_steps_launch_ahead_error" in extras:
            line["dependent_steps_launch_ahead_error"] = extras["dependent_steps_launch_ahead_error"]
        for key in ("phi_kernel_other_input_order", "phi_kernel_other_input_order_error", "posterior_10m_points", "posterior_10m_points_error",
                    "kronecker_config4_shape", "kronecker_config4_shape_error"):
            if key in extras:
                line[key] = extras[key]
        if "emulated_shard" in extras:
            es = {}
            for div, e in extras["emulated_shard"].items():
                if "res" in e:
                    sp = spread(e["res"]["ms"])
                    es["1_of_%s" % div] = {"points": e["points"], "ms_per_step": sp, "phi_kernel_us": e["res"]["kern_us"],
                                           "whole_job_value_if_all_ranks_did_this": N / (sp["median"] * 1e-3) / 1e6}
                else:
                    es["1_of_%s" % div] = e
            es["note"] = ("the `value` schedule (two buffer sets) WITH launch-ahead on the first N/2, N/4, N/8 points with the global N in the bound (world size 1, "
                          "no collective): one rank's step of a 2 / 4 / 8-rank strong-scaling run before any collective latency; not a multi-GPU "
                          "measurement (without launch-ahead these read 69-70 us)")
            line["emulated_shard"] = es
        if ind is not None:
            isp = spread(ind["ms"])
            line["independent_evaluations"] = {"value": mp(isp["median"]), "unit": "Mpoints/s", "ms_per_step": isp, "in_flight": args.in_flight,
                                               "max_rel_diff_vs_one_at_a_time": ind["max_rel_diff_vs_serial"],
                                               "note": "same theta in every step: a throughput of independent evaluations, not an optimiser's step rate"}
        if ind_error is not None:
            line["independent_evaluations_error"] = ind_error
        if weak is not None:
            line["weak_scaling_extra"] = weak
        if not args.no_cpu_baseline and world == 1:
            ns = min(args.cpu_sample, N)
            full = (ns == N)
            okind = {12: 0, 32: 1, 52: 2}[args.matern]
            line["cpu_baseline"], parity = cpu_baseline(x[:ns], y[:ns], M, theta0, okind, stats_host if full else None, out4 if full else None)
            line["cpu_baseline"]["host"] = host_description()
            if cpu_all is not None:
                line["cpu_baseline"]["all_cores"] = cpu_all
            if parity is not None:
                if "dependent_steps_all_gpu" in line:   # the all-GPU chain against the same long-double evaluation
                    eg = line["dependent_steps_all_gpu"]["fixed_theta_elbo_and_grad"]
                    ee = parity["elbo_oracle_long_double"]
                    line["dependent_steps_all_gpu"]["parity"] = {"abs_elbo_vs_long_double": abs(eg[0] - ee), "rel_elbo_vs_long_double": abs(eg[0] - ee) / abs(ee),
                                                                 "gate": "|ELBO - long double| <= 1e-9 |ELBO|"}
                line["parity"] = parity
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
