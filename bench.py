#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched VoltageControl hot path on N MI355X of one node.

One "step" = one pass of the hot path over the whole env batch of every rank:
    actions -> q clip -> Sbus -> Newton-Raphson power flow -> results/reward/info -> next profile row
    + noise -> get_obs                      (reference: step() + get_obs(), models/model.py:216,219)
Synthetic case141 (141 buses / 84 loads / 22 PV agents), 4096 envs per GPU (BASELINE.json configs[2],
the configuration `metric` is quoted on), float64 arithmetic.  Episodes are 240 steps
(var_voltage_control.yaml:16): whenever the batch terminates it is reset inside the timed region
(the reset's own power flow is extra work that is NOT counted as env-steps).

`python bench.py --gpus N` launches its own N ranks (one process per GPU, torch.distributed over RCCL) when it is
not already running under torch.distributed.run; under torchrun (WORLD_SIZE set) it is simply rank RANK.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCALE = {"case33": 0.8, "case141": 0.6, "case322": 0.8, "case141_deep": 0.6}     # reference train.py:34-42 (+ the depth-stress topology)
HBM_PEAK_GBS = 8000.0                                         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP64_PEAK_TFLOPS = 78.6                                       # MI355X f64 vector peak
NR_KERNEL = "k_nr"                                            # substring of the dominant kernel's name


def algorithmic_bytes_per_env_step(env):
    """compulsory per-env traffic of one step()+get_obs(); shared constants excluded.
    in : p_load, q_load [nl], p_pv, action [ns]                  f64
    out: vm, va [nb] f64; obs [n_agents, obs_size] f32; reward f64 + terminated u8 + info[11] f64"""
    nl, ns, nb = env.n_load, env.n_sgen, env.n_bus
    return 8 * (2 * nl + 2 * ns) + 8 * 2 * nb + 4 * env.n_agents * env.obs_size + (8 + 1 + 8 * 11)


def solver_bytes_per_env_step(env):
    """what the k_nr_tree launch ITSELF must move per env: read Sbus 16 n + q 8 ns; write e, f 16 n, pl 8 n_line,
    q 8 ns, reward / terminated / info 97 B (n = non-slack buses, n_line = n on a radial feeder).  The obs / state outputs and the
    profile rows belong to k_gather / k_advance: they are in algorithmic_bytes_per_env_step, not here."""
    n, ns = env.n_bus - 1, env.n_sgen
    return 16 * n + 8 * ns + 16 * n + 8 * n + 8 * ns + (8 + 1 + 8 * 11)


# ------------------------------------------------------------------------------------------------ CPU baseline
def _oracle_args(case):
    return dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type="bowl", seed=0)


def _cpu_worker(args):
    """One oracle env on one core for `seconds`: sequential step()+get_obs() (restated pandapower runpp + env logic)."""
    case, seconds, env_id = args
    import numpy as np
    from mapdn_amd.netspec import make_case
    from oracle.env_restated import VoltageControlOracle
    net, prof = make_case(case)
    env = VoltageControlOracle(net, prof, _oracle_args(case), env_id=env_id)
    rng = np.random.default_rng(env_id)
    for _ in range(3):
        env.step(rng.uniform(-SCALE[case], SCALE[case], net.n_sgen)); env.get_obs()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        _, term, _ = env.step(rng.uniform(-SCALE[case], SCALE[case], net.n_sgen))
        env.get_obs()
        n += 1
        if term:
            env.reset()
    return n, time.perf_counter() - t0


class BatchedOracleShard:
    """E oracle envs stepped together: the E power flows of a step are ONE batched-numpy Newton-Raphson (oracle/batched_np.py),
    everything around it (clip, reward, info, profile advance, obs) stays the per-env restatement of the reference's env code."""

    def __init__(self, case, n_envs, first_env_id=0):
        from mapdn_amd.netspec import make_case
        from oracle import env_restated
        from oracle.batched_np import BatchedRunpp
        net, prof = make_case(case)
        self.net = net
        self.envs = [env_restated.VoltageControlOracle(net, prof, _oracle_args(case), env_id=first_env_id + e) for e in range(n_envs)]
        self.solver = BatchedRunpp(net)

    def step(self, actions):
        import numpy as np
        envs = self.envs
        qs = np.stack([e._clip_reactive_power(np.asarray(a, dtype=np.float64), e.sgen_p) for e, a in zip(envs, actions)])
        res = self.solver(np.stack([e.load_p for e in envs]), np.stack([e.load_q for e in envs]),
                          np.stack([e.sgen_p for e in envs]), qs)
        out = []
        for e, a, r in zip(envs, actions, res):
            e._runpp_override = r                                       # the env's own runpp call takes its share of the batch (an
            rew, term, info = e.step(a)                                 # explicit per-env hook: no process-wide state is patched)
            e.get_obs()
            if term:
                e.reset()                                               # (rare) the restart's power flow: the one-env solver
            out.append((rew, term, info))
        return out


def _cpu_worker_batched(args):
    case, seconds, shard, envs_per_shard = args
    import numpy as np
    sh = BatchedOracleShard(case, envs_per_shard, first_env_id=shard * envs_per_shard)
    rng = np.random.default_rng(1000 + shard)
    ns = sh.net.n_sgen
    sh.step(rng.uniform(-SCALE[case], SCALE[case], (envs_per_shard, ns)))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        sh.step(rng.uniform(-SCALE[case], SCALE[case], (envs_per_shard, ns)))
        n += envs_per_shard
    return n, time.perf_counter() - t0


def effective_cores():
    """(cores in the affinity mask, CPU quota of the cgroup in cores or None).  os.sched_getaffinity alone over-reports on a
    quota-limited container: the kernel then throttles every process (round 2 saw 256 'cores' deliver 15x one core)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),):
        try:
            quota = parse(open(path).read())
        except (OSError, ValueError, IndexError):
            pass
    if quota is None:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / p if q > 0 else None
        except (OSError, ValueError):
            pass
    return aff, quota


def _run_pool(target, arglist, timeout=900):
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    t0 = time.perf_counter()
    procs = [ctx.Process(target=lambda a=a: q.put(target(a)), daemon=True) for a in arglist]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join()
    return res, time.perf_counter() - t0


def cpu_baseline(case, seconds=10.0, envs_per_shard=64):
    """Restated pandapower-equivalent CPU path (oracle/: numpy + scipy; pandapower 2.7.0 itself is not installable offline,
    SURVEY.md 8(c), hence kind='port'), three ways as SURVEY.md 8(d)(2) asks:
      (1) one env, one core: scipy sparse NR (SuperLU) per step;
      (2) one such env per core, all cores at once;
      (3) the batched-numpy form: a shard of `envs_per_shard` envs per core whose power flows are one vectorised NR.
    `value` is the best all-core figure.  Cores = min(affinity mask, cgroup CPU quota).  Called BEFORE this process touches
    the GPU (workers are forked); BLAS / OpenMP pools are pinned to one thread so that processes do not oversubscribe."""
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[v] = "1"
    aff, quota = effective_cores()
    cores = max(1, min(aff, int(quota)) if quota else aff)
    (n1, dt1), = _run_pool(_cpu_worker, [(case, seconds * 0.2, 0)])[0]
    res2, wall2 = _run_pool(_cpu_worker, [(case, seconds * 0.3, e) for e in range(cores)])
    res3, wall3 = _run_pool(_cpu_worker_batched, [(case, seconds * 0.5, e, envs_per_shard) for e in range(cores)])
    n2, busy2 = sum(r[0] for r in res2), max(r[1] for r in res2)
    n3, busy3 = sum(r[0] for r in res3), max(r[1] for r in res3)
    v2, v3 = n2 / busy2, n3 / busy3
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.pop(v, None)
    return {"value": max(v2, v3), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_core_value": n1 / dt1,
            "all_cores_one_env_per_process": {"value": v2, "per_process": v2 / cores, "processes": cores},
            "all_cores_batched_numpy": {"value": v3, "per_process": v3 / cores, "processes": cores, "envs_per_process": envs_per_shard},
            "affinity_cores": aff, "cgroup_cpu_quota_cores": quota,
            "parallel_efficiency_vs_single_core": v2 / cores / (n1 / dt1),
            "sample": f"{case}: (1) 1 env on 1 core, scipy SuperLU NR: {n1} step()+get_obs() in {dt1:.1f} s; (2) {cores} processes x 1 env: "
                      f"{n2} in {busy2:.1f} s (pool wall {wall2:.1f} s); (3) {cores} processes x {envs_per_shard} envs, batched-numpy NR: "
                      f"{n3} in {busy3:.1f} s (pool wall {wall3:.1f} s).  cores = min(affinity {aff}, cgroup quota {quota}); "
                      f"numpy/scipy restatement of pandapower runpp + env logic"}


# ------------------------------------------------------------------------------------------------ live PMC traffic
def _pmc_pass(counter, case, envs, steps, outdir):
    """One rocprofv3 pass of a short inner bench run; returns the mean counter value (KiB) over the NR launches."""
    import csv
    import glob
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", outdir, "-o", counter.lower(), "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--inner", "--case", case, "--envs", str(envs), "--steps", str(steps),
           "--warmup", "3"]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
    vals = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and NR_KERNEL in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    vals = [v for v in vals if v > 0.25 * max(vals)] if vals else vals        # drop the early-exit reset retries
    if not vals:
        raise RuntimeError("no NR dispatches in the counter file")
    return sum(vals) / len(vals), len(vals)


def measure_traffic(case, envs):
    """HBM-side bytes per NR launch: FETCH_SIZE and WRITE_SIZE need separate rocprofv3 passes (MI355X guide: TCC has 4
    slots, FETCH_SIZE takes 3, WRITE_SIZE 2), each a short sub-run of this file.  Counter KiB x 1024, with the gfx950
    correction CALIBRATED on this kernel's own access pattern (16-byte raw-buffer loads / stores, 256 contiguous bytes per 16-lane
    worker: tools/calibrate_traffic.py, profiles/r04_fetch_write_size_calibration.txt): FETCH_SIZE reports exactly half of the
    bytes read, WRITE_SIZE the bytes written — so bytes = 2 x FETCH + WRITE; the raw sum is kept as `bytes_raw`."""
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    d = tempfile.mkdtemp(prefix="mapdn_pmc_")
    try:
        f, nf = _pmc_pass("FETCH_SIZE", case, envs, 40, os.path.join(d, "f"))
        w, nw = _pmc_pass("WRITE_SIZE", case, envs, 40, os.path.join(d, "w"))
        return {"bytes": (2 * f + w) * 1024.0, "fetch_bytes": 2 * f * 1024.0, "write_bytes": w * 1024.0,
                "bytes_raw": (f + w) * 1024.0, "fetch_counter_bytes": f * 1024.0, "launches": [nf, nw],
                "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate sub-runs of bench.py; FETCH_SIZE x 2 "
                          "(calibrated: profiles/r04_fetch_write_size_calibration.txt)"}
    except Exception as e:                                   # profiler unavailable in this harness: say so
        return {"bytes": None, "error": f"{type(e).__name__}: {e}"[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def committed_traffic(case, envs):
    for tag in ("r06_final", "r05_final", "r04_final", "r03_final", "r02_final", "r02_base", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_traffic_{case}_b{envs}.json")
        if os.path.exists(path):
            return json.load(open(path))["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, backend="nccl"):
    """--gpus N without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`."""
    if backend == "nccl":                                    # one rank per GPU over RCCL: fail here, with a sentence, not in N ranks' init
        import torch
        have = torch.cuda.device_count()
        if n > have:
            raise SystemExit(f"bench.py --gpus {n}: this node exposes {have} GPU(s) and the nccl (RCCL) backend needs one per rank; "
                             f"run with --gpus <= {have}, or --backend gloo to exercise the {n}-rank path with ranks sharing GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--case", default="case141")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two live rocprofv3 PMC sub-runs (traffic from profiles/)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm); 'gloo' only to "
                                                        "exercise the N>1 path on a box with fewer GPUs than ranks")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the timed --steps block until the blocks add up to this")
    ap.add_argument("--repeats", type=int, default=0, help="exactly this many timed blocks (0: repeat until --min-seconds)")
    ap.add_argument("--inner", action="store_true", help="(internal) short un-instrumented loop for the PMC sub-runs")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the short case33 / case322 measurements appended to the default line")
    ap.add_argument("--env-id-offset", type=int, default=0, help="global id of this run's first env (a 1-rank run that covers the ids "
                                                                   "of rank r of an N-rank run: r x envs); ranks add rank x envs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end MADDPG block appended to the default line (BASELINE configs[4]'s "
                                                          "per-GPU shard: case322 x 8192 envs, 3 episodes at the reference's update intensity)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (and run the end-of-rollout gather, the max-over-ranks "
                                                              "all_reduce and the barriers) even at --gpus 1: first contact with RCCL on one GPU")
    ap.add_argument("--dump-returns", default=None, help="rank 0 writes the gathered per-env episode returns of the LAST timed block "
                                                         "to this .npy file (pre-flight checks of the N > 1 path)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus, a.backend)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and not a.no_cpu_baseline and not a.inner:
        # before any GPU work: the workers are forked.  (N > 1: the other ranks wait for rank 0 at the rendezvous meanwhile; the
        # sample is halved and the line says that the host was shared with their start-up.)
        cpu = cpu_baseline(a.case, a.cpu_seconds if world == 1 else 0.5 * a.cpu_seconds)
        if world > 1:
            cpu["note"] = f"measured on rank 0 of {world} while the other ranks were starting up on the same host"

    import numpy as np  # noqa: F401
    import torch
    dist = None
    if a.backend != "nccl":                       # test mode: ranks may share a GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                  # --force-dist without a launcher: a one-rank rendezvous on the loopback
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from mapdn_amd.env import VoltageControlBatch
    from mapdn_amd.netspec import make_case
    from mapdn_amd.sharding import gather_rollout

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(case, B, min_seconds, inner=False):
        """The timed loop on one (case, envs per GPU): returns the block times (max over ranks), the env and its NR timing."""
        net, prof = make_case(case)
        args = dict(episode_limit=240, action_scale=SCALE[case], action_bias=0.0, voltage_barrier_type="bowl", seed=0)
        id0 = a.env_id_offset + rank * B                     # weak scaling: rank r owns the global env ids [id0, id0 + B)
        env = VoltageControlBatch(net, prof, args, n_envs=B, device=dev, env_id_offset=id0)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234 + id0 // B)                     # actions keyed by the block of global ids, not by the rank: a 1-rank run
                                                             # with --env-id-offset r x B replays rank r of an N-rank run exactly
        scale = SCALE[case]
        # fresh random actions for every step, drawn on the device BEFORE the timed region so that the
        # (excluded) policy costs nothing inside it; a ring of `n_act` distinct action tensors
        n_act = min(a.steps + a.warmup + 60, 256)
        acts = torch.empty(n_act, B, env.n_sgen, dtype=torch.float32, device=dev).uniform_(-scale, scale, generator=gen)
        steps_in_ep, step_no, resets = [0], [0], [0]
        last_returns = [None]

        def one_step():
            act = acts[step_no[0] % n_act]
            step_no[0] += 1
            env.step(act)
            env.get_obs()
            steps_in_ep[0] += 1
            if steps_in_ep[0] >= env.episode_limit - 1:       # all envs terminate together (:204)
                env.reset()
                resets[0] += 1
                steps_in_ep[0] = 0

        def timed_block():
            """EXACTLY a.steps steps between two fences (barrier + device sync on both sides); max over ranks"""
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                one_step()
            if dist is not None:                              # end-of-rollout RCCL gather, inside the timed region
                ret = env.episode_returns()
                allret = gather_rollout(ret if a.backend == "nccl" else ret.cpu(), force=True)
                assert allret.shape[0] == world * B
                last_returns[0] = allret
            elif a.dump_returns:
                last_returns[0] = env.episode_returns()
            fence()
            dt = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        env.reset()
        for _ in range(a.warmup):
            one_step()
        if inner:
            timed_block()
            env.close()
            return None
        # A block of `--steps` steps can be a few milliseconds (the driver's --steps 20 is 2.3 ms): the block is repeated, each
        # repeat fenced and timed on its own, until the timed blocks add up to >= min_seconds (every rank runs the same number:
        # the decision uses the max-over-ranks times).  `ms_per_step` / `value` are the MEDIAN block; min / max are reported.
        resets[0] = 0
        blocks = []
        while (len(blocks) < a.repeats) if a.repeats > 0 else (len(blocks) < 3 or (sum(blocks) < min_seconds and len(blocks) < 2000)):
            blocks.append(timed_block())
        resets_in_region = resets[0]
        stats = env.stats()
        # ---- dominant kernel (NR solve) duration, HIP events on its launch stream, separate short pass
        # (>= 240 step launches whatever --steps is: 20 launches gave +-5 %; the power flow of a reset — another launch shape, no fused
        # prologue — is kept OUT of the average: timing is off around it)
        env.nr_timing(True)
        for _ in range(max(240, min(a.steps, 480))):
            act = acts[step_no[0] % n_act]
            step_no[0] += 1
            env.step(act)
            env.get_obs()
            steps_in_ep[0] += 1
            if steps_in_ep[0] >= env.episode_limit - 1:
                env.nr_timing(False)
                env.reset()
                env.nr_timing(True)
                steps_in_ep[0] = 0
        torch.cuda.synchronize(dev)
        nr_ms, nr_launches = env.nr_time_ms()
        env.nr_timing(False)
        nr_avg_rank_ms = nr_ms / max(nr_launches, 1)
        nr_per_rank = [nr_avg_rank_ms]
        if dist is not None:                                  # the dominant kernel's average duration on every rank; the roofline uses the slowest
            t = torch.zeros(world, dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            t[rank] = nr_avg_rank_ms
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            nr_per_rank = [float(x) for x in t.tolist()]
        return dict(env=env, blocks=blocks, resets=resets_in_region, stats=stats, nr_per_rank=nr_per_rank, nr_launches=nr_launches,
                    returns=last_returns[0])

    B = a.envs
    m = measure(a.case, B, a.min_seconds, inner=a.inner)
    if a.inner:
        return
    env, blocks, resets_in_region, stats = m["env"], m["blocks"], m["resets"], m["stats"]
    nr_per_rank, nr_launches = m["nr_per_rank"], m["nr_launches"]
    if a.dump_returns and rank == 0 and m["returns"] is not None:
        np.save(a.dump_returns, m["returns"].detach().cpu().numpy())
    blocks_sorted = sorted(blocks)
    dt = blocks_sorted[len(blocks) // 2]
    kname = "k_nr_tree"
    head = dict(n_bus=env.n_bus, n_agents=env.n_agents, obs_size=env.obs_size, bytes_step=algorithmic_bytes_per_env_step(env),
                solver_bytes=solver_bytes_per_env_step(env))
    env.close()

    # ---- the other shapes north_star names, in the same run and under the same clock (short: 0.2 s of timed blocks each):
    # case33 x 4096 (BASELINE configs[1]) and the per-GPU shards of the two case322 configurations (8192 / 8, 65536 / 8)
    shapes = []
    if not a.no_other_shapes and a.case == "case141" and B == 4096:
        for c2, b2 in (("case33", 4096), ("case322", 1024), ("case322", 8192), ("case141_deep", 4096)):
            m2 = measure(c2, b2, 0.2)
            e2 = m2["env"]
            bl = sorted(m2["blocks"])
            d2 = bl[len(bl) // 2]
            by2 = algorithmic_bytes_per_env_step(e2)
            nr2 = max(m2["nr_per_rank"]) * 1e-3
            tr2, trsrc2 = committed_traffic(c2, b2)
            shapes.append({"workload": f"{c2} ({e2.n_bus}-bus, {e2.n_agents} agents), {b2} envs per GPU, step()+get_obs()"
                                       + (" — a depth stress (45-bus trunk, radius 25), not one of the reference's scenarios" if c2 == "case141_deep" else ""),
                           "value": world * b2 * a.steps / d2, "unit": "env-steps/s", "ms_per_step": d2 / a.steps * 1e3, "repeats": len(bl),
                           "ms_per_step_repeats": {"min": bl[0] / a.steps * 1e3, "max": bl[-1] / a.steps * 1e3},
                           "nr_iterations": {"mean": m2["stats"]["mean_nr_iters"], "max": m2["stats"]["max_nr_iters"]},
                           "roofline": {"bound": "hbm", "achieved": by2 * b2 / nr2 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": by2 * b2 / nr2 / 1e9 / HBM_PEAK_GBS, "kernel_avg_ms": nr2 * 1e3,
                                        "frac_kernel_bytes": solver_bytes_per_env_step(e2) * b2 / nr2 / 1e9 / HBM_PEAK_GBS,
                                        "frac_step": by2 * b2 / (d2 / a.steps) / 1e9 / HBM_PEAK_GBS,
                                        "kernel_launches_timed": m2["nr_launches"],
                                        "algorithmic_bytes_per_env_step": by2, "traffic": tr2,
                                        "traffic_source": (f"committed rocprofv3 PMC passes: {trsrc2}" if trsrc2 else None)}})
            e2.close()

    # ---- BASELINE configs[4] (end-to-end MADDPG rollout + update with GPU-resident replay), its per-GPU shard: the loop the env feeds.
    # Default workload at N = 1 only; reported beside the headline, never part of `value`.  A failure here does not cost the line.
    e2e = None
    if world == 1 and not a.no_e2e and not a.no_other_shapes and a.case == "case141" and B == 4096:
        try:
            from mapdn_amd import e2e as _e2e
            torch.cuda.empty_cache()
            t_e2e = time.perf_counter()
            lines = _e2e.run(case="case322", envs=8192, alg="maddpg", episodes=3, intensity="reference", phases=True, device=dev)
            last = lines[-1]
            e2e = {"workload": "case322 (322-bus, 38 agents) x 8192 envs per GPU, MADDPG (shared recurrent agent + MLP critic), GPU-resident replay, "
                               "the reference's update intensity (models/model.py:39-52: 10 value + 1 policy update per 60 env-steps, batch = a window "
                               "of 32 steps of every env = 5.87 sampled transitions per env-step); 240-step episodes",
                   "value": last["env_steps_per_s"], "unit": "env-steps/s (training loop, per GPU)", "episodes": len(lines),
                   "env_steps_per_s_per_episode": [ln["env_steps_per_s"] for ln in lines],      # the first episode carries allocations / autotuning
                   "seconds_per_episode": [ln["seconds"] for ln in lines], "phase_share": last.get("phase_share"),
                   "phase_seconds": last.get("phase_seconds"), "sampled_transitions_per_env_step": last["sampled_transitions_per_env_step"],
                   "hbm_gb": last["hbm_gb"], "wall_seconds_total": time.perf_counter() - t_e2e,
                   "env_share_note": "rollout_and_host is where step()+get_obs()+policy forward live: the env is a single-digit share of this loop"}
            # the MFMA-bound kernels of that loop, timed on their own at the loop's size (32 steps x 8192 envs x 38 agents = 9.96 M rows of 64):
            # the critic head (csrc/critic.hip) forward and backward against the dense f32 MFMA peak (157.3 TFLOP/s)
            try:
                from mapdn_amd.learner import MLPCritic, critic_head, make_alg_args
                nb_, n_ = 32 * 8192, 38
                cr = MLPCritic(7, 1, make_alg_args(3, 5, 1)).to(dev)
                base = torch.randn(nb_, 64, device=dev, requires_grad=True); pern = torch.randn(n_, 64, device=dev, requires_grad=True)
                dv = torch.randn(nb_ * n_, 1, device=dev)

                def _ms(f, reps=5):
                    for _ in range(2):
                        f()
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record()
                    for _ in range(reps):
                        f()
                    b_.record(); torch.cuda.synchronize(dev)
                    return a_.elapsed_time(b_) / reps

                def _f():
                    with torch.no_grad():
                        critic_head(cr, base, pern)

                def _fb():
                    critic_head(cr, base, pern).backward(dv)
                tf, tfb = _ms(_f), _ms(_fb)
                gf = nb_ * n_ * 64 * 64 * 2 / 1e9
                e2e["critic_head"] = {"rows": nb_ * n_, "forward_ms": tf, "backward_ms": tfb - tf, "peak_tflops_f32_mfma": 157.3,
                                      "forward": {"products_per_row": 1, "achieved_tflops": gf / tf, "frac": gf / tf / 157.3},
                                      "backward": {"products_per_row": 3, "achieved_tflops": 3 * gf / (tfb - tf), "frac": 3 * gf / (tfb - tf) / 157.3},
                                      "note": "v_mfma_f32_16x16x4_f32; the backward recomputes the forward product (3 of 64 x 64 per row); counters: "
                                              "profiles/r06_final_critic_head_mfma_counters.txt"}
                del base, pern, dv, cr
            except Exception as exc:      # noqa: BLE001
                e2e["critic_head"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        except Exception as exc:          # noqa: BLE001 — the headline line must still be printed
            e2e = {"error": f"{type(exc).__name__}: {exc}"[:500]}
        finally:
            torch.cuda.empty_cache()

    if rank == 0:
        n_gpus = world
        value = n_gpus * B * a.steps / dt
        bytes_step = head["bytes_step"]
        nr_avg_s = max(nr_per_rank) * 1e-3
        achieved = bytes_step * B / nr_avg_s / 1e9
        traffic, tsrc, tdetail = None, None, None
        if world == 1 and not a.no_traffic:
            tdetail = measure_traffic(a.case, B)
            if tdetail and tdetail.get("bytes"):
                traffic, tsrc = tdetail["bytes"], tdetail["source"]
        if traffic is None:
            traffic, tsrc = committed_traffic(a.case, B)
            if tsrc:
                tsrc = f"committed rocprofv3 PMC passes: {tsrc}"
        flops_step = 184.0 * head["n_bus"] * (stats["mean_nr_iters"] + 1.0)
        tfl = flops_step * B / nr_avg_s / 1e12
        out = {
            "metric": "env-steps/sec (whole node), case141 batch=4096, at 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "repeats": len(blocks), "timed_seconds_total": sum(blocks),
            "ms_per_step_repeats": {"min": blocks_sorted[0] / a.steps * 1e3, "median": dt / a.steps * 1e3,
                                    "max": blocks_sorted[-1] / a.steps * 1e3},
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.case} ({head['n_bus']}-bus, {head['n_agents']} agents), {B} parallel envs per GPU, "
                                   f"bowl voltage barrier, step()+get_obs(), 240-step episodes; "
                                   + (f"{resets_in_region} whole-batch reset(s) fell inside the {len(blocks)} timed block(s) "
                                      f"(their power flows are extra work, not counted as env-steps)" if resets_in_region else
                                      f"no episode boundary fell inside the {len(blocks)} timed block(s) of {a.steps} steps"),
                       "resets_in_timed_region": resets_in_region,
                       "envs_per_gpu": B, "global_envs": n_gpus * B, "obs_size": head["obs_size"],
                       "parallelism": f"env-batch sharded x{n_gpus}, no data-path collective; one all_gather of episode "
                                      f"returns ({a.backend}) at the end of the rollout, inside the timed region"
                                      + ("" if dist is not None else " (N = 1: torch.distributed not initialised; --force-dist runs the same collectives on one rank)")},
            "nr_iterations": {"mean": stats["mean_nr_iters"], "max": stats["max_nr_iters"]},
            # `bound`: what the SQ counters say limits the kernel (profiles/*_nr_sq_counters.txt) — per-row latency of the
            # tree sweeps, neither roof.  The fraction is against the HBM roof as the contract prescribes (SURVEY 8(d):
            # compulsory traffic is tiny by construction); the f64 fraction is in `compute`.
            "roofline": {"bound": "hbm", "limited_by": "latency (rows x sweeps of the tree elimination; neither roof)", "kernel": kname,
                         "kernel_scope": "one k_nr_tree launch = PV-bus injection (prologue, since round 4: it was a launch of its own, "
                                         "6.8 us, before) + Newton-Raphson solve + reward / res_line / sgen commit epilogue",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # the same roof read two other ways: `frac` divides the WHOLE step's bytes (obs
                         # output included) by the solver launch's time; frac_kernel_bytes = the solver launch's own bytes / its time;
                         # frac_step = the whole step's bytes / the whole step's time (ms_per_step: k_nr_tree + k_advance + k_gather)
                         "frac_kernel_bytes": head["solver_bytes"] * B / nr_avg_s / 1e9 / HBM_PEAK_GBS,
                         "frac_step": bytes_step * B / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                         "kernel_bytes_per_env_step": head["solver_bytes"],
                         "traffic": traffic, "traffic_source": tsrc,
                         "traffic_detail": tdetail,
                         "algorithmic_bytes_per_launch": bytes_step * B,
                         "algorithmic_bytes_per_env_step": bytes_step, "envs_per_launch": B,
                         "kernel_avg_ms": nr_avg_s * 1e3, "kernel_avg_ms_per_rank": nr_per_rank, "kernel_launches_timed": nr_launches,
                         "kernel_timing": "HIP events on the launch stream around every step's k_nr_tree launch (>= 240 launches; the reset's "
                                          "power flow excluded); the events add ~3 us of command-processor time to rocprofv3's kernel duration"},
            # compute-side view: ~184 * nb f64 flops per NR iteration (SpMV, mismatch, Jacobian,
            # block-tree solve, update) x (iterations + 1 mismatch evaluation), against the f64 vector peak
            "compute": {"algorithmic_flops_per_env_step": flops_step, "achieved_tflops": tfl, "peak_tflops": FP64_PEAK_TFLOPS,
                        "frac": tfl / FP64_PEAK_TFLOPS,
                        "note": "f64 vector (no MFMA on the radial path: the Jacobian is eliminated without fill, there is no "
                                "dense contraction)"},
        }
        if dist is not None:
            maps = open("/proc/self/maps").read()
            out["dist"] = {"backend": a.backend, "world_size": world, "forced_at_one_rank": bool(a.force_dist and world == 1),
                           "rccl_loaded": "librccl" in maps, "collectives_in_timed_region": ["all_gather_into_tensor(episode returns)",
                           "all_reduce(MAX, block time)", "barrier x2"], "gathered_rows_last_block": int(m["returns"].shape[0]) if m["returns"] is not None else None}
        if shapes:
            out["other_shapes"] = shapes
        if e2e is not None:
            out["e2e"] = e2e
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
