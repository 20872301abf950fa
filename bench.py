#!/usr/bin/env python
"""Headline benchmark: images/sec of the open-set ImageNet training step (ResNet-50 + entropic open-set loss) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]        any N: for N > 1 this process starts its own N ranks (one child process
                                                               per GPU; see launch_ranks) and relays rank 0's JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the same ranks started by torchrun)

One step = the reference's inner-loop body (openset_imagenet/train.py:125-139): zero_grad, forward, loss, backward
(+ bucketed RCCL gradient all-reduce when N > 1), optimizer step — on a synthetic batch that is already resident in HBM.
Workload (all N, weak scaling): Protocol 2 shapes — C = 30 known classes, entropic open-set loss, batch 128 per GPU,
3x224x224 fp32 images in [0,1), labels -1 with probability 0.5, Adam lr 1e-3, fp32 arithmetic throughout.

Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement for the fields).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "openset-imagenet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _cpu_list(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _cpus_near_gpu(index):
    """CPUs local to the index-th GPU of the KFD topology (sysfs only — no GPU call): the GPU nodes of
    /sys/class/kfd/kfd/topology/nodes in node order are HIP's device order when no visibility mask reorders them; the node's PCI
    address -> /sys/bus/pci/devices/<bdf>/local_cpulist. None when any step does not resolve."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = dict(line.split(None, 1) for line in open(os.path.join(base, node, "properties")) if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        p = gpus[index]
        loc = int(p["location_id"])
        bdf = "%04x:%02x:%02x.%x" % (int(p.get("domain", "0")), (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        return _cpu_list(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read())
    except (OSError, ValueError, KeyError, IndexError):
        return None


def plan_rank_cpus(n_ranks, policy="auto"):
    """CPU set of every rank: its share of the CPUs this process may use (os.sched_getaffinity). "near" / "auto": the CPUs local to
    the rank's GPU (sysfs), dealt out evenly among the ranks that share a locality domain; "even": contiguous equal chunks of the
    sorted mask; "none": no binding. "auto" falls back to "even" when the topology does not resolve for every rank or a visibility
    mask (HIP_/ROCR_VISIBLE_DEVICES) may have reordered the devices. Returns (list of CPU lists or None, policy used)."""
    if policy == "none" or not hasattr(os, "sched_getaffinity"):
        return None, "none"
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < n_ranks:
        return None, "none"
    if policy in ("auto", "near") and not (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
                                           or os.environ.get("CUDA_VISIBLE_DEVICES")):
        near = [_cpus_near_gpu(r) for r in range(n_ranks)]
        if all(near):
            domains = {}
            for r, cpus in enumerate(near):
                domains.setdefault(tuple(sorted(set(cpus) & set(mine))), []).append(r)
            if all(len(cpus) >= len(ranks) for cpus, ranks in domains.items()):
                out = [None] * n_ranks
                for cpus, ranks in domains.items():
                    per = len(cpus) // len(ranks)
                    for i, r in enumerate(ranks):
                        out[r] = list(cpus[i * per:(i + 1) * per])
                return out, "near"
    per = len(mine) // n_ranks
    return [mine[r * per:(r + 1) * per] for r in range(n_ranks)], "even"


MIN_CPU_SHARE = 4     # a rank is only bound when its share holds its launch thread, RCCL's proxy thread and the copy / logging helpers


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no rank variables in the environment: start the N ranks as CHILD processes of this
    one (one process per GPU, the reference's intent: config/train.yaml:18,35-39, script/train_all.py:96-99), relay rank 0's single
    JSON line, and exit non-zero — after terminating the survivors — if any rank fails or the run exceeds --launch-timeout.

    This parent never touches the GPU (torch is not even imported): a process that has initialised the GPU must not be replaced by
    another, and RCCL wants the devices untouched by anybody but their rank. Visibility masks are passed through unchanged (every
    rank sees every device and takes LOCAL_RANK), so the `rccl` object of the record carries real PCI ids."""
    import signal
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("OSI_BENCH_LAUNCH_TIMEOUT", "1500")))
    ap.add_argument("--bind", default=os.environ.get("OSI_BENCH_BIND", "auto"), choices=("auto", "near", "even", "none"))
    la, _ = ap.parse_known_args(argv)
    with socket.socket() as s:                   # a free rendezvous port (a constant collides with a neighbour's run)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cpus, policy = plan_rank_cpus(n, la.bind)
    if cpus and min(len(c) for c in cpus) < MIN_CPU_SHARE:
        cpus, policy = None, f"none (a share would be < {MIN_CPU_SHARE} CPUs)"
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OSI_BENCH_LAUNCHED="1", OSI_BENCH_BIND_POLICY=policy)
        if cpus:
            env["OSI_BENCH_CPUS"] = ",".join(map(str, cpus[rank]))
        # rank 0's stdout is the record; the other ranks print nothing there (anything they do print goes to our stderr)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr.fileno()))
    print(f"[bench] started {n} ranks, pids {[p.pid for p in procs]}, rendezvous 127.0.0.1:{port}, cpu binding {policy}", file=sys.stderr, flush=True)

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)        # exactly the process groups started here
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        stop_all()
        raise SystemExit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, on_signal)

    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + la.launch_timeout
    failed = None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                failed = f"no result after {la.launch_timeout:.0f} s (--launch-timeout)"
                break
            time.sleep(0.1)
    finally:
        if failed or any(p.poll() is None for p in procs):
            stop_all(signal.SIGTERM)
            t_kill = time.monotonic() + 10
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(0.1)
            stop_all(signal.SIGKILL)
            for p in procs:
                p.wait()
    reader.join(timeout=5)
    if failed:
        print(f"bench.py: {failed}; the other ranks were terminated", file=sys.stderr, flush=True)
        return 1
    lines = [ln for ln in (out0[0] if out0 else b"").decode(errors="replace").splitlines() if ln.strip()]
    if len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} lines, expected exactly one JSON line", file=sys.stderr, flush=True)
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


def _gpus_arg(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and "-h" not in sys.argv \
        and "--help" not in sys.argv and _gpus_arg(sys.argv[1:]) > 1:
    sys.exit(launch_ranks(_gpus_arg(sys.argv[1:]), sys.argv[1:]))      # BEFORE torch is imported: the launcher makes no GPU call

# a rank started by launch_ranks binds itself to its CPU share before torch (and its thread pools) load
BOUND_CPUS = None
if os.environ.get("OSI_BENCH_CPUS") and hasattr(os, "sched_setaffinity"):
    try:
        os.sched_setaffinity(0, _cpu_list(os.environ["OSI_BENCH_CPUS"]))
        BOUND_CPUS = sorted(os.sched_getaffinity(0))
    except OSError:
        pass

# a rank started by somebody else's launcher (`python -m torch.distributed.run ... bench.py --gpus N`, the driver's form) takes the share
# the self-launcher would have given it: same plan, computed by every rank from the mask they all inherited (--bind none / OSI_BENCH_BIND=none: off)
elif int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("LOCAL_RANK") is not None and hasattr(os, "sched_setaffinity") \
        and os.environ.get("OSI_BENCH_LAUNCHED") != "1":         # (the self-launcher has decided for its own children)
    try:
        _bind = sys.argv[sys.argv.index("--bind") + 1] if "--bind" in sys.argv[:-1] else os.environ.get("OSI_BENCH_BIND", "auto")
        _lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"]))
        _plan, _policy = plan_rank_cpus(_lw, _bind if _bind in ("auto", "near", "even", "none") else "auto")
        if _plan is not None and min(len(c) for c in _plan) >= MIN_CPU_SHARE:
            os.sched_setaffinity(0, _plan[int(os.environ["LOCAL_RANK"])])
            BOUND_CPUS = sorted(os.sched_getaffinity(0))
            os.environ.setdefault("OSI_BENCH_BIND_POLICY", "rank-side:" + _policy)
    except (OSError, ValueError, IndexError):
        pass

# RCCL has no API for its channel count (= the workgroups a collective keeps resident on this GPU): every rank reads it from the
# communicator's own INIT log, written to a private file (stdout stays one JSON line). Set before torch loads RCCL.
RCCL_LOG = None
if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--force-dp" in sys.argv) \
        and os.environ.get("OSI_BENCH_BACKEND", "nccl") == "nccl" and "NCCL_DEBUG" not in os.environ:
    RCCL_LOG = f"/tmp/osi_rccl_init_{os.getpid()}.log"
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_FILE=RCCL_LOG)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work of the conv stack per image: fwd 8.1743, dgrad 7.9383 (no stem dgrad), wgrad 8.1743
CONV_GFLOP_FWD, CONV_GFLOP_DGRAD, CONV_GFLOP_WGRAD = 8.1743, 7.9383, 8.1743
CONV_GFLOP_PER_IMAGE = 24.287
MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"

WORKLOADS = {
    "p2": dict(name="Protocol 2, ResNet-50, entropic open-set loss, batch 128/GPU", C=30, B=128, loss="entropic", p_neg=0.5),
    "p1": dict(name="Protocol 1, ResNet-50, entropic open-set loss, batch 128/GPU", C=116, B=128, loss="entropic", p_neg=0.37),
    "p3": dict(name="Protocol 3, ResNet-50, background-class softmax, batch 256/GPU", C=152, B=256, loss="garbage", p_neg=0.39),
}


def synthetic_batch(B, C, p_neg, loss, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    labels = torch.randint(0, C if loss != "garbage" else C - 1, (B,), generator=g)
    neg = torch.rand(B, generator=g) < p_neg
    labels[neg] = -1 if loss != "garbage" else C - 1
    gd = torch.Generator(device=device).manual_seed(seed)
    images = torch.rand(B, 3, 224, 224, device=device, generator=gd)
    return images, labels.to(device)


def _cpu_leg(C, p_neg, loss, steps, B):
    """One CPU-oracle leg: fwd + loss + bwd + Adam at batch B; returns (best step seconds, images/sec)."""
    from oracle import resnet50_oracle as R, losses_oracle as L
    torch.manual_seed(42)
    sd = R.init_state(C, C, False)
    g = torch.Generator().manual_seed(42)
    x = torch.rand(B, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    if loss == "entropic":
        y[torch.rand(B, generator=g) < p_neg] = -1
        fn = lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0)
    else:   # plain softmax cross-entropy on known classes only (negatives are removed from the training set, train.py:291-293)
        fn = lambda lg, t, f: L.softmax_loss(lg, t)
    state, times = {}, []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        _, _, _, grads = R.forward_backward(sd, x, y, fn)
        R.adam_step(sd, grads, state, lr=1e-3)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu baseline ({loss}, C={C}) step {i}/{steps}: {times[-1]:.2f} s on {torch.get_num_threads()} threads", file=sys.stderr, flush=True)
        if times[-1] > 60 and i >= 1:      # a host this slow gets one timed step: the default run must end within minutes
            break
    best = min(times[1:])
    return best, B / best


def cpu_model():
    """CPU model string of the host (SURVEY.md §8d asks for core count AND model)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def usable_cpus(gpus_used=1, cpus_per_gpu=16):
    """How many CPUs' worth of time this process can get, and every limit that went into it. `cpus_per_gpu` is the lease policy of
    the pool (--cpus-per-gpu / OSI_CPUS_PER_GPU, 16 on this one), multiplied by the GPUs THIS RUN uses — not by the devices that
    happen to be visible."""
    n = os.cpu_count() or 1
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else n
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = max(1, -(-int(txt[0]) // int(txt[1])))
            elif int(txt[0]) > 0:
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = max(1, -(-int(txt[0]) // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    share = cpus_per_gpu * max(1, gpus_used)
    return {"threads": max(1, min(n, aff, quota or n, share)), "host_cpu_count": n, "affinity_cpus": aff, "cgroup_cpu_quota": quota,
            "cpus_per_gpu": cpus_per_gpu, "gpus_used": gpus_used, "lease_share": share}


def cpu_baseline(C, p_neg, steps=5, B=32, gpus_used=1, cpus_per_gpu=16):
    """The CPU oracle (torch-CPU restatement of the reference path) timed on this host's cores, the two legs SURVEY.md §8(d) names:
    the GPU workload's own loss at B = 32 (`value`) and BASELINE.json config 1 (Protocol 1, C = 116, softmax cross-entropy, B = 32)."""
    # SURVEY.md §8(d) says torch.set_num_threads(os.cpu_count()). On a GPU box that is wrong by an order of magnitude: the host has 256
    # logical CPUs but a one-GPU lease owns a share of them (16 per GPU on this pool), and 128-256 compute threads on 16 CPUs' worth of
    # time make every oneDNN primitive crawl (round 2: 10.2 s per step with torch's default 128 threads, against 3.7-5.3 s on 8 cores
    # in the build container; 256 threads did not finish a step in 7 minutes). The thread count is therefore the CPU time this process
    # can actually get: min(os.cpu_count(), affinity mask, cgroup CPU quota, --cpus-per-gpu x the GPUs this run uses).
    default_threads = torch.get_num_threads()
    limits = usable_cpus(gpus_used, cpus_per_gpu)
    want = limits["threads"]
    torch.set_num_threads(want)
    try:
        best, ips = _cpu_leg(C, p_neg, "entropic", steps, B)
        best1, ips1 = _cpu_leg(116, 0.0, "softmax", 3, B)
        used = torch.get_num_threads()
    finally:
        torch.set_num_threads(default_threads)
    return {"value": round(ips, 3), "unit": "images/sec", "cores": used, "cpu_model": cpu_model(), "kind": "port",
            "threads": {"used": used, "set_num_threads_applied": True, "set_to_os_cpu_count": want == (os.cpu_count() or 0),
                        "torch_default": default_threads, **limits,
                        "rule": "torch.set_num_threads(min(os.cpu_count(), affinity mask, cgroup CPU quota, cpus_per_gpu x gpus_used)): the CPUs "
                                "this process can actually get; SURVEY.md §8(d)'s plain os.cpu_count() oversubscribes a one-GPU lease 16x"},
            "sample": f"best of {steps} timed steps (after 1 warm-up) of batch {B}, same shapes/loss as the GPU workload, best step {best:.2f} s; "
                      f"oracle/resnet50_oracle.py (torch-CPU fp32 restatement; the reference package itself is not importable offline)",
            "config1_protocol1_softmax_b32": {"value": round(ips1, 3), "unit": "images/sec",
                                              "sample": f"best of 3 timed steps (after 1 warm-up) of batch {B}, C = 116, softmax "
                                                        f"cross-entropy, Adam; step {best1:.2f} s (BASELINE.json configs[0])"},
            "host_cpu_count": os.cpu_count()}


def _git_blob_id(path):
    """sha1 of the file as `git hash-object` computes it (no git needed on the GPU box): ties a number to the committed profile."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def committed_traffic(ms_step, B, workload):
    """HBM-side bytes per step of the conv kernels from the newest committed PMC summary (profiles/rNN_hbm_traffic_per_step.json),
    with its provenance: file, git blob id, the step time / workload the profile was taken at, and whether that still matches
    the run being reported (a stale or foreign profile yields traffic = None instead of a silently re-reported number)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_per_step.json")))
    if not files:
        return None, {"file": None}
    path = files[-1]
    t = json.load(open(path))
    meta = t.get("_meta", {})
    prov = {"file": os.path.relpath(path, ROOT), "git_blob": _git_blob_id(path), "profiled_ms_per_step": meta.get("ms_per_step"),
            "profiled_workload": meta.get("workload"), "profiled_commit": meta.get("commit")}
    same = meta.get("workload") == workload and meta.get("batch") == B
    fresh = same and meta.get("ms_per_step") and abs(meta["ms_per_step"] - ms_step) <= 0.15 * ms_step
    prov["matches_this_run"] = bool(fresh)
    if not fresh:
        return None, prov
    total = sum(t[k]["fetch_GB_x2_wide_read_correction"] + t[k]["write_GB"] for k in ("conv_fwd", "conv_dgrad", "conv_wgrad"))
    return round(total * 1e9), prov


def committed_mfma_util(ms_step, B, workload):
    """Matrix-pipe busy share per conv class (and the measured issued FLOPs) from the newest committed counter summary
    (profiles/rNN_mfma_util.json), under the provenance rule of committed_traffic: a profile of another workload, or one whose step time
    is more than 15 % off this run, is named but not reported."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util.json")))
    if not files:
        return None, None, {"file": None}
    path = files[-1]
    t = json.load(open(path))
    meta = t.get("_meta") or {}
    prov = {"file": os.path.relpath(path, ROOT), "git_blob": _git_blob_id(path), "profiled_ms_per_step": meta.get("ms_per_step"),
            "profiled_workload": meta.get("workload"), "profiled_commit": meta.get("commit")}
    fresh = meta.get("workload") == workload and meta.get("batch") == B and meta.get("ms_per_step") \
        and abs(meta["ms_per_step"] - ms_step) <= 0.15 * ms_step
    prov["matches_this_run"] = bool(fresh)
    if not fresh:
        return None, None, prov
    busy = {k: t["mfma_busy_percent"].get(k) for k in ("conv_fwd", "conv_dgrad", "conv_wgrad")}
    return busy, t.get("issued_gflop_per_step_measured"), prov


# the thirteen 3x3 / stride 1 / pad 1 layers of ResNet-50 (conv2 of every bottleneck without a stride): (channels, H, count)
WINO_LAYERS = ((64, 56, 3), (128, 28, 3), (256, 14, 5), (512, 7, 2))


def issued_conv_gflop(B, lib):
    """Matrix FLOPs the conv kernels ISSUE per step, per class. The direct kernels issue the convolution's own multiplies (SURVEY.md section 8d:
    what `roofline.achieved` counts); a Winograd layer issues 16 transform-domain products per 2x2 output tile and channel pair instead of
    36 — 4/9, times the tile padding where the image is not a whole number of tiles (7 x 7: 16 tiles cover 64 pixel slots for 49 pixels).
    Which directions run Winograd is read from the library's knobs."""
    knob = ctypes.c_int()
    out = {}
    for cls, gf, name in (("conv_fwd", CONV_GFLOP_FWD, b"fwd_wino"), ("conv_dgrad", CONV_GFLOP_DGRAD, b"dgrad_wino"), ("conv_wgrad", CONV_GFLOP_WGRAD, b"wgrad_wino")):
        total = gf * B
        lib.osi_get_tuning(name, ctypes.byref(knob))
        if knob.value:
            for C, H, count in WINO_LAYERS:
                direct = 2.0 * H * H * C * C * 9 * count * B / 1e9
                tiles = ((H + 1) // 2) ** 2
                total += 2.0 * tiles * 16 * C * C * count * B / 1e9 - direct
        out[cls] = total
    return out


def parity_probe(model, C, device):
    """Same-run logits parity on one shared batch with shared (freshly initialised) weights: HIP path vs the CPU oracle in fp32
    and in fp64 (the arbiter: torch-CPU fp32 is itself a few 1e-5 away from fp64 on this network)."""
    from oracle import resnet50_oracle as R
    g = torch.Generator().manual_seed(7)
    x = torch.rand(8, 3, 224, 224, generator=g)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.train()
    with torch.no_grad():
        lg, _ = model(x.to(device))
    lg = lg.cpu()
    ref32, _ = R.forward({k: v.clone() for k, v in sd.items()}, x, True)
    ref64, _ = R.forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(), True)
    model.load_state_dict({k: v.to(device) for k, v in sd.items()})  # undo the probe's running-stat update
    return {"max_abs_logit_err_vs_cpu_oracle_fp32": float((lg - ref32).abs().max()),
            "max_abs_logit_err_vs_cpu_oracle_fp64": float((lg.double() - ref64).abs().max()),
            "cpu_fp32_vs_fp64": float((ref32.double() - ref64).abs().max()), "tolerance": 1e-4,
            "probe": "train-mode forward, batch 8 at 224x224, shared weights at initialisation"}


def rccl_proof(model, net, backend, world, local, dev):
    """What lets a SCALE record PROVE that N distinct GPUs took part: every rank's device identity gathered over the process
    group itself, the communicator's backend / version and the per-step payload. Two ranks on one device under RCCL is an error
    (reference intent: one process per GPU, config/train.yaml:18,35-39)."""
    props = torch.cuda.get_device_properties(dev)
    ident = {"rank": int(os.environ.get("RANK", "0")), "local_rank": local, "device_index": dev.index,
             "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
             "pci_device_id": getattr(props, "pci_device_id", None), "name": props.name, "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, ident)
    # Two ranks provably share a device when an INFORMATIVE identifier coincides: the PCI bus id, or a uuid that is not all zeros. The
    # device index alone proves nothing either way (a launcher may give every rank its own visibility mask, all index 0), so it only
    # counts when nothing better exists. A shared device under RCCL is an error; ambiguous evidence is reported, never fatal.
    def informative(d):
        if d["pci_bus_id"] is not None:
            return "pci_bus_id", (d["pci_bus_id"], d["pci_device_id"])
        if d["uuid"] and set(d["uuid"]) - set("0-"):
            return "uuid", d["uuid"]
        return "device_index", d["device_index"]
    kinds = {informative(d)[0] for d in everyone}
    evidence = kinds.pop() if len(kinds) == 1 else "mixed"
    distinct = len({informative(d) for d in everyone})
    used = dist.get_backend()
    if backend == "nccl":
        assert used == "nccl", f"bench.py measures RCCL: the process group reports backend {used!r}"
        if distinct != world and evidence in ("pci_bus_id", "uuid"):
            raise SystemExit(f"bench.py: {world} ranks but only {distinct} distinct GPUs by {evidence} "
                             f"({[informative(d)[1] for d in everyone]}): one process per GPU")
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        ver = None
    buckets = model.gradient_buckets()
    return {"world": world, "effective_world_for_averaging": net.sync.world, "backend": used, "distinct_devices": distinct,
            "distinctness_evidence": evidence,
            "devices": [{k: d[k] for k in ("rank", "device_index", "uuid", "pci_bus_id", "name")} for d in everyone],
            "nccl_version": ver, "allreduce_bytes_per_step": int(sum(4 * (hi - lo) for lo, hi in buckets)),
            "buckets": [{"floats": int(hi - lo), "bytes": int(4 * (hi - lo))} for lo, hi in buckets],
            "collective": "all_reduce(AVG) per backward stage on a contiguous slice of the gradient arena, async beside the remaining backward"}


def rccl_channels(path):
    """Channel count of the RCCL communicator from its INIT log ("N coll channels", "Channel 00/N"), plus the environment overrides
    that would pin it. None when the log does not say."""
    import re
    info = {"NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"), "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
            "coll_channels": None, "source": None}
    import glob
    txt = ""
    for f in (sorted(glob.glob(path + "*")) if path else []):      # RCCL may append host / pid suffixes to NCCL_DEBUG_FILE
        try:
            txt += open(f, errors="replace").read()
        except OSError:
            pass
    m = re.findall(r"(\d+) coll channels", txt)
    if m:
        info.update(coll_channels=int(m[-1]), source='"N coll channels" line of the NCCL_DEBUG=INFO init log')
    else:
        m = re.findall(r"Channel \d+/(\d+)", txt)
        if m:
            info.update(coll_channels=int(m[-1]), source='"Channel xx/N" lines of the NCCL_DEBUG=INFO init log')
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each; the median window is reported")
    ap.add_argument("--workload", default="p2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (invalidates the headline config)")
    ap.add_argument("--cpus-per-gpu", type=int, default=int(os.environ.get("OSI_CPUS_PER_GPU", "16")),
                    help="CPU share of one GPU lease on this pool: caps the CPU baseline's thread count (x the GPUs this run uses)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event instrumentation of the executor")
    ap.add_argument("--force-dp", action="store_true", help="dev: with one rank, still run the data-parallel step (staged backward + "
                    "RCCL bucket all-reduce on a world-1 communicator) to price the N>1 code path on a one-GPU box")
    ap.add_argument("--sustained-steps", type=int, default=None,
                    help="further steps AFTER the timed windows (never part of `value`), reported as `sustained` in 50-step windows; 0 = off. "
                         "Default 600 (OSI_BENCH_SUSTAINED), and 0 for dev runs (--no-profile or --windows 1: A/B and counter passes)")
    ap.add_argument("--eval-steps", type=int, default=None, help="inference leg after everything else (never part of `value`): eval-mode forwards "
                    "timed per mode; 0 = off. Default 20, and 0 for dev runs (--no-profile or --windows 1)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="N > 1 self-launch: seconds before the ranks are terminated")
    ap.add_argument("--bind", default="auto", choices=("auto", "near", "even", "none"), help="N > 1 self-launch: CPU binding of the ranks")
    args = ap.parse_args()
    if args.sustained_steps is None:
        args.sustained_steps = 0 if (args.no_profile or args.windows == 1) else int(os.environ.get("OSI_BENCH_SUSTAINED", "600"))
    if args.eval_steps is None:
        args.eval_steps = 0 if (args.no_profile or args.windows == 1) else 20
    if args.windows < 1 or args.steps < 1 or args.warmup < 0 or args.cpus_per_gpu < 1 or args.sustained_steps < 0:
        ap.error("--windows, --steps and --cpus-per-gpu must be >= 1, --warmup and --sustained-steps >= 0")

    # stdout carries exactly ONE line (the JSON); anything libraries print on fd 1 (RCCL's version banner, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # fault injection for the launcher's tests (tests/test_bench_launcher.py): "RANK:exit" / "RANK:sleep", before anything touches the GPU
    for fault in filter(None, os.environ.get("OSI_BENCH_FAULT", "").split(",")):
        r, _, action = fault.partition(":")
        if int(r) == rank:
            if action == "exit":
                sys.exit(3)
            time.sleep(3600)
    if args.gpus != world:
        # `python bench.py --gpus N` starts its own ranks (launch_ranks, top of this file) and torch.distributed.run sets WORLD_SIZE = N;
        # anything else (a stray RANK without WORLD_SIZE, main() imported and called by hand) is a mis-launch
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: run `python bench.py --gpus N` with no rank variables in the "
                 "environment (it starts one child process per GPU) or launch the N ranks with torch.distributed.run")
    from openset_imagenet import ResNet50, EntropicOpensetLoss, GarbageLoss, optim, tools, _native as N
    from openset_imagenet.dp import DistributedDataParallel
    N.lib()  # fail loudly if the HIP library is missing
    # dev rehearsal of N > 1 on a box with fewer GPUs than ranks: OSI_BENCH_BACKEND=gloo lets several ranks share a device
    # (RCCL refuses that); the measured number is then meaningless, only the multi-rank control flow is exercised
    backend = os.environ.get("OSI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    dev = tools.set_device_gpu(local)
    use_dp = world > 1 or args.force_dp
    rccl_log = RCCL_LOG
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                sys.exit("bench.py: WORLD_SIZE > 1 without MASTER_PORT — the launcher (bench.py itself, or torch.distributed.run) sets it")
            import socket
            with socket.socket() as sk:      # --force-dp at world 1: any free port (a constant collides with a neighbour's run)
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    B, C = wl["B"], wl["C"]
    torch.manual_seed(42)
    model = tools.device(ResNet50(C, C, False))
    net = DistributedDataParallel(model) if use_dp else model
    if args.force_dp and world == 1:
        net.sync.world = 2   # AVG over one rank is the identity; the collectives are still enqueued
        if os.environ.get("OSI_BENCH_SKIP_COLLECTIVE") == "1":   # dev: price the staged backward + hand-off WITHOUT RCCL's kernels
            dist.all_reduce = lambda *a, **k: None
        elif os.environ.get("OSI_BENCH_SKIP_COLLECTIVE") == "2": # dev: the four staged backward calls alone (no hand-off either)
            net.sync.world = 1
    # Under RCCL the collectives keep `channels` workgroups resident beside the backward. The balanced-remainder plan of the forward /
    # input-gradient launches is told (knob dp_reserved_cus) unless the knob was set by hand; the communicator exists by now (the
    # wrapper's broadcast created it) and its INIT log names the channel count. MAX over the ranks: one plan everywhere.
    reserved = None
    if use_dp and backend == "nccl":
        from openset_imagenet.dp import reserved_cus_for_channels
        ch = rccl_channels(rccl_log)
        if os.environ.get("OSI_DP_RESERVED_CUS"):
            reserved = {"value": int(os.environ["OSI_DP_RESERVED_CUS"]), "source": "OSI_DP_RESERVED_CUS"}
        elif world > 1:
            # EVERY rank enters this collective, whatever its own log said (each rank parses its own INIT log; the "Channel xx/N" lines
            # are rank 0's only and a log may not be flushed yet): a rank that knows nothing contributes -1, MAX over the ranks decides,
            # and the knob is only touched when somebody knew. A collective some ranks skip would hang the ones that follow.
            mine_cus = -1 if ch["coll_channels"] is None else reserved_cus_for_channels(ch["coll_channels"])
            v = torch.tensor([mine_cus], device=dev, dtype=torch.int32)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            if int(v) >= 0:
                N.check(N.lib().osi_set_tuning(b"dp_reserved_cus", int(v)))
                reserved = {"value": int(v), "source": "ceil(RCCL channels / 8 workgroup slots per CU), MAX over ranks "
                                                       f"(this rank's INIT log: {ch['coll_channels']} channels)"}
            else:
                reserved = {"value": 0, "source": "default (no rank's RCCL INIT log named a channel count)"}
        else:
            reserved = {"value": 0, "source": "default (world 1: nothing is resident beside the backward)"}
    opt = optim.Adam(model.parameters(), lr=1e-3)
    images, labels = synthetic_batch(B, C, wl["p_neg"], wl["loss"], dev, 42 + rank)
    if wl["loss"] == "garbage":
        # class weights from a dataset-level label histogram (dataset.py:77-86), not from one batch: every known class present
        # (as in the protocol CSVs), the negatives at the workload's share
        import numpy as np
        from openset_imagenet.dataset import LabelTable
        known = np.repeat(np.arange(C - 1), 100)
        table = LabelTable(np.concatenate([known, -np.ones(int(len(known) * wl["p_neg"] / (1 - wl["p_neg"])), dtype=np.int64)]))
        table.replace_negative_label()
        loss_fn = GarbageLoss(table.calculate_class_weights())
    else:
        loss_fn = EntropicOpensetLoss(C, 1.0)

    def step():
        model.train()
        opt.zero_grad()
        logits, _ = net(images)
        j = loss_fn(logits, labels)
        j.backward()
        opt.step()
        return j

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rccl = rccl_proof(model, net, backend, world, local, dev) if use_dp else None
    parity = parity_probe(model, C, dev) if (world == 1 and not args.no_cpu_baseline) else None
    for _ in range(args.warmup):
        step()
    handle = model._net(B, 224, 224).h
    # Three back-to-back windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and reduced with
    # MAX over the ranks; the MEDIAN window is the reported one (`value`, `ms_per_step`), all three are listed. One 0.7 s window
    # moves by +-1.5 % between identical runs; the median of three resolves ~1 % levers with the same `steps` per window.
    windows, own_windows = [], []
    for _ in range(args.windows):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        torch.cuda.synchronize()
        own_windows.append(time.perf_counter() - t0)    # this rank's own finish time (before the barrier): a straggler shows here
        fence()
        w = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([w], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t)
        windows.append(w)
    elapsed = sorted(windows)[len(windows) // 2]
    loss_value = float(last.detach())
    # Sustained leg (never part of `value`): the timed region above is a ~2 s burst; this is the same step for --sustained-steps more
    # steps, one host synchronisation per 50-step window, so that clock / thermal drift shows as a series instead of hiding in a burst.
    sustained = None
    if args.sustained_steps >= 50:
        nwin = args.sustained_steps // 50
        series = []
        fence()
        t_all = time.perf_counter()
        for _ in range(nwin):
            t0 = time.perf_counter()
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            series.append((time.perf_counter() - t0) / 50 * 1e3)
        fence()
        total = time.perf_counter() - t_all
        if world > 1:
            t = torch.tensor(series + [total], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            series, total = [float(v) for v in t[:-1]], float(t[-1])
        srt = sorted(series)
        sustained = {"steps": nwin * 50, "seconds": round(total, 3), "images_per_sec": round(world * B * nwin * 50 / total, 2),
                     "ms_per_step_mean": round(total / (nwin * 50) * 1e3, 3),
                     "ms_per_step_p50": round(srt[len(srt) // 2], 3), "ms_per_step_p95": round(srt[min(len(srt) - 1, int(0.95 * len(srt)))], 3),
                     "first_vs_last_window": round(series[0] / series[-1], 4),
                     "window_ms_per_step": [round(v, 3) for v in series],
                     "how": "50-step windows run back to back after the timed region, one synchronize per window (MAX over ranks per window)"}
    if use_dp:     # communication leg: the same step with every bucket's collective bracketed by events (outside the timed windows)
        net.sync.timing(True)
        for _ in range(max(1, min(args.steps, 5))):
            step()
        torch.cuda.synchronize()
        comm = net.sync.read_timing()
        net.sync.timing(False)
        # what every rank saw, gathered over the process group itself: a SCALE record must show a straggler or one rank's exposed
        # communication, not only the MAX that `value` is computed from
        mine = {"rank": rank, "cpus": BOUND_CPUS and f"{len(BOUND_CPUS)} CPUs {BOUND_CPUS[0]}..{BOUND_CPUS[-1]}",
                "own_ms_per_step_windows": [round(w / args.steps * 1e3, 3) for w in own_windows],
                "exposed_comm_ms": None if comm["exposed_comm_ms"] is None else round(comm["exposed_comm_ms"], 3),
                "comm_ms_per_step": None if comm["comm_ms_per_step"] is None else round(comm["comm_ms_per_step"], 3)}
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        med = [sorted(r["own_ms_per_step_windows"])[len(r["own_ms_per_step_windows"]) // 2] for r in per_rank]
        knob = ctypes.c_int()
        plan = {}
        for name in ("tail_cus", "dp_reserved_cus", "tail_split"):
            N.check(N.lib().osi_get_tuning(name.encode(), ctypes.byref(knob)))
            plan[name] = knob.value
        hw = torch.cuda.get_device_properties(dev).multi_processor_count
        plan.update(dp_reserved_cus_source=None if reserved is None else reserved["source"])
        plan.update(hw_cus=hw, tail_plan_cus_in_effect=plan["tail_cus"] or max(8, hw - plan["dp_reserved_cus"]),
                    note="CU count the balanced-remainder plan of the forward / input-gradient launches assumes; dp_reserved_cus (OSI_DP_RESERVED_CUS) "
                         "leaves wave slots to RCCL's resident channel workgroups (one 256-thread workgroup per channel = 1/8 of a CU's slots)")
        rccl.update(per_rank=per_rank, per_rank_ms_per_step={"min": min(med), "median": sorted(med)[len(med) // 2], "max": max(med)},
                    launch_plan=plan, channels=rccl_channels(rccl_log) if backend == "nccl" else None)
        rccl.update(comm_ms_per_step=None if comm["comm_ms_per_step"] is None else round(comm["comm_ms_per_step"], 3),
                    per_bucket_comm_ms=None if comm["per_bucket_ms"] is None else [round(v, 3) for v in comm["per_bucket_ms"]],
                    exposed_comm_ms=None if comm["exposed_comm_ms"] is None else round(comm["exposed_comm_ms"], 3),
                    comm_steps=comm["steps"],
                    comm_how="instrumented steps after the timed windows: each bucket's all_reduce is issued from the communication stream "
                             "between two HIP events, the first recorded behind the hand-off (osi_resnet50_grads_ready: the stage's main- "
                             "and side-stream producers), so a bucket's figure is the collective's own duration; exposed = how long the "
                             "compute stream waited for the communication stream in finish(), after the whole backward was enqueued")
    # Roofline leg: the same step, right after the timed region, with one HIP event after every executor op on the launch
    # stream. The instrumented mode keeps every kernel on that one stream (weight gradients are NOT moved to the side stream),
    # so each class's duration is its own; the headline `value` above comes from the un-instrumented, overlapped steps.
    prof = None
    psteps = 0 if args.no_profile else max(1, min(args.steps, 5))
    if psteps:
        N.check(N.lib().osi_resnet50_profile(handle, 1))
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        ms = (ctypes.c_double * 7)()
        cnt = (ctypes.c_int * 7)()
        N.check(N.lib().osi_resnet50_profile_read(handle, ms, cnt))
        N.check(N.lib().osi_resnet50_profile(handle, 0))
        names = ["start", "conv_fwd", "conv_dgrad", "conv_wgrad", "bn_fwd", "bn_bwd", "other"]
        prof = {n: {"ms_per_step": ms[i] / psteps, "launch_groups_per_step": cnt[i] / psteps} for i, n in enumerate(names) if i}

    # Inference leg (never part of `value`): the forward of validate() / get_arrays() (reference train.py:142-234: model.eval(), no_grad)
    # on the same resident batch — the training topology on running statistics (executor option eval_fused = 0: pre-BN tensors, 16
    # block-output passes, bitmasks, 53 coefficient launches) against the inference forms (default: every BatchNorm + shortcut + ReLU in
    # its convolution's epilogue), and one validation step (forward + loss + confidence sums) as validate() issues it.
    evalrec = None
    if args.eval_steps > 0:
        def timed_forward(n, extra=None):
            with torch.no_grad():
                for _ in range(3):
                    lg, ft = model(images)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    lg, ft = model(images)
                    if extra is not None:
                        extra(lg)
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        def class_ms():
            N.check(N.lib().osi_resnet50_profile(handle, 1))
            with torch.no_grad():
                for _ in range(3):
                    model(images)
            torch.cuda.synchronize()
            ms, cnt = (ctypes.c_double * 7)(), (ctypes.c_int * 7)()
            N.check(N.lib().osi_resnet50_profile_read(handle, ms, cnt))
            N.check(N.lib().osi_resnet50_profile(handle, 0))
            return {"conv_ms": round(ms[1] / 3, 3), "bn_ms": round(ms[4] / 3, 3), "other_ms": round(ms[6] / 3, 3),
                    "launch_groups": int(sum(cnt[1:]) / 3)}
        model.eval()
        evalrec = {}
        for name, fused in (("training_topology", 0), ("fused", 1)):
            N.check(N.lib().osi_resnet50_set_option(handle, b"eval_fused", fused))
            t = timed_forward(args.eval_steps)
            evalrec[name] = {"ms_per_batch": round(t * 1e3, 3), "images_per_sec": round(B / t, 1)}
            if not args.no_profile:
                evalrec[name]["serialized"] = class_ms()
        acc4 = torch.zeros(4, dtype=torch.float64, device=dev)
        def val_tail(lg):
            loss_fn(lg, labels)
            if wl["loss"] == "garbage":
                N.ops().confidence_accumulate(lg, labels, 0.0, C - 1, -1, acc4)
            else:
                N.ops().confidence_accumulate(lg, labels, 1.0 / C, -1, 0, acc4)
        t = timed_forward(args.eval_steps, val_tail)
        evalrec["validate_step"] = {"ms_per_batch": round(t * 1e3, 3), "images_per_sec": round(B / t, 1),
                                    "what": "fused forward + loss + confidence sums per batch, as train.validate() issues them (no host sync inside)"}
        evalrec.update(batch=B, steps=args.eval_steps, speedup=round(evalrec["training_topology"]["ms_per_batch"] / evalrec["fused"]["ms_per_batch"], 3),
                       how=f"{args.eval_steps} eval-mode forwards of the resident batch after 3 warm-ups, one synchronize around the loop; per rank "
                           "(validation is sharded over the ranks by whole batches: the node's rate is world x this)")
        model.train()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        ips = world * B * args.steps / elapsed
        out = {
            "metric": "images/sec (whole node) ResNet-50 + entropic-open-set, Protocol 2, 1/2/4/8 GPUs",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "windows_ms_per_step": [round(w / args.steps * 1e3, 3) for w in windows],
            "timing": f"median of {len(windows)} back-to-back windows of {args.steps} steps each (barrier + synchronize around every window, "
                      "MAX over ranks per window)", "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (device-resident U[0,1) images, random-init weights seed 42)",
            "config": {"workload": wl["name"], "workload_key": args.workload, "classes": C, "batch_per_gpu": B, "global_batch": B * world, "image": "3x224x224",
                       "loss": wl["loss"], "optimizer": "adam lr=1e-3", "parallelism": f"dp{world}"},
            "final_loss": round(loss_value, 5),
            "step_mfma_frac": round(B * args.steps / elapsed * CONV_GFLOP_PER_IMAGE * 1e9 / (MFMA_F32_PEAK_TFLOPS * 1e12), 4),
        }
        if prof:
            conv_ms = sum(prof[k]["ms_per_step"] for k in ("conv_fwd", "conv_dgrad", "conv_wgrad"))
            achieved = B * CONV_GFLOP_PER_IMAGE * 1e9 / (conv_ms * 1e-3) / 1e12
            # HBM-side traffic of the conv kernels per step cannot be read from inside the process; it is the rocprofv3 PMC
            # measurement of this same command committed under profiles/ (2*FETCH_SIZE + WRITE_SIZE, KiB units, the x2 is the
            # gfx950 wide-read correction of MI355X_MICROARCH.md, validated on the Adam kernel's known 380/285 MB).
            traffic, tprov = committed_traffic(ms_step, B, args.workload)
            issued = issued_conv_gflop(B, N.lib())
            issued_total = sum(issued.values())
            busy, issued_measured, mprov = committed_mfma_util(ms_step, B, args.workload)
            out["roofline"] = {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                "frac_is": "EFFECTIVE rate: the direct convolution's FLOPs (SURVEY.md section 8d) over the conv kernels' time over the peak — the Winograd "
                           "layers' 2.25x fewer multiplies show as rate, so a single kernel can exceed 1; `issued_frac` is the matrix pipe's own utilisation",
                "issued_gflop_per_step": round(issued_total, 1),
                "issued_frac": round(issued_total * 1e9 / (conv_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "issued_per_class_gflop": {k: round(v, 1) for k, v in issued.items()},
                "issued_how": "matrix FLOPs the kernels issue: direct layers = the convolution's multiplies, Winograd layers (knobs fwd_wino / dgrad_wino / "
                              "wgrad_wino) = 16 products per 2x2 tile and channel pair incl. border-tile padding; over the same event-timed conv time",
                "mfma_busy_percent": busy, "issued_gflop_per_step_measured": issued_measured,
                "mfma_busy_provenance": mprov,
                "mfma_busy_note": "share of each class's launches during which a SIMD's matrix pipe is busy (rocprofv3 --pmc MfmaUtil) and the issued "
                                  "FLOPs measured as SQ_VALU_MFMA_BUSY_CYCLES x 64, from the committed summary named in mfma_busy_provenance; null when that "
                                  "profile is of another workload or its step time is >15% off this run",
                "traffic": traffic,
                "traffic_provenance": tprov,
                "traffic_note": "bytes per step at the L2<->fabric boundary for the three conv kernel classes, from the committed "
                                "rocprofv3 PMC summary named in traffic_provenance (--pmc FETCH_SIZE / WRITE_SIZE, separate passes); null "
                                "when that profile was taken on another workload or its step time is >15% off this run (stale); "
                                "algorithmic minimum ~36e9 (each conv input/output once fwd, dY+W / dY+X bwd)",
                "kernel": "conv stack, fp32 MFMA 32x32x2: implicit GEMM (k_conv_fwd + k_conv_dgrad + k_conv_wgrad incl. split-K reduce) for the 1x1 / "
                          "strided / stem layers, Winograd F(2x2,3x3) / F(3x3,2x2) (k_wino, k_wino_wgrad + weight transform, fix-up, reduce) for the "
                          "thirteen 3x3 stride-1 layers — `achieved` counts the DIRECT convolution's FLOPs (SURVEY.md section 8d) for both, so the "
                          "Winograd layers' 2.25x fewer multiplies show as rate",
                "how": f"24.287 GFLOP/img x {B} img per step / summed HIP-event duration of the conv launches per step ({conv_ms:.2f} ms), "
                       f"events recorded on the launch stream over {psteps} instrumented steps run straight after the timed region "
                       "(serialised: no side-stream overlap, so every class's time is its own)",
                "serialized_ms_per_step": round(sum(v["ms_per_step"] for v in prof.values()), 3),
                "per_class": {k: {"ms_per_step": round(v["ms_per_step"], 3),
                                  "tflops": round(B * g / v["ms_per_step"], 2) if g else None}
                              for (k, v), g in zip(prof.items(), (CONV_GFLOP_FWD, CONV_GFLOP_DGRAD, CONV_GFLOP_WGRAD, 0, 0, 0))},
            }
        if evalrec is not None:
            out["eval"] = evalrec
        if sustained is not None:
            sustained["vs_value"] = round(sustained["images_per_sec"] / out["value"], 4)
            out["sustained"] = sustained
        if rccl is not None:
            rccl["launcher"] = {"self_launched": os.environ.get("OSI_BENCH_LAUNCHED") == "1", "cpu_binding": os.environ.get("OSI_BENCH_BIND_POLICY")}
            out["rccl"] = rccl
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, wl["p_neg"], gpus_used=world, cpus_per_gpu=args.cpus_per_gpu)
            out["parity"] = parity
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
