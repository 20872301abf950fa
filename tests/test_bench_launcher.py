"""bench.py's self-launch of N > 1 ranks (`python bench.py --gpus N` with no rank variables in the environment): host logic only —
the launcher never touches a GPU, so its process handling is testable here. The N = 2 run itself is rehearsed on the GPU box
(tests/test_dp_two_ranks_gpu.py)."""
import importlib.util
import os
import re
import subprocess
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:      # a zombie still answers kill(0)
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_failing_rank_fails_the_launch_and_leaves_no_orphan():
    """Rank 1 exits with an error while rank 0 would wait forever: the parent must return non-zero quickly and terminate rank 0."""
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(OSI_BENCH_FAULT="1:exit,0:sleep"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 60
    assert not r.stdout.strip(), "no JSON line from a failed launch"
    assert "rank 1 exited with code 3" in r.stderr
    pids = [int(p) for p in re.search(r"pids \[([\d, ]+)\]", r.stderr).group(1).split(",")]
    assert len(pids) == 2
    time.sleep(0.5)
    assert not any(_alive(p) for p in pids), "a rank survived its failed launch"


def test_launch_timeout_terminates_the_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-timeout", "2"], env=_clean_env(OSI_BENCH_FAULT="0:sleep,1:sleep"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no result after 2 s" in r.stderr
    pids = [int(p) for p in re.search(r"pids \[([\d, ]+)\]", r.stderr).group(1).split(",")]
    time.sleep(0.5)
    assert not any(_alive(p) for p in pids)


def test_mismatched_world_is_a_mislaunch():
    """A rank environment that does not match --gpus is refused with a message (the old behaviour for EVERY plain N > 1 invocation)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_clean_env(RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE = 1" in r.stderr


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cpu_binding_plan_partitions_the_affinity_mask():
    b = _bench_module()
    assert b._cpu_list("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    mine = sorted(os.sched_getaffinity(0))
    for n in (1, 2, 4, 8):
        if n > len(mine):
            continue
        cpus, policy = b.plan_rank_cpus(n, "even")
        assert policy == "even" and len(cpus) == n
        flat = [c for part in cpus for c in part]
        assert len(set(flat)) == len(flat) and set(flat) <= set(mine), "disjoint shares of this process's own mask"
        assert len({len(part) for part in cpus}) == 1 and len(cpus[0]) == len(mine) // n
    assert b.plan_rank_cpus(2, "none") == (None, "none")
    cpus, policy = b.plan_rank_cpus(2, "auto")         # no KFD topology in the build container: falls back to the even split
    assert policy in ("even", "near") and len(cpus) == 2


def test_reserved_cus_follow_the_channel_count():
    sys.path.insert(0, os.path.join(ROOT, "openset-imagenet_amd"))
    from openset_imagenet.dp import reserved_cus_for_channels
    assert [reserved_cus_for_channels(c) for c in (None, 0, 1, 8, 9, 32, 128, 1000)] == [0, 0, 1, 1, 2, 4, 16, 32]


def test_rank_under_an_external_launcher_takes_its_own_cpu_share():
    """`python -m torch.distributed.run ... bench.py --gpus N` (the driver's form for N > 1) does not pass through launch_ranks: every
    rank then binds itself to the share the self-launcher would have dealt it, before torch loads; OSI_BENCH_BIND=none turns it off."""
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 2:
        return
    code = f"import importlib.util as u, os; s = u.spec_from_file_location('b', {BENCH!r}); m = u.module_from_spec(s); s.loader.exec_module(m); " \
           "print('CPUS', m.BOUND_CPUS, os.environ.get('OSI_BENCH_BIND_POLICY'))"
    shares = []
    for rank in (0, 1):
        r = subprocess.run([sys.executable, "-c", code], env=_clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                                                                         OSI_BENCH_BIND="even"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        line = [l for l in r.stdout.splitlines() if l.startswith("CPUS")][-1]
        shares.append(eval(line[5:line.rindex("]") + 1]))
        assert line.endswith("rank-side:even")
    assert shares[0] == mine[:len(mine) // 2] and shares[1] == mine[len(mine) // 2:2 * (len(mine) // 2)]
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", OSI_BENCH_BIND="none"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CPUS None None" in r.stdout


def test_near_binding_deals_a_locality_domain_out_among_its_ranks(monkeypatch):
    """The "near" plan on a made-up topology (sysfs is not consulted): ranks whose GPUs share a locality domain split that domain's
    CPUs evenly; a domain with fewer CPUs than ranks, an unresolved GPU, or a visibility mask fall back to the even split."""
    b = _bench_module()
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 4:
        return
    half = len(mine) // 2
    dom = [mine[:half], mine[half:2 * half]]
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(b, "_cpus_near_gpu", lambda r: dom[r // 2] if r < 4 else None)
    cpus, policy = b.plan_rank_cpus(4, "near")               # GPUs 0, 1 near domain 0; GPUs 2, 3 near domain 1
    if half >= 2:
        assert policy == "near"
        per = half // 2
        assert cpus == [dom[0][:per], dom[0][per:2 * per], dom[1][:per], dom[1][per:2 * per]]
    cpus, policy = b.plan_rank_cpus(8, "auto")               # GPUs 4..7 do not resolve
    assert policy in ("even", "none")
    monkeypatch.setattr(b, "_cpus_near_gpu", lambda r: mine[:1])     # eight ranks on a one-CPU domain
    assert b.plan_rank_cpus(2, "near")[1] == "even"
    monkeypatch.setattr(b, "_cpus_near_gpu", lambda r: dom[r % 2])
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")          # a mask may have reordered the devices: sysfs order is not HIP's
    assert b.plan_rank_cpus(2, "auto")[1] == "even"
