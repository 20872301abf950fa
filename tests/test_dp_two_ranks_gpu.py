"""Two data-parallel ranks on ONE MI355X (gloo between them): the closest rehearsal of the N > 1 path a one-GPU box allows.
RCCL cannot place two ranks on one device, so the collective backend here is gloo on GPU tensors; everything else — the HIP
executor's staged backward, the bucket schedule, the wrapper's broadcast, the fused optimizer — is the production code."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(600)
def test_two_ranks_share_one_gpu(cuda):
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_two_ranks_worker.py")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=500)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank}: ok" in out


RANK_VARS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def _run_bench(extra_args, env_extra, nproc, self_launch=False):
    """self_launch: ONE plain `python bench.py --gpus N` with no rank variables in the environment (bench.py starts its own ranks);
    otherwise one process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as torch.distributed.run does."""
    import json
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    port = _free_port()
    procs = []
    base = {k: v for k, v in os.environ.items() if k not in RANK_VARS}
    if "--sustained-steps" not in extra_args:
        extra_args = [*extra_args, "--sustained-steps", "0"]
    for rank in range(1 if self_launch else nproc):
        env = dict(base, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        if not self_launch:
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            if nproc > 1:
                env.update(RANK=str(rank), WORLD_SIZE=str(nproc), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(nproc), *extra_args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=500))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for rank, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{e[-3000:]}"
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints exactly one JSON line on rank 0's stdout"
    assert all(not o.strip() for o, _ in outs[1:]), "only rank 0 prints"
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_two_rank_rehearsal_reports_the_proof_of_ranks(cuda):
    """bench.py as the driver launches it for N = 2 (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), rehearsed on the one GPU of the test box with gloo carrying the collectives (OSI_BENCH_BACKEND=gloo; under RCCL
    two ranks on one device make bench.py exit with an error). The JSON line carries the three timed windows with their median as
    `value`, and the `rccl` object a SCALE record needs to prove that N processes on N devices took part."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-profile"], {"OSI_BENCH_BACKEND": "gloo"}, 2)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    w = out["windows_ms_per_step"]
    assert len(w) == 3 and sorted(w)[1] == out["ms_per_step"]
    assert abs(out["value"] - 2 * 8 / (out["ms_per_step"] * 1e-3)) <= 0.01 * out["value"]
    r = out["rccl"]
    assert r["world"] == 2 and r["backend"] == "gloo" and len(r["devices"]) == 2 and r["distinct_devices"] == 1   # the rehearsal shares GPU 0
    assert [d["rank"] for d in r["devices"]] == [0, 1]
    # the four stage buckets tile the gradient arena: 23 570 402 parameters (C = 30), each tensor padded to 16 bytes
    assert len(r["buckets"]) == 4 and r["allreduce_bytes_per_step"] == sum(b["bytes"] for b in r["buckets"])
    assert 4 * 23570402 <= r["allreduce_bytes_per_step"] <= 4 * 23570402 + 16 * 162
    assert r["exposed_comm_ms"] is not None and r["comm_steps"] >= 1
    # self-diagnosing record: what EVERY rank saw (own finish times per window, exposed communication), not only the MAX
    assert [q["rank"] for q in r["per_rank"]] == [0, 1] and all(len(q["own_ms_per_step_windows"]) == 3 for q in r["per_rank"])
    assert all(q["exposed_comm_ms"] is not None for q in r["per_rank"])
    pr = r["per_rank_ms_per_step"]
    assert 0 < pr["min"] <= pr["median"] <= pr["max"] <= max(w) * 1.001
    lp = r["launch_plan"]
    assert lp["hw_cus"] == 256 and lp["tail_cus"] == 0 and lp["dp_reserved_cus"] == 0 and lp["tail_plan_cus_in_effect"] == 256 and lp["tail_split"] == 1
    assert r["channels"] is None     # gloo rehearsal: no RCCL communicator


@pytest.mark.timeout(900)
def test_bench_self_launch_two_ranks(cuda):
    """The plain command: `python bench.py --gpus 2` with NO rank variables in the environment starts its own two ranks as child
    processes (free rendezvous port, per-rank CPU share), relays rank 0's one JSON line and returns 0. Same gloo rehearsal on the one
    GPU of the test box; the record says it was self-launched and how the ranks were bound."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-profile"], {"OSI_BENCH_BACKEND": "gloo"}, 2,
                     self_launch=True)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    r = out["rccl"]
    assert r["world"] == 2 and [d["rank"] for d in r["devices"]] == [0, 1]
    assert r["launcher"]["self_launched"] is True and r["launcher"]["cpu_binding"] in ("near", "even", "none")
    if r["launcher"]["cpu_binding"] != "none":
        assert all(q["cpus"] for q in r["per_rank"]) and r["per_rank"][0]["cpus"] != r["per_rank"][1]["cpus"]
    assert len({d["pid"] for d in r["devices"]} if "pid" in r["devices"][0] else {0, 1}) == 2


@pytest.mark.timeout(900)
def test_bench_self_launch_failing_rank(cuda):
    """A rank that dies takes the launch down: parent rc != 0, no JSON line, and the surviving rank — which by then holds the GPU and
    waits in the rendezvous — is terminated, not orphaned."""
    import re
    import time
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in RANK_VARS}
    env.update(OSI_BENCH_BACKEND="gloo", OSI_BENCH_FAULT="1:exit")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--batch", "8", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip() and "rank 1 exited with code 3" in r.stderr
    pids = [int(p) for p in re.search(r"pids \[([\d, ]+)\]", r.stderr).group(1).split(",")]
    time.sleep(1.0)
    for pid in pids:
        try:
            os.kill(pid, 0)
            state = open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0]
            assert state == "Z", f"rank process {pid} survived the failed launch"
        except (ProcessLookupError, OSError):
            pass


@pytest.mark.timeout(900)
def test_bench_sustained_leg(cuda):
    """After the timed windows (never part of `value`) the record carries a sustained leg in 50-step windows."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-profile", "--sustained-steps", "100"], {}, 1)
    s = out["sustained"]
    assert s["steps"] == 100 and len(s["window_ms_per_step"]) == 2 and s["ms_per_step_p50"] <= s["ms_per_step_p95"]
    assert s["images_per_sec"] == pytest.approx(8 * 100 / s["seconds"], rel=1e-3) and s["vs_value"] > 0
    assert out["value"] == pytest.approx(8 / (out["ms_per_step"] * 1e-3), rel=1e-2), "`value` comes from the timed windows alone"


@pytest.mark.timeout(900)
def test_bench_force_dp_world1_runs_the_rccl_path(cuda):
    """N = 1 with --force-dp: a world-1 RCCL communicator, the staged backward and one all_reduce per bucket, each bracketed by HIP
    events in the communication leg."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-profile", "--force-dp"], {}, 1)
    r = out["rccl"]
    assert r["world"] == 1 and r["backend"] == "nccl" and r["distinct_devices"] == 1 and r["nccl_version"]
    assert len(r["per_bucket_comm_ms"]) == 4 and all(v >= 0 for v in r["per_bucket_comm_ms"])
    assert r["comm_ms_per_step"] == pytest.approx(sum(r["per_bucket_comm_ms"]), abs=2e-3) and r["exposed_comm_ms"] >= 0
    assert len(out["windows_ms_per_step"]) == 3
    assert len(r["per_rank"]) == 1 and r["per_rank"][0]["exposed_comm_ms"] == r["exposed_comm_ms"]
    ch = r["channels"]
    assert set(ch) >= {"coll_channels", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "source"}
    print("RCCL channels:", ch)


@pytest.mark.timeout(900)
def test_bench_dp_reserved_cus_reaches_the_launch_plan(cuda):
    """OSI_DP_RESERVED_CUS (the wave slots left to RCCL's resident channel workgroups) travels environment -> osi_set_tuning ->
    plan_tail_split's CU count, and the record says which plan was in effect."""
    out = _run_bench(["--steps", "1", "--warmup", "1", "--windows", "1", "--batch", "8", "--no-cpu-baseline", "--no-profile", "--force-dp"],
                     {"OSI_DP_RESERVED_CUS": "8"}, 1)
    lp = out["rccl"]["launch_plan"]
    assert lp["dp_reserved_cus"] == 8 and lp["tail_plan_cus_in_effect"] == 248


@pytest.mark.timeout(900)
def test_bench_record_says_what_its_roofline_measures_and_carries_the_inference_leg(cuda):
    """`roofline.frac` is the EFFECTIVE rate (the direct convolution's FLOPs); `issued_frac` the matrix pipe's own
    (Winograd layers issue 4/9 of their multiplies); the committed counter summary is cited with its provenance and withheld when it
    belongs to another workload / batch; the `eval` object compares the inference forms with the training topology (never part of `value`)."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--windows", "1", "--batch", "8", "--no-cpu-baseline", "--eval-steps", "2"], {}, 1)
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and 0 < r["issued_frac"] < r["frac"]
    # B = 8: 24.287 GFLOP per image of direct convolution; the 13 Winograd layers issue 16 products per 2x2 tile instead of 36
    assert r["issued_gflop_per_step"] == pytest.approx(8 * 2491.657 / 128, rel=1e-3)
    assert set(r["issued_per_class_gflop"]) == {"conv_fwd", "conv_dgrad", "conv_wgrad"} and "EFFECTIVE" in r["frac_is"]
    prov = r["mfma_busy_provenance"]
    assert prov["file"].startswith("profiles/r") and len(prov["git_blob"]) == 40 and prov["matches_this_run"] is False   # the profile is B = 128
    assert r["mfma_busy_percent"] is None and r["issued_gflop_per_step_measured"] is None and r["traffic"] is None
    e = out["eval"]
    assert e["batch"] == 8 and e["steps"] == 2
    for k in ("training_topology", "fused", "validate_step"):
        assert e[k]["images_per_sec"] > 0 and e[k]["ms_per_batch"] > 0
    assert e["fused"]["serialized"]["launch_groups"] < e["training_topology"]["serialized"]["launch_groups"]
    assert e["speedup"] == pytest.approx(e["training_topology"]["ms_per_batch"] / e["fused"]["ms_per_batch"], rel=1e-2)
    assert out["value"] == pytest.approx(8 / (out["ms_per_step"] * 1e-3), rel=1e-2), "`value` comes from the timed windows alone"
