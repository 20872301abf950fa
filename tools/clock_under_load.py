"""Dev tool (GPU box): what the card reports about itself while the training step runs.

    python tools/clock_under_load.py [seconds] [workload]

Runs the bench step (Protocol-2 shapes, B = 128, the product path) in a loop for `seconds` while a host thread samples
`amd-smi metric --clock --power --temperature --json` (falling back to `rocm-smi --showclocks --showpower --json`) every ~0.3 s, and
once more when the GPU is idle before / after. Prints one JSON line: per-sample shader clocks, socket power, the power cap, and the
step rate measured over the same interval. It is the out-of-kernel counterpart of the in-kernel clock reads in
tools/probes/conv_ablate.hip (s_memtime / s_memrealtime; its `steady` and `long` modes) and of tools/clock_in_step.sh (GRBM_GUI_ACTIVE):
all three put the busy training step at ~2.35 GHz, well inside the power cap.
"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "openset-imagenet_amd")]


def _run(cmd):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
        return r.stdout if r.returncode == 0 else None
    except Exception:
        return None


def sample():
    """One reading as a small dict (whatever the installed tools expose); never raises."""
    out = {"t": time.time()}
    txt = _run(["amd-smi", "metric", "-g", "0", "--clock", "--power", "--temperature", "--json"])
    if txt:
        try:
            out["amd_smi"] = json.loads(txt)
            return out
        except Exception:
            out["amd_smi_raw"] = txt[:2000]
    txt = _run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"])
    if txt:
        try:
            out["rocm_smi"] = json.loads(txt)
        except Exception:
            out["rocm_smi_raw"] = txt[:2000]
    return out


def static_info():
    info = {}
    for name, cmd in (("amd_smi_static", ["amd-smi", "static", "-g", "0", "--limit", "--json"]),
                      ("rocm_smi_maxpower", ["rocm-smi", "-d", "0", "--showmaxpower", "--json"])):
        txt = _run(cmd)
        if txt:
            try:
                info[name] = json.loads(txt)
            except Exception:
                info[name + "_raw"] = txt[:2000]
    return info


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    import torch
    from bench import WORKLOADS, synthetic_batch
    from openset_imagenet import ResNet50, EntropicOpensetLoss, optim, tools
    wl = WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "p2"]
    dev = tools.set_device_gpu(0)
    B, C = wl["B"], wl["C"]
    torch.manual_seed(42)
    model = tools.device(ResNet50(C, C, False))
    opt = optim.Adam(model.parameters(), lr=1e-3)
    loss_fn = EntropicOpensetLoss(C, 1.0)
    images, labels = synthetic_batch(B, C, wl["p_neg"], "entropic", dev, 42)

    def step():
        model.train()
        opt.zero_grad()
        logits, _ = model(images)
        loss_fn(logits, labels).backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    result = {"workload": wl["name"], "static": static_info(), "idle_before": sample()}
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(sample())
            stop.wait(0.3)

    th = threading.Thread(target=poll, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        n += 10
    elapsed = time.perf_counter() - t0
    stop.set()
    th.join()
    time.sleep(2.0)
    result["idle_after"] = sample()
    result["steps"], result["ms_per_step"], result["images_per_s"] = n, elapsed / n * 1e3, n * B / elapsed
    result["under_load"] = samples
    print(json.dumps(result))


if __name__ == "__main__":
    main()
