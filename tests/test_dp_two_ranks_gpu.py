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
