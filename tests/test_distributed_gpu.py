"""The RCCL path on the GPU box: bench.py's multi-GPU timed region over a real `nccl` process group.  One GPU is all a
gpurun box has, so the group has one rank -- enough to run every call the N-rank job makes (communicator set-up, gather of every / the final
frame, in-place root slot, barrier, max-reduction of the times) with the HIP kernels rendering into the collective's buffers.  The N > 1
arithmetic of the same code (bands, padding, placement) is covered on CPU by tests/test_distributed_gloo.py at world sizes 2, 4 and 8."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_timed_loop_distributed_over_rccl_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # a fresh interpreter: the process group must exist before the first GPU call, and this pytest process has long made some
    p = subprocess.run([sys.executable, os.path.join(HERE, "dist_gpu_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["backend"] == "nccl"
    assert set(rec["seconds"]) >= {"no_clouds_32x8_direct/final", "no_clouds_32x8_direct/every", "clouds_high_rm/final", "clouds_high_rm/bands-every"}
