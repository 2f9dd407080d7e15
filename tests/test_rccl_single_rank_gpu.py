"""RCCL on the GPU box, one rank: library load, communicator creation as sr355.dist does it, and the collectives of bench.py /
tools/bench_train.py with their dtypes and sizes (tools/rccl_selftest.py).  More ranks need more GPUs than the builder's box has; the gloo
tests (tests/test_dist_cpu.py) cover the logic, this covers the backend."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_selftest():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "rccl_selftest.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "rccl selftest ok: backend nccl" in r.stdout, r.stdout[-500:]
