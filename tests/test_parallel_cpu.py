"""world_size-2 gloo test of the DP gradient bucket (the N>1 path of bench.py), on CPU."""
import json
import os
import socket
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_flat_bucket_allreduce_matches_single_process():
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    got = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][0]
        got.append(torch.tensor(json.loads(line[7:])))
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 1))
    x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10.0
    lin(x).sum().backward()
    want = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]) / world   # mean over ranks of shard sums
    assert torch.allclose(got[0], got[1])
    assert torch.allclose(got[0], want, atol=1e-6)
