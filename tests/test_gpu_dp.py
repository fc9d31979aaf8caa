"""Data-parallel training step (BASELINE config 4) on ONE GPU: two gloo ranks share cuda:0 and run the
DCANet training step of bench.py on their shards (batch 2 each) through `FlatGradBucket`.  What replaces the
reference's nn.DataParallel (main_dca.py:54-55) must (1) hand Adam the MEAN of the two ranks' gradients -- compared with the
gradients two single-process runs produce for the same shards (BatchNorm is per replica, as in the reference) --, (2) leave
parameters and Adam moments bitwise equal on both ranks after two steps, (3) not accumulate step 0's gradients into
step 1.  The same worker runs over RCCL on a multi-GPU node (DCA_DIST_BACKEND=nccl)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_dp_gpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(envs, outs):
    procs = [subprocess.Popen([sys.executable, WORKER, o], env=dict(os.environ, **e), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for e, o in zip(envs, outs)]
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    return [torch.load(o, weights_only=True) for o in outs]


@pytest.mark.timeout(900)
def test_two_rank_train_step_reduces_to_the_mean_of_single_process_gradients(tmp_path, capsys):
    port = _free_port()
    base = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DCA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the two ranks, concurrently on cuda:0
    r0, r1 = _run([dict(base, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2") for r in range(2)],
                  [str(tmp_path / f"rank{r}.pt") for r in range(2)])
    # single-process runs of the same shards (one after the other)
    s0, = _run([dict(base, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", DP_SHARD="0", DP_NSHARDS="2")], [str(tmp_path / "s0.pt")])
    s1, = _run([dict(base, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", DP_SHARD="1", DP_NSHARDS="2")], [str(tmp_path / "s1.pt")])
    assert r0["dist_world"] == r1["dist_world"] == 2 and r0["backend"] == "gloo"

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    # step 0: every rank's local gradient is what a single process computes for that shard (the HIP kernels' reductions are
    # order-fixed, so normally bitwise; MIOpen may pick another algorithm for the up-sampler's two 2D convolutions when two
    # processes share the GPU, hence a tolerance), and the reduced bucket is the mean of the two single-process gradients
    el = max(rel(r0["local0"].double(), s0["local0"].double()), rel(r1["local0"].double(), s1["local0"].double()))
    assert el <= 1e-5, el
    want0 = (s0["local0"].double() + s1["local0"].double()) / 2
    e0 = max(rel(r0["reduced0"].double(), want0), rel(r1["reduced0"].double(), want0))
    assert torch.equal(r0["reduced0"], r1["reduced0"])
    assert e0 <= max(1e-6, 2 * el), (e0, el)
    # ... and exactly (to fp32 rounding of one add and one divide) the mean of what the two ranks put into the bucket
    ex = rel(r0["reduced0"].double(), (r0["local0"].double() + r1["local0"].double()) / 2)
    assert ex <= 1e-6, ex
    assert r0["local0"].abs().max() > 0 and not torch.equal(r0["local0"], r1["local0"])
    # step 1 starts from the all-reduced update (differs from the single-process runs' parameters) and must not contain step 0
    want1 = (r0["local1"].double() + r1["local1"].double()) / 2
    e1 = max(rel(r0["reduced1"].double(), want1), rel(r1["reduced1"].double(), want1))
    assert e1 <= 1e-6 and torch.equal(r0["reduced1"], r1["reduced1"]), e1
    # identical optimizer trajectories on both ranks
    for k in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(r0[k], r1[k]), k
    assert not torch.equal(r0["params"], s0["params"])      # the reduced gradient, not the local one, was applied
    with capsys.disabled():
        print(f"\n[dp, 2 gloo ranks on one GPU] local vs single-process {el:.2e} (bitwise: "
              f"{torch.equal(r0['local0'], s0['local0']) and torch.equal(r1['local0'], s1['local0'])}); "
              f"|reduced - mean(single-process)| rel: step0 {e0:.2e}, vs own locals {ex:.2e}, step1 {e1:.2e}; "
              f"losses {r0['loss0']:.4f} / {r1['loss0']:.4f}; bucket {r0['reduced0'].numel()} floats")
