"""A/B of the conv3_f16x2 tile traversal (DCA_X2_ORDER=0/1, read once per process): time of the packed and fp32-operand launches
at the headline layer shape (32->32, 48x136x240), one sample and the batch of four."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcanet_amd import ops
dev = torch.device("cuda")
lib = ops._L()
for N in (1, 4):
    x = torch.randn(N, 32, 48, 136, 240, device=dev)
    wgt = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    y = torch.empty_like(x)
    for packed in (True, False):
        xin = ops.pack_x2(x) if packed else x
        xex = ops._exps_of(xin)
        w2 = torch.empty((lib.dca_conv3d_x2_weight_bytes(32, 32) // 2,), device=dev, dtype=torch.int16)
        ops._chk(lib.dca_conv3d_x2_prep_weight(ops._ptr(wgt), ops._ptr(w2), 32, 32, 0, 0, None, 0, ops._ptr(xex), ops._stream()), "prep")
        run = lambda: ops._chk(lib.dca_conv3d_x2_forward(ops._ptr(xin), int(packed), ops._ptr(xex), ops._ptr(w2), ops._ptr(y), None, None,
                                                          None, None, 1.0, None, N, 32, 32, 48, 136, 240, ops._stream()), "fwd")
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f"order {os.environ.get('DCA_X2_ORDER', '0')}  N={N} packed={packed}: {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
