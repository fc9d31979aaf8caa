"""GPU parity of the reduced-precision inference kernels (BASELINE configs 2 "bf16" / 5 "fp16"):
bf16 / fp16 storage, one MFMA product per multiply, fp32 accumulation.  Kernel-level gates compare with an fp64
evaluation of the SAME rounded operands (products of 2-byte operands are exact in fp32, so only the summation order and
the final rounding to the storage type differ); the end-to-end gate is SURVEY 8(d)'s |EPE_build - EPE_ref| <= 1e-3
against a common ground truth, with the mean-abs deviation from the fp32 oracle reported."""
import pytest
import torch
import torch.nn.functional as F

from oracle.seeded import seeded_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"
LPS = [torch.bfloat16, torch.float16]


def ulp(dtype):
    return 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11


CASES = [  # N, Cin, Cout, (D,H,W), in_f32, out_f32, affine, slope, pre, post
    (1, 32, 32, (4, 8, 16), False, False, True, 0.0, False, False),
    (2, 40, 32, (5, 7, 20), True, False, True, 0.0, False, False),     # dres0.0: fp32 volume in, W % 4 == 0
    (1, 64, 64, (3, 9, 36), False, False, True, 0.0, False, True),     # two output-channel blocks, res_post
    (1, 32, 32, (6, 10, 18), False, True, False, 1.0, False, False),   # W % 4 != 0 (unaligned staging), fp32 out
    (1, 16, 27, (6, 6, 12), True, True, True, 0.1, True, True),        # partial channel chunk / block, both residuals
    (1, 32, 32, (9, 17, 33), False, False, True, 0.0, True, False),    # odd sizes, partial tiles everywhere
]


@pytest.mark.parametrize("lp", LPS, ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=[str(c[:4]) for c in CASES])
def test_conv3d_lp(case, lp):
    from dcanet_amd import ops
    N, Cin, Cout, dims, in32, out32, aff, slope, pre, post = case
    x = seeded_tensor(f"lp.x{case}", (N, Cin) + dims)
    w = seeded_tensor(f"lp.w{case}", (Cout, Cin, 3, 3, 3)) * (2.0 / (27 * Cin)) ** 0.5
    odt = torch.float32 if out32 else lp
    scale = (torch.rand(Cout) + 0.5) if aff else None
    shift = torch.randn(Cout) * 0.1 if aff else None
    rp = seeded_tensor(f"lp.p{case}", (N, Cout) + dims).to(odt) if pre else None
    rq = seeded_tensor(f"lp.q{case}", (N, Cout) + dims).to(odt) if post else None
    xin = x if in32 else x.to(lp)
    ref = F.conv3d(xin.to(lp).double(), w.to(lp).double(), None, 1, 1)
    if aff:
        ref = ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if pre:
        ref = ref + rp.double()
    ref = torch.where(ref > 0, ref, ref * slope)
    if post:
        ref = ref + rq.double()
    g = lambda t: None if t is None else t.to(DEV)
    with torch.no_grad():
        got = ops.conv3d_lp(g(xin), g(w), lp, g(scale), g(shift), slope, g(rp), g(rq), odt)
    assert got.dtype == odt and got.shape == ref.shape
    err = (got.double().cpu() - ref).abs()
    lim = 2e-5 * max(1.0, ref.abs().max().item()) + (0 if out32 else ulp(lp)) * ref.abs()
    assert (err <= lim).all(), f"max err {err.max().item():.3e} at |ref| {ref.abs().flatten()[err.argmax()].item():.3e}"


def test_conv3d_lp_refuses_training():
    from dcanet_amd import ops
    x = torch.zeros(1, 32, 4, 8, 16, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(32, 32, 3, 3, 3, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError, match="inference only"):
        ops.conv3d_lp(x, w, torch.bfloat16)
