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
    (1, 128, 64, (4, 8, 20), False, False, True, 0.0, False, False),   # 8 chunks: streaming-weights mode
    (1, 80, 32, (5, 9, 17), True, False, False, 1.0, False, True),     # streaming mode, unaligned
    (2, 32, 32, (8, 16, 32), False, False, True, 0.0, True, True),     # several tiles per workgroup range, both residuals
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


C1_CASES = [  # N, C1, C2, Cout, dims, out_f32, affine, slope, pre, post
    (1, 32, 0, 32, (4, 8, 16), False, True, 0.1, False, False),
    (2, 32, 32, 32, (3, 5, 12), False, True, 1.0, False, False),      # fuse: two inputs
    (1, 32, 0, 27, (6, 7, 20), True, False, 1.0, False, False),       # tap expansion of a logit head: 27 fp32 outputs
    (1, 32, 0, 32, (5, 9, 28), False, True, 0.0, True, True),         # redir-style with both residuals, ragged last group
    (1, 40, 24, 20, (2, 3, 8), True, True, 0.1, True, False),         # partial chunks in both inputs
]


@pytest.mark.parametrize("lp", LPS, ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", C1_CASES, ids=[str(c[:5]) for c in C1_CASES])
def test_conv1x1_lp(case, lp):
    from dcanet_amd import ops
    N, C1, C2, Cout, dims, out32, aff, slope, pre, post = case
    x = seeded_tensor(f"lp1.x{case}", (N, C1) + dims).to(lp)
    x2 = seeded_tensor(f"lp1.y{case}", (N, C2) + dims).to(lp) if C2 else None
    w = seeded_tensor(f"lp1.w{case}", (Cout, C1 + C2, 1, 1, 1)) * (2.0 / (C1 + C2)) ** 0.5
    odt = torch.float32 if out32 else lp
    scale = (torch.rand(Cout) + 0.5) if aff else None
    shift = torch.randn(Cout) * 0.1 if aff else None
    rp = seeded_tensor(f"lp1.p{case}", (N, Cout) + dims).to(odt) if pre else None
    rq = seeded_tensor(f"lp1.q{case}", (N, Cout) + dims).to(odt) if post else None
    xin = x if x2 is None else torch.cat([x, x2], 1)
    ref = F.conv3d(xin.double(), w.to(lp).double())
    if aff:
        ref = ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if pre:
        ref = ref + rp.double()
    ref = torch.where(ref > 0, ref, ref * slope)
    if post:
        ref = ref + rq.double()
    g = lambda t: None if t is None else t.to(DEV)
    with torch.no_grad():
        got = ops.conv1x1_lp(g(x), g(w), lp, g(x2), g(scale), g(shift), slope, g(rp), g(rq), odt)
    assert got.dtype == odt and got.shape == ref.shape
    err = (got.double().cpu() - ref).abs()
    lim = 2e-5 * max(1.0, ref.abs().max().item()) + (0 if out32 else ulp(lp)) * ref.abs()
    assert (err <= lim).all(), f"max err {err.max().item():.3e}"


DC_CASES = [  # N, Cin, Cout, coarse dims, affine, slope, pre, post
    (1, 64, 32, (2, 8, 16), True, 0.0, True, False),       # exactly one tile, cost_agg.conv3 form
    (2, 64, 32, (3, 5, 12), True, 0.0, True, True),        # partial tiles, both residuals (cva1: + cost0)
    (1, 16, 27, (2, 3, 5), False, 1.0, False, False),      # unaligned W, partial channel block, one chunk
    (1, 48, 32, (5, 17, 36), True, 0.1, False, True),      # several tiles per workgroup range, 3 chunks
]


@pytest.mark.parametrize("lp", LPS, ids=["bf16", "fp16"])
@pytest.mark.parametrize("exact", [False, True], ids=["lp", "fp32mfma"])
@pytest.mark.parametrize("case", DC_CASES, ids=[str(c[:4]) for c in DC_CASES])
def test_deconv3d_lp(case, lp, exact):
    from dcanet_amd import ops
    N, Cin, Cout, dims, aff, slope, pre, post = case
    if exact and dims[2] % 4:
        pytest.skip("the mixed-storage fp32-MFMA form needs W % 4 == 0")
    odims = tuple(2 * d for d in dims)
    x = seeded_tensor(f"dc.x{case}", (N, Cin) + dims)
    w = seeded_tensor(f"dc.w{case}", (Cin, Cout, 3, 3, 3)) * (2.0 / (27 * Cin / 8)) ** 0.5
    scale = (torch.rand(Cout) + 0.5) if aff else None
    shift = torch.randn(Cout) * 0.1 if aff else None
    rp = seeded_tensor(f"dc.p{case}", (N, Cout) + odims).to(lp) if pre else None
    rq = seeded_tensor(f"dc.q{case}", (N, Cout) + odims).to(lp) if post else None
    xr, wr = (x.double(), w.double()) if exact else (x.to(lp).double(), w.to(lp).double())
    ref = F.conv_transpose3d(xr, wr, None, stride=2, padding=1, output_padding=1)
    if aff:
        ref = ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if pre:
        ref = ref + rp.double()
    ref = torch.where(ref > 0, ref, ref * slope)
    if post:
        ref = ref + rq.double()
    g = lambda t: None if t is None else t.to(DEV)
    with torch.no_grad():
        got = ops.deconv3d_lp(g(x), g(w), lp, g(scale), g(shift), slope, g(rp), g(rq), exact=exact)
    assert got.dtype == lp and got.shape == ref.shape
    err = (got.double().cpu() - ref).abs()
    lim = 2e-5 * max(1.0, ref.abs().max().item()) + ulp(lp) * ref.abs()
    assert (err <= lim).all(), f"max err {err.max().item():.3e}"


S2_CASES = [  # N, Cin, Cout, fine dims, affine, slope
    (1, 32, 64, (4, 16, 32), True, 0.0),       # exactly one tile, cost_agg.conv1 form
    (2, 32, 64, (6, 10, 24), True, 0.0),       # partial tiles
    (1, 40, 27, (5, 9, 20), False, 1.0),       # odd D / H (ceil), partial channel blocks, 5 chunks
    (1, 8, 64, (10, 36, 72), True, 0.1),       # several tiles per workgroup range, one chunk
]


@pytest.mark.parametrize("lp", LPS, ids=["bf16", "fp16"])
@pytest.mark.parametrize("exact", [False, True], ids=["lp", "fp32mfma"])
@pytest.mark.parametrize("case", S2_CASES, ids=[str(c[:4]) for c in S2_CASES])
def test_conv3d_s2_lp(case, lp, exact):
    from dcanet_amd import ops
    N, Cin, Cout, dims, aff, slope = case
    x = seeded_tensor(f"s2.x{case}", (N, Cin) + dims).to(lp)
    w = seeded_tensor(f"s2.w{case}", (Cout, Cin, 3, 3, 3)) * (2.0 / (27 * Cin)) ** 0.5
    scale = (torch.rand(Cout) + 0.5) if aff else None
    shift = torch.randn(Cout) * 0.1 if aff else None
    ref = F.conv3d(x.double(), (w if exact else w.to(lp)).double(), None, 2, 1)
    if aff:
        ref = ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * slope)
    g = lambda t: None if t is None else t.to(DEV)
    with torch.no_grad():
        got = ops.conv3d_s2_lp(g(x), g(w), g(scale), g(shift), slope, exact=exact)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    err = (got.double().cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f"max err {err:.3e}"


def test_conv3d_lp_refuses_training():
    from dcanet_amd import ops
    x = torch.zeros(1, 32, 4, 8, 16, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(32, 32, 3, 3, 3, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError, match="inference only"):
        ops.conv3d_lp(x, w, torch.bfloat16)


def _seeded_model(maxdisp):
    from oracle import dcanet_oracle as O
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    m = GwcNet(maxdisp, use_concat_volume=False)
    sd = O.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval(), sd


@pytest.mark.timeout(1200)
def test_reduced_precision_epe_gate_full_size(capsys):
    """SURVEY 8(d) gate for BASELINE configs 2 (bf16) and 5 (fp16) at 544x960 / D=192, eval forward from the 1/4-res
    features through the convex up-sampler: |EPE_build - EPE_ref| <= 1e-3 against a common dense synthetic ground truth
    (U(1,191), the bench's own), where ref = the fp32 CPU oracle on identical inputs and weights.  The mean / max
    per-pixel deviation from the fp32 oracle is reported (expected floor with 8-bit-mantissa operands through ~40 layers:
    ~1e-1 px bf16, ~1e-2 px fp16 -- SURVEY Appendix E.3), and bounded loosely so a broken kernel cannot hide in the EPE."""
    from oracle import dcanet_oracle as O
    from dcanet_amd import ops
    m, sd = _seeded_model(192)
    fL, fR = seeded_tensor("full.fL", (1, 320, 136, 240)), seeded_tensor("full.fR", (1, 320, 136, 240))
    guid = seeded_tensor("full.guid", (1, 64, 136, 240))
    gt = torch.rand(1, 1, 544, 960, generator=torch.Generator().manual_seed(5)) * 190.0 + 1.0
    with torch.no_grad():
        ref_q = O.hot_path(sd, fL, fR, 192, False)["pred4_q"]
        ref = O.prop(sd, guid, ref_q, False)
        f32 = m.prop(guid.to(DEV), m.hot_path(fL.to(DEV), fR.to(DEV))["pred4_q"]).cpu()
    near = ref + torch.randn(ref.shape, generator=torch.Generator().manual_seed(6))      # a ground truth close to the prediction
    epe = lambda x, g: (x - g).abs().mean().item()
    rows = [("fp32 kernels", f32)]
    for lp in LPS:
        with torch.no_grad(), ops.reduced_precision(lp):
            full = m.prop(guid.to(DEV), m.hot_path(fL.to(DEV), fR.to(DEV))["pred4_q"]).cpu()
        rows.append((str(lp)[6:], full))
    with capsys.disabled():
        print(f"\n[reduced precision, 544x960 D=192] EPE_ref(U(1,191) GT) = {epe(ref, gt):.4f}, EPE_ref(near GT) = {epe(ref, near):.4f}")
        for name, full in rows:
            dev = (full - ref).abs()
            print(f"   {name:12s} |dEPE| uniform GT {abs(epe(full, gt) - epe(ref, gt)):.2e}   near GT {abs(epe(full, near) - epe(ref, near)):.2e}"
                  f"   mean|d| {dev.mean().item():.3e}   max|d| {dev.max().item():.3e}")
    assert (f32 - ref).abs().max().item() <= 1e-3
    # Gates.  The SURVEY gate (|dEPE| <= 1e-3 against the uniform ground truth) is insensitive on its own: EPE_ref is ~48 px
    # there and sign-symmetric errors cancel.  So the per-pixel deviation from the fp32 oracle is bounded at ~1.5x what is
    # measured (bf16 mean 0.069 / max 0.67 px, fp16 0.0083 / 0.068 px), and |dEPE| is also gated against a ground truth
    # CLOSE to the prediction (oracle + N(0,1) px), where fp16 meets 1e-3 and bf16 does not (measured 2.9e-3: 8-bit
    # mantissas through ~40 layers) -- fp16 is the supported reduced-precision type for the 1e-3 EPE figure
    # (INTEGRATION.md section 5), bf16 is held to 5e-3 there.
    for name, full in rows[1:]:
        bf = name == "bfloat16"
        assert abs(epe(full, gt) - epe(ref, gt)) <= 1e-3, name
        assert abs(epe(full, near) - epe(ref, near)) <= (5e-3 if bf else 1e-3), name
        dev = (full - ref).abs()
        assert dev.mean().item() <= (0.1 if bf else 0.015), (name, dev.mean().item())
        assert dev.max().item() <= (1.0 if bf else 0.12), (name, dev.max().item())


def test_reduced_precision_is_inference_only():
    from dcanet_amd import ops
    m, _ = _seeded_model(32)
    fL = seeded_tensor("lp.tr.fL", (1, 320, 16, 32)).to(DEV)
    with ops.reduced_precision(torch.bfloat16):
        with pytest.raises(RuntimeError, match="inference only"):
            m.train().hot_path(fL, fL)
    with torch.no_grad():      # and the context leaves nothing behind
        a = m.eval().hot_path(fL, fL)["pred4_q"]
        with ops.reduced_precision(torch.float16):
            b = m.hot_path(fL, fL)["pred4_q"]
        c = m.hot_path(fL, fL)["pred4_q"]
    assert torch.equal(a, c) and not torch.equal(a, b)


@pytest.mark.timeout(900)
def test_config5_kitti_frame_fp16_hipgraph(capsys):
    """BASELINE config 5: KITTI 384x1248 frame (quarter-res 96 x 312), D=192, fp16, the 3D part replayed from a hipGraph.
    The replay must equal the eager reduced-precision run bit for bit, and pass the EPE gate against the fp32 CPU oracle."""
    from oracle import dcanet_oracle as O
    from dcanet_amd import ops
    from dcanet_amd.graph import GraphedHotPath
    m, sd = _seeded_model(192)
    fL, fR = seeded_tensor("kitti.fL", (1, 320, 96, 312)), seeded_tensor("kitti.fR", (1, 320, 96, 312))
    guid = seeded_tensor("kitti.guid", (1, 64, 96, 312))
    gt = torch.rand(1, 1, 384, 1248, generator=torch.Generator().manual_seed(9)) * 190.0 + 1.0
    with torch.no_grad():
        ref = O.prop(sd, guid, O.hot_path(sd, fL, fR, 192, False)["pred4_q"], False)
        with ops.reduced_precision(torch.float16):
            eager = m.hot_path(fL.to(DEV), fR.to(DEV))["pred4_q"].clone()
            g = GraphedHotPath(m, fL.to(DEV), fR.to(DEV))
        replay = g(fL.to(DEV), fR.to(DEV))["pred4_q"]          # outside the context: the graph holds the fp16 kernels
        assert torch.equal(replay, eager), f"replay differs by {(replay - eager).abs().max().item():.3e}"
        full = m.prop(guid.to(DEV), replay).cpu()
    epe = lambda x: (x - gt).abs().mean().item()
    dev = (full - ref).abs()
    with capsys.disabled():
        print(f"\n[config 5, 384x1248 fp16 graph] |dEPE| {abs(epe(full) - epe(ref)):.2e}  mean|d| {dev.mean().item():.3e}  max|d| {dev.max().item():.3e}")
    assert abs(epe(full) - epe(ref)) <= 1e-3 and dev.mean().item() <= 0.015 and dev.max().item() <= 0.15
