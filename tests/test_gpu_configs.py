"""GPU parity at the BASELINE.json configurations the small golden shapes do not reach (VERDICT r1 "configs untested"):
  * KITTI 384x1248 / D=192 frame (quarter-res 96 x 312: W is NOT a multiple of the 16-wide conv tile), eval forward,
    eager and as a hipGraph replay, vs the CPU oracle; and the my_img.py-style wrapper around it;
  * a TRAINING step at D=192 on a full-width row crop of the SceneFlow frame (the bench's own CPU-sample shape,
    features 2x320x34x240): every forward head and EVERY parameter gradient vs the oracle, G and GC;
  * weight-gradient kernels at the full 544x960 / D=192 size with batch 4: run-to-run bitwise determinism and
    additivity over the batch."""
import numpy as np
import pytest
import torch

from oracle import dcanet_oracle as O
from oracle.seeded import seeded_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"


def load_seeded(module):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(O.seeded_state_dict(shapes), strict=True)
    return module


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.timeout(900)
def test_kitti_frame_eval_matches_oracle_eager_and_graphed():
    from dcanet_amd import ops
    from dcanet_amd.graph import GraphedHotPath
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    m = load_seeded(GwcNet(192, use_concat_volume=False)).to(DEV).eval()
    fL, fR = seeded_tensor("kitti.fL", (1, 320, 96, 312)), seeded_tensor("kitti.fR", (1, 320, 96, 312))
    sd = O.seeded_state_dict(O.hot_path_shapes(False))
    with torch.no_grad():
        ref = O.hot_path(sd, fL, fR, 192, False)
        got = m.hot_path(fL.to(DEV), fR.to(DEV))
        eager = got["pred4_q"].clone()
        with ops.frozen_weights():
            frozen = m.hot_path(fL.to(DEV), fR.to(DEV))["pred4_q"].clone()
    assert ref["pred4_q"].std() > 1.0, "degenerate test: flat disparity map"
    err = (eager.cpu() - ref["pred4_q"]).abs()
    assert err.max().item() <= 2.5e-4, f"eager: max |pred4_q - oracle| = {err.max().item():.3e} (1e-3 abs at full res)"
    assert (got["prob_volume2"].cpu() - ref["prob_volume2"]).abs().max().item() <= 5e-5 * max(
        1.0, ref["prob_volume2"].abs().max().item())
    assert torch.equal(frozen, eager), "pre-packed weights change the result"
    g = GraphedHotPath(m, fL.to(DEV), fR.to(DEV))
    replay = g(fL.to(DEV), fR.to(DEV))["pred4_q"]
    assert torch.equal(replay, eager), f"graph replay differs by {(replay - eager).abs().max().item():.3e}"


def test_kitti_inference_wrapper_matches_forward():
    """dcanet_amd.inference.KittiInference (my_img.py:47-110) == normalise + pad + GwcNet.forward + crop, with the hot
    path replayed from a hipGraph; second call with other images reuses the graph."""
    from dcanet_amd.inference import KittiInference, crop_back, normalize_pair, pad_or_crop
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    model = torch.nn.DataParallel(load_seeded(GwcNet(32)), device_ids=[0]).cuda()    # my_img.py:34-38
    infer = KittiInference(model, crop_height=64, crop_width=128)
    rng = np.random.default_rng(11)
    for _ in range(2):
        left = rng.integers(0, 256, (50, 100, 3), dtype=np.uint8)
        right = rng.integers(0, 256, (50, 100, 3), dtype=np.uint8)
        disp = infer(left, right)
        assert disp.shape == (50, 100) and disp.dtype == np.float32
        L, R, h, w = pad_or_crop(normalize_pair(left, right), 64, 128)
        model.eval()
        with torch.no_grad():
            want = model(L.cuda(), R.cuda())[0]
        want = crop_back(want.squeeze().cpu().numpy(), h, w, 64, 128)
        assert np.abs(disp - want).max() <= 1e-4, np.abs(disp - want).max()
    assert disp.std() > 0.1


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("variant", ["g", "gc"])
def test_train_step_d192_row_crop_all_gradients(variant, capsys):
    """D=192, features (2,320[+12],34,240) -> 48 x 34 x 240 volumes: train-mode forward (all heads) and backward vs the
    CPU oracle.  Forward heads 2e-5; gradient of EVERY parameter tensor and of both feature maps on relative L2.  The
    per-layer table is printed (pytest -s / captured in the log)."""
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    concat = variant == "gc"
    B, h, w, D = 2, 34, 240, 192
    m = load_seeded(GwcNet(D, use_concat_volume=concat)).to(DEV).train()
    C = 320 + (12 if concat else 0)
    fL, fR = seeded_tensor("crop.fL", (B, C, h, w)), seeded_tensor("crop.fR", (B, C, h, w))
    sd = O.seeded_state_dict(O.hot_path_shapes(concat))
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        sd[k].requires_grad_()
    keys = ["pred0", "pred_dca1", "pred_dca2", "pred1", "pred2", "pred_dca3", "pred4_q"]

    def run(hot, a, b):
        if concat:
            r = hot(a[:, :320], b[:, :320], a[:, 320:], b[:, 320:])
        else:
            r = hot(a, b)
        loss = 0
        for i, k in enumerate(keys):
            loss = loss + (r[k] * seeded_tensor(f"crop.g{i}", r[k].shape).to(r[k].device)).sum()
        return r, loss

    a, b = fL.clone().requires_grad_(), fR.clone().requires_grad_()
    ref, ref_loss = run(lambda *t: O.hot_path(sd, t[0], t[1], D, True, 40, *(t[2:] if concat else (None, None))), a, b)
    ref_g = torch.autograd.grad(ref_loss, [a, b] + [sd[k] for k in names])
    ag, bg = fL.to(DEV).requires_grad_(), fR.to(DEV).requires_grad_()
    got, got_loss = run(m.hot_path, ag, bg)
    params = dict(m.named_parameters())
    got_g = torch.autograd.grad(got_loss, [ag, bg] + [params[k] for k in names])
    for k in keys:
        tol = 2e-5 * max(1.0, ref[k].abs().max().item())
        if k in ("pred_dca3", "pred4_q"):
            tol = 1e-3 if k == "pred_dca3" else 2.5e-4          # north_star: 1e-3 abs on full-res disparities
        err = (got[k].detach().cpu() - ref[k].detach()).abs().max().item()
        assert err <= tol, f"{k}: max err {err:.3e} > {tol:.1e}"
    rows = [(n, rel_l2(g_, r_), r_.norm().item()) for n, g_, r_ in zip(["fL", "fR"] + names, got_g, ref_g)]
    with capsys.disabled():
        print(f"\n[{variant}] per-tensor gradient rel-L2 vs oracle (D=192, 2x{C}x{h}x{w}):")
        for n, e, nr in sorted(rows, key=lambda t: -t[1])[:12]:
            print(f"   {n:60s} {e:.3e}   |ref| {nr:.3e}")
        errs = sorted(e for _, e, _ in rows)
        print(f"   median {errs[len(errs) // 2]:.3e}   max {errs[-1]:.3e}   tensors {len(errs)}")
    errs = sorted(e for _, e, _ in rows)
    # gates: see test_gpu_parity.py::test_golden_hot_path for why end-to-end train-mode gradients are gated on rel-L2
    # gates = ~2x what is measured (round 2: median 9.7e-4 / max 4.1e-3 (G), 1.5e-3 / 3.3e-3 (GC)): a 1 % backward bug in
    # one branch must not pass
    assert errs[len(errs) // 2] <= 3e-3, f"median per-tensor gradient error {errs[len(errs) // 2]:.3e}"
    bad = [(n, e) for n, e, nr in rows if e > 8e-3 and nr > 1e-6]
    assert not bad, bad


@pytest.mark.timeout(900)
def test_wgrad_full_size_batch4_is_deterministic_and_additive():
    """the four weight-gradient kernel families at the BASELINE volume size with per-GPU batch 4: two runs are bitwise
    equal (order-fixed slab reductions), and a batch of 4 copies of one sample gives 4x the single-sample gradient."""
    from dcanet_amd import ops
    d, h, w = 48, 136, 240
    cases = [("3x3x3 s1 32->32 (bf16x3)", 32, 32, (d, h, w), (d, h, w), 3, 1),
             ("3x3x3 s2 32->64 (fp32 MFMA)", 32, 64, (d, h, w), (d // 2, h // 2, w // 2), 3, 2),
             ("1x1x1 64->32", 64, 32, (d, h, w), (d, h, w), 1, 1),
             ("3x3x3 s1 64->64 @1/8 (bf16x3)", 64, 64, (d // 2, h // 2, w // 2), (d // 2, h // 2, w // 2), 3, 1)]
    for name, cx, cy, din, dout, k, s in cases:
        x1 = torch.randn(1, cx, *din, device=DEV)
        dy1 = torch.randn(1, cy, *dout, device=DEV)
        x4, dy4 = x1.expand(4, -1, -1, -1, -1).contiguous(), dy1.expand(4, -1, -1, -1, -1).contiguous()
        K = k ** 3
        def wg(x, dy):
            gw = torch.empty((cy, cx, k, k, k), device=DEV)
            ops._wgrad(x, dy, gw, 0, cx, cy, k, s, cx * K, K)
            return gw
        g4a, g4b, g1 = wg(x4, dy4), wg(x4, dy4), wg(x1, dy1)
        assert torch.equal(g4a, g4b), f"{name}: two runs differ"
        err = ((g4a - 4 * g1).norm() / (4 * g1).norm()).item()
        assert err <= 2e-5, f"{name}: batch-4 gradient is not 4x the single-sample gradient (rel {err:.2e})"
        del x4, dy4


def test_eval_forward_under_inference_mode_equals_no_grad():
    """the f16x2 operand-maxima tags read `t._version`, which inference tensors do not track -- the eval path must
    run (and give the same bits) under torch.inference_mode(), hot path alone and the whole GwcNet.forward."""
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    m = load_seeded(GwcNet(32, use_concat_volume=False)).to(DEV).eval()
    fL, fR = seeded_tensor("imode.fL", (1, 320, 16, 32)).to(DEV), seeded_tensor("imode.fR", (1, 320, 16, 32)).to(DEV)
    left, right = seeded_tensor("imode.l", (1, 3, 32, 64)).to(DEV), seeded_tensor("imode.r", (1, 3, 32, 64)).to(DEV)
    with torch.no_grad():
        a = m.hot_path(fL, fR)["pred4_q"].clone()
        wa = m(left, right)[0].clone()
    with torch.inference_mode():
        b = m.hot_path(fL, fR)["pred4_q"].clone()
        wb = m(left, right)[0].clone()
        fi = fL * 1.0                       # an inference tensor as INPUT
        c = m.hot_path(fi, fR)["pred4_q"].clone()
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(wa, wb)
