"""GPU parity at the REAL boundary: `GwcNet.forward(left, right[, disp_true])` (reference gwcnet_dca_g.py:209-282)
against outputs of the reference itself (tests/golden/whole_*.npz, written by oracle/make_golden.py::gen_whole_model),
two- and three-argument calls, train and eval return values, and the `nn.DataParallel(model, device_ids=[0]).cuda()`
+ `module.`-prefixed `load_state_dict(strict=True)` usage of main_dca.py:54-61,131,169.

Gates: full-res `pred4` 1e-3 abs (north_star), `prob_volume2` 2e-5, probabilities 2e-5, train-mode gradients on
relative L2 (see tests/test_gpu_parity.py::test_golden_hot_path for why those are looser)."""
import json
import os

import pytest
import torch

from oracle import dcanet_oracle as O
from oracle.seeded import seeded_tensor, thin

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, tol=2e-5, name="", abs_tol=None):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = abs_tol if abs_tol is not None else tol * max(1.0, b.abs().max().item())
    assert err <= lim, f"{name}: max err {err:.3e} (limit {lim:.3e})"


def close_l2(a, b, rel, name=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert err <= rel, f"{name}: rel L2 err {err:.3e}"


def grads_of(outs, tags, wrt):
    loss = 0
    for o, tag in zip(outs, tags):
        loss = loss + (o * seeded_tensor(tag, o.shape).to(o.device)).sum()
    return torch.autograd.grad(loss, wrt, allow_unused=True)


def whole_sd(variant):
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        shapes = json.load(f)[variant]
    return O.seeded_state_dict({k: tuple(v) for k, v in shapes.items()})


def make_model(variant, training):
    import dcanet_amd  # noqa: F401
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    m = GwcNet(32, use_concat_volume=(variant == "gc"))
    m.load_state_dict(whole_sd(variant), strict=True)
    return m.to(DEV).train(training)


def images(grad=False):
    L = seeded_tensor("whole.left", (1, 3, 64, 128)).to(DEV).requires_grad_(grad)
    R = seeded_tensor("whole.right", (1, 3, 64, 128)).to(DEV).requires_grad_(grad)
    return L, R


@pytest.mark.parametrize("variant", ["g", "gc"])
def test_forward_eval_matches_reference(golden, variant):
    """eval: `pred4, prob_volume2.squeeze(1) = model(left, right)` (gwcnet_dca_g.py:282; main_dca.py:169 two-arg call),
    under no_grad (fused inference kernels) and with autograd (unfused kernels + input gradients)."""
    g = golden(f"whole_{variant}_eval")
    m = make_model(variant, False)
    L, R = images()
    with torch.no_grad():
        pred4, prob2 = m(L, R)
        three = m(L, R, None)
    assert pred4.shape == (1, 1, 64, 128) and prob2.shape == (1, 4, 8, 16)
    close(pred4, g["pred4"], name="pred4 (1e-3 abs, north_star)", abs_tol=1e-3)
    close(prob2, g["prob_volume2"], 2e-5, "prob_volume2")
    # (not bitwise: MIOpen may settle on another 2D-conv algorithm after its first-call search)
    close(three[0], pred4, 1e-5, "3-arg call vs 2-arg call"); close(three[1], prob2, 1e-5, "3-arg call vs 2-arg call")
    L, R = images(True)
    pred4g, prob2g = m(L, R)
    close(pred4g, g["pred4"], name="pred4 (autograd path)", abs_tol=1e-3)
    close(prob2g, g["prob_volume2"], 2e-5, "prob_volume2 (autograd path)")
    gr = grads_of([pred4g], ["whole.g_eval"], [L, R])
    # image gradients cross ~80 MIOpen 2D layers (Winograd fp32 convolutions) plus the hot path: measured 1-4e-3
    close_l2(gr[0], g["gL"], 1e-2, "gL"); close_l2(gr[1], g["gR"], 1e-2, "gR")


GRAD_NAMES = ["g_fe_first_w", "g_fe_l4_w", "g_guid_start_w", "g_guid_out_w", "g_prop_w", "g_prop_bnb", "g_dres0_w",
              "g_cva2_deconv_w"]
GRAD_STRIDE = [1, 8, 1, 4, 8, 1, 1, 1]


def grad_params(m):
    return [m.feature_extraction.firstconv[0][0].weight, m.feature_extraction.layer4[2].conv2[0].weight,
            m.guidance.conv_start[0].weight, m.guidance.guidance.weight, m.prop.conv[2].weight, m.prop.conv[0][1].bias,
            m.dres0[0][0].weight, m.cva2.cost_agg.conv3[0].weight]


@pytest.mark.parametrize("variant", ["g", "gc"])
def test_forward_train_matches_reference(golden, variant):
    """train: `[pred0,pred_dca1,pred_dca2,pred1,pred2], [pred_dca3,pred4] = model(left, right)` (gwcnet_dca_g.py:277-278,
    main_dca.py:131) with batch-statistic BN everywhere, backward to the images and to parameters of every sub-network,
    running-stat updates of 2D and 3D BatchNorm."""
    g = golden(f"whole_{variant}_train")
    m = make_model(variant, True)
    L, R = images(True)
    probs, disps = m(L, R)
    assert len(probs) == 5 and len(disps) == 2
    for k, v in zip(["pred0", "pred_dca1", "pred_dca2", "pred1", "pred2"], probs):
        assert v.shape == (1, 8, 16, 32)
        close(v, g[k], 5e-5, k)
    close(disps[0], g["pred_dca3"], name="pred_dca3", abs_tol=1e-3)
    close(disps[1], g["pred4"], name="pred4 (1e-3 abs)", abs_tol=1e-3)
    gr = grads_of(list(probs) + list(disps), [f"whole.g{i}" for i in range(7)], [L, R] + grad_params(m))
    # end-to-end train-mode gradients through ~100 batch-stat BN layers (2D backbone on MIOpen + the HIP hot path):
    # relative-L2 gates, see test_gpu_parity.py::test_golden_hot_path
    close_l2(gr[0], g["gL"], 2e-2, "gL"); close_l2(gr[1], g["gR"], 2e-2, "gR")
    for got, name, st in zip(gr[2:], GRAD_NAMES, GRAD_STRIDE):
        close_l2(thin(got[::st]), g[name], 3e-2 if name.endswith("bnb") else 2e-2, name)
    close(m.feature_extraction.firstconv[0][1].running_mean, g["rm_fe_first"], 1e-5, "running_mean (2D BN, two calls)")
    close(m.prop.conv[0][1].running_var, g["rv_prop"], 1e-5, "running_var")
    assert int(m.dres0[0][1].num_batches_tracked) == int(g["nbt"]) == 1


def test_dataparallel_wrapper_and_prefixed_checkpoint(golden, tmp_path):
    """main_dca.py:54-61: `model = nn.DataParallel(model, device_ids=[0]); model.cuda();
    model.load_state_dict(torch.load(ckpt)['state_dict'], strict=True)` with `module.`-prefixed keys, then
    `model(imgL, imgR)` in eval (main_dca.py:169) and train (main_dca.py:131) mode; the checkpoint is written the way
    main_dca.py:275-281 writes it."""
    import dcanet_amd  # noqa: F401
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    g = golden("whole_gc_eval")
    ckpt = str(tmp_path / "checkpoint_0.tar")
    torch.save({"epoch": 0, "state_dict": {"module." + k: v for k, v in whole_sd("gc").items()}, "train_loss": 0.0}, ckpt)
    model = torch.nn.DataParallel(GwcNet(32), device_ids=[0])      # main_dca.py:53 builds the GC default
    model.cuda()
    state = torch.load(ckpt, weights_only=True)
    model.load_state_dict(state["state_dict"], strict=True)
    assert sorted(model.state_dict().keys()) == sorted(state["state_dict"].keys())
    L, R = images()
    model.eval()
    with torch.no_grad():
        pred4, prob2 = model(L, R)
    close(pred4, g["pred4"], name="pred4 through DataParallel", abs_tol=1e-3)
    close(prob2, g["prob_volume2"], 2e-5, "prob_volume2 through DataParallel")
    gt = golden("whole_gc_train")
    model.train()
    probs, disps = model(L, R)
    close(disps[1], gt["pred4"], name="train pred4 through DataParallel", abs_tol=1e-3)
    close(probs[4], gt["pred2"], 5e-5, "pred2 through DataParallel")
    (sum(d.mean() for d in disps) + sum(p.square().sum() for p in probs)).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("training", [False, True])
def test_guidance_and_convex_upsampler(golden, training):
    """`Guidance` (submodule.py:395-460) and `PropgationNet_4x` (submodule.py:357-373) as built here, on
    the GPU, vs the reference's outputs and gradients."""
    from dcanet_amd.models.submodule import Guidance, PropgationNet_4x
    g = golden(f"guidance_prop_{'train' if training else 'eval'}")
    sd = whole_sd("g")
    gnet = Guidance(64)
    gnet.load_state_dict({k[len("guidance."):]: v for k, v in sd.items() if k.startswith("guidance.")}, strict=True)
    gnet = gnet.to(DEV).train(training)
    x = seeded_tensor("guid.x", (2, 3, 32, 64)).to(DEV).requires_grad_()
    gout = gnet(x)["g"]
    close(gout[:, ::4], g["g"], 5e-5, "g")
    gg = grads_of([gout], ["guid.g"], [x, gnet.conv_start[0].weight])
    close_l2(gg[0], g["g_x"], 2e-3, "g_x"); close_l2(gg[1], g["g_start_w"], 2e-3, "g_start_w")
    prop = PropgationNet_4x(64)
    prop.load_state_dict({k[len("prop."):]: v for k, v in sd.items() if k.startswith("prop.")}, strict=True)
    prop = prop.to(DEV).train(training)
    gd = seeded_tensor("prop.guid", (2, 64, 6, 10)).to(DEV).requires_grad_()
    disp = (seeded_tensor("prop.disp", (2, 1, 6, 10)) * 2 + 5).to(DEV).requires_grad_()
    up = prop(gd, disp)
    close(up, g["up"], 2e-5, "up")
    gp = grads_of([up], ["prop.g"], [gd, disp, prop.conv[2].weight])
    close_l2(gp[0], g["gp_guid"], 2e-3, "gp_guid"); close_l2(gp[1], g["gp_disp"], 1e-4, "gp_disp")
    close_l2(gp[2][::8], g["gp_w"], 2e-3, "gp_w")
