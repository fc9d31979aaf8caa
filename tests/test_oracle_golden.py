"""CPU: the oracle restatement must reproduce the reference's own outputs (tests/golden, written by
oracle/make_golden.py from the imported reference).  This is what pins parity."""
import numpy as np
import pytest
import torch

from oracle import dcanet_oracle as O
from oracle.seeded import seeded_tensor, thin

T = lambda a: torch.from_numpy(np.asarray(a))


def close(a, b, tol=2e-5, name=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{name}: max err {err:.3e} (scale {scale:.3e})"


def close_l2(a, b, rel=2e-3, name=""):
    """Relative L2 gate for end-to-end train-mode gradients: ~40 stacked batch-stat BN layers make the
    fp32 gradient sensitive to summation order (the fp64 oracle agrees with the fp32 reference to 7e-4
    max-abs, two fp32 orders differ by up to 2e-2 on isolated elements of magnitude ~16)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert err <= rel, f"{name}: rel L2 err {err:.3e}"


def grads_of(outs, tags, wrt):
    loss = 0
    for o, tag in zip(outs, tags):
        loss = loss + (o * seeded_tensor(tag, o.shape)).sum()
    return torch.autograd.grad(loss, wrt, allow_unused=True)


def prefixed(shapes, p="m"):
    """Standalone reference modules have un-prefixed keys; the oracle API wants a prefix."""
    sd = O.seeded_state_dict(shapes)
    return {f"{p}.{k}": v for k, v in sd.items()}


def bn_shapes(s, p, c):
    s[p + ".weight"] = (c,); s[p + ".bias"] = (c,); s[p + ".running_mean"] = (c,)
    s[p + ".running_var"] = (c,); s[p + ".num_batches_tracked"] = ()


@pytest.mark.parametrize("tag", ["t0", "t1"])
def test_volumes(golden, tag):
    g = golden(f"volumes_{tag}")
    B, C, G, H, W, D, cc = [int(v) for v in g["shape"]]
    L = seeded_tensor(f"vol.{tag}.L", (B, C, H, W)).requires_grad_()
    R = seeded_tensor(f"vol.{tag}.R", (B, C, H, W)).requires_grad_()
    close(L[0, 0, 0, :4].detach(), g["L_fp"], 0, "input fingerprint")
    v = O.build_gwc_volume(L, R, D, G)
    close(v, g["gwc"], 1e-6, "gwc")
    gL, gR = grads_of([v], [f"vol.{tag}.gv"], [L, R])
    close(gL, g["gL"], 1e-6, "gL"); close(gR, g["gR"], 1e-6, "gR")
    cL, cR = L[:, :cc].detach().clone().requires_grad_(), R[:, :cc].detach().clone().requires_grad_()
    cv = O.build_concat_volume(cL, cR, D)
    close(cv, g["concat"], 0, "concat")
    gcL, gcR = grads_of([cv], [f"vol.{tag}.gcv"], [cL, cR])
    close(gcL, g["gcL"], 1e-6); close(gcR, g["gcR"], 1e-6)


def test_regression(golden):
    g = golden("regression")
    close(O.disparity_regression(T(g["p"]), 8), g["disp"], 1e-6)


@pytest.mark.parametrize("tag", ["t0", "t1"])
def test_context_inject(golden, tag):
    g = golden(f"context_inject_{tag}")
    shp = tuple(int(v) for v in g["shape"])
    x = seeded_tensor(f"inj.{tag}.x", shp).requires_grad_()
    preds = (seeded_tensor(f"inj.{tag}.p", (shp[0],) + shp[2:]) * 1.5).requires_grad_()
    key, kstar, _ = O.context_inject(x, preds)
    assert (kstar.numpy() == g["kstar"]).all()
    close(key, g["key"], 1e-6, "key")
    gx, gp = grads_of([key], [f"inj.{tag}.g"], [x, preds])
    close(gx, g["gx"], 1e-6, "gx"); close(gp, g["gp"], 2e-5, "gp")


def attention_shapes():
    s = {}
    for proj in ("key_project", "query_project"):
        for j in (0, 1):
            s[f"{proj}.{j}.0.weight"] = (32, 32, 1, 1, 1); bn_shapes(s, f"{proj}.{j}.1", 32)
    for proj in ("value_project", "out_project"):
        s[f"{proj}.0.weight"] = (32, 32, 1, 1, 1); bn_shapes(s, f"{proj}.1", 32)
    return s


@pytest.mark.parametrize("training", [False, True])
def test_attention(golden, training):
    g = golden(f"attention_{'train' if training else 'eval'}")
    sd = prefixed(attention_shapes())
    for k in ("m.query_project.0.0.weight", "m.value_project.0.weight", "m.out_project.1.weight",
              "m.key_project.1.1.bias"):
        sd[k].requires_grad_()
    shp = tuple(int(v) for v in g["shape"])
    q = seeded_tensor("att.q", shp).requires_grad_()
    k = seeded_tensor("att.k", shp).requires_grad_()
    out = O.self_attention_block(sd, "m", q, k, training)
    close(out, g["out"], 2e-5, "out")
    gr = grads_of([out], ["att.g"], [q, k, sd["m.query_project.0.0.weight"], sd["m.value_project.0.weight"],
                                     sd["m.out_project.1.weight"], sd["m.key_project.1.1.bias"]])
    for got, name in zip(gr, ["gq", "gk", "g_qp00w", "g_vp0w", "g_op1w", "g_kp11b"]):
        close(thin(got) if name.startswith("g_") else got, g[name], 5e-5, name)
    close(sd["m.query_project.0.1.running_mean"], g["rm_after"], 1e-6, "running_mean")


def cva_shapes():
    full = O.hot_path_shapes(False)
    return {k[len("cva1."):]: v for k, v in full.items() if k.startswith("cva1.")}


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("tag", ["t0", "t1"])
def test_cva(golden, tag, training):
    g = golden(f"cva_{tag}_{'train' if training else 'eval'}")
    sd = prefixed(cva_shapes())
    names = ["m.downsample.1.0.weight", "m.classify.2.weight", "m.fuse.0.0.weight", "m.cost_agg.conv1.0.0.weight",
             "m.cost_agg.conv3.0.weight", "m.cost_agg.conv3.1.weight", "m.cost_agg.redir.1.bias",
             "m.slc_net.cross_attention.key_project.0.0.weight"]
    for k in names:
        sd[k].requires_grad_()
    shp = tuple(int(v) for v in g["shape"])
    x = seeded_tensor(f"cva.{tag}.x", shp).requires_grad_()
    prob, aug = O.cva(sd, "m", x, training)
    close(prob, g["prob"], 2e-5, "prob"); close(aug, g["aug"], 2e-5, "aug")
    gr = grads_of([prob, aug], [f"cva.{tag}.gprob", f"cva.{tag}.gaug"], [x] + [sd[k] for k in names])
    gn = ["gx", "g_down_w", "g_cls2_w", "g_fuse_w", "g_agg1_w", "g_agg3_w", "g_agg3_bnw", "g_redir_bnb", "g_kp00_w"]
    for got, name in zip(gr, gn):
        close(thin(got) if name.startswith("g_") else got, g[name], 1e-4, name)
    close(sd["m.cost_agg.conv3.1.running_var"], g["rv_after"], 1e-5)


def magg_shapes():
    s = {}
    s["conv1.0.0.weight"] = (64, 32, 3, 3, 3); bn_shapes(s, "conv1.0.1", 64)
    s["conv2.0.0.weight"] = (64, 64, 3, 3, 3); bn_shapes(s, "conv2.0.1", 64)
    s["conv3.0.weight"] = (64, 32, 3, 3, 3); bn_shapes(s, "conv3.1", 32)
    s["redir.0.weight"] = (32, 32, 1, 1, 1); bn_shapes(s, "redir.1", 32)
    return s


@pytest.mark.parametrize("training", [False, True])
def test_multi_agg_hourglass(golden, training):
    t = "train" if training else "eval"
    g = golden(f"multi_agg_{t}")
    sd = prefixed(magg_shapes())
    names = ["m.conv3.0.weight", "m.conv1.0.0.weight", "m.redir.0.weight"]
    for k in names:
        sd[k].requires_grad_()
    x = seeded_tensor("magg.x", (2, 32, 4, 6, 10)).requires_grad_()
    y = O.multi_aggregation(sd, "m", x, training)
    close(y, g["y"], 2e-5)
    gr = grads_of([y], ["magg.g"], [x] + [sd[k] for k in names])
    for got, name in zip(gr, ["gx", "g_w3", "g_w1", "g_wr"]):
        close(thin(got) if name.startswith("g_") else got, g[name], 1e-4, name)
    g = golden(f"hourglass_{t}")
    shapes = {k[2:]: v for k, v in O.hourglass_shapes("h").items()}
    sd = prefixed(shapes)
    names = ["m.conv5.0.weight", "m.conv3.0.0.weight"]
    for k in names:
        sd[k].requires_grad_()
    x = seeded_tensor("hg.x", (1, 32, 8, 8, 12)).requires_grad_()
    y = O.hourglass(sd, "m", x, training)
    close(y, g["y"], 2e-5)
    gr = grads_of([y], ["hg.g"], [x] + [sd[k] for k in names])
    for got, name in zip(gr, ["gx", "g_w5", "g_w3"]):
        close(thin(got) if name.startswith("g_") else got, g[name], 1e-4, name)


@pytest.mark.parametrize("variant", ["g", "gc"])
@pytest.mark.parametrize("training", [False, True])
def test_hot_path(golden, variant, training):
    concat = variant == "gc"
    g = golden(f"hot_path_{variant}_{'train' if training else 'eval'}")
    sd = O.seeded_state_dict(O.hot_path_shapes(concat))
    C = 320 + (12 if concat else 0)
    fL = seeded_tensor("hot.fL", (2, C, 16, 32)).requires_grad_()
    fR = seeded_tensor("hot.fR", (2, C, 16, 32)).requires_grad_()
    names = ["dres0.0.0.weight", "dres1.2.1.weight", "cva2.cost_agg.conv3.0.weight",
             "cva1.slc_net.cross_attention.query_project.0.0.weight", "classif3.2.weight", "cva3.fuse.0.1.bias",
             "classif1.0.0.weight"]
    for k in names:
        sd[k].requires_grad_()
    cL = cR = None
    gl, gr_ = fL, fR
    if concat:
        gl, cL, gr_, cR = fL[:, :320], fL[:, 320:], fR[:, :320], fR[:, 320:]
    r = O.hot_path(sd, gl, gr_, 32, training, 40, cL, cR)
    close(r["pred4_q"], g["pred4_q"], 2e-5, "pred4_q")
    if training:
        keys = ["pred0", "pred_dca1", "pred_dca2", "pred1", "pred2", "pred_dca3", "pred4_q"]
        for k in keys:
            close(r[k], g[k], 2e-5, k)
        outs = [r[k] for k in keys]
        gr = grads_of(outs, [f"hot.g{i}" for i in range(7)], [fL, fR] + [sd[k] for k in names])
        close_l2(gr[0][:, ::16], g["gfL"], 2e-3, "gfL"); close_l2(gr[1][:, ::16], g["gfR"], 2e-3, "gfR")
        gn = ["g_dres0_w", "g_dres1_bn2_w", "g_cva2_deconv_w", "g_cva1_q00_w", "g_cls3_w", "g_cva3_fuse_bnb", "g_cls1_w"]
        for got, name in zip(gr[2:], gn):
            # a BN bias that feeds a batch-stat BN is almost exactly cancelled -> tiny, noisy gradient
            close_l2(thin(got), g[name], 1e-2 if name.endswith("bnb") else 2e-3, name)
        close(sd["dres0.0.1.running_mean"], g["rm_dres0"], 1e-5)
    else:
        close(r["prob_volume2"].squeeze(1), g["prob_volume2"], 2e-5, "prob_volume2")
        gr = grads_of([r["pred4_q"]], ["hot.g_eval"], [fL, fR])
        close(gr[0][:, ::16], g["gfL"], 2e-4); close(gr[1][:, ::16], g["gfR"], 2e-4)


def test_baseline_gwcnet(golden):
    """baseline gwcnet.GwcNet training branch (three stacked hourglasses), reference models/gwcnet.py:194-238"""
    g = golden("baseline_g_train")
    sd = O.seeded_state_dict(O.baseline_shapes())
    names = ["dres2.conv5.0.weight", "dres3.conv3.0.0.weight", "dres4.redir2.0.weight", "dres2.conv4.0.1.weight"]
    for k in names:
        sd[k].requires_grad_()
    fL = seeded_tensor("base.fL", (2, 320, 16, 32)).requires_grad_()
    fR = seeded_tensor("base.fR", (2, 320, 16, 32)).requires_grad_()
    preds = O.hot_path_baseline(sd, fL, fR, 32)
    for i, p in enumerate(preds):
        close(p, g[f"pred{i}"], 2e-5, f"pred{i}")
    gr = grads_of(preds, [f"base.g{i}" for i in range(4)], [fL, fR] + [sd[k] for k in names])
    close_l2(gr[0][:, ::16], g["gfL"], 2e-3, "gfL"); close_l2(gr[1][:, ::16], g["gfR"], 2e-3, "gfR")
    for got, name in zip(gr[2:], ["g_d2c5_w", "g_d3c3_w", "g_d4r2_w", "g_d2c4_bnw"]):
        close_l2(thin(got), g[name], 2e-3, name)


# ------------------------------------------------------------------ the real boundary, whole model
def _whole_sd(variant):
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        shapes = json.load(f)[variant]
    return O.seeded_state_dict({k: tuple(v) for k, v in shapes.items()})


WHOLE_GRAD_KEYS = ["feature_extraction.firstconv.0.0.weight", "feature_extraction.layer4.2.conv2.0.weight",
                   "guidance.conv_start.0.weight", "guidance.guidance.weight", "prop.conv.2.weight",
                   "prop.conv.0.1.bias", "dres0.0.0.weight", "cva2.cost_agg.conv3.0.weight"]
WHOLE_GRAD_NAMES = ["g_fe_first_w", "g_fe_l4_w", "g_guid_start_w", "g_guid_out_w", "g_prop_w", "g_prop_bnb", "g_dres0_w",
                    "g_cva2_deconv_w"]
WHOLE_GRAD_STRIDE = [1, 8, 1, 4, 8, 1, 1, 1]


@pytest.mark.parametrize("variant", ["g", "gc"])
@pytest.mark.parametrize("training", [False, True])
def test_whole_model(golden, variant, training):
    """GwcNet.forward(left, right, disp_true) of the reference (gwcnet_dca_g.py:209-282) on (1,3,64,128): 2D extractor,
    Guidance, hot path, convex up-sampler -- oracle restatement vs the reference's own outputs."""
    g = golden(f"whole_{variant}_{'train' if training else 'eval'}")
    sd = _whole_sd(variant)
    for k in WHOLE_GRAD_KEYS:
        sd[k].requires_grad_()
    L = seeded_tensor("whole.left", (1, 3, 64, 128)).requires_grad_()
    R = seeded_tensor("whole.right", (1, 3, 64, 128)).requires_grad_()
    out = O.whole_model(sd, L, R, 32, training)
    aux = out[2]
    close(aux["gwc_feature"][:, ::16], g["gwc_feature"], 2e-5, "gwc_feature")
    close(aux["guidance"][:, ::8], g["guidance"], 2e-5, "guidance")
    if training:
        probs, disps = out[0], out[1]
        for k, v in zip(["pred0", "pred_dca1", "pred_dca2", "pred1", "pred2"], probs):
            close(v, g[k], 2e-5, k)
        close(disps[0], g["pred_dca3"], 2e-5, "pred_dca3")
        close(disps[1], g["pred4"], 2e-5, "pred4")
        gr = grads_of(list(probs) + list(disps), [f"whole.g{i}" for i in range(7)], [L, R] + [sd[k] for k in WHOLE_GRAD_KEYS])
        close_l2(gr[0], g["gL"], 5e-3, "gL"); close_l2(gr[1], g["gR"], 5e-3, "gR")
        for got, name, st in zip(gr[2:], WHOLE_GRAD_NAMES, WHOLE_GRAD_STRIDE):
            close_l2(thin(got[::st]), g[name], 1e-2 if name.endswith("bnb") else 5e-3, name)
        close(sd["feature_extraction.firstconv.0.1.running_mean"], g["rm_fe_first"], 1e-5, "running_mean")
        close(sd["prop.conv.0.1.running_var"], g["rv_prop"], 1e-5, "running_var")
    else:
        close(out[0], g["pred4"], 2e-5, "pred4")
        close(out[1], g["prob_volume2"], 2e-5, "prob_volume2")
        gr = grads_of([out[0]], ["whole.g_eval"], [L, R])
        close(gr[0], g["gL"], 2e-4, "gL"); close(gr[1], g["gR"], 2e-4, "gR")


@pytest.mark.parametrize("training", [False, True])
def test_guidance_and_convex_upsampler(golden, training):
    """Guidance (submodule.py:395-460) and PropgationNet_4x (submodule.py:357-373) alone."""
    g = golden(f"guidance_prop_{'train' if training else 'eval'}")
    sd = _whole_sd("g")
    sd = {k: v.clone() for k, v in sd.items() if k.startswith(("guidance.", "prop."))}
    for k in ("guidance.conv_start.0.weight", "guidance.layer2.0.downsample.0.bias", "prop.conv.2.weight"):
        sd[k].requires_grad_()
    x = seeded_tensor("guid.x", (2, 3, 32, 64)).requires_grad_()
    gout = O.guidance(sd, x, training)
    close(gout[:, ::4], g["g"], 2e-5, "g")
    gg = grads_of([gout], ["guid.g"], [x, sd["guidance.conv_start.0.weight"], sd["guidance.layer2.0.downsample.0.bias"]])
    close_l2(gg[0], g["g_x"], 1e-3, "g_x"); close_l2(gg[1], g["g_start_w"], 1e-3, "g_start_w")
    if not training:     # under batch statistics a bias in front of a BN has a (numerically noisy) ~zero gradient
        close_l2(gg[2], g["g_ds_b"], 1e-3, "g_ds_b")
    gd = seeded_tensor("prop.guid", (2, 64, 6, 10)).requires_grad_()
    disp = (seeded_tensor("prop.disp", (2, 1, 6, 10)) * 2 + 5).requires_grad_()
    up = O.prop(sd, gd, disp, training)
    close(up, g["up"], 2e-5, "up")
    gp = grads_of([up], ["prop.g"], [gd, disp, sd["prop.conv.2.weight"]])
    close_l2(gp[0], g["gp_guid"], 1e-3, "gp_guid"); close_l2(gp[1], g["gp_disp"], 1e-4, "gp_disp")
    close_l2(gp[2][::8], g["gp_w"], 1e-3, "gp_w")


def test_losses(golden):
    """oracle focal_loss / model_loss vs the reference's (models/loss.py), incl. gradients and the sparse variant."""
    g = golden("losses")
    gt = T(g["gt"])
    ests = [torch.softmax(seeded_tensor(f"loss.e{i}", (2, 8, 8, 16)), 1).requires_grad_() for i in range(5)]
    fl = O.focal_loss(ests, gt, 32, 5.0, False)
    close(fl, g["focal"], 1e-5, "focal")
    close(O.focal_loss(ests, gt, 32, 5.0, True), g["focal_sparse"], 1e-5, "focal_sparse")
    ge = torch.autograd.grad(fl, ests)
    close(ge[0], g["gfocal0"], 1e-5, "gfocal0"); close(ge[4], g["gfocal4"], 1e-5, "gfocal4")
    d0 = (seeded_tensor("loss.d0", (2, 1, 32, 64)) * 3 + gt).requires_grad_()
    d1 = (seeded_tensor("loss.d1", (2, 1, 32, 64)) * 0.3 + gt).requires_grad_()
    ml = O.model_loss([d0, d1], gt, (gt < 32) & (gt > 0))
    close(ml, g["model"], 1e-5, "model")
    gd = torch.autograd.grad(ml, [d0, d1])
    close(gd[0], g["gd0"], 1e-5, "gd0"); close(gd[1], g["gd1"], 1e-5, "gd1")
