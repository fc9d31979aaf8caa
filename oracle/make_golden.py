"""Generate golden vectors by running the REFERENCE itself (build container only).

Imports /root/reference with the stub-package recipe of SURVEY.md 8(c) (skips
`models/__init__.py`, which imports a module whose source is not in the tree), loads the
key-seeded Appendix-D weights `strict=True` into the reference modules, runs them on seeded
inputs and stores inputs' fingerprints + outputs + gradients under tests/golden/*.npz.

Only DATA leaves this script; no reference source travels.  Run:
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python oracle/make_golden.py
"""
import os
import sys
import types
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)
_pkg = types.ModuleType("models")
_pkg.__path__ = [os.path.join(REF, "models")]
sys.modules["models"] = _pkg

import models.submodule as ref_sub            # noqa: E402
import models.gwcnet_dca_g as ref_dca         # noqa: E402
import models.gwcnet as ref_gwc               # noqa: E402
import models.augment.cva as ref_cva          # noqa: E402
import models.augment.semantic_level as ref_slc   # noqa: E402
import models.augment.SelfAttention_bn as ref_att  # noqa: E402

from oracle import dcanet_oracle as O         # noqa: E402
from oracle.seeded import seeded_tensor, thin       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def load_seeded(module: nn.Module, prefix_filter=None):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = O.seeded_state_dict(shapes)
    module.load_state_dict(sd, strict=True)
    return sd


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = (thin(v) if k.startswith("g_") else v).detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB", {k: tuple(v.shape) for k, v in out.items()})


def grads_of(outs, upstream_tags, wrt):
    """loss = sum_i <out_i, seeded(upstream_tag_i)>; returns grads wrt `wrt` tensors."""
    loss = 0
    for o, tag in zip(outs, upstream_tags):
        loss = loss + (o * seeded_tensor(tag, o.shape)).sum()
    return torch.autograd.grad(loss, wrt, allow_unused=True)


# ------------------------------------------------------------------ volumes (a1-a3, a7)
def gen_volumes():
    for tag, (B, C, G, H, W, D) in {"t0": (2, 32, 4, 5, 18, 6), "t1": (1, 320, 40, 3, 10, 12)}.items():
        L = seeded_tensor(f"vol.{tag}.L", (B, C, H, W)).requires_grad_()
        R = seeded_tensor(f"vol.{tag}.R", (B, C, H, W)).requires_grad_()
        v = ref_sub.build_gwc_volume(L, R, D, G)
        gL, gR = grads_of([v], [f"vol.{tag}.gv"], [L, R])
        cc = 12 if C >= 12 else C
        cL, cR = L[:, :cc].detach().clone().requires_grad_(), R[:, :cc].detach().clone().requires_grad_()
        cv = ref_sub.build_concat_volume(cL, cR, D)
        gcL, gcR = grads_of([cv], [f"vol.{tag}.gcv"], [cL, cR])
        npz(f"volumes_{tag}", shape=np.array([B, C, G, H, W, D, cc]), L_fp=L[0, 0, 0, :4], gwc=v, gL=gL, gR=gR,
            concat=cv, gcL=gcL, gcR=gcR)
    # soft-argmin
    x = seeded_tensor("reg.x", (2, 8, 5, 7))
    p = torch.softmax(x, 1)
    npz("regression", p=p, disp=ref_sub.disparity_regression(p, 8))


# ------------------------------------------------------------------ DCA sub-units
class _Capture(nn.Module):
    def forward(self, q, k):
        return k


def gen_context_inject():
    for tag, shp in {"t0": (2, 32, 4, 8, 16), "t1": (1, 32, 6, 5, 9)}.items():
        slc = ref_slc.SemanticLevelContext(32, 32)
        slc.cross_attention = _Capture()
        x = seeded_tensor(f"inj.{tag}.x", shp).requires_grad_()
        preds = (seeded_tensor(f"inj.{tag}.p", (shp[0],) + shp[2:]) * 1.5).requires_grad_()
        key = slc(x, preds)
        gx, gp = grads_of([key], [f"inj.{tag}.g"], [x, preds])
        npz(f"context_inject_{tag}", shape=np.array(shp), key=key, gx=gx, gp=gp,
            kstar=torch.softmax(preds, 1).argmax(1))


def gen_attention():
    for training in (False, True):
        att = ref_slc.SemanticLevelContext(32, 32).cross_attention
        load_seeded(att)
        att.train(training)
        shp = (2, 32, 4, 6, 10)
        q = seeded_tensor("att.q", shp).requires_grad_()
        k = seeded_tensor("att.k", shp).requires_grad_()
        out = att(q, k)
        params = [att.query_project[0][0].weight, att.value_project[0].weight, att.out_project[1].weight,
                  att.key_project[1][1].bias]
        g = grads_of([out], ["att.g"], [q, k] + params)
        npz(f"attention_{'train' if training else 'eval'}", shape=np.array(shp), out=out, gq=g[0], gk=g[1],
            g_qp00w=g[2], g_vp0w=g[3], g_op1w=g[4], g_kp11b=g[5],
            rm_after=att.query_project[0][1].running_mean)


def gen_cva():
    for training in (False, True):
        for tag, shp in {"t0": (2, 32, 8, 8, 16), "t1": (1, 32, 4, 6, 10)}.items():
            m = ref_cva.cva(32, 32)
            load_seeded(m)
            m.train(training)
            x = seeded_tensor(f"cva.{tag}.x", shp).requires_grad_()
            prob, aug = m(x)
            params = [m.downsample[1][0].weight, m.classify[2].weight, m.fuse[0][0].weight,
                      m.cost_agg.conv1[0][0].weight, m.cost_agg.conv3[0].weight, m.cost_agg.conv3[1].weight,
                      m.cost_agg.redir[1].bias, m.slc_net.cross_attention.key_project[0][0].weight]
            g = grads_of([prob, aug], [f"cva.{tag}.gprob", f"cva.{tag}.gaug"], [x] + params)
            npz(f"cva_{tag}_{'train' if training else 'eval'}", shape=np.array(shp), prob=prob, aug=aug, gx=g[0],
                g_down_w=g[1], g_cls2_w=g[2], g_fuse_w=g[3], g_agg1_w=g[4], g_agg3_w=g[5], g_agg3_bnw=g[6],
                g_redir_bnb=g[7], g_kp00_w=g[8], rv_after=m.cost_agg.conv3[1].running_var)


def gen_multi_agg_hourglass():
    for training in (False, True):
        m = ref_cva.Multi_Aggregation(32)
        load_seeded(m)
        m.train(training)
        x = seeded_tensor("magg.x", (2, 32, 4, 6, 10)).requires_grad_()
        y = m(x)
        g = grads_of([y], ["magg.g"], [x, m.conv3[0].weight, m.conv1[0][0].weight, m.redir[0].weight])
        npz(f"multi_agg_{'train' if training else 'eval'}", y=y, gx=g[0], g_w3=g[1], g_w1=g[2], g_wr=g[3])
        h = ref_gwc.hourglass(32)
        load_seeded(h)
        h.train(training)
        x = seeded_tensor("hg.x", (1, 32, 8, 8, 12)).requires_grad_()
        y = h(x)
        g = grads_of([y], ["hg.g"], [x, h.conv5[0].weight, h.conv3[0][0].weight])
        npz(f"hourglass_{'train' if training else 'eval'}", y=y, gx=g[0], g_w5=g[1], g_w3=g[2])


# ------------------------------------------------------------------ hot path end to end
class _FeatStub(nn.Module):
    def __init__(self, concat):
        super().__init__()
        self.concat = concat

    def forward(self, x):
        if self.concat:
            return {"gwc_feature": x[:, :320], "concat_feature": x[:, 320:]}
        return {"gwc_feature": x}


class _GuidStub(nn.Module):
    def forward(self, x):
        return {"g": None}


class _PropStub(nn.Module):
    def forward(self, g, d):
        return d


def gen_hot_path():
    """Runs the reference's own GwcNet.forward body (gwcnet_dca_g.py:209-282) with the 2D nets
    replaced by pass-through stubs, so inputs are the 1/4-res features."""
    for variant, concat in (("g", False), ("gc", True)):
        for training in (False, True):
            D = 32
            m = ref_dca.GwcNet(D, use_concat_volume=concat)
            m.feature_extraction, m.guidance, m.prop = _FeatStub(concat), _GuidStub(), _PropStub()
            sd = load_seeded(m)
            m.train(training)
            C = 320 + (12 if concat else 0)
            fL = seeded_tensor("hot.fL", (2, C, 16, 32)).requires_grad_()
            fR = seeded_tensor("hot.fR", (2, C, 16, 32)).requires_grad_()
            out = m(fL, fR, None)
            stem = f"hot_path_{variant}_{'train' if training else 'eval'}"
            if training:
                probs, disps = out
                outs = list(probs) + list(disps)
                tags = [f"hot.g{i}" for i in range(len(outs))]
                params = [m.dres0[0][0].weight, m.dres1[2][1].weight, m.cva2.cost_agg.conv3[0].weight,
                          m.cva1.slc_net.cross_attention.query_project[0][0].weight, m.classif3[2].weight,
                          m.cva3.fuse[0][1].bias, m.classif1[0][0].weight]
                g = grads_of(outs, tags, [fL, fR] + params)
                npz(stem, pred0=probs[0], pred_dca1=probs[1], pred_dca2=probs[2], pred1=probs[3], pred2=probs[4],
                    pred_dca3=disps[0], pred4_q=disps[1], gfL=g[0][:, ::16], gfR=g[1][:, ::16],
                    g_dres0_w=g[2], g_dres1_bn2_w=g[3], g_cva2_deconv_w=g[4], g_cva1_q00_w=g[5], g_cls3_w=g[6],
                    g_cva3_fuse_bnb=g[7], g_cls1_w=g[8], rm_dres0=m.dres0[0][1].running_mean)
            else:
                pred4, prob2 = out
                g = grads_of([pred4], ["hot.g_eval"], [fL, fR])
                npz(stem, pred4_q=pred4, prob_volume2=prob2, gfL=g[0][:, ::16], gfR=g[1][:, ::16])


# ------------------------------------------------------------------ whole model: GwcNet.forward(left, right, disp_true)
def gen_whole_model():
    """The real boundary: the reference's `GwcNet(32, G / GC)` -- 2D feature extractor, Guidance,
    hot path, convex up-sampler -- on a (1,3,64,128) pair with Appendix-D weights (incl. the residual-gamma x0.25 rule),
    eval tuple and train lists (gwcnet_dca_g.py:209-282), plus `Guidance` and `PropgationNet_4x` alone."""
    left = seeded_tensor("whole.left", (1, 3, 64, 128))
    right = seeded_tensor("whole.right", (1, 3, 64, 128))
    for variant, concat in (("g", False), ("gc", True)):
        for training in (False, True):
            m = ref_dca.GwcNet(32, use_concat_volume=concat)
            load_seeded(m)
            m.train(training)
            stem = f"whole_{variant}_{'train' if training else 'eval'}"
            L = left.clone().requires_grad_()
            R = right.clone().requires_grad_()
            feat = m.feature_extraction(L)["gwc_feature"].detach()
            guid = m.guidance(L)["g"].detach()
            if training:
                # the two probe calls above moved the BN running stats: reload, so the fixture = one clean forward
                load_seeded(m)
                m.train(True)
            out = m(L, R, None)
            if training:
                probs, disps = out
                outs = list(probs) + list(disps)
                tags = [f"whole.g{i}" for i in range(len(outs))]
                params = [m.feature_extraction.firstconv[0][0].weight, m.feature_extraction.layer4[2].conv2[0].weight,
                          m.guidance.conv_start[0].weight, m.guidance.guidance.weight, m.prop.conv[2].weight,
                          m.prop.conv[0][1].bias, m.dres0[0][0].weight, m.cva2.cost_agg.conv3[0].weight]
                g = grads_of(outs, tags, [L, R] + params)
                npz(stem, pred0=probs[0], pred_dca1=probs[1], pred_dca2=probs[2], pred1=probs[3], pred2=probs[4],
                    pred_dca3=disps[0], pred4=disps[1], gwc_feature=feat[:, ::16], guidance=guid[:, ::8],
                    gL=g[0], gR=g[1], g_fe_first_w=g[2], g_fe_l4_w=g[3][::8], g_guid_start_w=g[4], g_guid_out_w=g[5][::4],
                    g_prop_w=g[6][::8], g_prop_bnb=g[7], g_dres0_w=g[8], g_cva2_deconv_w=g[9],
                    rm_fe_first=m.feature_extraction.firstconv[0][1].running_mean,
                    rv_prop=m.prop.conv[0][1].running_var, nbt=m.dres0[0][1].num_batches_tracked)
            else:
                pred4, prob2 = out
                g = grads_of([pred4], ["whole.g_eval"], [L, R])
                npz(stem, pred4=pred4, prob_volume2=prob2, gwc_feature=feat[:, ::16], guidance=guid[:, ::8],
                    gL=g[0], gR=g[1])
    # Guidance and the convex up-sampler alone (submodule.py:395-460, 357-373)
    for training in (False, True):
        gnet = ref_sub.Guidance(64)
        sd = O.seeded_state_dict({"guidance." + k: tuple(v.shape) for k, v in gnet.state_dict().items()})
        gnet.load_state_dict({k[len("guidance."):]: v for k, v in sd.items()}, strict=True)
        gnet.train(training)
        x = seeded_tensor("guid.x", (2, 3, 32, 64)).requires_grad_()
        gout = gnet(x)["g"]
        gg = grads_of([gout], ["guid.g"], [x, gnet.conv_start[0].weight, gnet.layer2[0].downsample[0].bias])
        prop = ref_sub.PropgationNet_4x(64)
        sd = O.seeded_state_dict({"prop." + k: tuple(v.shape) for k, v in prop.state_dict().items()})
        prop.load_state_dict({k[len("prop."):]: v for k, v in sd.items()}, strict=True)
        prop.train(training)
        gd = seeded_tensor("prop.guid", (2, 64, 6, 10)).requires_grad_()
        disp = (seeded_tensor("prop.disp", (2, 1, 6, 10)) * 2 + 5).requires_grad_()
        up = prop(gd, disp)
        gp = grads_of([up], ["prop.g"], [gd, disp, prop.conv[2].weight])
        npz(f"guidance_prop_{'train' if training else 'eval'}", g=gout[:, ::4], g_x=gg[0], g_start_w=gg[1], g_ds_b=gg[2],
            up=up, gp_guid=gp[0], gp_disp=gp[1], gp_w=gp[2][::8])


def gen_init():
    """a10: the reference's weight init (gwcnet_dca_g.py:173-185) under a fixed torch seed: per-key fingerprints
    (sum, abs-sum, first 4 values) of the freshly constructed model's state dict."""
    out = {}
    for variant, concat in (("g", False), ("gc", True)):
        torch.manual_seed(1234)
        m = ref_dca.GwcNet(64, use_concat_volume=concat)
        keys = sorted(m.state_dict().keys())
        fp = []
        for k in keys:
            v = m.state_dict()[k].double().flatten()
            head = torch.zeros(4, dtype=torch.float64)
            head[:min(4, v.numel())] = v[:4]
            fp.append(torch.cat([v.sum().view(1), v.abs().sum().view(1), head]))
        out[f"{variant}_fp"] = torch.stack(fp)
    npz("init_fingerprint", **out)


def gen_baseline():
    """Baseline gwcnet.GwcNet (3 stacked hourglasses), training branch, from the 1/4-res features."""
    m = ref_gwc.GwcNet(32, use_concat_volume=False)
    m.feature_extraction = _FeatStub(False)
    load_seeded(m)
    m.train()
    fL = seeded_tensor("base.fL", (2, 320, 16, 32)).requires_grad_()
    fR = seeded_tensor("base.fR", (2, 320, 16, 32)).requires_grad_()
    left = torch.zeros(2, 3, 64, 128)       # only its size is read (gwcnet.py:219)
    m.feature_extraction = _FeatByShape(fL, fR)
    preds = m(left, left.clone() + 1, None)
    params = [m.dres2.conv5[0].weight, m.dres3.conv3[0][0].weight, m.dres4.redir2[0].weight, m.dres2.conv4[0][1].weight]
    g = grads_of(preds, [f"base.g{i}" for i in range(4)], [fL, fR] + params)
    npz("baseline_g_train", pred0=preds[0], pred1=preds[1], pred2=preds[2], pred3=preds[3], gfL=g[0][:, ::16],
        gfR=g[1][:, ::16], g_d2c5_w=g[2], g_d3c3_w=g[3], g_d4r2_w=g[4], g_d2c4_bnw=g[5])


class _FeatByShape(nn.Module):
    """returns the prepared 1/4-res features: first call -> left, second call -> right"""

    def __init__(self, fL, fR):
        super().__init__()
        self.feats, self.i = [fL, fR], 0

    def forward(self, x):
        f = self.feats[self.i % 2]
        self.i += 1
        return {"gwc_feature": f}


def gen_losses():
    import models.loss as ref_loss
    gt = torch.rand(2, 1, 32, 64, generator=torch.Generator().manual_seed(7)) * 40.0 - 2.0   # some invalid (<0, >=32)
    ests = [torch.softmax(seeded_tensor(f"loss.e{i}", (2, 8, 8, 16)), 1).requires_grad_() for i in range(5)]
    fl = ref_loss.focal_loss(ests, gt, 32, 5.0, False)
    fls = ref_loss.focal_loss(ests, gt, 32, 5.0, True)
    g = torch.autograd.grad(fl, ests)
    d0 = (seeded_tensor("loss.d0", (2, 1, 32, 64)) * 3 + gt).requires_grad_()
    d1 = (seeded_tensor("loss.d1", (2, 1, 32, 64)) * 0.3 + gt).requires_grad_()
    mask = (gt < 32) & (gt > 0)
    ml = ref_loss.model_loss([d0, d1], gt, mask)
    gd = torch.autograd.grad(ml, [d0, d1])
    npz("losses", gt=gt, focal=fl, focal_sparse=fls, gfocal0=g[0], gfocal4=g[4], model=ml, gd0=gd[0], gd1=gd[1])


def gen_state_dict_keys():
    """Key names + shapes of the reference model's state dict (the on-disk checkpoint format, main_dca.py:58-61)."""
    import json
    out = {}
    for variant, concat in (("g", False), ("gc", True)):
        m = ref_dca.GwcNet(192, use_concat_volume=concat)
        out[variant] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("state_dict_keys:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["volumes", "inject", "attention", "cva", "magg", "hot", "keys", "losses", "baseline", "whole", "init"]
    with torch.enable_grad():
        if "volumes" in which: gen_volumes()
        if "inject" in which: gen_context_inject()
        if "attention" in which: gen_attention()
        if "cva" in which: gen_cva()
        if "magg" in which: gen_multi_agg_hourglass()
        if "hot" in which: gen_hot_path()
        if "keys" in which: gen_state_dict_keys()
        if "losses" in which: gen_losses()
        if "baseline" in which: gen_baseline()
        if "whole" in which: gen_whole_model()
        if "init" in which: gen_init()
