"""CPU ORACLE for the DCANet cost-volume hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch CPU ops + closed forms, fp32 or fp64) of the
reference algorithm on the north-star path.  It is the *checker*: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product
package never imports anything from `oracle/` and fails loudly when its HIP library is
missing.

Parity status: PINNED.  Every function below is checked against outputs of the imported
reference itself (`oracle/make_golden.py` imports `/root/reference` in the build container
and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` replays them).

All functions are functional (state-dict + tensors in, tensors out) so that no reference
class is needed on the GPU box.  Citations are `file:line` into `/root/reference`.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# a1/a2/a3: cost-volume builders (models/submodule.py:134-167)
# --------------------------------------------------------------------------------------
def groupwise_correlation(fea1, fea2, num_groups):
    """models/submodule.py:148-154 -- mean over the C/G channels of each group."""
    B, C, H, W = fea1.shape
    assert C % num_groups == 0
    return (fea1 * fea2).view(B, num_groups, C // num_groups, H, W).mean(dim=2)


def build_gwc_volume(ref, tgt, maxdisp, num_groups):
    """models/submodule.py:157-167.  V[b,g,i,y,x] = mean_c ref[b,c,y,x]*tgt[b,c,y,x-i], 0 for x<i.

    Closed form: shift the target instead of slicing the output.
    """
    B, C, H, W = ref.shape
    vol = ref.new_zeros(B, num_groups, maxdisp, H, W)
    for i in range(min(maxdisp, W)):
        vol[:, :, i, :, i:] = groupwise_correlation(ref[..., i:], tgt[..., : W - i], num_groups)
    return vol


def build_concat_volume(ref, tgt, maxdisp):
    """models/submodule.py:134-145."""
    B, C, H, W = ref.shape
    vol = ref.new_zeros(B, 2 * C, maxdisp, H, W)
    for i in range(min(maxdisp, W)):
        vol[:, :C, i, :, i:] = ref[..., i:]
        vol[:, C:, i, :, i:] = tgt[..., : W - i]
    return vol


def disparity_regression(x, maxdisp):
    """models/submodule.py:127-131 -- sum_k k * x[:,k], keepdim."""
    assert x.dim() == 4
    k = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(x * k, 1, keepdim=True)


# --------------------------------------------------------------------------------------
# a4/a9: conv + BatchNorm3d blocks (models/submodule.py:121-124)
# --------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x, training: bool, update_stats: bool = True):
    """nn.BatchNorm3d defaults: eps 1e-5, momentum 0.1 (a9)."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training and not update_stats:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, 0.1, 1e-5)


def convbn_3d(sd: SD, p: str, x, stride, pad, training):
    """`convbn_3d` = Sequential(Conv3d(bias=False), BatchNorm3d); keys p.0.weight, p.1.*"""
    y = F.conv3d(x, sd[p + ".0.weight"], None, stride, pad)
    return _bn(sd, p + ".1", y, training)


def dres0(sd, p, x, training):
    """models/gwcnet_dca_g.py:141-144."""
    x = F.relu(convbn_3d(sd, p + ".0", x, 1, 1, training))
    return F.relu(convbn_3d(sd, p + ".2", x, 1, 1, training))


def dres1(sd, p, x, training):
    """models/gwcnet_dca_g.py:146-148 (residual add happens at :225, no ReLU after)."""
    y = F.relu(convbn_3d(sd, p + ".0", x, 1, 1, training))
    return convbn_3d(sd, p + ".2", y, 1, 1, training)


def classif(sd, p, x, training):
    """models/gwcnet_dca_g.py:154-168: convbn+ReLU, Conv3d 32->1."""
    y = F.relu(convbn_3d(sd, p + ".0", x, 1, 1, training))
    return F.conv3d(y, sd[p + ".2.weight"], None, 1, 1)


# --------------------------------------------------------------------------------------
# a5.3: homogeneous-region context injection (models/augment/semantic_level.py:96-126)
# --------------------------------------------------------------------------------------
def context_inject(x, preds):
    """Closed form of the per-(batch, class) python loop.

    p = softmax_k(preds); k* = argmax_k p; w = softmax over {pixels with the same (b,k*)}
    of p[., k*]; returns key_feats = feats_sl + x = x * (1 + onehot(k*) * w).
    Also returns (k*, w) for diagnostics.
    """
    B, C, n, h, w_ = x.shape
    p = F.softmax(preds, dim=1)                      # semantic_level.py:98
    kstar = p.argmax(dim=1)                          # (B,h,w), semantic_level.py:109
    m = p.gather(1, kstar.unsqueeze(1)).squeeze(1)
    e = torch.exp(m - 1.0)                           # any constant shift cancels in the ratio
    flat_k = kstar.reshape(B, -1)
    denom = torch.zeros(B, n, dtype=x.dtype).scatter_add_(1, flat_k, e.reshape(B, -1))
    wgt = e / denom.gather(1, flat_k).reshape(B, h, w_)
    onehot = F.one_hot(kstar, n).permute(0, 3, 1, 2).to(x.dtype)     # (B,n,h,w)
    scale = 1.0 + onehot * wgt.unsqueeze(1)
    return x * scale.unsqueeze(1), kstar, wgt


# --------------------------------------------------------------------------------------
# a5.4: per-pixel disparity attention (models/augment/SelfAttention_bn.py:62-98,136-160)
# --------------------------------------------------------------------------------------
def _proj_layer(sd, p, x, training):
    """One Sequential(Conv3d 1x1x1, BatchNorm3d, LeakyReLU(0.1)); keys p.0.weight, p.1.*"""
    y = F.conv3d(x, sd[p + ".0.weight"])
    return F.leaky_relu(_bn(sd, p + ".1", y, training), 0.1)


def _project(sd, p, x, num_convs, training):
    if num_convs == 1:
        return _proj_layer(sd, p, x, training)
    for i in range(num_convs):
        x = _proj_layer(sd, f"{p}.{i}", x, training)
    return x


def disparity_attention_core(q, k, v, head_dim=8):
    """SelfAttention_bn.py:70-94: per pixel, per head (8 channels), softmax(q k^T / sqrt 8) v
    over the disparity bins."""
    B, C, n, h, w = q.shape
    nh = C // head_dim
    qq = q.reshape(B, nh, head_dim, n, h * w).permute(0, 4, 1, 3, 2)   # B,hw,head,n,hc
    kk = k.reshape(B, nh, head_dim, n, h * w).permute(0, 4, 1, 2, 3)   # B,hw,head,hc,n
    vv = v.reshape(B, nh, head_dim, n, h * w).permute(0, 4, 1, 3, 2)   # B,hw,head,n,hc
    sim = torch.matmul(qq, kk) * (head_dim ** -0.5)
    sim = F.softmax(sim, dim=-1)
    ctx = torch.matmul(sim, vv)                                         # B,hw,head,n,hc
    ctx = ctx.permute(0, 2, 4, 3, 1).reshape(B, C, n, h, w)
    return ctx


def self_attention_block(sd, p, query_feats, key_feats, training):
    """SelfAttention_bn.py:62-98 with the ctor arguments of semantic_level.py:20-34."""
    q = _project(sd, p + ".query_project", query_feats, 2, training)
    k = _project(sd, p + ".key_project", key_feats, 2, training)
    v = _project(sd, p + ".value_project", key_feats, 1, training)
    ctx = disparity_attention_core(q, k, v)
    return _project(sd, p + ".out_project", ctx, 1, training)


def semantic_level_context(sd, p, x, preds, training):
    """models/augment/semantic_level.py:96-128."""
    key, _, _ = context_inject(x, preds)
    return self_attention_block(sd, p + ".cross_attention", x, key, training)


# --------------------------------------------------------------------------------------
# a5 / a5.7 / a6: DCA block, Multi_Aggregation, hourglass
# --------------------------------------------------------------------------------------
def multi_aggregation(sd, p, x, training):
    """models/augment/cva.py:13-31."""
    c1 = F.relu(convbn_3d(sd, p + ".conv1.0", x, 2, 1, training))
    c2 = F.relu(convbn_3d(sd, p + ".conv2.0", c1, 1, 1, training))
    c3 = F.conv_transpose3d(c2, sd[p + ".conv3.0.weight"], None, 2, 1, 1)
    c3 = _bn(sd, p + ".conv3.1", c3, training)
    r = F.conv3d(x, sd[p + ".redir.0.weight"])
    r = _bn(sd, p + ".redir.1", r, training)
    return F.relu(c3 + r)


def cva(sd, p, cost_volume, training):
    """models/augment/cva.py:59-72 (downsample=True branch).  Returns (prob_volume (B,1,n,h,w), aug)."""
    x = F.avg_pool3d(cost_volume, (3, 3, 3), stride=2, padding=1)              # cva.py:39
    cost_down = F.relu(convbn_3d(sd, p + ".downsample.1", x, 1, 1, training))  # cva.py:40-41
    prob = classif(sd, p + ".classify", cost_down, training).squeeze(1)        # cva.py:51-53,62
    aug_down = semantic_level_context(sd, p + ".slc_net", cost_down, prob, training)
    aug = F.interpolate(aug_down, scale_factor=(2, 2, 2), mode="trilinear")    # cva.py:64
    y = F.conv3d(torch.cat([aug, cost_volume], 1), sd[p + ".fuse.0.0.weight"])
    aug = _bn(sd, p + ".fuse.0.1", y, training)                                # cva.py:55,69
    aug = multi_aggregation(sd, p + ".cost_agg", aug, training)                # cva.py:70
    return prob.unsqueeze(1), aug


def hourglass(sd, p, x, training):
    """models/gwcnet.py:67-104 (identical copy at gwcnet_dca_g.py:69-106)."""
    c1 = F.relu(convbn_3d(sd, p + ".conv1.0", x, 2, 1, training))
    c2 = F.relu(convbn_3d(sd, p + ".conv2.0", c1, 1, 1, training))
    c3 = F.relu(convbn_3d(sd, p + ".conv3.0", c2, 2, 1, training))
    c4 = F.relu(convbn_3d(sd, p + ".conv4.0", c3, 1, 1, training))
    c5 = F.conv_transpose3d(c4, sd[p + ".conv5.0.weight"], None, 2, 1, 1)
    c5 = _bn(sd, p + ".conv5.1", c5, training)
    r2 = _bn(sd, p + ".redir2.1", F.conv3d(c2, sd[p + ".redir2.0.weight"]), training)
    c5 = F.relu(c5 + r2)
    c6 = F.conv_transpose3d(c5, sd[p + ".conv6.0.weight"], None, 2, 1, 1)
    c6 = _bn(sd, p + ".conv6.1", c6, training)
    r1 = _bn(sd, p + ".redir1.1", F.conv3d(x, sd[p + ".redir1.0.weight"]), training)
    return F.relu(c6 + r1)


# --------------------------------------------------------------------------------------
# The hot path of GwcNet.forward (models/gwcnet_dca_g.py:216-239, training heads :244-278)
# --------------------------------------------------------------------------------------
def hot_path(sd: SD, fL, fR, maxdisp: int, training: bool, num_groups: int = 40,
             cL: Optional[torch.Tensor] = None, cR: Optional[torch.Tensor] = None):
    """From 1/4-res features to the 1/4-res disparity (before `prop`) and all auxiliary heads.

    Returns a dict: pred4_q (B,1,h,w) [1/4-res pixel units], prob_volume{1,2,3} (B,1,n,h/2,w/2),
    cost3 (B,d,h,w) softmax, and in training mode pred0, pred_dca1, pred_dca2, pred1, pred2
    (B,d,h,w) and pred_dca3 (B,1,4h,4w).
    """
    d = maxdisp // 4
    vol = build_gwc_volume(fL, fR, d, num_groups)                       # :216
    if cL is not None:
        vol = torch.cat((vol, build_concat_volume(cL, cR, d)), 1)       # :217-220
    cost0 = dres0(sd, "dres0", vol, training)                           # :224
    cost0 = dres1(sd, "dres1", cost0, training) + cost0                 # :225
    prob1, aug = cva(sd, "cva1", cost0, training)                       # :228
    out1 = cost0 + aug                                                  # :229
    prob2, out2 = cva(sd, "cva2", out1, training)                       # :231
    prob3, out3 = cva(sd, "cva3", out2, training)                       # :232
    o3 = classif(sd, "classif3", out3, training).squeeze(1)            # :235-237
    cost3 = F.softmax(o3, dim=1)                                        # :238
    pred4_q = disparity_regression(cost3, d)                            # :239
    res = dict(pred4_q=pred4_q, cost3=cost3, prob_volume1=prob1, prob_volume2=prob2,
               prob_volume3=prob3, cost0=cost0, out1=out1, out2=out2, out3=out3)
    if training:
        res["pred0"] = F.softmax(classif(sd, "classif0", cost0, training).squeeze(1), dim=1)   # :245-248
        up = lambda t, s: F.interpolate(t, scale_factor=(s, s, s), mode="trilinear")
        res["pred_dca1"] = F.softmax(up(prob1, 2).squeeze(1), dim=1)    # :251-253
        res["pred_dca2"] = F.softmax(up(prob2, 2).squeeze(1), dim=1)    # :256-258
        p3 = F.softmax(up(prob3, 8).squeeze(1), dim=1)                  # :261-263
        res["pred_dca3"] = disparity_regression(p3, maxdisp)            # :264
        res["pred1"] = F.softmax(classif(sd, "classif1", out1, training).squeeze(1), dim=1)    # :266-269
        res["pred2"] = F.softmax(classif(sd, "classif2", out2, training).squeeze(1), dim=1)    # :272-275
    return res


def hot_path_baseline(sd: SD, fL, fR, maxdisp: int, num_groups: int = 40):
    """Training branch of the baseline gwcnet.GwcNet.forward (models/gwcnet.py:194-238) from the 1/4-res features:
    returns [pred0..pred3], each (B,1,4h,4w).  BN layers run in training mode (batch statistics)."""
    d = maxdisp // 4
    vol = build_gwc_volume(fL, fR, d, num_groups)
    cost0 = dres0(sd, "dres0", vol, True)
    cost0 = dres1(sd, "dres1", cost0, True) + cost0
    out1 = hourglass(sd, "dres2", cost0, True)
    out2 = hourglass(sd, "dres3", out1, True)
    out3 = hourglass(sd, "dres4", out2, True)
    preds = []
    for i, t in enumerate((cost0, out1, out2, out3)):
        c = classif(sd, f"classif{i}", t, True)
        c = F.interpolate(c, size=[maxdisp, 4 * fL.shape[2], 4 * fL.shape[3]], mode="trilinear").squeeze(1)
        preds.append(disparity_regression(F.softmax(c, dim=1), maxdisp))
    return preds


def baseline_shapes(num_groups: int = 40):
    s = {}

    def bn(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,); s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    def convbn(p, ci, co, k):
        s[p + ".0.weight"] = (co, ci, k, k, k); bn(p + ".1", co)

    convbn("dres0.0", num_groups, 32, 3); convbn("dres0.2", 32, 32, 3)
    convbn("dres1.0", 32, 32, 3); convbn("dres1.2", 32, 32, 3)
    for i in range(4):
        convbn(f"classif{i}.0", 32, 32, 3); s[f"classif{i}.2.weight"] = (1, 32, 3, 3, 3)
    for name in ("dres2", "dres3", "dres4"):
        s.update(hourglass_shapes(name, 32))
    return s


# --------------------------------------------------------------------------------------
# SURVEY 8(f) neighbours of the path: 2D feature extractor, Guidance, convex up-sampler, whole model, losses
# --------------------------------------------------------------------------------------
def _bn2(sd: SD, p: str, x, training: bool):
    """nn.BatchNorm2d defaults (same semantics as _bn)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training, 0.1, 1e-5)


def _convbn2(sd: SD, p: str, x, stride, pad, dilation, training):
    """`convbn` models/submodule.py:115-118: Conv2d(bias=False, padding = dilation if dilation > 1 else pad) + BN."""
    y = F.conv2d(x, sd[p + ".0.weight"], None, stride, dilation if dilation > 1 else pad, dilation)
    return _bn2(sd, p + ".1", y, training)


def _basic_block(sd: SD, p: str, x, stride, pad, dilation, training):
    """BasicBlock models/submodule.py:251-273: convbn+ReLU, convbn, (+ downsample), `out += x`, NO final ReLU."""
    out = F.relu(_convbn2(sd, p + ".conv1.0", x, stride, pad, dilation, training))
    out = _convbn2(sd, p + ".conv2", out, 1, pad, dilation, training)
    if (p + ".downsample.0.weight") in sd:
        x = _bn2(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), training)
    return out + x


def feature_extraction(sd: SD, x, training: bool, p: str = "feature_extraction"):
    """models/gwcnet_dca_g.py:13-66.  Returns (gwc_feature (B,320,H/4,W/4), concat_feature or None)."""
    for i, stride in ((0, 2), (2, 1), (4, 1)):                                      # firstconv :19-24
        x = F.relu(_convbn2(sd, f"{p}.firstconv.{i}", x, stride, 1, 1, training))
    for layer, blocks, stride, dil in (("layer1", 3, 1, 1), ("layer2", 16, 2, 1), ("layer3", 3, 1, 1),
                                       ("layer4", 3, 1, 2)):                        # :26-29
        for b in range(blocks):
            x = _basic_block(sd, f"{p}.{layer}.{b}", x, stride if b == 0 else 1, 1, dil, training)
        if layer == "layer2":
            l2 = x
        elif layer == "layer3":
            l3 = x
    gwc = torch.cat((l2, l3, x), dim=1)                                             # :60
    if (p + ".lastconv.2.weight") not in sd:
        return gwc, None
    c = F.relu(_convbn2(sd, p + ".lastconv.0", gwc, 1, 1, 1, training))              # :31-35
    return gwc, F.conv2d(c, sd[p + ".lastconv.2.weight"])


def _residual_block(sd: SD, p: str, x, stride, training):
    """ResidualBlock models/submodule.py:305-355, norm_fn='batch'.  `norm3` is the same module as `downsample.1`:
    load_state_dict fills `downsample.1.*` last, so those keys hold the live values."""
    y = F.relu(_bn2(sd, p + ".norm1", F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride, 1), training))
    y = F.relu(_bn2(sd, p + ".norm2", F.conv2d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], 1, 1), training))
    if stride != 1:
        x = _bn2(sd, p + ".downsample.1",
                 F.conv2d(x, sd[p + ".downsample.0.weight"], sd[p + ".downsample.0.bias"], stride), training)
    return F.relu(x + y)


def guidance(sd: SD, x, training: bool, p: str = "guidance"):
    """Guidance models/submodule.py:395-460 -> g (B,64,H/4,W/4).  (`norm1` is shared with `conv_start.1`, whose keys
    are loaded last.)"""
    x = F.relu(_bn2(sd, p + ".conv_start.1",
                    F.conv2d(x, sd[p + ".conv_start.0.weight"], sd[p + ".conv_start.0.bias"], 2, 3), training))
    for layer, stride in (("layer1", 1), ("layer2", 2)):
        x = _residual_block(sd, f"{p}.{layer}.0", x, stride, training)
        x = _residual_block(sd, f"{p}.{layer}.1", x, 1, training)
    for i in (0, 1):                                                                 # BasicConv :276-302
        x = F.relu(_bn2(sd, f"{p}.conv_g0.{i}.bn", F.conv2d(x, sd[f"{p}.conv_g0.{i}.conv.weight"], None, 1, 1), training))
    return F.conv2d(x, sd[p + ".guidance.weight"], None, 1, 1)


def convex_upsample(mask_logits, disp):
    """PropgationNet_4x.forward after its conv (models/submodule.py:366-373), written with explicit
    shifts instead of unfold: mask_logits (B,144,h,w) with channel = k*16 + i*4 + j, softmax over the 9 neighbours k,
    up[b,0,4y+i,4x+j] = sum_k mask[k,i,j,y,x] * 4*disp[y+k//3-1, x+k%3-1] (zero padded)."""
    b, _, h, w = disp.shape
    m = F.softmax(mask_logits.view(b, 9, 4, 4, h, w), dim=1)
    dp = F.pad(4 * disp, (1, 1, 1, 1))
    up = 0
    for k in range(9):
        ky, kx = k // 3, k % 3
        up = up + m[:, k] * dp[:, :, ky:ky + h, kx:kx + w].reshape(b, 1, 1, h, w)
    return up.permute(0, 3, 1, 4, 2).reshape(b, 1, 4 * h, 4 * w)


def prop(sd: SD, g, disp, training: bool, p: str = "prop"):
    """PropgationNet_4x models/submodule.py:357-373."""
    c = F.relu(_convbn2(sd, p + ".conv.0", g, 1, 1, 1, training))
    return convex_upsample(F.conv2d(c, sd[p + ".conv.2.weight"], None, 1, 1), disp)


def whole_model(sd: SD, left, right, maxdisp: int, training: bool, num_groups: int = 40):
    """GwcNet.forward (models/gwcnet_dca_g.py:209-282) end to end.  Returns the reference's own return value:
    train -> ([pred0,pred_dca1,pred_dca2,pred1,pred2], [pred_dca3,pred4]); eval -> (pred4, prob_volume2.squeeze(1)),
    plus a dict of intermediates as a third element."""
    fL, cL = feature_extraction(sd, left, training)                                   # :213
    fR, cR = feature_extraction(sd, right, training)                                  # :214
    g = guidance(sd, left, training)                                                  # :215
    r = hot_path(sd, fL, fR, maxdisp, training, num_groups, cL, cR)                   # :216-239, 244-275
    pred4 = prop(sd, g, r["pred4_q"], training)                                       # :240
    aux = dict(r, gwc_feature=fL, guidance=g)
    if training:
        return [r["pred0"], r["pred_dca1"], r["pred_dca2"], r["pred1"], r["pred2"]], [r["pred_dca3"], pred4], aux
    return pred4, r["prob_volume2"].squeeze(1), aux


def model_loss(disp_ests, disp_gt, mask):
    """models/loss.py:6-14: 1.8 / 2.1 weighted smooth-L1 (beta 1, mean) over the masked pixels."""
    weights = [1.8, 2.1]
    assert len(weights) == len(disp_ests)
    return sum(w * F.smooth_l1_loss(e[mask], disp_gt[mask], reduction="mean") for e, w in zip(disp_ests, weights))


def stereo_focal_loss_level(est, gt, max_disp, focal_coefficient, sparse):
    """StereoFocalLoss.loss_per_level with LaplaceDisp2Prob (models/loss.py:206-240, 60-128), line by line."""
    N, C, H, W = est.shape
    sgt, scale = gt.clone(), 1.0
    if gt.shape[-2] != H or gt.shape[-1] != W:                                        # :210-215
        scale = gt.shape[-1] / (W * 1.0)
        sgt = (F.adaptive_max_pool2d if sparse else F.adaptive_avg_pool2d)(gt.clone() / scale, (H, W))
    upper = int(max_disp / scale)                                                     # :220-221
    mask = ((sgt > 0) & (sgt < upper)).to(sgt.dtype)
    if mask.sum() < 1.0:                                                              # :224-227
        prob = torch.zeros_like(est)
    else:
        mgt = sgt * mask                                                              # :230
        nd = upper
        index = torch.arange(0, nd, dtype=mgt.dtype).view(1, nd, 1, 1)
        inner = ((mgt > 0) & (mgt < nd - 1)).to(mgt.dtype)                            # :87-89 (end_disp = nd - 1, :70)
        prob = F.softmax(-torch.abs(index - mgt * inner), dim=1) * inner + 1e-40      # :90-96, 124-126
    logp = F.log_softmax(est, dim=1)                                                  # :236
    weight = (1.0 - prob).pow(-focal_coefficient)                                     # :237
    return -((prob * logp) * weight * mask).sum(dim=1, keepdim=True).mean()           # :238


def focal_loss(disp_ests, disp_gt, maxdisp, focal_coefficient, sparse):
    """models/loss.py:16-24."""
    weights = [0.5, 0.7, 1.0, 1.2, 1.5]
    return sum(w * stereo_focal_loss_level(e, disp_gt, maxdisp, focal_coefficient, sparse)
               for e, w in zip(disp_ests, weights))


# --------------------------------------------------------------------------------------
# Deterministic, well-conditioned test weights
# --------------------------------------------------------------------------------------
GAIN = 1.5
LOGIT_SCALE = 0.3
SEED = 20241


def _gen(key: str) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ SEED) & 0x7FFFFFFF)


def seeded_state_dict(shapes: Dict[str, tuple]) -> SD:
    """Key-seeded weights: identical on every machine for the same torch CPU generator.

    `shapes` maps state-dict key -> shape.  Rules: conv/deconv weights
    ~ N(0, sqrt(GAIN/(k^3*Cout))); last 1-channel classifier convs x0.3; BN weight ~U(.75,1.25),
    bias ~N(0,.1), running_mean ~N(0,.1), running_var ~U(.75,1.25); closing BN gamma of 2D
    residual branches x0.25.
    """
    sd: SD = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        g = _gen(k)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shp, generator=g) * 0.5 + 0.75
        elif len(shp) == 1 and k.endswith(".weight"):
            v = torch.rand(shp, generator=g) * 0.5 + 0.75
            if _is_residual_closing_bn(k):
                v = v * 0.25
            sd[k] = v
        elif len(shp) == 1 and k.endswith(".bias"):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        elif len(shp) in (4, 5):
            kprod = 1
            for s in shp[2:]:
                kprod *= s
            cout = shp[1] if _is_transposed(k) else shp[0]
            v = torch.randn(shp, generator=g) * math.sqrt(GAIN / (kprod * cout))
            if _is_logit_conv(k):
                v = v * LOGIT_SCALE
            sd[k] = v
        else:
            raise ValueError(f"no rule for {k} {shp}")
    return sd


def _is_transposed(k: str) -> bool:
    return (".cost_agg.conv3.0.weight" in k) or k.endswith("conv5.0.weight") or k.endswith("conv6.0.weight")


def _is_logit_conv(k: str) -> bool:
    return (k.startswith("classif") and k.endswith(".2.weight")) or k.endswith("classify.2.weight")


def _is_residual_closing_bn(k: str) -> bool:
    if k.startswith("feature_extraction.layer") and k.endswith(".conv2.1.weight"):
        return True
    return k.startswith("guidance.layer") and k.endswith(".norm2.weight")


def hot_path_shapes(use_concat_volume: bool = False, num_groups: int = 40, concat_channels: int = 12):
    """State-dict key -> shape for the hot-path parameters of `gwcnet_dca_g.GwcNet`
    (dres0/1, cva1..3, classif0..3); matches SURVEY Appendix A.4."""
    s: Dict[str, tuple] = {}

    def bn(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,); s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    def convbn(p, ci, co, k):
        s[p + ".0.weight"] = (co, ci, k, k, k); bn(p + ".1", co)

    c0 = num_groups + (2 * concat_channels if use_concat_volume else 0)
    convbn("dres0.0", c0, 32, 3); convbn("dres0.2", 32, 32, 3)
    convbn("dres1.0", 32, 32, 3); convbn("dres1.2", 32, 32, 3)
    for i in range(4):
        convbn(f"classif{i}.0", 32, 32, 3); s[f"classif{i}.2.weight"] = (1, 32, 3, 3, 3)
    for i in (1, 2, 3):
        p = f"cva{i}"
        convbn(p + ".downsample.1", 32, 32, 3)
        convbn(p + ".classify.0", 32, 32, 3); s[p + ".classify.2.weight"] = (1, 32, 3, 3, 3)
        a = p + ".slc_net.cross_attention"
        for proj in ("key_project", "query_project"):
            for j in (0, 1):
                s[f"{a}.{proj}.{j}.0.weight"] = (32, 32, 1, 1, 1); bn(f"{a}.{proj}.{j}.1", 32)
        for proj in ("value_project", "out_project"):
            s[f"{a}.{proj}.0.weight"] = (32, 32, 1, 1, 1); bn(f"{a}.{proj}.1", 32)
        convbn(p + ".fuse.0", 64, 32, 1)
        convbn(p + ".cost_agg.conv1.0", 32, 64, 3)
        convbn(p + ".cost_agg.conv2.0", 64, 64, 3)
        s[p + ".cost_agg.conv3.0.weight"] = (64, 32, 3, 3, 3); bn(p + ".cost_agg.conv3.1", 32)
        s[p + ".cost_agg.redir.0.weight"] = (32, 32, 1, 1, 1); bn(p + ".cost_agg.redir.1", 32)
    return s


def hourglass_shapes(p: str, c: int = 32):
    s: Dict[str, tuple] = {}

    def bn(q, ch):
        s[q + ".weight"] = (ch,); s[q + ".bias"] = (ch,)
        s[q + ".running_mean"] = (ch,); s[q + ".running_var"] = (ch,)
        s[q + ".num_batches_tracked"] = ()

    def convbn(q, ci, co, k):
        s[q + ".0.weight"] = (co, ci, k, k, k); bn(q + ".1", co)

    convbn(p + ".conv1.0", c, 2 * c, 3); convbn(p + ".conv2.0", 2 * c, 2 * c, 3)
    convbn(p + ".conv3.0", 2 * c, 4 * c, 3); convbn(p + ".conv4.0", 4 * c, 4 * c, 3)
    s[p + ".conv5.0.weight"] = (4 * c, 2 * c, 3, 3, 3); bn(p + ".conv5.1", 2 * c)
    s[p + ".conv6.0.weight"] = (2 * c, c, 3, 3, 3); bn(p + ".conv6.1", c)
    convbn(p + ".redir1", c, c, 1); convbn(p + ".redir2", 2 * c, 2 * c, 1)
    return s


def clone_sd(sd: SD, dtype=None) -> SD:
    out = {}
    for k, v in sd.items():
        out[k] = v.clone() if (dtype is None or not v.is_floating_point()) else v.to(dtype).clone()
    return out
