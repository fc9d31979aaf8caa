"""Autograd operators of the DCANet hot path, each a thin host wrapper over the C ABI of
libdca_hip.so (include/dca_hip.h).  PyTorch is used for device memory, streams and autograd graph
bookkeeping only; every arithmetic step on the path runs in a hand-written HIP kernel.

There is no CPU / eager fallback: tensors must be fp32 on a ROCm device and the library must load.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

import torch

from . import _lib

_vp = ctypes.c_void_p


def _L():
    return _lib.load()


def _ptr(t: Optional[torch.Tensor]):
    return _vp(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's CURRENT stream on the current device as a hipStream_t (every launch goes there).  The raw C accessors
    cost ~0.3 us against ~9 us for torch.cuda.current_stream() -- there are ~1000 launches per training step."""
    if _raw_stream is not None and _raw_device is not None:
        return _vp(_raw_stream(_raw_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def _chk(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"{name} failed with hipError_t {rc}")


def _req(t: torch.Tensor, name: str, packed_ok: bool = False) -> torch.Tensor:
    if getattr(t, "_dca_px2", None) is not None and not packed_ok:
        raise RuntimeError(f"{name}: got a tensor in the packed px2 operand format (only the f16x2 3x3x3 stride-1 convolution "
                           "reads it; ask the producer for an fp32 result)")
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"{name}: the DCANet hot path runs only as HIP kernels on a ROCm device (got "
            f"{'a ' + str(t.device) + ' tensor' if isinstance(t, torch.Tensor) else type(t)}); there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def _opt(t, name):
    return None if t is None else _req(t, name)


# ------------------------------------------------------------------------------------------------
# cost volumes
# ------------------------------------------------------------------------------------------------
class _GwcVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, tgt, maxdisp, num_groups):
        ref, tgt = _req(ref, "build_gwc_volume"), _req(tgt, "build_gwc_volume")
        B, C, H, W = ref.shape
        vol = torch.empty((B, num_groups, maxdisp, H, W), device=ref.device, dtype=torch.float32)
        with torch.cuda.device_of(ref):
            _chk(_L().dca_gwc_volume_fwd(_ptr(ref), _ptr(tgt), _ptr(vol), B, C, H, W, maxdisp, num_groups, _stream()),
                 "dca_gwc_volume_fwd")
        ctx.save_for_backward(ref, tgt)
        ctx.meta = (maxdisp, num_groups)
        return vol

    @staticmethod
    def backward(ctx, gvol):
        ref, tgt = ctx.saved_tensors
        maxdisp, G = ctx.meta
        gvol = _req(gvol, "build_gwc_volume.backward")
        B, C, H, W = ref.shape
        gref, gtgt = torch.empty_like(ref), torch.empty_like(tgt)
        with torch.cuda.device_of(ref):
            _chk(_L().dca_gwc_volume_bwd(_ptr(gvol), _ptr(ref), _ptr(tgt), _ptr(gref), _ptr(gtgt), B, C, H, W,
                                         maxdisp, G, _stream()), "dca_gwc_volume_bwd")
        return gref, gtgt, None, None


class _ConcatVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, tgt, maxdisp):
        ref, tgt = _req(ref, "build_concat_volume"), _req(tgt, "build_concat_volume")
        B, C, H, W = ref.shape
        vol = torch.empty((B, 2 * C, maxdisp, H, W), device=ref.device, dtype=torch.float32)
        with torch.cuda.device_of(ref):
            _chk(_L().dca_concat_volume_fwd(_ptr(ref), _ptr(tgt), _ptr(vol), B, C, H, W, maxdisp, _stream()),
                 "dca_concat_volume_fwd")
        ctx.meta = (B, C, H, W, maxdisp)
        return vol

    @staticmethod
    def backward(ctx, gvol):
        B, C, H, W, maxdisp = ctx.meta
        gvol = _req(gvol, "build_concat_volume.backward")
        gref = torch.empty((B, C, H, W), device=gvol.device, dtype=torch.float32)
        gtgt = torch.empty_like(gref)
        with torch.cuda.device_of(gvol):
            _chk(_L().dca_concat_volume_bwd(_ptr(gvol), _ptr(gref), _ptr(gtgt), B, C, H, W, maxdisp, _stream()),
                 "dca_concat_volume_bwd")
        return gref, gtgt, None


def gwc_volume(ref, tgt, maxdisp, num_groups):
    return _GwcVolume.apply(ref, tgt, int(maxdisp), int(num_groups))


class _CostVolume(torch.autograd.Function):
    """Fused builder (csrc/volume_fused.hip): gwc volume from 1-3 channel segments per side + optional concat volume,
    one output tensor, fp32 or the reduced-precision storage type."""

    @staticmethod
    def forward(ctx, maxdisp, num_groups, out_dtype, nseg, has_concat, *tensors):
        refs = [_req(t, "cost_volume.ref") for t in tensors[:nseg]]
        tgts = [_req(t, "cost_volume.tgt") for t in tensors[nseg:2 * nseg]]
        cref = _req(tensors[2 * nseg], "cost_volume.cref") if has_concat else None
        ctgt = _req(tensors[2 * nseg + 1], "cost_volume.ctgt") if has_concat else None
        B, _, H, W = refs[0].shape
        segC = [t.shape[1] for t in refs]
        Cc = cref.shape[1] if has_concat else 0
        for r_, t_ in zip(refs, tgts):
            assert r_.shape == t_.shape and r_.shape[0] == B and tuple(r_.shape[2:]) == (H, W)
        vol = torch.empty((B, num_groups + 2 * Cc, maxdisp, H, W), device=refs[0].device, dtype=out_dtype)
        rp = (ctypes.c_void_p * nseg)(*[t.data_ptr() for t in refs])
        tp = (ctypes.c_void_p * nseg)(*[t.data_ptr() for t in tgts])
        sc = (ctypes.c_int * nseg)(*segC)
        code = 0 if out_dtype == torch.float32 else LP_DTYPES[out_dtype]
        # fp32 volume without concat part: the builder emits the per-channel maxima the first f16x2 convolution scales by
        vmax = _cslots(num_groups, vol.device) if (CONV_X2 and AMAX_EMIT and code == 0 and Cc == 0 and B * H <= CSLOTS) else None
        with torch.cuda.device_of(vol):
            _chk(_L().dca_cost_volume_fwd(rp, tp, sc, nseg, _ptr(cref), _ptr(ctgt), Cc, _ptr(vol), B, H, W, maxdisp,
                                          num_groups, code, _ptr(vmax), _stream()), "dca_cost_volume_fwd")
        _tls.last_vmax = (vmax, B * H)       # the caller tags the tensor it hands out (cost_volume below)
        ctx.save_for_backward(*refs, *tgts)
        ctx.meta = (maxdisp, num_groups, nseg, segC, Cc, (B, H, W))
        return vol

    @staticmethod
    def backward(ctx, gvol):
        maxdisp, G, nseg, segC, Cc, (B, H, W) = ctx.meta
        refs, tgts = ctx.saved_tensors[:nseg], ctx.saved_tensors[nseg:]
        gvol = _req(gvol, "cost_volume.backward")
        ref = refs[0] if nseg == 1 else torch.cat(refs, 1)
        tgt = tgts[0] if nseg == 1 else torch.cat(tgts, 1)
        gg = gvol if Cc == 0 else gvol[:, :G].contiguous()
        gref, gtgt = torch.empty_like(ref), torch.empty_like(tgt)
        lib = _L()
        with torch.cuda.device_of(gvol):
            _chk(lib.dca_gwc_volume_bwd(_ptr(gg), _ptr(ref), _ptr(tgt), _ptr(gref), _ptr(gtgt), B, ref.shape[1], H, W,
                                        maxdisp, G, _stream()), "dca_gwc_volume_bwd")
            grads = list(gref.split(segC, 1)) + list(gtgt.split(segC, 1))
            if Cc:
                gc = gvol[:, G:].contiguous()
                gcr = torch.empty((B, Cc, H, W), device=gvol.device, dtype=torch.float32)
                gct = torch.empty_like(gcr)
                _chk(lib.dca_concat_volume_bwd(_ptr(gc), _ptr(gcr), _ptr(gct), B, Cc, H, W, maxdisp, _stream()),
                     "dca_concat_volume_bwd")
                grads += [gcr, gct]
        return (None, None, None, None, None) + tuple(grads)


def cost_volume(ref, tgt, maxdisp, num_groups, cref=None, ctgt=None, out_dtype=torch.float32):
    """build_gwc_volume(ref, tgt) [cat build_concat_volume(cref, ctgt)] as ONE (B, G + 2*Cc, D, H, W) tensor.  `ref` /
    `tgt` may be tuples of channel segments (the extractor's l2 / l3 / l4 maps) that are read in place.  Falls back to
    the separate builders + torch.cat when the fused kernel's alignment needs (W % 4, D % 4) are not met."""
    refs = tuple(ref) if isinstance(ref, (tuple, list)) else (ref,)
    tgts = tuple(tgt) if isinstance(tgt, (tuple, list)) else (tgt,)
    W = refs[0].shape[-1]
    C = sum(t.shape[1] for t in refs)
    cpg = C // num_groups if num_groups else 0
    ok = (W % 4 == 0 and maxdisp % 4 == 0 and len(refs) <= 3 and cpg in (1, 2, 4, 8, 16) and C % num_groups == 0
          and all(t.shape[1] % cpg == 0 for t in refs))
    if not ok:
        if out_dtype != torch.float32:
            raise RuntimeError("cost_volume: the reduced-precision volume needs W % 4 == 0 and maxdisp % 4 == 0")
        r1 = refs[0] if len(refs) == 1 else torch.cat(refs, 1)
        t1 = tgts[0] if len(tgts) == 1 else torch.cat(tgts, 1)
        vol = gwc_volume(r1, t1, maxdisp, num_groups)
        return vol if cref is None else torch.cat((vol, concat_volume(cref, ctgt, maxdisp)), 1)
    extra = () if cref is None else (cref, ctgt)
    vol = _CostVolume.apply(int(maxdisp), int(num_groups), out_dtype, len(refs), cref is not None, *refs, *tgts, *extra)
    vmax, nslots = getattr(_tls, "last_vmax", (None, 0))
    _tls.last_vmax = (None, 0)
    if vmax is not None:
        _tag_cmax(vol, vmax, nslots)         # per-channel maxima from the builder: no read pass in front of dres0's first convolution
    return vol


def concat_volume(ref, tgt, maxdisp):
    return _ConcatVolume.apply(ref, tgt, int(maxdisp))


# ------------------------------------------------------------------------------------------------
# softmax(dim=1) / soft-argmin / disparity regression on (B, K, *spatial)
# ------------------------------------------------------------------------------------------------
class _SoftArgmin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = _req(x, "softargmin")
        B, K = x.shape[0], x.shape[1]
        HW = x[0, 0].numel()
        out = torch.empty_like(x) if mode == 0 else torch.empty((B, 1) + tuple(x.shape[2:]), device=x.device,
                                                                dtype=torch.float32)
        with torch.cuda.device_of(x):
            _chk(_L().dca_softargmin_fwd(_ptr(x), _ptr(out), B, K, HW, mode, _stream()), "dca_softargmin_fwd")
        ctx.mode = mode
        ctx.save_for_backward(out if mode == 0 else x)
        return out

    @staticmethod
    def backward(ctx, g):
        (aux,) = ctx.saved_tensors
        g = _req(g, "softargmin.backward")
        B, K = aux.shape[0], aux.shape[1]
        HW = aux[0, 0].numel()
        gx = torch.empty_like(aux)
        with torch.cuda.device_of(aux):
            _chk(_L().dca_softargmin_bwd(_ptr(aux), _ptr(g), _ptr(gx), B, K, HW, ctx.mode, _stream()),
                 "dca_softargmin_bwd")
        return gx, None


def softmax_dim1(x):
    return _SoftArgmin.apply(x, 0)


def softargmin(x):
    """disparity_regression(softmax(x, 1), K) fused: (B,K,...) logits -> (B,1,...)."""
    return _SoftArgmin.apply(x, 1)


def regression(x):
    """disparity_regression(x, K) for an arbitrary x: sum_k k*x[:,k], keepdim."""
    return _SoftArgmin.apply(x, 2)


class _UpSoftArgmin(torch.autograd.Function):
    """trilinear x s up-sampling + softmax(dim 1) + disparity regression, fused (gwcnet_dca_g.py:261-264)."""

    @staticmethod
    def forward(ctx, logits, scale):
        logits = _req(logits, "up_softargmin")
        B, n, hc, wc = logits.shape
        disp = torch.empty((B, 1, scale * hc, scale * wc), device=logits.device, dtype=torch.float32)
        with torch.cuda.device_of(logits):
            _chk(_L().dca_up_softargmin_fwd(_ptr(logits), _ptr(disp), B, n, hc, wc, scale, _stream()),
                 "dca_up_softargmin_fwd")
        ctx.save_for_backward(logits)
        ctx.scale = scale
        return disp

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        g = _req(g, "up_softargmin.backward")
        B, n, hc, wc = logits.shape
        s = ctx.scale
        g1 = torch.empty((B, n, s * hc, s * wc), device=logits.device, dtype=torch.float32)
        gl = torch.empty_like(logits)
        with torch.cuda.device_of(logits):
            _chk(_L().dca_up_softargmin_bwd(_ptr(logits), _ptr(g), _ptr(g1), _ptr(gl), B, n, hc, wc, s, _stream()),
                 "dca_up_softargmin_bwd")
        return gl, None


def up_softargmin(logits, scale):
    """(B,n,hc,wc) logits -> (B,1,s*hc,s*wc) expected disparity over s*n bins; falls back to the unfused kernels
    for n > 64 or scales other than 2 / 4 / 8."""
    if logits.shape[1] > 64 or logits.shape[1] < 2 or scale not in (2, 4, 8):
        return softargmin(trilinear_upsample(logits.unsqueeze(1), scale).squeeze(1))
    return _UpSoftArgmin.apply(logits, int(scale))


# ------------------------------------------------------------------------------------------------
# 3D convolutions
# ------------------------------------------------------------------------------------------------
def _round_up(a, m):
    return (a + m - 1) // m * m


# 3x3x3 stride-1 convolutions run on the bf16 matrix pipe with the exact three-way bf16 split of both operands
# ("bf16x3", conv3d_bf16x3.hip): fp32-grade accuracy (measured error vs fp64 slightly BELOW the fp32 MFMA kernel's)
# at 1.3-1.75x the speed on every shape the networks use, 1/4 to 1/16 resolution (tools/x3_vs_fp32.py).
# DCA_CONV=fp32 forces the fp32 MFMA kernel everywhere.
# DCA_CONV: "x2" (default) = 3x3x3 stride-1 convolutions and their weight gradients on the f16x2 split kernels
# (conv3d_f16x2.hip: two f16 terms per operand, three products, power-of-two operand scaling from the tensor's max |.|);
# "x3" = the bf16x3 split kernels for them; "fp32" = fp32 MFMA kernels everywhere.
_CONV_MODE = os.environ.get("DCA_CONV", "x2")
CONV_X3 = _CONV_MODE != "fp32"
CONV_X2 = _CONV_MODE == "x2"
WGRAD_S2_X2 = os.environ.get("DCA_WGRAD_S2", "x2") != "fp32"   # stride-2 / transposed weight gradient on the f16x2 split (A/B)
DECONV_X3 = os.environ.get("DCA_DECONV", "x3") != "fp32"
C1_WGRAD_FUSED = os.environ.get("DCA_C1_WGRAD", "fused") != "expand"   # logit heads: weight gradient without the 27-plane tensor (A/B)
CONV_S2_X2 = os.environ.get("DCA_CONV_S2", "x2") != "fp32"     # the stride-2 convolution itself on the f16x2 split (A/B)
BN_FUSE = os.environ.get("DCA_BN_FUSE", "1") != "0"        # BatchNorm batch statistics from the conv epilogue (training)   # the transposed-convolution member of the family alone (A/B timing)
_X3_MIN_WORKGROUPS = 1


# ---- operand scales of the f16x2 kernels --------------------------------------------------------------------------------
# Every CHANNEL of an operand is scaled by its own power of two (csrc/dca_common.h).  What travels with a tensor object:
#   t._dca_cmax = (slots, nslots, version): per-channel maxima the kernel that WROTE t emitted (BatchNorm apply / backward,
#                 the f16x2 convolution's fused epilogue): slots[c * CSLOTS + s], s < nslots.  No zero-initialisation.
#   t._dca_exps = (exps, version): the per-channel scale exponents (C ints) once somebody derived them (the weight-packing
#                 kernel of the first convolution over t does, on the way).
#   t._dca_px2  = (exps,): t is NOT fp32 data but the packed px2 image of an fp32 tensor of t's shape (the two scaled f16
#                 terms, [term][C/8][D][H][W][8]), written by a BatchNorm kernel for the f16x2 convolution that consumes it.
# `version` = t._version at tagging time: an in-place change of t invalidates the tag.  Tensors without a tag get one read
# pass (dca_cmax_f32).
AMAX_STATS = {"tagged": 0, "computed": 0, "packed": 0}
CSLOTS = 1024                                              # DCA_AMAX_CSLOTS of include/dca_hip.h
AMAX_EMIT = os.environ.get("DCA_AMAX_EMIT", "1") != "0"    # 0: no producer-side maxima, every operand gets its read pass (A/B)
PACK = os.environ.get("DCA_PACK", "1") != "0"              # 0: no packed px2 operands, fp32 tensors everywhere (A/B)


def _cslots(C, device):
    """uninitialised per-channel slot words for a tensor with C channels (None when producer-side maxima are switched off)"""
    if not AMAX_EMIT:
        return None
    return torch.empty((C * CSLOTS,), device=device, dtype=torch.int32)


def _ver(t):
    """version counter of t, or None for an inference tensor (torch.inference_mode(): such a tensor does not track
    versions -- and cannot be changed in place outside inference mode, so its tag stays valid)"""
    return None if t.is_inference() else t._version


def _tag_cmax(t, slots, nslots):
    if slots is not None:
        t._dca_cmax = (slots, int(nslots), _ver(t))
    return t


def _tag_px2(t, exps):
    t._dca_px2 = (exps,)
    return t


def _copy_tags(src, dst):
    """the operand tags of src on dst, a view of the same values (the alias output of _Conv3d)"""
    for k in ("_dca_cmax", "_dca_exps", "_dca_twin"):
        v = getattr(src, k, None)
        if v is not None:
            setattr(dst, k, v[:-1] + (_ver(dst),))


def _is_packed(t):
    return getattr(t, "_dca_px2", None) is not None


def _twin_of(t):
    """the packed px2 twin of the fp32 tensor t (written together with t by the BatchNorm apply pass), or None"""
    tw = getattr(t, "_dca_twin", None)
    if tw is not None and tw[0] is not None and tw[1] == _ver(t) and tw[0].device == t.device:
        return tw[0]
    return None


def _slots_of(t):
    """(slots, nslots) of the fp32 tensor t (N, C, ...): the producer's if t carries valid ones, else one read pass"""
    tag = getattr(t, "_dca_cmax", None)
    if tag is not None and tag[2] == _ver(t) and tag[0].device == t.device:
        AMAX_STATS["tagged"] += 1
        return tag[0], tag[1]
    AMAX_STATS["computed"] += 1
    N, C = t.shape[0], t.shape[1]
    S = t[0, 0].numel()
    slots = torch.empty((C * CSLOTS,), device=t.device, dtype=torch.int32)
    lib = _L()
    _chk(lib.dca_cmax_f32(_ptr(t), N, C, S, _ptr(slots), _stream()), "dca_cmax_f32")
    nslots = lib.dca_bn_num_chunks(C, S)
    t._dca_cmax = (slots, nslots, _ver(t))
    return slots, nslots


def _exps_cached(t):
    if _is_packed(t):
        return t._dca_px2[0]
    tag = getattr(t, "_dca_exps", None)
    if tag is not None and tag[1] == _ver(t) and tag[0].device == t.device:
        return tag[0]
    return None


def _exps_of(t):
    """per-channel scale exponents (C ints on the device) of operand t: packed tensor -> its own; cached; else from the slots"""
    ex = _exps_cached(t)
    if ex is not None:
        return ex
    slots, nslots = _slots_of(t)
    C = t.shape[1]
    ex = torch.empty((C,), device=t.device, dtype=torch.int32)
    _chk(_L().dca_cmax_exps(_ptr(slots), nslots, C, _ptr(ex), _stream()), "dca_cmax_exps")
    t._dca_exps = (ex, _ver(t))
    return ex


def pack_x2(x):
    """the packed px2 image of an fp32 tensor (N, C % 8 == 0, D, H, W) with exponents from its per-channel maxima (tests,
    micro-benchmarks; in the network the BatchNorm kernels write this format themselves)"""
    x = _req(x, "pack_x2")
    N, C = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    ex = _exps_of(x)
    xp = torch.empty_like(x)
    _chk(_L().dca_bn_apply_pack(_ptr(x), None, _ptr(ex), _ptr(xp), N, C, S, 1.0, None, None, None, None, None, _stream()),
         "dca_bn_apply_pack")
    return _tag_px2(xp, ex)


def _x3_eligible(x, x2, ksize, stride, transposed, A, B):
    if not CONV_X3 or ksize != 3 or stride != 1 or transposed or x2 is not None:
        return False
    N, _, D, H, W = x.shape
    tiles = N * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 15) // 16) * ((B + 31) // 32)
    return tiles >= _X3_MIN_WORKGROUPS and max(A, B) * D * H * W * 4 < 0x7ffffff0


def _dx3_eligible(x, x2, ksize, stride, transposed, A, B):
    """transposed 3x3x3 stride-2 convs on the bf16x3 kernel of deconv3d_x3.hip: <= 32 output channels, coarse width
    divisible by 4, 16-byte aligned input"""
    if not CONV_X3 or not DECONV_X3 or not transposed or ksize != 3 or stride != 2 or x2 is not None or B > 32 or A > 64:
        return False
    N, _, D, H, W = x.shape
    return W % 4 == 0 and x.data_ptr() % 16 == 0 and max(A, 8) * D * H * W * 4 < 0x7ffffff0 and 256 * D * H * W * 4 < 0x7ffffff0


def _s2x2_eligible(x, x2, ksize, stride, transposed, A, B, scale, res_pre, slope):
    """3x3x3 stride-2 convs on the f16x2 kernel of conv3d_s2_f16x2.hip: plain output or the inference epilogue (folded
    BatchNorm, activation, res_post; no res_pre), fine width divisible by 4, 16-byte aligned input, enough tiles to fill the
    chip (small volumes stay on the fp32 MFMA kernel)"""
    if not (CONV_X2 and CONV_X3 and CONV_S2_X2) or ksize != 3 or stride != 2 or transposed or x2 is not None:
        return False
    if res_pre is not None or A > 256 or A <= 4:
        return False
    N, _, D, H, W = x.shape
    Do, Ho, Wo = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    tiles = N * ((Do + 1) // 2) * ((Ho + 3) // 4) * ((Wo + 31) // 32) * ((B + 63) // 64)
    return (W % 4 == 0 and x.data_ptr() % 16 == 0 and tiles >= _X3_MIN_WORKGROUPS
            and (A + 3) * D * H * W * 4 < 0x7ffffff0 and (B + 63) * Do * Ho * Wo * 4 < 0x7ffffff0)


def _c1x3_eligible(x, x2, ksize, A, C1, y):
    """1x1x1 convs on the bf16x3 kernel of conv1_x3.hip (fp32-grade, LDS-free): channel layouts it is built for, voxel
    count divisible by 4, 16-byte aligned tensors"""
    if not CONV_X3 or ksize != 1:
        return False
    C2 = 0 if x2 is None else x2.shape[1]
    if (C1, C2) not in ((32, 0), (64, 0), (32, 32)) or A != C1 + C2:
        return False
    S = x[0, 0].numel()
    ptrs = [x.data_ptr(), y.data_ptr()] + ([] if x2 is None else [x2.data_ptr()])
    return S % 4 == 0 and all(p % 16 == 0 for p in ptrs) and 64 * S * 4 < 0x7ffffff0


def _slice_width(ksize, stride, transposed, B):
    """output channels one launch of dca_conv3d_forward produces (include/dca_hip.h)"""
    if ksize == 1 or transposed:
        return 32
    return 32 if (stride == 1 and B <= 32) else 64


def _prep_weight(w_src, A, B, K, src_ab, flip, ksize, stride, transposed, b_off=0, Bn=None):
    """wt[tap][Apad][Bpad] for the output-channel slice [b_off, b_off+Bn) (padding rules of include/dca_hip.h)."""
    Bn = B if Bn is None else Bn
    if ksize == 1:
        Apad = 32 if A <= 32 else 64
        Bpad = 32
    else:
        Apad = _round_up(A, 8)
        Bpad = 32 if (transposed or (stride == 1 and Bn <= 32)) else 64
    def build():
        wt = torch.empty((K, Apad, Bpad), device=w_src.device, dtype=torch.float32)
        _chk(_L().dca_conv3d_prep_weight(_ptr(w_src), _ptr(wt), A, Bn, Apad, Bpad, K, int(src_ab), int(flip), B, b_off,
                                         _stream()), "dca_conv3d_prep_weight")
        return wt
    return _memo(("prep", A, B, K, int(src_ab), int(flip), Apad, Bpad, b_off, Bn), (w_src,), build,
                 (0, A, Bn, Apad, Bpad, K, int(src_ab), int(flip), B, b_off)), Apad


def _out_dims(dims, ksize, stride, transposed):
    if ksize == 1 or stride == 1:
        return tuple(dims)
    if transposed:
        return tuple(2 * d for d in dims)
    return tuple((d + 1) // 2 for d in dims)


def _conv_sliced(x, x2, w_src, A, B, K, src_ab, flip, ksize, stride, transposed, scale=None, shift=None, slope=1.0,
                 res_pre=None, res_post=None, want_stats=False, emit_amax=False):
    """y = conv(x [, x2]) with A contraction channels and B output channels, as ceil(B / slice) launches that each
    write their channel slice of y (w_src is the PyTorch weight; src_ab / flip as in dca_conv3d_prep_weight).
    want_stats (no epilogue then): returns (y, part) where part holds the BatchNorm batch-statistics partials of y emitted
    by the convolution kernel itself (B * nchunk * 4 doubles, csrc/bn_fused_stats.h), or (y, None) when the kernel serving
    this shape has no such form."""
    y, part = _conv_sliced_impl(x, x2, w_src, A, B, K, src_ab, flip, ksize, stride, transposed, scale, shift, slope,
                                res_pre, res_post, want_stats, emit_amax)
    return (y, part) if want_stats else y


def _conv_sliced_impl(x, x2, w_src, A, B, K, src_ab, flip, ksize, stride, transposed, scale, shift, slope, res_pre,
                      res_post, want_stats, emit_amax=False):
    """x may be a packed px2 operand (f16x2 kernels only); emit_amax: tag y with its per-channel maxima where the kernel
    serving this shape can emit them (inference chains conv -> conv)"""
    packed = _is_packed(x)
    if packed and not (CONV_X2 and _x3_eligible(x, x2, ksize, stride, transposed, A, B)):
        raise RuntimeError("conv3d: a packed px2 operand can only feed the f16x2 3x3x3 stride-1 convolution")
    N = x.shape[0]
    Di, Hi, Wi = x.shape[2:]
    Do, Ho, Wo = _out_dims((Di, Hi, Wi), ksize, stride, transposed)
    y = torch.empty((N, B, Do, Ho, Wo), device=x.device, dtype=torch.float32)
    C1 = x.shape[1]
    width = _slice_width(ksize, stride, transposed, B)
    lib = _L()
    if CONV_X2 and _x3_eligible(x, x2, ksize, stride, transposed, A, B):
        # the weights are packed per launch: the image folds the operand's per-channel exponents in (dca_hip.h); the
        # packing kernel derives them from the operand's slots on the way and leaves them on the tensor for later users
        # (the weight gradient of this convolution, other convolutions over the same tensor)
        wx = torch.empty((lib.dca_conv3d_x2_weight_bytes(A, B) // 2,), device=x.device, dtype=torch.int16)
        xexps = _exps_cached(x)
        if xexps is not None:
            AMAX_STATS["packed" if packed else "tagged"] += 1
            slots, nslots = None, 0
        else:
            slots, nslots = _slots_of(x)
            xexps = torch.empty((A,), device=x.device, dtype=torch.int32)
        _chk(lib.dca_conv3d_x2_prep_weight(_ptr(w_src), _ptr(wx), A, B, int(src_ab), int(flip), _ptr(slots), nslots,
                                           _ptr(xexps), _stream()), "dca_conv3d_x2_prep_weight")
        if slots is not None:
            x._dca_exps = (xexps, _ver(x))
        if want_stats:
            nchunk = lib.dca_conv3d_x2_stats_chunks(N, B, Di, Hi, Wi)
            part = torch.empty((B * nchunk * 4,), device=x.device, dtype=torch.float64)
            _chk(lib.dca_conv3d_x2_forward_stats(_ptr(x), int(packed), _ptr(xexps), _ptr(wx), _ptr(y), _ptr(part), N, A, B,
                                                 Di, Hi, Wi, _stream()), "dca_conv3d_x2_forward_stats")
            return y, part
        ycm = _cslots(B, x.device) if emit_amax else None
        _chk(lib.dca_conv3d_x2_forward(_ptr(x), int(packed), _ptr(xexps), _ptr(wx), _ptr(y), _ptr(scale), _ptr(shift),
                                       _ptr(res_pre), _ptr(res_post), float(slope), _ptr(ycm), N, A, B, Di, Hi, Wi,
                                       _stream()), "dca_conv3d_x2_forward")
        _tag_cmax(y, ycm, lib.dca_conv3d_x2_stats_chunks(N, B, Di, Hi, Wi))
        return y, None
    if _x3_eligible(x, x2, ksize, stride, transposed, A, B):
        def build_x3():
            w3 = torch.empty((lib.dca_conv3d_x3_weight_bytes(A, B) // 2,), device=x.device, dtype=torch.int16)
            _chk(lib.dca_conv3d_x3_prep_weight(_ptr(w_src), _ptr(w3), A, B, int(src_ab), int(flip), _stream()),
                 "dca_conv3d_x3_prep_weight")
            return w3
        wx = _memo(("x3prep", A, B, int(src_ab), int(flip)), (w_src,), build_x3,
                   (1, A, B, 0, 0, 27, int(src_ab), int(flip), B, 0))
        if want_stats:
            nchunk = lib.dca_conv3d_x3_stats_chunks(N, B, Di, Hi, Wi)
            part = torch.empty((B * nchunk * 4,), device=x.device, dtype=torch.float64)
            _chk(lib.dca_conv3d_x3_forward_stats(_ptr(x), _ptr(wx), _ptr(y), _ptr(part), N, A, B, Di, Hi, Wi, _stream()),
                 "dca_conv3d_x3_forward_stats")
            return y, part
        _chk(lib.dca_conv3d_x3_forward(_ptr(x), _ptr(wx), _ptr(y), _ptr(scale), _ptr(shift), _ptr(res_pre),
                                       _ptr(res_post), float(slope), N, A, B, Di, Hi, Wi, _stream()),
             "dca_conv3d_x3_forward")
        return y, None
    if _s2x2_eligible(x, x2, ksize, stride, transposed, A, B, scale, res_pre, slope):
        wx = torch.empty((lib.dca_conv3d_s2x2_weight_bytes(A, B) // 2,), device=x.device, dtype=torch.int16)
        xexps = _exps_cached(x)
        if xexps is not None:
            AMAX_STATS["tagged"] += 1
            slots, nslots = None, 0
        else:
            slots, nslots = _slots_of(x)
            xexps = torch.empty((A,), device=x.device, dtype=torch.int32)
        _chk(lib.dca_conv3d_s2x2_prep_weight(_ptr(w_src), _ptr(wx), A, B, int(src_ab), int(flip), _ptr(slots), nslots,
                                             _ptr(xexps), _stream()), "dca_conv3d_s2x2_prep_weight")
        if slots is not None:
            x._dca_exps = (xexps, _ver(x))
        nsl = lib.dca_conv3d_s2x2_out_slots(N, B, Di, Hi, Wi)
        ycm = _cslots(B, x.device) if (emit_amax and nsl <= CSLOTS) else None
        _chk(lib.dca_conv3d_s2x2_forward(_ptr(x), _ptr(xexps), _ptr(wx), _ptr(y), _ptr(scale), _ptr(shift), float(slope),
                                         _ptr(res_post), _ptr(ycm), N, A, B, Di, Hi, Wi, _stream()), "dca_conv3d_s2x2_forward")
        if ycm is not None:
            _tag_cmax(y, ycm, nsl)
        return y, None
    if _dx3_eligible(x, x2, ksize, stride, transposed, A, B):
        def build_dx3():
            w3 = torch.empty((lib.dca_conv3d_x3_weight_bytes(A, B) // 2,), device=x.device, dtype=torch.int16)
            _chk(lib.dca_conv3d_x3_prep_weight(_ptr(w_src), _ptr(w3), A, B, int(src_ab), int(flip), _stream()),
                 "dca_conv3d_x3_prep_weight")
            return w3
        wx = _memo(("x3prep", A, B, int(src_ab), int(flip)), (w_src,), build_dx3,
                   (1, A, B, 0, 0, 27, int(src_ab), int(flip), B, 0))
        if want_stats:
            nchunk = lib.dca_deconv3d_x3_stats_chunks(N, Di, Hi, Wi)
            part = torch.empty((B * nchunk * 4,), device=x.device, dtype=torch.float64)
            _chk(lib.dca_deconv3d_x3_forward_stats(_ptr(x), _ptr(wx), _ptr(y), _ptr(part), N, A, B, Di, Hi, Wi,
                                                   _stream()), "dca_deconv3d_x3_forward_stats")
            return y, part
        _chk(lib.dca_deconv3d_x3_forward(_ptr(x), _ptr(wx), _ptr(y), _ptr(scale), _ptr(shift), _ptr(res_pre),
                                         _ptr(res_post), float(slope), N, A, B, Di, Hi, Wi, _stream()),
             "dca_deconv3d_x3_forward")
        return y, None
    if _c1x3_eligible(x, x2, ksize, A, C1, y):
        S = Do * Ho * Wo
        C2 = 0 if x2 is None else x2.shape[1]
        part = None
        if want_stats:
            nchunk = lib.dca_conv1_x3_stats_chunks(N, S)
            part = torch.empty((B * nchunk * 4,), device=x.device, dtype=torch.float64)
        for b0 in range(0, B, 32):
            bn = min(32, B - b0)

            def build_c1(b0=b0, bn=bn):
                wf = torch.empty((lib.dca_conv1_x3_weight_bytes(A) // 2,), device=x.device, dtype=torch.int16)
                _chk(lib.dca_conv1_x3_prep_weight(_ptr(w_src), _ptr(wf), A, bn, int(src_ab), B, b0, _stream()),
                     "dca_conv1_x3_prep_weight")
                return wf
            wf = _memo(("c1x3prep", A, B, int(src_ab), b0, bn), (w_src,), build_c1,
                       (2, A, bn, 0, 0, 1, int(src_ab), 0, B, b0))
            if part is not None:
                _chk(lib.dca_conv1_x3_forward_stats(_ptr(x), _ptr(x2), _ptr(wf), _ptr(y), _ptr(part), N, C1, C2, bn, B, b0,
                                                    S, _stream()), "dca_conv1_x3_forward_stats")
            else:
                _chk(lib.dca_conv1_x3_forward(_ptr(x), _ptr(x2), _ptr(wf), _ptr(y), _ptr(scale), _ptr(shift),
                                              _ptr(res_pre), _ptr(res_post), float(slope), N, C1, C2, bn, B, b0, S,
                                              _stream()), "dca_conv1_x3_forward")
        return y, part
    for b0 in range(0, B, width):
        bn = min(width, B - b0)
        wt, Apad = _prep_weight(w_src, A, B, K, src_ab, flip, ksize, stride, transposed, b0, bn)
        _chk(lib.dca_conv3d_forward(_ptr(x), _ptr(x2), _ptr(wt), _ptr(y), _ptr(scale), _ptr(shift), _ptr(res_pre),
                                    _ptr(res_post), float(slope), N, A, C1, bn, Apad, B, b0, Di, Hi, Wi, Do, Ho, Wo,
                                    ksize, stride, int(transposed), _stream()), "dca_conv3d_forward")
    return y, None


def conv3d_prepared(x, wt, A, Apad, B, ksize, stride, transposed):
    """single launch with an already laid-out weight (used by bench.py to time the bare kernel)"""
    N = x.shape[0]
    Di, Hi, Wi = x.shape[2:]
    Do, Ho, Wo = _out_dims((Di, Hi, Wi), ksize, stride, transposed)
    y = torch.empty((N, B, Do, Ho, Wo), device=x.device, dtype=torch.float32)
    _chk(_L().dca_conv3d_forward(_ptr(x), None, _ptr(wt), _ptr(y), None, None, None, None, 1.0, N, A, A, B, Apad, B, 0,
                                 Di, Hi, Wi, Do, Ho, Wo, ksize, stride, int(transposed), _stream()),
         "dca_conv3d_forward")
    return y


def _conv_forward_impl(x, x2, weight, stride, transposed, scale=None, shift=None, slope=1.0, res_pre=None,
                       res_post=None, want_stats=False, emit_amax=False):
    ksize = weight.shape[2]
    K = ksize ** 3
    if transposed:
        Cin, Cout = weight.shape[0], weight.shape[1]
        src_ab = 1
    else:
        Cout, Cin = weight.shape[0], weight.shape[1]
        src_ab = 0
    assert x.shape[1] + (x2.shape[1] if x2 is not None else 0) == Cin, "conv3d: channel mismatch"
    return _conv_sliced(x, x2, weight, Cin, Cout, K, src_ab, 0, ksize, stride, transposed, scale, shift, slope,
                        res_pre, res_post, want_stats, emit_amax)


def _wgrad(x, dy, dw_view_ptr_tensor, dst_offset, Cx, Cy, ksize, stride, s_cy, s_cx):
    """dw[cy*s_cy + cx*s_cx + k] (+dst_offset floats) = sum dy[cy] * x[cx] (see dca_hip.h); x / dy may be packed px2
    operands (3x3x3 stride 1 on the f16x2 kernel only)."""
    N = x.shape[0]
    Di, Hi, Wi = x.shape[2:]
    Do, Ho, Wo = dy.shape[2:]
    dst = _vp(dw_view_ptr_tensor.data_ptr() + 4 * dst_offset)
    lib = _L()
    xp, yp = _is_packed(x), _is_packed(dy)
    if (CONV_X2 and CONV_X3 and ksize == 3 and stride == 1 and (xp or Wi % 4 == 0) and (yp or Wi % 4 == 0)
            and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and max(Cx, Cy) * Di * Hi * Wi * 4 < 0x7ffffff0):
        xex, yex = _exps_of(x), _exps_of(dy)
        nws = lib.dca_conv3d_wgrad_x2_workspace(N, Cx, Cy, Di, Hi, Wi)
        part = torch.empty((nws,), device=x.device, dtype=torch.float32)
        _chk(lib.dca_conv3d_wgrad_x2(_ptr(x), int(xp), _ptr(xex), _ptr(dy), int(yp), _ptr(yex), _ptr(part), dst, N, Cx, Cy,
                                     Di, Hi, Wi, s_cy, s_cx, _stream()), "dca_conv3d_wgrad_x2")
        return
    if xp or yp:
        raise RuntimeError("weight gradient: a packed px2 operand can only feed the f16x2 3x3x3 stride-1 kernel")
    if (CONV_X3 and ksize == 3 and stride == 1 and Wi % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
            and max(Cx, Cy) * Di * Hi * Wi * 4 < 0x7ffffff0):
        nws = lib.dca_conv3d_wgrad_x3_workspace(N, Cx, Cy, Di, Hi, Wi)
        part = torch.empty((nws,), device=x.device, dtype=torch.float32)
        _chk(lib.dca_conv3d_wgrad_x3(_ptr(x), _ptr(dy), _ptr(part), dst, N, Cx, Cy, Di, Hi, Wi, s_cy, s_cx, _stream()),
             "dca_conv3d_wgrad_x3")
        return
    if (CONV_X2 and CONV_X3 and WGRAD_S2_X2 and ksize == 3 and stride == 2 and Wi % 4 == 0 and Wo % 4 == 0
            and (Do, Ho, Wo) == ((Di + 1) // 2, (Hi + 1) // 2, (Wi + 1) // 2)
            and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and max(Cx, Cy) * Di * Hi * Wi * 4 < 0x7ffffff0):
        # stride-2 convolution / transposed convolution: x = the fine tensor, dy = the coarse one (conv3d_wgrad_s2_f16x2.hip)
        xex, yex = _exps_of(x), _exps_of(dy)
        nws = lib.dca_conv3d_wgrad_s2_x2_workspace(N, Cx, Cy, Di, Hi, Wi)
        part = torch.empty((nws,), device=x.device, dtype=torch.float32)
        _chk(lib.dca_conv3d_wgrad_s2_x2(_ptr(x), _ptr(xex), _ptr(dy), _ptr(yex), _ptr(part), dst, N, Cx, Cy, Di, Hi, Wi,
                                        s_cy, s_cx, _stream()), "dca_conv3d_wgrad_s2_x2")
        return
    nws = lib.dca_conv3d_wgrad_workspace(N, Cx, Cy, Do, Ho, Wo, ksize, stride)
    part = torch.empty((nws,), device=x.device, dtype=torch.float32)
    _chk(lib.dca_conv3d_wgrad(_ptr(x), _ptr(dy), _ptr(part), dst, N, Cx, Cy, Di, Hi, Wi, Do, Ho, Wo, ksize, stride,
                              s_cy, s_cx, _stream()), "dca_conv3d_wgrad")


class _Conv3dC1(torch.autograd.Function):
    """nn.Conv3d(C, 1, 3, padding=1, bias=False): the logit heads.  The 27 taps become a GEMM axis so forward and
    weight gradient run on the 1x1x1 matrix-core kernels (see include/dca_hip.h); C must be 32 or 64."""

    @staticmethod
    def forward(ctx, x, weight):
        x, weight = _req(x, "conv3d"), _req(weight, "conv3d.weight")
        N, C, D, H, W = x.shape
        with torch.cuda.device_of(x):
            T = _conv_sliced(x, None, weight, C, 27, 1, 1, 0, 1, 1, False)      # wt[ci][tap] = w[0, ci, tap]; (N,27,D,H,W)
            y = torch.empty((N, 1, D, H, W), device=x.device, dtype=torch.float32)
            _chk(_L().dca_conv3d_c1_gather(_ptr(T), _ptr(y), N, D, H, W, _stream()), "dca_conv3d_c1_gather")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _req(dy, "conv3d.backward")
        N, C, D, H, W = x.shape
        gx = gw = None
        lib = _L()
        with torch.cuda.device_of(x):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _chk(lib.dca_conv3d_c1_bwd_data(_ptr(dy), _ptr(weight), _ptr(gx), N, C, D, H, W, _stream()),
                     "dca_conv3d_c1_bwd_data")
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(weight)
                if (C1_WGRAD_FUSED and W % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
                        and x.numel() * 4 < 0x7ffffff0):
                    # the tap-shifted views of dy are built inside the weight-gradient kernel: no 27-plane tensor
                    nws = lib.dca_conv3d_wgrad_workspace(N, C, 27, D, H, W, 1, 1)
                    part = torch.empty((nws,), device=x.device, dtype=torch.float32)
                    _chk(lib.dca_conv3d_c1_wgrad(_ptr(x), _ptr(dy), _ptr(part), _ptr(gw), N, C, D, H, W, _stream()),
                         "dca_conv3d_c1_wgrad")
                else:
                    G = torch.empty((N, 27, D, H, W), device=x.device, dtype=torch.float32)
                    _chk(lib.dca_conv3d_c1_expand(_ptr(dy), _ptr(G), N, D, H, W, _stream()), "dca_conv3d_c1_expand")
                    _wgrad(x, G, gw, 0, C, 27, 1, 1, 1, 27)                      # dw[ci*27 + tap]
        return gx, gw


class _Conv3d(torch.autograd.Function):
    """Conv3d (k=3 pad=1 stride 1|2, or k=1) / ConvTranspose3d (k=3 s=2 p=1 op=1), bias=False.
    Optional second input x2 = implicit channel concat for the 1x1x1 `fuse` conv."""

    @staticmethod
    def forward(ctx, x, x2, weight, stride, transposed, want_stats=False, alias=False, packed_dy=False):
        """want_stats: returns (y, part) -- part = the BatchNorm batch-statistics partials of y from the convolution
        kernel's own epilogue (not differentiable; csrc/bn_fused_stats.h), empty where the kernel serving this shape
        cannot produce them.
        alias (3x3x3 stride 1 only): one more output, x itself, for the OTHER consumers of x -- their summed gradient comes
        back as that output's gradient and is added inside this convolution's backward-data launch (epilogue `+ res_post`)
        instead of by autograd's separate accumulation pass."""
        x, weight = _req(x, "conv3d", packed_ok=True), _req(weight, "conv3d.weight")
        x2 = _opt(x2, "conv3d.x2")
        if _is_packed(x) and alias:
            raise RuntimeError("conv3d: alias output of a packed px2 operand")
        ctx.save_for_backward(x, x2, weight)
        ctx.meta = (stride, transposed)
        ctx.alias = bool(alias)
        ctx.set_materialize_grads(False)             # an unused output's gradient arrives as None, not as a tensor of zeros
        ctx.x_px2 = getattr(x, "_dca_px2", None)    # save_for_backward keeps the tensor, not its Python attributes
        ctx.packed_dy = bool(packed_dy)              # the gradient of y arrives as a packed px2 operand (_BnAct, pack_dy)
        ctx.x_exps = None
        # a packed twin of x (written beside it by its BatchNorm): this convolution and its weight gradient read the twin
        xt = None
        if (PACK and CONV_X2 and x2 is None and not transposed and stride == 1 and weight.shape[2] == 3 and weight.shape[0] > 1
                and _x3_eligible(x, None, 3, 1, False, weight.shape[1], weight.shape[0])):
            xt = _twin_of(x)
        ctx.x_twin = xt
        xin = x if xt is None else xt
        with torch.cuda.device_of(x):
            if not want_stats:
                y = _conv_forward_impl(xin, x2, weight, stride, transposed)
                ctx.x_exps = _exps_cached(xin)         # forward and weight gradient scale x by the same exponents
                if alias:
                    xa = x.view_as(x)
                    _copy_tags(x, xa)
                    return y, xa
                return y
            y, part = _conv_forward_impl(xin, x2, weight, stride, transposed, want_stats=True)
            ctx.x_exps = _exps_cached(xin)
        if part is None:
            part = torch.empty((0,), device=x.device, dtype=torch.float64)
        ctx.mark_non_differentiable(part)
        if alias:
            xa = x.view_as(x)
            _copy_tags(x, xa)
            return y, part, xa
        return y, part

    @staticmethod
    def backward(ctx, dy, *rest):
        x, x2, weight = ctx.saved_tensors
        xw = x                                      # the operand of the weight gradient
        if ctx.x_twin is not None:
            xw = ctx.x_twin                         # (tagged px2 when it was written)
        elif ctx.x_px2 is not None:
            x._dca_px2 = ctx.x_px2
        elif ctx.x_exps is not None and _exps_cached(x) is None:
            x._dca_exps = (ctx.x_exps, _ver(x))
        stride, transposed = ctx.meta
        g_alias = _opt(rest[-1], "conv3d.backward") if (ctx.alias and rest) else None
        if dy is None:       # y was not used: nothing flows through the convolution, only past it (alias)
            return g_alias, None, None, None, None, None, None, None
        if ctx.packed_dy and not _is_packed(dy):
            raise RuntimeError("conv3d.backward: expected the packed px2 gradient of the BatchNorm behind this convolution "
                               "(the tag was lost on the way through autograd)")
        dy = _req(dy, "conv3d.backward", packed_ok=True)
        ksize = weight.shape[2]
        K = ksize ** 3
        gx = gx2 = gw = None
        need_x, need_x2, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        with torch.cuda.device_of(x):
            if transposed:
                Cin, Cout = weight.shape[0], weight.shape[1]
                if need_x:  # stride-2 conv of dy with Wt read as a Conv3d weight [Cin][Cout][K]
                    gx = _conv_sliced(dy, None, weight, Cout, Cin, K, 0, 0, 3, 2, False)
                if need_w:
                    gw = torch.empty_like(weight)
                    _wgrad(dy, x, gw, 0, Cout, Cin, 3, 2, Cout * K, K)
            elif ksize == 3:
                Cout, Cin = weight.shape[0], weight.shape[1]
                if need_x:
                    if stride == 1:   # dy's max-|.| word: tag or one pass; + the other consumers' gradient of x (alias)
                        gx = _conv_sliced(dy, None, weight, Cout, Cin, K, 1, 1, 3, 1, False, res_post=g_alias)
                        g_alias = None
                    else:
                        gx = _conv_sliced(dy, None, weight, Cout, Cin, K, 1, 0, 3, 2, True)
                        if gx.shape != x.shape:
                            raise RuntimeError("stride-2 conv backward needs even input dims")
                if need_w:
                    gw = torch.empty_like(weight)
                    _wgrad(xw, dy, gw, 0, Cin, Cout, 3, stride, Cin * K, K)
            else:
                Cout, Cin = weight.shape[0], weight.shape[1]
                w2 = weight.reshape(Cout, Cin)
                C1 = x.shape[1]
                if Cout not in (32, 64):
                    raise RuntimeError("1x1x1 conv backward-data needs 32 or 64 output channels")
                if need_x:
                    wa = w2[:, :C1].contiguous()
                    gx = _conv_sliced(dy, None, wa, Cout, C1, 1, 1, 0, 1, 1, False)
                if x2 is not None and need_x2:
                    wb = w2[:, C1:].contiguous()
                    gx2 = _conv_sliced(dy, None, wb, Cout, Cin - C1, 1, 1, 0, 1, 1, False)
                if need_w:
                    gw = torch.empty_like(weight)
                    _wgrad(x, dy, gw, 0, C1, Cout, 1, 1, Cin, 1)
                    if x2 is not None:
                        _wgrad(x2, dy, gw, C1, Cin - C1, Cout, 1, 1, Cin, 1)
        if g_alias is not None:      # alias on a path without the fused form
            gx = g_alias if gx is None else gx + g_alias
        return gx, gx2, gw, None, None, None, None, None


class _ConvPair(torch.autograd.Function):
    """Two convolutions of ONE input -- A: 3x3x3 stride 2, B: 1x1x1 (`cost_agg.conv1` and `cost_agg.redir` of
    Multi_Aggregation, models/augment/cva.py:16-23) -- as one autograd node, so that the gradient of the shared input is
    formed inside the second backward-data launch (epilogue `+ res_post`) instead of by autograd's separate accumulation add
    (three passes over a 1/4-resolution tensor).  Same kernels, same values: (a + b) is one fp32 addition either way."""

    @staticmethod
    def forward(ctx, x, wa, wb, want_stats):
        x, wa, wb = _req(x, "conv_pair"), _req(wa, "conv_pair.weight_a"), _req(wb, "conv_pair.weight_b")
        ctx.save_for_backward(x, wa, wb)
        with torch.cuda.device_of(x):
            if want_stats:
                ya, pa = _conv_forward_impl(x, None, wa, 2, False, want_stats=True)
                yb, pb = _conv_forward_impl(x, None, wb, 1, False, want_stats=True)
            else:
                ya, pa = _conv_forward_impl(x, None, wa, 2, False), None
                yb, pb = _conv_forward_impl(x, None, wb, 1, False), None
        none = lambda: torch.empty((0,), device=x.device, dtype=torch.float64)
        pa, pb = (none() if pa is None else pa), (none() if pb is None else pb)
        ctx.mark_non_differentiable(pa, pb)
        return ya, pa, yb, pb

    @staticmethod
    def backward(ctx, dya, _dpa, dyb, _dpb):
        x, wa, wb = ctx.saved_tensors
        dya, dyb = _req(dya, "conv_pair.backward"), _req(dyb, "conv_pair.backward")
        Ca, Cin, Cb = wa.shape[0], wa.shape[1], wb.shape[0]
        if Cb not in (32, 64):
            raise RuntimeError("1x1x1 conv backward-data needs 32 or 64 output channels")
        with torch.cuda.device_of(x):
            gb = _conv_sliced(dyb, None, wb.reshape(Cb, Cin).contiguous(), Cb, Cin, 1, 1, 0, 1, 1, False)
            gx = _conv_sliced(dya, None, wa, Ca, Cin, 27, 1, 0, 3, 2, True, res_post=gb)     # d(x) = A^T dya + B^T dyb
            if gx.shape != x.shape:
                raise RuntimeError("stride-2 conv backward needs even input dims")
            gwa, gwb = torch.empty_like(wa), torch.empty_like(wb)
            _wgrad(x, dya, gwa, 0, Cin, Ca, 3, 2, Cin * 27, 27)
            _wgrad(x, dyb, gwb, 0, Cin, Cb, 1, 1, Cin, 1)
        return gx, gwa, gwb, None


PAIR_FUSE = os.environ.get("DCA_PAIR_FUSE", "1") != "0"


def convbn3d_pair(x, conv_a, bn_a, slope_a, conv_b, bn_b, slope_b, pack_a=False):
    """(act(BN_a(conv_a(x))), act(BN_b(conv_b(x)))) for conv_a = Conv3d(k3, s2, p1), conv_b = Conv3d(k1) over the SAME x;
    training path: one autograd node for the two convolutions (`_ConvPair`), otherwise two `convbn3d` calls.
    pack_a: the first result has one consumer, a 3x3x3 stride-1 convolution (see convbn3d, pack_out)"""
    inference = (not bn_a.training and not bn_b.training and not torch.is_grad_enabled())
    ok = (PAIR_FUSE and not inference and _lp_dtype() is None and conv_a.kernel_size[0] == 3 and conv_a.stride[0] == 2
          and conv_b.kernel_size[0] == 1 and conv_b.weight.shape[0] in (32, 64) and x.dtype == torch.float32
          and not isinstance(conv_a, torch.nn.ConvTranspose3d) and all(d % 2 == 0 for d in x.shape[2:]))
    if not ok:
        return convbn3d(x, conv_a, bn_a, slope_a, pack_out=pack_a), convbn3d(x, conv_b, bn_b, slope_b)
    stats = bool(BN_FUSE and bn_a.training and bn_b.training)
    ya, pa, yb, pb = _ConvPair.apply(x, conv_a.weight, conv_b.weight, stats)
    za = bn_act(ya, bn_a, slope_a, stats_part=pa if pa.numel() else None, pack_out=pack_a)
    zb = bn_act(yb, bn_b, slope_b, stats_part=pb if pb.numel() else None)
    return za, zb


def conv3d(x, weight, stride=1, transposed=False, x2=None):
    if (not transposed and x2 is None and weight.shape[0] == 1 and weight.shape[2] == 3 and int(stride) == 1
            and weight.shape[1] in (32, 64)):
        return _Conv3dC1.apply(x, weight)
    return _Conv3d.apply(x, x2, weight, int(stride), bool(transposed))


def conv3d_fused_inference(x, weight, stride, transposed, scale, shift, slope, res_pre=None, res_post=None, x2=None):
    """Forward-only conv with the affine (folded BatchNorm) + activation + residual epilogue fused."""
    x, weight = _req(x, "conv3d"), _req(weight, "conv3d.weight")
    with torch.cuda.device_of(x):
        return _conv_forward_impl(x, _opt(x2, "x2"), weight, int(stride), bool(transposed), _opt(scale, "scale"),
                                  _opt(shift, "shift"), slope, _opt(res_pre, "res_pre"), _opt(res_post, "res_post"),
                                  emit_amax=CONV_X2 and AMAX_EMIT)


# ------------------------------------------------------------------------------------------------
# BatchNorm3d + activation + residual
# ------------------------------------------------------------------------------------------------
def bn_stats_vector(y, gamma, beta, running_mean, running_var, training, momentum, eps, part=None, zexps=None,
                    rpre=None, rpost=None):
    """[mean | invstd | scale | shift] (4*C floats); updates the running stats in place when training.
    part: partial statistics the producing convolution already emitted (dca_*_forward_stats), or None.
    zexps (training only): C ints that receive the scale exponents of z = act(BN(y) + res_pre) + res_post for the packed px2
    output; rpre / rpost = (slots, nslots) of the residual tensors (their per-channel maxima enter the bound)."""
    rps, rpn = rpre if rpre is not None else (None, 0)
    rqs, rqn = rpost if rpost is not None else (None, 0)
    N, C = y.shape[0], y.shape[1]
    S = y[0, 0].numel()
    stats = torch.empty((4 * C,), device=y.device, dtype=torch.float32)
    lib = _L()
    if training:
        if part is not None:     # one self-centred partial {K, n, s, q} per (channel, workgroup) of the producing conv
            _chk(lib.dca_bn_finalize_centered(_ptr(part), part.numel() // (4 * C), _ptr(gamma), _ptr(beta),
                                              _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
                                              _ptr(stats), _ptr(zexps), _ptr(rps), rpn, _ptr(rqs), rqn, C, _stream()),
                 "dca_bn_finalize_centered")
            return stats
        nchunk = lib.dca_bn_num_chunks(C, S)
        part = torch.empty((C * nchunk * 2 + C,), device=y.device, dtype=torch.float64)   # partial sums + C shifts
        _chk(lib.dca_bn_stats(_ptr(y), _ptr(part), N, C, S, _stream()), "dca_bn_stats")
        _chk(lib.dca_bn_finalize(_ptr(part), nchunk, float(N * S), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                 _ptr(running_var), float(momentum), float(eps), 1, _ptr(stats), _ptr(zexps), _ptr(rps), rpn,
                                 _ptr(rqs), rqn, C, _stream()), "dca_bn_finalize")
    else:
        assert zexps is None
        _chk(lib.dca_bn_finalize(None, 0, float(N * S), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                 _ptr(running_var), float(momentum), float(eps), 0, _ptr(stats), None, None, 0, None, 0, C,
                                 _stream()), "dca_bn_finalize")
    return stats


def bn_eval_affine(bn):
    """[mean | invstd | scale | shift] of an eval-mode BatchNorm (running statistics), 4*C floats."""
    C = bn.num_features

    def build():
        stats = torch.empty((4 * C,), device=bn.running_mean.device, dtype=torch.float32)
        _chk(_L().dca_bn_finalize(None, 0, 1.0, _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean),
                                  _ptr(bn.running_var), 0.1, float(bn.eps), 0, _ptr(stats), None, None, 0, None, 0, C, _stream()),
             "dca_bn_finalize")
        return stats
    src = tuple(t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None)
    return _memo(("bnfold", float(bn.eps)), src, build)


class _BnAct(torch.autograd.Function):
    """z = act(BN(y) + res_pre) + res_post with nn.BatchNorm3d semantics.
    pack_z:  1: z is written in the packed px2 operand format (for ONE consumer: an f16x2 convolution) instead of fp32;
             2: both -- the fp32 z for all readers and a packed twin (it rides on z: _dca_twin) for the f16x2 convolution
             among them and its weight gradient.
    pack_dy: backward writes the gradient of y in the packed px2 format (y's producer is an f16x2 convolution: its
             backward-data and weight-gradient kernels are the only readers)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, slope, res_pre, res_post,
                part=None, zmax=None, pack_z=False, pack_dy=False):
        """zmax: per-channel slot words that receive max |z| (see _cslots); the caller tags z with them"""
        y = _req(y, "batch_norm")
        res_pre, res_post = _opt(res_pre, "res_pre"), _opt(res_post, "res_post")
        N, C = y.shape[0], y.shape[1]
        S = y[0, 0].numel()
        lib = _L()
        pack_z = int(pack_z) if (training and C % 8 == 0) else 0
        pack_dy = bool(pack_dy and res_pre is None and C % 8 == 0)
        with torch.cuda.device_of(y):
            zexps = torch.empty((C,), device=y.device, dtype=torch.int32) if pack_z else None
            rpre = _slots_of(res_pre) if (pack_z and res_pre is not None) else None
            rpost = _slots_of(res_post) if (pack_z and res_post is not None) else None
            stats = bn_stats_vector(y, gamma, beta, running_mean, running_var, training, momentum, eps, part, zexps, rpre, rpost)
            ymax = torch.empty((C * CSLOTS,), device=y.device, dtype=torch.int32) if pack_dy else None
            z = torch.empty_like(y)
            _tls.last_twin = None
            if pack_z:
                zp = z if pack_z == 1 else torch.empty_like(y)
                _chk(lib.dca_bn_apply_pack(_ptr(y), _ptr(stats), _ptr(zexps), _ptr(zp), N, C, S, float(slope), _ptr(ymax),
                                           _ptr(res_pre), _ptr(res_post), _ptr(z) if pack_z == 2 else None,
                                           _ptr(zmax) if pack_z == 2 else None, _stream()), "dca_bn_apply_pack")
                ymax_slots = lib.dca_bn_pack_chunks(C, S)
                if pack_z == 2:
                    _tls.last_twin = _tag_px2(zp, zexps)
            else:
                _chk(lib.dca_bn_apply(_ptr(y), _ptr(stats), _ptr(res_pre), _ptr(res_post), _ptr(z), N, C, S, float(slope),
                                      _ptr(zmax), _ptr(ymax), _stream()), "dca_bn_apply")
                ymax_slots = lib.dca_bn_num_chunks(C, S)
        ctx.save_for_backward(y, stats, res_pre if slope != 1.0 else None, ymax)
        ctx.meta = (training, slope, res_pre is not None, res_post is not None, pack_dy, ymax_slots)
        _tls.last_zexps = zexps       # picked up by bn_act right after apply() (same thread, synchronous)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, stats, res_pre, ymax = ctx.saved_tensors
        training, slope, has_pre, has_post, pack_dy, ymax_slots = ctx.meta
        dz = _req(dz, "batch_norm.backward")
        N, C = y.shape[0], y.shape[1]
        S = y[0, 0].numel()
        lib = _L()
        with torch.cuda.device_of(y):
            nchunk = lib.dca_bn_num_chunks(C, S)
            part = torch.empty((C * nchunk * 2,), device=y.device, dtype=torch.float64)
            dgb = torch.empty((4 * C,), device=y.device, dtype=torch.float32)
            dy = torch.empty_like(y)
            g_out = None
            if pack_dy:
                dyexps = torch.empty((C,), device=y.device, dtype=torch.int32)
                gmax = torch.empty((C * CSLOTS,), device=y.device, dtype=torch.int32)
                _chk(lib.dca_bn_backward_pack(_ptr(dz), _ptr(y), _ptr(stats), _ptr(part), _ptr(dgb), _ptr(dy), _ptr(dyexps),
                                              _ptr(gmax), _ptr(ymax), ymax_slots if ymax is not None else 0, N, C, S,
                                              float(slope), int(training), _stream()), "dca_bn_backward_pack")
                _tag_px2(dy, dyexps)
            else:
                want_g = has_pre and slope != 1.0 and ctx.needs_input_grad[9]
                g_out = torch.empty_like(y) if want_g else None
                dm = _cslots(C, y.device) if CONV_X2 else None     # per-channel max |dy| for the convolution's backward kernels
                _chk(lib.dca_bn_backward(_ptr(dz), _ptr(y), _ptr(res_pre), _ptr(stats), _ptr(part), _ptr(dgb), _ptr(dy),
                                         _ptr(g_out), N, C, S, float(slope), int(training), _ptr(dm), _stream()),
                     "dca_bn_backward")
                _tag_cmax(dy, dm, nchunk)
        g_pre = None
        if has_pre and ctx.needs_input_grad[9]:
            g_pre = g_out if g_out is not None else dz
        g_post = dz if (has_post and ctx.needs_input_grad[10]) else None
        return dy, dgb[:C], dgb[C:2 * C], None, None, None, None, None, None, g_pre, g_post, None, None, None, None


_tls = threading.local()


class frozen_weights:
    """Inference helper: inside this context the caller promises that parameters and BatchNorm buffers do not change, so
    the per-call weight re-layouts (dca_conv3d_prep_weight / dca_conv3d_x3_prep_weight) and folded eval-mode BatchNorm
    affines are computed once per tensor and reused (the usual weight pre-packing of inference engines; ~80 tiny launches
    = 0.5 ms of an 8.5 ms forward at 544x960).  Never active outside the context, per thread, dropped at exit; bypassed
    while a stream capture is running (a graph must contain its own prep kernels)."""

    def __enter__(self):
        self.prev = getattr(_tls, "frozen", None)
        _tls.frozen = {}
        return self

    def __exit__(self, *exc):
        _tls.frozen = self.prev
        return False


_PLAN_RECORDER = None   # PrepackPlan being recorded (process-wide, see the class)
_PLAN_ACTIVE = None     # {cache key: packed tensor} of the active PrepackPlan


class PrepackPlan:
    """Training helper: the weight re-layouts of a whole step as ONE launch (dca_conv3d_prep_many).

        plan = ops.PrepackPlan()
        with plan.recording():          # one ordinary step: notes every (parameter, layout) the step asks for
            step()
        plan.finalize()
        loop:   with plan.active(): step_forward_backward(); optimizer.step(); plan.refresh()

    `refresh()` re-packs all recorded layouts from the current parameter values; inside `active()` the convolutions pick
    the pre-packed images up instead of launching their own prep kernels.  Only layouts whose source is a leaf tensor
    that owns its storage (a parameter) are planned; anything else keeps packing per call.  The recording / active
    state is per PROCESS, not per thread: backward-data layouts are requested from the autograd engine's thread."""

    def __init__(self):
        self.entries = {}      # cache key -> (packed tensor, descriptor tuple, source tensors)
        self.table = None
        self.n = 0

    def recording(self):
        plan = self

        class _Rec:
            def __enter__(self_inner):
                global _PLAN_RECORDER
                self_inner.prev, _PLAN_RECORDER = _PLAN_RECORDER, plan

            def __exit__(self_inner, *exc):
                global _PLAN_RECORDER
                _PLAN_RECORDER = self_inner.prev
                return False
        return _Rec()

    def finalize(self):
        import struct
        rows = []
        for out, desc, tensors in self.entries.values():
            kind, A, Bn, Apad, Bpad, K, src_ab, flip, Btotal, b_off = desc
            nch = (A + 15) // 16
            total = out.numel() - (8 if kind == 3 else 0)    # kind 3: f16 elements in front of the 16-byte scale tail
            rows.append(struct.pack("<QQ12iq", tensors[0].data_ptr(), out.data_ptr(), kind, A, Bn, Apad, Bpad, K, src_ab,
                                    flip, Btotal, b_off, nch, 0, total))
        self.n = len(rows)
        if self.n:
            dev = next(iter(self.entries.values()))[0].device
            self.table = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(dev)
        return self

    def refresh(self):
        if self.n:
            _chk(_L().dca_conv3d_prep_many(_ptr(self.table), self.n, _stream()), "dca_conv3d_prep_many")

    def active(self):
        plan = self

        class _Act:
            def __enter__(self_inner):
                global _PLAN_ACTIVE
                self_inner.prev = _PLAN_ACTIVE
                _PLAN_ACTIVE = {k: out for k, (out, _, _t) in plan.entries.items()}

            def __exit__(self_inner, *exc):
                global _PLAN_ACTIVE
                _PLAN_ACTIVE = self_inner.prev
                return False
        return _Act()


def _memo(key, tensors, build, desc=None):
    """build() once per (key, identity of `tensors`) inside a frozen_weights() / PrepackPlan.active() context; plain
    build() otherwise (a PrepackPlan being recorded also notes `desc`, the dca_conv3d_prep_many descriptor)"""
    rec = _PLAN_RECORDER
    if rec is not None:
        out = build()
        t = tensors[0] if tensors else None
        if desc is not None and t is not None and t.is_leaf and t._base is None:
            rec.entries[(key,) + tuple((id(x), x.data_ptr()) for x in tensors)] = (out, desc, tensors)
        return out
    act = _PLAN_ACTIVE
    if act is not None and desc is not None:
        hit = act.get((key,) + tuple((id(x), x.data_ptr()) for x in tensors))
        if hit is not None:
            return hit
    cache = getattr(_tls, "frozen", None)
    if cache is None or (tensors and tensors[0].is_cuda and torch.cuda.is_current_stream_capturing()):
        return build()
    k = (key,) + tuple((id(t), t.data_ptr()) for t in tensors)
    hit = cache.get(k)
    if hit is None:
        hit = cache[k] = (build(), tensors)   # keeps the source tensors alive, so ids cannot be recycled
    return hit[0]


class batched_bn_counters:
    """Inside this context the `num_batches_tracked += 1` of every train-mode BatchNorm (nn.BatchNorm3d semantics) is
    deferred and applied at exit as ONE multi-tensor add instead of ~50 one-element kernel launches per step."""

    def __enter__(self):
        self.prev = getattr(_tls, "pending", None)
        _tls.pending = []
        return self

    def __exit__(self, *exc):
        pending, _tls.pending = _tls.pending, self.prev
        if pending:
            torch._foreach_add_(pending, 1)
        return False


def bn_act(y, bn, slope=1.0, res_pre=None, res_post=None, stats_part=None, pack_out=False, pack_dy=False):
    """Applies the nn.BatchNorm3d module `bn` (parameters/buffers only; its forward is never called).
    stats_part: batch-statistics partial sums of y from the producing convolution (`_Conv3d` with want_stats), if it made them.
    pack_out: True -- the result has ONE consumer, an f16x2 3x3x3 stride-1 convolution: write it in the packed px2 operand
    format instead of fp32; "both" -- several consumers, ONE of them such a convolution: fp32 result plus a packed twin that
    this convolution and its weight gradient pick up (training BatchNorm only; plain fp32 otherwise).  pack_dy: see _BnAct."""
    momentum = 0.1 if bn.momentum is None else bn.momentum
    training = bn.training or bn.running_mean is None
    if _lp_dtype() is not None:
        raise RuntimeError("ops.reduced_precision is inference only: call the model in eval mode under torch.no_grad()")
    C = y.shape[1]
    pack_z = 0
    if pack_out and PACK and CONV_X2 and training and C % 8 == 0 and torch.is_grad_enabled():
        pack_z = 2 if pack_out == "both" else 1
    pack_dy = bool(pack_dy and PACK and CONV_X2 and res_pre is None and C % 8 == 0)
    zm = _cslots(C, y.device) if (CONV_X2 and pack_z != 1) else None    # per-channel max |z|: the next convolution's operand scales
    z = _BnAct.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, float(slope),
                     res_pre, res_post, stats_part if training else None, zm, pack_z, pack_dy)
    if pack_z == 1:
        _tag_px2(z, _tls.last_zexps)
    elif pack_z == 2:
        z._dca_twin = (_tls.last_twin, _ver(z))
        _tag_cmax(z, zm, _L().dca_bn_pack_chunks(C, y[0, 0].numel()))
    elif zm is not None:
        _tag_cmax(z, zm, _L().dca_bn_num_chunks(C, y[0, 0].numel()))
    if bn.training and bn.num_batches_tracked is not None:
        pending = getattr(_tls, "pending", None)
        if pending is not None:
            pending.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    return z


class reduced_precision:
    """Inference-only context (BASELINE configs 2 "bf16" / 5 "fp16"): inside it the fused conv + BatchNorm inference
    launches of the 3x3x3 stride-1 convolutions use `dtype` (torch.bfloat16 / torch.float16) operands with ONE native
    MFMA product per multiply and fp32 accumulation (csrc/conv3d_lp.hip) instead of the fp32-grade six-product split.
    Softmax, soft-argmin, BatchNorm folding, the context injection's arg-max / region softmax and the 32 -> 1 logit
    heads stay fp32.  Per thread; never active outside the context; training (autograd) raises."""

    def __init__(self, dtype):
        if dtype not in LP_DTYPES:
            raise ValueError("reduced_precision: torch.bfloat16 or torch.float16")
        self.dtype = dtype

    def __enter__(self):
        self.prev = getattr(_tls, "lp", None)
        _tls.lp = self.dtype
        return self

    def __exit__(self, *exc):
        _tls.lp = self.prev
        return False


def _lp_dtype():
    return getattr(_tls, "lp", None)


def _pack_dy_ok(x, x2, conv, transposed, stride, res_pre):
    """may the gradient of this convolution's output travel as a packed px2 operand?  (its only readers are then the f16x2
    backward-data and weight-gradient kernels of this very convolution)"""
    if not (PACK and CONV_X2 and x2 is None and not transposed and stride == 1 and conv.kernel_size[0] == 3 and res_pre is None):
        return False
    Cout, Cin = conv.weight.shape[0], conv.weight.shape[1]
    if Cout % 8 or Cout < 8 or not _x3_eligible(x, None, 3, 1, False, Cin, Cout):
        return False
    return _is_packed(x) or (x.shape[-1] % 4 == 0 and x.data_ptr() % 16 == 0)


def convbn3d(x, conv, bn, slope=1.0, res_pre=None, res_post=None, x2=None, alias=False, pack_out=False):
    """`convbn_3d` (models/submodule.py:121-124) + activation + residual adds, on the HIP kernels.

    conv: nn.Conv3d / nn.ConvTranspose3d (bias=False), bn: nn.BatchNorm3d -- used as parameter holders.
    Inference (eval BN, no grad): one fused launch (BN folded into the conv epilogue).  Otherwise conv ->
    batch statistics -> apply, each with a HIP backward.
    pack_out: the caller promises that the result has ONE consumer and that it is a 3x3x3 stride-1 convolution through this
    function: in training the BatchNorm apply pass then writes the packed px2 operand format (csrc/dca_common.h) instead
    of fp32 (never with residuals; silently fp32 wherever the packed form does not apply)."""
    transposed = isinstance(conv, torch.nn.ConvTranspose3d)
    stride = conv.stride[0]
    if alias:
        # alias=True: returns (z, x') with x' = x for the other consumers of x (see _Conv3d.forward); plain (z, x) where the
        # fused form does not apply
        fuse = (PAIR_FUSE and torch.is_grad_enabled() and x.requires_grad and x2 is None and not transposed and stride == 1
                and conv.kernel_size[0] == 3 and conv.weight.shape[0] > 1 and _lp_dtype() is None and x.dtype == torch.float32)
        if not fuse:
            return convbn3d(x, conv, bn, slope, res_pre, res_post, x2), x
        stats = bool(BN_FUSE and bn.training)
        pdy = _pack_dy_ok(x, None, conv, False, 1, res_pre)
        out = _Conv3d.apply(x, None, conv.weight, 1, False, stats, True, pdy)
        y, part, xa = (out[0], out[1], out[2]) if stats else (out[0], None, out[1])
        z = bn_act(y, bn, slope, res_pre, res_post, part if (part is not None and part.numel()) else None,
                   pack_out=pack_out, pack_dy=pdy)
        return z, xa
    if not bn.training and not torch.is_grad_enabled():
        lp = _lp_dtype()
        if lp is not None and not transposed and stride == 1 and conv.kernel_size[0] == 3 and x2 is None:
            with torch.cuda.device_of(x):
                stats = bn_eval_affine(bn)
            C = bn.num_features
            return conv3d_lp(x, conv.weight, lp, stats[2 * C:3 * C], stats[3 * C:], slope, res_pre, res_post,
                             out_dtype=x.dtype)
        xx = _req(x, "convbn3d")
        with torch.cuda.device_of(xx):
            stats = bn_eval_affine(bn)
        C = bn.num_features
        return conv3d_fused_inference(xx, conv.weight, stride, transposed, stats[2 * C:3 * C], stats[3 * C:], slope,
                                      res_pre, res_post, x2)
    if BN_FUSE and bn.training and conv.weight.shape[0 if not transposed else 1] > 1 and _lp_dtype() is None:
        # the convolution kernel emits the batch statistics of its own output where it has such a form (the bf16x3 family):
        # no separate pass over y
        pdy = _pack_dy_ok(x, x2, conv, transposed, stride, res_pre)
        y, part = _Conv3d.apply(x, x2, conv.weight, int(stride), bool(transposed), True, False, pdy)
        return bn_act(y, bn, slope, res_pre, res_post, part if part.numel() else None, pack_out=pack_out, pack_dy=pdy)
    y = conv3d(x, conv.weight, stride, transposed, x2)
    return bn_act(y, bn, slope, res_pre, res_post)


# ------------------------------------------------------------------------------------------------
# pooling / interpolation
# ------------------------------------------------------------------------------------------------
class _AvgPool3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "avg_pool3d")
        N, C, D, H, W = x.shape
        y = torch.empty((N, C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), device=x.device, dtype=torch.float32)
        with torch.cuda.device_of(x):
            _chk(_L().dca_avgpool3d_fwd(_ptr(x), _ptr(y), N * C, D, H, W, _stream()), "dca_avgpool3d_fwd")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _req(gy, "avg_pool3d.backward")
        N, C, D, H, W = ctx.shape
        gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
        with torch.cuda.device_of(gy):
            _chk(_L().dca_avgpool3d_bwd(_ptr(gy), _ptr(gx), None, None, N * C, D, H, W, _stream()), "dca_avgpool3d_bwd")
        return gx


class _PoolFork(torch.autograd.Function):
    """(AvgPool3d(3, 2, 1)(x), x) as ONE autograd node: the second output is x itself, for a second consumer (a cva block
    pools its input AND feeds it to the `fuse` convolution, models/augment/cva.py:62-69).  Backward adds that consumer's
    gradient inside the pooling backward kernel (`res`) instead of leaving the sum to autograd's accumulation pass."""

    @staticmethod
    def forward(ctx, x, third):
        x = _req(x, "avg_pool3d")
        N, C, D, H, W = x.shape
        y = torch.empty((N, C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), device=x.device, dtype=torch.float32)
        with torch.cuda.device_of(x):
            _chk(_L().dca_avgpool3d_fwd(_ptr(x), _ptr(y), N * C, D, H, W, _stream()), "dca_avgpool3d_fwd")
        ctx.shape = tuple(x.shape)
        ctx.set_materialize_grads(False)
        if third:       # a third consumer of x (the block's `cost0 + augmented_cost`): its gradient joins inside the same kernel
            return y, x.view_as(x), x.view_as(x)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gx2, gx3=None):
        extra = [g for g in (gx2, gx3) if g is not None]
        if gy is None:
            return (extra[0] + extra[1] if len(extra) == 2 else (extra[0] if extra else None)), None
        gy = _req(gy, "avg_pool3d.backward")
        res = _opt(extra[0], "avg_pool3d.backward") if extra else None
        res2 = _opt(extra[1], "avg_pool3d.backward") if len(extra) == 2 else None
        N, C, D, H, W = ctx.shape
        gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
        with torch.cuda.device_of(gy):
            _chk(_L().dca_avgpool3d_bwd(_ptr(gy), _ptr(gx), _ptr(res), _ptr(res2), N * C, D, H, W, _stream()),
                 "dca_avgpool3d_bwd")
        return gx, None


def avg_pool3d_fork(x, third=False):
    """(avg_pool3d_k3s2p1(x), x'[, x'']) with x' = x'' = x for further consumers whose gradients are added inside the pooling
    backward kernel (training path; plain pooling and x itself otherwise)"""
    if PAIR_FUSE and torch.is_grad_enabled() and x.requires_grad and x.dtype == torch.float32:
        outs = _PoolFork.apply(x, bool(third))
        for o in outs[1:]:
            _copy_tags(x, o)        # the aliases are x: its per-channel maxima / packed twin stay valid (residual bounds)
        return outs
    return (avg_pool3d_k3s2p1(x), x, x) if third else (avg_pool3d_k3s2p1(x), x)


class _Trilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = _req(x, "interpolate")
        N, C, D, H, W = x.shape
        y = torch.empty((N, C, D * scale, H * scale, W * scale), device=x.device, dtype=torch.float32)
        with torch.cuda.device_of(x):
            _chk(_L().dca_trilinear_fwd(_ptr(x), _ptr(y), N * C, D, H, W, scale, _stream()), "dca_trilinear_fwd")
        ctx.meta = (tuple(x.shape), scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, scale = ctx.meta
        gy = _req(gy, "interpolate.backward")
        N, C, D, H, W = shape
        gx = torch.empty(shape, device=gy.device, dtype=torch.float32)
        with torch.cuda.device_of(gy):
            _chk(_L().dca_trilinear_bwd(_ptr(gy), _ptr(gx), N * C, D, H, W, scale, _stream()), "dca_trilinear_bwd")
        return gx, None


def avg_pool3d_k3s2p1(x):
    return _AvgPool3d.apply(x)


def trilinear_upsample(x, scale):
    return _Trilinear.apply(x, int(scale))


# ------------------------------------------------------------------------------------------------
# DCA: context injection and disparity attention
# ------------------------------------------------------------------------------------------------
class _ContextInject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, preds):
        x, preds = _req(x, "context_inject"), _req(preds, "context_inject.preds")
        B, C, n = x.shape[:3]
        HW = x.shape[3] * x.shape[4]
        key = torch.empty_like(x)
        kstar = torch.empty((B, HW), device=x.device, dtype=torch.int32)
        e = torch.empty((B, HW), device=x.device, dtype=torch.float32)
        pm = torch.empty_like(e)
        denom = torch.empty((B, n), device=x.device, dtype=torch.float32)
        part = torch.empty((B * ((HW + 255) // 256) * n,), device=x.device, dtype=torch.float32)
        with torch.cuda.device_of(x):
            _chk(_L().dca_context_inject_fwd(_ptr(x), _ptr(preds), _ptr(key), _ptr(kstar), _ptr(e), _ptr(pm),
                                             _ptr(denom), _ptr(part), B, C, n, HW, _stream()), "dca_context_inject_fwd")
        ctx.save_for_backward(x, preds, kstar, e, pm, denom)
        ctx.mark_non_differentiable(kstar)
        return key, kstar

    @staticmethod
    def backward(ctx, dkey, _unused):
        x, preds, kstar, e, pm, denom = ctx.saved_tensors
        dkey = _req(dkey, "context_inject.backward")
        B, C, n = x.shape[:3]
        HW = x.shape[3] * x.shape[4]
        dx, dpreds = torch.empty_like(x), torch.empty_like(preds)
        dw = torch.empty_like(e)
        T = torch.empty_like(denom)
        part = torch.empty((B * ((HW + 255) // 256) * n,), device=x.device, dtype=torch.float32)
        with torch.cuda.device_of(x):
            _chk(_L().dca_context_inject_bwd(_ptr(dkey), _ptr(x), _ptr(preds), _ptr(kstar), _ptr(e), _ptr(pm),
                                             _ptr(denom), _ptr(dx), _ptr(dpreds), _ptr(dw), _ptr(T), _ptr(part), B, C,
                                             n, HW, _stream()), "dca_context_inject_bwd")
        return dx, dpreds


class _DispAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = _req(q, "attention.q"), _req(k, "attention.k"), _req(v, "attention.v")
        B, C, n = q.shape[:3]
        HW = q.shape[3] * q.shape[4]
        out = torch.empty_like(q)
        with torch.cuda.device_of(q):
            _chk(_L().dca_disp_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, C, n, HW, _stream()),
                 "dca_disp_attention_fwd")
        ctx.save_for_backward(q, k, v)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v = ctx.saved_tensors
        dout = _req(dout, "attention.backward")
        B, C, n = q.shape[:3]
        HW = q.shape[3] * q.shape[4]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        with torch.cuda.device_of(q):
            _chk(_L().dca_disp_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(dout), _ptr(dq), _ptr(dk), _ptr(dv), B,
                                             C, n, HW, _stream()), "dca_disp_attention_bwd")
        return dq, dk, dv


def context_inject(x, preds):
    """SemanticLevelContext's feats_sl + inputs (semantic_level.py:96-126); returns (key_feats, kstar)."""
    return _ContextInject.apply(x, preds)


def disparity_attention(q, k, v):
    """softmax(q k^T / sqrt(8)) v along the disparity axis of every pixel, heads of 8 channels
    (SelfAttention_bn.py:70-94).  Limits of the kernels: channels % 8 == 0 and at most 64 disparity bins."""
    if q.dim() != 5 or q.shape[1] % 8 or q.shape[2] > 64:
        raise RuntimeError(f"disparity_attention: needs (B, C % 8 == 0, n <= 64, H, W) tensors, got {tuple(q.shape)} "
                           "(n = maxdisp/8 with the down-sampling cva, maxdisp/4 without)")
    return _DispAttention.apply(q, k, v)


# ------------------------------------------------------------------------------------------------
# convex x4 up-sampling and the stereo focal loss (csrc/heads2d.hip)
# ------------------------------------------------------------------------------------------------
class _ConvexUp4(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask_logits, disp):
        mask_logits, disp = _req(mask_logits, "convex_upsample4.mask"), _req(disp, "convex_upsample4.disp")
        B, C, h, w = mask_logits.shape
        if C != 144 or tuple(disp.shape) != (B, 1, h, w):
            raise RuntimeError(f"convex_upsample4: expected mask (B,144,h,w) and disp (B,1,h,w), got "
                               f"{tuple(mask_logits.shape)} and {tuple(disp.shape)}")
        up = torch.empty((B, 1, 4 * h, 4 * w), device=disp.device, dtype=torch.float32)
        with torch.cuda.device_of(disp):
            _chk(_L().dca_convex_up4_fwd(_ptr(mask_logits), _ptr(disp), _ptr(up), B, h, w, _stream()),
                 "dca_convex_up4_fwd")
        ctx.save_for_backward(mask_logits, disp)
        return up

    @staticmethod
    def backward(ctx, gup):
        mask_logits, disp = ctx.saved_tensors
        gup = _req(gup, "convex_upsample4.backward")
        B, _, h, w = mask_logits.shape
        gl, gd = torch.empty_like(mask_logits), torch.empty_like(disp)
        wk = torch.empty((B, 9, h, w), device=disp.device, dtype=torch.float32)
        with torch.cuda.device_of(disp):
            _chk(_L().dca_convex_up4_bwd(_ptr(mask_logits), _ptr(disp), _ptr(gup), _ptr(gl), _ptr(gd), _ptr(wk), B, h, w,
                                         _stream()), "dca_convex_up4_bwd")
        return gl, gd


def convex_upsample4(mask_logits, disp):
    """PropgationNet_4x.forward after its conv (models/submodule.py:366-373): (B,144,h,w), (B,1,h,w) -> (B,1,4h,4w)."""
    return _ConvexUp4.apply(mask_logits, disp)


class _FocalLevels(torch.autograd.Function):
    """sum_l w_l * StereoFocalLoss.loss_per_level(est_l, gt) for estimates of ONE resolution; `gt` is already pooled."""

    @staticmethod
    def forward(ctx, gt, coef, weights, *ests):
        ests = [_req(e, "focal_loss.est") for e in ests]
        gt = _req(gt, "focal_loss.gt")
        B, K = ests[0].shape[0], ests[0].shape[1]
        HW = ests[0][0, 0].numel()
        n = len(ests)
        if n > 8 or K > 256 or K < 2 or any(e.shape != ests[0].shape for e in ests) or gt.numel() != B * HW:
            raise RuntimeError("focal_loss: at most 8 equally shaped (B,K<=256,H,W) estimates per call and a ground truth "
                               "pooled to (B,1,H,W)")
        lib = _L()
        work = torch.empty((lib.dca_focal_loss_workspace(n, B, HW),), device=gt.device, dtype=torch.float64)
        out = torch.empty((n + 1,), device=gt.device, dtype=torch.float32)
        eptr = (ctypes.c_void_p * n)(*[e.data_ptr() for e in ests])
        wts = (ctypes.c_float * n)(*[float(w) for w in weights])
        with torch.cuda.device_of(gt):
            _chk(lib.dca_focal_loss_fwd(eptr, wts, n, _ptr(gt), _ptr(work), _ptr(out), B, K, HW, float(coef), _stream()),
                 "dca_focal_loss_fwd")
        ctx.save_for_backward(gt, work, *ests)
        ctx.meta = (float(coef), [float(w) for w in weights])
        return out[n]

    @staticmethod
    def backward(ctx, gloss):
        gt, work, *ests = ctx.saved_tensors
        coef, weights = ctx.meta
        n = len(ests)
        B, K = ests[0].shape[0], ests[0].shape[1]
        HW = ests[0][0, 0].numel()
        gloss = _req(gloss.reshape(1), "focal_loss.backward")
        gests = [torch.empty_like(e) for e in ests]
        eptr = (ctypes.c_void_p * n)(*[e.data_ptr() for e in ests])
        gptr = (ctypes.c_void_p * n)(*[g.data_ptr() for g in gests])
        wts = (ctypes.c_float * n)(*weights)
        with torch.cuda.device_of(gt):
            _chk(_L().dca_focal_loss_bwd(eptr, gptr, wts, n, _ptr(gt), _ptr(work), _ptr(gloss), B, K, HW, coef, _stream()),
                 "dca_focal_loss_bwd")
        return (None, None, None) + tuple(gests)


def focal_loss_levels(ests, gt_pooled, weights, focal_coefficient):
    """Weighted stereo focal loss (models/loss.py:206-240) of several estimates that share one resolution."""
    return _FocalLevels.apply(gt_pooled, float(focal_coefficient), tuple(float(w) for w in weights), *ests)


# ------------------------------------------------------------------------------------------------
# Reduced-precision inference (BASELINE configs 2 / 5): bf16 or fp16 storage, one MFMA product, fp32 accumulation
# ------------------------------------------------------------------------------------------------
LP_DTYPES = {torch.bfloat16: 1, torch.float16: 2}      # DCA_BF16 / DCA_FP16 of include/dca_hip.h


def _req_lp(t, name, lp):
    """contiguous ROCm tensor that is either fp32 or the 2-byte type `lp`"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: the DCANet hot path runs only as HIP kernels on a ROCm device; there is no CPU fallback")
    if t.dtype not in (torch.float32, lp):
        raise RuntimeError(f"{name}: expected float32 or {lp}, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def conv3d_lp(x, weight, lp, scale=None, shift=None, slope=1.0, res_pre=None, res_post=None, out_dtype=None,
              src_ab=0, flip=0):
    """3x3x3 stride-1 convolution with `lp` (torch.bfloat16 / torch.float16) operands and fp32 accumulation; the folded
    BatchNorm affine, activation and residual adds run in fp32 in the epilogue.  x may be fp32 (rounded on the fly) or
    `lp`; the result (and the residuals) are `out_dtype` = `lp` (default) or fp32.  Forward only."""
    code = LP_DTYPES[lp]
    out_dtype = lp if out_dtype is None else out_dtype
    x = _req_lp(x, "conv3d_lp", lp)
    weight = _req(weight, "conv3d_lp.weight")
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise RuntimeError("conv3d_lp: the reduced-precision path is inference only (wrap the call in torch.no_grad())")
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[1] if src_ab else weight.shape[0]
    assert (weight.shape[0] if src_ab else weight.shape[1]) == Cin and tuple(weight.shape[2:]) == (3, 3, 3)
    lib = _L()
    for r in (res_pre, res_post):
        if r is not None and (r.dtype != out_dtype or tuple(r.shape) != (N, Cout, D, H, W)):
            raise RuntimeError("conv3d_lp: residuals must have the output's shape and dtype")
    res_pre = None if res_pre is None else _req_lp(res_pre, "conv3d_lp.res_pre", lp)
    res_post = None if res_post is None else _req_lp(res_post, "conv3d_lp.res_post", lp)
    with torch.cuda.device_of(x):
        def build():
            wx = torch.empty((lib.dca_conv3d_lp_weight_bytes(Cin, Cout) // 2,), device=x.device, dtype=torch.int16)
            _chk(lib.dca_conv3d_lp_prep_weight(_ptr(weight), _ptr(wx), Cin, Cout, int(src_ab), int(flip), code, _stream()),
                 "dca_conv3d_lp_prep_weight")
            return wx
        wx = _memo(("lpprep", Cin, Cout, int(src_ab), int(flip), code), (weight,), build)
        y = torch.empty((N, Cout, D, H, W), device=x.device, dtype=out_dtype)
        _chk(lib.dca_conv3d_lp_forward(_ptr(x), _ptr(wx), _ptr(y), _ptr(_opt(scale, "scale")), _ptr(_opt(shift, "shift")),
                                       _ptr(res_pre), _ptr(res_post), float(slope), N, Cin, Cout, D, H, W, code,
                                       int(x.dtype == torch.float32), int(out_dtype == torch.float32), _stream()),
             "dca_conv3d_lp_forward")
    return y


def conv1x1_lp(x, weight, lp, x2=None, scale=None, shift=None, slope=1.0, res_pre=None, res_post=None, out_dtype=None):
    """1x1x1 convolution (Cout <= 32) over one or two `lp` inputs (implicit channel concat), fp32 accumulation, fused
    affine + activation + residual epilogue; result `lp` (default) or fp32.  weight: (Cout, C1 + C2[,1,1,1]) fp32."""
    code = LP_DTYPES[lp]
    out_dtype = lp if out_dtype is None else out_dtype
    x = _req_lp(x, "conv1x1_lp", lp)
    x2 = None if x2 is None else _req_lp(x2, "conv1x1_lp.x2", lp)
    if x.dtype != lp or (x2 is not None and x2.dtype != lp):
        raise RuntimeError(f"conv1x1_lp: inputs must be {lp}")
    weight = _req(weight, "conv1x1_lp.weight")
    N, C1 = x.shape[0], x.shape[1]
    C2 = 0 if x2 is None else x2.shape[1]
    Cout = weight.shape[0]
    S = x[0, 0].numel()
    assert weight[0].numel() == C1 + C2, "conv1x1_lp: channel mismatch"
    if Cout > 32 or S % 4:
        raise RuntimeError("conv1x1_lp: at most 32 output channels and a voxel count divisible by 4")
    lib = _L()
    oshape = (N, Cout) + tuple(x.shape[2:])
    for r in (res_pre, res_post):
        if r is not None and (r.dtype != out_dtype or tuple(r.shape) != oshape):
            raise RuntimeError("conv1x1_lp: residuals must have the output's shape and dtype")
    res_pre = None if res_pre is None else _req_lp(res_pre, "conv1x1_lp.res_pre", lp)
    res_post = None if res_post is None else _req_lp(res_post, "conv1x1_lp.res_post", lp)
    with torch.cuda.device_of(x):
        def build():
            wf = torch.empty((lib.dca_conv1_lp_weight_bytes(C1, C2) // 2,), device=x.device, dtype=torch.int16)
            _chk(lib.dca_conv1_lp_prep_weight(_ptr(weight), _ptr(wf), Cout, C1, C2, code, _stream()),
                 "dca_conv1_lp_prep_weight")
            return wf
        wf = _memo(("lp1prep", Cout, C1, C2, code), (weight,), build)
        y = torch.empty(oshape, device=x.device, dtype=out_dtype)
        _chk(lib.dca_conv1_lp_forward(_ptr(x), _ptr(x2), _ptr(wf), _ptr(y), _ptr(_opt(scale, "scale")),
                                      _ptr(_opt(shift, "shift")), _ptr(res_pre), _ptr(res_post), float(slope), N, C1, C2,
                                      Cout, S, code, int(out_dtype == torch.float32), _stream()), "dca_conv1_lp_forward")
    return y


def _lp_code(t):
    return LP_DTYPES[t.dtype]


def avg_pool3d_lp(x):
    """AvgPool3d(3, 2, 1) of a 2-byte (N,C,D,H,W) tensor -> fp32 (the 1/8-resolution interior of a DCA block is fp32)."""
    x = _req_lp(x, "avg_pool3d_lp", x.dtype)
    N, C, D, H, W = x.shape
    y = torch.empty((N, C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), device=x.device, dtype=torch.float32)
    with torch.cuda.device_of(x):
        _chk(_L().dca_avgpool3d_lp_fwd(_ptr(x), _ptr(y), N * C, D, H, W, _lp_code(x), _stream()), "dca_avgpool3d_lp_fwd")
    return y


def trilinear_up2_lp(x, lp):
    """x2 trilinear up-sampling of an fp32 tensor, written in the 2-byte type `lp`."""
    x = _req(x, "trilinear_up2_lp")
    N, C, D, H, W = x.shape
    y = torch.empty((N, C, 2 * D, 2 * H, 2 * W), device=x.device, dtype=lp)
    with torch.cuda.device_of(x):
        _chk(_L().dca_trilinear_up2_lp_fwd(_ptr(x), _ptr(y), N * C, D, H, W, LP_DTYPES[lp], _stream()),
             "dca_trilinear_up2_lp_fwd")
    return y


def conv3d_s2_lp(x, weight, scale, shift, slope, exact=False):
    """3x3x3 stride-2 conv + folded BN + activation reading a 2-byte x, fp32 result.  Default: weights rounded to x's
    type, one MFMA product (csrc/conv3d_s2_lp.hip); exact=True keeps the fp32 MFMA arithmetic (csrc/conv3d_mfma.hip)."""
    x = _req_lp(x, "conv3d_s2_lp", x.dtype)
    weight = _req(weight, "conv3d_s2_lp.weight")
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[0]
    Do, Ho, Wo = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    lib = _L()
    code = _lp_code(x)
    with torch.cuda.device_of(x):
        y = torch.empty((N, Cout, Do, Ho, Wo), device=x.device, dtype=torch.float32)
        if exact:
            wt, Apad = _prep_weight(weight, Cin, Cout, 27, 0, 0, 3, 2, False)
            _chk(lib.dca_conv3d_forward_mixed(_ptr(x), _ptr(wt), _ptr(y), _ptr(scale), _ptr(shift), None, None,
                                              float(slope), N, Cin, Cout, Apad, D, H, W, Do, Ho, Wo, 0, code, _stream()),
                 "dca_conv3d_forward_mixed")
            return y

        def build():
            wx = torch.empty((lib.dca_conv3d_s2_lp_weight_bytes(Cin) // 2,), device=x.device, dtype=torch.int16)
            _chk(lib.dca_conv3d_s2_lp_prep_weight(_ptr(weight), _ptr(wx), Cin, Cout, code, _stream()),
                 "dca_conv3d_s2_lp_prep_weight")
            return wx
        wx = _memo(("lps2", Cin, Cout, code), (weight,), build)
        _chk(lib.dca_conv3d_s2_lp_forward(_ptr(x), _ptr(wx), _ptr(y), _ptr(_opt(scale, "scale")), _ptr(_opt(shift, "shift")),
                                          float(slope), N, Cin, Cout, D, H, W, code, _stream()), "dca_conv3d_s2_lp_forward")
    return y


def deconv3d_lp(x, weight, lp, scale, shift, slope, res_pre=None, res_post=None, exact=False):
    """ConvTranspose3d(3, s2, p1, op1) + folded BN + residuals + activation: fp32 x -> 2-byte result / residuals.
    Default: operands rounded to `lp`, one MFMA product (csrc/deconv3d_lp.hip); exact=True keeps the fp32 MFMA
    arithmetic and only writes / reads the 2-byte tensors (csrc/conv3d_mfma.hip)."""
    x = _req(x, "deconv3d_lp")
    weight = _req(weight, "deconv3d_lp.weight")
    N, Cin, D, H, W = x.shape
    Cout = weight.shape[1]
    oshape = (N, Cout, 2 * D, 2 * H, 2 * W)
    for r in (res_pre, res_post):
        if r is not None and (r.dtype != lp or tuple(r.shape) != oshape):
            raise RuntimeError("deconv3d_lp: residuals must have the output's shape and dtype")
    res_pre = None if res_pre is None else _req_lp(res_pre, "deconv3d_lp.res_pre", lp)
    res_post = None if res_post is None else _req_lp(res_post, "deconv3d_lp.res_post", lp)
    lib = _L()
    code = LP_DTYPES[lp]
    with torch.cuda.device_of(x):
        y = torch.empty(oshape, device=x.device, dtype=lp)
        if exact or Cin > 64 or Cout > 32:
            wt, Apad = _prep_weight(weight, Cin, Cout, 27, 1, 0, 3, 2, True)
            _chk(lib.dca_conv3d_forward_mixed(_ptr(x), _ptr(wt), _ptr(y), _ptr(scale), _ptr(shift), _ptr(res_pre),
                                              _ptr(res_post), float(slope), N, Cin, Cout, Apad, D, H, W, 2 * D, 2 * H, 2 * W,
                                              1, code, _stream()), "dca_conv3d_forward_mixed")
            return y

        def build():
            wx = torch.empty((lib.dca_conv3d_lp_weight_bytes(Cin, Cout) // 2,), device=x.device, dtype=torch.int16)
            _chk(lib.dca_conv3d_lp_prep_weight(_ptr(weight), _ptr(wx), Cin, Cout, 1, 0, code, _stream()),
                 "dca_conv3d_lp_prep_weight")
            return wx
        wx = _memo(("lpdeconv", Cin, Cout, code), (weight,), build)
        _chk(lib.dca_deconv3d_lp_forward(_ptr(x), _ptr(wx), _ptr(y), _ptr(_opt(scale, "scale")), _ptr(_opt(shift, "shift")),
                                         _ptr(res_pre), _ptr(res_post), float(slope), N, Cin, Cout, D, H, W, code, _stream()),
             "dca_deconv3d_lp_forward")
    return y


def conv3d_c1_lp(x, weight):
    """nn.Conv3d(C, 1, 3, padding=1, bias=False) logit head on a 2-byte x: the 27 taps become the output axis of the
    reduced-precision 1x1x1 GEMM (fp32 accumulation, fp32 tap tensor), then the fp32 27-tap shifted gather."""
    lp = x.dtype
    weight = _req(weight, "conv3d_c1_lp.weight")
    N, C, D, H, W = x.shape
    w27 = _memo(("c1lp",), (weight,), lambda: weight[0].reshape(C, 27).t().contiguous())
    T = conv1x1_lp(x, w27, lp, out_dtype=torch.float32)
    y = torch.empty((N, 1, D, H, W), device=x.device, dtype=torch.float32)
    with torch.cuda.device_of(x):
        _chk(_L().dca_conv3d_c1_gather(_ptr(T), _ptr(y), N, D, H, W, _stream()), "dca_conv3d_c1_gather")
    return y
