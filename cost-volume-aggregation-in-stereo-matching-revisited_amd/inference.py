"""KITTI-style single-pair inference around the hipGraph-captured hot path (BASELINE config 5):
the host side of the reference's my_img.py:47-110 -- per-image per-channel mean/std normalisation (:59-68), top/right
zero padding to a fixed 384x1248 frame (:71-87), `model(left, right)` in eval mode (:95-101), crop back (:105-108)
and the uint16 x256 PNG (:110) -- with the 3D part replayed as ONE hipGraph (dcanet_amd.graph.GraphedHotPath).

The reference's script calls `.squeeze()` on the model's return value although eval `forward` returns a tuple
(gwcnet_dca_g.py:282); the disparity is element 0."""
from __future__ import annotations

import numpy as np
import torch

from .graph import GraphedHotPath


def normalize_pair(left_rgb: np.ndarray, right_rgb: np.ndarray) -> np.ndarray:
    """my_img.py:47-69 `load_data` after the file read: (H,W,3) uint8 x2 -> (6,H,W) float32, every colour plane
    shifted / scaled by its own mean and (population) standard deviation."""
    left_rgb, right_rgb = np.asarray(left_rgb), np.asarray(right_rgb)
    assert left_rgb.ndim == 3 and left_rgb.shape[2] >= 3 and left_rgb.shape[:2] == right_rgb.shape[:2]
    out = np.zeros((6,) + left_rgb.shape[:2], "float32")
    for i, img in enumerate((left_rgb, right_rgb)):
        for c in range(3):
            plane = img[:, :, c]
            out[3 * i + c] = (plane - np.mean(plane[:])) / np.std(plane[:])
    return out


def pad_or_crop(temp_data: np.ndarray, crop_height: int = 384, crop_width: int = 1248):
    """my_img.py:71-87 `my_transform`: images no larger than the frame go to its BOTTOM-LEFT corner (zero rows on top,
    zero columns on the right); larger ones are cropped (vertically centred, from column 0 -- the reference computes a
    horizontal offset and does not use it).  Returns (left (1,3,Hc,Wc), right, h, w)."""
    _, h, w = temp_data.shape
    if h <= crop_height and w <= crop_width:
        frame = np.zeros((6, crop_height, crop_width), "float32")
        frame[:, crop_height - h:crop_height, 0:w] = temp_data
    else:
        start_y = int((h - crop_height) / 2)
        frame = temp_data[:, start_y:start_y + crop_height, 0:crop_width]
    left = torch.from_numpy(np.ascontiguousarray(frame[None, 0:3])).float()
    right = torch.from_numpy(np.ascontiguousarray(frame[None, 3:6])).float()
    return left, right, h, w


def crop_back(disp: np.ndarray, h: int, w: int, crop_height: int = 384, crop_width: int = 1248) -> np.ndarray:
    """my_img.py:105-108"""
    if h <= crop_height and w <= crop_width:
        return disp[crop_height - h:crop_height, 0:w]
    return disp


def disparity_png(path: str, disp: np.ndarray) -> None:
    """my_img.py:110: `imsave(savename, (disp * 256).astype('uint16'))` (KITTI's 16-bit disparity format)."""
    from PIL import Image
    Image.fromarray((disp * 256).astype("uint16")).save(path, format="PNG")


class KittiInference:
    """`disp = KittiInference(model)(left_rgb, right_rgb)`: my_img.py:89-110 `my()` without the file I/O.

    `model` is a GwcNet (or the nn.DataParallel wrapper the reference builds, my_img.py:37) already on the GPU with its
    weights loaded.  The 2D networks run as ordinary PyTorch-ROCm launches; the cost-volume path (volume -> dres0/1 ->
    3 x cva -> classif3 -> soft-argmin) is captured once for the fixed frame size and replayed (`graph=False`: eager)."""

    def __init__(self, model, crop_height: int = 384, crop_width: int = 1248, graph: bool = True, dtype=None):
        """dtype: None (fp32) or torch.float16 / torch.bfloat16 -- the reduced-precision path of BASELINE config 5
        ("fp16, hipGraph-captured 3D hourglass"): the captured hot path runs under ops.reduced_precision(dtype)."""
        self.net = model.module if isinstance(model, torch.nn.DataParallel) else model
        self.crop_height, self.crop_width = crop_height, crop_width
        self.graph = graph
        self.dtype = dtype
        self._graphed = None
        self.net.eval()

    @torch.no_grad()
    def forward_frame(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        """(1,3,Hc,Wc) x2 on the GPU -> full-resolution disparity (1,1,Hc,Wc); = GwcNet.forward(...)[0] in eval mode"""
        net = self.net
        fl, fr = net.feature_extraction(left), net.feature_extraction(right)
        guidance = net.guidance(left)["g"]
        args = [fl["gwc_segments"], fr["gwc_segments"]]
        if net.use_concat_volume:
            args += [fl["concat_feature"], fr["concat_feature"]]
        import contextlib
        from . import ops
        ctx = ops.reduced_precision(self.dtype) if self.dtype is not None else contextlib.nullcontext()
        with ctx:      # (a replay needs no context: the captured launches are already the reduced-precision kernels)
            if self.graph:
                if self._graphed is None:
                    self._graphed = GraphedHotPath(net, *args)
                r = self._graphed(*args)
            else:
                r = net.hot_path(*args)
        return net.prop(guidance, r["pred4_q"])

    def __call__(self, left_rgb: np.ndarray, right_rgb: np.ndarray) -> np.ndarray:
        left, right, h, w = pad_or_crop(normalize_pair(left_rgb, right_rgb), self.crop_height, self.crop_width)
        dev = next(self.net.parameters()).device
        disp = self.forward_frame(left.to(dev), right.to(dev))
        return crop_back(disp.squeeze().cpu().numpy(), h, w, self.crop_height, self.crop_width)
