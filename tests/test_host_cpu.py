"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/dca_hip.h
declares (no compute calls without a GPU), the module mirror has the reference's state-dict format, and the
product path refuses to run without the HIP kernels (no CPU fallback)."""
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    import dcanet_amd
    from dcanet_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "dca_hip.h")).read()
    declared = set(re.findall(r"^(?:int|long)\s+(dca_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_capi_argument_counts_match_header():
    from dcanet_amd import _lib
    header = open(os.path.join(ROOT, "include", "dca_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(name + r"\s*\((.*?)\)\s*;", header, flags=re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]) == len(args), name


@pytest.mark.parametrize("variant", ["g", "gc"])
def test_state_dict_matches_reference(variant):
    """tests/golden/state_dict_keys.json was dumped from the reference model (oracle/make_golden.py)."""
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))[variant]
    got = {k: list(v.shape) for k, v in GwcNet(192, use_concat_volume=(variant == "gc")).state_dict().items()}
    assert list(got) == list(want)          # same keys, same order
    assert got == want                      # same shapes


def test_factories_and_registry():
    from dcanet_amd.models import __models__, GwcNet_G, GwcNet_GC
    assert __models__["gwcnet-g"] is GwcNet_G and __models__["gwcnet-gc"] is GwcNet_GC
    m = GwcNet_G(64)
    assert m.maxdisp == 64 and m.num_groups == 40 and m.concat_channels == 0 and not m.use_concat_volume
    assert GwcNet_GC(64).concat_channels == 12


def test_no_cpu_fallback():
    from dcanet_amd.models.submodule import build_gwc_volume, disparity_regression
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        build_gwc_volume(torch.zeros(1, 8, 2, 4), torch.zeros(1, 8, 2, 4), 2, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        disparity_regression(torch.zeros(1, 4, 2, 2), 4)
    m = GwcNet(32, False).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        m(torch.zeros(1, 3, 32, 64), torch.zeros(1, 3, 32, 64))


def test_reference_asserts_are_kept():
    from dcanet_amd.models.submodule import build_gwc_volume, disparity_regression
    with pytest.raises(AssertionError):
        disparity_regression(torch.zeros(4, 2, 2), 4)          # reference submodule.py:128
    with pytest.raises(AssertionError):
        build_gwc_volume(torch.zeros(1, 9, 2, 4), torch.zeros(1, 9, 2, 4), 2, 4)   # submodule.py:150


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cost-volume-aggregation-in-stereo-matching-revisited_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, os.path.join(dirpath, f)


def test_losses_match_reference(golden=None):
    """models/loss.py mirror vs the reference (tests/golden/losses.npz): `model_loss` is device-agnostic torch code and
    is checked here incl. its NaN behaviour at masked-out pixels; the focal loss is a HIP kernel (GPU test in
    tests/test_gpu_heads.py) and must refuse CPU tensors; the LR schedules are pure host logic."""
    import numpy as np
    from oracle.seeded import seeded_tensor
    from dcanet_amd.models.loss import StereoFocalLoss, focal_loss, model_loss
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "losses.npz")))
    gt = torch.from_numpy(g["gt"])
    ests = [torch.softmax(seeded_tensor(f"loss.e{i}", (2, 8, 8, 16)), 1).requires_grad_() for i in range(5)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        focal_loss(ests, gt, 32, 5.0, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        StereoFocalLoss(max_disp=32, focal_coefficient=5.0)(ests[0], gt, variance=1)
    d0 = (seeded_tensor("loss.d0", (2, 1, 32, 64)) * 3 + gt).requires_grad_()
    d1 = (seeded_tensor("loss.d1", (2, 1, 32, 64)) * 0.3 + gt).requires_grad_()
    mask = (gt < 32) & (gt > 0)
    ml = model_loss([d0, d1], gt, mask)
    assert abs(ml.item() - float(g["model"])) < 1e-5 * max(1, abs(float(g["model"])))
    gd = torch.autograd.grad(ml, [d0, d1])
    assert torch.allclose(gd[0], torch.from_numpy(g["gd0"]), atol=1e-7, rtol=1e-4)
    assert torch.allclose(gd[1], torch.from_numpy(g["gd1"]), atol=1e-7, rtol=1e-4)
    # a non-finite estimate at a masked-out pixel must not poison the loss (reference: est[mask] never sees it)
    bad = d0.detach().clone()
    bad[~mask] = float("inf")
    bad.requires_grad_()
    ml2 = model_loss([bad, d1.detach()], gt, mask)
    assert torch.isfinite(ml2) and abs(ml2.item() - ml.item()) < 1e-5 * max(1, abs(ml.item()))
    assert torch.isfinite(torch.autograd.grad(ml2, bad)[0]).all()


def test_lr_schedules():
    """adjust_learning_rate (utils/experiment.py:91-109, main_dca.py:254) and learning_rate_adjust (util.py:132-145)."""
    from dcanet_amd.utils import adjust_learning_rate, learning_rate_adjust
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    want = {0: 1e-3, 11: 1e-3, 12: 5e-4, 19: 5e-4, 20: 2.5e-4, 24: 1.25e-4, 27: 1.25e-4, 28: 6.25e-5, 39: 6.25e-5}
    for epoch, lr in want.items():
        assert adjust_learning_rate(opt, epoch, 1e-3, "12,20,24,28:2") == pytest.approx(lr)
        assert opt.param_groups[0]["lr"] == pytest.approx(lr)
    for epoch, lr in {0: 1e-3, 299: 1e-3, 300: 1e-4, 599: 1e-4, 600: 1e-5}.items():
        assert learning_rate_adjust(opt, epoch) == pytest.approx(lr)


def test_frozen_weights_and_batched_counters_host_logic():
    """ops.frozen_weights(): memoised per source-tensor identity inside the context only; ops.batched_bn_counters():
    deferred `num_batches_tracked += 1` applied once at exit (a module used twice counts twice)"""
    import dcanet_amd  # noqa: F401
    from dcanet_amd import ops
    t, u, calls = torch.zeros(3), torch.zeros(3), []

    def build():
        calls.append(1)
        return torch.ones(1)
    with ops.frozen_weights():
        a = ops._memo(("k", 1), (t,), build)
        assert ops._memo(("k", 1), (t,), build) is a and len(calls) == 1          # reused
        assert ops._memo(("k", 2), (t,), build) is not a and len(calls) == 2      # another layout of the same weight
        assert ops._memo(("k", 1), (u,), build) is not a and len(calls) == 3      # another tensor
        with ops.frozen_weights():                                                 # nested context: its own cache
            ops._memo(("k", 1), (t,), build)
            assert len(calls) == 4
        assert ops._memo(("k", 1), (t,), build) is a and len(calls) == 4
    ops._memo(("k", 1), (t,), build)
    assert len(calls) == 5 and getattr(ops._tls, "frozen", None) is None           # outside: always rebuilt

    n1, n2 = torch.zeros((), dtype=torch.long), torch.zeros((), dtype=torch.long)
    with ops.batched_bn_counters():
        ops._tls.pending += [n1, n1, n2]
        assert n1.item() == 0
    assert (n1.item(), n2.item()) == (2, 1) and getattr(ops._tls, "pending", None) is None


def test_weight_init_is_bit_identical_to_reference():
    """a10: under the same torch seed a freshly constructed model has the reference's initial weights, bit for bit
    (gwcnet_dca_g.py:173-185 incl. the ConvTranspose3d / Guidance exceptions and the RNG consumption order);
    fixture = per-key fingerprints of the reference's own freshly constructed model (oracle/make_golden.py::gen_init)."""
    import numpy as np
    import dcanet_amd  # noqa: F401
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "init_fingerprint.npz")))
    for variant, concat in (("g", False), ("gc", True)):
        torch.manual_seed(1234)
        sd = GwcNet(64, use_concat_volume=concat).state_dict()
        keys = sorted(sd.keys())
        assert len(keys) == g[f"{variant}_fp"].shape[0]
        for i, k in enumerate(keys):
            v = sd[k].double().flatten()
            head = torch.zeros(4, dtype=torch.float64)
            head[:min(4, v.numel())] = v[:4]
            fp = torch.cat([v.sum().view(1), v.abs().sum().view(1), head]).numpy()
            assert np.array_equal(fp, g[f"{variant}_fp"][i]), f"{variant}: {k} differs from the reference's init"


def test_2d_neighbours_match_reference_on_cpu():
    """the PyTorch 2D modules either side of the path (feature_extraction, Guidance) against the
    reference's outputs (tests/golden/whole_g_eval.npz) -- they are plain torch modules, so this runs without a GPU."""
    import json
    import numpy as np
    import dcanet_amd  # noqa: F401
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    from oracle import dcanet_oracle as O
    from oracle.seeded import seeded_tensor
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        shapes = json.load(f)["g"]
    m = GwcNet(32, use_concat_volume=False)
    m.load_state_dict(O.seeded_state_dict({k: tuple(v) for k, v in shapes.items()}), strict=True)
    m.eval()
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "whole_g_eval.npz")))
    left = seeded_tensor("whole.left", (1, 3, 64, 128))
    with torch.no_grad():
        feat = m.feature_extraction(left)["gwc_feature"]
        guid = m.guidance(left)["g"]
    assert (feat[:, ::16] - torch.from_numpy(g["gwc_feature"])).abs().max() < 2e-5
    assert (guid[:, ::8] - torch.from_numpy(g["guidance"])).abs().max() < 2e-5


def test_kitti_wrapper_host_logic(tmp_path):
    """dcanet_amd.inference: my_img.py:47-110's normalisation / padding / cropping / 16-bit PNG, against an independent
    plain-numpy restatement of those lines."""
    import numpy as np
    from PIL import Image
    from dcanet_amd.inference import crop_back, disparity_png, normalize_pair, pad_or_crop
    rng = np.random.default_rng(3)
    left = rng.integers(0, 256, (37, 121, 3), dtype=np.uint8)
    right = rng.integers(0, 256, (37, 121, 3), dtype=np.uint8)
    t = normalize_pair(left, right)
    assert t.shape == (6, 37, 121) and t.dtype == np.float32
    for i, img in enumerate((left, right)):
        for c in range(3):
            want = ((img[:, :, c] - img[:, :, c].mean()) / img[:, :, c].std()).astype("float32")
            assert np.array_equal(t[3 * i + c], want)
            assert abs(float(t[3 * i + c].mean())) < 1e-5 and abs(float(t[3 * i + c].std()) - 1) < 1e-4
    L, R, h, w = pad_or_crop(t, 48, 128)          # smaller than the frame: bottom-left corner, zeros top / right
    assert (h, w) == (37, 121) and L.shape == R.shape == (1, 3, 48, 128)
    assert np.array_equal(L[0, :, 11:, :121].numpy(), t[0:3]) and np.array_equal(R[0, :, 11:, :121].numpy(), t[3:6])
    assert float(L[0, :, :11].abs().max()) == 0 and float(L[0, :, :, 121:].abs().max()) == 0
    disp = rng.random((48, 128)).astype("float32") * 100
    assert np.array_equal(crop_back(disp, h, w, 48, 128), disp[11:, :121])
    L2, R2, h2, w2 = pad_or_crop(t, 32, 64)       # larger than the frame: vertically centred crop from column 0
    assert (h2, w2) == (37, 121) and np.array_equal(L2[0].numpy(), t[0:3, 2:34, 0:64])
    assert crop_back(disp, h2, w2, 32, 64) is disp
    png = str(tmp_path / "000000_10.png")
    disparity_png(png, disp)
    back = np.asarray(Image.open(png))
    assert back.dtype == np.uint16 and np.array_equal(back, (disp * 256).astype("uint16"))
