#!/usr/bin/env python3
"""DCANet cost-volume hot path benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode fwdbwd|fwd] [--batch B]

Per-GPU batch (weak scaling): 4 stereo pairs for the training step -- BASELINE.json configs[2] "fwd+bwd, batch=4" and
configs[3] "8 GPUs, batch=32" -- and 1 for the eval forward (configs[1]); `value` counts cost volumes, not steps.

One *step* = one pass of the hot path over one batch of synthetic SceneFlow-test-shaped input
(544x960, D=192 -> 1/4-res features (B,320,136,240) x2, seed 1234+rank, already resident in HBM):

  fwdbwd (default, BASELINE.json's metric "cost-volumes/sec (fwd+bwd)"):
      the reference's training step from the features on -- train-mode forward with all auxiliary heads
      (gwcnet_dca_g.py:216-278), the convex up-sampler, focal_loss + model_loss (main_dca.py:132-133),
      backward to the features and all parameters, ONE all-reduce of the flat fp32 gradient bucket (RCCL),
      Adam step (main_dca.py:64).
  fwd: eval-mode forward only (BASELINE config[1]), BN folded into the conv epilogues.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
measured live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle on the host cores,
rank 0 / N=1 only, on a bounded sample).
"""
import argparse
import copy
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_IMG, W_IMG, MAXDISP = 544, 960, 192
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def build_model(device):
    import dcanet_amd  # noqa: F401
    from dcanet_amd.models.gwcnet_dca_g import GwcNet
    torch.manual_seed(1)
    m = GwcNet(MAXDISP, use_concat_volume=False)       # "GwcNet-G" variant named by BASELINE configs
    # synthetic BN statistics / affine so eval mode is well conditioned
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.75)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                mod.running_mean.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.bias.shape, generator=g) * 0.5 + 0.75)
    return m.to(device)


def hot_params(m):
    mods = [m.dres0, m.dres1, m.cva1, m.cva2, m.cva3, m.classif0, m.classif1, m.classif2, m.classif3, m.prop]
    return [p for mod in mods for p in mod.parameters()]


def make_inputs(batch, rank, device, h4=H_IMG // 4, w4=W_IMG // 4):
    g = torch.Generator().manual_seed(1234 + rank)
    fL = torch.randn(batch, 320, h4, w4, generator=g).to(device)
    fR = torch.randn(batch, 320, h4, w4, generator=g).to(device)
    guid = torch.randn(batch, 64, h4, w4, generator=g).to(device)
    gt = (torch.rand(batch, 1, 4 * h4, 4 * w4, generator=g) * 190.0 + 1.0).to(device)
    return fL, fR, guid, gt


def _conv_x3():
    from dcanet_amd import ops
    return ops.CONV_X3


def train_local(m, fL, fR, guid, gt, bucket):
    """this rank's part of the step: forward (all heads), losses, backward, gradients gathered into the flat bucket"""
    from dcanet_amd.models.loss import focal_loss, model_loss
    bucket.zero()
    fL.grad = fR.grad = None
    r = m.hot_path(fL, fR)
    pred4 = m.prop(guid, r["pred4_q"])
    mask = (gt < MAXDISP) & (gt > 0)
    loss = focal_loss([r["pred0"], r["pred_dca1"], r["pred_dca2"], r["pred1"], r["pred2"]], gt, MAXDISP, 5.0, False) \
        + model_loss([r["pred_dca3"], pred4], gt, mask)
    loss.backward()
    bucket.gather()
    return loss.detach()


def train_step(m, fL, fR, guid, gt, bucket, opt):
    loss = train_local(m, fL, fR, guid, gt, bucket)
    bucket.reduce_flat()
    opt.step()
    return loss


def eval_step(m, fL, fR, guid):
    with torch.no_grad():
        r = m.hot_path(fL, fR)
        return m.prop(guid, r["pred4_q"])



PMC_FILE = "r03_pmc_traffic.json"       # written by tools/pmc_collect.sh + tools/pmc_summarise.py (separate --pmc passes)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# algorithmic work of the eval forward at 544x960 / D=192 (G variant), fp32
PATH_FWD_GFLOP, PATH_FWD_GB = 822.0, 10.71
# ... and its fwd+bwd convention: 3 x (forward + training heads: 3 x 89.3 GFLOP, 0.87 Gelem)
PATH_FB_GFLOP, PATH_FB_GB = 3 * (822.0 + 3 * 89.3), 3 * (2676.7 + 870.0) * 4e-3


def kernel_roofline(device, dtype="f32"):
    """Dominant kernels timed live with HIP events on the launch stream (torch's current stream is the stream the
    C ABI launches on), all on the 3x3x3 32->32 convolution at 1/4 resolution (dres0/1, classif*, the cva blocks and
    every stride-1 backward-data pass) and its weight gradient.  `achieved` is always ALGORITHMIC fp32 FLOP/s
    (2*27*Cin*Cout*voxels per launch).
      * conv3_bf16x3_kernel (the one the model runs at this size): every fp32 product costs six bf16 MFMA products,
        so the peak its algorithmic rate is held against is the dense bf16 peak / 6; `executed_bf16` is the same
        measurement in executed bf16 matrix FLOP/s, `vs_fp32_mfma_peak` the algorithmic rate over the fp32 MFMA peak.
      * conv3_mfma_kernel (fp32 MFMA; small volumes, stride 2, DCA_CONV=fp32) and wgrad3_kernel: fp32 MFMA peak."""
    from dcanet_amd import ops
    d, h, w = MAXDISP // 4, H_IMG // 4, W_IMG // 4
    x = torch.randn(1, 32, d, h, w, device=device)
    wgt = torch.randn(32, 32, 3, 3, 3, device=device) * 0.05
    flops = 2.0 * 27 * 32 * 32 * d * h * w
    out = {}
    with torch.no_grad():
        lib = ops._L()
        wt, cpad = ops._prep_weight(wgt, 32, 32, 27, 0, 0, 3, 1, False)
        wx = torch.empty((lib.dca_conv3d_x3_weight_bytes(32, 32) // 2,), device=device, dtype=torch.int16)
        ops._chk(lib.dca_conv3d_x3_prep_weight(ops._ptr(wgt), ops._ptr(wx), 32, 32, 0, 0, ops._stream()), "x3 prep")
        y = torch.empty_like(x)

        def run_x3():
            ops._chk(lib.dca_conv3d_x3_forward(ops._ptr(x), ops._ptr(wx), ops._ptr(y), None, None, None, None, 1.0, 1,
                                               32, 32, d, h, w, ops._stream()), "x3 forward")

        def run_x2_factory(packed):
            # the launch alone: in the network the per-channel exponents come with the operand (its producer, a BatchNorm
            # kernel, emits them) and the weights are packed by a ~3 us kernel per launch (dca_conv3d_x2_prep_weight)
            w2 = torch.empty((lib.dca_conv3d_x2_weight_bytes(32, 32) // 2,), device=device, dtype=torch.int16)
            xin = ops.pack_x2(x) if packed else x
            xex = ops._exps_of(xin)
            ops._chk(lib.dca_conv3d_x2_prep_weight(ops._ptr(wgt), ops._ptr(w2), 32, 32, 0, 0, None, 0, ops._ptr(xex),
                                                   ops._stream()), "x2 prep")

            def run_x2():
                ops._chk(lib.dca_conv3d_x2_forward(ops._ptr(xin), int(packed), ops._ptr(xex), ops._ptr(w2), ops._ptr(y), None,
                                                   None, None, None, 1.0, None, 1, 32, 32, d, h, w, ops._stream()), "x2 forward")
            return run_x2

        cases = []
        if dtype != "f32":
            # reduced-precision run: its dominant kernel is conv3_lp_kernel on 2-byte tensors.  86.6 GFLOP over 200.6 MB
            # = 432 FLOP/B > the 312 FLOP/B ridge (2.5 PFLOP/s / 8 TB/s): matrix-pipe bound on paper, so `frac` is against
            # the dense bf16 / fp16 MFMA peak; the HBM view of the same launch is given beside it.
            lp = torch.bfloat16 if dtype == "bf16" else torch.float16
            xl = x.to(lp)
            sc_, sh_ = torch.ones(32, device=device), torch.zeros(32, device=device)
            cases.append((f"conv3_lp_kernel<{dtype}> (3x3x3 32->32 @1/4 res, {dtype} storage in and out, one MFMA product per "
                          "multiply, fp32 accumulation, fused affine + ReLU epilogue)", "conv3_lp", PEAK_BF16_MFMA_TFLOPS,
                          lambda: ops.conv3d_lp(xl, wgt, lp, sc_, sh_, 0.0)))
        if ops.CONV_X3 and ops.CONV_X2:
            if ops.PACK:
                cases.append(("conv3_f16x2_kernel<packed operand> (3x3x3 32->32 @1/4 res; fp32-grade via per-channel power-of-two "
                              "scaling + 2-way f16 split, 3 f16 MFMA products per fp32 product; operand in the packed px2 format "
                              "its producing BatchNorm kernel wrote: every backward-data launch and the chained forward launches "
                              "of the training step)", "conv3_f16x2_px2", PEAK_BF16_MFMA_TFLOPS / 3.0, run_x2_factory(True)))
            cases.append(("conv3_f16x2_kernel (3x3x3 32->32 @1/4 res; same arithmetic, fp32 operand scaled and split while it is "
                          "staged)", "conv3_f16x2", PEAK_BF16_MFMA_TFLOPS / 3.0, run_x2_factory(False)))
        if ops.CONV_X3:
            cases.append(("conv3_bf16x3_kernel (3x3x3 32->32 @1/4 res; fp32 via exact 3-way bf16 split, 6 bf16 MFMA "
                          "products per fp32 product)", "conv3_bf16x3", PEAK_BF16_MFMA_TFLOPS / 6.0, run_x3))
        cases.append(("conv3_mfma_kernel<S1,Cout32,CK8,tile 4x8x16> (3x3x3 32->32 @1/4 res, fp32 MFMA)", "conv3_mfma",
                      PEAK_FP32_MFMA_TFLOPS, lambda: ops.conv3d_prepared(x, wt, 32, cpad, 32, 3, 1, False)))
        def run_wgrad(x3, x2=False, packed=False):
            xa = ops.pack_x2(x) if packed else x

            def f():
                keep = ops.CONV_X3, ops.CONV_X2
                ops.CONV_X3, ops.CONV_X2 = x3, x2
                try:
                    ops._wgrad(xa, xa, wgt.new_empty(wgt.shape), 0, 32, 32, 3, 1, 32 * 27, 27)
                finally:
                    ops.CONV_X3, ops.CONV_X2 = keep
            return f

        if ops.CONV_X3 and ops.CONV_X2:
            if ops.PACK:
                cases.append(("wgrad3_f16x2_kernel<packed x, packed dy> (+reduce) (dW of 3x3x3 32->32 @1/4 res; both operands in "
                              "the packed px2 format)", "wgrad3_f16x2_px2", PEAK_BF16_MFMA_TFLOPS / 3.0, run_wgrad(True, True, True)))
            cases.append(("wgrad3_f16x2_kernel (+reduce) (dW of 3x3x3 32->32 @1/4 res; same 3-product f16 split, fp32 operands)",
                          "wgrad3_f16x2", PEAK_BF16_MFMA_TFLOPS / 3.0, run_wgrad(True, True)))
        if ops.CONV_X3:
            cases.append(("wgrad3_bf16x3_kernel (+reduce) (dW of 3x3x3 32->32 @1/4 res; same 6-product bf16 split)",
                          "wgrad3_bf16x3", PEAK_BF16_MFMA_TFLOPS / 6.0, run_wgrad(True, False)))
        cases.append(("wgrad3_kernel<S1> (+reduce) (dW of 3x3x3 32->32 @1/4 res, fp32 MFMA)", "wgrad3",
                      PEAK_FP32_MFMA_TFLOPS, run_wgrad(False, False)))
        # weight gradient of the stride-2 convolution / transposed convolution of the cva blocks (32 <-> 64 channels)
        if ops.CONV_X3 and ops.CONV_X2 and ops.WGRAD_S2_X2:
            xc2 = torch.randn(1, 64, d // 2, h // 2, w // 2, device=device)
            gw2 = torch.empty(64, 32, 3, 3, 3, device=device)
            cases.append(("wgrad3s2_f16x2_kernel (+reduce) (dW of the 3x3x3 stride-2 conv 32->64, 1/4 -> 1/8 res; 3-product f16 "
                          "split)", "wgrad3s2_f16x2", PEAK_BF16_MFMA_TFLOPS / 3.0,
                          lambda: ops._wgrad(x, xc2, gw2, 0, 32, 64, 3, 2, 32 * 27, 27),
                          2.0 * 27 * 32 * 64 * (d // 2) * (h // 2) * (w // 2)))
        # the stride-2 convolution itself (cost_agg.conv1 forward, cost_agg.conv3 backward-data): 32 -> 64, 1/4 -> 1/8 res
        if ops.CONV_X3 and ops.CONV_X2 and ops.CONV_S2_X2:
            ws2 = torch.randn(64, 32, 3, 3, 3, device=device) * 0.05
            ops._exps_of(x)       # in the network the exponents come with the tensor (its producer emits the maxima)
            cases.append(("conv3s2_f16x2_kernel (3x3x3 stride-2 conv 32->64, 1/4 -> 1/8 res; 3-product f16 split, weights packed "
                          "per launch)", "conv3s2_f16x2", PEAK_BF16_MFMA_TFLOPS / 3.0,
                          lambda: ops._conv_sliced(x, None, ws2, 32, 64, 27, 0, 0, 3, 2, False),
                          2.0 * 27 * 32 * 64 * (d // 2) * (h // 2) * (w // 2)))
        # the transposed convolution of the cva blocks (64 -> 32, 1/8 -> 1/4 res), same 6-product split: its own FLOP count
        if ops.CONV_X3 and ops.DECONV_X3:
            xc = torch.randn(1, 64, d // 2, h // 2, w // 2, device=device)
            wdc = torch.randn(64, 32, 3, 3, 3, device=device) * 0.05
            cases.append(("deconv3_bf16x3_kernel (ConvTranspose3d 3x3x3 s2 64->32, 1/8 -> 1/4 res; 6 bf16 MFMA products per "
                          "fp32 product, half-group schedule)", "deconv3_bf16x3", PEAK_BF16_MFMA_TFLOPS / 6.0,
                          lambda: ops._conv_sliced(xc, None, wdc, 64, 32, 27, 1, 0, 3, 2, True),
                          2.0 * 27 * 64 * 32 * (d // 2) * (h // 2) * (w // 2)))
        # the 1x1x1 convolutions are bandwidth bound: 4*(Cin + Cout)*V4 algorithmic bytes per launch
        if ops.CONV_X3:
            w1 = torch.randn(32, 32, 1, 1, 1, device=device) * 0.1
            cases.append(("conv1_x3_kernel<2,0> (1x1x1 32->32 @1/4 res; fp32-grade on the bf16 pipe, LDS-free)", "conv1_x3",
                          None, lambda: ops._conv_sliced(x, None, w1, 32, 32, 1, 0, 0, 1, 1, False),
                          4.0 * 64 * d * h * w))
        # the cost-volume builder itself is bandwidth bound: 4*(2*320*hw + 40*V4) algorithmic bytes per launch
        fl, fr = torch.randn(1, 320, h, w, device=device), torch.randn(1, 320, h, w, device=device)
        gwc_bytes = 4.0 * (2 * 320 * h * w + 40 * d * h * w)
        cases.append(("gwc_fused_kernel<8> (build_gwc_volume, 40 groups x %d disparities @1/4 res, fp32 volume)" % d,
                      "gwc_fused", None, lambda: ops.cost_volume(fl, fr, d, 40), gwc_bytes))
        for case in cases:
            name, key, peak, fn = case[:4]
            work = case[4] if len(case) > 4 else None   # FLOP (mfma entries) or bytes (hbm entries) when not the default
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            reps = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            if peak is None:   # HBM-bound entry
                gbs = work / (ms * 1e-3) / 1e9
                out[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "ms_per_launch": round(ms, 4),
                             "algorithmic_bytes": work, "pmc_key": key}
                continue
            fl_launch = flops if work is None else work
            tf = fl_launch / (ms * 1e-3) / 1e12
            out[name] = {"bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(tf / peak, 4), "traffic": None, "ms_per_launch": round(ms, 4),
                         "flop_per_launch": fl_launch, "pmc_key": key}
            if key == "conv3_lp":
                lp_bytes = 2.0 * 2 * 32 * d * h * w
                out[name]["algorithmic_bytes"] = lp_bytes
                out[name]["hbm_view"] = {"achieved": round(lp_bytes / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                         "unit": "GB/s", "frac": round(lp_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            if key in ("conv3_f16x2", "wgrad3_f16x2", "wgrad3s2_f16x2", "conv3_f16x2_px2", "wgrad3_f16x2_px2", "conv3s2_f16x2"):
                out[name]["peak_note"] = "dense f16 MFMA peak (2500) / 3 products per fp32 product"
                out[name]["executed_f16"] = {"achieved": round(3 * tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                             "unit": "TFLOP/s", "frac": round(3 * tf / PEAK_BF16_MFMA_TFLOPS, 4)}
                out[name]["vs_fp32_mfma_peak"] = round(tf / PEAK_FP32_MFMA_TFLOPS, 4)
            if key in ("conv3_bf16x3", "wgrad3_bf16x3", "deconv3_bf16x3"):
                out[name]["peak_note"] = "dense bf16 MFMA peak (2500) / 6 products per fp32 product"
                out[name]["executed_bf16"] = {"achieved": round(6 * tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                              "unit": "TFLOP/s", "frac": round(6 * tf / PEAK_BF16_MFMA_TFLOPS, 4)}
                out[name]["vs_fp32_mfma_peak"] = round(tf / PEAK_FP32_MFMA_TFLOPS, 4)
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(m, mode, rows_q=16, iters=3):
    """The CPU oracle (torch CPU restatement of the reference, oracle/dcanet_oracle.py) timed on the host cores
    on a bounded sample: the same workload cropped to `rows_q` of the 136 quarter-res rows (full width, full
    disparity range), 1 warm-up + `iters` timed iterations, scaled by the row fraction."""
    from oracle import dcanet_oracle as O
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()
           if k.split(".")[0] in ("dres0", "dres1", "cva1", "cva2", "cva3", "classif0", "classif1", "classif2",
                                  "classif3", "prop")}
    fL, fR, guid, gt = make_inputs(1, 0, "cpu", rows_q, W_IMG // 4)
    threads = torch.get_num_threads()

    def once():
        sd = {k: v.clone() for k, v in sd0.items()}
        if mode == "fwd":
            with torch.no_grad():
                r = O.hot_path(sd, fL, fR, MAXDISP, False)
                O.prop(sd, guid, r["pred4_q"], False)
            return
        for k in sd:
            if sd[k].is_floating_point() and "running" not in k:
                sd[k].requires_grad_()
        a, b = fL.clone().requires_grad_(), fR.clone().requires_grad_()
        r = O.hot_path(sd, a, b, MAXDISP, True)
        pred4 = O.prop(sd, guid, r["pred4_q"], True)
        mask = (gt < MAXDISP) & (gt > 0)
        loss = O.focal_loss([r["pred0"], r["pred_dca1"], r["pred_dca2"], r["pred1"], r["pred2"]], gt, MAXDISP, 5.0,
                            False) + O.model_loss([r["pred_dca3"], pred4], gt, mask)
        loss.backward()

    once()                                  # warm-up (allocator, oneDNN primitive caches)
    times = []
    for _ in range(iters):
        t0 = time.time()
        once()
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    frac = rows_q / (H_IMG // 4)
    return {"value": round(frac / dt, 5), "unit": "cost-volumes/s", "cores": threads, "kind": "port",
            "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(),
            "sample": f"{mode} on a {4 * rows_q}x{W_IMG} crop ({rows_q}/{H_IMG // 4} of the rows, full width, D={MAXDISP}; "
                      f"the full frame does not fit the time limit), 1 warm-up + {iters} timed iterations, mean "
                      f"{dt:.2f} s (min {min(times):.2f}, max {max(times):.2f}), scaled by the row fraction"}


def launch_ranks(n):
    """`python bench.py --gpus N` without torchrun's environment: start N fresh rank processes (one per GPU, RCCL over
    xGMI) with `python -m torch.distributed.run`, BEFORE this process touches the GPU (a process that has initialised
    HIP must never exec or fork workers), relay their output (rank 0 prints the JSON line) and return their exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # does not initialise the GPU on this image
    if have < n and os.environ.get("DCA_DIST_BACKEND") != "gloo":
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (DCA_DIST_BACKEND=gloo lets several ranks share "
              "one GPU for a rehearsal)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None,
                    help="stereo pairs per GPU (weak scaling); default 4 for the training step -- BASELINE.json configs[2] "
                         "(fwd+bwd, batch=4) and configs[3] (8 GPUs, batch=32) -- and 1 for the eval forward (configs[1])")
    ap.add_argument("--mode", choices=["fwdbwd", "fwd"], default="fwdbwd")
    ap.add_argument("--dtype", choices=["f32", "bf16", "fp16"], default="f32",
                    help="f32 (default, the headline: fp32 storage, fp32-grade arithmetic) or the reduced-precision INFERENCE "
                         "path of BASELINE configs 2 / 5 (2-byte storage of the 1/4-res tensors, one MFMA product per "
                         "multiply, fp32 accumulation; --mode fwd only, EPE-gated in tests/test_gpu_lowprec.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=16)
    ap.add_argument("--no-prepack", action="store_true", help="fwdbwd mode: re-lay-out each weight in its own launch")
    ap.add_argument("--graph", action="store_true",
                    help="replay the eval hot path (fwd) / the whole training step (fwdbwd, dcanet_amd.graph."
                         "GraphedTrainStep) as captured hipGraphs; off by default: at this shape the step is GPU bound "
                         "and the replay measures the same as eager launches")
    ap.add_argument("--shape", default=None, help="HxWxD of a secondary workload (e.g. 384x1248x192 KITTI, "
                                                  "256x512x64 plumbing); default = BASELINE's 544x960x192")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    if args.dtype != "f32" and args.mode != "fwd":
        raise SystemExit("bench.py: --dtype bf16/fp16 is the reduced-precision INFERENCE path: use --mode fwd")
    if args.batch is None:
        args.batch = 4 if args.mode == "fwdbwd" else 1
    global H_IMG, W_IMG, MAXDISP
    if args.shape:
        H_IMG, W_IMG, MAXDISP = (int(v) for v in args.shape.lower().split("x"))
        assert H_IMG % 8 == 0 and W_IMG % 8 == 0 and MAXDISP % 8 == 0
        args.no_cpu_baseline = True
    from dcanet_amd.parallel import FlatGradBucket, init_from_env
    rank, local, world = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)

    m = build_model(device)
    fL, fR, guid, gt = make_inputs(args.batch, rank, device)
    train_graphed, graph_error, prepack_n = False, None, 0
    if args.mode == "fwdbwd":
        m.train()
        fL.requires_grad_(); fR.requires_grad_()
        params = hot_params(m)
        bucket = FlatGradBucket(params)
        opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), capturable=bool(args.graph))
        step = lambda: train_step(m, fL, fR, guid, gt, bucket, opt)
        if not args.graph and not args.no_prepack:
            # all weight re-layouts of a step (forward + backward-data images of every conv) as ONE launch right after
            # the optimizer update instead of ~130 five-microsecond launches (ops.PrepackPlan)
            from dcanet_amd import ops as _ops
            plan = _ops.PrepackPlan()
            with plan.recording():
                train_step(m, fL, fR, guid, gt, bucket, opt)
            plan.finalize()
            plan.refresh()

            def step():
                with plan.active():
                    loss = train_step(m, fL, fR, guid, gt, bucket, opt)
                plan.refresh()
                return loss
            prepack_n = plan.n
        if args.graph:
            # the whole step as hipGraph replays (dcanet_amd.graph.GraphedTrainStep); any capture problem -> eager
            try:
                from dcanet_amd.graph import GraphedTrainStep
                step = GraphedTrainStep(lambda: train_local(m, fL, fR, guid, gt, bucket), opt.step,
                                        bucket.reduce_flat if world > 1 else None)
                train_graphed = True
            except Exception as e:   # noqa: BLE001
                graph_error = f"{type(e).__name__}: {e}"[:200]
                torch.cuda.synchronize()
                step = lambda: train_step(m, fL, fR, guid, gt, bucket, opt)
    else:
        m.eval()
        if args.graph:
            from dcanet_amd.graph import GraphedHotPath
            graphed = GraphedHotPath(m, fL, fR)

            def step():
                with torch.no_grad():
                    return m.prop(guid, graphed(fL, fR)["pred4_q"])
        else:
            step = lambda: eval_step(m, fL, fR, guid)

    import contextlib
    frozen = contextlib.ExitStack()
    if args.mode == "fwd":
        # inference: parameters are frozen for the whole run, so weight re-layouts and BatchNorm folds are computed
        # once (during warm-up) instead of per call -- ordinary inference-engine weight pre-packing
        from dcanet_amd import ops as _ops
        frozen.enter_context(_ops.frozen_weights())
        if args.dtype != "f32":
            frozen.enter_context(_ops.reduced_precision(torch.bfloat16 if args.dtype == "bf16" else torch.float16))
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    from dcanet_amd import ops as _ops_c
    amax0 = dict(_ops_c.AMAX_STATS)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0   # wall time until the last launch is enqueued: includes BLOCKING on a full launch queue
    fence()
    dt = time.perf_counter() - t0
    frozen.close()
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([dt], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        dt = max(rank_ms) * args.steps / 1e3           # the job's time is its slowest rank's

    if rank == 0:
        volumes = args.steps * args.batch * world
        line = {
            "metric": f"cost-volumes/sec (fwd+bwd) at {H_IMG}x{W_IMG} D={MAXDISP}" if args.mode == "fwdbwd"
            else f"cost-volumes/sec (fwd only) at {H_IMG}x{W_IMG} D={MAXDISP}",
            "value": round(volumes / dt, 4), "unit": "cost-volumes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
            "host_note": "host_enqueue includes blocking on the full HIP launch queue; the step's real host cost is its time "
                         "at a shape with negligible GPU work: `bench.py --shape 64x128x32 --batch 1` = 10.2 ms per "
                         "training step (~900 launches), 2.2 ms per eval forward (DESIGN.md section 5)",
            "higher_is_better": True,
            # f16x2 kernels: per step, operand maxima that came with the tensor (emitted by its producer) / needed a read pass
            "f16x2_operand_maxima_per_step": {k: (_ops_c.AMAX_STATS[k] - amax0[k]) / args.steps for k in amax0},
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "dtype_note": ("fp32 tensors and fp32 accumulation everywhere; the 3x3x3 stride-1 convolutions and their weight "
                           "gradients scale every CHANNEL of an fp32 operand by its own power of two and split it into two f16 "
                           "terms (three f16 MFMA products, operand error <= 2^-22 per channel; DCA_CONV=x3: three bf16 terms, "
                           "six products); where a BatchNorm output or gradient has such a convolution as its only reader the "
                           "BatchNorm kernel writes those two f16 terms (4 bytes per element, like fp32) instead of the fp32 value "
                           "(packed px2 operand, DCA_PACK=0 switches it off); the transposed and 1x1x1 convolutions use the "
                           "three-term bf16 split: measured against fp64 as accurate as the fp32 MFMA kernels per output channel "
                           "(tests/test_gpu_parity.py::test_conv3d_bf16x3_is_fp32_grade, test_f16x2_per_channel_scales); "
                           "DCA_CONV=fp32 selects the fp32 MFMA kernels") if args.dtype == "f32" and _conv_x3() else
                          ("fp32 MFMA kernels everywhere (DCA_CONV=fp32)" if args.dtype == "f32" else
                           "reduced-precision inference path (NOT the headline): 1/4-res activations stored as " + args.dtype +
                           ", one native MFMA product per multiply with fp32 accumulation in the 3x3x3 stride-1 and 1x1x1 "
                           "convolutions, fp32 BN folding / softmax / soft-argmin / context injection / attention; gate "
                           "|EPE - EPE_fp32 oracle| <= 1e-3 (tests/test_gpu_lowprec.py)"),
            "data": "synthetic",
            "config": {"workload": "gwcnet_dca_g (GwcNet-G + 3 DCA blocks) hot path from 1/4-res features, "
                                   f"{H_IMG}x{W_IMG} D={MAXDISP}, " + ("train step: fwd(all heads)+focal/model loss+bwd+allreduce+Adam"
                                                        if args.mode == "fwdbwd" else "eval forward"),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "world_size": world, "collective_backend": (dist.get_backend() if world > 1 else None),
                       "ms_per_step_by_rank": [round(v, 3) for v in rank_ms],
                       "mode": args.mode, "hipgraph": bool(args.graph) if args.mode == "fwd" else train_graphed,
                       **({"hipgraph_error": graph_error} if graph_error else {}),
                       "weight_prepack": "once (ops.frozen_weights)" if args.mode == "fwd" else
                       (f"every step, {prepack_n} layouts in one launch (ops.PrepackPlan)" if prepack_n else
                        "every step, one launch per layout")},
        }
        roof = kernel_roofline(device, args.dtype) if not args.shape else {}
        names = list(roof)
        # HBM bytes per launch from the PMC passes committed under profiles/ (cannot be collected inside this run)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            for n in names:
                key = roof[n].pop("pmc_key", None)
                if key in pmc:
                    roof[n]["traffic"] = round(pmc[key]["hbm_bytes_per_launch"])
                    roof[n]["algorithmic_bytes"] = pmc[key]["algorithmic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        if not args.shape:
            # whole path against the two rooflines of SURVEY.md 8(d) (algorithmic bytes / FLOP per cost volume)
            ms = dt / args.steps * 1e3 / args.batch
            gflop, gb = (PATH_FWD_GFLOP, PATH_FWD_GB) if args.mode == "fwd" else (PATH_FB_GFLOP, PATH_FB_GB)
            line["path_roofline"] = {
                "algorithmic_gflop": round(gflop, 1), "algorithmic_gb": round(gb, 2),
                "hbm_ms": round(gb / PEAK_HBM_GBS * 1e3, 3), "hbm_frac": round(gb / PEAK_HBM_GBS * 1e3 / ms, 4),
                "fp32_mfma_ms": round(gflop / PEAK_FP32_MFMA_TFLOPS, 3),
                "fp32_mfma_frac": round(gflop / PEAK_FP32_MFMA_TFLOPS / ms, 4)}
        if names:
            line["roofline"] = dict(roof[names[0]], kernel=names[0])
            line["roofline_other"] = {n: roof[n] for n in names[1:]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(m, args.mode, args.cpu_rows)
            # `vs_baseline` stays null (BASELINE.md publishes no number for this metric); the ratio to the CPU oracle timed
            # beside it is reported under its own name -- a large ratio says nothing about kernel quality, `roofline.frac` does
            if line["cpu_baseline"].get("value"):
                line["vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
