"""Learning-rate schedules of the reference's training scripts."""


def adjust_learning_rate(optimizer, epoch, base_lr, lrepochs, verbose=False):
    """reference utils/experiment.py:91-109: `lrepochs` = "12,20,24,28:2" -> divide the rate by 2 at each listed epoch
    that has been reached (main_dca.py:254).  Returns the rate it set (the reference prints it)."""
    splits = lrepochs.split(":")
    assert len(splits) == 2
    downscale_epochs = [int(e) for e in splits[0].split(",")]
    downscale_rate = float(splits[1])
    lr = base_lr
    for eid in downscale_epochs:
        if epoch >= eid:
            lr /= downscale_rate
        else:
            break
    if verbose:
        print("setting learning rate to {}".format(lr))
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def learning_rate_adjust(optimizer, epoch, verbose=False):
    """reference util.py:132-145 (train_kitti.py:171, train_eth3d.py:156): 1e-3 below epoch 300, 1e-4 below 600, then
    1e-5."""
    lr = 0.001 if epoch < 300 else (0.0001 if epoch < 600 else 0.00001)
    if verbose:
        print("learning rate = %.5f" % lr)
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr
