"""Mirror of the reference's `utils` package for what the training scripts use on this path
(`from utils import *`, main_dca.py:11): the learning-rate schedule and the disparity metrics."""
from .experiment import adjust_learning_rate, learning_rate_adjust  # noqa: F401
from .metrics import D1_metric, EPE_metric, Thres_metric  # noqa: F401
