"""MHEntLoss with the reference's call surface (reference hand/criteria.py:42-173):
`MHEntLoss(loss_weights)(output, target) -> (total, losses, metrics)`; the metrics
block runs in one HIP kernel (csrc/metrics.hip)."""
import torch
from torch import nn

from . import ops

# row order of mhe_metrics_f32's [14,B] output
METRIC_KEYS = tuple(f"eucLoss_{sup}_rgb_{row}" for sup in ("3d", "2d")
                    for row in ("sample", "sample_std", "vis", "vis_std", "vis_mean", "invis", "invis_std"))


class MHEntLoss(nn.Module):
    def __init__(self, loss_weights=None):
        super().__init__()
        self.loss_weights = loss_weights

    def forward(self, output, target):
        losses = {"neg_log_p": -output["log_p"]}          # criteria.py:55
        metrics = {}
        if "xyz" in output:
            if "uv" not in output:
                raise NotImplementedError("uv from ground-truth s,t (criteria.py:100-104) is not on the MHEnt path")
            m = ops.metrics(output["xyz"].contiguous(), output["uv"].contiguous(), target["pose3d"].contiguous(),
                            target["scale"].contiguous(), target["crop_uv"].contiguous(), target["vis"].contiguous())
            metrics = {k: m[i] for i, k in enumerate(METRIC_KEYS)}
        return sum(v.mean() for v in losses.values()), losses, metrics
