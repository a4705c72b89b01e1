"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the criterion and metrics the reference evaluates on every
step (/root/reference/hand/criteria.py:42-173, helper hand/utils.py:21-30).
Pinned by tests/golden/criteria_*.npz.
"""
import torch


def mean_euclidean(pred, gt, scale):
    """utils.py:21-30 with reduction='none'."""
    pred = pred.view(pred.shape[0], -1, 3)
    gt = gt.reshape(pred.shape[0], -1, 3)
    d = torch.squeeze(torch.sqrt(torch.sum((pred - gt) ** 2, dim=2)))
    return d * torch.squeeze(scale).view(scale.shape[0], 1)


def _group_stats(stats, weight, B):
    """criteria.py:116-132."""
    num_vis = weight.sum(-1)
    mpj = (stats * weight).sum(-1) / (num_vis + 1e-16)
    if num_vis.dim() == 2:
        num_vis = num_vis[0]
    num_valid = (num_vis > 0.0).sum().item()
    return mpj * B / (num_valid + 1e-16) if num_valid else mpj * 0.0


def mhent_loss(output, target):
    """MHEntLoss.forward, criteria.py:47-173 (aligned=False, chamfer off).
    output: log_p (B,), xyz (N,B,63), uv (N,B,42).  Returns (total, losses, metrics)."""
    losses = {"neg_log_p": -output["log_p"]}
    metrics = {}
    N, B = output["xyz"].shape[:2]
    xyz = torch.flatten(output["xyz"], 0, 1)
    euc3d = mean_euclidean(xyz, target["pose3d"].repeat(N, 1), target["scale"].repeat(N)).reshape(N, B, -1)
    uv_gt = (target["crop_uv"] + 1.0) / 2.0 * 256
    euc2d = (output["uv"] - uv_gt).reshape(N, B, -1, 2).norm(p=2, dim=-1)
    weights = {
        "sample": torch.ones_like(target["vis"]),
        "vis": (target["vis"] == 1.0).float(),
        "invis": (target["vis"] != 1.0).float(),
    }
    weights["vis"][:, 12] = 0.0          # criteria.py:112-114
    weights["invis"][:, 12] = 0.0
    for sup, euc, D in (("3d", euc3d, 3), ("2d", euc2d, 2)):
        coord = output["xyz"] * target["scale"][:, None] if sup == "3d" else output["uv"]
        coord = coord.reshape(N, B, -1, D)
        for attr, w in weights.items():
            key = f"eucLoss_{sup}_rgb_{attr}"
            mpjpe = _group_stats(euc, w[None].repeat(N, 1, 1), B)
            metrics[key] = mpjpe.max(0)[0] if (sup == "2d" and attr == "vis") else mpjpe.min(0)[0]
            if N == 1:
                sp = torch.zeros(B, coord.shape[-2])
            else:
                sp = coord.std(0).prod(-1)
            sp = sp ** (1 / D) * (D ** 0.5)
            metrics[f"{key}_std"] = _group_stats(sp, w, B)
            if attr == "vis":
                metrics[f"{key}_mean"] = _group_stats(euc.mean(0), w, B)
    return sum(v.mean() for v in losses.values()), losses, metrics
