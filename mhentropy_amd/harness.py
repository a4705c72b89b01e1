"""Builders for the model exactly as the reference trainer assembles it
(reference hand/CrossModalHand.py:55-86: special_cfg / common_cfg) with the shipped
hyper-parameters of hand/configs/ho3d.yaml, plus the train-step plumbing the
"img/s" metric wraps (hand/CrossModalHand.py:191-203,455-470)."""
import torch

from .network import MHEnt
from .criteria import MHEntLoss


def mhent_cfgs(backbone="resnet50", h_dims=(512, 512), num_steps=6, tables=None, compute_dtype=torch.float32, flow="realnvp"):
    special = dict(
        q_z_giv_i_model=flow,                                                   # ho3d.yaml:39 ships "realnvp"
        q_z_giv_i_cfg=dict(dim=45, tsfm_on=512, kemb=False, jointN=21, h_dims=list(h_dims), num_steps=num_steps),
        ds="ho3d", image_size=[256, 256],
        mano_cfg=dict(flat_hand_mean=False, ncomps=45, use_pca=True, tables=tables),
        prior_cfg=dict(p_theta45_pth=None, th45_ref_alpha=50),                  # ho3d.yaml:40-41
        data_prior_cfg=dict(b_2d=0.03, w_prior_2d=0),                           # ho3d.yaml:42,44
        loss_cfg=dict(entropy=True, mode=False, w_reg_ds=0),
        kld_w=1, kld_w_annealing=[1, 20 * 1200], T=1.0)
    common = dict(n_latent=512, backbone=backbone, pretrained=False, conditional_p=False, K=21, D=3, feat_dim=None,
                  sigma_act="exp", deterministic=False, input="image", compute_dtype=compute_dtype)
    return special, common


def build_mhent(**kw):
    special, common = mhent_cfgs(**kw)
    model = MHEnt(special, **common)
    model.q_z_giv_i.compute_dtype = kw.get("compute_dtype", torch.float32)
    return model


# ---- trainer-shell parity -------------------------------------------------------
class AverageMeter:
    """reference hand/utils.py:75-91, including its quirk: `n` is overwritten by `int(val != 0)`, so zero values are
    skipped and every non-zero update counts once whatever `n` was passed."""
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        n = int(val != 0)
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count != 0 else 0


def save_model(path, model):
    """checkpoint in the reference's format (hand/CrossModalHand.py:573-587): {'decoderPose': ..., 'encoderRGB': state_dict}.
    The reference builds `decoderPose` as an empty nn.Sequential for MHEnt (its state_dict is empty)."""
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.save({"decoderPose": {}, "encoderRGB": sd}, path)


def load_model(path, model, map_location=None):
    """hand/CrossModalHand.py:589-602: loads check_point['encoderRGB'] into the model; like the reference it reports a
    key/shape mismatch instead of raising.  Works with a TrainStep attached (parameters are views into its flat buffer:
    load_state_dict copies in place; call trainer.repack() afterwards)."""
    ck = torch.load(path, map_location=map_location or "cpu")
    try:
        model.load_state_dict(ck["encoderRGB"])
    except RuntimeError as e:
        print(e)
    return ck


def multistep_lr(base_lr, epoch, milestones=(150, 250), gamma=0.1):
    """torch.optim.lr_scheduler.MultiStepLR as the reference configures it (hand/CrossModalHand.py:202, run per epoch)"""
    return base_lr * gamma ** sum(epoch >= m for m in milestones)
