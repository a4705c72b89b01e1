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


def save_model(path, model, rng_words=None):
    """checkpoint in the reference's format (hand/CrossModalHand.py:573-587): {'decoderPose': ..., 'encoderRGB': state_dict}.
    The reference builds `decoderPose` as an empty nn.Sequential for MHEnt (its state_dict is empty).  rng_words (optional, an extra
    key the reference's loader ignores): the device generator's {seed, counter, draws} (ops.rng_get_state) so that a resumed run
    continues the base-noise stream instead of replaying it from counter 0."""
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ck = {"decoderPose": {}, "encoderRGB": sd}
    if rng_words is not None:
        ck["mhe_rng_state"] = torch.as_tensor(rng_words, dtype=torch.int64).cpu().clone()
    torch.save(ck, path)


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


# ---- configuration ingestion (reference hand/configs/config.py:13-93 defaults + merge_from_file; hand/configs/ho3d.yaml) -----
class _Node(dict):
    """attribute access over nested dicts (what the reference gets from yacs' CfgNode / EasyDict)"""
    def __getattr__(self, k):          # AttributeError like yacs' CfgNode / EasyDict: getattr(cfg.x, k, default), hasattr and copy.deepcopy rely on it
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _node(d):
    return _Node({k: _node(v) if isinstance(v, dict) else v for k, v in d.items()})


# the defaults of the reference's config tree for the keys this path consumes (hand/configs/config.py:13-64)
CONFIG_DEFAULTS = {
    "info_interval": 200, "save_interval": 5, "eval_interval": 1, "eval_mscoco": False,
    "model_dir": "./model/", "pretrain_model": "./model/pretrain.pth", "final_model": "./model/final.pth",
    "dataset": {"dataset_name": "rhd", "image_size": [256, 256], "range_": [[-5., -5., -5.], [5., 5., 5.]], "pe": "3d", "jointN": 21},
    "training": {"mode": "pretrain", "seed": None, "view_correction": True, "batch_size": 32, "num_workers": 32, "pth": None,
                 "load_mod_names": None, "epochs": 80, "lr": 1e-4, "milestones": [30, 60], "warmups": 0, "criterion": "ELBOLoss"},
    "network": {"enc_type": "BasicEnc", "num_latent": 64, "nums_latent": None, "backbone": "resnet18", "resnet_pretrained": True,
                "conditional_p": False, "conditional_i": False, "feat_dim": None, "acts": "exp", "deterministic": False,
                "decoder_type": "mano", "pgm": None, "p_nf": None, "p_nf_dim": 3, "tsfm_on": None, "cond_mapping_dims": None,
                "iterative_refinement": False, "kemb": False, "h_dims": [64, 64], "num_steps": 3, "nf_res": None, "ddpm": False},
    "loss": {"kl": 0.0001},
}


def load_config(path):
    """a YAML file in the reference's schema merged over its defaults, like `update_cfg` (hand/configs/config.py:70-73);
    sections `training` and `network` accept new keys (CN(new_allowed=True), :29,:45)"""
    import copy
    import yaml
    cfg = copy.deepcopy(CONFIG_DEFAULTS)
    with open(path) as fh:
        user = yaml.safe_load(fh) or {}
    for k, v in user.items():
        if isinstance(v, dict):
            if k not in cfg:
                raise KeyError(f"config section {k!r} does not exist in the reference's tree")
            for kk, vv in v.items():
                if kk not in cfg[k] and k not in ("training", "network"):
                    raise KeyError(f"config key {k}.{kk} does not exist in the reference's tree")
                cfg[k][kk] = vv
        else:
            if k not in cfg or isinstance(cfg[k], dict):       # merge_from_file rejects keys the tree does not declare
                raise KeyError(f"config key {k!r} does not exist in the reference's tree")
            cfg[k] = v
    return _node(cfg)


def mhent_cfgs_from_config(cfg, tables=None, compute_dtype=torch.float32):
    """(special_cfg, common_cfg) exactly as hand/CrossModalHand.py:53-85 derives them from the config tree"""
    n, ds = cfg.network, cfg.dataset
    if n.enc_type != "MHEnt":
        raise NotImplementedError(f"enc_type {n.enc_type!r}: only the MHEnt encoder is part of this path")
    common = dict(n_latent=n.nums_latent if n.nums_latent else n.num_latent, backbone=n.backbone, pretrained=False,
                  conditional_p=n.conditional_p, K=ds.jointN, D=int(ds.pe[0]), feat_dim=n.feat_dim, sigma_act=n.acts,
                  deterministic=n.deterministic, input=n.get("input", "image"), compute_dtype=compute_dtype)
    special = dict(
        q_z_giv_i_model=n.regressor,
        q_z_giv_i_cfg=dict(dim=45, tsfm_on=n.num_latent, kemb=False, jointN=ds.jointN, h_dims=list(n.h_dims), num_steps=n.num_steps),
        ds=ds.dataset_name, image_size=list(ds.image_size),
        mano_cfg=dict(flat_hand_mean=False, ncomps=45, use_pca=True, tables=tables),
        prior_cfg=dict(p_theta45_pth=n.rot_prior, th45_ref_alpha=n.w_reg_th),
        data_prior_cfg=dict(b_2d=n.b_2d, w_prior_2d=n.w_prior_2d),
        loss_cfg=dict(entropy=n.entropy, mode=n.mode, w_reg_ds=n.w_reg_ds),
        kld_w=1, kld_w_annealing=[1, 20 * 1200], T=1.0)
    return special, common


class ScalarLog:
    """the scalars the reference sends to TensorBoard (hand/CrossModalHand.py:486-564), under its tags: `loss_it/*`, `metric_it/*`,
    `loss_avg/loss_total`, `metric_train|eval/eval_3d_rgb` (x1000), `param/{beta,theta}_norm`.  Written as JSON lines, and to a
    SummaryWriter as well when tensorboard is importable (tensorboardX, the reference's writer, is not a dependency)."""
    def __init__(self, path):
        self.fh = open(path, "a")
        self.tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            import os
            self.tb = SummaryWriter(os.path.dirname(os.path.abspath(path)))
        except Exception:
            pass

    def add_scalar(self, tag, value, global_step):
        import json
        v = float(value)
        self.fh.write(json.dumps({"tag": tag, "value": v, "step": int(global_step)}) + "\n")
        self.fh.flush()
        if self.tb is not None:
            self.tb.add_scalar(tag, v, global_step=global_step)

    def iteration(self, step, losses, metrics, out, train=True):
        """per-iteration scalars of one model_forward + criterion (hand/CrossModalHand.py:486-530)"""
        for k, v in losses.items():
            self.add_scalar(f"loss_it/{k}", v.mean(), step)
        for k in ("eucLoss_3d_rgb_sample", "eucLoss_2d_rgb_sample"):
            if k in metrics:
                self.add_scalar(f"metric_it/{k}", metrics[k].mean(), step)
        for name, key in (("theta", "th_norm"), ("beta", "bt_norm")):
            if key in out:
                self.add_scalar(f"param/{name}_norm", out[key].mean(), step)

    def epoch(self, step, loss_avg, eval_3d_avg, train=True):
        """hand/CrossModalHand.py:557-564"""
        if train:
            self.add_scalar("loss_avg/loss_total", loss_avg, step)
        self.add_scalar(f"metric_{'train' if train else 'eval'}/eval_3d_rgb", eval_3d_avg * 1000.0, step)

    def close(self):
        self.fh.close()
        if self.tb is not None:
            self.tb.close()
