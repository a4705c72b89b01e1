"""Training entry point with the shape of the reference's `hand/run.py` -> `CrossModalHand.train()` loop
(hand/CrossModalHand.py:205-225 epochs + MultiStepLR, :430-470 iteration: training_step_start, model_forward =
get_loss + the N-sample metrics pass, criterion, backward/clip/Adam, AverageMeter bookkeeping, :573-587 checkpoint),
on synthetic batches (no dataset ships with this repository; the HO3D pipeline is out of scope).

    python -m mhentropy_amd.run --backbone resnet50 --batch 256 --hyps 64 --dtype bf16 --epochs 1 --iters 20
    python -m torch.distributed.run --nproc-per-node 8 -m mhentropy_amd.run ...      # one process per GPU, RCCL
"""
import argparse
import json
import sys
import time

import torch

from . import dist as mdist, harness, synth
from .criteria import MHEntLoss
from .train import TrainStep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--backbone", default="resnet50", choices=["resnet18", "resnet50"])
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (configs/ho3d.yaml: 64)")
    ap.add_argument("--hyps", type=int, default=10, help="hypotheses per image in the loss (the reference hard-codes 10)")
    ap.add_argument("--test-samples", type=int, default=0, help="per-iteration metrics pass, training.test_samples (reference: 200)")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--flow-steps", type=int, default=6)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10, help="iterations per epoch")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--milestones", type=int, nargs="*", default=[150, 250])
    ap.add_argument("--save", default="", help="write a checkpoint in the reference's container format here (rank 0)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cfg", default="", help="a YAML file in the reference's schema (hand/configs/ho3d.yaml): network / training sections "
                                              "override the flags above (backbone, h_dims, num_steps, batch_size, lr, milestones, test_samples)")
    ap.add_argument("--load", default="", help="checkpoint in the reference's container format to start from (training.pth)")
    ap.add_argument("--scalars", default="", help="write the reference's TensorBoard scalars here as JSON lines (rank 0)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: replay the whole iteration (forward, reverse, all-reduce, clip, Adam) from HIP graphs (train.GraphedStep; batches are "
                         "copied into static tensors, the graphs are re-captured when MultiStepLR changes the rate)")
    ap.add_argument("--input-pipeline", action="store_true",
                    help="feed the step from DECODED synthetic HO3D samples through the GPU input pipeline (ho3d_dataloader.HO3DBatchPipeline: "
                         "crop, augmentation, visibility, compute_st; hand/dataloader/ho3d_dataloader.py:272-459) instead of ready-made batches; "
                         "image size is then 256")
    args = ap.parse_args(argv)
    cfg = None
    if args.cfg:
        cfg = harness.load_config(args.cfg)
        args.backbone, args.hidden, args.flow_steps = cfg.network.backbone, cfg.network.h_dims[0], cfg.network.num_steps
        args.batch, args.lr, args.milestones = cfg.training.batch_size, cfg.training.lr, list(cfg.training.milestones)
        args.test_samples = cfg.training.get("test_samples", args.test_samples)

    rank, local_rank, world, dist = mdist.init()
    if not torch.cuda.is_available():
        raise SystemExit("mhentropy_amd.run needs a HIP device (there is no CPU path)")
    torch.cuda.set_device(local_rank if world > 1 else 0)
    torch.manual_seed(args.seed)                      # same initial weights on every rank
    from . import ops
    ops.rng_state(torch.device("cuda", torch.cuda.current_device()), seed=args.seed + 7919 * rank)      # the device generator of the base noise
    cd = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if cfg is not None:
        from .network import MHEnt
        special, common = harness.mhent_cfgs_from_config(cfg, tables=synth.mano_tables(0), compute_dtype=cd)
        model = MHEnt(special, **common)
        model.q_z_giv_i.compute_dtype = cd
        model = model.cuda().train()
    else:
        model = harness.build_mhent(backbone=args.backbone, h_dims=(args.hidden, args.hidden), num_steps=args.flow_steps,
                                    tables=synth.mano_tables(0), compute_dtype=cd).cuda().train()
    if args.load:
        ck = harness.load_model(args.load, model, map_location="cuda")
        if isinstance(ck, dict) and ck.get("mhe_rng_state") is not None and rank == 0 and world == 1:
            # continue the checkpointed run's base-noise stream (single process; under data parallelism every rank keeps its own seed)
            ops.rng_set_state(torch.device("cuda", torch.cuda.current_device()), ck["mhe_rng_state"])
    scalars = harness.ScalarLog(args.scalars) if (args.scalars and rank == 0) else None
    trainer = TrainStep(model, lr=args.lr, max_norm=1.0, dist=dist)
    criterion = MHEntLoss()
    meters = {"loss": harness.AverageMeter(), "epe3d": harness.AverageMeter(), "epe2d": harness.AverageMeter()}
    step, log = 0, []
    graphed, graphed_lr, sx, sy = None, None, None, None
    if args.input_pipeline:
        import numpy as np
        from . import ho3d_dataloader as hd
        pipe = hd.HO3DBatchPipeline()
        pool = [synth.ho3d_sample(args.seed + 50 * rank + i, ((i * 37) % 400 - 200, (i * 53) % 300 - 150)) for i in range(8)]
        aug_rng = np.random.RandomState(args.seed + rank)
    for epoch in range(args.epochs):
        trainer.lr = harness.multistep_lr(args.lr, epoch, tuple(args.milestones))
        for m in meters.values():
            m.reset()
        t0 = time.time()
        it_losses = []
        for it in range(args.iters):
            if args.input_pipeline:
                order = aug_rng.randint(0, len(pool), args.batch)
                x, y = pipe(hd.collate_decoded([pool[i] for i in order]), aug=hd.draw_aug(args.batch, aug_rng))
            else:
                xn, yn = synth.batch(args.seed + 1000 * rank + step, args.batch, image_size=args.image_size)
                x = torch.as_tensor(xn).cuda()
                y = {k: torch.as_tensor(v).cuda() for k, v in yn.items()}
            model.training_step_start(step)
            if args.graph and not args.test_samples:
                from .train import GraphedStep
                yk = {k: y[k].contiguous() for k in ("crop_uv", "vis")}          # what the loss consumes
                if graphed is None or graphed_lr != trainer.lr:
                    # (re-)capture: GraphedStep's warm-up pass is one real step on this batch - it IS this iteration (replaying the
                    # same batch as well would apply two optimizer steps to it and advance Adam's count and the BatchNorm buffers twice)
                    sx, sy = x.clone(), {k: v.clone() for k, v in yk.items()}
                    graphed, graphed_lr = GraphedStep(trainer, sx, sy, N=args.hyps), trainer.lr
                    out = graphed.warm_out
                else:
                    sx.copy_(x)
                    for k, v in yk.items():
                        sy[k].copy_(v)
                    out = graphed.replay()
            else:
                out = trainer.step(x, y, N=args.hyps, test_samples=args.test_samples)
            with torch.no_grad():
                total, losses, metrics = criterion(dict(out), y)
            meters["loss"].update(float(total))
            it_losses.append(float(total))
            if scalars is not None:
                scalars.iteration(step, losses, metrics, out)
            if args.test_samples:
                meters["epe3d"].update(float(metrics["eucLoss_3d_rgb_sample"].mean()))
                meters["epe2d"].update(float(metrics["eucLoss_2d_rgb_sample"].mean()))
            step += 1
        torch.cuda.synchronize()
        rec = mdist.reduce_mean_scalars({k: float(m.avg) for k, m in meters.items()}, dist, device=torch.device("cuda"))
        rec.update(epoch=epoch, lr=trainer.lr, img_per_s=round(world * args.batch * args.iters / (time.time() - t0), 1),
                   it_losses=[round(v, 4) for v in it_losses])          # this rank's per-iteration totals (the resume test reads the first)
        log.append(rec)
        if scalars is not None:
            scalars.epoch(step, rec["loss"], rec["epe3d"])
        if rank == 0:
            print(json.dumps(rec), flush=True)
    if scalars is not None:
        scalars.close()
    if args.save and rank == 0:
        harness.save_model(args.save, model, rng_words=ops.rng_get_state(torch.device("cuda", torch.cuda.current_device())))
    if dist is not None:
        dist.destroy_process_group()
    main.last_trainer = trainer           # (tests: optimizer step count, parameters)
    return log


if __name__ == "__main__":
    main(sys.argv[1:])
