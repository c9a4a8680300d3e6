#!/usr/bin/env python3
"""Training entry point with the reference's flag set (scripts/image_train.py:87-113): every key of
`model_and_diffusion_defaults()` plus data_dir / schedule_sampler / lr / weight_decay / lr_anneal_steps / batch_size /
microbatch / ema_rate / log_interval / save_interval / resume_checkpoint / use_fp16 / fp16_scale_growth / rep_cond /
n_vars / causal_modeling / flow_based / in_channels / masking.  Additions: --log_dir (the reference edits a hard-coded
path in the source), --split, --host_feed (default: the dataset shard lives in HBM and batches are assembled on the GPU).

    python scripts/image_train.py --data_dir ../datasets/pendulum --image_size 96 --in_channels 4 --n_vars 4 \
        --rep_cond True --causal_modeling True --batch_size 32 --log_dir ../results/pendulum/causaldiffae
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/image_train.py ...
    DIFFUSION_TRAINING_TEST=1 python scripts/image_train.py --data_dir synthetic ...      # stops after the first save
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from improved_diffusion import dist_util, logger                                             # noqa: E402
from improved_diffusion.image_datasets import load_data                                      # noqa: E402
from improved_diffusion.resample import create_named_schedule_sampler                        # noqa: E402
from improved_diffusion.script_util import (add_dict_to_argparser, args_to_dict,            # noqa: E402
                                            create_model_and_diffusion, model_and_diffusion_defaults)
from improved_diffusion.train_util import TrainLoop                                          # noqa: E402


def create_argparser():
    defaults = dict(data_dir="", schedule_sampler="uniform", lr=1e-4, weight_decay=0.0, lr_anneal_steps=0, batch_size=1,
                    microbatch=-1, ema_rate="0.9999", log_interval=10, save_interval=5000, resume_checkpoint="",
                    use_fp16=False, fp16_scale_growth=1e-3, rep_cond=False, n_vars=None, causal_modeling=False,
                    flow_based=False, in_channels=3, masking=False,
                    log_dir="", split="train", host_feed=False)
    defaults.update(model_and_diffusion_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


def main():
    args = create_argparser().parse_args()
    if args.n_vars is not None:
        args.n_vars = int(args.n_vars)
    dist_util.setup_dist()
    logger.configure(dir=args.log_dir or None)
    logger.log("creating model and diffusion...")
    model, diffusion = create_model_and_diffusion(**args_to_dict(args, model_and_diffusion_defaults().keys()))
    model.to(dist_util.dev())
    schedule_sampler = create_named_schedule_sampler(args.schedule_sampler, diffusion)
    logger.log("creating data loader...")
    data = load_data(data_dir=args.data_dir, batch_size=args.batch_size, image_size=args.image_size, class_cond=args.class_cond,
                     split=args.split, device=None if args.host_feed else dist_util.dev(),
                     in_channels=args.in_channels, n_vars=args.n_vars or 4)
    logger.log("training...")
    TrainLoop(model=model, diffusion=diffusion, data=data, batch_size=args.batch_size, microbatch=args.microbatch, lr=args.lr,
              ema_rate=args.ema_rate, log_interval=args.log_interval, save_interval=args.save_interval,
              resume_checkpoint=args.resume_checkpoint, use_fp16=args.use_fp16, fp16_scale_growth=args.fp16_scale_growth,
              schedule_sampler=schedule_sampler, weight_decay=args.weight_decay, lr_anneal_steps=args.lr_anneal_steps,
              rep_cond=args.rep_cond, n_vars=args.n_vars, causal_modeling=args.causal_modeling, flow_based=args.flow_based,
              in_channels=args.in_channels, masking=args.masking).run_loop()


if __name__ == "__main__":
    main()
