#!/usr/bin/env python3
"""Counterfactual sampling driver (synthetic inputs): the reference's scripts/image_causaldae_test.py pattern on the
MI355X path.  Flags follow the reference's argparse names where they exist.

    python scripts/counterfactual_demo.py --image_size 64 --in_channels 4 --n_vars 4 --timestep_respacing ddim100 \
        --batch_size 16 --graph pendulum --var_index 0 --value 0.2 [--model_path model.pt]
    python -m torch.distributed.run --nproc-per-node 8 scripts/counterfactual_demo.py ... (batch sharded, gathered at the end)
"""
import argparse
import os
import sys
import time

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from improved_diffusion import dist_util, logger                                             # noqa: E402
from improved_diffusion.counterfactual import counterfactual_sample                          # noqa: E402
from improved_diffusion.script_util import (add_dict_to_argparser, args_to_dict,            # noqa: E402
                                            create_model_and_diffusion, model_and_diffusion_defaults)


def main():
    defaults = dict(clip_denoised=True, batch_size=16, use_ddim=True, model_path="", graph="pendulum", var_index=0, value=0.2, w=-1.0)
    defaults.update(model_and_diffusion_defaults())
    defaults.update(rep_cond=True, causal_modeling=True, timestep_respacing="ddim100")
    p = argparse.ArgumentParser()
    add_dict_to_argparser(p, defaults)
    args = p.parse_args()
    if "RANK" in os.environ:
        dist_util.setup_dist()
    model, diffusion = create_model_and_diffusion(**args_to_dict(args, model_and_diffusion_defaults().keys()))
    if args.model_path:
        model.load_state_dict(dist_util.load_state_dict(args.model_path, map_location="cpu"))
    model.to(dist_util.dev()).eval()
    g = th.Generator().manual_seed(0)
    batch = th.rand(args.batch_size, args.in_channels, args.image_size, args.image_size, generator=g)
    extra = {"y": th.zeros(args.batch_size, dtype=th.int64, device=dist_util.dev())} if args.class_cond else None
    t0 = time.perf_counter()
    out = counterfactual_sample(model, diffusion, batch, args.graph, args.var_index, args.value, use_ddim=args.use_ddim,
                                w=None if args.w < 0 else args.w, clip_denoised=args.clip_denoised, extra_kwargs=extra,
                                shard="RANK" in os.environ, gather="RANK" in os.environ)
    th.cuda.synchronize()
    logger.log(f"sampled {tuple(out.shape)} in {time.perf_counter() - t0:.2f}s; range [{out.min().item():.3f}, {out.max().item():.3f}]")


if __name__ == "__main__":
    main()
