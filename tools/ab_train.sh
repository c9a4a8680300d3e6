#!/bin/bash
# Dev: same-box A/B of the training step under an env switch:  bash tools/ab_train.sh CDAE_TRAIN_PRESPLIT 0 1
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  echo -n "$VAR=$v  "
  env $VAR=$v python tools/train_step.py 8 32 2>/dev/null | tail -1
done; done
