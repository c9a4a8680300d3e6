#!/bin/bash
# Dev tool (AB_CMD of tools/ab_lib.sh): the three training legs, short — config [1] on the 16-bit torso, C64 parity mode, C64 on the torso
python3 tools/train_step_m32.py 30 1 2>&1 | grep value | sed 's/^/m32_mixed16 value /'
STEPS=30 REGIONS=2 python3 tools/exp_train.py 2>&1 | grep -i "ms" | tail -2 | sed 's/^/c64_parity value /'
FP16=1 STEPS=30 REGIONS=2 python3 tools/exp_train.py 2>&1 | grep -i "ms" | tail -2 | sed 's/^/c64_torso value /'
