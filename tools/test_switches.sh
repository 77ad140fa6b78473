#!/bin/bash
# Run the GPU suite under every A/B / opt-in switch of the library and the engines (on the GPU box: gpurun -- 'bash tools/test_switches.sh').
# Each line must end with the same "N passed" as the default run; MS_STYLE_FUSED=0 skips the one test that asserts the single-read kernel ran.
set -u
for sw in "" MS_CONV_WIDE=0 MS_FUSE_ACTBWD=0 MS_FUSE_BNFIN=1 MS_INLINE_BN_BWD=1 MS_OVERLAP=1 MS_TRAIN_GRAPH=1 MS_STYLE_FUSED=0 MS_STYLE_FUSED_MIN_MB=2 \
          MS_XFIN=0 MS_XFIN_PRO=0 MS_FUSE_TAIL=0 MS_FUSE_HEAD_BWD=0 MS_LAZY_SEG_TAIL=0 MS_RIDE=0 MS_POOL_FUSE=0 MS_POOL_EPI=0 MS_LOOP_WINOGRAD=0 MS_SHARED_DEVICE=1; do
  echo -n "${sw:-default}: "
  env $sw timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -1
done
