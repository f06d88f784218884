#!/bin/bash
# sweeps the slab-form schedule parameters of the weight-gradient kernel (run on the GPU box)
L="dec_convT1_wgrad dec_convT2_wgrad dec_convT3_wgrad dec_last_wgrad enc_conv1_wgrad enc_conv2_wgrad enc_conv3_wgrad enc_conv4_wgrad"
for cap in 8388608 16777216 33554432; do for blocks in 1024 2048 4096; do
  echo "== cap=$cap blocks=$blocks"
  MMVAE_IMG=0 MMVAE_WGRAD_SLAB_CAP=$cap MMVAE_WGRAD_SLAB_BLOCKS=$blocks python tools/layer_bench.py $L 2>&1 | grep wgrad
done; done
