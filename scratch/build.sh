#!/bin/bash
cd /root/repo
python multimodal-vae_amd/build.py > /tmp/mmvae_build.log 2>&1
rc=$?
grep -E "error|Error" /tmp/mmvae_build.log | head -20
echo "build rc=$rc"
ls -la --time-style=full-iso multimodal-vae_amd/libmmvae_hip.so | awk '{print $5, $6, $7}'
date
