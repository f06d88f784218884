#!/bin/bash
cd /root/repo && python multimodal-vae_amd/build.py 2>&1 | grep -E "error|Error" | head -20
ls -la --time-style=full-iso multimodal-vae_amd/libmmvae_hip.so | awk '{print $6, $7}'
