"""Compile-time geometry of the image-resident conv kernels (multimodal-vae_amd/csrc/convres_geo.h) checked on the host: the header
is plain C++17, tests/host/convres_geo_check.cpp walks every (class, output pixel, tap) of every compiled layer shape (MultiMNIST,
CelebA, COCO) and compares the LDS address the kernels use (lane base + tap immediate) with the gather semantics of the
reference convolution (multimnist/model.py:160-216, celeba/model.py:101-150): in-image taps hit the staged pixel's cell, every
other tap a cell that is never written (the zero ring)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_convres_geometry_addresses(tmp_path):
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "geo_check")
    subprocess.run([cxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "multimodal-vae_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "convres_geo_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("ok"), out[-400:]
    assert "mm_conv2" in out and "ca_conv3" in out


def test_wgrad_ring_geometry_addresses(tmp_path):
    """Ring-staged weight-gradient kernel (csrc/wgrad_ring.hip, wgrad_geo.h): the host program replays the LDS-DMA fill of a
    slot and every fragment address of every (stride-parity class, pixel row, tap) and compares what the reads return with the
    weight gradient of the reference's Conv2d / ConvTranspose2d layers (multimnist/model.py:160-169,199-208)."""
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "wgeo_check")
    subprocess.run([cxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "multimodal-vae_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "wgrad_geo_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("ok"), out[-400:]
    assert "mm_convT3" in out and "mm_conv3" in out and "ca_conv3" in out
