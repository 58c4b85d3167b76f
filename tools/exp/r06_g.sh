#!/bin/bash
# round 6, batch G: the NTT pass with its register allocation sized for THREE waves per SIMD (133 VGPRs, no spill) instead of four
# (128 VGPRs, 4 spilled): same box, shipped build / variant / shipped build
export TAG=r06_g
tools/gpu.sh ab:base:ntt ab:ntt3w:ntt:bn254:ntt3w ab:base2:ntt
