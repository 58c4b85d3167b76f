#!/bin/bash
# round 6, the closing run on the FINAL library (after the one-shot upload priorities): suite, smoke, bench, stats + hbm of the headline leg
export TAG=r06_z
tools/gpu.sh tests smoke bench stats hbm bench2
