#!/bin/bash
# round 6, the closing run: one box, the final commit -- GPU suite, smoke, the bench line as the driver runs it (in-run counter passes
# included), the N = 2 line typed plainly (two ranks on this GPU over gloo), kernel stats, HBM counters, SQ accounting
export TAG=r06_z
tools/gpu.sh tests smoke bench bench2 stats hbm sq:bn:bn sq:ntt:ntt
