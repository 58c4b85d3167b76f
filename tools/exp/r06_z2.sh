#!/bin/bash
# round 6, closing run, second part (one box): smoke, the bench line, and -- on the same box -- the rocprofv3 kernel stats and HBM counter
# passes of the headline leg the line's roofline object is about
export TAG=r06_z
tools/gpu.sh smoke bench stats hbm
