#!/bin/bash
for rep in 1 2 3; do for a in 0 1; do echo "DVO_COPY_AHEAD=$a"; DVO_COPY_AHEAD=$a python tools/experiments/exp_copy_order.py 2>&1 | grep "ms per"; done; done
