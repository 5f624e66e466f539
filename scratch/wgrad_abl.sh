#!/bin/bash
for a in ${ABLS:-0 1 3 5 6}; do
  MT355_WGRAD_ABL=$a TAG=abl$a python3 scratch/wgrad_lab.py 2>&1 | grep "wgrad N32 256x256 128->128 k3"
done
