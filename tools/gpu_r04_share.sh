#!/bin/bash
mkdir -p gpurun_out/r04
for n in 8 4 2; do timeout -k 10 300 python tools/rank_share.py $n 1024 1024 256 4 || exit 1; done | tee gpurun_out/r04/rank_share.txt
