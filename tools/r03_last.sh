#!/bin/bash
# after the rocprofv3 summaries are in profiles/: the bench line once more (it quotes their PMC traffic), and the lanes A/B of the tree-walk configs
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
bash tools/r03_lanes.sh $1
