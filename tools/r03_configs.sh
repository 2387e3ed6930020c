#!/bin/bash
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
bash tools/configs.sh 2>&1 | tee $OUT/configs_full.jsonl
