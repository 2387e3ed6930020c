#!/bin/bash
# round 3, step 1: the GPU suite, then the divergence hunt on the three configs whose ray counts differ from the oracle's
set -o pipefail
mkdir -p gpurun_out/r03a
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03a/pytest.log
tail -3 gpurun_out/r03a/pytest.log
timeout -k 10 400 python tools/diverge.py C2_cornell_box 512 512 1024 --max-pixels 4 --tol 1e-4 --out gpurun_out/r03a/diverge_C2.json > gpurun_out/r03a/diverge_C2.log 2>&1; echo "C2 rc=$?"
timeout -k 10 300 python tools/diverge.py C3_suzanne 1280 720 16 --max-pixels 8 --tol 1e-4 --out gpurun_out/r03a/diverge_C3.json > gpurun_out/r03a/diverge_C3.log 2>&1; echo "C3 rc=$?"
timeout -k 10 300 python tools/diverge.py C5_part2_all 1920 1080 4 --max-pixels 8 --tol 1e-4 --out gpurun_out/r03a/diverge_C5.json > gpurun_out/r03a/diverge_C5.log 2>&1; echo "C5 rc=$?"
tail -2 gpurun_out/r03a/diverge_C*.log
