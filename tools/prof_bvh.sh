# rocprofv3 kernel statistics for the BVH configs at reduced spp (evidence for profiles/)
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/prof_bvh; mkdir -p $R/gpurun_out/prof_bvh && cd /tmp
for c in "C3_suzanne 64" "C5_part2_all 16" "C1_random_spheres 64"; do set -- $c
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bvh/$1 -- python3 $R/bench.py --config $1 --spp $2 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bvh/$1.log 2>&1
done
echo done
