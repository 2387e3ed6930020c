bash tools/prof.sh r02l_c3 --config C3_suzanne --spp 64 > gpurun_out/r02l_c3.log 2>&1; echo prof rc=$?
