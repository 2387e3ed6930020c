# build a library variant into firework_amd/lib/variants/lib_<name>.so:  tools/build_variant.sh <name> [-DFLAG=..]...
NAME=$1; shift
mkdir -p firework_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-function "$@" -x hip -shared \
  -o firework_amd/lib/variants/lib_$NAME.so firework_amd/csrc/fw_kernels.hip firework_amd/csrc/fw_runtime.cpp 2>&1 | grep -E "error|warning: v" ; echo built $NAME
