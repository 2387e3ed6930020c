# Builds the product library (HIP kernels + C ABI) and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: float expressions round exactly as written, on the device AND in the host-side
# scene/camera code, so hit/miss decisions follow the CPU oracle bit for bit.
# -fno-slp-vectorize: packed f32 (v_pk_mul/add/fma) issues at half the rate of the scalar forms on gfx950
# (tools/microbench.hip), so SLP-packed pairs gain nothing and their repacking v_movs cost: k_extend<false> 25.1 -> 20.3 ms.
HIPFLAGS ?= -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=$(ARCH) -Wall -Wno-unused-function
SRC = firework_amd/csrc/fw_kernels.hip firework_amd/csrc/fw_runtime.cpp
HDR = firework_amd/csrc/fw_device.h firework_amd/csrc/fw_libm.h include/firework_hip.h Makefile
LIB = firework_amd/lib/libfirework_hip.so

all: $(LIB) oracle examples

$(LIB): $(SRC) $(HDR)
	mkdir -p firework_amd/lib
	$(HIPCC) $(HIPFLAGS) -x hip -shared -o $@ $(SRC)

# the A/B build: the product plus the alternative kernels that were measured slower and their switches (fw_device.h: FW_AB);
# tests/ and tools/ load it with FIREWORK_LIB=firework_amd/lib/variants/lib_ab.so
ab: firework_amd/lib/variants/lib_ab.so
firework_amd/lib/variants/lib_ab.so: $(SRC) $(HDR)
	mkdir -p firework_amd/lib/variants
	$(HIPCC) $(HIPFLAGS) -DFW_AB=1 -x hip -shared -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle

# C++ host-side mirror of the reference API (include/firework.hpp): plain g++, links the C ABI only
examples: examples/cornell_box

examples/cornell_box: examples/cornell_box.cpp include/firework.hpp include/firework_hip.h $(LIB)
	g++ -O2 -std=c++17 -Iinclude -o $@ examples/cornell_box.cpp -Lfirework_amd/lib -lfirework_hip -Wl,-rpath,'$$ORIGIN/../firework_amd/lib'

clean:
	rm -f $(LIB) examples/cornell_box
	$(MAKE) -C oracle clean

.PHONY: all ab oracle examples clean
