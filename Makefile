# Builds the product library (HIP kernels + C ABI) and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: float expressions round exactly as written, on the device AND in the host-side
# scene/camera code, so hit/miss decisions follow the CPU oracle bit for bit.
HIPFLAGS ?= -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=$(ARCH) -Wall -Wno-unused-function
SRC = firework_amd/csrc/fw_kernels.hip firework_amd/csrc/fw_runtime.cpp
HDR = firework_amd/csrc/fw_device.h include/firework_hip.h
LIB = firework_amd/lib/libfirework_hip.so

all: $(LIB) oracle

$(LIB): $(SRC) $(HDR)
	mkdir -p firework_amd/lib
	$(HIPCC) $(HIPFLAGS) -x hip -shared -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
