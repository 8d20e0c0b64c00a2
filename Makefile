# Builds the product library (HIP, gfx950) and the CPU oracle (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := opencl_path_tracer_amd
CSRC    := $(PKG)/csrc
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -Iinclude -I$(CSRC) -Wall -Wno-unused-result

all: $(PKG)/libptamd.so oracle tests/cpp/dropin

$(PKG)/libptamd.so: $(CSRC)/pt_host.cpp $(CSRC)/pt_obj.cpp $(CSRC)/pt_kernels.hip $(CSRC)/pt_lbvh.hip $(CSRC)/pt_internal.hpp include/pt_api.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/pt_host.cpp $(CSRC)/pt_obj.cpp $(CSRC)/pt_kernels.hip $(CSRC)/pt_lbvh.hip

tests/cpp/dropin: tests/cpp/dropin_main.cpp include/pt_scene.hpp include/pt_api.h $(PKG)/libptamd.so
	g++ -O1 -std=c++14 -Iinclude -o $@ tests/cpp/dropin_main.cpp -L$(PKG) -lptamd -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -f $(PKG)/libptamd.so tests/cpp/dropin
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
