# Builds the product library (HIP, gfx950) and the CPU oracle (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := opencl_path_tracer_amd
CSRC    := $(PKG)/csrc
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -Iinclude -I$(CSRC) -Wall -Wno-unused-result

all: $(PKG)/libptamd.so oracle tests/cpp/dropin check-isa

SRCS    := $(CSRC)/pt_host.cpp $(CSRC)/pt_obj.cpp $(CSRC)/pt_kernels.hip $(CSRC)/pt_wavefront.hip $(CSRC)/pt_debug.hip $(CSRC)/pt_lbvh.hip $(CSRC)/pt_comm.hip $(CSRC)/pt_image.cpp $(CSRC)/pt_wide.cpp
HDRS    := $(CSRC)/pt_internal.hpp $(CSRC)/pt_device.hpp include/pt_api.h
OBJS    := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))

$(PKG)/libptamd.so: $(OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS) -ldl -lpthread

build/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(ABFLAGS) -c -o $@ $<

# A/B builds of the kernels for experiments: make ab AB=name ABFLAGS="-DPT_X=1" -> $(PKG)/libptamd_name.so
# (select it with PTAMD_LIB=... ; never loaded by default).  Always from scratch, every compile checked, explicit objects.
ABOBJS  = $(patsubst $(CSRC)/%,build/ab_$(AB)/%.o,$(SRCS))
ab:
	@test -n "$(AB)" || { echo "usage: make ab AB=name ABFLAGS=..."; exit 1; }
	rm -rf build/ab_$(AB) $(PKG)/libptamd_$(AB).so
	@mkdir -p build/ab_$(AB)
	$(MAKE) -j8 ab-objs AB=$(AB) ABFLAGS="$(ABFLAGS)"
	$(HIPCC) $(HIPFLAGS) -shared -o $(PKG)/libptamd_$(AB).so $(ABOBJS) -ldl -lpthread
ab-objs: $(ABOBJS)
build/ab_$(AB)/%.o: $(CSRC)/% $(HDRS)
	$(HIPCC) $(HIPFLAGS) $(ABFLAGS) -c -o $@ $<

# Static guard (no GPU needed): the ISA of every k_render / wf_intersect instance keeps its wave-uniform loop state in
# scalar registers (tools/check_isa.py)
build/isa/%.s: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/isa
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S -o $@ $< 2>/dev/null
check-isa: build/isa/pt_kernels.s build/isa/pt_wavefront.s tools/check_isa.py
	python3 tools/check_isa.py build/isa/pt_kernels.s build/isa/pt_wavefront.s > build/isa/check.log || { cat build/isa/check.log; exit 1; }
	@tail -1 build/isa/check.log

tests/cpp/dropin: tests/cpp/dropin_main.cpp include/pt_scene.hpp include/pt_api.h $(PKG)/libptamd.so
	g++ -O1 -std=c++14 -Iinclude -o $@ tests/cpp/dropin_main.cpp -L$(PKG) -lptamd -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -rf $(PKG)/libptamd*.so tests/cpp/dropin build
	$(MAKE) -C oracle clean

.PHONY: all oracle clean ab ab-objs check-isa
