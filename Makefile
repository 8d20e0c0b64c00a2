# Builds the product library (HIP, gfx950) and the CPU oracle (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := opencl_path_tracer_amd
CSRC    := $(PKG)/csrc
# -amdgpu-sdwa-peephole=0: an SDWA v_cndmask_b32 can only take its mask from VCC, so the compiler copies the SGPR mask there
# (s_mov_b64 vcc, ...) -- and a VCC-masked VOP2 / SDWA select whose VCC was not just written by a VALU compare costs 21 clocks
# instead of 4 on gfx950 (tools/micro/exec_ops.hip).  Two of them sat in the node loop of k_render: +2.7 % on the Cornell box.
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 -Iinclude -I$(CSRC) -Wall -Wno-unused-result

all: $(PKG)/libptamd.so oracle tests/cpp/dropin check-isa

SRCS    := $(CSRC)/pt_host.cpp $(CSRC)/pt_builder.cpp $(CSRC)/pt_launch.cpp $(CSRC)/pt_obj.cpp $(CSRC)/pt_kernels.hip $(CSRC)/pt_wavefront.hip $(CSRC)/pt_debug.hip $(CSRC)/pt_lbvh.hip $(CSRC)/pt_sahdev.hip $(CSRC)/pt_widedev.hip $(CSRC)/pt_comm.hip $(CSRC)/pt_image.cpp $(CSRC)/pt_wide.cpp
HDRS    := $(CSRC)/pt_internal.hpp $(CSRC)/pt_context.hpp $(CSRC)/pt_device.hpp include/pt_api.h
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

# Host-side scene path (pt_add_obj: parallel parse, threaded encounter ranks; pt_upload_triangles: thread pool, parallel SAH
# top, splice, 4-wide collapse) under ThreadSanitizer and under AddressSanitizer + UBSan, on a generated 200k-triangle OBJ,
# one context and then two at once.  CPU only (host-only contexts); the device objects are linked in unchanged.
HOSTSRC := pt_host.cpp pt_builder.cpp pt_launch.cpp pt_obj.cpp pt_wide.cpp pt_image.cpp
DEVOBJ  := build/pt_kernels.hip.o build/pt_wavefront.hip.o build/pt_debug.hip.o build/pt_lbvh.hip.o build/pt_sahdev.hip.o build/pt_widedev.hip.o build/pt_comm.hip.o
SANFLAGS := -O1 -g -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -Iinclude -I$(CSRC)
sanitize-host: $(DEVOBJ)
	@for san in thread address,undefined; do d=build/san_$$(echo $$san | tr , _); mkdir -p $$d; \
	  for f in $(HOSTSRC); do $(HIPCC) $(SANFLAGS) -fsanitize=$$san -c -o $$d/$$f.o $(CSRC)/$$f || exit 1; done; \
	  $(HIPCC) $(SANFLAGS) -fsanitize=$$san -shared -o $$d/libptamd_san.so $$(for f in $(HOSTSRC); do echo $$d/$$f.o; done) $(DEVOBJ) -ldl -lpthread || exit 1; \
	  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=$$san -Iinclude -o $$d/host_stress tests/cpp/host_stress.cpp -L$$d -lptamd_san -Wl,-rpath,$$PWD/$$d -Wl,-rpath,/opt/rocm/lib || exit 1; \
	  python3 -c "import sys; sys.path.insert(0, '.'); from opencl_path_tracer_amd import scenes; print(scenes.write_grid_mesh_obj(200000, '$$d', (40.0, -15.0, 25.0), (2.0, 2.0, 2.0), 10.0, 30.0)[0])" > $$d/obj_path.txt || exit 1; \
	  echo "== -fsanitize=$$san"; ASAN_OPTIONS=detect_leaks=0 $$d/host_stress $$(cat $$d/obj_path.txt) 2>&1 | tee $$d/report.txt; \
	  if grep -q "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $$d/report.txt; then echo "sanitizer findings in $$d/report.txt"; exit 1; fi; done

tests/cpp/dropin: tests/cpp/dropin_main.cpp include/pt_scene.hpp include/pt_api.h $(PKG)/libptamd.so
	g++ -O1 -std=c++14 -Iinclude -o $@ tests/cpp/dropin_main.cpp -L$(PKG) -lptamd -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -rf $(PKG)/libptamd*.so tests/cpp/dropin build
	$(MAKE) -C oracle clean

.PHONY: all oracle clean ab ab-objs check-isa sanitize-host
