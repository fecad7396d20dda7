# Builds the product (HIP library + viewer) and the test-only oracle.
#   make            -> esctp1raytracer_amd/lib/libesctp1rt.so, bin/ESCViewer2021, oracle/
#   make lib        -> the C-ABI shared library only (gfx950 code object inside)
# hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off is part of the arithmetic
# contract (see rt_kernels.hip); do not remove it.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := esctp1raytracer_amd
LIBDIR   := $(PKG)/lib
LIB      := $(LIBDIR)/libesctp1rt.so
VIEWER   := bin/ESCViewer2021

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off \
            -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize \
            -Iinclude -I$(PKG)/host -I$(PKG)/csrc -Wall -Wno-unused-function

LIB_SRC  := $(PKG)/csrc/rt_kernels.hip $(PKG)/csrc/rt_capi.cpp $(PKG)/csrc/rt_multi.cpp \
            $(PKG)/host/host_core.cpp $(PKG)/host/obj_loader.cpp $(PKG)/host/synth.cpp \
            $(PKG)/host/accel_build.cpp
LIB_HDR  := include/esctp1_rt.h $(PKG)/csrc/rt_device.h $(PKG)/csrc/rt_math.h $(PKG)/csrc/rt_brute.h $(PKG)/csrc/rt_accel.h $(PKG)/csrc/rt_lists.h $(PKG)/csrc/rt_tile_math.h $(PKG)/host/scene.h $(PKG)/host/accel_build.h

all: lib viewer oracle

lib: $(LIB)

$(LIB): $(LIB_SRC) $(LIB_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(LIB_SRC) -ldl

viewer: $(VIEWER)

$(VIEWER): $(PKG)/host/viewer_main.cpp include/esctp1_rt.h $(LIB)
	@mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -ffp-contract=off -Iinclude -o $@ $(PKG)/host/viewer_main.cpp \
	    -L$(LIBDIR) -lesctp1rt -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

oracle:
	$(MAKE) -C oracle

asm: # keep the ISA for inspection
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/rt_kernels.s $(PKG)/csrc/rt_kernels.hip \
	    -Rpass-analysis=kernel-resource-usage

clean:
	rm -rf $(LIBDIR) bin build
	$(MAKE) -C oracle clean

.PHONY: all lib viewer oracle asm clean
