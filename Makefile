# Top-level build: host library (g++), HIP library (hipcc, gfx950 only), CLI, oracle.
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $(EXTRA_HIPFLAGS)
HOSTFLAGS = -std=c++17 -O2 -ffp-contract=off -fopenmp -fPIC -Wall

LIBDIR = sigmap_amd/lib
CSRC = sigmap_amd/csrc
HOST = sigmap_amd/host
HIP_SRCS = $(CSRC)/api.hip $(CSRC)/signal.hip $(CSRC)/events.hip $(CSRC)/seed_lookup.hip $(CSRC)/sort.hip $(CSRC)/chain.hip $(CSRC)/sdtw.hip $(CSRC)/cwt.hip
HIP_OBJS = $(HIP_SRCS:$(CSRC)/%.hip=build/%.o)
HOST_SRCS = $(HOST)/kdtree.cc $(HOST)/pore_model.cc $(HOST)/sequence_batch.cc $(HOST)/signal_batch.cc \
            $(HOST)/fast5_reader.cc $(HOST)/blow5_reader.cc $(HOST)/spatial_index.cc $(HOST)/mapping_logic.cc $(HOST)/device_methods.cc \
            $(HOST)/host_abi.cc
# FAST5 input needs HDF5 (headers + libhdf5); this image ships it under /opt/conda.  Without it the
# library still builds and a .fast5 file is a loud run-time error.
HDF5_ROOT ?= /opt/conda
ifneq ($(wildcard $(HDF5_ROOT)/include/hdf5.h),)
ifneq ($(wildcard $(HDF5_ROOT)/lib/libhdf5.so),)
HDF5_FLAGS = -DSMHOST_HAVE_HDF5 -isystem $(HDF5_ROOT)/include
HDF5_LIBS = -L$(HDF5_ROOT)/lib -lhdf5 -Wl,-rpath,$(HDF5_ROOT)/lib
endif
endif

# zstd-compressed BLOW5 records: optional, like HDF5
ZSTD_ROOT ?= /opt/conda
ifneq ($(wildcard $(ZSTD_ROOT)/include/zstd.h),)
ifneq ($(wildcard $(ZSTD_ROOT)/lib/libzstd.so),)
ZSTD_FLAGS = -DSMHOST_HAVE_ZSTD -isystem $(ZSTD_ROOT)/include
ZSTD_LIBS = -L$(ZSTD_ROOT)/lib -lzstd -Wl,-rpath,$(ZSTD_ROOT)/lib
endif
endif

all: $(LIBDIR)/libsigmap_host.so $(LIBDIR)/libsigmap_hip.so cli oracle

$(LIBDIR)/libsigmap_host.so: $(HOST_SRCS) $(wildcard $(HOST)/*.h) include/sigmap_host.h include/sigmap_hip.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) $(HDF5_FLAGS) $(ZSTD_FLAGS) -shared -o $@ $(HOST_SRCS) -lz -ldl $(HDF5_LIBS) $(ZSTD_LIBS)

build/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/knobs.h include/sigmap_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# sort.hip prefetches with inline-assembly loads the compiler does not track: the device assembly of the same
# flags is checked for any use of a load's destination registers before the matching wait (tools/check_asm_prefetch.py)
build/sort.asmcheck: $(CSRC)/sort.hip $(CSRC)/common.h $(CSRC)/knobs.h include/sigmap_hip.h tools/check_asm_prefetch.py
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S $(CSRC)/sort.hip -o build/sort.device.s
	python3 tools/check_asm_prefetch.py build/sort.device.s
	@touch $@

$(LIBDIR)/libsigmap_hip.so: $(HIP_OBJS) build/sort.asmcheck
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS) -ldl

cli: $(LIBDIR)/libsigmap_host.so $(LIBDIR)/libsigmap_hip.so
	@mkdir -p sigmap_amd/bin
	$(HIPCC) -std=c++17 -O2 -ffp-contract=off -fopenmp -o sigmap_amd/bin/sigmap $(HOST)/sigmap.cc \
	   -L$(LIBDIR) -lsigmap_host -lsigmap_hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIBDIR)/*.so sigmap_amd/bin
	$(MAKE) -C oracle clean

.PHONY: all oracle clean cli
