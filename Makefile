# Builds the MI355X-native hot path (libtsdf_hip.so, gfx950 only) and the CPU oracle.
# hipcc cross-compiles for gfx950 without a GPU present.
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := $(ROOT)tracking_sdf_amd/csrc
LIBDIR  := $(ROOT)tracking_sdf_amd/lib
LIB     := $(LIBDIR)/libtsdf_hip.so

# -ffp-contract=off and no fast-math are REQUIRED for parity: every multiply-add of the reference is
# two roundings (its g++ build sets no -O/-march flags, src/CMakeLists.txt:97-98).
HIPEXTRA ?=
HIPFLAGS := $(HIPEXTRA) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
            -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wextra -Wno-unused-parameter
KERNELS := $(CSRC)/integrate_kernels.hip $(CSRC)/track_kernels.hip $(CSRC)/volume_kernels.hip $(CSRC)/preproc_kernels.hip $(CSRC)/mesh_kernels.hip
HOSTSRC := $(CSRC)/api_core.cpp $(CSRC)/api_frames.cpp $(CSRC)/api_hotpath.cpp $(CSRC)/api_comm.cpp $(CSRC)/api_volume.cpp \
           $(CSRC)/host_util.cpp $(CSRC)/rccl_dyn.cpp $(CSRC)/aql_queue.cpp
SRCS := $(KERNELS) $(HOSTSRC)
HDRS := $(CSRC)/handle.hpp $(CSRC)/host_util.hpp $(CSRC)/tsdf_device.h $(CSRC)/device_util.h $(CSRC)/mc_tables.h $(CSRC)/host_math.hpp $(CSRC)/rccl_dyn.hpp $(CSRC)/aql_queue.hpp $(ROOT)include/tsdf.h
# The tracker kernels once more as a stand-alone code object: the library's own AQL queue (csrc/aql_queue.cpp, opt-in with
# TSDF_AQL=1) dispatches track_kernel from it for Gauss-Newton passes >= 1.  Same source, same flags, next to the library.
# Both artefacts carry TSDF_BUILD_ID (a hash of the tracker's sources and the flags): AqlQueue::init refuses a code object
# whose id is not the library's, so a stale or foreign .hsaco next to the .so can never run in place of the library's kernel.
TRACK_DEPS := $(CSRC)/track_kernels.hip $(CSRC)/tsdf_device.h $(CSRC)/device_util.h $(CSRC)/aql_queue.hpp
HSACO := $(LIBDIR)/$(patsubst lib%_hip.so,%,$(notdir $(LIB)))_track.hsaco
BUILD_ID := $(shell (cat $(TRACK_DEPS); echo '$(HIPFLAGS)') | sha256sum | cut -c1-32)
IDFLAG := -DTSDF_BUILD_ID='"$(BUILD_ID)"'

.DELETE_ON_ERROR:
all: lib oracle
lib: $(LIB) $(HSACO)

$(LIB): $(SRCS) $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(IDFLAG) -shared -o $@ -x hip $(SRCS) -ldl -pthread -lhsa-runtime64

$(HSACO): $(TRACK_DEPS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(IDFLAG) --genco --no-gpu-bundle-output -o $@ -x hip $(CSRC)/track_kernels.hip

oracle:
	$(MAKE) -C $(ROOT)oracle

# kernel resource usage + ISA for inspection
asm:
	@mkdir -p $(ROOT)build
	for k in integrate_kernels track_kernels; do $(HIPCC) $(HIPFLAGS) -x hip -c $(CSRC)/$$k.hip --cuda-device-only -S -o $(ROOT)build/$$k.s \
	    -Rpass-analysis=kernel-resource-usage 2> $(ROOT)build/$$k.resource_usage.txt || true; done

clean:
	rm -f $(LIB) $(HSACO)
	$(MAKE) -C $(ROOT)oracle clean
.PHONY: all lib oracle asm clean

# C++ example of the drop-in shim (include/sdf_3d_reconstruction/hotpath.hpp): plain g++, links the C ABI only
shim_demo: $(LIB)
	@mkdir -p $(ROOT)build
	g++ -std=c++17 -O2 -Wall -Wextra -o $(ROOT)build/shim_demo $(ROOT)tools/shim_demo.cpp -L$(LIBDIR) -ltsdf_hip -Wl,-rpath,$(LIBDIR)
.PHONY: shim_demo

# C++ offline host for TUM RGB-D directories (depth PNGs -> GPU pre-processing -> track -> integrate); needs zlib
sdf_offline: $(LIB)
	@mkdir -p $(ROOT)build
	g++ -std=c++17 -O2 -Wall -Wextra -o $(ROOT)build/sdf_offline $(ROOT)tools/sdf_offline.cpp -L$(LIBDIR) -ltsdf_hip -lz -Wl,-rpath,$(LIBDIR)
.PHONY: sdf_offline

# measured HBM ceiling for the roofline (streaming RMW over volume-shaped arrays); run on the GPU box
rmw_probe:
	@mkdir -p $(ROOT)build
	$(HIPCC) --offload-arch=gfx950 -O3 -o $(ROOT)build/rmw_probe $(ROOT)tools/rmw_probe.hip
.PHONY: rmw_probe

# known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (tools/pmc_calibrate.py); run on the GPU box
pmc_calibrate:
	@mkdir -p $(ROOT)build
	$(HIPCC) --offload-arch=gfx950 -O3 -o $(ROOT)build/pmc_calibrate $(ROOT)tools/pmc_calibrate.hip
.PHONY: pmc_calibrate

# the reference's exact call signatures (hotpath.hpp -DTSDF_WITH_EIGEN_PCL -DTSDF_WITH_ROS) against the mock Eigen /
# PCL / ROS headers of tests/mock: test infrastructure (tests/test_cpp_shim.py)
refcall_demo: $(LIB)
	@mkdir -p $(ROOT)build
	g++ -std=c++11 -O2 -Wall -Wextra -DTSDF_WITH_EIGEN_PCL -DTSDF_WITH_ROS -I$(ROOT)tests/mock -I$(ROOT)include \
	    -o $(ROOT)build/refcall_demo $(ROOT)tests/mock/refcall_demo.cpp -L$(LIBDIR) -ltsdf_hip -Wl,-rpath,$(LIBDIR)
.PHONY: refcall_demo

# stand-in for librccl over POSIX shared memory (tests/mock_rccl): lets TWO ranks on the one GPU of the test box drive the
# in-library RCCL code path (select it with TSDF_RCCL_LIBRARY=build/libmock_rccl.so); test infrastructure only
mock_rccl:
	@mkdir -p $(ROOT)build
	gcc -O2 -Wall -Wextra -shared -fPIC -I/opt/rocm/include -o $(ROOT)build/libmock_rccl.so $(ROOT)tests/mock_rccl/mock_rccl.c \
	    -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib
.PHONY: mock_rccl

# field writes / constructor constants / cloud-reuse token of the exact-type shim (tests/test_cpp_shim.py, GPU)
shim_fields_demo: $(LIB)
	@mkdir -p $(ROOT)build
	g++ -std=c++11 -O2 -Wall -Wextra -DTSDF_WITH_EIGEN_PCL -DTSDF_WITH_ROS -I$(ROOT)tests/mock -I$(ROOT)include \
	    -o $(ROOT)build/shim_fields_demo $(ROOT)tests/mock/shim_fields_demo.cpp -L$(LIBDIR) -ltsdf_hip -Wl,-rpath,$(LIBDIR)
.PHONY: shim_fields_demo

# hand-written AQL dispatch on a user-mode queue of our own against hipLaunchKernel, for the shape of a tracker pass
# (tools/aql_probe.hip); run on the GPU box: build/aql_probe build/aql_probe_kernel.hsaco
aql_probe:
	@mkdir -p $(ROOT)build
	$(HIPCC) --offload-arch=gfx950 -O2 --genco --no-gpu-bundle-output -DPROBE_DEVICE_ONLY -o $(ROOT)build/aql_probe_kernel.hsaco $(ROOT)tools/aql_probe.hip
	$(HIPCC) --offload-arch=gfx950 -O2 -o $(ROOT)build/aql_probe $(ROOT)tools/aql_probe.hip -lhsa-runtime64
.PHONY: aql_probe
