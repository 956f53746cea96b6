# Build everything in-tree.  `make` = HIP kernel library + host mirror + oracle.
#   raytracer_amd/lib/librt_hip.so    HIP kernels + C ABI (include/rt_hip.h), gfx950
#   raytracer_amd/lib/librt_host.so   C++ mirror of the reference API + flatten (include/rt_host.h)
#   raytracer_amd/bin/rt_exercise     executable front-end: renders an exercise to a PPM on the GPU
#   oracle/librt_oracle.so            CPU oracle (tests / cpu_baseline only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

# No FMA contraction, no fast-math anywhere: decisions must be bit-identical
# between the oracle and the kernels.
FPFLAGS = -ffp-contract=off -fno-fast-math
CXXFLAGS = -std=c++17 -O2 -fPIC -Wall -Wextra $(FPFLAGS)
HIPFLAGS = -std=c++17 -O3 -fPIC --offload-arch=$(ARCH) $(FPFLAGS) -Wall -Wno-unused-function

LIBDIR = raytracer_amd/lib
BINDIR = raytracer_amd/bin
HOST_SRC = $(wildcard raytracer_amd/host/*.cpp)
HOST_HDR = $(wildcard raytracer_amd/host/*.h) include/rt_hip.h include/rt_host.h
HIP_SRC = raytracer_amd/csrc/rt_hip.hip
HIP_HDR = $(wildcard raytracer_amd/csrc/*.h) include/rt_hip.h

all: hip host oracle exercise

hip: $(LIBDIR)/librt_hip.so
host: $(LIBDIR)/librt_host.so
oracle: oracle/librt_oracle.so
exercise: $(BINDIR)/rt_exercise

$(LIBDIR)/librt_hip.so: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC)

$(LIBDIR)/librt_host.so: $(HOST_SRC) $(HOST_HDR) $(LIBDIR)/librt_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -L$(LIBDIR) -lrt_hip -Wl,-rpath,'$$ORIGIN'

$(BINDIR)/rt_exercise: raytracer_amd/exercises/main.cpp $(LIBDIR)/librt_host.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lrt_host -lrt_hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle/librt_oracle.so: oracle/rt_oracle.cpp
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) $(BINDIR) oracle/librt_oracle.so

.PHONY: all hip host oracle exercise clean
