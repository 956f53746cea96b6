# Build everything in-tree.  `make -j8` = HIP kernel library + host mirror + oracle.
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
HIPFLAGS = -std=c++17 -O3 -fPIC --offload-arch=$(ARCH) $(FPFLAGS) -Wall -Wno-unused-function $(EXTRA_HIPFLAGS)

LIBDIR = raytracer_amd/lib
BINDIR = raytracer_amd/bin
OBJDIR = build/obj
CSRC = raytracer_amd/csrc
HOST_SRC = $(wildcard raytracer_amd/host/*.cpp)
HOST_HDR = $(wildcard raytracer_amd/host/*.h) include/rt_hip.h include/rt_host.h
HIP_HDR = $(wildcard $(CSRC)/*.h) include/rt_hip.h

# The tracing kernels are explicit instantiations of the templates in rt_kernels.h, one object per
# scene class (0 general, 1 flat, 2 nested leaves) x counting variant x part (0 render_kernel, 1 wf_level_kernel,
# 2 batch kernels), so that the ~50 large kernels compile in parallel.
INST_OBJS = $(foreach c,0 1 2,$(foreach s,false true,$(foreach p,0 1,$(OBJDIR)/inst_c$(c)_$(s)_p$(p).o))) \
            $(foreach c,0 1 2,$(OBJDIR)/inst_c$(c)_false_p2.o)
HIP_OBJS = $(OBJDIR)/rt_hip.o $(OBJDIR)/rt_wf2.o $(OBJDIR)/rt_dirindex.o $(OBJDIR)/rt_hostmem.o $(OBJDIR)/rt_multi.o $(INST_OBJS)

all: hip host oracle exercise

hip: $(LIBDIR)/librt_hip.so
host: $(LIBDIR)/librt_host.so
oracle: oracle/librt_oracle.so
exercise: $(BINDIR)/rt_exercise

$(OBJDIR)/inst_c%.o: $(CSRC)/rt_inst.hip $(HIP_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DRT_INST_CLS=$(word 1,$(subst _, ,$*)) -DRT_INST_ST=$(word 2,$(subst _, ,$*)) \
	    -DRT_INST_PART=$(subst p,,$(word 3,$(subst _, ,$*))) -c -o $@ $<

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIBDIR)/librt_hip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(HIP_OBJS) -ldl

$(LIBDIR)/librt_host.so: $(HOST_SRC) $(HOST_HDR) $(LIBDIR)/librt_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -L$(LIBDIR) -lrt_hip -Wl,-rpath,'$$ORIGIN'

$(BINDIR)/rt_exercise: raytracer_amd/exercises/main.cpp $(LIBDIR)/librt_host.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lrt_host -lrt_hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle/librt_oracle.so: oracle/rt_oracle.cpp
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) $(BINDIR) $(OBJDIR) oracle/librt_oracle.so

.PHONY: all hip host oracle exercise clean
