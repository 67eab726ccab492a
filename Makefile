# Builds libmapperhip.so (gfx950 only) in-tree and the C pieces of the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mapperatorinator_amd/csrc
OBJDIR := build/obj
LIB := mapperatorinator_amd/lib/libmapperhip.so
SRCS := $(CSRC)/api.hip $(CSRC)/gemm.hip $(CSRC)/norm.hip $(CSRC)/attention.hip $(CSRC)/mel.hip $(CSRC)/conv.hip $(CSRC)/t5.hip $(CSRC)/dit.hip
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.hpp) include/mapperhip.h
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

# profiling build: the same sources with in-kernel phase stamps (tools/decode_phases.py); never loaded by the package
PROF_OBJDIR := build/prof
PROF_LIB := mapperatorinator_amd/lib/libmapperhip_prof.so
PROF_OBJS := $(patsubst $(CSRC)/%.hip,$(PROF_OBJDIR)/%.o,$(SRCS))

$(PROF_OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(PROF_OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DMH_PHASE_STAMPS -c $< -o $@

$(PROF_LIB): $(PROF_OBJS)
	@mkdir -p $(dir $(PROF_LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(PROF_OBJS) -o $@

prof: $(PROF_LIB)

clean:
	rm -rf build $(LIB) $(PROF_LIB)

.PHONY: all clean prof
