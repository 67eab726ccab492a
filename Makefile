# Builds libmapperhip.so (gfx950 only) in-tree and the C pieces of the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mapperatorinator_amd/csrc
OBJDIR := build/obj
LIB := mapperatorinator_amd/lib/libmapperhip.so
SRCS := $(CSRC)/api.hip $(CSRC)/gemm.hip $(CSRC)/mx8.hip $(CSRC)/norm.hip $(CSRC)/attention.hip $(CSRC)/mel.hip $(CSRC)/conv.hip $(CSRC)/t5.hip $(CSRC)/beam.hip $(CSRC)/dit.hip $(CSRC)/slider.hip
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.hpp) include/mapperhip.h
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs with the wave launch (gfx950 firmware)
# instead of through an s_load from the kernarg segment -- one memory round trip off every small dependent kernel
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14

# attention.hip: hipcc's SLP vectoriser turns `s * c + b * k` of the softmax into v_pk_mul_f32 of assembled register PAIRS plus
# an add of the halves (72 v_mov per key tile to build the pairs: the kernel is VALU-issue-bound) -- off for that file
EXTRA_attention := -fno-slp-vectorize

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

# profiling build: the same sources with in-kernel phase stamps (tools/decode_phases.py); never loaded by the package
PROF_OBJDIR := build/prof
PROF_LIB := mapperatorinator_amd/lib/libmapperhip_prof.so
PROF_OBJS := $(patsubst $(CSRC)/%.hip,$(PROF_OBJDIR)/%.o,$(SRCS))

$(PROF_OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(PROF_OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$*) -DMH_PHASE_STAMPS -c $< -o $@

$(PROF_LIB): $(PROF_OBJS)
	@mkdir -p $(dir $(PROF_LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(PROF_OBJS) -o $@

prof: $(PROF_LIB)

# A/B builds: `make variant NAME=x DEFS="-DMH_..."` -> mapperatorinator_amd/lib/libmapperhip_x.so (select with MAPPERHIP_LIB)
variant:
	@mkdir -p build/var_$(NAME)
	for f in $(SRCS); do b=$$(basename $$f .hip); x=; if [ $$b = attention ]; then x="$(EXTRA_attention)"; fi; $(HIPCC) $(HIPFLAGS) $$x $(DEFS) -c $$f -o build/var_$(NAME)/$$b.o & done; wait
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC build/var_$(NAME)/*.o -o mapperatorinator_amd/lib/libmapperhip_$(NAME).so

clean:
	rm -rf build $(LIB) $(PROF_LIB)

.PHONY: all clean prof variant
