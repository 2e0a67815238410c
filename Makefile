# Build of the MI355X-native DEXTRACTOR codecs.
#   make lib      dextractor_amd/libdexgpu.so   (HIP kernels + C-ABI, gfx950)
#   make cli      dextractor_amd/bin/{dexta,undexta,dexar,undexar,dexqv,undexqv}
#   make oracle   oracle/libdexref.so (+ oracle/_ref/* when /root/reference is present)  [tests only]
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CSRC     = dextractor_amd/csrc
BUILD    = build
LIB      = dextractor_amd/libdexgpu.so
EXTRA   ?=
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function $(EXTRA)
CFLAGS   = -O2 -fPIC -Iinclude -Wall -Wextra

HIP_SRC  = dx_ctx dx_pack2 dx_qv dx_qv_decode dx_synth dx_index dx_qv_walk
HIP_OBJ  = $(HIP_SRC:%=$(BUILD)/%.o)
C_OBJ    = $(BUILD)/dx_host.o $(BUILD)/dx_files.o $(BUILD)/dx_compat.o
TOOLS    = dexta undexta dexar undexar dexqv undexqv

all: lib cli

lib: $(LIB)

# every device compile also leaves the kernels' register / LDS / scratch use in $(BUILD)/<file>.res (compiler remarks);
# the library target condenses them into dextractor_amd/kernel_resources.txt, which tests/test_host.py checks:
# some kernels must stay under a register count to share a CU with another kernel
$(BUILD)/%.o: $(CSRC)/%.hip $(CSRC)/dx_internal.hpp $(CSRC)/dx_device.hpp $(CSRC)/dx_layout.h $(CSRC)/dx_walk.h $(CSRC)/dx_qv_fast.hpp $(CSRC)/dx_qv_short.hpp include/dexgpu.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(BUILD)/$*.res; rc=$$?; \
	  grep -v "kernel-resource-usage\|^ *[0-9]* | \|^ *| *^" $(BUILD)/$*.res >&2; exit $$rc

$(BUILD)/%.o: $(CSRC)/%.c $(CSRC)/dx_layout.h $(CSRC)/dx_walk.h include/dexgpu.h include/dexcompat.h
	@mkdir -p $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(HIP_OBJ) $(C_OBJ)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libdexgpu.so -lpthread
	@cat $(HIP_SRC:%=$(BUILD)/%.res) | sed -n 's/.*remark: *//p' | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
	  awk '/^Function Name:/ { if (n) print n, v, s, l, o; n = $$3 } /^VGPRs:/ { v = "vgprs=" $$2 } /^ScratchSize/ { s = "scratch=" $$NF } \
	       /^LDS Size/ { l = "lds=" $$NF } /^Occupancy/ { o = "waves_per_simd=" $$NF } END { if (n) print n, v, s, l, o }' \
	  > $(dir $(LIB))kernel_resources$(if $(filter $(LIB),dextractor_amd/libdexgpu.so),,_$(notdir $(basename $(LIB)))).txt

cli: $(LIB) $(TOOLS:%=dextractor_amd/bin/%)

dextractor_amd/bin/%: $(CSRC)/cli/%.c $(CSRC)/cli/cli_common.c $(CSRC)/cli/cli_common.h $(LIB)
	@mkdir -p dextractor_amd/bin
	$(CC) -O2 -Wall -Wextra -Iinclude -I$(CSRC)/cli -o $@ $< $(CSRC)/cli/cli_common.c \
	      -Ldextractor_amd -ldexgpu -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib -lm

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(BUILD) $(LIB) dextractor_amd/bin

.PHONY: all lib cli oracle clean
