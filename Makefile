# Builds the four command-line programs under the names and in the directories
# the reference's scripts expect (test/diff.sh, test/time.sh):
#
#   make                  -> debug/    (-g -DDEBUG: dumps go to ser/ par/ sergh/ pargh/)
#   make build=timing     -> timing/   (-O3 -DNO_WRITES)
#   make build=release    -> release/  (-O3)
#
#   stereomatch, stereomatch-ghost   CPU: the oracle's CLI (checker side of diff.sh)
#   stereopar,   stereopar-ghost     GPU: C host code over include/stereo_hip.h
#   stereopar-batch                  GPU: a list of pairs over all visible devices
#
# hipcc builds the device library once; everything else is gcc.

build ?= debug
MODE_FLAGS_debug   := -g -DDEBUG
MODE_FLAGS_timing  := -O3 -DNO_WRITES
MODE_FLAGS_release := -O3
ifeq ($(origin MODE_FLAGS_$(build)),undefined)
$(error error: invalid value for build)
endif

O      := $(build)
CC     := gcc
HIPCC  ?= /opt/rocm/bin/hipcc
CFLAGS := -std=gnu11 -Wall -Wextra -pedantic -Wno-unused-parameter -Iinclude -Ioracle $(MODE_FLAGS_$(build))

HOSTDIR := stereomatching_amd/host
CSRC    := stereomatching_amd/csrc
KERNELS := sm_match_bs_ds8 sm_match_bs sm_match_bs_duo8 sm_match_bs_duo sm_match_bs_ds4 sm_api sm_match sm_cost sm_cost_qs sm_cost_pc sm_cost_ssd sm_cost_mfma sm_cost_strip sm_gather
DEVOBJ  := $(addprefix stereomatching_amd/obj/product/,$(addsuffix .o,$(KERNELS)))
# (the same flags, in the same order, as HIPCC_FLAGS of stereomatching_amd/build.py: the two share
#  stereomatching_amd/obj/product and its flags.txt stamp, so neither rebuilds what the other built)
HIPFLAGS_BARE := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result
HIPFLAGS := $(HIPFLAGS_BARE) -Iinclude -I$(CSRC)
DEVSTAMP := stereomatching_amd/obj/product/flags.txt
DEVLIB  := stereomatching_amd/libstereo_hip.so
LINKDEV := -Lstereomatching_amd -lstereo_hip -Wl,-rpath,'$$ORIGIN/../stereomatching_amd'

CPU_PROGRAMS := $(O)/stereomatch $(O)/stereomatch-ghost
GPU_PROGRAMS := $(O)/stereopar $(O)/stereopar-ghost
BATCH_PROGRAM := $(O)/stereopar-batch

all: $(CPU_PROGRAMS) $(GPU_PROGRAMS) $(BATCH_PROGRAM) $(BATCH_PROGRAM)-testhooks

$(O):
	mkdir -p $@

# (the bit-sliced kernel's builds are four translation units: `make -j4` compiles them side by side)
# objects of another flag set are not reused: the stamp changes (and with it every object) only
# when the flags do
$(DEVSTAMP): FORCE
	@mkdir -p $(dir $@)
	@if [ "$$(cat $@ 2>/dev/null)" != "$(HIPFLAGS_BARE)" ]; then rm -f $(dir $@)*.o; echo "$(HIPFLAGS_BARE)" > $@; fi
FORCE:
stereomatching_amd/obj/product/%.o: $(CSRC)/%.hip $(CSRC)/sm_internal.h $(CSRC)/sm_match_bs_kernel.h $(CSRC)/sm_cost.h include/stereo_hip.h $(DEVSTAMP)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(DEVLIB): $(DEVOBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $^ -o $@

$(O)/image.o: $(HOSTDIR)/image.c include/image.h | $(O)
	$(CC) $(CFLAGS) -c $< -o $@
$(O)/image_gpu.o: $(HOSTDIR)/image_gpu.c include/image.h include/stereo_hip.h | $(O)
	$(CC) $(CFLAGS) -c $< -o $@
$(O)/oracle.o: oracle/stereo_oracle.c oracle/stereo_oracle.h | $(O)
	$(CC) $(CFLAGS) -ffp-contract=off -c $< -o $@

# -ghost programs are the same sources with -DGHOST
$(CPU_PROGRAMS): oracle/stereomatch_cli.c $(O)/image.o $(O)/oracle.o
	$(CC) $(CFLAGS) $(if $(findstring ghost,$@),-DGHOST) $^ -o $@ -lm

$(GPU_PROGRAMS): $(HOSTDIR)/stereopar.c $(O)/image.o $(O)/image_gpu.o $(DEVLIB)
	$(CC) $(CFLAGS) $(if $(findstring ghost,$@),-DGHOST) $(filter %.c %.o,$^) -o $@ $(LINKDEV) -lm

# a batch of pairs over all visible GPUs (not in the reference: its programs do one pair)
$(BATCH_PROGRAM): $(HOSTDIR)/stereopar_batch.c $(HOSTDIR)/batch_index.h $(O)/image.o $(DEVLIB)
	$(CC) $(CFLAGS) $(filter %.c %.o,$^) -o $@ $(LINKDEV) -lm -lpthread
# the same program with its failure-injection option (-x): for the tests only, not the product
$(BATCH_PROGRAM)-testhooks: $(HOSTDIR)/stereopar_batch.c $(HOSTDIR)/batch_index.h $(O)/image.o $(DEVLIB)
	$(CC) $(CFLAGS) -DSTEREOPAR_BATCH_TEST_HOOKS $(filter %.c %.o,$^) -o $@ $(LINKDEV) -lm -lpthread

clean:
	rm -rf debug timing release *.ppm

.PHONY: all clean FORCE
