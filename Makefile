# Top-level build, same interface as the reference's Makefile
# (/root/reference/Makefile): `make` (debug), `make build=timing`,
# `make build=release` produce, in debug/ timing/ release/, the four programs
#   stereomatch  stereomatch-ghost      CPU (the oracle's CLI: the checker side
#                                       of test/diff.sh, built from oracle/)
#   stereopar    stereopar-ghost        GPU: C host code over the C ABI of
#                                       include/stereo_hip.h (hipcc replaces nvcc)
# so that the reference's test/diff.sh and test/time.sh run unchanged.

name           := stereomatch
name_par       := stereopar
name_ghost     := stereomatch-ghost
name_ghost_par := stereopar-ghost

build  := debug
CC     := gcc
HIPCC  ?= /opt/rocm/bin/hipcc
CFLAGS := -Wall -Wextra -pedantic -std=gnu11 -Wno-unused-parameter -Iinclude -Ioracle
LDLIBS := -lm

ifeq ($(build),debug)
    outdir := debug
    CFLAGS += -g -DDEBUG
else ifeq ($(build),timing)
    outdir := timing
    CFLAGS += -O3 -DNO_WRITES
else ifeq ($(build),release)
    outdir := release
    CFLAGS += -O3
else
    $(error error: invalid value for build)
endif

HOST    := stereomatching_amd/host
HIPLIB  := stereomatching_amd/libstereo_hip.so
HIPSRC  := stereomatching_amd/csrc/sm_api.hip stereomatching_amd/csrc/sm_match.hip stereomatching_amd/csrc/sm_match_bs.hip stereomatching_amd/csrc/sm_cost.hip
HIPDEP  := $(HIPSRC) stereomatching_amd/csrc/sm_internal.h include/stereo_hip.h
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -Istereomatching_amd/csrc
# the programs find the library next to the package wherever the tree is copied
RPATH   := -Wl,-rpath,'$$ORIGIN/../stereomatching_amd'

all: $(outdir) $(outdir)/$(name) $(outdir)/$(name_ghost) $(outdir)/$(name_par) $(outdir)/$(name_ghost_par)

$(outdir):
	mkdir -p $(outdir)

$(HIPLIB): $(HIPDEP)
	$(HIPCC) $(HIPFLAGS) $(HIPSRC) -o $@

$(outdir)/image.o: $(HOST)/image.c include/image.h
	$(CC) $(CFLAGS) -c $< -o $@

$(outdir)/image_gpu.o: $(HOST)/image_gpu.c include/image.h include/stereo_hip.h
	$(CC) $(CFLAGS) -c $< -o $@

$(outdir)/oracle.o: oracle/stereo_oracle.c oracle/stereo_oracle.h
	$(CC) $(CFLAGS) -ffp-contract=off -c $< -o $@

$(outdir)/$(name): oracle/stereomatch_cli.c $(outdir)/image.o $(outdir)/oracle.o
	$(CC) $(CFLAGS) $^ -o $@ $(LDLIBS)

$(outdir)/$(name_ghost): oracle/stereomatch_cli.c $(outdir)/image.o $(outdir)/oracle.o
	$(CC) $(CFLAGS) -DGHOST $^ -o $@ $(LDLIBS)

$(outdir)/$(name_par): $(HOST)/stereopar.c $(outdir)/image.o $(outdir)/image_gpu.o $(HIPLIB)
	$(CC) $(CFLAGS) $(HOST)/stereopar.c $(outdir)/image.o $(outdir)/image_gpu.o -o $@ -Lstereomatching_amd -lstereo_hip $(RPATH) $(LDLIBS)

$(outdir)/$(name_ghost_par): $(HOST)/stereopar.c $(outdir)/image.o $(outdir)/image_gpu.o $(HIPLIB)
	$(CC) $(CFLAGS) -DGHOST $(HOST)/stereopar.c $(outdir)/image.o $(outdir)/image_gpu.o -o $@ -Lstereomatching_amd -lstereo_hip $(RPATH) $(LDLIBS)

clean:
	-rm -rf *.ppm debug timing release

.PHONY: all clean
