# Build everything that ships: the gfx950 library, the C++11 host program, and
# (test infrastructure) the oracle.  No cmake, no reference build system.
#
#   make            -> lib + cli + oracle (+ oracle/_ref when /root/reference exists)
#   make lib        -> compairr_amd/lib/libcompairr_hip.so   (hipcc, gfx950)
#   make cli        -> bin/compairr                          (g++ -std=c++11, dlopens the lib)
#   make oracle     -> oracle/liboracle.so, oracle/_ref/compairr, tests/bin/*

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function
CXXFLAGS ?= -O2 -std=c++11 -Wall -Wextra -pedantic

LIB      = compairr_amd/lib/libcompairr_hip.so
CLI      = bin/compairr
HOST_CORE = compairr_amd/host/airr_tsv.cc compairr_amd/host/options.cc compairr_amd/host/overlap_host.cc \
            compairr_amd/host/cluster_host.cc
HOST_SRC = $(HOST_CORE) compairr_amd/host/hip_backend.cc
HOST_HDR = $(wildcard compairr_amd/host/*.h) include/compairr_hip.h
KERN_DIR = compairr_amd/csrc
KERN_HDR = $(KERN_DIR)/kernels.h $(KERN_DIR)/kernels_sliced.h $(KERN_DIR)/kernels_rows.h \
           $(KERN_DIR)/kernels_pairs2.h \
           $(KERN_DIR)/layout.h $(KERN_DIR)/select.h $(KERN_DIR)/context.h include/compairr_hip.h
OBJ_DIR  = compairr_amd/lib/obj
# the probe kernels are instantiated per (kernel variant, waves per workgroup) in
# their own translation units, so that `make -j` compiles them side by side
TU_OBJS  = $(OBJ_DIR)/probe_v0.o $(OBJ_DIR)/resolve.o $(OBJ_DIR)/probe_pairs2.o \
           $(OBJ_DIR)/probe_v1_nw4.o $(OBJ_DIR)/probe_v1_nw8.o $(OBJ_DIR)/probe_v1_nw16.o \
           $(OBJ_DIR)/probe_v2_nw4.o $(OBJ_DIR)/probe_v2_nw8.o $(OBJ_DIR)/probe_v2_nw16.o \
           $(OBJ_DIR)/probe_v2i_nw4.o $(OBJ_DIR)/probe_v2i_nw8.o $(OBJ_DIR)/probe_v2i_nw16.o \
           $(OBJ_DIR)/probe_v2w_nw4.o $(OBJ_DIR)/probe_v2w_nw8.o $(OBJ_DIR)/probe_v2w_nw16.o \
           $(OBJ_DIR)/probe_v2wi_nw4.o $(OBJ_DIR)/probe_v2wi_nw8.o $(OBJ_DIR)/probe_v2wi_nw16.o

all: lib cli oracle

lib:
	$(MAKE) -j8 $(LIB) LIB=$(LIB) OBJ_DIR=$(OBJ_DIR)

$(OBJ_DIR)/main.o: $(KERN_DIR)/compairr_hip.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJ_DIR)/query_layout.o: $(KERN_DIR)/query_layout.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJ_DIR)/ref_index.o: $(KERN_DIR)/ref_index.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJ_DIR)/probe_v0.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=0 -c -o $@ $<

$(OBJ_DIR)/probe_pairs2.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=3 -c -o $@ $<

$(OBJ_DIR)/resolve.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=9 -c -o $@ $<

$(OBJ_DIR)/probe_v1_nw%.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=1 -DTU_NW=$* -c -o $@ $<

$(OBJ_DIR)/probe_v2_nw%.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=2 -DTU_NW=$* -DTU_INLINE=0 -c -o $@ $<

$(OBJ_DIR)/probe_v2i_nw%.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=2 -DTU_NW=$* -DTU_INLINE=1 -c -o $@ $<

$(OBJ_DIR)/probe_v2w_nw%.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=2 -DTU_NW=$* -DTU_INLINE=0 -DTU_WIDE -c -o $@ $<

$(OBJ_DIR)/probe_v2wi_nw%.o: $(KERN_DIR)/probe_tu.hip $(KERN_HDR)
	@mkdir -p $(OBJ_DIR)
	$(HIPCC) $(HIPFLAGS) -DTU_VARIANT=2 -DTU_NW=$* -DTU_INLINE=1 -DTU_WIDE -c -o $@ $<

$(LIB): $(OBJ_DIR)/main.o $(OBJ_DIR)/query_layout.o $(OBJ_DIR)/ref_index.o $(TU_OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $^

# diagnostic build with per-phase cycle counters in the probe kernel (tools/phase_timing.py)
timing:
	$(MAKE) -j8 lib LIB=compairr_amd/lib/libcompairr_hip_timing.so OBJ_DIR=compairr_amd/lib/obj_timing \
	    HIPFLAGS="$(HIPFLAGS) -DCMPR_PHASE_TIMING"

# ablation build: the "debug" tunable skips parts of the kernels (timing experiments only,
# results become wrong; tools/ablation.sh)
ablation:
	$(MAKE) -j8 lib LIB=compairr_amd/lib/libcompairr_hip_ablation.so OBJ_DIR=compairr_amd/lib/obj_ablation \
	    HIPFLAGS="$(HIPFLAGS) -DCMPR_ABLATION"

cli: $(CLI)

$(CLI): $(HOST_SRC) $(HOST_HDR) compairr_amd/host/main/compairr_main.cc
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) -Iinclude -Icompairr_amd/host -o $@ $(HOST_SRC) \
	    compairr_amd/host/main/compairr_main.cc -ldl -lpthread

oracle:
	$(MAKE) -C oracle all
	$(MAKE) tests/bin/compairr_oracle_cli tests/bin/compairr_oracle_cli_asan

asan: tests/bin/compairr_oracle_cli_asan

# the same TEST binary under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only:
# SURVEY section 5; the golden cases run through it in the CPU suite)
SAN = -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1
tests/bin/compairr_oracle_cli_asan: $(HOST_CORE) $(HOST_HDR) tests/oracle_cli_main.cc oracle/compairr_oracle.c oracle/compairr_oracle.h
	@mkdir -p tests/bin
	gcc $(SAN) -std=c11 -pthread -c -o tests/bin/compairr_oracle_asan.o oracle/compairr_oracle.c
	$(CXX) $(SAN) -std=c++11 -Wall -Wextra -pedantic -Iinclude -Icompairr_amd/host -Ioracle -o $@ $(HOST_CORE) \
	    tests/oracle_cli_main.cc tests/bin/compairr_oracle_asan.o -lpthread

# host logic + oracle backend: a TEST binary (lives under tests/, never shipped)
tests/bin/compairr_oracle_cli: $(HOST_CORE) $(HOST_HDR) tests/oracle_cli_main.cc oracle/compairr_oracle.c oracle/compairr_oracle.h
	@mkdir -p tests/bin
	gcc -O2 -std=c11 -pthread -c -o tests/bin/compairr_oracle.o oracle/compairr_oracle.c
	$(CXX) $(CXXFLAGS) -Iinclude -Icompairr_amd/host -Ioracle -o $@ $(HOST_CORE) \
	    tests/oracle_cli_main.cc tests/bin/compairr_oracle.o -lpthread

# the microbenchmarks profiles/rNN/calibration.json and extra/lds_conflicts.txt come from (tools/reprofile.sh,
# tools/r06_lds_conflicts.sh run them on the GPU box; the binaries travel with the snapshot)
TOOL_BINS = tools/bin/calib tools/bin/lds_conflicts tools/bin/atom tools/bin/chase
tools: $(TOOL_BINS)
tools/bin/calib: tools/calib.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
tools/bin/lds_conflicts: tools/lds_conflicts.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
tools/bin/atom: tools/atomics_bench.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
tools/bin/chase: tools/chase_bench.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<

clean:
	rm -rf compairr_amd/lib bin tests/bin
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle asan timing ablation tools clean
