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
KERN_SRC = compairr_amd/csrc/compairr_hip.hip
KERN_HDR = compairr_amd/csrc/kernels.h compairr_amd/csrc/kernels_sliced.h compairr_amd/csrc/layout.h include/compairr_hip.h

all: lib cli oracle

lib: $(LIB)

$(LIB): $(KERN_SRC) $(KERN_HDR)
	@mkdir -p compairr_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(KERN_SRC)

cli: $(CLI)

$(CLI): $(HOST_SRC) $(HOST_HDR) compairr_amd/host/main/compairr_main.cc
	@mkdir -p bin
	$(CXX) $(CXXFLAGS) -Iinclude -Icompairr_amd/host -o $@ $(HOST_SRC) \
	    compairr_amd/host/main/compairr_main.cc -ldl -lpthread

oracle:
	$(MAKE) -C oracle all
	$(MAKE) tests/bin/compairr_oracle_cli

# host logic + oracle backend: a TEST binary (lives under tests/, never shipped)
tests/bin/compairr_oracle_cli: $(HOST_CORE) $(HOST_HDR) tests/oracle_cli_main.cc oracle/compairr_oracle.c oracle/compairr_oracle.h
	@mkdir -p tests/bin
	gcc -O2 -std=c11 -pthread -c -o tests/bin/compairr_oracle.o oracle/compairr_oracle.c
	$(CXX) $(CXXFLAGS) -Iinclude -Icompairr_amd/host -Ioracle -o $@ $(HOST_CORE) \
	    tests/oracle_cli_main.cc tests/bin/compairr_oracle.o -lpthread

clean:
	rm -rf compairr_amd/lib bin tests/bin
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle clean
