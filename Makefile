# Plain-make build of the library and the command-line tools (the same commands __graft_entry__.build() runs).
#   make            libecckd_hip.so (hipcc, gfx950) + bin/<tool> (g++, linked against the library only)
#   make oracle     the test-only CPU oracle and, when /root/reference is present, oracle/_ref
#   make check      CPU test suite;  make check-gpu  the parity tests (needs an MI355X)
-include config.mk
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
prefix ?= /usr/local
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function
LIB   := ecckd_amd/libecckd_hip.so
CSRC  := $(sort $(wildcard ecckd_amd/csrc/*.hip) $(wildcard ecckd_amd/csrc/*.cpp))
CHDR  := $(wildcard ecckd_amd/csrc/*.hpp) include/ecckd_hip.h
TOOLS := $(patsubst ecckd_amd/cli/%.cpp,bin/%,$(wildcard ecckd_amd/cli/*.cpp))

all: $(LIB) $(TOOLS) srclinks

# the reference's scripts look for the executables in ../src/ecckd (test/config.h:38-47)
srclinks: $(TOOLS)
	@mkdir -p src/ecckd
	@for t in $(notdir $(TOOLS)); do ln -sf ../../bin/$$t src/ecckd/$$t; done

install: all
	mkdir -p $(prefix)/bin $(prefix)/lib
	cp $(TOOLS) $(prefix)/bin/
	cp $(LIB) $(prefix)/lib/

$(LIB): $(CSRC) $(CHDR)
	$(HIPCC) $(HIPFLAGS) -o $@ $(CSRC)

bin/%: ecckd_amd/cli/%.cpp $(wildcard ecckd_amd/cli/*.hpp) include/ecckd_hip.h $(LIB)
	@mkdir -p bin
	$(CXX) -std=c++17 -O2 -Wall -o $@ $< -Lecckd_amd -lecckd_hip '-Wl,-rpath,$$ORIGIN/../ecckd_amd'

oracle:
	$(MAKE) -s -C oracle

check: all oracle
	python -m pytest tests -q -m "not gpu"

check-gpu: all oracle
	python -m pytest tests -q -m gpu

clean:
	rm -f $(LIB) $(TOOLS)

.PHONY: all oracle check check-gpu clean srclinks install
