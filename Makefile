# Convenience targets; the Python entry points (mapdn_amd.build, __graft_entry__.build) do the same build.
HIPCC ?= /opt/rocm/bin/hipcc
LIB   := mapdn_amd/libmapdn_hip.so
SRC   := mapdn_amd/csrc/plan.cpp mapdn_amd/csrc/kernels.hip mapdn_amd/csrc/dense.hip mapdn_amd/csrc/sparse.hip mapdn_amd/csrc/policy.hip mapdn_amd/csrc/capi.hip
HDR   := mapdn_amd/csrc/plan.hpp mapdn_amd/csrc/kernels.hpp mapdn_amd/csrc/nr_common.hpp include/mapdn.h

lib: $(LIB)

$(LIB): $(SRC) $(HDR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -o $@.tmp $(SRC)
	mv $@.tmp $@

c_abi_host: examples/c_abi_host.c $(LIB)
	gcc -std=c99 -Wall -Wextra -pedantic -Iinclude $< -o $@ -Lmapdn_amd -lmapdn_hip -Wl,-rpath,$(CURDIR)/mapdn_amd

test-cpu: $(LIB)
	python -m pytest tests -q -m "not gpu"

test-gpu: $(LIB)
	python -m pytest tests -q -m gpu

bench: $(LIB)
	python bench.py

.PHONY: lib test-cpu test-gpu bench
