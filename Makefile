# Convenience targets; the Python entry points (mapdn_amd.build, __graft_entry__.build) do the same build.
# `make -j8 lib` compiles the translation units in parallel (k_nr_tree's instantiations are four objects from one source).
HIPCC ?= /opt/rocm/bin/hipcc
FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result
LIB   := mapdn_amd/libmapdn_hip.so
CSRC  := mapdn_amd/csrc
SRC   := plan.cpp kernels.hip dense.hip sparse.hip policy.hip policy_bwd.hip critic.hip rollout.hip capi.hip
HDR   := $(CSRC)/plan.hpp $(CSRC)/colstats.hpp $(CSRC)/rowtile.hpp $(CSRC)/kernels.hpp $(CSRC)/philox.hpp $(CSRC)/nrmath.hpp $(CSRC)/nr_common.hpp $(CSRC)/nr_tree.hpp $(CSRC)/nr_inst_list.hpp include/mapdn.h
OBJ   := $(addprefix build/,$(addsuffix .o,$(basename $(SRC)))) build/nr_inst_0.o build/nr_inst_1.o build/nr_inst_2.o build/nr_inst_3.o

lib: $(LIB)

build/%.o: $(CSRC)/%.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@
build/%.o: $(CSRC)/%.cpp $(HDR)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@
build/nr_inst_%.o: $(CSRC)/nr_inst.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -DNR_INST_PART=$* -c $< -o $@

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@.tmp $(OBJ)
	mv $@.tmp $@

c_abi_host: examples/c_abi_host.c $(LIB)
	gcc -std=c99 -Wall -Wextra -pedantic -Iinclude $< -o $@ -Lmapdn_amd -lmapdn_hip -Wl,-rpath,$(CURDIR)/mapdn_amd

test-cpu: $(LIB)
	python -m pytest tests -q -m "not gpu"

test-gpu: $(LIB)
	python -m pytest tests -q -m gpu

bench: $(LIB)
	python bench.py

# SURVEY section 5 "sanitizers" row: the HOST C++ of the library (plan.cpp's index arithmetic, the C ABI in capi.hip) under
# AddressSanitizer + UndefinedBehaviorSanitizer (device code is not instrumented: -fno-gpu-sanitize; GPU ASan is not available on
# this pool).  Every CPU test that reaches the plan through mapdn_create(..., device = -1) — the 24 random trees and forests of
# tests/test_topology_stress.py, the meshed nets of tests/test_general_topology.py, bus fusion, ingestion, the C-ABI checks — runs
# against build/libmapdn_hip_asan.so with the sanitizer runtime preloaded into python; any report is fatal.
ASAN_RT := $(shell /opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
asan:
	@mkdir -p build
	MAPDN_BUILD_OUT=$(CURDIR)/build/libmapdn_hip_asan.so \
	MAPDN_EXTRA_FLAGS="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan" python -m mapdn_amd.build --force
	LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	MAPDN_LIB_PATH=$(CURDIR)/build/libmapdn_hip_asan.so python -m pytest tests/test_topology_stress.py tests/test_general_topology.py \
	    tests/test_capi_cpu.py tests/test_bus_fusion.py tests/test_data_ingestion.py tests/test_data_formats.py -q -m "not gpu" -p no:cacheprovider

.PHONY: lib test-cpu test-gpu bench asan
