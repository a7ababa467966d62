# Maintainer build without Python (python -c "import __graft_entry__ as g; g.build()" does the same, incrementally).
#   make            libokrt.so (HIP, gfx950), libokrt_host.so (C++ host layer), examples/headless_host, examples/headless_host_mgpu
#   make oracle     the CPU checker used by tests/ (and the compiled reference pieces when /root/reference exists)
#   make asan       ASan + UBSan builds of the host layer and the checker, and the CPU test suite run against them
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
PKG      := gpu-raytracer_amd
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -Wall -Wno-class-memaccess

DEV_SRC  := $(wildcard $(PKG)/csrc/*.hip) $(wildcard $(PKG)/csrc/*.cpp)
DEV_OBJ  := $(patsubst $(PKG)/csrc/%,$(PKG)/_build/%.o,$(DEV_SRC))
HOST_SRC := $(wildcard $(PKG)/host/*.cpp)

all: $(PKG)/libokrt.so $(PKG)/libokrt_host.so examples/headless_host examples/headless_host_mgpu

$(PKG)/_build/%.o: $(PKG)/csrc/% $(wildcard $(PKG)/csrc/*.h) include/okrt.h
	@mkdir -p $(PKG)/_build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/libokrt.so: $(DEV_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -ldl

$(PKG)/libokrt_host.so: $(HOST_SRC) $(wildcard $(PKG)/host/*.h) include/okrt.h
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC)

examples/headless_host: examples/headless_host.cpp $(PKG)/libokrt.so $(PKG)/libokrt_host.so
	$(CXX) -O2 -std=c++17 -ffp-contract=off $< -Iinclude -I$(PKG)/host -L$(PKG) -lokrt_host -lokrt -L/opt/rocm/lib -lamdhip64 \
	    '-Wl,-rpath,$$ORIGIN/../$(PKG)' -Wl,-rpath,/opt/rocm/lib -o $@

examples/headless_host_mgpu: examples/headless_host_mgpu.cpp $(PKG)/libokrt.so $(PKG)/libokrt_host.so
	$(CXX) -O2 -std=c++17 -ffp-contract=off $< -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Iinclude -I$(PKG)/host -L$(PKG) -lokrt_host -lokrt \
	    -L/opt/rocm/lib -lamdhip64 '-Wl,-rpath,$$ORIGIN/../$(PKG)' -Wl,-rpath,/opt/rocm/lib -o $@

oracle:
	$(MAKE) -C oracle all
	@if [ -d /root/reference/source ]; then $(MAKE) -C oracle ref; fi

# CPU side under AddressSanitizer + UndefinedBehaviorSanitizer: the C++ host layer (file parsers, PNG encoder,
# BVH / octree builders) and the CPU checker, then the whole CPU test suite against those builds. Nothing here touches a GPU.
SAN := -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
asan:
	$(CXX) $(filter-out -O2,$(CXXFLAGS)) $(SAN) -shared -o $(PKG)/libokrt_host_asan.so $(HOST_SRC)
	$(CXX) -std=c++17 -fPIC -mfma -ffp-contract=off -fno-fast-math -pthread $(SAN) -shared -o oracle/liboracle_asan.so oracle/okrt_oracle.cpp
	LD_PRELOAD="$$($(CXX) -print-file-name=libasan.so) $$($(CXX) -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	    OKRT_HOST_LIBRARY=$(abspath $(PKG)/libokrt_host_asan.so) OKRT_ORACLE_LIBRARY=$(abspath oracle/liboracle_asan.so) \
	    python3 -m pytest tests -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf $(PKG)/_build $(PKG)/libokrt.so $(PKG)/libokrt_host.so examples/headless_host examples/headless_host_mgpu

.PHONY: all oracle asan clean
