# Build recipe for the product libraries (host front end + HIP path) and the CLI.
# `python -c "import __graft_entry__ as g; g.build()"` drives this.
PKG      := pbrt-v3-spectral_amd
HOSTSRC  := $(wildcard $(PKG)/csrc/host/*.cpp)
HOSTSRC  := $(filter-out $(PKG)/csrc/host/main.cpp,$(HOSTSRC))
CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
CXXFLAGS := -std=c++17 -O2 -fPIC -pthread -ffp-contract=off -Wall -Wno-unused-function -Iinclude
HIPFLAGS := --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Iinclude -Wno-unused-result -Wno-unused-value

all: host hip cli oracle

host: $(PKG)/libmipt_host.so
hip: $(PKG)/libmipt_hip.so
cli: $(PKG)/pbrt_amd

$(PKG)/libmipt_host.so: $(HOSTSRC) $(wildcard $(PKG)/csrc/host/*.h) $(wildcard $(PKG)/csrc/host/*.inc) include/mi_pt.h include/mi_scene.h
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOSTSRC) -ldl -lz

# pt_kernels.hip is compiled four times side by side (MIPT_PART: the other kernels + host code, and the k_shade instances in
# three groups), hlbvh.hip once; `make -j` runs them together.
DEVDEPS  := $(wildcard $(PKG)/csrc/device/*.hip) $(wildcard $(PKG)/csrc/device/*.h) include/mi_pt.h
DEVOBJ   := build/pt_part0.o build/pt_part1.o build/pt_part2.o build/pt_part3.o build/hlbvh.o
build/pt_part%.o: $(DEVDEPS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DMIPT_PART=$* -c -o $@ $(PKG)/csrc/device/pt_kernels.hip
build/hlbvh.o: $(DEVDEPS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(PKG)/csrc/device/hlbvh.hip
$(PKG)/libmipt_hip.so: $(DEVOBJ)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(DEVOBJ)

$(PKG)/pbrt_amd: $(PKG)/csrc/host/main.cpp $(PKG)/libmipt_host.so
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(PKG) -lmipt_host -Wl,-rpath,'$$ORIGIN' -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build; rm -f $(PKG)/*.so $(PKG)/pbrt_amd oracle/*.so oracle/*.o

.PHONY: all host hip cli oracle clean
