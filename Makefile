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

$(PKG)/libmipt_hip.so: $(wildcard $(PKG)/csrc/device/*.hip) $(wildcard $(PKG)/csrc/device/*.h) include/mi_pt.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(wildcard $(PKG)/csrc/device/*.hip)

$(PKG)/pbrt_amd: $(PKG)/csrc/host/main.cpp $(PKG)/libmipt_host.so
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(PKG) -lmipt_host -Wl,-rpath,'$$ORIGIN' -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(PKG)/*.so $(PKG)/pbrt_amd oracle/*.so oracle/*.o

.PHONY: all host hip cli oracle clean
