# Top-level build: host library, HIP library (gfx950), oracle (test-only).
#   make host     -> raytracer_amd/libraytracer_host.so   (scene loader, PNG I/O; no GPU)
#   make hip      -> raytracer_amd/librt_hip.so           (HIP kernels + C ABI of include/rt_hip.h)
#   make cli      -> raytracer_amd/rt_amd                 (reference-compatible driver)
#   make oracle   -> oracle/liboracle.so (+ oracle/_ref when /root/reference exists)
CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
HOSTFLAGS = -O2 -std=c++17 -pthread -ffp-contract=off -fPIC -Wall -Wextra
HOST_SRCS = raytracer_amd/host/scene_loader.cpp raytracer_amd/host/png_io.cpp raytracer_amd/host/host_capi.cpp
HOST_HDRS = $(wildcard raytracer_amd/host/*.h) include/rt_hip.h
# -fno-slp-vectorize: the SLP vectoriser packs the scalar float code of the object loop and the shading code into
# v_pk_* instructions on 64-bit register pairs behind v_mov shuffles; without it trace_kernel needs 129 VGPRs
# instead of 155, its dense variant and photon_kernel spill nothing, trace_julia_kernel spills 11 VGPRs instead of 68
# [MI355X: C3 8.95 -> 7.92 ms, C5 render 0.67 -> 0.58, C2 1.75 -> 1.70].  Hand-packed code (the Julia estimator) is inline asm.
HIPFLAGS  = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall

all: host hip cli oracle

host: raytracer_amd/libraytracer_host.so
raytracer_amd/libraytracer_host.so: $(HOST_SRCS) $(HOST_HDRS)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOST_SRCS) -lz

hip: raytracer_amd/librt_hip.so
HIP_SRCS = $(wildcard raytracer_amd/csrc/*.hip)
HIP_OBJS = $(patsubst raytracer_amd/csrc/%.hip,build/%.o,$(HIP_SRCS))
build/%.o: raytracer_amd/csrc/%.hip $(wildcard raytracer_amd/csrc/*.h) include/rt_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
raytracer_amd/librt_hip.so: $(HIP_OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_OBJS)

cli: raytracer_amd/rt_amd
raytracer_amd/rt_amd: $(wildcard raytracer_amd/host/*.cpp) $(HOST_HDRS) raytracer_amd/librt_hip.so
	$(CXX) $(HOSTFLAGS) -o $@ raytracer_amd/host/cli_main.cpp raytracer_amd/host/tracer_shim.cpp \
	    raytracer_amd/host/scene_loader.cpp raytracer_amd/host/png_io.cpp \
	    -Lraytracer_amd -lrt_hip -lz -rdynamic -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

# ---- sanitizer builds of the host code (reference Makefile:10-11 has the flags commented out) ------------
#   make asan      host loader + PNG/APNG + oracle under AddressSanitizer + UBSan: tools/sanitize_host (also run by
#                  tests/test_sanitizers.py), then librt_hip's host code (BVH builder) the same way: tools/sanitize_bvh
#   make tsan      the same host program under ThreadSanitizer (four loader threads sharing the asset cache)
SAN_SRCS = tools/sanitize_host.cpp $(HOST_SRCS) oracle/rt_oracle.cpp
build/asan/sanitize_host: $(SAN_SRCS) $(HOST_HDRS)
	@mkdir -p build/asan
	$(CXX) -O1 -g -std=c++17 -pthread -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o $@ $(SAN_SRCS) -lz
build/tsan/sanitize_host: $(SAN_SRCS) $(HOST_HDRS)
	@mkdir -p build/tsan
	$(CXX) -O1 -g -std=c++17 -pthread -ffp-contract=off -fsanitize=thread -o $@ $(SAN_SRCS) -lz
build/asan/sanitize_bvh: tools/sanitize_bvh.cpp $(HIP_SRCS) $(wildcard raytracer_amd/csrc/*.h) include/rt_hip.h
	@mkdir -p build/asan
	$(HIPCC) $(HIPFLAGS) -g -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -o $@ tools/sanitize_bvh.cpp $(HIP_SRCS)
SCENE_DIRS = scenes/examples scenes/examples2 scenes/examples3
asan-host: build/asan/sanitize_host
	ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 build/asan/sanitize_host $(SCENE_DIRS)
asan-bvh: build/asan/sanitize_bvh
	ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 build/asan/sanitize_bvh
asan: asan-host asan-bvh
tsan: build/tsan/sanitize_host
	build/tsan/sanitize_host $(SCENE_DIRS)

clean:
	rm -f raytracer_amd/*.so raytracer_amd/rt_amd
	rm -rf build
	$(MAKE) -C oracle clean
.PHONY: all host hip cli oracle clean asan asan-host asan-bvh tsan
