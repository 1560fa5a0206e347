# Convenience targets; __graft_entry__.build() performs the same hipcc / oracle builds.
HIPCC ?= /opt/rocm/bin/hipcc
CSRC  := poulpy_amd/csrc
LIB   := poulpy_amd/libpoulpy_hip.so

all: $(LIB) oracle

UNITS := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(CSRC)/_obj/%.o,$(UNITS))
RCCL  := -ldl   # RCCL itself is dlopen-ed at the first pz_comm_* call (api_dist.hip)

# one object per translation unit (make -j8 compiles them in parallel; launch_br.hip is the long pole, ~75 s)
$(CSRC)/_obj/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/poulpy_hip.h
	@mkdir -p $(CSRC)/_obj
	cd $(CSRC) && $(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c $(notdir $<) -o _obj/$(notdir $@)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(OBJS) $(RCCL)

oracle:
	$(MAKE) -s -C oracle

# C++ client of the ABI (tests/cpp/test_abi.cpp); run it on a GPU box
cpp-test: $(LIB) oracle
	g++ -O1 -std=c++17 -I include -I oracle tests/cpp/test_abi.cpp -o tests/cpp/test_abi $(LIB) oracle/_build/libpoulpy_oracle.so \
	    -Wl,-rpath,$(CURDIR)/poulpy_amd -Wl,-rpath,$(CURDIR)/oracle/_build -Wl,-rpath,/opt/rocm/lib

# register / scratch / occupancy table of every kernel
kres:
	python tools/kres.py

test-cpu:
	python -m pytest tests -x -q -m "not gpu"

# on the GPU box (through gpurun from the build container)
test-gpu:
	python -m pytest tests -x -q -m gpu

bench:
	python bench.py

clean:
	rm -f $(LIB) tests/cpp/test_abi
	rm -rf $(CSRC)/_obj
	rm -rf oracle/_build

.PHONY: all oracle cpp-test kres test-cpu test-gpu bench clean
