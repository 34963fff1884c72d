# Build the gfx950 shared library (C ABI in include/isopoints.h) and the CPU
# oracle.  `python __graft_entry__.py build` drives this same file.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
HIPFLAGS   ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off \
              -Wall -Wno-unused-function -Iinclude
CSRC       := iso_points_amd/csrc
SRCS       := $(wildcard $(CSRC)/*.hip)
OBJS       := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB        := iso_points_amd/libisopoints_hip.so

all: $(LIB) oracle

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS)

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/isopoints.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
