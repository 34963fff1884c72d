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

# Per-file scheduler strategy (measured, tools/build_variant.sh + tools/siren_eval_bench.py): the SIREN step is one
# long software-pipelined loop body; LLVM's "max-memory-clause" machine-scheduler strategy keeps its loads clustered
# and leaves 25 instead of 39 spilled registers: 2.97 -> 2.83 ms per 1 M evaluations on one box, 3.01 -> 2.92 on another
# (max-ilp 2.88, iterative-ilp 2.93; no effect on idr / idr_x16 / bricks / splat, siren.hip slower).  Results are bit-identical.
FLAGS_siren_x3 := -mllvm -amdgpu-sched-strategy=max-memory-clause

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/isopoints.h Makefile
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
