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

# The SLP vectoriser of this toolchain (ROCm 7.2, clang 22.0.0git) miscompiles the (z, id, q) swap-chain insertion of a
# loop-carried K-best list: it pairs the (q, z) / (id, id) registers and gives a swap taken by the TIE rule (equal depth,
# lower id) the values of the no-swap path -- round 5's "wrong lists on depth ties" build, taken apart in round 6
# (profiles/HISTORY.md; tools/probes/tie_merge.hip reproduces it in 150 lines and is correct with this flag;
# tests/test_ties_gpu.py; tools/structurize_scan.py looks for the defect in the built library).  Every file but siren_x3.hip
# is built without it: all files with selection lists (cfg 3a: 0.957-0.960 ms either way; k_fps_grid loses 922 spilled
# registers), and the other MFMA files, which are as fast or faster without (IDR 8x512: 27.0 -> 25.9-26.2 ms per 1 M
# evaluations; the f32-MFMA SIREN step unchanged); the split-fp16 SIREN step is 0.7 % faster WITH it and holds no list.
SLP_KEPT   := siren_x3
NOSLP      := -fno-slp-vectorize

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/isopoints.h Makefile
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(if $(filter $*,$(SLP_KEPT)),,$(NOSLP)) $(FLAGS_$*) -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
