# Builds libpacingpseudo_hip.so (gfx950 only) and nothing else.  `python -c "import __graft_entry__ as g; g.build()"`
# drives this Makefile.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := pacingpseudo_amd/csrc
OUT := pacingpseudo_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wall -Wno-unused-function $(EXTRA)
SRCS := $(CSRC)/pp_conv.hip $(CSRC)/pp_wino.hip $(CSRC)/pp_norm.hip $(CSRC)/pp_spatial.hip $(CSRC)/pp_loss.hip $(CSRC)/pp_optim.hip $(CSRC)/pp_augment.hip
OBJS := $(patsubst $(CSRC)/%.hip,$(OUT)/%.o,$(SRCS)) $(OUT)/pp_runtime.o

all: $(OUT)/libpacingpseudo_hip.so

$(OUT)/%.o: $(CSRC)/%.hip $(CSRC)/pp_common.h include/pacingpseudo_hip.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT)/pp_runtime.o: $(CSRC)/pp_runtime.cpp $(CSRC)/pp_common.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT)/libpacingpseudo_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# study build for tests/studies/halo_phase_trace.py: the halo convolution kernels instrumented with s_memtime per phase
# (a second library beside the product one; select it with PP_LIB_PATH)
trace: all
	@mkdir -p $(OUT)/trace
	$(HIPCC) $(HIPFLAGS) -DPP_HALO_TRACE -c $(CSRC)/pp_conv.hip -o $(OUT)/trace/pp_conv.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(OUT)/trace/libpacingpseudo_hip.so $(OUT)/trace/pp_conv.o $(filter-out $(OUT)/pp_conv.o,$(OBJS))

clean:
	rm -rf $(OUT)

.PHONY: all clean trace asan

# AddressSanitizer build of the HOST side (pp_runtime.cpp + the argument checks of every launch wrapper), no GPU needed:
# the HOST half of every source is instrumented (-Xarch_host: the device half is compiled as usual, GPU ASAN is not available on
# this pool) and linked with tests/native/asan_args.cpp, which feeds invalid arguments to a representative entry point of
# every source file.  `make asan` builds and runs it (tests/test_abi.py does the same).
ASAN_OUT := $(OUT)/asan
ASANFLAGS := --offload-arch=$(ARCH) -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -O1 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wno-unused-function
ASAN_OBJS := $(patsubst $(CSRC)/%.hip,$(ASAN_OUT)/%.o,$(SRCS)) $(ASAN_OUT)/pp_runtime.o

$(ASAN_OUT)/%.o: $(CSRC)/%.hip $(CSRC)/pp_common.h include/pacingpseudo_hip.h
	@mkdir -p $(ASAN_OUT)
	$(HIPCC) $(ASANFLAGS) -c $< -o $@

$(ASAN_OUT)/pp_runtime.o: $(CSRC)/pp_runtime.cpp $(CSRC)/pp_common.h
	@mkdir -p $(ASAN_OUT)
	$(HIPCC) $(ASANFLAGS) -x hip -c $< -o $@

$(ASAN_OUT)/asan_args.o: tests/native/asan_args.cpp include/pacingpseudo_hip.h
	@mkdir -p $(ASAN_OUT)
	/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address -fno-omit-frame-pointer -O1 -g -std=c++17 -Iinclude -c $< -o $@

$(ASAN_OUT)/asan_args: $(ASAN_OUT)/asan_args.o $(ASAN_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fsanitize=address $(ASAN_OUT)/asan_args.o $(ASAN_OBJS) -o $@ -ldl

asan: $(ASAN_OUT)/asan_args
	ASAN_OPTIONS=detect_leaks=0 $(ASAN_OUT)/asan_args
