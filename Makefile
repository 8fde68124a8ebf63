# Build everything in-tree (the .so files travel to the GPU box with the gpurun snapshot).
#   make hip     -> lt-mapper_amd/libltm_hip.so   (product: C ABI + gfx950 kernels)
#   make oracle  -> oracle/libltm_oracle.so       (test infrastructure: CPU restatement of the reference)
#   make host    -> lt-mapper_amd/host/ltm_run    (C++ mirror of Removerter/Session + CLI)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
PKG    = lt-mapper_amd
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Iinclude -I$(PKG)/csrc -Wall -Wno-unused-result

all: hip oracle host

hip: $(PKG)/libltm_hip.so
# kernels by stage (projection + vote, streaming helpers, voxel grid, kNN), one translation unit each, then the C ABI
KOBJ = $(PKG)/csrc/ltm_k_projection.o $(PKG)/csrc/ltm_k_stream.o $(PKG)/csrc/ltm_k_voxel.o $(PKG)/csrc/ltm_k_knn.o
$(PKG)/csrc/ltm_k_%.o: $(PKG)/csrc/ltm_k_%.hip $(PKG)/csrc/ltm_kernels_common.h $(PKG)/csrc/ltm_kernels.h $(PKG)/csrc/ltm_device_math.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
AOBJ = $(PKG)/csrc/ltm_api_core.o $(PKG)/csrc/ltm_api_vote.o $(PKG)/csrc/ltm_api_voxel.o $(PKG)/csrc/ltm_api_knn.o
$(PKG)/csrc/ltm_api_%.o: $(PKG)/csrc/ltm_api_%.cpp $(PKG)/csrc/ltm_internal.h $(PKG)/csrc/ltm_kernels.h $(PKG)/csrc/ltm_pclsort.h include/ltm.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(PKG)/libltm_hip.so: $(KOBJ) $(AOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

host:
	@if [ -f $(PKG)/host/Makefile ]; then $(MAKE) -C $(PKG)/host; fi

ubench: tools/ubench/valu_rate tools/ubench/mfma_overlap tools/ubench/stream_priority
tools/ubench/valu_rate: tools/ubench/valu_rate.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -w $< -o $@
tools/ubench/stream_priority: tools/ubench/stream_priority.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -w $< -o $@
tools/ubench/mfma_overlap: tools/ubench/mfma_overlap.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -w $< -o $@

clean:
	rm -f $(PKG)/csrc/*.o $(PKG)/libltm_hip.so tools/ubench/valu_rate tools/ubench/mfma_overlap tools/ubench/stream_priority
	$(MAKE) -C oracle clean
.PHONY: all hip oracle host ubench clean
