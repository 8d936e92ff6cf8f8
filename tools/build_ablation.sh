#!/bin/bash
# builds gpurun_out-independent ablation variants of libartamd.so into audio_resampler_amd/_abl/
set -e
cd "$(dirname "$0")/.."
mkdir -p audio_resampler_amd/_abl
for v in ${ABL_VARIANTS:-NOFLUSH NOMFMA NOLOAD BL1}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DABL_$v -I include -I audio_resampler_amd/csrc -c audio_resampler_amd/csrc/sinc_fir.hip -o audio_resampler_amd/_abl/sinc_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o audio_resampler_amd/_abl/libartamd_$v.so audio_resampler_amd/_obj/resampler_host.c.o audio_resampler_amd/_obj/pcm_host.c.o audio_resampler_amd/_obj/extrapolate_host.c.o audio_resampler_amd/_obj/stretch_host.c.o audio_resampler_amd/_obj/device_rt.hip.o audio_resampler_amd/_abl/sinc_$v.o audio_resampler_amd/_obj/pcm_kernels.hip.o audio_resampler_amd/_obj/stretch_kernels.hip.o -lm -lpthread
done
