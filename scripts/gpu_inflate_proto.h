// gpu_inflate_proto.h -- the stand-alone measurement build of the GPU-side DEFLATE kernel (kindel_amd/csrc/kd_gpu_inflate.h):
// scripts/gpu_inflate_proto.hip / .py time k_gpu_inflate alone against the host decoder, tests/emu/gpu_inflate_emu.cpp checks it
// against zlib on the CPU emulator.
#pragma once
#include "../kindel_amd/csrc/kd_gpu_inflate.h"
