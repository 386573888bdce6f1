// pt_common.h -- launch geometry shared by the render kernels and the stream-compaction library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptk {
#ifndef PT_BLOCK
#define PT_BLOCK 256
#endif
constexpr int kBlock = PT_BLOCK;     // threads per workgroup = paths per tile (4 wave64; PT_BLOCK: tile-size experiments only)
constexpr int kWaves = kBlock / 64;
}  // namespace ptk
