// pt_common.h -- launch geometry shared by the render kernels and the stream-compaction library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptk {
constexpr int kBlock = 256;          // threads per workgroup = paths per tile (4 wave64)
constexpr int kWaves = kBlock / 64;
}  // namespace ptk
