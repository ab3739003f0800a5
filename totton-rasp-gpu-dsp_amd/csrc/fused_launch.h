// Launch entry points of the fused kernel, one per transform length, each compiled in a translation unit of its own
// (csrc/fused_size.hip, once per size): as ONE unit the fused kernels are ~4 minutes of hipcc, as eleven parallel units
// about one. engine.hip only dispatches on Geometry::log2k.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "device/common.h"

namespace miups {

#define MI_DECLARE_FUSED(n)                                                                                              \
  bool LaunchFusedK##n(const Geometry &g, const IoDesc &io, const FusedTables &ft, bool narrow, bool r32, unsigned items, \
                       hipStream_t st, std::string *error);
MI_DECLARE_FUSED(5)
MI_DECLARE_FUSED(6)
MI_DECLARE_FUSED(7)
MI_DECLARE_FUSED(8)
MI_DECLARE_FUSED(9)
MI_DECLARE_FUSED(10)
MI_DECLARE_FUSED(11)
MI_DECLARE_FUSED(12)
MI_DECLARE_FUSED(13)
MI_DECLARE_FUSED(14)
#undef MI_DECLARE_FUSED
// block transform length 32768 = two 16384-point transforms through the LDS (fused_split_kernel<14>)
bool LaunchFusedSplitK14(const Geometry &g, const IoDesc &io, const FusedTables &ft, unsigned items, hipStream_t st,
                         std::string *error);

}  // namespace miups
