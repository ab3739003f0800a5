// Template bodies behind fused_launch.h: which fused_kernel instantiation a call takes, and its launch.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "device/kernel_fused.h"
#include "hip_check.h"

namespace miups {

template <int LOG2K, bool EXT, int W, bool R32 = false>
bool LaunchFusedVariant(const Geometry &g, const IoDesc &io, const FusedTables &ft, unsigned items, hipStream_t st,
                        std::string *error) {
  using Cfg = FusedCfg<LOG2K, W, R32>;
  static bool attr_set[64] = {};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (Cfg::LDS_BYTES > 64 * 1024 && dev < 64 && !attr_set[dev]) {
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_kernel<LOG2K, EXT, W, R32>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((fused_kernel<LOG2K, EXT, W, R32>), dim3(items), dim3(Cfg::T), Cfg::LDS_BYTES, st, g, io, ft);
  return HipOk(hipGetLastError(), "fused_kernel launch", error);
}

// small calls: io.phase_parts workgroups per work item (fused_parts_kernel), sizes 2^10 .. 2^14
template <int LOG2K>
bool LaunchFusedParts(const Geometry &g, const IoDesc &io, const FusedTables &ft, unsigned items, hipStream_t st,
                      std::string *error) {
  using Cfg = FusedCfg<LOG2K, 2>;
  static bool attr_set[64] = {};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (Cfg::LDS_BYTES > 64 * 1024 && dev < 64 && !attr_set[dev]) {
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_parts_kernel<LOG2K>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((fused_parts_kernel<LOG2K>), dim3(items), dim3(Cfg::T), Cfg::LDS_BYTES, st, g, io, ft);
  return HipOk(hipGetLastError(), "fused_parts_kernel launch", error);
}

// narrow = the tables are in the narrow layout (K >= 1024): one butterfly per thread; r32 = the tables follow the
// radix-32 pass plan (wide form, K = 8192 / 16384)
template <int LOG2K>
bool LaunchFused(const Geometry &g, const IoDesc &io, const FusedTables &ft, bool narrow, bool r32, unsigned items,
                 hipStream_t st, std::string *error) {
  if constexpr (LOG2K >= 10) {
#if !defined(MIUPS_NO_NARROW)
    if (narrow) {
      return io.ext_epilogue ? LaunchFusedVariant<LOG2K, true, 1>(g, io, ft, items, st, error)
                             : LaunchFusedVariant<LOG2K, false, 1>(g, io, ft, items, st, error);
    }
#endif
  }
  if (narrow) {
    if (error) {
      *error = "narrow tables without a narrow kernel";
    }
    return false;
  }
  if (io.phase_parts > 1) {  // `items` counts workgroups: work items * phase_parts
    if constexpr (LOG2K >= kPartsMinLog2K) {
      if (!r32 && io.ext_epilogue && g.P % io.phase_parts == 0) {
        return LaunchFusedParts<LOG2K>(g, io, ft, items, st, error);
      }
    }
    if (error) {
      *error = "phase-split launch asked of a kernel that has no such form";
    }
    return false;
  }
#if defined(MIUPS_WITH_R32)  // experiment builds only (scripts/build_variant.sh NAME -DMIUPS_WITH_R32): the radix-32 plan
  if constexpr (fused_plan_r32_exists(LOG2K, 2)) {
    if (r32) {
      return io.ext_epilogue ? LaunchFusedVariant<LOG2K, true, 2, true>(g, io, ft, items, st, error)
                             : LaunchFusedVariant<LOG2K, false, 2, true>(g, io, ft, items, st, error);
    }
  }
#endif
  if (r32) {
    if (error) {
      *error = "radix-32 tables (MIUPS_EXP_R32) but this build holds no radix-32 kernel (-DMIUPS_WITH_R32)";
    }
    return false;
  }
  return io.ext_epilogue ? LaunchFusedVariant<LOG2K, true, 2>(g, io, ft, items, st, error)
                         : LaunchFusedVariant<LOG2K, false, 2>(g, io, ft, items, st, error);
}

template <int LOG2K>
bool LaunchFusedSplit(const Geometry &g, const IoDesc &io, const FusedTables &ft, unsigned items, hipStream_t st,
                      std::string *error) {
  using Cfg = FusedCfg<LOG2K>;
  static bool attr_set[64] = {};
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  if (Cfg::LDS_BYTES_SPLIT > 64 * 1024 && dev < 64 && !attr_set[dev]) {
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_split_kernel<LOG2K>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES_SPLIT));
    attr_set[dev] = true;
  }
  if (io.phase_parts > 1) {  // small calls: `items` counts workgroups = work items * phase_parts (a divisor of 2P)
    static bool attr_parts[64] = {};
    if (Cfg::LDS_BYTES_SPLIT > 64 * 1024 && dev < 64 && !attr_parts[dev]) {
      MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_split_parts_kernel<LOG2K>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES_SPLIT));
      attr_parts[dev] = true;
    }
    hipLaunchKernelGGL((fused_split_parts_kernel<LOG2K>), dim3(items), dim3(Cfg::T), Cfg::LDS_BYTES_SPLIT, st, g, io, ft);
    return HipOk(hipGetLastError(), "fused_split_parts_kernel launch", error);
  }
  hipLaunchKernelGGL((fused_split_kernel<LOG2K>), dim3(items), dim3(Cfg::T), Cfg::LDS_BYTES_SPLIT, st, g, io, ft);
  return HipOk(hipGetLastError(), "fused_split_kernel launch", error);
}

}  // namespace miups
