// TEST INFRASTRUCTURE ONLY -- shared by the translation units of the thread-emulation driver (emu_driver.cpp: shim, staged /
// two-level / frame kernels, main; emu_fused.cpp: the fused kernels, one transform length per compile so that the build
// runs in parallel -- as one unit it took four minutes of g++ under the sanitizers).
#pragma once

#include <cstddef>
#include <functional>

#include "device/common.h"
#include "host/spectrum.h"

namespace miups_emu {
// run f() once per (block, thread); with barriers every thread of a block is a real OS thread
void launch(unsigned grid, unsigned block, size_t shmem, bool barriers, const std::function<void()> &f);
}  // namespace miups_emu

#define MI_EMU_DECLARE_FUSED(n) \
  void EmuFusedK##n(const miups::Geometry &g, const miups::IoDesc &io, const miups::FilterTables &t, unsigned items);
MI_EMU_DECLARE_FUSED(5)
MI_EMU_DECLARE_FUSED(6)
MI_EMU_DECLARE_FUSED(7)
MI_EMU_DECLARE_FUSED(8)
MI_EMU_DECLARE_FUSED(9)
MI_EMU_DECLARE_FUSED(10)
MI_EMU_DECLARE_FUSED(11)
MI_EMU_DECLARE_FUSED(12)
MI_EMU_DECLARE_FUSED(13)
MI_EMU_DECLARE_FUSED(14)
#undef MI_EMU_DECLARE_FUSED
#define MI_EMU_DECLARE_SPLIT(n) \
  void EmuFusedSplitK##n(const miups::Geometry &g, const miups::IoDesc &io, const miups::FilterTables &t, unsigned items);
MI_EMU_DECLARE_SPLIT(10)
MI_EMU_DECLARE_SPLIT(11)
MI_EMU_DECLARE_SPLIT(12)
MI_EMU_DECLARE_SPLIT(13)
MI_EMU_DECLARE_SPLIT(14)
#undef MI_EMU_DECLARE_SPLIT
