// TEST INFRASTRUCTURE ONLY -- the fused kernels of ONE transform length (-DEMU_TU_LOG2K=n, 5..14) for the thread-emulation
// driver; see emu_launch.h.
#include <cstdio>
#include <cstdlib>

#include "device/kernel_fused.h"

#include "emu_launch.h"

using namespace miups;

namespace {

template <int LOG2K>
void EmuFused(const Geometry &g, const IoDesc &io, const FilterTables &t, unsigned items) {
  const FusedTables ft{t.tw.data(), t.WmT.data(), t.blockB.data(), t.GT.data(), t.G0.data(), t.Wb, nullptr, t.Wself};
  if constexpr (LOG2K >= 10) {
    if (t.fusedNarrow) {  // one butterfly per thread (experiment form, EMU_NARROW)
      using CfgN = FusedCfg<LOG2K, 1>;
      if (io.ext_epilogue) {
        miups_emu::launch(items, CfgN::T, CfgN::LDS_BYTES, true, [&]() { fused_kernel<LOG2K, true, 1>(g, io, ft); });
      } else {
        miups_emu::launch(items, CfgN::T, CfgN::LDS_BYTES, true, [&]() { fused_kernel<LOG2K, false, 1>(g, io, ft); });
      }
      return;
    }
  }
  using Cfg = FusedCfg<LOG2K, 2, false>;
  if constexpr (fused_plan_r32_exists(LOG2K, 2)) {
    if (t.fusedR32) {  // radix-32 pass plan (experiment, EMU_R32)
      if (io.ext_epilogue) {
        miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES, true, [&]() { fused_kernel<LOG2K, true, 2, true>(g, io, ft); });
      } else {
        miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES, true, [&]() { fused_kernel<LOG2K, false, 2, true>(g, io, ft); });
      }
      return;
    }
  }
  if (io.phase_parts > 1) {  // small-call form (EMU_PARTS): `items` counts workgroups = work items * phase_parts
    if constexpr (LOG2K >= kPartsMinLog2K) {
      miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES, true, [&]() { fused_parts_kernel<LOG2K>(g, io, ft); });
      return;
    }
    std::fprintf(stderr, "EMU_PARTS: no phase-split kernel for this transform length\n");
    std::exit(2);
  }
  if (io.ext_epilogue) {
    miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES, true, [&]() { fused_kernel<LOG2K, true, 2, false>(g, io, ft); });
  } else {
    miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES, true, [&]() { fused_kernel<LOG2K, false, 2, false>(g, io, ft); });
  }
}

template <int LOG2K>
void EmuFusedSplit(const Geometry &g, const IoDesc &io, const FilterTables &t, unsigned items) {
  using Cfg = FusedCfg<LOG2K>;
  const FusedTables ft{t.tw.data(), t.WmT.data(), t.blockB.data(), t.GT.data(), t.G0.data(), t.Wb, t.selfW.data(), t.Wself};
  if (io.phase_parts > 1) {  // small-call form (EMU_PARTS): `items` counts workgroups
    miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES_SPLIT, true, [&]() { fused_split_parts_kernel<LOG2K>(g, io, ft); });
    return;
  }
  miups_emu::launch(items, Cfg::T, Cfg::LDS_BYTES_SPLIT, true, [&]() { fused_split_kernel<LOG2K>(g, io, ft); });
}

}  // namespace

#define MI_EMU_DEFINE_(n)                                                                                   \
  void EmuFusedK##n(const Geometry &g, const IoDesc &io, const FilterTables &t, unsigned items) { EmuFused<n>(g, io, t, items); }
#define MI_EMU_DEFINE(n) MI_EMU_DEFINE_(n)
MI_EMU_DEFINE(EMU_TU_LOG2K)
#if EMU_TU_LOG2K >= 10
#define MI_EMU_DEFINE_SPLIT_(n)                                                                                 \
  void EmuFusedSplitK##n(const Geometry &g, const IoDesc &io, const FilterTables &t, unsigned items) { \
    EmuFusedSplit<n>(g, io, t, items);                                                                           \
  }
#define MI_EMU_DEFINE_SPLIT(n) MI_EMU_DEFINE_SPLIT_(n)
MI_EMU_DEFINE_SPLIT(EMU_TU_LOG2K)
#endif
