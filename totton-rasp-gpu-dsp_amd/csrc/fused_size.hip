// One transform length of the fused kernel per compile: -DMIUPS_TU_LOG2K=n (5..14) or -DMIUPS_TU_SPLIT (the split form).
#include "fused_launch.h"
#include "fused_launch_impl.h"

namespace miups {

#if defined(MIUPS_TU_SPLIT)
bool LaunchFusedSplitK14(const Geometry &g, const IoDesc &io, const FusedTables &ft, unsigned items, hipStream_t st,
                         std::string *error) {
  return LaunchFusedSplit<14>(g, io, ft, items, st, error);
}
#else
#define MI_DEFINE_FUSED_(n)                                                                                              \
  bool LaunchFusedK##n(const Geometry &g, const IoDesc &io, const FusedTables &ft, bool narrow, bool r32, unsigned items, \
                       hipStream_t st, std::string *error) {                                                          \
    return LaunchFused<n>(g, io, ft, narrow, r32, items, st, error);                                                  \
  }
#define MI_DEFINE_FUSED(n) MI_DEFINE_FUSED_(n)
MI_DEFINE_FUSED(MIUPS_TU_LOG2K)
#endif

}  // namespace miups
