// HIP error plumbing shared by the translation units that launch kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <sstream>
#include <string>

namespace miups {

inline bool HipOk(hipError_t e, const char *what, std::string *error) {
  if (e == hipSuccess) {
    return true;
  }
  std::ostringstream os;
  os << what << ": " << hipGetErrorString(e);
  if (error) {
    *error = os.str();
  }
  return false;
}

}  // namespace miups

#define MI_HIP(call)                              \
  do {                                            \
    if (!::miups::HipOk((call), #call, error)) {  \
      return false;                               \
    }                                             \
  } while (0)
