// OPRA EQ record -> Equalizer APO text: the step between the OPRA catalogue and the EQ path.
// Reference: scripts/integration/opra.py:50-100 (EqProfile.to_apo_format), :103-125 (slope_to_q), :128-174
// (convert_opra_band), :177-205 (convert_opra_to_apo), :208-244 (apply_modern_target_correction; band constants in
// scripts/modern_target.py:43-49). The catalogue itself (download, cache, search) is out of scope.
#pragma once

#include <string>

namespace miups {

// eqJson: one OPRA EQ record, {"name", "author", "details", "parameters": {"gain_db", "bands": [{"type", "frequency",
// "gain_db", "q", "slope"}, ...]}}. Returns the APO text the reference writes for it (no trailing newline; empty for a
// record without preamp and bands). modernTarget: append the KB5000_7 correction band and lower the preamp by its gain.
// null in a numeric field counts as absent (the reference would raise on it).
bool OpraToApo(const std::string &eqJson, bool modernTarget, std::string *apoText, std::string *error);

}  // namespace miups
