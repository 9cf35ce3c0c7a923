// Developer switches of the library (host side; no HIP dependency: conv3x3_spatial.hpp is also compiled by the CPU tests).
#pragma once
#include <stdlib.h>

namespace tia {

// Developer switches (A/B measurements, parity audits of fall-back paths: TIA_CONV_NO_RING, TIA_NO_CCL_TILE, TIA_MORPH_FORCE_UF, ...)
// are read only when TIA_DEV=1 is set as well: without it the library has ONE configuration -- the tested one -- whatever else is
// in the environment.
inline const char* dev_env(const char* name) {
    static const bool on = [] {
        const char* e = getenv("TIA_DEV");
        return e != nullptr && e[0] == '1';
    }();
    return on ? getenv(name) : nullptr;
}

}  // namespace tia
