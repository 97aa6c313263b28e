// experiments_absent.cc - what the product library holds in place of csrc/decode_mega.hip (the whole decode step in one launch, round 1) and
// csrc/decode_engine.hip (one persistent launch per layer, round 3).  Both lost their A/Bs against the per-launch path and are kept in the
// tree only as experiments: build.py compiles them when MI355_BUILD_EXPERIMENTS=1 is set and this file otherwise, so that the default library carries
// neither their kernels nor their launchers.  host/runtime.cc asks `*_applicable` before it prepares either path: false here, it never gets further.
#include "kernels.h"

namespace mi355 {

bool experiments_built() { return false; }

int mega_blocks() { return 0; }
bool decode_mega_applicable(int, int, int, int, int) { return false; }
hipError_t launch_decode_mega(const MegaLayer *, int, int, int, const AttnArgs &, const float *, int, const float *, const float *, const int32_t *, unsigned *, unsigned *,
                              int *, unsigned long long *, size_t, hipStream_t) {
    return hipErrorNotSupported;
}

bool decode_engine_applicable(const EngineLayer &, int, int) { return false; }
void decode_engine_plan(EngineLayer &) {}
size_t decode_engine_granule_words(int, int) { return 0; }
void decode_engine_set_error_word(unsigned *) {}
hipError_t launch_decode_engine(const EngineLayer *, int, int, unsigned long long *, const unsigned *, int, unsigned long long *, hipStream_t) { return hipErrorNotSupported; }

}  // namespace mi355
