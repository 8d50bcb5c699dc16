// U-Net decoder pieces (reference common/network_ao.py:48-55): learned 3x3
// stride-2 transposed convolution + BN + ReLU.  Placeholder until the 4-phase
// sub-pixel kernel lands; the engine reports UKBB_EARCH for U-Net models.
#include "kernels.h"

namespace ukbb {

hipError_t launch_tconv(const TconvArgs &, hipStream_t) { return hipErrorNotSupported; }
size_t pack_tconv_weights(const float *, int, int, float *) { return 0; }

}  // namespace ukbb
