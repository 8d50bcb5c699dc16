// U-Net decoder support (reference common/network_ao.py:48-55): the learned 3x3
// stride-2 transposed convolution is evaluated by conv_mfma_kernel (kernels_conv.hip)
// as a 2x2-tap convolution over the input grid whose 4*Cout output channels are the
// four sub-pixel phases; see kernels.h.  This file holds the host-side filter rewrite.
#include "kernels.h"

namespace ukbb {

void tconv_as_conv2x2(const float *w, int cin, int cout, float *dst) {
    const int c4 = 4 * cout;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int ci = 0; ci < cin; ++ci)
                for (int py = 0; py < 2; ++py)
                    for (int px = 0; px < 2; ++px) {
                        const int kh = py == 0 ? (a == 0 ? 2 : 0) : (a == 1 ? 1 : -1);
                        const int kw = px == 0 ? (b == 0 ? 2 : 0) : (b == 1 ? 1 : -1);
                        for (int co = 0; co < cout; ++co) {
                            const float v = (kh < 0 || kw < 0) ? 0.f : w[((size_t)(kh * 3 + kw) * cin + ci) * cout + co];
                            dst[((size_t)(a * 2 + b) * cin + ci) * c4 + (py * 2 + px) * cout + co] = v;
                        }
                    }
}

}  // namespace ukbb
