// Host unit test of the RF_* knob table (retinaface_amd/csrc/knobs.cpp), compiled twice by tests/test_host.py: as the product (no RF_PROBES)
// and as the probe build (-DRF_PROBES).  The environment is set by the test; this program prints what knob() returns.
#include <cstdio>

#include "../../retinaface_amd/csrc/knobs.cpp"

int main() {
    using namespace rf;
    printf("probes %d\n", probes_compiled() ? 1 : 0);
    printf("RF_STEM2_V2 %d\n", knob(K_STEM2_V2));
    printf("RF_CONV3WS %d\n", knob(K_CONV3WS));
    printf("RF_TILE128 %d\n", knob(K_TILE128));
    printf("RF_WIDE_I8 %d\n", knob(K_WIDE_I8));
    printf("RF_FORCE_SCATTER %d\n", knob(K_FORCE_SCATTER));
    printf("RF_PREBUILD_LANES %d\n", knob(K_PREBUILD_LANES));
    printf("RF_BLEND_FP32 %d\n", knob(K_BLEND_FP32));
    printf("min_rounds %.2f\n", knob_persist_min_rounds());
    for (int k = 0; k < K_COUNT; k++)
        if (!knob_name((Knob)k) || knob_name((Knob)k)[0] != 'R') { printf("table row %d has no name\n", k); return 1; }
    return 0;
}
