#include "context.h"
namespace lw {
int ntt_bb_device(Context &, lw_layout_t, lw_dir_t, const void *, void *, uint32_t, uint32_t, uint64_t, const void *, hipStream_t) {
    set_error("BabyBear NTT not built yet");
    return LW_ERR_BAD_ARG;
}
int msm_device(Context &, lw_curve_t, const uint64_t *, const void *, size_t, void *, hipStream_t) {
    set_error("MSM not built yet");
    return LW_ERR_BAD_ARG;
}
}
