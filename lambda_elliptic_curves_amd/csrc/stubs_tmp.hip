#include "context.h"
namespace lw {
int ntt_bb_device(Context &, lw_layout_t, lw_dir_t, const void *, void *, uint32_t, uint32_t, uint64_t, const void *, hipStream_t) {
    set_error("BabyBear NTT not built yet");
    return LW_ERR_BAD_ARG;
}
}
