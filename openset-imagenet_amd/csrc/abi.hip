// ABI bookkeeping for libosi_hip.so.
#include "osi_common.h"

#include <cstring>

OsiTuning g_osi_tuning = {/*wgrad_tile*/ 0, /*wgrad_blocks*/ 2048, /*wgrad_nst*/ 1, /*bn_grid*/ 1024, /*bn_single_p*/ 128, /*wgrad3*/ 2, /*wgrad3_blocks*/ 768, /*fwd_wide*/ 0, /*dgrad_wide*/ 0, /*wgrad_group*/ 2, /*tail_split*/ 1, /*tail_cus*/ 0, /*tail_smax*/ 8, /*tail_mint*/ 16, /*stem_direct*/ 1, /*tail_gain*/ 8, /*tail_qmax*/ 8, /*bn_grid_bwd*/ 1024, /*bn_wide_p*/ 2048};

namespace {
int* tuning_slot(const char* name) {
    if (!name) return nullptr;
    if (!strcmp(name, "wgrad_tile")) return &g_osi_tuning.wgrad_tile;
    if (!strcmp(name, "wgrad_blocks")) return &g_osi_tuning.wgrad_blocks;
    if (!strcmp(name, "wgrad_nst")) return &g_osi_tuning.wgrad_nst;
    if (!strcmp(name, "bn_grid")) return &g_osi_tuning.bn_grid;
    if (!strcmp(name, "bn_single_p")) return &g_osi_tuning.bn_single_p;
    if (!strcmp(name, "wgrad_group")) return &g_osi_tuning.wgrad_group;
    if (!strcmp(name, "wgrad3")) return &g_osi_tuning.wgrad3;
    if (!strcmp(name, "fwd_wide")) return &g_osi_tuning.fwd_wide;
    if (!strcmp(name, "dgrad_wide")) return &g_osi_tuning.dgrad_wide;
    if (!strcmp(name, "wgrad3_blocks")) return &g_osi_tuning.wgrad3_blocks;
    if (!strcmp(name, "tail_split")) return &g_osi_tuning.tail_split;
    if (!strcmp(name, "tail_cus")) return &g_osi_tuning.tail_cus;
    if (!strcmp(name, "tail_smax")) return &g_osi_tuning.tail_smax;
    if (!strcmp(name, "tail_mint")) return &g_osi_tuning.tail_mint;
    if (!strcmp(name, "stem_direct")) return &g_osi_tuning.stem_direct;
    if (!strcmp(name, "bn_wide_p")) return &g_osi_tuning.bn_wide_p;
    if (!strcmp(name, "bn_grid_bwd")) return &g_osi_tuning.bn_grid_bwd;
    if (!strcmp(name, "tail_gain")) return &g_osi_tuning.tail_gain;
    if (!strcmp(name, "tail_qmax")) return &g_osi_tuning.tail_qmax;
    return nullptr;
}
}  // namespace

extern "C" {
int osi_abi_version(void) { return 4; }   // 4: osi_dgrad_fusion.addend_stride, osi_conv_dgrad accumulate = 2 (3: gate read-out, geometry, stem kernels)
int osi_set_tuning(const char* name, int value) {
    int* s = tuning_slot(name);
    if (!s) return OSI_ERR_ARG;
    if ((s == &g_osi_tuning.wgrad_blocks || s == &g_osi_tuning.wgrad3_blocks) && value < 1) return OSI_ERR_ARG;
    if (s == &g_osi_tuning.wgrad_nst && value != 1 && value != 2) return OSI_ERR_ARG;
    if ((s == &g_osi_tuning.bn_grid || s == &g_osi_tuning.bn_grid_bwd) && value < 1) return OSI_ERR_ARG;
    if (s == &g_osi_tuning.bn_single_p && value < 1) return OSI_ERR_ARG;
    if (s == &g_osi_tuning.bn_wide_p && value < 0) return OSI_ERR_ARG;
    *s = value;
    return OSI_OK;
}
int osi_get_tuning(const char* name, int* value) {
    int* s = tuning_slot(name);
    if (!s || !value) return OSI_ERR_ARG;
    *value = *s;
    return OSI_OK;
}
const char* osi_build_arch(void) { return "gfx950"; }
const char* osi_strerror(int code) {
    switch (code) {
        case OSI_OK: return "ok";
        case OSI_ERR_ARG: return "invalid argument (shape, pointer or alignment precondition)";
        case OSI_ERR_LAUNCH: return "HIP launch failed";
        case OSI_ERR_STATE: return "executor called out of order";
        default: return "unknown error";
    }
}
}
