// ABI bookkeeping for libosi_hip.so.
#include "osi_common.h"

#include <cstring>

OsiTuning g_osi_tuning = {/*wgrad_tile*/ 0, /*wgrad_blocks*/ 2048, /*wgrad_nst*/ 1, /*bn_grid*/ 1024, /*bn_single_p*/ 128, /*wgrad3*/ 2, /*wgrad3_blocks*/ 768, /*fwd_wide*/ 0, /*dgrad_wide*/ 0, /*wgrad_group*/ 2, /*tail_split*/ 1, /*tail_cus*/ 0, /*tail_smax*/ 8, /*tail_mint*/ 16, /*stem_direct*/ 1, /*tail_gain*/ 8, /*tail_qmax*/ 8, /*bn_grid_bwd*/ 1024, /*bn_wide_p*/ 2048, /*fwd_rows*/ 1, /*fwd_w3*/ 1, /*dgrad_w3*/ 1, /*fwd_wino*/ 1, /*dgrad_wino*/ 1, /*wgrad_wino*/ 1, /*wino_wide*/ 1, /*wino_streamk*/ 2, /*dp_reserved_cus*/ 0};

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
    if (!strcmp(name, "dp_reserved_cus")) return &g_osi_tuning.dp_reserved_cus;
    if (!strcmp(name, "fwd_rows")) return &g_osi_tuning.fwd_rows;
    if (!strcmp(name, "fwd_w3")) return &g_osi_tuning.fwd_w3;
    if (!strcmp(name, "dgrad_w3")) return &g_osi_tuning.dgrad_w3;
    if (!strcmp(name, "fwd_wino")) return &g_osi_tuning.fwd_wino;
    if (!strcmp(name, "dgrad_wino")) return &g_osi_tuning.dgrad_wino;
    if (!strcmp(name, "wino_streamk")) return &g_osi_tuning.wino_streamk;
    if (!strcmp(name, "wgrad_wino")) return &g_osi_tuning.wgrad_wino;
    if (!strcmp(name, "wino_wide")) return &g_osi_tuning.wino_wide;
    return nullptr;
}
}  // namespace

extern "C" {
int osi_abi_version(void) { return 7; }   // 7: inference forms (osi_conv_fwd_epilogue, osi_conv_fwd_wino_epilogue_pre, osi_bn_eval_coeffs_multi), executor option "eval_fused"; 6: Winograd forms (osi_conv_*_wino), knobs "fwd_wino" / "dgrad_wino"; 5: osi_resnet50_grads_ready, executor option "stage_join", knob "dp_reserved_cus", range-checked knobs, plan snapshot (4: addend_stride, accumulate = 2)
int osi_set_tuning(const char* name, int value) {
    int* s = tuning_slot(name);
    if (!s) return OSI_ERR_ARG;
    // every knob has a range; a value outside it is refused instead of silently switching a plan off
    OsiTuning& t = g_osi_tuning;
    auto in = [&](int lo, int hi) { return value >= lo && value <= hi; };
    bool good = true;
    if (s == &t.wgrad_tile) good = value == 0 || value == 64;
    else if (s == &t.wgrad_blocks || s == &t.wgrad3_blocks) good = in(1, 1 << 20);
    else if (s == &t.wgrad_nst) good = in(1, 2);
    else if (s == &t.bn_grid || s == &t.bn_grid_bwd) good = in(1, 1 << 20);
    else if (s == &t.bn_single_p) good = in(1, 1 << 20);
    else if (s == &t.bn_wide_p) good = in(0, 2048);
    else if (s == &t.wgrad3 || s == &t.wgrad_group) good = in(0, 2);
    else if (s == &t.fwd_wide || s == &t.dgrad_wide || s == &t.tail_split || s == &t.stem_direct || s == &t.fwd_w3 || s == &t.dgrad_w3 || s == &t.fwd_wino || s == &t.dgrad_wino || s == &t.wgrad_wino || s == &t.wino_wide) good = in(0, 1);
    else if (s == &t.wino_streamk) good = in(0, 3);
    else if (s == &t.tail_cus) good = in(0, 4096);            // 0 = ask the device
    else if (s == &t.tail_smax) good = in(1, 64);
    else if (s == &t.tail_mint) good = in(1, 4096);
    else if (s == &t.tail_gain) good = in(0, 100);
    else if (s == &t.tail_qmax) good = in(0, 4096);
    else if (s == &t.dp_reserved_cus) good = in(0, 128);
    else if (s == &t.fwd_rows) good = in(0, 2);
    if (!good) return OSI_ERR_ARG;
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
        case OSI_ERR_STATE: return "executor called out of order, or a plan-relevant tuning knob changed after osi_resnet50_create";
        default: return "unknown error";
    }
}
}
