// Whole-network executor: one C call enqueues the complete ResNet-50 forward (or a range of backward stages)
// of the reference model on a HIP stream — no per-layer host round trips, graph-capturable.
//
// Topology restated from the reference's only constructor call, openset_imagenet/model.py:17-26
// (torchvision.models.resnet50 = ResNet v1.5: stride on the 3x3 of the first block of stages 2-4, Bottleneck
// expansion 4, 1x1-stride-s + BN downsample on block 0 of each stage; then fc = Linear(2048, fc_dim) and
// logits = Linear(fc_dim, out_features, bias=logit_bias)); forward = model.py:28-39; backward = the autograd
// graph that openset_imagenet/train.py:138 (j.backward()) walks. Parameter order and names = nn.Module
// registration order, i.e. the reference checkpoint's state_dict keys.
#include "osi_common.h"
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>

#define OSI_TRY(x)                 \
    do {                           \
        int e__ = (x);             \
        if (e__ != OSI_OK) return e__; \
    } while (0)

namespace {

struct Tensor {
    std::string name;
    int ndim; int shape[4];
    size_t off, numel;  // floats in the param arena
};
struct BN {
    std::string prefix;
    int C, M;
    size_t g_off, b_off;    // param arena
    size_t rm_off, rv_off;  // buffer arena
    size_t mean, invstd, scale, shift;  // workspace (floats)
};
struct Conv {
    osi_conv_desc d;
    size_t w_off;  // param arena
    int bn;
    size_t y;      // workspace: conv output (pre-BN)
    size_t a;      // workspace: BN(+res)+ReLU output (SIZE_MAX for downsample: goes to scratch)
    size_t mask;   // workspace: ReLU bitmask of `a` (1 bit per element)
    size_t u_fw = (size_t)-1, u_bw = (size_t)-1;   // workspace: Winograd-transformed weights (forward / input-gradient form), 3x3 stride-1 layers only
};
struct Block {
    int c1, c2, c3, ds;  // conv indices, ds = -1 if identity skip
    size_t x_in;         // workspace offset of the block input
    size_t out;          // = convs[c3].a
    int stage;           // backward stage this block belongs to
};

constexpr size_t ALIGN_F = 64;  // floats (256 B)
// The executor's INTERNAL events only order streams of ONE device against each other (fork / join of the weight-gradient side stream,
// reader events of the scratch buffers). By default a HIP event performs a SYSTEM-scope fence when it is recorded — a cache writeback +
// invalidate that makes device memory visible to the host and to other devices — which none of them needs: the kernels on both sides
// carry their own agent-scope acquire / release. ~115 records per step.
constexpr unsigned EV_FLAGS = hipEventDisableTiming | hipEventDisableSystemFence;
// The two events of the data-parallel hand-off (osi_resnet50_grads_ready) are different: their consumer is RCCL's all-reduce, whose
// peers read this device's gradient arena over xGMI. They keep the system-scope release (<= 8 records per step).
constexpr unsigned EV_FLAGS_HANDOFF = hipEventDisableTiming;
static size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static int device_cus() {   // CUs of the CURRENT device (the launch plans are balanced for them): 0 when no device answers
    int v = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return v;
}

}  // namespace

struct osi_resnet50 {
    int B, H, W, F, O, logit_bias;
    std::vector<Tensor> tensors;
    std::vector<BN> bns;
    std::vector<Conv> convs;
    std::vector<Block> blocks;
    size_t param_floats = 0, buffer_floats = 0, ws_floats = 0;
    // head tensors
    int t_fc_w, t_fc_b, t_lg_w, t_lg_b;
    // stem
    int Hs, Ws, Hp, Wp;              // stem conv output, maxpool output
    size_t x4, wpack, gpack, a_pool, pool_idx, pooled, feat, logits_ws;
    size_t bn_ws, bn_ws2, bn_ws_bytes, wg_ws, wg_ws_bytes, dg_ws, dg_ws_bytes;   // bn_ws2: BatchNorm scratch of the side-stream branch
    size_t stem_ws = 0, stem_ws_bytes = 0;
    size_t wino_ws = 0, wino_ws_bytes = 0;   // transformed weights of the Winograd forms (main stream only: forward conv2, in-block input gradients)
    static constexpr int NSCR = 12;   // scratch activations-gradient buffers (each = largest activation)
    size_t scratch[NSCR], scratch_floats;
    size_t dfeat, dpooled;
    int Hf, Wf;                      // final spatial size
    // stage bookkeeping
    int n_stages = 4;
    size_t stage_lo[4], stage_hi[4];
    // run state
    bool fwd_done = false;
    bool any_fwd = false;            // the workspace holds the ReLU / arg-max decisions of a forward (osi_resnet50_debug_gate refuses to read an
                                     // empty workspace — or one an inference forward ran in: that one stores activations, not pre-BN tensors or bitmasks)
    int next_stage = 0;
    int cur_grad = -1;               // scratch index holding the upstream gradient between stages
    bool go_fused = false;           // cur_grad is already ReLU-masked and its BatchNorm reductions wait in dg_ws
    int fused_P = 0;                 // row tiles of the partials in dg_ws
    std::vector<int> free_list;
    // optional HIP-event instrumentation: one event after every op, tagged with the op's class
    bool prof_on = false;
    bool prof_timeline = false;      // mode 2: keep the side-stream overlap, record where each op ran (osi_resnet50_timeline_read)
    std::vector<int> prof_side;      // per event: 1 = recorded on the side stream
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_cls;
    int prof_n = 0;
    int mark(int cls, hipStream_t st) {
        if (!prof_on) return OSI_OK;
        if (prof_n == (int)prof_ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return OSI_ERR_LAUNCH;
            prof_ev.push_back(e); prof_cls.push_back(0); prof_side.push_back(0);
        }
        prof_cls[prof_n] = cls;
        prof_side[prof_n] = (side != nullptr && st == side) ? 1 : 0;
        if (hipEventRecord(prof_ev[prof_n], st) != hipSuccess) return OSI_ERR_LAUNCH;
        ++prof_n;
        return OSI_OK;
    }

    size_t add_tensor(const std::string& name, int ndim, const int* shape) {
        Tensor t; t.name = name; t.ndim = ndim; t.numel = 1;
        for (int i = 0; i < 4; ++i) t.shape[i] = i < ndim ? shape[i] : 1;
        for (int i = 0; i < ndim; ++i) t.numel *= (size_t)shape[i];
        t.off = param_floats;
        param_floats = up(param_floats + t.numel, 4);
        tensors.push_back(t);
        return t.off;
    }
    size_t ws_alloc(size_t floats) {
        size_t o = ws_floats;
        ws_floats = up(ws_floats + floats, ALIGN_F);
        return o;
    }
    int add_conv_bn(const std::string& cname, const std::string& bname, int Bn, int Hin, int Win, int Cin, int Cout, int k, int s,
                    int pad, bool keep_act) {
        Conv c{};
        c.d.B = Bn; c.d.H = Hin; c.d.W = Win; c.d.Cin = Cin; c.d.Cout = Cout; c.d.R = k; c.d.S = k; c.d.stride = s; c.d.pad = pad;
        c.d.Ho = (Hin + 2 * pad - k) / s + 1; c.d.Wo = (Win + 2 * pad - k) / s + 1;
        int shp[4] = {Cout, Cin == 4 && k == 7 ? 3 : Cin, k, k};
        c.w_off = add_tensor(cname + ".weight", 4, shp);
        BN b; b.prefix = bname; b.C = Cout; b.M = Bn * c.d.Ho * c.d.Wo;
        int s1[1] = {Cout};
        b.g_off = add_tensor(bname + ".weight", 1, s1);
        b.b_off = add_tensor(bname + ".bias", 1, s1);
        b.rm_off = buffer_floats; buffer_floats += up(Cout, 4);
        b.rv_off = buffer_floats; buffer_floats += up(Cout, 4);
        b.mean = ws_alloc(Cout); b.invstd = ws_alloc(Cout); b.scale = ws_alloc(Cout); b.shift = ws_alloc(Cout);
        c.bn = (int)bns.size();
        bns.push_back(b);
        size_t n = (size_t)b.M * Cout;
        c.y = ws_alloc(n);
        c.a = keep_act ? ws_alloc(n) : (size_t)-1;
        c.mask = keep_act ? ws_alloc(osi_bn_relu_mask_bytes(b.M, Cout) / sizeof(float)) : (size_t)-1;
        convs.push_back(c);
        return (int)convs.size() - 1;
    }
    // Weight gradients run on a low-priority side stream: they are off the critical path of backward (nothing downstream
    // reads them before the optimizer) and fill the matrix pipes while the main stream is in HBM-bound BatchNorm kernels
    // or in the ragged last round of a dgrad launch. buf_ev[i] = last side-stream reader of scratch buffer i.
    void* staged_ws = nullptr;   // workspace whose input buffer was filled by osi_resnet50_stage_input_u8 (consumed by one forward)
    bool overlap = true;
    bool fwd_fork = true;            // projection shortcut of the forward pass on the side stream
    bool fwd_recompute = false;      // identity-shortcut block outputs recomputed by the next conv1, their pass moved off the critical
                                     // path. Built, bit-exact, and measured: NO gain (35.48 vs 35.45 ms) — conv1 with the second operand
                                     // stream is 5-27 % slower, which eats the hidden pass. Off by default; kept for A/B.
    bool side_prio_normal = false;   // side stream at default instead of lowest priority (read when the stream is created)
    const float* x4_ext = nullptr;   // external NHWC4 input bound by osi_resnet50_bind_input_nhwc4 (consumed by one forward)
    const float* x4_cur = nullptr;   // input of the step in flight (forward sets it, the stem weight gradient reads it)
    bool stem_pool_stats = true;     // option "stem_pool_stats": bn1's backward reductions come out of layer1.0.conv1's dgrad epilogue
    bool ds_sparse = true;           // option "ds_sparse": stride-2 shortcut gradients write / are read at the even-even pixels only
    bool stem_wgrad_main = true;     // option "stem_wgrad_main": the fused stem weight gradient runs on the main stream (own workspace)
    int stem_stats_P = 0;            // > 0: bn1's backward partial sums wait in dg_ws (left by the pool-mode epilogue of layer1.0.conv1's dgrad)
    bool stem_fused = true;          // option "stem_fused": conv1's weight gradient builds dY in its operand loader (osi_stem_wgrad_fused)
                                     // behind the BatchNorm reductions: no 112x112x64 gradient tensor, no apply pass (step -0.15 ms)
    bool stagger = false;            // option "stagger": a weight gradient starts when the input gradient of the SAME layer has finished
                                     // (beside the next BatchNorm-backward kernels) and the next input gradient waits for it: matrix-bound
                                     // kernels never co-run, only HBM-bound work overlaps them. A/B against the default co-running schedule.
    struct PendingW { bool on = false; int ci = 0, gi = 0, in_bn = -1; const float* conv_in = nullptr; float* grads = nullptr; float* ws = nullptr; } pend;
    bool w_inflight = false;
#ifdef OSI_DIAG                       // `make -C csrc diag` (libosi_hip_diag.so, tools only): the product library has no such switch
    int dbg_fwd_count = 0;           // training forwards so far (dbg_skip bit 3)
    int dbg_skip = 0;                // option "dbg_skip" (TIMING EXPERIMENTS ONLY, results are wrong): bit 0 = the BatchNorm-backward apply passes
                                     // are not launched (their reductions still are), bit 1 = the block-output passes of the forward are
                                     // not launched, bit 2 = with "fwd_recompute" the deferred block-output pass is not launched either — upper bounds for what folding
                                     // those passes into their consumers could buy; bit 3 = the forward's BatchNorm finalize launches are not
                                     // launched (round 6: the ceiling of merging them)
#else
    static constexpr int dbg_skip = 0;
    static constexpr int dbg_fwd_count = 0;
#endif
    bool eval_fused = true;          // option "eval_fused": a forward with training = 0 runs the inference forms (forward_eval_fused); 0 = the
                                     // training topology on running statistics (A/B, same bits)
    bool stage_join = true;          // option "stage_join": a staged backward call (stage_hi < stages) ends by joining the side stream into
                                     // the caller's stream. 0 (data parallel): only the LAST stage joins; the caller hands each finished
                                     // stage to its communication stream with osi_resnet50_grads_ready, and the compute stream runs on
    OsiTuning plan_knobs;            // the process-wide knobs the workspace was sized for (osi_resnet50_create); a launch under other
    int plan_hw_cus = 0;             // values (or on a device with another CU count) is refused (OSI_ERR_STATE) instead of running a
                                     // plan the workspace does not fit
    bool plan_unchanged() const {
        const OsiTuning &a = plan_knobs, &b = g_osi_tuning;
        return plan_hw_cus == device_cus() && a.wgrad_tile == b.wgrad_tile && a.wgrad_blocks == b.wgrad_blocks && a.wgrad3 == b.wgrad3 && a.wgrad3_blocks == b.wgrad3_blocks &&
               a.tail_split == b.tail_split && a.tail_cus == b.tail_cus && a.tail_smax == b.tail_smax && a.tail_mint == b.tail_mint &&
               a.tail_gain == b.tail_gain && a.tail_qmax == b.tail_qmax && a.stem_direct == b.stem_direct && a.wgrad_group == b.wgrad_group &&
               a.dp_reserved_cus == b.dp_reserved_cus && a.fwd_wino == b.fwd_wino && a.dgrad_wino == b.dgrad_wino && a.wgrad_wino == b.wgrad_wino;
    }
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_wdone = nullptr, ev_rmain = nullptr, ev_rside = nullptr, ev_wt = nullptr;
    bool wt_aside = true;            // option "wino_weights_aside": the Winograd weight transforms of a training forward run on the side stream (A/B: 0 = main)
    bool wt_pending = false;         // the Winograd weight transforms of this forward run on the side stream: the first consumer waits for ev_wt
    hipEvent_t buf_ev[NSCR] = {};
    bool buf_pending[NSCR] = {};
    bool side_dirty = false;
    int ensure_side() {
        if (side) return OSI_OK;
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return OSI_ERR_LAUNCH;
        const int prio = side_prio_normal ? 0 : lo;   // option "side_priority_normal": default priority for the side stream (A/B)
        if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipEventCreateWithFlags(&ev_fork, EV_FLAGS) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipEventCreateWithFlags(&ev_join, EV_FLAGS) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipEventCreateWithFlags(&ev_wdone, EV_FLAGS) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipEventCreateWithFlags(&ev_wt, EV_FLAGS) != hipSuccess) return OSI_ERR_LAUNCH;
        for (int i = 0; i < NSCR; ++i)
            if (hipEventCreateWithFlags(&buf_ev[i], EV_FLAGS) != hipSuccess) return OSI_ERR_LAUNCH;
        return OSI_OK;
    }
    bool async_wgrad() const { return overlap && (!prof_on || prof_timeline) && side != nullptr; }
    // a scratch buffer may only be rewritten on `st` after its last side-stream reader has finished
    // returns the buffer index, or a negative OSI_ERR_* code
    int take(hipStream_t st) {
        if (free_list.empty()) return OSI_ERR_STATE;
        // FIFO: hand out the buffer that was released longest ago, so its side-stream reader has most likely finished
        int i = free_list.front(); free_list.erase(free_list.begin());
        if (buf_pending[i]) {
            if (hipStreamWaitEvent(st, buf_ev[i], 0) != hipSuccess) return OSI_ERR_LAUNCH;
            buf_pending[i] = false;
        }
        return i;
    }
    void give(int i) { free_list.push_back(i); }
    int join_side(hipStream_t st) {
        if (!side_dirty) return OSI_OK;
        if (hipEventRecord(ev_join, side) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        for (int i = 0; i < NSCR; ++i) buf_pending[i] = false;
        side_dirty = false;
        return OSI_OK;
    }
    ~osi_resnet50() {
        for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e);
        if (side) {
            (void)hipStreamSynchronize(side);
            (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); (void)hipEventDestroy(ev_wdone); (void)hipEventDestroy(ev_wt);
            for (int i = 0; i < NSCR; ++i) (void)hipEventDestroy(buf_ev[i]);
            (void)hipStreamDestroy(side);
        }
        if (ev_rmain) (void)hipEventDestroy(ev_rmain);
        if (ev_rside) (void)hipEventDestroy(ev_rside);
    }
};

extern "C" {

int osi_resnet50_create(osi_resnet50_t* out, int B, int H, int W, int fc_dim, int out_features, int logit_bias) {
    OSI_REQUIRE(out && B > 0 && H >= 32 && W >= 32 && fc_dim > 0 && out_features > 0);
    osi_resnet50* n = new osi_resnet50();
    n->B = B; n->H = H; n->W = W; n->F = fc_dim; n->O = out_features; n->logit_bias = logit_bias ? 1 : 0;
    const std::string rb = "resnet_base.";
    // stem
    n->x4 = n->ws_alloc((size_t)B * H * W * 4);
    n->wpack = n->ws_alloc(64 * 224);
    n->gpack = n->ws_alloc(64 * 224);
    int stem = n->add_conv_bn(rb + "conv1", rb + "bn1", B, H, W, 4, 64, 7, 2, 3, false);   // its activation only exists max-pooled
    n->Hs = n->convs[stem].d.Ho; n->Ws = n->convs[stem].d.Wo;
    n->Hp = (n->Hs + 2 - 3) / 2 + 1; n->Wp = (n->Ws + 2 - 3) / 2 + 1;
    n->a_pool = n->ws_alloc((size_t)B * n->Hp * n->Wp * 64);
    n->pool_idx = n->ws_alloc((size_t)B * n->Hp * n->Wp * 64 / 4);
    // stages
    const int planes[4] = {64, 128, 256, 512}, nblk[4] = {3, 4, 6, 3}, strides[4] = {1, 2, 2, 2};
    int inpl = 64, h = n->Hp, w = n->Wp;
    size_t x = n->a_pool;
    for (int s = 0; s < 4; ++s) {
        size_t lo = n->param_floats;
        for (int b = 0; b < nblk[s]; ++b) {
            const std::string pre = rb + "layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".";
            const int st = b == 0 ? strides[s] : 1;
            Block blk{};
            blk.x_in = x; blk.stage = 3 - s;
            // conv1 / conv2: only the pre-BN output is kept (their activation is recomputed in the consumers' loaders)
            blk.c1 = n->add_conv_bn(pre + "conv1", pre + "bn1", B, h, w, inpl, planes[s], 1, 1, 0, false);
            blk.c2 = n->add_conv_bn(pre + "conv2", pre + "bn2", B, h, w, planes[s], planes[s], 3, st, 1, false);
            const int ho = n->convs[blk.c2].d.Ho, wo = n->convs[blk.c2].d.Wo;
            blk.c3 = n->add_conv_bn(pre + "conv3", pre + "bn3", B, ho, wo, planes[s], planes[s] * 4, 1, 1, 0, true);
            blk.ds = -1;
            if (b == 0) blk.ds = n->add_conv_bn(pre + "downsample.0", pre + "downsample.1", B, h, w, inpl, planes[s] * 4, 1, st, 0, false);
            blk.out = n->convs[blk.c3].a;
            n->blocks.push_back(blk);
            x = blk.out; inpl = planes[s] * 4; h = ho; w = wo;
        }
        n->stage_lo[3 - s] = lo; n->stage_hi[3 - s] = n->param_floats;
    }
    n->stage_lo[3] = 0;  // the stem belongs to the last backward stage
    n->Hf = h; n->Wf = w;
    // head
    { int shp[2] = {fc_dim, 2048}; n->t_fc_w = (int)n->tensors.size(); n->add_tensor(rb + "fc.weight", 2, shp); }
    { int shp[1] = {fc_dim}; n->t_fc_b = (int)n->tensors.size(); n->add_tensor(rb + "fc.bias", 1, shp); }
    { int shp[2] = {out_features, fc_dim}; n->t_lg_w = (int)n->tensors.size(); n->add_tensor("logits.weight", 2, shp); }
    n->t_lg_b = -1;
    if (n->logit_bias) { int shp[1] = {out_features}; n->t_lg_b = (int)n->tensors.size(); n->add_tensor("logits.bias", 1, shp); }
    n->stage_hi[0] = n->param_floats;  // head gradients are final after stage 0 (head + layer4)
    n->pooled = n->ws_alloc((size_t)B * 2048);
    n->feat = n->ws_alloc((size_t)B * fc_dim);
    n->logits_ws = n->ws_alloc((size_t)B * out_features);
    n->dfeat = n->ws_alloc((size_t)B * fc_dim);
    n->dpooled = n->ws_alloc((size_t)B * 2048);
    // scratch sizing
    size_t maxact = 0, bnws = 0, wgws = 0, dgws = 0, winows = 0;
    for (auto& c : n->convs) {
        size_t e = (size_t)c.d.B * c.d.Ho * c.d.Wo * c.d.Cout;
        if (e > maxact) maxact = e;
        size_t ein = (size_t)c.d.B * c.d.H * c.d.W * c.d.Cin;
        if (ein > maxact) maxact = ein;
        size_t b1 = osi_bn_workspace(n->bns[c.bn].M, c.d.Cout), b2 = osi_bn_backward_workspace(n->bns[c.bn].M, c.d.Cout);
        if (b1 > bnws) bnws = b1;
        if (b2 > bnws) bnws = b2;
        size_t b3 = osi_conv_fwd_bnstats_workspace(&c.d);
        if (b3 > bnws) bnws = b3;
        size_t wg = osi_conv_wgrad_workspace(&c.d);
        if (wg > wgws) wgws = wg;
        wg = osi_stem_wgrad_direct_workspace(&c.d);
        if (wg > wgws) wgws = wg;
        wg = osi_conv_wgrad_wino_workspace(&c.d);
        if (wg > wgws) wgws = wg;
        if (!(c.d.Cin == 4 && c.d.R == 7)) { size_t dg = osi_conv_dgrad_fused_workspace(&c.d); if (dg > dgws) dgws = dg; }
        if (osi_conv_wino_eligible(&c.d, 0) || osi_conv_wino_eligible(&c.d, 1)) {
            // Winograd forms: one (mean, M2) / (sum g, sum g xhat) partial per 16 tiles (+ the merge scratch behind the statistics); the
            // transformed weights of both directions live per layer (built on the side stream at the start of a forward pass)
            const size_t Pw = ((size_t)c.d.B * ((c.d.H + 1) / 2) * ((c.d.W + 1) / 2) + 15) / 16;
            if (osi_conv_wino_eligible(&c.d, 0)) c.u_fw = n->ws_alloc(osi_conv_wino_weights_bytes(&c.d) / sizeof(float));
            if (osi_conv_wino_eligible(&c.d, 1)) c.u_bw = n->ws_alloc(osi_conv_wino_weights_bytes(&c.d) / sizeof(float));
            size_t w = osi_conv_wino_slab_bytes();
            if (w > winows) winows = w;
            w = (2 * Pw + 64) * c.d.Cout * sizeof(float);
            if (w > bnws) bnws = w;
            w = 3 * Pw * c.d.Cin * sizeof(float);
            if (w > dgws) dgws = w;
        }
    }
    n->wino_ws_bytes = winows; n->wino_ws = n->ws_alloc(winows / 4 + 4);
    n->bn_ws_bytes = bnws; n->bn_ws = n->ws_alloc(bnws / 4 + 4); n->bn_ws2 = n->ws_alloc(bnws / 4 + 4);
    n->wg_ws_bytes = wgws; n->wg_ws = n->ws_alloc(wgws / 4 + 4);
    // the fused stem weight gradient has its own slab: it may run on the main stream while the side stream still owns wg_ws
    // (0 bytes = a geometry / knob setting the fused form does not take: the executor then never calls it — see stem_fused_ok)
    n->stem_ws_bytes = osi_stem_wgrad_fused_workspace(&n->convs[0].d);
    n->stem_ws = n->stem_ws_bytes ? n->ws_alloc(n->stem_ws_bytes / 4 + 4) : n->wg_ws;
    n->plan_knobs = g_osi_tuning;
    n->plan_hw_cus = device_cus();
    n->dg_ws_bytes = dgws; n->dg_ws = n->ws_alloc(dgws / 4 + 4);
    n->scratch_floats = maxact;
    for (int i = 0; i < osi_resnet50::NSCR; ++i) n->scratch[i] = n->ws_alloc(maxact);
    *out = n;
    return OSI_OK;
}

void osi_resnet50_destroy(osi_resnet50_t net) { delete net; }
int osi_resnet50_num_tensors(osi_resnet50_t net) { return net ? (int)net->tensors.size() : 0; }
size_t osi_resnet50_param_floats(osi_resnet50_t net) { return net ? net->param_floats : 0; }
int osi_resnet50_num_bn(osi_resnet50_t net) { return net ? (int)net->bns.size() : 0; }
size_t osi_resnet50_buffer_floats(osi_resnet50_t net) { return net ? net->buffer_floats : 0; }
size_t osi_resnet50_workspace_bytes(osi_resnet50_t net) { return net ? net->ws_floats * sizeof(float) : 0; }
int osi_resnet50_num_stages(osi_resnet50_t net) { return net ? net->n_stages : 0; }
int osi_resnet50_geometry(osi_resnet50_t net, int* B, int* H, int* W) {
    OSI_REQUIRE(net);
    if (B) *B = net->B;
    if (H) *H = net->H;
    if (W) *W = net->W;
    return OSI_OK;
}

int osi_resnet50_tensor_info(osi_resnet50_t net, int i, char* name, int name_cap, int* ndim, int* shape, size_t* offset,
                             size_t* numel) {
    OSI_REQUIRE(net && i >= 0 && i < (int)net->tensors.size());
    const Tensor& t = net->tensors[i];
    if (name && name_cap > 0) { std::strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ndim) *ndim = t.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = t.shape[k];
    if (offset) *offset = t.off;
    if (numel) *numel = t.numel;
    return OSI_OK;
}
int osi_resnet50_bn_info(osi_resnet50_t net, int j, char* prefix, int cap, int* C, size_t* rm_offset, size_t* rv_offset) {
    OSI_REQUIRE(net && j >= 0 && j < (int)net->bns.size());
    const BN& b = net->bns[j];
    if (prefix && cap > 0) { std::strncpy(prefix, b.prefix.c_str(), cap - 1); prefix[cap - 1] = 0; }
    if (C) *C = b.C;
    if (rm_offset) *rm_offset = b.rm_off;
    if (rv_offset) *rv_offset = b.rv_off;
    return OSI_OK;
}
int osi_resnet50_stage_grad_range(osi_resnet50_t net, int s, size_t* lo, size_t* hi) {
    OSI_REQUIRE(net && s >= 0 && s < net->n_stages && lo && hi);
    *lo = net->stage_lo[s]; *hi = net->stage_hi[s];
    return OSI_OK;
}

int osi_resnet50_profile(osi_resnet50_t n, int enable) {
    OSI_REQUIRE(n);
    n->prof_on = enable != 0;
    n->prof_timeline = enable == 2;
    n->prof_n = 0;
    return OSI_OK;
}
// Timeline mode (osi_resnet50_profile(net, 2)): the overlapped schedule is kept and every op's completion event carries the stream
// it ran on. Host-synchronising read-out: t_ms[i] = completion time of op i relative to the first recorded event.
int osi_resnet50_timeline_read(osi_resnet50_t n, double* t_ms, int* cls, int* on_side, int cap, int* count) {
    OSI_REQUIRE(n && t_ms && cls && on_side && count && cap > 0);
    *count = 0;
    if (n->prof_n == 0) return OSI_OK;
    for (int i = 0; i < n->prof_n; ++i)
        if (hipEventSynchronize(n->prof_ev[i]) != hipSuccess) return OSI_ERR_LAUNCH;
    const int m = n->prof_n < cap ? n->prof_n : cap;
    for (int i = 0; i < m; ++i) {
        float t = 0.f;
        if (i > 0 && hipEventElapsedTime(&t, n->prof_ev[0], n->prof_ev[i]) != hipSuccess) return OSI_ERR_LAUNCH;
        t_ms[i] = t; cls[i] = n->prof_cls[i]; on_side[i] = n->prof_side[i];
    }
    *count = m;
    n->prof_n = 0;
    return OSI_OK;
}
// Host-synchronising read-out (not a launch function): per-class elapsed milliseconds and op counts since enable/last read.
int osi_resnet50_profile_read(osi_resnet50_t n, double* ms, int* count) {
    OSI_REQUIRE(n && ms && count);
    for (int k = 0; k < OSI_PROF_NCLASS; ++k) { ms[k] = 0; count[k] = 0; }
    if (n->prof_n == 0) return OSI_OK;
    if (hipEventSynchronize(n->prof_ev[n->prof_n - 1]) != hipSuccess) return OSI_ERR_LAUNCH;
    for (int i = 1; i < n->prof_n; ++i) {
        int c = n->prof_cls[i];
        if (c == OSI_PROF_START) continue;  // gap between two executor calls (loss kernel, host work): not attributed
        float t = 0.f;
        if (hipEventElapsedTime(&t, n->prof_ev[i - 1], n->prof_ev[i]) != hipSuccess) return OSI_ERR_LAUNCH;
        ms[c] += t; count[c] += 1;
    }
    n->prof_n = 0;
    return OSI_OK;
}

// conv ci + its BatchNorm statistics. in_bn >= 0: the conv's input is the PRE-BN output of the layer whose BatchNorm is `in_bn`;
// that BatchNorm + ReLU is applied inside the conv's operand loader (osi_conv_fwd_act) — the activation never exists in HBM.
// in_res != NULL (with in_bn): x is conv3's pre-BN output of the previous bottleneck and in_res its identity shortcut — the whole block
// output relu(bn3(x) + in_res) is recomputed in the loader (osi_conv_fwd_act2).
static int conv_bn_fwd(osi_resnet50* n, int ci, const float* params, float* buffers, float* ws, const float* x, const float* w,
                       int training, hipStream_t st, size_t bn_ws_off, int in_bn = -1, const float* in_res = nullptr) {
    Conv& c = n->convs[ci];
    BN& b = n->bns[c.bn];
    const float* isc = in_bn >= 0 ? ws + n->bns[in_bn].scale : nullptr;
    const float* ish = in_bn >= 0 ? ws + n->bns[in_bn].shift : nullptr;
    // 3x3 / stride 1 (conv2 of a bottleneck without a stride): Winograd F(2x2,3x3), 2.25x fewer multiplies (csrc/conv_wino.hip); main stream only
    const bool wino = n->plan_knobs.fwd_wino && isc && !in_res && (n->side == nullptr || st != n->side) && c.u_fw != (size_t)-1;
    if (wino && n->wt_pending) {      // the transformed weights come from the side stream
        if (hipStreamWaitEvent(st, n->ev_wt, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        n->wt_pending = false;
    }
    if (training) {
        // batch statistics come out of the conv epilogue (per row tile), only a tiny per-channel merge follows
        int P = 0, rows = 0;
        if (isc && in_res) OSI_TRY(osi_conv_fwd_act2(&c.d, x, isc, ish, in_res, w, ws + c.y, OSI_TILE_AUTO, ws + bn_ws_off, n->bn_ws_bytes, &P, &rows, st));
        else if (isc && wino) OSI_TRY(osi_conv_fwd_wino_pre(&c.d, x, isc, ish, ws + c.u_fw, ws + c.y, ws + n->wino_ws, n->wino_ws_bytes, ws + bn_ws_off, n->bn_ws_bytes, &P, &rows, st));
        else if (isc) OSI_TRY(osi_conv_fwd_act(&c.d, x, isc, ish, w, ws + c.y, OSI_TILE_AUTO, ws + bn_ws_off, n->bn_ws_bytes, &P, &rows, st));
        else OSI_TRY(osi_conv_fwd_bnstats(&c.d, x, w, ws + c.y, OSI_TILE_AUTO, ws + bn_ws_off, n->bn_ws_bytes, &P, &rows, st));
        OSI_TRY(n->mark(OSI_PROF_CONV_FWD, st));
        // (bit 3, diagnostic build only: no finalize launches from the 4th training forward on — the coefficients of the third step stay
        // in the workspace, so the data, and with it the clock the chip holds, stay realistic: the ceiling of merging / folding these launches)
        if (!((n->dbg_skip & 8) && n->dbg_fwd_count > 3))
        OSI_TRY(osi_bn_finalize_stats(ws + bn_ws_off, n->bn_ws_bytes, P, rows, b.M, b.C, params + b.g_off, params + b.b_off, 1e-5f, 0.1f,
                                      buffers + b.rm_off, buffers + b.rv_off, ws + b.mean, ws + b.invstd, ws + b.scale,
                                      ws + b.shift, st));
    } else {
        if (isc && in_res) OSI_TRY(osi_conv_fwd_act2(&c.d, x, isc, ish, in_res, w, ws + c.y, OSI_TILE_AUTO, nullptr, 0, nullptr, nullptr, st));
        else if (isc && wino) OSI_TRY(osi_conv_fwd_wino_pre(&c.d, x, isc, ish, ws + c.u_fw, ws + c.y, ws + n->wino_ws, n->wino_ws_bytes, nullptr, 0, nullptr, nullptr, st));
        else if (isc) OSI_TRY(osi_conv_fwd_act(&c.d, x, isc, ish, w, ws + c.y, OSI_TILE_AUTO, nullptr, 0, nullptr, nullptr, st));
        else OSI_TRY(osi_conv_fwd(&c.d, x, w, ws + c.y, OSI_TILE_AUTO, st));
        OSI_TRY(n->mark(OSI_PROF_CONV_FWD, st));
        OSI_TRY(osi_bn_eval_coeffs(buffers + b.rm_off, buffers + b.rv_off, params + b.g_off, params + b.b_off, 1e-5f, b.C,
                                   ws + b.scale, ws + b.shift, st));
    }
    OSI_TRY(n->mark(OSI_PROF_BN_FWD, st));
    return OSI_OK;
}

// Input staged from a uint8 [B][H][W][3] batch (+ optional per-image horizontal flip flags) straight into the executor's NHWC4
// input buffer; the following osi_resnet50_forward call passes image = NULL.
int osi_resnet50_stage_input_u8(osi_resnet50_t n, const unsigned char* images_u8_nhwc, const unsigned char* flip, void* workspace,
                                osi_stream_t stream) {
    OSI_REQUIRE(n && images_u8_nhwc && workspace);
    OSI_TRY(osi_u8hwc3_to_nhwc4(images_u8_nhwc, flip, (float*)workspace + n->x4, n->B, n->H, n->W, stream));
    n->staged_ws = workspace;
    return OSI_OK;
}

int osi_resnet50_bind_input_nhwc4(osi_resnet50_t n, const float* x_nhwc4) {
    OSI_REQUIRE(n && x_nhwc4 && ((size_t)x_nhwc4 & 15) == 0);
    n->x4_ext = x_nhwc4;
    return OSI_OK;
}

// avgpool -> fc -> logits, copies to the caller's tensors
static int head_fwd(osi_resnet50* n, const float* params, float* ws, float* logits, float* features, hipStream_t st) {
    const float* last = ws + n->blocks.back().out;
    OSI_TRY(osi_avgpool_fwd(last, ws + n->pooled, n->B, n->Hf * n->Wf, 2048, st));
    const Tensor& fw = n->tensors[n->t_fc_w]; const Tensor& fb = n->tensors[n->t_fc_b]; const Tensor& lw = n->tensors[n->t_lg_w];
    OSI_TRY(osi_linear_fwd(ws + n->pooled, params + fw.off, params + fb.off, ws + n->feat, n->B, 2048, n->F, st));
    const float* lb = n->t_lg_b >= 0 ? params + n->tensors[n->t_lg_b].off : nullptr;
    OSI_TRY(osi_linear_fwd(ws + n->feat, params + lw.off, lb, ws + n->logits_ws, n->B, n->F, n->O, st));
    if (hipMemcpyAsync(features, ws + n->feat, (size_t)n->B * n->F * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return OSI_ERR_LAUNCH;
    if (hipMemcpyAsync(logits, ws + n->logits_ws, (size_t)n->B * n->O * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return OSI_ERR_LAUNCH;
    return OSI_OK;
}

// Inference forward (training == 0; validate() / get_arrays(), reference train.py:142-234: model.eval() under no_grad). In eval mode
// every BatchNorm's scale / shift exist before its convolution is launched, so nothing of the training topology's bookkeeping is
// needed: ONE launch computes the coefficients of all 53 BatchNorms, and every convolution applies its own BatchNorm in its epilogue —
// conv1 / conv2 write relu(bn(conv)), the projection shortcut writes bn(conv), conv3 writes the block output relu(bn3(conv3) + shortcut)
// (osi_conv_fwd_epilogue / osi_conv_fwd_wino_epilogue_pre). No pre-BN tensor, no block-output pass, no ReLU bitmask, no fused-loader
// activation on the consumer side (every consumer reads a finished activation with its plain loader): 53 + 16 + 53 launches fewer, the 16
// HBM-bound block-output passes and the loader-side activation arithmetic gone. Values: the same fmas on the same accumulators as the
// training topology on running statistics (option eval_fused = 0) — bit-identical outputs.
// Buffers: conv1 / conv2 / shortcut activations live where the training forward keeps those layers' pre-BN tensors, the block outputs
// where it keeps them; the workspace then holds no backward state (fwd_done and any_fwd are cleared).
static int forward_eval_fused(osi_resnet50* n, const float* params, const float* buffers, float* ws, const float* x4, float* logits,
                              float* features, hipStream_t st) {
    {
        osi_bn_eval_layer tab[OSI_BN_MULTI_MAX];
        const int nb = (int)n->bns.size();
        if (nb > OSI_BN_MULTI_MAX) return OSI_ERR_STATE;
        for (int j = 0; j < nb; ++j) {
            const BN& b = n->bns[j];
            tab[j] = osi_bn_eval_layer{buffers + b.rm_off, buffers + b.rv_off, params + b.g_off, params + b.b_off, ws + b.scale, ws + b.shift, b.C};
        }
        OSI_TRY(osi_bn_eval_coeffs_multi(tab, nb, 1e-5f, st));
        OSI_TRY(n->mark(OSI_PROF_BN_FWD, st));
    }
    Conv& c0 = n->convs[0];
    OSI_TRY(osi_conv_fwd(&c0.d, x4, ws + n->wpack, ws + c0.y, OSI_TILE_AUTO, st));
    OSI_TRY(n->mark(OSI_PROF_CONV_FWD, st));
    BN& b0 = n->bns[c0.bn];
    OSI_TRY(osi_bn_relu_maxpool_fwd(ws + c0.y, ws + b0.scale, ws + b0.shift, ws + n->a_pool, ws + n->pool_idx, n->B, n->Hs, n->Ws, 64, st));
    OSI_TRY(n->mark(OSI_PROF_BN_FWD, st));
    auto epi = [&](int ci, const float* res, int relu) {
        const BN& b = n->bns[n->convs[ci].bn];
        return osi_conv_epilogue{ws + b.scale, ws + b.shift, res, relu};
    };
    for (Block& k : n->blocks) {
        const float* x = ws + k.x_in;
        Conv &c1 = n->convs[k.c1], &c2 = n->convs[k.c2], &c3 = n->convs[k.c3];
        const bool fork = k.ds >= 0 && n->fwd_fork && n->async_wgrad();
        if (k.ds >= 0) {            // the projection shortcut only depends on the block input: beside the main branch where a side stream exists
            Conv& cd = n->convs[k.ds];
            hipStream_t ds_st = st;
            if (fork) {
                if (hipEventRecord(n->ev_fork, st) != hipSuccess) return OSI_ERR_LAUNCH;
                if (hipStreamWaitEvent(n->side, n->ev_fork, 0) != hipSuccess) return OSI_ERR_LAUNCH;
                ds_st = n->side;
            }
            const osi_conv_epilogue e = epi(k.ds, nullptr, 0);
            OSI_TRY(osi_conv_fwd_epilogue(&cd.d, x, params + cd.w_off, ws + cd.y, &e, ws + (fork ? n->bn_ws2 : n->bn_ws), n->bn_ws_bytes, ds_st));
            OSI_TRY(n->mark(OSI_PROF_CONV_FWD, ds_st));
            if (fork && hipEventRecord(n->ev_join, n->side) != hipSuccess) return OSI_ERR_LAUNCH;
        }
        {
            const osi_conv_epilogue e = epi(k.c1, nullptr, 1);
            OSI_TRY(osi_conv_fwd_epilogue(&c1.d, x, params + c1.w_off, ws + c1.y, &e, ws + n->bn_ws, n->bn_ws_bytes, st));
            OSI_TRY(n->mark(OSI_PROF_CONV_FWD, st));
        }
        {
            const osi_conv_epilogue e = epi(k.c2, nullptr, 1);
            if (n->plan_knobs.fwd_wino && c2.u_fw != (size_t)-1) {     // 3x3 / stride 1: Winograd F(2x2,3x3)
                if (n->wt_pending) {      // the transformed weights come from the side stream
                    if (hipStreamWaitEvent(st, n->ev_wt, 0) != hipSuccess) return OSI_ERR_LAUNCH;
                    n->wt_pending = false;
                }
                OSI_TRY(osi_conv_fwd_wino_epilogue_pre(&c2.d, ws + c1.y, ws + c2.u_fw, ws + c2.y, &e, ws + n->wino_ws, n->wino_ws_bytes, st));
            } else {
                OSI_TRY(osi_conv_fwd_epilogue(&c2.d, ws + c1.y, params + c2.w_off, ws + c2.y, &e, ws + n->bn_ws, n->bn_ws_bytes, st));
            }
            OSI_TRY(n->mark(OSI_PROF_CONV_FWD, st));
        }
        if (fork && hipStreamWaitEvent(st, n->ev_join, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        {
            const osi_conv_epilogue e = epi(k.c3, k.ds >= 0 ? ws + n->convs[k.ds].y : x, 1);
            OSI_TRY(osi_conv_fwd_epilogue(&c3.d, ws + c2.y, params + c3.w_off, ws + c3.a, &e, ws + n->bn_ws, n->bn_ws_bytes, st));
            OSI_TRY(n->mark(OSI_PROF_CONV_FWD, st));
        }
    }
    OSI_TRY(head_fwd(n, params, ws, logits, features, st));
    n->any_fwd = false;           // no pre-BN tensors, bitmasks or arg-max decisions of a training forward remain (osi_resnet50_debug_gate)
    OSI_TRY(n->mark(OSI_PROF_OTHER, st));
    return OSI_OK;
}

int osi_resnet50_forward(osi_resnet50_t n, const float* params, float* buffers, long long* nbt, const float* image,
                         void* workspace, float* logits, float* features, int training, osi_stream_t stream) {
    OSI_REQUIRE(n && params && buffers && workspace && logits && features);
    OSI_REQUIRE(!training || nbt);
    if (!n->plan_unchanged()) return OSI_ERR_STATE;   // a plan-relevant knob changed after create: the workspace no longer fits the plans
    const float* ext = n->x4_ext;
    n->x4_ext = nullptr;
    if (!image && !ext && n->staged_ws != workspace) return OSI_ERR_STATE;   // image = NULL needs a staged or bound input
    n->staged_ws = nullptr;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const float* x4 = (!image && ext) ? ext : ws + n->x4;
    n->x4_cur = x4;
    n->fwd_done = false;
#ifdef OSI_DIAG
    if (training) ++n->dbg_fwd_count;
#endif
    if (training && n->overlap && (!n->prof_on || n->prof_timeline)) OSI_TRY(n->ensure_side());
    OSI_TRY(n->mark(OSI_PROF_START, st));
    // Winograd weight transforms of every 3x3 stride-1 layer, both directions: the weights are the same for this forward and its backward.
    // On the side stream beside the stem (26 launches of 4 - 16 us that used to sit in front of their convolutions); the first Winograd
    // convolution waits for them.
    if (n->plan_knobs.fwd_wino || (training && n->plan_knobs.dgrad_wino)) {
        // (training forwards only: an inference forward has 13 transforms and nothing but the stem beside them — the fork / join costs more
        // than they do: 7.96 in line vs 8.00 ms aside per batch of 128, profiles/NOTES_r06.md)
        const bool aside = training && n->async_wgrad() && n->wt_aside;
        hipStream_t wt = aside ? n->side : st;
        if (aside) {
            if (hipEventRecord(n->ev_fork, st) != hipSuccess) return OSI_ERR_LAUNCH;
            if (hipStreamWaitEvent(n->side, n->ev_fork, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        }
        for (auto& c : n->convs)
            if (n->plan_knobs.fwd_wino && c.u_fw != (size_t)-1)
                OSI_TRY(osi_conv_wino_transform_weights(&c.d, params + c.w_off, 0, ws + c.u_fw, osi_conv_wino_weights_bytes(&c.d), wt));
        OSI_TRY(n->mark(OSI_PROF_CONV_FWD, wt));
        if (training && n->plan_knobs.dgrad_wino) {
            for (auto& c : n->convs)
                if (c.u_bw != (size_t)-1)
                    OSI_TRY(osi_conv_wino_transform_weights(&c.d, params + c.w_off, 1, ws + c.u_bw, osi_conv_wino_weights_bytes(&c.d), wt));
            OSI_TRY(n->mark(OSI_PROF_CONV_DGRAD, wt));
        }
        if (aside) {
            if (hipEventRecord(n->ev_wt, n->side) != hipSuccess) return OSI_ERR_LAUNCH;
            n->wt_pending = true;
        }
    }
    // stem
    if (image) OSI_TRY(osi_nchw3_to_nhwc4(image, ws + n->x4, n->B, n->H, n->W, st));
    Conv& c0 = n->convs[0];
    OSI_TRY(osi_stem_weight_pack(params + c0.w_off, ws + n->wpack, 64, st));
    OSI_TRY(n->mark(OSI_PROF_OTHER, st));
    if (!training && n->eval_fused) return forward_eval_fused(n, params, buffers, ws, x4, logits, features, st);
    OSI_TRY(conv_bn_fwd(n, 0, params, buffers, ws, x4, ws + n->wpack, training, st, n->bn_ws));
    BN& b0 = n->bns[c0.bn];
    // bn1 + relu + maxpool in one pass: the 112x112x64 post-ReLU tensor is never materialised
    OSI_TRY(osi_bn_relu_maxpool_fwd(ws + c0.y, ws + b0.scale, ws + b0.shift, ws + n->a_pool, ws + n->pool_idx, n->B, n->Hs, n->Ws, 64, st));
    OSI_TRY(n->mark(OSI_PROF_BN_FWD, st));
    // bottleneck blocks. Only the block outputs (residual sums) are materialised: conv2 / conv3 read the pre-BN output of the conv
    // before them and apply its BatchNorm + ReLU in their operand loader; the projection shortcut's BatchNorm is applied inside the
    // block-output kernel. Per block: 3 (4) convs + one block-output pass instead of 3 (4) convs + 3 (4) apply passes.
    // The block-output pass itself is HBM-bound and sits between two matrix-bound convolutions. Option "fwd_recompute" takes it off the
    // critical path for blocks with an identity shortcut: conv1 of the next block recomputes relu(bn3(y3) + x) in its loader
    // (osi_conv_fwd_act2) and the pass that materialises the tensor (for the next shortcut, the projection conv and the backward) runs
    // beside it on the side stream. Default off (no measured gain, see the member's comment).
    const int nb = (int)n->blocks.size();
    bool deferred = false;          // the previous block's output pass has not been enqueued yet
    for (int bi = 0; bi < nb; ++bi) {
        Block& k = n->blocks[bi];
        const float* x = ws + k.x_in;
        const bool async = n->async_wgrad();
        hipStream_t sd = async ? n->side : st;      // stream of the work that runs beside the main branch
        bool forked = false;
        if ((deferred || (k.ds >= 0 && n->fwd_fork)) && async) {
            if (hipEventRecord(n->ev_fork, st) != hipSuccess) return OSI_ERR_LAUNCH;
            if (hipStreamWaitEvent(n->side, n->ev_fork, 0) != hipSuccess) return OSI_ERR_LAUNCH;
            forked = true;
        }
        const Block* pk = bi > 0 ? &n->blocks[bi - 1] : nullptr;
        if (deferred && !(n->dbg_skip & 4)) {   // materialise the previous block's output (this block's shortcut / projection input)
            Conv& c3p = n->convs[pk->c3];
            BN& b3p = n->bns[c3p.bn];
            OSI_TRY(osi_bn_apply_relu_mask(ws + c3p.y, ws + pk->x_in, ws + b3p.scale, ws + b3p.shift, ws + c3p.a, ws + c3p.mask, b3p.M,
                                           b3p.C, sd));
            OSI_TRY(n->mark(OSI_PROF_BN_FWD, sd));
        }
        if (k.ds >= 0) {            // the projection shortcut only depends on the block input: beside the main branch (own BN scratch)
            Conv& c = n->convs[k.ds];
            hipStream_t ds_st = (n->fwd_fork || deferred) ? sd : st;   // behind a deferred pass it must follow it on that stream
            OSI_TRY(conv_bn_fwd(n, k.ds, params, buffers, ws, x, params + c.w_off, training, ds_st, ds_st != st ? n->bn_ws2 : n->bn_ws));
        }
        if (forked && hipEventRecord(n->ev_join, n->side) != hipSuccess) return OSI_ERR_LAUNCH;
        Conv &c1 = n->convs[k.c1], &c2 = n->convs[k.c2], &c3 = n->convs[k.c3];
        if (deferred) {
            Conv& c3p = n->convs[pk->c3];
            OSI_TRY(conv_bn_fwd(n, k.c1, params, buffers, ws, ws + c3p.y, params + c1.w_off, training, st, n->bn_ws, c3p.bn, ws + pk->x_in));
        } else {
            OSI_TRY(conv_bn_fwd(n, k.c1, params, buffers, ws, x, params + c1.w_off, training, st, n->bn_ws));
        }
        OSI_TRY(conv_bn_fwd(n, k.c2, params, buffers, ws, ws + c1.y, params + c2.w_off, training, st, n->bn_ws, c1.bn));
        OSI_TRY(conv_bn_fwd(n, k.c3, params, buffers, ws, ws + c2.y, params + c3.w_off, training, st, n->bn_ws, c2.bn));
        if (forked && hipStreamWaitEvent(st, n->ev_join, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        BN& b3 = n->bns[c3.bn];
        deferred = n->fwd_recompute && k.ds < 0 && bi + 1 < nb;
        if (deferred) continue;     // the next iteration enqueues this block's output pass beside its conv1
        if (n->dbg_skip & 2) continue;   // timing experiment: no block-output pass (the next block reads stale data)
        if (k.ds >= 0) {
            Conv& cd = n->convs[k.ds];
            BN& bd = n->bns[cd.bn];
            OSI_TRY(osi_bn_apply_relu_mask2(ws + c3.y, ws + b3.scale, ws + b3.shift, ws + cd.y, ws + bd.scale, ws + bd.shift, ws + c3.a,
                                            ws + c3.mask, b3.M, b3.C, st));
        } else {
            OSI_TRY(osi_bn_apply_relu_mask(ws + c3.y, x, ws + b3.scale, ws + b3.shift, ws + c3.a, ws + c3.mask, b3.M, b3.C, st));
        }
        OSI_TRY(n->mark(OSI_PROF_BN_FWD, st));
    }
    OSI_TRY(head_fwd(n, params, ws, logits, features, st));
    n->any_fwd = true;
    if (training) {
        OSI_TRY(osi_i64_add(nbt, (int)n->bns.size(), 1, st));
        n->fwd_done = true;
        n->next_stage = 0;
    }
    OSI_TRY(n->mark(OSI_PROF_OTHER, st));
    return OSI_OK;
}

// weight gradient of conv `ci` from dy (scratch buffer index gi): on the side stream when overlap is on
// in_bn >= 0: conv_in is the PRE-BN output of the layer with BatchNorm `in_bn`; its BN + ReLU is applied in the loader
static int wgrad_launch(osi_resnet50* n, int ci, float* grads, float* ws, int gi, const float* conv_in, hipStream_t st, int in_bn,
                        bool async) {
    Conv& c = n->convs[ci];
    const float* dy = ws + n->scratch[gi];
    hipStream_t ws_st = st;
    if (async) {
        if (hipEventRecord(n->ev_fork, st) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipStreamWaitEvent(n->side, n->ev_fork, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        ws_st = n->side;
    }
    if (ci == 0) {
        if (osi_stem_wgrad_direct_workspace(&c.d) > 0) {      // direct form: the parameter-layout gradient in one go
            OSI_TRY(osi_stem_wgrad_direct(&c.d, dy, conv_in, grads + c.w_off, ws + n->wg_ws, n->wg_ws_bytes, ws_st));
        } else {
            OSI_TRY(osi_conv_wgrad(&c.d, dy, conv_in, ws + n->gpack, ws + n->wg_ws, n->wg_ws_bytes, ws_st));
            OSI_TRY(osi_stem_grad_unpack(ws + n->gpack, grads + c.w_off, 64, ws_st));
        }
    } else if (in_bn >= 0 && n->plan_knobs.wgrad_wino && osi_conv_wgrad_wino_workspace(&c.d) > 0) {
        // 3x3 / stride 1 (conv2 of a bottleneck without a stride): Winograd F(3x3, 2x2), the sum over tiles in the transformed domain
        OSI_TRY(osi_conv_wgrad_wino(&c.d, dy, conv_in, ws + n->bns[in_bn].scale, ws + n->bns[in_bn].shift, grads + c.w_off, ws + n->wg_ws,
                                    n->wg_ws_bytes, ws_st));
    } else if (in_bn >= 0) {
        OSI_TRY(osi_conv_wgrad_act(&c.d, dy, conv_in, ws + n->bns[in_bn].scale, ws + n->bns[in_bn].shift, grads + c.w_off, ws + n->wg_ws,
                                   n->wg_ws_bytes, ws_st));
    } else {
        OSI_TRY(osi_conv_wgrad(&c.d, dy, conv_in, grads + c.w_off, ws + n->wg_ws, n->wg_ws_bytes, ws_st));
    }
    if (async) {
        if (hipEventRecord(n->buf_ev[gi], n->side) != hipSuccess) return OSI_ERR_LAUNCH;
        n->buf_pending[gi] = true;
        n->side_dirty = true;
    }
    OSI_TRY(n->mark(OSI_PROF_CONV_WGRAD, ws_st));
    return OSI_OK;
}

// staggered schedule: issue the weight gradient that was held back, behind everything enqueued on `st` so far
static int flush_wgrad(osi_resnet50* n, hipStream_t st) {
    if (!n->pend.on) return OSI_OK;
    osi_resnet50::PendingW p = n->pend;
    n->pend.on = false;
    OSI_TRY(wgrad_launch(n, p.ci, p.grads, p.ws, p.gi, p.conv_in, st, p.in_bn, true));
    if (hipEventRecord(n->ev_wdone, n->side) != hipSuccess) return OSI_ERR_LAUNCH;
    n->w_inflight = true;
    return OSI_OK;
}

// staggered schedule: an input gradient may only start once the weight gradient issued before it has finished
static int before_dgrad(osi_resnet50* n, hipStream_t st) {
    if (!n->w_inflight) return OSI_OK;
    if (hipStreamWaitEvent(st, n->ev_wdone, 0) != hipSuccess) return OSI_ERR_LAUNCH;
    n->w_inflight = false;
    return OSI_OK;
}

static int wgrad(osi_resnet50* n, int ci, float* grads, float* ws, int gi, const float* conv_in, hipStream_t st, int in_bn = -1) {
    const bool async = n->async_wgrad();
    if (!(async && n->stagger)) return wgrad_launch(n, ci, grads, ws, gi, conv_in, st, in_bn, async);
    OSI_TRY(flush_wgrad(n, st));       // two weight gradients with no input gradient between them: the older one goes now
    n->pend.on = true; n->pend.ci = ci; n->pend.gi = gi; n->pend.in_bn = in_bn; n->pend.conv_in = conv_in; n->pend.grads = grads; n->pend.ws = ws;
    // every caller issues the input gradient that flushes this launch BEFORE giving the dy buffer back (the stem, which has no input
    // gradient, is flushed at the end of the call, before the join), so the buffer's reader event exists by the time it can be taken
    return OSI_OK;
}

// plain input gradient (no fused epilogue) with the staggered-schedule hooks
static int dgrad_plain(osi_resnet50* n, const osi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, hipStream_t st) {
    OSI_TRY(before_dgrad(n, st));
    OSI_TRY(osi_conv_dgrad(d, dy, w, dx, accumulate, OSI_TILE_AUTO, st));
    OSI_TRY(n->mark(OSI_PROF_CONV_DGRAD, st));
    return flush_wgrad(n, st);
}

// backward of conv+BN(+ReLU mask): dout (in scratch buffer gi) -> dy in place, then wgrad; returns with dy still in the buffer
static int bn_conv_wgrad(osi_resnet50* n, int ci, const float* params, float* grads, float* ws, int gi,
                         float* gmasked, const float* conv_in, hipStream_t st) {
    Conv& c = n->convs[ci];
    BN& b = n->bns[c.bn];
    float* g = ws + n->scratch[gi];
    OSI_TRY(osi_bn_backward_relu_mask(g, ws + c.mask, ws + c.y, ws + b.mean, ws + b.invstd, params + b.g_off, g, gmasked,
                                      grads + b.g_off, grads + b.b_off, b.M, b.C, ws + n->bn_ws, n->bn_ws_bytes, st));
    OSI_TRY(n->mark(OSI_PROF_BN_BWD, st));
    return wgrad(n, ci, grads, ws, gi, conv_in, st);
}

// BatchNorm backward of conv `ci` from a gradient buffer that the producing dgrad epilogue already ReLU-masked, with the
// reductions waiting in dg_ws (column `which` = 0: main branch, 1: downsample branch). dy goes to buffer `dyi` (may equal gi).
static int bn_bwd_fused(osi_resnet50* n, int ci, const float* params, float* grads, float* ws, int gi, int dyi, int which,
                        hipStream_t st) {
    Conv& c = n->convs[ci];
    BN& b = n->bns[c.bn];
    const float* psum_g = ws + n->dg_ws;
    const float* psum_gx = psum_g + (size_t)(1 + which) * n->fused_P * b.C;
    if (n->dbg_skip & 1) {   // timing experiment: reductions only, the consumers read whatever the dy buffer holds
        OSI_TRY(osi_bn_backward_reduce(psum_g, psum_gx, n->fused_P, grads + b.g_off, grads + b.b_off, b.M, b.C, ws + n->bn_ws, n->bn_ws_bytes, st));
        return n->mark(OSI_PROF_BN_BWD, st);
    }
    OSI_TRY(osi_bn_backward_fused(ws + n->scratch[gi], ws + c.y, ws + b.mean, ws + b.invstd, params + b.g_off, psum_g, psum_gx,
                                  n->fused_P, ws + n->scratch[dyi], grads + b.g_off, grads + b.b_off, b.M, b.C, ws + n->bn_ws,
                                  n->bn_ws_bytes, st));
    OSI_TRY(n->mark(OSI_PROF_BN_BWD, st));
    return OSI_OK;
}

// dgrad of conv `ci` (dy in buffer dyi) into buffer dxi, adding buffer addi (-1: none), with the epilogue fused for the layer
// that produced this conv's input: its ReLU bitmask and the BatchNorm reductions of conv `pc` (and `pd`, the downsample twin).
static int dgrad_fused(osi_resnet50* n, int ci, const float* params, float* ws, int dyi, int dxi, int addi, int pc, int pd,
                       hipStream_t st, bool add_even = false) {
    Conv& c = n->convs[ci];
    Conv& p0 = n->convs[pc];
    BN& b0 = n->bns[p0.bn];
    osi_dgrad_fusion f{};
    if (p0.mask != (size_t)-1) f.relu_mask = ws + p0.mask;           // block output: stored ReLU bitmask
    else { f.scale0 = ws + b0.scale; f.shift0 = ws + b0.shift; }     // in-block activation: gate recomputed from y0
    f.y0 = ws + p0.y; f.mean0 = ws + b0.mean; f.invstd0 = ws + b0.invstd;
    if (pd >= 0) {
        Conv& p1 = n->convs[pd];
        BN& b1 = n->bns[p1.bn];
        f.y1 = ws + p1.y; f.mean1 = ws + b1.mean; f.invstd1 = ws + b1.invstd;
    }
    f.partials = ws + n->dg_ws; f.partials_bytes = n->dg_ws_bytes;
    f.addend_stride = add_even ? 2 : 1;
    int P = 0;
    OSI_TRY(before_dgrad(n, st));
    // the in-block 3x3 / stride 1 input gradients (gate recomputed, one consumer, no addend) take the Winograd form
    if (n->wt_pending) {          // (forward Winograd off: nobody has waited for the side-stream weight transforms yet)
        if (hipStreamWaitEvent(st, n->ev_wt, 0) != hipSuccess) return OSI_ERR_LAUNCH;
        n->wt_pending = false;
    }
    if (n->plan_knobs.dgrad_wino && f.scale0 && pd < 0 && addi < 0 && c.u_bw != (size_t)-1)
        OSI_TRY(osi_conv_dgrad_fused_wino_pre(&c.d, ws + n->scratch[dyi], ws + c.u_bw, ws + n->scratch[dxi], &f, ws + n->wino_ws,
                                              n->wino_ws_bytes, &P, st));
    else
    OSI_TRY(osi_conv_dgrad_fused(&c.d, ws + n->scratch[dyi], params + c.w_off, ws + n->scratch[dxi],
                                 addi >= 0 ? ws + n->scratch[addi] : nullptr, &f, OSI_TILE_AUTO, &P, st));
    n->fused_P = P;
    OSI_TRY(n->mark(OSI_PROF_CONV_DGRAD, st));
    return flush_wgrad(n, st);
}

// One bottleneck block of the backward pass. On entry n->cur_grad holds the gradient w.r.t. the block output: raw (stage entry
// from the average pool) or, when n->go_fused, already masked by the block's final ReLU with the bn3 / downsample-BN reductions
// in dg_ws (left there by the conv1 dgrad epilogue of the block above).
static int block_backward(osi_resnet50* n, int bi, const float* params, float* grads, float* ws, hipStream_t st) {
    Block& k = n->blocks[bi];
    auto S = [&](int i) { return ws + n->scratch[i]; };
    Conv &c1 = n->convs[k.c1], &c2 = n->convs[k.c2], &c3 = n->convs[k.c3];
    const float* x = ws + k.x_in;
    const bool has_ds = k.ds >= 0;
    // A stride-2 1x1 shortcut reaches only the even-even pixels of the block input: its input gradient writes just those (a quarter
    // of the tensor, no zero fill) and conv1's input gradient, which completes the sum in place, reads the addend only there.
    // bi == 0 keeps the dense form (pool mode); so does the wide-tile A/B switch.
    const bool ds_sparse = has_ds && bi > 0 && n->ds_sparse && n->convs[k.ds].d.stride == 2 && n->convs[k.ds].d.R == 1;
    int go = n->cur_grad;
    int d3 = -1, dxbase = -1;
    if (n->go_fused) {
        if (has_ds) {
            Conv& cd = n->convs[k.ds];
            int t1 = n->take(st);
            if (t1 < 0) return t1;
            OSI_TRY(bn_bwd_fused(n, k.ds, params, grads, ws, go, t1, 1, st));
            OSI_TRY(wgrad(n, k.ds, grads, ws, t1, x, st));
            dxbase = n->take(st);
            if (dxbase < 0) return dxbase;
            OSI_TRY(dgrad_plain(n, &cd.d, S(t1), params + cd.w_off, S(dxbase), ds_sparse ? 2 : 0, st));
            n->give(t1);
        }
        d3 = n->take(st);
        if (d3 < 0) return d3;
        OSI_TRY(bn_bwd_fused(n, k.c3, params, grads, ws, go, d3, 0, st));
        if (has_ds) n->give(go);
        else dxbase = go;          // identity skip: the masked gradient itself continues to the block input
    } else {
        dxbase = n->take(st);
        if (dxbase < 0) return dxbase;
        if (has_ds) {
            Conv& cd = n->convs[k.ds];
            BN& bd = n->bns[cd.bn];
            int t1 = n->take(st);
            if (t1 < 0) return t1;
            OSI_TRY(osi_bn_backward_relu_mask(S(go), ws + c3.mask, ws + cd.y, ws + bd.mean, ws + bd.invstd, params + bd.g_off,
                                              S(t1), nullptr, grads + bd.g_off, grads + bd.b_off, bd.M, bd.C, ws + n->bn_ws,
                                              n->bn_ws_bytes, st));
            OSI_TRY(n->mark(OSI_PROF_BN_BWD, st));
            OSI_TRY(wgrad(n, k.ds, grads, ws, t1, x, st));
            OSI_TRY(dgrad_plain(n, &cd.d, S(t1), params + cd.w_off, S(dxbase), ds_sparse ? 2 : 0, st));
            n->give(t1);
        }
        BN& b3 = n->bns[c3.bn];
        OSI_TRY(osi_bn_backward_relu_mask(S(go), ws + c3.mask, ws + c3.y, ws + b3.mean, ws + b3.invstd, params + b3.g_off, S(go),
                                          has_ds ? nullptr : S(dxbase), grads + b3.g_off, grads + b3.b_off, b3.M, b3.C,
                                          ws + n->bn_ws, n->bn_ws_bytes, st));
        OSI_TRY(n->mark(OSI_PROF_BN_BWD, st));
        d3 = go;
    }
    // conv3 -> (mask a2, bn2) -> conv2 -> (mask a1, bn1) -> conv1
    OSI_TRY(wgrad(n, k.c3, grads, ws, d3, ws + c2.y, st, c2.bn));
    int t2 = n->take(st);
    if (t2 < 0) return t2;
    OSI_TRY(dgrad_fused(n, k.c3, params, ws, d3, t2, -1, k.c2, -1, st));
    n->give(d3);
    OSI_TRY(bn_bwd_fused(n, k.c2, params, grads, ws, t2, t2, 0, st));
    OSI_TRY(wgrad(n, k.c2, grads, ws, t2, ws + c1.y, st, c1.bn));
    int t3 = n->take(st);
    if (t3 < 0) return t3;
    OSI_TRY(dgrad_fused(n, k.c2, params, ws, t2, t3, -1, k.c1, -1, st));
    n->give(t2);
    OSI_TRY(bn_bwd_fused(n, k.c1, params, grads, ws, t3, t3, 0, st));
    OSI_TRY(wgrad(n, k.c1, grads, ws, t3, x, st));
    if (bi > 0) {
        // the block input is the previous block's output: fuse that block's final ReLU mask and its bn3 (+ downsample BN) reductions
        Block& pk = n->blocks[bi - 1];
        int dxn = has_ds ? dxbase : n->take(st);   // downsample case: add in place (each lane reads then writes its own element)
        if (dxn < 0) return dxn;
        OSI_TRY(dgrad_fused(n, k.c1, params, ws, t3, dxn, dxbase, pk.c3, pk.ds, st, ds_sparse));
        if (dxn != dxbase) n->give(dxbase);
        n->cur_grad = dxn;
        n->go_fused = true;
    } else {
        // first block: its input is the stem's max-pooled activation. With the fused stem tail the epilogue of this LAST input gradient
        // (it completes the gradient w.r.t. the pooled activation) also emits bn1's backward reductions through the arg-max bytes, so the
        // stem needs no reduction pass over its 112 x 112 tensor (pool mode of osi_conv_dgrad_fused)
        Conv& c0 = n->convs[0];
        n->stem_stats_P = 0;
        if (n->stem_fused && n->stem_pool_stats && n->stem_ws_bytes > 0) {
            BN& b0 = n->bns[c0.bn];
            osi_dgrad_fusion f{};
            f.y0 = ws + c0.y; f.mean0 = ws + b0.mean; f.invstd0 = ws + b0.invstd;
            f.partials = ws + n->dg_ws; f.partials_bytes = n->dg_ws_bytes;
            f.pool_idx = ws + n->pool_idx; f.pool_H = n->Hs; f.pool_W = n->Ws;
            int P = 0;
            OSI_TRY(before_dgrad(n, st));
            OSI_TRY(osi_conv_dgrad_fused(&c1.d, S(t3), params + c1.w_off, S(dxbase), S(dxbase), &f, OSI_TILE_AUTO, &P, st));
            n->stem_stats_P = P;
            OSI_TRY(n->mark(OSI_PROF_CONV_DGRAD, st));
            OSI_TRY(flush_wgrad(n, st));
        } else {
            OSI_TRY(dgrad_plain(n, &c1.d, S(t3), params + c1.w_off, S(dxbase), 1, st));
        }
        n->cur_grad = dxbase;
        n->go_fused = false;
    }
    n->give(t3);
    return OSI_OK;
}

int osi_resnet50_backward(osi_resnet50_t n, const float* params, float* grads, void* workspace, const float* dlogits,
                          const float* dfeatures, int stage_lo, int stage_hi, osi_stream_t stream) {
    OSI_REQUIRE(n && params && grads && workspace);
    OSI_REQUIRE(stage_lo >= 0 && stage_lo < stage_hi && stage_hi <= n->n_stages);
    if (!n->fwd_done || stage_lo != n->next_stage || !n->plan_unchanged()) return OSI_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    auto S = [&](int i) { return ws + n->scratch[i]; };
    if (n->overlap && (!n->prof_on || n->prof_timeline)) OSI_TRY(n->ensure_side());
    OSI_TRY(n->mark(OSI_PROF_START, st));

    for (int stage = stage_lo; stage < stage_hi; ++stage) {
        if (stage == 0) {
            OSI_REQUIRE(dlogits);
            n->free_list.clear();
            for (int i = 0; i < osi_resnet50::NSCR; ++i) { n->free_list.push_back(i); n->buf_pending[i] = false; }
            const Tensor& fw = n->tensors[n->t_fc_w]; const Tensor& fb = n->tensors[n->t_fc_b]; const Tensor& lw = n->tensors[n->t_lg_w];
            float* dfeat = ws + n->dfeat;
            int acc = 0;
            if (dfeatures) {
                if (hipMemcpyAsync(dfeat, dfeatures, (size_t)n->B * n->F * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return OSI_ERR_LAUNCH;
                acc = 1;
            }
            float* dlb = n->t_lg_b >= 0 ? grads + n->tensors[n->t_lg_b].off : nullptr;
            OSI_TRY(osi_linear_bwd(dlogits, ws + n->feat, params + lw.off, dfeat, acc, grads + lw.off, dlb, n->B, n->F, n->O, st));
            OSI_TRY(osi_linear_bwd(dfeat, ws + n->pooled, params + fw.off, ws + n->dpooled, 0, grads + fw.off, grads + fb.off, n->B,
                                   2048, n->F, st));
            int g = n->take(st);
            if (g < 0) return g;
            OSI_TRY(osi_avgpool_bwd(ws + n->dpooled, S(g), n->B, n->Hf * n->Wf, 2048, st));
            OSI_TRY(n->mark(OSI_PROF_OTHER, st));
            n->cur_grad = g;
            n->go_fused = false;
        }
        for (int bi = (int)n->blocks.size() - 1; bi >= 0; --bi) {
            Block& k = n->blocks[bi];
            if (k.stage != stage) continue;
            OSI_TRY(block_backward(n, bi, params, grads, ws, st));
        }
        if (stage == n->n_stages - 1) {
            // maxpool + stem
            Conv& c0 = n->convs[0];
            int go = n->cur_grad;
            int t = n->take(st);
            if (t < 0) return t;
            BN& b0 = n->bns[c0.bn];
            const float* x4c = n->x4_cur ? n->x4_cur : ws + n->x4;
            if (n->stem_fused && n->stem_ws_bytes > 0) {      // its own slab, sized at create: never the side stream's wg_ws
                // reductions of bn1's backward (dgamma, dbeta) on the main stream, then the weight gradient with the max-pool scatter,
                // ReLU gate and BatchNorm backward applied inside its operand loader: the 112x112x64 gradient is never written
                n->give(t);
                if (n->stem_stats_P > 0) {   // the reductions arrived with the last input gradient: two tiny merge launches
                    const float* psum_g = ws + n->dg_ws;
                    OSI_TRY(osi_bn_backward_reduce(psum_g, psum_g + (size_t)n->stem_stats_P * 64, n->stem_stats_P, grads + b0.g_off,
                                                   grads + b0.b_off, n->B * n->Hs * n->Ws, 64, ws + n->bn_ws, n->bn_ws_bytes, st));
                    n->stem_stats_P = 0;
                } else {
                    OSI_TRY(osi_bn_relu_maxpool_bwd(S(go), ws + n->pool_idx, ws + c0.y, ws + b0.mean, ws + b0.invstd, params + b0.g_off,
                                                    nullptr, grads + b0.g_off, grads + b0.b_off, n->B, n->Hs, n->Ws, 64, ws + n->bn_ws,
                                                    n->bn_ws_bytes, st));
                }
                OSI_TRY(n->mark(OSI_PROF_BN_BWD, st));
                // the stem's weight gradient is the last kernel of the step; on the MAIN stream it runs beside the side stream's
                // backlog (layer1's weight gradients) instead of behind it
                const bool async = n->async_wgrad() && !n->stem_wgrad_main;
                hipStream_t gs = st;
                if (async) {
                    if (hipEventRecord(n->ev_fork, st) != hipSuccess) return OSI_ERR_LAUNCH;
                    if (hipStreamWaitEvent(n->side, n->ev_fork, 0) != hipSuccess) return OSI_ERR_LAUNCH;
                    gs = n->side;
                }
                OSI_TRY(osi_stem_wgrad_fused(&c0.d, S(go), ws + n->pool_idx, ws + c0.y, x4c, params + b0.g_off, ws + b0.mean, ws + b0.invstd,
                                             grads + b0.g_off, grads + b0.b_off, grads + c0.w_off, ws + n->stem_ws, n->stem_ws_bytes, gs));
                if (async) {
                    if (hipEventRecord(n->buf_ev[go], n->side) != hipSuccess) return OSI_ERR_LAUNCH;
                    n->buf_pending[go] = true;
                    n->side_dirty = true;
                }
                OSI_TRY(n->mark(OSI_PROF_CONV_WGRAD, gs));
                n->give(go);
            } else {
                // max-pool scatter + ReLU gate + bn1 backward gathered on the fly from the pooled gradient (no 112x112x64 gradient)
                OSI_TRY(osi_bn_relu_maxpool_bwd(S(go), ws + n->pool_idx, ws + c0.y, ws + b0.mean, ws + b0.invstd, params + b0.g_off, S(t),
                                                grads + b0.g_off, grads + b0.b_off, n->B, n->Hs, n->Ws, 64, ws + n->bn_ws, n->bn_ws_bytes, st));
                OSI_TRY(n->mark(OSI_PROF_BN_BWD, st));
                n->give(go);
                OSI_TRY(wgrad(n, 0, grads, ws, t, x4c, st));
                n->give(t);
            }
            n->cur_grad = -1;
            n->fwd_done = false;
        }
        n->next_stage = stage + 1;
    }
    // Join once per call: every gradient of the stages just run is final on `st` from here on. A data-parallel caller issues
    // one stage per call (and reduces that slice next); a single-GPU caller issues all stages in one call and pays one join.
    OSI_TRY(flush_wgrad(n, st));
    if (n->stage_join || stage_hi == n->n_stages) {
        n->w_inflight = false;         // the join below covers it
        OSI_TRY(n->join_side(st));
    }
    return OSI_OK;
}

// Hand the gradients of the stages enqueued so far to another stream WITHOUT stalling the compute stream: `waiter` waits for the
// work enqueued on `main` up to now and for the side stream's weight gradients; `main` waits for nothing.
int osi_resnet50_grads_ready(osi_resnet50_t n, osi_stream_t main_stream, osi_stream_t waiter_stream) {
    OSI_REQUIRE(n);
    hipStream_t mn = (hipStream_t)main_stream, wt = (hipStream_t)waiter_stream;
    if (!n->ev_rmain) {
        if (hipEventCreateWithFlags(&n->ev_rmain, EV_FLAGS_HANDOFF) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipEventCreateWithFlags(&n->ev_rside, EV_FLAGS_HANDOFF) != hipSuccess) return OSI_ERR_LAUNCH;
    }
    if (hipEventRecord(n->ev_rmain, mn) != hipSuccess) return OSI_ERR_LAUNCH;
    if (hipStreamWaitEvent(wt, n->ev_rmain, 0) != hipSuccess) return OSI_ERR_LAUNCH;
    if (n->side && n->side_dirty) {
        if (hipEventRecord(n->ev_rside, n->side) != hipSuccess) return OSI_ERR_LAUNCH;
        if (hipStreamWaitEvent(wt, n->ev_rside, 0) != hipSuccess) return OSI_ERR_LAUNCH;
    }
    return OSI_OK;
}

// ---- debug: the non-smooth decisions of the latest forward (tests only; the product path never calls these) -------------------
// The network's only non-differentiable points are its 49 ReLUs and the max-pool's arg-max. A whole-network gradient check
// against an fp64 oracle is dominated by the few elements where fp32 rounding flips such a decision (2e-2 relative); with the
// oracle taking THIS run's decisions the comparison is 400x tighter (tests/test_gate_pinned_gpu.py). Each gate is read from what
// the backward itself consumes: the stored bitmask for block outputs, fma(y0, scale0, shift0) > 0 for the in-block activations
// that were never materialised (the expression the forward loader and the dgrad epilogue evaluate), bit 7 / the low bits of the
// arg-max byte for the fused stem tail. Output order is the oracle's NCHW.
namespace {
__global__ __launch_bounds__(256) void k_dbg_gate_bits(const unsigned long long* __restrict__ bits, unsigned char* __restrict__ out,
                                                       size_t n, int HW, int C) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;   // NHWC element index
    if (e >= n) return;
    const size_t i4 = e >> 2;
    const unsigned long long w = bits[(i4 >> 6) * 4 + (e & 3)];
    const size_t pix = e / C;
    const int c = (int)(e - pix * C);
    const size_t b = pix / HW, hw = pix - b * HW;
    out[(b * C + c) * HW + hw] = (unsigned char)((w >> (i4 & 63)) & 1);
}
__global__ __launch_bounds__(256) void k_dbg_gate_fma(const float* __restrict__ y, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, unsigned char* __restrict__ out, size_t n, int HW,
                                                      int C) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const size_t pix = e / C;
    const int c = (int)(e - pix * C);
    const size_t b = pix / HW, hw = pix - b * HW;
    out[(b * C + c) * HW + hw] = __builtin_fmaf(y[e], scale[c], shift[c]) > 0.f ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_dbg_gate_pool(const uint32_t* __restrict__ idx, unsigned char* __restrict__ gate,
                                                       int* __restrict__ argmax, size_t n, int Ws, int Hp, int Wp, int C) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;   // NHWC element of the pooled tensor
    if (e >= n) return;
    const uint32_t byte = (idx[e >> 2] >> (8 * (e & 3))) & 0xffu;
    const size_t pix = e / C;
    const int c = (int)(e - pix * C);
    const size_t b = pix / ((size_t)Hp * Wp), hw = pix - b * Hp * Wp;
    const int ho = (int)(hw / Wp), wo = (int)(hw - (size_t)ho * Wp);
    const int tap = byte & 0x7f, r = tap / 3, s = tap - 3 * r;
    const size_t o = (b * C + c) * ((size_t)Hp * Wp) + hw;
    gate[o] = (unsigned char)(byte >> 7);
    if (argmax) argmax[o] = (ho * 2 - 1 + r) * Ws + (wo * 2 - 1 + s);
}
}  // namespace

int osi_resnet50_debug_num_gates(osi_resnet50_t n) { return n ? 1 + 3 * (int)n->blocks.size() : 0; }

int osi_resnet50_debug_gate_shape(osi_resnet50_t n, int i, int* C, int* H, int* W) {
    OSI_REQUIRE(n && C && H && W && i >= 0 && i < 1 + 3 * (int)n->blocks.size());
    if (i == 0) { *C = 64; *H = n->Hp; *W = n->Wp; return OSI_OK; }
    const Block& k = n->blocks[(i - 1) / 3];
    const int which = (i - 1) % 3;
    const osi_conv_desc& d = n->convs[which == 0 ? k.c1 : which == 1 ? k.c2 : k.c3].d;
    *C = d.Cout; *H = d.Ho; *W = d.Wo;
    return OSI_OK;
}

int osi_resnet50_debug_gate(osi_resnet50_t n, void* workspace, int i, unsigned char* gate_nchw, int* pool_argmax_nchw,
                            osi_stream_t stream) {
    OSI_REQUIRE(n && workspace && gate_nchw && i >= 0 && i < 1 + 3 * (int)n->blocks.size());
    OSI_REQUIRE(i == 0 || !pool_argmax_nchw);
    if (!n->any_fwd) return OSI_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (i == 0) {
        const size_t e = (size_t)n->B * n->Hp * n->Wp * 64;
        hipLaunchKernelGGL(k_dbg_gate_pool, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, st, (const uint32_t*)(ws + n->pool_idx),
                           gate_nchw, pool_argmax_nchw, e, n->Ws, n->Hp, n->Wp, 64);
        OSI_LAUNCH_CHECK();
        return OSI_OK;
    }
    const Block& k = n->blocks[(i - 1) / 3];
    const int which = (i - 1) % 3;
    const Conv& c = n->convs[which == 0 ? k.c1 : which == 1 ? k.c2 : k.c3];
    const BN& b = n->bns[c.bn];
    const size_t e = (size_t)b.M * b.C;
    const int HW = c.d.Ho * c.d.Wo;
    if (which == 2)
        hipLaunchKernelGGL(k_dbg_gate_bits, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, st,
                           (const unsigned long long*)(ws + c.mask), gate_nchw, e, HW, b.C);
    else
        hipLaunchKernelGGL(k_dbg_gate_fma, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, st, ws + c.y, ws + b.scale, ws + b.shift,
                           gate_nchw, e, HW, b.C);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_resnet50_set_overlap(osi_resnet50_t n, int enable) {
    OSI_REQUIRE(n);
    n->overlap = enable != 0;
    return OSI_OK;
}

int osi_resnet50_set_option(osi_resnet50_t n, const char* name, int value) {
    OSI_REQUIRE(n && name);
    if (!strcmp(name, "overlap")) n->overlap = value != 0;
    else if (!strcmp(name, "fwd_fork")) n->fwd_fork = value != 0;
    else if (!strcmp(name, "stagger")) n->stagger = value != 0;
    else if (!strcmp(name, "fwd_recompute")) n->fwd_recompute = value != 0;
    else if (!strcmp(name, "stem_fused")) n->stem_fused = value != 0;
    else if (!strcmp(name, "stem_pool_stats")) n->stem_pool_stats = value != 0;
    else if (!strcmp(name, "ds_sparse")) n->ds_sparse = value != 0;
    else if (!strcmp(name, "stem_wgrad_main")) n->stem_wgrad_main = value != 0;
    else if (!strcmp(name, "stage_join")) n->stage_join = value != 0;
    else if (!strcmp(name, "eval_fused")) n->eval_fused = value != 0;
    else if (!strcmp(name, "wino_weights_aside")) n->wt_aside = value != 0;
#ifdef OSI_DIAG
    else if (!strcmp(name, "dbg_skip")) n->dbg_skip = value;
#endif
    else if (!strcmp(name, "side_priority_normal")) {
        if (n->side) return OSI_ERR_STATE;   // the side stream already exists with the other priority
        n->side_prio_normal = value != 0;
    } else return OSI_ERR_ARG;
    return OSI_OK;
}

}  // extern "C"
