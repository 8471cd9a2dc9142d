// Shared by the translation units of libeasyhybrid_hip.so (eh_api.hip: entry points and their kernels; eh_comm.hip: the library's
// own collectives -- RCCL binding, local groups, the peer-to-peer exchange): the handle behind the opaque eh_handle pointer, the
// error convention and the few helpers both sides call.  Not installed; the public ABI is include/easyhybrid_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is bound at run time, by the first eh_comm_* call (see EhRccl in eh_comm.hip)

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "eh_arch.hpp"
#include "eh_jit.hpp"

// Where the optimiser mirrors theta into the padded parameter image the step kernel stages.
struct EhImg {
    float* image;
    const int* imap;     // canonical index -> image offset (-1 for the raw globals)
    int g_off, phi_off;
    int glob_par[EH_MAX_PARAMS];   // global g -> canonical mech parameter j
    float glo[EH_MAX_PARAMS], ghi[EH_MAX_PARAMS];
    // extra loss lambda * weight_l2(ps; normalize) (src/utils/extract_weights.jl:69-91): l2c = lambda or lambda / #weights;
    // the Dense weight matrices are the canonical entries whose image offset lies below the bias block
    float l2c;
    int b_off;
    const unsigned char* wflag;   // layer-wise form (no image, imap == nullptr): 1 at the canonical positions of Dense weights
    // several weight_l2 terms (one per network of a MultiNNHybridModel, each with its own lambda / normalisation, or the biases:
    // extract_weights.jl:64 `l2_Rb = lambda * weight_l2(ps.Rb; normalize = true)`): one coefficient per canonical entry,
    // extra loss = sum_i l2w[i] theta_i^2 (eh_set_weight_l2_coef); nullptr: the one-lambda form above
    const float* l2w;
    int n_theta;
    // agg = mean (src/config/TrainingConfig.jl:76-77; compute_loss.jl:31-34,50-53): the training loss is
    // mean([mean_t(L_t), extra terms...]) = agg_a * sum_t L_t + l2s * sum of the extra terms, agg_a = 1 / (T (1 + E)), l2s = 1 / (1 + E);
    // agg = sum: both 1
    float agg_a, l2s;
};

// --------------------------------------------------------------------------------------------
// handle
// --------------------------------------------------------------------------------------------
constexpr int EH_MULTI_MAX = 256;         // steps per launch of the multi-step kernel (EH_MODE_TRAIN_MULTI): bounds a launch to a millisecond or two
constexpr int EH_EVAL_BLOCKS = 1024;      // evaluation passes of the per-wave kernels: up to four workgroups per CU (eval_grid_for, eh_api.hip)

struct EhSplit {
    float* recs = nullptr;
    long long n = 0;
    float shift[EH_MAX_TARG] = {0, 0, 0, 0};
};

// kernel selector handed to EhVariant::launch: the fast-path bits, or 4 = the EH_MECH_PROGRAM kernels
#define KFAST(h) ((h)->net.mech == EH_MECH_PROGRAM ? 4 : (h)->fast)
#define TH(h) ((h)->thb[(h)->cur])
#define MM(h) ((h)->mb[(h)->cur])
#define VV(h) ((h)->vb[(h)->cur])

struct eh_handle_s {
    eh_model_desc desc;
    EhNet net;
    const EhArchInfo* arch = nullptr;
    const EhArchInfo* arch_alt = nullptr;   // the other kernel family built for this shape ("row_split" option), if any
    int variant = 0, act = 0, fast = 0;
    float* image = nullptr;
    int* imap = nullptr;
    int* rmap = nullptr;            // reduction map for the current kernel family / variant / fast-path flags (v3: the inverse map)
    size_t rmap_cap = 0;
    // block placement of every net inside the padded (block-diagonal) MLP
    int n_nets = 1;                                     // 1 for SingleNN
    int net_P[EH_MAX_NETS] = {0}, net_K[EH_MAX_NETS] = {0};
    int net_w[EH_MAX_NETS][EH_MAX_HIDDEN] = {{0}};      // hidden widths of net k
    int net_d[EH_MAX_NETS] = {0};                       // hidden layers of net k (layers past them: identity blocks, not in theta)
    char* mech_ws = nullptr;                            // eh_mech_loss_vjp: [counts | out | partial rows]
    size_t mech_ws_bytes = 0;
    int net_c0[EH_MAX_NETS] = {0};                      // first predictor row of net k
    int net_r0[EH_MAX_NETS][EH_MAX_HIDDEN + 1] = {{0}}; // first row of net k in layer l (l == n_hidden: output row)
    int tot_w[EH_MAX_HIDDEN] = {0};                     // total (summed) hidden widths
    EhImg img{};
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    int C = 0, n_acc = 0, n_par = 0;
    float *thb[2] = {nullptr, nullptr}, *mb[2] = {nullptr, nullptr}, *vb[2] = {nullptr, nullptr};   // parameter sets (fused mode ping-pongs them)
    float* pset = nullptr;          // backing allocation of thb/mb/vb/sc
    float* sc = nullptr;            // [2][2] running beta products, ping-pong
    int cur = 0, sc_sel = 0;
    // fused-update mode
    bool fused = false, pending = false, fused_det = false;   // fused_det ("fused_update" 2): one kernel per step only where one workgroup covers the minibatch
    float* gacc = nullptr;          // [3][EH_GSHARDS][n_acc] rotating gradient accumulators
    // cross-GPU exchange (EhP2P): an uncached, IPC-exported receive buffer of {value, sequence} words next to gacc
    bool p2p_on = false, p2p_alloc = false;
    unsigned long long* p2p_recv = nullptr;
    int p2p_world = 0, p2p_rank = 0;
    unsigned p2p_seq = 0;
    float* p2p_stage = nullptr;
    unsigned* p2p_ctr = nullptr;    // [0] top-level ticket, [1] error flag, [2] self-test mismatches, [32 (1 + g)] group tickets (eh_p2p_publish)
    EhP2P* p2p_dev = nullptr;
    EhP2P p2p_host{};               // the same descriptor, handed to the step kernels by value
    void* p2p_peer[EH_GSHARDS] = {nullptr};
    eh_handle_s* p2p_group[EH_GSHARDS] = {nullptr};   // eh_p2p_init_local: the members, by rank
    bool p2p_local = false;         // the peers are handles of this process (eh_p2p_init_local): plain pointers, nothing to unmap
    long long gstep = 0;
    float* pending_loss = nullptr;
    // input BatchNorm
    bool bn_on = false;
    float* bn_part = nullptr;       // [32][64] partials + c[32]
    float* bn_run = nullptr;        // [2][32] running mean / var
    float* bn_shift = nullptr;      // [32] common shift of the cross-GPU statistics (eh_set_bn_shift)
    float* bn_stat = nullptr;       // [65] sum d | sum d^2 | n of the current step, all-reduced by the host (EH_BUF_BNSTAT)
    float* tcount = nullptr;        // [EH_MAX_TARG][3] n_t | sum (y - c) | sum (y - c)^2 of the current step's shard, all-reduced by the caller (EH_BUF_TCOUNT)
    float* mombuf = nullptr;        // [EH_MAX_TARG][EH_EVAL_STATS] this shard's moments of (yhat, y) of the current step (eh_dp_moments), all-reduced by the caller (EH_BUF_MOMENT)
    int mom_stage = 0;              // eh_dp_moments: 0 = none, 1 = the sums about the shift are out, 2 = the sums about the global centre are out
    bool dp_moments = false;        // the step being launched takes its two-pass coefficients from the all-reduced moments (no local statistics passes)
    bool dp_weights = false;        // the step being launched takes its per-target weights from the all-reduced sums (no local counting pass)
    bool tcount_ready = false;      // eh_dp_counts ran for the step eh_dp_grad is about to take
    bool bn_ext = false;            // bn_stat holds the statistics of the step about to run
    bool bn_dp_update = false;
    bool opt_ready = false;
    // layer-wise execution form (eh_lform.hpp): networks no fused kernel holds
    bool lform = false;
    // one entry per network (SingleNN: one; MultiNN: one single-output network per neural parameter, each on its own predictor rows)
    struct LNet { int nl = 0, c0 = 0, orow = 0, act = 0; int lact[EH_MAX_HIDDEN + 1] = {0}; int in[EH_MAX_HIDDEN + 1] = {0}, out[EH_MAX_HIDDEN + 1] = {0}, woff[EH_MAX_HIDDEN + 1] = {0}, boff[EH_MAX_HIDDEN + 1] = {0}; };
    int l_nnets = 0;                                     // 0: no network at all (no neural parameter)
    LNet l_net[EH_MAX_NETS];
    float* l_split = nullptr; size_t l_split_cap = 0;    // split-K partial products of the small-batch GEMMs
    unsigned* l_lprog = nullptr; int l_lprog_gen = -1;  // the recorded loss programs as the layer-wise form interprets them (device copy, generation it was made from)
    float* l_dk = nullptr; size_t l_dk_cap = 0;          // every layer's delta of a small-batch step (the weight gradients then run as one grouped launch)
    bool l_job_done = false;                              // lform_gemm -> lform_train: a few-rows product took the side job of summing the chain's partial rows
    struct EhLApply* l_apply = nullptr; bool l_applied = false;   // do_step -> lform_train: the optimiser may run in the epilogue of the grouped weight gradients / it did
    bool l_tail_fn[12] = {false};
    float* l_ws = nullptr;                               // [Xb | H_0 .. H_{NL-1} | D0 | D1 | O | mech partial rows]
    long long l_cap = 0;                                 // samples the workspace holds
    unsigned char* wflag = nullptr;
    int slab_rows = 256;
    ncclComm_t comm = nullptr;      // eh_comm_init: the library's own RCCL communicator (data parallelism without a host-side collective library)
    int comm_world = 0, comm_rank = 0;
    struct EhLocalGroup* lgroup = nullptr;   // eh_comm_init_local: handles of ONE process exchange through peer-mapped device memory, no RCCL
    unsigned roles = 0;             // eh_set_target_roles: 2 bits per target -- 0 a data target, 1 / 2 an entry of the extra loss (mean / sum over all samples of a recorded function of the prediction)
    int agg = 0, n_extra = 0;       // eh_set_option "agg" (0 = sum, 1 = mean) / "extra_terms" (entries the extra loss returns); img.agg_a / img.l2s follow
    EhOpt opt{};
    EhSplit split[2];
    float *slab = nullptr, *gradbuf = nullptr, *inv_n = nullptr, *loss_hist = nullptr;
    long long loss_cap = 0;
    int* perm = nullptr;
    long long perm_cap = 0;
    bool perm_valid = false;
    int fast_user = 3;              // what the fast_paths option allows (default: all)
    unsigned* prog = nullptr;       // EH_MECH_PROGRAM: device copy of the program (EhStepArgs::prog layout)
    // EH_MECH_PROGRAM: kernels compiled at run time around the program (eh_jit.hpp), one entry per (kernel family, variant) used
    // state: 0 = being compiled by `worker` ("specialize" = 2: the steps run the kernels built ahead of time meanwhile), 1 = ready, -1 = failed
    // verified: the kernel has been run next to the one built ahead of time on one window of the user's data and agreed (jit_verify, eh_api.hip)
    struct JitEntry { const EhArchInfo* arch; int variant, fast; bool spec, p2p; EhNet net; int loss_gen; std::atomic<int> state{0}; EhJitKernel k; std::thread worker; std::string log; bool verified = false; };
    std::vector<std::unique_ptr<JitEntry>> jit;
    bool check_idx = false;         // "check_idx" option: range-check device-side minibatch indices before the step (debug)
    bool empty_nan = false;         // "empty_target_nan" option: eh_loss_and_grad reports NaN when a target of a non-empty batch has no valid sample (the reference's value; gradient unchanged)
    bool aot_spec = true;           // "aot_spec" option / EH_NO_AOT_SPEC: run the kernel specialised ahead of time when the descriptor is a canonical one (eh_spec.hip)
    const struct EhSpecKernel* spec_used = nullptr;      // ... the one the last launch ran (eh_jit_status reports it)
    bool specialize_async = false;  // "specialize" = 2
    bool jit_on = true;             // "jit" option / EH_JIT=0: 0 = the interpreting kernels built ahead of time
    bool jit_failed = false;
    bool specialize = false;        // "specialize" option: every model gets kernels compiled around its descriptor
    EhLossProg loss_prog;           // eh_set_loss_program (EH_LOSS_PROGRAM)
    std::string jit_log;
    float* l2val = nullptr;         // lambda * weight_l2 of the current parameters (device scalar)
    float* l2w = nullptr;           // eh_set_weight_l2_coef: one coefficient per canonical entry (device)
    int n_weights = 0;
    struct GraphRec { hipGraphExec_t exec; bool fused; int gslot, cur, sc_sel; };
    std::vector<GraphRec> graphs;         // eh_graph_*: captured step sequences + the rotation state they start (and must end) in
    bool capturing = false;
    GraphRec cap{};
    int max_blocks = 256;
    bool multi_step = true;         // "multi_step" option (eh_train_epoch: several one-workgroup steps per launch)
    bool bn_no_self = false;        // "bn_in_kernel" 0
    int eval_blocks = 0;            // "eval_blocks" option: workgroups of eh_eval / eh_forward (0 = per kernel family, eval_grid_for)
    int mech_blocks = 0;            // "mech_blocks" option: cap on the streaming kernel's workgroups (0: none -- one workgroup per `mech_tiles` tiles)
    int mech_tiles = 0;             // "mech_tiles" option: consecutive 256 V-sample tiles per workgroup (0: by model -- 2, multi-output models 8)
    // scratch for forward / eval outputs
    float* out_buf = nullptr;
    float* eval_host = nullptr;          // pinned, device-visible: the evaluation kernels store their per-workgroup sums straight into it
    float* eval_host_dev = nullptr;
    size_t eval_host_cap = 0;            // floats
    long long out_cap = 0;
    int* idx_buf = nullptr;
    long long idx_cap = 0;
    // profiling
    bool prof = false;
    int prof_stride = 1;            // events bracket bursts of this many steps (1 = every kernel of every step)
    long long prof_k = 0;
    std::vector<hipEvent_t> ev;   // 3 per step: before step kernel, between, after reduce
    size_t ev_used = 0;
    unsigned long long* stamps = nullptr;   // diagnostic builds only
    std::string err;
};


// error convention: the message goes to the handle (or, without one, to the slot eh_last_error(NULL) reads); the code comes back
int fail(eh_handle* h, int code, const char* fmt, ...);
extern std::string g_create_err;
#define HIPCHK(h, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(h, e_ == hipErrorOutOfMemory ? EH_ENOMEM : EH_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// fused-update mode: apply the pending gradient so theta / m / v / image are current (eh_api.hip)
int flush_pending(eh_handle* h);
#define FLUSH(h)                          \
    do {                                  \
        int rc_ = flush_pending(h);       \
        if (rc_) return rc_;              \
    } while (0)

// bit t set: target t's training loss needs batch moments of the predictions ahead of the pass (two forward-only passes; eh_api.hip)
unsigned eh_two_pass_mask(const eh_handle* h);

// eh_destroy's share of eh_comm.hip: leave the communicator / local group, unmap the peers, free the exchange buffers
void eh_comm_release(eh_handle* h);
