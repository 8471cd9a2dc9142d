// C ABI of libeasyhybrid_hip.so (declared in include/easyhybrid_hip.h): model, data, parameters, forward / eval, the training
// step and epoch, the optimiser, graphs, the data-parallel seam eh_dp_*.  The small kernels live in eh_kernels.hpp, the handle in
// eh_internal.hpp, the library's own collectives (eh_comm_*, eh_p2p_*) in eh_comm.hip.  Everything here runs on one HIP stream
// per handle; there is no CPU compute path.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "eh_internal.hpp"
#include "eh_kernels.hpp"
#include "eh_wide.hpp"
#include "eh_lform.hpp"

std::string g_create_err;

int fail(eh_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

static unsigned two_pass_mask(const EhNet& net);

// ---- fused-update mode: apply the pending gradient so theta / m / v / image are current -----------
static int flush_one(eh_handle* h);
int flush_pending(eh_handle* h) {
    if (!h->pending) return EH_OK;
    // Publishing mode 1: a step's sums reach the peers from the NEXT kernel on each rank's stream -- the next step or this flush -- so a
    // drain is collective: the flush kernel waits for words that only the peers' own next kernels store.  Members of a one-process
    // local group are drained together here, all launches before anybody synchronises (advisor r05: eh_get_params / eh_eval / a loss
    // read / eh_set_option on ONE member used to run into the 2 s deadline and apply the step with the peers' sums read as 0).  Rank
    // processes cannot be reached from here: there every draining call has to be made on all ranks (dp.py).
    if (h->p2p_on && h->p2p_local && h->p2p_host.mode == 1) {
        for (int r = 0; r < h->p2p_world; ++r) {
            eh_handle* g = h->p2p_group[r];
            if (!g || g == h || !g->pending) continue;
            HIPCHK(g, hipSetDevice(g->device));
            if (int rc = flush_one(g)) return rc;
        }
        HIPCHK(h, hipSetDevice(h->device));
    }
    return flush_one(h);
}
static int flush_one(eh_handle* h) {
    const int nt = h->net.n_theta;
    const float* g_prev = h->gacc + (size_t)((h->gstep + 2) % 3) * EH_GSHARDS * h->n_acc;
    float* sc_in = h->sc + 2 * h->sc_sel;
    float* sc_out = h->sc + 2 * (h->sc_sel ^ 1);
    hipLaunchKernelGGL(eh_fused_flush_kernel, dim3((nt + 255) / 256), dim3(256), 0, h->stream, g_prev, h->n_acc, nt, TH(h), MM(h), VV(h), sc_in, sc_out,
                       h->opt, h->pending_loss, h->img, h->net.loss, h->p2p_on ? h->p2p_dev : nullptr, (int)((h->gstep + 2) % 3), h->p2p_seq, h->net.T);
    HIPCHK(h, hipGetLastError());
    // Single GPU: nothing to clear -- the rotation keeps itself clean (the step that accumulates into slot g clears slot g + 1 in
    // its prologue and nobody reads a slot before the step after its clearing has filled it).  Under EhP2P the workgroups add
    // into a staging copy that the publishing workgroup reads whole: cleared here.
    if (h->p2p_on) HIPCHK(h, hipMemsetAsync(h->p2p_stage, 0, (size_t)3 * EH_GSHARDS * h->n_acc * sizeof(float), h->stream));
    h->sc_sel ^= 1;
    h->pending = false;
    h->pending_loss = nullptr;
    return EH_OK;
}
// Every trainable scalar of the model as (canonical flat index, layer, row, col) in the padded
// block-diagonal MLP.  col < 0 marks a bias.  SingleNN = one net covering everything.
struct EhEntry { int canon, l, row, col; };
static std::vector<EhEntry> enumerate_entries(const eh_handle* h) {
    std::vector<EhEntry> v;
    const int nl = h->desc.n_hidden;
    int off = 0;
    for (int k = 0; k < h->n_nets; ++k) {
        int in = h->net_P[k];
        for (int l = 0; l <= nl; ++l) {
            const int o = l < nl ? h->net_w[k][l] : h->net_K[k];
            const int r0 = h->net_r0[k][l], c0 = l == 0 ? h->net_c0[k] : h->net_r0[k][l - 1];
            if (l < nl && l >= h->net_d[k]) continue;            // identity block of a shallower net
            for (int col = 0; col < in; ++col)
                for (int row = 0; row < o; ++row) v.push_back({off + row + o * col, l, r0 + row, c0 + col});
            off += o * in;
            for (int row = 0; row < o; ++row) v.push_back({off + row, l, r0 + row, -1});
            off += o;
            in = o;
        }
    }
    return v;
}

// canonical index -> parameter-image offset; canonical index <-> accumulator element of the step
// kernel (rmap: where the element sits among the parked accumulators; see eh_acc_layout)
static int build_maps(eh_handle* h, bool with_imap) {
    const eh_model_desc& d = h->desc;
    const EhNet& n = h->net;
    const EhArchInfo* A = h->arch;
    const int nbi = A->nbi, nbh = A->nbh, nl = A->nl, fast = h->fast;
    const EhAccLayout L = eh_acc_layout(nbi, nbh, nl, fast);
    const std::vector<EhEntry> ent = enumerate_entries(h);
    const int nwv = A->wide ? A->var[h->variant].nw : 4;
    const EhWideLayout WL = eh_wide_layout(nbi, nbh, nl, nwv);
    std::vector<int> imap((size_t)n.n_theta, -1), rmap((size_t)h->n_acc, 0);
    // v2 region[k][g][r][c] where one wave workspace holds a wave's raw accumulators, else v3 region[k][c][g][r] (eh_step_body, workgroup reduction)
    const bool v2 = A->wide || L.rw <= A->var[h->variant].red_floats / A->var[h->variant].nw;
    auto at = [v2](int k, int lane, int r) { return v2 ? k * 256 + (lane >> 4) * 64 + r * 16 + (lane & 15) : k * 256 + (lane & 15) * 16 + (lane >> 4) * 4 + r; };
    for (const EhEntry& e : ent) {
        const int m = e.row / 16, g = (e.row % 16) / 4, r = e.row % 4;
        int img, k, lane, rr, nlan;
        if (e.col < 0) {                                   // bias of layer l
            img = A->b_off + e.l * A->hp + e.row;
            if (e.l < nl) { k = L.kb + e.l * nbh + m; lane = 16 * g; rr = r; nlan = 16; }
            else if (fast & 1) { k = -1; lane = 0; rr = 0; nlan = 1; }          // K1: scalar in the tail
            else { k = L.kbo; lane = 16 * (e.row / 4); rr = e.row % 4; nlan = 16; }
        } else if (e.l == 0) {
            img = A->w0_off + e.row * A->s0 + e.col;
            if (fast & 2) { k = L.kw0 + m * 4 + e.col; lane = 16 * g; rr = r; nlan = 16; }
            else { k = L.kw0 + m * nbi + e.col / 16; lane = 16 * g + e.col % 16; rr = r; nlan = 1; }
        } else if (e.l < nl) {
            img = A->wh_off + (e.l - 1) * A->hp * A->sh + e.row * A->sh + e.col;
            k = L.kwh + ((e.l - 1) * nbh + m) * nbh + e.col / 16; lane = 16 * g + e.col % 16; rr = r; nlan = 1;
        } else {
            img = A->wo_off + e.row * A->sh + e.col;
            if (fast & 1) { k = L.kwo + e.col / 16; lane = 16 * ((e.col % 16) / 4); rr = e.col % 4; nlan = 16; }
            else { k = L.kwo + e.col / 16; lane = 16 * (e.row / 4) + e.col % 16; rr = e.row % 4; nlan = 1; }
        }
        imap[e.canon] = img;
        if (A->wide) {
            // row-split kernel: wave w owns feature blocks [w*mb, (w+1)*mb) of every layer
            int w = 0, kk;
            const int mb = WL.mb;
            if (e.l < nl) {
                w = m / mb;
                const int mm = m % mb;
                if (e.col < 0) { kk = WL.kb + e.l * mb + mm; lane = 16 * g; rr = r; }
                else if (e.l == 0) { kk = WL.kw0 + mm * nbi + e.col / 16; lane = 16 * g + e.col % 16; rr = r; }
                else { kk = WL.kwh + ((e.l - 1) * mb + mm) * nbh + e.col / 16; lane = 16 * g + e.col % 16; rr = r; }
            } else if (e.col < 0) {
                kk = WL.kbo; lane = 16 * (e.row / 4); rr = e.row % 4;
            } else {
                const int q = e.col / 16;
                w = q / mb;
                kk = WL.kwo + q % mb; lane = 16 * (e.row / 4) + e.col % 16; rr = e.row % 4;
            }
            rmap[e.canon] = (w * WL.na + kk) * 256 + lane * 4 + rr;      // position in the kernel's LDS staging
            continue;
        }
        if (k < 0) { rmap[e.canon] = (L.na * 256 + 13) | (1 << 24); }
        else rmap[e.canon] = at(k, lane, rr) | (nlan << 24);
    }
    for (int j = 0; j < d.n_params; ++j)
        if (d.param_kind[j] == EH_PAR_GLOBAL) rmap[n.g_off + d.param_index[j]] = (L.na * 256 + j) | (1 << 24);
    rmap[n.n_theta] = (L.na * 256 + 8) | (1 << 24);
    for (int t = 0; t < n.T; ++t) rmap[n.n_theta + 1 + t] = (L.na * 256 + 9 + t) | (1 << 24);
    rmap[n.n_theta + 1 + n.T] = (L.na * 256 + 14) | (1 << 24);
    rmap[n.n_theta + 2 + n.T] = (L.na * 256 + 15) | (1 << 24);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (with_imap) {
        if (!h->imap) HIPCHK(h, hipMalloc(&h->imap, imap.size() * sizeof(int)));
        HIPCHK(h, hipMemcpy(h->imap, imap.data(), imap.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // the per-wave kernel's v3 reduction gathers in accumulator order: it wants the INVERSE map, word of a parked accumulator -> canonical
    // index (-1: padding, structural zeros, the columns of a row sum but the first), then the 16 tail words
    const size_t emap_n = (size_t)L.na * 256 + 16;
    if (!A->wide && !v2) {
        std::vector<int> emap(emap_n, -1);
        for (int e = 0; e < h->n_acc; ++e) {
            const int pos = rmap[(size_t)e] & 0xFFFFFF;
            if (pos >= 0 && (size_t)pos < emap_n) emap[(size_t)pos] = e;
        }
        rmap.swap(emap);
    }
    if (rmap.size() > h->rmap_cap) {           // (the kernel family / variant / fast-path options change the layout, and with it the size)
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->rmap); h->rmap = nullptr; h->rmap_cap = 0;
        const size_t want = std::max(rmap.size(), std::max((size_t)h->n_acc, emap_n));
        HIPCHK(h, hipMalloc(&h->rmap, want * sizeof(int)));
        h->rmap_cap = want;
    }
    HIPCHK(h, hipMemcpy(h->rmap, rmap.data(), rmap.size() * sizeof(int), hipMemcpyHostToDevice));
    return EH_OK;
}

// Which vector-ALU fast paths a model may use: bit 0 = single NN output (K == 1), bit 1 = P <= 4 predictors.  Single-target
// models only (the K == 1 kernels keep one residual per sample).  (Round 1 confined the P <= 4 path to the one-block shapes
// because the wider ReLU kernels lost the loss sum; that was the SLP vectoriser -- see the Makefile -- and is gone with it.)
static int fast_wanted(const EhArchInfo* A, int K, int P, int T, int mech) {
    if (!A->has_fast || T != 1 || K != 1 || mech == EH_MECH_PROGRAM) return 0;
    return 1 | (P <= 4 ? 2 : 0);
}

struct MechInfo { int n_par, n_forc, n_out; };
static bool mech_info(int mech, MechInfo* mi, const eh_model_desc* d = nullptr) {
    switch (mech) {
        case EH_MECH_PROGRAM: if (!d) return false; *mi = {d->n_params, d->prog_n_forc, d->prog_n_out}; return true;
        case EH_MECH_RBQ10: *mi = {2, 1, 1}; return true;
        case EH_MECH_EXPO: *mi = {2, 1, 1}; return true;
        case EH_MECH_LINEAR: *mi = {2, 1, 1}; return true;
        case EH_MECH_EXPO2POOL: *mi = {4, 1, 1}; return true;
        case EH_MECH_RS_COMPONENTS: *mi = {6, 1, 1}; return true;
        case EH_MECH_RS_COMPONENTS3F: *mi = {6, 3, 1}; return true;
        case EH_MECH_FLUXPART: *mi = {3, 2, 3}; return true;
        default: return false;
    }
}

static const EhArchInfo* find_arch(int nbi, int nbh, int nl) {
#define EH_ARCH_TRY(a, b, c) if (nbi == a && nbh == b && nl == c) return eh_arch_##a##_##b##_##c();
    EH_ARCH_LIST(EH_ARCH_TRY)
#undef EH_ARCH_TRY
    return nullptr;
}
static const EhArchInfo* find_wide(int nbi, int nbh, int nl) {
#define EH_ARCH_TRY(a, b, c) if (nbi == a && nbh == b && nl == c) return eh_wide_##a##_##b##_##c();
    EH_WIDE_LIST(EH_ARCH_TRY)
#undef EH_ARCH_TRY
    return nullptr;
}
// The "shape" of a model that runs layer by layer (eh_lform.hpp): no fused kernel, no parameter image beyond the EH_IMG_* block
static hipError_t lform_prepare(void) { return hipSuccess; }
static hipError_t lform_launch(int, int, int, int, hipStream_t, const EhNet*, const EhStepArgs*) { return hipErrorNotSupported; }
static const EhArchInfo g_lform_arch = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, /*phi_off*/ 0, /*img_floats*/ EH_IMG_META, /*has_fast*/ 0, /*nvar*/ 1,
                                        {{4, 4, 0, 1 << 30, &lform_prepare, &lform_launch, 1, 0}, {}, {}, {}}, /*wide*/ 1};
enum { EH_LFORM_ROWS = 64 };      // partial slabs of the weight gradients (split over the samples of a minibatch)

static bool arch_fits(const EhArchInfo* A, int need) {
    for (int vi = 0; vi < A->nvar; ++vi)
        if ((long long)(A->var[vi].nw / 2) * need > A->var[vi].red_floats) return false;      // (family choice, not a hard limit: a gradient this large relative to the workspaces is summed across the waves in several rounds, the row-split kernel needs no such sum)
    return true;
}

namespace {
struct EvalHostBuf { float* host; float* dev; size_t cap; int device; };
std::mutex g_eval_pool_mu;
std::vector<EvalHostBuf> g_eval_pool;
}
// (see eval_host_acquire)
static void eval_host_release(eh_handle* h) {
    if (!h->eval_host) return;
    {
        std::lock_guard<std::mutex> lk(g_eval_pool_mu);
        if (g_eval_pool.size() < 8) { g_eval_pool.push_back({h->eval_host, h->eval_host_dev, h->eval_host_cap, h->device}); h->eval_host = nullptr; }
    }
    if (h->eval_host) (void)hipHostFree(h->eval_host);
    h->eval_host = nullptr; h->eval_host_dev = nullptr; h->eval_host_cap = 0;
}
// Launches the step kernel of the handle's (family, variant).  A recorded closure, or any model with the "specialize" option,
// runs a kernel compiled at run time (eh_jit.hpp; built on first use, one per descriptor state; a failed build or launch
// switches the handle to the kernels built ahead of time for good); everything else runs the table entry.
static bool jit_wanted(const eh_handle* h, int mode) {
    const bool prog = h->net.mech == EH_MECH_PROGRAM, closs = h->net.loss == EH_LOSS_PROGRAM;
    if (mode == EH_MODE_TRAIN_P2P && h->act == EH_ACT_PER_NET) return false;
    if (mode == EH_MODE_TRAIN_P2P) return h->jit_on && !h->jit_failed && h->specialize && !prog && !closs && !h->arch->wide;
    if (closs || h->act == EH_ACT_PER_NET) return mode != EH_MODE_TRAIN_P2P;          // a recorded loss / per-net activations exist in run-time compiled kernels only
    return h->jit_on && !h->jit_failed && (h->specialize || prog);
}
// the compiled kernels for the handle's current (family, variant, descriptor); builds them on first use; nullptr = not available
static eh_handle_s::JitEntry* jit_entry(eh_handle* h) {
    const int kf = KFAST(h);
    const bool closs = h->net.loss == EH_LOSS_PROGRAM, prog = h->net.mech == EH_MECH_PROGRAM;
    const bool spec = h->specialize || prog || closs || h->act == EH_ACT_PER_NET;        // a model that is compiled anyway gets its descriptor baked in as well
    const bool want_p2p = h->specialize && h->p2p_on && !closs && !prog;
    const int lgen = closs ? h->loss_prog.gen : 0;
    for (auto& up : h->jit) {
        eh_handle_s::JitEntry& e = *up;
        if (e.arch == h->arch && e.variant == h->variant && e.fast == kf && e.spec == spec && (e.p2p || !want_p2p) && e.loss_gen == lgen &&
            (!e.spec || !memcmp(&e.net, &h->net, sizeof(EhNet)))) {
            const int st = e.state.load(std::memory_order_acquire);
            if (st < 0 && !h->jit_failed) { h->jit_log = e.log; h->jit_failed = true; }       // (a background build that failed: reported like a synchronous one)
            return st > 0 ? &e : nullptr;
        }
    }
    h->jit.emplace_back(new eh_handle_s::JitEntry());
    eh_handle_s::JitEntry* je = h->jit.back().get();
    je->arch = h->arch; je->variant = h->variant; je->fast = kf; je->spec = spec; je->p2p = want_p2p; je->net = h->net; je->loss_gen = lgen;
    // Only a kernel that merely REPLACES one built ahead of time may arrive later; a recorded closure / loss / per-net activation has no other form
    const bool async = h->specialize_async && !prog && !closs && h->act != EH_ACT_PER_NET && !want_p2p && !h->capturing;
    if (async) {
        const eh_model_desc desc = h->desc;
        const int act = h->act, device = h->device;
        je->worker = std::thread([je, desc, act, device]() {
            (void)hipSetDevice(device);
            const bool ok = eh_jit_build(desc, je->arch, je->variant, act, je->fast, &je->net, false, nullptr, &je->k, &je->log, /*allow_slp*/ false);
            je->state.store(ok ? 1 : -1, std::memory_order_release);
        });
        return nullptr;
    }
    const bool ok = eh_jit_build(h->desc, h->arch, h->variant, h->act, kf, spec ? &h->net : nullptr, want_p2p, closs ? &h->loss_prog : nullptr, &je->k, &je->log);
    je->state.store(ok ? 1 : -1, std::memory_order_release);
    if (!ok) { h->jit_log = je->log; h->jit_failed = true; return nullptr; }
    return je;
}
// A kernel compiled at run time that merely REPLACES one built ahead of time (the "specialize" option on a registry model) is checked
// against it before it takes over: both run the training pass on one window of the step's own data (no update, no state touched),
// their partial sums are reduced, and loss sum, counts and un-normalised gradient must agree to 1e-5 of the gradient's largest
// entry.  The two are the same source -- constants folded, dead branches gone -- but different binaries (hiprtc; the SLP vectoriser
// on for the one-block shapes, which miscompiled a sibling group of kernels in round 2: a loss sum lost in a packed accumulator),
// and the retry ladder of the build only catches compiler refusals, not silent miscompiles.  On disagreement the handle keeps the
// kernels built ahead of time and says so in eh_jit_status.  ~100 us, once per compiled kernel.
static int grid_for(const eh_handle* h, long long count);
static bool jit_verify(eh_handle* h, eh_handle_s::JitEntry* je, const EhStepArgs* a) {
    const EhNet& net = h->net;
    EhStepArgs v = *a;
    v.fz.gacc = nullptr;                         // the two-kernel form of the pass: one slab row per workgroup, nothing else written
    v.count = std::min<long long>(a->count, 8192);
    v.bn_update = 0; v.stamps = nullptr; v.slab = h->slab; v.n_acc = h->n_acc;
    v.yhat = nullptr; v.pout = nullptr;
    if (v.count <= 0) return true;
    const int grid = grid_for(h, v.count);
    const size_t nb = (size_t)h->n_acc * sizeof(float);
    float* dev = nullptr;
    if (hipMalloc(&dev, 2 * nb) != hipSuccess) { (void)hipGetLastError(); return true; }      // (cannot check: not a reason to refuse the kernel)
    std::vector<float> host(2 * (size_t)h->n_acc);
    float* sc = h->sc;                           // (read by nothing: APPLY = false)
    bool launched = true;
    for (int k = 0; k < 2 && launched; ++k) {
        hipError_t e = k == 0 ? h->arch->var[h->variant].launch(EH_MODE_TRAIN, h->act, KFAST(h), grid, h->stream, &net, &v)
                              : eh_jit_launch(&je->k, EH_MODE_TRAIN, grid, h->stream, &net, &v);
        if (e != hipSuccess) { (void)hipGetLastError(); launched = false; break; }
        hipLaunchKernelGGL((eh_reduce_kernel<false, 16>), dim3((h->n_acc + 15) / 16), dim3(256), 0, h->stream, h->slab, grid, h->n_acc, net.n_theta, net.T, 0,
                           dev + (size_t)k * h->n_acc, TH(h), MM(h), VV(h), sc, sc, h->opt, (float*)nullptr, h->img, net.loss, (const float*)nullptr, (const float*)nullptr, 0u);
        if (hipGetLastError() != hipSuccess) { launched = false; break; }
    }
    bool ok = true;
    std::string why;
    if (launched && hipMemcpyAsync(host.data(), dev, 2 * nb, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) {
        const float* A = host.data(); const float* B = A + h->n_acc;
        if (getenv("EH_DEBUG_JIT_SKEW")) host[(size_t)h->n_acc + net.n_theta] *= 1.01f;        // tests: the fall-back path without a second compiler run
        double gmax = 0.0, dmax = 0.0;
        for (int i = 0; i < net.n_theta; ++i) { gmax = std::max(gmax, (double)fabsf(A[i])); dmax = std::max(dmax, (double)fabsf(A[i] - B[i])); }
        const double sa = A[net.n_theta], sb = B[net.n_theta];
        bool counts = true;
        for (int t = 0; t < net.T; ++t) counts = counts && A[net.n_theta + 1 + t] == B[net.n_theta + 1 + t];
        const bool loss_ok = (std::isnan(sa) && std::isnan(sb)) || fabs(sa - sb) <= 1e-5 * fabs(sa) + 1e-30;      // (a window without a valid target: NaN on both sides)
        if (!(dmax <= 1e-5 * gmax + 1e-30) || !loss_ok || !counts || !std::isfinite(gmax)) {
            ok = false;
            char b[320];
            snprintf(b, sizeof b, "the kernel compiled at run time disagrees with the one built ahead of time on %lld samples of this model's data "
                     "(gradient: max |difference| %.3g against a largest entry of %.3g; loss sum %.9g vs %.9g; valid counts %s): the handle keeps the kernels built ahead of time",
                     v.count, dmax, gmax, sb, sa, counts ? "equal" : "DIFFERENT");
            why = b;
        }
    } else (void)hipGetLastError();
    (void)hipFree(dev);
    if (!ok) { je->state.store(-1); h->jit_failed = true; h->jit_log = why; }
    return ok;
}

// agg = mean: the data loss enters with 1 / (T_data (1 + E)), every entry of the extra loss with 1 / (1 + E) (EhImg); targets that stand
// for entries of the extra loss (eh_set_target_roles) are not data targets
static void eh_agg_factors(eh_handle* h) {
    int tdata = 0;
    for (int t = 0; t < h->net.T; ++t) tdata += ((h->roles >> (2 * t)) & 3u) == 0u ? 1 : 0;
    h->img.agg_a = h->agg ? 1.0f / ((float)std::max(1, tdata) * (float)(1 + h->n_extra)) : 1.0f;
    h->img.l2s = h->agg ? 1.0f / (float)(1 + h->n_extra) : 1.0f;
}

// the kernel specialised AHEAD OF TIME for this handle's descriptor, if it is one of the canonical ones (eh_spec.hip); nullptr otherwise
static const EhSpecKernel* spec_lookup(const eh_handle* h) {
    static const EhSpecKernel* const list[] = {
#define EH_SPEC_ITEM(k) eh_spec_##k(),
        EH_SPEC_LIST(EH_SPEC_ITEM)
#undef EH_SPEC_ITEM
    };
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per DEVICE: several handles of one process on different devices
    // (eh_p2p_init_local, eh_comm_init_local) each need the raised LDS limit (advisor, round 4: a per-process flag left every device but
    // the first with a kernel that failed to launch on every step and fell back silently)
    static std::mutex mu;
    static uint64_t prepared[sizeof list / sizeof list[0]] = {};          // bit d: prepared on device d
    if (!h->aot_spec || h->lform || h->act == EH_ACT_PER_NET) return nullptr;
    const EhArchInfo* A = h->arch;
    const EhVariant& V = A->var[h->variant];
    const int kf = KFAST(h);
    for (size_t i = 0; i < sizeof list / sizeof list[0]; ++i) {
        const EhSpecKernel* k = list[i];
        if (k->wide != (A->wide != 0) || k->bf16 != V.bf16 || k->nbi != A->nbi || k->nbh != A->nbh || k->nl != A->nl || k->nt != V.nt || k->nw != V.nw ||
            k->act != h->act || k->fast != kf || (k->so != 0) != (V.so != 0) || memcmp(&k->net, &h->net, sizeof(EhNet)) != 0) continue;
        {
            std::lock_guard<std::mutex> lk(mu);
            const uint64_t bit = 1ull << (h->device & 63);
            if (!(prepared[i] & bit)) {
                if (hipSetDevice(h->device) != hipSuccess || k->prepare() != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                prepared[i] |= bit;
            }
        }
        return k;
    }
    return nullptr;
}

static hipError_t step_launch(eh_handle* h, int mode, int grid, const EhStepArgs* a) {
    if (mode == EH_MODE_TRAIN_MULTI) {     // (built ahead of time only -- specialised for the canonical descriptors, generic otherwise; the caller checked: multi_ok)
        if (const EhSpecKernel* sk = spec_lookup(h)) {
            if (sk->launch(mode, grid, h->stream, &h->net, a) == hipSuccess) { h->spec_used = sk; return hipSuccess; }
            (void)hipGetLastError();
        } else if (jit_wanted(h, EH_MODE_TRAIN)) {      // the kernel compiled at run time, once it has been checked against the generic one (the single-step path does that)
            eh_handle_s::JitEntry* je = jit_entry(h);
            if (je && je->verified && je->k.fn[EH_MODE_TRAIN_MULTI]) {
                if (eh_jit_launch(&je->k, mode, grid, h->stream, &h->net, a) == hipSuccess) return hipSuccess;
                (void)hipGetLastError();
            }
        }
        return h->arch->var[h->variant].launch(mode, h->act, KFAST(h), grid, h->stream, &h->net, a);
    }
    if (const EhSpecKernel* sk = spec_lookup(h)) {
        const hipError_t e = sk->launch(mode, grid, h->stream, &h->net, a);
        if (e == hipSuccess) { h->spec_used = sk; return e; }
        (void)hipGetLastError();                 // (a mode the specialised unit does not hold: the paths below)
    }
    if (jit_wanted(h, mode)) {
        eh_handle_s::JitEntry* je = jit_entry(h);
        if (je && !je->verified && mode != EH_MODE_EVAL && !h->capturing && je->spec && h->net.mech != EH_MECH_PROGRAM && h->net.loss != EH_LOSS_PROGRAM &&
            h->act != EH_ACT_PER_NET && !getenv("EH_JIT_NO_VERIFY")) {
            bool has_lprog = false;
            for (int t = 0; t < h->net.T; ++t) has_lprog = has_lprog || ((h->net.loss_t >> (4 * t)) & 15u) == (unsigned)EH_LOSS_PROGRAM;
            if (has_lprog || jit_verify(h, je, a)) je->verified = true; else je = nullptr;
        }
        if (je) {
            const hipError_t e = eh_jit_launch(&je->k, mode, grid, h->stream, &h->net, a);
            if (e == hipSuccess) return e;
            (void)hipGetLastError();
            je->state.store(-1); h->jit_failed = true;
            h->jit_log = std::string("launch of the run-time compiled kernel failed: ") + hipGetErrorString(e);
        }
        if (h->net.loss == EH_LOSS_PROGRAM && mode != EH_MODE_EVAL) return hipErrorNotSupported;     // no other form of a recorded loss exists
    }
    if (h->act == EH_ACT_PER_NET) return hipErrorNotSupported;       // (the table below has no such kernel; eh_create made sure the compiled one exists)
    return h->arch->var[h->variant].launch(mode, h->act, KFAST(h), grid, h->stream, &h->net, a);
}

extern "C" {

int32_t eh_version(void) { return EH_ABI_VERSION; }

const char* eh_last_error(const eh_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

// Streams of destroyed handles are kept for the next handle on the same device: creating one takes 1.4-1.9 ms and destroying it as long
// again -- a fifth of a whole train() call on the reference's tutorial data set, which creates and destroys one engine per call.  At most
// eight per device are kept (more are destroyed); the kept ones live until the process ends.
static std::mutex g_stage_mu;                 // eh_set_data's pinned staging pair (host memory: any device)
static float* g_stage[2] = {nullptr, nullptr};
static size_t g_stage_bytes = 0;
static int copy_out(eh_handle* h, float* dst, const float* src_dev, size_t n);
static int copy_in(eh_handle* h, float* dst_dev, const float* src, size_t n);
static std::mutex g_stream_pool_mu;
static std::vector<std::pair<int, hipStream_t>> g_stream_pool;
static hipStream_t stream_pool_take(int device) {
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); ++i)
        if (g_stream_pool[i].first == device) { hipStream_t s = g_stream_pool[i].second; g_stream_pool.erase(g_stream_pool.begin() + (long)i); return s; }
    return nullptr;
}
static void stream_pool_give(int device, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        int kept = 0;
        for (auto& e : g_stream_pool) kept += e.first == device;
        if (kept < 8 && !getenv("EH_NO_STREAM_POOL")) { g_stream_pool.emplace_back(device, s); return; }
    }
    (void)hipStreamDestroy(s);
}

int32_t eh_create(const eh_model_desc* d, eh_handle** out) {
    if (!d || !out) return fail(nullptr, EH_EINVAL, "eh_create: null argument");
    *out = nullptr;
    if (d->struct_size != (int32_t)sizeof(eh_model_desc)) return fail(nullptr, EH_EINVAL, "eh_create: struct_size %d != %zu", d->struct_size, sizeof(eh_model_desc));
    MechInfo mi;
    if (!mech_info(d->mech, &mi, d)) return fail(nullptr, EH_EUNSUPPORTED, "eh_create: unknown mechanistic model id %d (no silent fallback)", d->mech);
    if (d->mech == EH_MECH_PROGRAM) {
        // every operand must name a slot that holds a value when the instruction runs: the kernel indexes a per-lane array with them
        if (d->n_params < 1 || d->n_params > EH_MAX_PARAMS) return fail(nullptr, EH_EINVAL, "eh_create: a program takes 1..%d parameters, descriptor has %d", EH_MAX_PARAMS, d->n_params);
        if (d->prog_len < 1 || d->prog_len > EH_MAX_PROG) return fail(nullptr, EH_EUNSUPPORTED, "eh_create: program of %d instructions (1..%d)", d->prog_len, EH_MAX_PROG);
        if (d->prog_n_const < 0 || d->prog_n_const > EH_MAX_PROG_CONST) return fail(nullptr, EH_EUNSUPPORTED, "eh_create: program with %d constants (0..%d)", d->prog_n_const, EH_MAX_PROG_CONST);
        if (d->prog_n_forc < 0 || d->prog_n_forc > EH_MAX_FORC) return fail(nullptr, EH_EINVAL, "eh_create: program with %d forcings (0..%d)", d->prog_n_forc, EH_MAX_FORC);
        if (d->prog_n_out < 1 || d->prog_n_out > EH_MAX_PROG_OUT) return fail(nullptr, EH_EUNSUPPORTED, "eh_create: program with %d outputs (1..%d)", d->prog_n_out, EH_MAX_PROG_OUT);
        auto slot_ok = [&](unsigned sl, int upto) {
            if (sl < EH_PROG_SLOT_FORC) return (int)sl < d->n_params;
            if (sl < EH_PROG_SLOT_CONST) return (int)sl - EH_PROG_SLOT_FORC < d->prog_n_forc;
            if (sl < EH_PROG_SLOT_INSTR) return (int)sl - EH_PROG_SLOT_CONST < d->prog_n_const;
            return (int)sl - EH_PROG_SLOT_INSTR < upto;
        };
        for (int i = 0; i < d->prog_len; ++i) {
            const unsigned w = d->prog_code[i], op = w & 255u;
            if (op >= EH_OP_COUNT) return fail(nullptr, EH_EUNSUPPORTED, "eh_create: program instruction %d has unknown opcode %u", i, op);
            const int nop = (op == EH_OP_SELECT) ? 3 : (op == EH_OP_NEG || op == EH_OP_EXP || op == EH_OP_LOG || op == EH_OP_SQRT || op == EH_OP_TANH ||
                                                        op == EH_OP_SIGMOID || op == EH_OP_ABS || op == EH_OP_SIN || op == EH_OP_COS) ? 1 : 2;
            const unsigned sl[3] = {(w >> 8) & 255u, (w >> 16) & 255u, w >> 24};
            for (int k = 0; k < 3; ++k) {
                if (k < nop ? !slot_ok(sl[k], i) : sl[k] != 0u)
                    return fail(nullptr, EH_EINVAL, "eh_create: program instruction %d, operand %d names slot %u (undefined at that point, or a non-zero unused operand)", i, k, sl[k]);
            }
        }
        for (int o = 0; o < d->prog_n_out; ++o)
            if (d->prog_out[o] < 0 || !slot_ok((unsigned)d->prog_out[o], d->prog_len)) return fail(nullptr, EH_EINVAL, "eh_create: program output %d names slot %d", o, d->prog_out[o]);
    }
    if (d->activation < 0 || d->activation > EH_ACT_PER_NET) return fail(nullptr, EH_EUNSUPPORTED, "eh_create: unknown activation id %d", d->activation);
    int act = d->activation;
    if (act == EH_ACT_PER_NET) {
        // n_nets >= 1: activation::NamedTuple of the MultiNN constructor (GenericHybridModel.jl:168-176), one activation per network;
        // n_nets == 0: `hidden_layers::Chain` of the single-network constructor (NNModels.jl:145-219: Dense layers that carry activations of
        // their own), one activation per hidden LAYER.  Both run on kernels compiled at run time around eh_row_act(layer, row).
        if (d->n_nets < 0 || d->n_nets > EH_MAX_NETS) return fail(nullptr, EH_EINVAL, "eh_create: n_nets = %d (0..%d)", d->n_nets, EH_MAX_NETS);
        const int na = d->n_nets > 0 ? d->n_nets : d->n_hidden;
        if (na < 1 || na > EH_MAX_HIDDEN) return fail(nullptr, EH_EINVAL, "eh_create: per-layer activations need 1..%d hidden layers (n_hidden = %d)", EH_MAX_HIDDEN, d->n_hidden);
        bool same = true;
        for (int k = 0; k < na; ++k) {
            if (d->net_activation[k] < 0 || d->net_activation[k] > EH_ACT_IDENTITY)
                return fail(nullptr, EH_EUNSUPPORTED, "eh_create: unknown activation id %d for %s %d", d->net_activation[k], d->n_nets > 0 ? "net" : "hidden layer", k);
            same = same && d->net_activation[k] == d->net_activation[0];
        }
        if (same) act = d->net_activation[0];          // one activation after all: the kernels built ahead of time
    }
    if (d->n_params != mi.n_par) return fail(nullptr, EH_EINVAL, "eh_create: model %d takes %d parameters, descriptor has %d", d->mech, mi.n_par, d->n_params);
    if (d->n_predictors < 0 || d->n_hidden < 0 || d->n_hidden > EH_MAX_HIDDEN) return fail(nullptr, EH_EINVAL, "eh_create: n_predictors %d, n_hidden %d (0..%d)", d->n_predictors, d->n_hidden, EH_MAX_HIDDEN);
    if (d->n_forcings < mi.n_forc || d->n_forcings > EH_MAX_FORC) return fail(nullptr, EH_EINVAL, "eh_create: n_forcings %d (model needs %d, max %d)", d->n_forcings, mi.n_forc, EH_MAX_FORC);
    if (d->n_targets < 1 || d->n_targets > EH_MAX_TARG) return fail(nullptr, EH_EINVAL, "eh_create: n_targets must be 1..%d", EH_MAX_TARG);
    int K = 0, G = 0, maxw = 0;
    bool seenK[EH_MAX_PARAMS] = {}, seenG[EH_MAX_PARAMS] = {};
    for (int j = 0; j < d->n_params; ++j) {
        const int k = d->param_kind[j], ix = d->param_index[j];
        if (k == EH_PAR_NEURAL || k == EH_PAR_GLOBAL) {
            if (ix < 0 || ix >= EH_MAX_PARAMS) return fail(nullptr, EH_EINVAL, "eh_create: param_index[%d] = %d out of range", j, ix);
            bool* seen = k == EH_PAR_NEURAL ? seenK : seenG;
            if (seen[ix]) return fail(nullptr, EH_EINVAL, "eh_create: duplicate param_index %d", ix);
            seen[ix] = true;
            (k == EH_PAR_NEURAL ? K : G)++;
            if (!(d->param_upper[j] > d->param_lower[j]) && (k == EH_PAR_GLOBAL || d->scale_nn_outputs))
                return fail(nullptr, EH_EINVAL, "eh_create: parameter %d needs upper > lower for sigmoid scaling", j);
        } else if (k != EH_PAR_FIXED) {
            return fail(nullptr, EH_EINVAL, "eh_create: param_kind[%d] = %d", j, k);
        }
    }
    for (int i = 0; i < K; ++i) if (!seenK[i]) return fail(nullptr, EH_EINVAL, "eh_create: neural param_index values must be 0..K-1");
    for (int i = 0; i < G; ++i) if (!seenG[i]) return fail(nullptr, EH_EINVAL, "eh_create: global param_index values must be 0..G-1");
    // No neural parameter: the reference builds no network at all (`NN = Chain()`, GenericHybridModel.jl:112-125) and its forward
    // is global / fixed parameters -> M (:376-406).  Here: the layer-wise form with zero Dense layers (mechanistic stage + reduce).
    const bool no_nn = K == 0;
    if (no_nn) {
        if (G < 1) return fail(nullptr, EH_EINVAL, "eh_create: a model with neither neural nor global parameters has nothing to train");
        if (d->n_hidden != 0 || d->n_nets != 0 || d->input_batchnorm) return fail(nullptr, EH_EINVAL, "eh_create: no neural parameter: n_hidden, n_nets and input_batchnorm must be 0");
    } else {
        if (d->n_predictors < 1) return fail(nullptr, EH_EINVAL, "eh_create: n_predictors must be >= 1");
        if (d->n_hidden < 1) return fail(nullptr, EH_EINVAL, "eh_create: n_hidden must be 1..%d", EH_MAX_HIDDEN);
    }
    // nets and their block placement (SingleNN = one net with K outputs)
    const int nl = d->n_hidden;
    int n_nets = 1, net_P[EH_MAX_NETS] = {0}, net_K[EH_MAX_NETS] = {0}, net_w[EH_MAX_NETS][EH_MAX_HIDDEN] = {{0}}, tot_w[EH_MAX_HIDDEN] = {0};
    int net_d[EH_MAX_NETS];                  // hidden layers of net k; layers past them are identity blocks as wide as its last one
    for (int k = 0; k < EH_MAX_NETS; ++k) net_d[k] = nl;
    bool mixed_depth = false;
    if (d->n_nets < 0 || d->n_nets > EH_MAX_NETS) return fail(nullptr, EH_EINVAL, "eh_create: n_nets = %d (0..%d)", d->n_nets, EH_MAX_NETS);
    if (d->n_nets > 0) {
        if (d->n_nets != K) return fail(nullptr, EH_EINVAL, "eh_create: MultiNN needs one net per neural parameter (%d nets, %d neural parameters)", d->n_nets, K);
        n_nets = d->n_nets;
        int ptot = 0;
        for (int k = 0; k < n_nets; ++k) {
            if (d->net_n_predictors[k] < 1) return fail(nullptr, EH_EINVAL, "eh_create: net %d has no predictors", k);
            net_P[k] = d->net_n_predictors[k]; net_K[k] = 1; ptot += net_P[k];
            if (d->net_depth[k] < 0 || d->net_depth[k] > nl) return fail(nullptr, EH_EINVAL, "eh_create: net_depth[%d] = %d (n_hidden = %d)", k, d->net_depth[k], nl);
            if (d->net_depth[k] > 0) net_d[k] = d->net_depth[k];
            mixed_depth = mixed_depth || net_d[k] != nl;
            for (int l = 0; l < nl; ++l) {
                if (l < net_d[k] && d->net_hidden[k][l] < 1) return fail(nullptr, EH_EINVAL, "eh_create: net_hidden[%d][%d] = %d", k, l, d->net_hidden[k][l]);
                net_w[k][l] = d->net_hidden[k][l < net_d[k] ? l : net_d[k] - 1]; tot_w[l] += net_w[k][l];
            }
        }
        if (mixed_depth) {
            bool full = false;
            for (int k = 0; k < n_nets; ++k) full = full || net_d[k] == nl;
            if (!full) return fail(nullptr, EH_EINVAL, "eh_create: n_hidden = %d but no net is that deep", nl);
            act = EH_ACT_PER_NET;            // the identity blocks need the row-dependent activation (kernels compiled at run time)
        }
        if (ptot != d->n_predictors) return fail(nullptr, EH_EINVAL, "eh_create: net_n_predictors sum to %d, n_predictors = %d", ptot, d->n_predictors);
    } else {
        net_P[0] = d->n_predictors; net_K[0] = K;
        for (int l = 0; l < nl; ++l) {
            if (d->hidden[l] < 1) return fail(nullptr, EH_EINVAL, "eh_create: hidden[%d] = %d", l, d->hidden[l]);
            net_w[0][l] = d->hidden[l]; tot_w[l] = d->hidden[l];
        }
    }
    for (int l = 0; l < nl; ++l) maxw = std::max(maxw, tot_w[l]);
    for (int f = 0; f < mi.n_forc; ++f)
        if (d->forcing_index[f] < 0 || d->forcing_index[f] >= d->n_forcings) return fail(nullptr, EH_EINVAL, "eh_create: forcing_index[%d] out of range", f);
    for (int t = 0; t < d->n_targets; ++t)
        if (d->target_output[t] < 0 || d->target_output[t] >= mi.n_out) return fail(nullptr, EH_EINVAL, "eh_create: target_output[%d] = %d (model has %d outputs)", t, d->target_output[t], mi.n_out);
    const int nbi = (d->n_predictors + 15) / 16, nbh_raw = (maxw + 15) / 16;
    const int nbh = nbh_raw <= 1 ? 1 : nbh_raw <= 2 ? 2 : nbh_raw <= 4 ? 4 : nbh_raw <= 8 ? 8 : 0;
    const EhArchInfo* arch = (nbh && K <= 16 && !no_nn) ? find_arch(nbi, nbh, d->n_hidden) : nullptr;
    const EhArchInfo* const wide_arch = (nbh && K <= 16 && !no_nn) ? find_wide(nbi, nbh, d->n_hidden) : nullptr;
    if (!arch) arch = wide_arch;
    bool lform = false;
    if (no_nn) { arch = &g_lform_arch; lform = true; }
    else if (!arch) {
        // no fused kernel holds this network: run it layer by layer (eh_lform.hpp) where that form is built
        // (every activation, SingleNN and MultiNN alike: the layer-wise form runs each network as its own chain of products)
        if (K > 16 || (d->input_batchnorm && d->n_predictors > 32))
            return fail(nullptr, EH_EUNSUPPORTED, "eh_create: no kernel for P=%d, hidden max width %d%s, %d hidden layers, K=%d, activation %d (fused kernels: P<=32, K<=16, "
                        "width<=64 with <=3 layers or width<=128 with <=2; layer-wise form: K<=16, input BatchNorm with P<=32)",
                        d->n_predictors, maxw, d->n_nets > 0 ? " (nets side by side)" : "", d->n_hidden, K, act);
        arch = &g_lform_arch;
        lform = true;
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, EH_EHIP, "eh_create: no HIP device (this library has no CPU path)");
    if (d->device < 0 || d->device >= ndev) return fail(nullptr, EH_EINVAL, "eh_create: device %d of %d", d->device, ndev);

    eh_handle* h = new eh_handle_s();
    h->desc = *d;
    if (d->n_nets > 0) {                     // the descriptor the run-time compiler reads: depths and identity-block widths spelled out
        for (int k = 0; k < n_nets; ++k) {
            h->desc.net_depth[k] = net_d[k];
            for (int l = 0; l < nl; ++l) h->desc.net_hidden[k][l] = net_w[k][l];
            if (act == EH_ACT_PER_NET) h->desc.net_activation[k] = d->activation == EH_ACT_PER_NET ? d->net_activation[k] : d->activation;
        }
        if (act == EH_ACT_PER_NET) h->desc.activation = EH_ACT_PER_NET;
    }
    h->device = d->device;
    h->arch = arch;
    h->lform = lform;
    h->variant = (arch->nvar > 1 && !arch->wide) ? 1 : 0;   // narrow nets: two waves per SIMD hide the latency of the short tile
    EhNet& n = h->net;
    memset(&n, 0, sizeof n);
    n.P = d->n_predictors; n.K = K; n.G = G; n.T = d->n_targets; n.F = d->n_forcings;
    h->n_nets = n_nets;
    {
        int c0 = 0, r0[EH_MAX_HIDDEN] = {0};
        for (int k = 0; k < n_nets; ++k) {
            h->net_P[k] = net_P[k]; h->net_K[k] = net_K[k]; h->net_c0[k] = c0; c0 += net_P[k];
            for (int l = 0; l < nl; ++l) { h->net_w[k][l] = net_w[k][l]; h->net_r0[k][l] = r0[l]; r0[l] += net_w[k][l]; }
            h->net_r0[k][nl] = d->n_nets > 0 ? k : 0;            // output row
            h->net_d[k] = net_d[k];
        }
        for (int l = 0; l < nl; ++l) h->tot_w[l] = tot_w[l];
    }
    int lw_off[EH_MAX_HIDDEN + 1], lb_off[EH_MAX_HIDDEN + 1];     // canonical offsets of net 0 (all there is for SingleNN)
    int off = 0;
    for (int k = 0; k < (no_nn ? 0 : n_nets); ++k) {
        int in = net_P[k];
        for (int l = 0; l <= nl; ++l) {
            const int o = l < nl ? net_w[k][l] : net_K[k];
            if (k == 0) lw_off[l] = off;
            if (l < nl && l >= net_d[k]) { if (k == 0) lb_off[l] = off; continue; }       // identity block: nothing of it in theta
            off += o * in;
            if (k == 0) lb_off[l] = off;
            off += o;
            in = o;
        }
    }
    n.g_off = off;
    n.n_theta = off + G;
    if (lform && !no_nn) {    // Dense layers of every network: shapes and canonical offsets (no network: l_nnets stays 0)
        h->l_nnets = n_nets;
        int o2 = 0, c0 = 0;
        for (int k = 0; k < n_nets; ++k) {
            eh_handle_s::LNet& L = h->l_net[k];
            L.nl = net_d[k] + 1; L.c0 = c0; L.orow = d->n_nets > 0 ? k : 0;
            L.act = (d->n_nets > 0 && d->activation == EH_ACT_PER_NET) ? d->net_activation[k] : (act == EH_ACT_PER_NET ? d->activation : act);
            for (int l = 0; l <= EH_MAX_HIDDEN; ++l)      // (an activation per hidden layer: the single network of a `hidden_layers::Chain`)
                L.lact[l] = (d->n_nets == 0 && act == EH_ACT_PER_NET && l < nl) ? d->net_activation[l] : L.act;
            int in = net_P[k];
            for (int l = 0; l < L.nl; ++l) {
                const int o = l + 1 < L.nl ? net_w[k][l] : net_K[k];
                L.in[l] = in; L.out[l] = o; L.woff[l] = o2; L.boff[l] = o2 + o * in;
                o2 += o * in + o; in = o;
            }
            c0 += net_P[k];
        }
    }
    h->act = act; n.scale_nn = d->scale_nn_outputs ? 1 : 0;
    n.mech = d->mech; n.n_par = d->n_params;
    for (int j = 0; j < d->n_params; ++j) {
        n.par_kind |= (unsigned)d->param_kind[j] << (2 * j);
        n.par_idx |= (unsigned)(d->param_kind[j] == EH_PAR_FIXED ? 0 : d->param_index[j]) << (4 * j);
    }
    for (int j = d->n_params; j < EH_MAX_PARAMS; ++j) n.par_kind |= (unsigned)EH_PAR_FIXED << (2 * j);
    n.forc_col = 0xFFFFFFFFu;
    for (int f = 0; f < mi.n_forc; ++f) n.forc_col = (n.forc_col & ~(0xFFu << (8 * f))) | ((unsigned)d->forcing_index[f] << (8 * f));
    n.n_out = mi.n_out;
    for (int t = 0; t < d->n_targets; ++t) n.targ_out |= (unsigned)d->target_output[t] << (2 * t);
    h->fast = fast_wanted(arch, K, n.P, n.T, d->mech);
    h->C = n.P + n.F + n.T;
    h->n_par = d->n_params;
    h->n_acc = n.n_theta + 1 + n.T + 2;      // [grad | S | n_valid per target | Sy | Syy]
    if (!lform && !arch_fits(arch, std::max(h->n_acc, EH_EVAL_STATS * n.T))) {
        // the per-wave kernel parks every wave's accumulators in LDS to sum them; the row-split kernel needs no such sum
        if (!wide_arch) {
            delete h;
            return fail(nullptr, EH_EUNSUPPORTED, "eh_create: %d accumulators exceed the kernel's reduction space", n.n_theta + 1 + n.T);
        }
        h->arch = arch = wide_arch;
        h->variant = 0;
        h->fast = 0;
    }
    h->arch_alt = arch == wide_arch ? nullptr : wide_arch;
#define HIPCHK_C(expr)                                                            \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            fail(nullptr, e_ == hipErrorOutOfMemory ? EH_ENOMEM : EH_EHIP, "eh_create: %s: %s", #expr, hipGetErrorString(e_)); \
            eh_destroy(h);                                                        \
            return e_ == hipErrorOutOfMemory ? EH_ENOMEM : EH_EHIP;               \
        }                                                                         \
    } while (0)
    const bool ctrace = getenv("EH_CREATE_TRACE") != nullptr;          // diagnostics: where eh_create's milliseconds go
    auto ct0 = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!ctrace) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[eh_create] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - ct0).count());
        ct0 = t;
    };
    HIPCHK_C(hipSetDevice(h->device));
    tick("hipSetDevice");
    if (!(h->own_stream = stream_pool_take(h->device))) HIPCHK_C(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    tick("stream");
    h->stream = h->own_stream;
    for (int vi = 0; vi < arch->nvar; ++vi) HIPCHK_C(arch->var[vi].prepare());
    tick("kernel attributes");
    if (const char* ej = getenv("EH_JIT")) h->jit_on = atoi(ej) != 0;
    if (const char* es = getenv("EH_SPECIALIZE")) h->specialize = atoi(es) != 0;      // (test runs: the whole suite on specialised kernels)
    if (getenv("EH_NO_AOT_SPEC")) h->aot_spec = false;                                // (A/B: the generic / run-time compiled kernels for a canonical descriptor)
    if (d->mech == EH_MECH_PROGRAM) {
        std::vector<unsigned> pb(EH_PROG_HDR + EH_MAX_PROG, 0u);
        pb[0] = (unsigned)d->prog_len; pb[1] = (unsigned)d->prog_n_out;
        for (int o = 0; o < EH_MAX_PROG_OUT; ++o) pb[2 + o] = (unsigned)(o < d->prog_n_out ? d->prog_out[o] : d->prog_out[0]);
        for (int k = 0; k < d->prog_n_const; ++k) memcpy(&pb[8 + k], &d->prog_const[k], sizeof(float));
        for (int i = 0; i < d->prog_len; ++i) pb[EH_PROG_HDR + i] = d->prog_code[i];
        HIPCHK_C(hipMalloc(&h->prog, pb.size() * sizeof(unsigned)));
        HIPCHK_C(hipMemcpy(h->prog, pb.data(), pb.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    const size_t nt = (size_t)n.n_theta;
    // one allocation: [2][3][n_theta] parameter sets {theta, m, v}, then the [2][2] running beta products
    HIPCHK_C(hipMalloc(&h->pset, (6 * nt + 4) * sizeof(float)));
    HIPCHK_C(hipMemset(h->pset, 0, (6 * nt + 4) * sizeof(float)));
    for (int k = 0; k < 2; ++k) { h->thb[k] = h->pset + (size_t)k * 3 * nt; h->mb[k] = h->thb[k] + nt; h->vb[k] = h->thb[k] + 2 * nt; }
    h->sc = h->pset + 6 * nt;
    h->bn_on = d->input_batchnorm != 0;
    if (h->bn_on) {
        HIPCHK_C(hipMalloc(&h->bn_part, (32 * 64 + 32) * sizeof(float)));
        HIPCHK_C(hipMalloc(&h->bn_run, 64 * sizeof(float)));
        HIPCHK_C(hipMalloc(&h->bn_shift, 32 * sizeof(float)));
        HIPCHK_C(hipMemset(h->bn_shift, 0, 32 * sizeof(float)));
        HIPCHK_C(hipMalloc(&h->bn_stat, 68 * sizeof(float)));
        HIPCHK_C(hipMemset(h->bn_stat, 0, 68 * sizeof(float)));
        float run0[64];
        for (int p = 0; p < 32; ++p) { run0[p] = 0.0f; run0[32 + p] = 1.0f; }     // LuxCore.initialstates(BatchNorm)
        HIPCHK_C(hipMemcpy(h->bn_run, run0, sizeof run0, hipMemcpyHostToDevice));
    }
    if (!lform) {             // (the fused-update accumulators: that mode exists for the per-wave kernels only)
        HIPCHK_C(hipMalloc(&h->gacc, ((size_t)3 * EH_GSHARDS * h->n_acc + 4) * sizeof(float)));      // (+4: the fused prologue reads five tail floats of every shard whatever T is)
        HIPCHK_C(hipMemset(h->gacc, 0, ((size_t)3 * EH_GSHARDS * h->n_acc + 4) * sizeof(float)));
    }
    tick("pset / bn / gacc");
    h->slab_rows = lform ? (int)EH_LFORM_ROWS : h->max_blocks;
    HIPCHK_C(hipMalloc(&h->slab, (std::max((size_t)h->slab_rows * std::max(h->n_acc, EH_EVAL_STATS * n.T), (size_t)1 << 20) + 16) * sizeof(float)));      // (>= 4 MB: the evaluation passes park their per-workgroup metric sums here)
    HIPCHK_C(hipMalloc(&h->gradbuf, (size_t)h->n_acc * sizeof(float)));
    HIPCHK_C(hipMalloc(&h->mombuf, EH_MAX_TARG * EH_EVAL_STATS * sizeof(float)));
    HIPCHK_C(hipMemset(h->mombuf, 0, EH_MAX_TARG * EH_EVAL_STATS * sizeof(float)));
    HIPCHK_C(hipMalloc(&h->tcount, 3 * EH_MAX_TARG * sizeof(float)));
    HIPCHK_C(hipMemset(h->tcount, 0, 3 * EH_MAX_TARG * sizeof(float)));
    HIPCHK_C(hipMalloc(&h->inv_n, EH_TT * EH_MAX_TARG * sizeof(float)));      // the per-target table (EhStepArgs::inv_n): weight, centre of yhat, k0 k1 k2, loss of every target
    HIPCHK_C(hipMemset(h->inv_n, 0, EH_TT * EH_MAX_TARG * sizeof(float)));
    HIPCHK_C(hipMemset(h->gradbuf, 0, (size_t)h->n_acc * sizeof(float)));
    tick("slab + small buffers");
    if (!lform) { if (int rc = build_maps(h, true)) { g_create_err = h->err; eh_destroy(h); return rc; } }
    else {
        std::vector<unsigned char> wf((size_t)n.n_theta, 0);
        for (int k = 0; k < h->l_nnets; ++k)
            for (int l = 0; l < h->l_net[k].nl; ++l) std::fill(wf.begin() + h->l_net[k].woff[l], wf.begin() + h->l_net[k].boff[l], (unsigned char)1);
        HIPCHK_C(hipMalloc(&h->wflag, wf.size()));
        HIPCHK_C(hipMemcpy(h->wflag, wf.data(), wf.size(), hipMemcpyHostToDevice));
    }
    tick("maps");
    {   // parameter image (constant parts; theta is mirrored into it by eh_image_kernel / the optimiser)
        std::vector<float> img0((size_t)arch->img_floats, 0.0f);
        for (int j = 0; j < d->n_params; ++j) {
            if (d->param_kind[j] == EH_PAR_FIXED) img0[arch->phi_off + EH_IMG_PHI + j] = d->param_default[j];   // st.fixed, GenericHybridModel.jl:289-303
            img0[arch->phi_off + EH_IMG_LO + j] = d->param_lower[j];
            img0[arch->phi_off + EH_IMG_SC + j] = d->param_upper[j] - d->param_lower[j];
        }
        for (int p = 0; p < 32; ++p) { img0[arch->phi_off + EH_IMG_BNM + p] = 0.0f; img0[arch->phi_off + EH_IMG_BNR + p] = d->input_batchnorm ? 1.0f / std::sqrt(1.0f + EH_BN_EPS) : 1.0f; }
        for (int k = 0; k < (lform ? 0 : n_nets); ++k)      // identity blocks that carry a shallower net to the output layer (fused envelope only: the layer-wise form runs every net at its own depth)
            for (int l = net_d[k]; l < nl; ++l)
                for (int i = 0; i < net_w[k][l]; ++i)
                    img0[arch->wh_off + (size_t)(l - 1) * arch->hp * arch->sh + (size_t)(h->net_r0[k][l] + i) * arch->sh + h->net_r0[k][l - 1] + i] = 1.0f;
        auto put_int = [&](int slot, int v) { memcpy(&img0[arch->phi_off + slot], &v, sizeof(int)); };
        if (!lform) {         // (read by the fused kernels' end-of-kernel reduction only; sized for their four layers)
            for (int l = 0; l <= d->n_hidden; ++l) { put_int(EH_IMG_WOFF + l, lw_off[l]); put_int(EH_IMG_BOFF + l, lb_off[l]); }
            for (int l = 0; l < d->n_hidden; ++l) put_int(EH_IMG_WIDTH + l, tot_w[l]);
        }
        for (int j = 0; j < d->n_params; ++j)
            if (d->param_kind[j] == EH_PAR_GLOBAL) put_int(EH_IMG_GPAR + d->param_index[j], j);
        HIPCHK_C(hipMalloc(&h->image, img0.size() * sizeof(float)));
        HIPCHK_C(hipMemcpy(h->image, img0.data(), img0.size() * sizeof(float), hipMemcpyHostToDevice));
        EhImg& im = h->img;
        im.image = h->image; im.imap = h->imap; im.g_off = n.g_off; im.phi_off = arch->phi_off;
        im.l2c = 0.0f; im.l2w = nullptr; im.n_theta = n.n_theta; im.b_off = arch->b_off; im.wflag = h->wflag;
        im.agg_a = 1.0f; im.l2s = 1.0f;          // agg = sum
        HIPCHK_C(hipMalloc(&h->l2val, sizeof(float)));
        HIPCHK_C(hipMemset(h->l2val, 0, sizeof(float)));
        {
            int nw = 0;
            if (lform) { for (int k = 0; k < h->l_nnets; ++k) for (int l = 0; l < h->l_net[k].nl; ++l) nw += h->l_net[k].in[l] * h->l_net[k].out[l]; }
            else for (const EhEntry& e : enumerate_entries(h)) nw += e.col >= 0 ? 1 : 0;
            h->n_weights = nw;
        }
        for (int j = 0; j < d->n_params; ++j)
            if (d->param_kind[j] == EH_PAR_GLOBAL) {
                const int g = d->param_index[j];
                im.glob_par[g] = j; im.glo[g] = d->param_lower[j]; im.ghi[g] = d->param_upper[j];
            }
        tick("image");
        hipLaunchKernelGGL(eh_image_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream, TH(h), (int)nt, h->img);
        HIPCHK_C(hipGetLastError());
        tick("image kernel launch");
        HIPCHK_C(hipStreamSynchronize(h->stream));
        tick("synchronize");
    }
#undef HIPCHK_C
    if (h->act == EH_ACT_PER_NET && !h->lform && !jit_entry(h)) {         // built now, so that a missing run-time compiler is an error of the constructor
        const std::string log = h->jit_log;
        eh_destroy(h);
        return fail(nullptr, EH_EUNSUPPORTED, "eh_create: per-net activations need the run-time compiled kernel, which failed to build: %.600s", log.c_str());
    }
    *out = h;
    return EH_OK;
}

int32_t eh_destroy(eh_handle* h) {
    if (!h) return EH_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto& e : h->jit) { if (e->worker.joinable()) e->worker.join(); eh_jit_release(&e->k); }
    eh_comm_release(h);             // communicator / local group / peer-to-peer mappings and buffers (eh_comm.hip)
    (void)hipSetDevice(h->device);
    (void)hipFree(h->pset);
    (void)hipFree(h->gacc); (void)hipFree(h->bn_part); (void)hipFree(h->bn_run); (void)hipFree(h->bn_shift); (void)hipFree(h->bn_stat); (void)hipFree(h->tcount); (void)hipFree(h->mombuf); (void)hipFree(h->slab); (void)hipFree(h->gradbuf); (void)hipFree(h->inv_n);
    (void)hipFree(h->prog); (void)hipFree(h->l2val); (void)hipFree(h->l2w); (void)hipFree(h->loss_hist); (void)hipFree(h->perm); (void)hipFree(h->out_buf); (void)hipFree(h->idx_buf);
    eval_host_release(h);
    (void)hipFree(h->mech_ws); (void)hipFree(h->l_ws); (void)hipFree(h->l_split); (void)hipFree(h->l_dk); (void)hipFree(h->l_lprog); (void)hipFree(h->wflag);
    (void)hipFree(h->stamps); (void)hipFree(h->image); (void)hipFree(h->imap); (void)hipFree(h->rmap);
    (void)hipFree(h->split[0].recs); (void)hipFree(h->split[1].recs);
    if (h->own_stream) stream_pool_give(h->device, h->own_stream);      // (drained above; the next handle on this device takes it over)
    delete h;
    return EH_OK;
}

int32_t eh_n_theta(const eh_handle* h, int64_t* n) {
    if (!h || !n) return EH_EINVAL;
    *n = h->net.n_theta;
    return EH_OK;
}

int32_t eh_set_stream(eh_handle* h, void* s) {
    if (!h) return EH_EINVAL;
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return EH_OK;
}

int32_t eh_synchronize(eh_handle* h) {
    if (!h) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));      // (polling hipStreamQuery instead was measured and is slower: 12.1-13.4 against 11.4-11.6 us per step in the 20-step run)
    if (h->p2p_on) {
        unsigned c[3] = {0, 0, 0};
        HIPCHK(h, hipMemcpy(c, h->p2p_ctr, sizeof c, hipMemcpyDeviceToHost));
        if (c[1]) return fail(h, EH_EHIP, "eh_synchronize: a cross-GPU exchange ran into its 2 s deadline (a rank is missing or out of step); results are invalid");
    }
    return EH_OK;
}

// target < 0: the program of every target that has none of its own; 0 <= target < T: that target's own
static int set_loss_program(eh_handle* h, int target, const uint32_t* code, int32_t n_instr, const float* consts, int32_t n_const, int32_t out_slot, const char* who) {
    if (!h || !code || (n_const > 0 && !consts)) return EH_EINVAL;
    if (n_instr < 1 || n_instr > EH_MAX_PROG) return fail(h, EH_EUNSUPPORTED, "%s: %d instructions (1..%d)", who, n_instr, EH_MAX_PROG);
    if (n_const < 0 || n_const > EH_MAX_PROG_CONST) return fail(h, EH_EUNSUPPORTED, "%s: %d constants (0..%d)", who, n_const, EH_MAX_PROG_CONST);
    auto slot_ok = [&](unsigned sl, int upto) {
        if (sl < EH_PROG_SLOT_CONST) return sl < 2u;                                   // yhat, y
        if (sl < EH_PROG_SLOT_INSTR) return (int)sl - EH_PROG_SLOT_CONST < n_const;
        return (int)sl - EH_PROG_SLOT_INSTR < upto;
    };
    for (int i = 0; i < n_instr; ++i) {
        const unsigned w = code[i], op = w & 255u;
        if (op >= EH_OP_COUNT) return fail(h, EH_EUNSUPPORTED, "%s: instruction %d has unknown opcode %u", who, i, op);
        const int nop = (op == EH_OP_SELECT) ? 3 : (op == EH_OP_NEG || op == EH_OP_EXP || op == EH_OP_LOG || op == EH_OP_SQRT || op == EH_OP_TANH ||
                                                    op == EH_OP_SIGMOID || op == EH_OP_ABS || op == EH_OP_SIN || op == EH_OP_COS) ? 1 : 2;
        const unsigned sl[3] = {(w >> 8) & 255u, (w >> 16) & 255u, w >> 24};
        for (int k = 0; k < 3; ++k)
            if (k < nop ? !slot_ok(sl[k], i) : sl[k] != 0u) return fail(h, EH_EINVAL, "%s: instruction %d, operand %d names slot %u (undefined at that point, or a non-zero unused operand)", who, i, k, sl[k]);
    }
    if (out_slot < 0 || !slot_ok((unsigned)out_slot, n_instr)) return fail(h, EH_EINVAL, "%s: output slot %d", who, out_slot);
    if (h->net.loss == EH_LOSS_PROGRAM) { HIPCHK(h, hipSetDevice(h->device)); FLUSH(h); }
    EhLossProg1& lp = target < 0 ? static_cast<EhLossProg1&>(h->loss_prog) : h->loss_prog.per[target];
    lp.code.assign(code, code + n_instr);
    lp.consts.assign(consts, consts + n_const);
    lp.out = out_slot;
    h->loss_prog.gen++;
    return EH_OK;
}
int32_t eh_set_loss_program(eh_handle* h, const uint32_t* code, int32_t n_instr, const float* consts, int32_t n_const, int32_t out_slot) {
    if (h) for (int t = 0; t < EH_MAX_TARG; ++t) h->loss_prog.per[t] = EhLossProg1();        // one function for all targets again
    return set_loss_program(h, -1, code, n_instr, consts, n_const, out_slot, "eh_set_loss_program");
}
int32_t eh_set_target_loss_program(eh_handle* h, int32_t target, const uint32_t* code, int32_t n_instr, const float* consts, int32_t n_const, int32_t out_slot) {
    if (!h) return EH_EINVAL;
    if (target < 0 || target >= h->net.T) return fail(h, EH_EINVAL, "eh_set_target_loss_program: target %d of %d", target, h->net.T);
    return set_loss_program(h, target, code, n_instr, consts, n_const, out_slot, "eh_set_target_loss_program");
}

int32_t eh_set_target_losses(eh_handle* h, const int32_t* kinds, int32_t n) {
    if (!h || !kinds) return EH_EINVAL;
    if (n != h->net.T) return fail(h, EH_EINVAL, "eh_set_target_losses: %d losses for %d targets", n, h->net.T);
    unsigned lt = 0;
    bool same = true;
    bool any_prog = false, any_two = false;
    for (int t = 0; t < n; ++t) {
        if (kinds[t] < EH_LOSS_MSE || kinds[t] > EH_LOSS_PROGRAM) return fail(h, EH_EUNSUPPORTED, "eh_set_target_losses: loss %d for target %d is not implemented on the device", kinds[t], t);
        lt |= (unsigned)kinds[t] << (4 * t);
        same = same && kinds[t] == kinds[0];
        any_prog = any_prog || kinds[t] == EH_LOSS_PROGRAM;
        if (kinds[t] == EH_LOSS_PROGRAM && !h->loss_prog.has(t)) return fail(h, EH_ESTATE, "eh_set_target_losses: target %d: EH_LOSS_PROGRAM without a program (eh_set_loss_program / eh_set_target_loss_program first)", t);
        any_two = any_two || (kinds[t] >= EH_LOSS_PEARSONLOSS && kinds[t] <= EH_LOSS_PBKGELOSS) || (kinds[t] == EH_LOSS_RMSE && n > 1);
    }
    if (same) return eh_set_option(h, "training_loss", kinds[0]);
    if (any_two && h->fused) return fail(h, EH_EUNSUPPORTED, "rmse (on a multi-target model) / pearson / kge training losses take forward passes ahead of the step: switch fused_update off first");
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    // (what the single-kind code paths read -- the run-time compiler's "is there a recorded loss", the deferred normalisation of
    //  single-target steps; the kernels take every target's kind from loss_t)
    h->net.loss = any_prog ? EH_LOSS_PROGRAM : EH_LOSS_MSE;
    h->net.loss_t = lt;
    {   // two-pass losses exist in the generic kernels only (the K == 1 / P <= 4 fast paths stay untouched by them)
        const int want = h->lform ? 0 : fast_wanted(h->arch, h->net.K, h->net.P, h->net.T, h->net.mech);
        const int fast = any_two ? 0 : (want & h->fast_user);
        if (!h->lform && fast != h->fast) { h->fast = fast; return build_maps(h, false); }
    }
    return EH_OK;
}

// extra_loss as a function of the predictions (src/losses/compute_loss.jl:31-34; the reference's own test:
// `extra_loss = (yhat, ps) -> [sum(abs, yhat.var1), sum(abs, yhat.var2)]`, test/test_compute_loss.jl:257-285).  An entry that is the sum
// or the mean over ALL samples of the batch of a per-sample function f(yhat_o) of one model output is carried as one more TARGET: it
// observes output o (eh_model_desc::target_output), its data column holds no NaN (the host binding passes zeros), its per-sample loss is the
// recorded f (eh_set_target_loss_program, kind EH_LOSS_PROGRAM in eh_set_target_losses) -- and this call says so:
// roles[t] = 0 a data target | 1 an extra-loss entry, mean over all samples | 2 an extra-loss entry, sum over all samples.
// Such a target takes the extra loss's factor under agg = mean, no 1 / n when it is a sum, and does not count as a data target.
int32_t eh_set_target_roles(eh_handle* h, const int32_t* roles, int32_t n) {
    if (!h || !roles) return EH_EINVAL;
    if (n != h->net.T) return fail(h, EH_EINVAL, "eh_set_target_roles: %d roles for %d targets", n, h->net.T);
    unsigned r = 0;
    int ndata = 0;
    for (int t = 0; t < n; ++t) {
        if (roles[t] < 0 || roles[t] > 2) return fail(h, EH_EINVAL, "eh_set_target_roles: role %d of target %d (0 data, 1 extra-loss mean, 2 extra-loss sum)", roles[t], t);
        r |= (unsigned)roles[t] << (2 * t);
        ndata += roles[t] == 0;
    }
    if (ndata == 0) return fail(h, EH_EINVAL, "eh_set_target_roles: at least one data target");
    if (r != 0 && h->net.T < 2) return fail(h, EH_EINVAL, "eh_set_target_roles: an extra-loss entry is a target of its own");
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    h->roles = r;
    eh_agg_factors(h);
    return EH_OK;
}

int32_t eh_jit_status(eh_handle* h, int32_t* n_compiled, char* log, int64_t log_bytes) {
    if (!h || !n_compiled) return EH_EINVAL;
    int n = 0;
    for (auto& e : h->jit) n += e->state.load(std::memory_order_acquire) > 0 ? 1 : 0;
    // a canonical descriptor runs the kernel specialised AHEAD of time (eh_spec.hip): counted like a compiled pair, the log says which it is
    std::string msg = h->jit_log;
    if (h->spec_used && h->aot_spec) { n += 1; msg = std::string("ahead-of-time: ") + h->spec_used->what + (msg.empty() ? "" : "; ") + msg; }
    *n_compiled = n;
    if (log && log_bytes > 0) {
        const size_t m = std::min((size_t)log_bytes - 1, msg.size());
        memcpy(log, msg.data(), m);
        log[m] = 0;
    }
    return EH_OK;
}

// may this handle train on the sample-owned bf16 kernel (eh_bf16_sample.hpp)?  One network (its slab row is the plain canonical order); the
// A/B switches of the measurement tools are read once and mean the same to the "precision" and the "variant" option (advisor r05)
static bool so_allowed(const eh_handle* h) {
    static const bool no_so = getenv("EH_NO_DIRECT_STORE") != nullptr || getenv("EH_NO_SAMPLE_OWNED") != nullptr;
    return !no_so && h->n_nets == 1 && h->desc.n_nets == 0;
}

int32_t eh_set_option(eh_handle* h, const char* name, int64_t value) {
    if (!h || !name) return EH_EINVAL;
    if (h->lform && (!strcmp(name, "fast_paths") || !strcmp(name, "row_split") || !strcmp(name, "variant") || !strcmp(name, "precision"))) {
        if (!strcmp(name, "precision") && value) return fail(h, EH_EUNSUPPORTED, "precision: the layer-wise form computes in fp32");
        return EH_OK;                        // tile / kernel-family knobs of the fused kernels: nothing to choose in the layer-wise form
    }
    if (!strcmp(name, "max_blocks")) {
        if (value < 1 || value > 256) return fail(h, EH_EINVAL, "max_blocks must be 1..256 (one workgroup per CU)");
        h->max_blocks = (int)value;
        return EH_OK;
    }
    if (!strcmp(name, "bn_in_kernel")) {     // 0: always launch eh_bn_stats_kernel in front of a step with input BatchNorm (A/B, tests); 1 (default): small minibatches take their statistics inside the step kernel
        h->bn_no_self = value == 0;
        return EH_OK;
    }
    if (!strcmp(name, "multi_step")) {       // 1 (default): eh_train_epoch runs the steps of small minibatches (one workgroup each) several per launch; 0: one launch per step
        h->multi_step = value != 0;
        return EH_OK;
    }
    if (!strcmp(name, "eval_blocks")) {      // workgroups of the evaluation passes (eh_eval / eh_forward); 0 = the default of the kernel family
        if (value < 0 || value > 4096) return fail(h, EH_EINVAL, "eval_blocks must be 0 (default) .. 4096");
        h->eval_blocks = (int)value;
        return EH_OK;
    }
    if (!strcmp(name, "mech_blocks")) {      // workgroups of the stand-alone mechanistic stage (eh_mech_loss_vjp): rows of partials the finish kernel folds
        if (value < 0 || value > EH_MECH_MAXROWS) return fail(h, EH_EINVAL, "mech_blocks must be 0 (no cap) .. %d", (int)EH_MECH_MAXROWS);
        h->mech_blocks = (int)value;
        return EH_OK;
    }
    if (!strcmp(name, "mech_tiles")) {       // consecutive tiles (1 024 samples each) per workgroup of the stand-alone mechanistic stage
        if (value < 1 || value > 1024) return fail(h, EH_EINVAL, "mech_tiles must be 1..1024");
        h->mech_tiles = (int)value;
        return EH_OK;
    }
    if (!strcmp(name, "fast_paths")) {       // 0 forces the generic MFMA kernels (A/B testing)
        const int want = fast_wanted(h->arch, h->net.K, h->net.P, h->net.T, h->net.mech);
        h->fast_user = value ? (int)value : 0;
        h->fast = (h->net.loss >= EH_LOSS_PEARSONLOSS) ? 0 : (want & h->fast_user);
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);
        return build_maps(h, false);
    }
    if (!strcmp(name, "fused_update")) {     // 1: one kernel per step (float-atomic accumulation, not bitwise reproducible)
        if (value && two_pass_mask(h->net)) return fail(h, EH_EUNSUPPORTED, "fused_update: rmse (on a multi-target model) / pearson / kge training losses take forward passes ahead of the step");
        if (value && (h->img.l2c != 0.0f || h->img.l2w)) return fail(h, EH_EUNSUPPORTED, "fused_update: the weight_l2 extra loss is not built for it");
        if (value && h->arch->wide) return fail(h, EH_EUNSUPPORTED, "fused_update is not built for hidden widths above 64");
        if (!value && h->p2p_alloc) return fail(h, EH_ESTATE, "fused_update: eh_p2p_disable first");
        if (value < 0 || value > 2) return fail(h, EH_EINVAL, "fused_update must be 0, 1 or 2");
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);
        h->fused = value != 0;
        // 2 = "where it is reproducible": one kernel per step only for minibatches ONE workgroup covers -- its sums meet in one fixed order
        // (several steps per launch with the state in LDS, or one add per accumulator) -- and the deterministic step + reduce pair for
        // every larger minibatch, whose workgroups' float atomics land in no fixed order.  What train() asks for when random_seed is set
        // (the reference's default, src/config/TrainingConfig.jl:85-86: a seeded CPU run IS reproducible).
        h->fused_det = value == 2;
        return EH_OK;
    }
    if (!strcmp(name, "training_loss")) {
        if (value < EH_LOSS_MSE || value > EH_LOSS_PROGRAM) return fail(h, EH_EUNSUPPORTED, "training_loss %lld is not implemented on the device", (long long)value);
        if (value == EH_LOSS_PROGRAM)
            for (int t = 0; t < h->net.T; ++t)
                if (!h->loss_prog.has(t)) return fail(h, EH_ESTATE, "training_loss EH_LOSS_PROGRAM: call eh_set_loss_program first");
        const bool two = (value >= EH_LOSS_PEARSONLOSS && value <= EH_LOSS_PBKGELOSS) || (value == EH_LOSS_RMSE && h->net.T > 1);
        if (two && h->fused) return fail(h, EH_EUNSUPPORTED, "rmse (on a multi-target model) / pearson / kge training losses take forward passes ahead of the step: switch fused_update off first");
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);
        h->net.loss = (int)value;
        h->net.loss_t = 0;
        for (int t = 0; t < h->net.T; ++t) h->net.loss_t |= (unsigned)value << (4 * t);
        if (!h->lform) {   // the two-pass losses exist in the generic kernels only (the K == 1 / P <= 4 fast paths stay untouched by them)
            const int want = fast_wanted(h->arch, h->net.K, h->net.P, h->net.T, h->net.mech);
            const int fast = (two || value >= EH_LOSS_PEARSONLOSS) ? 0 : (want & h->fast_user);
            if (fast != h->fast) { h->fast = fast; return build_maps(h, false); }
        }
        return EH_OK;
    }
    if (!strcmp(name, "jit")) {              // recorded closures: 1 = kernels compiled at run time around the program (default), 0 = the interpreter
        h->jit_on = value != 0;
        return EH_OK;
    }
    if (!strcmp(name, "agg") || !strcmp(name, "extra_terms")) {
        // `agg::Function` of the training configuration (src/config/TrainingConfig.jl:76-77): the training loss is
        // agg([agg(per-target losses), extra loss entries...]) (src/losses/compute_loss.jl:31-34,50-53).  0 = sum (default), 1 = mean;
        // "extra_terms" = the number of entries the extra loss returns (what eh_set_weight_l2 / eh_set_weight_l2_coef stand for: one
        // entry, or as many weight_l2 terms as the host folded into the coefficients) -- only `mean` needs it.
        if (!strcmp(name, "agg")) { if (value != 0 && value != 1) return fail(h, EH_EINVAL, "agg must be 0 (sum) or 1 (mean)"); }
        else if (value < 0 || value > 64) return fail(h, EH_EINVAL, "extra_terms must be 0..64");
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);                                 // (a pending fused update was computed under the old setting)
        if (!strcmp(name, "agg")) h->agg = (int)value; else h->n_extra = (int)value;
        eh_agg_factors(h);
        return EH_OK;
    }
    if (!strcmp(name, "check_idx")) {        // debug: range-check minibatch indices that live on the DEVICE (eh_train_step, idx_on_device != 0) before every step --
        h->check_idx = value != 0;           // a small kernel + one synchronisation per step; host indices are always checked
        return EH_OK;
    }
    if (!strcmp(name, "p2p_mode")) {         // who publishes a step's sums to the peers (EhP2P::mode, eh_device.hpp): 0 = the step's last workgroup to finish, elected by a
                                             // two-level ticket; 1 = workgroup 0 of the next kernel on the stream, no election.  EVERY rank switches at the same step.
        if (value != 0 && value != 1) return fail(h, EH_EINVAL, "p2p_mode must be 0 (election in the step's epilogue) or 1 (publish from the next kernel's prologue)");
        if (!h->p2p_on) return fail(h, EH_ESTATE, "p2p_mode: no peer-to-peer exchange on this handle (eh_p2p_init / eh_p2p_init_local first)");
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);                                // (no step pending across the switch: whoever was to publish it has done so)
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->p2p_host.mode = (int)value;
        HIPCHK(h, hipMemcpy(h->p2p_dev, &h->p2p_host, sizeof(EhP2P), hipMemcpyHostToDevice));
        return EH_OK;
    }
    if (!strcmp(name, "empty_target_nan")) { // a target with no valid sample inside a batch that has some: the gradient is the reference's either way (that target adds
        h->empty_nan = value != 0;           // nothing); the VALUE is the sum of the other targets (0, default) or the reference's NaN = mean over an empty selection
        return EH_OK;                        // (1; src/losses/loss_fn.jl:61-63), reported by eh_loss_and_grad -- the seam where the reference's objective is called directly
    }
    if (!strcmp(name, "aot_spec")) {         // 0 = never the kernels specialised ahead of time for the canonical descriptors (eh_spec.hip): tests of the other paths, A/B
        h->aot_spec = value != 0;
        return EH_OK;
    }
    if (!strcmp(name, "specialize")) {       // 1 = step kernels compiled at run time with the model descriptor as a compile-time constant;
        h->specialize = value != 0;          // 2 = the same in a background thread: steps run the kernels built ahead of time until the
        h->specialize_async = value == 2;    //     compiled one is ready (~1 s, or a disk-cache hit), then switch -- same arithmetic, same results
        return EH_OK;
    }
    if (!strcmp(name, "row_split")) {        // A/B: the row-split kernel family (eh_wide.hpp) where both are built
        const bool now = h->arch->wide != 0;
        if ((value != 0) == now) return EH_OK;
        if (!h->arch_alt) return fail(h, EH_EUNSUPPORTED, "row_split: this model has only the %s kernel", now ? "row-split" : "per-wave");
        if (value && h->fused) return fail(h, EH_EUNSUPPORTED, "row_split: switch fused_update off first");
        if (h->arch->var[h->variant].bf16) return fail(h, EH_ESTATE, "row_split: the bf16-forward kernels exist in the row-split family only (set precision 0 first)");
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);
        std::swap(h->arch, h->arch_alt);
        h->variant = (h->arch->nvar > 1 && !h->arch->wide) ? 1 : 0;
        h->fast = (h->net.loss >= EH_LOSS_PEARSONLOSS) ? 0 : (fast_wanted(h->arch, h->net.K, h->net.P, h->net.T, h->net.mech) & h->fast_user);
        for (int vi = 0; vi < h->arch->nvar; ++vi) HIPCHK(h, h->arch->var[vi].prepare());
        return build_maps(h, false);
    }
    if (!strcmp(name, "precision")) {        // 0 = fp32 end to end (the reference's arithmetic); eh_wide_bf16.hpp: 1 = bf16 forward products, fp32 accumulate, fp32-exact
                                             // backward; 2 = bf16 operands in both passes (every backward delta rounded once), fp32 accumulate
        if (value < 0 || value > 2) return fail(h, EH_EINVAL, "precision must be 0 (f32), 1 (bf16 forward / fp32-exact backward) or 2 (bf16 operands in both passes, fp32 accumulate)");
        const int now = h->arch->var[h->variant].bf16;
        if ((int)value == now) return EH_OK;
        if (value) {
            if (h->act == EH_ACT_SWISH || h->act == EH_ACT_PER_NET)
                return fail(h, EH_EUNSUPPORTED, "precision: the bf16 kernels keep only the rounded activation (tanh / sigmoid / relu / identity; not swish, not per-net activations)");
            if (h->fused) return fail(h, EH_EUNSUPPORTED, "precision: switch fused_update off first (the row-split kernels have no such mode)");
            const EhArchInfo* W = h->arch->wide ? h->arch : h->arch_alt;
            // one network (the slab row in plain canonical order): the sample-owned training kernel (eh_bf16_sample.hpp); else the row-split one
            const bool so_ok = so_allowed(h);
            int vb = -1;
            if (W && so_ok) for (int vi = 0; vi < W->nvar; ++vi) if (W->var[vi].bf16 == (int)value && W->var[vi].so) { vb = vi; break; }
            if (W && vb < 0) for (int vi = 0; vi < W->nvar; ++vi) if (W->var[vi].bf16 == (int)value && !W->var[vi].so) { vb = vi; break; }
            if (vb < 0) return fail(h, EH_EUNSUPPORTED, "precision: no bf16 kernel is built for this shape (row-split shapes only: hidden width 33..128)");
            HIPCHK(h, hipSetDevice(h->device));
            FLUSH(h);
            if (W != h->arch) { std::swap(h->arch, h->arch_alt); h->fast = 0; }
            h->variant = vb;
            for (int vi = 0; vi < h->arch->nvar; ++vi) HIPCHK(h, h->arch->var[vi].prepare());
        } else {
            HIPCHK(h, hipSetDevice(h->device));
            FLUSH(h);
            h->variant = 0;
        }
        return build_maps(h, false);
    }
    if (!strcmp(name, "variant")) {
        if (value < 0 || value >= h->arch->nvar) return fail(h, EH_EINVAL, "variant must be 0..%d for this shape", h->arch->nvar - 1);
        if (h->arch->var[value].bf16 != h->arch->var[h->variant].bf16) return fail(h, EH_EINVAL, "variant %lld belongs to the other precision (set the \"precision\" option)", (long long)value);
        if (h->arch->var[value].so && !so_allowed(h))
            return fail(h, EH_EUNSUPPORTED, "variant %lld is the sample-owned training kernel: one-network models only (and not with EH_NO_DIRECT_STORE / EH_NO_SAMPLE_OWNED set)", (long long)value);
        h->variant = (int)value;
        HIPCHK(h, hipSetDevice(h->device));      // the reduction map depends on the variant (waves of a row-split workgroup; layout of the parked accumulators)
        return build_maps(h, false);
    }
    return fail(h, EH_EINVAL, "unknown option %s", name);
}

int32_t eh_set_data(eh_handle* h, int32_t split, int64_t n, const float* x, const float* const* forcings, const float* const* targets,
                    int32_t on_device) {
    if (!h) return EH_EINVAL;
    if (split != EH_SPLIT_TRAIN && split != EH_SPLIT_VAL) return fail(h, EH_EINVAL, "eh_set_data: split %d", split);
    if (n < 0 || n > 0x7fffffffLL) return fail(h, EH_EINVAL, "eh_set_data: n = %lld", (long long)n);
    if (on_device & ~(EH_DATA_ON_DEVICE | EH_DATA_X_PLANES | EH_DATA_X_ROWS)) return fail(h, EH_EINVAL, "eh_set_data: flags %d", on_device);
    if ((on_device & EH_DATA_X_ROWS) && (on_device & (EH_DATA_ON_DEVICE | EH_DATA_X_PLANES))) return fail(h, EH_EINVAL, "eh_set_data: EH_DATA_X_ROWS goes with host arrays only, and not with EH_DATA_X_PLANES");
    if (n > 0 && ((!x && h->net.P > 0) || !forcings || !targets)) return fail(h, EH_EINVAL, "eh_set_data: null array");
    const EhNet& net = h->net;
    HIPCHK(h, hipSetDevice(h->device));
    EhSplit& sp = h->split[split];
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(sp.recs);
    sp.recs = nullptr; sp.n = 0;
    if (split == EH_SPLIT_TRAIN) h->perm_valid = false;
    if (n == 0) return EH_OK;
    const int C = h->C;
    static const bool dbg_t = getenv("EH_DEBUG_SET_DATA") != nullptr;      // (diagnostic: where an upload's wall clock goes)
    const auto t_a = std::chrono::steady_clock::now();
    HIPCHK(h, hipMalloc(&sp.recs, (size_t)n * C * sizeof(float)));
    const auto t_b = std::chrono::steady_clock::now();
    const bool planes = (on_device & EH_DATA_X_PLANES) != 0;       // x as P arrays of N (row-major P x N: what a NumPy host holds) instead of N records of P
    const bool xrows = (on_device & EH_DATA_X_ROWS) != 0;          // x as P POINTERS to arrays of N: the caller's own columns, never stacked into a matrix
    const float* const* const xr = reinterpret_cast<const float* const*>(x);
    if (xrows) for (int j = 0; j < net.P; ++j) if (!xr[j]) return fail(h, EH_EINVAL, "eh_set_data: predictor row %d is null", j);
    on_device &= EH_DATA_ON_DEVICE;
    if (on_device) {
        EhPackArgs pa{};
        pa.x = x;
        for (int f = 0; f < net.F; ++f) pa.forc[f] = forcings[f];
        for (int t = 0; t < net.T; ++t) pa.targ[t] = targets[t];
        const long long tot = (long long)n * C;
        hipLaunchKernelGGL(eh_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, pa, sp.recs, (long long)n, net.P, net.F, net.T, planes ? 1 : 0);
        HIPCHK(h, hipGetLastError());
        // metric shift: mean of the first valid targets, computed from a small host copy
        std::vector<float> tmp((size_t)std::min<int64_t>(n, 4096));
        for (int t = 0; t < net.T; ++t) {
            HIPCHK(h, hipMemcpy(tmp.data(), targets[t], tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
            double s = 0; long long c = 0;
            for (float vv : tmp) if (!std::isnan(vv)) { s += vv; ++c; }
            sp.shift[t] = c ? (float)(s / c) : 0.0f;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
        // records interleaved on the host in chunks of <= 1 M samples, through two pinned staging buffers: chunk k is packed (two
        // to eight threads, each on its own run of samples) while chunk k - 1 is on its way over PCIe.  (Rounds 1-3: one pageable 67 MB vector, one thread, one
        // blocking copy: 56 ms for the headline data set; the one-time cost a user's train() call pays before its first step.)
        for (int t = 0; t < net.T; ++t) {
            double sum = 0; long long c = 0;
            for (int64_t s = 0; s < std::min<int64_t>(n, 4096); ++s) if (!std::isnan(targets[t][s])) { sum += targets[t][s]; ++c; }
            sp.shift[t] = c ? (float)(sum / c) : 0.0f;
        }
        const int64_t CH = std::min<int64_t>(n, (int64_t)1 << 20);
        // (the two pinned buffers are kept for the next call of the process -- pinning 2 x 16 MB costs as much as the upload they stage;
        //  a second thread uploading at the same time gets buffers of its own)
        float* stage[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        const size_t stage_bytes = (size_t)CH * C * sizeof(float);
        std::unique_lock<std::mutex> pool_lk(g_stage_mu, std::try_to_lock);
        const bool pooled = pool_lk.owns_lock();
        bool pinned;
        if (pooled) {
            if (g_stage_bytes < stage_bytes) {
                for (int k = 0; k < 2; ++k) { if (g_stage[k]) (void)hipHostFree(g_stage[k]); g_stage[k] = nullptr; }
                g_stage_bytes = 0;
                if (hipHostMalloc((void**)&g_stage[0], stage_bytes, hipHostMallocPortable) == hipSuccess &&          // (portable: the next handle may sit on another device)
                    hipHostMalloc((void**)&g_stage[1], stage_bytes, hipHostMallocPortable) == hipSuccess) g_stage_bytes = stage_bytes;
                else { (void)hipGetLastError(); for (int k = 0; k < 2; ++k) { if (g_stage[k]) (void)hipHostFree(g_stage[k]); g_stage[k] = nullptr; } }
            }
            pinned = g_stage_bytes >= stage_bytes;
            if (pinned) { stage[0] = g_stage[0]; stage[1] = g_stage[1]; }
        } else {
            pinned = hipHostMalloc((void**)&stage[0], stage_bytes, hipHostMallocDefault) == hipSuccess &&
                     hipHostMalloc((void**)&stage[1], stage_bytes, hipHostMallocDefault) == hipSuccess;
        }
        std::vector<float> pageable;
        if (!pinned) {                                   // (no pinned memory to be had: the plain path)
            (void)hipGetLastError();
            if (!pooled) { if (stage[0]) (void)hipHostFree(stage[0]); if (stage[1]) (void)hipHostFree(stage[1]); }
            pageable.resize((size_t)CH * C);
            stage[0] = stage[1] = pageable.data();
        } else {
            // (an event that cannot be made: the first one is not leaked and a one-off staging pair not kept; advisor, round 4)
            hipError_t ee = hipEventCreateWithFlags(&done[0], hipEventDisableTiming);
            if (ee == hipSuccess) { ee = hipEventCreateWithFlags(&done[1], hipEventDisableTiming); if (ee != hipSuccess) (void)hipEventDestroy(done[0]); }
            if (ee != hipSuccess) { if (!pooled) { (void)hipHostFree(stage[0]); (void)hipHostFree(stage[1]); } HIPCHK(h, ee); }
        }
        const auto t_c = std::chrono::steady_clock::now();
        const int P = net.P, F = net.F, T = net.T;
        auto pack = [&](float* dst, int64_t s0, int64_t s1, int64_t base) {
            for (int64_t s = s0; s < s1; ++s) {
                float* r = dst + (size_t)(s - base) * C;
                if (xrows) for (int j = 0; j < P; ++j) r[j] = xr[j][s];
                else if (planes) for (int j = 0; j < P; ++j) r[j] = x[(size_t)j * (size_t)n + (size_t)s];
                else for (int j = 0; j < P; ++j) r[j] = x[(size_t)s * P + j];
                for (int f = 0; f < F; ++f) r[P + f] = forcings[f][s];
                for (int t = 0; t < T; ++t) r[P + F + t] = targets[t][s];
            }
        };
        hipError_t err = hipSuccess;
        int k = 0;
        for (int64_t s0 = 0; s0 < n && err == hipSuccess; s0 += CH, ++k) {
            const int64_t cnt = std::min(CH, n - s0);
            float* const buf = stage[k & 1];
            const auto tq0 = std::chrono::steady_clock::now();
            if (pinned && k >= 2) err = hipEventSynchronize(done[k & 1]);       // the copy that last read this buffer is through
            if (err != hipSuccess) break;
            const auto tq1 = std::chrono::steady_clock::now();
            if (cnt >= 65536) {          // a chunk is interleaved by up to eight threads, each on its own run of samples
                const unsigned hw = std::thread::hardware_concurrency();
                static const int pack_max = getenv("EH_PACK_THREADS") ? std::max(1, atoi(getenv("EH_PACK_THREADS"))) : 8;
                const int nthr = (int)std::max<int64_t>(2, std::min<int64_t>({(int64_t)pack_max, (int64_t)(hw ? hw / 2 : 2), cnt / 32768}));
                const int64_t per = (cnt + nthr - 1) / nthr;
                std::vector<std::thread> others;
                int64_t mine_end = std::min(s0 + cnt, s0 + per);
                try {                    // (no exception crosses the C ABI: a thread that cannot be started leaves its run of samples to this one)
                    for (int w = 1; w < nthr; ++w) {
                        const int64_t a0 = s0 + w * per, a1 = std::min(s0 + cnt, a0 + per);
                        if (a0 < a1) others.emplace_back([&, a0, a1] { pack(buf, a0, a1, s0); });
                    }
                } catch (...) {
                    for (auto& t : others) t.join();
                    others.clear();
                    mine_end = s0 + cnt;         // single-threaded: everything (the threads that did start packed the same values into the same places)
                }
                pack(buf, s0, mine_end, s0);
                for (auto& t : others) t.join();
            } else pack(buf, s0, s0 + cnt, s0);
            const auto tq2 = std::chrono::steady_clock::now();
            if (pinned) {
                err = hipMemcpyAsync(sp.recs + (size_t)s0 * C, buf, (size_t)cnt * C * sizeof(float), hipMemcpyHostToDevice, h->stream);
                if (err == hipSuccess) err = hipEventRecord(done[k & 1], h->stream);
                if (dbg_t) {
                    auto ms = [](std::chrono::steady_clock::time_point u, std::chrono::steady_clock::time_point v) { return std::chrono::duration<double, std::milli>(v - u).count(); };
                    fprintf(stderr, "   chunk %d: wait %.2f pack %.2f enqueue %.2f ms\n", k, ms(tq0, tq1), ms(tq1, tq2), ms(tq2, std::chrono::steady_clock::now()));
                }
            } else err = hipMemcpy(sp.recs + (size_t)s0 * C, buf, (size_t)cnt * C * sizeof(float), hipMemcpyHostToDevice);
        }
        // always drained before the staging pair is released or handed to the next caller -- also after an error: copies of earlier chunks
        // may still be reading it (advisor, round 4)
        const auto t_d = std::chrono::steady_clock::now();
        { const hipError_t es = hipStreamSynchronize(h->stream); if (err == hipSuccess) err = es; }
        if (dbg_t) {
            auto ms = [](std::chrono::steady_clock::time_point u, std::chrono::steady_clock::time_point v) { return std::chrono::duration<double, std::milli>(v - u).count(); };
            fprintf(stderr, "eh_set_data n=%lld: hipMalloc %.2f ms, staging (pooled %d pinned %d) %.2f ms, pack + enqueue %.2f ms, drain %.2f ms\n", (long long)n, ms(t_a, t_b), (int)pooled, (int)pinned,
                    ms(t_b, t_c), ms(t_c, t_d), ms(t_d, std::chrono::steady_clock::now()));
        }
        if (pinned) { (void)hipEventDestroy(done[0]); (void)hipEventDestroy(done[1]); if (!pooled) { (void)hipHostFree(stage[0]); (void)hipHostFree(stage[1]); } }
        HIPCHK(h, err);
    }
    sp.n = n;
    return EH_OK;
}

int32_t eh_set_params(eh_handle* h, const float* theta, int64_t n) {
    if (!h || !theta) return EH_EINVAL;
    if (n != h->net.n_theta) return fail(h, EH_EINVAL, "eh_set_params: n = %lld, model has %d", (long long)n, h->net.n_theta);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (int rc = copy_in(h, TH(h), theta, (size_t)n)) return rc;
    hipLaunchKernelGGL(eh_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, TH(h), (int)n, h->img);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EH_OK;
}

int32_t eh_get_params(eh_handle* h, float* theta, int64_t n) {
    if (!h || !theta) return EH_EINVAL;
    if (n != h->net.n_theta) return fail(h, EH_EINVAL, "eh_get_params: n = %lld, model has %d", (long long)n, h->net.n_theta);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return copy_out(h, theta, TH(h), (size_t)n);
}

}   // extern "C"

// ---- internal launch helpers ---------------------------------------------------------------------
static int ensure_events(eh_handle* h, size_t need);
static int grid_for(const eh_handle* h, long long count) {
    const EhVariant& v = h->arch->var[h->variant];
    const long long mt = 16LL * v.nt;
    const long long ntiles = (count + mt - 1) / mt;
    return (int)std::max<long long>(1, std::min<long long>((ntiles + (v.tiles ? v.tiles : v.nw) - 1) / (v.tiles ? v.tiles : v.nw), h->max_blocks));
}

// The evaluation passes hold no gradient accumulators: the per-wave kernels need a quarter to a half of the registers of their training
// form (headline shape: 60 against 128 VGPRs), so several workgroups fit a CU, and a forward over a whole split is a long chain of
// dependent per-tile latencies that more resident waves hide.  The row-split kernels fill a CU's LDS with one workgroup: no gain, and every
// extra workgroup pays the image prologue again.  The metric rows (EH_EVAL_STATS * T floats per workgroup) fit the slab's 4 MB floor.
static int eval_grid_for(const eh_handle* h, long long count) {
    const EhVariant& v = h->arch->var[h->variant];
    const long long mt = 16LL * v.nt, per = v.tiles ? v.tiles : v.nw;
    const long long ntiles = (count + mt - 1) / mt;
    int cap = h->max_blocks;
    if (h->eval_blocks > 0) cap = h->eval_blocks;
    else if (!h->arch->wide && h->max_blocks == 256) cap = EH_EVAL_BLOCKS;
    return (int)std::max<long long>(1, std::min<long long>((ntiles + per - 1) / per, cap));
}

// input BatchNorm: statistics of the minibatch [first, first+count) -> a.bn_* (train-mode kernels only)
// consume = false: a forward-only pass in front of the step's training pass (eh_dp_moments) -- the global statistics stay for the passes behind it
static int bn_prepare(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long count, bool update, EhStepArgs* a, bool consume = true) {
    a->bn_part = nullptr; a->bn_nblk = 0; a->bn_update = 0; a->bn_run = h->bn_run; a->image_out = h->image;
    a->bn_c = nullptr; a->bn_n = nullptr;
    if (!h->bn_on) return EH_OK;
    if (h->bn_ext) {          // statistics of the GLOBAL batch, summed over the GPUs by the host since eh_dp_bn_stats
        a->bn_part = h->bn_stat; a->bn_nblk = 1; a->bn_c = h->bn_shift; a->bn_n = h->bn_stat + 64;
        a->bn_update = (update || h->bn_dp_update) ? 1 : 0;
        if (consume) { h->bn_ext = false; h->bn_dp_update = false; }
        return EH_OK;
    }
    if (count <= 0) return EH_OK;
    if (!h->arch->wide && !h->lform && count <= EH_BN_SELF_MAX && !h->bn_no_self) {      // small minibatch, per-wave kernel: the step kernel takes the statistics itself
        a->bn_nblk = -1; a->bn_update = update ? 1 : 0;
        return EH_OK;
    }
    const int nblk = (int)std::max<long long>(1, std::min<long long>(32, (count + 1023) / 1024));
    hipLaunchKernelGGL(eh_bn_stats_kernel, dim3(nblk), dim3(1024), 0, h->stream, sp.recs, h->C, h->net.P, idx, (int)first, (int)count, h->bn_part, nullptr);
    HIPCHK(h, hipGetLastError());
    a->bn_part = h->bn_part; a->bn_nblk = nblk; a->bn_c = h->bn_part + nblk * 64; a->bn_update = update ? 1 : 0;
    return EH_OK;
}

// ---- layer-wise execution form (eh_lform.hpp) ---------------------------------------------------------------------------
struct EhLWs { float *Xb, *H[EH_MAX_NETS][EH_MAX_HIDDEN], *Z[EH_MAX_NETS][EH_MAX_HIDDEN], *D[2], *O, *part; long long ldo; };
static long long lform_floats_per_sample(const eh_handle* h) {
    long long w = h->net.P + 16, maxw = 0;
    for (int k = 0; k < h->l_nnets; ++k)
        for (int l = 0; l + 1 < h->l_net[k].nl; ++l) {
            w += h->l_net[k].out[l] * (h->l_net[k].lact[l] == EH_ACT_SWISH ? 2 : 1);      // swish keeps the pre-activation too
            maxw = std::max<long long>(maxw, h->l_net[k].out[l]);
        }
    return w + 2 * maxw;
}
// an allocation would be needed while a graph is being recorded: drop the recording (the engine stays usable) and say what to do
static int lform_alloc_in_capture(eh_handle* h, const char* what) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(h->stream, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    h->capturing = false;
    return fail(h, EH_ESTATE, "%s: the layer-wise form sizes this buffer at the first step that needs it -- run one step of this batch size (and training loss) "
                              "before eh_graph_begin, then record; the recording was dropped", what);
}
static int lform_workspace(eh_handle* h, long long count, EhLWs* W) {
    const long long cap_need = (count + 127) / 128 * 128;
    if (cap_need > h->l_cap) {
        if (h->capturing) return lform_alloc_in_capture(h, "workspace");
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->l_ws);
        h->l_ws = nullptr; h->l_cap = 0;
        HIPCHK(h, hipMalloc(&h->l_ws, ((size_t)cap_need * lform_floats_per_sample(h) + (size_t)4096 * EH_EVAL_STATS * EH_MAX_TARG) * sizeof(float)));
        h->l_cap = cap_need;
    }
    const long long cap = h->l_cap;
    float* p = h->l_ws;
    long long maxw = 0;
    W->Xb = p; p += cap * h->net.P;
    for (int k = 0; k < h->l_nnets; ++k)
        for (int l = 0; l + 1 < h->l_net[k].nl; ++l) {
            const long long o = h->l_net[k].out[l];
            maxw = std::max(maxw, o);
            W->H[k][l] = p; p += cap * o;
            W->Z[k][l] = nullptr;
            if (h->l_net[k].lact[l] == EH_ACT_SWISH) { W->Z[k][l] = p; p += cap * o; }
        }
    W->D[0] = p; p += cap * maxw;
    W->D[1] = p; p += cap * maxw;
    W->O = p; p += cap * 16; W->ldo = cap;
    W->part = p;
    return EH_OK;
}
static const bool g_gemm_novec = getenv("EH_GEMM_NOVEC") != nullptr;      // (A/B switch of the measurement tools)
// a weight gradient (A transposed, plain store) with a thin side runs as a streaming kernel (eh_thin_gemm_kernel): its arguments
static bool lform_thin_args(const EhGemmArgs& g, bool btr, EhThinArgs* out) {
    if (g_gemm_novec || !(g.M <= 8 || g.N <= 8) || (g.M <= 8 ? btr : !btr)) return false;
    EhThinArgs t{};
    t.K = g.K; t.kchunk = g.kchunk; t.C = g.C; t.c_z = g.c_zstride; t.cs_z = g.c_zstride;
    if (g.M <= 8) {          // few inputs: wide = B (dZ [B x out]), thin = A (the layer's input [B x in])
        t.wide = g.B; t.ldw = g.ldb; t.ncols = g.N; t.thin = g.A; t.tsb = g.lda; t.tsj = 1; t.J = g.M; t.c_col = 1; t.c_j = g.ldc; t.cs_wide = g.colsum;
    } else {                 // few outputs, dO stored [K][ldo]: wide = A (the layer's input [B x in]), thin = B
        t.wide = g.A; t.ldw = g.lda; t.ncols = g.M; t.thin = g.B; t.tsb = 1; t.tsj = g.ldb; t.J = g.N; t.c_col = g.ldc; t.c_j = 1; t.cs_thin = g.colsum;
    }
    *out = t;
    return true;
}
template <bool ATR, bool BTR, int EPI>
static void lform_gemm(eh_handle* h, const EhGemmArgs& g, int nz) {
    const bool novec = g_gemm_novec;
    if constexpr (EPI == EH_GEPI_STORE && ATR) {
        EhThinArgs t;
        if (lform_thin_args(g, BTR, &t)) {
            hipLaunchKernelGGL(eh_thin_gemm_kernel, dim3((unsigned)((t.ncols + 63) / 64), (unsigned)nz), dim3(256), 0, h->stream, t);
            return;
        }
    }
    if (!novec && nz == 1 && g.kchunk >= g.K) {          // products with a degenerate dimension: streaming kernels (eh_lform.hpp)
        const long long tot = (long long)g.M * g.N;
        const unsigned sg = (unsigned)std::max<long long>(1, std::min<long long>(2048, (tot + 255) / 256));
        if (EPI == EH_GEPI_BIAS_ACT && !ATR && !BTR && g.K <= 8) { hipLaunchKernelGGL(eh_thin_fwd_k_kernel, dim3(sg), dim3(256), 0, h->stream, g); return; }
        if (EPI == EH_GEPI_BIAS_T && !ATR && !BTR && g.N <= 16 && g.M <= 16384) {
            hipLaunchKernelGGL(eh_thin_fwd_n_kernel, dim3((unsigned)std::max(1, std::min(1024, (g.M + 3) / 4))), dim3(256), 0, h->stream, g);
            return;
        }
        if (EPI == EH_GEPI_DACT && ATR && BTR && g.K <= 16) { hipLaunchKernelGGL(eh_thin_dact_kernel, dim3(sg), dim3(256), 0, h->stream, g); return; }
    }
    if constexpr (EPI == EH_GEPI_BIAS_ACT || EPI == EH_GEPI_DACT) {
        // few rows, deep k: split-K into partial products + a combine pass (eh_splitk_combine_kernel)
        static const bool nosplit = getenv("EH_GEMM_NOSPLIT") != nullptr;
        static const bool nofew = getenv("EH_GEMM_NOFEWROWS") != nullptr;
        static const int fewmax = getenv("EH_GEMM_FEWROWS_MAX") ? atoi(getenv("EH_GEMM_FEWROWS_MAX")) : 1024;
        if constexpr (!ATR) {
            // ... or, without the combine pass, one 16 x 16 tile per workgroup with the k range split over its waves (eh_fewrows_gemm_kernel)
            if (!novec && !nofew && nz == 1 && g.M <= fewmax && g.K >= 64 && g.kchunk >= g.K && eh_gemm_vec_ok(g, ATR, BTR)) {
                EhGemmArgs p = g;
                p.kchunk = std::max(16, ((g.K + 15) / 16 + 15) / 16 * 16);         // <= 16 slices of whole 16-deep groups
                const int nw = (g.K + p.kchunk - 1) / p.kchunk;
                static const int few32 = getenv("EH_GEMM_FEW32_MIN") ? atoi(getenv("EH_GEMM_FEW32_MIN")) : 512;       // rows from which the 32 x 32 tile runs
                if (g.M >= few32) hipLaunchKernelGGL((eh_fewrows32_gemm_kernel<BTR, EPI>), dim3((unsigned)((g.N + 31) / 32), (unsigned)((g.M + 31) / 32)), dim3(64u * (unsigned)nw), 0, h->stream, p);
                else {
                    hipLaunchKernelGGL((eh_fewrows_gemm_kernel<BTR, EPI>), dim3((unsigned)((g.N + 15) / 16), (unsigned)((g.M + 15) / 16)), dim3(64u * (unsigned)nw), 0, h->stream, p);
                    if (p.job_part) h->l_job_done = true;      // (the one product kernel that takes the side job, EhGemmArgs::job_part)
                }
                return;
            }
        }
        if (!novec && !nosplit && nz == 1 && g.M <= 256 && g.K >= 128 && g.kchunk >= g.K && eh_gemm_vec_ok(g, ATR, BTR)) {
            int kc = std::max(32, ((g.K + 15) / 16 + 15) / 16 * 16);          // ~16 parts, whole 16-deep steps
            const int ns = (g.K + kc - 1) / kc;
            const size_t need = (size_t)ns * g.M * g.N;
            if (ns > 1) {
                if (need > h->l_split_cap && !h->capturing) {      // (while a graph is being recorded: no allocation -- the tiled kernel below runs instead)
                    if (hipStreamSynchronize(h->stream) == hipSuccess) { (void)hipFree(h->l_split); h->l_split = nullptr; h->l_split_cap = 0; }
                    if (hipMalloc(&h->l_split, need * sizeof(float)) == hipSuccess) h->l_split_cap = need; else (void)hipGetLastError();
                }
                if (need <= h->l_split_cap) {
                    EhGemmArgs p = g;
                    p.C = h->l_split; p.ldc = g.N; p.c_zstride = (long long)g.M * g.N; p.kchunk = kc; p.colsum = nullptr; p.bias = nullptr; p.H = nullptr; p.Z = nullptr;
                    const dim3 g64((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)ns);
                    hipLaunchKernelGGL((eh_gemm_kernel<ATR, BTR, EH_GEPI_STORE, true, 64>), g64, dim3(256), 0, h->stream, p);
                    const long long tot = (long long)g.M * g.N;
                    hipLaunchKernelGGL(eh_splitk_combine_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(1024, (tot + 255) / 256))), dim3(256), 0, h->stream,
                                       h->l_split, ns, (int)EPI, g);
                    return;
                }
            }
        }
    }
    const dim3 grid((unsigned)((g.N + 127) / 128), (unsigned)((g.M + 127) / 128), (unsigned)nz);
    const bool vec = !novec && eh_gemm_vec_ok(g, ATR, BTR);
    static const long long t128_min = getenv("EH_GEMM_T128_MIN") ? atoll(getenv("EH_GEMM_T128_MIN")) : 2048;
    if (vec && (long long)grid.x * grid.y * grid.z < t128_min) {      // fewer 128 x 128 tiles than CUs: 64 x 64 ones (four times the workgroups, a quarter of the work each)
        const dim3 grid64((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)nz);
        hipLaunchKernelGGL((eh_gemm_kernel<ATR, BTR, EPI, true, 64>), grid64, dim3(256), 0, h->stream, g);
        return;
    }
    if (vec) hipLaunchKernelGGL((eh_gemm_kernel<ATR, BTR, EPI, true>), grid, dim3(256), 0, h->stream, g);
    else hipLaunchKernelGGL((eh_gemm_kernel<ATR, BTR, EPI, false>), grid, dim3(256), 0, h->stream, g);
}
// minibatch -> Xb (input BatchNorm applied), forward through every Dense layer; O^T [K][ldo] = the raw NN outputs
// (lstop >= 0: a one-network model, only its layers below `lstop` run here -- the rest belongs to eh_lform_tailchain_kernel)
static int lform_forward(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long count, bool train_mode, bool bn_update, const EhLWs& W, int lstop = -1) {
    const EhNet& net = h->net;
    const int B = (int)count;
    if (h->l_nnets == 0) return EH_OK;        // no network (no neural parameter): the mechanistic stage reads the records itself
    EhStepArgs bn{};
    // (small minibatches: the prep kernel takes the batch statistics itself -- one dependent launch fewer)
    static const bool nofuse = getenv("EH_LFORM_NOFUSE") != nullptr;
    const bool bn_self = train_mode && h->bn_on && !h->bn_ext && count > 0 && count <= 256 && !nofuse;
    if (train_mode && !bn_self) { if (int rc = bn_prepare(h, sp, idx, first, count, bn_update, &bn)) return rc; }
    EhLPrepArgs pa{};
    pa.recs = sp.recs; pa.C = h->C; pa.P = net.P; pa.idx = idx; pa.first = first; pa.count = B; pa.Xb = W.Xb; pa.meta = h->image;
    pa.bn_part = bn.bn_part; pa.bn_nblk = bn.bn_nblk; pa.bn_c = bn.bn_c; pa.bn_n = bn.bn_n; pa.bn_update = bn.bn_update; pa.bn_run = h->bn_run;
    if (bn_self) { pa.bn_self = 1; pa.bn_update = bn_update ? 1 : 0; }
    static const bool stamp_first = getenv("EH_STAMP_FIRST") != nullptr;      // (diagnostic builds: the stamps of the first launch instead of the chain kernel's)
    pa.stamps = stamp_first ? h->stamps : nullptr;
    const long long tot = (long long)B * net.P;
    const float* theta = TH(h);
    // Few rows: minibatch matrix + BatchNorm + first layer + the SECOND layer's few-rows product as one launch (eh_fewrows_first_kernel) where
    // the second layer is a hidden layer that would run eh_fewrows_gemm_kernel anyway
    bool fused01 = false;
    {
        static const bool nofirst = getenv("EH_LFORM_NOFIRST") != nullptr;
        static const bool nofew = getenv("EH_GEMM_NOFEWROWS") != nullptr;
        static const int first_maxb = getenv("EH_LFORM_FIRST_MAXB") ? atoi(getenv("EH_LFORM_FIRST_MAXB")) : 64;   // every workgroup takes the statistics itself: a loss from ~128 rows up
        const eh_handle_s::LNet& L0 = h->l_net[0];
        const int stop = lstop >= 0 ? lstop : L0.nl;
        auto al16 = [](const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0; };
        if (!nofirst && !nofuse && !nofew && !g_gemm_novec && L0.nl >= 3 && stop >= 2 && L0.in[0] <= 4 && B >= 1 && B <= first_maxb &&
            L0.out[0] >= 64 && (L0.out[0] % 16) == 0 && (L0.out[1] % 4) == 0 &&
            al16(theta + L0.woff[0]) && al16(theta + L0.boff[0]) && al16(theta + L0.woff[1])) {
            EhGemmArgs f{};
            f.lda = net.P; f.B = theta + L0.woff[0]; f.ldb = L0.out[0]; f.M = B; f.N = L0.out[0]; f.K = L0.in[0]; f.kchunk = f.K;
            f.bias = theta + L0.boff[0]; f.act = L0.lact[0]; f.C = W.H[0][0]; f.ldc = L0.out[0]; f.Z = W.Z[0][0];
            EhGemmArgs g{};
            g.A = nullptr; g.lda = L0.in[1]; g.B = theta + L0.woff[1]; g.ldb = L0.out[1];
            g.M = B; g.N = L0.out[1]; g.K = L0.in[1]; g.c_zstride = 0;
            g.bias = theta + L0.boff[1]; g.act = L0.lact[1]; g.C = W.H[0][1]; g.ldc = L0.out[1]; g.Z = W.Z[0][1];
            g.kchunk = std::max(16, ((g.K + 15) / 16 + 15) / 16 * 16);         // <= 16 slices of whole 16-deep groups (lform_gemm's few-rows rule)
            const int nwv = (g.K + g.kchunk - 1) / g.kchunk;
            if (f.K <= 2) hipLaunchKernelGGL((eh_fewrows_first_kernel<EH_GEPI_BIAS_ACT, 2>), dim3((unsigned)((g.N + 15) / 16), (unsigned)((g.M + 15) / 16)), dim3(64u * (unsigned)nwv), 0, h->stream, pa, f, g, L0.c0);
            else hipLaunchKernelGGL((eh_fewrows_first_kernel<EH_GEPI_BIAS_ACT, 4>), dim3((unsigned)((g.N + 15) / 16), (unsigned)((g.M + 15) / 16)), dim3(64u * (unsigned)nwv), 0, h->stream, pa, f, g, L0.c0);
            HIPCHK(h, hipGetLastError());
            fused01 = true;
        }
    }
    // the first network's first layer rides along when it is one of the few-predictor products (K <= 8, hidden layer behind it)
    bool fused0 = false;
    if (!fused01) {
        const eh_handle_s::LNet& L0 = h->l_net[0];
        if (!nofuse && !g_gemm_novec && L0.nl >= 2 && L0.in[0] <= 8 && lstop != 0) {
            EhGemmArgs g{};
            g.A = nullptr; g.lda = net.P; g.B = theta + L0.woff[0]; g.ldb = L0.out[0];
            g.M = B; g.N = L0.out[0]; g.K = L0.in[0]; g.kchunk = g.K; g.bias = theta + L0.boff[0]; g.act = L0.lact[0];
            g.C = W.H[0][0]; g.ldc = L0.out[0]; g.Z = W.Z[0][0];
            const long long totc = (long long)g.M * g.N;
            hipLaunchKernelGGL((eh_lform_prep_kernel<true>), dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (std::max(tot, totc) + 255) / 256))), dim3(256), 0, h->stream, pa, g, L0.c0);
            fused0 = true;
        }
    }
    if (!fused0 && !fused01) hipLaunchKernelGGL((eh_lform_prep_kernel<false>), dim3((unsigned)std::max<long long>(1, std::min<long long>(1024, (tot + 255) / 256))), dim3(256), 0, h->stream, pa, EhGemmArgs{}, 0);
    HIPCHK(h, hipGetLastError());
    for (int k = 0; k < h->l_nnets; ++k) {
        const eh_handle_s::LNet& L = h->l_net[k];
        for (int l = 0; l < (lstop >= 0 ? lstop : L.nl); ++l) {
            if ((fused0 || fused01) && k == 0 && l == 0) continue;
            if (fused01 && k == 0 && l == 1) continue;
            EhGemmArgs g{};
            g.A = l == 0 ? W.Xb + L.c0 : W.H[k][l - 1]; g.lda = l == 0 ? net.P : L.in[l];      // (a network's predictors: its columns of the minibatch matrix)
            g.B = theta + L.woff[l]; g.ldb = L.out[l];                   // canonical (out, in) column-major == [in][out] row-major
            g.M = B; g.N = L.out[l]; g.K = L.in[l]; g.kchunk = g.K; g.c_zstride = 0;
            g.bias = theta + L.boff[l]; g.act = L.lact[l];
            if (l + 1 < L.nl) { g.C = W.H[k][l]; g.ldc = L.out[l]; g.Z = W.Z[k][l]; lform_gemm<false, false, EH_GEPI_BIAS_ACT>(h, g, 1); }
            else { g.C = W.O + (long long)L.orow * W.ldo; g.ldc = W.ldo; lform_gemm<false, false, EH_GEPI_BIAS_T>(h, g, 1); }
            HIPCHK(h, hipGetLastError());
        }
    }
    return EH_OK;
}
// targets whose training loss takes two forward passes ahead of the training pass (batch statistics of yhat): bit t
static unsigned two_pass_mask(const EhNet& net) {
    unsigned m = 0;
    for (int t = 0; t < net.T; ++t) {
        const unsigned k = (net.loss_t >> (4 * t)) & 15u;
        if ((k >= (unsigned)EH_LOSS_PEARSONLOSS && k <= (unsigned)EH_LOSS_PBKGELOSS) || (k == (unsigned)EH_LOSS_RMSE && net.T > 1)) m |= 1u << t;
    }
    return m;
}
unsigned eh_two_pass_mask(const eh_handle* h) { return two_pass_mask(h->net); }
// one training step's gradient sums into `rows` partial slab rows (the contract of the fused step kernels: eh_reduce_kernel follows)
static int lform_train(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long count, int* rows_out, bool bn_update) {
    const EhNet& net = h->net;
    EhLWs W;
    if (int rc = lform_workspace(h, std::max<long long>(count, 1), &W)) return rc;
    const int B = (int)count;
    // slab rows = sample chunks the weight-gradient products are split over (their only source of parallelism beyond the tiles of the
    // weight matrix itself): >= EH_LFORM_CHUNK samples each, at most EH_LFORM_ROWS of them
    static const long long lchunk = getenv("EH_LFORM_CHUNK") ? std::max(64, atoi(getenv("EH_LFORM_CHUNK"))) : 128;
    const int rows = (int)std::max<long long>(1, std::min<long long>(EH_LFORM_ROWS, (count + lchunk - 1) / lchunk));
    const int chunk = std::max(16, (int)(((count + rows - 1) / rows + 15) / 16 * 16));
    *rows_out = rows;
    if (net.T > 1 && !h->dp_weights) {        // (data-parallel step: eh_dp_grad has just filled inv_n with the weights of the GLOBAL batch)
        EhShift4 sh4; for (int t = 0; t < EH_MAX_TARG; ++t) sh4.c[t] = sp.shift[t];
        hipLaunchKernelGGL(eh_count_kernel, dim3(net.T), dim3(256), 0, h->stream, sp.recs, h->C, net.P + net.F, net.T, idx, first, count, h->inv_n, net.loss_t, sh4, h->img.agg_a, h->roles, h->img.l2s);
        HIPCHK(h, hipGetLastError());
    }
    if (count <= 0) {                          // nothing to do: an all-zero partial (the reduce kernel then skips the update)
        HIPCHK(h, hipMemsetAsync(h->slab, 0, (size_t)h->n_acc * sizeof(float), h->stream));
        return EH_OK;
    }
    const unsigned tpm = two_pass_mask(net);
    // Small minibatches: every layer's delta is kept (l_dk), the chain of delta products runs first and the weight gradients -- a
    // handful of tiles each, 4-6 us per dependent launch -- follow as ONE grouped launch of the tiled ones and one of the thin ones.
    static const long long lgroup_max = getenv("EH_LFORM_GROUP_MAX") ? atoll(getenv("EH_LFORM_GROUP_MAX")) : 4096;
    bool grouped = count <= lgroup_max && h->l_nnets > 0;
    long long dk_floats = 0;
    // (sized by THIS minibatch, rounded up to a power of two -- not by the largest one the path serves: a B = 64 step of a deep, wide
    //  MultiNN model used to ask for up to the 1 GiB cap; advisor, round 3.  A larger batch later re-grows it, outside a capture.)
    long long dk_rows = 64;
    while (dk_rows < count) dk_rows *= 2;
    if (grouped) {
        for (int k = 0; k < h->l_nnets; ++k)
            for (int l = 0; l + 1 < h->l_net[k].nl; ++l) dk_floats += dk_rows * h->l_net[k].out[l];
        if (dk_floats > (1ll << 28)) { grouped = false; h->jit_log = "layer-wise form: kept deltas of this minibatch exceed 1 GiB -- weight gradients run layer by layer (slower small-batch steps)"; }
        else if ((size_t)dk_floats > h->l_dk_cap) {
            if (h->capturing) return lform_alloc_in_capture(h, "kept deltas");
            HIPCHK(h, hipStreamSynchronize(h->stream));
            (void)hipFree(h->l_dk); h->l_dk = nullptr; h->l_dk_cap = 0;
            if (hipMalloc(&h->l_dk, (size_t)dk_floats * sizeof(float)) == hipSuccess) h->l_dk_cap = (size_t)dk_floats;
            else { (void)hipGetLastError(); grouped = false; h->jit_log = "layer-wise form: no memory for the kept deltas of the grouped small-batch path -- weight gradients run layer by layer (slower small-batch steps)"; }
        }
    }
    // Few rows, one network: the narrow end of the network -- every layer from `tail_s` on, the mechanistic stage and the deltas back
    // down to layer tail_s - 1 -- is ONE launch with a workgroup per row (eh_lform_tailchain_kernel, eh_lform.hpp)
    static const bool notail = getenv("EH_LFORM_NOTAIL") != nullptr;
    int tail_s = -1;
    // (a workgroup per ROW up to 256 rows: every CU streams the suffix's weights once whatever the number of rows -- 57.5 / 61.3 / 70.1 / 77.5 us
    //  per step at 96 / 128 / 192 / 256 rows against 70.9 / 74.2 / 80.1 / 86.6 with the products launched one by one.  With FOUR rows per
    //  workgroup, the form of the round's first half for more than 64 rows, it was slower than either: 75.5 / 78.6 / 90.2 / 96.7; those
    //  instantiations are gone.)
    if (!notail && grouped && h->l_nnets == 1 && count <= 256 && !tpm && !g_gemm_novec) {
        const eh_handle_s::LNet& L = h->l_net[0];
        long long wsum = 0;
        for (int l = L.nl - 1; l >= 0; --l) {
            if (L.in[l] > (int)EH_LTAIL_MAXW || L.out[l] > (int)EH_LTAIL_MAXW || wsum + (long long)L.in[l] * L.out[l] > (96ll << 10)) break;
            wsum += (long long)L.in[l] * L.out[l];
            tail_s = l;
        }
    }
    if (int rc = lform_forward(h, sp, idx, first, count, true, bn_update, W, tail_s)) return rc;
    EhStepArgs a{};
    a.prog = h->prog; a.recs = sp.recs; a.C = h->C; a.idx = idx; a.first = first; a.count = count; a.stamps = (getenv("EH_STAMP_DW") || getenv("EH_STAMP_FIRST")) ? nullptr : h->stamps;
    for (int t = 0; t < EH_MAX_TARG; ++t) a.shift[t] = sp.shift[t];
    if (tpm) {
        // losses that need batch statistics of yhat (see launch_train_kernel): the NN outputs O stay where the forward left them, so
        // the two statistics passes are two runs of the mechanistic stage alone (eval form), not two forwards
        EhStepArgs e = a;
        e.inv_n = nullptr; e.yld = count;
        const int egrid = (int)std::min<long long>(256, (count + 255) / 256);
        EhLMechArgs me{W.O, W.ldo, h->slab};                   // [egrid][EH_EVAL_STATS * T] (the slab is rewritten by the training pass afterwards)
        EhShift4 s4; for (int t = 0; t < EH_MAX_TARG; ++t) s4.c[t] = sp.shift[t];
        for (int pass = 0; pass < 2; ++pass) {
            if (net.mech == EH_MECH_PROGRAM) hipLaunchKernelGGL((eh_lform_mech_kernel<false, true>), dim3(egrid), dim3(256), 0, h->stream, net, e, me, h->image);
            else hipLaunchKernelGGL((eh_lform_mech_kernel<false, false>), dim3(egrid), dim3(256), 0, h->stream, net, e, me, h->image);
            HIPCHK(h, hipGetLastError());
            if (pass == 0) hipLaunchKernelGGL(eh_moment_centre_kernel, dim3(net.T), dim3(64), 0, h->stream, h->slab, egrid, net.T, net.loss_t, s4, h->inv_n);
            else hipLaunchKernelGGL(eh_moment_coef_kernel, dim3(net.T), dim3(64), 0, h->stream, h->slab, egrid, net.T, net.loss_t, s4, h->inv_n, h->img.agg_a);
            HIPCHK(h, hipGetLastError());
            e.inv_n = h->inv_n;
        }
    }
    a.inv_n = (net.T > 1 || tpm) ? h->inv_n : nullptr;
    // recorded loss functions: no kernel is compiled at run time in this form -- the mechanistic kernel interprets their tapes
    bool lprog = false;
    for (int t = 0; t < net.T; ++t) lprog = lprog || ((net.loss_t >> (4 * t)) & 15u) == (unsigned)EH_LOSS_PROGRAM;
    if (lprog) {
        if (h->l_lprog_gen != h->loss_prog.gen) {
            if (h->capturing) return lform_alloc_in_capture(h, "recorded loss programs");
            std::vector<unsigned> img((size_t)EH_MAX_TARG * EH_LPROG_WORDS, 0u);
            for (int t = 0; t < net.T; ++t) {
                if (!h->loss_prog.has(t)) continue;
                const EhLossProg1& lp = h->loss_prog.of(t);
                unsigned* q = img.data() + (size_t)t * EH_LPROG_WORDS;
                q[0] = (unsigned)lp.code.size(); q[1] = 1u; q[2] = (unsigned)lp.out;
                for (size_t k = 0; k < lp.consts.size() && k < (size_t)EH_MAX_PROG_CONST; ++k) memcpy(&q[8 + k], &lp.consts[k], sizeof(float));
                for (size_t i = 0; i < lp.code.size() && i < (size_t)EH_MAX_PROG; ++i) q[24 + i] = lp.code[i];
            }
            if (!h->l_lprog) HIPCHK(h, hipMalloc(&h->l_lprog, img.size() * sizeof(unsigned)));
            HIPCHK(h, hipStreamSynchronize(h->stream));          // (steps in flight read the old programs)
            HIPCHK(h, hipMemcpy(h->l_lprog, img.data(), img.size() * sizeof(unsigned), hipMemcpyHostToDevice));
            h->l_lprog_gen = h->loss_prog.gen;
        }
        a.lprog = h->l_lprog;
    }
    EhLMechArgs m{W.O, W.ldo, W.part, nullptr, 0, 0};
    const int mgrid = (int)std::min<long long>(2048, (count + 255) / 256);
    static const bool nofuse = getenv("EH_LFORM_NOFUSE") != nullptr;
    if (mgrid == 1 && !nofuse) { m.slab = h->slab; m.nrows = rows; m.n_acc = (long long)h->n_acc; }      // one workgroup: it writes the slab's tail columns itself
    EhLTailJob tjob{nullptr, 0, nullptr, 0, 0};
    if (tail_s >= 0) {
        const eh_handle_s::LNet& L = h->l_net[0];
        const float* theta = TH(h);
        EhLTailArgs t{};
        t.nl = L.nl - tail_s;
        // dZ_l of the hidden layers where the grouped weight gradients will look for them (the order of the backward loop below: from the top)
        float* dptr[EH_MAX_HIDDEN + 1] = {nullptr};
        { float* q = h->l_dk; for (int l = L.nl - 1; l >= 1; --l) { dptr[l - 1] = q; q += dk_rows * L.in[l]; } }
        int wmax = 4;
        for (int j = 0; j < t.nl; ++j) {
            const int l = tail_s + j;
            EhLTailLayer& T = t.L[j];
            T.W = theta + L.woff[l]; T.b = theta + L.boff[l]; T.in = L.in[l]; T.out = L.out[l]; T.act = L.lact[l];
            T.vec = ((reinterpret_cast<unsigned long long>(T.W) & 15ull) == 0 && (T.out & 3) == 0) ? 1 : 0;
            const bool hidden = l + 1 < L.nl;
            T.H = hidden ? W.H[0][l] : nullptr; T.Z = hidden ? W.Z[0][l] : nullptr; T.D = hidden ? dptr[l] : nullptr;
            eh_ltail_geometry(T);
            if (hidden && T.act == EH_ACT_SWISH) t.any_swish = 1;
            wmax = std::max(wmax, std::max(T.in, T.out));
        }
        (void)wmax;
        t.wmax = (int)EH_LTAIL_MAXW;        // (rows of EH_LTAIL_MAXW + EH_LTAIL_PAD floats whatever the widths: a product reads on past a row's width into zeros, up to the next power of two)
        t.Hin = tail_s == 0 ? W.Xb + L.c0 : W.H[0][tail_s - 1]; t.ldin = tail_s == 0 ? net.P : L.in[tail_s];
        t.act_below = tail_s > 0 ? L.lact[tail_s - 1] : EH_ACT_IDENTITY;
        t.Zin = (tail_s > 0 && t.act_below == EH_ACT_SWISH) ? W.Z[0][tail_s - 1] : nullptr;
        if (t.Zin) t.any_swish = 1;
        t.Dbelow = tail_s > 0 ? dptr[tail_s - 1] : nullptr;
        t.O = W.O + (long long)L.orow * W.ldo; t.ldo = W.ldo; t.part = W.part;
        const int R = 1;
        const int tgrid = (int)((count + R - 1) / R);
        const size_t lds0 = eh_ltail_lds_bytes(R, t.nl, t.wmax, t.any_swish != 0);
        const bool mp = net.mech == EH_MECH_PROGRAM;
        const void* fn = lprog ? (mp ? (const void*)&eh_lform_tailchain_kernel<1, true, true> : (const void*)&eh_lform_tailchain_kernel<1, false, true>)
                               : (mp ? (const void*)&eh_lform_tailchain_kernel<1, true, false> : (const void*)&eh_lform_tailchain_kernel<1, false, false>);
        // one or two hidden layers + the output layer whose weights fit the threads' registers: they stay there for the delta products (eh_lform_tailkeep_kernel)
        static const bool nokeep = getenv("EH_LFORM_NOKEEP") != nullptr;
        int trf = 0;
        const bool keep = !nokeep && R == 1 && eh_ltail_keep_ok(t, (int)count, &trf);
        size_t lds = lds0;
        if (keep) {
            lds = lds0 + ((size_t)trf + 8) * sizeof(float);
            fn = lprog ? (mp ? (const void*)&eh_lform_tailkeep_kernel<true, true> : (const void*)&eh_lform_tailkeep_kernel<false, true>)
                       : (mp ? (const void*)&eh_lform_tailkeep_kernel<true, false> : (const void*)&eh_lform_tailkeep_kernel<false, false>);
        }
        if (lds > EH_LDS_LIMIT) return fail(h, EH_EUNSUPPORTED, "layer-wise form: %zu bytes of LDS for the tail chain", lds);
        bool& prepared = h->l_tail_fn[(keep ? 8 : 0) + (lprog ? 2 : 0) + (mp ? 1 : 0)];      // (per handle = per device)
        if (!prepared) { HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EH_LDS_LIMIT)); prepared = true; }
        void* kargs[] = {(void*)&net, (void*)&a, (void*)&t, (void*)&h->image};
        HIPCHK(h, hipLaunchKernel(fn, dim3((unsigned)tgrid), dim3(EH_LTAIL_THREADS), kargs, lds, h->stream));
        tjob = EhLTailJob{W.part, tgrid, h->slab, rows, (long long)h->n_acc};
    } else
    if (lprog) {
        if (net.mech == EH_MECH_PROGRAM) hipLaunchKernelGGL((eh_lform_mech_kernel<true, true, true>), dim3(mgrid), dim3(256), 0, h->stream, net, a, m, h->image);
        else hipLaunchKernelGGL((eh_lform_mech_kernel<true, false, true>), dim3(mgrid), dim3(256), 0, h->stream, net, a, m, h->image);
    }
    else if (net.mech == EH_MECH_PROGRAM) hipLaunchKernelGGL((eh_lform_mech_kernel<true, true>), dim3(mgrid), dim3(256), 0, h->stream, net, a, m, h->image);
    else hipLaunchKernelGGL((eh_lform_mech_kernel<true, false>), dim3(mgrid), dim3(256), 0, h->stream, net, a, m, h->image);
    HIPCHK(h, hipGetLastError());
    if (!m.slab && tail_s < 0) {
        hipLaunchKernelGGL(eh_lform_tail_kernel, dim3(1), dim3(256), 0, h->stream, W.part, mgrid, net, h->slab, rows, (long long)h->n_acc);
        HIPCHK(h, hipGetLastError());
    }
    // backward, every network from its output layer down; dZ of the output layer = its rows of d loss / d O^T, still [K][ldo]
    const float* theta = TH(h);
    EhGemmGroup GG{}; EhThinGroup TG{};
    float* tot_job = nullptr;              // where a delta product's side job leaves the sums of the chain's partial rows (only the few-rows product kernel takes the job: l_job_done)
    h->l_job_done = false;
    bool all_grouped = true;               // every weight-gradient product of the step sits in GG / TG (none launched on its own, no group flushed early)
    auto flush_tiled = [&]() {
        if (GG.n > 0) hipLaunchKernelGGL((eh_gemm_group_kernel<true, false, EH_GEPI_STORE, true, 64>), dim3((unsigned)GG.t0[GG.n]), dim3(256), 0, h->stream, GG);
        GG.n = 0;
    };
    auto flush_thin = [&]() {
        if (TG.n > 0) hipLaunchKernelGGL(eh_thin_gemm_group_kernel, dim3((unsigned)TG.t0[TG.n]), dim3(256), 0, h->stream, TG);
        TG.n = 0;
    };
    float* dkp = h->l_dk;
    for (int k = 0; k < h->l_nnets; ++k) {
        const eh_handle_s::LNet& L = h->l_net[k];
        const float* dZ = W.O + (long long)L.orow * W.ldo;
        bool dz_t = true;                      // dZ stored transposed ([out][B])
        int which = 0;
        for (int l = L.nl - 1; l >= 0; --l) {
            const int in = L.in[l], out = L.out[l];
            const float* Hprev = l == 0 ? W.Xb + L.c0 : W.H[k][l - 1];
            const long long ldp = l == 0 ? net.P : in;
            EhGemmArgs g{};                    // dW_l^T [in x out] = Hprev^T [in x B] * dZ_l [B x out], split over the samples
            g.A = Hprev; g.lda = ldp; g.B = dZ; g.ldb = dz_t ? W.ldo : out;
            g.C = h->slab + L.woff[l]; g.ldc = out; g.M = in; g.N = out; g.K = B; g.kchunk = chunk; g.c_zstride = h->n_acc;
            g.colsum = h->slab + L.boff[l];      // db_l = column sums of dZ_l, from the same tiles
            EhThinArgs ta;
            if (grouped && lform_thin_args(g, dz_t, &ta)) {
                if (TG.n == EH_GEMM_GROUP) { flush_thin(); all_grouped = false; HIPCHK(h, hipGetLastError()); }
                const int gx = (ta.ncols + 63) / 64;
                TG.a[TG.n] = ta; TG.gx[TG.n] = gx; TG.t0[TG.n + 1] = TG.t0[TG.n] + gx * rows; ++TG.n;
            } else if (grouped && !dz_t && eh_gemm_vec_ok(g, true, false)) {
                if (GG.n == EH_GEMM_GROUP) { flush_tiled(); all_grouped = false; HIPCHK(h, hipGetLastError()); }
                const int gx = (g.N + 63) / 64, gy = (g.M + 63) / 64;
                GG.g[GG.n] = g; GG.gx[GG.n] = gx; GG.gy[GG.n] = gy; GG.t0[GG.n + 1] = GG.t0[GG.n] + gx * gy * rows; ++GG.n;
            } else {
                // (in grouped mode too: dZ_l stays where it is until the step ends)
                all_grouped = false;
                if (dz_t) lform_gemm<true, true, EH_GEPI_STORE>(h, g, rows); else lform_gemm<true, false, EH_GEPI_STORE>(h, g, rows);
                HIPCHK(h, hipGetLastError());
            }
            if (l > 0) {                       // dZ_{l-1} [B x in] = (dZ_l [B x out] * W_l [out x in]) .* act'(H_{l-1})  (swish: act' from the stored Z_{l-1})
                EhGemmArgs b{};
                b.A = dZ; b.lda = dz_t ? W.ldo : out; b.B = theta + L.woff[l]; b.ldb = out;        // W_l element (k = out, n = in) at n * out + k
                float* const dnext = grouped ? dkp : W.D[which];
                if (grouped) dkp += dk_rows * in;
                if (!(tail_s >= 0 && l >= tail_s)) {      // (the tail chain has left this delta where the weight gradients look for it)
                    if (tjob.part && !h->l_job_done && h->l_apply && rows == 1) {      // (its first workgroup also adds up the chain's rows of partial sums: eh_dw_apply_kernel finds them ready)
                        b.job_part = tjob.part; b.job_nblk = tjob.nblk; b.job_out = W.part + (size_t)2048 * EH_LMECH_PART;
                        tot_job = b.job_out;
                    }
                    b.C = dnext; b.ldc = in; b.M = B; b.N = in; b.K = out; b.kchunk = out; b.c_zstride = 0;
                    b.H = L.lact[l - 1] == EH_ACT_SWISH ? W.Z[k][l - 1] : W.H[k][l - 1]; b.ldh = in; b.act = L.lact[l - 1];
                    if (dz_t) lform_gemm<true, true, EH_GEPI_DACT>(h, b, 1); else lform_gemm<false, true, EH_GEPI_DACT>(h, b, 1);
                    HIPCHK(h, hipGetLastError());
                }
                dZ = dnext; dz_t = false; which ^= 1;
            }
        }
    }
    static const bool nodwmerge = getenv("EH_LFORM_NODWMERGE") != nullptr;
    static const bool noapply = getenv("EH_LFORM_NOAPPLY") != nullptr;
    if (h->l_apply && !noapply && tjob.part && all_grouped && rows == 1) {
        // one slab row: the products ARE the gradient -- the optimiser runs in their epilogues (eh_dw_apply_kernel), no reduce launch behind them
        EhLApply ap = *h->l_apply;
        ap.slab = h->slab; ap.part = tjob.part; ap.nblk = tjob.nblk; ap.tot = h->l_job_done ? tot_job : nullptr;
        static const bool stamp_dw = getenv("EH_STAMP_DW") != nullptr;      // (diagnostic builds: the stamps of this launch instead of the chain kernel's)
        ap.stamps = stamp_dw ? h->stamps : nullptr;
        ap.stamp_wg = stamp_dw ? atoi(getenv("EH_STAMP_DW")) : 0;
        if (ap.stamp_wg < 0) ap.stamp_wg += TG.t0[TG.n] + GG.t0[GG.n] + 1;
        static const bool noapply64 = getenv("EH_LFORM_NOAPPLY64") != nullptr;
        if (B <= 64 && ap.tot && !noapply64) {       // everything requested at once, 32 x 32 tiles with the samples split over the waves (eh_dw_apply64_kernel)
            for (int i = 0; i < GG.n; ++i) {
                GG.gx[i] = (GG.g[i].N + 31) / 32; GG.gy[i] = (GG.g[i].M + 31) / 32;
                GG.t0[i + 1] = GG.t0[i] + GG.gx[i] * GG.gy[i];
            }
            if (ap.stamp_wg < 0) ap.stamp_wg = TG.t0[TG.n] + GG.t0[GG.n];
#ifdef EH_STAMPS
            if (ap.stamps) { hipMemsetAsync(ap.stamps + 28, 0xff, 8, h->stream); hipMemsetAsync(ap.stamps + 30, 0, 8, h->stream); }
#endif
            hipLaunchKernelGGL(eh_dw_apply64_kernel, dim3((unsigned)(TG.t0[TG.n] + 8 * ((GG.t0[GG.n] + 7) / 8) + 1)), dim3(256), 0, h->stream, GG, TG, net, ap);
        } else
        hipLaunchKernelGGL(eh_dw_apply_kernel, dim3((unsigned)(TG.t0[TG.n] + GG.t0[GG.n] + 1)), dim3(256), 0, h->stream, GG, TG, net, ap);
        HIPCHK(h, hipGetLastError());
        h->l_applied = true;
        return EH_OK;
    }
    if (tjob.part || (TG.n > 0 && GG.n > 0 && !nodwmerge)) {       // what is left of both groups: one launch (+ the workgroup that sums the tail chain's partial rows)
        hipLaunchKernelGGL(eh_dw_group_kernel, dim3((unsigned)(TG.t0[TG.n] + GG.t0[GG.n] + (tjob.part ? 1 : 0))), dim3(256), 0, h->stream, GG, TG, tjob, net);
        TG.n = 0; GG.n = 0;
        HIPCHK(h, hipGetLastError());
    }
    flush_thin(); HIPCHK(h, hipGetLastError());
    flush_tiled(); HIPCHK(h, hipGetLastError());
    return EH_OK;
}
// forward + metric sums of samples [first, first+count) in chunks; per-workgroup rows of EH_EVAL_STATS * T sums land in the slab
static int lform_eval(eh_handle* h, const EhSplit& sp, long long first, long long count, float* yhat, float* pout, int* rows_out) {
    const EhNet& net = h->net;
    const long long CH = 65536;
    int rows = 0;
    const int ncol = EH_EVAL_STATS * net.T;
    for (long long c0 = 0; c0 < count; c0 += CH) {
        const long long n = std::min(CH, count - c0);
        EhLWs W;
        if (int rc = lform_workspace(h, n, &W)) return rc;
        if (int rc = lform_forward(h, sp, nullptr, first + c0, n, false, false, W)) return rc;
        EhStepArgs a{};
        a.prog = h->prog; a.recs = sp.recs; a.C = h->C; a.idx = nullptr; a.first = first + c0; a.count = n;
        a.yhat = yhat ? yhat + c0 : nullptr; a.pout = pout ? pout + c0 : nullptr; a.yld = count;
        for (int t = 0; t < EH_MAX_TARG; ++t) a.shift[t] = sp.shift[t];
        const int mgrid = (int)std::min<long long>(256, (n + 255) / 256);
        if ((size_t)(rows + mgrid) * ncol > std::max((size_t)h->slab_rows * std::max(h->n_acc, ncol), (size_t)1 << 20)) return fail(h, EH_ENOMEM, "eh_eval: window too large for the metric rows");
        EhLMechArgs m{W.O, W.ldo, h->slab + (size_t)rows * ncol};
        if (net.mech == EH_MECH_PROGRAM) hipLaunchKernelGGL((eh_lform_mech_kernel<false, true>), dim3(mgrid), dim3(256), 0, h->stream, net, a, m, h->image);
        else hipLaunchKernelGGL((eh_lform_mech_kernel<false, false>), dim3(mgrid), dim3(256), 0, h->stream, net, a, m, h->image);
        HIPCHK(h, hipGetLastError());
        rows += mgrid;
    }
    *rows_out = std::max(rows, 0);
    return EH_OK;
}

static int launch_train_kernel(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long count, int* grid_out, bool bn_update) {
    if (h->lform) return lform_train(h, sp, idx, first, count, grid_out, bn_update);
    const EhNet& net = h->net;
    if (net.T > 1 && !h->dp_weights) {
        EhShift4 sh4; for (int t = 0; t < EH_MAX_TARG; ++t) sh4.c[t] = sp.shift[t];
        hipLaunchKernelGGL(eh_count_kernel, dim3(net.T), dim3(256), 0, h->stream, sp.recs, h->C, net.P + net.F, net.T, idx, first, count, h->inv_n, net.loss_t, sh4, h->img.agg_a, h->roles, h->img.l2s);
        HIPCHK(h, hipGetLastError());
    }
    const bool moment_loss = two_pass_mask(net) != 0;
    if (moment_loss && !h->dp_moments) {      // (data-parallel step: eh_dp_grad has just made the coefficients from the moments of the GLOBAL batch)
        // forward-only passes (train-mode BatchNorm statistics included): the batch mean of yhat, then the moments of
        // (yhat, y) about the means -> the coefficients of the per-sample d loss / d yhat that the training pass multiplies
        // into the VJP
        EhStepArgs e{};
        e.prog = h->prog;
        e.recs = sp.recs; e.C = h->C; e.idx = idx; e.first = first; e.count = count;
        e.image = h->image; e.slab = h->slab; e.n_acc = EH_EVAL_STATS * net.T; e.rmap = h->rmap; e.stamps = nullptr;
        e.yld = count;
        for (int t = 0; t < EH_MAX_TARG; ++t) e.shift[t] = sp.shift[t];
        if (int rc = bn_prepare(h, sp, idx, first, count, false, &e)) return rc;
        const int egrid = count > 0 ? grid_for(h, count) : 1;
        HIPCHK(h, step_launch(h, EH_MODE_EVAL, egrid, &e));       // -> mean of yhat
        EhShift4 s4; for (int t = 0; t < EH_MAX_TARG; ++t) s4.c[t] = sp.shift[t];
        hipLaunchKernelGGL(eh_moment_centre_kernel, dim3(net.T), dim3(64), 0, h->stream, h->slab, egrid, net.T, net.loss_t, s4, h->inv_n);
        HIPCHK(h, hipGetLastError());
        e.inv_n = h->inv_n;                                                                                               // -> moments about it
        HIPCHK(h, step_launch(h, EH_MODE_EVAL, egrid, &e));
        hipLaunchKernelGGL(eh_moment_coef_kernel, dim3(net.T), dim3(64), 0, h->stream, h->slab, egrid, net.T, net.loss_t, s4, h->inv_n, h->img.agg_a);
        HIPCHK(h, hipGetLastError());
    }
    EhStepArgs a{};
    a.prog = h->prog;
    a.recs = sp.recs; a.C = h->C; a.idx = idx; a.first = first; a.count = count;
    a.image = h->image; a.slab = h->slab; a.n_acc = h->n_acc;
    a.inv_n = (net.T > 1 || moment_loss) ? h->inv_n : nullptr;
    for (int t = 0; t < EH_MAX_TARG; ++t) a.shift[t] = sp.shift[t];
    a.rmap = h->rmap;
    // one network on a row-split bf16 kernel: the accumulators go to the slab row straight from the registers (canonical order is plain
    // column-major per layer; eh_wide_bf16.hpp) -- signalled by a null map
    static const bool no_direct = getenv("EH_NO_DIRECT_STORE") != nullptr;      // (A/B switch of the measurement tools)
    if (!no_direct && h->arch->wide && h->arch->var[h->variant].bf16 && h->n_nets == 1 && h->desc.n_nets == 0) a.rmap = nullptr;
    a.stamps = h->stamps;
    a.fz.gacc = nullptr;
    if (int rc = bn_prepare(h, sp, idx, first, count, bn_update, &a)) return rc;
    const int grid = grid_for(h, count);
    *grid_out = grid;
    HIPCHK(h, step_launch(h, EH_MODE_TRAIN, grid, &a));
    return EH_OK;
}

// one fused kernel: prologue applies the previous step's update, epilogue accumulates this step's sums
static int do_fused_step(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long count, float* loss_slot_for_this_step) {
    const bool prof = h->prof && h->ev_used + 3 <= 3 * 8192;
    const bool burst_first = h->prof_k % h->prof_stride == 0, burst_last = h->prof_k % h->prof_stride == h->prof_stride - 1;
    if (prof) {
        int rc = ensure_events(h, h->ev_used + 3);
        if (rc) return rc;
        if (burst_first) HIPCHK(h, hipEventRecord(h->ev[h->ev_used], h->stream));
        h->prof_k++;
    }
    EhStepArgs a{};
    a.prog = h->prog;
    a.recs = sp.recs; a.C = h->C; a.idx = idx; a.first = first; a.count = count;
    a.image = h->image; a.slab = h->slab; a.n_acc = h->n_acc; a.inv_n = nullptr; a.rmap = h->rmap; a.stamps = h->stamps;
    for (int t = 0; t < EH_MAX_TARG; ++t) a.shift[t] = sp.shift[t];
    if (h->net.T > 1) {      // multi-target: the per-target weights (1 / n_t, 1 / sum (y - ybar)^2) have to be known inside the streaming pass
        EhShift4 sh4; for (int t = 0; t < EH_MAX_TARG; ++t) sh4.c[t] = sp.shift[t];
        hipLaunchKernelGGL(eh_count_kernel, dim3(h->net.T), dim3(256), 0, h->stream, sp.recs, h->C, h->net.P + h->net.F, h->net.T, idx, first, count, h->inv_n, h->net.loss_t, sh4, h->img.agg_a, h->roles, h->img.l2s);
        HIPCHK(h, hipGetLastError());
        a.inv_n = h->inv_n;
    }
    EhFused& z = a.fz;
    z.gacc = h->gacc; z.pset = h->pset; z.imap = h->imap; z.loss_slot = h->pending_loss;
    z.gslot = (int)(h->gstep % 3); z.cur = h->cur; z.sc_sel = h->sc_sel; z.pending = h->pending ? 1 : 0; z.opt = h->opt; z.agg_a = h->img.agg_a;
    a.p2p = h->p2p_on ? h->p2p_dev : nullptr;
    if (h->p2p_on) a.p2pv = h->p2p_host;
    a.p2p_seq = h->p2p_on ? ++h->p2p_seq : 0u;
    if (int rc = bn_prepare(h, sp, idx, first, count, true, &a)) return rc;
    const int grid = grid_for(h, count);
    HIPCHK(h, step_launch(h, h->p2p_on ? EH_MODE_TRAIN_P2P : EH_MODE_TRAIN, grid, &a));
    h->cur ^= 1; h->sc_sel ^= 1; h->gstep++;
    h->pending = true;
    h->pending_loss = loss_slot_for_this_step;
    if (prof && burst_last) {
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 2], h->stream));
        h->ev_used += 3;
    }
    return EH_OK;
}

// Several fused-update steps in one launch (EH_MODE_TRAIN_MULTI, eh_device.hpp): minibatches one workgroup covers -- the reference's
// default batch of 64 among them -- with the step-to-step state in LDS.
static bool multi_ok(const eh_handle* h, long long batch) {
    if (!h->fused || !h->multi_step || h->lform || h->arch->wide || h->p2p_on || h->prof || h->capturing) return false;
    if (h->net.T != 1 || h->net.mech == EH_MECH_PROGRAM || h->net.loss == EH_LOSS_PROGRAM || h->act == EH_ACT_PER_NET) return false;
    if (h->bn_on && (h->bn_ext || h->bn_no_self || batch > EH_BN_SELF_MAX)) return false;
    if (h->arch->var[h->variant].lds_bytes + sizeof(float) * (size_t)eh_ms_extra_floats(h->net.n_theta, h->n_acc) > EH_LDS_LIMIT) return false;
    if (grid_for(h, batch) != 1) return false;
    if (!spec_lookup(h) && jit_wanted(h, EH_MODE_TRAIN)) {      // a model on kernels compiled at run time: its multi-step kernel, or one launch per step
        eh_handle_s::JitEntry* je = jit_entry(const_cast<eh_handle*>(h));      // (a compiled single-step kernel beats the GENERIC multi-step one: 7.4 against 9.0 us)
        if (je && (!je->verified || !je->k.fn[EH_MODE_TRAIN_MULTI])) return false;
    }
    return true;
}
static int do_fused_multi(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long batch, long long end, int nsteps, float* loss_slots) {
    EhStepArgs a{};
    a.prog = h->prog;
    a.recs = sp.recs; a.C = h->C; a.idx = idx; a.first = first; a.count = std::min(batch, end - first);
    a.image = h->image; a.slab = h->slab; a.n_acc = h->n_acc; a.inv_n = nullptr; a.rmap = h->rmap; a.stamps = h->stamps;      // (diagnostic builds: the stamps of the launch's LAST step remain)
    for (int t = 0; t < EH_MAX_TARG; ++t) a.shift[t] = sp.shift[t];
    EhFused& z = a.fz;
    z.gacc = h->gacc; z.pset = h->pset; z.imap = h->imap; z.loss_slot = h->pending_loss;
    z.gslot = (int)(h->gstep % 3); z.cur = h->cur; z.sc_sel = h->sc_sel; z.pending = h->pending ? 1 : 0; z.opt = h->opt; z.agg_a = h->img.agg_a;
    a.p2p = nullptr; a.p2p_seq = 0u;
    if (int rc = bn_prepare(h, sp, idx, first, a.count, true, &a)) return rc;       // (input BatchNorm: statistics inside the kernel, multi_ok made sure)
    a.ms_nsteps = nsteps; a.ms_batch = (int)batch; a.ms_end = end; a.ms_loss = loss_slots;
    HIPCHK(h, step_launch(h, EH_MODE_TRAIN_MULTI, 1, &a));
    // every step of the launch has applied its own update and written its own loss (eh_ms_apply): nothing is pending behind it; the first
    // step's prologue flipped the parameter set and rotated the accumulator slots once
    h->cur ^= 1; h->sc_sel ^= 1;
    h->gstep += 1;
    h->pending = false;
    h->pending_loss = nullptr;
    return EH_OK;
}

static int ensure_events(eh_handle* h, size_t need) {
    while (h->ev.size() < need) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->ev.push_back(e);
    }
    return EH_OK;
}

// fused step on the train split.  apply = update theta; loss_slot = device float for the loss.
static int do_step(eh_handle* h, const EhSplit& sp, const int* idx, long long first, long long count, bool apply, bool raw, float* loss_slot) {
    const EhNet& net = h->net;
    const bool prof = h->prof && h->ev_used + 3 <= 3 * 8192;
    const bool burst = h->prof_stride > 1, burst_first = h->prof_k % h->prof_stride == 0, burst_last = h->prof_k % h->prof_stride == h->prof_stride - 1;
    if (prof) {
        int rc = ensure_events(h, h->ev_used + 3);
        if (rc) return rc;
        if (burst_first) HIPCHK(h, hipEventRecord(h->ev[h->ev_used], h->stream));
        h->prof_k++;
    }
    int grid = 1;
    const int deferred = (net.T == 1 && !raw) ? 1 : 0;
    const unsigned tp_mask = two_pass_mask(net);
    const bool moment_loss = tp_mask != 0;
    const bool l2 = (h->img.l2c != 0.0f || h->img.l2w) && !raw;      // (data-parallel seam: raw sums only -- the extra loss is added once, in eh_dp_apply)
    float* sc_in = h->sc + 2 * h->sc_sel;
    float* sc_out = h->sc + 2 * (h->sc_sel ^ 1);
    // layer-wise form, few rows: the optimiser may run in the epilogue of the grouped weight-gradient launch (lform_train decides; eh_lform.hpp EhLApply)
    EhLApply lap{};
    h->l_apply = nullptr; h->l_applied = false;
    if (h->lform && apply && deferred && !l2 && !moment_loss && net.loss != EH_LOSS_PROGRAM) {
        lap.theta = TH(h); lap.m = MM(h); lap.v = VV(h); lap.sc_in = sc_in; lap.sc_out = sc_out; lap.o = h->opt; lap.loss_slot = loss_slot; lap.gradbuf = h->gradbuf;
        lap.im = h->img; lap.loss_kind = h->net.loss; lap.n_theta = net.n_theta; lap.g_off = net.g_off;
        h->l_apply = &lap;
    }
    int rc = launch_train_kernel(h, sp, idx, first, count, &grid, apply);
    h->l_apply = nullptr;
    if (rc) return rc;
    if (prof && !burst) HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
    if (h->l_applied) {                  // theta, the moments, the loss and the beta products are done
        h->l_applied = false;
        h->sc_sel ^= 1;
        if (prof && burst_last) {
            if (burst) HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
            HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 2], h->stream));
            h->ev_used += 3;
        }
        return EH_OK;
    }
    if (l2) {
        hipLaunchKernelGGL(eh_weight_l2_kernel, dim3(1), dim3(256), 0, h->stream, TH(h), h->img, h->l2val);
        HIPCHK(h, hipGetLastError());
    }
    static const int cw_env = getenv("EH_REDUCE_CW") ? atoi(getenv("EH_REDUCE_CW")) : 0;      // (A/B switch of the measurement tools)
    const bool big = cw_env ? cw_env == 64 : h->n_acc >= 8192;          // enough columns to fill the chip with 64-column blocks
    // layer-wise form: few slab rows (<= 32) under very many columns (the tutorial net: 700 k): one column per thread -- with 64-column
    // blocks 3 of 4 threads had no row to read and the per-block part (counts, barriers) ran 11 k times: 29.6 us of a 250 us step at
    // B = 64 (tools/lform_trace.sh)
    const bool tall = cw_env ? cw_env == 256 : (h->lform && grid <= (int)EH_LFORM_ROWS && h->n_acc >= 16384);
    static const bool nc1 = getenv("EH_REDUCE_NC1") != nullptr;
    const bool tall4 = tall && !nc1;                                   // four columns per thread
    const int rgrid = tall4 ? (h->n_acc + 1023) / 1024 : tall ? (h->n_acc + 255) / 256 : big ? (h->n_acc + 63) / 64 : (h->n_acc + 15) / 16;
#define EH_REDUCE_GO(AP, ...)                                                                                                                       \
    hipLaunchKernelGGL((eh_reduce_kernel<AP, __VA_ARGS__>), dim3(rgrid), dim3(256), 0, h->stream, h->slab, grid, h->n_acc, net.n_theta, net.T, deferred, h->gradbuf, \
                       TH(h), MM(h), VV(h), sc_in, sc_out, h->opt, loss_slot, h->img, h->net.loss, (moment_loss && !raw) ? h->inv_n : nullptr, l2 ? h->l2val : nullptr, tp_mask)
    if (apply) {
        if (tall4) EH_REDUCE_GO(true, 256, 4, true); else if (tall) EH_REDUCE_GO(true, 256); else if (big) EH_REDUCE_GO(true, 64); else EH_REDUCE_GO(true, 16);
        h->sc_sel ^= 1;
    } else {
        if (tall4) EH_REDUCE_GO(false, 256, 4, true); else if (tall) EH_REDUCE_GO(false, 256); else if (big) EH_REDUCE_GO(false, 64); else EH_REDUCE_GO(false, 16);
    }
#undef EH_REDUCE_GO
    HIPCHK(h, hipGetLastError());
    if (prof && burst_last) {
        if (burst) HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 2], h->stream));
        h->ev_used += 3;
    }
    return EH_OK;
}

static int ensure_loss_hist(eh_handle* h, long long need) {
    if (h->loss_cap >= need) return EH_OK;
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->loss_hist);
    h->loss_hist = nullptr; h->loss_cap = 0;
    HIPCHK(h, hipMalloc(&h->loss_hist, (size_t)need * sizeof(float)));
    h->loss_cap = need;
    return EH_OK;
}

static int check_window(eh_handle* h, const EhSplit& sp, long long first, long long count, const char* who) {
    if (!sp.recs || sp.n == 0) return fail(h, EH_ESTATE, "%s: no data set for this split (call eh_set_data)", who);
    if (first < 0 || count < 0 || first + count > sp.n) return fail(h, EH_EINVAL, "%s: window [%lld, %lld) outside 0..%lld", who, first, first + count, sp.n);
    return EH_OK;
}

// host minibatch indices -> the handle's device scratch (range-checked); the step then gathers through it from offset 0
static int stage_host_idx(eh_handle* h, const EhSplit& sp, const int32_t* idx, int64_t first, int64_t count, const char* who) {
    if (!sp.recs) return fail(h, EH_ESTATE, "%s: no data set for this split", who);
    if (count < 0 || first < 0) return fail(h, EH_EINVAL, "%s: first %lld, count %lld", who, (long long)first, (long long)count);
    if (h->capturing) return fail(h, EH_ESTATE, "%s: host indices cannot be recorded into a graph (upload them and pass idx_on_device = 1)", who);
    for (int64_t i = 0; i < count; ++i)
        if (idx[first + i] < 0 || idx[first + i] >= sp.n) return fail(h, EH_EINVAL, "%s: idx[%lld] = %d outside 0..%lld", who, (long long)(first + i), idx[first + i], sp.n);
    if (count > h->idx_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->idx_buf);
        h->idx_buf = nullptr; h->idx_cap = 0;
        HIPCHK(h, hipMalloc(&h->idx_buf, (size_t)std::max<int64_t>(count, 1) * sizeof(int)));
        h->idx_cap = count;
    }
    // (the previous step that read idx_buf is ahead of this copy in stream order)
    HIPCHK(h, hipMemcpyAsync(h->idx_buf, idx + first, (size_t)count * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return EH_OK;
}

// forward / eval on a window; stats (host, double) per target may be null
// The pinned buffers the evaluation kernels store their sums into outlive their handles in a small pool: pinning host memory costs
// some hundred microseconds, and train() creates (and closes) an engine per call -- 10 ms in all for the tutorial's 20 epochs.
static int eval_host_acquire(eh_handle* h, size_t need) {
    {
        std::lock_guard<std::mutex> lk(g_eval_pool_mu);
        for (size_t i = 0; i < g_eval_pool.size(); ++i)
            if (g_eval_pool[i].device == h->device && g_eval_pool[i].cap >= need) {
                h->eval_host = g_eval_pool[i].host; h->eval_host_dev = g_eval_pool[i].dev; h->eval_host_cap = g_eval_pool[i].cap;
                g_eval_pool.erase(g_eval_pool.begin() + (long)i);
                return EH_OK;
            }
    }
    const size_t cap = std::max<size_t>(need, (size_t)4096 * EH_EVAL_STATS * EH_MAX_TARG);
    HIPCHK(h, hipHostMalloc((void**)&h->eval_host, cap * sizeof(float), hipHostMallocMapped));
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->eval_host_dev, h->eval_host, 0));
    h->eval_host_cap = cap;
    return EH_OK;
}
// A large result array to the caller's (pageable) memory: through the pinned staging pair of eh_set_data, chunk k + 1 on its way over
// PCIe while chunk k is copied out by the CPU.  (hipMemcpy straight into pageable memory pins the destination pages behind the
// scenes; with 50 MB of predictions per train() call on the headline data set that cost 1-27 ms per call depending on where the
// allocator had put the arrays -- and whatever the runtime queued to undo it made the NEXT call's first upload chunk wait 15-28 ms
// (tools/e2e_breakdown2.py).)  Small arrays, or a staging pair that is busy or cannot be had: the plain copy.
static int copy_out(eh_handle* h, float* dst, const float* src_dev, size_t n) {
    const size_t bytes = n * sizeof(float);
    std::unique_lock<std::mutex> lk(g_stage_mu, std::try_to_lock);
    if (bytes < ((size_t)256 << 10) || !lk.owns_lock()) { HIPCHK(h, hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost)); return EH_OK; }
    const size_t want = (size_t)8 << 20;                     // 8 MB chunks
    if (g_stage_bytes < want) {
        for (int k = 0; k < 2; ++k) { if (g_stage[k]) (void)hipHostFree(g_stage[k]); g_stage[k] = nullptr; }
        g_stage_bytes = 0;
        if (hipHostMalloc((void**)&g_stage[0], want, hipHostMallocPortable) == hipSuccess && hipHostMalloc((void**)&g_stage[1], want, hipHostMallocPortable) == hipSuccess) g_stage_bytes = want;
        else { (void)hipGetLastError(); for (int k = 0; k < 2; ++k) { if (g_stage[k]) (void)hipHostFree(g_stage[k]); g_stage[k] = nullptr; } }
    }
    if (g_stage_bytes < want) { HIPCHK(h, hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost)); return EH_OK; }
    const size_t CH = std::min(g_stage_bytes, want) / sizeof(float);
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIPCHK(h, hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    { const hipError_t e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming); if (e != hipSuccess) { (void)hipEventDestroy(ev[0]); HIPCHK(h, e); } }
    hipError_t err = hipSuccess;
    const size_t nch = (n + CH - 1) / CH;
    for (size_t k = 0; k <= nch && err == hipSuccess; ++k) {
        if (k < nch) {                                       // chunk k: device -> stage[k & 1]
            const size_t off = k * CH, cnt = std::min(CH, n - off);
            err = hipMemcpyAsync(g_stage[k & 1], src_dev + off, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream);
            if (err == hipSuccess) err = hipEventRecord(ev[k & 1], h->stream);
        }
        if (k >= 1 && err == hipSuccess) {                   // chunk k - 1: stage -> caller
            const size_t off = (k - 1) * CH, cnt = std::min(CH, n - off);
            err = hipEventSynchronize(ev[(k - 1) & 1]);
            if (err == hipSuccess) memcpy(dst + off, g_stage[(k - 1) & 1], cnt * sizeof(float));
        }
    }
    { const hipError_t es = hipStreamSynchronize(h->stream); if (err == hipSuccess) err = es; }      // (the pair is not handed on with a copy in flight)
    (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
    HIPCHK(h, err);
    return EH_OK;
}
// ... and the other way (the parameter vector of a wide model, eh_set_params: 2.8 MB for the tutorial's large net -- a plain hipMemcpy from
// pageable memory was 3 ms of every train() call and 33 ms of one call in eight, tools/e2e_spikes.py)
static int copy_in(eh_handle* h, float* dst_dev, const float* src, size_t n) {
    const size_t bytes = n * sizeof(float);
    std::unique_lock<std::mutex> lk(g_stage_mu, std::try_to_lock);
    const size_t want = (size_t)8 << 20;
    if (bytes < ((size_t)256 << 10) || !lk.owns_lock()) { HIPCHK(h, hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice)); return EH_OK; }
    if (g_stage_bytes < want) {
        for (int k = 0; k < 2; ++k) { if (g_stage[k]) (void)hipHostFree(g_stage[k]); g_stage[k] = nullptr; }
        g_stage_bytes = 0;
        if (hipHostMalloc((void**)&g_stage[0], want, hipHostMallocPortable) == hipSuccess && hipHostMalloc((void**)&g_stage[1], want, hipHostMallocPortable) == hipSuccess) g_stage_bytes = want;
        else { (void)hipGetLastError(); for (int k = 0; k < 2; ++k) { if (g_stage[k]) (void)hipHostFree(g_stage[k]); g_stage[k] = nullptr; } }
    }
    if (g_stage_bytes < want) { HIPCHK(h, hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice)); return EH_OK; }
    const size_t CH = want / sizeof(float);
    hipError_t err = hipSuccess;
    for (size_t off = 0, k = 0; off < n && err == hipSuccess; off += CH, ++k) {
        const size_t cnt = std::min(CH, n - off);
        if (k >= 2) err = hipStreamSynchronize(h->stream);          // (the buffer about to be refilled has been read)
        if (err != hipSuccess) break;
        memcpy(g_stage[k & 1], src + off, cnt * sizeof(float));
        err = hipMemcpyAsync(dst_dev + off, g_stage[k & 1], cnt * sizeof(float), hipMemcpyHostToDevice, h->stream);
    }
    { const hipError_t es = hipStreamSynchronize(h->stream); if (err == hipSuccess) err = es; }
    HIPCHK(h, err);
    return EH_OK;
}

static int do_eval(eh_handle* h, int split, long long first, long long count, double* stats, float* const* yhat, float* const* params) {
    const EhNet& net = h->net;
    EhSplit& sp = h->split[split];
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_window(h, sp, first, count, "eh_eval");
    if (rc) return rc;
    FLUSH(h);
    const long long need = (long long)(yhat ? net.T : 0) * count + (long long)(params ? h->n_par : 0) * count;
    if (need > h->out_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->out_buf);
        h->out_buf = nullptr; h->out_cap = 0;
        HIPCHK(h, hipMalloc(&h->out_buf, (size_t)need * sizeof(float)));
        h->out_cap = need;
    }
    EhStepArgs a{};
    a.prog = h->prog;
    a.recs = sp.recs; a.C = h->C; a.idx = nullptr; a.first = first; a.count = count;
    a.image = h->image; a.slab = h->slab; a.n_acc = EH_EVAL_STATS * net.T;
    a.yhat = yhat ? h->out_buf : nullptr;
    a.pout = params ? h->out_buf + (yhat ? (long long)net.T * count : 0) : nullptr;
    a.yld = count;
    for (int t = 0; t < EH_MAX_TARG; ++t) a.shift[t] = sp.shift[t];
    int grid = count > 0 ? (h->lform ? grid_for(h, count) : eval_grid_for(h, count)) : 1;
    // The per-workgroup sums (grid x 8 T floats) go to the host for the final fold in double.  The step kernels store them straight into
    // pinned, device-visible host memory -- the call is the kernel and one synchronisation; as a copy out of the slab into pageable
    // memory behind the kernel it was a staged transfer of its own, 25-30 us of a 105 us call over the headline split.
    const float* part = nullptr;
    std::vector<float> part_copy;
    static const bool no_zero_copy = getenv("EH_EVAL_COPY") != nullptr;      // (A/B switch of the measurement tools)
    if (!h->lform && !no_zero_copy) {
        const size_t need_host = (size_t)grid * a.n_acc;
        if (need_host > h->eval_host_cap) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            eval_host_release(h);
            if (int rc2 = eval_host_acquire(h, need_host)) return rc2;
        }
        a.slab = h->eval_host_dev;
        part = h->eval_host;
    }
    // (eh_profile_enable: HIP events on the engine's stream around the evaluation kernel too -- the kernel's own time beside the call's,
    //  which also holds the launch, the synchronisation and the reading of the sums; bench.py eh_eval_roofline)
    const bool prof_ev = h->prof && h->ev_used + 3 <= 3 * 8192 && ensure_events(h, h->ev_used + 3) == EH_OK;
    if (prof_ev) HIPCHK(h, hipEventRecord(h->ev[h->ev_used], h->stream));
    if (h->lform) { if ((rc = lform_eval(h, sp, first, count, a.yhat, a.pout, &grid))) return rc; }
    else HIPCHK(h, step_launch(h, EH_MODE_EVAL, grid, &a));
    if (prof_ev) {
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 2], h->stream));
        h->ev_used += 3;
    }
    if (!part) {
        part_copy.resize((size_t)grid * a.n_acc);
        HIPCHK(h, hipMemcpyAsync(part_copy.data(), h->slab, part_copy.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        part = part_copy.data();
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (stats) {
        for (int k = 0; k < a.n_acc; ++k) {
            double s = 0;
            for (int b = 0; b < grid; ++b) s += part[(size_t)b * a.n_acc + k];
            stats[k] = s;
        }
    }
    if (yhat)
        for (int t = 0; t < net.T; ++t)
            if (yhat[t]) { if (int rc2 = copy_out(h, yhat[t], a.yhat + (long long)t * count, (size_t)count)) return rc2; }
    if (params)
        for (int j = 0; j < h->n_par; ++j)
            if (params[j]) { if (int rc2 = copy_out(h, params[j], a.pout + (long long)j * count, (size_t)count)) return rc2; }
    return EH_OK;
}

extern "C" {

int32_t eh_forward(eh_handle* h, int32_t split, int64_t first, int64_t count, float* const* yhat, float* const* params) {
    if (!h) return EH_EINVAL;
    if (split != EH_SPLIT_TRAIN && split != EH_SPLIT_VAL) return fail(h, EH_EINVAL, "eh_forward: split %d", split);
    return do_eval(h, split, first, count, nullptr, yhat, params);
}

int32_t eh_eval(eh_handle* h, int32_t split, int64_t first, int64_t count, eh_target_metrics* out, float* const* yhat, float* const* params) {
    if (!h || !out) return EH_EINVAL;
    if (split != EH_SPLIT_TRAIN && split != EH_SPLIT_VAL) return fail(h, EH_EINVAL, "eh_eval: split %d", split);
    double st[EH_MAX_TARG * EH_EVAL_STATS] = {0};
    int rc = do_eval(h, split, first, count, st, yhat, params);
    if (rc) return rc;
    const double nan = std::nan("");
    for (int t = 0; t < h->net.T; ++t) {
        const double* s = st + t * EH_EVAL_STATS;
        const double c = h->split[split].shift[t];
        const double S = s[0], Sy = s[1], Syy = s[2], n = s[3], Sh = s[4], Shh = s[5], Shy = s[6], A = s[7];
        eh_target_metrics& o = out[t];
        o.n = n; o.sse = S;
        if (n <= 0) { o.mse = o.rmse = o.mae = o.r2 = o.nse = o.pearson = o.kge = o.pbkge = o.beta = o.alpha = nan; continue; }
        const double ssy = Syy - Sy * Sy / n, ssh = Shh - Sh * Sh / n, shy = Shy - Sh * Sy / n;
        o.mse = S / n; o.rmse = std::sqrt(o.mse); o.mae = A / n;
        o.r2 = 1.0 - S / ssy;           // loss_fn(Val(:r2)), src/losses/loss_fn.jl:71-73
        o.nse = o.r2;                   // :nse has the same closed form (:85-86)
        o.pearson = shy / std::sqrt(ssh * ssy);
        o.alpha = std::sqrt(ssh / ssy); // std ratio (the n-1 cancels)
        o.beta = (c + Sh / n) / (c + Sy / n);
        o.kge = 1.0 - std::sqrt((o.pearson - 1) * (o.pearson - 1) + (o.alpha - 1) * (o.alpha - 1) + (o.beta - 1) * (o.beta - 1));
        o.pbkge = 1.0 - std::sqrt((o.pearson - 1) * (o.pearson - 1) + (o.beta - 1) * (o.beta - 1));
    }
    return EH_OK;
}

int32_t eh_mech_loss_vjp(eh_handle* h, int64_t count, int64_t ld, const float* o_dev, const float* const* forcings_dev, const float* const* targets_dev,
                         const int64_t* n_valid_in, float* d_o_dev, float* yhat_dev, float* loss, float* grad_global, int64_t* n_valid) {
    if (!h || !o_dev || !forcings_dev || !targets_dev || !d_o_dev) return EH_EINVAL;
    const EhNet& net = h->net;
    if (net.K == 0) return fail(h, EH_EUNSUPPORTED, "eh_mech_loss_vjp: the model has no neural parameter (no NN outputs to differentiate by): eh_loss_and_grad / eh_train_step run it whole");
    if (net.loss != EH_LOSS_MSE && net.loss != EH_LOSS_MAE) return fail(h, EH_EUNSUPPORTED, "eh_mech_loss_vjp: training loss %d (built: mse, mae)", net.loss);
    if (count < 1 || ld < count) return fail(h, EH_EINVAL, "eh_mech_loss_vjp: count %lld, ld %lld", (long long)count, (long long)ld);
    for (int f = 0; f < net.F; ++f) if (!forcings_dev[f]) return fail(h, EH_EINVAL, "eh_mech_loss_vjp: forcing %d is null", f);
    for (int t = 0; t < net.T; ++t) if (!targets_dev[t]) return fail(h, EH_EINVAL, "eh_mech_loss_vjp: target %d is null", t);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    EhMechArgs a{};
    a.o = o_dev; a.d_o = d_o_dev; a.yhat = yhat_dev; a.n = count; a.ld = ld; a.meta = h->image + h->arch->phi_off;
    bool vec = count % 4 == 0 && ld % 4 == 0 && ((uintptr_t)o_dev | (uintptr_t)d_o_dev | (uintptr_t)yhat_dev) % 16 == 0;
    for (int f = 0; f < net.F; ++f) { a.frc[f] = forcings_dev[f]; vec = vec && (uintptr_t)forcings_dev[f] % 16 == 0; }
    for (int t = 0; t < net.T; ++t) { a.y[t] = targets_dev[t]; vec = vec && (uintptr_t)targets_dev[t] % 16 == 0; }
    if (net.mech == EH_MECH_PROGRAM) { vec = false; a.prog = h->prog; }        // the interpreter keeps its tape in scratch: one sample per lane
    const int per = vec ? 1024 : 256;                                      // samples per workgroup and tile
    if (net.n_out == 1 && net.T > 1) return fail(h, EH_EUNSUPPORTED, "eh_mech_loss_vjp: %d targets on a single-output model", net.T);
    // Many short workgroups, each on its own consecutive tiles: the accesses then move through the arrays as one front, in dispatch
    // order, which is what the memory system serves best (tools/ubench/stream31.hip: 5.8 TB/s for this 3 : 1 mix, against 4.8-5.3 for
    // a persistent grid of 1-4 k workgroups striding through it).  Rows of partials: one per workgroup, <= EH_MECH_MAXROWS.
    const long long ntile = (count + per - 1) / per;
    // (default: two tiles per workgroup; models with several outputs -- twice the streams, more registers per lane, fewer resident waves --
    //  eight: FluxPart 0.59 -> 0.63 of the roof per call, RbQ10 best at two, Expo2Pool indifferent; tools/bench_mech.py --tiles)
    int tiles = h->mech_tiles > 0 ? h->mech_tiles : (net.n_out > 1 ? 8 : 2);
    while ((ntile + tiles - 1) / tiles > EH_MECH_MAXROWS) tiles *= 2;
    int nblk = (int)((ntile + tiles - 1) / tiles);
    if (h->mech_blocks > 0 && nblk > h->mech_blocks) { nblk = h->mech_blocks; tiles = 0; }      // "mech_blocks" option: a capped, grid-striding launch
    a.tiles = tiles;
    a.agg_a = h->img.agg_a;
    const int nfold = nblk <= 2048 ? 0 : std::min(64, (nblk + 511) / 512);      // workgroups of the fold kernel (512 rows and more each)
    const int rows_per = nfold ? (nblk + nfold - 1) / nfold : 0;
    const size_t ws_bytes = EH_MAX_TARG * sizeof(unsigned long long) + 64 + 64 + (size_t)(64 + EH_MECH_MAXROWS) * EH_MECH_PART * sizeof(float);
    if (ws_bytes > h->mech_ws_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->mech_ws);
        h->mech_ws = nullptr; h->mech_ws_bytes = 0;
        HIPCHK(h, hipMalloc(&h->mech_ws, ws_bytes));
        h->mech_ws_bytes = ws_bytes;
    }
    // [counts (4 x u64) | out: loss + 8 gradients, padded to 32 floats | part2 [64][16] | part [rows][16]]
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(h->mech_ws);
    float* out = reinterpret_cast<float*>(h->mech_ws + EH_MAX_TARG * sizeof(unsigned long long));
    float* part2 = out + 32;
    a.part = part2 + 64 * EH_MECH_PART;
    a.counts = counts;
    unsigned long long hc[EH_MAX_TARG] = {0, 0, 0, 0};
    if (n_valid_in) {                        // masks are a property of the data set (src/training/train.jl:221-232): the caller may know the counts
        for (int t = 0; t < net.T; ++t) { if (n_valid_in[t] < 0 || n_valid_in[t] > count) return fail(h, EH_EINVAL, "eh_mech_loss_vjp: n_valid_in[%d] = %lld", t, (long long)n_valid_in[t]); hc[t] = (unsigned long long)n_valid_in[t]; }
        for (int t = 0; t < EH_MAX_TARG; ++t) a.counts_v[t] = hc[t];
        a.use_v = 1;                                             // travels in the kernarg segment: no copy, no counting pass
    } else {
        HIPCHK(h, hipMemsetAsync(counts, 0, sizeof hc, h->stream));
        const unsigned ncb = (unsigned)std::min(nblk, 512);      // (one atomic per workgroup and target on ONE address: they serialise, ~13 ns each)
        if (vec) hipLaunchKernelGGL(eh_count_valid_kernel<4>, dim3(ncb), dim3(256), 0, h->stream, a, net.T, counts);
        else hipLaunchKernelGGL(eh_count_valid_kernel<1>, dim3(ncb), dim3(256), 0, h->stream, a, net.T, counts);
        HIPCHK(h, hipGetLastError());
    }
#define EH_MECH_GO(M)                                                                                                              \
    case M:                                                                                                                      \
        if (vec) hipLaunchKernelGGL((eh_mech_vjp_kernel<4, M>), dim3((unsigned)nblk), dim3(256), 0, h->stream, net, a);           \
        else hipLaunchKernelGGL((eh_mech_vjp_kernel<1, M>), dim3((unsigned)nblk), dim3(256), 0, h->stream, net, a);               \
        break;
    switch (net.mech) {
        EH_MECH_GO(EH_MECH_RBQ10) EH_MECH_GO(EH_MECH_EXPO) EH_MECH_GO(EH_MECH_LINEAR) EH_MECH_GO(EH_MECH_EXPO2POOL)
        EH_MECH_GO(EH_MECH_RS_COMPONENTS) EH_MECH_GO(EH_MECH_RS_COMPONENTS3F) EH_MECH_GO(EH_MECH_FLUXPART)
        case EH_MECH_PROGRAM: hipLaunchKernelGGL((eh_mech_vjp_kernel<1, EH_MECH_PROGRAM>), dim3((unsigned)nblk), dim3(256), 0, h->stream, net, a); break;
        default: return fail(h, EH_EUNSUPPORTED, "eh_mech_loss_vjp: mechanistic model %d", net.mech);
    }
#undef EH_MECH_GO
    HIPCHK(h, hipGetLastError());
    if (nfold) {
        hipLaunchKernelGGL(eh_mech_fold_kernel, dim3((unsigned)nfold), dim3(1024), 0, h->stream, a.part, nblk, rows_per, part2);
        HIPCHK(h, hipGetLastError());
    }
    hipLaunchKernelGGL(eh_mech_finish_kernel, dim3(1), dim3(1024), 0, h->stream, nfold ? part2 : a.part, nfold ? nfold : nblk, net, a.meta, a, out);
    HIPCHK(h, hipGetLastError());
    if (!loss && !grad_global && !n_valid) return EH_OK;         // asynchronous use: results stay on the device, ordered on the handle's stream
    float ho[9];
    HIPCHK(h, hipMemcpyAsync(ho, out, sizeof ho, hipMemcpyDeviceToHost, h->stream));
    if (!a.use_v) HIPCHK(h, hipMemcpyAsync(hc, counts, sizeof hc, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    long long nv = 0;
    for (int t = 0; t < net.T; ++t) nv += (long long)hc[t];
    if (n_valid) *n_valid = nv;
    if (loss) *loss = nv > 0 ? ho[0] : __builtin_nanf("");      // all-masked batch: skipped (epoch.jl:17-19)
    if (grad_global)
        for (int j = 0; j < h->desc.n_params; ++j)
            if (h->desc.param_kind[j] == EH_PAR_GLOBAL) grad_global[h->desc.param_index[j]] = ho[1 + j];
    return EH_OK;
}

int32_t eh_loss_and_grad(eh_handle* h, int32_t split, const int32_t* idx, int64_t first, int64_t count, float* loss, float* grad, int64_t* n_valid) {
    if (!h) return EH_EINVAL;
    if (split != EH_SPLIT_TRAIN && split != EH_SPLIT_VAL) return fail(h, EH_EINVAL, "eh_loss_and_grad: split %d", split);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    EhSplit& sp = h->split[split];
    int rc;
    const int* didx = nullptr;
    if (idx) {
        if ((rc = stage_host_idx(h, sp, idx, 0, count, "eh_loss_and_grad"))) return rc;
        didx = h->idx_buf;
        first = 0;
    } else {
        rc = check_window(h, sp, first, count, "eh_loss_and_grad");
        if (rc) return rc;
    }
    rc = do_step(h, sp, didx, first, count, false, false, nullptr);
    if (rc) return rc;
    std::vector<float> host((size_t)h->n_acc);
    HIPCHK(h, hipMemcpyAsync(host.data(), h->gradbuf, host.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int nt = h->net.n_theta;
    if (grad) memcpy(grad, host.data(), (size_t)nt * sizeof(float));
    if (loss) {
        *loss = host[nt];
        if (h->empty_nan && h->net.T > 1) {
            bool some = false, empty = false;
            for (int t = 0; t < h->net.T; ++t) { some = some || host[nt + 1 + t] > 0.0f; empty = empty || !(host[nt + 1 + t] > 0.0f); }
            if (some && empty) *loss = std::nanf("");
        }
    }
    if (n_valid) {
        double c = 0;
        for (int t = 0; t < h->net.T; ++t) c += host[nt + 1 + t];
        *n_valid = (int64_t)std::llround(c);
    }
    return EH_OK;
}

int32_t eh_get_bn_state(eh_handle* h, float* running_mean, float* running_var, int64_t n) {
    if (!h || !running_mean || !running_var) return EH_EINVAL;
    if (!h->bn_on) return fail(h, EH_ESTATE, "eh_get_bn_state: the model has no input BatchNorm");
    if (n != h->net.P) return fail(h, EH_EINVAL, "eh_get_bn_state: n = %lld, model has %d predictors", (long long)n, h->net.P);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(running_mean, h->bn_run, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(running_var, h->bn_run + 32, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return EH_OK;
}

int32_t eh_set_bn_state(eh_handle* h, const float* running_mean, const float* running_var, int64_t n) {
    if (!h || !running_mean || !running_var) return EH_EINVAL;
    if (!h->bn_on) return fail(h, EH_ESTATE, "eh_set_bn_state: the model has no input BatchNorm");
    if (n != h->net.P) return fail(h, EH_EINVAL, "eh_set_bn_state: n = %lld, model has %d predictors", (long long)n, h->net.P);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<float> rstd((size_t)n);
    for (int64_t p = 0; p < n; ++p) rstd[p] = 1.0f / std::sqrt(running_var[p] + EH_BN_EPS);
    HIPCHK(h, hipMemcpy(h->bn_run, running_mean, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->bn_run + 32, running_var, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->image + h->arch->phi_off + EH_IMG_BNM, running_mean, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->image + h->arch->phi_off + EH_IMG_BNR, rstd.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return EH_OK;
}

int32_t eh_opt_init(eh_handle* h, int32_t rule, float lr, float beta1, float beta2, float eps, float weight_decay) {
    if (!h) return EH_EINVAL;
    if (rule < EH_OPT_ADAM || rule > EH_OPT_DESCENT) return fail(h, EH_EUNSUPPORTED, "eh_opt_init: unknown rule %d", rule);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->opt = EhOpt{rule, lr, beta1, beta2, eps, weight_decay};
    const size_t nt = (size_t)h->net.n_theta;
    HIPCHK(h, hipMemset(MM(h), 0, nt * sizeof(float)));
    HIPCHK(h, hipMemset(VV(h), 0, nt * sizeof(float)));
    const float sc[4] = {beta1, beta2, beta1, beta2};   // Optimisers.jl starts the running product at beta (t = 1)
    HIPCHK(h, hipMemcpy(h->sc, sc, sizeof sc, hipMemcpyHostToDevice));
    h->sc_sel = 0;
    h->opt_ready = true;
    return EH_OK;
}

int32_t eh_get_opt_state(eh_handle* h, float* m, float* v, int64_t n, float* beta_t) {
    if (!h) return EH_EINVAL;
    if (n != h->net.n_theta) return fail(h, EH_EINVAL, "eh_get_opt_state: n");
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (m) { if (int rc = copy_out(h, m, MM(h), (size_t)n)) return rc; }
    if (v) { if (int rc = copy_out(h, v, VV(h), (size_t)n)) return rc; }
    if (beta_t) HIPCHK(h, hipMemcpy(beta_t, h->sc + 2 * h->sc_sel, 2 * sizeof(float), hipMemcpyDeviceToHost));
    return EH_OK;
}

int32_t eh_set_opt_state(eh_handle* h, const float* m, const float* v, int64_t n, const float* beta_t) {
    if (!h) return EH_EINVAL;
    if (!h->opt_ready) return fail(h, EH_ESTATE, "eh_set_opt_state: call eh_opt_init first");
    if (n != h->net.n_theta) return fail(h, EH_EINVAL, "eh_set_opt_state: n");
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (m) { if (int rc = copy_in(h, MM(h), m, (size_t)n)) return rc; }
    if (v) { if (int rc = copy_in(h, VV(h), v, (size_t)n)) return rc; }
    if (beta_t) HIPCHK(h, hipMemcpy(h->sc + 2 * h->sc_sel, beta_t, 2 * sizeof(float), hipMemcpyHostToDevice));
    return EH_OK;
}

int32_t eh_train_step(eh_handle* h, const int32_t* idx, int32_t idx_on_device, int64_t first, int64_t count, float* loss_out) {
    if (!h) return EH_EINVAL;
    if (!h->opt_ready) return fail(h, EH_ESTATE, "eh_train_step: call eh_opt_init first");
    HIPCHK(h, hipSetDevice(h->device));
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    int rc;
    const int* didx = nullptr;
    if (idx && !idx_on_device) {
        if ((rc = stage_host_idx(h, sp, idx, first, count, "eh_train_step"))) return rc;
        didx = h->idx_buf;
        first = 0;
    } else if (idx) {
        if (!sp.recs || sp.n == 0) return fail(h, EH_ESTATE, "eh_train_step: no data set for this split (call eh_set_data)");
        if (first < 0 || count < 0) return fail(h, EH_EINVAL, "eh_train_step: first %lld, count %lld", (long long)first, (long long)count);
        didx = idx;
        if (h->check_idx && count > 0) {        // "check_idx" (debug): the entries idx[first .. first + count) of the caller's DEVICE array against the split's size
            if (h->capturing) return fail(h, EH_ESTATE, "eh_train_step: the check_idx option synchronises: not while a graph is recorded");
            unsigned* bad = nullptr;
            HIPCHK(h, hipMalloc(&bad, 2 * sizeof(unsigned)));
            HIPCHK(h, hipMemsetAsync(bad, 0, sizeof(unsigned), h->stream));
            HIPCHK(h, hipMemsetAsync(bad + 1, 0xFF, sizeof(unsigned), h->stream));      // (position of the first offender: a minimum)
            hipLaunchKernelGGL(eh_idx_check_kernel, dim3((unsigned)std::min<long long>(1024, (count + 255) / 256)), dim3(256), 0, h->stream, idx, (long long)first, (long long)count, (long long)sp.n, bad);
            unsigned res[2] = {0, 0};
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(res, bad, sizeof res, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            (void)hipFree(bad);
            HIPCHK(h, e);
            if (res[0]) return fail(h, EH_EINVAL, "eh_train_step: %u of the %lld device-side indices are outside 0..%lld (first offender: idx[%lld])", res[0], (long long)count, (long long)sp.n - 1, (long long)first + (long long)res[1]);
        }
    } else if ((rc = check_window(h, sp, first, count, "eh_train_step"))) return rc;
    rc = ensure_loss_hist(h, 1);
    if (rc) return rc;
    if (h->fused && !(h->fused_det && grid_for(h, count) != 1)) {
        rc = do_fused_step(h, sp, didx, first, count, loss_out ? h->loss_hist : nullptr);
        if (rc) return rc;
        if (loss_out) FLUSH(h);
    } else {
        FLUSH(h);                              // ("fused_update" 2: a one-workgroup step may be pending in front of this larger one)
        rc = do_step(h, sp, didx, first, count, true, false, loss_out ? h->loss_hist : nullptr);
        if (rc) return rc;
    }
    if (loss_out) {
        HIPCHK(h, hipMemcpyAsync(loss_out, h->loss_hist, sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return EH_OK;
}

// device-side keyed permutation of the train split's sample indices into h->perm
static int make_permutation(eh_handle* h, long long N, uint64_t seed) {
    if (h->perm_cap < N) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->perm);
        h->perm = nullptr; h->perm_cap = 0;
        HIPCHK(h, hipMalloc(&h->perm, (size_t)N * sizeof(int)));
        h->perm_cap = N;
    }
    int bits = 1;
    while ((1LL << bits) < N) ++bits;
    const int hb = std::max(1, (bits + 1) / 2);
    hipLaunchKernelGGL(eh_perm_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->perm, (uint32_t)N, hb, seed);
    HIPCHK(h, hipGetLastError());
    return EH_OK;
}

// ---- hipGraph capture of a sequence of training steps -------------------------------------------------
int32_t eh_graph_begin(eh_handle* h) {
    if (!h) return EH_EINVAL;
    if (h->capturing) return fail(h, EH_ESTATE, "eh_graph_begin: already capturing");
    // a fused-mode step applies the update of the step before it: the recorded sequence has to start (and every replay
    // has to find the engine) with such an update pending, or its first kernel would skip / re-apply one
    if (h->fused && !h->pending) return fail(h, EH_ESTATE, "eh_graph_begin: fused_update mode: run one training step first (and do not synchronize before capturing)");
    h->cap = {nullptr, h->fused, (int)(h->gstep % 3), h->cur, h->sc_sel};
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_loss_hist(h, 1);
    if (rc) return rc;
    if (jit_wanted(h, EH_MODE_TRAIN)) {
        // compile and load NOW, synchronously: hiprtc / hipModuleLoadData must not run beside a relaxed capture, and the kernel must
        // not change in the middle of the recorded sequence ("specialize" = 2 would otherwise hand the build to a worker thread)
        h->capturing = true;                    // (jit_entry builds in the calling thread while this is set)
        (void)jit_entry(h);
        h->capturing = false;
        for (auto& e : h->jit) if (e->worker.joinable()) e->worker.join();      // a build already in flight: wait for it
    }
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
    h->capturing = true;
    return EH_OK;
}

int32_t eh_graph_end(eh_handle* h, int32_t* graph_id) {
    if (!h || !graph_id) return EH_EINVAL;
    if (!h->capturing) return fail(h, EH_ESTATE, "eh_graph_end: not capturing");
    h->capturing = false;
    hipGraph_t g = nullptr;
    HIPCHK(h, hipStreamEndCapture(h->stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(h, EH_EHIP, "eh_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    if (h->fused != h->cap.fused || (int)(h->gstep % 3) != h->cap.gslot || h->cur != h->cap.cur || h->sc_sel != h->cap.sc_sel) {
        (void)hipGraphExecDestroy(ex);
        return fail(h, EH_EINVAL, "eh_graph_end: the recorded sequence does not bring the engine's rotation state back (record a multiple of 6 steps in fused_update mode, of 2 otherwise)");
    }
    h->cap.exec = ex;
    h->graphs.push_back(h->cap);
    *graph_id = (int32_t)h->graphs.size() - 1;
    return EH_OK;
}

int32_t eh_graph_launch(eh_handle* h, int32_t graph_id) {
    if (!h) return EH_EINVAL;
    if (graph_id < 0 || graph_id >= (int32_t)h->graphs.size()) return fail(h, EH_EINVAL, "eh_graph_launch: graph %d", graph_id);
    const eh_handle::GraphRec& g = h->graphs[(size_t)graph_id];
    if (g.fused != h->fused || g.gslot != (int)(h->gstep % 3) || g.cur != h->cur || g.sc_sel != h->sc_sel || (g.fused && !h->pending))
        return fail(h, EH_ESTATE, "eh_graph_launch: the engine is not in the state the graph was recorded in (steps / synchronize in between: run steps until it is, with an update pending in fused_update mode)");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipGraphLaunch(g.exec, h->stream));
    return EH_OK;
}

int32_t eh_train_epoch(eh_handle* h, int64_t batchsize, uint64_t seed, int32_t shuffle, float* mean_loss, int64_t* n_steps) {
    if (!h) return EH_EINVAL;
    if (!h->opt_ready) return fail(h, EH_ESTATE, "eh_train_epoch: call eh_opt_init first");
    if (batchsize < 1) return fail(h, EH_EINVAL, "eh_train_epoch: batchsize %lld", (long long)batchsize);
    HIPCHK(h, hipSetDevice(h->device));
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    if (!sp.recs || sp.n == 0) return fail(h, EH_ESTATE, "eh_train_epoch: no training data");
    const long long N = sp.n;
    if (shuffle) {
        if (int rc = make_permutation(h, N, seed)) return rc;
        h->perm_valid = false;          // (the data-parallel window order of eh_dp_shuffle is gone)
    }
    const long long steps = (N + batchsize - 1) / batchsize;
    int rc = ensure_loss_hist(h, steps);
    if (rc) return rc;
    if (multi_ok(h, batchsize)) {          // one workgroup per step: up to EH_MULTI_MAX steps per launch
        for (long long s = 0; s < steps; s += EH_MULTI_MAX) {
            const int n = (int)std::min<long long>(EH_MULTI_MAX, steps - s);
            if ((rc = do_fused_multi(h, sp, shuffle ? h->perm : nullptr, s * batchsize, batchsize, N, n, h->loss_hist + s))) return rc;
        }
    } else
    for (long long s = 0; s < steps; ++s) {
        const long long first = s * batchsize, count = std::min<long long>(batchsize, N - first);
        const bool one_kernel = h->fused && !(h->fused_det && grid_for(h, count) != 1);      // ("fused_update" 2: only where one workgroup covers the minibatch)
        if (!one_kernel) FLUSH(h);
        rc = one_kernel ? do_fused_step(h, sp, shuffle ? h->perm : nullptr, first, count, h->loss_hist + s)
                        : do_step(h, sp, shuffle ? h->perm : nullptr, first, count, true, false, h->loss_hist + s);
        if (rc) return rc;
    }
    if (mean_loss) FLUSH(h);
    if (n_steps) *n_steps = steps;
    if (mean_loss) {
        std::vector<float> l((size_t)steps);
        HIPCHK(h, hipMemcpyAsync(l.data(), h->loss_hist, l.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        double sum = 0; long long c = 0;
        for (float x : l) if (!std::isnan(x)) { sum += x; ++c; }
        *mean_loss = c ? (float)(sum / c) : std::nanf("");
    }
    return EH_OK;
}

int32_t eh_set_weight_l2(eh_handle* h, float lambda, int32_t normalize) {
    if (!h) return EH_EINVAL;
    if (!(lambda >= 0.0f)) return fail(h, EH_EINVAL, "eh_set_weight_l2: lambda = %g", (double)lambda);
    if (lambda != 0.0f && h->fused) return fail(h, EH_EUNSUPPORTED, "eh_set_weight_l2: not built for the fused_update mode: switch it off first");
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    h->img.l2c = (normalize && h->n_weights > 0) ? lambda / (float)h->n_weights : lambda;
    h->img.l2w = nullptr;
    return EH_OK;
}

int32_t eh_set_weight_l2_coef(eh_handle* h, const float* coef, int64_t n) {
    if (!h) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    if (!coef || n == 0) { h->img.l2w = nullptr; h->img.l2c = 0.0f; return EH_OK; }
    if (n != h->net.n_theta) return fail(h, EH_EINVAL, "eh_set_weight_l2_coef: %lld coefficients for %d parameters", (long long)n, h->net.n_theta);
    bool any = false;
    for (int64_t i = 0; i < n; ++i) {
        if (!(coef[i] >= 0.0f) || std::isinf(coef[i])) return fail(h, EH_EINVAL, "eh_set_weight_l2_coef: coefficient %lld = %g", (long long)i, (double)coef[i]);
        any = any || coef[i] != 0.0f;
    }
    if (any && h->fused) return fail(h, EH_EUNSUPPORTED, "eh_set_weight_l2_coef: not built for the fused_update mode: switch it off first");
    if (!any) { h->img.l2w = nullptr; h->img.l2c = 0.0f; return EH_OK; }
    if (!h->l2w) HIPCHK(h, hipMalloc(&h->l2w, (size_t)n * sizeof(float)));
    HIPCHK(h, hipMemcpyAsync(h->l2w, coef, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));          // (the caller's array may go away)
    h->img.l2w = h->l2w; h->img.l2c = 0.0f; h->img.n_theta = h->net.n_theta;
    return EH_OK;
}

int32_t eh_dp_shuffle(eh_handle* h, uint64_t seed, int32_t on) {
    if (!h) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (!on) { h->perm_valid = false; return EH_OK; }      // (stream order keeps earlier steps on the old permutation)
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    if (!sp.recs || sp.n == 0) return fail(h, EH_ESTATE, "eh_dp_shuffle: no training data");
    if (int rc = make_permutation(h, sp.n, seed)) return rc;
    h->perm_valid = true;
    return EH_OK;
}

int32_t eh_dp_grad(eh_handle* h, int64_t first, int64_t count) {
    if (!h) return EH_EINVAL;
    if (h->net.T != 1 && !h->tcount_ready) return fail(h, EH_ESTATE, "eh_dp_grad: multi-target model: call eh_dp_counts for this window and all-reduce EH_BUF_TCOUNT first");
    const unsigned tpm_dp = two_pass_mask(h->net);
    if (tpm_dp && h->mom_stage != 2) return fail(h, EH_ESTATE, "eh_dp_grad: rmse (multi-target) / pearson / kge training losses need the moments of the GLOBAL batch's predictions first: eh_dp_moments stage 0, all-reduce EH_BUF_MOMENT, stage 1, all-reduce");
    if (h->bn_on && !h->bn_ext) return fail(h, EH_ESTATE, "eh_dp_grad: input BatchNorm needs the global batch statistics: call eh_dp_bn_stats and all-reduce EH_BUF_BNSTAT first");
    h->bn_dp_update = h->bn_on;
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    int rc = check_window(h, sp, first, count, "eh_dp_grad");
    if (rc) return rc;
    if (h->net.T != 1) {      // the weights of the GLOBAL batch from the all-reduced sums; the step kernel then normalises exactly
        hipLaunchKernelGGL(eh_weights_from_counts_kernel, dim3(1), dim3(64), 0, h->stream, h->tcount, h->net.T, h->net.loss_t, h->inv_n, h->img.agg_a, h->roles, h->img.l2s);
        HIPCHK(h, hipGetLastError());
        h->dp_weights = true;
    }
    if (tpm_dp) {             // the all-reduced moments about the global centre -> k0 k1 k2 and the loss value of every two-pass target
        EhShift4 s4; for (int t = 0; t < EH_MAX_TARG; ++t) s4.c[t] = sp.shift[t];
        hipLaunchKernelGGL(eh_moment_coef_kernel, dim3(h->net.T), dim3(64), 0, h->stream, h->mombuf, 1, h->net.T, h->net.loss_t, s4, h->inv_n, h->img.agg_a);
        HIPCHK(h, hipGetLastError());
        h->dp_moments = true;
    }
    rc = do_step(h, sp, h->perm_valid ? h->perm : nullptr, first, count, false, true, nullptr);
    h->dp_weights = false; h->tcount_ready = false; h->dp_moments = false; h->mom_stage = 0;
    return rc;
}

// multi-target models under data parallelism: this shard's per-target sums of the window into EH_BUF_TCOUNT
// ([n_t | sum (y - c) | sum (y - c)^2] per target, c = the split's target shift: the ranks must share it, eh_set_target_shift).
// The caller all-reduces the 12 floats; eh_dp_grad turns them into the weights 1 / n_t (mse, mae) or 1 / sum (y - ybar)^2 (nseLoss).
int32_t eh_dp_counts(eh_handle* h, int64_t first, int64_t count) {
    if (!h) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    int rc = check_window(h, sp, first, count, "eh_dp_counts");
    if (rc) return rc;
    const EhNet& net = h->net;
    EhShift4 sh4; for (int t = 0; t < EH_MAX_TARG; ++t) sh4.c[t] = sp.shift[t];
    HIPCHK(h, hipMemsetAsync(h->tcount, 0, 3 * EH_MAX_TARG * sizeof(float), h->stream));
    hipLaunchKernelGGL(eh_count_kernel, dim3(net.T), dim3(256), 0, h->stream, sp.recs, h->C, net.P + net.F, net.T, h->perm_valid ? h->perm : nullptr, first, count,
                       h->inv_n, net.loss_t, sh4, h->img.agg_a, h->roles, h->img.l2s, h->tcount);
    HIPCHK(h, hipGetLastError());
    h->tcount_ready = true;
    return EH_OK;
}

// Two-pass training losses (pearsonLoss / kgeLoss / pbkgeLoss, rmse on a multi-target model; src/losses/loss_fn.jl:58-60,105-174)
// under data parallelism: d loss / d yhat_i = k0 + k1 (yhat_i - centre) + k2 (y_i - c) with coefficients made of the moments of the
// GLOBAL batch, so the shards exchange their moment sums twice ahead of the pass --
//   stage 0: forward-only pass over the window, this shard's sums about the common shift c into EH_BUF_MOMENT
//            ([T][EH_EVAL_STATS] floats: S, sum (y-c), sum (y-c)^2, n, sum (yhat-c), ...)          (caller: all-reduce)
//   stage 1: centre of yhat = c + sum (yhat - c) / n of the global batch; forward-only pass about it, sums into EH_BUF_MOMENT (all-reduce)
// -- and eh_dp_grad turns the all-reduced moments into the coefficients (same kernel as the single-GPU step) and runs the training
// pass with them: the sums in EH_BUF_GRAD are final, as for multi-target models.  Fused kernel families only (the layer-wise form
// keeps its statistics passes inside its own forward).
int32_t eh_dp_moments(eh_handle* h, int64_t first, int64_t count, int32_t stage) {
    if (!h) return EH_EINVAL;
    const EhNet& net = h->net;
    if (!two_pass_mask(net)) return fail(h, EH_ESTATE, "eh_dp_moments: the training loss needs no batch moments of the predictions");
    if (h->lform) return fail(h, EH_EUNSUPPORTED, "eh_dp_moments: the layer-wise form has no data-parallel seam for the two-pass training losses");
    if (stage != 0 && stage != 1) return fail(h, EH_EINVAL, "eh_dp_moments: stage %d (0 or 1)", stage);
    if (stage == 1 && h->mom_stage != 1) return fail(h, EH_ESTATE, "eh_dp_moments: stage 1 follows stage 0 (and the all-reduce of EH_BUF_MOMENT)");
    if (h->bn_on && !h->bn_ext) return fail(h, EH_ESTATE, "eh_dp_moments: input BatchNorm needs the global batch statistics: call eh_dp_bn_stats and all-reduce EH_BUF_BNSTAT first");
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    if (int rc = check_window(h, sp, first, count, "eh_dp_moments")) return rc;
    const int* idx = h->perm_valid ? h->perm : nullptr;
    EhStepArgs e{};
    e.prog = h->prog;
    e.recs = sp.recs; e.C = h->C; e.idx = idx; e.first = first; e.count = count;
    e.image = h->image; e.slab = h->slab; e.n_acc = EH_EVAL_STATS * net.T; e.rmap = h->rmap; e.stamps = nullptr;
    e.yld = count;
    for (int t = 0; t < EH_MAX_TARG; ++t) e.shift[t] = sp.shift[t];
    // (the global BatchNorm statistics of eh_dp_bn_stats serve both moment passes AND the training pass behind them: eh_dp_grad /
    //  eh_dp_fused_step consume them; advisor, round 4: stage 0 used to clear them and every later pass of the step failed)
    if (int rc = bn_prepare(h, sp, idx, first, count, false, &e, /*consume*/ false)) return rc;
    EhShift4 s4; for (int t = 0; t < EH_MAX_TARG; ++t) s4.c[t] = sp.shift[t];
    if (stage == 1) {        // the all-reduced sums about the shift -> the centre of yhat of the global batch
        hipLaunchKernelGGL(eh_moment_centre_kernel, dim3(net.T), dim3(64), 0, h->stream, h->mombuf, 1, net.T, net.loss_t, s4, h->inv_n);
        HIPCHK(h, hipGetLastError());
        e.inv_n = h->inv_n;
    }
    const int egrid = count > 0 ? grid_for(h, count) : 1;
    if (count > 0) HIPCHK(h, step_launch(h, EH_MODE_EVAL, egrid, &e));
    else HIPCHK(h, hipMemsetAsync(h->slab, 0, (size_t)EH_EVAL_STATS * net.T * sizeof(float), h->stream));      // an empty shard window adds nothing
    hipLaunchKernelGGL(eh_moment_fold_kernel, dim3(1), dim3(64), 0, h->stream, h->slab, egrid, net.T, h->mombuf);
    HIPCHK(h, hipGetLastError());
    h->mom_stage = stage + 1;
    return EH_OK;
}

// common shift of the shifted target sums (nseLoss, metrics) -- under data parallelism every rank passes the same vector, e.g. the
// mean of each target over the global training set; eh_set_data resets it to the shard's own means
int32_t eh_set_target_shift(eh_handle* h, int32_t split, const float* shift, int64_t n) {
    if (!h || !shift) return EH_EINVAL;
    if (split != EH_SPLIT_TRAIN && split != EH_SPLIT_VAL) return fail(h, EH_EINVAL, "eh_set_target_shift: split %d", split);
    if (n != h->net.T) return fail(h, EH_EINVAL, "eh_set_target_shift: %lld values for %d targets", (long long)n, h->net.T);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    for (int t = 0; t < h->net.T; ++t) h->split[split].shift[t] = shift[t];
    return EH_OK;
}

int32_t eh_dp_fused_step(eh_handle* h, int64_t first, int64_t count, int32_t* buffer_index) {
    if (!h || !buffer_index) return EH_EINVAL;
    if (!h->fused) return fail(h, EH_ESTATE, "eh_dp_fused_step: set the fused_update option first");
    if (h->net.T != 1) return fail(h, EH_EUNSUPPORTED, "eh_dp_fused_step: multi-target models need the global per-target counts before the pass: use eh_dp_counts + eh_dp_grad (fused_update off)");
    if (h->bn_on && !h->bn_ext) return fail(h, EH_ESTATE, "eh_dp_fused_step: input BatchNorm needs the global batch statistics: call eh_dp_bn_stats and all-reduce EH_BUF_BNSTAT first");
    if (!h->opt_ready) return fail(h, EH_ESTATE, "eh_dp_fused_step: call eh_opt_init first");
    HIPCHK(h, hipSetDevice(h->device));
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    int rc = check_window(h, sp, first, count, "eh_dp_fused_step");
    if (rc) return rc;
    *buffer_index = h->p2p_on ? -1 : (int32_t)(h->gstep % 3);        // -1: the kernels exchange the sums themselves (eh_p2p_attach)
    return do_fused_step(h, sp, h->perm_valid ? h->perm : nullptr, first, count, nullptr);
}

int32_t eh_set_bn_shift(eh_handle* h, const float* shift, int64_t n) {
    if (!h || !shift) return EH_EINVAL;
    if (!h->bn_on) return fail(h, EH_ESTATE, "eh_set_bn_shift: the model has no input BatchNorm");
    if (n != h->net.P) return fail(h, EH_EINVAL, "eh_set_bn_shift: n = %lld, model has %d predictors", (long long)n, h->net.P);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->bn_shift, shift, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return EH_OK;
}

int32_t eh_dp_bn_stats(eh_handle* h, int64_t first, int64_t count) {
    if (!h) return EH_EINVAL;
    if (!h->bn_on) return fail(h, EH_ESTATE, "eh_dp_bn_stats: the model has no input BatchNorm");
    HIPCHK(h, hipSetDevice(h->device));
    EhSplit& sp = h->split[EH_SPLIT_TRAIN];
    int rc = check_window(h, sp, first, count, "eh_dp_bn_stats");
    if (rc) return rc;
    const int* idx = h->perm_valid ? h->perm : nullptr;      // the same samples eh_dp_grad / eh_dp_fused_step will read
    const int nblk = (int)std::max<long long>(1, std::min<long long>(32, (count + 1023) / 1024));
    hipLaunchKernelGGL(eh_bn_stats_kernel, dim3(nblk), dim3(1024), 0, h->stream, sp.recs, h->C, h->net.P, idx, (int)first, (int)count, h->bn_part, h->bn_shift);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(eh_bn_fold_kernel, dim3(1), dim3(64), 0, h->stream, h->bn_part, nblk, (int)count, h->bn_stat);
    HIPCHK(h, hipGetLastError());
    h->bn_ext = true;
    return EH_OK;
}

int32_t eh_dp_apply(eh_handle* h, float* loss_out) {
    if (!h) return EH_EINVAL;
    if (!h->opt_ready) return fail(h, EH_ESTATE, "eh_dp_apply: call eh_opt_init first");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_loss_hist(h, 1);
    if (rc) return rc;
    float* sc_in = h->sc + 2 * h->sc_sel;
    float* sc_out = h->sc + 2 * (h->sc_sel ^ 1);
    const int nt = h->net.n_theta;
    const bool l2 = h->img.l2c != 0.0f || h->img.l2w;
    if (l2) {                                                  // the extra loss is a function of the replicated parameters: every rank adds the same term
        hipLaunchKernelGGL(eh_weight_l2_kernel, dim3(1), dim3(256), 0, h->stream, TH(h), h->img, h->l2val);
        HIPCHK(h, hipGetLastError());
    }
    hipLaunchKernelGGL(eh_apply_kernel, dim3((nt + 255) / 256), dim3(256), 0, h->stream, h->gradbuf, nt, TH(h), MM(h), VV(h), sc_in, sc_out, h->opt,
                       h->loss_hist, h->img, h->net.loss, h->net.T, l2 ? h->l2val : nullptr, two_pass_mask(h->net) ? h->inv_n : nullptr, two_pass_mask(h->net));
    HIPCHK(h, hipGetLastError());
    h->sc_sel ^= 1;
    if (loss_out) {
        HIPCHK(h, hipMemcpyAsync(loss_out, h->loss_hist, sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return EH_OK;
}

int32_t eh_device_buffer(eh_handle* h, int32_t which, void** dev_ptr, int64_t* n_floats) {
    if (!h || !dev_ptr || !n_floats) return EH_EINVAL;
    if (h->fused && which != EH_BUF_GRAD && which != EH_BUF_GACC && which != EH_BUF_BNSTAT && which != EH_BUF_TCOUNT && which != EH_BUF_MOMENT) return fail(h, EH_ESTATE, "eh_device_buffer: parameter buffers ping-pong in fused_update mode; switch it off first");
    switch (which) {
        case EH_BUF_GRAD: *dev_ptr = h->gradbuf; *n_floats = h->n_acc; return EH_OK;
        case EH_BUF_THETA: *dev_ptr = TH(h); *n_floats = h->net.n_theta; return EH_OK;
        case EH_BUF_OPT_M: *dev_ptr = MM(h); *n_floats = h->net.n_theta; return EH_OK;
        case EH_BUF_OPT_V: *dev_ptr = VV(h); *n_floats = h->net.n_theta; return EH_OK;
        case EH_BUF_GACC: *dev_ptr = h->gacc; *n_floats = (int64_t)3 * EH_GSHARDS * h->n_acc; return EH_OK;
        case EH_BUF_BNSTAT:
            if (!h->bn_on) return fail(h, EH_ESTATE, "eh_device_buffer: the model has no input BatchNorm");
            *dev_ptr = h->bn_stat; *n_floats = 65; return EH_OK;
        case EH_BUF_TCOUNT: *dev_ptr = h->tcount; *n_floats = 3 * EH_MAX_TARG; return EH_OK;
        case EH_BUF_MOMENT: *dev_ptr = h->mombuf; *n_floats = EH_MAX_TARG * EH_EVAL_STATS; return EH_OK;
        default: return fail(h, EH_EINVAL, "eh_device_buffer: which = %d", which);
    }
}

int32_t eh_debug_stamps(eh_handle* h, uint64_t* out, int32_t n) {
    if (!h || !out || n < 0 || n > 32) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->stamps) {   // first call arms the buffer; later calls read it
        HIPCHK(h, hipMalloc(&h->stamps, 32 * sizeof(unsigned long long)));
        HIPCHK(h, hipMemset(h->stamps, 0, 32 * sizeof(unsigned long long)));
        memset(out, 0, (size_t)n * sizeof(uint64_t));
        return EH_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->stamps, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return EH_OK;
}

int32_t eh_profile_enable(eh_handle* h, int32_t on) {
    if (!h) return EH_EINVAL;
    h->prof = on != 0;
    h->prof_stride = on > 1 ? on : 1;        // on = S > 1: one event pair around every burst of S consecutive steps
    h->prof_k = 0;
    h->ev_used = 0;
    return EH_OK;
}

int32_t eh_profile_samples(eh_handle* h, double* ms, int64_t cap, int64_t* n_out) {
    if (!h || !n_out || (cap > 0 && !ms)) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int64_t n = std::min<int64_t>((int64_t)(h->ev_used / 3), cap);
    for (int64_t i = 0; i < n; ++i) {
        float t = 0;
        HIPCHK(h, hipEventElapsedTime(&t, h->ev[3 * i], h->ev[3 * i + 2]));
        ms[i] = t;
    }
    *n_out = n;
    return EH_OK;
}

int32_t eh_profile_read(eh_handle* h, int64_t* n_launches, double* mean_ms_step_kernel, double* mean_ms_reduce_kernel) {
    if (!h) return EH_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double a = 0, b = 0;
    const size_t n = h->ev_used / 3;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[3 * i], h->ev[3 * i + 1]));
        a += ms;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[3 * i + 1], h->ev[3 * i + 2]));
        b += ms;
    }
    if (n_launches) *n_launches = (int64_t)n;
    if (mean_ms_step_kernel) *mean_ms_step_kernel = n ? a / n : 0.0;
    if (mean_ms_reduce_kernel) *mean_ms_reduce_kernel = n ? b / n : 0.0;
    h->ev_used = 0;
    return EH_OK;
}

}   // extern "C"

