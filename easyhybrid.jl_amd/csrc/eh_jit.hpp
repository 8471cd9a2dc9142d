// Run-time compilation of the step kernels for a recorded closure (EH_MECH_PROGRAM).  The kernels built ahead of time
// interpret the program (per-lane value array in scratch memory, eh_device.hpp `eh_prog_forward`); here the same program is
// written out as straight-line C++ and the kernel template is instantiated around it with hiprtc, so the tape lives in
// registers and the mechanistic stage costs what a hand-derived registry model costs.  Falls back to the interpreter when
// hiprtc is unavailable or refuses (reported by eh_jit_status).
#pragma once
#include <string>
#include <vector>

#include "eh_arch.hpp"

struct EhJitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn[4] = {nullptr, nullptr, nullptr, nullptr};   // EH_MODE_TRAIN, EH_MODE_EVAL, EH_MODE_TRAIN_P2P (when asked for), EH_MODE_TRAIN_MULTI (per-wave registry models with one target)
    int nw = 0;
    size_t lds_bytes = 0;
    size_t lds_eval_bytes = 0;      // forward / evaluation kernel (0: lds_bytes)
};

// a recorded custom training loss (eh_set_loss_program): value slot 0 = yhat, 1 = y
struct EhLossProg1 {
    std::vector<unsigned> code;
    std::vector<float> consts;
    int out = 0;
};
struct EhLossProg : EhLossProg1 {          // the program every target without one of its own uses (eh_set_loss_program) ...
    EhLossProg1 per[4];                    // ... and the targets' own (eh_set_target_loss_program; PerTarget((f, g)), compute_loss.jl:128-145)
    int gen = 0;          // bumped by every eh_set_*loss_program: part of the cache key of the compiled kernels
    const EhLossProg1& of(int t) const { return per[t].code.empty() ? static_cast<const EhLossProg1&>(*this) : per[t]; }
    bool has(int t) const { return !of(t).code.empty(); }
};
// the generated eh_jit_loss.inc: float eh_jit_loss(int target, float yhat, float yobs, float& dl)
std::string eh_jit_loss_source(const EhLossProg& lp);
// the generated eh_jit_mech.inc (EhJitTape, eh_jit_fwd, eh_jit_rev) for a validated descriptor
std::string eh_jit_mech_source(const eh_model_desc& d);
// Compiles the train + eval (+ cross-GPU train) kernels of (arch, variant, activation, fast-path flags); false + log on failure.
// `spec` (optional) bakes the model descriptor into the kernels as a compile-time constant.
// allow_slp = false: never the SLP vectoriser (the flags of the kernels built ahead of time: a kernel that REPLACES one of those in the
// middle of a run -- "specialize" = 2 -- must give the same bits)
bool eh_jit_build(const eh_model_desc& d, const EhArchInfo* A, int variant, int act, int fast, const EhNet* spec, bool with_p2p,
                  const EhLossProg* loss, EhJitKernel* out, std::string* log, bool allow_slp = true);
hipError_t eh_jit_launch(const EhJitKernel* k, int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args);
void eh_jit_release(EhJitKernel* k);
