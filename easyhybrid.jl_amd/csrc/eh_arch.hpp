// Table of compiled kernel shapes.  One translation unit (eh_arch.hip, built once per
// (EH_NBI, EH_NBH, EH_NL) by the Makefile) instantiates eh_step_kernel for one padded MLP shape:
//   NBI  input  blocks of 16  (P <= 16*NBI)
//   NBH  hidden blocks of 16  (every hidden width <= 16*NBH)
//   NL   hidden layers
// for every activation and both modes.  A shape has one or more (NT, NW) variants: NT x16 samples per
// wave macro-tile, NW waves per workgroup; variant 0 is the default (largest NT in {4,2,1} whose LDS
// image fits the 160 KiB of a gfx950 CU with a 4-wave workgroup).
#pragma once
#include "eh_device.hpp"

struct EhVariant {
    int nt, nw;
    size_t lds_bytes;        // dynamic LDS per workgroup
    int red_floats;          // floats available to the end-of-kernel reduction (needs nw * n_acc)
    hipError_t (*prepare)(void);   // raises the dynamic-LDS limit of every kernel of the variant
    // fast: bit 0 = single NN output (K == 1), bit 1 = P <= 4; only honoured by shapes built with EH_FAST_PATHS
    hipError_t (*launch)(int mode, int act, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args);
    int tiles;               // macro-tiles a workgroup works on at a time; 0 = nw (one per wave)
    int bf16;                // eh_widebf_kernel (eh_wide_bf16.hpp), the "precision" option: 1 = bf16 forward products, fp32-exact backward (three-term deltas);
                             // 2 = bf16 operands in both passes (deltas rounded once); 0 = the fp32 kernels
    int so;                  // != 0: the TRAIN kernel is the sample-owned eh_bfs_kernel (eh_bf16_sample.hpp; one-network models only, 16 * nw-sample
                             // tiles: nt == nw); evaluation passes run eh_widebf_kernel with so x 16-sample tiles.  What "precision" selects.
    size_t lds_eval;         // != 0: dynamic LDS of the variant's forward / evaluation kernels (per-wave family: no hidden images); 0 = lds_bytes
};

struct EhArchInfo {
    int nbi, nbh, nl;
    // parameter-image geometry (EhGeom; independent of the variant)
    int ip, hp, s0, sh, w0_off, wh_off, wo_off, b_off, phi_off, img_floats;
    int has_fast;            // K1 / small-P kernels compiled for this shape
    int nvar;
    EhVariant var[8];
    int wide;                // eh_wide_kernel (eh_wide.hpp): the four waves of a workgroup share one tile and split the layers by rows
};

// A step kernel with ONE model descriptor baked in at build time (eh_spec.hip, one translation unit per canonical descriptor: the
// BASELINE.json configurations): what the "specialize" option compiles at run time, without a run-time compiler.  A handle whose
// descriptor, kernel family, variant, activation and fast-path flags match runs it instead of the generic kernel.
struct EhSpecKernel {
    EhNet net;
    int wide, bf16, nbi, nbh, nl, nt, nw, act, fast;
    int so;                  // the TRAIN kernel is the sample-owned one (EhVariant::so; family 3 of eh_spec.hip): part of the match, the (nt, nw, bf16) signature alone does not tell the families apart
    size_t lds_bytes;
    const char* what;
    hipError_t (*prepare)(void);
    hipError_t (*launch)(int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args);
};
#define EH_SPEC_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#define EH_SPEC_DECL(k) extern "C" const EhSpecKernel* eh_spec_##k(void);
EH_SPEC_LIST(EH_SPEC_DECL)
#undef EH_SPEC_DECL

constexpr size_t EH_LDS_LIMIT = 160 * 1024;

template <int NBI, int NBH, int NL>
constexpr int eh_pick_nt() {
    if (sizeof(float) * EhGeom<NBI, NBH, NL, 4, 4>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 4;
    if (sizeof(float) * EhGeom<NBI, NBH, NL, 2, 4>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 2;
    return 1;
}

#define EH_ARCH_LIST(X) \
    X(1, 1, 1) X(1, 1, 2) X(1, 1, 3) \
    X(1, 2, 1) X(1, 2, 2) X(1, 2, 3) \
    X(1, 4, 1) X(1, 4, 2) X(1, 4, 3) \
    X(2, 1, 1) X(2, 1, 2) X(2, 1, 3) \
    X(2, 2, 1) X(2, 2, 2) X(2, 2, 3) \
    X(2, 4, 1) X(2, 4, 2) X(2, 4, 3)

// Row-split kernel (eh_wide.hpp).  Hidden widths 65..128: the only kernel, at most two hidden layers
// fit the LDS.  Widths 33..64: the fallback when the per-wave reduction space of the kernel above is
// too small for the model's gradient (P > 16 with full-width layers), selectable with the
// "row_split" option otherwise.
#define EH_WIDE_LIST(X) \
    X(1, 4, 1) X(1, 4, 2) X(1, 4, 3) X(2, 4, 1) X(2, 4, 2) X(2, 4, 3) \
    X(1, 8, 1) X(1, 8, 2) X(2, 8, 1) X(2, 8, 2)

#define EH_ARCH_DECL(a, b, c) extern "C" const EhArchInfo* eh_arch_##a##_##b##_##c(void);
EH_ARCH_LIST(EH_ARCH_DECL)
#undef EH_ARCH_DECL
#define EH_ARCH_DECL(a, b, c) extern "C" const EhArchInfo* eh_wide_##a##_##b##_##c(void);
EH_WIDE_LIST(EH_ARCH_DECL)
#undef EH_ARCH_DECL
