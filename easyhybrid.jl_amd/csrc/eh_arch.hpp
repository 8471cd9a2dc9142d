// Table of compiled kernel shapes.  One translation unit (eh_arch.hip, built once per
// (EH_NBI, EH_NBH, EH_NL) by the Makefile) instantiates eh_step_kernel for one padded MLP shape:
//   NBI  input  blocks of 16  (P <= 16*NBI)
//   NBH  hidden blocks of 16  (every hidden width <= 16*NBH)
//   NL   hidden layers
// and the macro-tile size NT (x16 samples per wave) is the largest of {4,2,1} whose LDS image fits
// the 160 KiB of a gfx950 CU with one 4-wave workgroup per CU.
#pragma once
#include "eh_device.hpp"

struct EhArchInfo {
    int nbi, nbh, nl, nt;
    size_t lds_bytes;        // dynamic LDS per workgroup
    int red_floats;          // floats available to the end-of-kernel reduction (must be >= n_acc)
    hipError_t (*prepare)(void);   // raises the dynamic-LDS limit of both kernels
    hipError_t (*launch)(int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args);
};

constexpr size_t EH_LDS_LIMIT = 160 * 1024;

template <int NBI, int NBH, int NL>
constexpr int eh_pick_nt() {
    if (sizeof(float) * EhGeom<NBI, NBH, NL, 4>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 4;
    if (sizeof(float) * EhGeom<NBI, NBH, NL, 2>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 2;
    return 1;
}

#define EH_ARCH_LIST(X) \
    X(1, 1, 1) X(1, 1, 2) X(1, 1, 3) \
    X(1, 2, 1) X(1, 2, 2) X(1, 2, 3) \
    X(1, 4, 1) X(1, 4, 2) X(1, 4, 3) \
    X(2, 1, 1) X(2, 1, 2) X(2, 1, 3) \
    X(2, 2, 1) X(2, 2, 2) X(2, 2, 3) \
    X(2, 4, 1) X(2, 4, 2) X(2, 4, 3)

#define EH_ARCH_DECL(a, b, c) extern "C" const EhArchInfo* eh_arch_##a##_##b##_##c(void);
EH_ARCH_LIST(EH_ARCH_DECL)
#undef EH_ARCH_DECL
