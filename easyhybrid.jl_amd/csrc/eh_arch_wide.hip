// One compiled shape of the row-split wide kernel; built with -DEH_NBI=.. -DEH_NBH=.. -DEH_NL=.. (see Makefile).
// NBH = 8 gets two variants: 8 waves per workgroup (one feature block each, two waves per SIMD so that
// one wave's LDS round trips hide behind the other's MFMAs) and 4 waves (two blocks each).
#include "eh_arch.hpp"
#include "eh_wide.hpp"
#include "eh_wide_bf16.hpp"
#include "eh_bf16_sample.hpp"

#ifndef EH_NBI
#error "build with -DEH_NBI -DEH_NBH -DEH_NL"
#endif

namespace {
constexpr int pick_nt() {
    if (sizeof(float) * EhWideGeom<EH_NBI, EH_NBH, EH_NL, 4, 4>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 4;
    if (sizeof(float) * EhWideGeom<EH_NBI, EH_NBH, EH_NL, 2, 4>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 2;
    return 1;
}
#ifdef EH_WIDE_NT
constexpr int NT = EH_WIDE_NT;
#else
constexpr int NT = pick_nt();
#endif

template <int NWV>
struct Var {
    using Geom = EhWideGeom<EH_NBI, EH_NBH, EH_NL, NT, NWV>;
    static constexpr size_t LDS = sizeof(float) * Geom::TOTAL_FLOATS;
    static_assert(LDS <= EH_LDS_LIMIT, "kernel shape does not fit the 160 KiB LDS of a gfx950 CU");

    template <int ACT, int MODE, bool PROG>
    static hipError_t prep1() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_wide_kernel<EH_NBI, EH_NBH, EH_NL, NT, NWV, ACT, MODE, PROG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    template <int ACT>
    static hipError_t prep2() {
        hipError_t e = prep1<ACT, EH_MODE_TRAIN, false>();
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_EVAL, false>();
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN, true>();      // EH_MECH_PROGRAM (fast == 4)
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_EVAL, true>();
        return e;
    }
    static hipError_t prepare() {
        hipError_t e;
        if ((e = prep2<EH_ACT_TANH>()) != hipSuccess) return e;
        if ((e = prep2<EH_ACT_SIGMOID>()) != hipSuccess) return e;
        if ((e = prep2<EH_ACT_RELU>()) != hipSuccess) return e;
        if ((e = prep2<EH_ACT_SWISH>()) != hipSuccess) return e;
        return prep2<EH_ACT_IDENTITY>();
    }
#define EH_GO(MODE, PROG) hipLaunchKernelGGL((eh_wide_kernel<EH_NBI, EH_NBH, EH_NL, NT, NWV, ACT, MODE, PROG>), dim3(grid), dim3(64 * NWV), LDS, stream, *net, *args)
    template <int ACT>
    static void go(int mode, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if (fast & 4) { if (mode == EH_MODE_TRAIN) EH_GO(EH_MODE_TRAIN, true); else EH_GO(EH_MODE_EVAL, true); }
        else { if (mode == EH_MODE_TRAIN) EH_GO(EH_MODE_TRAIN, false); else EH_GO(EH_MODE_EVAL, false); }
    }
#undef EH_GO
    static hipError_t launch(int mode, int act, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if (mode != EH_MODE_TRAIN && mode != EH_MODE_EVAL) return hipErrorNotSupported;
        switch (act) {
            case EH_ACT_TANH: go<EH_ACT_TANH>(mode, fast, grid, stream, net, args); break;
            case EH_ACT_SIGMOID: go<EH_ACT_SIGMOID>(mode, fast, grid, stream, net, args); break;
            case EH_ACT_RELU: go<EH_ACT_RELU>(mode, fast, grid, stream, net, args); break;
            case EH_ACT_SWISH: go<EH_ACT_SWISH>(mode, fast, grid, stream, net, args); break;
            case EH_ACT_IDENTITY: go<EH_ACT_IDENTITY>(mode, fast, grid, stream, net, args); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    static constexpr EhVariant info() { return EhVariant{NT, NWV, LDS, 1 << 30, &prepare, &launch, 1, 0}; }
};

// the bf16-forward kernels (eh_wide_bf16.hpp): NWV waves, the largest sample tile whose LDS map fits; tanh / sigmoid / relu /
// identity, registry models (a recorded closure gets its kernel compiled at run time, eh_jit.hip)
template <int NWV>
constexpr int pick_nt_bf() {
    if (sizeof(float) * EhBfGeom<EH_NBI, EH_NBH, EH_NL, 4, NWV>::TOTAL_FLOATS <= EH_LDS_LIMIT) return 4;
    return 2;
}
template <int NWV, int NTW = 0, int NS = 3>
struct VarBf {
    static constexpr int NTB = NTW ? NTW : pick_nt_bf<NWV>();
    using Geom = EhBfGeom<EH_NBI, EH_NBH, EH_NL, NTB, NWV, NS>;
    static constexpr size_t LDS = sizeof(float) * Geom::TOTAL_FLOATS;
    static_assert(LDS <= EH_LDS_LIMIT, "kernel shape does not fit the 160 KiB LDS of a gfx950 CU");
    template <int ACT, int MODE>
    static hipError_t prep1() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_widebf_kernel<EH_NBI, EH_NBH, EH_NL, NTB, NWV, ACT, MODE, false, NS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    template <int ACT>
    static hipError_t prep2() {
        hipError_t e = prep1<ACT, EH_MODE_TRAIN>();
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_EVAL>();
        return e;
    }
    static hipError_t prepare() {
        hipError_t e;
        if ((e = prep2<EH_ACT_TANH>()) != hipSuccess) return e;
        if ((e = prep2<EH_ACT_SIGMOID>()) != hipSuccess) return e;
        if ((e = prep2<EH_ACT_RELU>()) != hipSuccess) return e;
        return prep2<EH_ACT_IDENTITY>();
    }
#define EH_GO(MODE) hipLaunchKernelGGL((eh_widebf_kernel<EH_NBI, EH_NBH, EH_NL, NTB, NWV, ACT, MODE, false, NS>), dim3(grid), dim3(64 * NWV), LDS, stream, *net, *args)
    template <int ACT>
    static void go(int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if (mode == EH_MODE_TRAIN) EH_GO(EH_MODE_TRAIN); else EH_GO(EH_MODE_EVAL);
    }
#undef EH_GO
    static hipError_t launch(int mode, int act, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if ((mode != EH_MODE_TRAIN && mode != EH_MODE_EVAL) || (fast & 4)) return hipErrorNotSupported;
        switch (act) {
            case EH_ACT_TANH: go<EH_ACT_TANH>(mode, grid, stream, net, args); break;
            case EH_ACT_SIGMOID: go<EH_ACT_SIGMOID>(mode, grid, stream, net, args); break;
            case EH_ACT_RELU: go<EH_ACT_RELU>(mode, grid, stream, net, args); break;
            case EH_ACT_IDENTITY: go<EH_ACT_IDENTITY>(mode, grid, stream, net, args); break;
            default: return hipErrorNotSupported;
        }
        return hipGetLastError();
    }
    static constexpr EhVariant info() { return EhVariant{NTB, NWV, LDS, 1 << 30, &prepare, &launch, 1, NS == 3 ? 1 : 2}; }
};

// the sample-owned training kernel of the bf16 modes (eh_bf16_sample.hpp): what "precision" selects for a one-network model.  Its
// evaluation passes -- and nothing else: the handle keeps models with a mapped slab row off this variant -- run the row-split kernel.
template <int NWV, int NS>
struct VarBfs {
    using Old = VarBf<NWV, 0, NS>;
    using Geom = EhBfsGeom<EH_NBI, EH_NBH, EH_NL, NWV, NS>;
    static constexpr size_t LDS_T = sizeof(float) * Geom::TOTAL_FLOATS;
    static constexpr size_t LDS = LDS_T > Old::LDS ? LDS_T : Old::LDS;      // (one figure per variant: what a run-time build of its kernels is launched with)
    static_assert(LDS_T <= EH_LDS_LIMIT, "kernel shape does not fit the 160 KiB LDS of a gfx950 CU");
    template <int ACT>
    static hipError_t prep1() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_bfs_kernel<EH_NBI, EH_NBH, EH_NL, NWV, ACT, false, NS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_T);
    }
    static hipError_t prepare() {
        hipError_t e;
        if ((e = Old::prepare()) != hipSuccess) return e;
        if ((e = prep1<EH_ACT_TANH>()) != hipSuccess) return e;
        if ((e = prep1<EH_ACT_SIGMOID>()) != hipSuccess) return e;
        if ((e = prep1<EH_ACT_RELU>()) != hipSuccess) return e;
        return prep1<EH_ACT_IDENTITY>();
    }
#define EH_GO(ACT) hipLaunchKernelGGL((eh_bfs_kernel<EH_NBI, EH_NBH, EH_NL, NWV, ACT, false, NS>), dim3(grid), dim3(64 * NWV), LDS_T, stream, *net, *args)
    static hipError_t launch(int mode, int act, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if (mode == EH_MODE_EVAL) return Old::launch(mode, act, fast, grid, stream, net, args);
        if (mode != EH_MODE_TRAIN || (fast & 4) || args->rmap) return hipErrorNotSupported;
        switch (act) {
            case EH_ACT_TANH: EH_GO(EH_ACT_TANH); break;
            case EH_ACT_SIGMOID: EH_GO(EH_ACT_SIGMOID); break;
            case EH_ACT_RELU: EH_GO(EH_ACT_RELU); break;
            case EH_ACT_IDENTITY: EH_GO(EH_ACT_IDENTITY); break;
            default: return hipErrorNotSupported;
        }
        return hipGetLastError();
    }
#undef EH_GO
    static constexpr EhVariant info() { return EhVariant{NWV, NWV, LDS, 1 << 30, &prepare, &launch, 1, NS == 3 ? 1 : 2, Old::NTB}; }
};

using G0 = EhWideGeom<EH_NBI, EH_NBH, EH_NL, NT, 4>;
const EhArchInfo info = {
    EH_NBI, EH_NBH, EH_NL,
    G0::IP, G0::HP, G0::S0, G0::SH, G0::W0_OFF, G0::WH_OFF, G0::WO_OFF, G0::B_OFF, G0::PHI_OFF, G0::IMG_FLOATS,
    0,
#if EH_NBH == 8
    7, {Var<8>::info(), Var<4>::info(), VarBf<8>::info(), VarBf<8, 2>::info(), VarBf<8, 0, 1>::info(), VarBf<8, 2, 1>::info(), VarBfs<8, 1>::info()},
#else
    4, {Var<4>::info(), VarBf<4>::info(), VarBf<4, 0, 1>::info(), VarBfs<4, 1>::info(), {}, {}, {}, {}},
#endif
    1,
};
}   // namespace

#define EH_CAT_(a, b, c) eh_wide_##a##_##b##_##c
#define EH_CAT(a, b, c) EH_CAT_(a, b, c)
extern "C" const EhArchInfo* EH_CAT(EH_NBI, EH_NBH, EH_NL)(void) { return &info; }
