// One compiled kernel shape; built with -DEH_NBI=.. -DEH_NBH=.. -DEH_NL=.. (see Makefile).
// -DEH_EXTRA_VARIANTS adds the (NT, NW) variants used for tuning the small shapes.
#include "eh_arch.hpp"

#ifndef EH_NBI
#error "build with -DEH_NBI -DEH_NBH -DEH_NL"
#endif

namespace {
constexpr int NT0 = eh_pick_nt<EH_NBI, EH_NBH, EH_NL>();

template <int NT, int NW>
struct Var {
    using Geom = EhGeom<EH_NBI, EH_NBH, EH_NL, NT, NW>;
    static constexpr size_t LDS = sizeof(float) * Geom::TOTAL_FLOATS;
    static_assert(LDS <= EH_LDS_LIMIT, "kernel shape does not fit the 160 KiB LDS of a gfx950 CU");
    static constexpr size_t LDS_EVAL = sizeof(float) * Geom::TOTAL_FLOATS_EVAL;      // (forward / evaluation kernels: no hidden images, eh_device.hpp)
    static constexpr size_t LDS_EVAL_K1 = sizeof(float) * Geom::TOTAL_FLOATS_EVAL_K1;
    static constexpr bool HASPS = true;      // the P <= 4 kernels (FAST = 3) of every shape built with the fast paths
    // the cross-GPU (EH_MODE_TRAIN_P2P) kernels are built for the default variant of a shape only
#ifdef EH_EXTRA_VARIANTS
    static constexpr bool HASP2P = NT == 2 && NW == 8;
#else
    static constexpr bool HASP2P = true;
#endif

    template <int ACT, int MODE, int FAST>
    static hipError_t prep1() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, NW, ACT, MODE, FAST>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    }
    template <int ACT, int FAST>
    static hipError_t prepm() {           // the multi-step kernels keep their step-to-step state behind the work space: whatever LDS a CU has
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, NW, ACT, EH_MODE_TRAIN_MULTI, FAST>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)EH_LDS_LIMIT);
    }
    template <int ACT>
    static hipError_t prep2() {
        hipError_t e = prep1<ACT, EH_MODE_TRAIN, 0>();
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_EVAL, 0>();
        if constexpr (HASP2P) { if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN_P2P, 0>(); }
        if (e == hipSuccess) e = prepm<ACT, 0>();                     // several steps of one workgroup per launch (small minibatches)
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN, 4>();      // EH_MECH_PROGRAM kernels (no cross-GPU variant)
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_EVAL, 4>();
#ifdef EH_FAST_PATHS
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN, 1>();
        if (e == hipSuccess) e = prepm<ACT, 1>();
        if constexpr (HASPS) { if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN, 3>(); }
        if constexpr (HASPS) { if (e == hipSuccess) e = prepm<ACT, 3>(); }
        if (e == hipSuccess) e = prep1<ACT, EH_MODE_EVAL, 1>();
        if constexpr (HASP2P) {
            if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN_P2P, 1>();
            if constexpr (HASPS) { if (e == hipSuccess) e = prep1<ACT, EH_MODE_TRAIN_P2P, 3>(); }
        }
#endif
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
#define EH_GO(MODE, FAST) hipLaunchKernelGGL((eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, NW, ACT, MODE, FAST>), dim3(grid), dim3(64 * NW), (MODE) == EH_MODE_EVAL ? (((FAST) & 1) ? LDS_EVAL_K1 : LDS_EVAL) : LDS, stream, *net, *args)
    template <int ACT>
    static void go(int mode, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if (fast & 4) {
            if (mode == EH_MODE_TRAIN) EH_GO(EH_MODE_TRAIN, 4); else EH_GO(EH_MODE_EVAL, 4);
            return;
        }
        if (mode == EH_MODE_TRAIN_MULTI) {
            const size_t lds_ms = LDS + sizeof(float) * (size_t)eh_ms_extra_floats(net->n_theta, args->n_acc);
#define EH_GOM(FAST) hipLaunchKernelGGL((eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, NW, ACT, EH_MODE_TRAIN_MULTI, FAST>), dim3(1), dim3(64 * NW), lds_ms, stream, *net, *args)
#ifdef EH_FAST_PATHS
            if constexpr (HASPS) { if (fast == 3) { EH_GOM(3); return; } }
            if (fast & 1) { EH_GOM(1); return; }
#endif
            EH_GOM(0);
#undef EH_GOM
            return;
        }
        if (mode == EH_MODE_TRAIN_P2P) {
            if constexpr (HASP2P) {
#ifdef EH_FAST_PATHS
                if constexpr (HASPS) { if (fast == 3) { EH_GO(EH_MODE_TRAIN_P2P, 3); return; } }
                if (fast & 1) { EH_GO(EH_MODE_TRAIN_P2P, 1); return; }
#endif
                EH_GO(EH_MODE_TRAIN_P2P, 0);
            }
            return;
        }
#ifdef EH_FAST_PATHS
        if constexpr (HASPS) { if (mode == EH_MODE_TRAIN && fast == 3) { EH_GO(EH_MODE_TRAIN, 3); return; } }
        if (mode == EH_MODE_TRAIN && (fast & 1)) { EH_GO(EH_MODE_TRAIN, 1); return; }
        if (mode == EH_MODE_EVAL && (fast & 1)) { EH_GO(EH_MODE_EVAL, 1); return; }
#endif
        if (mode == EH_MODE_TRAIN) EH_GO(EH_MODE_TRAIN, 0); else EH_GO(EH_MODE_EVAL, 0);
    }
#undef EH_GO
    static hipError_t launch(int mode, int act, int fast, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
        if (mode == EH_MODE_TRAIN_P2P && !HASP2P) return hipErrorNotSupported;
        if (fast == 3 && !HASPS) return hipErrorNotSupported;
        if ((fast & 4) && (fast != 4 || mode == EH_MODE_TRAIN_P2P || mode == EH_MODE_TRAIN_MULTI)) return hipErrorNotSupported;
        if (mode == EH_MODE_TRAIN_MULTI && (grid != 1 || LDS + sizeof(float) * (size_t)eh_ms_extra_floats(net->n_theta, args->n_acc) > EH_LDS_LIMIT)) return hipErrorInvalidValue;
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
    static constexpr EhVariant info() { return EhVariant{NT, NW, LDS, NW * Geom::WAVE_WS, &prepare, &launch, 0, 0, 0, LDS_EVAL}; }
};

using G0 = EhGeom<EH_NBI, EH_NBH, EH_NL, NT0, 4>;
const EhArchInfo info = {
    EH_NBI, EH_NBH, EH_NL,
    G0::IP, G0::HP, G0::S0, G0::SH, G0::W0_OFF, G0::WH_OFF, G0::WO_OFF, G0::B_OFF, G0::PHI_OFF, G0::IMG_FLOATS,
#ifdef EH_FAST_PATHS
    1,
#else
    0,
#endif
#ifdef EH_EXTRA_VARIANTS
    4, {Var<NT0, 4>::info(), Var<2, 8>::info(), Var<1, 16>::info(), Var<1, 8>::info()}
#else
    1, {Var<NT0, 4>::info(), {}, {}, {}}
#endif
};
}   // namespace

#define EH_CAT_(a, b, c) eh_arch_##a##_##b##_##c
#define EH_CAT(a, b, c) EH_CAT_(a, b, c)
extern "C" const EhArchInfo* EH_CAT(EH_NBI, EH_NBH, EH_NL)(void) { return &info; }
