// One compiled kernel shape; built with -DEH_NBI=.. -DEH_NBH=.. -DEH_NL=.. (see Makefile).
#include "eh_arch.hpp"

#ifndef EH_NBI
#error "build with -DEH_NBI -DEH_NBH -DEH_NL"
#endif

namespace {
constexpr int NT = eh_pick_nt<EH_NBI, EH_NBH, EH_NL>();
using Geom = EhGeom<EH_NBI, EH_NBH, EH_NL, NT>;
static_assert(sizeof(float) * Geom::TOTAL_FLOATS <= EH_LDS_LIMIT, "kernel shape does not fit the 160 KiB LDS of a gfx950 CU");

hipError_t prepare() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, EH_MODE_TRAIN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * Geom::TOTAL_FLOATS));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, EH_MODE_EVAL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * Geom::TOTAL_FLOATS));
}

hipError_t launch(int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
    const size_t lds = sizeof(float) * Geom::TOTAL_FLOATS;
    if (mode == EH_MODE_TRAIN)
        hipLaunchKernelGGL((eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, EH_MODE_TRAIN>), dim3(grid), dim3(256), lds, stream, *net, *args);
    else
        hipLaunchKernelGGL((eh_step_kernel<EH_NBI, EH_NBH, EH_NL, NT, EH_MODE_EVAL>), dim3(grid), dim3(256), lds, stream, *net, *args);
    return hipGetLastError();
}

const EhArchInfo info = {EH_NBI, EH_NBH, EH_NL, NT, sizeof(float) * Geom::TOTAL_FLOATS, 4 * Geom::WAVE_WS, &prepare, &launch};
}   // namespace

#define EH_CAT_(a, b, c) eh_arch_##a##_##b##_##c
#define EH_CAT(a, b, c) EH_CAT_(a, b, c)
extern "C" const EhArchInfo* EH_CAT(EH_NBI, EH_NBH, EH_NL)(void) { return &info; }
