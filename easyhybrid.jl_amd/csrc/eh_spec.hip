// One canonical model descriptor baked into its step kernels AHEAD OF TIME: the headline of BASELINE.json must
// not depend on a run-time compiler being present and willing.  Built once per descriptor by the Makefile with
//   -DEH_SPEC_ID=k -DEH_SPEC_NS=eh_spec_ns_k -DEH_SPEC_NET='P,K,G,T,F,n_theta,g_off,scale_nn,mech,n_par,loss,n_out,targ_out,par_kind,par_idx,forc_col,loss_t'
//   -DEH_SPEC_FAMILY=0|1|2|3 (per-wave | row-split | row-split bf16 | sample-owned bf16: NT = NW, evaluation on the row-split kernel with -DEH_SPEC_EVAL_NT)
//   -DEH_SPEC_SHAPE=NBI,NBH,NL,NT,NW  -DEH_SPEC_ACT=a  -DEH_SPEC_FAST=f  [-DEH_SPEC_NSPLIT=3|1]
// EH_SPEC_NET is the macro the run-time specialiser defines (eh_jit.hip): the kernel sources are the same, the constants fold the same
// way.  The descriptor strings in the Makefile are what `EH_JIT_TRACE=1` prints for the BASELINE configurations
// (tests/test_gpu_headline.py::test_canonical_descriptors_run_kernels_specialised_ahead_of_time checks that they still match).
#if EH_SPEC_FAMILY == 3
#include "eh_arch.hpp"
#include "eh_bf16_sample.hpp"
#elif EH_SPEC_FAMILY == 2
#include "eh_arch.hpp"
#include "eh_wide_bf16.hpp"
#elif EH_SPEC_FAMILY == 1
#include "eh_arch.hpp"
#include "eh_wide.hpp"
#else
#include "eh_arch.hpp"
#endif

namespace {
constexpr int SH[5] = {EH_SPEC_SHAPE};
constexpr int NBI = SH[0], NBH = SH[1], NL = SH[2], NT = SH[3], NW = SH[4], ACT = EH_SPEC_ACT, FAST = EH_SPEC_FAST;
#ifndef EH_SPEC_NSPLIT
#define EH_SPEC_NSPLIT 3
#endif
#if EH_SPEC_FAMILY == 3
using GeomT = EhBfsGeom<NBI, NBH, NL, NW, EH_SPEC_NSPLIT>;
using GeomE = EhBfGeom<NBI, NBH, NL, EH_SPEC_EVAL_NT, NW, EH_SPEC_NSPLIT>;
struct Geom { static constexpr int TOTAL_FLOATS = GeomT::TOTAL_FLOATS > GeomE::TOTAL_FLOATS ? GeomT::TOTAL_FLOATS : GeomE::TOTAL_FLOATS; };
template <int MODE> struct SpecK;
template <> struct SpecK<EH_MODE_TRAIN> { static constexpr auto fn = &eh_bfs_kernel<NBI, NBH, NL, NW, ACT, false, EH_SPEC_NSPLIT>; };
template <> struct SpecK<EH_MODE_EVAL> { static constexpr auto fn = &eh_widebf_kernel<NBI, NBH, NL, EH_SPEC_EVAL_NT, NW, ACT, EH_MODE_EVAL, false, EH_SPEC_NSPLIT>; };
#define EH_SPEC_KERNEL(MODE) (*SpecK<MODE>::fn)
constexpr bool HASP2P = false;
#elif EH_SPEC_FAMILY == 2
using Geom = EhBfGeom<NBI, NBH, NL, NT, NW, EH_SPEC_NSPLIT>;
#define EH_SPEC_KERNEL(MODE) eh_widebf_kernel<NBI, NBH, NL, NT, NW, ACT, MODE, false, EH_SPEC_NSPLIT>
constexpr bool HASP2P = false;
#elif EH_SPEC_FAMILY == 1
using Geom = EhWideGeom<NBI, NBH, NL, NT, NW>;
#define EH_SPEC_KERNEL(MODE) eh_wide_kernel<NBI, NBH, NL, NT, NW, ACT, MODE, false>
constexpr bool HASP2P = false;
#else
using Geom = EhGeom<NBI, NBH, NL, NT, NW>;
#define EH_SPEC_KERNEL(MODE) eh_step_kernel<NBI, NBH, NL, NT, NW, ACT, MODE, ((MODE) == EH_MODE_EVAL ? (FAST & 5) : FAST)>
constexpr bool HASP2P = (FAST & 4) == 0;
constexpr bool HASMULTI = (FAST & 4) == 0 && EhNet{EH_SPEC_NET}.T == 1;
#endif
constexpr size_t LDS = sizeof(float) * Geom::TOTAL_FLOATS;
#if EH_SPEC_FAMILY == 0
constexpr size_t LDS_EVAL = sizeof(float) * ((FAST & 1) ? Geom::TOTAL_FLOATS_EVAL_K1 : Geom::TOTAL_FLOATS_EVAL);      // (the per-wave forward / evaluation kernels: no hidden images, eh_device.hpp)
#else
constexpr size_t LDS_EVAL = LDS;
#endif
static_assert(LDS <= EH_LDS_LIMIT, "kernel shape does not fit the 160 KiB LDS of a gfx950 CU");

hipError_t prepare() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&EH_SPEC_KERNEL(EH_MODE_TRAIN)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&EH_SPEC_KERNEL(EH_MODE_EVAL)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
#if EH_SPEC_FAMILY == 0
    if constexpr (HASP2P) { if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&EH_SPEC_KERNEL(EH_MODE_TRAIN_P2P)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS); }
    if constexpr (HASMULTI) { if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&EH_SPEC_KERNEL(EH_MODE_TRAIN_MULTI)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)EH_LDS_LIMIT); }
#endif
    return e;
}
hipError_t launch(int mode, int grid, hipStream_t stream, const EhNet* net, const EhStepArgs* args) {
#if EH_SPEC_FAMILY == 3
    if (mode == EH_MODE_TRAIN && args->rmap) return hipErrorNotSupported;      // (the sample-owned kernel writes plain canonical slab rows)
#endif
    if (mode == EH_MODE_TRAIN) hipLaunchKernelGGL((EH_SPEC_KERNEL(EH_MODE_TRAIN)), dim3(grid), dim3(64 * NW), LDS, stream, *net, *args);
    else if (mode == EH_MODE_EVAL) hipLaunchKernelGGL((EH_SPEC_KERNEL(EH_MODE_EVAL)), dim3(grid), dim3(64 * NW), LDS_EVAL, stream, *net, *args);
#if EH_SPEC_FAMILY == 0
    else if (mode == EH_MODE_TRAIN_P2P && HASP2P) hipLaunchKernelGGL((EH_SPEC_KERNEL(EH_MODE_TRAIN_P2P)), dim3(grid), dim3(64 * NW), LDS, stream, *net, *args);
    else if (mode == EH_MODE_TRAIN_MULTI && HASMULTI) {      // several steps of one workgroup per launch, the state between them in LDS behind the work space
        const size_t lds_ms = LDS + sizeof(float) * (size_t)eh_ms_extra_floats(net->n_theta, args->n_acc);
        if (grid != 1 || lds_ms > EH_LDS_LIMIT) return hipErrorInvalidValue;
        hipLaunchKernelGGL((EH_SPEC_KERNEL(EH_MODE_TRAIN_MULTI)), dim3(1), dim3(64 * NW), lds_ms, stream, *net, *args);
    }
#endif
    else return hipErrorNotSupported;
    return hipGetLastError();
}
#define EH_STR_(x) #x
#define EH_STR(x) EH_STR_(x)
const EhSpecKernel spec = {EhNet{EH_SPEC_NET}, EH_SPEC_FAMILY != 0, EH_SPEC_FAMILY >= 2 ? (EH_SPEC_NSPLIT == 3 ? 1 : 2) : 0, NBI, NBH, NL, NT, NW, ACT, FAST, EH_SPEC_FAMILY == 3 ? 1 : 0, LDS,
                           "descriptor " EH_STR(EH_SPEC_ID) " of csrc/Makefile, specialised ahead of time", &prepare, &launch};
}   // namespace

#define EH_CAT_(a) eh_spec_##a
#define EH_CAT(a) EH_CAT_(a)
extern "C" const EhSpecKernel* EH_CAT(EH_SPEC_ID)(void) { return &spec; }
