// The collectives inside libeasyhybrid_hip.so (declared in include/easyhybrid_hip.h): the RCCL communicator a
// handle can own (bound with dlopen on first use), the local group of ONE process driving several handles, and the peer-to-peer
// exchange of the fused step kernel (EhP2P, csrc/eh_device.hpp) -- between processes over HIP IPC, or between the handles of one
// process by plain pointers.  The kernels that produce what is exchanged are launched from eh_api.hip (eh_dp_*).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>

#include "eh_internal.hpp"

// RCCL is bound with dlopen by the first eh_comm_* call, not linked: a process that never asks for the library's own communicator
// (every single-GPU user; a host that brings its own collective, like the torch.distributed harness) loads neither RCCL nor
// the rocm_smi it drags in -- whose static destructors were seen to abort at exit -- and a host that already has an RCCL in the
// process (torch bundles one) shares that copy instead of getting a second one.
struct EhRccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)(void) = nullptr;
    ncclResult_t (*GroupEnd)(void) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static EhRccl g_rccl;
static bool rccl_bind(std::string* why) {
    if (g_rccl.so) return true;
    void* so = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((so = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!so) { *why = std::string("librccl not found: ") + dlerror(); return false; }
    EhRccl r;
    r.so = so;
#define EH_SYM(field, sym) *(void**)(&r.field) = dlsym(so, sym); if (!r.field) { *why = std::string("librccl lacks ") + sym; return false; }
    EH_SYM(GetUniqueId, "ncclGetUniqueId") EH_SYM(CommInitRank, "ncclCommInitRank") EH_SYM(CommDestroy, "ncclCommDestroy")
    EH_SYM(AllReduce, "ncclAllReduce") EH_SYM(GroupStart, "ncclGroupStart") EH_SYM(GroupEnd, "ncclGroupEnd") EH_SYM(GetErrorString, "ncclGetErrorString")
#undef EH_SYM
    g_rccl = r;                      // (never unloaded)
    return true;
}
#define RCCL_BIND(h)                                                                           \
    do {                                                                                       \
        std::string why_;                                                                      \
        if (!rccl_bind(&why_)) return fail(h, EH_ERCCL, "RCCL: %s", why_.c_str());             \
    } while (0)

// ---- local communicator: the handles of ONE process (one host thread issuing to several devices / streams, SURVEY section 8(b)
// threading row) sum their buffers without RCCL.  The buffers are a few KB: every member's stream waits (events) until all
// members' producers have run, one small kernel per member then reads ALL members' buffers -- directly, over peer-mapped device
// memory (xGMI) when they live on different GPUs -- and adds them in rank order, so every replica gets bit-identical sums; a
// second event round keeps a member from overwriting its buffer while a peer still reads it.
struct EhLocalGroup {
    int n = 0;
    eh_handle* m[EH_GSHARDS] = {nullptr};
    hipEvent_t ready[EH_GSHARDS] = {nullptr}, done[EH_GSHARDS] = {nullptr};
    float* sum[EH_GSHARDS] = {nullptr};       // per member, on its device: where its kernel leaves the sums before they replace the buffer
    size_t cap = 0;                           // floats each of them holds
};
struct EhLocalPtrs { const float* p[EH_GSHARDS]; };
__global__ void __launch_bounds__(256) eh_lgroup_sum_kernel(EhLocalPtrs src, int world, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v[EH_GSHARDS];
#pragma unroll
    for (int r = 0; r < EH_GSHARDS; ++r) v[r] = r < world ? __builtin_nontemporal_load(src.p[r] + i) : 0.0f;     // all loads in flight together; rank order below
    float acc = v[0];
#pragma unroll
    for (int r = 1; r < EH_GSHARDS; ++r) if (r < world) acc += v[r];
    out[i] = acc;
}
struct EhLocalReq { eh_handle* h; float* buf; size_t n; };
static thread_local int g_group_depth = 0;                 // eh_comm_group_begin nesting of this host thread
static thread_local bool g_group_rccl = false;             // ncclGroupStart was issued for the open bracket
static thread_local std::vector<EhLocalReq> g_group_reqs;  // all-reduces of local-group members, run at eh_comm_group_end

// eh_p2p_selftest: one exchange round of the EhP2P protocol with a known vector per rank
__device__ __forceinline__ float eh_p2p_test_value(int rank, int i, unsigned seq) { return (float)((rank + 1) * 1000 + (i % 97) + (int)(seq & 255u)); }
__global__ __launch_bounds__(256) void eh_p2p_test_kernel(const EhP2P* P, int slot, unsigned seq, int n_acc, int* bad) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n_acc; i += 256) {
        const unsigned long long w = eh_ll_pack(eh_p2p_test_value(P->rank, i, seq), seq);
        for (int r = 0; r < P->world; ++r)
            __hip_atomic_store(&P->peer_recv[r][((long long)slot * EH_GSHARDS + P->rank) * n_acc + i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int i = tid; i < n_acc; i += 256) {
        auto ad = [&](int r) -> const unsigned long long* {
            return r < P->world ? P->peer_recv[P->rank] + ((long long)slot * EH_GSHARDS + r) * n_acc + i : nullptr;
        };
        unsigned long long w[EH_GSHARDS];
        float got[EH_GSHARDS];
        eh_ll_issue(ad, seq, w);
        eh_ll_finish(P, ad, seq, w, got, 5ull * EH_P2P_DEADLINE_TICKS / 2);        // 5 s: at start-up the ranks may be a while apart
        float sum = 0.0f, want = 0.0f;
        for (int r = 0; r < P->world; ++r) { sum += got[r]; want += eh_p2p_test_value(r, i, seq); }
        if (sum != want) atomicAdd(bad, 1);
    }
}

// every member's kernels read (local group) or write (peer-to-peer exchange) every other member's buffers: peer access between the
// distinct devices of the list
static int eh_enable_peer_access(eh_handle* const* handles, int n, const char* who) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int a = handles[i]->device, b = handles[j]->device;
            if (a == b) continue;
            int can = 0;
            HIPCHK(handles[i], hipDeviceCanAccessPeer(&can, a, b));
            if (!can) return fail(handles[i], EH_EUNSUPPORTED, "%s: device %d cannot map the memory of device %d (no peer access): use eh_comm_init (RCCL)", who, a, b);
            HIPCHK(handles[i], hipSetDevice(a));
            hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess) return fail(handles[i], EH_EHIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", a, b, hipGetErrorString(e));
        }
    return EH_OK;
}

extern "C" {

// ---- cross-GPU exchange without a collective call (EhP2P, csrc/eh_device.hpp) ----------------------
static size_t p2p_recv_words(const eh_handle* h) { return (size_t)3 * EH_GSHARDS * h->n_acc; }

// the buffers of one rank: the uncached receive buffer the peers store into, the local staging accumulators, the ticket counters
// and the descriptor; handle_out != nullptr: also export the receive buffer over HIP IPC (peers in other processes)
static int p2p_alloc(eh_handle* h, int32_t world, int32_t rank, hipIpcMemHandle_t* handle_out, const char* who) {
    // (world == 1 is a loopback: the rank publishes to and reads from itself -- measures the cost of the machinery)
    if (world < 1 || world > EH_GSHARDS || rank < 0 || rank >= world) return fail(h, EH_EINVAL, "%s: world %d (1..%d), rank %d", who, world, EH_GSHARDS, rank);
    if (!h->fused) return fail(h, EH_ESTATE, "%s: set the fused_update option first", who);
    if (h->net.T != 1) return fail(h, EH_EUNSUPPORTED, "%s: the peer-to-peer exchange is built for single-target models (use the all-reduce seam)", who);
    if (h->net.mech == EH_MECH_PROGRAM) return fail(h, EH_EUNSUPPORTED, "%s: the program kernels have no cross-GPU variant (use the all-reduce seam)", who);
    if (h->act == EH_ACT_PER_NET) return fail(h, EH_EUNSUPPORTED, "%s: the per-net activation kernels have no cross-GPU variant (use the all-reduce seam)", who);
    if (h->p2p_on || h->p2p_alloc) return fail(h, EH_ESTATE, "%s: already initialised", who);
    HIPCHK(h, hipSetDevice(h->device));
    FLUSH(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // other GPUs store into the receive buffer: nothing of it may linger in a cache of this one
    const size_t bytes = p2p_recv_words(h) * sizeof(unsigned long long);
    unsigned long long* buf = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, EH_EUNSUPPORTED, "%s: no uncached / fine-grained device memory (%s)", who, hipGetErrorString(e)); }
    HIPCHK(h, hipMemset(buf, 0, bytes));                  // sequence 0 everywhere: nothing has arrived
    if (handle_out) {
        e = hipIpcGetMemHandle(handle_out, buf);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(buf); return fail(h, EH_EUNSUPPORTED, "%s: hipIpcGetMemHandle: %s", who, hipGetErrorString(e)); }
    }
    h->p2p_recv = buf; h->p2p_local = handle_out == nullptr;      // (known from here on: an error below leaves buffers that the caller's clean-up frees)
    HIPCHK(h, hipMalloc(&h->p2p_stage, ((size_t)3 * EH_GSHARDS * h->n_acc + 4) * sizeof(float)));      // (+4: the prologue reads five scalars behind every shard's gradient, whatever T)
    HIPCHK(h, hipMemset(h->p2p_stage, 0, ((size_t)3 * EH_GSHARDS * h->n_acc + 4) * sizeof(float)));
    HIPCHK(h, hipMalloc(&h->p2p_ctr, 32 * (1 + EH_P2P_GROUPS) * sizeof(unsigned)));      // [0] top ticket, [1] error flag, [2] self-test mismatches, [32 (1 + g)] group tickets
    HIPCHK(h, hipMemset(h->p2p_ctr, 0, 32 * (1 + EH_P2P_GROUPS) * sizeof(unsigned)));
    HIPCHK(h, hipMalloc(&h->p2p_dev, sizeof(EhP2P)));
    HIPCHK(h, hipDeviceSynchronize());        // the memsets ran on the null stream, which the engine's non-blocking stream does not wait for
    h->p2p_world = world; h->p2p_rank = rank; h->p2p_alloc = true; h->p2p_seq = 0; h->p2p_local = handle_out == nullptr;
    return EH_OK;
}

// peers known (h->p2p_peer[]): the descriptor goes to the device and the step kernels switch to the exchanging variant
static int p2p_wire(eh_handle* h) {
    EhP2P P;
    memset(&P, 0, sizeof P);
    for (int r = 0; r < h->p2p_world; ++r) P.peer_recv[r] = (unsigned long long*)h->p2p_peer[r];
    P.stage = h->p2p_stage; P.counter = h->p2p_ctr; P.err = (int*)(h->p2p_ctr + 1);
    P.world = h->p2p_world; P.rank = h->p2p_rank;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->p2p_dev, &P, sizeof P, hipMemcpyHostToDevice));
    h->p2p_host = P;
    h->p2p_on = true;
    return EH_OK;
}

int32_t eh_p2p_init(eh_handle* h, int32_t world, int32_t rank, void* handle_out, int64_t handle_bytes) {
    if (!h || !handle_out) return EH_EINVAL;
    if (handle_bytes < (int64_t)sizeof(hipIpcMemHandle_t)) return fail(h, EH_EINVAL, "eh_p2p_init: handle buffer of %lld bytes, need %zu", (long long)handle_bytes, sizeof(hipIpcMemHandle_t));
    hipIpcMemHandle_t hd;
    if (int rc = p2p_alloc(h, world, rank, &hd, "eh_p2p_init")) return rc;
    memcpy(handle_out, &hd, sizeof hd);
    return EH_OK;
}

int32_t eh_p2p_attach(eh_handle* h, const void* handles, int64_t handle_stride) {
    if (!h || !handles) return EH_EINVAL;
    if (!h->p2p_alloc || h->p2p_on) return fail(h, EH_ESTATE, "eh_p2p_attach: call eh_p2p_init first (once)");
    if (handle_stride < (int64_t)sizeof(hipIpcMemHandle_t)) return fail(h, EH_EINVAL, "eh_p2p_attach: handle stride %lld", (long long)handle_stride);
    HIPCHK(h, hipSetDevice(h->device));
    for (int r = 0; r < h->p2p_world; ++r) {
        void* ptr = h->p2p_recv;
        if (r != h->p2p_rank) {
            hipIpcMemHandle_t hd;
            memcpy(&hd, (const char*)handles + (size_t)r * handle_stride, sizeof hd);
            hipError_t e = hipIpcOpenMemHandle(&ptr, hd, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, EH_EUNSUPPORTED, "eh_p2p_attach: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e)); }
        }
        h->p2p_peer[r] = ptr;
    }
    return p2p_wire(h);
}

// `rounds` exchanges of a known vector per rank through the real buffers; EVERY rank must call it at the same point.
static int p2p_selftest_launch(eh_handle* h, int k) {
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(eh_p2p_test_kernel, dim3(1), dim3(256), 0, h->stream, h->p2p_dev, k % 3, ++h->p2p_seq, h->n_acc, (int*)(h->p2p_ctr + 2));
    HIPCHK(h, hipGetLastError());
    return EH_OK;
}
static int p2p_selftest_collect(eh_handle* h, int32_t* ok) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unsigned c[3] = {0, 0, 0};
    HIPCHK(h, hipMemcpy(c, h->p2p_ctr, sizeof c, hipMemcpyDeviceToHost));
    *ok = (c[1] == 0 && c[2] == 0) ? 1 : 0;
    return EH_OK;
}
int32_t eh_p2p_selftest(eh_handle* h, int32_t rounds, int32_t* ok) {
    if (!h || !ok) return EH_EINVAL;
    *ok = 0;
    if (!h->p2p_on) return fail(h, EH_ESTATE, "eh_p2p_selftest: call eh_p2p_attach first");
    if (h->pending) return fail(h, EH_ESTATE, "eh_p2p_selftest: a training step is pending");
    for (int k = 0; k < rounds; ++k)
        if (int rc = p2p_selftest_launch(h, k)) return rc;
    return p2p_selftest_collect(h, ok);
}

// back to the host-side all-reduce (RCCL) of EH_BUF_GACC
static int p2p_disable_one(eh_handle* h) {
    HIPCHK(h, hipSetDevice(h->device));
    for (int r = 0; r < EH_GSHARDS; ++r) {
        if (h->p2p_peer[r] && r != h->p2p_rank && !h->p2p_local) (void)hipIpcCloseMemHandle(h->p2p_peer[r]);
        h->p2p_peer[r] = nullptr; h->p2p_group[r] = nullptr;
    }
    h->p2p_on = false; h->p2p_alloc = false; h->p2p_local = false;
    (void)hipFree(h->p2p_recv); h->p2p_recv = nullptr;
    (void)hipFree(h->p2p_stage); h->p2p_stage = nullptr;
    (void)hipFree(h->p2p_ctr); h->p2p_ctr = nullptr;
    (void)hipFree(h->p2p_dev); h->p2p_dev = nullptr;
    HIPCHK(h, hipMemsetAsync(h->gacc, 0, (size_t)3 * EH_GSHARDS * h->n_acc * sizeof(float), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EH_OK;
}

// back to the all-reduce of EH_BUF_GACC (RCCL / local group).  A handle wired by eh_p2p_init_local takes its whole group along:
// the members hold plain pointers to each other's receive buffers, so every member is drained before any buffer is freed.
int32_t eh_p2p_disable(eh_handle* h) {
    if (!h) return EH_EINVAL;
    if (!h->p2p_alloc) return EH_OK;
    eh_handle* grp[EH_GSHARDS] = {h};
    int n = 1;
    if (h->p2p_local && h->p2p_on) { n = 0; for (int r = 0; r < h->p2p_world; ++r) if (h->p2p_group[r]) grp[n++] = h->p2p_group[r]; }
    int rc = EH_OK;
    for (int i = 0; i < n; ++i) {           // every member's pending update applied and its stream idle (a member that ran into the deadline returns at once)
        eh_handle* m = grp[i];
        HIPCHK(m, hipSetDevice(m->device));
        if (int r2 = flush_pending(m)) rc = rc ? rc : r2;
        HIPCHK(m, hipStreamSynchronize(m->stream));
    }
    for (int i = 0; i < n; ++i)
        if (int r2 = p2p_disable_one(grp[i])) rc = rc ? rc : r2;
    (void)hipSetDevice(h->device);
    return rc;
}

// ---- the same exchange between the handles of ONE process (one host thread, one handle per device: SURVEY section 8(b), threading
// row): no IPC -- every member's descriptor holds plain device pointers to the others' receive buffers (peer access enabled
// between distinct devices).  rank = position in the list.  Runs the start-up self-test on all members together (`selftest_rounds`
// exchanges of known vectors through the real buffers; a single thread could not run eh_p2p_selftest member by member: the first
// would wait for peers that have not been launched yet).  *ok = 0: the test failed and every member is back on the all-reduce.
int32_t eh_p2p_init_local(eh_handle* const* handles, int32_t n, int32_t selftest_rounds, int32_t* ok) {
    if (!handles || !ok || n < 1 || n > EH_GSHARDS) return fail(nullptr, EH_EINVAL, "eh_p2p_init_local: %d handles (1..%d)", n, EH_GSHARDS);
    *ok = 0;
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail(nullptr, EH_EINVAL, "eh_p2p_init_local: handle %d is NULL", i);
        for (int j = 0; j < i; ++j) if (handles[j] == handles[i]) return fail(handles[i], EH_EINVAL, "eh_p2p_init_local: handle %d is listed twice", i);
        if (handles[i]->net.n_theta != handles[0]->net.n_theta || handles[i]->n_acc != handles[0]->n_acc)
            return fail(handles[i], EH_EINVAL, "eh_p2p_init_local: handle %d is a different model (%d parameters, handle 0 has %d)", i, handles[i]->net.n_theta, handles[0]->net.n_theta);
        if (handles[i]->p2p_on || handles[i]->p2p_alloc) return fail(handles[i], EH_ESTATE, "eh_p2p_init_local: handle %d already has a peer-to-peer exchange (eh_p2p_disable first)", i);
        if (handles[i]->pending) return fail(handles[i], EH_ESTATE, "eh_p2p_init_local: handle %d has a training step pending (eh_synchronize first)", i);
    }
    if (int rc = eh_enable_peer_access(handles, n, "eh_p2p_init_local")) return rc;
    // (every member's stream is drained BEFORE any buffer is freed: self-test kernels of the earlier members may still be storing into their
    //  peers' receive buffers -- as eh_p2p_disable does; advisor, round 4)
    auto undo = [&](int rc) {
        for (int i = 0; i < n; ++i) { (void)hipSetDevice(handles[i]->device); (void)hipStreamSynchronize(handles[i]->stream); }
        for (int i = 0; i < n; ++i) if (handles[i]->p2p_alloc || handles[i]->p2p_recv || handles[i]->p2p_stage || handles[i]->p2p_ctr || handles[i]->p2p_dev) { handles[i]->p2p_on = false; (void)p2p_disable_one(handles[i]); }
        (void)hipGetLastError();
        return rc;
    };
    for (int i = 0; i < n; ++i)
        if (int rc = p2p_alloc(handles[i], n, i, nullptr, "eh_p2p_init_local")) return undo(rc);
    for (int i = 0; i < n; ++i) {
        for (int r = 0; r < n; ++r) { handles[i]->p2p_peer[r] = handles[r]->p2p_recv; handles[i]->p2p_group[r] = handles[r]; }
        if (int rc = p2p_wire(handles[i])) return undo(rc);
    }
    int all = 1;
    for (int k = 0; k < selftest_rounds; ++k)          // round k of every member is in flight before anyone waits for it
        for (int i = 0; i < n; ++i)
            if (int rc = p2p_selftest_launch(handles[i], k)) return undo(rc);
    for (int i = 0; i < n; ++i) {
        int32_t one = 0;
        if (int rc = p2p_selftest_collect(handles[i], &one)) return undo(rc);
        all = all && one;
    }
    if (getenv("EH_DEBUG_P2P_FAIL_SELFTEST")) all = 0;      // tests: walk the refusal path
    if (!all) { (void)eh_p2p_disable(handles[0]); return EH_OK; }
    *ok = 1;
    return EH_OK;
}

// Drain every member and look at the deadline flags.  *healthy = 1: no exchange was missed.  Otherwise (a member stepped without
// the others, a device fell behind by more than the 2 s deadline): every member leaves the peer-to-peer exchange -- later
// eh_dp_train_step_group calls go through the members' communicator (local group / RCCL) -- and takes member 0's parameters AND
// optimiser state (a rank that missed an exchange substituted zeros, so theta, the moments and the beta products can all differ).
int32_t eh_p2p_check_local(eh_handle* const* handles, int32_t n, int32_t* healthy) {
    if (!handles || !healthy || n < 1 || n > EH_GSHARDS) return fail(nullptr, EH_EINVAL, "eh_p2p_check_local: %d handles (1..%d)", n, EH_GSHARDS);
    *healthy = 1;
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail(nullptr, EH_EINVAL, "eh_p2p_check_local: handle %d is NULL", i);
        if (!handles[i]->p2p_on || !handles[i]->p2p_local || handles[i]->p2p_world != n || handles[i]->p2p_group[i] != handles[i])
            return fail(handles[i], EH_ESTATE, "eh_p2p_check_local: handle %d is not member %d of a local peer-to-peer group of %d (eh_p2p_init_local)", i, i, n);
    }
    for (int i = 0; i < n; ++i) {
        eh_handle* h = handles[i];
        HIPCHK(h, hipSetDevice(h->device));
        FLUSH(h);
    }
    for (int i = 0; i < n; ++i) {
        eh_handle* h = handles[i];
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        unsigned c[3] = {0, 0, 0};
        HIPCHK(h, hipMemcpy(c, h->p2p_ctr, sizeof c, hipMemcpyDeviceToHost));
        if (c[1]) *healthy = 0;
    }
    if (*healthy) return EH_OK;
    if (int rc = eh_p2p_disable(handles[0])) return rc;
    eh_handle* h0 = handles[0];
    const int64_t nt = h0->net.n_theta;
    std::vector<float> th(nt), m(nt), v(nt);
    float bt[2];
    int rc;
    if ((rc = eh_get_params(h0, th.data(), nt)) || (rc = eh_get_opt_state(h0, m.data(), v.data(), nt, bt))) return rc;
    for (int i = 1; i < n; ++i)
        if ((rc = eh_set_params(handles[i], th.data(), nt)) || (rc = eh_set_opt_state(handles[i], m.data(), v.data(), nt, bt))) return rc;
    return EH_OK;
}

// ---- the collective inside the library (RCCL) -------------------------------------------------------------------------
#define NCCLCHK(h, expr)                                                                                     \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return fail(h, EH_ERCCL, "%s: %s", #expr, g_rccl.GetErrorString(r_));            \
    } while (0)

int32_t eh_comm_unique_id(void* id_out, int64_t id_bytes) {
    if (!id_out || id_bytes < (int64_t)sizeof(ncclUniqueId)) return fail(nullptr, EH_EINVAL, "eh_comm_unique_id: buffer of %lld bytes, need %zu", (long long)id_bytes, sizeof(ncclUniqueId));
    RCCL_BIND(nullptr);
    ncclUniqueId id;
    NCCLCHK(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return EH_OK;
}

int32_t eh_comm_init(eh_handle* h, const void* unique_id, int64_t id_bytes, int32_t world, int32_t rank) {
    if (!h || !unique_id) return EH_EINVAL;
    if (id_bytes < (int64_t)sizeof(ncclUniqueId)) return fail(h, EH_EINVAL, "eh_comm_init: id of %lld bytes, need %zu", (long long)id_bytes, sizeof(ncclUniqueId));
    if (world < 1 || rank < 0 || rank >= world) return fail(h, EH_EINVAL, "eh_comm_init: world %d, rank %d", world, rank);
    if (h->comm || h->lgroup) return fail(h, EH_ESTATE, "eh_comm_init: the handle already has a communicator (eh_comm_destroy first)");
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    RCCL_BIND(h);
    if (g_group_depth > 0 && !g_group_rccl) { NCCLCHK(h, g_rccl.GroupStart()); g_group_rccl = true; }      // the bracket was opened before RCCL was in the process
    NCCLCHK(h, g_rccl.CommInitRank(&h->comm, world, id, rank));
    h->comm_world = world; h->comm_rank = rank;
    return EH_OK;
}

// the largest buffer eh_dp_allreduce can be asked for on this handle
static size_t lgroup_floats(const eh_handle* h) {
    size_t n = std::max<size_t>((size_t)h->n_acc, 3 * EH_MAX_TARG);
    if (h->gacc) n = std::max(n, (size_t)EH_GSHARDS * h->n_acc);
    return std::max<size_t>(n, 68);
}

int32_t eh_comm_init_local(eh_handle* const* handles, int32_t n) {
    if (!handles || n < 1 || n > EH_GSHARDS) return fail(nullptr, EH_EINVAL, "eh_comm_init_local: %d handles (1..%d)", n, EH_GSHARDS);
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail(nullptr, EH_EINVAL, "eh_comm_init_local: handle %d is NULL", i);
        for (int j = 0; j < i; ++j) if (handles[j] == handles[i]) return fail(handles[i], EH_EINVAL, "eh_comm_init_local: handle %d is listed twice", i);
        if (handles[i]->comm || handles[i]->lgroup) return fail(handles[i], EH_ESTATE, "eh_comm_init_local: handle %d already has a communicator (eh_comm_destroy first)", i);
        if (handles[i]->net.n_theta != handles[0]->net.n_theta || handles[i]->n_acc != handles[0]->n_acc)
            return fail(handles[i], EH_EINVAL, "eh_comm_init_local: handle %d is a different model (%d parameters, handle 0 has %d)", i, handles[i]->net.n_theta, handles[0]->net.n_theta);
    }
    if (int rc = eh_enable_peer_access(handles, n, "eh_comm_init_local")) return rc;
    EhLocalGroup* g = new EhLocalGroup();
    g->n = n;
    g->cap = lgroup_floats(handles[0]);
    auto undo = [&](eh_handle* h, hipError_t e, const char* what) {
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(handles[i]->device);
            if (g->ready[i]) (void)hipEventDestroy(g->ready[i]);
            if (g->done[i]) (void)hipEventDestroy(g->done[i]);
            (void)hipFree(g->sum[i]);
        }
        delete g;
        return fail(h, e == hipErrorOutOfMemory ? EH_ENOMEM : EH_EHIP, "eh_comm_init_local: %s: %s", what, hipGetErrorString(e));
    };
    for (int i = 0; i < n; ++i) {
        hipError_t e;
        if ((e = hipSetDevice(handles[i]->device)) != hipSuccess) return undo(handles[i], e, "hipSetDevice");
        if ((e = hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming)) != hipSuccess) return undo(handles[i], e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming)) != hipSuccess) return undo(handles[i], e, "hipEventCreate");
        if ((e = hipMalloc(&g->sum[i], g->cap * sizeof(float))) != hipSuccess) return undo(handles[i], e, "hipMalloc");
    }
    for (int i = 0; i < n; ++i) {
        g->m[i] = handles[i];
        handles[i]->lgroup = g; handles[i]->comm_world = n; handles[i]->comm_rank = i;
    }
    return EH_OK;
}

int32_t eh_comm_destroy(eh_handle* h) {
    if (!h) return EH_EINVAL;
    if (h->lgroup) {                            // a local group lives and dies as a whole: every member leaves it
        EhLocalGroup* g = h->lgroup;
        for (int i = 0; i < g->n; ++i) {
            eh_handle* m = g->m[i];
            (void)hipSetDevice(m->device);
            (void)hipStreamSynchronize(m->stream);
        }
        for (int i = 0; i < g->n; ++i) {
            eh_handle* m = g->m[i];
            (void)hipSetDevice(m->device);
            (void)hipEventDestroy(g->ready[i]); (void)hipEventDestroy(g->done[i]); (void)hipFree(g->sum[i]);
            m->lgroup = nullptr; m->comm_world = 0; m->comm_rank = 0;
        }
        for (size_t k = 0; k < g_group_reqs.size();)
            if (g_group_reqs[k].h->lgroup == nullptr) g_group_reqs.erase(g_group_reqs.begin() + k); else ++k;
        delete g;
        (void)hipSetDevice(h->device);
        return EH_OK;
    }
    if (!h->comm) return EH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    NCCLCHK(h, g_rccl.CommDestroy(h->comm));
    h->comm = nullptr; h->comm_world = 0;
    return EH_OK;
}

// the queued all-reduces of one local group, all members present: two event rounds and one small kernel per member
static int lgroup_run(EhLocalGroup* g, const std::vector<EhLocalReq>& reqs) {
    eh_handle* h0 = reqs[0].h;
    const size_t n = reqs[0].n;
    const EhLocalReq* by_rank[EH_GSHARDS] = {nullptr};
    for (const EhLocalReq& r : reqs) {
        if (by_rank[r.h->comm_rank]) return fail(r.h, EH_ESTATE, "eh_comm_group_end: rank %d of the local group asked for two all-reduces in one bracket", r.h->comm_rank);
        if (r.n != n) return fail(r.h, EH_ESTATE, "eh_comm_group_end: the members of the local group ask for different buffers (%zu vs %zu floats)", r.n, n);
        by_rank[r.h->comm_rank] = &r;
    }
    for (int i = 0; i < g->n; ++i)
        if (!by_rank[i]) return fail(h0, EH_ESTATE, "eh_comm_group_end: rank %d of the local group did not call eh_dp_allreduce inside the bracket (every member must)", i);
    if (n > g->cap) return fail(h0, EH_EINVAL, "eh_comm_group_end: %zu floats, the group's scratch holds %zu", n, g->cap);
    EhLocalPtrs src;
    for (int i = 0; i < EH_GSHARDS; ++i) src.p[i] = by_rank[i < g->n ? i : 0]->buf;
    for (int i = 0; i < g->n; ++i) {           // round 1: every member's buffer is complete
        eh_handle* m = g->m[i];
        HIPCHK(m, hipSetDevice(m->device));
        HIPCHK(m, hipEventRecord(g->ready[i], m->stream));
    }
    for (int i = 0; i < g->n; ++i) {
        eh_handle* m = g->m[i];
        HIPCHK(m, hipSetDevice(m->device));
        for (int p = 0; p < g->n; ++p) if (p != i) HIPCHK(m, hipStreamWaitEvent(m->stream, g->ready[p], 0));
        hipLaunchKernelGGL(eh_lgroup_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, src, g->n, (long long)n, g->sum[i]);
        HIPCHK(m, hipGetLastError());
        HIPCHK(m, hipEventRecord(g->done[i], m->stream));
    }
    for (int i = 0; i < g->n; ++i) {           // round 2: nobody reads a buffer any more; the sums replace it
        eh_handle* m = g->m[i];
        HIPCHK(m, hipSetDevice(m->device));
        for (int p = 0; p < g->n; ++p) if (p != i) HIPCHK(m, hipStreamWaitEvent(m->stream, g->done[p], 0));
        HIPCHK(m, hipMemcpyAsync(by_rank[i]->buf, g->sum[i], n * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
    }
    return EH_OK;
}

int32_t eh_comm_group_begin(void) {
    if (g_group_depth++ == 0) {
        g_group_rccl = false;
        g_group_reqs.clear();
        if (g_rccl.so) { NCCLCHK(nullptr, g_rccl.GroupStart()); g_group_rccl = true; }      // (no RCCL in the process yet: eh_comm_init opens the RCCL bracket itself)
    }
    return EH_OK;
}
int32_t eh_comm_group_end(void) {
    if (g_group_depth <= 0) return fail(nullptr, EH_ESTATE, "eh_comm_group_end without eh_comm_group_begin");
    if (--g_group_depth > 0) return EH_OK;
    int rc = EH_OK;
    if (g_group_rccl) {
        g_group_rccl = false;
        ncclResult_t r = g_rccl.GroupEnd();
        if (r != ncclSuccess) rc = fail(nullptr, EH_ERCCL, "ncclGroupEnd: %s", g_rccl.GetErrorString(r));
    }
    std::vector<EhLocalReq> reqs;
    reqs.swap(g_group_reqs);
    while (!reqs.empty() && rc == EH_OK) {
        EhLocalGroup* g = reqs[0].h->lgroup;
        std::vector<EhLocalReq> mine, rest;
        for (const EhLocalReq& r : reqs) (r.h->lgroup == g ? mine : rest).push_back(r);
        rc = lgroup_run(g, mine);
        if (rc != EH_OK) g_create_err = mine[0].h->err.empty() ? g_create_err : mine[0].h->err;
        reqs.swap(rest);
    }
    return rc;
}

int32_t eh_dp_allreduce(eh_handle* h, int32_t which, int32_t index) {
    if (!h) return EH_EINVAL;
    if (!h->comm && !h->lgroup) return fail(h, EH_ESTATE, "eh_dp_allreduce: call eh_comm_init (or eh_comm_init_local) first");
    float* buf = nullptr;
    size_t n = 0;
    switch (which) {
        case EH_BUF_GRAD: buf = h->gradbuf; n = (size_t)h->n_acc; break;
        case EH_BUF_GACC:
            if (index < 0 || index > 2) return fail(h, EH_EINVAL, "eh_dp_allreduce: accumulator %d (0..2)", index);
            if (h->p2p_on) return fail(h, EH_ESTATE, "eh_dp_allreduce: the step kernels exchange their sums themselves (eh_p2p_attach); nothing to reduce");
            if (!h->gacc) return fail(h, EH_ESTATE, "eh_dp_allreduce: this model has no fused_update accumulators");
            n = (size_t)EH_GSHARDS * h->n_acc; buf = h->gacc + (size_t)index * n; break;
        case EH_BUF_BNSTAT:
            if (!h->bn_on) return fail(h, EH_ESTATE, "eh_dp_allreduce: the model has no input BatchNorm");
            buf = h->bn_stat; n = 65; break;
        case EH_BUF_TCOUNT: buf = h->tcount; n = 3 * EH_MAX_TARG; break;
        case EH_BUF_MOMENT: buf = h->mombuf; n = EH_MAX_TARG * EH_EVAL_STATS; break;
        default: return fail(h, EH_EINVAL, "eh_dp_allreduce: buffer %d (EH_BUF_GRAD, EH_BUF_GACC, EH_BUF_BNSTAT, EH_BUF_TCOUNT or EH_BUF_MOMENT)", which);
    }
    if (h->lgroup) {
        if (h->lgroup->n == 1) return EH_OK;               // a world of one: the sum is the buffer
        if (g_group_depth <= 0) return fail(h, EH_ESTATE, "eh_dp_allreduce: a local group's members meet at eh_comm_group_end: bracket the calls of all members with eh_comm_group_begin / eh_comm_group_end");
        g_group_reqs.push_back({h, buf, n});
        return EH_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    NCCLCHK(h, g_rccl.AllReduce(buf, buf, n, ncclFloat, ncclSum, h->comm, h->stream));
    return EH_OK;
}

int32_t eh_dp_train_step(eh_handle* h, int64_t first, int64_t count, float* loss_out) {
    if (!h) return EH_EINVAL;
    if (!h->comm && !h->lgroup) return fail(h, EH_ESTATE, "eh_dp_train_step: call eh_comm_init first");
    if (h->lgroup && h->lgroup->n > 1) return fail(h, EH_ESTATE, "eh_dp_train_step: the members of a local group step together: eh_dp_train_step_group");
    int rc;
    if (h->bn_on) {
        if ((rc = eh_dp_bn_stats(h, first, count))) return rc;
        if ((rc = eh_dp_allreduce(h, EH_BUF_BNSTAT, 0))) return rc;
    }
    if (h->fused && h->net.T == 1) {
        if (loss_out) return fail(h, EH_EINVAL, "eh_dp_train_step: fused_update mode reports no per-step loss (pass NULL)");
        int32_t k = 0;
        if ((rc = eh_dp_fused_step(h, first, count, &k))) return rc;
        return k >= 0 ? eh_dp_allreduce(h, EH_BUF_GACC, k) : EH_OK;
    }
    if (eh_two_pass_mask(h)) {           // the moments of the GLOBAL batch's predictions, two small exchanges ahead of the pass (eh_dp_moments)
        for (int stage = 0; stage < 2; ++stage) {
            if ((rc = eh_dp_moments(h, first, count, stage))) return rc;
            if ((rc = eh_dp_allreduce(h, EH_BUF_MOMENT, 0))) return rc;
        }
    }
    if (h->net.T != 1) {
        if ((rc = eh_dp_counts(h, first, count))) return rc;
        if ((rc = eh_dp_allreduce(h, EH_BUF_TCOUNT, 0))) return rc;
    }
    if ((rc = eh_dp_grad(h, first, count))) return rc;
    if ((rc = eh_dp_allreduce(h, EH_BUF_GRAD, 0))) return rc;
    return eh_dp_apply(h, loss_out);
}

// One host thread, several handles (one per device): a whole data-parallel step of all of them.  Every phase is issued to all
// members before the exchange that follows it, and every exchange sits in one eh_comm_group_begin / eh_comm_group_end bracket
// (RCCL communicators: ncclGroupStart / End as RCCL requires of a single thread; local groups: the members meet at the end).
int32_t eh_dp_train_step_group(eh_handle* const* hs, int32_t n, const int64_t* first, int64_t count, float* loss_out) {
    if (!hs || !first || n < 1 || n > EH_GSHARDS) return fail(nullptr, EH_EINVAL, "eh_dp_train_step_group: %d handles (1..%d)", n, EH_GSHARDS);
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) return fail(nullptr, EH_EINVAL, "eh_dp_train_step_group: handle %d is NULL", i);
        if (!hs[i]->comm && !hs[i]->lgroup) return fail(hs[i], EH_ESTATE, "eh_dp_train_step_group: handle %d has no communicator (eh_comm_init / eh_comm_init_local)", i);
        if (hs[i]->fused != hs[0]->fused || hs[i]->net.T != hs[0]->net.T || hs[i]->bn_on != hs[0]->bn_on || hs[i]->p2p_on != hs[0]->p2p_on)
            return fail(hs[i], EH_EINVAL, "eh_dp_train_step_group: handle %d runs a different step mode than handle 0", i);
        // the list must be exactly one communicator's membership in rank order -- found here, not at the first exchange, when BatchNorm
        // sums / counts / gradient kernels of some members have already advanced their state
        if (hs[0]->lgroup) {
            if (hs[i]->lgroup != hs[0]->lgroup || n != hs[0]->lgroup->n || hs[i]->comm_rank != i)
                return fail(hs[i], EH_EINVAL, "eh_dp_train_step_group: the handles must be all %d members of ONE local group in rank order (handle %d is %s)", hs[0]->lgroup->n, i,
                            hs[i]->lgroup != hs[0]->lgroup ? "not in handle 0's group" : "out of order");
        } else if (!hs[i]->comm) {
            return fail(hs[i], EH_EINVAL, "eh_dp_train_step_group: handle 0 has an RCCL communicator, handle %d a local group: one kind per call", i);
        }
        if (hs[0]->p2p_on && hs[0]->p2p_local && (hs[i]->p2p_group[i] != hs[i] || hs[i]->p2p_group[0] != hs[0] || hs[i]->p2p_world != n))
            return fail(hs[i], EH_EINVAL, "eh_dp_train_step_group: the handles must be all %d members of ONE local peer-to-peer group in rank order", hs[0]->p2p_world);
    }
    int32_t k[EH_GSHARDS] = {0};
    auto exchange = [&](int which, bool per_handle_index) -> int {
        int rc = eh_comm_group_begin();
        if (rc) return rc;
        for (int i = 0; i < n && !rc; ++i) rc = eh_dp_allreduce(hs[i], which, per_handle_index ? k[i] : 0);
        const int rc2 = eh_comm_group_end();                // (always closes the bracket)
        return rc ? rc : rc2;
    };
    int rc;
    eh_handle* h0 = hs[0];
    if (h0->bn_on) {
        for (int i = 0; i < n; ++i) if ((rc = eh_dp_bn_stats(hs[i], first[i], count))) return rc;
        if ((rc = exchange(EH_BUF_BNSTAT, false))) return rc;
    }
    if (h0->fused && h0->net.T == 1) {
        if (loss_out) return fail(h0, EH_EINVAL, "eh_dp_train_step_group: fused_update mode reports no per-step loss (pass NULL)");
        for (int i = 0; i < n; ++i) if ((rc = eh_dp_fused_step(hs[i], first[i], count, &k[i]))) return rc;
        return k[0] >= 0 ? exchange(EH_BUF_GACC, true) : EH_OK;
    }
    if (eh_two_pass_mask(h0)) {
        for (int stage = 0; stage < 2; ++stage) {
            for (int i = 0; i < n; ++i) if ((rc = eh_dp_moments(hs[i], first[i], count, stage))) return rc;
            if ((rc = exchange(EH_BUF_MOMENT, false))) return rc;
        }
    }
    if (h0->net.T != 1) {
        for (int i = 0; i < n; ++i) if ((rc = eh_dp_counts(hs[i], first[i], count))) return rc;
        if ((rc = exchange(EH_BUF_TCOUNT, false))) return rc;
    }
    for (int i = 0; i < n; ++i) if ((rc = eh_dp_grad(hs[i], first[i], count))) return rc;
    if ((rc = exchange(EH_BUF_GRAD, false))) return rc;
    for (int i = 0; i < n; ++i) if ((rc = eh_dp_apply(hs[i], i == 0 ? loss_out : nullptr))) return rc;
    return EH_OK;
}

}   // extern "C"

void eh_comm_release(eh_handle* h) {
    (void)hipSetDevice(h->device);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
    if (h->lgroup) (void)eh_comm_destroy(h);
    if (h->p2p_alloc) (void)eh_p2p_disable(h);          // (a member of a local peer-to-peer group takes the group along)
}
