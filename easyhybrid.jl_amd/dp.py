"""Data-parallel training over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-GPU code at all; samples are independent through
the NN, the mechanistic model and the per-sample loss terms, so the path shards over samples.  The
only exchange per step is ONE sum all-reduce of

    [ grad_unnormalised (n_theta) | sum_i m_i (yhat_i - y_i)^2 | n_valid ]        (n_theta + 2 floats)

after which every rank divides by the GLOBAL valid count -- the mean the reference takes over the
whole batch (src/losses/loss_fn.jl:61-63) -- and applies the same optimiser update to its replica.
Shards never exchange per-shard means (they would weight shards with more NaN targets wrongly).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous sample range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_partials(buf, group=None):
    """SUM all-reduce of the raw partial vector (torch tensor, CPU or GPU), in place."""
    import torch.distributed as dist
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def normalise(buf, n_theta: int, kind: str = "mse"):
    """[grad | S | n | Sy | Syy] raw sums -> (gradient of the loss, loss, n); n == 0 -> (zeros, nan, 0):
    batch skipped.  Same arithmetic as eh_loss_finish in csrc/eh_device.hpp."""
    n = float(buf[n_theta + 1])
    if n <= 0:
        return buf[:n_theta] * 0, float("nan"), 0.0
    S = float(buf[n_theta])
    if kind == "rmse":
        loss = (S / n) ** 0.5
        return buf[:n_theta] / (2 * n * loss), loss, n
    if kind == "nseLoss":
        D = float(buf[n_theta + 3]) - float(buf[n_theta + 2]) ** 2 / n
        return buf[:n_theta] / D, S / D, n
    return buf[:n_theta] / n, S / n, n


def target_weights(tcount, kinds):
    """per-target sums [n_t | sum (y-c) | sum (y-c)^2] (T x 3) of the GLOBAL batch (EH_BUF_TCOUNT summed over the ranks) -> the
    weights w_t the residual terms of target t enter the loss with: 1 / n_t (mse, mae), 1 / sum (y - ybar)^2 (nseLoss).
    Same arithmetic as eh_weights_from_counts_kernel (csrc/eh_api.hip)."""
    import numpy as np
    tc = np.asarray(tcount, np.float64).reshape(-1, 3)
    kinds = [kinds] * len(tc) if isinstance(kinds, str) else list(kinds)
    w = np.zeros(len(tc))
    for t, (n, s1, s2) in enumerate(tc):
        if n > 0:
            w[t] = 1.0 / (s2 - s1 * s1 / n) if kinds[t] == "nseLoss" else 1.0 / n
    return w


def bn_moments(stat, shift):
    """[sum (x-c) (32) | sum (x-c)^2 (32) | n] summed over the ranks -> (mean, biased variance) per predictor.
    Same arithmetic as the BatchNorm prologue of the step kernel (csrc/eh_device.hpp)."""
    P = len(shift)
    n = float(stat[64])
    d = stat[:P] / n
    var = (stat[32:32 + P] / n - d * d).clip(0)
    return shift + d, var


class _DevArray:
    """__cuda_array_interface__ view of a library-owned device buffer (no copy, no torch types in the ABI)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class DataParallel:
    """Drives one replica.  `engine` already holds this rank's shard as its train split."""

    def __init__(self, engine, group=None, fused: bool = True, p2p="auto", specialize: bool = False, debug_fail_selftest_rank=None):
        """fused=True: one kernel + one exchange per step (the update of step s is applied in the
        prologue of step s+1; `engine.synchronize()` applies the last one).  The exchange is the
        engine's own peer-to-peer store protocol over xGMI when `p2p` is on and its start-up self-test
        passes on every rank (no collective call per step at all), otherwise one RCCL all-reduce.
        fused=False: step kernel, deterministic reduction, all-reduce, optimiser kernel.
        p2p="prologue" (publishing mode 1): a step's sums leave this rank from the NEXT kernel on its stream, which then waits for the
        peers' -- every draining call (`synchronize`, `get_params`, `evaluate`, `forward`, a loss read, `set_option`) is a collective
        then and has to be made on ALL ranks at the same step; a one-sided drain runs into the 2 s deadline."""
        import os
        import numpy as np
        import torch
        from . import _lib as L
        self.engine, self.group, self.fused = engine, group, fused
        self._debug_fail_rank = debug_fail_selftest_rank            # tests only: this rank reports a failed self-test (exercises the fallback negotiation)
        dev = torch.device("cuda", torch.cuda.current_device())
        self._dev = dev
        ptr, n = engine.device_buffer(L.EH_BUF_GRAD)
        self.buf = torch.as_tensor(_DevArray(ptr, n), device=dev)
        self.p2p = False
        self.p2p_mode = 0
        self.p2p_report = None                              # set by the negotiation: how far it got on every rank
        if fused and int(engine.desc.n_targets) > 1:        # the per-target weights need the GLOBAL counts before the pass: eh_dp_counts + three-kernel path
            fused = self.fused = False
        if fused:
            try:
                engine.set_option("fused_update", 1)
            except NotImplementedError:                     # hidden widths above 64: three-kernel path
                fused = self.fused = False
        if fused:
            # p2p = "prologue": the peer-to-peer exchange with the sums published by the NEXT kernel's first workgroup instead of the step's
            # elected last one (engine option "p2p_mode" = 1; csrc/eh_device.hpp EhP2P::mode); True / "auto": the elected form; calibrate()
            # times both next to the collective and keeps the fastest
            self.p2p_mode = 1 if p2p == "prologue" else int(os.environ.get("EH_DP_P2P_MODE", "0"))
            if p2p == "auto":
                p2p = os.environ.get("EH_DP_P2P", "1") != "0"
            if p2p:
                self.p2p = self._negotiate_p2p()
                if self.p2p and self.p2p_mode:
                    engine.set_option("p2p_mode", self.p2p_mode)
            gptr, gn = engine.device_buffer(L.EH_BUF_GACC)      # (after the negotiation: it re-allocates the accumulators)
            self.gacc = [torch.as_tensor(_DevArray(gptr + 4 * k * (gn // 3), gn // 3), device=dev) for k in range(3)]
        # run the engine on torch's current stream so kernels and the collective are ordered
        engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.bn = bool(engine.desc.input_batchnorm)
        if self.bn:
            # input BatchNorm normalises with the statistics of the GLOBAL minibatch: a second, 65-float
            # all-reduce per step ahead of the step kernel.  Every rank shifts its sums by the same
            # vector (the mean of the whole training set) so they can be added.
            bptr, bn_n = engine.device_buffer(L.EH_BUF_BNSTAT)
            self.bnbuf = torch.as_tensor(_DevArray(bptr, bn_n), device=dev)
            sx, n = engine.x_sum.get(L.EH_SPLIT_TRAIN, (None, 0))
            if sx is None:
                raise RuntimeError("DataParallel with input BatchNorm needs the train split uploaded through set_data first")
            tot = torch.tensor(list(sx) + [float(n)], dtype=torch.float64, device=dev)
            allreduce_partials(tot, group)
            engine.set_bn_shift((tot[:-1] / tot[-1]).cpu().numpy())
        # the shifted target sums (nseLoss normaliser; per-target weights of multi-target models) are added across ranks: every
        # rank shifts by the same vector, the mean of each target over the global training set
        self.multi = int(engine.desc.n_targets) > 1
        ys = getattr(engine, "y_sum", {}).get(L.EH_SPLIT_TRAIN)
        if ys is not None:
            tot = torch.tensor(np.asarray(ys, np.float64).reshape(-1), dtype=torch.float64, device=dev)
            allreduce_partials(tot, group)
            tot = tot.cpu().numpy().reshape(-1, 2)
            engine.set_target_shift(np.where(tot[:, 1] > 0, tot[:, 0] / np.maximum(tot[:, 1], 1), 0.0))
        elif self.multi or engine.two_pass_loss:
            # (advisor, round 4: a single-target pearson / kge engine fed through set_data_device kept its shard's own shift -- the mean of
            #  its first 4 096 samples -- and the all-reduced moments then mixed different centres: a silently wrong gradient)
            raise RuntimeError("DataParallel with a multi-target model or a two-pass training loss (pearson / kge / pbkge) needs the train split "
                               "uploaded through set_data first: the shifted sums are added across ranks about ONE common shift")
        if self.multi:
            cptr, cn = engine.device_buffer(L.EH_BUF_TCOUNT)
            self.cbuf = torch.as_tensor(_DevArray(cptr, cn), device=dev)
        # two-pass training losses (pearson / kge / pbkge; rmse on a multi-target model): the moments of the GLOBAL batch's predictions
        # go round twice ahead of the pass (32 floats each: eh_dp_moments)
        mptr, mn = engine.device_buffer(L.EH_BUF_MOMENT)
        self.mbuf = torch.as_tensor(_DevArray(mptr, mn), device=dev)
        if specialize:
            engine.set_option("specialize", 1)
            self.prepare()

    def prepare(self):
        """Compile the run-time kernels of this configuration now, on every rank, before any step: a forward over one sample
        builds train / eval / cross-GPU kernels in one go.  (A peer that is still compiling while the others step would run
        them into the deadline of the peer-to-peer exchange.)"""
        import torch.distributed as dist
        from . import _lib as L
        self.engine.forward(L.EH_SPLIT_TRAIN, 0, 1, params=False)
        self.engine.synchronize()
        dist.barrier(group=self.group)

    def _all_agree(self, ok: bool) -> bool:
        import torch
        import torch.distributed as dist
        backend = dist.get_backend(self.group)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _negotiate_p2p(self) -> bool:
        """Export / map the receive buffers and run the self-test; every rank ends with the same answer.  `self.p2p_report` says, rank by
        rank, how far the negotiation got and why it stopped (what bench.py --gpus N prints: on real multi-GPU hardware the first run has
        to tell which of IPC export, IPC mapping of every peer's buffer, or the store / load self-test over xGMI is the one that failed)."""
        import os
        import torch.distributed as dist
        eng = self.engine
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        info = {"rank": rank, "export": None, "attach": None, "selftest": None, "error": None}

        def done(enabled, stage):
            infos = [None] * world
            dist.all_gather_object(infos, info, group=self.group)
            self.p2p_report = {"enabled": enabled, "stopped_at": None if enabled else stage, "world": world,
                               "stages": "export = uncached receive buffer + IPC handle; attach = every peer's buffer mapped (world - 1 IPC attachments per rank); "
                                         "selftest = 8 exchange rounds, every word of every peer checked", "ranks": infos}
            return enabled
        if world < 2 or world > 8:
            self.p2p_report = {"enabled": False, "stopped_at": "world size %d (2..8)" % world, "world": world, "ranks": []}
            return False
        try:
            handle = eng.p2p_init(world, rank)
            info["export"] = True
        except Exception as e:                              # no uncached memory / IPC on this system
            handle = None
            info["export"], info["error"] = False, repr(e)[:200]
        if not self._all_agree(handle is not None):
            eng.p2p_disable()
            return done(False, "export")
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=self.group)
        try:
            eng.p2p_attach(handles)
            ok = True
        except Exception as e:
            ok = False
            info["error"] = repr(e)[:200]
        info["attach"] = ok
        if not self._all_agree(ok):
            eng.p2p_disable()
            return done(False, "attach")
        dist.barrier(group=self.group)                      # every rank has every buffer mapped before anyone stores
        try:
            ok = eng.p2p_selftest(8)
        except Exception as e:
            ok = False
            info["error"] = repr(e)[:200]
        if self._debug_fail_rank is not None and int(self._debug_fail_rank) == rank:
            ok = False
            info["error"] = "debug_fail_selftest_rank"
        info["selftest"] = bool(ok)
        if not self._all_agree(ok):
            dist.barrier(group=self.group)                  # nobody unmaps while a peer's test kernel may still store
            eng.p2p_disable()
            return done(False, "selftest")
        return done(True, None)

    def _refresh_gacc(self):
        import torch
        from . import _lib as L
        gptr, gn = self.engine.device_buffer(L.EH_BUF_GACC)
        self.gacc = [torch.as_tensor(_DevArray(gptr + 4 * k * (gn // 3), gn // 3), device=self._dev) for k in range(3)]

    def calibrate(self, first: int, count: int, nsteps: int = 300) -> dict:
        """Collective: time `nsteps` training steps on samples [first, first+count) with the peer-to-peer exchange
        and with the RCCL all-reduce and keep the faster one (measure, don't guess: the store latency over xGMI
        and the collective's launch cost both depend on the node).  Trains the model by 2 x nsteps steps."""
        import time
        import torch
        import torch.distributed as dist
        if not self.p2p:
            return {"p2p_us": None, "p2p_prologue_us": None, "collective_us": None, "chosen": "collective"}

        def timed():
            # every rank runs the same sequence of collectives whatever happens locally (an exception on one rank
            # must not leave its peers inside a barrier)
            ok = True
            try:
                for _ in range(20):
                    self.step(first, count)
                self.engine.synchronize()
            except Exception:
                ok = False
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            try:
                for _ in range(nsteps):
                    self.step(first, count)
                self.engine.synchronize()
            except Exception:
                ok = False
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                             device=self._dev if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return 1e6 * float(t.item()) / nsteps, ok
        # three exchanges on this node, same steps each: the peer-to-peer form with the election in the step's epilogue, the one published
        # from the next kernel's prologue, the collective.  (Every rank switches modes at the same point: between two drained steps.)
        t_modes, ok = {}, True
        for mode in (0, 1):
            dist.barrier(group=self.group)
            self.engine.set_option("p2p_mode", mode)
            t_modes[mode], ok1 = timed()
            ok = ok and ok1
        ok = self._all_agree(ok)
        dist.barrier(group=self.group)
        self.engine.p2p_disable()
        self.p2p = False
        self._refresh_gacc()
        if not ok:
            self.broadcast_state(0)
            return {"p2p_us": None, "p2p_prologue_us": None, "collective_us": None, "chosen": "collective (peer-to-peer exchange failed)"}
        t_col, _ = timed()
        best_mode = 0 if t_modes[0] <= t_modes[1] else 1
        if t_modes[best_mode] < t_col:
            self.p2p = self._negotiate_p2p()
            self._refresh_gacc()
            if self.p2p:
                self.p2p_mode = best_mode
                self.engine.set_option("p2p_mode", best_mode)
        return {"p2p_us": t_modes[0], "p2p_prologue_us": t_modes[1], "collective_us": t_col,
                "chosen": ("p2p, published from the next prologue" if best_mode else "p2p, elected publisher") if self.p2p else "collective"}

    def check(self) -> bool:
        """Collective: drain the engine and make sure no peer-to-peer wait ran into its deadline anywhere.
        If one did, every rank drops back to the RCCL all-reduce, rank 0's parameters are re-broadcast and
        False is returned (the steps since the last check are then not trustworthy)."""
        import torch
        import torch.distributed as dist
        from . import _lib as L
        try:
            self.engine.synchronize()
            ok = True
        except Exception:
            ok = False
        if not self.p2p:
            return ok
        if self._all_agree(ok):
            return True
        dist.barrier(group=self.group)
        self.engine.p2p_disable()
        self.p2p = False
        self._refresh_gacc()
        self.broadcast_state(0)
        return False

    def step(self, first: int, count: int, want_loss: bool = False):
        if self.bn:
            self.engine.dp_bn_stats(first, count)
            allreduce_partials(self.bnbuf, self.group)
        if self.fused and not want_loss:
            k = self.engine.dp_fused_step(first, count)
            if k >= 0:                                        # k < 0: the kernels exchange the sums themselves (p2p)
                allreduce_partials(self.gacc[k], self.group)  # 8 shards x (n_theta + 2) raw sums
            return None
        if self.engine.two_pass_loss:                         # moments of the GLOBAL batch's predictions: about the common shift, then about their global mean
            for stage in (0, 1):
                self.engine.dp_moments(first, count, stage)
                allreduce_partials(self.mbuf, self.group)
        if self.multi:                                        # per-target normalisers of the GLOBAL batch ahead of the pass (12 floats)
            self.engine.dp_counts(first, count)
            allreduce_partials(self.cbuf, self.group)
        self.engine.dp_grad(first, count)
        allreduce_partials(self.buf, self.group)
        return self.engine.dp_apply(want_loss)

    def broadcast_params(self, src: int = 0):
        import torch
        import torch.distributed as dist
        from . import _lib as L
        ptr, n = self.engine.device_buffer(L.EH_BUF_THETA)
        t = torch.as_tensor(_DevArray(ptr, n), device=self.buf.device)
        dist.broadcast(t, src=src, group=self.group)

    def broadcast_state(self, src: int = 0):
        """Collective: every replica takes rank `src`'s parameters AND optimiser state (moments, running beta products) -- after a
        missed exchange a rank substitutes zeros for what never arrived, so all of them can differ, and replicas that only agreed
        on theta would drift apart again with the next update.  Goes through the host API (get / set): the engine applies any
        pending update first and refreshes the parameter image the step kernels read."""
        import numpy as np
        import torch
        import torch.distributed as dist
        eng = self.engine
        theta = eng.get_params()
        m, v, bt = eng.get_opt_state()
        pack = np.concatenate([theta, m, v, np.asarray(bt, np.float32)]).astype(np.float32)
        backend = dist.get_backend(self.group)
        t = torch.from_numpy(pack).to(self._dev if backend == "nccl" else "cpu")
        dist.broadcast(t, src=src, group=self.group)
        pack = t.cpu().numpy()
        n = theta.size
        eng.set_params(pack[:n])
        eng.set_opt_state(pack[n:2 * n], pack[2 * n:3 * n], pack[3 * n:3 * n + 2])
        if self.fused:
            self._refresh_gacc()
