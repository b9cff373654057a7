"""The PPO update with its non-GEMM part in hand-written HIP and the whole epoch replayed as one hipGraph
(include/openroborl_learner.h, csrc/orr_learner.hip).

Same arithmetic as ppo.PPO.update (the plain PyTorch restatement of agents/ppo_imitation.py:156-258 + Adam, which stays the
fp32 reference the tests compare this against), different execution:
  * parameters, gradients and Adam moments live in ONE flat buffer each (the model's tensors become views of it): the gradient
    all-reduce needs no packing and Adam is one launch over 434 k floats;
  * forward = six GEMMs with bias(+ReLU) epilogues writing into static activations; backward = ten GEMMs written out by hand
    (no autograd graph) with the ReLU masks, the gradient through the two output layers (fan-out 12 and 1: no GEMM), every bias
    gradient and the loss head in a handful of HIP launches (7 per minibatch) - autograd spends ~70 elementwise / reduction launches per
    minibatch on the same work, more GPU time than the GEMMs;
  * one rank: an epoch (gathers of all minibatches, forward, backward, Adam) is captured once and replayed - ~30 launches per
    minibatch cost no host time; several ranks: one graph per minibatch, the all-reduce (RCCL) and Adam between replays.
    ORR_FORCE_DIST=1 with an initialised process group takes the several-ranks path even for ONE rank (a one-GPU box can then run
    per-minibatch graphs + the RCCL all-reduce + Adam between replays; the result equals the one-graph path bit for bit).
Scalar hyper-parameters (lr, betas, eps, clip, vf_coef, the policy's std, the world size) are by-value kernel arguments, i.e. frozen
into a captured graph: update() compares them with the values the graphs were captured with and captures again when one changed
(the reference feeds optim_stepsize * cur_lrmult every step: ppo1/pposgd_simple.py; run.py uses schedule='constant').
There is no fallback: without a GPU and the HIP library the constructor raises.
"""
import ctypes as C
import os

from . import _abi, _lib

NETS = ("pi", "vf")


class FusedPPO(object):
    def __init__(self, model, clip_param=0.2, lr=1e-5, adam_eps=1e-5, minibatch=4096, vf_coef=1.0, group=None, betas=(0.9, 0.999),
                 mpi_adam_epsilon=False, use_graph=True):
        import torch
        self.torch = torch
        self.model = model
        self.dev = model.device
        if self.dev.type != "cuda":
            raise RuntimeError("FusedPPO needs a GPU device (the torch reference is ppo.PPO)")
        self.L = _lib.load()
        self.clip, self.lr, self.eps, self.vf_coef = float(clip_param), float(lr), float(adam_eps), float(vf_coef)
        self.b1, self.b2 = float(betas[0]), float(betas[1])
        self.minibatch = int(minibatch)
        self.group = group
        self.adam_flags = 1 if mpi_adam_epsilon else 0          # ORR_ADAM_MPI_EPSILON
        self.use_graph = bool(use_graph)
        # ---- one flat buffer for the parameters (16-byte aligned slots), the model's tensors become views of it ----
        names = sorted(model.p)
        self.slots, off = {}, 0
        for k in names:
            n = model.p[k].numel()
            self.slots[k] = (off, n, tuple(model.p[k].shape))
            off += (n + 3) // 4 * 4
        self.total = off
        z = lambda: torch.zeros(self.total, dtype=torch.float32, device=self.dev)   # noqa: E731
        self.flat_p, self.flat_g, self.m, self.v = z(), z(), z(), z()
        self.state = torch.zeros(2, dtype=torch.int32, device=self.dev)
        self.w, self.g = {}, {}
        with torch.no_grad():
            for k in names:
                o, n, shape = self.slots[k]
                self.w[k] = self.flat_p[o:o + n].view(shape)
                self.w[k].copy_(model.p[k].detach())
                self.g[k] = self.flat_g[o:o + n].view(shape)
                model.p[k] = self.w[k].detach().requires_grad_(True)        # same storage: the torch path and the zip export keep working
        if hasattr(model, "mark_updated"):
            model.mark_updated()
        self._plans = {}

    # ---- launches ------------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def _several(self):
        """The several-ranks execution (per-minibatch graphs, all-reduce and Adam between replays): more than one rank, or
        ORR_FORCE_DIST=1 with a process group (one-rank rehearsal of exactly that path, e.g. on RCCL with the one GPU of a test box)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or (os.environ.get("ORR_FORCE_DIST", "0") == "1")

    def _graph_key(self, several):
        return (self.lr, self.b1, self.b2, self.eps, self.clip, self.vf_coef, float(self.model.std), self.adam_flags, self._world(), several)

    @staticmethod
    def fit_minibatch(num_samples, minibatch):
        """Largest minibatch size <= `minibatch` that divides `num_samples` (update() needs equal minibatches: static buffers)."""
        m = max(1, min(int(minibatch), int(num_samples)))
        while num_samples % m:
            m -= 1
        return m

    def _adam(self, world):
        _lib.check(self.L.orr_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.total,
                                        self.lr, self.b1, self.b2, self.eps, 1.0 / world, self.adam_flags, self.state.data_ptr(),
                                        self._stream()), self.L)

    def _wgrad(self, x, g, part, out):
        """out = x^T g.  Tall batches: 16 partial products in one batched GEMM (see ppo._wgrad: the library's plain GEMM fills a
        third of the chip for these K = batch shapes); their sum is one of the jobs of the minibatch's orr_colsum_finish."""
        t = self.torch
        if part is not None:
            M = x.shape[0]
            t.bmm(x.view(16, M // 16, x.shape[1]).transpose(1, 2), g.view(16, M // 16, g.shape[1]), out=part)
        else:
            t.mm(x.t(), g, out=out)

    def _plan(self, B, M):
        """Static buffers for B samples per update in minibatches of M."""
        key = (B, M)
        if key in self._plans:
            return self._plans[key]
        t = self.torch
        f = lambda *shape: t.empty(shape, dtype=t.float32, device=self.dev)   # noqa: E731
        split = M >= 4096 and M % 16 == 0
        P = {"B": B, "M": M, "nmb": B // M, "obs": f(B, 160), "aux": f(B, 16), "perm": t.empty(B, dtype=t.int64, device=self.dev),
             "x": f(M, 160), "batch": f(M, 16), "gz2": f(M, 256), "gh1": f(M, 512), "g_pi": f(M, 12), "g_vf": f(M),
             "stats": t.zeros(B // M, 2, dtype=t.float32, device=self.dev),
             "ws": f(int(self.L.orr_learner_workspace_floats(M, 16))), "graphs": None}
        P["aux"].zero_()
        # the column sums behind the six bias gradients and the two output-layer weight gradients of a minibatch are deferred: each
        # producer leaves its per-workgroup partial sums in its own buffer and ONE launch adds them all up (orr_colsum_finish)
        rows = int(self.L.orr_learner_partial_rows(M))
        jobs = (_abi.OrrColsumJob * 10)()
        for i, (net, k) in enumerate((("pi", 12), ("vf", 1))):
            P["h1_" + net], P["h2_" + net], P["y_" + net] = f(M, 512), f(M, 256), f(M, k)
            P["part1_" + net], P["part0_" + net] = (f(16, 512, 256), f(16, 160, 512)) if split else (None, None)
            if split:
                jobs[6 + 2 * i] = _abi.OrrColsumJob(P["part1_" + net].data_ptr(), self.g["model/%s_fc1/w:0" % net].data_ptr(), 16, 512 * 256)
                jobs[7 + 2 * i] = _abi.OrrColsumJob(P["part0_" + net].data_ptr(), self.g["model/%s_fc0/w:0" % net].data_ptr(), 16, 160 * 512)
            P["ws2_" + net] = f(int(self.L.orr_learner_workspace_floats(M, 256)))      # [rows][256] | [rows][256 * 12]
            P["ws1_" + net] = f(rows * 512)
            for j, (buf, off, out, cols) in enumerate(((P["ws2_" + net], 0, self.g["model/%s_fc1/b:0" % net], 256),
                                                       (P["ws2_" + net], rows * 256, self.g["model/%s/w:0" % net], 256 * k),
                                                       (P["ws1_" + net], 0, self.g["model/%s_fc0/b:0" % net], 512))):
                jobs[3 * i + j] = _abi.OrrColsumJob(buf.data_ptr() + 4 * off, out.data_ptr(), rows, cols)
        P["jobs"], P["n_jobs"] = jobs, 10 if split else 6
        self._plans[key] = P
        return P

    def _minibatch(self, P, s):
        """Gather, forward, loss head and backward of minibatch s of the current permutation; gradients land in flat_g."""
        t, L, M, w, g = self.torch, self.L, P["M"], self.w, self.g
        st, ws = self._stream(), P["ws"].data_ptr()
        idx = P["perm"][s * M:(s + 1) * M]
        t.index_select(P["obs"], 0, idx, out=P["x"])
        t.index_select(P["aux"], 0, idx, out=P["batch"])
        x = P["x"]
        for net in NETS:
            t._addmm_activation(w["model/%s_fc0/b:0" % net], x, w["model/%s_fc0/w:0" % net], out=P["h1_" + net])
            t._addmm_activation(w["model/%s_fc1/b:0" % net], P["h1_" + net], w["model/%s_fc1/w:0" % net], out=P["h2_" + net])
            t.addmm(w["model/%s/b:0" % net], P["h2_" + net], w["model/%s/w:0" % net], out=P["y_" + net])
        _lib.check(L.orr_ppo_head(P["y_pi"].data_ptr(), P["y_vf"].data_ptr(), P["batch"].data_ptr(), M, float(self.model.std), self.clip,
                                  self.vf_coef, P["g_pi"].data_ptr(), P["g_vf"].data_ptr(), g["model/pi/b:0"].data_ptr(),
                                  g["model/vf/b:0"].data_ptr(), P["stats"][s].data_ptr(), ws, st), L)
        for net, k, gy in (("pi", 12, P["g_pi"]), ("vf", 1, P["g_vf"])):
            h1, h2 = P["h1_" + net], P["h2_" + net]
            _lib.check(L.orr_head_backward(gy.data_ptr(), k, w["model/%s/w:0" % net].data_ptr(), h2.data_ptr(), M, 256, P["gz2"].data_ptr(),
                                           None, None, P["ws2_" + net].data_ptr(), st), L)
            self._wgrad(h1, P["gz2"], P["part1_" + net], g["model/%s_fc1/w:0" % net])
            t.mm(P["gz2"], w["model/%s_fc1/w:0" % net].t(), out=P["gh1"])
            _lib.check(L.orr_relu_backward(P["gh1"].data_ptr(), h1.data_ptr(), M, 512, None, P["ws1_" + net].data_ptr(), st), L)
            self._wgrad(x, P["gh1"], P["part0_" + net], g["model/%s_fc0/w:0" % net])
        _lib.check(L.orr_colsum_finish(P["jobs"], P["n_jobs"], st), L)

    def _allreduce(self):
        """Sum of the flat gradient over the ranks; orr_adam_step divides by the world size (mpi_adam.py:51-53)."""
        import torch.distributed as dist
        if dist.get_backend(self.group) == "gloo":            # rehearsal of the multi-rank path on a one-GPU box: stage through the host
            tmp = self.flat_g.cpu()
            dist.all_reduce(tmp, group=self.group)
            self.flat_g.copy_(tmp)
        else:
            dist.all_reduce(self.flat_g, group=self.group)   # RCCL, in place on the device

    def _epoch_eager(self, P, world, several):
        for s in range(P["nmb"]):
            self._minibatch(P, s)
            if several:
                self._allreduce()
            self._adam(world)

    def _capture(self, P, several):
        """One rank: one graph for the whole epoch.  Several ranks: one graph per minibatch (the collective runs between replays)."""
        t = self.torch
        keep = [x.clone() for x in (self.flat_p, self.m, self.v, self.state)]
        P["perm"].copy_(t.arange(P["B"], device=self.dev))
        side = t.cuda.Stream(device=self.dev)
        side.wait_stream(t.cuda.current_stream(self.dev))
        with t.cuda.stream(side):                    # warm-up outside the capture: library handles, workspaces, kernel selection
            for _ in range(2):
                self._minibatch(P, 0)
                self._adam(1)
        t.cuda.current_stream(self.dev).wait_stream(side)
        graphs = []
        if not several:
            gr = t.cuda.CUDAGraph()
            with t.cuda.graph(gr):
                for s in range(P["nmb"]):
                    self._minibatch(P, s)
                    self._adam(1)
            graphs.append(gr)
        else:
            pool = None
            for s in range(P["nmb"]):
                gr = t.cuda.CUDAGraph()
                with t.cuda.graph(gr, pool=pool):
                    self._minibatch(P, s)
                pool = pool or gr.pool()
                graphs.append(gr)
        for dst, src in zip((self.flat_p, self.m, self.v, self.state), keep):     # the warm-up took real optimiser steps: undo them
            dst.copy_(src)
        P["graphs"] = graphs

    # ---- the learner interface (ppo.PPO.update) --------------------------------------------------------------------------
    def update(self, obs, actions, adv, ret, old_logp=None, epochs=1, generator=None):
        """obs [B,160], actions [B,12] (unclipped samples), adv [B] (already normalised), ret [B] (TD(lambda) targets);
        returns the mean (surrogate, value loss) over the minibatches like ppo.PPO.update."""
        t = self.torch
        B = int(obs.shape[0])
        M = min(self.minibatch, B)
        if B % M:
            raise ValueError("FusedPPO: the number of samples (%d) must be a multiple of the minibatch size (%d); "
                             "FusedPPO.fit_minibatch(%d, %d) = %d is the largest size that divides it" % (B, M, B, M, self.fit_minibatch(B, M)))
        world, several = self._world(), self._several()
        with t.no_grad():
            if old_logp is None:
                old_logp = self.model.log_prob(obs, actions)
            P = self._plan(B, M)
            P["obs"].copy_(obs)
            aux = P["aux"]
            aux[:, :12].copy_(actions)
            aux[:, 12].copy_(old_logp)
            aux[:, 13].copy_(adv)
            aux[:, 14].copy_(ret)
            if self.use_graph and (P["graphs"] is None or P.get("graph_key") != self._graph_key(several)):
                self._capture(P, several)
                P["graph_key"] = self._graph_key(several)
            total = t.zeros(2, dtype=t.float32, device=self.dev)
            for _ in range(epochs):
                P["perm"].copy_(t.randperm(B, device=self.dev, generator=generator))
                if not self.use_graph:
                    self._epoch_eager(P, world, several)
                elif not several:
                    P["graphs"][0].replay()
                else:
                    for gr in P["graphs"]:
                        gr.replay()
                        self._allreduce()
                        self._adam(world)
                total += P["stats"].sum(dim=0)
        if hasattr(self.model, "mark_updated"):
            self.model.mark_updated()
        return (total / float(epochs * P["nmb"])).cpu().numpy()

    def steps_taken(self):
        return int(self.state[0].item())

    # ---- replica consistency (MpiAdam.sync / check_synced, stable_baselines/common/mpi_adam.py:64-82) -----------------------
    def _bcast_root(self):
        import torch.distributed as dist
        root = self.flat_p.clone()
        # `src` of a broadcast is a GLOBAL rank: the root of a subgroup is its own rank 0, whatever global rank that is
        src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
        if dist.get_backend(self.group) == "gloo":
            host = root.cpu()
            dist.broadcast(host, src=src, group=self.group)
            root.copy_(host)
        else:
            dist.broadcast(root, src=src, group=self.group)
        return root

    def sync(self):
        """Every rank takes rank 0's parameters (mpi_adam.py:64-70; the reference calls it once before training)."""
        if self._several():
            with self.torch.no_grad():
                self.flat_p.copy_(self._bcast_root())
            if hasattr(self.model, "mark_updated"):
                self.model.mark_updated()

    def check_synced(self):
        """Raises unless this rank's parameters equal rank 0's bit for bit (mpi_adam.py:72-82; the reference checks every 100 steps)."""
        if self._several() and not bool(self.torch.equal(self._bcast_root(), self.flat_p)):
            raise RuntimeError("FusedPPO.check_synced: this rank's parameters differ from rank 0's")
