"""Data-parallel WT-PSE training: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference is single-device (train.py:33); this is new functionality.  The multi-source-domain
mini-batch is domain-major [D0 x pb, D1 x pb, D2 x pb] (Trainer.py:45-55); rank g of G takes rows
[d*pb + g*pb/G, d*pb + (g+1)*pb/G) of EVERY domain, so its local batch is domain-major too and sees all domains.

Gradient exchange: all-reduces (mean) of contiguous RANGES of a network's flat gradient buffer — 25.5 MB (WT_PSE) /
12.8 MB (student) in two to four pieces — instead of ~390 per-tensor reductions.  Convention: every rank differentiates
its rank-local loss L_r with mean_r L_r = L_global, so the exchange is a plain average.
Overlap ("Collective 1"): the backward schedules announce a range as soon as nothing will write it any
more (`HipNet.grads_ready`: the 1x1 heads, the decoder, the teacher ... in the order the backward finishes them); its
all-reduce is issued behind the streams that produced it and runs on RCCL's stream beside the rest of the backward;
`allreduce_grads` — called where the single collective used to be — reduces what is left and joins the pieces.

Two modes
  * `bn_sync=False` (throughput; standard DDP semantics): BatchNorm statistics, the MMD term and the OC pos_weight
    are per-rank; the only collective is the gradient all-reduce.
  * `bn_sync=True` (exact): reproduces the single-device result on the global batch.  The four global-batch
    couplings of SURVEY.md §8e are synchronised: BatchNorm (sum, sum^2) forward and (sum dy, sum dy*xhat) backward
    (2*C floats per layer), an all-gather of the [B,120] upper-triangle vectors for the MMD, the two pos_weight
    sums, and sampling noise drawn from one global Philox stream indexed by global row.
"""
import torch
import torch.distributed as dist

from . import ops


# ------------------------------------------------------------------------------------------------ pure index logic
def local_rows(per_domain_global, domains, world, rank):
    """Global row indices (domain-major global batch) owned by `rank`, in local (domain-major) order."""
    assert per_domain_global % world == 0, "per-domain batch %d is not divisible by %d ranks" % (per_domain_global, world)
    n_l = per_domain_global // world
    return [d * per_domain_global + rank * n_l + i for d in range(domains) for i in range(n_l)]


def gathered_to_global_index(n_local, domains, world):
    """Index tensor `idx` with global[j] = gathered.reshape(world * B_used, ...)[idx[j]] where `gathered` stacks the
    ranks' domain-major local rows (rank-major) and `global` is domain-major over the global batch."""
    idx = []
    b_used = domains * n_local
    for d in range(domains):
        for g in range(world):
            for i in range(n_local):
                idx.append(g * b_used + d * n_local + i)
    return torch.tensor(idx, dtype=torch.long)


class DataParallel:
    def __init__(self, world, rank, device, bn_sync=False, group=None, domains=3, overlap=None):
        self.world, self.rank, self.device = int(world), int(rank), device
        # overlap=None: on in the throughput mode, off in the exact mode — there the early bucket all-reduces would interleave with
        # ~100 BatchNorm-statistics collectives per pass on the same communicator, an order no multi-GPU run has exercised yet
        self.overlap = (not bn_sync) if overlap is None else bool(overlap)
        self._pieces = {}            # id(net) -> [(lo, hi, work, view)]: ranges whose all-reduce is in flight
        self._issue = {}             # device -> stream the early all-reduces are issued from
        self.bn_sync = bool(bn_sync)
        self.exact = self.bn_sync
        self.group = group
        self.domains = domains
        self._avg_native = dist.get_backend(group) == "nccl"
        self._idx_cache = {}

    # -------------------------------------------------------------------------------------------- collectives
    def allreduce_mean_(self, t):
        if self._avg_native:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:                                   # gloo (CPU rehearsal / tests) has no AVG
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.mul_(1.0 / self.world)
        return t

    def allreduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def bucket_ready(self, net, gflat, lo, hi, streams=()):
        """gflat[lo:hi] is final once the work queued so far on `streams` has run: start its all-reduce now, behind them,
        beside whatever the caller enqueues next.  (No-op without `overlap`.)"""
        if not self.overlap or hi <= lo:
            return
        view = gflat[lo:hi]
        if gflat.is_cuda:
            cs = self._issue.get(gflat.device)
            if cs is None:
                cs = self._issue[gflat.device] = torch.cuda.Stream(device=gflat.device)
            for st in streams:
                cs.wait_stream(st)
            with torch.cuda.stream(cs):        # RCCL's own stream picks the dependency up from the stream the call is made on
                work = dist.all_reduce(view, op=dist.ReduceOp.AVG if self._avg_native else dist.ReduceOp.SUM, group=self.group,
                                       async_op=True)
        else:
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pieces.setdefault(id(net), []).append((lo, hi, work, view))

    def allreduce_grads(self, net, gflat):
        """Finish a network's gradient exchange: all-reduce the ranges no bucket_ready() covered (all of it without
        overlap: one collective per backward) and make the current stream wait for the pieces in flight."""
        pieces = self._pieces.pop(id(net), [])
        pos, n = 0, gflat.numel()
        for lo, hi, _, _ in sorted(pieces, key=lambda p: p[0]):
            if lo > pos:
                self.allreduce_mean_(gflat[pos:lo])
            pos = max(pos, hi)
        if pos < n:
            self.allreduce_mean_(gflat[pos:n])
        for _, _, work, view in pieces:
            work.wait()                           # RCCL: the current stream waits for the collective; gloo: the host does
            if not (gflat.is_cuda and self._avg_native):
                view.mul_(1.0 / self.world)

    def broadcast_params(self, nets):
        for n in nets:
            dist.broadcast(n.flat_params(), src=0, group=self.group)
            n.invalidate_packed()
            n.ensure_ready()
            bufs = [b for b in n.buffers()]
            if bufs:
                fl = torch.cat([b.detach().reshape(-1).to(torch.float64) for b in bufs])
                dist.broadcast(fl, src=0, group=self.group)
                o = 0
                with torch.no_grad():
                    for b in bufs:
                        b.copy_(fl[o:o + b.numel()].reshape(b.shape).to(b.dtype))
                        o += b.numel()

    # -------------------------------------------------------------------------------------------- BatchNorm (exact mode)
    def sync_bn_stats(self, stats, count):
        """Per-workgroup (sum, sum^2) partials [nblk, C, 2] -> global sums as a 1-slab partial, global count."""
        nblk, C, _ = stats.shape
        folded = torch.empty((1, C, 2), dtype=torch.float32, device=stats.device)
        ops.lib().call("wtpse_reduce_rows", stats.data_ptr(), nblk, 2 * C, folded.data_ptr(), 0, 1.0, ops.stream_ptr())
        self.allreduce_sum(folded)
        return folded, count * self.world

    def bn_bwd_synced(self, dz, t, bn, root):
        B, C, H, W = t.y.shape
        L = ops.lib()
        ns = L.query("wtpse_bn_bwd_nsplit", B, C, H * W)
        partial = ops.workspace("bn_bwd_partial", ns * C * 2, dz.device)
        coef = ops.workspace("bn_bwd_coef", C * 3, dz.device)
        s_local = torch.empty((C, 2), dtype=torch.float32, device=dz.device)
        L.call("wtpse_bn_bwd_reduce", dz.data_ptr(), t.y.data_ptr(), t.ss.data_ptr(), int(t.relu), t.mean.data_ptr(),
               t.invstd.data_ptr(), partial.data_ptr(), s_local.data_ptr(), B, C, H * W, ops.stream_ptr())
        s_global = self.allreduce_sum(s_local.clone())
        dy = torch.empty_like(t.y)
        dy.wt_amax = ops._amax_table(dy.device)
        L.call("wtpse_bn_bwd_apply", dz.data_ptr(), t.y.data_ptr(), t.ss.data_ptr(), int(t.relu), bn.weight.data_ptr(),
               t.mean.data_ptr(), t.invstd.data_ptr(), s_local.data_ptr(), s_global.data_ptr(), B * H * W * self.world,
               coef.data_ptr(), root.gview(bn.weight).data_ptr(), root.gview(bn.bias).data_ptr(), 0, dy.data_ptr(), B, C,
               H * W, ops.ptr(dy.wt_amax), ops.stream_ptr())
        return dy

    # -------------------------------------------------------------------------------------------- WT loss
    def _index(self, n_local, D, device):
        key = (n_local, D, str(device))
        if key not in self._idx_cache:
            to_global = gathered_to_global_index(n_local, D, self.world).to(device)
            mine = torch.tensor(local_rows(n_local * self.world, D, self.world, self.rank), dtype=torch.long, device=device)
            self._idx_cache[key] = (to_global, mine)
        return self._idx_cache[key]

    def gather_domain_major(self, v_local, n_local, D):
        """[B_local, F] (first D*n_local rows domain-major) on every rank -> [D*n_local*world, F] domain-major global."""
        used = v_local[: D * n_local].contiguous()
        out = [torch.empty_like(used) for _ in range(self.world)]
        dist.all_gather(out, used, group=self.group)
        to_global, _ = self._index(n_local, D, v_local.device)
        return torch.cat(out, 0).index_select(0, to_global)

    def wt_loss_fwd(self, z, D, n_local, margin, eps, losses_out):
        if not self.exact:
            return ops.wt_loss_fwd(z, D, n_local, margin, eps, losses_out)
        B, C, H, W = z.shape
        HW = H * W
        dev = z.device
        L = ops.lib()
        S = L.query("wtpse_wt_split", B, HW, 0)
        st = ops.WtLossState()
        partial = ops.workspace("wt_partial", B * S * 256, dev)
        st.gram = torch.empty((B, 256), dtype=torch.float32, device=dev)
        st.v = torch.empty((B, 120), dtype=torch.float32, device=dev)
        st.offdiag = torch.empty((B,), dtype=torch.float32, device=dev)
        st.diag = torch.empty((B,), dtype=torch.float32, device=dev)
        L.call("wtpse_wt_gram_fwd", z.data_ptr(), B, C, HW, float(eps), partial.data_ptr(), st.gram.data_ptr(),
               st.v.data_ptr(), st.offdiag.data_ptr(), st.diag.data_ptr(), ops.stream_ptr())
        v_g = self.gather_domain_major(st.v, n_local, D)
        n_g = n_local * self.world
        R_g = D * n_g
        rowval = torch.empty((R_g,), dtype=torch.float64, device=dev)
        dmmd_g = torch.empty((R_g, 120), dtype=torch.float32, device=dev)
        L.call("wtpse_mmd_fwd", v_g.data_ptr(), D, n_g, rowval.data_ptr(), dmmd_g.data_ptr(), ops.stream_ptr())
        st.losses = losses_out if losses_out is not None else torch.empty((3,), dtype=torch.float32, device=dev)
        L.call("wtpse_wt_final", st.offdiag.data_ptr(), st.diag.data_ptr(), B, B * self.world, float(margin),
               rowval.data_ptr(), R_g, st.losses.data_ptr(), ops.stream_ptr())
        self.allreduce_sum(st.losses[0:2])                  # instance means over the global batch
        _, mine = self._index(n_local, D, dev)
        st.dmmd_dv = dmmd_g.index_select(0, mine).contiguous()   # d dom_global / d v for this rank's rows
        st.z, st.B, st.HW, st.D, st.n, st.margin = z, B, HW, D, n_local, float(margin)
        return st

    def wt_loss_bwd(self, st, dz, accumulate, g_off=None, g_diag=None, g_dom=None, w_off=1.0, w_diag=1.0, w_dom=1.0):
        # gradients are averaged over ranks afterwards: the instance terms are already rank-local means; the MMD term
        # is the global one and each rank carries only its own rows' share of it -> weight it by `world`
        if self.exact:
            w_dom = w_dom * self.world
        return ops.wt_loss_bwd(st, dz, accumulate, g_off, g_diag, g_dom, w_off, w_diag, w_dom)

    # -------------------------------------------------------------------------------------------- sampling noise
    def noise(self, shape, seed, counter):
        """-> (eps [B_local, ...], elements consumed from the global stream); `counter`: device uint64 position of the
        stream (the caller advances it).  Exact mode: row r of the local batch takes the slice of the single global
        stream that belongs to its global row, so G ranks reproduce G = 1."""
        B = int(shape[0])
        per_row = 1
        for s in shape[1:]:
            per_row *= int(s)
        if not self.exact:
            n = B * per_row
            return ops.randn(shape, self.device, seed + 7919 * (self.rank + 1), 0, counter), (n + 3) & ~3
        assert per_row % 4 == 0, "rows must be a multiple of 4 elements for the global Philox stream"
        D = self.domains
        n_l = B // D
        assert n_l * D == B, "exact data-parallel sampling needs a batch that is a multiple of the domain count"
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        n_g = n_l * self.world
        for d in range(D):
            g0 = d * n_g + self.rank * n_l
            seg = out[d * n_l:(d + 1) * n_l]
            ops.lib().call("wtpse_randn", seg.data_ptr(), n_l * per_row, int(seed) & 0xFFFFFFFFFFFFFFFF,
                           g0 * per_row, counter.data_ptr(), ops.stream_ptr())
        return out, D * n_g * per_row
