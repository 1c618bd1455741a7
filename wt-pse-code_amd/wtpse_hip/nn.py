"""Host-side engine of the WT-PSE hot path on MI355X.

* Parameter containers reproduce the reference's ``state_dict`` key names so its checkpoints load.
* A root network (``HipNet``) keeps ALL its parameters in one flat device buffer (``nn.Parameter``s are views),
  one flat gradient buffer (``.grad``s are views), and one packed-weight buffer in the kernels' layouts, refreshed
  by a single launch whenever the parameters change.  Flat buffers make the optimiser step one kernel and the
  data-parallel gradient exchange one RCCL all-reduce per network.
* Forward and backward of every block are explicit schedules of C-ABI calls (ops.py) with a small tape; torch
  autograd only sees one Function per ``update()`` (algorithms.py / shape_networks.py).

Reference blocks mirrored here: ConvD / ConvU (algorithms.py:877-962), DoubleConv (:398-413), DeepWT / DoubleConvWT
(:416-428,1080-1117), ShapeVariationalDist_y_x (:979-1075), attention_layer (:1120-1129), heads (:1006-1012,1199-1201).
"""
import math
import os

import torch
import torch.nn as nn

from . import ops


# ================================================================================================ parameter containers
class ConvP(nn.Module):
    """weight [Cout,Cin,k,k] + bias [Cout], initialised like nn.Conv2d (kaiming_uniform(a=sqrt(5)))."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(self.bias, -bound, bound)
        self.wf_off = self.wd_off = -1
        self.xf_off = self.xd_off = -1      # offsets (unsigned shorts) into the root's x3-packed weights; -1: fp32-MFMA path
        self.x16f_off = self.x16d_off = -1  # same, the 16-channel x3 layout (csrc/conv.hip MODE 3)


class BNP(nn.Module):
    """nn.BatchNorm2d state: weight, bias, running_mean, running_var, num_batches_tracked."""

    def __init__(self, c):
        super().__init__()
        self.c = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Seq(nn.Module):
    """Children under the numeric names nn.Sequential would give them (activations occupy the gaps)."""

    def __init__(self, **children):
        super().__init__()
        for name, mod in children.items():
            self.add_module(name.lstrip("_"), mod)

    def __getitem__(self, i):
        return getattr(self, str(i))


class ConvDBlock(nn.Module):
    def __init__(self, cin, c, first=False):
        super().__init__()
        self.first = first
        self.conv1, self.bn1 = ConvP(cin, c, 3), BNP(c)
        self.conv2, self.bn2 = ConvP(c, c, 3), BNP(c)
        self.conv3, self.bn3 = ConvP(c, c, 3), BNP(c)


class ConvUBlock(nn.Module):
    def __init__(self, planes, first=False):
        super().__init__()
        self.first = first
        if not first:
            self.conv1, self.bn1 = ConvP(2 * planes, planes, 3), BNP(planes)
        self.conv2, self.bn2 = ConvP(planes, planes // 2, 1), BNP(planes // 2)
        self.conv3, self.bn3 = ConvP(planes, planes, 3), BNP(planes)


class DoubleConvWTP(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.double_conv = Seq(_0=ConvP(cin, cout, 3), _2=ConvP(cout, cout, 3))


class DeepWTP(nn.Module):
    """DeepWT (algorithms.py:1080-1117): 3->16->16, ReLU, 16->16->16; returns [z1, z2, relu(z2)].
    relu(z2) is never materialised: consumers take z2 with ReLU-on-load."""

    def __init__(self, cin=3, c=16):
        super().__init__()
        self.DoubleConv = DoubleConvWTP(cin, c)
        self.DoubleConv2 = DoubleConvWTP(c, c)

    def forward(self, x):
        self._root.ensure_ready(repack=True)
        with ops.fwd_scope(x.device):
            t = deepwt_fwd(self, x.detach().to(torch.float32).contiguous(), want_tape=False)
            return [t.z1, t.z2, Act(t.z2, None, True).dense()]      # (the maps carry their amax tables: Act)


class DoubleConvP(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.double_conv = Seq(_0=ConvP(cin, cout, 3), _1=BNP(cout), _3=ConvP(cout, cout, 3), _4=BNP(cout))


def head_p(n=16, n_classes=1):
    return Seq(_0=ConvP(2 * n, 2 * n, 1), _2=ConvP(2 * n, 8, 1), _4=ConvP(8, n_classes, 1))


class UNetBody(nn.Module):
    """down1-4 / up1-4 shared by the segmentation net, the teacher and the student."""

    def _make_body(self, n=16):
        self.down1 = ConvDBlock(n, 2 * n)
        self.down2 = ConvDBlock(2 * n, 4 * n)
        self.down3 = ConvDBlock(4 * n, 8 * n)
        self.down4 = ConvDBlock(8 * n, 16 * n)
        self.up1 = ConvUBlock(16 * n, first=True)
        self.up2 = ConvUBlock(8 * n)
        self.up3 = ConvUBlock(4 * n)
        self.up4 = ConvUBlock(2 * n)


class AttentionP(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer1 = ConvP(1, 1, 1)

    def forward(self, x):
        """attention_layer.forward (algorithms.py:1126-1129) -> (sigmoid(conv(x)), conv(x))."""
        root = self.layer1._root
        root.ensure_ready(repack=True)
        dummy = x.new_zeros((x.shape[0], 1, x.shape[2], x.shape[3]))
        att, pre, _, _ = ops.attn_fuse_fwd(x.contiguous(), self.layer1.weight.data_ptr(), dummy, 0.0, True, True, False)
        return att, pre


class TeacherP(UNetBody):
    """ShapeVariationalDist_y_x with whitening=True (algorithms.py:979-1075)."""

    def __init__(self, n=16):
        super().__init__()
        self.inc = DoubleConvP(1, n)
        self.fusion = Seq(_0=ConvP(2 * n, n, 1))
        self._make_body(n)
        self.mu_prior = head_p(n)
        self.logvar_prior = head_p(n)

    def sample_forward(self, inputs, mask=None, training=True):
        """(z, mu) in training, mu otherwise; `inputs` = W[-1] = relu(z2) materialised by the caller."""
        root = self.fusion[0]._root
        root.ensure_ready(repack=True)
        with ops.fwd_scope(inputs.device):
            t = teacher_fwd(self, inputs.contiguous(), mask.contiguous(), self.training, want_logvar=training,
                            want_tape=False)
        if not training:
            return t.mu
        eps = root.next_noise(t.mu.shape)
        return ops.reparam_fwd(t.mu, t.logvar, eps), t.mu


# ================================================================================================ root network
def _invalidate_on_load(module, incompatible_keys):
    root = module.__dict__.get("_root")
    if root is not None:
        root.invalidate_packed()


class HipNet(nn.Module):
    """Root of a parameter tree: owns the flat parameter / gradient / packed-weight buffers."""

    def _finish_init(self):
        object.__setattr__(self, "_flat", None)
        object.__setattr__(self, "_gflat", None)
        object.__setattr__(self, "_gwork", None)
        object.__setattr__(self, "_packed", None)
        object.__setattr__(self, "_x3", None)
        object.__setattr__(self, "_xdesc", None)
        object.__setattr__(self, "_x16desc", None)
        object.__setattr__(self, "_packed_version", -1)
        object.__setattr__(self, "_desc", None)
        object.__setattr__(self, "_touched", [])
        object.__setattr__(self, "_noise_queue", [])
        object.__setattr__(self, "_noise_seed", 0x5eed)
        object.__setattr__(self, "_noise_ctr", None)
        object.__setattr__(self, "_dp", None)
        object.__setattr__(self, "_flag", None)
        object.__setattr__(self, "_packed_valid", False)
        object.__setattr__(self, "_attach_grads", True)
        self._convs = [m for m in self.modules() if isinstance(m, ConvP)]
        for m in self.modules():
            if isinstance(m, (ConvP, BNP, AttentionP, TeacherP, DeepWTP)):
                object.__setattr__(m, "_root", self)
        # A checkpoint loaded into a live network (test_visulization.py:132-193: filtered load_state_dict, also on sub-modules)
        # changes the flat parameters behind the packed copies: every parameter holder reports the load to the root, which then
        # repacks on the next entry even when a step harness vouches for the packed weights (`_packed_valid`).
        for m in self.modules():
            if isinstance(m, (ConvP, BNP)):
                m.register_load_state_dict_post_hook(_invalidate_on_load)
        # weights of the layers that run on the split-bf16 ("x3") convolution (csrc/conv_x3.hip), pre-split into bf16 triples
        off = 0
        self._x3_convs = []
        for c in self._convs:
            fwd = x3_eligible(c.cin, c.cout, c.k) and os.environ.get("WTPSE_X3_FWD", "1") != "0"
            bwd = x3_eligible(c.cout, c.cin, c.k) and os.environ.get("WTPSE_X3_DGRAD", "1") != "0"
            if fwd:
                c.xf_off = off
                off += ops.x3_packed_size(c.cout, c.cin, c.k * c.k)
            if bwd:
                c.xd_off = off
                off += ops.x3_packed_size(c.cin, c.cout, c.k * c.k)
            if fwd or bwd:
                self._x3_convs.append(c)
        # the 16-channel 3x3 layers (inc, DeepWT, the last decoder conv): x3 arithmetic with register-resident weight fragments
        # (csrc/conv.hip MODE 3); same buffer, own layout
        self._x16_convs = []
        for c in self._convs:
            c.x16f_off = c.x16d_off = -1
            if X16 and c.k == 3 and c.cin <= 16 and c.cout <= 16:
                if c.cin >= X16_MIN_K:
                    c.x16f_off = off
                    off += ops.X16_SIZE
                if c.cout >= X16_MIN_K:
                    c.x16d_off = off
                    off += ops.X16_SIZE
                if c.x16f_off >= 0 or c.x16d_off >= 0:
                    self._x16_convs.append(c)
        self._x3_size = off
        # fp32-input MFMA layouts: only the directions that do not run on the x3 kernels (re-packing all 6.4 M weights of a
        # network into layouts nobody reads cost 120 us per network and step)
        off = 0
        for c in self._convs:
            t = c.k * c.k
            c.wf_off = c.wd_off = -1
            if c.xf_off < 0:
                c.wf_off = off
                off += ((c.cin + 3) & ~3) * t * ((c.cout + 15) & ~15)
            if c.xd_off < 0:
                c.wd_off = off
                off += ((c.cout + 3) & ~3) * t * ((c.cin + 15) & ~15)
        self._packed_size = max(off, 4)

    # ---- flat storage --------------------------------------------------------------------------------------
    def _is_flat(self):
        f = self._flat
        if f is None:
            return False
        # first and last parameter still views of the flat buffer (a `.to()` / `load_state_dict(assign=True)` replaces all
        # of them); walking every parameter on every entry cost ~5 ms of host time per step
        p0, pl = self._plist[0], self._plist[-1]
        return (p0.device == f.device and p0.data_ptr() == f.data_ptr()
                and pl.data_ptr() == f.data_ptr() + 4 * self._offsets[-1])

    def _flatten(self):
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("the WT-PSE MI355X path runs in device memory only: move the module to a HIP device "
                               "(`.to('cuda')`); there is no CPU fallback")
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        offsets, off = [], 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).to(dev, torch.float32))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = None
                offsets.append(off)
                off += n
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_offsets", offsets)
        object.__setattr__(self, "_pindex", {id(p): i for i, p in enumerate(params)})
        object.__setattr__(self, "_plist", params)
        object.__setattr__(self, "_gflat", torch.zeros(total, dtype=torch.float32, device=dev))
        object.__setattr__(self, "_gwork", None)
        object.__setattr__(self, "_packed", torch.empty(self._packed_size, dtype=torch.float32, device=dev))
        desc = []
        for c in self._convs:
            desc += [self._offsets[self._pindex[id(c.weight)]], c.cout, c.cin, c.k * c.k, c.wf_off, c.wd_off, 0, 0]
        object.__setattr__(self, "_desc", torch.tensor(desc, dtype=torch.int32).to(dev))
        object.__setattr__(self, "_packed_version", -1)
        object.__setattr__(self, "_x3", torch.empty(max(self._x3_size, 8), dtype=torch.int16, device=dev))
        xdesc = []
        for c in self._x3_convs:
            xdesc += [self._offsets[self._pindex[id(c.weight)]], c.cout, c.cin, c.k * c.k, c.xf_off, c.xd_off, 0, 0]
        object.__setattr__(self, "_xdesc", torch.tensor(xdesc, dtype=torch.int32).to(dev) if xdesc else None)
        xdesc = []
        for c in self._x16_convs:
            xdesc += [self._offsets[self._pindex[id(c.weight)]], c.cout, c.cin, 9, c.x16f_off, c.x16d_off, 0, 0]
        object.__setattr__(self, "_x16desc", torch.tensor(xdesc, dtype=torch.int32).to(dev) if xdesc else None)
        object.__setattr__(self, "_flag", torch.zeros(1, dtype=torch.int32, device=dev))
        ops._tickets(0, dev)          # the ticket ring of the self-folding data gradients: created outside any stream capture
        ops.amax_begin(dev)           # ... and the arena of the gradients' amax tables (x2h arithmetic)
        # position in this network's Philox stream.  It lives in device memory (advanced by a one-thread launch after each
        # draw) so that nothing that changes from step to step is passed to a kernel by value: a captured step (hipGraph)
        # then draws fresh noise on every replay, and eager and captured runs see the same stream.
        object.__setattr__(self, "_noise_ctr", torch.zeros(1, dtype=torch.int64, device=dev))
        for b in self.buffers():
            if b.device != dev:
                raise RuntimeError("parameters and buffers live on different devices")

    def ensure_ready(self, repack=False):
        """Flatten on first use / after `.to()`; refresh the packed weights.  `nn.Parameter.data` views do not share a
        version counter with the flat buffer, so a foreign optimiser's in-place step cannot be detected: public entry
        points (update / predict) repack unconditionally (one ~20 us launch) unless the step harness, which owns the
        optimiser, vouches for the packed copy through `_packed_valid`."""
        if not self._is_flat():
            self._flatten()
        terms = ops.x3_terms()       # the packed x3 weights are in the format of the arithmetic they were packed under
        if self._packed_version < 0 or (repack and not self._packed_valid) or self._packed_version != terms:
            ops.lib().call("wtpse_pack_conv_weights", self._flat.data_ptr(), self._desc.data_ptr(), len(self._convs),
                           self._packed.data_ptr(), ops.stream_ptr())
            if self._xdesc is not None:
                ops.lib().call("wtpse_pack_conv_weights_x3", self._flat.data_ptr(), self._xdesc.data_ptr(), len(self._x3_convs),
                               self._x3.data_ptr(), ops.stream_ptr())
            if self._x16desc is not None:
                ops.lib().call("wtpse_pack_conv16_x3", self._flat.data_ptr(), self._x16desc.data_ptr(), len(self._x16_convs),
                               self._x3.data_ptr(), ops.stream_ptr())
            object.__setattr__(self, "_packed_version", terms)

    def invalidate_packed(self):
        object.__setattr__(self, "_packed_version", -1)

    def packed_ptr(self, off):
        return self._packed.data_ptr() + 4 * off

    def x3_ptr(self, off):
        return self._x3.data_ptr() + 2 * off

    def flat_params(self):
        self.ensure_ready()
        return self._flat

    def flat_grads(self):
        self.ensure_ready()
        return self._gflat

    def param_offset(self, p):
        return self._offsets[self._pindex[id(p)]]

    # ---- gradients -----------------------------------------------------------------------------------------
    def begin_backward(self):
        """Choose the buffer this backward writes into: the attached flat gradient when every .grad is None
        (the normal zero_grad(set_to_none=True) flow), a work buffer otherwise (accumulation semantics)."""
        self._touched.clear()
        ops.amax_begin(self._flat.device)      # the amax tables of the previous pass are free again (zeroed: one launch)
        clean = all(p.grad is None for p in self._plist)
        if clean:
            object.__setattr__(self, "_gtarget", self._gflat)
        else:
            if self._gwork is None:
                object.__setattr__(self, "_gwork", torch.zeros_like(self._gflat))
            object.__setattr__(self, "_gtarget", self._gwork)

    def gview(self, p):
        """View of the current gradient target for parameter p; marks p as reached by this backward."""
        i = self._pindex[id(p)]
        self._touched.append(p)
        off = self._offsets[i]
        return self._gtarget[off:off + p.numel()]

    def grads_ready(self, first, last=None):
        """Data-parallel overlap: the parameter gradients of the consecutive submodules `first` .. `last` (a contiguous range of
        the flat buffer) are complete once the work queued so far on the current stream and on its weight-gradient side stream
        has run — their all-reduce can start now, beside the rest of the backward (dp.py).  Without data parallelism, while a
        launch plan is being recorded (collectives stay outside plans) or when gradients accumulate: nothing happens."""
        dp = self._dp
        if dp is None or not dp.overlap or self._gtarget is not self._gflat or ops.lib()._rec is not None:
            return
        if self._gflat.is_cuda and torch.cuda.is_current_stream_capturing():
            # TrainStep(graph=True): an asynchronous collective issued here would join the capture from the issue stream and
            # be waited for only after the capture has ended (unjoined work / an illegal host sync with gloo).  Captured
            # stretches keep their collectives between them (step.py): the whole buffer goes in allreduce_grads.
            return
        last = first if last is None else last
        pf, pl = next(first.parameters()), list(last.parameters())[-1]
        lo = self._offsets[self._pindex[id(pf)]]
        hi = self._offsets[self._pindex[id(pl)]] + pl.numel()
        cur = torch.cuda.current_stream()
        streams = [cur]
        side = _SIDE.get((self._gflat.device, ops.stream_ptr()))
        if side is not None:
            streams.append(side)
        dp.bucket_ready(self, self._gtarget, lo, hi, streams)

    def end_backward(self):
        joins = self.__dict__.get("_join")
        if joins:                                   # helper streams this backward put work on (weight gradients, prior chain)
            cur = torch.cuda.current_stream()
            for st in joins:
                stream_wait(cur, st)
            joins.clear()
        if self._dp is not None and not self.__dict__.get("_defer_allreduce"):     # the step harness issues it itself
            self._dp.allreduce_grads(self, self._gtarget)
        direct = self._gtarget is self._gflat
        if direct and not self._attach_grads:   # the step harness reads the flat buffer itself
            self._touched.clear()
            return
        for p in self._touched:
            off = self._offsets[self._pindex[id(p)]]
            g = self._gtarget[off:off + p.numel()].view(p.shape)
            if direct:
                p.grad = g
            elif p.grad is None:
                p.grad = g.clone()
            else:
                ops.axpy(p.grad, g.contiguous(), 1.0) if p.grad.is_contiguous() else p.grad.add_(g)
        self._touched.clear()

    # ---- sampling noise --------------------------------------------------------------------------------------
    def set_noise(self, tensors):
        """Inject the standard-normal draws of the next sampling calls (parity tests); [] returns to Philox."""
        object.__setattr__(self, "_noise_queue", list(tensors))

    def seed_noise(self, seed):
        self.ensure_ready()
        object.__setattr__(self, "_noise_seed", int(seed))
        ops.zero_(self._noise_ctr)

    def next_noise(self, shape):
        if self._noise_queue:
            e = self._noise_queue.pop(0)
            assert tuple(e.shape) == tuple(shape), (e.shape, shape)
            return e.to(self._flat.device, torch.float32).contiguous()
        ctr = self._noise_ctr
        if self._dp is not None:   # exact mode: one global stream indexed by global row (any sharding reproduces G = 1)
            eps, used = self._dp.noise(tuple(int(v) for v in shape), self._noise_seed, ctr)
        else:
            n = 1
            for s in shape:
                n *= int(s)
            eps, used = ops.randn(shape, self._flat.device, self._noise_seed, 0, ctr), (n + 3) & ~3
        ops.counter_add(ctr, used)
        return eps


# ================================================================================================ block schedules
class Tape:
    """Attribute bag holding what a block's backward needs."""
    pass


class Act:
    """An activation as its producer left it in HBM: the stored tensor `t`, still to be passed through
    act(t * scale + shift) — BatchNorm-apply (`pro` = [C,2] scale/shift or None) and ReLU (`relu`) — which every consumer
    (conv / weight-gradient loaders, max-pool, bilinear upsample) applies as it loads.  BatchNorm outputs, relu(z2) and the
    two halves of a torch.cat therefore never make a round trip through memory.
    `amax` (x2h arithmetic): the amax table holding a BOUND of the largest magnitude of the activation as loaded — the scale its x2h
    consumers load it with (include/wtpse_hip.h, wtpse_x3_terms).  Given by whoever made the Act (the BatchNorm finalize: no look at
    the data); a tensor that is loaded as stored inherits the table its producer's epilogue attached (`t.wt_amax`); unknown (None):
    act_amax() finds it with one extra pass the first time a consumer asks."""
    __slots__ = ("t", "pro", "relu", "amax")

    def __init__(self, t, pro=None, relu=False, amax=None):
        self.t, self.pro, self.relu = t, pro, bool(relu)
        self.amax = amax if amax is not None else (getattr(t, "wt_amax", None) if pro is None else None)

    def dense(self):
        if self.pro is None and not self.relu:
            return self.t
        z = ops.affine_act(self.t, self.pro, self.relu)
        if self.amax is not None:
            z.wt_amax = self.amax
        return z


def as_act(x):
    return x if isinstance(x, Act) else Act(x)


# WTPSE_FWD_AMAX=0: the consumers fall back to round 5's fixed 2^2 input scale (A/B of what the data-driven scale costs; not a mode to run in)
FWD_AMAX = os.environ.get("WTPSE_FWD_AMAX", "1") != "0"


def act_amax(a):
    """x2h arithmetic: the amax table of the activation `a` as loaded (Act.amax), None otherwise.  An activation nobody supplied a bound
    for (a caller's own tensor entering a block) pays one pass over its data here, once: the table stays on the Act."""
    if a is None or ops.x3_terms() != 2 or not FWD_AMAX:
        return None
    if a.amax is None:
        raw = ops.amax_of(a.t)
        a.amax = raw if a.pro is None else ops.act_bound(a.pro, raw)
    return a.amax


def _relu_bits(a0, a1):
    return (1 if a0.relu else 0) | (2 if (a1 is not None and a1.relu) else 0)


# Convolutions with more than 16 output channels and a K dimension of at least 16 run on the bf16 matrix cores at fp32
# accuracy (three bf16 terms per fp32 operand, six products, fp32 accumulation: csrc/conv_x3.hip — 1.3-1.8x the fp32-MFMA
# kernel on these layers); the 16-output-channel layers, the 1-/3-channel input layers and the small 1x1 convs stay on the
# fp32-input MFMA (csrc/conv.hip).  WTPSE_X3=0 routes everything to the fp32-input MFMA.
X3 = os.environ.get("WTPSE_X3", "1") != "0"
X3_WGRAD = os.environ.get("WTPSE_X3_WGRAD", "1") != "0"
WGRAD_R = os.environ.get("WTPSE_WGRAD_R", "1") != "0"      # =0: the LDS-based weight-gradient kernels of rounds 1-2
# 16-channel 3x3 layers (<= 16 in, <= 16 out) in the x3 arithmetic (csrc/conv.hip MODE 3); =0: fp32-input MFMA.  Layers whose
# reduction has fewer than X16_MIN_K channels (the 1- and 3-channel inputs) are bound by their output write on either path and
# stay on the fp32-input MFMA.
X16 = os.environ.get("WTPSE_X16", "1") != "0" and X3
X16_MIN_K = int(os.environ.get("WTPSE_X16_MIN_K", "8"))


def x3_eligible(k_dim, rows, ksize):
    """rows = output channels of the launch (Cout forward, Cin for a data gradient), k_dim = its reduction channels."""
    return X3 and rows > 16 and 16 <= k_dim <= 256 and (ksize == 3 or k_dim >= 64)     # 256: wtpse_conv_fwd_x3 (include/wtpse_hip.h)


def _conv(layer, a0, a1=None, relu_out=False, want_stats=False, want_amax=False):
    """want_amax (x2h arithmetic): the launch also leaves the amax table of its output on the result (`y.wt_amax`) — for outputs
    that reach an x2h consumer without a train-mode BatchNorm in between (DeepWT's maps, the fusion conv, eval-mode BatchNorm)."""
    root = layer._root
    a0 = as_act(a0)
    a1 = as_act(a1) if a1 is not None else None
    out_amax = ops.fwd_amax_table(a0.t.device) if want_amax else None
    if layer.x16f_off >= 0 and a1 is None:
        y, stats, _ = ops.conv16_x3(a0.t, root.x3_ptr(layer.x16f_off), layer.bias, layer.cout, a0.pro, _relu_bits(a0, None),
                                    relu_out, want_stats, in_amax=act_amax(a0), out_amax=out_amax)
        return y, stats
    if layer.xf_off >= 0:
        y, _, stats = ops.conv_fwd_x3(a0.t, a1.t if a1 is not None else None, root.x3_ptr(layer.xf_off), layer.bias, layer.cout,
                                      layer.k, a0.pro, _relu_bits(a0, a1), relu_out, want_stats, None, None,
                                      a1.pro if a1 is not None else None, act_amax(a0), act_amax(a1), out_amax)
        return y, stats
    y, _, stats = ops.conv_fwd(a0.t, a1.t if a1 is not None else None, root.packed_ptr(layer.wf_off), layer.bias, layer.cout,
                               layer.k, a0.pro, _relu_bits(a0, a1), relu_out, want_stats, None, None,
                               a1.pro if a1 is not None else None, out_amax)
    return y, stats


class PreBN:
    """A gradient wrt the activated output of a conv + BatchNorm (+ReLU) layer as the data gradient that produced it left it:
    already masked with that layer's ReLU (`g`), with the two reductions of the BatchNorm backward as per-workgroup partials
    (`stats`): convbn_bwd / upbn_bwd finish the BatchNorm backward with one elementwise pass (ops.bn_bwd_from_stats) instead
    of reduce + apply (5 -> 3 HBM passes over the layer's map)."""
    __slots__ = ("g", "stats", "coef")

    def __init__(self, g, stats, coef=None):
        self.g, self.stats, self.coef = g, stats, coef      # coef: the launch also folded the partials (BN_TAIL)


# BatchNorm-backward reductions in the epilogue of the producing data gradient (csrc/conv_x3.hip, conv.hip: EPI 2).
# WTPSE_BN_FUSED_STATS=0: the stand-alone reduce pass everywhere.
BN_FUSED_STATS = os.environ.get("WTPSE_BN_FUSED_STATS", "1") != "0"
# ... and the fold of those partials into the BatchNorm-backward coefficients by the same launch's last workgroups
# (csrc/common.h: bnb_tail).  WTPSE_BN_TAIL=0: the stand-alone fold (bn_bwd_finalize_k) in front of the apply pass.
BN_TAIL = os.environ.get("WTPSE_BN_TAIL", "1") != "0"


def _gamax(dy):
    """x2h arithmetic (ops.x3_terms() == 2): the device slot with the largest magnitude of the gradient `dy` — left on the tensor by
    its producer, or computed by one extra pass (ops.amax_of).  None otherwise."""
    return ops.amax_of(dy) if ops.x3_terms() == 2 else None


def _dgrad(layer, dy, split=None, mask_ref=None, below0=None, below1=None):
    """Data gradient; mask_ref fuses the ReLU backward of the tensor the gradient flows into (d * [ref > 0]).
    below0 / below1: tape of the conv + BatchNorm layer whose activated output the first / second returned gradient is taken
    with respect to (at most one of them): that gradient comes back as a PreBN."""
    root = layer._root
    below = below0 if below0 is not None else below1
    if (BN_FUSED_STATS and below is not None and mask_ref is None and below.mean is not None
            and not (root._dp is not None and root._dp.bn_sync) and (below1 is None or split is not None)):
        bn = below.bn
        tail = (bn.weight, below.invstd, root.gview(bn.weight), root.gview(bn.bias)) if BN_TAIL else None
        if layer.x16d_off >= 0 and split is None:
            layout, wptr = 2, root.x3_ptr(layer.x16d_off)
        elif layer.xd_off >= 0:
            layout, wptr = 1, root.x3_ptr(layer.xd_off)
        else:
            layout, wptr = 0, root.packed_ptr(layer.wd_off)
        d0, d1, stats, coef = ops.dgrad_bnb(dy, wptr, layout, layer.cin, layer.k, below.y, below.ss, below.mean, below.relu, split,
                                            below1 is not None and below0 is None, tail,
                                            _gamax(dy) if layout == 1 else getattr(dy, "wt_amax", None) if layout == 2 else None)
        if below0 is not None:
            return PreBN(d0, stats, coef), d1
        return d0, PreBN(d1, stats, coef)
    if layer.x16d_off >= 0 and split is None:
        return ops.conv16_x3(dy, root.x3_ptr(layer.x16d_off), None, layer.cin, mask_ref=mask_ref, grad_in=True)[0], None
    if layer.xd_off >= 0:
        return ops.conv_fwd_x3(dy, None, root.x3_ptr(layer.xd_off), None, layer.cin, layer.k, None, 0, False, False, split,
                               mask_ref, None, _gamax(dy))[:2]
    return ops.conv_fwd(dy, None, root.packed_ptr(layer.wd_off), None, layer.cin, layer.k, None, 0, False, False, split,
                        mask_ref)[:2]


# The weight gradient of a conv+BatchNorm block depends only on dy and the saved input, and nothing downstream in the
# backward pass reads it, so it runs on a side stream next to the data gradients that follow (small-grid launches of the
# deep levels leave CUs idle that the other stream's workgroups fill).  end_backward() joins the side stream.
_SIDE = {}
WGRAD_SIDE_STREAM = os.environ.get("WTPSE_WGRAD_STREAM", "1") != "0"


def _side_stream(device, owner=None):
    """One weight-gradient stream per stream it is fed from (the main stream and the prior-chain stream of a segmentation update)."""
    key = (device, ops.stream_ptr() if owner is None else owner)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=device)
    return st


_SECOND = {}
TEACHER_SIDE_STREAM = os.environ.get("WTPSE_TEACHER_STREAM", "1") != "0"


def second_stream(device):
    """Stream for forward-only work that is independent of the main schedule (the teacher in the student's update)."""
    if not TEACHER_SIDE_STREAM:
        return None
    key = (device, ops.stream_ptr())          # per calling stream
    st = _SECOND.get(key)
    if st is None:
        st = _SECOND[key] = torch.cuda.Stream(device=device)
    return st


def stream_wait(waiter, waited):
    """waiter.wait_stream(waited), also entered into the launch plan being recorded (wtpse_hip/lib.py), if any."""
    waiter.wait_stream(waited)
    ops.lib().plan_wait(waiter.cuda_stream, waited.cuda_stream)


def note_join(root, stream):
    """end_backward() of `root` must wait for `stream`."""
    j = root.__dict__.get("_join")
    if j is None:
        j = []
        object.__setattr__(root, "_join", j)
    if all(s is not stream for s in j):
        j.append(stream)


def _wgrad_side(layer, dy, a0, a1=None):
    """_wgrad(..., with_bias=False) on the side stream.  dy must not be written again by the caller (it is a fresh
    BatchNorm-backward result in both callers)."""
    root = layer._root
    if not WGRAD_SIDE_STREAM:
        return _wgrad(layer, dy, a0, a1, with_bias=False)
    amax = _gamax(dy)                # (on the main stream, in front of the fork: the data gradient that follows uses the same slot)
    # (likewise — normally there since the forward pass; only where the launch takes the x2h weight gradient: wgrad_r)
    xam = (act_amax(as_act(a0)), act_amax(as_act(a1)) if a1 is not None else None) if _takes_wgrad_r(layer, as_act(a0), a1) else (None, None)
    main = torch.cuda.current_stream()
    side = _side_stream(dy.device)
    stream_wait(side, main)
    with torch.cuda.stream(side):
        _wgrad(layer, dy, a0, a1, with_bias=False)
    a0 = as_act(a0)
    for t in (dy, amax, a0.t, a0.pro) + xam + ((as_act(a1).t, as_act(a1).pro) if a1 is not None else ()):
        if t is not None:
            t.record_stream(side)       # the caching allocator must not hand these out again before the side stream is done
    # (Under stream capture the allocator keeps record_stream'ed blocks out of circulation until the capture ends.  Holding the
    # tensors instead until the main stream has waited for the weight gradient that reads them — lagged by 2-3 layers, marks in
    # the launch plan — was built and measured in round 3: plan 577 / 571 images/s against 588 with record_stream, eager
    # unchanged; it is not where the replayed step loses its 1 % against the eager one.)
    note_join(root, side)


def _takes_wgrad_r(layer, a0, a1, with_bias=False):
    """3x3 layers on maps that are a multiple of 32 pixels (or exactly 16) wide: the x3 weight gradient with register-resident operands
    (csrc/wgrad_r.hip; bias gradient only in its 16 x 16-channel form: the DeepWT layers)"""
    return (X3 and X3_WGRAD and WGRAD_R and layer.k == 3 and
            ops.wgrad_r_supported(layer.cin, layer.cout, 3, a0.t.shape[1] if a1 is not None else 16, a0.t.shape[3]) and
            (not with_bias or (layer.cin % 32 != 0 and layer.cout % 32 != 0)))


def _wgrad(layer, dy, a0, a1=None, with_bias=True):
    root = layer._root
    a0 = as_act(a0)
    a1 = as_act(a1) if a1 is not None else None
    dw = root.gview(layer.weight)
    db = root.gview(layer.bias) if with_bias else None
    if _takes_wgrad_r(layer, a0, a1, with_bias):
        # (the 16 x 16-channel blocks are HBM-bound: an extra pass over dY to find its scale costs more than x2h gains there — they
        # take the table their dY's producer attached, or run in x3)
        small = layer.cin % 32 != 0 and layer.cout % 32 != 0
        ops.conv_wgrad_r(dy, a0.t, a1.t if a1 is not None else None, dw, db, a0.pro, _relu_bits(a0, a1), False,
                         a1.pro if a1 is not None else None, getattr(dy, "wt_amax", None) if small else _gamax(dy),
                         act_amax(a0), act_amax(a1))
        return
    if (X3 and X3_WGRAD and db is None and
            ops.wgrad_x3_supported(layer.cin, layer.cout, layer.k, a0.t.shape[1] if a1 is not None else 8)):
        ops.conv_wgrad_x3(dy, a0.t, a1.t if a1 is not None else None, layer.k, dw, a0.pro, _relu_bits(a0, a1), False,
                          a1.pro if a1 is not None else None)
        return
    ops.conv_wgrad(dy, a0.t, a1.t if a1 is not None else None, layer.k, dw, db, a0.pro, _relu_bits(a0, a1), False,
                   a1.pro if a1 is not None else None)


# ---- conv + BatchNorm (+ReLU): the result stays virtual (raw conv output + per-channel scale/shift) --------------
def convbn_fwd(conv, bn, a0, a1, relu, training, want_tape=True):
    root = conv._root
    a0 = as_act(a0)
    a1 = as_act(a1) if a1 is not None else None
    # (x2h) the bound of the activated output travels with it to its consumers: from the statistics' fold in train mode
    # (|gamma| sqrt(N - 1) + |beta|: common.h, bn_act_bound), from the stored data's amax behind an eval-mode BatchNorm
    tab = ops.fwd_amax_table(a0.t.device) if training else None
    if training and BN_TAIL and not (root._dp is not None and root._dp.bn_sync):
        # the convolution finishes its own statistics (csrc/common.h: bnf_tail): no finalize launch
        if conv.x16f_off >= 0 and a1 is None:
            layout, wptr = 2, root.x3_ptr(conv.x16f_off)
        elif conv.xf_off >= 0:
            layout, wptr = 1, root.x3_ptr(conv.xf_off)
        else:
            layout, wptr = 0, root.packed_ptr(conv.wf_off)
        y, ss, mean, invstd = ops.conv_fwd_bnf(a0.t, a1.t if a1 is not None else None, wptr, layout, conv.bias, conv.cout, conv.k,
                                               a0.pro, _relu_bits(a0, a1), a1.pro if a1 is not None else None, bn.weight, bn.bias,
                                               bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                               in_amax0=act_amax(a0) if layout else None,
                                               in_amax1=act_amax(a1) if layout == 1 else None, act_amax=tab)
    elif training:
        y, stats = _conv(conv, a0, a1, False, True)
        B, _, H, W = y.shape
        if root._dp is not None and root._dp.bn_sync:
            stats, count = root._dp.sync_bn_stats(stats, B * H * W)
        else:
            count = B * H * W
        ss, mean, invstd = ops.bn_finalize(stats, count, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                           bn.num_batches_tracked, act_amax=tab)
    else:
        y, _ = _conv(conv, a0, a1, False, False, want_amax=True)
        ss = ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var)
        tab = ops.act_bound(ss, y.wt_amax) if getattr(y, "wt_amax", None) is not None else None
        mean = invstd = None
    z = Act(y, ss, relu, tab)
    if not want_tape:
        return z, None
    t = Tape()
    t.a0, t.a1, t.y, t.ss, t.mean, t.invstd, t.relu = a0, a1, y, ss, mean, invstd, relu
    t.bn = bn
    return z, t


# The second half of a BatchNorm backward (dy = k1 g + k2 y + k3 per channel, from the coefficients the producing launch folded)
# formed by the CONSUMER of dy as it loads, instead of by an elementwise pass of its own (3 HBM passes over the layer's map).
# WTPSE_BN_IN=0: the stand-alone apply pass (wtpse_bn_bwd_apply_coef) everywhere.
BN_IN = os.environ.get("WTPSE_BN_IN", "1") != "0"
def _bn_bwd(bn, t, dz, root):
    """BatchNorm (+ReLU) backward of a convbn / upbn tape: dz = gradient wrt the activated output, plain or PreBN."""
    if isinstance(dz, PreBN):
        if dz.coef is not None:
            return ops.bn_bwd_apply_coef(dz.g, t.y, dz.coef)
        return ops.bn_bwd_from_stats(dz.g, t.y, dz.stats, bn.weight, t.mean, t.invstd, root.gview(bn.weight), root.gview(bn.bias))
    if root._dp is not None and root._dp.bn_sync:
        return root._dp.bn_bwd_synced(dz, t, bn, root)
    return ops.bn_bwd(dz, t.y, t.ss, t.relu, bn.weight, t.mean, t.invstd, root.gview(bn.weight), root.gview(bn.bias))


def convbn_bwd(conv, bn, t, dz, need_dx=True, below0=None, below1=None):
    """dz: gradient wrt the activated output (plain or PreBN).  -> (dx0, dx1): gradients wrt the inputs AS LOADED (activated);
    below0 / below1: see _dgrad."""
    root = conv._root
    dy = _bn_bwd(bn, t, dz, root)
    # the conv bias in front of a train-mode BatchNorm has an exactly-zero gradient (sum of dy over the batch
    # vanishes); the reference carries rounding noise there. It is left at 0.
    _wgrad_side(conv, dy, t.a0, t.a1)
    if not need_dx:
        return None, None
    split = t.a0.t.shape[1] if t.a1 is not None else None
    return _dgrad(conv, dy, split, below0=below0, below1=below1)


# ---- ConvD (algorithms.py:897-917) ---------------------------------------------------------------------------
def convd_fwd(blk, x, training, want_tape=True):
    t = Tape()
    x = as_act(x)
    t.x = x
    h = x if blk.first else Act(ops.maxpool2_fwd(x.t, x.pro, x.relu), amax=x.amax)      # (a pooled tensor is bounded like its source)
    a, t.c1 = convbn_fwd(blk.conv1, blk.bn1, h, None, False, training, want_tape)
    b, t.c2 = convbn_fwd(blk.conv2, blk.bn2, a, None, True, training, want_tape)
    c, t.c3 = convbn_fwd(blk.conv3, blk.bn3, b, None, True, training, want_tape)
    return c, t


def convd_bwd(blk, t, dz, dx_accum=None, need_dx=True, mask_x=False, below=None):
    """dx_accum: existing gradient buffer of x (skip connection) to accumulate into, or None.  mask_x: x is the output of a ReLU
    whose backward is applied to the returned gradient here (in the max-pool backward, which reads x anyway).  below: tape of the
    conv + BatchNorm + ReLU layer that produced x: the max-pool backward then also forms that layer's BatchNorm-backward
    reductions and the gradient comes back as a PreBN."""
    d, _ = convbn_bwd(blk.conv3, blk.bn3, t.c3, dz, below0=t.c2)
    d, _ = convbn_bwd(blk.conv2, blk.bn2, t.c2, d, below0=t.c1)
    d, _ = convbn_bwd(blk.conv1, blk.bn1, t.c1, d, need_dx=need_dx)
    if not need_dx:
        return None
    if blk.first:
        if dx_accum is not None:
            ops.axpy(dx_accum, d)
            d = dx_accum
        return ops.relu_mask(d, t.x.t) if mask_x else d
    root = blk.conv1._root
    if (BN_FUSED_STATS and below is not None and below.mean is not None and t.x.relu and t.x.pro is not None
            and not (root._dp is not None and root._dp.bn_sync)):
        r = ops.maxpool2_bwd_bnb(t.x.t, d, dx_accum, t.x.pro, t.x.relu, below.mean)
        if r is not None:
            return PreBN(r[0], r[1])
    return ops.maxpool2_bwd(t.x.t, d, dx_accum, dx_accum is not None, t.x.pro, t.x.relu, mask=mask_x)


# ---- ConvU (algorithms.py:941-962) ---------------------------------------------------------------------------
# The reference runs  upsample(x2, bilinear) -> conv2 (1x1) -> bn2 -> ReLU  (algorithms.py:949-951).  A 1x1 convolution is
# an affine map per pixel and bilinear interpolation an affine combination of pixels with weights summing to 1, so the
# two commute exactly in real arithmetic: conv2(up(x)) = up(conv2(x)).  The conv therefore runs on the low-resolution
# tensor (a quarter of the pixels; forward, data gradient and weight gradient alike) and the upsampling moves half as
# many channels; BatchNorm sees the same tensor as in the reference and takes its batch statistics from the upsampling
# kernel.  In fp32 the result differs from the reference order by rounding only (~1e-7 relative).
CONVU_CONV_FIRST = os.environ.get("WTPSE_CONVU_REFERENCE_ORDER", "0") != "1"    # =1: upsample -> conv2, as written in the reference


def upbn_fwd(conv, bn, a0, training, want_tape=True):
    root = conv._root
    z, _ = _conv(conv, a0, None, False, False, want_amax=not training)      # low resolution, pre-BatchNorm
    if training:
        y, stats = ops.upsample2x_fwd_stats(z)
        B, _, H, W = y.shape
        if root._dp is not None and root._dp.bn_sync:
            stats, count = root._dp.sync_bn_stats(stats, B * H * W)
        else:
            count = B * H * W
        tab = ops.fwd_amax_table(y.device)
        ss, mean, invstd = ops.bn_finalize(stats, count, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                           bn.num_batches_tracked, act_amax=tab)
    else:
        y = ops.upsample2x_fwd(z)
        ss = ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var)
        # (bilinear interpolation is a convex combination: the upsampled map is bounded by the amax of the low-resolution one)
        tab = ops.act_bound(ss, z.wt_amax) if getattr(z, "wt_amax", None) is not None else None
        mean = invstd = None
    out = Act(y, ss, True, tab)
    if not want_tape:
        return out, None
    t = Tape()
    t.a0, t.a1, t.y, t.ss, t.mean, t.invstd, t.relu = a0, None, y, ss, mean, invstd, True
    t.bn = bn
    return out, t


def upbn_bwd(conv, bn, t, dz, below=None):
    """-> gradient wrt the (activated, low-resolution) conv input (a PreBN if `below`, the tape of the layer it came from)."""
    root = conv._root
    if BN_IN and isinstance(dz, PreBN) and dz.coef is not None and dz.g.shape[3] % 8 == 0:
        dzl = ops.upsample2x_bwd_bn(dz.g, t.y, dz.coef)       # the apply pass on load: dy is never written
    else:
        dzl = ops.upsample2x_bwd(_bn_bwd(bn, t, dz, root))
    _wgrad_side(conv, dzl, t.a0, None)                         # bias in front of a train-mode BatchNorm: see convbn_bwd
    return _dgrad(conv, dzl, below0=below)[0]


def convu_fwd(blk, x, prev, training, want_tape=True):
    t = Tape()
    x = as_act(x)
    if not blk.first:
        x, t.c1 = convbn_fwd(blk.conv1, blk.bn1, x, None, True, training, want_tape)
    B, _, H, W = x.t.shape
    t.swapped = CONVU_CONV_FIRST and W % 2 == 0 and B * blk.conv2.cout < 32768   # what wtpse_upsample2x_fwd_stats takes
    if t.swapped:
        y, t.c2 = upbn_fwd(blk.conv2, blk.bn2, x, training, want_tape)
    else:                                                        # odd widths: the reference order
        u = Act(ops.upsample2x_fwd(x.t, x.pro, x.relu), amax=x.amax)
        y, t.c2 = convbn_fwd(blk.conv2, blk.bn2, u, None, True, training, want_tape)
    out, t.c3 = convbn_fwd(blk.conv3, blk.bn3, prev, y, True, training, want_tape)
    return out, t


def convu_bwd(blk, t, dout, below_x=None):
    """-> (dx, dprev); both wrt the activated tensors.  below_x: tape of the conv + BatchNorm layer that produced the block's
    input x, when x feeds nothing else (dx then comes back as a PreBN)."""
    dprev, dy = convbn_bwd(blk.conv3, blk.bn3, t.c3, dout, below1=t.c2)
    if t.swapped:
        dx = upbn_bwd(blk.conv2, blk.bn2, t.c2, dy, below=(below_x if blk.first else t.c1))
    else:
        du, _ = convbn_bwd(blk.conv2, blk.bn2, t.c2, dy)
        dx = ops.upsample2x_bwd(du)
    if not blk.first:
        dx, _ = convbn_bwd(blk.conv1, blk.bn1, t.c1, dx, below0=below_x)
    return dx, dprev


# ---- U-Net body ------------------------------------------------------------------------------------------------
def unet_fwd(net, x1, training, want_tape=True):
    t = Tape()
    x1 = as_act(x1)
    x2, t.d1 = convd_fwd(net.down1, x1, training, want_tape)
    x3, t.d2 = convd_fwd(net.down2, x2, training, want_tape)
    x4, t.d3 = convd_fwd(net.down3, x3, training, want_tape)
    x5, t.d4 = convd_fwd(net.down4, x4, training, want_tape)
    x, t.u1 = convu_fwd(net.up1, x5, x4, training, want_tape)
    x, t.u2 = convu_fwd(net.up2, x, x3, training, want_tape)
    x, t.u3 = convu_fwd(net.up3, x, x2, training, want_tape)
    x, t.u4 = convu_fwd(net.up4, x, x1, training, want_tape)
    return x, t


def unet_bwd(net, t, dfeat, need_dx1=True, decoder_done=None, mask_x=None, below_x1=None):
    """-> gradient wrt the activated x1 (None if not needed).  decoder_done(): called once up4 .. up1 have been queued (their
    parameter gradients are then complete: the data-parallel exchange of that range starts beside the encoder's backward)."""
    # (the output of up3 / up2 / up1 / down4 feeds only the next ConvU: its gradient carries that layer's BatchNorm statistics)
    d, g1 = convu_bwd(net.up4, t.u4, dfeat, below_x=t.u3.c3)
    d, g2 = convu_bwd(net.up3, t.u3, d, below_x=t.u2.c3)
    d, g3 = convu_bwd(net.up2, t.u2, d, below_x=t.u1.c3)
    g5, g4 = convu_bwd(net.up1, t.u1, d, below_x=t.d4.c3)
    if decoder_done is not None:
        decoder_done()
    # (the skip gradients g4 .. g1 get their pooled share added by the max-pool backward, which then also forms the BatchNorm-backward
    # reductions of the conv3 that produced the tensor: they come back as PreBN)
    g4 = convd_bwd(net.down4, t.d4, g5, g4, below=t.d3.c3)
    g3 = convd_bwd(net.down3, t.d3, g4, g3, below=t.d2.c3)
    g2 = convd_bwd(net.down2, t.d2, g3, g2, below=t.d1.c3)
    # mask_x: x1 came out of a ReLU (the teacher's fusion conv): its backward rides in the last kernel that writes g1;
    # below_x1: x1 came out of a conv + BatchNorm + ReLU layer (the segmentation net's inc block)
    g1 = convd_bwd(net.down1, t.d1, g2, g1, mask_x=mask_x is not None, below=below_x1)
    return g1 if need_dx1 else None


# ---- 1x1 heads: conv, ReLU, conv, ReLU, conv (algorithms.py:1006-1012) / conv, ReLU, conv (:1199-1200) -------------
def _head_fusable(seq, idxs, x):
    """The fused kernels (csrc/head.hip) cover 32 -> 32 -> 8 [-> nc <= 4] heads on maps with H*W % 32 == 0 whose parameters
    sit back to back in the flat buffers (they do: registration order); anything else takes the per-layer path."""
    ls = [seq[i] for i in idxs]
    if len(ls) not in (2, 3) or (ls[0].cin, ls[0].cout, ls[1].cin, ls[1].cout) != (32, 32, 32, 8):
        return None
    if len(ls) == 3 and not (ls[2].cin == 8 and 1 <= ls[2].cout <= 4):
        return None
    if any(l.k != 1 for l in ls) or (x.t.shape[2] * x.t.shape[3]) % 32 != 0:
        return None
    root = ls[0]._root
    off = root._offsets[root._pindex[id(ls[0].weight)]]
    for l in ls:
        for prm in (l.weight, l.bias):
            if root._offsets[root._pindex[id(prm)]] != off:
                return None
            off += prm.numel()
    return ls


def head_fwd(seq, x, idxs, want_tape=True):
    t = Tape()
    t.x, t.acts = as_act(x), []
    ls = _head_fusable(seq, idxs, t.x)
    t.fused = ls is not None
    if t.fused:
        w3, b3 = (ls[2].weight, ls[2].bias) if len(ls) == 3 else (None, None)
        out, t.h1, t.h2 = ops.head_fwd(t.x.t, t.x.pro, t.x.relu, ls[0].weight, ls[0].bias, ls[1].weight, ls[1].bias, w3, b3,
                                       want_tape, x_amax=act_amax(t.x))
        return out, t
    h = t.x
    for n, i in enumerate(idxs):
        last = n == len(idxs) - 1
        h, _ = _conv(seq[i], h, None, not last, False)
        t.acts.append(h)
    return h, t


def head_bwd(seq, t, d, idxs):
    """-> gradient wrt the activated head input."""
    if t.fused:
        ls = [seq[i] for i in idxs]
        root = ls[0]._root
        views = [root.gview(prm) for l in ls for prm in (l.weight, l.bias)]      # marks them reached; contiguous range
        total = sum(v.numel() for v in views)
        off = root._offsets[root._pindex[id(ls[0].weight)]]
        dparams = root._gtarget[off:off + total]
        return ops.head_bwd(d, t.x.t, t.x.pro, t.x.relu, t.h1, t.h2, ls[0].weight, ls[1].weight,
                            ls[2].weight if len(ls) == 3 else None, dparams, b1=ls[0].bias, x_amax=act_amax(t.x), dy_amax=_gamax(d))
    for n in reversed(range(len(idxs))):
        layer = seq[idxs[n]]
        inp = t.x if n == 0 else t.acts[n - 1]
        _wgrad(layer, d, inp)
        # the ReLU backward of the activation below rides in this data gradient's epilogue
        d, _ = _dgrad(layer, d, mask_ref=(t.acts[n - 1] if n > 0 else None))
    return d


# ---- DeepWT (algorithms.py:1091-1117) -----------------------------------------------------------------------------
# Gram partials of z1 / z2 in the epilogue of the convs that write them: the WT loss then never reads the maps from HBM
# (134 MB per map at B=32, 256x256).  WTPSE_WT_FUSED_GRAM=0: the stand-alone gram_partial_k pass.
WT_FUSED_GRAM = os.environ.get("WTPSE_WT_FUSED_GRAM", "1") != "0"


def _conv_gram(layer, a0):
    """3x3, 16 output channels, no BatchNorm: -> (z, (partial Grams, S))."""
    a0 = as_act(a0)
    out_amax = ops.fwd_amax_table(a0.t.device)       # z1 / z2 reach their x2h consumers as stored
    if layer.x16f_off >= 0:
        y, _, g = ops.conv16_x3(a0.t, layer._root.x3_ptr(layer.x16f_off), layer.bias, 16, a0.pro, _relu_bits(a0, None), False,
                                want_gram=True, in_amax=act_amax(a0), out_amax=out_amax)
        return y, g
    return ops.conv_fwd_gram(a0.t, layer._root.packed_ptr(layer.wf_off), layer.bias, a0.pro, _relu_bits(a0, None), False, out_amax)


def deepwt_fwd(wt, x, want_tape=True, want_gram=False):
    """want_gram: the caller computes the WT loss on z1 / z2 (training updates): t.g1 / t.g2 = their partial Grams."""
    t = Tape()
    a, b = wt.DoubleConv.double_conv, wt.DoubleConv2.double_conv
    t.x = x
    t.g1 = t.g2 = None
    fused = want_gram and WT_FUSED_GRAM and a[2].cout == 16 and a[2].k == 3 and b[2].cout == 16
    # (no BatchNorm anywhere in DeepWT: every map carries the amax of its stored data to its x2h consumers)
    t.h1, _ = _conv(a[0], x, None, True, want_amax=True)
    if fused:
        t.z1, t.g1 = _conv_gram(a[2], t.h1)
    else:
        t.z1, _ = _conv(a[2], t.h1, want_amax=True)
    t.h2, _ = _conv(b[0], Act(t.z1, None, True), None, True, want_amax=True)     # ReLU(z1) on load
    if fused:
        t.z2, t.g2 = _conv_gram(b[2], t.h2)
    else:
        t.z2, _ = _conv(b[2], t.h2, want_amax=True)
    return t


def deepwt_bwd(wt, t, dz2, dz1_extra=None):
    """dz2: total gradient wrt z2 (raw).  dz1_extra(dz1): callback adding the WT-loss gradient of z1 in place."""
    a, b = wt.DoubleConv.double_conv, wt.DoubleConv2.double_conv
    _wgrad(b[2], dz2, t.h2)
    d, _ = _dgrad(b[2], dz2, mask_ref=t.h2)
    _wgrad(b[0], d, Act(t.z1, None, True))
    dz1, _ = _dgrad(b[0], d, mask_ref=t.z1)
    if dz1_extra is not None:
        dz1_extra(dz1)
    _wgrad(a[2], dz1, t.h1)
    d, _ = _dgrad(a[2], dz1, mask_ref=t.h1)
    _wgrad(a[0], d, t.x)


# ---- teacher: ShapeVariationalDist_y_x (algorithms.py:1014-1033,1055-1075) ---------------------------------------------
def teacher_fwd(tn, feat, mask, training, want_logvar=True, want_tape=True):
    """feat: Act — z2 with ReLU-on-load, or an already-activated tensor."""
    t = Tape()
    inc = tn.inc.double_conv
    feat = as_act(feat)
    m1, t.i0 = convbn_fwd(inc[0], inc[1], mask, None, True, training, want_tape)
    m2, t.i3 = convbn_fwd(inc[3], inc[4], m1, None, True, training, want_tape)
    t.m2, t.feat = m2, feat
    t.xf, _ = _conv(tn.fusion[0], m2, feat, True, want_amax=True)
    fmap, t.unet = unet_fwd(tn, t.xf, training, want_tape)
    t.mu, t.hmu = head_fwd(tn.mu_prior, fmap, (0, 2, 4), want_tape)
    if want_logvar:
        t.logvar, t.hlv = head_fwd(tn.logvar_prior, fmap, (0, 2, 4), want_tape)
    return t


def teacher_bwd(tn, t, dmu, dlogvar):
    """-> gradient wrt the activated feat (i.e. wrt relu(z2) when feat carries ReLU-on-load)."""
    d = head_bwd(tn.mu_prior, t.hmu, dmu, (0, 2, 4))
    if dlogvar is not None:
        # (adding the second head's input gradient inside its kernel was measured: the loads in front of the stores cost the
        # MFMA-bound kernel 290 us, the separate 16-byte add 120)
        d2 = head_bwd(tn.logvar_prior, t.hlv, dlogvar, (0, 2, 4))
        ops.axpy(d, d2)
    dxf = unet_bwd(tn, t.unet, d, mask_x=t.xf)      # the ReLU of the fusion conv rides in the last data gradient's epilogue
    _wgrad(tn.fusion[0], dxf, t.m2, t.feat)
    dm2, dfeat = _dgrad(tn.fusion[0], dxf, split=t.m2.t.shape[1], below0=t.i3)
    inc = tn.inc.double_conv
    dm1, _ = convbn_bwd(inc[3], inc[4], t.i3, dm2, below0=t.i0)
    convbn_bwd(inc[0], inc[1], t.i0, dm1, need_dx=False)
    return dfeat
