"""Drop-in for the reference's ``algorithms.py`` on MI355X: same module name, same ``get_algorithm_class`` lookup,
same ``WT_PSE`` constructor / ``update()`` / ``predict()`` signature and returns, same ``state_dict`` keys
(reference algorithms.py:21-25, 1134-1353; consumed by train.py:82-98 and Trainer.py:779,856,170,182).

Underneath, every operation is a hand-written gfx950 kernel reached through the C ABI of libwtpse_hip.so
(include/wtpse_hip.h); torch provides device memory, the stream and one autograd node per ``update()``.
There is no CPU path: modules must live on a HIP device.

Only the reachable configuration space of the reference is supported: ``shape_prior`` and ``whitening`` both on
(the WT-PSE method) or both off (plain segmentation U-Net, BASELINE.json configs[1]); the mixed settings crash in
the reference itself, and so does ``shape_attention=False`` (see ``__init__``).  The reachable
non-default branch ``cat_shape=True`` (``outc`` over ``cat(fuse_embedding, z_posterior)``, reference :1192,1253,1348)
is built and pinned by tests/golden/catshape.npz.
"""
import torch
import torch.nn as nn

from wtpse_hip import nn as E
from wtpse_hip import ops

__all__ = ["get_algorithm_class", "WT_PSE", "compute_MMD"]


def get_algorithm_class(algorithm_name):
    """Return the algorithm class with the given name (reference algorithms.py:21-25)."""
    if algorithm_name not in globals():
        raise NotImplementedError("Algorithm not found: {}".format(algorithm_name))
    return globals()[algorithm_name]


class compute_MMD(object):
    """Pairwise Gaussian-kernel MMD over contiguous domain row blocks (reference algorithms.py:59-121)."""

    def __init__(self, domain_num, batch_size):
        self.domain_num = domain_num
        self.batch_size = batch_size
        self.kernel_type = "gaussian"

    def forward(self, inputs, **kwargs):
        v = inputs.contiguous()
        R = self.domain_num * self.batch_size
        rowval = torch.empty((R,), dtype=torch.float64, device=v.device)
        dv = torch.empty((R, 120), dtype=torch.float32, device=v.device)
        ops.lib().call("wtpse_mmd_fwd", v.data_ptr(), self.domain_num, self.batch_size, rowval.data_ptr(), dv.data_ptr(),
                       ops.stream_ptr())
        out = torch.empty((1,), dtype=torch.float32, device=v.device)
        tmp = rowval.to(torch.float32).reshape(R, 1).contiguous()
        ops.lib().call("wtpse_reduce_rows", tmp.data_ptr(), R, 1, out.data_ptr(), 0, 1.0, ops.stream_ptr())
        return out[0]


class _WtLossFn(torch.autograd.Function):
    """compute_whitening_loss as an autograd node: forward = fused Gram + masked L1 + MMD kernels, backward = gram_bwd_k with
    the three upstream scalars read on the device (no host sync).  -> (ins_offdiag, ins_diag, domain)."""

    @staticmethod
    def forward(ctx, z, domain_num, per_domain, margin, eps):
        zc = z.detach().to(torch.float32).contiguous()
        st = ops.wt_loss_fwd(zc, domain_num, per_domain, margin, eps)
        ctx.st = st
        return st.losses[0].clone(), st.losses[1].clone(), st.losses[2].clone()

    @staticmethod
    def backward(ctx, g_off, g_diag, g_dom):
        st = ctx.st
        dz = torch.empty_like(st.z)
        c = lambda g: None if g is None else g.detach().to(torch.float32).contiguous()
        ops.wt_loss_bwd(st, dz, False, c(g_off), c(g_diag), c(g_dom), w_off=0.0 if g_off is None else 1.0,
                        w_diag=0.0 if g_diag is None else 1.0, w_dom=0.0 if g_dom is None else 1.0)
        return dz, None, None, None, None


class _UpdateFn(torch.autograd.Function):
    """One autograd node per WT_PSE.update(): forward runs the fused schedule and keeps a tape, backward replays it."""

    @staticmethod
    def forward(ctx, anchor, net, inputs, mask, wt_inputs):
        res, tape = net._forward_update(inputs, mask, wt_inputs, want_tape=True)
        ctx.net, ctx.tape = net, tape
        if tape.shape_prior:
            out, att_mask, scal = res
            ctx.mark_non_differentiable(att_mask)
            return out, att_mask, scal[0], scal[3]
        return res[0]

    @staticmethod
    def backward(ctx, d_out, *rest):
        d_ins = rest[1] if len(rest) > 1 else None
        d_dom = rest[2] if len(rest) > 2 else None
        # a missing upstream gradient means the caller's loss does not use that output: weight 0
        ctx.net._backward_update(ctx.tape, d_out, d_ins, d_dom, w_ins=0.0 if d_ins is None else 1.0,
                                 w_dom=0.0 if d_dom is None else 1.0)
        ctx.tape = None
        return None, None, None, None, None


class WT_PSE(E.HipNet, E.UNetBody):
    def __init__(self, n_channels, n_classes, hparams, device, two_step, per_domain_batch=8, source_domain_num=3,
                 feature_dim=8, bilinear=True):
        super(WT_PSE, self).__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.device = device
        self.hparams = hparams
        self.two_step = two_step
        self.per_domain_batch = per_domain_batch
        self.number_source_domain = source_domain_num
        self.num_domains = 3
        self.feature_dim = feature_dim
        self.bilinear = bilinear
        self.eps = 1e-5
        self.dim = 16
        self.whitening = hparams['whitening']
        self.start_shape_step = hparams['shape_start']
        self.cat_shape = hparams['cat_shape']
        self.margin = hparams['margin']
        self.mmd_operator = compute_MMD(domain_num=source_domain_num, batch_size=per_domain_batch)
        if bool(hparams['shape_prior']) != bool(hparams['whitening']):
            raise NotImplementedError("shape_prior and whitening must be switched together (the mixed settings fail in "
                                      "the reference: algorithms.py:994,1022-1023,1235)")
        if not hparams['shape_attention']:
            raise NotImplementedError("shape_attention=False fails in the reference itself: update() returns "
                                      "z_posterior_attention_mask and predict() no_sigmoid_embeddings, which only the "
                                      "attention branch assigns (algorithms.py:1241-1272, 1340-1352: UnboundLocalError)")
        n = 16
        # registration order == reference state_dict order (algorithms.py:1161-1204)
        if self.whitening:
            self.wt_model = E.DeepWTP(3, n)
        self.inc = E.ConvDBlock(n_channels, n, first=True)
        self._make_body(n)
        if hparams['shape_prior']:
            self.prior_dist = E.TeacherP(n)
        self.mu = E.Seq(_0=E.ConvP(2 * n, 2 * n, 1), _2=E.ConvP(2 * n, feature_dim, 1))
        # cat_shape (algorithms.py:1190-1193,1253,1348): outc also sees z_posterior as one more input channel
        fuse_dim = feature_dim + 1 if (hparams['shape_prior'] and self.cat_shape) else feature_dim
        self.outc = E.Seq(_0=E.ConvP(fuse_dim, n_classes, 1))
        self.attention_layer = E.AttentionP()
        self.global_step = 0
        self._finish_init()

    # ------------------------------------------------------------------------------------------------ public API
    def update(self, inputs, mask, step=0, plot_show=0, two_stage_inputs=None, sp_mask=None, two_step=False):
        """Reference algorithms.py:1216-1275.  Returns (logits, att_mask, att_mask, ins_wt_loss, dom_wt_loss)."""
        self.ensure_ready(repack=True)
        inputs = self._as_input(inputs)
        mask = self._as_input(mask)
        wt_in = self._as_input(two_stage_inputs) if (two_step and two_stage_inputs is not None) else inputs
        if torch.is_grad_enabled():
            res = _UpdateFn.apply(self._get_anchor(), self, inputs, mask, wt_in)
            if self.hparams['shape_prior']:
                out, att_mask, ins, dom = res
                return out, att_mask, att_mask, ins, dom
            return res, 0, 0, 0, 0
        res, _ = self._forward_update(inputs, mask, wt_in, want_tape=False)
        if self.hparams['shape_prior']:
            out, att_mask, scal = res
            return out, att_mask, att_mask, scal[0], scal[3]
        return res[0], 0, 0, 0, 0

    def predict(self, learn_x_network, inputs_all):
        """Reference algorithms.py:1311-1353 (uses the STUDENT's wt_model and shape net). -> (logits, pre-sigmoid attention)."""
        self.ensure_ready(repack=True)
        if self.two_step:
            inputs, wt_in = self._as_input(inputs_all[0]), self._as_input(inputs_all[1])
        else:
            inputs = wt_in = self._as_input(inputs_all)
        training = self.training
        with ops.fwd_scope(inputs.device):           # the amax tables of this pass's activations (x2h arithmetic)
            emb = self._embedding(inputs, training, None)
            if not self.hparams['shape_prior']:
                out, _ = E._conv(self.outc[0], emb)
                return out, None
            learn_x_network.ensure_ready(repack=True)
            w = E.deepwt_fwd(learn_x_network.wt_model, wt_in, want_tape=False)
            z = learn_x_network._student_mu(E.Act(w.z2, None, True), learn_x_network.training, None)
            _, pre, _, fuse = ops.attn_fuse_fwd(z, self.attention_layer.layer1.weight.data_ptr(), emb,
                                                float(self.hparams['shape_attention_coeffient']), False, True, False)
            out, _ = E._conv(self.outc[0], self._outc_input(fuse, z))
            return out, pre

    def _outc_input(self, fuse, z):
        """cat_shape: torch.cat([fuse_embedding, z_posterior], 1) (algorithms.py:1253,1348).  This non-default branch
        materialises the 9-channel tensor (a device copy, no arithmetic): the two-pointer loader wants its first part in
        whole 16-channel chunks."""
        return torch.cat([fuse, z], 1) if self.cat_shape else fuse

    def compute_whitening_loss(self, z):
        """Reference algorithms.py:1277-1309 -> (instance_loss, domain_loss), connected to autograd through `z`
        (gradients flow as in the reference when a caller differentiates the returned scalars)."""
        ins_off, ins_diag, dom = _WtLossFn.apply(z, self.number_source_domain, self.per_domain_batch, float(self.margin), self.eps)
        return ins_off + ins_diag, dom

    # ------------------------------------------------------------------------------------------------ schedules
    def _as_input(self, t):
        if not t.is_cuda:
            raise RuntimeError("WT_PSE on MI355X takes device tensors (got %s)" % t.device)
        return t.detach().to(torch.float32).contiguous()

    def _get_anchor(self):
        a = self.__dict__.get("_anchor")
        if a is None or a.device != self._flat.device:
            a = torch.zeros(1, device=self._flat.device, requires_grad=True)
            object.__setattr__(self, "_anchor", a)
        return a

    def _embedding(self, inputs, training, tape):
        want = tape is not None
        x1, c_inc = E.convd_fwd(self.inc, inputs, training, want)
        feat, c_unet = E.unet_fwd(self, x1, training, want)
        emb, c_mu = E.head_fwd(self.mu, feat, (0, 2), want)
        if want:
            tape.inc, tape.unet, tape.mu = c_inc, c_unet, c_mu
        return emb

    def _forward_update(self, inputs, mask, wt_in, want_tape):
        # one scope of amax tables per forward pass (x2h arithmetic: ops.fwd_scope), opened on the stream the pass starts on — in
        # front of the fork to the second stream — and kept alive by the activations on the tape until the backward pass is done
        with ops.fwd_scope(inputs.device):
            return self._forward_update_body(inputs, mask, wt_in, want_tape)

    def _forward_update_body(self, inputs, mask, wt_in, want_tape):
        hp = self.hparams
        t = E.Tape()
        t.shape_prior = bool(hp['shape_prior'])
        training = self.training
        if not t.shape_prior:
            emb = self._embedding(inputs, training, t if want_tape else None)
            out, _ = E._conv(self.outc[0], emb)
            t.emb = emb
            return (out,), t
        coef = float(hp['shape_attention_coeffient'])
        D, n = self.number_source_domain, self.per_domain_batch

        def prior_chain():
            """DeepWT -> teacher -> posterior sample, and the WT loss on the first two maps divided by
            len([z1, z2, relu(z2)]) = 3 (algorithms.py:1259-1267).  Shares only the inputs with the embedding above."""
            w = E.deepwt_fwd(self.wt_model, wt_in, want_tape, want_gram=not (self._dp is not None and self._dp.exact))
            th = E.teacher_fwd(self.prior_dist, E.Act(w.z2, None, True), mask, training, True, want_tape)
            eps = self.next_noise(th.mu.shape)
            z_post = ops.reparam_fwd(th.mu, th.logvar, eps)
            losses = torch.empty((2, 3), dtype=torch.float32, device=inputs.device)
            st1 = self._wt_loss(w.z1, D, n, losses[0], w.g1)
            st2 = self._wt_loss(w.z2, D, n, losses[1], w.g2)
            return w, th, eps, z_post, st1, st2, ops.wt_combine(losses, 3.0, 0)

        # The prior chain runs on the second stream beside the segmentation U-Net (small-grid layers of the two networks
        # fill each other's idle CUs); what the main stream reads of it afterwards (z_post, the loss scalars) is handed over
        # at the join, everything else is only touched again by the chain's own backward, on the same stream.
        second = None if (self._dp is not None and self._dp.exact) else E.second_stream(inputs.device)
        if second is not None:
            main = torch.cuda.current_stream()
            E.stream_wait(second, main)
            with torch.cuda.stream(second):
                w, th, eps, z_post, st1, st2, scal = prior_chain()
        emb = self._embedding(inputs, training, t if want_tape else None)
        if second is None:
            w, th, eps, z_post, st1, st2, scal = prior_chain()
        else:
            E.stream_wait(main, second)
            z_post.record_stream(main)
            scal.record_stream(main)
        att, _, att_mask, fuse = ops.attn_fuse_fwd(z_post, self.attention_layer.layer1.weight.data_ptr(), emb, coef,
                                                   True, False, True)
        fuse_in = self._outc_input(fuse, z_post)
        out, _ = E._conv(self.outc[0], fuse_in)
        if want_tape:
            t.w, t.th, t.eps, t.z_post, t.att, t.emb, t.fuse, t.st1, t.st2, t.coef = w, th, eps, z_post, att, emb, fuse_in, st1, st2, coef
        return (out, att_mask, scal), t

    def _wt_loss(self, z, D, n, losses_out, gram=None):
        if self._dp is not None and self._dp.exact:
            return self._dp.wt_loss_fwd(z, D, n, float(self.margin), self.eps, losses_out)
        return ops.wt_loss_fwd(z, D, n, float(self.margin), self.eps, losses_out, gram)

    def _wt_loss_bwd(self, st, dz, mask_in=False, **kw):
        """mask_in: dz arrives as the gradient wrt relu(z) and is masked with [z > 0] inside the same pass."""
        acc = 3 if mask_in else 1
        if self._dp is not None:
            return self._dp.wt_loss_bwd(st, dz, acc, **kw)
        return ops.wt_loss_bwd(st, dz, acc, **kw)

    def _backward_update(self, t, d_out, d_ins=None, d_dom=None, w_ins=1.0, w_dom=1.0):
        """d_out: gradient wrt the logits; d_ins / d_dom: device scalars (None -> 1) scaled by the host weights
        w_ins / w_dom (0 disables a term).  Writes the parameter gradients into the flat gradient buffer."""
        self.begin_backward()
        if d_out is None:
            d_out = ops.zero_(torch.empty((t.emb.shape[0], self.n_classes) + tuple(t.emb.shape[2:]), dtype=torch.float32, device=t.emb.device))
        d_out = d_out.contiguous()
        outc = self.outc[0]
        if not t.shape_prior:
            E._wgrad(outc, d_out, t.emb)
            demb, _ = E._dgrad(outc, d_out)
        else:
            E._wgrad(outc, d_out, t.fuse)
            dfuse, _ = E._dgrad(outc, d_out)
            dz_cat = None
            if self.cat_shape:          # gradient of the concat: the first feature_dim channels belong to fuse, the last to z_posterior
                dz_cat = dfuse[:, self.feature_dim:].contiguous()
                dfuse = dfuse[:, :self.feature_dim].contiguous()
            al = self.attention_layer.layer1
            d_wb = self.gview(al.weight)
            self.gview(al.bias)
            demb, dz_post = ops.attn_fuse_bwd(dfuse, t.z_post, t.emb, t.att, al.weight.data_ptr(), t.coef, d_wb.data_ptr(), True)
            if dz_cat is not None:
                ops.axpy(dz_post, dz_cat)
            gi = d_ins.contiguous() if d_ins is not None else None
            gd = d_dom.contiguous() if d_dom is not None else None
            kw = dict(g_off=gi, g_diag=gi, g_dom=gd, w_off=w_ins / 3.0, w_diag=w_ins / 3.0, w_dom=w_dom / 3.0)

            def prior_chain_bwd():
                dlogvar = ops.reparam_bwd(dz_post, t.th.logvar, t.eps)
                d_relu_z2 = E.teacher_bwd(self.prior_dist, t.th, dz_post, dlogvar)
                self.grads_ready(self.prior_dist)           # data-parallel overlap: 12.7 MB go out beside DeepWT's backward
                dz2 = d_relu_z2                              # masked with [z2 > 0] inside the WT-loss backward pass
                self._wt_loss_bwd(t.st2, dz2, mask_in=True, **kw)
                E.deepwt_bwd(self.wt_model, t.w, dz2, lambda dz1: self._wt_loss_bwd(t.st1, dz1, **kw))

            # the prior chain's backward on the second stream beside the segmentation U-Net's (see _forward_update); it
            # only produces parameter gradients, which end_backward() waits for
            second = None if (self._dp is not None and self._dp.exact) else E.second_stream(dz_post.device)
            if second is not None:
                main = torch.cuda.current_stream()
                E.stream_wait(second, main)
                with torch.cuda.stream(second):
                    prior_chain_bwd()
                for g in (dz_post, gi, gd):
                    if g is not None:
                        g.record_stream(second)
                E.note_join(self, second)
            else:
                prior_chain_bwd()
        dfeat = E.head_bwd(self.mu, t.mu, demb, (0, 2))
        self.grads_ready(self.mu, self.attention_layer)      # heads: mu, outc, attention_layer (registration order)
        dx1 = E.unet_bwd(self, t.unet, dfeat, decoder_done=lambda: self.grads_ready(self.up1, self.up4), below_x1=t.inc.c3)
        E.convd_bwd(self.inc, t.inc, dx1, need_dx=False)
        self.end_backward()
