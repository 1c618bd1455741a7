"""Drop-in for the reference's ``shape_networks.py`` on MI355X: ``ShapeVariationalDist_x`` with the reference's
constructor, ``update()`` / ``sample_forward()`` surface and ``state_dict`` keys
(reference shape_networks.py:415-597; consumed by train.py:99-100,113 and Trainer.py:816,901, algorithms.py:1333-1338).

The student distils the teacher's mean (KD = MSE of means) and carries its own WT loss.  Quirks kept on purpose
(SURVEY.md §8a-8, Appendix A): the ``ins_diag`` accumulator overwrite (:546-548), the 2-of-3 averaging (:551-554),
a hard-coded 3-domain MMD (:448), sampling ``z = normal(mu, std)*std + mu`` (:507-509).  Skipped because they cannot
change any result: the teacher's backward in this call (its gradients are zeroed before use, Trainer.py:767-768)
and the two attention calls whose outputs are discarded (:533-535).
"""
import torch
import torch.nn as nn

from wtpse_hip import nn as E
from wtpse_hip import ops

__all__ = ["ShapeVariationalDist_x"]


class _ShapeUpdateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, net, main_network, x, mask):
        scal, tape = net._forward_update(main_network, x, mask, want_tape=True)
        ctx.net, ctx.tape = net, tape
        return scal[0], scal[1], scal[2], scal[3], scal[4]

    @staticmethod
    def backward(ctx, d_kd, d_tot, d_off, d_diag, d_dom):
        # ins_total = ins_off + ins_diag: fold upstream gradients of the three instance outputs (device scalars)
        g_off = d_tot if d_off is None else (d_off if d_tot is None else d_tot + d_off)
        g_diag = d_tot if d_diag is None else (d_diag if d_tot is None else d_tot + d_diag)
        ctx.net._backward_update(ctx.tape, d_kd, g_off, g_diag, d_dom,
                                 w_kd=0.0 if d_kd is None else 1.0, w_off=0.0 if g_off is None else 1.0,
                                 w_diag=0.0 if g_diag is None else 1.0, w_dom=0.0 if d_dom is None else 1.0)
        ctx.tape = None
        return None, None, None, None, None


class ShapeVariationalDist_x(E.HipNet, E.UNetBody):
    def __init__(self, hparams, device, n_classes, number_source_domain=3, batch_size=3):
        super(ShapeVariationalDist_x, self).__init__()
        self.device = device
        self.batch_size = batch_size
        self.hparams = hparams
        self.wt = hparams['whitening']
        self.whitening = hparams['whitening']
        self.momentum = 0.99
        self.number_source_domain = number_source_domain
        self.eps = 1e-5
        self.margin = hparams['margin']
        if not self.wt:
            raise NotImplementedError("the shape network is only constructed with whitening=True in the reference flow "
                                      "(train.py:99; Trainer.py:810 skips it otherwise)")
        n = 16
        self.dim = n
        self.wt_model = E.DeepWTP(3, n)
        self._make_body(n)
        self.mu_prior = E.head_p(n, n_classes)
        self.logvar_prior = E.head_p(n, n_classes)
        self._finish_init()

    # ------------------------------------------------------------------------------------------------ public API
    def update(self, main_network, inputs, mask, step=0, plot_show=0, two_stage_inputs=None, two_step=False):
        """Reference shape_networks.py:512-558 -> (kd, ins_total, ins_offdiag, ins_diag, dom) scalar tensors."""
        self.ensure_ready(repack=True)
        main_network.ensure_ready(repack=True)
        x = two_stage_inputs if (two_step and two_stage_inputs is not None) else inputs
        x = self._as_input(x)
        mask = self._as_input(mask)
        if torch.is_grad_enabled():
            return _ShapeUpdateFn.apply(self._get_anchor(), self, main_network, x, mask)
        scal, _ = self._forward_update(main_network, x, mask, want_tape=False)
        return scal[0], scal[1], scal[2], scal[3], scal[4]

    def sample_forward(self, inputs, training):
        """Reference shape_networks.py:483-500.  `inputs` = W[-1] = relu(z2), materialised by the caller."""
        self.ensure_ready(repack=True)
        x = self._as_input(inputs)
        if getattr(inputs, "wt_amax", None) is not None and x.data_ptr() == inputs.data_ptr():
            x.wt_amax = inputs.wt_amax          # (the caller's W[-1] from DeepWTP.forward carries the amax table of its data)
        with ops.fwd_scope(x.device):
            mu, fmap = self._student_mu(E.Act(x), self.training, None, want_fmap=True)
        if not training:
            return mu
        logvar, _ = E.head_fwd(self.logvar_prior, fmap, (0, 2, 4), False)
        eps = self.next_noise(mu.shape)
        return ops.reparam_student(mu, logvar, eps, self._flag), mu

    def compute_whitening_loss(self, z):
        """Reference shape_networks.py:561-594 -> (ins_offdiag, ins_diag, domain), connected to autograd through `z`."""
        from algorithms import _WtLossFn
        return _WtLossFn.apply(z, 3, self.batch_size, float(self.margin), self.eps)

    # ------------------------------------------------------------------------------------------------ schedules
    def _as_input(self, t):
        if not t.is_cuda:
            raise RuntimeError("ShapeVariationalDist_x on MI355X takes device tensors (got %s)" % t.device)
        return t.detach().to(torch.float32).contiguous()

    def _get_anchor(self):
        a = self.__dict__.get("_anchor")
        if a is None or a.device != self._flat.device:
            a = torch.zeros(1, device=self._flat.device, requires_grad=True)
            object.__setattr__(self, "_anchor", a)
        return a

    def _student_mu(self, feat, training, tape, want_fmap=False):
        """unet_extractor + mu_prior + NaN scrub (shape_networks.py:468-492).  feat: Act (z2 with ReLU-on-load, or activated)."""
        want = tape is not None
        fmap, c_unet = E.unet_fwd(self, feat, training, want)
        mu, c_mu = E.head_fwd(self.mu_prior, fmap, (0, 2, 4), want)
        ops.nan_scrub_(mu, self._flag)
        if want:
            tape.unet, tape.hmu = c_unet, c_mu
        return (mu, fmap) if want_fmap else mu

    def _forward_update(self, main_network, x, mask, want_tape):
        # one scope of amax tables per forward pass (see WT_PSE._forward_update)
        with ops.fwd_scope(x.device):
            return self._forward_update_body(main_network, x, mask, want_tape)

    def _forward_update_body(self, main_network, x, mask, want_tape):
        t = E.Tape()
        training = self.training
        # teacher side: forward only (train-mode BatchNorm still advances its running statistics, as in the reference).
        # It shares nothing with the student's forward but the inputs, keeps no tape and uses no scratch buffers, so it runs
        # on a second stream beside it (the small-grid layers of the two U-Nets fill each other's idle CUs); its one
        # result, mu_t, is handed to the main stream at the join.
        def teacher():
            w1 = E.deepwt_fwd(main_network.wt_model, x, want_tape=False)
            th = E.teacher_fwd(main_network.prior_dist, E.Act(w1.z2, None, True), mask, main_network.training,
                               want_logvar=False, want_tape=False)
            return th.mu
        synced = main_network._dp is not None and main_network._dp.bn_sync      # collectives stay on one stream
        side = None if synced else E.second_stream(x.device)
        if side is not None:
            main = torch.cuda.current_stream()
            E.stream_wait(side, main)
            with torch.cuda.stream(side):
                mu_t = teacher()
        else:
            mu_t = teacher()
        # student side
        w2 = E.deepwt_fwd(self.wt_model, x, want_tape, want_gram=not (self._dp is not None and self._dp.exact))
        mu_s = self._student_mu(E.Act(w2.z2, None, True), training, t if want_tape else None)
        if side is not None:
            E.stream_wait(main, side)
            mu_t.record_stream(main)
        scal = torch.empty((5,), dtype=torch.float32, device=x.device)   # (kd, ins_total, ins_off, ins_diag, dom)
        ops.mse_fwd(mu_t, mu_s, out=scal[0:1])
        losses = torch.empty((2, 3), dtype=torch.float32, device=x.device)
        n = self.batch_size
        st1 = self._wt_loss(w2.z1, 3, n, losses[0], w2.g1)
        st2 = self._wt_loss(w2.z2, 3, n, losses[1], w2.g2)
        ops.wt_combine(losses, 3.0, 1, out=scal[1:5])
        if want_tape:
            t.w2, t.mu_t, t.mu_s, t.st1, t.st2 = w2, mu_t, mu_s, st1, st2
        return scal, t

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

    def _backward_update(self, t, g_kd=None, g_off=None, g_diag=None, g_dom=None, w_kd=1.0, w_off=1.0, w_diag=1.0, w_dom=1.0):
        """g_*: device scalars (None -> 1), w_*: host weights.  Instance terms follow the reference's bookkeeping:
        ins_off = (off_1 + off_2)/3, ins_diag = 2*diag_2/3 (accumulator overwrite), dom = (dom_1 + dom_2)/3."""
        self.begin_backward()
        c = lambda g: g.contiguous() if g is not None else None
        g_kd, g_off, g_diag, g_dom = c(g_kd), c(g_off), c(g_diag), c(g_dom)
        dmu = ops.mse_bwd(t.mu_s, t.mu_t, g_kd, w_kd)
        dfmap = E.head_bwd(self.mu_prior, t.hmu, dmu, (0, 2, 4))
        # data-parallel overlap: up1 .. logvar_prior (registration order: the decoder and the two heads) are complete after the
        # decoder's backward and go out beside the encoder's and DeepWT's
        d_relu_z2 = E.unet_bwd(self, t.unet, dfmap, decoder_done=lambda: self.grads_ready(self.up1, self.logvar_prior))
        dz2 = d_relu_z2                                      # masked with [z2 > 0] inside the WT-loss backward pass
        self._wt_loss_bwd(t.st2, dz2, mask_in=True, g_off=g_off, g_diag=g_diag, g_dom=g_dom, w_off=w_off / 3.0, w_diag=2.0 * w_diag / 3.0,
                          w_dom=w_dom / 3.0)
        E.deepwt_bwd(self.wt_model, t.w2, dz2,
                     lambda dz1: self._wt_loss_bwd(t.st1, dz1, g_off=g_off, g_diag=g_diag, g_dom=g_dom, w_off=w_off / 3.0,
                                                   w_diag=0.0, w_dom=w_dom / 3.0))
        self.end_backward()
