"""The training step: counterpart of the body of the reference's hot loop (Trainer.py:766-914).

One `TrainStep.step()` = the four `update()` calls A-D, their four backward passes and four Adam steps over the
four networks of train.py:91-138, in the reference's order:

    A  seg-net OD   : BCELoss(sigmoid(out), target_od) + gm_i*ins + gm_d*dom            (Trainer.py:779-805)
    B  shape-net OD : kd + gm_i*ins_total + gm_d*dom                                     (:810-825)
       ROI          : od_pred = sigmoid(out) > 0.75 ; roi = (image+1)*od_pred - 1       (:842-853)
    C  seg-net OC   : BCEWithLogits(out*od_pred, target_oc, pos_weight) + wt terms       (:856-892)
    D  shape-net OC : as B on the ROI                                                    (:894-914)

Differences from the reference loop that cannot change a result: no per-iteration `.item()` host syncs or
tensorboard scalars (losses stay on the device), fused Adam over the flat parameter buffer instead of ~390
per-tensor updates, and the dead work listed in shape_networks.py's header is skipped.

Replaying a recorded step: a step is ~1 900 kernel launches from one Python thread (tens of milliseconds of host time,
about as long as the GPU needs for them).  The C ABI allocates nothing, never synchronises and takes nothing that changes
from step to step by value (Adam's step number and the Philox stream position live in device memory), so a step can be
recorded once and replayed:
  * `TrainStep(..., graph="plan")`: the step is recorded under stream capture — which pins every device address: the
    allocator serves the capture from a private pool and defers cross-stream frees — into native launch plans
    (csrc/plan.hip: entry point + argument values + stream of every call, plus the cross-stream waits) and replayed with one
    host call per plan; the captured HIP graph itself is only kept alive as the owner of the memory pool.
  * `TrainStep(..., graph=True)`: the captured HIP graphs themselves are replayed (hipGraphLaunch).  Measured slower than
    eager launches on this runtime (profiles/r02_hipgraph_vs_eager.txt); kept for comparison.
With data-parallel training the gradient all-reduces stay outside (RCCL runs them eagerly between the replayed stretches):
a step is then five plans / graphs with four collectives between them.
"""
import torch

from . import ops


class FlatAdam:
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0) over a network's flat buffers: one launch per step.
    The step count is a device integer (see wtpse_adam in include/wtpse_hip.h)."""

    def __init__(self, net, lr=5e-4, betas=(0.9, 0.99), eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, lr, betas, eps
        self.m = self.v = self.t_dev = None

    def ready(self):
        if self.m is None:
            p = self.net.flat_params()
            self.m = ops.zero_(torch.empty_like(p))
            self.v = ops.zero_(torch.empty_like(p))
            self.t_dev = torch.zeros(1, dtype=torch.int32, device=p.device)      # completed steps

    def step(self):
        net = self.net
        p, g = net.flat_params(), net.flat_grads()
        self.ready()
        ops.adam_step(p, g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, 1, self.t_dev)
        ops.counter_add(self.t_dev, 1)
        net.invalidate_packed()
        net.ensure_ready()

    @property
    def t(self):
        return 0 if self.t_dev is None else int(self.t_dev.item())


class TrainStep:
    def __init__(self, model_od, shape_od, model_oc, shape_oc, hparams, lr=5e-4, betas=(0.9, 0.99), dp=None, graph=False):
        self.hp = hparams
        self.full = bool(hparams['whitening'])
        self.nets = [model_od, model_oc] + ([shape_od, shape_oc] if self.full else [])
        self.model_od, self.shape_od, self.model_oc, self.shape_oc = model_od, shape_od, model_oc, shape_oc
        self.dp = dp
        for n in self.nets:
            n.train()
            n.ensure_ready(repack=True)
            object.__setattr__(n, "_packed_valid", True)     # this harness owns the optimiser and repacks after each step
            object.__setattr__(n, "_attach_grads", False)
            object.__setattr__(n, "_dp", dp)
            object.__setattr__(n, "_defer_allreduce", True)  # the gradient exchange is issued here, between backward and Adam
        if dp is not None:
            dp.broadcast_params(self.nets)
        self.opt = {id(n): FlatAdam(n, lr, betas) for n in self.nets}
        for o in self.opt.values():
            o.ready()
        self.last_od_pred = None
        # exact data-parallel mode runs ~100 small collectives inside every forward: it stays eager
        self.graph = bool(graph) and not (dp is not None and dp.exact)
        self.plan = self.graph and graph == "plan"
        self._graphs = None
        self._static = None

    # ------------------------------------------------------------------------------------------------ schedule
    def _seg_call(self, model, x, target, noise, od_pred):
        """forward + loss + backward of one segmentation network, then (after the caller's gradient exchange) Adam.
        Generator: yields the network whose flat gradient is ready to be exchanged; returns (logits, losses)."""
        gi, gd = float(self.hp['instance_wt_gm']), float(self.hp['domain_wt_gm'])
        if noise is not None:
            model.set_noise([noise])
        res, tape = model._forward_update(x, target, x, want_tape=True)
        out = res[0]
        if od_pred is None:
            loss = ops.bce_sigmoid_fwd(out, target)
            d_out = ops.bce_sigmoid_bwd(out, target)
        else:
            sums, pw = ops.pos_weight_sums(od_pred, target)
            if self.dp is not None and self.dp.exact:      # pos_weight over the global batch (Trainer.py:865)
                pw = ops.pos_weight_from_sums(self.dp.allreduce_sum(sums))
            loss = ops.bce_logits_pw_fwd(out, od_pred, target, pw)
            d_out = ops.bce_logits_pw_bwd(out, od_pred, target, pw)
        model._backward_update(tape, d_out, None, None, w_ins=gi, w_dom=gd)
        del tape
        yield model
        self.opt[id(model)].step()
        r = {"seg": loss}
        if self.full:
            r["ins"], r["dom"] = res[2][0], res[2][3]
        return out, r

    def _shape_call(self, shape, model, x, target):
        gi, gd = float(self.hp['instance_wt_gm']), float(self.hp['domain_wt_gm'])
        r = None
        for _ in range(int(self.hp['multi-turn'])):
            scal, tape = shape._forward_update(model, x, target, want_tape=True)
            shape._backward_update(tape, None, None, None, None, w_kd=1.0, w_off=gi, w_diag=gi, w_dom=gd)
            del tape
            yield shape
            self.opt[id(shape)].step()
            r = {"kd": scal[0], "ins_total": scal[1], "ins_off": scal[2], "ins_diag": scal[3], "dom": scal[4]}
        return r

    def _schedule(self, image, target_od, target_oc, noise):
        """The step as a generator that yields at the points where a network's gradient is complete and not yet used."""
        out, ra = yield from self._seg_call(self.model_od, image, target_od, noise.get("a"), None)
        res = {"seg_od": ra["seg"]}
        if self.full:
            res.update(ins_od=ra["ins"], dom_od=ra["dom"])
            rb = yield from self._shape_call(self.shape_od, self.model_od, image, target_od)
            res.update(kd_od=rb["kd"], ins_shape_od=rb["ins_total"], ins_ij_od=rb["ins_off"], ins_ii_od=rb["ins_diag"],
                       dom_shape_od=rb["dom"])
        roi, od_pred = ops.roi(image, out)
        self.last_od_pred = od_pred          # [B,1,H,W] in {0,1}: lets a caller see how much of the image the ROI keeps
        out_oc, rc = yield from self._seg_call(self.model_oc, roi, target_oc, noise.get("c"), od_pred)
        res["seg_oc"] = rc["seg"]
        if self.full:
            res.update(ins_oc=rc["ins"], dom_oc=rc["dom"])
            rd = yield from self._shape_call(self.shape_oc, self.model_oc, roi, target_oc)
            res.update(kd_oc=rd["kd"], ins_shape_oc=rd["ins_total"], dom_shape_oc=rd["dom"])
        return res

    def _exchange(self, net):
        if self.dp is not None:
            self.dp.allreduce_grads(net, net.flat_grads())

    # ------------------------------------------------------------------------------------------------ hipGraph
    def _capture(self, image, target_od, target_oc):
        """Record the step into HIP graphs / launch plans (one per stretch between gradient exchanges).  Nothing executes
        while a stretch is recorded; the eager collectives between two stretches run on stale buffers and are harmless."""
        self._static = tuple(t.clone() for t in (image, target_od, target_oc))
        pool = torch.cuda.graph_pool_handle()
        self._cap_stream = torch.cuda.Stream(device=image.device)
        gen = self._schedule(*self._static, {})
        graphs, res = [], None
        L = ops.lib()
        done = False
        # the recorded launches' ticket slices: a buffer of the recording's own, never the eager ring (ops.ticket_scope)
        self._ticket_scope = ops.ticket_scope(image.device)
        with self._ticket_scope:
            self._capture_stretches(gen, pool, L, graphs)
        res = self._capture_result
        self._graphs, self._result = graphs, res

    def _capture_stretches(self, gen, pool, L, graphs):
        res, done = None, False
        while not done:
            g = torch.cuda.CUDAGraph()
            net = plan = None
            if self.plan:
                L.plan_begin()
            try:
                # thread_local: another thread's HIP calls (the RCCL watchdog polling the events of the parameter broadcast that
                # ran before the capture) must not invalidate the capture — in "global" mode they did, now and then (status 901)
                with torch.cuda.graph(g, pool=pool, stream=self._cap_stream, capture_error_mode="thread_local"):
                    try:
                        net = next(gen)
                    except StopIteration as stop:
                        res, done = stop.value, True
                    except BaseException:
                        # leaving a broken capture can crash the runtime before Python reports anything: say why first
                        import traceback
                        traceback.print_exc()
                        raise
            finally:
                if self.plan:
                    plan = L.plan_end()
            graphs.append((g, net, plan))
            if net is not None:
                self._exchange(net)
        self._capture_result = res

    def _replay(self):
        L = ops.lib()
        for g, net, plan in self._graphs:
            if plan is not None:
                cur = torch.cuda.current_stream()
                self._cap_stream.wait_stream(cur)       # the plan's streams start behind the caller's stream ...
                try:
                    L.plan_replay(plan)
                except Exception:
                    # a recorded call failed mid-plan: the side streams forked inside the plan were never joined, so nothing
                    # may run unordered behind them — drain the device before the error propagates
                    torch.cuda.synchronize()
                    raise
                finally:
                    cur.wait_stream(self._cap_stream)   # ... and the caller's stream continues behind the plan
            else:
                g.replay()
            if net is not None:
                self._exchange(net)

    def close(self):
        """Release the native launch plans (and their hipEvents) of a recorded step."""
        graphs, self._graphs = self._graphs, None
        if graphs:
            L = ops.lib()
            for _, _, plan in graphs:
                if plan is not None:
                    L.plan_destroy(plan)
            self._ticket_scope = None       # the recorded launches' ticket buffer goes with them

    def __del__(self):
        import sys
        if sys.is_finalizing():       # the HIP runtime may already be gone: destroying events then aborts the process
            return
        try:
            self.close()
        except Exception:
            pass

    def step(self, image, target_od, target_oc, noise=None):
        """image [B,3,H,W] in [-1,1], targets [B,1,H,W] in {0,1}; all device fp32, rows domain-major.
        noise: optional {'a': eps, 'c': eps} standard-normal [B,1,H,W] (parity runs); default Philox.
        Returns {name: 0-dim device tensor}; nothing is synchronised with the host.  With graph=True the returned tensors
        are the graphs' own output buffers: read them before the next step() overwrites them."""
        image = image.contiguous()
        if self.graph and not noise:
            if self._graphs is None:
                self._capture(image, target_od, target_oc)
            for dst, src in zip(self._static, (image, target_od, target_oc)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            self._replay()
            return self._result
        gen = self._schedule(image, target_od, target_oc, noise or {})
        try:
            while True:
                self._exchange(next(gen))
        except StopIteration as stop:
            return stop.value
