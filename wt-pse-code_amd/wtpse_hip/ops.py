"""Thin host wrappers over the C ABI (include/wtpse_hip.h): allocate outputs/scratch with torch (device
memory + stream plumbing only) and launch on torch's current HIP stream.  No arithmetic happens here."""
import torch

from .lib import lib

_WS = {}


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """Raw hipStream_t of torch's current stream (the C getter: torch.cuda.current_stream() builds a Python Stream object
    on every call, ~8 us, and this runs once per launch)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, name="tensor"):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous fp32 tensor in device memory (got %s, %s, contiguous=%s)"
                         % (name, t.device, t.dtype, t.is_contiguous()))


def workspace(tag, nfloats, device):
    """Persistent scratch, one buffer per (device, stream, tag), grown on demand: independent parts of the schedule run on
    different streams (nn.py) and must not share scratch."""
    key = (str(device), stream_ptr(), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1024), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# ----------------------------------------------------------------------------------------------- convolution
def conv_fwd(in0, in1, wpacked_ptr, bias, cout, ksize, pro0=None, pro_relu=0, relu_out=False, want_stats=False,
             split=None, mask_ref=None, pro1=None, out_amax=None):
    """-> (out0, out1 or None, stats or None).  `wpacked_ptr` is a raw device pointer into the packed-weight buffer.
    out_amax: a zeroed amax table (fwd_amax_table) that receives the largest magnitude of the stored output."""
    _chk(in0, "in0"); _chk(in1, "in1"); _chk(pro0, "pro0"); _chk(pro1, "pro1")
    B, C0, H, W = in0.shape
    C1 = 0 if in1 is None else in1.shape[1]
    L = lib()
    if split is None:
        out0 = torch.empty((B, cout, H, W), dtype=torch.float32, device=in0.device)
        out1 = None
        csplit = cout
    else:
        csplit = int(split)
        out0 = torch.empty((B, csplit, H, W), dtype=torch.float32, device=in0.device)
        out1 = torch.empty((B, cout - csplit, H, W), dtype=torch.float32, device=in0.device)
    stats = None
    if want_stats:
        nblk = L.query("wtpse_conv_stats_blocks", B, H, W)
        stats = torch.empty((nblk, cout, 2), dtype=torch.float32, device=in0.device)
    L.call("wtpse_conv_fwd", ptr(in0), C0, ptr(in1), C1, wpacked_ptr, ptr(bias), ptr(pro0), ptr(pro1), int(pro_relu), ptr(out0),
           ptr(out1), csplit, ptr(stats), B, H, W, cout, ksize, int(relu_out), ptr(mask_ref), ptr(out_amax), stream_ptr())
    if out_amax is not None:
        out0.wt_amax = out_amax
    return out0, out1, stats


def conv_fwd_gram(in0, wpacked_ptr, bias, pro0=None, pro_relu=0, relu_out=False, out_amax=None):
    """3x3 conv with 16 output channels that also returns the per-tile partial Grams of its output.
    -> (out, (partial [B*S,256], S)) with S = tiles per image: what wt_loss_fwd(..., gram_partial=) takes."""
    _chk(in0, "in0"); _chk(pro0, "pro0")
    B, C0, H, W = in0.shape
    L = lib()
    out = torch.empty((B, 16, H, W), dtype=torch.float32, device=in0.device)
    nblk = L.query("wtpse_conv_stats_blocks", B, H, W)
    partial = torch.empty((nblk, 256), dtype=torch.float32, device=in0.device)
    L.call("wtpse_conv_fwd_gram", ptr(in0), C0, wpacked_ptr, ptr(bias), ptr(pro0), int(pro_relu), ptr(out), ptr(partial), B, H, W,
           16, int(relu_out), ptr(out_amax), stream_ptr())
    if out_amax is not None:
        out.wt_amax = out_amax
    return out, (partial, nblk // B)


X16_SIZE = 8 + 5 * 3 * 64 * 8 + 5 * 2 * 64 * 8      # unsigned shorts of one direction of one conv in the 16-channel layout: header, x3 and x2h fragments


def conv16_x3(in0, wx16_ptr, bias, cout, pro0=None, pro_relu=0, relu_out=False, want_stats=False, want_gram=False, mask_ref=None,
              bnb=None, grad_in=False, in_amax=None, out_amax=None):
    """3x3 conv with at most 16 input and 16 output channels in the x3 / x2h arithmetic (csrc/conv.hip MODE 3 / 4; include/wtpse_hip.h,
    wtpse_conv16_x3).  bnb = (bn_y, bn_ss, bn_mean, bn_relu): the BatchNorm-backward epilogue of dgrad_bnb.  grad_in: in0 is a
    gradient (a data-gradient launch): x2h only with the amax table its producer attached (`in0.wt_amax`), x3 otherwise — never an
    extra pass for it.  in_amax (forward launches): the bound of the input as loaded; out_amax: as conv_fwd.
    -> (out, stats or None, (gram partial, tiles per image) or None)."""
    _chk(in0, "in0"); _chk(pro0, "pro0"); _chk(mask_ref, "mask_ref")
    B, C0, H, W = in0.shape
    L = lib()
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=in0.device)
    nblk = L.query("wtpse_conv_stats_blocks", B, H, W)
    stats = torch.empty((nblk, cout, 2), dtype=torch.float32, device=in0.device) if (want_stats or bnb is not None) else None
    gram = torch.empty((nblk, 256), dtype=torch.float32, device=in0.device) if want_gram else None
    bn_ss = bn_mean = None
    bn_relu = 0
    if bnb is not None:
        mask_ref, bn_ss, bn_mean, bn_relu = bnb
        _chk(mask_ref, "bn_y"); _chk(bn_ss, "bn_ss"); _chk(bn_mean, "bn_mean")
        assert mask_ref.shape == out.shape
    L.call("wtpse_conv16_x3", ptr(in0), C0, wx16_ptr, ptr(bias), ptr(pro0), int(pro_relu), ptr(out), ptr(stats), ptr(gram),
           ptr(mask_ref), ptr(bn_ss), ptr(bn_mean), int(bool(bn_relu)), B, H, W, cout, int(relu_out), int(bool(grad_in)),
           ptr(getattr(in0, "wt_amax", None)) if grad_in else ptr(in_amax), ptr(out_amax), stream_ptr())
    if out_amax is not None:
        out.wt_amax = out_amax
    return out, stats, ((gram, nblk // B) if want_gram else None)


def x3_packed_size(rows, k, taps):
    """unsigned shorts of one direction of one conv in the x3 layout (include/wtpse_hip.h): 64 bytes of header + the term slots."""
    return 32 + ((k + 15) & ~15) * ((rows + 31) & ~31) * taps * 3


def x3_terms():
    """16-bit terms per fp32 operand in the x3 kernels right now (wtpse_x3_terms: 3 x3, 2 x2h, 1 bf16 mode)."""
    return lib().query("wtpse_x3_terms", -1)


AMAX_WORDS = 1024           # unsigneds of one amax table (include/wtpse_hip.h, wtpse_amax): 64 shards, one per 64-byte line
_AMAX = {}
_AMAX_TABLES = 512          # tables per arena (2 MB): a backward pass of the largest network hands out ~120


def amax_begin(device):
    """Start of a backward pass: a fresh arena of zeroed amax tables for the gradients it produces.  ONE launch zeroes the WHOLE arena,
    unconditionally — the call is part of recorded launch plans / captured graphs, so what it zeroes must not depend on how many
    tables the pass before it happened to use when the step was recorded (a conditional, sized zeroing left the replayed steps with
    the tables of earlier steps: scales that only ever grew, results that differed from the eager step in the last bits).
    Round 6: the arena is a buffer of the PASS, not of the device — rounds 5's one arena per device was zeroed wholesale
    by whichever network began a backward pass next, under the feet of a pass of another network still in flight.  The producers of a
    pass (bn_bwd_*, upsample2x_bwd*) take their tables with _amax_table(); the tables (views, attached to the gradients as
    `wt_amax`) keep their arena alive, and the last few arenas are held a little longer for work still queued on side streams."""
    st = _AMAX.get(device)
    buf = zero_(torch.empty(_AMAX_TABLES * AMAX_WORDS, dtype=torch.int32, device=device))     # (2 MB, ~2 us; recorded steps: from the capture's pool)
    if st is None:
        st = _AMAX[device] = [buf, 0, []]
    else:
        st[2].append(st[0])
        del st[2][:-3]
        st[0], st[1] = buf, 0


def _amax_table(device):
    """A zeroed amax table for a producer, or None when the x2h arithmetic is off (nothing would read it)."""
    if x3_terms() != 2:
        return None
    st = _AMAX.get(device)
    if st is None or st[1] >= _AMAX_TABLES:        # no arena yet / exhausted (micro-benchmarks looping without begin_backward)
        return zero_(torch.empty(AMAX_WORDS, dtype=torch.int32, device=device))
    t = st[0][st[1] * AMAX_WORDS:(st[1] + 1) * AMAX_WORDS]
    st[1] += 1
    return t


_FWD_SCOPES = []
FWD_SCOPE_TABLES = 256      # per forward pass (1 MB): a WT_PSE.update hands out ~60, an eval-mode predict ~170


class fwd_scope:
    """The amax tables of ONE forward pass (an update() / predict() call): forward activations carry a bound of their largest
    magnitude to their x2h consumers in tables like the gradients' (include/wtpse_hip.h, wtpse_x3_terms), and those tables must be zero
    when their producer runs.  A scope owns one buffer, zeroed by ONE launch when the pass starts (on the stream the pass starts on,
    in front of every fork), and hands out slices; the slices (held by the activations on the tape) keep the buffer alive until the
    backward pass is done with them.  Recorded steps: the buffer comes from the capture's pool and the zeroing is a recorded launch.
    Without a scope (block-level callers, tests) a table is a fresh zeroed tensor of its own."""

    def __init__(self, device):
        self.buf = None
        if x3_terms() == 2:
            self.buf = zero_(torch.empty(FWD_SCOPE_TABLES * AMAX_WORDS, dtype=torch.int32, device=device))
        self.next = 0

    def __enter__(self):
        _FWD_SCOPES.append(self)
        return self

    def __exit__(self, *exc):
        assert _FWD_SCOPES and _FWD_SCOPES[-1] is self
        _FWD_SCOPES.pop()
        return False


def fwd_amax_table(device):
    """A zeroed amax table for the producer of a forward activation, or None when the x2h arithmetic is off (nobody would read it)."""
    if x3_terms() != 2:
        return None
    sc = _FWD_SCOPES[-1] if _FWD_SCOPES else None
    if sc is None or sc.buf is None or sc.buf.device != device or sc.next >= FWD_SCOPE_TABLES:
        return zero_(torch.empty(AMAX_WORDS, dtype=torch.int32, device=device))
    t = sc.buf[sc.next * AMAX_WORDS:(sc.next + 1) * AMAX_WORDS]
    sc.next += 1
    return t


def act_bound(ss, raw_amax):
    """Amax table of |ss[c,0] * y + ss[c,1]| over a tensor y whose amax table is raw_amax (include/wtpse_hip.h, wtpse_act_bound)."""
    tab = torch.empty(AMAX_WORDS, dtype=torch.int32, device=ss.device)
    lib().call("wtpse_act_bound", ptr(ss), ss.shape[0], ptr(raw_amax), ptr(tab), stream_ptr())
    return tab


def amax_of(t):
    """The amax table (int32 tensor [AMAX_WORDS]) of the tensor t as stored: the scale source of an operand of the x2h kernels.
    Producers that fill it as they write t (bn_bwd_*, upsample2x_bwd*; the convolutions' epilogues for un-normalised forward maps) attach
    it to their result as `t.wt_amax`; anything else pays one extra pass here (and keeps the table on the tensor for its other consumers)."""
    tab = getattr(t, "wt_amax", None)
    if tab is None:
        tab = torch.empty(AMAX_WORDS, dtype=torch.int32, device=t.device)
        lib().call("wtpse_amax", ptr(t), t.numel(), ptr(tab), stream_ptr())
        t.wt_amax = tab
    return tab


def conv_fwd_x3(in0, in1, wpacked_ptr, bias, cout, ksize, pro0=None, pro_relu=0, relu_out=False, want_stats=False,
                split=None, mask_ref=None, pro1=None, in_amax=None, in_amax1=None, out_amax=None):
    """conv_fwd on the 16-bit matrix cores at fp32 accuracy (csrc/conv_x3.hip); `wpacked_ptr` points into the x3-packed weights.
    in_amax / in_amax1: the amax tables of in0 / in1 AS LOADED (a gradient's amax_of; a forward activation's bound, nn.act_amax) —
    include/wtpse_hip.h, wtpse_x3_terms; out_amax: as conv_fwd."""
    _chk(in0, "in0"); _chk(in1, "in1"); _chk(pro0, "pro0"); _chk(pro1, "pro1")
    B, C0, H, W = in0.shape
    C1 = 0 if in1 is None else in1.shape[1]
    L = lib()
    if split is None:
        out0 = torch.empty((B, cout, H, W), dtype=torch.float32, device=in0.device)
        out1 = None
        csplit = cout
    else:
        csplit = int(split)
        out0 = torch.empty((B, csplit, H, W), dtype=torch.float32, device=in0.device)
        out1 = torch.empty((B, cout - csplit, H, W), dtype=torch.float32, device=in0.device)
    stats = None
    if want_stats:
        nblk = L.query("wtpse_conv_x3_stats_blocks", B, H, W, cout, int(ksize))
        stats = torch.empty((nblk, cout, 2), dtype=torch.float32, device=in0.device)
    L.call("wtpse_conv_fwd_x3", ptr(in0), C0, ptr(in1), C1, wpacked_ptr, ptr(bias), ptr(pro0), ptr(pro1), int(pro_relu), ptr(out0),
           ptr(out1), csplit, ptr(stats), B, H, W, cout, ksize, int(relu_out), ptr(mask_ref), ptr(in_amax), ptr(in_amax1),
           ptr(out_amax), stream_ptr())
    if out_amax is not None:
        out0.wt_amax = out_amax
    return out0, out1, stats


_TICKETS = {}
_TICKET_RING = 1 << 22      # unsigneds (16 MB): ~1900 launches of the largest layer (5 training steps) before a slice comes round again; the
                            # streams of a step are joined at its end, so two users of a slice never run at the same time
_TICKET_SCOPE = None        # [tensor, next]: a private ticket buffer while a step is being recorded (ticket_scope)
_DEBUG = __import__("os").environ.get("WTPSE_DEBUG", "0") == "1"


class ticket_scope:
    """While a step is recorded into a launch plan / captured into a graph, its launches take their ticket slices from a buffer of
    their OWN (allocated here, before the capture starts; owned by the recorded step and freed with it) instead of the eager ring:
    a replayed launch keeps its slice for as long as the recording lives, and the eager ring's pointer, which comes round every
    ~1900 launches, must never land on it (an eager step between two replays would have shared slices with the
    replayed launches that were in flight on other streams)."""

    def __init__(self, device, n=1 << 21):
        self.buf = torch.zeros(n, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()

    def __enter__(self):
        global _TICKET_SCOPE
        assert _TICKET_SCOPE is None, "ticket scopes do not nest"
        _TICKET_SCOPE = [self.buf, 0]
        return self

    def __exit__(self, *exc):
        global _TICKET_SCOPE
        _TICKET_SCOPE = None
        return False


def _tickets(n, device):
    """n zeroed unsigneds for a launch that leaves them zeroed (include/wtpse_hip.h: wtpse_dgrad_bnb_coef).  Eager launches: slices
    of one ring per device — a slice is handed out again only after ~1900 further launches, long after its launch has finished.
    Launches recorded inside a ticket_scope: consecutive slices of the scope's private buffer, never reused.  -> (pointer, view)."""
    n = (n + 63) & ~63
    if _TICKET_SCOPE is not None:
        buf, off = _TICKET_SCOPE
        if off + n > buf.numel():
            raise RuntimeError("ticket_scope exhausted (%d unsigneds): a recorded step uses more ticket words than reserved" % buf.numel())
        _TICKET_SCOPE[1] = off + n
        return buf.data_ptr() + 4 * off, buf[off:off + n]
    st = _TICKETS.get(device)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            # the zero fill would be captured instead of executed (and the ring would live in the capture's private pool)
            raise RuntimeError("the ticket ring must exist before a step is captured: HipNet.ensure_ready() creates it")
        st = _TICKETS[device] = [torch.zeros(_TICKET_RING, dtype=torch.int32, device=device), 0]
        torch.cuda.current_stream(device).synchronize()
    if st[1] + n > _TICKET_RING:
        st[1] = 0
    off = st[1]
    st[1] += n
    view = st[0][off:off + n]
    if _DEBUG and n and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize(device)
        assert not bool(view.any()), "ticket slice [%d, %d) is not zero on entry: a launch that used it failed or is still running" % (off, off + n)
    return st[0].data_ptr() + 4 * off, view


def _ticket_call(view, name, *args):
    """lib().call for a launch that owns the ticket slice `view`: a launch that fails (rejected arguments after the slice was
    taken, a launch error) may leave its slice half-counted — it is re-zeroed (after a device sync) before the error propagates,
    so that the launch that gets the slice next does not fold its statistics one arrival early."""
    try:
        lib().call(name, *args)
    except Exception:
        if view.numel() and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize(view.device)
            zero_(view)
        raise


def dgrad_bnb(dy, wpacked_ptr, layout, cout, ksize, bn_y, bn_ss, bn_mean, bn_relu, split=None, bn_second=False, tail=None,
              in_amax=None):
    """Data gradient (`wpacked_ptr`: the layer's data-gradient weights; layout 0 fp32, 1 x3, 2 the 16-channel x3 fragments) whose
    epilogue masks the result with the ReLU of the conv + BatchNorm layer it flows into and forms that layer's BatchNorm-backward
    reductions (include/wtpse_hip.h, wtpse_dgrad_bnb).  With a split the BatchNorm'd tensor is out0, or out1 if `bn_second`.
    tail = (gamma, invstd, dgamma, dbeta) of that BatchNorm: the launch also folds the partials and leaves the coefficients
    (wtpse_dgrad_bnb_coef).
    -> (out0, out1 or None, stats [nblk, Cbn, 2], coef [Cbn, 3] or None)."""
    _chk(dy, "dy"); _chk(bn_y, "bn_y"); _chk(bn_ss, "bn_ss"); _chk(bn_mean, "bn_mean")
    layout = int(layout)
    B, C, H, W = dy.shape
    L = lib()
    if split is None:
        csplit, c0, c1 = cout, 0, cout
        out0 = torch.empty((B, cout, H, W), dtype=torch.float32, device=dy.device)
        out1 = None
    else:
        csplit = int(split)
        c0, c1 = (csplit, cout) if bn_second else (0, csplit)
        out0 = torch.empty((B, csplit, H, W), dtype=torch.float32, device=dy.device)
        out1 = torch.empty((B, cout - csplit, H, W), dtype=torch.float32, device=dy.device)
    assert bn_y.shape == (B, c1 - c0, H, W), (bn_y.shape, (B, c1 - c0, H, W))
    nblk = L.query("wtpse_conv_x3_stats_blocks", B, H, W, cout, int(ksize)) if layout == 1 else L.query("wtpse_conv_stats_blocks", B, H, W)
    stats = torch.empty((nblk, c1 - c0, 2), dtype=torch.float32, device=dy.device)
    if tail is not None:
        gamma, invstd, dgamma, dbeta = tail
        coef = torch.empty((c1 - c0, 3), dtype=torch.float32, device=dy.device)
        partial2 = torch.empty(L.query("wtpse_bnb_tail_partial2", nblk, cout), dtype=torch.float64, device=dy.device)
        tickets, tview = _tickets(L.query("wtpse_bnb_tail_tickets", nblk, cout), dy.device)
        _ticket_call(tview, "wtpse_dgrad_bnb_coef", ptr(dy), C, wpacked_ptr, layout, ptr(out0), ptr(out1), csplit, ptr(bn_y), ptr(bn_ss),
               ptr(bn_mean), int(bool(bn_relu)), c0, c1, ptr(stats), ptr(gamma), ptr(invstd), ptr(coef), ptr(dgamma), ptr(dbeta), 0,
               ptr(partial2), tickets, B, H, W, cout, ksize, ptr(in_amax), stream_ptr())
        return out0, out1, stats, coef
    if layout == 2:
        L.call("wtpse_conv16_x3", ptr(dy), C, wpacked_ptr, 0, 0, 0, ptr(out0), ptr(stats), 0, ptr(bn_y), ptr(bn_ss), ptr(bn_mean),
               int(bool(bn_relu)), B, H, W, cout, 0, 1, ptr(in_amax), 0, stream_ptr())
    elif layout == 1:
        L.call("wtpse_dgrad_x3_bnb", ptr(dy), C, wpacked_ptr, ptr(out0), ptr(out1), csplit, ptr(bn_y), ptr(bn_ss), ptr(bn_mean),
               int(bool(bn_relu)), c0, c1, ptr(stats), B, H, W, cout, ksize, ptr(in_amax), stream_ptr())
    else:
        L.call("wtpse_dgrad_bnb", ptr(dy), C, wpacked_ptr, ptr(out0), ptr(out1), csplit,
               ptr(bn_y), ptr(bn_ss), ptr(bn_mean), int(bool(bn_relu)), c0, c1, ptr(stats), B, H, W, cout, ksize, stream_ptr())
    return out0, out1, stats, None


def conv_wgrad(dy, x0, x1, ksize, dw, dbias, pro0=None, pro_relu=0, accumulate=False, pro1=None):
    """dw / dbias are views into the flat gradient buffer ([Cout,Cin,k,k] / [Cout] or None)."""
    _chk(dy, "dy"); _chk(x0, "x0"); _chk(x1, "x1")
    B, cout, H, W = dy.shape
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    cin = C0 + C1
    L = lib()
    ks = L.query("wtpse_wgrad_ksplit", B, H, W, cin, cout)
    slab = workspace("wgrad_slab", ks * cout * cin * ksize * ksize, dy.device)
    dbs = workspace("wgrad_dbias", ks * cout, dy.device) if dbias is not None else None
    L.call("wtpse_conv_wgrad", ptr(dy), ptr(x0), C0, ptr(x1), C1, ptr(pro0), ptr(pro1), int(pro_relu), ptr(slab), ptr(dbs), ks,
           ptr(dw), ptr(dbias), int(accumulate), B, H, W, cout, ksize, stream_ptr())


def wgrad_x3_supported(cin, cout, ksize, c0):
    return bool(lib().query("wtpse_wgrad_x3_supported", int(cin), int(cout), int(ksize), int(c0)))


def conv_wgrad_x3(dy, x0, x1, ksize, dw, pro0=None, pro_relu=0, accumulate=False, pro1=None):
    """conv_wgrad in the x3 arithmetic (csrc/conv_x3.hip); no bias gradient."""
    _chk(dy, "dy"); _chk(x0, "x0"); _chk(x1, "x1")
    B, cout, H, W = dy.shape
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    cin = C0 + C1
    L = lib()
    ks = L.query("wtpse_wgrad_x3_ksplit", B, H, W, cin, cout)
    slab = workspace("wgrad_slab", ks * cout * cin * ksize * ksize, dy.device)
    L.call("wtpse_conv_wgrad_x3", ptr(dy), ptr(x0), C0, ptr(x1), C1, ptr(pro0), ptr(pro1), int(pro_relu), ptr(slab), ks, ptr(dw),
           int(accumulate), B, H, W, cout, ksize, stream_ptr())


def wgrad_r_supported(cin, cout, ksize, c0, w):
    return bool(lib().query("wtpse_wgrad_r_supported", int(cin), int(cout), int(ksize), int(c0), int(w)))


def conv_wgrad_r(dy, x0, x1, dw, dbias=None, pro0=None, pro_relu=0, accumulate=False, pro1=None, dy_amax=None, x_amax0=None,
                 x_amax1=None):
    """3x3 conv_wgrad in the x3 arithmetic with register-resident operands (csrc/wgrad_r.hip); with bias gradient.
    dy_amax: amax_of(dy) (x2h: the scale of the gradient operand); x_amax0 / x_amax1: the bounds of x0 / x1 as loaded."""
    _chk(dy, "dy"); _chk(x0, "x0"); _chk(x1, "x1")
    B, cout, H, W = dy.shape
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    cin = C0 + C1
    L = lib()
    ns = L.query("wtpse_wgrad_r_slabs", B, H, W, cin, cout)
    slab = workspace("wgrad_slab", ns * cout * cin * 9, dy.device)
    dbs = workspace("wgrad_dbias", ns * cout, dy.device) if dbias is not None else None
    L.call("wtpse_conv_wgrad_r", ptr(dy), ptr(x0), C0, ptr(x1), C1, ptr(pro0), ptr(pro1), int(pro_relu), ptr(slab), ptr(dbs), ns,
           ptr(dw), ptr(dbias), int(accumulate), B, H, W, cout, ptr(dy_amax), ptr(x_amax0), ptr(x_amax1), stream_ptr())


def conv_wgrad_r_bn(g, bn_y, coef, x0, x1, dw, pro0=None, pro_relu=0, accumulate=False, pro1=None):
    """conv_wgrad_r with dY = coef[:,0] * g + coef[:,1] * bn_y + coef[:,2] formed on load (no bias gradient)."""
    _chk(g, "g"); _chk(bn_y, "bn_y"); _chk(x0, "x0"); _chk(x1, "x1")
    B, cout, H, W = g.shape
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    cin = C0 + C1
    L = lib()
    ns = L.query("wtpse_wgrad_r_slabs", B, H, W, cin, cout)
    slab = workspace("wgrad_slab", ns * cout * cin * 9, g.device)
    L.call("wtpse_conv_wgrad_r_bn", ptr(g), ptr(bn_y), ptr(coef), ptr(x0), C0, ptr(x1), C1, ptr(pro0), ptr(pro1), int(pro_relu),
           ptr(slab), ns, ptr(dw), int(accumulate), B, H, W, cout, stream_ptr())


# ----------------------------------------------------------------------------------------------- batch norm
def conv_fwd_bnf(in0, in1, wpacked_ptr, layout, bias, cout, ksize, pro0, pro_relu, pro1, gamma, beta, rmean, rvar, nbt,
                 momentum=0.1, eps=1e-5, in_amax0=None, in_amax1=None, act_amax=None):
    """Forward convolution in front of a train-mode BatchNorm, statistics finished by the same launch (include/wtpse_hip.h,
    wtpse_conv_fwd_bnf).  layout 0 fp32, 1 x3, 2 the 16-channel x3 fragments.  in_amax0 / in_amax1: the bounds of the inputs as
    loaded; act_amax: a zeroed table that receives the bound of the BatchNorm's output.  -> (y, ss [C,2], mean, invstd)."""
    _chk(in0, "in0"); _chk(in1, "in1"); _chk(pro0, "pro0"); _chk(pro1, "pro1")
    layout = int(layout)
    B, C0, H, W = in0.shape
    C1 = 0 if in1 is None else in1.shape[1]
    L = lib()
    dev = in0.device
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=dev)
    nblk = L.query("wtpse_conv_x3_stats_blocks", B, H, W, cout, int(ksize)) if layout == 1 else L.query("wtpse_conv_stats_blocks", B, H, W)
    stats = torch.empty((nblk, cout, 2), dtype=torch.float32, device=dev)
    ss = torch.empty((cout, 2), dtype=torch.float32, device=dev)
    mean = torch.empty((cout,), dtype=torch.float32, device=dev)
    invstd = torch.empty((cout,), dtype=torch.float32, device=dev)
    partial2 = torch.empty(L.query("wtpse_bnb_tail_partial2", nblk, cout), dtype=torch.float64, device=dev)
    tickets, tview = _tickets(L.query("wtpse_bnb_tail_tickets", nblk, cout), dev)
    _ticket_call(tview, "wtpse_conv_fwd_bnf", ptr(in0), C0, ptr(in1), C1, wpacked_ptr, layout, ptr(bias), ptr(pro0), ptr(pro1), int(pro_relu),
           ptr(out), ptr(stats), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ptr(nbt), float(momentum), float(eps), ptr(ss),
           ptr(mean), ptr(invstd), ptr(partial2), tickets, B, H, W, cout, ksize, ptr(in_amax0), ptr(in_amax1), ptr(act_amax),
           stream_ptr())
    return out, ss, mean, invstd


def bn_finalize(stats, count, gamma, beta, rmean, rvar, nbt, momentum=0.1, eps=1e-5, act_amax=None):
    nblk, C, _ = stats.shape
    dev = stats.device
    ss = torch.empty((C, 2), dtype=torch.float32, device=dev)
    mean = torch.empty((C,), dtype=torch.float32, device=dev)
    invstd = torch.empty((C,), dtype=torch.float32, device=dev)
    lib().call("wtpse_bn_finalize", ptr(stats), nblk, C, int(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
               ptr(nbt), float(momentum), float(eps), ptr(ss), ptr(mean), ptr(invstd), ptr(act_amax), stream_ptr())
    return ss, mean, invstd


def bn_eval_coeffs(gamma, beta, rmean, rvar, eps=1e-5):
    C = gamma.numel()
    ss = torch.empty((C, 2), dtype=torch.float32, device=gamma.device)
    lib().call("wtpse_bn_eval_coeffs", ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), float(eps), C, ptr(ss), stream_ptr())
    return ss


def affine_act(y, ss, relu):
    _chk(y, "y")
    B, C, H, W = y.shape
    z = torch.empty_like(y)
    lib().call("wtpse_affine_act", ptr(y), ptr(ss), int(relu), ptr(z), B, C, H * W, stream_ptr())
    return z


def bn_bwd(dz, y, ss, relu, gamma, mean, invstd, dgamma, dbeta, accumulate=False):
    _chk(dz, "dz"); _chk(y, "y")
    B, C, H, W = y.shape
    L = lib()
    ns = L.query("wtpse_bn_bwd_nsplit", B, C, H * W)
    partial = workspace("bn_bwd_partial", ns * C * 2, y.device)
    coef = workspace("bn_bwd_coef", C * 3, y.device)
    dy = torch.empty_like(y)
    dy.wt_amax = _amax_table(y.device)
    L.call("wtpse_bn_bwd", ptr(dz), ptr(y), ptr(ss), int(relu), ptr(gamma), ptr(mean), ptr(invstd), ptr(partial),
           ptr(coef), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(dy), B, C, H * W, ptr(dy.wt_amax), stream_ptr())
    return dy


def bn_bwd_from_stats(g, y, stats, gamma, mean, invstd, dgamma, dbeta, accumulate=False):
    """Second half of a BatchNorm backward after dgrad_bnb: g = masked incoming gradient, stats = its partials.  -> dy."""
    _chk(g, "g"); _chk(y, "y"); _chk(stats, "stats")
    B, C, H, W = y.shape
    assert stats.shape[1] == C
    coef = workspace("bn_bwd_coef", C * 3, y.device)
    dy = torch.empty_like(y)
    dy.wt_amax = _amax_table(y.device)
    lib().call("wtpse_bn_bwd_from_stats", ptr(g), ptr(y), ptr(stats), stats.shape[0], ptr(gamma), ptr(mean), ptr(invstd), ptr(coef),
               ptr(dgamma), ptr(dbeta), int(accumulate), ptr(dy), B, C, H * W, ptr(dy.wt_amax), stream_ptr())
    return dy


# ----------------------------------------------------------------------------------------------- WT loss
class WtLossState:
    __slots__ = ("z", "gram", "offdiag", "diag", "dmmd_dv", "losses", "v", "B", "HW", "D", "n", "margin")


def bn_bwd_apply_coef(g, y, coef):
    """dy = k1 g + k2 y + k3 with the coefficients a dgrad_bnb(..., tail=) launch left."""
    _chk(g, "g"); _chk(y, "y")
    B, C, H, W = y.shape
    dy = torch.empty_like(y)
    dy.wt_amax = _amax_table(y.device)
    lib().call("wtpse_bn_bwd_apply_coef", ptr(g), ptr(y), ptr(coef), ptr(dy), B, C, H * W, ptr(dy.wt_amax), stream_ptr())
    return dy


def wt_loss_fwd(z, domain_num, per_domain, margin=0.0, eps=1e-5, losses_out=None, gram_partial=None):
    """-> WtLossState; st.losses = device [3] = (ins_offdiag, ins_diag, domain).
    gram_partial: (partial, S) from conv_fwd_gram — the producer of z already formed the per-tile Grams: z is not read."""
    _chk(z, "z")
    B, C, H, W = z.shape
    HW = H * W
    dev = z.device
    L = lib()
    S = L.query("wtpse_wt_split", B, HW, 0)
    R = domain_num * per_domain
    st = WtLossState()
    partial = workspace("wt_partial", B * S * 256, dev) if gram_partial is None else None
    st.gram = torch.empty((B, 256), dtype=torch.float32, device=dev)
    st.v = torch.empty((B, 120), dtype=torch.float32, device=dev)
    st.offdiag = torch.empty((B,), dtype=torch.float32, device=dev)
    st.diag = torch.empty((B,), dtype=torch.float32, device=dev)
    rowval = torch.empty((max(R, 1) + 1,), dtype=torch.float64, device=dev)     # + the tail launch's ticket word
    st.dmmd_dv = torch.empty((max(R, 1), 120), dtype=torch.float32, device=dev)
    st.losses = losses_out if losses_out is not None else torch.empty((3,), dtype=torch.float32, device=dev)
    if gram_partial is not None:
        gp, gs = gram_partial
        assert C == 16 and gp.shape[0] == B * gs
        L.call("wtpse_wt_loss_fwd_partials", ptr(gp), int(gs), B, HW, float(eps), float(margin), int(domain_num), int(per_domain),
               ptr(st.gram), ptr(st.v), ptr(st.offdiag), ptr(st.diag), ptr(rowval), ptr(st.dmmd_dv), ptr(st.losses), stream_ptr())
    else:
        L.call("wtpse_wt_loss_fwd", ptr(z), B, C, HW, float(eps), float(margin), int(domain_num), int(per_domain),
               ptr(partial), ptr(st.gram), ptr(st.v), ptr(st.offdiag), ptr(st.diag), ptr(rowval), ptr(st.dmmd_dv),
               ptr(st.losses), stream_ptr())
    st.z, st.B, st.HW, st.D, st.n, st.margin = z, B, HW, domain_num, per_domain, float(margin)
    return st


def wt_loss_bwd(st, dz, accumulate, g_off=None, g_diag=None, g_dom=None, w_off=1.0, w_diag=1.0, w_dom=1.0):
    _chk(dz, "dz")
    _amax_stale(dz)
    Mws = workspace("wt_M", st.B * 256, dz.device)
    lib().call("wtpse_wt_loss_bwd", ptr(st.z), st.B, 16, st.HW, st.margin, st.D, st.n, ptr(st.gram), ptr(st.offdiag),
               ptr(st.diag), ptr(st.dmmd_dv), ptr(g_off), ptr(g_diag), ptr(g_dom), float(w_off), float(w_diag),
               float(w_dom), ptr(Mws), ptr(dz), int(accumulate), stream_ptr())


def wt_combine(losses, den, mode, out=None):
    """losses [nmaps,3] -> out[4] = (ins_total, ins_off, ins_diag, dom); mode 1 = student accumulator quirk."""
    if out is None:
        out = torch.empty((4,), dtype=torch.float32, device=losses.device)
    lib().call("wtpse_wt_combine", ptr(losses), losses.shape[0], float(den), int(mode), ptr(out), stream_ptr())
    return out


# ----------------------------------------------------------------------------------------------- pooling etc.
def maxpool2_fwd(x, pro=None, relu=False):
    _chk(x, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    lib().call("wtpse_maxpool2_fwd", ptr(x), ptr(pro), int(relu), ptr(out), B, C, H, W, stream_ptr())
    return out


def maxpool2_bwd(x, dout, dx, accumulate, pro=None, relu=False, mask=False):
    """mask: the result is also multiplied with [act(x) > 0] (the ReLU that produced x)."""
    B, C, H, W = x.shape
    if dx is None:
        dx = torch.empty_like(x)
        accumulate = False
    _amax_stale(dx)
    lib().call("wtpse_maxpool2_bwd", ptr(x), ptr(pro), int(relu), ptr(dout), ptr(dx), int(bool(accumulate)) | (2 if mask else 0),
               B, C, H, W, stream_ptr())
    return dx


def maxpool2_bwd_bnb(x, dout, dx, pro, relu, mean):
    """maxpool2_bwd(mask=True) on the raw output of a conv + BatchNorm + ReLU layer that also forms the BatchNorm-backward
    reductions of that layer.  -> (masked gradient, stats [nblk, C, 2]) or None when the shape is not supported."""
    B, C, H, W = x.shape
    if W % 4 or H % 2 or B * C >= 32768 or pro is None:
        return None
    L = lib()
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    _amax_stale(dx)
    stats = torch.empty((L.query("wtpse_maxpool2_bwd_stats_blocks", B, H, W), C, 2), dtype=torch.float32, device=x.device)
    L.call("wtpse_maxpool2_bwd_bnb", ptr(x), ptr(pro), int(relu), ptr(dout), ptr(dx), int(acc), ptr(mean), ptr(stats), B, C, H, W,
           stream_ptr())
    return dx, stats


def upsample2x_fwd(x, pro=None, relu=False):
    _chk(x, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    lib().call("wtpse_upsample2x_fwd", ptr(x), ptr(pro), int(relu), ptr(out), B, C, H, W, stream_ptr())
    return out


def upsample2x_fwd_stats(x):
    """-> (out, stats [nblk, C, 2]): bilinear x2 plus the (sum, sum of squares) partials of the output."""
    _chk(x, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    nblk = lib().query("wtpse_upsample2x_stats_blocks", B, H, W)
    stats = torch.empty((nblk, C, 2), dtype=torch.float32, device=x.device)
    lib().call("wtpse_upsample2x_fwd_stats", ptr(x), ptr(out), ptr(stats), B, C, H, W, stream_ptr())
    return out, stats


def upsample2x_bwd(dout):
    _chk(dout, "dout")
    B, C, Ho, Wo = dout.shape
    dx = torch.empty((B, C, Ho // 2, Wo // 2), dtype=torch.float32, device=dout.device)
    dx.wt_amax = _amax_table(dout.device)
    lib().call("wtpse_upsample2x_bwd", ptr(dout), ptr(dx), 0, B, C, Ho // 2, Wo // 2, ptr(dx.wt_amax), stream_ptr())
    return dx


def upsample2x_bwd_bn(g, bn_y, coef):
    """upsample2x_bwd of dout = coef[:,0] * g + coef[:,1] * bn_y + coef[:,2] formed on load (no apply pass, dout never written)."""
    _chk(g, "g"); _chk(bn_y, "bn_y"); _chk(coef, "coef")
    B, C, Ho, Wo = g.shape
    assert bn_y.shape == g.shape and coef.shape == (C, 3) and Wo % 8 == 0, (g.shape, bn_y.shape, coef.shape)
    dx = torch.empty((B, C, Ho // 2, Wo // 2), dtype=torch.float32, device=g.device)
    dx.wt_amax = _amax_table(g.device)
    lib().call("wtpse_upsample2x_bwd_bn", ptr(g), ptr(bn_y), ptr(coef), ptr(dx), B, C, Ho // 2, Wo // 2, ptr(dx.wt_amax), stream_ptr())
    return dx


def resize_bilinear(x, size):
    _chk(x, "x")
    B, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    lib().call("wtpse_resize_bilinear", ptr(x), ptr(out), B, C, H, W, Ho, Wo, stream_ptr())
    return out


def relu_mask(dz, ref, out=None, accumulate=False):
    _chk(dz, "dz"); _chk(ref, "ref")
    if out is None:
        out = torch.empty_like(dz)
        accumulate = False
    _amax_stale(out)
    lib().call("wtpse_relu_mask", ptr(dz), ptr(ref), ptr(out), int(accumulate), dz.numel(), stream_ptr())
    return out


def _amax_stale(t):
    """t is about to be modified in place: an amax table its producer attached no longer describes it."""
    if t is not None and getattr(t, "wt_amax", None) is not None:
        t.wt_amax = None


def axpy(dst, src, alpha=1.0):
    _amax_stale(dst)
    lib().call("wtpse_axpy", ptr(dst), ptr(src), float(alpha), dst.numel(), stream_ptr())
    return dst


def zero_(t):
    lib().call("wtpse_zero", ptr(t), t.numel() * t.element_size(), stream_ptr())
    return t


# ----------------------------------------------------------------------------------------------- fused 1x1 heads
def head_fwd(x, pro, relu, w1, b1, w2, b2, w3, b3, want_tape=True, x_amax=None, want_h1=False):
    """32 -> 32 (ReLU) -> 8 [-> (ReLU) -> nc] in one kernel.  -> (out, h1 or None, h2): out is y [B,nc,H,W] for a
    three-layer head (w3 given), the 8-channel h2 for a two-layer head."""
    _chk(x, "x"); _chk(pro, "pro")
    B, C, H, W = x.shape
    three = w3 is not None
    nc = w3.shape[0] if three else 0
    # (x2h arithmetic: the backward forms layer 1 again from x — no h1 tape, 128 bytes per pixel less written here and read there)
    h1 = torch.empty((B, 32, H, W), dtype=torch.float32, device=x.device) if (want_tape and (want_h1 or x3_terms() != 2)) else None
    h2 = torch.empty((B, 8, H, W), dtype=torch.float32, device=x.device) if (want_tape or not three) else None
    y = torch.empty((B, nc, H, W), dtype=torch.float32, device=x.device) if three else None
    lib().call("wtpse_head_fwd", ptr(x), ptr(pro), int(bool(relu)), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), nc,
               ptr(h1), ptr(h2), ptr(y), ptr(x_amax), B, H * W, stream_ptr())
    return (y if three else h2), h1, h2


def head_bwd(dy, x, pro, relu, h1, h2, w1, w2, w3, dparams, accumulate=False, b1=None, x_amax=None, dy_amax=None):
    """-> dx (gradient wrt the activated input); the parameter gradients land in `dparams`, the contiguous flat-buffer
    range (dW1, db1, dW2, db2[, dW3, db3]).  x2h arithmetic (the default): h1 is not read — the kernel forms layer 1 again from
    x (b1 required) — and the operand scales come from the amax tables of the activated x (None: the fixed forward scale) and
    of dy (None: one extra pass over dy here); otherwise the fp32-input kernel, which reads the h1 tape."""
    _chk(dy, "dy"); _chk(x, "x"); _chk(h1, "h1"); _chk(h2, "h2")
    if x3_terms() == 2:
        if b1 is None:
            raise ValueError("head_bwd (x2h): b1 is required — layer 1 is formed again from x")
        if dy_amax is None:
            dy_amax = amax_of(dy)
    elif h1 is None:
        raise ValueError("head_bwd: the fp32-input kernel reads the layer-1 tape h1 (head_fwd(..., want_h1=True))")
    B, C, H, W = x.shape
    nc = w3.shape[0] if w3 is not None else 0
    L = lib()
    ns = 1024 + 32 + 256 + 8 + 9 * nc
    assert dparams.numel() == ns and dparams.is_contiguous()
    slab = workspace("head_slab", L.query("wtpse_head_slabs", B, H * W) * ns, x.device)
    dx = torch.empty_like(x)
    L.call("wtpse_head_bwd", ptr(dy), ptr(x), ptr(pro), int(bool(relu)), ptr(h1), ptr(h2), ptr(w1), ptr(b1), ptr(w2), ptr(w3), nc,
           ptr(dx), ptr(slab), ptr(dparams), int(accumulate), ptr(x_amax), ptr(dy_amax), B, H * W, stream_ptr())
    return dx


# ----------------------------------------------------------------------------------------------- attention / sampling
def attn_fuse_fwd(z, wb_ptr, emb, coef, want_att=True, want_pre=False, want_mask=False):
    _chk(z, "z"); _chk(emb, "emb")
    B, CE, H, W = emb.shape
    att = torch.empty_like(z) if want_att else None
    pre = torch.empty_like(z) if want_pre else None
    mask = torch.empty_like(z) if want_mask else None
    fuse = torch.empty_like(emb)
    lib().call("wtpse_attn_fuse_fwd", ptr(z), wb_ptr, ptr(emb), float(coef), ptr(att), ptr(pre), ptr(mask), ptr(fuse),
               B, CE, H * W, stream_ptr())
    return att, pre, mask, fuse


def attn_fuse_bwd(dfuse, z, emb, att, wb_ptr, coef, d_wb_ptr, want_dz=True, accumulate=False):
    B, CE, H, W = emb.shape
    demb = torch.empty_like(emb)
    dz = torch.empty_like(z) if want_dz else None
    nb = (B * H * W + 255) // 256
    partial = workspace("attn_partial", 2 * nb, emb.device)
    lib().call("wtpse_attn_fuse_bwd", ptr(dfuse), ptr(z), ptr(emb), ptr(att), wb_ptr, float(coef), ptr(demb), ptr(dz),
               ptr(partial), d_wb_ptr, int(accumulate), B, CE, H * W, stream_ptr())
    return demb, dz


def reparam_fwd(mu, logvar, eps):
    z = torch.empty_like(mu)
    lib().call("wtpse_reparam_fwd", ptr(mu), ptr(logvar), ptr(eps), ptr(z), mu.numel(), stream_ptr())
    return z


def reparam_bwd(dz, logvar, eps):
    dlv = torch.empty_like(dz)
    lib().call("wtpse_reparam_bwd", ptr(dz), ptr(logvar), ptr(eps), ptr(dlv), dz.numel(), stream_ptr())
    return dlv


def reparam_student(mu, logvar, eps, flag):
    std = torch.empty_like(mu)
    lib().call("wtpse_exp_half", ptr(logvar), ptr(std), mu.numel(), stream_ptr())
    nan_scrub_(std, flag)
    z = torch.empty_like(mu)
    lib().call("wtpse_reparam_student", ptr(mu), ptr(std), ptr(eps), ptr(z), mu.numel(), stream_ptr())
    return z


def nan_scrub_(x, flag):
    lib().call("wtpse_nan_scrub", ptr(x), x.numel(), flag.data_ptr(), stream_ptr())
    return x


def randn(shape, device, seed, offset=0, offset_dev=None):
    """Standard normals from the Philox stream `seed` at position offset + *offset_dev (device uint64 counter or None)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    lib().call("wtpse_randn", ptr(out), out.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset), ptr(offset_dev), stream_ptr())
    return out


def counter_add(counter, inc):
    """*counter += inc on the current stream (device int32 or int64 scalar tensor)."""
    lib().call("wtpse_counter_add", ptr(counter), int(inc), int(counter.dtype == torch.int64), stream_ptr())


# ----------------------------------------------------------------------------------------------- losses / optimiser
def _red_ws(n, device, mult=1):
    nb = lib().query("wtpse_reduce_blocks", n)
    return workspace("loss_partial", nb * mult, device)


def bce_sigmoid_fwd(x, t):
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    lib().call("wtpse_bce_sigmoid_fwd", ptr(x), ptr(t), x.numel(), ptr(_red_ws(x.numel(), x.device)), ptr(loss), stream_ptr())
    return loss


def bce_sigmoid_bwd(x, t, g=None, w=1.0):
    dx = torch.empty_like(x)
    lib().call("wtpse_bce_sigmoid_bwd", ptr(x), ptr(t), ptr(g), float(w), x.numel(), ptr(dx), stream_ptr())
    return dx


def pos_weight_sums(mask, t):
    """device [2] = (sum mask, sum mask*t) of THIS rank's rows (all-reduce before pos_weight_from_sums under DP)."""
    sums = torch.empty((2,), dtype=torch.float32, device=mask.device)
    pw = torch.empty((), dtype=torch.float32, device=mask.device)
    lib().call("wtpse_pos_weight", ptr(mask), ptr(t), mask.numel(), ptr(_red_ws(mask.numel(), mask.device, 2)), ptr(sums),
               ptr(pw), stream_ptr())
    return sums, pw


def pos_weight_from_sums(sums):
    pw = torch.empty((), dtype=torch.float32, device=sums.device)
    lib().call("wtpse_pos_weight_from_sums", ptr(sums), ptr(pw), stream_ptr())
    return pw


def bce_logits_pw_fwd(x, mask, t, pw):
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    lib().call("wtpse_bce_logits_pw_fwd", ptr(x), ptr(mask), ptr(t), ptr(pw), x.numel(), ptr(_red_ws(x.numel(), x.device)),
               ptr(loss), stream_ptr())
    return loss


def bce_logits_pw_bwd(x, mask, t, pw, g=None, w=1.0):
    dx = torch.empty_like(x)
    lib().call("wtpse_bce_logits_pw_bwd", ptr(x), ptr(mask), ptr(t), ptr(pw), ptr(g), float(w), x.numel(), ptr(dx), stream_ptr())
    return dx


def mse_fwd(a, b, out=None):
    loss = out if out is not None else torch.empty((), dtype=torch.float32, device=a.device)
    lib().call("wtpse_mse_fwd", ptr(a), ptr(b), a.numel(), ptr(_red_ws(a.numel(), a.device)), ptr(loss), stream_ptr())
    return loss


def mse_bwd(a, b, g=None, w=1.0):
    da = torch.empty_like(a)
    lib().call("wtpse_mse_bwd", ptr(a), ptr(b), ptr(g), float(w), a.numel(), ptr(da), stream_ptr())
    return da


def roi(image, logit):
    B, C, H, W = image.shape
    out = torch.empty_like(image)
    od = torch.empty_like(logit)
    lib().call("wtpse_roi", ptr(image), ptr(logit), ptr(out), ptr(od), B, C, H * W, stream_ptr())
    return out, od


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, step_dev=None):
    """Step number t = step + *step_dev (step_dev: device int32 count of completed steps, or None)."""
    lib().call("wtpse_adam", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
               int(step), ptr(step_dev), stream_ptr())
