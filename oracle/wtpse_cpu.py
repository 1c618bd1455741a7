"""CPU restatement of the WT-PSE training hot path (TEST INFRASTRUCTURE — see oracle/__init__.py).

Functional style: a network is a plain ``{state_dict key: tensor}`` mapping (``sd``),
every layer is a function of ``(sd, key prefix, inputs)``, all arithmetic is stock
PyTorch fp32 ops on the host.  Device-agnostic (no ``.cuda()``), sampling noise is
an explicit argument.  Pinned against the imported reference by
``tests/test_oracle_golden.py`` + ``tests/golden/*.npz``.

Citations are file:line into the reference (tonyckc/WT-PSE-code @ 2024-12-23).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# hparams the hot path reads (hparams_registry.py:71-93 defaults)
DEFAULT_HPARAMS = {
    "whitening": True,
    "shape_prior": True,
    "shape_attention": True,
    "shape_attention_coeffient": 0.3,
    "cat_shape": False,
    "margin": 0,
    "shape_start": 0.5,
    "instance_wt_gm": 1,
    "domain_wt_gm": 1,
    "multi-turn": 1,
}

BN_EPS = 1e-5        # nn.BatchNorm2d default (algorithms.py:862-864)
BN_MOMENTUM = 0.1
WT_EPS = 1e-5        # algorithms.py:1140
WT_DIM = 16          # algorithms.py:1157
THRESH = 0.75        # algorithms.py:1244, Trainer.py:842


# ----------------------------------------------------------------------------- state helpers
def is_buffer(name: str) -> bool:
    leaf = name.rsplit(".", 1)[-1]
    return leaf in ("running_mean", "running_var", "num_batches_tracked")


def param_names(sd):
    return [k for k in sd if not is_buffer(k)]


def as_leaves(sd):
    """Clone `sd` to CPU fp32 and mark parameters as autograd leaves."""
    out = {}
    for k, v in sd.items():
        t = v.detach().to("cpu").clone()
        if not is_buffer(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# ----------------------------------------------------------------------------- layers
def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def _bn(sd, name, x, training):
    # nn.BatchNorm2d(planes): eps 1e-5, momentum 0.1, affine, track_running_stats
    if training:
        sd[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training, BN_MOMENTUM, BN_EPS)


def conv_d(sd, pre, x, first, training):
    """ConvD.forward — algorithms.py:897-917 (dup shape_networks.py:347-367).
    [maxpool2] -> conv+bn (NO activation) -> conv+bn+relu -> conv+bn+relu."""
    if not first:
        x = F.max_pool2d(x, 2)
    x = _bn(sd, pre + "bn1", _conv(sd, pre + "conv1", x, 1), training)
    y = F.relu(_bn(sd, pre + "bn2", _conv(sd, pre + "conv2", x, 1), training))
    z = F.relu(_bn(sd, pre + "bn3", _conv(sd, pre + "conv3", y, 1), training))
    return z


def conv_u(sd, pre, x, prev, first, training):
    """ConvU.forward — algorithms.py:941-962 (dup shape_networks.py:391-412).
    [conv3x3+bn+relu] -> bilinear x2 (align_corners=False) -> conv1x1+bn+relu -> cat(prev,y) -> conv3x3+bn+relu."""
    if not first:
        x = F.relu(_bn(sd, pre + "bn1", _conv(sd, pre + "conv1", x, 1), training))
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.relu(_bn(sd, pre + "bn2", _conv(sd, pre + "conv2", y, 0), training))
    y = torch.cat([prev, y], 1)
    y = F.relu(_bn(sd, pre + "bn3", _conv(sd, pre + "conv3", y, 1), training))
    return y


def unet_body(sd, pre, x1, training):
    """down1-4 / up1-4 — algorithms.py:1025-1033, 1219-1226; shape_networks.py:473-481."""
    x2 = conv_d(sd, pre + "down1.", x1, False, training)
    x3 = conv_d(sd, pre + "down2.", x2, False, training)
    x4 = conv_d(sd, pre + "down3.", x3, False, training)
    x5 = conv_d(sd, pre + "down4.", x4, False, training)
    x = conv_u(sd, pre + "up1.", x5, x4, True, training)
    x = conv_u(sd, pre + "up2.", x, x3, False, training)
    x = conv_u(sd, pre + "up3.", x, x2, False, training)
    x = conv_u(sd, pre + "up4.", x, x1, False, training)
    return x


def deep_wt(sd, pre, x):
    """DeepWT.forward / DoubleConvWT — algorithms.py:1091-1117, 416-428.
    Returns [z1, z2, relu(z2)]; no normalisation is applied (IN modules are constructed, never called)."""
    a = pre + "DoubleConv.double_conv."
    b = pre + "DoubleConv2.double_conv."
    z1 = _conv(sd, a + "2", F.relu(_conv(sd, a + "0", x, 1)), 1)
    z2 = _conv(sd, b + "2", F.relu(_conv(sd, b + "0", F.relu(z1), 1)), 1)
    return [z1, z2, F.relu(z2)]


def head3(sd, pre, x):
    """mu_prior / logvar_prior — algorithms.py:1006-1012; shape_networks.py:459-465. 1x1 conv chain 32->32->8->n."""
    x = F.relu(_conv(sd, pre + "0", x, 0))
    x = F.relu(_conv(sd, pre + "2", x, 0))
    return _conv(sd, pre + "4", x, 0)


def attention(sd, pre, x):
    """attention_layer.forward — algorithms.py:1126-1129: (sigmoid(conv1x1(x)), conv1x1(x))."""
    x1 = _conv(sd, pre + "layer1", x, 0)
    return torch.sigmoid(x1), x1


# ----------------------------------------------------------------------------- WT loss
def mmd_pair(x, y):
    """compute_MMD.mmd / gaussian_kernel / my_cdist — algorithms.py:65-88. gamma = [1]."""
    def k(a, b):
        an = a.pow(2).sum(-1, keepdim=True)
        bn = b.pow(2).sum(-1, keepdim=True)
        d = torch.addmm(bn.transpose(-2, -1), a, b.transpose(-2, -1), alpha=-2).add(an)
        return torch.exp(-d.clamp_min(1e-30))
    return k(x, x).mean() + k(y, y).mean() - 2 * k(x, y).mean()


def mmd(v, domain_num, batch_size):
    """compute_MMD.forward — algorithms.py:102-121: contiguous row blocks, mean over unordered pairs."""
    feats = [v[batch_size * i: batch_size * (i + 1)] for i in range(domain_num)]
    penalty = 0
    for i in range(domain_num):
        for j in range(i + 1, domain_num):
            penalty = penalty + mmd_pair(feats[i], feats[j])
    if domain_num > 1:
        penalty = penalty / (domain_num * (domain_num - 1) / 2)
    return penalty


def gram(z, eps=WT_EPS):
    """algorithms.py:1278-1283: uncentred z zT / (HW-1) + eps*I."""
    B, C, H, W = z.shape
    f = z.contiguous().view(B, C, -1)
    return torch.bmm(f, f.transpose(1, 2)).div(H * W - 1) + eps * torch.eye(C, dtype=z.dtype)


def whitening_loss(z, domain_num, batch_size, margin=0.0, eps=WT_EPS):
    """compute_whitening_loss — algorithms.py:1277-1309; shape_networks.py:561-594.
    Returns (ins_offdiag, ins_diag, domain).  WT_PSE returns (ins_offdiag + ins_diag, domain)."""
    B, C, H, W = z.shape
    g = gram(z, eps)
    triu = torch.ones(C, C, dtype=z.dtype).triu(diagonal=1)
    eye = torch.eye(C, dtype=z.dtype)
    g_off = g * triu
    g_diag = g * eye
    off = g_off.abs().sum(dim=(1, 2)) - margin
    ins_off = torch.clamp(off / triu.sum(), min=0).sum() / B
    dg = (g_diag - eye).abs().sum(dim=(1, 2)) - margin
    ins_diag = torch.clamp(dg / C, min=0).sum() / B
    iu = torch.triu_indices(C, C, 1)
    v = g_off[:, iu[0], iu[1]]
    dom = mmd(v, domain_num, batch_size)
    return ins_off, ins_diag, dom


# ----------------------------------------------------------------------------- shape nets
def _scrub_nan(t):
    """shape_networks.py:490-492,504-506: only when a NaN is present, nan_to_num the whole tensor
    (the follow-up `t[t == inf] = 0` can never fire after nan_to_num)."""
    if torch.isnan(t).any():
        t = torch.nan_to_num(t)
    return t


def teacher_forward(sd, pre, feat, mask, training, eps_noise=None):
    """ShapeVariationalDist_y_x.unet_extractor/.sample_forward/.reparameterization — algorithms.py:1014-1033,1055-1075.
    training: returns (mu + exp(logvar/2)*eps, mu); eval: mu."""
    a = pre + "inc.double_conv."
    m = F.relu(_bn(sd, a + "1", _conv(sd, a + "0", mask, 1), training))
    m = F.relu(_bn(sd, a + "4", _conv(sd, a + "3", m, 1), training))
    x1 = F.relu(_conv(sd, pre + "fusion.0", torch.cat([m, feat], 1), 0))
    fmap = unet_body(sd, pre, x1, training)
    mu = head3(sd, pre + "mu_prior.", fmap)
    logvar = head3(sd, pre + "logvar_prior.", fmap)
    if not training:
        return mu
    std = torch.exp(logvar / 2)
    if eps_noise is None:
        eps_noise = torch.randn_like(std)
    return mu + std * eps_noise, mu


def student_forward(sd, feat, training, eps_noise=None):
    """ShapeVariationalDist_x.unet_extractor/.sample_forward/.reparameterization — shape_networks.py:468-510.
    Sampling quirk kept: s = normal(mu, std) (no grad through s); z = s*std + mu."""
    fmap = unet_body(sd, "", feat, training)
    mu = head3(sd, "mu_prior.", fmap)
    logvar = head3(sd, "logvar_prior.", fmap)
    mu = _scrub_nan(mu)
    if not training:
        return mu
    std = _scrub_nan(torch.exp(logvar / 2))
    if eps_noise is None:
        eps_noise = torch.randn_like(std)
    s = (mu + std * eps_noise).detach()     # torch.normal(mu, std) is not differentiable
    return s * std + mu, mu


# ----------------------------------------------------------------------------- WT_PSE
def main_unet(sd, x, training):
    """algorithms.py:1218-1227: inc, down1-4, up1-4, mu head (1x1 32->32, ReLU, 1x1 32->8)."""
    x1 = conv_d(sd, "inc.", x, True, training)
    f = unet_body(sd, "", x1, training)
    return _conv(sd, "mu.2", F.relu(_conv(sd, "mu.0", f, 0)), 0)


def wt_pse_update(sd, hp, inputs, mask, two_stage_inputs=None, two_step=False, noise=None,
                  domain_num=3, per_domain_batch=1):
    """WT_PSE.update — algorithms.py:1216-1275 (train mode)."""
    emb = main_unet(sd, inputs, True)
    if not hp["shape_prior"]:
        return _conv(sd, "outc.0", emb, 0), 0, 0, 0, 0
    w = deep_wt(sd, "wt_model.", two_stage_inputs if two_step else inputs)
    z_post, _z_mu = teacher_forward(sd, "prior_dist.", w[-1], mask, True, noise)
    att, _ = attention(sd, "attention_layer.", z_post)
    att_mask = (att > THRESH).float()
    fuse = hp["shape_attention_coeffient"] * emb + att * emb
    if hp["cat_shape"]:                 # algorithms.py:1192,1253: outc takes feature_dim + 1 channels
        fuse = torch.cat([fuse, z_post], 1)
    # quirk (algorithms.py:1259-1267): two terms summed, divided by len(list) == 3
    ins = 0
    dom = 0
    for e in range(len(w) - 1):
        off, dg, d = whitening_loss(w[e], domain_num, per_domain_batch, hp["margin"])
        ins = ins + (off + dg)
        dom = dom + d
    ins = ins / len(w)
    dom = dom / len(w)
    out = _conv(sd, "outc.0", fuse, 0)
    return out, att_mask, att_mask, ins, dom


def wt_pse_predict(sd, sd_shape, hp, inputs_all, two_step):
    """WT_PSE.predict — algorithms.py:1311-1353 (eval mode; uses the STUDENT's wt_model and shape net)."""
    if two_step:
        inputs, two_stage_inputs = inputs_all[0], inputs_all[1]
    else:
        inputs = two_stage_inputs = inputs_all
    emb = main_unet(sd, inputs, False)
    if not hp["shape_prior"]:
        return _conv(sd, "outc.0", emb, 0), None
    w = deep_wt(sd_shape, "wt_model.", two_stage_inputs)
    z = student_forward(sd_shape, w[-1], False)
    att, pre_sig = attention(sd, "attention_layer.", z)
    fuse = hp["shape_attention_coeffient"] * emb + att * emb
    if hp["cat_shape"]:                 # algorithms.py:1348
        fuse = torch.cat([fuse, z], 1)
    return _conv(sd, "outc.0", fuse, 0), pre_sig


def shape_update(sd_shape, sd_main, hp, inputs, mask, two_stage_inputs=None, two_step=False,
                 noise_teacher=None, noise_student=None, per_domain_batch=1):
    """ShapeVariationalDist_x.update — shape_networks.py:512-558 (train mode).
    Returns (kd, ins_total, ins_off, ins_diag, dom) with the accumulator quirk of :546-548."""
    x = two_stage_inputs if two_step else inputs
    w1 = deep_wt(sd_main, "wt_model.", x)
    w2 = deep_wt(sd_shape, "wt_model.", x)
    _z_post, mu_t = teacher_forward(sd_main, "prior_dist.", w1[-1], mask, True, noise_teacher)
    _z_pre, mu_s = student_forward(sd_shape, w2[-1], True, noise_student)
    kd = F.mse_loss(mu_t, mu_s)
    # attention calls at :533-535 are dead compute
    n = len(w2)
    ins_off = 0
    dom = 0
    ins_diag = 0
    for e in range(n - 1):
        off, dg, d = whitening_loss(w2[e], 3, per_domain_batch, hp["margin"])   # student MMD: 3 domains, :448
        ins_off = ins_off + off
        ins_diag = dg + dg            # `a, ins2, c = f(); ins2 += ins2` overwrites the accumulator each pass
        dom = dom + d
    ins_off = ins_off / n
    ins_diag = ins_diag / n
    dom = dom / n
    return kd, ins_off + ins_diag, ins_off, ins_diag, dom


# ----------------------------------------------------------------------------- caller glue (a-11)
def seg_loss_od(output, target):
    """Trainer.py:19,787: BCELoss(sigmoid(output), target), mean."""
    return F.binary_cross_entropy(torch.sigmoid(output), target)


def roi_from_od(image, output):
    """Trainer.py:842-853: od_pred = sigmoid(out) > 0.75 ; roi = (image+1)*od_pred - 1."""
    od_pred = (torch.sigmoid(output) > THRESH).float().detach()
    return (image + 1) * od_pred - 1, od_pred


def seg_loss_oc(output_oc, od_pred, target_oc):
    """Trainer.py:865-871: pos_weight = sum(od_pred)/sum(od_pred*target_oc) (1 if inf/nan); BCE-with-logits of out*od_pred."""
    pw = od_pred.sum() / (od_pred * target_oc).sum()
    if torch.isinf(pw) or torch.isnan(pw):
        pw = torch.tensor(1.0)
    return F.binary_cross_entropy_with_logits(output_oc * od_pred, target_oc, pos_weight=pw)


def _f(v):
    return v.item() if torch.is_tensor(v) else float(v)


class Nets:
    """The four networks + four Adam optimisers of train.py:91-138 as plain state mappings."""

    def __init__(self, sd_od, sd_shape_od, sd_oc, sd_shape_oc, lr=5e-4):
        self.od, self.shape_od, self.oc, self.shape_oc = (as_leaves(s) for s in (sd_od, sd_shape_od, sd_oc, sd_shape_oc))
        mk = lambda sd: torch.optim.Adam([sd[k] for k in param_names(sd)], lr=lr, betas=(0.9, 0.99))
        self.opt_od, self.opt_shape_od, self.opt_oc, self.opt_shape_oc = (
            mk(self.od), mk(self.shape_od), mk(self.oc), mk(self.shape_oc))

    @staticmethod
    def zero(sd):
        for k in param_names(sd):
            sd[k].grad = None


def train_iteration(nets, hp, image, target_od, target_oc, noise, per_domain_batch, domain_num=3,
                    skip_dead_teacher_backward=False):
    """One body of the hot loop, Trainer.py:766-914 (A: seg OD, B: shape OD, ROI, C: seg OC, D: shape OC).
    `noise` = dict with keys a, b_t, b_s, c, d_t, d_s -> [B,1,H,W] standard-normal fixtures (or None).
    Returns the scalar losses of the iteration (python floats)."""
    noise = noise or {}
    image = image.clone()
    res = {}
    # ---- A
    Nets.zero(nets.od)
    out, _, _, ins, dom = wt_pse_update(nets.od, hp, image, target_od, image, True, noise.get("a"),
                                        domain_num, per_domain_batch)
    l_seg = seg_loss_od(out, target_od)
    loss = l_seg + hp["instance_wt_gm"] * ins + hp["domain_wt_gm"] * dom
    loss.backward()
    nets.opt_od.step()
    res.update(seg_od=l_seg.item(), ins_od=_f(ins), dom_od=_f(dom), main_od=loss.item())
    # ---- B
    if hp["whitening"]:
        for _ in range(hp["multi-turn"]):
            Nets.zero(nets.shape_od)
            kd, ins_t, ins_ij, ins_ii, dom_s = shape_update(nets.shape_od, nets.od, hp, image, target_od, image, True,
                                                            noise.get("b_t"), noise.get("b_s"), per_domain_batch)
            loss_s = kd + hp["instance_wt_gm"] * ins_t + hp["domain_wt_gm"] * dom_s
            loss_s.backward()
            nets.opt_shape_od.step()
        res.update(kd_od=kd.item(), ins_shape_od=ins_t.item(), ins_ij_od=ins_ij.item(), ins_ii_od=ins_ii.item(),
                   dom_shape_od=dom_s.item(), shape_od=loss_s.item())
    # ---- ROI
    roi, od_pred = roi_from_od(image, out)
    # ---- C
    Nets.zero(nets.oc)
    out_oc, _, _, ins_c, dom_c = wt_pse_update(nets.oc, hp, roi, target_oc, roi, True, noise.get("c"),
                                               domain_num, per_domain_batch)
    l_seg_oc = seg_loss_oc(out_oc, od_pred, target_oc)
    loss_oc = l_seg_oc + hp["instance_wt_gm"] * ins_c + hp["domain_wt_gm"] * dom_c
    loss_oc.backward()
    nets.opt_oc.step()
    res.update(seg_oc=l_seg_oc.item(), ins_oc=_f(ins_c), dom_oc=_f(dom_c), main_oc=loss_oc.item())
    # ---- D
    if hp["whitening"]:
        for _ in range(hp["multi-turn"]):
            Nets.zero(nets.shape_oc)
            kd2, ins_t2, ins_ij2, ins_ii2, dom_s2 = shape_update(nets.shape_oc, nets.oc, hp, roi, target_oc, roi, True,
                                                                 noise.get("d_t"), noise.get("d_s"), per_domain_batch)
            loss_s2 = kd2 + hp["instance_wt_gm"] * ins_t2 + hp["domain_wt_gm"] * dom_s2
            loss_s2.backward()
            nets.opt_shape_oc.step()
        res.update(kd_oc=kd2.item(), ins_shape_oc=ins_t2.item(), dom_shape_oc=dom_s2.item(), shape_oc=loss_s2.item())
    return res


# ----------------------------------------------------------------------------- metrics
def dice_coefficient(seg, gt):
    """metrics.py:68-97: (2*|A&B| + 1) / (|A| + |B| + 1) on boolean masks."""
    seg = np.asarray(seg, dtype=np.bool_)
    gt = np.asarray(gt, dtype=np.bool_)
    inter = float(np.logical_and(seg, gt).sum())
    return (2 * inter + 1.0) / (1.0 + float(seg.sum()) + float(gt.sum()))


def checksum(t):
    """Per-tensor fingerprint stored in fixtures instead of multi-MB tensors: (sum, sum|.|, 32 strided samples)."""
    f = t.detach().double().reshape(-1)
    n = f.numel()
    idx = torch.linspace(0, n - 1, steps=min(32, n)).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()]).astype(np.float64)


# ----------------------------------------------------------------------------- validation front half (Trainer.py:170-209)
def validate_predict(sd_od, sd_shape_od, sd_oc, sd_shape_oc, hp, data, label_size):
    """predictions (OD) and predictions_oc * od_pred (OC), both resized to the label size — Trainer.py:170-209."""
    pred, _ = wt_pse_predict(sd_od, sd_shape_od, hp, data, False)
    od_pred = (torch.sigmoid(pred) > THRESH).float()
    roi = (data + 1) * od_pred - 1
    pred_oc, _ = wt_pse_predict(sd_oc, sd_shape_oc, hp, torch.stack((roi, roi), 0), True)
    pred_oc = pred_oc * od_pred
    pred = F.interpolate(pred, size=tuple(label_size), mode="bilinear")
    pred_oc = F.interpolate(pred_oc, size=tuple(label_size), mode="bilinear")
    return pred, pred_oc
