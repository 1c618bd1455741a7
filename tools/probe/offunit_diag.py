"""Diagnostic (GPU box): where does the off-unit-statistics parity case stand against the reference's own fp32 noise?
HIP (x2h / x3) and the CPU oracle in fp32, all against the oracle in fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
import torch
import test_parity_gpu as T
from oracle import wtpse_cpu as O
from oracle.inputs import make_inputs, make_noise
from wtpse_hip import ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, pb = 6, 2
img, od, oc = make_inputs(177, B, H, H)
eps = make_noise(178, (B, 1, H, H))
for variant in ("plain", "offunit"):
    main, shape, _, _ = T.build_nets(pb)
    if variant == "offunit":
        T._off_unit_statistics(main, 5)
        T._off_unit_statistics(shape, 6)
    sd = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    to64 = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}
    with torch.no_grad():
        r32 = O.wt_pse_update(dict(sd), T.HP, img, od, img, True, eps, 3, pb)[0]
        r64 = O.wt_pse_update(to64(sd), T.HP, img.double(), od.double(), img.double(), True, eps.double(), 3, pb)[0]
    res = {}
    for terms in (2, 3):
        ops.lib().query("wtpse_x3_terms", terms)
        main.load_state_dict(sd)
        main.train(); main.set_noise([eps])
        with torch.no_grad():
            out = main.update(img.to("cuda"), od.to("cuda"), two_stage_inputs=img.to("cuda"), two_step=True)[0]
        res[terms] = out.cpu().double()
    ops.lib().query("wtpse_x3_terms", 2)
    sc = float(r64.abs().max())
    print("%s H=%d scale %.3g | max |d| vs fp64: oracle32 %.3e  x2h %.3e  x3 %.3e | x2h vs oracle32 %.3e" % (
        variant, H, sc, float((r32.double() - r64).abs().max()), float((res[2] - r64).abs().max()), float((res[3] - r64).abs().max()),
        float((res[2] - r32.double()).abs().max())))
