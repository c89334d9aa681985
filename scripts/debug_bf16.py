"""Debug helper (GPU box): where does the bf16 path drift from the fp32 path?  Runs the BASELINE-size model in both modes
(train-mode BN) and prints, per saved tensor, the error of the NORMALISED value z = x*scale+shift (what the next layer sees).
Usage: python scripts/debug_bf16.py [warm_steps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from tests_helpers import PARENTS
from model.gast_net import SpatioTemporalModel
from bench import adj_from_parents      # (the product's callers build adj themselves; nothing under oracle/ is used here)

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
adj = adj_from_parents(PARENTS[17])
torch.manual_seed(0)
m = SpatioTemporalModel(adj, 17, 2, 17, filter_widths=[3, 3, 3], channels=128, dropout=0.0).cuda()
gen = torch.Generator().manual_seed(1234)
x = (torch.rand(128, 27, 17, 2, generator=gen) * 2 - 1).cuda()
os.environ['GAST_HIP_DTYPE'] = 'fp32'
m.train()
with torch.no_grad():
    for _ in range(warm):
        m(x)
eng = m._runner.engine
orig = eng.forward
cap = {}
def wrapped(*a, **k):
    pred, sv = orig(*a, **k)
    cap['sv'] = sv
    return pred, sv
eng.forward = wrapped
res = {}
for mode in ('fp32', 'bf16'):
    os.environ['GAST_HIP_DTYPE'] = mode
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        y = m(x)
    res[mode] = (y.clone(), cap['sv'])
    m.load_state_dict(sd)
print('centered:', eng.centered, ' out diff %.3e' % (res['fp32'][0] - res['bf16'][0]).abs().max().item())
f = lambda t: t.detach().double()
def z(t, bn, sl=None):
    sc, sh = f(bn.scale), f(bn.shift)
    return f(t) * sc[None, :] + sh[None, :]
def report(tag, ta, bna, tb, bnb):
    za, zb = z(ta, bna), z(tb, bnb)
    xa = f(ta)
    ratio = (xa.mean(0).abs() / (xa.std(0) + 1e-12)).max().item()
    xb = f(tb)
    ratio_b = (xb.mean(0).abs() / (xb.std(0) + 1e-12)).max().item()
    print('%-14s z max diff %.3e  rms diff %.3e  (z rms %.2f)  max|mean|/std stored: fp32 %.1f bf16 %.1f  scale max %.1f' % (
        tag, (za - zb).abs().max().item(), (za - zb).pow(2).mean().sqrt().item(), za.pow(2).mean().sqrt().item(), ratio, ratio_b,
        f(bna.scale).abs().max().item()))
sa, sb = res['fp32'][1], res['bf16'][1]
report('E', sa['E'], sa['bnE'], sb['E'], sb['bnE'])
for s, (a, b) in enumerate(zip(sa['stages'], sb['stages'])):
    if s > 0:
        la, lb = sa['levels'][s - 1], sb['levels'][s - 1]
        report('L%d.T1' % s, la['T1'], la['bn1'], lb['T1'], lb['bn1'])
        report('L%d.T2' % s, la['T2'], la['bn2'], lb['T2'], lb['bn2'])
    print('S%d.X          diff %.3e (rms %.2f)' % (s, (f(a['X']) - f(b['X'])).abs().max().item(), f(a['X']).pow(2).mean().sqrt().item()))
    print('S%d.H          diff %.3e (rms %.2f)' % (s, (f(a['H']) - f(b['H'])).abs().max().item(), f(a['H']).pow(2).mean().sqrt().item()))
    report('S%d.Y' % s, a['Y'], a['bnY'], b['Y'], b['bnY'])
    print('S%d.Ya         diff %.3e (rms %.2f)' % (s, (f(a['Ya']) - f(b['Ya'])).abs().max().item(), f(a['Ya']).pow(2).mean().sqrt().item()))
    report('S%d.LG' % s, a['LG'], a['bnLG'], b['LG'], b['bnLG'])      # [local | global] pre-BN tensor, one BN state for both halves
    report('S%d.O' % s, a['O'], a['bnO'], b['O'], b['bnO'])
