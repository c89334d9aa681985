"""The ReLU / LeakyReLU decisions a forward pass of the product's plan took, read off its saved pre-activations -- TEST INFRASTRUCTURE (like the rest of oracle/: only tests/, __graft_entry__.smoke() and
bench.py's checker legs may import it).

`y.grad_fn.sv` of a model output is the engine's saved state (gast_hip/engine.py::Engine.forward): every pre-BatchNorm tensor in
position-major layout plus the scale / shift the consumers applied.  A kernel decides `fmaf(x, scale, shift) > 0`; the exact product
plus one rounding has the sign of the float64 evaluation below, so these ARE the kernels' decisions.  They are returned in the
oracle's call order and layouts (oracle/gast_oracle.py::OracleModel.forward) for `np_autograd.TIES['forced']`:

    expand_bn                                  (B, C0, T0, J)
    per level s >= 1:  layers_bn[2s-2], [2s-1] (B, C, T_s, J) each
    per block s:       bn_1, bn_2, local cat_bn (B, C, T_s, J); 4 heads' LeakyReLU(a_i + c_j) (B*T_s, J, J); global cat_bn (B, C, T_s, J);
                       block cat_bn (B, 2C, T_s, J)
"""
import torch

NHEADS = 4


def plan_decisions(sv, J):
    B, T = sv['B'], sv['T']
    out = []

    def relu_mask(X, st, lo, hi, Tn):
        z = X[:, lo:hi].double() * st.scale[lo:hi].double() + st.shift[lo:hi].double()
        return (z > 0).view(B, Tn, J, hi - lo).permute(0, 3, 1, 2).contiguous()

    C0 = sv['E'].shape[1]
    out.append(relu_mask(sv['E'], sv['bnE'], 0, C0, T[0]))
    for s, st in enumerate(sv['stages']):
        C, Tn = st['C'], st['Tn']
        if s > 0:
            lv = sv['levels'][s - 1]
            out.append(relu_mask(lv['T1'], lv['bn1'], 0, C, Tn))
            out.append(relu_mask(lv['T2'], lv['bn2'], 0, C, Tn))
        out.append(relu_mask(st['Y'], st['bnY'], 0, C, Tn))
        out.append(relu_mask(st['Y'], st['bnY'], C, 2 * C, Tn))
        out.append(relu_mask(st['LG'], st['bnLG'], 0, C, Tn))
        H = st['H']
        F = B * Tn
        for h in range(NHEADS):
            a = H[:, 5 * C + h].float().view(F, J)
            c = H[:, 5 * C + NHEADS + h].float().view(F, J)
            out.append((a[:, :, None] + c[:, None, :]) > 0)
        out.append(relu_mask(st['LG'], st['bnLG'], C, 2 * C, Tn))
        out.append(relu_mask(st['O'], st['bnO'], 0, 2 * C, Tn))
    return out
