"""Flat Adam / AMSGrad for the GAST-Net training step.

Drop-in for the reference's `optim.Adam(model_pos.parameters(), lr=lr, amsgrad=True)` (reference trainval.py:78, stepped at
main.py:238): same constructor arguments, same update rule (torch/optim/adam.py `_single_tensor_adam`), same
`zero_grad()/step()/state_dict()` surface.  Parameters are re-homed into ONE flat fp32 buffer (each `p.data` becomes a view
of it), gradients into another (shared with `gast_hip.dist.FlatGradAllReduce` when one is attached), moments are flat, so the
whole update is one HIP launch over a contiguous stream (csrc/optim_ops.hip) instead of torch's 8 multi-tensor launches.

There is no CPU implementation: `step()` on CPU parameters raises.
"""
import torch

from gast_hip.packer import note_raw_parameter_write


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, ops=None, skip_nonfinite=None):
        """skip_nonfinite: skip the whole update (parameters, moments, step counter) when the flat gradient holds an inf / NaN, and count
        it in `skipped_steps` -- what torch.cuda.amp.GradScaler does around an optimizer.  Default: on in the loss-scaled 16-bit mode
        (GAST_HIP_DTYPE=f16, where an overflow of the scaled activation gradients shows up exactly like that), off otherwise.  The check
        and the counter stay on the device (hipGraph-capturable); the loss scale itself is static (GAST_F16_LOSS_SCALE)."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameters')
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        self._ops = ops
        if skip_nonfinite is None:
            import os
            skip_nonfinite = os.environ.get('GAST_HIP_DTYPE', '').lower() == 'f16'
        self.skip_nonfinite = bool(skip_nonfinite)
        self.skipped_steps = None      # device int64 counter (created with the first guarded step)
        self.grad_scale = 1.0      # gradients are multiplied by this inside the update kernel (1/world after a SUM all-reduce)
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            if any(p.dtype != torch.float32 for p in ps):
                raise TypeError('FlatAdam: fp32 parameters only')
            dev = ps[0].device
            n = sum(p.numel() for p in ps)     # tightly packed in parameter order: the layout of FlatGradAllReduce / the packer
            P = torch.zeros(n, dtype=torch.float32, device=dev)
            offs, off = [], 0
            with torch.no_grad():
                for p in ps:
                    P[off:off + p.numel()].copy_(p.detach().reshape(-1))
                    p.data = P[off:off + p.numel()].view(p.shape)
                    offs.append(off)
                    off += p.numel()
            st = dict(params=ps, offs=offs, n=n, P=P, G=None,
                      m=torch.zeros(n, dtype=torch.float32, device=dev), v=torch.zeros(n, dtype=torch.float32, device=dev),
                      vmax=torch.zeros(n, dtype=torch.float32, device=dev) if group['amsgrad'] else None,
                      step=torch.zeros(1, dtype=torch.int32, device=dev))
            self._flat.append(st)

    # ------------------------------------------------------------------ gradients
    def _grad_buffer(self, st):
        """Flat gradient buffer whose slices ARE the p.grad tensors.  Adopts an existing layout (FlatGradAllReduce: tightly
        packed) when every p.grad already is a view of one buffer at our offsets; otherwise installs our own views."""
        ps, offs = st['params'], st['offs']
        G = st['G']
        if G is not None and all(p.grad is not None and p.grad.data_ptr() == G.data_ptr() + 4 * o for p, o in zip(ps, offs)):
            return G
        g0 = ps[0].grad
        if g0 is not None:
            base = g0.data_ptr() - 4 * offs[0]
            root = g0._base if g0._base is not None else g0
            if (all(p.grad is not None and p.grad.is_contiguous() and p.grad.data_ptr() == base + 4 * o for p, o in zip(ps, offs))
                    and root.dtype == torch.float32 and root.data_ptr() <= base
                    and base + 4 * st['n'] <= root.data_ptr() + 4 * root.numel()):
                start = (base - root.data_ptr()) // 4
                st['G'] = root.view(-1)[start:start + st['n']]
                return st['G']
        G = torch.zeros(st['n'], dtype=torch.float32, device=st['P'].device)
        for p, o in zip(ps, offs):
            view = G[o:o + p.numel()].view(p.shape)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
        st['G'] = G
        return G

    def flat_grads(self):
        """The flat gradient buffers (one per parameter group) -- e.g. to all-reduce them in one collective."""
        return [self._grad_buffer(st) for st in self._flat if st is not None]

    def zero_grad(self, set_to_none=False):
        """One fill per group.  (`set_to_none=True` would detach the gradients from the flat buffer again; it is ignored.)"""
        for st in self._flat:
            if st is not None:
                self._grad_buffer(st).zero_()

    # ------------------------------------------------------------------ step
    def _get_ops(self):
        if self._ops is None:
            from gast_hip.binding import HipOps
            self._ops = HipOps()
        return self._ops

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, st in zip(self.param_groups, self._flat):
            if st is None:
                continue
            if not st['P'].is_cuda:
                raise RuntimeError('FlatAdam: parameters are on %s; the HIP path needs device tensors (no CPU fallback)' % st['P'].device)
            for p, o in zip(st['params'], st['offs']):
                if p.data_ptr() != st['P'].data_ptr() + 4 * o:
                    raise RuntimeError('FlatAdam: a parameter was moved out of the flat buffer (model.to()/load after construction?); '
                                       'build the optimizer after the model is on its device')
            G = self._grad_buffer(st)
            b1, b2 = group['betas']
            skip = None
            ops = self._get_ops()
            if self.skip_nonfinite:
                if self.skipped_steps is None:
                    self.skipped_steps = torch.zeros(1, dtype=torch.int64, device=G.device)
                    self._flags = torch.zeros(ops.NONFINITE_FLAGS, dtype=torch.int32, device=G.device)
                skip = self._flags
                ops.nonfinite_scan(G, skip)          # one launch over the flat gradient: a verdict per block, read by the update kernel
            ops.adam_step(st['P'], G, st['m'], st['v'], st['vmax'] if group['amsgrad'] else None, st['step'],
                          float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']),
                          grad_scale=float(self.grad_scale), skip=skip, skipped=self.skipped_steps if skip is not None else None)
            note_raw_parameter_write()      # (the kernel writes parameter memory behind torch's version counters)
        return loss

    # ------------------------------------------------------------------ (de)serialisation in torch.optim.Adam's per-parameter format
    def state_dict(self):
        state, idx = {}, 0
        groups = []
        for group, st in zip(self.param_groups, self._flat):
            ids = []
            if st is not None:
                step = st['step'].to(torch.float32).reshape(()).clone()
                for p, o in zip(st['params'], st['offs']):
                    n = p.numel()
                    ent = {'step': step.clone(), 'exp_avg': st['m'][o:o + n].view(p.shape).clone(),
                           'exp_avg_sq': st['v'][o:o + n].view(p.shape).clone()}
                    if group['amsgrad']:
                        ent['max_exp_avg_sq'] = st['vmax'][o:o + n].view(p.shape).clone()
                    state[idx] = ent
                    ids.append(idx)
                    idx += 1
            g = {k: v for k, v in group.items() if k != 'params'}
            g['params'] = ids
            groups.append(g)
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        for group, st, g in zip(self.param_groups, self._flat, sd['param_groups']):
            for k, v in g.items():
                if k != 'params':
                    group[k] = v
            if st is None:
                continue
            for p, o, i in zip(st['params'], st['offs'], g['params']):
                ent = sd['state'].get(i)
                if ent is None:
                    continue
                n = p.numel()
                st['m'][o:o + n].copy_(ent['exp_avg'].reshape(-1))
                st['v'][o:o + n].copy_(ent['exp_avg_sq'].reshape(-1))
                if group['amsgrad'] and 'max_exp_avg_sq' in ent:
                    if st['vmax'] is None:
                        st['vmax'] = torch.zeros_like(st['v'])
                    st['vmax'][o:o + n].copy_(ent['max_exp_avg_sq'].reshape(-1))
                st['step'].fill_(int(ent['step']))
