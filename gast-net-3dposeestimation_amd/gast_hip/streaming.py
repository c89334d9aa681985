"""Causal streaming inference: one new frame per call, per-level frame buffers instead of re-running the window.

The reference's real-time demo (gen_skes.py:43-69 `load_model_realtime`, tools/inference.py:73-91 `gen_pose_frame`) feeds the last
`receptive_field` frames (edge-padded) through the causal single-frame model for EVERY new frame -- 27 or 81 frames of work per
pose.  A causal model only ever needs, per temporal level s, the last (k_s - 1) d_s + 1 frames of the previous block's output:

    level 0   raw inputs of the last k_0 frames  -> expand conv -> GraphAttentionBlock on ONE frame  -> O_0[t]
    level s   O_{s-1}[t], O_{s-1}[t - d], O_{s-1}[t - 2d] -> dilated conv + 1x1 + residual (the newest frame, gast_net.py:170)
              -> GraphAttentionBlock on one frame -> O_s[t]
    shrink    O_{L-1}[t] -> pose[t]

so a new frame costs one frame through every layer (1/27 resp. 1/81 of the window forward) and the buffers hold the pre-BatchNorm
block outputs exactly as the fused plan stores them.  The step launches the same kernels as the window forward (the taps are
row maps into the frame buffers); it is captured as a hipGraph and replayed per frame.  Weights are interchangeable between
`SpatioTemporalModel(causal=True)` and `SpatioTemporalModelOptimized1f(causal=True)` (reference gast_net.py:180-251): both are
accepted and give the same stream.  Flip test-time augmentation (reference main.py:313-318, tools/inference.py:80-84: mirrored
copy, un-mirror, average) runs as a second half of the batch inside the same step.
"""
import contextlib

import torch

from gast_hip.engine import BNState, Engine, PRO_BNRELU, RowMap, ident, inp_bn


class CausalStream:
    """stream = CausalStream(model, batch=1, flip=(kps_left, kps_right, joints_left, joints_right) or None)
       stream.reset(first_frame)      # left edge replication of the reference's generators (np.pad 'edge')
       pose = stream.push(frame)      # frame: (batch, J, 2) float32 on the model's device -> (batch, J, 3), the pose of THIS frame
    """

    def __init__(self, model, batch=1, flip=None, graph=True):
        from model.gast_net import ModelSpec, bn_buffers
        from gast_hip.packer import Packer
        runner = model._runner
        sp0 = runner.spec
        if not sp0.causal or sp0.dense:
            raise ValueError('CausalStream needs a causal model (constructor argument causal=True, dense=False): a symmetric model '
                             'needs future frames')
        self.model = model
        self.dev = next(model.parameters()).device
        if self.dev.type != 'cuda':
            raise RuntimeError('CausalStream (MI355X build): move the model to the GPU first; there is no CPU fallback')
        adj = model.layers_graph_conv[0].global_graph_layer.attentions[0].adj
        # the dilated twin of the model's plan (same parameters, taps `dilation` apart)
        self.spec = ModelSpec(adj, sp0.fw, sp0.channels, True, False, sp0.in_features)
        from gast_hip.binding import HipOps
        self.ops = HipOps()
        self.engine = Engine(self.spec, self.ops)
        self.packer = Packer(model, self.spec)
        self.bufs_fn = lambda: bn_buffers(model)
        self.runner = runner
        self.B_user = int(batch)
        self.flip = flip
        self.B = self.B_user * (2 if flip is not None else 1)
        self.J, self.F = self.spec.J, self.spec.in_features
        self.use_graph = bool(graph)
        if flip is not None:
            kl, kr, jl, jr = (list(v) for v in flip)
            perm_in = list(range(self.J))
            for a, b in zip(kl, kr):
                perm_in[a], perm_in[b] = b, a
            perm_out = list(range(self.J))
            for a, b in zip(jl, jr):
                perm_out[a], perm_out[b] = b, a
            self.perm_in = torch.tensor(perm_in, device=self.dev)
            self.perm_out = torch.tensor(perm_out, device=self.dev)
        self._graph = None
        self.refresh()

    # ------------------------------------------------------------------------------------------ parameters -> operands
    def refresh(self):
        """Re-pack the parameters and the eval-mode BatchNorm tables (call after load_state_dict / an optimizer step)."""
        sp, ops, dev = self.spec, self.ops, self.dev
        self.dt = self.runner.act_dtype
        ops.x3 = self.runner.x3
        if self.dt != torch.float32 and hasattr(ops, 'set_h16'):
            ops.set_h16(self.dt)          # (bfloat16 or binary16 flavour of the library for every launch of this stream)
        with torch.cuda.device(dev):
            self.st = self.packer.state(dev, self.dt, x3=self.runner.x3)
            ops.run_pack(self.packer, self.st)
            self.inp = self.packer.inputs(self.st)
            self.bufs = self.bufs_fn()
            self.engine._pre = self.engine._eval_table(self.inp, self.bufs, dev)
            L = len(sp.fw)
            C0 = sp.channels
            self.adjs, jobs = [], []
            for s in range(L):
                Cs = C0 * 2 ** s
                A_s = torch.empty(sp.nnz_sym + 1, Cs, dtype=torch.float32, device=dev)
                A_c = torch.empty(sp.nnz_con + 1, Cs, dtype=torch.float32, device=dev)
                jobs += [(self.inp['g%d.e_sym' % s], sp.pat_sym(dev), A_s), (self.inp['g%d.e_con' % s], sp.pat_con(dev), A_c)]
                self.adjs.append((A_s, A_c))
            ops.semch_adj_fwd_multi(jobs)
            # frame buffers: raw inputs of the last k0 frames; per level s >= 1 the last (k - 1) d + 1 block outputs of level s - 1
            self.Tb = [sp.fw[0]] + [(sp.fw[s] - 1) * sp.dil[s] + 1 for s in range(1, L)]
            self.xwin = torch.zeros(self.B, self.Tb[0], self.J, self.F, dtype=torch.float32, device=dev)
            self.obuf = [torch.zeros(self.B * self.Tb[s] * self.J, C0 * 2 ** s, dtype=self.dt, device=dev) for s in range(1, L)]
            self.x_in = torch.zeros(self.B_user, self.J, self.F, dtype=torch.float32, device=dev)
            self.pred = torch.zeros(self.B_user, self.J, 3, dtype=torch.float32, device=dev)
        self._graph = None
        self.frames = 0

    def reset(self, first_frame=None):
        """Forget the history.  With `first_frame` (batch, J, 2) the history is filled with copies of it -- the left edge
        replication `np.pad(..., 'edge')` of the reference's generators (common/generators.py:197-203) -- so that the next
        push() returns the pose of the clip's first frame."""
        self.xwin.zero_()
        for b in self.obuf:
            b.zero_()
        self.frames = 0
        if first_frame is not None:
            for _ in range(self.spec.receptive_field - 1):
                self.push(first_frame)

    # ------------------------------------------------------------------------------------------ one frame
    def _shift_in(self, buf3, new):
        """buf3: (B, Tb, X) window, new: (B, X): drop the oldest frame, append the newest -- in place, one HIP launch
        (gast_stream_shift_multi; static addresses: graph-capturable).  The raw-input window and every level's window are shifted by
        the same kernel; a level's shift needs the previous block's output of THIS frame, so it is one launch per level (round 2 used
        a torch.cat + copy_ pair per window).  bf16 windows (GAST_HIP_DTYPE=bf16) keep the torch pair."""
        # (the kernel wants an fp32, contiguous window and 16-byte aligned unit-stride rows; anything else takes the torch pair below
        #  instead of raising from the binding -- ADVICE round 3)
        if (buf3.dtype == torch.float32 and new.dtype == torch.float32 and buf3.shape[2] % 4 == 0 and new.stride(0) % 4 == 0
                and buf3.is_contiguous() and new.stride(-1) == 1 and buf3.data_ptr() % 16 == 0 and new.data_ptr() % 16 == 0):
            self.ops.stream_shift_multi([(buf3, new)])
        elif buf3.shape[1] > 1:
            buf3.copy_(torch.cat([buf3[:, 1:], new.unsqueeze(1)], dim=1))
        else:
            buf3[:, 0] = new

    def _step(self):
        sp, ops, eng = self.spec, self.ops, self.engine
        inp, bufs, dt, dev = self.inp, self.bufs, self.dt, self.dev
        B, J, F_in = self.B, self.J, self.F
        pre = eng._pre
        L = len(sp.fw)
        C0 = sp.channels
        eng.za.begin(('stream', B, dt), dev)
        x = self.x_in
        if self.flip is not None:
            xm = x[:, self.perm_in].clone()
            xm[..., 0] *= -1
            x = torch.cat([x, xm], dim=0)
        self._shift_in(self.xwin.view(B, self.Tb[0], J * F_in), x.reshape(B, J * F_in))
        # ---- level 0: init_bn + expand conv on the k0-frame window -> one frame (gast_net.py:163-164)
        k0 = sp.fw[0]
        P = B * J
        E = torch.empty(P, C0, dtype=dt, device=dev)
        partE = torch.empty(ops.rowwise_blocks(P, C0), C0, 2, dtype=torch.float32, device=dev)
        sc0, sh0 = pre['bn0'][0], pre['bn0'][1]
        ops.expand_fwd(self.xwin, B, k0, J, F_in, k0, 1, inp['expand_w'], sc0, sh0, C0, E, partE, center=None)
        X = torch.empty(P, C0, dtype=dt, device=dev)
        ops.bnrelu_apply(E, P, C0, pre['bnE'][0], pre['bnE'][1], X)
        stg = eng._gab_forward(0, X, B, 1, J, C0, inp, bufs, False, dt, None, False, self.adjs[0])
        for s in range(1, L):
            C = C0 * 2 ** s
            Tb, d, k = self.Tb[s], sp.dil[s], sp.fw[s]
            ob = self.obuf[s - 1]
            self._shift_in(ob.view(B, Tb, J * C), stg['O'].view(B, J * C))
            scO, shO = stg['bnO'].scale, stg['bnO'].shift
            Wc, W1 = inp['l%d.conv' % s], inp['l%d.conv1' % s]
            T1 = torch.empty(P, C, dtype=dt, device=dev)
            segs = [dict(A=ob, K=C, map=RowMap(Tb, 1, tap * d), W=Wc[:, tap * C:(tap + 1) * C], pro=PRO_BNRELU, scale=scO, shift=shO)
                    for tap in range(k)]
            ops.gemm((B, 1, J), C, segs, T1, ident(1))
            b1, b2 = pre['l%d.bn1' % s], pre['l%d.bn2' % s]
            T2 = torch.empty(P, C, dtype=dt, device=dev)
            ops.gemm((B, 1, J), C, [dict(A=T1, K=C, map=ident(1), W=W1, pro=PRO_BNRELU, scale=b1[0], shift=b1[1])], T2, ident(1))
            X = torch.empty(P, C, dtype=dt, device=dev)
            # causal residual = the newest frame (pad + causal_shift = (k - 1) d, gast_net.py:142-143,170)
            ops.residual_fwd(ob, RowMap(Tb, 1, Tb - 1), scO, shO, T2, b2[0], b2[1], False, 0, None, B, 1, J, C, X)
            stg = eng._gab_forward(s, X, B, 1, J, C, inp, bufs, False, dt, None, False, self.adjs[s])
        CL = 2 * C0 * 2 ** (L - 1)
        pred = torch.empty(P, 3, dtype=torch.float32, device=dev)
        ops.gemm((B, 1, J), 3, [dict(A=stg['O'], K=CL, map=ident(1), W=inp['shrink'], pro=PRO_BNRELU, scale=stg['bnO'].scale,
                                     shift=stg['bnO'].shift)], pred, ident(1))
        pred = pred.view(B, J, 3)
        if self.flip is not None:
            pm = pred[self.B_user:, self.perm_out].clone()
            pm[..., 0] *= -1
            pred = (pred[:self.B_user] + pm) * 0.5
        self.pred.copy_(pred)
        eng.za.end()

    def push(self, frame):
        """frame: (batch, J, in_features) -> pose (batch, J, 3) of this frame (a new tensor)."""
        if tuple(frame.shape) != (self.B_user, self.J, self.F):
            raise RuntimeError('CausalStream.push: expected a frame of shape %s, got %s' % ((self.B_user, self.J, self.F), tuple(frame.shape)))
        with torch.cuda.device(self.dev), torch.no_grad():
            ops = self.engine.ops
            if self.dt != torch.float32 and hasattr(ops, 'set_h16'):
                ops.set_h16(self.dt)      # (the storage flavour is per THREAD: a stream pushed from another thread than the one that built it)
            self.x_in.copy_(frame)
            if not self.use_graph:
                self._step()
            elif self._graph is None:
                if self.frames < 2:
                    self._step()                 # eager warm-up (lazy workspaces) before the capture
                else:
                    g = torch.cuda.CUDAGraph()
                    from model.gast_net import _no_gc          # (no cyclic collection inside a capture)
                    with _no_gc(), torch.cuda.graph(g, **({'capture_error_mode': 'thread_local'} if (torch.distributed.is_available() and torch.distributed.is_initialized()) else {})):
                        self._step()
                    self._graph = g
                    # (the capture does not execute: run the step it recorded)
                    g.replay()
            else:
                self._graph.replay()
        self.frames += 1
        return self.pred.clone()

    def run(self, clip):
        """clip: (batch, T, J, in_features) -> (batch, T, J, 3): reset with left edge replication, then push every frame."""
        self.reset(clip[:, 0])
        out = [self.push(clip[:, t]) for t in range(clip.shape[1])]
        return torch.stack(out, dim=1)
