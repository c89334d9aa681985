"""Repeat one BNRELU_BWD GEMM case many times and compare outputs bit for bit (a nondeterminism hunt: tests/test_kernels_gpu.py::
test_gemm_bwd_second_output failed once in a full-suite run and never alone)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
os.chdir(os.path.join(ROOT, 'tests'))
import torch
import test_kernels_gpu as tk
from gast_hip.binding import HipOps
ops = HipOps()
if tk.H16 == torch.float16:
    ops.set_h16(torch.float16)
name = sys.argv[1] if len(sys.argv) > 1 else 'big_dgrad_gather_bwd'
mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
case = [c for c in tk.BWD_CASES if c[0] == name][0]
dt = tk.MM_DT[mode]
bad = 0
for it in range(int(os.environ.get('REPS', 150))):
    jd, jh, bufs = tk._gemm_case(case, dt)
    with tk.x3_mode(ops, mode):
        ops.gemm(**jd)
        torch.cuda.synchronize()
        ref = bufs[0].clone(); pref = bufs[2].clone()
        junk = [torch.empty(int(torch.randint(1, 4000, (1,))) * 8, device='cuda') for _ in range(3)]      # perturb the allocator
        bufs[0].fill_(7.0); bufs[2].zero_()
        C2 = torch.full_like(bufs[0], 7.0)
        ops.gemm(**dict(jd, C2=C2[:, :case[2]]))
        torch.cuda.synchronize()
    if not torch.equal(bufs[0], ref):
        d = (bufs[0].float() - ref.float()).abs()
        bad += 1
        print('iter', it, 'C differs: n =', int((d > 0).sum()), 'max', float(d.max()), 'C2 ptr %x' % C2.data_ptr(), flush=True)
    if not torch.equal(bufs[2], pref):
        print('iter', it, 'partials differ', float((bufs[2] - pref).abs().max()), flush=True)
print(name, mode, 'flavour', tk.H16, 'mismatching iterations:', bad)
