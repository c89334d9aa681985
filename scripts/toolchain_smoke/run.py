import ctypes, os, sys, torch, time
here = os.path.dirname(os.path.abspath(__file__))
out = {}
for tag in ('default', 'cov5'):
    lib = ctypes.CDLL(os.path.join(here, 'probe_%s.so' % tag))
    lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.probe_memset.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    s = torch.cuda.current_stream().cuda_stream
    for which, dt, K in ((0, torch.float32, 24), (1, torch.bfloat16, 48)):
        g = torch.Generator().manual_seed(3)
        A = torch.randn(32, K, generator=g).to(dt).cuda()
        Bt = torch.randn(32, K, generator=g).to(dt).cuda()
        D = torch.full((32, 32), 7.0, device='cuda')
        rc0 = lib.probe_memset(D.data_ptr(), D.numel() * 4, s)
        rc = lib.probe_launch(which, A.data_ptr(), Bt.data_ptr(), D.data_ptr(), K, s)
        torch.cuda.synchronize()
        ref = A.float() @ Bt.float().t()
        err = (D - ref).abs().max().item()
        print(tag, 'which', which, 'rc', rc0, rc, 'max err', err, 'ref max', ref.abs().max().item(), flush=True)
print(torch.cuda.get_device_name(0), torch.version.hip)
# graph capture of a ctypes launch on the torch stream
lib = ctypes.CDLL(os.path.join(here, 'probe_default.so'))
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
A = torch.randn(32, 24, device='cuda'); Bt = torch.randn(32, 24, device='cuda'); D = torch.zeros(32, 32, device='cuda')
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    lib.probe_launch(0, A.data_ptr(), Bt.data_ptr(), D.data_ptr(), 24, torch.cuda.current_stream().cuda_stream)
torch.cuda.current_stream().wait_stream(st)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lib.probe_launch(0, A.data_ptr(), Bt.data_ptr(), D.data_ptr(), 24, torch.cuda.current_stream().cuda_stream)
A.copy_(torch.randn(32, 24, device='cuda')); D.zero_()
g.replay(); torch.cuda.synchronize()
print('graph replay err', (D - A @ Bt.t()).abs().max().item())
import subprocess
print(subprocess.run('nproc; lscpu | grep "Model name"; rocminfo | grep -m3 gfx', shell=True, capture_output=True, text=True).stdout)
