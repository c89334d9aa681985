"""ctypes binding of libgast_hip.so (include/gast_hip.h) for torch tensors living on an MI355X.

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device, these functions raise.
Every op enqueues on torch's current stream (so `torch.cuda.graph` capture and stream semantics just work) and
never synchronises.
"""
import ctypes as C
import os
import threading
from collections import namedtuple

import torch

from gast_hip.packer import F8Weight, X3Weight

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgast_hip.so')

GAST_F32, GAST_BF16, GAST_F32X3, GAST_F32X3H = 0, 1, 2, 3
MAX_SEG = 8
PRO_NONE, PRO_BNRELU, PRO_BNRELU_DROP = 0, 1, 2
EPI_PLAIN, EPI_STATS, EPI_BNRELU_BWD = 0, 1, 2

RowMap = namedtuple('RowMap', 'T_total t_stride t_off')
IDENT = lambda T: RowMap(T, 1, 0)  # noqa: E731


class Dropout(namedtuple('Dropout', 'seed thresh inv_keep')):
    """seed: int32/uint32 device tensor with one element; thresh = round(p*65536); inv_keep = 65536/(65536-thresh)."""
    __slots__ = ()


def dropout_params(p):
    if not 0.0 <= p <= 1.0:
        raise ValueError('dropout probability has to be between 0 and 1, got %r' % (p,))
    thresh = int(round(p * 65536))
    if thresh >= 65536:            # p = 1: everything is dropped (nn.Dropout(1.0) returns zeros)
        return 65536, 0.0
    inv_keep = 65536.0 / (65536 - thresh) if thresh else 1.0
    return thresh, inv_keep


class _RowMap(C.Structure):
    _fields_ = [('T_total', C.c_int), ('t_stride', C.c_int), ('t_off', C.c_int)]


class _Dropout(C.Structure):
    _fields_ = [('seed', C.c_void_p), ('thresh', C.c_uint32), ('inv_keep', C.c_float)]


class _GemmSeg(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int), ('K', C.c_int), ('map', _RowMap), ('W', C.c_void_p), ('ldw', C.c_int),
                ('pro', C.c_int), ('scale', C.c_void_p), ('shift', C.c_void_p), ('salt', C.c_uint32), ('Wx', C.c_void_p), ('ldwx', C.c_int)]


class _X3ImageJob(C.Structure):
    _fields_ = [('W', C.c_void_p), ('R', C.c_int), ('K', C.c_int), ('ldw', C.c_int), ('img', C.c_void_p), ('ldimg', C.c_int),
                ('f16', C.c_int)]


class _GemmArgs(C.Structure):
    _fields_ = [('dtype', C.c_int), ('out_f32', C.c_int), ('B', C.c_int), ('Tn', C.c_int), ('J', C.c_int), ('N', C.c_int),
                ('nseg', C.c_int), ('seg', _GemmSeg * MAX_SEG), ('C', C.c_void_p), ('ldc', C.c_int), ('cmap', _RowMap),
                ('bias', C.c_void_p), ('bias_neg', C.c_int), ('addend', C.c_void_p), ('ldadd', C.c_int), ('addmap', _RowMap), ('epi', C.c_int),
                ('partials', C.c_void_p), ('X', C.c_void_p), ('ldx', C.c_int), ('xscale', C.c_void_p), ('xshift', C.c_void_p),
                ('xdrop', C.c_int), ('xsalt', C.c_uint32), ('drop', _Dropout), ('C2', C.c_void_p), ('ldc2', C.c_int), ('f8_scale', C.c_void_p)]


class _F8ScaleJob(C.Structure):
    _fields_ = [('W', C.c_void_p), ('R', C.c_int), ('K', C.c_int), ('ldw', C.c_int), ('out', C.c_void_p)]


class _WgradSeg(C.Structure):
    _fields_ = [('Q', C.c_void_p), ('ldq', C.c_int), ('S', C.c_int), ('map', _RowMap), ('pro', C.c_int), ('scale', C.c_void_p),
                ('shift', C.c_void_p), ('salt', C.c_uint32), ('wcol0', C.c_int)]


class _WgradArgs(C.Structure):
    _fields_ = [('dtype', C.c_int), ('B', C.c_int), ('Tn', C.c_int), ('J', C.c_int), ('P', C.c_void_p), ('ldp', C.c_int),
                ('R', C.c_int), ('pmap', _RowMap), ('nseg', C.c_int), ('seg', _WgradSeg * MAX_SEG), ('dW', C.c_void_p),
                ('ldw', C.c_int), ('zero_first', C.c_int), ('drop', _Dropout)]


class _AdjJob(C.Structure):
    _fields_ = [('e', C.c_void_p), ('C', C.c_int), ('pat', C.c_void_p), ('A_t', C.c_void_p), ('dA_t', C.c_void_p)]


class _StreamShiftJob(C.Structure):
    _fields_ = [('buf', C.c_void_p), ('newest', C.c_void_p), ('B', C.c_int), ('Tb', C.c_int), ('X', C.c_int), ('ldnew', C.c_int)]


class _RowsumJob(C.Structure):
    _fields_ = [('ws', C.c_void_p), ('nrow', C.c_int), ('ncol', C.c_long), ('nb', C.c_long), ('out0', C.c_void_p), ('out1', C.c_void_p),
                ('accumulate', C.c_int)]


class _ZeroJob(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('bytes', C.c_long)]


PREP_MAX_ZERO = 6


class _PrepArgs(C.Structure):
    _fields_ = [('zero', _ZeroJob * PREP_MAX_ZERO), ('nzero', C.c_int), ('seed_ctr', C.c_void_p), ('seed_out', C.c_void_p),
                ('pad_src', C.c_void_p), ('pad_dst', C.c_void_p), ('pad_rows', C.c_long), ('pad_cols_src', C.c_int),
                ('pad_cols_dst', C.c_int), ('pad_scale', C.c_float), ('pad_dst_h16', C.c_int)]


class _BnEvalJob(C.Structure):
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('N', C.c_int),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('centered', C.c_int)]


class _BnFinJob(C.Structure):
    _fields_ = [('partials', C.c_void_p), ('nblk', C.c_int), ('ncol_total', C.c_int), ('col0', C.c_int), ('N', C.c_int),
                ('count', C.c_double), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p),
                ('running_var', C.c_void_p), ('num_batches_tracked', C.c_void_p), ('momentum', C.c_float), ('eps', C.c_float),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('centered', C.c_int)]


class _BnBwdFinJob(C.Structure):
    _fields_ = [('partials', C.c_void_p), ('nblk', C.c_int), ('ncol_total', C.c_int), ('col0', C.c_int), ('N', C.c_int),
                ('count', C.c_double), ('gamma', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('dgamma', C.c_void_p),
                ('dbeta', C.c_void_p), ('ka', C.c_void_p), ('kb', C.c_void_p), ('kc', C.c_void_p), ('accumulate', C.c_int)]


class _BnBwdJob(C.Structure):
    _fields_ = [('f', _BnBwdFinJob), ('dz', C.c_void_p), ('lddz', C.c_int), ('X', C.c_void_p), ('ldx', C.c_int), ('rows', C.c_long)]


# The library exists in two 16-bit STORAGE flavours built from the same sources (csrc/build.sh, csrc/common.h): libgast_hip.so keeps
# GAST_BF16 tensors as bfloat16, libgast_hip_f16.so as IEEE binary16 (GAST_HIP_DTYPE=f16).  Same ABI; the process-wide flavour
# (`set_h16`, per thread) decides which one the op set calls and which 16-bit torch dtype `_dt` accepts -- a tensor of the other kind raises.
LIB_PATH_F16 = os.path.join(_HERE, 'libgast_hip_f16.so')
_libs = {}


class _Flavour(threading.local):
    """The 16-bit storage flavour is per THREAD: every entry point of the plan (forward, backward -- which autograd runs
    on its own thread --, the causal stream, DataParallel replica threads) selects it for its own launches, so two models of different
    16-bit dtypes driven from different threads cannot flip it under each other."""
    dtype = torch.bfloat16


_H16F = _Flavour()


class _H16View:
    """(dict-style access kept for the call sites below)"""
    def __getitem__(self, k):
        return _H16F.dtype

    def __setitem__(self, k, v):
        _H16F.dtype = v


_H16 = _H16View()


def set_h16(dtype):
    """Select the 16-bit storage flavour for every following call: torch.bfloat16 (default) or torch.float16."""
    if dtype not in (torch.bfloat16, torch.float16):
        raise ValueError('gast_hip: the 16-bit storage type is torch.bfloat16 or torch.float16, got %r' % (dtype,))
    if dtype == torch.float16:
        load_library(torch.float16)        # (fail here, loudly, if the flavour was not built)
    _H16['dtype'] = dtype


def h16_dtype():
    return _H16['dtype']


def load_library(h16=torch.bfloat16):
    """Load libgast_hip.so (or its binary16 flavour) or raise -- the HIP path is the only path."""
    lib = _libs.get(h16)
    if lib is not None:
        return lib
    path = LIB_PATH_F16 if h16 == torch.float16 else LIB_PATH
    if h16 != torch.float16 and os.environ.get('GAST_HIP_LIB_EXPERIMENT'):      # kernel A/B builds (scripts/): a sibling library by suffix
        path = os.path.join(_HERE, 'libgast_hip_%s.so' % os.environ['GAST_HIP_LIB_EXPERIMENT'])
    if not os.path.exists(path):
        raise RuntimeError('gast_hip: %s not found -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.' % path)
    lib = C.CDLL(path)
    vp, ci, cl, cf, cd, cu = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_uint32
    sig = {
        'gast_gemm': [C.POINTER(_GemmArgs), vp],
        'gast_gemm_row_blocks': [ci],
        'gast_gemm_ws': [C.POINTER(_GemmArgs), vp, cl, vp],
        'gast_gemm_multi': [C.POINTER(_GemmArgs), ci, vp, cl, vp],
        'gast_gemm_splitk_ws_bytes': [cl, ci],
        'gast_gemm_path': [C.POINTER(_GemmArgs)],
        'gast_f8_scale_multi': [C.POINTER(_F8ScaleJob), ci, vp],
        'gast_x3_image_multi': [C.POINTER(_X3ImageJob), ci, vp],
        'gast_x3_image_ld': [ci],
        'gast_wgrad': [C.POINTER(_WgradArgs), vp],
        'gast_wgrad_multi': [C.POINTER(_WgradArgs), ci, vp],
        'gast_semch_adj_fwd': [vp, ci, vp, vp, vp],
        'gast_semch_adj_bwd': [vp, vp, ci, vp, vp, vp],
        'gast_semch_adj_multi': [C.POINTER(_AdjJob), ci, ci, vp],
        'gast_semch_agg_fwd': [ci, vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp, ci, vp, vp, vp, vp],
        'gast_semch_agg_blocks': [ci, ci],
        'gast_semch_agg_bwd': [ci, vp, ci, vp, ci, ci, ci, ci, vp, vp, ci, ci, vp, vp, ci, ci, vp, ci, vp, vp, vp],
        'gast_semch_agg_bwd_ws_floats': [ci, ci, ci, ci],
        'gast_semch_agg_bwd_fuses_bn': [ci, ci, ci, ci, ci, ci, ci, ci],
        'gast_semch_agg_bwd_bn': [ci, vp, ci, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, vp, vp, ci, ci, vp, ci, vp, vp,
                                  C.POINTER(_RowsumJob), vp],
        'gast_attn_fwd': [ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp],
        'gast_attn_bwd': [ci, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, vp, vp, vp, vp],
        'gast_attn_bwd_ws_floats': [ci, ci, ci, ci],
        'gast_rowsum_multi': [C.POINTER(_RowsumJob), ci, vp],
        'gast_attn_bwd_deferred': [ci, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, vp, vp, vp, C.POINTER(_RowsumJob), vp],
        'gast_semch_agg_bwd_deferred': [ci, vp, ci, vp, ci, ci, ci, ci, vp, vp, ci, ci, vp, vp, ci, ci, vp, ci, vp, vp, C.POINTER(_RowsumJob), vp],
        'gast_bn_finalize': [vp, ci, ci, ci, ci, cd, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, ci, vp],
        'gast_bn_eval': [vp, vp, vp, vp, cf, ci, vp, vp, ci, vp],
        'gast_bn_finalize_multi': [C.POINTER(_BnFinJob), ci, vp],
        'gast_bn_eval_multi': [C.POINTER(_BnEvalJob), ci, cf, vp],
        'gast_bn_bwd_finalize_multi': [C.POINTER(_BnBwdFinJob), ci, vp],
        'gast_bn_bwd_fused_multi': [ci, C.POINTER(_BnBwdJob), ci, vp],
        'gast_bn_bwd_finalize': [vp, ci, ci, ci, ci, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        'gast_bn_bwd_apply': [ci, vp, ci, vp, ci, cl, ci, vp, vp, vp, vp],
        'gast_bn_bwd_apply_frames': [ci, vp, ci, vp, ci, cl, ci, vp, vp, vp, ci, ci, C.c_ulonglong, vp],
        'gast_bnrelu_apply': [ci, vp, ci, cl, ci, vp, vp, vp, ci, ci, cu, _Dropout, vp],
        'gast_bnrelu_bwd_mask': [ci, vp, ci, vp, ci, cl, ci, vp, vp, ci, cu, _Dropout, vp, ci, vp, vp],
        'gast_shrink_fwd': [ci, vp, ci, cl, ci, vp, vp, vp, ci, ci, vp, ci, vp],
        'gast_shrink_bwd_blocks': [cl],
        'gast_shrink_bwd': [ci, vp, ci, vp, ci, ci, vp, ci, vp, vp, cl, ci, vp, ci, vp, vp],
        'gast_rowwise_blocks': [cl, ci],
        'gast_residual_fwd': [ci, vp, ci, _RowMap, vp, vp, vp, ci, vp, vp, ci, cu, _Dropout, ci, ci, ci, ci, vp, ci, vp],
        'gast_input_stats': [vp, cl, ci, vp, vp, vp],
        'gast_input_stats_blocks': [cl],
        'gast_expand_fwd': [ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, ci, vp, vp, vp],
        'gast_expand_bwd': [ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp],
        'gast_expand_bwd_ws_floats': [cl, ci, ci, ci],
        'gast_expand_bwd_bn': [ci, vp, ci, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp],
        'gast_colsum': [ci, vp, ci, cl, ci, vp, ci, vp],
        'gast_strided_copy': [vp, vp, ci, vp, vp],
        'gast_fold': [vp, ci, ci, vp, vp],
        'gast_pack_all': [vp, vp, ci, vp, ci, ci, vp, vp],
        'gast_unfold': [vp, ci, ci, vp, vp],
        'gast_mpjpe': [vp, vp, cl, ci, vp, vp, vp],
        'gast_adam_step': [vp, vp, vp, vp, vp, cl, vp, cf, cf, cf, cf, cf, cf, vp],
        'gast_adam_step_guarded': [vp, vp, vp, vp, vp, cl, vp, cf, cf, cf, cf, cf, cf, vp, vp, vp],
        'gast_nonfinite_scan': [vp, cl, vp, vp],
        'gast_null_launch': [vp],
        'gast_prep': [C.POINTER(_PrepArgs), vp],
        'gast_chunk_gather': [vp, vp, vp, vp, vp, cl, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp],
        'gast_stream_shift_multi': [C.POINTER(_StreamShiftJob), ci, vp],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ci
    lib.gast_semch_agg_bwd_ws_floats.restype = C.c_long
    lib.gast_gemm_splitk_ws_bytes.restype = C.c_long
    lib.gast_x3_image_ld.restype = C.c_long
    lib.gast_expand_bwd_ws_floats.restype = C.c_long
    lib.gast_version.restype = C.c_char_p
    lib.gast_version.argtypes = []
    _libs[h16] = lib
    return lib


EXPORTED_SYMBOLS = ['gast_gemm', 'gast_gemm_ws', 'gast_gemm_multi', 'gast_gemm_splitk_ws_bytes', 'gast_gemm_row_blocks', 'gast_gemm_path', 'gast_f8_scale_multi', 'gast_x3_image_multi', 'gast_x3_image_ld', 'gast_wgrad', 'gast_wgrad_multi', 'gast_semch_adj_fwd', 'gast_semch_adj_bwd', 'gast_semch_adj_multi',
                    'gast_semch_agg_fwd', 'gast_semch_agg_blocks', 'gast_semch_agg_bwd', 'gast_semch_agg_bwd_ws_floats', 'gast_semch_agg_bwd_fuses_bn', 'gast_semch_agg_bwd_bn', 'gast_attn_fwd', 'gast_attn_bwd', 'gast_attn_bwd_ws_floats', 'gast_rowsum_multi', 'gast_attn_bwd_deferred', 'gast_semch_agg_bwd_deferred',
                    'gast_bn_finalize', 'gast_bn_finalize_multi', 'gast_bn_eval', 'gast_bn_eval_multi', 'gast_bn_bwd_finalize', 'gast_bn_bwd_finalize_multi', 'gast_bn_bwd_fused_multi', 'gast_bn_bwd_apply', 'gast_bn_bwd_apply_frames', 'gast_bnrelu_apply',
                    'gast_bnrelu_bwd_mask', 'gast_shrink_fwd', 'gast_shrink_bwd_blocks', 'gast_shrink_bwd', 'gast_rowwise_blocks', 'gast_residual_fwd', 'gast_input_stats',
                    'gast_input_stats_blocks', 'gast_expand_fwd', 'gast_expand_bwd', 'gast_expand_bwd_bn', 'gast_expand_bwd_ws_floats', 'gast_colsum', 'gast_strided_copy', 'gast_pack_all', 'gast_fold',
                    'gast_unfold', 'gast_mpjpe', 'gast_adam_step', 'gast_adam_step_guarded', 'gast_nonfinite_scan', 'gast_prep', 'gast_null_launch', 'gast_chunk_gather', 'gast_stream_shift_multi', 'gast_version']


def _check(rc, what):
    if rc != 0:
        names = {-1: 'GAST_EINVAL', -2: 'GAST_EALIGN', -3: 'GAST_ERANGE'}
        raise RuntimeError('gast_hip: %s failed with %s' % (what, names.get(rc, 'hipError %d' % rc)))


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('gast_hip: tensor is on %s; the HIP path needs device tensors (no CPU fallback)' % t.device)
    return t.data_ptr()


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError('gast_hip: expected a 2-D row-major view, got shape %s strides %s' % (tuple(t.shape), t.stride()))
    return t.stride(0)


def _dt(t):
    if t.dtype == torch.float32:
        return GAST_F32
    if t.dtype in (torch.bfloat16, torch.float16):
        if t.dtype != _H16['dtype']:
            raise RuntimeError('gast_hip: %s tensor, but the selected 16-bit storage flavour is %s (gast_hip.binding.set_h16)'
                               % (t.dtype, _H16['dtype']))
        return GAST_BF16          # ("the 16-bit storage type of the loaded flavour")
    raise RuntimeError('gast_hip: unsupported dtype %s' % t.dtype)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rm(m):
    return _RowMap(int(m.T_total), int(m.t_stride), int(m.t_off))


def _drop(d):
    if d is None:
        return _Dropout(None, 0, 1.0)
    return _Dropout(_p(d.seed), int(d.thresh), float(d.inv_keep))


class HipOps:
    """The op set of include/gast_hip.h on torch device tensors.  tests/fake_backend.py mirrors this interface on
    CPU tensors with the numpy oracle (tests only)."""
    name = 'hip'

    @property
    def lib(self):
        return load_library(_H16['dtype'])

    set_h16 = staticmethod(set_h16)

    def __init__(self):
        load_library()
        self.launches = 0
        # fp32 tensors: run the MFMA GEMMs / weight gradients on split-bf16 products (GAST_F32X3, include/gast_hip.h) instead of
        # the fp32 matrix instruction.  Set by the model runner from GAST_HIP_DTYPE=bf16x3; storage stays fp32 everywhere.
        self.x3 = False
        # bf16 tensors: GEMMs whose weight operands all carry an fp8 scale (the forward GEMMs, gast_hip/packer.py) run with e4m3
        # operands on v_mfma_f32_32x32x16_fp8_fp8 (GAST_HIP_DTYPE=fp8, BASELINE.json configs[4]); gradients stay bf16
        self.f8 = False
        self._ws = {}            # per (device, stream): fp32 split-K workspace (allocated once, before any graph capture)

    SPLITK_WS_BYTES = 96 << 20

    def _splitk_ws(self, dev):
        """One workspace per launch stream: the engine runs independent branches of the plan on a side stream."""
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.empty(self.SPLITK_WS_BYTES // 4, dtype=torch.float32, device=dev)
        return ws

    # -- GEMM family
    def gemm_row_blocks(self, M):
        return self.lib.gast_gemm_row_blocks(int(M))

    def _gemm_args(self, a, dom, N, segs, C_, cmap, bias=None, addend=None, addmap=None, epi=EPI_PLAIN, partials=None, X=None,
                   xscale=None, xshift=None, xdrop=False, xsalt=0, drop=None, bias_neg=False, C2=None):
        a.dtype = _dt(segs[0]['A'])
        a.out_f32 = 1 if (C_.dtype == torch.float32 and a.dtype == GAST_BF16) else 0
        st_dtype = a.dtype
        a.B, a.Tn, a.J = (int(v) for v in dom)
        a.N = int(N)
        a.nseg = len(segs)
        f8_scales = []
        nf16 = 0
        for i, s in enumerate(segs):
            g = a.seg[i]
            W, img = s['W'], None
            if isinstance(W, X3Weight):
                nf16 += W.f16
                W, img = W.t, W.img
            elif isinstance(W, F8Weight):
                f8_scales.append(W.scale)
                W = W.t
            if _dt(s['A']) != st_dtype or _dt(W) != st_dtype:
                raise RuntimeError('gast_hip: mixed operand dtypes in gemm')
            g.A, g.lda, g.K, g.map = _p(s['A']), _ld(s['A']), int(s['K']), _rm(s['map'])
            g.W, g.ldw = _p(W), _ld(W)
            if img is not None and (self.x3 or st_dtype == GAST_BF16):
                g.Wx, g.ldwx = _p(img), img.stride(0)
            g.pro = int(s.get('pro', PRO_NONE))
            g.scale, g.shift = _p(s.get('scale')), _p(s.get('shift'))
            g.salt = int(s.get('salt', 0))
        a.C, a.ldc, a.cmap = _p(C_), _ld(C_), _rm(cmap)
        a.bias = _p(bias)
        a.bias_neg = int(bool(bias_neg))
        if addend is not None:
            a.addend, a.ldadd, a.addmap = _p(addend), _ld(addend), _rm(addmap)
        a.epi = int(epi)
        a.partials = _p(partials)
        if X is not None:
            a.X, a.ldx = _p(X), _ld(X)
        a.xscale, a.xshift = _p(xscale), _p(xshift)
        a.xdrop, a.xsalt = int(bool(xdrop)), int(xsalt)
        a.drop = _drop(drop)
        if C2 is not None:          # BNRELU_BWD: second output, the value before the mask (same row map as C_)
            if C2.dtype != C_.dtype or C2.shape[0] != C_.shape[0]:
                raise RuntimeError('gast_hip: gemm C2 must have the dtype and rows of C_')
            a.C2, a.ldc2 = _p(C2), _ld(C2)
        if self.x3 and st_dtype == GAST_F32:
            # fp16 pairs when the weight operands say so (the forward operands of the plan, gast_hip/packer.py); one GEMM, one kind
            if nf16 not in (0, len(segs)):
                raise RuntimeError('gast_hip: fp16-pair and bf16-pair weight operands in one gemm')
            a.dtype = GAST_F32X3H if nf16 else GAST_F32X3
        if (self.f8 and st_dtype == GAST_BF16 and not a.out_f32 and len(f8_scales) == len(segs)
                and all(sc.data_ptr() == f8_scales[0].data_ptr() for sc in f8_scales)):
            a.f8_scale = _p(f8_scales[0])

    def gemm(self, dom, N, segs, C_, cmap, **kw):
        a = _GemmArgs()
        self._gemm_args(a, dom, N, segs, C_, cmap, **kw)
        self.launches += 1
        ws = self._splitk_ws(C_.device)
        _check(self.lib.gast_gemm_ws(C.byref(a), ws.data_ptr(), ws.numel() * 4, _stream()), 'gast_gemm')

    GEMM_MAX_BATCH = 3

    def gemm_path(self, dom, N, segs, C_, cmap, **kw):
        """0 / 1: the kernel gemm() would launch for these arguments (gast_gemm_path)"""
        a = _GemmArgs()
        self._gemm_args(a, dom, N, segs, C_, cmap, **kw)
        return self.lib.gast_gemm_path(C.byref(a))

    def x3_weight(self, W, f16=False):
        """fp32 [N][K] row-major operand -> X3Weight carrying a freshly built pre-split image (gast_x3_image_multi; bf16 pairs, or
        fp16 pairs for a forward operand): what Packer.inputs() hands the engine for every packed operand in GAST_F32X3 mode."""
        if W.dtype != torch.float32:
            raise RuntimeError('gast_hip: x3_weight needs an fp32 operand')
        ld = int(self.lib.gast_x3_image_ld(int(W.shape[0])))
        img = torch.zeros((W.shape[1] + 15) // 16, ld // 32, 32, dtype=torch.bfloat16, device=W.device)
        job = (_X3ImageJob * 1)()
        job[0].W, job[0].R, job[0].K, job[0].ldw, job[0].img, job[0].ldimg = _p(W), W.shape[0], W.shape[1], _ld(W), _p(img), ld
        job[0].f16 = int(bool(f16))
        self.launches += 1
        _check(self.lib.gast_x3_image_multi(job, 1, _stream()), 'gast_x3_image_multi')
        return X3Weight(W, img, f16)

    def h16_weight(self, W):
        """16-bit [N][K] row-major operand (K % 8 == 0) -> X3Weight(group = 32) carrying its k-group-major layout image
        (gast_x3_image_multi, kind 2): what Packer.inputs() hands the engine in the 16-bit modes."""
        if W.dtype == torch.float32:
            raise RuntimeError('gast_hip: h16_weight needs a 16-bit operand')
        ld = int(self.lib.gast_x3_image_ld(int(W.shape[0])))
        img = torch.zeros((W.shape[1] + 31) // 32, ld // 32, 32, dtype=W.dtype, device=W.device)
        job = (_X3ImageJob * 1)()
        job[0].W, job[0].R, job[0].K, job[0].ldw, job[0].img, job[0].ldimg = _p(W), W.shape[0], W.shape[1], _ld(W), _p(img), ld
        job[0].f16 = 2
        self.launches += 1
        _check(self.lib.gast_x3_image_multi(job, 1, _stream()), 'gast_x3_image_multi')
        return X3Weight(W, img, False, 32)

    def gemm_multi(self, jobs):
        """jobs: list of dicts with the arguments of gemm() (keys dom, N, segs, C_, cmap + keywords): independent GEMMs of one plan
        step, launched as one grid (GEMM_MAX_BATCH at a time)."""
        for i0 in range(0, len(jobs), self.GEMM_MAX_BATCH):
            chunk = jobs[i0:i0 + self.GEMM_MAX_BATCH]
            arr = (_GemmArgs * len(chunk))()
            for a, j in zip(arr, chunk):
                self._gemm_args(a, **j)
            self.launches += 1
            ws = self._splitk_ws(chunk[0]['C_'].device)
            _check(self.lib.gast_gemm_multi(arr, len(chunk), ws.data_ptr(), ws.numel() * 4, _stream()), 'gast_gemm_multi')

    def _wgrad_args(self, a, dom, P, R, pmap, segs, dW, drop=None, zero_first=True):
        a.dtype = st_dtype = _dt(P)
        a.B, a.Tn, a.J = (int(v) for v in dom)
        a.P, a.ldp, a.R, a.pmap = _p(P), _ld(P), int(R), _rm(pmap)
        a.nseg = len(segs)
        for i, s in enumerate(segs):
            g = a.seg[i]
            if _dt(s['Q']) != st_dtype:
                raise RuntimeError('gast_hip: mixed operand dtypes in wgrad')
            g.Q, g.ldq, g.S, g.map = _p(s['Q']), _ld(s['Q']), int(s['S']), _rm(s['map'])
            g.pro = int(s.get('pro', PRO_NONE))
            g.scale, g.shift = _p(s.get('scale')), _p(s.get('shift'))
            g.salt, g.wcol0 = int(s.get('salt', 0)), int(s['wcol0'])
        if dW.dtype != torch.float32:
            raise RuntimeError('gast_hip: dW must be fp32')
        a.dW, a.ldw, a.zero_first = _p(dW), _ld(dW), int(bool(zero_first))
        a.drop = _drop(drop)
        if self.x3 and st_dtype == GAST_F32:
            a.dtype = GAST_F32X3

    def wgrad(self, dom, P, R, pmap, segs, dW, drop=None, zero_first=True):
        a = _WgradArgs()
        self._wgrad_args(a, dom, P, R, pmap, segs, dW, drop, zero_first)
        self.launches += 1
        _check(self.lib.gast_wgrad(C.byref(a), _stream()), 'gast_wgrad')

    WGRAD_MAX_BATCH = 8

    def wgrad_multi(self, jobs):
        """jobs: list of dicts with the keyword arguments of wgrad(); launched WGRAD_MAX_BATCH at a time."""
        for i0 in range(0, len(jobs), self.WGRAD_MAX_BATCH):
            chunk = jobs[i0:i0 + self.WGRAD_MAX_BATCH]
            arr = (_WgradArgs * len(chunk))()
            for a, j in zip(arr, chunk):
                self._wgrad_args(a, **j)
            self.launches += 1
            _check(self.lib.gast_wgrad_multi(arr, len(chunk), _stream()), 'gast_wgrad_multi')

    # -- SemCH graph conv
    def semch_adj_fwd(self, e, pat, A_t):
        self.launches += 1
        _check(self.lib.gast_semch_adj_fwd(_p(e), e.shape[0], _p(pat), _p(A_t), _stream()), 'gast_semch_adj_fwd')

    def semch_adj_bwd(self, dA_t, A_t, pat, de):
        self.launches += 1
        _check(self.lib.gast_semch_adj_bwd(_p(dA_t), _p(A_t), de.shape[0], _p(pat), _p(de), _stream()), 'gast_semch_adj_bwd')

    ADJ_MAX_BATCH = 8

    def semch_adj_fwd_multi(self, jobs):
        """jobs: (e, pat, A_t) triples -- every adjacency softmax of the forward pass in one launch."""
        for i0 in range(0, len(jobs), self.ADJ_MAX_BATCH):
            chunk = jobs[i0:i0 + self.ADJ_MAX_BATCH]
            arr = (_AdjJob * len(chunk))()
            for a, (e, pat, A_t) in zip(arr, chunk):
                a.e, a.C, a.pat, a.A_t, a.dA_t = _p(e), e.shape[0], _p(pat), _p(A_t), None
            self.launches += 1
            _check(self.lib.gast_semch_adj_multi(arr, len(chunk), 0, _stream()), 'gast_semch_adj_multi')

    def semch_adj_bwd_multi(self, jobs, accumulate=False):
        """jobs: (dA_t, A_t, pat, de) tuples -- every adjacency softmax backward of the pass in one launch (de written, or +=)."""
        for i0 in range(0, len(jobs), self.ADJ_MAX_BATCH):
            chunk = jobs[i0:i0 + self.ADJ_MAX_BATCH]
            arr = (_AdjJob * len(chunk))()
            for a, (dA_t, A_t, pat, de) in zip(arr, chunk):
                a.e, a.C, a.pat, a.A_t, a.dA_t = _p(de), de.shape[0], _p(pat), _p(A_t), _p(dA_t)
            self.launches += 1
            _check(self.lib.gast_semch_adj_multi(arr, len(chunk), 2 if accumulate else 1, _stream()), 'gast_semch_adj_multi')

    def semch_agg_blocks(self, F, C_):
        return self.lib.gast_semch_agg_blocks(int(F), int(C_))

    def semch_agg_fwd(self, H, F, J, C_, A_sym, pat_sym, A_con, pat_con, Y, partials, deg=(0, 0), center=(None, None)):
        """A_*: [nnz+1][C] (row nnz all zero); deg = (Dr_sym, Dr_con) of the pattern tables selects the unrolled kernels;
        center = (bn_1.running_mean, bn_2.running_mean) or Nones: subtracted from the stored outputs."""
        self.launches += 1
        _check(self.lib.gast_semch_agg_fwd(_dt(H), _p(H), _ld(H), F, J, C_, _p(A_sym), _p(pat_sym), int(deg[0]), _p(A_con),
                                           _p(pat_con), int(deg[1]), _p(Y), _ld(Y), _p(partials), _p(center[0]), _p(center[1]),
                                           _stream()), 'gast_semch_agg_fwd')

    def semch_agg_bwd_ws(self, F, C_, nnz_sym, nnz_con):
        return self.lib.gast_semch_agg_bwd_ws_floats(int(F), int(C_), int(nnz_sym), int(nnz_con))

    ROWSUM_MAX_BATCH = 8

    def rowsum_multi(self, jobs):
        """jobs: the (job struct, tensors kept alive) pairs the deferred forms of attn_bwd / semch_agg_bwd appended: their finishes
        (dbias / dC_k += column sums of the partial rows, dA = column sums) as one launch per ROWSUM_MAX_BATCH."""
        for i0 in range(0, len(jobs), self.ROWSUM_MAX_BATCH):
            chunk = jobs[i0:i0 + self.ROWSUM_MAX_BATCH]
            arr = (_RowsumJob * len(chunk))()
            for a, (j, _keep) in zip(arr, chunk):
                a.ws, a.nrow, a.ncol, a.nb, a.out0, a.out1, a.accumulate = j.ws, j.nrow, j.ncol, j.nb, j.out0, j.out1, j.accumulate
            self.launches += 1
            _check(self.lib.gast_rowsum_multi(arr, len(chunk), _stream()), 'gast_rowsum_multi')

    def semch_agg_bwd_fuses_bn(self, H, F, J, C_, A_sym, A_con, cdeg):
        """can semch_agg_bwd(bn=...) apply the BatchNorm backward of dY while it stages it (the LDS-staged kernel: fp32 storage, the
        shipped skeletons' column degrees)?"""
        return bool(self.lib.gast_semch_agg_bwd_fuses_bn(_dt(H), int(F), int(J), int(C_), A_sym.shape[0] - 1, int(cdeg[0]), A_con.shape[0] - 1,
                                                         int(cdeg[1])))

    def semch_agg_bwd(self, dY, H, F, J, C_, A_sym, pat_sym, A_con, pat_con, dH, dA, ws, cdeg=(0, 0), defer=None, bn=None):
        """dA: [nnz_sym + nnz_con][C] fp32 (sym rows first), fully written; ws: workspace of semch_agg_bwd_ws() floats;
        cdeg = (Dc_sym, Dc_con).  defer: a list -- the final row reduction into dA is not launched but appended to it (rowsum_multi).
        bn = (Ypre, ka, kb, kc): dY is the gradient BEFORE the BatchNorm backward of bn_1 | bn_2 and the kernel applies
        ka*dY + kb*Ypre + kc while it stages dY (ask semch_agg_bwd_fuses_bn first)."""
        if bn is not None:
            Yp, ka, kb, kc = bn
            job = _RowsumJob()
            self.launches += 1 if defer is not None else 2
            _check(self.lib.gast_semch_agg_bwd_bn(_dt(H), _p(dY), _ld(dY), _p(Yp), _ld(Yp), _p(ka), _p(kb), _p(kc), _p(H), _ld(H),
                                                  F, J, C_, _p(A_sym), _p(pat_sym), A_sym.shape[0] - 1, int(cdeg[0]), _p(A_con), _p(pat_con),
                                                  A_con.shape[0] - 1, int(cdeg[1]), _p(dH), _ld(dH), _p(dA), _p(ws),
                                                  C.byref(job) if defer is not None else None, _stream()), 'gast_semch_agg_bwd_bn')
            if defer is not None and job.ws:
                defer.append((job, (ws, dA)))
            return
        if defer is not None:
            job = _RowsumJob()
            self.launches += 1
            _check(self.lib.gast_semch_agg_bwd_deferred(_dt(H), _p(dY), _ld(dY), _p(H), _ld(H), F, J, C_, _p(A_sym), _p(pat_sym),
                                                        A_sym.shape[0] - 1, int(cdeg[0]), _p(A_con), _p(pat_con), A_con.shape[0] - 1,
                                                        int(cdeg[1]), _p(dH), _ld(dH), _p(dA), _p(ws), C.byref(job), _stream()),
                   'gast_semch_agg_bwd_deferred')
            if job.ws:
                defer.append((job, (ws, dA)))
            return
        self.launches += 2
        _check(self.lib.gast_semch_agg_bwd(_dt(H), _p(dY), _ld(dY), _p(H), _ld(H), F, J, C_, _p(A_sym), _p(pat_sym),
                                           A_sym.shape[0] - 1, int(cdeg[0]), _p(A_con), _p(pat_con), A_con.shape[0] - 1, int(cdeg[1]),
                                           _p(dH), _ld(dH), _p(dA), _p(ws), _stream()), 'gast_semch_agg_bwd')

    # -- global attention
    def attn_fwd(self, G, AC, C_k, F, J, C_, nheads, Y):
        self.launches += 1
        _check(self.lib.gast_attn_fwd(_dt(G), _p(G), _ld(G), _p(AC), _ld(AC), _p(C_k), F, J, C_, nheads, _p(Y), _ld(Y), _stream()),
               'gast_attn_fwd')

    def attn_bwd(self, dY, G, AC, C_k, F, J, C_, nheads, dG, dAC, dC_k, dbias=None, generic=False, defer=None):
        """dC_k and dbias ([C + 2*nheads]: column sums of [dG | dAC]) are accumulated into (zero-filled by the caller).
        defer: a list -- the reduction of the per-wave partial rows into dbias / dC_k is appended to it instead of launched."""
        self.launches += 2
        ws = None
        if not generic:
            ws = torch.empty(max(1, self.lib.gast_attn_bwd_ws_floats(F, J, C_, nheads)), dtype=torch.float32, device=G.device)
        if defer is not None:
            job = _RowsumJob()
            self.launches -= 1
            _check(self.lib.gast_attn_bwd_deferred(_dt(G), _p(dY), _ld(dY), _p(G), _ld(G), _p(AC), _ld(AC), _p(C_k), F, J, C_, nheads,
                                                   _p(dG), _ld(dG), _p(dAC), _ld(dAC), _p(dC_k), _p(dbias), _p(ws), C.byref(job), _stream()),
                   'gast_attn_bwd_deferred')
            if job.ws:
                defer.append((job, (ws, dC_k, dbias)))
            return
        _check(self.lib.gast_attn_bwd(_dt(G), _p(dY), _ld(dY), _p(G), _ld(G), _p(AC), _ld(AC), _p(C_k), F, J, C_, nheads,
                                      _p(dG), _ld(dG), _p(dAC), _ld(dAC), _p(dC_k), _p(dbias), _p(ws), _stream()), 'gast_attn_bwd')

    # -- BatchNorm pieces
    def bn_finalize(self, partials, nblk, col0, N, count, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                    scale, shift, mean, rstd, centered=False):
        self.launches += 1
        _check(self.lib.gast_bn_finalize(_p(partials), nblk, partials.shape[1], col0, N, float(count), _p(gamma), _p(beta),
                                         _p(running_mean), _p(running_var), _p(nbt), momentum, eps, _p(scale), _p(shift),
                                         _p(mean), _p(rstd), int(bool(centered)), _stream()), 'gast_bn_finalize')

    BN_MAX_BATCH = 4

    @staticmethod
    def _fill_fin(a, j):
        pt = j['partials']
        a.partials, a.nblk, a.ncol_total, a.col0, a.N, a.count = _p(pt), j['nblk'], pt.shape[1], j['col0'], j['N'], float(j['count'])
        a.gamma, a.beta = _p(j['gamma']), _p(j['beta'])
        a.running_mean, a.running_var, a.num_batches_tracked = _p(j['running_mean']), _p(j['running_var']), _p(j['nbt'])
        a.momentum, a.eps = j['momentum'], j['eps']
        a.scale, a.shift, a.mean, a.rstd = _p(j['scale']), _p(j['shift']), _p(j['mean']), _p(j['rstd'])
        a.centered = int(bool(j.get('centered', False)))

    prep_pads_h16 = True  # gast_prep pads d loss / d pred into the 16-bit storage type, with the loss scale (round 6)
    fuses_bn_bwd = True  # semch_agg_bwd(bn=...) / expand_bwd(bn=...) apply the BatchNorm backward of their input gradient on load

    def bn_finalize_multi(self, jobs):
        """jobs: dicts with the arguments of bn_finalize(); BN_MAX_BATCH per launch."""
        for i0 in range(0, len(jobs), self.BN_MAX_BATCH):
            chunk = jobs[i0:i0 + self.BN_MAX_BATCH]
            arr = (_BnFinJob * len(chunk))()
            for a, j in zip(arr, chunk):
                self._fill_fin(a, j)
            self.launches += 1
            _check(self.lib.gast_bn_finalize_multi(arr, len(chunk), _stream()), 'gast_bn_finalize_multi')

    def bn_bwd_finalize_multi(self, jobs):
        """jobs: dicts with the arguments of bn_bwd_finalize(); BN_MAX_BATCH per launch."""
        for i0 in range(0, len(jobs), self.BN_MAX_BATCH):
            chunk = jobs[i0:i0 + self.BN_MAX_BATCH]
            arr = (_BnBwdFinJob * len(chunk))()
            for a, j in zip(arr, chunk):
                pt = j['partials']
                a.partials, a.nblk, a.ncol_total, a.col0, a.N, a.count = _p(pt), j['nblk'], pt.shape[1], j['col0'], j['N'], float(j['count'])
                a.gamma, a.mean, a.rstd = _p(j['gamma']), _p(j['mean']), _p(j['rstd'])
                a.dgamma, a.dbeta, a.ka, a.kb, a.kc = _p(j['dgamma']), _p(j['dbeta']), _p(j['ka']), _p(j['kb']), _p(j['kc'])
                a.accumulate = int(bool(j.get('accumulate', False)))
            self.launches += 1
            _check(self.lib.gast_bn_bwd_finalize_multi(arr, len(chunk), _stream()), 'gast_bn_bwd_finalize_multi')

    @staticmethod
    def _fill_bwd_fin(a, j):
        pt = j['partials']
        a.partials, a.nblk, a.ncol_total, a.col0, a.N, a.count = _p(pt), j['nblk'], pt.shape[1], j['col0'], j['N'], float(j['count'])
        a.gamma, a.mean, a.rstd = _p(j['gamma']), _p(j['mean']), _p(j['rstd'])
        a.dgamma, a.dbeta = _p(j['dgamma']), _p(j['dbeta'])
        a.ka, a.kb, a.kc = _p(j.get('ka')), _p(j.get('kb')), _p(j.get('kc'))
        a.accumulate = int(bool(j.get('accumulate', False)))

    def bn_bwd_fused_multi(self, jobs):
        """jobs: dicts of bn_bwd_finalize() arguments (without ka/kb/kc) + dz, X, rows: finalize and in-place apply in one launch per
        BN_MAX_BATCH jobs -- for short tensors."""
        for i0 in range(0, len(jobs), self.BN_MAX_BATCH):
            chunk = jobs[i0:i0 + self.BN_MAX_BATCH]
            arr = (_BnBwdJob * len(chunk))()
            for a, j in zip(arr, chunk):
                self._fill_bwd_fin(a.f, j)
                a.dz, a.lddz, a.X, a.ldx, a.rows = _p(j['dz']), _ld(j['dz']), _p(j['X']), _ld(j['X']), int(j['rows'])
            self.launches += 1
            _check(self.lib.gast_bn_bwd_fused_multi(_dt(chunk[0]['dz']), arr, len(chunk), _stream()), 'gast_bn_bwd_fused_multi')

    BN_EVAL_MAX_BATCH = 32

    def bn_eval_multi(self, jobs, eps):
        """jobs: (gamma, beta, running_mean, running_var, scale, shift, centered) tuples: every eval-mode BatchNorm of the model in one
        launch."""
        for i0 in range(0, len(jobs), self.BN_EVAL_MAX_BATCH):
            chunk = jobs[i0:i0 + self.BN_EVAL_MAX_BATCH]
            arr = (_BnEvalJob * len(chunk))()
            for a, (gamma, beta, rm, rv, scale, shift, centered) in zip(arr, chunk):
                a.gamma, a.beta, a.running_mean, a.running_var, a.N = _p(gamma), _p(beta), _p(rm), _p(rv), gamma.numel()
                a.scale, a.shift, a.centered = _p(scale), _p(shift), int(bool(centered))
            self.launches += 1
            _check(self.lib.gast_bn_eval_multi(arr, len(chunk), eps, _stream()), 'gast_bn_eval_multi')

    def bn_eval(self, gamma, beta, rm, rv, eps, N, scale, shift, centered=False):
        self.launches += 1
        _check(self.lib.gast_bn_eval(_p(gamma), _p(beta), _p(rm), _p(rv), eps, N, _p(scale), _p(shift), int(bool(centered)),
                                     _stream()), 'gast_bn_eval')

    def bn_bwd_finalize(self, partials, nblk, col0, N, count, gamma, mean, rstd, dgamma, dbeta, ka, kb, kc):
        self.launches += 1
        _check(self.lib.gast_bn_bwd_finalize(_p(partials), nblk, partials.shape[1], col0, N, float(count), _p(gamma), _p(mean),
                                             _p(rstd), _p(dgamma), _p(dbeta), _p(ka), _p(kb), _p(kc), _stream()),
               'gast_bn_bwd_finalize')

    def bn_bwd_apply(self, dz, X, rows, N, ka, kb, kc):
        self.launches += 1
        _check(self.lib.gast_bn_bwd_apply(_dt(dz), _p(dz), _ld(dz), _p(X), _ld(X), rows, N, _p(ka), _p(kb), _p(kc), _stream()), 'gast_bn_bwd_apply')

    def bn_bwd_apply_frames(self, dz, X, rows, N, ka, kb, kc, T_total, J, frames):
        """bn_bwd_apply for rows = B * T_total * J of which only the frames in the bit mask `frames` carry a gradient: the rest of dz is
        taken as zero without being read (it may be uninitialised)"""
        self.launches += 1
        _check(self.lib.gast_bn_bwd_apply_frames(_dt(dz), _p(dz), _ld(dz), _p(X), _ld(X), rows, N, _p(ka), _p(kb), _p(kc), int(T_total), int(J),
                                                 int(frames), _stream()), 'gast_bn_bwd_apply_frames')

    def bnrelu_apply(self, X, rows, N, scale, shift, Y, use_drop=False, salt=0, drop=None):
        """Y = drop(relu(scale*X + shift)); the dropout stream `salt` is indexed by the element offset in X."""
        self.launches += 1
        _check(self.lib.gast_bnrelu_apply(_dt(X), _p(X), _ld(X), rows, N, _p(scale), _p(shift), _p(Y), _ld(Y), int(bool(use_drop)),
                                               int(salt), _drop(drop), _stream()), 'gast_bnrelu_apply')

    def rowwise_blocks(self, rows, N):
        return self.lib.gast_rowwise_blocks(int(rows), int(N))

    def bnrelu_bwd_mask(self, dY, X, rows, N, scale, shift, use_drop, salt, drop, dz, partials):
        self.launches += 1
        _check(self.lib.gast_bnrelu_bwd_mask(_dt(X), _p(dY), _ld(dY), _p(X), _ld(X), rows, N, _p(scale), _p(shift),
                                             int(bool(use_drop)), int(salt), _drop(drop), _p(dz), _ld(dz), _p(partials), _stream()),
               'gast_bnrelu_bwd_mask')

    def shrink_fwd(self, O, rows, K, scale, shift, W, pred):
        """pred[r, :] = relu(scale*O[r, :] + shift) . W^T  (W [D][K] of O's dtype, pred fp32 [rows][D]): the shrink layer as ONE row-wise launch"""
        W = W if torch.is_tensor(W) else W.t          # (an X3Weight / F8Weight wrapper: the plain operand)
        self.launches += 1
        _check(self.lib.gast_shrink_fwd(_dt(O), _p(O), _ld(O), rows, K, _p(scale), _p(shift), _p(W), _ld(W), W.shape[0], _p(pred), _ld(pred),
                                        _stream()), 'gast_shrink_fwd')

    def shrink_bwd_blocks(self, rows):
        return self.lib.gast_shrink_bwd_blocks(int(rows))

    def shrink_bwd(self, dp, W, O, rows, K, scale, shift, dO, partials):
        """dO = [scale*O + shift > 0] * (dp[:, :D] . W) + the BatchNorm-backward column sums per row block (partials [shrink_bwd_blocks][K][2])"""
        W = W if torch.is_tensor(W) else W.t
        self.launches += 1
        _check(self.lib.gast_shrink_bwd(_dt(O), _p(dp), _ld(dp), _p(W), _ld(W), W.shape[0], _p(O), _ld(O), _p(scale), _p(shift), rows, K,
                                        _p(dO), _ld(dO), _p(partials), _stream()), 'gast_shrink_bwd')

    def residual_fwd(self, O, omap, scO, shO, T2, sc2, sh2, use_drop, salt, drop, B, Tn, J, N, Xn):
        self.launches += 1
        _check(self.lib.gast_residual_fwd(_dt(O), _p(O), _ld(O), _rm(omap), _p(scO), _p(shO), _p(T2), _ld(T2), _p(sc2), _p(sh2),
                                               int(bool(use_drop)), int(salt), _drop(drop), B, Tn, J, N, _p(Xn), _ld(Xn), _stream()), 'gast_residual_fwd')

    def colsum(self, X, rows, N, out, zero_first=True):
        self.launches += 1
        _check(self.lib.gast_colsum(_dt(X), _p(X), _ld(X), rows, N, _p(out), int(bool(zero_first)), _stream()), 'gast_colsum')

    # -- input side
    def input_stats_blocks(self, rows):
        return self.lib.gast_input_stats_blocks(int(rows))

    def input_stats(self, x, rows, F_in, partials):
        self.launches += 1
        _check(self.lib.gast_input_stats(_p(x), rows, F_in, _p(partials), None, _stream()), 'gast_input_stats')

    def expand_fwd(self, x, B, T_in, J, F_in, k0, t_stride, W, sc0, sh0, C_, E, partials, center=None):
        self.launches += 1
        _check(self.lib.gast_expand_fwd(_dt(E), _p(x), B, T_in, J, F_in, k0, t_stride, _p(W), _p(sc0), _p(sh0), C_, _p(E), _ld(E),
                                             _p(partials), _p(center), _stream()), 'gast_expand_fwd')

    def expand_bwd(self, dE, x, B, T_in, J, F_in, k0, t_stride, mean0, rstd0, C_, W, gamma0, beta0, dW, dgamma0, dbeta0, accumulate=False,
                   bn=None):
        """dW written (or += with accumulate); dgamma0 / dbeta0 always accumulated (zero-filled by the caller).
        bn = (Epre, ka, kb, kc): dE is the gradient BEFORE the backward of expand_bn; the kernel applies ka*dE + kb*Epre + kc on load."""
        self.launches += 2
        T_out = (T_in - k0) // t_stride + 1
        n = self.lib.gast_expand_bwd_ws_floats(B * T_out * J, C_, F_in, k0)
        ws = torch.empty(n, dtype=torch.float32, device=dE.device)
        Ep, ka, kb, kc = bn if bn is not None else (None, None, None, None)
        _check(self.lib.gast_expand_bwd_bn(_dt(dE), _p(dE), _ld(dE), _p(Ep), _ld(Ep) if Ep is not None else 0, _p(ka), _p(kb), _p(kc),
                                           _p(x), B, T_in, J, F_in, k0, t_stride, _p(mean0), _p(rstd0), C_,
                                           _p(W), _p(gamma0), _p(beta0), _p(dW), _p(dgamma0), _p(dbeta0), _p(ws), int(bool(accumulate)), _stream()),
               'gast_expand_bwd')

    # -- parameter packing / gradient unpacking (gast_hip/packer.py job lists -> device tables, one launch per list)
    @staticmethod
    def _word(ref, itemsize):
        from gast_hip.packer import BASE_ABS
        if ref.base == BASE_ABS:
            return ((ref.tensor.data_ptr() + ref.off * ref.tensor.element_size()) << 4) | 0
        return ((ref.off * itemsize) << 4) | ref.base

    def _copy_table(self, jobs, dev, w_itemsize, accumulate):
        """jobs: (src Ref, dst Ref, R, S[, fp32dst]) -> (int64 job table, int32 tile table, ntiles)"""
        from gast_hip.packer import BASE_W
        words, tiles = [], []
        for ji, job in enumerate(jobs):
            src, dst, R, S = job[0], job[1], job[2], job[3]
            src_size = w_itemsize if src.base == BASE_W else 4
            dst_size = w_itemsize if dst.base == BASE_W else 4
            flags = (1 if src_size == 2 else 0) | (2 if dst_size == 2 else 0) | (4 if accumulate else 0)
            words += [self._word(src, src_size), self._word(dst, dst_size), R, S, src.rs, src.cs, dst.rs, dst.cs, flags, 0]
            for tr in range((R + 31) // 32):
                for tc in range((S + 31) // 32):
                    tiles += [ji, tr, tc]
        jt = torch.tensor(words, dtype=torch.int64).to(dev)
        tt = torch.tensor(tiles, dtype=torch.int32).to(dev)
        return jt, tt, len(tiles) // 3

    def _tables(self, packer, st, dev):
        tb = st.get('tables')
        if tb is None:
            wsize = st['Wb'].element_size()
            tb = {'pack': self._copy_table(packer.copy_jobs, dev, wsize, False)}
            fw = []
            for j in packer.fold_jobs:
                wt = j['w']
                fw += [(j['W'].data_ptr() << 4), ((wt.data_ptr() + j['woff'] * 4) << 4), (j['b'].data_ptr() << 4), j['Ci'], j['C'],
                       self._word(j['row'], wsize), j['row'].cs, self._word(j['col'], wsize), j['col'].cs, self._word(j['bias'], 4),
                       1 if wsize == 2 else 0, 0]
            tb['fold'] = (torch.tensor(fw, dtype=torch.int64).to(dev), len(packer.fold_jobs))
            st['tables'] = tb
        return tb

    def _packx_tables(self, packer, st, dev):
        """gast_pack_all tables (GAST_F32X3 states): the copy jobs with the pre-split image of the operand each one lands in, the
        fold jobs with the images of their two destinations -- one launch packs operands AND images."""
        from gast_hip.packer import BASE_W
        wsize = st['Wb'].element_size()
        regions = sorted((off, r, c, n) for n, (off, r, c) in packer.W.regions.items())
        noimg = st.get('Xb') is None          # (a state without images: every image word stays 0 = "none")
        h16img = bool(st.get('h16img'))       # (16-bit storage: layout images, kind 2, 32 K positions per row; some operands have none)

        def region_of(off):
            for o, r, c, n in regions:
                if o <= off < o + r * c:
                    return o, c, n
            raise RuntimeError('gast_hip: packed destination outside every operand region')
        words, tiles = [], []
        for ji, job in enumerate(packer.copy_jobs):
            src, dst, R, S = job[0], job[1], job[2], job[3]
            src_size = wsize if src.base == BASE_W else 4
            dst_size = wsize if dst.base == BASE_W else 4
            flags = (1 if src_size == 2 else 0) | (2 if dst_size == 2 else 0)
            ext = [0, 0, 0, 0, 0, 0]
            if dst.base == BASE_W and not noimg and (not h16img or packer._h16_ok(region_of(dst.off)[2])):
                roff, K, name = region_of(dst.off)
                if dst.rs != 1 and dst.cs != 1:
                    raise RuntimeError('gast_hip: an operand destination must be K-contiguous or a transposed twin')
                img = packer._image(st, name)
                ext = [(img.data_ptr() << 4) | 0, img.stride(0), dst.off - roff, K, 2 if h16img else int(packer._f16(st, name)), 0]
            words += [self._word(src, src_size), self._word(dst, dst_size), R, S, src.rs, src.cs, dst.rs, dst.cs, flags, 0] + ext
            for tr in range((R + 31) // 32):
                for tc in range((S + 31) // 32):
                    tiles += [ji, tr, tc]
        fw = []
        for j in packer.fold_jobs:
            wt = j['w']
            if noimg:
                fw += [(j['W'].data_ptr() << 4), ((wt.data_ptr() + j['woff'] * 4) << 4), (j['b'].data_ptr() << 4), j['Ci'], j['C'],
                       self._word(j['row'], wsize), j['row'].cs, self._word(j['col'], wsize), j['col'].cs, self._word(j['bias'], 4),
                       1 if wsize == 2 else 0, 0, 0, 0, 0, 0, 0, 0]
                continue
            roff_r, K_r, name_r = region_of(j['row'].off)
            roff_c, K_c, name_c = region_of(j['col'].off)
            row_r, k0_r = divmod(j['row'].off - roff_r, K_r)          # the row destination: operand row, first K position (0)
            row_c, k_c = divmod(j['col'].off - roff_c, K_c)           # the column destination: first operand row (0), K position
            if k0_r != 0 or row_c != 0 or j['row'].cs != 1 or j['col'].cs != K_c:
                raise RuntimeError('gast_hip: unexpected fold destination layout')
            if h16img:
                ir = ic = ld_r = 0
                if packer._h16_ok(name_r):
                    img_r = packer._image(st, name_r)
                    ir, ld_r = img_r.data_ptr() + 2 * (row_r * 32), img_r.stride(0)
                if packer._h16_ok(name_c):
                    img_c = packer._image(st, name_c)
                    ic = img_c.data_ptr() + 2 * ((k_c >> 5) * img_c.stride(0) + (k_c & 31))
                fw += [(j['W'].data_ptr() << 4), ((wt.data_ptr() + j['woff'] * 4) << 4), (j['b'].data_ptr() << 4), j['Ci'], j['C'],
                       self._word(j['row'], wsize), j['row'].cs, self._word(j['col'], wsize), j['col'].cs, self._word(j['bias'], 4),
                       1 if wsize == 2 else 0, 0, (ir << 4), ld_r, (ic << 4), 0, 0, 1]
                continue
            img_r, img_c = packer._image(st, name_r), packer._image(st, name_c)
            ir = img_r.data_ptr() + 2 * (row_r * 32)
            ic = img_c.data_ptr() + 2 * ((k_c >> 4) * img_c.stride(0) + (k_c & 15))
            fw += [(j['W'].data_ptr() << 4), ((wt.data_ptr() + j['woff'] * 4) << 4), (j['b'].data_ptr() << 4), j['Ci'], j['C'],
                   self._word(j['row'], wsize), j['row'].cs, self._word(j['col'], wsize), j['col'].cs, self._word(j['bias'], 4),
                   1 if wsize == 2 else 0, 0,
                   (ir << 4), img_r.stride(0), (ic << 4), 0, int(packer._f16(st, name_r)) | (int(packer._f16(st, name_c)) << 1), 0]
        return (torch.tensor(words, dtype=torch.int64).to(dev), torch.tensor(tiles, dtype=torch.int32).to(dev), len(tiles) // 3,
                torch.tensor(fw, dtype=torch.int64).to(dev) if fw else None, len(packer.fold_jobs))

    def _unpack_tables(self, packer, st, dev, accumulate, bucket=None):
        tb = self._tables(packer, st, dev)
        key = 'unpack%d' % int(accumulate) + ('' if bucket is None else ':%d' % bucket)
        if key not in tb:
            from gast_hip.packer import BASE_G
            cjobs = packer.unpack_jobs if bucket is None else packer.unpack_by_bucket[bucket]
            ujobs = packer.unfold_jobs if bucket is None else packer.unfold_by_bucket[bucket]
            cp = self._copy_table(cjobs, dev, 4, accumulate) if cjobs else (None, None, 0)
            uw = []
            for j in ujobs:
                gW = packer.goff[packer.index[id(j['W'])]] * 4
                gw = (packer.goff[packer.index[id(j['w'])]] + j['woff']) * 4
                gb = packer.goff[packer.index[id(j['b'])]] * 4
                uw += [self._word(j['dv'], 4), self._word(j['da'], 4), (j['W'].data_ptr() << 4), ((j['w'].data_ptr() + j['woff'] * 4) << 4),
                       (j['b'].data_ptr() << 4), (gW << 4) | BASE_G, (gw << 4) | BASE_G, (gb << 4) | BASE_G, j['Ci'], j['C'],
                       int(accumulate), 0]
            tb[key] = (cp, torch.tensor(uw, dtype=torch.int64).to(dev) if uw else None, len(ujobs),
                       max((j['Ci'] for j in ujobs), default=0))
        return tb[key]

    def _bases(self, **kw):
        from gast_hip.packer import BASE_W, BASE_F, BASE_S, BASE_G
        arr = (C.c_int64 * 8)()
        for name, idx in (('W', BASE_W), ('F', BASE_F), ('S', BASE_S), ('G', BASE_G)):
            t = kw.get(name)
            arr[idx] = t.data_ptr() if t is not None else 0
        return arr

    # -- training-step tail
    def mpjpe(self, pred, target, loss, dirs):
        rows, D = pred.numel() // pred.shape[-1], pred.shape[-1]
        self.launches += 1
        _check(self.lib.gast_mpjpe(_p(pred), _p(target), rows, D, _p(loss), _p(dirs), _stream()), 'gast_mpjpe')

    NONFINITE_FLAGS = 256

    def nonfinite_scan(self, g, flags):
        """flags (NONFINITE_FLAGS int32): one verdict per block of the scan -- 1 when its slice of the flat fp32 buffer g holds an inf / NaN"""
        if flags.numel() < self.NONFINITE_FLAGS or flags.dtype != torch.int32 or g.dtype != torch.float32:
            raise RuntimeError('gast_hip: nonfinite_scan needs %d int32 flag words and an fp32 buffer' % self.NONFINITE_FLAGS)
        self.launches += 1
        _check(self.lib.gast_nonfinite_scan(_p(g), g.numel(), _p(flags), _stream()), 'gast_nonfinite_scan')

    def adam_step(self, p, g, m, v, vmax, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, skip=None, skipped=None):
        """skip: optional flag words of nonfinite_scan; any set = leave parameters, moments and the step counter alone and count the
        step in `skipped` (device int64)"""
        self.launches += 2
        _check(self.lib.gast_adam_step_guarded(_p(p), _p(g), _p(m), _p(v), _p(vmax), p.numel(), _p(step), lr, beta1, beta2, eps,
                                               weight_decay, grad_scale, _p(skip), _p(skipped), _stream()), 'gast_adam_step')

    def chunk_gather(self, poses2d, poses3d, cams, seq_off, pairs, first_pair, B, chunk, pad, causal_shift, perm2d, perm3d, out2d, out3d,
                     outcam):
        self.launches += 1 + (poses3d is not None) + (cams is not None)
        J2, F2 = poses2d.shape[-2], poses2d.shape[-1]
        J3, F3 = (poses3d.shape[-2], poses3d.shape[-1]) if poses3d is not None else (0, 0)
        ncam = cams.shape[-1] if cams is not None else 0
        _check(self.lib.gast_chunk_gather(_p(poses2d), _p(poses3d), _p(cams), _p(seq_off), _p(pairs), int(first_pair), int(B), int(chunk),
                                          int(pad), int(causal_shift), J2, F2, J3, F3, ncam, _p(perm2d), _p(perm3d), _p(out2d), _p(out3d),
                                          _p(outcam), _stream()), 'gast_chunk_gather')

    STREAM_SHIFT_MAX = 8

    def stream_shift_multi(self, jobs):
        """jobs: (buf (B, Tb, X) fp32 contiguous, newest (B, X) fp32 row-contiguous): every frame window advances by one frame in
        place and takes `newest` as its last frame -- one launch for all of them (gast_stream_shift_multi)."""
        for i0 in range(0, len(jobs), self.STREAM_SHIFT_MAX):
            chunk = jobs[i0:i0 + self.STREAM_SHIFT_MAX]
            arr = (_StreamShiftJob * len(chunk))()
            for a, (buf, newest) in zip(arr, chunk):
                if buf.dtype != torch.float32 or newest.dtype != torch.float32 or not buf.is_contiguous() or newest.stride(-1) != 1:
                    raise RuntimeError('gast_hip: stream_shift_multi needs fp32 tensors, a contiguous window and unit-stride rows')
                a.buf, a.newest = _p(buf), _p(newest)
                a.B, a.Tb, a.X, a.ldnew = int(buf.shape[0]), int(buf.shape[1]), int(buf.shape[2]), int(newest.stride(0))
            self.launches += 1
            _check(self.lib.gast_stream_shift_multi(arr, len(chunk), _stream()), 'gast_stream_shift_multi')

    def prep(self, zero, seed=None, pad=None):
        """Pass prologue (gast_prep): zero-fill the tensors of `zero` (contiguous device tensors), seed = (counter, per-pass copy):
        *copy = ++*counter (int32 / uint32 one-element tensors), pad = (src, dst, rows, cols_src, cols_dst[, scale]): contiguous, fp32 source, fp32 or 16-bit destination.  One
        launch per PREP_MAX_ZERO regions; a region that is not 16-byte granular is zeroed by torch instead."""
        jobs = []
        for t in zero:
            if not t.is_contiguous():
                raise RuntimeError('gast_hip: prep needs contiguous tensors')
            nbytes = t.numel() * t.element_size()
            if t.data_ptr() % 16 or nbytes % 16:
                t.zero_()
                self.launches += 1
            elif nbytes:
                jobs.append((_p(t), nbytes))
        first = True
        while first or jobs:
            a = _PrepArgs()
            chunk, jobs = jobs[:PREP_MAX_ZERO], jobs[PREP_MAX_ZERO:]
            for i, (ptr, nb) in enumerate(chunk):
                a.zero[i].ptr, a.zero[i].bytes = ptr, nb
            a.nzero = len(chunk)
            if first and seed is not None:
                a.seed_ctr, a.seed_out = _p(seed[0]), _p(seed[1])
            if first and pad is not None:
                src, dst, rows, cs, cd = pad[:5]
                if src.dtype != torch.float32 or dst.dtype not in (torch.float32, h16_dtype()) or not src.is_contiguous() or not dst.is_contiguous():
                    raise RuntimeError('gast_hip: prep pads a contiguous fp32 tensor into fp32 or the 16-bit storage type')
                a.pad_src, a.pad_dst, a.pad_rows, a.pad_cols_src, a.pad_cols_dst = _p(src), _p(dst), int(rows), int(cs), int(cd)
                a.pad_scale = float(pad[5]) if len(pad) > 5 else 1.0
                a.pad_dst_h16 = int(dst.dtype != torch.float32)
            if a.nzero or a.seed_ctr or a.pad_rows:
                self.launches += 1
                _check(self.lib.gast_prep(C.byref(a), _stream()), 'gast_prep')
            first = False

    def null_launch(self):
        _check(self.lib.gast_null_launch(_stream()), 'gast_null_launch')

    def run_pack(self, packer, st):
        dev = st['Wb'].device
        tb = self._tables(packer, st, dev)
        bases = self._bases(W=st['Wb'], F=st['Fb'])
        fused_ok = st.get('Xb') is not None or st.get('F8s') is None
        if fused_ok and os.environ.get('GAST_PACK_FUSED', '1') not in ('0', ''):
            # GAST_F32X3: operands, folds and their pre-split images in ONE launch (gast_pack_all; GAST_PACK_FUSED=0: the three
            # launches of rounds 2-3).  Round 6: also the states WITHOUT images (fp32): copy tiles and fold blocks in one grid, image
            # words zero -- and the 16-bit states, whose LAYOUT images (kind 2) are written by the same tiles
            px = tb.get('packx')
            if px is None:
                px = tb['packx'] = self._packx_tables(packer, st, dev)
            jt, tt, nt, ft, nf = px
            self.launches += 1
            _check(self.lib.gast_pack_all(_p(jt), _p(tt), nt, _p(ft), nf, max((j['C'] for j in packer.fold_jobs), default=1),
                                          C.cast(bases, C.c_void_p), _stream()), 'gast_pack_all')
            return
        jt, tt, nt = tb['pack']
        self.launches += 2
        _check(self.lib.gast_strided_copy(_p(jt), _p(tt), nt, C.cast(bases, C.c_void_p), _stream()), 'gast_strided_copy')
        ft, nf = tb['fold']
        _check(self.lib.gast_fold(_p(ft), nf, max(j['C'] for j in packer.fold_jobs), C.cast(bases, C.c_void_p), _stream()), 'gast_fold')
        if st.get('F8s') is not None:         # fp8 mode: per-tensor power-of-two scales of the forward weight operands, one launch
            arr = tb.get('f8scale')
            if arr is None:
                jobs = packer.f8_jobs(st)
                arr = (_F8ScaleJob * len(jobs))()
                for a, (wv, sv) in zip(arr, jobs):
                    a.W, a.R, a.K, a.ldw, a.out = _p(wv), wv.shape[0], wv.shape[1], _ld(wv), _p(sv)
                tb['f8scale'] = arr
            self.launches += 1
            _check(self.lib.gast_f8_scale_multi(arr, len(arr), _stream()), 'gast_f8_scale_multi')
        if st.get('Xb') is not None:          # GAST_F32X3: the pre-split images of all packed operands, one launch (16-bit modes: layout images)
            arr = tb.get('x3img')
            if arr is None:
                jobs = packer.image_jobs(st)
                arr = (_X3ImageJob * len(jobs))()
                for a, (wv, iv, f16) in zip(arr, jobs):
                    a.W, a.R, a.K, a.ldw, a.img, a.ldimg = _p(wv), wv.shape[0], wv.shape[1], _ld(wv), _p(iv), iv.stride(0)
                    a.f16 = int(f16)         # (image kind: 0 / 1 / 2, packer.image_jobs)
                tb['x3img'] = arr
            self.launches += 1
            _check(self.lib.gast_x3_image_multi(arr, len(arr), _stream()), 'gast_x3_image_multi')

    def run_unpack(self, packer, st, Sb, G, accumulate, bucket=None):
        """packed gradient scratch -> parameter-shaped gradients in the flat buffer; bucket = i: only the jobs whose destination
        lies in packer.bucket_ranges[i] (Packer.set_buckets)."""
        dev = G.device
        (jt, tt, nt), ut, nu, maxci = self._unpack_tables(packer, st, dev, accumulate, bucket)
        bases = self._bases(S=Sb, G=G)
        if nt:
            self.launches += 1
            _check(self.lib.gast_strided_copy(_p(jt), _p(tt), nt, C.cast(bases, C.c_void_p), _stream()), 'gast_strided_copy')
        if nu:
            self.launches += 1
            _check(self.lib.gast_unfold(_p(ut), nu, maxci, C.cast(bases, C.c_void_p), _stream()), 'gast_unfold')
