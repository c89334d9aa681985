"""Host-side contract of the drop-in boundary (no GPU): constructor surface, state_dict keys/shapes, namespace leak,
error behaviour, C-ABI symbols exported by libgast_hip.so."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, PKG
from tests_helpers import PARENTS


def adj(J):
    from oracle.gast_oracle import adj_from_parents
    return torch.from_numpy(adj_from_parents(PARENTS[J]))


def test_state_dict_contract_j17():
    from model.gast_net import SpatioTemporalModel, SpatioTemporalModelOptimized1f
    with open(os.path.join(GOLDEN, 'state_dict_contract_j17_a333_c128.json')) as f:
        ref = json.load(f)
    for cls in (SpatioTemporalModel, SpatioTemporalModelOptimized1f):
        m = cls(adj(17), 17, 2, 17, filter_widths=[3, 3, 3], channels=128)
        sd = m.state_dict()
        assert list(sd.keys()) == list(ref.keys())         # same names in the same registration order
        for k, v in sd.items():
            assert list(v.shape) == ref[k], k
        assert len(sd) == 228
        assert m.receptive_field() == 27


def test_param_counts_known_answers():
    from model.gast_net import SpatioTemporalModel
    with open(os.path.join(GOLDEN, 'index.json')) as f:
        counts = json.load(f)['_param_counts']
    for J in (17, 19, 15):
        m = SpatioTemporalModel(adj(J), J, 2, J, filter_widths=[3, 3, 3], channels=128)
        assert sum(p.numel() for p in m.parameters()) == counts['J%d_a333_c128' % J]


def test_namespace_and_errors():
    import model.gast_net as g
    for name in ('torch', 'nn', 'LocalGraph', 'MultiGlobalGraph', 'SingleGlobalGraph', 'SpatioTemporalModel',
                 'SpatioTemporalModelOptimized1f', 'GraphAttentionBlock', 'SpatioTemporalModelBase'):
        assert hasattr(g, name), name
    assert not hasattr(g, '__all__')
    with pytest.raises(AssertionError):
        g.SpatioTemporalModel(adj(17), 17, 2, 17, filter_widths=[3, 4, 3])
    with pytest.raises(KeyError):
        g.SpatioTemporalModel(torch.eye(14), 14, 2, 14, filter_widths=[3, 3])
    a = adj(17)
    a0 = a.clone()
    m = g.SpatioTemporalModel(a, 17, 2, 17, filter_widths=[3, 3, 3], causal=True, dropout=0.05, channels=32)
    assert torch.equal(a, a0)                                # constructor must not mutate adj
    assert m.receptive_field() == 27 and m.pad == [1, 3, 9] and m.causal_shift == [1, 3, 9]
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(2, 27, 17, 2))
    with pytest.raises(AssertionError):
        m(torch.zeros(2, 27, 16, 2))


@pytest.mark.skipif(not os.path.isdir('/root/reference/model'), reason='reference checkout not present')
def test_initialisation_matches_reference_seed_for_seed():
    """torch.manual_seed(s) + construction gives bit-identical initial weights (same initialisers, same RNG order)."""
    code = r'''
import sys, types, torch, hashlib
stub = types.ModuleType('torchsummary'); stub.summary = lambda *a, **k: None; sys.modules['torchsummary'] = stub
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[2])
from oracle.gast_oracle import adj_from_parents
from model.gast_net import SpatioTemporalModel, SpatioTemporalModelOptimized1f
adj = torch.from_numpy(adj_from_parents([-1,0,1,2,0,4,5,0,7,8,9,8,11,12,8,14,15]))
for cls in (SpatioTemporalModel, SpatioTemporalModelOptimized1f):
    torch.manual_seed(0)
    m = cls(adj, 17, 2, 17, filter_widths=[3,3,3], causal=False, dropout=0.05, channels=32)
    h = hashlib.sha256()
    for k, v in m.state_dict().items():
        h.update(k.encode()); h.update(v.numpy().tobytes())
    print(h.hexdigest())
'''
    outs = []
    for first in ('/root/reference', PKG):
        r = subprocess.run(['python', '-c', code, first, ROOT], capture_output=True, text=True, check=True)
        outs.append(r.stdout.strip().splitlines())
    assert outs[0] == outs[1] and len(outs[0]) == 2


def test_library_exports_every_declared_symbol():
    from gast_hip.binding import EXPORTED_SYMBOLS, LIB_PATH
    assert os.path.exists(LIB_PATH), 'run __graft_entry__.build() first'
    with open(os.path.join(ROOT, 'include', 'gast_hip.h')) as f:
        header = f.read()
    declared = sorted(set(re.findall(r'\b(gast_[a-z0-9_]+)\s*\(', header)))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(LIB_PATH)            # loads without a GPU; no compute call is made
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(EXPORTED_SYMBOLS) == declared
    lib.gast_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.gast_version() and b'bfloat16' in lib.gast_version()
    # the binary16 storage flavour (same sources, -DGAST_H16_F16: GAST_HIP_DTYPE=f16) has the same ABI
    from gast_hip.binding import LIB_PATH_F16
    assert os.path.exists(LIB_PATH_F16), 'run __graft_entry__.build() first'
    lib16 = ctypes.CDLL(LIB_PATH_F16)
    for sym in declared:
        assert hasattr(lib16, sym), sym
    lib16.gast_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib16.gast_version() and b'binary16' in lib16.gast_version()
    # size helpers are pure host functions: callable without a GPU
    assert lib.gast_gemm_row_blocks(54400) == 425


def test_pattern_tables_match_contract():
    from model.local_attention import pattern_table, skeleton_patterns
    from oracle import kernel_contract as kc
    from oracle.gast_oracle import local_graph_adjacencies, adj_from_parents
    for J in (15, 16, 17, 19):
        a = adj(J)
        sym, con = skeleton_patterns(a)
        s2, c2 = local_graph_adjacencies(adj_from_parents(PARENTS[J]))
        assert np.array_equal(sym.numpy() > 0, s2 > 0) and np.array_equal(con.numpy() > 0, c2 > 0)
        for ours, ref in ((sym, s2), (con, c2)):
            tab, nnz = pattern_table(ours)
            assert np.array_equal(tab.numpy(), kc.build_pattern(ref)) and nnz == int((ref > 0).sum())


def test_binding_argument_counts_match_header():
    """every ctypes signature has as many arguments as the C declaration (a short list silently drops the stream pointer)"""
    import re
    from gast_hip import binding
    lib = binding.load_library()
    h = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'gast_hip.h')).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    for m in re.finditer(r'\b(?:int|long|const char\*)\s+(gast_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;', h):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        fn = getattr(lib, name)
        assert fn.argtypes is not None, name
        assert len(fn.argtypes) == n, (name, len(fn.argtypes), n)


def test_segment_limit_is_the_same_everywhere():
    """The engine chunks K segments (dense=True taps) by MAX_SEG: it must equal the ABI's GAST_MAX_SEG and the binding's array size."""
    import re
    from gast_hip import engine, binding
    hdr = open(os.path.join(ROOT, 'include', 'gast_hip.h')).read()
    n = int(re.search(r'#define\s+GAST_MAX_SEG\s+(\d+)', hdr).group(1))
    assert engine.MAX_SEG == binding.MAX_SEG == n


def test_operand_images_follow_their_slices():
    """packer.X3Weight: a GEMM operand travels with its k-group-major image (include/gast_hip.h, gast_x3_image_multi).  A row slice keeps
    the rows behind it (a tile reads 256 image rows from its first row), a column slice aligned to the K groups -- 16 values for the
    split images of fp32 operands, 32 for the layout image of a 16-bit operand (round 5) -- moves the group index, any other column
    slice DROPS the image (the GEMM then takes the 128 x 128 kernel, which needs none).  Host logic only: no device."""
    from gast_hip.packer import X3Weight, x3_image_rows
    R, K = 40, 96
    for group, dt in ((16, torch.float32), (32, torch.float16)):
        w = torch.arange(R * K, dtype=torch.float32).reshape(R, K).to(dt)
        img = torch.zeros((K + group - 1) // group, x3_image_rows(R), 32, dtype=torch.bfloat16 if group == 16 else dt)
        x = X3Weight(w, img, False, group)
        a = x[8:24]                                            # rows: the operand is the slice, the image starts at row 8 and keeps its tail
        assert a.t.shape == (16, K) and a.img.shape == (img.shape[0], img.shape[1] - 8, 32) and a.img.data_ptr() == img[:, 8:].data_ptr()
        b = x[:, 2 * group:]                                   # aligned column slice: group index 2
        assert b.t.shape == (R, K - 2 * group) and b.img.shape[0] == img.shape[0] - 2 and b.img.data_ptr() == img[2:].data_ptr()
        c = x[:, group:2 * group][4:]                          # both, chained
        assert c.t.shape == (R - 4, group) and c.img.shape[0] == 1 and c.img.data_ptr() == img[1:2, 4:].data_ptr() and c.group == group
        assert x[:, 8:40].img is None and x[:, 8:40].t.shape == (R, 32)          # not on a group boundary: no image
        assert x[:, :group + 8].img is None                                       # ragged end inside the operand: no image
        assert x[:, 3 * group:].img is not None                                   # ... but the operand's own K tail is fine
        assert x[::2].img is None                                                 # strided rows: no image
    # (a 16-value boundary is NOT a boundary of the 32-value layout image)
    w16 = torch.zeros(R, K, dtype=torch.float16)
    assert X3Weight(w16, torch.zeros(3, x3_image_rows(R), 32, dtype=torch.float16), False, 32)[:, 16:48].img is None


def test_asm_load_hazard_scanner_modes(tmp_path):
    """scripts/asm_load_hazard.py on two synthetic listings.  (1) A register of an inline-asm load is read between the load and the wait
    the source wrote, behind a compiler-inserted wait that sits on a conditionally executed path: the default walk lets that wait retire the
    load (no report), --strict does not (one report) -- the shape of the round-5 miss.  (2) A load issued BEFORE a short loop stays pending
    when the walk leaves the loop; loads issued inside it are forgotten at its exit (their wait is the loop's own)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('asm_load_hazard', os.path.join(ROOT, 'scripts', 'asm_load_hazard.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    one = tmp_path / 'one.s'
    one.write_text('\n'.join([
        'kern_a:', '\t;;#ASMSTART', '\tglobal_load_dwordx4 v[6:9], v3, s[4:5]', '\t;;#ASMEND',
        '\ts_cbranch_execz .LBB0_2',
        '\tglobal_load_dword v20, v[12:13], off', '\ts_waitcnt vmcnt(0)',          # the compiler's own load and wait, conditional
        '.LBB0_2:',
        '\tv_cndmask_b32_e64 v6, v6, 0, s[2:3]',                                     # <- reads v6 before the source's wait
        '\t;;#ASMSTART', '\ts_waitcnt vmcnt(0)', '\t;;#ASMEND',
        '\tv_add_f32_e32 v1, v6, v7', '\ts_endpgm', '']))
    mod.STRICT = False
    assert mod.scan(str(one)) == []
    mod.STRICT = True
    rep = mod.scan(str(one))
    assert len(rep) == 1 and rep[0][3] == [6] and 'v_cndmask' in rep[0][2]
    two = tmp_path / 'two.s'
    two.write_text('\n'.join([
        'kern_b:', '\t;;#ASMSTART', '\tglobal_load_dwordx4 v[6:9], v3, s[4:5]', '\t;;#ASMEND',
        '.LBB1_1:                                ; =>This Loop Header: Depth=1',
        '\t;;#ASMSTART', '\tglobal_load_dwordx4 v[30:33], v4, s[4:5]', '\t;;#ASMEND',
        '\ts_cbranch_scc1 .LBB1_1',
        '.LBB1_2:',
        '\tv_mov_b32_e32 v40, v30',                                                  # issued inside the loop: forgotten at its exit
        '\tv_mov_b32_e32 v41, v8',                                                   # <- issued before the loop: still pending
        '\t;;#ASMSTART', '\ts_waitcnt vmcnt(0)', '\t;;#ASMEND', '\ts_endpgm', '']))
    rep = mod.scan(str(two))
    assert len(rep) == 1 and rep[0][3] == [8]
    mod.STRICT = False
    # (3) round 6, --sgpr: a scalar base written by v_readfirstlane right in front of an inline-asm load is reported; five wait states
    # (other instructions, or s_nop 4 inside the statement) clear it; a SALU-written base is no hazard
    three = tmp_path / 'three.s'
    three.write_text('\n'.join([
        'kern_c:', '\tv_readfirstlane_b32 s11, v117', '\tv_add_u32_e32 v1, v2, v3', '\tv_readfirstlane_b32 s10, v116',
        '\t;;#ASMSTART', '\tglobal_load_dwordx4 v[38:41], v11, s[10:11]', '\t;;#ASMEND',                      # <- 0 and 2 wait states
        '\tv_readfirstlane_b32 s12, v20', '\tv_readfirstlane_b32 s13, v21',
        '\t;;#ASMSTART', '\ts_nop 4', '\tglobal_load_dwordx4 v[42:45], v11, s[12:13]', '\t;;#ASMEND',           # wait states inside the statement
        '\ts_add_u32 s14, s14, 128', '\ts_addc_u32 s15, s15, 0',
        '\t;;#ASMSTART', '\tglobal_load_dwordx4 v[46:49], v11, s[14:15]', '\t;;#ASMEND', '\ts_endpgm', '']))     # SALU-written base
    rep = mod.scan_sgpr(str(three))
    assert len(rep) == 2 and all('s[10:11]' in r[2] for r in rep), rep


def test_compiled_kernels_pass_the_strict_load_hazard_screen(tmp_path):
    """hipcc's gfx950 code for the kernels that issue asynchronous loads from inline asm (csrc/common.h: gload16 / gload_wait_n), screened
    with scripts/asm_load_hazard.py --strict: no instruction may name a register of an in-flight load before a wait the SOURCE wrote.
    gemm.hip and wgrad.hip compile in about a minute side by side; gemm_big.hip (two minutes by itself) joins with GAST_TEST_ASM_SCAN_ALL=1
    (it is screened by hand whenever it changes: DESIGN.md section 4, round 5)."""
    import shutil
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    csrc = os.path.join(PKG, 'csrc')
    files = ['gemm', 'wgrad', 'gemm_bj'] + (['gemm_big', 'wgrad_wide'] if os.environ.get('GAST_TEST_ASM_SCAN_ALL') else [])
    procs = []
    for f in files:
        out = str(tmp_path / (f + '.s'))
        procs.append((f, out, subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment', '-S', '--cuda-device-only',
                                                os.path.join(csrc, f + '.hip'), '-o', out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for f, out, p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0 and os.path.exists(out), '%s.hip did not compile: %s' % (f, err.decode()[-800:])
    for f, out, _ in procs:
        r = subprocess.run(['python', os.path.join(ROOT, 'scripts', 'asm_load_hazard.py'), '--strict', out], capture_output=True, text=True)
        assert r.returncode == 0 and '0 suspicious instruction(s)' in r.stdout, '%s.hip: %s' % (f, r.stdout[-1500:])
        # round 6: no VALU-written SGPR may feed an inline-asm VMEM instruction within 5 wait states (the cause of the memory faults that
        # came and went with unrelated code changes: hipcc's hazard recognizer does not look inside an asm statement)
        r = subprocess.run(['python', os.path.join(ROOT, 'scripts', 'asm_load_hazard.py'), '--sgpr', out], capture_output=True, text=True)
        assert r.returncode == 0 and '0 VALU->SGPR->VMEM hazard(s)' in r.stdout, '%s.hip: %s' % (f, r.stdout[-1500:])
