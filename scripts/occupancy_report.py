"""Register / LDS / scratch report of every kernel in csrc/ from the ISA metadata (no GPU needed):
    python scripts/occupancy_report.py [file.hip ...] [--filter substr]
Columns: VGPRs (-> waves per SIMD at 512 VGPRs per SIMD lane slot), spilled VGPRs / SGPRs, scratch bytes, static LDS.
Round 2b found three regressions only this table shows: a dead split-K exit that cost the BNRELU_BWD variants of gemm_big 50 VGPRs
(3 -> 2 blocks per CU), look-ahead registers that pushed the 32-channel-head attention backward past 168 VGPRs, and a kernel that
started to use scratch (~6 us more per dispatch)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gast-net-3dposeestimation_amd', 'csrc')
args = [a for a in sys.argv[1:] if not a.startswith('--')]
flt = None
if '--filter' in sys.argv:
    flt = sys.argv[sys.argv.index('--filter') + 1]
    args = [a for a in args if a != flt]
files = [os.path.abspath(a) for a in args] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
print('%-78s %5s %5s %6s %6s %8s %8s' % ('kernel', 'vgpr', 'w/SIMD', 'vspill', 'sspill', 'scratch', 'lds'))
for f in files:
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment', '-S', '--cuda-device-only', '-o', tmp.name, f],
                           capture_output=True, text=True, cwd=CSRC)
        if r.returncode:
            print('%s: %s' % (f, r.stderr.splitlines()[-1] if r.stderr else 'failed'))
            continue
        txt = open(tmp.name).read()
    for blk in txt.split('  - .agpr_count:')[1:]:
        def field(name, default='0'):
            m = re.search(r'\.%s:\s+(\S+)' % name, blk)
            return m.group(1) if m else default
        name = field('name', '?')
        try:
            name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        name = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if flt and flt not in name:
            continue
        v = int(field('vgpr_count'))
        waves = min(8, 512 // max(v, 1)) if v else 8
        print('%-78s %5d %5d %6s %6s %8s %8s' % (name[:78], v, waves, field('vgpr_spill_count'), field('sgpr_spill_count'),
                                               field('private_segment_fixed_size'), field('group_segment_fixed_size')))
