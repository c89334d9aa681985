#!/usr/bin/env python3
"""Static screen of hipcc's gfx950 assembly for the inline-asm asynchronous-load hazard.

csrc/common.h issues 16-byte global loads from inline asm (gload16 / gload16s) and waits for them with a separate asm statement
(gload_wait_n).  The compiler does not know the destination registers are still in flight between the two, so it may legally
COPY or otherwise touch them there (a phi copy at a loop back edge, a spill, a re-materialisation): the copy then reads whatever
the register held before the load landed -- a rare, timing-dependent wrong value.  This script walks every function of a `.s`
file in program order, keeps the FIFO of outstanding VMEM operations (gfx9: loads and stores share vmcnt), and reports every
instruction that names a register of a still-outstanding inline-asm load before an s_waitcnt vmcnt(N) has retired it.

  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only csrc/gemm.hip -o /tmp/gemm.s && python scripts/asm_load_hazard.py /tmp/gemm.s
  ... asm_load_hazard.py --strict /tmp/gemm.s     (compiler-inserted waits do not count: see scan())
Second screen (round 6), `--sgpr`: the gfx9 data hazard "VALU writes an SGPR -> VMEM reads that SGPR" needs 5 wait states.  hipcc's hazard
recognizer inserts them for the instructions it generates, but it cannot see INSIDE an inline-asm statement: when the scalar base of
gload16s / glds16 reaches the statement through v_readfirstlane (a pointer the divergence analysis kept in vector registers), nothing
separates the VALU write from the asm's global_load.  The load then uses the OLD register content -- a wild address: the "memory access
fault that comes and goes with unrelated code changes" of rounds 5 and 6.  Reported: every inline-asm VMEM instruction whose saddr pair
was written by a VALU instruction fewer than 5 instruction slots (s_nop N counts N + 1) earlier in layout order.

Linear scan (blocks are walked in layout order; loads issued inside a loop are forgotten at its exit block), so a report is a lead to read, not a proof; no report on a loop whose
layout order is its execution order is strong evidence.  Exit code 1 when anything was reported.
"""
import re
import sys

VMEM = re.compile(r'^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|scratch_load|scratch_store|flat_load|flat_store|flat_atomic)')
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


STRICT = False


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    fn, in_asm, fifo, reports, in_loop = None, False, [], [], None
    for ln, line in enumerate(open(path, errors='replace'), 1):
        s = line.split(';')[0].rstrip() if not line.lstrip().startswith(';;#') else line.strip()
        if re.match(r'^[_A-Za-z][\w$.]*:\s*(;.*)?$', line) and not line.startswith('.L'):
            fn, fifo, in_loop = line.split(':')[0], [], None
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        s = s.strip()
        if line.startswith('.LBB'):
            # leaving a loop (the next block is not annotated as part of it): the loop's exit edge follows its last counted wait at run
            # time, although the block is laid out behind the loop's trailing loads -- forget the loads issued INSIDE that loop.  Loads
            # issued before it stay pending (round 5: a short table-building loop between the first loads of a kernel and their wait used
            # to wipe them, and the conversion that hipcc hoisted above that wait went unreported)
            m = re.search(r'Header=(\w+)', line)
            now_loop = m.group(1) if m else (line.split(':')[0].lstrip('.L') if 'Loop Header' in line else None)
            if in_loop is not None and now_loop != in_loop:
                fifo = [(d, l, t) for d, l, t in fifo if t != in_loop]
            in_loop = now_loop
            continue
        if not s or s.startswith('.') or s.endswith(':'):
            continue
        m = re.match(r's_waitcnt\b(.*)', s)
        if m:
            v = re.search(r'vmcnt\((\d+)\)', m.group(1))
            # --strict: only the waits the SOURCE wrote (gload_wait_n, inside ASMSTART / ASMEND) retire in-flight loads.  A wait the
            # compiler inserted for a load of its own may sit on a conditionally executed path (the lazy BatchNorm finalize in front of
            # gemm_big's tables): the linear walk would count it, the hardware may never execute it
            if v and (in_asm or not STRICT):
                n = int(v.group(1))
                fifo = fifo[len(fifo) - n:] if n else []
            continue
        if s.startswith('s_endpgm'):
            fifo = []
            continue
        if VMEM.match(s):
            ops = s.split(None, 1)[1] if ' ' in s else ''
            dst = regs(ops.split(',')[0]) if in_asm and s.startswith('global_load') and 'lds' not in s.split()[0] else set()
            used = regs(ops) - dst
            pending = set().union(*[d for d, _, _ in fifo]) if fifo else set()
            if used & pending:
                reports.append((fn, ln, s, sorted(used & pending)))
            fifo.append((dst, ln, in_loop))
            continue
        pending = set().union(*[d for d, _, _ in fifo]) if fifo else set()
        if pending:
            hit = regs(s) & pending
            if hit:
                reports.append((fn, ln, s, sorted(hit)))
    return reports


SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')


def sregs(tok):
    out = set()
    for m in SREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan_sgpr(path):
    """VALU write of an SGPR followed by an inline-asm VMEM read of it with fewer than 5 wait states in between"""
    fn, in_asm, reports = None, False, []
    recent = []          # (wait states elapsed since, sgprs written by a VALU instruction, line, text)
    for ln, line in enumerate(open(path, errors='replace'), 1):
        if re.match(r'^[_A-Za-z][\w$.]*:\s*(;.*)?$', line) and not line.startswith('.L'):
            fn, recent = line.split(':')[0], []
            continue
        st = line.strip()
        if st.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if st.startswith(';;#ASMEND'):
            in_asm = False
            continue
        s_ = line.split(';')[0].strip()
        if not s_ or s_.startswith('.') or s_.endswith(':'):
            continue
        op = s_.split()[0]
        if in_asm and VMEM.match(s_):
            ops = s_.split(None, 1)[1] if ' ' in s_ else ''
            used = sregs(ops)
            for age, wr, wl, wt in recent:
                if age < 5 and used & wr:
                    reports.append((fn, ln, s_, 'saddr s%s written %d wait state(s) earlier by: %s (line %d)' % (sorted(used & wr), age, wt, wl)))
        step = 1
        m = re.match(r's_nop\s+(\d+)', s_)
        if m:
            step = int(m.group(1)) + 1
        recent = [(a + step, w, l, t) for a, w, l, t in recent if a + step < 8]
        if op.startswith('v_'):
            dst = s_.split(None, 1)[1].split(',')[0] if ' ' in s_ else ''
            w = sregs(dst)
            if op.startswith('v_cmp') and not w and 'vcc' not in dst:
                w = set()
            if w:
                recent.append((0, w, ln, s_))
    return reports


if __name__ == '__main__':
    bad = 0
    if '--sgpr' in sys.argv:
        sys.argv.remove('--sgpr')
        for p in sys.argv[1:]:
            rep = scan_sgpr(p)
            for fn, ln, s, why in rep[:400]:
                print('%s:%d  %s\n    in %s   %s' % (p, ln, s, fn, why))
            print('%s: %d VALU->SGPR->VMEM hazard(s) inside inline asm' % (p, len(rep)))
            bad += len(rep)
        sys.exit(1 if bad else 0)
    if '--strict' in sys.argv:
        STRICT = True
        sys.argv.remove('--strict')
    for p in sys.argv[1:]:
        rep = scan(p)
        for fn, ln, s, hit in rep[:400]:
            print('%s:%d  %s\n    in %s   touches in-flight v%s' % (p, ln, s, fn, hit))
        print('%s: %d suspicious instruction(s)' % (p, len(rep)))
        bad += len(rep)
    sys.exit(1 if bad else 0)
