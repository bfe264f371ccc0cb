#!/usr/bin/env python3
"""Build-time ISA check for kernels that keep a ring of in-flight global loads in ordinary asm outputs (csrc/chain_dest.hip.h:
FARNN_RD_ISSUE_, "=&v" destinations, released later by a counted `s_waitcnt vmcnt(N)` with "+v" operands).

The compiler does not know those registers are being written asynchronously between the issue and the wait.  The failure mode
(documented in compact_tag.hip.h, where the ring is pinned for it): register allocation inserts a COPY of a ring register at a
control-flow join -- v_mov_b32 / v_accvgpr_* / a scratch spill -- while its load is still in flight; the copy reads the stale
value and the tags are silently wrong.  This script fails the build when that happens.

For every kernel of a device assembly file (hipcc -save-temps=obj: <unit>-hip-amdgcn-amd-amdhsa-gfx950.s):
  ring registers = the destination VGPRs of every `global_load_dword*` that sits INSIDE an inline-asm block (;;#ASMSTART .. ;;#ASMEND);
  a finding      = a compiler-generated instruction (outside every asm block) that moves or spills a ring register:
                   v_mov_b32 / v_mov_b64 / v_swap_b32 / v_accvgpr_write / v_accvgpr_read / scratch_* / buffer_store* with a ring
                   register as source or destination.
Arithmetic that consumes a ring register (v_fma, v_pk_fma, ...) is what the kernel does behind its waits and is not a finding.

    python scripts/check_ring_registers.py <file.s> [--kernels SUBSTRING] [--verbose]
Exit status 1 on a finding.  csrc/build.py runs it on every unit that asks for it (`// build-check: ring-registers <substring>`).
"""
import re
import sys

MOVES = ('v_mov_b32', 'v_mov_b64', 'v_swap_b32', 'v_accvgpr_write', 'v_accvgpr_read', 'v_accvgpr_mov', 'scratch_', 'buffer_store',
         'v_readlane', 'v_writelane', 'v_readfirstlane')
VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(path):
    """yields (name, [(line_no, text, in_asm)])"""
    name, body, in_asm = None, [], False
    with open(path) as f:
        for no, line in enumerate(f, 1):
            s = line.strip()
            m = re.match(r'^(_Z\w+):', line)
            if m and name is None:
                name, body, in_asm = m.group(1), [], False
                continue
            if name is None:
                continue
            if s.startswith('.end_amdhsa_kernel') or s.startswith('s_endpgm') and False:
                pass
            if s.startswith('.section') or s.startswith('.Lfunc_end'):
                yield name, body
                name = None
                continue
            if s.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if s.startswith(';;#ASMEND'):
                in_asm = False
                continue
            m2 = re.match(r'^(\.LBB\w+):', s)
            if m2:
                body.append((no, m2.group(1) + ':', False))
                continue
            if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
                continue
            body.append((no, s.split(';')[0].strip(), in_asm))
    if name is not None:
        yield name, body


WAIT = re.compile(r's_waitcnt\b.*?vmcnt\((\d+)\)')
VM_OPS = ('global_load', 'global_store', 'global_atomic', 'buffer_load', 'buffer_store', 'buffer_atomic', 'flat_load', 'flat_store',
          'scratch_load', 'scratch_store')


def blocks_of(body):
    """basic blocks: {label or index: (instructions, successors)}; a block ends at a branch, s_endpgm or in front of a label"""
    blocks, order, cur, name = {}, [], [], '^entry'
    def close(nxt):
        blocks[name] = [cur, nxt]
        order.append(name)
    for no, text, in_asm in body:
        if text.endswith(':') and text.startswith('.LBB'):
            lab = text[:-1]
            close([lab])                             # fall through into the label
            cur, name = [], lab
            continue
        cur.append((no, text, in_asm))
        if in_asm:
            continue
        if text.startswith('s_branch'):
            close([text.split()[1]])
            cur, name = [], '^after{}'.format(no)
        elif text.startswith('s_cbranch'):
            nxt = '^after{}'.format(no)
            close([text.split()[1], nxt])
            cur, name = [], nxt
        elif text.startswith('s_endpgm'):
            close([])
            cur, name = [], '^after{}'.format(no)
    close([])
    return blocks


def walk_block(insts, pending, bad, ring):
    """the vector-memory counter as the hardware keeps it: outstanding operations oldest first, each with the registers an
    asm-issued load will write; `s_waitcnt vmcnt(N)` = all but the N youngest have completed"""
    pending = list(pending)
    for no, text, in_asm in insts:
        m = WAIT.search(text)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            continue
        if text.startswith('s_waitcnt'):
            continue
        if text.startswith(VM_OPS):
            dst = frozenset(vregs(text.split(',')[0])) if (in_asm and '_load' in text.split()[0]) else frozenset()
            if not in_asm and text.startswith(('scratch_store', 'buffer_store')):   # a spill of an in-flight register
                hot = set().union(*pending) if pending else set()
                if vregs(text) & hot:
                    bad.add((no, text))
            ring |= dst
            pending.append(dst)
            if len(pending) > 64:
                pending = pending[-64:]               # (the counter saturates at 63)
            continue
        if in_asm or not pending or not text.startswith(MOVES):
            continue
        hot = set().union(*pending)
        srcs = vregs(text.split(',', 1)[1]) if ',' in text and not text.startswith(('scratch_store', 'buffer_store')) else vregs(text)
        if hot and srcs & hot:
            bad.add((no, text))
    while pending and not pending[0]:                 # operations older than the oldest asm-issued load never matter again
        pending.pop(0)
    return tuple(pending)


WAIT = re.compile(r's_waitcnt\b.*?vmcnt\((\d+)\)')
VM_OPS = ('global_load', 'global_store', 'global_atomic', 'buffer_load', 'buffer_store', 'buffer_atomic', 'flat_load', 'flat_store',
          'scratch_load', 'scratch_store')
STATE_CAP = 400000


def check(path, only=None, verbose=False):
    """Forward exploration of each kernel's control-flow graph, one state per (block, outstanding-operation list) pair.  An operation
    issued from an asm block carries its destination registers; while it is outstanding on SOME path, no compiler-generated
    instruction may read or write them.  Compiler-tracked operations (its own loads / stores and waits) take part in the counter
    like the hardware's.  A kernel whose state space exceeds STATE_CAP is reported as not fully explored (never silently passed)."""
    findings, checked, incomplete = [], 0, []
    for name, body in kernels(path):
        if only and only not in name:
            continue
        if not any(in_asm and text.startswith('global_load_dword') for _, text, in_asm in body):
            continue
        checked += 1
        blocks = blocks_of(body)
        bad, ring = set(), set()
        seen, work = set(), [('^entry', ())]
        while work and len(seen) < STATE_CAP:
            blk, state = work.pop()
            if (blk, state) in seen or blk not in blocks:
                continue
            seen.add((blk, state))
            insts, succ = blocks[blk]
            out = walk_block(insts, state, bad, ring)
            for s_ in succ:
                work.append((s_, out))
        if work:
            incomplete.append(name)
        if verbose:
            print('{}: {} blocks, {} states, {} ring registers, {} finding(s){}'.format(
                name, len(blocks), len(seen), len(ring), len(bad), ' -- NOT fully explored' if work else ''))
        findings += [(name, no, text) for no, text in sorted(bad)]
    for n in incomplete:
        print('# {}: state space above {} -- not fully explored'.format(n, STATE_CAP))
    return checked, findings, incomplete


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    only = None
    if '--kernels' in sys.argv:
        only = sys.argv[sys.argv.index('--kernels') + 1]
        args = [a for a in args if a != only]
    checked, findings, incomplete = check(args[0], only, '--verbose' in sys.argv)
    for name, no, text in findings:
        print('{}:{}: `{}` touches a register whose asm-issued load is still in flight, in {}'.format(args[0], no, text, name))
    print('# ring-register check: {} kernel(s) with an asm-issued load ring, {} finding(s)'.format(checked, len(findings)))
    return 1 if (findings or incomplete) else 0


if __name__ == '__main__':
    sys.exit(main())
