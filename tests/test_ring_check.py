"""scripts/check_ring_registers.py (the build-time ISA check of the kernels that keep a ring of asm-issued global loads in ordinary
registers, csrc/chain_dest.hip.h): it must SEE the failure it guards against -- a compiler-generated copy or spill of a register
whose load is still in flight (the regalloc-at-a-join failure documented in compact_tag.hip.h) -- and pass clean code, counted
waits and branches included.  Synthetic assembly: no compiler needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, 'scripts', 'check_ring_registers.py')

HEAD = '_ZN5farnn4testILb1EEEvNS_10RegsParamsE:\n'
TAIL = '\ts_endpgm\n\t.section\t.rodata\n'
ISSUE = ('\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v1, s[4:5]\n\tglobal_load_dwordx4 v[14:17], v2, s[4:5]\n\t;;#ASMEND\n')


def run(body, tmp_path):
    f = tmp_path / 'k.s'
    f.write_text(HEAD + body + TAIL)
    return subprocess.run([sys.executable, SCRIPT, str(f), '--verbose'], capture_output=True, text=True)


def test_clean_ring_passes(tmp_path):
    body = ISSUE + '\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n\tv_fma_f32 v20, v10, v11, v20\n' \
                   '\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v30, v14\n'
    r = run(body, tmp_path)
    assert r.returncode == 0 and '0 finding' in r.stdout, r.stdout + r.stderr


def test_copy_of_an_in_flight_register_is_a_finding(tmp_path):
    # the counted wait releases the OLDER load only: v[14:17] is still in flight when the compiler copies v15
    body = ISSUE + '\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n\tv_mov_b32_e32 v30, v15\n'
    r = run(body, tmp_path)
    assert r.returncode == 1 and 'v_mov_b32_e32 v30, v15' in r.stdout, r.stdout + r.stderr


def test_copy_at_a_join_behind_a_branch_is_a_finding(tmp_path):
    body = ISSUE + '\ts_cbranch_scc1 .LBB0_2\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n.LBB0_2:\n\tv_mov_b64_e32 v[40:41], v[12:13]\n'
    r = run(body, tmp_path)
    assert r.returncode == 1 and 'v_mov_b64_e32' in r.stdout, r.stdout + r.stderr


def test_spill_of_an_in_flight_register_is_a_finding(tmp_path):
    body = ISSUE + '\tscratch_store_dword off, v16, s32\n'
    r = run(body, tmp_path)
    assert r.returncode == 1 and 'scratch_store_dword' in r.stdout, r.stdout + r.stderr


def test_the_compilers_own_loads_count_in_the_counter(tmp_path):
    # a compiler-issued load behind the ring's: vmcnt(1) then leaves IT outstanding and releases both ring loads
    body = ISSUE + '\tglobal_load_dword v50, v3, s[6:7]\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v30, v15\n'
    r = run(body, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
