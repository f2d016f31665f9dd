"""CPU (build-time): the hand-scheduled code of csrc/gnn.hip, checked in the ISA hipcc produced for gfx950.

Three sites are written as instructions because the compiler's own forms were slower or (the lane-swap builtins of
this toolchain) wrong, and each carries wait states the hazard recognizer does not insert for inline asm:
  * eh_dpp_xor4_add: two bank-masked `v_add_f32_dpp` into one register (the 8-head walk's last score reduction);
  * eh_xor_sum<16/32>: `v_permlane16_swap_b32` / `v_permlane32_swap_b32` (slot merge of the 1/2/4-head walks);
  * EH_BLOCK_SYNC: `s_waitcnt lgkmcnt(0); s_barrier` instead of __syncthreads() in the fused layer kernel.
A toolchain bump that reorders or drops what surrounds them would change results only at near-tie decisions; this
test pins the instruction patterns themselves (tests/test_gpu_handscheduled.py pins their results)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'graph-generation-edge_amd', 'csrc')
ASM = os.path.join(ROOT, 'graph-generation-edge_amd', 'lib', 'gnn.s')


@pytest.fixture(scope='module')
def isa():
    src_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in ('gnn.hip', 'gemm.h', 'common.h', 'gemv.h'))
    if not os.path.exists(ASM) or os.path.getmtime(ASM) < src_time:
        subprocess.check_call(['make', '-C', CSRC, ASM], stdout=subprocess.DEVNULL)
    lines = []
    for ln in open(ASM):
        ln = ln.strip()
        if ln.startswith(';;#ASM'):
            lines.append(ln)
            continue
        ln = ln.split(';')[0].strip()                          # trailing comments (labels carry `; @name`)
        if ln and not ln.startswith('.'):
            lines.append(ln)
    return lines


def _function_ranges(lines, prefix):
    """[(start, end)] of the kernels whose mangled name contains `prefix`: from its label to the next function label
    (a kernel has several s_endpgm: early exits)."""
    labels = [i for i, ln in enumerate(lines) if ln.endswith(':') and ln.startswith('_Z')]
    ends = labels[1:] + [len(lines)]
    return [(a, b) for a, b in zip(labels, ends) if prefix in lines[a]]


def test_bank_masked_dpp_adds_keep_their_wait_states(isa):
    sites = [i for i, ln in enumerate(isa) if ln.startswith('v_add_f32_dpp') and 'bank_mask:0x5' in ln]
    assert len(sites) >= 2                                     # k_layer<8>: the stream walk and the long-row walk
    for i in sites:
        m = re.match(r'v_add_f32_dpp (v\d+), (v\d+), (v\d+) row_shl:4 row_mask:0xf bank_mask:0x5$', isa[i])
        assert m, isa[i]
        dst, a, b = m.groups()
        assert a == b and dst != a, isa[i]                     # early-clobber output: the second add reads the source again
        assert isa[i - 2] == ';;#ASMSTART' and isa[i - 1] == 's_nop 1', isa[i - 2:i + 1]
        assert isa[i + 1] == f'v_add_f32_dpp {dst}, {a}, {a} row_shr:4 row_mask:0xf bank_mask:0xa', isa[i + 1]
        assert isa[i + 2] == 's_nop 1' and isa[i + 3] == ';;#ASMEND', isa[i:i + 4]


def test_lane_swaps_keep_their_wait_states(isa):
    sites = [i for i, ln in enumerate(isa) if ln.startswith('v_permlane16_swap_b32') or ln.startswith('v_permlane32_swap_b32')]
    assert len(sites) >= 18                                    # 9 values (8 channels + the weight total) per stage and kernel
    for i in sites:
        m = re.match(r'v_permlane(16|32)_swap_b32 (v\d+), (v\d+)$', isa[i])
        assert m and m.group(2) != m.group(3), isa[i]          # two different registers: both halves of the exchange
        assert isa[i - 2] == ';;#ASMSTART' and isa[i - 1] == 's_nop 1', isa[i - 2:i + 1]
        assert isa[i + 1] == 's_nop 1' and isa[i + 2] == ';;#ASMEND', isa[i:i + 3]


def test_fused_layer_kernel_barriers_wait_for_lds_only(isa):
    """Every block barrier of k_layer<h> is EH_BLOCK_SYNC: `s_waitcnt lgkmcnt(0)` directly before `s_barrier`, nothing
    in between, and no vmcnt(0) drain in front of it (that is what __syncthreads() would add)."""
    ranges = _function_ranges(isa, 'k_layer')
    assert len(ranges) == 4                                    # heads 1, 2, 4, 8
    for lo, hi in ranges:
        bars = [i for i in range(lo, hi) if isa[i] == 's_barrier']
        assert len(bars) >= 8
        for i in bars:
            assert isa[i - 1] == 's_waitcnt lgkmcnt(0)' and isa[i - 2] == ';;#ASMSTART' and isa[i + 1] == ';;#ASMEND', isa[i - 2:i + 2]
            assert 'vmcnt(0)' not in isa[i - 3], isa[i - 3:i + 1]


def test_product_source_carries_no_timing_switches():
    """the wrong-results-by-construction timing switches and the debug array live in
    tools/experiments/k_layer_timing_switches.patch, not in the product."""
    src = open(os.path.join(CSRC, 'gnn.hip')).read()
    for word in ('EH_EXP_', 'eh_dbg', 'EH_STAMP'):
        assert word not in src, word
