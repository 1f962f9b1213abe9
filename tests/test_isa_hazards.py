"""ISA-level guards on the product's gfx950 code objects (CPU: llvm-objdump on what hipcc built, no GPU; tools/isa_check.py).

Two hazards that neither hipcc nor the hardware catches were found in round 4 and were guarded by exact-product tests on the
GPU only; the original store bug was invisible in the product's own tile configurations "by scheduling luck" (EXPERIMENTS.md
R4.7 item 3), so a toolchain bump could move it.  These tests read the instruction streams themselves."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_check as ic  # noqa: E402

OBJ = os.path.join(ROOT, 'jamie_amd', 'csrc', '_obj')


@pytest.fixture(scope='module')
def built():
    sys.path.insert(0, ROOT)
    from jamie_amd.build import build_library
    build_library()                       # (no-op when the objects are current; hipcc cross-compiles without a GPU)
    if not os.path.exists(ic.OBJDUMP):
        pytest.skip('llvm-objdump not found')
    return {n: ic.device_disassembly(os.path.join(OBJ, n + '.o')) for n in ('gemm_bf16', 'gemm_f32')}


def test_checker_sees_what_it_should_on_a_synthetic_stream():
    """The checker itself: a destination touched before the wait, a clean read, a loop whose back edge carries a pending read to
    a wait at the loop top, EXEC == 0 paths not walked; wide / narrow buffer stores with an SGPR offset."""
    def line(addr, text, enc='BF800000'):
        return f'\t{text:60s} // {addr:012X}: {enc}'
    body = [
        '0000000000001000 <k>:',
        line(0x1000, 'ds_read_b64_tr_b16 v[2:3], v9', 'D9C60000 02000009'),
        line(0x1008, 'v_mov_b32_e32 v4, v2', '7E080302'),                       # reads a destination before the wait: flagged
        line(0x100C, 's_waitcnt lgkmcnt(0)', 'BF8CC07F'),
        line(0x1010, 'ds_read_b64_tr_b16 v[6:7], v9', 'D9C60000 06000009'),
        line(0x1018, 'v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[14:17], a[0:15]', 'D3B60000 04021D0A'),
        line(0x1020, 's_cbranch_scc1 65531', 'BF85FFFB'),                       # back to 0x1010: another read of v[6:7]: same destination
        line(0x1024, 's_waitcnt vmcnt(0) lgkmcnt(0)', 'BF8C0070'),
        line(0x1028, 'v_mov_b32_e32 v8, v6', '7E100306'),                       # behind the wait: fine
        line(0x102C, 'ds_read_b64_tr_b16 v[20:21], v9', 'D9C60000 14000009'),
        line(0x1034, 's_cbranch_execnz 2', 'BF890002'),                         # fall-through = EXEC 0: not walked
        line(0x1038, 'v_mov_b32_e32 v20, 0', '7E280280'),
        line(0x103C, 's_nop 0', 'BF800000'),
        line(0x1040, 's_waitcnt lgkmcnt(0)', 'BF8CC07F'),
        line(0x1044, 'buffer_store_dwordx4 v[30:33], v40, s[8:11], s5 offen', 'E07C1000 05021E28'),
        line(0x104C, 'buffer_store_dwordx2 v[34:35], v40, s[8:11], s5 offen nt', 'E0761000 05022228'),
        line(0x1054, 'v_add_f32_e32 v35, v1, v2', '02460501'),                  # overwrites a data register one instruction later
        line(0x1058, 'buffer_store_dwordx4 v[30:33], v40, s[8:11], 0 offen', 'E07C1000 80021E28'),
        line(0x1060, 's_endpgm', 'BF810000'),
    ]
    text = '\n'.join(body)
    bad = ic.asm_read_hazards(text)
    assert [(r.addr, o.addr) for r, o in bad] == [(0x1000, 0x1008), (0x1010, 0x1010)], bad
    wide, narrow = ic.store_data_hazards(text)
    assert [w.addr for w in wide] == [0x1044]
    assert [(s.addr, w.addr, d) for s, w, d in narrow] == [(0x104C, 0x1054, 1)]


def test_no_instruction_touches_an_asm_read_destination_before_its_wait(built):
    """gemm_bf16.hip issues ds_read_b64_tr_b16 from inline asm (the intrinsic made hipcc drain the LDS-DMA queue every k-step);
    hipcc then holds the destination written when the statement ends.  On every path from such a read to the next
    `s_waitcnt lgkmcnt(0)` nothing may read or write its destination -- no register-allocator copy for the 64 -> 128-bit
    concatenation, no spill, no reuse."""
    text = built['gemm_bf16']
    reads = [i for i in ic.parse(text) if i.op == 'ds_read_b64_tr_b16']
    assert len(reads) >= 400                                   # the dX / dW instantiations are there
    bad = ic.asm_read_hazards(text)
    assert not bad, [(str(r), str(o), r.func[:70]) for r, o in bad[:8]]


def test_no_buffer_store_with_a_scalar_offset(built):
    """A buffer store of more than 64 bits with an SGPR `soffset` lets the wave overwrite its data registers before the store has
    read them (found in round 4: wrong .y elements in fixed lanes; hipcc's hazard recogniser exempts exactly that form).  The ISA
    holds stores of up to 64 bits free of that hazard, and round 4 kept the scalar offset on them on that reading (the data
    register of such a store IS overwritten by the very next instruction in those streams); since round 5 every lean store
    path carries its row offset in the vector offset, so the exempted form does not occur in the product at all -- wide or
    narrow -- and a toolchain that schedules differently has nothing to break.  (A/B: -DJAMIE_STORE_SOFF, profiles/r05_ab_store_voff*.)"""
    for name, text in built.items():
        assert sum(1 for i in ic.parse(text) if i.op.startswith('buffer_store_')) > 100, name
        hits = ic.sgpr_offset_stores(text)
        assert not hits, (name, len(hits), [str(w) for w in hits[:8]])
        wide, narrow = ic.store_data_hazards(text)
        assert not wide and not narrow
