"""ISA-level guards on the product's gfx950 code objects (CPU side: llvm-objdump on what hipcc built; no GPU).

Two hazards neither hipcc nor the hardware catches (EXPERIMENTS.md R4.1 item 2, R4.7 item 3):

 * asm_read_hazards():  `ds_read_b64_tr_b16` is issued from INLINE ASM in gemm_bf16.hip (the intrinsic drained the LDS-DMA queue every
   k-step), so hipcc believes its destination registers are written when the statement ends.  Between such a read and the next
   `s_waitcnt ... lgkmcnt(0)` on EVERY path no instruction may read or write its destination registers (a register-allocator copy,
   a spill, a reuse): found by walking the control-flow graph from every read.
 * store_data_hazards():  a `buffer_store_*` whose `soffset` is an SGPR lets the wave run on before the store has read its data
   registers when the data is wider than 64 bits (hipcc's hazard recogniser exempts exactly that form; gfx950 does not).  The product
   must not contain a > 64-bit buffer store with an SGPR offset at all, and behind a <= 64-bit one (which the ISA holds free of
   the hazard) no vector instruction may overwrite a data register within `window` instructions -- a margin, so that a toolchain
   that moves the schedule is noticed here and not in a gradient.

`python tools/isa_check.py [objects...]` prints a report; tests/test_isa_hazards.py asserts on the functions."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get('LLVM_OBJDUMP', '/opt/rocm/lib/llvm/bin/llvm-objdump')
_REG = re.compile(r'\b([vas])(?:(\d+)\b|\[(\d+):(\d+)\])')
_LINE = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f ]+)')
_FUNC = re.compile(r'^[0-9a-fA-F]+ <(.+)>:$')


def device_disassembly(obj):
    """Disassembly text of the gfx950 code object bundled in host object `obj` (or of `obj` itself if it is one)."""
    tmp = tempfile.mkdtemp(prefix='jamie_isa_')
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, '--offloading', local], check=True, capture_output=True, text=True)
        dev = [f for f in os.listdir(tmp) if 'amdgcn' in f]
        target = os.path.join(tmp, dev[0]) if dev else local
        return subprocess.run([OBJDUMP, '-d', target], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class Inst:
    __slots__ = ('addr', 'size', 'op', 'args', 'func')

    def __init__(self, addr, size, op, args, func):
        self.addr, self.size, self.op, self.args, self.func = addr, size, op, args, func

    def __repr__(self):
        return f'{self.addr:#x}: {self.op} {self.args}'


def parse(text):
    """[Inst] in address order (all kernels of the object; `func` = the enclosing symbol)."""
    out, func = [], None
    for ln in text.splitlines():
        m = _FUNC.match(ln)
        if m:
            func = m.group(1)
            continue
        m = _LINE.match(ln)
        if not m or func is None:
            continue
        op, args, addr, enc = m.group(1), m.group(2), int(m.group(3), 16), m.group(4).split()
        out.append(Inst(addr, 4 * len(enc), op, args, func))
    return out


def regs(operand_text, kinds='v'):
    """Set of (kind, index) registers named in an operand string."""
    s = set()
    for m in _REG.finditer(operand_text):
        k = m.group(1)
        if k not in kinds:
            continue
        if m.group(2) is not None:
            s.add((k, int(m.group(2))))
        else:
            s.update((k, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return s


def _split_operands(args):
    out, depth, cur = [], 0, ''
    for ch in args:
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur.strip())
            cur = ''
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _branch_target(ins, enc_words):
    simm = enc_words & 0xFFFF
    if simm & 0x8000:
        simm -= 0x10000
    return ins.addr + 4 + 4 * simm


def _waits_lgkm0(ins):
    if ins.op != 's_waitcnt':
        return False
    a = ins.args.replace(' ', '')
    if 'lgkmcnt(0)' in a:
        return True
    return a in ('0', '0x0')          # s_waitcnt 0: every counter


def asm_read_hazards(text, max_path=6000):
    """[(read Inst, offending Inst)]: an instruction that touches a destination of a ds_read_b64_tr_b16 before the wait."""
    insts = parse(text)
    index = {i.addr: n for n, i in enumerate(insts)}
    enc = {}
    for ln in text.splitlines():
        m = _LINE.match(ln)
        if m:
            enc[int(m.group(3), 16)] = int(m.group(4).split()[0], 16)
    bad = []
    for n, ins in enumerate(insts):
        if ins.op != 'ds_read_b64_tr_b16':
            continue
        dst = regs(_split_operands(ins.args)[0])
        seen, stack, steps = set(), [n + 1], 0
        while stack and steps < max_path:
            k = stack.pop()
            while k < len(insts) and k not in seen and steps < max_path:
                seen.add(k)
                steps += 1
                cur = insts[k]
                if cur.func != ins.func or cur.op in ('s_endpgm',):
                    break
                if _waits_lgkm0(cur):
                    break
                if cur.op == 'ds_read_b64_tr_b16':
                    # another transposed read: its own destination must differ, its address may not be a pending destination
                    if regs(cur.args) & dst:
                        bad.append((ins, cur))
                        break
                elif regs(cur.args) & dst:
                    bad.append((ins, cur))
                    break
                if cur.op.startswith('s_cbranch') or cur.op == 's_branch':
                    tgt = index.get(_branch_target(cur, enc[cur.addr]))
                    # EXEC == 0 paths are not walked: behind the fall-through of s_cbranch_execnz and behind the taken
                    # s_cbranch_execz no lane is active, so vector instructions there neither read nor write a register
                    if cur.op == 's_cbranch_execz':
                        k += 1
                        continue
                    if tgt is not None:
                        stack.append(tgt)
                    if cur.op in ('s_branch', 's_cbranch_execnz'):
                        break
                k += 1
    return bad


_VALU_PREFIX = ('v_',)


def store_data_hazards(text, window=8):
    """(wide, narrow): `wide` = buffer stores of more than 64 bits with an SGPR soffset (must be empty);
    `narrow` = [(store, writer)]: a vector instruction overwriting a data register of a <= 64-bit buffer store with an SGPR
    soffset within `window` instructions behind it (straight-line; a branch ends the window)."""
    insts = parse(text)
    wide, narrow = [], []
    for n, ins in enumerate(insts):
        if not ins.op.startswith('buffer_store_'):
            continue
        ops = _split_operands(ins.args.split(' offen')[0].split(' idxen')[0].split(' offset:')[0].split(' sc')[0].split(' nt')[0])
        if len(ops) < 4:
            continue
        soff = ops[3].split()[0]
        if not re.fullmatch(r's\d+|m0|ttmp\d+', soff):
            continue                                   # 0 / a literal: not the exempted form
        data = regs(ops[0])
        if len(data) > 2:
            wide.append(ins)
            continue
        for k in range(n + 1, min(n + 1 + window, len(insts))):
            cur = insts[k]
            if cur.func != ins.func or cur.op.startswith('s_cbranch') or cur.op in ('s_branch', 's_endpgm', 's_barrier'):
                break
            if cur.op.startswith(_VALU_PREFIX) or cur.op.startswith(('ds_read', 'buffer_load', 'global_load')):
                o = _split_operands(cur.args)
                if o and regs(o[0]) & data and not cur.op.startswith(('v_cmp', 'v_cmpx')):
                    narrow.append((ins, cur, k - n))
                    break
    return wide, narrow


def sgpr_offset_stores(text):
    """Every buffer store whose soffset is a register (the form hipcc's hazard recogniser exempts), of any width."""
    out = []
    for ins in parse(text):
        if not ins.op.startswith('buffer_store_'):
            continue
        ops = _split_operands(ins.args.split(' offen')[0].split(' idxen')[0].split(' offset:')[0].split(' sc')[0].split(' nt')[0])
        if len(ops) >= 4 and re.fullmatch(r's\d+|m0|ttmp\d+', ops[3].split()[0]):
            out.append(ins)
    return out


def product_objects():
    d = os.path.join(ROOT, 'jamie_amd', 'csrc', '_obj')
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith('.o'))


if __name__ == '__main__':
    for obj in (sys.argv[1:] or product_objects()):
        text = device_disassembly(obj) if obj.endswith('.o') else open(obj).read()
        n_tr = sum(1 for i in parse(text) if i.op == 'ds_read_b64_tr_b16')
        bad = asm_read_hazards(text)
        wide, narrow = store_data_hazards(text)
        n_so = sum(1 for i in parse(text) if i.op.startswith('buffer_store_'))
        print(f'{os.path.basename(obj)}: {n_tr} transposed reads, {len(bad)} touched before their wait; {n_so} buffer stores, '
              f'{len(wide)} wide with an SGPR offset, {len(narrow)} narrow ones with a data register overwritten within the window')
        for r, o in bad[:6]:
            print('   READ ', r, ' <-', o, ' in', r.func[:60])
        for s_ in wide[:6]:
            print('   WIDE ', s_, ' in', s_.func[:60])
        for s_, w, dist in narrow[:10]:
            print(f'   NARROW {s_}  <- {w} (+{dist}) in {s_.func[:60]}')
