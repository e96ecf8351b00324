"""Scan gfx950 assembly / disassembly for packed fp32 instructions that broadcast one half of a VGPR pair through op_sel /
op_sel_hi - the instruction form tools/hazard_repro2.hip shows to compute with the wrong register now and then while other waves
of the CU execute MFMAs.  The library is built so that none exists (edge_attn.cuh: bc_v, split.cuh:
splat4, -fno-slp-vectorize); tests/test_boundary_cpu.py runs this scan on the built libinfgen_hip.so.

    python tools/pk_opsel_scan.py file.s [...]          (compiler -S output or llvm-objdump -d output)
    python tools/pk_opsel_scan.py --lib infgen_amd/libinfgen_hip.so
"""
import os
import re
import subprocess
import sys
import tempfile

PAT = re.compile(r'^\s*(?:[0-9a-f]+:\s+)?(v_pk_(?:fma|mul|add)_f32)\s+(.*?)(?:\s*//.*)?$')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def scan_text(lines):
    """-> (packed fp32 instructions, those that broadcast a half of a VGPR pair by op_sel, {function: count})"""
    tot = risky = 0
    func, per_func = None, {}
    for line in lines:
        m0 = re.match(r'^(?:[0-9a-f]+ <)?(_Z\w+)>?:', line)
        if m0:
            func = m0.group(1)
        m = PAT.match(line)
        if not m:
            continue
        tot += 1
        rest = m.group(2)
        ops_part = re.split(r'\s+op_sel', rest)[0]
        ops = [o.strip() for o in re.split(r',\s*(?![^\[]*\])', ops_part)][1:]          # the sources
        sel = re.search(r'op_sel:\[([0-9,]+)\]', rest)
        selh = re.search(r'op_sel_hi:\[([0-9,]+)\]', rest)
        n = len(ops)
        s = [int(x) for x in sel.group(1).split(',')] if sel else [0] * n
        h = [int(x) for x in selh.group(1).split(',')] if selh else [1] * n
        if any(i < len(s) and i < len(h) and (s[i], h[i]) != (0, 1) and o.startswith('v') for i, o in enumerate(ops)):
            risky += 1
            per_func[func] = per_func.get(func, 0) + 1
    return tot, risky, per_func


def scan_library(path):
    """disassemble every gfx950 code object bundled in a shared library / object file and scan it"""
    tot = risky = 0
    per_func = {}
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, os.path.basename(path))
        with open(path, 'rb') as f, open(lib, 'wb') as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, '--offloading', lib], cwd=d, check=True, capture_output=True)
        objs = [os.path.join(d, n) for n in os.listdir(d) if n.endswith('gfx950')]
        for o in objs:
            dis = subprocess.run([OBJDUMP, '-d', o], check=True, capture_output=True, text=True).stdout
            t, r, pf = scan_text(dis.splitlines())
            tot += t
            risky += r
            for k, v in pf.items():
                per_func[k] = per_func.get(k, 0) + v
    return tot, risky, per_func, len(objs)


def scan_m0(lines):
    """the hand-written LDS-DMA of split.cuh (lds_dma16) sets M0 itself, which is not a legal inline-asm clobber: every write of
    M0 has to be the one in front of a global_load_lds within the next three instructions, and nothing else may read M0
    -> (M0 writes, LDS-DMA loads, offending lines)"""
    ins = []
    for line in lines:
        m = re.match(r'^\s*(?:[0-9a-f]+:\s+)?([sv]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)\s*(.*?)(?:\s*//.*)?$', line)
        if m:
            ins.append((m.group(1), m.group(2)))
    writes = dma = 0
    bad = []
    for i, (op, rest) in enumerate(ins):
        if op.startswith('global_load_lds'):
            dma += 1
        if not re.search(r'\bm0\b', rest):
            continue
        if op == 's_mov_b32' and rest.split(',')[0].strip() == 'm0':
            writes += 1
            if not any(o.startswith('global_load_lds') for o, _ in ins[i + 1:i + 4]):
                bad.append(f'{op} {rest}')
        else:
            bad.append(f'{op} {rest}')
    return writes, dma, bad


def scan_library_m0(path):
    writes = dma = 0
    bad = []
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, os.path.basename(path))
        with open(path, 'rb') as f, open(lib, 'wb') as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, '--offloading', lib], cwd=d, check=True, capture_output=True)
        for n in os.listdir(d):
            if n.endswith('gfx950'):
                dis = subprocess.run([OBJDUMP, '-d', os.path.join(d, n)], check=True, capture_output=True, text=True).stdout
                w, m, b = scan_m0(dis.splitlines())
                writes += w
                dma += m
                bad += b
    return writes, dma, bad


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--lib':
        t, r, pf, n = scan_library(sys.argv[2])
        print(f'{sys.argv[2]}: {n} gfx950 code objects, {t} packed fp32 instructions, {r} broadcast a half of a VGPR pair by op_sel')
        for k, v in sorted(pf.items(), key=lambda kv: -kv[1])[:12]:
            print(f'    {v:5d}  {k[:100]}')
        sys.exit(1 if r else 0)
    for f in sys.argv[1:]:
        t, r, pf = scan_text(open(f))
        print(f'{f}: {t} packed fp32 instructions, {r} broadcast a half of a VGPR pair by op_sel')
        for k, v in sorted(pf.items(), key=lambda kv: -kv[1])[:8]:
            print(f'    {v:5d}  {k[:100]}')
