"""Loops of one kernel in hipcc's assembly (-S --offload-device-only): instructions, scratch (spill) traffic, global and LDS
accesses per loop body.  Usage: python tools/asm_loops.py file.s first_line last_line"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')[int(sys.argv[2]) - 1:int(sys.argv[3])]
lab = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: lab[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in lab and lab[t] < i: loops.append((lab[t], i))
def is_instr(l): return l.startswith('\t') and not l.strip().startswith(('.', ';'))
print("kernel: %d instructions, %d scratch loads, %d scratch stores" % (sum(map(is_instr, lines)), sum('scratch_load' in l for l in lines), sum('scratch_store' in l for l in lines)))
for a, b in sorted(loops):
    body = [l for l in lines[a:b + 1] if is_instr(l)]
    print("loop %6d-%6d: %5d instr, scratch ld/st %2d/%2d, global ld/st %2d/%2d, ds %3d, valu %4d" % (
        a, b, len(body), sum('scratch_load' in l for l in body), sum('scratch_store' in l for l in body),
        sum('global_load' in l for l in body), sum('global_store' in l or 'global_atomic' in l for l in body),
        sum(l.strip().startswith('ds_') for l in body), sum(l.strip().startswith('v_') for l in body)))
