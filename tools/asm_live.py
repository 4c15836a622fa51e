#!/usr/bin/env python3
"""VGPR liveness of one kernel in a hipcc -S listing: backward dataflow over the basic blocks, then the
number of live VGPRs before every instruction.  Prints the high-water mark per basic block (blocks above
a threshold) so that the code region that sets a kernel's register demand can be found.
usage: asm_live.py file.s mangled_name_substring [min_live_to_print [block_label_to_list]]
(approximate: a write under a partial exec mask is treated as a full definition)"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dump = sys.argv[4] if len(sys.argv) > 4 else None   # label of a block to list instruction by instruction
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
a = m.start()
b = s.index('.Lfunc_end', a)

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(tok):
    out = set()
    for mm in REG.finditer(tok):
        if mm.group(1) is not None:
            out.add(int(mm.group(1)))
        else:
            out.update(range(int(mm.group(2)), int(mm.group(3)) + 1))
    return out


NO_DEF = ('scratch_store', 'global_store', 'ds_write', 'ds_add', 'ds_or', 'buffer_store', 'flat_store', 'global_atomic',
          'v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane', 's_', 'global_load_lds', 'ds_bpermute_NO')


def defuse(ins):
    op, _, rest = ins.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    if op.startswith('global_load_lds'):
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if op.startswith(NO_DEF) and not (op.startswith('global_atomic') and 'sc0' in ins):
        u = set()
        for o in ops:
            u |= regs(o)
        return set(), u
    d = regs(ops[0]) if ops else set()
    u = set()
    for o in ops[1:]:
        u |= regs(o)
    if op.startswith('v_writelane') or 'dpp' in op or 'row_' in ins or 'wave_sh' in ins or 'quad_perm' in ins or op.startswith(('v_mac', 'v_fmac', 'v_dot')) and len(ops) == 3:
        u |= d   # read-modify-write of the destination (bound_ctrl:0 DPP keeps old lanes)
    return d, u


blocks = []   # (label, [instrs])
cur = ('entry', [])
for l in s[a:b].split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        blocks.append(cur)
        cur = (mm.group(1), [])
    else:
        t = l.split(';')[0].strip()
        if t and not t.startswith(('.', '//')) and not t.endswith(':'):
            cur[1].append(t)
blocks.append(cur)
idx = {name: i for i, (name, _) in enumerate(blocks)}
succ = []
for i, (name, ins) in enumerate(blocks):
    sc = set()
    fall = True
    for t in ins:
        op = t.split()[0]
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = t.split()[-1]
            if tgt in idx:
                sc.add(idx[tgt])
            if op == 's_branch':
                fall = False
        if op == 's_endpgm':
            fall = False
    if fall and i + 1 < len(blocks):
        sc.add(i + 1)
    succ.append(sc)
du = [[defuse(t) for t in ins] for _, ins in blocks]
live_in = [set() for _ in blocks]
changed = True
while changed:
    changed = False
    for i in range(len(blocks) - 1, -1, -1):
        live = set()
        for j in succ[i]:
            live |= live_in[j]
        for d, u in reversed(du[i]):
            live = (live - d) | u
        if live != live_in[i]:
            live_in[i] = live
            changed = True
line = 0
peak_all = 0
for i, (name, ins) in enumerate(blocks):
    live = set()
    for j in succ[i]:
        live |= live_in[j]
    counts = []
    for d, u in reversed(du[i]):
        live = (live - d) | u
        counts.append(len(live))
    counts.reverse()
    if dump == name:
        for c, t in zip(counts, ins):
            print(f'   {c:3d}  {t}')
    if counts:
        pk = max(counts)
        peak_all = max(peak_all, pk)
        if pk >= thr:
            k = counts.index(pk)
            print(f"{name:12s} instrs {len(ins):5d} live-in {len(live_in[i]):3d} peak {pk:3d} at +{k}: {ins[k][:70]}")
print('peak live VGPRs', peak_all)
