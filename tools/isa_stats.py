#!/usr/bin/env python3
"""Instruction statistics of the kernels in a hipcc --save-temps .s file whose name contains a pattern.
usage: python tools/isa_stats.py file.s k_shade"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
for m in re.finditer(r'^(\S+): *; @\S+\n', s, re.M):
    fn = m.group(1)
    if pat not in fn:
        continue
    body = s[m.end():]
    body = body[:body.index('.Lfunc_end')]
    insts = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(insts)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    vg = re.search(re.escape(fn) + r'\.num_vgpr, (\d+)', s)
    sc = re.search(re.escape(fn) + r'\.private_seg_size, (\d+)', s)
    short = re.sub(r'^_ZN2tr12_GLOBAL__N_1\d+', '', fn)[:40]
    print(f"{short:40s} insts {len(insts):6d} valu {grp('v_'):6d} vgpr {vg.group(1) if vg else '?':>4s} scratch_B {sc.group(1) if sc else '?':>5s} | div_scale {c.get('v_div_scale_f32', 0):4d} "
          f"div_fmas {c.get('v_div_fmas_f32', 0):4d} rcp {grp('v_rcp'):4d} sqrt {grp('v_sqrt'):4d} rsq {grp('v_rsq'):4d} sin/cos {grp('v_sin') + grp('v_cos'):3d} exp/log {grp('v_exp') + grp('v_log'):3d} "
          f"scratch_ops {grp('scratch_'):4d} vmem {grp('global_') + grp('buffer_') + grp('flat_'):4d}")
