"""Static instruction mix of the kernels in a hipcc -S listing.  usage: isa_stats.py file.s [name-substring]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
for k, (i, name) in enumerate(starts):
    if want not in name:
        continue
    end = next((j for j in range(i, len(txt)) if txt[j].startswith('.Lfunc_end')), len(txt))
    ops = collections.Counter()
    for l in txt[i:end]:
        m = re.match(r'\s+([a-z][a-z0-9_]+)', l)
        if m:
            ops[m.group(1)] += 1
    total = sum(ops.values())
    def grp(pred):
        return sum(v for o, v in ops.items() if pred(o))
    print(name[:100])
    print(f"  total {total}  valu {grp(lambda o: o.startswith('v_'))}  salu {grp(lambda o: o.startswith('s_') and not o.startswith('s_load') and not o.startswith('s_waitcnt'))}"
          f"  f64 {grp(lambda o: o.endswith('_f64'))}  scratch_ld {grp(lambda o: o.startswith('scratch_load'))} scratch_st {grp(lambda o: o.startswith('scratch_store'))}"
          f"  ds {grp(lambda o: o.startswith('ds_'))}  flat {grp(lambda o: o.startswith('flat_'))}  global {grp(lambda o: o.startswith('global_'))}  s_load {grp(lambda o: o.startswith('s_load'))}"
          f"  branch {grp(lambda o: o.startswith('s_cbranch') or o == 's_branch')}  waitcnt {ops['s_waitcnt']}")
    print("  top:", ', '.join(f"{o} {v}" for o, v in ops.most_common(24)))
