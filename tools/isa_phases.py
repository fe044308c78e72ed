"""Static instruction mix per phase (the '; phase N' markers of RT_STAMP) of one kernel in a hipcc -S listing.
usage: isa_phases.py <f32|f64> <mangled-name-prefix>      e.g. isa_phases.py f64 _ZN2rt21render_persist_kernelIdLb0ELi3"""
import collections, os, re, subprocess, sys
unit, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "csrc")
out = f"/tmp/rt_{unit}.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
                f"-ffp-contract={'off' if unit == 'f64' else 'on'}", "-S", "--cuda-device-only", f"rt_{unit}.hip", "-o", out]
               + sys.argv[3:], cwd=src, check=True, stderr=subprocess.DEVNULL)
txt = open(out).read().split('\n')
start = [i for i, l in enumerate(txt) if l.startswith(prefix)][0]
end = next(j for j in range(start, len(txt)) if txt[j].startswith('.Lfunc_end'))
cnt = collections.OrderedDict()
cur = 'pre'
for l in txt[start:end]:
    m = re.search(r'; phase (\d+)', l)
    if m:
        cur = 'after phase ' + m.group(1)
        continue
    mm = re.match(r'\s+([a-z][a-z0-9_]+)', l)
    if not mm:
        continue
    op = mm.group(1)
    d = cnt.setdefault(cur, collections.Counter())
    d['total'] += 1
    for key, pred in (('valu', op.startswith('v_')), ('f64', op.endswith('_f64') or '_f64_' in op),
                      ('lane', op in ('v_readlane_b32', 'v_writelane_b32')), ('scratch', op.startswith('scratch_')),
                      ('div', op.startswith('v_div_fixup')), ('sqrt', op.startswith('v_rsq_f64') or op.startswith('v_sqrt_f64')), ('cnd', op.startswith('v_cndmask')), ('mov', op.startswith('v_mov')),
                      ('gmem', op.startswith('global_') or op.startswith('flat_')), ('ds', op.startswith('ds_')),
                      ('salu', op.startswith('s_') and not op.startswith('s_waitcnt') and not op.startswith('s_nop'))):
        if pred:
            d[key] += 1
for k, d in cnt.items():
    print(f"{k:14s}", ' '.join(f"{a} {b}" for a, b in d.items()))
