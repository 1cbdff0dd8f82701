import re, itertools, subprocess, sys
src = sys.argv[1]
names = sys.argv[2:]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", "/tmp/k.s"], check=True, capture_output=True)
s = open('/tmp/k.s').read()
for name in names:
    cands = [m.group(1) for m in re.finditer(r'^(_Z\S*' + name + r'\S*):', s, re.M)]
    for nm in cands[:1]:
        i = s.index(nm + ':'); j = s.index('.Lfunc_end', i)
        body = [l.strip() for l in s[i:j].split('\n')]
        seq = []
        for l in body:
            if 'global_load_dwordx4' in l: seq.append('L')
            elif 'global_load_dwordx2' in l: seq.append('l')
            elif 's_waitcnt' in l and 'vmcnt' in l: seq.append('W(' + re.search(r'vmcnt\((\d+)\)', l).group(1) + ')')
            elif 'v_mfma' in l: seq.append('M')
            elif 's_barrier' in l: seq.append('|')
            elif 'global_store' in l: seq.append('S')
            elif 'scratch_' in l: seq.append('X')
        comp = ''.join(f"{k}{len(list(g))}" if k in 'MLlSX' else k * len(list(g)) for k, g in itertools.groupby(seq))
        meta = s[j:j + 400000]
        vg = re.search(r'\.vgpr_count:\s+(\d+)', s[s.index(nm, j):] if nm in s[j:] else meta)
        print(nm[:90], "len", len(body))
        print(comp[-600:])
