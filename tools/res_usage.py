"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin): name, VGPR, AGPR, spill, scratch."""
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = dict(name=m.group(1)); rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
only = len(sys.argv) > 1 and sys.argv[1] == "spills"
for r in rows:
    if only and not (r.get("spill") or r.get("scratch")): continue
    name = r["name"].replace("_ZN12_GLOBAL__N_1", "").replace("EEEvNS_5ConvKE", "").replace("EEEvNS_3WgKE", "")
    print(f"{name[:90]:90s} v{r.get('vgpr')} a{r.get('agpr')} spill{r.get('spill')} scratch{r.get('scratch')} occ{r.get('occ')}")
