"""instruction mix per kernel of a gfx950 assembly file (tools/kres.sh leaves them in /tmp/isa):
python tools/isa_mix.py /tmp/isa/afx_cwt.s [name filter]"""
import re
import sys
from collections import Counter

flt = sys.argv[2] if len(sys.argv) > 2 else ""
name, c = None, None
for l in open(sys.argv[1]).read().split("\n"):
    m = re.match(r"^(_Z\S+):", l)
    if m:
        name, c = m.group(1), Counter()
        continue
    t = l.strip()
    if name is None or not l.startswith("\t") or not t or t[0] in ".;":
        continue
    i = t.split()[0]
    if i == "s_endpgm":
        if re.search(flt, name):
            print(name[:70], sum(c.values()), dict(c))
        name = None
    elif i.startswith("v_pk"):
        c["v_pk"] += 1
    elif i.startswith("v_mfma"):
        c["mfma"] += 1
    elif i.startswith("v_"):
        c["valu"] += 1
    elif i.startswith("s_"):
        c["salu"] += 1
    elif i.startswith("ds_"):
        c["lds"] += 1
    elif i.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] += 1
    else:
        c[i] += 1
