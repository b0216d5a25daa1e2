"""Instruction mix of the MFMA-heavy loops of one kernel in a hipcc --save-temps .s file (all basic blocks of each loop summed):
   python tools/loop_mix.py file.s kernel_name_substring [min_mfma]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m0 = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)     # the function's own label line
i = m0.end()
j = s.index(".Lfunc_end", i)
body = s[i:j]
for m in re.finditer(r"^\.L(BB\d+_\d+):.*Loop Header.*$", body, re.M):
    name = m.group(1)
    ends = [e.end() for e in re.finditer(r"s_cbranch\S*\s+\.L" + name + r"\b|s_branch\s+\.L" + name + r"\b", body)]
    if not ends:
        continue
    txt = body[m.start():max(ends)]
    n = len(re.findall(r"v_mfma", txt))
    if n < min_mfma:
        continue
    ins = [l.strip().split()[0] for l in txt.split("\n") if l.strip() and not l.strip().startswith((".", ";", "//"))]
    c = Counter(ins)
    print("loop", name, "insts", len(ins), "mfma", n)
    for k, v in c.most_common(70):
        print("   ", k, v)
