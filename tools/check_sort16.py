"""The 16-input sorting network of csrc/assemble.hip (sort16: 60 comparators, 10 layers), checked by the 0-1 principle: a comparator
network sorts every input iff it sorts all 2^16 sequences of zeros and ones.   python tools/check_sort16.py"""
import re, os
import numpy as np
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "large-steps-pytorch_amd", "csrc", "assemble.hip")).read()
body = src[src.index("void sort16("):src.index("#undef LS_CE")]
ces = [(int(a), int(b)) for a, b in re.findall(r"LS_CE\((\d+), (\d+)\)", body)]
x = ((np.arange(1 << 16)[:, None] >> np.arange(16)) & 1).astype(np.int8)
for i, j in ces:
    lo, hi = np.minimum(x[:, i], x[:, j]), np.maximum(x[:, i], x[:, j])
    x[:, i], x[:, j] = lo, hi
ok = bool((np.diff(x, axis=1) >= 0).all())
print(f"{len(ces)} comparators, sorts all 65536 0/1 inputs: {ok}")
raise SystemExit(0 if ok and len(ces) == 60 else 1)
