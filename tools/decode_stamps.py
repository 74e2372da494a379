"""Reads the decode-lane trace written by libccx when CCX_DEC_STAMPS=<file> is set (csrc/whisper.hip, dec_stamp_kernel): per
lane, how long the cross attention of a layer takes while the other lanes run their chains, and how long the chain between two
cross attentions takes while another lane streams.  Usage: python tools/decode_stamps.py <file> [decode index]"""
import sys
from collections import defaultdict

import numpy as np

blocks, cur = [], None
for line in open(sys.argv[1]):
    p = line.split()
    if p[0] == "decode":
        cur = {"B": int(p[2]), "lanes": {}}
        blocks.append(cur)
    elif p[0] == "lane" and cur is not None:
        v = np.array([int(x) for x in p[3:]], dtype=np.uint64)
        cur["lanes"][int(p[1])] = ((v >> np.uint64(8)).astype(np.int64) * 10, (v & np.uint64(255)).astype(np.int64))   # ns, tag
which = int(sys.argv[2]) if len(sys.argv) > 2 else max(range(len(blocks)), key=lambda i: sum(len(t) for t, _ in blocks[i]["lanes"].values()))
blk = blocks[which]
print(f"decode #{which}: B = {blk['B']}, {len(blk['lanes'])} lanes")
names = {16: "LN1+QKV", 17: "self attention", 18: "Wo", 1: "LNc+Wcq", 2: "cross attention", 19: "Wco", 20: "LN2+fc1", 21: "fc2", 3: "final LN + logits + select"}
for lane, (t, tag) in sorted(blk["lanes"].items()):
    seg = defaultdict(list)
    for k in range(1, len(t)):
        seg[int(tag[k])].append((t[k] - t[k - 1]) / 1e3)          # us since the previous stamp, attributed to the stamp's tag
    ends = t[tag == 3]
    step = np.diff(ends) / 1e3
    skip = len(step) // 4
    print(f"lane {lane}: {len(ends)} steps, step time median {np.median(step[skip:]):.0f} us")
    for tg in sorted(seg, key=lambda x: (x != 1, x != 2, x)):
        a = np.array(seg[tg][len(seg[tg]) // 4:])
        what = names.get(tg, str(tg))
        if tg == 1 and 16 not in seg:
            what = "chain between two cross attentions (fc1/fc2 of the last layer, LN1..Wcq of this one)"
        if tg == 3 and 16 not in seg:
            what = "last layer's Wco..fc2 + final LN + logits + select"
        print(f"   {what:90s} n {len(a):6d}  median {np.median(a):7.1f} us  p90 {np.percentile(a, 90):7.1f}  sum/step {a.sum() / max(1, len(step) - skip):8.1f} us")
