import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else 'dec_'
rows = list(c.execute("select start,end,name,stream_id from kernels where name like '%" + pat + "%' order by start"))
sids = sorted(set(r[3] for r in rows))
tot = 0
lo = min(r[0] for r in rows); hi = max(r[1] for r in rows)
for s in sids:
    rr = [r for r in rows if r[3] == s]
    b = sum(r[1]-r[0] for r in rr)
    tot += b
    print("stream", s, "n", len(rr), "busy ms", b/1e6, "first", (rr[0][0]-lo)/1e6, "last", (rr[-1][1]-lo)/1e6)
print("span ms", (hi-lo)/1e6, "sum busy / span =", tot/(hi-lo))
