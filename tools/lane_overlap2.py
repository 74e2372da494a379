import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start,end,name,stream_id,queue_id,grid_x,workgroup_x from kernels where name like '%lane_probe_spin%' order by start"))
t0 = rows[0][0]
by = collections.defaultdict(list)
for r in rows: by[(r[3], r[4])].append(r)
for k in sorted(by):
    rr = by[k]
    cnt = collections.Counter((r[5], r[6]) for r in rr)
    print("stream/queue", k, "n", len(rr), dict(cnt), "first %.2f ms last %.2f ms" % ((rr[0][0]-t0)/1e6, (rr[-1][1]-t0)/1e6))
# timeline of the first 40 kernels overall
for r in rows[:60]:
    print("%.1f %.1f s%d q%d grid %d" % ((r[0]-t0)/1e3, (r[1]-t0)/1e3, r[3], r[4], r[5]))
