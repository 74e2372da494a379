cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pk2 /tmp/pk6
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk2 -o ks -- python3 bench.py --schedule sequential --steps 2 --warmup 1 --no-cpu-baseline > /tmp/b2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk6 -o ks -- python3 bench.py --schedule sequential --steps 6 --warmup 1 --no-cpu-baseline > /tmp/b6.log 2>&1 || exit 1
python3 - <<'P'
import csv,glob
def load(d):
    f=glob.glob(d+'/**/ks_kernel_stats.csv', recursive=True)[0]
    return {r['Name']:(int(r['Calls']),float(r['TotalDurationNs'])/1e6) for r in csv.DictReader(open(f))}
a,b=load('/tmp/pk2'),load('/tmp/pk6')
rows=[]
for n,(c6,t6) in b.items():
    c2,t2=a.get(n,(0,0.0))
    rows.append(((t6-t2)/4,(c6-c2)/4,n))
rows.sort(reverse=True)
tot=0
for ms,c,n in rows:
    if not ('dec_' in n[:40]): tot+=ms
    if ms>0.3 and not ('dec_' in n[:40]): print(f"{ms:8.2f} ms/step {c:8.1f} calls/step  {n[:90]}")
print("front-end kernel ms per step (differenced):", round(tot,1))
P
