cd $GRAFT_REPO_ROOT; OUT=gpurun_out
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extra-configs --config poisson3d_128 --ordering builtin --profile-out $OUT/lt_p3d.txt > $OUT/b.json 2> $OUT/b.err; tail -2 $OUT/b.err
python - <<'PY'
import collections
rows=[l.split() for l in open('gpurun_out/lt_p3d.txt').read().strip().split('\n')[1:]]
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for r in rows:
    c=r[7]; agg[c][0]+=1; agg[c][1]+=float(r[6]); agg[c][2]+=float(r[5])
for c,(n,ms,gf) in sorted(agg.items(), key=lambda x:-x[1][1]):
    print(f"{c:12s} n={n:4d} ms={ms:8.3f} gflop={gf:9.2f} avg_us={ms/n*1e3:8.1f} tflops={gf/ms if ms else 0:6.2f}")
print("total", sum(v[1] for v in agg.values()))
lev=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(int)
for r in rows:
    lev[int(r[2])][r[7]]+=float(r[6])
    if r[7]=='chain': cnt[int(r[2])]+=int(r[3])
for L in sorted(lev):
    d=lev[L]; print(L, "units", cnt[L], " ".join(f"{k}={v:.2f}" for k,v in sorted(d.items())))
PY
