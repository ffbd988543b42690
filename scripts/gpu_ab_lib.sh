#!/bin/bash
# A/B of two builds of the library on ONE box: bash scripts/gpu_ab_lib.sh tag libA.so libB.so [rounds]
# (each round: bench.py with libA, then with libB; per-launch tables of the last round kept)
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/ablib_$1
mkdir -p $OUT
R=${4:-2}
for r in $(seq 1 $R); do
  for v in A B; do
    lib=$2; [ $v = B ] && lib=$3
    cp $lib spllt_amd/libspllt_hip.so
    timeout -k 10 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-check --no-extra-configs ${BENCH_ARGS} --profile-out $OUT/table_$v.txt > $OUT/b_$v.json 2> $OUT/b_$v.err || { tail -3 $OUT/b_$v.err; exit 1; }
    python - $v $lib $OUT/b_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("%s %-50s ms %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"]), flush=True)
PY
  done
done
python - $OUT <<'PY'
import sys, collections
def load(f): return [l.split() for l in open(f).read().strip().splitlines()[1:]]
a=load(sys.argv[1]+"/table_A.txt"); b=load(sys.argv[1]+"/table_B.txt")
ca=collections.defaultdict(float); cb=collections.defaultdict(float)
for x in a: ca[x[7]]+=float(x[6])
for x in b: cb[x[7]]+=float(x[6])
for k in ca: print("%-10s A %.3f  B %.3f ms (launches alone)" % (k, ca[k], cb.get(k,0)))
for lev in sorted(set(int(x[2]) for x in a)):
    sa=sum(float(x[6]) for x in a if x[7]=="between" and int(x[2])==lev); sb=sum(float(x[6]) for x in b if x[7]=="between" and int(x[2])==lev)
    if sa: print("between, level %2d: A %.3f  B %.3f ms" % (lev, sa, sb))
PY
