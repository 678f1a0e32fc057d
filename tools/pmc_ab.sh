#!/bin/bash
# usage (GPU box, repo root): tools/pmc_ab.sh <tag> <lib.so> ...   HBM bytes per launch of K1 / K3 / K4 on the bench shape
# (tools/ab_solve.py), FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes; prints full-launch means per kernel.
R=$GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  name=$(basename $so .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag/$name/$c -o p -- python3 $R/tools/ab_solve.py $R/$so > $R/gpurun_out/pmc_$tag/$name.$c.log 2>&1
  done
  python3 - "$R/gpurun_out/pmc_$tag/$name" "$name" <<'PY'
import csv, glob, sys, collections
root, name = sys.argv[1:3]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    tot[c] = d
for k in sorted(tot["FETCH_SIZE"]):
    if not any(x in k for x in ("k_linearize_imu", "k_assemble", "k_band_solve", "k_linearize_between")):
        continue
    f, w = sorted(tot["FETCH_SIZE"][k]), sorted(tot["WRITE_SIZE"].get(k, [0]))
    fh, wh = f[len(f) // 2:], w[len(w) // 2:]          # the larger half = full launches
    rd, wr = 2 * sum(fh) / len(fh) * 1024 / 1e9, sum(wh) / len(wh) * 1024 / 1e9
    print(f"{name:28s} {k:32s} launches {len(f):3d}  read {rd:7.3f} GB (2*FETCH)  write {wr:7.3f} GB")
PY
done
