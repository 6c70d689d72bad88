#!/bin/bash
# kernel trace of the 16 384-point UNet forward (run through gpurun): bash tests/probe/unet_trace.sh <tag>
TAG=${1:-unet}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python $ROOT/tests/probe/unet_time.py 16384 5 > $OUT/${TAG}_time.log 2>&1
python - <<PY
import glob, sqlite3
f = sorted(glob.glob("$OUT/${TAG}_trace/*/*_results.db"))[-1]
con = sqlite3.connect(f)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open("$OUT/${TAG}_kernel_stats.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python tests/probe/unet_time.py 16384 5   name | calls | total_us | avg_us | pct\n")
    for r in rows[:45]:
        o.write(f"{r[0][:110]:110s} {r[1]:6d} {r[2] / 1e3:12.1f} {r[3] / 1e3:10.2f} {r[4]:6.2f}\n")
PY
tail -4 $OUT/${TAG}_time.log
