cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/calib_traffic.hip -o /tmp/calib_traffic || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/cal_$c -o cal -- /tmp/calib_traffic > /dev/null 2>&1
  db=$(find /tmp/cal_$c -name "*.db" | head -1)
  echo "== $c (KiB per dispatch, avg over 3 reps)"
  python3 - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for k, cn, a in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
    print("%-40s %-12s %12.1f KiB" % (k.split("(")[0][-40:], cn, a))
PY
done
/tmp/calib_traffic
