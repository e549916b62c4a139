#!/bin/bash
# HBM counter calibration on known byte counts (tools/microbench/ubench_fetch.hip): plain run for GB/s, then one
# rocprofv3 --pmc pass per counter group.  usage (through gpurun): bash tools/fetch_calibration.sh <tag>
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetchcal
rm -rf $O && mkdir -p $O
cd $R
B=tools/microbench/ubench_fetch
$B > $O/plain.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/p1 -- $B > $O/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p2 -- $B > $O/p2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum -d $O/p3 -- $B > $O/p3.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum -d $O/p4 -- $B > $O/p4.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_MISS_sum -d $O/p5 -- $B > $O/p5.log 2>&1 || true
python3 - "$O" > $R/gpurun_out/${TAG}_fetch_calibration.txt <<'PY'
import collections, glob, os, sqlite3, sys
root = sys.argv[1]
print(open(os.path.join(root, "plain.txt")).read())
c = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    con = sqlite3.connect(db)
    try:
        for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
            if "sweep" in name:
                c[name][counter].append(float(value))
    except sqlite3.Error:
        pass
true = float(2 << 30)
pat = ["coalesced", "lane64", "lane64u", "lane16", "quad64", "oct128"]
print("# counters: mean per launch (3 launches each); ratios are counter bytes / true bytes (2 GiB, every byte touched once)")
for k in sorted(c):
    a = k[k.index("<") + 1:k.index(">")].split(",")
    name = "%-10s %s" % (pat[int(a[0])], "write" if "true" in a[1] else "read ")
    e = {n: sum(v) / len(v) for n, v in c[k].items()}
    parts = []
    if "FETCH_SIZE" in e: parts.append("FETCH_SIZE %.3fx" % (e["FETCH_SIZE"] * 1024 / true))
    if "WRITE_SIZE" in e: parts.append("WRITE_SIZE %.3fx" % (e["WRITE_SIZE"] * 1024 / true))
    for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum",
              "TCC_EA0_WRREQ_DRAM_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCC_REQ_sum", "TCC_MISS_sum"):
        if n in e: parts.append("%s %.4f per 64 B" % (n.replace("_sum", ""), e[n] * 64 / true))
    print(name, "; ".join(parts))
PY
cat $R/gpurun_out/${TAG}_fetch_calibration.txt
find $O -name "*.db" -delete
