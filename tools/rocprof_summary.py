#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db) as text: per-kernel call count / avg / min /
max / total duration (the --stats view) and, when counters were collected (--pmc), per-kernel
counter averages.  Usage: tools/rocprof_summary.py <results.db> [--tail N] > profiles/<name>.txt"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    tail = int(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 0
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# source: %s" % path)
    print("## kernel stats (ns)")
    print("%-70s %8s %12s %10s %10s %14s %6s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "%"))
    rows = list(cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                            "from kernels group by name order by 6 desc"))
    tot = sum(r[5] for r in rows) or 1
    for r in rows:
        print("%-70s %8d %12.1f %10d %10d %14d %6.2f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100.0 * r[5] / tot))
    if tail:
        print("## steady state: last %d dispatches of each k_tick / k_rollout / k_actor kernel" % tail)
        for (name,) in list(cur.execute("select distinct name from kernels where name like '%k_tick%' or name like '%k_rollout%' "
                                        "or name like '%k_actor%'")):
            d = [x[0] for x in cur.execute("select end-start from kernels where name=? order by start", (name,))][-tail:]
            print("%-70s n=%d avg_ns=%.1f min=%d max=%d" % (name[:70], len(d), sum(d) / len(d), min(d), max(d)))
        # a persistent launch (k_rollout<.., PERS = true>) is ONE dispatch per call: the run's earlier dispatches are the prefill /
        # warm-up calls (other lengths); the LAST one is the timed --steps ticks = bench.py's roofline.launch_ms
        print("## last dispatch of each k_tick / k_rollout / k_actor kernel (a persistent launch: the timed call)")
        for (name,) in list(cur.execute("select distinct name from kernels where name like '%k_tick%' or name like '%k_rollout%' "
                                        "or name like '%k_actor%'")):
            d = [x[0] for x in cur.execute("select end-start from kernels where name=? order by start", (name,))]
            print("%-70s last_ns=%d (dispatch %d of %d)" % (name[:70], d[-1], len(d), len(d)))
    for r in cur.execute("select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x "
                         "from kernels where name like '%k_tick%' or name like '%k_rollout%' or name like '%k_actor%'"):
        # rocprofv3's vgpr_count field is NOT the per-lane allocation the occupancy follows (it reports 48 for a kernel the
        # compiler allocates 96 registers to); the compiler's numbers are in profiles/rNN_resource_usage.txt
        print("## dispatch record %s: rocprofv3 vgpr_count field=%s accum=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s "
              "(compiler allocation: see the resource_usage file)" % ((r[0][:40],) + tuple(r[1:])))
    try:
        rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                                "from counters_collection group by kernel_name, counter_name order by 1, 2"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("## counters (per dispatch)")
        print("%-60s %-24s %7s %16s %16s %16s" % ("kernel", "counter", "n", "avg", "min", "max"))
        for r in rows:
            print("%-60s %-24s %7d %16.3f %16.3f %16.3f" % (r[0][:60], r[1], r[2], r[3], r[4], r[5]))
        if tail:
            print("## counters, last %d dispatches per kernel family" % tail)
            for (cn,) in list(cur.execute("select distinct counter_name from counters_collection")):
                for kn in ("k_tick", "k_rollout", "k_actor"):
                    d = [x[0] for x in cur.execute("select value from counters_collection where kernel_name like ? "
                                                   "and counter_name=? order by start", ("%" + kn + "%", cn))][-tail:]
                    if d:
                        print("%-12s %-24s n=%d avg=%.3f" % (kn, cn, len(d), sum(d) / len(d)))


if __name__ == "__main__":
    main()
