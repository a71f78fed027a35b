#!/usr/bin/env python3
"""gpurun_out/<run>/ (scratch) -> profiles/<prefix>_* (committed text summaries + traffic json).
Usage: python tools/make_profile_summaries.py gpurun_out/r1 r01"""
import json
import os
import shutil
import sqlite3
import subprocess
import sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
prof = os.path.join(root, "profiles")
summ = os.path.join(root, "tools", "rocprof_summary.py")


def summary(db, out, tail):
    if os.path.isfile(db):
        with open(os.path.join(prof, out), "w") as f:
            f.write(subprocess.check_output([sys.executable, summ, db, "--tail", str(tail)], text=True))


summary(os.path.join(src, "stats", "r_results.db"), prefix + "_kernel_stats.txt", 1000)
summary(os.path.join(src, "stats_single", "r_results.db"), prefix + "_kernel_stats_single_launch.txt", 1000)
summary(os.path.join(src, "stats_actor", "r_results.db"), prefix + "_kernel_stats_actor.txt", 300)
summary(os.path.join(src, "stats_lanes8", "r_results.db"), prefix + "_kernel_stats_lanes8.txt", 300)
summary(os.path.join(src, "fetch", "r_results.db"), prefix + "_pmc_fetch.txt", 30)
summary(os.path.join(src, "write", "r_results.db"), prefix + "_pmc_write.txt", 30)
summary(os.path.join(src, "probe", "r_results.db"), prefix + "_pmc_probe_calibration.txt", 10)
for f in ("bench_default.json", "bench_single_launch.json", "bench_cap64.json", "bench_actor.json", "phase_profile.txt", "bench_lanes8.json",
          "bench_lanes4.json"):
    p = os.path.join(src, f)
    if os.path.isfile(p):
        shutil.copyfile(p, os.path.join(prof, prefix + "_" + f))


def last_avg(db, counter, kernel, n):
    cur = sqlite3.connect(db).cursor()
    d = [x[0] for x in cur.execute("select value from counters_collection where kernel_name like ? and counter_name=? "
                                   "order by start", ("%" + kernel + "%", counter))][-n:]
    return sum(d) / len(d)


try:
    probe = last_avg(os.path.join(src, "probe", "r_results.db"), "FETCH_SIZE", "k_probe", 10)
    known_kib = 4096 * 128 * 72 / 1024.0
    fetch = last_avg(os.path.join(src, "fetch", "r_results.db"), "FETCH_SIZE", "k_tick", 30)
    write = last_avg(os.path.join(src, "write", "r_results.db"), "WRITE_SIZE", "k_tick", 30)
    corr = known_kib / probe
    try:
        bl = json.load(open(os.path.join(src, "bench_default.json")))
        envs_per_launch = int(bl["roofline"].get("envs_per_launch", 4096))
    except Exception:  # noqa
        envs_per_launch = 4096
    traffic = dict(
        kernel="k_tick<128>", envs_per_launch=envs_per_launch,
        workload="%d envs x 128 slots per launch, default bench outputs, steady state (last 30 launches)" % envs_per_launch,
        fetch_size_kib_reported=fetch, write_size_kib_reported=write,
        fetch_calibration=dict(kernel="k_probe<128>", known_kib=known_kib, reported_kib=probe, correction=corr,
                               note="same 8 B / 4 B per-lane SoA load pattern as the tick's load phase; gfx950 FETCH_SIZE "
                                    "counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section)"),
        hbm_bytes_per_launch=(fetch * corr + write) * 1024.0)
    json.dump(traffic, open(os.path.join(prof, prefix + "_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
except Exception as e:  # noqa
    print("traffic json not written:", e)
