#!/usr/bin/env python3
"""gpurun_out/<run>/ (scratch) -> profiles/<prefix>_* (committed text summaries + traffic json).
Usage: python tools/make_profile_summaries.py gpurun_out/r2 r02"""
import json
import os
import shutil
import sqlite3
import subprocess
import sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
prof = os.path.join(root, "profiles")
summ = os.path.join(root, "tools", "rocprof_summary.py")


def summary(db, out, tail):
    if os.path.isfile(db):
        with open(os.path.join(prof, out), "w") as f:
            f.write(subprocess.check_output([sys.executable, summ, db, "--tail", str(tail)], text=True))


summary(os.path.join(src, "stats", "r_results.db"), prefix + "_kernel_stats.txt", 30)
summary(os.path.join(src, "stats_step", "r_results.db"), prefix + "_kernel_stats_step.txt", 1000)
summary(os.path.join(src, "stats_driver_like", "r_results.db"), prefix + "_kernel_stats_driver_like.txt", 8)
summary(os.path.join(src, "stats_cap64", "r_results.db"), prefix + "_kernel_stats_cap64.txt", 30)
summary(os.path.join(src, "stats_actor", "r_results.db"), prefix + "_kernel_stats_actor.txt", 30)
summary(os.path.join(src, "stats_actor_step", "r_results.db"), prefix + "_kernel_stats_actor_step.txt", 300)
summary(os.path.join(src, "stats_actor_lanes4", "r_results.db"), prefix + "_kernel_stats_actor_lanes4.txt", 12)
summary(os.path.join(src, "stats_lanes8", "r_results.db"), prefix + "_kernel_stats_lanes8.txt", 12)
summary(os.path.join(src, "stats_lanes4", "r_results.db"), prefix + "_kernel_stats_lanes4.txt", 4)
summary(os.path.join(src, "stats_chunked", "r_results.db"), prefix + "_kernel_stats_chunked.txt", 30)
for extra in ("phase_counters.txt", "pmc_ta_persist.txt", "persistent_trace.txt", "ab_launch_shapes.txt", "ab_launch_shapes_cap64.txt", "soak_random.txt", "soak_random_many.txt"):
    if os.path.isfile(os.path.join(src, extra)):
        shutil.copyfile(os.path.join(src, extra), os.path.join(prof, prefix + "_" + extra))
for m in ("persist", "persist_short", "rollout", "step"):
    summary(os.path.join(src, "fetch_" + m, "r_results.db"), prefix + "_pmc_fetch_%s.txt" % m, 30)
    summary(os.path.join(src, "write_" + m, "r_results.db"), prefix + "_pmc_write_%s.txt" % m, 30)
summary(os.path.join(src, "probe", "r_results.db"), prefix + "_pmc_probe_calibration.txt", 10)
for f in sorted(os.listdir(src)):
    if (f.startswith("bench_") and f.endswith(".json")) or f.startswith("phase_profile"):
        shutil.copyfile(os.path.join(src, f), os.path.join(prof, prefix + "_" + f))
with open(os.path.join(prof, prefix + "_resource_usage.txt"), "w") as f:
    f.write(subprocess.run([sys.executable, os.path.join(root, "tools", "resource_usage.py")], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True).stdout)


def last_avg(db, counter, kernel, n):
    cur = sqlite3.connect(db).cursor()
    d = [x[0] for x in cur.execute("select value from counters_collection where kernel_name like ? and counter_name=? "
                                   "order by start", ("%" + kernel + "%", counter))][-n:]
    return sum(d) / len(d)


import bench  # noqa: E402  (csrc fingerprint: a traffic figure only applies to the build it was measured on)


def parse_sq(path, field="mean/launch"):
    """{kernel prefix: {counter: value per launch}} from a tools/pmc_sq.sh summary (field: "mean/launch" or "last")"""
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = out.setdefault(line.strip(), {})
        elif cur is not None and "mean/launch" in line:
            f = line.split()
            cur[f[0]] = float(f[f.index(field) + 1]) if field in f else float(f[2])
    return out


def shader_clock(path, default=2.4):
    try:
        for line in open(path):
            if line.startswith("shader clock during the launch"):
                return float(line.split(":")[1].split("MHz")[0]) / 1000.0
    except OSError:
        pass
    return default


# ---- what binds the kernels: SQ instruction counts per wave and tick, LDS conflicts, wait share (bench.py: roofline.binding)
try:
    binding = {}
    clk = shader_clock(os.path.join(src, "phase_profile_rollout.txt"))
    # (tag, kernel name prefix, profile key, ticks of the counted launch, envs per launch, which launch)
    for tag, kkey, mode, tpl, envs, field in (
            # (the default command gathers BASELINE.md 3's tape by vehicle id: the IDT variants)
            # (round 6: the HOME build k_rollout<128, 5, ..> is what the queue form launches for 128 slots)
            ("persist", "void k_rollout<128, 5, false, false, false, true, true>", "persist", 100, 4096, "last"),
            ("persist_short", "void k_rollout<128, 5, false, false, false, true, true>", "persist_short", 20, 4096, "last"),
            ("rollout", "void k_rollout<128, 4, false, false, false, true, false>", "rollout", 25, 2048, "mean/launch"),
            ("step", "void k_tick<128>", "step", 1, 2048, "mean/launch"),
            ("actor", "void k_rollout<128, 4, false, true, false, false, true>", "actor_persist", 100, 4096, "last")):
        f = os.path.join(src, "pmc_sq_%s.txt" % tag)
        if not os.path.isfile(f):
            continue
        shutil.copyfile(f, os.path.join(prof, prefix + "_pmc_sq_%s.txt" % tag))
        sq = parse_sq(f, field)
        k = next((v for n, v in sq.items() if n.startswith(kkey[:58])), None)
        if not k:
            continue
        wt = envs * 2 * tpl                                   # waves x ticks of one launch (capacity 128: 2 waves per env)
        binding[mode] = dict(kernel=kkey.replace("void ", ""), csrc_sha=bench.csrc_sha(), envs_per_launch=envs, ticks_per_launch=tpl,
                             valu_per_wave_tick=k["SQ_INSTS_VALU"] / wt, salu_per_wave_tick=k["SQ_INSTS_SALU"] / wt,
                             lds_per_wave_tick=k["SQ_INSTS_LDS"] / wt, vmem_per_wave_tick=k["SQ_INSTS_VMEM"] / wt,
                             lds_bank_conflict_frac=k["SQ_LDS_BANK_CONFLICT"] / k["SQ_ACTIVE_INST_LDS"],
                             lds_bank_conflict_cycles_per_wave_tick=k["SQ_LDS_BANK_CONFLICT"] / wt,
                             lds_idx_active_cycles_per_wave_tick=k["SQ_LDS_IDX_ACTIVE"] / wt,
                             wait_frac=k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"],
                             valu_lane_utilisation=k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64.0) if k.get("SQ_ACTIVE_INST_VALU") else None,
                             shader_clock_ghz=clk, source=prefix + "_pmc_sq_%s.txt" % tag)
    json.dump(binding, open(os.path.join(prof, prefix + "_binding.json"), "w"), indent=1)
    print(json.dumps(binding, indent=1))
except Exception as e:  # noqa
    print("binding json not written:", e)
try:
    probe = last_avg(os.path.join(src, "probe", "r_results.db"), "FETCH_SIZE", "k_probe", 10)
    known_kib = 4096 * 128 * 72 / 1024.0
    corr = known_kib / probe
    out = {}
    for mode, kernel, bl, nlast in (("persist", "k_rollout", "bench_pmc_persist.json", 1), ("persist_short", "k_rollout", "bench_pmc_persist_short.json", 1),
                                    ("rollout", "k_rollout", "bench_pmc_rollout.json", 3), ("step", "k_tick", "bench_pmc_step.json", 60)):
        if not os.path.isfile(os.path.join(src, "fetch_" + mode, "r_results.db")) or not os.path.isfile(os.path.join(src, bl)):
            continue
        fetch = last_avg(os.path.join(src, "fetch_" + mode, "r_results.db"), "FETCH_SIZE", kernel, nlast)
        write = last_avg(os.path.join(src, "write_" + mode, "r_results.db"), "WRITE_SIZE", kernel, nlast)
        b = json.load(open(os.path.join(src, bl)))
        envs_per_launch = int(b["roofline"].get("envs_per_launch", 4096))
        tpl = int(b["config"].get("ticks_per_launch", 1))
        out[mode] = dict(
            kernel=kernel + "<128>", mode=mode, envs_per_launch=envs_per_launch, ticks_per_launch=tpl, csrc_sha=bench.csrc_sha(),
            workload="%d envs x 128 slots x %d tick(s) per launch, default bench outputs, steady state (last launch(es) of the run)"
                     % (envs_per_launch, tpl),
            fetch_size_kib_reported=fetch, write_size_kib_reported=write,
            fetch_calibration=dict(kernel="k_probe<128>", known_kib=known_kib, reported_kib=probe, correction=corr,
                                   note="same 8 B / 4 B per-lane SoA load pattern as the tick's load phase; gfx950 FETCH_SIZE "
                                        "counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section)"),
            hbm_bytes_per_launch=(fetch * corr + write) * 1024.0,
            hbm_bytes_per_tick=(fetch * corr + write) * 1024.0 / tpl)
    json.dump(out, open(os.path.join(prof, prefix + "_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
except Exception as e:  # noqa
    print("traffic json not written:", e)
