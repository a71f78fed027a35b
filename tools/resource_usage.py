#!/usr/bin/env python3
"""Compiler view of every kernel of libpveenv.so (hipcc -Rpass-analysis=kernel-resource-usage) as one table:
VGPRs, AGPRs, spills, scratch, LDS, occupancy (waves per SIMD).  Usage: python tools/resource_usage.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "pve-mcc_for_unsignalized_intersection_amd", "csrc")


def collect():
    out = subprocess.run(["make", "-C", CSRC, "-s", "-B", "resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[a-zA-Z/]+\])?):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
        if "warning:" in line:
            print(line.strip(), file=sys.stderr)
    return rows


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%-62s %5s %5s %6s %6s %8s %7s %5s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "LDS", "occ"))
    for r in collect():
        if flt in r["name"]:
            print("%-62s %5d %5d %6d %6d %8d %7d %5d" % (r["name"][:62], r.get("VGPRs", -1), r.get("AGPRs", -1),
                  r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1),
                  r.get("LDS Size [bytes/block]", -1), r.get("Occupancy [waves/SIMD]", -1)))
