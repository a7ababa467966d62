#!/usr/bin/env python3
"""Copy the PMC summaries tools/pmc_traffic.py left under gpurun_out/ into profiles/ (tracked), adding the git head they
were measured at. usage: python tools/stamp_profile.py gpurun_out/pmc r02"""
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], sys.argv[2]
head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "gpu-raytracer_amd/csrc", "include", "bench.py"], cwd=REPO, capture_output=True, text=True).stdout.strip())
for f in sorted(glob.glob(os.path.join(src, "pmc_*.json"))):
    j = json.load(open(f))
    j["git_head"] = head + ("+uncommitted" if dirty else "")
    name = f"{rnd}_" + os.path.basename(f)
    json.dump(j, open(os.path.join(REPO, "profiles", name), "w"), indent=1)
    txt = f[:-5] + ".txt"
    if os.path.exists(txt):
        shutil.copy(txt, os.path.join(REPO, "profiles", name[:-5] + ".txt"))
    print("profiles/" + name, j["source_sha"], j["git_head"])
