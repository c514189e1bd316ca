#!/usr/bin/env python3
"""PMC collection for one kernel of one command (run on the GPU box):

    python3 tools/pmc_collect.py TAG KERNEL_SUBSTRING[,SUBSTRING...] -- python3 bench.py ...

Runs the command under `rocprofv3 --kernel-trace --pmc ...` once per counter group (separate passes, as the microarch
guide prescribes; never together with other trace domains), averages every counter over the dispatches whose kernel name
contains one of the substrings, prints a table (-> profiles/r02_pmc_TAG.txt) and merges the figures bench.py needs into
profiles/counters.json under the key TAG:
  hbm_bytes       = 2 x FETCH_SIZE[KB] x 1024 + WRITE_SIZE[KB] x 1024   (gfx950: FETCH_SIZE reports half of a 16 B/lane stream)
  valu_insts      = SQ_INSTS_VALU (wave-instructions per dispatch)
  clock_ghz       = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
This parent process never touches the GPU."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("sq1", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
             "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU"]),
    ("sq2", ["SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU",
             "SQ_ACTIVE_INST_SCA", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_LDS"]),
    ("sq3", ["SQ_BUSY_CU_CYCLES", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_MISC",
             "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU"]),
    ("grbm", ["GRBM_GUI_ACTIVE"]),
]


def main():
    tag, subs = sys.argv[1], sys.argv[2].split(",")
    cmd = sys.argv[sys.argv.index("--") + 1:]
    groups = [g for g in GROUPS if not os.environ.get("PMC_GROUPS") or g[0] in os.environ["PMC_GROUPS"].split(",")]
    os.chdir(ROOT)
    os.makedirs("gpurun_out", exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    lines, means, durs = [], {}, {}
    for name, counters in groups:
        d = f"gpurun_out/pmc_{tag}_{name}"
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--"] + cmd,
                           env=env, stdout=open(f"{d}.log", "w"), stderr=subprocess.STDOUT)
        lines.append(f"== {name} rc={r.returncode}")
        print(lines[-1], flush=True)
        files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(files[0])):
            if not any(s in row["Kernel_Name"] for s in subs):
                continue
            a = acc[row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
        for k, (n, v) in acc.items():
            means[k] = v / n
            lines.append(f"{name:6s} {k:24s} dispatches={n:4d} mean={v / n:.6g}")
        kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
        if kt:
            dd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))
                  if any(s in r["Kernel_Name"] for s in subs)]
            if dd:
                durs[name] = sum(dd) / len(dd) / 1e3
                lines.append(f"{name:6s} kernel duration mean {durs[name]:.2f} us over {len(dd)} dispatches")
    entry = {"kernel": subs, "command": " ".join(cmd)}
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        entry["fetch_kb"], entry["write_kb"] = means["FETCH_SIZE"], means["WRITE_SIZE"]
        entry["hbm_bytes"] = (2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024
    if "SQ_INSTS_VALU" in means:
        entry["valu_insts"] = means["SQ_INSTS_VALU"]
        entry["duration_us_sq1_pass"] = durs.get("sq1")
    if "GRBM_GUI_ACTIVE" in means and "grbm" in durs:
        entry["clock_ghz"] = means["GRBM_GUI_ACTIVE"] / 8 / (durs["grbm"] * 1e3)
        entry["duration_us_grbm_pass"] = durs["grbm"]
    lines.append("summary " + json.dumps(entry))
    text = "\n".join(lines) + "\n"
    print(text)
    os.makedirs("gpurun_out/pmc_summaries", exist_ok=True)
    rnd = os.environ.get("PMC_ROUND", "r03")
    open(f"gpurun_out/pmc_summaries/{rnd}_pmc_{tag}.txt", "w").write(text)
    open(f"gpurun_out/pmc_summaries/{tag}.json", "w").write(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
