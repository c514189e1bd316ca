#!/bin/bash
# PMC collection for the tiled kernel (development aid).  usage: bash tools/pmc_k1.sh <tag> <L> ; env TSU_TILE_VARIANT / TSU_K1_DEBUG pass through
tag=${1:-run}; L=${2:-8192}; K=${3:-4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python bench.py --L $L --sweeps-per-launch $K --sweeps-per-step ${SPS:-16} --steps 3 --warmup 1 --ramp-steps 2 --no-cpu-baseline --no-extra > gpurun_out/pmc_${tag}_$name.log 2>&1; echo "== $name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS
run sq3 SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run grbm GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for name in ("fetch","write","sq1","sq2","sq3","grbm"):
    files = glob.glob(f"gpurun_out/pmc_${tag}_{name}/**/*counter_collection.csv", recursive=True)
    if not files: print(name, "no file"); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(files[0])):
        if "k1_tiled" not in row["Kernel_Name"] and "k1_resident" not in row["Kernel_Name"]: continue
        k = row["Counter_Name"]; acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    for k, (n, v) in acc.items(): print(f"{name:6s} {k:24s} dispatches={n:4d} mean={v/n:.6g}")
    kt = glob.glob(f"gpurun_out/pmc_${tag}_{name}/**/*kernel_trace.csv", recursive=True)
    if kt and name == "sq1":
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "k1_tiled" in r["Kernel_Name"] or "k1_resident" in r["Kernel_Name"]]
        print(f"kernel duration mean {sum(d)/len(d)/1e3:.2f} us over {len(d)} dispatches")
PY
