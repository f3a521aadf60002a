cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL"; do
  n=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds_$n -- python3 $R/tools/quick_ir_bench.py 1000 > $R/gpurun_out/pmc_lds_$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_lds_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0,0.0])
    for row in csv.DictReader(open(f)):
        k=(row["Kernel_Name"][:40], row["Counter_Name"])
        acc[k][0]+=1; acc[k][1]+=float(row["Counter_Value"])
    for k,v in acc.items():
        if "k_robot_sweep" in k[0]: print(k, "dispatches", v[0], "per dispatch %.3e" % (v[1]/v[0]))
PY
