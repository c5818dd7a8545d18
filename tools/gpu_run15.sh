mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_view_a -- python3 tools/view_bench.py --iters 3 > /dev/null 2> gpurun_out/pmc_view_a.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_view_b -- python3 tools/view_bench.py --iters 3 > /dev/null 2> gpurun_out/pmc_view_b.err
tail -2 gpurun_out/pmc_view_a.err gpurun_out/pmc_view_b.err
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pmc_view_a","gpurun_out/pmc_view_b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mg_gen_obs" in r["Kernel_Name"]:
                key=r["Kernel_Name"].split("<")[1].split(">")[0]+" grid="+r["Grid_Size"]
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(d.split("_")[-1], k, {c: "%.3g" % (sum(x)/len(x)) for c,x in v.items()}, "calls", len(next(iter(v.values()))))
PY
