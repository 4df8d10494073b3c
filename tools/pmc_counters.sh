#!/bin/bash
# SQ / LDS / TCP counters of both passes for the current build (GPU box, via gpurun): outputs gpurun_out/$1/pmc_summary.txt
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r02pmc}; rm -rf $out; mkdir -p $out
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-extras}"
PROG="${PMC_PROG:-$R/bench.py $ARGS}"      # PMC_PROG="$GRAFT_REPO_ROOT/tools/wide_bench.py": the counters of another workload
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $PROG > $out/p$i.log 2>&1 < /dev/null
done
cd $R && python3 - "$out" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
with open(out+"/pmc_summary.txt","w") as fo:
  for f in sorted(glob.glob(out+"/p*/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("void ","").split("(")[0][:16], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        if k[0].startswith(("k_total","k_accum")): print("%-10s %-32s %.6g"%(k[0],k[1],sum(v)/len(v)), file=fo)
PY
cat $out/pmc_summary.txt
