R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r05prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
name=cfg2_prob
rm -rf $out/$name; mkdir -p $out/$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras --prob > $out/$name/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --prob > $out/$name/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --prob > $out/$name/write.log 2>&1
tail -1 $out/$name/stats.log | cut -c1-300
