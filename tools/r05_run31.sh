cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05fin
bash tools/r05_profiles.sh > gpurun_out/r05fin/profiles.log 2>&1
tail -3 gpurun_out/r05fin/profiles.log | cut -c1-400
