cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_final
bash profiles/run_profile.sh r06_final > gpurun_out/r06_final_run.log 2>&1
tail -8 gpurun_out/r06_final_run.log | cut -c1-700
