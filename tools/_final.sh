bash tools/run_checks_r05.sh
bash tools/run_profiles_r05.sh > gpurun_out/r05/profiles.log 2>&1
tail -3 gpurun_out/r05/profiles.log
